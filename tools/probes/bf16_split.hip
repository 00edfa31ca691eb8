// fp32 products on the bf16 matrix pipe: x = x1 + x2 + x3 with three bf16 terms (8 + 8 + 8 mantissa bits: exact for normal fp32 values), the
// product of two such sums as bf16 x bf16 MFMAs with fp32 accumulation.  `v_mfma_f32_16x16x32_bf16` issues every ~17 cycles per SIMD for a
// 16 x 16 x 32 slab, `v_mfma_f32_16x16x4_f32` every 32 cycles for a 16 x 16 x 4 one: a K = 32 slab costs 256 cycles natively and 17 per term
// kept.  Which terms:  x1 w1 (1), x1 w2, x2 w1 (2^-8), x2 w2, x1 w3, x3 w1 (2^-16) -- six terms, everything dropped is <= 2^-24 of the
// product, the size of an fp32 rounding -- or all nine.  This probe measures both things that decide whether the Regulation / attention
// products should move there: the issue rate of the six- and nine-term groups against the native instruction (one and two waves per SIMD),
// and the error of [16 x K] . [K x 16] products against fp64 for native fp32, 3, 6 and 9 terms (K = 128 and 1024, N(0, 1) operands).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/bf16_split.hip -o build/bf16_split && build/bf16_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }      // round to nearest even
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
    a = to_bf16(x);
    const float r1 = x - (float)a;
    b = to_bf16(r1);
    const float r2 = r1 - (float)b;
    c = to_bf16(r2);
}

// C[16 x 16] = A[16 x K] . B[K x 16], one wave.  A row-major [16][K], B row-major [K][16].
// TERMS: 0 = native fp32 MFMA, 3 / 6 / 9 = bf16 terms kept.
template <int TERMS>
__global__ void k_product(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (TERMS == 0) {
        for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k0 + q], B[(k0 + q) * 16 + r], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            bf16x8 a[3], b[3];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 t0, t1, t2;
                split3(A[r * K + k0 + 8 * q + j], t0, t1, t2);
                a[0][j] = t0, a[1][j] = t1, a[2][j] = t2;
                split3(B[(k0 + 8 * q + j) * 16 + r], t0, t1, t2);
                b[0][j] = t0, b[1][j] = t1, b[2][j] = t2;
            }
            // smallest terms first: their sum is formed before it meets the large one
            if (TERMS >= 9) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[1], acc, 0, 0, 0);
            }
            if (TERMS >= 6) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) C[(4 * q + i) * 16 + r] = acc[i];
}

// issue rate: `groups` x 4 K = 32 slabs per wave, four independent accumulators (static registers), operands fixed in registers
template <int TERMS>
__device__ __forceinline__ void slab(f32x4& c, const bf16x8 (&a)[3], const bf16x8 (&b)[3], float fa, float fb) {
    if (TERMS == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c, 0, 0, 0);
    } else {
        if (TERMS >= 9) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[1], c, 0, 0, 0);
        }
        if (TERMS >= 6) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
    }
}
// INTERLEAVE: the terms of four slabs issued round-robin over the four accumulators (no two consecutive MFMAs on one accumulator)
template <int TERMS>
__global__ void k_rate(float* out, unsigned long long* ticks, int groups) {
    const int lane = threadIdx.x & 63;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    bf16x8 a[3], b[3];
    for (int t = 0; t < 3; ++t)
        for (int j = 0; j < 8; ++j) a[t][j] = (__bf16)(0.001f * (lane + j + t)), b[t][j] = (__bf16)(0.002f * (lane - j + t));
    const float fa = 0.001f * lane, fb = 0.002f * lane;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int g = 0; g < groups; ++g) {
        slab<TERMS>(c0, a, b, fa, fb);
        slab<TERMS>(c1, a, b, fa, fb);
        slab<TERMS>(c2, a, b, fa, fb);
        slab<TERMS>(c3, a, b, fa, fb);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const f32x4 s4 = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s4[0] + s4[1] + s4[2] + s4[3];
    if (lane == 0) ticks[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            printf("%s: %s\n", #x, hipGetErrorString(e_));        \
            return 1;                                             \
        }                                                         \
    } while (0)

template <int TERMS>
static int accuracy(int K, const std::vector<float>& A, const std::vector<float>& B, const std::vector<double>& ref, double scale) {
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4));
    CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dC, 256 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_product<TERMS>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    std::vector<float> C(256);
    CK(hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost));
    double worst = 0, rms = 0;
    for (int i = 0; i < 256; ++i) {
        const double e = std::fabs((double)C[i] - ref[i]);
        worst = std::max(worst, e);
        rms += e * e;
    }
    printf("    %-22s max |err| %.3e  rms %.3e   (in units of sqrt(K) 2^-24: %.2f / %.2f)\n",
           TERMS == 0 ? "native fp32 MFMA" : TERMS == 3 ? "3 bf16 terms" : TERMS == 6 ? "6 bf16 terms" : "9 bf16 terms", worst, std::sqrt(rms / 256), worst / scale,
           std::sqrt(rms / 256) / scale);
    (void)hipFree(dA), (void)hipFree(dB), (void)hipFree(dC);
    return 0;
}

template <int TERMS>
static int rate(int waves_per_simd) {
    const int groups = 1024, threads = 256 * waves_per_simd;      // one workgroup on one CU: 4 SIMDs x waves_per_simd; 4 slabs per group
    float* out;
    unsigned long long* ticks;
    CK(hipMalloc(&out, threads * 4));
    CK(hipMalloc(&ticks, 64 * 8));
    hipLaunchKernelGGL(k_rate<TERMS>, dim3(1), dim3(threads), 0, 0, out, ticks, groups);
    hipLaunchKernelGGL(k_rate<TERMS>, dim3(1), dim3(threads), 0, 0, out, ticks, groups);
    unsigned long long t[16];
    CK(hipMemcpy(t, ticks, (threads / 64) * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (int i = 0; i < threads / 64; ++i) mx = std::max(mx, (double)t[i]);
    printf("    %-22s %d wave(s) per SIMD: %7.1f ticks per K = 32 slab and wave, %7.1f per slab and SIMD\n",
           TERMS == 0 ? "native fp32 MFMA (x8)" : TERMS == 3 ? "3 bf16 terms" : TERMS == 6 ? "6 bf16 terms" : "9 bf16 terms", waves_per_simd, mx / (4.0 * groups),
           mx / (4.0 * groups) / waves_per_simd);
    (void)hipFree(out), (void)hipFree(ticks);
    return 0;
}

int main() {
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int K : {128, 1024}) {
        std::vector<float> A(16 * K), B(K * 16);
        for (auto& v : A) v = nd(gen);
        for (auto& v : B) v = nd(gen);
        std::vector<double> ref(256, 0.0);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 16 + j];
                ref[i * 16 + j] = s;
            }
        printf("[16 x %d] . [%d x 16], N(0, 1) operands, against fp64:\n", K, K);
        const double scale = std::sqrt((double)K) * std::ldexp(1.0, -24);
        if (accuracy<0>(K, A, B, ref, scale) || accuracy<3>(K, A, B, ref, scale) || accuracy<6>(K, A, B, ref, scale) || accuracy<9>(K, A, B, ref, scale)) return 1;
    }
    printf("issue rate on one CU (s_memtime ticks = shader cycles):\n");
    for (int w : {1, 2})
        if (rate<0>(w) || rate<3>(w) || rate<6>(w) || rate<9>(w)) return 1;
    return 0;
}

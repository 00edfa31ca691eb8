// Skinny products [ROWS x K] . [K x N] with ROWS << 16 (the Regulation stack's 9 tokens, the Pairwise chains' 8 pairs, the
// Embedding layer's single row): the vector ALUs with the row operand in SGPRs against the 16-row f32 matrix-core tile.
//   VALU:  lane = output column(s); the weights stream from L2 as 16-byte loads in a [K/4][N][4] layout (unit stride across lanes);
//          x[k][m] comes through the scalar cache (s_load) and is a scalar operand of v_fmac_f32: ROWS FMAs per weight element
//   MFMA:  v_mfma_f32_16x16x4_f32 on 16 x 16 fragment-order tiles (what the kernels of csrc/ do), 16 rows whatever ROWS is
// One workgroup of eight waves per CU (as the chain kernels), N = 1024 columns, K = 128, `layers` different weight matrices.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rows.hip -o build/valu_rows && build/valu_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* cptr;
typedef const __attribute__((address_space(1))) f32x4* g4ptr;

constexpr int K = 128, N = 1024;

template <int ROWS>
__global__ __launch_bounds__(512) void k_valu(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ out,
                                              unsigned long long* ticks, int layers) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n0 = wave * 128 + lane, n1 = n0 + 64;
    float acc0[ROWS], acc1[ROWS];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) acc0[m] = acc1[m] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < layers; ++l) {
        g4ptr Wl = (g4ptr)(W + (size_t)l * K * N);
        cptr Xl = (cptr)(X + ((size_t)blockIdx.x * layers + l) * K * ROWS);
        // weight ring: 4 k-quads ahead
        f32x4 r0[4], r1[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            r0[p] = Wl[(size_t)p * N + n0];
            r1[p] = Wl[(size_t)p * N + n1];
        }
#pragma unroll 4
        for (int k4 = 0; k4 < K / 4; ++k4) {
            const f32x4 w0 = r0[k4 & 3], w1 = r1[k4 & 3];
            r0[k4 & 3] = Wl[(size_t)(k4 + 4) * N + n0];      // (runs into the next matrix / the padding behind the last: harmless)
            r1[k4 & 3] = Wl[(size_t)(k4 + 4) * N + n1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int m = 0; m < ROWS; ++m) {
                    const float x = Xl[(k4 * 4 + i) * ROWS + m];
                    acc0[m] = fmaf(x, w0[i], acc0[m]);
                    acc1[m] = fmaf(x, w1[i], acc1[m]);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
        out[((size_t)blockIdx.x * ROWS + m) * N + n0] = acc0[m];
        out[((size_t)blockIdx.x * ROWS + m) * N + n1] = acc1[m];
    }
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

// the matrix-core form: wave = 128 columns = 8 N-tiles, A (16 rows x K) from LDS, B tiles in fragment order: tile (nt, k16) is 256
// floats, lane (n = lane & 15, kq = lane >> 4) reads float4 = W[k16 * 16 + kq * 4 .. + 3][nt * 16 + n]
__global__ __launch_bounds__(512) void k_mfma(const float* __restrict__ Wt, const float* __restrict__ X16, float* __restrict__ out,
                                              unsigned long long* ticks, int layers) {
    __shared__ float xs[16][K + 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < 16 * K; i += 512) xs[i / K][i % K] = X16[(size_t)blockIdx.x * 16 * K + i];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < layers; ++l) {
        g4ptr Wl = (g4ptr)(Wt + (size_t)l * K * N);
        // tile index: (nt * (K / 16) + k16) * 64 + lane   (float4 units)
        f32x4 ring[2][8];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int t = 0; t < 8; ++t) ring[p][t] = Wl[((size_t)(wave * 8 + t) * (K / 16) + p) * 64 + lane];
#pragma unroll 2
        for (int k16 = 0; k16 < K / 16; ++k16) {
            f32x4 b[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) b[t] = ring[k16 & 1][t];
#pragma unroll
            for (int t = 0; t < 8; ++t) ring[k16 & 1][t] = Wl[((size_t)(wave * 8 + t) * (K / 16) + k16 + 2) * 64 + lane];
            const f32x4 a = *reinterpret_cast<const f32x4*>(&xs[lane & 15][k16 * 16 + (lane >> 4) * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[t][i], acc[t], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[((size_t)blockIdx.x * 16 + (lane >> 4) * 4 + i) * N + (wave * 8 + t) * 16 + (lane & 15)] = acc[t][i];
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

static void report(const char* what, unsigned long long* dticks, int nwg, int layers, float ms) {
    std::vector<unsigned long long> h(nwg * 8);
    hipMemcpy(h.data(), dticks, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (auto v : h) {
        mean += (double)v;
        if ((double)v > mx) mx = (double)v;
    }
    mean /= h.size();
    printf("%-28s %4d workgroups  %8.0f ticks per product (mean), %8.0f (slowest wave)   launch %.1f us\n", what, nwg, mean / layers, mx / layers, ms * 1e3);
}

int main() {
    const int layers = 12, nwg_max = 256;
    float *W, *X, *out;
    unsigned long long* ticks;
    hipMalloc(&W, (size_t)(layers + 1) * K * N * 4);
    hipMalloc(&X, (size_t)nwg_max * layers * K * 16 * 4);
    hipMalloc(&out, (size_t)nwg_max * 16 * N * 4);
    hipMalloc(&ticks, nwg_max * 8 * 8);
    std::vector<float> h((size_t)layers * K * N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hx((size_t)nwg_max * layers * K * 16);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 40503u) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms;
#define RUN(name, kern, nwg)                                                       \
    for (int rep = 0; rep < 2; ++rep) {                                            \
        hipEventRecord(e0);                                                        \
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 0, 0, W, X, out, ticks, layers); \
        hipEventRecord(e1);                                                        \
        hipDeviceSynchronize();                                                    \
        hipEventElapsedTime(&ms, e0, e1);                                          \
    }                                                                              \
    report(name, ticks, nwg, layers, ms);
    for (int nwg : {1, 192}) {
        RUN("mfma 16x16x4 (16-row tile)", k_mfma, nwg);
        RUN("valu  1 row", k_valu<1>, nwg);
        RUN("valu  8 rows", k_valu<8>, nwg);
        RUN("valu  9 rows", k_valu<9>, nwg);
        RUN("valu 12 rows", k_valu<12>, nwg);
    }
    return 0;
}

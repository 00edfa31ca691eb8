// What would the weight products of a Regulation layer cost on the bf16 matrix pipe?  A skeleton of k_reg8_fwd's per-layer work --
// 192 workgroups of eight waves, each wave streaming its own 112 KB (fp32) of tiled weights per layer through an 8-unit register
// ring that runs 7 units ahead across product, barrier and layer boundaries; four products (8 column tiles x K = 128, 1 x 256,
// 2 x 128, 1 x 256) with a workgroup barrier and a small LDS epilogue behind each -- in three forms:
//   MODE 0  native:  v_mfma_f32_16x16x4_f32, unit = two 1 KB blocks, 8 instructions (what the kernels do today)
//   MODE 1  six bf16 terms, weights PRE-SPLIT into three bf16 planes (6 bytes per element): unit = three 1 KB blocks, 6 x
//           v_mfma_f32_16x16x32_bf16; the A operand as three bf16 planes in LDS
//   MODE 2  six bf16 terms, weights fp32 in memory (4 bytes per element), split in registers (~44 VALU operations per unit)
// No numerics here (tools/probes/bf16_split.hip has the error figures); only the instruction mix and the memory pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/reg_layer_stream.hip -o build/reg_layer_stream && build/reg_layer_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ float4 ldg4(const char* p) {
    const f32x4 v = *(const GLOBAL f32x4*)(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ bf16x8 as_bf(const float4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mf32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mbf(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    const uint32_t u0 = __builtin_bit_cast(uint32_t, x0), u1 = __builtin_bit_cast(uint32_t, x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const uint32_t v0 = __builtin_bit_cast(uint32_t, r0), v1 = __builtin_bit_cast(uint32_t, r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __builtin_bit_cast(float, v0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, s1), __builtin_bit_cast(uint32_t, s0), 0x07060302u);
}
struct B3 {
    bf16x8 h, m, l;
};
__device__ __forceinline__ B3 split8(const float4& a, const float4& b) {
    uint32_t h[4], m[4], l[4];
    split_pair(a.x, a.y, h[0], m[0], l[0]);
    split_pair(a.z, a.w, h[1], m[1], l[1]);
    split_pair(b.x, b.y, h[2], m[2], l[2]);
    split_pair(b.z, b.w, h[3], m[3], l[3]);
    B3 o;
    o.h = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
    o.m = __builtin_bit_cast(bf16x8, make_uint4(m[0], m[1], m[2], m[3]));
    o.l = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    return o;
}
// six terms, smallest first, on two accumulators
__device__ __forceinline__ void six(const B3& a, const B3& b, f32x4& c0, f32x4& c1) {
    c0 = mbf(a.l, b.h, c0);
    c1 = mbf(a.h, b.l, c1);
    c0 = mbf(a.m, b.m, c0);
    c1 = mbf(a.m, b.h, c1);
    c0 = mbf(a.h, b.m, c0);
    c1 = mbf(a.h, b.h, c1);
}

constexpr int kUnits = 56;      // per wave and layer: 32 + 8 + 8 + 8

template <int MODE, int PRE>
__global__ __launch_bounds__(512) void k_layer(const char* __restrict__ W, int layers, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = (MODE == 1 || MODE >= 3) ? 3 : 2, UB = NB * 1024, R = 8;      // (mixed modes: every unit gets a 3 KB slot, native units use 2 KB of it)
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    // A operand: fp32 tile [16][260] (MODE 0) or three bf16 planes [16][264] (272 / 528-byte rows: conflict-free 16-byte reads)
    float* af = reinterpret_cast<float*>(smem);
    char* ab = smem;
    constexpr int PLANE = 16 * 528;
    for (int i = tid; i < 16 * 260; i += 512) af[i] = 0.001f * (i % 97);
    __syncthreads();
    const char* wp = W + ((size_t)w * kUnits) * UB + lane * 16;
    const size_t lstride = (size_t)8 * kUnits * UB;
    float4 ring[R][NB];
    auto req = [&](int slot, const char* p) {
#pragma unroll
        for (int j = 0; j < NB; ++j) ring[slot][j] = ldg4(p + j * 1024);
    };
    auto req2 = [&](int slot, const char* p) {
#pragma unroll
        for (int j = 0; j < 2; ++j) ring[slot][j] = ldg4(p + j * 1024);
    };
    auto is_bf = [](int u) { return MODE == 1 || MODE == 2 || (MODE == 3 && u < 32) || (MODE == 4 && u >= 32); };
    auto reqm = [&](int slot, const char* p, int u) {
        if (MODE >= 3 && !is_bf(u))
            req2(slot, p);
        else
            req(slot, p);
    };
#pragma unroll
    for (int u = 0; u < PRE; ++u) reqm(u, wp + (size_t)u * UB, u);
    f32x4 acc[8][2];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t][0] = acc[t][1] = f32x4{0, 0, 0, 0};
    float keep = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < layers; ++l) {
        const char* lp = wp + (size_t)l * lstride;
        const bool more = l + 1 < layers;
        B3 a1[4];      // A of the first product: all four K = 32 slabs stay in registers for its eight column tiles
        if (is_bf(0)) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a1[s].h = as_bf(*reinterpret_cast<const float4*>(ab + r * 528 + s * 64 + q * 16));
                a1[s].m = as_bf(*reinterpret_cast<const float4*>(ab + PLANE + r * 528 + s * 64 + q * 16));
                a1[s].l = as_bf(*reinterpret_cast<const float4*>(ab + 2 * PLANE + r * 528 + s * 64 + q * 16));
            }
        }
        auto body = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (u + PRE < kUnits)
                reqm((u + PRE) % R, lp + (size_t)(u + PRE) * UB, u + PRE);
            else if (more)
                reqm((u + PRE) % R, lp + lstride + (size_t)(u + PRE - kUnits) * UB, u + PRE - kUnits);
            __builtin_amdgcn_sched_barrier(0);
            // which tile / slab this unit is
            constexpr int prod = u < 32 ? 0 : (u - 32) / 8 + 1, v = u < 32 ? u : (u - 32) % 8;
            constexpr int t = prod == 0 ? v / 4 : (prod == 2 ? v / 4 : 0);
            if (!is_bf(u)) {
                const float4 x0 = *reinterpret_cast<const float4*>(af + (MODE >= 3 ? 3 * 16 * 528 / 4 : 0) + r * 260 + ((2 * v) % 16) * 16 + q * 4);
                const float4 x1 = *reinterpret_cast<const float4*>(af + (MODE >= 3 ? 3 * 16 * 528 / 4 : 0) + r * 260 + ((2 * v + 1) % 16) * 16 + q * 4);
                const float4 b0 = ring[u % R][0], b1 = ring[u % R][1];
                acc[t][0] = mf32(x0.x, b0.x, acc[t][0]);
                acc[t][1] = mf32(x1.x, b1.x, acc[t][1]);
                acc[t][0] = mf32(x0.y, b0.y, acc[t][0]);
                acc[t][1] = mf32(x1.y, b1.y, acc[t][1]);
                acc[t][0] = mf32(x0.z, b0.z, acc[t][0]);
                acc[t][1] = mf32(x1.z, b1.z, acc[t][1]);
                acc[t][0] = mf32(x0.w, b0.w, acc[t][0]);
                acc[t][1] = mf32(x1.w, b1.w, acc[t][1]);
            } else {
                B3 a;
                if (prod == 0) {
                    a = a1[v % 4];
                } else {
                    constexpr int s = prod == 2 ? v % 4 : v;
                    a.h = as_bf(*reinterpret_cast<const float4*>(ab + r * 528 + s * 64 + q * 16));
                    a.m = as_bf(*reinterpret_cast<const float4*>(ab + PLANE + r * 528 + s * 64 + q * 16));
                    a.l = as_bf(*reinterpret_cast<const float4*>(ab + 2 * PLANE + r * 528 + s * 64 + q * 16));
                }
                B3 b;
                if (MODE != 2) {
                    b.h = as_bf(ring[u % R][0]);
                    b.m = as_bf(ring[u % R][1]);
                    b.l = as_bf(ring[u % R][NB - 1]);
                } else {
                    b = split8(ring[u % R][0], ring[u % R][1]);
                }
                six(a, b, acc[t][0], acc[t][1]);
            }
            if (u == 31 || u == 39 || u == 47 || u == 55) {      // end of a product: epilogue into LDS, barrier
                constexpr int nt = u == 31 ? 8 : (u == 47 ? 2 : 1);
#pragma unroll
                for (int tt = 0; tt < nt; ++tt) {
                    const f32x4 s = acc[tt][0] + acc[tt][1];
                    keep += s[0] + s[1] + s[2] + s[3];
                    if (!is_bf((u + 1) % kUnits)) {
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) (af + (MODE >= 3 ? 3 * 16 * 528 / 4 : 0))[(q * 4 + ii) * 260 + ((w * 16 + tt * 16 + r) & 255)] = s[ii] * 1e-3f;
                    } else {      // split + three 2-byte stores per value
#pragma unroll
                        for (int ii = 0; ii < 4; ii += 2) {
                            uint32_t h, m, lo;
                            split_pair(s[ii] * 1e-3f, s[ii + 1] * 1e-3f, h, m, lo);
                            const int c = (w * 16 + tt * 16 + r) & 255;
                            uint16_t* p0 = reinterpret_cast<uint16_t*>(ab + (q * 4 + ii) * 528) + c;
                            uint16_t* p1 = reinterpret_cast<uint16_t*>(ab + (q * 4 + ii + 1) * 528) + c;
                            p0[0] = (uint16_t)h, p1[0] = (uint16_t)(h >> 16);
                            p0[PLANE / 2] = (uint16_t)m, p1[PLANE / 2] = (uint16_t)(m >> 16);
                            p0[PLANE] = (uint16_t)lo, p1[PLANE] = (uint16_t)(lo >> 16);
                        }
                    }
                    acc[tt][0] = acc[tt][1] = f32x4{0, 0, 0, 0};
                }
                __syncthreads();
            }
        };
        [&]<int... I>(std::integer_sequence<int, I...>) { (body(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, kUnits>{});
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (keep == 1234.5f) out[tid] = keep;
    if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <int MODE, int PRE>
void run(const char* name, const char* W, float* out, unsigned long long* cyc) {
    const int layers = 6;
    const size_t smem = 3 * 16 * 528 + 16 * 260 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k_layer<MODE, PRE><<<192, 512, smem>>>(W, layers, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k_layer<MODE, PRE><<<192, 512, smem>>>(W, layers, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double bytes = 8.0 * (MODE == 1 ? kUnits * 3072 : MODE == 3 ? 32 * 3072 + 24 * 2048 : MODE == 4 ? 32 * 2048 + 24 * 3072 : kUnits * 2048);
    printf("%-28s pre %d: %7.1f us per launch (6 layers), %6.1f K ticks per layer (100 MHz x ~21-24), %5.1f B per ns per CU\n", name, PRE, ms * 100, c / 6e3,
           bytes * 6 / (ms * 1e5));
}

int main() {
    char* W;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&W, 64 << 20);
    hipMalloc(&out, 1 << 16);
    hipMalloc(&cyc, 64);
    hipMemset(W, 0x11, 64 << 20);
    run<0, 7>("native f32", W, out, cyc);
    run<1, 7>("six bf16 terms, 3 planes", W, out, cyc);
    run<1, 5>("six bf16 terms, 3 planes", W, out, cyc);
    run<2, 7>("six bf16 terms, split here", W, out, cyc);
    run<2, 5>("six bf16 terms, split here", W, out, cyc);
    run<3, 7>("K=128 x 8 tiles bf16, rest f32", W, out, cyc);
    run<3, 5>("K=128 x 8 tiles bf16, rest f32", W, out, cyc);
    run<4, 7>("first f32, three small bf16", W, out, cyc);
    return 0;
}

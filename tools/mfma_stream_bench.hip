// Micro-benchmark of the row-tile product used by every chain kernel: C[16, NT*16] per wave,
// A (16 x K) in LDS, B streamed from a TILED L2-resident weight through a register ring.
// Reports shader cycles per MFMA for one workgroup per CU (192 workgroups), K = 128 / 1024.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_stream_bench.hip -o tools/msb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// MODE 0: B from registers only (no loads)   1: tiled loads through the ring
template <int NT, int RING, int CK, int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const float* __restrict__ Wt, int KS, int reps, float* out, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float As[16][1028];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    for (int i = tid; i < 16 * 1028; i += WAVES * 64) (&As[0][0])[i] = 0.001f * (i % 97);
    __syncthreads();
    const float* ap = &As[r][q * 4];
    const float* wp = Wt + (size_t)(w * NT) * KS * 256 + lane * 4;      // wave owns NT n-tiles; tile stride KS*256
    f32x4 acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float4 ring[RING][CK][NT];
    const int NC = KS / CK;
    auto fetch = [&](int slot, int c) {
#pragma unroll
        for (int k = 0; k < CK; ++k)
#pragma unroll
            for (int t = 0; t < NT; ++t) ring[slot][k][t] = *reinterpret_cast<const float4*>(wp + (size_t)t * KS * 256 + (c * CK + k) * 256);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < RING - 1; ++c) fetch(c, c);
        } else {
#pragma unroll
            for (int s = 0; s < RING; ++s)
#pragma unroll
                for (int k = 0; k < CK; ++k)
#pragma unroll
                    for (int t = 0; t < NT; ++t) ring[s][k][t] = make_float4(1.f + s, 2.f + k, 3.f + t, 4.f + rep);
        }
        for (int c0 = 0; c0 < NC; c0 += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const int c = c0 + u;
                if (MODE == 1 && c + RING - 1 < NC) fetch((u + RING - 1) % RING, c + RING - 1);
#pragma unroll
                for (int k = 0; k < CK; ++k) {
                    const float4 a = *reinterpret_cast<const float4*>(ap + ((c * CK + k) * 16) % 1024);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float4 b = ring[u][k][t];
                        acc[t] = mfma4(a.x, b.x, acc[t]);
                        acc[t] = mfma4(a.y, b.y, acc[t]);
                        acc[t] = mfma4(a.z, b.z, acc[t]);
                        acc[t] = mfma4(a.w, b.w, acc[t]);
                    }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 1234.5f) out[tid] = s;
    if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <int NT, int RING, int CK, int MODE, int WAVES>
void run(const char* name, const float* W, float* out, unsigned long long* cyc, int KS) {
    const int reps = 8;
    k<NT, RING, CK, MODE, WAVES><<<192, WAVES * 64>>>(W, KS, reps, out, cyc);
    hipDeviceSynchronize();
    k<NT, RING, CK, MODE, WAVES><<<192, WAVES * 64>>>(W, KS, reps, out, cyc);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = (double)reps * KS * 4 * NT;
    printf("%-22s NT %d ring %d ck %d waves %d KS %3d : %8llu cyc  %6.1f cyc/MFMA per wave (ideal %d)\n", name, NT, RING, CK, WAVES, KS, c, c / mf,
           WAVES == 4 ? 32 : 64);
}

int main() {
    float *W, *out;
    unsigned long long* cyc;
    hipMalloc(&W, 64 << 20); hipMalloc(&out, 1 << 16); hipMalloc(&cyc, 64);
    hipMemset(W, 0, 64 << 20);
    for (int KS : {8, 64}) {
        run<4, 4, 2, 0, 4>("regs only", W, out, cyc, KS);
        run<2, 4, 2, 0, 4>("regs only", W, out, cyc, KS);
        run<4, 3, 2, 1, 4>("tiled loads", W, out, cyc, KS);
        run<4, 4, 2, 1, 4>("tiled loads", W, out, cyc, KS);
        run<4, 4, 1, 1, 4>("tiled loads", W, out, cyc, KS);
        run<4, 8, 1, 1, 4>("tiled loads", W, out, cyc, KS);
        run<2, 4, 2, 1, 4>("tiled loads", W, out, cyc, KS);
        run<2, 8, 2, 1, 4>("tiled loads", W, out, cyc, KS);
        run<2, 4, 2, 1, 8>("tiled loads", W, out, cyc, KS);
        run<1, 8, 2, 1, 8>("tiled loads", W, out, cyc, KS);
        run<4, 4, 2, 1, 8>("tiled loads", W, out, cyc, KS);
    }
    return 0;
}

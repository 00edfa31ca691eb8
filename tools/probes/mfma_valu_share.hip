// What a wave's non-MFMA work costs while the other wave of its SIMD streams MFMAs (gfx950): 512-thread workgroup, waves 0..3 run
// a VALU / DPP / LDS / transcendental chain, waves 4..7 (same SIMDs) run v_mfma_f32_16x16x4_f32 with NACC accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_share.hip -o build/mfma_valu_share && build/mfma_valu_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int NACC, bool MFMA_ON>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int iters) {
    __shared__ float lds[4096];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = threadIdx.x;
    lds[threadIdx.x + 512] = threadIdx.x;
    __syncthreads();
    float s = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w >= 4) {
        if (MFMA_ON) {
            f32x4 acc[NACC];
            for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
            float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else {
        float x = lane * 0.01f + 1.0f, y = 0.5f;
        const int n = iters * 8 * NACC / 4;      // ~ the same wall time as the MFMA waves if a chain step took 128 cycles
        for (int it = 0; it < n; ++it) {
            if (KIND == 0) {             // 32 dependent FMAs
#pragma unroll
                for (int j = 0; j < 32; ++j) x = __builtin_fmaf(x, 0.999f, y);
            } else if (KIND == 1) {      // 8 x (DPP row rotate + add)
#pragma unroll
                for (int j = 0; j < 8; ++j) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));
            } else if (KIND == 2) {      // 4 x dependent LDS round trip (write, read)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lds[1024 + w * 128 + lane] = x;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    x = lds[1024 + w * 128 + (lane ^ 1)] + 1.0f;
                }
            } else if (KIND == 3) {      // 8 dependent exp
#pragma unroll
                for (int j = 0; j < 8; ++j) x = __expf(x * 0.001f);
            } else if (KIND == 4) {      // 4 dependent MFMAs (the attention's own small products)
                f32x4 c = {x, x, x, x};
#pragma unroll
                for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0);
                x = c[0] * 1e-6f;
            }
        }
        s = x;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) ticks[w] = t1 - t0;
}
template <int KIND, int NACC>
void run(const char* what, int steps_per_it, float* out, unsigned long long* ticks) {
    const int iters = 400;
    unsigned long long h[2][8];
    for (int on = 0; on < 2; ++on) {
        for (int rep = 0; rep < 2; ++rep) {
            if (on)
                hipLaunchKernelGGL((k<KIND, NACC, true>), dim3(1), dim3(512), 0, 0, out, ticks, iters);
            else
                hipLaunchKernelGGL((k<KIND, NACC, false>), dim3(1), dim3(512), 0, 0, out, ticks, iters);
        }
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h[on], ticks, sizeof(h[on]), hipMemcpyDeviceToHost);
    }
    const double n = (double)(iters * 8 * NACC / 4) * steps_per_it;
    printf("%-28s NACC %d: chain step %.1f ticks alone, %.1f beside MFMAs;  MFMA wave: %.1f ticks per MFMA\n", what, NACC, h[0][0] / n, h[1][0] / n,
           h[1][4] / ((double)iters * 8 * NACC));
}
int main() {
    float* out;
    unsigned long long* ticks;
    (void)hipMalloc(&out, 1 << 20);
    (void)hipMalloc(&ticks, 1024);
    run<0, 2>("dependent v_fma", 32, out, ticks);
    run<0, 4>("dependent v_fma", 32, out, ticks);
    run<1, 2>("DPP rotate + add", 8, out, ticks);
    run<1, 4>("DPP rotate + add", 8, out, ticks);
    run<2, 2>("LDS write -> read round trip", 4, out, ticks);
    run<2, 4>("LDS write -> read round trip", 4, out, ticks);
    run<3, 2>("dependent v_exp", 8, out, ticks);
    run<3, 4>("dependent v_exp", 8, out, ticks);
    run<4, 2>("dependent MFMA", 4, out, ticks);
    run<4, 4>("dependent MFMA", 4, out, ticks);
    return 0;
}

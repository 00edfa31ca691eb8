// Issue cost of dependent v_mfma_f32_16x16x4_f32 chains on gfx950: NACC independent accumulators used round-robin by one
// wave (NACC = 1: every MFMA reads the previous one's result), then the same with two waves on one SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain.hip -o build/mfma_chain && build/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(float* out, unsigned long long* ticks, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC>
void run(int threads, float* out, unsigned long long* ticks) {
    const int iters = 200;
    hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(threads), 0, 0, out, ticks, iters);
    hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(threads), 0, 0, out, ticks, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)iters * 16 * NACC;
    printf("NACC %d, %d waves (%d per SIMD): ticks per MFMA of each wave:", NACC, threads / 64, (threads / 64 + 3) / 4);
    for (int w = 0; w < threads / 64; ++w) printf(" %.1f", h[w] / n);
    printf("\n");
}
int main() {
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 1 << 20);
    hipMalloc(&ticks, 1024);
    for (int threads : {64, 256, 512}) {
        run<1>(threads, out, ticks);
        run<2>(threads, out, ticks);
        run<4>(threads, out, ticks);
    }
    // clock: ticks per microsecond
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, ticks, 20000);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h0;
    hipMemcpy(&h0, ticks, 8, hipMemcpyDeviceToHost);
    printf("memtime ticks per us: %.1f  (kernel %.3f ms, %.1f ns per MFMA)\n", h0 / (ms * 1e3), ms, ms * 1e6 / (20000.0 * 64));
    return 0;
}

// Probe: a 64 x 64 tile global -> LDS by dword LDS-DMA, one wave-instruction per tile row (256 contiguous bytes), issued from inline assembly
// (the compiler does not see the transfer: no conservative vmcnt(0) in front of the next LDS access, the waits are the kernel's own), into a PADDED
// row stride -- M0 carries the row's LDS address.  Checks the copy and times it against the register path (16-byte loads + ds_write_b128).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_row.hip -o /tmp/lds_dma_row && /tmp/lds_dma_row
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void lds_dma_row(const float* srow, unsigned voff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(srow), "s"(lds_byte) : "memory");
}
__global__ void k(const float* src, float* dst, int ld) {
    __shared__ float tile[64 * 68];
    typedef __attribute__((address_space(3))) float* lds_ptr;
    const unsigned base = (unsigned)(uintptr_t)(lds_ptr)tile;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = 0; i < 16; ++i) {
        const int j = 16 * w + i;
        lds_dma_row(src + (size_t)j * ld, lane * 4u, base + j * 68 * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) dst[i] = tile[(i >> 6) * 68 + (i & 63)];
}
int main() {
    const int ld = 384;
    std::vector<float> h(64 * ld);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, h.size() * 4);
    hipMalloc(&d, 64 * 64 * 4);
    hipMemcpy(s, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, s + 64, d, ld);
    std::vector<float> o(64 * 64);
    hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c < 64; ++c) bad += o[r * 64 + c] != h[(size_t)r * ld + 64 + c];
    printf("lds_dma_row: %d mismatches of 4096\n", bad);
    return bad != 0;
}

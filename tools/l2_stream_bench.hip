// Micro-benchmark: how fast can one workgroup per CU stream a (shared, L2-resident) weight
// matrix into registers, for the three lane->address patterns the MFMA products can use?
//   P0 "nt":    lane (r = l&15, q = l>>4) reads 16 B at W[(n0 + r) * K + k0 + 4q]   (16 rows per quarter-wave)
//   P1 "nn":    lane reads 16 B at W[(k0 + q) * N + n0 + 4r]                        (256 B contiguous per quarter-wave)
//   P2 "tiled": lane l reads 16 B at base + l * 16                                  (1 KB contiguous per instruction)
// hipcc --offload-arch=gfx950 -O3 tools/l2_stream_bench.hip -o /tmp/l2b && /tmp/l2b
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PAT, int DEPTH>
__global__ __launch_bounds__(256) void stream(const float* __restrict__ W, int words_per_wave, float* out, int shared, int reps) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    // each wave owns a contiguous slice of the matrix; matrix = [1024 rows][128 floats]
    const float* base = W + (size_t)(shared ? 0 : blockIdx.x) * 4 * words_per_wave + (size_t)w * words_per_wave;
    float4 acc = make_float4(0, 0, 0, 0);
    const int n_instr = words_per_wave / 256;    // 256 floats (1 KB) per instruction
    for (int rep = 0; rep < reps; ++rep) {
        float4 ring[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) ring[d] = make_float4(0, 0, 0, 0);
        for (int i0 = 0; i0 < n_instr; i0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int i = i0 + d;
                size_t off;
                if (PAT == 0) {          // instr i covers rows (i/8)*16.. +16, k chunk (i%8)*16
                    off = (size_t)((i >> 3) * 16 + r) * 128 + (i & 7) * 16 + q * 4;
                } else if (PAT == 1) {   // matrix viewed [K=128.. rows][N cols = 64]: instr i covers 4 k-rows x 64 cols
                    off = (size_t)(i * 4 + q) * 64 + 4 * r;
                } else {
                    off = (size_t)i * 256 + lane * 4;
                }
                const float4 v = *reinterpret_cast<const float4*>(base + off);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int PAT, int DEPTH>
void run(const char* name, const float* W, float* out, int shared, int nblk) {
    const int words_per_wave = 32768;   // 128 KB per wave, 512 KB per workgroup (one 1024x128 matrix)
    const int reps = 20;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    stream<PAT, DEPTH><<<nblk, 256>>>(W, words_per_wave, out, shared, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    stream<PAT, DEPTH><<<nblk, 256>>>(W, words_per_wave, out, shared, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nblk * 4 * words_per_wave * 4 * reps;
    printf("%-6s depth %2d %s blocks %3d: %8.1f us  %7.2f TB/s aggregate  %6.1f B/clk/CU (2.4 GHz)\n", name, DEPTH, shared ? "shared " : "private",
           nblk, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / nblk / (ms * 1e-3 * 2.4e9));
}

int main() {
    float *W, *out;
    const size_t n = (size_t)256 * 4 * 32768;   // 128 MB: private slices for 256 blocks
    hipMalloc(&W, n * 4); hipMalloc(&out, 4096);
    hipMemset(W, 0, n * 4);
    for (int shared = 1; shared >= 0; --shared)
        for (int nblk : {192, 24}) {
            run<0, 8>("nt", W, out, shared, nblk);
            run<1, 8>("nn", W, out, shared, nblk);
            run<2, 8>("tiled", W, out, shared, nblk);
            run<0, 16>("nt", W, out, shared, nblk);
            run<2, 16>("tiled", W, out, shared, nblk);
            run<2, 2>("tiled", W, out, shared, nblk);
        }
    return 0;
}

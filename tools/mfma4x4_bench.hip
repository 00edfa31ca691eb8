// Probe of v_mfma_f32_4x4x1_16b_f32 with the A block broadcast to all 16 blocks (cbsz = 4): one instruction is then
// D[4 x 64] += A[4 x 1] . B[1 x 64] -- a 4-row granule instead of the 16 rows of v_mfma_f32_16x16x4_f32, which
// matters when only T = 9 of 16 rows of a tile are live.  Checks the lane mapping against a host model and measures
// issue cycles per instruction at one and two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_bench.hip -o tools/m44
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ABID>
__device__ __forceinline__ f32x4 mm(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0); }

__global__ void k_sem(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c0 = {0, 0, 0, 0}, c5 = {0, 0, 0, 0}, cn = {0, 0, 0, 0};
    c0 = mm<0>(a[l], b[l], c0);
    c5 = mm<5>(a[l], b[l], c5);
    cn = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], cn, 0, 0, 0);
    for (int i = 0; i < 4; ++i) {
        d[l * 4 + i] = c0[i];
        d[256 + l * 4 + i] = c5[i];
        d[512 + l * 4 + i] = cn[i];
    }
}

// 3 row groups x NC column chunks of 64: per k: 3*NC instructions, A regs hold 16 (group,k) blocks each
template <int NC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_rate(const float* src, int reps, float* out, unsigned long long* cyc) {
    const int tid = threadIdx.x;
    float av[3], bv[NC][4];
    for (int i = 0; i < 3; ++i) av[i] = src[tid + i * 64];
    for (int c = 0; c < NC; ++c)
        for (int k = 0; k < 4; ++k) bv[c][k] = src[(tid + c * 4 + k) & 1023];
    f32x4 acc[3][NC];
    for (int g = 0; g < 3; ++g)
        for (int c = 0; c < NC; ++c) acc[g][c] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[0][c] = mm<0>(av[0], bv[c][0], acc[0][c]);
            acc[1][c] = mm<1>(av[0], bv[c][0], acc[1][c]);
            acc[2][c] = mm<2>(av[0], bv[c][0], acc[2][c]);
            acc[0][c] = mm<3>(av[0], bv[c][1], acc[0][c]);
            acc[1][c] = mm<4>(av[0], bv[c][1], acc[1][c]);
            acc[2][c] = mm<5>(av[0], bv[c][1], acc[2][c]);
            acc[0][c] = mm<6>(av[1], bv[c][2], acc[0][c]);
            acc[1][c] = mm<7>(av[1], bv[c][2], acc[1][c]);
            acc[2][c] = mm<8>(av[1], bv[c][2], acc[2][c]);
            acc[0][c] = mm<9>(av[2], bv[c][3], acc[0][c]);
            acc[1][c] = mm<10>(av[2], bv[c][3], acc[1][c]);
            acc[2][c] = mm<11>(av[2], bv[c][3], acc[2][c]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int g = 0; g < 3; ++g)
        for (int c = 0; c < NC; ++c) s += acc[g][c][0] + acc[g][c][1] + acc[g][c][2] + acc[g][c][3];
    if (s == 1234.5f) out[tid] = s;
    if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

// the same work on 16x16x4: NC*4 column tiles of 16, K = 4 per instruction
template <int NC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_rate16(const float* src, int reps, float* out, unsigned long long* cyc) {
    const int tid = threadIdx.x;
    float av = src[tid], bv[NC * 4];
    for (int c = 0; c < NC * 4; ++c) bv[c] = src[(tid + c) & 1023];
    f32x4 acc[NC * 4];
    for (int c = 0; c < NC * 4; ++c) acc[c] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int c = 0; c < NC * 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[c], acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < NC * 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 1234.5f) out[tid] = s;
    if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <class F>
double timed(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float ha[64], hb[64], hd[768], *a, *b, *d, *src, *out;
    unsigned long long* cyc;
    for (int l = 0; l < 64; ++l) {
        ha[l] = 1.0f + l;
        hb[l] = 0.5f + 0.25f * l;
    }
    hipMalloc(&a, 256), hipMalloc(&b, 256), hipMalloc(&d, 768 * 4), hipMalloc(&src, 4096 * 4), hipMalloc(&out, 4096 * 4), hipMalloc(&cyc, 8);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice);
    hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    float hs[4096];
    for (int i = 0; i < 4096; ++i) hs[i] = 0.001f * (i % 97) - 0.03f;
    hipMemcpy(src, hs, sizeof hs, hipMemcpyHostToDevice);
    k_sem<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
    // model: lane l = (block l/4, j = l%4); D reg i of lane l = A_blk[row i] * B_blk[col j]; with cbsz = 4 the A block is ABID's
    int bad0 = 0, bad5 = 0, badn = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int blk = l / 4;
            bad0 += hd[l * 4 + i] != ha[0 * 4 + i] * hb[l];
            bad5 += hd[256 + l * 4 + i] != ha[5 * 4 + i] * hb[l];
            badn += hd[512 + l * 4 + i] != ha[blk * 4 + i] * hb[l];
        }
    printf("semantics: cbsz4/abid0 mismatches %d, cbsz4/abid5 %d, no-broadcast %d (of 256 each)\n", bad0, bad5, badn);
    printf("  lane 9 regs (abid5): %g %g %g %g   expect A[20..23]*B[9] = %g %g %g %g\n", hd[256 + 36], hd[256 + 37], hd[256 + 38], hd[256 + 39],
           ha[20] * hb[9], ha[21] * hb[9], ha[22] * hb[9], ha[23] * hb[9]);
    const int reps = 2000;
    auto report = [&](const char* name, int nc, int waves, double ms, int per_rep, int ideal) {
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-10s NC %d waves/WG %d: %9llu cyc, %6.2f cyc per instruction per wave (pipe floor %d at %d wave/SIMD), %.3f ms\n", name, nc, waves, c,
               (double)c / ((double)reps * per_rep), ideal * (waves / 4), waves / 4, ms);
    };
    double ms;
    ms = timed([&] { k_rate<2, 4><<<256, 256>>>(src, reps, out, cyc); });
    report("4x4x1", 2, 4, ms, 24, 8);
    ms = timed([&] { k_rate<2, 8><<<256, 512>>>(src, reps, out, cyc); });
    report("4x4x1", 2, 8, ms, 24, 8);
    ms = timed([&] { k_rate<4, 4><<<256, 256>>>(src, reps, out, cyc); });
    report("4x4x1", 4, 4, ms, 48, 8);
    ms = timed([&] { k_rate<4, 8><<<256, 512>>>(src, reps, out, cyc); });
    report("4x4x1", 4, 8, ms, 48, 8);
    ms = timed([&] { k_rate16<2, 4><<<256, 256>>>(src, reps, out, cyc); });
    report("16x16x4", 2, 4, ms, 8, 32);
    ms = timed([&] { k_rate16<2, 8><<<256, 512>>>(src, reps, out, cyc); });
    report("16x16x4", 2, 8, ms, 8, 32);
    return 0;
}

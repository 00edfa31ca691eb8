// What does it cost four workgroups on four CUs to sum a [48 x 128] fp32 partial tile between them, inside one launch?
// (The question behind splitting a Regulation layer's output columns over a TEAM of workgroups: every CU then streams a quarter of the
// layer's weights and the 64 CUs a 192-workgroup launch leaves idle get work, at the price of two such sums per layer.)
//
// 256 workgroups of 512 threads, one per CU (100 KB of LDS each), teams of four.  One exchange: every member writes its 24 KB partial to its
// slot, counts itself in on the team's counter, waits for the other three, reads all four slots (fixed order: the sum is the same bits
// in every member) -- `iters` times, double-buffered slots, shader-clock ticks per exchange from wave 0 of every workgroup.
//   placement same : the members of a team are 8 workgroup ids apart (ids go round-robin over the 8 XCDs: one L2 for the team);
//                    data with plain stores (the vector L1 writes through to the L2), loads at agent scope (never the L1)
//   placement same + agent stores : the same teams, data written through with agent-scope stores as well
//   placement cross: the members are neighbours in id (four different XCDs); agent-scope stores and loads, slots never reused
//                    (a slot read once stays in the reader's L2: private L2s are coherent for this memory only between launches)
// Every variant checks the sums it reads against what the members wrote (mismatches are counted and printed).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/team_exchange.hip -o build/team_exchange && build/team_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int kRows = 48, kCols = 128, kTileF = kRows * kCols;      // 24 KB
constexpr int kThreads = 512, kPerThread = kTileF / kThreads / 4;   // float4s per thread: 3

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}
// 16-byte accesses with an explicit cache policy: raw buffer instructions (aux bit 4 = sc1: agent scope on gfx94x / gfx950 -- a load never hits
// the CU's vector L1, a store is written through; aux 0: a plain access)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kSc1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(float* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000); }
template <int AUX>
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t rs, unsigned byte, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, rs, byte, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ float4 ld16(__amdgpu_buffer_rsrc_t rs, unsigned byte) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte, 0, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// MODE 0: plain stores, agent loads (same-XCD teams); 1: agent stores + loads (same-XCD teams); 2: agent stores + loads, cross-XCD teams, fresh slots
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_exchange(float* __restrict__ slots, int* __restrict__ cnt, unsigned long long* __restrict__ ticks,
                                                       unsigned* __restrict__ xcc, int* __restrict__ bad, int iters) {
    extern __shared__ float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    int team, member;
    if (MODE == 2) {
        team = b >> 2, member = b & 3;
    } else {
        team = (b >> 5) * 8 + (b & 7), member = (b >> 3) & 3;
    }
    if (tid == 0) xcc[b] = xcc_id();
    const int nbuf = MODE == 2 ? iters : 2;
    float* tslots = slots + (size_t)team * nbuf * 4 * kTileF;
    int* tc = cnt + team * 32;      // a counter per team, 128 bytes apart
    lds[tid] = 0.f;
    __syncthreads();
    int wrong = 0;
    unsigned long long t_sum = 0;
    for (int it = 0; it < iters; ++it) {
        const __amdgpu_buffer_rsrc_t rs = rsrc(tslots + (size_t)(it % nbuf) * 4 * kTileF, 4 * kTileF * sizeof(float));      // the four slots of this exchange
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const int i = (k * kThreads + tid) * 4;      // float index inside the tile
            const float v0 = (float)(member + 1) * 0.25f + (float)(it & 63);
            const float4 v = make_float4(v0 + (float)(i & 7) * 0.125f, v0 + (float)((i + 1) & 7) * 0.125f, v0 + (float)((i + 2) & 7) * 0.125f, v0 + (float)((i + 3) & 7) * 0.125f);
            st16<MODE == 0 ? 0 : kSc1>(rs, (member * kTileF + i) * 4, v);
        }
        __builtin_amdgcn_s_waitcnt(0);      // the stores have left the CU (vmcnt = 0: acknowledged by the L2 / by memory)
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(tc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (relaxed: a release here is an L2 write-back, ~16 us; the data is already out)
            while (__hip_atomic_load(tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4 * (it + 1)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const int i = (k * kThreads + tid) * 4;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < 4; ++m) {      // fixed order: every member forms the same bits
                const float4 p = ld16<kSc1>(rs, (m * kTileF + i) * 4);
                s.x += p.x, s.y += p.y, s.z += p.z, s.w += p.w;
            }
            const float w0 = 2.5f + 4.f * (float)(it & 63);
            if (s.x != w0 + 0.5f * (float)(i & 7) || s.y != w0 + 0.5f * (float)((i + 1) & 7) || s.z != w0 + 0.5f * (float)((i + 2) & 7) ||
                s.w != w0 + 0.5f * (float)((i + 3) & 7))
                ++wrong;
            acc += s.x + s.y + s.z + s.w;
        }
        lds[tid] += acc;
        __syncthreads();      // (nobody overwrites a slot of parity (it % 2) before everybody has read it: the next write to it is two exchanges on,
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();      //  behind the counter wait of the exchange in between)
        t_sum += t1 - t0;
    }
    if (wrong) atomicAdd(bad, wrong);
    if (tid == 0) ticks[b] = t_sum;
    if (lds[tid] == 12345.f) ticks[b] = 0;
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
            return 1;                                                              \
        }                                                                          \
    } while (0)

template <int MODE>
static int run(const char* name, int iters) {
    const int nwg = 256, teams = nwg / 4, nbuf = MODE == 2 ? iters : 2;
    float* slots;
    int *cnt, *bad;
    unsigned long long* ticks;
    unsigned* xcc;
    CK(hipMalloc(&slots, (size_t)teams * nbuf * 4 * kTileF * sizeof(float)));
    CK(hipMalloc(&cnt, teams * 32 * sizeof(int)));
    CK(hipMalloc(&bad, sizeof(int)));
    CK(hipMalloc(&ticks, nwg * sizeof(unsigned long long)));
    CK(hipMalloc(&xcc, nwg * sizeof(unsigned)));
    CK(hipFuncSetAttribute((const void*)k_exchange<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    std::vector<unsigned long long> t(nwg);
    std::vector<unsigned> x(nwg);
    double best = 1e30, med = 0, us_launch = 1e30;
    int nbad = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(cnt, 0, teams * 32 * sizeof(int)));
        CK(hipMemset(bad, 0, sizeof(int)));
        CK(hipMemset(slots, 0, (size_t)teams * nbuf * 4 * kTileF * sizeof(float)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_exchange<MODE>, dim3(nwg), dim3(kThreads), 100 * 1024, 0, slots, cnt, ticks, xcc, bad, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us_launch = std::min(us_launch, (double)ms * 1e3 / iters);
        CK(hipMemcpy(t.data(), ticks, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        CK(hipMemcpy(x.data(), xcc, nwg * sizeof(unsigned), hipMemcpyDeviceToHost));
        int bb;
        CK(hipMemcpy(&bb, bad, sizeof(int), hipMemcpyDeviceToHost));
        nbad += bb;
        std::vector<double> per(nwg);
        for (int i = 0; i < nwg; ++i) per[i] = (double)t[i] / iters;
        std::sort(per.begin(), per.end());
        best = std::min(best, per[0]);
        med = per[nwg / 2];
    }
    int same = 0;
    for (int tm = 0; tm < teams; ++tm) {
        int ids[4];
        for (int m = 0; m < 4; ++m) ids[m] = MODE == 2 ? tm * 4 + m : (tm >> 3) * 32 + (tm & 7) + 8 * m;
        same += (x[ids[0]] == x[ids[1]] && x[ids[1]] == x[ids[2]] && x[ids[2]] == x[ids[3]]);
    }
    printf("%-44s %8.0f s_memtime ticks per exchange (median workgroup; fastest %.0f); launch / exchanges = %.2f us; teams on one XCD: %d of %d; wrong sums: %d\n",
           name, med, best, us_launch, same, teams, nbad);
    printf("    XCC id of workgroups 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %u", x[i]);
    printf("\n");
    hipFree(slots), hipFree(cnt), hipFree(bad), hipFree(ticks), hipFree(xcc);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64;
    printf("four workgroups sum a [48 x 128] fp32 tile (24 KB written, 96 KB read per member), %d exchanges per launch\n", iters);
    if (run<0>("same XCD, plain stores / agent loads", iters)) return 1;
    if (run<1>("same XCD, agent stores / agent loads", iters)) return 1;
    if (run<2>("four XCDs, agent stores / loads, fresh slots", iters)) return 1;
    return 0;
}

// cf_kernels.h -- gfx950 (MI355X) device code of the Chromoformer hot path.
//
// All arithmetic is fp32.  Dense contractions run on the f32-input matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, bit-equal to an fmaf chain), everything else on
// the VALU with 64-lane wavefront shuffles for the reductions.  One workgroup = 256
// threads = 4 wavefronts.  "Row tile" kernels own 16 activation rows (one MFMA M-tile)
// and walk a chain of Linear / LayerNorm ops with the activations resident in LDS and
// the weights streamed from L2 straight into MFMA B fragments.
//
// Reference semantics (file:line under /root/reference/chromoformer):
//   attention scores / mask / softmax ....... modules.py:58-77, 170-189
//   gate, frequency bias ..................... modules.py:63-69, 80-81
//   out-projection + residual + LayerNorm .... modules.py:28-30, 150-152
//   FeedForward .............................. modules.py:100-101
//   centre row, positional table ............. net.py:23-29, 59, 138
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#ifndef CF_TRUNK_NT      // streaming hints for what the centre-row kernels only save for the backward pass (1: stores, 2: loads)
#define CF_TRUNK_NT 0
#endif

// Every read of threadIdx.x is OPAQUE to the optimiser (CF_TID_OPAQUE=0: the plain builtin).  The fused trunk kernels run ~20 phase bodies one
// behind the other in straight-line code; each body starts from `tid = threadIdx.x` and derives its lane / row / column offsets from it.
// Left alone the compiler recognises that tid >> 3, tid << 5, (tid & 63) * 16 ... of phase 17 were already computed in phase 2, keeps them
// alive through the 244-register attention bodies in between and SPILLS them there (k_trunk_bwd: 71 spilled VGPRs, every one of them a
// shift of the thread index; tools/kernel_resources.sh).  Read through an empty `asm volatile` the thread index of one phase has nothing in
// common with that of the next: each phase recomputes its three or four shifts and nothing lives across a phase boundary.
#ifndef CF_TID_OPAQUE
#define CF_TID_OPAQUE 1
#endif
#if CF_TID_OPAQUE
namespace cf {
struct TidProxy {
    struct X {
        __device__ __forceinline__ operator unsigned() const {
            unsigned t = __builtin_amdgcn_workitem_id_x();
            asm volatile("" : "+v"(t));
            return t;
        }
    } x;
    struct Y { __device__ __forceinline__ operator unsigned() const { return __builtin_amdgcn_workitem_id_y(); } } y;
    struct Z { __device__ __forceinline__ operator unsigned() const { return __builtin_amdgcn_workitem_id_z(); } } z;
};
}  // namespace cf
// (scoped to the kernel headers: #undef'd again at the end of this file, behind the last of them)
#define threadIdx (cf::TidProxy{})
#endif

namespace cf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kD = 128;        // d_emb
constexpr int kTile = 16;      // rows per workgroup (MFMA M)
constexpr int kMaxRes = 3;
constexpr float kLnEps = 1e-5f;
constexpr float kMaskFill = -1e9f;

// physical row = (row / div) * mul + (row % div) * rem + off
struct RowMap {
    int div, mul, rem, off;
};
__host__ __device__ inline RowMap identity_map() { return RowMap{1, 1, 0, 0}; }
__device__ __forceinline__ int map_row(const RowMap& m, int row) {
    return (row / m.div) * m.mul + (row % m.div) * m.rem + m.off;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Explicit global-address-space accesses.  A pointer that was itself loaded from memory (a table
// entry, a by-value struct copied through registers) is "generic" to the compiler, which then emits
// flat_load/flat_store: those tick BOTH vmcnt and lgkmcnt, so every wait around them degenerates to
// vmcnt(0) lgkmcnt(0) and no load can stay in flight across an LDS access.  These helpers pin the
// address space so that global_load/global_store are emitted.
#define CF_GLOBAL __attribute__((address_space(1)))
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ldg(const float* p) { return *(const CF_GLOBAL float*)(p); }
__device__ __forceinline__ float2 ldg2(const float* p) {
    const v2f v = *(const CF_GLOBAL v2f*)(p);
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 ldg4(const float* p) {
    const v4f v = *(const CF_GLOBAL v4f*)(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
// streaming forms (nt): data that is touched once -- the optimiser state under the riders -- should not push the tables of the kernel
// that runs beside them out of the L2
__device__ __forceinline__ float4 ldg4_nt(const float* p) {
    const v4f v = __builtin_nontemporal_load((const CF_GLOBAL v4f*)(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ldg_nt(const float* p) { return __builtin_nontemporal_load((const CF_GLOBAL float*)(p)); }
__device__ __forceinline__ void stg_nt(float* p, float v) { __builtin_nontemporal_store(v, (CF_GLOBAL float*)(p)); }
__device__ __forceinline__ void stg4_nt(float* p, float4 v) { __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, (CF_GLOBAL v4f*)(p)); }
__device__ __forceinline__ void stg(float* p, float v) { *(CF_GLOBAL float*)(p) = v; }
__device__ __forceinline__ void stg2(float* p, float2 v) { *(CF_GLOBAL v2f*)(p) = v2f{v.x, v.y}; }
__device__ __forceinline__ void stg4(float* p, float4 v) { *(CF_GLOBAL v4f*)(p) = v4f{v.x, v.y, v.z, v.w}; }
template <int NT>
struct LdgN;
template <>
struct LdgN<1> {
    static __device__ __forceinline__ float ld(const float* p) { return ldg(p); }
    static __device__ __forceinline__ void st(float* p, float v) { stg(p, v); }
};
template <>
struct LdgN<2> {
    static __device__ __forceinline__ float2 ld(const float* p) { return ldg2(p); }
    static __device__ __forceinline__ void st(float* p, float2 v) { stg2(p, v); }
};
template <>
struct LdgN<4> {
    static __device__ __forceinline__ float4 ld(const float* p) { return ldg4(p); }
    static __device__ __forceinline__ void st(float* p, float4 v) { stg4(p, v); }
};

// Reductions inside a row of 16 lanes run on DPP row rotations (one VALU op per step) instead of ds_bpermute.  Every
// lane ends with the same bits: at each step the two lanes of a pair add the same two values, only commuted.
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float group16_sum(float v) {
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, row_ror<8>(v));
    v = fmaxf(v, row_ror<4>(v));
    v = fmaxf(v, row_ror<2>(v));
    v = fmaxf(v, row_ror<1>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = group16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// ---------------------------------------------------------------------------------------
// MFMA tile products.  A is a 16-row tile in LDS (row stride lda floats, 16-byte aligned
// rows).  Lane l = (r = l & 15, q = l >> 4).  The 16x16x4 instruction wants
// A[i = r][k = q] and B[k = q][j = r]; the reduction index is assigned to lanes in
// blocks of four (k = k0 + 4q + i for the i-th of four MFMAs) so that the A operand is
// read 16 bytes at a time.  Result fragment: acc[t][i] = C[row 4q + i][col(t, r)].
//
// The weights are not shared between the waves of a workgroup (each wave owns its own
// output columns), so they go straight from L2 into registers: a wave first issues ALL
// loads of its B fragments for the product (up to 128 VGPRs, fully unrolled, nothing
// waits), typically before the barrier / LayerNorm that precedes the product, and then
// runs the MFMAs as the data lands.  With ~100 resident workgroups on 256 CUs this
// memory-level parallelism is what hides the L2 latency.
// ---------------------------------------------------------------------------------------

// Operand streaming.  A product walks K in chunks of 32 (two MFMA k-steps); the B operand of
// chunk c+2 is requested from L2 while chunk c is multiplied (3-slot register ring, at most
// 2*NT*2 sixteen-byte loads outstanding, well inside the 6-bit vmcnt counter so the compiler
// can emit counted waits instead of draining the queue).  frag_load_*() issues the first two
// chunks -- typically before the barrier / LayerNorm that precedes the product -- and
// frag_mma_*() runs the pipeline.

// "NT" fragments: C = A . W^T for an nn.Linear weight W[n][K].  Reading W row-major for the MFMA
// B operand puts 16 different rows (cache lines) into every quarter-wave of a load, which the
// texture addresser serialises: 15 B/clk/CU measured, half of what the matrix pipes consume
// (tools/l2_stream_bench.hip).  The forward products therefore read a TILED copy of the weights
// (k_retile, refreshed at the start of every forward): each 16(n) x 16(k) block is stored as 256
// contiguous floats in fragment order [q][r][4], so a wave's load is one contiguous 1 KB
// (55-60 B/clk/CU measured).  A sub-matrix starting at a row multiple of 16 has the same offset in
// both layouts.  Column map: col(t, r) = t*16 + r.  KS = K / 16, ldw = K of the full tensor.
#ifndef CF_APREFETCH
#define CF_APREFETCH 1      // read the A operand of a k-step one step ahead (experiment switch)
#endif
constexpr int kRing = 4;      // register ring slots: chunks c+1 .. c+kRing-1 are in flight while chunk c is multiplied
// RING: slots of this product's ring (default kRing).  A chunk of a narrow product (NT = 1: eight MFMAs, ~256 cycles) covers little
// of the ~2 K cycles of an L2 round trip; where a wave is alone with its product (the head: 4 workgroups) requesting everything at
// once paid (cf_head.h: -7 us); in the chain bodies, where the second wave of the SIMD fills the gaps, RING = 8 for the K = 256
// products changed nothing (k_trunk_fwd 100.0 vs 98.9 us, k_trunk_bwd 138.0 vs 139.0).
template <int NT, int KS, int RING = kRing>
struct FragNT {
    float4 ring[RING][2][NT];
    const float* wp;
    int tstride;      // floats between consecutive 16-row tiles: (ldw / 16) * 256
};
template <int NT, int KS, int RING>
__device__ __forceinline__ void frag_chunk_nt(FragNT<NT, KS, RING>& f, int slot, int chunk) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int t = 0; t < NT; ++t)
            f.ring[slot][k][t] = ldg4(f.wp + (size_t)t * f.tstride + (chunk * 2 + k) * 256);
}
// Wt: TILED address of the first output row of this wave (row index a multiple of 16), plus
// (k0 / 16) * 256 for a reduction offset k0; ldw: K of the full tensor.
template <int NT, int KS, int RING>
__device__ __forceinline__ void frag_load_nt(FragNT<NT, KS, RING>& f, const float* __restrict__ Wt, int ldw) {
    static_assert(KS % 2 == 0 && KS >= 2, "K must be a multiple of 32");
    f.wp = Wt + (threadIdx.x & 63) * 4;
    f.tstride = ldw * 16;
#pragma unroll
    for (int c = 0; c < RING - 1; ++c)
        if (c < KS / 2) frag_chunk_nt(f, c, c);
    __builtin_amdgcn_sched_barrier(0);
}
template <int NT, int KS, int RING>
__device__ __forceinline__ void frag_mma_nt(FragNT<NT, KS, RING>& f, const float* As, int lda, f32x4 (&acc)[NT]) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const float* ap = As + r * lda + q * 4;
    constexpr int NC = KS / 2;
    float4 a_nxt = *reinterpret_cast<const float4*>(ap);      // A operand one k-step ahead: its LDS latency hides behind the MFMAs
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + RING - 1 < NC) frag_chunk_nt(f, (c + RING - 1) % RING, c + RING - 1);
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of this chunk's MFMAs (the scheduler would sink it)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float4 a = CF_APREFETCH ? a_nxt : *reinterpret_cast<const float4*>(ap + (c * 2 + k) * 16);
            if (CF_APREFETCH && c * 2 + k + 1 < KS) a_nxt = *reinterpret_cast<const float4*>(ap + (c * 2 + k + 1) * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 b = f.ring[c % RING][k][t];
                acc[t] = mfma4(a.x, b.x, acc[t]);
                acc[t] = mfma4(a.y, b.y, acc[t]);
                acc[t] = mfma4(a.z, b.z, acc[t]);
                acc[t] = mfma4(a.w, b.w, acc[t]);
            }
        }
    }
}
__device__ __forceinline__ int col_nt(int t, int r) { return t * 16 + r; }

// "NN" fragments: Bm row-major [k][ldb], C = A . Bm.  A lane reads NT consecutive floats
// of one k-row, which feed NT different tiles, so tile t holds the STRIDED columns
// col(t, r) = NT*r + t of the wave's 16*NT-column range (one 8/16-byte load per NT MFMAs).
template <int NT>
struct VecN;
template <>
struct VecN<1> {
    typedef float type;
};
template <>
struct VecN<2> {
    typedef float2 type;
};
template <>
struct VecN<4> {
    typedef float4 type;
};
template <int NT, int KS, int RING = kRing>
struct FragNN {
    typename VecN<NT>::type ring[RING][2][4];
    const float* bp;
    int ldb;
};
template <int NT, int KS, int RING>
__device__ __forceinline__ void frag_chunk_nn(FragNN<NT, KS, RING>& f, int slot, int chunk) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            f.ring[slot][k][i] = LdgN<NT>::ld(f.bp + (size_t)((chunk * 2 + k) * 16 + i) * f.ldb);
}
template <int NT, int KS, int RING>
__device__ __forceinline__ void frag_load_nn(FragNN<NT, KS, RING>& f, const float* __restrict__ Bm, int ldb) {
    static_assert(KS % 2 == 0 && KS >= 2, "K must be a multiple of 32");
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    f.bp = Bm + (size_t)(q * 4) * ldb + NT * r;
    f.ldb = ldb;
#pragma unroll
    for (int c = 0; c < RING - 1; ++c)
        if (c < KS / 2) frag_chunk_nn(f, c, c);
    __builtin_amdgcn_sched_barrier(0);
}
template <int NT, int KS, int RING>
__device__ __forceinline__ void frag_mma_nn(FragNN<NT, KS, RING>& f, const float* As, int lda, f32x4 (&acc)[NT]) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const float* ap = As + r * lda + q * 4;
    constexpr int NC = KS / 2;
    float4 a_nxt = *reinterpret_cast<const float4*>(ap);      // A operand one k-step ahead
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + RING - 1 < NC) frag_chunk_nn(f, (c + RING - 1) % RING, c + RING - 1);
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of this chunk's MFMAs (the scheduler would sink it)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float4 a = CF_APREFETCH ? a_nxt : *reinterpret_cast<const float4*>(ap + (c * 2 + k) * 16);
            if (CF_APREFETCH && c * 2 + k + 1 < KS) a_nxt = *reinterpret_cast<const float4*>(ap + (c * 2 + k + 1) * 16);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* bv = reinterpret_cast<const float*>(&f.ring[c % RING][k][i]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma4(av[i], bv[t], acc[t]);
            }
        }
    }
}
template <int NT>
__device__ __forceinline__ int col_nn(int t, int r) { return NT * r + t; }

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// Load a [16, ncols] tile (ncols % 4 == 0) of a row-major global matrix into LDS.
__device__ __forceinline__ void load_tile(float* dst, int ldd, const float* __restrict__ src, int lds_, int ncols,
                                          int row0, int nrows, const RowMap& map) {
    const int c4n = ncols >> 2;
    for (int i = threadIdx.x; i < kTile * c4n; i += blockDim.x) {
        const int rr = i / c4n, c4 = i - rr * c4n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + rr < nrows) v = ldg4(src + (size_t)map_row(map, row0 + rr) * lds_ + c4 * 4);
        *reinterpret_cast<float4*>(dst + rr * ldd + c4 * 4) = v;
    }
}

// The same in two phases -- request into registers, put into LDS -- for the prologue of a chain kernel: loads return in
// order, so the tile a kernel's first barrier waits for is requested FIRST, the operand rings of the products behind it, and
// the put waits for the tile alone.  No branch around and no select directly behind the loads (rows past the end are
// clamped and zeroed at the put): the wait count stays exact.
template <int NCOLS, int NTHR>
struct TileReq {
    static constexpr int C4 = NCOLS / 4, NV = (kTile * C4 + NTHR - 1) / NTHR;
    float4 v[NV];
};
template <int NCOLS, int NTHR>
__device__ __forceinline__ void tile_req(TileReq<NCOLS, NTHR>& t, const float* __restrict__ src, int lds_, int row0, int nrows,
                                         const RowMap& map) {
    constexpr int C4 = NCOLS / 4;
#pragma unroll
    for (int u = 0; u < TileReq<NCOLS, NTHR>::NV; ++u) {
        const int i = min((int)threadIdx.x + u * NTHR, kTile * C4 - 1), rr = i / C4, c4 = i - rr * C4;
        t.v[u] = ldg4(src + (size_t)map_row(map, min(row0 + rr, nrows - 1)) * lds_ + c4 * 4);
    }
}
template <int NCOLS, int NTHR>
__device__ __forceinline__ void tile_put(const TileReq<NCOLS, NTHR>& t, float* dst, int ldd, int row0, int nrows) {
    constexpr int C4 = NCOLS / 4;
#pragma unroll
    for (int u = 0; u < TileReq<NCOLS, NTHR>::NV; ++u) {
        const int i = (int)threadIdx.x + u * NTHR, rr = i / C4, c4 = i - rr * C4;
        if (i < kTile * C4)
            *reinterpret_cast<float4*>(dst + rr * ldd + c4 * 4) = row0 + rr < nrows ? t.v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// LayerNorm forward over the 128 columns of the 4 rows this wave owns (rows 4w..4w+3 of
// an LDS tile).  y = xhat * g + b is written back into the tile; xhat / rstd / y go to
// global (row-major [N,128]) when the pointers are non-null.
__device__ __forceinline__ void ln_fwd_rows(float* ts, int ld, const float* __restrict__ g, const float* __restrict__ b,
                                            int row0, int nvalid, float* xhat_g, float* rstd_g, float* y_g,
                                            const RowMap& ymap) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float g0 = ldg(g + lane), g1 = ldg(g + lane + 64), b0 = ldg(b + lane), b1 = ldg(b + lane + 64);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = w * 4 + rr;
        const float v0 = ts[row * ld + lane], v1 = ts[row * ld + lane + 64];
        const float mean = wave_sum(v0 + v1) * (1.0f / kD);
        const float d0 = v0 - mean, d1 = v1 - mean;
        const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / kD);
        const float rstd = 1.0f / sqrtf(var + kLnEps);
        const float x0 = d0 * rstd, x1 = d1 * rstd;
        const float y0 = x0 * g0 + b0, y1 = x1 * g1 + b1;
        ts[row * ld + lane] = y0;
        ts[row * ld + lane + 64] = y1;
        const int n = row0 + row;
        if (row < nvalid) {
            if (xhat_g) {
                stg(xhat_g + (size_t)n * kD + lane, x0);
                stg(xhat_g + (size_t)n * kD + lane + 64, x1);
                if (lane == 0) stg(rstd_g + n, rstd);
            }
            if (y_g) {
                const size_t o = (size_t)map_row(ymap, n) * kD;
                stg(y_g + o + lane, y0);
                stg(y_g + o + lane + 64, y1);
            }
        }
    }
}

// same, with the lane's four LayerNorm parameters {g[lane], g[lane+64], b[lane], b[lane+64]} already in registers
__device__ __forceinline__ void ln_fwd_rows_r(float* ts, int ld, const float (&gb)[4], int row0, int nvalid, float* xhat_g,
                                              float* rstd_g, float* y_g, const RowMap& ymap) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = w * 4 + rr;
        const float v0 = ts[row * ld + lane], v1 = ts[row * ld + lane + 64];
        const float mean = wave_sum(v0 + v1) * (1.0f / kD);
        const float d0 = v0 - mean, d1 = v1 - mean;
        const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / kD);
        const float rstd = 1.0f / sqrtf(var + kLnEps);
        const float x0 = d0 * rstd, x1 = d1 * rstd;
        const float y0 = x0 * gb[0] + gb[2], y1 = x1 * gb[1] + gb[3];
        ts[row * ld + lane] = y0;
        ts[row * ld + lane + 64] = y1;
        const int n = row0 + row;
        if (row < nvalid) {
            if (xhat_g) {
                stg(xhat_g + (size_t)n * kD + lane, x0);
                stg(xhat_g + (size_t)n * kD + lane + 64, x1);
                if (lane == 0) stg(rstd_g + n, rstd);
            }
            if (y_g) {
                const size_t o = (size_t)map_row(ymap, n) * kD;
                stg(y_g + o + lane, y0);
                stg(y_g + o + lane + 64, y1);
            }
        }
    }
}

// LayerNorm backward for the wave's 4 rows: dys holds dL/dy (LDS), xh the saved xhat
// (LDS); dL/dx overwrites dys and is stored to dx_g.
__device__ __forceinline__ void ln_bwd_rows(float* dys, int ld, const float* xh, int ldx, const float* __restrict__ g,
                                            const float* __restrict__ rstd_g, int row0, int nvalid, float* dx_g) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float g0 = ldg(g + lane), g1 = ldg(g + lane + 64);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = w * 4 + rr, n = row0 + row;
        const float a0 = dys[row * ld + lane] * g0, a1 = dys[row * ld + lane + 64] * g1;
        const float x0 = xh[row * ldx + lane], x1 = xh[row * ldx + lane + 64];
        const float m1 = wave_sum(a0 + a1) * (1.0f / kD);
        const float m2 = wave_sum(a0 * x0 + a1 * x1) * (1.0f / kD);
        const float rs = (row < nvalid) ? ldg(rstd_g + n) : 0.f;
        const float o0 = rs * (a0 - m1 - x0 * m2), o1 = rs * (a1 - m1 - x1 * m2);
        dys[row * ld + lane] = o0;
        dys[row * ld + lane + 64] = o1;
        if (row < nvalid) {
            stg(dx_g + (size_t)n * kD + lane, o0);
            stg(dx_g + (size_t)n * kD + lane + 64, o1);
        }
    }
}

// ---------------------------------------------------------------------------------------
// 16-lanes-per-row LayerNorm (used by the fused kernels): a wave normalises its 4 rows at once,
// lane (row = lane >> 4, sub = lane & 15) owns columns 8*sub .. 8*sub+7, so a row reduction is four
// xor-shuffle steps inside a 16-lane group instead of six full-wave steps per row.
// ---------------------------------------------------------------------------------------
struct LnParams {
    float4 g0, g1, b0, b1;
};
__device__ __forceinline__ LnParams ln_params_load(const float* g, const float* b) {
    const int sub = threadIdx.x & 15;
    LnParams p;
    p.g0 = ldg4(g + sub * 8);
    p.g1 = ldg4(g + sub * 8 + 4);
    p.b0 = ldg4(b + sub * 8);
    p.b1 = ldg4(b + sub * 8 + 4);
    return p;
}
__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ float4 f4_sub(float4 v, float m) { return make_float4(v.x - m, v.y - m, v.z - m, v.w - m); }
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4_fma(float4 a, float4 b, float4 c) {
    return make_float4(a.x * b.x + c.x, a.y * b.y + c.y, a.z * b.z + c.z, a.w * b.w + c.w);
}
// forward: ts (LDS tile, in place -> y); xhat / rstd / y to global rows row0 + row when row < nvalid
template <bool NT = false>      // NT: streaming stores (rows nobody reads before they have left the L2 anyway: the Regulation kernels' saves)
__device__ __forceinline__ void ln_fwd_tile16(float* ts, int ld, const LnParams& P, int row0, int nvalid, float* xhat_g, float* rstd_g,
                                              float* y_g, const RowMap ymap = RowMap{1, 1, 0, 0}) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, row = w * 4 + (lane >> 4), sub = lane & 15;
    float* tp = ts + row * ld + sub * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(tp), v1 = *reinterpret_cast<const float4*>(tp + 4);
    const float mean = group16_sum(sum4(v0) + sum4(v1)) * (1.0f / kD);
    const float4 d0 = f4_sub(v0, mean), d1 = f4_sub(v1, mean);
    const float var = group16_sum(sum4(f4_mul(d0, d0)) + sum4(f4_mul(d1, d1))) * (1.0f / kD);
    const float rstd = 1.0f / sqrtf(var + kLnEps);
    const float4 x0 = f4_scale(d0, rstd), x1 = f4_scale(d1, rstd);
    const float4 y0 = f4_fma(x0, P.g0, P.b0), y1 = f4_fma(x1, P.g1, P.b1);
    *reinterpret_cast<float4*>(tp) = y0;
    *reinterpret_cast<float4*>(tp + 4) = y1;
    if (row < nvalid) {
        const size_t o = (size_t)(row0 + row) * kD + sub * 8;
        if (xhat_g) {
            if (NT) {
                stg4_nt(xhat_g + o, x0);
                stg4_nt(xhat_g + o + 4, x1);
            } else {
                stg4(xhat_g + o, x0);
                stg4(xhat_g + o + 4, x1);
            }
            if (sub == 0) stg(rstd_g + row0 + row, rstd);
        }
        if (y_g) {
            const size_t oy = (size_t)map_row(ymap, row0 + row) * kD + sub * 8;
            if (NT) {
                stg4_nt(y_g + oy, y0);
                stg4_nt(y_g + oy + 4, y1);
            } else {
                stg4(y_g + oy, y0);
                stg4(y_g + oy + 4, y1);
            }
        }
    }
}
// backward: dst = rstd * (a - mean(a) - xhat * mean(a * xhat)),  a = dy * g   (src, dst, xh: LDS tiles)
__device__ __forceinline__ void ln_bwd_tile16(const float* src, float* dst, int ld, const float* xh, int ldx, float4 g0, float4 g1,
                                              const float* rstd_g, int row0, int nvalid, float* dx_g) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, row = w * 4 + (lane >> 4), sub = lane & 15;
    const float* sp = src + row * ld + sub * 8;
    const float* xp = xh + row * ldx + sub * 8;
    const float4 a0 = f4_mul(*reinterpret_cast<const float4*>(sp), g0), a1 = f4_mul(*reinterpret_cast<const float4*>(sp + 4), g1);
    const float4 x0 = *reinterpret_cast<const float4*>(xp), x1 = *reinterpret_cast<const float4*>(xp + 4);
    const float m1 = group16_sum(sum4(a0) + sum4(a1)) * (1.0f / kD);
    const float m2 = group16_sum(sum4(f4_mul(a0, x0)) + sum4(f4_mul(a1, x1))) * (1.0f / kD);
    const float rs = row < nvalid ? ldg(rstd_g + row0 + row) : 0.f;
    const float4 o0 = make_float4(rs * (a0.x - m1 - x0.x * m2), rs * (a0.y - m1 - x0.y * m2), rs * (a0.z - m1 - x0.z * m2), rs * (a0.w - m1 - x0.w * m2));
    const float4 o1 = make_float4(rs * (a1.x - m1 - x1.x * m2), rs * (a1.y - m1 - x1.y * m2), rs * (a1.z - m1 - x1.z * m2), rs * (a1.w - m1 - x1.w * m2));
    float* dp = dst + row * ld + sub * 8;
    *reinterpret_cast<float4*>(dp) = o0;
    *reinterpret_cast<float4*>(dp + 4) = o1;
    if (row < nvalid) {
        const size_t o = (size_t)(row0 + row) * kD + sub * 8;
        stg4(dx_g + o, o0);
        stg4(dx_g + o + 4, o1);
    }
}
// ---- Rows wider than 128 (d_emb = 256, round 5): the same 16-lanes-per-row LayerNorm over D / 128 chunks of 8 floats per lane.  The
// 128-wide functions above stay as they are (the default shape compiles to the code it always had); the chain bodies pick by `D`.
// column of the i-th float4 of lane `sub` (16 lanes per row): D = 64: one float4 at 4 sub; D = 128 c: chunks of 128 columns, two float4 at 8 sub each
template <int D>
__device__ __forceinline__ int ln_col_w(int sub, int i) {
    return D == 64 ? sub * 4 : (i >> 1) * 128 + sub * 8 + (i & 1) * 4;
}
template <int D>
struct LnParamsW {
    float4 g[D / 64], b[D / 64];
};
template <int D>
__device__ __forceinline__ LnParamsW<D> ln_params_load_w(const float* g, const float* b) {
    static_assert(D == 64 || D % 128 == 0, "row widths: 64 or a multiple of 128");
    const int sub = threadIdx.x & 15;
    LnParamsW<D> p;
#pragma unroll
    for (int i = 0; i < D / 64; ++i) {
        p.g[i] = ldg4(g + ln_col_w<D>(sub, i));
        p.b[i] = ldg4(b + ln_col_w<D>(sub, i));
    }
    return p;
}
template <int D>
__device__ __forceinline__ void ln_fwd_tile16_w(float* ts, int ld, const LnParamsW<D>& P, int row0, int nvalid, float* xhat_g, float* rstd_g,
                                                float* y_g, const RowMap ymap = RowMap{1, 1, 0, 0}) {
    constexpr int NV = D / 64;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, row = w * 4 + (lane >> 4), sub = lane & 15;
    float* tp = ts + row * ld;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const float4*>(tp + ln_col_w<D>(sub, i));
        s += sum4(v[i]);
    }
    const float mean = group16_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = f4_sub(v[i], mean);
        q += sum4(f4_mul(v[i], v[i]));
    }
    const float var = group16_sum(q) * (1.0f / D);
    const float rstd = 1.0f / sqrtf(var + kLnEps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int co = ln_col_w<D>(sub, i);
        const float4 x = f4_scale(v[i], rstd), y = f4_fma(x, P.g[i], P.b[i]);
        *reinterpret_cast<float4*>(tp + co) = y;
        if (row < nvalid) {
            if (xhat_g) stg4(xhat_g + (size_t)(row0 + row) * D + co, x);
            if (y_g) stg4(y_g + (size_t)map_row(ymap, row0 + row) * D + co, y);
        }
    }
    if (row < nvalid && xhat_g && sub == 0) stg(rstd_g + row0 + row, rstd);
}
template <int D>
__device__ __forceinline__ void ln_bwd_tile16_w(const float* src, float* dst, int ld, const float* xh, int ldx, const float* __restrict__ g,
                                                const float* rstd_g, int row0, int nvalid, float* dx_g) {
    constexpr int NV = D / 64;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, row = w * 4 + (lane >> 4), sub = lane & 15;
    float4 a[NV], x[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int co = ln_col_w<D>(sub, i);
        a[i] = f4_mul(*reinterpret_cast<const float4*>(src + row * ld + co), ldg4(g + co));
        x[i] = *reinterpret_cast<const float4*>(xh + row * ldx + co);
        s1 += sum4(a[i]);
        s2 += sum4(f4_mul(a[i], x[i]));
    }
    const float m1 = group16_sum(s1) * (1.0f / D), m2 = group16_sum(s2) * (1.0f / D);
    const float rs = row < nvalid ? ldg(rstd_g + row0 + row) : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int co = ln_col_w<D>(sub, i);
        const float4 o = make_float4(rs * (a[i].x - m1 - x[i].x * m2), rs * (a[i].y - m1 - x[i].y * m2), rs * (a[i].z - m1 - x[i].z * m2),
                                     rs * (a[i].w - m1 - x[i].w * m2));
        *reinterpret_cast<float4*>(dst + row * ld + co) = o;
        if (row < nvalid) stg4(dx_g + (size_t)(row0 + row) * D + co, o);
    }
}
// column sums over the first nrows rows of an LDS tile
__device__ __forceinline__ void colsum_rows(const float* A, int lda, const float* Bt, int ldb, int ncols, int nrows, float* out) {
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
        float s = 0.f;
        for (int rr = 0; rr < nrows; ++rr) s += Bt ? A[rr * lda + c] * Bt[rr * ldb + c] : A[rr * lda + c];
        stg(out + c, s);
    }
}

// column sums over the 16 rows of an LDS tile: out[c] = sum_r A[r][c] (* Bt[r][c]); threads t0, t0 + 1, ... of the workgroup
// take the columns (t0 = 0: all threads), so that different waves can run different sums side by side
__device__ __forceinline__ void colsum16(const float* A, int lda, const float* Bt, int ldb, int ncols, float* out, int t0 = 0) {
    if ((int)threadIdx.x < t0) return;
    for (int c = threadIdx.x - t0; c < ncols; c += blockDim.x - t0) {
        float s = 0.f;
#pragma unroll
        for (int rr = 0; rr < kTile; ++rr) s += Bt ? A[rr * lda + c] * Bt[rr * ldb + c] : A[rr * lda + c];
        stg(out + c, s);
    }
}

// the same with ALL threads of the workgroup: four adjacent lanes share a column, each sums four of the 16 rows and the quad adds
// up on two DPP swaps -- 4 LDS reads per thread instead of 16 on a quarter of the threads (the sum is the same for every lane
// of the quad; fixed order: ((r0+r1+r2+r3) + (r4..r7)) + ((r8..r11) + (r12..r15)))
__device__ __forceinline__ void colsum16q(const float* A, int lda, const float* Bt, int ldb, int ncols, float* out) {
    const int rg = threadIdx.x & 3;
    for (int c = threadIdx.x >> 2; c < ncols; c += blockDim.x >> 2) {
        float s = 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = rg * 4 + rr;
            s += Bt ? A[row * lda + c] * Bt[row * ldb + c] : A[row * lda + c];
        }
        s += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
        s += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
        if (rg == 0) stg(out + c, s);
    }
}

// =======================================================================================
// Embedding: centre-row input  x0 = Wlp f_c + PE_c   (net.py:42-53 at row L//2)
// also emits the centre features padded to 8 (operand of the lin_proj weight gradient)
// =======================================================================================
struct X0Args {
    const float* feats[kMaxRes];
    const float* pe[kMaxRes];
    const float* wlp[kMaxRes];
    float* x0[kMaxRes];
    float* featc[kMaxRes];
    int L[kMaxRes];
    int F;
};
template <int D = 128>
__device__ __forceinline__ void embed_x0_row(const X0Args& a, int r, int n, int e) {
    constexpr int kD = D;      // (row width: shadows cf::kD)
    const int L = a.L[r], c = L / 2, F = a.F;
    const float* f = a.feats[r] + ((size_t)n * L + c) * F;
    float acc = 0.f;
    for (int i = 0; i < F; ++i) acc = fmaf(ldg(f + i), ldg(a.wlp[r] + e * F + i), acc);
    stg(a.x0[r] + (size_t)n * kD + e, acc + ldg(a.pe[r] + (size_t)c * kD + e));
    if (e < 8) stg(a.featc[r] + (size_t)n * 8 + e, e < F ? ldg(f + e) : 0.f);
}
template <int D = 128>
__global__ __launch_bounds__(D) void k_embed_x0(X0Args a) { embed_x0_row<D>(a, blockIdx.y, blockIdx.x, threadIdx.x); }

// =======================================================================================
// Query chain:  q = x Wq^T ;  qt[h] = q[h] Wk[h]   (the key projection absorbed into the
// single query of each sequence; modules.py:38-58 / 159-170 for one query row)
// =======================================================================================
struct QChainArgs {
    const float* x[kMaxRes];
    RowMap xmap;
    const float* wq[kMaxRes];   // [128,128]
    const float* wk[kMaxRes];   // [128,128] rows = h*64+d
    float* q[kMaxRes];          // [N,128]
    float* qt[kMaxRes];         // [N,NH,128]
    float* xcopy[kMaxRes];      // optional: the (row-mapped) input rows, materialised [N,128]
    int N;
};
// The chain kernels are written as BODIES over one 16-row tile (rows row0 .. row0 + 15 of resolution r, valid below N; LDS tiles
// passed in) with thin __global__ wrappers: the fused centre-row kernels of cf_trunk.h run the same bodies back to back in one
// workgroup per gene.
// NH: heads of the layer (d_head = 128 / NH; the default 2 everywhere but in the stand-alone launches of a configuration with 1 or 4 heads)
template <int NWV, int NH = 2, int D = 128>
__device__ __forceinline__ void qchain_fwd_body(const QChainArgs& a, const int r, const int row0, float (*xs)[D + 4], float (*qs)[D + 4]) {
    constexpr int kD = D;      // (row width: shadows cf::kD)
    constexpr int DH = kD / NH, QW = NH * kD;
    constexpr int CW = kD / NWV, NTC = CW / 16;           // q columns per wave
    constexpr int EW = QW / NWV, NTE = EW / 16;           // qt columns per wave (NH heads x 128)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int h = (w * EW) / kD, e0 = (w * EW) % kD;
    TileReq<kD, NWV * 64> tx;
    tile_req(tx, a.x[r], kD, row0, a.N, a.xmap);
    FragNT<NTC, kD / 16> fq;
    frag_load_nt(fq, a.wq[r] + (size_t)(w * CW) * kD, kD);
    FragNN<NTE, DH / 16> fk;
    frag_load_nn(fk, a.wk[r] + (size_t)(h * DH) * kD + e0, kD);
    tile_put(tx, &xs[0][0], kD + 4, row0, a.N);
    __syncthreads();
    if (a.xcopy[r])
        for (int i = threadIdx.x; i < kTile * kD; i += NWV * 64)
            if (row0 + (i / kD) < a.N) stg(a.xcopy[r] + (size_t)(row0 + (i / kD)) * kD + (i % kD), xs[i / kD][i % kD]);
    {
        f32x4 acc[NTC];
        zero_acc(acc);
        frag_mma_nt(fq, &xs[0][0], kD + 4, acc);
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * CW + col_nt(t, lr);
                qs[row][col] = acc[t][i];
                if (row0 + row < a.N) stg(a.q[r] + (size_t)(row0 + row) * kD + col, acc[t][i]);
            }
    }
    __syncthreads();
    {
        f32x4 acc[NTE];
        zero_acc(acc);
        frag_mma_nn(fk, &qs[0][h * DH], kD + 4, acc);
        typedef typename VecN<NTE>::type vec_t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lq * 4 + i;
            float v[NTE];
#pragma unroll
            for (int t = 0; t < NTE; ++t) v[t] = acc[t][i];
            if (row0 + row < a.N)
                LdgN<NTE>::st(a.qt[r] + (size_t)(row0 + row) * QW + h * kD + e0 + NTE * lr, *reinterpret_cast<const vec_t*>(v));
        }
    }
}

template <int NWV, int NH = 2, int D = 128>
__global__ __launch_bounds__(NWV * 64) void k_qchain_fwd(QChainArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[kTile][D + 4];
    __shared__ __attribute__((aligned(16))) float qs[kTile][D + 4];
    qchain_fwd_body<NWV, NH, D>(a, blockIdx.y, blockIdx.x * kTile, xs, qs);
}

struct QBwdArgs {
    const float* dqt[kMaxRes];   // [N,2,128]
    const float* dres[kMaxRes];  // [N,128] gradient arriving through the residual
    const float* wk[kMaxRes];
    const float* wq[kMaxRes];
    float* dq[kMaxRes];          // [N,128]
    float* dx[kMaxRes];          // [N,128]
    int N;
};
template <int NWV, int NH = 2, int D = 128>
__device__ __forceinline__ void qchain_bwd_body(const QBwdArgs& a, const int r, const int row0, float (*ds)[NH * D + 4], float (*qs)[D + 4]) {
    constexpr int kD = D;      // (row width: shadows cf::kD)
    constexpr int DH = kD / NH, QW = NH * kD;
    constexpr int CW = kD / NWV, NTC = CW / 16;           // output columns / tiles per wave
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int h = (w * CW) / DH;
    TileReq<QW, NWV * 64> td;
    tile_req(td, a.dqt[r], QW, row0, a.N, identity_map());
    FragNT<NTC, kD / 16> fk;
    frag_load_nt(fk, a.wk[r] + (size_t)(w * CW) * kD, kD);
    FragNN<NTC, kD / 16> fq;
    frag_load_nn(fq, a.wq[r] + w * CW, kD);
    tile_put(td, &ds[0][0], QW + 4, row0, a.N);
    __syncthreads();
    {   // dq[:, h*dh+d] = sum_e dqt[:, h, e] Wk[h*dh+d, e]
        f32x4 acc[NTC];
        zero_acc(acc);
        frag_mma_nt(fk, &ds[0][h * kD], QW + 4, acc);
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * CW + col_nt(t, lr);
                qs[row][col] = acc[t][i];
                if (row0 + row < a.N) stg(a.dq[r] + (size_t)(row0 + row) * kD + col, acc[t][i]);
            }
    }
    __syncthreads();
    {   // dx = dres + dq Wq
        f32x4 acc[NTC];
        zero_acc(acc);
        frag_mma_nn(fq, &qs[0][0], kD + 4, acc);
        typedef typename VecN<NTC>::type vec_t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lq * 4 + i;
            if (row0 + row < a.N) {
                const size_t o = (size_t)(row0 + row) * kD + w * CW + NTC * lr;
                const vec_t drv = LdgN<NTC>::ld(a.dres[r] + o);
                const float* dr = reinterpret_cast<const float*>(&drv);
                float v[NTC];
#pragma unroll
                for (int t = 0; t < NTC; ++t) v[t] = acc[t][i] + dr[t];
                LdgN<NTC>::st(a.dx[r] + o, *reinterpret_cast<const vec_t*>(v));
            }
        }
    }
}

template <int NWV, int NH = 2, int D = 128>
__global__ __launch_bounds__(NWV * 64) void k_qchain_bwd(QBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float ds[kTile][NH * D + 4];
    __shared__ __attribute__((aligned(16))) float qs[kTile][D + 4];
    qchain_bwd_body<NWV, NH, D>(a, blockIdx.y, blockIdx.x * kTile, ds, qs);
}

// =======================================================================================
// Centre-row attention over the bins of one region (one workgroup per sequence).
//   forward : score_j = (f_j . u + PE_j . qt) / sqrt(dh),  u = Wlp^T qt
//             p = softmax(masked_fill(score, -1e9));  xbar = Wlp (sum p_j f_j) + sum p_j PE_j
//   backward: the same five passes with (dxbar, p) in and (dqt, du) out.
// The features are the only HBM stream (L*F floats per sequence, read once, coalesced);
// the positional table is shared by every sequence and stays in L2.  Two layouts of it
// are kept: PE^T [128][L] for the bin-parallel pass and PE [L][128] for the
// channel-parallel pass, so that both are unit-stride across lanes.
// =======================================================================================
struct AttcArgs {
    const float* feats[kMaxRes];
    const uint8_t* mask[kMaxRes];
    long long mstride[kMaxRes];
    const float* pe[kMaxRes];
    const float* pet[kMaxRes];
    const float* wlp[kMaxRes];
    const float* vin[kMaxRes];   // fwd: qt, bwd: dxbar   [N,NH,128]
    float* p[kMaxRes];           // [N,NH,L]  fwd: out, bwd: in
    float* w[kMaxRes];           // [N,NH,8]  fwd: w (out), bwd: du (out)
    float* vout[kMaxRes];        // fwd: xbar, bwd: dqt   [N,NH,128]
    int L[kMaxRes];
    int F;
    float scale;                 // sqrt(d_head)
};

// NH simultaneous block reductions (one per head) over the four waves of the workgroup; red: 4 * NH floats of LDS scratch
template <bool IS_MAX, int NH>
__device__ __forceinline__ void block_reduce_heads(float (&v)[NH], float* red) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int h = 0; h < NH; ++h) v[h] = IS_MAX ? wave_max(v[h]) : wave_sum(v[h]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) red[4 * h + w] = v[h];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float* rh = red + 4 * h;
        v[h] = IS_MAX ? fmaxf(fmaxf(rh[0], rh[1]), fmaxf(rh[2], rh[3])) : (rh[0] + rh[1]) + (rh[2] + rh[3]);
    }
}

// The stand-alone form of the centre-row attention: one workgroup per sequence, vector ALUs, any bin count the LDS holds, NH heads
// (vin / vout [N, NH, 128], p [N, NH, L], w [N, NH, 8]).  The default configuration takes the kernels of cf_attc1.h / cf_attc2.h (two
// heads); this one runs where their LDS image does not fit and for the other head counts (1, 4).
template <bool BWD, int NH = 2, int D = 128>
__global__ __launch_bounds__(256) void k_attc(AttcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int kD = D;      // (row width: shadows cf::kD)
    static_assert(D == 64 || D == 128 || D == 256, "the channel-parallel pass splits the bins over 256 / D thread groups");
    const int r = blockIdx.y, n = blockIdx.x, tid = threadIdx.x;
    const int L = a.L[r], F = a.F;
    constexpr int QW = NH * kD;
    float* vin_s = smem;              // NH * D
    float* half_s = vin_s + QW;       // max(1, 256 / D - 1) x NH * D
    float* u_s = half_s + QW * (256 / kD > 2 ? 256 / kD - 1 : 1);         // NH * 8  (h*8+f)
    float* w_s = u_s + NH * 8;        // NH * 8
    float* red_s = w_s + NH * 8;      // NH * 4
    float* sc_s = red_s + NH * 4;     // NH*L  scores -> p (fwd) / p -> ds (bwd)
    float* dp_s = sc_s + NH * L;      // NH*L  (bwd only)
    float* feats_s = dp_s + (BWD ? NH * L : 0);   // L*F

    for (int i = tid; i < QW; i += 256) vin_s[i] = a.vin[r][(size_t)n * QW + i];
    if (tid < NH * 8) {
        u_s[tid] = 0.f;
        w_s[tid] = 0.f;
    }
    const float* fg = a.feats[r] + (size_t)n * L * F;
    for (int i = tid; i < L * F; i += 256) feats_s[i] = fg[i];
    if (BWD) {
        const float* pg = a.p[r] + (size_t)n * NH * L;
        for (int i = tid; i < NH * L; i += 256) sc_s[i] = pg[i];
    }
    __syncthreads();

    const float* wlp = a.wlp[r];
    const int grp = tid >> 4, sub = tid & 15;
    // (1) u[h][f] = sum_e vin[h][e] Wlp[e][f]
    for (int gi = grp; gi < NH * F; gi += 16) {
        const int gh = gi / F, gf = gi - gh * F;
        float s = 0.f;
        for (int e = sub; e < kD; e += 16) s = fmaf(vin_s[gh * kD + e], wlp[e * F + gf], s);
        s = group16_sum(s);
        if (sub == 0) u_s[gh * 8 + gf] = s;
    }
    __syncthreads();

    // (2) per-bin pass: t[h][j] = f_j . u[h] + PE_j . vin[h]
    const uint8_t* mrow = a.mask[r] + (size_t)n * a.mstride[r];
    const float* pet = a.pet[r];
    for (int j = tid; j < L; j += 256) {
        float sv[NH], tv[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) sv[h] = 0.f, tv[h] = 0.f;
        for (int f = 0; f < F; ++f) {
            const float fv = feats_s[j * F + f];
#pragma unroll
            for (int h = 0; h < NH; ++h) sv[h] = fmaf(fv, u_s[h * 8 + f], sv[h]);
        }
#pragma unroll 8
        for (int e = 0; e < kD; ++e) {
            const float pv = pet[(size_t)e * L + j];
#pragma unroll
            for (int h = 0; h < NH; ++h) tv[h] = fmaf(pv, vin_s[h * kD + e], tv[h]);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float sh = sv[h] + tv[h];
            if (!BWD) {
                sh = sh / a.scale;
                if (mrow[j]) sh = kMaskFill;
                sc_s[h * L + j] = sh;
            } else {
                dp_s[h * L + j] = sh;
            }
        }
    }
    __syncthreads();

    // (3) softmax (fwd) / softmax backward (bwd) over the bins, all heads at once
    if (!BWD) {
        float m[NH], z[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) m[h] = -INFINITY, z[h] = 0.f;
        for (int j = tid; j < L; j += 256)
#pragma unroll
            for (int h = 0; h < NH; ++h) m[h] = fmaxf(m[h], sc_s[h * L + j]);
        block_reduce_heads<true, NH>(m, red_s);
        for (int j = tid; j < L; j += 256)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float e = expf(sc_s[h * L + j] - m[h]);
                sc_s[h * L + j] = e;
                z[h] += e;
            }
        block_reduce_heads<false, NH>(z, red_s);
        float* pg = a.p[r] + (size_t)n * NH * L;
        for (int j = tid; j < L; j += 256)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float pv = sc_s[h * L + j] / z[h];
                sc_s[h * L + j] = pv;
                pg[h * L + j] = pv;
            }
    } else {
        float d[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) d[h] = 0.f;
        for (int j = tid; j < L; j += 256)
#pragma unroll
            for (int h = 0; h < NH; ++h) d[h] = fmaf(sc_s[h * L + j], dp_s[h * L + j], d[h]);
        block_reduce_heads<false, NH>(d, red_s);
        for (int j = tid; j < L; j += 256) {
            const bool mk = mrow[j] != 0;     // masked_fill passes no gradient
#pragma unroll
            for (int h = 0; h < NH; ++h) sc_s[h * L + j] = mk ? 0.f : sc_s[h * L + j] * (dp_s[h * L + j] - d[h]) / a.scale;
        }
    }
    __syncthreads();

    // (4) w[h][f] = sum_j sc[h][j] f_j[f]
    for (int gi = grp; gi < NH * F; gi += 16) {
        const int gh = gi / F, gf = gi - gh * F;
        float s = 0.f;
        for (int j = sub; j < L; j += 16) s = fmaf(sc_s[gh * L + j], feats_s[j * F + gf], s);
        s = group16_sum(s);
        if (sub == 0) w_s[gh * 8 + gf] = s;
    }
    // (5) pi[h][e] = sum_j sc[h][j] PE[j][e]   (256 / D parts of the bin range: two halves at the default width)
    constexpr int PARTS = 256 / kD;
    const int e = tid % kD, half = tid / kD;
    const int Lh = (L + PARTS - 1) / PARTS;
    const int j0 = half * Lh, j1 = min(L, j0 + Lh);
    const float* pe = a.pe[r];
    float cv[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) cv[h] = 0.f;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const float pv = pe[(size_t)j * kD + e];
#pragma unroll
        for (int h = 0; h < NH; ++h) cv[h] = fmaf(sc_s[h * L + j], pv, cv[h]);
    }
    if (half >= 1) {      // (parts 1 .. PARTS - 1 go through LDS, [part - 1][NH * D]; the smem budget reserves (PARTS - 1) x NH x D floats at half_s)
#pragma unroll
        for (int h = 0; h < NH; ++h) half_s[(half - 1) * QW + h * kD + e] = cv[h];
    }
    __syncthreads();
    if (tid < NH * 8) a.w[r][(size_t)n * NH * 8 + tid] = w_s[tid];
    if (half == 0) {
        float xv[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int pp = 1; pp < PARTS; ++pp) cv[h] += half_s[(pp - 1) * QW + h * kD + e];
            xv[h] = 0.f;
        }
        for (int f = 0; f < F; ++f) {
            const float wv = wlp[e * F + f];
#pragma unroll
            for (int h = 0; h < NH; ++h) xv[h] = fmaf(w_s[h * 8 + f], wv, xv[h]);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) a.vout[r][(size_t)n * QW + h * kD + e] = xv[h] + cv[h];
    }
}

// (tiled weight copies: layout comment at k_retile below)
struct RetileUnit {
    long long off;     // float offset of the unit's first row in the flat parameter buffer
    long long toff;    // float offset of the tensor
    int K, N, n0, tr;  // row length, rows of the tensor, first row of the unit, emit the transposed tiling too
};
// tp: LDS patch of NWV x 16 x 20 floats (one 16 x 16 transpose per wave)
template <int NWV>
__device__ __forceinline__ void retile_unit_lds(const float* __restrict__ params, float* __restrict__ tiled, float* __restrict__ tiledT,
                                                const RetileUnit u, float (*tp)[16][20]) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const float* src = params + u.off + (size_t)r * u.K + q * 4;
    float* dst = tiled + u.off + lane * 4;
    const bool tr = u.tr && tiledT;
    float* dstT = tiledT + u.toff + (size_t)(u.n0 / 16) * 256 + lane * 4;
    for (int kt = w; kt < u.K / 16; kt += NWV) {
        const float4 vv = ldg4(src + kt * 16);
        stg4(dst + (size_t)kt * 256, vv);
        if (tr) {      // 16 x 16 transpose through a wave-private LDS patch (LDS operations of a wave execute in order)
            *reinterpret_cast<float4*>(&tp[w][r][q * 4]) = vv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float4 o = make_float4(tp[w][q * 4][r], tp[w][q * 4 + 1][r], tp[w][q * 4 + 2][r], tp[w][q * 4 + 3][r]);
            stg4(dstT + (size_t)kt * (u.N / 16) * 256, o);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}
template <int NWV = 4>
__device__ __forceinline__ void retile_unit(const float* __restrict__ params, float* __restrict__ tiled, float* __restrict__ tiledT,
                                            const RetileUnit u) {
    __shared__ __attribute__((aligned(16))) float tp[NWV][16][20];
    retile_unit_lds<NWV>(params, tiled, tiledT, u, tp);
}
// =======================================================================================
// Post-attention chain on a 16-row tile:
//   [a = xbar[h] Wv[h]^T]  ->  y1 = LN(x + a Wo^T + bo)  ->  h = relu(y1 W1^T + b1)
//   ->  out = LN(y1 + h W2^T + b2)          (modules.py:28-30 / 150-152, 100-101)
// =======================================================================================
struct PostArgs {
    const float* x[kMaxRes];
    RowMap xmap;
    const float* ain[kMaxRes];   // VPROJ: xbar [N,2,128]   else: a [N,DM]
    const float* wv[kMaxRes];
    const float* wo[kMaxRes];
    const float* bo[kMaxRes];
    const float* g1[kMaxRes];
    const float* be1[kMaxRes];
    const float* w1[kMaxRes];
    const float* b1[kMaxRes];
    const float* w2[kMaxRes];
    const float* b2[kMaxRes];
    const float* g2[kMaxRes];
    const float* be2[kMaxRes];
    float* a_out[kMaxRes];
    float* xh1[kMaxRes];
    float* rs1[kMaxRes];
    float* y1[kMaxRes];
    float* hdn[kMaxRes];
    float* xh2[kMaxRes];
    float* rs2[kMaxRes];
    float* out[kMaxRes];
    RowMap omap;
    int N;
    int save;    // 0: inference, skip the activation saves
    const float* lin_w[kMaxRes] = {};   // optional trailing Linear without bias on the chain's output rows (tiled [128,128] copy):
    float* lin_y[kMaxRes] = {};         // lin_y = out . lin_w^T, rows [N,128] (the Embedding's layer feeds lin_proj_p, net.py:118)
    // optional: the query chain of the NEXT layer on the chain's output rows (q = out Wq^T, qt[h] = q[h] Wk[h]; k_qchain_fwd's
    // work on the tile that is in LDS anyway: one launch and one round trip through memory less per layer boundary)
    const float* nq_wq[kMaxRes] = {};   // next layer's Wq, tiled copy
    const float* nq_wk[kMaxRes] = {};   // next layer's Wk, row-major
    float* nq_q[kMaxRes] = {};          // [N,128]
    float* nq_qt[kMaxRes] = {};         // [N,2,128]
    // optional: tiled-copy units riding in this launch (workgroups with blockIdx.y >= rt_y0; unit = (blockIdx.y - rt_y0) * gridDim.x
    // + blockIdx.x).  The Embedding layer's chain kernel runs on a dozen workgroups: the Regulation + head weights, which nothing
    // reads before the Regulation stack, are re-tiled on the idle CUs beside it instead of in front of the whole forward pass.
    const RetileUnit* rt_units = nullptr;
    const float* rt_params = nullptr;
    float* rt_tiled = nullptr;
    float* rt_tiledT = nullptr;
    int rt_n = 0, rt_y0 = 1 << 30;
};

// NWV waves per workgroup (4 or 8): wave w owns 128 / NWV output columns of every product (DFF / NWV of the hidden layer), so with
// eight waves two of them share a SIMD and one's operand waits hide behind the other's products; the LayerNorms run on waves 0..3.
template <int DFF, int NH = 2, int D = 128>
struct PostFwdLds {      // LDS tiles of post_fwd_body
    static constexpr int HW0 = (DFF > 256 ? DFF : 256);
    static constexpr int HW = (NH * D > HW0 ? NH * D : HW0);      // (the hidden tile first holds the NH x D attention rows)
};
template <bool VPROJ, int DM, int DFF, int NWV, int NH = 2, int D = 128>
__device__ __forceinline__ void post_fwd_body(const PostArgs& a, const int r, const int row0, const int N, float (*xs)[D + 4],
                                              float (*as_)[DM + 4], float (*ts)[D + 4], float (*hs)[PostFwdLds<DFF, NH, D>::HW + 4]) {
    constexpr int kD = D;      // (row width: shadows cf::kD)
    constexpr int HW = PostFwdLds<DFF, NH, D>::HW;
    constexpr int DH = kD / NH, QW = NH * kD;
    constexpr int CW = kD / NWV, NTC = CW / 16;           // output columns / column tiles per wave
    constexpr int CH = DFF / NWV, NT1 = CH / 16;          // hidden columns / tiles per wave
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    TileReq<kD, NWV * 64> tx;
    TileReq<QW, NWV * 64> th;
    tile_req(tx, a.x[r], kD, row0, N, a.xmap);
    if (VPROJ) tile_req(th, a.ain[r], QW, row0, N, identity_map());
    FragNT<NTC, DM / 16> fo;
    FragNT<NTC, kD / 16> fv;
    if (VPROJ) frag_load_nt(fv, a.wv[r] + (size_t)(w * CW) * kD, kD);
    frag_load_nt(fo, a.wo[r] + (size_t)(w * CW) * DM, DM);
    typename std::conditional<D == 128, LnParams, LnParamsW<D>>::type lnp1, lnp2;
    if (w < 4) {
        if constexpr (D == 128) {
            lnp1 = ln_params_load(a.g1[r], a.be1[r]);
            lnp2 = ln_params_load(a.g2[r], a.be2[r]);
        } else {
            lnp1 = ln_params_load_w<D>(a.g1[r], a.be1[r]);
            lnp2 = ln_params_load_w<D>(a.g2[r], a.be2[r]);
        }
    }
    tile_put(tx, &xs[0][0], kD + 4, row0, N);
    if (VPROJ) {
        tile_put(th, &hs[0][0], HW + 4, row0, N);
        __syncthreads();
        const int h = (w * CW) / DH;
        f32x4 acc[NTC];
        zero_acc(acc);
        frag_mma_nt(fv, &hs[0][h * kD], HW + 4, acc);
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * CW + col_nt(t, lr);
                as_[row][col] = acc[t][i];
                if (a.save && row0 + row < N) {
                    if (CF_TRUNK_NT & 1) stg_nt(a.a_out[r] + (size_t)(row0 + row) * DM + col, acc[t][i]);
                    else stg(a.a_out[r] + (size_t)(row0 + row) * DM + col, acc[t][i]);
                }
            }
    } else {
        load_tile(&as_[0][0], DM + 4, a.ain[r], DM, DM, row0, N, identity_map());
    }
    __syncthreads();
    {   // t1 = x + a Wo^T + bo
        f32x4 acc[NTC];
        zero_acc(acc);
        frag_mma_nt(fo, &as_[0][0], DM + 4, acc);
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * CW + col_nt(t, lr);
                ts[row][col] = acc[t][i] + ldg(a.bo[r] + col) + xs[row][col];
            }
    }
    FragNT<NT1, kD / 16> f1;      // issued before the LayerNorm so the L2 latency hides behind it
    frag_load_nt(f1, a.w1[r] + (size_t)(w * CH) * kD, kD);
    __syncthreads();
    if (w < 4) {
        if constexpr (D == 128)
            ln_fwd_tile16<(CF_TRUNK_NT & 1) != 0>(&ts[0][0], kD + 4, lnp1, row0, min(kTile, N - row0), a.save ? a.xh1[r] : nullptr, a.rs1[r], a.save ? a.y1[r] : nullptr);
        else
            ln_fwd_tile16_w<D>(&ts[0][0], kD + 4, lnp1, row0, min(kTile, N - row0), a.save ? a.xh1[r] : nullptr, a.rs1[r], a.save ? a.y1[r] : nullptr);
    }
    __syncthreads();
    {   // hdn = relu(y1 W1^T + b1)
        f32x4 acc[NT1];
        zero_acc(acc);
        frag_mma_nt(f1, &ts[0][0], kD + 4, acc);
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * CH + col_nt(t, lr);
                const float v = fmaxf(acc[t][i] + ldg(a.b1[r] + col), 0.f);
                hs[row][col] = v;
                if (a.save && row0 + row < N) {
                    if (CF_TRUNK_NT & 1) stg_nt(a.hdn[r] + (size_t)(row0 + row) * DFF + col, v);
                    else stg(a.hdn[r] + (size_t)(row0 + row) * DFF + col, v);
                }
            }
    }
    FragNT<NTC, DFF / 16> f2;      // (RING = 8 for this K = 256 product, everything in flight at once: measured, no change)
    frag_load_nt(f2, a.w2[r] + (size_t)(w * CW) * DFF, DFF);
    __syncthreads();
    {   // t2 = y1 + hdn W2^T + b2
        f32x4 acc[NTC];
        zero_acc(acc);
        frag_mma_nt(f2, &hs[0][0], HW + 4, acc);
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * CW + col_nt(t, lr);
                xs[row][col] = acc[t][i] + ldg(a.b2[r] + col) + ts[row][col];
            }
    }
    const bool lin = a.lin_w[r] != nullptr, nq = a.nq_wq[r] != nullptr;
    FragNT<NTC, kD / 16> fl;      // (operand ring of the trailing Linear, or of the next layer's q projection: never both)
    if (lin) frag_load_nt(fl, a.lin_w[r] + (size_t)(w * CW) * kD, kD);
    if (nq) frag_load_nt(fl, a.nq_wq[r] + (size_t)(w * CW) * kD, kD);
    __syncthreads();
    if (w < 4) {
        if constexpr (D == 128) ln_fwd_tile16(&xs[0][0], kD + 4, lnp2, row0, min(kTile, N - row0), a.save ? a.xh2[r] : nullptr, a.rs2[r], a.out[r], a.omap);
        else ln_fwd_tile16_w<D>(&xs[0][0], kD + 4, lnp2, row0, min(kTile, N - row0), a.save ? a.xh2[r] : nullptr, a.rs2[r], a.out[r], a.omap);
    }
    if (nq) {
        constexpr int EW = QW / NWV, NTE = EW / 16;           // qt columns per wave (NH heads x D)
        const int hq = (w * EW) / kD, e0 = (w * EW) % kD;
        FragNN<NTE, DH / 16> fk;
        frag_load_nn(fk, a.nq_wk[r] + (size_t)(hq * DH) * kD + e0, kD);
        __syncthreads();
        {   // q = out Wq^T   (ts is free: the last reader was the FFN's second product)
            f32x4 acc[NTC];
            zero_acc(acc);
            frag_mma_nt(fl, &xs[0][0], kD + 4, acc);
#pragma unroll
            for (int t = 0; t < NTC; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = lq * 4 + i, col = w * CW + col_nt(t, lr);
                    ts[row][col] = acc[t][i];
                    if (row0 + row < N) stg(a.nq_q[r] + (size_t)(row0 + row) * kD + col, acc[t][i]);
                }
        }
        __syncthreads();
        {
            f32x4 acc[NTE];
            zero_acc(acc);
            frag_mma_nn(fk, &ts[0][hq * DH], kD + 4, acc);
            typedef typename VecN<NTE>::type vec_t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i;
                float v[NTE];
#pragma unroll
                for (int t = 0; t < NTE; ++t) v[t] = acc[t][i];
                if (row0 + row < N)
                    LdgN<NTE>::st(a.nq_qt[r] + (size_t)(row0 + row) * QW + hq * kD + e0 + NTE * lr, *reinterpret_cast<const vec_t*>(v));
            }
        }
    }
    if (lin) {      // one launch less than a separate Linear kernel on 4 row tiles
        __syncthreads();
        f32x4 acc[NTC];
        zero_acc(acc);
        frag_mma_nt(fl, &xs[0][0], kD + 4, acc);
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * CW + col_nt(t, lr);
                if (row0 + row < N) stg(a.lin_y[r] + (size_t)(row0 + row) * kD + col, acc[t][i]);
            }
    }
}

// RT: the instantiation that hosts tiled-copy units (PostArgs::rt_units) -- a template parameter, so that the other instantiations
// keep their code (the extra path cost k_post_fwd<true, 128, 256, 8> 2.4 us per launch when it was a run-time branch in all of them)
template <bool VPROJ, int DM, int DFF, int NWV, bool RT = false, int NH = 2, int D = 128>
__global__ __launch_bounds__(NWV * 64) void k_post_fwd(PostArgs a) {
    constexpr int HW = PostFwdLds<DFF, NH, D>::HW;
    __shared__ __attribute__((aligned(16))) float xs[kTile][D + 4];
    __shared__ __attribute__((aligned(16))) float as_[kTile][DM + 4];
    __shared__ __attribute__((aligned(16))) float ts[kTile][D + 4];
    __shared__ __attribute__((aligned(16))) float hs[kTile][HW + 4];
    if (RT && (int)blockIdx.y >= a.rt_y0) {      // a tiled-copy unit riding in this launch
        const int u = ((int)blockIdx.y - a.rt_y0) * gridDim.x + blockIdx.x;
        if (u < a.rt_n) retile_unit<NWV>(a.rt_params, a.rt_tiled, a.rt_tiledT, a.rt_units[u]);
        return;
    }
    post_fwd_body<VPROJ, DM, DFF, NWV, NH, D>(a, blockIdx.y, blockIdx.x * kTile, a.N, xs, as_, ts, hs);
}

// backward of the chain.  Per tile it also emits the column sums that make up the
// bias / LayerNorm gradients:  partial[tile] = [dg2 | db2' | dbias2 | dbias1(DFF) | dg1 | db1' | dbo]
struct PostBwdArgs {
    const float* dout[kMaxRes];
    RowMap dmap;
    const float* xh2[kMaxRes];
    const float* rs2[kMaxRes];
    const float* g2[kMaxRes];
    const float* hdn[kMaxRes];
    const float* w2[kMaxRes];
    const float* w1[kMaxRes];
    const float* xh1[kMaxRes];
    const float* rs1[kMaxRes];
    const float* g1[kMaxRes];
    const float* wo[kMaxRes];
    const float* wv[kMaxRes];
    float* dt2[kMaxRes];
    float* dpre1[kMaxRes];
    float* dt1[kMaxRes];
    float* da[kMaxRes];
    float* dxbar[kMaxRes];
    float* partial[kMaxRes];
    int N;
};
__host__ __device__ constexpr int post_partial_width(int dff, int d = 128) { return 6 * d + dff; }

// part_row: row of the partial buffer this tile's column sums go to (the tile index; the gene in the fused kernels)
template <bool VPROJ, int DM, int DFF, int NWV, int NH = 2, int D = 128>
__device__ __forceinline__ void post_bwd_body(const PostBwdArgs& a, const int r, const int row0, const int N, const int part_row,
                                              float (*ds)[D + 4], float (*xh)[D + 4], float (*t2)[D + 4],
                                              float (*wide)[(DFF > DM ? DFF : DM) + 4]) {
    constexpr int kD = D;      // (row width: shadows cf::kD)
    constexpr int WW = (DFF > DM ? DFF : DM);
    constexpr int CW = kD / NWV, NTC = CW / 16;           // columns / tiles per wave of a 128-wide product
    constexpr int CH = DFF / NWV, NT2 = CH / 16;          // ... of the hidden layer
    constexpr int CO = DM / NWV, NTO = CO / 16;           // ... of the attention output
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    float* part = a.partial[r] + (size_t)part_row * post_partial_width(DFF, kD);
    // With eight waves the LayerNorm backwards run on waves 0..3 while waves 4..7 take the column sums of the same phase.
    constexpr bool SPLIT = NWV == 8;
    TileReq<kD, NWV * 64> td, tx, tx1;
    tile_req(td, a.dout[r], kD, row0, N, a.dmap);
    tile_req(tx, a.xh2[r], kD, row0, N, identity_map());
    FragNN<NT2, kD / 16> fw2;
    frag_load_nn(fw2, a.w2[r] + w * CH, DFF);
    tile_req(tx1, a.xh1[r], kD, row0, N, identity_map());      // xhat1 rows: needed three phases later, requested now
    tile_put(td, &ds[0][0], kD + 4, row0, N);
    tile_put(tx, &xh[0][0], kD + 4, row0, N);
    __syncthreads();
    colsum16(&ds[0][0], kD + 4, &xh[0][0], kD + 4, kD, part + 0, SPLIT ? 256 : 0);      // d ln2.weight  (waves 4, 5)
    colsum16(&ds[0][0], kD + 4, nullptr, 0, kD, part + kD, SPLIT ? 384 : 0);            // d ln2.bias    (waves 6, 7)
    if (w < 4) {
        const int sub = threadIdx.x & 15;
        if constexpr (D == 128)
            ln_bwd_tile16(&ds[0][0], &t2[0][0], kD + 4, &xh[0][0], kD + 4, ldg4(a.g2[r] + sub * 8), ldg4(a.g2[r] + sub * 8 + 4), a.rs2[r], row0,
                          min(kTile, N - row0), a.dt2[r]);   // t2 = dt2
        else
            ln_bwd_tile16_w<D>(&ds[0][0], &t2[0][0], kD + 4, &xh[0][0], kD + 4, a.g2[r], a.rs2[r], row0, min(kTile, N - row0), a.dt2[r]);
    }
    FragNN<NTC, DFF / 16> fw1;      // (RING = 8 for this K = 256 product: measured, no change)
    frag_load_nn(fw1, a.w1[r] + w * CW, kD);
    __syncthreads();
    colsum16q(&t2[0][0], kD + 4, nullptr, 0, kD, part + 2 * kD);       // d l2.bias
    {   // dpre1 = (dt2 W2) * (hdn > 0)
        f32x4 acc[NT2];
        zero_acc(acc);
        frag_mma_nn(fw2, &t2[0][0], kD + 4, acc);
        typedef typename VecN<NT2>::type vec_t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lq * 4 + i, col = w * CH + NT2 * lr;
            float v[NT2];
#pragma unroll
            for (int t = 0; t < NT2; ++t) v[t] = 0.f;
            if (row0 + row < N) {
                const size_t o = (size_t)(row0 + row) * DFF + col;
                const vec_t hv = LdgN<NT2>::ld(a.hdn[r] + o);
                const float* hp = reinterpret_cast<const float*>(&hv);
#pragma unroll
                for (int t = 0; t < NT2; ++t) v[t] = hp[t] > 0.f ? acc[t][i] : 0.f;
                LdgN<NT2>::st(a.dpre1[r] + o, *reinterpret_cast<const vec_t*>(v));
            }
            *reinterpret_cast<vec_t*>(&wide[row][col]) = *reinterpret_cast<const vec_t*>(v);
        }
    }
    tile_put(tx1, &xh[0][0], kD + 4, row0, N);   // xhat2 is dead now
    FragNN<NTO, kD / 16> fwo;
    frag_load_nn(fwo, a.wo[r] + w * CO, DM);
    __syncthreads();
    colsum16q(&wide[0][0], WW + 4, nullptr, 0, DFF, part + 3 * kD);    // d l1.bias
    {   // dy1 = dt2 + dpre1 W1
        f32x4 acc[NTC];
        zero_acc(acc);
        frag_mma_nn(fw1, &wide[0][0], WW + 4, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lq * 4 + i, col = w * CW + NTC * lr;
#pragma unroll
            for (int t = 0; t < NTC; ++t) ds[row][col + t] = acc[t][i] + t2[row][col + t];
        }
    }
    __syncthreads();
    colsum16(&ds[0][0], kD + 4, &xh[0][0], kD + 4, kD, part + 3 * kD + DFF, SPLIT ? 256 : 0);   // d ln1.weight
    colsum16(&ds[0][0], kD + 4, nullptr, 0, kD, part + 4 * kD + DFF, SPLIT ? 384 : 0);          // d ln1.bias
    float (*d1)[kD + 4] = SPLIT ? t2 : ds;      // dt1: into the dead dt2 tile when the sums above still read dy1, else in place
    if (!SPLIT) __syncthreads();
    if (w < 4) {
        const int sub = threadIdx.x & 15;
        if constexpr (D == 128)
            ln_bwd_tile16(&ds[0][0], &d1[0][0], kD + 4, &xh[0][0], kD + 4, ldg4(a.g1[r] + sub * 8), ldg4(a.g1[r] + sub * 8 + 4), a.rs1[r], row0,
                          min(kTile, N - row0), a.dt1[r]);
        else
            ln_bwd_tile16_w<D>(&ds[0][0], &d1[0][0], kD + 4, &xh[0][0], kD + 4, a.g1[r], a.rs1[r], row0, min(kTile, N - row0), a.dt1[r]);
    }
    __syncthreads();
    colsum16q(&d1[0][0], kD + 4, nullptr, 0, kD, part + 5 * kD + DFF);    // d out-proj bias
    {   // da = dt1 Wo
        f32x4 acc[NTO];
        zero_acc(acc);
        frag_mma_nn(fwo, &d1[0][0], kD + 4, acc);
        typedef typename VecN<NTO>::type vec_t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lq * 4 + i, col = w * CO + NTO * lr;
            float v[NTO];
#pragma unroll
            for (int t = 0; t < NTO; ++t) v[t] = acc[t][i];
            *reinterpret_cast<vec_t*>(&wide[row][col]) = *reinterpret_cast<const vec_t*>(v);
            if (row0 + row < N) LdgN<NTO>::st(a.da[r] + (size_t)(row0 + row) * DM + col, *reinterpret_cast<const vec_t*>(v));
        }
    }
    if (VPROJ) {   // dxbar[:, h, e] = sum_d da[:, h*dh+d] Wv[h*dh+d, e]
        constexpr int DH = kD / NH, QW = NH * kD;
        constexpr int EW = QW / NWV, NTE = EW / 16;
        const int h = (w * EW) / kD, e0 = (w * EW) % kD;
        FragNN<NTE, DH / 16> fwv;
        frag_load_nn(fwv, a.wv[r] + (size_t)(h * DH) * kD + e0, kD);
        __syncthreads();
        f32x4 acc[NTE];
        zero_acc(acc);
        frag_mma_nn(fwv, &wide[0][h * DH], WW + 4, acc);
        typedef typename VecN<NTE>::type vec_t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lq * 4 + i;
            float v[NTE];
#pragma unroll
            for (int t = 0; t < NTE; ++t) v[t] = acc[t][i];
            if (row0 + row < N)
                LdgN<NTE>::st(a.dxbar[r] + (size_t)(row0 + row) * QW + h * kD + e0 + NTE * lr, *reinterpret_cast<const vec_t*>(v));
        }
    }
}

template <bool VPROJ, int DM, int DFF, int NWV, int NH = 2, int D = 128>
__global__ __launch_bounds__(NWV * 64) void k_post_bwd(PostBwdArgs a) {
    constexpr int WW = (DFF > DM ? DFF : DM);
    __shared__ __attribute__((aligned(16))) float ds[kTile][D + 4];    // dout -> dy1 -> dt1
    __shared__ __attribute__((aligned(16))) float xh[kTile][D + 4];    // xhat2 -> xhat1
    __shared__ __attribute__((aligned(16))) float t2[kTile][D + 4];    // dt2
    __shared__ __attribute__((aligned(16))) float wide[kTile][WW + 4];  // dpre1 -> da
    post_bwd_body<VPROJ, DM, DFF, NWV, NH, D>(a, blockIdx.y, blockIdx.x * kTile, a.N, blockIdx.x, ds, xh, t2, wide);
}

// =======================================================================================
// Generic Linear forward / input-gradient on 16-row tiles (used where no chain applies:
// Regulation q|k|v|gate projection, lin_proj_p, the head).
// =======================================================================================
struct LinArgs {
    const float* x[kMaxRes];
    RowMap xmap;
    int ldx;
    const float* w[kMaxRes];     // [Nout, K]
    const float* b[kMaxRes];     // may be null
    float* y[kMaxRes];
    int ldy;
    int N, K, Nout, relu;
};
constexpr int kKChunk0 = 128;   // reduction columns per fragment set / LDS stage (KCH = 64 for K = 64: rows of d_emb = 64)
template <int NT, int KCH = kKChunk0>
__global__ __launch_bounds__(256) void k_linear_fwd(LinArgs a) {
    constexpr int kKChunk = KCH;
    __shared__ __attribute__((aligned(16))) float xs[2][kTile][kKChunk + 4];
    const int r = blockIdx.z, row0 = blockIdx.x * kTile, K = a.K;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int col0 = blockIdx.y * (64 * NT) + w * (16 * NT);
    const bool active = col0 < a.Nout;
    const float* wp = a.w[r] + (size_t)(active ? col0 : 0) * K;
    f32x4 acc[NT];
    zero_acc(acc);
    int buf = 0;
    for (int kc = 0; kc < K; kc += kKChunk, buf ^= 1) {
        FragNT<NT, kKChunk / 16> f;
        frag_load_nt(f, wp + (size_t)kc * 16, K);     // tiled layout: reduction offset kc -> (kc/16)*256 floats
        load_tile(&xs[buf][0][0], kKChunk + 4, a.x[r] + kc, a.ldx, kKChunk, row0, a.N, a.xmap);
        __syncthreads();
        frag_mma_nt(f, &xs[buf][0][0], kKChunk + 4, acc);
    }
    if (!active) return;
    const float* bias = a.b[r];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = lq * 4 + i, col = col0 + col_nt(t, lr);
            if (row0 + row < a.N) {
                float v = acc[t][i] + (bias ? bias[col] : 0.f);
                if (a.relu) v = fmaxf(v, 0.f);
                a.y[r][(size_t)(row0 + row) * a.ldy + col] = v;
            }
        }
}

struct DgradArgs {
    const float* dy[kMaxRes];    // [N, K]
    int lddy;
    const float* w[kMaxRes];     // [K, ldw]  (the Linear weight [out=K, in=Ncols])
    int ldw;
    const float* res[kMaxRes];   // optional residual gradient, [.., ldres] through rmap
    RowMap rmap;
    int ldres;
    float* dx[kMaxRes];
    int lddx;
    int N, K, Ncols;
};
// One workgroup = 16 rows x 32 output columns; the reduction dimension K is split over the
// four waves (K/4 each, operands straight from L2 into registers, no LDS staging because
// nothing is shared), partial tiles are summed through LDS.  K % 64 == 0, KS = K/64.
template <int KS>
__global__ __launch_bounds__(256) void k_dgrad(DgradArgs a) {
    __shared__ __attribute__((aligned(16))) float red[4][kTile][32 + 1];
    const int r = blockIdx.z, row0 = blockIdx.x * kTile, col0 = blockIdx.y * 32;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int kw = w * KS * 16;                       // this wave's slice of K
    const bool rv = row0 + lr < a.N;
    const float* ap = a.dy[r] + (size_t)(rv ? row0 + lr : 0) * a.lddy + kw + lq * 4;
    const float* bp = a.w[r] + (size_t)(kw + lq * 4) * a.ldw + col0 + 2 * lr;
    float4 ra[3][2];
    float2 rb[3][2][4];
    auto fetch = [&](int slot, int chunk) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            ra[slot][k] = *reinterpret_cast<const float4*>(ap + (chunk * 2 + k) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rb[slot][k][i] = *reinterpret_cast<const float2*>(bp + (size_t)((chunk * 2 + k) * 16 + i) * a.ldw);
        }
    };
    constexpr int NC = KS / 2;
    fetch(0, 0);
    if (NC > 1) fetch(1, 1);
    f32x4 acc[2];
    zero_acc(acc);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + 2 < NC) fetch((c + 2) % 3, c + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float4 av = ra[c % 3][k];
            const float a4[4] = {rv ? av.x : 0.f, rv ? av.y : 0.f, rv ? av.z : 0.f, rv ? av.w : 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0] = mfma4(a4[i], rb[c % 3][k][i].x, acc[0]);
                acc[1] = mfma4(a4[i], rb[c % 3][k][i].y, acc[1]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[w][lq * 4 + i][2 * lr + t] = acc[t][i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < kTile * 32; idx += 256) {
        const int row = idx >> 5, c = idx & 31, col = col0 + c;
        if (row0 + row < a.N && col < a.Ncols) {
            float v = (red[0][row][c] + red[1][row][c]) + (red[2][row][c] + red[3][row][c]);
            if (a.res[r]) v += a.res[r][(size_t)map_row(a.rmap, row0 + row) * a.ldres + col];
            a.dx[r][(size_t)(row0 + row) * a.lddx + col] = v;
        }
    }
}

// =======================================================================================
// Regulation attention: T <= 17 tokens, H heads x DM / H, gate + frequency bias
// (modules.py:38-81).  One workgroup per (gene, resolution).  The default shape (8 heads x 32, d_model 256) with T <= 16 runs
// inside the fused Regulation kernels (cf_reg8.h / cf_reg_fused.h); this kernel is the stand-alone stage of the layer-by-layer
// path: longer token rows, and the other head counts / widths check_config accepts (H in {4, 8}, DM in {128, 256}: run-time here).
// =======================================================================================
constexpr int kRH = 8, kRDh = 32, kRDm = 256, kRW = 1024, kRMaxT = 17;      // the default shape (what the fused kernels are written for)
struct AttrArgs {
    const float* qkvg[kMaxRes];      // [B*T, 4 DM]  q | k | v | gate
    const uint8_t* mask[kMaxRes];    // [B, T, T]
    const float* freq;               // [B, T, T]
    const float* gamma[kMaxRes];     // [H]
    float* p[kMaxRes];               // [B, H, T, T]
    float* a[kMaxRes];               // fwd out [B*T, DM]; bwd in: da
    float* dqkvg[kMaxRes];           // bwd out
    float* dgam[kMaxRes];            // bwd out [B, H] per-gene partials
    int T;
    int H, DM;                       // heads, d_model (H * d_head)
};

template <bool BWD>
__global__ __launch_bounds__(256) void k_attr(AttrArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int r = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, T = a.T, TT = T * T;
    const int H = a.H, DM = a.DM, RW = 4 * DM, DH = DM / H;
    float* x_s = smem;                       // [T][4 DM]
    float* p_s = x_s + T * RW;               // [H][T][T]
    float* ds_s = p_s + H * TT;              // [H][T][T]     (bwd)
    float* do_s = ds_s + (BWD ? H * TT : 0);     // [T][DM]   (bwd)
    float* red_s = do_s + (BWD ? T * DM : 0);    // H*T       (bwd)
    const float scale = sqrtf((float)DH);
    const float* xg = a.qkvg[r] + (size_t)g * T * RW;
    for (int i = tid; i < T * RW / 4; i += 256) reinterpret_cast<float4*>(x_s)[i] = reinterpret_cast<const float4*>(xg)[i];
    if (BWD) {
        const float* pg = a.p[r] + (size_t)g * H * TT;
        for (int i = tid; i < H * TT; i += 256) p_s[i] = pg[i];
    }
    __syncthreads();
    const uint8_t* mk = a.mask[r] + (size_t)g * TT;
    const float* fq = a.freq + (size_t)g * TT;
    if (!BWD) {
        for (int idx = tid; idx < H * TT; idx += 256) {
            const int h = idx / TT, ij = idx - h * TT, i = ij / T, j = ij - i * T;
            const float* qp = x_s + i * RW + h * DH;
            const float* kp = x_s + j * RW + DM + h * DH;
            float s = 0.f;
#pragma unroll 16
            for (int d = 0; d < DH; ++d) s = fmaf(qp[d], kp[d], s);
            s = s / scale + a.gamma[r][h] * fq[ij];
            if (mk[ij]) s = kMaskFill;
            p_s[idx] = s;
        }
        __syncthreads();
        for (int row = tid; row < H * T; row += 256) {
            float* pr = p_s + row * T;
            float m = -INFINITY;
            for (int j = 0; j < T; ++j) m = fmaxf(m, pr[j]);
            float z = 0.f;
            for (int j = 0; j < T; ++j) {
                const float e = expf(pr[j] - m);
                pr[j] = e;
                z += e;
            }
            for (int j = 0; j < T; ++j) pr[j] = pr[j] / z;
        }
        __syncthreads();
        float* pg = a.p[r] + (size_t)g * H * TT;
        for (int i = tid; i < H * TT; i += 256) pg[i] = p_s[i];
        for (int idx = tid; idx < T * DM; idx += 256) {
            const int i = idx / DM, c = idx - i * DM, h = c / DH;
            const float* pr = p_s + (h * T + i) * T;
            float o = 0.f;
            for (int j = 0; j < T; ++j) o = fmaf(pr[j], x_s[j * RW + 2 * DM + c], o);
            const float gt = x_s[i * RW + 3 * DM + c];
            const float sg = 1.0f / (1.0f + expf(-gt));
            a.a[r][((size_t)g * T + i) * DM + c] = o * sg;
        }
    } else {
        float* dx = a.dqkvg[r] + (size_t)g * T * RW;
        const float* dag = a.a[r] + (size_t)g * T * DM;
        // gate / value-side: do = da * sigmoid(g); dgate = da * o * s(1-s)
        for (int idx = tid; idx < T * DM; idx += 256) {
            const int i = idx / DM, c = idx - i * DM, h = c / DH;
            const float* pr = p_s + (h * T + i) * T;
            float o = 0.f;
            for (int j = 0; j < T; ++j) o = fmaf(pr[j], x_s[j * RW + 2 * DM + c], o);
            const float gt = x_s[i * RW + 3 * DM + c];
            const float sg = 1.0f / (1.0f + expf(-gt));
            const float da = dag[idx];
            do_s[idx] = da * sg;
            dx[i * RW + 3 * DM + c] = da * o * sg * (1.0f - sg);
        }
        __syncthreads();
        // dp[h][i][j] = do[i][h] . v[j][h]
        for (int idx = tid; idx < H * TT; idx += 256) {
            const int h = idx / TT, ij = idx - h * TT, i = ij / T, j = ij - i * T;
            const float* dp = do_s + i * DM + h * DH;
            const float* vp = x_s + j * RW + 2 * DM + h * DH;
            float s = 0.f;
#pragma unroll 16
            for (int d = 0; d < DH; ++d) s = fmaf(dp[d], vp[d], s);
            ds_s[idx] = s;
        }
        __syncthreads();
        for (int row = tid; row < H * T; row += 256) {
            const float* pr = p_s + row * T;
            float* dr = ds_s + row * T;
            const int i = row % T;
            float dot = 0.f;
            for (int j = 0; j < T; ++j) dot = fmaf(pr[j], dr[j], dot);
            float gsum = 0.f;
            for (int j = 0; j < T; ++j) {
                const float v = mk[i * T + j] ? 0.f : pr[j] * (dr[j] - dot);
                gsum = fmaf(v, fq[i * T + j], gsum);
                dr[j] = v;
            }
            red_s[row] = gsum;
        }
        __syncthreads();
        if (tid < H) {
            float s = 0.f;
            for (int i = 0; i < T; ++i) s += red_s[tid * T + i];
            a.dgam[r][(size_t)g * H + tid] = s;
        }
        // dq, dk, dv
        for (int idx = tid; idx < T * DM; idx += 256) {
            const int i = idx / DM, c = idx - i * DM, h = c / DH;
            float dq = 0.f, dk = 0.f, dv = 0.f;
            for (int j = 0; j < T; ++j) {
                dq = fmaf(ds_s[(h * T + i) * T + j], x_s[j * RW + DM + c], dq);     // ds[i][j] k[j]
                dk = fmaf(ds_s[(h * T + j) * T + i], x_s[j * RW + c], dk);          // ds[j][i] q[j]
                dv = fmaf(p_s[(h * T + j) * T + i], do_s[j * DM + c], dv);          // p[j][i] do[j]
            }
            dx[i * RW + c] = dq / scale;
            dx[i * RW + DM + c] = dk / scale;
            dx[i * RW + 2 * DM + c] = dv;
        }
    }
}

// =======================================================================================
// Head  (net.py:377-380; loss train.py:156, 193)
// =======================================================================================
// (the head kernels live in cf_head.h)

// Joins the gradient streams that meet at the promoter embedding e_c of a gene:
//   dxp0[g] = sum_s dxP[g*S+s]       (lin_proj_p output is broadcast over the S slots)
//   resid[g] = dX0[g, token 0] + dhin[g, r]   (regulation input + final skip, net.py:378)
struct JoinArgs {
    const float* dxp[kMaxRes];   // [B*S,128]
    const float* dx0[kMaxRes];   // [B*T,128]
    const float* dhin;           // [B, n_res*128]
    float* dxp0[kMaxRes];        // [B,128]
    float* resid[kMaxRes];       // [B,128]
    int S, T, n_res;
};
template <int D = 128>
__global__ __launch_bounds__(D) void k_join(JoinArgs a) {
    constexpr int kD = D;      // (row width: shadows cf::kD)
    const int g = blockIdx.x, r = blockIdx.y, e = threadIdx.x;
    float s = 0.f;
    for (int i = 0; i < a.S; ++i) s += a.dxp[r][((size_t)g * a.S + i) * kD + e];
    a.dxp0[r][(size_t)g * kD + e] = s;
    a.resid[r][(size_t)g * kD + e] = a.dx0[r][(size_t)g * a.T * kD + e] + a.dhin[(size_t)g * (a.n_res * kD) + r * kD + e];
}

// The join and the input gradient of lin_proj_p in one launch: dxp0[g] = sum_slots dxp[g, slot] (written for the weight
// gradient by the first column block), edout[g] = dxp0[g] . W + (dx0[g, token 0] + dhin[g, r]).  One workgroup = 16 genes x 32
// output columns, K = 128 split over the four waves as in k_dgrad<2>; the A operand is summed from the S slot rows on the fly
// (same order as k_join: slot 0 first), so the result is bit-identical to k_join + k_dgrad.
struct JoinDgradArgs {
    const float* dxp[kMaxRes];   // [B*S,128]
    const float* dx0[kMaxRes];   // [B*T,128]
    const float* dhin;           // [B, n_res*128]
    const float* w[kMaxRes];     // lin_proj_p.weight [128,128] row-major
    float* dxp0[kMaxRes];        // [B,128]
    float* dx[kMaxRes];          // [B,128]
    int B, S, T, n_res;
};
template <int D = 128>
__global__ __launch_bounds__(256) void k_join_dgrad(JoinDgradArgs a) {
    constexpr int kD = D, KB = D / 64;      // (row width: shadows cf::kD; 16-deep k-blocks per wave)
    __shared__ __attribute__((aligned(16))) float red[4][kTile][32 + 1];
    const int r = blockIdx.z, row0 = blockIdx.x * kTile, col0 = blockIdx.y * 32;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int kw = w * (kD / 4);                      // this wave's slice of K = D
    const bool rv = row0 + lr < a.B;
    const int g = rv ? row0 + lr : 0;
    const float* bp = a.w[r] + (size_t)(kw + lq * 4) * kD + col0 + 2 * lr;
    float2 rb[KB][4];
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[k][i] = ldg2(bp + (size_t)(k * 16 + i) * kD);
    float4 ra[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const float* ap = a.dxp[r] + (size_t)g * a.S * kD + kw + k * 16 + lq * 4;
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < a.S; ++i) {
            const float4 v = ldg4(ap + (size_t)i * kD);
            sacc = make_float4(sacc.x + v.x, sacc.y + v.y, sacc.z + v.z, sacc.w + v.w);
        }
        ra[k] = rv ? sacc : make_float4(0.f, 0.f, 0.f, 0.f);
        if (rv && blockIdx.y == 0) stg4(a.dxp0[r] + (size_t)g * kD + kw + k * 16 + lq * 4, sacc);
    }
    f32x4 acc[2];
    zero_acc(acc);
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const float a4[4] = {ra[k].x, ra[k].y, ra[k].z, ra[k].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[0] = mfma4(a4[i], rb[k][i].x, acc[0]);
            acc[1] = mfma4(a4[i], rb[k][i].y, acc[1]);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[w][lq * 4 + i][2 * lr + t] = acc[t][i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < kTile * 32; idx += 256) {
        const int row = idx >> 5, c = idx & 31, col = col0 + c, gg = row0 + row;
        if (gg < a.B) {
            const float v = (red[0][row][c] + red[1][row][c]) + (red[2][row][c] + red[3][row][c]);
            const float res = a.dx0[r][(size_t)gg * a.T * kD + col] + a.dhin[(size_t)gg * (a.n_res * kD) + r * kD + col];
            a.dx[r][(size_t)gg * kD + col] = v + res;
        }
    }
}

// =======================================================================================
// Deferred weight gradients: one launch, a table of 64x64 output tiles.
//   C[n][k] = sum over segments, sum_m A[m][n] * Bm[m][k]      (dW = dY^T X)
// =======================================================================================
template <int N, class F>
__device__ __forceinline__ void static_for(F f) {      // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
// AdamW folded into the epilogue of the gradient reductions (single GPU: nothing stands between a finished gradient element and
// its update): every element of a bucket's gradients is produced exactly once, by one weight-gradient or column-sum tile, so the
// tile that holds it in registers applies torch.optim.AdamW's update (adamw_range below, same operation order) to the
// parameter / moment elements at the same offset of the flat buffers.  The 21 MB of gradients, the 116 + 32 MB optimiser
// streams of two launches and their launch boundaries become part of the two reductions.
struct AdamFuse {
    float *p, *m, *v;        // flat parameter / moment buffers
    const float* gbase;      // flat gradient buffer: a tile's output pointer minus this is the element's offset in all four
    float decay, one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps;
    int keep_grads;          // also store the gradient (callers that read it back)
    float* tiled;            // optional: the tiled weight copies (same flat offsets as p); tiles that carry a tensor offset (WgTile::toff >= 0) write
                             // the updated elements there too, in fragment order -- the next forward pass then has nothing to re-tile (cf_keep_tiled)
};
__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, const AdamFuse& o) {
#pragma clang fp contract(off)      // every product and sum rounded on its own, wherever this is inlined: the stand-alone and the fused form agree bit for bit
    p = p * o.decay;
    m = m + o.one_m_b1 * (g - m);
    v = v * o.b2 + o.one_m_b2 * g * g;
    const float denom = sqrtf(v) / o.bc2_sqrt + o.eps;
    p = p - o.step_size * (m / denom);
}
struct WgSeg {
    const float* A;
    const float* B;
    int lda, ldb;
    int rows_per_gene;    // M = rows_per_gene * batch
};
struct WgTile {
    WgSeg seg[4];
    int nseg;
    float* C;
    int ldc, Nn, Kk;
    int n0, k0;
    long long toff;       // flat offset of the tensor C lies in when that tensor has a tiled copy kept fresh by the optimiser epilogue, else -1
    int trow0, pad_;      // row of the tensor C's row 0 is
};
// four consecutive floats of a row, zero beyond `ncols`; 16-byte load when the address allows it
__device__ __forceinline__ float4 wg_load4(const float* __restrict__ p, int col, int ncols, bool vec_ok) {
    if (vec_ok && col + 3 < ncols) return ldg4(p + col);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < ncols) v.x = ldg(p + col);
    if (col + 1 < ncols) v.y = ldg(p + col + 1);
    if (col + 2 < ncols) v.z = ldg(p + col + 2);
    if (col + 3 < ncols) v.w = ldg(p + col + 3);
    return v;
}
#ifndef CF_WG_TK
#define CF_WG_TK 64      // 128 measured: 91.6 vs 78.1 us per step (half as many workgroups: the barriers of a stage are no longer hidden)
#endif
// Streaming (nt) loads and stores for the optimiser state in the tile epilogues: every element is touched once per step, and as plain
// accesses the 128 MB of them sweep the L2s that the operand rows of the other tiles -- and, under the riders, the tables of the trunk --
// live in.  A/B on one box: riders only 0.568 -> 0.565 ms, all epilogues 0.567 -> 0.558.
#ifndef CF_RIDER_OPND_NT      // the riders' operand rows as well (round 4, 151 MB streamed through the L2s the trunk works in: 0.5415 -> 0.5388 ms).  Round 6, on the
#define CF_RIDER_OPND_NT 0    // kernels as they are now, the other way round: plain loads 0.5035 -> 0.5006 ms and 40 MB less HBM-side traffic per step; with plain
#endif                        // accesses for the riders' optimiser state too (CF_RIDER_NT 0) 0.4984 ms, 274 -> 225 MB in `k_trunk_bwd` (profiles/r06k_traffic_ab.txt)
#ifndef CF_ADAM_NT      // the stand-alone AdamW stream (data-parallel path) likewise
#define CF_ADAM_NT 1
#endif
#ifndef CF_OPT_NT
#define CF_OPT_NT 1
#endif
#ifndef CF_RIDER_NT
#define CF_RIDER_NT 0
#endif
#ifndef CF_WG_M
#define CF_WG_M 32
#endif
constexpr int kWgM = CF_WG_M;       // reduction rows per LDS stage
constexpr int kWgTk = CF_WG_TK;     // tile width along K (columns of dW): 64 or 128; tiles are 64 (n) x kWgTk (k)
constexpr int kWgLdA = 64 + 16;     // A stage row stride: a half-wave's scalar reads (two rows x 16 columns) hit 32 distinct banks
constexpr int kWgLdB = kWgTk + 4;
// (component-wise: a select between two float4 OBJECTS is compiled to a select between their addresses -- both spilled to scratch)
__device__ __forceinline__ float4 f4_keep_if(bool keep, const float4& v) {
    return make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f);
}
// A tile whose 64 x kWgTk block lies inside the matrix and whose operands allow 16-byte loads (every tile of the default
// configuration but the head's last layer) fetches its stages WITHOUT branches: wg_load4's edge handling puts every load into a
// conditional of its own, and the compiler then waits for each one before it issues the next -- a stage of four loads per thread
// was four round trips, 6 K cycles around 1 K cycles of MFMA issue (tools/reg_stamps-style stamps of the riders; the instruction
// mix of a stage: 27 branches, 24 scalar and 4 vector loads).
__device__ __forceinline__ bool wg_tile_interior(const WgTile& t) {
    bool ok = t.n0 + 64 <= t.Nn && t.k0 + kWgTk <= t.Kk;
    for (int s = 0; s < t.nseg; ++s) {
        const WgSeg& sg = t.seg[s];
        ok = ok && ((sg.lda | sg.ldb | t.n0 | t.k0) & 3) == 0 && ((reinterpret_cast<uintptr_t>(sg.A) | reinterpret_cast<uintptr_t>(sg.B)) & 15) == 0;
    }
    return ok;
}
// As / Bs: the LDS stage of the 256 threads that work on this tile, tid: their index (the tile of a 256-thread workgroup, or one of
// the two teams of a 512-thread rider workgroup of k_trunk_bwd, which walk equally long tiles in lock step: the barriers are the
// workgroup's)
// D: stages whose global loads are in flight (a register ring).  1 where several workgroups share a CU and hide each other's round
// trips; the riders of k_trunk_bwd are alone on their CU (two teams, cold operands: a stage took 6 K cycles for 1 K of MFMA issue
// with D = 1) and run with D = 4.
template <bool OPT, int D = 1>
__device__ __forceinline__ void wgrad_tile_impl(const WgTile& t, int batch, const AdamFuse* o, float (*As)[kWgM * kWgLdA], float (*Bs)[kWgM * kWgLdB],
                                                const int tid) {
    // 64 x kWgTk output tile; wave w owns rows n0+16w..+15 and all the tile's columns.  Both operands are staged through LDS
    // (every element of dY is used by one wave but every element of X by all four: reading X straight from L2 in each wave
    // made the kernel L1-bound at 37 % of the MFMA peak), one stage of 32 reduction rows (single-buffered: several workgroups
    // per CU hide the two barriers better than a double buffer with fewer did); the global loads of stage i+1 are in
    // flight while stage i is multiplied.  B is read 16 bytes per lane, so accumulator 4 h + c holds the strided columns
    // k0 + 64 h + 4 r + c.  A 64 x 128 tile moves (64 + 128) x 32 floats per 2 x 64 x 128 x 32 flops: 21 flop per byte out of
    // L2 against 16 for a 64 x 64 tile.
    constexpr int NH = kWgTk / 64;
    const int w = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int kc = t.k0 + 4 * lr;
    f32x4 acc[4 * NH];
    zero_acc(acc);
    const int sm = tid >> 4, sc = (tid & 15) * 4;          // staging: rows sm, sm + 16; columns sc .. sc + 3 (+ 64 h)
    const bool interior = wg_tile_interior(t);
    auto run = [&](auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    for (int s = 0; s < t.nseg; ++s) {
        const WgSeg sg = t.seg[s];
        const int M = sg.rows_per_gene * batch, nst = (M + kWgM - 1) / kWgM;
        const bool va = ((sg.lda | t.n0) & 3) == 0 && (reinterpret_cast<uintptr_t>(sg.A) & 15) == 0;
        const bool vb = ((sg.ldb | t.k0) & 3) == 0 && (reinterpret_cast<uintptr_t>(sg.B) & 15) == 0;
        float4 ra[D][kWgM / 16], rb[D][kWgM / 16][NH];
        auto fetch = [&](int st, auto slot_c) {
            constexpr int sl = decltype(slot_c)::value;
#pragma unroll
            for (int p = 0; p < kWgM / 16; ++p) {
                const int m = st * kWgM + sm + 16 * p;
                if (FAST) {      // no branch around a load (see wg_tile_interior): rows past the end re-read the last one and are zeroed by a select
                    const int mc = min(m, M - 1);      // (the select sits in put(): nothing here needs a loaded value)
                    ra[sl][p] = ldg4(sg.A + (size_t)mc * sg.lda + t.n0 + sc);
#pragma unroll
                    for (int h = 0; h < NH; ++h) rb[sl][p][h] = ldg4(sg.B + (size_t)mc * sg.ldb + t.k0 + 64 * h + sc);
                } else if (m < M) {
                    ra[sl][p] = wg_load4(sg.A + (size_t)m * sg.lda, t.n0 + sc, t.Nn, va);
#pragma unroll
                    for (int h = 0; h < NH; ++h) rb[sl][p][h] = wg_load4(sg.B + (size_t)m * sg.ldb, t.k0 + 64 * h + sc, t.Kk, vb);
                } else {
                    ra[sl][p] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int h = 0; h < NH; ++h) rb[sl][p][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto put = [&](int st, auto slot_c) {
            constexpr int sl = decltype(slot_c)::value;
#pragma unroll
            for (int p = 0; p < kWgM / 16; ++p) {
                const bool live = !FAST || st * kWgM + sm + 16 * p < M;
                *reinterpret_cast<float4*>(&As[0][(sm + 16 * p) * kWgLdA + sc]) = f4_keep_if(live, ra[sl][p]);
#pragma unroll
                for (int h = 0; h < NH; ++h) *reinterpret_cast<float4*>(&Bs[0][(sm + 16 * p) * kWgLdB + 64 * h + sc]) = f4_keep_if(live, rb[sl][p][h]);
            }
        };
        static_for<D>([&](auto u) {
            if (decltype(u)::value < nst) fetch(decltype(u)::value, u);
        });
        for (int st0 = 0; st0 < nst; st0 += D) static_for<D>([&](auto u) {
            const int st = st0 + decltype(u)::value;
            if (st >= nst) return;
            __syncthreads();                                   // the previous stage has been consumed
            put(st, u);
            __syncthreads();
            if (st + D < nst) fetch(st + D, u);                // in flight during the next D stages' multiplies
            const float* ap = &As[0][lq * kWgLdA + 16 * w + lr];
            const float* bp = &Bs[0][lq * kWgLdB + 4 * lr];
#pragma unroll
            for (int i = 0; i < kWgM / 4; ++i) {
                const float av = ap[4 * i * kWgLdA];
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float4 bv = *reinterpret_cast<const float4*>(bp + 4 * i * kWgLdB + 64 * h);
                    acc[4 * h + 0] = mfma4(av, bv.x, acc[4 * h + 0]);
                    acc[4 * h + 1] = mfma4(av, bv.y, acc[4 * h + 1]);
                    acc[4 * h + 2] = mfma4(av, bv.z, acc[4 * h + 2]);
                    acc[4 * h + 3] = mfma4(av, bv.w, acc[4 * h + 3]);
                }
            }
        });
    }
    };
    if (interior) run(std::true_type{});
    else run(std::false_type{});
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        if (kc + 64 * h >= t.Kk) continue;
        // OPT: the optimiser state of all four rows is requested at once (one round trip, not four: a load inside the row loop is
        // waited for before the next row's is issued); rows past the end re-read the last one and are not stored
        float4 pp[4], mm[4], vv[4];
        if (OPT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = min(t.n0 + w * 16 + lq * 4 + i, t.Nn - 1);
                const size_t off = (size_t)(t.C + (size_t)row * t.ldc + kc + 64 * h - o->gbase);
                pp[i] = CF_OPT_NT ? ldg4_nt(o->p + off) : ldg4(o->p + off);
                mm[i] = CF_OPT_NT ? ldg4_nt(o->m + off) : ldg4(o->m + off);
                vv[i] = CF_OPT_NT ? ldg4_nt(o->v + off) : ldg4(o->v + off);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = t.n0 + w * 16 + lq * 4 + i;
            if (row < t.Nn) {
                float* cp = t.C + (size_t)row * t.ldc + kc + 64 * h;
                const float4 g4 = make_float4(acc[4 * h + 0][i], acc[4 * h + 1][i], acc[4 * h + 2][i], acc[4 * h + 3][i]);
                if (OPT) {
                    const size_t off = (size_t)(cp - o->gbase);
                    adamw_elem(pp[i].x, g4.x, mm[i].x, vv[i].x, *o);
                    adamw_elem(pp[i].y, g4.y, mm[i].y, vv[i].y, *o);
                    adamw_elem(pp[i].z, g4.z, mm[i].z, vv[i].z, *o);
                    adamw_elem(pp[i].w, g4.w, mm[i].w, vv[i].w, *o);
                    if (CF_OPT_NT) stg4_nt(o->p + off, pp[i]); else stg4(o->p + off, pp[i]);
                    if (CF_OPT_NT) stg4_nt(o->m + off, mm[i]); else stg4(o->m + off, mm[i]);
                    if (CF_OPT_NT) stg4_nt(o->v + off, vv[i]); else stg4(o->v + off, vv[i]);
                    if (o->keep_grads) stg4(cp, g4);
                    if (o->tiled && t.toff >= 0) {      // the same four elements in the tiled copy: block (row / 16, k / 16), lane slot (k % 16 / 4, row % 16)
                        const int ra = t.trow0 + row, k = kc + 64 * h;
                        stg4(o->tiled + t.toff + ((size_t)(ra >> 4) * (t.ldc >> 4) + (k >> 4)) * 256 + (((k & 15) >> 2) * 16 + (ra & 15)) * 4, pp[i]);
                    }
                } else {
                    stg4(cp, g4);
                }
            }
        }
    }
}
// The same tile by ONE wave (the riders of k_trunk_bwd, cf_trunk.h): all four 16-row blocks of the tile in sixteen accumulators, a
// wave-private LDS stage (kWgWaveLds floats), no workgroup barrier -- eight waves of a workgroup walk eight tiles at their own pace
// instead of two four-wave teams in lock step (a team's stage is ~6 K cycles of round trips and barriers around 1 K cycles of MFMA
// issue; only other tiles on the same CU fill that, and a rider workgroup is alone on its CU).  Every element sums its products in
// the order of the four-wave routine: bit-identical results.  A float4 of B feeds 16 MFMAs instead of 4.
constexpr int kWgWaveLds = kWgM * kWgLdA + kWgM * kWgLdB;
template <bool OPT>
__device__ __forceinline__ void wgrad_tile_wave(const WgTile& t, int batch, const AdamFuse* o, float* lds, unsigned long long* stamp = nullptr) {
    static_assert(kWgTk == 64, "one 64 x 64 tile per wave");
    const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    float* As = lds;
    float* Bs = lds + kWgM * kWgLdA;
    const int kc = t.k0 + 4 * lr;
    f32x4 acc[4][4];      // [16-row block][column quarter c: columns k0 + 4 lr + c]
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) zero_acc(acc[rt]);
    const int sm = lane >> 4, sc = (lane & 15) * 4;          // staging: rows sm + 4 p; columns sc .. sc + 3
    constexpr int NP = kWgM / 4;
    const bool interior = wg_tile_interior(t);
    auto run = [&](auto fast_c) {
        constexpr bool FAST = decltype(fast_c)::value;
        for (int s = 0; s < t.nseg; ++s) {
            const WgSeg sg = t.seg[s];
            const int M = sg.rows_per_gene * batch, nst = (M + kWgM - 1) / kWgM;
            const bool va = ((sg.lda | t.n0) & 3) == 0 && (reinterpret_cast<uintptr_t>(sg.A) & 15) == 0;
            const bool vb = ((sg.ldb | t.k0) & 3) == 0 && (reinterpret_cast<uintptr_t>(sg.B) & 15) == 0;
            float4 ra[NP], rb[NP];
            auto fetch = [&](int st) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int m = st * kWgM + sm + 4 * p;
                    if (FAST) {      // (see wg_tile_interior; rows past the end re-read the last one and are zeroed on their way into LDS)
                        const int mc = min(m, M - 1);
                        ra[p] = CF_RIDER_OPND_NT ? ldg4_nt(sg.A + (size_t)mc * sg.lda + t.n0 + sc) : ldg4(sg.A + (size_t)mc * sg.lda + t.n0 + sc);
                        rb[p] = CF_RIDER_OPND_NT ? ldg4_nt(sg.B + (size_t)mc * sg.ldb + t.k0 + sc) : ldg4(sg.B + (size_t)mc * sg.ldb + t.k0 + sc);
                    } else {
                        ra[p] = rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (m < M) {
                            ra[p] = wg_load4(sg.A + (size_t)m * sg.lda, t.n0 + sc, t.Nn, va);
                            rb[p] = wg_load4(sg.B + (size_t)m * sg.ldb, t.k0 + sc, t.Kk, vb);
                        }
                    }
                }
            };
            fetch(0);
            for (int st = 0; st < nst; ++st) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (LDS operations of a wave execute in order; these keep the compiler from reordering them)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const bool live = !FAST || st * kWgM + sm + 4 * p < M;
                    *reinterpret_cast<float4*>(&As[(sm + 4 * p) * kWgLdA + sc]) = f4_keep_if(live, ra[p]);
                    *reinterpret_cast<float4*>(&Bs[(sm + 4 * p) * kWgLdB + sc]) = f4_keep_if(live, rb[p]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_sched_barrier(0);      // (the next stage's loads stay BEHIND the LDS writes: hoisted above them, the writes' wait would cover one of the new loads)
                if (st + 1 < nst) fetch(st + 1);
                __builtin_amdgcn_sched_barrier(0);
                const float* ap = &As[lq * kWgLdA + lr];
                const float* bp = &Bs[lq * kWgLdB + 4 * lr];
#pragma unroll
                for (int i = 0; i < kWgM / 4; ++i) {
                    const float4 bv = *reinterpret_cast<const float4*>(bp + 4 * i * kWgLdB);
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        const float av = ap[4 * i * kWgLdA + 16 * rt];
                        acc[rt][0] = mfma4(av, bv.x, acc[rt][0]);
                        acc[rt][1] = mfma4(av, bv.y, acc[rt][1]);
                        acc[rt][2] = mfma4(av, bv.z, acc[rt][2]);
                        acc[rt][3] = mfma4(av, bv.w, acc[rt][3]);
                    }
                }
            }
        }
    };
    if (stamp) stamp[0] = __builtin_amdgcn_s_memtime();
    if (interior) run(std::true_type{});
    else run(std::false_type{});
    if (stamp) stamp[1] = __builtin_amdgcn_s_memtime();
    if (kc >= t.Kk) return;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        float4 pp[4], mm[4], vv[4];      // (the optimiser state of a 16-row block at once: four round trips per tile instead of sixteen)
        if (OPT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = min(t.n0 + rt * 16 + lq * 4 + i, t.Nn - 1);
                const size_t off = (size_t)(t.C + (size_t)row * t.ldc + kc - o->gbase);
                pp[i] = CF_RIDER_NT ? ldg4_nt(o->p + off) : ldg4(o->p + off);
                mm[i] = CF_RIDER_NT ? ldg4_nt(o->m + off) : ldg4(o->m + off);
                vv[i] = CF_RIDER_NT ? ldg4_nt(o->v + off) : ldg4(o->v + off);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = t.n0 + rt * 16 + lq * 4 + i;
            if (row < t.Nn) {
                float* cp = t.C + (size_t)row * t.ldc + kc;
                const float4 g4 = make_float4(acc[rt][0][i], acc[rt][1][i], acc[rt][2][i], acc[rt][3][i]);
                if (OPT) {
                    const size_t off = (size_t)(cp - o->gbase);
                    adamw_elem(pp[i].x, g4.x, mm[i].x, vv[i].x, *o);
                    adamw_elem(pp[i].y, g4.y, mm[i].y, vv[i].y, *o);
                    adamw_elem(pp[i].z, g4.z, mm[i].z, vv[i].z, *o);
                    adamw_elem(pp[i].w, g4.w, mm[i].w, vv[i].w, *o);
                    if (CF_RIDER_NT) stg4_nt(o->p + off, pp[i]); else stg4(o->p + off, pp[i]);
                    if (CF_RIDER_NT) stg4_nt(o->m + off, mm[i]); else stg4(o->m + off, mm[i]);
                    if (CF_RIDER_NT) stg4_nt(o->v + off, vv[i]); else stg4(o->v + off, vv[i]);
                    if (o->keep_grads) stg4(cp, g4);
                } else {
                    stg4(cp, g4);
                }
            }
        }
    }
}
template <bool OPT = false>
__device__ __forceinline__ void wgrad_tile(const WgTile& t, int batch, const AdamFuse* o = nullptr) {
    __shared__ __attribute__((aligned(16))) float As[1][kWgM * kWgLdA];
    __shared__ __attribute__((aligned(16))) float Bs[1][kWgM * kWgLdB];
    wgrad_tile_impl<OPT>(t, batch, o, As, Bs, threadIdx.x);
}

// XCD-aware tile order.  Consecutive workgroup ids go round-robin over the 8 XCDs (each with a private L2), and the tiles of one
// weight-gradient job -- (n0, k0) blocks of the same dY^T X product, adjacent in the table -- share their operands: every X
// column block is read by Nn / 64 tiles, every dY column block by Kk / 64.  With the table order as launch order a job's tiles
// are spread over all eight L2s and each fetches its own copy (PMC: 2.6x the algorithmic bytes).  Workgroup b therefore takes
// tile (b mod 8) * per + b / 8, per = ceil(n / 8): XCD x works through a contiguous eighth of the table, so a job's tiles meet
// in one L2.  The launch has 8 * per workgroups; those past the table's end return.
// MEASURED (round 3): slower -- k_reduce 38.4 -> 40.6 us per launch on a box whose other kernels ran 2.7 % faster: with a job's
// tiles in one XCD all of them walk the same rows of the same arrays in lock step and queue on the same L2 channels, and the
// re-reads the table order causes are served by the 256 MB Infinity Cache anyway (the arrays were written by the kernel in
// front).  Off by default there (`on` = 0: table order).  In the fused reduction + AdamW launch of the single-GPU step
// (k_reduce_opt behind the riders: 796 tiles + the optimiser state, 312 MB HBM-side in 56 us = 5.5 TB/s) bytes are what binds, and the
// same order PAID in round 3: 0.5683 -> 0.5633 ms over two A/B rounds.  Re-measured in round 6 (riders with cached accesses, 5-launch step): table order
// 0.4892 against 0.5012 ms -- off by default for that launch too.  CF_XCD_REDUCE=0 / 1 forces both.
__device__ __forceinline__ int xcd_tile(int b, int n, int on) {
    const int per = (n + 7) >> 3;
    return on ? (b & 7) * per + (b >> 3) : b;
}
__host__ __device__ inline int xcd_grid(int n) { return 8 * ((n + 7) >> 3); }
__global__ __launch_bounds__(256) void k_wgrad(const WgTile* __restrict__ tiles, int n, int batch, int xcd) {
    const int t = xcd_tile(blockIdx.x, n, xcd);
    if (t < n) wgrad_tile(tiles[t], batch);
}

// Weight gradient of the 7-mark projections (lin_proj / lin_proj_pcre, [128, F]): too narrow
// for an MFMA tile.  One workgroup sums a chunk of 8 genes into partial[chunk][128*F]; the
// chunks are added by k_colsum.   dW[e][f] = sum over segments, sum_m A[m][e] * B[m][f]
constexpr int kLpMaxSeg = 16;     // two terms per Pairwise layer: up to eight layers
struct LpJob {
    WgSeg seg[kLpMaxSeg];
    int nseg;
    float* partial;
    int F;
};
constexpr int kLpGenes = 1;
// the partial of genes [g0, g0 + kLpGenes) of one job, by 256 threads (t256 = 0 .. 255)
__device__ __forceinline__ void wgrad_lp_body(const LpJob& j, const int chunk, const int batch, const int t256, const int D = kD) {
    const int g0 = chunk * kLpGenes, g1 = min(batch, g0 + kLpGenes);
    if (g0 >= batch) return;
    const int fh = t256 >> 7;
    for (int e = t256 & 127; e < D; e += 128) {      // (one pass at the default row width)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < j.nseg; ++s) {
        const WgSeg& sg = j.seg[s];
        const int m1 = g1 * sg.rows_per_gene;
#pragma unroll 8
        for (int m = g0 * sg.rows_per_gene; m < m1; ++m) {      // (global-address-space loads: pointers out of a device table are generic)
            const float av = ldg(sg.A + (size_t)m * sg.lda + e);
            const float4 bv = ldg4(sg.B + (size_t)m * sg.ldb + fh * 4);
            acc[0] = fmaf(av, bv.x, acc[0]);
            acc[1] = fmaf(av, bv.y, acc[1]);
            acc[2] = fmaf(av, bv.z, acc[2]);
            acc[3] = fmaf(av, bv.w, acc[3]);
        }
    }
    float* out = j.partial + (size_t)chunk * (D * j.F) + e * j.F;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (fh * 4 + k < j.F) stg(out + fh * 4 + k, acc[k]);
    }
}
__global__ __launch_bounds__(256) void k_wgrad_lp(const LpJob* __restrict__ jobs, int batch, int D) {
    wgrad_lp_body(jobs[blockIdx.y], blockIdx.x, batch, threadIdx.x, D);
}

// Deferred column sums (bias / LayerNorm / gamma gradients): out[c] = sum_m src[m][c] (* src2[m][c])
struct CsTile {
    const float* src;
    float* out;
    int ld, ncols, c0;
    int rows_per_gene, div;      // M = ceil(rows_per_gene * batch / div)
    const float* src2;           // optional elementwise factor (same leading dimension): LayerNorm weight gradients
};
template <bool OPT = false>
__device__ __forceinline__ void colsum_tile(const CsTile& t, int batch, const AdamFuse* o = nullptr) {
    __shared__ float red[4][64];
    const int M = (t.rows_per_gene * batch + t.div - 1) / t.div;
    const int c = t.c0 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float s = 0.f;
    if (c < t.ncols) {
        const float* p1 = t.src + c;
        if (t.src2) {
            const float* p2 = t.src2 + c;
#pragma unroll 8
            for (int m = ph; m < M; m += 4) s += ldg(p1 + (size_t)m * t.ld) * ldg(p2 + (size_t)m * t.ld);
        } else {
#pragma unroll 8
            for (int m = ph; m < M; m += 4) s += ldg(p1 + (size_t)m * t.ld);
        }
    }
    red[ph][threadIdx.x & 63] = s;
    __syncthreads();
    if (ph == 0 && c < t.ncols) {
        const float g = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (OPT) {
            const size_t off = (size_t)(t.out + c - o->gbase);
            float pp = ldg(o->p + off), mm = ldg(o->m + off), vv = ldg(o->v + off);
            adamw_elem(pp, g, mm, vv, *o);
            stg(o->p + off, pp);
            stg(o->m + off, mm);
            stg(o->v + off, vv);
            if (o->keep_grads) stg(t.out + c, g);
        } else {
            stg(t.out + c, g);
        }
    }
}
__global__ __launch_bounds__(256) void k_colsum(const CsTile* __restrict__ tiles, int batch) { colsum_tile(tiles[blockIdx.x], batch); }
// Both reductions of a gradient bucket in ONE launch: workgroups [0, n_wg) take weight-gradient tiles, the rest column-sum
// tiles (independent of each other; the bandwidth-bound column sums run beside the matrix products instead of behind them,
// and one launch boundary goes)
__global__ __launch_bounds__(256) void k_reduce(const WgTile* __restrict__ wg, int n_wg, const CsTile* __restrict__ cs, int batch, int xcd) {
    const int nb = xcd_grid(n_wg);
    if ((int)blockIdx.x < nb) {
        const int t = xcd_tile(blockIdx.x, n_wg, xcd);
        if (t < n_wg) wgrad_tile(wg[t], batch);
    } else {
        colsum_tile(cs[blockIdx.x - nb], batch);
    }
}

// The reductions of a bucket with the AdamW update of the same bucket in their epilogues (AdamFuse above)
// (skip: the window of the table the riders of k_trunk_bwd have reduced and stepped already -- tile t of this launch is table entry t, or
//  t + skip.y from entry skip.x on)
__global__ __launch_bounds__(256) void k_reduce_opt(const WgTile* __restrict__ wg, int n_wg, const CsTile* __restrict__ cs, int batch, int xcd, AdamFuse o, int2 skip) {
    const int nb = xcd_grid(n_wg);
    if ((int)blockIdx.x < nb) {
        const int t = xcd_tile(blockIdx.x, n_wg, xcd);
        if (t < n_wg) wgrad_tile<true>(wg[t >= skip.x ? t + skip.y : t], batch, &o);
    } else {
        colsum_tile<true>(cs[blockIdx.x - nb], batch, &o);
    }
}

// =======================================================================================
// AdamW (torch.optim.AdamW defaults as used at train.py:157): decoupled decay, then the
// moment updates and the bias-corrected step, in the same operation order as torch.
// =======================================================================================
__device__ __forceinline__ void adamw_range(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                            long long n4, float decay, float one_m_b1, float b2, float one_m_b2, float step_size, float bc2_sqrt,
                                            float eps, long long first, long long stride) {
    for (long long i = first; i < n4; i += stride) {
        float4 pp = CF_ADAM_NT ? ldg4_nt(p + 4 * i) : reinterpret_cast<float4*>(p)[i];
        const float4 gg = CF_ADAM_NT ? ldg4_nt(g + 4 * i) : reinterpret_cast<const float4*>(g)[i];
        float4 mm = CF_ADAM_NT ? ldg4_nt(m + 4 * i) : reinterpret_cast<float4*>(m)[i];
        float4 vv = CF_ADAM_NT ? ldg4_nt(v + 4 * i) : reinterpret_cast<float4*>(v)[i];
        float* pa = reinterpret_cast<float*>(&pp);
        const float* ga = reinterpret_cast<const float*>(&gg);
        float* ma = reinterpret_cast<float*>(&mm);
        float* va = reinterpret_cast<float*>(&vv);
        const AdamFuse o{nullptr, nullptr, nullptr, nullptr, decay, one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) adamw_elem(pa[k], ga[k], ma[k], va[k], o);      // (one definition of the update for both forms)
        if (CF_ADAM_NT) {
            stg4_nt(p + 4 * i, pp);
            stg4_nt(m + 4 * i, mm);
            stg4_nt(v + 4 * i, vv);
        } else {
            reinterpret_cast<float4*>(p)[i] = pp;
            reinterpret_cast<float4*>(m)[i] = mm;
            reinterpret_cast<float4*>(v)[i] = vv;
        }
    }
}
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, long long n4, float decay, float one_m_b1, float b2,
                                               float one_m_b2, float step_size, float bc2_sqrt, float eps) {
    adamw_range(p, g, m, v, n4, decay, one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps, (long long)blockIdx.x * blockDim.x + threadIdx.x,
                (long long)gridDim.x * blockDim.x);
}
// The same stream over the Embedding + Pairwise range with the stepped parameters ALSO written into their tiled copies (cf_keep_tiled
// under data parallelism, where AdamW is a launch of its own behind the all-reduce): tmap[i] = float4 index of flat float4 i in the tiled
// buffer, -1 where the tensor has no tiled copy.  A flat float4 W[n][4 j .. 4 j + 3] is one float4 of the tiled layout too (element
// [q][r][0..3] of a 16 x 16 block), so the copy costs one 4-byte index load and one 16-byte store per float4.
__global__ __launch_bounds__(256) void k_adamw_tiled(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, long long n4, float decay, float one_m_b1, float b2,
                                                     float one_m_b2, float step_size, float bc2_sqrt, float eps, const int* __restrict__ tmap,
                                                     float* __restrict__ tiled) {
    const AdamFuse o{nullptr, nullptr, nullptr, nullptr, decay, one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        const int t = tmap[i];
        float* pa = reinterpret_cast<float*>(&pp);
        const float* ga = reinterpret_cast<const float*>(&gg);
        float* ma = reinterpret_cast<float*>(&mm);
        float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int k = 0; k < 4; ++k) adamw_elem(pa[k], ga[k], ma[k], va[k], o);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (t >= 0) reinterpret_cast<float4*>(tiled)[t] = pp;
    }
}
// The gradient reductions of one bucket and the AdamW update of ANOTHER bucket's (already reduced) range in one launch: workgroups
// [0, n_red) take the weight-gradient / column-sum tiles, the rest stream the optimiser state.  The Embedding + Pairwise bucket's
// reduction is a launch of ~300 latency-bound tiles (21 us with most of the chip idle), the Regulation + head bucket's AdamW a pure
// 116 MB stream (19 us): side by side they take about as long as the longer one.
struct AdamArgs {
    float *p, *m, *v;
    const float* g;
    long long n4;
    float decay, one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps;
};
__global__ __launch_bounds__(256) void k_reduce_adamw(const WgTile* __restrict__ wg, int n_wg, const CsTile* __restrict__ cs, int n_cs, int batch,
                                                      int xcd, AdamArgs o) {
    const int nb = xcd_grid(n_wg), n_red = nb + n_cs;
    if ((int)blockIdx.x < nb) {
        const int t = xcd_tile(blockIdx.x, n_wg, xcd);
        if (t < n_wg) wgrad_tile(wg[t], batch);
    } else if ((int)blockIdx.x < n_red) {
        colsum_tile(cs[blockIdx.x - nb], batch);
    } else {
        const long long nblk = gridDim.x - n_red;
        adamw_range(o.p, o.g, o.m, o.v, o.n4, o.decay, o.one_m_b1, o.b2, o.one_m_b2, o.step_size, o.bc2_sqrt, o.eps,
                    (long long)(blockIdx.x - n_red) * blockDim.x + threadIdx.x, nblk * blockDim.x);
    }
}
// The same update with the step-dependent scalars read from device memory, so that the launch can sit inside a
// replayed hipGraph (kernel arguments are frozen at capture): k_adamw_set writes them each step, outside the graph.
struct AdamHyper {
    float decay, one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps, pad;
};
__global__ void k_adamw_set(AdamHyper* dst, AdamHyper v) { *dst = v; }
__global__ __launch_bounds__(256) void k_adamw_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n4, const AdamHyper* __restrict__ hp) {
    const AdamHyper h = *hp;
    adamw_range(p, g, m, v, n4, h.decay, h.one_m_b1, h.b2, h.one_m_b2, h.step_size, h.bc2_sqrt, h.eps, (long long)blockIdx.x * blockDim.x + threadIdx.x,
                (long long)gridDim.x * blockDim.x);
}

// =======================================================================================
// Tiled copies of the Linear weights -- see FragNT.  One workgroup per 16-row unit of a tensor W[N][K] (retile_unit, above):
//   tiled:  block (n / 16, k / 16) at (n/16 * K/16 + k/16) * 256, element [q][r][m] = W[n0 + r][k0 + 4q + m]
//           (B operand of the forward products  y = x W^T)
//   tiledT: the same tiling of W^T[K][N]: block (k / 16, n / 16) at (k/16 * N/16 + n/16) * 256, element
//           [q][r][m] = W[n0 + 4q + m][k0 + r]   (B operand of the backward products  dx = dy W; Regulation only)
// =======================================================================================
__global__ __launch_bounds__(256) void k_retile(const float* __restrict__ params, float* __restrict__ tiled, float* __restrict__ tiledT,
                                                const RetileUnit* __restrict__ units) {
    retile_unit(params, tiled, tiledT, units[blockIdx.x]);
}
// First launch of a forward pass: the tiled weight copies (workgroups [0, n_units)) and, independent of them, the Embedding's
// centre-row input x0 (one workgroup per (gene, resolution) behind them) -- one launch boundary less
template <int D = 128>
__global__ __launch_bounds__(256) void k_fwd_prologue(const float* __restrict__ params, float* __restrict__ tiled, float* __restrict__ tiledT,
                                                      const RetileUnit* __restrict__ units, int n_units, X0Args a, int B) {
    if ((int)blockIdx.x < n_units) {
        retile_unit(params, tiled, tiledT, units[blockIdx.x]);
    } else if (threadIdx.x < D) {
        const int i = blockIdx.x - n_units;
        embed_x0_row<D>(a, i / B, i % B, threadIdx.x);
    }
}
// the same tiling of a stand-alone row-major matrix W[rows][K] (rows, K multiples of 16): workgroup b takes rows 16 b ..
__global__ __launch_bounds__(256) void k_retile_rows(const float* __restrict__ src, float* __restrict__ dst, int K) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const float* sp = src + (size_t)blockIdx.x * 16 * K + (size_t)r * K + q * 4;
    float* dp = dst + (size_t)blockIdx.x * 16 * K + lane * 4;
    for (int kt = w; kt < K / 16; kt += 4) stg4(dp + (size_t)kt * 256, ldg4(sp + kt * 16));
}

// Tile tables of the dense layer's split-K weight gradients, written ON the device (the arguments travel by value, so the
// launch can be captured; an upload from a host temporary needs a stream synchronisation per call).  Same entries and
// order as push_wg / push_cs on the host: tile t = (split, n0 / 64, k0 / 64).
struct DenseWgTab {
    const float* dY;
    const float* X;
    float* part;
    float* dW;
    WgTile* tiles;
    CsTile* cs;
    long long rows;
    int lddy, ldx, N_, K_, splits, split_rows;
};
__global__ __launch_bounds__(256) void k_dense_wg_tables(DenseWgTab a) {
    const int tn = (a.N_ + 63) / 64, tk = (a.K_ + kWgTk - 1) / kWgTk, per = tn * tk, nt = per * a.splits;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < nt; t += gridDim.x * 256) {
        const int s = t / per, rem = t - s * per;
        const long long r0 = (long long)s * a.split_rows;
        WgTile w;
        for (int i = 0; i < 4; ++i) w.seg[i] = WgSeg{nullptr, nullptr, 0, 0, 0};
        w.seg[0] = WgSeg{a.dY + r0 * a.lddy, a.X + r0 * a.ldx, a.lddy, a.ldx, (int)min((long long)a.split_rows, a.rows - r0)};
        w.nseg = 1;
        w.C = a.part + (size_t)s * a.N_ * a.K_;
        w.ldc = a.K_;
        w.Nn = a.N_;
        w.Kk = a.K_;
        w.n0 = (rem / tk) * 64;
        w.k0 = (rem % tk) * kWgTk;
        a.tiles[t] = w;
    }
    const int ncs = (a.N_ * a.K_ + 63) / 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ncs; i += gridDim.x * 256)
        a.cs[i] = CsTile{a.part, a.dW, a.N_ * a.K_, a.N_ * a.K_, i * 64, a.splits, 1, nullptr};
}
// Column-sum tables of the dense layer's bias / LayerNorm gradients: stage 1 sums chunks of `chunk_rows` partial rows
// (width pw) into part2[chunk][pw]; stage 2 sums the chunk rows of seven column ranges into their gradient tensors.
struct DenseCsTab {
    const float* partial;
    float* part2;
    CsTile* cs1;
    CsTile* cs2;
    int tiles_q, chunk_rows, nchunks, pw;
    int off[7], ncols[7];
    float* dst[7];
};
__global__ __launch_bounds__(256) void k_dense_cs_tables(DenseCsTab a) {
    const int per = (a.pw + 63) / 64;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.nchunks * per; i += gridDim.x * 256) {
        const int ch = i / per, c0 = (i - ch * per) * 64;
        const int rows = min(a.chunk_rows, a.tiles_q - ch * a.chunk_rows);
        a.cs1[i] = CsTile{a.partial + (size_t)ch * a.chunk_rows * a.pw, a.part2 + (size_t)ch * a.pw, a.pw, a.pw, c0, rows, 1, nullptr};
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int n = 0;
        for (int k = 0; k < 7; ++k)
            for (int c0 = 0; c0 < a.ncols[k]; c0 += 64) a.cs2[n++] = CsTile{a.part2 + a.off[k], a.dst[k], a.pw, a.ncols[k], c0, a.nchunks, 1, nullptr};
    }
}

}  // namespace cf

#include "cf_valu_mv.h"
#include "cf_head_ride.h"
#include "cf_reg_args.h"
#include "cf_reg8.h"
#include "cf_attc2.h"
#include "cf_attc1.h"
#include "cf_gather.h"
#include "cf_trunk.h"
#include "cf_head.h"
#include "cf_attn.h"
#include "cf_bin.h"
#include "cf_embed_full.h"
#if CF_TID_OPAQUE
#undef threadIdx      // the opaque thread index is a property of the kernels above, not of whatever includes this header
#endif

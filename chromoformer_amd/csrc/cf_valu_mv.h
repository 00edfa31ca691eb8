// cf_valu_mv.h -- one-row products y = W x / y = W^T v, LayerNorm of one row: the vector-ALU routines of the phases that work on ONE row
// per workgroup (the Embedding layer inside the fused trunk, cf_trunk_e.h; the prediction head of a gene at the tail of the Regulation
// forward, cf_head_ride.h).  512-thread workgroups.  (Included by cf_kernels.h in front of the kernels that use it.)
#pragma once

namespace cf {

constexpr int kMvT = 512;      // threads of the workgroups these routines are laid out for

// ---- NT form: N rows of K columns, read from the TILED copy of the weight (cf_kernels.h, k_retile: every 16 x 16 block contiguous in
//      fragment order [q][r][4], so a wave's 16-byte-per-lane load is one contiguous KB -- the row-major tensor would put 64 different
//      cache lines into every load of this mapping: measured 3.7 K cycles per product instead of ~1.5 K).  Wave w owns rows
//      16 TW w .. 16 TW (w + 1) - 1; lane (r = lane & 15, q = lane >> 4) holds W[row r of the tile][16 kt + 4 q .. + 3] for every k-block kt.
template <int N, int K>
struct NtW {
    static constexpr int TW = N / 16 / (kMvT / 64), KB = K / 16;      // 16-row tiles per wave, k-blocks
    static_assert(N % (16 * (kMvT / 64)) == 0 && K % 16 == 0, "tile split");
    float4 v[TW][KB];
};
template <int N, int K>
__device__ __forceinline__ void nt_request(NtW<N, K>& w, const float* __restrict__ Wt) {
    constexpr int TW = NtW<N, K>::TW, KB = NtW<N, K>::KB;
    const float* p = Wt + (size_t)((threadIdx.x >> 6) * TW) * 16 * K + (threadIdx.x & 63) * 4;
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int kt = 0; kt < KB; ++kt) w.v[t][kt] = ldg4(p + (size_t)t * 16 * K + kt * 256);
}
// row of tile t this lane contributes to / (after nt_dot) holds the full sum of
template <int N, int K>
__device__ __forceinline__ int nt_row(int t) {
    return ((threadIdx.x >> 6) * NtW<N, K>::TW + t) * 16 + (threadIdx.x & 15);
}
// x: LDS vector of K floats (16-byte aligned).  out[t] = y[nt_row(t)], the same bits in the four lanes of the row.
template <int N, int K>
__device__ __forceinline__ void nt_dot(const NtW<N, K>& w, const float* x, float (&out)[NtW<N, K>::TW]) {
    constexpr int TW = NtW<N, K>::TW, KB = NtW<N, K>::KB;
    const float* xp = x + ((threadIdx.x & 63) >> 4) * 4;
    float s[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) s[t] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KB; ++kt) {
        const float4 xv = *reinterpret_cast<const float4*>(xp + kt * 16);
#pragma unroll
        for (int t = 0; t < TW; ++t) s[t] = fmaf(w.v[t][kt].w, xv.w, fmaf(w.v[t][kt].z, xv.z, fmaf(w.v[t][kt].y, xv.y, fmaf(w.v[t][kt].x, xv.x, s[t]))));
    }
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        float v = s[t];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        out[t] = v;
    }
}
__device__ __forceinline__ bool nt_writer() { return (threadIdx.x & 63) < 16; }

// ---- T form: NR rows of NC columns; thread = (4 columns, one of NG row groups)
template <int NR, int NC>
struct TrW {
    static constexpr int CG = NC / 4, NG = kMvT / CG, RPG = NR / NG;
    static_assert(kMvT % CG == 0 && NR % NG == 0, "column split");
    float4 v[RPG];
};
template <int NR, int NC>
__device__ __forceinline__ void tr_request(TrW<NR, NC>& w, const float* __restrict__ W) {
    constexpr int CG = TrW<NR, NC>::CG, RPG = TrW<NR, NC>::RPG;
    const int c4 = threadIdx.x % CG, ng = threadIdx.x / CG;
    const float* p = W + (size_t)(ng * RPG) * NC + 4 * c4;
#pragma unroll
    for (int i = 0; i < RPG; ++i) w.v[i] = ldg4(p + (size_t)i * NC);
}
// v: LDS vector of NR floats; part: LDS [NG][NC].  Partial sums of the thread's row group (rows in index order).
template <int NR, int NC>
__device__ __forceinline__ void tr_partial(const TrW<NR, NC>& w, const float* v, float* part) {
    constexpr int CG = TrW<NR, NC>::CG, RPG = TrW<NR, NC>::RPG;
    const int c4 = threadIdx.x % CG, ng = threadIdx.x / CG;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const float s = v[ng * RPG + i];
        a = make_float4(fmaf(s, w.v[i].x, a.x), fmaf(s, w.v[i].y, a.y), fmaf(s, w.v[i].z, a.z), fmaf(s, w.v[i].w, a.w));
    }
    *reinterpret_cast<float4*>(part + ng * NC + 4 * c4) = a;
}
// sum of groups g0 .. g0 + n - 1 of column c, in group order
template <int NC>
__device__ __forceinline__ float tr_sum(const float* part, int g0, int n, int c) {
    float s = part[g0 * NC + c];
    for (int g = 1; g < n; ++g) s += part[(g0 + g) * NC + c];
    return s;
}

// LayerNorm of one 128-wide row by one wave (ln_fwd_rows' arithmetic): lane owns elements lane and lane + 64
struct LnRow {
    float x0, x1, y0, y1, rstd;
};
__device__ __forceinline__ LnRow ln_row_fwd(float v0, float v1, float g0, float g1, float b0, float b1) {
    const float mean = wave_sum(v0 + v1) * (1.0f / kD);
    const float d0 = v0 - mean, d1 = v1 - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.0f / kD);
    LnRow o;
    o.rstd = 1.0f / sqrtf(var + kLnEps);
    o.x0 = d0 * o.rstd, o.x1 = d1 * o.rstd;
    o.y0 = o.x0 * g0 + b0, o.y1 = o.x1 * g1 + b1;
    return o;
}
// dx = rstd * (a - mean(a) - xhat * mean(a * xhat)), a = dy * g
__device__ __forceinline__ void ln_row_bwd(float dy0, float dy1, float g0, float g1, float xh0, float xh1, float rs, float& o0, float& o1) {
    const float a0 = dy0 * g0, a1 = dy1 * g1;
    const float m1 = wave_sum(a0 + a1) * (1.0f / kD);
    const float m2 = wave_sum(a0 * xh0 + a1 * xh1) * (1.0f / kD);
    o0 = rs * (a0 - m1 - xh0 * m2);
    o1 = rs * (a1 - m1 - xh1 * m2);
}

}  // namespace cf

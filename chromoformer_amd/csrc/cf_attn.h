// cf_attn.h -- dense (all query rows) multi-head attention core, forward and backward, f32 on the matrix cores
// (included by cf_kernels.h).
//
// The training hot path never needs it (only the centre query row is consumed); it is the
// kernel of the stress configuration (800-bin sequences, all L x L rows, attention-bound)
// and of any consumer of a full-length embedding.  Semantics of MultiHeadAttention._attention /
// PairwiseMultiHeadAttention (modules.py:58-77, 170-188):
//     S = Q K^T / sqrt(dh);  S[masked] = -1e9 (replace, not add);  P = softmax(S);  O = P V
// with mask = not (qvalid x kvalid) (what the dataset produces, data.py:156-161, 180-185) or an arbitrary byte
// mask [N, Lq, Lk].  A fully masked row is a uniform distribution over all Lk keys, never NaN.
//
// Flash-style: one workgroup = 64 query rows of one (sequence, head), 16 rows per wave; keys / values stream
// through LDS in tiles of 64; the score tile never leaves registers except for the transposition of P (D layout
// -> A layout) through a per-wave LDS patch.  Backward recomputes P from the saved row statistics: one kernel per
// output owner (k_attn_bwd_kv: dK, dV for a key tile, looping over query tiles; k_attn_bwd_q: dQ for a query
// tile, looping over key tiles), so that no gradient needs atomics and the result is bit-reproducible.
#pragma once

namespace cf {

constexpr int kADh = 64;        // head width of the Embedding / Pairwise stacks
constexpr int kABq = 64;        // query rows per workgroup
constexpr int kABk = 64;        // keys per LDS tile
constexpr int kALd = kADh + 4;  // LDS row stride (floats)

struct AttnArgs {
    const float *q, *k, *v;            // [N, L, ld]: head h occupies columns [h*64, h*64 + 64)
    int ldq, ldk, ldv;
    const uint8_t *qvalid, *kvalid;    // [N, Lq], [N, Lk], 1 = real bin; null = all real
    const uint8_t* mask;               // optional [N, Lq, Lk], 1 = masked; takes precedence over the valid vectors
    float* o;                          // [N, Lq, ldo]
    int ldo;
    float* stats;                      // [N, H, Lq, 2]  row maximum m and 1 / sum exp(s - m) of the (masked, scaled) scores
                                       // (kept apart: m = -1e9 on a fully masked row would swallow log(sum) in one float)
    // backward only
    const float* d_o;                  // [N, Lq, ldo]
    float *dq, *dk, *dv;               // same layouts as q, k, v
    float* delta;                      // [N, H, Lq]  rowsum(dO * O), workspace
    int N, H, Lq, Lk;
    float rscale;                      // 1 / sqrt(dh)
};

__device__ __forceinline__ bool attn_masked(const AttnArgs& a, int n, int i, int j, bool qv, bool kv) {
    if (a.mask) return a.mask[((size_t)n * a.Lq + i) * a.Lk + j] != 0;
    return !(qv && kv);
}
__device__ __forceinline__ bool attn_kvalid(const AttnArgs& a, int n, int j) {
    return j < a.Lk && (a.kvalid ? a.kvalid[(size_t)n * a.Lk + j] != 0 : true);
}

// [64 x 64] tile of rows r0.. of one head: global -> registers -> LDS (rows beyond `rows` are zero), in two halves,
// so that the global loads of tile i+1 are in flight while tile i is multiplied
struct AttnTileRegs {
    float4 v[4];
};
__device__ __forceinline__ void attn_fetch(AttnTileRegs& t, const float* __restrict__ src, int ld, int r0, int rows) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = threadIdx.x + 256 * p, j = i >> 4, d4 = i & 15;
        t.v[p] = r0 + j < rows ? ldg4(src + (size_t)(r0 + j) * ld + d4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// (NT: streaming loads -- the one-pass backward walks 615 KB of Q / dO / dQ per workgroup once per key tile, 768 workgroups at a time: nothing it
//  fetches is found again in a cache)
template <bool NT>
__device__ __forceinline__ void attn_fetch_t(AttnTileRegs& t, const float* __restrict__ src, int ld, int r0, int rows) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = threadIdx.x + 256 * p, j = i >> 4, d4 = i & 15;
        const float* sp = src + (size_t)min(r0 + j, rows - 1) * ld + d4 * 4;
        t.v[p] = f4_keep_if(r0 + j < rows, NT ? ldg4_nt(sp) : ldg4(sp));
    }
}
__device__ __forceinline__ void attn_put(float* dst, const AttnTileRegs& t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = threadIdx.x + 256 * p, j = i >> 4, d4 = i & 15;
        *reinterpret_cast<float4*>(dst + j * kALd + d4 * 4) = t.v[p];
    }
}

// the same fetch as (scalar base of the (sequence, head)) + (32-bit lane offset): no 64-bit per-lane address lives across the tile loop
__device__ __forceinline__ void attn_fetch_s(AttnTileRegs& t, const float* seq, int ld, int r0, int rows) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = threadIdx.x + 256 * p, j = r0 + (i >> 4), d4 = i & 15;
        const float4 v = ldg4(lane_at(sbase(seq, 0), (unsigned)(min(j, rows - 1) * ld + d4 * 4) * 4u));
        t.v[p] = f4_keep_if(j < rows, v);
    }
}
// acc[t] (16 x 16 tiles, t = 0..3) += A[16 x 64] . B^T where B is a [64 x 64] LDS tile read along its rows:
// out column 16t + r <-> B row 16t + r, reduction over the 64 columns of A and B (float4 reads, permuted k order).
__device__ __forceinline__ void attn_mma_nt(const float4 (&af)[4], const float* Bs, f32x4 (&acc)[4]) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 b = *reinterpret_cast<const float4*>(Bs + (16 * t + r) * kALd + 16 * u + 4 * q);
            acc[t] = mfma4(af[u].x, b.x, acc[t]);
            acc[t] = mfma4(af[u].y, b.y, acc[t]);
            acc[t] = mfma4(af[u].z, b.z, acc[t]);
            acc[t] = mfma4(af[u].w, b.w, acc[t]);
        }
}
// acc[t] += A[16 x 64] . B where B is a [64 x 64] LDS tile in natural [k][n] layout: out column 16t + r, the
// reduction index runs over B's rows (scalar reads), A comes from an LDS patch [16][kALd] (float4 along k).
__device__ __forceinline__ void attn_mma_nn(const float* As, const float* Bs, f32x4 (&acc)[4]) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float4 av = *reinterpret_cast<const float4*>(As + r * kALd + 16 * u + 4 * q);
        const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma4(a4[i], Bs[(16 * u + 4 * q + i) * kALd + 16 * t + r], acc[t]);
    }
}
// store a D-layout tile set (rows 4q + reg, columns 16t + r) into an LDS patch [16][kALd]
__device__ __forceinline__ void attn_store_d(float* Ps, const f32x4 (&acc)[4]) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) Ps[(4 * q + g) * kALd + 16 * t + r] = acc[t][g];
}

__global__ __launch_bounds__(256, 3) void k_attn_fwd_v1(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[kABk * kALd];
    __shared__ __attribute__((aligned(16))) float Vs[kABk * kALd];
    __shared__ __attribute__((aligned(16))) float Ps[4][16 * kALd];
    __shared__ float Kvs[kABk];      // validity of the tile's keys: travels with the tile (a global byte load at the point of use stalls the softmax)
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * kABq;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, qd = lane >> 4;
    const int row_a = q0 + 16 * w + r;                  // the row this lane feeds as A operand
    float4 qf[4];
    {
        const float* qp = a.q + ((size_t)n * a.Lq + min(row_a, a.Lq - 1)) * a.ldq + h * kADh;
#pragma unroll
        for (int u = 0; u < 4; ++u) qf[u] = row_a < a.Lq ? ldg4(qp + 16 * u + 4 * qd) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int rowi[4];
    bool qv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        rowi[g] = q0 + 16 * w + 4 * qd + g;              // the rows this lane holds in D layout
        qv[g] = rowi[g] < a.Lq && (a.qvalid ? a.qvalid[(size_t)n * a.Lq + rowi[g]] != 0 : true);
    }
    float m[4], l[4];
    f32x4 o[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        m[g] = -INFINITY;
        l[g] = 0.f;
    }
    zero_acc(o);
    const float* kbase = a.k + (size_t)n * a.Lk * a.ldk + h * kADh;
    const float* vbase = a.v + (size_t)n * a.Lk * a.ldv + h * kADh;
    AttnTileRegs kr, vr;
    attn_fetch(kr, kbase, a.ldk, 0, a.Lk);
    attn_fetch(vr, vbase, a.ldv, 0, a.Lk);
    const auto kv_fetch = [&](int k0n) {
        const int j = k0n + (int)threadIdx.x;
        return threadIdx.x < kABk && attn_kvalid(a, n, min(j, a.Lk - 1)) && j < a.Lk ? 1.f : 0.f;
    };
    float kvn = kv_fetch(0);
    for (int k0 = 0; k0 < a.Lk; k0 += kABk) {
        __syncthreads();
        attn_put(Ks, kr);
        attn_put(Vs, vr);
        if (threadIdx.x < kABk) Kvs[threadIdx.x] = kvn;
        __syncthreads();
        if (k0 + kABk < a.Lk) {
            attn_fetch(kr, kbase, a.ldk, k0 + kABk, a.Lk);
            attn_fetch(vr, vbase, a.ldv, k0 + kABk, a.Lk);
            kvn = kv_fetch(k0 + kABk);
        }
        f32x4 s[4];
        zero_acc(s);
        attn_mma_nt(qf, Ks, s);
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = k0 + 16 * t + r;
            const bool kvj = Kvs[16 * t + r] != 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x = s[t][g] * a.rscale;
                if (j < a.Lk && rowi[g] < a.Lq) {
                    if (attn_masked(a, n, rowi[g], j, qv[g], kvj)) x = kMaskFill;
                } else {
                    x = -INFINITY;                        // beyond the tensor: not part of the softmax at all
                }
                s[t][g] = x;
                mx[g] = fmaxf(mx[g], x);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mx[g] = group16_max(mx[g]);
            const float mn = fmaxf(m[g], mx[g]);
            const float alpha = mn == -INFINITY ? 1.f : __expf(m[g] - mn);
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float p = mn == -INFINITY ? 0.f : __expf(s[t][g] - mn);
                s[t][g] = p;
                ps += p;
            }
            ps = group16_sum(ps);
            l[g] = l[g] * alpha + ps;
            m[g] = mn;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t][g] *= alpha;
        }
        attn_store_d(&Ps[w][0], s);                       // per-wave patch: only this wave reads it back
        __builtin_amdgcn_wave_barrier();
        attn_mma_nn(&Ps[w][0], Vs, o);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (rowi[g] >= a.Lq) continue;
        const float inv = 1.0f / l[g];
        float* op = a.o + ((size_t)n * a.Lq + rowi[g]) * a.ldo + h * kADh;
#pragma unroll
        for (int t = 0; t < 4; ++t) op[16 * t + r] = o[t][g] * inv;
        if (a.stats && r == 0) {
            float* sp = a.stats + (((size_t)n * a.H + h) * a.Lq + rowi[g]) * 2;
            sp[0] = m[g];
            sp[1] = inv;
        }
    }
}

// Round 5: the forward on TRANSPOSED score tiles.  S^T = K Q^T puts, in the accumulator layout, query row r of a 16-row block into
// lanes (r, quarter q) with keys 4 q + g of a 16-key block in its four registers -- which is exactly the B operand of the second
// product when that one is written transposed as well, O^T[dh x queries] += V^T[dh x keys] . P^T[keys x queries], with the reduction
// walking the keys of a block in the order (4 q + g: g outer, q across the lanes) instead of (4 g' + q).  P never leaves its registers:
// no per-wave LDS patch, no store / wave barrier / reload between the softmax and the second product (k_attn_fwd_v1 above: 16 stores +
// 4 wide loads per tile and wave, 17 KB of LDS per workgroup).  A query row's statistics are one value per lane (the maximum is
// combined over the four quarter lanes with two cross-lane steps, the sum stays a per-lane partial until the end), the output leaves as
// 16-byte stores.  A wave owns TWO 16-row blocks (a workgroup 128 query rows): every K row / V element read from LDS feeds two MFMAs.
// Key blocks beyond the tensor (the last tile of L = 800 holds 32 keys) and waves without a live row issue no products.
constexpr int kABq2 = 128;      // query rows per workgroup of k_attn_fwd
__device__ __forceinline__ float quarter_max(float v) {      // over the four lanes (r, q = 0..3) of a query row
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quarter_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
// HAS_MASK: the arbitrary byte mask [N, Lq, Lk] (one byte load per score); without it the mask is the outer product of the two validity vectors and
// the whole replacement is ONE fused multiply-add per score: x = s * sel + bias with, per key, (sel, bias) = (1 / sqrt(dh), 0) for a real bin,
// (0, -1e9) for a padded one, (0, -inf) beyond the tensor -- the pair travels with the K / V tile -- and, per query row, sel *= 0 / bias = min(bias, -1e9)
// where the row itself is padding.  (The per-score form of this -- `a.mask ? byte : !(qv && kv)` inside the loops -- compiled to a scalar branch
// sequence per score, 34 of them per tile and wave whether a mask was there or not: a fifth of the kernel's time.)
template <bool HAS_MASK>
__global__ __launch_bounds__(256, HAS_MASK ? 2 : 3) void k_attn_fwd(AttnArgs a) {      // (the byte-mask form keeps 64-bit mask addresses: two workgroups per CU, no scratch)
    __shared__ __attribute__((aligned(16))) float Ks[kABk * kALd];
    __shared__ __attribute__((aligned(16))) float Vs[kABk * kALd];
    __shared__ __attribute__((aligned(16))) float Ksel[kABk];      // per key of the tile: the score's factor ...
    __shared__ __attribute__((aligned(16))) float Kbias[kABk];     // ... and what is added (see above)
    // (an XCD-aware order of the workgroups -- the query tiles of one (sequence, head), which stream the same K / V, on one L2 -- changed nothing:
    //  6.89-6.92 ms either way, gpurun_out/r05h_attn_xcd.txt)
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * kABq2;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, qd = lane >> 4;
    const int qw = q0 + 32 * w;                           // first row of this wave
    const bool live_w = qw < a.Lq;
    float4 qf[2][4];                                      // row qw + 16 b + r, head columns 16 u + 4 qd ..: the B operand of S^T
    int row[2];
    float qsel[2], qbias[2];                              // a padded query row (or one beyond the tensor, never stored): every score is the mask fill
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        row[b] = qw + 16 * b + r;
        const float* qp = a.q + ((size_t)n * a.Lq + min(row[b], a.Lq - 1)) * a.ldq + h * kADh + 4 * qd;
#pragma unroll
        for (int u = 0; u < 4; ++u) qf[b][u] = row[b] < a.Lq ? ldg4(qp + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        const bool qv = row[b] < a.Lq && (!HAS_MASK && a.qvalid ? a.qvalid[(size_t)n * a.Lq + row[b]] != 0 : true);      // (a byte mask takes precedence)
        qsel[b] = qv ? 1.f : 0.f;
        qbias[b] = qv ? 0.f : kMaskFill;
    }
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    f32x4 o[2][4];                                        // O^T: lane (r, qd) holds head columns 16 t' + 4 qd + g of row r
    zero_acc(o[0]);
    zero_acc(o[1]);
    const float* kbase = a.k + (size_t)n * a.Lk * a.ldk + h * kADh;
    const float* vbase = a.v + (size_t)n * a.Lk * a.ldv + h * kADh;
    AttnTileRegs kr, vr;
    attn_fetch_s(kr, kbase, a.ldk, 0, a.Lk);
    attn_fetch_s(vr, vbase, a.ldv, 0, a.Lk);
    const auto kv_fetch = [&](int k0n) {                  // 1: real bin, 0: padded, -1: beyond the tensor
        const int j = k0n + (int)threadIdx.x;
        if (threadIdx.x >= kABk || j >= a.Lk) return -1.f;
        return (HAS_MASK || attn_kvalid(a, n, j)) ? 1.f : 0.f;
    };
    float kvn = kv_fetch(0);
    for (int k0 = 0; k0 < a.Lk; k0 += kABk) {
        __syncthreads();
        attn_put(Ks, kr);
        attn_put(Vs, vr);
        if (threadIdx.x < kABk) {
            Ksel[threadIdx.x] = kvn > 0.f ? a.rscale : 0.f;
            Kbias[threadIdx.x] = kvn > 0.f ? 0.f : (kvn < 0.f ? -INFINITY : kMaskFill);
        }
        __syncthreads();
        const int nkb = min(4, (a.Lk - k0 + 15) >> 4);    // key blocks of this tile that hold a key
        f32x4 s[2][4];
        zero_acc(s[0]);
        zero_acc(s[1]);
        // (the sched barriers keep the LDS reads of ONE key block in front of its products: left alone the scheduler issues all 16 wide reads
        //  of the tile first -- 64 registers -- and the kernel spills)
        if (live_w) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < nkb) {
                    float4 kk[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) kk[u] = *reinterpret_cast<const float4*>(Ks + (16 * t + r) * kALd + 16 * u + 4 * qd);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            s[b][t] = mfma4(kk[u].x, qf[b][u].x, s[b][t]);
                            s[b][t] = mfma4(kk[u].y, qf[b][u].y, s[b][t]);
                            s[b][t] = mfma4(kk[u].z, qf[b][u].z, s[b][t]);
                            s[b][t] = mfma4(kk[u].w, qf[b][u].w, s[b][t]);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the next tile is requested between the softmax and the second product: in front of the score products its 32 registers meet the K rows
        // of a key block, in front of the softmax the per-key factors
        const auto prefetch = [&]() {
            if (k0 + kABk < a.Lk) {
                attn_fetch_s(kr, kbase, a.ldk, k0 + kABk, a.Lk);
                attn_fetch_s(vr, vbase, a.ldv, k0 + kABk, a.Lk);
                kvn = kv_fetch(k0 + kABk);
            }
        };
        if (!live_w) {                                    // (uniform per wave; the barriers above are the loop's only ones)
            prefetch();
            continue;
        }
        float mxs[2] = {-INFINITY, -INFINITY};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 ks4 = *reinterpret_cast<const float4*>(Ksel + 16 * t + 4 * qd);
            const float4 kb4 = *reinterpret_cast<const float4*>(Kbias + 16 * t + 4 * qd);
            const float ksel[4] = {ks4.x, ks4.y, ks4.z, ks4.w}, kbias[4] = {kb4.x, kb4.y, kb4.z, kb4.w};
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float x = fmaf(s[b][t][g], ksel[g] * qsel[b], fminf(kbias[g], qbias[b]));
                    if (HAS_MASK) {                       // (the byte decides; beyond the tensor stays -inf)
                        const int j = k0 + 16 * t + 4 * qd + g;
                        if (j < a.Lk && row[b] < a.Lq && a.mask[((size_t)n * a.Lq + row[b]) * a.Lk + j] != 0) x = kMaskFill;
                    }
                    s[b][t][g] = x;
                    mxs[b] = fmaxf(mxs[b], x);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float mx = quarter_max(mxs[b]);
            const float mn = fmaxf(m[b], mx);             // (finite from the first tile on: it holds a key of the tensor)
            const float alpha = __expf(m[b] - mn);
            float ps = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float p = __expf(s[b][t][g] - mn);
                    s[b][t][g] = p;
                    ps += p;
                }
            l[b] = l[b] * alpha + ps;                     // this lane's keys only; the quarters are summed behind the loop
            m[b] = mn;
#pragma unroll
            for (int t = 0; t < 4; ++t) o[b][t] *= alpha;
            __builtin_amdgcn_sched_barrier(0);
        }
        prefetch();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < nkb) {
#pragma unroll
                for (int gh = 0; gh < 2; ++gh) {
                    float av[2][4];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) av[g][tp] = Vs[(16 * t + 4 * qd + 2 * gh + g) * kALd + r + 16 * tp];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) {
                            o[0][tp] = mfma4(av[g][tp], s[0][t][2 * gh + g], o[0][tp]);
                            o[1][tp] = mfma4(av[g][tp], s[1][t][2 * gh + g], o[1][tp]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (!live_w) return;
    // (the output addresses are formed HERE from a fresh read of the thread index, which the optimiser cannot see through (CF_TID_OPAQUE): formed
    //  from r / qd it computes them in front of the loop and keeps them, or r / qd themselves, in scratch across it)
    const int le = threadIdx.x & 63, re = le & 15, qe = le >> 4, qwe = q0 + 32 * (int)(threadIdx.x >> 6);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float lt = quarter_sum(l[b]);
        const int rowe = qwe + 16 * b + re;
        if (rowe >= a.Lq) continue;
        const float inv = 1.0f / lt;
        float* op = a.o + ((size_t)n * a.Lq + rowe) * a.ldo + h * kADh + 4 * qe;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const float4 v = make_float4(o[b][tp][0] * inv, o[b][tp][1] * inv, o[b][tp][2] * inv, o[b][tp][3] * inv);
            if ((a.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.o) & 15) == 0) {
                stg4(op + 16 * tp, v);
            } else {
                op[16 * tp] = v.x;
                op[16 * tp + 1] = v.y;
                op[16 * tp + 2] = v.z;
                op[16 * tp + 3] = v.w;
            }
        }
        if (a.stats && qe == 0) {
            float* sp = a.stats + (((size_t)n * a.H + h) * a.Lq + rowe) * 2;
            sp[0] = m[b];
            sp[1] = inv;
        }
    }
}

// delta[n, h, i] = sum_d dO[n, i, h, d] * O[n, i, h, d]
__global__ __launch_bounds__(256) void k_attn_delta(AttnArgs a) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15, h = blockIdx.y, n = blockIdx.z;
    if (row >= a.Lq) return;
    const size_t o = ((size_t)n * a.Lq + row) * a.ldo + h * kADh + 4 * sub;
    const float4 x = ldg4(a.o + o), y = ldg4(a.d_o + o);
    float s = (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    s = group16_sum(s);
    if (sub == 0) a.delta[((size_t)n * a.H + h) * a.Lq + row] = s;
}

// Recompute the probability tile of 16 query rows (this wave) x 64 keys in D layout and turn it into dS.
//   p  = exp(s - m) / sum       (0 beyond the tensor; 1/Lk for a fully masked row, like the forward)
//   ds = p * (dp - delta) / sqrt(dh), and 0 where the score was replaced by the mask fill
__device__ __forceinline__ void attn_p_ds(const AttnArgs& a, int n, int k0, const int (&rowi)[4], const bool (&qv)[4],
                                          const float (&mrow)[4], const float (&linv)[4], const float (&dl)[4], f32x4 (&s)[4],
                                          f32x4 (&dp)[4]) {
    const int r = threadIdx.x & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = k0 + 16 * t + r;
        const bool kvj = attn_kvalid(a, n, j);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float p = 0.f, d = 0.f;
            if (j < a.Lk && rowi[g] < a.Lq) {
                const bool mk = attn_masked(a, n, rowi[g], j, qv[g], kvj);
                const float x = mk ? kMaskFill : s[t][g] * a.rscale;
                p = __expf(x - mrow[g]) * linv[g];
                d = mk ? 0.f : p * (dp[t][g] - dl[g]) * a.rscale;
            }
            s[t][g] = p;
            dp[t][g] = d;                 // dS overwrites dP
        }
    }
}

// dK, dV of one key tile: loop over the query tiles.  Wave w owns keys 16w .. 16w+15 of the tile.
//   dV[j] += sum_i p[i][j] dO[i]        dK[j] += sum_i ds[i][j] Q[i]
__global__ __launch_bounds__(256, 3) void k_attn_bwd_kv(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Qs[kABq * kALd];
    __shared__ __attribute__((aligned(16))) float Gs[kABq * kALd];      // dO tile
    __shared__ __attribute__((aligned(16))) float Pt[4][16 * kALd];     // per-wave patches: p^T / ds^T [16 keys][64 rows]
    __shared__ __attribute__((aligned(16))) float Ss[kABq * 4];         // row statistics of the query tile (as in k_attn_bwd below)
    const int n = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * kABk;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, qd = lane >> 4;
    // this wave's 16 keys as the A operand of S^T = K Q^T and dP^T = V dO^T
    const int key_a = k0 + 16 * w + r;
    float4 kf[4], vf[4];
    {
        const float* kp = a.k + ((size_t)n * a.Lk + min(key_a, a.Lk - 1)) * a.ldk + h * kADh;
        const float* vp = a.v + ((size_t)n * a.Lk + min(key_a, a.Lk - 1)) * a.ldv + h * kADh;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kf[u] = key_a < a.Lk ? ldg4(kp + 16 * u + 4 * qd) : make_float4(0.f, 0.f, 0.f, 0.f);
            vf[u] = key_a < a.Lk ? ldg4(vp + 16 * u + 4 * qd) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    int keyi[4];
    bool kv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        keyi[g] = k0 + 16 * w + 4 * qd + g;              // keys this lane holds in D layout (rows of S^T)
        kv[g] = keyi[g] < a.Lk && (a.kvalid ? a.kvalid[(size_t)n * a.Lk + keyi[g]] != 0 : true);
    }
    f32x4 dk[4], dv[4];
    zero_acc(dk);
    zero_acc(dv);
    const float* qbase = a.q + (size_t)n * a.Lq * a.ldq + h * kADh;
    const float* gbase = a.d_o + (size_t)n * a.Lq * a.ldo + h * kADh;
    const float* stp = a.stats + ((size_t)n * a.H + h) * a.Lq * 2;
    const float* dlp = a.delta + ((size_t)n * a.H + h) * a.Lq;
    for (int q0 = 0; q0 < a.Lq; q0 += kABq) {
        __syncthreads();
        {   // three workgroups per CU hide this latency; a register prefetch of the next tile would cost the third one
            AttnTileRegs qr, gr;
            attn_fetch(qr, qbase, a.ldq, q0, a.Lq);
            attn_fetch(gr, gbase, a.ldo, q0, a.Lq);
            float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (threadIdx.x < kABq) {      // max, 1 / sum, delta, valid of row q0 + thread: the softmax below reads them from LDS
                const int ic = min(q0 + (int)threadIdx.x, a.Lq - 1);
                const float2 ml = ldg2(stp + 2 * ic);
                sv = make_float4(ml.x, ml.y, ldg(dlp + ic), (a.qvalid ? a.qvalid[(size_t)n * a.Lq + ic] != 0 : true) ? 1.f : 0.f);
            }
            attn_put(Qs, qr);
            attn_put(Gs, gr);
            if (threadIdx.x < kABq) *reinterpret_cast<float4*>(Ss + threadIdx.x * 4) = sv;
        }
        __syncthreads();
        f32x4 st[4], dpt[4];                               // S^T, dP^T: rows = keys 4qd+g, columns = query 16t + r
        zero_acc(st);
        zero_acc(dpt);
        attn_mma_nt(kf, Qs, st);
        attn_mma_nt(vf, Gs, dpt);
        // p^T overwrites S^T and dS^T overwrites dP^T (each element is read once, by the lane that rewrites it)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = q0 + 16 * t + r;
            const bool iv = i < a.Lq;
            const float4 sv = *reinterpret_cast<const float4*>(Ss + (16 * t + r) * 4);
            const float mrow = sv.x, linv = sv.y, dl = sv.z;
            const bool qvi = iv && sv.w != 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float p = 0.f, d = 0.f;
                if (iv && keyi[g] < a.Lk) {
                    const bool mk = a.mask ? a.mask[((size_t)n * a.Lq + i) * a.Lk + keyi[g]] != 0 : !(qvi && kv[g]);
                    const float x = mk ? kMaskFill : st[t][g] * a.rscale;
                    p = __expf(x - mrow) * linv;
                    d = mk ? 0.f : p * (dpt[t][g] - dl) * a.rscale;
                }
                st[t][g] = p;
                dpt[t][g] = d;
            }
        }
        attn_store_d(&Pt[w][0], st);
        __builtin_amdgcn_wave_barrier();
        attn_mma_nn(&Pt[w][0], Gs, dv);                   // dV[16 keys x 64] += P^T[16 x 64 rows] . dO[64 rows x 64]
        __builtin_amdgcn_wave_barrier();
        attn_store_d(&Pt[w][0], dpt);
        __builtin_amdgcn_wave_barrier();
        attn_mma_nn(&Pt[w][0], Qs, dk);                   // dK += dS^T . Q
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (keyi[g] >= a.Lk) continue;
        float* dkp = a.dk + ((size_t)n * a.Lk + keyi[g]) * a.ldk + h * kADh;
        float* dvp = a.dv + ((size_t)n * a.Lk + keyi[g]) * a.ldv + h * kADh;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            dkp[16 * t + r] = dk[t][g];
            dvp[16 * t + r] = dv[t][g];
        }
    }
}

// dQ of one query tile: loop over the key tiles.   dQ[i] += sum_j ds[i][j] K[j]
__global__ __launch_bounds__(256, 3) void k_attn_bwd_q(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[kABk * kALd];
    __shared__ __attribute__((aligned(16))) float Vs[kABk * kALd];
    __shared__ __attribute__((aligned(16))) float Ds[4][16 * kALd];
    const int n = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * kABq;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, qd = lane >> 4;
    const int row_a = q0 + 16 * w + r;
    float4 qf[4], gf[4];
    {
        const size_t ro = (size_t)n * a.Lq + min(row_a, a.Lq - 1);
        const float* qp = a.q + ro * a.ldq + h * kADh;
        const float* gp = a.d_o + ro * a.ldo + h * kADh;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            qf[u] = row_a < a.Lq ? ldg4(qp + 16 * u + 4 * qd) : make_float4(0.f, 0.f, 0.f, 0.f);
            gf[u] = row_a < a.Lq ? ldg4(gp + 16 * u + 4 * qd) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    int rowi[4];
    bool qv[4];
    float mrow[4], linv[4], dl[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        rowi[g] = q0 + 16 * w + 4 * qd + g;
        const bool iv = rowi[g] < a.Lq;
        const size_t so = ((size_t)n * a.H + h) * a.Lq + (iv ? rowi[g] : 0);
        qv[g] = iv && (a.qvalid ? a.qvalid[(size_t)n * a.Lq + rowi[g]] != 0 : true);
        mrow[g] = iv ? a.stats[2 * so] : 0.f;
        linv[g] = iv ? a.stats[2 * so + 1] : 0.f;
        dl[g] = iv ? a.delta[so] : 0.f;
    }
    f32x4 dq[4];
    zero_acc(dq);
    const float* kbase = a.k + (size_t)n * a.Lk * a.ldk + h * kADh;
    const float* vbase = a.v + (size_t)n * a.Lk * a.ldv + h * kADh;
    for (int k0 = 0; k0 < a.Lk; k0 += kABk) {
        __syncthreads();
        {
            AttnTileRegs kr, vr;
            attn_fetch(kr, kbase, a.ldk, k0, a.Lk);
            attn_fetch(vr, vbase, a.ldv, k0, a.Lk);
            attn_put(Ks, kr);
            attn_put(Vs, vr);
        }
        __syncthreads();
        f32x4 s[4], dp[4];
        zero_acc(s);
        zero_acc(dp);
        attn_mma_nt(qf, Ks, s);
        attn_mma_nt(gf, Vs, dp);                          // dP = dO V^T
        attn_p_ds(a, n, k0, rowi, qv, mrow, linv, dl, s, dp);
        attn_store_d(&Ds[w][0], dp);
        __builtin_amdgcn_wave_barrier();
        attn_mma_nn(&Ds[w][0], Ks, dq);                   // dQ += dS . K
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (rowi[g] >= a.Lq) continue;
        float* dqp = a.dq + ((size_t)n * a.Lq + rowi[g]) * a.ldq + h * kADh;
#pragma unroll
        for (int t = 0; t < 4; ++t) dqp[16 * t + r] = dq[t][g];
    }
}

// dQ, dK, dV in ONE pass (round 4): a workgroup owns one (sequence, head) and walks the key tiles (outer) and the query tiles (inner).
// The split kernels above recompute S and dP in both of them: 7 tile products per (key tile, query tile) pair for 5 of algorithmic
// work.  Here a pair is S^T, dP^T, dV += P^T dO, dK += dS^T Q (wave w = keys 16 w .. 16 w + 15 of the tile, as in k_attn_bwd_kv) and
//   dQ^T[columns 16 w .. + 15 of the head][64 rows of the query tile] += K^T[those columns][64 keys] . dS^T[64 keys][64 rows]
// -- the transposed form keeps everything a wave needs where it is: the A operand (16 columns of the K tile, 16 registers, loaded once
// per key tile) in registers, the B operand = the four waves' dS^T patches in natural layout (one more barrier per pair), and the
// result in the accumulator layout is, per query row, FOUR CONSECUTIVE head columns: the update of dQ in global memory is one 16-byte
// load and one 16-byte store per lane and 16-row block.  dQ is complete only after the last key tile: every pair adds to the rows in
// global memory (the first key tile stores without reading).  One workgroup per (sequence, head) owns those rows and the key tiles
// follow each other in a fixed order: no atomics, bit-reproducible.  THREE workgroups per CU since round 5 (52 KB of LDS each; 166
// registers): in round 4 the kernel needed 256 registers at two per CU and spilled 131 at three -- most of that was thread-index
// arithmetic kept alive across the loops (cf_kernels.h, CF_TID_OPAQUE) and 64-bit per-lane base addresses (now scalar base + 32-bit lane
// offset).  Stress configuration: backward 24.2 (split kernels) -> 21.0 (one pass) -> 19.4 (statistics in LDS) -> 18.2 ms (three per CU):
// 73.5 -> 98 TFLOP/s of algorithmic work.
#ifndef CF_ATTN_NT
#define CF_ATTN_NT 1      // 1: the Q / dO tile fetches (17.6 against 18.1 ms), 2: dQ's old values, 4: dQ's stores (no change: off)
#endif
__global__ __launch_bounds__(256, 3) void k_attn_bwd(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Qs[kABq * kALd];
    __shared__ __attribute__((aligned(16))) float Gs[kABq * kALd];      // dO tile
    __shared__ __attribute__((aligned(16))) float Pt[4][16 * kALd];     // per-wave patches: p^T / ds^T [16 keys][64 rows]
    __shared__ __attribute__((aligned(16))) float Ss[kABq * 4];         // row statistics of the query tile: max, 1 / sum, delta, valid
    const int n = blockIdx.y, h = blockIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, qd = lane >> 4;
    const float* qbase = a.q + (size_t)n * a.Lq * a.ldq + h * kADh;
    const float* gbase = a.d_o + (size_t)n * a.Lq * a.ldo + h * kADh;
    const float* stp = a.stats + ((size_t)n * a.H + h) * a.Lq * 2;
    const float* dlp = a.delta + ((size_t)n * a.H + h) * a.Lq;
    float* dqb = a.dq + (size_t)n * a.Lq * a.ldq + h * kADh + 16 * w + 4 * qd;      // this lane's four columns
    // the statistics of a query tile travel with it (thread i < 64: row i): the softmax of a pair reads them from LDS instead of waiting for L2
    auto stat_fetch = [&](int q0n) {
        const int i = q0n + (int)threadIdx.x, ic = min(i, a.Lq - 1);
        float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (threadIdx.x < kABq) {
            const float2 ml = ldg2(stp + 2 * ic);
            sv = make_float4(ml.x, ml.y, ldg(dlp + ic), (a.qvalid ? a.qvalid[(size_t)n * a.Lq + ic] != 0 : true) ? 1.f : 0.f);
        }
        return sv;
    };
    for (int k0 = 0; k0 < a.Lk; k0 += kABk) {
        // this wave's 16 keys as the A operand of S^T = K Q^T and dP^T = V dO^T
        const int key_a = k0 + 16 * w + r;
        float4 kf[4], vf[4];
        {   // (scalar base of the (sequence, head) + 32-bit lane offset: as 64-bit per-lane addresses these bases were the kernel's only spills)
            const float* kp = lane_at(sbase(a.k + (size_t)n * a.Lk * a.ldk + h * kADh, 0), (unsigned)(min(key_a, a.Lk - 1) * a.ldk + 4 * qd) * 4u);
            const float* vp = lane_at(sbase(a.v + (size_t)n * a.Lk * a.ldv + h * kADh, 0), (unsigned)(min(key_a, a.Lk - 1) * a.ldv + 4 * qd) * 4u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                kf[u] = key_a < a.Lk ? ldg4(kp + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
                vf[u] = key_a < a.Lk ? ldg4(vp + 16 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // column 16 w + r of the K tile as the A operand of dQ^T = K^T dS^T: kt[u] = K[keys k0 + 16 u + 4 qd .. + 3][that column]
        float4 kt[4];
        {
            const float* kc = a.k + (size_t)n * a.Lk * a.ldk + h * kADh + 16 * w + r;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = k0 + 16 * u + 4 * qd + i;
                    v4[i] = key < a.Lk ? ldg(kc + (size_t)key * a.ldk) : 0.f;
                }
                kt[u] = make_float4(v4[0], v4[1], v4[2], v4[3]);
            }
        }
        int keyi[4];
        bool kv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            keyi[g] = k0 + 16 * w + 4 * qd + g;              // keys this lane holds in D layout (rows of S^T)
            kv[g] = keyi[g] < a.Lk && (a.kvalid ? a.kvalid[(size_t)n * a.Lk + keyi[g]] != 0 : true);
        }
        f32x4 dk[4], dv[4];
        zero_acc(dk);
        zero_acc(dv);
        {   // first query tile of this key tile (the barrier that ended the previous key tile's last pair has freed Qs / Gs)
            AttnTileRegs qr, gr;
            attn_fetch_t<CF_ATTN_NT & 1>(qr, qbase, a.ldq, 0, a.Lq);
            attn_fetch_t<CF_ATTN_NT & 1>(gr, gbase, a.ldo, 0, a.Lq);
            const float4 sv = stat_fetch(0);
            attn_put(Qs, qr);
            attn_put(Gs, gr);
            if (threadIdx.x < kABq) *reinterpret_cast<float4*>(Ss + threadIdx.x * 4) = sv;
        }
        __syncthreads();
        for (int q0 = 0; q0 < a.Lq; q0 += kABq) {
            f32x4 st[4], dpt[4];                               // S^T, dP^T: rows = keys 4qd+g, columns = query 16t + r
            zero_acc(st);
            zero_acc(dpt);
            attn_mma_nt(kf, Qs, st);
            attn_mma_nt(vf, Gs, dpt);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = q0 + 16 * t + r;
                const bool iv = i < a.Lq;
                const float4 sv = *reinterpret_cast<const float4*>(Ss + (16 * t + r) * 4);
                const float mrow = sv.x, linv = sv.y, dl = sv.z;
                const bool qvi = iv && sv.w != 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float p = 0.f, d = 0.f;
                    if (iv && keyi[g] < a.Lk) {
                        const bool mk = a.mask ? a.mask[((size_t)n * a.Lq + i) * a.Lk + keyi[g]] != 0 : !(qvi && kv[g]);
                        const float x = mk ? kMaskFill : st[t][g] * a.rscale;
                        p = __expf(x - mrow) * linv;
                        d = mk ? 0.f : p * (dpt[t][g] - dl) * a.rscale;
                    }
                    st[t][g] = p;
                    dpt[t][g] = d;
                }
            }
            attn_store_d(&Pt[w][0], st);
            __builtin_amdgcn_wave_barrier();
            attn_mma_nn(&Pt[w][0], Gs, dv);                   // dV[16 keys x 64] += P^T[16 x 64 rows] . dO[64 rows x 64]
            __builtin_amdgcn_wave_barrier();
            attn_store_d(&Pt[w][0], dpt);
            // the rows of dQ this lane adds to (query row q0 + 16 t + r, its four columns): requested here, behind the last use of S^T / dP^T
            float4 dqo[4];
            if (k0 > 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) dqo[t] = (CF_ATTN_NT & 2) ? ldg4_nt(dqb + (size_t)min(q0 + 16 * t + r, a.Lq - 1) * a.ldq) : ldg4(dqb + (size_t)min(q0 + 16 * t + r, a.Lq - 1) * a.ldq);
            }
            __builtin_amdgcn_wave_barrier();
            attn_mma_nn(&Pt[w][0], Qs, dk);                   // dK += dS^T . Q
            __syncthreads();                                  // all four dS^T patches are written; nobody reads Qs / Gs any more
            AttnTileRegs qr, gr;
            const bool more = q0 + kABq < a.Lq;
            float4 svn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (more) svn = stat_fetch(q0 + kABq);
            f32x4 dq[4];                                      // dQ^T: rows = head columns 16 w + 4 qd + g, columns = query row 16 t + r
            zero_acc(dq);
#pragma unroll
            for (int u = 0; u < 4; ++u) {                     // keys 16 u .. 16 u + 15: wave u's patch
                const float a4[4] = {kt[u].x, kt[u].y, kt[u].z, kt[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) dq[t] = mfma4(a4[i], Pt[u][(4 * qd + i) * kALd + 16 * t + r], dq[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = q0 + 16 * t + r;
                if (row < a.Lq) {
                    float4 v = make_float4(dq[t][0], dq[t][1], dq[t][2], dq[t][3]);
                    if (k0 > 0) v = make_float4(v.x + dqo[t].x, v.y + dqo[t].y, v.z + dqo[t].z, v.w + dqo[t].w);
                    if (CF_ATTN_NT & 4) stg4_nt(dqb + (size_t)row * a.ldq, v); else stg4(dqb + (size_t)row * a.ldq, v);
                }
            }
            if (more) {      // (a register prefetch of the next tile under the dQ^T product lost in rounds 4 and 5: 19.1 against 18.2 ms)
                attn_fetch_t<CF_ATTN_NT & 1>(qr, qbase, a.ldq, q0 + kABq, a.Lq);
                attn_fetch_t<CF_ATTN_NT & 1>(gr, gbase, a.ldo, q0 + kABq, a.Lq);
                attn_put(Qs, qr);
                attn_put(Gs, gr);
                if (threadIdx.x < kABq) *reinterpret_cast<float4*>(Ss + threadIdx.x * 4) = svn;
            }
            __syncthreads();                                  // the next tile is in LDS; every wave is done with the patches
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (keyi[g] >= a.Lk) continue;
            float* dkp = a.dk + ((size_t)n * a.Lk + keyi[g]) * a.ldk + h * kADh;
            float* dvp = a.dv + ((size_t)n * a.Lk + keyi[g]) * a.ldv + h * kADh;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                dkp[16 * t + r] = dk[t][g];
                dvp[16 * t + r] = dv[t][g];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 6: the one-pass backward with 128 KEYS PER PASS (k_attn_bwd2).  k_attn_bwd above walks Q / dO / dQ of its (sequence, head) once per
// 64-key tile: 64 KB of HBM traffic per (key tile, query tile) pair, 47 GB per launch at the stress shape, and its timing probes say that this
// traffic -- not the matrix pipe, not latency -- is what it waits for (LAB_NOTES.md, round 5).  Here a pass covers 128 keys: half the passes,
// half the bytes.  Eight waves (one workgroup per CU, two waves per SIMD); wave w owns keys 16 w .. 16 w + 15 of the pass.
//
//  * NON-transposed score tiles: S = Q K^T with the Q tile as the A operand (LDS) and the wave's K rows as the B operand (registers, loaded once
//    per pass).  In the accumulator layout a lane then holds S[query 16 t + 4 qd + g][key r] -- which IS the B operand of the transposed
//    gradient products  dV^T[dh x keys] += dO^T[dh x queries] P[queries x keys],  dK^T += Q^T dS  (reduction over the queries in the order
//    4 qd + j: j outer, qd across the lanes), so P and dS feed them straight from the registers: no P^T / dS^T patch, no wave barrier between
//    the softmax and the products; the A operands are scalar reads of the dO / Q tiles in natural layout (conflict-free: lanes r walk a row).
//  * dS goes to LDS once, as one 16-byte store per lane and 16-row block ([key][query] patches), for the only product that reduces over
//    the keys:  dQ^T[16 head columns x 32 rows] += K^T[16 x 128 keys] dS^T[128 x 32]  per wave (wave = (column block w & 3, row half w >> 2)),
//    added to the rows of dQ in global memory as before (one owner per element, fixed order of the passes: no atomics, bit-reproducible).
//  * ONE workgroup barrier per pair: the Q / dO / statistics tiles and the dS patches are double-buffered in LDS (141 KB), the next pair's
//    tile is requested into registers two phases ahead (behind the barrier of the previous pair) and written into the other buffer at the
//    end of this pair's first phase -- its HBM round trip lies under ~10 K cycles of products instead of under the 2 K of the dQ^T product.
//  * 16-row blocks of the last query tile beyond the tensor issue no products (L = 800: half of tile 13), waves whose keys lie beyond it
//    skip their first phase (they publish a zero patch).
#ifndef CF_AB2_SCHED
#define CF_AB2_SCHED 7      // scheduling fences inside the first phase (1: between the k-chunks of S / dP, 2: behind the softmax, 4: between the row blocks of dV / dK)
#endif
#ifndef CF_AB2_M0_KEEP
#define CF_AB2_M0_KEEP 0
#endif
#ifndef CF_AB2_STAGGER
#define CF_AB2_STAGGER 1    // waves 4-7 run the two phases of an interval in the opposite order (see the schedule inside the kernel)
#endif
#ifndef CF_AB2_PROBE
#define CF_AB2_PROBE 0      // timing probes (WRONG results): 1 no p / dS arithmetic, 2 no tile transfers after the first, 4 no dQ update in global memory,
#endif                      // 8 no barriers, 16 no dV / dK products, 32 no dQ^T product, 64 no S / dP products
constexpr int kAB2K = 128;                                   // keys per pass
constexpr int kAB2Tile = kABq * kALd;                        // floats of a [64][68] tile
constexpr int kAB2Patch = 16 * kALd;                         // floats of one wave's dS patch [16 keys][68]
constexpr int kAB2LdsFloats = 4 * kAB2Tile + 16 * kAB2Patch + 2 * kABq * 4;
constexpr size_t kAB2Smem = (size_t)kAB2LdsFloats * sizeof(float);

template <bool HAS_MASK>
__global__ __launch_bounds__(512, 1) void k_attn_bwd2(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Qs = smem;                                  // [2][64][68]
    float* const Gs = smem + 2 * kAB2Tile;                   // [2][64][68]  dO
    float* const Ds = smem + 4 * kAB2Tile;                   // [2][8][16][68]  dS patches: [key of the wave][query of the tile]
    float* const Ss = Ds + 16 * kAB2Patch;                   // [2][64][4]  max, 1 / sum (0 beyond the tensor), delta, query valid
    const int n = blockIdx.y, h = blockIdx.x;
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 15, qd = lane >> 4;
    const int cw = w & 3, t0 = 2 * (w >> 2);                 // dQ^T: head columns 16 cw .. + 15, row blocks t0, t0 + 1 of the query tile
    const float* qseq = a.q + (size_t)n * a.Lq * a.ldq + h * kADh;
    const float* gseq = a.d_o + (size_t)n * a.Lq * a.ldo + h * kADh;
    const float* kseq = a.k + (size_t)n * a.Lk * a.ldk + h * kADh;
    const float* vseq = a.v + (size_t)n * a.Lk * a.ldv + h * kADh;
    const float* stp = a.stats + ((size_t)n * a.H + h) * a.Lq * 2;
    const float* dlp = a.delta + ((size_t)n * a.H + h) * a.Lq;
    float* dqseq = a.dq + (size_t)n * a.Lq * a.ldq + h * kADh + 16 * cw + 4 * qd;      // this lane's four columns
    const int NQ = (a.Lq + kABq - 1) / kABq;
    const int sj = tid >> 4, sd = (tid & 15) * 4;            // staging: rows sj, sj + 32; columns sd .. sd + 3

    // A query tile travels global -> LDS by dword LDS-DMA, one wave instruction per tile row (256 contiguous bytes into a padded LDS row: M0 carries the
    // row's LDS address), issued from inline assembly: no staging registers, and the compiler -- which does not see the transfer -- puts no
    // conservative wait in front of the next LDS access; the wait is this kernel's own `s_waitcnt vmcnt(0)` in front of the barrier that publishes
    // the tile.  Wave w moves rows 8 w .. 8 w + 7 of Q and of dO.  Rows beyond the tensor re-read the last one: their 1 / sum is 0 (below), so
    // whatever they hold contributes exact zeros.  The statistics of the tile (64 rows x 4 floats) go through one register set of threads 0 .. 63.
    typedef __attribute__((address_space(3))) float* lds_ptr_t;
    const unsigned lds_q = (unsigned)(uintptr_t)(lds_ptr_t)Qs, lds_g = (unsigned)(uintptr_t)(lds_ptr_t)Gs;
    // (M0 is written in the statement that uses it and not put back: the compiler keeps nothing in M0 across statements -- no other LDS-DMA, no
    //  `s_movrel`, no `ds_*_addtid` in this kernel; CF_AB2_M0_KEEP=1: the save / restore form of tools/probes/lds_dma_row.hip)
    auto dma_row = [&](const float* sb, unsigned voff, unsigned lds_byte) {
#if CF_AB2_M0_KEEP
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sb), "s"(lds_byte) : "memory");
#else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sb), "s"(lds_byte) : "memory");
#endif
    };
    auto dma_tile = [&](int q0, int b) {
        const float* qb = sbase(qseq, 0);
        const float* gb = sbase(gseq, 0);
        const unsigned lq = lds_q + (unsigned)((b * kAB2Tile + 8 * w * kALd) * 4), lg = lds_g + (unsigned)((b * kAB2Tile + 8 * w * kALd) * 4);
        if (q0 + kABq <= a.Lq) {      // a whole tile: one lane offset, the row in the (scalar) base
            const unsigned vq = (unsigned)((q0 + 8 * w) * a.ldq + lane) * 4u, vg = (unsigned)((q0 + 8 * w) * a.ldo + lane) * 4u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                dma_row(qb + (size_t)i * a.ldq, vq, lq + (unsigned)(i * kALd * 4));
                dma_row(gb + (size_t)i * a.ldo, vg, lg + (unsigned)(i * kALd * 4));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = min(q0 + 8 * w + i, a.Lq - 1);
                dma_row(qb, (unsigned)(row * a.ldq + lane) * 4u, lq + (unsigned)(i * kALd * 4));
                dma_row(gb, (unsigned)(row * a.ldo + lane) * 4u, lg + (unsigned)(i * kALd * 4));
            }
        }
    };
    float4 ps;
    auto fetch_stats = [&](int q0) {
        ps = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < kABq) {
            const int i = q0 + tid, ic = min(i, a.Lq - 1);
            const float2 ml = ldg2(stp + 2 * ic);
            const bool qv = a.qvalid ? a.qvalid[(size_t)n * a.Lq + ic] != 0 : true;
            // (beyond the tensor: "masked" with maximum 0 and 1 / sum 0, so that p = exp(-1e9 - 0) * 0 = 0 whatever the clamped row holds)
            ps = i < a.Lq ? make_float4(ml.x, ml.y, ldg(dlp + ic), qv ? 1.f : 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto put_stats = [&](int b) {
        if (tid < kABq) *reinterpret_cast<float4*>(Ss + b * kABq * 4 + tid * 4) = ps;
    };
    auto publish = [&]() {      // this wave's transfers have landed; the barrier makes every wave's visible
        if (CF_AB2_PROBE & 8) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // ---- the schedule.  Interval p (between the barriers that end pairs p - 1 and p) holds two independent pieces of work: the SECOND phase of
    // pair p - 1 (dQ^T from the dS patches that barrier published) and the FIRST phase of pair p (S, dP, p / dS, dV^T, dK^T on the tile that barrier
    // published).  Waves 0-3 run them in that order, waves 4-7 in the opposite one -- a SIMD holds one wave of each group, so while one of them
    // does the softmax arithmetic, updates dQ in global memory or issues the next tile's transfers, the other one has matrix-core work
    // (CF_AB2_STAGGER=0: every wave in the first order).  The first interval of a pass runs second-phase-first in every wave: that phase still
    // needs the previous pass's K^T operand, the first phase the new pass's.
    const int NK = (a.Lk + kAB2K - 1) / kAB2K, P = NQ * NK;
    const bool grp_b = CF_AB2_STAGGER && w >= 4;
    float4 kf[4], vf[4], kt[8];
    f32x4 dk[4], dv[4];                                      // dK^T, dV^T: rows = head columns 16 c + 4 qd + g, column = key r of the wave
    int key = 0;
    bool wave_live = false, klive = false, kvr = false;

    auto load_pass = [&](int k0) {      // the wave's 16 keys as rows of K and V (B operands of S and dP), column 16 cw + r of the K tile (A operand of dQ^T)
        key = k0 + 16 * w + r;                               // the key of this lane's score column
        wave_live = k0 + 16 * w < a.Lk;
        klive = key < a.Lk;
        kvr = klive && (a.kvalid ? a.kvalid[(size_t)n * a.Lk + min(key, a.Lk - 1)] != 0 : true);
        const unsigned ko = (unsigned)(min(key, a.Lk - 1) * a.ldk + 4 * qd) * 4u, vo = (unsigned)(min(key, a.Lk - 1) * a.ldv + 4 * qd) * 4u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kf[u] = f4_keep_if(klive, ldg4(lane_at(sbase(kseq, 0), ko) + 16 * u));
            vf[u] = f4_keep_if(klive, ldg4(lane_at(sbase(vseq, 0), vo) + 16 * u));
        }
        // (scalar base + 32-bit lane offsets, no branch around a load: as 64-bit per-lane addresses these 32 gathers were the kernel's spills)
        const float* kc = sbase(kseq + 16 * cw, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float v4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = k0 + 16 * u + 4 * qd + i;
                const float x = ldg(lane_at(kc, (unsigned)(min(kk, a.Lk - 1) * a.ldk + r) * 4u));
                v4[i] = kk < a.Lk ? x : 0.f;
            }
            kt[u] = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
        zero_acc(dk);
        zero_acc(dv);
    };
    auto store_pass = [&]() {
        if (klive) {
            float* dkp = a.dk + ((size_t)n * a.Lk + key) * a.ldk + h * kADh + 4 * qd;
            float* dvp = a.dv + ((size_t)n * a.Lk + key) * a.ldv + h * kADh + 4 * qd;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                stg4(dkp + 16 * c, make_float4(dk[c][0], dk[c][1], dk[c][2], dk[c][3]));
                stg4(dvp + 16 * c, make_float4(dv[c][0], dv[c][1], dv[c][2], dv[c][3]));
            }
        }
    };

    // ---------------- first phase of pair p: S, dP, p / dS, dV^T, dK^T of this wave's 16 keys, its dS patch
    auto phase_a = [&](int p, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int b = p & 1, q0 = (p % NQ) * kABq;
        const int nt = min(4, (a.Lq - q0 + 15) >> 4);        // live 16-row blocks of this query tile
        const float* Qb = Qs + b * kAB2Tile;
        const float* Gb = Gs + b * kAB2Tile;
        const float* Sb = Ss + b * kABq * 4;
        float* Dw = Ds + (b * 8 + w) * kAB2Patch;
        if (wave_live) {
            f32x4 s[4], dp[4];
            zero_acc(s);
            zero_acc(dp);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (CF_AB2_PROBE & 64) break;
                if (CF_AB2_SCHED & 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!FULL && t >= nt) continue;
                    const float4 qa = *reinterpret_cast<const float4*>(Qb + (16 * t + r) * kALd + 16 * u + 4 * qd);
                    const float4 ga = *reinterpret_cast<const float4*>(Gb + (16 * t + r) * kALd + 16 * u + 4 * qd);
                    s[t] = mfma4(qa.x, kf[u].x, s[t]);
                    dp[t] = mfma4(ga.x, vf[u].x, dp[t]);
                    s[t] = mfma4(qa.y, kf[u].y, s[t]);
                    dp[t] = mfma4(ga.y, vf[u].y, dp[t]);
                    s[t] = mfma4(qa.z, kf[u].z, s[t]);
                    dp[t] = mfma4(ga.z, vf[u].z, dp[t]);
                    s[t] = mfma4(qa.w, kf[u].w, s[t]);
                    dp[t] = mfma4(ga.w, vf[u].w, dp[t]);
                }
            }
            if (CF_AB2_SCHED & 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!FULL && t >= nt) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (CF_AB2_PROBE & 1) break;
                    const float4 sv = *reinterpret_cast<const float4*>(Sb + (16 * t + 4 * qd + g) * 4);
                    bool mk;
                    if (HAS_MASK) mk = a.mask[((size_t)n * a.Lq + min(q0 + 16 * t + 4 * qd + g, a.Lq - 1)) * a.Lk + min(key, a.Lk - 1)] != 0;
                    else mk = !(sv.w != 0.f && kvr);
                    const float x = mk ? kMaskFill : s[t][g] * a.rscale;
                    float pr = __expf(x - sv.x) * sv.y;
                    pr = klive ? pr : 0.f;
                    s[t][g] = pr;
                    dp[t][g] = mk ? 0.f : pr * (dp[t][g] - sv.z) * a.rscale;      // dS overwrites dP
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!FULL && t >= nt) continue;
                if (CF_AB2_SCHED & 4) __builtin_amdgcn_sched_barrier(0);
                *reinterpret_cast<float4*>(Dw + r * kALd + 16 * t + 4 * qd) = make_float4(dp[t][0], dp[t][1], dp[t][2], dp[t][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (CF_AB2_PROBE & 16) break;
                    const float* grow = Gb + (16 * t + 4 * qd + j) * kALd + r;
                    const float* qrow = Qb + (16 * t + 4 * qd + j) * kALd + r;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        dv[c] = mfma4(grow[16 * c], s[t][j], dv[c]);
                        dk[c] = mfma4(qrow[16 * c], dp[t][j], dk[c]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(Dw + r * kALd + 16 * t + 4 * qd) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // ---------------- second phase of pair p: dQ^T of this wave's (column block, row half), added to dQ in global memory
    auto phase_b = [&](int p) {
        const int b = p & 1, q0 = (p % NQ) * kABq;
        const bool first_pass = p < NQ;
        const int nt = min(4, (a.Lq - q0 + 15) >> 4);
        const bool r0 = t0 < nt, r1 = t0 + 1 < nt;
        float4 dqo[2];
        if (!first_pass && !(CF_AB2_PROBE & 4)) {
            dqo[0] = ldg4(dqseq + (size_t)min(q0 + 16 * t0 + r, a.Lq - 1) * a.ldq);
            dqo[1] = ldg4(dqseq + (size_t)min(q0 + 16 * t0 + 16 + r, a.Lq - 1) * a.ldq);
        }
        f32x4 dq[2];
        zero_acc(dq);
        const float* Db = Ds + b * 8 * kAB2Patch + 4 * qd * kALd + 16 * t0 + r;
        if (r1) {          // both row blocks live (every tile but a ragged last one)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (CF_AB2_PROBE & 32) break;
                const float a4[4] = {kt[u].x, kt[u].y, kt[u].z, kt[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dq[0] = mfma4(a4[i], Db[u * kAB2Patch + i * kALd], dq[0]);
                    dq[1] = mfma4(a4[i], Db[u * kAB2Patch + i * kALd + 16], dq[1]);
                }
            }
        } else if (r0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float a4[4] = {kt[u].x, kt[u].y, kt[u].z, kt[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) dq[0] = mfma4(a4[i], Db[u * kAB2Patch + i * kALd], dq[0]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int row = q0 + 16 * (t0 + tt) + r;
            if (row < a.Lq && (!(CF_AB2_PROBE & 4) || p >= P - NQ)) {
                float4 v = make_float4(dq[tt][0], dq[tt][1], dq[tt][2], dq[tt][3]);
                if (!first_pass && !(CF_AB2_PROBE & 4)) v = make_float4(v.x + dqo[tt].x, v.y + dqo[tt].y, v.z + dqo[tt].z, v.w + dqo[tt].w);
                stg4(dqseq + (size_t)row * a.ldq, v);
            }
        }
    };
    // pair p + 1's tile into the buffers pair p - 1 used (every wave is past the barrier that ended that pair's first phase), its statistics into `ps`
    auto send_next = [&](int p) {
        if (p + 1 < P && !((CF_AB2_PROBE & 2) && p > 0)) {
            const int q0n = ((p + 1) % NQ) * kABq;
            fetch_stats(q0n);
            dma_tile(q0n, (p + 1) & 1);
        }
    };

    fetch_stats(0);
    dma_tile(0, 0);
    put_stats(0);
    publish();
    for (int p = 0; p <= P; ++p) {
        const bool pass_start = p < P && p % NQ == 0;
        const bool b_first = !grp_b || pass_start || p == P;
        if (b_first && p > 0) phase_b(p - 1);                // (behind it: a compiler-counted wait for its dQ loads must not cover the transfers below)
        if (p % NQ == 0 && p > 0) store_pass();              // the previous pass's dK, dV are complete
        if (p == P) break;
        if (pass_start) load_pass((p / NQ) * kAB2K);
        send_next(p);
        if (a.Lq - (p % NQ) * kABq >= kABq) phase_a(p, std::true_type{});
        else phase_a(p, std::false_type{});
        if (!b_first) phase_b(p - 1);
        if (p + 1 < P) put_stats((p + 1) & 1);
        publish();                                           // every dS patch of pair p and pair p + 1's tile are in LDS
    }
}

}  // namespace cf

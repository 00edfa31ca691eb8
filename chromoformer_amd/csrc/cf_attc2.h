// cf_attc2.h -- centre-row attention, eight regions per workgroup, on the matrix cores
// (included by cf_kernels.h).
//
// Same mathematics as k_attc (one region per workgroup, VALU): for every (region n, head h)
//     forward : score_j = (f_j . u + PE_j . qt) / sqrt(dh),  u = Wlp^T qt;  p = softmax(masked score)
//               xbar = Wlp (sum_j p_j f_j) + sum_j p_j PE_j
//     backward: dp_j = f_j . dw + PE_j . dxbar,  dw = Wlp^T dxbar;  ds = p (dp - <p,dp>) / sqrt(dh) (0 where masked)
//               dqt = Wlp (sum_j ds_j f_j) + sum_j ds_j PE_j
// but the 8 regions x 2 heads of a workgroup form one 16-row MFMA tile, so the two passes over the
// positional table become [16 x 128] . [128 x L] and [16 x L] . [L x 128] products and each table
// element fetched from L2 is used for 16 rows instead of 2.  The 7-mark features of the 8 regions
// (the only HBM stream) are staged once in LDS and serve both feature passes.
//
// Phases of a workgroup (barriers b1 .. b5; per-wave shader-clock stamps: tools/attc_stamps.py):
//   stage   small loads (operand rows, Wlp, masks) -> LDS; the feature strips and PE^T chunk 0 are requested behind them   | b1
//   (2a)    t = vin . PE^T on the matrix cores (wave = 64-column block), u = vin Wlp by the last wave; features -> LDS;
//           pass 5's PE operand is requested where the early waves wait for the late ones                                  | b2
//   (2b)    epilogue in the accumulator layout: + f.u, scale, mask; per (row, block) softmax statistics (fwd) / <p,dp> (bwd)| b3
//   (3)     every lane combines the <= 16 block statistics of its rows: p / ds -> score tile (+ global p)                  | b4
//   (4),(5) w = sum_j sc_j f_j (VALU), out = sc . PE on the matrix cores (2 column groups x 4 K quarters)                  | b5
//   tail    K-split partials through LDS, each wave finishes a quarter of the rows: + Wlp w, store
// Measured at L = 400, 8 regions (cycles of the 2 GHz shader clock): the two matrix passes are bound by the MFMA pipe (2 x ~9 K
// with two waves per SIMD, no-load experiment: 9 K), the table loads add ~2.5 K, the first barrier costs the 3-4 K of one
// HBM round trip; 43 K -> 35 K (backward), 51 K -> 47 K (forward) against the version that staged everything up front.
#pragma once
#include <type_traits>

namespace cf {

struct Attc2Args {
    const float* feats[kMaxRes];
    const uint8_t* mask[kMaxRes];
    long long mstride[kMaxRes];
    const float* pe[kMaxRes];    // [Lpad][128], rows >= L zero
    const float* pet[kMaxRes];   // [128][LT],   columns >= L zero
    const float* wlp[kMaxRes];
    const float* vin[kMaxRes];   // fwd: qt, bwd: dxbar   [N,2,128]
    float* p[kMaxRes];           // [N,2,L]   fwd: out, bwd: in
    float* w[kMaxRes];           // [N,2,8]   fwd: w (out), bwd: du (out)
    float* vout[kMaxRes];        // fwd: xbar, bwd: dqt   [N,2,128]
    int L[kMaxRes], Lpad[kMaxRes], LT[kMaxRes];
    int N, F;
    float scale, rscale;         // sqrt(dh) and its reciprocal
    unsigned long long* tdbg;    // optional shader-clock stamps (workgroup 0 of the last resolution)
    unsigned long long* tall;    // optional: (start, end) of EVERY workgroup (wave 0), [gridDim.y][gridDim.x][2]
};
#define CF_STAMP2(slot)                                                                                   \
    do {                                                                                                  \
        if (a.tdbg && stamp_wg == 0 && (threadIdx.x & 63) == 0)                                            \
            a.tdbg[(threadIdx.x >> 6) * 32 + (BWD ? 16 : 0) + (slot)] = __builtin_amdgcn_s_memtime();      \
    } while (0)
constexpr int kAGMax = 8;                                // regions per workgroup: 8, 4, 2 or 1 (template parameter AG)
constexpr int kAT = 512;                                 // threads per workgroup (8 waves)
__host__ __device__ inline int attc2_lpad(int L) { return (L + 255) / 256 * 256; }
__host__ __device__ inline int attc2_lt(int L) { return (L + 63) / 64 * 64 + 64; }
__host__ __device__ inline size_t attc2_smem(int L, int F, int AG) {
    return (size_t)(kTile * (attc2_lpad(L) + 4) + 2 * kTile * (kD + 4) + 2 * kTile * 8 + kD * 8 + AG * L * F + 8) * sizeof(float) + (size_t)AG * attc2_lpad(L);
}

// C[16, 64] += A[16, 32*nchunks] . Bm (row-major [k][ldb]) with the 4-slot operand ring, any nchunks >= 1.
// Split in two so that the first ring slots can be requested long before the A operand exists.
__device__ __forceinline__ void frag_load_nn_rt(FragNN<4, 8>& f, const float* __restrict__ Bm, int ldb, int nchunks) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    f.bp = Bm + (size_t)(q * 4) * ldb + 4 * r;
    f.ldb = ldb;
#pragma unroll
    for (int c = 0; c < kRing - 1; ++c) frag_chunk_nn(f, c, min(c, nchunks - 1));     // unconditional: no undefined ring slots
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void frag_mma_nn_rt(FragNN<4, 8>& f, const float* As, int lda, int nchunks, f32x4 (&acc)[4]) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const float* ap = As + r * lda + q * 4;
    float4 a_nxt = *reinterpret_cast<const float4*>(ap);      // A operand one k-step ahead (rows are padded: reading one past is in bounds)
    for (int c0 = 0; c0 < nchunks; c0 += kRing) {
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            const int c = c0 + u;
            if (c + kRing - 1 < nchunks) frag_chunk_nn(f, (u + kRing - 1) % kRing, c + kRing - 1);
            __builtin_amdgcn_sched_barrier(0);
            if (c < nchunks)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float4 a = a_nxt;
                    a_nxt = *reinterpret_cast<const float4*>(ap + (c * 2 + k + 1) * 16);
                    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float* bv = reinterpret_cast<const float*>(&f.ring[u][k][i]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t] = mfma4(av[i], bv[t], acc[t]);
                    }
                }
        }
    }
}

// LDS of the body: a SCRATCH part [sc | vin | red | u | w] that is dead when the body returns, and a PERSISTENT part
// [Wlp | features | masks] that depends only on the regions (not on the layer or the direction): a caller that runs several
// layers over the same regions (cf_trunk.h) stages it once and passes STAGED = true afterwards.
__host__ __device__ inline size_t attc2_scratch_floats(int L) { return (size_t)kTile * (attc2_lpad(L) + 4) + 2 * kTile * (kD + 4) + 2 * kTile * 8; }
__host__ __device__ inline size_t attc2_persist_bytes(int L, int F, int AG) {
    return (size_t)(kD * 8 + AG * L * F + 8) * sizeof(float) + (size_t)AG * attc2_lpad(L);
}
// The persistent part filled AHEAD of the first attention over a group of regions (cf_trunk.h: at the start of the launch, under the
// Embedding layer / the last layer's chain): Wlp and the masks through registers, the feature strips -- the long HBM stream -- by
// 16-byte loads straight into LDS (global_load_lds_dwordx4: no registers in flight, nothing to wait for here).  The caller waits
// (`s_waitcnt vmcnt(0)`, then a workgroup barrier) before the first attention, which then runs with STAGED = true.
template <int AG>
__device__ __forceinline__ void attc2_stage(const float* feats, const uint8_t* mask, long long mstride, const float* wlp, int L, int Lpad, int F,
                                            const int n0, const int N, float* persist) {
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* wlp_s = persist;
    float* feats_s = wlp_s + kD * 8;
    uint8_t* mk_s = reinterpret_cast<uint8_t*>(feats_s + AG * L * F + 8);
    const int nreg = min(AG, N - n0);
    const float* fg = feats + (size_t)n0 * L * F;
    const int nf = nreg * L * F;
    constexpr int NFL = AG == 8 ? 12 : AG == 4 ? 6 : AG == 2 ? 3 : 2;
    const bool f16b = ((L * F) & 3) == 0 && AG * L * F <= NFL * 4 * kAT && (reinterpret_cast<uintptr_t>(fg) & 15) == 0;
    if (f16b) {
        const int n4 = (AG * L * F) >> 2, last4 = (nf >> 2) - 1;      // strips of absent regions re-read the last one (never used)
#pragma unroll
        for (int u = 0; u < NFL; ++u) {
            const int i = tid + u * kAT;
            typedef __attribute__((address_space(3))) void* lds_void;
            typedef const __attribute__((address_space(1))) void* glb_void;
            if (i < n4)
                __builtin_amdgcn_global_load_lds((glb_void)(fg + (size_t)min(i, last4) * 4), (lds_void)(feats_s + (size_t)(w * 64 + u * kAT) * 4), 16, 0, 0);
        }
    } else {
        for (int i = tid; i < AG * L * F; i += kAT) feats_s[i] = i < nf ? ldg(fg + i) : 0.f;
    }
    if (tid < 8) feats_s[AG * L * F + tid] = 0.f;
    static_assert(kD * 8 == 2 * kAT, "wlp staging assumes two entries per thread");
    wlp_s[tid] = (tid & 7) < F ? ldg(wlp + (tid >> 3) * F + (tid & 7)) : 0.f;
    wlp_s[tid + kAT] = (tid & 7) < F ? ldg(wlp + ((tid + kAT) >> 3) * F + (tid & 7)) : 0.f;
    const uint8_t* mg = mask + (size_t)n0 * mstride;
    const bool words = ((L | (int)mstride | (int)(reinterpret_cast<uintptr_t>(mask))) & 3) == 0;
    if (words) {
        const int LW = Lpad >> 2, lw = L >> 2;
        uint32_t* mk_w = reinterpret_cast<uint32_t*>(mk_s);
        for (int idx = tid; idx < AG * LW; idx += kAT) {
            const int sreg = idx / LW, jw = idx - sreg * LW;
            mk_w[idx] = (sreg < nreg && jw < lw) ? *(const CF_GLOBAL uint32_t*)(mg + (size_t)sreg * mstride + 4 * jw) : 0x01010101u;
        }
    } else {
        for (int s = 0; s < AG; ++s)
            for (int j = tid; j < Lpad; j += kAT) mk_s[s * Lpad + j] = (s < nreg && j < L) ? *(const CF_GLOBAL uint8_t*)(mg + (size_t)s * mstride + j) : (uint8_t)1;
    }
}
template <bool BWD, int AG, bool STAGED = false>
__device__ __forceinline__ void attc2_body(const Attc2Args& a, const int r, const int n0, const int N, float* scratch, float* persist,
                                           const int stamp_wg = -1) {
    constexpr int kAG = AG;                   // rows 2*AG .. 15 of the MFMA tile are dead (zero operand rows)
    const int tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int L = a.L[r], Lpad = a.Lpad[r], LT = a.LT[r], F = a.F, LS = Lpad + 4;
    constexpr int LD = kD + 4, NW = kAT / 64;
    float* sc_s = scratch;                    // [16][LS]  scores -> p (fwd) / dp -> ds (bwd); zero for j >= L
    float* vin_s = sc_s + kTile * LS;         // [16][LD]  qt / dxbar rows (m = 2*region + head)
    float* red_s = vin_s + kTile * LD;        // [16][LD]  K-split partial of the second table product
    float* u_s = red_s + kTile * LD;          // [16][8]
    float* w_s = u_s + kTile * 8;             // [16][8]
    float* wlp_s = persist;                   // [128][8]  Wlp, zero padded to 8 marks (all loops over marks run to 8)
    float* feats_s = wlp_s + kD * 8;          // [8][L][F] (+8 floats of slack: the padded mark loops read one past)
    uint8_t* mk_s = reinterpret_cast<uint8_t*>(feats_s + kAG * L * F + 8);   // [8][Lpad]
    const int nreg = min(kAG, N - n0);        // regions present in this workgroup

    CF_STAMP2(0);
    if (a.tall && tid == 0 && stamp_wg >= 0) a.tall[stamp_wg * 2] = __builtin_amdgcn_s_memtime();
    const int nblk = (L + 63) / 64;
    float* stat_s = red_s;                    // [2][16][16] per-(row, column block) softmax statistics (free until pass 5's reduction)
    static_assert(2 * kTile * 16 <= kTile * LD, "statistics fit in the K-split partial buffer");
    // The workgroup's time goes to the CU's vector-memory path (64 bytes per clock; ~620 KB per L = 400 workgroup, most of it the
    // two positional tables out of L2) and to its matrix pipes; a wave stalls at ISSUE once the path's queue is full.  So the
    // first barrier stands in front of the bulk loads, not behind them: only the operand rows, Wlp and the masks are fetched
    // before it, the table operand and the features are requested behind it and the products start on the first chunk to land.
    FragNN<4, 8> ft0;                         // PE^T operand of this wave's first column block of pass 2: all of K = 128 in registers
    const float* fg = a.feats[r] + (size_t)n0 * L * F;
    const int nf = nreg * L * F;
    constexpr int NFL = AG == 8 ? 12 : AG == 4 ? 6 : AG == 2 ? 3 : 2;         // 16-byte feature loads per thread: L <= 438 (AG = 1: 585)
    const bool f16b = ((L * F) & 3) == 0 && kAG * L * F <= NFL * 4 * kAT;     // one round of copies (L = 400, 8 regions: 22,400 of 24,576 floats)
    float4 fv12[NFL];
    {
        const uint8_t* mg = a.mask[r] + (size_t)n0 * a.mstride[r];
        const bool words = ((L | (int)a.mstride[r] | (int)(reinterpret_cast<uintptr_t>(a.mask[r]))) & 3) == 0;
        const int LW = Lpad >> 2, lw = L >> 2;
        constexpr int MW = kAG >= 2 ? kAG / 2 : 1;       // mask words per thread: kAG * (Lpad / 4 <= 256) / 512
        float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            const int m = tid >> 5, c4 = tid & 31;               // 16 rows x 32 float4
            if ((m >> 1) < nreg) vv = ldg4(a.vin[r] + (size_t)n0 * 256 + m * kD + c4 * 4);
        }
        const float wl0 = !STAGED && (tid & 7) < F ? ldg(a.wlp[r] + (tid >> 3) * F + (tid & 7)) : 0.f;
        const float wl1 = !STAGED && (tid & 7) < F ? ldg(a.wlp[r] + ((tid + kAT) >> 3) * F + (tid & 7)) : 0.f;      // kD * 8 = 2 * kAT entries
        uint32_t mwv[MW];
        if (!STAGED && words) {
#pragma unroll
            for (int k = 0; k < MW; ++k) {
                const int idx = tid + k * kAT, sreg = idx / LW, jw = idx - sreg * LW;
                mwv[k] = (idx < kAG * LW && sreg < nreg && jw < lw) ? *(const CF_GLOBAL uint32_t*)(mg + (size_t)sreg * a.mstride[r] + 4 * jw) : 0x01010101u;
            }
        }
        // the feature strips (the HBM stream: longest latency, needed last) and the first chunk of the PE^T operand are requested
        // here, behind the small loads the barrier waits for and in front of it: ~27 KB per wave, about as long to issue as the
        // operand rows take to arrive.  Always NFL feature loads, no branch around them and no select behind them: the wait counts
        // stay exact (strips past the end re-read the last one; when the 16-byte path does not apply the values are not used).
        {
            if (!STAGED) {
                const float* fa = reinterpret_cast<const float*>(reinterpret_cast<uintptr_t>(fg) & ~(uintptr_t)15);
                const int last4 = f16b ? (nf >> 2) - 1 : max((nf >> 2) - 2, 0);      // (aligned down: stay one strip inside)
#pragma unroll
                for (int u = 0; u < NFL; ++u) fv12[u] = ldg4(fa + (size_t)min(tid + u * kAT, last4) * 4);
            }
            const float* bm_ = a.pet[r] + min(w, nblk - 1) * 64;
            ft0.bp = bm_ + (size_t)(lq * 4) * LT + 4 * lr;
            ft0.ldb = LT;
            frag_chunk_nn(ft0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        stat_s[tid] = tid < kTile * 16 ? 0.f : -INFINITY;      // block sums | block maxima of column blocks nobody owns
        if (tid < kTile) u_s[tid * 8 + 7] = 0.f;               // (mark 7 does not exist: the epilogue multiplies it by a zero feature)
        {
            const int m = tid >> 5;                                // zero tail of the score rows (K padding of pass 5)
            for (int j = L + (tid & 31); j < Lpad; j += 32) sc_s[m * LS + j] = 0.f;
        }
        *reinterpret_cast<float4*>(vin_s + (tid >> 5) * LD + (tid & 31) * 4) = vv;
        static_assert(kD * 8 == 2 * kAT, "wlp staging assumes two entries per thread");
        if (!STAGED) {
            wlp_s[tid] = wl0;
            wlp_s[tid + kAT] = wl1;
        }
        if (STAGED) {
        } else if (words) {
            uint32_t* mk_w = reinterpret_cast<uint32_t*>(mk_s);
#pragma unroll
            for (int k = 0; k < MW; ++k)
                if (tid + k * kAT < kAG * LW) mk_w[tid + k * kAT] = mwv[k];
        } else {
            for (int s = 0; s < kAG; ++s)
                for (int j = tid; j < Lpad; j += kAT)
                    mk_s[s * Lpad + j] = (s < nreg && j < L) ? *(const CF_GLOBAL uint8_t*)(mg + (size_t)s * a.mstride[r] + j) : (uint8_t)1;
        }
    }
    CF_STAMP2(10);
    __syncthreads();
    CF_STAMP2(1);
    // ---- (1) u[m][f] = sum_e vin[m][e] Wlp[e][f] on the matrix cores, by the last wave (the one with the fewest column blocks):
    //      one [16 x 128] . [128 x 16] tile whose B operand comes straight from global (columns f >= F repeat column F - 1 and
    //      are not stored).  Its loads are issued first, so they return first.
    float ub[32];
    if (w == NW - 1) {
        const float* wp_ = a.wlp[r] + (size_t)(lq * 4) * F + min(lr, F - 1);
#pragma unroll
        for (int st8 = 0; st8 < 8; ++st8)
#pragma unroll
            for (int i = 0; i < 4; ++i) ub[st8 * 4 + i] = ldg(wp_ + (size_t)(st8 * 16 + i) * F);
    }
    // ---- (2a) t[m][j] = vin[m] . PE^T[:, j];  wave w takes column blocks w and w + 8 (L <= 1024).  Loads and products are
    //      interleaved in program order (a wave that meets a full memory queue stalls where it stands).
    f32x4 acc2[2][4];
    zero_acc(acc2[0]);
    zero_acc(acc2[1]);
    {
        const float* ap = vin_s + lr * LD + lq * 4;
        auto products = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            if (w < nblk)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float4 av4 = *reinterpret_cast<const float4*>(ap + (c * 2 + k) * 16);
                    const float av[4] = {av4.x, av4.y, av4.z, av4.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float* bv = reinterpret_cast<const float*>(&ft0.ring[c][k][i]);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[0][t] = mfma4(av[i], bv[t], acc2[0][t]);
                    }
                }
        };
        frag_chunk_nn(ft0, 1, 1);
        frag_chunk_nn(ft0, 2, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (w == NW - 1) {
            f32x4 ua = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st8 = 0; st8 < 8; ++st8) {
                const float4 av = *reinterpret_cast<const float4*>(ap + st8 * 16);
                ua = mfma4(av.x, ub[st8 * 4], ua);
                ua = mfma4(av.y, ub[st8 * 4 + 1], ua);
                ua = mfma4(av.z, ub[st8 * 4 + 2], ua);
                ua = mfma4(av.w, ub[st8 * 4 + 3], ua);
            }
            if (lr < F)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) u_s[(lq * 4 + ii) * 8 + lr] = ua[ii];
        }
        __builtin_amdgcn_sched_barrier(0);
        products(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        frag_chunk_nn(ft0, 3, 3);
        __builtin_amdgcn_sched_barrier(0);
        products(std::integral_constant<int, 1>{});
        products(std::integral_constant<int, 2>{});
        products(std::integral_constant<int, 3>{});
    }
    if (w + NW < nblk) {
        FragNN<4, 8> ft;
        frag_load_nn(ft, a.pet[r] + (w + NW) * 64, LT);
        frag_mma_nn(ft, vin_s, LD, acc2[1]);
    }
    CF_STAMP2(8);
    // ---- features to LDS (behind the products)
    if (STAGED) {
    } else if (f16b) {
        const int n4 = (kAG * L * F) >> 2;
#pragma unroll
        for (int u = 0; u < NFL; ++u) {
            const int i = tid + u * kAT;
            if (i < n4) *reinterpret_cast<float4*>(feats_s + (size_t)i * 4) = fv12[u];
        }
        if (tid < 8) feats_s[kAG * L * F + tid] = 0.f;
    } else {
        for (int i = tid; i < kAG * L * F + 8; i += kAT) feats_s[i] = i < nf ? ldg(fg + i) : 0.f;
    }
    // pass 5's PE operand (2 column groups x 4 quarters of the bin range): requested here, where the waves that finished their
    // products wait for the others anyway (the issue itself stalls ~3 K cycles on the memory queue); in flight across the
    // epilogue, the softmax and pass 4
    const int cg = w & 1, kh = w >> 1, kper = Lpad / 4;
    FragNN<4, 8> fp5;
    frag_load_nn_rt(fp5, a.pe[r] + (size_t)(kh * kper) * kD + cg * 64, kD, kper / 32);
    CF_STAMP2(9);
    __syncthreads();
    CF_STAMP2(2);
    // ---- (2b) epilogue of pass 2 in the accumulator layout (lane: rows 4 lq + ii, columns j0 .. j0 + 3 of its block):
    //      + f_j . u[m]; forward: scale, mask, then the block's share of the softmax (block maximum, exponentials, block sum);
    //      backward: the block's share of <p, dp>.  One barrier later every lane combines the <= 16 block statistics of its rows
    //      and writes p (forward; also to global) / ds (backward) into the score tile, the A operand of pass 5.
    float xs[2][4][4];                           // [block][ii][t]: exp(score - block maximum) (fwd) / dp (bwd)
    float4 pv[2][4];                             // backward: saved p of the lane's entries
    float bstat[2][4];                           // forward: block maximum of row ii
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        const int jb = w + bi * NW;
        if (jb >= nblk) continue;
        const int j0 = jb * 64 + 4 * lr;
        if (BWD) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int m = lq * 4 + ii;
                pv[bi][ii] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((m >> 1) < nreg && (m >> 1) < kAG) {
                    const float* pg = a.p[r] + ((size_t)n0 * 2 + m) * L;
                    if (j0 + 3 < L && (L & 3) == 0) pv[bi][ii] = (CF_TRUNK_NT & 2) ? ldg4_nt(pg + j0) : ldg4(pg + j0);
                    else {
                        if (j0 < L) pv[bi][ii].x = ldg(pg + j0);
                        if (j0 + 1 < L) pv[bi][ii].y = ldg(pg + j0 + 1);
                        if (j0 + 2 < L) pv[bi][ii].z = ldg(pg + j0 + 2);
                        if (j0 + 3 < L) pv[bi][ii].w = ldg(pg + j0 + 3);
                    }
                }
            }
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {            // the lane's rows 4q..4q+3 = regions 2q, 2q+1 (two heads each)
            const int s = lq * 2 + sp;
            if (s >= kAG) continue;
            float fv[4][8];                           // marks of bins j0..j0+3 of region s: 28 consecutive floats
            if (j0 + 3 < L && F == 7) {
                const float4* fp4 = reinterpret_cast<const float4*>(feats_s + (size_t)(s * L + j0) * 7);
                float tmp[28];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const float4 t4 = fp4[k];
                    tmp[4 * k] = t4.x;
                    tmp[4 * k + 1] = t4.y;
                    tmp[4 * k + 2] = t4.z;
                    tmp[4 * k + 3] = t4.w;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int f = 0; f < 8; ++f) fv[t][f] = f < 7 ? tmp[t * 7 + f] : 0.f;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int f = 0; f < 8; ++f) fv[t][f] = (j0 + t < L && f < F) ? feats_s[(size_t)(s * L + j0 + t) * F + f] : 0.f;
            }
            const uint32_t mw = *reinterpret_cast<const uint32_t*>(mk_s + s * Lpad + j0);      // j0 + 3 < Lpad (Lpad >= 64 nblk)
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                const int ii = sp * 2 + hd, m = lq * 4 + ii;
                const float4 u0 = *reinterpret_cast<const float4*>(u_s + m * 8), u1 = *reinterpret_cast<const float4*>(u_s + m * 8 + 4);
                float bm = -INFINITY;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float x = acc2[bi][t][ii];
                    x = fmaf(fv[t][0], u0.x, x);
                    x = fmaf(fv[t][1], u0.y, x);
                    x = fmaf(fv[t][2], u0.z, x);
                    x = fmaf(fv[t][3], u0.w, x);
                    x = fmaf(fv[t][4], u1.x, x);
                    x = fmaf(fv[t][5], u1.y, x);
                    x = fmaf(fv[t][6], u1.z, x);
                    x = fmaf(fv[t][7], u1.w, x);
                    if (!BWD) {
                        x = x * a.rscale;
                        if ((mw >> (8 * t)) & 0xffu) x = kMaskFill;
                        if (j0 + t < L) bm = fmaxf(bm, x);
                    }
                    xs[bi][ii][t] = x;
                }
                float part;
                if (!BWD) {
                    bm = group16_max(bm);                    // finite: the block holds at least one column j < L
                    part = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float e = j0 + t < L ? __expf(xs[bi][ii][t] - bm) : 0.f;
                        xs[bi][ii][t] = e;
                        part += e;
                    }
                    bstat[bi][ii] = bm;
                } else {
                    const float4 p4 = pv[bi][ii];
                    part = (p4.x * xs[bi][ii][0] + p4.y * xs[bi][ii][1]) + (p4.z * xs[bi][ii][2] + p4.w * xs[bi][ii][3]);
                }
                part = group16_sum(part);
                if (lr == 0) {
                    stat_s[m * 16 + jb] = part;
                    if (!BWD) stat_s[kTile * 16 + m * 16 + jb] = bm;
                }
            }
        }
    }
    __syncthreads();
    CF_STAMP2(3);
    // ---- (3) softmax over the bins (fwd) / its backward (bwd): combine the block statistics of the lane's rows
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int m = lq * 4 + ii, s = m >> 1;
        if (s >= kAG || w >= nblk) continue;
        const bool present = s < nreg;
        float fac[2] = {0.f, 0.f}, dot = 0.f;
        const float bsum = stat_s[m * 16 + lr];      // lane lr <-> column block lr of row m (unowned blocks: sum 0, maximum -inf)
        if (!BWD) {
            const float bmax = stat_s[kTile * 16 + m * 16 + lr];
            const float mx = group16_max(bmax);
            const float z = group16_sum(bsum * __expf(bmax - mx));
            const float rz = 1.0f / z;
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) fac[bi] = (w + bi * NW < nblk) ? __expf(bstat[bi][ii] - mx) * rz : 0.f;
        } else {
            dot = group16_sum(bsum);
        }
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
            const int jb = w + bi * NW;
            if (jb >= nblk) continue;
            const int j0 = jb * 64 + 4 * lr;
            float4 o;
            if (!BWD) {
                o = make_float4(xs[bi][ii][0] * fac[bi], xs[bi][ii][1] * fac[bi], xs[bi][ii][2] * fac[bi], xs[bi][ii][3] * fac[bi]);
                if (present) {
                    float* pg = a.p[r] + ((size_t)n0 * 2 + m) * L;
                    if (j0 + 3 < L && (L & 3) == 0) {
                        if (CF_TRUNK_NT & 1) stg4_nt(pg + j0, o);
                        else stg4(pg + j0, o);
                    }
                    else {
                        if (j0 < L) stg(pg + j0, o.x);
                        if (j0 + 1 < L) stg(pg + j0 + 1, o.y);
                        if (j0 + 2 < L) stg(pg + j0 + 2, o.z);
                        if (j0 + 3 < L) stg(pg + j0 + 3, o.w);
                    }
                }
            } else {
                const uint32_t mw = *reinterpret_cast<const uint32_t*>(mk_s + s * Lpad + j0);
                const float4 p4 = pv[bi][ii];
                o.x = (mw & 0xffu) ? 0.f : p4.x * (xs[bi][ii][0] - dot) * a.rscale;
                o.y = (mw & 0xff00u) ? 0.f : p4.y * (xs[bi][ii][1] - dot) * a.rscale;
                o.z = (mw & 0xff0000u) ? 0.f : p4.z * (xs[bi][ii][2] - dot) * a.rscale;
                o.w = (mw & 0xff000000u) ? 0.f : p4.w * (xs[bi][ii][3] - dot) * a.rscale;      // j >= L: p = 0 there
            }
            *reinterpret_cast<float4*>(sc_s + m * LS + j0) = o;      // j0 + 3 < Lpad
        }
    }
    // LPR lanes per row m: 32 with all 16 rows live, a whole wave per row when at most 8 are
    constexpr int LPR = AG == 8 ? 32 : 64, LSH = AG == 8 ? 5 : 6;
    const int gm = tid >> LSH, sub = tid & (LPR - 1);
    const bool row_live = gm < 2 * kAG;
    __syncthreads();
    CF_STAMP2(4);
    // ---- (4) w[m][f] = sum_j sc[m][j] f_j[f]
    {
        float acc[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = 0.f;
        const float* fp = feats_s + (row_live ? gm >> 1 : 0) * L * F;
        if (F == 7) {      // four bins per step: their 28 marks are seven 16-byte LDS reads (the tail past L is multiplied by zeros)
            for (int j = 4 * sub; j < (row_live ? L : 0); j += 4 * LPR) {
                const float4 p4 = *reinterpret_cast<const float4*>(sc_s + gm * LS + j);
                if (j + 3 < L) {
                    const float4* f4 = reinterpret_cast<const float4*>(fp + (size_t)j * 7);
                    float tmp[28];
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        const float4 t4 = f4[k];
                        tmp[4 * k] = t4.x;
                        tmp[4 * k + 1] = t4.y;
                        tmp[4 * k + 2] = t4.z;
                        tmp[4 * k + 3] = t4.w;
                    }
                    const float pj[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int f = 0; f < 7; ++f) acc[f] = fmaf(pj[t], tmp[t * 7 + f], acc[f]);
                } else {
                    const float pj[4] = {p4.x, p4.y, p4.z, p4.w};
                    for (int t = 0; t < 4 && j + t < L; ++t)
#pragma unroll
                        for (int f = 0; f < 7; ++f) acc[f] = fmaf(pj[t], fp[(size_t)(j + t) * 7 + f], acc[f]);
                }
            }
        } else {
            for (int j = sub; j < (row_live ? L : 0); j += LPR) {
                const float pv = sc_s[gm * LS + j];
#pragma unroll
                for (int f = 0; f < 8; ++f) acc[f] = fmaf(pv, fp[j * F + f], acc[f]);           // entries f >= F are dropped below
            }
        }
        // 8 sums over LPR lanes in 9 (10) exchanges: each step trades half of the values with the partner lane
        float h4[4], h2[2], h1;
        {
            const bool up = sub & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float send = up ? acc[i] : acc[4 + i], keep = up ? acc[4 + i] : acc[i];
                h4[i] = keep + __shfl_xor(send, 1, 64);
            }
        }
        {
            const bool up = sub & 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float send = up ? h4[i] : h4[2 + i], keep = up ? h4[2 + i] : h4[i];
                h2[i] = keep + __shfl_xor(send, 2, 64);
            }
        }
        {
            const bool up = sub & 4;
            const float send = up ? h2[0] : h2[1], keep = up ? h2[1] : h2[0];
            h1 = keep + __shfl_xor(send, 4, 64);
        }
        h1 += __shfl_xor(h1, 8, 64);
        h1 += __shfl_xor(h1, 16, 64);
        if (LPR == 64) h1 += __shfl_xor(h1, 32, 64);
        if (sub < 8 && row_live) {
            const int f = (sub & 1) * 4 + (sub & 2) + ((sub >> 2) & 1);
            const float sv = f < F ? h1 : 0.f;
            w_s[gm * 8 + f] = sv;
            if ((gm >> 1) < nreg) stg(a.w[r] + ((size_t)n0 * 2 + gm) * 8 + f, sv);
        }
    }
    CF_STAMP2(5);
    // ---- (5) out[m][e] = sum_j sc[m][j] PE[j][e]: 2 column groups x 4 quarters of the bin range
    {
        f32x4 acc[4];
        zero_acc(acc);
        frag_mma_nn_rt(fp5, sc_s + kh * kper, LS, kper / 32, acc);
        CF_STAMP2(6);
        __syncthreads();                       // the score tile, the operand rows and the marks are free now
        // K-split partials of all four quarters to LDS ([16][LD] each); then wave (cg, kh) finishes the rows 4 lq + kh of its
        // column group: three partials + its own, + Wlp w, store -- a quarter of the rows per wave instead of all of them in two
        // (four [16][LD] buffers laid over the score tile, the operand rows, the partial buffer and the first 64 floats of u:
        //  contiguous in that order and all free now; 16 LS + 32 LD + 128 >= 64 LD for every Lpad >= 256)
        float* const parts[4] = {sc_s, sc_s + kTile * LD, sc_s + 2 * kTile * LD, sc_s + 3 * kTile * LD};
        {
            float* part = sc_s + kh * kTile * LD;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                *reinterpret_cast<float4*>(part + (lq * 4 + ii) * LD + cg * 64 + 4 * lr) = make_float4(acc[0][ii], acc[1][ii], acc[2][ii], acc[3][ii]);
        }
        __syncthreads();
        {
            const int m = lq * 4 + kh, e0 = cg * 64 + 4 * lr;
            // Wlp rows of this lane's four output columns: 8 sixteen-byte reads instead of 32 scalar ones
            float wl[4][8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 w0 = *reinterpret_cast<const float4*>(wlp_s + (e0 + t) * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(wlp_s + (e0 + t) * 8 + 4);
                wl[t][0] = w0.x, wl[t][1] = w0.y, wl[t][2] = w0.z, wl[t][3] = w0.w;
                wl[t][4] = w1.x, wl[t][5] = w1.y, wl[t][6] = w1.z, wl[t][7] = w1.w;
            }
            const float4 wa = *reinterpret_cast<const float4*>(w_s + m * 8), wb = *reinterpret_cast<const float4*>(w_s + m * 8 + 4);
            const float wm[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            const float4 r0 = *reinterpret_cast<const float4*>(parts[0] + m * LD + e0);
            const float4 r1 = *reinterpret_cast<const float4*>(parts[1] + m * LD + e0);
            const float4 r2 = *reinterpret_cast<const float4*>(parts[2] + m * LD + e0);
            const float4 r3 = *reinterpret_cast<const float4*>(parts[3] + m * LD + e0);
            float v[4] = {(r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z), (r0.w + r1.w) + (r2.w + r3.w)};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int f = 0; f < 8; ++f) v[t] = fmaf(wm[f], wl[t][f], v[t]);
            if ((m >> 1) < nreg && (m >> 1) < kAG) stg4(a.vout[r] + ((size_t)n0 * 2 + m) * kD + e0, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
    CF_STAMP2(7);
    if (a.tall && tid == 0 && stamp_wg >= 0) a.tall[stamp_wg * 2 + 1] = __builtin_amdgcn_s_memtime();
}

template <bool BWD, int AG>
__global__ __launch_bounds__(kAT) void k_attc2(Attc2Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // resolutions in reverse launch order: the long-sequence workgroups (last binsize) are dispatched first
    const int r = gridDim.y - 1 - blockIdx.y;
    attc2_body<BWD, AG>(a, r, blockIdx.x * AG, a.N, smem, smem + attc2_scratch_floats(a.L[r]), blockIdx.y * gridDim.x + blockIdx.x);
}

// Regions per workgroup.  These kernels are latency-bound (a handful of workgroups per CU, dependent phases): the
// time of a workgroup is ~19 K cycles + 5 K per region, and dead MFMA rows cost nothing that matters.  So take the
// fewest regions per workgroup that still lets every long-sequence workgroup be resident at once.
inline int attc2_regions_per_wg(int N, int cap) {
    for (int ag = 1; ag < kAGMax; ag *= 2)
        if ((N + ag - 1) / ag <= cap) return ag;
    return kAGMax;
}
template <bool BWD>
inline const void* attc2_kernel(int ag) {
    switch (ag) {
        case 1: return (const void*)k_attc2<BWD, 1>;
        case 2: return (const void*)k_attc2<BWD, 2>;
        case 4: return (const void*)k_attc2<BWD, 4>;
        default: return (const void*)k_attc2<BWD, 8>;
    }
}

}  // namespace cf

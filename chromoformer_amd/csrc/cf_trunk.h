// cf_trunk.h -- the centre-row trunk (Embedding layer + Pairwise stack) of ONE gene at one resolution in ONE workgroup
// (included by cf_kernels.h behind the kernels whose bodies it runs).
//
// Every row of the centre-row stage belongs to one gene: the promoter's centre embedding (net.py:59), the query rows of its
// i_max (promoter, pCRE) pairs through the Pairwise layers (modules.py:150-195, 205-206; net.py:105-139), and -- backwards --
// the join of their gradients at the promoter embedding.  Nothing crosses genes before the weight-gradient reductions.  Round 2
// ran the stage as 8 + 10 launches (query chain -> attention -> post chain per layer and direction, the join, the 7-mark
// projection partials), each one wave of <= 192 workgroups that started with an HBM round trip for rows the launch in front
// had just written.  Here workgroup (gene g, resolution r) walks the whole chain itself:
//
//   forward   x0 -> [q chain -> attention (1 region, vector ALUs) -> post chain + lin_proj_p]             Embedding layer
//                -> [q chain] -> { attention (S regions, matrix cores) -> post chain (+ next layer's q chain) } x layers
//   backward  { post chain -> attention -> q chain } x layers (last to first) -> join + lin_proj_p backward
//                -> [post chain -> attention -> q chain] of the Embedding layer -> 7-mark projection partials of the gene
//
// The Pairwise phases and both attentions are the BODIES of the stand-alone kernels (cf_kernels.h, cf_attc1.h, cf_attc2.h), unchanged
// arithmetic in unchanged order; the Embedding layer's one-row chains run on the vector ALUs (cf_trunk_e.h, round 4: same products,
// another summation order -- the fused and the stand-alone path agree to fp32 rounding).  Between two phases the rows travel
// through global memory exactly as between two launches -- but written and read by the same workgroup, i.e. out of the CU's
// own L1 / the XCD's L2 behind a workgroup barrier (workgroup-scope release / acquire: all waves of a workgroup share the L1)
// instead of across a launch boundary -- and the S regions' features, masks and Wlp are staged in LDS ONCE for all Pairwise
// layers of a direction (x_pcre is never updated, modules.py:218-219).
#pragma once

namespace cf {

constexpr int kMaxPairLayers = 4;      // (of the fused trunk kernels; the stand-alone kernels take up to kLpMaxSeg / 2 = 8)

// one centre-row layer of one resolution: operands, saved activations, gradient buffers (device pointers, fixed at cf_bind)
struct CentreLayerDev {
    const float *wq_t, *wk, *wv_t, *wo_t, *bo, *g1, *be1, *w1_t, *b1, *w2_t, *b2, *g2, *be2;      // forward (_t: tiled copies)
    const float *wq, *wk_t, *wv, *wo, *w1, *w2;                                                     // backward
    float *q, *qt, *p, *w, *xbar, *a, *xh1, *rs1, *y1, *hdn, *xh2, *rs2, *out, *xin;
    float *dt2, *dpre1, *dt1, *da, *dxbar, *dqt, *du, *dq, *dx, *partial;
};
struct TrunkResDev {
    CentreLayerDev E, P[kMaxPairLayers];
    const float *pe, *pe2, *pet2;      // positional table [L][128]; padded layouts of the attention bodies (cf_attc2.h)
    const float *wlp_e, *wlp_p;        // lin_proj / lin_proj_pcre  [128, F]
    const float *lin_p_t, *lin_p;      // lin_proj_p [128, 128]: tiled copy (forward), row-major (backward)
    float *ex0, *featc, *xp0, *dxp0, *edout, *rx0, *drx0;
    float *lp_part_e, *lp_part_p;
    int L, Lpad, LT, pad_;
};
struct TrunkArgs {
    const TrunkResDev* tab;            // [n_res]
    const float* pfeats[kMaxRes];
    const uint8_t* pmask[kMaxRes];
    long long pmstride[kMaxRes];
    const float* cfeats[kMaxRes];
    const uint8_t* cmask[kMaxRes];
    long long cmstride[kMaxRes];
    const float* dhin;                 // [B, n_res * 128]   (backward)
    int B, S, T, F, n_res, pair_layers, save;
    float scale, rscale;
    // tiled-copy units riding in the forward launch (workgroups of the row blockIdx.y == n_res loop over them): the Regulation +
    // head weights, which nothing reads before the Regulation stack
    const RetileUnit* rt_units;
    const float* rt_params;
    float* rt_tiled;
    float* rt_tiledT;
    int rt_n;
    RecordArgs rec;                    // backward, optional (rec.cursor != null): the step's logits / labels / loss into the epoch logs (cf_record_step_bwd)
    int* adv_cursor;                   // forward, optional: the batch cursor of a gather that shared the launch in front (k_prologue_gather): advanced here
    const LpJob* lp_jobs;              // backward: the 7-mark projection jobs, [2 r] = Embedding, [2 r + 1] = Pairwise of resolution r
    // backward, single-GPU training: the first rd_n weight-gradient tiles of the Regulation bucket (all equally long), with AdamW in
    // their epilogues, walked by an extra row of workgroups on the CUs the trunk leaves idle (k_trunk_bwd; cf_rider_arm)
    const WgTile* rd_tiles;
    int rd_n, rd_batch;
    AdamFuse rd_opt;
    unsigned long long* tdbg;          // optional shader-clock stamps of workgroup (gene 0, longest resolution), one per phase boundary
};

// a table field through the constant address space: the pointer stays in SGPRs (cf_reg8.h, load_layer)
template <class T>
__device__ __forceinline__ T ldc(const T* p) {
    static_assert(sizeof(T) == 8 || sizeof(T) == 4, "pointer or int fields");
    if constexpr (sizeof(T) == 8) {
        const unsigned long long v = *(const __attribute__((address_space(4))) unsigned long long*)p;
        return __builtin_bit_cast(T, v);
    } else {
        const unsigned v = *(const __attribute__((address_space(4))) unsigned*)p;
        return __builtin_bit_cast(T, v);
    }
}
#define TF(field) ldc(&R->field)

// LDS of the trunk kernels: [scratch | persistent attention part]; the chain bodies' tiles and the one-region attention (which
// runs before / after everything that needs the persistent part) lie over it from the start.
__host__ __device__ inline size_t trunk_scratch_floats(int L, int dff_max) {
    const size_t chain = (size_t)3 * kTile * (kD + 4) + (size_t)kTile * ((dff_max > 256 ? dff_max : 256) + 4);      // post_fwd / post_bwd tiles
    const size_t att = attc2_scratch_floats(L);
    return (chain > att ? chain : att) + 4;
}
__host__ __device__ inline size_t trunk_smem(int L, int F, int dff_max) {
    const size_t two = trunk_scratch_floats(L, dff_max) * sizeof(float) + attc2_persist_bytes(L, F, kAGMax);
    const size_t one = attc1_smem(L, F);
    const size_t rt = (size_t)(kAT / 64) * 16 * 20 * sizeof(float);
    const size_t lp = (size_t)80 * (kD + 8) * sizeof(float);      // trunk_lp's operand rows (kLpRowsMax)
    size_t m = two > one ? two : one;
    m = m > rt ? m : rt;
    return m > lp ? m : lp;
}

// How the phases are put together matters more than anything in them (all measured / compiled, round 3):
//   * as NOINLINE calls each body keeps the register allocation it has as a stand-alone kernel -- but the callee saves the 108
//     callee-saved VGPRs of the calling convention it uses (every 256-register body: 108 scratch stores + loads per lane and call)
//     and receives its uniform arguments in VGPRs: every table field becomes a vector load, the attention body has 212 flat
//     accesses and 287 divergent branches.  The fused forward kernel took 134 us against 122 us for the eight launches it replaces;
//   * inlined one behind the other as they are, the compiler merges the address arithmetic the phases have in common and keeps it
//     alive through the 244-register attention bodies: 477 spilled VGPRs (two attentions back to back alone: 104);
//   * inlined into the cases of a switch inside a loop over a phase counter, the common part is hoisted out of the loop: 218 spills.
// What the kernels below do: every phase forceinline, STRAIGHT-LINE code (the number of Pairwise layers is a template parameter), and
// launder() in front of every phase -- the context passes through an empty `asm volatile`, so each phase recomputes what it needs
// from opaque inputs, as the stand-alone kernels do.  21 (forward) / 72 (backward) spilled VGPRs remain, none inside a loop.
#ifndef CF_STAMP_Y      // which row of workgroups writes the phase stamps (0: the longest resolution; tools/trunk_stamps.py)
#define CF_STAMP_Y 0
#endif
#define CF_PHASE static __device__ __forceinline__

struct TrunkCtx {      // what every phase needs, by value (uniform: lives in SGPRs)
    const TrunkResDev* R;
    const float* feats;        // the phase's feature array (promoter or pCRE) of resolution r
    const uint8_t* mask;
    long long mstride;
    int g, S, T, F, save;
    float scale, rscale;
};
__device__ __forceinline__ void launder(TrunkCtx& c, float*& smem, float*& persist) {
    asm volatile("" : "+s"(c.R), "+s"(c.feats), "+s"(c.mask), "+s"(c.mstride), "+s"(c.g), "+s"(c.S), "+s"(c.T), "+s"(c.F), "+s"(c.save),
                 "+s"(c.scale), "+s"(c.rscale));
    // LDS pointers: through the local address space (32-bit offsets) and back -- a generic pointer truncated to 32 bits and
    // widened again is NOT an LDS address any more (its aperture bits are gone)
    typedef __attribute__((address_space(3))) float* lds_ptr;
    unsigned a = (unsigned)(uintptr_t)(lds_ptr)smem, b = (unsigned)(uintptr_t)(lds_ptr)persist;
    asm volatile("" : "+s"(a), "+s"(b));
    smem = (float*)(lds_ptr)(uintptr_t)a;
    persist = (float*)(lds_ptr)(uintptr_t)b;
}
__device__ __forceinline__ void trunk_attc_args(Attc2Args& at, const TrunkCtx& c, const float* wlp, const float* vin, float* p, float* w, float* vout,
                                                int N) {
    const TrunkResDev* R = c.R;
    at.feats[0] = c.feats;
    at.mask[0] = c.mask;
    at.mstride[0] = c.mstride;
    at.pe[0] = TF(pe2);
    at.pet[0] = TF(pet2);
    at.wlp[0] = wlp;
    at.vin[0] = vin;
    at.p[0] = p;
    at.w[0] = w;
    at.vout[0] = vout;
    at.L[0] = TF(L);
    at.Lpad[0] = TF(Lpad);
    at.LT[0] = TF(LT);
    at.N = N;
    at.F = c.F;
    at.scale = c.scale;
    at.rscale = c.rscale;
    at.tdbg = nullptr;
    at.tall = nullptr;
}
// chain tiles over the scratch part of the LDS
#define CF_CHAIN_TILES(scratch)                                                                       \
    float (*xs)[kD + 4] = reinterpret_cast<float (*)[kD + 4]>(scratch);                               \
    float (*as_)[kD + 4] = reinterpret_cast<float (*)[kD + 4]>((scratch) + kTile * (kD + 4));         \
    float (*ts)[kD + 4] = reinterpret_cast<float (*)[kD + 4]>((scratch) + 2 * kTile * (kD + 4));      \
    float* wide_raw = (scratch) + 3 * kTile * (kD + 4);                                               \
    (void)xs, (void)as_, (void)ts, (void)wide_raw

}  // namespace cf
#include "cf_trunk_e.h"      // the Embedding layer's one-row chains on the vector ALUs
namespace cf {

// ---- forward phases
template <bool BWD>
CF_PHASE void trunk_attc1_e(TrunkCtx c, float* smem) {
    const TrunkResDev* R = c.R;
    Attc2Args at;
    if (!BWD) trunk_attc_args(at, c, TF(wlp_e), TF(E.qt), TF(E.p), TF(E.w), TF(E.xbar), c.g + 1);
    else trunk_attc_args(at, c, TF(wlp_e), TF(E.dxbar), TF(E.p), TF(E.du), TF(E.dqt), c.g + 1);
    attc1_body<BWD>(at, 0, c.g, smem);
}
__device__ __forceinline__ void trunk_post_args(PostArgs& po, const CentreLayerDev* P, int save) {
    po.ain[0] = ldc(&P->xbar);
    po.wv[0] = ldc(&P->wv_t);
    po.wo[0] = ldc(&P->wo_t);
    po.bo[0] = ldc(&P->bo);
    po.g1[0] = ldc(&P->g1);
    po.be1[0] = ldc(&P->be1);
    po.w1[0] = ldc(&P->w1_t);
    po.b1[0] = ldc(&P->b1);
    po.w2[0] = ldc(&P->w2_t);
    po.b2[0] = ldc(&P->b2);
    po.g2[0] = ldc(&P->g2);
    po.be2[0] = ldc(&P->be2);
    po.a_out[0] = ldc(&P->a);
    po.xh1[0] = ldc(&P->xh1);
    po.rs1[0] = ldc(&P->rs1);
    po.y1[0] = ldc(&P->y1);
    po.hdn[0] = ldc(&P->hdn);
    po.xh2[0] = ldc(&P->xh2);
    po.rs2[0] = ldc(&P->rs2);
    po.save = save;
    po.lin_w[0] = nullptr;
    po.nq_wq[0] = nullptr;
}
CF_PHASE void trunk_qchain_p0(TrunkCtx c, float* smem) {
    const TrunkResDev* R = c.R;
    CF_CHAIN_TILES(smem);
    const CentreLayerDev* P0 = &R->P[0];
    QChainArgs q;
    q.x[0] = TF(xp0);
    q.xmap = RowMap{c.S, 1, 0, 0};         // the promoter row, once per pair (net.py:114-118)
    q.wq[0] = ldc(&P0->wq_t);
    q.wk[0] = ldc(&P0->wk);
    q.q[0] = ldc(&P0->q);
    q.qt[0] = ldc(&P0->qt);
    q.xcopy[0] = ldc(&P0->xin);
    q.N = c.g * c.S + c.S;
    qchain_fwd_body<kAT / 64>(q, 0, c.g * c.S, xs, as_);
}
// attention of Pairwise layer l over the gene's S regions; STAGED: features, masks and Wlp are in the persistent LDS part already
template <bool BWD, bool STAGED>
CF_PHASE void trunk_attc2_p(TrunkCtx c, int l, float* scratch, float* persist) {
    const TrunkResDev* R = c.R;
    const CentreLayerDev* P = &R->P[l];
    const int row0 = c.g * c.S, NB = row0 + c.S;
    Attc2Args at;
    if (!BWD) trunk_attc_args(at, c, TF(wlp_p), ldc(&P->qt), ldc(&P->p), ldc(&P->w), ldc(&P->xbar), NB);
    else trunk_attc_args(at, c, TF(wlp_p), ldc(&P->dxbar), ldc(&P->p), ldc(&P->du), ldc(&P->dqt), NB);
    attc2_body<BWD, kAGMax, STAGED>(at, 0, row0, NB, scratch, persist);
}
// features, masks and Wlp of the gene's S regions into the persistent LDS part, requested at the START of a launch: the feature
// strips (90 KB per L = 400 workgroup, straight out of HBM) land under the phases in front of the first Pairwise attention, which
// used to wait for them (tools/trunk_stamps.py: that attention took 54.6 K ticks against 44.1 K for the second layer's)
CF_PHASE void trunk_stage_p(TrunkCtx c, float* persist) {
    const TrunkResDev* R = c.R;
    attc2_stage<kAGMax>(c.feats, c.mask, c.mstride, TF(wlp_p), TF(L), TF(Lpad), c.F, c.g * c.S, c.g * c.S + c.S, persist);
}
template <int DFF>
CF_PHASE void trunk_post_p(TrunkCtx c, int l, int n_layers, float* smem) {
    const TrunkResDev* R = c.R;
    const CentreLayerDev* P = &R->P[l];
    const bool last = l + 1 == n_layers;
    const int S = c.S, row0 = c.g * S, NB = row0 + S;
    CF_CHAIN_TILES(smem);
    PostArgs po;
    trunk_post_args(po, P, c.save);
    po.x[0] = l == 0 ? (const float*)TF(xp0) : (const float*)ldc(&R->P[l > 0 ? l - 1 : 0].out);
    po.xmap = l == 0 ? RowMap{S, 1, 0, 0} : identity_map();
    po.out[0] = last ? TF(rx0) : ldc(&P->out);
    po.omap = last ? RowMap{S, c.T, 1, 1} : identity_map();      // tokens 1 .. S of the gene's Regulation input
    po.N = NB;
    if (!last) {      // the next layer's query chain on the tile the chain still holds
        const CentreLayerDev* Pn = &R->P[l + 1];
        po.nq_wq[0] = ldc(&Pn->wq_t);
        po.nq_wk[0] = ldc(&Pn->wk);
        po.nq_q[0] = ldc(&Pn->q);
        po.nq_qt[0] = ldc(&Pn->qt);
    }
    post_fwd_body<true, 128, DFF, kAT / 64>(po, 0, row0, NB, xs, as_, ts, reinterpret_cast<float (*)[PostFwdLds<DFF>::HW + 4]>(wide_raw));
}

// PL: pairwise_interaction.n_layers as a template parameter -- the phases follow each other in STRAIGHT-LINE code.  (As a loop over
// a phase counter with the bodies in the cases of a switch the compiler hoists what the phases have in common out of the loop
// and keeps it alive through all of them: 218 spilled VGPRs against 10 for the straight line.)
template <int DFF_E, int DFF_P, int PL>
__global__ __launch_bounds__(kAT) void k_trunk_fwd(TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NWV = kAT / 64;
    if ((int)blockIdx.y >= a.n_res) {      // tiled-copy units on the CUs the trunk leaves idle
        float (*tp)[16][20] = reinterpret_cast<float (*)[16][20]>(smem);
        for (int u = blockIdx.x; u < a.rt_n; u += gridDim.x) {
            retile_unit_lds<NWV>(a.rt_params, a.rt_tiled, a.rt_tiledT, a.rt_units[u], tp);
            __syncthreads();
        }
        return;
    }
    if (a.adv_cursor && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && a.adv_cursor[0] < a.adv_cursor[1]) a.adv_cursor[0] += 1;
    // resolutions in reverse launch order: the long-sequence workgroups (last binsize) are dispatched first
    const int r = a.n_res - 1 - (int)blockIdx.y;
    const TrunkResDev* R = a.tab + r;
    constexpr int DFF_MAX = DFF_E > DFF_P ? DFF_E : DFF_P;
    float* sm = smem;
    float* persist = smem + trunk_scratch_floats(TF(L), DFF_MAX);
    TrunkCtx c{R, a.pfeats[r], a.pmask[r], a.pmstride[r], (int)blockIdx.x, a.S, a.T, a.F, a.save, a.scale, a.rscale};
    int stamp_i = 0;
    if (a.tdbg && blockIdx.x == 0 && blockIdx.y == CF_STAMP_Y && threadIdx.x == 0) a.tdbg[stamp_i++] = __builtin_amdgcn_s_memtime();
#define CF_NEXT_PHASE                                                                                     \
    __syncthreads();                                                                                      \
    if (a.tdbg && blockIdx.x == 0 && blockIdx.y == CF_STAMP_Y && threadIdx.x == 0) a.tdbg[stamp_i++] = __builtin_amdgcn_s_memtime(); \
    launder(c, sm, persist)
    {
        TrunkCtx cs{R, a.cfeats[r], a.cmask[r], a.cmstride[r], (int)blockIdx.x, a.S, a.T, a.F, a.save, a.scale, a.rscale};
        trunk_stage_p(cs, persist);
        launder(c, sm, persist);
    }
    // ---------------------------------------------------------------- Embedding layer: one row (the promoter's centre bin)
    trunk_e_front(c, sm);
    CF_NEXT_PHASE;
    trunk_attc1_e<false>(c, sm);
    CF_NEXT_PHASE;
    trunk_e_post_fwd<DFF_E>(c, sm);
    // ---------------------------------------------------------------- Pairwise stack: S rows (the gene's pairs)
    c.feats = a.cfeats[r];
    c.mask = a.cmask[r];
    c.mstride = a.cmstride[r];
    CF_NEXT_PHASE;
    trunk_qchain_p0(c, sm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the feature strips of trunk_stage_p: every wave's share, then the barrier
    CF_NEXT_PHASE;
    trunk_attc2_p<false, true>(c, 0, sm, persist);       // features, masks, Wlp are in LDS (trunk_stage_p)
    CF_NEXT_PHASE;
    trunk_post_p<DFF_P>(c, 0, PL, sm);
#pragma unroll
    for (int l = 1; l < PL; ++l) {
        CF_NEXT_PHASE;
        trunk_attc2_p<false, true>(c, l, sm, persist);   // later layers: they are in LDS
        CF_NEXT_PHASE;
        trunk_post_p<DFF_P>(c, l, PL, sm);
    }
    CF_NEXT_PHASE;
}

// ---- backward phases
__device__ __forceinline__ void trunk_post_bwd_args(PostBwdArgs& pb, const CentreLayerDev* P) {
    pb.xh2[0] = ldc(&P->xh2);
    pb.rs2[0] = ldc(&P->rs2);
    pb.g2[0] = ldc(&P->g2);
    pb.hdn[0] = ldc(&P->hdn);
    pb.w2[0] = ldc(&P->w2);
    pb.w1[0] = ldc(&P->w1);
    pb.xh1[0] = ldc(&P->xh1);
    pb.rs1[0] = ldc(&P->rs1);
    pb.g1[0] = ldc(&P->g1);
    pb.wo[0] = ldc(&P->wo);
    pb.wv[0] = ldc(&P->wv);
    pb.dt2[0] = ldc(&P->dt2);
    pb.dpre1[0] = ldc(&P->dpre1);
    pb.dt1[0] = ldc(&P->dt1);
    pb.da[0] = ldc(&P->da);
    pb.dxbar[0] = ldc(&P->dxbar);
    pb.partial[0] = ldc(&P->partial);
}
#define CF_BWD_TILES(DFF)                                                                                       \
    constexpr int WW_ = (DFF) > 128 ? (DFF) : 128;                                                              \
    float (*wide)[WW_ + 4] = reinterpret_cast<float (*)[WW_ + 4]>(wide_raw)
template <int DFF>
CF_PHASE void trunk_post_bwd_p(TrunkCtx c, int l, int n_layers, float* smem) {
    const TrunkResDev* R = c.R;
    const CentreLayerDev* P = &R->P[l];
    const bool last = l + 1 == n_layers;
    const int S = c.S, row0 = c.g * S, NB = row0 + S;
    CF_CHAIN_TILES(smem);
    CF_BWD_TILES(DFF);
    PostBwdArgs pb;
    trunk_post_bwd_args(pb, P);
    pb.dout[0] = last ? (const float*)TF(drx0) : (const float*)ldc(&R->P[last ? l : l + 1].dx);
    pb.dmap = last ? RowMap{S, c.T, 1, 1} : identity_map();
    pb.N = NB;
    post_bwd_body<true, 128, DFF, kAT / 64>(pb, 0, row0, NB, c.g, xs, as_, ts, wide);      // one partial row per gene
}
CF_PHASE void trunk_qchain_bwd(TrunkCtx c, const CentreLayerDev* P, int row0, int NB, float* smem) {
    float (*ds)[256 + 4] = reinterpret_cast<float (*)[256 + 4]>(smem);
    float (*qs)[kD + 4] = reinterpret_cast<float (*)[kD + 4]>(smem + kTile * (256 + 4));
    QBwdArgs qb;
    qb.dqt[0] = ldc(&P->dqt);
    qb.dres[0] = ldc(&P->dt1);
    qb.wk[0] = ldc(&P->wk_t);      // NT product in the backward: tiled copy
    qb.wq[0] = ldc(&P->wq);
    qb.dq[0] = ldc(&P->dq);
    qb.dx[0] = ldc(&P->dx);
    qb.N = NB;
    (void)c;
    qchain_bwd_body<kAT / 64>(qb, 0, row0, ds, qs);
}
// The join of the gradient streams at the promoter embedding and lin_proj_p backward for ONE gene (k_join_dgrad's arithmetic):
//   dxp0[g] = sum_s dxP0[g S + s]   (slot 0 first);   edout[g] = dxp0[g] . W + (dX0[g, token 0] + dhin[g, r])
// thread = (column, quarter of K = 128): fmaf chains in the order of the matrix-core version, quarters summed as (0 + 1) + (2 + 3)
CF_PHASE void trunk_join(TrunkCtx c, const float* dhin, int r, int n_res, float* smem) {
    const TrunkResDev* R = c.R;
    const int g = c.g, S = c.S, tid = threadIdx.x;
    float* xk = smem;                  // [128]  dxp0 of the gene
    float* red = smem + kD;            // [4][128]
    if (tid < kD) {
        const float* dxp = ldc(&R->P[0].dx) + (size_t)g * S * kD + tid;
        float s = 0.f;
        for (int i = 0; i < S; ++i) s += ldg(dxp + (size_t)i * kD);
        xk[tid] = s;
        stg(TF(dxp0) + (size_t)g * kD + tid, s);
    }
    __syncthreads();
    {
        const int col = tid & (kD - 1), kw = (tid >> 7) * 32;
        const float* w = TF(lin_p) + (size_t)kw * kD + col;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = k * 16 + q * 4 + i;
                    acc = fmaf(xk[kw + kk], ldg(w + (size_t)kk * kD), acc);
                }
        red[(tid >> 7) * kD + col] = acc;
    }
    __syncthreads();
    if (tid < kD) {
        const float v = (red[tid] + red[kD + tid]) + (red[2 * kD + tid] + red[3 * kD + tid]);
        const float res = ldg(TF(drx0) + (size_t)g * c.T * kD + tid) + ldg(dhin + (size_t)g * (n_res * kD) + r * kD + tid);
        stg(TF(edout) + (size_t)g * kD + tid, v + res);
    }
}
// The 7-mark projection partials of the gene (k_wgrad_lp's arithmetic, same order): dW[e][f] = sum over the job's segments and
// rows of A[m][e] B[m][f].  The <= 80 operand rows of both jobs (this workgroup wrote them a few phases ago) are staged in LDS
// in ONE round trip -- thread-private loops over 64 rows of dependent global loads took 8 us here --, then thread (job, e, half
// of the marks) walks them in the table's order.
constexpr int kLpRowsMax = 80;
CF_PHASE void trunk_lp(const LpJob* jobs, int r, int g, int batch, float* smem) {
    const int tid = threadIdx.x;
    const LpJob* J = jobs + 2 * r;      // [0] Embedding (<= 3 segments), [1] Pairwise (<= 2 kMaxPairLayers)
    float* As = smem;                   // [rows][128]
    float* Bs = smem + kLpRowsMax * kD; // [rows][8]
    (void)batch;
    constexpr int NSEG = 3 + 2 * kMaxPairLayers;
    const int ns0 = ldc(&J[0].nseg), ns1 = ldc(&J[1].nseg);
    // every segment holds <= 16 rows of 128 floats = one float4 per thread: all segments are requested before anything is put
    float4 va[NSEG], vb[NSEG];
    int rpg[NSEG];
#pragma unroll
    for (int u = 0; u < NSEG; ++u) {
        const int jj = u < 3 ? 0 : 1, s = u < 3 ? u : u - 3;
        const bool on = s < (jj ? ns1 : ns0);
        const WgSeg* sp = &J[jj].seg[on ? s : 0];
        const float* A = ldc(&sp->A);
        const float* Bp = ldc(&sp->B);
        const int lda = ldc(&sp->lda), ldb = ldc(&sp->ldb), n = ldc(&sp->rows_per_gene);
        rpg[u] = on ? n : 0;
        const size_t m0 = (size_t)g * n;
        va[u] = ldg4(A + (m0 + min(tid >> 5, n - 1)) * lda + (tid & 31) * 4);
        vb[u] = ldg4(Bp + (m0 + min(tid >> 1, n - 1)) * ldb + (tid & 1) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    int base = 0, n0 = 0;
#pragma unroll
    for (int u = 0; u < NSEG; ++u) {
        if (u == 3) n0 = base;
        if ((tid >> 5) < rpg[u]) *reinterpret_cast<float4*>(As + (base + (tid >> 5)) * kD + (tid & 31) * 4) = va[u];
        if ((tid >> 1) < rpg[u]) *reinterpret_cast<float4*>(Bs + (base + (tid >> 1)) * 8 + (tid & 1) * 4) = vb[u];
        base += rpg[u];
    }
    __syncthreads();
    const int job = tid >> 8, t256 = tid & 255, e = t256 & 127, fh = t256 >> 7;
    const int lo = job ? n0 : 0, hi = job ? base : n0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int row = lo; row < hi; ++row) {
        const float av = As[row * kD + e];
        const float4 bv = *reinterpret_cast<const float4*>(Bs + row * 8 + fh * 4);
        acc[0] = fmaf(av, bv.x, acc[0]);
        acc[1] = fmaf(av, bv.y, acc[1]);
        acc[2] = fmaf(av, bv.z, acc[2]);
        acc[3] = fmaf(av, bv.w, acc[3]);
    }
    const int F = ldc(&J[0].F);
    float* const p0 = ldc(&J[0].partial);
    float* const p1 = ldc(&J[1].partial);
    float* out = (job ? p1 : p0) + (size_t)g * (kD * F) + e * F;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (fh * 4 + k < F) stg(out + fh * 4 + k, acc[k]);
}

template <int DFF_E, int DFF_P, int PL>
__global__ __launch_bounds__(kAT) void k_trunk_bwd(TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.y >= a.n_res) {
        // Riders: the trunk's 192 workgroups fill one CU each (141 KB of LDS, 2 x 256 registers per SIMD lane) for ~130 us of
        // latency chains; the other 64 CUs take Regulation weight-gradient tiles meanwhile -- those depend on the Regulation
        // backward only, and neither the trunk nor anything before the next step's Regulation forward reads what their epilogues
        // update.  One tile per WAVE at a time (wgrad_tile_wave: no workgroup barrier, eight tiles in flight per CU).
        const int w = threadIdx.x >> 6;
        const int t = (((int)blockIdx.y - a.n_res) * gridDim.x + blockIdx.x) * (kAT / 64) + w;      // (rows of riders: one tile per wave)
        unsigned long long* stamp = a.tdbg && blockIdx.x == 0 && (int)blockIdx.y == a.n_res && (threadIdx.x & 63) == 0 ? a.tdbg + 64 + 4 * w : nullptr;
        if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
        if (t < a.rd_n) wgrad_tile_wave<true>(a.rd_tiles[t], a.rd_batch, &a.rd_opt, smem + w * kWgWaveLds, stamp);
        if (stamp) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp[2] = __builtin_amdgcn_s_memtime();
        }
        return;
    }
    if (a.rec.cursor && blockIdx.x == 0 && blockIdx.y == 0) record_block(a.rec);      // (the loss is final: the Regulation backward launch has run)
    const int r = a.n_res - 1 - (int)blockIdx.y;
    const TrunkResDev* R = a.tab + r;
    constexpr int DFF_MAX = DFF_E > DFF_P ? DFF_E : DFF_P;
    float* sm = smem;
    float* persist = smem + trunk_scratch_floats(TF(L), DFF_MAX);
    TrunkCtx c{R, a.cfeats[r], a.cmask[r], a.cmstride[r], (int)blockIdx.x, a.S, a.T, a.F, 1, a.scale, a.rscale};
    int stamp_i = 32;
    if (a.tdbg && blockIdx.x == 0 && blockIdx.y == CF_STAMP_Y && threadIdx.x == 0) a.tdbg[stamp_i++] = __builtin_amdgcn_s_memtime();
    const unsigned long long t_wg0 = __builtin_amdgcn_s_memtime();      // (tools/rider_stamps.py: how long a workgroup of each resolution lasts)
    trunk_stage_p(c, persist);
    launder(c, sm, persist);
    // ---------------------------------------------------------------- Pairwise stack, last layer first
    trunk_post_bwd_p<DFF_P>(c, PL - 1, PL, sm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the feature strips of trunk_stage_p
    CF_NEXT_PHASE;
    trunk_attc2_p<true, true>(c, PL - 1, sm, persist);        // features, masks, Wlp are in LDS (trunk_stage_p)
    CF_NEXT_PHASE;
    trunk_qchain_bwd(c, &c.R->P[PL - 1], c.g * c.S, c.g * c.S + c.S, sm);
#pragma unroll
    for (int l = PL - 2; l >= 0; --l) {
        CF_NEXT_PHASE;
        trunk_post_bwd_p<DFF_P>(c, l, PL, sm);
        CF_NEXT_PHASE;
        trunk_attc2_p<true, true>(c, l, sm, persist);         // they are in LDS
        CF_NEXT_PHASE;
        trunk_qchain_bwd(c, &c.R->P[l], c.g * c.S, c.g * c.S + c.S, sm);
    }
    // ---------------------------------------------------------------- join at the promoter embedding, lin_proj_p backward
    CF_NEXT_PHASE;
    trunk_join(c, a.dhin, r, a.n_res, sm);
    // ---------------------------------------------------------------- Embedding layer
    c.feats = a.pfeats[r];
    c.mask = a.pmask[r];
    c.mstride = a.pmstride[r];
    CF_NEXT_PHASE;
    trunk_e_post_bwd<DFF_E>(c, sm);
    CF_NEXT_PHASE;
    trunk_attc1_e<true>(c, sm);
    CF_NEXT_PHASE;
    trunk_e_q_bwd(c, sm);
    // ---------------------------------------------------------------- 7-mark projection partials of this gene (k_wgrad_lp)
    CF_NEXT_PHASE;
    trunk_lp(a.lp_jobs, r, c.g, a.B, sm);
    CF_NEXT_PHASE;
    if (a.tdbg && blockIdx.x == 0 && threadIdx.x == 0) a.tdbg[100 + blockIdx.y] = __builtin_amdgcn_s_memtime() - t_wg0;
}

}  // namespace cf

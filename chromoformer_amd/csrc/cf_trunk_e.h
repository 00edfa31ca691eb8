// cf_trunk_e.h -- the Embedding layer's row-tile chains inside the fused trunk, on the vector ALUs (included by cf_trunk.h).
//
// The Embedding layer of the centre-row stage works on ONE row per (gene, resolution): the promoter's centre bin (net.py:59).  As phases
// of the trunk it used to run the 16-row matrix-core bodies of the stand-alone kernels (qchain_fwd_body, post_fwd_body, post_bwd_body,
// qchain_bwd_body) with one live row of sixteen: every product cost a full tile on the matrix pipes (~1 K cycles of issue per wave and
// product, two waves per SIMD) plus an operand ring start-up and a barrier, 37.7 K cycles forward and 34.3 K backward per workgroup
// (profiles/r03f_trunk_stamps.txt) for 7 + 6 products of 16 K multiply-adds each.  Here a product is what it is -- a 128 x 128 matrix
// times one vector:
//   NT form   y[n] = sum_k W[n][k] x[k]     wave = 16 rows, lane = (row, quarter of every 16-wide k block) of the TILED weight copy: eight
//                                           16-byte loads, each one contiguous KB per wave, 32 FMAs against x in LDS, two shuffles
//   T form    y[k] = sum_n v[n] W[n][k]     thread = (four columns k, sixteenth of the rows): eight 16-byte loads (a wave reads two row
//                                           groups x 512 contiguous bytes), partial sums meet in LDS in a fixed order
// Every load instruction of either form is unit-stride across the wave, every product's weights are requested one product ahead (the loads
// land under the previous product and its barrier), LayerNorm and its backward run redundantly in every wave on a wave-private copy
// of the row (no barrier), and nothing but the rows other phases or the weight-gradient tiles read goes to global memory.
// Same mathematics as the matrix-core bodies (modules.py:28-58, 91-101; net.py:31-59), other summation order: results agree to fp32
// rounding (tests/test_full_size_gpu.py compares the fused trunk with the stand-alone kernels), not bit for bit.
#pragma once

namespace cf {

static_assert(kAT == kMvT, "the one-row products of cf_valu_mv.h are laid out for the trunk's workgroup size");

// LDS map of the Embedding chains (floats from the start of the trunk's scratch; the attention phase between them owns the scratch
// in its own layout, so nothing survives it in LDS: the few rows that cross it go through global memory, as before)
constexpr int kEvX = 0;                 // [128]  x0 / edout
constexpr int kEvB = 128;               // [256]  xbar / dqt
constexpr int kEvA = 384;               // [128]  a / dy1 / da / dq
constexpr int kEvT = 512;               // [128]  t1, t2 (pre-LayerNorm rows) / xhat rows
constexpr int kEvH = 640;               // [256]  hidden row / dpre1
constexpr int kEvX2 = 896;              // [128]  xhat1 (backward)
constexpr int kEvW = 1024;              // [8][128] wave-private copies of a LayerNorm output / input gradient
constexpr int kEvW2 = 2048;             // [8][128] a second set
constexpr int kEvP = 3072;              // [16][256] partial sums of the T-form products
constexpr int kEvEnd = kEvP + 16 * 256;

// ---- forward, in front of the attention:  x0 = Wlp f_c + PE_c;  q = Wq x0;  qt[h] = Wk[h]^T q[h]     (net.py:42-53 at the centre bin, modules.py:38-58)
CF_PHASE void trunk_e_front(TrunkCtx c, float* smem) {
    const TrunkResDev* R = c.R;
    const int g = c.g, tid = threadIdx.x;
    float* xs = smem + kEvX;
    float* qs = smem + kEvA;
    float* part = smem + kEvP;
    NtW<kD, kD> wq;
    nt_request(wq, TF(E.wq_t));
    TrW<kD, kD> wk;
    tr_request(wk, TF(E.wk));
    __builtin_amdgcn_sched_barrier(0);
    if (tid < kD) {      // embed_x0_row's arithmetic, the row kept in LDS
        const int L = TF(L), cb = L / 2, F = c.F;
        const float* f = c.feats + ((size_t)g * L + cb) * F;
        const float* wl = TF(wlp_e) + tid * F;
        float acc = 0.f;
        for (int i = 0; i < F; ++i) acc = fmaf(ldg(f + i), ldg(wl + i), acc);
        const float x = acc + ldg(TF(pe) + (size_t)cb * kD + tid);
        xs[tid] = x;
        stg(TF(ex0) + (size_t)g * kD + tid, x);
        if (tid < 8) stg(TF(featc) + (size_t)g * 8 + tid, tid < F ? ldg(f + tid) : 0.f);
    }
    __syncthreads();
    {
        float q[1];
        nt_dot(wq, xs, q);
        if (nt_writer()) {
            const int n = nt_row<kD, kD>(0);
            qs[n] = q[0];
            stg(TF(E.q) + (size_t)g * kD + n, q[0]);
        }
    }
    __syncthreads();
    tr_partial(wk, qs, part);      // row groups 0..7: head 0 (rows 0..63), 8..15: head 1
    __syncthreads();
    if (tid < 2 * kD) {
        const int h = tid >> 7, e = tid & (kD - 1);
        stg(TF(E.qt) + (size_t)g * 2 * kD + tid, tr_sum<kD>(part, h * 8, 8, e));
    }
}

// ---- forward, behind the attention:  a[h] = Wv[h] xbar[h];  y1 = LN(x0 + Wo a + bo);  out = LN(y1 + W2 relu(W1 y1 + b1) + b2);  xp0 = lin_proj_p out
//      (modules.py:28-30, 100-101; net.py:118)
template <int DFF>
CF_PHASE void trunk_e_post_fwd(TrunkCtx c, float* smem) {
    const TrunkResDev* R = c.R;
    const int g = c.g, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const bool save = c.save != 0;
    float* xs = smem + kEvX;
    float* xb = smem + kEvB;
    float* as_ = smem + kEvA;
    float* ts = smem + kEvT;
    float* hs = smem + kEvH;
    float* y1w = smem + kEvW + w * kD;       // this wave's copy of y1
    float* ow = smem + kEvW2 + w * kD;       // ... of the layer output
    static_assert(DFF == 128 || DFF == 256, "hidden width of the Embedding layer");
    NtW<kD, kD> wv, wo;
    nt_request(wv, TF(E.wv_t));
    float rx = 0.f, rb = 0.f;
    if (tid < kD) rx = ldg(TF(ex0) + (size_t)g * kD + tid);
    if (tid < 2 * kD) rb = ldg(TF(E.xbar) + (size_t)g * 2 * kD + tid);
    nt_request(wo, TF(E.wo_t));
    // LayerNorm parameters of the lane's two elements and the biases of the row this thread finishes
    const int n = nt_row<kD, kD>(0);
    const float g1a = ldg(TF(E.g1) + lane), g1b = ldg(TF(E.g1) + lane + 64), b1a = ldg(TF(E.be1) + lane), b1b = ldg(TF(E.be1) + lane + 64);
    const float g2a = ldg(TF(E.g2) + lane), g2b = ldg(TF(E.g2) + lane + 64), b2a = ldg(TF(E.be2) + lane), b2b = ldg(TF(E.be2) + lane + 64);
    const float bo = ldg(TF(E.bo) + n), b2 = ldg(TF(E.b2) + n);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < kD) xs[tid] = rx;
    if (tid < 2 * kD) xb[tid] = rb;
    __syncthreads();
    {   // a = Wv xbar (row n belongs to head n >> 6: one head per wave)
        float a[1];
        nt_dot(wv, xb + (n >> 6) * kD, a);
        if (nt_writer()) {
            as_[n] = a[0];
            if (save) stg(TF(E.a) + (size_t)g * kD + n, a[0]);
        }
    }
    NtW<DFF, kD> w1;
    nt_request(w1, TF(E.w1_t));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    {   // t1 = x0 + Wo a + bo
        float t[1];
        nt_dot(wo, as_, t);
        if (nt_writer()) ts[n] = t[0] + bo + xs[n];
    }
    NtW<kD, DFF> w2;
    nt_request(w2, TF(E.w2_t));
    float b1[NtW<DFF, kD>::TW];
#pragma unroll
    for (int t = 0; t < NtW<DFF, kD>::TW; ++t) b1[t] = ldg(TF(E.b1) + nt_row<DFF, kD>(t));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    {   // y1 = LN(t1): every wave, on its own copy
        const LnRow o = ln_row_fwd(ts[lane], ts[lane + 64], g1a, g1b, b1a, b1b);
        y1w[lane] = o.y0;
        y1w[lane + 64] = o.y1;
        if (w == 0 && save) {
            stg(TF(E.xh1) + (size_t)g * kD + lane, o.x0);
            stg(TF(E.xh1) + (size_t)g * kD + lane + 64, o.x1);
            stg(TF(E.y1) + (size_t)g * kD + lane, o.y0);
            stg(TF(E.y1) + (size_t)g * kD + lane + 64, o.y1);
            if (lane == 0) stg(TF(E.rs1) + g, o.rstd);
        }
    }
    __builtin_amdgcn_wave_barrier();
    {   // hdn = relu(W1 y1 + b1)
        float v[NtW<DFF, kD>::TW];
        nt_dot(w1, y1w, v);
        if (nt_writer())
#pragma unroll
            for (int t = 0; t < NtW<DFF, kD>::TW; ++t) {
                const int j = nt_row<DFF, kD>(t);
                const float hv = fmaxf(v[t] + b1[t], 0.f);
                hs[j] = hv;
                if (save) stg(TF(E.hdn) + (size_t)g * DFF + j, hv);
            }
    }
    NtW<kD, kD> wl;
    nt_request(wl, TF(lin_p_t));
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();      // (also: every wave is past its read of ts)
    {   // t2 = y1 + W2 hdn + b2
        float t[1];
        nt_dot(w2, hs, t);
        if (nt_writer()) ts[n] = t[0] + b2 + y1w[n];
    }
    __syncthreads();
    {   // out = LN(t2) -> token 0 of the gene's Regulation input
        const LnRow o = ln_row_fwd(ts[lane], ts[lane + 64], g2a, g2b, b2a, b2b);
        ow[lane] = o.y0;
        ow[lane + 64] = o.y1;
        if (w == 0) {
            if (save) {
                stg(TF(E.xh2) + (size_t)g * kD + lane, o.x0);
                stg(TF(E.xh2) + (size_t)g * kD + lane + 64, o.x1);
                if (lane == 0) stg(TF(E.rs2) + g, o.rstd);
            }
            float* rx0 = TF(rx0) + (size_t)g * c.T * kD;
            stg(rx0 + lane, o.y0);
            stg(rx0 + lane + 64, o.y1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    {   // xp0 = lin_proj_p out
        float v[1];
        nt_dot(wl, ow, v);
        if (nt_writer()) stg(TF(xp0) + (size_t)g * kD + n, v[0]);
    }
}

// ---- backward, in front of the attention backward: edout -> dt2, dpre1, dt1, da, dxbar and the gene's row of bias / LayerNorm sums
//      partial row = [dg2 | db2' | dbias2 | dbias1 (DFF) | dg1 | db1' | dbo]  (post_bwd_body's layout; one row: the sums are the row itself)
template <int DFF>
CF_PHASE void trunk_e_post_bwd(TrunkCtx c, float* smem) {
    const TrunkResDev* R = c.R;
    const int g = c.g, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    float* dys = smem + kEvA;                 // dy1
    float* das = smem + kEvA;                 // da (dy1 is dead by then)
    float* hs = smem + kEvH;                  // dpre1
    float* dt2w = smem + kEvW + w * kD;
    float* dt1w = smem + kEvW2 + w * kD;
    float* part = smem + kEvP;
    float* prow = TF(E.partial) + (size_t)g * post_partial_width(DFF);
    TrW<kD, DFF> w2;
    tr_request(w2, TF(E.w2));                 // dhdn[j] = sum_n dt2[n] W2[n][j]
    const float d0 = ldg(TF(edout) + (size_t)g * kD + lane), d1 = ldg(TF(edout) + (size_t)g * kD + lane + 64);
    const float x20 = ldg(TF(E.xh2) + (size_t)g * kD + lane), x21 = ldg(TF(E.xh2) + (size_t)g * kD + lane + 64);
    const float x10 = ldg(TF(E.xh1) + (size_t)g * kD + lane), x11 = ldg(TF(E.xh1) + (size_t)g * kD + lane + 64);
    const float rs2 = ldg(TF(E.rs2) + g), rs1 = ldg(TF(E.rs1) + g);
    const float g2a = ldg(TF(E.g2) + lane), g2b = ldg(TF(E.g2) + lane + 64), g1a = ldg(TF(E.g1) + lane), g1b = ldg(TF(E.g1) + lane + 64);
    float hv = 0.f;
    if (tid < DFF) hv = ldg(TF(E.hdn) + (size_t)g * DFF + tid);
    TrW<DFF, kD> w1;
    tr_request(w1, TF(E.w1));                 // dy1[k] = dt2[k] + sum_j dpre1[j] W1[j][k]
    __builtin_amdgcn_sched_barrier(0);
    float t20, t21;
    ln_row_bwd(d0, d1, g2a, g2b, x20, x21, rs2, t20, t21);      // dt2, every wave on its own copy
    dt2w[lane] = t20;
    dt2w[lane + 64] = t21;
    if (w == 0) {
        stg(TF(E.dt2) + (size_t)g * kD + lane, t20);
        stg(TF(E.dt2) + (size_t)g * kD + lane + 64, t21);
        stg(prow + lane, d0 * x20);
        stg(prow + lane + 64, d1 * x21);
        stg(prow + 128 + lane, d0);
        stg(prow + 128 + lane + 64, d1);
        stg(prow + 256 + lane, t20);
        stg(prow + 256 + lane + 64, t21);
    }
    __builtin_amdgcn_wave_barrier();
    tr_partial(w2, dt2w, part);
    TrW<kD, kD> wo;
    tr_request(wo, TF(E.wo));                 // da[k] = sum_n dt1[n] Wo[n][k]
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (tid < DFF) {
        const float s = tr_sum<DFF>(part, 0, TrW<kD, DFF>::NG, tid);
        const float dp = hv > 0.f ? s : 0.f;
        hs[tid] = dp;
        stg(TF(E.dpre1) + (size_t)g * DFF + tid, dp);
        stg(prow + 384 + tid, dp);
    }
    __syncthreads();
    tr_partial(w1, hs, part);
    __syncthreads();
    if (tid < kD) dys[tid] = dt2w[tid] + tr_sum<kD>(part, 0, TrW<DFF, kD>::NG, tid);
    TrW<kD, kD> wv;
    tr_request(wv, TF(E.wv));                 // dxbar[h][e] = sum_d da[h 64 + d] Wv[h 64 + d][e]
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    {
        const float y0 = dys[lane], y1 = dys[lane + 64];
        float t10, t11;
        ln_row_bwd(y0, y1, g1a, g1b, x10, x11, rs1, t10, t11);      // dt1
        dt1w[lane] = t10;
        dt1w[lane + 64] = t11;
        if (w == 0) {
            stg(TF(E.dt1) + (size_t)g * kD + lane, t10);
            stg(TF(E.dt1) + (size_t)g * kD + lane + 64, t11);
            stg(prow + 384 + DFF + lane, y0 * x10);
            stg(prow + 384 + DFF + lane + 64, y1 * x11);
            stg(prow + 512 + DFF + lane, y0);
            stg(prow + 512 + DFF + lane + 64, y1);
            stg(prow + 640 + DFF + lane, t10);
            stg(prow + 640 + DFF + lane + 64, t11);
        }
    }
    __builtin_amdgcn_wave_barrier();
    tr_partial(wo, dt1w, part);      // (the last readers of `part` are behind the barrier in front of the LayerNorm backward)
    __syncthreads();
    if (tid < kD) {
        const float s = tr_sum<kD>(part, 0, TrW<kD, kD>::NG, tid);
        das[tid] = s;
        stg(TF(E.da) + (size_t)g * kD + tid, s);
    }
    __syncthreads();
    tr_partial(wv, das, part);
    __syncthreads();
    if (tid < 2 * kD) {
        const int h = tid >> 7, e = tid & (kD - 1);
        stg(TF(E.dxbar) + (size_t)g * 2 * kD + tid, tr_sum<kD>(part, h * 8, 8, e));
    }
}

// ---- backward, behind the attention backward:  dq[n] = sum_e dqt[h(n)][e] Wk[n][e];  dx = dt1 + Wq^T dq     (qchain_bwd_body)
CF_PHASE void trunk_e_q_bwd(TrunkCtx c, float* smem) {
    const TrunkResDev* R = c.R;
    const int g = c.g, tid = threadIdx.x;
    float* db = smem + kEvB;       // dqt [2][128]
    float* dqs = smem + kEvA;
    float* part = smem + kEvP;
    NtW<kD, kD> wk;
    nt_request(wk, TF(E.wk_t));
    float rb = 0.f, rr = 0.f;
    if (tid < 2 * kD) rb = ldg(TF(E.dqt) + (size_t)g * 2 * kD + tid);
    if (tid < kD) rr = ldg(TF(E.dt1) + (size_t)g * kD + tid);
    TrW<kD, kD> wq;
    tr_request(wq, TF(E.wq));
    __builtin_amdgcn_sched_barrier(0);
    if (tid < 2 * kD) db[tid] = rb;
    __syncthreads();
    {
        const int n = nt_row<kD, kD>(0);
        float q[1];
        nt_dot(wk, db + (n >> 6) * kD, q);
        if (nt_writer()) {
            dqs[n] = q[0];
            stg(TF(E.dq) + (size_t)g * kD + n, q[0]);
        }
    }
    __syncthreads();
    tr_partial(wq, dqs, part);
    __syncthreads();
    if (tid < kD) stg(TF(E.dx) + (size_t)g * kD + tid, rr + tr_sum<kD>(part, 0, TrW<kD, kD>::NG, tid));
}

}  // namespace cf

// cf_reg_fused.h -- the Regulation stack as two launches (included by cf_kernels.h).
//
// One workgroup per (gene, resolution) walks ALL layers.  The gene's T <= 11 tokens are one
// MFMA row tile (rows >= T are dead weight, which is free: 192 workgroups on 256 CUs); the
// token embeddings stay in LDS from layer to layer, only what the backward needs is written
// out.  Per layer: q|k|v|gate projection (4 column chunks) -> attention in LDS ->
// out-proj + LN -> FFN + LN   (modules.py:28-88, 100-101, 121-124; net.py:148-153).
#pragma once

namespace cf {

struct RegLayerDev {
    const float *watt, *gamma, *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2;
    const float *watt_t, *wo_t, *w1_t, *w2_t;      // tiled copies (forward products)
    const float *watt_tt, *wo_tt, *w1_tt, *w2_tt;  // tiled copies of the transposes (backward products of cf_reg8.h)
    float *xin, *qkvg, *p, *a, *xh1, *rs1, *y1, *hdn, *xh2, *rs2, *xout;
    float *dxout, *dt2, *dpre1, *dt1, *da, *dqkvg, *dxin, *partial, *dgam;
    float *hq, *dy1;                               // cf_reg8.h: per-(gene, head) attention operands; d(LayerNorm 1 output)
};
struct RegArgs {
    const RegLayerDev* tab;          // [n_res][n_layers]
    int n_layers, T, B, n_res, xcd_map;
    const uint8_t* mask[kMaxRes];    // [B,T,T]
    const float* freq;               // [B,T,T]
    int save;
    unsigned long long* tdbg;        // optional: shader-clock stamps of workgroup (0,0), 16 per layer
    HeadRide head;                   // forward (cf_reg8.h): the gene's prediction head at the tail of its last workgroup (cf_head_ride.h)
    float* team_slots;               // cf_regq.h: exchange slots of the four-workgroup teams, [units][2][4][16 x 128]
    int* team_cnt;                   //            arrival / departure counters, [units][32]
    int row0_last;                   // cf_reg8.h: only token 0 of the LAST layer's output is consumed (net.py:375): that layer computes row 0 only
};
#define CF_STAMP(slot)                                                                                 \
    do {                                                                                               \
        if (a.tdbg && g == 0 && r == 0 && threadIdx.x == 0)                                              \
            a.tdbg[l * 16 + (slot)] = __builtin_amdgcn_s_memtime();                                    \
    } while (0)

constexpr int kQkLd = kRW + 4;
__host__ __device__ constexpr size_t reg_fwd_smem(int T) {
    return (size_t)(2 * kTile * (kD + 4) + 2 * kTile * (kRDm + 4) + T * kQkLd + kRH * T * T + 2 * T * T + 16) * sizeof(float);
}
__host__ __device__ constexpr size_t reg_bwd_smem(int T) {
    return (size_t)(3 * kTile * (kD + 4) + kTile * (kRDm + 4) + T * kQkLd + kTile * kQkLd + 2 * kRH * T * T + kRH * T + 2 * T * T + 16) *
           sizeof(float);
}

// Small per-head products of the attention on the matrix cores: D[16 x 16] = A[16 x 4*KS] . B[4*KS x 16],
// operands fetched element-wise through branch-free functors: indices are clamped into valid data (rows / columns
// >= T only produce results that are discarded) and reduction indices >= T read a zero word on the A side.
// Lane (r, q): A(row = r, k = 4*ks + q), B(k = 4*ks + q, col = r); result reg ii = D[4q + ii][r].
template <int KS, class FA, class FB>
__device__ __forceinline__ f32x4 mfma_small(FA fa, FB fb) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    float av[KS], bv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {      // all operand reads first (branch-free functors), then the MFMA chain
        av[ks] = fa(r, 4 * ks + q);
        bv[ks] = fb(4 * ks + q, r);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = mfma4(av[ks], bv[ks], acc);
    return acc;
}
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// TC: compile-time token count (0 = take a.T at run time); with TC > 0 every loop over tokens unrolls.
template <int DFF, int TC>
__global__ __launch_bounds__(256) void k_reg_fwd(RegArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own 4 MB L2.  XCD x
    // takes the contiguous slice [x * per, (x + 1) * per) of the (resolution-major) sequence list, so six XCDs stream
    // the weights of one resolution (5.5 MB) and two of them those of two, instead of all eight streaming all three.
    const int per = gridDim.x >> 3, v = a.xcd_map ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : (int)blockIdx.x;
    if (v >= a.B * a.n_res) return;
    const int g = v % a.B, r = v / a.B, T = TC > 0 ? TC : a.T, TT = T * T, row0 = g * T, tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    constexpr int LD = kD + 4, LW = kRDm + 4;
    float* xs = smem;                    // [16][LD]   layer input / output
    float* ts = xs + kTile * LD;         // [16][LD]
    float* as_ = ts + kTile * LD;        // [16][LW]   gated attention output
    float* hs = as_ + kTile * LW;        // [16][LW]   FFN hidden
    float* qk = hs + kTile * LW;         // [T][kQkLd] q | k | v | gate of the gene
    float* p_s = qk + T * kQkLd;         // [8][T][T]
    float* fq_s = p_s + kRH * TT;        // [T][T]     interaction frequencies of the gene
    float* mk_s = fq_s + TT;             // [T][T]     interaction mask (1 = masked)
    float* gam_s = mk_s + TT;            // [8] + one zero word read by the padded reduction indices
    const int ZI = (int)(gam_s + 8 - p_s);
    if (tid == 0) gam_s[8] = 0.f;
    const RegLayerDev* tab = a.tab + (size_t)r * a.n_layers;
    const float scale = sqrtf((float)kRDh);
    {
        const float* x0 = tab[0].xin + (size_t)row0 * kD;
        for (int i = tid; i < kTile * (kD / 4); i += 256) {
            const int row = i >> 5, c4 = i & 31;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < T) v = ldg4(x0 + row * kD + c4 * 4);
            *reinterpret_cast<float4*>(xs + row * LD + c4 * 4) = v;
        }
        for (int i = tid; i < TT; i += 256) {
            fq_s[i] = ldg(a.freq + (size_t)g * TT + i);
            mk_s[i] = *(const CF_GLOBAL uint8_t*)(a.mask[r] + (size_t)g * TT + i) ? 1.f : 0.f;
        }
    }
    for (int i = tid; i < (kTile - T) * LW; i += 256) as_[T * LW + i] = 0.f;      // dead rows of the attention output
    // q|k|v|gate projection: one operand ring for the whole [16 x 1024] product.  Step i = (column
    // chunk i/4 of 64 columns, K-chunk i%4 of 32); the ring runs 3 steps ahead and across layers.
    float4 qring[kRing][2][4];
    auto qfetch = [&](int slot, int step, const float* watt_t) {
        const float* base = watt_t + (size_t)(w * 256 + (step >> 2) * 64) * kD + (step & 3) * 512 + lane * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t) qring[slot][k][t] = ldg4(base + (size_t)t * (kD * 16) + k * 256);
    };
#pragma unroll
    for (int i = 0; i < kRing - 1; ++i) qfetch(i, i, tab[0].watt_t);
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l) {
        const RegLayerDev P = tab[l];      // by value: the pointers live in SGPRs (no reload after every store)
        CF_STAMP(0);
        // per-lane copies of the small vectors of this layer (their latency hides behind the projection)
        const float bo0 = ldg(P.bo + w * 32 + lr), bo1 = ldg(P.bo + w * 32 + 16 + lr);
        const float b20 = ldg(P.b2 + w * 32 + lr), b21 = ldg(P.b2 + w * 32 + 16 + lr);
        float b1v[DFF / 64];
#pragma unroll
        for (int t = 0; t < DFF / 64; ++t) b1v[t] = ldg(P.b1 + w * (DFF / 4) + t * 16 + lr);
        const LnParams ln1 = ln_params_load(P.g1, P.be1), ln2 = ln_params_load(P.g2, P.be2);
        float gam = 0.f;
        if (lane < kRH) gam = ldg(P.gamma + lane);
        {
            f32x4 acc[4];
            const float* ap = xs + lr * LD + lq * 4;
            float4 av8[8];      // the A operand (this layer's input tile) is the same for all four column chunks: read it once
#pragma unroll
            for (int k = 0; k < 8; ++k) av8[k] = *reinterpret_cast<const float4*>(ap + k * 16);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + kRing - 1 < 16) qfetch((i + kRing - 1) % kRing, i + kRing - 1, P.watt_t);
                __builtin_amdgcn_sched_barrier(0);
                if ((i & 3) == 0) zero_acc(acc);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float4 av = av8[(i & 3) * 2 + k];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 b = qring[i % kRing][k][t];
                        acc[t] = mfma4(av.x, b.x, acc[t]);
                        acc[t] = mfma4(av.y, b.y, acc[t]);
                        acc[t] = mfma4(av.z, b.z, acc[t]);
                        acc[t] = mfma4(av.w, b.w, acc[t]);
                    }
                }
                if ((i & 3) == 3) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) {
                            const int row = lq * 4 + ii, col = w * 256 + (i >> 2) * 64 + col_nt(t, lr);
                            if (row < T) qk[row * kQkLd + col] = acc[t][ii];
                        }
                }
            }
        }
        FragNT<2, kRDm / 16> fo;       // out-projection weights: in flight during the attention
        frag_load_nt(fo, P.wo_t + (size_t)(w * 32) * kRDm, kRDm);
        CF_STAMP(1);
        __syncthreads();
        CF_STAMP(2);
        if (a.save) {                   // coalesced copy of the projection for the backward pass
            float* qg = P.qkvg + (size_t)row0 * kRW;
            for (int i = tid; i < T * (kRW / 4); i += 256) {
                const int row = i >> 8, c4 = i & 255;
                stg4(qg + (size_t)row * kRW + c4 * 4, *reinterpret_cast<const float4*>(qk + row * kQkLd + c4 * 4));
            }
        }
        // ---- attention (modules.py:58-81) on the matrix cores: wave w owns heads 2w, 2w+1
        constexpr int KT = TC > 0 ? (TC + 3) / 4 : (kRMaxT + 3) / 4;     // k-steps covering the tokens
        if (tid < kRH) gam_s[tid] = gam;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * w + hh;
            const f32x4 sc = mfma_small<kRDh / 4>(
                [&](int i, int d) { return qk[min(i, T - 1) * kQkLd + h * kRDh + d]; },
                [&](int d, int j) { return qk[min(j, T - 1) * kQkLd + kRDm + h * kRDh + d]; });
            const float gm = ldg(P.gamma + h);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = lq * 4 + ii, j = lr;
                if (i < T && j < T) {
                    float v = sc[ii] / scale + gm * fq_s[i * T + j];
                    if (mk_s[i * T + j] != 0.f) v = kMaskFill;
                    p_s[(h * T + i) * T + j] = v;
                }
            }
        }
        CF_STAMP(3);
        __syncthreads();
        for (int row = tid; row < kRH * T; row += 256) {
            float* pr = p_s + row * T;
            float m = -INFINITY;
            _Pragma("unroll") for (int j = 0; j < T; ++j) m = fmaxf(m, pr[j]);
            float e[TC > 0 ? TC : kRMaxT];
            float z = 0.f;
            _Pragma("unroll") for (int j = 0; j < (TC > 0 ? TC : kRMaxT); ++j) {
                e[j] = j < T ? expf(pr[j] - m) : 0.f;
                z += e[j];
            }
            const float rz = 1.0f / z;
            _Pragma("unroll") for (int j = 0; j < (TC > 0 ? TC : kRMaxT); ++j)
                if (j < T) pr[j] = e[j] * rz;
        }
        CF_STAMP(4);
        __syncthreads();
        if (a.save) {
            float* pg = P.p + (size_t)g * kRH * TT;
            for (int i = tid; i < kRH * TT; i += 256) stg(pg + i, p_s[i]);
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int h = 2 * w + hh, col = h * kRDh + nt * 16 + lr;
                const f32x4 o = mfma_small<KT>(
                    [&](int i, int j) { return p_s[j < T ? (h * T + min(i, T - 1)) * T + j : ZI]; },
                    [&](int j, int n) { return qk[min(j, T - 1) * kQkLd + 2 * kRDm + h * kRDh + nt * 16 + n]; });
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = lq * 4 + ii;
                    if (i < T) {
                        const float gt = qk[i * kQkLd + 3 * kRDm + col];
                        const float v = o[ii] * fast_sigmoid(gt);
                        if (a.save) stg(P.a + (size_t)(row0 + i) * kRDm + col, v);
                        as_[i * LW + col] = v;
                    }
                }
            }
        CF_STAMP(5);
        __syncthreads();
        CF_STAMP(6);
        // ---- out-projection + residual + LN
        {
            f32x4 acc[2];
            zero_acc(acc);
            frag_mma_nt(fo, as_, LW, acc);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = lq * 4 + i, col = w * 32 + col_nt(t, lr);
                    ts[row * LD + col] = acc[t][i] + (t ? bo1 : bo0) + xs[row * LD + col];
                }
        }
        constexpr int NT1 = DFF / 64;
        FragNT<NT1, 8> f1;
        frag_load_nt(f1, P.w1_t + (size_t)(w * (DFF / 4)) * kD, kD);
        CF_STAMP(7);
        __syncthreads();
        ln_fwd_tile16(ts, LD, ln1, row0, T, a.save ? P.xh1 : nullptr, P.rs1, a.save ? P.y1 : nullptr);
        CF_STAMP(8);
        __syncthreads();
        {
            f32x4 acc[NT1];
            zero_acc(acc);
            frag_mma_nt(f1, ts, LD, acc);
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = lq * 4 + i, col = w * (DFF / 4) + col_nt(t, lr);
                    const float v = fmaxf(acc[t][i] + b1v[t], 0.f);
                    hs[row * LW + col] = v;
                    if (a.save && row < T) stg(P.hdn + (size_t)(row0 + row) * DFF + col, v);
                }
        }
        FragNT<2, DFF / 16> f2;
        frag_load_nt(f2, P.w2_t + (size_t)(w * 32) * DFF, DFF);
        CF_STAMP(9);
        __syncthreads();
        {
            f32x4 acc[2];
            zero_acc(acc);
            frag_mma_nt(f2, hs, LW, acc);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = lq * 4 + i, col = w * 32 + col_nt(t, lr);
                    xs[row * LD + col] = acc[t][i] + (t ? b21 : b20) + ts[row * LD + col];
                }
        }
        if (l + 1 < a.n_layers) {                 // first operand chunks of the next layer's projection
#pragma unroll
            for (int i = 0; i < kRing - 1; ++i) qfetch(i, i, tab[l + 1].watt_t);
        }
        CF_STAMP(10);
        __syncthreads();
        ln_fwd_tile16(xs, LD, ln2, row0, T, a.save ? P.xh2 : nullptr, P.rs2, P.xout);
        CF_STAMP(11);
        __syncthreads();
    }
}

template <int DFF, int TC>
__global__ __launch_bounds__(256) void k_reg_bwd(RegArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own 4 MB L2.  XCD x
    // takes the contiguous slice [x * per, (x + 1) * per) of the (resolution-major) sequence list, so six XCDs stream
    // the weights of one resolution (5.5 MB) and two of them those of two, instead of all eight streaming all three.
    const int per = gridDim.x >> 3, v = a.xcd_map ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : (int)blockIdx.x;
    if (v >= a.B * a.n_res) return;
    const int g = v % a.B, r = v / a.B, T = TC > 0 ? TC : a.T, TT = T * T, row0 = g * T, tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    constexpr int LD = kD + 4, LW = kRDm + 4, NT2 = DFF / 64, PW = post_partial_width(DFF);
    float* ds = smem;                    // [16][LD]  d(layer output) -> dy1 -> dt1 -> d(layer input)
    float* xh = ds + kTile * LD;         // [16][LD]  xhat2 -> xhat1
    float* t2 = xh + kTile * LD;         // [16][LD]  dt2
    float* wide = t2 + kTile * LD;       // [16][LW]  dpre1 -> da
    float* qk = wide + kTile * LW;       // [T][kQkLd]
    float* dqk = qk + T * kQkLd;         // [16][kQkLd]
    float* p_s = dqk + kTile * kQkLd;    // [8][T][T]
    float* s_s = p_s + kRH * TT;         // [8][T][T]
    float* red_s = s_s + kRH * TT;       // [8*T]
    float* fq_s = red_s + kRH * T;       // [T][T]
    float* mk_s = fq_s + TT;             // [T][T]
    float* zero_s = mk_s + TT;           // one zero word read by the padded reduction indices
    const int ZI = (int)(zero_s - p_s), ZS = (int)(zero_s - s_s);
    if (tid == 0) zero_s[0] = 0.f;
    float* do_s = xh;                    // [T][256] aliases xh|t2 (both dead during the attention backward)
    const RegLayerDev* tab = a.tab + (size_t)r * a.n_layers;
    const float scale = sqrtf((float)kRDh);
    auto load_rows = [&](float* dst, const float* src) {      // [T,128] rows of this gene -> 16-row LDS tile
        for (int i = tid; i < kTile * (kD / 4); i += 256) {
            const int row = i >> 5, c4 = i & 31;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < T) v = ldg4(src + (size_t)(row0 + row) * kD + c4 * 4);
            *reinterpret_cast<float4*>(dst + row * LD + c4 * 4) = v;
        }
    };
    for (int i = tid; i < (kTile - T) * kQkLd; i += 256) dqk[T * kQkLd + i] = 0.f;      // dead rows of the MFMA operand
    for (int i = tid; i < TT; i += 256) {
        fq_s[i] = ldg(a.freq + (size_t)g * TT + i);
        mk_s[i] = *(const CF_GLOBAL uint8_t*)(a.mask[r] + (size_t)g * TT + i) ? 1.f : 0.f;
    }
    load_rows(ds, tab[a.n_layers - 1].dxout);
    for (int l = a.n_layers - 1; l >= 0; --l) {
        const RegLayerDev P = tab[l];
        float* part = P.partial + (size_t)g * PW;
        CF_STAMP(0);
        const float4 lg2a = ldg4(P.g2 + (lane & 15) * 8), lg2b = ldg4(P.g2 + (lane & 15) * 8 + 4);
        const float4 lg1a = ldg4(P.g1 + (lane & 15) * 8), lg1b = ldg4(P.g1 + (lane & 15) * 8 + 4);
        FragNN<NT2, 8> fw2;
        frag_load_nn(fw2, P.w2 + w * (DFF / 4), DFF);
        load_rows(xh, P.xh2);
        __syncthreads();
        colsum_rows(ds, LD, xh, LD, kD, T, part + 0);
        colsum_rows(ds, LD, nullptr, 0, kD, T, part + 128);
        CF_STAMP(1);
        ln_bwd_tile16(ds, t2, LD, xh, LD, lg2a, lg2b, P.rs2, row0, T, P.dt2);      // t2 = dt2
        CF_STAMP(2);
        const int cg = w & 1, kh = w >> 1;      // products with 128 output columns: 2 column groups x 2 K halves
        FragNN<4, DFF / 32> fw1;
        frag_load_nn(fw1, P.w1 + (size_t)(kh * (DFF / 2)) * kD + cg * 64, kD);
        __syncthreads();
        colsum_rows(t2, LD, nullptr, 0, kD, T, part + 256);
        {
            f32x4 acc[NT2];
            zero_acc(acc);
            frag_mma_nn(fw2, t2, LD, acc);
            typedef typename VecN<NT2>::type vec_t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * (DFF / 4) + NT2 * lr;
                float v[NT2];
#pragma unroll
                for (int t = 0; t < NT2; ++t) v[t] = 0.f;
                if (row < T) {
                    const size_t o = (size_t)(row0 + row) * DFF + col;
                    const vec_t hv = LdgN<NT2>::ld(P.hdn + o);
                    const float* hp = reinterpret_cast<const float*>(&hv);
#pragma unroll
                    for (int t = 0; t < NT2; ++t) v[t] = hp[t] > 0.f ? acc[t][i] : 0.f;
                    LdgN<NT2>::st(P.dpre1 + o, *reinterpret_cast<const vec_t*>(v));
                }
                *reinterpret_cast<vec_t*>(wide + row * LW + col) = *reinterpret_cast<const vec_t*>(v);
            }
        }
        CF_STAMP(3);
        load_rows(xh, P.xh1);
        FragNN<4, 8> fwo;
        frag_load_nn(fwo, P.wo + w * (kRDm / 4), kRDm);
        __syncthreads();
        colsum_rows(wide, LW, nullptr, 0, DFF, T, part + 384);
        {
            f32x4 acc[4];
            zero_acc(acc);
            frag_mma_nn(fw1, wide + kh * (DFF / 2), LW, acc);
            float* red = qk;                    // [16][LD] scratch (the q|k|v|g tile is loaded later)
            if (kh == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(red + (lq * 4 + i) * LD + cg * 64 + 4 * lr) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = (lq * 4 + i) * LD + cg * 64 + 4 * lr;
                    const float4 rv = *reinterpret_cast<const float4*>(red + o), tv = *reinterpret_cast<const float4*>(t2 + o);
                    *reinterpret_cast<float4*>(ds + o) = make_float4(acc[0][i] + rv.x + tv.x, acc[1][i] + rv.y + tv.y, acc[2][i] + rv.z + tv.z,
                                                                    acc[3][i] + rv.w + tv.w);
                }
            }
        }
        __syncthreads();
        CF_STAMP(4);
        colsum_rows(ds, LD, xh, LD, kD, T, part + 384 + DFF);
        colsum_rows(ds, LD, nullptr, 0, kD, T, part + 512 + DFF);
        __syncthreads();
        ln_bwd_tile16(ds, ds, LD, xh, LD, lg1a, lg1b, P.rs1, row0, T, P.dt1);       // ds = dt1 (each lane rewrites only what it read)
        CF_STAMP(5);
        // operands of the attention backward: in flight during the out-projection product
        {
            const float* qg = P.qkvg + (size_t)row0 * kRW;
            for (int i = tid; i < T * (kRW / 4); i += 256) {
                const int row = i >> 8, c4 = i & 255;
                *reinterpret_cast<float4*>(qk + row * kQkLd + c4 * 4) = ldg4(qg + (size_t)row * kRW + c4 * 4);
            }
            const float* pg = P.p + (size_t)g * kRH * TT;
            for (int i = tid; i < kRH * TT; i += 256) p_s[i] = ldg(pg + i);
        }
        __syncthreads();
        colsum_rows(ds, LD, nullptr, 0, kD, T, part + 640 + DFF);
        {
            f32x4 acc[4];
            zero_acc(acc);
            frag_mma_nn(fwo, ds, LD, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * (kRDm / 4) + 4 * lr;
                const float4 v = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
                *reinterpret_cast<float4*>(wide + row * LW + col) = v;
                if (row < T) stg4(P.da + (size_t)(row0 + row) * kRDm + col, v);
            }
        }
        CF_STAMP(6);
        FragNN<4, 32> fd;                 // input-gradient product: this wave's K half of the q|k|v|g weight
        frag_load_nn(fd, P.watt + (size_t)(kh * 512) * kD + cg * 64, kD);
        __syncthreads();
        CF_STAMP(7);
        // ---- attention backward on the matrix cores (wave w owns heads 2w, 2w+1)
        constexpr int KT = TC > 0 ? (TC + 3) / 4 : (kRMaxT + 3) / 4;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {       // o = p v (recomputed), gate: do = da s(g), dg = da o s(1-s)
                const int h = 2 * w + hh, col = h * kRDh + nt * 16 + lr;
                const f32x4 o = mfma_small<KT>(
                    [&](int i, int j) { return p_s[j < T ? (h * T + min(i, T - 1)) * T + j : ZI]; },
                    [&](int j, int n) { return qk[min(j, T - 1) * kQkLd + 2 * kRDm + h * kRDh + nt * 16 + n]; });
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = lq * 4 + ii;
                    if (i < T) {
                        const float gt = qk[i * kQkLd + 3 * kRDm + col];
                        const float sg = fast_sigmoid(gt);
                        const float da = wide[i * LW + col];
                        do_s[i * kRDm + col] = da * sg;
                        dqk[i * kQkLd + 3 * kRDm + col] = da * o[ii] * sg * (1.0f - sg);
                    }
                }
            }
        CF_STAMP(8);
        __syncthreads();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {           // dp[i][j] = do[i] . v[j]
            const int h = 2 * w + hh;
            const f32x4 dp = mfma_small<kRDh / 4>(
                [&](int i, int d) { return do_s[min(i, T - 1) * kRDm + h * kRDh + d]; },
                [&](int d, int j) { return qk[min(j, T - 1) * kQkLd + 2 * kRDm + h * kRDh + d]; });
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = lq * 4 + ii, j = lr;
                if (i < T && j < T) s_s[(h * T + i) * T + j] = dp[ii];
            }
        }
        CF_STAMP(9);
        __syncthreads();
        for (int row = tid; row < kRH * T; row += 256) {
            const float* pr = p_s + row * T;
            float* dr = s_s + row * T;
            const int i = row % T;
            float dot = 0.f;
            _Pragma("unroll") for (int j = 0; j < T; ++j) dot = fmaf(pr[j], dr[j], dot);
            float gsum = 0.f;
            _Pragma("unroll") for (int j = 0; j < T; ++j) {
                const float v = mk_s[i * T + j] != 0.f ? 0.f : pr[j] * (dr[j] - dot);
                gsum = fmaf(v, fq_s[i * T + j], gsum);
                dr[j] = v;
            }
            red_s[row] = gsum;
        }
        CF_STAMP(10);
        __syncthreads();
        if (tid < kRH) {
            float sm = 0.f;
            _Pragma("unroll") for (int i = 0; i < T; ++i) sm += red_s[tid * T + i];
            stg(P.dgam + (size_t)g * kRH + tid, sm);
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int h = 2 * w + hh, cb = h * kRDh + nt * 16, col = cb + lr;
                // dq[i] = sum_j ds[i][j] k[j];  dk[j] = sum_i ds[i][j] q[i];  dv[j] = sum_i p[i][j] do[i]
                const f32x4 dq = mfma_small<KT>(
                    [&](int i, int j) { return s_s[j < T ? (h * T + min(i, T - 1)) * T + j : ZS]; },
                    [&](int j, int n) { return qk[min(j, T - 1) * kQkLd + kRDm + cb + n]; });
                const f32x4 dk = mfma_small<KT>(
                    [&](int j, int i) { return s_s[i < T ? (h * T + i) * T + min(j, T - 1) : ZS]; },
                    [&](int i, int n) { return qk[min(i, T - 1) * kQkLd + cb + n]; });
                const f32x4 dv = mfma_small<KT>(
                    [&](int j, int i) { return p_s[i < T ? (h * T + i) * T + min(j, T - 1) : ZI]; },
                    [&](int i, int n) { return do_s[min(i, T - 1) * kRDm + cb + n]; });
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int row = lq * 4 + ii;
                    if (row < T) {
                        dqk[row * kQkLd + col] = dq[ii] / scale;
                        dqk[row * kQkLd + kRDm + col] = dk[ii] / scale;
                        dqk[row * kQkLd + 2 * kRDm + col] = dv[ii];
                    }
                }
            }
        __syncthreads();
        {
            float* dg = P.dqkvg + (size_t)row0 * kRW;
            for (int i = tid; i < T * (kRW / 4); i += 256) {
                const int row = i >> 8, c4 = i & 255;
                stg4(dg + (size_t)row * kRW + c4 * 4, *reinterpret_cast<const float4*>(dqk + row * kQkLd + c4 * 4));
            }
        }
        CF_STAMP(11);
        // ---- d(layer input) = dt1 + dqkvg Watt   (K = 1024: two halves x two column groups)
        {
            f32x4 acc[4];
            zero_acc(acc);
            frag_mma_nn(fd, dqk + kh * 512, kQkLd, acc);
            float* red = wide;                  // [16][LD] scratch (da is dead)
            if (kh == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(red + (lq * 4 + i) * LD + cg * 64 + 4 * lr) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
            }
            __syncthreads();
            if (kh == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = lq * 4 + i, o = row * LD + cg * 64 + 4 * lr;
                    const float4 rv = *reinterpret_cast<const float4*>(red + o), tv = *reinterpret_cast<const float4*>(ds + o);
                    const float4 v = make_float4(acc[0][i] + rv.x + tv.x, acc[1][i] + rv.y + tv.y, acc[2][i] + rv.z + tv.z, acc[3][i] + rv.w + tv.w);
                    *reinterpret_cast<float4*>(ds + o) = v;
                    if (row < T) stg4(P.dxin + (size_t)(row0 + row) * kD + cg * 64 + 4 * lr, v);
                }
            }
        }
        CF_STAMP(12);
        __syncthreads();
    }
}

}  // namespace cf

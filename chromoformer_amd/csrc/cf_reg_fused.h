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
    float *xin, *qkvg, *p, *a, *xh1, *rs1, *y1, *hdn, *xh2, *rs2, *xout;
    float *dxout, *dt2, *dpre1, *dt1, *da, *dqkvg, *dxin, *partial, *dgam;
};
struct RegArgs {
    const RegLayerDev* tab;          // [n_res][n_layers]
    int n_layers, T;
    const uint8_t* mask[kMaxRes];    // [B,T,T]
    const float* freq;               // [B,T,T]
    int save;
    unsigned long long* tdbg;        // optional: shader-clock stamps of workgroup (0,0), 16 per layer
};
#define CF_STAMP(slot)                                                                                 \
    do {                                                                                               \
        if (a.tdbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)                          \
            a.tdbg[l * 16 + (slot)] = __builtin_amdgcn_s_memtime();                                    \
    } while (0)

constexpr int kQkLd = kRW + 4;
__host__ __device__ constexpr size_t reg_fwd_smem(int T) {
    return (size_t)(2 * kTile * (kD + 4) + 2 * kTile * (kRDm + 4) + T * kQkLd + kRH * T * T + 2 * T * T) * sizeof(float);
}
__host__ __device__ constexpr size_t reg_bwd_smem(int T) {
    return (size_t)(3 * kTile * (kD + 4) + kTile * (kRDm + 4) + T * kQkLd + kTile * kQkLd + 2 * kRH * T * T + kRH * T + 2 * T * T) *
           sizeof(float);
}

template <int DFF>
__global__ __launch_bounds__(256) void k_reg_fwd(RegArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x, r = blockIdx.y, T = a.T, TT = T * T, row0 = g * T, tid = threadIdx.x;
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
    const RegLayerDev* tab = a.tab + (size_t)r * a.n_layers;
    const float scale = sqrtf((float)kRDh);
    {
        const float* x0 = tab[0].xin + (size_t)row0 * kD;
        for (int i = tid; i < kTile * (kD / 4); i += 256) {
            const int row = i >> 5, c4 = i & 31;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < T) v = *reinterpret_cast<const float4*>(x0 + row * kD + c4 * 4);
            *reinterpret_cast<float4*>(xs + row * LD + c4 * 4) = v;
        }
        for (int i = tid; i < TT; i += 256) {
            fq_s[i] = a.freq[(size_t)g * TT + i];
            mk_s[i] = a.mask[r][(size_t)g * TT + i] ? 1.f : 0.f;
        }
    }
    FragNT<4, 8> fa, fb;
    frag_load_nt(fa, tab[0].watt_t + (size_t)(w * 256) * kD, kD);
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l) {
        const RegLayerDev P = tab[l];      // by value: the pointers live in SGPRs (no reload after every store)
        CF_STAMP(0);
        // ---- q|k|v|gate projection: wave w owns columns [w*256, w*256+256) in 4 chunks of 64
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            FragNT<4, 8>& cur = (c & 1) ? fb : fa;
            FragNT<4, 8>& nxt = (c & 1) ? fa : fb;
            if (c < 3) frag_load_nt(nxt, P.watt_t + (size_t)(w * 256 + (c + 1) * 64) * kD, kD);
            f32x4 acc[4];
            zero_acc(acc);
            frag_mma_nt(cur, xs, LD, acc);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = lq * 4 + i, col = w * 256 + c * 64 + col_nt(t, lr);
                    if (row < T) qk[row * kQkLd + col] = acc[t][i];
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
                *reinterpret_cast<float4*>(qg + (size_t)row * kRW + c4 * 4) = *reinterpret_cast<const float4*>(qk + row * kQkLd + c4 * 4);
            }
        }
        // ---- attention (modules.py:58-81)
        for (int idx = tid; idx < kRH * TT; idx += 256) {
            const int h = idx / TT, ij = idx - h * TT, i = ij / T, j = ij - i * T;
            const float4* qp = reinterpret_cast<const float4*>(qk + i * kQkLd + h * kRDh);
            const float4* kp = reinterpret_cast<const float4*>(qk + j * kQkLd + kRDm + h * kRDh);
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < kRDh / 4; ++d) {
                const float4 qv = qp[d], kv = kp[d];
                sc = fmaf(qv.x, kv.x, sc);
                sc = fmaf(qv.y, kv.y, sc);
                sc = fmaf(qv.z, kv.z, sc);
                sc = fmaf(qv.w, kv.w, sc);
            }
            sc = sc / scale + P.gamma[h] * fq_s[ij];
            if (mk_s[ij] != 0.f) sc = kMaskFill;
            p_s[idx] = sc;
        }
        CF_STAMP(3);
        __syncthreads();
        for (int row = tid; row < kRH * T; row += 256) {
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
        CF_STAMP(4);
        __syncthreads();
        if (a.save) {
            float* pg = P.p + (size_t)g * kRH * TT;
            for (int i = tid; i < kRH * TT; i += 256) pg[i] = p_s[i];
        }
        for (int idx = tid; idx < kTile * kRDm; idx += 256) {
            const int i = idx >> 8, c = idx & 255, h = c >> 5;
            float v = 0.f;
            if (i < T) {
                const float* pr = p_s + (h * T + i) * T;
                float o = 0.f;
                for (int j = 0; j < T; ++j) o = fmaf(pr[j], qk[j * kQkLd + 2 * kRDm + c], o);
                const float gt = qk[i * kQkLd + 3 * kRDm + c];
                v = o * (1.0f / (1.0f + expf(-gt)));
                if (a.save) P.a[(size_t)(row0 + i) * kRDm + c] = v;
            }
            as_[i * LW + c] = v;
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
                    ts[row * LD + col] = acc[t][i] + P.bo[col] + xs[row * LD + col];
                }
        }
        constexpr int NT1 = DFF / 64;
        FragNT<NT1, 8> f1;
        frag_load_nt(f1, P.w1_t + (size_t)(w * (DFF / 4)) * kD, kD);
        CF_STAMP(7);
        __syncthreads();
        ln_fwd_rows(ts, LD, P.g1, P.be1, row0, T, a.save ? P.xh1 : nullptr, P.rs1, a.save ? P.y1 : nullptr, identity_map());
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
                    const float v = fmaxf(acc[t][i] + P.b1[col], 0.f);
                    hs[row * LW + col] = v;
                    if (a.save && row < T) P.hdn[(size_t)(row0 + row) * DFF + col] = v;
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
                    xs[row * LD + col] = acc[t][i] + P.b2[col] + ts[row * LD + col];
                }
        }
        if (l + 1 < a.n_layers) frag_load_nt(fa, tab[l + 1].watt_t + (size_t)(w * 256) * kD, kD);   // next layer, chunk 0
        CF_STAMP(10);
        __syncthreads();
        ln_fwd_rows(xs, LD, P.g2, P.be2, row0, T, a.save ? P.xh2 : nullptr, P.rs2, P.xout, identity_map());
        CF_STAMP(11);
        __syncthreads();
    }
}

template <int DFF>
__global__ __launch_bounds__(256) void k_reg_bwd(RegArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x, r = blockIdx.y, T = a.T, TT = T * T, row0 = g * T, tid = threadIdx.x;
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
    float* do_s = xh;                    // [T][256] aliases xh|t2 (both dead during the attention backward)
    const RegLayerDev* tab = a.tab + (size_t)r * a.n_layers;
    const float scale = sqrtf((float)kRDh);
    auto load_rows = [&](float* dst, const float* src) {      // [T,128] rows of this gene -> 16-row LDS tile
        for (int i = tid; i < kTile * (kD / 4); i += 256) {
            const int row = i >> 5, c4 = i & 31;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < T) v = *reinterpret_cast<const float4*>(src + (size_t)(row0 + row) * kD + c4 * 4);
            *reinterpret_cast<float4*>(dst + row * LD + c4 * 4) = v;
        }
    };
    for (int i = tid; i < (kTile - T) * kQkLd; i += 256) dqk[T * kQkLd + i] = 0.f;      // dead rows of the MFMA operand
    for (int i = tid; i < TT; i += 256) {
        fq_s[i] = a.freq[(size_t)g * TT + i];
        mk_s[i] = a.mask[r][(size_t)g * TT + i] ? 1.f : 0.f;
    }
    load_rows(ds, tab[a.n_layers - 1].dxout);
    for (int l = a.n_layers - 1; l >= 0; --l) {
        const RegLayerDev P = tab[l];
        float* part = P.partial + (size_t)g * PW;
        FragNN<NT2, 8> fw2;
        frag_load_nn(fw2, P.w2 + w * (DFF / 4), DFF);
        load_rows(xh, P.xh2);
        __syncthreads();
        colsum16(ds, LD, xh, LD, kD, part + 0);
        colsum16(ds, LD, nullptr, 0, kD, part + 128);
        for (int i = tid; i < kTile * kD; i += 256) t2[(i >> 7) * LD + (i & 127)] = ds[(i >> 7) * LD + (i & 127)];
        __syncthreads();
        ln_bwd_rows(t2, LD, xh, LD, P.g2, P.rs2, row0, T, P.dt2);
        FragNN<2, DFF / 16> fw1;
        frag_load_nn(fw1, P.w1 + w * 32, kD);
        __syncthreads();
        colsum16(t2, LD, nullptr, 0, kD, part + 256);
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
                    const vec_t hv = *reinterpret_cast<const vec_t*>(P.hdn + o);
                    const float* hp = reinterpret_cast<const float*>(&hv);
#pragma unroll
                    for (int t = 0; t < NT2; ++t) v[t] = hp[t] > 0.f ? acc[t][i] : 0.f;
                    *reinterpret_cast<vec_t*>(P.dpre1 + o) = *reinterpret_cast<const vec_t*>(v);
                }
                *reinterpret_cast<vec_t*>(wide + row * LW + col) = *reinterpret_cast<const vec_t*>(v);
            }
        }
        load_rows(xh, P.xh1);
        FragNN<4, 8> fwo;
        frag_load_nn(fwo, P.wo + w * (kRDm / 4), kRDm);
        __syncthreads();
        colsum16(wide, LW, nullptr, 0, DFF, part + 384);
        {
            f32x4 acc[2];
            zero_acc(acc);
            frag_mma_nn(fw1, wide, LW, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * 32 + 2 * lr;
                ds[row * LD + col] = acc[0][i] + t2[row * LD + col];
                ds[row * LD + col + 1] = acc[1][i] + t2[row * LD + col + 1];
            }
        }
        __syncthreads();
        colsum16(ds, LD, xh, LD, kD, part + 384 + DFF);
        colsum16(ds, LD, nullptr, 0, kD, part + 512 + DFF);
        __syncthreads();
        ln_bwd_rows(ds, LD, xh, LD, P.g1, P.rs1, row0, T, P.dt1);       // ds = dt1
        // operands of the attention backward: in flight during the out-projection product
        {
            const float* qg = P.qkvg + (size_t)row0 * kRW;
            for (int i = tid; i < T * (kRW / 4); i += 256) {
                const int row = i >> 8, c4 = i & 255;
                *reinterpret_cast<float4*>(qk + row * kQkLd + c4 * 4) = *reinterpret_cast<const float4*>(qg + (size_t)row * kRW + c4 * 4);
            }
            const float* pg = P.p + (size_t)g * kRH * TT;
            for (int i = tid; i < kRH * TT; i += 256) p_s[i] = pg[i];
        }
        __syncthreads();
        colsum16(ds, LD, nullptr, 0, kD, part + 640 + DFF);
        {
            f32x4 acc[4];
            zero_acc(acc);
            frag_mma_nn(fwo, ds, LD, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * (kRDm / 4) + 4 * lr;
                const float4 v = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
                *reinterpret_cast<float4*>(wide + row * LW + col) = v;
                if (row < T) *reinterpret_cast<float4*>(P.da + (size_t)(row0 + row) * kRDm + col) = v;
            }
        }
        FragNN<2, 16> fda, fdb;           // first K-chunk of the input-gradient product
        frag_load_nn(fda, P.watt + w * 32, kD);
        __syncthreads();
        // ---- attention backward (gate, value, softmax, score sides)
        for (int idx = tid; idx < T * kRDm; idx += 256) {
            const int i = idx >> 8, c = idx & 255, h = c >> 5;
            const float* pr = p_s + (h * T + i) * T;
            float o = 0.f;
            for (int j = 0; j < T; ++j) o = fmaf(pr[j], qk[j * kQkLd + 2 * kRDm + c], o);
            const float gt = qk[i * kQkLd + 3 * kRDm + c];
            const float sg = 1.0f / (1.0f + expf(-gt));
            const float da = wide[i * LW + c];
            do_s[idx] = da * sg;
            dqk[i * kQkLd + 3 * kRDm + c] = da * o * sg * (1.0f - sg);
        }
        __syncthreads();
        for (int idx = tid; idx < kRH * TT; idx += 256) {
            const int h = idx / TT, ij = idx - h * TT, i = ij / T, j = ij - i * T;
            const float4* dp = reinterpret_cast<const float4*>(do_s + i * kRDm + h * kRDh);
            const float4* vp = reinterpret_cast<const float4*>(qk + j * kQkLd + 2 * kRDm + h * kRDh);
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < kRDh / 4; ++d) {
                const float4 x = dp[d], y = vp[d];
                sc = fmaf(x.x, y.x, sc);
                sc = fmaf(x.y, y.y, sc);
                sc = fmaf(x.z, y.z, sc);
                sc = fmaf(x.w, y.w, sc);
            }
            s_s[idx] = sc;
        }
        __syncthreads();
        for (int row = tid; row < kRH * T; row += 256) {
            const float* pr = p_s + row * T;
            float* dr = s_s + row * T;
            const int i = row % T;
            float dot = 0.f;
            for (int j = 0; j < T; ++j) dot = fmaf(pr[j], dr[j], dot);
            float gsum = 0.f;
            for (int j = 0; j < T; ++j) {
                const float v = mk_s[i * T + j] != 0.f ? 0.f : pr[j] * (dr[j] - dot);
                gsum = fmaf(v, fq_s[i * T + j], gsum);
                dr[j] = v;
            }
            red_s[row] = gsum;
        }
        __syncthreads();
        if (tid < kRH) {
            float sm = 0.f;
            for (int i = 0; i < T; ++i) sm += red_s[tid * T + i];
            P.dgam[(size_t)g * kRH + tid] = sm;
        }
        for (int idx = tid; idx < T * kRDm; idx += 256) {
            const int i = idx >> 8, c = idx & 255, h = c >> 5;
            float dq = 0.f, dk = 0.f, dv = 0.f;
            for (int j = 0; j < T; ++j) {
                dq = fmaf(s_s[(h * T + i) * T + j], qk[j * kQkLd + kRDm + c], dq);
                dk = fmaf(s_s[(h * T + j) * T + i], qk[j * kQkLd + c], dk);
                dv = fmaf(p_s[(h * T + j) * T + i], do_s[j * kRDm + c], dv);
            }
            dqk[i * kQkLd + c] = dq / scale;
            dqk[i * kQkLd + kRDm + c] = dk / scale;
            dqk[i * kQkLd + 2 * kRDm + c] = dv;
        }
        __syncthreads();
        {
            float* dg = P.dqkvg + (size_t)row0 * kRW;
            for (int i = tid; i < T * (kRW / 4); i += 256) {
                const int row = i >> 8, c4 = i & 255;
                *reinterpret_cast<float4*>(dg + (size_t)row * kRW + c4 * 4) = *reinterpret_cast<const float4*>(dqk + row * kQkLd + c4 * 4);
            }
        }
        // ---- d(layer input) = dt1 + dqkvg Watt   (K = 1024 in four chunks)
        {
            f32x4 acc[2];
            zero_acc(acc);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                FragNN<2, 16>& cur = (c & 1) ? fdb : fda;
                FragNN<2, 16>& nxt = (c & 1) ? fda : fdb;
                if (c < 3) frag_load_nn(nxt, P.watt + (size_t)((c + 1) * 256) * kD + w * 32, kD);
                frag_mma_nn(cur, dqk + c * 256, kQkLd, acc);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lq * 4 + i, col = w * 32 + 2 * lr;
                const float v0 = acc[0][i] + ds[row * LD + col], v1 = acc[1][i] + ds[row * LD + col + 1];
                ds[row * LD + col] = v0;
                ds[row * LD + col + 1] = v1;
                if (row < T) *reinterpret_cast<float2*>(P.dxin + (size_t)(row0 + row) * kD + col) = make_float2(v0, v1);
            }
        }
        __syncthreads();
    }
}

}  // namespace cf

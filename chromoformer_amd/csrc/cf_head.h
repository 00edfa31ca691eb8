// cf_head.h -- the prediction head, one kernel per direction (included by cf_kernels.h).
//
//   forward  (net.py:377-380): hin = cat_r(x_out[r][:, 0] + x_in[r][:, 0]);  h1 = relu(W1 hin + b1);  logits = W2 h1 + b2
//   backward (train.py:156, 193-195): CrossEntropyLoss / MSELoss (mean over the batch) -> dlogits -> dh1 -> dhin, which
//            is also the gradient of token 0 of every Regulation output
//
// One workgroup per 16 genes.  The head is a few MFLOP: what matters is that it is two launches instead of nine.
#pragma once
// (the head kernels keep the plain thread index: with the opaque one of cf_kernels.h -- CF_TID_OPAQUE, there for the fused trunk -- k_head_bwd
// went from 174 registers to 256 with 16 spilled)
#pragma push_macro("threadIdx")
#undef threadIdx

namespace cf {

struct HeadFwdArgs {
    const float* xl[kMaxRes];    // Regulation output [B*T, 128]
    const float* x0[kMaxRes];    // Regulation input  [B*T, 128]
    const float* w1_t;           // fc_head.0.weight [128, n_res*128], tiled copy
    const float *b1, *w2, *b2;   // fc_head.0.bias, fc_head.2.weight [n_out, 128], fc_head.2.bias
    float *hin, *h1, *logits;    // saved for the backward pass: [B, n_res*128], [B, 128], [B, n_out]
    float* logits_user;          // caller's copy (may be null)
    int B, T, n_res, n_out;
    unsigned long long* tdbg;    // optional shader-clock stamps of workgroup 0 (tools/head_stamps.py)
};
constexpr int kHeadThreads = 512;      // eight waves: wave w owns 16 of the 128 hidden columns
#define CF_HSTAMP(slot)                                                                              \
    do {                                                                                             \
        if (a.tdbg && blockIdx.x == 0 && threadIdx.x == 0) a.tdbg[slot] = __builtin_amdgcn_s_memtime(); \
    } while (0)
__device__ __forceinline__ void head_fwd_body(const HeadFwdArgs& a) {
    // all n_res resolution chunks of the concatenated input are fetched at once into one [16][n_res*128] tile (one barrier), then
    // every wave runs ONE product over the whole reduction range with a continuous operand ring
    constexpr int KMAX = kMaxRes * kD;
    __shared__ __attribute__((aligned(16))) float xs[kTile][KMAX + 4];
    __shared__ __attribute__((aligned(16))) float hs[kTile][kD + 4];
    const int row0 = blockIdx.x * kTile, K = a.n_res * kD, tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int col0 = w * 16;
    CF_HSTAMP(0);
    // The whole operand stream of the wave (16 hidden columns x K = n_res * 128: 1 KB per 16-deep k-block) is requested at once:
    // a 3-chunk ring covers ~800 cycles of MFMA work against ~2 K cycles of L2 latency, i.e. the product used to stall on every
    // chunk (12 x ~1.3 K cycles); 24 sixteen-byte loads in flight cost 96 registers this kernel has to spare.
    constexpr int KB = kMaxRes * kD / 16;            // 16-deep k-blocks at most
    float4 wb[KB];
    const float* wp = a.w1_t + (size_t)col0 * K + lane * 4;
    const int nkb = K / 16;
    // input rows first (loads return in order: the tile's puts wait for these alone), the operand stream behind them
    constexpr int NTL = kTile * (KMAX / 4) / kHeadThreads;      // float4 pairs per thread: 3
    float4 tp[NTL], tq[NTL];
#pragma unroll
    for (int u = 0; u < NTL; ++u) {
        const int i = min(tid + u * kHeadThreads, kTile * (K / 4) - 1);
        const int row = i / (K / 4), c4k = i - row * (K / 4), r = c4k >> 5, c4 = c4k & 31, g = min(row0 + row, a.B - 1);
        const size_t o = (size_t)g * a.T * kD + c4 * 4;
        tp[u] = ldg4(a.xl[r] + o);
        tq[u] = ldg4(a.x0[r] + o);
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) wb[kb] = ldg4(wp + (size_t)min(kb, nkb - 1) * 256);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NTL; ++u) {
        const int i = tid + u * kHeadThreads;
        if (i < kTile * (K / 4)) {
            const int row = i / (K / 4), c4k = i - row * (K / 4), r = c4k >> 5, c4 = c4k & 31, g = row0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g < a.B) {
                v = make_float4(tp[u].x + tq[u].x, tp[u].y + tq[u].y, tp[u].z + tq[u].z, tp[u].w + tq[u].w);
                stg4(a.hin + (size_t)g * K + r * kD + c4 * 4, v);
            }
            *reinterpret_cast<float4*>(&xs[row][c4k * 4]) = v;
        }
    }
    // operands of the logits (fc_head.2) and the first bias: requested here, consumed two barriers later
    const float b1v = ldg(a.b1 + col0 + lr);
    float w2v[2][kD / 16];
    {
        const int sub = tid & 15;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < kD / 16; ++k) w2v[c][k] = ldg(a.w2 + min(c, a.n_out - 1) * kD + sub + 16 * k);
    }
    __syncthreads();
    CF_HSTAMP(1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        const float* ap = &xs[lr][lq * 4];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            if (kb < nkb) {
                const float4 av = *reinterpret_cast<const float4*>(ap + kb * 16);
                const float4 b = wb[kb];
                acc = mfma4(av.x, b.x, acc);
                acc = mfma4(av.y, b.y, acc);
                acc = mfma4(av.z, b.z, acc);
                acc = mfma4(av.w, b.w, acc);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = lq * 4 + i, col = col0 + lr;
        const float v = fmaxf(acc[i] + b1v, 0.f);
        hs[row][col] = v;
        if (row0 + row < a.B) stg(a.h1 + (size_t)(row0 + row) * kD + col, v);
    }
    CF_HSTAMP(2);
    __syncthreads();
    if (tid < 256) {   // logits: 16 lanes per gene
        const int row = tid >> 4, sub = tid & 15, g = row0 + row;
        for (int c = 0; c < a.n_out; ++c) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < kD / 16; ++k) s = fmaf(hs[row][sub + 16 * k], w2v[c & 1][k], s);
            s = group16_sum(s);
            if (sub == 0 && g < a.B) {
                const float v = s + a.b2[c];
                a.logits[g * a.n_out + c] = v;
                if (a.logits_user) a.logits_user[g * a.n_out + c] = v;
            }
        }
    }
}

__global__ __launch_bounds__(kHeadThreads) void k_head_fwd(HeadFwdArgs a) { head_fwd_body(a); }

struct HeadBwdArgs {
    const float* logits;         // [B, n_out]
    const void* labels;          // int64 [B] (n_out = 2) / float [B] (n_out = 1); null: dlogits is given
    const float *w2, *h1;        // fc_head.2.weight, saved hidden layer
    const float* w1;             // fc_head.0.weight [128, n_res*128], row-major
    float *dlogits, *dh1, *dhin; // [B, n_out], [B, 128], [B, n_res*128]  (kept for the weight gradients / k_join)
    float* dxl[kMaxRes];         // Regulation output gradient [B*T, 128]: token 0 receives dhin (the other rows stay zero)
    float* loss;                 // [0] mean loss, [1] user copy flag unused, [2] arrival counter (uint), workspace
    float* loss_part;            // [tiles]
    float* loss_user;            // may be null
    float gscale;
    int B, T, n_res, n_out;
    unsigned long long* tdbg;
};
__device__ __forceinline__ void head_bwd_body(const HeadBwdArgs& a) {
    __shared__ __attribute__((aligned(16))) float ds[kTile][kD + 4];
    __shared__ float dl[kTile][2];
    __shared__ float li[kTile];
    const int row0 = blockIdx.x * kTile, K = a.n_res * kD, tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    // what the later phases read from global memory is requested up front (every phase used to start with a round trip of its
    // own): the W1 operand of this wave's first column block of the dhin product, the W2 rows and the ReLU mask of dh1
    FragNN<4, 8> f0;
    frag_load_nn(f0, a.w1 + min(w, K / 64 - 1) * 64, K);
    constexpr int NDH = kTile * kD / kHeadThreads;      // dh1 entries per thread: 4
    float w2a[NDH], w2b[NDH], h1v[NDH];
#pragma unroll
    for (int u = 0; u < NDH; ++u) {
        const int i = tid + u * kHeadThreads, row = i >> 7, j = i & 127, g = min(row0 + row, a.B - 1);
        w2a[u] = ldg(a.w2 + j);
        w2b[u] = ldg(a.w2 + (a.n_out == 2 ? kD : 0) + j);
        h1v[u] = ldg(a.h1 + (size_t)g * kD + j);
    }
    CF_HSTAMP(3);
    if (tid < kTile) {        // loss and d loss / d logits of one gene
        const int g = row0 + tid;
        float l = 0.f, d0 = 0.f, d1 = 0.f;
        if (g < a.B) {
            if (!a.labels) {
                d0 = a.dlogits[g * a.n_out];
                if (a.n_out == 2) d1 = a.dlogits[g * 2 + 1];
            } else if (a.n_out == 1) {
                const float y = reinterpret_cast<const float*>(a.labels)[g];
                const float d = a.logits[g] - y;
                l = d * d;
                d0 = 2.0f * d * a.gscale / (float)a.B;
                a.dlogits[g] = d0;
            } else {
                const int y = (int)reinterpret_cast<const long long*>(a.labels)[g];
                const float z0 = a.logits[g * 2], z1 = a.logits[g * 2 + 1];
                const float m = fmaxf(z0, z1);
                const float lse = m + logf(expf(z0 - m) + expf(z1 - m));
                l = lse - (y ? z1 : z0);
                d0 = (expf(z0 - lse) - (y == 0 ? 1.f : 0.f)) * a.gscale / (float)a.B;
                d1 = (expf(z1 - lse) - (y == 1 ? 1.f : 0.f)) * a.gscale / (float)a.B;
                a.dlogits[g * 2] = d0;
                a.dlogits[g * 2 + 1] = d1;
            }
        }
        dl[tid][0] = d0;
        dl[tid][1] = d1;
        li[tid] = l;
    }
    CF_HSTAMP(4);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NDH; ++u) {      // dh1 = (dlogits W2) * (h1 > 0)
        const int i = tid + u * kHeadThreads, row = i >> 7, j = i & 127, g = row0 + row;
        float s = dl[row][0] * w2a[u];
        if (a.n_out == 2) s = fmaf(dl[row][1], w2b[u], s);
        const float v = (g < a.B && h1v[u] > 0.f) ? s : 0.f;
        ds[row][j] = v;
        if (g < a.B) stg(a.dh1 + (size_t)g * kD + j, v);
    }
    CF_HSTAMP(5);
    __syncthreads();
    for (int jb = w; jb < K / 64; jb += kHeadThreads / 64) {          // dhin[:, 64 jb ..] = dh1 . W1[:, 64 jb ..]
        FragNN<4, 8> f;
        if (jb == w) f = f0;
        else frag_load_nn(f, a.w1 + jb * 64, K);
        f32x4 acc[4];
        zero_acc(acc);
        frag_mma_nn(f, &ds[0][0], kD + 4, acc);
        const int r = jb >> 1, e0 = (jb & 1) * 64 + 4 * lr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = row0 + lq * 4 + i;
            if (g < a.B) {
                const float4 v = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
                stg4(a.dhin + (size_t)g * K + jb * 64 + 4 * lr, v);
                stg4(a.dxl[r] + (size_t)g * a.T * kD + e0, v);
            }
        }
    }
    CF_HSTAMP(6);
    // (last: the fences of this block cost a few thousand cycles, and nothing in the kernel waits for the loss)
    if (a.labels && tid == 0) {       // mean loss: per-workgroup partials, summed in workgroup order by the last to arrive
        float s = 0.f;
        for (int i = 0; i < kTile; ++i) s += li[i];
        a.loss_part[blockIdx.x] = s;
        __threadfence();
        unsigned* counter = reinterpret_cast<unsigned*>(a.loss + 2);
        if (atomicAdd(counter, 1u) == gridDim.x - 1) {
            __threadfence();
            float tot = 0.f;
            for (unsigned i = 0; i < gridDim.x; ++i) tot += *(volatile float*)(a.loss_part + i);
            tot /= (float)a.B;
            a.loss[0] = tot;
            if (a.loss_user) a.loss_user[0] = tot;
            *counter = 0u;
        }
    }
    CF_HSTAMP(7);
}

__global__ __launch_bounds__(kHeadThreads) void k_head_bwd(HeadBwdArgs a) { head_bwd_body(a); }
// Training step: head forward, loss and head backward of a 16-gene tile in ONE launch -- nothing of the backward half depends on
// another tile (the mean loss is only reported), and the two launches were 12 + 13 us of latency for a few MFLOP.  The logits
// and the hidden layer travel through global memory (they are saved anyway) behind a workgroup barrier.
__global__ __launch_bounds__(kHeadThreads) void k_head_train(HeadFwdArgs f, HeadBwdArgs b) {
    head_fwd_body(f);
    __syncthreads();
    head_bwd_body(b);
}

// ---- any other hidden width (net.py:278 leaves d_head free): one 256-thread workgroup per gene, vector ALUs, two launches.  The
//      default width runs the matrix-core kernels above; this path costs ~25 us more per step and exists so that such a model runs at all.
struct HeadGenArgs {
    const float* xl[kMaxRes];    // Regulation output [B*T, 128]
    const float* x0[kMaxRes];    // Regulation input  [B*T, 128]
    const float *w1, *b1, *w2, *b2;      // fc_head.0.weight [DH, n_res*128] (row-major), .bias, fc_head.2.weight [n_out, DH], .bias
    float *hin, *h1, *logits, *logits_user;
    const void* labels;
    float *dlogits, *dh1, *dhin;
    float* dxl[kMaxRes];
    float *loss, *loss_part, *loss_user;
    float gscale;
    int B, T, n_res, n_out, DH;
    int D;                       // row width of the Regulation rows (d_emb: 128, or 256 since round 5)
};
constexpr int kHeadGenMaxDH = 1024;
constexpr int kHeadGenMaxD = 256;
__global__ __launch_bounds__(256) void k_head_gen_fwd(HeadGenArgs a) {
    __shared__ float xs[kMaxRes * kHeadGenMaxD];
    __shared__ float hs[kHeadGenMaxDH];
    const int kD = a.D;      // (row width: shadows cf::kD)
    const int g = blockIdx.x, tid = threadIdx.x, K = a.n_res * kD, DH = a.DH;
    for (int k = tid; k < K; k += 256) {
        const int r = k / kD, c = k % kD;
        const size_t o = (size_t)g * a.T * kD + c;
        const float v = ldg(a.xl[r] + o) + ldg(a.x0[r] + o);
        xs[k] = v;
        stg(a.hin + (size_t)g * K + k, v);
    }
    __syncthreads();
    const int part = tid & 3, kq = K >> 2;      // four lanes per hidden unit, a quarter of the reduction each
    for (int j0 = 0; j0 < DH; j0 += 64) {
        const int j = j0 + (tid >> 2);
        float s = 0.f;
        if (j < DH) {
            const float* wr = a.w1 + (size_t)j * K + part * kq;
            const float* xp = xs + part * kq;
            for (int k = 0; k < kq; ++k) s = fmaf(ldg(wr + k), xp[k], s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (j < DH && part == 0) {
            const float h = fmaxf(s + ldg(a.b1 + j), 0.f);
            hs[j] = h;
            stg(a.h1 + (size_t)g * DH + j, h);
        }
    }
    __syncthreads();
    if (tid < 64)
        for (int c = 0; c < a.n_out; ++c) {
            float s = 0.f;
            for (int j = tid; j < DH; j += 64) s = fmaf(hs[j], ldg(a.w2 + (size_t)c * DH + j), s);
            s = wave_sum(s);
            if (tid == 0) {
                const float v = s + ldg(a.b2 + c);
                a.logits[g * a.n_out + c] = v;
                if (a.logits_user) a.logits_user[g * a.n_out + c] = v;
            }
        }
}
__global__ __launch_bounds__(256) void k_head_gen_bwd(HeadGenArgs a) {
    __shared__ float ds[kHeadGenMaxDH];
    __shared__ float dl[2];
    const int kD = a.D;      // (row width: shadows cf::kD)
    const int g = blockIdx.x, tid = threadIdx.x, K = a.n_res * kD, DH = a.DH;
    if (tid == 0) {      // loss and d loss / d logits of the gene (CrossEntropyLoss / MSELoss, mean over the batch: train.py:156, 193)
        float l = 0.f, d0 = 0.f, d1 = 0.f;
        if (!a.labels) {
            d0 = a.dlogits[g * a.n_out];
            if (a.n_out == 2) d1 = a.dlogits[g * 2 + 1];
        } else if (a.n_out == 1) {
            const float d = a.logits[g] - reinterpret_cast<const float*>(a.labels)[g];
            l = d * d;
            d0 = 2.0f * d * a.gscale / (float)a.B;
            a.dlogits[g] = d0;
        } else {
            const int y = (int)reinterpret_cast<const long long*>(a.labels)[g];
            const float z0 = a.logits[g * 2], z1 = a.logits[g * 2 + 1];
            const float m = fmaxf(z0, z1);
            const float lse = m + logf(expf(z0 - m) + expf(z1 - m));
            l = lse - (y ? z1 : z0);
            d0 = (expf(z0 - lse) - (y == 0 ? 1.f : 0.f)) * a.gscale / (float)a.B;
            d1 = (expf(z1 - lse) - (y == 1 ? 1.f : 0.f)) * a.gscale / (float)a.B;
            a.dlogits[g * 2] = d0;
            a.dlogits[g * 2 + 1] = d1;
        }
        dl[0] = d0;
        dl[1] = d1;
        a.loss_part[g] = l;
    }
    __syncthreads();
    for (int j = tid; j < DH; j += 256) {      // dh1 = (dlogits W2) * (h1 > 0)
        float s = dl[0] * ldg(a.w2 + j);
        if (a.n_out == 2) s = fmaf(dl[1], ldg(a.w2 + DH + j), s);
        const float v = ldg(a.h1 + (size_t)g * DH + j) > 0.f ? s : 0.f;
        ds[j] = v;
        stg(a.dh1 + (size_t)g * DH + j, v);
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {       // dhin = dh1 W1; token 0 of the Regulation output gradient receives it
        float s = 0.f;
        for (int j = 0; j < DH; ++j) s = fmaf(ds[j], ldg(a.w1 + (size_t)j * K + k), s);
        stg(a.dhin + (size_t)g * K + k, s);
        stg(a.dxl[k / kD] + (size_t)g * a.T * kD + (k % kD), s);
    }
    if (a.labels && tid == 0) {                // mean loss: summed in gene order by the last workgroup to arrive
        __threadfence();
        unsigned* counter = reinterpret_cast<unsigned*>(a.loss + 2);
        if (atomicAdd(counter, 1u) == gridDim.x - 1) {
            __threadfence();
            float tot = 0.f;
            for (int i = 0; i < a.B; ++i) tot += *(volatile float*)(a.loss_part + i);
            tot /= (float)a.B;
            a.loss[0] = tot;
            if (a.loss_user) a.loss_user[0] = tot;
            *counter = 0u;
        }
    }
}

}  // namespace cf
#pragma pop_macro("threadIdx")

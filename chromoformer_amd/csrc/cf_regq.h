// cf_regq.h -- the Regulation stack forward (net.py:142-153) with a TEAM of four 256-thread workgroups per (gene, resolution)
// (included by cf_kernels.h behind cf_reg8.h, whose operand-stream helpers and saved-activation layouts it shares).
//
// k_reg8_fwd runs one 512-thread workgroup per (gene, resolution): 192 workgroups on 256 CUs, eight wave-jobs (one head each) of
// 16 K cycles of matrix-pipe issue per layer on every occupied SIMD pair and nothing on a quarter of the chip.  Here a unit is split
// over four workgroups -- member j owns heads 2 j, 2 j + 1 and a quarter of the FFN's hidden columns -- of four waves each, and a
// head over two waves (q | k and the softmax; v | gate and p v): 3,072 half-jobs, three workgroups per CU, three waves per SIMD.
//   * attention stays inside a member (the two waves of a head hand the probabilities over in LDS);
//   * the two products whose reduction dimension is split over the members (out-projection: K = this member's 64 attention
//     columns; FFN 2: K = its 64 hidden columns) leave a partial [T x 128] tile per member; the members SUM them through global memory:
//     plain stores, `s_waitcnt vmcnt(0)`, a relaxed agent-scope counter, agent-scope (sc1: never the vector L1) loads of the four
//     partials in member order -- every member forms the same bits and carries the full row tile on.  The four members of a team are 8
//     workgroup ids apart: workgroup ids go round-robin over the 8 XCDs (tools/probes/team_exchange.hip: HW_REG_XCC_ID = id mod 8), so
//     a team lives behind ONE L2 and an exchange costs ~2 us which the other two workgroups of the CU fill (the same exchange
//     across XCDs would need write-through stores and never-reused slots: 5 us).  cf_create checks the id -> XCD rule on the device
//     before it enables this kernel.
//   * LayerNorms are computed by every member (16 rows: nothing); only member 0 writes their saves and the layer output.
// Saved activations: exactly k_reg8_fwd's (cf_reg8.h header), so k_reg8_bwd runs behind either forward.  Sums over the split
// reduction dimensions are formed in another order than in k_reg8_fwd: the two forwards agree to fp32 rounding, not bit for bit.
#pragma once

#ifndef CF_TQ_PRE
#define CF_TQ_PRE 3
#endif
#ifndef CF_TQ_SLEEP
#define CF_TQ_SLEEP 2
#endif
namespace cf {

constexpr int kTqLA = 64 + 4;                                   // row stride of the member's 64-column tiles
constexpr int kTqSlot = kTile * kD;                             // floats of one member's partial tile
constexpr int kTqCnt = 32;                                      // ints per team: [0] arrivals, [16] departures
__host__ __device__ constexpr size_t regq_fwd_smem() {
    return (size_t)(2 * kTile * (kD + 4) + 2 * kTile * kTqLA + 2 * kPatchF) * sizeof(float);
}
__host__ __device__ constexpr size_t regq_slot_floats(int units) { return (size_t)units * 2 * 4 * kTqSlot; }

// 16-byte agent-scope load (sc1) through a raw buffer descriptor: the compiler tracks its wait count like any other load
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t rs, unsigned byte) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// One exchange: this member's partial rows [T][128] are in its slot already (plain stores by all waves).  Returns with the four slots
// readable.  cnt[0] counts arrivals over the whole launch (4 per exchange).
// One exchange: this member's partial rows [T][128] are in its slot (plain stores by all waves).  team_arrive: the stores are out, one
// arrival per member; team_wait: the four slots are readable.  What the caller requests between the two lands under the wait.
// (Per-wave arrivals and per-wave spinning -- no workgroup barrier around an exchange -- were measured: 16 pollers per team instead of 4 on the
// same counter line, the launch 176 us instead of 121.)
__device__ __forceinline__ void team_arrive(int* cnt) {
    __builtin_amdgcn_s_waitcnt(0);      // this wave's stores have been acknowledged by the L2
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void team_wait(int* cnt, int target) {
    if (threadIdx.x == 0)
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(CF_TQ_SLEEP);
    __syncthreads();
}

#define CF_STAMPQ(slot)                                                                                   \
    do {                                                                                                  \
        if (a.tdbg && v == 0 && mj == 0 && tid == 0 && l < 6) a.tdbg[l * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
template <bool SAVE>
__global__ __launch_bounds__(256, 3) void k_regq_fwd(RegArgs a) {
    constexpr int DFF = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int team = (b >> 5) * 8 + (b & 7), mj = __builtin_amdgcn_readfirstlane((b >> 3) & 3);
    const int per = gridDim.x >> 5;                                    // teams per XCD
    const int v = a.xcd_map ? (team & 7) * per + (team >> 3) : team;   // units of one resolution share an XCD's L2 (as in k_reg8_fwd)
    if (v >= a.B * a.n_res) return;
    if (a.tdbg && threadIdx.x == 0 && b < 464) a.tdbg[96 + 2 * b] = __builtin_amdgcn_s_memtime();      // (tools/regq_stamps.py: start / end of every workgroup)
    const int g = __builtin_amdgcn_readfirstlane(v % a.B), r = __builtin_amdgcn_readfirstlane(v / a.B), T = a.T, TT = T * T, row0 = g * T, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int hl = w >> 1, hh = 2 * mj + hl;      // head of this wave: local index in the member, index in the layer
    const bool role_v = (w & 1) != 0;             // false: q | k, scores, softmax;  true: v | gate, p v, gating
    constexpr int LD = kD + 4, LA = kTqLA;
    float* xs = smem;                    // [16][LD]   layer input / output
    float* ts = xs + kTile * LD;         // [16][LD]
    float* as_ = ts + kTile * LD;        // [16][LA]   gated attention output of the member's two heads
    float* hs = as_ + kTile * LA;        // [16][LA]   the member's 64 hidden columns
    float* qs = hs + kTile * LA + hl * kPatchF;     // per head: q rows [16][36]
    float* ks = qs + 16 * 36;                       //           k rows [16][36]
    float* ps = ks + 16 * 36;                       //           p      [16][20]
    const RegLayerDev* tab = a.tab + (size_t)r * a.n_layers;
    const float scale = sqrtf((float)kRDh);
    {
        const float* x0 = load_layer(tab).xin + (size_t)row0 * kD;
        for (int i = tid; i < kTile * (kD / 4); i += 256) {
            const int row = i >> 5, c4 = i & 31;
            float4 t = f4z();
            if (row < T) t = ldg4(x0 + row * kD + c4 * 4);
            *reinterpret_cast<float4*>(xs + row * LD + c4 * 4) = t;
        }
        for (int i = tid; i < 2 * kTile * LA; i += 256) as_[i] = 0.f;      // (as_ and hs) rows >= T stay zero (MFMA operand)
    }
    const int oD36 = lq * 4 * 36 + lr, oA36 = lr * 36 + lq * 4, oD20 = lq * 4 * 20 + lr, oA20 = lr * 20 + lq * 4;
    const int oALD = lr * LD + lq * 4, oALA = lr * LA + lq * 4;
    const int hq0 = (g * kRH + hh) * kHqFloats;
    const unsigned bT = (lq * 64 + lr * 4) * 4, bV = (lq * 4 * 32 + lr) * 4;
    const unsigned zA = (lq * 4 * kRDm + lr) * 4, zH = (lq * 4 * DFF + lr) * 4;
    const unsigned wl = lane * 16;
    bool rok[4];
    float fqv[4];
    bool mkv[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int i = lq * 4 + ii;
        rok[ii] = i < T;
        const bool in = i < T && lr < T;
        fqv[ii] = in ? ldg(a.freq + (size_t)g * TT + i * T + lr) : 0.f;
        mkv[ii] = in ? *(const CF_GLOBAL uint8_t*)(a.mask[r] + (size_t)g * TT + i * T + lr) != 0 : false;
    }
    // exchange slots of the unit: [parity][member][16][128]; this member's partial rows are written straight from the accumulators
    float* uslots = a.team_slots + (size_t)v * 2 * 4 * kTqSlot;
    int* ucnt = a.team_cnt + (size_t)v * kTqCnt;
    const unsigned pD = ((lq * 4) * kD + w * 32 + lr) * 4;      // byte offset of (row 4 lq, column 32 w + lr) in a partial tile
    int n_ex = 0;
    // The sum of the four partials + bias + residual and the LayerNorm behind it, in ONE pass in registers: thread = (row, 8 columns) -- the
    // LayerNorm's own layout (16 lanes per row), so the row statistics are two 16-lane reductions and nothing goes through LDS in between.
    // The normalised row is written to the LDS tile `dst`; saves (xhat, rstd, y) go to global memory where the pointers are given.
    const int lrow = w * 4 + lq, lsub = lr;      // row of this lane in the tile, its 8-column group
    const auto team_sum_ln = [&](int par, const float* bias, const float* resid, float* dst, const LnParams& lp, float* xhat_g, float* rstd_g, float* y_g) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uslots + (size_t)par * 4 * kTqSlot), 0, 4 * kTqSlot * 4, 0x00020000);
        const unsigned o = (lrow * kD + lsub * 8) * 4;
        float4 pa[4], pb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {      // (rows >= T: nothing was exchanged, nothing is read -- they carry the residual on, finite and never stored)
            pa[m] = pb[m] = f4z();
            if (lrow < T) {
                pa[m] = ld16_sc1(rs, m * kTqSlot * 4 + o);
                pb[m] = ld16_sc1(rs, m * kTqSlot * 4 + o + 16);
            }
        }
        const float4 ba = ldg4(bias + lsub * 8), bb = ldg4(bias + lsub * 8 + 4);
        const float4 ra = lds4(resid + lrow * LD + lsub * 8), rb = lds4(resid + lrow * LD + lsub * 8 + 4);
        float4 v0, v1;
        v0.x = (((pa[0].x + pa[1].x) + pa[2].x) + pa[3].x) + ba.x + ra.x;
        v0.y = (((pa[0].y + pa[1].y) + pa[2].y) + pa[3].y) + ba.y + ra.y;
        v0.z = (((pa[0].z + pa[1].z) + pa[2].z) + pa[3].z) + ba.z + ra.z;
        v0.w = (((pa[0].w + pa[1].w) + pa[2].w) + pa[3].w) + ba.w + ra.w;
        v1.x = (((pb[0].x + pb[1].x) + pb[2].x) + pb[3].x) + bb.x + rb.x;
        v1.y = (((pb[0].y + pb[1].y) + pb[2].y) + pb[3].y) + bb.y + rb.y;
        v1.z = (((pb[0].z + pb[1].z) + pb[2].z) + pb[3].z) + bb.z + rb.z;
        v1.w = (((pb[0].w + pb[1].w) + pb[2].w) + pb[3].w) + bb.w + rb.w;
        const float mean = group16_sum(sum4(v0) + sum4(v1)) * (1.0f / kD);
        const float4 d0 = f4_sub(v0, mean), d1 = f4_sub(v1, mean);
        const float var = group16_sum(sum4(f4_mul(d0, d0)) + sum4(f4_mul(d1, d1))) * (1.0f / kD);
        const float rstd = 1.0f / sqrtf(var + kLnEps);
        const float4 x0 = f4_scale(d0, rstd), x1 = f4_scale(d1, rstd);
        const float4 y0 = f4_fma(x0, lp.g0, lp.b0), y1 = f4_fma(x1, lp.g1, lp.b1);
        *reinterpret_cast<float4*>(dst + lrow * LD + lsub * 8) = y0;
        *reinterpret_cast<float4*>(dst + lrow * LD + lsub * 8 + 4) = y1;
        if (lrow < T) {
            const size_t og = (size_t)(row0 + lrow) * kD + lsub * 8;
            if (xhat_g) {
                SAVE_ST4(xhat_g + og, x0);
                SAVE_ST4(xhat_g + og + 4, x1);
                if (lsub == 0) stg(rstd_g + row0 + lrow, rstd);
            }
            if (y_g) {
                stg4(y_g + og, y0);
                stg4(y_g + og + 4, y1);
            }
        }
    };
    const auto qidx = [](int u, int j) { return (16 * (u >> 3) + j) * 8 + (u & 7); };      // chunk u / 8, k-block u % 8, tile j of the head
    constexpr int kPre = CF_TQ_PRE, kRingQ = CF_TQ_PRE + 1;      // projection ring: units in flight ahead of the matrix pipe
    const int wq = (2 * hh * 8 + (role_v ? 32 * 8 : 0)) * 256;      // this wave's two chunks: q | k or v | gate of head hh
    BBuf<16, kRingQ> rq;
    b_issue<0, kPre>(rq, load_layer(tab).watt_t + wq, wl, qidx);
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l) {
        const RegLayerDev P = load_layer(tab + l);
        const float gam = ldg(P.gamma + hh);
        float* hq = P.hq + hq0;
        CF_STAMPQ(0);
        // ---- two projection chunks of head hh: 2 x (2 tiles x K = 128), one ring over the 16 units
        f32x4 vacc[2], gacc[2];
        {
            const float* qb = P.watt_t + wq;
            const float* ap = xs + oALD;
            float4 an = lds4(ap);
            f32x4 pacc[2];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u + kPre < 16) {
                    rq.s[(u + kPre) % kRingQ][0] = ldg_blk(qb, wl, qidx(u + kPre, 0));
                    rq.s[(u + kPre) % kRingQ][1] = ldg_blk(qb, wl, qidx(u + kPre, 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                if ((u & 7) == 0) zero_acc(pacc);
                const float4 av = an;
                an = lds4(ap + ((u + 1) & 7) * 16);
                mma_unit(av, av, rq.s[u % kRingQ][0], rq.s[u % kRingQ][1], pacc[0], pacc[1]);
                if ((u & 7) == 7) {
                    const int c = u >> 3;
                    if (!role_v) {      // q (c = 0), k (c = 1): rows into the head's patch, transposed tiles to global
                        float* dst = (c == 0 ? qs : ks) + oD36;
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
#pragma unroll
                            for (int ii = 0; ii < 4; ++ii) dst[ii * 36 + t * 16] = pacc[t][ii];
                            float* tp_ = lane_at(sbase(hq, (c == 0 ? kHqQ : kHqK) + t * 256), bT);
                            if (SAVE && rok[0]) SAVE_ST4(tp_, acc4(pacc[t]));
                        }
                    } else if (c == 0) {      // v: stays in registers (B operand of p v); rows to global for the backward
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            vacc[t] = pacc[t];
                            if (SAVE) {
                                float* vp_ = lane_at(sbase(hq, kHqV), bV);
#pragma unroll
                                for (int ii = 0; ii < 4; ++ii)
                                    if (rok[ii]) SAVE_ST(vp_ + ii * 32 + t * 16, pacc[t][ii]);
                            }
                        }
                    } else {                  // gate
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            float* tp_ = lane_at(sbase(hq, kHqG + t * 256), bT);
                            if (SAVE && rok[0]) SAVE_ST4(tp_, acc4(pacc[t]));
#pragma unroll
                            for (int ii = 0; ii < 4; ++ii) gacc[t][ii] = fast_sigmoid(pacc[t][ii]);      // (the gate's sigmoid: formed while the head's other wave is at its softmax)
                        }
                    }
                }
            }
        }
        // out-projection operands of this wave: output tiles 2 w, 2 w + 1, k-blocks 4 mj .. 4 mj + 3 of K = 256 (requested under the attention)
        float4 bo_[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) bo_[t][i] = ldg_blk(P.wo_t, wl, (2 * w + t) * 16 + 4 * mj + i);
        CF_STAMPQ(1);
        if (!role_v) {      // scores, mask, softmax of the head (modules.py:58-72) -> p in the head's patch
            wave_lds_sync();
            f32x4 sc[2];
            zero_acc(sc);
            mma_unit(lds4(qs + oA36), lds4(qs + oA36 + 16), lds4(ks + oA36), lds4(ks + oA36 + 16), sc[0], sc[1]);
            float p[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                float sv = (sc[0][ii] + sc[1][ii]) / scale + gam * fqv[ii];
                if (mkv[ii]) sv = kMaskFill;
                if (lr >= T) sv = -INFINITY;
                const float m = group16_max(sv);
                const float e = lr < T ? __expf(sv - m) : 0.f;
                const float z = group16_sum(e);
                p[ii] = rok[ii] ? e * __builtin_amdgcn_rcpf(z) : 0.f;
                ps[oD20 + ii * 20] = p[ii];
            }
            float* pp_ = lane_at(sbase(hq, kHqP), bT);
            if (SAVE && rok[0] && lr < T) SAVE_ST4(pp_, make_float4(p[0], p[1], p[2], p[3]));
        }
        CF_STAMPQ(2);
        __syncthreads();
        CF_STAMPQ(3);
        if (role_v) {       // o = p v, gated (modules.py:74-81) -> the member's attention tile
            const float4 pa = lds4(ps + oA20);
            f32x4 o[2];
            zero_acc(o);
            mma_unit(pa, pa, acc4(vacc[0]), acc4(vacc[1]), o[0], o[1]);
            float* ap_ = lane_at(sbase(P.a, row0 * kRDm + hh * 32), zA);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const float val = o[t][ii] * gacc[t][ii];
                    if (rok[ii]) {
                        as_[(lq * 4 + ii) * LA + hl * 32 + t * 16 + lr] = val;
                        if (SAVE) SAVE_ST(ap_ + ii * kRDm + t * 16, val);
                    }
                }
        }
        __syncthreads();
        CF_STAMPQ(4);
        // ---- out-projection over this member's 64 attention columns: partial tile -> slot, team sum, + bias + residual, LayerNorm
        {
            f32x4 acc[2];
            zero_acc(acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 av = lds4(as_ + oALA + i * 16);
                mma_unit(av, av, bo_[0][i], bo_[1][i], acc[0], acc[1]);
            }
            float* sp = lane_at(sbase(uslots, ((n_ex & 1) * 4 + mj) * kTqSlot), pD);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    if (rok[ii]) stg(sp + ii * kD + t * 16, acc[t][ii]);
        }
        CF_STAMPQ(5);
        team_arrive(ucnt);
        // FFN 1 operands of this wave: hidden tile 4 mj + w, K = 128; the LayerNorm's parameters (requested under the wait)
        float4 b1_[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) b1_[i] = ldg_blk(P.w1_t, wl, (4 * mj + w) * 8 + i);
        const float b1v = ldg(lane_at(P.b1 + mj * 64 + w * 16, lr * 4));
        const LnParams ln1 = ln_params_load(P.g1, P.be1);
        team_wait(ucnt, 4 * (n_ex + 1));
        CF_STAMPQ(6);
        team_sum_ln(n_ex & 1, P.bo, xs, ts, ln1, SAVE && mj == 0 ? P.xh1 : nullptr, P.rs1, SAVE && mj == 0 ? P.y1 : nullptr);
        ++n_ex;
        CF_STAMPQ(7);
        __syncthreads();
        CF_STAMPQ(8);
        // ---- FFN (modules.py:100-101): this member's 64 hidden columns, then the K-split second product
        float4 b2_[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) b2_[t][i] = ldg_blk(P.w2_t, wl, (2 * w + t) * 16 + 4 * mj + i);
        {
            f32x4 acc[2];
            zero_acc(acc);
#pragma unroll
            for (int u = 0; u < 4; ++u) mma_unit(lds4(ts + oALD + (2 * u) * 16), lds4(ts + oALD + (2 * u + 1) * 16), b1_[2 * u], b1_[2 * u + 1], acc[0], acc[1]);
            float* hp = lane_at(sbase(P.hdn, row0 * DFF + mj * 64 + w * 16), zH);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const float hv = fmaxf((acc[0][ii] + acc[1][ii]) + b1v, 0.f);
                hs[(lq * 4 + ii) * LA + w * 16 + lr] = hv;
                if (SAVE && rok[ii]) SAVE_ST(hp + ii * DFF, hv);
            }
        }
        __syncthreads();
        CF_STAMPQ(9);
        {
            f32x4 acc[2];
            zero_acc(acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 av = lds4(hs + oALA + i * 16);
                mma_unit(av, av, b2_[0][i], b2_[1][i], acc[0], acc[1]);
            }
            float* sp = lane_at(sbase(uslots, ((n_ex & 1) * 4 + mj) * kTqSlot), pD);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    if (rok[ii]) stg(sp + ii * kD + t * 16, acc[t][ii]);
        }
        CF_STAMPQ(10);
        team_arrive(ucnt);
        // the next layer's projection ring and the LayerNorm's parameters (requested under the wait)
        const bool more = l + 1 < a.n_layers;
        if (more) b_issue<0, kPre>(rq, load_layer(tab + l + 1).watt_t + wq, wl, qidx);
        const LnParams ln2 = ln_params_load(P.g2, P.be2);
        team_wait(ucnt, 4 * (n_ex + 1));
        CF_STAMPQ(11);
        team_sum_ln(n_ex & 1, P.b2, ts, xs, ln2, SAVE && mj == 0 ? P.xh2 : nullptr, P.rs2, mj == 0 ? P.xout : nullptr);
        ++n_ex;
        CF_STAMPQ(12);
        __syncthreads();
        CF_STAMPQ(13);
    }
    if (a.tdbg && tid == 0 && b < 464) a.tdbg[97 + 2 * b] = __builtin_amdgcn_s_memtime();
    // the last member to leave resets the team's counters for the next launch (everybody is past its last wait by then)
    if (tid == 0) {
        const int d = __hip_atomic_fetch_add(ucnt + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == 3) {      // (one departure per member)
            __hip_atomic_store(ucnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ucnt + 16, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Which XCD does workgroup i of a launch land on?  (cf_create: the team kernel is enabled only where the answer is i mod 8)
__global__ void k_xcc_probe(unsigned* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v & 15;
}

}  // namespace cf

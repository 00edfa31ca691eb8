// cf_reg8.h -- the Regulation stack (net.py:142-153) as two launches of 512-thread workgroups (included by cf_kernels.h).
//
// One workgroup per (gene, resolution) walks ALL layers; the gene's T <= 16 tokens are one MFMA row tile that stays in
// LDS from layer to layer.  Eight waves = two per SIMD: while one wave of a pair waits for an operand, a barrier or an
// LDS round trip, the other one issues its MFMAs (one wave per SIMD left the matrix pipes 46 % / 32 % busy).
//
//   * wave w owns HEAD w: it computes the q | k | v | gate columns of its head (4 x 32 of the 1024 projection columns),
//     so the whole attention of a head -- scores, mask, softmax, p v, gate -- and its backward are wave-local: the only
//     exchange is a D-layout -> A-layout transpose through a wave-private LDS patch (no workgroup barrier), v and the
//     gate never leave the accumulator registers (the D layout of v IS the B operand of p v).
//   * every weight is read through a tiled copy whose 16 x 16 blocks are 1 KB contiguous in fragment order (k_retile):
//     the forward products read the tiling of W, the backward (dX = dY W) products the tiling of W^T.  A "unit" of the
//     operand stream is two such blocks = 8 MFMAs; a product requests all its units (64 registers) one product ahead,
//     the two long products (K = 128 x 4 chunks forward, K = 1024 backward) run an 8-unit ring 7 units ahead.
//   * what the backward needs of the attention is saved per (gene, head) in the layouts its operands want: q^T, k^T,
//     gate^T as transposed tiles (= the forward's accumulator registers, one float4 of four tokens per lane), v as
//     [16][32] rows, p^T as a [16][16] tile.  The tiles are quarter-major -- [token quarter][column][4 tokens], 256
//     contiguous bytes per quarter -- and only quarters / rows that hold a token are written or read back (T = 9: 6.3 KB
//     of the 9 KB block, whole 64-byte pieces): the forward is sensitive to its store volume (37 MB more per launch cost
//     10 us: 3.7 TB/s at the margin) and the backward waits for these reads, which every workgroup of the launch issues
//     in the same microsecond (tools/reg_stamps.py bwd: the two LayerNorm phases).
//   * bias / LayerNorm gradients are column sums of arrays the weight-gradient launch needs anyway (dt2, dpre1, dt1,
//     dy1, d(layer output)): k_colsum takes them from there, nothing is reduced inside this kernel.
//
// Reference math: modules.py:28-88 (self-attention block with gate and frequency bias), 100-101 (FeedForward),
// 121-124 (layer stack).  Six workgroup barriers per layer and direction.
#pragma once
#ifndef CF_STAGGER      // 1: waves 4..7 run the attention BEFORE the gate chunk (opposite to their SIMD partner).  Measured: no change
#define CF_STAGGER 0    // (116.8 vs 117.0 us, three interleaved rounds) -- the two waves of a SIMD drift apart on their own.
#endif
// The activations saved for the backward pass are written once and read once, 100+ us later: as plain accesses they sweep the L2 that
// the weight tiles of the layer stream through (every workgroup of an XCD reads the same 0.9 MB per layer); as streaming (nt) stores /
// loads they do not.  CF_SAVE_NT=0: plain.
#ifndef CF_SAVE_NT      // bits: 1 forward saves of the attention operands / a / hdn, 2 their loads in the backward, 4 the LayerNorm saves,
#define CF_SAVE_NT 11   //       8 the backward's row-level outputs, 16 the backward's LayerNorm operand loads.  Measured (ms per step, two A/B
#endif                  //       rounds): 0 0.561, 1 0.553, 2 0.550, 3 0.545 / 0.543, 7 0.545, 11 0.5405, 19 0.544, 31 0.542
#if CF_SAVE_NT & 1
#define SAVE_ST stg_nt
#define SAVE_ST4 stg4_nt
#else
#define SAVE_ST stg
#define SAVE_ST4 stg4
#endif
#if CF_SAVE_NT & 2
#define SAVE_LD ldg_nt
#define SAVE_LD4 ldg4_nt
#else
#define SAVE_LD ldg
#define SAVE_LD4 ldg4
#endif
#if CF_SAVE_NT & 8
#define BWD_ST stg_nt
#define BWD_ST4 stg4_nt
#else
#define BWD_ST stg
#define BWD_ST4 stg4
#endif
#if CF_SAVE_NT & 16
#define BWD_LD4 ldg4_nt
#else
#define BWD_LD4 ldg4
#endif
#ifndef CF_QPRE
#define CF_QPRE 7
#endif

namespace cf {

constexpr int kHqQ = 0, kHqK = 512, kHqG = 1024, kHqV = 1536, kHqP = 2048, kHqFloats = 2304;      // saved block of one (gene, head)
constexpr int kPatchF = 16 * 36 * 2 + 16 * 20;      // forward patch of a wave: q rows, k rows, p
constexpr int kPatchB = 16 * 36 + 16 * 20;          // backward patch: do rows, ds

__host__ __device__ constexpr size_t reg8_fwd_smem(int dff) {
    return (size_t)(2 * kTile * (kD + 4) + kTile * (kRDm + 4) + kTile * (dff + 4) + 8 * kPatchF) * sizeof(float);
}
__host__ __device__ constexpr size_t reg8_bwd_smem(int dff) {
    return (size_t)(2 * kTile * (kD + 4) + kTile * (dff + 4) + kTile * kQkLd + 8 * kPatchB) * sizeof(float);
}

__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 f4z() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// LDS operations of one wave execute in order; this only keeps the compiler from moving a read above the write it depends on
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// 8 MFMAs on two independent accumulators: (a0, b0) -> c0, (a1, b1) -> c1.  Lane (r, q) supplies A[row r][k = 4q + m] and
// B[k = 4q + m][col r] to the m-th instruction of a 16-deep block.
__device__ __forceinline__ void mma_unit(const float4& a0, const float4& a1, const float4& b0, const float4& b1, f32x4& c0, f32x4& c1) {
    c0 = mfma4(a0.x, b0.x, c0);
    c1 = mfma4(a1.x, b1.x, c1);
    c0 = mfma4(a0.y, b0.y, c0);
    c1 = mfma4(a1.y, b1.y, c1);
    c0 = mfma4(a0.z, b0.z, c0);
    c1 = mfma4(a1.z, b1.z, c1);
    c0 = mfma4(a0.w, b0.w, c0);
    c1 = mfma4(a1.w, b1.w, c1);
}
__device__ __forceinline__ float4 acc4(const f32x4& c) { return make_float4(c[0], c[1], c[2], c[3]); }

// Global addressing as (scalar base + compile-time constant) + 32-bit per-lane byte offset.  The constant is folded into
// the SCALAR base (two SALU instructions) behind an opaque barrier, so the access is `global_load v, v_off, s[base]`:
// left to itself the compiler folds constants into the per-lane part instead, materialises one 64-bit VGPR address per
// distinct constant and hoists them all out of the layer loop (128 registers for the 64 blocks of one operand stream).
__device__ __forceinline__ const float* sbase(const float* p, int elems) {
    const float* q = p + elems;
    asm("" : "+s"(q));
    return q;
}
__device__ __forceinline__ float* sbase(float* p, int elems) {
    float* q = p + elems;
    asm("" : "+s"(q));
    return q;
}
// (the lane offset passes through an opaque barrier too: instruction selection works per basic block and only folds a
// zero-extended 32-bit offset it can SEE -- a zero-extension hoisted out of the loop makes it fall back to 64-bit adds)
__device__ __forceinline__ const float* lane_at(const float* sb, unsigned vbyte) {
    asm("" : "+v"(vbyte));
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(sb) + vbyte);
}
__device__ __forceinline__ float* lane_at(float* sb, unsigned vbyte) {
    asm("" : "+v"(vbyte));
    return reinterpret_cast<float*>(reinterpret_cast<char*>(sb) + vbyte);
}
// block i (1 KB) of a tiled weight stream: 4 KB groups go into the scalar base, the rest into the instruction's immediate
__device__ __forceinline__ float4 ldg_blk(const float* base, unsigned vbyte, int i) { return ldg4(lane_at(sbase(base, (i >> 2) * 1024), vbyte) + (i & 3) * 256); }

// One row of the layer table through the CONSTANT address space: scalar loads whatever stores precede them (a plain
// global load after a store is not provably unclobbered, the compiler then fetches the row with vector loads and every
// pointer in it lives in VGPRs).
#define CF_CONST __attribute__((address_space(4)))
__device__ __forceinline__ RegLayerDev load_layer(const RegLayerDev* p) {
    struct Words {
        unsigned long long w[sizeof(RegLayerDev) / 8];
    } t;
    const CF_CONST unsigned long long* s = (const CF_CONST unsigned long long*)p;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(RegLayerDev) / 8); ++i) t.w[i] = s[i];
    return __builtin_bit_cast(RegLayerDev, t);
}

// Operand stream of one product: NU units of two 1 KB blocks, ring of R units.  IDX(u, j): block index of the j-th block
// of unit u relative to the wave's base (compile-time after unrolling).
template <int NU, int R>
struct BBuf {
    float4 s[R][2];
};
template <int U0, int U1, int NU, int R, class IDX>
__device__ __forceinline__ void b_issue(BBuf<NU, R>& b, const float* base, unsigned vb, IDX idx) {
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        b.s[u % R][0] = ldg_blk(base, vb, idx(u, 0));
        b.s[u % R][1] = ldg_blk(base, vb, idx(u, 1));
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int NU, int R, class IDX>
__device__ __forceinline__ void b_issue1(BBuf<NU, R>& b, const float* base, unsigned vb, IDX idx, int u) {      // u: compile-time after unrolling
    b.s[u % R][0] = ldg_blk(base, vb, idx(u, 0));
    b.s[u % R][1] = ldg_blk(base, vb, idx(u, 1));
}
// TWO = true: a unit is one 16-deep k-block of two column tiles (acc[0], acc[1] = the two tiles);
// TWO = false: two consecutive k-blocks of one column tile (acc[0] + acc[1] (+ acc[2] + acc[3]) = the tile).
// Units [0, PRE) were requested by the caller; ap = A tile + r * lda + 4 q.  co(u) runs before the MFMAs of unit u: the
// requests of the NEXT product's operand stream are spread over this product's units (a burst of 16 one-KB loads per wave
// keeps the CU's 64 B/clk load path -- and every wave queued behind it -- busy for 2 K cycles).
struct NoCo {
    __device__ __forceinline__ void operator()(int) const {}
};
template <int PRE, bool TWO, int NACC, int NU, int R, class IDX, class CO = NoCo>
__device__ __forceinline__ void b_run(BBuf<NU, R>& b, const float* base, unsigned vb, IDX idx, const float* ap, f32x4 (&acc)[NACC], CO co = CO()) {
    static_assert(PRE <= R && (PRE == NU || PRE < R), "ring too small for the prefetch distance");
    float4 a0n = lds4(ap), a1n = TWO ? a0n : lds4(ap + 16);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (u + PRE < NU) {
            b.s[(u + PRE) % R][0] = ldg_blk(base, vb, idx(u + PRE, 0));
            b.s[(u + PRE) % R][1] = ldg_blk(base, vb, idx(u + PRE, 1));
        }
        co(u);
        __builtin_amdgcn_sched_barrier(0);      // keep the request ahead of this unit's MFMAs (the scheduler would sink it)
        const float4 a0 = a0n, a1 = TWO ? a0n : a1n;
        if (u + 1 < NU) {
            if (TWO) {
                a0n = lds4(ap + (u + 1) * 16);
            } else {
                a0n = lds4(ap + (2 * u + 2) * 16);
                a1n = lds4(ap + (2 * u + 3) * 16);
            }
        }
        const int k = NACC == 4 ? (u & 1) * 2 : 0;
        mma_unit(a0, a1, b.s[u % R][0], b.s[u % R][1], acc[k], acc[k + 1]);
    }
}

// ---- The LAST layer (RegArgs::row0_last).  The head reads token 0 of the stack's output and nothing else (net.py:375-376), so in the
// last layer only the keys and values need all T rows: its query, gate, out-projection, FFN and both LayerNorms are needed for row 0
// alone, and in the backward pass every gradient above the attention has one live row.  A 16-row MFMA tile costs the same for one row
// as for sixteen; on the vector ALUs a one-row product is 4 FMAs per 1 KB weight block and lane: lane (r, q) of a tiled block holds
// W[n0 + r][16 kb + 4 q ..], so y[n0 + r] = sum over the blocks of the lane's 4-term dot products, summed over the four lanes r, r + 16,
// r + 32, r + 48 at the end.  Same operand stream, same ring, same epilogues (the result goes into row 0 of an otherwise zero
// accumulator tile): ~330 of a wave's ~470 MFMAs of that layer disappear in either direction, its 0.9 MB weight stream stays.
// Exact: what is dropped is never read (forward) or multiplied by an exact zero (backward).
__device__ __forceinline__ float dot4(const float4& w, const float4& x, float acc) {
    acc = fmaf(w.x, x.x, acc);
    acc = fmaf(w.y, x.y, acc);
    acc = fmaf(w.z, x.z, acc);
    return fmaf(w.w, x.w, acc);
}
__device__ __forceinline__ float quarters_sum(float v) {      // over the lanes r, r + 16, r + 32, r + 48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ f32x4 row0_acc(float v) {      // accumulator layout: lane (r, q = 0), register 0 = row 0, column r
    f32x4 c = {(threadIdx.x & 48) == 0 ? v : 0.f, 0.f, 0.f, 0.f};
    return c;
}
// b_run for ONE live row: a0 = row 0 of the A tile + 4 q.  Result in acc[0] (TWO: acc[0], acc[1] = the two column tiles), the rest zero.
template <int PRE, bool TWO, int NACC, int NU, int R, class IDX, class CO = NoCo>
__device__ __forceinline__ void b_run_row0(BBuf<NU, R>& b, const float* base, unsigned vb, IDX idx, const float* a0, f32x4 (&acc)[NACC], CO co = CO()) {
    static_assert(PRE <= R && (PRE == NU || PRE < R), "ring too small for the prefetch distance");
    float p0 = 0.f, p1 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (u + PRE < NU) {
            b.s[(u + PRE) % R][0] = ldg_blk(base, vb, idx(u + PRE, 0));
            b.s[(u + PRE) % R][1] = ldg_blk(base, vb, idx(u + PRE, 1));
        }
        co(u);
        __builtin_amdgcn_sched_barrier(0);
        const float4 x0 = lds4(a0 + (TWO ? u : 2 * u) * 16);
        p0 = dot4(b.s[u % R][0], x0, p0);
        p1 = dot4(b.s[u % R][1], TWO ? x0 : lds4(a0 + (2 * u + 1) * 16), p1);
    }
    zero_acc(acc);
    if (TWO) {
        acc[0] = row0_acc(quarters_sum(p0));
        acc[1] = row0_acc(quarters_sum(p1));
    } else {
        acc[0] = row0_acc(quarters_sum(p0 + p1));
    }
}
// The K = 1024 input-gradient product of the last layer: d(q | k | v | gate) has all rows in its k and v quarters (units 8 .. 23: matrix
// cores) and row 0 only in its q and gate quarters (units 0 .. 7, 24 .. 31: vector ALUs).  ap = A tile + r * lda + 4 q, a0 = row 0 + 4 q.
template <int PRE, int NU, int R, class IDX>
__device__ __forceinline__ void b_run_kv_rows(BBuf<NU, R>& b, const float* base, unsigned vb, IDX idx, const float* ap, const float* a0, f32x4 (&acc)[4]) {
    static_assert(NU == 32 && PRE < R, "q | k | v | gate: eight units each");
    float p0 = 0.f, p1 = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (u + PRE < NU) {
            b.s[(u + PRE) % R][0] = ldg_blk(base, vb, idx(u + PRE, 0));
            b.s[(u + PRE) % R][1] = ldg_blk(base, vb, idx(u + PRE, 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (u < 8 || u >= 24) {
            p0 = dot4(b.s[u % R][0], lds4(a0 + (2 * u) * 16), p0);
            p1 = dot4(b.s[u % R][1], lds4(a0 + (2 * u + 1) * 16), p1);
        } else {
            const int k = (u & 1) * 2;
            mma_unit(lds4(ap + (2 * u) * 16), lds4(ap + (2 * u + 1) * 16), b.s[u % R][0], b.s[u % R][1], acc[k], acc[k + 1]);
        }
    }
    const float v = quarters_sum(p0 + p1);
    if ((threadIdx.x & 48) == 0) acc[0][0] += v;
}

// LayerNorm backward of the 16-row tile, 16 lanes per row (waves 0..3), xhat / rstd / gamma of the lane in registers:
// dst = rstd * (a - mean(a) - xhat * mean(a * xhat)),  a = dy * g
__device__ __forceinline__ void ln_bwd_tile16_r(const float* src, float* dst, int ld, float4 x0, float4 x1, float4 g0, float4 g1, float rs,
                                                bool live, float* dx_lane) {
    const int lane = threadIdx.x & 63, row = (threadIdx.x >> 6) * 4 + (lane >> 4), sub = lane & 15;
    const float* sp = src + row * ld + sub * 8;
    const float4 a0 = f4_mul(lds4(sp), g0), a1 = f4_mul(lds4(sp + 4), g1);
    const float m1 = group16_sum(sum4(a0) + sum4(a1)) * (1.0f / kD);
    const float m2 = group16_sum(sum4(f4_mul(a0, x0)) + sum4(f4_mul(a1, x1))) * (1.0f / kD);
    const float4 o0 = make_float4(rs * (a0.x - m1 - x0.x * m2), rs * (a0.y - m1 - x0.y * m2), rs * (a0.z - m1 - x0.z * m2), rs * (a0.w - m1 - x0.w * m2));
    const float4 o1 = make_float4(rs * (a1.x - m1 - x1.x * m2), rs * (a1.y - m1 - x1.y * m2), rs * (a1.z - m1 - x1.z * m2), rs * (a1.w - m1 - x1.w * m2));
    float* dp = dst + row * ld + sub * 8;
    *reinterpret_cast<float4*>(dp) = o0;
    *reinterpret_cast<float4*>(dp + 4) = o1;
    if (live) {
        BWD_ST4(dx_lane, o0);
        BWD_ST4(dx_lane + 4, o1);
    }
}

#define CF_STAMP8(slot)                                                                     \
    do {                                                                                    \
        if (a.tdbg && g == 0 && r == 0 && l < 6 && lane == 0) a.tdbg[(w * 6 + l) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

template <int DFF, bool SAVE>
__global__ __launch_bounds__(512) void k_reg8_fwd(RegArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own 4 MB L2.  XCD x
    // takes the contiguous slice [x * per, (x + 1) * per) of the (resolution-major) sequence list, so six XCDs stream
    // the weights of one resolution (5.5 MB) and two of them those of two, instead of all eight streaming all three.
    if (SAVE && a.head.on && blockIdx.x == 0) head_ride_begin(a.head, a.B);
    const int per = gridDim.x >> 3, v = a.xcd_map ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : (int)blockIdx.x;
    if (v >= a.B * a.n_res) return;
    const int g = __builtin_amdgcn_readfirstlane(v % a.B), r = __builtin_amdgcn_readfirstlane(v / a.B), T = a.T, TT = T * T, row0 = g * T, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    constexpr int LD = kD + 4, LW = kRDm + 4, LH = DFF + 4;
    constexpr int NU1 = DFF == 256 ? 8 : 4, NU2 = DFF / 32;      // units of the two FFN products
    constexpr int HW = DFF == 256 ? 32 : 16;                     // FFN hidden columns per wave
    constexpr int kQPre = CF_QPRE;                               // units the projection ring runs ahead
    float* xs = smem;                    // [16][LD]   layer input / output
    float* ts = xs + kTile * LD;         // [16][LD]
    float* as_ = ts + kTile * LD;        // [16][LW]   gated attention output
    float* hs = as_ + kTile * LW;        // [16][LH]   FFN hidden
    float* qs = hs + kTile * LH + w * kPatchF;      // wave-private: q rows [16][36]
    float* ks = qs + 16 * 36;                       //               k rows [16][36]
    float* ps = ks + 16 * 36;                       //               p      [16][20]
    const RegLayerDev* tab = a.tab + (size_t)r * a.n_layers;
    const float scale = sqrtf((float)kRDh);
    {
        const float* x0 = load_layer(tab).xin + (size_t)row0 * kD;
        for (int i = tid; i < kTile * (kD / 4); i += 512) {
            const int row = i >> 5, c4 = i & 31;
            float4 t = f4z();
            if (row < T) t = ldg4(x0 + row * kD + c4 * 4);
            *reinterpret_cast<float4*>(xs + row * LD + c4 * 4) = t;
        }
        for (int i = tid; i < kTile * LW; i += 512) as_[i] = 0.f;      // rows >= T stay zero (MFMA operand)
    }
    // Per-lane addressing, computed once.  "D" offsets address the accumulator layout (lane (lr, lq), register ii = row
    // 4 lq + ii, column lr of a 16-column tile), "A" offsets the operand layout (row lr, 4 consecutive k at 4 lq): every
    // access below is one of these bases plus a compile-time constant.
    const int oD36 = lq * 4 * 36 + lr, oA36 = lr * 36 + lq * 4, oD20 = lq * 4 * 20 + lr, oA20 = lr * 20 + lq * 4;
    const int oDLD = lq * 4 * LD + w * 16 + lr, oALD = lr * LD + lq * 4;
    const int oDLW = lq * 4 * LW + w * 32 + lr, oALW = lr * LW + lq * 4;
    const int oDLH = lq * 4 * LH + w * HW + lr, oALH = lr * LH + lq * 4;
    // global: per-lane BYTE offsets (the wave / gene part of an address is scalar and goes into the base, see sbase())
    const int hq0 = (g * kRH + w) * kHqFloats;                       // (gene, head) block of the saved attention operands
    const unsigned bT = (lq * 64 + lr * 4) * 4;                      // transposed tiles / p^T, one float4 per lane, quarter-major (see header)
    const unsigned bV = (lq * 4 * 32 + lr) * 4;                      // v rows
    const unsigned zA = (lq * 4 * kRDm + lr) * 4, zH = (lq * 4 * DFF + lr) * 4;
    const unsigned wl = lane * 16;
    bool rok[4];
    // interaction frequencies / mask of the gene in the score layout: lane (j = lr, lq) holds rows i = 4 lq + ii
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
    const auto qidx = [](int u, int j) { return (16 * (u >> 3) + j) * 8 + (u & 7); };      // chunk u / 8, k-block u % 8, tile j of the head
    const auto idx_two = [](int u, int j) { return j * 8 + u; };                           // two tiles (K = 128), k-block u
    const auto idx_one = [](int u, int j) { return 2 * u + j; };                           // one tile, k-blocks 2u, 2u + 1
    const int wq = 2 * w * 8 * 256, wo16 = w * 16 * 256, w1o = (DFF == 256 ? 2 * w : w) * 8 * 256, w2o = w * (DFF / 16) * 256;      // wave's blocks
    BBuf<32, 8> rq;
    b_issue<0, kQPre>(rq, load_layer(tab).watt_t + wq, wl, qidx);
    __syncthreads();
    // (the layer as a generic lambda instantiated twice -- all rows / the last layer's row 0 -- instead of run-time branches in one body:
    // with both forms alive in one loop body the 241-register kernel spilled 46 VGPRs)
    auto layer = [&](auto r0_c, const int l) __attribute__((always_inline)) {
        constexpr bool r0 = decltype(r0_c)::value;      // the last layer: row 0 only (see b_run_row0)
        const RegLayerDev P = load_layer(tab + l);      // by value: the pointers live in SGPRs
        CF_STAMP8(0);
        const float gam = ldg(P.gamma + w);
        const float bov = ldg(lane_at(P.bo + w * 16, lr * 4)), b2v = ldg(lane_at(P.b2 + w * 16, lr * 4));
        float b1v[2];
        b1v[0] = ldg(lane_at(P.b1 + w * HW, lr * 4));
        b1v[1] = DFF == 256 ? ldg(lane_at(P.b1 + w * HW, lr * 4) + 16) : 0.f;
        float* hq = P.hq + hq0;
        float* ag = P.a + row0 * kRDm + w * 32;
        float* hg = P.hdn + row0 * DFF + w * HW;
        // ---- q | k | v | gate of head w: 4 chunks x (2 tiles x K = 128), one ring over all 32 units; then the attention of the
        //      head (modules.py:58-81), wave-local.  (CF_STAGGER: the two waves of a SIMD, w and w + 4, can run the gate chunk
        //      and the attention in opposite orders; see the macro for the measurement.)
        f32x4 vacc[2], gacc[2], o[2];
        BBuf<8, 8> bwo;
        const float* qb = P.watt_t + wq;
        const float* ap = xs + oALD;
        float4 an = lds4(ap);
        f32x4 pacc[2];
        float pr0 = 0.f, pr1 = 0.f;
        auto units = [&](auto u0_c, auto u1_c) {
            constexpr int U0 = decltype(u0_c)::value, U1 = decltype(u1_c)::value;
#pragma unroll
            for (int u = U0; u < U1; ++u) {
                if (u + kQPre < 32) {
                    rq.s[(u + kQPre) % 8][0] = ldg_blk(qb, wl, qidx(u + kQPre, 0));
                    rq.s[(u + kQPre) % 8][1] = ldg_blk(qb, wl, qidx(u + kQPre, 1));
                }
                if (u >= 24) b_issue1(bwo, P.wo_t + wo16, wl, idx_one, u - 24);      // out-projection weights, spread over the gate chunk
                __builtin_amdgcn_sched_barrier(0);
                if ((u & 7) == 0) {
                    zero_acc(pacc);
                    pr0 = pr1 = 0.f;
                }
                const float4 av = an;
                an = lds4(ap + ((u + 1) & 7) * 16);
                const bool one = r0 && ((u >> 3) == 0 || (u >> 3) == 3);      // q / gate of the last layer: row 0 on the vector ALUs
                if (one) {
                    const float4 x0 = lds4(xs + lq * 4 + (u & 7) * 16);
                    pr0 = dot4(rq.s[u % 8][0], x0, pr0);
                    pr1 = dot4(rq.s[u % 8][1], x0, pr1);
                } else {
                    mma_unit(av, av, rq.s[u % 8][0], rq.s[u % 8][1], pacc[0], pacc[1]);
                }
                if ((u & 7) == 7) {
                    const int c = u >> 3;
                    if (one) {
                        pacc[0] = row0_acc(quarters_sum(pr0));
                        pacc[1] = row0_acc(quarters_sum(pr1));
                    }
#ifdef CF_STAMP_CHUNKS      // (tools/reg_stamps.py: ends of the q, k, v chunks)
                    if (c < 3) CF_STAMP8(9 + c);
#endif
                    if (c < 2) {            // q, k: row-major into the wave's patch (operands of the scores), transposed tiles to global
                        float* dst = (c == 0 ? qs : ks) + oD36;
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
#pragma unroll
                            for (int ii = 0; ii < 4; ++ii) dst[ii * 36 + t * 16] = pacc[t][ii];
                            float* tp_ = lane_at(sbase(hq, (c == 0 ? kHqQ : kHqK) + t * 256), bT);      // (scalar bases are formed outside the conditionals)
                            if (SAVE && rok[0]) SAVE_ST4(tp_, acc4(pacc[t]));
                        }
                    } else if (c == 2) {    // v: stays in registers (B operand of p v); rows to global for the backward
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
                    } else {                // gate
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            gacc[t] = pacc[t];
                            float* tp_ = lane_at(sbase(hq, kHqG + t * 256), bT);
                            if (SAVE && rok[0]) SAVE_ST4(tp_, acc4(pacc[t]));
                        }
                    }
                }
            }
        };
        auto attend = [&]() {      // scores, mask, softmax, p v of the head -> o
            wave_lds_sync();
            f32x4 sc[2];
            zero_acc(sc);
            mma_unit(lds4(qs + oA36), lds4(qs + oA36 + 16), lds4(ks + oA36), lds4(ks + oA36 + 16), sc[0], sc[1]);
            float p[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {      // lane (j = lr, lq): score of row i = 4 lq + ii
                float sv = (sc[0][ii] + sc[1][ii]) / scale + gam * fqv[ii];
                if (mkv[ii]) sv = kMaskFill;
                if (lr >= T) sv = -INFINITY;
                const float m = group16_max(sv);
                const float e = lr < T ? __expf(sv - m) : 0.f;      // v_exp_f32: 1 ulp; the logits keep two orders of headroom to 1e-4
                const float z = group16_sum(e);
                p[ii] = rok[ii] ? e * __builtin_amdgcn_rcpf(z) : 0.f;
                ps[oD20 + ii * 20] = p[ii];
            }
            float* pp_ = lane_at(sbase(hq, kHqP), bT);
            if (SAVE && rok[0] && lr < T) SAVE_ST4(pp_, make_float4(p[0], p[1], p[2], p[3]));      // p^T tile
            wave_lds_sync();
            const float4 pa = lds4(ps + oA20);      // A[i = lr][j = 4 lq + m]; B[j = 4 lq + m][d] = the v accumulators
            zero_acc(o);
            mma_unit(pa, pa, acc4(vacc[0]), acc4(vacc[1]), o[0], o[1]);
        };
        units(std::integral_constant<int, 0>{}, std::integral_constant<int, 24>{});
        if (CF_STAGGER && w >= 4) {
            attend();
            units(std::integral_constant<int, 24>{}, std::integral_constant<int, 32>{});
        } else {
            units(std::integral_constant<int, 24>{}, std::integral_constant<int, 32>{});
            CF_STAMP8(1);
            attend();
        }
        {
            float* ap_ = lane_at(sbase(ag, 0), zA);      // (scalar bases are formed outside the per-row conditionals)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const float val = o[t][ii] * fast_sigmoid(gacc[t][ii]);
                    if (rok[ii]) {
                        as_[oDLW + ii * LW + t * 16] = val;
                        if (SAVE) SAVE_ST(ap_ + ii * kRDm + t * 16, val);
                    }
                }
        }
        LnParams ln1, ln2;
        if (w < 4) ln1 = ln_params_load(P.g1, P.be1);
        BBuf<NU1, NU1> bw1;
        CF_STAMP8(2);
        __syncthreads();
        CF_STAMP8(3);
        // ---- out-projection + residual, LayerNorm
        {
            f32x4 acc[2];
            zero_acc(acc);
            const auto co_w1 = [&](int u) {      // FFN weights 1, one unit per unit
                if (u < NU1) {
                    if (DFF == 256)
                        b_issue1(bw1, P.w1_t + w1o, wl, idx_two, u);
                    else
                        b_issue1(bw1, P.w1_t + w1o, wl, idx_one, u);
                }
            };
            if (r0)
                b_run_row0<8, false, 2>(bwo, P.wo_t + wo16, wl, idx_one, as_ + lq * 4, acc, co_w1);
            else
                b_run<8, false, 2>(bwo, P.wo_t + wo16, wl, idx_one, as_ + oALW, acc, co_w1);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) ts[oDLD + ii * LD] = (acc[0][ii] + acc[1][ii]) + bov + xs[oDLD + ii * LD];
        }
        CF_STAMP8(4);
        __syncthreads();
        if (w < 4) ln_fwd_tile16<(CF_SAVE_NT & 4) != 0>(ts, LD, ln1, row0, T, SAVE ? P.xh1 : nullptr, P.rs1, SAVE ? P.y1 : nullptr);
        CF_STAMP8(5);
        __syncthreads();
        // ---- FFN (modules.py:100-101)
        BBuf<NU2, NU2> bw2;
        const auto co_w2 = [&](int u) {
            if (u < NU2) b_issue1(bw2, P.w2_t + w2o, wl, idx_one, u);
        };
        {
            f32x4 acc[2];
            zero_acc(acc);
            if (DFF == 256) {
                if (r0)
                    b_run_row0<NU1, true, 2>(bw1, P.w1_t + w1o, wl, idx_two, ts + lq * 4, acc, co_w2);
                else
                    b_run<NU1, true, 2>(bw1, P.w1_t + w1o, wl, idx_two, ts + oALD, acc, co_w2);
                float* hp = lane_at(sbase(hg, 0), zH);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float hv[4];
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        hv[ii] = fmaxf(acc[t][ii] + b1v[t], 0.f);
                        hs[oDLH + ii * LH + t * 16] = hv[ii];
                        if (SAVE && rok[ii]) SAVE_ST(hp + ii * DFF + t * 16, hv[ii]);
                    }
                }
            } else {
                if (r0)
                    b_run_row0<NU1, false, 2>(bw1, P.w1_t + w1o, wl, idx_one, ts + lq * 4, acc, co_w2);
                else
                    b_run<NU1, false, 2>(bw1, P.w1_t + w1o, wl, idx_one, ts + oALD, acc, co_w2);
                if (NU2 > NU1) b_issue<NU1, NU2>(bw2, P.w2_t + w2o, wl, idx_one);
                float* hp = lane_at(sbase(hg, 0), zH);
                float hv[4];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    hv[ii] = fmaxf((acc[0][ii] + acc[1][ii]) + b1v[0], 0.f);
                    hs[oDLH + ii * LH] = hv[ii];
                    if (SAVE && rok[ii]) SAVE_ST(hp + ii * DFF, hv[ii]);
                }
            }
        }
        if (w < 4) ln2 = ln_params_load(P.g2, P.be2);
        CF_STAMP8(6);
        __syncthreads();
        {
            f32x4 acc[2];
            zero_acc(acc);
            const bool more = l + 1 < a.n_layers;
            const float* nq = load_layer(tab + (more ? l + 1 : l)).watt_t + wq;
            const auto co_nq = [&](int u) {      // next layer's projection ring
                if (u < kQPre && more) b_issue1(rq, nq, wl, qidx, u);
            };
            if (r0)
                b_run_row0<NU2, false, 2>(bw2, P.w2_t + w2o, wl, idx_one, hs + lq * 4, acc, co_nq);
            else
                b_run<NU2, false, 2>(bw2, P.w2_t + w2o, wl, idx_one, hs + oALH, acc, co_nq);
            if (kQPre > NU2 && more) b_issue<NU2, kQPre>(rq, nq, wl, qidx);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) xs[oDLD + ii * LD] = (acc[0][ii] + acc[1][ii]) + b2v + ts[oDLD + ii * LD];
        }
        CF_STAMP8(7);
        __syncthreads();
        if (w < 4) ln_fwd_tile16<(CF_SAVE_NT & 4) != 0>(xs, LD, ln2, row0, T, SAVE ? P.xh2 : nullptr, P.rs2, P.xout);
        CF_STAMP8(8);
        __syncthreads();
    };
    const int n_full = a.n_layers - (a.row0_last ? 1 : 0);
    for (int l = 0; l < n_full; ++l) layer(std::false_type{}, l);
    if (a.row0_last) layer(std::true_type{}, n_full);
    // the prediction head of the gene, forward + loss + backward, by the last of its resolutions' workgroups to get here (cf_head_ride.h);
    // row 0 of xs is the stack's output for token 0, everything behind it in the LDS is free
    if (SAVE && a.head.on) head_ride_tail(a.head, g, r, a.B, T, a.n_res, xs, load_layer(tab).xin + (size_t)row0 * kD, as_);
}

// Backward of the same stack.  LDS: ds = d(layer output) -> dy1 -> dt1 -> d(layer input); t2 = dt2; wide = dpre1;
// dqk = d(q | k | v | gate) of the gene (A operand of the K = 1024 input-gradient product).
template <int DFF>
__global__ __launch_bounds__(512) void k_reg8_bwd(RegArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool ride_loss = a.head.on && blockIdx.x == 0 && (threadIdx.x >> 6) == 7;      // (the head ran at the tail of the forward launch: cf_head_ride.h)
    const float ride_l = ride_loss ? head_ride_loss_request(a.head, a.B) : 0.f;
    const int per = gridDim.x >> 3, v = a.xcd_map ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : (int)blockIdx.x;
    if (v >= a.B * a.n_res) return;
    const int g = __builtin_amdgcn_readfirstlane(v % a.B), r = __builtin_amdgcn_readfirstlane(v / a.B), T = a.T, TT = T * T, row0 = g * T, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    constexpr int LD = kD + 4, LH = DFF + 4;
    constexpr int NUA = DFF == 256 ? 8 : 4, NUB = DFF / 32;      // units of the W2^T and W1^T products
    constexpr int HW = DFF == 256 ? 32 : 16;
    float* ds = smem;                    // [16][LD]
    float* t2 = ds + kTile * LD;         // [16][LD]
    float* wide = t2 + kTile * LD;       // [16][LH]
    float* dqk = wide + kTile * LH;      // [16][kQkLd]
    float* dos = dqk + kTile * kQkLd + w * kPatchB;      // wave-private: do rows [16][36]
    float* dss = dos + 16 * 36;                          //               ds      [16][20]
    const RegLayerDev* tab = a.tab + (size_t)r * a.n_layers;
    const float scale = sqrtf((float)kRDh);
    const bool from_top = a.l_top + 1 == a.n_layers;      // (a launch over the lower layers starts from what the one above it stored as d(layer input))
    {
        const float* d0 = load_layer(tab + a.l_top).dxout + (size_t)row0 * kD;
        const int Tin = a.row0_last && from_top ? 1 : T;      // (only token 0 of the stack's output has a gradient, see b_run_row0)
        for (int i = tid; i < kTile * (kD / 4); i += 512) {
            const int row = i >> 5, c4 = i & 31;
            float4 t = f4z();
            if (row < Tin) t = ldg4(d0 + row * kD + c4 * 4);
            *reinterpret_cast<float4*>(ds + row * LD + c4 * 4) = t;
        }
        for (int i = tid; i < kTile * kQkLd; i += 512) dqk[i] = 0.f;      // rows >= T stay zero (MFMA operand)
    }
    const int oD36 = lq * 4 * 36 + lr, oA36 = lr * 36 + lq * 4, oD20 = lq * 4 * 20 + lr, oA20 = lr * 20 + lq * 4;
    const int oDLD = lq * 4 * LD + w * 16 + lr, oALD = lr * LD + lq * 4;
    const int oDLH = lq * 4 * LH + w * HW + lr, oALH = lr * LH + lq * 4;
    const int oDQ = lq * 4 * kQkLd + w * 32 + lr, oAQ = lr * kQkLd + lq * 4;
    const int hq0 = (g * kRH + w) * kHqFloats;
    const unsigned bT = (lq * 64 + lr * 4) * 4;                      // transposed tiles / p^T, one float4 per lane, quarter-major
    const unsigned bVr = (lr * 32 + lq * 4) * 4;                     // B-operand rows of v: v[j = lr][4 lq ..]
    const unsigned zA = (lq * 4 * kRDm + lr) * 4, zH = (lq * 4 * DFF + lr) * 4, zD = (lq * 4 * kD + lr) * 4, zQ = (lq * 4 * kRW + lr) * 4;
    const unsigned wl = lane * 16;
    const int wAo = (DFF == 256 ? 2 * w : w) * 8 * 256, wBo = w * (DFF / 16) * 256, wCo = 2 * w * 8 * 256, wDo = w * 64 * 256;
    bool rok[4];
    float fqv[4];
    bool mkv[4];
    unsigned zAc[4], zHc[4];      // byte offsets of the lane's rows of a / hdn, clamped into the gene's rows (branch-free loads)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int i = lq * 4 + ii;
        rok[ii] = i < T;
        const bool in = i < T && lr < T;
        fqv[ii] = in ? ldg(a.freq + (size_t)g * TT + i * T + lr) : 0.f;
        mkv[ii] = in ? *(const CF_GLOBAL uint8_t*)(a.mask[r] + (size_t)g * TT + i * T + lr) != 0 : true;      // outside the block: no gradient
        zAc[ii] = (min(i, T - 1) * kRDm + lr) * 4;
        zHc[ii] = (min(i, T - 1) * DFF + lr) * 4;
    }
    const auto idx_two = [](int u, int j) { return j * 8 + u; };
    const auto idx_one = [](int u, int j) { return 2 * u + j; };
    const int lrow = (w & 3) * 4 + lq;               // LayerNorm row of this lane (waves 0..3)
    const bool llive = w < 4 && lrow < T;
    const unsigned bL = (lrow * kD + lr * 8) * 4, bR = lrow * 4;      // LayerNorm lane: 8 columns of row lrow
    // operands of a layer's first phase (LayerNorm 2 backward on waves 0..3, the W2^T stream for everybody), requested one
    // phase ahead: during the input-gradient product of the layer above, or here for the top layer
    BBuf<NUA, NUA> bA;
    float4 l2ga = f4z(), l2gb = f4z(), l2x0 = f4z(), l2x1 = f4z();
    float l2rs = 0.f;
    auto request_stream = [&](const RegLayerDev& Q) {
        if (DFF == 256)
            b_issue<0, NUA>(bA, Q.w2_tt + wAo, wl, idx_two);
        else
            b_issue<0, NUA>(bA, Q.w2_tt + wAo, wl, idx_one);
    };
    auto request_ln2 = [&](const RegLayerDev& Q) {
        if (w < 4) {
            l2ga = ldg4(lane_at(Q.g2, lr * 32));
            l2gb = ldg4(lane_at(Q.g2, lr * 32) + 4);
            const float* xp = lane_at(sbase(Q.xh2, row0 * kD), bL);
            const float* rp = lane_at(sbase(Q.rs2, row0), bR);
            if (llive) {
                l2x0 = BWD_LD4(xp);
                l2x1 = BWD_LD4(xp + 4);
                l2rs = ldg(rp);
            }
        }
    };
    request_ln2(load_layer(tab + a.l_top));
    __syncthreads();
    auto layer = [&](auto r0_c, const int l) __attribute__((always_inline)) {
        constexpr bool r0 = decltype(r0_c)::value;      // the last layer: one live row above the attention (see b_run_row0)
        const RegLayerDev P = load_layer(tab + l);
        CF_STAMP8(0);
        // ---- LayerNorm 2 backward (waves 0..3) first, then everybody's requests for the next phase
        if (w < 4) ln_bwd_tile16_r(ds, t2, LD, l2x0, l2x1, l2ga, l2gb, l2rs, llive, lane_at(sbase(P.dt2, row0 * kD), bL));
        request_stream(P);
        float hm[2][4];      // FFN hidden of this lane's outputs (ReLU mask of the next phase); rows >= T read row T - 1 (their gradient is zero)
#pragma unroll
        for (int t = 0; t < (DFF == 256 ? 2 : 1); ++t)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) hm[t][ii] = SAVE_LD(lane_at(sbase(P.hdn, row0 * DFF + w * HW), zHc[ii]) + t * 16);
        CF_STAMP8(1);
        __syncthreads();
        // ---- dpre1 = (dt2 W2) . relu'
        BBuf<NUB, NUB> bB;
        const auto co_B = [&](int u) {
            if (u < NUB) b_issue1(bB, P.w1_tt + wBo, wl, idx_one, u);
        };
        float* dpp = lane_at(sbase(P.dpre1, row0 * DFF + w * HW), zH);
        {
            f32x4 acc[2];
            zero_acc(acc);
            if (DFF == 256) {
                if (r0)
                    b_run_row0<NUA, true, 2>(bA, P.w2_tt + wAo, wl, idx_two, t2 + lq * 4, acc, co_B);
                else
                    b_run<NUA, true, 2>(bA, P.w2_tt + wAo, wl, idx_two, t2 + oALD, acc, co_B);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const float vv = hm[t][ii] > 0.f ? acc[t][ii] : 0.f;
                        wide[oDLH + ii * LH + t * 16] = vv;
                        if (rok[ii]) BWD_ST(dpp + ii * DFF + t * 16, vv);
                    }
                }
            } else {
                if (r0)
                    b_run_row0<NUA, false, 2>(bA, P.w2_tt + wAo, wl, idx_one, t2 + lq * 4, acc, co_B);
                else
                    b_run<NUA, false, 2>(bA, P.w2_tt + wAo, wl, idx_one, t2 + oALD, acc, co_B);
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const float vv = hm[0][ii] > 0.f ? acc[0][ii] + acc[1][ii] : 0.f;
                    wide[oDLH + ii * LH] = vv;
                    if (rok[ii]) BWD_ST(dpp + ii * DFF, vv);
                }
            }
        }
        CF_STAMP8(2);
        __syncthreads();
        // ---- dy1 = dpre1 W1 + dt2
        BBuf<8, 8> bC;
        float4 l1ga = f4z(), l1gb = f4z(), l1x0 = f4z(), l1x1 = f4z();
        float l1rs = 0.f;
        if (w < 4) {
            l1ga = ldg4(lane_at(P.g1, lr * 32));
            l1gb = ldg4(lane_at(P.g1, lr * 32) + 4);
            if (llive) {
                l1x0 = BWD_LD4(lane_at(sbase(P.xh1, row0 * kD), bL));
                l1x1 = BWD_LD4(lane_at(sbase(P.xh1, row0 * kD), bL) + 4);
                l1rs = ldg(lane_at(sbase(P.rs1, row0), bR));
            }
        }
        {
            f32x4 acc[2];
            zero_acc(acc);
            const auto co_C = [&](int u) { b_issue1(bC, P.wo_tt + wCo, wl, idx_two, u); };
            if (r0)
                b_run_row0<NUB, false, 2>(bB, P.w1_tt + wBo, wl, idx_one, wide + lq * 4, acc, co_C);
            else
                b_run<NUB, false, 2>(bB, P.w1_tt + wBo, wl, idx_one, wide + oALH, acc, co_C);
            if (NUB < 8) b_issue<NUB, 8>(bC, P.wo_tt + wCo, wl, idx_two);
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const float vv = (acc[0][ii] + acc[1][ii]) + t2[oDLD + ii * LD];
                ds[oDLD + ii * LD] = vv;
                if (rok[ii]) BWD_ST(lane_at(sbase(P.dy1, row0 * kD + w * 16), zD) + ii * kD, vv);
            }
        }
        CF_STAMP8(3);
        __syncthreads();
        // ---- LayerNorm 1 backward (in place, waves 0..3) first; then everybody requests the operands of its head's attention backward
        if (w < 4) ln_bwd_tile16_r(ds, ds, LD, l1x0, l1x1, l1ga, l1gb, l1rs, llive, lane_at(sbase(P.dt1, row0 * kD), bL));
        float4 gT[2], kT[2], qT[2], vr[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {      // (only pieces that hold a token: the rest of the block is never written nor read)
            const float* gp = lane_at(sbase(P.hq, hq0 + kHqG + t * 256), bT);
            const float* kp = lane_at(sbase(P.hq, hq0 + kHqK + t * 256), bT);
            const float* qp = lane_at(sbase(P.hq, hq0 + kHqQ + t * 256), bT);
            const float* vp = lane_at(sbase(P.hq, hq0 + kHqV), bVr) + t * 16;
            gT[t] = kT[t] = qT[t] = vr[t] = f4z();
            if (rok[0]) {
                gT[t] = SAVE_LD4(gp);
                kT[t] = SAVE_LD4(kp);
                qT[t] = SAVE_LD4(qp);
            }
            if (lr < T) vr[t] = SAVE_LD4(vp);
        }
        float4 pT = f4z();      // lane (j = lr, lq): p[4 lq + ii][j]
        {
            const float* pp = lane_at(sbase(P.hq, hq0 + kHqP), bT);
            if (rok[0] && lr < T) pT = SAVE_LD4(pp);
        }
        float av[2][4];      // gated attention output of the head (forward); rows >= T read row T - 1
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) av[t][ii] = SAVE_LD(lane_at(sbase(P.a, row0 * kRDm + w * 32), zAc[ii]) + t * 16);
        CF_STAMP8(4);
        __syncthreads();
        // ---- da = dt1 Wo (the 32 columns of head w), then the attention backward of the head, wave-local
        BBuf<32, 8> bD;
        {
            f32x4 da[2];
            zero_acc(da);
            const auto co_D = [&](int u) {
                if (u < 7) b_issue1(bD, P.watt_tt + wDo, wl, idx_one, u);
            };
            if (r0)
                b_run_row0<8, true, 2>(bC, P.wo_tt + wCo, wl, idx_two, ds + lq * 4, da, co_D);
            else
                b_run<8, true, 2>(bC, P.wo_tt + wCo, wl, idx_two, ds + oALD, da, co_D);
            float* dg = P.dqkvg + row0 * kRW + w * 32;
            f32x4 dov[2];      // do = da . sigmoid(gate): D layout = B operand of dv
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float gt[4] = {gT[t].x, gT[t].y, gT[t].z, gT[t].w};
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const float sg = fast_sigmoid(gt[ii]);
                    const float dgate = da[t][ii] * av[t][ii] * (1.0f - sg);      // da . o . s (1 - s), a = o s
                    dov[t][ii] = da[t][ii] * sg;
                    dos[oD36 + ii * 36 + t * 16] = dov[t][ii];
                    if (rok[ii]) {
                        dqk[oDQ + ii * kQkLd + 3 * kRDm + t * 16] = dgate;
                        BWD_ST(lane_at(sbase(dg, ii * kRW + 3 * kRDm), zQ) + t * 16, dgate);
                    }
                }
            }
            wave_lds_sync();
            // dp[i][j] = do[i] . v[j]
            f32x4 dp[2];
            zero_acc(dp);
            mma_unit(lds4(dos + oA36), lds4(dos + oA36 + 16), vr[0], vr[1], dp[0], dp[1]);
            const float pv[4] = {pT.x, pT.y, pT.z, pT.w};
            f32x4 dsv;      // lane (j = lr, lq): ds[i = 4 lq + ii][j], already divided by sqrt(dh)
            float gsum = 0.f;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const float d = dp[0][ii] + dp[1][ii];
                const float dot = group16_sum(pv[ii] * d);
                const float sv = mkv[ii] ? 0.f : pv[ii] * (d - dot);
                gsum = fmaf(sv, fqv[ii], gsum);
                dsv[ii] = sv / scale;
                dss[oD20 + ii * 20] = dsv[ii];
            }
            gsum = wave_sum(gsum);
            if (lane == 0) stg(P.dgam + (size_t)g * kRH + w, gsum);
            wave_lds_sync();
            const float4 dsa = lds4(dss + oA20);      // A[i = lr][j = 4 lq + m]
            f32x4 dq[2], dk[2], dv[2];
            zero_acc(dq);
            zero_acc(dk);
            zero_acc(dv);
            mma_unit(dsa, dsa, kT[0], kT[1], dq[0], dq[1]);                         // dq[i][d] = sum_j ds[i][j] k[j][d]
            mma_unit(acc4(dsv), acc4(dsv), qT[0], qT[1], dk[0], dk[1]);             // dk[j][d] = sum_i ds[i][j] q[i][d]
            mma_unit(pT, pT, acc4(dov[0]), acc4(dov[1]), dv[0], dv[1]);             // dv[j][d] = sum_i p[i][j] do[i][d]
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii)
                    if (rok[ii]) {
                        const int o = oDQ + ii * kQkLd + t * 16;
                        dqk[o] = dq[t][ii];
                        dqk[o + kRDm] = dk[t][ii];
                        dqk[o + 2 * kRDm] = dv[t][ii];
                        float* dgi = lane_at(sbase(dg, ii * kRW), zQ) + t * 16;
                        BWD_ST(dgi, dq[t][ii]);
                        BWD_ST(dgi + kRDm, dk[t][ii]);
                        BWD_ST(dgi + 2 * kRDm, dv[t][ii]);
                    }
        }
        CF_STAMP8(5);
        __syncthreads();
        // ---- d(layer input) = dt1 + d(q|k|v|gate) Watt   (K = 1024: 32 units, four accumulators)
        if (l > a.l_bot) request_ln2(load_layer(tab + l - 1));
        {
            f32x4 acc[4];
            zero_acc(acc);
            if (r0)
                b_run_kv_rows<7>(bD, P.watt_tt + wDo, wl, idx_one, dqk + oAQ, dqk + lq * 4, acc);
            else
                b_run<7, false, 4>(bD, P.watt_tt + wDo, wl, idx_one, dqk + oAQ, acc);

#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const float vv = ((acc[0][ii] + acc[1][ii]) + (acc[2][ii] + acc[3][ii])) + ds[oDLD + ii * LD];
                ds[oDLD + ii * LD] = vv;
                if (rok[ii]) BWD_ST(lane_at(sbase(P.dxin, row0 * kD + w * 16), zD) + ii * kD, vv);
            }
        }
        CF_STAMP8(6);
        __syncthreads();
    };
    const bool top_row0 = a.row0_last && from_top;
    if (top_row0) layer(std::true_type{}, a.l_top);
    for (int l = a.l_top - (top_row0 ? 1 : 0); l >= a.l_bot; --l) layer(std::false_type{}, l);
    if (ride_loss) head_ride_loss(a.head, a.B, ride_l);
}

}  // namespace cf

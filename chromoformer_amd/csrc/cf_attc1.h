// cf_attc1.h -- centre-row attention, ONE region per workgroup, on the vector ALUs (included by cf_kernels.h behind cf_attc2.h).
//
// Same mathematics and the same argument block as k_attc2 (cf_attc2.h) for the launches that run one region per workgroup
// (the Embedding stage at bsz 64; every centre-row launch of a small batch).  With one region only 2 of the 16 rows of an MFMA
// tile are live, and the two passes over the positional table cost a full tile each (2 x ~9 K cycles of the matrix pipe per
// workgroup, whatever the number of live rows).  On the vector ALUs the same passes are 2 x 128 x L fused multiply-adds over 512
// threads -- a few hundred per thread -- and what is left is the table stream itself (2 x L x 128 floats through the CU's load
// path, ~3 K cycles each at 64 bytes per clock):
//   (2) thread = (4 bins, quarter of the channels):   t[h][j] = sum_e vin[h][e] PE^T[e][j]     16-byte loads, unit-stride across lanes
//   (5) thread = (4 channels, sixteenth of the bins): out[h][e] = sum_j sc[h][j] PE[j][e]     16-byte loads, unit-stride across lanes
// Each thread requests its whole share of the table in two rounds (the passes are latency-bound otherwise: one round trip per
// 8 rows made the first version exactly as slow as the MFMA kernel), partial sums meet in LDS.
// The softmax (forward) / its backward run as workgroup-wide reductions over the bins (wave reduction + 8 partials in LDS).
#pragma once

namespace cf {

__host__ __device__ inline size_t attc1_smem(int L, int F) {
    const int Lpad = attc2_lpad(L);
    return (size_t)(2 * kD + 16 + 16 + kD * 8 + ((L * F + 8 + 3) & ~3) + 2 * (Lpad + 4) + 2 * Lpad + 64 + 16 * 2 * kD) * sizeof(float) + (size_t)Lpad;
}

#define CF_STAMP1(slot)                                                                                   \
    do {                                                                                                  \
        if (a.tdbg && stamp_wg == 0 && threadIdx.x == 0)                                                   \
            a.tdbg[(BWD ? 16 : 0) + (slot)] = __builtin_amdgcn_s_memtime();                               \
    } while (0)
template <bool BWD>
__device__ __forceinline__ void attc1_body(const Attc2Args& a, const int r, const int n, float* smem, const int stamp_wg = -1) {
    const int tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63;
    const int L = a.L[r], Lpad = a.Lpad[r], LT = a.LT[r], F = a.F, LS = Lpad + 4;
    float* vin_s = smem;                      // [2][128]   qt / dxbar
    float* u_s = vin_s + 2 * kD;              // [2][8]
    float* w_s = u_s + 16;                    // [2][8]
    float* wlp_s = w_s + 16;                  // [128][8]   zero padded to 8 marks
    float* feats_s = wlp_s + kD * 8;          // [L][F] (+8)
    float* sc_s = feats_s + ((L * F + 8 + 3) & ~3);      // [2][LS] (16-byte aligned)    scores -> p (fwd) / dp -> ds (bwd); zero for j >= L
    float* p_s = sc_s + 2 * LS;               // [2][Lpad]  backward: the saved p
    float* red_s = p_s + 2 * Lpad;            // [8 waves][2 heads][4]: partials of the workgroup-wide reductions
    float* part_s = red_s + 64;               // [16][2][128] K-split partials of pass 5; [4][2][512] partials of pass 2
    uint8_t* mk_s = reinterpret_cast<uint8_t*>(part_s + 16 * 2 * kD);     // [Lpad]

    CF_STAMP1(0);
    // ---- stage: everything a load round trip away is requested before anything is put (loads return in order: one round trip
    //      for the lot), and the first half of this thread's share of PE^T (pass 2, block 0) behind it
    const int eg = tid >> 7, j4 = tid & 127;      // pass 2: channels 32 eg .. 32 eg + 31, bins 4 j4 .. 4 j4 + 3 of a 512-bin block
    float4 tb[16];      // half of the thread's share of a table pass (two rounds of loads per pass: 32 at once stall the issue and spill)
    {
        const float* fg = a.feats[r] + (size_t)n * L * F;
        const uint8_t* mg = a.mask[r] + (size_t)n * a.mstride[r];
        const float wl0 = (tid & 7) < F ? ldg(a.wlp[r] + (tid >> 3) * F + (tid & 7)) : 0.f;
        const float wl1 = (tid & 7) < F ? ldg(a.wlp[r] + ((tid + kAT) >> 3) * F + (tid & 7)) : 0.f;
        float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 64) vv = ldg4(a.vin[r] + (size_t)n * 256 + tid * 4);
        uint8_t mb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = tid + u * kAT;
            mb[u] = *(const CF_GLOBAL uint8_t*)(mg + min(j, L - 1));      // Lpad <= 1024; no select behind a load (it would wait for it)
        }
        float pb[4];
        if (BWD) {
            const float* pg = a.p[r] + (size_t)n * 2 * L;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = tid + u * kAT, h = i / Lpad, j = i - h * Lpad;
                pb[u] = ldg(pg + min(h, 1) * L + min(j, L - 1));
            }
        }
        float fb[8];
        const int nfe = L * F;
#pragma unroll
        for (int u = 0; u < 8; ++u) fb[u] = ldg(fg + min(tid + u * kAT, nfe - 1));      // L * F <= 4096 floats in this round
        {
            const float* pt = a.pet[r] + (size_t)(eg * 32) * LT + min(4 * j4, LT - 4);
#pragma unroll
            for (int k = 0; k < 16; ++k) tb[k] = ldg4(pt + (size_t)k * LT);
        }
        __builtin_amdgcn_sched_barrier(0);
        wlp_s[tid] = wl0;
        wlp_s[tid + kAT] = wl1;
        if (tid < 64) *reinterpret_cast<float4*>(vin_s + tid * 4) = vv;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tid + u * kAT < Lpad) mk_s[tid + u * kAT] = tid + u * kAT < L ? mb[u] : (uint8_t)1;
        if (BWD)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (tid + u * kAT < 2 * Lpad) p_s[tid + u * kAT] = (tid + u * kAT) % Lpad < L ? pb[u] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (tid + u * kAT < nfe + 8) feats_s[tid + u * kAT] = tid + u * kAT < nfe ? fb[u] : 0.f;
        for (int i = tid + 8 * kAT; i < nfe + 8; i += kAT) feats_s[i] = i < nfe ? ldg(fg + i) : 0.f;      // (longer regions: the slow way)
        for (int i = tid; i < 2 * LS; i += kAT) {      // zero tail of the score rows (pass 5 reads them 16 bytes at a time)
            const int j = i % LS;
            if (j >= L) sc_s[i] = 0.f;
        }
    }
    __syncthreads();
    CF_STAMP1(1);
    // ---- (1) u[h][f] = sum_e vin[h][e] Wlp[e][f]: thread = (h, f, 1/32 of e), half-wave reduction
    {
        const int o = tid >> 5, part = tid & 31, h = o >> 3, f = o & 7;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = fmaf(vin_s[h * kD + part * 4 + i], wlp_s[(part * 4 + i) * 8 + f], acc);
        acc = group16_sum(acc);
        acc += __shfl_xor(acc, 16, 64);
        if (part == 0) u_s[o] = acc;
    }
    // ---- (2) t[h][j] = sum_e vin[h][e] PE^T[e][j]: thread = (bins 4 j4 .. 4 j4 + 3 of a 512-bin block, channels 32 eg .. 32 eg + 31);
    //      the four channel groups meet in LDS, then thread = bin j (and j + 512 when L > 512)
    float t0[2] = {0.f, 0.f}, t1[2] = {0.f, 0.f};
    for (int c = 0; c * kAT < L; ++c) {
        const int jc = min(c * kAT + 4 * j4, LT - 4);      // (columns up to LT, a multiple of 64, exist: zero past L)
        const float* pt = a.pet[r] + (size_t)(eg * 32) * LT + jc;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 b[16];
            if (c == 0) {      // block 0: the first half was requested in the stage, the second is requested before the first is multiplied
#pragma unroll
                for (int k = 0; k < 16; ++k) b[k] = tb[k];
                if (half == 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) tb[k] = ldg4(pt + (size_t)(16 + k) * LT);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) b[k] = ldg4(pt + (size_t)(half * 16 + k) * LT);
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const float4 v0 = *reinterpret_cast<const float4*>(vin_s + eg * 32 + half * 16 + 4 * k4);
                const float4 v1 = *reinterpret_cast<const float4*>(vin_s + kD + eg * 32 + half * 16 + 4 * k4);
                const float q0[4] = {v0.x, v0.y, v0.z, v0.w}, q1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 bb = b[4 * k4 + i];
                    s0 = make_float4(fmaf(q0[i], bb.x, s0.x), fmaf(q0[i], bb.y, s0.y), fmaf(q0[i], bb.z, s0.z), fmaf(q0[i], bb.w, s0.w));
                    s1 = make_float4(fmaf(q1[i], bb.x, s1.x), fmaf(q1[i], bb.y, s1.y), fmaf(q1[i], bb.z, s1.z), fmaf(q1[i], bb.w, s1.w));
                }
            }
        }
        if (c) __syncthreads();      // (the previous block's partials have been read)
        *reinterpret_cast<float4*>(part_s + (eg * 2) * kAT + 4 * j4) = s0;
        *reinterpret_cast<float4*>(part_s + (eg * 2 + 1) * kAT + 4 * j4) = s1;
        __syncthreads();
        t0[c] = (part_s[tid] + part_s[2 * kAT + tid]) + (part_s[4 * kAT + tid] + part_s[6 * kAT + tid]);
        t1[c] = (part_s[kAT + tid] + part_s[3 * kAT + tid]) + (part_s[5 * kAT + tid] + part_s[7 * kAT + tid]);
    }
    // pass 5's first round of table rows (channels 4 e4 .. 4 e4 + 3, a sixteenth of the bins) is requested here and lands under the softmax
    const int e4 = tid & 31, jg = tid >> 5;
    const int per5 = (L + 15) / 16;              // bins per group
    const int j05 = jg * per5, nj5 = max(0, min(L, j05 + per5) - j05);
    const float* pp5 = a.pe[r] + (size_t)min(j05, Lpad - 1) * kD + 4 * e4;      // (rows up to Lpad exist in the padded table)
#pragma unroll
    for (int k = 0; k < 16; ++k) tb[k] = ldg4(pp5 + (size_t)min(k, max(nj5 - 1, 0)) * kD);
    __syncthreads();      // u
    CF_STAMP1(2);
    // ---- epilogue of pass 2 + softmax / its backward over the bins of the two rows
    auto block_reduce = [&](float v0, float v1, bool is_max, float& o0, float& o1) {
        v0 = is_max ? wave_max(v0) : wave_sum(v0);
        v1 = is_max ? wave_max(v1) : wave_sum(v1);
        __syncthreads();      // (the previous round's partials have been read)
        if (lane == 0) {
            red_s[w * 2] = v0;
            red_s[w * 2 + 1] = v1;
        }
        __syncthreads();
        o0 = red_s[0];
        o1 = red_s[1];
#pragma unroll
        for (int k = 1; k < kAT / 64; ++k) {
            o0 = is_max ? fmaxf(o0, red_s[k * 2]) : o0 + red_s[k * 2];
            o1 = is_max ? fmaxf(o1, red_s[k * 2 + 1]) : o1 + red_s[k * 2 + 1];
        }
    };
    {
        float x0[2], x1[2];
        bool valid[2], masked[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = tid + c * kAT;
            valid[c] = j < L;
            masked[c] = valid[c] && mk_s[min(j, Lpad - 1)];
            float f0 = 0.f, f1 = 0.f;
            if (valid[c])
                for (int f = 0; f < F; ++f) {
                    const float fv = feats_s[(size_t)j * F + f];
                    f0 = fmaf(fv, u_s[f], f0);
                    f1 = fmaf(fv, u_s[8 + f], f1);
                }
            x0[c] = t0[c] + f0;
            x1[c] = t1[c] + f1;
        }
        if (!BWD) {
            float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                x0[c] = masked[c] ? kMaskFill : x0[c] * a.rscale;
                x1[c] = masked[c] ? kMaskFill : x1[c] * a.rscale;
                if (valid[c]) {
                    m0 = fmaxf(m0, x0[c]);
                    m1 = fmaxf(m1, x1[c]);
                }
            }
            float M0, M1, Z0, Z1;
            block_reduce(m0, m1, true, M0, M1);
            float z0 = 0.f, z1 = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                x0[c] = valid[c] ? __expf(x0[c] - M0) : 0.f;
                x1[c] = valid[c] ? __expf(x1[c] - M1) : 0.f;
                z0 += x0[c];
                z1 += x1[c];
            }
            block_reduce(z0, z1, false, Z0, Z1);
            const float r0 = 1.0f / Z0, r1 = 1.0f / Z1;
            float* pg = a.p[r] + (size_t)n * 2 * L;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = tid + c * kAT;
                if (valid[c]) {
                    const float p0 = x0[c] * r0, p1 = x1[c] * r1;
                    sc_s[j] = p0;
                    sc_s[LS + j] = p1;
                    stg(pg + j, p0);
                    stg(pg + L + j, p1);
                }
            }
        } else {
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = tid + c * kAT;
                if (valid[c]) {
                    d0 = fmaf(p_s[j], x0[c], d0);
                    d1 = fmaf(p_s[Lpad + j], x1[c], d1);
                }
            }
            float D0, D1;
            block_reduce(d0, d1, false, D0, D1);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int j = tid + c * kAT;
                if (valid[c]) {
                    sc_s[j] = masked[c] ? 0.f : p_s[j] * (x0[c] - D0) * a.rscale;
                    sc_s[LS + j] = masked[c] ? 0.f : p_s[Lpad + j] * (x1[c] - D1) * a.rscale;
                }
            }
        }
    }
    __syncthreads();
    CF_STAMP1(3);
    // ---- (4) w[h][f] = sum_j sc[h][j] f_j[f]: thread = (h, f, 1/32 of the bins), half-wave reduction
    {
        const int o = tid >> 5, part = tid & 31, h = o >> 3, f = o & 7;
        float acc = 0.f;
        if (f < F)
            for (int j = part; j < L; j += 32) acc = fmaf(sc_s[h * LS + j], feats_s[(size_t)j * F + f], acc);
        acc = group16_sum(acc);
        acc += __shfl_xor(acc, 16, 64);
        if (part == 0) {
            w_s[o] = acc;
            stg(a.w[r] + (size_t)n * 16 + o, acc);
        }
    }
    CF_STAMP1(4);
    CF_STAMP1(5);
    // ---- (5) out[h][e] = sum_j sc[h][j] PE[j][e]: thread = (channels 4 e4 .. 4 e4 + 3, sixteenth of the bins)
    {
        float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
        for (int jb = 0; jb < nj5; jb += 16) {       // (two rounds for L = 400)
            float4 b[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = tb[k];
            if (jb + 16 < nj5) {
#pragma unroll
                for (int k = 0; k < 16; ++k) tb[k] = ldg4(pp5 + (size_t)min(jb + 16 + k, nj5 - 1) * kD);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool in = jb + k < nj5;
                const float ph = in ? sc_s[j05 + jb + k] : 0.f, qh = in ? sc_s[LS + j05 + jb + k] : 0.f;
                o0 = make_float4(fmaf(ph, b[k].x, o0.x), fmaf(ph, b[k].y, o0.y), fmaf(ph, b[k].z, o0.z), fmaf(ph, b[k].w, o0.w));
                o1 = make_float4(fmaf(qh, b[k].x, o1.x), fmaf(qh, b[k].y, o1.y), fmaf(qh, b[k].z, o1.z), fmaf(qh, b[k].w, o1.w));
            }
        }
        __syncthreads();      // (pass 2's partials in the same buffer have long been read; pass 4 is done with its LDS reads)
        *reinterpret_cast<float4*>(part_s + (jg * 2) * kD + 4 * e4) = o0;
        *reinterpret_cast<float4*>(part_s + (jg * 2 + 1) * kD + 4 * e4) = o1;
    }
    __syncthreads();
    CF_STAMP1(6);
    if (tid < 2 * kD) {
        const int h = tid >> 7, e = tid & 127;
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) v += part_s[(g * 2 + h) * kD + e];
#pragma unroll
        for (int f = 0; f < 8; ++f) v = fmaf(w_s[h * 8 + f], wlp_s[e * 8 + f], v);
        stg(a.vout[r] + (size_t)n * 256 + tid, v);
    }
    CF_STAMP1(7);
}
template <bool BWD>
__global__ __launch_bounds__(kAT) void k_attc1(Attc2Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    attc1_body<BWD>(a, gridDim.y - 1 - blockIdx.y, blockIdx.x, smem, blockIdx.y * gridDim.x + blockIdx.x);      // long-sequence workgroups first
}

}  // namespace cf

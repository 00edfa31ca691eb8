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
};
#define CF_STAMP2(slot)                                                                                   \
    do {                                                                                                  \
        if (a.tdbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)                              \
            a.tdbg[(BWD ? 16 : 0) + (slot)] = __builtin_amdgcn_s_memtime();                               \
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

template <bool BWD, int AG>
__global__ __launch_bounds__(kAT) void k_attc2(Attc2Args a) {
    constexpr int kAG = AG;                   // rows 2*AG .. 15 of the MFMA tile are dead (zero operand rows)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // resolutions in reverse launch order: the long-sequence workgroups (last binsize) are dispatched first
    const int r = gridDim.y - 1 - blockIdx.y, n0 = blockIdx.x * kAG, tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int L = a.L[r], Lpad = a.Lpad[r], LT = a.LT[r], F = a.F, LS = Lpad + 4, N = a.N;
    constexpr int LD = kD + 4, NW = kAT / 64;
    float* sc_s = smem;                       // [16][LS]  scores -> p (fwd) / dp -> ds (bwd); zero for j >= L
    float* vin_s = sc_s + kTile * LS;         // [16][LD]  qt / dxbar rows (m = 2*region + head)
    float* red_s = vin_s + kTile * LD;        // [16][LD]  K-split partial of the second table product
    float* u_s = red_s + kTile * LD;          // [16][8]
    float* w_s = u_s + kTile * 8;             // [16][8]
    float* wlp_s = w_s + kTile * 8;           // [128][8]  Wlp, zero padded to 8 marks (all loops over marks run to 8)
    float* feats_s = wlp_s + kD * 8;          // [8][L][F] (+8 floats of slack: the padded mark loops read one past)
    uint8_t* mk_s = reinterpret_cast<uint8_t*>(feats_s + kAG * L * F + 8);   // [8][Lpad]
    const int nreg = min(kAG, N - n0);        // regions present in this workgroup

    CF_STAMP2(0);
    const int nblk = (L + 63) / 64;
    FragNN<4, 8> ft0;                         // PE^T operand ring of this wave's first block of pass 2
    // ---- stage: features (the HBM stream), operand rows, masks
    {
        const float* fg = a.feats[r] + (size_t)n0 * L * F;
        const int nf = nreg * L * F;
        const uint8_t* mg = a.mask[r] + (size_t)n0 * a.mstride[r];
        const bool words = ((L | (int)a.mstride[r] | (int)(reinterpret_cast<uintptr_t>(a.mask[r]))) & 3) == 0;
        const int LW = Lpad >> 2, lw = L >> 2;
        constexpr int MW = kAG >= 2 ? kAG / 2 : 1;       // mask words per thread: kAG * (Lpad / 4 <= 256) / 512
        // every load of the stage is issued before the first LDS store waits for one (loads return in order, so a
        // load -> store -> load -> store sequence would pay the L2 latency once per array)
        const float wl0 = (tid & 7) < F ? ldg(a.wlp[r] + (tid >> 3) * F + (tid & 7)) : 0.f;
        const float wl1 = (tid & 7) < F ? ldg(a.wlp[r] + ((tid + kAT) >> 3) * F + (tid & 7)) : 0.f;      // kD * 8 = 2 * kAT entries
        float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
        {
            const int m = tid >> 5, c4 = tid & 31;               // 16 rows x 32 float4
            if ((m >> 1) < nreg) vv = ldg4(a.vin[r] + (size_t)n0 * 256 + m * kD + c4 * 4);
        }
        uint32_t mwv[MW];
        if (words) {
#pragma unroll
            for (int k = 0; k < MW; ++k) {
                const int idx = tid + k * kAT, sreg = idx / LW, jw = idx - sreg * LW;
                mwv[k] = (idx < kAG * LW && sreg < nreg && jw < lw) ? *(const CF_GLOBAL uint32_t*)(mg + (size_t)sreg * a.mstride[r] + 4 * jw) : 0x01010101u;
            }
        }
        if (((L * F) & 3) == 0) {                      // 16-byte copies, 12 in flight per thread
            const int n4 = (kAG * L * F) >> 2, nf4 = nf >> 2;
            {                                          // first round: covers L*F <= 3072 (L = 400: 2800)
                float4 v[12];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = tid + u * kAT;
                    v[u] = (u * kAT < n4 && i < nf4) ? ldg4(fg + (size_t)i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int i = tid + u * kAT;
                    if (i < n4) *reinterpret_cast<float4*>(feats_s + (size_t)i * 4) = v[u];
                }
            }
            for (int i = tid + 12 * kAT; i < n4; i += kAT)
                *reinterpret_cast<float4*>(feats_s + (size_t)i * 4) = i < nf4 ? ldg4(fg + (size_t)i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < 8) feats_s[kAG * L * F + tid] = 0.f;
        } else {
            for (int i = tid; i < kAG * L * F + 8; i += kAT) feats_s[i] = i < nf ? ldg(fg + i) : 0.f;
        }
        static_assert(kD * 8 == 2 * kAT, "wlp staging assumes two entries per thread");
        wlp_s[tid] = wl0;
        wlp_s[tid + kAT] = wl1;
        *reinterpret_cast<float4*>(vin_s + (tid >> 5) * LD + (tid & 31) * 4) = vv;
        if (words) {
            uint32_t* mk_w = reinterpret_cast<uint32_t*>(mk_s);
#pragma unroll
            for (int k = 0; k < MW; ++k)
                if (tid + k * kAT < kAG * LW) mk_w[tid + k * kAT] = mwv[k];
        } else {
            for (int s = 0; s < kAG; ++s)
                for (int j = tid; j < Lpad; j += kAT)
                    mk_s[s * Lpad + j] = (s < nreg && j < L) ? *(const CF_GLOBAL uint8_t*)(mg + (size_t)s * a.mstride[r] + j) : (uint8_t)1;
        }
        {
            const int m = tid >> 5;                                // zero tail of the score rows (K padding of pass 5)
            for (int j = L + (tid & 31); j < Lpad; j += 32) sc_s[m * LS + j] = 0.f;
        }
        // PE^T operand ring of pass 2: requested last (loads return in order: anything issued before the staging loads
        // would have to land before their data can be stored), in flight across the barrier and pass 1
        frag_load_nn(ft0, a.pet[r] + min(w, nblk - 1) * 64, LT);
    }
    __syncthreads();
    CF_STAMP2(1);
    // ---- (1) u[m][f] = sum_e vin[m][e] Wlp[e][f]: thread = (m, f, quarter of e)
    {
        const int m = tid >> 5, f = (tid >> 2) & 7, part = tid & 3;
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < kD / 4; ++i) {
            const int e = part * (kD / 4) + i;
            acc = fmaf(vin_s[m * LD + e], wlp_s[e * 8 + f], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (part == 0) u_s[m * 8 + f] = acc;
    }
    __syncthreads();
    CF_STAMP2(2);
    // ---- (2) t[m][j] = vin[m] . PE^T[:, j]  (+ f_j . u[m]);  wave w takes column blocks w, w+8, ...
    {
        for (int jb = w; jb < nblk; jb += NW) {
            f32x4 acc[4];
            zero_acc(acc);
            if (jb == w) frag_mma_nn(ft0, vin_s, LD, acc);
            else {
                FragNN<4, 8> ft;
                frag_load_nn(ft, a.pet[r] + jb * 64, LT);
                frag_mma_nn(ft, vin_s, LD, acc);
            }
            const int j0 = jb * 64 + 4 * lr;
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
#pragma unroll
                for (int hd = 0; hd < 2; ++hd) {
                    const int ii = sp * 2 + hd, m = lq * 4 + ii;
                    const float4 u0 = *reinterpret_cast<const float4*>(u_s + m * 8), u1 = *reinterpret_cast<const float4*>(u_s + m * 8 + 4);
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int j = j0 + t;
                        float x = acc[t][ii];
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
                            if (mk_s[s * Lpad + min(j, Lpad - 1)]) x = kMaskFill;
                        }
                        v[t] = j < L ? x : 0.f;
                    }
                    if (j0 < Lpad) *reinterpret_cast<float4*>(sc_s + m * LS + j0) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    __syncthreads();
    CF_STAMP2(3);
    const int cg = w & 1, kh = w >> 1, kper = Lpad / 4;      // pass 5: 2 column groups x 4 quarters of the bin range
    FragNN<4, 8> fp5;
    frag_load_nn_rt(fp5, a.pe[r] + (size_t)(kh * kper) * kD + cg * 64, kD, kper / 32);   // in flight across passes 3, 4
    // LPR lanes per row m: 32 with all 16 rows live, a whole wave per row when at most 8 are
    constexpr int LPR = AG == 8 ? 32 : 64, LSH = AG == 8 ? 5 : 6;
    const int gm = tid >> LSH, sub = tid & (LPR - 1);
    const bool row_live = gm < 2 * kAG;
    auto sum32 = [](float v) {
        v = group16_sum(v);
        v += __shfl_xor(v, 16, 64);
        if (LPR == 64) v += __shfl_xor(v, 32, 64);
        return v;
    };
    // ---- (3) softmax over the bins (fwd) / its backward (bwd): LPR lanes per row, each lane owns the 16-byte
    //      strips j = 4*sub + 4*LPR*k; everything a lane needs is fetched before the arithmetic starts
    auto softmax_pass = [&](auto kmax_c) {
        constexpr int KMAX = decltype(kmax_c)::value;      // strips per lane
        float* row = sc_s + gm * LS;
        const bool present = row_live && (gm >> 1) < nreg;
        const int nk = (L + 4 * LPR - 1) / (4 * LPR);
        float4 xv[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < nk) xv[k] = *reinterpret_cast<const float4*>(row + 4 * sub + 4 * LPR * k);      // entries j >= L are 0
        if (!BWD) {
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < nk) {
                    const int j = 4 * sub + 4 * LPR * k;
                    if (j < L) mx = fmaxf(mx, xv[k].x);
                    if (j + 1 < L) mx = fmaxf(mx, xv[k].y);
                    if (j + 2 < L) mx = fmaxf(mx, xv[k].z);
                    if (j + 3 < L) mx = fmaxf(mx, xv[k].w);
                }
            mx = group16_max(mx);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            if (LPR == 64) mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float z = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < nk) {
                    const int j = 4 * sub + 4 * LPR * k;
                    xv[k].x = j < L ? __expf(xv[k].x - mx) : 0.f;
                    xv[k].y = j + 1 < L ? __expf(xv[k].y - mx) : 0.f;
                    xv[k].z = j + 2 < L ? __expf(xv[k].z - mx) : 0.f;
                    xv[k].w = j + 3 < L ? __expf(xv[k].w - mx) : 0.f;
                    z += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);
                }
            z = sum32(z);
            const float rz = 1.0f / z;
            float* pg = a.p[r] + ((size_t)n0 * 2 + gm) * L;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < nk) {
                    const int j = 4 * sub + 4 * LPR * k;
                    const float4 pv = make_float4(xv[k].x * rz, xv[k].y * rz, xv[k].z * rz, xv[k].w * rz);
                    *reinterpret_cast<float4*>(row + j) = pv;
                    if (present) {
                        if (j + 3 < L && (L & 3) == 0) stg4(pg + j, pv);
                        else {
                            if (j < L) stg(pg + j, pv.x);
                            if (j + 1 < L) stg(pg + j + 1, pv.y);
                            if (j + 2 < L) stg(pg + j + 2, pv.z);
                            if (j + 3 < L) stg(pg + j + 3, pv.w);
                        }
                    }
                }
        } else {
            const float* pg = a.p[r] + ((size_t)n0 * 2 + gm) * L;
            float4 pv[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < nk) {
                    const int j = 4 * sub + 4 * LPR * k;
                    pv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (present) {
                        if (j + 3 < L && (L & 3) == 0) pv[k] = ldg4(pg + j);
                        else {
                            if (j < L) pv[k].x = ldg(pg + j);
                            if (j + 1 < L) pv[k].y = ldg(pg + j + 1);
                            if (j + 2 < L) pv[k].z = ldg(pg + j + 2);
                            if (j + 3 < L) pv[k].w = ldg(pg + j + 3);
                        }
                    }
                }
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < nk) dot += (pv[k].x * xv[k].x + pv[k].y * xv[k].y) + (pv[k].z * xv[k].z + pv[k].w * xv[k].w);
            dot = sum32(dot);
            const uint8_t* mk = mk_s + (gm >> 1) * Lpad;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < nk) {
                    const int j = 4 * sub + 4 * LPR * k;
                    const uint32_t mw = *reinterpret_cast<const uint32_t*>(mk + j);
                    float4 d;
                    d.x = (mw & 0xffu) ? 0.f : pv[k].x * (xv[k].x - dot) * a.rscale;
                    d.y = (mw & 0xff00u) ? 0.f : pv[k].y * (xv[k].y - dot) * a.rscale;
                    d.z = (mw & 0xff0000u) ? 0.f : pv[k].z * (xv[k].z - dot) * a.rscale;
                    d.w = (mw & 0xff000000u) ? 0.f : pv[k].w * (xv[k].w - dot) * a.rscale;
                    *reinterpret_cast<float4*>(row + j) = d;      // j >= L: p = 0 there, so 0 is written back
                }
        }
    };
    if (row_live) {
        if (L <= 4 * LPR * 4) softmax_pass(std::integral_constant<int, 4>{});
        else softmax_pass(std::integral_constant<int, 8>{});      // L <= 1024 (checked on the host)
    }
    __syncthreads();
    CF_STAMP2(4);
    // ---- (4) w[m][f] = sum_j sc[m][j] f_j[f]
    {
        float acc[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[f] = 0.f;
        const float* fp = feats_s + (row_live ? gm >> 1 : 0) * L * F;
        for (int j = sub; j < (row_live ? L : 0); j += LPR) {
            const float pv = sc_s[gm * LS + j];
#pragma unroll
            for (int f = 0; f < 8; ++f) acc[f] = fmaf(pv, fp[j * F + f], acc[f]);           // entries f >= F are dropped below
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
        float* part = kh == 1 ? red_s : kh == 2 ? vin_s : sc_s;      // [16][LD] each (LS >= LD)
        if (kh != 0) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                *reinterpret_cast<float4*>(part + (lq * 4 + ii) * LD + cg * 64 + 4 * lr) = make_float4(acc[0][ii], acc[1][ii], acc[2][ii], acc[3][ii]);
        }
        __syncthreads();
        if (kh == 0) {
            // Wlp rows of this lane's four output columns: 8 sixteen-byte reads instead of 32 scalar ones per row
            float wl[4][8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 w0 = *reinterpret_cast<const float4*>(wlp_s + (cg * 64 + 4 * lr + t) * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(wlp_s + (cg * 64 + 4 * lr + t) * 8 + 4);
                wl[t][0] = w0.x, wl[t][1] = w0.y, wl[t][2] = w0.z, wl[t][3] = w0.w;
                wl[t][4] = w1.x, wl[t][5] = w1.y, wl[t][6] = w1.z, wl[t][7] = w1.w;
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int m = lq * 4 + ii, e0 = cg * 64 + 4 * lr;
                const float4 wa = *reinterpret_cast<const float4*>(w_s + m * 8), wb = *reinterpret_cast<const float4*>(w_s + m * 8 + 4);
                const float wm[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                const float4 r1 = *reinterpret_cast<const float4*>(red_s + m * LD + e0);
                const float4 r2 = *reinterpret_cast<const float4*>(vin_s + m * LD + e0);
                const float4 r3 = *reinterpret_cast<const float4*>(sc_s + m * LD + e0);
                float v[4] = {acc[0][ii] + r1.x + r2.x + r3.x, acc[1][ii] + r1.y + r2.y + r3.y, acc[2][ii] + r1.z + r2.z + r3.z,
                              acc[3][ii] + r1.w + r2.w + r3.w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int f = 0; f < 8; ++f) v[t] = fmaf(wm[f], wl[t][f], v[t]);
                if ((m >> 1) < nreg) stg4(a.vout[r] + ((size_t)n0 * 2 + m) * kD + e0, make_float4(v[0], v[1], v[2], v[3]));
            }
        }
    }
    CF_STAMP2(7);
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

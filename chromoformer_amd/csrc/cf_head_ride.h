// cf_head_ride.h -- the prediction head of a gene, forward + loss + backward, at the TAIL of the Regulation forward launch (round 4;
// included by cf_kernels.h in front of cf_reg8.h).
//
// net.py:377-380, train.py:156, 193-195: hin = cat_r(x_out[r][0] + x_in[r][0]); h1 = relu(W1 hin + b1); logits = W2 h1 + b2; loss; and
// back: dlogits, dh1 = (W2^T dlogits) . relu', dhin = W1^T dh1 = the gradient of token 0 of every Regulation output.  As a launch of its
// own (k_head_train, cf_head.h) that is 17 us for a chain of four dependent steps on four workgroups.  The three resolutions of a
// gene meet nowhere else: here each Regulation workgroup (gene g, resolution r) writes its 128-float chunk of hin THROUGH to the
// device's coherence point (agent-scope relaxed stores: `global_store_dword sc1` -- 512 bytes, not a cache write-back), counts itself
// in on a per-gene counter (relaxed agent-scope atomic behind `s_waitcnt vmcnt(0)`), and the workgroup that arrives LAST runs the
// gene's head on the vector ALUs (one row: the matrix-vector routines of cf_valu_mv.h), reading the other two chunks back with
// agent-scope loads.  The counters are zeroed by workgroup 0 at the start of the launch (the kernel is part of a replayed graph); the
// mean loss is summed in gene order by one wave at the start of the Regulation backward launch.  No spinning, no fence that writes
// back an L2; a first version that let the last gene to finish sum the losses and rewind the counters spent 14 us in the tail
// (65 dependent agent-scope accesses), the launch it replaces takes 17.
#pragma once

namespace cf {

struct HeadRide {
    int on;                      // 0: the launch ends with the Regulation output, the head is somebody else's
    int n_out;
    float gscale;                // loss scale (1 / world under data parallelism)
    const void* labels;          // int64 [B] (n_out = 2) / float [B] (n_out = 1)
    const float* w1_t;           // fc_head.0.weight [128, 384], tiled copy
    const float* w1;             // ... row-major
    const float *b1, *w2, *b2;   // fc_head.0.bias, fc_head.2.weight [n_out, 128], fc_head.2.bias
    float *hin, *h1, *logits, *logits_user, *dlogits, *dh1, *dhin;      // [B, 384], [B, 128], [B, n_out] x 3, [B, 128], [B, 384]
    float* dxl[kMaxRes];         // Regulation output gradient [B * T, 128]: token 0 receives dhin (the other rows stay zero)
    float *loss, *loss_part, *loss_user;
    int* cnt;                    // [B] arrivals per gene (zeroed by workgroup 0 of the forward launch at its start)
};

// The per-gene arrival counters are MONOTONIC (round 5): a launch adds n_res to each, the workgroup that reads a value congruent to n_res - 1
// runs the gene's head -- nobody ever writes a counter but the arrivals themselves.  (Round 4 had workgroup 0 zero them at the start of the
// launch, "100 us before anybody arrives", and this round first let the last arriver rewind its counter: with two processes on one device --
// the two-rank tests -- a gene's head was skipped about once in ten runs, stale logits and all; with CF_HEAD_RIDE=0 never.  Zeroed at cf_create;
// 2^32 is not a multiple of 3, so a counter would drift after 1.4e9 launches: every 2^28 launches the host puts them back to zero with a
// stream-ordered memset in FRONT of a launch (cf_api.hip, ride_tick), where every counter is a multiple of n_res and nobody is arriving.)
__device__ __forceinline__ void head_ride_begin(const HeadRide&, const int) {}
// The Regulation BACKWARD launch, one wave of workgroup 0: the mean loss of the batch, summed in gene order (the genes' losses were
// written by whichever workgroup ran each gene's head; a launch boundary lies in between).  The first 64 values are requested at the
// start of the launch (head_ride_loss_request) and added at its end: as the first thing the wave does, the cold loads held its whole
// workgroup -- and with it the launch -- back by 2 us.
__device__ __forceinline__ float head_ride_loss_request(const HeadRide& hd, const int B) {
    const int lane = threadIdx.x & 63;
    return lane < B ? ldg(hd.loss_part + lane) : 0.f;
}
__device__ __forceinline__ void head_ride_loss(const HeadRide& hd, const int B, const float first) {
    const int lane = threadIdx.x & 63;
    float tot = 0.f;
    for (int i0 = 0; i0 < B; i0 += 64) {
        const float l = i0 == 0 ? first : (i0 + lane < B ? ldg(hd.loss_part + i0 + lane) : 0.f);
        for (int i = 0; i < min(64, B - i0); ++i) tot += __shfl(l, i, 64);
    }
    if (lane == 0) {
        tot /= (float)B;
        hd.loss[0] = tot;
        if (hd.loss_user) hd.loss_user[0] = tot;
    }
}
// xs0: the workgroup's Regulation output row of token 0 (LDS); xin0: its Regulation input row of token 0 (global); lds: >= 16 * 384 + 1024
// floats of the workgroup's LDS, free now.  All 512 threads of the workgroup call this, after a barrier behind the last LayerNorm.
__device__ __forceinline__ void head_ride_tail(const HeadRide& hd, const int g, const int r, const int B, const int T, const int n_res,
                                               const float* xs0, const float* xin0, float* lds) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    constexpr int K = kMaxRes * kD;
    float* hs = lds;                 // [384] hin of the gene
    float* h1s = hs + K;             // [128]
    float* dhs = h1s + kD;           // [128] dh1
    float* sc = dhs + kD;            // [8]   dlogits, flags
    float* part = sc + 8;            // [16][384] partial sums of dhin
    int* flag = reinterpret_cast<int*>(sc + 4);
    // the operand streams of both products are requested before anybody knows who will need them: the loser of the count returns
    // a microsecond later without having waited for them, the winner finds them there
    if (tid < kD) __hip_atomic_store(hd.hin + (size_t)g * K + r * kD + tid, xs0[tid] + ldg(xin0 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The chunk must have reached the coherence point before the workgroup counts itself in.  Nothing else of this wave is in flight here (the
    // last layer's operand rings have drained), so a full wait costs the write-through latency of this one store and depends on nothing the
    // compiler may do to the loads behind it (round 4 waited with vmcnt(24) for "everything but the 24 operand loads requested after the
    // store": right for the code the compiler emitted then, silently wrong if it ever splits, merges or hoists one of them).
#ifndef CF_HEAD_RIDE_COUNTED_WAIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NtW<kD, K> w1t;
    nt_request(w1t, hd.w1_t);
#else
    asm volatile("" ::: "memory");
    NtW<kD, K> w1t;
    nt_request(w1t, hd.w1_t);
    static_assert(NtW<kD, K>::TW * NtW<kD, K>::KB == 24, "loads of the operand stream behind the store");
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
#endif
    __syncthreads();
    if (tid == 0) *flag = __hip_atomic_fetch_add(hd.cnt + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if ((unsigned)*flag % (unsigned)n_res != (unsigned)n_res - 1u) return;
    // ---- the gene's last workgroup: forward
    if (tid < K) hs[tid] = __hip_atomic_load(hd.hin + (size_t)g * K + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n = nt_row<kD, K>(0);
    const float b1 = ldg(hd.b1 + n);
    const float w2a = tid < kD ? ldg(hd.w2 + tid) : 0.f, w2b = tid < kD && hd.n_out == 2 ? ldg(hd.w2 + kD + tid) : 0.f;
    __syncthreads();
    {
        float z[1];
        nt_dot(w1t, hs, z);
        if (nt_writer()) {
            const float v = fmaxf(z[0] + b1, 0.f);
            h1s[n] = v;
            stg(hd.h1 + (size_t)g * kD + n, v);
        }
    }
    // dhin[k] = sum_j dh1[j] W1[j][k]: thread = (four columns of one 128-wide chunk, sixteenth of the rows), three chunks; requested now
    float4 wr[kMaxRes][8];
    {
        const int c4 = tid & 31, ng = tid >> 5;
        const float* p = hd.w1 + (size_t)(ng * 8) * K + 4 * c4;
#pragma unroll
        for (int ch = 0; ch < kMaxRes; ++ch)
#pragma unroll
            for (int i = 0; i < 8; ++i) wr[ch][i] = ldg4(p + (size_t)i * K + ch * kD);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (w == 0) {      // logits, loss, d loss / d logits (wave 0)
        float s0 = fmaf(h1s[lane], ldg(hd.w2 + lane), h1s[lane + 64] * ldg(hd.w2 + lane + 64));
        float s1 = hd.n_out == 2 ? fmaf(h1s[lane], ldg(hd.w2 + kD + lane), h1s[lane + 64] * ldg(hd.w2 + kD + lane + 64)) : 0.f;
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) {
            const float z0 = s0 + ldg(hd.b2), z1 = hd.n_out == 2 ? s1 + ldg(hd.b2 + 1) : 0.f;
            float l, d0, d1 = 0.f;
            if (hd.n_out == 1) {
                const float y = reinterpret_cast<const float*>(hd.labels)[g];
                const float d = z0 - y;
                l = d * d;
                d0 = 2.0f * d * hd.gscale / (float)B;
                hd.logits[g] = z0;
                if (hd.logits_user) hd.logits_user[g] = z0;
                hd.dlogits[g] = d0;
            } else {
                const int y = (int)reinterpret_cast<const long long*>(hd.labels)[g];
                const float m = fmaxf(z0, z1);
                const float lse = m + logf(expf(z0 - m) + expf(z1 - m));
                l = lse - (y ? z1 : z0);
                d0 = (expf(z0 - lse) - (y == 0 ? 1.f : 0.f)) * hd.gscale / (float)B;
                d1 = (expf(z1 - lse) - (y == 1 ? 1.f : 0.f)) * hd.gscale / (float)B;
                hd.logits[g * 2] = z0;
                hd.logits[g * 2 + 1] = z1;
                if (hd.logits_user) hd.logits_user[g * 2] = z0, hd.logits_user[g * 2 + 1] = z1;
                hd.dlogits[g * 2] = d0;
                hd.dlogits[g * 2 + 1] = d1;
            }
            sc[0] = d0;
            sc[1] = d1;
            hd.loss_part[g] = l;      // (summed over the genes by the backward launch: head_ride_loss)
        }
    }
    __syncthreads();
    if (tid < kD) {      // dh1 = (dlogits W2) . relu'
        float s = sc[0] * w2a;
        if (hd.n_out == 2) s = fmaf(sc[1], w2b, s);
        const float v = h1s[tid] > 0.f ? s : 0.f;
        dhs[tid] = v;
        stg(hd.dh1 + (size_t)g * kD + tid, v);
    }
    __syncthreads();
    {
        const int c4 = tid & 31, ng = tid >> 5;
#pragma unroll
        for (int ch = 0; ch < kMaxRes; ++ch) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float s = dhs[ng * 8 + i];
                acc = make_float4(fmaf(s, wr[ch][i].x, acc.x), fmaf(s, wr[ch][i].y, acc.y), fmaf(s, wr[ch][i].z, acc.z), fmaf(s, wr[ch][i].w, acc.w));
            }
            *reinterpret_cast<float4*>(part + ng * K + ch * kD + 4 * c4) = acc;
        }
    }
    __syncthreads();
    if (tid < K) {
        float s = part[tid];
        for (int gq = 1; gq < 16; ++gq) s += part[gq * K + tid];
        stg(hd.dhin + (size_t)g * K + tid, s);
        const int rr = tid >> 7, e = tid & (kD - 1);
        if (rr < n_res) stg(hd.dxl[rr] + (size_t)g * T * kD + e, s);
    }
}

}  // namespace cf

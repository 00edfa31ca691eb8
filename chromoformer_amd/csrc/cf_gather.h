// cf_gather.h -- the batch of a training step taken from the resident split inside the step's graph (included by cf_kernels.h).
//
// Replaces the DataLoader and the per-tensor .cuda() copies of the reference's loop (train.py:137-140, 171-177): the
// epoch's gene order sits in device memory, a device-side cursor says which slice of it is the current batch, one launch
// copies those genes' arrays into the batch buffers the captured kernels read and advances the cursor.
#pragma once

namespace cf {

constexpr int kGatherMaxSeg = 72, kGatherChunk = 4096;      // one 16-byte load + store per thread and chunk
struct GatherSeg {
    const char* src;      // store array
    char* dst;            // batch array
    int gene_bytes;       // bytes per gene in both
    int off, len;         // this workgroup's byte range inside a gene
    int pad;
};
struct GatherArgs {
    GatherSeg seg[kGatherMaxSeg];
    const int* order;     // [n_batches * B] gene indices of the epoch, batch-major
    int* cursor;          // [0] = next batch, [1] = batches uploaded (bound), [2] = error flags (1: ran past the epoch, 2: bad gene index)
    long long n_genes;    // genes in the store
    int B;
};
// one 4 KB chunk (segment s) of gene slot b of the batch
__device__ __forceinline__ void gather_block(const GatherArgs& a, const int b, const int segi) {
    const GatherSeg s = a.seg[segi];
    const int cur = a.cursor[0];
    if (cur < 0 || cur >= a.cursor[1]) {      // a step past the uploaded epoch: nothing is read (the batch buffers keep the last batch)
        if (threadIdx.x == 0 && b == 0 && segi == 0) a.cursor[2] |= 1;
        return;
    }
    const long long gene = a.order[(long long)cur * a.B + b];
    if (gene < 0 || gene >= a.n_genes) {
        if (threadIdx.x == 0) a.cursor[2] |= 2;      // (same value from every writer)
        return;
    }
    const char* src = s.src + gene * s.gene_bytes + s.off;
    char* dst = s.dst + (long long)b * s.gene_bytes + s.off;
    if ((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)s.len) & 15) == 0) {
        const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src);
        uint4* __restrict__ d4 = reinterpret_cast<uint4*>(dst);
        for (int i = threadIdx.x; i < s.len / 16; i += 256) d4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < s.len; i += 256) dst[i] = src[i];
    }
}
__global__ __launch_bounds__(256) void k_gather_batch(GatherArgs a) { gather_block(a, blockIdx.x, blockIdx.y); }
// The gather of a training step and the prologue of its forward pass (the tiled weight copies the first kernels read) do not depend on
// each other: one launch (cf_gather_batch_fwd + the cf_forward that follows; the cursor is advanced by the trunk's forward launch
// behind it).  6.5 + 1 + 6 us as three launches in front of every step of the training loop.
__global__ __launch_bounds__(256) void k_prologue_gather(const float* __restrict__ params, float* __restrict__ tiled, float* __restrict__ tiledT,
                                                         const RetileUnit* __restrict__ units, int n_units, GatherArgs ga) {
    if ((int)blockIdx.x < n_units) {
        retile_unit(params, tiled, tiledT, units[blockIdx.x]);
    } else {
        const int i = (int)blockIdx.x - n_units;
        gather_block(ga, i % ga.B, i / ga.B);
    }
}
// The cursor is advanced by a launch of its own behind the gather (a ticket counter that lets the last workgroup do it
// costs one device-scope atomic per workgroup on a single word: 17 us for 1,500 workgroups, 80 us for 2,900).
__global__ void k_gather_advance(int* cursor) {
    if (cursor[0] < cursor[1]) cursor[0] += 1;
}

// k_reduce_opt (cf_kernels.h: the step's last launch, the remaining weight-gradient / column-sum tiles with AdamW in their epilogues)
// ... and behind its tiles the gather of the NEXT step's batch (cf_gather_batch_next): nothing in this launch reads the batch buffers, the
// 4 KB copy blocks fill the CUs the last tiles leave idle, and the next step starts without a launch in front of it
__global__ __launch_bounds__(256) void k_reduce_opt_gather(const WgTile* __restrict__ wg, int n_wg, const CsTile* __restrict__ cs, int n_cs, int batch, int xcd,
                                                           AdamFuse o, GatherArgs ga, int2 skip) {
    const int nb = xcd_grid(n_wg);
    if ((int)blockIdx.x < nb) {
        const int t = xcd_tile(blockIdx.x, n_wg, xcd);
        if (t < n_wg) wgrad_tile<true>(wg[t >= skip.x ? t + skip.y : t], batch, &o);
    } else if ((int)blockIdx.x < nb + n_cs) {
        colsum_tile<true>(cs[blockIdx.x - nb], batch, &o);
    } else {
        const int i = (int)blockIdx.x - nb - n_cs;
        gather_block(ga, i % ga.B, i / ga.B);
    }
}

struct RecordArgs {
    const int* cursor;
    const float* logits;
    const char* labels;
    const float* loss;
    float* logits_log;
    char* labels_log;
    float* loss_log;
    int B, n_out, label_bytes;
};
// (by one workgroup of any size)
__device__ __forceinline__ void record_block(const RecordArgs& a) {
    const long long row = (long long)a.cursor[0] - 1;
    if (row < 0 || row >= a.cursor[1] || (a.cursor[2] & 1)) return;      // the logs hold cursor[1] rows; a step past the epoch logs nothing
    for (int i = threadIdx.x; i < a.B * a.n_out; i += blockDim.x) a.logits_log[row * a.B * a.n_out + i] = a.logits[i];
    for (int i = threadIdx.x; i < a.B * a.label_bytes; i += blockDim.x) a.labels_log[row * a.B * a.label_bytes + i] = a.labels[i];
    if (threadIdx.x == 0) a.loss_log[row] = a.loss[0];
}
__global__ __launch_bounds__(256) void k_record_step(RecordArgs a) { record_block(a); }

}  // namespace cf

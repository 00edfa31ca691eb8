// cf_bin.h -- raw histone signal -> model input, on the GPU (included by cf_kernels.h).
//
// ChromoformerDataset._bin_and_pad + the strand flip (data.py:68-113): for one region (fp16 [n_feats, len] on
// disk, preprocessing/scripts/extract_signals.py:66-71) and one bin size
//     window -> per-bin mean (the last bin may be short) -> natural log(1 + x) -> centred zero padding to L bins
//     (ceil left / floor right) -> mirrored for a '-' strand promoter (which swaps the pads)
// written straight into the batch layout the kernels consume: features [L, n_feats] fp32 and the pad mask of
// the region as bytes [L] (1 = padding).  A pure HBM scan: 2 bytes in per sample, 4 * n_feats bytes out per bin.
// One workgroup per region; every sum is reduced in a fixed order (deterministic).
#pragma once

namespace cf {

struct BinJob {                 // = cf_bin_job of the C ABI
    const void* raw;            // device, fp16 [n_feats, ld]
    long long ld;               // samples per feature row
    int col0, ncols;            // window of samples that is binned: [col0, col0 + ncols)
    int flip, reserved;         // flip = 1: mirror the padded result ('-' strand promoter)
    float* out;                 // [L, n_feats]
    unsigned char* mask;        // [L] or null
};

// Generic path: a wave per output bin, 2-byte loads.  Used when the fast path's alignment conditions do not hold
// (pCRE files of odd length, odd bin sizes).
__device__ __forceinline__ void bin_region_scalar(const BinJob& j, int F, int b, int L) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_full = j.ncols / b, tail = j.ncols - n_full * b;
    const int n_bins = min(n_full + (tail > 0 ? 1 : 0), L);
    const int left = (L - n_bins + 1) / 2;                       // ceil((L - n) / 2), data.py:87
    const _Float16* raw = reinterpret_cast<const _Float16*>(j.raw);
    for (int p = w; p < L; p += 4) {                             // output position (after the optional mirror)
        const int q = j.flip ? L - 1 - p : p, bin = q - left;    // position before the mirror, bin index
        float val = 0.f;
        const bool real = bin >= 0 && bin < n_bins;
        if (real) {
            const int cnt = bin < n_full ? b : tail;
            for (int f = 0; f < F; ++f) {
                const _Float16* row = raw + (size_t)f * j.ld + j.col0 + (size_t)bin * b;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int c = lane;
                for (; c + 192 < cnt; c += 256) {
                    s0 += (float)*(const CF_GLOBAL _Float16*)(row + c);
                    s1 += (float)*(const CF_GLOBAL _Float16*)(row + c + 64);
                    s2 += (float)*(const CF_GLOBAL _Float16*)(row + c + 128);
                    s3 += (float)*(const CF_GLOBAL _Float16*)(row + c + 192);
                }
                for (; c < cnt; c += 64) s0 += (float)*(const CF_GLOBAL _Float16*)(row + c);
                const float s = wave_sum((s0 + s1) + (s2 + s3));
                if (lane == f) val = logf(1.0f + s / (float)cnt);
            }
        }
        if (lane < F) j.out[(size_t)p * F + lane] = val;
        if (lane == 0 && j.mask) j.mask[p] = real ? 0 : 1;
    }
}

// Fast path (rows, window start and bin size multiples of 4 samples: every promoter file and the default bin sizes):
// a feature row is streamed in tiles of 4,096 samples, 8 bytes (4 samples) per lane and load, so a chunk never
// straddles a bin; chunk sums go through LDS, then a wave per bin of the tile adds the bin's chunks (strided over
// lanes, fixed-order wave reduction) into the bin's accumulator.  5x fewer load instructions than the generic path;
// wide bins (> 1,024 samples) stay on the generic path, where a wave already reads long contiguous runs.  (Requesting the
// next tile before reducing the current one was measured 15 % slower and is not done.)
constexpr int kBinTile = 1024;          // chunks of 4 samples per tile
constexpr int kBinMaxBins = 1024;       // n_bins_out limit of the library
__global__ __launch_bounds__(256) void k_bin_regions(const BinJob* __restrict__ jobs, int F, int b, int L) {
    __shared__ float part[kBinTile];
    __shared__ float acc[kBinMaxBins];           // per-bin sums of the feature row being streamed
    const BinJob j = jobs[blockIdx.x];
    // measured (tools/bin_bench.py, TB/s, generic / fast): 2000-sample bins 4.28 / 4.05, 500: 3.0 / 3.75, 100: 1.4 / 2.6
    const bool fast = ((j.ld | j.col0 | b) & 3) == 0 && b <= 1024 && (reinterpret_cast<uintptr_t>(j.raw) & 7) == 0 && L <= kBinMaxBins;
    if (!fast) {
        bin_region_scalar(j, F, b, L);
        return;
    }
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int n_full = j.ncols / b, tail = j.ncols - n_full * b;
    const int n_bins = min(n_full + (tail > 0 ? 1 : 0), L);
    const int left = (L - n_bins + 1) / 2;
    const int nchunk = (min(j.ncols, n_bins * b) + 3) >> 2, cpb = b >> 2;      // chunks in the window, chunks per bin
    const _Float16* raw = reinterpret_cast<const _Float16*>(j.raw);
    for (int f = 0; f < F; ++f) {
        const _Float16* row = raw + (size_t)f * j.ld + j.col0;
        for (int i = tid; i < n_bins; i += 256) acc[i] = 0.f;
        float v[kBinTile / 256];
        auto fetch = [&](int c0) {
#pragma unroll
            for (int k = 0; k < kBinTile / 256; ++k) {
                const int c = c0 + tid + 256 * k;
                float s = 0.f;
                if (c < nchunk) {
                    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
                    const v2u u = *(const CF_GLOBAL v2u*)(row + 4 * c);
                    const _Float16* hp = reinterpret_cast<const _Float16*>(&u);
                    const int rem = j.ncols - 4 * c;          // samples of the window left from this chunk on
                    s = (float)hp[0];
                    if (rem > 1) s += (float)hp[1];
                    if (rem > 2) s += (float)hp[2];
                    if (rem > 3) s += (float)hp[3];
                }
                v[k] = s;
            }
        };
        for (int c0 = 0; c0 < nchunk; c0 += kBinTile) {
            fetch(c0);
            __syncthreads();                                   // the previous tile's partials have been consumed
#pragma unroll
            for (int k = 0; k < kBinTile / 256; ++k) part[tid + 256 * k] = v[k];
            __syncthreads();
            const int c1 = min(c0 + kBinTile, nchunk);        // chunks [c0, c1) are in `part`
            const int bin0 = c0 / cpb, bin1 = (c1 - 1) / cpb;
            for (int bin = bin0 + w; bin <= bin1; bin += 4) {
                const int lo = max(bin * cpb, c0), hi = min((bin + 1) * cpb, c1);
                float s = 0.f;
                for (int c = lo + lane; c < hi; c += 64) s += part[c - c0];
                s = wave_sum(s);
                if (lane == 0) acc[bin] += s;                  // one writer per (tile, bin); tiles are sequential
            }
        }
        __syncthreads();
        for (int p = tid; p < L; p += 256) {
            const int q = j.flip ? L - 1 - p : p, bin = q - left;
            const bool real = bin >= 0 && bin < n_bins;
            float val = 0.f;
            if (real) val = logf(1.0f + acc[bin] / (float)(bin < n_full ? b : tail));
            j.out[(size_t)p * F + f] = val;
            if (f == 0 && j.mask) j.mask[p] = real ? 0 : 1;
        }
        __syncthreads();
    }
}

}  // namespace cf

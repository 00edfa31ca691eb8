// cf_bin.h -- raw histone signal -> model input, on the GPU (included by cf_kernels.h).
//
// ChromoformerDataset._bin_and_pad + the strand flip (data.py:68-113): for one region (fp16 [n_feats, len] on
// disk, preprocessing/scripts/extract_signals.py:66-71) and one bin size
//     window -> per-bin mean (the last bin may be short) -> natural log(1 + x) -> centred zero padding to L bins
//     (ceil left / floor right) -> mirrored for a '-' strand promoter (which swaps the pads)
// written straight into the batch layout the kernels consume: features [L, n_feats] fp32 and the pad mask of
// the region as bytes [L] (1 = padding).  A pure HBM scan: 2 bytes in per sample, 4 * n_feats bytes out per bin.
// One workgroup per region; a wave owns an output bin at a time, lanes stride over its samples (coalesced
// 128-byte reads), sums are reduced in a fixed order (deterministic).
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

__global__ __launch_bounds__(256) void k_bin_regions(const BinJob* __restrict__ jobs, int F, int b, int L) {
    const BinJob j = jobs[blockIdx.x];
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

}  // namespace cf

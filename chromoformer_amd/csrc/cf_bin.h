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

// =======================================================================================
// All resolutions of a region in ONE pass over its raw bytes (data.py:68-113 calls _bin_and_pad once
// per bin size on the same window).  For nested bin sizes b0 = k1 b1 = k1 k2 b2 (the default 2000 / 500 / 100) the sums of a
// coarse bin are sums of the fine bins inside it, so the raw fp16 row is read once: a WAVE owns one unit = one coarsest bin
// (b0 samples) of one region and walks its feature rows --
//   loads    4 samples (8 bytes) per lane and load, unit-stride across lanes; a chunk never straddles a bin (b2 % 4 == 0); the
//            whole row of the unit is requested at once and the NEXT row before the current one is reduced (two rows in flight)
//   reduce   chunk sums -> wave-private LDS -> lane j adds the chunks of finest bin j in index order -> lane m adds the finest
//            bins of middle bin m -> lane 0 the middle bins: every sum in a fixed order, no workgroup barrier anywhere
//   write    log(1 + sum / count) of the unit's bins of every resolution staged per feature row and written [bin][feature]
//            contiguous at the end; the wave of unit 0 also writes the zero padding and the pad-mask bytes of the region
// One launch for everything a region needs; algorithmic traffic 2 bytes per sample in + 29 bytes per output bin.  Regions
// whose rows are not 8-byte aligned (odd pCRE lengths) or configurations without nesting take bin_region_scalar per resolution.
// =======================================================================================
constexpr int kBinMaxRes = 3;
constexpr int kBinMaxF = 8;              // feature rows of the one-pass path (the output staging is sized for it)
constexpr int kBinMaxLoads = 16;         // 64-lane loads of 4 samples per unit: b0 <= 4,096 samples
struct BinJobMulti {                     // = cf_bin_job_multi of the C ABI
    const void* raw;
    long long ld;
    int col0, ncols;
    int flip, reserved;
    float* out[kBinMaxRes];
    unsigned char* mask[kBinMaxRes];
};
struct BinPlan {
    int n_res, F;
    int b[kBinMaxRes];                   // bin sizes, coarsest first
    int L[kBinMaxRes];                   // output bins per region
    int nested;                          // b[r] % b[r + 1] == 0, finest % 4 == 0, unit limits hold: the one-pass path applies
};
__host__ __device__ constexpr int bin_wave_lds_floats(int nload) { return nload * 64 + 64 + 64 + (64 + 64 + 1) * kBinMaxF; }

template <int NLOAD>
__device__ __forceinline__ void bin_unit_wave(const BinJobMulti& j, const BinPlan& pl, const int unit, float* lds) {
    const int lane = threadIdx.x & 63;
    const int F = pl.F, nres = pl.n_res;
    const int b0 = pl.b[0], bf = nres == 3 ? pl.b[2] : (nres == 2 ? pl.b[1] : pl.b[0]);      // coarsest, finest (static indices: no scratch)
    const int cpu = b0 >> 2, cpf = bf >> 2;                      // chunks per unit / per finest bin
    const int nf = b0 / bf;                                      // finest bins per unit (<= 64)
    const int nm = nres == 3 ? b0 / pl.b[1] : 0;                 // middle bins per unit
    const int fpm = nres == 3 ? pl.b[1] / bf : 0;                // finest bins per middle bin
    float* cs = lds;                                             // [NLOAD * 64] chunk sums of the row
    float* fs = cs + NLOAD * 64;                                 // [64] finest-bin sums
    float* ms = fs + 64;                                         // [64] middle-bin sums
    float* stf = ms + 64;                                        // [64][F] staged outputs, finest resolution
    float* stm = stf + 64 * kBinMaxF;                            // [64][F] middle
    float* stc = stm + 64 * kBinMaxF;                            // [F]     coarsest
    const int s0 = unit * b0;                                    // first sample of the unit inside the window
    const int ncols = j.ncols;
    const _Float16* base = reinterpret_cast<const _Float16*>(j.raw) + j.col0 + s0;
    typedef unsigned int v2u __attribute__((ext_vector_type(2)));
    v2u cur[NLOAD], nxt[NLOAD];
    auto request = [&](v2u (&dst)[NLOAD], int f) {
        const _Float16* row = base + (size_t)f * j.ld;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int c = k * 64 + lane;
            // chunks past the unit / the window are clamped to the last one that exists (no select behind a load) and zeroed below
            const int cc = min(c, min(cpu - 1, max((ncols - s0 - 1) >> 2, 0)));
            dst[k] = __builtin_nontemporal_load((const CF_GLOBAL v2u*)(row + 4 * cc));
        }
    };
    const int Frows = ncols > 0 ? F : 0;                          // (an empty window: padding only, nothing is read)
    if (Frows) request(cur, 0);
    for (int f = 0; f < Frows; ++f) {
        if (f + 1 < F) request(nxt, f + 1);
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int c = k * 64 + lane;
            const int rem = ncols - s0 - 4 * c;                   // samples of the window left from this chunk on
            const _Float16* hp = reinterpret_cast<const _Float16*>(&cur[k]);
            float s = 0.f;
            if (c < cpu && rem > 0) {
                s = (float)hp[0];
                if (rem > 1) s += (float)hp[1];
                if (rem > 2) s += (float)hp[2];
                if (rem > 3) s += (float)hp[3];
            }
            if (c < cpu) cs[c] = s;
        }
        __builtin_amdgcn_wave_barrier();
        {   // finest bins of the unit: lane j adds its chunks in index order
            float s = 0.f;
            if (lane < nf)
                for (int i = 0; i < cpf; ++i) s += cs[lane * cpf + i];
            const int g = unit * nf + lane;                      // bin index inside the region
            const int cnt = min(bf, ncols - g * bf);
            if (lane < nf) {
                fs[lane] = s;
                stf[lane * F + f] = cnt > 0 ? logf(1.0f + s / (float)cnt) : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (nres == 3) {
            float s = 0.f;
            if (lane < nm)
                for (int i = 0; i < fpm; ++i) s += fs[lane * fpm + i];
            const int g = unit * nm + lane;
            const int cnt = min(pl.b[1], ncols - g * pl.b[1]);
            if (lane < nm) {
                ms[lane] = s;
                stm[lane * F + f] = cnt > 0 ? logf(1.0f + s / (float)cnt) : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (nres >= 2 && lane == 0) {
            const float* src = nres == 3 ? ms : fs;
            const int n = nres == 3 ? nm : nf;
            float s = 0.f;
            for (int i = 0; i < n; ++i) s += src[i];
            const int cnt = min(b0, ncols - s0);
            stc[f] = logf(1.0f + s / (float)cnt);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) cur[k] = nxt[k];
    }
    // ---- the unit's bins of every resolution, [bin][feature] contiguous (mirrored regions: bins backwards, features forwards)
#pragma unroll
    for (int r = 0; r < kBinMaxRes; ++r) {
        if (r >= nres) break;
        const int br = pl.b[r], L = pl.L[r];
        const int n_all = min((ncols + br - 1) / br, L), left = (L - n_all + 1) / 2;      // data.py:87
        const int per = b0 / br;                                                         // bins of this resolution per unit
        const float* st = nres == 1 ? stf : (r == 0 ? stc : (r == nres - 1 ? stf : stm));
        const int g0 = unit * per, nv = max(0, min(per, n_all - g0));
        for (int i = lane; i < nv * F; i += 64) {
            const int bl = i / F, f = i - bl * F;
            const int q = left + g0 + bl, p = j.flip ? L - 1 - q : q;
            __builtin_nontemporal_store(st[bl * F + f], (CF_GLOBAL float*)(j.out[r] + (size_t)p * F + f));
        }
        if (j.mask[r])
            for (int bl = lane; bl < nv; bl += 64) {
                const int q = left + g0 + bl, p = j.flip ? L - 1 - q : q;
                *(CF_GLOBAL unsigned char*)(j.mask[r] + p) = 0;
            }
        if (unit == 0) {      // the padding of the region: zero rows, mask bytes 1 (positions outside [left, left + n_all))
            const int npad = L - n_all;
            for (int i = lane; i < npad * F; i += 64) {
                const int k = i / F, f = i - k * F;
                const int q = k < left ? k : n_all + k, p = j.flip ? L - 1 - q : q;
                __builtin_nontemporal_store(0.f, (CF_GLOBAL float*)(j.out[r] + (size_t)p * F + f));
            }
            if (j.mask[r])
                for (int k = lane; k < npad; k += 64) {
                    const int q = k < left ? k : n_all + k, p = j.flip ? L - 1 - q : q;
                    *(CF_GLOBAL unsigned char*)(j.mask[r] + p) = 1;
                }
        }
    }
}

// grid = (ceil(units of the longest window / 4), regions); a workgroup is four independent waves (no barrier, wave-private LDS)
template <int NLOAD>
__global__ __launch_bounds__(256) void k_bin_multi(const BinJobMulti* __restrict__ jobs, BinPlan pl) {
    __shared__ float lds[4 * bin_wave_lds_floats(NLOAD)];
    const BinJobMulti j = jobs[blockIdx.y];
    const bool fast = pl.nested && ((j.ld | j.col0) & 3) == 0 && (reinterpret_cast<uintptr_t>(j.raw) & 7) == 0;
    if (!fast) {      // rows that are not 8-byte aligned: the generic kernel's wave-per-bin walk, one resolution after the other
        if (blockIdx.x == 0) {
#pragma unroll
            for (int r = 0; r < kBinMaxRes; ++r) {
                if (r >= pl.n_res) break;
                BinJob one{j.raw, j.ld, j.col0, j.ncols, j.flip, 0, j.out[r], j.mask[r]};
                bin_region_scalar(one, pl.F, pl.b[r], pl.L[r]);
            }
        }
        return;
    }
    const int w = threadIdx.x >> 6;
    const int unit = blockIdx.x * 4 + w;
    const int n_units = max(1, (min(j.ncols, pl.L[0] * pl.b[0]) + pl.b[0] - 1) / pl.b[0]);      // (an empty window still pads)
    if (unit >= n_units) return;
    bin_unit_wave<NLOAD>(j, pl, unit, lds + w * bin_wave_lds_floats(NLOAD));
}

}  // namespace cf

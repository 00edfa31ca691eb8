// cf_embed_full.h -- the Embedding stack over ALL promoter bins (included by cf_kernels.h).
//
// The default configuration evaluates the single Embedding layer for the centre query row only.
// With embed.n_layers > 1 the keys and values of layer l + 1 are every row of layer l's output, and a consumer of the
// full promoter embedding (EmbeddingTransformer's first return value, net.py:57-59) wants every row too: both go through
// the dense transformer layer (cf_op_dense_layer_*) on the token embeddings built here.
//   k_embed_tokens      x0[n][j] = Wlp f[n][j] + PE[j]                                  (net.py:42-53)
//   k_embed_tokens_wgrad dWlp[e][f] += sum_{n,j} dx0[n][j][e] f[n][j][f]               (two-stage, fixed order)
//   k_rows_gather / k_rows_scatter: the centre row L/2 of every sequence <-> a compact [N,128] array
#pragma once

namespace cf {

struct EmbTokArgs {
    const float* feats;   // [N, L, F]
    const float* pe;      // [L, 128]
    const float* wlp;     // [128, F]
    float* x0;            // [N, L, 128]
    int N, L, F;
};
__global__ __launch_bounds__(128) void k_embed_tokens(EmbTokArgs a) {
    __shared__ float w_s[kD * 8];
    const int e = threadIdx.x;
    for (int f = 0; f < 8; ++f) w_s[e * 8 + f] = f < a.F ? a.wlp[e * a.F + f] : 0.f;
    __syncthreads();
    const long long rows = (long long)a.N * a.L;
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* f = a.feats + row * a.F;
        float acc = 0.f;
        for (int i = 0; i < a.F; ++i) acc = fmaf(f[i], w_s[e * 8 + i], acc);
        a.x0[row * kD + e] = acc + a.pe[(row % a.L) * kD + e];
    }
}

constexpr int kEmbWgRows = 256;      // rows per first-stage workgroup
struct EmbTokWgArgs {
    const float* dx0;     // [N*L, 128]
    const float* feats;   // [N*L, F]
    float* partial;       // [chunks][128*8]
    long long rows;
    int F;
};
__global__ __launch_bounds__(128) void k_embed_tokens_wgrad(EmbTokWgArgs a) {
    const int e = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * kEmbWgRows, r1 = min(a.rows, r0 + kEmbWgRows);
    float acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = 0.f;
    for (long long row = r0; row < r1; ++row) {
        const float d = a.dx0[row * kD + e];
        const float* f = a.feats + row * a.F;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < a.F) acc[i] = fmaf(d, f[i], acc[i]);
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) a.partial[(size_t)blockIdx.x * (kD * 8) + e * 8 + f] = acc[f];
}
// second stage: dW[e][f] = sum over chunks of partial[chunk][e*8 + f], chunks in order
__global__ __launch_bounds__(128) void k_embed_tokens_wgrad2(const float* partial, int chunks, int F, float* dW) {
    const int e = threadIdx.x;
    for (int f = 0; f < F; ++f) {
        float s = 0.f;
        for (int c = 0; c < chunks; ++c) s += partial[(size_t)c * (kD * 8) + e * 8 + f];
        dW[e * F + f] = s;
    }
}

// dst[n] = src[n][row][:] (gather) / dst[n][j][:] = j == row ? src[n] : 0 (scatter); row-major [N, L, 128] <-> [N(*ldn), 128]
__global__ __launch_bounds__(128) void k_rows_gather(const float* src, int L, int row, float* dst, int ld_dst) {
    dst[(size_t)blockIdx.x * ld_dst + threadIdx.x] = src[((size_t)blockIdx.x * L + row) * kD + threadIdx.x];
}
__global__ __launch_bounds__(128) void k_rows_scatter(const float* src, int L, int row, float* dst) {
    const int n = blockIdx.x / L, j = blockIdx.x % L;
    dst[(size_t)blockIdx.x * kD + threadIdx.x] = j == row ? src[(size_t)n * kD + threadIdx.x] : 0.f;
}
// a += b over n floats (the two input gradients of a self-attention layer)
__global__ __launch_bounds__(256) void k_add_inplace(float* a, const float* b, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] += b[i];
}
// valid[n][j] = !masked[n * stride + j]   (centre mask row of a structured pad mask -> the dense layer's validity bytes)
__global__ __launch_bounds__(256) void k_mask_to_valid(const uint8_t* mask, long long stride, int L, uint8_t* valid) {
    for (int j = threadIdx.x; j < L; j += 256) valid[(size_t)blockIdx.x * L + j] = mask[(size_t)blockIdx.x * stride + j] ? 0 : 1;
}

}  // namespace cf

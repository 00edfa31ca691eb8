/*
 * chromoformer_hip.h  --  C ABI of libchromoformer_hip.so (MI355X / gfx950).
 *
 * The reference (dohlee/chromoformer) has no native boundary: its hot path is the
 * Python call  model(promoter_feats, promoter_pad_masks, pcre_feats, pcre_pad_masks,
 * interaction_masks, interaction_freq)  ->  loss.backward()  ->  optimizer.step()
 * (chromoformer/train.py:182-196, chromoformer/net.py:332-380).  This header is the
 * boundary a binding for that path uses: plain pointers and sizes, no torch types.
 * Every entry point names the reference interface it replaces.
 *
 * Conventions
 *   - all data pointers are DEVICE pointers unless a comment says "host";
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it and
 *     never allocates (workspace is sized in cf_create for cfg.max_batch genes);
 *   - return value 0 = ok, negative = error; cf_last_error() gives the message;
 *   - a handle is thread-compatible (one thread at a time), handles are independent.
 */
#ifndef CHROMOFORMER_HIP_H
#define CHROMOFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_MAX_RES 3
#define CF_ABI_VERSION 1

/* Model / problem description.  Mirrors the constructor arguments of
 * ChromoformerBase (net.py:273-299) and the data-processing keys of
 * configs/default.yaml:9-12 (i_max, w_max / binsizes -> n_bins). */
typedef struct cf_config {
    int n_feats;            /* histone marks per bin (7)                          net.py:276  */
    int d_emb;              /* 128                                                net.py:277  */
    int d_head;             /* 128 (any multiple of 4 <= 1024; != 128: slower head) net.py:278  */
    int n_out;              /* 2 = classifier, 1 = regressor                      net.py:329,427 */
    int n_res;              /* number of resolutions (3)                          net.py:297  */
    int binsizes[CF_MAX_RES];   /* e.g. 2000, 500, 100 (only used for names)               */
    int n_bins[CF_MAX_RES];     /* L_b = w_max / binsize, e.g. 20, 80, 400    data.py:140  */
    int i_max;              /* pCRE slots per gene (8)                            data.py:63  */
    int embed_layers, embed_heads, embed_dmodel, embed_dff;      /* net.py:279-284 */
    int pair_layers, pair_heads, pair_dmodel, pair_dff;          /* net.py:285-290 */
    int reg_layers, reg_heads, reg_dmodel, reg_dff;              /* net.py:291-296 */
    int max_batch;          /* largest B a call will pass                                     */
} cf_config;

/* One entry of the parameter table, in the reference's state_dict() order
 * (net.py:311-330).  `offset` indexes the flat fp32 parameter buffer; the
 * same offset addresses the gradient and the two AdamW moment buffers.  Tensors that
 * never receive a gradient in the reference (w_bias, Embedding/Pairwise gamma_f,
 * Pairwise ln; train.py:157 skips them) have trainable = 0 and live after
 * cf_layout.n_active so that optimiser and all-reduce run on one contiguous range. */
typedef struct cf_param_desc {
    char name[120];
    int ndim;
    int shape[2];
    long long offset;       /* in floats */
    long long numel;
    int trainable;
} cf_param_desc;

typedef struct cf_layout {
    int n_tensors;
    long long n_total;      /* floats in the flat buffer (incl. alignment padding)      */
    long long n_active;     /* floats [0, n_active) hold every trainable tensor          */
    long long n_elems;      /* sum of numel (5,342,672 for the default config)           */
} cf_layout;

/* One batch, reference layout (net.py:332-340; produced by data.py:124-212).
 * Masks are the reference's bool tensors viewed as bytes (non-zero = masked).  Only
 * the centre query row of the two pad masks is ever consumed, so
 * they are passed as "row pointers": element (sequence n, key bin j) is read at
 *     mask[n * stride + j].
 * For the reference's [B,1,1,L,L] / [B,S,1,L,L] tensors pass  ptr + (L/2)*L  and
 * stride = L*L (zero-copy); a compact [B(,S),L] row array uses stride = L. */
typedef struct cf_batch {
    int B;
    const float*   promoter_feats[CF_MAX_RES];      /* [B,1,L,F]  fp32                 */
    const float*   pcre_feats[CF_MAX_RES];          /* [B,S,L,F]  fp32                 */
    const uint8_t* promoter_mask_row[CF_MAX_RES];   /* see above; n = gene            */
    long long      promoter_mask_stride[CF_MAX_RES];
    const uint8_t* pcre_mask_row[CF_MAX_RES];       /* n = gene * S + slot            */
    long long      pcre_mask_stride[CF_MAX_RES];
    const uint8_t* interaction_mask[CF_MAX_RES];    /* [B,1,T,T]  T = i_max + 1       */
    const float*   interaction_freq;                /* [B,T,T]                        */
} cf_batch;

typedef struct cf_handle cf_handle;

/* ---- host-only helpers (no GPU needed) ------------------------------------------ */
int         cf_abi_version(void);
const char* cf_last_error(void);
/* Parameter table for a config: fills `layout`, writes up to `cap` entries into `table`
 * (may be NULL to query the count).  Replaces walking model.state_dict(). */
int cf_param_layout(const cf_config* cfg, cf_layout* layout, cf_param_desc* table, int cap);

/* ---- lifetime --------------------------------------------------------------------- */
/* Replaces Model(...).cuda() (train.py:142-154): allocates the workspace for
 * cfg.max_batch genes and the positional tables (net.py:23-29) which the caller supplies
 * from host memory: pe[r] is float[n_bins[r] * d_emb], row-major, same values the
 * reference adds at net.py:47-53 / :124-130. */
int  cf_create(const cf_config* cfg, const float* const* pe_host, cf_handle** out);
void cf_destroy(cf_handle* h);
/* Flat fp32 buffers owned by the caller (cf_layout.n_total floats each).  grads / m / v
 * may be NULL for inference-only use.  Replaces model.parameters() / optimizer state. */
int cf_bind(cf_handle* h, float* params, float* grads, float* exp_avg, float* exp_avg_sq);

/* ---- hot path ------------------------------------------------------------------- */
/* ChromoformerBase.forward (net.py:332-380).  logits: [B, n_out].  save_for_backward: 0 = inference only; 1 = keep what
 * cf_backward needs; 2 = as 1, and the prediction head (net.py:377-380) is LEFT to the cf_backward / cf_backward_part(parts & 1)
 * call that must follow with labels: head forward, loss and head backward then run as one launch, and `logits` (which has to
 * stay valid until then) is written by that call -- the training step of a caller that does not need the logits in between. */
int cf_forward(cf_handle* h, const cf_batch* batch, float* logits, int save_for_backward, void* stream);
/* cf_forward(save_for_backward = 2) for a training step whose labels are known at forward time (train.py:182-193: they always are).
 * Where cf_head_rides(h) -- the 512-thread Regulation kernels, three resolutions, d_head = 128 -- the prediction head (net.py:377-380),
 * the loss (train.py:156) and the head's backward run at the TAIL of the Regulation forward launch, on the last of a gene's three
 * workgroups to finish, instead of in a launch of their own; the cf_backward_part / cf_backward call that follows skips the head
 * (its labels / loss arguments are ignored).  Elsewhere identical to cf_forward(..., 2, ...).  logits / loss_out may be null. */
int cf_forward_train(cf_handle* h, const cf_batch* batch, float* logits, const void* labels, float loss_scale,
                     float* loss_out, void* stream);
int cf_head_rides(cf_handle* h);
/* criterion(out, label) + loss.backward() (train.py:156, 193-195).  labels: int64 [B]
 * (n_out = 2, CrossEntropyLoss) or float [B] (n_out = 1, MSELoss).  `loss_scale`
 * multiplies the mean-over-B loss gradient (1.0 single GPU; 1/world for an all-reduce SUM).
 * Writes the scalar loss to loss_out (device, 1 float) and every trainable gradient into
 * the bound grads buffer (overwrites; nothing accumulates across calls).  Must follow a
 * cf_forward(..., save_for_backward = 1 or 2) on the same batch. */
int cf_backward(cf_handle* h, const cf_batch* batch, const void* labels, float loss_scale,
                float* loss_out, void* stream);
/* The two halves of cf_backward, for callers that replay the first as a hipGraph:
 *   cf_backward_chain  -- loss + the activation-gradient chain (head -> Regulation ->
 *                         Pairwise -> Embedding); every dY a weight gradient needs is left
 *                         in the workspace;
 *   cf_backward_reduce -- the deferred reductions: all weight gradients (dW = dY^T X, one
 *                         launch over a tile table) and all bias / LayerNorm / gamma_f
 *                         gradients (one launch). */
int cf_backward_chain(cf_handle* h, const cf_batch* batch, const void* labels, float loss_scale,
                      float* loss_out, void* stream);
int cf_backward_reduce(cf_handle* h, int B, void* stream);
/* Gradient buckets.  The trainable parameters lie in state_dict order in the flat buffers, so the model splits
 * into two adjacent ranges: CF_BUCKET_PE = Embedding + Pairwise stacks, [0, split), whose gradients exist only
 * after the whole backward chain, and CF_BUCKET_REG = Regulation stacks + fc_head, [split, n_active), 78 % of the
 * bytes, whose gradients can be reduced as soon as piece 2 of cf_backward_part has run.  A data-parallel caller
 * reduces + all-reduces (+ steps) CF_BUCKET_REG on a second stream while piece 4 still runs on the first.
 *   cf_grad_bucket          -- the flat range (in floats) of one bucket;
 *   cf_backward_reduce_part -- cf_backward_reduce restricted to a bucket mask. */
#define CF_BUCKET_REG 1
#define CF_BUCKET_PE 2
/* Round 5: the Regulation bucket in two halves, for a data-parallel caller that wants bytes on the wire while the backward pass still runs.
 * Trainable tensors lie [Embedding | Pairwise | Regulation layers below n_layers / 2 (every resolution) | the layers from there up | fc_head]
 * in the flat buffers (cf_param_layout reports every offset; the state_dict order of the table is unchanged):
 *   CF_BUCKET_REG_HI = upper Regulation layers + fc_head -- complete after cf_backward_part(parts = 1 | CF_PART_REG_HI);
 *   CF_BUCKET_REG_LO = lower Regulation layers           -- complete after cf_backward_part(parts = CF_PART_REG_LO);
 *   CF_BUCKET_REG    = both (one adjacent range), as before.
 * cf_reg_halves(h) returns the first layer of the upper half when the model's Regulation backward can run as two launches (the fused
 * kernels, two layers or more), else 0: then only CF_BUCKET_REG / parts = 2 exist. */
#define CF_BUCKET_REG_HI 4
#define CF_BUCKET_REG_LO 8
#define CF_PART_REG_HI 8
#define CF_PART_REG_LO 16
int cf_reg_halves(cf_handle* h);
int cf_grad_bucket(cf_handle* h, int bucket, long long* offset, long long* numel);
int cf_backward_reduce_part(cf_handle* h, int B, int buckets, void* stream);
/* cf_backward_chain in pieces (bit mask `parts`: 1 = loss + head, 2 = Regulation stack, 4 = Pairwise +
 * Embedding), so that a caller replaying graphs can launch one piece eagerly between two graphs (bench.py times
 * the Regulation backward kernel with HIP events that way).  Pieces must run in the order 1, 2, 4. */
int cf_backward_part(cf_handle* h, const cf_batch* batch, const void* labels, float loss_scale,
                     float* loss_out, int parts, void* stream);
/* Same, but from a caller-supplied d(loss)/d(logits) [B, n_out]. */
int cf_backward_from(cf_handle* h, const cf_batch* batch, const float* dlogits, void* stream);
/* torch.optim.AdamW.step (train.py:157, 196): decoupled weight decay, bias correction
 * from `step` (1-based), over [0, n_active). */
int cf_adamw_step(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay,
                  long long step, void* stream);
/* The same update restricted to a bucket mask (elementwise, so two calls with complementary masks and the same
 * `step` equal one cf_adamw_step). */
int cf_adamw_step_part(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay,
                       long long step, int buckets, void* stream);
/* The optimiser inside a replayed graph.  Kernel arguments are frozen at capture, so the step-dependent scalars
 * live in device memory: cf_adamw_set (never captured; once per step, before the replay, on the stream the replay
 * is launched on) writes them, cf_adamw_step_dev (capturable) applies the update to a bucket mask.  Same
 * arithmetic, same bits as cf_adamw_step_part. */
int cf_adamw_set(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay,
                 long long step, void* stream);
int cf_adamw_step_dev(cf_handle* h, int buckets, void* stream);
/* Fork / join of a side stream: `waiter` waits for everything enqueued on `signaller` so far.  Under capture the
 * other stream becomes a parallel branch of the graph (it has to be joined back before cf_capture_end). */
int cf_stream_wait(cf_handle* h, void* waiter, void* signaller);

/* EmbeddingTransformer.forward's first return value (net.py:57-59): the embedding of EVERY promoter bin, out[r] =
 * [B, 1, L_r, 128] (a NULL entry is skipped).  The training path never needs it (only the centre row is consumed,
 * net.py:59, 138); it is computed by the all-rows path -- token embeddings + the dense transformer layer of
 * cf_op_dense_layer_fwd per Embedding layer -- which is also what cf_forward / cf_backward run for the whole Embedding
 * stack when embed.n_layers > 1 (keys and values of a layer are then all rows of the previous one).  Pad masks: the full
 * [B,1,1,L,L] tensor (mask stride L*L) is honoured entry by entry; a compact centre row is expanded as the dataset's
 * structured mask not(valid x valid).  Forward only. */
int cf_embed_full(cf_handle* h, const cf_batch* batch, float* const* out, void* stream);

/* ---- hipGraph capture ---------------------------------------------------------------- */
/* The launch sequence of a step is static, so it can be captured once and replayed: every
 * library call issued on `stream` between begin and end is recorded (nothing executes);
 * cf_graph_launch replays it.  Pointers are baked in: replay reads the same device buffers,
 * so callers refill those buffers (hipMemcpyAsync) instead of passing new ones. */
int cf_capture_begin(cf_handle* h, void* stream);
int cf_capture_end(cf_handle* h, void* stream, int* graph_id);
int cf_graph_launch(cf_handle* h, int graph_id, void* stream);

/* ---- introspection (tests / profiling) -------------------------------------------- */
/* HIP-event timing of one eagerly launched kernel ("k_wgrad", "k_colsum", "k_adamw", "k_reg_fwd", "k_reg_bwd", "k_trunk_fwd", "k_trunk_bwd"; NULL
 * = off): events are recorded on the launch stream around every launch of that kernel;
 * cf_timing_read waits for them and returns the summed duration and the launch count. */
int cf_timing_select(cf_handle* h, const char* kernel);
int cf_timing_read(cf_handle* h, float* total_ms, int* count);
/* Under capture the selected kernel (k_reg_fwd / k_reg_bwd) is not recorded: the capture is split around it and
 * cf_graph_launch replays [piece 1] -> the kernel, eagerly, between two events -> [piece 2].  Select the kernel
 * before cf_capture_begin. */
/* Executed flops (2*M*N*K summed over the tile table) of one k_wgrad launch at batch B. */
double cf_wgrad_flops(cf_handle* h, int B);
/* Algorithmic flops (2 * MAC over the valid rows) of one launch of "k_wgrad", "k_reg_fwd", "k_reg_bwd", "k_trunk_fwd" or
 * "k_trunk_bwd" (the centre-row trunk: row products + two-pass centre-row attention of the Embedding row and the i_max Pairwise
 * rows of every gene and resolution; riders of the launch are not counted). */
double cf_kernel_flops(cf_handle* h, const char* kernel, int B);
/* Compute units of the device the handle lives on (hipDeviceAttributeMultiprocessorCount): callers that add rider workgroups to
 * a launch of one-per-CU workgroups (cf_rider_arm) size them with it. */
int cf_cu_count(cf_handle* h);
/* Copies a named workspace buffer (e.g. "E0.qt", "dP2.1.xbar") to dst (device);
 * *n_floats receives its size; dst may be NULL to query. */
int cf_debug_copy(cf_handle* h, const char* name, float* dst, long long* n_floats, void* stream);
/* Names of all workspace buffers, '\n' separated (host string owned by the handle). */
const char* cf_debug_names(cf_handle* h);
/* cf_backward_reduce_part(reduce_buckets) and cf_adamw_step_part(adam_buckets) as ONE launch: the tiles of the bucket under
 * reduction and the optimiser stream of the OTHER bucket (whose gradients must be complete, and all-reduced under data
 * parallelism) run side by side.  reduce_buckets: exactly one bucket; adam_buckets: disjoint from it.  Same results as the two
 * separate calls, bit for bit.  Not capturable (the optimiser's scalars are launch arguments). */
int cf_reduce_adamw_part(cf_handle* h, int B, int reduce_buckets, float lr, float beta1, float beta2, float eps,
                         float weight_decay, long long step, int adam_buckets, void* stream);
/* cf_backward_reduce_part(bucket) with cf_adamw_step_part(bucket) folded into it: the tile that finishes a gradient element applies
 * torch.optim.AdamW's update (train.py:157, 194) to the parameter and moment elements at the same offset -- one launch, no second
 * pass over gradients and optimiser state.  For single-GPU training (under data parallelism the all-reduce stands between the two).
 * bucket: one bucket or both (CF_BUCKET_REG | CF_BUCKET_PE: every tile of the step in one launch, behind the whole backward pass);
 * keep_grads != 0 also stores the gradients (otherwise the flat gradient buffer keeps what it held).  Same
 * parameters / moments as the two separate calls, bit for bit.  Fails when the all-rows Embedding path is active (embed n_layers > 1:
 * its gradients are written outside the reduction tables).  Not capturable (the optimiser's scalars are launch arguments). */
int cf_reduce_opt_part(cf_handle* h, int B, int bucket, float lr, float beta1, float beta2, float eps, float weight_decay,
                       long long step, int keep_grads, void* stream);
/* Arms "riders" for the NEXT cf_backward_part(parts & 4) call: while the Pairwise + Embedding backward occupies 192 of the 256 CUs
 * with latency chains, an extra row of workgroups of the same launch reduces up to max_tiles leading weight-gradient tiles of the
 * Regulation + head bucket (they depend on the Regulation backward only) and applies this step's AdamW update in their epilogues,
 * as cf_reduce_opt_part does.  The cf_reduce_opt_part call that follows (same `step`, a mask with CF_BUCKET_REG) skips those tiles;
 * it must follow, and nothing else may read or reduce that bucket in between.  Requires the fused trunk kernels (default
 * configuration); not capturable.  Same parameters and moments as without riders, bit for bit. */
int cf_rider_arm(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay, long long step, int keep_grads,
                 int max_tiles);
/* Number of kernels the LAST forward / backward (chain pieces + bucket reductions) / optimiser step launched, counted at the
 * launch sites (zero before the first call; calls replayed from a graph do not pass the host code and leave the counts
 * of the capture pass). */
int cf_launch_counts(cf_handle* h, int* fwd, int* bwd, int* opt);

/* ---- standalone operators (same kernels, for unit tests and micro-benchmarks) ------ */
/* C[M,N] = A[M,K] . W[N,K]^T (+ bias[N]) (relu)      -- the Linear of modules.py:20-24 */
int cf_op_linear(const float* A, const float* W, const float* bias, float* C, int M, int N, int K,
                 int relu, void* stream);
/* dW[N,K] = dY[M,N]^T . X[M,K]                       -- its weight gradient           */
int cf_op_wgrad(const float* dY, const float* X, float* dW, int M, int N, int K, void* stream);
/* dX[M,K] = dY[M,N] . W[N,K]                         -- its input gradient            */
int cf_op_dgrad(const float* dY, const float* W, float* dX, int M, int N, int K, void* stream);

/* ---- dense (all rows) transformer layer, forward --------------------------------------- */
/* One AttentionBlock / PairwiseAttentionBlock over ALL query rows (modules.py:104-112, 198-208), d_emb = d_model =
 * 128, 2 heads of 64:   a = Attn(x_q Wq^T, x_kv Wk^T, x_kv Wv^T);  y1 = LN(x_q + a Wo^T + bo);
 * y = LN(y1 + relu(y1 W1^T + b1) W2^T + b2).  Self-attention: x_q = x_kv and (wq, wkv) are the row blocks
 * [0,128) and [128,384) of `att.weight`; pairwise: wq = p_att.weight, wkv = c_att.weight.  (The training path of the
 * model never needs all rows; these operators serve the stress configuration and consumers of full embeddings.)  `ws` is a device
 * workspace of cf_op_dense_layer_workspace(...) floats; masks as in cf_op_attention_fwd. */
typedef struct cf_dense_layer {
    const float *wq, *wkv;            /* [128,128], [256,128] */
    const float *wo, *bo, *ln1_g, *ln1_b;
    const float *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
    int d_ff;                         /* 128 or 256 */
} cf_dense_layer;
long long cf_op_dense_layer_workspace(int N, int Lq, int Lk, int d_ff);
int cf_op_dense_layer_fwd(const cf_dense_layer* w, const float* x_q, const float* x_kv, const unsigned char* qvalid,
                          const unsigned char* kvalid, const unsigned char* mask, int N, int Lq, int Lk, float* y,
                          float* ws, void* stream);
/* Training form: cf_op_dense_layer_fwd_train keeps the activations the backward pass needs in `ws`
 * (cf_op_dense_layer_train_workspace floats); cf_op_dense_layer_bwd turns dy [N*Lq, 128] into dx_q [N*Lq, 128],
 * dx_kv [N*Lk, 128] (a self-attention caller adds the two) and the gradients of all twelve tensors (`grads`, same
 * shapes as the weights, overwritten).  `tables`: device scratch of 4 MiB for the tile tables.  Weight gradients
 * are split-K sums in a fixed order: bit-reproducible.  The tables are written by a device kernel from by-value arguments:
 * nothing synchronises, the call can be captured. */
typedef struct cf_dense_layer_grads {
    float *wq, *wkv, *wo, *bo, *ln1_g, *ln1_b, *w1, *b1, *w2, *b2, *ln2_g, *ln2_b;
} cf_dense_layer_grads;
long long cf_op_dense_layer_train_workspace(int N, int Lq, int Lk, int d_ff);
int cf_op_dense_layer_fwd_train(const cf_dense_layer* w, const float* x_q, const float* x_kv,
                                const unsigned char* qvalid, const unsigned char* kvalid, const unsigned char* mask,
                                int N, int Lq, int Lk, float* y, float* ws, void* stream);
int cf_op_dense_layer_bwd(const cf_dense_layer* w, const float* x_q, const float* x_kv, const unsigned char* qvalid,
                          const unsigned char* kvalid, const unsigned char* mask, int N, int Lq, int Lk,
                          const float* dy, float* dx_q, float* dx_kv, const cf_dense_layer_grads* grads, float* ws,
                          float* tables, void* stream);

/* ---- input pipeline ------------------------------------------------------------------ */
/* ChromoformerDataset._bin_and_pad + strand flip (data.py:68-113) on the device: per region, the window
 * [col0, col0 + ncols) of the raw fp16 signal [n_feats, ld] is averaged over bins of `bin_size` samples (short last
 * bin included), passed through log(1 + x), centred in n_bins_out bins (ceil-left / floor-right zero padding) and
 * mirrored if `flip`; out receives [n_bins_out, n_feats] fp32 (the slot of the region in a batch / store array),
 * mask (optional) the pad bytes [n_bins_out] (1 = padding).  `jobs` is a DEVICE array.  The caller guarantees
 * ceil(ncols / bin_size) <= n_bins_out (the reference raises for longer regions, data.py:168-169). */
typedef struct cf_bin_job {
    const void* raw;
    long long ld;
    int col0, ncols;
    int flip, reserved;
    float* out;
    unsigned char* mask;
} cf_bin_job;
int cf_bin_regions(const cf_bin_job* jobs, int n_jobs, int n_feats, int bin_size, int n_bins_out, void* stream);
/* All resolutions of every region in ONE launch and -- when the bin sizes nest (each a multiple of the next, the finest a
 * multiple of 4 samples: the default 2000 / 500 / 100) and the rows are 8-byte aligned -- ONE pass over the raw bytes: the
 * reference calls _bin_and_pad once per bin size on the same window (data.py:134-140, 168-176 -> 68-100); here a wave reads a
 * coarsest bin's samples once and derives the finer resolutions' sums from the same chunk sums.  bin_sizes / n_bins_out:
 * host arrays [n_res], coarsest resolution first; out[r] / mask[r] of a job as in cf_bin_job, per resolution; max_cols: an
 * upper bound of `ncols` over the jobs (sizes the launch).  Same results as n_res calls of cf_bin_regions up to the order
 * of the fp32 additions inside a bin (1e-7 relative).  Other configurations / unaligned rows: the per-resolution walk of
 * cf_bin_regions inside the same launch. */
typedef struct cf_bin_job_multi {
    const void* raw;
    long long ld;
    int col0, ncols;
    int flip, reserved;
    float* out[3];
    unsigned char* mask[3];
} cf_bin_job_multi;
int cf_bin_regions_multi(const cf_bin_job_multi* jobs, int n_jobs, int n_feats, int n_res, const int* bin_sizes,
                         const int* n_bins_out, int max_cols, void* stream);

/* ---- resident split, batch gather inside the step graph --------------------------------- */
/* The binned genes of a split, resident in HBM (what chromoformer_amd.data.GeneStore builds with cf_bin_regions):
 * the arrays of cf_batch with a leading gene dimension, pad masks as compact centre rows, one interaction mask per
 * gene (it is the same at every resolution, data.py:200-203). */
typedef struct cf_store {
    long long n_genes;
    const float*   promoter_feats[CF_MAX_RES];      /* [n,1,L,F]                       */
    const float*   pcre_feats[CF_MAX_RES];          /* [n,S,L,F]                       */
    const uint8_t* promoter_mask[CF_MAX_RES];       /* [n,L]    centre query row       */
    const uint8_t* pcre_mask[CF_MAX_RES];           /* [n,S,L]  centre query row       */
    const uint8_t* interaction_mask;                /* [n,T,T]                         */
    const float*   interaction_freq;                /* [n,T,T]                         */
    const void*    labels;                          /* int64 [n] (n_out = 2) or float [n] */
} cf_store;
/* The DataLoader and the per-tensor .cuda() copies of a step (train.py:137-140, 171-177) as ONE capturable launch:
 * copies genes order[cursor[0] * B ... + B) of `store` into the batch buffers `dst` (which must use compact mask
 * rows, stride L, and are written despite the const in cf_batch) and their labels into labels_dst, then advances
 * cursor[0] (a second, one-thread launch).  `order` (int32, batch-major) and `cursor` (int32[4]) are device memory: a
 * replayed graph walks the epoch without any host involvement.  cursor[0] = next batch (zero it whenever a new order is
 * uploaded), cursor[1] = number of batches in `order` (the bound: a call with cursor[0] >= cursor[1] copies nothing, does
 * not advance and sets bit 0 of cursor[2]; a gene index outside [0, store->n_genes) is skipped and sets bit 1),
 * cursor[2] = error flags (sticky until the caller clears them), cursor[3] reserved. */
int cf_gather_batch(cf_handle* h, const cf_store* store, const int* order, int* cursor, const cf_batch* dst,
                    void* labels_dst, void* stream);
/* cf_gather_batch for the batch of a TRAINING step: launches nothing; the cf_forward / cf_forward_train that must follow on the same
 * stream (with `dst` as its batch) copies the genes in the launch that refreshes its tiled weight copies -- neither depends on the other --
 * and its trunk launch advances the cursor: the three launches in front of every step of the training loop become one.  Configurations
 * without the fused trunk kernels: the forward issues the two gather launches itself, as cf_gather_batch would. */
int cf_gather_batch_fwd(cf_handle* h, const cf_store* store, const int* order, int* cursor, const cf_batch* dst,
                        void* labels_dst, void* stream);
/* Appends the step's logits [B, n_out], labels [B] and loss to per-epoch device logs at row cursor[0] - 1 (capturable;
 * what the loop's running metrics, train.py:198-232, read every tenth step instead of cloning tensors every step).
 * The logs must hold cursor[1] rows; nothing is written for a step past the epoch (cursor[2] bit 0). */
int cf_record_step(cf_handle* h, const int* cursor, const float* logits, const void* labels, const float* loss, int B,
                   float* logits_log, void* labels_log, float* loss_log, void* stream);
/* cf_record_step without a launch of its own: the cf_backward_part(parts & 4) that must follow on the same stream writes the rows at the
 * start of the Pairwise + Embedding backward launch (the step's loss is final by then). */
int cf_record_step_bwd(cf_handle* h, const int* cursor, const float* logits, const void* labels, const float* loss, int B,
                       float* logits_log, void* labels_log, float* loss_log, void* stream);

/* Dense attention core with ALL query rows (MultiHeadAttention._attention, modules.py:58-77;
 * PairwiseMultiHeadAttention, modules.py:170-188), head width 64:
 *     O = softmax(masked_fill(Q K^T / sqrt(64), mask, -1e9)) V        per (sequence n, head h)
 * q, k, v, o are [N, L, ld*] row-major with head h in columns [64 h, 64 h + 64) -- i.e. the chunks of the
 * reference's fused projection can be passed in place (pointer offset + ld).  mask = not (qvalid x kvalid) from
 * two byte vectors (1 = real bin; null = all real), or an arbitrary byte mask [N, Lq, Lk] (1 = masked).
 * stats [N, H, Lq, 2] receives the softmax row statistics the backward pass needs.  The training path does not
 * call this (it evaluates the centre query row only); it serves the 800-bin stress configuration and consumers
 * of full-length embeddings. */
typedef struct cf_attn_shape {
    int N, H, Lq, Lk;
    int ldq, ldk, ldv, ldo;
} cf_attn_shape;
int cf_op_attention_fwd(const cf_attn_shape* shape, const float* q, const float* k, const float* v,
                        const unsigned char* qvalid, const unsigned char* kvalid, const unsigned char* mask,
                        float* o, float* stats, void* stream);
/* Its backward: dq, dk, dv in the layouts of q, k, v (only the 64 H columns of the heads are written);
 * delta_ws: workspace of N*H*Lq floats.  No atomics: bit-reproducible. */
int cf_op_attention_bwd(const cf_attn_shape* shape, const float* q, const float* k, const float* v,
                        const unsigned char* qvalid, const unsigned char* kvalid, const unsigned char* mask,
                        const float* o, const float* stats, const float* d_o, float* dq, float* dk, float* dv,
                        float* delta_ws, void* stream);

/* Round 4.  The fused optimiser (cf_reduce_opt_part over CF_BUCKET_PE) keeps the tiled copies of the Embedding + Pairwise weights fresh:
 * its epilogue writes every stepped element in both layouts and forward passes skip the re-tiling of those tensors (no launch in front of
 * a training step, or one that only gathers the batch).  Contract: whoever else writes parameters calls cf_params_changed before the next
 * forward pass, which then re-tiles once (cf_retile_early does it at once, e.g. before replaying a hipGraph captured without it).  cf_bind
 * and the library's separate AdamW launches clear the flag themselves.  cf_keep_tiled(h, 1) returns -1 where the reduction tiles do not
 * cover those tensors (embed n_layers > 1, no gradient buffer bound).  No reference counterpart (train.py:194 steps torch parameters). */
/* cf_gather_batch for a training loop without a launch in front of a step: cf_gather_batch_next stores its arguments and the cf_reduce_opt_part
 * that follows carries the copy of the NEXT step's batch behind its tiles; cf_gather_batch_only copies now (the first step of an epoch).  In
 * both cases the cursor is advanced by the forward pass that follows (its trunk launch).  Replaces train.py:137-140, 171-177 as cf_gather_batch does. */
int cf_gather_batch_next(cf_handle* h, const cf_store* store, const int* order, int* cursor, const cf_batch* dst, void* labels_dst, void* stream);
int cf_gather_batch_only(cf_handle* h, const cf_store* store, const int* order, int* cursor, const cf_batch* dst, void* labels_dst, void* stream);
int cf_keep_tiled(cf_handle* h, int on);
int cf_params_changed(cf_handle* h);
int cf_retile_early(cf_handle* h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CHROMOFORMER_HIP_H */

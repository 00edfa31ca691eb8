// cf_api.hip -- host side of libchromoformer_hip.so: parameter layout, workspace,
// launch sequences for forward / backward / AdamW, and the C ABI of
// include/chromoformer_hip.h.  No torch, no allocation on the hot path.
#include "../../include/chromoformer_hip.h"
#include "cf_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace cf;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
// every kernel launch of the library passes through LAUNCH_CHECK exactly once: the counter behind cf_launch_counts
static thread_local long long g_launches = 0;
#define LAUNCH_CHECK(name)                                                                     \
    do {                                                                                       \
        ++g_launches;                                                                          \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) return fail("launch %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------
// parameter layout (host only)
// ------------------------------------------------------------------------------------
constexpr int kMaxEmbedLayers = 8;

struct PDesc {
    std::string name;
    int ndim;
    int shape[2];
    long long numel, offset;
    bool trainable;
};

static int getenv_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static std::string fmt(const char* f, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}
static void add(std::vector<PDesc>& v, const std::string& name, int d0, int d1, bool trainable) {
    PDesc p;
    p.name = name;
    p.ndim = d1 > 0 ? 2 : 1;
    p.shape[0] = d0;
    p.shape[1] = d1 > 0 ? d1 : 0;
    p.numel = (long long)d0 * (d1 > 0 ? d1 : 1);
    p.offset = -1;
    p.trainable = trainable;
    v.push_back(p);
}
// the seven tensors of an attention block, four/three-chunk self attention (modules.py:16-25)
static void add_self_att(std::vector<PDesc>& v, const std::string& pre, int d_emb, int heads, int dm, bool gate, bool gamma_trains) {
    add(v, pre + "gamma_f", heads, 0, gamma_trains);
    add(v, pre + "w_bias.weight", heads, 2, false);
    add(v, pre + "att.weight", (gate ? 4 : 3) * dm, d_emb, true);
    add(v, pre + "ff.weight", d_emb, dm, true);
    add(v, pre + "ff.bias", d_emb, 0, true);
    add(v, pre + "ln.weight", d_emb, 0, true);
    add(v, pre + "ln.bias", d_emb, 0, true);
}
static void add_ffn(std::vector<PDesc>& v, const std::string& pre, int d_emb, int dff) {
    add(v, pre + "l1.weight", dff, d_emb, true);
    add(v, pre + "l1.bias", dff, 0, true);
    add(v, pre + "l2.weight", d_emb, dff, true);
    add(v, pre + "l2.bias", d_emb, 0, true);
    add(v, pre + "ln.weight", d_emb, 0, true);
    add(v, pre + "ln.bias", d_emb, 0, true);
}

static int check_config(const cf_config& c) {
    // (net.py:277-278 leave d_emb and d_head free; every kernel of this library is written for 128-wide rows -- one MFMA row tile
    //  of 16 x 128 in LDS, eight waves x 16 columns -- so other widths are refused here, by name, instead of failing later)
    // round 5: d_emb = 256 as well, through the stand-alone kernels (row-tile chains, one-sequence attention, layer-by-layer Regulation, the
    // vector-ALU head), which carry the row width as a template parameter; the fused kernels are written for 128-wide rows
    if (c.d_emb != 64 && c.d_emb != 128 && c.d_emb != 256)
        return fail("d_emb = %d is not supported: the HIP path implements d_emb = 128 (the reference's default, net.py:277; every fused kernel) and "
                    "64 / 256 (stand-alone kernels)", c.d_emb);
    if (c.d_emb != 128 && (c.embed_layers != 1 || c.embed_heads > 2 || c.pair_heads > 2))
        return fail("d_emb = %d: one Embedding layer and at most two heads in the Embedding / Pairwise stacks are implemented at this width "
                    "(got embed.n_layers = %d, n_heads = %d / %d)", c.d_emb, c.embed_layers, c.embed_heads, c.pair_heads);
    if (c.d_head < 4 || c.d_head > kHeadGenMaxDH || (c.d_head & 3))
        return fail("d_head = %d is not supported (net.py:278): multiples of 4 in 4..%d (128, the reference's default, runs the matrix-core head "
                    "kernels, other widths a vector-ALU head)", c.d_head, kHeadGenMaxDH);
    if (c.n_feats < 1 || c.n_feats > 8) return fail("n_feats must be in 1..8 (got %d)", c.n_feats);
    if (c.n_out != 1 && c.n_out != 2) return fail("n_out must be 1 or 2");
    if (c.n_res != 3) return fail("exactly 3 resolutions are supported (fc_head is Linear(3*d_emb, .), net.py:327)");
    if (c.i_max < 1 || c.i_max > 16) return fail("i_max must be in 1..16");
    if (c.embed_layers < 1 || c.embed_layers > kMaxEmbedLayers)
        return fail("embed.n_layers must be in 1..%d (got %d); more than one layer runs the all-rows path", kMaxEmbedLayers, c.embed_layers);
    // (heads: 2 is what the fused trunk and the gene-batched attention kernels are written for; 1 and 4 run the stand-alone chain
    //  kernels instantiated for that head count and the one-sequence-per-workgroup attention)
    auto heads_ok = [](int n) { return n == 1 || n == 2 || n == 4; };
    if (!heads_ok(c.embed_heads) || c.embed_dmodel != c.d_emb)
        return fail("embed: n_heads in {1, 2, 4} and d_model = d_emb (net.py:305) are supported (got n_heads = %d, d_model = %d)", c.embed_heads, c.embed_dmodel);
    if (!heads_ok(c.pair_heads) || c.pair_dmodel != c.d_emb)
        return fail("pairwise_interaction: n_heads in {1, 2, 4} and d_model = d_emb = %d are supported -- the Pairwise rows are concatenated with the promoter "
                    "embedding (net.py:361-370), so the two widths must agree (got n_heads = %d, d_model = %d)", c.d_emb, c.pair_heads, c.pair_dmodel);
    if (c.embed_layers > 1 && c.embed_heads != 2)
        return fail("embed: n_layers > 1 (the all-rows path) is implemented for n_heads = 2 only (got n_heads = %d)", c.embed_heads);
    if (c.pair_layers < 1 || 2 * c.pair_layers > kLpMaxSeg) return fail("pairwise_interaction.n_layers must be in 1..%d (got %d)", kLpMaxSeg / 2, c.pair_layers);
    // (the fused Regulation kernels are written for 8 heads x 32; the other shapes run the layer-by-layer kernels, whose attention
    //  stage takes heads and width at run time and whose products are instantiated for both widths)
    // (round 6: the layer-by-layer attention stage, k_attr, takes any head count that divides the width; 1, 2 and 16 are tested beside 4 and 8)
    auto reg_heads_ok = [](int n) { return n == 1 || n == 2 || n == 4 || n == 8 || n == 16; };
    if (!reg_heads_ok(c.reg_heads) || (c.reg_dmodel != 128 && c.reg_dmodel != 256))
        return fail("regulation: n_heads in {1, 2, 4, 8, 16} and d_model in {128, 256} are supported (got n_heads = %d, d_model = %d)", c.reg_heads, c.reg_dmodel);
    if (c.reg_layers < 1 || c.reg_layers > 32) return fail("regulation.n_layers must be in 1..32");
    const int dffs[3] = {c.embed_dff, c.pair_dff, c.reg_dff};
    for (int d : dffs)
        if (d != 128 && d != 256) return fail("d_ff must be 128 or 256 (got %d)", d);
    for (int r = 0; r < c.n_res; ++r)
        if (c.n_bins[r] < 1 || c.n_bins[r] > 1024) return fail("n_bins[%d] must be in 1..1024", r);
    if (c.max_batch < 1 || c.max_batch > 4096) return fail("max_batch must be in 1..4096");
    return 0;
}

static int build_layout(const cf_config& c, std::vector<PDesc>& v, cf_layout& lay) {
    if (check_config(c)) return -1;
    v.clear();
    const int D = c.d_emb;
    for (int r = 0; r < c.n_res; ++r) {
        const std::string pre = fmt("embed.%d.", c.binsizes[r]);
        add(v, pre + "lin_proj.weight", D, c.n_feats, true);
        for (int l = 0; l < c.embed_layers; ++l) {
            const std::string lp = pre + fmt("transformer.layers.%d.", l);
            add_self_att(v, lp + "self_att.", D, c.embed_heads, c.embed_dmodel, false, false);
            add_ffn(v, lp + "ff.", D, c.embed_dff);
        }
    }
    for (int r = 0; r < c.n_res; ++r) {
        const std::string pre = fmt("pairwise_interaction.%d.", c.binsizes[r]);
        add(v, pre + "ln.weight", c.pair_dmodel, 0, false);
        add(v, pre + "ln.bias", c.pair_dmodel, 0, false);
        add(v, pre + "lin_proj_p.weight", c.pair_dmodel, D, true);
        add(v, pre + "lin_proj_pcre.weight", c.pair_dmodel, c.n_feats, true);
        for (int l = 0; l < c.pair_layers; ++l) {
            const std::string lp = pre + fmt("transformer.layers.%d.", l);
            add(v, lp + "self_att.gamma_f", c.pair_heads, 0, false);
            add(v, lp + "self_att.p_att.weight", c.pair_dmodel, c.pair_dmodel, true);
            add(v, lp + "self_att.c_att.weight", 2 * c.pair_dmodel, c.pair_dmodel, true);
            add(v, lp + "self_att.ff.weight", c.pair_dmodel, c.pair_dmodel, true);
            add(v, lp + "self_att.ff.bias", c.pair_dmodel, 0, true);
            add(v, lp + "self_att.ln.weight", c.pair_dmodel, 0, true);
            add(v, lp + "self_att.ln.bias", c.pair_dmodel, 0, true);
            add_ffn(v, lp + "ff.", c.pair_dmodel, c.pair_dff);
        }
    }
    for (int r = 0; r < c.n_res; ++r) {
        const std::string pre = fmt("regulation.%d.", c.binsizes[r]);
        for (int l = 0; l < c.reg_layers; ++l) {
            const std::string lp = pre + fmt("transformer.layers.%d.", l);
            add_self_att(v, lp + "self_att.", D, c.reg_heads, c.reg_dmodel, true, true);
            add_ffn(v, lp + "ff.", D, c.reg_dff);
        }
    }
    add(v, "fc_head.0.weight", c.d_head, 3 * D, true);
    add(v, "fc_head.0.bias", c.d_head, 0, true);
    add(v, "fc_head.2.weight", c.n_out, c.d_head, true);
    add(v, "fc_head.2.bias", c.n_out, 0, true);

    // Offsets: trainable tensors first -- Embedding, Pairwise (state_dict order), then the Regulation layers BELOW reg_layers / 2 of every
    // resolution, then the layers from there up, then fc_head (three adjacent gradient buckets: cf_grad_bucket) --, never-trained ones behind.
    // The TABLE stays in state_dict order; only where a tensor lies in the flat buffers follows the order its gradient becomes complete in.
    const auto group = [&](const PDesc& p) {
        if (!p.trainable) return 4;
        if (p.name.rfind("fc_head.", 0) == 0) return 3;
        if (p.name.rfind("regulation.", 0) != 0) return 0;
        const size_t at = p.name.find(".transformer.layers.");
        const int l = at == std::string::npos ? 0 : atoi(p.name.c_str() + at + 20);
        return l < c.reg_layers / 2 ? 1 : 2;
    };
    long long off = 0, elems = 0;
    for (int pass = 0; pass < 5; ++pass) {
        for (auto& p : v) {
            if (group(p) != pass) continue;
            p.offset = off;
            off += (p.numel + 3) / 4 * 4;     // 16-byte aligned tensors
            elems += p.numel;
        }
        if (pass == 3) lay.n_active = off;
    }
    lay.n_tensors = (int)v.size();
    lay.n_total = off;
    lay.n_elems = elems;
    return 0;
}

// ------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------
struct CentreBuf {   // one centre-row attention layer of one resolution
    float *q, *qt, *p, *w, *xbar, *a, *xh1, *rs1, *y1, *hdn, *xh2, *rs2, *out, *xin;
    float *dt2, *dpre1, *dt1, *da, *dxbar, *dqt, *du, *dq, *dx, *partial;
};
struct RegBuf {
    float *qkvg, *p, *a, *xh1, *rs1, *y1, *hdn, *xh2, *rs2;
    float *dt2, *dpre1, *dt1, *da, *dqkvg, *partial, *dgam;
    float *hq, *dy1;
};
struct WsEntry {
    std::string name;
    size_t n, off;
};

struct cf_handle {
    cf_config cfg;
    std::vector<PDesc> table;
    std::map<std::string, int> index;
    cf_layout lay;
    float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
    float* tiled = nullptr;              // tiled copy of the Linear weights (forward products), same offsets
    float* tiledT = nullptr;             // tiled copy of the transposed Regulation weights (backward products), same offsets
    bool reg8 = false;                   // Regulation stack on the 512-thread kernels of cf_reg8.h
    bool reg_row0 = true;                // ... whose last layer computes only what token 0 of its output needs (CF_REG_ROW0=0: all rows, the cross-check)
    bool keep_tiled_ok = false;          // (build_tables: the reduction tiles cover every tensor of that group)
    bool keep_tiled = false;             // cf_keep_tiled: the fused optimiser keeps the Embedding + Pairwise tiled copies fresh, forward passes do not re-tile them
    int* tiled_map = nullptr;            // [bucket_split / 4]: where each flat float4 of the Embedding + Pairwise range lies in the tiled buffer (k_adamw_tiled)
    bool tiled_pe_fresh = false;         // ... and they ARE fresh (cleared by whatever else writes parameters: cf_bind, cf_params_changed, the separate AdamW launches)
    // Embedding stack over ALL promoter bins (cf_embed_full.h + the dense transformer layer): used by the model path when
    // embed.n_layers > 1 and by cf_embed_full; device buffers outside the arena, allocated on first need
    struct EmbedDense {
        bool ready = false;
        int B = 0;
        float* x[kMaxRes][kMaxEmbedLayers + 1] = {};      // token embeddings / layer outputs [B, L, 128]
        float* ws[kMaxRes][kMaxEmbedLayers] = {};          // dense-layer workspaces (training size)
        float* dy[kMaxRes][3] = {};                        // gradient ping-pong [B, L, 128]
        float* lp_partial[kMaxRes] = {};
        uint8_t* valid[kMaxRes] = {};
        float* tables = nullptr;                           // 4 MiB of tile tables for the backward pass
        std::vector<void*> owned;
    } ed;
    bool embed_dense = false;            // the training path goes through it (embed.n_layers > 1)
    RetileUnit* retile_units = nullptr;
    int n_retile = 0, n_retile_early = 0;      // all units | the leading ones a forward pass needs at once (Embedding + Pairwise)
    TrunkResDev* trunk_tab = nullptr;          // fused centre-row trunk (cf_trunk.h): device table, one entry per resolution
    bool trunk = false;                        // the Embedding + Pairwise stage runs as k_trunk_fwd / k_trunk_bwd (CF_TRUNK=0: the stand-alone kernels)
    size_t trunk_smem_bytes = 0;
    bool head_deferred = false;                // cf_forward(save = 2) left the head to cf_backward_part (k_head_train)
    bool head_done = false;                    // cf_forward_train ran head forward + loss + head backward at the tail of the Regulation launch
    bool head_loss_due = false;                // ... and the mean loss is still to be summed (by the Regulation backward launch)
    bool pend_record = false;                  // cf_record_step_bwd: the step log rides in the trunk's backward launch
    RecordArgs pend_rec;
    bool pend_gnext = false;                   // cf_gather_batch_next: the NEXT step's gather rides in this step's reduction launch (cf_reduce_opt_part)
    GatherArgs pend_gn;
    int pend_gn_n = 0;
    int* adv_next = nullptr;                   // a batch gathered without advancing the cursor: the next forward pass advances it (trunk launch)
    bool pend_gather = false;                  // cf_gather_batch_fwd: the gather of the step shares a launch with the next forward's prologue
    const void* pend_key = nullptr;            // the batch (its first feature array) the pending gather / cursor advance above belongs to: a forward
                                               // pass over ANOTHER batch (validation between two steps of a fed epoch) must not consume them
    GatherArgs pend_ga;
    int pend_ga_n = 0;
    bool head_ride = true;                     // CF_HEAD_RIDE=0 (read at cf_create): the head stays a launch of its own (k_head_train)
    HeadRide ride;
    int* head_cnt = nullptr;
    unsigned long long ride_launches = 0;      // launches that added to head_cnt since it was last zero (ride_tick)
    unsigned long long ride_reset_every = 1ull << 28;
    std::vector<hipStream_t> ride_streams;     // every stream a head-ride launch of this handle was issued on (ride_tick orders its reset against all of them)
    hipEvent_t ride_ev = nullptr;
    float* deferred_logits_user = nullptr;
    // riders of the next k_trunk_bwd launch (cf_rider_arm): leading Regulation weight-gradient tiles with AdamW in their epilogues
    struct Rider {
        bool armed = false;
        int max_tiles = 0, done = 0;           // done: tiles the last trunk launch took (the reduction call that follows skips them)
        long long step = -1;
        AdamFuse o;
    } rider;
    int xcd_reduce = 0;                        // XCD-aware order of the weight-gradient tiles (measured slower: cf_kernels.h, xcd_tile)
    int xcd_reduce_opt = 0;                    // ... in the fused reduction + AdamW launch it paid in round 3 (0.568 -> 0.563 ms); on the round-6 kernels (riders with
                                               // cached accesses) the table order wins: 0.5012 -> 0.4892 ms (profiles/r06n_env_ab.txt); CF_XCD_REDUCE=1 forces it on
    int defer_retile = 1;                      // Regulation + head units ride in the Embedding layer's chain launch (CF_DEFER_RETILE=0: all in the prologue)
    // workspace
    float* arena = nullptr;
    size_t arena_floats = 0;
    std::vector<WsEntry> ws;
    std::map<std::string, int> ws_index;
    std::string ws_names;
    bool planning = true;
    // stage buffers
    float *pe[kMaxRes], *pet[kMaxRes];
    float *pe2[kMaxRes], *pet2[kMaxRes];      // padded layouts of the gene-batched attention kernel (cf_attc2.h)
    bool attc2 = false;
    bool attc1 = true;                        // one-region launches on the vector-ALU kernel (CF_ATTC1=0: k_attc2<., 1>)
    AdamHyper* hyper = nullptr;               // step-dependent AdamW scalars for the graph-replayed optimiser launch
    std::vector<hipEvent_t> sync_ev;          // cf_stream_wait
    size_t sync_ev_used = 0;
    int attc_cap = 64;                        // most workgroups per resolution for which attc2 trades regions per workgroup for parallelism
    int xcd_map = 1;                          // XCD-aware placement of the Regulation workgroups (CF_XCD_MAP=0 turns it off)
    int n_wg_r = 0, n_cs_r = 0;               // leading entries of wg_tiles / cs_tiles that belong to the Regulation + head bucket
    long long bucket_split = 0;               // flat offset of the first Regulation parameter (bucket boundary)
    long long bucket_split_hi = 0;            // ... of the first parameter of the upper Regulation layers (CF_BUCKET_REG_HI = [this, n_active))
    int n_wg_hi = 0, n_cs_hi = 0;             // tiles of CF_BUCKET_REG_HI at the front of the tables
    int n_wg_short = 0;                       // ... and, at the front of those, the SHORT ones (64 reduction rows: the head's, the last Regulation layer's one-row gradients)
    float *featc[kMaxRes], *ex0[kMaxRes], *edx0[kMaxRes], *edout[kMaxRes];
    CentreBuf E[kMaxRes];
    float *xp0[kMaxRes], *dxp0[kMaxRes], *resid[kMaxRes];
    std::vector<CentreBuf> P[kMaxRes];
    std::vector<float*> Rx[kMaxRes], dRx[kMaxRes];
    std::vector<RegBuf> R[kMaxRes];
    float *hin, *h1, *logits, *dlogits, *dh1, *dhin, *loss, *loss_part, *tdbg;
    // deferred-gradient tile tables
    WgTile* wg_tiles = nullptr;
    int n_wg = 0;
    CsTile* cs_tiles = nullptr;
    int n_cs = 0;
    LpJob* lp_jobs = nullptr;
    int n_lp = 0;
    RegLayerDev* reg_tab = nullptr;      // fused Regulation stack (one workgroup per gene), when it fits in LDS
    bool reg_fused = false;
    float *lp_part_e[kMaxRes], *lp_part_p[kMaxRes];
    int last_fwd_B = 0;
    int n_fwd = 0, n_bwd = 0, n_opt = 0;
    int n_cu = 256;                     // hipDeviceAttributeMultiprocessorCount of the device the handle was created on
    double wg_flops_per_gene = 0.0;     // 2*M*N*K summed over the weight-gradient jobs, per gene
    // optional HIP-event timing of one of the eagerly launched kernels
    std::string timed;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    // captured launch sequences
    bool capturing = false;

    // A captured sequence is replayed as [graph piece 1] -> [the timed kernel, launched eagerly between two HIP
    // events] -> [graph piece 2] when it contains the kernel selected with cf_timing_select (event-record nodes
    // inside a graph cost ~60 us per replay on this runtime, an eager launch between two graphs costs nothing
    // measurable), and as one graph otherwise.
    struct Hole {
        const void* func = nullptr;
        dim3 grid, block;
        size_t smem = 0;
        RegArgs args;
    };
    struct Replay {
        hipGraphExec_t first = nullptr, second = nullptr;
        bool has_hole = false;
        Hole hole;
        // launch accounting of the captured sequence (cf_launch_counts must describe ONE step under replay as well): the forward
        // launches it holds (-1: none), whether it starts a backward pass (head: the per-step counters restart), the backward launches it holds
        int n_fwd = -1, n_bwd = 0;
        bool starts_bwd = false;
    };
    std::vector<Replay> replays;
    Replay cap;                    // under construction

    void time_mark(const char* name, hipStream_t st) {
        if (timed.empty() || capturing || timed != name) return;
        if (ev_used >= 16384) return;      // nobody is reading: stop recording rather than grow without bound
        if (ev_used == ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            ev.push_back(e);
        }
        (void)hipEventRecord(ev[ev_used++], st);
    }

    float* ws_get(const std::string& name, size_t n) {
        n = (n + 3) / 4 * 4;
        if (planning) {
            ws_index[name] = (int)ws.size();
            ws.push_back(WsEntry{name, n, arena_floats});
            arena_floats += n;
            return nullptr;
        }
        const WsEntry& e = ws[ws_index.at(name)];
        return arena + e.off;
    }
    const float* P_(const std::string& name, long long extra = 0) const { return params + table[index.at(name)].offset + extra; }
    const float* T_(const std::string& name, long long extra = 0) const { return tiled + table[index.at(name)].offset + extra; }
    const float* TT_(const std::string& name) const { return tiledT + table[index.at(name)].offset; }
    float* G_(const std::string& name, long long extra = 0) const { return grads + table[index.at(name)].offset + extra; }
};

static void plan_centre(cf_handle* h, CentreBuf& b, const std::string& pre, size_t N, int L, int dff, bool own_out, bool own_xin, int nh) {
    const size_t kD = h->cfg.d_emb;      // (row width: shadows cf::kD in this function)
    const size_t tiles = (N + kTile - 1) / kTile;
    b.q = h->ws_get(pre + "q", N * kD);
    b.qt = h->ws_get(pre + "qt", N * nh * kD);
    b.p = h->ws_get(pre + "p", N * nh * L);
    b.w = h->ws_get(pre + "w", N * nh * 8);
    b.xbar = h->ws_get(pre + "xbar", N * nh * kD);
    b.a = h->ws_get(pre + "a", N * kD);
    b.xh1 = h->ws_get(pre + "xh1", N * kD);
    b.rs1 = h->ws_get(pre + "rs1", N);
    b.y1 = h->ws_get(pre + "y1", N * kD);
    b.hdn = h->ws_get(pre + "hdn", N * dff);
    b.xh2 = h->ws_get(pre + "xh2", N * kD);
    b.rs2 = h->ws_get(pre + "rs2", N);
    b.out = own_out ? h->ws_get(pre + "out", N * kD) : nullptr;
    b.xin = own_xin ? h->ws_get(pre + "xin", N * kD) : nullptr;
    const std::string d = "d" + pre;
    b.dt2 = h->ws_get(d + "t2", N * kD);
    b.dpre1 = h->ws_get(d + "pre1", N * dff);
    b.dt1 = h->ws_get(d + "t1", N * kD);
    b.da = h->ws_get(d + "a", N * kD);
    b.dxbar = h->ws_get(d + "xbar", N * nh * kD);
    b.dqt = h->ws_get(d + "qt", N * nh * kD);
    b.du = h->ws_get(d + "u", N * nh * 8);
    b.dq = h->ws_get(d + "q", N * kD);
    b.dx = h->ws_get(d + "x", N * kD);
    b.partial = h->ws_get(d + "partial", std::max(tiles, (size_t)h->cfg.max_batch) * post_partial_width(dff, (int)kD));      // (the fused trunk writes one row per gene)
}

// executed twice: once to size the arena, once to hand out pointers
static void plan_workspace(cf_handle* h) {
    const cf_config& c = h->cfg;
    const size_t kD = c.d_emb;      // (row width: shadows cf::kD in this function)
    const size_t MB = c.max_batch, S = c.i_max, T = S + 1;
    const size_t NE = MB, NP = MB * S, NR = MB * T;
    for (int r = 0; r < c.n_res; ++r) {
        const int L = c.n_bins[r];
        h->pe[r] = h->ws_get(fmt("pe%d", r), (size_t)L * kD);
        h->pet[r] = h->ws_get(fmt("pet%d", r), (size_t)L * kD);
        h->pe2[r] = h->ws_get(fmt("pe2_%d", r), (size_t)attc2_lpad(L) * kD);
        h->pet2[r] = h->ws_get(fmt("pet2_%d", r), (size_t)kD * attc2_lt(L));
        h->featc[r] = h->ws_get(fmt("E%d.featc", r), NE * 8);
        h->lp_part_e[r] = h->ws_get(fmt("dE%d.lp_partial", r), ((MB + kLpGenes - 1) / kLpGenes) * kD * 8);
        h->lp_part_p[r] = h->ws_get(fmt("dP%d.lp_partial", r), ((MB + kLpGenes - 1) / kLpGenes) * kD * 8);
        h->ex0[r] = h->ws_get(fmt("E%d.x0", r), NE * kD);
        plan_centre(h, h->E[r], fmt("E%d.", r), NE, L, c.embed_dff, false, false, c.embed_heads);
        h->edout[r] = h->ws_get(fmt("dE%d.out", r), NE * kD);
        h->xp0[r] = h->ws_get(fmt("P%d.xp0", r), NE * kD);
        h->dxp0[r] = h->ws_get(fmt("dP%d.xp0", r), NE * kD);
        h->resid[r] = h->ws_get(fmt("dE%d.resid", r), NE * kD);
        h->P[r].resize(c.pair_layers);
        for (int l = 0; l < c.pair_layers; ++l)
            plan_centre(h, h->P[r][l], fmt("P%d.%d.", r, l), NP, L, c.pair_dff, l + 1 < c.pair_layers, l == 0, c.pair_heads);
        h->Rx[r].resize(c.reg_layers + 1);
        h->dRx[r].resize(c.reg_layers + 1);
        for (int l = 0; l <= c.reg_layers; ++l) {
            h->Rx[r][l] = h->ws_get(fmt("R%d.x%d", r, l), NR * kD);
            h->dRx[r][l] = h->ws_get(fmt("dR%d.x%d", r, l), NR * kD);
        }
        h->R[r].resize(c.reg_layers);
        for (int l = 0; l < c.reg_layers; ++l) {
            RegBuf& b = h->R[r][l];
            const std::string pre = fmt("R%d.%d.", r, l), d = "d" + pre;
            const int dff = c.reg_dff;
            const int RH = c.reg_heads, RDm = c.reg_dmodel, RW = 4 * RDm;
            b.qkvg = h->ws_get(pre + "qkvg", NR * RW);
            b.p = h->ws_get(pre + "p", MB * RH * T * T);
            b.a = h->ws_get(pre + "a", NR * RDm);
            b.xh1 = h->ws_get(pre + "xh1", NR * kD);
            b.rs1 = h->ws_get(pre + "rs1", NR);
            b.y1 = h->ws_get(pre + "y1", NR * kD);
            b.hdn = h->ws_get(pre + "hdn", NR * dff);
            b.xh2 = h->ws_get(pre + "xh2", NR * kD);
            b.rs2 = h->ws_get(pre + "rs2", NR);
            b.dt2 = h->ws_get(d + "t2", NR * kD);
            b.dpre1 = h->ws_get(d + "pre1", NR * dff);
            b.dt1 = h->ws_get(d + "t1", NR * kD);
            b.da = h->ws_get(d + "a", NR * RDm);
            b.dqkvg = h->ws_get(d + "qkvg", NR * RW);
            b.partial = h->ws_get(d + "partial", std::max((NR + kTile - 1) / kTile, MB) * post_partial_width(dff, (int)kD));
            b.dgam = h->ws_get(d + "gam", MB * RH);
            b.hq = h->ws_get(pre + "hq", MB * RH * kHqFloats);
            b.dy1 = h->ws_get(d + "y1", NR * kD);
        }
    }
    h->hin = h->ws_get("H.in", MB * 3 * kD);
    h->h1 = h->ws_get("H.h1", MB * c.d_head);
    h->logits = h->ws_get("H.logits", MB * c.n_out);
    h->dlogits = h->ws_get("dH.logits", MB * c.n_out);
    h->dh1 = h->ws_get("dH.h1", MB * c.d_head);
    h->dhin = h->ws_get("dH.in", MB * 3 * kD);
    h->loss = h->ws_get("H.loss", 4);
    h->loss_part = h->ws_get("H.loss_part", MB + 1);      // (per 16-gene tile; per gene in the generic-width head)
    h->head_cnt = reinterpret_cast<int*>(h->ws_get("H.cnt", MB + 1));      // arrivals per gene (cf_head_ride.h): monotonic, see ride_tick
    h->ride_reset_every = (unsigned long long)std::max(1, getenv_int("CF_RIDE_RESET_EVERY", 1 << 28));
    h->tdbg = h->ws_get("reg_tdbg", 2 * 16 * 64);      // shader-clock stamps (uint64) of the fused Regulation kernels
}

// ------------------------------------------------------------------------------------
// deferred gradient tables
// ------------------------------------------------------------------------------------
struct WgJob {
    WgSeg seg[4];
    int nseg;
    float* C;
    int ldc, Nn, Kk;
};
static void push_wg(std::vector<WgTile>& out, const WgJob& j) {
    for (int n0 = 0; n0 < j.Nn; n0 += 64)
        for (int k0 = 0; k0 < j.Kk; k0 += kWgTk) {
            WgTile t;
            memset(&t, 0, sizeof t);
            for (int s = 0; s < j.nseg; ++s) t.seg[s] = j.seg[s];
            t.nseg = j.nseg;
            t.C = j.C;
            t.ldc = j.ldc;
            t.Nn = j.Nn;
            t.Kk = j.Kk;
            t.n0 = n0;
            t.k0 = k0;
            t.toff = -1;
            out.push_back(t);
        }
}
static WgJob wg1(const float* A, int lda, const float* B, int ldb, int rpg, float* C, int ldc, int Nn, int Kk) {
    WgJob j;
    memset(&j, 0, sizeof j);
    j.seg[0] = WgSeg{A, B, lda, ldb, rpg};
    j.nseg = 1;
    j.C = C;
    j.ldc = ldc;
    j.Nn = Nn;
    j.Kk = Kk;
    return j;
}
static void push_cs(std::vector<CsTile>& out, const float* src, int ld, int ncols, int rpg, int div, float* dst, const float* src2 = nullptr) {
    for (int c0 = 0; c0 < ncols; c0 += 64) out.push_back(CsTile{src, dst, ld, ncols, c0, rpg, div, src2});
}
// the bias / LayerNorm gradients carried by one post-chain partial buffer
// (rows of the partial buffer: one per 16-row tile, M = ceil(rpg * batch / 16); one per gene with the fused trunk: rpg = div = 1)
static void push_post_cs(std::vector<CsTile>& out, const cf_handle* h, const float* part, int dff, int rpg,
                         const std::string& att_pre, const std::string& ff_pre, int div = kTile) {
    const int kD = h->cfg.d_emb;      // (row width: shadows cf::kD in this function)
    const int pw = post_partial_width(dff, kD);
    push_cs(out, part + 0, pw, kD, rpg, div, h->G_(ff_pre + "ln.weight"));
    push_cs(out, part + kD, pw, kD, rpg, div, h->G_(ff_pre + "ln.bias"));
    push_cs(out, part + 2 * kD, pw, kD, rpg, div, h->G_(ff_pre + "l2.bias"));
    push_cs(out, part + 3 * kD, pw, dff, rpg, div, h->G_(ff_pre + "l1.bias"));
    push_cs(out, part + 3 * kD + dff, pw, kD, rpg, div, h->G_(att_pre + "ln.weight"));
    push_cs(out, part + 4 * kD + dff, pw, kD, rpg, div, h->G_(att_pre + "ln.bias"));
    push_cs(out, part + 5 * kD + dff, pw, kD, rpg, div, h->G_(att_pre + "ff.bias"));
}
// weight gradients of one centre-row layer (q / k / v projections, out-projection, FFN)
static void push_centre_wg(std::vector<WgTile>& out, const cf_handle* h, const CentreBuf& b, const float* xin, int ldxin,
                           int rpg, int dff, float* gWq, float* gWk, float* gWv, const std::string& att_pre,
                           const std::string& ff_pre, int nh) {
    const int kD = h->cfg.d_emb;      // (row width: shadows cf::kD in this function)
    const int dh = kD / nh, qw = nh * kD;
    push_wg(out, wg1(b.dq, kD, xin, ldxin, rpg, gWq, kD, kD, kD));
    for (int hd = 0; hd < nh; ++hd) {
        push_wg(out, wg1(b.q + hd * dh, kD, b.dqt + hd * kD, qw, rpg, gWk + (size_t)hd * dh * kD, kD, dh, kD));
        push_wg(out, wg1(b.da + hd * dh, kD, b.xbar + hd * kD, qw, rpg, gWv + (size_t)hd * dh * kD, kD, dh, kD));
    }
    push_wg(out, wg1(b.dt1, kD, b.a, kD, rpg, h->G_(att_pre + "ff.weight"), kD, kD, kD));
    push_wg(out, wg1(b.dpre1, dff, b.y1, kD, rpg, h->G_(ff_pre + "l1.weight"), kD, dff, kD));
    push_wg(out, wg1(b.dt2, kD, b.hdn, dff, rpg, h->G_(ff_pre + "l2.weight"), dff, kD, dff));
}

// the fused Regulation kernels (cf_reg8.h): forward with / without the activation saves, backward; per FFN width
static const void* reg_kernel(bool bwd, int dff, bool save = true) {
    if (!bwd) {
        if (save) return dff == 128 ? (const void*)k_reg8_fwd<128, true> : (const void*)k_reg8_fwd<256, true>;
        return dff == 128 ? (const void*)k_reg8_fwd<128, false> : (const void*)k_reg8_fwd<256, false>;
    }
    return dff == 128 ? (const void*)k_reg8_bwd<128> : (const void*)k_reg8_bwd<256>;
}

static int build_reg_table(cf_handle* h) {
    const cf_config& c = h->cfg;
    if (h->reg_fused) {
        std::vector<RegLayerDev> rt;
        for (int r = 0; r < c.n_res; ++r)
            for (int l = 0; l < c.reg_layers; ++l) {
                const std::string lp = fmt("regulation.%d.transformer.layers.%d.", c.binsizes[r], l);
                const RegBuf& b = h->R[r][l];
                RegLayerDev d;
                d.watt = h->P_(lp + "self_att.att.weight");
                d.watt_t = h->T_(lp + "self_att.att.weight");
                d.wo_t = h->T_(lp + "self_att.ff.weight");
                d.w1_t = h->T_(lp + "ff.l1.weight");
                d.w2_t = h->T_(lp + "ff.l2.weight");
                d.watt_tt = h->TT_(lp + "self_att.att.weight");
                d.wo_tt = h->TT_(lp + "self_att.ff.weight");
                d.w1_tt = h->TT_(lp + "ff.l1.weight");
                d.w2_tt = h->TT_(lp + "ff.l2.weight");
                d.gamma = h->P_(lp + "self_att.gamma_f");
                d.wo = h->P_(lp + "self_att.ff.weight");
                d.bo = h->P_(lp + "self_att.ff.bias");
                d.g1 = h->P_(lp + "self_att.ln.weight");
                d.be1 = h->P_(lp + "self_att.ln.bias");
                d.w1 = h->P_(lp + "ff.l1.weight");
                d.b1 = h->P_(lp + "ff.l1.bias");
                d.w2 = h->P_(lp + "ff.l2.weight");
                d.b2 = h->P_(lp + "ff.l2.bias");
                d.g2 = h->P_(lp + "ff.ln.weight");
                d.be2 = h->P_(lp + "ff.ln.bias");
                d.xin = h->Rx[r][l];
                d.qkvg = b.qkvg;
                d.p = b.p;
                d.a = b.a;
                d.xh1 = b.xh1;
                d.rs1 = b.rs1;
                d.y1 = b.y1;
                d.hdn = b.hdn;
                d.xh2 = b.xh2;
                d.rs2 = b.rs2;
                d.xout = h->Rx[r][l + 1];
                d.dxout = h->dRx[r][l + 1];
                d.dt2 = b.dt2;
                d.dpre1 = b.dpre1;
                d.dt1 = b.dt1;
                d.da = b.da;
                d.dqkvg = b.dqkvg;
                d.dxin = h->dRx[r][l];
                d.partial = b.partial;
                d.dgam = b.dgam;
                d.hq = b.hq;
                d.dy1 = b.dy1;
                rt.push_back(d);
            }
        if (h->reg_tab) (void)hipFree(h->reg_tab);
        HIP_TRY(hipMalloc(&h->reg_tab, rt.size() * sizeof(RegLayerDev)));
        HIP_TRY(hipMemcpy(h->reg_tab, rt.data(), rt.size() * sizeof(RegLayerDev), hipMemcpyHostToDevice));
    }
    return 0;
}

static int build_tables(cf_handle* h) {
    const cf_config& c = h->cfg;
    const int kD = c.d_emb;      // (row width: shadows cf::kD in this function)
    const int S = c.i_max, T = S + 1, F = c.n_feats;
    // two gradient buckets: `wg` / `cs` take Embedding + Pairwise (ready after the whole backward chain), `wgR` / `csR`
    // the Regulation stacks and the head (ready after k_reg_bwd, i.e. before Pairwise + Embedding backward starts)
    std::vector<WgTile> wg, wgHi, wgLo, wgShort;
    std::vector<CsTile> cs, csHi, csLo;
    std::vector<LpJob> lpj;
    for (int r = 0; r < c.n_res; ++r) {
        const int bs = c.binsizes[r];
        if (!h->embed_dense) {   // Embedding (the all-rows path writes its gradients itself)
            const std::string pre = fmt("embed.%d.", bs), lp = pre + "transformer.layers.0.";
            const CentreBuf& b = h->E[r];
            LpJob j;
            memset(&j, 0, sizeof j);
            j.seg[0] = WgSeg{h->edx0[r], h->featc[r], kD, 8, 1};
            j.seg[1] = WgSeg{b.dxbar, b.w, kD, 8, c.embed_heads};      // (rows of [N, heads, .] arrays: heads per gene)
            j.seg[2] = WgSeg{b.qt, b.du, kD, 8, c.embed_heads};
            j.nseg = 3;
            j.partial = h->lp_part_e[r];
            j.F = F;
            lpj.push_back(j);
            push_cs(cs, j.partial, kD * F, kD * F, 1, kLpGenes, h->G_(pre + "lin_proj.weight"));
            float* gatt = h->G_(lp + "self_att.att.weight");
            push_centre_wg(wg, h, b, h->ex0[r], kD, 1, c.embed_dff, gatt, gatt + (size_t)kD * kD, gatt + (size_t)2 * kD * kD,
                           lp + "self_att.", lp + "ff.", c.embed_heads);
            if (h->trunk) push_post_cs(cs, h, b.partial, c.embed_dff, 1, lp + "self_att.", lp + "ff.", 1);
            else push_post_cs(cs, h, b.partial, c.embed_dff, 1, lp + "self_att.", lp + "ff.");
        }
        {   // Pairwise
            const std::string pre = fmt("pairwise_interaction.%d.", bs);
            push_wg(wg, wg1(h->dxp0[r], kD, h->Rx[r][0], T * kD, 1, h->G_(pre + "lin_proj_p.weight"), kD, kD, kD));
            LpJob j;
            memset(&j, 0, sizeof j);
            // lin_proj_pcre collects two terms per layer (pair_layers <= 8 -> <= kLpMaxSeg segments)
            int ns = 0;
            for (int l = 0; l < c.pair_layers; ++l) {
                j.seg[ns++] = WgSeg{h->P[r][l].dxbar, h->P[r][l].w, kD, 8, c.pair_heads * S};
                j.seg[ns++] = WgSeg{h->P[r][l].qt, h->P[r][l].du, kD, 8, c.pair_heads * S};
            }
            j.nseg = ns;
            j.partial = h->lp_part_p[r];
            j.F = F;
            lpj.push_back(j);
            push_cs(cs, j.partial, kD * F, kD * F, 1, kLpGenes, h->G_(pre + "lin_proj_pcre.weight"));
            for (int l = 0; l < c.pair_layers; ++l) {
                const std::string lp = pre + fmt("transformer.layers.%d.", l);
                const CentreBuf& b = h->P[r][l];
                const float* xin = l == 0 ? b.xin : h->P[r][l - 1].out;
                float* gc = h->G_(lp + "self_att.c_att.weight");
                push_centre_wg(wg, h, b, xin, kD, S, c.pair_dff, h->G_(lp + "self_att.p_att.weight"), gc, gc + (size_t)kD * kD,
                               lp + "self_att.", lp + "ff.", c.pair_heads);
                if (h->trunk) push_post_cs(cs, h, b.partial, c.pair_dff, 1, lp + "self_att.", lp + "ff.", 1);
                else push_post_cs(cs, h, b.partial, c.pair_dff, S, lp + "self_att.", lp + "ff.");
            }
        }
        for (int l = 0; l < c.reg_layers; ++l) {   // Regulation: the upper half of the stack (complete first in the backward pass) and the lower one
            std::vector<WgTile>& wgR = l >= c.reg_layers / 2 ? wgHi : wgLo;
            std::vector<CsTile>& csR = l >= c.reg_layers / 2 ? csHi : csLo;
            const std::string lp = fmt("regulation.%d.transformer.layers.%d.", bs, l);
            const RegBuf& b = h->R[r][l];
            const int dff = c.reg_dff;
            const int RDm = c.reg_dmodel, RW = 4 * RDm;
            if (l + 1 == c.reg_layers && h->reg_row0 && h->reg8) {
                // The last layer, reduced to what token 0 of its output needs (cf_reg8.h: b_run_row0 / b_run_kv_rows): every gradient above the attention
                // -- out-projection, FFN, the query and gate quarters of the input projection -- has ONE live row per gene, the rows of tokens 1 .. T - 1
                // are zeros the kernel writes.  Their reductions walk that row alone (rows_per_gene = 1 at a stride of T rows: 64 reduction rows
                // instead of 576, a ninth of the operand bytes); the key and value quarters keep all rows.  Same sums: what is left out are exact zeros.
                float* ga = h->G_(lp + "self_att.att.weight");
                push_wg(wgShort, wg1(b.dqkvg, T * RW, h->Rx[r][l], T * kD, 1, ga, kD, RDm, kD));                                                            // q
                push_wg(wgR, wg1(b.dqkvg + RDm, RW, h->Rx[r][l], kD, T, ga + (size_t)RDm * kD, kD, 2 * RDm, kD));                                           // k | v
                push_wg(wgShort, wg1(b.dqkvg + 3 * RDm, T * RW, h->Rx[r][l], T * kD, 1, ga + (size_t)3 * RDm * kD, kD, RDm, kD));                            // gate
                push_wg(wgShort, wg1(b.dt1, T * kD, b.a, T * RDm, 1, h->G_(lp + "self_att.ff.weight"), RDm, kD, RDm));
                push_wg(wgShort, wg1(b.dpre1, T * dff, b.y1, T * kD, 1, h->G_(lp + "ff.l1.weight"), kD, dff, kD));
                push_wg(wgShort, wg1(b.dt2, T * kD, b.hdn, T * dff, 1, h->G_(lp + "ff.l2.weight"), dff, kD, dff));
            } else {
            push_wg(wgR, wg1(b.dqkvg, RW, h->Rx[r][l], kD, T, h->G_(lp + "self_att.att.weight"), kD, RW, kD));
            push_wg(wgR, wg1(b.dt1, kD, b.a, RDm, T, h->G_(lp + "self_att.ff.weight"), RDm, kD, RDm));
            push_wg(wgR, wg1(b.dpre1, dff, b.y1, kD, T, h->G_(lp + "ff.l1.weight"), kD, dff, kD));
            push_wg(wgR, wg1(b.dt2, kD, b.hdn, dff, T, h->G_(lp + "ff.l2.weight"), dff, kD, dff));
            }
            if (h->reg8) {            // column sums straight from the row-level arrays the backward kernel writes anyway
                const std::string ap = lp + "self_att.", fp = lp + "ff.";
                push_cs(csR, h->dRx[r][l + 1], kD, kD, T, 1, h->G_(fp + "ln.weight"), b.xh2);
                push_cs(csR, h->dRx[r][l + 1], kD, kD, T, 1, h->G_(fp + "ln.bias"));
                push_cs(csR, b.dt2, kD, kD, T, 1, h->G_(fp + "l2.bias"));
                push_cs(csR, b.dpre1, dff, dff, T, 1, h->G_(fp + "l1.bias"));
                push_cs(csR, b.dy1, kD, kD, T, 1, h->G_(ap + "ln.weight"), b.xh1);
                push_cs(csR, b.dy1, kD, kD, T, 1, h->G_(ap + "ln.bias"));
                push_cs(csR, b.dt1, kD, kD, T, 1, h->G_(ap + "ff.bias"));
            } else {
                push_post_cs(csR, h, b.partial, dff, T, lp + "self_att.", lp + "ff.");
            }
            push_cs(csR, b.dgam, c.reg_heads, c.reg_heads, 1, 1, h->G_(lp + "self_att.gamma_f"));
        }
    }
    push_wg(wgShort, wg1(h->dh1, c.d_head, h->hin, 3 * kD, 1, h->G_("fc_head.0.weight"), 3 * kD, c.d_head, 3 * kD));
    push_wg(wgShort, wg1(h->dlogits, c.n_out, h->h1, c.d_head, 1, h->G_("fc_head.2.weight"), c.d_head, c.n_out, c.d_head));
    // the short tiles -- one reduction row per gene -- lead the upper bucket: the riders of k_trunk_bwd take the window BEHIND them (a rider is a
    // single wave: a long tile each keeps the rider waves equally busy), the reduction launch what lies on either side of that window
    h->n_wg_short = (int)wgShort.size();
    wgHi.insert(wgHi.begin(), wgShort.begin(), wgShort.end());
    push_cs(csHi, h->dh1, c.d_head, c.d_head, 1, 1, h->G_("fc_head.0.bias"));
    push_cs(csHi, h->dlogits, c.n_out, c.n_out, 1, 1, h->G_("fc_head.2.bias"));

    if (h->lp_jobs) (void)hipFree(h->lp_jobs);
    h->n_lp = (int)lpj.size();
    HIP_TRY(hipMalloc(&h->lp_jobs, lpj.size() * sizeof(LpJob)));
    HIP_TRY(hipMemcpy(h->lp_jobs, lpj.data(), lpj.size() * sizeof(LpJob), hipMemcpyHostToDevice));
    // Embedding + Pairwise tiles whose tensor has a tiled copy: where the tensor starts in the flat buffers and which of its rows the tile's row 0 is
    // (AdamFuse::tiled: the optimiser epilogue writes the stepped elements into the tiled copy as well)
    for (WgTile& t : wg) {
        const long long e0 = t.C - h->grads;
        for (const PDesc& p : h->table) {
            if (e0 < p.offset || e0 >= p.offset + p.numel) continue;
            const bool tiled_copy = p.ndim == 2 && p.shape[0] % 16 == 0 && p.shape[1] % 16 == 0 && p.trainable;
            if (tiled_copy && t.ldc == p.shape[1] && (e0 - p.offset) % p.shape[1] == 0) {
                t.toff = p.offset;
                t.trow0 = (int)((e0 - p.offset) / p.shape[1]);
            }
            break;
        }
    }
    {   // ... and every tensor the forward pass re-tiles up front must be covered completely, or the mode is not offered (cf_keep_tiled)
        h->keep_tiled_ok = !h->embed_dense;
        for (const PDesc& p : h->table) {
            const bool is_late = p.name.rfind("regulation.", 0) == 0 || p.name.rfind("fc_head.", 0) == 0;
            if (is_late || !(p.ndim == 2 && p.shape[0] % 16 == 0 && p.shape[1] % 16 == 0 && p.trainable)) continue;
            long long covered = 0;
            for (const WgTile& t : wg)
                if (t.toff == p.offset) covered += (long long)std::min(64, t.Nn - t.n0) * std::min(kWgTk, t.Kk - t.k0);
            if (covered != p.numel) h->keep_tiled_ok = false;
        }
    }
    if (h->keep_tiled_ok) {      // the float4 map of the separate AdamW launch (data parallel): flat -> tiled, for the same tensors
        std::vector<int> map((size_t)h->bucket_split / 4, -1);
        for (const PDesc& p : h->table) {
            const bool is_late = p.name.rfind("regulation.", 0) == 0 || p.name.rfind("fc_head.", 0) == 0;
            if (is_late || !(p.ndim == 2 && p.shape[0] % 16 == 0 && p.shape[1] % 16 == 0 && p.trainable)) continue;
            const int N = p.shape[0], K = p.shape[1];
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; k += 4) {
                    const long long flat = p.offset + (long long)n * K + k;
                    const long long til = p.offset + ((long long)(n / 16) * (K / 16) + k / 16) * 256 + (((k % 16) / 4) * 16 + n % 16) * 4;
                    map[(size_t)(flat / 4)] = (int)(til / 4);
                }
        }
        if (h->tiled_map) (void)hipFree(h->tiled_map);
        HIP_TRY(hipMalloc(&h->tiled_map, map.size() * sizeof(int)));
        HIP_TRY(hipMemcpy(h->tiled_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    h->n_wg_hi = (int)wgHi.size();
    h->n_cs_hi = (int)csHi.size();
    h->n_wg_r = (int)(wgHi.size() + wgLo.size());
    h->n_cs_r = (int)(csHi.size() + csLo.size());
    wg.insert(wg.begin(), wgLo.begin(), wgLo.end());      // table layout: [upper Regulation layers + head | lower Regulation layers | Embedding + Pairwise]
    wg.insert(wg.begin(), wgHi.begin(), wgHi.end());
    cs.insert(cs.begin(), csLo.begin(), csLo.end());
    cs.insert(cs.begin(), csHi.begin(), csHi.end());
    h->wg_flops_per_gene = 0.0;
    for (const WgTile& t : wg) {
        if (t.n0 || t.k0) continue;      // count each job once
        for (int sgi = 0; sgi < t.nseg; ++sgi) h->wg_flops_per_gene += 2.0 * t.seg[sgi].rows_per_gene * (double)t.Nn * t.Kk;
    }
    if (h->wg_tiles) (void)hipFree(h->wg_tiles);
    if (h->cs_tiles) (void)hipFree(h->cs_tiles);
    h->n_wg = (int)wg.size();
    h->n_cs = (int)cs.size();
    HIP_TRY(hipMalloc(&h->wg_tiles, wg.size() * sizeof(WgTile)));
    HIP_TRY(hipMalloc(&h->cs_tiles, cs.size() * sizeof(CsTile)));
    HIP_TRY(hipMemcpy(h->wg_tiles, wg.data(), wg.size() * sizeof(WgTile), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->cs_tiles, cs.data(), cs.size() * sizeof(CsTile), hipMemcpyHostToDevice));
    return 0;
}

// ------------------------------------------------------------------------------------
// C ABI: host-only part
// ------------------------------------------------------------------------------------
extern "C" int cf_abi_version(void) { return CF_ABI_VERSION; }
extern "C" const char* cf_last_error(void) { return g_err.c_str(); }

extern "C" int cf_param_layout(const cf_config* cfg, cf_layout* layout, cf_param_desc* table, int cap) {
    if (!cfg || !layout) return fail("cf_param_layout: null argument");
    std::vector<PDesc> v;
    if (build_layout(*cfg, v, *layout)) return -1;
    if (table) {
        if (cap < (int)v.size()) return fail("cf_param_layout: table capacity %d < %d", cap, (int)v.size());
        for (size_t i = 0; i < v.size(); ++i) {
            cf_param_desc& d = table[i];
            memset(&d, 0, sizeof d);
            snprintf(d.name, sizeof d.name, "%s", v[i].name.c_str());
            d.ndim = v[i].ndim;
            d.shape[0] = v[i].shape[0];
            d.shape[1] = v[i].shape[1];
            d.offset = v[i].offset;
            d.numel = v[i].numel;
            d.trainable = v[i].trainable ? 1 : 0;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------
static int embed_dense_alloc(cf_handle* h);      // buffers of the all-rows Embedding path (end of this file)
extern "C" void cf_destroy(cf_handle* h);

extern "C" int cf_create(const cf_config* cfg, const float* const* pe_host, cf_handle** out) {
    if (!cfg || !pe_host || !out) return fail("cf_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("cf_create: no HIP device visible -- libchromoformer_hip has no CPU fallback");
    cf_handle* h = new cf_handle();
    h->cfg = *cfg;
    {   // compute units of the current device: sizes the rows of rider workgroups that share a launch with one-per-CU workgroups
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0)
            h->n_cu = ncu;
    }
    if (build_layout(h->cfg, h->table, h->lay)) {
        delete h;
        return -1;
    }
    for (size_t i = 0; i < h->table.size(); ++i) h->index[h->table[i].name] = (int)i;
    {   // bucket boundary: trainable tensors are laid out in state_dict order (embed | pairwise | regulation | fc_head)
        long long split = -1, pe_end = 0;
        for (const PDesc& p : h->table) {
            if (!p.trainable) continue;
            const bool late = p.name.rfind("regulation.", 0) == 0 || p.name.rfind("fc_head.", 0) == 0;
            if (late && split < 0) split = p.offset;
            if (!late) pe_end = std::max(pe_end, p.offset + (p.numel + 3) / 4 * 4);
        }
        if (split < 0 || pe_end > split) {
            delete h;
            return fail("cf_create: parameter layout does not split into [embed, pairwise | regulation, head] ranges");
        }
        h->bucket_split = split;
        h->bucket_split_hi = h->lay.n_active;
        for (const PDesc& p : h->table) {
            if (!p.trainable || p.name.rfind("regulation.", 0) != 0) continue;
            const size_t at = p.name.find(".transformer.layers.");
            if (at != std::string::npos && atoi(p.name.c_str() + at + 20) >= h->cfg.reg_layers / 2) h->bucket_split_hi = std::min(h->bucket_split_hi, p.offset);
        }
    }
    {
        // units of the Embedding + Pairwise weights first (needed by the first kernels of a forward pass), Regulation + head behind
        // them (needed ~250 us later: they can ride in a later, under-filled launch, see PostArgs::rt_units)
        std::vector<RetileUnit> units;
        for (int late = 0; late < 2; ++late) {
            for (const PDesc& p : h->table) {
                const bool is_late = p.name.rfind("regulation.", 0) == 0 || p.name.rfind("fc_head.", 0) == 0;
                if (is_late != (late == 1)) continue;
                if (p.ndim == 2 && p.shape[0] % 16 == 0 && p.shape[1] % 16 == 0 && p.trainable)
                    for (int n0 = 0; n0 < p.shape[0]; n0 += 16) units.push_back(RetileUnit{p.offset + (long long)n0 * p.shape[1], p.offset, p.shape[1], p.shape[0], n0,
                                                    p.name.rfind("regulation.", 0) == 0 ? 1 : 0});
            }
            if (late == 0) h->n_retile_early = (int)units.size();
        }
        h->n_retile = (int)units.size();
        if (hipMalloc(&h->hyper, sizeof(AdamHyper)) != hipSuccess || hipMalloc(&h->tiled, h->lay.n_total * sizeof(float)) != hipSuccess ||
            hipMalloc(&h->tiledT, h->lay.n_total * sizeof(float)) != hipSuccess ||
            hipMalloc(&h->retile_units, units.size() * sizeof(RetileUnit)) != hipSuccess ||
            hipMemcpy(h->retile_units, units.data(), units.size() * sizeof(RetileUnit), hipMemcpyHostToDevice) != hipSuccess) {
            delete h;
            return fail("cf_create: allocation of the tiled weight copy failed");
        }
    }
    h->embed_dense = h->cfg.embed_layers > 1;
    h->planning = true;
    plan_workspace(h);
    hipError_t e = hipMalloc(&h->arena, h->arena_floats * sizeof(float));
    if (e != hipSuccess) {
        delete h;
        return fail("cf_create: hipMalloc(%zu bytes) failed: %s", h->arena_floats * sizeof(float), hipGetErrorString(e));
    }
    e = hipMemset(h->arena, 0, h->arena_floats * sizeof(float));
    h->planning = false;
    plan_workspace(h);
    for (const auto& en : h->ws) h->ws_names += en.name + "\n";
    for (int r = 0; r < h->cfg.n_res; ++r) {
        const int L = h->cfg.n_bins[r];
        const int kD = h->cfg.d_emb;      // (row width of the positional table: shadows cf::kD here)
        h->edx0[r] = h->E[r].dx;
        std::vector<float> t((size_t)L * kD);
        for (int j = 0; j < L; ++j)
            for (int d = 0; d < kD; ++d) t[(size_t)d * L + j] = pe_host[r][(size_t)j * kD + d];
        std::vector<float> t2((size_t)kD * attc2_lt(L), 0.f);
        for (int j = 0; j < L; ++j)
            for (int d = 0; d < kD; ++d) t2[(size_t)d * attc2_lt(L) + j] = pe_host[r][(size_t)j * kD + d];
        if (hipMemcpy(h->pe2[r], pe_host[r], (size_t)L * kD * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||       // rows >= L stay zero
            hipMemcpy(h->pet2[r], t2.data(), t2.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->pe[r], pe_host[r], (size_t)L * kD * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->pet[r], t.data(), (size_t)L * kD * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(h->arena);
            delete h;
            return fail("cf_create: positional table upload failed");
        }
    }
    const cf_config& c = h->cfg;
    {   // the fused Regulation kernels need up to ~150 KB of dynamic LDS
        const int T = c.i_max + 1;
        // 512-thread kernels (cf_reg8.h) for the default head count / width and T <= 16 tokens; CF_REG_FUSED=0 runs the stack layer by layer
        // on the stand-alone kernels instead (k_attr + the row-tile chains: what every other shape runs) -- the cross-check implementation
        h->reg_row0 = getenv_int("CF_REG_ROW0", 1) != 0;
        const size_t need = std::max(reg8_fwd_smem(c.reg_dff), reg8_bwd_smem(c.reg_dff));
        h->reg_fused = T <= kTile && need <= 160 * 1024 && c.reg_heads == kRH && c.reg_dmodel == kRDm && c.d_emb == kD && getenv_int("CF_REG_FUSED", 1) != 0;
        if (h->reg_fused) {
            const size_t sf = reg8_fwd_smem(c.reg_dff), sb = reg8_bwd_smem(c.reg_dff);
            hipError_t e1 = hipFuncSetAttribute(reg_kernel(false, c.reg_dff), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf);
            if (e1 == hipSuccess) e1 = hipFuncSetAttribute(reg_kernel(false, c.reg_dff, false), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf);
            hipError_t e2 = hipFuncSetAttribute(reg_kernel(true, c.reg_dff), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb);
            if (e1 != hipSuccess || e2 != hipSuccess) h->reg_fused = false;
        }
        h->reg8 = h->reg_fused;
    }
    {   // gene-batched attention kernel when its LDS image (8 regions of features + 16 score rows) fits
        size_t need = 0;
        for (int r = 0; r < c.n_res; ++r) need = std::max(need, attc2_smem(c.n_bins[r], c.n_feats, kAGMax));
        h->attc2 = need <= 160 * 1024 && c.d_emb == kD;      // (and the default row width: cf_attc1.h / cf_attc2.h are written for 128)
        if (h->attc2) {
            for (int ag = 1; ag <= kAGMax; ag *= 2) {
                size_t nd = 0;
                for (int r = 0; r < c.n_res; ++r) nd = std::max(nd, attc2_smem(c.n_bins[r], c.n_feats, ag));
                hipError_t e1 = hipFuncSetAttribute(attc2_kernel<false>(ag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nd);
                hipError_t e2 = hipFuncSetAttribute(attc2_kernel<true>(ag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nd);
                if (e1 != hipSuccess || e2 != hipSuccess) h->attc2 = false;
            }
        }
    }
    if (h->embed_dense && embed_dense_alloc(h)) {      // the training path needs the all-rows buffers: allocate them now, not on the hot path
        cf_destroy(h);
        return -1;
    }
    if (const char* e = getenv("CF_XCD_MAP")) h->xcd_map = atoi(e) != 0;
    if (const char* e = getenv("CF_ATTC_CAP")) h->attc_cap = atoi(e);
    if (const char* e = getenv("CF_ATTC1")) h->attc1 = atoi(e) != 0;
    if (const char* e = getenv("CF_DEFER_RETILE")) h->defer_retile = atoi(e) != 0;
    if (const char* e = getenv("CF_HEAD_RIDE")) h->head_ride = atoi(e) != 0;
    if (const char* e = getenv("CF_XCD_REDUCE")) h->xcd_reduce = h->xcd_reduce_opt = atoi(e) != 0;
    h->n_fwd = h->n_bwd = h->n_opt = 0;      // counted at the launch sites by the first calls (cf_launch_counts)
    *out = h;
    return 0;
}

extern "C" void cf_destroy(cf_handle* h) {
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    if (h->wg_tiles) (void)hipFree(h->wg_tiles);
    if (h->tiled_map) (void)hipFree(h->tiled_map);
    if (h->cs_tiles) (void)hipFree(h->cs_tiles);
    if (h->lp_jobs) (void)hipFree(h->lp_jobs);
    if (h->reg_tab) (void)hipFree(h->reg_tab);
    if (h->trunk_tab) (void)hipFree(h->trunk_tab);
    if (h->tiled) (void)hipFree(h->tiled);
    if (h->tiledT) (void)hipFree(h->tiledT);
    for (void* q : h->ed.owned) (void)hipFree(q);
    if (h->retile_units) (void)hipFree(h->retile_units);
    for (cf_handle::Replay& rp : h->replays) {
        if (rp.first) (void)hipGraphExecDestroy(rp.first);
        if (rp.second) (void)hipGraphExecDestroy(rp.second);
    }
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->sync_ev) (void)hipEventDestroy(e);
    if (h->ride_ev) (void)hipEventDestroy(h->ride_ev);
    if (h->hyper) (void)hipFree(h->hyper);
    delete h;
}

static int build_trunk_table(cf_handle* h);      // (behind the centre-row parameter helpers below)
extern "C" int cf_bind(cf_handle* h, float* params, float* grads, float* exp_avg, float* exp_avg_sq) {
    if (!h || !params) return fail("cf_bind: null handle / params");
    h->tiled_pe_fresh = false;
    h->params = params;
    h->grads = grads;
    h->m = exp_avg;
    h->v = exp_avg_sq;
    if (build_reg_table(h) || build_trunk_table(h)) return -1;
    if (grads) return build_tables(h);
    return 0;
}

// the all-rows Embedding path (defined behind the dense-layer operators at the end of this file)
static int embed_dense_forward(cf_handle* h, const cf_batch* bt, bool train, hipStream_t st);
static int embed_dense_backward(cf_handle* h, const cf_batch* bt, hipStream_t st);

// ------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------
static inline int tiles_of(int n) { return (n + kTile - 1) / kTile; }
static size_t attc_smem(int L, int F, bool bwd, int nh = 2, int D = kD) {
    const int parts = 256 / D > 2 ? 256 / D - 1 : 1;
    return (size_t)((1 + parts) * nh * D + 20 * nh + nh * L + (bwd ? nh * L : 0) + L * F) * sizeof(float);
}
static size_t attr_smem(int T, int H, int DM, bool bwd) {
    return (size_t)(T * 4 * DM + H * T * T + (bwd ? H * T * T + T * DM + H * T : 0)) * sizeof(float);
}

#ifndef CF_POST_WAVES
#define CF_POST_WAVES 8
#endif
constexpr int kPostWaves = CF_POST_WAVES;      // waves per workgroup of the row-tile chains (k_post_*, k_qchain_*): 4 or 8
template <bool VPROJ, int DM, int D = 128>
static void launch_post_fwd(int dff, dim3 grid, hipStream_t st, const PostArgs& a) {
    if constexpr (D != 128) {      // (rows of another width: the chain kernels with the width as a template parameter; four waves at 64)
        constexpr int NWV = D == 64 ? 4 : 8;
        if (dff == 128) hipLaunchKernelGGL((k_post_fwd<VPROJ, DM, 128, NWV, false, 2, D>), grid, dim3(NWV * 64), 0, st, a);
        else hipLaunchKernelGGL((k_post_fwd<VPROJ, DM, 256, NWV, false, 2, D>), grid, dim3(NWV * 64), 0, st, a);
        return;
    }
    if constexpr (VPROJ && DM == 128) {      // the hosting instantiation (eight waves; the Embedding layer's launch asks for it, nobody else)
        if (a.rt_units) {
            if (dff == 128) hipLaunchKernelGGL((k_post_fwd<VPROJ, DM, 128, 8, true>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_post_fwd<VPROJ, DM, 256, 8, true>), grid, dim3(512), 0, st, a);
            return;
        }
    }
    // (kPostWaves is a compile-time switch: only the selected form is instantiated)
    if (dff == 128) hipLaunchKernelGGL((k_post_fwd<VPROJ, DM, 128, kPostWaves>), grid, dim3(kPostWaves * 64), 0, st, a);
    else hipLaunchKernelGGL((k_post_fwd<VPROJ, DM, 256, kPostWaves>), grid, dim3(kPostWaves * 64), 0, st, a);
}
template <bool VPROJ, int DM, int D = 128>
static void launch_post_bwd(int dff, dim3 grid, hipStream_t st, const PostBwdArgs& a) {
    if constexpr (D != 128) {
        constexpr int NWV = D == 64 ? 4 : 8;
        if (dff == 128) hipLaunchKernelGGL((k_post_bwd<VPROJ, DM, 128, NWV, 2, D>), grid, dim3(NWV * 64), 0, st, a);
        else hipLaunchKernelGGL((k_post_bwd<VPROJ, DM, 256, NWV, 2, D>), grid, dim3(NWV * 64), 0, st, a);
        return;
    }
    if (dff == 128) hipLaunchKernelGGL((k_post_bwd<VPROJ, DM, 128, kPostWaves>), grid, dim3(kPostWaves * 64), 0, st, a);
    else hipLaunchKernelGGL((k_post_bwd<VPROJ, DM, 256, kPostWaves>), grid, dim3(kPostWaves * 64), 0, st, a);
}

// The stand-alone stages of one centre-row layer for a head count other than 2 (eight-wave chain kernels, the one-sequence-per-workgroup
// attention): the same argument structures as the default launches, [N, NH, .] arrays.
// D: the row width (d_emb; the Embedding / Pairwise attention width is the same, net.py:305 / 361-370).  Every shape but the default
// (two heads, 128) takes this route.
template <int NH, int D = 128>
static int centre_fwd_heads(hipStream_t st, int N, int nres, int dff, bool q_done, const QChainArgs& q, const AttcArgs& at, size_t smem,
                            const PostArgs& po) {
    constexpr int NWV = D == 64 ? 4 : 8;      // (a wave owns at least one 16-column tile of a D-wide product)
    if (!q_done) {
        hipLaunchKernelGGL((k_qchain_fwd<NWV, NH, D>), dim3(tiles_of(N), nres), dim3(NWV * 64), 0, st, q);
        LAUNCH_CHECK("k_qchain_fwd");
    }
    hipLaunchKernelGGL((k_attc<false, NH, D>), dim3(N, nres), dim3(256), smem, st, at);
    LAUNCH_CHECK("k_attc<fwd>");
    if (dff == 128) hipLaunchKernelGGL((k_post_fwd<true, D, 128, NWV, false, NH, D>), dim3(tiles_of(N), nres), dim3(NWV * 64), 0, st, po);
    else hipLaunchKernelGGL((k_post_fwd<true, D, 256, NWV, false, NH, D>), dim3(tiles_of(N), nres), dim3(NWV * 64), 0, st, po);
    LAUNCH_CHECK("k_post_fwd<centre>");
    return 0;
}
template <int NH, int D = 128>
static int centre_bwd_heads(hipStream_t st, int N, int nres, int dff, const PostBwdArgs& pb, const AttcArgs& at, size_t smem, const QBwdArgs& qb) {
    constexpr int NWV = D == 64 ? 4 : 8;
    if (dff == 128) hipLaunchKernelGGL((k_post_bwd<true, D, 128, NWV, NH, D>), dim3(tiles_of(N), nres), dim3(NWV * 64), 0, st, pb);
    else hipLaunchKernelGGL((k_post_bwd<true, D, 256, NWV, NH, D>), dim3(tiles_of(N), nres), dim3(NWV * 64), 0, st, pb);
    LAUNCH_CHECK("k_post_bwd<centre>");
    hipLaunchKernelGGL((k_attc<true, NH, D>), dim3(N, nres), dim3(256), smem, st, at);
    LAUNCH_CHECK("k_attc<bwd>");
    hipLaunchKernelGGL((k_qchain_bwd<NWV, NH, D>), dim3(tiles_of(N), nres), dim3(NWV * 64), 0, st, qb);
    LAUNCH_CHECK("k_qchain_bwd");
    return 0;
}

struct CentreParams {   // weights of one centre-row layer (_t: tiled copies for the forward products)
    const float *wq, *wk, *wv, *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2, *wlp;
    const float *wq_t, *wk_t, *wv_t, *wo_t, *w1_t, *w2_t;
};
static CentreParams centre_params(const cf_handle* h, const std::string& att_pre, const std::string& ff_pre, const float* wq,
                                  const float* wk, const float* wv, const float* wlp) {
    CentreParams p;
    p.wq = wq;
    p.wk = wk;
    p.wv = wv;
    p.wlp = wlp;
    const long long td = h->tiled - h->params;     // same offsets in both buffers
    p.wq_t = wq + td;
    p.wk_t = wk + td;
    p.wv_t = wv + td;
    p.wo = h->P_(att_pre + "ff.weight");
    p.wo_t = p.wo + td;
    p.bo = h->P_(att_pre + "ff.bias");
    p.g1 = h->P_(att_pre + "ln.weight");
    p.be1 = h->P_(att_pre + "ln.bias");
    p.w1 = h->P_(ff_pre + "l1.weight");
    p.w1_t = p.w1 + td;
    p.b1 = h->P_(ff_pre + "l1.bias");
    p.w2 = h->P_(ff_pre + "l2.weight");
    p.w2_t = p.w2 + td;
    p.b2 = h->P_(ff_pre + "l2.bias");
    p.g2 = h->P_(ff_pre + "ln.weight");
    p.be2 = h->P_(ff_pre + "ln.bias");
    return p;
}
static CentreParams embed_params(const cf_handle* h, int r) {
    const std::string pre = fmt("embed.%d.", h->cfg.binsizes[r]), lp = pre + "transformer.layers.0.";
    const float* att = h->P_(lp + "self_att.att.weight");
    const size_t kD = h->cfg.d_emb;      // (row width: shadows cf::kD)
    return centre_params(h, lp + "self_att.", lp + "ff.", att, att + (size_t)kD * kD, att + (size_t)2 * kD * kD,
                         h->P_(pre + "lin_proj.weight"));
}
static CentreParams pair_params(const cf_handle* h, int r, int l) {
    const std::string pre = fmt("pairwise_interaction.%d.", h->cfg.binsizes[r]), lp = pre + fmt("transformer.layers.%d.", l);
    const float* c_att = h->P_(lp + "self_att.c_att.weight");
    const size_t kD = h->cfg.d_emb;      // (row width: shadows cf::kD)
    return centre_params(h, lp + "self_att.", lp + "ff.", h->P_(lp + "self_att.p_att.weight"), c_att, c_att + (size_t)kD * kD,
                         h->P_(pre + "lin_proj_pcre.weight"));
}

// device table of the fused centre-row trunk (cf_trunk.h)
static const void* trunk_kernel(bool bwd, int dff_e, int dff_p, int pair_layers) {
    if (dff_e == 128 && dff_p == 256 && pair_layers == 2) return bwd ? (const void*)k_trunk_bwd<128, 256, 2> : (const void*)k_trunk_fwd<128, 256, 2>;
    return nullptr;      // (other shapes run the stand-alone kernels)
}
static void fill_centre_dev(CentreLayerDev& d, const CentreParams& p, const CentreBuf& b) {
    d.wq_t = p.wq_t, d.wk = p.wk, d.wv_t = p.wv_t, d.wo_t = p.wo_t, d.bo = p.bo, d.g1 = p.g1, d.be1 = p.be1;
    d.w1_t = p.w1_t, d.b1 = p.b1, d.w2_t = p.w2_t, d.b2 = p.b2, d.g2 = p.g2, d.be2 = p.be2;
    d.wq = p.wq, d.wk_t = p.wk_t, d.wv = p.wv, d.wo = p.wo, d.w1 = p.w1, d.w2 = p.w2;
    d.q = b.q, d.qt = b.qt, d.p = b.p, d.w = b.w, d.xbar = b.xbar, d.a = b.a, d.xh1 = b.xh1, d.rs1 = b.rs1, d.y1 = b.y1;
    d.hdn = b.hdn, d.xh2 = b.xh2, d.rs2 = b.rs2, d.out = b.out, d.xin = b.xin;
    d.dt2 = b.dt2, d.dpre1 = b.dpre1, d.dt1 = b.dt1, d.da = b.da, d.dxbar = b.dxbar, d.dqt = b.dqt, d.du = b.du, d.dq = b.dq;
    d.dx = b.dx, d.partial = b.partial;
}
static int build_trunk_table(cf_handle* h) {
    const cf_config& c = h->cfg;
    h->trunk = false;
    if (h->embed_dense || !h->attc2 || c.i_max > kAGMax || c.pair_layers > kMaxPairLayers || kPostWaves != 8) return 0;
    if (c.embed_heads != 2 || c.pair_heads != 2 || c.d_emb != kD) return 0;      // (the fused kernels are written for two heads and 128-wide rows)
    if (!trunk_kernel(false, c.embed_dff, c.pair_dff, c.pair_layers)) return 0;
    if (const char* e = getenv("CF_TRUNK"))      // CF_TRUNK=0: the stand-alone kernels (A/B runs, cross-checks in the tests)
        if (atoi(e) == 0) return 0;
    size_t need = 0;
    for (int r = 0; r < c.n_res; ++r) need = std::max(need, trunk_smem(c.n_bins[r], c.n_feats, std::max(c.embed_dff, c.pair_dff)));
    if (need > 160 * 1024) return 0;
    const size_t need_bwd = std::max(need, (size_t)(kAT / 64) * kWgWaveLds * sizeof(float));      // (riders of the backward launch: cf_rider_arm)
    if (need_bwd > 160 * 1024) return 0;
    if (hipFuncSetAttribute(trunk_kernel(false, c.embed_dff, c.pair_dff, c.pair_layers), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need) != hipSuccess ||
        hipFuncSetAttribute(trunk_kernel(true, c.embed_dff, c.pair_dff, c.pair_layers), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need_bwd) != hipSuccess)
        return 0;
    std::vector<TrunkResDev> tab(c.n_res);
    for (int r = 0; r < c.n_res; ++r) {
        TrunkResDev& t = tab[r];
        memset(&t, 0, sizeof t);
        fill_centre_dev(t.E, embed_params(h, r), h->E[r]);
        for (int l = 0; l < c.pair_layers; ++l) fill_centre_dev(t.P[l], pair_params(h, r, l), h->P[r][l]);
        t.pe = h->pe[r], t.pe2 = h->pe2[r], t.pet2 = h->pet2[r];
        t.wlp_e = embed_params(h, r).wlp;
        t.wlp_p = pair_params(h, r, 0).wlp;
        t.lin_p = h->P_(fmt("pairwise_interaction.%d.lin_proj_p.weight", c.binsizes[r]));
        t.lin_p_t = h->T_(fmt("pairwise_interaction.%d.lin_proj_p.weight", c.binsizes[r]));
        t.ex0 = h->ex0[r], t.featc = h->featc[r], t.xp0 = h->xp0[r], t.dxp0 = h->dxp0[r], t.edout = h->edout[r];
        t.rx0 = h->Rx[r][0], t.drx0 = h->dRx[r][0];
        t.lp_part_e = h->lp_part_e[r], t.lp_part_p = h->lp_part_p[r];
        t.L = c.n_bins[r], t.Lpad = attc2_lpad(c.n_bins[r]), t.LT = attc2_lt(c.n_bins[r]);
    }
    if (h->trunk_tab) (void)hipFree(h->trunk_tab);
    HIP_TRY(hipMalloc(&h->trunk_tab, tab.size() * sizeof(TrunkResDev)));
    HIP_TRY(hipMemcpy(h->trunk_tab, tab.data(), tab.size() * sizeof(TrunkResDev), hipMemcpyHostToDevice));
    h->trunk_smem_bytes = need;
    h->trunk = true;
    return 0;
}
static void trunk_args(const cf_handle* h, const cf_batch* bt, TrunkArgs& a, int save) {
    const cf_config& c = h->cfg;
    memset(&a, 0, sizeof a);
    a.tab = h->trunk_tab;
    for (int r = 0; r < c.n_res; ++r) {
        a.pfeats[r] = bt->promoter_feats[r], a.pmask[r] = bt->promoter_mask_row[r], a.pmstride[r] = bt->promoter_mask_stride[r];
        a.cfeats[r] = bt->pcre_feats[r], a.cmask[r] = bt->pcre_mask_row[r], a.cmstride[r] = bt->pcre_mask_stride[r];
    }
    a.dhin = h->dhin;
    a.B = bt->B, a.S = c.i_max, a.T = c.i_max + 1, a.F = c.n_feats, a.n_res = c.n_res, a.pair_layers = c.pair_layers, a.save = save;
    a.scale = sqrtf(64.f);
    a.rscale = 1.0f / a.scale;
    a.tdbg = getenv("CF_STAMP_TRUNK") ? reinterpret_cast<unsigned long long*>(h->tdbg) : nullptr;      // tools/trunk_stamps.py
}

static int check_batch(const cf_handle* h, const cf_batch* b) {
    if (!h || !b) return fail("null handle / batch");
    if (!h->params) return fail("cf_bind has not been called");
    if (b->B < 1 || b->B > h->cfg.max_batch) return fail("batch size %d outside 1..max_batch=%d", b->B, h->cfg.max_batch);
    for (int r = 0; r < h->cfg.n_res; ++r)
        if (!b->promoter_feats[r] || !b->pcre_feats[r] || !b->promoter_mask_row[r] || !b->pcre_mask_row[r] || !b->interaction_mask[r])
            return fail("batch pointer for resolution %d is null", r);
    if (!b->interaction_freq) return fail("interaction_freq is null");
    return 0;
}

// Launch of a fused Regulation kernel.  Under capture, if it is the kernel selected with cf_timing_select, the
// capture is split around it: the launch is remembered instead of recorded and cf_graph_launch issues it eagerly,
// between two HIP events, between the two graph pieces.
static int launch_reg(cf_handle* h, const char* name, const void* fn, dim3 grid, size_t smem, RegArgs& ra, hipStream_t st) {
    const dim3 block(512);
    if (h->capturing && h->timed == name && !h->cap.has_hole) {
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamEndCapture(st, &g));
        hipError_t e = hipGraphInstantiate(&h->cap.first, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        h->cap.has_hole = true;
        h->cap.hole.func = fn;
        h->cap.hole.grid = grid;
        h->cap.hole.block = block;
        h->cap.hole.smem = smem;
        h->cap.hole.args = ra;
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        ++g_launches;      // (issued by cf_graph_launch between the two graph pieces)
        return 0;
    }
    void* kargs[] = {&ra};
    h->time_mark(name, st);
    HIP_TRY(hipLaunchKernel(fn, grid, block, kargs, smem, st));
    h->time_mark(name, st);
    LAUNCH_CHECK(name);
    return 0;
}

static void head_gen_args(const cf_handle* h, int B, float* logits_user, HeadGenArgs& a) {      // d_head != 128: the vector-ALU head (cf_head.h)
    const cf_config& c = h->cfg;
    memset(&a, 0, sizeof a);
    for (int r = 0; r < c.n_res; ++r) {
        a.xl[r] = h->Rx[r][c.reg_layers];
        a.x0[r] = h->Rx[r][0];
        a.dxl[r] = h->dRx[r][c.reg_layers];
    }
    a.w1 = h->P_("fc_head.0.weight");
    a.b1 = h->P_("fc_head.0.bias");
    a.w2 = h->P_("fc_head.2.weight");
    a.b2 = h->P_("fc_head.2.bias");
    a.hin = h->hin, a.h1 = h->h1, a.logits = h->logits, a.logits_user = logits_user;
    a.dlogits = h->dlogits, a.dh1 = h->dh1, a.dhin = h->dhin;
    a.loss = h->loss, a.loss_part = h->loss_part;
    a.B = B, a.T = c.i_max + 1, a.n_res = c.n_res, a.n_out = c.n_out, a.DH = c.d_head;
    a.D = c.d_emb;
}
static void head_fwd_args(const cf_handle* h, int B, float* logits_user, HeadFwdArgs& a) {
    const cf_config& c = h->cfg;
    for (int r = 0; r < c.n_res; ++r) {
        a.xl[r] = h->Rx[r][c.reg_layers];
        a.x0[r] = h->Rx[r][0];
    }
    a.w1_t = h->T_("fc_head.0.weight");
    a.b1 = h->P_("fc_head.0.bias");
    a.w2 = h->P_("fc_head.2.weight");
    a.b2 = h->P_("fc_head.2.bias");
    a.hin = h->hin;
    a.h1 = h->h1;
    a.logits = h->logits;
    a.logits_user = logits_user;
    a.B = B;
    a.T = c.i_max + 1;
    a.n_res = c.n_res;
    a.n_out = c.n_out;
    a.tdbg = getenv("CF_STAMP_HEAD") ? reinterpret_cast<unsigned long long*>(h->tdbg) + 128 : nullptr;      // tools/head_stamps.py
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
static int gather_launch(const GatherArgs& ga, int n, hipStream_t st);      // (with the gather entry points below)
// the tiled copies of the Embedding + Pairwise weights, now (cf_keep_tiled: whenever somebody other than the fused optimiser has written them)
static int retile_early(cf_handle* h, hipStream_t st) {
    if (h->n_retile_early > 0) {
        hipLaunchKernelGGL(k_retile, dim3(h->n_retile_early), dim3(256), 0, st, (const float*)h->params, h->tiled, h->reg8 ? h->tiledT : (float*)nullptr,
                           (const RetileUnit*)h->retile_units);
        LAUNCH_CHECK("k_retile");
    }
    h->tiled_pe_fresh = true;
    return 0;
}
static int forward_impl(cf_handle* h, const cf_batch* bt, float* logits, int save, void* stream, const HeadRide* ride) {
    if (check_batch(h, bt)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const long long launches0 = g_launches;
    const cf_config& c = h->cfg;
    const int kD = c.d_emb;      // (row width: shadows cf::kD in this function; 128, or 256 through the stand-alone kernels)
    const int B = bt->B, S = c.i_max, T = S + 1, nres = c.n_res, F = c.n_feats;
    const int NE = B, NP = B * S, NR = B * T;
    CentreParams ep[kMaxRes], pp[kMaxRes];
    for (int r = 0; r < nres; ++r) ep[r] = embed_params(h, r);
    const bool defer = h->defer_retile && !h->embed_dense && kPostWaves == 8 && c.embed_heads == 2 && c.d_emb == 128;      // (kD is shadowed in this function) Regulation + head units ride in the Embedding layer's chain launch
    const bool trunk = h->trunk;                                                   // Embedding + Pairwise stage as ONE launch (cf_trunk.h)
    int* adv_cursor = nullptr;
    {   // refresh the tiled weight copies (the parameters may have been changed by anyone since the last call) and, in the same
        // launch, the Embedding centre-row input
        X0Args a;
        for (int r = 0; r < nres; ++r) {
            a.feats[r] = bt->promoter_feats[r];
            a.pe[r] = h->pe[r];
            a.wlp[r] = ep[r].wlp;
            a.x0[r] = h->ex0[r];
            a.featc[r] = h->featc[r];
            a.L[r] = c.n_bins[r];
        }
        a.F = F;
        // cf_keep_tiled: the Embedding + Pairwise units (the leading ones) are kept fresh by the optimiser epilogue; if something else has
        // written parameters since (cf_params_changed, cf_bind, a separate AdamW launch) they are re-tiled here, once, in a launch of their own
        if (h->keep_tiled && !h->tiled_pe_fresh && retile_early(h, st)) return -1;
        if (h->pend_gnext) {      // (a cf_gather_batch_next that found no reduction launch to ride in: a launch of its own, here)
            hipLaunchKernelGGL(k_gather_batch, dim3(h->pend_gn.B, h->pend_gn_n), dim3(256), 0, st, h->pend_gn);
            LAUNCH_CHECK("k_gather_batch");
            h->adv_next = h->pend_gn.cursor;
            h->pend_gnext = false;
        }
        // The pre-gathered feed's pending state belongs to ONE batch: only the forward pass over that batch takes the gather into its
        // launch / moves the cursor on.  A pass over another batch (Trainer.evaluate_store or model(...) between two steps of a fed epoch)
        // leaves it for the step it was queued for -- consumed here it advanced the cursor a second time under graph replay (the
        // captured trunk launch advances it by itself) and the epoch silently skipped a batch.
        const bool mine = !h->pend_key || h->pend_key == (const void*)bt->promoter_feats[0];
        const int u0 = h->keep_tiled ? h->n_retile_early : 0;
        const int n_now = (defer ? h->n_retile_early : h->n_retile) - u0;
        const RetileUnit* units = (const RetileUnit*)h->retile_units + u0;
        const bool with_gather = mine && h->pend_gather && trunk && h->pend_ga.B == B;
        if (mine && h->pend_gather && !with_gather && gather_launch(h->pend_ga, h->pend_ga_n, st)) return -1;      // (launches of their own, as cf_gather_batch)
        if (with_gather) {      // the step's batch gather in the same launch (cf_gather_batch_fwd); the trunk's forward launch advances the cursor
            hipLaunchKernelGGL(k_prologue_gather, dim3(n_now + B * h->pend_ga_n), dim3(256), 0, st, (const float*)h->params, h->tiled,
                               h->reg8 ? h->tiledT : (float*)nullptr, units, n_now, h->pend_ga);
            LAUNCH_CHECK("k_prologue_gather");
        } else if (n_now + (trunk ? 0 : B * nres) > 0) {      // (nothing to re-tile and nothing to gather: no launch)
            // (the fused trunk computes the Embedding input row itself: no x0 workgroups then)
            if (kD == 256)
                hipLaunchKernelGGL(k_fwd_prologue<256>, dim3(n_now + (trunk ? 0 : B * nres)), dim3(256), 0, st, (const float*)h->params, h->tiled,
                                   h->reg8 ? h->tiledT : (float*)nullptr, units, n_now, a, B);
            else if (kD == 64)
                hipLaunchKernelGGL(k_fwd_prologue<64>, dim3(n_now + (trunk ? 0 : B * nres)), dim3(256), 0, st, (const float*)h->params, h->tiled,
                                   h->reg8 ? h->tiledT : (float*)nullptr, units, n_now, a, B);
            else
                hipLaunchKernelGGL(k_fwd_prologue<128>, dim3(n_now + (trunk ? 0 : B * nres)), dim3(256), 0, st, (const float*)h->params, h->tiled,
                                   h->reg8 ? h->tiledT : (float*)nullptr, units, n_now, a, B);
            LAUNCH_CHECK("k_fwd_prologue");
        }
        if (mine) h->pend_gather = false;
        if (with_gather) adv_cursor = h->pend_ga.cursor;
        else if (mine && h->adv_next) {      // the batch is in place already (cf_gather_batch_only / cf_gather_batch_next): only the cursor moves on
            if (trunk) adv_cursor = h->adv_next;
            else {
                hipLaunchKernelGGL(k_gather_advance, dim3(1), dim3(1), 0, st, h->adv_next);
                LAUNCH_CHECK("k_gather_advance");
            }
        }
        if (mine) {
            h->adv_next = nullptr;
            h->pend_key = nullptr;
        }
    }
    if (trunk) {
        TrunkArgs ta;
        trunk_args(h, bt, ta, save);
        if (defer) {
            ta.rt_units = h->retile_units + h->n_retile_early;
            ta.rt_n = h->n_retile - h->n_retile_early;
            ta.rt_params = h->params;
            ta.rt_tiled = h->tiled;
            ta.rt_tiledT = h->reg8 ? h->tiledT : nullptr;
        }
        ta.adv_cursor = adv_cursor;
        void* kargs[] = {&ta};
        h->time_mark("k_trunk_fwd", st);
        HIP_TRY(hipLaunchKernel(trunk_kernel(false, c.embed_dff, c.pair_dff, c.pair_layers), dim3(B, nres + (defer && ta.rt_n > 0 ? 1 : 0)), dim3(kAT), kargs,
                                h->trunk_smem_bytes, st));
        h->time_mark("k_trunk_fwd", st);
        LAUNCH_CHECK("k_trunk_fwd");
    }
    // one centre-row layer: query chain -> attention -> post chain
    auto centre_layer = [&](CentreBuf* bufs[kMaxRes], const CentreParams* prm, const float* const* xin, RowMap xmap,
                            const float* const* feats, const uint8_t* const* mask, const long long* mstride, int N, int dff,
                            float* const* outp, RowMap omap, bool copy_x, int nh, const float* const* lin_w = nullptr,
                            float* const* lin_y = nullptr, bool q_done = false, const CentreParams* next_prm = nullptr,
                            CentreBuf* const* next_bufs = nullptr, bool host_retile = false) -> int {
        const float scale_c = sqrtf((float)(kD / nh));      // sqrt(d_head), modules.py:60-61
        // q_done: the previous layer's chain kernel has already run this layer's query chain; next_prm / next_bufs: run the next
        // layer's query chain at the end of this layer's chain kernel (only for identity row maps: the output tile IS its input)
        QChainArgs q;
        AttcArgs at;
        PostArgs po;
        for (int r = 0; r < kMaxRes; ++r) {
            po.lin_w[r] = lin_w && r < nres ? lin_w[r] : nullptr;
            po.lin_y[r] = lin_y && r < nres ? lin_y[r] : nullptr;
            if (next_prm && r < nres) {
                po.nq_wq[r] = next_prm[r].wq_t;
                po.nq_wk[r] = next_prm[r].wk;
                po.nq_q[r] = next_bufs[r]->q;
                po.nq_qt[r] = next_bufs[r]->qt;
            }
        }
        size_t smem = 0;
        for (int r = 0; r < nres; ++r) {
            CentreBuf& b = *bufs[r];
            q.x[r] = xin[r];
            q.wq[r] = prm[r].wq_t;      // NT product: tiled copy
            q.wk[r] = prm[r].wk;        // NN product: row-major
            q.q[r] = b.q;
            q.qt[r] = b.qt;
            q.xcopy[r] = copy_x ? b.xin : nullptr;
            at.feats[r] = feats[r];
            at.mask[r] = mask[r];
            at.mstride[r] = mstride[r];
            at.pe[r] = h->pe[r];
            at.pet[r] = h->pet[r];
            at.wlp[r] = prm[r].wlp;
            at.vin[r] = b.qt;
            at.p[r] = b.p;
            at.w[r] = b.w;
            at.vout[r] = b.xbar;
            at.L[r] = c.n_bins[r];
            smem = std::max(smem, attc_smem(c.n_bins[r], F, false, nh, kD));
            po.x[r] = xin[r];
            po.ain[r] = b.xbar;
            po.wv[r] = prm[r].wv_t;
            po.wo[r] = prm[r].wo_t;
            po.bo[r] = prm[r].bo;
            po.g1[r] = prm[r].g1;
            po.be1[r] = prm[r].be1;
            po.w1[r] = prm[r].w1_t;
            po.b1[r] = prm[r].b1;
            po.w2[r] = prm[r].w2_t;
            po.b2[r] = prm[r].b2;
            po.g2[r] = prm[r].g2;
            po.be2[r] = prm[r].be2;
            po.a_out[r] = b.a;
            po.xh1[r] = b.xh1;
            po.rs1[r] = b.rs1;
            po.y1[r] = b.y1;
            po.hdn[r] = b.hdn;
            po.xh2[r] = b.xh2;
            po.rs2[r] = b.rs2;
            po.out[r] = outp[r];
        }
        q.xmap = xmap;
        q.N = N;
        at.F = F;
        at.scale = scale_c;
        po.xmap = xmap;
        po.omap = omap;
        po.N = N;
        po.save = save;
        if (kD != 128) {      // (rows of 64 / 256: the stand-alone kernels with the width as a template parameter; one or two heads)
            if (smem > 64 * 1024) return fail("cf_forward: a region of %d bins does not fit the one-sequence attention at d_emb = %d", c.n_bins[nres - 1], kD);
            if (kD == 256) return nh == 1 ? centre_fwd_heads<1, 256>(st, N, nres, dff, q_done, q, at, smem, po) : centre_fwd_heads<2, 256>(st, N, nres, dff, q_done, q, at, smem, po);
            return nh == 1 ? centre_fwd_heads<1, 64>(st, N, nres, dff, q_done, q, at, smem, po) : centre_fwd_heads<2, 64>(st, N, nres, dff, q_done, q, at, smem, po);
        }
        if (nh == 1) return centre_fwd_heads<1>(st, N, nres, dff, q_done, q, at, smem, po);
        if (nh == 4) return centre_fwd_heads<4>(st, N, nres, dff, q_done, q, at, smem, po);
        if (!q_done) {
            hipLaunchKernelGGL((k_qchain_fwd<kPostWaves>), dim3(tiles_of(N), nres), dim3(kPostWaves * 64), 0, st, q);
            LAUNCH_CHECK("k_qchain_fwd");
        }
        if (h->attc2) {
            const int ag = attc2_regions_per_wg(N, h->attc_cap);
            Attc2Args a2;
            size_t sm2 = 0;
            for (int r = 0; r < nres; ++r) {
                a2.feats[r] = at.feats[r];
                a2.mask[r] = at.mask[r];
                a2.mstride[r] = at.mstride[r];
                a2.pe[r] = h->pe2[r];
                a2.pet[r] = h->pet2[r];
                a2.wlp[r] = at.wlp[r];
                a2.vin[r] = at.vin[r];
                a2.p[r] = at.p[r];
                a2.w[r] = at.w[r];
                a2.vout[r] = at.vout[r];
                a2.L[r] = at.L[r];
                a2.Lpad[r] = attc2_lpad(at.L[r]);
                a2.LT[r] = attc2_lt(at.L[r]);
                sm2 = std::max(sm2, attc2_smem(at.L[r], F, ag));
            }
            a2.N = N;
            a2.F = F;
            a2.scale = scale_c;
            a2.rscale = 1.0f / a2.scale;
            a2.tdbg = (getenv("CF_STAMP_ATTC") && (!getenv("CF_STAMP_ATTC_AG") || atoi(getenv("CF_STAMP_ATTC_AG")) == ag))
                          ? reinterpret_cast<unsigned long long*>(h->tdbg) + 256 : nullptr;
            a2.tall = (getenv("CF_STAMP_ATTC_ALL") && atoi(getenv("CF_STAMP_ATTC_ALL")) == 0 && (!getenv("CF_STAMP_ATTC_AG") || atoi(getenv("CF_STAMP_ATTC_AG")) == ag))
                          ? reinterpret_cast<unsigned long long*>(h->tdbg) + 256 : nullptr;
            void* kargs2[] = {&a2};
            if (ag == 1 && h->attc1) {      // one region per workgroup: the vector-ALU kernel (cf_attc1.h)
                size_t sm1 = 0;
                for (int r = 0; r < nres; ++r) sm1 = std::max(sm1, attc1_smem(at.L[r], F));
                HIP_TRY(hipLaunchKernel((const void*)k_attc1<false>, dim3(N, nres), dim3(kAT), kargs2, sm1, st));
            } else {
                HIP_TRY(hipLaunchKernel(attc2_kernel<false>(ag), dim3((N + ag - 1) / ag, nres), dim3(kAT), kargs2, sm2, st));
            }
        } else {
            hipLaunchKernelGGL((k_attc<false>), dim3(N, nres), dim3(256), smem, st, at);
        }
        LAUNCH_CHECK("k_attc<fwd>");
        dim3 pgrid(tiles_of(N), nres);
        if (host_retile) {
            po.rt_units = h->retile_units + h->n_retile_early;
            po.rt_n = h->n_retile - h->n_retile_early;
            po.rt_params = h->params;
            po.rt_tiled = h->tiled;
            po.rt_tiledT = h->reg8 ? h->tiledT : nullptr;
            po.rt_y0 = nres;
            pgrid.y += (po.rt_n + pgrid.x - 1) / pgrid.x;
        }
        launch_post_fwd<true, 128>(dff, pgrid, st, po);
        LAUNCH_CHECK("k_post_fwd<centre>");
        return 0;
    };

    if (trunk) {
    } else if (h->embed_dense) {   // Embedding with more than one layer: every row of every layer (cf_embed_full.h + dense layers)
        if (embed_dense_forward(h, bt, save != 0, st)) return -1;
    } else {   // Embedding
        CentreBuf* bufs[kMaxRes];
        const float* xin[kMaxRes];
        float* outp[kMaxRes];
        for (int r = 0; r < nres; ++r) {
            bufs[r] = &h->E[r];
            xin[r] = h->ex0[r];
            outp[r] = h->Rx[r][0];
        }
        // ... with lin_proj_p on the promoter centre embedding (net.py:118) as the last product of the layer's chain kernel
        const float* lw[kMaxRes];
        float* ly[kMaxRes];
        for (int r = 0; r < nres; ++r) {
            lw[r] = h->T_(fmt("pairwise_interaction.%d.lin_proj_p.weight", c.binsizes[r]));
            ly[r] = h->xp0[r];
        }
        if (centre_layer(bufs, ep, xin, identity_map(), bt->promoter_feats, bt->promoter_mask_row, bt->promoter_mask_stride, NE,
                         c.embed_dff, outp, RowMap{1, T, 0, 0}, false, c.embed_heads, lw, ly, false, nullptr, nullptr, defer))
            return -1;
    }
    if (h->embed_dense) {   // lin_proj_p on the promoter centre embedding (net.py:118)
        LinArgs a;
        for (int r = 0; r < nres; ++r) {
            a.x[r] = h->Rx[r][0];
            a.w[r] = h->T_(fmt("pairwise_interaction.%d.lin_proj_p.weight", c.binsizes[r]));
            a.b[r] = nullptr;
            a.y[r] = h->xp0[r];
        }
        a.xmap = RowMap{1, T, 0, 0};
        a.ldx = kD;
        a.ldy = kD;
        a.N = NE;
        a.K = kD;
        a.Nout = kD;
        a.relu = 0;
        hipLaunchKernelGGL((k_linear_fwd<2>), dim3(tiles_of(NE), 1, nres), dim3(256), 0, st, a);
        LAUNCH_CHECK("k_linear_fwd<lin_proj_p>");
    }
    for (int l = 0; l < (trunk ? 0 : c.pair_layers); ++l) {   // Pairwise layers
        CentreBuf* bufs[kMaxRes];
        const float* xin[kMaxRes];
        float* outp[kMaxRes];
        const bool last = l + 1 == c.pair_layers;
        for (int r = 0; r < nres; ++r) {
            pp[r] = pair_params(h, r, l);
            bufs[r] = &h->P[r][l];
            xin[r] = l == 0 ? h->xp0[r] : h->P[r][l - 1].out;
            outp[r] = last ? h->Rx[r][0] : h->P[r][l].out;
        }
        const RowMap xmap = l == 0 ? RowMap{S, 1, 0, 0} : identity_map();
        const RowMap omap = last ? RowMap{S, T, 1, 1} : identity_map();
        CentreParams npp[kMaxRes];
        CentreBuf* nbufs[kMaxRes];
        if (!last)
            for (int r = 0; r < nres; ++r) {
                npp[r] = pair_params(h, r, l + 1);
                nbufs[r] = &h->P[r][l + 1];
            }
        if (centre_layer(bufs, pp, xin, xmap, bt->pcre_feats, bt->pcre_mask_row, bt->pcre_mask_stride, NP, c.pair_dff, outp, omap,
                         l == 0, c.pair_heads, nullptr, nullptr, l > 0, last ? nullptr : npp, last ? nullptr : nbufs))
            return -1;
    }
    if (h->reg_fused) {   // Regulation: all layers in one launch, one workgroup per (gene, resolution)
        RegArgs ra;
        ra.tab = h->reg_tab;
        ra.n_layers = c.reg_layers;
        ra.T = T;
        ra.B = B;
        ra.n_res = nres;
        ra.xcd_map = h->xcd_map;
        for (int r = 0; r < nres; ++r) ra.mask[r] = bt->interaction_mask[r];
        ra.freq = bt->interaction_freq;
        ra.save = save;
        ra.tdbg = getenv("CF_STAMP") ? reinterpret_cast<unsigned long long*>(h->tdbg) : nullptr;
        memset(&ra.head, 0, sizeof ra.head);
        if (ride) ra.head = *ride;
        ra.row0_last = h->reg_row0 ? 1 : 0;
        ra.l_top = c.reg_layers - 1;
        ra.l_bot = 0;
        if (launch_reg(h, "k_reg_fwd", reg_kernel(false, c.reg_dff, save != 0), dim3(8 * ((B * nres + 7) / 8)), reg8_fwd_smem(c.reg_dff), ra, st)) return -1;
    }
    for (int l = 0; l < (h->reg_fused ? 0 : c.reg_layers); ++l) {   // Regulation layers, unfused fallback (T > 11)
        LinArgs la;
        AttrArgs at;
        PostArgs po;
        for (int r = 0; r < nres; ++r) {
            const std::string lp = fmt("regulation.%d.transformer.layers.%d.", c.binsizes[r], l);
            RegBuf& b = h->R[r][l];
            la.x[r] = h->Rx[r][l];
            la.w[r] = h->T_(lp + "self_att.att.weight");
            la.b[r] = nullptr;
            la.y[r] = b.qkvg;
            at.qkvg[r] = b.qkvg;
            at.mask[r] = bt->interaction_mask[r];
            at.gamma[r] = h->P_(lp + "self_att.gamma_f");
            at.p[r] = b.p;
            at.a[r] = b.a;
            at.dqkvg[r] = nullptr;
            at.dgam[r] = nullptr;
            po.x[r] = h->Rx[r][l];
            po.ain[r] = b.a;
            po.wv[r] = nullptr;
            po.wo[r] = h->T_(lp + "self_att.ff.weight");
            po.bo[r] = h->P_(lp + "self_att.ff.bias");
            po.g1[r] = h->P_(lp + "self_att.ln.weight");
            po.be1[r] = h->P_(lp + "self_att.ln.bias");
            po.w1[r] = h->T_(lp + "ff.l1.weight");
            po.b1[r] = h->P_(lp + "ff.l1.bias");
            po.w2[r] = h->T_(lp + "ff.l2.weight");
            po.b2[r] = h->P_(lp + "ff.l2.bias");
            po.g2[r] = h->P_(lp + "ff.ln.weight");
            po.be2[r] = h->P_(lp + "ff.ln.bias");
            po.a_out[r] = nullptr;
            po.xh1[r] = b.xh1;
            po.rs1[r] = b.rs1;
            po.y1[r] = b.y1;
            po.hdn[r] = b.hdn;
            po.xh2[r] = b.xh2;
            po.rs2[r] = b.rs2;
            po.out[r] = h->Rx[r][l + 1];
        }
        la.xmap = identity_map();
        la.ldx = kD;
        const int RDm = c.reg_dmodel, RW = 4 * RDm;
        la.ldy = RW;
        la.N = NR;
        la.K = kD;
        la.Nout = RW;
        la.relu = 0;
        at.freq = bt->interaction_freq;
        at.T = T;
        at.H = c.reg_heads;
        at.DM = RDm;
        po.xmap = identity_map();
        po.omap = identity_map();
        po.N = NR;
        po.save = save;
        if (kD == 64) hipLaunchKernelGGL((k_linear_fwd<4, 64>), dim3(tiles_of(NR), RW / 256, nres), dim3(256), 0, st, la);
        else hipLaunchKernelGGL((k_linear_fwd<4>), dim3(tiles_of(NR), RW / 256, nres), dim3(256), 0, st, la);
        LAUNCH_CHECK("k_linear_fwd<qkvg>");
        hipLaunchKernelGGL((k_attr<false>), dim3(B, nres), dim3(256), attr_smem(T, at.H, RDm, false), st, at);
        LAUNCH_CHECK("k_attr<fwd>");
        if (kD == 256) {
            if (RDm == 128) launch_post_fwd<false, 128, 256>(c.reg_dff, dim3(tiles_of(NR), nres), st, po);
            else launch_post_fwd<false, 256, 256>(c.reg_dff, dim3(tiles_of(NR), nres), st, po);
        } else if (kD == 64) {
            if (RDm == 128) launch_post_fwd<false, 128, 64>(c.reg_dff, dim3(tiles_of(NR), nres), st, po);
            else launch_post_fwd<false, 256, 64>(c.reg_dff, dim3(tiles_of(NR), nres), st, po);
        } else if (RDm == 128) launch_post_fwd<false, 128>(c.reg_dff, dim3(tiles_of(NR), nres), st, po);
        else launch_post_fwd<false, 256>(c.reg_dff, dim3(tiles_of(NR), nres), st, po);
        LAUNCH_CHECK("k_post_fwd<reg>");
    }
    h->head_deferred = save == 2 && !ride;
    h->head_done = ride != nullptr;
    h->deferred_logits_user = logits;
    if (save != 2 && (c.d_head != 128 || c.d_emb != 128)) {
        HeadGenArgs a;
        head_gen_args(h, B, logits, a);
        hipLaunchKernelGGL(k_head_gen_fwd, dim3(B), dim3(256), 0, st, a);
        LAUNCH_CHECK("k_head_gen_fwd");
    } else if (save != 2) {   // head (save = 2: together with the loss and its backward in cf_backward_part, one launch)
        HeadFwdArgs a;
        head_fwd_args(h, B, logits, a);
        hipLaunchKernelGGL(k_head_fwd, dim3(tiles_of(B)), dim3(kHeadThreads), 0, st, a);
        LAUNCH_CHECK("k_head_fwd");
    }
    h->last_fwd_B = save ? B : 0;
    h->n_fwd = (int)(g_launches - launches0);
    if (h->capturing) h->cap.n_fwd = h->n_fwd;
    return 0;
}

// ------------------------------------------------------------------------------------
// backward
extern "C" int cf_forward(cf_handle* h, const cf_batch* bt, float* logits, int save, void* stream) {
    return forward_impl(h, bt, logits, save, stream, nullptr);
}
// Can cf_forward_train run the head at the tail of the Regulation launch?  (512-thread Regulation kernels, three resolutions, the
// default head width; CF_HEAD_RIDE=0 at cf_create switches it off: A/B runs, cross-checks.)
extern "C" int cf_head_rides(cf_handle* h) {
    if (!h) return 0;
    return h->reg_fused && h->cfg.n_res == kMaxRes && h->cfg.d_head == kD && h->cfg.d_emb == kD && h->head_ride ? 1 : 0;
}
// cf_forward(save_for_backward = 2) for a training step whose labels are known at forward time: where cf_head_rides(h), the
// prediction head -- forward, loss, its backward down to the gradient of token 0 of every Regulation output -- runs at the tail of the
// Regulation forward launch, by the last of a gene's three workgroups to finish (cf_head_ride.h), and the cf_backward_part /
// cf_backward call that follows skips the head (its labels / loss arguments are ignored then).  Elsewhere it is cf_forward(save = 2).
// logits: caller's [B, n_out] buffer (may be null); loss_out: one float (may be null); loss_scale as in cf_backward.
// The head ride's per-gene arrival counters are monotonic (cf_head_ride.h): every launch adds n_res to each, and 2^32 is no multiple of 3 -- after
// 1.4e9 launches (eight days of uninterrupted steps) the winner test would drift.  Every 2^28 launches the counters are put back to zero by a
// stream-ordered memset IN FRONT of a launch: between two launches of a stream every counter is a multiple of n_res and nobody is arriving (zeroing
// from inside the launch, round 4, raced with the arrivals of a second process on the device).
// The launches of a handle may come in on more than one stream (two Trainers on one model, a caller's own stream): the reset waits for what the
// OTHER streams have queued so far and they wait for the reset, so it can never land under a ride launch in flight elsewhere.
static int ride_tick(cf_handle* h, hipStream_t st) {
    if (h->capturing) return 0;      // (a captured launch is counted when its graph is replayed: cf_graph_launch)
    if (std::find(h->ride_streams.begin(), h->ride_streams.end(), st) == h->ride_streams.end()) {
        if (h->ride_streams.size() >= 16) h->ride_streams.erase(h->ride_streams.begin());
        h->ride_streams.push_back(st);
    }
    if (++h->ride_launches >= h->ride_reset_every) {      // (CF_RIDE_RESET_EVERY at cf_create: the tests run with a handful)
        if (h->ride_streams.size() > 1) {
            if (!h->ride_ev) HIP_TRY(hipEventCreateWithFlags(&h->ride_ev, hipEventDisableTiming));
            for (size_t i = 0; i < h->ride_streams.size();) {
                hipStream_t o = h->ride_streams[i];
                if (o != st) {
                    if (hipEventRecord(h->ride_ev, o) != hipSuccess) {      // a stream its owner has destroyed since: nothing of it can be in flight
                        (void)hipGetLastError();
                        h->ride_streams.erase(h->ride_streams.begin() + i);
                        continue;
                    }
                    HIP_TRY(hipStreamWaitEvent(st, h->ride_ev, 0));
                }
                ++i;
            }
        }
        HIP_TRY(hipMemsetAsync(h->head_cnt, 0, (size_t)(h->cfg.max_batch + 1) * sizeof(int), st));
        if (h->ride_streams.size() > 1) {
            HIP_TRY(hipEventRecord(h->ride_ev, st));
            for (hipStream_t o : h->ride_streams)
                if (o != st) HIP_TRY(hipStreamWaitEvent(o, h->ride_ev, 0));
        }
        h->ride_launches = 0;
    }
    return 0;
}
extern "C" int cf_forward_train(cf_handle* h, const cf_batch* bt, float* logits, const void* labels, float loss_scale, float* loss_out,
                                void* stream) {
    if (!h) return fail("null handle");
    if (!labels || !cf_head_rides(h)) return forward_impl(h, bt, logits, 2, stream, nullptr);
    if (ride_tick(h, (hipStream_t)stream)) return -1;
    if (!h->grads) return fail("cf_forward_train: no gradient buffer bound");
    const cf_config& c = h->cfg;
    HeadRide hd;
    memset(&hd, 0, sizeof hd);
    hd.on = 1;
    hd.n_out = c.n_out;
    hd.gscale = loss_scale;
    hd.labels = labels;
    hd.w1_t = h->T_("fc_head.0.weight");
    hd.w1 = h->P_("fc_head.0.weight");
    hd.b1 = h->P_("fc_head.0.bias");
    hd.w2 = h->P_("fc_head.2.weight");
    hd.b2 = h->P_("fc_head.2.bias");
    hd.hin = h->hin, hd.h1 = h->h1, hd.logits = h->logits, hd.logits_user = logits, hd.dlogits = h->dlogits, hd.dh1 = h->dh1, hd.dhin = h->dhin;
    for (int r = 0; r < c.n_res; ++r) hd.dxl[r] = h->dRx[r][c.reg_layers];
    hd.loss = h->loss, hd.loss_part = h->loss_part, hd.loss_user = loss_out;
    hd.cnt = h->head_cnt;
    h->ride = hd;                 // (the backward launch finishes the mean loss)
    return forward_impl(h, bt, logits, 2, stream, &hd);
}

// ------------------------------------------------------------------------------------
// parts: 1 = head, 2 = Regulation stack, 4 = Pairwise + Embedding (the activation-gradient chain in order)
static int backward_impl(cf_handle* h, const cf_batch* bt, hipStream_t st, int parts = 7, const void* labels = nullptr,
                         float loss_scale = 1.f, float* loss_out = nullptr) {
    const cf_config& c = h->cfg;
    const int kD = c.d_emb;      // (row width: shadows cf::kD in this function)
    const int B = bt->B, S = c.i_max, T = S + 1, nres = c.n_res, F = c.n_feats;
    const int NE = B, NP = B * S, NR = B * T;
    if ((parts & 1) && h->head_done) {         // cf_forward_train has run head forward, loss and head backward already
        h->head_done = false;
        h->head_loss_due = true;
        parts &= ~1;
    }
    if ((parts & 1) && (c.d_head != 128 || c.d_emb != 128)) {      // loss + head, any hidden width / row width
        if (h->head_deferred && !labels) return fail("cf_backward: cf_forward(save_for_backward = 2) needs the fused loss (labels)");
        HeadGenArgs a;
        head_gen_args(h, B, h->head_deferred ? h->deferred_logits_user : nullptr, a);
        a.labels = labels;
        a.loss_user = loss_out;
        a.gscale = loss_scale;
        if (h->head_deferred) {
            hipLaunchKernelGGL(k_head_gen_fwd, dim3(B), dim3(256), 0, st, a);
            LAUNCH_CHECK("k_head_gen_fwd");
            h->head_deferred = false;
        }
        hipLaunchKernelGGL(k_head_gen_bwd, dim3(B), dim3(256), 0, st, a);
        LAUNCH_CHECK("k_head_gen_bwd");
    } else if (parts & 1) {   // loss + head
        HeadBwdArgs a;
        a.logits = h->logits;
        a.labels = labels;
        a.w2 = h->P_("fc_head.2.weight");
        a.h1 = h->h1;
        a.w1 = h->P_("fc_head.0.weight");
        a.dlogits = h->dlogits;
        a.dh1 = h->dh1;
        a.dhin = h->dhin;
        for (int r = 0; r < nres; ++r) a.dxl[r] = h->dRx[r][c.reg_layers];
        a.loss = h->loss;
        a.loss_part = h->loss_part;
        a.loss_user = loss_out;
        a.gscale = loss_scale;
        a.B = B;
        a.T = T;
        a.n_res = nres;
        a.n_out = c.n_out;
        a.tdbg = getenv("CF_STAMP_HEAD") ? reinterpret_cast<unsigned long long*>(h->tdbg) + 128 : nullptr;
        if (h->head_deferred) {      // the forward pass left the head to this call: forward, loss, backward in one launch
            if (!labels) return fail("cf_backward: cf_forward(save_for_backward = 2) needs the fused loss (labels)");
            HeadFwdArgs f;
            head_fwd_args(h, B, h->deferred_logits_user, f);
            hipLaunchKernelGGL(k_head_train, dim3(tiles_of(B)), dim3(kHeadThreads), 0, st, f, a);
            LAUNCH_CHECK("k_head_train");
            h->head_deferred = false;
        } else {
            hipLaunchKernelGGL(k_head_bwd, dim3(tiles_of(B)), dim3(kHeadThreads), 0, st, a);
            LAUNCH_CHECK("k_head_bwd");
        }
    }
    if ((parts & (CF_PART_REG_HI | CF_PART_REG_LO)) && !(parts & 2)) {      // the Regulation backward in halves (data-parallel schedule, cf_reg_halves)
        if (!cf_reg_halves(h)) return fail("cf_backward_part: this model's Regulation backward does not come in halves (cf_reg_halves)");
        if ((parts & (CF_PART_REG_HI | CF_PART_REG_LO)) == (CF_PART_REG_HI | CF_PART_REG_LO)) parts |= 2;
    }
    if ((parts & (2 | CF_PART_REG_HI | CF_PART_REG_LO)) && h->reg_fused) {
        const int half = c.reg_layers / 2;
        const int l_top = (parts & 2) || (parts & CF_PART_REG_HI) ? c.reg_layers - 1 : half - 1;
        const int l_bot = (parts & 2) || (parts & CF_PART_REG_LO) ? 0 : half;
        RegArgs ra;
        ra.l_top = l_top;
        ra.l_bot = l_bot;
        ra.tab = h->reg_tab;
        ra.n_layers = c.reg_layers;
        ra.T = T;
        ra.B = B;
        ra.n_res = nres;
        ra.xcd_map = h->xcd_map;
        for (int r = 0; r < nres; ++r) ra.mask[r] = bt->interaction_mask[r];
        ra.freq = bt->interaction_freq;
        ra.save = 1;
        ra.tdbg = getenv("CF_STAMP_BWD") ? reinterpret_cast<unsigned long long*>(h->tdbg) : nullptr;
        memset(&ra.head, 0, sizeof ra.head);
        if (h->head_loss_due) {
            ra.head = h->ride;
            if (loss_out) ra.head.loss_user = loss_out;
            h->head_loss_due = false;
        }
        ra.row0_last = h->reg_row0 ? 1 : 0;
        if (launch_reg(h, "k_reg_bwd", reg_kernel(true, c.reg_dff), dim3(8 * ((B * nres + 7) / 8)), reg8_bwd_smem(c.reg_dff), ra, st)) return -1;
    }
    for (int l = ((h->reg_fused || !(parts & 2)) ? -1 : c.reg_layers - 1); l >= 0; --l) {   // Regulation, unfused fallback
        PostBwdArgs pb;
        AttrArgs at;
        DgradArgs dg;
        for (int r = 0; r < nres; ++r) {
            const std::string lp = fmt("regulation.%d.transformer.layers.%d.", c.binsizes[r], l);
            RegBuf& b = h->R[r][l];
            pb.dout[r] = h->dRx[r][l + 1];
            pb.xh2[r] = b.xh2;
            pb.rs2[r] = b.rs2;
            pb.g2[r] = h->P_(lp + "ff.ln.weight");
            pb.hdn[r] = b.hdn;
            pb.w2[r] = h->P_(lp + "ff.l2.weight");
            pb.w1[r] = h->P_(lp + "ff.l1.weight");
            pb.xh1[r] = b.xh1;
            pb.rs1[r] = b.rs1;
            pb.g1[r] = h->P_(lp + "self_att.ln.weight");
            pb.wo[r] = h->P_(lp + "self_att.ff.weight");
            pb.wv[r] = nullptr;
            pb.dt2[r] = b.dt2;
            pb.dpre1[r] = b.dpre1;
            pb.dt1[r] = b.dt1;
            pb.da[r] = b.da;
            pb.dxbar[r] = nullptr;
            pb.partial[r] = b.partial;
            at.qkvg[r] = b.qkvg;
            at.mask[r] = bt->interaction_mask[r];
            at.gamma[r] = h->P_(lp + "self_att.gamma_f");
            at.p[r] = b.p;
            at.a[r] = b.da;
            at.dqkvg[r] = b.dqkvg;
            at.dgam[r] = b.dgam;
            dg.dy[r] = b.dqkvg;
            dg.w[r] = h->P_(lp + "self_att.att.weight");
            dg.res[r] = b.dt1;
            dg.dx[r] = h->dRx[r][l];
        }
        pb.dmap = identity_map();
        pb.N = NR;
        const int RDm = c.reg_dmodel, RW = 4 * RDm;
        at.freq = bt->interaction_freq;
        at.T = T;
        at.H = c.reg_heads;
        at.DM = RDm;
        dg.lddy = RW;
        dg.ldw = kD;
        dg.rmap = identity_map();
        dg.ldres = kD;
        dg.lddx = kD;
        dg.N = NR;
        dg.K = RW;
        dg.Ncols = kD;
        if (kD == 256) {
            if (RDm == 128) launch_post_bwd<false, 128, 256>(c.reg_dff, dim3(tiles_of(NR), nres), st, pb);
            else launch_post_bwd<false, 256, 256>(c.reg_dff, dim3(tiles_of(NR), nres), st, pb);
        } else if (kD == 64) {
            if (RDm == 128) launch_post_bwd<false, 128, 64>(c.reg_dff, dim3(tiles_of(NR), nres), st, pb);
            else launch_post_bwd<false, 256, 64>(c.reg_dff, dim3(tiles_of(NR), nres), st, pb);
        } else if (RDm == 128) launch_post_bwd<false, 128>(c.reg_dff, dim3(tiles_of(NR), nres), st, pb);
        else launch_post_bwd<false, 256>(c.reg_dff, dim3(tiles_of(NR), nres), st, pb);
        LAUNCH_CHECK("k_post_bwd<reg>");
        hipLaunchKernelGGL((k_attr<true>), dim3(B, nres), dim3(256), attr_smem(T, at.H, RDm, true), st, at);
        LAUNCH_CHECK("k_attr<bwd>");
        if (RDm == 128) hipLaunchKernelGGL((k_dgrad<8>), dim3(tiles_of(NR), kD / 32, nres), dim3(256), 0, st, dg);
        else hipLaunchKernelGGL((k_dgrad<16>), dim3(tiles_of(NR), kD / 32, nres), dim3(256), 0, st, dg);
        LAUNCH_CHECK("k_dgrad<qkvg>");
    }
    if (!(parts & 4)) return 0;
    if (h->pend_record && !h->trunk) {      // (cf_record_step_bwd without the fused trunk: a launch of its own, here)
        hipLaunchKernelGGL(k_record_step, dim3(1), dim3(256), 0, st, h->pend_rec);
        LAUNCH_CHECK("k_record_step");
        h->pend_record = false;
    }
    if (h->trunk) {      // Pairwise + Embedding backward, the join and the 7-mark projection partials: one launch (cf_trunk.h)
        TrunkArgs ta;
        trunk_args(h, bt, ta, 1);
        if (h->pend_record) {
            ta.rec = h->pend_rec;
            h->pend_record = false;
        }
        ta.lp_jobs = h->lp_jobs;
        ta.rd_tiles = nullptr;
        ta.rd_n = 0;
        ta.rd_batch = B;
        memset(&ta.rd_opt, 0, sizeof ta.rd_opt);
        const bool riding = h->rider.armed;
        if (riding) {      // (cf_rider_arm) one tile per rider wave at a time: any leading part of the bucket's table
            if (h->capturing) return fail("cf_backward_part: armed riders carry this step's AdamW scalars as launch arguments and cannot be captured");
            ta.rd_n = std::min(h->rider.max_tiles, h->n_wg_r - h->n_wg_short);
            ta.rd_tiles = h->wg_tiles + h->n_wg_short;
            ta.rd_opt = h->rider.o;
        }
        void* kargs[] = {&ta};
        const size_t rider_lds = (size_t)(kAT / 64) * kWgWaveLds * sizeof(float);      // eight wave-private stages
        // one tile per rider wave: rows of B workgroups x 8 waves until every tile has a wave.  The first (CUs - 3 B) workgroups start at once on
        // the idle CUs, the others as the short-resolution workgroups of the trunk (dispatched last, done first) leave theirs
        const int rider_rows = ta.rd_n > 0 ? (ta.rd_n + B * (kAT / 64) - 1) / (B * (kAT / 64)) : 0;
        h->time_mark("k_trunk_bwd", st);
        const hipError_t le = hipLaunchKernel(trunk_kernel(true, c.embed_dff, c.pair_dff, c.pair_layers), dim3(B, nres + rider_rows), dim3(kAT), kargs,
                                              ta.rd_n > 0 ? std::max(h->trunk_smem_bytes, rider_lds) : h->trunk_smem_bytes, st);
        h->time_mark("k_trunk_bwd", st);
        ++g_launches;
        if (le != hipSuccess || hipGetLastError() != hipSuccess) {
            // nothing was reduced or stepped: the riders stay armed for a retry, rider.done stays 0 and the reduction launch of the
            // step covers every tile
            return fail("launch k_trunk_bwd failed: %s", hipGetErrorString(le));
        }
        if (riding) {      // only a launch that was accepted counts as having reduced (and stepped) its tiles
            h->rider.armed = false;
            h->rider.done = ta.rd_n;
        }
        return 0;
    }
    // one centre-row layer backward: post chain -> attention -> query chain
    auto centre_bwd = [&](CentreBuf* bufs[kMaxRes], const CentreParams* prm, const float* const* dout, RowMap dmap,
                          const float* const* feats, const uint8_t* const* mask, const long long* mstride, int N, int dff, int nh) -> int {
        const float scale_c = sqrtf((float)(kD / nh));
        PostBwdArgs pb;
        AttcArgs at;
        QBwdArgs qb;
        size_t smem = 0;
        for (int r = 0; r < nres; ++r) {
            CentreBuf& b = *bufs[r];
            pb.dout[r] = dout[r];
            pb.xh2[r] = b.xh2;
            pb.rs2[r] = b.rs2;
            pb.g2[r] = prm[r].g2;
            pb.hdn[r] = b.hdn;
            pb.w2[r] = prm[r].w2;
            pb.w1[r] = prm[r].w1;
            pb.xh1[r] = b.xh1;
            pb.rs1[r] = b.rs1;
            pb.g1[r] = prm[r].g1;
            pb.wo[r] = prm[r].wo;
            pb.wv[r] = prm[r].wv;
            pb.dt2[r] = b.dt2;
            pb.dpre1[r] = b.dpre1;
            pb.dt1[r] = b.dt1;
            pb.da[r] = b.da;
            pb.dxbar[r] = b.dxbar;
            pb.partial[r] = b.partial;
            at.feats[r] = feats[r];
            at.mask[r] = mask[r];
            at.mstride[r] = mstride[r];
            at.pe[r] = h->pe[r];
            at.pet[r] = h->pet[r];
            at.wlp[r] = prm[r].wlp;
            at.vin[r] = b.dxbar;
            at.p[r] = b.p;
            at.w[r] = b.du;
            at.vout[r] = b.dqt;
            at.L[r] = c.n_bins[r];
            smem = std::max(smem, attc_smem(c.n_bins[r], F, true, nh, kD));
            qb.dqt[r] = b.dqt;
            qb.dres[r] = b.dt1;
            qb.wk[r] = prm[r].wk_t;     // NT product in the backward: tiled copy
            qb.wq[r] = prm[r].wq;
            qb.dq[r] = b.dq;
            qb.dx[r] = b.dx;
        }
        pb.dmap = dmap;
        pb.N = N;
        at.F = F;
        at.scale = scale_c;
        qb.N = N;
        if (kD == 256) return nh == 1 ? centre_bwd_heads<1, 256>(st, N, nres, dff, pb, at, smem, qb) : centre_bwd_heads<2, 256>(st, N, nres, dff, pb, at, smem, qb);
        if (kD == 64) return nh == 1 ? centre_bwd_heads<1, 64>(st, N, nres, dff, pb, at, smem, qb) : centre_bwd_heads<2, 64>(st, N, nres, dff, pb, at, smem, qb);
        if (nh == 1) return centre_bwd_heads<1>(st, N, nres, dff, pb, at, smem, qb);
        if (nh == 4) return centre_bwd_heads<4>(st, N, nres, dff, pb, at, smem, qb);
        launch_post_bwd<true, 128>(dff, dim3(tiles_of(N), nres), st, pb);
        LAUNCH_CHECK("k_post_bwd<centre>");
        if (h->attc2) {
            const int ag = attc2_regions_per_wg(N, h->attc_cap);
            Attc2Args a2;
            size_t sm2 = 0;
            for (int r = 0; r < nres; ++r) {
                a2.feats[r] = at.feats[r];
                a2.mask[r] = at.mask[r];
                a2.mstride[r] = at.mstride[r];
                a2.pe[r] = h->pe2[r];
                a2.pet[r] = h->pet2[r];
                a2.wlp[r] = at.wlp[r];
                a2.vin[r] = at.vin[r];
                a2.p[r] = at.p[r];
                a2.w[r] = at.w[r];
                a2.vout[r] = at.vout[r];
                a2.L[r] = at.L[r];
                a2.Lpad[r] = attc2_lpad(at.L[r]);
                a2.LT[r] = attc2_lt(at.L[r]);
                sm2 = std::max(sm2, attc2_smem(at.L[r], F, ag));
            }
            a2.N = N;
            a2.F = F;
            a2.scale = scale_c;
            a2.rscale = 1.0f / a2.scale;
            a2.tdbg = (getenv("CF_STAMP_ATTC") && (!getenv("CF_STAMP_ATTC_AG") || atoi(getenv("CF_STAMP_ATTC_AG")) == ag))
                          ? reinterpret_cast<unsigned long long*>(h->tdbg) + 256 : nullptr;
            a2.tall = (getenv("CF_STAMP_ATTC_ALL") && atoi(getenv("CF_STAMP_ATTC_ALL")) == 1 && (!getenv("CF_STAMP_ATTC_AG") || atoi(getenv("CF_STAMP_ATTC_AG")) == ag))
                          ? reinterpret_cast<unsigned long long*>(h->tdbg) + 256 : nullptr;
            void* kargs2[] = {&a2};
            if (ag == 1 && h->attc1) {
                size_t sm1 = 0;
                for (int r = 0; r < nres; ++r) sm1 = std::max(sm1, attc1_smem(at.L[r], F));
                HIP_TRY(hipLaunchKernel((const void*)k_attc1<true>, dim3(N, nres), dim3(kAT), kargs2, sm1, st));
            } else {
                HIP_TRY(hipLaunchKernel(attc2_kernel<true>(ag), dim3((N + ag - 1) / ag, nres), dim3(kAT), kargs2, sm2, st));
            }
        } else {
            hipLaunchKernelGGL((k_attc<true>), dim3(N, nres), dim3(256), smem, st, at);
        }
        LAUNCH_CHECK("k_attc<bwd>");
        hipLaunchKernelGGL((k_qchain_bwd<kPostWaves>), dim3(tiles_of(N), nres), dim3(kPostWaves * 64), 0, st, qb);
        LAUNCH_CHECK("k_qchain_bwd");
        return 0;
    };
    CentreParams prm[kMaxRes];
    for (int l = c.pair_layers - 1; l >= 0; --l) {   // Pairwise
        CentreBuf* bufs[kMaxRes];
        const float* dout[kMaxRes];
        const bool last = l + 1 == c.pair_layers;
        for (int r = 0; r < nres; ++r) {
            prm[r] = pair_params(h, r, l);
            bufs[r] = &h->P[r][l];
            dout[r] = last ? h->dRx[r][0] : h->P[r][l + 1].dx;
        }
        if (centre_bwd(bufs, prm, dout, last ? RowMap{S, T, 1, 1} : identity_map(), bt->pcre_feats, bt->pcre_mask_row,
                       bt->pcre_mask_stride, NP, c.pair_dff, c.pair_heads))
            return -1;
    }
    {   // join the streams meeting at the promoter embedding, back through lin_proj_p: one launch
        JoinDgradArgs a;
        for (int r = 0; r < nres; ++r) {
            a.dxp[r] = h->P[r][0].dx;
            a.dx0[r] = h->dRx[r][0];
            a.w[r] = h->P_(fmt("pairwise_interaction.%d.lin_proj_p.weight", c.binsizes[r]));
            a.dxp0[r] = h->dxp0[r];
            a.dx[r] = h->edout[r];
        }
        a.dhin = h->dhin;
        a.B = B;
        a.S = S;
        a.T = T;
        a.n_res = nres;
        if (kD == 256) hipLaunchKernelGGL(k_join_dgrad<256>, dim3(tiles_of(NE), kD / 32, nres), dim3(256), 0, st, a);
        else if (kD == 64) hipLaunchKernelGGL(k_join_dgrad<64>, dim3(tiles_of(NE), kD / 32, nres), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_join_dgrad<128>, dim3(tiles_of(NE), kD / 32, nres), dim3(256), 0, st, a);
        LAUNCH_CHECK("k_join_dgrad");
    }
    if (h->embed_dense) {
        if (embed_dense_backward(h, bt, st)) return -1;      // writes the Embedding gradients directly (no deferred tiles)
    } else {   // Embedding
        CentreBuf* bufs[kMaxRes];
        const float* dout[kMaxRes];
        for (int r = 0; r < nres; ++r) {
            prm[r] = embed_params(h, r);
            bufs[r] = &h->E[r];
            dout[r] = h->edout[r];
        }
        if (centre_bwd(bufs, prm, dout, identity_map(), bt->promoter_feats, bt->promoter_mask_row, bt->promoter_mask_stride, NE,
                       c.embed_dff, c.embed_heads))
            return -1;
    }
    return 0;
}

#ifndef CF_MERGE_REDUCE
#define CF_MERGE_REDUCE 1
#endif
constexpr bool kMergeReduce = CF_MERGE_REDUCE;
// deferred weight / bias gradients: one launch per bucket over the two tile tables
static int reduce_impl(cf_handle* h, int B, hipStream_t st, int buckets = CF_BUCKET_REG | CF_BUCKET_PE) {
    if (h->rider.done) return fail("gradient reduction: the riders of step %lld have updated part of the Regulation + head bucket; finish the step with cf_reduce_opt_part", h->rider.step);
    if ((buckets & CF_BUCKET_PE) && !h->trunk) {      // (the fused trunk backward writes these partials itself)
        hipLaunchKernelGGL(k_wgrad_lp, dim3((B + kLpGenes - 1) / kLpGenes, h->n_lp), dim3(256), 0, st, (const LpJob*)h->lp_jobs, B, h->cfg.d_emb);
        LAUNCH_CHECK("k_wgrad_lp");
    }
    if (buckets & CF_BUCKET_REG) buckets |= CF_BUCKET_REG_HI | CF_BUCKET_REG_LO;
    if ((buckets & (CF_BUCKET_REG_HI | CF_BUCKET_REG_LO)) == (CF_BUCKET_REG_HI | CF_BUCKET_REG_LO)) buckets |= CF_BUCKET_REG;
    for (int bk = 0; bk < 4; ++bk) {      // the whole Regulation + head bucket (one launch), else its halves; Embedding + Pairwise
        int w0, wn, c0, cn;
        if (bk == 0) {
            if (!(buckets & CF_BUCKET_REG)) continue;
            w0 = 0, wn = h->n_wg_r, c0 = 0, cn = h->n_cs_r;
        } else if (bk == 1) {
            if ((buckets & CF_BUCKET_REG) || !(buckets & CF_BUCKET_REG_HI)) continue;
            w0 = 0, wn = h->n_wg_hi, c0 = 0, cn = h->n_cs_hi;
        } else if (bk == 2) {
            if ((buckets & CF_BUCKET_REG) || !(buckets & CF_BUCKET_REG_LO)) continue;
            w0 = h->n_wg_hi, wn = h->n_wg_r - h->n_wg_hi, c0 = h->n_cs_hi, cn = h->n_cs_r - h->n_cs_hi;
        } else {
            if (!(buckets & CF_BUCKET_PE)) continue;
            w0 = h->n_wg_r, wn = h->n_wg - h->n_wg_r, c0 = h->n_cs_r, cn = h->n_cs - h->n_cs_r;
        }
        if (!kMergeReduce || h->timed == "k_wgrad" || h->timed == "k_colsum") {      // timed separately
            h->time_mark("k_wgrad", st);
            hipLaunchKernelGGL(k_wgrad, dim3(xcd_grid(wn)), dim3(256), 0, st, (const WgTile*)h->wg_tiles + w0, wn, B, h->xcd_reduce);
            h->time_mark("k_wgrad", st);
            LAUNCH_CHECK("k_wgrad");
            h->time_mark("k_colsum", st);
            hipLaunchKernelGGL(k_colsum, dim3(cn), dim3(256), 0, st, (const CsTile*)h->cs_tiles + c0, B);
            h->time_mark("k_colsum", st);
        } else {
            hipLaunchKernelGGL(k_reduce, dim3(xcd_grid(wn) + cn), dim3(256), 0, st, (const WgTile*)h->wg_tiles + w0, wn, (const CsTile*)h->cs_tiles + c0, B, h->xcd_reduce);
        }
        LAUNCH_CHECK("k_colsum");
    }
    return 0;
}

// `first`: the call starts a backward pass (later pieces may follow a replayed graph, which bypasses the host-side record)
static int check_bwd(cf_handle* h, const cf_batch* bt, bool first = true) {
    if (check_batch(h, bt)) return -1;
    if (!h->grads) return fail("cf_backward: no gradient buffer bound");
    if (first && h->last_fwd_B != bt->B) return fail("cf_backward must follow cf_forward(save_for_backward=1) on the same batch");
    return 0;
}

extern "C" int cf_backward_part(cf_handle* h, const cf_batch* bt, const void* labels, float loss_scale, float* loss_out, int parts,
                                void* stream) {
    if (check_bwd(h, bt, (parts & 1) != 0)) return -1;
    hipStream_t st = (hipStream_t)stream;
    if ((parts & 1) && !labels && !h->head_done) return fail("cf_backward: labels is null");
    const long long launches0 = g_launches;
    if (parts & 1) h->n_bwd = h->n_opt = 0;      // a backward pass starts with the head: its pieces, the bucket reductions and the optimiser launches add up
    const int rc = backward_impl(h, bt, st, parts, labels, loss_scale, loss_out);
    h->n_bwd += (int)(g_launches - launches0);
    if (h->capturing) {
        if (parts & 1) h->cap.starts_bwd = true, h->cap.n_bwd = 0;
        h->cap.n_bwd += (int)(g_launches - launches0);
    }
    return rc;
}

extern "C" int cf_backward_chain(cf_handle* h, const cf_batch* bt, const void* labels, float loss_scale, float* loss_out,
                                 void* stream) {
    return cf_backward_part(h, bt, labels, loss_scale, loss_out, 7, stream);
}

extern "C" int cf_backward_reduce(cf_handle* h, int B, void* stream) {
    if (!h || !h->grads) return fail("cf_backward_reduce: no gradient buffer bound");
    if (B < 1 || B > h->cfg.max_batch) return fail("cf_backward_reduce: bad batch size %d", B);
    const long long launches0 = g_launches;
    const int rc = reduce_impl(h, B, (hipStream_t)stream);
    h->n_bwd += (int)(g_launches - launches0);
    return rc;
}

extern "C" int cf_backward_reduce_part(cf_handle* h, int B, int buckets, void* stream) {
    if (!h || !h->grads) return fail("cf_backward_reduce_part: no gradient buffer bound");
    if (B < 1 || B > h->cfg.max_batch) return fail("cf_backward_reduce_part: bad batch size %d", B);
    if (buckets & ~(CF_BUCKET_REG | CF_BUCKET_PE | CF_BUCKET_REG_HI | CF_BUCKET_REG_LO)) return fail("cf_backward_reduce_part: bad bucket mask %d", buckets);
    if ((buckets & (CF_BUCKET_REG_HI | CF_BUCKET_REG_LO)) && !(buckets & CF_BUCKET_REG) && !cf_reg_halves(h))
        return fail("cf_backward_reduce_part: this model's Regulation bucket does not come in halves (cf_reg_halves)");
    const long long launches0 = g_launches;
    const int rc = reduce_impl(h, B, (hipStream_t)stream, buckets);
    h->n_bwd += (int)(g_launches - launches0);
    if (h->capturing) h->cap.n_bwd += (int)(g_launches - launches0);
    return rc;
}

extern "C" int cf_reg_halves(cf_handle* h) { return h && h->reg_fused && h->cfg.reg_layers >= 2 ? h->cfg.reg_layers / 2 : 0; }
extern "C" int cf_grad_bucket(cf_handle* h, int bucket, long long* offset, long long* numel) {
    if (!h || !offset || !numel) return fail("cf_grad_bucket: null argument");
    if (bucket == CF_BUCKET_PE) {
        *offset = 0;
        *numel = h->bucket_split;
    } else if (bucket == CF_BUCKET_REG) {
        *offset = h->bucket_split;
        *numel = h->lay.n_active - h->bucket_split;
    } else if (bucket == CF_BUCKET_REG_HI) {
        *offset = h->bucket_split_hi;
        *numel = h->lay.n_active - h->bucket_split_hi;
    } else if (bucket == CF_BUCKET_REG_LO) {
        *offset = h->bucket_split;
        *numel = h->bucket_split_hi - h->bucket_split;
    } else {
        return fail("cf_grad_bucket: bucket must be CF_BUCKET_PE, CF_BUCKET_REG, CF_BUCKET_REG_HI or CF_BUCKET_REG_LO");
    }
    return 0;
}

extern "C" int cf_backward(cf_handle* h, const cf_batch* bt, const void* labels, float loss_scale, float* loss_out, void* stream) {
    if (cf_backward_chain(h, bt, labels, loss_scale, loss_out, stream)) return -1;
    return reduce_impl(h, bt->B, (hipStream_t)stream);
}

extern "C" int cf_backward_from(cf_handle* h, const cf_batch* bt, const float* dlogits, void* stream) {
    if (check_bwd(h, bt)) return -1;
    if (!dlogits) return fail("cf_backward_from: dlogits is null");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(h->dlogits, dlogits, (size_t)bt->B * h->cfg.n_out * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (backward_impl(h, bt, st)) return -1;
    return reduce_impl(h, bt->B, st);
}

// ------------------------------------------------------------------------------------
// hipGraph capture of launch sequences (the per-step sequence is static)
// ------------------------------------------------------------------------------------
extern "C" int cf_capture_begin(cf_handle* h, void* stream) {
    if (!h) return fail("null handle");
    if (h->capturing) return fail("cf_capture_begin: already capturing");
    h->cap = cf_handle::Replay();
    HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
    h->capturing = true;
    return 0;
}
extern "C" int cf_capture_end(cf_handle* h, void* stream, int* graph_id) {
    if (!h || !graph_id) return fail("null argument");
    if (!h->capturing) return fail("cf_capture_end: not capturing");
    h->capturing = false;
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    if (h->cap.has_hole) h->cap.second = ge;
    else h->cap.first = ge;
    h->replays.push_back(h->cap);
    *graph_id = (int)h->replays.size() - 1;
    return 0;
}
extern "C" int cf_graph_launch(cf_handle* h, int graph_id, void* stream) {
    if (!h || graph_id < 0 || graph_id >= (int)h->replays.size()) return fail("cf_graph_launch: bad graph id %d", graph_id);
    cf_handle::Replay& rp = h->replays[graph_id];
    hipStream_t st = (hipStream_t)stream;
    if (rp.n_fwd >= 0) {
        h->n_fwd = rp.n_fwd;
        h->adv_next = nullptr;      // (a replayed forward pass moves the cursor on by itself if it was captured that way)
        h->pend_key = nullptr;
    }
    if (rp.starts_bwd) h->n_bwd = h->n_opt = 0;
    h->n_bwd += rp.n_bwd;
    if (rp.n_fwd >= 0 && h->head_cnt && ride_tick(h, st)) return -1;      // (a replayed forward pass may hold a head ride: counted as one)
    HIP_TRY(hipGraphLaunch(rp.first, st));
    if (rp.has_hole) {
        void* kargs[] = {&rp.hole.args};
        h->time_mark(h->timed.c_str(), st);
        HIP_TRY(hipLaunchKernel(rp.hole.func, rp.hole.grid, rp.hole.block, kargs, rp.hole.smem, st));
        h->time_mark(h->timed.c_str(), st);
        HIP_TRY(hipGraphLaunch(rp.second, st));
    }
    return 0;
}

// ------------------------------------------------------------------------------------
// HIP-event timing of one eagerly launched kernel ("k_wgrad", "k_colsum", "k_adamw")
// ------------------------------------------------------------------------------------
extern "C" int cf_timing_select(cf_handle* h, const char* kernel) {
    if (!h) return fail("null handle");
    h->timed = kernel ? kernel : "";
    h->ev_used = 0;
    return 0;
}
extern "C" int cf_timing_read(cf_handle* h, float* total_ms, int* count) {
    if (!h || !total_ms || !count) return fail("null argument");
    float tot = 0.f;
    int n = 0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(h->ev[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        tot += ms;
        ++n;
    }
    h->ev_used = 0;
    *total_ms = tot;
    *count = n;
    return 0;
}
extern "C" double cf_wgrad_flops(cf_handle* h, int B) { return h ? h->wg_flops_per_gene * B : 0.0; }
// Algorithmic flops (2 * MAC, valid rows only -- the dead rows of a 16-row tile are not counted) of one launch.
extern "C" double cf_kernel_flops(cf_handle* h, const char* kernel, int B) {
    if (!h || !kernel) return 0.0;
    const cf_config& c = h->cfg;
    const double kD = c.d_emb;      // (row width: shadows cf::kD in this function)
    const double T = c.i_max + 1, dff = c.reg_dff;
    const std::string k = kernel;
    if (k == "k_wgrad") return h->wg_flops_per_gene * B;
    const double RDm = c.reg_dmodel;
    const double lin_fwd = 2.0 * T * (kD * 4.0 * RDm + RDm * (double)kD + kD * dff + dff * kD);        // q|k|v|g, out-proj, FFN
    const double att_fwd = 2.0 * T * T * RDm * 2.0;                                                    // q k^T and p v
    // With the last layer reduced to what token 0 of its output needs (cf_reg8.h, b_run_row0) that layer's algorithmic work is smaller and is
    // counted as such: keys / values (forward) and the k, v quarters of the input-gradient product (backward) over all T rows, everything else
    // -- q, gate, out-projection, FFN, the attention products -- for ONE row.
    const bool row0 = h->reg_fused && h->reg_row0;
    const double full_layers = c.reg_layers - (row0 ? 1 : 0);
    const double lin_last = 2.0 * (T * kD * 2.0 * RDm + kD * 2.0 * RDm + RDm * (double)kD + kD * dff + dff * kD);
    if (k == "k_reg_fwd") return ((lin_fwd + att_fwd) * full_layers + (row0 ? lin_last + 2.0 * T * RDm * 2.0 : 0.0)) * c.n_res * B;
    if (k == "k_reg_bwd")                                                                              // dX products + p v, dp, dq, dk, dv
        return ((lin_fwd + 2.0 * T * T * RDm * 5.0) * full_layers + (row0 ? lin_last + 2.0 * T * RDm * 5.0 : 0.0)) * c.n_res * B;
    if (k == "k_trunk_fwd" || k == "k_trunk_bwd") {
        // The centre-row trunk of one (gene, resolution): ONE Embedding row and S = i_max Pairwise rows per layer.  Per row and layer the
        // 128-wide products q = x Wq^T, qt[h] = q[h] Wk[h], a[h] = xbar[h] Wv[h]^T, a Wo^T (4 x 128 x 128 MACs), the FFN (2 x 128 x d_ff) and the
        // attention of 2 heads over the L bins of the region in two passes of (128 + F) MACs per bin and head (scores, weighted sum; the key /
        // value projections are absorbed into the query); lin_proj_p on the Embedding row (128 x 128), the
        // Embedding input row (F x 128, forward only).  Backward: the dX product of each of these maps and the attention backward
        // (the same two passes with (dxbar, p) in and (dqt, du) out); weight gradients belong to k_wgrad, riders are not counted.
        const double S = c.i_max, F = c.n_feats, PL = c.pair_layers;
        const bool bwd = k == "k_trunk_bwd";
        double mac = 0.0;
        for (int r = 0; r < c.n_res; ++r) {
            const double L = c.n_bins[r];
            const double row_e = 4.0 * kD * kD + 2.0 * kD * c.embed_dff + kD * kD + (bwd ? 0.0 : F * kD);
            const double row_p = 4.0 * kD * kD + 2.0 * kD * c.pair_dff;
            const double att = 2.0 * 2.0 * L * (kD + F);
            mac += row_e + att + S * PL * (row_p + att);
        }
        return 2.0 * mac * B;
    }
    return 0.0;
}
extern "C" int cf_cu_count(cf_handle* h) { return h ? h->n_cu : 0; }

// ------------------------------------------------------------------------------------
// optimiser
// ------------------------------------------------------------------------------------
static int adam_hyper(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay, long long step, AdamHyper& hy) {
    if (!h || !h->params || !h->grads || !h->m || !h->v) return fail("cf_adamw_step: params / grads / moments not bound");
    if (step < 1) return fail("cf_adamw_step: step is 1-based");
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    hy.decay = (float)(1.0 - (double)lr * (double)weight_decay);
    hy.one_m_b1 = (float)(1.0 - (double)beta1);
    hy.b2 = beta2;
    hy.one_m_b2 = (float)(1.0 - (double)beta2);
    hy.step_size = (float)((double)lr / bc1);
    hy.bc2_sqrt = (float)std::sqrt(bc2);
    hy.eps = eps;
    hy.pad = 0.f;
    return 0;
}
// the buckets are adjacent ranges of the flat buffers: [0, split) = Embedding + Pairwise, [split, n_active) = Regulation + head;
// every tensor starts 16-byte aligned, so both bounds are multiples of 4
static int adam_range(cf_handle* h, int buckets, long long& lo, long long& n4, int& grid) {
    if (!buckets || (buckets & ~(CF_BUCKET_REG | CF_BUCKET_PE | CF_BUCKET_REG_HI | CF_BUCKET_REG_LO))) return fail("cf_adamw_step_part: bad bucket mask %d", buckets);
    if (buckets & CF_BUCKET_REG) buckets |= CF_BUCKET_REG_HI | CF_BUCKET_REG_LO;
    // the buckets lie [PE | REG_LO | REG_HI]: the mask must name an adjacent run of them
    const bool pe = buckets & CF_BUCKET_PE, lo_ = buckets & CF_BUCKET_REG_LO, hi_ = buckets & CF_BUCKET_REG_HI;
    if (pe && hi_ && !lo_) return fail("cf_adamw_step_part: the buckets of mask %d are not adjacent in the flat buffers", buckets);
    lo = pe ? 0 : (lo_ ? h->bucket_split : h->bucket_split_hi);
    const long long hi = hi_ ? h->lay.n_active : (lo_ ? h->bucket_split_hi : h->bucket_split);
    n4 = (hi - lo) / 4;
    grid = (int)std::min<long long>((n4 + 255) / 256, 256 * 8);
    return 0;
}
extern "C" int cf_adamw_step_part(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay, long long step,
                                  int buckets, void* stream) {
    AdamHyper hy;
    long long lo, n4;
    int grid;
    if (adam_hyper(h, lr, beta1, beta2, eps, weight_decay, step, hy) || adam_range(h, buckets, lo, n4, grid)) return -1;
    // cf_keep_tiled: a launch over exactly the Embedding + Pairwise bucket writes the tiled copies of what it steps (k_adamw_tiled); any other
    // range that touches that bucket steps the parameters only and leaves the copies stale
    const bool tiled = h->keep_tiled && h->tiled_map && buckets == CF_BUCKET_PE;
    if (buckets & CF_BUCKET_PE) h->tiled_pe_fresh = tiled;
    h->time_mark("k_adamw", (hipStream_t)stream);
    if (tiled)
        hipLaunchKernelGGL(k_adamw_tiled, dim3(grid), dim3(256), 0, (hipStream_t)stream, h->params + lo, (const float*)h->grads + lo, h->m + lo,
                           h->v + lo, n4, hy.decay, hy.one_m_b1, hy.b2, hy.one_m_b2, hy.step_size, hy.bc2_sqrt, hy.eps, (const int*)h->tiled_map, h->tiled);
    else
        hipLaunchKernelGGL(k_adamw, dim3(grid), dim3(256), 0, (hipStream_t)stream, h->params + lo, (const float*)h->grads + lo, h->m + lo,
                           h->v + lo, n4, hy.decay, hy.one_m_b1, hy.b2, hy.one_m_b2, hy.step_size, hy.bc2_sqrt, hy.eps);
    h->time_mark("k_adamw", (hipStream_t)stream);
    LAUNCH_CHECK("k_adamw");
    h->n_opt += 1;      // (reset when a backward pass starts: the optimiser launches of a step add up)
    return 0;
}
// The deferred gradients of `reduce_buckets` and the AdamW update of `adam_buckets` (disjoint from them, gradients already
// complete) in ONE launch (k_reduce_adamw).
extern "C" int cf_reduce_adamw_part(cf_handle* h, int B, int reduce_buckets, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    long long step, int adam_buckets, void* stream) {
    if (!h || !h->grads) return fail("cf_reduce_adamw_part: no gradient buffer bound");
    if (B < 1 || B > h->cfg.max_batch) return fail("cf_reduce_adamw_part: bad batch size %d", B);
    if (reduce_buckets != CF_BUCKET_PE && reduce_buckets != CF_BUCKET_REG) return fail("cf_reduce_adamw_part: exactly one reduction bucket");
    if (adam_buckets & reduce_buckets) return fail("cf_reduce_adamw_part: the optimiser range overlaps the bucket under reduction");
    if (adam_buckets & CF_BUCKET_PE) h->tiled_pe_fresh = false;
    AdamHyper hy;
    long long lo, n4;
    int agrid;
    if (adam_hyper(h, lr, beta1, beta2, eps, weight_decay, step, hy) || adam_range(h, adam_buckets, lo, n4, agrid)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const long long launches0 = g_launches;
    if ((reduce_buckets & CF_BUCKET_PE) && !h->trunk) {
        hipLaunchKernelGGL(k_wgrad_lp, dim3((B + kLpGenes - 1) / kLpGenes, h->n_lp), dim3(256), 0, st, (const LpJob*)h->lp_jobs, B, h->cfg.d_emb);
        LAUNCH_CHECK("k_wgrad_lp");
    }
    const bool reg = reduce_buckets == CF_BUCKET_REG;
    const int w0 = reg ? 0 : h->n_wg_r, wn = reg ? h->n_wg_r : h->n_wg - h->n_wg_r;
    const int c0 = reg ? 0 : h->n_cs_r, cn = reg ? h->n_cs_r : h->n_cs - h->n_cs_r;
    AdamArgs o{h->params + lo, h->m + lo, h->v + lo, h->grads + lo, n4, hy.decay, hy.one_m_b1, hy.b2, hy.one_m_b2, hy.step_size, hy.bc2_sqrt, hy.eps};
    hipLaunchKernelGGL(k_reduce_adamw, dim3(xcd_grid(wn) + cn + agrid), dim3(256), 0, st, (const WgTile*)h->wg_tiles + w0, wn,
                       (const CsTile*)h->cs_tiles + c0, cn, B, h->xcd_reduce, o);
    LAUNCH_CHECK("k_reduce_adamw");
    h->n_bwd += (int)(g_launches - launches0);
    return 0;
}
// The deferred gradients of ONE bucket and the AdamW update of the same bucket in the epilogues of the reduction tiles
// (k_reduce_opt): single-GPU training, where nothing stands between a gradient element and its update.
extern "C" int cf_reduce_opt_part(cf_handle* h, int B, int bucket, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  long long step, int keep_grads, void* stream) {
    if (!h || !h->grads || !h->params || !h->m || !h->v) return fail("cf_reduce_opt_part: params / grads / moments not bound");
    if (B < 1 || B > h->cfg.max_batch) return fail("cf_reduce_opt_part: bad batch size %d", B);
    if (!bucket || (bucket & ~(CF_BUCKET_REG | CF_BUCKET_PE))) return fail("cf_reduce_opt_part: bad bucket mask %d", bucket);
    if (h->embed_dense) return fail("cf_reduce_opt_part: the all-rows Embedding path (embed n_layers > 1) writes its gradients outside the reduction tables; use cf_backward_reduce_part + cf_adamw_step_part");
    AdamHyper hy;
    if (adam_hyper(h, lr, beta1, beta2, eps, weight_decay, step, hy)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const long long launches0 = g_launches;
    if ((bucket & CF_BUCKET_PE) && !h->trunk) {
        hipLaunchKernelGGL(k_wgrad_lp, dim3((B + kLpGenes - 1) / kLpGenes, h->n_lp), dim3(256), 0, st, (const LpJob*)h->lp_jobs, B, h->cfg.d_emb);
        LAUNCH_CHECK("k_wgrad_lp");
    }
    // the tables hold the Regulation + head bucket's tiles first: one bucket is a prefix / suffix, both are everything
    const bool reg = (bucket & CF_BUCKET_REG) != 0, pe = (bucket & CF_BUCKET_PE) != 0;
    int w0 = reg ? 0 : h->n_wg_r, wn = (reg ? h->n_wg_r : 0) + (pe ? h->n_wg - h->n_wg_r : 0);
    const int c0 = reg ? 0 : h->n_cs_r, cn = (reg ? h->n_cs_r : 0) + (pe ? h->n_cs - h->n_cs_r : 0);
    int2 skip = make_int2(0x7fffffff, 0);
    if (h->rider.done) {       // tiles the riders of the last k_trunk_bwd launch have reduced and stepped already: the window behind the short tiles
        if (!reg || h->rider.step != step) return fail("cf_reduce_opt_part: riders were armed for step %lld of the Regulation + head bucket; this call must finish that step", h->rider.step);
        skip = make_int2(h->n_wg_short, h->rider.done);
        wn -= h->rider.done;
        h->rider.done = 0;
    }
    // cf_keep_tiled: the Embedding + Pairwise bucket's tiles write the tiled copies of the weights they step (their next reader is the next
    // forward pass, which then skips the re-tiling of those tensors)
    AdamFuse o{h->params, h->m, h->v, h->grads, hy.decay, hy.one_m_b1, hy.b2, hy.one_m_b2, hy.step_size, hy.bc2_sqrt, hy.eps, keep_grads ? 1 : 0,
               h->keep_tiled && pe ? h->tiled : nullptr};
    if (h->pend_gnext) {      // the next step's batch gather behind the tiles of this launch (cf_gather_batch_next)
        hipLaunchKernelGGL(k_reduce_opt_gather, dim3(xcd_grid(wn) + cn + h->pend_gn.B * h->pend_gn_n), dim3(256), 0, st, (const WgTile*)h->wg_tiles + w0, wn,
                           (const CsTile*)h->cs_tiles + c0, cn, B, h->xcd_reduce_opt, o, h->pend_gn, skip);
        h->adv_next = h->pend_gn.cursor;
        h->pend_gnext = false;
    } else {
        hipLaunchKernelGGL(k_reduce_opt, dim3(xcd_grid(wn) + cn), dim3(256), 0, st, (const WgTile*)h->wg_tiles + w0, wn, (const CsTile*)h->cs_tiles + c0, B,
                           h->xcd_reduce_opt, o, skip);
    }
    LAUNCH_CHECK("k_reduce_opt");
    if (pe) h->tiled_pe_fresh = h->keep_tiled;      // (every tensor of that bucket with a tiled copy has just been rewritten in both forms -- or in one only)
    h->n_bwd += (int)(g_launches - launches0);
    return 0;
}
// Arms the riders of the NEXT cf_backward_part(parts & 4) call (k_trunk_bwd; cf_trunk.h): up to max_tiles leading weight-gradient
// tiles of the Regulation + head bucket are reduced, with this step's AdamW update in their epilogues, on the CUs the trunk leaves
// idle.  The cf_reduce_opt_part call of the same step (a mask that contains CF_BUCKET_REG) then skips them.
extern "C" int cf_rider_arm(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay, long long step, int keep_grads,
                            int max_tiles) {
    if (!h || !h->grads || !h->params || !h->m || !h->v) return fail("cf_rider_arm: params / grads / moments not bound");
    if (!h->trunk) return fail("cf_rider_arm: the fused trunk kernels are not in use for this configuration");
    if (h->embed_dense) return fail("cf_rider_arm: not with the all-rows Embedding path");
    if (h->rider.done) return fail("cf_rider_arm: the riders of step %lld have not been followed by cf_reduce_opt_part", h->rider.step);
    AdamHyper hy;
    if (adam_hyper(h, lr, beta1, beta2, eps, weight_decay, step, hy)) return -1;
    h->rider.o = AdamFuse{h->params, h->m, h->v, h->grads, hy.decay, hy.one_m_b1, hy.b2, hy.one_m_b2, hy.step_size, hy.bc2_sqrt, hy.eps, keep_grads ? 1 : 0, nullptr};
    h->rider.max_tiles = std::max(0, max_tiles);
    h->rider.step = step;
    h->rider.armed = max_tiles > 0;      // (max_tiles <= 0: disarms; the checks above tell a caller whether riders are available at all)
    return 0;
}
// Split form for callers that replay the optimiser launch from a hipGraph: cf_adamw_set (eager, once per step, before the
// replay) writes the step's scalars to device memory, cf_adamw_step_dev (capturable) reads them.
extern "C" int cf_adamw_set(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay, long long step, void* stream) {
    AdamHyper hy;
    if (adam_hyper(h, lr, beta1, beta2, eps, weight_decay, step, hy)) return -1;
    if (h->capturing) return fail("cf_adamw_set must not be captured: its arguments change every step");
    hipLaunchKernelGGL(k_adamw_set, dim3(1), dim3(1), 0, (hipStream_t)stream, h->hyper, hy);
    LAUNCH_CHECK("k_adamw_set");
    return 0;
}
extern "C" int cf_adamw_step_dev(cf_handle* h, int buckets, void* stream) {
    if (!h || !h->params || !h->grads || !h->m || !h->v) return fail("cf_adamw_step: params / grads / moments not bound");
    long long lo, n4;
    int grid;
    if (adam_range(h, buckets, lo, n4, grid)) return -1;
    if (buckets & CF_BUCKET_PE) h->tiled_pe_fresh = false;
    h->time_mark("k_adamw", (hipStream_t)stream);
    hipLaunchKernelGGL(k_adamw_dev, dim3(grid), dim3(256), 0, (hipStream_t)stream, h->params + lo, (const float*)h->grads + lo, h->m + lo,
                       h->v + lo, n4, (const AdamHyper*)h->hyper);
    h->time_mark("k_adamw", (hipStream_t)stream);
    LAUNCH_CHECK("k_adamw_dev");
    h->n_opt += 1;
    return 0;
}
// `waiter` waits for everything enqueued on `signaller` so far (fork / join of a side stream; under capture this pulls
// the other stream into the graph as a parallel branch).
extern "C" int cf_stream_wait(cf_handle* h, void* waiter, void* signaller) {
    if (!h) return fail("null handle");
    const size_t slot = h->sync_ev_used++ % 64;           // a small ring: an event is re-recorded long after its waiters passed
    if (slot >= h->sync_ev.size()) {
        hipEvent_t ne;
        HIP_TRY(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
        h->sync_ev.push_back(ne);
    }
    hipEvent_t e = h->sync_ev[slot];
    HIP_TRY(hipEventRecord(e, (hipStream_t)signaller));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)waiter, e, 0));
    return 0;
}
// The fused optimiser (cf_reduce_opt_part over the Embedding + Pairwise bucket) can keep the tiled copies of that bucket's weights fresh: its
// epilogue writes every stepped element in both layouts, and forward passes then skip the re-tiling of those tensors -- the launch in front
// of a training step disappears (or carries only the batch gather).  The caller's side of the contract: whoever else writes parameters --
// a checkpoint load, an in-place edit, another optimiser -- says so with cf_params_changed before the next forward pass (the Python layer
// compares the version counter of its flat parameter tensor; the library's own separate AdamW launches and cf_bind clear the flag
// themselves); the next forward pass then re-tiles once, in a launch of its own.  Not offered (-1) where the reduction tiles do not cover those
// tensors (the all-rows Embedding path).
extern "C" int cf_keep_tiled(cf_handle* h, int on) {
    if (!h) return fail("null handle");
    if (on && !h->keep_tiled_ok) return fail("cf_keep_tiled: not available for this configuration (gradient buffer not bound, or the all-rows Embedding path)");
    h->keep_tiled = on != 0;
    h->tiled_pe_fresh = false;
    return 0;
}
extern "C" int cf_params_changed(cf_handle* h) {
    if (!h) return fail("null handle");
    h->tiled_pe_fresh = false;
    return 0;
}
// the re-tiling a forward pass would do on finding the flag cleared, now (before a hipGraph that was captured without it is replayed)
extern "C" int cf_retile_early(cf_handle* h, void* stream) {
    if (!h || !h->params) return fail("cf_retile_early: no parameters bound");
    return retile_early(h, (hipStream_t)stream);
}
extern "C" int cf_adamw_step(cf_handle* h, float lr, float beta1, float beta2, float eps, float weight_decay, long long step,
                             void* stream) {
    return cf_adamw_step_part(h, lr, beta1, beta2, eps, weight_decay, step, CF_BUCKET_REG | CF_BUCKET_PE, stream);
}

// ------------------------------------------------------------------------------------
// introspection
// ------------------------------------------------------------------------------------
extern "C" int cf_debug_copy(cf_handle* h, const char* name, float* dst, long long* n_floats, void* stream) {
    if (!h || !name) return fail("cf_debug_copy: null argument");
    auto it = h->ws_index.find(name);
    if (it == h->ws_index.end()) return fail("cf_debug_copy: no workspace buffer named '%s'", name);
    const WsEntry& e = h->ws[it->second];
    if (n_floats) *n_floats = (long long)e.n;
    if (dst) HIP_TRY(hipMemcpyAsync(dst, h->arena + e.off, e.n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
extern "C" const char* cf_debug_names(cf_handle* h) { return h ? h->ws_names.c_str() : ""; }
extern "C" int cf_launch_counts(cf_handle* h, int* fwd, int* bwd, int* opt) {
    if (!h) return fail("null handle");
    if (fwd) *fwd = h->n_fwd;
    if (bwd) *bwd = h->n_bwd;
    if (opt) *opt = h->n_opt;
    return 0;
}

// ------------------------------------------------------------------------------------
// resident split: batch gather / step log inside the graph
// ------------------------------------------------------------------------------------
static int gather_args(cf_handle* h, const cf_store* st_, const int* order, int* cursor, const cf_batch* dst, void* labels_dst, GatherArgs& ga, int& n) {
    if (!h || !st_ || !order || !cursor || !dst) return fail("cf_gather_batch: null argument");
    const cf_config& c = h->cfg;
    const int B = dst->B, S = c.i_max, T = S + 1, F = c.n_feats;
    if (B < 1 || B > c.max_batch) return fail("cf_gather_batch: B = %d outside [1, max_batch = %d]", B, c.max_batch);
    memset(&ga, 0, sizeof ga);
    n = 0;
    bool overflow = false;
    auto push = [&](const void* src, const void* d, long long gene_bytes) {
        const int chunk = kGatherChunk;
        for (long long off = 0; off < gene_bytes; off += chunk) {
            if (n >= kGatherMaxSeg) {
                overflow = true;
                return;
            }
            ga.seg[n++] = GatherSeg{(const char*)src, (char*)const_cast<void*>(d), (int)gene_bytes, (int)off, (int)std::min<long long>(chunk, gene_bytes - off), 0};
        }
    };
    for (int r = 0; r < c.n_res; ++r) {
        const long long L = c.n_bins[r];
        if (dst->promoter_mask_stride[r] != L || dst->pcre_mask_stride[r] != L)
            return fail("cf_gather_batch: the destination batch must use compact mask rows (stride = n_bins)");
        push(st_->promoter_feats[r], dst->promoter_feats[r], L * F * 4);
        push(st_->pcre_feats[r], dst->pcre_feats[r], (long long)S * L * F * 4);
        push(st_->promoter_mask[r], dst->promoter_mask_row[r], L);
        push(st_->pcre_mask[r], dst->pcre_mask_row[r], (long long)S * L);
        push(st_->interaction_mask, dst->interaction_mask[r], (long long)T * T);
    }
    push(st_->interaction_freq, dst->interaction_freq, (long long)T * T * 4);
    if (labels_dst) push(st_->labels, labels_dst, c.n_out == 1 ? 4 : 8);
    if (overflow) return fail("cf_gather_batch: segment table overflow");
    ga.order = order;
    ga.cursor = cursor;
    ga.n_genes = st_->n_genes;
    ga.B = B;
    return 0;
}
static int gather_launch(const GatherArgs& ga, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_batch, dim3(ga.B, n), dim3(256), 0, st, ga);
    LAUNCH_CHECK("k_gather_batch");
    hipLaunchKernelGGL(k_gather_advance, dim3(1), dim3(1), 0, st, ga.cursor);
    LAUNCH_CHECK("k_gather_advance");
    return 0;
}
extern "C" int cf_gather_batch(cf_handle* h, const cf_store* st_, const int* order, int* cursor, const cf_batch* dst, void* labels_dst,
                               void* stream) {
    GatherArgs ga;
    int n;
    if (gather_args(h, st_, order, cursor, dst, labels_dst, ga, n)) return -1;
    return gather_launch(ga, n, (hipStream_t)stream);
}
// cf_gather_batch for the batch of a TRAINING step: nothing is launched here; the cf_forward / cf_forward_train that must follow on
// the same stream (same batch buffers) copies the genes in the launch that refreshes its tiled weight copies -- the two do not depend
// on each other -- and advances the cursor.  Three launches in front of every step of the training loop become one.
extern "C" int cf_gather_batch_fwd(cf_handle* h, const cf_store* st_, const int* order, int* cursor, const cf_batch* dst, void* labels_dst,
                                   void* stream) {
    (void)stream;
    int n;
    if (gather_args(h, st_, order, cursor, dst, labels_dst, h->pend_ga, n)) return -1;
    h->pend_ga_n = n;
    h->pend_gather = true;
    h->pend_key = dst->promoter_feats[0];
    return 0;
}

// The batch of the NEXT step, gathered while this step ends: nothing is launched here; the cf_reduce_opt_part that follows on the same stream
// carries the copy blocks behind its tiles (nothing in that launch reads the batch buffers, and every kernel of this step that does has
// finished), and the cursor -- which the gather reads, so it cannot move in the same launch -- is advanced by the trunk's forward launch
// of the next cf_forward / cf_forward_train.  A step of the training loop then has no launch in front of it.  cf_gather_batch_only is the
// same for the first step of an epoch: the copy as a launch of its own, now, the cursor left to the forward pass that follows.
extern "C" int cf_gather_batch_next(cf_handle* h, const cf_store* st_, const int* order, int* cursor, const cf_batch* dst, void* labels_dst,
                                    void* stream) {
    (void)stream;
    int n;
    if (gather_args(h, st_, order, cursor, dst, labels_dst, h->pend_gn, n)) return -1;
    h->pend_gn_n = n;
    h->pend_gnext = true;
    h->pend_key = dst->promoter_feats[0];
    return 0;
}
extern "C" int cf_gather_batch_only(cf_handle* h, const cf_store* st_, const int* order, int* cursor, const cf_batch* dst, void* labels_dst,
                                    void* stream) {
    GatherArgs ga;
    int n;
    if (gather_args(h, st_, order, cursor, dst, labels_dst, ga, n)) return -1;
    hipLaunchKernelGGL(k_gather_batch, dim3(ga.B, n), dim3(256), 0, (hipStream_t)stream, ga);
    LAUNCH_CHECK("k_gather_batch");
    h->adv_next = cursor;
    h->pend_key = dst->promoter_feats[0];
    return 0;
}

extern "C" int cf_record_step(cf_handle* h, const int* cursor, const float* logits, const void* labels, const float* loss, int B,
                              float* logits_log, void* labels_log, float* loss_log, void* stream) {
    if (!h || !cursor || !logits || !labels || !loss || !logits_log || !labels_log || !loss_log) return fail("cf_record_step: null argument");
    RecordArgs ra{cursor, logits, (const char*)labels, loss, logits_log, (char*)labels_log, loss_log, B, h->cfg.n_out, h->cfg.n_out == 1 ? 4 : 8};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_record_step, dim3(1), dim3(256), 0, st, ra);
    LAUNCH_CHECK("k_record_step");
    return 0;
}
// cf_record_step without a launch of its own: the cf_backward_part(parts & 4) that must follow on the same stream writes the log rows at the
// start of the trunk's backward launch (the loss is final by then).  Configurations without the fused trunk kernels: that call issues
// k_record_step itself.
extern "C" int cf_record_step_bwd(cf_handle* h, const int* cursor, const float* logits, const void* labels, const float* loss, int B,
                                  float* logits_log, void* labels_log, float* loss_log, void* stream) {
    (void)stream;
    if (!h || !cursor || !logits || !labels || !loss || !logits_log || !labels_log || !loss_log) return fail("cf_record_step_bwd: null argument");
    h->pend_rec = RecordArgs{cursor, logits, (const char*)labels, loss, logits_log, (char*)labels_log, loss_log, B, h->cfg.n_out, h->cfg.n_out == 1 ? 4 : 8};
    h->pend_record = true;
    return 0;
}

// ------------------------------------------------------------------------------------
// standalone operators
// ------------------------------------------------------------------------------------
extern "C" int cf_op_linear(const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int relu, void* stream) {
    if (K % 128 || N % 32) return fail("cf_op_linear: K %% 128 == 0 and N %% 32 == 0 required");
    // standalone use: build the tiled copy of W on the fly (test / micro-benchmark helper, synchronous)
    std::vector<RetileUnit> units;
    for (int n0 = 0; n0 < N; n0 += 16) units.push_back(RetileUnit{(long long)n0 * K, 0, K, N, n0, 0});
    float* Wt = nullptr;
    RetileUnit* du = nullptr;
    HIP_TRY(hipMalloc(&Wt, (size_t)N * K * sizeof(float)));
    HIP_TRY(hipMalloc(&du, units.size() * sizeof(RetileUnit)));
    HIP_TRY(hipMemcpy(du, units.data(), units.size() * sizeof(RetileUnit), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_retile, dim3((int)units.size()), dim3(256), 0, (hipStream_t)stream, W, Wt, (float*)nullptr, (const RetileUnit*)du);
    LinArgs a;
    memset(&a, 0, sizeof a);
    a.x[0] = A;
    a.w[0] = Wt;
    a.b[0] = bias;
    a.y[0] = C;
    a.xmap = identity_map();
    a.ldx = K;
    a.ldy = N;
    a.N = M;
    a.K = K;
    a.Nout = N;
    a.relu = relu;
    hipLaunchKernelGGL((k_linear_fwd<2>), dim3(tiles_of(M), (N + 127) / 128, 1), dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK("cf_op_linear");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipFree(Wt));
    HIP_TRY(hipFree(du));
    return 0;
}
extern "C" int cf_op_dgrad(const float* dY, const float* W, float* dX, int M, int N, int K, void* stream) {
    if ((N != 128 && N != 256 && N != 1024) || K % 32) return fail("cf_op_dgrad: N in {128, 256, 1024} and K %% 32 == 0 required");
    DgradArgs a;
    memset(&a, 0, sizeof a);
    a.dy[0] = dY;
    a.lddy = N;
    a.w[0] = W;
    a.ldw = K;
    a.rmap = identity_map();
    a.dx[0] = dX;
    a.lddx = K;
    a.N = M;
    a.K = N;
    a.Ncols = K;
    const dim3 grid(tiles_of(M), K / 32, 1);
    if (N == 128) hipLaunchKernelGGL((k_dgrad<2>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (N == 256) hipLaunchKernelGGL((k_dgrad<4>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_dgrad<16>), grid, dim3(256), 0, (hipStream_t)stream, a);
    LAUNCH_CHECK("cf_op_dgrad");
    return 0;
}
extern "C" int cf_op_wgrad(const float* dY, const float* X, float* dW, int M, int N, int K, void* stream) {
    if (K % 4) return fail("cf_op_wgrad: K %% 4 == 0 required");
    std::vector<WgTile> tiles;
    push_wg(tiles, wg1(dY, N, X, K, M, dW, K, N, K));
    WgTile* d = nullptr;
    HIP_TRY(hipMalloc(&d, tiles.size() * sizeof(WgTile)));
    HIP_TRY(hipMemcpy(d, tiles.data(), tiles.size() * sizeof(WgTile), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_wgrad, dim3(xcd_grid((int)tiles.size())), dim3(256), 0, (hipStream_t)stream, (const WgTile*)d, (int)tiles.size(), 1, 0);
    LAUNCH_CHECK("cf_op_wgrad");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipFree(d));
    return 0;
}

static int attn_args(const cf_attn_shape* sh, AttnArgs& a) {
    if (!sh) return fail("cf_op_attention: null shape");
    if (sh->N < 1 || sh->H < 1 || sh->Lq < 1 || sh->Lk < 1) return fail("cf_op_attention: bad shape");
    if (sh->N > 65535 || sh->H > 65535) return fail("cf_op_attention: N and H are grid dimensions (<= 65535)");
    if ((sh->ldq | sh->ldk | sh->ldv | sh->ldo) & 3) return fail("cf_op_attention: row strides must be multiples of 4 floats");
    if (sh->ldq < sh->H * kADh || sh->ldk < sh->H * kADh || sh->ldv < sh->H * kADh || sh->ldo < sh->H * kADh)
        return fail("cf_op_attention: row stride smaller than H * 64");
    memset(&a, 0, sizeof a);
    a.N = sh->N;
    a.H = sh->H;
    a.Lq = sh->Lq;
    a.Lk = sh->Lk;
    a.ldq = sh->ldq;
    a.ldk = sh->ldk;
    a.ldv = sh->ldv;
    a.ldo = sh->ldo;
    a.rscale = 1.0f / sqrtf((float)kADh);
    return 0;
}
// The dense attention forward: k_attn_fwd (round 5: transposed score tiles, 128 query rows per workgroup); CF_ATTN_FWD_V1=1 runs the round-1
// kernel (64 rows per workgroup, P through a per-wave LDS patch) -- the cross-check of the tests.  Same results up to the order of the fp32
// additions inside P V.
static int attn_fwd_launch(const AttnArgs& a, hipStream_t st) {
    if (getenv_int("CF_ATTN_FWD_V1", 0)) {
        hipLaunchKernelGGL(k_attn_fwd_v1, dim3((a.Lq + kABq - 1) / kABq, a.H, a.N), dim3(256), 0, st, a);
        LAUNCH_CHECK("k_attn_fwd_v1");
        return 0;
    }
    const dim3 grid((a.Lq + kABq2 - 1) / kABq2, a.H, a.N);
    if (a.mask) hipLaunchKernelGGL(k_attn_fwd<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_attn_fwd<false>, grid, dim3(256), 0, st, a);
    LAUNCH_CHECK("k_attn_fwd");
    return 0;
}
extern "C" int cf_op_attention_fwd(const cf_attn_shape* sh, const float* q, const float* k, const float* v, const unsigned char* qvalid,
                                   const unsigned char* kvalid, const unsigned char* mask, float* o, float* stats, void* stream) {
    AttnArgs a;
    if (attn_args(sh, a)) return -1;
    if (!q || !k || !v || !o) return fail("cf_op_attention_fwd: null tensor");
    a.q = q;
    a.k = k;
    a.v = v;
    a.qvalid = qvalid;
    a.kvalid = kvalid;
    a.mask = mask;
    a.o = o;
    a.stats = stats;
    if (attn_fwd_launch(a, (hipStream_t)stream)) return -1;
    return 0;
}
// dQ, dK, dV of the dense attention core (delta = rowsum(dO * O) is in a.delta already).  One fused pass per (sequence, head)
// (k_attn_bwd: 5 tile products per key / query tile pair) when the launch has enough (sequence, head) workgroups to fill the chip;
// otherwise the two kernels split by output owner (7 products per pair, but (Lk / 64 + Lq / 64) workgroups per sequence and head).
// CF_ATTN_BWD_SPLIT=1 forces the split kernels, -1 the fused one (A/B runs, cross-checks in the tests; read at every call).  Same
// results either way up to the order of the fp32 additions inside dQ.
static int attn_bwd_launch(const AttnArgs& a, hipStream_t st) {
    const int mode = getenv_int("CF_ATTN_BWD_SPLIT", 0);
    const bool vec_ok = (a.ldq & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dq) & 15) == 0;      // (the dQ update is 16 bytes per lane)
    if (vec_ok && (mode < 0 || (mode == 0 && (long long)a.N * a.H >= 512))) {
        // round 6: 128 keys per pass, eight waves, 141 KB of dynamic LDS (k_attn_bwd2); CF_ATTN_BWD_V1=1 runs the 64-key kernel of rounds 4-5 (the
        // cross-check of the tests: same results up to the order of the fp32 additions inside dQ)
        const bool v2_ok = ((reinterpret_cast<uintptr_t>(a.dk) | reinterpret_cast<uintptr_t>(a.dv)) & 15) == 0 && getenv_int("CF_ATTN_BWD_V1", 0) == 0;
        if (v2_ok) {
            static int ready = 0;      // 0: not tried, 1: attribute set, -1: refused by the runtime (fall through to the 64-key kernel)
            if (ready == 0) {
                hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAB2Smem);
                hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAB2Smem);
                ready = (e1 == hipSuccess && e2 == hipSuccess) ? 1 : -1;
                if (ready < 0) (void)hipGetLastError();
            }
            if (ready > 0) {
                if (a.mask) hipLaunchKernelGGL(k_attn_bwd2<true>, dim3(a.H, a.N), dim3(512), kAB2Smem, st, a);
                else hipLaunchKernelGGL(k_attn_bwd2<false>, dim3(a.H, a.N), dim3(512), kAB2Smem, st, a);
                LAUNCH_CHECK("k_attn_bwd2");
                return 0;
            }
        }
        hipLaunchKernelGGL(k_attn_bwd, dim3(a.H, a.N), dim3(256), 0, st, a);
        LAUNCH_CHECK("k_attn_bwd");
        return 0;
    }
    hipLaunchKernelGGL(k_attn_bwd_kv, dim3((a.Lk + kABk - 1) / kABk, a.H, a.N), dim3(256), 0, st, a);
    LAUNCH_CHECK("k_attn_bwd_kv");
    hipLaunchKernelGGL(k_attn_bwd_q, dim3((a.Lq + kABq - 1) / kABq, a.H, a.N), dim3(256), 0, st, a);
    LAUNCH_CHECK("k_attn_bwd_q");
    return 0;
}
extern "C" int cf_op_attention_bwd(const cf_attn_shape* sh, const float* q, const float* k, const float* v, const unsigned char* qvalid,
                                   const unsigned char* kvalid, const unsigned char* mask, const float* o, const float* stats,
                                   const float* d_o, float* dq, float* dk, float* dv, float* delta_ws, void* stream) {
    AttnArgs a;
    if (attn_args(sh, a)) return -1;
    if (!q || !k || !v || !o || !stats || !d_o || !dq || !dk || !dv || !delta_ws) return fail("cf_op_attention_bwd: null tensor");
    a.q = q;
    a.k = k;
    a.v = v;
    a.qvalid = qvalid;
    a.kvalid = kvalid;
    a.mask = mask;
    a.o = const_cast<float*>(o);
    a.stats = const_cast<float*>(stats);
    a.d_o = d_o;
    a.dq = dq;
    a.dk = dk;
    a.dv = dv;
    a.delta = delta_ws;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_attn_delta, dim3((a.Lq + 15) / 16, a.H, a.N), dim3(256), 0, st, a);
    LAUNCH_CHECK("k_attn_delta");
    if (attn_bwd_launch(a, st)) return -1;
    return 0;
}

static_assert(sizeof(cf_bin_job) == sizeof(BinJob), "cf_bin_job layout");
extern "C" int cf_bin_regions(const cf_bin_job* jobs, int n_jobs, int n_feats, int bin_size, int n_bins_out, void* stream) {
    if (!jobs) return fail("cf_bin_regions: null job table");
    if (n_jobs < 0 || n_feats < 1 || n_feats > 64 || bin_size < 1 || n_bins_out < 1) return fail("cf_bin_regions: bad argument");
    if (n_jobs == 0) return 0;
    hipLaunchKernelGGL(k_bin_regions, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const BinJob*>(jobs), n_feats,
                       bin_size, n_bins_out);
    LAUNCH_CHECK("k_bin_regions");
    return 0;
}

static_assert(sizeof(cf_bin_job_multi) == sizeof(BinJobMulti), "cf_bin_job_multi layout");
extern "C" int cf_bin_regions_multi(const cf_bin_job_multi* jobs, int n_jobs, int n_feats, int n_res, const int* bin_sizes, const int* n_bins_out,
                                    int max_cols, void* stream) {
    if (!jobs || !bin_sizes || !n_bins_out) return fail("cf_bin_regions_multi: null argument");
    if (n_jobs < 0 || n_feats < 1 || n_feats > 64 || n_res < 1 || n_res > kBinMaxRes || max_cols < 0) return fail("cf_bin_regions_multi: bad argument");
    if (n_jobs == 0) return 0;
    BinPlan pl;
    memset(&pl, 0, sizeof pl);
    pl.n_res = n_res;
    pl.F = n_feats;
    for (int r = 0; r < n_res; ++r) {
        if (bin_sizes[r] < 1 || n_bins_out[r] < 1 || n_bins_out[r] > kBinMaxBins) return fail("cf_bin_regions_multi: bad bin size / bin count at resolution %d", r);
        if (r && bin_sizes[r] >= bin_sizes[r - 1]) return fail("cf_bin_regions_multi: bin sizes must be listed coarsest first (%d after %d)", bin_sizes[r], bin_sizes[r - 1]);
        pl.b[r] = bin_sizes[r];
        pl.L[r] = n_bins_out[r];
    }
    // one pass over the raw bytes when the bins nest and a unit (one coarsest bin) fits a wave's registers and lanes
    bool nested = n_res >= 2 && n_feats <= kBinMaxF && (pl.b[n_res - 1] & 3) == 0 && pl.b[0] <= kBinMaxLoads * 256 && pl.b[0] / pl.b[n_res - 1] <= 64;
    for (int r = 0; r + 1 < n_res; ++r) nested = nested && pl.b[r] % pl.b[r + 1] == 0;
    pl.nested = nested ? 1 : 0;
    const int units = nested ? std::max(1, (max_cols + pl.b[0] - 1) / pl.b[0]) : 1;
    const dim3 grid((units + 3) / 4, n_jobs);
    const int nload = nested ? (pl.b[0] / 4 + 63) / 64 : 4;
    const BinJobMulti* jm = reinterpret_cast<const BinJobMulti*>(jobs);
    if (nload <= 4) hipLaunchKernelGGL(k_bin_multi<4>, grid, dim3(256), 0, (hipStream_t)stream, jm, pl);
    else if (nload <= 8) hipLaunchKernelGGL(k_bin_multi<8>, grid, dim3(256), 0, (hipStream_t)stream, jm, pl);
    else hipLaunchKernelGGL(k_bin_multi<16>, grid, dim3(256), 0, (hipStream_t)stream, jm, pl);
    LAUNCH_CHECK("k_bin_multi");
    return 0;
}

// ------------------------------------------------------------------------------------
// dense (all rows) layer: projections -> attention core -> out-projection / LN / FFN / LN chain, and its backward
// ------------------------------------------------------------------------------------
namespace {
struct DenseWs {       // workspace layout in floats; `train` adds what the backward pass needs
    long long wq_t, wkv_t, wo_t, w1_t, w2_t, q, kv, o, tab;                                  // forward
    long long stats, xh1, rs1, y1, hdn, xh2, rs2;                                             // saved
    long long dt2, dpre1, dt1, da, dq, dkv, delta, partial, wpart;                           // backward
    long long total;
    int splits;
};
constexpr int kDenseTab = 16384;        // floats reserved for the unit / tile tables
constexpr int kDenseSplitRows = 4096;   // reduction rows per split-K chunk of the weight gradients
DenseWs dense_ws(int N, int Lq, int Lk, int dff, bool train) {
    const long long rq = (long long)N * Lq, rk = (long long)N * Lk;
    DenseWs w;
    long long o = 0;
    auto take = [&](long long n) {
        const long long at = o;
        o += (n + 3) / 4 * 4;
        return at;
    };
    w.wq_t = take(128 * 128);
    w.wkv_t = take(256 * 128);
    w.wo_t = take(128 * 128);
    w.w1_t = take((long long)dff * 128);
    w.w2_t = take((long long)128 * dff);
    w.q = take(rq * 128);
    w.kv = take(rk * 256);
    w.o = take(rq * 128);
    w.tab = take(kDenseTab);
    w.splits = 0;
    if (train) {
        const long long tiles = (rq + kTile - 1) / kTile;
        w.stats = take((long long)N * 2 * Lq * 2);
        w.xh1 = take(rq * 128);
        w.rs1 = take(rq);
        w.y1 = take(rq * 128);
        w.hdn = take(rq * dff);
        w.xh2 = take(rq * 128);
        w.rs2 = take(rq);
        w.dt2 = take(rq * 128);
        w.dpre1 = take(rq * dff);
        w.dt1 = take(rq * 128);
        w.da = take(rq * 128);
        w.dq = take(rq * 128);
        w.dkv = take(rk * 256);
        w.delta = take((long long)N * 2 * Lq);
        w.partial = take(tiles * post_partial_width(dff));
        w.splits = (int)((std::max(rq, rk) + kDenseSplitRows - 1) / kDenseSplitRows);
        w.wpart = take((long long)w.splits * 256 * 128);        // the largest weight is 256 x 128 (or 128 x 256)
    }
    w.total = o;
    return w;
}
}  // namespace

extern "C" long long cf_op_dense_layer_workspace(int N, int Lq, int Lk, int d_ff) { return dense_ws(N, Lq, Lk, d_ff, false).total; }
extern "C" long long cf_op_dense_layer_train_workspace(int N, int Lq, int Lk, int d_ff) { return dense_ws(N, Lq, Lk, d_ff, true).total; }

static int dense_layer_fwd(const cf_dense_layer* w, const float* x_q, const float* x_kv, const unsigned char* qvalid, const unsigned char* kvalid,
                           const unsigned char* mask, int N, int Lq, int Lk, float* y, float* ws, bool train, hipStream_t st) {
    if (!w || !x_q || !x_kv || !y || !ws) return fail("cf_op_dense_layer_fwd: null argument");
    if (w->d_ff != 128 && w->d_ff != 256) return fail("cf_op_dense_layer_fwd: d_ff must be 128 or 256");
    if (N < 1 || Lq < 1 || Lk < 1 || N > 65535) return fail("cf_op_dense_layer_fwd: bad shape");
    const int dff = w->d_ff;
    const long long rq = (long long)N * Lq, rk = (long long)N * Lk;
    if (rq > 0x7fffffffLL / 256 || rk > 0x7fffffffLL / 256) return fail("cf_op_dense_layer_fwd: too many rows");
    const DenseWs L = dense_ws(N, Lq, Lk, dff, train);
    // tiled copies of the five weights (no table: workgroup b of a launch takes rows 16 b .. of its matrix)
    struct Job { const float* src; float* dst; int rows, K; } jobs[5] = {{w->wq, ws + L.wq_t, 128, 128}, {w->wkv, ws + L.wkv_t, 256, 128},
                                                                         {w->wo, ws + L.wo_t, 128, 128}, {w->w1, ws + L.w1_t, dff, 128},
                                                                         {w->w2, ws + L.w2_t, 128, dff}};
    for (int j = 0; j < 5; ++j) {
        hipLaunchKernelGGL(k_retile_rows, dim3(jobs[j].rows / 16), dim3(256), 0, st, jobs[j].src, jobs[j].dst, jobs[j].K);
        LAUNCH_CHECK("k_retile_rows<dense layer>");
    }
    auto linear = [&](const float* x, const float* wt, float* out, long long rows, int nout) {
        LinArgs a;
        memset(&a, 0, sizeof a);
        a.x[0] = x;
        a.w[0] = wt;
        a.y[0] = out;
        a.xmap = identity_map();
        a.ldx = 128;
        a.ldy = nout;
        a.N = (int)rows;
        a.K = 128;
        a.Nout = nout;
        hipLaunchKernelGGL((k_linear_fwd<2>), dim3(tiles_of((int)rows), (nout + 127) / 128, 1), dim3(256), 0, st, a);
    };
    linear(x_q, ws + L.wq_t, ws + L.q, rq, 128);
    LAUNCH_CHECK("k_linear_fwd<q>");
    linear(x_kv, ws + L.wkv_t, ws + L.kv, rk, 256);
    LAUNCH_CHECK("k_linear_fwd<kv>");
    {
        AttnArgs a;
        cf_attn_shape sh = {N, 2, Lq, Lk, 128, 256, 256, 128};
        if (attn_args(&sh, a)) return -1;
        a.q = ws + L.q;
        a.k = ws + L.kv;
        a.v = ws + L.kv + 128;
        a.qvalid = qvalid;
        a.kvalid = kvalid;
        a.mask = mask;
        a.o = ws + L.o;
        a.stats = train ? ws + L.stats : nullptr;
        if (attn_fwd_launch(a, st)) return -1;
    }
    {
        PostArgs p;
        memset(&p, 0, sizeof p);
        p.x[0] = x_q;
        p.xmap = identity_map();
        p.ain[0] = ws + L.o;
        p.wo[0] = ws + L.wo_t;
        p.bo[0] = w->bo;
        p.g1[0] = w->ln1_g;
        p.be1[0] = w->ln1_b;
        p.w1[0] = ws + L.w1_t;
        p.b1[0] = w->b1;
        p.w2[0] = ws + L.w2_t;
        p.b2[0] = w->b2;
        p.g2[0] = w->ln2_g;
        p.be2[0] = w->ln2_b;
        if (train) {
            p.xh1[0] = ws + L.xh1;
            p.rs1[0] = ws + L.rs1;
            p.y1[0] = ws + L.y1;
            p.hdn[0] = ws + L.hdn;
            p.xh2[0] = ws + L.xh2;
            p.rs2[0] = ws + L.rs2;
        }
        p.out[0] = y;
        p.omap = identity_map();
        p.N = (int)rq;
        p.save = train ? 1 : 0;
        launch_post_fwd<false, 128>(dff, dim3(tiles_of((int)rq), 1), st, p);
        LAUNCH_CHECK("k_post_fwd<dense layer>");
    }
    return 0;
}
extern "C" int cf_op_dense_layer_fwd(const cf_dense_layer* w, const float* x_q, const float* x_kv, const unsigned char* qvalid,
                                     const unsigned char* kvalid, const unsigned char* mask, int N, int Lq, int Lk, float* y, float* ws,
                                     void* stream) {
    return dense_layer_fwd(w, x_q, x_kv, qvalid, kvalid, mask, N, Lq, Lk, y, ws, false, (hipStream_t)stream);
}
extern "C" int cf_op_dense_layer_fwd_train(const cf_dense_layer* w, const float* x_q, const float* x_kv, const unsigned char* qvalid,
                                           const unsigned char* kvalid, const unsigned char* mask, int N, int Lq, int Lk, float* y, float* ws,
                                           void* stream) {
    return dense_layer_fwd(w, x_q, x_kv, qvalid, kvalid, mask, N, Lq, Lk, y, ws, true, (hipStream_t)stream);
}

// dW[N_, K_] = dY^T X over `rows` rows: split-K over chunks of kDenseSplitRows rows (one k_wgrad tile per (n0, k0, chunk), partial
// results in `part`), then a column sum over the chunks.  Fixed order: deterministic.
static int dense_wgrad(const float* dY, int lddy, const float* X, int ldx, long long rows, float* dW, int N_, int K_, float* part,
                       WgTile* tiles_d, CsTile* cs_d, hipStream_t st) {
    const int splits = (int)((rows + kDenseSplitRows - 1) / kDenseSplitRows);
    const int ntiles = ((N_ + 63) / 64) * ((K_ + kWgTk - 1) / kWgTk) * splits, ncs = (N_ * K_ + 63) / 64;
    if ((size_t)ntiles * sizeof(WgTile) > (size_t)(kDenseTab / 2) * sizeof(float) * 64 || (size_t)ncs * sizeof(CsTile) > (size_t)(kDenseTab / 2) * sizeof(float) * 64)
        return fail("cf_op_dense_layer_bwd: tile table overflow");
    DenseWgTab tb{dY, X, part, dW, tiles_d, cs_d, rows, lddy, ldx, N_, K_, splits, kDenseSplitRows};
    hipLaunchKernelGGL(k_dense_wg_tables, dim3(std::max(1, std::min(64, (ntiles + 255) / 256))), dim3(256), 0, st, tb);
    LAUNCH_CHECK("k_dense_wg_tables");
    hipLaunchKernelGGL(k_wgrad, dim3(xcd_grid(ntiles)), dim3(256), 0, st, (const WgTile*)tiles_d, ntiles, 1, 0);
    LAUNCH_CHECK("k_wgrad<dense layer>");
    hipLaunchKernelGGL(k_colsum, dim3(ncs), dim3(256), 0, st, (const CsTile*)cs_d, 1);
    LAUNCH_CHECK("k_colsum<dense layer>");
    return 0;
}

extern "C" int cf_op_dense_layer_bwd(const cf_dense_layer* w, const float* x_q, const float* x_kv, const unsigned char* qvalid,
                                     const unsigned char* kvalid, const unsigned char* mask, int N, int Lq, int Lk, const float* dy,
                                     float* dx_q, float* dx_kv, const cf_dense_layer_grads* g, float* ws, float* tables, void* stream) {
    if (!w || !x_q || !x_kv || !dy || !dx_q || !dx_kv || !g || !ws || !tables) return fail("cf_op_dense_layer_bwd: null argument");
    if (w->d_ff != 128 && w->d_ff != 256) return fail("cf_op_dense_layer_bwd: d_ff must be 128 or 256");
    hipStream_t st = (hipStream_t)stream;
    const int dff = w->d_ff;
    const long long rq = (long long)N * Lq, rk = (long long)N * Lk;
    const DenseWs L = dense_ws(N, Lq, Lk, dff, true);
    const int tiles_q = tiles_of((int)rq);
    {   // out-projection / LN / FFN / LN chain
        PostBwdArgs p;
        memset(&p, 0, sizeof p);
        p.dout[0] = dy;
        p.dmap = identity_map();
        p.xh2[0] = ws + L.xh2;
        p.rs2[0] = ws + L.rs2;
        p.g2[0] = w->ln2_g;
        p.hdn[0] = ws + L.hdn;
        p.w2[0] = w->w2;
        p.w1[0] = w->w1;
        p.xh1[0] = ws + L.xh1;
        p.rs1[0] = ws + L.rs1;
        p.g1[0] = w->ln1_g;
        p.wo[0] = w->wo;
        p.dt2[0] = ws + L.dt2;
        p.dpre1[0] = ws + L.dpre1;
        p.dt1[0] = ws + L.dt1;
        p.da[0] = ws + L.da;
        p.partial[0] = ws + L.partial;
        p.N = (int)rq;
        launch_post_bwd<false, 128>(dff, dim3(tiles_q, 1), st, p);
        LAUNCH_CHECK("k_post_bwd<dense layer>");
    }
    {   // attention core
        AttnArgs a;
        cf_attn_shape sh = {N, 2, Lq, Lk, 128, 256, 256, 128};
        if (attn_args(&sh, a)) return -1;
        a.q = ws + L.q;
        a.k = ws + L.kv;
        a.v = ws + L.kv + 128;
        a.qvalid = qvalid;
        a.kvalid = kvalid;
        a.mask = mask;
        a.o = ws + L.o;
        a.stats = ws + L.stats;
        a.d_o = ws + L.da;
        a.dq = ws + L.dq;
        a.dk = ws + L.dkv;
        a.dv = ws + L.dkv + 128;
        a.delta = ws + L.delta;
        hipLaunchKernelGGL(k_attn_delta, dim3((Lq + 15) / 16, 2, N), dim3(256), 0, st, a);
        LAUNCH_CHECK("k_attn_delta");
        if (attn_bwd_launch(a, st)) return -1;
    }
    {   // input gradients: dx_q = dt1 (residual) + dq Wq;  dx_kv = dkv Wkv
        DgradArgs d;
        memset(&d, 0, sizeof d);
        d.dy[0] = ws + L.dq;
        d.lddy = 128;
        d.w[0] = w->wq;
        d.ldw = 128;
        d.res[0] = ws + L.dt1;
        d.rmap = identity_map();
        d.ldres = 128;
        d.dx[0] = dx_q;
        d.lddx = 128;
        d.N = (int)rq;
        d.K = 128;
        d.Ncols = 128;
        hipLaunchKernelGGL((k_dgrad<2>), dim3(tiles_q, 128 / 32, 1), dim3(256), 0, st, d);
        LAUNCH_CHECK("k_dgrad<q>");
        memset(&d, 0, sizeof d);
        d.dy[0] = ws + L.dkv;
        d.lddy = 256;
        d.w[0] = w->wkv;
        d.ldw = 128;
        d.rmap = identity_map();
        d.dx[0] = dx_kv;
        d.lddx = 128;
        d.N = (int)rk;
        d.K = 256;
        d.Ncols = 128;
        hipLaunchKernelGGL((k_dgrad<4>), dim3(tiles_of((int)rk), 128 / 32, 1), dim3(256), 0, st, d);
        LAUNCH_CHECK("k_dgrad<kv>");
    }
    // weight gradients (split-K), bias / LayerNorm gradients (column sums of the per-tile partials)
    WgTile* tiles_d = reinterpret_cast<WgTile*>(tables);
    CsTile* cs_d = reinterpret_cast<CsTile*>(tables + (size_t)(kDenseTab / 2) * 64);
    float* part = ws + L.wpart;
    if (dense_wgrad(ws + L.dq, 128, x_q, 128, rq, g->wq, 128, 128, part, tiles_d, cs_d, st)) return -1;
    if (dense_wgrad(ws + L.dkv, 256, x_kv, 128, rk, g->wkv, 256, 128, part, tiles_d, cs_d, st)) return -1;
    if (dense_wgrad(ws + L.dt1, 128, ws + L.o, 128, rq, g->wo, 128, 128, part, tiles_d, cs_d, st)) return -1;
    if (dense_wgrad(ws + L.dpre1, dff, ws + L.y1, 128, rq, g->w1, dff, 128, part, tiles_d, cs_d, st)) return -1;
    if (dense_wgrad(ws + L.dt2, 128, ws + L.hdn, dff, rq, g->w2, 128, dff, part, tiles_d, cs_d, st)) return -1;
    {
        // two stages: the per-tile partial rows (one per 16 input rows: 100,000 of them at the stress shape) are first summed in
        // chunks of 512 rows by many workgroups, then the chunk sums per quantity -- one workgroup walking 100,000 rows took 11 ms
        const int pw = post_partial_width(dff);
        const float* pp = ws + L.partial;
        constexpr int kChunkRows = 512;
        const int nchunks = (tiles_q + kChunkRows - 1) / kChunkRows;
        float* part2 = ws + L.wpart;                 // free again: the weight gradients above are done with it
        if ((long long)nchunks * pw > (long long)L.splits * 256 * 128) return fail("cf_op_dense_layer_bwd: chunk buffer too small");
        const int per = (pw + 63) / 64, n1 = nchunks * per;
        int n2 = 0;
        DenseCsTab tb;
        tb.partial = pp;
        tb.part2 = part2;
        tb.cs1 = reinterpret_cast<CsTile*>(tiles_d);
        tb.cs2 = cs_d;
        tb.tiles_q = tiles_q;
        tb.chunk_rows = kChunkRows;
        tb.nchunks = nchunks;
        tb.pw = pw;
        const int offs[7] = {0, 128, 256, 384, 384 + dff, 512 + dff, 640 + dff}, ncl[7] = {kD, kD, kD, dff, kD, kD, kD};
        float* dsts[7] = {g->ln2_g, g->ln2_b, g->b2, g->b1, g->ln1_g, g->ln1_b, g->bo};
        for (int k = 0; k < 7; ++k) {
            tb.off[k] = offs[k];
            tb.ncols[k] = ncl[k];
            tb.dst[k] = dsts[k];
            n2 += (ncl[k] + 63) / 64;
        }
        if ((size_t)n1 * sizeof(CsTile) > (size_t)(kDenseTab / 2) * sizeof(float) * 64) return fail("cf_op_dense_layer_bwd: tile table overflow");
        hipLaunchKernelGGL(k_dense_cs_tables, dim3(std::max(1, std::min(64, (n1 + 255) / 256))), dim3(256), 0, st, tb);
        LAUNCH_CHECK("k_dense_cs_tables");
        hipLaunchKernelGGL(k_colsum, dim3(n1), dim3(256), 0, st, (const CsTile*)tb.cs1, 1);
        LAUNCH_CHECK("k_colsum<dense layer bias, stage 1>");
        hipLaunchKernelGGL(k_colsum, dim3(n2), dim3(256), 0, st, (const CsTile*)cs_d, 1);
        LAUNCH_CHECK("k_colsum<dense layer bias>");
    }
    return 0;
}

// ------------------------------------------------------------------------------------
// Embedding over all promoter bins (embed.n_layers > 1, cf_embed_full): net.py:9-59 without the centre-row shortcut
// ------------------------------------------------------------------------------------
static int embed_dense_alloc(cf_handle* h) {
    cf_handle::EmbedDense& e = h->ed;
    if (e.ready) return 0;
    const cf_config& c = h->cfg;
    const int B = c.max_batch, E = c.embed_layers;
    auto get = [&](size_t floats) -> float* {
        void* q = nullptr;
        if (hipMalloc(&q, floats * sizeof(float)) != hipSuccess) return nullptr;
        (void)hipMemset(q, 0, floats * sizeof(float));
        h->ed.owned.push_back(q);
        return (float*)q;
    };
    for (int r = 0; r < c.n_res; ++r) {
        const int L = c.n_bins[r];
        const size_t rows = (size_t)B * L;
        for (int l = 0; l <= E; ++l)
            if (!(e.x[r][l] = get(rows * kD))) return fail("embed_dense_alloc: out of memory");
        for (int l = 0; l < E; ++l)
            if (!(e.ws[r][l] = get((size_t)dense_ws(B, L, L, c.embed_dff, true).total))) return fail("embed_dense_alloc: out of memory");
        for (int k = 0; k < 3; ++k)
            if (!(e.dy[r][k] = get(rows * kD))) return fail("embed_dense_alloc: out of memory");
        if (!(e.lp_partial[r] = get(((rows + kEmbWgRows - 1) / kEmbWgRows) * kD * 8))) return fail("embed_dense_alloc: out of memory");
        if (!(e.valid[r] = reinterpret_cast<uint8_t*>(get((rows + 3) / 4 + 4)))) return fail("embed_dense_alloc: out of memory");
    }
    if (!(e.tables = get((size_t)1 << 20))) return fail("embed_dense_alloc: out of memory");
    e.B = B;
    e.ready = true;
    return 0;
}
static cf_dense_layer embed_dense_weights(const cf_handle* h, int r, int l, const float* base) {
    const std::string lp = fmt("embed.%d.transformer.layers.%d.", h->cfg.binsizes[r], l);
    auto at = [&](const std::string& n) { return base + h->table[h->index.at(lp + n)].offset; };
    cf_dense_layer w;
    const float* att = at("self_att.att.weight");      // rows [q | k | v] of the fused projection (modules.py:38)
    w.wq = att;
    w.wkv = att + (size_t)kD * kD;
    w.wo = at("self_att.ff.weight");
    w.bo = at("self_att.ff.bias");
    w.ln1_g = at("self_att.ln.weight");
    w.ln1_b = at("self_att.ln.bias");
    w.w1 = at("ff.l1.weight");
    w.b1 = at("ff.l1.bias");
    w.w2 = at("ff.l2.weight");
    w.b2 = at("ff.l2.bias");
    w.ln2_g = at("ff.ln.weight");
    w.ln2_b = at("ff.ln.bias");
    w.d_ff = h->cfg.embed_dff;
    return w;
}
// Pad mask of the promoters as the dense layer wants it: the full [B, L, L] byte mask when the caller passed the reference's
// tensor (row stride L * L: the centre-row pointer is L/2 rows into it), else validity bytes from the compact centre row
// -- exact for the dataset's structured masks  not(valid x valid)  with a real centre bin (data.py:156-161).
static void embed_dense_mask(cf_handle* h, const cf_batch* bt, int r, const uint8_t** full, const uint8_t** valid, hipStream_t st) {
    const int L = h->cfg.n_bins[r];
    if (bt->promoter_mask_stride[r] == (long long)L * L) {
        *full = bt->promoter_mask_row[r] - (size_t)(L / 2) * L;
        *valid = nullptr;
    } else {
        hipLaunchKernelGGL(k_mask_to_valid, dim3(bt->B), dim3(256), 0, st, bt->promoter_mask_row[r], bt->promoter_mask_stride[r], L, h->ed.valid[r]);
        *full = nullptr;
        *valid = h->ed.valid[r];
    }
}
static int embed_dense_forward(cf_handle* h, const cf_batch* bt, bool train, hipStream_t st) {
    if (embed_dense_alloc(h)) return -1;
    const cf_config& c = h->cfg;
    const int B = bt->B, E = c.embed_layers, T = c.i_max + 1;
    for (int r = 0; r < c.n_res; ++r) {
        const int L = c.n_bins[r];
        EmbTokArgs ta{bt->promoter_feats[r], h->pe[r], h->P_(fmt("embed.%d.lin_proj.weight", c.binsizes[r])), h->ed.x[r][0], B, L, c.n_feats};
        hipLaunchKernelGGL(k_embed_tokens, dim3(std::min<long long>((long long)B * L, 4096)), dim3(128), 0, st, ta);
        LAUNCH_CHECK("k_embed_tokens");
        const uint8_t *full, *valid;
        embed_dense_mask(h, bt, r, &full, &valid, st);
        for (int l = 0; l < E; ++l) {
            const cf_dense_layer w = embed_dense_weights(h, r, l, h->params);
            if (dense_layer_fwd(&w, h->ed.x[r][l], h->ed.x[r][l], valid, valid, full, B, L, L, h->ed.x[r][l + 1], h->ed.ws[r][l], train, st)) return -1;
        }
        hipLaunchKernelGGL(k_rows_gather, dim3(B), dim3(128), 0, st, (const float*)h->ed.x[r][E], L, L / 2, h->Rx[r][0], T * kD);
        LAUNCH_CHECK("k_rows_gather");
    }
    return 0;
}
static int embed_dense_backward(cf_handle* h, const cf_batch* bt, hipStream_t st) {
    const cf_config& c = h->cfg;
    const int B = bt->B, E = c.embed_layers;
    for (int r = 0; r < c.n_res; ++r) {
        const int L = c.n_bins[r];
        const long long n = (long long)B * L * kD;
        float *dy = h->ed.dy[r][0], *da = h->ed.dy[r][1], *db = h->ed.dy[r][2];
        hipLaunchKernelGGL(k_rows_scatter, dim3(B * L), dim3(128), 0, st, (const float*)h->edout[r], L, L / 2, dy);      // only the centre row is consumed (net.py:59)
        LAUNCH_CHECK("k_rows_scatter");
        const uint8_t *full, *valid;
        embed_dense_mask(h, bt, r, &full, &valid, st);
        for (int l = E - 1; l >= 0; --l) {
            const cf_dense_layer w = embed_dense_weights(h, r, l, h->params);
            const cf_dense_layer gw = embed_dense_weights(h, r, l, h->grads);
            cf_dense_layer_grads g{const_cast<float*>(gw.wq), const_cast<float*>(gw.wkv), const_cast<float*>(gw.wo), const_cast<float*>(gw.bo),
                                   const_cast<float*>(gw.ln1_g), const_cast<float*>(gw.ln1_b), const_cast<float*>(gw.w1), const_cast<float*>(gw.b1),
                                   const_cast<float*>(gw.w2), const_cast<float*>(gw.b2), const_cast<float*>(gw.ln2_g), const_cast<float*>(gw.ln2_b)};
            if (cf_op_dense_layer_bwd(&w, h->ed.x[r][l], h->ed.x[r][l], valid, valid, full, B, L, L, dy, da, db, &g, h->ed.ws[r][l], h->ed.tables, st)) return -1;
            hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, da, (const float*)db, n);      // self-attention: x is query and key/value input
            LAUNCH_CHECK("k_add_inplace");
            std::swap(dy, da);
        }
        const long long rows = (long long)B * L;
        const int chunks = (int)((rows + kEmbWgRows - 1) / kEmbWgRows);
        EmbTokWgArgs wa{dy, bt->promoter_feats[r], h->ed.lp_partial[r], rows, c.n_feats};
        hipLaunchKernelGGL(k_embed_tokens_wgrad, dim3(chunks), dim3(128), 0, st, wa);
        hipLaunchKernelGGL(k_embed_tokens_wgrad2, dim3(1), dim3(128), 0, st, (const float*)h->ed.lp_partial[r], chunks, c.n_feats,
                           h->G_(fmt("embed.%d.lin_proj.weight", c.binsizes[r])));
        LAUNCH_CHECK("k_embed_tokens_wgrad");
    }
    return 0;
}

// EmbeddingTransformer.forward's FIRST return value (net.py:57-59): the embeddings of every promoter bin, [B, 1, L, 128] per
// resolution, for consumers that want more than the centre row.  Forward only; not capturable.
extern "C" int cf_embed_full(cf_handle* h, const cf_batch* bt, float* const* out, void* stream) {
    if (check_batch(h, bt) || !out) return fail("cf_embed_full: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (embed_dense_forward(h, bt, false, st)) return -1;
    for (int r = 0; r < h->cfg.n_res; ++r) {
        if (!out[r]) continue;
        HIP_TRY(hipMemcpyAsync(out[r], h->ed.x[r][h->cfg.embed_layers], (size_t)bt->B * h->cfg.n_bins[r] * kD * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

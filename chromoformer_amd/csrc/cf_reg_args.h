// cf_reg_args.h -- argument structures of the fused Regulation kernels (cf_reg8.h; included by cf_kernels.h): the per-layer device table
// and the launch arguments, nothing else.  The kernels themselves are in cf_reg8.h (modules.py:28-88, 100-101, 121-124; net.py:148-153).
#pragma once

namespace cf {

struct RegLayerDev {
    const float *watt, *gamma, *wo, *bo, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2;
    const float *watt_t, *wo_t, *w1_t, *w2_t;      // tiled copies (forward products)
    const float *watt_tt, *wo_tt, *w1_tt, *w2_tt;  // tiled copies of the transposes (backward products of cf_reg8.h)
    float *xin, *qkvg, *p, *a, *xh1, *rs1, *y1, *hdn, *xh2, *rs2, *xout;
    float *dxout, *dt2, *dpre1, *dt1, *da, *dqkvg, *dxin, *partial, *dgam;
    float *hq, *dy1;                               // cf_reg8.h: per-(gene, head) attention operands; d(LayerNorm 1 output)
};
struct RegArgs {
    const RegLayerDev* tab;          // [n_res][n_layers]
    int n_layers, T, B, n_res, xcd_map;
    const uint8_t* mask[kMaxRes];    // [B,T,T]
    const float* freq;               // [B,T,T]
    int save;
    unsigned long long* tdbg;        // optional: shader-clock stamps of workgroup (0,0), 16 per layer
    HeadRide head;                   // forward (cf_reg8.h): the gene's prediction head at the tail of its last workgroup (cf_head_ride.h)
    int row0_last;                   // cf_reg8.h: only token 0 of the LAST layer's output is consumed (net.py:375): that layer computes row 0 only
    int l_top, l_bot;                // backward: the launch walks layers l_top down to l_bot (the whole stack: n_layers - 1 .. 0; the data-parallel step
                                     // runs the upper and the lower half as two launches, so that the upper half's gradients can be on the wire earlier)
};

constexpr int kQkLd = kRW + 4;
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

}  // namespace cf

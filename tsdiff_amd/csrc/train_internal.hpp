// train_internal.hpp -- shared between kernels_train.hip and train_step.hip (not part of the C ABI)
#pragma once
#include "common.hpp"

namespace tsd {

// What a dense layer's epilogue does with the product row r, column n (v = acc + bias[n]):
//   v *= C(mask_dist[r])               CFConv cutoff weight (schnet.py:92-99), forward of the filter MLP
//   v *= act'(act_pre[r,n])            adjoint of the activation that FED this layer (dgrad only)
//   v += R[r,n]                        residual / gradient accumulation (R may alias Y)
//   Y[r,n] = v;  Y2[r,n] = act(v)      the activation that FOLLOWS this layer, written next to the
//                                      pre-activation the backward pass needs
// kinds: 0 swish, 1 shifted softplus, 2 ReLU, 3 softplus (as tsd_act_fwd)
struct LinEpi {
    const float* bias = nullptr;
    const float* R = nullptr;
    const float* mask_dist = nullptr;
    float cutoff = 0.0f;
    int smooth = 0;
    const float* act_pre = nullptr;
    int dact_kind = 0;
    float* Y2 = nullptr;
    int act_kind = 0;
};

__device__ __forceinline__ float act_apply(int kind, float v) {
    return kind == 0 ? swishf(v) : kind == 1 ? sspf(v) : kind == 2 ? fmaxf(v, 0.0f) : sspf(v) + 0.69314718055994530942f;
}
__device__ __forceinline__ float act_deriv(int kind, float v) {
    const float sg = __builtin_amdgcn_rcpf(1.0f + fast_exp(-v));  // sigmoid(v)
    // d swish = sg * (1 + v * (1 - sg)) ; d ssp = d softplus = sg
    return kind == 0 ? sg * (1.0f + v * (1.0f - sg)) : kind == 2 ? (v > 0.0f ? 1.0f : 0.0f) : sg;
}
__device__ __forceinline__ void epi_store(const LinEpi& e, float v, int row, int col, size_t o, float* Y) {
    if (e.bias) v += e.bias[col];
    if (e.mask_dist) v *= cutoff_weight(e.mask_dist[row], e.cutoff, e.smooth);
    if (e.act_pre) v *= act_deriv(e.dact_kind, e.act_pre[o]);
    if (e.R) v += e.R[o];
    Y[o] = v;
    if (e.Y2) e.Y2[o] = act_apply(e.act_kind, v);
}

int linear_fwd_impl(int rows, int in, int out, const float* X, const float* W, const float* Wp, const LinEpi& epi,
                    float* Y, float* scratch, size_t scratch_floats, hipStream_t st);
// flags: 1 = dX += (sets epi.R = dX), 2 = dW / db +=, 4 = epi.R is the caller's (dX = dY W + R, R != dX).
// `epi` applies to dX (bias is ignored).  dW == NULL / db == NULL: that gradient is not computed.
int linear_bwd_impl(int rows, int in, int out, const float* X, const float* W, const float* Wp_t, const float* dY,
                    float* dX, float* dW, float* db, int flags, LinEpi epi, float* scratch, size_t scratch_floats,
                    hipStream_t st);
size_t linear_scratch_floats(int in, int out);
// one list of the edge embedding's backward chain (kernels_mlp.hip::embed_bwd_kernel): rows of [.,H] except dc [.,2H]
struct EmbedBwdList {
    tsd_edges e;
    const float *d_ea, *c0, *l0;
    float *dc0, *dc, *de, *dl0;
};
int launch_embed_bwd(int H, int rows_a, const EmbedBwdList& la, int rows_b, const EmbedBwdList& lb, const float* bond_emb,
                     const float* W1t, const float* W0t, const float* Wmt, hipStream_t st, float* amax_h2 = nullptr);
int launch_pair_bwd(int H, int rows, tsd_edges e, const int32_t* attr_row, const float* ds, const float* w2,
                    const float* g1, const float* g0, const float* W1t, const float* W0t, float* dg1, float* dg0,
                    float* dp, float* d_ea, int attr_from, int attr_shift, hipStream_t st, float* amax_h2 = nullptr);
int launch_row_gather(int H, int N, tsd_edges e, const float* W, const float* x, float* out, hipStream_t st);
struct NodeAmax {  // device words (non-negative floats, atomic max) or NULL
    float *in = nullptr, *dx1 = nullptr, *dh = nullptr, *dx2 = nullptr;
};
int launch_block_bwd(int H, int N, int first, int last, tsd_edges enc, const float* Wf, const float* dagg_in,
                     const float* dh_up, const float* w_lin1_t, const float* w_lin_t, const float* w_lin2_t,
                     const float* x2_prev, float* dx1, float* dh, float* dx2_prev, float* dagg_prev, int filter_rows,
                     tsd_edges enc_u, const float* x1, const float* f0, const float* W2t, const float* W0t, float cutoff,
                     int smooth, float* dWf, float* df0, float* d_ea, hipStream_t st, float* amax_h2 = nullptr,
                     NodeAmax node_amax = NodeAmax{});
int launch_split_reduce(int64_t n, int S, const float* part, float* dst, int accumulate, hipStream_t st);
int wgrad_batch_splits(int m, int blocks, int rows);
size_t wgrad_batch_scratch_floats(int n, int rows, int in, int out);
int launch_wgrad_batch(int n, int rows, int in, int out, const float* const* dY, const float* const* X, float* const* dW,
                       float* const* db, int accumulate, float* part, hipStream_t st,
                       const float* const* amax_h2 = nullptr);

// the optimizer rewrites every weight every step: all layout conversions of a step in a few launches.
// mode 0: W [out,in] -> forward MFMA layout [in/4][out][in%4]; 1: dgrad layout (W^T); 2: copy of `out` floats;
// 3 / 4: the forward / dgrad matrix as f16 planes (split16.hpp, the layout of tsd_pack_weights16; in, out multiples of 16)
struct PackItem {
    const float* src;
    float* dst;
    int out, in, mode;
};
int launch_pack_items(int n, const PackItem* items, hipStream_t st);

}  // namespace tsd

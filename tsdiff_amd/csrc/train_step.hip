// train_step.hip -- the training step of the condensed network as TWO library calls (reference train.py:124-152,
// models/epsnet/condensenc.py:178-239, 267-328): tsd_train_forward evaluates get_loss on the perturbed
// geometry and keeps every activation the backward pass needs in a caller-provided workspace;
// tsd_train_backward turns d(loss)/d(loss_i) into the gradient of every parameter (flat, in the order of the
// flat parameter vector).  The forward pass IS the inference forward (the fused per-tile kernels of kernels_mlp.hip /
// kernels_combo.hip in their SAVE instantiations, which write the activations next to the results; the edge
// embedding is evaluated once per undirected pair and shared by the encoder and output lists, as in api.hip);
// the backward pass is the primitive kernels of the autograd form (tsdiff_amd/train_ops.py) sequenced here
// instead of by ~450 Python-level autograd nodes per step, residual adds and gradient accumulations riding in
// the GEMM epilogues.
//
// Flat parameter vector `raw` (fp32, no padding), order of tsdiff_amd/engine.py::raw_param_names:
//   edge_encoder.bond_emb.weight [100,H], edge_encoder.mlp.layers.0.{weight [H,1], bias}, .1.{weight [H,H], bias},
//   atom_embedding.weight [100,H/2], atom_feat_embedding.weight [H/2,F],
//   per block: conv.lin1.weight, conv.lin2.{weight,bias}, conv.nn.0.{weight,bias}, conv.nn.2.{weight,bias},
//              lin.{weight,bias},
//   grad_dist_mlp.layers.{0 [H,2H], 1 [H/2,H], 2 [1,H/2]}.{weight,bias}, edge_cat.{0 [H,2H], 2 [H,H]}.{weight,bias}
#include <math.h>

#include <vector>

#include <algorithm>
#include "train_internal.hpp"

namespace tsd {

// kernels_train.hip
int launch_aggregate_bwd_filter(int H, int capacity_u, tsd_edges enc_u, const float* dagg, const float* x1, float* dWf,
                                int masked, float cutoff, int smooth, hipStream_t st);
// kernels_graph.hip
int launch_geometry(const tsd_model_cfg&, int, int, int, const float*, const int32_t*, const int32_t*, const int32_t*,
                    const uint16_t*, tsd_geometry, hipStream_t);

namespace {

struct RawLayout {
    size_t bond_emb, emlp_w0, emlp_b0, emlp_w1, emlp_b1, atom_emb, atom_feat;
    size_t layer0, layer_stride, L_lin1_w, L_lin2_w, L_lin2_b, L_nn0_w, L_nn0_b, L_nn2_w, L_nn2_b, L_lin_w, L_lin_b;
    size_t out_w0, out_b0, out_w1, out_b1, out_w2, out_b2, ecat_w0, ecat_b0, ecat_w1, ecat_b1, total;
};
RawLayout raw_layout(const tsd_model_cfg& c) {
    RawLayout L;
    const size_t H = c.hidden, F = c.feat_dim, HH = H * H;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += n; return r; };
    L.bond_emb = take(100 * H);
    L.emlp_w0 = take(H);
    L.emlp_b0 = take(H);
    L.emlp_w1 = take(HH);
    L.emlp_b1 = take(H);
    L.atom_emb = take(100 * (H / 2));
    L.atom_feat = take((H / 2) * F);
    L.layer0 = o;
    size_t lo = 0;
    auto lt = [&](size_t n) { size_t r = lo; lo += n; return r; };
    L.L_lin1_w = lt(HH);
    L.L_lin2_w = lt(HH);
    L.L_lin2_b = lt(H);
    L.L_nn0_w = lt(HH);
    L.L_nn0_b = lt(H);
    L.L_nn2_w = lt(HH);
    L.L_nn2_b = lt(H);
    L.L_lin_w = lt(HH);
    L.L_lin_b = lt(H);
    L.layer_stride = lo;
    o += lo * (size_t)c.num_convs;
    L.out_w0 = take(2 * HH);
    L.out_b0 = take(H);
    L.out_w1 = take(HH / 2);
    L.out_b1 = take(H / 2);
    L.out_w2 = take(H / 2);
    L.out_b2 = take(1);
    L.ecat_w0 = take(2 * HH);
    L.ecat_b0 = take(H);
    L.ecat_w1 = take(HH);
    L.ecat_b1 = take(H);
    L.total = o;
    return L;
}

struct Work {
    // forward state (per-edge arrays sized by CAPACITY PU = P/2; per-block arrays strided by it)
    float *featR, *featP;          // [N,F] fp32 one-hot features
    float* h;                      // [(L+1), N, H]
    float *x1, *agg, *x2, *xs;     // [L, N, H] each
    EmbedSave emb;                 // [2 PU, .]: rows [0, Eu) enc_u edges, rows [Eu, Eu + Ed) separately embedded out edges
                                   // (ONE row range: every weight gradient of the embedding is one problem of Eu + Ed rows)
    float* ea;                     // [2 PU, H] edge attributes, same rows (condensenc.py:156-176, edge.py:58-68)
    float *f0, *fs, *Wf;           // [L, PU, H] each
    float *hp, *g0, *gs0, *g1, *gs1, *s_u;  // pair MLP
    float *node_eq, *pos_target, *d_target;
    float* pack_inf;               // packed weights, layout of tsd_pack_weights (the fused forward kernels)
    float* pack_t;                 // dgrad layouts of the dense weights, offsets = raw offsets
    float* pack16;                 // f16-plane image of pack_inf (tsd_pack_weights16): the split-f16 block launches
    float* pack_t16;               // f16-plane images of the filter MLPs' dgrad matrices, offsets = raw offsets
    float* amax;                   // [L][2] running max |dWf|, |df0| per block, then [2] |dg1|, |dg0| of the pair MLP, then
                                   // [3] |d_ea|, |dc0|, |de| of the edge embedding, then the node chain's: [1] |dh_L|, [L] |dh_l|,
                                   // [L] |dx2_l|, [L] |dx1_l|
                                   // (split-f16 backward: the dY scales of the weight-gradient launches)
    float* scratch;                // linear scratch
    float* scratch2;               // the same again for the side lane of the split-f16 backward (small gradient kernels)
    size_t scratch_floats;
    // backward temporaries
    float *eA, *eB;                // [PU, 2H] each
    float *d_ea;                   // [2 PU, H], rows as `ea`
    float *nA, *nB, *nC, *dh;      // [N, H] each
    float *dhs, *dx2s, *dx1s;      // [(L+1), N, H], [L, N, H], [L, N, H]: the node-level dY of every block (batched wgrad)
    float* wpart;                  // split partials of the batched wgrads (node level, then filter level)
    float *dWfs, *df0s;            // [L, PU, H] each: the filter MLP's dY of every block (batched wgrad)
    float *e_dc0, *e_dc, *e_de, *e_dl0;  // [2 PU, H] ([2 PU, 2H] for e_dc): the edge embedding's dY, rows as `ea`
    size_t total;
};

Work carve(const tsd_model_cfg& c, int N, size_t PU, float* base) {
    Work w;
    const size_t H = c.hidden, L = c.num_convs, F = c.feat_dim;
    size_t o = 0;
    auto take = [&](size_t n) { float* r = base ? base + o : nullptr; o += (n + 63) & ~size_t(63); return r; };
    w.featR = take(2 * (size_t)N * F);  // [2 N, F]: reactant rows, then product rows (one weight-gradient problem)
    w.featP = base ? w.featR + (size_t)N * F : nullptr;
    w.h = take((L + 1) * N * H);
    w.x1 = take(L * N * H);
    w.agg = take(L * N * H);
    w.x2 = take(L * N * H);
    w.xs = take(L * N * H);
    w.emb.l0 = take(2 * PU * H);
    w.emb.s0 = take(2 * PU * H);
    w.emb.e = take(2 * PU * H);
    w.emb.c = take(2 * PU * 2 * H);
    w.emb.c0 = take(2 * PU * H);
    w.emb.s1 = take(2 * PU * H);
    w.emb.d = take(2 * PU);
    w.emb.tr = reinterpret_cast<uint8_t*>(take((2 * PU + 3) / 4));
    w.emb.tp = reinterpret_cast<uint8_t*>(take((2 * PU + 3) / 4));
    w.ea = take(2 * PU * H);
    w.f0 = take(L * PU * H);
    w.fs = take(L * PU * H);
    w.Wf = take(L * PU * H);
    w.hp = take(PU * 2 * H);
    w.g0 = take(PU * H);
    w.gs0 = take(PU * H);
    w.g1 = take(PU * (H / 2));
    w.gs1 = take(PU * (H / 2));
    w.s_u = take(PU);
    w.node_eq = take((size_t)N * 3);
    w.pos_target = take((size_t)N * 3);
    w.d_target = take(PU);
    const RawLayout R = raw_layout(c);
    w.pack_inf = take(weight_layout(c).total);
    w.pack_t = take(R.total);
    w.pack16 = take(weight_layout(c).total);
    w.pack_t16 = take(R.total);
    w.amax = take(2 * L + 2 + 3 + 1 + 3 * L);
    w.scratch_floats = linear_scratch_floats((int)(2 * H), (int)H);
    if (w.scratch_floats < (size_t)512 * 32 * H) w.scratch_floats = (size_t)512 * 32 * H;  // embedding-gradient partials
    w.scratch = take(w.scratch_floats);
    w.scratch2 = take(w.scratch_floats);
    w.eA = take(PU * 2 * H);
    w.eB = take(PU * 2 * H);
    w.d_ea = take(2 * PU * H);
    w.nA = take((size_t)N * H);
    w.nB = take((size_t)N * H);
    w.nC = take((size_t)N * H);
    w.dh = take((size_t)N * H);
    w.dhs = take((L + 1) * N * H);
    w.dx2s = take(L * N * H);
    w.dx1s = take(L * N * H);
    {
        size_t a = 0, b = 0, e2 = 0;
        if (N > 0 && H % 128 == 0) {
            a = wgrad_batch_scratch_floats((int)(3 * L), N, (int)H, (int)H);
            b = PU > 0 ? wgrad_batch_scratch_floats((int)(2 * L), (int)PU, (int)H, (int)H) : 0;
            e2 = PU > 0 ? wgrad_batch_scratch_floats(2, (int)(2 * PU), (int)H, (int)H) : 0;  // the embedding's two H x H layers
            if (PU > 0 && (H / 2) % 128 == 0) e2 = std::max(e2, wgrad_batch_scratch_floats(1, (int)PU, (int)H, (int)(H / 2)));  // pair MLP layers.1
            if (PU > 0) e2 = std::max(e2, wgrad_batch_scratch_floats(1, (int)(2 * PU), (int)(2 * H), (int)H));  // pair MLP layers.0, edge_cat.0 (split-f16 step)
        }
        w.wpart = take(std::max(a, std::max(b, e2)));
    }
    w.dWfs = take(H % 128 == 0 ? L * PU * H : 0);
    w.df0s = take(H % 128 == 0 ? L * PU * H : 0);
    w.e_dc0 = take(H % 128 == 0 ? 2 * PU * H : 0);
    w.e_dc = take(H % 128 == 0 ? 2 * PU * 2 * H : 0);
    w.e_de = take(H % 128 == 0 ? 2 * PU * H : 0);
    w.e_dl0 = take(H % 128 == 0 ? 2 * PU * H : 0);
    w.total = o;
    return w;
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

// ---- small kernels ----------------------------------------------------------------------------
__global__ void feats_to_float_kernel(int64_t n, const int64_t* __restrict__ r, const int64_t* __restrict__ p,
                                      float* __restrict__ fr, float* __restrict__ fp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        fr[i] = (float)r[i];
        fp[i] = (float)p[i];
    }
}
// z = [Emb[atom] + Wf r , Wf p - Wf r]                                          condensenc.py:193-198
__global__ void node_embed_raw_kernel(int N, int Hh, int F, const float* __restrict__ atom_emb,
                                      const float* __restrict__ Wfeat, const int64_t* __restrict__ atom,
                                      const float* __restrict__ fr, const float* __restrict__ fp,
                                      float* __restrict__ z) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * Hh) return;
    const int i = (int)(t / Hh), c = (int)(t % Hh);
    float a = 0.0f, b = 0.0f;
    for (int f = 0; f < F; ++f) {
        const float w = Wfeat[(size_t)c * F + f];
        a = fmaf(fr[(size_t)i * F + f], w, a);
        b = fmaf(fp[(size_t)i * F + f], w, b);
    }
    z[(size_t)i * 2 * Hh + c] = atom_emb[(size_t)atom[i] * Hh + c] + a;
    z[(size_t)i * 2 * Hh + Hh + c] = b - a;
}
// dz -> d(Wf r) = dz_lo - dz_hi, d(Wf p) = dz_hi
__global__ void node_embed_bwd_kernel(int N, int Hh, const float* __restrict__ dz, float* __restrict__ dfr,
                                      float* __restrict__ dfp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * Hh) return;
    const int i = (int)(t / Hh), c = (int)(t % Hh);
    const float lo = dz[(size_t)i * 2 * Hh + c], hi = dz[(size_t)i * 2 * Hh + Hh + c];
    dfr[t] = lo - hi;
    dfp[t] = hi;
}
// d atom_emb[a, c] = sum over the nodes with atom type a of dz[i, c], in node order, without atomics:
// grid (100 embedding rows, node chunks); partial [chunk][100][Hh], summed in chunk order by the second kernel.
constexpr int AE_CHUNKS = 64;
__global__ void atom_emb_grad_kernel(int N, int Hh, const int64_t* __restrict__ atom, const float* __restrict__ dz,
                                     float* __restrict__ part) {
    const int a = blockIdx.x, chunk = blockIdx.y;
    const int per = (N + AE_CHUNKS - 1) / AE_CHUNKS;
    const int i0 = chunk * per, i1 = min(N, i0 + per);
    for (int c = threadIdx.x; c < Hh; c += blockDim.x) {
        float s = 0.0f;
        for (int i = i0; i < i1; ++i)
            if (atom[i] == a) s += dz[(size_t)i * 2 * Hh + c];
        part[((size_t)chunk * 100 + a) * Hh + c] = s;
    }
}
__global__ void atom_emb_grad_reduce_kernel(int n, const float* __restrict__ part, float* __restrict__ g) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float s = 0.0f;
    for (int k = 0; k < AE_CHUNKS; ++k) s += part[(size_t)k * n + t];
    g[t] += s;
}
// de = dc_lo * emb[tr] + dc_hi * emb[tp];  d emb[t] = sum of dc_lo * e over the rows with tr == t plus dc_hi * e over
// those with tp == t.  Deterministic: every wave owns a private [ET][64] accumulator in LDS and walks its rows in
// order, the four waves are combined in a fixed order into partial[chunk][ET][H], a second kernel sums the
// chunks in order.  ET = 32 edge types: bond 1..21, 22 + hop - 1 for hop <= 7 (tsd_model_cfg orders are <= 7).
constexpr int ET = 32;
// row chunks of the embedding-table gradient: 16 rows per wave (the per-row LDS accumulation is a latency chain),
// a multiple of 16 for the split reduction (chunks past the rows write zero partials)
inline int emb_chunks(int E) {
    const int c = (E + 63) / 64 < 512 ? (E + 63) / 64 : 512;
    return (c + 15) & ~15;
}
__global__ __launch_bounds__(256) void emb_mul2_bwd_kernel(int rows, int H, int rows_per_wg,
                                                           const float* __restrict__ e, const float* __restrict__ emb,
                                                           const uint8_t* __restrict__ tr,
                                                           const uint8_t* __restrict__ tp,
                                                           const float* __restrict__ dc, float* __restrict__ de,
                                                           float* __restrict__ part) {
    __shared__ float acc[4][ET][64];
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + lane;
    const int w = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < 4 * ET * 64; t += 256) (&acc[0][0][0])[t] = 0.0f;
    __syncthreads();
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    if (c < H) {
        // four rows of the wave per trip: their loads are in flight together (rows past the end clamped), the
        // accumulations follow in row order
        for (int rb = r0 + w; rb < r1; rb += 16) {
            int a[4], b[4];
            float glo[4], ghi[4], ev[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = min(rb + 4 * k, r1 - 1);
                a[k] = min((int)tr[r], ET - 1);
                b[k] = min((int)tp[r], ET - 1);
                glo[k] = dc[(size_t)r * 2 * H + c];
                ghi[k] = dc[(size_t)r * 2 * H + H + c];
                ev[k] = e[(size_t)r * H + c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rb + 4 * k;
                if (r < r1) {
                    if (de) de[(size_t)r * H + c] = glo[k] * emb[(size_t)a[k] * H + c] + ghi[k] * emb[(size_t)b[k] * H + c];
                    acc[w][a[k]][lane] += glo[k] * ev[k];  // private to (wave, lane): plain read-modify-write
                    acc[w][b[k]][lane] += ghi[k] * ev[k];
                }
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ET * 64; t += 256) {
        const int ty = t / 64, l = t % 64, cc = blockIdx.x * 64 + l;
        if (cc < H)
            part[((size_t)blockIdx.y * ET + ty) * H + cc] = (acc[0][ty][l] + acc[1][ty][l]) + (acc[2][ty][l] + acc[3][ty][l]);
    }
}
__global__ void copy2d_kernel(int64_t rows, int cols, const float* __restrict__ src, int lds_, float* __restrict__ dst,
                              int ldd) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * cols) return;
    const int64_t r = t / cols;
    const int c = (int)(t % cols);
    dst[r * ldd + c] = src[r * lds_ + c];
}
// dst[row[r]] = src[r] for rows with distinct targets (the out edges' share of the edge-attribute gradient:
// every out_u edge owns one row of the attribute matrix, geo.attr_row)
// (targets >= from lie `shift` rows lower: PairSave::attr_from / attr_shift)
__global__ void scatter_rows_kernel(int64_t rows, int cols, const float* __restrict__ src, int lds_,
                                    const int32_t* __restrict__ row, int from, int shift, float* __restrict__ dst) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * cols) return;
    const int64_t r = t / cols;
    const int c = (int)(t % cols);
    int target = row[r];
    if (target >= from) target -= shift;
    dst[(size_t)target * cols + c] = src[r * lds_ + c];
}
// The loss head of the forward in ONE launch (d_target, both eq_transforms and the squared difference; thread = node):
//   d_target_e = (|pos0_i - pos0_j| - d_e) / sqrt(1 - a) * sqrt(a)                          condensenc.py:309-318
//   node_eq = eq_transform(s), pos_target = eq_transform(d_target)                          geometry.py:22-30
//   loss_i = |node_eq_i - pos_target_i|^2                                                   condensenc.py:326-328
// (a directed edge and its reverse give the same d_target: both distances are sums of the same squares)
__global__ void loss_head_kernel(int N, tsd_edges ed, const float* __restrict__ pos, const float* __restrict__ pos0,
                                 const int32_t* __restrict__ node_graph, const float* __restrict__ a_graph,
                                 const float* __restrict__ s, float* __restrict__ node_eq,
                                 float* __restrict__ pos_target, float* __restrict__ loss) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float a = a_graph[node_graph[i]];
    const float ca = sqrtf(1.0f - a), sa = sqrtf(a);
    const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
    const float qx = pos0[3 * i], qy = pos0[3 * i + 1], qz = pos0[3 * i + 2];
    float ax = 0.f, ay = 0.f, az = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
    const int e1 = ed.row_ptr[i + 1];
    for (int e = ed.row_ptr[i]; e < e1; ++e) {
        const int j = ed.dst[e];
        const float d = ed.dist[e];
        const float dx = px - pos[3 * j], dy = py - pos[3 * j + 1], dz = pz - pos[3 * j + 2];
        const float gx = qx - pos0[3 * j], gy = qy - pos0[3 * j + 1], gz = qz - pos0[3 * j + 2];
        const float d_gt = sqrtf(gx * gx + gy * gy + gz * gz);
        const float w = s[ed.umap[e]] / d, wt = ((d_gt - d) / ca * sa) / d;
        ax += dx * w;
        ay += dy * w;
        az += dz * w;
        tx += dx * wt;
        ty += dy * wt;
        tz += dz * wt;
    }
    const float ex = ax + ax, ey = ay + ay, ez = az + az, ux = tx + tx, uy = ty + ty, uz = tz + tz;
    node_eq[3 * i] = ex; node_eq[3 * i + 1] = ey; node_eq[3 * i + 2] = ez;
    pos_target[3 * i] = ux; pos_target[3 * i + 1] = uy; pos_target[3 * i + 2] = uz;
    const float l0 = ex - ux, l1 = ey - uy, l2 = ez - uz;
    loss[i] = l0 * l0 + l1 * l1 + l2 * l2;
}
// The head of the backward in ONE launch: workgroups [0, edge_blocks) take the loss gradient to the undirected pair
// scores -- g_i = 2 (node_eq_i - pos_target_i) dloss_i, ds[u] = 2 ((pos_i - pos_j) / d_u) . (g_i - g_j) -- and the
// remaining workgroups zero the gradient accumulators of the step (the flat parameter gradient, the attribute
// gradient, dh when no pair writes it): six launches of the primitive form (two kernels, four memsets).
struct ZeroRange {
    float* p;
    size_t n;
};
__global__ __launch_bounds__(256) void bwd_head_kernel(tsd_edges eu, int edge_blocks, const float* __restrict__ pos,
                                                       const float* __restrict__ eq, const float* __restrict__ tg,
                                                       const float* __restrict__ dloss, float* __restrict__ ds,
                                                       ZeroRange z0, ZeroRange z1, ZeroRange z2) {
    if ((int)blockIdx.x < edge_blocks) {
        const int u = blockIdx.x * blockDim.x + threadIdx.x;
        if (u >= *eu.count) return;
        const int i = eu.src[u], j = eu.dst[u];
        const float inv = 2.0f / eu.dist[u];
        const float li = dloss[i], lj = dloss[j];
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gi = 2.0f * (eq[3 * i + k] - tg[3 * i + k]) * li, gj = 2.0f * (eq[3 * j + k] - tg[3 * j + k]) * lj;
            acc += (pos[3 * i + k] - pos[3 * j + k]) * (gi - gj);
        }
        ds[u] = inv * acc;
        return;
    }
    const size_t gt = (size_t)((int)blockIdx.x - edge_blocks) * blockDim.x + threadIdx.x;
    const size_t gs = (size_t)((int)gridDim.x - edge_blocks) * blockDim.x;
    const ZeroRange z[3] = {z0, z1, z2};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const size_t n4 = z[r].n / 4;
        f32x4* p4 = reinterpret_cast<f32x4*>(z[r].p);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (size_t k = gt; k < n4; k += gs) p4[k] = zero;
        for (size_t k = n4 * 4 + gt; k < z[r].n; k += gs) z[r].p[k] = 0.0f;
    }
}
// get_loss's forward diffusion (condensenc.py:292-297): thread = atom; the first G threads also write a_graph
__global__ void diffuse_kernel(int N, int G, int T, const float* __restrict__ alphas, const int64_t* __restrict__ time_step,
                               const int64_t* __restrict__ node_graph, const float* __restrict__ pos,
                               const float* __restrict__ noise, float* __restrict__ out, float* __restrict__ a_graph) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G) a_graph[i] = alphas[min(max(time_step[i], (int64_t)0), (int64_t)T - 1)];
    if (i >= N) return;
    const int64_t g = min(max(node_graph[i], (int64_t)0), (int64_t)G - 1);
    const float a = alphas[min(max(time_step[g], (int64_t)0), (int64_t)T - 1)];
    const float s1 = sqrtf(1.0f - a), s2 = sqrtf(a);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[3 * i + k] = pos[3 * i + k] + (noise[3 * i + k] * s1) / s2;
}

// the four words tsd_train_forward needs on the host -- [enc_u count, out_u count, topology status, diff_u count] -- written
// straight into pinned (device-visible) host memory: no staging copy, one launch
__global__ void counts_to_host_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b, const int32_t* __restrict__ st,
                                      const int32_t* __restrict__ d, int32_t* __restrict__ host4) {
    if (threadIdx.x == 0) {
        host4[0] = *a;
        host4[1] = *b;
        host4[2] = st ? *st : 0;
        host4[3] = *d;
        __threadfence_system();
    }
}

// ---- optimizer on the flat parameter / gradient vectors (reference train.py:144-145, utils/common.py:58-68) ----
// |g|: fixed-order two-stage sum of squares (NORM_WG workgroups, then one), no atomics
constexpr int NORM_WG = 1024;
__global__ __launch_bounds__(256) void sumsq_stage1_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ part) {
    __shared__ float sm[256];
    const int64_t per = ((n + NORM_WG - 1) / NORM_WG + 3) & ~int64_t(3);
    const int64_t i0 = blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    float s = 0.0f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) s = fmaf(g[i], g[i], s);
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}
__global__ __launch_bounds__(256) void sumsq_stage2_kernel(const float* __restrict__ part, float* __restrict__ norm) {
    __shared__ float sm[256];
    float s = 0.0f;
    for (int k = threadIdx.x; k < NORM_WG; k += 256) s += part[k];  // fixed order per thread
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) norm[0] = sqrtf(sm[0]);
}
// g *= min(1, max_norm / (norm + 1e-6))                               torch.nn.utils.clip_grad_norm_
__global__ void clip_scale_kernel(int64_t n, float* __restrict__ g, const float* __restrict__ norm, float max_norm) {
    // torch: clip_coef_clamped = clamp(max_norm / (total_norm + 1e-6), max=1.0), every gradient multiplied by it --
    // a NaN total norm gives a NaN coefficient (torch.clamp keeps NaN) and NaN gradients.  Only the exact no-op
    // (coef >= 1) is skipped.
    const float coef = max_norm / (norm[0] + 1e-6f);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(coef >= 1.0f)) g[i] *= coef;
}
// torch.optim.Adam (no amsgrad, minimise): g += wd p; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr, float b1, float b2, float eps, float wd, float step_size,
                            float inv_sqrt_bc2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.0f) gi = fmaf(wd, pi, gi);
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}

struct SideLane {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, fork2 = nullptr, join = nullptr;
};
struct Ctx {
    const tsd_model_cfg* c;
    const tsd_batch* b;
    RawLayout R;
    Work w;
    const float* raw;
    float* grad;  // backward only
    hipStream_t st;
    int N, H, L, F, PU, Eu, Eo, Ed;
    bool packed(int in, int out) const { return (in == 128 || in == 256 || in == 512) && (out == 128 || out == 256 || out == 512); }
    // all parameter gradients accumulate (the flat gradient is zeroed once per backward).
    // dact_kind >= 0: dX is multiplied by act'(act_pre) -- the adjoint of the activation that fed this layer
    int lin_bwd(int rows, int in, int out, const float* X, size_t w_off, long b_off, const float* dY, float* dX,
                bool accumulate_dx, int dact_kind = -1, const float* act_pre = nullptr) const {
        const float* Wpt = packed(out, in) ? w.pack_t + w_off : nullptr;
        LinEpi e;
        if (dact_kind >= 0) {
            e.act_pre = act_pre;
            e.dact_kind = dact_kind;
        }
        return linear_bwd_impl(rows, in, out, X, raw + w_off, Wpt, dY, dX, grad + w_off, b_off >= 0 ? grad + b_off : nullptr,
                               2 | (accumulate_dx ? 1 : 0), e, w.scratch, w.scratch_floats, st);
    }
    // dX = dY W (* act'(act_pre)) (+ Rsd) only: the weight gradient of the layer is taken later, batched
    int dgrad(int rows, int in, int out, size_t w_off, const float* dY, float* dX, const float* Rsd, int dact_kind = -1,
              const float* act_pre = nullptr) const {
        const float* Wpt = packed(out, in) ? w.pack_t + w_off : nullptr;
        LinEpi e;
        if (dact_kind >= 0) {
            e.act_pre = act_pre;
            e.dact_kind = dact_kind;
        }
        e.R = Rsd;
        return linear_bwd_impl(rows, in, out, nullptr, raw + w_off, Wpt, dY, dX, nullptr, nullptr, 4, e, w.scratch,
                               w.scratch_floats, st);
    }
};

#define TSD_TRY(expr)          \
    do {                       \
        int _r = (expr);       \
        if (_r) return _r;     \
    } while (0)

// Every layout conversion of the step in two launches: the packed arena the fused forward kernels read
// (layout of tsd_pack_weights) and the dgrad layouts of the dense weights (offsets of the raw vector).
int pack_all(const Ctx& x, bool h2) {
    const int H = x.H, L = x.L;
    const WeightLayout I = weight_layout(*x.c);
    const RawLayout& R = x.R;
    std::vector<PackItem> it;
    it.reserve(128);
    float* D = x.w.pack_inf;
    // h2: the interaction blocks' matrices and biases also as the f16-plane arena of the split-f16 block launches
    // (same offsets as the fp32 arena: tsd_pack_weights16's image of these ranges), in the same launch
    float* D16 = x.w.pack16;
    bool blocks = true;  // (every matrix and vector of the forward also goes to the f16 arena)
    auto cp = [&](size_t dst, size_t src, int n) {
        it.push_back({x.raw + src, D + dst, n, 1, 2});
        if (h2 && blocks) it.push_back({x.raw + src, D16 + dst, n, 1, 2});
    };
    auto pk = [&](size_t dst, size_t src, int out, int in, bool t16 = false) {
        it.push_back({x.raw + src, D + dst, out, in, 0});
        if (x.packed(out, in)) it.push_back({x.raw + src, x.w.pack_t + src, out, in, 1});
        if (h2 && blocks) it.push_back({x.raw + src, D16 + dst, out, in, 3});
        if (h2 && t16) it.push_back({x.raw + src, x.w.pack_t16 + src, out, in, 4});
    };
    cp(I.bond_emb, R.bond_emb, 100 * H);
    cp(I.emlp_w0, R.emlp_w0, H);
    cp(I.emlp_b0, R.emlp_b0, H);
    pk(I.emlp_w1, R.emlp_w1, H, H, true);
    cp(I.emlp_b1, R.emlp_b1, H);
    pk(I.ecat_w0, R.ecat_w0, H, 2 * H, true);
    cp(I.ecat_b0, R.ecat_b0, H);
    pk(I.ecat_w1, R.ecat_w1, H, H, true);
    cp(I.ecat_b1, R.ecat_b1, H);
    for (int l = 0; l < L; ++l) {
        const size_t i = I.layer0 + (size_t)l * I.layer_stride, r = R.layer0 + (size_t)l * R.layer_stride;
        pk(i + I.L_lin1_w, r + R.L_lin1_w, H, H);
        pk(i + I.L_lin2_w, r + R.L_lin2_w, H, H);
        cp(i + I.L_lin2_b, r + R.L_lin2_b, H);
        pk(i + I.L_nn0_w, r + R.L_nn0_w, H, H, true);
        cp(i + I.L_nn0_b, r + R.L_nn0_b, H);
        pk(i + I.L_nn2_w, r + R.L_nn2_w, H, H, true);
        cp(i + I.L_nn2_b, r + R.L_nn2_b, H);
        pk(i + I.L_lin_w, r + R.L_lin_w, H, H);
        cp(i + I.L_lin_b, r + R.L_lin_b, H);
    }
    // (the pair MLP as the blocks: forward planes in the f16 arena; dgrad planes below)
    pk(I.out_w0, R.out_w0, H, 2 * H, true);
    cp(I.out_b0, R.out_b0, H);
    pk(I.out_w1, R.out_w1, H / 2, H, true);
    cp(I.out_b1, R.out_b1, H / 2);
    cp(I.out_w2, R.out_w2, H / 2);
    cp(I.out_b2, R.out_b2, 1);
    return launch_pack_items((int)it.size(), it.data(), x.st);
}

// d_ea [E,H] -> parameter gradients (eA / eB are its temporaries: d_ea must not alias them)
int embed_bwd(const Ctx& x, const tsd_edges& lst, int E, const EmbedSave& s, const float* d_ea) {
    if (E == 0) return TSD_OK;
    const int H = x.H;
    float* tA = x.w.eA;             // [E,2H]
    float* tB = x.w.eB;             // [E,2H]
    TSD_TRY(x.lin_bwd(E, H, H, s.s1, x.R.ecat_w1, (long)x.R.ecat_b1, d_ea, tB, false, 0, s.c0));   // dc0
    TSD_TRY(x.lin_bwd(E, 2 * H, H, s.c, x.R.ecat_w0, (long)x.R.ecat_b0, tB, tA, false));           // dc [E,2H]
    const int chunks = emb_chunks(E);
    hipLaunchKernelGGL(emb_mul2_bwd_kernel, dim3((H + 63) / 64, chunks), dim3(256), 0, x.st, E, H,
                       (E + chunks - 1) / chunks, s.e, x.raw + x.R.bond_emb, lst.type_r, lst.type_p, tA, tB,
                       x.w.scratch);                                                                // de -> tB
    TSD_TRY(launch_split_reduce((int64_t)ET * H, chunks, x.w.scratch, x.grad + x.R.bond_emb, 1, x.st));  // rows [0, ET) of the [100, H] table
    TSD_TRY(x.lin_bwd(E, H, H, s.s0, x.R.emlp_w1, (long)x.R.emlp_b1, tB, tA, false, 0, s.l0));     // dl0
    TSD_TRY(x.lin_bwd(E, 1, H, lst.dist, x.R.emlp_w0, (long)x.R.emlp_b0, tA, nullptr, false));
    TSD_LAUNCH_CHECK("embed_bwd");
    return TSD_OK;
}

// MFMA sizes: the dgrad chain of BOTH lists in one tile-kernel launch (launch_embed_bwd), then the weight / table
// gradients per list from the dY it wrote.  d_ea: [2 PU, H], rows as the edge-attribute matrix.
int embed_bwd_fused(const Ctx& x, const tsd_geometry& g, const float* d_ea, bool h2, SideLane* side = nullptr) {
    const int H = x.H;
    const Work& w = x.w;
    const size_t o1 = (size_t)x.Eu * H, o2 = (size_t)x.Eu * 2 * H;  // first row of the second list
    const EmbedBwdList la{g.enc_u, d_ea, w.emb.c0, w.emb.l0, w.e_dc0, w.e_dc, w.e_de, w.e_dl0};
    const EmbedBwdList lb{g.diff_u, d_ea + o1, w.emb.c0 + o1, w.emb.l0 + o1, w.e_dc0 + o1, w.e_dc + o2, w.e_de + o1, w.e_dl0 + o1};
    // (split-f16 step: the chain on f16 MFMA; its three dY tensors keep their maxima in amax[2 L + 2 ..] for the weight
    // gradients below, whose X -- s1, s0, c -- passed the range check of the split-f16 forward's embedding tiles)
    const float* Wt = h2 ? w.pack_t16 : w.pack_t;
    float* eamax = w.amax + 2 * x.L + 2;
    TSD_TRY(launch_embed_bwd(H, x.Eu, la, x.Ed, lb, x.raw + x.R.bond_emb, Wt + x.R.ecat_w1, Wt + x.R.ecat_w0,
                             Wt + x.R.emlp_w1, x.st, h2 ? eamax : nullptr));
    if (side != nullptr) {
        TSD_HIP(hipEventRecord(side->fork2, x.st));
        TSD_HIP(hipStreamWaitEvent(side->s, side->fork2, 0));
    }
    // weight gradients only (dX == NULL): X, dY per layer, the rows of BOTH lists as one problem (they are contiguous:
    // the forward saved the second list's rows, distances and types right behind the first's)
    const int E = x.Eu + x.Ed;
    if (E > 0) {
        {   // the two H x H layers (edge_cat.2, the distance MLP's second layer) share a launch
            const float* dYs[2] = {d_ea, w.e_de};
            const float* Xs[2] = {w.emb.s1, w.emb.s0};
            float* dWs[2] = {x.grad + x.R.ecat_w1, x.grad + x.R.emlp_w1};
            float* dbs[2] = {x.grad + x.R.ecat_b1, x.grad + x.R.emlp_b1};
            const float* am[2] = {eamax, eamax + 2};
            TSD_TRY(launch_wgrad_batch(2, E, H, H, dYs, Xs, dWs, dbs, 1, w.wpart, x.st, h2 ? am : nullptr));
        }
        if (h2) {
            const float* dYs[1] = {w.e_dc0};
            const float* Xs[1] = {w.emb.c};
            float* dWs[1] = {x.grad + x.R.ecat_w0};
            float* dbs[1] = {x.grad + x.R.ecat_b0};
            const float* am[1] = {eamax + 1};
            TSD_TRY(launch_wgrad_batch(1, E, 2 * H, H, dYs, Xs, dWs, dbs, 1, w.wpart, x.st, am));
        } else {
            TSD_TRY(x.lin_bwd(E, 2 * H, H, w.emb.c, x.R.ecat_w0, (long)x.R.ecat_b0, w.e_dc0, nullptr, false));
        }
        // (side lane: the table / narrow-layer gradients only read what the chain kernel above wrote)
        Ctx xs = x;
        if (side != nullptr) {
            xs.st = side->s;
            xs.w.scratch = w.scratch2;
        }
        const int chunks = emb_chunks(E);
        hipLaunchKernelGGL(emb_mul2_bwd_kernel, dim3((H + 63) / 64, chunks), dim3(256), 0, xs.st, E, H,
                           (E + chunks - 1) / chunks, w.emb.e, x.raw + x.R.bond_emb, w.emb.tr, w.emb.tp, w.e_dc,
                           (float*)nullptr, xs.w.scratch);
        TSD_TRY(launch_split_reduce((int64_t)ET * H, chunks, xs.w.scratch, x.grad + x.R.bond_emb, 1, xs.st));  // rows [0, ET) of the [100, H] table
        TSD_TRY(xs.lin_bwd(E, 1, H, w.emb.d, x.R.emlp_w0, (long)x.R.emlp_b0, w.e_dl0, nullptr, false));
    }
    TSD_LAUNCH_CHECK("embed_bwd_fused");
    return TSD_OK;
}

// the save arrays of the edge embedding from row `row` on
EmbedSave embed_rows(const EmbedSave& s, size_t row, size_t H) {
    return EmbedSave{s.l0 + row * H, s.s0 + row * H, s.e + row * H, s.c + row * 2 * H, s.c0 + row * H, s.s1 + row * H};
}

// node embedding: dz = d loss / d h_0
// (d(Wf r) rows then d(Wf p) rows in nA [2 N, H/2], against featR | featP [2 N, F]: atom_feat_embedding's weight
// gradient is one problem of 2 N rows)
int node_embed_grads(const Ctx& x, const float* dz, const int64_t* atom_type, hipStream_t st, float* scratch) {
    const Work& w = x.w;
    const int N = x.N, H = x.H, F = x.F;
    hipLaunchKernelGGL(node_embed_bwd_kernel, dim3(nblk((int64_t)N * (H / 2))), dim3(256), 0, st, N, H / 2, dz, w.nA,
                       w.nA + (size_t)N * (H / 2));
    hipLaunchKernelGGL(atom_emb_grad_kernel, dim3(100, AE_CHUNKS), dim3(H / 2 < 256 ? H / 2 : 256), 0, st, N, H / 2,
                       atom_type, dz, scratch);
    hipLaunchKernelGGL(atom_emb_grad_reduce_kernel, dim3(nblk(100 * (H / 2))), dim3(256), 0, st, 100 * (H / 2), scratch,
                       x.grad + x.R.atom_emb);
    return linear_bwd_impl(2 * N, F, H / 2, w.featR, x.raw + x.R.atom_feat, nullptr, w.nA, nullptr, x.grad + x.R.atom_feat,
                           nullptr, 2, LinEpi(), scratch, w.scratch_floats, st);
}

// A second stream of the library's own for the backward's small, latency-bound gradient launches (embedding tables, narrow
// layers: nine launches of a few workgroups each, ~115 us one behind the other): forked where their inputs are final, they
// run beside the batched weight-gradient launches of the caller's stream and join it before the call returns.
int side_lane(SideLane** out) {
    static thread_local SideLane lanes[16];
    int dev = 0;
    TSD_HIP(hipGetDevice(&dev));
    SideLane& l = lanes[dev & 15];
    if (l.s == nullptr) {
        TSD_HIP(hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
        TSD_HIP(hipEventCreateWithFlags(&l.fork, hipEventDisableTiming));
        TSD_HIP(hipEventCreateWithFlags(&l.fork2, hipEventDisableTiming));
        TSD_HIP(hipEventCreateWithFlags(&l.join, hipEventDisableTiming));
    }
    *out = &l;
    return TSD_OK;
}

// tsd_batch.reserved bit 5 asks for the split-f16 step; its tile kernels exist for the shipped width (hidden = 256), any other
// width runs the fp32-MFMA step whatever the bit says
bool train_h2(const tsd_model_cfg* cfg, const tsd_batch* b) {
    return cfg && b && (b->reserved & 32) != 0 && b->status != nullptr && cfg->hidden == 256;
}

int make_ctx(Ctx& x, const tsd_model_cfg* cfg, const tsd_batch* b, const float* raw, float* ws, size_t ws_floats,
             const int32_t* counts_host, hipStream_t st) {
    TSD_REQUIRE(cfg && b && raw && ws && counts_host, "null pointer");
    TSD_REQUIRE(hidden_supported(cfg->hidden), "hidden=%d unsupported (64/128/256)", cfg->hidden);
    x.c = cfg;
    x.b = b;
    x.R = raw_layout(*cfg);
    x.raw = raw;
    x.grad = nullptr;
    x.st = st;
    x.N = b->num_nodes;
    x.H = cfg->hidden;
    x.L = cfg->num_convs;
    x.F = cfg->feat_dim;
    x.PU = b->num_pairs / 2;
    x.Eu = counts_host[0];
    x.Eo = counts_host[1];
    x.Ed = counts_host[3];
    TSD_REQUIRE(x.Eu >= 0 && x.Eu <= x.PU && x.Eo >= 0 && x.Eo <= x.PU && x.Ed >= 0 && x.Ed <= x.PU,
                "edge counts (%d, %d, %d) outside the capacity %d", x.Eu, x.Eo, x.Ed, x.PU);
    x.w = carve(*cfg, x.N, (size_t)x.PU, ws);
    TSD_REQUIRE(x.w.total <= ws_floats, "training workspace too small: %zu < %zu floats", ws_floats, x.w.total);
    return TSD_OK;
}

}  // namespace
}  // namespace tsd

using namespace tsd;

extern "C" {

size_t tsd_train_raw_floats(const tsd_model_cfg* cfg) { return cfg ? raw_layout(*cfg).total : 0; }

size_t tsd_train_workspace_floats(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_pairs) {
    if (!cfg) return 0;
    return carve(*cfg, num_nodes, (size_t)num_pairs / 2, nullptr).total;
}

int tsd_grad_norm_clip(int64_t n, float* grad, float max_norm, float* scratch, float* norm, void* stream) {
    TraceRange range("tsd:grad_norm_clip");
    hipStream_t st = (hipStream_t)stream;
    TSD_REQUIRE(n >= 0 && grad && scratch && norm, "null pointer");
    hipLaunchKernelGGL(sumsq_stage1_kernel, dim3(NORM_WG), dim3(256), 0, st, n, grad, scratch);
    hipLaunchKernelGGL(sumsq_stage2_kernel, dim3(1), dim3(256), 0, st, scratch, norm);
    if (max_norm > 0.0f && n > 0)
        hipLaunchKernelGGL(clip_scale_kernel, dim3(nblk(n)), dim3(256), 0, st, n, grad, norm, max_norm);
    TSD_LAUNCH_CHECK("grad_norm_clip");
    return TSD_OK;
}

int tsd_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, void* stream) {
    TraceRange range("tsd:adam_step");
    TSD_REQUIRE(n >= 0 && param && grad && exp_avg && exp_avg_sq && step >= 1, "bad argument");
    if (n == 0) return TSD_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, n, param, grad, exp_avg, exp_avg_sq,
                       lr, beta1, beta2, eps, weight_decay, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)));
    TSD_LAUNCH_CHECK("adam_step");
    return TSD_OK;
}

int tsd_diffuse_positions(int32_t num_nodes, int32_t num_graphs, int32_t num_timesteps, const float* alphas,
                          const int64_t* time_step, const int64_t* node_graph, const float* pos, const float* noise,
                          float* pos_perturbed, float* a_graph, void* stream) {
    TSD_REQUIRE(num_nodes >= 0 && num_graphs >= 0 && num_timesteps > 0, "bad sizes");
    const int n = num_nodes > num_graphs ? num_nodes : num_graphs;
    if (n == 0) return TSD_OK;
    TSD_REQUIRE(alphas && time_step && a_graph && (num_nodes == 0 || (node_graph && pos && noise && pos_perturbed)),
                "null pointer");
    hipLaunchKernelGGL(diffuse_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, num_nodes, num_graphs,
                       num_timesteps, alphas, time_step, node_graph, pos, noise, pos_perturbed, a_graph);
    TSD_LAUNCH_CHECK("diffuse_positions");
    return TSD_OK;
}

int tsd_geometry_counts_async(tsd_geometry geo, const int32_t* topo_status, int32_t* counts_pinned, void* stream) {
    TSD_REQUIRE(geo.enc_u.count && geo.out_u.count && geo.diff_u.count && counts_pinned, "null pointer");
    hipLaunchKernelGGL(counts_to_host_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, geo.enc_u.count, geo.out_u.count,
                       topo_status, geo.diff_u.count, counts_pinned);
    TSD_LAUNCH_CHECK("geometry_counts_async");
    return TSD_OK;
}

int tsd_train_forward(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* raw, const int64_t* atom_type,
                      const int64_t* r_feat, const int64_t* p_feat, const float* pos0, const float* pos,
                      const float* a_graph, const int32_t* topo_status, float* workspace, size_t workspace_floats,
                      float* loss, int32_t* counts_host, void* stream) {
    TraceRange range("tsd:train_forward");
    hipStream_t st = (hipStream_t)stream;
    TSD_REQUIRE(cfg && batch && raw && atom_type && r_feat && p_feat && pos0 && pos && a_graph && workspace && loss &&
                    counts_host, "null pointer");
    const tsd_geometry& g = batch->geo;
    // tsd_batch.reserved bit 7 (round 6): the caller built the edge lists of `pos` ahead (tsd_geometry_build on these
    // buffers, e.g. on a side stream beside the previous step -- CondenseEncoderEpsNetwork.prefetch_batch(pos=...)) and
    // hands the counts over in counts_host[0, 1, 3] (+ the topology status word in [2]): no build, and no host wait in
    // this call -- the forward's launches queue up behind the previous step's backward pass
    if (!(batch->reserved & 128)) {
    TSD_TRY(launch_geometry(*cfg, batch->num_nodes, batch->num_graphs, batch->num_pairs, pos, batch->graph_ptr,
                            batch->node_graph, batch->pair_ptr, batch->pair_code, g, st));
    // the edge counts size the launches of the backward pass: the one host sync of the step.  They land in a pinned
    // staging block (a device-to-host copy into pageable memory is a blocking staged copy each: four of them cost
    // ~80 us of idle GPU per step)
    static thread_local int32_t* pinned = nullptr;
    if (!pinned) TSD_HIP(hipHostMalloc(reinterpret_cast<void**>(&pinned), 4 * sizeof(int32_t), hipHostMallocDefault));
    pinned[2] = 0;
    TSD_HIP(hipMemcpyAsync(&pinned[0], g.enc_u.count, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TSD_HIP(hipMemcpyAsync(&pinned[1], g.out_u.count, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TSD_HIP(hipMemcpyAsync(&pinned[3], g.diff_u.count, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (topo_status) TSD_HIP(hipMemcpyAsync(&pinned[2], topo_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TSD_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 4; ++k) counts_host[k] = pinned[k];
    }
    if (counts_host[2] & (TSD_STATUS_BAD_BOND | TSD_STATUS_ASYMMETRIC)) return TSD_OK;  // the caller raises
    Ctx x;
    TSD_TRY(make_ctx(x, cfg, batch, raw, workspace, workspace_floats, counts_host, st));
    const int N = x.N, H = x.H, L = x.L, F = x.F, PU = x.PU, Eu = x.Eu, Eo = x.Eo, Ed = x.Ed;
    const Work& w = x.w;
    if (N == 0) return TSD_OK;
    {
        TraceRange ph("tsd:train_forward/pack_weights");
        TSD_TRY(pack_all(x, train_h2(cfg, batch)));
    }
    const float* W = w.pack_inf;
    // tsd_batch.reserved bit 5: the interaction blocks on the f16 MFMA pipes (split-f16 operands, split16.hpp; fp32
    // accumulation and fp32 saved activations), range flag in tsd_batch.status as in the inference forward
    const bool h2 = train_h2(cfg, batch);
    Prec prec;
    if (h2) {
        prec.mode = PREC_H2;
        prec.range_status = batch->status;
    }
    // node embedding
    hipLaunchKernelGGL(feats_to_float_kernel, dim3(nblk((int64_t)N * F)), dim3(256), 0, st, (int64_t)N * F, r_feat, p_feat,
                       w.featR, w.featP);
    hipLaunchKernelGGL(node_embed_raw_kernel, dim3(nblk((int64_t)N * (H / 2))), dim3(256), 0, st, N, H / 2, F,
                       raw + x.R.atom_emb, raw + x.R.atom_feat, atom_type, w.featR, w.featP, w.h);
    TSD_LAUNCH_CHECK("node_embed_raw");
    const size_t NH = (size_t)N * H, CH = (size_t)PU * H;  // strides of the per-block node / edge arrays
    (void)F;
    // edge attributes of every undirected pair once: the enc_u rows, then the out edges that differ (geo.attr_row)
    // (rows [0, Eu) and [Eu, Eu + Ed) of the attribute matrix and of every saved activation: geo.attr_row numbers the
    // second range from PU on, the pair kernels shift it down by PU - Eu)
    if (h2) {
        TSD_TRY(launch_edge_embed_save_h(*cfg, w.pack16, Eu, g.enc_u, w.ea, Ed, g.diff_u, w.ea + (size_t)Eu * H, st, w.emb, Eu,
                                         batch->status));
    } else {
        TSD_TRY(launch_edge_embed2(*cfg, W, Eu, g.enc_u, w.ea, Ed, g.diff_u, w.ea + (size_t)Eu * H, 1, 0, st, nullptr, &w.emb, Eu));
    }
    // one launch per interaction block: node chain of block l || filter GEMMs of block l+1   schnet.py:88-128, 223-224
    const int tpl = filter_tiles_per_layer(PU);
    const FilterSave fsv{w.f0, w.fs};
    for (int j = 0; j <= L; ++j) {
        const int layer = j - 1;  // -1: x1_0 = lin1_0(h_0) only
        const int lc = layer < 0 ? 0 : layer;
        const NodeSave nsv{w.agg + lc * NH, w.x2 + lc * NH, w.xs + lc * NH};
        TSD_TRY(launch_layer_combo(*cfg, h2 ? w.pack16 : W, layer, N, g.enc, w.Wf + lc * CH, w.x1 + lc * NH, w.h + lc * NH,
                                   w.h + (size_t)(layer + 1) * NH, layer + 1 < L ? w.x1 + (size_t)(layer + 1) * NH : w.nA,
                                   0, j * tpl, j < L ? tpl : 0, PU, g.enc_u, w.ea, w.Wf, L, 1, 0, 0, 0, st, nullptr, 0,
                                   &fsv, &nsv, false, prec));
    }
    // pair MLP on [h_i * h_j , edge_attr_out]                                       common.py:226-229
    if (Eo > 0) {
        PairSave psv{w.hp, w.g0, w.gs0, w.g1, w.gs1};
        psv.attr_from = PU;
        psv.attr_shift = PU - Eu;
        TSD_TRY(launch_pair_output(*cfg, h2 ? w.pack16 : W, Eo, g.out_u, w.h + (size_t)L * NH, w.ea, g.attr_row, w.s_u, 1,
                                   0, 0, 0, st, nullptr, 0, &psv, false, prec));
    }
    // loss                                                                      condensenc.py:303-328
    hipLaunchKernelGGL(loss_head_kernel, dim3((N + 63) / 64), dim3(64), 0, st, N, g.out, pos, pos0, batch->node_graph,
                       a_graph, w.s_u, w.node_eq, w.pos_target, loss);
    TSD_LAUNCH_CHECK("train_forward");
    return TSD_OK;
}

int tsd_train_grad_buckets(const tsd_model_cfg* cfg, size_t* out) {
    TSD_REQUIRE(cfg != nullptr && out != nullptr, "null pointer");
    const RawLayout R = raw_layout(*cfg);
    out[0] = R.layer0;
    out[1] = R.layer_stride * (size_t)cfg->num_convs;
    out[2] = R.total;
    return TSD_OK;
}

int tsd_train_backward(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* raw, const int64_t* atom_type,
                       const float* pos, float* workspace, size_t workspace_floats, const int32_t* counts_host,
                       const float* dloss, float* grad, void* stream) {
    return tsd_train_backward2(cfg, batch, raw, atom_type, pos, workspace, workspace_floats, counts_host, dloss, grad, nullptr,
                               stream);
}

int tsd_train_backward2(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* raw, const int64_t* atom_type,
                        const float* pos, float* workspace, size_t workspace_floats, const int32_t* counts_host,
                        const float* dloss, float* grad, void* blocks_done_event, void* stream) {
    TraceRange range("tsd:train_backward");
    hipStream_t st = (hipStream_t)stream;
    TSD_REQUIRE(dloss && grad && atom_type && pos, "null pointer");
    TSD_REQUIRE((reinterpret_cast<uintptr_t>(grad) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                "grad and workspace must be 16-byte aligned");
    const bool h2 = train_h2(cfg, batch);
    if (h2) {
        // The forward of this step ran on split-f16 operands: its range flag decides whether the saved activations can be
        // differentiated.  One 4-byte read behind the forward's last kernel, from C++ so that the first backward launch
        // follows the answer by microseconds (the host is ahead of the GPU here: it only waits for the forward to end)
        static thread_local int32_t* pinned = nullptr;
        if (!pinned) TSD_HIP(hipHostMalloc(reinterpret_cast<void**>(&pinned), sizeof(int32_t), hipHostMallocDefault));
        TSD_HIP(hipMemcpyAsync(pinned, batch->status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        TSD_HIP(hipStreamSynchronize(st));
        if (*pinned & TSD_STATUS_RANGE) {
            set_error("split-f16 training step: an activation of the forward left the f16 range (TSD_STATUS_RANGE)");
            return TSD_ERR_RANGE;
        }
    }
    Ctx x;
    TSD_TRY(make_ctx(x, cfg, batch, raw, workspace, workspace_floats, counts_host, st));
    x.grad = grad;
    const tsd_geometry& g = batch->geo;
    const int N = x.N, H = x.H, L = x.L, F = x.F, PU = x.PU, Eu = x.Eu, Eo = x.Eo, Ed = x.Ed;
    const Work& w = x.w;
    if (N == 0) {   // an empty shard: all-zero gradient; the blocks' range is final behind the fill (the caller's side-stream
                    // all-reduce waits for this event: it must not run beside the memset)
        TSD_HIP(hipMemsetAsync(grad, 0, x.R.total * sizeof(float), st));
        if (blocks_done_event != nullptr) TSD_HIP(hipEventRecord((hipEvent_t)blocks_done_event, st));
        return TSD_OK;
    }
    const size_t NH = (size_t)N * H, EH = (size_t)PU * H;  // strides of the per-block node / edge arrays
    float* ds = w.d_target;  // [Eo]
    {
        // loss -> node_eq -> s_u, and the zero fill of the accumulators: the flat gradient (every weight gradient
        // accumulates), the attribute gradient of every embedded pair (rows [0, Eu) enc_u edges, rows [Eu, Eu + Ed)
        // the out edges' own), dh when there is no pair to write it
        const int edge_blocks = (int)nblk(Eo);
        // (third range: dh when there is no pair to write it -- else, split-f16 step, the running-max words of the backward's
        // dY tensors, which were a memset launch of their own until round 6)
        const bool z_amax = h2 && Eo > 0 && H == 256;
        const ZeroRange z0{grad, x.R.total}, z1{w.d_ea, (size_t)(Eu + Ed) * H},
            z2{z_amax ? w.amax : w.dh, z_amax ? 5 * (size_t)L + 6 : (Eo > 0 ? (size_t)0 : NH)};
        const size_t zfloats = z0.n + z1.n + z2.n;
        const int zero_blocks = (int)std::min<size_t>(2048, (zfloats / 4 + 255) / 256 + 1);
        hipLaunchKernelGGL(bwd_head_kernel, dim3(edge_blocks + zero_blocks), dim3(256), 0, st, g.out_u, edge_blocks, pos,
                           w.node_eq, w.pos_target, dloss, ds, z0, z1, z2);
    }
    if (Eo > 0) {
        float* dp = w.eA;  // [Eo,H]
        if (H == 256) {
            // the three dgrads, the split of dhp and the scatter of its right half as ONE tile kernel; the weight
            // gradients from the dY it wrote
            float *dg1 = w.eA + (size_t)PU * H, *dg0 = w.eB;  // [Eo,H/2] behind dp, [Eo,H]
            // (split-f16 step: the chain on f16 MFMA, the weight gradients with dY scaled by the maxima it kept; their X --
            // gs0, hp -- passed the range check of the split-f16 forward's pair tiles)
            float* pamax = w.amax + 2 * L;
            TSD_TRY(launch_pair_bwd(H, Eo, g.out_u, g.attr_row, ds, raw + x.R.out_w2, w.g1, w.g0,
                                    (h2 ? w.pack_t16 : w.pack_t) + x.R.out_w1, (h2 ? w.pack_t16 : w.pack_t) + x.R.out_w0, dg1,
                                    dg0, dp, w.d_ea, PU, PU - Eu, st, h2 ? pamax : nullptr));
            TSD_TRY(x.lin_bwd(Eo, H / 2, 1, w.gs1, x.R.out_w2, (long)x.R.out_b2, ds, nullptr, false));
            {   // [H/2 x H]: two output blocks -- split finer than the single-problem form (128 ways: one workgroup per CU)
                const float* dYs[1] = {dg1};
                const float* Xs[1] = {w.gs0};
                float* dWs[1] = {grad + x.R.out_w1};
                float* dbs[1] = {grad + x.R.out_b1};
                const float* am[1] = {pamax};
                TSD_TRY(launch_wgrad_batch(1, Eo, H, H / 2, dYs, Xs, dWs, dbs, 1, w.wpart, st, h2 ? am : nullptr));
            }
            if (h2) {
                const float* dYs[1] = {dg0};
                const float* Xs[1] = {w.hp};
                float* dWs[1] = {grad + x.R.out_w0};
                float* dbs[1] = {grad + x.R.out_b0};
                const float* am[1] = {pamax + 1};
                TSD_TRY(launch_wgrad_batch(1, Eo, 2 * H, H, dYs, Xs, dWs, dbs, 1, w.wpart, st, am));
            } else {
                TSD_TRY(x.lin_bwd(Eo, 2 * H, H, w.hp, x.R.out_w0, (long)x.R.out_b0, dg0, nullptr, false));
            }
        } else {
            TSD_TRY(x.lin_bwd(Eo, H / 2, 1, w.gs1, x.R.out_w2, (long)x.R.out_b2, ds, w.eB, false, 0, w.g1));   // dg1
            TSD_TRY(x.lin_bwd(Eo, H, H / 2, w.gs0, x.R.out_w1, (long)x.R.out_b1, w.eB, w.eA, false, 0, w.g0)); // dg0
            TSD_TRY(x.lin_bwd(Eo, 2 * H, H, w.hp, x.R.out_w0, (long)x.R.out_b0, w.eA, w.eB, false));           // dhp [Eo,2H]
            // dp (left half) -> dh ; d edge_attr_out (right half) -> the out edges' rows of the attribute gradient
            hipLaunchKernelGGL(copy2d_kernel, dim3(nblk((int64_t)Eo * H)), dim3(256), 0, st, (int64_t)Eo, H, w.eB, 2 * H, dp, H);
            hipLaunchKernelGGL(scatter_rows_kernel, dim3(nblk((int64_t)Eo * H)), dim3(256), 0, st, (int64_t)Eo, H, w.eB + H,
                               2 * H, g.attr_row, PU, PU - Eu, w.d_ea);
        }
        TSD_TRY(launch_row_gather(H, N, g.out, dp, w.h + (size_t)L * NH, w.dh, st));  // dh_i = sum_e dp[umap e] * h[dst e]
    }
    // The node-level layers (N rows) keep their dY per block and take their weight gradients in ONE batched launch
    // after the loop: 3 L problems of N rows each are launch-latency bound one by one (13 + 5 us each, 21 per step).
    const bool batch_wg = (H % 128 == 0);
    const float* dh_cur = w.dh;  // d loss / d h_{l+1}
    if (batch_wg) {
        // MFMA sizes: ONE launch per block (launch_block_bwd: the node chain -- gather + three 16-row GEMMs, the mirror
        // image of the forward's node role -- beside the filter MLP's backward chain) instead of 4 + 3 primitive launches
        float *dagg = w.nA, *dagg_other = w.nB;
        auto wt = [&](int l, size_t off) { return w.pack_t + x.R.layer0 + (size_t)l * x.R.layer_stride + off; };
        auto wt16 = [&](int l, size_t off) { return w.pack_t16 + x.R.layer0 + (size_t)l * x.R.layer_stride + off; };
        if (h2 && Eo == 0) TSD_HIP(hipMemsetAsync(w.amax, 0, (5 * (size_t)L + 6) * sizeof(float), st));
        const tsd_edges none{};
        // (split-f16 step: the node chain keeps the running maxima of its dY tensors for the node-level weight gradients)
        float* namax = w.amax + 2 * L + 5;  // [0] dh_L, [1 + l] dh_l, [1 + L + l] dx2_l, [1 + 2 L + l] dx1_l
        auto nam = [&](int l_dx1, int l_dh, int l_dx2, bool top) {
            NodeAmax m;
            if (!h2) return m;
            if (top) m.in = namax;
            if (l_dx1 >= 0) m.dx1 = namax + 1 + 2 * L + l_dx1;
            if (l_dh >= 0) m.dh = namax + 1 + l_dh;
            if (l_dx2 >= 0) m.dx2 = namax + 1 + L + l_dx2;
            return m;
        };
        TSD_TRY(launch_block_bwd(H, N, 1, 0, g.enc, nullptr, nullptr, w.dh, nullptr, wt(L - 1, x.R.L_lin_w),
                                 wt(L - 1, x.R.L_lin2_w), w.x2 + (size_t)(L - 1) * NH, nullptr, nullptr,
                                 w.dx2s + (size_t)(L - 1) * NH, dagg, 0, none, nullptr, nullptr, nullptr, nullptr, 0.f, 0,
                                 nullptr, nullptr, nullptr, st, nullptr, nam(-1, -1, L - 1, true)));
        for (int l = L - 1; l >= 0; --l) {
            const int lp = l > 0 ? l - 1 : 0;
            TSD_TRY(launch_block_bwd(H, N, 0, l == 0, g.enc, w.Wf + l * EH, dagg, dh_cur, wt(l, x.R.L_lin1_w),
                                     wt(lp, x.R.L_lin_w), wt(lp, x.R.L_lin2_w), w.x2 + (size_t)lp * NH, w.dx1s + l * NH,
                                     w.dhs + l * NH, w.dx2s + (size_t)lp * NH, dagg_other, Eu, g.enc_u, w.x1 + l * NH,
                                     w.f0 + l * EH, h2 ? wt16(l, x.R.L_nn2_w) : wt(l, x.R.L_nn2_w),
                                     h2 ? wt16(l, x.R.L_nn0_w) : wt(l, x.R.L_nn0_w), cfg->conv_cutoff,
                                     cfg->smooth_conv, w.dWfs + l * EH, w.df0s + l * EH, w.d_ea, st,
                                     h2 ? w.amax + 2 * l : nullptr, nam(l, l, l > 0 ? l - 1 : -1, false)));
            dh_cur = w.dhs + l * NH;
            float* t = dagg;
            dagg = dagg_other;
            dagg_other = t;
        }
    }
    for (int l = L - 1; l >= 0 && !batch_wg; --l) {  // sizes without MFMA instances: primitive by primitive
        const size_t o = x.R.layer0 + (size_t)l * x.R.layer_stride;
        float *f0 = w.f0 + l * EH, *fs = w.fs + l * EH, *Wf = w.Wf + l * EH;
        float* hl = w.h + l * NH;
        float *x1 = w.x1 + l * NH, *agg = w.agg + l * NH, *x2 = w.x2 + l * NH, *xs = w.xs + l * NH;
        // h_{l+1} = h_l + lin(ssp(lin2(agg)))
        TSD_TRY(x.lin_bwd(N, H, H, xs, o + x.R.L_lin_w, (long)(o + x.R.L_lin_b), w.dh, w.nB, false, 1, x2));  // dx2
        TSD_TRY(x.lin_bwd(N, H, H, agg, o + x.R.L_lin2_w, (long)(o + x.R.L_lin2_b), w.nB, w.nA, false));     // dagg
        // agg = aggregate(x1, Wf): symmetric edge set and filter => the adjoint w.r.t. x1 is the same gather of dagg
        TSD_TRY(tsd_cfconv_aggregate(H, N, g.enc.row_ptr, g.enc.dst, g.enc.umap, Wf, w.nA, w.nC, stream));    // dx1
        if (Eu > 0) {
            TSD_TRY(launch_aggregate_bwd_filter(H, Eu, g.enc_u, w.nA, x1, w.eA, 1, cfg->conv_cutoff, cfg->smooth_conv, st));
            TSD_TRY(x.lin_bwd(Eu, H, H, fs, o + x.R.L_nn2_w, (long)(o + x.R.L_nn2_b), w.eA, w.eB, false, 1, f0));  // df0
            TSD_TRY(x.lin_bwd(Eu, H, H, w.ea, o + x.R.L_nn0_w, (long)(o + x.R.L_nn0_b), w.eB, w.d_ea, true));
        }
        TSD_TRY(x.lin_bwd(N, H, H, hl, o + x.R.L_lin1_w, -1, w.nC, w.dh, true));  // dh += dx1 W_lin1 (residual keeps dh)
    }
    // split-f16 step: the node-embedding gradients need only dh_0, which is final here -- they go to the side lane, beside the
    // batched weight-gradient launches below (nA / nB are free again, the lane has its own scratch)
    SideLane* side = nullptr;
    if (h2 && batch_wg && !(batch->reserved & 64)) {
        TSD_TRY(side_lane(&side));
        TSD_HIP(hipEventRecord(side->fork, st));
        TSD_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
        TSD_TRY(node_embed_grads(x, dh_cur, atom_type, side->s, w.scratch2));
    }
    if (batch_wg) {
        std::vector<const float*> dYs, Xs;
        std::vector<float*> dWs, dbs;
        for (int l = 0; l < L; ++l) {
            const size_t o = x.R.layer0 + (size_t)l * x.R.layer_stride;
            const float* dh_up = l == L - 1 ? w.dh : w.dhs + (size_t)(l + 1) * NH;  // d loss / d h_{l+1}
            dYs.push_back(dh_up);              Xs.push_back(w.xs + l * NH);  dWs.push_back(grad + o + x.R.L_lin_w);  dbs.push_back(grad + o + x.R.L_lin_b);
            dYs.push_back(w.dx2s + l * NH);    Xs.push_back(w.agg + l * NH); dWs.push_back(grad + o + x.R.L_lin2_w); dbs.push_back(grad + o + x.R.L_lin2_b);
            dYs.push_back(w.dx1s + l * NH);    Xs.push_back(w.h + l * NH);   dWs.push_back(grad + o + x.R.L_lin1_w); dbs.push_back(nullptr);
        }
        std::vector<const float*> namx;
        for (int l = 0; l < L; ++l) {  // (order as above: dh_{l+1}, dx2_l, dx1_l; xs / agg / h passed the forward's range check)
            const float* na = w.amax + 2 * L + 5;
            namx.push_back(l == L - 1 ? na : na + 1 + (l + 1));
            namx.push_back(na + 1 + L + l);
            namx.push_back(na + 1 + 2 * L + l);
        }
        TSD_TRY(launch_wgrad_batch((int)dYs.size(), N, H, H, dYs.data(), Xs.data(), dWs.data(), dbs.data(), 1, w.wpart, st,
                                   h2 ? namx.data() : nullptr));
        if (Eu > 0) {  // the filter MLPs of all blocks: 2 L problems of Eu rows
            dYs.clear(), Xs.clear(), dWs.clear(), dbs.clear();
            std::vector<const float*> amx;
            for (int l = 0; l < L; ++l) {
                const size_t o = x.R.layer0 + (size_t)l * x.R.layer_stride;
                dYs.push_back(w.dWfs + l * EH); Xs.push_back(w.fs + l * EH); dWs.push_back(grad + o + x.R.L_nn2_w); dbs.push_back(grad + o + x.R.L_nn2_b);
                dYs.push_back(w.df0s + l * EH); Xs.push_back(w.ea);          dWs.push_back(grad + o + x.R.L_nn0_w); dbs.push_back(grad + o + x.R.L_nn0_b);
                amx.push_back(w.amax + 2 * l);
                amx.push_back(w.amax + 2 * l + 1);
            }
            // (split-f16 step: fs / ea passed the forward's range check, the dY tensors carry their block launch's maxima)
            TSD_TRY(launch_wgrad_batch((int)dYs.size(), Eu, H, H, dYs.data(), Xs.data(), dWs.data(), dbs.data(), 1, w.wpart, st,
                                       h2 ? amx.data() : nullptr));
        }
    }
    // every gradient of the interaction blocks (tsd_train_grad_buckets: 83 % of the flat vector) is final here: a
    // data-parallel caller starts their all-reduce on another stream behind this event, beside the embedding's backward
    // chain and the node-embedding gradients below
    if (blocks_done_event != nullptr) TSD_HIP(hipEventRecord((hipEvent_t)blocks_done_event, st));
    const float* dz = dh_cur;  // d loss / d h_0
    if (batch_wg) {
        TSD_TRY(embed_bwd_fused(x, g, w.d_ea, h2, side));
    } else {
        if (Eu > 0) TSD_TRY(embed_bwd(x, g.enc_u, Eu, w.emb, w.d_ea));
        if (Ed > 0) TSD_TRY(embed_bwd(x, g.diff_u, Ed, embed_rows(w.emb, (size_t)Eu, (size_t)H), w.d_ea + (size_t)Eu * H));
    }
    // node embedding: dz = dh
    // (d(Wf r) rows then d(Wf p) rows in nA [2 N, H/2], against featR | featP [2 N, F]: atom_feat_embedding's weight
    // gradient is one problem of 2 N rows)
    if (side == nullptr) TSD_TRY(node_embed_grads(x, dz, atom_type, st, w.scratch));
    if (side != nullptr) {  // the caller's stream continues behind the side lane's last launch
        TSD_HIP(hipEventRecord(side->join, side->s));
        TSD_HIP(hipStreamWaitEvent(st, side->join, 0));
    }
    TSD_LAUNCH_CHECK("train_backward");
    return TSD_OK;
}

}  // extern "C"

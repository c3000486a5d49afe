// api.hip -- the extern "C" surface of libtsdiff_hip.so (declared in include/tsdiff_hip.h) and the
// orchestration of one score-network forward / one sampling step / the device-resident loop.
#include <stdarg.h>
#include <stdlib.h>

#include "common.hpp"

namespace tsd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return TSD_OK;
    set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
    return TSD_ERR_HIP;
}

// launchers implemented in the kernel translation units
int launch_edge_embed(const tsd_model_cfg&, const float*, int, tsd_edges, float*, hipStream_t);
int launch_cfconv_layer(const tsd_model_cfg&, const float*, int, int, tsd_edges, const float*, const float*,
                        float*, float*, hipStream_t);
int launch_node_update(const tsd_model_cfg&, const float*, int, int, int, const int32_t*, const float*,
                       const float*, float*, float*, hipStream_t);
int launch_node_lin1(const tsd_model_cfg&, const float*, int, int, const float*, float*, hipStream_t);
int launch_pair_output(const tsd_model_cfg&, const float*, int, tsd_edges, const float*, const float*,
                       const int32_t*, float*, int, size_t, size_t, size_t, hipStream_t, const float*, size_t);
size_t raw_weight_floats(const tsd_model_cfg&);
int launch_pack_weights(const tsd_model_cfg&, const float*, float*, hipStream_t);
int launch_topology(int, int, int, int64_t, const int32_t*, const int32_t*, const int64_t*, const int64_t*, int,
                    int, int32_t*, int32_t*, uint16_t*, int32_t*, hipStream_t);
size_t geometry_scratch_ints(int, int);
int launch_geometry(const tsd_model_cfg&, int, int, int, const float*, const int32_t*, const int32_t*,
                    const int32_t*, const uint16_t*, tsd_geometry, hipStream_t);
int launch_geometry_count(const tsd_model_cfg&, int, const float*, const int32_t*, const int32_t*, const int32_t*,
                          const uint16_t*, tsd_geometry, hipStream_t);
int launch_geometry_lists(const tsd_model_cfg&, int, int, const float*, const int32_t*, const int32_t*,
                          const int32_t*, const uint16_t*, tsd_geometry, int32_t*, hipStream_t, bool);
int launch_step_post(const tsd_model_cfg&, int, int, int, int, int, const int32_t*, const int32_t*, const uint16_t*,
                     tsd_geometry, const float*, const float*, const float*, float, float, float*, float*, int32_t*,
                     const int32_t*, hipStream_t);
int launch_filter_gen(const tsd_model_cfg&, const float*, int, tsd_edges, const float*, float*, int, int,
                      hipStream_t);
int launch_edge_embed2(const tsd_model_cfg&, const float*, int, tsd_edges, float*, int, tsd_edges, float*, int,
                       size_t, hipStream_t, const UmapRole*);
int launch_layer_combo(const tsd_model_cfg&, const float*, int, int, tsd_edges, const float*, const float*,
                       const float*, float*, float*, int, int, int, int, tsd_edges, const float*, float*, int, size_t,
                       size_t, size_t, hipStream_t, const ComboPre*, size_t);
int filter_tiles_per_layer(int);
int launch_node_embed(const tsd_model_cfg&, const float*, int, const int64_t*, const int64_t*, const int64_t*,
                      float*, hipStream_t);
int launch_cfconv_aggregate(int, int, const int32_t*, const int32_t*, const int32_t*, const float*,
                            const float*, float*, hipStream_t);
int launch_eq_transform_atomic(int, int64_t, const float*, const float*, const int64_t*, const float*, float*,
                               hipStream_t);
int launch_eq_transform_rows(int, const float*, const int32_t*, const int32_t*, const int32_t*, tsd_edges,
                             const int32_t*, const float*, float*, hipStream_t);
int launch_ensemble_mean(int, int, tsd_edges, const float*, float*, hipStream_t);
int launch_sampler_step(int, int, int, const int32_t*, const float*, const float*, const float*, float, float,
                        float*, float*, int32_t*, const int32_t*, hipStream_t);

extern int g_filter_rows;
extern int g_combo_cols;
extern int g_node_run;
extern int g_combo_prefetch;

static int check_cfg(const tsd_model_cfg* c) {
    TSD_REQUIRE(c != nullptr, "cfg is null");
    TSD_REQUIRE(hidden_supported(c->hidden), "hidden=%d unsupported (64/128/256)", c->hidden);
    TSD_REQUIRE(c->num_convs >= 1 && c->num_convs <= 64, "num_convs=%d out of range", c->num_convs);
    TSD_REQUIRE(c->feat_dim >= 1, "feat_dim=%d", c->feat_dim);
    TSD_REQUIRE(c->edge_order >= 1 && c->edge_order <= 7 && c->pred_edge_order >= 1 && c->pred_edge_order <= 7,
                "edge orders (%d,%d) outside 1..7", c->edge_order, c->pred_edge_order);
    return TSD_OK;
}

struct Workspace {
    // every array holds one block per checkpoint (stride_* floats apart): the M forwards of an ensemble
    // run in the SAME launches (grid.y = checkpoint), which removes the tile quantisation of batch-100
    // launches (508 workgroups on 256 CUs) and the per-checkpoint launch boundaries
    float *ea;   // [M][P, H]: rows 0..P/2-1 enc_u edges, rows P/2.. separately embedded (diff_u) out edges
    float *wf;   // [M][L, P/2, H]: CFConv filters of every layer on the undirected enc list
    float *h, *x1, *x1b;  // [M][N, H]
    float *agg;  // [N, H] (piecewise path only)
    float *pre;  // [M][P/2, H]: node-independent half of the pair MLP's first layer (ComboPre)
    size_t stride_ea, stride_wf, stride_nh, stride_pre;
    size_t total;
};

static Workspace carve(const tsd_model_cfg& c, int N, int P, int M, float* base) {
    Workspace w;
    const size_t H = c.hidden, PU = (size_t)P / 2;
    size_t o = 0;
    auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += n; return p; };
    auto pad = [](size_t n) { return (n + 63) & ~size_t(63); };
    w.stride_ea = pad(2 * PU * H);
    w.stride_wf = pad((size_t)c.num_convs * PU * H);
    w.stride_nh = pad((size_t)N * H);
    w.ea = take(w.stride_ea * M);
    w.wf = take(w.stride_wf * M);
    w.h = take(w.stride_nh * M);
    w.x1 = take(w.stride_nh * M);
    w.x1b = take(w.stride_nh * M);
    w.agg = take(w.stride_nh);
    w.stride_pre = pad(PU * H);
    w.pre = take(w.stride_pre * M);
    w.total = o;
    return w;
}

// TSDIFF_FORWARD=serial selects the piecewise path (filter_gen for all layers, then aggregate + node_update
// per block, checkpoints one after the other) -- the A/B baseline of the fused per-block launches.
// (A two-stream variant, node chain || next block's filters joined by events, measured 1.07 ms/step under
// hipGraph replay against 0.92 serial and was removed in favour of the in-kernel fusion, kernels_combo.hip.)
// TSDIFF_PAIR_PRE=0 keeps the whole pair MLP in pair_output_kernel (A/B of the ComboPre role)
static bool pre_role_enabled() {
    static int cached = -1;
    if (cached < 0) {
        const char* env = getenv("TSDIFF_PAIR_PRE");
        cached = (env && env[0] == '0') ? 0 : 1;
    }
    return cached == 1 && g_filter_rows != 64 && g_combo_cols != 64;
}

static bool use_fused_path() {
    static int cached = -1;
    if (cached < 0) {
        const char* env = getenv("TSDIFF_FORWARD");
        cached = (env && env[0] == 's') ? 0 : 1;
    }
    return cached == 1;
}

// One forward per checkpoint on the current positions.  Every per-edge MLP runs on the UNDIRECTED
// lists (half the edges of the reference's directed list: edge_attr, W and edge_inv are symmetric);
// the directed CSR list only drives the aggregation and eq_transform through `umap`.
// counts_ready: the per-row member counts are already in geo.scratch (written by the previous step's
// step_post_kernel); advance: device step counter bumped by the scan kernel (sampling loop only).
static int forward_impl(const tsd_model_cfg& c, const tsd_batch& b, const float* pos, hipStream_t st,
                        bool counts_ready = false, int32_t* advance = nullptr) {
    const int N = b.num_nodes, P = b.num_pairs, M = b.num_models;
    const int PU = P / 2, L = c.num_convs;
    const size_t H = c.hidden;
    const tsd_geometry& g = b.geo;
    int r;
    if (!counts_ready) {
        if ((r = launch_geometry_count(c, N, pos, b.graph_ptr, b.node_graph, b.pair_ptr, b.pair_code, g, st))) return r;
    }
    // fused path: the directed-edge -> undirected-pair map is not needed before the first block launch, so it
    // runs as an extra role of the edge-embedding launch instead of a launch of its own on the critical path
    const bool fused = use_fused_path();
    if ((r = launch_geometry_lists(c, N, P, pos, b.graph_ptr, b.node_graph, b.pair_ptr, b.pair_code, g, advance, st,
                                   fused)))
        return r;
    const Workspace w = carve(c, N, P, M, b.workspace);
    const size_t wfloats = weight_layout(c).total;
    if (fused) {
        // one launch per interaction block: node chain of block l || filter GEMMs of block l+1;
        // all M checkpoints in the same launches (grid.y)
        const float* W = b.weights;
        UmapRole um{};
        um.g = g;
        um.graph_ptr = b.graph_ptr;
        um.node_graph = b.node_graph;
        um.pair_ptr = b.pair_ptr;
        um.P = P;
        if ((r = launch_edge_embed2(c, W, PU, g.enc_u, w.ea, PU, g.diff_u, w.ea + (size_t)PU * H, M, w.stride_ea, st,
                                    &um)))
            return r;
        // block 0 reads z (residual input) and x1_0 = lin1_0(z) straight from the per-batch arrays -- both are
        // pos independent (computed at bind time) -- so no per-step copy of z and no lin1 launch
        if (w.stride_nh != (size_t)N * H) {
            set_error("internal: node stride mismatch");
            return TSD_ERR_INVALID;
        }
        // The filter tiles of all blocks form one queue (they depend on the geometry only); launch j takes block
        // j's tiles, so that they are complete before block j's node chain runs in launch j+1.  (Re-cutting the
        // queue into whole chip rounds -- 512,512,256,... tiles instead of 7 x 408 at batch 100 -- measured
        // slower, 0.597 vs 0.534 ms/step: 408 filter + 100 node workgroups already are two full rounds of 256.)
        const int tpl = filter_tiles_per_layer(PU);
        const long total = (long)L * tpl;
        long cum = 0;
        const float* xin = b.x1_0;
        float* xout = w.x1;
        // the last launch has no filter tiles left: its free CUs compute the node-independent half of the pair
        // MLP's first layer (ComboPre), which pair_output_kernel then only completes
        const WeightLayout WL = weight_layout(c);
        ComboPre pre{};
        pre.tiles = (PU + TSD_EDGE_TILE - 1) / TSD_EDGE_TILE;
        pre.e = g.out_u;
        pre.edge_attr = w.ea;
        pre.attr_row = g.attr_row;
        pre.w0b = W + WL.out_w0 + H * H;  // packed [k/4][out][k%4]: the k >= H half is contiguous
        pre.b0 = W + WL.out_b0;
        pre.out = w.pre;
        // (only when the node chain of the last block leaves most of the chip idle: at batch 100 it occupies ~100
        // of the 256 CUs; with an ensemble or a large batch the launch is full and the extra role only adds work:
        // C2 0.518 -> 0.511 ms/step, C5 51.1 -> 51.6, M = 8 3.10 -> 3.13)
        const bool use_pre = pre_role_enabled() && (long)((N + TSD_NODE_TILE - 1) / TSD_NODE_TILE) * M <= 256;
        for (int j = 0; j <= L; ++j) {
            const long n = (j < L) ? tpl : total - cum;
            const int layer = j == 0 ? -2 : j - 1;
            if ((r = launch_layer_combo(c, W, layer, N, g.enc, layer >= 0 ? w.wf + (size_t)layer * PU * H : nullptr, xin,
                                        layer == 0 ? b.z : w.h, w.h, xout, 0, (int)cum, (int)n, PU, g.enc_u, w.ea, w.wf,
                                        M, w.stride_nh, w.stride_ea, w.stride_wf, st,
                                        (use_pre && j == L) ? &pre : nullptr, w.stride_pre)))
                return r;
            cum += n;
            if (layer >= 0) {
                xin = xout;
                xout = (xout == w.x1) ? w.x1b : w.x1;
            }
        }
        return launch_pair_output(c, W, PU, g.out_u, w.h, w.ea, g.attr_row, b.edge_inv_u, M, w.stride_nh, w.stride_ea,
                                  (size_t)PU, st, use_pre ? w.pre : nullptr, w.stride_pre);
    }
    for (int m = 0; m < M; ++m) {  // piecewise path
        const float* W = b.weights + (size_t)m * wfloats;
        if ((r = launch_edge_embed(c, W, PU, g.enc_u, w.ea, st))) return r;
        if ((r = launch_filter_gen(c, W, PU, g.enc_u, w.ea, w.wf, 0, L, st))) return r;
        TSD_HIP(hipMemcpyAsync(w.h, b.z + (size_t)m * N * H, (size_t)N * H * sizeof(float),
                               hipMemcpyDeviceToDevice, st));
        if ((r = launch_node_lin1(c, W, 0, N, w.h, w.x1, st))) return r;
        for (int l = 0; l < L; ++l) {
            if ((r = launch_cfconv_aggregate(c.hidden, N, g.enc.row_ptr, g.enc.dst, g.enc.umap,
                                             w.wf + (size_t)l * PU * H, w.x1, w.agg, st)))
                return r;
            if ((r = launch_node_update(c, W, l, (l + 1 < L) ? l + 1 : -1, N, nullptr, w.agg, nullptr, w.h, w.x1,
                                        st)))
                return r;
        }
        if ((r = launch_edge_embed(c, W, PU, g.diff_u, w.ea + (size_t)PU * H, st))) return r;
        if ((r = launch_pair_output(c, W, PU, g.out_u, w.h, w.ea, g.attr_row, b.edge_inv_u + (size_t)m * PU, 1, 0, 0,
                                    0, st, nullptr, 0)))
            return r;
    }
    return TSD_OK;
}

// one sampling step of the device-resident loop: lists (from the counts of the previous step's tail) ->
// M forwards -> fused tail (mean, eq_transform, update, centre, next step's counts)
static int step_impl(const tsd_model_cfg& c, const tsd_batch& b, int kind, const float* coefs,
                     const float* noises, float clip, float clip_pos, float* pos, float* traj,
                     int32_t* status, int32_t* step_ctr, bool advance, hipStream_t st) {
    int r;
    if ((r = forward_impl(c, b, pos, st, true, advance ? step_ctr : nullptr))) return r;
    return launch_step_post(c, kind, b.num_nodes, b.num_graphs, b.num_models, b.num_pairs, b.graph_ptr, b.pair_ptr,
                            b.pair_code, b.geo, b.edge_inv_u, noises, coefs, clip, clip_pos, pos, traj, status,
                            step_ctr, st);
}

}  // namespace tsd

using namespace tsd;

extern "C" {

int tsd_set_filter_tile(int32_t rows) {
    TSD_REQUIRE(rows == 0 || rows == 32 || rows == 64, "filter tile rows must be 0 (auto), 32 or 64");
    g_filter_rows = rows;
    return TSD_OK;
}

int tsd_set_combo_prefetch(int32_t kblocks) {
    TSD_REQUIRE(kblocks == 0 || kblocks == 4 || kblocks == 8, "prefetch chunk must be 0 (default), 4 or 8 k-blocks");
    g_combo_prefetch = kblocks;
    return TSD_OK;
}

int tsd_set_node_run(int32_t run) {
    TSD_REQUIRE(run >= 1 && run <= 64, "node tiles per XCD run must be in 1..64");
    g_node_run = run;
    return TSD_OK;
}

int tsd_set_combo_cols(int32_t cols) {
    TSD_REQUIRE(cols == 0 || cols == 32 || cols == 64, "columns per wave must be 0 (auto), 32 or 64");
    g_combo_cols = cols;
    return TSD_OK;
}

const char* tsd_version(void) { return "tsdiff_hip 0.1 (gfx950, fp32 MFMA)"; }
const char* tsd_last_error(void) { return g_err; }

size_t tsd_raw_weight_floats(const tsd_model_cfg* cfg) {
    if (check_cfg(cfg)) return 0;
    return raw_weight_floats(*cfg);
}
size_t tsd_packed_weight_floats(const tsd_model_cfg* cfg) {
    if (check_cfg(cfg)) return 0;
    return weight_layout(*cfg).total;
}
int tsd_pack_weights(const tsd_model_cfg* cfg, const float* raw, float* packed, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(raw && packed, "null weight pointer");
    return launch_pack_weights(*cfg, raw, packed, (hipStream_t)stream);
}

int tsd_topology_build(int32_t num_nodes, int32_t num_graphs, int32_t num_pairs, int64_t num_bonds,
                       const int32_t* graph_ptr, const int32_t* pair_base, const int64_t* bond_index,
                       const int64_t* bond_type, int32_t max_order, int32_t max_graph_nodes_host,
                       int32_t* node_graph, int32_t* pair_ptr, uint16_t* pair_code, int32_t* status,
                       void* stream) {
    TSD_REQUIRE(num_nodes >= 0 && num_graphs >= 0 && num_pairs >= 0 && num_bonds >= 0, "negative size");
    TSD_REQUIRE(graph_ptr && pair_base && node_graph && pair_ptr && status, "null pointer");
    TSD_REQUIRE(num_pairs == 0 || pair_code, "null pair_code");
    TSD_REQUIRE(num_bonds == 0 || (bond_index && bond_type), "null bond arrays");
    return launch_topology(num_nodes, num_graphs, num_pairs, num_bonds, graph_ptr, pair_base, bond_index,
                           bond_type, max_order, max_graph_nodes_host, node_graph, pair_ptr, pair_code, status,
                           (hipStream_t)stream);
}

size_t tsd_geometry_scratch_ints(int32_t num_nodes, int32_t num_pairs) {
    return geometry_scratch_ints(num_nodes, num_pairs);
}

int tsd_geometry_build(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_graphs, int32_t num_pairs,
                       const float* pos, const int32_t* graph_ptr, const int32_t* node_graph,
                       const int32_t* pair_ptr, const uint16_t* pair_code, tsd_geometry geo, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(pos && graph_ptr && node_graph && pair_ptr && geo.scratch, "null pointer");
    TSD_REQUIRE(geo.enc.count && geo.enc.row_ptr && geo.out.count && geo.out.row_ptr && geo.enc_u.count &&
                    geo.enc_u.row_ptr && geo.out_u.count && geo.out_u.row_ptr && geo.diff_u.count &&
                    geo.diff_u.row_ptr && geo.attr_row && geo.pair2out && geo.pair2u && geo.enc.umap &&
                    geo.out.umap,
                "null edge list");
    TSD_REQUIRE(num_pairs % 2 == 0, "num_pairs must be sum n(n-1)");
    return launch_geometry(*cfg, num_nodes, num_graphs, num_pairs, pos, graph_ptr, node_graph, pair_ptr,
                           pair_code, geo, (hipStream_t)stream);
}

int tsd_node_embed(const tsd_model_cfg* cfg, const float* w, int32_t num_nodes, const int64_t* atom_type,
                   const int64_t* r_feat, const int64_t* p_feat, float* z, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    return launch_node_embed(*cfg, w, num_nodes, atom_type, r_feat, p_feat, z, (hipStream_t)stream);
}

int tsd_edge_embed(const tsd_model_cfg* cfg, const float* w, int32_t capacity, tsd_edges edges,
                   float* edge_attr, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    return launch_edge_embed(*cfg, w, capacity, edges, edge_attr, (hipStream_t)stream);
}

int tsd_node_lin1(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t num_nodes, const float* h,
                  float* x1, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(layer >= 0 && layer < cfg->num_convs, "layer %d out of range", layer);
    return launch_node_lin1(*cfg, w, layer, num_nodes, h, x1, (hipStream_t)stream);
}

int tsd_cfconv_layer(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t capacity, tsd_edges enc,
                     const float* edge_attr, const float* x1, float* agg, float* part, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(layer >= 0 && layer < cfg->num_convs, "layer %d out of range", layer);
    return launch_cfconv_layer(*cfg, w, layer, capacity, enc, edge_attr, x1, agg, part, (hipStream_t)stream);
}

int tsd_filter_gen(const tsd_model_cfg* cfg, const float* w, int32_t capacity, tsd_edges edges,
                   const float* edge_attr, float* Wf, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(w && edge_attr && Wf && edges.count && edges.dist, "null pointer");
    return launch_filter_gen(*cfg, w, capacity, edges, edge_attr, Wf, 0, cfg->num_convs, (hipStream_t)stream);
}

int tsd_interaction_block(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t num_nodes,
                          tsd_edges enc, const float* Wf_layer, const float* x1_in, float* h, float* x1_out,
                          int32_t filter_layer, int32_t capacity_u, tsd_edges enc_u, const float* edge_attr,
                          float* Wf_out, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(w && (layer == -2 || (h && x1_out)), "null pointer");
    TSD_REQUIRE(layer >= -2 && layer < cfg->num_convs && filter_layer >= -1 && filter_layer < cfg->num_convs,
                "layer out of range");
    TSD_REQUIRE(layer < 0 || (Wf_layer && x1_in && enc.row_ptr && enc.dst && enc.umap && x1_in != x1_out),
                "node role needs Wf_layer, x1_in != x1_out and the directed enc list");
    TSD_REQUIRE(filter_layer < 0 || (edge_attr && Wf_out && enc_u.count && enc_u.dist), "filter role: null pointer");
    return launch_layer_combo(*cfg, w, layer, num_nodes, enc, Wf_layer, x1_in, nullptr, h, x1_out,
                              filter_layer < 0 ? 0 : filter_layer, 0,
                              filter_layer < 0 ? 0 : filter_tiles_per_layer(capacity_u), capacity_u, enc_u, edge_attr,
                              Wf_out, 1, 0, 0, 0, (hipStream_t)stream, nullptr, 0);
}

int tsd_cfconv_aggregate(int32_t hidden, int32_t num_nodes, const int32_t* row_ptr, const int32_t* dst,
                         const int32_t* umap, const float* W, const float* x1, float* out, void* stream) {
    TSD_REQUIRE(row_ptr && dst && W && x1 && out, "null pointer");
    return launch_cfconv_aggregate(hidden, num_nodes, row_ptr, dst, umap, W, x1, out, (hipStream_t)stream);
}

int tsd_node_update(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t next_layer,
                    int32_t num_nodes, const int32_t* enc_row_ptr, const float* agg, const float* part,
                    float* h, float* x1, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(layer >= 0 && layer < cfg->num_convs && next_layer < cfg->num_convs, "layer out of range");
    return launch_node_update(*cfg, w, layer, next_layer, num_nodes, enc_row_ptr, agg, part, h, x1,
                              (hipStream_t)stream);
}

int tsd_pair_output(const tsd_model_cfg* cfg, const float* w, int32_t capacity, tsd_edges out, const float* h,
                    const float* edge_attr, const int32_t* attr_row, float* edge_inv, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    return launch_pair_output(*cfg, w, capacity, out, h, edge_attr, attr_row, edge_inv, 1, 0, 0, 0,
                              (hipStream_t)stream, nullptr, 0);
}

int tsd_eq_transform(int32_t num_nodes, int64_t num_edges, const float* score_d, const float* pos,
                     const int64_t* edge_index, const float* edge_length, float* score_pos, void* stream) {
    TSD_REQUIRE(num_edges == 0 || (score_d && pos && edge_index && edge_length && score_pos), "null pointer");
    return launch_eq_transform_atomic(num_nodes, num_edges, score_d, pos, edge_index, edge_length, score_pos,
                                      (hipStream_t)stream);
}

size_t tsd_forward_workspace_floats(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_pairs,
                                    int32_t num_models) {
    if (check_cfg(cfg)) return 0;
    return carve(*cfg, num_nodes, num_pairs, num_models < 1 ? 1 : num_models, nullptr).total;
}

int tsd_score_forward(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* pos, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(batch && pos, "null pointer");
    TSD_REQUIRE(batch->num_models >= 1, "num_models=%d", batch->num_models);
    TSD_REQUIRE(batch->z && batch->x1_0 && batch->weights && batch->workspace && batch->edge_inv_u, "null batch pointer");
    return forward_impl(*cfg, *batch, pos, (hipStream_t)stream);
}

int tsd_ensemble_mean(int32_t num_models, int32_t num_pairs, tsd_edges out, const float* edge_inv_u,
                      float* edge_inv, void* stream) {
    TSD_REQUIRE(num_models >= 1 && out.count && out.umap && edge_inv_u && edge_inv, "bad argument");
    return launch_ensemble_mean(num_models, num_pairs, out, edge_inv_u, edge_inv, (hipStream_t)stream);
}

int tsd_eq_transform_rows(int32_t num_nodes, const float* pos, const int32_t* pair_ptr,
                          const int32_t* graph_ptr, const int32_t* node_graph, tsd_edges out,
                          const int32_t* pair2out, const float* score_d, float* score_pos, void* stream) {
    return launch_eq_transform_rows(num_nodes, pos, pair_ptr, graph_ptr, node_graph, out, pair2out, score_d,
                                    score_pos, (hipStream_t)stream);
}

int tsd_sampler_step(int32_t kind, int32_t num_nodes, int32_t num_graphs, const int32_t* graph_ptr,
                     const float* score_pos, const float* noise, const float* coefs, float clip, float clip_pos,
                     float* pos, int32_t* status, void* stream) {
    TSD_REQUIRE(kind == 0 || kind == 1, "kind=%d (0 = ld, 1 = ddpm)", kind);
    return launch_sampler_step(kind, num_nodes, num_graphs, graph_ptr, score_pos, noise, coefs, clip, clip_pos,
                               pos, nullptr, status, nullptr, (hipStream_t)stream);
}

int tsd_sampler_run(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t kind, int32_t n_steps,
                    const float* coefs, const float* noises, float clip, float clip_pos, float* pos, float* traj,
                    int32_t* status, int32_t use_graph, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(batch && coefs && noises && pos && status, "null pointer");
    TSD_REQUIRE(kind == 0 || kind == 1, "kind=%d (0 = ld, 1 = ddpm)", kind);
    TSD_REQUIRE(n_steps >= 0, "n_steps=%d", n_steps);
    hipStream_t st = (hipStream_t)stream;
    if (n_steps == 0) return TSD_OK;
    int32_t* step_ctr = status + 1;  // status[1]: device-side step counter (offsets into coefs/noises/traj)
    const tsd_batch& b = *batch;
    TSD_HIP(hipMemsetAsync(step_ctr, 0, sizeof(int32_t), st));
    if ((r = launch_geometry_count(*cfg, b.num_nodes, pos, b.graph_ptr, b.node_graph, b.pair_ptr, b.pair_code, b.geo,
                                   st)))
        return r;
    // step 0 runs eagerly (also performs every one-time function-attribute set-up outside capture)
    if ((r = step_impl(*cfg, b, kind, coefs, noises, clip, clip_pos, pos, traj, status, step_ctr, false, st))) return r;
    if (n_steps == 1) return TSD_OK;
    if (!use_graph) {
        for (int k = 1; k < n_steps; ++k)
            if ((r = step_impl(*cfg, b, kind, coefs, noises, clip, clip_pos, pos, traj, status, step_ctr, true, st)))
                return r;
        return TSD_OK;
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    TSD_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    r = step_impl(*cfg, b, kind, coefs, noises, clip, clip_pos, pos, traj, status, step_ctr, true, st);
    hipError_t ce = hipStreamEndCapture(st, &graph);
    if (r) {
        if (graph) (void)hipGraphDestroy(graph);
        return r;
    }
    TSD_HIP(ce);
    hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (ie != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return check_hip(ie, "hipGraphInstantiate");
    }
    for (int k = 1; k < n_steps && r == TSD_OK; ++k) r = check_hip(hipGraphLaunch(exec, st), "hipGraphLaunch");
    // the exec object must outlive its launches: wait for the stream before destroying it
    hipError_t se = hipStreamSynchronize(st);
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    if (r) return r;
    return check_hip(se, "hipStreamSynchronize");
}

}  // extern "C"

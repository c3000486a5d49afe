// api.hip -- the extern "C" surface of libtsdiff_hip.so (declared in include/tsdiff_hip.h) and the
// orchestration of one score-network forward / one sampling step / the device-resident loop.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>

#include "common.hpp"

namespace tsd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_roctx_push = nullptr;  // written once (call_once), read-only afterwards
roctx_pop_fn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
}  // namespace
TraceRange::TraceRange(const char* name) {
    std::call_once(g_roctx_once, [] {
        void* push = dlsym(RTLD_DEFAULT, "roctxRangePushA");
        void* pop = dlsym(RTLD_DEFAULT, "roctxRangePop");
        if (push && pop) {
            g_roctx_push = (roctx_push_fn)push;
            g_roctx_pop = (roctx_pop_fn)pop;
        }
    });
    on = g_roctx_push != nullptr;
    if (on) g_roctx_push(name);
}
TraceRange::~TraceRange() {
    if (on) g_roctx_pop();
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
                     tsd_geometry, const float*, float, float, float*, tsd_sampler_state*, hipStream_t);
int launch_set_run_args(tsd_sampler_state*, const tsd_run_args&, hipStream_t);
size_t typed_tiles_capacity(int);
int launch_typed_tiles_build(const tsd_model_cfg&, int, int, const int32_t*, const int32_t*, const int32_t*,
                             const uint16_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*,
                             int32_t*, int32_t*, int32_t*, int32_t*, hipStream_t);
int launch_bucket_weights(const tsd_model_cfg&, const float*, int, const int32_t*, float*, hipStream_t);
int launch_typed_embed(const tsd_model_cfg&, const float*, const tsd_batch&, const float*, float*, int, size_t,
                       hipStream_t, const UmapRole*, const EmbedFuse0*, Prec);
bool step_tail_supported(int, int, int);
int launch_step_tail_reset(int, tsd_geometry, hipStream_t);
int launch_step_tail(const tsd_model_cfg&, int, int, int, int, int, int, const int32_t*, const int32_t*, const uint16_t*,
                     tsd_geometry, const float*, float, float, float*, tsd_sampler_state*, hipStream_t);
int launch_philox_normal(uint64_t, uint64_t, int64_t, float*, hipStream_t);
int launch_filter_gen(const tsd_model_cfg&, const float*, int, tsd_edges, const float*, float*, int, int,
                      hipStream_t);
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

static int check_cfg(const tsd_model_cfg* c) {
    TSD_REQUIRE(c != nullptr, "cfg is null");
    TSD_REQUIRE(hidden_supported(c->hidden), "hidden=%d unsupported (64/128/256)", c->hidden);
    TSD_REQUIRE(c->num_convs >= 1 && c->num_convs <= 64, "num_convs=%d out of range", c->num_convs);
    TSD_REQUIRE(c->feat_dim >= 1, "feat_dim=%d", c->feat_dim);
    TSD_REQUIRE(c->edge_order >= 1 && c->edge_order <= 7 && c->pred_edge_order >= 1 && c->pred_edge_order <= 7,
                "edge orders (%d,%d) outside 1..7", c->edge_order, c->pred_edge_order);
    return TSD_OK;
}

// Capacity: the per-edge matrices are indexed row * H with 64-bit arithmetic in the kernels, but tile and row
// bookkeeping is 32-bit; the largest batch any kernel has been run and checked at is configs[4] (P H = 1.06e9).
// Everything at or past 2^31 elements of one [P, H] matrix is refused up front (TSD_ERR_UNSUPPORTED) instead of
// trusted: split the batch (graphs never interact; tsdiff_amd.distributed shards them).
static bool capacity_ok(const tsd_model_cfg& c, int64_t N, int64_t P) {
    const int64_t lim = (int64_t)1 << 31;
    return P * c.hidden < lim && N * c.hidden < lim;
}
#define TSD_CAPACITY(c, N, P)                                                                                       \
    do {                                                                                                            \
        if (!capacity_ok((c), (N), (P))) {                                                                          \
            set_error("batch too large: %lld ordered pairs x hidden %d >= 2^31 elements per edge matrix (largest "  \
                      "verified size: 1024 x 64-atom graphs); split the batch", (long long)(P), (c).hidden);        \
            return TSD_ERR_UNSUPPORTED;                                                                             \
        }                                                                                                           \
    } while (0)

struct Workspace {
    // every array holds one block per checkpoint (stride_* floats apart): the M forwards of an ensemble
    // run in the SAME launches (grid.y = checkpoint), which removes the tile quantisation of batch-100
    // launches (508 workgroups on 256 CUs) and the per-checkpoint launch boundaries
    float *ea;   // [M][P, H]: rows 0..P/2-1 enc_u edges, rows P/2.. separately embedded (diff_u) out edges
    float *wf;   // [M][S, P/2, H]: CFConv filters on the undirected enc list, a ring of S = min(L, 2) layer slots:
                 // launch j writes block j's filters, launch j+1 is their only reader
    float *h, *x1, *x1b;  // [M][N, H]
    float *pre;  // [M][P/2, H]: node-independent half of the pair MLP's first layer (ComboPre)
    int32_t* ready;  // [M][node tiles] readiness flags of the last launch (pair role), zeroed by the embedding launch
    int32_t* ctl;    // control words of the one-launch forward (kernels_combo.hip MegaCtl), zeroed by the host per run
    size_t ctl_words;
    float* x1m;      // one-launch forward: [L - 1][N, H] x1 of every block (a buffer is written once per launch)
    size_t stride_ea, stride_wf, stride_nh, stride_pre, stride_ctl /* int32 words */, stride_x1m;
    int wf_slots;
    size_t total;
};

constexpr int WF_RING = 2;
#ifndef TSD_FOLD
#define TSD_FOLD 1  // the inference forward on the folded weights (common.hpp, FOLDED WEIGHTS); 0: A/B variant builds
#endif
constexpr bool kFold = TSD_FOLD != 0;

#ifndef TSD_MEGA
#define TSD_MEGA 1  // 0 (A/B variant builds): the split-f16 forward as one launch per block
#endif
// Shapes the one-launch forward takes: every node workgroup of a checkpoint resident at once on at most HALF of the
// device's slots for that kernel (occupancy x compute units, queried per device: 512 on a whole MI355X, less on a
// partitioned or smaller part), and a filter arena of all L blocks (no ring) that stays small.
// An ensemble (round 5) runs in the SAME launch as groups of G checkpoints -- as many as fit that half of the slots --
// staggered in the grid (kernels_combo.hip forward_mega_kernel), when it is whole groups: 2 checkpoints at batch 100 0.381 -> 0.312 ms/step, 8 at
// batch 25 0.426 -> 0.345 (one group each), 8 at batch 100 1.236 -> 1.206 (four groups of two), 4 x 200 graphs 1.191 ->
// 1.159 (four of one); a last group that is not full loses (3 checkpoints at batch 100 as 2 + 1: 0.475 -> 0.547), so
// such ensembles stay on the launch-per-block forms.
#ifndef TSD_MEGA_ENSEMBLE
#define TSD_MEGA_ENSEMBLE 1  // 0 (A/B variant builds): ensembles on the launch-per-block forms, as before round 5
#endif
// checkpoints per group of the one-launch forward
static int mega_group(const tsd_model_cfg& c, int N, int M) {
    const int node_wgs = (N + mega_node_rows() - 1) / mega_node_rows();
#ifndef TSD_MEGA_GROUP_DIV
#define TSD_MEGA_GROUP_DIV 2  // a group's node workgroups on at most 1 / DIV of the slots (4: measured slower, kernels_combo.hip)
#endif
    const int fit = node_wgs > 0 ? mega_slots(c.hidden) / TSD_MEGA_GROUP_DIV / node_wgs : 1;
    return fit < 1 ? 1 : (fit > M ? M : fit);
}
static bool mega_shape(const tsd_model_cfg& c, int N, int P, int M) {
    if (TSD_MEGA == 0 || M < 1 || (M != 1 && !TSD_MEGA_ENSEMBLE) || N <= 0 || c.num_convs > 60) return false;
    const int node_wgs = (N + mega_node_rows() - 1) / mega_node_rows();
    if (2 * node_wgs > mega_slots(c.hidden)) return false;
#ifndef TSD_MEGA_WHOLE_GROUPS
#define TSD_MEGA_WHOLE_GROUPS 1  // (0, A/B builds: also with a last group that is not full -- 3 checkpoints at batch 100 0.485 ->
                                 // 0.570 ms/step, 5: 0.826 -> 0.859, 4 x 80 graphs 0.513 -> 0.588, 7: 1.195 -> 1.149)
#endif
    if (TSD_MEGA_WHOLE_GROUPS && M > 1 && M % mega_group(c, N, M) != 0) return false;
    return (size_t)M * (P / 2) * c.hidden * c.num_convs * sizeof(float) <= ((size_t)2 << 30);
}

static Workspace carve(const tsd_model_cfg& c, int N, int P, int M, float* base) {
    Workspace w;
    const size_t H = c.hidden, PU = (size_t)P / 2;
    size_t o = 0;
    auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += n; return p; };
    auto pad = [](size_t n) { return (n + 63) & ~size_t(63); };
    w.stride_ea = pad(2 * PU * H);
    w.wf_slots = c.num_convs < WF_RING ? c.num_convs : WF_RING;
    // the one-launch forward (one checkpoint, <= 256 node tiles) keeps the filters of every block (no ring)
    if (mega_shape(c, N, P, M)) w.wf_slots = c.num_convs;
    w.stride_wf = pad((size_t)w.wf_slots * PU * H);
    w.stride_nh = pad((size_t)N * H);
    w.ea = take(w.stride_ea * M);
    w.wf = take(w.stride_wf * M);
    w.h = take(w.stride_nh * M);
    w.x1 = take(w.stride_nh * M);
    w.x1b = take(w.stride_nh * M);
    w.stride_pre = pad(PU * H);
    w.pre = take(w.stride_pre * M);
    w.ready = reinterpret_cast<int32_t*>(take(pad((size_t)M * ((N + TSD_NODE_TILE - 1) / TSD_NODE_TILE))));
    // (one control block and one x1m block per checkpoint)
    w.stride_ctl = pad(mega_shape(c, N, P, M) ? mega_ctl_words(filter_tiles_per_layer((int)PU), c.num_convs) : 64);
    w.ctl_words = w.stride_ctl * (mega_shape(c, N, P, M) ? (size_t)M : 1);
    w.ctl = reinterpret_cast<int32_t*>(take(w.ctl_words));
    w.stride_x1m = mega_shape(c, N, P, M) && c.num_convs > 1 ? w.stride_nh * (size_t)(c.num_convs - 1) : 0;
    w.x1m = take(w.stride_x1m * M);
    w.total = o;
    return w;
}

// One forward per checkpoint on the current positions.  Every per-edge MLP runs on the UNDIRECTED
// lists (half the edges of the reference's directed list: edge_attr, W and edge_inv are symmetric);
// the directed CSR list only drives the aggregation and eq_transform through `umap`.
// counts_ready: the per-row member counts are already in geo.scratch (written by the previous step's
// step_post_kernel); advance: device step counter bumped by the scan kernel (sampling loop only).
// lists_ready: the five lists of `pos` are complete but for the directed -> undirected map (the fused step tail of
// the previous step built them): no count / scan / fill launches at all.
// status: device word for TSD_STATUS_INTERNAL (sampling loop: the state block's flags); NULL: the pair MLP runs as its
// own launch (no in-launch waits anywhere in the forward).
// epoch_src / epoch_bias (sampling loop): the device word and offset that number the forwards of a run 1, 2, ... (the
// one-launch forward's hand-off words are monotonic and zeroed once per run); NULL: a stand-alone forward zeroes them.
static int forward_impl(const tsd_model_cfg& c, const tsd_batch& b, const float* pos, hipStream_t st,
                        bool counts_ready = false, int32_t* advance = nullptr, bool lists_ready = false,
                        int32_t* status = nullptr, const int32_t* epoch_src = nullptr, int epoch_bias = 1) {
    const int N = b.num_nodes, P = b.num_pairs, M = b.num_models;
    const int PU = P / 2, L = c.num_convs;
    const size_t H = c.hidden;
    const tsd_geometry& g = b.geo;
    int r;
    TraceRange range("tsd:score_forward");
    if (!counts_ready && !lists_ready) {
        if ((r = launch_geometry_count(c, N, pos, b.graph_ptr, b.node_graph, b.pair_ptr, b.pair_code, g, st))) return r;
    }
    // the directed-edge -> undirected-pair map is not needed before the first block launch, so it runs as an
    // extra role of the edge-embedding launch instead of a launch of its own on the critical path
    if (!lists_ready &&
        (r = launch_geometry_lists(c, N, P, pos, b.graph_ptr, b.node_graph, b.pair_ptr, b.pair_code, g, advance, st,
                                   true)))
        return r;
    // one launch per interaction block: node chain of block l || filter GEMMs of block l+1;
    // all M checkpoints in the same launches (grid.y)
    // Arithmetic of the tile GEMMs: split-f16 operands on the f16 MFMA pipes when the batch carries the f16-plane
    // arenas (csrc/split16.hpp; needs the typed embedding and the folded weights), else the fp32-input MFMA.  Every
    // tensor in memory is fp32 either way.  The range word: the sampling loop's flags, else the batch's.
    const bool typed = kFold && b.enc_tiles.num_tiles > 0 && b.bucket_weights != nullptr;
    Prec prec{};
    if (kFold && typed && b.weights16 != nullptr && b.bucket_weights16 != nullptr) {
        prec.mode = PREC_H2;
        prec.range_status = status ? status : b.status;
        prec.narrow_filter_tiles = (b.reserved & 2) != 0;
    }
    const bool h2 = prec.mode == PREC_H2;
    const float* W = h2 ? b.weights16 : b.weights;
    // The fused per-unit encoder (kernels_unit.hip: all L blocks in one launch, the CFConv filters never written to memory)
    // where a launch fills the chip -- ensembles, batches past the one-launch form's size -- and the batch carries its
    // unit partition (every graph <= TSD_UNIT_MAX_NODES atoms); tsd_batch.reserved bit 2 switches it off, bit 4 asks for
    // it also where the one-launch form would apply (tests, A/B).  Same bits as the materialising forms.
    const bool fused_ok = h2 && unit_encoder_supported(c) && b.unit_node != nullptr && b.num_units > 0 && P > 0 &&
                          prec.range_status != nullptr && !(b.reserved & 4);  // (a status word: the kernel validates its units)
    const bool mega_ok = h2 && mega_shape(c, N, P, M) && prec.range_status != nullptr && P > 0 && !(b.reserved & 1);
    const bool fused = fused_ok && (!mega_ok || (b.reserved & 16));
    const bool mega = mega_ok && !fused;
    UmapRole um{};
    um.g = g;
    um.graph_ptr = b.graph_ptr;
    um.node_graph = b.node_graph;
    um.pair_ptr = b.pair_ptr;
    um.P = P;
    const Workspace w = carve(c, N, P, M, b.workspace);
    const int node_tiles_all = (N + TSD_NODE_TILE - 1) / TSD_NODE_TILE;
    // small forwards (the node chain of a block leaves most of the chip idle): the LAST block launch also runs the
    // whole pair MLP behind per-node-tile readiness flags, which the embedding launch zeroes.  The split-f16 forward does so
    // up to 4096 node tiles x checkpoints (round 5: one launch boundary less is worth 300 graphs 0.463 -> 0.458 ms/step,
    // 600: 0.871 -> 0.857, 8 checkpoints at batch 100 1.258 -> 1.251, 200 x 4: 1.214 -> 1.205; not measured beyond)
#ifndef TSD_SMALL_FWD_MAX
#define TSD_SMALL_FWD_MAX 4096
#endif
    const bool small_fwd = (long)node_tiles_all * M <= (h2 ? TSD_SMALL_FWD_MAX : 256);
    if (small_fwd) {
        um.zero_words = w.ready;
        um.n_zero = node_tiles_all * M;
    }
    // the filter GEMMs of block 0 ride in the embedding launch (a tile's filters need only that tile's attributes):
    // the first per-block launch, which had no node chain to run beside them, disappears
    const WeightLayout WL0 = weight_layout(c);
    EmbedFuse0 f0{};
    // (the inference forward runs on the FOLDED weights: the attribute rows hold s1, common.hpp)
    f0.nn0_w = W + WL0.layer0 + (kFold ? WL0.L_nn0f_w : WL0.L_nn0_w);
    f0.nn0_b = W + WL0.layer0 + (kFold ? WL0.L_nn0f_b : WL0.L_nn0_b);
    f0.nn2_w = W + WL0.layer0 + WL0.L_nn2_w;
    f0.nn2_b = W + WL0.layer0 + WL0.L_nn2_b;
    f0.conv_cutoff = c.conv_cutoff;
    f0.smooth = c.smooth_conv;
    f0.wf = w.wf;
    f0.wf_stride = w.stride_wf;
    // Block 0's filters ride with the embedding tiles (the attribute tile is in LDS already: two more GEMMs instead of a
    // launch that reloads it).  With the typed embedding's small tile: everywhere (round 5, three workgroups per CU:
    // 300 graphs 0.484 -> 0.454 ms/step, 600: 0.867 -> 0.865, 8 checkpoints at batch 100 1.251 -> 1.247, 200 graphs x 4
    // checkpoints 1.206 -> 1.199; round 3 had it at batch-100 sizes and >= 2048 node tiles only); with the generic
    // embedding kernel at batch-100 sizes only (configs[4]: 50.5 -> 52.8 ms/step with it, round 2)
    const bool fuse_block0 = !fused && (small_fwd || mega || typed);
    if (typed) {
        if ((r = launch_typed_embed(c, W, b, pos, w.ea, M, w.stride_ea, st, &um, fuse_block0 ? &f0 : nullptr, prec))) return r;
    } else if ((r = launch_edge_embed2(c, W, PU, g.enc_u, w.ea, PU, g.diff_u, w.ea + (size_t)PU * H, M, w.stride_ea, st,
                                       &um, nullptr, 0, fuse_block0 ? &f0 : nullptr, kFold)))
        return r;
    if (mega) {
        // the L block launches and the pair MLP as ONE launch (kernels_combo.hip forward_mega_kernel): node workgroups
        // persistent over the blocks, filter tiles, pair tiles, in-launch hand-offs instead of launch boundaries
        if (epoch_src == nullptr) {
            TSD_HIP(hipMemsetAsync(w.ctl, 0, w.ctl_words * sizeof(int32_t), st));
            epoch_src = w.ctl + 32;  // (MegaCtl::ZERO: a word nothing writes)
            epoch_bias = 1;
        }
        // an ensemble: groups of as many checkpoints as keep the node workgroups within half of the slots, group after
        // group in the same grid
        MegaGroup mg;
        mg.M = M;
        mg.G = mega_group(c, N, M);
        mg.s_nh = w.stride_nh;
        mg.s_x1m = w.stride_x1m;
        mg.s_wf = w.stride_wf;
        mg.s_ea = w.stride_ea;
        mg.s_ctl = (int)w.stride_ctl;
        return launch_forward_mega(c, b, pos, W, w.ea, w.wf, w.h, w.x1m, w.stride_nh, w.ctl, epoch_src, epoch_bias,
                                   prec.range_status, st, mg);
    }
    if (fused) {
        // embedding launch (attribute rows) -> the whole encoder as one launch -> pair MLP launch
        if ((r = launch_unit_encoder(c, b, W, w.ea, w.stride_ea, w.h, w.stride_nh, 0, L, w.x1, prec.range_status, st)))
            return r;
        return launch_pair_output(c, W, PU, g.out_u, w.h, w.ea, g.attr_row, b.edge_inv_u, M, w.stride_nh, w.stride_ea,
                                  (size_t)PU, st, nullptr, w.stride_pre, nullptr, kFold, prec);
    }
    // block 0 reads z (residual input) and x1_0 = lin1_0(z) straight from the per-batch arrays -- both are
    // pos independent (computed at bind time) -- so no per-step copy of z and no lin1 launch
    if (w.stride_nh != (size_t)N * H) {
        set_error("internal: node stride mismatch");
        return TSD_ERR_INVALID;
    }
    // The filter tiles of all blocks form one queue (they depend on the geometry only); launch j takes block
    // j's tiles, so that they are complete before block j's node chain runs in launch j+1.  Launch j+1 is the
    // only reader of block j's filters, so they live in a ring of two layer slots (block l -> slot l % 2) --
    // the working set of a launch pair instead of all L layers (r01: 14.8 GB at config C5, now 4.2 GB).
    const int tpl = filter_tiles_per_layer(PU);
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
    pre.w0b = kFold ? W + WL.out_w0f : W + WL.out_w0 + H * H;  // packed [k/4][out][k%4]: the k >= H half is contiguous
    pre.b0 = W + (kFold ? WL.out_b0f : WL.out_b0);
    pre.out = w.pre;
    // (only when the node chain of the last block leaves most of the chip idle: at batch 100 it occupies ~100
    // of the 256 CUs; with an ensemble or a large batch the launch is full and the extra role only adds work:
    // C2 0.518 -> 0.511 ms/step, C5 51.1 -> 51.6, M = 8 3.10 -> 3.13)
    // (the split-f16 forward has no separate pre role: its pair MLP is either the pair role or the stand-alone kernel)
#ifndef TSD_PAIR_ROLE
#define TSD_PAIR_ROLE 1  // 0 (A/B variant builds): pre role + separate pair_output launch, as in round 2
#endif
    // (tsd_batch.reserved bit 0 -- the form the host falls back to after TSD_STATUS_INTERNAL -- has NO in-kernel wait:
    // its pair MLP is the stand-alone launch)
    const bool pair_role = small_fwd && TSD_PAIR_ROLE != 0 && status != nullptr && !(b.reserved & 1);
    const bool use_pre = small_fwd && (pair_role || !h2);
    if (pair_role) {
        // 64-row pair tiles where the role is many rounds deep (as the stand-alone pair launch: kernels_combo.hip
        // TSD_PAIR_OUT_WIDE_MIN; round 5: 1000 graphs, 8 x 300 graphs)
        if (h2 && H == 256 && !prec.narrow_filter_tiles && (long)pre.tiles * M >= 4096) {
            pre.rows = 2 * TSD_EDGE_TILE;
            pre.tiles = (PU + pre.rows - 1) / pre.rows;
        }
        pre.pair = 1;
        pre.w0a = W + WL.out_w0;
        pre.w1 = W + WL.out_w1;
        pre.b1 = W + WL.out_b1;
        pre.w2 = W + WL.out_w2;
        pre.b2 = W + WL.out_b2;
        pre.h = w.h;
        pre.edge_inv = b.edge_inv_u;
        pre.ready = w.ready;
        pre.status = status;
        pre.inv_stride = (size_t)PU;
    }
    for (int j = fuse_block0 ? 1 : 0; j <= L; ++j) {  // (fused: the filters of block 0 came with the embedding launch)
        const int layer = j == 0 ? -2 : j - 1;  // node chain of this launch; its filters were written by launch j-1
        const float* wf_read = layer >= 0 ? w.wf + (size_t)(layer % w.wf_slots) * PU * H : nullptr;
        if ((r = launch_layer_combo(c, W, layer, N, g.enc, wf_read, xin, layer == 0 ? b.z : w.h, w.h, xout, 0,
                                    j * tpl, j < L ? tpl : 0, PU, g.enc_u, w.ea, w.wf, w.wf_slots, M, w.stride_nh,
                                    w.stride_ea, w.stride_wf, st, (use_pre && j == L) ? &pre : nullptr, w.stride_pre,
                                    nullptr, nullptr, kFold, prec)))
            return r;
        if (layer >= 0) {
            xin = xout;
            xout = (xout == w.x1) ? w.x1b : w.x1;
        }
    }
    if (pair_role) return TSD_OK;  // the last block launch ran the pair MLP
    return launch_pair_output(c, W, PU, g.out_u, w.h, w.ea, g.attr_row, b.edge_inv_u, M, w.stride_nh, w.stride_ea,
                              (size_t)PU, st, use_pre ? w.pre : nullptr, w.stride_pre, nullptr, kFold, prec);
}

// one sampling step of the device-resident loop: lists (from the counts of the previous step's tail) ->
// M forwards -> fused tail (mean, eq_transform, update, centre, next step's counts)
// The fused tail (one launch: update + the NEXT step's lists) is used whenever the batch qualifies.
static bool use_step_tail(const tsd_batch& b) {
    return step_tail_supported(b.num_nodes, b.num_graphs, b.max_graph_nodes);
}
static int step_impl(const tsd_model_cfg& c, const tsd_batch& b, int kind, float clip, float clip_pos, float* pos,
                     tsd_sampler_state* state, hipStream_t st) {
    int r;
    if (use_step_tail(b)) {
        // (the fused tail leaves step = -1 after its list-only launch and + 1 per step: forward k of a run reads k - 2)
        if ((r = forward_impl(c, b, pos, st, true, nullptr, true, &state->flags, &state->step, 2))) return r;
        return launch_step_tail(c, kind, b.num_nodes, b.num_graphs, b.num_models, b.num_pairs, b.max_graph_nodes,
                                b.graph_ptr, b.pair_ptr, b.pair_code, b.geo, b.edge_inv_u, clip, clip_pos, pos, state, st);
    }
    // (the scan kernel of the list build advances step from -1 before the forward: forward k of a run reads k - 1)
    if ((r = forward_impl(c, b, pos, st, true, &state->step, false, &state->flags, &state->step, 1))) return r;
    return launch_step_post(c, kind, b.num_nodes, b.num_graphs, b.num_models, b.num_pairs, b.graph_ptr, b.pair_ptr,
                            b.pair_code, b.geo, b.edge_inv_u, clip, clip_pos, pos, state, st);
}

}  // namespace tsd

// the plan: one captured + instantiated step, replayed by every call (include/tsdiff_hip.h)
// (round 3: a second graph of TSD_PLAN_MULTI consecutive steps -- consecutive graph launches are ~9 us apart on the
// device, kernels inside a graph are back to back -- captured on the first call that runs that many steps; every
// per-step value comes from the device-side step counter / ticket, so a step is the same wherever it is replayed from)
// (round 6: graphs of 8, 4, 2 and 1 steps -- a call of n steps is n / 8 launches of the longest and at most three more, where
// the 8 + 1 form of rounds 3-5 ran the remainder step by step (the driver's 20-step call: 3 launches instead of 6) -- and
// tsd_sampler_plan_create captures them ALL, so that the first long call of a plan no longer pays for a capture.)
constexpr int TSD_PLAN_MULTI = 8;
constexpr int TSD_PLAN_SIZES = 4;                                  // graphs of 8, 4, 2, 1 steps
constexpr int TSD_PLAN_STEPS[TSD_PLAN_SIZES] = {TSD_PLAN_MULTI, 4, 2, 1};
struct tsd_sampler_plan {
    tsd_model_cfg cfg;
    tsd_batch batch;
    int kind;
    float clip, clip_pos;
    float* pos;
    tsd_sampler_state* state;
    hipGraph_t graph[TSD_PLAN_SIZES];
    hipGraphExec_t exec[TSD_PLAN_SIZES];
};

namespace tsd {
static int check_batch(const tsd_model_cfg& c, const tsd_batch* batch) {
    TSD_REQUIRE(batch != nullptr, "batch is null");
    TSD_CAPACITY(c, batch->num_nodes, batch->num_pairs);
    TSD_REQUIRE(batch->num_models >= 1, "num_models=%d", batch->num_models);
    TSD_REQUIRE(batch->z && batch->x1_0 && batch->weights && batch->workspace && batch->edge_inv_u,
                "null batch pointer");
    return TSD_OK;
}
}  // namespace tsd

using namespace tsd;

extern "C" {

const char* tsd_version(void) { return "tsdiff_hip 0.7 (gfx950; fp32 MFMA, split-f16 MFMA inference forward and training step, fused per-unit encoder, transposed-accumulator tile GEMMs, ensembles in the one-launch forward, windowed CFConv aggregation)"; }
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
    TraceRange range("tsd:pack_weights");
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(raw && packed, "null weight pointer");
    return launch_pack_weights(*cfg, raw, packed, (hipStream_t)stream);
}

int tsd_pack_weights16(const tsd_model_cfg* cfg, const float* packed, float* packed16, void* stream) {
    TraceRange range("tsd:pack_weights16");
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(packed && packed16 && packed != packed16, "null / aliased weight pointer");
    return launch_pack_weights16(*cfg, packed, packed16, (hipStream_t)stream);
}

int tsd_weights16_preflight(const float* weights, size_t num_floats, float* out8, void* stream) {
    TSD_REQUIRE(out8 != nullptr && (num_floats == 0 || weights != nullptr), "null pointer");
    return launch_weights_preflight(weights, num_floats, out8, (hipStream_t)stream);
}

int tsd_bucket_weights16(const tsd_model_cfg* cfg, const float* bucket_weights, int32_t num_slots, float* out16,
                         void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(num_slots >= 0 && (num_slots == 0 || (bucket_weights && out16 && bucket_weights != out16)), "bad argument");
    return launch_bucket_weights16(*cfg, bucket_weights, num_slots, out16, (hipStream_t)stream);
}

int tsd_topology_build(int32_t num_nodes, int32_t num_graphs, int32_t num_pairs, int64_t num_bonds,
                       const int32_t* graph_ptr, const int32_t* pair_base, const int64_t* bond_index,
                       const int64_t* bond_type, int32_t max_order, int32_t max_graph_nodes_host,
                       int32_t* node_graph, int32_t* pair_ptr, uint16_t* pair_code, int32_t* status,
                       void* stream) {
    TraceRange range("tsd:topology_build");
    TSD_REQUIRE(num_nodes >= 0 && num_graphs >= 0 && num_pairs >= 0 && num_bonds >= 0, "negative size");
    TSD_REQUIRE(graph_ptr && pair_base && node_graph && pair_ptr && status, "null pointer");
    TSD_REQUIRE(num_pairs == 0 || pair_code, "null pair_code");
    TSD_REQUIRE(num_bonds == 0 || (bond_index && bond_type), "null bond arrays");
    return launch_topology(num_nodes, num_graphs, num_pairs, num_bonds, graph_ptr, pair_base, bond_index,
                           bond_type, max_order, max_graph_nodes_host, node_graph, pair_ptr, pair_code, status,
                           (hipStream_t)stream);
}

size_t tsd_typed_tiles_capacity(int32_t num_pairs) { return typed_tiles_capacity(num_pairs); }

int tsd_typed_tiles_build(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_pairs, const int32_t* graph_ptr,
                          const int32_t* node_graph, const int32_t* pair_ptr, const uint16_t* pair_code,
                          int32_t* enc_pair, int32_t* enc_i, int32_t* enc_j, int32_t* enc_tile, int32_t* diff_pair,
                          int32_t* diff_i, int32_t* diff_j, int32_t* diff_tile, int32_t* keys, int32_t* counts_dev,
                          int32_t* scratch, void* stream) {
    TraceRange range("tsd:typed_tiles_build");
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(graph_ptr && node_graph && pair_ptr && keys && counts_dev && scratch, "null pointer");
    TSD_REQUIRE(num_pairs == 0 || (pair_code && enc_pair && enc_i && enc_j && enc_tile && diff_pair && diff_i &&
                                   diff_j && diff_tile),
                "null pointer");
    return launch_typed_tiles_build(*cfg, num_nodes, num_pairs, graph_ptr, node_graph, pair_ptr, pair_code, enc_pair,
                                    enc_i, enc_j, enc_tile, diff_pair, diff_i, diff_j, diff_tile, keys, counts_dev,
                                    scratch, (hipStream_t)stream);
}

size_t tsd_bucket_weights_floats(const tsd_model_cfg* cfg, int32_t num_slots) {
    if (check_cfg(cfg) || num_slots < 0) return 0;
    return (size_t)num_slots * ((size_t)cfg->hidden * cfg->hidden + cfg->hidden);
}

int tsd_bucket_weights_build(const tsd_model_cfg* cfg, const float* packed_weights, int32_t num_slots,
                             const int32_t* keys_dev, float* out, void* stream) {
    TraceRange range("tsd:bucket_weights_build");
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(num_slots >= 0 && (num_slots == 0 || (packed_weights && keys_dev && out)), "bad argument");
    return launch_bucket_weights(*cfg, packed_weights, num_slots, keys_dev, out, (hipStream_t)stream);
}

size_t tsd_geometry_scratch_ints(int32_t num_nodes, int32_t num_pairs) {
    return geometry_scratch_ints(num_nodes, num_pairs);
}

int tsd_geometry_build(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_graphs, int32_t num_pairs,
                       const float* pos, const int32_t* graph_ptr, const int32_t* node_graph,
                       const int32_t* pair_ptr, const uint16_t* pair_code, tsd_geometry geo, void* stream) {
    TraceRange range("tsd:geometry_build");
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
    TraceRange range("tsd:node_embed");
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

static int interaction_block_impl(int h2, int32_t* range_status, const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t num_nodes,
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
    Prec prec{};
    if (h2) {
        prec.mode = PREC_H2;
        prec.range_status = range_status;
    }
#ifdef TSD_TRACE
    extern int g_tsd_debug_prec;  // (kernels_combo.hip, variant builds: the traced launch in the split-f16 arithmetic)
    if (g_tsd_debug_prec) prec.mode = g_tsd_debug_prec;
#endif
    return launch_layer_combo(*cfg, w, layer, num_nodes, enc, Wf_layer, x1_in, nullptr, h, x1_out,
                              filter_layer < 0 ? 0 : filter_layer, 0,
                              filter_layer < 0 ? 0 : filter_tiles_per_layer(capacity_u), capacity_u, enc_u, edge_attr,
                              Wf_out, 1, 1, 0, 0, 0, (hipStream_t)stream, nullptr, 0, nullptr, nullptr, h2 != 0, prec);
}

int tsd_interaction_block(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t num_nodes, tsd_edges enc,
                          const float* Wf_layer, const float* x1_in, float* h, float* x1_out, int32_t filter_layer,
                          int32_t capacity_u, tsd_edges enc_u, const float* edge_attr, float* Wf_out, void* stream) {
    return interaction_block_impl(0, nullptr, cfg, w, layer, num_nodes, enc, Wf_layer, x1_in, h, x1_out, filter_layer,
                                  capacity_u, enc_u, edge_attr, Wf_out, stream);
}
int tsd_attr_planes(int32_t hidden, int64_t rows, const float* edge_attr, float* edge_attr16, int32_t* range_status,
                    void* stream) {
    TraceRange range("tsd:attr_planes");
    TSD_REQUIRE(rows >= 0 && (rows == 0 || (edge_attr && edge_attr16)), "null pointer");
    TSD_REQUIRE(edge_attr != edge_attr16, "in-place conversion is not supported");
    return launch_attr_planes(hidden, rows, edge_attr, edge_attr16, range_status, (hipStream_t)stream);
}

int tsd_interaction_block16(const tsd_model_cfg* cfg, const float* w16, int32_t layer, int32_t num_nodes, tsd_edges enc,
                            const float* Wf_layer, const float* x1_in, float* h, float* x1_out, int32_t filter_layer,
                            int32_t capacity_u, tsd_edges enc_u, const float* edge_attr, float* Wf_out,
                            int32_t* range_status, void* stream) {
    return interaction_block_impl(1, range_status, cfg, w16, layer, num_nodes, enc, Wf_layer, x1_in, h, x1_out, filter_layer,
                                  capacity_u, enc_u, edge_attr, Wf_out, stream);
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
    if (!capacity_ok(*cfg, num_nodes, num_pairs)) {
        set_error("batch too large: %d ordered pairs x hidden %d >= 2^31 elements per edge matrix; split the batch",
                  num_pairs, cfg->hidden);
        return 0;
    }
    return carve(*cfg, num_nodes, num_pairs, num_models < 1 ? 1 : num_models, nullptr).total;
}

int tsd_forward_workspace_layout(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_pairs, int32_t num_models,
                                 size_t* out) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(out != nullptr && num_nodes >= 0 && num_pairs >= 0, "bad argument");
    TSD_CAPACITY(*cfg, num_nodes, num_pairs);
    float* base = reinterpret_cast<float*>(sizeof(float));  // (carve only does pointer arithmetic)
    const Workspace w = carve(*cfg, num_nodes, num_pairs, num_models < 1 ? 1 : num_models, base);
    out[0] = (size_t)(w.ea - base);
    out[1] = (size_t)(w.wf - base);
    out[2] = (size_t)(w.h - base);
    out[3] = (size_t)(w.x1 - base);
    out[4] = (size_t)(w.x1b - base);
    out[5] = w.stride_nh;
    out[6] = (size_t)w.wf_slots;
    out[7] = w.total;
    return TSD_OK;
}

int tsd_forward_work(const tsd_model_cfg* cfg, int32_t num_nodes, int64_t enc_edges, int64_t out_edges,
                     int64_t diff_pairs, tsd_work* out) {
    int r = check_cfg(cfg);
    if (r) return r;
    TSD_REQUIRE(out && num_nodes >= 0 && enc_edges >= 0 && out_edges >= 0 && diff_pairs >= 0, "bad argument");
    const double H = cfg->hidden, L = cfg->num_convs, N = num_nodes;
    const double Eu = (double)(enc_edges / 2), Ou = (double)(out_edges / 2), E = (double)enc_edges, O = (double)out_edges;
    const double embed_row = (2 * H + 2 * H * H) + 6 * H * H;          // Linear(1,H), Linear(H,H) | edge_cat 2H->H->H
    const double embed_row_typed = 2 * H + 2 * H * H;                   // Linear(1,H), ONE folded H x H GEMM (typed tiles)
    const double filter_row = 4 * H * H;                                // nn.0, nn.2
    const double pair_row = 4 * H * H + H * H + H + H;                  // 2H->H, H->H/2, H/2->1, h_i * h_j
    const double node_block = 6 * H * H;                                // lin1, lin2, lin
    out->flops_edge_embed = (Eu + (double)diff_pairs) * embed_row_typed;
    out->flops_blocks = L * (Eu * filter_row + E * 2 * H + N * node_block);
    out->flops_pair_output = Ou * pair_row;
    out->flops_other = N * 13000.0;
    out->flops_executed = out->flops_edge_embed + out->flops_blocks + out->flops_pair_output + out->flops_other;
    out->flops_reference = E * (embed_row + L * (filter_row + 2 * H)) + O * (embed_row + pair_row) + N * (L * node_block + 13000.0);
    out->flops_block_launch = L * (Eu * (filter_row + H) + E * 2 * H + N * node_block) / (L + 1);
    out->bytes_aggregate = (4 * H + 4) * E + 8 * H * N + 4;            // stand-alone CFConv aggregation (HBM form)
    // the training step's forward keeps the reference's operation order (no folded weights): three GEMMs more per
    // embedded edge than the inference forward
    out->flops_train_forward = out->flops_executed + (Eu + (double)diff_pairs) * (embed_row - embed_row_typed);
    return TSD_OK;
}

int tsd_score_forward(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* pos, void* stream) {
    int r = check_cfg(cfg);
    if (r) return r;
    if ((r = check_batch(*cfg, batch))) return r;
    TSD_REQUIRE(pos, "null pointer");
    return forward_impl(*cfg, *batch, pos, (hipStream_t)stream);
}

int tsd_forward_blocks(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t epoch, void* stream) {
    TraceRange range("tsd:forward_blocks");
    int r = check_cfg(cfg);
    if (r) return r;
    if ((r = check_batch(*cfg, batch))) return r;
    TSD_REQUIRE(epoch >= 1, "epoch=%d (1, 2, ... since the first call)", epoch);
    const tsd_batch& b = *batch;
    const bool typed = kFold && b.enc_tiles.num_tiles > 0 && b.bucket_weights != nullptr;
    if (!(typed && b.weights16 && b.bucket_weights16 && b.status && b.num_models == 1 &&
          mega_shape(*cfg, b.num_nodes, b.num_pairs, b.num_models) && b.num_pairs > 0)) {
        set_error("tsd_forward_blocks: the batch does not take the one-launch split-f16 forward");
        return TSD_ERR_UNSUPPORTED;
    }
    const Workspace w = carve(*cfg, b.num_nodes, b.num_pairs, b.num_models, b.workspace);
    hipStream_t st = (hipStream_t)stream;
    if (epoch == 1) TSD_HIP(hipMemsetAsync(w.ctl, 0, w.ctl_words * sizeof(int32_t), st));
    return launch_forward_mega(*cfg, b, nullptr, b.weights16, w.ea, w.wf, w.h, w.x1m, w.stride_nh, w.ctl, w.ctl + 32, epoch,
                               b.status, st);
}

int tsd_forward_encoder(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t l_begin, int32_t l_end, void* stream) {
    TraceRange range("tsd:forward_encoder");
    int r = check_cfg(cfg);
    if (r) return r;
    if ((r = check_batch(*cfg, batch))) return r;
    const tsd_batch& b = *batch;
    const bool typed = kFold && b.enc_tiles.num_tiles > 0 && b.bucket_weights != nullptr;
    if (!(typed && b.weights16 && b.bucket_weights16 && b.status && unit_encoder_supported(*cfg) && b.unit_node &&
          b.num_units > 0 && b.num_pairs > 0)) {
        set_error("tsd_forward_encoder: the batch does not take the fused per-unit encoder");
        return TSD_ERR_UNSUPPORTED;
    }
    TSD_REQUIRE(l_begin >= 0 && l_begin < l_end && l_end <= cfg->num_convs, "blocks [%d, %d) outside [0, %d)", l_begin,
                l_end, cfg->num_convs);
    const Workspace w = carve(*cfg, b.num_nodes, b.num_pairs, b.num_models, b.workspace);
    return launch_unit_encoder(*cfg, b, b.weights16, w.ea, w.stride_ea, w.h, w.stride_nh, l_begin, l_end, w.x1, b.status,
                               (hipStream_t)stream);
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

int tsd_philox_normal(uint64_t seed, uint64_t offset, int64_t n_atoms, float* out, void* stream) {
    TSD_REQUIRE(n_atoms >= 0 && (n_atoms == 0 || out), "bad argument");
    return launch_philox_normal(seed, offset, n_atoms, out, (hipStream_t)stream);
}

// capture + instantiate the graph of TSD_PLAN_STEPS[i] consecutive steps (nothing executes)
static int plan_capture(tsd_sampler_plan* p, int i, hipStream_t st) {
    if (p->exec[i] != nullptr) return TSD_OK;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return check_hip(e, "hipStreamBeginCapture");
    int r = TSD_OK;
    for (int q = 0; q < TSD_PLAN_STEPS[i] && r == TSD_OK; ++q)
        r = step_impl(p->cfg, p->batch, p->kind, p->clip, p->clip_pos, p->pos, p->state, st);
    e = hipStreamEndCapture(st, &p->graph[i]);
    if (r == TSD_OK) r = check_hip(e, "hipStreamEndCapture");
    if (r == TSD_OK) r = check_hip(hipGraphInstantiate(&p->exec[i], p->graph[i], nullptr, nullptr, 0), "hipGraphInstantiate");
    // (the executable graph's one-time upload now, not inside its first launch; best effort)
    if (r == TSD_OK) (void)hipGraphUpload(p->exec[i], st);
    return r;
}

int tsd_sampler_plan_create(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t kind, float clip,
                            float clip_pos, float* pos, tsd_sampler_state* state, void* stream,
                            tsd_sampler_plan** plan_out) {
    TraceRange range("tsd:sampler_plan_create");
    int r = check_cfg(cfg);
    if (r) return r;
    if ((r = check_batch(*cfg, batch))) return r;
    TSD_REQUIRE(pos && state && plan_out, "null pointer");
    TSD_REQUIRE(kind == 0 || kind == 1, "kind=%d (0 = ld, 1 = ddpm)", kind);
    TSD_REQUIRE(stream != nullptr, "stream capture is illegal on the legacy default stream: pass a created stream");
    hipStream_t st = (hipStream_t)stream;
    tsd_sampler_plan* p = new tsd_sampler_plan{*cfg, *batch, kind, clip, clip_pos, pos, state, {}, {}};
    for (int i = TSD_PLAN_SIZES - 1; i >= 0 && r == TSD_OK; --i) r = plan_capture(p, i, st);  // (1 step first: errors surface early)
    if (r != TSD_OK) {
        tsd_sampler_plan_destroy(p);
        return r;
    }
    *plan_out = p;
    return TSD_OK;
}

int tsd_sampler_plan_run(tsd_sampler_plan* plan, int32_t n_steps, const tsd_run_args* args, int32_t use_graph,
                         void* stream) {
    TraceRange range("tsd:sampler_plan_run");
    TSD_REQUIRE(plan && args, "null pointer");
    TSD_REQUIRE(n_steps >= 0, "n_steps=%d", n_steps);
    TSD_REQUIRE(n_steps == 0 || args->coefs, "null coefs");
    if (n_steps == 0) return TSD_OK;
    hipStream_t st = (hipStream_t)stream;
    const tsd_batch& b = plan->batch;
    int r;
    if ((r = launch_set_run_args(plan->state, *args, st))) return r;
    {   // the hand-off words of the one-launch forward count the forwards of THIS run
        const Workspace wm = carve(plan->cfg, b.num_nodes, b.num_pairs, b.num_models, b.workspace);
        TSD_HIP(hipMemsetAsync(wm.ctl, 0, wm.ctl_words * sizeof(int32_t), st));
    }
    if (use_step_tail(b)) {
        // the first step's lists: the fused tail in its list-only form (epoch 1 of the run's ticket counter); every
        // later step finds the lists its predecessor's tail built
        TSD_REQUIRE(((int64_t)n_steps + 2) * b.num_graphs < (int64_t)1 << 31, "n_steps=%d too large for one call", n_steps);
        if ((r = launch_step_tail_reset(b.num_graphs, b.geo, st))) return r;
        if ((r = launch_step_tail(plan->cfg, -1, b.num_nodes, b.num_graphs, b.num_models, b.num_pairs, b.max_graph_nodes,
                                  b.graph_ptr, b.pair_ptr, b.pair_code, b.geo, b.edge_inv_u, plan->clip, plan->clip_pos,
                                  plan->pos, plan->state, st)))
            return r;
    } else if ((r = launch_geometry_count(plan->cfg, b.num_nodes, plan->pos, b.graph_ptr, b.node_graph, b.pair_ptr,
                                          b.pair_code, b.geo, st))) {
        // member counts of the first step's lists (later steps get them from the previous step's tail kernel)
        return r;
    }
    int k = 0;
    if (use_graph) {
        // longest graphs first: n / 8 launches of the 8-step graph, then at most one each of 4, 2, 1
        for (int i = 0; i < TSD_PLAN_SIZES; ++i) {
            const int sz = TSD_PLAN_STEPS[i];
            if (k + sz > n_steps) continue;
            if ((r = plan_capture(plan, i, st))) return r;  // (a plan from tsd_sampler_plan_create holds them all already)
            for (; k + sz <= n_steps; k += sz) TSD_HIP(hipGraphLaunch(plan->exec[i], st));
        }
    }
    for (; k < n_steps; ++k)
        if ((r = step_impl(plan->cfg, b, plan->kind, plan->clip, plan->clip_pos, plan->pos, plan->state, st))) return r;
    return TSD_OK;
}

void tsd_sampler_plan_destroy(tsd_sampler_plan* plan) {
    if (!plan) return;
    for (int i = 0; i < TSD_PLAN_SIZES; ++i) {
        if (plan->exec[i]) (void)hipGraphExecDestroy(plan->exec[i]);
        if (plan->graph[i]) (void)hipGraphDestroy(plan->graph[i]);
    }
    delete plan;
}

int tsd_sampler_run(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t kind, int32_t n_steps,
                    const float* coefs, const float* noises, uint64_t seed, uint64_t offset, float clip,
                    float clip_pos, float* pos, float* traj, tsd_sampler_state* state, int32_t use_graph,
                    void* stream) {
    TraceRange range("tsd:sampler_run");
    tsd_sampler_plan* plan = nullptr;
    int r;
    // a bare plan: the one-shot form captures only the graphs its own n_steps needs (tsd_sampler_plan_run does, lazily);
    // use_graph == 0: no capture at all, the same kernels launched one by one
    if ((r = check_cfg(cfg))) return r;
    if ((r = check_batch(*cfg, batch))) return r;
    TSD_REQUIRE(pos && state, "null pointer");
    TSD_REQUIRE(kind == 0 || kind == 1, "kind=%d (0 = ld, 1 = ddpm)", kind);
    TSD_REQUIRE(!use_graph || stream != nullptr, "stream capture is illegal on the legacy default stream: pass a created stream");
    plan = new tsd_sampler_plan{*cfg, *batch, kind, clip, clip_pos, pos, state, {}, {}};
    const tsd_run_args args{coefs, noises, traj, seed, offset};
    r = tsd_sampler_plan_run(plan, n_steps, &args, use_graph, stream);
    // the exec object must outlive its launches: wait for the stream before destroying it
    const hipError_t se = use_graph ? hipStreamSynchronize((hipStream_t)stream) : hipSuccess;
    tsd_sampler_plan_destroy(plan);
    if (r) return r;
    return check_hip(se, "hipStreamSynchronize");
}

}  // extern "C"

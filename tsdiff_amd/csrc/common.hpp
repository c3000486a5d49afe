// common.hpp -- shared device/host helpers of libtsdiff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tsdiff_hip.h"

namespace tsd {

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

// A named range around a library call or one of its phases, visible to `rocprofv3 --marker-trace` (and to
// --kernel-rename): roctxRangePushA / roctxRangePop are looked up in the process image once -- rocprofv3 preloads
// librocprofiler-sdk-roctx.so when markers are requested -- and the range is a no-op when they are absent, so the
// library neither links nor loads a profiler.
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
    bool on;
};
int check_hip(hipError_t e, const char* what);

#define TSD_HIP(call)                                              \
    do {                                                           \
        int _r = ::tsd::check_hip((call), #call);                  \
        if (_r != TSD_OK) return _r;                               \
    } while (0)

#define TSD_LAUNCH_CHECK(name) TSD_HIP((hipGetLastError()))

#define TSD_REQUIRE(cond, ...)                                     \
    do {                                                           \
        if (!(cond)) {                                             \
            ::tsd::set_error(__VA_ARGS__);                         \
            return TSD_ERR_INVALID;                                \
        }                                                          \
    } while (0)

// One-time set-up per DEVICE (a kernel's dynamic-LDS limit is a per-device function attribute, and one process
// may drive several GPUs): first_on_current_device() is true exactly once per device id for each instance.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool first_on_current_device() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return true;
        const uint64_t bit = 1ull << (d & 63);
        return !(mask.fetch_or(bit) & bit);
    }
};
template <typename K>
static inline int allow_lds(K kernel, size_t bytes, DeviceOnce& once) {
    if (bytes > 48 * 1024 && once.first_on_current_device())
        TSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// packed weight arena (floats).  Dense matrix W[out][in] is stored as Bp[in/4][out][in%4] so that
// MFMA B-operand lane (kgroup, col) fetches one aligned float4 and a wave fetches 1 KiB contiguous.
// ---------------------------------------------------------------------------------------------
struct WeightLayout {
    size_t bond_emb, emlp_w0, emlp_b0, emlp_w1, emlp_b1, atom_emb, atom_feat;
    size_t ecat_w0, ecat_b0, ecat_w1, ecat_b1;
    size_t layer0, layer_stride;  // per layer: see L_* offsets below
    size_t out_w0, out_b0, out_w1, out_b1, out_w2, out_b2;
    size_t out_w0f, out_b0f;  // folded (see below): W0[:, H:] . W_cat2 [H x H], W0[:, H:] . b_cat2 + b0
    size_t total;
    // offsets inside one layer block
    size_t L_nn0_w, L_nn0_b, L_nn2_w, L_nn2_b, L_lin1_w, L_lin2_w, L_lin2_b, L_lin_w, L_lin_b;
    size_t L_nn0f_w, L_nn0f_b;  // folded: W_nn0 . W_cat2, W_nn0 . b_cat2 + b_nn0
};
// FOLDED WEIGHTS (round 3).  edge_cat.2 is a Linear whose output feeds only Linears -- nn.0 of every interaction block
// (schnet.py:77) and the edge half of grad_dist_mlp.0 (condensenc.py:236, common.py:226-229) -- with no activation in
// between (condensenc.py:105-115).  The inference forward therefore keeps s1 = swish(edge_cat.0(...)) as its "edge
// attribute" and uses W' = W . W_cat2, b' = W . b_cat2 + b (products accumulated in fp64 at pack time, rounded once):
// one H x H GEMM per embedded edge less (17 % of the embedding launch).  Same function, another fp32 association
// (measured distance to an fp64 evaluation unchanged, profiles/r03_parity_report.md); the training step and the
// piecewise entry points (tsd_edge_embed, tsd_filter_gen, tsd_interaction_block, tsd_pair_output) keep the
// reference's operation order.

inline WeightLayout weight_layout(const tsd_model_cfg& c) {
    WeightLayout L;
    const size_t H = c.hidden, F = c.feat_dim, HH = H * H;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~size_t(3); return r; };  // 16-B aligned
    L.bond_emb = take(100 * H);
    L.emlp_w0 = take(H);
    L.emlp_b0 = take(H);
    L.emlp_w1 = take(HH);
    L.emlp_b1 = take(H);
    L.atom_emb = take(100 * (H / 2));
    L.atom_feat = take((H / 2) * F);
    L.ecat_w0 = take(2 * HH);
    L.ecat_b0 = take(H);
    L.ecat_w1 = take(HH);
    L.ecat_b1 = take(H);
    L.layer0 = o;
    size_t lo = 0;
    auto ltake = [&](size_t n) { size_t r = lo; lo += (n + 3) & ~size_t(3); return r; };
    L.L_nn0_w = ltake(HH);
    L.L_nn0_b = ltake(H);
    L.L_nn2_w = ltake(HH);
    L.L_nn2_b = ltake(H);
    L.L_lin1_w = ltake(HH);
    L.L_lin2_w = ltake(HH);
    L.L_lin2_b = ltake(H);
    L.L_lin_w = ltake(HH);
    L.L_lin_b = ltake(H);
    L.L_nn0f_w = ltake(HH);
    L.L_nn0f_b = ltake(H);
    L.layer_stride = lo;
    o += lo * (size_t)c.num_convs;
    L.out_w0 = take(2 * HH);
    L.out_b0 = take(H);
    L.out_w1 = take(HH / 2);
    L.out_b1 = take(H / 2);
    L.out_w2 = take(H / 2);
    L.out_b2 = take(4);
    L.out_w0f = take(HH);
    L.out_b0f = take(H);
    L.total = o;
    return L;
}

// directed edge e -> index of its undirected pair in the matching *_u list (needs every row's fill to be
// complete).  Stand-alone: edge_umap_kernel (kernels_graph.hip); in the fused forward it rides as an extra role
// of the edge-embedding launch, which needs the undirected lists only (UmapRole, kernels_mlp.hip).
__device__ __forceinline__ void edge_umap_body(const tsd_geometry& g, const int32_t* __restrict__ graph_ptr,
                                               const int32_t* __restrict__ node_graph,
                                               const int32_t* __restrict__ pair_ptr, int P, int e) {
    const int Ee = *g.enc.count, Eo = *g.out.count;
    if (e < Ee) {
        const int i = g.enc.src[e], j = g.enc.dst[e];
        int p = g.enc.pair_id[e];
        if (i > j) {
            const int lo = graph_ptr[node_graph[i]];
            const int il = i - lo, jl = j - lo;
            p = pair_ptr[j] + il - (il > jl ? 1 : 0);
        }
        g.enc.umap[e] = g.pair2u[p];
    }
    if (e < Eo) {
        const int i = g.out.src[e], j = g.out.dst[e];
        int p = g.out.pair_id[e];
        if (i > j) {
            const int lo = graph_ptr[node_graph[i]];
            const int il = i - lo, jl = j - lo;
            p = pair_ptr[j] + il - (il > jl ? 1 : 0);
        }
        g.out.umap[e] = g.pair2u[(size_t)P + p];
    }
}
struct UmapRole {
    int blocks;  // 0: no umap role
    tsd_geometry g;
    const int32_t *graph_ptr, *node_graph, *pair_ptr;
    int P;
    int32_t* zero_words;  // words zeroed by the role (the node-tile readiness flags of the forward's last launch), or NULL
    int n_zero;
};

// Arithmetic of the tile GEMMs of a launch: PREC_F32 = fp32-input MFMA (exact fp32 fma chains: training, piecewise entry
// points, fallback), PREC_H2 = split-f16 operands on the f16 MFMA pipes (split16.hpp; the inference forward's default).
// A PREC_H2 launch takes the f16-plane weight arena (tsd_pack_weights16) where the fp32 one is documented, and a device
// word that receives TSD_STATUS_RANGE when an operand left the f16 range.
// ATTRIBUTE ROWS AS f16 PLANES (round 5).  The split-f16 embedding launch (kernels_typed.hip) writes a row of the edge
// attribute matrix not as H floats but as the two f16 planes its consumers need -- bytes [0, 2H): the high plane,
// [2H, 4H): the low plane scaled by 2^11 (split16.hpp), the same 4 H bytes -- so that the filter tiles of every block, the
// fused encoder's tiles and the pair tiles copy 16-byte chunks straight into their LDS planes instead of converting the same
// fp32 row L + 1 times per forward (the conversion was 10 % of a fused-encoder tile).  The planes hold exactly what the
// consumers computed before: results are bit-identical.  The range / low-side checks of these rows moved to the producer.
// Every INFERENCE role of the split-f16 arithmetic (filter_role_h, pair_role_h, the unit encoder) takes plane rows; the
// saving (training) forms, the fp32 kernels and the fp32 piecewise entry points keep fp32 rows; tsd_attr_planes converts
// an fp32 attribute matrix for the piecewise split-f16 entry point tsd_interaction_block16.
enum : int { PREC_F32 = 0, PREC_H2 = 1 };
struct Prec {
    int mode = PREC_F32;
    int32_t* range_status = nullptr;
    bool narrow_filter_tiles = false;  // 32-row filter tiles also where a launch would take 64-row ones (tsd_batch.reserved bit 1)
};

// optional third role of the per-block launch (kernels_combo.hip), filled by the forward in api.hip
struct ComboPre {
    int tiles;  // 0: no pre role
    int rows = TSD_EDGE_TILE;  // pairs per tile of the pair role: 32, or 64 (split-f16, hidden 256, launches many rounds deep)
    tsd_edges e;
    const float* edge_attr;
    const int32_t* attr_row;  // edge_attr row of out edge e (NULL: e)
    const float *w0b, *b0;    // packed W0 rows k in [H, 2H), bias
    float* out;               // [capacity_u, H]  (pre role only)
    // ---- pair role (round 3): the WHOLE pair MLP of a tile of 32 undirected out edges inside the forward's last
    // block launch -- the node-independent half first, then, behind per-node-tile readiness flags published by the
    // node role of the same launch (cdna_hip_programming.md Guideline 16, R1: write-through payload, drained flag), the
    // h_i * h_j half and the rest of pair_output_kernel.  Replaces the pre role + the pair_output launch.
    int pair;                 // != 0: pair role (needs the fields below)
    const float *w0a, *w1, *b1, *w2, *b2;  // packed W0 rows k in [0, H); layers 1, 2
    const float* h;           // final node states [N, H] (written by the node role of this launch)
    float* edge_inv;          // [capacity_u]
    int32_t* ready;           // [M][node tiles] flags: >= ready_target = the tile's rows of h are in memory
    int ready_target = 1;     // (per-block launches: flags are 0 / 1; one-launch forward: epoch * 64 + blocks)
    int ready_div = TSD_NODE_TILE;  // atoms per flag (node tile rows: 16; the one-launch forward's node tiles: 8)
    int32_t* status;          // TSD_STATUS_INTERNAL on a wait that gave up
    size_t inv_stride;        // per-checkpoint stride of edge_inv
};

// Side outputs of the fused forward kernels for the training step (train_step.hip): the activations the
// backward pass reads, written next to the inference results by the SAVE instantiations of the same kernels.
struct EmbedSave {    // rows are edge-attribute rows: enc_u edge e -> e, k-th diff_u edge -> save_b_row + k
    float *l0, *s0;   // Linear(1,H)(d) and its swish                         [rows,H]
    float *e, *c;     // e = mlp(d) [rows,H]; c = [e * emb[type_r], e * emb[type_p]] [rows,2H]
    float *c0, *s1;   // edge_cat.0(c) and its swish                           [rows,H]
    float* d;         // the row's distance                                    [rows]   (the two lists' inputs as ONE
    uint8_t *tr, *tp; // the row's bond types                                  [rows]    row range for the wgrads)
};
struct FilterSave {   // per interaction block: [layer][capacity_u, H]
    float *f0, *fs;   // nn.0 output and its shifted softplus
};
struct NodeSave {     // the block of this launch, [N,H] each
    float *agg, *x2, *xs;  // CFConv aggregate; lin2 output and its shifted softplus
};
struct PairSave {     // [capacity_u, .]
    float* hp;        // [2H]: h_i * h_j || edge_attr_out
    float *g0, *gs0;  // [H]: first layer of the pair MLP and its swish
    float *g1, *gs1;  // [H/2]: second layer and its swish
    // attribute rows >= attr_from lie attr_shift rows lower than geo.attr_row says (the training step packs the
    // separately embedded out edges right behind the enc_u rows: one contiguous row range for the weight gradients)
    int attr_from = 0x7fffffff, attr_shift = 0;
};

// The filter GEMMs of interaction block 0 appended to the enc-list tiles of the edge-embedding launch (inference
// forward: a tile's filters need only that tile's edge attributes, which are still in LDS -- one launch and one
// read of the attribute rows less per forward).  Same arithmetic in the same order as the filter role: bit-identical.
struct EmbedFuse0 {
    const float *nn0_w, *nn0_b, *nn2_w, *nn2_b;  // block 0 (checkpoint m at + m * weight stride)
    float conv_cutoff;
    int smooth;
    float* wf;          // filter slot of block 0 (checkpoint m at + m * wf_stride)
    size_t wf_stride;
};

// launchers of the fused forward kernels shared by the inference forward (api.hip) and the training step
// (train_step.hip); `save` != NULL selects the SAVE instantiation
// fold: the "edge attribute" written is s1 = swish(edge_cat.0(...)) (consumers use the folded weights)
// (the launchers below that take a `Prec` run their GEMMs in that arithmetic; W is then the matching arena)
int launch_edge_embed2(const tsd_model_cfg& c, const float* W, int cap_a, tsd_edges ea, float* out_a, int cap_b,
                       tsd_edges eb, float* out_b, int M, size_t out_stride, hipStream_t st, const UmapRole* umap,
                       const EmbedSave* save = nullptr, int save_b_row = 0, const EmbedFuse0* fuse0 = nullptr,
                       bool fold = false);
int launch_layer_combo(const tsd_model_cfg& c, const float* W, int layer, int N, tsd_edges enc, const float* Wf_layer,
                       const float* x1_in, const float* h_in, float* h, float* x1_out, int layer_w0, int g_begin,
                       int g_count, int capacity_u, tsd_edges enc_u, const float* edge_attr, float* wf_base,
                       int wf_slots, int M, size_t nh_stride, size_t ea_stride, size_t wf_stride, hipStream_t st,
                       const ComboPre* pre, size_t pre_stride, const FilterSave* fsave = nullptr,
                       const NodeSave* nsave = nullptr, bool folded = false, Prec prec = Prec{});
int filter_tiles_per_layer(int capacity_u);
int launch_pair_output(const tsd_model_cfg& c, const float* W, int capacity, tsd_edges e, const float* h,
                       const float* edge_attr, const int32_t* attr_row, float* edge_inv, int M, size_t h_stride,
                       size_t ea_stride, size_t inv_stride, hipStream_t st, const float* pre, size_t pre_stride,
                       const PairSave* save = nullptr, bool folded = false, Prec prec = Prec{});

int launch_edge_embed_save_h(const tsd_model_cfg& c, const float* W16, int cap_a, tsd_edges ea, float* out_a, int cap_b,
                             tsd_edges eb, float* out_b, hipStream_t st, const EmbedSave& save, int save_b_row,
                             int32_t* range_status);
int launch_pair_output_h(const tsd_model_cfg& c, const float* W16, int capacity, tsd_edges e, const float* h,
                         const float* edge_attr, const int32_t* attr_row, float* edge_inv, int M, size_t h_stride,
                         size_t ea_stride, size_t inv_stride, hipStream_t st, bool folded, int32_t* range_status,
                         const PairSave* save = nullptr, bool narrow = false);
int launch_pack_weights16(const tsd_model_cfg& c, const float* packed, float* packed16, hipStream_t st);
int launch_attr_planes(int H, int64_t rows, const float* src, float* dst, int32_t* range_status, hipStream_t st);
int launch_weights_preflight(const float* w, size_t n, float* out8, hipStream_t st);
// the whole split-f16 forward of one checkpoint as ONE launch (kernels_combo.hip, small batches)
struct MegaGroup {   // the M checkpoints of a batch in groups of G in one launch of the one-launch forward; strides in floats / int32 words
    int M = 1, G = 1;
    size_t s_nh = 0, s_x1m = 0, s_wf = 0, s_ea = 0;
    int s_ctl = 0;
};
int launch_forward_mega(const tsd_model_cfg& c, const tsd_batch& b, const float* pos, const float* W16, float* ea, float* wf,
                        float* h, float* x1m, size_t x1_stride, int32_t* ctl, const int32_t* epoch_src, int epoch_bias,
                        int32_t* status, hipStream_t st, const MegaGroup& mg = MegaGroup{});
size_t mega_ctl_words(int tiles_per_layer, int L);
int mega_node_rows();
int mega_slots(int H);  // resident workgroup slots of the one-launch kernel on the current device (0: unknown)
// the whole SchNet encoder as one launch of per-unit workgroups, filters never materialised (kernels_unit.hip)
bool unit_encoder_supported(const tsd_model_cfg& c);
int launch_unit_encoder(const tsd_model_cfg& c, const tsd_batch& b, const float* W16, const float* ea, size_t ea_stride,
                        float* h, size_t nh_stride, int l_begin, int l_end, float* x1_io, int32_t* status, hipStream_t st);
int launch_bucket_weights16(const tsd_model_cfg& c, const float* bucket, int num_slots, float* out16, hipStream_t st);

inline bool hidden_supported(int H) { return H == 64 || H == 128 || H == 256; }

// Workgroup -> (item, checkpoint) on the 1-D grids of the ensemble launches (round 5).  M > 0: checkpoint = id % M --
// workgroups go to the 8 XCDs round robin (id % 8, MI355X_MICROARCH.md), so the 8 checkpoints of a production ensemble run
// one per XCD and each 4-MB L2 holds ONE checkpoint's weight images of the running block; M < 0: checkpoint-major (id /
// items), the order of rounds 1-4.  The launchers interleave while a checkpoint's share of the grid is less than two
// chip-fulls of workgroup slots (ckpt_grid_m): there the checkpoint-major order has three or four checkpoints resident at
// a time and every XCD sees them all (measured, tools/ab_step.py: 8 checkpoints at batch 100 1.391 -> 1.326 ms/step,
// 4 checkpoints 0.710 -> 0.702); with 300 graphs x 8 checkpoints a checkpoint's items are several chip-fulls, one or two
// checkpoints are resident either way, and the interleaved order measured 1 % slower (3.621 vs 3.590).  A speed
// assumption only.
__device__ __forceinline__ void wg_item_ckpt(int M, int& item, size_t& m) {
    const unsigned id = blockIdx.x;
    if (M > 0) {
        item = (int)(id / (unsigned)M);
        m = id % (unsigned)M;
    } else {
        const unsigned items = gridDim.x / (unsigned)(-M);
        item = (int)(id % items);
        m = id / items;
    }
}
#ifndef TSD_CKPT_INTERLEAVE_MAX
#define TSD_CKPT_INTERLEAVE_MAX 1024  // items per checkpoint below which the grid is interleaved (0: never)
#endif
inline int ckpt_grid_m(int M, long items_per_ckpt) { return (M > 1 && items_per_ckpt < TSD_CKPT_INTERLEAVE_MAX) ? M : -M; }

// ---------------------------------------------------------------------------------------------
// device math (fp32, no fast-math)
// ---------------------------------------------------------------------------------------------
// Activations use the hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each).
// Absolute error of either function is ~1e-7 for every input, the same class as the fp32 rounding of
// the reference's own softplus(x) - log 2 cancellation; the library versions cost ~100 VALU ops per
// element, which was 30 % of the fused CFConv kernel (profiles/r01_a_kernel_stats.md).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float swishf(float x) {  // reference utils/activation_functions.py:10-11
    return x * __builtin_amdgcn_rcpf(1.0f + fast_exp(-x));  // x * sigmoid(x)
}
__device__ __forceinline__ float sspf(float x) {  // reference models/encoder/schnet.py:65-71
    // F.softplus(beta=1, threshold=20) - log(2) = max(x,0) + log1p(exp(-|x|)) - log 2
    const float t = fast_exp(-fabsf(x));
    const float l = __builtin_amdgcn_logf(1.0f + t) * 0.69314718055994530942f;
    return (fmaxf(x, 0.0f) + l) - 0.69314718055994530942f;
}

// Philox4x32-10 (Salmon et al., SC'11): counter-based generator for the Gaussian draws of the sampling loop
// (reference models/sampler.py:213 torch.randn_like): key = seed, counter = (ctr, 0), four u32 per call.
struct u32x4 { uint32_t x, y, z, w; };
__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint64_t ctr, uint64_t seed) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}
// three standard normals of counter `ctr` (Box-Muller on the four uniforms; the fourth normal is dropped)
__device__ __forceinline__ void philox_normal3(uint64_t ctr, uint64_t seed, float (&z)[3]) {
    const u32x4 r = philox4x32_10(ctr, seed);
    const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1), 24 bits
    const float u1 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(r.w >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float sa, ca, sb, cb;
    sincospif(2.0f * u1, &sa, &ca);
    sincospif(2.0f * u3, &sb, &cb);
    (void)sb;
    z[0] = ra * ca;
    z[1] = ra * sa;
    z[2] = rb * cb;
}

// CFConv cutoff weight C(d), reference models/encoder/schnet.py:92-98
__device__ __forceinline__ float cutoff_weight(float d, float cutoff, int smooth) {
    if (!(d <= cutoff)) return 0.0f;
    if (!smooth) return 1.0f;
    return d >= 0.0f ? 0.5f * (cosf(d * 3.14159265358979323846f / cutoff) + 1.0f) : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// MFMA tile GEMM:  acc[T x NOUT] += A[T x K] * W^T   (A in LDS, W packed in global/L2)
//
// 32x32x2 f32 MFMA: A operand lane l holds A[i = l&31][k = l>>5], B operand B[k = l>>5][j = l&31],
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16).
// k is permuted in groups of 8: MFMA step s of k-block kb consumes k = kb*8 + s (lanes 0-31) and
// k = kb*8 + 4 + s (lanes 32-63); A and B use the same permutation so the contraction is unchanged.
// Each lane therefore reads ONE float4 of A (LDS, ds_read_b128) and one float4 of B per column
// block (global_load_dwordx4, a wave reads 1 KiB contiguous) per 4 MFMAs.
//
// RB = T/32 row blocks, CB = column blocks (32 wide) owned by this wave starting at col0.
//
// The B feed is a RING of R k-blocks in flight per wave, issued by inline asm with counted `s_waitcnt vmcnt`:
// hipcc's scheduler collapses every source-level prefetch of this loop to two loads in flight, each re-issued
// into the registers its MFMAs have just consumed, with 64-bit VALU address arithmetic between the dependent
// MFMAs (r01: 62-67 % of the MFMA rate in the GEMM phases).  The asm loads are invisible to that scheduler:
// wave-uniform base in SGPRs (advanced by SALU), one 32-bit lane offset in a VGPR, R loads deep.  Measured in
// tools/mfma_probe3.hip in this kernel shape (8 waves x 32 columns, one accumulator per wave, 2 workgroups per
// CU): compiler-scheduled loop 135 TFLOP/s, ring of 4 or 8: 152 (registers-only ceiling: 152).
//
// Contract of the asm loads (cdna_hip_programming.md 5.7): a destination counts as written at the asm statement,
// so every consumer takes its operand from the `s_waitcnt` statement that names the registers "+v" (a true data
// dependency: no MFMA can be scheduled above its wait); loads return in issue order, so compiler-issued loads
// that are older or younger than the ring only make the counted waits conservative; the kernels issue no
// global STORES while a ring is in flight.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one B fragment (float4 per lane) for CB column blocks of k-block `kb`: CB loads, column block cb at +cb*cb_bytes
template <int CB>
__device__ __forceinline__ void ring_issue(f32x4 (&b)[CB], const char* __restrict__ sbase /* wave-uniform */,
                                           unsigned voff /* lane byte offset */, int cb_bytes) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const char* sb = sbase + (size_t)cb * cb_bytes;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[cb]) : "v"(voff), "s"(sb) : "memory");
    }
}
// wait until at most N younger loads are outstanding and hand the fragment to its consumers
template <int N, int CB>
__device__ __forceinline__ void ring_wait(f32x4 (&b)[CB]) {
    static_assert(CB >= 1 && CB <= 4, "");
    if constexpr (CB == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(b[0]) : "n"(N) : "memory");
    if constexpr (CB == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
    if constexpr (CB == 3)
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N) : "memory");
    if constexpr (CB == 4)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}

template <int V>
struct IntC { static constexpr int value = V; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<I + 1, N>(f);
    }
}

// ring depth: as many k-blocks as fit 32 VGPRs of fragments
template <int CB>
constexpr int ring_depth() { return CB == 1 ? 8 : (CB == 2 ? 4 : 2); }

// The ring split in two calls, so that a kernel can put the first R k-blocks in flight BEFORE a barrier /
// epilogue and consume them after it (ring_start ... gemm_tile_ring); gemm_tile is start + consume.
template <int CB, int R>
struct BRing {
    f32x4 b[R][CB];
    const char* base;   // wave-uniform: packed matrix + this GEMM's first k-block
    unsigned voff;      // lane byte offset inside a k-block row pair
    int kb_bytes, cb_bytes;
};

template <int CB, int R, int KB>
__device__ __forceinline__ void ring_start(BRing<CB, R>& r, const float* __restrict__ Bp, int nout, unsigned lane_f4,
                                           int kb_rows /* k rows of float4 per k-block: 2 (32x32x2) or 4 (16x16x4) */,
                                           int cb_f4 /* float4 per column block: 32 or 16 */) {
    r.base = reinterpret_cast<const char*>(Bp);
    r.voff = lane_f4 * 16u;
    r.kb_bytes = kb_rows * nout * 16;
    r.cb_bytes = cb_f4 * 16;
    static_for<0, (R < KB ? R : KB)>([&](auto i) {
        constexpr int I = decltype(i)::value;
        ring_issue<CB>(r.b[I], r.base + (size_t)I * r.kb_bytes, r.voff, r.cb_bytes);
    });
}

template <int RB, int CB, int K, int R>
__device__ __forceinline__ void gemm_tile_ring(BRing<CB, R>& r, const float* __restrict__ ldsA, int lda,
                                               f32x16 (&acc)[RB][CB]) {
    const int lane = threadIdx.x & 63;
    const float* aptr = ldsA + (lane & 31) * lda + (lane >> 5) * 4;
    constexpr int KB = K / 8;
    static_for<0, KB>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        constexpr int slot = kb % R;
        constexpr int younger = ((kb + R <= KB) ? R : KB - kb) - 1;  // k-blocks issued after this one, still wanted
        f32x4 a[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(aptr + rb * 32 * lda + kb * 8);
        ring_wait<younger * CB, CB>(r.b[slot]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][s], r.b[slot][cb][s], acc[rb][cb], 0, 0, 0);
        // refill the slot: its MFMAs have been issued (operands are read at issue), the data lands an L2 latency later
        if constexpr (kb + R < KB) ring_issue<CB>(r.b[slot], r.base + (size_t)(kb + R) * r.kb_bytes, r.voff, r.cb_bytes);
    });
}

template <int RB, int CB, int K>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ ldsA, int lda,
                                          const float* __restrict__ Bp, int nout, int col0,
                                          f32x16 (&acc)[RB][CB]) {
    constexpr int R = ring_depth<CB>();
    const int lane = threadIdx.x & 63;
    BRing<CB, R> r;
    ring_start<CB, R, K / 8>(r, Bp, nout, (unsigned)((lane >> 5) * nout + col0 + (lane & 31)), 2, 32);
    gemm_tile_ring<RB, CB, K, R>(r, ldsA, lda, acc);
}

// gemm_tile in two halves: gemm_ring_start puts the first ring_depth k-blocks of B in flight (they depend on
// nothing the kernel computes), gemm_ring_run consumes them.  A kernel starts the ring BEFORE the staging barrier /
// the epilogue that precedes the GEMM, so that the first MFMA does not wait an L2 round trip.
template <int CB, int K>
__device__ __forceinline__ void gemm_ring_start(BRing<CB, ring_depth<CB>()>& r, const float* __restrict__ Bp, int nout,
                                                int col0) {
    const int lane = threadIdx.x & 63;
    ring_start<CB, ring_depth<CB>(), K / 8>(r, Bp, nout, (unsigned)((lane >> 5) * nout + col0 + (lane & 31)), 2, 32);
}
template <int RB, int CB, int K>
__device__ __forceinline__ void gemm_ring_run(BRing<CB, ring_depth<CB>()>& r, const float* __restrict__ ldsA, int lda,
                                              f32x16 (&acc)[RB][CB]) {
    gemm_tile_ring<RB, CB, K, ring_depth<CB>()>(r, ldsA, lda, acc);
}

// 16-row variant on the 16x16x4 f32 MFMA (same rate, half the rows): A lane l holds A[i = l&15][k = l>>4],
// B lane l holds B[k = l>>4][j = l&15], C/D: col = l&15, row = (l>>4)*4 + r, r in [0,4).
// k is permuted in groups of 16: step s of k-block kb consumes k = kb*16 + q*4 + s on lane quarter q, so a
// lane reads one float4 of A and one float4 of B per column block per 4 MFMAs, from the SAME packed
// weight layout ([k/4][out][k%4]) as the 32-row variant.  CB = 16-wide column blocks of this wave.
template <int CB, int K, int R>
__device__ __forceinline__ void gemm_tile16_ring(BRing<CB, R>& r, const float* __restrict__ ldsA, int lda,
                                                 f32x4 (&acc)[CB]) {
    const int lane = threadIdx.x & 63;
    const float* aptr = ldsA + (lane & 15) * lda + (lane >> 4) * 4;
    constexpr int KB = K / 16;
    static_for<0, KB>([&](auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        constexpr int slot = kb % R;
        constexpr int younger = ((kb + R <= KB) ? R : KB - kb) - 1;
        const f32x4 a = *reinterpret_cast<const f32x4*>(aptr + kb * 16);
        ring_wait<younger * CB, CB>(r.b[slot]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], r.b[slot][cb][s], acc[cb], 0, 0, 0);
        if constexpr (kb + R < KB) ring_issue<CB>(r.b[slot], r.base + (size_t)(kb + R) * r.kb_bytes, r.voff, r.cb_bytes);
    });
}

template <int CB, int K>
__device__ __forceinline__ void gemm_tile16(const float* __restrict__ ldsA, int lda,
                                            const float* __restrict__ Bp, int nout, int col0,
                                            f32x4 (&acc)[CB]) {
    constexpr int R = ring_depth<CB>();
    const int lane = threadIdx.x & 63;
    BRing<CB, R> r;
    ring_start<CB, R, K / 16>(r, Bp, nout, (unsigned)((lane >> 4) * nout + col0 + (lane & 15)), 4, 16);
    gemm_tile16_ring<CB, K, R>(r, ldsA, lda, acc);
}

template <int CB, int K>
__device__ __forceinline__ void gemm16_ring_start(BRing<CB, ring_depth<CB>()>& r, const float* __restrict__ Bp,
                                                  int nout, int col0) {
    const int lane = threadIdx.x & 63;
    ring_start<CB, ring_depth<CB>(), K / 16>(r, Bp, nout, (unsigned)((lane >> 4) * nout + col0 + (lane & 15)), 4, 16);
}
template <int CB, int K>
__device__ __forceinline__ void gemm16_ring_run(BRing<CB, ring_depth<CB>()>& r, const float* __restrict__ ldsA, int lda,
                                                f32x4 (&acc)[CB]) {
    gemm_tile16_ring<CB, K, ring_depth<CB>()>(r, ldsA, lda, acc);
}

// 16-byte store of a streamed output (CFConv filter rows).  Cache policy: 1 = write-through (sc1): the bytes leave the
// XCD's L2 as they are written instead of at the kernel's end-of-launch write-back (MI355X_MICROARCH.md price list,
// rows boundary / publish-large); 0 = plain; 2 = non-temporal (measured slower: 0.448 ms/step).  Variant builds:
// tools/build_variant.sh NAME "-DTSD_WF_STORE=0"
#ifndef TSD_WF_STORE
#define TSD_WF_STORE 1  // measured at batch 100 (tools/ab_step.py, round 3): 0.4303 vs 0.4334 ms/step with plain stores
#endif
__device__ __forceinline__ void store_stream16(float* p, const f32x4 v) {
#if TSD_WF_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif TSD_WF_STORE == 2
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}

template <int RB, int CB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RB][CB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.0f;
}

// row of accumulator register r for this lane (within a 32-row block)
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

}  // namespace tsd

#include "split16.hpp"

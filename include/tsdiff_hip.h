/* tsdiff_hip.h -- C ABI of libtsdiff_hip.so: the MI355X (gfx950) implementation of the
 * TSDiff score-network denoising hot path.
 *
 * Every entry point replaces one Python-level operation of the reference
 * (seonghann/tsdiff, paths relative to the reference root) and is what a
 * ctypes / cffi binding on the reference side would bind:
 *
 *   tsd_topology_*      models/common.py:115-202   _extend_ts_graph_order (pos-independent part)
 *   tsd_geometry_build  models/common.py:205-223,328-384 + models/epsnet/condensenc.py:117-154
 *                       + models/geometry.py:18-19 (radius graph, union, types, edge_length)
 *   tsd_node_embed      models/epsnet/condensenc.py:193-198
 *   tsd_edge_embed      models/epsnet/condensenc.py:156-176, models/encoder/edge.py:58-68
 *   tsd_cfconv_layer    models/encoder/schnet.py:88-107 (filter MLP + message + scatter-add, fused)
 *   tsd_cfconv_aggregate models/encoder/schnet.py:102,106 (MessagePassing aggr="add" alone)
 *   tsd_node_update     models/encoder/schnet.py:103,123-127,223-224 (lin2, ssp, lin, residual, next lin1)
 *   tsd_filter_gen      models/encoder/schnet.py:94-99 (CFConv filters of all layers, once per undirected pair)
 *   tsd_interaction_block  one launch per block: node chain of block l || filters of block l+1
 *   tsd_pair_output     models/common.py:226-229 + models/epsnet/condensenc.py:236-237
 *   tsd_eq_transform*   models/geometry.py:22-30
 *   tsd_sampler_step    models/sampler.py:208-251 (clip_norm, LD/DDPM update, NaN flag, center_pos)
 *   tsd_score_forward   models/epsnet/condensenc.py:178-239 + models/sampler.py:58-116 (M checkpoints)
 *   tsd_sampler_run     models/sampler.py:187-254 (the 5000-step loop, device resident, hipGraph replay)
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - the library never allocates or frees device memory: outputs and scratch are caller
 *     allocated (torch's caching allocator on the Python side);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no
 *     host synchronisation (exceptions, each documented at its declaration: the one-shot tsd_sampler_run
 *     and tsd_train_forward) and returns 0 on success or a negative TSD_ERR_* code;
 *     tsd_last_error() returns a thread-local message for the last failure;
 *   - no mutable process-global state: the only globals are that thread-local error string,
 *     per-device "kernel attributes set" bits and the roctx entry points resolved once from the process image
 *     (the forward, sampling-loop, geometry / topology and training entry points run inside "tsd:<name>"
 *     ranges that `rocprofv3 --marker-trace` records; absent a profiler the ranges are no-ops); a process may drive several devices (the device current
 *     at the call is used), one stream per call;
 *   - every tensor at this boundary is fp32; the GEMMs run on the fp32-input MFMA (exact-f32 FMA chains) or, in the
 *     inference forward when the batch carries `weights16`, on the f16 MFMA with split operands and fp32 accumulation
 *     (22-bit operands: the fp32 error class, see tsd_pack_weights16); indices at this
 *     boundary are int32 except where the reference surface hands over int64 tensors
 *     (bond_index, bond_type, atom_type, r_feat, p_feat);
 *   - hidden size H must be 64, 128 or 256 (reference config: 256).
 */
#ifndef TSDIFF_HIP_H
#define TSDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSD_OK 0
#define TSD_ERR_INVALID (-1)     /* bad argument (shape, hidden size, null pointer) */
#define TSD_ERR_HIP (-2)         /* a HIP runtime call failed; see tsd_last_error() */
#define TSD_ERR_UNSUPPORTED (-3) /* e.g. a graph with more than TSD_MAX_GRAPH_NODES atoms */
#define TSD_ERR_NAN (-4)         /* reported through status words, mapped to FloatingPointError */
#define TSD_ERR_RANGE (-5)       /* tsd_train_backward2 of a split-f16 step (tsd_batch.reserved bit 5) whose forward raised
                                    TSD_STATUS_RANGE: nothing was launched; run tsd_train_forward again without the bit
                                    (same workspace, same loss buffer), then the backward */

#define TSD_EDGE_TILE 32         /* edges per workgroup tile of every per-edge kernel */
#define TSD_NODE_TILE 16         /* nodes per workgroup tile of the per-node kernels */
#define TSD_EDGE_PAD 8           /* spare entries every tsd_edges array must carry past its capacity */
#define TSD_MAX_GRAPH_NODES 255  /* two u8 hop matrices of n*n must fit in the 160 KiB LDS */
#define TSD_UNIT_MAX_NODES 64    /* atoms of one unit of the fused encoder (its x1 rows live in LDS) */
#define TSD_NUM_BOND_TYPES 22    /* len(rdkit BondType.names), reference utils/chem.py:21 */

/* status word bits (device int32 written by kernels, read by the caller when it chooses) */
#define TSD_STATUS_NAN 1           /* NaN in positions after an update (sampler.py:248-250) */
#define TSD_STATUS_BAD_BOND 2      /* bond across graphs / self loop / index out of range */
#define TSD_STATUS_ASYMMETRIC 4    /* bond list is not symmetric (A0 contract: both directions) */
#define TSD_STATUS_INTERNAL 8      /* a bounded in-kernel wait (fused step tail, one-launch forward) gave up -- e.g. another
                                      tenant of the GPU held the workgroup slots the waited-for tiles needed; the results
                                      of the call are invalid: rerun it with tsd_batch.reserved bit 0 set and
                                      max_graph_nodes = 0 (forms without in-kernel waits; the Python host does) */
#define TSD_STATUS_RANGE 16        /* split-f16 forward (tsd_batch.weights16): an activation left the f16 range
                                      (|a| > 65504); the results of the call are invalid, rerun it with weights16 = NULL */

typedef struct tsd_model_cfg {
    int32_t hidden;          /* config.hidden_dim == config.encoder.hidden_dim */
    int32_t num_convs;       /* config.encoder.num_convs */
    int32_t feat_dim;        /* config.feat_dim */
    int32_t edge_order;      /* config.edge_order (encoder graph) */
    int32_t pred_edge_order; /* config.pred_edge_order (output graph) */
    float edge_cutoff;       /* config.edge_cutoff (radius graph) */
    float conv_cutoff;       /* config.encoder.cutoff (CFConv mask C) */
    int32_t smooth_conv;     /* config.encoder.smooth_conv: C = 0.5 (cos(pi d / cutoff) + 1) inside the cutoff (schnet.py:92-96) */
} tsd_model_cfg;

/* One extended-graph edge list in device memory (capacity = num_pairs entries + TSD_EDGE_PAD spare ones: the
 * node role of tsd_interaction_block reads dst / umap eight edges at a time and may touch, never use, up to
 * seven entries past the last edge).  Sorted row-major by (src, dst) exactly like the reference's edge_index. */
typedef struct tsd_edges {
    int32_t* count;    /* [1]   number of edges E */
    int32_t* row_ptr;  /* [N+1] CSR over src */
    int32_t* src;      /* [cap] edge_index[0] */
    int32_t* dst;      /* [cap] edge_index[1] */
    float* dist;       /* [cap] edge_length */
    uint8_t* type_r;   /* [cap] edge type in the reactant graph (0, 1..21, 22+hop-1) */
    uint8_t* type_p;   /* [cap] same for the product graph */
    int32_t* pair_id;  /* [cap] index of the edge's ordered pair in the topology */
    int32_t* umap;     /* [cap] directed lists only: index of the edge's undirected pair {i,j} in the
                                matching undirected list (enc -> enc_u, out -> out_u) */
} tsd_edges;

/* Everything tsd_geometry_build produces for one set of positions.
 * The extended edge set, edge_length and the types are symmetric under (i,j) <-> (j,i), hence so are
 * the edge embedding, the CFConv filter W and edge_inv: the per-edge MLPs run once per UNDIRECTED
 * pair (lists *_u, src < dst, capacity P/2) and the directed lists (reference order, what the
 * Python surface returns) index into them through `umap`. */
typedef struct tsd_geometry {
    tsd_edges enc, out;              /* directed, order edge_order / pred_edge_order */
    tsd_edges enc_u, out_u, diff_u;  /* undirected; diff_u: out_u edges whose (d, type_r, type_p) differ
                                        from their enc_u edge (or have none) and need their own embedding */
    int32_t* attr_row;   /* [P/2] per out_u edge: row of the [P,H] edge-attribute matrix holding its
                                  embedding: its enc_u index, or P/2 + k for the k-th diff_u edge */
    int32_t* pair2out;   /* [P] directed out edge index of every ordered pair, -1 if not an edge */
    int32_t* pair2u;     /* [2P] scratch: undirected enc / out index of every ordered pair with src < dst */
    int32_t* scratch;    /* [tsd_geometry_scratch_ints] */
} tsd_geometry;

const char* tsd_version(void);
const char* tsd_last_error(void);

/* ---- weights ---------------------------------------------------------------------------
 * `raw` is the reference state_dict flattened, each tensor row-major [out,in], in this order:
 *   edge_encoder.bond_emb.weight, edge_encoder.mlp.layers.0.{weight,bias}, .1.{weight,bias},
 *   atom_embedding.weight, atom_feat_embedding.weight,
 *   for l in 0..L-1: encoder.interactions.l.conv.lin1.weight, conv.lin2.{weight,bias},
 *                    conv.nn.0.{weight,bias}, conv.nn.2.{weight,bias}, lin.{weight,bias},
 *   grad_dist_mlp.layers.{0,1,2}.{weight,bias}, edge_cat.0.{weight,bias}, edge_cat.2.{weight,bias}
 * `packed` receives the MFMA-friendly layout ([k/4][out][k%4] for every dense matrix). */
size_t tsd_raw_weight_floats(const tsd_model_cfg* cfg);
size_t tsd_packed_weight_floats(const tsd_model_cfg* cfg);
int tsd_pack_weights(const tsd_model_cfg* cfg, const float* raw, float* packed, void* stream);
/* The split-f16 image of a packed arena (0.4): the inference forward's tile GEMMs run on the f16 MFMA pipes (16x the
 * fp32-input MFMA rate on gfx950) with every fp32 operand split into two f16 planes, a = hi + lo * 2^-11, three f16
 * MFMAs per product, fp32 accumulation: 22 significant bits per operand, the same error class as the fp32 fma chain
 * (csrc/split16.hpp; measured in profiles/).  `packed16` has tsd_packed_weight_floats floats: every dense matrix
 * rewritten in place of its fp32 image as [k/16][plane][k%16/8][out][k%8] f16, everything else copied. */
int tsd_pack_weights16(const tsd_model_cfg* cfg, const float* packed, float* packed16, void* stream);

/* Pre-flight of a weight arena for the split-f16 arithmetic (a real checkpoint before its first forward): over
 * `num_floats` floats -- the packed arena of tsd_pack_weights, which holds the folded matrices the forward multiplies with,
 * or a bucket arena of tsd_bucket_weights_build -- out8 (DEVICE, 8 floats) receives [0] max |w|, [1] the number of
 * non-zero weights below 2^-14 = 6.1e-5 (their high plane is an f16 subnormal: the operand keeps an absolute precision of
 * ~1.5e-11 instead of 22 bits -- harmless in small numbers, a reason to run `weights16 = NULL` if a whole layer sits
 * there), [2] the number beyond the f16 range 65504 (the forward would report TSD_STATUS_RANGE at once), [3] num_floats.
 * Activations cannot be pre-flighted: the forward tracks their range itself (TSD_STATUS_RANGE). */
int tsd_weights16_preflight(const float* weights, size_t num_floats, float* out8 /* device [8] */, void* stream);

/* ---- work model (the figures every roofline fraction in bench.py is quoted against) -----------
 * Arithmetic of ONE forward of one checkpoint for a batch with the given edge counts: `enc_edges` / `out_edges`
 * directed edges of the encoder / output lists, `diff_pairs` undirected output pairs embedded separately. */
typedef struct tsd_work {
    double flops_edge_embed, flops_blocks, flops_pair_output, flops_other; /* executed by the INFERENCE forward: per-edge
                                   MLPs once per undirected pair, embedding on the type-folded matrices (tsd_typed_tiles) */
    double flops_executed;      /* their sum */
    double flops_reference;     /* the reference's directed formulation of the same forward */
    double flops_block_launch;  /* average executed flops of one of the L + 1 per-block launches */
    double bytes_aggregate;     /* algorithmic HBM bytes of one stand-alone CFConv aggregation (tsd_cfconv_aggregate) */
    double flops_train_forward; /* executed by the TRAINING step's forward (reference operation order, undirected pairs) */
} tsd_work;
int tsd_forward_work(const tsd_model_cfg* cfg, int32_t num_nodes, int64_t enc_edges, int64_t out_edges,
                     int64_t diff_pairs, tsd_work* out);

/* ---- topology (once per batch; pos independent) -------------------------------------------
 * graph_ptr [G+1]: node offsets; pair_base [G+1]: prefix sums of n_g*(n_g-1).
 * Outputs: node_graph [N], pair_ptr [N+1], pair_code [P] (u16: bondR | bondP<<5 | hopR<<10 | hopP<<13),
 * status [1] (TSD_STATUS_* bits OR-ed in). max_order = max(edge_order, pred_edge_order) <= 7. */
int tsd_topology_build(int32_t num_nodes, int32_t num_graphs, int32_t num_pairs, int64_t num_bonds,
                       const int32_t* graph_ptr, const int32_t* pair_base,
                       const int64_t* bond_index /* [2,num_bonds] */, const int64_t* bond_type,
                       int32_t max_order, int32_t max_graph_nodes_host,
                       int32_t* node_graph, int32_t* pair_ptr, uint16_t* pair_code,
                       int32_t* status, void* stream);

/* ---- geometry (every step) ---------------------------------------------------------------
 * Builds the directed and undirected edge lists of `pos` (see tsd_geometry). No host sync. */
size_t tsd_geometry_scratch_ints(int32_t num_nodes, int32_t num_pairs);
int tsd_geometry_build(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_graphs, int32_t num_pairs,
                       const float* pos, const int32_t* graph_ptr, const int32_t* node_graph,
                       const int32_t* pair_ptr, const uint16_t* pair_code, tsd_geometry geo, void* stream);

/* ---- network pieces (one checkpoint each; `w` = packed weights of that checkpoint) --------- */
int tsd_node_embed(const tsd_model_cfg* cfg, const float* w, int32_t num_nodes,
                   const int64_t* atom_type, const int64_t* r_feat, const int64_t* p_feat,
                   float* z /* [N,H] */, void* stream);

int tsd_edge_embed(const tsd_model_cfg* cfg, const float* w, int32_t capacity, tsd_edges edges,
                   float* edge_attr /* [cap,H] */, void* stream);

/* x1 = lin1_layer(h) for layer 0 (no bias). */
int tsd_node_lin1(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t num_nodes,
                  const float* h, float* x1, void* stream);

/* Fused CFConv message pass of `layer`: W = nn(edge_attr) * C; agg[i] = sum_{e in row i} x1[dst e] * W_e.
 * Complete rows go to agg [N,H]; rows cut by a tile boundary go to part [ceil(cap/32),2,H]
 * (deterministic two-level reduction, finished by tsd_node_update). */
int tsd_cfconv_layer(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t capacity,
                     tsd_edges enc, const float* edge_attr, const float* x1,
                     float* agg, float* part, void* stream);

/* CFConv filters of ALL layers for an (undirected) edge list in one launch:
 * Wf[l][e] = nn_l(edge_attr[e]) * (dist[e] <= conv_cutoff)   -> Wf [num_convs, capacity, H]
 * (reference schnet.py:94-99; the filter does not depend on the node states, so the seven layers'
 * GEMMs are batched: 7x the tiles of one layer keep all 256 CUs busy at batch-100 sizes). */
int tsd_filter_gen(const tsd_model_cfg* cfg, const float* w, int32_t capacity, tsd_edges edges,
                   const float* edge_attr, float* Wf, void* stream);

/* The scatter-add with a materialised filter (T5; HBM/L2-bound):
 * out[i] = sum_{e: row_ptr[i] <= e < row_ptr[i+1]} x1[dst[e]] * W[umap ? umap[e] : e], edges in order. */
int tsd_cfconv_aggregate(int32_t hidden, int32_t num_nodes, const int32_t* row_ptr, const int32_t* dst,
                         const int32_t* umap, const float* W, const float* x1, float* out, void* stream);

/* One interaction block in ONE launch (the production path): workgroups [0, ceil(N/16)) run the node
 * chain of `layer` -- agg[i] = sum_{e in row i} x1_in[dst e] * Wf_layer[enc.umap[e]] in edge order, then
 * h += lin(ssp(lin2(agg))), x1_out = lin1_{layer+1}(h) -- and the remaining workgroups generate the
 * CFConv filters of `filter_layer` (normally layer+1; -1: none) into Wf_out on the CUs the short node
 * chain leaves idle.  layer == -1: the node role is only x1_out = lin1_0(h); layer == -2: no node role.
 * x1_in and x1_out must be different buffers.  reference schnet.py:94-107,123-127,223-224 */
int tsd_interaction_block(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t num_nodes,
                          tsd_edges enc, const float* Wf_layer, const float* x1_in, float* h, float* x1_out,
                          int32_t filter_layer, int32_t capacity_u, tsd_edges enc_u, const float* edge_attr,
                          float* Wf_out, void* stream);

/* The same launch in the split-f16 arithmetic of the inference forward (w16: tsd_pack_weights16 image; the filter role
 * then takes edge_attr = s1 and the FOLDED nn.0 of tsd_pack_weights, as the forward does); range_status: device word
 * for TSD_STATUS_RANGE or NULL.  Node-side inputs and all outputs are the fp32 ones; `edge_attr` is in the form the
 * split-f16 embedding launch leaves it in (0.6): every row = its two f16 planes, bytes [0, 2H) the high plane, [2H, 4H)
 * the low plane x 2^11 -- tsd_attr_planes converts an fp32 matrix and, as the embedding launch does for its rows, raises
 * TSD_STATUS_RANGE in *range_status (device word or NULL) for a value beyond the f16 range or a run of 8 channels that is
 * tiny throughout (csrc/split16.hpp); tsd_interaction_block16 does not re-check rows it receives as planes (0.7). */
int tsd_attr_planes(int32_t hidden, int64_t rows, const float* edge_attr /* [rows, H] fp32 */,
                    float* edge_attr16 /* [rows, H] floats: plane rows */, int32_t* range_status, void* stream);
int tsd_interaction_block16(const tsd_model_cfg* cfg, const float* w16, int32_t layer, int32_t num_nodes,
                            tsd_edges enc, const float* Wf_layer, const float* x1_in, float* h, float* x1_out,
                            int32_t filter_layer, int32_t capacity_u, tsd_edges enc_u, const float* edge_attr,
                            float* Wf_out, int32_t* range_status, void* stream);

/* h += lin(ssp(lin2(agg) )); if next_layer >= 0 also x1 = lin1_{next_layer}(h).
 * part / enc_row_ptr: only for agg produced by tsd_cfconv_layer (rows cut by tile edges); pass NULL
 * for a complete agg (tsd_cfconv_aggregate). */
int tsd_node_update(const tsd_model_cfg* cfg, const float* w, int32_t layer, int32_t next_layer,
                    int32_t num_nodes, const int32_t* enc_row_ptr, const float* agg, const float* part,
                    float* h, float* x1, void* stream);

/* edge_attr: edge-attribute matrix, row attr_row[e] belongs to out edge e (attr_row NULL: row e). */
int tsd_pair_output(const tsd_model_cfg* cfg, const float* w, int32_t capacity, tsd_edges out,
                    const float* h, const float* edge_attr, const int32_t* attr_row,
                    float* edge_inv /* [cap] */, void* stream);

/* ---- distance score -> Cartesian score -------------------------------------------------- */
/* generic (any edge list, int64 like the reference surface); fp32 atomics. score must be zeroed. */
int tsd_eq_transform(int32_t num_nodes, int64_t num_edges, const float* score_d, const float* pos,
                     const int64_t* edge_index /* [2,E] */, const float* edge_length,
                     float* score_pos /* [N,3] */, void* stream);

/* ---- static type-sorted embedding tiles (round 3; pos independent, once per batch) ----------------
 * The edge embedding of the reference (models/encoder/edge.py:58-68 + condensenc.py:156-176) is
 *   s1 = swish(edge_cat.0([mlp(d) * emb[type_r], mlp(d) * emb[type_p]])),   mlp(d) = W1 swish(w0 d + b0) + b1
 * and for a FIXED (type_r, type_p) the two products and edge_cat.0 collapse into ONE H x H matrix:
 *   s1 = swish(Wt swish(w0 d + b0) + bt),  Wt = (Wc0[:, :H] diag(emb[type_r]) + Wc0[:, H:] diag(emb[type_p])) W1.
 * Types are topology: the undirected candidate pairs of a batch are sorted by (type_r, type_p) ONCE, cut into
 * tiles of <= 32 pairs of one type pair, and every step's embedding launch walks these static tiles (a pair that is
 * not an edge at this step is computed and dropped) with the per-bucket matrices of tsd_bucket_weights_build:
 * one GEMM per embedded edge instead of three.  `enc`: every pair with the encoder-graph types; `diff`: the pairs
 * whose output-graph types differ from them (the only ones that ever need an embedding of their own). */
typedef struct tsd_typed_tiles {
    int32_t num_tiles;         /* HOST value (read back once after tsd_typed_tiles_build); 0: not built */
    int32_t num_buckets;       /* HOST value: distinct (type_r, type_p) pairs of this list */
    const int32_t* tile_slot;  /* [num_tiles] slot of the tile's bucket in the bucket-weight arena */
    const int32_t* tile_start; /* [num_tiles] first entry of the tile in pair / node_i / node_j */
    const int32_t* tile_count; /* [num_tiles] 1..32 */
    const int32_t* pair;       /* [P/2] ordered-pair index (src < dst) of the candidates, grouped by bucket */
    const int32_t* node_i;     /* [P/2] src atom */
    const int32_t* node_j;     /* [P/2] dst atom */
} tsd_typed_tiles;
/* Scratch / output sizes: pair, node_i, node_j: P/2 ints each per list; tile arrays: tsd_typed_tiles_capacity ints
 * each per list; keys: [2][1024] ints (bucket_key of slot s of list l at keys[l * 1024 + s]: type_r * 32 + type_p);
 * counts_dev [4] = {enc tiles, enc buckets, diff tiles, diff buckets} (device; the caller reads them back once);
 * scratch: 8192 ints.  Diff-list slots are numbered after the enc list's (slot = enc buckets + k). */
size_t tsd_typed_tiles_capacity(int32_t num_pairs);
int tsd_typed_tiles_build(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_pairs, const int32_t* graph_ptr,
                          const int32_t* node_graph, const int32_t* pair_ptr, const uint16_t* pair_code,
                          int32_t* enc_pair, int32_t* enc_i, int32_t* enc_j, int32_t* enc_tile /* [3][capacity] */,
                          int32_t* diff_pair, int32_t* diff_i, int32_t* diff_j, int32_t* diff_tile /* [3][capacity] */,
                          int32_t* keys, int32_t* counts_dev, int32_t* scratch, void* stream);
/* Per-bucket folded matrices of ONE checkpoint: out [num_slots][H*H + H] (packed [k/4][out][k%4], then the bias),
 * slot s from keys_dev[s] (enc slots, then diff slots: pass the concatenated key list).  fp64 accumulation. */
size_t tsd_bucket_weights_floats(const tsd_model_cfg* cfg, int32_t num_slots);
int tsd_bucket_weights_build(const tsd_model_cfg* cfg, const float* packed_weights, int32_t num_slots,
                             const int32_t* keys_dev, float* out, void* stream);
/* f16-plane image of a bucket arena (same size and slot layout, biases copied): see tsd_pack_weights16 */
int tsd_bucket_weights16(const tsd_model_cfg* cfg, const float* bucket_weights, int32_t num_slots, float* out16,
                         void* stream);

/* ---- whole forward for M checkpoints ---------------------------------------------------- */
typedef struct tsd_batch {
    int32_t num_nodes, num_graphs, num_pairs, num_models;
    const int32_t* graph_ptr;   /* [G+1] */
    const int32_t* node_graph;  /* [N] */
    const int32_t* pair_ptr;    /* [N+1] */
    const uint16_t* pair_code;  /* [P] */
    const float* weights;       /* [M, packed_floats] */
    const float* z;             /* [M, N, H] node embeddings (tsd_node_embed), pos independent */
    const float* x1_0;          /* [M, N, H] lin1 of block 0 applied to z (tsd_node_lin1), pos independent */
    tsd_geometry geo;
    float* workspace;           /* tsd_forward_workspace_floats */
    float* edge_inv_u;          /* [M, P/2] per-checkpoint output on the undirected out list */
    int32_t max_graph_nodes;    /* atoms of the largest graph (host knowledge), or 0 = unknown: the sampling loop then
                                   runs its step tail as three launches instead of the fused one (<= 64-atom graphs) */
    int32_t reserved;           /* flags; bit 0: run the forward as one launch per block even where the one-launch form
                                   applies (also what a caller sets after a TSD_STATUS_INTERNAL report: that form has no
                                   in-kernel waits); bit 1: 32-row filter and pair tiles also where a split-f16 launch
                                   (block launch, one-launch forward, pair output) would take 64-row ones (A/B and
                                   cross-check switches; results are bit-identical); bit 2: no
                                   fused per-unit encoder (kernels_unit.hip) where it would apply: one launch per block
                                   with materialised filters instead (bit-identical results); bit 4: the fused
                                   encoder also where the one-launch form would apply (tests, A/B);
                                   bit 3 (tests only): fault injection -- the one-launch forward skips its last filter
                                   tile, so that one bounded wait gives up and TSD_STATUS_INTERNAL is reported;
                                   bit 5 (tsd_train_forward / tsd_train_backward2, hidden = 256, `status` set): the training
                                   step's tile GEMMs on split-f16 operands -- gradient operands scaled by exact powers of
                                   two, fp32 accumulation and saved activations; the forward raises TSD_STATUS_RANGE in
                                   `status` when an activation left the f16 range, tsd_train_backward2 then returns
                                   TSD_ERR_RANGE.  Pass the same bit to both calls of a step; bit 6 (with bit 5): the
                                   backward keeps its small gradient launches (embedding tables, narrow layers) on the
                                   caller's stream instead of the library's side stream (A/B switch, same results);
                                   bit 7 (tsd_train_forward, 0.7): the edge lists of `geo` are current for the call's `pos`
                                   (tsd_geometry_build ran on them, e.g. ahead on a side stream) and counts_host[0, 1, 3]
                                   (+ [2]: topology status) hold their counts: the call neither rebuilds them nor waits for
                                   the device */
    tsd_typed_tiles enc_tiles, diff_tiles;  /* static type-sorted embedding tiles, or num_tiles = 0: generic embedding */
    const float* bucket_weights;            /* [M][(enc + diff buckets) * (H*H + H)] (tsd_bucket_weights_build) or NULL */
    /* ---- appended in 0.4: the split-f16 inference forward (see tsd_pack_weights16) ---- */
    const float* weights16;                 /* [M, packed_floats] f16-plane image of `weights` (tsd_pack_weights16), or NULL:
                                               the forward runs on the fp32-input MFMA */
    const float* bucket_weights16;          /* [M][...] f16-plane image of `bucket_weights` (tsd_bucket_weights16); needed
                                               with weights16 when bucket_weights is set */
    int32_t* status;                        /* device word that receives TSD_STATUS_RANGE (sticky, OR-ed; the caller zeroes
                                               and reads it); may be NULL: no range report.  In the sampling loop the
                                               state block's flags word is used instead */
    /* ---- appended in 0.5: the fused per-unit encoder (csrc/kernels_unit.hip) ---- */
    const int32_t* unit_node;               /* [num_units + 1] node offsets of the UNITS of the batch: consecutive runs of whole
                                               graphs with at most TSD_UNIT_MAX_NODES atoms each, covering [0, N) (host
                                               knowledge: graphs never interact, so any such partition is valid; the Python
                                               host balances it by pair count).  NULL: no fused encoder */
    int32_t num_units;
    int32_t reserved2;                      /* 0 (0.5 selected experimental two-team forms of the fused encoder here; they were
                                               removed in 0.6: measured equal to the general kernel, docs/NOTEBOOK.md) */
} tsd_batch;

size_t tsd_forward_workspace_floats(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_pairs,
                                    int32_t num_models);
/* Where a forward leaves its intermediates inside `workspace` (tests, debugging tools, integrators that want the
 * node states): float offsets out[0..4] = edge attributes [M][P, H], CFConv filter slots [M][slots, P/2, H], node states h
 * [M][N, H] (the encoder's output), the two x1 buffers; out[5] = per-checkpoint stride of the [N, H] arrays, out[6] =
 * number of filter slots, out[7] = tsd_forward_workspace_floats.  Same device as the forward (the layout of small
 * batches depends on the device's workgroup slots). */
int tsd_forward_workspace_layout(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_pairs, int32_t num_models,
                                 size_t* out /* [8] host */);
/* geometry + M forwards; edge_inv_u[m] valid for the first *geo.out_u.count entries. */
int tsd_score_forward(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* pos, void* stream);

/* Measurement entry (bench.py's roofline of the dominant kernel): re-runs ONLY the one-launch kernel of the split-f16
 * forward -- the L interaction blocks (models/encoder/schnet.py:203-225) and the pair MLP (models/common.py:226-229) --
 * on the state the last tsd_score_forward of this batch left in the workspace (edge attributes, block-0 filters, edge
 * lists).  `epoch` numbers the calls 1, 2, 3, ... (the hand-off words of the launch are monotonic; call 1 zeroes them).
 * TSD_ERR_UNSUPPORTED when the batch does not take that path (no weights16 / status, several checkpoints -- an ensemble's
 * one-launch forward, groups of checkpoints in one grid, runs inside tsd_score_forward only --, > 256 node tiles). */
int tsd_forward_blocks(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t epoch, void* stream);

/* Measurement / cross-check entry of the fused per-unit encoder (csrc/kernels_unit.hip; models/encoder/schnet.py:74-128,
 * 203-225): runs interaction blocks [l_begin, l_end) of every bound checkpoint as ONE launch -- one workgroup per unit
 * (tsd_batch.unit_node) computes the CFConv filters of its pairs tile by tile, accumulates both end points' messages in
 * registers and runs the node chain on its <= 64 rows; the filters are never written to memory -- on the attribute rows
 * and edge lists the last tsd_score_forward of this batch left in the workspace.  l_begin = 0 starts from z / x1_0;
 * a later start reads h and x1 of block l_begin from the workspace (what a run that ended at l_begin left there).
 * TSD_ERR_UNSUPPORTED when the batch does not take that path (no weights16 / unit partition, hidden != 256). */
int tsd_forward_encoder(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t l_begin, int32_t l_end, void* stream);

/* mean over checkpoints in the reference's order, expanded to the directed out list:
 * edge_inv[e] = ((inv_u[0][u] + inv_u[1][u]) + ...)/M with u = out.umap[e]   -> edge_inv [P] */
int tsd_ensemble_mean(int32_t num_models, int32_t num_pairs, tsd_edges out,
                      const float* edge_inv_u /* [M, P/2] */, float* edge_inv, void* stream);

/* eq_transform on the library's own out-edge list (deterministic, no atomics):
 * score[i] = sum_{e in row i} u_e s_e + sum_{e in row i} u_e s_{(dst e, i)}. */
int tsd_eq_transform_rows(int32_t num_nodes, const float* pos, const int32_t* pair_ptr,
                          const int32_t* graph_ptr, const int32_t* node_graph,
                          tsd_edges out, const int32_t* pair2out, const float* score_d,
                          float* score_pos, void* stream);

/* ---- sampler ------------------------------------------------------------------------- */
/* per-step coefficients, computed by the host exactly as the reference's fp32 tensor ops do:
 * LD   (kind 0): c[0]=step_size, c[1]=sigma_i, c[2]=sqrt(2*step_size)
 * DDPM (kind 1): c[0]=sqrt(at), c[1]=sqrt(1/at), c[2]=sqrt(1/at-1), c[3]=sqrt(atm1)*beta_t,
 *                c[4]=sqrt(1-beta_t)*(1-atm1), c[5]=1-at, c[6]=mask*exp(0.5*log(beta_t)), c[7]=sqrt(atm1) */
#define TSD_STEP_COEFS 8
int tsd_sampler_step(int32_t kind, int32_t num_nodes, int32_t num_graphs, const int32_t* graph_ptr,
                     const float* score_pos, const float* noise, const float* coefs /* [8] device */,
                     float clip, float clip_pos /* <0: none */, float* pos, int32_t* status, void* stream);

/* ---- the device-resident sampling loop (reference models/sampler.py:187-254) --------------------
 * One step = geometry lists, M forwards, ensemble mean, eq_transform, clip, LD/DDPM update, NaN flag,
 * centring, and the member counts of the next step's lists.  A *plan* is a host object holding ONE such step
 * captured into a hipGraph and instantiated; it is created once per (batch, bound checkpoints, kind, clip)
 * and replayed by every later call, so that a call costs its launches only (r01: capture + instantiate +
 * destroy were ~11 ms inside every call).  Everything a call may change is read by the captured kernels
 * from `tsd_sampler_state`, a 64-byte block of DEVICE memory owned by the caller. */
typedef struct tsd_run_args {
    const float* coefs;   /* [n_steps, TSD_STEP_COEFS] */
    const float* noises;  /* [n_steps, N, 3] injected Gaussian draws (parity tests), or NULL: the draws of
                             sampler.py:213 are generated on the device, Philox4x32-10 keyed by `seed`,
                             counter = offset + step * N + atom, Box-Muller -> 3 of 4 normals */
    float* traj;          /* [n_steps, N, 3] or NULL */
    uint64_t seed;
    uint64_t offset;
} tsd_run_args;
typedef struct tsd_sampler_state {
    int32_t flags;        /* TSD_STATUS_* bits, sticky; the caller zeroes and reads them */
    int32_t step;         /* device-side step counter (row of coefs / noises / traj) */
    int32_t reserved[2];
    tsd_run_args args;    /* written by tsd_sampler_plan_run */
    int32_t pad[2];
} tsd_sampler_state;
typedef struct tsd_sampler_plan tsd_sampler_plan;

/* Captures the step on `stream` (must not be the legacy default stream) into graphs of 8, 4, 2 and 1 consecutive steps
 * (0.7: all of them here, so that no later call pays for a capture; a run of n steps is n / 8 launches of the longest and at
 * most one each of the others); nothing executes.
 * `batch`'s arrays, `pos` [N,3] (updated in place by every step) and `state` must stay alive and unmoved
 * for the life of the plan.  kind 0 = LD, 1 = DDPM; clip_pos < 0: no clamp. */
int tsd_sampler_plan_create(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t kind, float clip,
                            float clip_pos, float* pos, tsd_sampler_state* state, void* stream,
                            tsd_sampler_plan** plan_out);
/* n_steps steps from the current contents of `pos`; asynchronous, no host sync.  use_graph == 0 launches the
 * same kernels eagerly (bit-identical; tests).  Calls on one plan must be stream-ordered. */
int tsd_sampler_plan_run(tsd_sampler_plan* plan, int32_t n_steps, const tsd_run_args* args, int32_t use_graph,
                         void* stream);
/* The caller must have synchronised with every stream the plan was run on. */
void tsd_sampler_plan_destroy(tsd_sampler_plan* plan);

/* One-shot form: a plan that captures only the graphs this call's n_steps needs + plan_run + stream synchronise +
 * plan_destroy (use the plan calls to keep the graphs across calls).  noises NULL: device Philox with (seed, offset). */
int tsd_sampler_run(const tsd_model_cfg* cfg, const tsd_batch* batch, int32_t kind, int32_t n_steps,
                    const float* coefs, const float* noises, uint64_t seed, uint64_t offset, float clip,
                    float clip_pos, float* pos, float* traj, tsd_sampler_state* state, int32_t use_graph,
                    void* stream);
/* The normal draws of the device generator alone: out[k] for k in [0, 3 n): atom i = k / 3 of counter
 * offset + i (what step s of a run with N atoms uses at offset + s * N).  Tests / reproducibility. */
int tsd_philox_normal(uint64_t seed, uint64_t offset, int64_t n_atoms, float* out /* [n_atoms, 3] */, void* stream);

/* ---- training primitives (BASELINE config 4; reference train.py:124-152, condensenc.py:267-328) -----
 * One launch per operation: every dense layer's forward / dgrad / wgrad on the fp32 MFMA, the graph-shaped
 * operations and their adjoints as plain kernels.  tsd_train_forward / tsd_train_backward (further down)
 * sequence them for a whole step; the Python host also exposes them one by one as torch.autograd.Function
 * nodes (tsdiff_amd/train_ops.py: the differentiable forward(), the legacy network, cross-checks).
 * All matrices row-major fp32. */
/* Y[rows,out] = X[rows,in] W[out,in]^T + b.  scratch >= in*out floats: W is packed there and the product
 * runs on the fp32 MFMA (in, out in {128,256,512}); otherwise / without scratch a plain VALU kernel runs. */
int tsd_linear_fwd(int32_t rows, int32_t in, int32_t out, const float* X, const float* W, const float* b,
                   float* Y, float* scratch, size_t scratch_floats, void* stream);
/* The optimizer rewrites every weight every step: pack all dense weights of a step with ONE launch.
 * Item k: W[k] [out_dim k, in_dim k] row major -> dst[k] (out*in floats) in the forward layout
 * (transposed[k] == 0, for tsd_linear_fwd_packed) or the dgrad layout (transposed[k] != 0, `Wp_t` of
 * tsd_linear_bwd).  W / dst / the three int arrays are HOST arrays of n entries (device pointers inside).
 * tsd_linear_packable: 1 if (in, out) runs on the MFMA kernels (both in {128, 256, 512}). */
int tsd_linear_packable(int32_t in, int32_t out);
int tsd_pack_linear_batch(int32_t n, const float* const* W_host, float* const* dst_host, const int32_t* out_dim_host,
                          const int32_t* in_dim_host, const int32_t* transposed_host, void* stream);
int tsd_linear_fwd_packed(int32_t rows, int32_t in, int32_t out, const float* X, const float* Wp, const float* b,
                          float* Y, void* stream);
/* dX = dY W (NULL: skip); dW = dY^T X (NULL: skip); db = column sums of dY (NULL: skip).
 * Wp_t: W packed in the dgrad layout (tsd_pack_linear_batch) or NULL (packed here into scratch).
 * scratch layout: 64*out floats (bias partials) | in*out floats (W packed for the MFMA dgrad) |
 * 64*out*in floats (row-split wgrad partials, summed in a fixed order: deterministic).
 * With less scratch, or shapes that are not multiples of 128, plain VALU kernels run instead. */
int tsd_linear_bwd(int32_t rows, int32_t in, int32_t out, const float* X, const float* W, const float* Wp_t,
                   const float* dY, float* dX, float* dW, float* db, float* scratch, size_t scratch_floats,
                   void* stream);
/* kind 0: swish (utils/activation_functions.py), 1: shifted softplus (schnet.py:65-71), 2: ReLU, 3: softplus;
 * x = pre-activation */
int tsd_act_fwd(int32_t kind, int64_t n, const float* x, float* y, void* stream);
int tsd_act_bwd(int32_t kind, int64_t n, const float* x, const float* dy, float* dx, void* stream);
/* y[r,:] = x[r,:] * emb[idx[r],:] (edge.py:66-68); backward also accumulates demb (zeroed by the caller) */
int tsd_emb_mul_fwd(int32_t rows, int32_t H, const float* x, const float* emb, const uint8_t* idx, float* y,
                    void* stream);
int tsd_emb_mul_bwd(int32_t rows, int32_t H, const float* x, const float* emb, const uint8_t* idx,
                    const float* dy, float* dx, float* demb, void* stream);
/* y[r,:] = table[idx[r],:] ; dtable[idx[r],:] += dy[r,:] (atom_embedding, condensenc.py:193) */
int tsd_gather_rows(int32_t rows, int32_t H, const float* table, const int64_t* idx, float* y, void* stream);
int tsd_scatter_rows_add(int32_t rows, int32_t H, const float* dy, const int64_t* idx, float* dtable,
                         void* stream);
/* x[r,:] *= C(dist[r]): CFConv cutoff weight (hard mask, or cosine when smooth != 0), forward and backward
 * (schnet.py:92-99) */
int tsd_row_mask(int32_t rows, int32_t H, const float* dist, float cutoff, int32_t smooth, float* x, void* stream);
/* adjoint of tsd_cfconv_aggregate w.r.t. the filter: dWf[u] = dagg[i]*x1[j] + dagg[j]*x1[i]
 * (w.r.t. x1 it is tsd_cfconv_aggregate itself with dagg in place of x1) */
int tsd_aggregate_bwd_filter(int32_t H, int32_t capacity_u, tsd_edges enc_u, const float* dagg, const float* x1,
                             float* dWf, void* stream);
/* p[u,:] = h[src u,:] * h[dst u,:] (common.py:226-229) and dh[i,:] = sum_{e in row i} dp[umap e,:] * h[dst e,:] */
int tsd_pair_product_fwd(int32_t H, int32_t capacity_u, tsd_edges out_u, const float* h, float* p, void* stream);
int tsd_pair_product_bwd(int32_t num_nodes, int32_t H, tsd_edges out, const float* dp, const float* h, float* dh,
                         void* stream);
/* eq_transform (geometry.py:22-30) of a per-undirected-pair score and its adjoint w.r.t. the score */
int tsd_eq_und_fwd(int32_t num_nodes, tsd_edges out, const float* pos, const float* s_u, float* node_eq,
                   void* stream);
int tsd_eq_und_bwd(int32_t capacity_u, tsd_edges out_u, const float* pos, const float* g, float* ds_u,
                   void* stream);
/* d[u] = |pos[src u] - pos[dst u]| (get_distance on another geometry, condensenc.py:313) */
int tsd_pair_distance(int32_t capacity_u, tsd_edges list_u, const float* pos, float* d, void* stream);

/* ---- legacy dual-encoder pieces (secondary: no shipped TSDiff config runs them) ----
 * GINEConv message + sum aggregation + self term (reference models/encoder/gin.py:61-73):
 *   out[i] = (1+eps) x[i] + sum_{e: edge_index[1][e] = i} act(x[edge_index[0][e]] + edge_attr[e])
 * activation 0 none, 1 ReLU, 2 softplus; arbitrary directed int64 edge list; fp32 atomics. */
int tsd_gine_aggregate(int32_t num_nodes, int64_t num_edges, int32_t H, int32_t activation, float eps,
                       const float* x, const int64_t* edge_index, const float* edge_attr, float* out,
                       void* stream);
/* ---- the training step as two calls (reference train.py:124-152, condensenc.py:267-328) -------------
 * `raw`: all trainable parameters as ONE flat fp32 vector in the order of tsd_pack_weights' `raw` argument
 * (no padding; tsd_train_raw_floats floats).  tsd_train_forward builds the geometry of `pos` (the perturbed
 * positions), evaluates the network and loss[i] = |eq_transform(edge_inv) - eq_transform(d_target)|^2 per node
 * (d_target from the clean positions `pos0` and the per-graph alpha `a_graph` [G]), and keeps every
 * activation in `workspace` (tsd_train_workspace_floats(cfg, N, P) floats).  counts_host [4] (HOST memory)
 * receives the undirected edge counts of the encoder and output lists, the topology status word and the number of
 * output edges embedded separately (the call synchronises the stream once to read them; a fresh batch needs no
 * other host read) and must be
 * handed unchanged to tsd_train_backward, which turns dloss [N] = d(objective)/d(loss) into
 * grad [tsd_train_raw_floats], the gradient of every parameter in the layout of `raw` (grad and workspace 16-byte
 * aligned). */
/* The forward diffusion of get_loss (condensenc.py:292-297) in one launch: a_graph[g] = alphas[time_step[g]],
 * pos_perturbed[i] = pos[i] + noise[i] * sqrt(1 - a) / sqrt(a) with a = a_graph[node_graph[i]] -- the operations of
 * the reference's expression in its order (eight elementwise / gather launches there).  time_step [G] and
 * node_graph [N] are int64 as the reference holds them (indices outside their tables are clamped, not reported);
 * the draws (time_step, noise) stay the caller's. */
int tsd_diffuse_positions(int32_t num_nodes, int32_t num_graphs, int32_t num_timesteps, const float* alphas,
                          const int64_t* time_step, const int64_t* node_graph, const float* pos, const float* noise,
                          float* pos_perturbed /* [N,3] */, float* a_graph /* [G] */, void* stream);
/* The four host words of tsd_train_forward's `counts_host` for edge lists built ahead (tsd_geometry_build on `geo`, e.g. on a
 * side stream beside the previous step): [enc_u count, out_u count, *topo_status (or 0), diff_u count] written by one small
 * kernel into PINNED, device-visible host memory `counts_pinned` [4] on `stream`.  Record an event behind the call and read
 * the words when it has fired; hand them to tsd_train_forward with tsd_batch.reserved bit 7 (0.7). */
int tsd_geometry_counts_async(tsd_geometry geo, const int32_t* topo_status, int32_t* counts_pinned /* [4], pinned host */,
                              void* stream);
size_t tsd_train_raw_floats(const tsd_model_cfg* cfg);
size_t tsd_train_workspace_floats(const tsd_model_cfg* cfg, int32_t num_nodes, int32_t num_pairs);
int tsd_train_forward(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* raw, const int64_t* atom_type,
                      const int64_t* r_feat, const int64_t* p_feat, const float* pos0, const float* pos,
                      const float* a_graph, const int32_t* topo_status /* tsd_topology_build's status word or NULL */,
                      float* workspace, size_t workspace_floats, float* loss,
                      int32_t* counts_host /* [4] */, void* stream);
int tsd_train_backward(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* raw, const int64_t* atom_type,
                       const float* pos, float* workspace, size_t workspace_floats, const int32_t* counts_host,
                       const float* dloss, float* grad, void* stream);

/* The same call for a data-parallel caller (reference train.py:138-145 on N GPUs): `blocks_done_event` (a hipEvent_t,
 * or NULL) is recorded on `stream` as soon as every gradient of the interaction blocks is final -- the contiguous range
 * tsd_train_grad_buckets reports ({offset, count, total} floats of the flat gradient: 83 % of it for the shipped config) --
 * so that their all-reduce can start on another stream beside the embedding's backward chain; the rest of the vector is
 * final when the call's last kernel is. */
int tsd_train_grad_buckets(const tsd_model_cfg* cfg, size_t* out /* [3] host */);
int tsd_train_backward2(const tsd_model_cfg* cfg, const tsd_batch* batch, const float* raw, const int64_t* atom_type,
                        const float* pos, float* workspace, size_t workspace_floats, const int32_t* counts_host,
                        const float* dloss, float* grad, void* blocks_done_event, void* stream);

/* ---- optimizer step on the flat vectors (reference train.py:144-145, utils/common.py:58-68) ---------------
 * tsd_grad_norm_clip: norm[0] = |grad|_2 (fixed-order two-stage sum, deterministic), then -- max_norm > 0 --
 * grad *= min(1, max_norm / (norm + 1e-6)), torch.nn.utils.clip_grad_norm_'s rule; scratch: 1024 floats.
 * tsd_adam_step: torch.optim.Adam's update (no amsgrad) of all n parameters in one launch; `step` counts from 1. */
int tsd_grad_norm_clip(int64_t n, float* grad, float max_norm, float* scratch, float* norm, void* stream);
int tsd_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, void* stream);

/* ---- dual-encoder network, reference models/epsnet/dualenc.py ----------------
 * GINE message pass of the local head over the rows of the symmetric extended edge list `enc`,
 * restricted to the local edges (enc.type_r > 0, dualenc.py:1222-1223), edge attributes once per
 * undirected pair (ea_u [capacity_u, H], indexed through enc.umap):
 *   out[i] = sum_{e in row i, local} act(x[dst e] + ea_u[umap e]) + (1 + eps) x[i]     gin.py:61-73
 * and its adjoints (dx [N,H], dea_u [capacity_u,H]; either may be NULL). Deterministic, no atomics. */
int tsd_gine_csr_fwd(int32_t num_nodes, int32_t H, int32_t activation, float eps, tsd_edges enc,
                     const float* ea_u, const float* x, float* out, void* stream);
int tsd_gine_csr_bwd(int32_t num_nodes, int32_t capacity_u, int32_t H, int32_t activation, float eps, tsd_edges enc,
                     tsd_edges enc_u, const float* ea_u, const float* x, const float* dout, float* dx, float* dea_u,
                     void* stream);
/* torch.nn.Embedding(max_norm) side effect of the global SchNet's node embedding (schnet.py:151): rows
 * table[idx[k]] with 2-norm > max_norm are rescaled in place by max_norm / (norm + 1e-7).
 * scratch: num_rows int32. */
int tsd_embedding_renorm(int32_t num_rows, int32_t H, int32_t n, const int64_t* idx, float max_norm, float* table,
                         int32_t* scratch, void* stream);
/* eps_pos of the dual-encoder sampler (dualenc.py:826-849):
 *   out = clip_norm(eq_local, clip_local) + clip_norm(eq_global, clip_global) * w_global
 * a negative clip disables that clip; eq_global NULL drops the global term. */
int tsd_dual_score(int32_t num_nodes, const float* eq_local, const float* eq_global, float clip_local,
                   float clip_global, float w_global, float* out, void* stream);

/* GaussianSmearingEdgeEncoder (models/encoder/edge.py:18-41): the radial-basis expansion
 *   out[e] = [exp(coeff (d_e - offset_k)^2) for k < K, bond_emb[type_e]]   -> [E, 2K] */
int tsd_gaussian_edge_encode(int64_t num_edges, int32_t K, float coeff, const float* d, const float* offset,
                             const int64_t* type, const float* bond_emb, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSDIFF_HIP_H */

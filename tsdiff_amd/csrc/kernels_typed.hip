// kernels_typed.hip -- the edge embedding on static TYPE-SORTED tiles (round 3; include/tsdiff_hip.h, tsd_typed_tiles).
//
// reference: models/encoder/edge.py:58-68 (mlp(d) * bond_emb[type]) and models/epsnet/condensenc.py:156-176,105-115
// (edge_cat over the reactant / product halves).  For a fixed (type_r, type_p) the chain
//   Linear(H,H) -> * emb[type_r] || * emb[type_p] -> Linear(2H,H)
// is ONE H x H matrix Wt (three GEMM units of an embedded edge become one).  Edge types are topology, so the
// candidate pairs of a batch are bucketed by type pair once (tsd_typed_tiles_build: histogram, scan, scatter -- no
// sort, the order inside a bucket is irrelevant: a row's result depends on that row only) and the per-bucket matrices
// are folded once per (batch, checkpoint) with fp64 accumulation (tsd_bucket_weights_build).
#include "train_internal.hpp"
#include "typed_tile.hpp"

namespace tsd {

constexpr int T = TSD_EDGE_TILE;
constexpr int NKEY = 1024;  // type_r * 32 + type_p, types < 32

__device__ __forceinline__ int tt_type_of(int bond, int hop, int order) {  // kernels_graph.hip::type_of
    return bond ? bond : ((hop >= 2 && hop <= order) ? (TSD_NUM_BOND_TYPES + hop - 1) : 0);
}
__device__ __forceinline__ void tt_keys(int code, int order_enc, int order_out, int& key_enc, int& key_out) {
    const int bR = code & 31, bP = (code >> 5) & 31, hR = (code >> 10) & 7, hP = (code >> 13) & 7;
    key_enc = tt_type_of(bR, hR, order_enc) * 32 + tt_type_of(bP, hP, order_enc);
    key_out = tt_type_of(bR, hR, order_out) * 32 + tt_type_of(bP, hP, order_out);
}

// pass 0: histogram of the bucket keys; pass 1: scatter to bin_off[key] + cursor++.  One wave per row (atom i), its
// lanes over the row's pairs with j > i.
template <int PASS>
__global__ __launch_bounds__(256) void typed_pairs_kernel(int N, const int32_t* __restrict__ graph_ptr,
                                                          const int32_t* __restrict__ node_graph,
                                                          const int32_t* __restrict__ pair_ptr,
                                                          const uint16_t* __restrict__ pair_code, int order_enc,
                                                          int order_out, int32_t* __restrict__ hist /* [2][NKEY] */,
                                                          const int32_t* __restrict__ bin_off /* [2][NKEY] */,
                                                          int32_t* __restrict__ cursor /* [2][NKEY] */,
                                                          int32_t* __restrict__ enc_pair, int32_t* __restrict__ enc_i,
                                                          int32_t* __restrict__ enc_j, int32_t* __restrict__ diff_pair,
                                                          int32_t* __restrict__ diff_i, int32_t* __restrict__ diff_j) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int lo = graph_ptr[node_graph[i]];
    const int il = i - lo;
    const int p0 = pair_ptr[i], np = pair_ptr[i + 1] - p0;
    for (int k = il + lane; k < np; k += 64) {  // k >= il  <=>  j = lo + k + 1 > i
        const int j = lo + k + 1;
        int ke, ko;
        tt_keys(pair_code[p0 + k], order_enc, order_out, ke, ko);
        if (PASS == 0) {
            atomicAdd(hist + ke, 1);
            if (ko != ke) atomicAdd(hist + NKEY + ko, 1);
        } else {
            const int s = bin_off[ke] + atomicAdd(cursor + ke, 1);
            enc_pair[s] = p0 + k;
            enc_i[s] = i;
            enc_j[s] = j;
            if (ko != ke) {
                const int q = bin_off[NKEY + ko] + atomicAdd(cursor + NKEY + ko, 1);
                diff_pair[q] = p0 + k;
                diff_i[q] = i;
                diff_j[q] = j;
            }
        }
    }
}

// one workgroup of NKEY threads per list (blockIdx.x = list): bin -> entry offset, tile offset, slot; tile tables
__global__ __launch_bounds__(NKEY) void typed_scan_kernel(const int32_t* __restrict__ hist, int32_t* __restrict__ bin_off,
                                                          int32_t* __restrict__ keys, int32_t* __restrict__ enc_tile,
                                                          int32_t* __restrict__ diff_tile, int cap,
                                                          int32_t* __restrict__ counts) {
    __shared__ int s_ent[NKEY], s_til[NKEY], s_slot[NKEY], s_w[3][16];
    const int l = blockIdx.x, b = threadIdx.x, lane = b & 63, wave = b >> 6;
    const int c = hist[l * NKEY + b];
    int v[3] = {c, (c + T - 1) / T, c > 0 ? 1 : 0};
    int inc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int x = v[q];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(x, off);
            if (lane >= off) x += u;
        }
        inc[q] = x;
        if (lane == 63) s_w[q][wave] = x;
    }
    __syncthreads();
    int ex[3], tot[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int base = 0, all = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) base += s_w[q][w];
            all += s_w[q][w];
        }
        ex[q] = base + inc[q] - v[q];
        tot[q] = all;
    }
    s_ent[b] = ex[0];
    s_til[b] = ex[1];
    s_slot[b] = ex[2];
    bin_off[l * NKEY + b] = ex[0];
    if (c > 0) keys[l * NKEY + ex[2]] = b;
    // the diff list's slots follow the enc list's in the bucket-weight arena
    __shared__ int s_slot0;
    if (b == 0) s_slot0 = 0;
    __syncthreads();
    if (b == 0) {
        counts[2 * l] = tot[1];
        counts[2 * l + 1] = tot[2];
    }
    // (slot offset of the diff list = number of enc buckets: recomputed from the enc histogram by this workgroup)
    int enc_buckets = 0;
    if (l == 1) {
        int mine = hist[b] > 0 ? 1 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0) atomicAdd(&s_slot0, mine);
        __syncthreads();
        enc_buckets = s_slot0;
    }
    int32_t* tile = l == 0 ? enc_tile : diff_tile;
    const int ntiles = tot[1];
    for (int t = b; t < ntiles; t += NKEY) {  // bin of tile t: the last bin whose tile offset is <= t and that has tiles
        int lo_ = 0, hi_ = NKEY - 1;
        while (lo_ < hi_) {
            const int mid = (lo_ + hi_ + 1) >> 1;
            if (s_til[mid] <= t) lo_ = mid; else hi_ = mid - 1;
        }
        // (bins without tiles share their offset with the next bin that has some; the LAST bin with offset <= t is the
        // one that owns t: every later bin starts past t)
        const int bin = lo_;
        const int k = t - s_til[bin];
        if (t < cap) {
            tile[t] = enc_buckets + s_slot[bin];
            tile[cap + t] = s_ent[bin] + k * T;
            tile[2 * cap + t] = min(T, hist[l * NKEY + bin] - k * T);
        }
    }
}

size_t typed_tiles_capacity(int P) { return (size_t)(P / 2 + T - 1) / T + NKEY; }

int launch_typed_tiles_build(const tsd_model_cfg& c, int N, int P, const int32_t* graph_ptr, const int32_t* node_graph,
                             const int32_t* pair_ptr, const uint16_t* pair_code, int32_t* enc_pair, int32_t* enc_i,
                             int32_t* enc_j, int32_t* enc_tile, int32_t* diff_pair, int32_t* diff_i, int32_t* diff_j,
                             int32_t* diff_tile, int32_t* keys, int32_t* counts, int32_t* scratch, hipStream_t st) {
    int32_t *hist = scratch, *bin_off = scratch + 2 * NKEY, *cursor = scratch + 4 * NKEY;
    TSD_HIP(hipMemsetAsync(scratch, 0, (size_t)6 * NKEY * sizeof(int32_t), st));
    TSD_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(int32_t), st));
    if (N == 0 || P == 0) return TSD_OK;
    const int cap = (int)typed_tiles_capacity(P);
    hipLaunchKernelGGL(typed_pairs_kernel<0>, dim3((N + 3) / 4), dim3(256), 0, st, N, graph_ptr, node_graph, pair_ptr,
                       pair_code, c.edge_order, c.pred_edge_order, hist, bin_off, cursor, enc_pair, enc_i, enc_j,
                       diff_pair, diff_i, diff_j);
    hipLaunchKernelGGL(typed_scan_kernel, dim3(2), dim3(NKEY), 0, st, hist, bin_off, keys, enc_tile, diff_tile, cap,
                       counts);
    hipLaunchKernelGGL(typed_pairs_kernel<1>, dim3((N + 3) / 4), dim3(256), 0, st, N, graph_ptr, node_graph, pair_ptr,
                       pair_code, c.edge_order, c.pred_edge_order, hist, bin_off, cursor, enc_pair, enc_i, enc_j,
                       diff_pair, diff_i, diff_j);
    TSD_LAUNCH_CHECK("typed_tiles_build");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// per-bucket folded matrices: Wt = (Wc0[:, :H] diag(emb[tr]) + Wc0[:, H:] diag(emb[tp])) W1,  bt = (...) b1 + bc0
// read from the PACKED arena ([k/4][out][k%4]: the same fp32 values as the raw tensors), accumulated in fp64
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float packed_at(const float* __restrict__ Bp, int nout, int o, int k) {
    return Bp[((size_t)(k >> 2) * nout + o) * 4 + (k & 3)];
}
__global__ void bucket_weights_kernel(int H, const float* __restrict__ bond_emb, const float* __restrict__ w1p,
                                      const float* __restrict__ b1, const float* __restrict__ cw0p,
                                      const float* __restrict__ cb0, const int32_t* __restrict__ keys,
                                      float* __restrict__ out) {
    const int slot = blockIdx.y;
    const int key = keys[slot];
    const float* er = bond_emb + (size_t)(key >> 5) * H;
    const float* ep = bond_emb + (size_t)(key & 31) * H;
    float* dst = out + (size_t)slot * ((size_t)H * H + H);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < H * H) {
        const int s = idx & 3, o = (idx >> 2) % H, k = ((idx >> 2) / H) * 4 + s;
        double acc = 0.0;
        for (int j = 0; j < H; ++j) {
            const double a = (double)packed_at(cw0p, H, o, j) * (double)er[j] + (double)packed_at(cw0p, H, o, H + j) * (double)ep[j];
            acc += a * (double)packed_at(w1p, H, j, k);
        }
        dst[idx] = (float)acc;
    }
    if (idx < H) {
        double acc = (double)cb0[idx];
        for (int j = 0; j < H; ++j) {
            const double a = (double)packed_at(cw0p, H, idx, j) * (double)er[j] + (double)packed_at(cw0p, H, idx, H + j) * (double)ep[j];
            acc += a * (double)b1[j];
        }
        dst[(size_t)H * H + idx] = (float)acc;
    }
}

int launch_bucket_weights(const tsd_model_cfg& c, const float* W, int num_slots, const int32_t* keys, float* out,
                          hipStream_t st) {
    if (num_slots <= 0) return TSD_OK;
    const WeightLayout L = weight_layout(c);
    const int H = c.hidden;
    hipLaunchKernelGGL(bucket_weights_kernel, dim3((H * H + 255) / 256, num_slots), dim3(256), 0, st, H, W + L.bond_emb,
                       W + L.emlp_w1, W + L.emlp_b1, W + L.ecat_w0, W + L.ecat_b0, keys, out);
    TSD_LAUNCH_CHECK("bucket_weights");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// the embedding launch on static typed tiles: tiles [0, ta.n) enc list (+ the filters of block 0 when FUSE0),
// [ta.n, ta.n + tb.n) diff list, then the directed -> undirected map role (UmapRole).  The attribute rows written
// hold s1 (the inference forward's folded form, common.hpp FOLDED WEIGHTS).
// ---------------------------------------------------------------------------------------------
template <int H, bool FUSE0>
__global__ __launch_bounds__(2 * H) void typed_embed_kernel(TypedEmbedW w, TypedList ta, TypedList tb,
                                                            const float* __restrict__ pos,
                                                            const int32_t* __restrict__ pair2u, int P,
                                                            const int32_t* __restrict__ attr_row,
                                                            float* __restrict__ edge_attr, size_t out_stride,
                                                            UmapRole um, EmbedFuse0 f0) {
    constexpr int LDA = H + 4, NT = 2 * H, C4 = H / 4;
    const int embed_tiles = ta.n + tb.n;
    if ((int)blockIdx.x >= embed_tiles) {  // extra role: directed-edge -> undirected-pair map (checkpoint 0 only)
        if (blockIdx.y == 0) {
            const int t = ((int)blockIdx.x - embed_tiles) * NT + (int)threadIdx.x;
            edge_umap_body(um.g, um.graph_ptr, um.node_graph, um.pair_ptr, um.P, t);
            if (t < um.n_zero) um.zero_words[t] = 0;
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf = smem;
    float* s_d = smem + T * LDA;
    int* s_row = reinterpret_cast<int*>(s_d + T);
    const bool second = (int)blockIdx.x >= ta.n;
    const TypedList& tl = second ? tb : ta;
    const int t = second ? (int)blockIdx.x - ta.n : (int)blockIdx.x;
    const int PU = P / 2;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32, col = col0 + l31;
    const size_t m = blockIdx.y;
    const int slot = tl.slot[t], start = tl.start[t], nrows = tl.count[t];
    const float* Wt = w.bucket + m * w.bstride + (size_t)slot * ((size_t)H * H + H);
    const float* bt = Wt + (size_t)H * H;
    const float *w0 = w.w0 + m * w.wstride, *b0 = w.b0 + m * w.wstride;
    edge_attr += m * out_stride;
    if (tid < T) {
        const int s = start + min(tid, nrows - 1);
        const int p = tl.pair[s];
        const int i = tl.ni[s], j = tl.nj[s];
        int row = -1;
        if (tid < nrows) {
            if (!second) {
                row = pair2u[p];                                  // enc_u index of the pair, -1: not an edge now
            } else {
                const int eo = pair2u[(size_t)P + p];             // out_u index
                const int ar = eo >= 0 ? attr_row[eo] : -1;
                row = ar >= PU ? ar : -1;                         // separately embedded now?
            }
        }
        // edge_length exactly as the list build computes it (kernels_graph.hip::eval_pair_xyz, src = i < j = dst)
        const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        s_d[tid] = sqrtf(d2);
        s_row[tid] = row;
    }
    __syncthreads();
    // (a tile none of whose pairs is an edge at this step: nothing to do)
    {
        int any = 0;
        for (int r = 0; r < nrows; ++r) any |= s_row[r] >= 0;
        if (!any) return;
    }
    {  // Linear(1,H) + swish: thread = (channel, half of the tile's rows)
        const int c = tid % H, r0 = (tid / H) * (T / 2);
        const float ww = w0[c], bb = b0[c];
#pragma unroll 8
        for (int r = r0; r < r0 + T / 2; ++r) buf[r * LDA + c] = swishf(ww * s_d[r] + bb);
    }
    __syncthreads();
    f32x16 acc[1][1];
    zero_acc(acc);
    gemm_tile<1, 1, H>(buf, LDA, Wt, H, col0, acc);
    __syncthreads();
    {
        const float b = bt[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[acc_row(r, hi) * LDA + col] = swishf(acc[0][0][r] + b);
    }
    __syncthreads();
    // s1 (in LDS) is the attribute tile: whole 1-KiB rows to the rows of the edges the pairs are at this step
    for (int idx = tid; idx < nrows * C4; idx += NT) {
        const int r = idx / C4, c4 = idx % C4;
        const int row = s_row[r];
        if (row >= 0)
            store_stream16(edge_attr + (size_t)row * H + c4 * 4, *reinterpret_cast<const f32x4*>(buf + r * LDA + c4 * 4));
    }
    if constexpr (FUSE0) {
        if (second) return;
        // filter role of interaction block 0 on this tile (kernels_combo.hip::filter_role, same order of operations)
        const float *nn0_w = f0.nn0_w + m * w.wstride, *nn0_b = f0.nn0_b + m * w.wstride;
        const float *nn2_w = f0.nn2_w + m * w.wstride, *nn2_b = f0.nn2_b + m * w.wstride;
        float* wf = f0.wf + m * f0.wf_stride;
        zero_acc(acc);
        gemm_tile<1, 1, H>(buf, LDA, nn0_w, H, col0, acc);
        __syncthreads();
        {
            const float b = nn0_b[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) buf[acc_row(r, hi) * LDA + col] = sspf(acc[0][0][r] + b);
        }
        __syncthreads();
        zero_acc(acc);
        gemm_tile<1, 1, H>(buf, LDA, nn2_w, H, col0, acc);
        __syncthreads();
        {
            const float b = nn2_b[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, hi);
                const float cw = row < nrows ? cutoff_weight(s_d[row], f0.conv_cutoff, f0.smooth) : 0.0f;
                buf[row * LDA + col] = (acc[0][0][r] + b) * cw;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < nrows * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            const int row = s_row[r];
            if (row >= 0)
                store_stream16(wf + (size_t)row * H + c4 * 4, *reinterpret_cast<const f32x4*>(buf + r * LDA + c4 * 4));
        }
    }
}


// the embedding launch on the f16 MFMA pipes: typed_tile.hpp::typed_embed_tile_h per workgroup
template <int H, bool FUSE0>
__global__ __launch_bounds__(2 * H) void typed_embed_h_kernel(TypedEmbedW w, TypedList ta, TypedList tb,
                                                              const float* __restrict__ pos,
                                                              const int32_t* __restrict__ pair2u, int P,
                                                              const int32_t* __restrict__ attr_row,
                                                              float* __restrict__ edge_attr, size_t out_stride,
                                                              UmapRole um, EmbedFuse0 f0, int32_t* range_status, int M) {
    constexpr int NT = 2 * H;
    const int embed_tiles = ta.n + tb.n;
    int bx;  // (1-D grid, checkpoint = id % M: common.hpp wg_item_ckpt)
    size_t m;
    wg_item_ckpt(M, bx, m);
    if (bx >= embed_tiles) {  // extra role: directed-edge -> undirected-pair map (checkpoint 0 only)
        if (m == 0) {
            const int t = (bx - embed_tiles) * NT + (int)threadIdx.x;
            edge_umap_body(um.g, um.graph_ptr, um.node_graph, um.pair_ptr, um.P, t);
            if (t < um.n_zero) um.zero_words[t] = 0;
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typed_embed_tile_h<H, FUSE0>(w, ta, tb, bx, m, pos, pair2u, P, attr_row, edge_attr, out_stride, f0,
                                 range_status, smem);
}

int launch_typed_embed(const tsd_model_cfg& c, const float* W, const tsd_batch& b, const float* pos, float* edge_attr,
                       int M, size_t out_stride, hipStream_t st, const UmapRole* umap, const EmbedFuse0* fuse0, Prec prec) {
    // (prec.mode == PREC_H2: W and *fuse0 point into the f16-plane arena, the bucket arena is b.bucket_weights16)
    const WeightLayout L = weight_layout(c);
    const int H = c.hidden;
    const int cap = (int)typed_tiles_capacity(b.num_pairs);
    (void)cap;
    TypedList ta{b.enc_tiles.num_tiles, b.enc_tiles.tile_slot, b.enc_tiles.tile_start, b.enc_tiles.tile_count,
                 b.enc_tiles.pair, b.enc_tiles.node_i, b.enc_tiles.node_j};
    TypedList tb{b.diff_tiles.num_tiles, b.diff_tiles.tile_slot, b.diff_tiles.tile_start, b.diff_tiles.tile_count,
                 b.diff_tiles.pair, b.diff_tiles.node_i, b.diff_tiles.node_j};
    const int nslots = b.enc_tiles.num_buckets + b.diff_tiles.num_buckets;
    const bool h2 = prec.mode == PREC_H2;
    if (h2 && b.bucket_weights16 == nullptr) {
        set_error("internal: split-f16 typed embedding without bucket_weights16");
        return TSD_ERR_INVALID;
    }
    TypedEmbedW w{W + L.emlp_w0, W + L.emlp_b0, h2 ? b.bucket_weights16 : b.bucket_weights, L.total,
                  (size_t)nslots * ((size_t)H * H + H)};
    UmapRole um{};
    if (umap && umap->P > 0) {
        um = *umap;
        um.blocks = ((um.P > um.n_zero ? um.P : um.n_zero) + 2 * H - 1) / (2 * H);
    }
    const int grid = ta.n + tb.n + um.blocks;
    if (grid == 0) return TSD_OK;
    const size_t lds = h2 ? (size_t)(T * ldh_of(H) + 3 * T + 3 * H /* biases */) * 4 : (size_t)(T * (H + 4) + 2 * T) * 4;
#define TSD_TEH(HH, FU, FARG)                                                                                  \
    {                                                                                                          \
        static DeviceOnce once;                                                                                \
        int r = allow_lds(typed_embed_h_kernel<HH, FU>, lds, once);                                            \
        if (r) return r;                                                                                       \
        hipLaunchKernelGGL((typed_embed_h_kernel<HH, FU>), dim3(grid * M), dim3(2 * HH), lds, st, w, ta, tb, pos, \
                           b.geo.pair2u, b.num_pairs, b.geo.attr_row, edge_attr, out_stride, um, FARG,         \
                           prec.range_status, ckpt_grid_m(M, grid));                                           \
    }
#define TSD_TE(HH, FU, FARG)                                                                                   \
    {                                                                                                          \
        static DeviceOnce once;                                                                                \
        int r = allow_lds(typed_embed_kernel<HH, FU>, lds, once);                                              \
        if (r) return r;                                                                                       \
        hipLaunchKernelGGL((typed_embed_kernel<HH, FU>), dim3(grid, M), dim3(2 * HH), lds, st, w, ta, tb, pos,  \
                           b.geo.pair2u, b.num_pairs, b.geo.attr_row, edge_attr, out_stride, um, FARG);        \
    }
    if (h2) {
        switch (H) {
            case 64: if (fuse0) TSD_TEH(64, true, *fuse0) else TSD_TEH(64, false, EmbedFuse0{}) break;
            case 128: if (fuse0) TSD_TEH(128, true, *fuse0) else TSD_TEH(128, false, EmbedFuse0{}) break;
            case 256: if (fuse0) TSD_TEH(256, true, *fuse0) else TSD_TEH(256, false, EmbedFuse0{}) break;
            default: set_error("hidden=%d unsupported (64/128/256)", H); return TSD_ERR_INVALID;
        }
    } else {
        switch (H) {
            case 64: if (fuse0) TSD_TE(64, true, *fuse0) else TSD_TE(64, false, EmbedFuse0{}) break;
            case 128: if (fuse0) TSD_TE(128, true, *fuse0) else TSD_TE(128, false, EmbedFuse0{}) break;
            case 256: if (fuse0) TSD_TE(256, true, *fuse0) else TSD_TE(256, false, EmbedFuse0{}) break;
            default: set_error("hidden=%d unsupported (64/128/256)", H); return TSD_ERR_INVALID;
        }
    }
#undef TSD_TEH
#undef TSD_TE
    TSD_LAUNCH_CHECK("typed_embed");
    return TSD_OK;
}

}  // namespace tsd

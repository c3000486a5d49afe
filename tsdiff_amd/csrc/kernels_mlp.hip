// kernels_mlp.hip -- the MFMA-bound kernels of the score network (gfx950, fp32 MFMA 32x32x2).
//
// Every kernel processes one tile of 32 edges (or 32 nodes) per workgroup of H threads
// (H/64 waves, each wave owns 64 output columns), keeps the whole chain of dense layers of
// that tile on chip (activations in LDS, accumulators in AGPR/VGPR) and streams the packed
// weights from L2 with 1-KiB coalesced wave loads.  Reference call sites are cited per kernel.
#include "train_internal.hpp"
#include "split16.hpp"

namespace tsd {

constexpr int T = TSD_EDGE_TILE;   // 32 edges per tile in the per-edge kernels
constexpr int TN = TSD_NODE_TILE;  // 16 nodes per tile in the per-node kernels

struct EdgeEmbedW {
    const float *bond_emb, *w0, *b0, *w1, *b1, *cw0, *cb0, *cw1, *cb1;
};
struct CfconvW {
    const float *nn0_w, *nn0_b, *nn2_w, *nn2_b;
};
struct NodeW {
    const float *lin2_w, *lin2_b, *lin_w, *lin_b, *lin1_next_w;
};
struct PairW {
    const float *w0, *b0, *w1, *b1, *w2, *b2;
    const float *w0e, *b0e;  // edge-attribute half of layer 0 ([H x H] packed) and the bias that goes with it: W0[:, H:]
                             // and b0, or their folded forms when the attribute rows hold s1 (common.hpp)
};

// ---------------------------------------------------------------------------------------------
// A7 + A8: edge_attr = edge_cat([mlp(d) * emb[type_r], mlp(d) * emb[type_p]])
// reference models/encoder/edge.py:58-68, models/epsnet/condensenc.py:156-176,105-115
// mlp(d) is evaluated once and shared by the r/p branches (identical inputs in the reference).
// ---------------------------------------------------------------------------------------------
// Two lists may share one launch (tiles [0, tiles_a) -> list a, the rest -> list b): the encoder list and
// the few output-graph edges that need their own embedding fill the chip together.
// SAVE: the training step's instantiation also writes the intermediate activations (EmbedSave, row = edge-attribute
// row: list b starts at row `save_b_row`).
// FUSE0: the tiles of list a go on to the filter GEMMs of interaction block 0 (EmbedFuse0, common.hpp).
// FOLD: the chain stops at s1 = swish(edge_cat.0(...)), which is written as the tile's attribute rows; consumers hold
// edge_cat.2 folded into their weights (common.hpp, FOLDED WEIGHTS).  f0's nn0 pointers are then the folded ones.
template <int H, bool SAVE, bool FUSE0, bool FOLD>
__global__ __launch_bounds__(2 * H) void edge_embed_kernel(EdgeEmbedW w, tsd_edges ea_, float* __restrict__ out_a,
                                                       int tiles_a, tsd_edges eb_, float* __restrict__ out_b,
                                                       size_t wstride, size_t out_stride, int embed_tiles,
                                                       UmapRole um, EmbedSave sv, int save_b_row, EmbedFuse0 f0) {
    constexpr int LDA = 2 * H + 4;
    if ((int)blockIdx.x >= embed_tiles) {  // extra role: directed-edge -> undirected-pair map (checkpoint 0 only)
        if (blockIdx.y == 0) {
            const int t = ((int)blockIdx.x - embed_tiles) * (2 * H) + (int)threadIdx.x;
            edge_umap_body(um.g, um.graph_ptr, um.node_graph, um.pair_ptr, um.P, t);
            if (t < um.n_zero) um.zero_words[t] = 0;  // readiness flags of the forward's last launch (api.hip)
        }
        return;
    }
    {  // blockIdx.y = checkpoint of the ensemble: its weight arena and its output block
        const size_t wo = (size_t)blockIdx.y * wstride, oo = (size_t)blockIdx.y * out_stride;
        w.bond_emb += wo; w.w0 += wo; w.b0 += wo; w.w1 += wo; w.b1 += wo;
        w.cw0 += wo; w.cb0 += wo; w.cw1 += wo; w.cb1 += wo;
        out_a += oo; out_b += oo;
        if constexpr (FUSE0) {
            f0.nn0_w += wo; f0.nn0_b += wo; f0.nn2_w += wo; f0.nn2_b += wo;
            f0.wf += (size_t)blockIdx.y * f0.wf_stride;
        }
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf = smem;
    float* s_d = smem + T * LDA;
    int* s_tr = reinterpret_cast<int*>(s_d + T);
    int* s_tp = s_tr + T;

    const bool second = (int)blockIdx.x >= tiles_a;
    const tsd_edges& e = second ? eb_ : ea_;
    float* __restrict__ edge_attr = second ? out_b : out_a;
    const int E = *e.count;
    const int e0 = (second ? (int)blockIdx.x - tiles_a : (int)blockIdx.x) * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32;  // 2H threads = H/32 waves x 32 columns (measured faster than H/64 x 64)
    const size_t srow0 = (size_t)(second ? save_b_row : 0) + e0;  // SAVE: first row of this tile in the save arrays

    if (tid < T) {
        const int ee = e0 + tid;
        const bool v = ee < E;
        s_d[tid] = v ? e.dist[ee] : 0.0f;
        s_tr[tid] = v ? (int)e.type_r[ee] : 0;
        s_tp[tid] = v ? (int)e.type_p[ee] : 0;
        if constexpr (SAVE) {
            if (v) {
                sv.d[srow0 + tid] = s_d[tid];
                sv.tr[srow0 + tid] = (uint8_t)s_tr[tid];
                sv.tp[srow0 + tid] = (uint8_t)s_tp[tid];
            }
        }
    }
    __syncthreads();
    {  // Linear(1,H) + swish: thread = (channel, half of the tile's rows)
        const int c = tid % H, r0 = (tid / H) * (T / 2);
        const float w0 = w.w0[c], b0 = w.b0[c];
#pragma unroll 8
        for (int r = r0; r < r0 + T / 2; ++r) {
            const float l = w0 * s_d[r] + b0, sl = swishf(l);
            buf[r * LDA + c] = sl;
            if constexpr (SAVE) {
                if (e0 + r < E) {
                    sv.l0[(srow0 + r) * H + c] = l;
                    sv.s0[(srow0 + r) * H + c] = sl;
                }
            }
        }
    }
    __syncthreads();

    f32x16 acc[1][1];
    zero_acc(acc);
    gemm_tile<1, 1, H>(buf, LDA, w.w1, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 1; ++cb) {
        const int col = col0 + cb * 32 + l31;
        const float b = w.b1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            const float v = acc[0][cb][r] + b;
            const float vr = v * w.bond_emb[s_tr[row] * H + col], vp = v * w.bond_emb[s_tp[row] * H + col];
            buf[row * LDA + col] = vr;
            buf[row * LDA + H + col] = vp;
            if constexpr (SAVE) {
                if (e0 + row < E) {
                    sv.e[(srow0 + row) * H + col] = v;
                    sv.c[(srow0 + row) * 2 * H + col] = vr;
                    sv.c[(srow0 + row) * 2 * H + H + col] = vp;
                }
            }
        }
    }
    __syncthreads();

    zero_acc(acc);
    gemm_tile<1, 1, 2 * H>(buf, LDA, w.cw0, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 1; ++cb) {
        const int col = col0 + cb * 32 + l31;
        const float b = w.cb0[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            const float v = acc[0][cb][r] + b, sv1 = swishf(v);
            buf[row * LDA + col] = sv1;
            if constexpr (SAVE) {
                if (e0 + row < E) {
                    sv.c0[(srow0 + row) * H + col] = v;
                    sv.s1[(srow0 + row) * H + col] = sv1;
                }
            }
        }
    }
    __syncthreads();

    const int nrows = min(T, E - e0);
    const int col = col0 + l31;
    if constexpr (FOLD) {
        // s1 (in LDS) IS the attribute tile: whole 1-KiB rows, float4 per lane
        constexpr int C4 = H / 4;
        for (int idx = tid; idx < nrows * C4; idx += 2 * H) {
            const int r = idx / C4, c4 = idx % C4;
            store_stream16(edge_attr + (size_t)(e0 + r) * H + c4 * 4, *reinterpret_cast<const f32x4*>(buf + r * LDA + c4 * 4));
        }
        if constexpr (!FUSE0) return;
        if (second) return;
        // (rows past the end hold finite values computed from d = 0; their filter rows are never stored)
    } else {
        zero_acc(acc);
        gemm_tile<1, 1, H>(buf, LDA, w.cw1, H, col0, acc);
        {
            const float b = w.cb1[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ee = e0 + acc_row(r, hi);
                if (ee < E) edge_attr[(size_t)ee * H + col] = acc[0][0][r] + b;
            }
            if constexpr (FUSE0) {  // keep the attribute tile for the filter GEMMs below
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] += b;
            }
        }
        if constexpr (FUSE0) {
            if (second) return;
            __syncthreads();  // every wave is done reading buf
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, hi);
                buf[row * LDA + col] = row < nrows ? acc[0][0][r] : 0.0f;
            }
            __syncthreads();
        }
    }
    if constexpr (FUSE0) {
        // filter role of interaction block 0 on this tile (kernels_combo.hip::filter_role, same order of operations)
        zero_acc(acc);
        gemm_tile<1, 1, H>(buf, LDA, f0.nn0_w, H, col0, acc);
        __syncthreads();
        {
            const float b = f0.nn0_b[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) buf[acc_row(r, hi) * LDA + col] = sspf(acc[0][0][r] + b);
        }
        __syncthreads();
        zero_acc(acc);
        gemm_tile<1, 1, H>(buf, LDA, f0.nn2_w, H, col0, acc);
        __syncthreads();
        {
            const float b = f0.nn2_b[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, hi);
                const float cw = row < nrows ? cutoff_weight(s_d[row], f0.conv_cutoff, f0.smooth) : 0.0f;
                buf[row * LDA + col] = (acc[0][0][r] + b) * cw;
            }
        }
        __syncthreads();
        constexpr int C4 = H / 4;
        for (int idx = tid; idx < nrows * C4; idx += 2 * H) {
            const int r = idx / C4, c4 = idx % C4;
            store_stream16(f0.wf + (size_t)(e0 + r) * H + c4 * 4, *reinterpret_cast<const f32x4*>(buf + r * LDA + c4 * 4));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The edge embedding's backward chain for one tile of 32 embedded edges (the adjoint of edge_embed_kernel;
// condensenc.py:156-176, edge.py:58-68 backwards), both lists in one launch like the forward:
//   dc0 = (d_ea . W_cat1) * swish'(c0)                    -> global (weight gradient of edge_cat.0), LDS
//   dc  = dc0 . W_cat0  [2H]                               -> global (bond-embedding gradient)
//   de  = dc_lo * emb[type_r] + dc_hi * emb[type_p]        -> global (weight gradient of mlp.1), LDS
//   dl0 = (de . W_mlp1) * swish'(l0)                       -> global (weight gradient of mlp.0)
// Weights in the dgrad layout.  The weight / embedding-table gradients are separate (row-split) launches.
// ---------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(2 * H) void embed_bwd_kernel(EmbedBwdList la, int tiles_a, EmbedBwdList lb,
                                                         const float* __restrict__ bond_emb,
                                                         const float* __restrict__ W1t, const float* __restrict__ W0t,
                                                         const float* __restrict__ Wmt) {
    constexpr int LDA = H + 4, NT = 2 * H, C4 = H / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf = smem;
    int* s_tr = reinterpret_cast<int*>(smem + T * LDA);
    int* s_tp = s_tr + T;
    const bool second = (int)blockIdx.x >= tiles_a;
    const EmbedBwdList& L = second ? lb : la;
    const int E = *L.e.count;
    const int e0 = (second ? (int)blockIdx.x - tiles_a : (int)blockIdx.x) * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32, col = col0 + l31;
    const int nrows = min(T, E - e0);
    if (tid < T) {
        const bool v = tid < nrows;
        s_tr[tid] = v ? (int)L.e.type_r[e0 + tid] : 0;
        s_tp[tid] = v ? (int)L.e.type_p[e0 + tid] : 0;
    }
    {
        constexpr int NIT = T * C4 / NT;
        static_assert(T * C4 % NT == 0, "tile / block mismatch");
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            v[it] = *reinterpret_cast<const f32x4*>(L.d_ea + (size_t)(e0 + min(r, nrows - 1)) * H + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = r < nrows ? v[it] : z;
        }
    }
    __syncthreads();
    // (pre-activations are requested before the GEMM that precedes their use, rows clamped: no guarded loads)
    float pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = L.c0[(size_t)(e0 + min(acc_row(r, hi), nrows - 1)) * H + col];
    f32x16 acc[1][1], acc2[1][1];
    zero_acc(acc);
    gemm_tile<1, 1, H>(buf, LDA, W1t, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        float v = 0.0f;
        if (row < nrows) {
            v = acc[0][0][r] * act_deriv(0, pre[r]);
            L.dc0[(size_t)(e0 + row) * H + col] = v;
        }
        buf[row * LDA + col] = v;
    }
    __syncthreads();
    zero_acc(acc);
    zero_acc(acc2);
    gemm_tile<1, 1, H>(buf, LDA, W0t, 2 * H, col0, acc);       // d(e * emb[type_r]) columns
    gemm_tile<1, 1, H>(buf, LDA, W0t, 2 * H, col0 + H, acc2);  // d(e * emb[type_p]) columns
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        float v = 0.0f;
        if (row < nrows) {
            const float lo = acc[0][0][r], hh = acc2[0][0][r];
            L.dc[(size_t)(e0 + row) * 2 * H + col] = lo;
            L.dc[(size_t)(e0 + row) * 2 * H + H + col] = hh;
            v = lo * bond_emb[s_tr[row] * H + col] + hh * bond_emb[s_tp[row] * H + col];
            L.de[(size_t)(e0 + row) * H + col] = v;
        }
        buf[row * LDA + col] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = L.l0[(size_t)(e0 + min(acc_row(r, hi), nrows - 1)) * H + col];
    zero_acc(acc);
    gemm_tile<1, 1, H>(buf, LDA, Wmt, H, col0, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        if (row < nrows) L.dl0[(size_t)(e0 + row) * H + col] = acc[0][0][r] * act_deriv(0, pre[r]);
    }
}

// The same chain on the f16 MFMA pipes (split16.hpp, GRADIENT operands; W1t / W0t / Wmt: f16-plane images of the dgrad
// matrices).  The d_ea rows are scaled row by row (a row is staged by ONE wave), dc0 and de by their tile's max.
// amax[0..2]: running max of |d_ea|, |dc0|, |de| for the weight-gradient launches.
template <int H>
__global__ __launch_bounds__(2 * H) void embed_bwd_h_kernel(EmbedBwdList la, int tiles_a, EmbedBwdList lb,
                                                           const float* __restrict__ bond_emb,
                                                           const float* __restrict__ W1t, const float* __restrict__ W0t,
                                                           const float* __restrict__ Wmt, float* __restrict__ amax) {
    constexpr int LDH = ldh_of(H), NT = 2 * H, C4 = H / 4, NW = NT / 64;
    static_assert(C4 == 64, "a row is staged by one wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Planes pl = planes_at(smem, T, LDH);
    float* s_inv = smem + T * LDH;  // [T] 2^e of the d_ea rows
    float* s_rmax = s_inv + T;      // [T] max |d_ea| of the rows
    float* s_wmax = s_rmax + T;     // [NW]
    int* s_tr = reinterpret_cast<int*>(s_wmax + NW);
    int* s_tp = s_tr + T;
    const bool second = (int)blockIdx.x >= tiles_a;
    const EmbedBwdList& L = second ? lb : la;
    const int E = *L.e.count;
    const int e0 = (second ? (int)blockIdx.x - tiles_a : (int)blockIdx.x) * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = wave * 32, col = col0 + l31;
    const int nrows = min(T, E - e0);
    if (tid < T) {
        const bool v = tid < nrows;
        s_tr[tid] = v ? (int)L.e.type_r[e0 + tid] : 0;
        s_tp[tid] = v ? (int)L.e.type_p[e0 + tid] : 0;
    }
    float dummy = 0.0f;
    {
        constexpr int NIT = T * C4 / NT;
        static_assert(T * C4 % NT == 0, "tile / block mismatch");
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            v[it] = *reinterpret_cast<const f32x4*>(L.d_ea + (size_t)(e0 + min(r, nrows - 1)) * H + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 a = r < nrows ? v[it] : z;
            const float m = max64(fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));  // the row's max
            float inv;
            const float sc = pow2_scale(m, inv);
            if (c4 == 0) {
                s_inv[r] = inv;
                s_rmax[r] = m;
            }
            planes_store4(pl, r * LDH + c4 * 4, a * sc, dummy);
        }
    }
    __syncthreads();
    f32x16 accm[1][1], accx[1][1];
    hzero(accm, accx);
    hgemm_tile<1, 1, H>(pl, LDH, W1t, H, col0, accm, accx);
    // (SGPR tile bases + 32-bit lane offsets, computed behind the GEMM's asm statements: see filter_bwd_role_h)
    unsigned off[16];
    int hi_p = hi;
    asm volatile("" : "+v"(hi_p));
#pragma unroll
    for (int r = 0; r < 16; ++r) off[r] = (unsigned)(min(acc_row(r, hi_p), nrows - 1) * H + col) * 4u;
    auto at = [](const float* base, unsigned o) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + o); };
    auto atw = [](float* base, unsigned o) { return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + o); };
    const float* c0t = L.c0 + (size_t)e0 * H;
    float* dc0t = L.dc0 + (size_t)e0 * H;
    float pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = *at(c0t, off[r]);
    float v0[16], m0 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        v0[r] = 0.0f;
        if (row < nrows) {
            v0[r] = hval(accm[0][0], accx[0][0], r) * s_inv[row] * act_deriv(0, pre[r]);
            *atw(dc0t, off[r]) = v0[r];
        }
        m0 = fmaxf(m0, fabsf(v0[r]));
    }
    m0 = max64(m0);
    if (lane == 0) s_wmax[wave] = m0;
    __syncthreads();  // (every wave is done reading the planes, the wave maxima are in place)
    float tmax1 = s_wmax[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) tmax1 = fmaxf(tmax1, s_wmax[k]);
    float inv2;
    const float sc2 = pow2_scale(tmax1, inv2);
#pragma unroll
    for (int r = 0; r < 16; ++r) planes_store1(pl, acc_row(r, hi) * LDH + col, v0[r] * sc2, dummy);
    __syncthreads();
    hzero(accm, accx);
    hgemm_tile<1, 1, H>(pl, LDH, W0t, 2 * H, col0, accm, accx);  // d(e * emb[type_r]) columns
    float* dct = L.dc + (size_t)e0 * 2 * H;
    hi_p = hi;
    asm volatile("" : "+v"(hi_p));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi_p);
        v0[r] = hval(accm[0][0], accx[0][0], r) * inv2;
        if (row < nrows) *atw(dct, (unsigned)(row * 2 * H + col) * 4u) = v0[r];
    }
    hzero(accm, accx);
    hgemm_tile<1, 1, H>(pl, LDH, W0t, 2 * H, col0 + H, accm, accx);  // d(e * emb[type_p]) columns
    float* det = L.de + (size_t)e0 * H;
    hi_p = hi;
    asm volatile("" : "+v"(hi_p));
    m0 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi_p);
        const float hh = hval(accm[0][0], accx[0][0], r) * inv2;
        float v = 0.0f;
        if (row < nrows) {
            *atw(dct, (unsigned)(row * 2 * H + H + col) * 4u) = hh;
            v = v0[r] * bond_emb[s_tr[row] * H + col] + hh * bond_emb[s_tp[row] * H + col];
            *atw(det, (unsigned)(row * H + col) * 4u) = v;
        }
        v0[r] = v;
        m0 = fmaxf(m0, fabsf(v));
    }
    m0 = max64(m0);
    __syncthreads();  // (the planes and the wave maxima of the previous stage are consumed)
    if (lane == 0) s_wmax[wave] = m0;
    __syncthreads();
    float tmax2 = s_wmax[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) tmax2 = fmaxf(tmax2, s_wmax[k]);
    float inv3;
    const float sc3 = pow2_scale(tmax2, inv3);
#pragma unroll
    for (int r = 0; r < 16; ++r) planes_store1(pl, acc_row(r, hi) * LDH + col, v0[r] * sc3, dummy);
    __syncthreads();
    hzero(accm, accx);
    hgemm_tile<1, 1, H>(pl, LDH, Wmt, H, col0, accm, accx);
    const float* l0t = L.l0 + (size_t)e0 * H;
    float* dl0t = L.dl0 + (size_t)e0 * H;
    hi_p = hi;
    asm volatile("" : "+v"(hi_p));
#pragma unroll
    for (int r = 0; r < 16; ++r) off[r] = (unsigned)(min(acc_row(r, hi_p), nrows - 1) * H + col) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = *at(l0t, off[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        if (row < nrows) *atw(dl0t, off[r]) = hval(accm[0][0], accx[0][0], r) * inv3 * act_deriv(0, pre[r]);
    }
    if (amax != nullptr) {  // one (conditional) atomic per tile and word
        if (wave == 0) {
            const float m = max32(lane < nrows && lane < T ? s_rmax[lane] : 0.0f);
            if (lane == 0) atomic_amax(amax, m);
        }
        if (tid == 64) atomic_amax(amax + 1, tmax1);
        if (tid == 128) atomic_amax(amax + 2, tmax2);
    }
}

int launch_embed_bwd(int H, int rows_a, const EmbedBwdList& la, int rows_b, const EmbedBwdList& lb, const float* bond_emb,
                     const float* W1t, const float* W0t, const float* Wmt, hipStream_t st, float* amax_h2) {
    const int tiles_a = (rows_a + T - 1) / T, tiles_b = (rows_b + T - 1) / T;
    if (tiles_a + tiles_b == 0) return TSD_OK;
    if (amax_h2 != nullptr) {  // split-f16 form: W1t / W0t / Wmt are f16-plane images
        if (H != 256) {
            set_error("embed_bwd: hidden=%d has no split-f16 instance", H);
            return TSD_ERR_INVALID;
        }
        const size_t lds_h = (size_t)(T * ldh_of(256) + 2 * T + 8) * 4 + 2 * T * sizeof(int);
        static DeviceOnce once_h;
        int r = allow_lds(embed_bwd_h_kernel<256>, lds_h, once_h);
        if (r) return r;
        hipLaunchKernelGGL(embed_bwd_h_kernel<256>, dim3(tiles_a + tiles_b), dim3(512), lds_h, st, la, tiles_a, lb, bond_emb,
                           W1t, W0t, Wmt, amax_h2);
        TSD_LAUNCH_CHECK("embed_bwd_h");
        return TSD_OK;
    }
    const size_t lds = (size_t)(T * (H + 4)) * 4 + 2 * T * sizeof(int);
#define TSD_EB(HH)                                                                                                  \
    {                                                                                                               \
        static DeviceOnce once;                                                                                     \
        int r = allow_lds(embed_bwd_kernel<HH>, lds, once);                                                         \
        if (r) return r;                                                                                            \
        hipLaunchKernelGGL(embed_bwd_kernel<HH>, dim3(tiles_a + tiles_b), dim3(2 * HH), lds, st, la, tiles_a, lb,    \
                           bond_emb, W1t, W0t, Wmt);                                                                \
    }
    if (H == 128) TSD_EB(128) else if (H == 256) TSD_EB(256) else {
        set_error("embed_bwd: hidden=%d has no MFMA instance", H);
        return TSD_ERR_INVALID;
    }
#undef TSD_EB
    TSD_LAUNCH_CHECK("embed_bwd");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// The pair MLP's backward chain for one tile of 32 undirected out edges (the adjoint of pair_output_kernel;
// common.py:226-229 backwards):
//   dg1 = ds * w2 * swish'(g1)          [H/2]  -> global (weight gradient of layers.1), LDS
//   dg0 = (dg1 . W1) * swish'(g0)       [H]    -> global (weight gradient of layers.0), LDS
//   dhp = dg0 . W0                      [2H]:  left half dp -> global [rows,H] (adjoint of h_i * h_j),
//                                               right half -> row attr_row[e] of the edge-attribute gradient
// ---------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(2 * H) void pair_bwd_kernel(tsd_edges e, const int32_t* __restrict__ attr_row,
                                                        const float* __restrict__ ds, const float* __restrict__ w2,
                                                        const float* __restrict__ g1, const float* __restrict__ g0,
                                                        const float* __restrict__ W1t, const float* __restrict__ W0t,
                                                        float* __restrict__ dg1, float* __restrict__ dg0,
                                                        float* __restrict__ dp, float* __restrict__ d_ea,
                                                        int attr_from, int attr_shift) {
    constexpr int LDA = H + 4, NT = 2 * H, HH = H / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf = smem;
    float* s_ds = smem + T * LDA;
    int* s_row = reinterpret_cast<int*>(s_ds + T);
    const int E = *e.count;
    const int e0 = blockIdx.x * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32, col = col0 + l31;
    const int nrows = min(T, E - e0);
    if (tid < T) {
        const bool v = tid < nrows;
        s_ds[tid] = v ? ds[e0 + tid] : 0.0f;
        int row = v ? attr_row[e0 + tid] : 0;
        if (row >= attr_from) row -= attr_shift;  // (as PairSave::attr_from / attr_shift of the forward)
        s_row[tid] = row;
    }
    __syncthreads();
    {   // dg1 tile: every load of a thread in flight together (rows clamped)
        constexpr int NIT = T * HH / NT;
        static_assert(T * HH % NT == 0, "tile / block mismatch");
        float gv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = min(idx / HH, nrows - 1), c = idx % HH;
            gv[it] = g1[(size_t)(e0 + r) * HH + c];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / HH, c = idx % HH;
            float v = 0.0f;
            if (r < nrows) {
                v = s_ds[r] * w2[c] * act_deriv(0, gv[it]);
                dg1[(size_t)(e0 + r) * HH + c] = v;
            }
            buf[r * LDA + c] = v;
        }
    }
    __syncthreads();
    float pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = g0[(size_t)(e0 + min(acc_row(r, hi), nrows - 1)) * H + col];
    f32x16 acc[1][1], acc2[1][1];
    zero_acc(acc);
    gemm_tile<1, 1, HH>(buf, LDA, W1t, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        float v = 0.0f;
        if (row < nrows) {
            v = acc[0][0][r] * act_deriv(0, pre[r]);
            dg0[(size_t)(e0 + row) * H + col] = v;
        }
        buf[row * LDA + col] = v;
    }
    __syncthreads();
    zero_acc(acc);
    zero_acc(acc2);
    gemm_tile<1, 1, H>(buf, LDA, W0t, 2 * H, col0, acc);
    gemm_tile<1, 1, H>(buf, LDA, W0t, 2 * H, col0 + H, acc2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        if (row < nrows) {
            dp[(size_t)(e0 + row) * H + col] = acc[0][0][r];
            d_ea[(size_t)s_row[row] * H + col] = acc2[0][0][r];  // every out edge owns its attribute row
        }
    }
}

#ifndef TSD_BWD_PIN
#define TSD_BWD_PIN false
#endif
#ifndef TSD_BWD_RING
#define TSD_BWD_RING 3  // weight k-steps in flight per wave in the backward tile chains (they have the registers)
#endif
// The same chain on the f16 MFMA pipes (split16.hpp, GRADIENT operands; W1t / W0t: f16-plane images of the dgrad
// matrices).  dg1 rows are scaled by 2^-e of the bound |ds| max|w2| 1.1 >= max |row| (swish' <= 1.0999: no reduction
// needed), dg0 by the tile's max (through LDS beside the barrier the planes need).  amax[0] / amax[1]: running max of
// |dg1| / |dg0| for the weight-gradient launches.
template <int H>
__global__ __launch_bounds__(2 * H) void pair_bwd_h_kernel(tsd_edges e, const int32_t* __restrict__ attr_row,
                                                          const float* __restrict__ ds, const float* __restrict__ w2,
                                                          const float* __restrict__ g1, const float* __restrict__ g0,
                                                          const float* __restrict__ W1t, const float* __restrict__ W0t,
                                                          float* __restrict__ dg1, float* __restrict__ dg0,
                                                          float* __restrict__ dp, float* __restrict__ d_ea,
                                                          int attr_from, int attr_shift, float* __restrict__ amax) {
    constexpr int LDH = ldh_of(H), NT = 2 * H, HH = H / 2, NW = NT / 64;
    static_assert(HH == 128, "the w2 max below reads two values per lane");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Planes pl = planes_at(smem, T, LDH);
    float* s_ds = smem + T * LDH;
    float* s_sc = s_ds + T;    // [T] 2^-e of the dg1 rows
    float* s_inv = s_sc + T;   // [T] 2^e
    float* s_wmax = s_inv + T; // [NW]
    int* s_row = reinterpret_cast<int*>(s_wmax + NW);
    const int E = *e.count;
    const int e0 = blockIdx.x * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = wave * 32, col = col0 + l31;
    const int nrows = min(T, E - e0);
    const float w2m = max64(fmaxf(fabsf(w2[lane]), fabsf(w2[lane + 64])));
    if (tid < T) {
        const bool v = tid < nrows;
        const float d = v ? ds[e0 + tid] : 0.0f;
        s_ds[tid] = d;
        float inv;
        s_sc[tid] = pow2_scale(fabsf(d) * w2m * 1.1f, inv);
        s_inv[tid] = inv;
        int row = v ? attr_row[e0 + tid] : 0;
        if (row >= attr_from) row -= attr_shift;
        s_row[tid] = row;
    }
    __syncthreads();
    float dummy = 0.0f, m1 = 0.0f;
    {
        constexpr int NIT = T * HH / NT;
        static_assert(T * HH % NT == 0, "tile / block mismatch");
        float gv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = min(idx / HH, nrows - 1), c = idx % HH;
            gv[it] = g1[(size_t)(e0 + r) * HH + c];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / HH, c = idx % HH;
            float v = 0.0f;
            if (r < nrows) {
                v = s_ds[r] * w2[c] * act_deriv(0, gv[it]);
                dg1[(size_t)(e0 + r) * HH + c] = v;
            }
            amax_upd(m1, v);
            planes_store1(pl, r * LDH + c, v * s_sc[r], dummy);
        }
    }
    __syncthreads();
    f32x16 accm[1][1], accx[1][1];
    hzero(accm, accx);
    hgemm_tile<1, 1, HH, TSD_BWD_PIN, TSD_BWD_RING>(pl, LDH, W1t, H, col0, accm, accx);
    // (tile base pointers are wave-uniform: SGPR base + one 32-bit lane offset per row; computed behind the GEMM's asm
    // statements -- held across the MFMA stream the values cost the second workgroup of the CU, see filter_bwd_role_h)
    const float* g0t = g0 + (size_t)e0 * H;
    float* dg0t = dg0 + (size_t)e0 * H;
    float* dpt = dp + (size_t)e0 * H;
    unsigned off[16];
    int hi_p = hi;
    asm volatile("" : "+v"(hi_p));
#pragma unroll
    for (int r = 0; r < 16; ++r) off[r] = (unsigned)(min(acc_row(r, hi_p), nrows - 1) * H + col) * 4u;
    float pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(g0t) + off[r]);
    float v0[16], m0 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        v0[r] = 0.0f;
        if (row < nrows) {
            v0[r] = hval(accm[0][0], accx[0][0], r) * s_inv[row] * act_deriv(0, pre[r]);
            *reinterpret_cast<float*>(reinterpret_cast<char*>(dg0t) + off[r]) = v0[r];
        }
        m0 = fmaxf(m0, fabsf(v0[r]));
    }
    m0 = max64(m0);
    if (lane == 0) s_wmax[wave] = m0;
    __syncthreads();  // (every wave is done reading the planes, the wave maxima are in place)
    float tmax = s_wmax[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) tmax = fmaxf(tmax, s_wmax[k]);
    float inv2;
    const float sc2 = pow2_scale(tmax, inv2);
#pragma unroll
    for (int r = 0; r < 16; ++r) planes_store1(pl, acc_row(r, hi) * LDH + col, v0[r] * sc2, dummy);
    __syncthreads();
    hzero(accm, accx);
    hgemm_tile<1, 1, H, TSD_BWD_PIN, TSD_BWD_RING>(pl, LDH, W0t, 2 * H, col0, accm, accx);
    hi_p = hi;
    asm volatile("" : "+v"(hi_p));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi_p);
        if (row < nrows) *reinterpret_cast<float*>(reinterpret_cast<char*>(dpt) + (unsigned)(row * H + col) * 4u) =
            hval(accm[0][0], accx[0][0], r) * inv2;
    }
    hzero(accm, accx);
    hgemm_tile<1, 1, H, TSD_BWD_PIN, TSD_BWD_RING>(pl, LDH, W0t, 2 * H, col0 + H, accm, accx);
    hi_p = hi;
    asm volatile("" : "+v"(hi_p));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi_p);
        if (row < nrows) d_ea[(size_t)s_row[row] * H + col] = hval(accm[0][0], accx[0][0], r) * inv2;  // every out edge owns its attribute row
    }
    if (amax != nullptr) {  // one (conditional) atomic per tile and word
        m1 = max64(m1);
        __syncthreads();
        if (lane == 0) s_wmax[wave] = m1;
        __syncthreads();
        if (tid == 0) {
            float t = s_wmax[0];
#pragma unroll
            for (int k = 1; k < NW; ++k) t = fmaxf(t, s_wmax[k]);
            atomic_amax(amax, t);
        }
        if (tid == 64) atomic_amax(amax + 1, tmax);
    }
}

int launch_pair_bwd(int H, int rows, tsd_edges e, const int32_t* attr_row, const float* ds, const float* w2,
                    const float* g1, const float* g0, const float* W1t, const float* W0t, float* dg1, float* dg0,
                    float* dp, float* d_ea, int attr_from, int attr_shift, hipStream_t st, float* amax_h2) {
    if (rows == 0) return TSD_OK;
    const size_t lds = (size_t)(T * (H + 4) + T) * 4 + T * sizeof(int);
    if (H != 256) {
        set_error("pair_bwd: hidden=%d has no MFMA instance", H);
        return TSD_ERR_INVALID;
    }
    if (amax_h2 != nullptr) {  // split-f16 form: W1t / W0t are f16-plane images
        const size_t lds_h = (size_t)(T * ldh_of(H) + 3 * T + 8) * 4 + T * sizeof(int);
        static DeviceOnce once_h;
        int r = allow_lds(pair_bwd_h_kernel<256>, lds_h, once_h);
        if (r) return r;
        hipLaunchKernelGGL(pair_bwd_h_kernel<256>, dim3((rows + T - 1) / T), dim3(512), lds_h, st, e, attr_row, ds, w2, g1, g0,
                           W1t, W0t, dg1, dg0, dp, d_ea, attr_from, attr_shift, amax_h2);
        TSD_LAUNCH_CHECK("pair_bwd_h");
        return TSD_OK;
    }
    static DeviceOnce once;
    int r = allow_lds(pair_bwd_kernel<256>, lds, once);
    if (r) return r;
    hipLaunchKernelGGL(pair_bwd_kernel<256>, dim3((rows + T - 1) / T), dim3(512), lds, st, e, attr_row, ds, w2, g1, g0,
                       W1t, W0t, dg1, dg0, dp, d_ea, attr_from, attr_shift);
    TSD_LAUNCH_CHECK("pair_bwd");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// A9 / T5 fused: W = nn2(ssp(nn0(edge_attr))) * C ; msg = x1[dst] * W ; agg[src] += msg
// reference models/encoder/schnet.py:88-107 (CFConv.forward/message, aggr="add").
// The reference aggregates messages x1[edge_index[0]] * W at edge_index[1]; the extended edge set
// and W are symmetric under (i,j) <-> (j,i), so the sum over incoming edges of node i equals the
// sum over row i of the sorted list, x1 gathered at the column -- a contiguous segment, reduced
// here in edge order without atomics (rows cut by the tile boundary finish in node_update).
// ---------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(H) void cfconv_layer_kernel(CfconvW w, float conv_cutoff, int smooth, tsd_edges e,
                                                         const float* __restrict__ edge_attr,
                                                         const float* __restrict__ x1,
                                                         float* __restrict__ agg, float* __restrict__ part) {
    constexpr int LDA = H + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf = smem;
    float* s_c = smem + T * LDA;
    int* s_src = reinterpret_cast<int*>(s_c + T);
    int* s_dst = s_src + T;
    int* s_rp0 = s_dst + T;
    int* s_rp1 = s_rp0 + T;

    const int E = *e.count;
    const int e0 = blockIdx.x * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 64;
    const int nrows = min(T, E - e0);

    if (tid < T) {
        const bool v = tid < nrows;
        const int ee = e0 + tid;
        const int s = v ? e.src[ee] : 0;
        s_src[tid] = s;
        s_dst[tid] = v ? e.dst[ee] : 0;
        s_c[tid] = v ? cutoff_weight(e.dist[ee], conv_cutoff, smooth) : 0.0f;  // schnet.py:92-98
        s_rp0[tid] = e.row_ptr[s];
        s_rp1[tid] = e.row_ptr[s + 1];
    }
    {  // edge_attr tile -> LDS (float4, coalesced)
        constexpr int C4 = H / 4;
        for (int idx = tid; idx < T * C4; idx += H) {
            const int r = idx / C4, c4 = idx % C4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < nrows) v = *reinterpret_cast<const f32x4*>(edge_attr + (size_t)(e0 + r) * H + c4 * 4);
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = v;
        }
    }
    __syncthreads();

    f32x16 acc[1][2];
    zero_acc(acc);
    gemm_tile<1, 2, H>(buf, LDA, w.nn0_w, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = col0 + cb * 32 + l31;
        const float b = w.nn0_b[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[acc_row(r, hi) * LDA + col] = sspf(acc[0][cb][r] + b);
    }
    __syncthreads();

    // gather x1[dst] for this lane's 16 rows x 2 column blocks while the second GEMM runs
    float xg[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            xg[cb][r] = x1[(size_t)s_dst[acc_row(r, hi)] * H + col0 + cb * 32 + l31];

    zero_acc(acc);
    gemm_tile<1, 2, H>(buf, LDA, w.nn2_w, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = col0 + cb * 32 + l31;
        const float b = w.nn2_b[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            const float Wv = (acc[0][cb][r] + b) * s_c[row];  // W = nn(edge_attr) * C
            buf[row * LDA + col] = xg[cb][r] * Wv;             // message x_j * W
        }
    }
    __syncthreads();

    {  // segmented sum over the tile's rows, one channel per thread, edge order
        const int c = tid;
        float sum = 0.0f;
        int cur = s_src[0];
        int seg_first = 0;
        for (int r = 0; r <= nrows; ++r) {
            const int s = (r < nrows) ? s_src[r] : -1;
            if (s != cur) {
                const bool complete = (s_rp0[seg_first] >= e0) && (s_rp1[seg_first] <= e0 + T);
                if (complete)
                    agg[(size_t)cur * H + c] = sum;
                else
                    part[((size_t)blockIdx.x * 2 + (seg_first == 0 ? 0 : 1)) * H + c] = sum;
                cur = s;
                sum = 0.0f;
                seg_first = r;
            }
            if (r < nrows) sum += buf[r * LDA + c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CFConv filters of every layer, one launch: Wf[l][e] = nn2_l(ssp(nn0_l(edge_attr[e]))) * C(e)
// reference models/encoder/schnet.py:94-99.  grid = (edge tiles, layers).  The filter is independent
// of the node states, so all layers' GEMMs are batched; it is evaluated once per UNDIRECTED pair
// (edge_attr and C are symmetric) and consumed by tsd_cfconv_aggregate through `umap`.
// ---------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(H) void filter_gen_kernel(const float* __restrict__ Wl0, size_t layer_stride,
                                                       size_t o_nn0_w, size_t o_nn0_b, size_t o_nn2_w,
                                                       size_t o_nn2_b, float conv_cutoff, int smooth, tsd_edges e,
                                                       const float* __restrict__ edge_attr,
                                                       float* __restrict__ Wf, size_t wf_layer_stride,
                                                       int layer_base) {
    constexpr int LDA = H + 4;
    constexpr int C4 = H / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf = smem;
    float* s_c = smem + T * LDA;

    const int E = *e.count;
    const int e0 = blockIdx.x * T;
    if (e0 >= E) return;
    const int layer = blockIdx.y + layer_base;
    const float* Wb = Wl0 + (size_t)layer * layer_stride;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 64;
    const int nrows = min(T, E - e0);

    if (tid < T) s_c[tid] = tid < nrows ? cutoff_weight(e.dist[e0 + tid], conv_cutoff, smooth) : 0.0f;  // schnet.py:92-98
    for (int idx = tid; idx < T * C4; idx += H) {
        const int r = idx / C4, c4 = idx % C4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < nrows) v = *reinterpret_cast<const f32x4*>(edge_attr + (size_t)(e0 + r) * H + c4 * 4);
        *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = v;
    }
    __syncthreads();

    f32x16 acc[1][2];
    zero_acc(acc);
    gemm_tile<1, 2, H>(buf, LDA, Wb + o_nn0_w, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = col0 + cb * 32 + l31;
        const float b = Wb[o_nn0_b + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[acc_row(r, hi) * LDA + col] = sspf(acc[0][cb][r] + b);
    }
    __syncthreads();

    zero_acc(acc);
    gemm_tile<1, 2, H>(buf, LDA, Wb + o_nn2_w, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = col0 + cb * 32 + l31;
        const float b = Wb[o_nn2_b + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            buf[row * LDA + col] = (acc[0][cb][r] + b) * s_c[row];  // W = nn(edge_attr) * C
        }
    }
    __syncthreads();
    float* out = Wf + (size_t)layer * wf_layer_stride;  // whole 1-KiB rows, float4 per lane
    for (int idx = tid; idx < nrows * C4; idx += H) {
        const int r = idx / C4, c4 = idx % C4;
        *reinterpret_cast<f32x4*>(out + (size_t)(e0 + r) * H + c4 * 4) =
            *reinterpret_cast<const f32x4*>(buf + r * LDA + c4 * 4);
    }
}

// ---------------------------------------------------------------------------------------------
// node side of an interaction block + the next block's lin1:
//   a = assemble(agg, part); h += lin(ssp(lin2(a))); x1 = lin1_next(h)
// reference models/encoder/schnet.py:103 (lin2), :123-127 (act, lin), :223-224 (residual),
// :101 (next layer's lin1).   MODE 0: full update; MODE 1: only x1 = lin1(h) (first layer).
// N is small (1600 atoms at batch 100): 16-row tiles on the 16x16x4 MFMA and 2H threads (H/32 waves,
// two per SIMD at H=256, 32 columns each) spread the three dependent GEMMs over 2x the CUs with
// 2x the waves per CU of the 32-row form (75 -> see profiles/ for the measured time per layer).
// ---------------------------------------------------------------------------------------------
template <int H, int MODE>
__global__ __launch_bounds__(2 * H) void node_update_kernel(NodeW w, int N, const int32_t* __restrict__ row_ptr,
                                                            const float* __restrict__ agg,
                                                            const float* __restrict__ part,
                                                            float* __restrict__ h, float* __restrict__ x1) {
    constexpr int LDA = H + 4;
    constexpr int NT = 2 * H;
    constexpr int C4 = H / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf = smem;
    int* s_rp0 = reinterpret_cast<int*>(smem + TN * LDA);
    int* s_rp1 = s_rp0 + TN;

    const int n0 = blockIdx.x * TN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int col0 = (tid >> 6) * 32;
    const int nrows = min(TN, N - n0);
    f32x4 acc[2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    if (MODE == 0) {
        if (tid < TN) {
            const bool v = tid < nrows;
            // without a row_ptr every row counts as non-empty and complete (tsd_cfconv_aggregate output)
            s_rp0[tid] = (v && row_ptr) ? row_ptr[n0 + tid] : 0;
            s_rp1[tid] = v ? (row_ptr ? row_ptr[n0 + tid + 1] : 1) : 0;
        }
        __syncthreads();
        for (int idx = tid; idx < TN * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            const int rp0 = s_rp0[r], rp1 = s_rp1[r];
            f32x4 v = zero4;
            if (rp1 > rp0) {
                const int t0 = rp0 / T, t1 = (rp1 - 1) / T;
                if (part == nullptr || t0 == t1) {
                    v = *reinterpret_cast<const f32x4*>(agg + (size_t)(n0 + r) * H + c4 * 4);
                } else {  // row cut by edge-tile boundaries: add the tiles' partial sums in order
                    for (int t = t0; t <= t1; ++t) {
                        const int slot = (t == t0 && rp0 != t0 * T) ? 1 : 0;
                        v += *reinterpret_cast<const f32x4*>(part + ((size_t)t * 2 + slot) * H + c4 * 4);
                    }
                }
            }
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = v;
        }
        __syncthreads();

        acc[0] = zero4; acc[1] = zero4;
        gemm_tile16<2, H>(buf, LDA, w.lin2_w, H, col0, acc);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = col0 + cb * 16 + l15;
            const float b = w.lin2_b[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[(q * 4 + r) * LDA + col] = sspf(acc[cb][r] + b);
        }
        __syncthreads();

        acc[0] = zero4; acc[1] = zero4;
        gemm_tile16<2, H>(buf, LDA, w.lin_w, H, col0, acc);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = col0 + cb * 16 + l15;
            const float b = w.lin_b[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = q * 4 + r;
                float hn = 0.0f;
                if (row < nrows) {
                    const size_t o = (size_t)(n0 + row) * H + col;
                    hn = h[o] + (acc[cb][r] + b);
                    h[o] = hn;
                }
                buf[row * LDA + col] = hn;
            }
        }
        if (w.lin1_next_w == nullptr) return;
        __syncthreads();
    } else {
        for (int idx = tid; idx < TN * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            f32x4 v = zero4;
            if (r < nrows) v = *reinterpret_cast<const f32x4*>(h + (size_t)(n0 + r) * H + c4 * 4);
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = v;
        }
        __syncthreads();
    }

    acc[0] = zero4; acc[1] = zero4;
    gemm_tile16<2, H>(buf, LDA, w.lin1_next_w, H, col0, acc);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = col0 + cb * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            if (row < nrows) x1[(size_t)(n0 + row) * H + col] = acc[cb][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// A10: edge_inv = grad_dist_mlp([h_src * h_dst, edge_attr_out])     (2H -> H -> H/2 -> 1, swish)
// reference models/common.py:226-229, models/epsnet/condensenc.py:72-76,236-237
// ---------------------------------------------------------------------------------------------
// SAVE (training step, `pre` == NULL): also writes the staged input rows and both layers' pre-/post-activations.
template <int H, bool SAVE>
__global__ __launch_bounds__(2 * H) void pair_output_kernel(PairW w, tsd_edges e, const float* __restrict__ h,
                                                        const float* __restrict__ edge_attr,
                                                        const int32_t* __restrict__ attr_row,
                                                        float* __restrict__ edge_inv, size_t wstride,
                                                        size_t h_stride, size_t ea_stride, size_t inv_stride,
                                                        const float* __restrict__ pre, size_t pre_stride,
                                                        PairSave sv) {
    constexpr int LDA = 2 * H + 4;
    constexpr int NW = H / 64;   // waves of the second GEMM (H/2 columns, 32 per wave); the block has 2 NW waves
    {  // blockIdx.y = checkpoint of the ensemble
        const size_t m = blockIdx.y, wo = m * wstride;
        w.w0 += wo; w.b0 += wo; w.w1 += wo; w.b1 += wo; w.w2 += wo; w.b2 += wo; w.w0e += wo; w.b0e += wo;
        h += m * h_stride; edge_attr += m * ea_stride; edge_inv += m * inv_stride;
        if (pre) pre += m * pre_stride;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf = smem;
    float* s_red = smem + T * LDA;  // [NW][T]
    int* s_src = reinterpret_cast<int*>(s_red + NW * T);
    int* s_dst = s_src + T;
    int* s_row = s_dst + T;

    const int E = *e.count;
    const int e0 = blockIdx.x * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int nrows = min(T, E - e0);

    if (tid < T) {
        const bool v = tid < nrows;
        s_src[tid] = v ? e.src[e0 + tid] : 0;
        s_dst[tid] = v ? e.dst[e0 + tid] : 0;
        int row = v ? (attr_row ? attr_row[e0 + tid] : e0 + tid) : 0;
        if constexpr (SAVE) {
            if (row >= sv.attr_from) row -= sv.attr_shift;
        }
        s_row[tid] = row;
    }
    __syncthreads();
    // the precomputed first-layer half of this lane's outputs is requested first: it arrives under the staging below
    float pre_v[16];
    if (pre) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            pre_v[r] = pre[(size_t)(e0 + min(acc_row(r, hi), nrows - 1)) * H + wave * 32 + l31];
    }
    {  // h_src * h_dst || edge_attr row -> LDS, float4 per lane, every load of a thread in flight together (rows past
       // the end clamped; a guarded load per iteration makes the compiler wait for each one)
        constexpr int C4 = H / 4;
        constexpr int NIT = T * C4 / (2 * H);
        static_assert(T * C4 % (2 * H) == 0, "tile / block mismatch");
        f32x4 hs[NIT], hd[NIT], bb[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 2 * H, r = min(idx / C4, nrows - 1), c4 = idx % C4;
            hs[it] = *reinterpret_cast<const f32x4*>(h + (size_t)s_src[r] * H + c4 * 4);
            hd[it] = *reinterpret_cast<const f32x4*>(h + (size_t)s_dst[r] * H + c4 * 4);
            if (!pre) bb[it] = *reinterpret_cast<const f32x4*>(edge_attr + (size_t)s_row[r] * H + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 2 * H, r = idx / C4, c4 = idx % C4;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (r < nrows) {
                a = hs[it] * hd[it];
                if (!pre) b = bb[it];
            }
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = a;
            if (!pre) *reinterpret_cast<f32x4*>(buf + r * LDA + H + c4 * 4) = b;
            if constexpr (SAVE) {
                if (r < nrows) {
                    *reinterpret_cast<f32x4*>(sv.hp + (size_t)(e0 + r) * 2 * H + c4 * 4) = a;
                    *reinterpret_cast<f32x4*>(sv.hp + (size_t)(e0 + r) * 2 * H + H + c4 * 4) = b;
                }
            }
        }
    }
    __syncthreads();

    {   // 2H -> H on H/32 waves x 32 columns, summed as (edge_attr half + bias) then + (h_i*h_j half): the first
        // part is node independent and normally arrives precomputed in `pre` (ComboPre role of the last block
        // launch); without it the same two GEMMs run here, in the same order, so both forms are bit-identical
        f32x16 acc[1][1];
        const int col0 = wave * 32;
        const int col = col0 + l31;
        if (pre) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = acc_row(r, hi) < nrows ? pre_v[r] : 0.0f;
        } else {
            zero_acc(acc);
            gemm_tile<1, 1, H>(buf + H, LDA, w.w0e, H, col0, acc);
            const float b = w.b0e[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += b;
        }
        gemm_tile<1, 1, H>(buf, LDA, w.w0, H, col0, acc);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            const float v = acc[0][0][r], sg = swishf(v);
            buf[row * LDA + col] = sg;
            if constexpr (SAVE) {
                if (row < nrows) {
                    sv.g0[(size_t)(e0 + row) * H + col] = v;
                    sv.gs0[(size_t)(e0 + row) * H + col] = sg;
                }
            }
        }
        __syncthreads();
    }
    if (wave < NW) {  // H -> H/2 and the final dot: the first H/64 waves
        f32x16 acc[1][1];
        const int col = wave * 32 + l31;
        zero_acc(acc);
        gemm_tile<1, 1, H>(buf, LDA, w.w1, H / 2, wave * 32, acc);
        const float b = w.b1[col], w2 = w.w2[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float g = acc[0][0][r] + b, sg = swishf(g);
            if constexpr (SAVE) {
                const int row = acc_row(r, hi);
                if (row < nrows) {
                    sv.g1[(size_t)(e0 + row) * (H / 2) + col] = g;
                    sv.gs1[(size_t)(e0 + row) * (H / 2) + col] = sg;
                }
            }
            float v = sg * w2;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            if (l31 == 0) s_red[wave * T + acc_row(r, hi)] = v;
        }
    }
    __syncthreads();
    if (tid < nrows) {
        float v = s_red[tid];
#pragma unroll
        for (int k = 1; k < NW; ++k) v += s_red[k * T + tid];
        edge_inv[e0 + tid] = v + w.b2[0];
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing: W[out][in] row-major -> Bp[in/4][out][in%4]
// ---------------------------------------------------------------------------------------------
__global__ void pack_linear_kernel(const float* __restrict__ W, float* __restrict__ Bp, int nout, int nin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nout * nin) return;
    const int s = idx & 3;
    const int j = (idx >> 2) % nout;
    const int k4 = (idx >> 2) / nout;
    Bp[idx] = W[(size_t)j * nin + k4 * 4 + s];
}
__global__ void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) dst[idx] = src[idx];
}

// Folded weights (common.hpp, FOLDED WEIGHTS): Cp = pack(A . B) with A [nout x lda] (columns a0 .. a0 + mid), B [mid x nin]
// row major, products accumulated in fp64 and rounded once; bias' = A . bb + ba likewise.
__global__ void fold_linear_kernel(const float* __restrict__ A, int lda, int a0, const float* __restrict__ B,
                                   const float* __restrict__ bb, const float* __restrict__ ba, int nout, int mid, int nin,
                                   float* __restrict__ Cp, float* __restrict__ bias) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nout * nin) {
        const int s = idx & 3;
        const int o = (idx >> 2) % nout;
        const int k = ((idx >> 2) / nout) * 4 + s;
        double acc = 0.0;
        for (int j = 0; j < mid; ++j) acc += (double)A[(size_t)o * lda + a0 + j] * (double)B[(size_t)j * nin + k];
        Cp[idx] = (float)acc;
    }
    if (idx < nout) {
        double acc = (double)ba[idx];
        for (int j = 0; j < mid; ++j) acc += (double)A[(size_t)idx * lda + a0 + j] * (double)bb[j];
        bias[idx] = (float)acc;
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static inline size_t lds_edge_embed(int H) { return (size_t)(T * (2 * H + 4) + 3 * T) * 4; }
static inline size_t lds_cfconv(int H) { return (size_t)(T * (H + 4) + 5 * T) * 4; }
static inline size_t lds_node(int H) { return (size_t)(TN * (H + 4) + 2 * TN) * 4; }
static inline size_t lds_pair(int H) { return (size_t)(T * (2 * H + 4) + (H / 64) * T + 3 * T) * 4; }

#define TSD_DISPATCH_H(H_, ...)                                     \
    switch (H_) {                                                   \
        case 64: { constexpr int HH = 64; __VA_ARGS__; } break;     \
        case 128: { constexpr int HH = 128; __VA_ARGS__; } break;   \
        case 256: { constexpr int HH = 256; __VA_ARGS__; } break;   \
        default: set_error("hidden=%d unsupported (64/128/256)", H_); return TSD_ERR_INVALID; \
    }

// The saving edge embedding of the training step on the f16 MFMA pipes (split16.hpp): the same chain in the reference's
// operation order (no fold), every GEMM operand tile as two f16 planes in LDS ([T][2H + 8] each), the dense matrices
// from the f16-plane arena, fp32 accumulation, fp32 results and saves.  Every conversion feeds the role's running max:
// the workgroup raises TSD_STATUS_RANGE when a value left the f16 range (the step is then recomputed in fp32).
template <int H>
__global__ __launch_bounds__(2 * H) void edge_embed_save_h_kernel(EdgeEmbedW w, tsd_edges ea_, float* __restrict__ out_a,
                                                                  int tiles_a, tsd_edges eb_, float* __restrict__ out_b,
                                                                  EmbedSave sv, int save_b_row, int32_t* range_status) {
    constexpr int LDH = ldh_of(2 * H);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Planes pl = planes_at(smem, T, LDH);
    float* s_d = smem + T * LDH;
    int* s_tr = reinterpret_cast<int*>(s_d + T);
    int* s_tp = s_tr + T;
    const bool second = (int)blockIdx.x >= tiles_a;
    const tsd_edges& e = second ? eb_ : ea_;
    float* __restrict__ edge_attr = second ? out_b : out_a;
    const int E = *e.count;
    const int e0 = (second ? (int)blockIdx.x - tiles_a : (int)blockIdx.x) * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32, col = col0 + l31;
    const int nrows = min(T, E - e0);
    const size_t srow0 = (size_t)(second ? save_b_row : 0) + e0;
    float amax = 0.0f;
    if (tid < T) {
        const int ee = e0 + tid;
        const bool v = ee < E;
        s_d[tid] = v ? e.dist[ee] : 0.0f;
        s_tr[tid] = v ? (int)e.type_r[ee] : 0;
        s_tp[tid] = v ? (int)e.type_p[ee] : 0;
        if (v) {
            sv.d[srow0 + tid] = s_d[tid];
            sv.tr[srow0 + tid] = (uint8_t)s_tr[tid];
            sv.tp[srow0 + tid] = (uint8_t)s_tp[tid];
        }
    }
    __syncthreads();
    {  // Linear(1,H) + swish: thread = (channel, half of the tile's rows)
        const int c = tid % H, r0 = (tid / H) * (T / 2);
        const float w0 = w.w0[c], b0 = w.b0[c];
#pragma unroll 8
        for (int r = r0; r < r0 + T / 2; ++r) {
            const float l = w0 * s_d[r] + b0, sl = swishf(l);
            planes_store1(pl, r * LDH + c, sl, amax);
            if (r < nrows) {
                sv.l0[(srow0 + r) * H + c] = l;
                sv.s0[(srow0 + r) * H + c] = sl;
            }
        }
    }
    __syncthreads();
    f32x16 accm[1][1], accx[1][1];
    hzero(accm, accx);
    hgemm_tile<1, 1, H, true>(pl, LDH, w.w1, H, col0, accm, accx);
    __syncthreads();
    {
        const float b = w.b1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            const float v = hval(accm[0][0], accx[0][0], r) + b;
            const float vr = v * w.bond_emb[s_tr[row] * H + col], vp = v * w.bond_emb[s_tp[row] * H + col];
            planes_store1(pl, row * LDH + col, vr, amax);
            planes_store1(pl, row * LDH + H + col, vp, amax);
            if (row < nrows) {
                sv.e[(srow0 + row) * H + col] = v;
                sv.c[(srow0 + row) * 2 * H + col] = vr;
                sv.c[(srow0 + row) * 2 * H + H + col] = vp;
            }
        }
    }
    __syncthreads();
    hzero(accm, accx);
    hgemm_tile<1, 1, 2 * H, true>(pl, LDH, w.cw0, H, col0, accm, accx);
    __syncthreads();
    {
        const float b = w.cb0[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            const float v = hval(accm[0][0], accx[0][0], r) + b, s1 = swishf(v);
            planes_store1(pl, row * LDH + col, s1, amax);
            if (row < nrows) {
                sv.c0[(srow0 + row) * H + col] = v;
                sv.s1[(srow0 + row) * H + col] = s1;
            }
        }
    }
    __syncthreads();
    hzero(accm, accx);
    hgemm_tile<1, 1, H, true>(pl, LDH, w.cw1, H, col0, accm, accx);
    {
        const float b = w.cb1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            if (row < nrows) edge_attr[(size_t)(e0 + row) * H + col] = hval(accm[0][0], accx[0][0], r) + b;
        }
    }
    range_report(amax, range_status);
}
int launch_edge_embed_save_h(const tsd_model_cfg& c, const float* W16, int cap_a, tsd_edges ea, float* out_a, int cap_b,
                             tsd_edges eb, float* out_b, hipStream_t st, const EmbedSave& save, int save_b_row,
                             int32_t* range_status) {
    const WeightLayout L = weight_layout(c);
    EdgeEmbedW w{W16 + L.bond_emb, W16 + L.emlp_w0, W16 + L.emlp_b0, W16 + L.emlp_w1, W16 + L.emlp_b1,
                 W16 + L.ecat_w0, W16 + L.ecat_b0, W16 + L.ecat_w1, W16 + L.ecat_b1};
    const int tiles_a = (cap_a + T - 1) / T, tiles_b = (cap_b + T - 1) / T;
    if (tiles_a + tiles_b == 0) return TSD_OK;
    if (c.hidden != 256) {
        set_error("edge_embed_save_h: hidden=%d has no split-f16 instance", c.hidden);
        return TSD_ERR_INVALID;
    }
    const size_t lds = (size_t)(T * ldh_of(2 * 256) + T) * 4 + 2 * T * sizeof(int);
    static DeviceOnce once;
    int r = allow_lds(edge_embed_save_h_kernel<256>, lds, once);
    if (r) return r;
    hipLaunchKernelGGL(edge_embed_save_h_kernel<256>, dim3(tiles_a + tiles_b), dim3(512), lds, st, w, ea, out_a, tiles_a, eb,
                       out_b, save, save_b_row, range_status);
    TSD_LAUNCH_CHECK("edge_embed_save_h");
    return TSD_OK;
}

int launch_edge_embed2(const tsd_model_cfg& c, const float* W, int cap_a, tsd_edges ea, float* out_a, int cap_b,
                       tsd_edges eb, float* out_b, int M, size_t out_stride, hipStream_t st, const UmapRole* umap,
                       const EmbedSave* save, int save_b_row, const EmbedFuse0* fuse0, bool fold) {
    const WeightLayout L = weight_layout(c);
    EdgeEmbedW w{W + L.bond_emb, W + L.emlp_w0, W + L.emlp_b0, W + L.emlp_w1, W + L.emlp_b1,
                 W + L.ecat_w0, W + L.ecat_b0, W + L.ecat_w1, W + L.ecat_b1};
    const int tiles_a = (cap_a + T - 1) / T, tiles_b = (cap_b + T - 1) / T;
    UmapRole um{};
    if (umap && umap->P > 0) {
        um = *umap;
        um.blocks = ((um.P > um.n_zero ? um.P : um.n_zero) + 2 * c.hidden - 1) / (2 * c.hidden);
    }
    if (tiles_a + tiles_b + um.blocks == 0) return TSD_OK;
    const size_t lds = lds_edge_embed(c.hidden);
    const dim3 grid(tiles_a + tiles_b + um.blocks, M);
    if (save && fuse0) {
        set_error("internal: edge_embed saves and the fused block-0 filters are exclusive");
        return TSD_ERR_INVALID;
    }
#define TSD_EE(HH, SV, FU, FO, SARG, FARG)                                                                     \
    {                                                                                                          \
        static DeviceOnce once;                                                                                \
        int r = allow_lds(edge_embed_kernel<HH, SV, FU, FO>, lds, once);                                       \
        if (r) return r;                                                                                       \
        hipLaunchKernelGGL((edge_embed_kernel<HH, SV, FU, FO>), grid, dim3(2 * HH), lds, st, w, ea, out_a, tiles_a, eb, \
                           out_b, L.total, out_stride, tiles_a + tiles_b, um, SARG, save_b_row, FARG);          \
    }
    if (save && fold) {
        set_error("internal: the saving edge embedding keeps the reference's operation order (no fold)");
        return TSD_ERR_INVALID;
    }
    if (save) {
        TSD_DISPATCH_H(c.hidden, TSD_EE(HH, true, false, false, *save, EmbedFuse0{}));
    } else if (fuse0 && fold) {
        TSD_DISPATCH_H(c.hidden, TSD_EE(HH, false, true, true, EmbedSave{}, *fuse0));
    } else if (fuse0) {
        TSD_DISPATCH_H(c.hidden, TSD_EE(HH, false, true, false, EmbedSave{}, *fuse0));
    } else if (fold) {
        TSD_DISPATCH_H(c.hidden, TSD_EE(HH, false, false, true, EmbedSave{}, EmbedFuse0{}));
    } else {
        TSD_DISPATCH_H(c.hidden, TSD_EE(HH, false, false, false, EmbedSave{}, EmbedFuse0{}));
    }
#undef TSD_EE
    TSD_LAUNCH_CHECK("edge_embed");
    return TSD_OK;
}

int launch_edge_embed(const tsd_model_cfg& c, const float* W, int capacity, tsd_edges e, float* edge_attr,
                      hipStream_t st) {
    return launch_edge_embed2(c, W, capacity, e, edge_attr, 0, e, edge_attr, 1, 0, st, nullptr, nullptr, 0, nullptr);
}

int launch_cfconv_layer(const tsd_model_cfg& c, const float* W, int layer, int capacity, tsd_edges e,
                        const float* edge_attr, const float* x1, float* agg, float* part, hipStream_t st) {
    const WeightLayout L = weight_layout(c);
    const float* B = W + L.layer0 + (size_t)layer * L.layer_stride;
    CfconvW w{B + L.L_nn0_w, B + L.L_nn0_b, B + L.L_nn2_w, B + L.L_nn2_b};
    const int tiles = (capacity + T - 1) / T;
    if (tiles == 0) return TSD_OK;
    const size_t lds = lds_cfconv(c.hidden);
    TSD_DISPATCH_H(c.hidden, {
        static DeviceOnce once; int r = allow_lds(cfconv_layer_kernel<HH>, lds, once);
        if (r) return r;
        hipLaunchKernelGGL(cfconv_layer_kernel<HH>, dim3(tiles), dim3(HH), lds, st, w, c.conv_cutoff, c.smooth_conv, e,
                           edge_attr, x1, agg, part);
    });
    TSD_LAUNCH_CHECK("cfconv_layer");
    return TSD_OK;
}

static inline size_t lds_filter(int H) { return (size_t)(T * (H + 4) + T) * 4; }

int launch_filter_gen(const tsd_model_cfg& c, const float* W, int capacity, tsd_edges e, const float* edge_attr,
                      float* Wf, int layer_base, int nlayers, hipStream_t st) {
    const WeightLayout L = weight_layout(c);
    const int tiles = (capacity + T - 1) / T;
    if (tiles == 0 || nlayers <= 0) return TSD_OK;
    const size_t lds = lds_filter(c.hidden);
    TSD_DISPATCH_H(c.hidden, {
        static DeviceOnce once; int r = allow_lds(filter_gen_kernel<HH>, lds, once);
        if (r) return r;
        hipLaunchKernelGGL(filter_gen_kernel<HH>, dim3(tiles, nlayers), dim3(HH), lds, st, W + L.layer0,
                           L.layer_stride, L.L_nn0_w, L.L_nn0_b, L.L_nn2_w, L.L_nn2_b, c.conv_cutoff, c.smooth_conv, e, edge_attr,
                           Wf, (size_t)capacity * HH, layer_base);
    });
    TSD_LAUNCH_CHECK("filter_gen");
    return TSD_OK;
}

int launch_node_update(const tsd_model_cfg& c, const float* W, int layer, int next_layer, int N,
                       const int32_t* row_ptr, const float* agg, const float* part, float* h, float* x1,
                       hipStream_t st) {
    const WeightLayout L = weight_layout(c);
    const float* B = W + L.layer0 + (size_t)layer * L.layer_stride;
    const float* Bn = next_layer >= 0 ? W + L.layer0 + (size_t)next_layer * L.layer_stride + L.L_lin1_w : nullptr;
    NodeW w{B + L.L_lin2_w, B + L.L_lin2_b, B + L.L_lin_w, B + L.L_lin_b, Bn};
    const int tiles = (N + TN - 1) / TN;
    if (tiles == 0) return TSD_OK;
    const size_t lds = lds_node(c.hidden);
    TSD_DISPATCH_H(c.hidden, {
        static DeviceOnce once; int r = allow_lds(node_update_kernel<HH, 0>, lds, once);
        if (r) return r;
        hipLaunchKernelGGL((node_update_kernel<HH, 0>), dim3(tiles), dim3(2 * HH), lds, st, w, N, row_ptr, agg, part,
                           h, x1);
    });
    TSD_LAUNCH_CHECK("node_update");
    return TSD_OK;
}

int launch_node_lin1(const tsd_model_cfg& c, const float* W, int layer, int N, const float* h, float* x1,
                     hipStream_t st) {
    const WeightLayout L = weight_layout(c);
    const float* B = W + L.layer0 + (size_t)layer * L.layer_stride;
    NodeW w{nullptr, nullptr, nullptr, nullptr, B + L.L_lin1_w};
    const int tiles = (N + TN - 1) / TN;
    if (tiles == 0) return TSD_OK;
    const size_t lds = lds_node(c.hidden);
    TSD_DISPATCH_H(c.hidden, {
        static DeviceOnce once; int r = allow_lds(node_update_kernel<HH, 1>, lds, once);
        if (r) return r;
        hipLaunchKernelGGL((node_update_kernel<HH, 1>), dim3(tiles), dim3(2 * HH), lds, st, w, N,
                           (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           const_cast<float*>(h), x1);
    });
    TSD_LAUNCH_CHECK("node_lin1");
    return TSD_OK;
}

int launch_pair_output(const tsd_model_cfg& c, const float* W, int capacity, tsd_edges e, const float* h,
                       const float* edge_attr, const int32_t* attr_row, float* edge_inv, int M, size_t h_stride,
                       size_t ea_stride, size_t inv_stride, hipStream_t st, const float* pre, size_t pre_stride,
                       const PairSave* save, bool folded, Prec prec) {
    if (prec.mode == PREC_H2) {
        if (pre) {
            set_error("internal: the split-f16 pair output takes no precomputed half");
            return TSD_ERR_INVALID;
        }
        return launch_pair_output_h(c, W, capacity, e, h, edge_attr, attr_row, edge_inv, M, h_stride, ea_stride, inv_stride,
                                    st, folded, prec.range_status, save, prec.narrow_filter_tiles);
    }
    const WeightLayout L = weight_layout(c);
    PairW w{W + L.out_w0, W + L.out_b0, W + L.out_w1, W + L.out_b1, W + L.out_w2, W + L.out_b2,
            folded ? W + L.out_w0f : W + L.out_w0 + (size_t)c.hidden * c.hidden,  // packed [k/4][out][k%4]: the k >= H half is contiguous
            folded ? W + L.out_b0f : W + L.out_b0};
    const int tiles = (capacity + T - 1) / T;
    if (tiles == 0) return TSD_OK;
    const size_t lds = lds_pair(c.hidden);
    if (save) {
        if (pre) {
            set_error("internal: pair_output saves need the unsplit first layer");
            return TSD_ERR_INVALID;
        }
        TSD_DISPATCH_H(c.hidden, {
            static DeviceOnce once; int r = allow_lds(pair_output_kernel<HH, true>, lds, once);
            if (r) return r;
            hipLaunchKernelGGL((pair_output_kernel<HH, true>), dim3(tiles, M), dim3(2 * HH), lds, st, w, e, h, edge_attr,
                               attr_row, edge_inv, L.total, h_stride, ea_stride, inv_stride, pre, pre_stride, *save);
        });
    } else {
        TSD_DISPATCH_H(c.hidden, {
            static DeviceOnce once; int r = allow_lds(pair_output_kernel<HH, false>, lds, once);
            if (r) return r;
            hipLaunchKernelGGL((pair_output_kernel<HH, false>), dim3(tiles, M), dim3(2 * HH), lds, st, w, e, h, edge_attr,
                               attr_row, edge_inv, L.total, h_stride, ea_stride, inv_stride, pre, pre_stride, PairSave{});
        });
    }
    TSD_LAUNCH_CHECK("pair_output");
    return TSD_OK;
}

static int pack_one(const float* src, float* dst, int nout, int nin, hipStream_t st) {
    const int n = nout * nin;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, nout, nin);
    TSD_LAUNCH_CHECK("pack_linear");
    return TSD_OK;
}
static int copy_one(const float* src, float* dst, int n, hipStream_t st) {
    hipLaunchKernelGGL(copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, n);
    TSD_LAUNCH_CHECK("copy");
    return TSD_OK;
}


// ---------------------------------------------------------------------------------------------
// f16-plane image of packed matrices (split16.hpp): fp32 packed [k/4][out][k%4]  ->  [k/16][plane][k%16/8][out][k%8] f16
// in the same bytes.  One thread per element; `count` equally shaped matrices `stride` floats apart (blockIdx.y).
// ---------------------------------------------------------------------------------------------
__global__ void split16_kernel(const float* __restrict__ src, float* __restrict__ dst, int nout, int nin, size_t stride) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nout * nin) return;
    src += (size_t)blockIdx.y * stride;
    f16* d = reinterpret_cast<f16*>(dst + (size_t)blockIdx.y * stride);
    const int col = idx % nout, k = idx / nout;
    const float a = src[((size_t)(k >> 2) * nout + col) * 4 + (k & 3)];
    const f16 h = (f16)a;
    const f16 l = (f16)((a - (float)h) * SPLIT_SCALE);
    const int ks = k >> 4, half = (k >> 3) & 1, e = k & 7;
    d[((((size_t)ks * 2 + 0) * 2 + half) * nout + col) * 8 + e] = h;
    d[((((size_t)ks * 2 + 1) * 2 + half) * nout + col) * 8 + e] = l;
}
// fp32 attribute rows [rows, H] -> plane rows (common.hpp ATTRIBUTE ROWS AS f16 PLANES): one thread per 8 channels
__global__ void attr_planes_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, long rows,
                                   int32_t* __restrict__ range_status) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = H / 8;
    if (idx >= rows * c8) return;
    const long r = idx / c8;
    const int c = (int)(idx % c8);
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + r * H + c * 8), a1 = *reinterpret_cast<const f32x4*>(src + r * H + c * 8 + 4);
    f16x8 h, l;
    // range check as in the producers of the fused forward (split16.hpp): |a| > 65504 and a conversion site -- here the 8
    // consecutive channels of one row that a thread converts -- whose non-zero values all lie below 2^-12
    float amax = 0.0f, m = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f16 hh, ll;
        const float a = i < 4 ? a0[i] : a1[i - 4];
        amax_upd(m, a);
        split1(a, hh, ll);
        h[i] = hh;
        l[i] = ll;
    }
    site_close(amax, m);
    range_report(amax, range_status);
    f16* d = reinterpret_cast<f16*>(dst + r * H);
    *reinterpret_cast<f16x8*>(d + c * 8) = h;
    *reinterpret_cast<f16x8*>(d + H + c * 8) = l;
}
int launch_attr_planes(int H, int64_t rows, const float* src, float* dst, int32_t* range_status, hipStream_t st) {
    if (rows <= 0) return TSD_OK;
    if (!hidden_supported(H)) {
        set_error("hidden=%d unsupported (64/128/256)", H);
        return TSD_ERR_INVALID;
    }
    const long n = rows * (H / 8);
    hipLaunchKernelGGL(attr_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, H, (long)rows,
                       range_status);
    TSD_LAUNCH_CHECK("attr_planes");
    return TSD_OK;
}
static int split_mats(const float* src, float* dst, int nout, int nin, int count, size_t stride, hipStream_t st) {
    if (count <= 0) return TSD_OK;
    if (nin % 16 != 0) {
        set_error("internal: split16 needs K %% 16 == 0 (K = %d)", nin);
        return TSD_ERR_INVALID;
    }
    const int n = nout * nin;
    hipLaunchKernelGGL(split16_kernel, dim3((n + 255) / 256, count), dim3(256), 0, st, src, dst, nout, nin, stride);
    TSD_LAUNCH_CHECK("split16");
    return TSD_OK;
}
int launch_pack_weights16(const tsd_model_cfg& c, const float* packed, float* packed16, hipStream_t st) {
    const WeightLayout L = weight_layout(c);
    const int H = c.hidden;
    int r;
    TSD_HIP(hipMemcpyAsync(packed16, packed, L.total * sizeof(float), hipMemcpyDeviceToDevice, st));
    if ((r = split_mats(packed + L.emlp_w1, packed16 + L.emlp_w1, H, H, 1, 0, st))) return r;
    if ((r = split_mats(packed + L.ecat_w0, packed16 + L.ecat_w0, H, 2 * H, 1, 0, st))) return r;
    if ((r = split_mats(packed + L.ecat_w1, packed16 + L.ecat_w1, H, H, 1, 0, st))) return r;
    const size_t lo[6] = {L.L_nn0_w, L.L_nn2_w, L.L_lin1_w, L.L_lin2_w, L.L_lin_w, L.L_nn0f_w};
    for (int i = 0; i < 6; ++i)
        if ((r = split_mats(packed + L.layer0 + lo[i], packed16 + L.layer0 + lo[i], H, H, c.num_convs, L.layer_stride, st)))
            return r;
    if ((r = split_mats(packed + L.out_w0, packed16 + L.out_w0, H, 2 * H, 1, 0, st))) return r;
    if ((r = split_mats(packed + L.out_w1, packed16 + L.out_w1, H / 2, H, 1, 0, st))) return r;
    if ((r = split_mats(packed + L.out_w0f, packed16 + L.out_w0f, H, H, 1, 0, st))) return r;
    return TSD_OK;
}
// Pre-flight of a weight arena for the split-f16 arithmetic: over n floats, out[0] = max |w|, out[1] = number of non-zero
// weights below 2^-14 (the f16 normal range: their hi plane is subnormal, the value keeps ~1.5e-11 absolute precision
// instead of 22 bits), out[2] = number beyond 65504 (not representable: the forward would report TSD_STATUS_RANGE),
// out[3] = n.  One workgroup-level reduction per block, atomics on four words (max on the non-negative bit pattern).
__global__ void weights_preflight_kernel(const float* __restrict__ w, size_t n, float* __restrict__ out) {
    float mx = 0.0f;
    unsigned small = 0, big = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = fabsf(w[i]);
        mx = fmaxf(mx, a);
        small += (a != 0.0f && a < 6.103515625e-05f) ? 1u : 0u;
        big += (!(a <= F16_MAX)) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off));
        small += __shfl_xor(small, off);
        big += __shfl_xor(big, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(mx));
        atomicAdd(reinterpret_cast<unsigned*>(out) + 4, small);
        atomicAdd(reinterpret_cast<unsigned*>(out) + 5, big);
    }
}
__global__ void weights_preflight_finish(float* out, size_t n) {
    const unsigned* c = reinterpret_cast<const unsigned*>(out) + 4;
    out[1] = (float)c[0];
    out[2] = (float)c[1];
    out[3] = (float)n;
}
int launch_weights_preflight(const float* w, size_t n, float* out8, hipStream_t st) {
    TSD_HIP(hipMemsetAsync(out8, 0, 8 * sizeof(float), st));
    if (n > 0) {
        hipLaunchKernelGGL(weights_preflight_kernel, dim3(256), dim3(256), 0, st, w, n, out8);
        TSD_LAUNCH_CHECK("weights_preflight");
    }
    hipLaunchKernelGGL(weights_preflight_finish, dim3(1), dim3(1), 0, st, out8, n);
    TSD_LAUNCH_CHECK("weights_preflight_finish");
    return TSD_OK;
}
int launch_bucket_weights16(const tsd_model_cfg& c, const float* bucket, int num_slots, float* out16, hipStream_t st) {
    if (num_slots <= 0) return TSD_OK;
    const size_t H = c.hidden, per = H * H + H;
    TSD_HIP(hipMemcpyAsync(out16, bucket, (size_t)num_slots * per * sizeof(float), hipMemcpyDeviceToDevice, st));
    return split_mats(bucket, out16, c.hidden, c.hidden, num_slots, per, st);
}

size_t raw_weight_floats(const tsd_model_cfg& c) {
    const size_t H = c.hidden, F = c.feat_dim, HH = H * H;
    size_t n = 100 * H + H + H + HH + H + 100 * (H / 2) + (H / 2) * F;
    n += (size_t)c.num_convs * (5 * HH + 4 * H);
    n += 2 * HH + H + HH / 2 + H / 2 + H / 2 + 1 + 2 * HH + H + HH + H;
    return n;
}

int launch_pack_weights(const tsd_model_cfg& c, const float* raw, float* packed, hipStream_t st) {
    const WeightLayout L = weight_layout(c);
    const int H = c.hidden, F = c.feat_dim, HH = H * H;
    const float* s = raw;
    int r;
    const float* raw_nn0_w[64];
    const float* raw_nn0_b[64];
    const float *raw_w0 = nullptr, *raw_b0 = nullptr, *raw_c2w = nullptr, *raw_c2b = nullptr;
#define CP(dst, n) { if ((r = copy_one(s, packed + (dst), (n), st))) return r; s += (n); }
#define PK(dst, nout, nin) { if ((r = pack_one(s, packed + (dst), (nout), (nin), st))) return r; s += (size_t)(nout) * (nin); }
    CP(L.bond_emb, 100 * H);
    CP(L.emlp_w0, H);  // Linear(1,H).weight is [H,1]
    CP(L.emlp_b0, H);
    PK(L.emlp_w1, H, H);
    CP(L.emlp_b1, H);
    CP(L.atom_emb, 100 * (H / 2));
    CP(L.atom_feat, (H / 2) * F);
    for (int l = 0; l < c.num_convs; ++l) {
        const size_t B = L.layer0 + (size_t)l * L.layer_stride;
        PK(B + L.L_lin1_w, H, H);
        PK(B + L.L_lin2_w, H, H);
        CP(B + L.L_lin2_b, H);
        raw_nn0_w[l] = s;
        PK(B + L.L_nn0_w, H, H);
        raw_nn0_b[l] = s;
        CP(B + L.L_nn0_b, H);
        PK(B + L.L_nn2_w, H, H);
        CP(B + L.L_nn2_b, H);
        PK(B + L.L_lin_w, H, H);
        CP(B + L.L_lin_b, H);
    }
    raw_w0 = s;
    PK(L.out_w0, H, 2 * H);
    raw_b0 = s;
    CP(L.out_b0, H);
    PK(L.out_w1, H / 2, H);
    CP(L.out_b1, H / 2);
    CP(L.out_w2, H / 2);
    CP(L.out_b2, 1);
    PK(L.ecat_w0, H, 2 * H);
    CP(L.ecat_b0, H);
    raw_c2w = s;
    PK(L.ecat_w1, H, H);
    raw_c2b = s;
    CP(L.ecat_b1, H);
    // folded forms: nn.0 of every block and the edge half of grad_dist_mlp.0 absorb edge_cat.2
    {
        const int nblk = (HH + 255) / 256;
        for (int l = 0; l < c.num_convs; ++l) {
            const size_t B = L.layer0 + (size_t)l * L.layer_stride;
            hipLaunchKernelGGL(fold_linear_kernel, dim3(nblk), dim3(256), 0, st, raw_nn0_w[l], H, 0, raw_c2w, raw_c2b,
                               raw_nn0_b[l], H, H, H, packed + B + L.L_nn0f_w, packed + B + L.L_nn0f_b);
        }
        hipLaunchKernelGGL(fold_linear_kernel, dim3(nblk), dim3(256), 0, st, raw_w0, 2 * H, H, raw_c2w, raw_c2b, raw_b0, H,
                           H, H, packed + L.out_w0f, packed + L.out_b0f);
        TSD_LAUNCH_CHECK("fold_linear");
    }
#undef CP
#undef PK
    (void)HH;
    if ((size_t)(s - raw) != raw_weight_floats(c)) {
        set_error("internal: raw weight walk mismatch");
        return TSD_ERR_INVALID;
    }
    return TSD_OK;
}

}  // namespace tsd

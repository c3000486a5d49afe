// kernels_misc.hip -- the HBM/latency-bound kernels around the MFMA chain:
// node embedding, stand-alone CFConv aggregation (T5), eq_transform, ensemble mean, sampler update.
#include "common.hpp"

namespace tsd {

template <int V>
struct VecOf {  // V consecutive channels of a row held by one lane
    typedef float type __attribute__((ext_vector_type(V)));
    static __device__ __forceinline__ float get(const type& x, int v) { return x[v]; }
};
template <>
struct VecOf<1> {
    typedef float type;
    static __device__ __forceinline__ float get(const type& x, int) { return x; }
};

// ---------------------------------------------------------------------------------------------
// A2: z = [Emb[atom] + Wf r_feat , Wf p_feat - Wf r_feat]      reference condensenc.py:193-198
// ---------------------------------------------------------------------------------------------
__global__ void node_embed_kernel(int N, int H, int F, const float* __restrict__ atom_emb,
                                  const float* __restrict__ wf, const int64_t* __restrict__ atom_type,
                                  const int64_t* __restrict__ r_feat, const int64_t* __restrict__ p_feat,
                                  float* __restrict__ z) {
    const int half = H / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * half) return;
    const int i = idx / half, c = idx % half;
    int a = (int)atom_type[i];
    a = a < 0 ? 0 : (a > 99 ? 99 : a);
    float fr = 0.0f, fp = 0.0f;
    for (int k = 0; k < F; ++k) {
        const float w = wf[c * F + k];
        fr = fmaf((float)r_feat[(size_t)i * F + k], w, fr);
        fp = fmaf((float)p_feat[(size_t)i * F + k], w, fp);
    }
    z[(size_t)i * H + c] = atom_emb[a * half + c] + fr;
    z[(size_t)i * H + half + c] = fp - fr;
}

int launch_node_embed(const tsd_model_cfg& c, const float* W, int N, const int64_t* atom_type,
                      const int64_t* r_feat, const int64_t* p_feat, float* z, hipStream_t st) {
    const WeightLayout L = weight_layout(c);
    const int n = N * (c.hidden / 2);
    if (n == 0) return TSD_OK;
    hipLaunchKernelGGL(node_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, st, N, c.hidden, c.feat_dim,
                       W + L.atom_emb, W + L.atom_feat, atom_type, r_feat, p_feat, z);
    TSD_LAUNCH_CHECK("node_embed");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// T5 alone: out[i] = sum_{e in row i} x1[dst[e]] * W[e]          reference schnet.py:102,106
// HBM-bound: W is streamed once (4H bytes/edge), x1 rows come from L2/MALL, one wave per row,
// each lane owns H/64 consecutive channels, edges in order (bit-identical to a sequential
// scatter_add: product rounded, then added).
// ---------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(256) void cfconv_aggregate_kernel(int N, const int32_t* __restrict__ row_ptr,
                                                               const int32_t* __restrict__ dst,
                                                               const int32_t* __restrict__ umap,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ x1,
                                                               float* __restrict__ out) {
    constexpr int V = H / 64;  // floats per lane: 4 (H=256), 2, 1
    typedef typename VecOf<V>::type vrow;
    // Row order.  Workgroups are dealt round-robin to the 8 XCDs, each with a private L2, and the rows of ONE graph
    // gather the same x1 rows: with the identity order the 16 workgroups of a 64-atom graph land on all 8 XCDs and
    // every L2 fetches the graph's x1 rows (r02 PMC: 4.87 GB fetched for 4.38 GB algorithmic).  The XCD-aware order
    // (TSD_AGG_SWZ=1: XCD x takes the contiguous range [x B/8, (x+1) B/8) of 4-row groups) was measured SLOWER at
    // configs[4] size, 785 vs 753 us per launch (tools/ab_agg.py, round 3): eight far-apart W streams cost more than
    // the re-fetched x1 rows (11 % of the bytes) save.  Identity order kept.
    const int B = gridDim.x, b = blockIdx.x;
    const int per = B >> 3;
#ifndef TSD_AGG_SWZ
#define TSD_AGG_SWZ 0
#endif
#ifndef TSD_AGG_U
#define TSD_AGG_U 8
#endif
#ifndef TSD_AGG_RUN
#define TSD_AGG_RUN 16  // consecutive 4-row groups kept on one XCD by TSD_AGG_SWZ == 2 (16 groups = one 64-atom graph)
#endif
    int grp = b;
    if (TSD_AGG_SWZ == 1 && b < (per << 3)) {
        grp = (b & 7) * per + (b >> 3);  // XCD x takes the contiguous range [x B/8, (x+1) B/8)
    } else if (TSD_AGG_SWZ == 2) {
        // runs of TSD_AGG_RUN consecutive groups go to ONE XCD, consecutive runs to consecutive XCDs: the XCDs work on
        // neighbouring graphs (one W window, as the identity order) and a graph's x1 rows are fetched by one L2
        constexpr int RUN = TSD_AGG_RUN;
        const int full = B / (8 * RUN) * (8 * RUN);
        if (b < full) {
            const int x = b & 7, r = b >> 3;
            grp = ((r / RUN) * 8 + x) * RUN + r % RUN;
        }
    }
    const int i = grp * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (wave-uniform: scalar row offsets)
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int e0 = row_ptr[i], e1 = row_ptr[i + 1];
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.0f;
    // The row's indices come by ONE coalesced load per 64 edges (lane l holds edge eb + l) and reach the row loads
    // through v_readlane (wave-uniform row bases in SGPRs); then U edges = 2 U row loads in flight per wave, the last
    // batch clamped to the row's last edge (its extra slots are loaded, not added).  The per-edge index loads of the
    // r02 form sat between the row loads with a vmcnt(0) each (ISA), and 63-edge rows spent 3 of 18 round trips in a
    // one-edge tail loop.
    constexpr int U = TSD_AGG_U;
    for (int eb = e0; eb < e1; eb += 64) {
        const int cnt = min(64, e1 - eb);
        const int ee = eb + min(lane, cnt - 1);
        const int jv = dst[ee];
        const int wv = umap ? umap[ee] : ee;
        for (int k = 0; k < cnt; k += U) {
            vrow w[U], x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = min(k + u, cnt - 1);
                const int j = __builtin_amdgcn_readlane(jv, kk);
                const int we = __builtin_amdgcn_readlane(wv, kk);
                w[u] = __builtin_nontemporal_load(reinterpret_cast<const vrow*>(W + (size_t)we * H + lane * V));
                x[u] = *reinterpret_cast<const vrow*>(x1 + (size_t)j * H + lane * V);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k + u < cnt) {
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        acc[v] = __fadd_rn(acc[v], __fmul_rn(VecOf<V>::get(x[u], v), VecOf<V>::get(w[u], v)));
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) out[(size_t)i * H + lane * V + v] = acc[v];
}

// ---------------------------------------------------------------------------------------------
// T5, windowed form (round 6): the same sums for launches that are many chip-fulls of rows (BASELINE configs[4]: 65 536
// rows, 4.2 GB of filters).  A workgroup owns AGW_ROWS consecutive destination rows.  Graphs are contiguous node ranges, so
// the source rows its edges name lie in a short window [lo, hi] of x1: the workgroup finds the window (min / max over its
// slice of `dst`, 16 KB), copies it into LDS ONCE (<= AGW_WIN rows = 64 KB at H = 256: a 64-atom graph) and every edge then
// costs one streamed filter row from HBM and one ds_read from LDS -- in the one-wave-per-row form above each edge's x1 row
// is a second 1-KB vector load served by L2, and every XCD's L2 fetches every graph's x1 rows (PMC: 4.86 GB per launch for
// 4.38 GB algorithmic, profiles/r05_pmc_traffic_c5.md).  A window that does not fit (a graph of more than AGW_WIN atoms, or
// rows that straddle several graphs) takes the global gather for that workgroup.  A wave walks the CONTIGUOUS edge range of
// its AGW_ROWS / AGW_WAVES consecutive rows in batches of U edges and closes a row where the CSR says so (wave-uniform
// control flow): no clamped tail batch per row.  Per row the edges are added in list order, product rounded, then added,
// from 0: bit-identical to the form above and to a sequential scatter_add.
// Measured at configs[4] size (tools/ab_agg.py, interleaved child processes, three boxes of the pool, round 6): the
// one-wave-per-row form 749 / 800 / 769 us per launch (0.73 / 0.68 / 0.71 of 8 TB/s), this form 677-713 us (0.77-0.81) --
// 6.4 TB/s, the rate a plain copy reaches on this chip (MI355X_MICROARCH.md).  Shape of the workgroup: 16 rows x 16 waves
// (one row per wave) 677 us, 16 x 8: 686, 32 x 8: 707, 64 x 16: 710; 8 edges in flight per wave 691 against 16: 707;
// a second batch in flight (TSD_AGW_DB) and XCD-aware group orders (TSD_AGW_SWZ): no gain (716 vs 713, 701 vs 686).
// ---------------------------------------------------------------------------------------------
#ifndef TSD_AGW_ROWS
#define TSD_AGW_ROWS 16
#endif
#ifndef TSD_AGW_WAVES
#define TSD_AGW_WAVES 16
#endif
#ifndef TSD_AGW_U
#define TSD_AGW_U 8
#endif
#ifndef TSD_AGW_SWZ
#define TSD_AGW_SWZ 0
#endif
#ifndef TSD_AGW_DB
#define TSD_AGW_DB 0  // 1: two batches of U filter rows in flight per wave (double buffer)
#endif
#ifndef TSD_AGW_MIN_ROWS
#define TSD_AGW_MIN_ROWS 16384  // rows of a launch from which the windowed form runs (0: never); below, one wave per row
#endif
constexpr int AGW_ROWS = TSD_AGW_ROWS, AGW_WAVES = TSD_AGW_WAVES, AGW_WIN = 64, AGW_RW = AGW_ROWS / AGW_WAVES;
static_assert(AGW_ROWS % AGW_WAVES == 0, "rows per wave");
constexpr size_t agw_lds_bytes(int H) { return (size_t)AGW_WIN * H * 4 + (AGW_ROWS + 1 + 2 * AGW_WAVES + 3) / 4 * 16; }

template <int H, bool XL>
__device__ __forceinline__ void agw_wave(const int* __restrict__ rp /* LDS: row_ptr[r0 ..] */, int ra, int rb, int r0, int lo,
                                         const float* __restrict__ xs, const int32_t* __restrict__ dst,
                                         const int32_t* __restrict__ umap, const float* __restrict__ W,
                                         const float* __restrict__ x1, float* __restrict__ out) {
    constexpr int V = H / 64, U = XL ? TSD_AGW_U : TSD_AGW_U / 2;  // (the global-gather fallback holds two rows per edge)
    typedef typename VecOf<V>::type vrow;
    const int lane = threadIdx.x & 63;
    const int ea = __builtin_amdgcn_readfirstlane(rp[ra - r0]), ez = __builtin_amdgcn_readfirstlane(rp[rb - r0]);
    int cur = ra, ce = __builtin_amdgcn_readfirstlane(rp[ra + 1 - r0]);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.0f;
    // rows that end at edge `enext` (the row just summed, and empty rows behind it) are stored and closed
    auto flush = [&](int enext) {
        while (cur < rb && ce == enext) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                out[(size_t)cur * H + lane * V + v] = acc[v];
                acc[v] = 0.0f;
            }
            ++cur;
            if (cur < rb) ce = __builtin_amdgcn_readfirstlane(rp[cur + 1 - r0]);
        }
    };
    flush(ea);
    for (int eb = ea; eb < ez; eb += 64) {
        const int cnt = min(64, ez - eb);
        const int ee = eb + min(lane, cnt - 1);
        const int jv = dst[ee] - (XL ? lo : 0);
        const int wv = umap ? umap[ee] : ee;
        auto issue = [&](vrow (&w)[U], vrow (&x)[XL ? 1 : U], int k) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = min(k + u, cnt - 1);
                const int we = __builtin_amdgcn_readlane(wv, kk);
                w[u] = __builtin_nontemporal_load(reinterpret_cast<const vrow*>(W + (size_t)we * H + lane * V));
                if constexpr (!XL) {
                    const int j = __builtin_amdgcn_readlane(jv, kk);
                    x[u] = *reinterpret_cast<const vrow*>(x1 + (size_t)j * H + lane * V);
                }
            }
        };
        // (the x rows of four edges are read from LDS together, ahead of their use: a ds_read per edge right before its
        // multiply would expose the LDS latency once per edge -- the row-closing branches keep the compiler from hoisting)
        auto consume = [&](vrow (&w)[U], vrow (&x)[XL ? 1 : U], int k) {
#pragma unroll
            for (int u0 = 0; u0 < U; u0 += 4) {
                vrow xg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (XL) {
                        const int j = __builtin_amdgcn_readlane(jv, min(k + u0 + g, cnt - 1));
                        xg[g] = *reinterpret_cast<const vrow*>(xs + j * H + lane * V);
                    } else {
                        xg[g] = x[u0 + g];
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int u = u0 + g;
                    if (k + u < cnt) {
#pragma unroll
                        for (int v = 0; v < V; ++v)
                            acc[v] = __fadd_rn(acc[v], __fmul_rn(VecOf<V>::get(xg[g], v), VecOf<V>::get(w[u], v)));
                        flush(eb + k + u + 1);
                    }
                }
            }
        };
        if constexpr (XL && TSD_AGW_DB) {
            // two batches in flight: batch k + U is requested before batch k is summed (the stream never drains inside a
            // 64-edge chunk)
            vrow wa[U], wb[U], xd[1];
            issue(wa, xd, 0);
            for (int k = 0; k < cnt; k += 2 * U) {
                if (k + U < cnt) issue(wb, xd, k + U);
                consume(wa, xd, k);
                if (k + 2 * U < cnt) issue(wa, xd, k + 2 * U);
                if (k + U < cnt) consume(wb, xd, k + U);
            }
        } else {
            for (int k = 0; k < cnt; k += U) {
                vrow w[U], x[XL ? 1 : U];
                issue(w, x, k);
                consume(w, x, k);
            }
        }
    }
}

template <int H>
__global__ __launch_bounds__(64 * AGW_WAVES) void cfconv_aggregate_win_kernel(int N, const int32_t* __restrict__ row_ptr,
                                                                              const int32_t* __restrict__ dst,
                                                                              const int32_t* __restrict__ umap,
                                                                              const float* __restrict__ W,
                                                                              const float* __restrict__ x1,
                                                                              float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float agw_smem[];
    constexpr int T = 64 * AGW_WAVES;
    float* xs = agw_smem;
    int* rp = reinterpret_cast<int*>(agw_smem + (size_t)AGW_WIN * H);
    int* red = rp + AGW_ROWS + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> row group.  Workgroups are dealt round-robin to the 8 XCDs; TSD_AGW_SWZ = 1 (variant builds) sends runs of
    // AGW_WIN / AGW_ROWS consecutive groups (the destination rows of one 64-atom graph: the same x1 window) to ONE XCD.
    int grp = blockIdx.x;
    if (TSD_AGW_SWZ) {
        constexpr int RUN = AGW_WIN / AGW_ROWS;
        const int full = (int)gridDim.x / (8 * RUN) * (8 * RUN);
        if (grp < full) {
            const int x = grp & 7, r = grp >> 3;
            grp = ((r / RUN) * 8 + x) * RUN + r % RUN;
        }
    }
    const int r0 = grp * AGW_ROWS, r1 = min(N, r0 + AGW_ROWS);
    if (tid <= r1 - r0) rp[tid] = row_ptr[r0 + tid];
    __syncthreads();
    const int e_lo = rp[0], e_hi = rp[r1 - r0];
    int lo = 0x7fffffff, hi = -1;
    for (int e = e_lo + tid; e < e_hi; e += T) {
        const int j = dst[e];
        lo = min(lo, j);
        hi = max(hi, j);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    if (lane == 0) {
        red[wave] = lo;
        red[AGW_WAVES + wave] = hi;
    }
    __syncthreads();
    lo = red[0];
    hi = red[AGW_WAVES];
#pragma unroll
    for (int w = 1; w < AGW_WAVES; ++w) {
        lo = min(lo, red[w]);
        hi = max(hi, red[AGW_WAVES + w]);
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    const bool xl = hi >= lo && hi - lo < AGW_WIN;  // (workgroup-uniform)
    if (xl) {
        const f32x4* src = reinterpret_cast<const f32x4*>(x1 + (size_t)lo * H);
        f32x4* d = reinterpret_cast<f32x4*>(xs);
        const int chunks = (hi - lo + 1) * (H / 4);
        for (int c = tid; c < chunks; c += T) d[c] = src[c];
        __syncthreads();
    }
    const int ra = r0 + wave * AGW_RW, rb = min(r1, ra + AGW_RW);
    if (ra >= rb) return;  // (no barrier below)
    if (xl) agw_wave<H, true>(rp, ra, rb, r0, lo, xs, dst, umap, W, x1, out);
    else agw_wave<H, false>(rp, ra, rb, r0, 0, xs, dst, umap, W, x1, out);
}

int launch_cfconv_aggregate(int H, int N, const int32_t* row_ptr, const int32_t* dst, const int32_t* umap,
                            const float* W, const float* x1, float* out, hipStream_t st) {
    if (N == 0) return TSD_OK;
    if (!hidden_supported(H)) {
        set_error("hidden=%d unsupported (64/128/256)", H);
        return TSD_ERR_INVALID;
    }
    if (TSD_AGW_MIN_ROWS > 0 && N >= TSD_AGW_MIN_ROWS) {
        const int blocks = (N + AGW_ROWS - 1) / AGW_ROWS;
        const size_t lds = agw_lds_bytes(H);
#define TSD_AGW(HH)                                                                                              \
    {                                                                                                            \
        static DeviceOnce once;                                                                                  \
        int r = allow_lds(cfconv_aggregate_win_kernel<HH>, lds, once);                                           \
        if (r) return r;                                                                                         \
        hipLaunchKernelGGL(cfconv_aggregate_win_kernel<HH>, dim3(blocks), dim3(64 * AGW_WAVES), lds, st, N, row_ptr, dst, \
                           umap, W, x1, out);                                                                    \
    }
        switch (H) {
            case 64: TSD_AGW(64) break;
            case 128: TSD_AGW(128) break;
            default: TSD_AGW(256) break;
        }
#undef TSD_AGW
        TSD_LAUNCH_CHECK("cfconv_aggregate_win");
        return TSD_OK;
    }
    const int blocks = (N + 3) / 4;
    switch (H) {
        case 64: hipLaunchKernelGGL(cfconv_aggregate_kernel<64>, dim3(blocks), dim3(256), 0, st, N, row_ptr, dst, umap, W, x1, out); break;
        case 128: hipLaunchKernelGGL(cfconv_aggregate_kernel<128>, dim3(blocks), dim3(256), 0, st, N, row_ptr, dst, umap, W, x1, out); break;
        default: hipLaunchKernelGGL(cfconv_aggregate_kernel<256>, dim3(blocks), dim3(256), 0, st, N, row_ptr, dst, umap, W, x1, out); break;
    }
    TSD_LAUNCH_CHECK("cfconv_aggregate");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// eq_transform                                                   reference models/geometry.py:22-30
// ---------------------------------------------------------------------------------------------
__global__ void eq_transform_atomic_kernel(int64_t E, const float* __restrict__ sd, const float* __restrict__ pos,
                                           const int64_t* __restrict__ ei, const float* __restrict__ len,
                                           float* __restrict__ score) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t i = ei[e], j = ei[E + e];
    const float inv = 1.0f / len[e], s = sd[e];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = __fmul_rn(__fmul_rn(inv, pos[3 * i + k] - pos[3 * j + k]), s);
        atomicAdd(score + 3 * i + k, v);
        atomicAdd(score + 3 * j + k, -v);
    }
}

int launch_eq_transform_atomic(int N, int64_t E, const float* sd, const float* pos, const int64_t* ei,
                               const float* len, float* score, hipStream_t st) {
    (void)N;
    if (E == 0) return TSD_OK;
    hipLaunchKernelGGL(eq_transform_atomic_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, E, sd, pos,
                       ei, len, score);
    TSD_LAUNCH_CHECK("eq_transform_atomic");
    return TSD_OK;
}

// deterministic form on the library's own out-edge list: one thread per node walks its row.
// first term: edges (i,j) of row i in order; second term: edges (j,i), j ascending == same order,
// -dd_dr(j,i) == dd_dr(i,j) exactly, score_d looked up through pair2out.
__global__ void eq_transform_rows_kernel(int N, const float* __restrict__ pos, const int32_t* __restrict__ pair_ptr,
                                         const int32_t* __restrict__ graph_ptr,
                                         const int32_t* __restrict__ node_graph, tsd_edges out,
                                         const int32_t* __restrict__ pair2out, const float* __restrict__ sd,
                                         float* __restrict__ score) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int lo = graph_ptr[node_graph[i]];
    const int il = i - lo;
    const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
    float ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    const int e1 = out.row_ptr[i + 1];
    for (int e = out.row_ptr[i]; e < e1; ++e) {
        const int j = out.dst[e];
        const int jl = j - lo;
        const float inv = 1.0f / out.dist[e];
        const float ux = __fmul_rn(inv, px - pos[3 * j]), uy = __fmul_rn(inv, py - pos[3 * j + 1]),
                    uz = __fmul_rn(inv, pz - pos[3 * j + 2]);
        const float s1 = sd[e];
        const int et = pair2out[pair_ptr[j] + il - (il > jl ? 1 : 0)];
        const float s2 = et >= 0 ? sd[et] : 0.0f;
        ax = __fadd_rn(ax, __fmul_rn(ux, s1));
        ay = __fadd_rn(ay, __fmul_rn(uy, s1));
        az = __fadd_rn(az, __fmul_rn(uz, s1));
        bx = __fadd_rn(bx, __fmul_rn(ux, s2));
        by = __fadd_rn(by, __fmul_rn(uy, s2));
        bz = __fadd_rn(bz, __fmul_rn(uz, s2));
    }
    score[3 * i] = ax + bx;
    score[3 * i + 1] = ay + by;
    score[3 * i + 2] = az + bz;
}

int launch_eq_transform_rows(int N, const float* pos, const int32_t* pair_ptr, const int32_t* graph_ptr,
                             const int32_t* node_graph, tsd_edges out, const int32_t* pair2out,
                             const float* sd, float* score, hipStream_t st) {
    if (N == 0) return TSD_OK;
    hipLaunchKernelGGL(eq_transform_rows_kernel, dim3((N + 127) / 128), dim3(128), 0, st, N, pos, pair_ptr,
                       graph_ptr, node_graph, out, pair2out, sd, score);
    TSD_LAUNCH_CHECK("eq_transform_rows");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// ensemble mean, reference sampler.py:96-111: edge_inv += out[0] (in order), then /= M
// ---------------------------------------------------------------------------------------------
__global__ void ensemble_mean_kernel(int M, int PU, tsd_edges out, const float* __restrict__ inv_u,
                                     float* __restrict__ mean) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= *out.count) return;
    const int u = out.umap[e];
    float s = inv_u[u];
    for (int m = 1; m < M; ++m) s = __fadd_rn(s, inv_u[(size_t)m * PU + u]);
    mean[e] = s / (float)M;
}

int launch_ensemble_mean(int M, int P, tsd_edges out, const float* inv_u, float* mean, hipStream_t st) {
    if (P == 0) return TSD_OK;
    hipLaunchKernelGGL(ensemble_mean_kernel, dim3((P + 255) / 256), dim3(256), 0, st, M, P / 2, out, inv_u, mean);
    TSD_LAUNCH_CHECK("ensemble_mean");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// sampler update: clip_norm, LD / DDPM step, NaN flag, centre per graph, optional clamp
// reference models/sampler.py:208-253, 260-268.  One wave per graph.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sampler_step_kernel(int kind, const int32_t* __restrict__ graph_ptr,
                                                          const float* __restrict__ score,
                                                          const float* __restrict__ noise,
                                                          const float* __restrict__ coefs, float clip,
                                                          float clip_pos, float* __restrict__ pos,
                                                          float* __restrict__ traj, int32_t* __restrict__ status,
                                                          const int32_t* __restrict__ step_ctr, int N) {
    if (step_ctr) {  // device-resident loop: this step's slices of the per-step tables
        const size_t k = (size_t)*step_ctr;
        coefs += k * TSD_STEP_COEFS;
        noise += k * 3 * (size_t)N;
        if (traj) traj += k * 3 * (size_t)N;
    }
    const int g = blockIdx.x;
    const int lo = graph_ptr[g], hi = graph_ptr[g + 1];
    const int lane = threadIdx.x;
    float c[TSD_STEP_COEFS];
#pragma unroll
    for (int k = 0; k < TSD_STEP_COEFS; ++k) c[k] = coefs[k];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    bool bad = false;
    for (int i = lo + lane; i < hi; i += 64) {
        float v[3], p[3], nz[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[k] = score[3 * i + k];
            p[k] = pos[3 * i + k];
            nz[k] = noise[3 * i + k];
        }
        // clip_norm (sampler.py:265-268)
        const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2])));
        const float denom = norm > clip ? clip / norm : 1.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float eps = __fmul_rn(v[k], denom);
            float nx;
            if (kind == 0) {  // LD, sampler.py:238-244
                nx = __fadd_rn(__fadd_rn(p[k], __fmul_rn(c[0], eps) / c[1]), __fmul_rn(nz[k], c[2]));
            } else {  // DDPM, sampler.py:215-236
                const float e = -eps;
                const float pos_C = __fmul_rn(c[0], p[k]);
                const float pos0 = __fsub_rn(__fmul_rn(c[1], pos_C), __fmul_rn(c[2], e));
                const float mean = __fadd_rn(__fmul_rn(c[3], pos0), __fmul_rn(c[4], pos_C)) / c[5];
                nx = __fadd_rn(mean, __fmul_rn(c[6], nz[k])) / c[7];
            }
            bad |= (nx != nx);
            p[k] = nx;
        }
        pos[3 * i] = p[0];
        pos[3 * i + 1] = p[1];
        pos[3 * i + 2] = p[2];
        sx += p[0];
        sy += p[1];
        sz += p[2];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sx += __shfl_xor(sx, off);
        sy += __shfl_xor(sy, off);
        sz += __shfl_xor(sz, off);
    }
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(status, TSD_STATUS_NAN);
    const float cnt = (float)max(hi - lo, 1);
    const float mx = sx / cnt, my = sy / cnt, mz = sz / cnt;
    for (int i = lo + lane; i < hi; i += 64) {  // center_pos (sampler.py:260-262); same lane wrote these
        float x = pos[3 * i] - mx, y = pos[3 * i + 1] - my, z = pos[3 * i + 2] - mz;
        if (clip_pos >= 0.0f) {
            x = fminf(fmaxf(x, -clip_pos), clip_pos);
            y = fminf(fmaxf(y, -clip_pos), clip_pos);
            z = fminf(fmaxf(z, -clip_pos), clip_pos);
        }
        pos[3 * i] = x;
        pos[3 * i + 1] = y;
        pos[3 * i + 2] = z;
        if (traj) {
            traj[3 * i] = x;
            traj[3 * i + 1] = y;
            traj[3 * i + 2] = z;
        }
    }
}

int launch_sampler_step(int kind, int N, int G, const int32_t* graph_ptr, const float* score, const float* noise,
                        const float* coefs, float clip, float clip_pos, float* pos, float* traj,
                        int32_t* status, const int32_t* step_ctr, hipStream_t st) {
    if (G == 0) return TSD_OK;
    hipLaunchKernelGGL(sampler_step_kernel, dim3(G), dim3(64), 0, st, kind, graph_ptr, score, noise, coefs, clip,
                       clip_pos, pos, traj, status, step_ctr, N);
    TSD_LAUNCH_CHECK("sampler_step");
    return TSD_OK;
}

}  // namespace tsd

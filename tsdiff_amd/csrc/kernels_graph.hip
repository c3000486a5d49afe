// kernels_graph.hip -- device-side graph construction (HBM/latency-bound integer + fp32 compare work).
//
//   topology  (once per batch) : k-hop bond-graph extension of the reactant and product graphs
//                                reference models/common.py:115-202 (_extend_ts_graph_order)
//   geometry  (every step)     : radius membership, union with the local edges, types, edge_length,
//                                row-major compaction   reference models/common.py:205-223,328-384,
//                                models/epsnet/condensenc.py:117-154, models/geometry.py:18-19
//
// The reference builds dense (N,N) matrices over the whole batch and takes matrix powers; graphs
// never interact, so here every graph is handled on its own: ordered intra-graph pairs (i != j) are
// enumerated row-major -- pair p of row i is the k-th other node of i's graph -- which is exactly the
// reference's (edge_index[0], edge_index[1]) sort order, so stream compaction of the member pairs
// yields the reference edge list without a sort.
#include "common.hpp"

namespace tsd {

__device__ __forceinline__ int type_of(int bond, int hop, int order) {
    // common.py:163-167: bond type, or NUM_BOND_TYPES + hop - 1 for 2 <= hop <= order, else 0
    return bond ? bond : ((hop >= 2 && hop <= order) ? (TSD_NUM_BOND_TYPES + hop - 1) : 0);
}

// ---------------------------------------------------------------------------------------------
// topology
// ---------------------------------------------------------------------------------------------
__global__ void node_map_kernel(int N, int G, const int32_t* __restrict__ graph_ptr,
                                const int32_t* __restrict__ pair_base, int32_t* __restrict__ node_graph,
                                int32_t* __restrict__ pair_ptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    if (i == N) {
        pair_ptr[N] = pair_base[G];
        return;
    }
    int lo = 0, hi = G;  // largest g with graph_ptr[g] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (graph_ptr[mid] <= i) lo = mid; else hi = mid;
    }
    const int n = graph_ptr[lo + 1] - graph_ptr[lo];
    node_graph[i] = lo;
    pair_ptr[i] = pair_base[lo] + (i - graph_ptr[lo]) * (n - 1);
}

__global__ void bond_scatter_kernel(int N, int64_t nb, const int64_t* __restrict__ bond_index,
                                    const int64_t* __restrict__ bond_type,
                                    const int32_t* __restrict__ graph_ptr,
                                    const int32_t* __restrict__ node_graph,
                                    const int32_t* __restrict__ pair_ptr, uint16_t* __restrict__ pair_code,
                                    int32_t* __restrict__ status) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int64_t a = bond_index[b], c = bond_index[nb + b], t = bond_type[b];
    if (a < 0 || a >= N || c < 0 || c >= N || a == c || t < 0 ||
        t >= TSD_NUM_BOND_TYPES * TSD_NUM_BOND_TYPES || node_graph[a] != node_graph[c]) {
        atomicOr(status, TSD_STATUS_BAD_BOND);
        return;
    }
    const int lo = graph_ptr[node_graph[a]];
    const int il = (int)a - lo, jl = (int)c - lo;
    const int r = (int)(t / TSD_NUM_BOND_TYPES), p = (int)(t % TSD_NUM_BOND_TYPES);  // common.py:148,153
    pair_code[pair_ptr[a] + jl - (jl > il ? 1 : 0)] = (uint16_t)(r | (p << 5));
}

// one workgroup per graph; LDS: two n*n u8 hop matrices (255 = not reached yet)
__global__ __launch_bounds__(256) void hop_kernel(const int32_t* __restrict__ graph_ptr,
                                                  const int32_t* __restrict__ pair_base,
                                                  uint16_t* __restrict__ pair_code, int max_order,
                                                  int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int g = blockIdx.x;
    const int n = graph_ptr[g + 1] - graph_ptr[g];
    const int nn = n * n;
    unsigned char* hopR = hsm;
    unsigned char* hopP = hsm + nn;
    uint16_t* code = pair_code + pair_base[g];
    for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
        const int i = idx / n, j = idx % n;
        unsigned char r = 0, p = 0;
        if (i != j) {
            const int c = code[i * (n - 1) + j - (j > i ? 1 : 0)];
            r = (c & 31) ? 1 : 255;
            p = ((c >> 5) & 31) ? 1 : 255;
        }
        hopR[idx] = r;
        hopP[idx] = p;
    }
    __syncthreads();
    // level-synchronous BFS == shortest directed path == first power of (A+I) that reaches (i,j)
    for (int level = 2; level <= max_order; ++level) {
        for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
            const int i = idx / n, j = idx % n;
            // branch-free over k (no early exit: the LDS reads of a trip pipeline instead of forming a chain of
            // n dependent round trips; the largest graph of the batch sets the kernel's duration)
            const bool needR = hopR[idx] == 255, needP = hopP[idx] == 255;
            if (needR || needP) {
                int fr = 0, fp = 0;
                const unsigned char* rowR = hopR + i * n;
                const unsigned char* rowP = hopP + i * n;
#pragma unroll 8
                for (int k = 0; k < n; ++k) {
                    fr |= (int)(rowR[k] == level - 1) & (int)(hopR[k * n + j] == 1);
                    fp |= (int)(rowP[k] == level - 1) & (int)(hopP[k * n + j] == 1);
                }
                // (entries written this level hold `level`, never `level - 1` or 1: readers of this level are unaffected)
                if (needR && fr) hopR[idx] = (unsigned char)level;
                if (needP && fp) hopP[idx] = (unsigned char)level;
            }
        }
        __syncthreads();
    }
    bool asym = false;
    for (int idx = threadIdx.x; idx < nn; idx += blockDim.x) {
        const int i = idx / n, j = idx % n;
        if (i == j) continue;
        const int o = i * (n - 1) + j - (j > i ? 1 : 0);
        const int ot = j * (n - 1) + i - (i > j ? 1 : 0);
        const int hr = hopR[idx] == 255 ? 0 : hopR[idx];
        const int hp = hopP[idx] == 255 ? 0 : hopP[idx];
        const int hrt = hopR[j * n + i] == 255 ? 0 : hopR[j * n + i];
        const int hpt = hopP[j * n + i] == 255 ? 0 : hopP[j * n + i];
        const int c = code[o] & 1023, ct = code[ot] & 1023;
        if (c != ct || hr != hrt || hp != hpt) asym = true;
        // distinct idx write distinct o; the low 10 bits read by other threads are unchanged
        code[o] = (uint16_t)(c | (hr << 10) | (hp << 13));
    }
    if (asym) atomicOr(status, TSD_STATUS_ASYMMETRIC);
}

int launch_topology(int N, int G, int P, int64_t nb, const int32_t* graph_ptr, const int32_t* pair_base,
                    const int64_t* bond_index, const int64_t* bond_type, int max_order, int max_n,
                    int32_t* node_graph, int32_t* pair_ptr, uint16_t* pair_code, int32_t* status,
                    hipStream_t st) {
    if (max_order < 1 || max_order > 7) {
        set_error("edge order %d outside 1..7", max_order);
        return TSD_ERR_INVALID;
    }
    if (max_n > TSD_MAX_GRAPH_NODES) {
        set_error("graph with %d atoms exceeds TSD_MAX_GRAPH_NODES=%d", max_n, TSD_MAX_GRAPH_NODES);
        return TSD_ERR_UNSUPPORTED;
    }
    if (P > 0) TSD_HIP(hipMemsetAsync(pair_code, 0, (size_t)P * sizeof(uint16_t), st));
    hipLaunchKernelGGL(node_map_kernel, dim3((N + 1 + 255) / 256), dim3(256), 0, st, N, G, graph_ptr, pair_base,
                       node_graph, pair_ptr);
    TSD_LAUNCH_CHECK("node_map");
    if (nb > 0) {
        hipLaunchKernelGGL(bond_scatter_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, N, nb,
                           bond_index, bond_type, graph_ptr, node_graph, pair_ptr, pair_code, status);
        TSD_LAUNCH_CHECK("bond_scatter");
    }
    if (G > 0) {
        const size_t lds = (size_t)2 * max_n * max_n + 16;
        static DeviceOnce once;
        if (once.first_on_current_device())
            TSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hop_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(hop_kernel, dim3(G), dim3(256), lds, st, graph_ptr, pair_base, pair_code, max_order,
                           status);
        TSD_LAUNCH_CHECK("hop");
    }
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// geometry: one wave per row (node i); the row's pairs are the other nodes of its graph in order
// ---------------------------------------------------------------------------------------------
struct PairEval {
    float d2, d;
    int tr_enc, tp_enc, tr_out, tp_out;
    bool in_enc, in_out;
    // an out edge whose (d, type_r, type_p) equal its enc edge's has the SAME edge embedding
    // (same kernel, same inputs, same weights); only the others are embedded a second time
    __device__ __forceinline__ bool needs_own_attr() const {
        return in_out && (!in_enc || tr_enc != tr_out || tp_enc != tp_out);
    }
};

__device__ __forceinline__ PairEval eval_pair_xyz(float xi, float yi, float zi, float xj, float yj, float zj,
                                                  int code, int order_enc, int order_out, float cut2) {
    PairEval r;
    // models/geometry.py:18-19  (pos[row] - pos[col]).norm()
    const float dx = xi - xj, dy = yi - yj, dz = zi - zj;
    r.d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    r.d = sqrtf(r.d2);
    const int bR = code & 31, bP = (code >> 5) & 31, hR = (code >> 10) & 7, hP = (code >> 13) & 7;
    r.tr_enc = type_of(bR, hR, order_enc);
    r.tp_enc = type_of(bP, hP, order_enc);
    r.tr_out = type_of(bR, hR, order_out);
    r.tp_out = type_of(bP, hP, order_out);
    const bool in_radius = r.d2 < cut2;  // torch_cluster radius: dist^2 < r^2, no self loops
    r.in_enc = in_radius || (r.tr_enc | r.tp_enc) != 0;
    r.in_out = in_radius || (r.tr_out | r.tp_out) != 0;
    return r;
}

__device__ __forceinline__ PairEval eval_pair(const float* __restrict__ pos, int i, int j, int code,
                                              int order_enc, int order_out, float cut2) {
    return eval_pair_xyz(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], pos[3 * j], pos[3 * j + 1], pos[3 * j + 2], code,
                         order_enc, order_out, cut2);
}

// per-row member counts of the five lists: 0 enc, 1 out, 2 enc_u, 3 out_u, 4 diff_u  (u: j > i only)
constexpr int NLIST = 5;

__global__ __launch_bounds__(256) void pair_count_kernel(int N, const float* __restrict__ pos,
                                                         const int32_t* __restrict__ graph_ptr,
                                                         const int32_t* __restrict__ node_graph,
                                                         const int32_t* __restrict__ pair_ptr,
                                                         const uint16_t* __restrict__ pair_code,
                                                         int order_enc, int order_out, float cut2,
                                                         int32_t* __restrict__ cnt /* [NLIST][N+1] */) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int lo = graph_ptr[node_graph[i]];
    const int il = i - lo;
    const int p0 = pair_ptr[i], np = pair_ptr[i + 1] - p0;
    int c[NLIST] = {0, 0, 0, 0, 0};
    for (int k0 = 0; k0 < np; k0 += 64) {
        const int k = k0 + lane;
        bool m[NLIST] = {false, false, false, false, false};
        if (k < np) {
            const int j = lo + k + (k >= il ? 1 : 0);
            const PairEval r = eval_pair(pos, i, j, pair_code[p0 + k], order_enc, order_out, cut2);
            const bool up = j > i;
            m[0] = r.in_enc;
            m[1] = r.in_out;
            m[2] = r.in_enc && up;
            m[3] = r.in_out && up;
            m[4] = r.needs_own_attr() && up;
        }
#pragma unroll
        for (int q = 0; q < NLIST; ++q) c[q] += __popcll(__ballot(m[q]));
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NLIST; ++q) cnt[(size_t)q * (N + 1) + i] = c[q];
    }
}

struct ScanOut {
    int32_t* row_ptr[NLIST];
    int32_t* total[NLIST];
};

// exclusive scan of NLIST int arrays of length N (+ total at [N]); single workgroup of 1024 threads:
// per-thread chunk sums, inclusive scan inside each wave by shuffles, 16 wave totals combined through LDS
__global__ __launch_bounds__(1024) void scan_kernel(int N, const int32_t* __restrict__ cnt, ScanOut o,
                                                    int32_t* __restrict__ advance) {
    __shared__ int wtot[NLIST][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (advance && t == 0) *advance += 1;  // device-side step counter of the sampling loop
    const int per = (N + 1023) / 1024;
    const int beg = min(N, t * per), end = min(N, beg + per);
    int x[NLIST], inc[NLIST];
#pragma unroll
    for (int q = 0; q < NLIST; ++q) {
        int a = 0;
        for (int i = beg; i < end; ++i) a += cnt[(size_t)q * (N + 1) + i];
        x[q] = a;
        int v = a;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(v, off);
            if (lane >= off) v += u;
        }
        inc[q] = v;
        if (lane == 63) wtot[q][wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NLIST; ++q) {
        int base = 0, all = 0;
        for (int w = 0; w < 16; ++w) {
            const int v = wtot[q][w];
            if (w < wave) base += v;
            all += v;
        }
        int r = base + inc[q] - x[q];  // exclusive prefix of this thread's chunk
        for (int i = beg; i < end; ++i) {
            o.row_ptr[q][i] = r;
            r += cnt[(size_t)q * (N + 1) + i];
        }
        if (t == 0) {
            o.row_ptr[q][N] = all;
            *o.total[q] = all;
        }
    }
}

__global__ __launch_bounds__(256) void pair_fill_kernel(int N, const float* __restrict__ pos,
                                                        const int32_t* __restrict__ graph_ptr,
                                                        const int32_t* __restrict__ node_graph,
                                                        const int32_t* __restrict__ pair_ptr,
                                                        const uint16_t* __restrict__ pair_code,
                                                        int order_enc, int order_out, float cut2,
                                                        tsd_geometry g, int P) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int lo = graph_ptr[node_graph[i]];
    const int il = i - lo;
    const int p0 = pair_ptr[i], np = pair_ptr[i + 1] - p0;
    const int PU = P / 2;
    int base[NLIST] = {g.enc.row_ptr[i], g.out.row_ptr[i], g.enc_u.row_ptr[i], g.out_u.row_ptr[i],
                       g.diff_u.row_ptr[i]};
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (int k0 = 0; k0 < np; k0 += 64) {
        const int k = k0 + lane;
        PairEval r;
        r.in_enc = r.in_out = false;
        int j = 0;
        bool up = false;
        if (k < np) {
            j = lo + k + (k >= il ? 1 : 0);
            r = eval_pair(pos, i, j, pair_code[p0 + k], order_enc, order_out, cut2);
            up = j > i;
        }
        const bool m[NLIST] = {r.in_enc, r.in_out, r.in_enc && up, r.in_out && up,
                               (k < np) && up && r.needs_own_attr()};
        int at[NLIST];
#pragma unroll
        for (int q = 0; q < NLIST; ++q) {
            const unsigned long long b = __ballot(m[q]);
            at[q] = base[q] + __popcll(b & lower);
            base[q] += __popcll(b);
        }
        if (m[0]) {
            const int e = at[0];
            g.enc.src[e] = i;
            g.enc.dst[e] = j;
            g.enc.dist[e] = r.d;
            g.enc.type_r[e] = (uint8_t)r.tr_enc;
            g.enc.type_p[e] = (uint8_t)r.tp_enc;
            g.enc.pair_id[e] = p0 + k;
        }
        if (m[1]) {
            const int e = at[1];
            g.out.src[e] = i;
            g.out.dst[e] = j;
            g.out.dist[e] = r.d;
            g.out.type_r[e] = (uint8_t)r.tr_out;
            g.out.type_p[e] = (uint8_t)r.tp_out;
            g.out.pair_id[e] = p0 + k;
        }
        if (m[2]) {
            const int e = at[2];
            g.enc_u.src[e] = i;
            g.enc_u.dst[e] = j;
            g.enc_u.dist[e] = r.d;
            g.enc_u.type_r[e] = (uint8_t)r.tr_enc;
            g.enc_u.type_p[e] = (uint8_t)r.tp_enc;
            g.enc_u.pair_id[e] = p0 + k;
        }
        if (m[3]) {
            const int e = at[3];
            g.out_u.src[e] = i;
            g.out_u.dst[e] = j;
            g.out_u.dist[e] = r.d;
            g.out_u.type_r[e] = (uint8_t)r.tr_out;
            g.out_u.type_p[e] = (uint8_t)r.tp_out;
            g.out_u.pair_id[e] = p0 + k;
            g.attr_row[e] = m[4] ? PU + at[4] : at[2];  // own embedding, or the enc_u edge's row
        }
        if (m[4]) {
            const int e = at[4];
            g.diff_u.dist[e] = r.d;
            g.diff_u.type_r[e] = (uint8_t)r.tr_out;
            g.diff_u.type_p[e] = (uint8_t)r.tp_out;
        }
        if (k < np) {
            g.pair2out[p0 + k] = m[1] ? at[1] : -1;
            if (up) {
                g.pair2u[p0 + k] = m[2] ? at[2] : -1;
                g.pair2u[(size_t)P + p0 + k] = m[3] ? at[3] : -1;
            }
        }
    }
}

// directed edge -> index of its undirected pair (needs every row's fill to be complete: own launch)
__global__ void edge_umap_kernel(tsd_geometry g, const int32_t* __restrict__ graph_ptr,
                                 const int32_t* __restrict__ node_graph, const int32_t* __restrict__ pair_ptr,
                                 int P) {
    edge_umap_body(g, graph_ptr, node_graph, pair_ptr, P, blockIdx.x * blockDim.x + threadIdx.x);
}

size_t geometry_scratch_ints(int N, int P) {
    (void)P;
    return (size_t)NLIST * (N + 1);
}

int launch_geometry_count(const tsd_model_cfg& c, int N, const float* pos, const int32_t* graph_ptr,
                          const int32_t* node_graph, const int32_t* pair_ptr, const uint16_t* pair_code,
                          tsd_geometry g, hipStream_t st) {
    if (N == 0) return TSD_OK;
    const float cut2 = c.edge_cutoff * c.edge_cutoff;
    hipLaunchKernelGGL(pair_count_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, pos, graph_ptr, node_graph, pair_ptr,
                       pair_code, c.edge_order, c.pred_edge_order, cut2, g.scratch);
    TSD_LAUNCH_CHECK("pair_count");
    return TSD_OK;
}

// scan of the per-row counts (already in g.scratch) + fill + umap
int launch_geometry_lists(const tsd_model_cfg& c, int N, int P, const float* pos, const int32_t* graph_ptr,
                          const int32_t* node_graph, const int32_t* pair_ptr, const uint16_t* pair_code,
                          tsd_geometry g, int32_t* advance, hipStream_t st, bool skip_umap) {
    const float cut2 = c.edge_cutoff * c.edge_cutoff;
    ScanOut so;
    tsd_edges* lists[NLIST] = {&g.enc, &g.out, &g.enc_u, &g.out_u, &g.diff_u};
    for (int q = 0; q < NLIST; ++q) {
        so.row_ptr[q] = lists[q]->row_ptr;
        so.total[q] = lists[q]->count;
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, N, g.scratch, so, advance);
    TSD_LAUNCH_CHECK("scan");
    if (N > 0) {
        hipLaunchKernelGGL(pair_fill_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, pos, graph_ptr, node_graph,
                           pair_ptr, pair_code, c.edge_order, c.pred_edge_order, cut2, g, P);
        TSD_LAUNCH_CHECK("pair_fill");
    }
    if (P > 0 && !skip_umap) {  // (skipped when the caller runs the umap as a role of the embedding launch)
        hipLaunchKernelGGL(edge_umap_kernel, dim3((P + 255) / 256), dim3(256), 0, st, g, graph_ptr, node_graph,
                           pair_ptr, P);
        TSD_LAUNCH_CHECK("edge_umap");
    }
    return TSD_OK;
}

int launch_geometry(const tsd_model_cfg& c, int N, int G, int P, const float* pos, const int32_t* graph_ptr,
                    const int32_t* node_graph, const int32_t* pair_ptr, const uint16_t* pair_code,
                    tsd_geometry g, hipStream_t st) {
    (void)G;
    int r = launch_geometry_count(c, N, pos, graph_ptr, node_graph, pair_ptr, pair_code, g, st);
    if (r) return r;
    return launch_geometry_lists(c, N, P, pos, graph_ptr, node_graph, pair_ptr, pair_code, g, nullptr, st, false);
}

// ---------------------------------------------------------------------------------------------
// End of a sampling step, one wave per graph, fused: ensemble mean (reference sampler.py:96-111),
// eq_transform (geometry.py:22-30), clip_norm + LD/DDPM update + NaN flag + centring (sampler.py:208-253)
// and -- on the new positions, still in LDS -- the per-row member counts of the NEXT step's edge lists
// (pair_count).  Replaces five launches (mean, eq_transform, update, advance, count) by one; the step
// counter is advanced by the next step's scan kernel.
// eq_transform: edge_inv is evaluated once per undirected pair, so s(i,j) == s(j,i) bit for bit and the
// reference's two scatter terms are equal: score_i = A_i + A_i with A_i = sum_{e in row i} u_e s_e (edge order).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void step_post_kernel(int kind, int N, int M, int PU,
                                                       const int32_t* __restrict__ graph_ptr,
                                                       const int32_t* __restrict__ pair_ptr,
                                                       const uint16_t* __restrict__ pair_code, tsd_edges out,
                                                       const float* __restrict__ inv_u, float clip, float clip_pos,
                                                       float* __restrict__ pos, tsd_sampler_state* __restrict__ ss,
                                                       int order_enc, int order_out, float cut2,
                                                       int32_t* __restrict__ cnt) {
    __shared__ float old_s[3 * (TSD_MAX_GRAPH_NODES + 1)];
    __shared__ float new_s[3 * (TSD_MAX_GRAPH_NODES + 1)];
    __shared__ int cnt_s[NLIST * (TSD_MAX_GRAPH_NODES + 1)];
    // everything a call may change comes from the device-resident state block (the captured graph is reused
    // by every call): the row of the coefficient / noise / trajectory arrays is the device-side step counter
    const size_t k_step = (size_t)ss->step;
    const float* __restrict__ coefs = ss->args.coefs + k_step * TSD_STEP_COEFS;
    const float* __restrict__ noise = ss->args.noises ? ss->args.noises + k_step * 3 * (size_t)N : nullptr;
    float* __restrict__ traj = ss->args.traj ? ss->args.traj + k_step * 3 * (size_t)N : nullptr;
    const uint64_t seed = ss->args.seed, ctr0 = ss->args.offset + k_step * (uint64_t)N;
    int32_t* status = &ss->flags;
    const int g = blockIdx.x;
    const int lo = graph_ptr[g], hi = graph_ptr[g + 1];
    const int n = hi - lo;
    const int lane = threadIdx.x;
    float c[TSD_STEP_COEFS];
#pragma unroll
    for (int k = 0; k < TSD_STEP_COEFS; ++k) c[k] = coefs[k];
    for (int t = lane; t < 3 * n; t += 64) old_s[t] = pos[3 * lo + t];
    __syncthreads();

    float sx = 0.f, sy = 0.f, sz = 0.f;
    bool bad = false;
    for (int il = lane; il < n; il += 64) {
        const int i = lo + il;
        const float px = old_s[3 * il], py = old_s[3 * il + 1], pz = old_s[3 * il + 2];
        float ax = 0.f, ay = 0.f, az = 0.f;
        const int e1 = out.row_ptr[i + 1];
        int e = out.row_ptr[i];
        // 4 edges per round: their index / distance loads and then their edge_inv gathers are independent,
        // only the adds keep the edge order (the serial chain was 1 us per edge)
        for (; e + 4 <= e1; e += 4) {
            int jl[4], u[4];
            float dd[4], sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                jl[q] = out.dst[e + q] - lo;
                u[q] = out.umap[e + q];
                dd[q] = out.dist[e + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = inv_u[u[q]];
                for (int m = 1; m < M; ++m) s = __fadd_rn(s, inv_u[(size_t)m * PU + u[q]]);
                sv[q] = s / (float)M;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float inv = 1.0f / dd[q];
                ax = __fadd_rn(ax, __fmul_rn(__fmul_rn(inv, px - old_s[3 * jl[q]]), sv[q]));
                ay = __fadd_rn(ay, __fmul_rn(__fmul_rn(inv, py - old_s[3 * jl[q] + 1]), sv[q]));
                az = __fadd_rn(az, __fmul_rn(__fmul_rn(inv, pz - old_s[3 * jl[q] + 2]), sv[q]));
            }
        }
        for (; e < e1; ++e) {
            const int jl = out.dst[e] - lo;
            const int u = out.umap[e];
            float s = inv_u[u];
            for (int m = 1; m < M; ++m) s = __fadd_rn(s, inv_u[(size_t)m * PU + u]);
            s = s / (float)M;
            const float inv = 1.0f / out.dist[e];
            ax = __fadd_rn(ax, __fmul_rn(__fmul_rn(inv, px - old_s[3 * jl]), s));
            ay = __fadd_rn(ay, __fmul_rn(__fmul_rn(inv, py - old_s[3 * jl + 1]), s));
            az = __fadd_rn(az, __fmul_rn(__fmul_rn(inv, pz - old_s[3 * jl + 2]), s));
        }
        float v[3] = {ax + ax, ay + ay, az + az};
        float p[3] = {px, py, pz};
        // clip_norm (sampler.py:265-268)
        const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2])));
        const float denom = norm > clip ? clip / norm : 1.0f;
        float zn[3];
        if (noise) {
            zn[0] = noise[3 * i]; zn[1] = noise[3 * i + 1]; zn[2] = noise[3 * i + 2];
        } else {
            philox_normal3(ctr0 + (uint64_t)i, seed, zn);  // sampler.py:213 randn_like, generated in place
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float eps = __fmul_rn(v[k], denom);
            const float nz = zn[k];
            float nx;
            if (kind == 0) {  // LD, sampler.py:238-244
                nx = __fadd_rn(__fadd_rn(p[k], __fmul_rn(c[0], eps) / c[1]), __fmul_rn(nz, c[2]));
            } else {  // DDPM, sampler.py:215-236
                const float e = -eps;
                const float pos_C = __fmul_rn(c[0], p[k]);
                const float pos0 = __fsub_rn(__fmul_rn(c[1], pos_C), __fmul_rn(c[2], e));
                const float mean = __fadd_rn(__fmul_rn(c[3], pos0), __fmul_rn(c[4], pos_C)) / c[5];
                nx = __fadd_rn(mean, __fmul_rn(c[6], nz)) / c[7];
            }
            bad |= (nx != nx);
            p[k] = nx;
        }
        new_s[3 * il] = p[0];
        new_s[3 * il + 1] = p[1];
        new_s[3 * il + 2] = p[2];
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
    const float cntf = (float)max(n, 1);
    const float mx = sx / cntf, my = sy / cntf, mz = sz / cntf;
    for (int il = lane; il < n; il += 64) {  // center_pos (sampler.py:260-262) + optional clamp
        float x = new_s[3 * il] - mx, y = new_s[3 * il + 1] - my, z = new_s[3 * il + 2] - mz;
        if (clip_pos >= 0.0f) {
            x = fminf(fmaxf(x, -clip_pos), clip_pos);
            y = fminf(fmaxf(y, -clip_pos), clip_pos);
            z = fminf(fmaxf(z, -clip_pos), clip_pos);
        }
        new_s[3 * il] = x;
        new_s[3 * il + 1] = y;
        new_s[3 * il + 2] = z;
        const int i = lo + il;
        pos[3 * i] = x;
        pos[3 * i + 1] = y;
        pos[3 * i + 2] = z;
        if (traj) {
            traj[3 * i] = x;
            traj[3 * i + 1] = y;
            traj[3 * i + 2] = z;
        }
    }
    __syncthreads();
    // member counts of the next step's lists on the new positions (same arithmetic as pair_count_kernel):
    // the graph's n(n-1) ordered pairs spread over the lanes, per-row counters in LDS
    const int pg0 = pair_ptr[lo];
    const int npairs = n * (n - 1);
    for (int t = lane; t < n * NLIST; t += 64) cnt_s[t] = 0;
    __syncthreads();
    for (int p = lane; p < npairs; p += 64) {
        const int il = p / (n - 1), k = p - il * (n - 1);
        const int jl = k + (k >= il ? 1 : 0);
        const PairEval r = eval_pair_xyz(new_s[3 * il], new_s[3 * il + 1], new_s[3 * il + 2], new_s[3 * jl],
                                         new_s[3 * jl + 1], new_s[3 * jl + 2], pair_code[pg0 + p], order_enc, order_out,
                                         cut2);
        const bool up = jl > il;
        if (r.in_enc) atomicAdd(&cnt_s[il * NLIST + 0], 1);
        if (r.in_out) atomicAdd(&cnt_s[il * NLIST + 1], 1);
        if (r.in_enc && up) atomicAdd(&cnt_s[il * NLIST + 2], 1);
        if (r.in_out && up) atomicAdd(&cnt_s[il * NLIST + 3], 1);
        if (r.needs_own_attr() && up) atomicAdd(&cnt_s[il * NLIST + 4], 1);
    }
    __syncthreads();
    for (int t = lane; t < n * NLIST; t += 64) {
        const int il = t / NLIST, q = t - il * NLIST;
        cnt[(size_t)q * (N + 1) + lo + il] = cnt_s[t];
    }
}

int launch_step_post(const tsd_model_cfg& c, int kind, int N, int G, int M, int P, const int32_t* graph_ptr,
                     const int32_t* pair_ptr, const uint16_t* pair_code, tsd_geometry g, const float* inv_u,
                     float clip, float clip_pos, float* pos, tsd_sampler_state* state, hipStream_t st) {
    if (G > 0) {
        hipLaunchKernelGGL(step_post_kernel, dim3(G), dim3(64), 0, st, kind, N, M, P / 2, graph_ptr, pair_ptr,
                           pair_code, g.out, inv_u, clip, clip_pos, pos, state, c.edge_order, c.pred_edge_order,
                           c.edge_cutoff * c.edge_cutoff, g.scratch);
        TSD_LAUNCH_CHECK("step_post");
    }
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// The whole step tail in ONE launch (round 3; replaces step_post + scan + pair_fill = 25 us and two kernel
// boundaries of a 450-us step at batch 100): one workgroup per graph does
//   1. ensemble mean + eq_transform + clip + LD/DDPM update + NaN flag + centring   (as step_post_kernel)
//   2. membership of the graph's ordered pairs on the NEW positions, per 64-pair chunk counts
//   3. the graph's totals -> ONE 8-byte record {tag, enc_u, out_u, diff_u} (16 bits each: a graph has at most
//      255*254/2 undirected pairs) stored write-through; the offsets of the graph in the five lists = the sum of
//      the records of the graphs before it (the directed lists hold exactly twice the undirected counts of a
//      graph: membership is symmetric), polled 64 records per sweep -- "the data is the flag" (cdna_hip_programming.md
//      Guideline 16, R2: relaxed agent-scope 8-byte granules, no fences, bounded spins)
//   4. stream compaction of the member pairs into the five lists of the NEXT step (as pair_fill_kernel; the pairs
//      of a graph in pair order ARE the reference's row-major edge order), row_ptr from the first pair of each row.
// Graph <-> workgroup by a ticket (atomicAdd), not blockIdx: every graph a workgroup waits for has been taken by a
// workgroup that is already running, whatever the dispatch order.  ticket / G + 1 is the launch's epoch within the
// run (tsd_sampler_plan_run zeroes the ticket and the records): epoch 1 = the list-only launch on the initial
// positions (kind < 0), epoch k + 2 = step k, whose row of the coefficient / noise / trajectory tables it selects;
// the record tag is the epoch's low 16 bits (a record is rewritten every launch, so a stale tag never matches).
// Bit-identical lists and positions to the three-kernel path (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------
constexpr int TAIL_MAX_N = 64;      // larger graphs: three-kernel path (the eq_transform terms of a graph live in LDS)
constexpr int TAIL_MAX_G = 4096;    // every workgroup sums all records before its own: ceil(g / 64) sweeps
constexpr int TAIL_CTL_INTS = 4;    // ticket + padding ahead of the records in geo.scratch
constexpr unsigned TAIL_SPIN_LIMIT = 400000u;  // ~0.3 s of polling: then TSD_STATUS_INTERNAL instead of a hang

typedef __attribute__((address_space(1))) unsigned long long gu64;

template <int NT>
__global__ __launch_bounds__(NT) void step_tail_kernel(int kind, int N, int G, int M, int P,
                                                       const int32_t* __restrict__ graph_ptr,
                                                       const int32_t* __restrict__ pair_ptr,
                                                       const uint16_t* __restrict__ pair_code, tsd_geometry gq,
                                                       const float* __restrict__ inv_u, float clip, float clip_pos,
                                                       float* __restrict__ pos, tsd_sampler_state* __restrict__ ss,
                                                       int order_enc, int order_out, float cut2,
                                                       int max_pairs /* LDS capacity in ordered pairs */) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    float* old_s = reinterpret_cast<float*>(tsm);          // [3 * TAIL_MAX_N]
    float* new_s = old_s + 3 * TAIL_MAX_N;                 // [3 * TAIL_MAX_N]
    int* ccnt = reinterpret_cast<int*>(new_s + 3 * TAIL_MAX_N);  // [NLIST][64] per-chunk member counts -> exclusive prefix
    int* misc = ccnt + NLIST * 64;                         // [16]
    float* terms = reinterpret_cast<float*>(misc + 16);    // [3 * max_n (max_n - 1)] u_e s_e of every ordered pair
    uint16_t* code_s = reinterpret_cast<uint16_t*>(terms + 3 * max_pairs);  // [max_n (max_n - 1)] pair codes

    // The kernel is a chain of dependent memory round trips (everything it reads was written by the kernels just
    // before it: ~2 us each): keep the chain short -- [ticket || graph offsets of the expected graph] -> [positions,
    // pair offset, coefficients, noise] -> [pair codes, undirected out index of every pair] -> [edge_inv gathers].
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PU = P / 2;
    unsigned* ticket = reinterpret_cast<unsigned*>(gq.scratch);
    gu64* rec = (gu64*)(gq.scratch + TAIL_CTL_INTS);
    if (tid == 0) misc[0] = (int)atomicAdd(ticket, 1u);
    // (workgroups are dispatched in blockIdx order in practice: the ticket then equals it and these loads are used)
    const int g_spec = (int)blockIdx.x;
    int lo = graph_ptr[g_spec], hi = graph_ptr[g_spec + 1];
    const tsd_run_args ra = ss->args;
    __syncthreads();
    const unsigned tk = (unsigned)misc[0];
    const int g = (int)(tk % (unsigned)G);
    const unsigned epoch = tk / (unsigned)G + 1u;
    const unsigned long long tag = epoch & 0xffffu;
    if (g != g_spec) {
        lo = graph_ptr[g];
        hi = graph_ptr[g + 1];
    }
    const int n = hi - lo;
    const int nm1 = max(n - 1, 1);
    const int npairs = n * (n - 1);
    // row of ordered pair p = p / (n - 1) without the ~40-instruction runtime division: exact for p (n - 1) < 2^32
    const unsigned div_magic = nm1 > 1 ? 0xFFFFFFFFu / (unsigned)nm1 + 1u : 0u;
    auto row_of = [&](int p) { return nm1 > 1 ? (int)__umulhi((unsigned)p, div_magic) : p; };
    int32_t* status = &ss->flags;
    const size_t k_step = kind >= 0 ? (size_t)(epoch - 2u) : 0;
    const float* __restrict__ coefs = ra.coefs + k_step * TSD_STEP_COEFS;
    const float* __restrict__ noise = ra.noises ? ra.noises + k_step * 3 * (size_t)N : nullptr;
    float* __restrict__ traj = ra.traj ? ra.traj + k_step * 3 * (size_t)N : nullptr;

    // second round trip: positions, the graph's pair offset, this lane's noise row, the step's coefficients
    const int pg0 = pair_ptr[lo];
    float c[TSD_STEP_COEFS];
    float zn[3] = {0.f, 0.f, 0.f};
    if (kind >= 0 && wave == 0) {
#pragma unroll
        for (int k = 0; k < TSD_STEP_COEFS; ++k) c[k] = coefs[k];
        if (noise && lane < n) {
            zn[0] = noise[3 * (lo + lane)]; zn[1] = noise[3 * (lo + lane) + 1]; zn[2] = noise[3 * (lo + lane) + 2];
        }
    }
    for (int t = tid; t < 3 * n; t += NT) old_s[t] = pos[3 * lo + t];
    __syncthreads();

    // third + fourth round trip, PB pairs per thread in flight: pair code (-> LDS, the membership tests below read
    // it twice) and -- update launches -- the pair's undirected out edge, then its edge_inv; u_e s_e -> LDS for
    // EVERY ordered pair (+0 for non-edges: the ordered row sums below are unchanged by adding +0)
    constexpr int PB = NT >= 256 ? 4 : 8;
    for (int p0 = 0; p0 < npairs; p0 += PB * NT) {
        int code[PB], uu[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int p = min(p0 + q * NT + tid, npairs - 1);
            code[q] = pair_code[pg0 + p];
            uu[q] = -1;
            if (kind >= 0) {
                const int il = row_of(p), k = p - il * nm1;
                const int jl = k + (k >= il ? 1 : 0);
                const int p_up = jl > il ? p : jl * nm1 + il - 1;  // the pair with src < dst
                uu[q] = gq.pair2u[(size_t)P + pg0 + p_up];
            }
        }
        float sv[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            sv[q] = 0.0f;
            if (uu[q] >= 0) {
                float s = inv_u[uu[q]];
                for (int m = 1; m < M; ++m) s = __fadd_rn(s, inv_u[(size_t)m * PU + uu[q]]);
                sv[q] = s / (float)M;
            }
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int p = p0 + q * NT + tid;
            if (p < npairs) {
                code_s[p] = (uint16_t)code[q];
                if (kind >= 0) {
                    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
                    if (uu[q] >= 0) {
                        const int il = row_of(p), k = p - il * nm1;
                        const int jl = k + (k >= il ? 1 : 0);
                        const float dx = old_s[3 * il] - old_s[3 * jl], dy = old_s[3 * il + 1] - old_s[3 * jl + 1],
                                    dz = old_s[3 * il + 2] - old_s[3 * jl + 2];
                        // edge_length as the list build computed it (eval_pair_xyz), 1 / d as eq_transform
                        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                        const float inv = 1.0f / sqrtf(d2);
                        t0 = __fmul_rn(__fmul_rn(inv, dx), sv[q]);
                        t1 = __fmul_rn(__fmul_rn(inv, dy), sv[q]);
                        t2 = __fmul_rn(__fmul_rn(inv, dz), sv[q]);
                    }
                    terms[3 * p] = t0;
                    terms[3 * p + 1] = t1;
                    terms[3 * p + 2] = t2;
                }
            }
        }
    }
    __syncthreads();
    if (kind >= 0) {
        const uint64_t seed = ra.seed, ctr0 = ra.offset + k_step * (uint64_t)N;
        if (wave == 0) {  // n <= 64: one atom per lane of wave 0 (same lane <-> atom map and sums as step_post_kernel)
            float sx = 0.f, sy = 0.f, sz = 0.f;
            float p[3] = {0.f, 0.f, 0.f};
            bool bad = false;
            const int il = lane;
            if (il < n) {
                const int i = lo + il;
                p[0] = old_s[3 * il]; p[1] = old_s[3 * il + 1]; p[2] = old_s[3 * il + 2];
                float ax = 0.f, ay = 0.f, az = 0.f;
                const float* tr = terms + 3 * il * (n - 1);
                for (int r = 0; r < n - 1; ++r) {  // the row's pairs in order = the row's edges in order
                    ax = __fadd_rn(ax, tr[3 * r]);
                    ay = __fadd_rn(ay, tr[3 * r + 1]);
                    az = __fadd_rn(az, tr[3 * r + 2]);
                }
                const float v[3] = {ax + ax, ay + ay, az + az};
                // clip_norm (sampler.py:265-268)
                const float norm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2])));
                const float denom = norm > clip ? clip / norm : 1.0f;
                if (!noise) philox_normal3(ctr0 + (uint64_t)i, seed, zn);  // sampler.py:213 randn_like, generated in place
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float eps = __fmul_rn(v[k], denom);
                    const float nz = zn[k];
                    float nx;
                    if (kind == 0) {  // LD, sampler.py:238-244
                        nx = __fadd_rn(__fadd_rn(p[k], __fmul_rn(c[0], eps) / c[1]), __fmul_rn(nz, c[2]));
                    } else {  // DDPM, sampler.py:215-236
                        const float e = -eps;
                        const float pos_C = __fmul_rn(c[0], p[k]);
                        const float pos0 = __fsub_rn(__fmul_rn(c[1], pos_C), __fmul_rn(c[2], e));
                        const float mean = __fadd_rn(__fmul_rn(c[3], pos0), __fmul_rn(c[4], pos_C)) / c[5];
                        nx = __fadd_rn(mean, __fmul_rn(c[6], nz)) / c[7];
                    }
                    bad |= (nx != nx);
                    p[k] = nx;
                }
                sx = p[0]; sy = p[1]; sz = p[2];
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                sx += __shfl_xor(sx, off);
                sy += __shfl_xor(sy, off);
                sz += __shfl_xor(sz, off);
            }
            if (__ballot(bad) != 0ull && lane == 0) atomicOr(status, TSD_STATUS_NAN);
            const float cntf = (float)max(n, 1);
            const float mx = sx / cntf, my = sy / cntf, mz = sz / cntf;
            if (il < n) {  // center_pos (sampler.py:260-262) + optional clamp
                float x = p[0] - mx, y = p[1] - my, z = p[2] - mz;
                if (clip_pos >= 0.0f) {
                    x = fminf(fmaxf(x, -clip_pos), clip_pos);
                    y = fminf(fmaxf(y, -clip_pos), clip_pos);
                    z = fminf(fmaxf(z, -clip_pos), clip_pos);
                }
                new_s[3 * il] = x;
                new_s[3 * il + 1] = y;
                new_s[3 * il + 2] = z;
                const int i = lo + il;
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
        __syncthreads();
    } else {
        __syncthreads();
        for (int t = tid; t < 3 * n; t += NT) new_s[t] = old_s[t];
        __syncthreads();
    }

    // ---- membership on the new positions: per-chunk counts of the five lists (chunk = 64 consecutive ordered pairs)
    const int nchunks = (npairs + 63) >> 6;  // <= 63 for n <= 64
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (int c = wave; c < nchunks; c += NW) {
        const int p = c * 64 + lane;
        bool m[NLIST] = {false, false, false, false, false};
        if (p < npairs) {
            const int il = row_of(p), k = p - il * nm1;
            const int jl = k + (k >= il ? 1 : 0);
            const PairEval r = eval_pair_xyz(new_s[3 * il], new_s[3 * il + 1], new_s[3 * il + 2], new_s[3 * jl],
                                             new_s[3 * jl + 1], new_s[3 * jl + 2], code_s[p], order_enc, order_out, cut2);
            const bool up = jl > il;
            m[0] = r.in_enc;
            m[1] = r.in_out;
            m[2] = r.in_enc && up;
            m[3] = r.in_out && up;
            m[4] = r.needs_own_attr() && up;
        }
#pragma unroll
        for (int q = 0; q < NLIST; ++q) {
            const int cq = __popcll(__ballot(m[q]));
            if (lane == 0) ccnt[q * 64 + c] = cq;
        }
    }
    __syncthreads();
    if (wave == 0) {
        // exclusive prefix over the chunks (one chunk per lane), the graph's totals, its record, the look-back
        int tot[NLIST];
#pragma unroll
        for (int q = 0; q < NLIST; ++q) {
            const int x = lane < nchunks ? ccnt[q * 64 + lane] : 0;
            int v = x;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(v, off);
                if (lane >= off) v += u;
            }
            ccnt[q * 64 + lane] = v - x;
            tot[q] = __shfl(v, 63);
        }
        if (lane == 0) {
            const unsigned long long r = tag | ((unsigned long long)tot[2] << 16) | ((unsigned long long)tot[3] << 32) |
                                         ((unsigned long long)tot[4] << 48);
            __hip_atomic_store(rec + g, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int base[3] = {0, 0, 0};
        bool gave_up = false;
        constexpr int LB = 4;  // records per lane per sweep (their loads are in flight together): 256 graphs
        for (int g0 = 0; g0 < g && !gave_up; g0 += 64 * LB) {
            unsigned long long r[LB];
#pragma unroll
            for (int q = 0; q < LB; ++q) r[q] = tag;
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    const int gg = g0 + q * 64 + lane;
                    // (a record that has matched is not loaded again)
                    if (gg < g && (spins == 0 || (r[q] & 0xffffull) != tag))
                        r[q] = __hip_atomic_load(rec + gg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int q = 0; q < LB; ++q) ok &= (r[q] & 0xffffull) == tag;
                if (__all(ok)) break;
                if (spins > TAIL_SPIN_LIMIT) {  // (wave-uniform: the exit condition is a ballot)
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            int a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
            for (int q = 0; q < LB; ++q) {
                if (g0 + q * 64 + lane < g) {
                    a0 += (int)((r[q] >> 16) & 0xffffull);
                    a1 += (int)((r[q] >> 32) & 0xffffull);
                    a2 += (int)((r[q] >> 48) & 0xffffull);
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                a0 += __shfl_xor(a0, off);
                a1 += __shfl_xor(a1, off);
                a2 += __shfl_xor(a2, off);
            }
            base[0] += a0;
            base[1] += a1;
            base[2] += a2;
        }
        if (lane == 0) {
            if (gave_up) atomicOr(status, TSD_STATUS_INTERNAL);
            misc[1] = 2 * base[0];  // enc
            misc[2] = 2 * base[1];  // out
            misc[3] = base[0];      // enc_u
            misc[4] = base[1];      // out_u
            misc[5] = base[2];      // diff_u
            if (g == G - 1) {  // the last graph closes the lists
                const int end[NLIST] = {misc[1] + tot[0], misc[2] + tot[1], misc[3] + tot[2], misc[4] + tot[3],
                                        misc[5] + tot[4]};
                tsd_edges* lists[NLIST] = {&gq.enc, &gq.out, &gq.enc_u, &gq.out_u, &gq.diff_u};
#pragma unroll
                for (int q = 0; q < NLIST; ++q) {
                    lists[q]->row_ptr[N] = end[q];
                    *lists[q]->count = end[q];
                }
                ss->step = (int32_t)epoch - 2;  // row of the step tables this launch used (-1: the list-only launch)
            }
        }
    }
    __syncthreads();

    // ---- fill: the member pairs of chunk c land at base + (members in earlier chunks) + (members among lower lanes)
    const int gbase[NLIST] = {misc[1], misc[2], misc[3], misc[4], misc[5]};
    if (npairs == 0) {
        if (tid < n) {  // (n == 1: a row without pairs)
            gq.enc.row_ptr[lo + tid] = gbase[0];
            gq.out.row_ptr[lo + tid] = gbase[1];
            gq.enc_u.row_ptr[lo + tid] = gbase[2];
            gq.out_u.row_ptr[lo + tid] = gbase[3];
            gq.diff_u.row_ptr[lo + tid] = gbase[4];
        }
        return;
    }
    for (int c = wave; c < nchunks; c += NW) {
        const int p = c * 64 + lane;
        const bool valid = p < npairs;
        PairEval r;
        r.in_enc = r.in_out = false;
        r.d = 0.f;
        r.tr_enc = r.tp_enc = r.tr_out = r.tp_out = 0;
        int il = 0, k = 0, jl = 0;
        bool up = false;
        if (valid) {
            il = row_of(p);
            k = p - il * nm1;
            jl = k + (k >= il ? 1 : 0);
            r = eval_pair_xyz(new_s[3 * il], new_s[3 * il + 1], new_s[3 * il + 2], new_s[3 * jl], new_s[3 * jl + 1],
                              new_s[3 * jl + 2], code_s[p], order_enc, order_out, cut2);
            up = jl > il;
        }
        const bool m[NLIST] = {r.in_enc, r.in_out, r.in_enc && up, r.in_out && up, valid && up && r.needs_own_attr()};
        int at[NLIST];
#pragma unroll
        for (int q = 0; q < NLIST; ++q)
            at[q] = gbase[q] + ccnt[q * 64 + c] + __popcll(__ballot(m[q]) & lower);
        const int i = lo + il, j = lo + jl;
        if (valid && k == 0) {  // the first pair of a row: its would-be positions are the row's offsets
            gq.enc.row_ptr[i] = at[0];
            gq.out.row_ptr[i] = at[1];
            gq.enc_u.row_ptr[i] = at[2];
            gq.out_u.row_ptr[i] = at[3];
            gq.diff_u.row_ptr[i] = at[4];
        }
        if (m[0]) {
            const int e = at[0];
            gq.enc.src[e] = i;
            gq.enc.dst[e] = j;
            gq.enc.dist[e] = r.d;
            gq.enc.type_r[e] = (uint8_t)r.tr_enc;
            gq.enc.type_p[e] = (uint8_t)r.tp_enc;
            gq.enc.pair_id[e] = pg0 + p;
        }
        if (m[1]) {
            const int e = at[1];
            gq.out.src[e] = i;
            gq.out.dst[e] = j;
            gq.out.dist[e] = r.d;
            gq.out.type_r[e] = (uint8_t)r.tr_out;
            gq.out.type_p[e] = (uint8_t)r.tp_out;
            gq.out.pair_id[e] = pg0 + p;
        }
        if (m[2]) {
            const int e = at[2];
            gq.enc_u.src[e] = i;
            gq.enc_u.dst[e] = j;
            gq.enc_u.dist[e] = r.d;
            gq.enc_u.type_r[e] = (uint8_t)r.tr_enc;
            gq.enc_u.type_p[e] = (uint8_t)r.tp_enc;
            gq.enc_u.pair_id[e] = pg0 + p;
        }
        if (m[3]) {
            const int e = at[3];
            gq.out_u.src[e] = i;
            gq.out_u.dst[e] = j;
            gq.out_u.dist[e] = r.d;
            gq.out_u.type_r[e] = (uint8_t)r.tr_out;
            gq.out_u.type_p[e] = (uint8_t)r.tp_out;
            gq.out_u.pair_id[e] = pg0 + p;
            gq.attr_row[e] = m[4] ? PU + at[4] : at[2];  // own embedding, or the enc_u edge's row
        }
        if (m[4]) {
            const int e = at[4];
            gq.diff_u.dist[e] = r.d;
            gq.diff_u.type_r[e] = (uint8_t)r.tr_out;
            gq.diff_u.type_p[e] = (uint8_t)r.tp_out;
        }
        if (valid) {
            gq.pair2out[pg0 + p] = m[1] ? at[1] : -1;
            if (up) {
                gq.pair2u[pg0 + p] = m[2] ? at[2] : -1;
                gq.pair2u[(size_t)P + pg0 + p] = m[3] ? at[3] : -1;
            }
        }
    }
}

bool step_tail_supported(int N, int G, int max_n) {
    return G >= 1 && N >= 1 && max_n >= 1 && max_n <= TAIL_MAX_N && G <= TAIL_MAX_G &&
           (size_t)TAIL_CTL_INTS + 2 * (size_t)G <= geometry_scratch_ints(N, 0);
}

// zero the ticket and the records: once per run, ahead of the list-only launch (outside the captured step)
int launch_step_tail_reset(int G, tsd_geometry g, hipStream_t st) {
    TSD_HIP(hipMemsetAsync(g.scratch, 0, ((size_t)TAIL_CTL_INTS + 2 * (size_t)G) * sizeof(int32_t), st));
    return TSD_OK;
}

// kind < 0: lists of `pos` only (first step of a run); kind 0 / 1: the LD / DDPM step tail, then the lists
int launch_step_tail(const tsd_model_cfg& c, int kind, int N, int G, int M, int P, int max_n,
                     const int32_t* graph_ptr, const int32_t* pair_ptr, const uint16_t* pair_code, tsd_geometry g,
                     const float* inv_u, float clip, float clip_pos, float* pos, tsd_sampler_state* state,
                     hipStream_t st) {
    if (!step_tail_supported(N, G, max_n)) {
        set_error("internal: fused step tail launched for an unsupported batch (N=%d G=%d max_n=%d)", N, G, max_n);
        return TSD_ERR_INVALID;
    }
    const size_t fixed = (size_t)(6 * TAIL_MAX_N) * 4 + (size_t)(NLIST * 64 + 16) * 4;
    auto lds_for = [&](int mn) { return fixed + (size_t)14 * mn * (mn - 1) + 16; };  // 3 floats + 1 u16 per ordered pair
    const size_t lds = lds_for(max_n);
    const int max_pairs = max_n * (max_n - 1);
    const float cut2 = c.edge_cutoff * c.edge_cutoff;
    if (max_n <= 12) {  // (<= 132 ordered pairs: one wave; else four waves share the pair loops)
        hipLaunchKernelGGL(step_tail_kernel<64>, dim3(G), dim3(64), lds, st, kind, N, G, M, P, graph_ptr, pair_ptr,
                           pair_code, g, inv_u, clip, clip_pos, pos, state, c.edge_order, c.pred_edge_order, cut2,
                           max_pairs);
    } else {
        static DeviceOnce once;  // (the attribute is set once per device: ask for the largest graph's worth)
        int r = allow_lds(step_tail_kernel<256>, lds_for(TAIL_MAX_N), once);
        if (r) return r;
        hipLaunchKernelGGL(step_tail_kernel<256>, dim3(G), dim3(256), lds, st, kind, N, G, M, P, graph_ptr, pair_ptr,
                           pair_code, g, inv_u, clip, clip_pos, pos, state, c.edge_order, c.pred_edge_order, cut2,
                           max_pairs);
    }
    TSD_LAUNCH_CHECK("step_tail");
    return TSD_OK;
}

// per-call inputs of the captured step -> the device-resident state block (flags untouched); step = -1: the
// scan kernel of every step, the first included, advances it
__global__ void set_run_args_kernel(tsd_sampler_state* ss, tsd_run_args a) {
    ss->args = a;
    ss->step = -1;
}
int launch_set_run_args(tsd_sampler_state* ss, const tsd_run_args& a, hipStream_t st) {
    hipLaunchKernelGGL(set_run_args_kernel, dim3(1), dim3(1), 0, st, ss, a);
    TSD_LAUNCH_CHECK("set_run_args");
    return TSD_OK;
}

__global__ void philox_normal_kernel(uint64_t seed, uint64_t offset, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float z[3];
    philox_normal3(offset + (uint64_t)i, seed, z);
    out[3 * i] = z[0];
    out[3 * i + 1] = z[1];
    out[3 * i + 2] = z[2];
}
int launch_philox_normal(uint64_t seed, uint64_t offset, int64_t n, float* out, hipStream_t st) {
    if (n <= 0) return TSD_OK;
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, offset, n, out);
    TSD_LAUNCH_CHECK("philox_normal");
    return TSD_OK;
}

}  // namespace tsd

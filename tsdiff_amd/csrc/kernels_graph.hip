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
            if (hopR[idx] == 255) {
                bool f = false;
                for (int k = 0; k < n && !f; ++k) f = (hopR[i * n + k] == level - 1) && (hopR[k * n + j] == 1);
                if (f) hopR[idx] = (unsigned char)level;
            }
            if (hopP[idx] == 255) {
                bool f = false;
                for (int k = 0; k < n && !f; ++k) f = (hopP[i * n + k] == level - 1) && (hopP[k * n + j] == 1);
                if (f) hopP[idx] = (unsigned char)level;
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
        static bool done = false;
        if (!done || lds > 48 * 1024) {
            TSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hop_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            done = true;
        }
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

__device__ __forceinline__ PairEval eval_pair(const float* __restrict__ pos, int i, int j, int code,
                                              int order_enc, int order_out, float cut2) {
    PairEval r;
    // models/geometry.py:18-19  (pos[row] - pos[col]).norm()
    const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1],
                dz = pos[3 * i + 2] - pos[3 * j + 2];
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

// exclusive scan of NLIST int arrays of length N (+ total at [N]); single workgroup of 1024 threads
__global__ __launch_bounds__(1024) void scan_kernel(int N, const int32_t* __restrict__ cnt, ScanOut o) {
    __shared__ int sm[NLIST][1024];
    const int t = threadIdx.x;
    const int per = (N + 1023) / 1024;
    const int beg = min(N, t * per), end = min(N, beg + per);
    int x[NLIST];
#pragma unroll
    for (int q = 0; q < NLIST; ++q) {
        int a = 0;
        for (int i = beg; i < end; ++i) a += cnt[(size_t)q * (N + 1) + i];
        x[q] = a;
        sm[q][t] = a;
    }
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        int v[NLIST];
#pragma unroll
        for (int q = 0; q < NLIST; ++q) v[q] = (t >= off) ? sm[q][t - off] : 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NLIST; ++q) sm[q][t] += v[q];
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NLIST; ++q) {
        int r = sm[q][t] - x[q];  // exclusive prefix of this thread's chunk
        for (int i = beg; i < end; ++i) {
            o.row_ptr[q][i] = r;
            r += cnt[(size_t)q * (N + 1) + i];
        }
        if (t == 1023) {
            o.row_ptr[q][N] = sm[q][1023];
            *o.total[q] = sm[q][1023];
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
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
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

size_t geometry_scratch_ints(int N, int P) {
    (void)P;
    return (size_t)NLIST * (N + 1);
}

int launch_geometry(const tsd_model_cfg& c, int N, int G, int P, const float* pos, const int32_t* graph_ptr,
                    const int32_t* node_graph, const int32_t* pair_ptr, const uint16_t* pair_code,
                    tsd_geometry g, hipStream_t st) {
    (void)G;
    int32_t* cnt = g.scratch;
    const float cut2 = c.edge_cutoff * c.edge_cutoff;
    const int blocks = (N + 3) / 4;
    if (N > 0) {
        hipLaunchKernelGGL(pair_count_kernel, dim3(blocks), dim3(256), 0, st, N, pos, graph_ptr, node_graph,
                           pair_ptr, pair_code, c.edge_order, c.pred_edge_order, cut2, cnt);
        TSD_LAUNCH_CHECK("pair_count");
    }
    ScanOut so;
    tsd_edges* lists[NLIST] = {&g.enc, &g.out, &g.enc_u, &g.out_u, &g.diff_u};
    for (int q = 0; q < NLIST; ++q) {
        so.row_ptr[q] = lists[q]->row_ptr;
        so.total[q] = lists[q]->count;
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, N, cnt, so);
    TSD_LAUNCH_CHECK("scan");
    if (N > 0) {
        hipLaunchKernelGGL(pair_fill_kernel, dim3(blocks), dim3(256), 0, st, N, pos, graph_ptr, node_graph,
                           pair_ptr, pair_code, c.edge_order, c.pred_edge_order, cut2, g, P);
        TSD_LAUNCH_CHECK("pair_fill");
    }
    if (P > 0) {
        hipLaunchKernelGGL(edge_umap_kernel, dim3((P + 255) / 256), dim3(256), 0, st, g, graph_ptr, node_graph,
                           pair_ptr, P);
        TSD_LAUNCH_CHECK("edge_umap");
    }
    return TSD_OK;
}

}  // namespace tsd

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

__global__ __launch_bounds__(256) void pair_count_kernel(int N, const float* __restrict__ pos,
                                                         const int32_t* __restrict__ graph_ptr,
                                                         const int32_t* __restrict__ node_graph,
                                                         const int32_t* __restrict__ pair_ptr,
                                                         const uint16_t* __restrict__ pair_code,
                                                         int order_enc, int order_out, float cut2,
                                                         int32_t* __restrict__ cnt_enc,
                                                         int32_t* __restrict__ cnt_out,
                                                         int32_t* __restrict__ cnt_diff) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int lo = graph_ptr[node_graph[i]];
    const int il = i - lo;
    const int p0 = pair_ptr[i], np = pair_ptr[i + 1] - p0;
    int ce = 0, co = 0, cd = 0;
    for (int k0 = 0; k0 < np; k0 += 64) {
        const int k = k0 + lane;
        bool me = false, mo = false, md = false;
        if (k < np) {
            const int j = lo + k + (k >= il ? 1 : 0);
            const PairEval r = eval_pair(pos, i, j, pair_code[p0 + k], order_enc, order_out, cut2);
            me = r.in_enc;
            mo = r.in_out;
            md = r.needs_own_attr();
        }
        ce += __popcll(__ballot(me));
        co += __popcll(__ballot(mo));
        cd += __popcll(__ballot(md));
    }
    if (lane == 0) {
        cnt_enc[i] = ce;
        cnt_out[i] = co;
        cnt_diff[i] = cd;
    }
}

// exclusive scan of three int arrays of length N (+ total at [N]); single workgroup of 1024 threads
__global__ __launch_bounds__(1024) void scan3_kernel(int N, const int32_t* __restrict__ a_in,
                                                     const int32_t* __restrict__ b_in,
                                                     const int32_t* __restrict__ c_in,
                                                     int32_t* __restrict__ a_out, int32_t* __restrict__ b_out,
                                                     int32_t* __restrict__ c_out,
                                                     int32_t* __restrict__ a_total, int32_t* __restrict__ b_total,
                                                     int32_t* __restrict__ c_total) {
    __shared__ int sa[1024], sb[1024], sc[1024];
    const int t = threadIdx.x;
    const int per = (N + 1023) / 1024;
    const int beg = min(N, t * per), end = min(N, beg + per);
    int xa = 0, xb = 0, xc = 0;
    for (int i = beg; i < end; ++i) {
        xa += a_in[i];
        xb += b_in[i];
        xc += c_in[i];
    }
    sa[t] = xa;
    sb[t] = xb;
    sc[t] = xc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        int va = 0, vb = 0, vc = 0;
        if (t >= off) {
            va = sa[t - off];
            vb = sb[t - off];
            vc = sc[t - off];
        }
        __syncthreads();
        sa[t] += va;
        sb[t] += vb;
        sc[t] += vc;
        __syncthreads();
    }
    int ra = sa[t] - xa, rb = sb[t] - xb, rc = sc[t] - xc;  // exclusive prefix of this thread's chunk
    for (int i = beg; i < end; ++i) {
        const int va = a_in[i], vb = b_in[i], vc = c_in[i];
        a_out[i] = ra;
        b_out[i] = rb;
        c_out[i] = rc;
        ra += va;
        rb += vb;
        rc += vc;
    }
    if (t == 1023) {
        a_out[N] = sa[1023];
        b_out[N] = sb[1023];
        c_out[N] = sc[1023];
        *a_total = sa[1023];
        *b_total = sb[1023];
        *c_total = sc[1023];
    }
}

__global__ __launch_bounds__(256) void pair_fill_kernel(int N, const float* __restrict__ pos,
                                                        const int32_t* __restrict__ graph_ptr,
                                                        const int32_t* __restrict__ node_graph,
                                                        const int32_t* __restrict__ pair_ptr,
                                                        const uint16_t* __restrict__ pair_code,
                                                        int order_enc, int order_out, float cut2,
                                                        tsd_edges enc, tsd_edges out, tsd_edges diff,
                                                        int32_t* __restrict__ attr_row, int P,
                                                        int32_t* __restrict__ pair2out) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int lo = graph_ptr[node_graph[i]];
    const int il = i - lo;
    const int p0 = pair_ptr[i], np = pair_ptr[i + 1] - p0;
    int be = enc.row_ptr[i], bo = out.row_ptr[i], bd = diff.row_ptr[i];
    const unsigned long long lower = (1ull << lane) - 1ull;
    for (int k0 = 0; k0 < np; k0 += 64) {
        const int k = k0 + lane;
        PairEval r;
        r.in_enc = r.in_out = false;
        int j = 0;
        if (k < np) {
            j = lo + k + (k >= il ? 1 : 0);
            r = eval_pair(pos, i, j, pair_code[p0 + k], order_enc, order_out, cut2);
        }
        const bool own = (k < np) && r.needs_own_attr();
        const unsigned long long me = __ballot(r.in_enc), mo = __ballot(r.in_out), md = __ballot(own);
        const int ie = be + __popcll(me & lower), io = bo + __popcll(mo & lower), id = bd + __popcll(md & lower);
        if (r.in_enc) {
            enc.src[ie] = i;
            enc.dst[ie] = j;
            enc.dist[ie] = r.d;
            enc.type_r[ie] = (uint8_t)r.tr_enc;
            enc.type_p[ie] = (uint8_t)r.tp_enc;
            enc.pair_id[ie] = p0 + k;
        }
        if (r.in_out) {
            out.src[io] = i;
            out.dst[io] = j;
            out.dist[io] = r.d;
            out.type_r[io] = (uint8_t)r.tr_out;
            out.type_p[io] = (uint8_t)r.tp_out;
            out.pair_id[io] = p0 + k;
            attr_row[io] = own ? P + id : ie;  // row of the [2P,H] edge-attribute matrix
        }
        if (own) {
            diff.dist[id] = r.d;
            diff.type_r[id] = (uint8_t)r.tr_out;
            diff.type_p[id] = (uint8_t)r.tp_out;
        }
        if (k < np) pair2out[p0 + k] = r.in_out ? io : -1;
        be += __popcll(me);
        bo += __popcll(mo);
        bd += __popcll(md);
    }
}

size_t geometry_scratch_ints(int N, int P) {
    (void)P;
    return (size_t)3 * (N + 1);
}

int launch_geometry(const tsd_model_cfg& c, int N, int G, int P, const float* pos, const int32_t* graph_ptr,
                    const int32_t* node_graph, const int32_t* pair_ptr, const uint16_t* pair_code,
                    tsd_edges enc, tsd_edges out, tsd_edges diff, int32_t* attr_row, int32_t* pair2out,
                    int32_t* scratch, hipStream_t st) {
    (void)G;
    int32_t* cnt_enc = scratch;
    int32_t* cnt_out = scratch + (N + 1);
    int32_t* cnt_diff = scratch + 2 * (N + 1);
    const float cut2 = c.edge_cutoff * c.edge_cutoff;
    const int blocks = (N + 3) / 4;
    if (N > 0) {
        hipLaunchKernelGGL(pair_count_kernel, dim3(blocks), dim3(256), 0, st, N, pos, graph_ptr, node_graph,
                           pair_ptr, pair_code, c.edge_order, c.pred_edge_order, cut2, cnt_enc, cnt_out, cnt_diff);
        TSD_LAUNCH_CHECK("pair_count");
    }
    hipLaunchKernelGGL(scan3_kernel, dim3(1), dim3(1024), 0, st, N, cnt_enc, cnt_out, cnt_diff, enc.row_ptr,
                       out.row_ptr, diff.row_ptr, enc.count, out.count, diff.count);
    TSD_LAUNCH_CHECK("scan3");
    if (N > 0) {
        hipLaunchKernelGGL(pair_fill_kernel, dim3(blocks), dim3(256), 0, st, N, pos, graph_ptr, node_graph,
                           pair_ptr, pair_code, c.edge_order, c.pred_edge_order, cut2, enc, out, diff, attr_row, P,
                           pair2out);
        TSD_LAUNCH_CHECK("pair_fill");
    }
    return TSD_OK;
}

}  // namespace tsd

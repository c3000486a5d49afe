// kernels_combo.hip -- one launch per interaction block: the node-side chain of block l and the filter
// GEMMs of block l+1 share a grid ("horizontal fusion").
//
// At batch-100 sizes the node chain (aggregate -> lin2 -> ssp -> lin -> +h -> next lin1 on N = 1600
// rows) is a short dependent sequence that can occupy ~100 of the 256 CUs, while the CFConv filters
// of the NEXT block (reference models/encoder/schnet.py:94-99) do not depend on it.  Two streams joined
// by events were measured slower under hipGraph replay (1.07 vs 0.92 ms/step) and only 6 % faster
// eagerly, so the overlap is done inside ONE kernel: workgroups [0, node_tiles) take the node role
// (they are dispatched first: critical path), the rest take the filter role and fill the idle CUs.
// Roles never communicate; the kernel boundary orders block l's filter before block l's aggregation.
//
// Both roles run with 2H threads = H/32 waves, 32 output columns per wave.
#include "train_internal.hpp"
#include "typed_tile.hpp"

// Phase timeline of the per-block launch (variant builds only: tools/build_variant.sh trace "-DTSD_TRACE";
// tools/trace_combo.py reads it).  32 u64 slots per workgroup: [0..7] s_memtime at phase boundaries (wave 0), [8..15] / [16..23] every wave's end of its
// first / second GEMM, slot 31 = HW_ID | XCC_ID << 32 | role << 40.
#ifdef TSD_TRACE
static unsigned long long* g_tsd_trace_host = nullptr;  // variant builds only
extern "C" int tsd_debug_trace(void* buf) {
    g_tsd_trace_host = (unsigned long long*)buf;
    return 0;
}
int g_tsd_debug_prec = 0;  // variant builds only: tsd_interaction_block runs the split-f16 roles (w = the f16-plane arena)
extern "C" int tsd_debug_prec(int prec) {
    g_tsd_debug_prec = prec;
    return 0;
}
#define TSD_TRACE_AT(slot)                                                                                      \
    do {                                                                                                        \
        if (trace_buf && threadIdx.x == 0)                                                                      \
            trace_buf[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define TSD_TRACE_WAVE(base)                                                                                    \
    do {                                                                                                        \
        if (trace_buf && (threadIdx.x & 63) == 0)                                                               \
            trace_buf[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 32 + (base) + (threadIdx.x >> 6)] = \
                __builtin_amdgcn_s_memtime();                                                                   \
    } while (0)
#define TSD_TRACE_ID(role)                                                                                      \
    do {                                                                                                        \
        if (trace_buf && threadIdx.x == 0)                                                                      \
            trace_buf[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 32 + 31] =                        \
                (unsigned long long)__builtin_amdgcn_s_getreg(63492) |                                          \
                ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15) << 32) | ((unsigned long long)(role) << 40); \
    } while (0)
#define TSD_TRACE_REAL(slot)                                                                                    \
    do {                                                                                                        \
        if (trace_buf && threadIdx.x == 0)                                                                      \
            trace_buf[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 32 + (slot)] = wall_clock64();    \
    } while (0)
#define TSD_TRACE_ARG , unsigned long long* trace_buf
#define TSD_TRACE_PASS , trace_buf
#define TSD_TRACE_NULL , nullptr
#else
#define TSD_TRACE_AT(slot)
#define TSD_TRACE_WAVE(base)
#define TSD_TRACE_ID(role)
#define TSD_TRACE_REAL(slot)
#define TSD_TRACE_ARG
#define TSD_TRACE_PASS
#define TSD_TRACE_NULL
#endif

namespace tsd {

#ifndef TSD_FILTER_TRANS
#define TSD_FILTER_TRANS 1  // transposed accumulators in the split-f16 tile roles (0: the round-4 layout; A/B builds)
#endif
constexpr int T = TSD_EDGE_TILE;   // 32 edges per filter tile
constexpr int TN = TSD_NODE_TILE;  // 16 nodes per node tile
#ifndef TSD_AGG_PREFETCH
#define TSD_AGG_PREFETCH 1
#endif

struct ComboNode {
    int mode;  // 0: aggregate + update (+ next lin1), 1: x1_out = lin1(h) only, -1: no node role
    int N;
    const int32_t *row_ptr, *dst, *umap;
    const float* Wf;  // this block's filters on the undirected list [Eu, H]
    const float* x1_in;
    const float* h_in;  // residual input (block 0 reads the pos-independent node embedding z directly)
    float* h;           // residual output
    float* x1_out;
    const float *lin2_w, *lin2_b, *lin_w, *lin_b, *lin1_next_w;
    int32_t* ready;  // != NULL (last launch of a small forward): publish h write-through, then ready[tile] = 1
};

// Filter work is a flat queue of items g = layer * tiles_per_layer + tile over ALL layers (it depends on the
// geometry only, not on the node states); a launch takes the range [g_begin, g_begin + tiles).
struct ComboFilter {
    int tiles;            // filter work items of this launch (0: no filter role)
    int g_begin;          // first item
    int tiles_per_layer;  // ceil(capacity / tile rows)
    int layer0;           // layer of item 0 for the WEIGHTS (0 in the forward; the layer itself for single-layer calls)
    const float* Wl0;     // packed weights of interaction block 0; block l at + l * layer_stride
    size_t layer_stride, o_nn0_w, o_nn0_b, o_nn2_w, o_nn2_b;
    float conv_cutoff;
    int smooth;
    tsd_edges e;
    const float* edge_attr;
    float* wf;            // filter slot 0; the filters of item layer l go to slot l % wf_slots
    size_t wf_layer_stride;
    int wf_slots;
};

template <int V>
struct VRow {  // V consecutive channels of a row held by one lane (a register tuple the asm loads can name)
    typedef float type __attribute__((ext_vector_type(V)));
    static __device__ __forceinline__ float get(const type& x, int v) { return x[v]; }
};
template <>
struct VRow<1> {
    typedef float type;
    static __device__ __forceinline__ float get(const type& x, int) { return x; }
};

// -------------------------------------------------------------------------------------------------
// The aggregation of one tile of TN nodes: buf[r] = sum_{e in row n0 + r} x[dst e] * Wf[umap e] (edge order, product
// rounded then added: bit-identical to a sequential scatter_add), rows past N zero.  Shared by the node role of the
// per-block launch (x = x1, the CFConv message; schnet.py:101-107) and by the backward node chain of the
// training step (x = d loss / d agg: the same gather is the adjoint w.r.t. x1).  SAVE: rows also go to `save` [N,H].
// -------------------------------------------------------------------------------------------------
// SPLIT: the tile goes to LDS as the two f16 planes of split16.hpp ([TN][H + 8] each, at `buf`) instead of fp32 rows;
// *amax collects max |sum| for the range check.
// SC1: the rows are loaded with `sc1` (from L2, never this CU's L1): the one-launch forward gathers rows that other CUs
// wrote during the launch into buffers this CU may have read an older version of.
// TR: rows of the tile that are aggregated (default TN; the one-launch forward's node workgroups take 8: rows TR .. TN - 1
// of the LDS tile are then not written here).
template <int H, bool SAVE, int NW = 2 * H / 64 /* waves of the workgroup */, int U = 8 /* edges in flight per wave */,
          bool SPLIT = false, bool SC1 = false, int TR = TN, int PROWS = TN /* rows of the LDS tile (planes) */>
__device__ __forceinline__ void aggregate_tile(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ dst,
                                               const int32_t* __restrict__ umap, const float* __restrict__ Wf,
                                               const float* __restrict__ x, int N, int n0, float* buf,
                                               float* __restrict__ save, float* amax = nullptr) {
    constexpr int LDA = H + 4;
    constexpr int LDH = ldh_of(H);
    const Planes pl = planes_at(buf, PROWS, LDH);
    float amx = 0.0f;
    // (the asm loads below take these through "s" operands: wave-uniform as the compiler can see it, split16.hpp)
    dst = reinterpret_cast<const int32_t*>(uniform_ptr(dst));
    umap = reinterpret_cast<const int32_t*>(uniform_ptr(umap));
    Wf = reinterpret_cast<const float*>(uniform_ptr(Wf));
    x = reinterpret_cast<const float*>(uniform_ptr(x));
    constexpr int RPW = TR / NW;  // rows aggregated per wave
    constexpr int V = H / 64;     // channels per lane during aggregation
    static_assert(TR % NW == 0 && TR <= PROWS, "");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // The RPW rows of this wave are consecutive, so their edges are ONE contiguous CSR range
    // [rp[first], rp[first + RPW]): one load of the RPW + 1 row offsets, one coalesced load of the indices per
    // 64 edges, then batches of U edges in flight across the row boundaries -- 2 + ceil(edges / U) dependent
    // memory round trips per wave (the per-row form with its 4 / 1-edge tail loops took ~14: the aggregation
    // was 22 of the 36 us of a node tile at batch 100, and the node chain is the critical path of the launch).
    // Sums stay per row, in edge order, product rounded then added: bit-identical to a sequential scatter_add.
    const int first = n0 + wave * RPW;
    const int rpv = row_ptr[min(first + min(lane, RPW), N)];  // lanes 0..RPW: offsets of this wave's rows
    const int E0 = __builtin_amdgcn_readlane(rpv, 0), E1 = __builtin_amdgcn_readlane(rpv, RPW);
    int rr = 0;                                        // current row (wave-uniform)
    int row_end = __builtin_amdgcn_readlane(rpv, 1);   // end of the current row's edges
    float s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0.0f;
    auto flush = [&]() {  // row rr is complete: its sums go to the LDS tile, the next row starts
        if constexpr (SPLIT) {
            const int off = (wave * RPW + rr) * LDH + lane * V;
            if constexpr (V == 4) {
                const f32x4 sv = {s[0], s[1], s[2], s[3]};
                planes_store4(pl, off, sv, amx);
            } else if constexpr (V == 2) {
                planes_store2(pl, off, s[0], s[1], amx);
            } else {
                planes_store1(pl, off, s[0], amx);
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if constexpr (!SPLIT) buf[(wave * RPW + rr) * LDA + lane * V + v] = s[v];
            if constexpr (SAVE) {
                if (first + rr < N) save[(size_t)(first + rr) * H + lane * V + v] = s[v];
            }
            s[v] = 0.0f;
        }
        ++rr;
        row_end = __builtin_amdgcn_readlane(rpv, min(rr + 1, RPW));
    };
    // The gather runs beside the MFMA streams of the filter tiles on the same SIMDs, and an fp32 MFMA holds the
    // SIMD's vector issue port for its 64 cycles: every VALU instruction of this wave waits for an MFMA boundary
    // (traced at batch 100: the aggregation takes 6 us alone on its CU, 21 us beside a filter tile -- the same
    // with the filter's weight loads removed, 7 us with its MFMAs removed).  So the loop is written to need almost
    // no VALU work: the edge indices come by SCALAR loads (8 consecutive edges per s_load_dwordx8), the row bases
    // are SALU arithmetic, the row loads take an SGPR base + one lane-offset VGPR (inline asm: hipcc builds a
    // 64-bit VGPR address per load), which leaves the products and sums.
    // U edges in flight per wave (2 x U row loads of H floats)
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    typedef typename VRow<V>::type vrow;
    const unsigned lane_b = (unsigned)lane * (V * 4u);  // this lane's byte offset inside a row
#if TSD_AGG_PREFETCH
    // the indices of batch k + 1 are requested while the rows of batch k are in flight: one dependent scalar round trip
    // less per batch of a chain of (2 + edges / U) round trips
    i32x8 jd, ud, jn, un;
    if (E0 < E1) {
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(jn) : "s"(dst + E0) : "memory");
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(un) : "s"(umap + E0) : "memory");
    }
#endif
    for (int e = E0; e < E1; e += U) {
#if TSD_AGG_PREFETCH
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(jn), "+s"(un)::"memory");
        jd = jn;
        ud = un;
#else
        i32x8 jd, ud;  // dst / umap of edges e .. e+7 (the lists carry 8 spare entries: tsdiff_hip.h)
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(jd) : "s"(dst + e) : "memory");
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(ud) : "s"(umap + e) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(jd), "+s"(ud)::"memory");
#endif
        vrow wv[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // slots past this wave's range hold other rows' (or no) edges: they re-read slot 0 and are not added
            const bool live = e + u < E1;
            const int we = live ? ud[u] : ud[0], j = live ? jd[u] : jd[0];
            const float* wrow = Wf + (size_t)we * H;
            const float* xrow = x + (size_t)j * H;
            if constexpr (V == 4 && SC1) {
                asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(wv[u]) : "v"(lane_b), "s"(wrow) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(xv[u]) : "v"(lane_b), "s"(xrow) : "memory");
            } else if constexpr (V == 2 && SC1) {
                asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(wv[u]) : "v"(lane_b), "s"(wrow) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(xv[u]) : "v"(lane_b), "s"(xrow) : "memory");
            } else if constexpr (SC1) {
                asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(wv[u]) : "v"(lane_b), "s"(wrow) : "memory");
                asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(xv[u]) : "v"(lane_b), "s"(xrow) : "memory");
            } else if constexpr (V == 4) {
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wv[u]) : "v"(lane_b), "s"(wrow) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xv[u]) : "v"(lane_b), "s"(xrow) : "memory");
            } else if constexpr (V == 2) {
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(wv[u]) : "v"(lane_b), "s"(wrow) : "memory");
                asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(xv[u]) : "v"(lane_b), "s"(xrow) : "memory");
            } else {
                asm volatile("global_load_dword %0, %1, %2" : "=v"(wv[u]) : "v"(lane_b), "s"(wrow) : "memory");
                asm volatile("global_load_dword %0, %1, %2" : "=v"(xv[u]) : "v"(lane_b), "s"(xrow) : "memory");
            }
        }
#if TSD_AGG_PREFETCH
        if (e + U < E1) {
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(jn) : "s"(dst + e + U) : "memory");
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(un) : "s"(umap + e + U) : "memory");
        }
#endif
        // one wait for the batch, naming every destination (the consumers below depend on this statement)
        if constexpr (U == 8)
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]), "+v"(wv[4]), "+v"(wv[5]), "+v"(wv[6]),
                           "+v"(wv[7]), "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]),
                           "+v"(xv[6]), "+v"(xv[7])::"memory");
        else
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]), "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]),
                           "+v"(xv[3])::"memory");
        static_assert(U == 8 || U == 4, "the wait statement names U + U registers");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u < E1) {
                while (e + u >= row_end) flush();  // (also steps over rows without edges)
#pragma unroll
                for (int v = 0; v < V; ++v)
                    s[v] = __fadd_rn(s[v], __fmul_rn(VRow<V>::get(xv[u], v), VRow<V>::get(wv[u], v)));
            }
        }
    }
    while (rr < RPW) flush();  // the last row, and rows past it without edges (or past the last node): zeros
    if constexpr (SPLIT) {
        if (amax) *amax = fmaxf(*amax, amx);
    }
}

// -------------------------------------------------------------------------------------------------
// The same aggregation for the node workgroups of the one-launch forward (H = 256, split planes, rows stored by other
// workgroups of the launch), wrapped around the wait for the neighbour tiles (`wait_for_x`, which ends with a barrier).
// Traced (tools/trace_mega.py): every dependent round trip of the gather is 1.5-2 us there -- the rows were stored
// write-through by other CUs and come from beyond the XCD's L2 -- and aggregate_tile makes five or six of them behind
// the wait (row offsets, indices, four batches of 8 edges = 16 row loads).  Here:
//   1. BEFORE the wait (the filters of the block are complete): row offsets, indices, and the filter rows of the wave's
//      first 8 edges -- all of it depends on the geometry only.
//   2. behind the wait: the columns a tile gathers x at are the atoms of the graphs its own atoms belong to, rows
//      [xbase, xbase + nloc) of x, a few dozen at molecule sizes, each wanted by up to TN rows of the tile: they are
//      copied to LDS (`xs`) by LDS-DMA, one round trip, no registers.  The per-edge loads are then the filter rows
//      alone -- half the bytes through the CU's L2 port -- and
//   3. with the 64 row registers of a wave all on filter rows the remaining edges take one round trip per 16.
// Two round trips behind the wait for up to 24 edges per wave, three for up to 40.  Sums as in aggregate_tile: per
// row, in edge order, product rounded then added -- bit-identical.  Every wave of the workgroup calls it.
// (Variants that did not fit 128 VGPRs without an occupancy cap, i.e. two workgroups per CU: 16 edges requested before
// the wait; x rows and 16 filter rows in ONE round trip behind it; the next batch's indices requested early.)
// -------------------------------------------------------------------------------------------------
constexpr int XS_ROWS = 60;  // rows of x the LDS copy holds (a 16-row tile between two 23-atom graphs: 16 + 2 * 22)
typedef int xl_i32x8 __attribute__((ext_vector_type(8)));
template <int H, int NW, int TR, class WaitFn>
__device__ __forceinline__ void xl_gather(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ dst,
                                          const int32_t* __restrict__ umap, const float* __restrict__ Wf,
                                          const float* __restrict__ x, int xbase, int nloc, float* xs, int N, int n0,
                                          float* buf, float& amax, WaitFn&& wait_for_x) {
    static_assert(H == 256, "one 16-byte load per lane and row");
    constexpr int V = 4, U = 8, LDH = ldh_of(H), RPW = TR / NW;
    static_assert(TR % NW == 0 && TR <= TN, "");
    const Planes pl = planes_at(buf, TN, LDH);
    dst = reinterpret_cast<const int32_t*>(uniform_ptr(dst));
    umap = reinterpret_cast<const int32_t*>(uniform_ptr(umap));
    Wf = reinterpret_cast<const float*>(uniform_ptr(Wf));
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lane_b = (unsigned)lane * (V * 4u);
    const int first = n0 + wave * RPW;
    static_assert(RPW == 2, "the row offsets of a wave are three scalars");
    int E0, Em, E1;  // edges of the wave's first row: [E0, Em), of its second: [Em, E1)
    {
        const int rpv = row_ptr[min(first + min(lane, RPW), N)];  // lanes 0 .. RPW: row offsets of this wave's rows
        E0 = __builtin_amdgcn_readlane(rpv, 0), Em = __builtin_amdgcn_readlane(rpv, 1), E1 = __builtin_amdgcn_readlane(rpv, 2);
    }
    // step 1 (before the wait for the x rows): the filter rows of the wave's first 8 edges -- indices and row loads
    // depend on the geometry only.  (8, not 16: with 64 row registers held across the wait the kernel does not fit the
    // 128 VGPRs of two workgroups per CU.)  The index lists carry 8 spare entries (tsdiff_hip.h); slots past the range
    // re-read slot 0 and are not added.
    f32x4 w0[U];
    xl_i32x8 ja;  // their x columns
    if (E0 < E1) {
        xl_i32x8 ua;
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(ua) : "s"(umap + E0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ua)::"memory");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int we = E0 + u < E1 ? ua[u] : ua[0];
            asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(w0[u]) : "v"(lane_b), "s"(Wf + (size_t)we * H) : "memory");
        }
    }
    wait_for_x();  // (ends with a barrier)
    if (E0 < E1) asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(ja) : "s"(dst + E0) : "memory");
    {   // step 2: the rows of x -> LDS by LDS-DMA (one 1-KiB row per wave instruction, no registers: the row registers
        // of step 1 are in flight); `sc1` as every load of rows another CU stored during this launch.  The wait also
        // covers the filter rows of step 1.
        static_assert(XS_ROWS <= 8 * NW, "eight rows per wave");
        const char* xg = reinterpret_cast<const char*>(x + (size_t)xbase * H) + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave + i * NW;
            if (r < nloc)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xg + (size_t)r * (H * 4)),
                                                 (__attribute__((address_space(3))) void*)(xs + (size_t)r * H), 16, 0, 16 /* sc1 */);
        }
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(w0[0]), "+v"(w0[1]), "+v"(w0[2]), "+v"(w0[3]), "+v"(w0[4]), "+v"(w0[5]), "+v"(w0[6]), "+v"(w0[7])::"memory");
    }
    int rr = 0;
    int row_end = Em;
    float s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0.0f;
    auto flush = [&]() {
        const f32x4 sv = {s[0], s[1], s[2], s[3]};
        planes_store4(pl, (wave * RPW + rr) * LDH + lane * V, sv, amax);
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] = 0.0f;
        ++rr;
        row_end = E1;  // (rr >= 1: the second row, or past it)
    };
    auto consume = [&](const f32x4 (&w)[U], const xl_i32x8& jd, int e) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e + u < E1) {
                while (e + u >= row_end) flush();  // (also steps over rows without edges)
                const f32x4 xr = *reinterpret_cast<const f32x4*>(xs + (size_t)(jd[u] - xbase) * H + lane * V);
#pragma unroll
                for (int v = 0; v < V; ++v) s[v] = __fadd_rn(s[v], __fmul_rn(xr[v], w[u][v]));
            }
        }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ja)::"memory");
    __syncthreads();  // the LDS copy of x is complete
    if (E0 < E1) consume(w0, ja, E0);
    // step 3: the remaining edges, 16 per round trip (two groups of 8: a group is only requested when its first edge is
    // inside the range)
    for (int e = E0 + U; e < E1; e += 2 * U) {
        const bool two = e + U < E1;
        xl_i32x8 jd0, ud0, jd1, ud1;
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(jd0) : "s"(dst + e) : "memory");
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(ud0) : "s"(umap + e) : "memory");
        if (two) {
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(jd1) : "s"(dst + e + U) : "memory");
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(ud1) : "s"(umap + e + U) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(jd0), "+s"(ud0), "+s"(jd1), "+s"(ud1)::"memory");
        f32x4 wa[U], wb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int we = e + u < E1 ? ud0[u] : ud0[0];
            asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(wa[u]) : "v"(lane_b), "s"(Wf + (size_t)we * H) : "memory");
        }
        if (two) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int we = e + U + u < E1 ? ud1[u] : ud1[0];
                asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(wb[u]) : "v"(lane_b), "s"(Wf + (size_t)we * H) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2]), "+v"(wa[3]), "+v"(wa[4]), "+v"(wa[5]), "+v"(wa[6]), "+v"(wa[7]),
                       "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]), "+v"(wb[3]), "+v"(wb[4]), "+v"(wb[5]), "+v"(wb[6]), "+v"(wb[7])::"memory");
        consume(wa, jd0, e);
        if (two) consume(wb, jd1, e + U);
    }
    while (rr < RPW) flush();  // the last row, and rows past it without edges (or past the last node): zeros
    // (no low-side site check here -- split16.hpp site_close: the one-launch kernel has no register to spare; the
    // attribute rows that feed these filters are checked where the filter tiles convert them)
}

// -------------------------------------------------------------------------------------------------
// node role: agg[i] = sum_{e in row i} x1[dst e] * Wf[umap e]  (edge order, product rounded then added:
// bit-identical to a sequential scatter_add), then the three dense layers of tsd_node_update.
// reference schnet.py:101-107 (message/aggregate), :103 (lin2), :123-127, :223-224
// -------------------------------------------------------------------------------------------------
template <int H, bool SAVE>
__device__ __forceinline__ void node_role(const ComboNode& a, int tile, float* smem, const NodeSave& ns TSD_TRACE_ARG) {
    constexpr int LDA = H + 4;
    constexpr int NT = 2 * H;
    constexpr int CB16 = 2;  // 16-wide column blocks per wave
    constexpr int C4 = H / 4;
    float* buf = smem;

    const int n0 = tile * TN;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int col0 = wave * 32;
    const int nrows = min(TN, a.N - n0);
    f32x4 acc[CB16];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    TSD_TRACE_ID(1);
    TSD_TRACE_AT(0);
    // biases and the residual input of this lane's outputs: requested first, they arrive under the aggregation
    float b_lin2[CB16], b_lin[CB16], h_res[CB16][4];
    if (a.mode == 0) {
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
            b_lin2[cb] = a.lin2_b[col];
            b_lin[cb] = a.lin_b[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = q * 4 + r;
                h_res[cb][r] = row < nrows ? a.h_in[(size_t)(n0 + row) * H + col] : 0.0f;
            }
        }
    }
    if (a.mode == 0) {
        aggregate_tile<H, SAVE>(a.row_ptr, a.dst, a.umap, a.Wf, a.x1_in, a.N, n0, buf, SAVE ? ns.agg : nullptr);
        TSD_TRACE_WAVE(16);
        __syncthreads();
        TSD_TRACE_AT(1);

#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) acc[cb] = zero4;
        gemm_tile16<CB16, H>(buf, LDA, a.lin2_w, H, col0, acc);
        TSD_TRACE_WAVE(8);
        TSD_TRACE_AT(2);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
            const float b = b_lin2[cb];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[cb][r] + b, sv = sspf(v);
                buf[(q * 4 + r) * LDA + col] = sv;
                if constexpr (SAVE) {
                    if (q * 4 + r < nrows) {
                        ns.x2[(size_t)(n0 + q * 4 + r) * H + col] = v;
                        ns.xs[(size_t)(n0 + q * 4 + r) * H + col] = sv;
                    }
                }
            }
        }
        __syncthreads();
        TSD_TRACE_AT(3);

#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) acc[cb] = zero4;
        gemm_tile16<CB16, H>(buf, LDA, a.lin_w, H, col0, acc);
        TSD_TRACE_AT(4);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
            const float b = b_lin[cb];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = q * 4 + r;
                float hn = 0.0f;
                if (row < nrows) {
                    hn = h_res[cb][r] + (acc[cb][r] + b);
                    if (a.ready)  // write-through (sc1): the pair tiles of this launch read the row from other CUs
                        __hip_atomic_store(a.h + (size_t)(n0 + row) * H + col, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        a.h[(size_t)(n0 + row) * H + col] = hn;
                }
                buf[row * LDA + col] = hn;
            }
        }
        if (a.ready) {  // Guideline 16 R1: every storing wave drains, the workgroup meets, ONE lane raises the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.ready + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.lin1_next_w == nullptr) return;
        __syncthreads();
    } else {
        for (int idx = tid; idx < TN * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            f32x4 v = zero4;
            if (r < nrows) v = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)(n0 + r) * H + c4 * 4);
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = v;
        }
        __syncthreads();
    }

    TSD_TRACE_AT(5);
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) acc[cb] = zero4;
    gemm_tile16<CB16, H>(buf, LDA, a.lin1_next_w, H, col0, acc);
    TSD_TRACE_AT(6);
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) {
        const int col = col0 + cb * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            if (row < nrows) a.x1_out[(size_t)(n0 + row) * H + col] = acc[cb][r];
        }
    }
    TSD_TRACE_AT(7);
}

// -------------------------------------------------------------------------------------------------
// filter role: Wf[e] = nn2(ssp(nn0(edge_attr[e]))) * C(e) for one tile of 32 undirected edges,
// H/32 waves x 32 columns (the 256-thread stand-alone form is filter_gen_kernel in kernels_mlp.hip)
// -------------------------------------------------------------------------------------------------
template <int H, bool SAVE>
__device__ __forceinline__ void filter_role(const ComboFilter& f, int item, float* smem, const FilterSave& fsv TSD_TRACE_ARG) {
    const int g = f.g_begin + item;
    const int lrel = g / f.tiles_per_layer, tile = g - lrel * f.tiles_per_layer;
    const float* Wb = f.Wl0 + (size_t)(f.layer0 + lrel) * f.layer_stride;
    const float *nn0_w = Wb + f.o_nn0_w, *nn0_b = Wb + f.o_nn0_b, *nn2_w = Wb + f.o_nn2_w, *nn2_b = Wb + f.o_nn2_b;
    float* out = f.wf + (size_t)(lrel % f.wf_slots) * f.wf_layer_stride;
    constexpr int LDA = H + 4;
    constexpr int NT = 2 * H;
    constexpr int C4 = H / 4;
    float* buf = smem;
    float* s_c = smem + T * LDA;

    const int E = *f.e.count;
    const int e0 = tile * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32;
    const int nrows = min(T, E - e0);
    TSD_TRACE_ID(2);
    TSD_TRACE_AT(0);

    // the first k-blocks of both GEMMs' weights are requested ahead of the barrier / epilogue that precedes them
    // (they depend on nothing computed here): the first MFMA of a GEMM does not wait an L2 round trip
    BRing<1, ring_depth<1>()> rg;
    gemm_ring_start<1, H>(rg, nn0_w, H, col0);
    if (tid < T) s_c[tid] = tid < nrows ? cutoff_weight(f.e.dist[e0 + tid], f.conv_cutoff, f.smooth) : 0.0f;
    {   // edge_attr tile -> LDS with every load of a thread in flight together (rows past the end clamped)
        constexpr int NIT = T * C4 / NT;
        static_assert(T * C4 % NT == 0, "tile / block mismatch");
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            v[it] = *reinterpret_cast<const f32x4*>(f.edge_attr + (size_t)(e0 + min(r, nrows - 1)) * H + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = r < nrows ? v[it] : z;
        }
    }
    __syncthreads();
    TSD_TRACE_AT(1);

    f32x16 acc[1][1];
    zero_acc(acc);
    gemm_ring_run<1, 1, H>(rg, buf, LDA, acc);
    gemm_ring_start<1, H>(rg, nn2_w, H, col0);
    TSD_TRACE_WAVE(8);
    TSD_TRACE_AT(2);
    __syncthreads();
    {
        const int col = col0 + l31;
        const float b = nn0_b[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            const float v = acc[0][0][r] + b, sv = sspf(v);
            buf[row * LDA + col] = sv;
            if constexpr (SAVE) {  // the training step keeps every block's filters and activations: slot = block
                if (row < nrows) {
                    const size_t o = (size_t)lrel * f.wf_layer_stride + (size_t)(e0 + row) * H + col;
                    fsv.f0[o] = v;
                    fsv.fs[o] = sv;
                }
            }
        }
    }
    __syncthreads();
    TSD_TRACE_AT(3);

    zero_acc(acc);
    gemm_ring_run<1, 1, H>(rg, buf, LDA, acc);
    TSD_TRACE_WAVE(16);
    TSD_TRACE_AT(4);
    __syncthreads();
    {
        const int col = col0 + l31;
        const float b = nn2_b[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            buf[row * LDA + col] = (acc[0][0][r] + b) * s_c[row];
        }
    }
    __syncthreads();
    TSD_TRACE_AT(5);
    for (int idx = tid; idx < nrows * C4; idx += NT) {
        const int r = idx / C4, c4 = idx % C4;
        store_stream16(out + (size_t)(e0 + r) * H + c4 * 4, *reinterpret_cast<const f32x4*>(buf + r * LDA + c4 * 4));
    }
    TSD_TRACE_AT(6);
}

// "pre" role: the half of the pair MLP's first layer that does not depend on the node states,
//   pre[e] = edge_attr_out[e] . W0[:, H:2H]^T + b0                        (common.py:226-229, condensenc.py:236)
// for a tile of 32 undirected out edges.  It rides in the LAST block launch, whose filter slot is empty (the node
// chain of the last block occupies ~100 of the 256 CUs), and halves the first GEMM of pair_output_kernel.

template <int H>
__device__ __forceinline__ void pre_role(const ComboPre& q, int tile, float* smem) {
    constexpr int LDA = H + 4, NT = 2 * H, C4 = H / 4;
    float* buf = smem;
    int* s_row = reinterpret_cast<int*>(smem + T * LDA);
    const int E = *q.e.count;
    const int e0 = tile * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32;
    const int nrows = min(T, E - e0);
    if (tid < T) s_row[tid] = q.attr_row ? q.attr_row[e0 + min(tid, nrows - 1)] : e0 + min(tid, nrows - 1);
    __syncthreads();
    for (int idx = tid; idx < T * C4; idx += NT) {
        const int r = idx / C4, c4 = idx % C4;
        *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) =
            *reinterpret_cast<const f32x4*>(q.edge_attr + (size_t)s_row[r] * H + c4 * 4);
    }
    __syncthreads();
    f32x16 acc[1][1];
    zero_acc(acc);
    gemm_tile<1, 1, H>(buf, LDA, q.w0b, H, col0, acc);
    const int col = col0 + l31;
    const float b = q.b0[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        if (row < nrows) q.out[(size_t)(e0 + row) * H + col] = acc[0][0][r] + b;
    }
}

// "pair" role: the pair MLP of one tile of 32 undirected out edges inside the forward's LAST block launch
// (ComboPre, common.hpp): pre = edge_attr_out . W0[:, H:]^T + b0 while the node chain of the last block is still
// running, then -- once the node tiles that own the tile's atoms have published h -- h_i * h_j . W0[:, :H]^T on top,
// swish, the H -> H/2 layer, swish, the dot with w2: the arithmetic of pre_role + pair_output_kernel in the same
// order (bit-identical), without the round trip of `pre` through memory and without the pair_output launch.
#ifndef TSD_PAIR_SLEEP
#define TSD_PAIR_SLEEP 48  // x64 cycles between two polls of a waiting pair tile (~1.3 us)
#endif
#ifndef TSD_PAIR_DEFER
#define TSD_PAIR_DEFER 1
#endif
constexpr unsigned PAIR_SPIN_LIMIT = 2000000u;  // ~1 s of polling: then TSD_STATUS_INTERNAL instead of a hang

// defer_pre: the node-independent GEMM runs AFTER the wait -- for the pair tiles that share a CU with a node tile (an
// fp32 MFMA stream beside it slows the node chain's aggregation 3x, and the node chain is what everyone waits for)
template <int H>
__device__ __forceinline__ void pair_role(const ComboPre& q, int tile, int node_tiles, float* smem, bool defer_pre) {
    constexpr int LDA = H + 4, NT = 2 * H, C4 = H / 4, NW = H / 64;
    float* buf = smem;
    float* s_red = smem + T * LDA;  // [NW][T]
    int* s_src = reinterpret_cast<int*>(s_red + NW * T);
    int* s_dst = s_src + T;
    int* s_row = s_dst + T;
    const int E = *q.e.count;
    const int e0 = tile * T;
    if (e0 >= E) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = wave * 32, col = col0 + l31;
    const int nrows = min(T, E - e0);
    if (tid < T) {
        const int ee = e0 + min(tid, nrows - 1);
        s_src[tid] = q.e.src[ee];
        s_dst[tid] = q.e.dst[ee];
        s_row[tid] = q.attr_row ? q.attr_row[ee] : ee;
    }
    __syncthreads();
    f32x16 acc[1][1];
    auto pre_gemm = [&]() {
        for (int idx = tid; idx < T * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) =
                *reinterpret_cast<const f32x4*>(q.edge_attr + (size_t)s_row[r] * H + c4 * 4);
        }
        __syncthreads();
        zero_acc(acc);
        gemm_tile<1, 1, H>(buf, LDA, q.w0b, H, col0, acc);
        const float b = q.b0[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = acc_row(r, hi) < nrows ? acc[0][0][r] + b : 0.0f;
    };
    if (!defer_pre) pre_gemm();
    // wait for the node tiles that hold this tile's atoms (min .. max node id over both end points)
    if (wave == 0) {
        int lo = lane < T ? min(s_src[lane], s_dst[lane]) : 0x7fffffff;
        int hn = lane < T ? max(s_src[lane], s_dst[lane]) : -1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off));
            hn = max(hn, __shfl_xor(hn, off));
        }
        const int t_lo = lo / q.ready_div, t_hi = hn / q.ready_div;
        bool gave_up = false;
        for (int t0 = t_lo; t0 <= t_hi && !gave_up; t0 += 64) {
            const int t = t0 + lane;
            const bool need = t <= t_hi && t < node_tiles;
            for (unsigned spins = 0;; ++spins) {
                const int f = need ? __hip_atomic_load(q.ready + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
                if (__all(f >= q.ready_target)) break;
                // (a launch in which another wait has already given up unwinds quickly: see mega_wait_failed)
                if (spins > PAIR_SPIN_LIMIT ||
                    ((spins & 1023u) == 1023u &&
                     (__hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & TSD_STATUS_INTERNAL))) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(TSD_PAIR_SLEEP);
            }
        }
        if (gave_up && lane == 0) atomicOr(q.status, TSD_STATUS_INTERNAL);
    }
    __syncthreads();  // (also: every wave is done reading buf)
    if (defer_pre) {
        pre_gemm();
        __syncthreads();
    }
    {   // h_src * h_dst -> LDS.  The rows were written by other CUs during this launch: sc1 loads (L2, never this
        // CU's L1, which may hold the rows' previous contents from a node tile's residual read)
        constexpr int NIT = T * C4 / NT;
        static_assert(T * C4 % NT == 0, "tile / block mismatch");
        f32x4 hs[NIT], hd[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            const float* ps = q.h + (size_t)s_src[r] * H + c4 * 4;
            const float* pd = q.h + (size_t)s_dst[r] * H + c4 * 4;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hs[it]) : "v"(ps) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hd[it]) : "v"(pd) : "memory");
        }
        static_assert(NIT == 4, "the wait statement names 2 x 4 registers");
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(hs[0]), "+v"(hs[1]), "+v"(hs[2]), "+v"(hs[3]), "+v"(hd[0]), "+v"(hd[1]), "+v"(hd[2]), "+v"(hd[3])
                     :: "memory");
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = r < nrows ? hs[it] * hd[it] : z;
        }
    }
    __syncthreads();
    gemm_tile<1, 1, H>(buf, LDA, q.w0a, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[acc_row(r, hi) * LDA + col] = swishf(acc[0][0][r]);
    __syncthreads();
    if (wave < NW) {  // H -> H/2 and the final dot: the first H/64 waves
        f32x16 a2[1][1];
        const int c2 = wave * 32 + l31;
        zero_acc(a2);
        gemm_tile<1, 1, H>(buf, LDA, q.w1, H / 2, wave * 32, a2);
        const float b = q.b1[c2], w2 = q.w2[c2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float g = a2[0][0][r] + b, sg = swishf(g);
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
        q.edge_inv[e0 + tid] = v + q.b2[0];
    }
}


// -------------------------------------------------------------------------------------------------
// The three roles of the per-block launch on the f16 MFMA pipes (PREC_H2, split16.hpp): the same dataflow as
// node_role / filter_role / pair_role above -- same inputs and outputs in memory (fp32), same LDS tile shapes -- with
// every GEMM operand tile held in LDS as two f16 planes and every weight matrix read from the f16-plane arena.
// -------------------------------------------------------------------------------------------------
template <int H, bool SAVE = false>
__device__ __forceinline__ void node_role_h(const ComboNode& a, int tile, float* smem, int32_t* range_status,
                                            const NodeSave& ns TSD_TRACE_ARG) {
    constexpr int LDH = ldh_of(H);
    constexpr int NT = 2 * H, CB16 = 2, C4 = H / 4;
    const Planes pl = planes_at(smem, TN, LDH);
    const int n0 = tile * TN;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int col0 = wave * 32;
    const int nrows = min(TN, a.N - n0);
    f32x4 accm[CB16], accx[CB16];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float amax = 0.0f;
    HRing<CB16, HRING16_R> rg;
    // transposed accumulators (split16.hpp hgemm16_ring_run<..., TRANS>; inference tiles): lane = tile row l15, the four
    // consecutive channels 4 q .. 4 q + 3 of each 16-column block: 8-byte plane stores, 16-byte h / x1 accesses.  The two
    // bias vectors go through LDS (behind the planes), written by the first H threads behind the gather.
    constexpr bool TR = !SAVE && TSD_FILTER_TRANS != 0;
    float* s_bias = smem + TN * LDH;  // [2][H] (TR)

    float b_lin2[CB16], b_lin[CB16], h_res[CB16][4];
    f32x4 h4[CB16];  // (TR: the residual rows as 16-byte loads)
    float bl2 = 0.0f, bl = 0.0f;
    TSD_TRACE_ID(1);
    TSD_TRACE_AT(0);
    if (a.mode == 0) {
        if constexpr (TR) {
            if (tid < H) {
                bl2 = a.lin2_b[tid];
                bl = a.lin_b[tid];
            }
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb)
                h4[cb] = l15 < nrows ? *reinterpret_cast<const f32x4*>(a.h_in + (size_t)(n0 + l15) * H + col0 + cb * 16 + q * 4) : zero4;
        } else {
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
            b_lin2[cb] = a.lin2_b[col];
            b_lin[cb] = a.lin_b[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = q * 4 + r;
                h_res[cb][r] = row < nrows ? a.h_in[(size_t)(n0 + row) * H + col] : 0.0f;
            }
        }
        }
        aggregate_tile<H, SAVE, 2 * H / 64, 8, true>(a.row_ptr, a.dst, a.umap, a.Wf, a.x1_in, a.N, n0, smem,
                                                      SAVE ? ns.agg : nullptr, &amax);
        if constexpr (TR) {
            if (tid < H) {
                s_bias[tid] = bl2;
                s_bias[H + tid] = bl;
            }
        }
        hgemm16_ring_start<CB16, H>(rg, a.lin2_w, H, col0);  // (after the gather: its registers are the gather's)
        TSD_TRACE_WAVE(16);
        __syncthreads();
        TSD_TRACE_AT(1);
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) accm[cb] = accx[cb] = zero4;
        hgemm16_ring_run<CB16, H, TR>(rg, pl, LDH, accm, accx);
        hgemm16_ring_start<CB16, H>(rg, a.lin_w, H, col0);
        TSD_TRACE_WAVE(8);
        TSD_TRACE_AT(2);
        __syncthreads();
        if constexpr (TR) {
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const int c = col0 + cb * 16 + q * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(s_bias + c);
                f32x4 y4;
#pragma unroll
                for (int r = 0; r < 4; ++r) y4[r] = sspf(hval4(accm[cb], accx[cb], r) + b[r]);
                planes_store4(pl, l15 * LDH + c, y4, amax);
            }
        } else
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
            const float b = b_lin2[cb];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = hval4(accm[cb], accx[cb], r) + b, sv = sspf(v);
                planes_store1(pl, (q * 4 + r) * LDH + col, sv, amax);
                if constexpr (SAVE) {  // (the training step keeps the pre-activation and the activation: as node_role)
                    if (q * 4 + r < nrows) {
                        ns.x2[(size_t)(n0 + q * 4 + r) * H + col] = v;
                        ns.xs[(size_t)(n0 + q * 4 + r) * H + col] = sv;
                    }
                }
            }
        }
        __syncthreads();
        TSD_TRACE_AT(3);
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) accm[cb] = accx[cb] = zero4;
        hgemm16_ring_run<CB16, H, TR>(rg, pl, LDH, accm, accx);
        if (a.lin1_next_w != nullptr) hgemm16_ring_start<CB16, H>(rg, a.lin1_next_w, H, col0);
        TSD_TRACE_AT(4);
        __syncthreads();
        if constexpr (TR) {
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const int c = col0 + cb * 16 + q * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(s_bias + H + c);
                f32x4 hn = zero4;
                if (l15 < nrows) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) hn[r] = h4[cb][r] + (hval4(accm[cb], accx[cb], r) + b[r]);
                    float* hp = a.h + (size_t)(n0 + l15) * H + c;
                    if (a.ready)  // write-through (sc1): the pair tiles of this launch read the row from other CUs
                        store_stream16(hp, hn);
                    else
                        *reinterpret_cast<f32x4*>(hp) = hn;
                }
                planes_store4(pl, l15 * LDH + c, hn, amax);
            }
        } else
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
            const float b = b_lin[cb];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = q * 4 + r;
                float hn = 0.0f;
                if (row < nrows) {
                    hn = h_res[cb][r] + (hval4(accm[cb], accx[cb], r) + b);
                    if (a.ready)  // write-through (sc1): the pair tiles of this launch read the row from other CUs
                        __hip_atomic_store(a.h + (size_t)(n0 + row) * H + col, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        a.h[(size_t)(n0 + row) * H + col] = hn;
                }
                planes_store1(pl, row * LDH + col, hn, amax);
            }
        }
        if (a.ready) {  // Guideline 16 R1: every storing wave drains, the workgroup meets, ONE lane raises the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.ready + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.lin1_next_w == nullptr) {
            range_report(amax, range_status);
            return;
        }
        __syncthreads();
    } else {
        hgemm16_ring_start<CB16, H>(rg, a.lin1_next_w, H, col0);
        float site_m = 0.0f;
        for (int idx = tid; idx < TN * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            f32x4 v = zero4;
            if (r < nrows) v = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)(n0 + r) * H + c4 * 4);
            planes_store4(pl, r * LDH + c4 * 4, v, site_m);
        }
        site_close(amax, site_m);
        __syncthreads();
    }
    TSD_TRACE_AT(5);
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) accm[cb] = accx[cb] = zero4;
    hgemm16_ring_run<CB16, H, TR>(rg, pl, LDH, accm, accx);
    TSD_TRACE_AT(6);
    if constexpr (TR) {
        if (l15 < nrows) {
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                f32x4 x4;
#pragma unroll
                for (int r = 0; r < 4; ++r) x4[r] = hval4(accm[cb], accx[cb], r);
                *reinterpret_cast<f32x4*>(a.x1_out + (size_t)(n0 + l15) * H + col0 + cb * 16 + q * 4) = x4;
            }
        }
    } else
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) {
        const int col = col0 + cb * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            if (row < nrows) a.x1_out[(size_t)(n0 + row) * H + col] = hval4(accm[cb], accx[cb], r);
        }
    }
    TSD_TRACE_AT(7);
    range_report(amax, range_status);
}

// The node role on WIDE tiles (NRB 16-row blocks per workgroup; round 5): in a launch that is many chip-fulls deep -- the
// block launches of an 8-checkpoint ensemble carry 800 node tiles beside 1632 filter tiles -- a 16-row node tile streams
// the three 256-KiB weight images of the chain through its CU for 16 rows of work and holds a workgroup slot for as long as
// a 64-row filter tile does (profiles/r05_kernel_stats_ens8_base.md: the node tiles take as much slot time as the filter
// tiles).  Here a weight fragment of the ring feeds NRB row blocks (the unit encoder's hgemm16_ring_run_rb) and a wave
// gathers 2 NRB consecutive rows in one pass over its contiguous edge range.  Same arithmetic per row and element as
// node_role_h (transposed accumulators): bit-identical; inference only.  The small-launch forms keep 16 rows: there the
// chain's latency, not its slot time, is what counts.
#ifndef TSD_NODE_WIDE_RING
#define TSD_NODE_WIDE_RING 2
#endif
template <int H, int NRB>
__device__ __forceinline__ void node_role_hw(const ComboNode& a, int tile, float* smem, int32_t* range_status) {
    constexpr int LDH = ldh_of(H), TNR = TN * NRB;
    constexpr int NT = 2 * H, CB16 = 2, C4 = H / 4;
    const Planes pl = planes_at(smem, TNR, LDH);
    float* s_bias = smem + TNR * LDH;  // [2][H]
    const int n0 = tile * TNR;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int col0 = wave * 32;
    const int nrows = min(TNR, a.N - n0);
    const int nrb = (nrows + TN - 1) / TN;  // row blocks that hold atoms (uniform)
    f32x4 accm[NRB][CB16], accx[NRB][CB16];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float amax = 0.0f;
    constexpr int RW = TSD_NODE_WIDE_RING;  // ring depth (k-steps of weights in flight)
    HRing<CB16, RW> rg;
    auto zero_acc4 = [&]() {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) accm[rb][cb] = accx[rb][cb] = zero4;
    };
    if (a.mode == 0) {
        f32x4 h4[NRB][CB16];
        float bl2 = 0.0f, bl = 0.0f;
        if (tid < H) {
            bl2 = a.lin2_b[tid];
            bl = a.lin_b[tid];
        }
        aggregate_tile<H, false, 2 * H / 64, 8, true, false, TNR, TNR>(a.row_ptr, a.dst, a.umap, a.Wf, a.x1_in, a.N, n0, smem,
                                                                      nullptr, &amax);
        if (tid < H) {
            s_bias[tid] = bl2;
            s_bias[H + tid] = bl;
        }
        hgemm16_ring_start<CB16, H, RW>(rg, a.lin2_w, H, col0);
        __syncthreads();
        zero_acc4();
        hgemm16_ring_run_rb<NRB, CB16, H, true, RW>(rg, pl, LDH, accm, accx, nrb);
        hgemm16_ring_start<CB16, H, RW>(rg, a.lin_w, H, col0);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int c = col0 + cb * 16 + q * 4;
            const f32x4 b = *reinterpret_cast<const f32x4*>(s_bias + c);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
                if (rb < nrb) {
                    f32x4 y4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) y4[r] = sspf(hval4(accm[rb][cb], accx[rb][cb], r) + b[r]);
                    planes_store4(pl, (rb * TN + l15) * LDH + c, y4, amax);
                }
        }
        __syncthreads();
        // (the residual rows are requested here, not before the gather: they arrive under this GEMM, and the role has no
        // registers to hold them across the first one)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb)
                h4[rb][cb] = rb * TN + l15 < nrows
                                 ? *reinterpret_cast<const f32x4*>(a.h_in + (size_t)(n0 + rb * TN + l15) * H + col0 + cb * 16 + q * 4)
                                 : zero4;
        zero_acc4();
        hgemm16_ring_run_rb<NRB, CB16, H, true, RW>(rg, pl, LDH, accm, accx, nrb);
        if (a.lin1_next_w != nullptr) hgemm16_ring_start<CB16, H, RW>(rg, a.lin1_next_w, H, col0);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int c = col0 + cb * 16 + q * 4;
            const f32x4 b = *reinterpret_cast<const f32x4*>(s_bias + H + c);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
                if (rb < nrb) {
                    const int row = rb * TN + l15;
                    f32x4 hn = zero4;
                    if (row < nrows) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) hn[r] = h4[rb][cb][r] + (hval4(accm[rb][cb], accx[rb][cb], r) + b[r]);
                        *reinterpret_cast<f32x4*>(a.h + (size_t)(n0 + row) * H + c) = hn;
                    }
                    planes_store4(pl, row * LDH + c, hn, amax);
                }
        }
        if (a.lin1_next_w == nullptr) {
            range_report(amax, range_status);
            return;
        }
        __syncthreads();
    } else {
        hgemm16_ring_start<CB16, H, RW>(rg, a.lin1_next_w, H, col0);
        float site_m = 0.0f;
        for (int idx = tid; idx < TNR * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            f32x4 v = zero4;
            if (r < nrows) v = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)(n0 + r) * H + c4 * 4);
            planes_store4(pl, r * LDH + c4 * 4, v, site_m);
        }
        site_close(amax, site_m);
        __syncthreads();
    }
    zero_acc4();
    hgemm16_ring_run_rb<NRB, CB16, H, true, RW>(rg, pl, LDH, accm, accx, nrb);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
        if (rb < nrb && rb * TN + l15 < nrows) {
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                f32x4 x4;
#pragma unroll
                for (int r = 0; r < 4; ++r) x4[r] = hval4(accm[rb][cb], accx[rb][cb], r);
                *reinterpret_cast<f32x4*>(a.x1_out + (size_t)(n0 + rb * TN + l15) * H + col0 + cb * 16 + q * 4) = x4;
            }
        }
    range_report(amax, range_status);
}

// RB = 32-row blocks of one tile: 1 in the small launches (more, shorter tiles), 2 where a launch is many chip-fulls
// of filter tiles (a weight fragment from the ring then feeds two row blocks: the tile is bound by the weight feed).
template <int H, int RB, bool SAVE = false>
__device__ __forceinline__ void filter_role_h(const ComboFilter& f, int item, float* smem, int32_t* range_status,
                                              const FilterSave& fsv TSD_TRACE_ARG) {
    constexpr int TT = T * RB;
    const int g = f.g_begin + item;
    const int lrel = g / f.tiles_per_layer, tile = g - lrel * f.tiles_per_layer;
    const float* Wb = f.Wl0 + (size_t)(f.layer0 + lrel) * f.layer_stride;
    const float *nn0_w = Wb + f.o_nn0_w, *nn0_b = Wb + f.o_nn0_b, *nn2_w = Wb + f.o_nn2_w, *nn2_b = Wb + f.o_nn2_b;
    float* out = f.wf + (size_t)(lrel % f.wf_slots) * f.wf_layer_stride;
    constexpr int LDH = ldh_of(H), LDA = H + 4;
    constexpr int NT = 2 * H, C4 = H / 4;
    const Planes pl = planes_at(smem, TT, LDH);
    float* buf = smem;             // the finished filter tile as fp32 rows (over the planes: TT (H + 4) <= TT (H + 8) floats)
    float* s_c = smem + TT * LDH;  // (planes: 2 x TT x LDH f16 = TT x LDH floats)
    // TRANSPOSED ACCUMULATORS (split16.hpp hgemm_ring_run<..., TRANS>; inference tiles): a lane holds four runs of four
    // consecutive channels of ONE tile row -- 8-byte plane stores, 16-byte filter-tile stores, one cutoff weight per lane;
    // the 16 channels' biases come from LDS.  The saving form keeps the plain layout (its side stores are row-contiguous
    // per wave that way).  Bit-identical either way.
#ifndef TSD_FROLE_TRANS
#define TSD_FROLE_TRANS TSD_FILTER_TRANS
#endif
    constexpr bool TR = !SAVE && TSD_FROLE_TRANS != 0;
    float* s_bias = s_c + TT;      // [2][H] nn.0 / nn.2 biases (TR)

    const int E = *f.e.count;
    const int e0 = tile * TT;
    if (e0 >= E) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32, col = col0 + l31;
    const int nrows = min(TT, E - e0);
    float amax = 0.0f;
    TSD_TRACE_ID(2);
    TSD_TRACE_AT(0);

    HRing<1, HRING_R> rg;
    hgemm_ring_start<1, H>(rg, nn0_w, H, col0);
    for (int r = tid; r < TT; r += NT) s_c[r] = r < nrows ? cutoff_weight(f.e.dist[e0 + r], f.conv_cutoff, f.smooth) : 0.0f;
    float b0 = 0.0f, b2 = 0.0f;
    if constexpr (TR) {
        for (int c = tid; c < H; c += NT) {
            s_bias[c] = nn0_b[c];
            s_bias[H + c] = nn2_b[c];
        }
    } else {
        b0 = nn0_b[col];
        b2 = nn2_b[col];
    }
    {   // edge_attr tile -> LDS planes with every load of a thread in flight together (rows past the end clamped)
        constexpr int NIT = TT * C4 / NT;
        static_assert(TT * C4 % NT == 0, "tile / block mismatch");
        static_assert(NIT == 4 * RB, "ld16_wait4 names four registers");
        f32x4 v[RB][4];  // (sc1: in the one-launch forward the rows were stored by other workgroups of this launch)
        float site_m = 0.0f;  // (max |a| of one conversion site: split16.hpp site_close)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            ld16_sc1(v[it / 4][it % 4], f.edge_attr + (size_t)(e0 + min(r, nrows - 1)) * H + c4 * 4);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) ld16_wait4(v[rb]);
        if constexpr (!SAVE) {   // the rows ARE the planes (common.hpp ATTRIBUTE ROWS AS f16 PLANES; checked where they were written): 16-byte copies
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                planes_put_chunk<H>(pl, r * LDH, c4, r < nrows ? v[it / 4][it % 4] : z);
            }
        } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            planes_store4(pl, r * LDH + c4 * 4, r < nrows ? v[it / 4][it % 4] : z, site_m);
        }
        site_close(amax, site_m);
        }
    }
    __syncthreads();
    TSD_TRACE_AT(1);

    f32x16 accm[RB][1], accx[RB][1];
    hzero(accm, accx);
    hgemm_ring_run<RB, 1, H, false, TR>(rg, pl, LDH, accm, accx);
    hgemm_ring_start<1, H>(rg, nn2_w, H, col0);
    TSD_TRACE_WAVE(8);
    TSD_TRACE_AT(2);
    __syncthreads();
    if constexpr (TR) {
        // (group-major: the next group's bias vector is requested before this group's stores -- a wave's LDS accesses stay in
        // program order, and the role has no registers for all 16 biases at once: two workgroups per CU = 128 VGPRs)
        const float* bb = s_bias + col0 + 4 * hi;
        f32x4 bn = *reinterpret_cast<const f32x4*>(bb);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 bv = bn;
            if (g4 < 3) bn = *reinterpret_cast<const f32x4*>(bb + 8 * (g4 + 1));
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f32x4 y4;
#pragma unroll
                for (int r = 0; r < 4; ++r) y4[r] = sspf(hval(accm[rb][0], accx[rb][0], 4 * g4 + r) + bv[r]);
                planes_store4(pl, (rb * T + l31) * LDH + col0 + 8 * g4 + 4 * hi, y4, amax);
            }
        }
    } else
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * T + acc_row(r, hi);
            const float v = hval(accm[rb][0], accx[rb][0], r) + b0, sv = sspf(v);
            planes_store1(pl, row * LDH + col, sv, amax);
            if constexpr (SAVE) {  // the training step keeps every block's filters and activations: slot = block
                if (row < nrows) {
                    const size_t o = (size_t)lrel * f.wf_layer_stride + (size_t)(e0 + row) * H + col;
                    fsv.f0[o] = v;
                    fsv.fs[o] = sv;
                }
            }
        }
    __syncthreads();
    TSD_TRACE_AT(3);

    hzero(accm, accx);
    hgemm_ring_run<RB, 1, H, false, TR>(rg, pl, LDH, accm, accx);
    TSD_TRACE_WAVE(16);
    TSD_TRACE_AT(4);
    __syncthreads();
    if constexpr (TR) {
        const float* bb = s_bias + H + col0 + 4 * hi;
        float cw[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) cw[rb] = s_c[rb * T + l31];
        f32x4 bn = *reinterpret_cast<const f32x4*>(bb);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 bv = bn;
            if (g4 < 3) bn = *reinterpret_cast<const f32x4*>(bb + 8 * (g4 + 1));
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f32x4 w4;
#pragma unroll
                for (int r = 0; r < 4; ++r) w4[r] = (hval(accm[rb][0], accx[rb][0], 4 * g4 + r) + bv[r]) * cw[rb];
                *reinterpret_cast<f32x4*>(buf + (rb * T + l31) * LDA + col0 + 8 * g4 + 4 * hi) = w4;
            }
        }
    } else
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb * T + acc_row(r, hi);
            buf[row * LDA + col] = (hval(accm[rb][0], accx[rb][0], r) + b2) * s_c[row];
        }
    __syncthreads();
    TSD_TRACE_AT(5);
    for (int idx = tid; idx < nrows * C4; idx += NT) {
        const int r = idx / C4, c4 = idx % C4;
        store_stream16(out + (size_t)(e0 + r) * H + c4 * 4, *reinterpret_cast<const f32x4*>(buf + r * LDA + c4 * 4));
    }
    TSD_TRACE_AT(6);
    range_report(amax, range_status);
}

// SAVE (training step): the tile's inputs and pre-/post-activations also go to `sv` (as pair_output_kernel<H, true>).
// HSPLIT (the stand-alone launch, whose budget is 80 VGPRs = three workgroups per CU): the h rows in two batches of two
// chunks per thread (16 registers instead of 32 beside the live accumulators; the other workgroups of the CU hide the
// second round trip).
// RB: 32-row blocks per pair tile (2: the one-launch forward's pair tiles where they fill the chip -- when the node chain ends
// every pair tile of the launch streams the MLP's 384 KB of weights at once; a fragment that feeds two row blocks halves that).
template <int H, bool SAVE = false, bool HSPLIT = false, int RB = 1>
__device__ __forceinline__ void pair_role_h(const ComboPre& q, int tile, int node_tiles, float* smem, bool defer_pre,
                                            int32_t* range_status, const PairSave& sv = PairSave{}) {
    constexpr int LDH = ldh_of(H), NT = 2 * H, C4 = H / 4, NW = H / 64, TT = T * RB;
    const Planes pl = planes_at(smem, TT, LDH);
    float* s_red = smem + TT * LDH;  // [NW][TT]
    int* s_src = reinterpret_cast<int*>(s_red + NW * TT);
    int* s_dst = s_src + TT;
    int* s_row = s_dst + TT;
    const int E = *q.e.count;
    const int e0 = tile * TT;
    if (e0 >= E) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = wave * 32, col = col0 + l31;
    const int nrows = min(TT, E - e0);
    float amax = 0.0f;
#ifndef TSD_PAIR_TRANS
#define TSD_PAIR_TRANS TSD_FILTER_TRANS
#endif
    constexpr bool TR = !SAVE && TSD_PAIR_TRANS != 0;  // transposed accumulators (see filter_role_h): lane = tile row l31
#ifndef TSD_PAIR_TR3
#define TSD_PAIR_TR3 1
#endif
#ifndef TSD_PAIR_TR12
#define TSD_PAIR_TR12 1
#endif
    constexpr bool TR3 = TR && TSD_PAIR_TR3 != 0, TR12 = TR && TSD_PAIR_TR12 != 0;
    static_assert(RB == 1 || (TR3 && TR12 && !HSPLIT), "64-row pair tiles: the transposed inference form");
    if (tid < TT) {
        const int ee = e0 + min(tid, nrows - 1);
        s_src[tid] = q.e.src[ee];
        s_dst[tid] = q.e.dst[ee];
        int row = q.attr_row ? q.attr_row[ee] : ee;
        if constexpr (SAVE) {
            if (row >= sv.attr_from) row -= sv.attr_shift;
        }
        s_row[tid] = row;
    }
    __syncthreads();
    // ONE accumulation chain for Linear(2H, H) of the concatenation [h_src * h_dst, attr] (common.py:226-229): the attribute
    // half first (it does not depend on h: the one-launch forward runs it while the node tiles finish), then the h half
    // into the SAME accumulators -- no 16 registers of intermediate sums beside them (80 VGPRs: three workgroups per CU)
    f32x16 accm[RB][1], accx[RB][1];
    constexpr int PR = HSPLIT ? 2 : HRING_R;  // weight k-steps in flight (HSPLIT: six waves per SIMD hide the rest)
    constexpr int HB = HSPLIT ? 2 : 4;  // chunks of the gathered h rows per thread and batch (NIT = 4 in all)
    f32x4 hs[HB], hd[HB];
    auto issue_h = [&](int it0) {
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            const int idx = tid + (it0 + k) * NT, r = idx / C4, c4 = idx % C4;
            const float* ps = q.h + (size_t)s_src[r] * H + c4 * 4;
            const float* pd = q.h + (size_t)s_dst[r] * H + c4 * 4;
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hs[k]) : "v"(ps) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hd[k]) : "v"(pd) : "memory");
        }
    };
    auto pre_gemm = [&]() {
        HRing<1, PR> rg;
        constexpr int NIT = TT * C4 / NT;
        static_assert(NIT == 4 * RB, "ld16_wait4 names four registers");
        float site_m = 0.0f;
#pragma unroll
        for (int ib = 0; ib < NIT; ib += 4) {   // (four chunks per thread = 32 tile rows per round trip)
        f32x4 v[4];  // (sc1: see filter_role_h)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + (ib + it) * NT, r = idx / C4, c4 = idx % C4;
            ld16_sc1(v[it], q.edge_attr + (size_t)s_row[r] * H + c4 * 4);
        }
        ld16_wait4(v);
        if constexpr (!SAVE) {   // (rows stored as f16 planes: common.hpp ATTRIBUTE ROWS AS f16 PLANES)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = tid + (ib + it) * NT, r = idx / C4, c4 = idx % C4;
                planes_put_chunk<H>(pl, r * LDH, c4, v[it]);
            }
        } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + (ib + it) * NT, r = idx / C4, c4 = idx % C4;
            planes_store4(pl, r * LDH + c4 * 4, v[it], site_m);
            if constexpr (SAVE) {
                if (r < nrows) *reinterpret_cast<f32x4*>(sv.hp + (size_t)(e0 + r) * 2 * H + H + c4 * 4) = v[it];
            }
        }
        }
        }
        if constexpr (SAVE) site_close(amax, site_m);
        hgemm_ring_start<1, H>(rg, q.w0b, H, col0);  // (behind the staging: its registers are free now)
        __syncthreads();
        hzero(accm, accx);
        hgemm_ring_run<RB, 1, H, false, TR12>(rg, pl, LDH, accm, accx);
    };
    // the first layer's biases: the one-launch forward (whose pair tiles end its dependency chain) requests them here, ahead of
    // the wait for the node tiles; the stand-alone launch (80 VGPRs) where it adds them
    f32x4 b0v[4];
    constexpr bool B0_EARLY = TR12 && !HSPLIT && RB == 1;  // (the 64-row form has no registers for them across the GEMMs)
    if constexpr (B0_EARLY) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) b0v[g4] = *reinterpret_cast<const f32x4*>(q.b0 + col0 + 8 * g4 + 4 * hi);
    }
    if (!defer_pre) pre_gemm();
    // wait for the node tiles that hold this tile's atoms (min .. max node id over both end points)
    if (wave == 0 && q.ready != nullptr) {  // (ready == NULL: the stand-alone pair output, h is complete)
        int lo = lane < TT ? min(s_src[lane], s_dst[lane]) : 0x7fffffff;
        int hn = lane < TT ? max(s_src[lane], s_dst[lane]) : -1;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off));
            hn = max(hn, __shfl_xor(hn, off));
        }
        const int t_lo = lo / q.ready_div, t_hi = hn / q.ready_div;
        bool gave_up = false;
        for (int t0 = t_lo; t0 <= t_hi && !gave_up; t0 += 64) {
            const int t = t0 + lane;
            const bool need = t <= t_hi && t < node_tiles;
            for (unsigned spins = 0;; ++spins) {
                const int f = need ? __hip_atomic_load(q.ready + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
                if (__all(f >= q.ready_target)) break;
                // (a launch in which another wait has already given up unwinds quickly: see mega_wait_failed)
                if (spins > PAIR_SPIN_LIMIT ||
                    ((spins & 1023u) == 1023u &&
                     (__hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & TSD_STATUS_INTERNAL))) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(TSD_PAIR_SLEEP);
            }
        }
        if (gave_up && lane == 0) atomicOr(q.status, TSD_STATUS_INTERNAL);
    }
    __syncthreads();  // (also: every wave is done reading the planes)
    if (defer_pre) {
        pre_gemm();
        __syncthreads();
    }
    HRing<1, PR> rg;
    {   // h_src * h_dst -> LDS planes.  The rows were written by other CUs during this launch: sc1 loads
        constexpr int NIT = TT * C4 / NT;
        static_assert(TT * C4 % NT == 0, "tile / block mismatch");
        static_assert(NIT == 4 * RB, "the wait statement names 2 x 4 registers");
        // (requesting these rows together with the attribute rows in the stand-alone launch -- one round trip instead of two,
        // 32 more registers across the first GEMM -- measured nothing: 1.372 vs 1.377 ms/step with 8 checkpoints, round 5)
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += HB) {
            issue_h(it0);
            if constexpr (HSPLIT)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(hs[0]), "+v"(hs[1]), "+v"(hd[0]), "+v"(hd[1]) :: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(hs[0]), "+v"(hs[1]), "+v"(hs[HB - 2]), "+v"(hs[HB - 1]), "+v"(hd[0]), "+v"(hd[1]),
                               "+v"(hd[HB - 2]), "+v"(hd[HB - 1])
                             :: "memory");
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int idx = tid + (it0 + k) * NT, r = idx / C4, c4 = idx % C4;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                planes_store4(pl, r * LDH + c4 * 4, r < nrows ? hs[k] * hd[k] : z, amax);
                if constexpr (SAVE) {
                    if (r < nrows) *reinterpret_cast<f32x4*>(sv.hp + (size_t)(e0 + r) * 2 * H + c4 * 4) = hs[k] * hd[k];
                }
            }
        }
    }
    hgemm_ring_start<1, H>(rg, q.w0a, H, col0);  // (behind the staging: its registers are free now)
    __syncthreads();
    hgemm_ring_run<RB, 1, H, false, TR12>(rg, pl, LDH, accm, accx);  // (on top of the attribute half's sums)
    __syncthreads();
    if constexpr (TR12) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 bv;
            if constexpr (!B0_EARLY) bv = *reinterpret_cast<const f32x4*>(q.b0 + col0 + 8 * g4 + 4 * hi);
            else bv = b0v[g4];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bool live = rb * T + l31 < nrows;  // (rows past the end: zeros, as their h rows)
                f32x4 s4;
#pragma unroll
                for (int r = 0; r < 4; ++r) s4[r] = swishf(live ? hval(accm[rb][0], accx[rb][0], 4 * g4 + r) + bv[r] : 0.0f);
                planes_store4(pl, (rb * T + l31) * LDH + col0 + 8 * g4 + 4 * hi, s4, amax);
            }
        }
    } else
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = acc_row(r, hi) < nrows ? hval(accm[0][0], accx[0][0], r) + q.b0[col] : 0.0f, sg = swishf(v);
        planes_store1(pl, acc_row(r, hi) * LDH + col, sg, amax);
        if constexpr (SAVE) {
            const int row = acc_row(r, hi);
            if (row < nrows) {
                sv.g0[(size_t)(e0 + row) * H + col] = v;
                sv.gs0[(size_t)(e0 + row) * H + col] = sg;
            }
        }
    }
    __syncthreads();
    if (wave < NW) {  // H -> H/2 and the final dot: the first H/64 waves
        const int c2 = wave * 32 + l31;
        hzero(accm, accx);
        hgemm_tile<RB, 1, H, false, PR, TR3>(pl, LDH, q.w1, H / 2, wave * 32, accm, accx);
        if constexpr (TR3) {   // the lane's 16 channels of row l31 summed in register order, then the other half-wave's 16
            float v[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) v[rb] = 0.0f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(q.b1 + wave * 32 + 8 * g4 + 4 * hi);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(q.w2 + wave * 32 + 8 * g4 + 4 * hi);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[rb] += swishf(hval(accm[rb][0], accx[rb][0], 4 * g4 + r) + bv[r]) * wv[r];
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                v[rb] += __shfl_xor(v[rb], 32);
                if (hi == 0) s_red[wave * TT + rb * T + l31] = v[rb];
            }
        } else {
        const float b = q.b1[c2], w2 = q.w2[c2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float g = hval(accm[0][0], accx[0][0], r) + b, sg = swishf(g);
            if constexpr (SAVE) {
                const int row = acc_row(r, hi);
                if (row < nrows) {
                    sv.g1[(size_t)(e0 + row) * (H / 2) + c2] = g;
                    sv.gs1[(size_t)(e0 + row) * (H / 2) + c2] = sg;
                }
            }
            float v = sg * w2;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            if (l31 == 0) s_red[wave * TT + acc_row(r, hi)] = v;
        }
        }
    }
    __syncthreads();
    if (tid < nrows) {
        float v = s_red[tid];
#pragma unroll
        for (int k = 1; k < NW; ++k) v += s_red[k * TT + tid];
        q.edge_inv[e0 + tid] = v + q.b2[0];
    }
    range_report(amax, range_status);
}

// -------------------------------------------------------------------------------------------------
// Backward node chain of the training step between two blocks, one tile of TN nodes per workgroup -- the adjoint
// of the node role above, same shape (one gather + three 16-row GEMMs):
//   dx1_l  = sum_e dagg_l[dst e] * Wf_l[umap e]            (the symmetric edge set makes the adjoint the same gather)
//   dh_l   = dh_{l+1} + dx1_l . W_lin1_l                   (residual + lin1 dgrad; schnet.py:223-224)
//   dx2_{l-1}  = (dh_l . W_lin_{l-1}) * ssp'(x2_{l-1})
//   dagg_{l-1} = dx2_{l-1} . W_lin2_{l-1}
// first != 0: the chain starts at dh_l (given), no gather (the top block); last != 0: it stops at dh_0.
// The weight gradients of the three layers read dx1, dh, dx2 later (batched); weights in the dgrad layout.
// -------------------------------------------------------------------------------------------------
struct NodeBwd {
    int N, first, last;
    const int32_t *row_ptr, *dst, *umap;
    const float *Wf, *dagg_in;       // block l
    const float *dh_up;              // d loss / d h_{l+1}  (first: d loss / d h_l itself)
    const float *w_lin1_t;           // block l
    const float *w_lin_t, *w_lin2_t; // block l-1
    const float *x2_prev;            // block l-1 pre-activation
    float *dx1, *dh, *dx2_prev, *dagg_prev;
    NodeAmax am;                     // split-f16 step: running maxima of what this chain writes (or NULL pointers)
};
// AMAX: the tile's max |dh_up| (first), |dx1|, |dh|, |dx2_prev| go to a.am (the node-level weight gradients scale their dY
// operands by them: split16.hpp, GRADIENT operands) -- one conditional atomic per tile and word
template <int H, bool AMAX = false>
__device__ __forceinline__ void node_bwd_role(const NodeBwd& a, int tile, float* smem) {
    constexpr int LDA = H + 4, NT = 2 * H, CB16 = 2, C4 = H / 4;
    float* buf = smem;
    const int n0 = tile * TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int col0 = wave * 32;
    const int nrows = min(TN, a.N - n0);
    f32x4 acc[CB16];
    float pre[CB16][4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float m_in = 0.0f, m_dx1 = 0.0f, m_dh = 0.0f, m_dx2 = 0.0f;
    auto flush = [&]() {  // (before every return of the role)
        if constexpr (AMAX) {
            float* s_m = smem + TN * LDA;  // [4][NT / 64] behind the tile
            constexpr int NWV = NT / 64;
            m_in = max64(m_in), m_dx1 = max64(m_dx1), m_dh = max64(m_dh), m_dx2 = max64(m_dx2);
            __syncthreads();
            if (lane == 0) {
                s_m[wave] = m_in;
                s_m[NWV + wave] = m_dx1;
                s_m[2 * NWV + wave] = m_dh;
                s_m[3 * NWV + wave] = m_dx2;
            }
            __syncthreads();
            if (tid < 4) {
                float* slot = tid == 0 ? a.am.in : tid == 1 ? a.am.dx1 : tid == 2 ? a.am.dh : a.am.dx2;
                float t = s_m[tid * NWV];
#pragma unroll
                for (int k = 1; k < NWV; ++k) t = fmaxf(t, s_m[tid * NWV + k]);
                if (slot != nullptr && t > 0.0f) atomic_amax(slot, t);
            }
        }
    };
    if (!a.first) {
        aggregate_tile<H, true>(a.row_ptr, a.dst, a.umap, a.Wf, a.dagg_in, a.N, n0, buf, a.dx1);
        __syncthreads();
        if constexpr (AMAX) {  // the aggregated dx1 tile (rows past N are zero)
            for (int idx = tid; idx < TN * C4; idx += NT) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(buf + (idx / C4) * LDA + (idx % C4) * 4);
                amax_upd2(m_dx1, v[0], v[1]);
                amax_upd2(m_dx1, v[2], v[3]);
            }
        }
        // (values the epilogues read are requested before the GEMM that precedes them, rows clamped: no guarded loads)
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pre[cb][r] = a.dh_up[(size_t)(n0 + min(q * 4 + r, nrows - 1)) * H + col0 + cb * 16 + l15];
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) acc[cb] = zero4;
        gemm_tile16<CB16, H>(buf, LDA, a.w_lin1_t, H, col0, acc);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = q * 4 + r;
                float v = 0.0f;
                if (row < nrows) {
                    v = pre[cb][r] + acc[cb][r];
                    a.dh[(size_t)(n0 + row) * H + col] = v;
                }
                if constexpr (AMAX) amax_upd(m_dh, v);
                buf[row * LDA + col] = v;
            }
        }
    } else {
        for (int idx = tid; idx < TN * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            f32x4 v = zero4;
            if (r < nrows) v = *reinterpret_cast<const f32x4*>(a.dh_up + (size_t)(n0 + r) * H + c4 * 4);
            if constexpr (AMAX) {
                amax_upd2(m_in, v[0], v[1]);
                amax_upd2(m_in, v[2], v[3]);
            }
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = v;
        }
    }
    if (a.last) {
        flush();
        return;
    }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            pre[cb][r] = a.x2_prev[(size_t)(n0 + min(q * 4 + r, nrows - 1)) * H + col0 + cb * 16 + l15];
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) acc[cb] = zero4;
    gemm_tile16<CB16, H>(buf, LDA, a.w_lin_t, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) {
        const int col = col0 + cb * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            float v = 0.0f;
            if (row < nrows) {
                v = acc[cb][r] * act_deriv(1, pre[cb][r]);
                a.dx2_prev[(size_t)(n0 + row) * H + col] = v;
            }
            if constexpr (AMAX) amax_upd(m_dx2, v);
            buf[row * LDA + col] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) acc[cb] = zero4;
    gemm_tile16<CB16, H>(buf, LDA, a.w_lin2_t, H, col0, acc);
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) {
        const int col = col0 + cb * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            if (row < nrows) a.dagg_prev[(size_t)(n0 + row) * H + col] = acc[cb][r];
        }
    }
    flush();
}
// -------------------------------------------------------------------------------------------------
// The filter MLP's whole backward chain for one tile of 32 undirected edges (the adjoint of the filter role;
// schnet.py:94-99 backwards):
//   dWf = (dagg_i * x1_j + dagg_j * x1_i) * C(d)     -> global (the weight gradient of nn.2 reads it), LDS
//   df0 = (dWf . W_nn2) * ssp'(f0)                    -> global (weight gradient of nn.0), LDS
//   d_ea += df0 . W_nn0                               (read-modify-write of the tile's own rows: deterministic)
// instead of aggregate_bwd_filter + two dgrad launches with dWf / df0 read back from HBM in between.
// -------------------------------------------------------------------------------------------------
struct FilterBwd {
    int tiles;  // 0: no filter role
    tsd_edges eu;
    const float *dagg, *x1, *f0, *W2t, *W0t;
    float cutoff;
    int smooth;
    float *dWf, *df0, *d_ea;
    float* amax;  // split-f16 form: [2] running max |dWf|, max |df0| of the block (non-negative floats, atomicMax), or NULL
};
template <int H>
__device__ __forceinline__ void filter_bwd_role(const FilterBwd& f, int tile, float* smem) {
    constexpr int TT = 32, LDA = H + 4, NT = 2 * H, C4 = H / 4;
    float* buf = smem;
    float* s_c = smem + TT * LDA;
    int* s_i = reinterpret_cast<int*>(s_c + TT);
    int* s_j = s_i + TT;
    const int E = *f.eu.count;
    const int e0 = tile * TT;
    if (e0 >= E) return;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32;
    const int nrows = min(TT, E - e0);
    if (tid < TT) {
        const bool v = tid < nrows;
        s_i[tid] = v ? f.eu.src[e0 + tid] : 0;
        s_j[tid] = v ? f.eu.dst[e0 + tid] : 0;
        s_c[tid] = v ? cutoff_weight(f.eu.dist[e0 + tid], f.cutoff, f.smooth) : 0.0f;
    }
    __syncthreads();
    {   // every load of a thread in flight together (rows past the end are clamped; a guarded load per iteration
        // makes the compiler wait for each one: 4 serial L2 round trips per tile)
        constexpr int NIT = TT * C4 / NT;
        static_assert(TT * C4 % NT == 0, "tile / block mismatch");
        f32x4 di[NIT], dj[NIT], xi[NIT], xj[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = min(idx / C4, nrows - 1), c4 = idx % C4;
            const size_t oi = (size_t)s_i[r] * H + c4 * 4, oj = (size_t)s_j[r] * H + c4 * 4;
            di[it] = *reinterpret_cast<const f32x4*>(f.dagg + oi);
            dj[it] = *reinterpret_cast<const f32x4*>(f.dagg + oj);
            xi[it] = *reinterpret_cast<const f32x4*>(f.x1 + oi);
            xj[it] = *reinterpret_cast<const f32x4*>(f.x1 + oj);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / C4, c4 = idx % C4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < nrows) {
                v = (di[it] * xj[it] + dj[it] * xi[it]) * s_c[r];
                *reinterpret_cast<f32x4*>(f.dWf + (size_t)(e0 + r) * H + c4 * 4) = v;
            }
            *reinterpret_cast<f32x4*>(buf + r * LDA + c4 * 4) = v;
        }
    }
    __syncthreads();
    const int col = col0 + l31;
    // the pre-activations (and, below, the old attribute-gradient values) of this lane's outputs are requested
    // BEFORE the GEMM that needs them afterwards: their latency hides under its MFMAs
    float pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = f.f0[(size_t)(e0 + min(acc_row(r, hi), nrows - 1)) * H + col];
    f32x16 acc[1][1];
    zero_acc(acc);
    gemm_tile<1, 1, H>(buf, LDA, f.W2t, H, col0, acc);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        float v = 0.0f;
        if (row < nrows) {
            v = acc[0][0][r] * act_deriv(1, pre[r]);
            f.df0[(size_t)(e0 + row) * H + col] = v;
        }
        buf[row * LDA + col] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = f.d_ea[(size_t)(e0 + min(acc_row(r, hi), nrows - 1)) * H + col];
    zero_acc(acc);
    gemm_tile<1, 1, H>(buf, LDA, f.W0t, H, col0, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        if (row < nrows) f.d_ea[(size_t)(e0 + row) * H + col] = pre[r] + acc[0][0][r];
    }
}

// The same chain on the f16 MFMA pipes (split16.hpp, GRADIENT operands): the dWf tile is converted row by row scaled to
// [1, 2) (a row is held by ONE wave during staging: the row max is a wave reduction), df0 with one scale per tile (its
// rows are spread over the waves: the tile max goes through LDS beside the barrier the planes need anyway); W2t16 / W0t16
// are the f16-plane images of the dgrad matrices (pack mode 4).  The running maxima of dWf and df0 (true values) go to
// amax[0] / amax[1]: the batched weight-gradient launch scales its dY operands by them.
#ifndef TSD_FBWD_RING
#define TSD_FBWD_RING 5  // weight k-steps in flight per wave of the backward filter chain
#endif
template <int H>
__device__ __forceinline__ void filter_bwd_role_h(const FilterBwd& f, int tile, float* smem) {
    constexpr int TT = 32, LDH = ldh_of(H), NT = 2 * H, C4 = H / 4, NW = NT / 64;
    const Planes pl = planes_at(smem, TT, LDH);
    float* s_c = smem + TT * LDH;
    float* s_inv = s_c + TT;    // [TT] 2^e of the dWf rows
    float* s_wmax = s_inv + TT; // [NW] per-wave max of the df0 tile
    float* s_rmax = s_wmax + NW; // [TT] max |dWf| of the rows
    int* s_i = reinterpret_cast<int*>(s_rmax + TT);
    int* s_j = s_i + TT;
    const int E = *f.eu.count;
    const int e0 = tile * TT;
    if (e0 >= E) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = wave * 32;
    const int nrows = min(TT, E - e0);
    if (tid < TT) {
        const bool v = tid < nrows;
        s_i[tid] = v ? f.eu.src[e0 + tid] : 0;
        s_j[tid] = v ? f.eu.dst[e0 + tid] : 0;
        s_c[tid] = v ? cutoff_weight(f.eu.dist[e0 + tid], f.cutoff, f.smooth) : 0.0f;
    }
    __syncthreads();
    float dummy = 0.0f;
    {
        constexpr int NIT = TT * C4 / NT;
        static_assert(TT * C4 % NT == 0 && (C4 == 64 || C4 == 32), "a row is held by one wave (or half of one)");
        // (two halves: all sixteen row loads of a thread in flight together hold 64 registers beside the conversion's own)
        constexpr int HB = NIT;  // (NIT / 2: two halves, 32 registers less in flight)
        static_assert(NIT % HB == 0, "");
#pragma unroll
        for (int h0 = 0; h0 < NIT; h0 += HB) {
            f32x4 di[HB], dj[HB], xi[HB], xj[HB];
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int idx = tid + (h0 + k) * NT, r = min(idx / C4, nrows - 1), c4 = idx % C4;
                const size_t oi = (size_t)s_i[r] * H + c4 * 4, oj = (size_t)s_j[r] * H + c4 * 4;
                di[k] = *reinterpret_cast<const f32x4*>(f.dagg + oi);
                dj[k] = *reinterpret_cast<const f32x4*>(f.dagg + oj);
                xi[k] = *reinterpret_cast<const f32x4*>(f.x1 + oi);
                xj[k] = *reinterpret_cast<const f32x4*>(f.x1 + oj);
            }
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const int idx = tid + (h0 + k) * NT, r = idx / C4, c4 = idx % C4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < nrows) {
                    v = (di[k] * xj[k] + dj[k] * xi[k]) * s_c[r];
                    *reinterpret_cast<f32x4*>(f.dWf + (size_t)(e0 + r) * H + c4 * 4) = v;
                }
                float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                m = C4 == 64 ? max64(m) : max32(m);  // the row's max
                float inv;
                const float sc = pow2_scale(m, inv);
                if (c4 == 0) {
                    s_inv[r] = inv;
                    s_rmax[r] = m;
                }
                planes_store4(pl, r * LDH + c4 * 4, v * sc, dummy);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int col = col0 + l31;
    float pre[16];
    __syncthreads();
    f32x16 accm[1][1], accx[1][1];
    hzero(accm, accx);
    hgemm_tile<1, 1, H, true, TSD_FBWD_RING>(pl, LDH, f.W2t, H, col0, accm, accx);
    // (the pre-activations, and below the old attribute-gradient values, are read BEHIND the GEMM that precedes their
    // use: sixteen values held across the MFMA stream put the kernel over the 128-register line of two workgroups
    // per CU, whose interleaving hides this latency anyway)
    // (tile base pointers are wave-uniform: SGPR base + one 32-bit lane offset per row, shared by the four accesses below)
    const float* f0 = f.f0 + (size_t)e0 * H;
    float* df0 = f.df0 + (size_t)e0 * H;
    float* d_ea = f.d_ea + (size_t)e0 * H;
    unsigned off[16];  // bytes: SGPR base + 32-bit lane offset
    int hi_p = hi;     // (through an asm statement: the offsets and the loads stay behind the GEMM's asm statements)
    asm volatile("" : "+v"(hi_p));
#pragma unroll
    for (int r = 0; r < 16; ++r) off[r] = (unsigned)(min(acc_row(r, hi_p), nrows - 1) * H + col) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(f0) + off[r]);
    float v0[16], m0 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        v0[r] = 0.0f;
        if (row < nrows) {
            v0[r] = hval(accm[0][0], accx[0][0], r) * s_inv[row] * act_deriv(1, pre[r]);
            *reinterpret_cast<float*>(reinterpret_cast<char*>(df0) + off[r]) = v0[r];
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
    hgemm_tile<1, 1, H, true, TSD_FBWD_RING>(pl, LDH, f.W0t, H, col0, accm, accx);
    hi_p = hi;
    asm volatile("" : "+v"(hi_p));
#pragma unroll
    for (int r = 0; r < 16; ++r) off[r] = (unsigned)(min(acc_row(r, hi_p), nrows - 1) * H + col) * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) pre[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(d_ea) + off[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        if (row < nrows) *reinterpret_cast<float*>(reinterpret_cast<char*>(d_ea) + off[r]) = pre[r] + hval(accm[0][0], accx[0][0], r) * inv2;
    }
    if (f.amax != nullptr) {
        if (wave == 0) {
            const float m = max32(lane < TT && lane < nrows ? s_rmax[lane] : 0.0f);
            if (lane == 0 && m > 0.0f) atomic_amax(f.amax, m);
        }
        if (tid == 64) atomic_amax(f.amax + 1, tmax);
    }
}

// One launch per block of the backward pass: workgroups [0, node_tiles) run the node chain (the critical path),
// the rest the filter chain of the same block -- both read d loss / d agg of the block and nothing of each other.
// PREC_H2: the filter chain on the f16 MFMA pipes (W2t / W0t are f16-plane images); the node chain stays fp32 (16-row
// tiles bound by the weight feed, the same bytes in either form)
template <int H, int PREC = PREC_F32>
__global__ __launch_bounds__(2 * H) void block_bwd_kernel(NodeBwd a, int node_tiles, FilterBwd f) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < node_tiles) {
        __builtin_amdgcn_s_setprio(3);
        node_bwd_role<H, PREC == PREC_H2>(a, b, smem);
    } else {
        if constexpr (PREC == PREC_H2) filter_bwd_role_h<H>(f, b - node_tiles, smem);
        else filter_bwd_role<H>(f, b - node_tiles, smem);
    }
}
// first: the chain starts at dh (top block, no gather); last: it stops at dh_0.  filter_rows == 0: no filter role.
int launch_block_bwd(int H, int N, int first, int last, tsd_edges enc, const float* Wf, const float* dagg_in,
                     const float* dh_up, const float* w_lin1_t, const float* w_lin_t, const float* w_lin2_t,
                     const float* x2_prev, float* dx1, float* dh, float* dx2_prev, float* dagg_prev, int filter_rows,
                     tsd_edges enc_u, const float* x1, const float* f0, const float* W2t, const float* W0t, float cutoff,
                     int smooth, float* dWf, float* df0, float* d_ea, hipStream_t st, float* amax_h2, NodeAmax node_amax) {
    NodeBwd a{N, first, last, enc.row_ptr, enc.dst, enc.umap, Wf, dagg_in, dh_up, w_lin1_t, w_lin_t, w_lin2_t, x2_prev,
              dx1, dh, dx2_prev, dagg_prev, node_amax};
    FilterBwd f{};
    f.tiles = (filter_rows + 31) / 32;
    if (f.tiles) {
        f.eu = enc_u;
        f.dagg = dagg_in;
        f.x1 = x1;
        f.f0 = f0;
        f.W2t = W2t;
        f.W0t = W0t;
        f.cutoff = cutoff;
        f.smooth = smooth;
        f.dWf = dWf;
        f.df0 = df0;
        f.d_ea = d_ea;
        f.amax = amax_h2;
    }
    const int node_tiles = (N + TN - 1) / TN;
    if (node_tiles + f.tiles == 0) return TSD_OK;
    // (h2: W2t / W0t are f16-plane images; the top launch of a split-f16 step has no filter role but keeps the node maxima)
    const bool h2 = amax_h2 != nullptr || node_amax.in != nullptr || node_amax.dh != nullptr || node_amax.dx2 != nullptr;
    const size_t lds_n = (size_t)TN * (H + 4) * 4 + 4 * 8 * 4, lds_f = (size_t)(32 * (H + 4) + 32) * 4 + 2 * 32 * sizeof(int);
    const size_t lds_h = (size_t)(32 * ldh_of(H) + 4 * 32) * 4 + 2 * 32 * sizeof(int);
    const size_t lds = std::max(lds_n, h2 ? lds_h : lds_f);
#define TSD_NB(HH)                                                                                              \
    {                                                                                                           \
        static DeviceOnce once, once_h;                                                                         \
        int r = h2 ? allow_lds(block_bwd_kernel<HH, PREC_H2>, lds, once_h) : allow_lds(block_bwd_kernel<HH>, lds, once); \
        if (r) return r;                                                                                        \
        if (h2) hipLaunchKernelGGL((block_bwd_kernel<HH, PREC_H2>), dim3(node_tiles + f.tiles), dim3(2 * HH), lds, st, a, node_tiles, f); \
        else hipLaunchKernelGGL(block_bwd_kernel<HH>, dim3(node_tiles + f.tiles), dim3(2 * HH), lds, st, a, node_tiles, f); \
    }
    if (H == 128) TSD_NB(128) else if (H == 256) TSD_NB(256) else {
        set_error("block_bwd: hidden=%d has no MFMA instance", H);
        return TSD_ERR_INVALID;
    }
#undef TSD_NB
    TSD_LAUNCH_CHECK("block_bwd");
    return TSD_OK;
}

// out[i] = sum_{e in row i} x[dst e] * W[umap e] for every node, by the node role's gather code (tiles of TN nodes):
// the adjoint of the pair product h_i * h_j w.r.t. h in the training step (W = d loss / d product on the undirected
// out list, x = h; common.py:226-229 backwards)
template <int H>
__global__ __launch_bounds__(2 * H) void row_gather_kernel(int N, const int32_t* __restrict__ row_ptr,
                                                           const int32_t* __restrict__ dst,
                                                           const int32_t* __restrict__ umap, const float* __restrict__ W,
                                                           const float* __restrict__ x, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    aggregate_tile<H, true>(row_ptr, dst, umap, W, x, N, blockIdx.x * TN, smem, out);
}
int launch_row_gather(int H, int N, tsd_edges e, const float* W, const float* x, float* out, hipStream_t st) {
    if (N == 0) return TSD_OK;
    const size_t lds = (size_t)TN * (H + 4) * 4;
    const int tiles = (N + TN - 1) / TN;
#define TSD_RG(HH)                                                                                              \
    {                                                                                                           \
        static DeviceOnce once;                                                                                 \
        int r = allow_lds(row_gather_kernel<HH>, lds, once);                                                    \
        if (r) return r;                                                                                        \
        hipLaunchKernelGGL(row_gather_kernel<HH>, dim3(tiles), dim3(2 * HH), lds, st, N, e.row_ptr, e.dst, e.umap, \
                           W, x, out);                                                                          \
    }
    if (H == 64) TSD_RG(64) else if (H == 128) TSD_RG(128) else if (H == 256) TSD_RG(256) else {
        set_error("row_gather: hidden=%d unsupported", H);
        return TSD_ERR_INVALID;
    }
#undef TSD_RG
    TSD_LAUNCH_CHECK("row_gather");
    return TSD_OK;
}

struct ComboStride {  // per-checkpoint strides (checkpoint = blockIdx.x % M)
    size_t w, nh, ea, wf, pre;
    int M;            // checkpoints of the launch, signed: common.hpp wg_item_ckpt / ckpt_grid_m
    int node_stride;  // 1: node tiles are the first workgroups; S > 1 (odd): node tile j is workgroup j * S
    int32_t* range_status;  // PREC_H2 launches: device word for TSD_STATUS_RANGE (or NULL)
};

#ifndef TSD_FILTER_WIDE_MIN
#define TSD_FILTER_WIDE_MIN 1024  // filter tiles (x checkpoints) of a block launch from which they are 64 rows; 0: never
#endif
constexpr int NODE_RUN = 4;  // consecutive node tiles kept on one XCD (one 64-atom graph = 4 tiles)

// TAIL: the instantiation of the launches that carry a third role behind the filter tiles (the pair MLP of the last
// block launch, or the pre-GEMM of the piecewise pair output): a kernel of its own name in a profile, and the plain
// block launches do not carry its code and registers.
// PREC: PREC_F32 = fp32-input MFMA roles; PREC_H2 = the split-f16 roles (SAVE form: the training step's block launches,
// and no pre role: the piecewise pair output computes both halves itself).
// FRB: 32-row blocks per filter tile of the split-f16 filter role (filter_role_h).
// NRB: 16-row blocks per node tile of the split-f16 node role (1: node_role_h; 2: node_role_hw, chip-full launches).
template <int H, bool SAVE, bool TAIL, int PREC = PREC_F32, int FRB = 1, int NRB = 1>
__global__ __launch_bounds__(2 * H) void layer_combo_kernel(ComboNode a, int node_tiles, ComboFilter f,
                                                            ComboStride sd, ComboPre q, FilterSave fsv,
                                                            NodeSave ns TSD_TRACE_ARG) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bx;      // (1-D grid, checkpoint = id % M: common.hpp wg_item_ckpt)
    size_t m_;
    wg_item_ckpt(sd.M, bx, m_);
    {
        const size_t m = m_;
        const size_t wo = m * sd.w, no = m * sd.nh;
        a.Wf += m * sd.wf; a.x1_in += no; a.h_in += no; a.h += no; a.x1_out += no;
        a.lin2_w += wo; a.lin2_b += wo; a.lin_w += wo; a.lin_b += wo;
        if (a.lin1_next_w) a.lin1_next_w += wo;
        f.Wl0 += wo;
        f.edge_attr += m * sd.ea; f.wf += m * sd.wf;
        if (TAIL && q.tiles) {
            q.edge_attr += m * sd.ea; q.w0b += wo; q.b0 += wo;
            if (q.pair) {
                q.w0a += wo; q.w1 += wo; q.b1 += wo; q.w2 += wo; q.b2 += wo;
                q.h += no; q.edge_inv += m * q.inv_stride; q.ready += m * (size_t)node_tiles;
            } else {
                q.out += m * sd.pre;
            }
        }
        if (a.ready) a.ready += m * (size_t)node_tiles;
    }
    // Workgroup -> role.  Small launches (every workgroup resident at once, S = 1) put the node tiles first; large
    // ones (S > 1, odd) spread them through the grid, so that the node role's HBM-bound gather of a big batch and the
    // MFMA-bound filter tiles at least share the chip in time.  (Measured and dropped at batch 100: pairing the node
    // tiles with each other on a CU -- workgroups b and b + 256 share one -- instead of with filter tiles: a node
    // tile gathers 3x slower beside an fp32 MFMA stream, 6 -> 21 us, but the launch stays bound by the CUs that run
    // two filter tiles, 40.4 vs 39.5 us.)
    const int S = sd.node_stride;
    const int b = bx;
    const bool is_node = S > 1 ? (b % S == 0 && b / S < node_tiles) : b < node_tiles;
    const int node_id = S > 1 ? b / S : b;
    const int others_before = S > 1 ? b - min(node_tiles, b / S + 1) : b - node_tiles;
    if (is_node) {
        // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), and the node
        // tiles of one graph read the same x1 rows and -- from both endpoints -- the same filter rows.  Runs of
        // NODE_RUN consecutive tiles therefore go to ONE XCD: node workgroup j runs on XCD (j * S) & 7, a bijection of
        // j & 7 for odd S, so the q-th node workgroup of residue x = j & 7 takes tile ((q / run) * 8 + x) * run + q % run.
        // The tail that does not fill 8 * run keeps the identity order.
        int tile = node_id;
        const int full = node_tiles / (8 * NODE_RUN) * (8 * NODE_RUN);
        if (tile < full) {
            const int x = tile & 7, qq = tile >> 3;
            tile = ((qq / NODE_RUN) * 8 + x) * NODE_RUN + qq % NODE_RUN;
        }
        // the node chain is the critical path of the launch: its waves get issue priority over the filter waves
        // they share a SIMD with
        __builtin_amdgcn_s_setprio(3);
        TSD_TRACE_REAL(24);
        if constexpr (PREC == PREC_H2 && NRB > 1) node_role_hw<H, NRB>(a, tile, smem, sd.range_status);
        else if constexpr (PREC == PREC_H2) node_role_h<H, SAVE>(a, tile, smem, sd.range_status, ns TSD_TRACE_PASS);
        else node_role<H, SAVE>(a, tile, smem, ns TSD_TRACE_PASS);
        TSD_TRACE_REAL(25);
    } else {
        const int item = others_before;
        if (item >= f.tiles) {
            if constexpr (!TAIL) return;
            else if (q.pair) {
                // workgroups b and b + 256 of a launch share a CU (measured; a speed assumption only): these pair tiles
                // sit beside the node tiles
                const bool defer = TSD_PAIR_DEFER && b >= 256 && b - 256 < node_tiles;
                if constexpr (PREC == PREC_H2) {
                    if constexpr (H == 256) {
                        if (q.rows == 2 * T) pair_role_h<H, false, false, 2>(q, item - f.tiles, node_tiles, smem, defer, sd.range_status);
                        else pair_role_h<H>(q, item - f.tiles, node_tiles, smem, defer, sd.range_status);
                    } else {
                        pair_role_h<H>(q, item - f.tiles, node_tiles, smem, defer, sd.range_status);
                    }
                }
                else pair_role<H>(q, item - f.tiles, node_tiles, smem, defer);
            } else if constexpr (PREC == PREC_F32) {
                pre_role<H>(q, item - f.tiles, smem);
            }
            return;
        }
        TSD_TRACE_REAL(24);
        if constexpr (PREC == PREC_H2) filter_role_h<H, FRB, SAVE>(f, item, smem, sd.range_status, fsv TSD_TRACE_PASS);
        else filter_role<H, SAVE>(f, item, smem, fsv TSD_TRACE_PASS);
        TSD_TRACE_REAL(25);
    }
}


int filter_tiles_per_layer(int capacity_u);

// The stand-alone pair output (tsd_pair_output's place in a forward that does not run it inside the last block
// launch) on the f16 MFMA pipes: pair_role_h without the wait.
// (inference form: three workgroups of 8 waves per CU = 6 waves per SIMD = 80 VGPRs; the compiler's free allocation is 84, the
// cap costs no spill -- tools/check_regs.py holds it to 80 registers and no scratch)
// RB = 2: 64-row tiles for launches many rounds deep (configs[4]: 64 k tiles) -- 128 VGPRs under the cap (132 free), two
// workgroups per CU, half the weight stream per row: configs[4] 18.64 -> 18.30 ms/step.
template <int H, bool SAVE = false, int RB = 1>
__global__ __launch_bounds__(2 * H) __attribute__((amdgpu_waves_per_eu(SAVE || H != 256 ? 1 : (RB > 1 ? 4 : 6)))) void pair_output_h_kernel(ComboPre q, size_t wstride, size_t h_stride, size_t ea_stride,
                                                              int32_t* range_status, PairSave sv, int M) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int tile;
    size_t m;
    wg_item_ckpt(M, tile, m);
    const size_t wo = m * wstride;
    q.edge_attr += m * ea_stride; q.w0b += wo; q.b0 += wo;
    q.w0a += wo; q.w1 += wo; q.b1 += wo; q.w2 += wo; q.b2 += wo;
    q.h += m * h_stride; q.edge_inv += m * q.inv_stride;
    pair_role_h<H, SAVE, (!SAVE && RB == 1), RB>(q, tile, 0, smem, false, range_status, sv);
}
int launch_pair_output_h(const tsd_model_cfg& c, const float* W16, int capacity, tsd_edges e, const float* h,
                         const float* edge_attr, const int32_t* attr_row, float* edge_inv, int M, size_t h_stride,
                         size_t ea_stride, size_t inv_stride, hipStream_t st, bool folded, int32_t* range_status,
                         const PairSave* save, bool narrow) {
    const WeightLayout L = weight_layout(c);
    const size_t H = c.hidden;
    const int tiles = (capacity + T - 1) / T;
    if (tiles == 0) return TSD_OK;
#ifndef TSD_PAIR_OUT_WIDE_MIN
#define TSD_PAIR_OUT_WIDE_MIN 4096  // 32-row pair tiles (x checkpoints) of the stand-alone launch from which they are 64 rows; 0: never
#endif
    const bool wide = !save && !narrow && c.hidden == 256 && TSD_PAIR_OUT_WIDE_MIN > 0 && (long)tiles * M >= TSD_PAIR_OUT_WIDE_MIN;
    ComboPre q{};
    q.tiles = tiles;
    q.e = e;
    q.edge_attr = edge_attr;
    q.attr_row = attr_row;
    q.w0b = folded ? W16 + L.out_w0f : W16 + L.out_w0 + H * H;
    q.b0 = W16 + (folded ? L.out_b0f : L.out_b0);
    q.pair = 1;
    q.w0a = W16 + L.out_w0;
    q.w1 = W16 + L.out_w1;
    q.b1 = W16 + L.out_b1;
    q.w2 = W16 + L.out_w2;
    q.b2 = W16 + L.out_b2;
    q.h = h;
    q.edge_inv = edge_inv;
    q.ready = nullptr;
    q.status = nullptr;
    q.inv_stride = inv_stride;
    const size_t lds = (size_t)(T * ldh_of(c.hidden) + (c.hidden / 64) * T + 3 * T) * 4;
    if (save && (folded || M != 1)) {
        set_error("internal: the saving pair output takes the unfolded first layer of one checkpoint");
        return TSD_ERR_INVALID;
    }
    if (wide) {
        static DeviceOnce once_w;
        const int tiles2 = (capacity + 2 * T - 1) / (2 * T);
        q.tiles = tiles2;
        const size_t lds2 = (size_t)(2 * T * ldh_of(256) + (256 / 64) * 2 * T + 3 * 2 * T) * 4;
        int r = allow_lds(pair_output_h_kernel<256, false, 2>, lds2, once_w);
        if (r) return r;
        hipLaunchKernelGGL((pair_output_h_kernel<256, false, 2>), dim3(tiles2 * M), dim3(512), lds2, st, q, L.total, h_stride,
                           ea_stride, range_status, PairSave{}, ckpt_grid_m(M, tiles2));
        TSD_LAUNCH_CHECK("pair_output_h (64-row tiles)");
        return TSD_OK;
    }
#define TSD_POH(HH)                                                                                              \
    {                                                                                                            \
        static DeviceOnce once, once_s;                                                                          \
        int r = save ? allow_lds(pair_output_h_kernel<HH, true>, lds, once_s) : allow_lds(pair_output_h_kernel<HH>, lds, once); \
        if (r) return r;                                                                                         \
        if (save) hipLaunchKernelGGL((pair_output_h_kernel<HH, true>), dim3(tiles * M), dim3(2 * HH), lds, st, q, L.total, h_stride, \
                           ea_stride, range_status, *save, ckpt_grid_m(M, tiles));                               \
        else hipLaunchKernelGGL(pair_output_h_kernel<HH>, dim3(tiles * M), dim3(2 * HH), lds, st, q, L.total, h_stride, \
                           ea_stride, range_status, PairSave{}, ckpt_grid_m(M, tiles));                          \
    }
    switch (c.hidden) {
        case 64: TSD_POH(64) break;
        case 128: TSD_POH(128) break;
        case 256: TSD_POH(256) break;
        default: set_error("hidden=%d unsupported (64/128/256)", c.hidden); return TSD_ERR_INVALID;
    }
#undef TSD_POH
    TSD_LAUNCH_CHECK("pair_output_h");
    return TSD_OK;
}

static inline size_t lds_combo(int H, int prec, int frb = 1, int nrb = 1) {
    const int ld = prec == PREC_H2 ? ldh_of(H) : H + 4;  // floats per tile row (two f16 planes of H + 8 = H + 8 floats)
    const size_t node = (size_t)TN * nrb * ld * 4 + (prec == PREC_H2 ? (size_t)2 * H * 4 : 0);  // (+ the transposed form's biases)
    const size_t filt = (size_t)(T * ld + T) * 4 * frb + (prec == PREC_H2 ? (size_t)2 * H * 4 : 0);  // (+ the transposed form's biases)
    const size_t pair = (size_t)(T * ld + (H / 64) * T + 3 * T) * 4 * (prec == PREC_H2 ? frb : 1);  // pair role (H >= 64; 64-row tiles with the filter role's)
    return node > filt ? (node > pair ? node : pair) : (filt > pair ? filt : pair);
}

int filter_tiles_per_layer(int capacity_u) { return (capacity_u + T - 1) / T; }

// =================================================================================================
// THE INTERACTION BLOCKS AND THE PAIR MLP AS ONE LAUNCH (split-f16 arithmetic, small batches, one checkpoint).
//
// With the GEMMs on the f16 MFMA pipes a per-block launch at batch 100 is 25 us of which ~7 us are the launch itself
// (its 508 workgroups are dispatched at ~8 ns each and the kernel boundary drains and refills the chip) and ~17 us the
// node chain, a latency chain that occupies 100 of the 256 CUs.  Here the L launches behind the embedding launch become
// roles of one grid, ordered by blockIdx (= dispatch order):
//   [node workgroups: ONE per node tile, persistent over all L blocks]
//   [filter tiles of blocks 1 .. L-1, block-major]  [pair tiles]      (the first n_node pair tiles sit at positions
//    256 .. 256 + n_node - 1, i.e. on the node workgroups' CUs, where they sleep until the node chain is done)
// (the embedding launch before it still carries the filters of block 0 and the directed -> undirected map), and the
// launch boundaries become hand-offs (cdna_hip_programming.md Guideline 16: 16-byte write-through payload, every
// storing wave drains, ONE lane publishes; relaxed agent-scope polls by one wave; consumers read handed-off rows with
// sc1 loads, and no buffer is written twice within the launch -- an sc1 load is served by the XCD's L2, which would
// keep the line of an earlier read):
//   filter_done[l][f] = epoch once filter tile f of block l >= 1 has its rows in memory
//                -> the node workgroups whose edges use rows of that tile (a graph's pairs are contiguous in the
//                   undirected list: a node tile waits for the ~5 filter tiles [min umap / 32, max umap / 32] of its own
//                   edges, not for the slowest of the layer's 408)
//   node_done[t] = epoch * 64 + (blocks tile t has published x1 / h for)
//                -> the node tiles that hold atoms of the same graphs (block l + 1 gathers x1 rows of the whole graph),
//                   pair tiles (final h)
// All words are monotonic: a counter's target is epoch * count, where epoch = the number of this launch since the host
// zeroed the block (once per run / per stand-alone forward), derived from the device-side step counter of the sampling
// loop (it advances between two launches; graph replay freezes kernel arguments): nothing is zeroed per launch.
// Waits are bounded (TSD_STATUS_INTERNAL instead of a hang).  No workgroup waits for one that is dispatched after it,
// except the node workgroups among themselves and for filter tiles -- they are all resident (node tiles <= 256 is a
// launch condition; filter tiles never wait), so in-order dispatch is enough for progress.
// Results are bit-identical to the launch-per-block split-f16 forward (same GEMMs, same gather order).
//
// REGISTERS.  The kernel must fit 128 VGPRs (two resident workgroups per CU) WITHOUT an occupancy cap: under
// __launch_bounds__(2 H, 4) the compiler spills and splits live ranges beside the asm-issued ring loads, whose
// destination registers it believes written at the statement (measured: wrong results in one build, 3x slower tiles in
// another).  tools/check_async_loads.py proves the absence of such accesses for the compiled binary.
// =================================================================================================
#ifdef TSD_MEGA_TRACE  // (variant builds: tools/trace_mega.py) 4 u64 per workgroup: role, start, end, aux
__device__ unsigned long long g_mega_trace[8192 * 4];
#define TSD_MEGA_T(slot, val)                                                                           \
    do {                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_mega_trace[(size_t)blockIdx.x * 4 + (slot)] = (val); \
    } while (0)
extern "C" int tsd_debug_mega_trace(void* host_buf) {  // copies the buffer out and clears it
    hipError_t e = hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_mega_trace), sizeof(g_mega_trace));
    if (e != hipSuccess) return (int)e;
    static unsigned long long zeros[8192 * 4];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_mega_trace), zeros, sizeof(zeros));
}
// phase stamps of the node workgroups: [tile][block][phase] (wall clock, 100 MHz)
#ifndef TSD_MEGA_P_MASK
#define TSD_MEGA_P_MASK 255  // (every stamp costs scalar registers: a build that must stay at 128 VGPRs takes fewer)
#endif
__device__ unsigned long long g_mega_phase[256 * 8 * 8];
#define TSD_MEGA_P(tile, l, ph)                                                                                 \
    do {                                                                                                        \
        if (TSD_MEGA_P_MASK >> (ph) & 1 && threadIdx.x == 0 && (tile) < 256 && (l) < 8) g_mega_phase[((size_t)(tile) * 8 + (l)) * 8 + (ph)] = wall_clock64(); \
    } while (0)
extern "C" int tsd_debug_mega_phase(void* host_buf) {
    return (int)hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_mega_phase), sizeof(g_mega_phase));
}
#else
#define TSD_MEGA_T(slot, val)
#define TSD_MEGA_P(tile, l, ph)
#endif
struct MegaCtl {  // int32 words in the forward workspace
    // ZERO: a word nothing ever writes (epoch source of a stand-alone forward); NODE0: one word per node tile (<= 256);
    // FILTER0: one word per filter tile of blocks 0 .. L-1 (block-major; block 0's are never written here)
    static constexpr int ZERO = 32, NODE0 = 64, FILTER0 = 64 + 256;
};
struct MegaArgs {
    // roles' grid ranges
    int n_node, n_filter, n_pair, tiles_per_layer;
    int filter_rows;           // pairs per filter tile: 32, or 64 (hidden 256, a block's tiles fill the chip)
    int pair_rows;             // pairs per pair tile: likewise
    // checkpoints of this launch (an ensemble's forwards in groups whose node workgroups fit the resident half of the chip):
    // workgroup id = item * G + g, every pointer below is checkpoint 0's of the group and moves by g * its stride
    int G, M, per_group;       // G checkpoints per group, M in all (the grid holds ceil(M / G) groups of per_group workgroups each, group-major)
    size_t s_w, s_nh, s_x1m, s_wf, s_ea, s_inv;  // floats: weight arena, node arrays, x1m block, filter arena, attribute rows, edge_inv
    int s_ctl;                 // int32 words of one checkpoint's control block
    int L, N;
    int half_slots;            // half of the device's resident-workgroup slots for this kernel (mega_slots() / 2: 256 on a whole MI355X)
    const int32_t* epoch_src;  // device word: epoch = *epoch_src + epoch_bias (>= 1, + 1 per launch since the block was zeroed)
    int epoch_bias;
    int32_t* ctl;
    int32_t* status;           // TSD_STATUS_INTERNAL / TSD_STATUS_RANGE
    // node role
    const int32_t *graph_ptr, *node_graph;
    const int32_t *row_ptr, *dst, *umap;
    const float* W;            // f16-plane arena
    size_t layer0, layer_stride, o_lin1, o_lin2_w, o_lin2_b, o_lin_w, o_lin_b;
    const float *z, *x1_0;
    float *x1m, *h;            // x1m: [L - 1][N, H], x1 of block l + 1 in slot l (every buffer is written ONCE per launch)
    size_t x1_stride;
    const float* wf;           // filters, slot = block
    size_t wf_layer_stride;
    // filter role (blocks 1 .. L-1) and pair role
    ComboFilter f;
    ComboPre q;
};

// Rows per filter tile of the one-launch forward (MegaArgs.filter_rows: 32 or 64, hidden 256).  At batch 100 the launch is
// paced by its filter tiles -- 2448 of 32 rows on the ~312 slots the node workgroups and the parked pair tiles leave them
// run until 132 us of a 157-us launch, and every block's node chain waits for the last tiles of its layer (tools/trace_mega.py)
// -- and a weight fragment that feeds two row blocks is the cheaper tile per row, as in the chip-full launches: with 64-row
// tiles the last filter tile ends at 111 us, the launch at 150 (traced builds); 0.1884 -> 0.1876 / 0.1889 -> 0.1869 ms/step
// untraced.  Half-empty launches (50 graphs: +1.2 %) keep the shorter tiles: 64 rows from TSD_MEGA_WIDE_MIN 32-row tiles per
// block on.
#ifndef TSD_MEGA_WIDE_MIN
#define TSD_MEGA_WIDE_MIN 384  // (0: never)
#endif
constexpr unsigned MEGA_SPIN_LIMIT = 4000000u;
#ifndef TSD_MEGA_XLDS
#define TSD_MEGA_XLDS 1  // the node workgroups gather x from an LDS copy of their graphs' rows (aggregate_tile_xl); 0: from L2
#endif
#ifndef TSD_MEGA_POLL_SLEEP
#define TSD_MEGA_POLL_SLEEP 12  // x64 cycles between two polls (~0.3 us): a hundred waves polling one word at full rate throttle the L2 channel it lives in
#endif
constexpr int MEGA_POLL_SLEEP = TSD_MEGA_POLL_SLEEP;

// one wave: wait until *p >= target (monotonic word); false after the bound
// (a launch in which one wait has given up unwinds quickly: every MEGA_ABORT_POLLS polls a waiting wave looks at the
// status word, and a wait that finds TSD_STATUS_INTERNAL there gives up as well -- the results of such a launch are void,
// the host reruns the forward as one launch per block)
constexpr unsigned MEGA_ABORT_POLLS = 256u;
__device__ __forceinline__ bool mega_wait_failed(unsigned spins, int32_t* status) {
    if (spins > MEGA_SPIN_LIMIT) {
        if ((threadIdx.x & 63) == 0) atomicOr(status, TSD_STATUS_INTERNAL);
        return true;
    }
    return (spins & (MEGA_ABORT_POLLS - 1)) == MEGA_ABORT_POLLS - 1 &&
           (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & TSD_STATUS_INTERNAL) != 0;
}
__device__ __forceinline__ bool mega_wait_ge(const int32_t* p, int target, int32_t* status) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        if (mega_wait_failed(spins, status)) {
            return false;
        }
        __builtin_amdgcn_s_sleep(MEGA_POLL_SLEEP);
    }
}
// one wave: wait until p[t] >= target for every t in [lo, hi]
__device__ __forceinline__ bool mega_wait_range_ge(const int32_t* p, int lo, int hi, int target, int32_t* status) {
    const int lane = threadIdx.x & 63;
    for (int t0 = lo; t0 <= hi; t0 += 64) {
        const int t = t0 + lane;
        for (unsigned spins = 0;; ++spins) {
            const int v = t <= hi ? __hip_atomic_load(p + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
            if (__all(v >= target)) break;
            if (mega_wait_failed(spins, status)) return false;
            __builtin_amdgcn_s_sleep(MEGA_POLL_SLEEP);
        }
    }
    return true;
}
// every storing wave drains, the workgroup meets, ONE lane counts (Guideline 16 R1, counter form)
__device__ __forceinline__ void mega_arrive(int32_t* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The node workgroup of tile `tile`: all L blocks (node_role_h's chain per block, h kept in registers, x1 published).
// Tile rows MEGA_TR: 16 as in the per-block launches.  8 (one row per wave in the gather: two dependent round trips for
// ~16 edges instead of four for ~32; the MFMA's other rows are free) is bit-identical and was measured SLOWER at batch
// 100: 0.254 vs 0.223 ms/step -- 200 node workgroups leave the filter tiles 312 of the 512 slots, a layer of 408 tiles
// then takes 1.3 rounds and the node chain waits for its filters (tools/trace_mega.py: 188 vs 157 us).
#ifndef TSD_MEGA_TR
#define TSD_MEGA_TR 16  // (8: built and measured, see below)
#endif
constexpr int MEGA_TR = TSD_MEGA_TR;
template <int H>
__device__ __forceinline__ void node_persist_h(const MegaArgs& A, int tile, int epoch, float* smem) {
    constexpr int LDH = ldh_of(H), LDA = H + 4;
    constexpr int NT = 2 * H, CB16 = 2, C4 = H / 4;
    const Planes pl = planes_at(smem, TN, LDH);
    float* xst = smem;  // the finished x1 tile as fp32 rows (over the planes: TN (H + 4) <= TN (H + 8) floats)
    constexpr int TR = MEGA_TR < 2 * H / 64 ? 2 * H / 64 : MEGA_TR;  // (at least one row per wave)
    const int n0 = tile * TR;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));  // (opaque: lane arithmetic is not hoisted above the role branch of the one-launch kernel)
    const int tid = tid_;
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, l15 = lane & 15;
    const int col0 = wave * 32;
    const int nrows = min(TR, A.N - n0);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float amax = 0.0f;
    int32_t* node_done = A.ctl + MegaCtl::NODE0;
    // transposed accumulators (as node_role_h): lane = tile row l15, channels 4 q .. 4 q + 3 per 16-column block; the
    // block's two bias vectors in LDS behind the planes and the x copy, x1 published straight from the registers
#ifndef TSD_MEGA_NODE_TRANS
#define TSD_MEGA_NODE_TRANS 1
#endif
#ifndef TSD_MEGA_DIRECT_X1
#define TSD_MEGA_DIRECT_X1 1
#endif
    constexpr bool TRN = TSD_FILTER_TRANS != 0 && TSD_MEGA_NODE_TRANS != 0;
    constexpr int XS_F = (H == 256 && TR == TN && TSD_MEGA_XLDS) ? XS_ROWS * 256 : 0;
    float* s_bias = smem + TN * LDH + XS_F;  // [2][H]
    // the node tiles that hold atoms of the graphs this tile's atoms belong to
    const int g_first = A.node_graph[n0], g_last = A.node_graph[n0 + nrows - 1];
    // the atoms of those graphs (workgroup-uniform values the compiler would keep in vector registers)
    const int x_base = __builtin_amdgcn_readfirstlane(A.graph_ptr[g_first]);
    const int x_nloc = __builtin_amdgcn_readfirstlane(A.graph_ptr[g_last + 1]) - x_base;
    const int t_lo = x_base / TR, t_hi = (x_base + x_nloc - 1) / TR;
    // residual input of block 0: the pos-independent node embedding z
    float h_res[CB16][4];
#pragma unroll
    for (int cb = 0; cb < CB16; ++cb) {
        if constexpr (TRN) {
            const f32x4 z4 = l15 < nrows ? *reinterpret_cast<const f32x4*>(A.z + (size_t)(n0 + l15) * H + col0 + cb * 16 + q * 4) : zero4;
#pragma unroll
            for (int r = 0; r < 4; ++r) h_res[cb][r] = z4[r];
        } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = q * 4 + r;
            h_res[cb][r] = row < nrows ? A.z[(size_t)(n0 + row) * H + col0 + cb * 16 + l15] : 0.0f;
        }
        }
    }
    // the filter tiles this tile's edges read rows of (geometry only: the same range in every block)
    int f_lo = 0x7fffffff, f_hi = -1;
    if (wave == 0) {
        const int Et0 = A.row_ptr[n0], Et1 = A.row_ptr[min(n0 + TR, A.N)];
        int lo = 0x7fffffff, hn = -1;
        for (int e = Et0 + lane; e < Et1; e += 64) {
            const int u = A.umap[e];
            lo = min(lo, u);
            hn = max(hn, u);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off));
            hn = max(hn, __shfl_xor(hn, off));
        }
        if (hn >= 0) {
            f_lo = lo / A.filter_rows;
            f_hi = hn / A.filter_rows;
        }
    }
    bool use_xl = false;
    if constexpr (H == 256 && TR == TN && TSD_MEGA_XLDS) use_xl = x_nloc <= XS_ROWS;  // (workgroup-uniform)
    for (int l = 0; l < A.L; ++l) {
        TSD_MEGA_P(tile, l, 0);
        const float* Wl = A.W + A.layer0 + (size_t)l * A.layer_stride;
        const float* x_in = l == 0 ? A.x1_0 : A.x1m + (size_t)(l - 1) * A.x1_stride;
        float* x_out = A.x1m + (size_t)l * A.x1_stride;
        const float* wf_l = A.wf + (size_t)l * A.wf_layer_stride;
        const bool last = l + 1 == A.L;
        HRing<CB16, HRING16_R> rg;
        f32x4 accm[CB16], accx[CB16];
        if (use_xl) {
            // the filters of this block first (complete long before the neighbours' x rows as a rule): the first
            // edges' filter rows of every wave are in flight while wave 0 waits for the node tiles
            if (l > 0 && wave == 0 && f_hi >= 0)
                mega_wait_range_ge(A.ctl + MegaCtl::FILTER0 + (size_t)l * A.tiles_per_layer, f_lo, f_hi, epoch, A.status);
            __syncthreads();
            if constexpr (H == 256)
                xl_gather<H, 2 * H / 64, TR>(A.row_ptr, A.dst, A.umap, wf_l, x_in, x_base, x_nloc, smem + TN * LDH, A.N, n0, smem,
                                             amax, [&]() {
                                                 if (l > 0 && wave == 0) mega_wait_range_ge(node_done, t_lo, t_hi, epoch * 64 + l, A.status);
                                                 __syncthreads();
                                                 TSD_MEGA_P(tile, l, 1);
                                             });
        } else {
            if (l > 0 && wave == 0) {
#ifndef TSD_MEGA_NOWAIT_LAYER  // (timing experiments only: wrong results)
                if (f_hi >= 0)
                    mega_wait_range_ge(A.ctl + MegaCtl::FILTER0 + (size_t)l * A.tiles_per_layer, f_lo, f_hi, epoch, A.status);
#endif
#ifndef TSD_MEGA_NOWAIT_NODE
                mega_wait_range_ge(node_done, t_lo, t_hi, epoch * 64 + l, A.status);
#endif
            }
            __syncthreads();
            TSD_MEGA_P(tile, l, 1);
            if constexpr (TR < TN) {  // the MFMA's rows past the tile: the gather does not write them and the x1 staging of
                // the previous block lies over them (fp32 bits read as f16 may be inf): zero, so that they stay finite
                for (int idx = tid; idx < (TN - TR) * (LDH / 2); idx += NT) {
                    reinterpret_cast<uint32_t*>(pl.hi + TR * LDH)[idx] = 0u;
                    reinterpret_cast<uint32_t*>(pl.lo + TR * LDH)[idx] = 0u;
                }
            }
#ifdef TSD_MEGA_PLAIN_GATHER
            constexpr bool kSc1 = false;
#else
            constexpr bool kSc1 = true;
#endif
            aggregate_tile<H, false, 2 * H / 64, 8, true, kSc1, TR>(A.row_ptr, A.dst, A.umap, wf_l, x_in, A.N, n0, smem, nullptr, &amax);
        }
        TSD_MEGA_P(tile, l, 2);
        float b_lin2[CB16], b_lin[CB16];  // (requested behind the gather: they arrive under the first GEMM)
        float bl2 = 0.0f, bl = 0.0f;      // (TRN: one channel's two biases per thread, to LDS behind the first GEMM)
        if constexpr (TRN) {
            if (tid < H) {
                bl2 = Wl[A.o_lin2_b + tid];
                bl = Wl[A.o_lin_b + tid];
            }
        } else {
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            b_lin2[cb] = Wl[A.o_lin2_b + col0 + cb * 16 + l15];
            b_lin[cb] = Wl[A.o_lin_b + col0 + cb * 16 + l15];
        }
        }
        hgemm16_ring_start<CB16, H>(rg, Wl + A.o_lin2_w, H, col0);
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) accm[cb] = accx[cb] = zero4;
        hgemm16_ring_run<CB16, H, TRN>(rg, pl, LDH, accm, accx);
        if constexpr (TRN) {
            if (tid < H) {  // (the previous block's readers are two barriers back)
                s_bias[tid] = bl2;
                s_bias[H + tid] = bl;
            }
        }
        hgemm16_ring_start<CB16, H>(rg, Wl + A.o_lin_w, H, col0);
        __syncthreads();
        TSD_MEGA_P(tile, l, 3);
        if constexpr (TRN) {
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const int c = col0 + cb * 16 + q * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(s_bias + c);
                f32x4 y4;
#pragma unroll
                for (int r = 0; r < 4; ++r) y4[r] = sspf(hval4(accm[cb], accx[cb], r) + b[r]);
                planes_store4(pl, l15 * LDH + c, y4, amax);
            }
        } else
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                planes_store1(pl, (q * 4 + r) * LDH + col, sspf(hval4(accm[cb], accx[cb], r) + b_lin2[cb]), amax);
        }
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) accm[cb] = accx[cb] = zero4;
        TSD_MEGA_P(tile, l, 4);
        hgemm16_ring_run<CB16, H, TRN>(rg, pl, LDH, accm, accx);
        if (!last) hgemm16_ring_start<CB16, H>(rg, Wl + A.layer_stride + A.o_lin1, H, col0);
        __syncthreads();
        TSD_MEGA_P(tile, l, 5);
        if constexpr (TRN) {
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const int c = col0 + cb * 16 + q * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(s_bias + H + c);
                f32x4 hn = zero4;
                if (l15 < nrows) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) hn[r] = h_res[cb][r] + (hval4(accm[cb], accx[cb], r) + b[r]);
                    if (last)  // the final node states: write-through, the pair tiles read them from other CUs
                        store_stream16(A.h + (size_t)(n0 + l15) * H + c, hn);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) h_res[cb][r] = hn[r];
                if (!last) planes_store4(pl, l15 * LDH + c, hn, amax);
            }
        } else
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = q * 4 + r;
                float hn = 0.0f;
                if (row < nrows) {
                    hn = h_res[cb][r] + (hval4(accm[cb], accx[cb], r) + b_lin[cb]);
                    if (last)  // the final node states: write-through, the pair tiles read them from other CUs
                        __hip_atomic_store(A.h + (size_t)(n0 + row) * H + col, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                h_res[cb][r] = hn;
                if (!last) planes_store1(pl, row * LDH + col, hn, amax);
            }
        }
        if (last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(node_done + tile, epoch * 64 + A.L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) accm[cb] = accx[cb] = zero4;
        hgemm16_ring_run<CB16, H, TRN>(rg, pl, LDH, accm, accx);
        if constexpr (TRN && TSD_MEGA_DIRECT_X1) {   // a lane holds 16 consecutive bytes of row l15: write-through stores straight from the
                               // accumulators (no fp32 staging tile, one barrier less per block)
            TSD_MEGA_P(tile, l, 6);
            if (l15 < nrows) {
#pragma unroll
                for (int cb = 0; cb < CB16; ++cb) {
                    f32x4 x4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) x4[r] = hval4(accm[cb], accx[cb], r);
                    store_stream16(x_out + (size_t)(n0 + l15) * H + col0 + cb * 16 + q * 4, x4);
                }
            }
        } else {
        __syncthreads();  // every wave is done reading the planes: the x1 tile goes over them as fp32 rows
        TSD_MEGA_P(tile, l, 6);
#pragma unroll
        for (int cb = 0; cb < CB16; ++cb) {
            const int col = col0 + cb * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (TRN) xst[l15 * LDA + col0 + cb * 16 + q * 4 + r] = hval4(accm[cb], accx[cb], r);
                else xst[(q * 4 + r) * LDA + col] = hval4(accm[cb], accx[cb], r);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < nrows * C4; idx += NT) {  // whole 1-KiB rows, write-through
            const int r = idx / C4, c4 = idx % C4;
            store_stream16(x_out + (size_t)(n0 + r) * H + c4 * 4, *reinterpret_cast<const f32x4*>(xst + r * LDA + c4 * 4));
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // (also: the staging rows are free for the next block's planes)
        if (tid == 0) __hip_atomic_store(node_done + tile, epoch * 64 + l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        TSD_MEGA_P(tile, l, 7);
    }
    range_report(amax, A.status);
}


template <int H>
__global__ __launch_bounds__(2 * H) void forward_mega_kernel(MegaArgs A_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    TSD_MEGA_T(1, wall_clock64());
    MegaArgs A = A_;
    int b = blockIdx.x;
    if (A.M > 1) {
        // An ensemble: groups of G checkpoints, a group's checkpoints interleaved -- one checkpoint's workgroups on the XCDs
        // id % G (common.hpp wg_item_ckpt).  The groups are STAGGERED in the grid,
        //   [N0 F0] [N1 F1 P0] [N2 F2 P1] ... [N(K-1) F(K-1) P(K-2)] [P(K-1)]     (N node workgroups, F filter tiles, P pair tiles)
        // so that group k + 1's node chain starts while group k's runs its last blocks and its pair tiles -- which sleep until
        // that chain is done -- do not hold the slots of the next group's filter tiles.  A workgroup still waits only for
        // workgroups of its own checkpoint, which precede it in the grid or never wait (F(k) follows N(k) directly, and N(k)
        // needs nothing behind F(k)): in-order dispatch is enough for progress even when the node workgroups of two groups
        // fill every slot -- N(k) then simply finishes first.  Round 5: 8 checkpoints at batch 100 1.218 -> 1.199 ms/step,
        // 4: 0.628 -> 0.616, 4 x 200 graphs 1.163 -> 1.147 against group after group; groups of half the size (two node
        // sets = half of the slots) lose: 1.237.
        unsigned id = blockIdx.x, grp, j;
        int base;
        const unsigned sN = (unsigned)A.n_node * A.G, sF = (unsigned)A.n_filter * A.G, sP = (unsigned)A.n_pair * A.G;
#ifndef TSD_MEGA_STAGGER
#define TSD_MEGA_STAGGER 1  // 0 (A/B variant builds): group after group
#endif
        if (TSD_MEGA_STAGGER) {
            const unsigned K = ((unsigned)A.M + A.G - 1) / (unsigned)A.G, blk = sN + sF + sP;
            if (id < sN + sF) {
                grp = 0;
                if (id < sN) { j = id; base = 0; } else { j = id - sN; base = A.n_node; }
            } else {
                id -= sN + sF;
                const unsigned k = id / blk + 1, r = id % blk;
                if (k < K) {
                    if (r < sN) { grp = k; j = r; base = 0; }
                    else if (r < sN + sF) { grp = k; j = r - sN; base = A.n_node; }
                    else { grp = k - 1; j = r - sN - sF; base = A.n_node + A.n_filter; }
                } else {
                    grp = K - 1; j = id - (K - 1) * blk; base = A.n_node + A.n_filter;
                }
            }
        } else {
            grp = id / (unsigned)A.per_group;
            j = id % (unsigned)A.per_group;
            base = 0;
        }
        const size_t g = (size_t)grp * A.G + j % (unsigned)A.G;
        b = base + (int)(j / (unsigned)A.G);
        if (g >= (size_t)A.M) return;   // (the last group of an ensemble that is no multiple of G)
        const size_t ow = g * A.s_w, on = g * A.s_nh, owf = g * A.s_wf, oea = g * A.s_ea;
        A.ctl += g * (size_t)A.s_ctl;
        A.W += ow; A.z += on; A.x1_0 += on; A.x1m += g * A.s_x1m; A.h += on; A.wf += owf;
        A.f.Wl0 += ow; A.f.edge_attr += oea; A.f.wf += owf;
        A.q.edge_attr += oea; A.q.w0b += ow; A.q.b0 += ow; A.q.w0a += ow; A.q.w1 += ow; A.q.b1 += ow; A.q.w2 += ow; A.q.b2 += ow;
        A.q.h += on; A.q.edge_inv += g * A.s_inv; A.q.ready += g * (size_t)A.s_ctl;
        A.half_slots = 0;      // (no parked pair tiles: positions in the grid say nothing about CUs here)
    }
    const int epoch =  // (wave-uniform: kept in an SGPR)
        __builtin_amdgcn_readfirstlane(__hip_atomic_load(A.epoch_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + A.epoch_bias;
    if (b < A.n_node) {
        __builtin_amdgcn_s_setprio(3);
        node_persist_h<H>(A, b, epoch, smem);
        TSD_MEGA_T(0, 2);
        TSD_MEGA_T(2, wall_clock64());
        return;
    }
    b -= A.n_node;
    // Workgroups b and b + 256 of a launch share a CU (measured; a speed assumption only).  The node workgroups took the
    // first slot of n_node CUs; whatever is dispatched at positions 256 .. 256 + n_node - 1 becomes their neighbour.  A
    // filter tile there costs the node chain -- the critical path of the launch -- a third of its speed (17.5 us per
    // block alone, 23 us beside filter tiles: gather and epilogues share the CU's L1 and issue slots).  So those
    // positions go to PAIR tiles, which only sleep until the node chain has finished (their node-independent GEMM
    // deferred behind the wait): the node workgroups get their CUs to themselves.
#ifndef TSD_MEGA_PARK
#define TSD_MEGA_PARK 1
#endif
    // (only while the node workgroups take at most half of the CUs: every parked tile is a slot the filter tiles lose;
    // measured at batch 100: 0.2184 vs 0.2197 ms/step -- the node chain is slowed by the memory system's load, which
    // stays, far more than by its CU neighbour)
    const bool park = TSD_MEGA_PARK != 0 && 2 * A.n_node <= A.half_slots;
    const int first_filters = park ? min(A.n_filter, A.half_slots - A.n_node) : A.n_filter;
    const int parked = park ? min(A.n_pair, A.n_node) : 0;
    int filter_item = -1, pair_item = -1;
    if (b < first_filters) filter_item = b;
    else if (b < first_filters + parked) pair_item = b - first_filters;
    else if (b < A.n_filter + parked) filter_item = b - parked;
    else if (b < A.n_filter + A.n_pair) pair_item = b - A.n_filter;
    if (filter_item >= 0) {  // filter tiles of blocks 1 .. L-1
#ifndef TSD_MEGA_SKIP_FILTER  // (timing experiments only: wrong results)
        if constexpr (H == 256) {
            if (A.filter_rows == 2 * T) filter_role_h<H, 2>(A.f, filter_item, smem, A.status, FilterSave{} TSD_TRACE_NULL);
            else filter_role_h<H, 1>(A.f, filter_item, smem, A.status, FilterSave{} TSD_TRACE_NULL);
        } else {
            filter_role_h<H, 1>(A.f, filter_item, smem, A.status, FilterSave{} TSD_TRACE_NULL);
        }
#endif
        // every storing wave drains, the workgroup meets, ONE lane publishes the tile (Guideline 16 R1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(A.ctl + MegaCtl::FILTER0 + filter_item + A.tiles_per_layer, epoch, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);  // (filter_item counts from block 1)
        TSD_MEGA_T(0, 3);
        TSD_MEGA_T(2, wall_clock64());
        TSD_MEGA_T(3, (unsigned long long)(1 + filter_item / A.tiles_per_layer));
        return;
    }
    if (pair_item >= 0) {
        ComboPre q = A.q;
        q.ready_target = epoch * 64 + A.L;
        if constexpr (H == 256) {
            if (A.pair_rows == 2 * T) pair_role_h<H, false, false, 2>(q, pair_item, A.n_node, smem, /* defer_pre */ pair_item < parked, A.status);
            else pair_role_h<H>(q, pair_item, A.n_node, smem, /* defer_pre */ pair_item < parked, A.status);
        } else {
            pair_role_h<H>(q, pair_item, A.n_node, smem, /* defer_pre */ pair_item < parked, A.status);
        }
        TSD_MEGA_T(0, 4);
        TSD_MEGA_T(2, wall_clock64());
    }
}

// Resident-workgroup slots of the one-launch kernel on the current device: occupancy of the kernel (its LDS and
// registers) x compute units, queried once per device.  The launch condition (api.hip mega_shape) keeps the node
// workgroups -- which wait for workgroups dispatched after them -- within HALF of these slots, so that on a partitioned
// or smaller part the filter tiles they wait for always find a free slot.  (A second tenant on the same GPU can still
// hold slots: that is what the bounded waits and the host's per-block rerun are for.)
static size_t mega_lds_bytes(int H) {
    size_t lds = lds_combo(H, PREC_H2, H == 256 && TSD_MEGA_WIDE_MIN > 0 ? 2 : 1);
    if (H == 256 && MEGA_TR == TN && TSD_MEGA_XLDS)  // the node workgroups' LDS copy of x (xl_gather) behind their planes
        lds = std::max(lds, (size_t)(TN * ldh_of(256) + XS_ROWS * 256 + 2 * 256 /* s_bias */) * 4);
    return lds;
}
int mega_slots(int H) {
    static std::atomic<int> cache[3][64];
    const int hi = H == 64 ? 0 : (H == 128 ? 1 : 2);
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return 0;
    d &= 63;
    int v = cache[hi][d].load();
    if (v > 0) return v;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 0;
    int per_cu = 0;
    const size_t lds = mega_lds_bytes(H);
    hipError_t e = hipErrorInvalidValue;
#define TSD_OCC(HH)                                                                                             \
    {                                                                                                           \
        if (lds > 48 * 1024)                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(forward_mega_kernel<HH>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, forward_mega_kernel<HH>, 2 * HH, lds);        \
    }
    if (H == 64) TSD_OCC(64) else if (H == 128) TSD_OCC(128) else if (H == 256) TSD_OCC(256)
#undef TSD_OCC
    if (e != hipSuccess || per_cu <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    v = per_cu * prop.multiProcessorCount;
    cache[hi][d].store(v);
    return v;
}
// The forwards of the batch's M checkpoints in ONE launch, in groups of G (MegaGroup: per-checkpoint strides of the arrays;
// M = G = 1: the single-checkpoint form).  G * node workgroups must fit half of the device's slots (api.hip mega_group).
int launch_forward_mega(const tsd_model_cfg& c, const tsd_batch& b, const float* pos, const float* W16, float* ea, float* wf,
                        float* h, float* x1m, size_t x1_stride, int32_t* ctl, const int32_t* epoch_src, int epoch_bias,
                        int32_t* status, hipStream_t st, const MegaGroup& mg) {
    (void)pos;
    const WeightLayout WL = weight_layout(c);
    const int H = c.hidden, L = c.num_convs, N = b.num_nodes, P = b.num_pairs, PU = P / 2;
    const tsd_geometry& g = b.geo;
    if (mg.G < 1 || mg.M < 1) {
        set_error("internal: checkpoint groups (%d of %d)", mg.G, mg.M);
        return TSD_ERR_INVALID;
    }
    MegaArgs A{};
    A.G = mg.G;
    A.M = mg.M;
    A.s_w = WL.total;
    A.s_nh = mg.s_nh;
    A.s_x1m = mg.s_x1m;
    A.s_wf = mg.s_wf;
    A.s_ea = mg.s_ea;
    A.s_inv = (size_t)PU;
    A.s_ctl = mg.s_ctl;
    A.L = L;
    A.N = N;
    A.epoch_src = epoch_src;
    A.epoch_bias = epoch_bias;
    A.ctl = ctl;
    A.status = status;
    // node role
    A.graph_ptr = b.graph_ptr;
    A.node_graph = b.node_graph;
    A.row_ptr = g.enc.row_ptr;
    A.dst = g.enc.dst;
    A.umap = g.enc.umap;
    A.W = W16;
    A.layer0 = WL.layer0;
    A.layer_stride = WL.layer_stride;
    A.o_lin1 = WL.L_lin1_w;
    A.o_lin2_w = WL.L_lin2_w;
    A.o_lin2_b = WL.L_lin2_b;
    A.o_lin_w = WL.L_lin_w;
    A.o_lin_b = WL.L_lin_b;
    A.z = b.z;
    A.x1_0 = b.x1_0;
    A.x1m = x1m;
    A.x1_stride = x1_stride;
    A.h = h;
    A.wf = wf;
    A.wf_layer_stride = (size_t)PU * H;
    // filter role: the queue of kernels_combo's per-block launches from block 1 on, slot = block
    // (64-row filter tiles by the GROUP's tiles per block: they share the slots)
    const bool narrow = (b.reserved & 2) != 0;   // (tsd_batch.reserved bit 1: 32-row tiles everywhere; bit-identical -- tests, A/B)
    const int frb = (H == 256 && !narrow && TSD_MEGA_WIDE_MIN > 0 && (long)filter_tiles_per_layer(PU) * mg.G >= TSD_MEGA_WIDE_MIN) ? 2 : 1;
    A.filter_rows = T * frb;
    A.tiles_per_layer = (PU + T * frb - 1) / (T * frb);
    A.f.tiles = (L - 1) * A.tiles_per_layer;
    A.f.g_begin = A.tiles_per_layer;
    A.f.tiles_per_layer = A.tiles_per_layer;
    A.f.layer0 = 0;
    A.f.Wl0 = W16 + WL.layer0;
    A.f.layer_stride = WL.layer_stride;
    A.f.o_nn0_w = WL.L_nn0f_w;
    A.f.o_nn0_b = WL.L_nn0f_b;
    A.f.o_nn2_w = WL.L_nn2_w;
    A.f.o_nn2_b = WL.L_nn2_b;
    A.f.conv_cutoff = c.conv_cutoff;
    A.f.smooth = c.smooth_conv;
    A.f.e = g.enc_u;
    A.f.edge_attr = ea;
    A.f.wf = wf;
    A.f.wf_layer_stride = (size_t)PU * H;
    A.f.wf_slots = L;
    // pair role
    // 64-row pair tiles where a group's pair tiles are more than a round of the chip (8 checkpoints at batch 100: 1.199 ->
    // 1.182 ms/step; 200 graphs, 2 checkpoints: equal); a single batch-100 forward ends with ONE partial round of pair tiles,
    // which the longer tile only stretches (0.2007 -> 0.2046)
#ifndef TSD_MEGA_PAIR_WIDE_MIN
#define TSD_MEGA_PAIR_WIDE_MIN 768  // 32-row pair tiles of a group from which they are 64 rows (0: never)
#endif
    A.pair_rows = (H == 256 && !narrow && TSD_MEGA_PAIR_WIDE_MIN > 0 && (long)((PU + T - 1) / T) * mg.G >= TSD_MEGA_PAIR_WIDE_MIN) ? 2 * T : T;
    A.q.tiles = (PU + A.pair_rows - 1) / A.pair_rows;
    A.q.e = g.out_u;
    A.q.edge_attr = ea;
    A.q.attr_row = g.attr_row;
    A.q.w0b = W16 + WL.out_w0f;
    A.q.b0 = W16 + WL.out_b0f;
    A.q.pair = 1;
    A.q.w0a = W16 + WL.out_w0;
    A.q.w1 = W16 + WL.out_w1;
    A.q.b1 = W16 + WL.out_b1;
    A.q.w2 = W16 + WL.out_w2;
    A.q.b2 = W16 + WL.out_b2;
    A.q.h = h;
    A.q.edge_inv = b.edge_inv_u;
    A.q.ready = ctl + MegaCtl::NODE0;
    A.q.ready_div = MEGA_TR;
    A.q.status = status;
    A.q.inv_stride = (size_t)PU;
    // grid
    constexpr int TRH = MEGA_TR;  // (H = 64 / 128 / 256: 2 / 4 / 8 waves, all <= MEGA_TR rows)
    A.n_node = (N + TRH - 1) / TRH;
    A.n_filter = A.f.tiles;
    A.n_pair = A.q.tiles;
    A.half_slots = mega_slots(H) / 2;
    if ((long)A.n_node * A.G > A.half_slots) {
        set_error("internal: %d x %d node workgroups on a device with %d resident slots", A.G, A.n_node, 2 * A.half_slots);
        return TSD_ERR_INVALID;
    }
    // (tests, tsd_batch.reserved bit 3: the last filter tile of the last block is never run, so the node workgroup that
    // reads its rows waits until the bound -- the TSD_STATUS_INTERNAL path end to end)
    if ((b.reserved & 8) && A.n_filter > 0) --A.n_filter;
    const int grid = A.n_node + A.n_filter + A.n_pair;
    if (grid == 0 || A.n_node == 0) return TSD_OK;
    const size_t lds = mega_lds_bytes(H);
    A.per_group = grid * A.G;
    const int groups = (A.M + A.G - 1) / A.G;
#define TSD_MEGA(HH)                                                                                         \
    {                                                                                                        \
        static DeviceOnce once;                                                                              \
        int r = allow_lds(forward_mega_kernel<HH>, lds, once);                                               \
        if (r) return r;                                                                                     \
        hipLaunchKernelGGL(forward_mega_kernel<HH>, dim3(A.per_group * groups), dim3(2 * HH), lds, st, A);   \
    }
    switch (H) {
        case 64: TSD_MEGA(64) break;
        case 128: TSD_MEGA(128) break;
        case 256: TSD_MEGA(256) break;
        default: set_error("hidden=%d unsupported (64/128/256)", H); return TSD_ERR_INVALID;
    }
#undef TSD_MEGA
    TSD_LAUNCH_CHECK("forward_mega");
    return TSD_OK;
}
size_t mega_ctl_words(int tiles_per_layer, int L) { return (size_t)MegaCtl::FILTER0 + (size_t)tiles_per_layer * (size_t)(L > 0 ? L : 1); }
int mega_node_rows() { return MEGA_TR; }


// layer == -1: node role = lin1 of block 0 only; layer == -2: no node role.
// Filter role: items [g_begin, g_begin + g_count) of the queue (layer_w0 + g / tiles_per_layer, g % tiles_per_layer);
// wf_base = filter slot 0, layer l of the queue writes slot l % wf_slots.  g_count == 0: no filter role.
int launch_layer_combo(const tsd_model_cfg& c, const float* W, int layer, int N, tsd_edges enc,
                       const float* Wf_layer, const float* x1_in, const float* h_in, float* h, float* x1_out,
                       int layer_w0, int g_begin, int g_count,
                       int capacity_u, tsd_edges enc_u, const float* edge_attr, float* wf_base, int wf_slots, int M,
                       size_t nh_stride, size_t ea_stride, size_t wf_stride, hipStream_t st, const ComboPre* pre,
                       size_t pre_stride, const FilterSave* fsave, const NodeSave* nsave, bool folded, Prec prec) {
    const WeightLayout L = weight_layout(c);
    ComboNode a{};
    a.N = N;
    a.row_ptr = enc.row_ptr;
    a.dst = enc.dst;
    a.umap = enc.umap;
    a.Wf = Wf_layer;
    a.x1_in = x1_in;
    a.h_in = h_in ? h_in : h;
    a.h = h;
    a.x1_out = x1_out;
    if (layer < 0) {
        a.mode = 1;
        a.lin1_next_w = W + L.layer0 + L.L_lin1_w;
    } else {
        const float* B = W + L.layer0 + (size_t)layer * L.layer_stride;
        a.mode = 0;
        a.lin2_w = B + L.L_lin2_w;
        a.lin2_b = B + L.L_lin2_b;
        a.lin_w = B + L.L_lin_w;
        a.lin_b = B + L.L_lin_b;
        a.lin1_next_w = (layer + 1 < c.num_convs) ? B + L.layer_stride + L.L_lin1_w : nullptr;
    }
    ComboFilter f{};
    f.tiles = 0;
    if (g_count > 0) {
        f.tiles = g_count;
        f.g_begin = g_begin;
        f.tiles_per_layer = filter_tiles_per_layer(capacity_u);
        f.layer0 = layer_w0;
        f.Wl0 = W + L.layer0;
        f.layer_stride = L.layer_stride;
        f.o_nn0_w = folded ? L.L_nn0f_w : L.L_nn0_w;  // folded: `edge_attr` holds s1 (common.hpp, FOLDED WEIGHTS)
        f.o_nn0_b = folded ? L.L_nn0f_b : L.L_nn0_b;
        f.o_nn2_w = L.L_nn2_w;
        f.o_nn2_b = L.L_nn2_b;
        f.conv_cutoff = c.conv_cutoff;
        f.smooth = c.smooth_conv;
        f.e = enc_u;
        f.edge_attr = edge_attr;
        f.wf = wf_base;
        f.wf_layer_stride = (size_t)capacity_u * c.hidden;
        f.wf_slots = wf_slots < 1 ? 1 : wf_slots;
    }
    ComboPre q{};
    if (pre && pre->tiles > 0) q = *pre;
    // Wide node tiles (32 rows: node_role_hw) where the launch's node tiles alone are a chip-full of workgroup slots: their
    // slot time, not the chain's latency, is what the launch pays for.  Inference form of the production width, ONE
    // checkpoint, from the size at which the one-launch forward no longer applies (> 256 node tiles): measured
    // (tools/ab_step.py) 300 graphs 0.514 -> 0.486 ms/step, 400: 0.636 -> 0.600, 500: 0.756 -> 0.742, 600: 0.942 -> 0.920;
    // an 8-checkpoint ensemble at batch 100 LOST with the checkpoint-major grid (1.405 -> 1.454: eight weight sets do not
    // stay in an XCD's L2 and the shallower weight ring of the wide role -- two k-steps in flight: three do not fit 128
    // VGPRs -- then shows) and WINS since the checkpoints are interleaved over the XCDs (common.hpp wg_item_ckpt):
    // 1.320 -> 1.285; 300 graphs x 8 checkpoints 3.633 -> 3.527.
    int nrb = 1;
#ifndef TSD_NODE_WIDE_ANY_M
#define TSD_NODE_WIDE_ANY_M 1  // wide node tiles for ensembles too (0: single-checkpoint launches only; A/B builds)
#endif
#ifndef TSD_NODE_WIDE_MIN
#define TSD_NODE_WIDE_MIN 257  // 16-row node tiles of a (single-checkpoint) launch from which they are 32 rows; 0: never
#endif
    if (prec.mode == PREC_H2 && c.hidden == 256 && !fsave && !nsave && q.tiles == 0 && layer != -2 && TSD_NODE_WIDE_MIN > 0 &&
        !prec.narrow_filter_tiles && (M == 1 || TSD_NODE_WIDE_ANY_M) && (long)((N + TN - 1) / TN) * M >= TSD_NODE_WIDE_MIN)
        nrb = 2;
    const int node_tiles = layer == -2 ? 0 : (N + TN * nrb - 1) / (TN * nrb);
    if (q.pair) a.ready = q.ready;  // the node role of this launch publishes h to the pair tiles
    // Whole layers of filter tiles, many chip-fulls of them (big batches / ensembles): 64-row tiles.  A weight fragment
    // of the ring then feeds two row blocks, and the launch is bound by that feed.
    int frb = 1;
    // (the saving form keeps 32-row tiles: with two row blocks AND the saves it does not fit two workgroups per CU)
    if (prec.mode == PREC_H2 && !prec.narrow_filter_tiles && !fsave && f.tiles > 0 && q.tiles == 0 && TSD_FILTER_WIDE_MIN > 0 && f.tiles % f.tiles_per_layer == 0 &&
        f.g_begin % f.tiles_per_layer == 0 && (long)f.tiles * M >= TSD_FILTER_WIDE_MIN) {
        const int tpl2 = (capacity_u + 2 * T - 1) / (2 * T);
        f.g_begin = f.g_begin / f.tiles_per_layer * tpl2;
        f.tiles = f.tiles / f.tiles_per_layer * tpl2;
        f.tiles_per_layer = tpl2;
        frb = 2;
    }
    const int grid = node_tiles + f.tiles + q.tiles;
    if (grid == 0) return TSD_OK;
    size_t lds = lds_combo(c.hidden, prec.mode, frb, nrb);
    if (q.tiles > 0 && q.pair && q.rows == 2 * T) lds = std::max(lds, lds_combo(c.hidden, prec.mode, 2, 1));  // (64-row pair tiles)
    // interleave only when the launch is many chip-fulls deep (the node tiles alone over-subscribe the chip)
    // (never with the waiting pair role: its tiles spin on the node tiles' ready flags, so every node tile has to be
    // DISPATCHED before any of them -- node tiles first, S = 1; a spread node tile behind a slot-filling crowd of
    // waiting pair tiles would never start: ADVICE r05, tests/test_gpu_round6.py)
    int node_stride = 1;
    if (!q.pair && node_tiles >= 1024 && grid >= 3 * node_tiles) {
        node_stride = grid / node_tiles;
        if (node_stride % 2 == 0) --node_stride;
    }
    const ComboStride sd{L.total, nh_stride, ea_stride, wf_stride, pre_stride, ckpt_grid_m(M, grid), node_stride, prec.range_status};
    if (prec.mode == PREC_H2 && q.tiles > 0 && !q.pair) {
        set_error("internal: the split-f16 block launch has no pre role");
        return TSD_ERR_INVALID;
    }
#ifdef TSD_TRACE
#define TSD_TRACE_HOST , g_tsd_trace_host
#else
#define TSD_TRACE_HOST
#endif
#define TSD_COMBO_I(HH, SV, TL, PR, FR, ...)                                                                \
    {                                                                                                       \
        static DeviceOnce once;                                                                             \
        int r = allow_lds(layer_combo_kernel<HH, SV, TL, PR, FR, ##__VA_ARGS__>, lds, once);                \
        if (r) return r;                                                                                    \
        hipLaunchKernelGGL((layer_combo_kernel<HH, SV, TL, PR, FR, ##__VA_ARGS__>), dim3(grid * M), dim3(2 * HH), lds, st, a, \
                           node_tiles, f, sd, q, fsv, nsv TSD_TRACE_HOST);                                  \
    }
#define TSD_COMBO(HH)                                                                                       \
    if (prec.mode == PREC_H2) {                                                                             \
        if (save) TSD_COMBO_I(HH, true, false, PREC_H2, 1)                                                  \
        else if (q.tiles > 0) TSD_COMBO_I(HH, false, true, PREC_H2, 1)                                      \
        else if (nrb == 2 && HH == 256 && frb == 2) TSD_COMBO_I(256, false, false, PREC_H2, 2, 2)           \
        else if (nrb == 2 && HH == 256) TSD_COMBO_I(256, false, false, PREC_H2, 1, 2)                       \
        else if (frb == 2) TSD_COMBO_I(HH, false, false, PREC_H2, 2)                                        \
        else TSD_COMBO_I(HH, false, false, PREC_H2, 1)                                                      \
    } else if (save) TSD_COMBO_I(HH, true, false, PREC_F32, 1)                                              \
    else if (q.tiles > 0) TSD_COMBO_I(HH, false, true, PREC_F32, 1) else TSD_COMBO_I(HH, false, false, PREC_F32, 1)
    const bool save = fsave != nullptr || nsave != nullptr;
    const FilterSave fsv = fsave ? *fsave : FilterSave{};
    const NodeSave nsv = nsave ? *nsave : NodeSave{};
    if (save && ((f.tiles > 0 && !fsave) || (node_tiles > 0 && a.mode == 0 && !nsave) || M != 1 || q.tiles > 0)) {
        set_error("internal: the saving block launch needs both save sets and one checkpoint");
        return TSD_ERR_INVALID;
    }
    switch (c.hidden) {
        case 64: TSD_COMBO(64) break;
        case 128: TSD_COMBO(128) break;
        case 256: TSD_COMBO(256) break;
        default: set_error("hidden=%d unsupported (64/128/256)", c.hidden); return TSD_ERR_INVALID;
    }
#undef TSD_COMBO
#undef TSD_COMBO_I
    TSD_LAUNCH_CHECK("layer_combo");
    return TSD_OK;
}


}  // namespace tsd


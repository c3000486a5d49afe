// kernels_unit.hip -- the SchNet encoder (all L interaction blocks) as ONE launch of per-unit workgroups that never
// write the CFConv filters to memory: reference models/encoder/schnet.py:74-128,203-225
//   per block:  W = nn(edge_attr) * C ;  agg_i = sum_{e: (j -> i)} x1[j] * W_e ;  h += lin(ssp(lin2(agg))) ;  x1 = lin1_next(h)
//
// WHY.  Every other form of the forward materialises the filter W [E_u, H] fp32 per block: one write and two reads
// (both end points gather it), 267 of the 391 MB a batch-100 launch moves and 8.25 GB per block launch at BASELINE
// configs[4] -- there the block launch is 2.1 ms against 0.65 ms of MFMA time, and an HBM-bound gather (the node
// role) shares the chip with the MFMA-bound filter tiles.  Here a UNIT -- a run of whole graphs with at most 64 atoms
// in total; graphs never interact -- is owned by one workgroup that owns a CU (145 KB of LDS, up to 256 VGPRs):
//   LDS:  x1 of the unit's atoms [64][H] fp32 (64 KB) | one tile of 64 undirected pairs as two f16 planes (66 KB), over
//         which the finished filter tile lies as fp32 rows | cutoff weights and local end points of the unit's pairs
//   VGPR: the aggregate `agg` of the unit's atoms (32 registers per lane), accumulators, weight ring
// Per block and tile of 64 consecutive undirected pairs: attribute rows (prefetched one tile ahead) -> planes; GEMM nn.0;
// ssp; GEMM nn.2; x C -> fp32 tile in LDS; both end points' messages x1[other] * W accumulated into `agg`.  Then the
// node chain of the block on the unit's (<= 64) rows, whose lin1 output lands in the LDS x1 of the next block.  Nothing
// crosses workgroups: no flags, no in-kernel waits, no launch boundaries inside the encoder, and the only per-block HBM
// traffic is ONE read of the attribute rows (2.1 GB at configs[4]) -- the filters and x1 never leave the CU.
//
// SAME BITS as the materialising forms.  The undirected list is sorted by (src, dst), so walking it IN ORDER and adding
// pair (i, j)'s two messages to row i and row j visits the partners of every row in ascending order -- exactly the
// directed CSR order of aggregate_tile (kernels_combo.hip) -- with the product rounded before the add.  The filter rows
// come from the same MFMA sequences as filter_role_h (every row of a tile is the same chain of MFMAs whatever the tile
// size), the node chain runs node_role_h's 16-row MFMAs on four row blocks.  So the encoder's h is BIT-IDENTICAL to the
// launch-per-block and one-launch split-f16 forwards (tests/test_gpu_round4.py asserts torch.equal).
//
// Row ownership in the accumulation: wave w holds agg of the rows r with r % 8 == w (4 columns per lane, agg[r / 8]: a
// uniform dynamic index into eight 4-vectors, s_set_gpr_idx moves).  No atomics; the two tile forms are described at
// `blk` in the kernel.
#include "common.hpp"

namespace tsd {

constexpr int UT = 64;         // undirected pairs per filter tile (two 32-row MFMA blocks)
constexpr int UNA = 64;        // atoms per unit (TSD_UNIT_MAX_NODES)
constexpr int UE_MAX = 2016;   // undirected pairs per unit: one complete 64-atom graph
constexpr int U_ROWS = 36 * 64;  // tile rows of a unit: 31.5 list tiles, or 36 block tiles (8 blocks: 8 * 9 / 2)

struct UnitArgs {
    int L, N, num_units, M;  // M: checkpoints, signed (common.hpp wg_item_ckpt)
    const int32_t* unit_node;  // [num_units + 1] node offsets
    const int32_t *node_graph, *graph_ptr, *pair_ptr, *pair2u;  // topology / geometry tables (block mode: tile row -> pair of the list)
    tsd_edges eu;              // undirected encoder list (row_ptr, src, dst, dist)
    const float* W;            // f16-plane weight arena of checkpoint 0 (checkpoint m at + m * w_stride)
    size_t w_stride;
    size_t layer0, layer_stride, o_nn0_w, o_nn0_b, o_nn2_w, o_nn2_b, o_lin1, o_lin2_w, o_lin2_b, o_lin_w, o_lin_b;
    const float* ea;           // edge attributes (s1 rows of the typed embedding) [M][.., H]
    size_t ea_stride;
    const float *z, *x1_0;     // [M][N, H] pos-independent inputs of block 0
    float* h;                  // [M][N, H] node states (written per block, final value read by the pair MLP)
    size_t nh_stride;
    float conv_cutoff;
    int smooth;
    int l_begin, l_end;        // blocks [l_begin, l_end) (the whole encoder: 0, L); l_begin > 0 reads h / x1 from memory
    float* x1_io;              // [M][N, H] x1 in (l_begin > 0) / out (l_end < L); may be NULL for the whole encoder
    int32_t* status;           // TSD_STATUS_RANGE / TSD_STATUS_INTERNAL (a unit that breaks the size contract)
};

// TRANSPOSED ACCUMULATORS (round 5, split16.hpp hgemm_ring_run<..., TRANS>): the weight fragment is the MFMA's A operand, so
// a lane ends up with runs of four consecutive CHANNELS of one tile row -- the ssp epilogue writes f16x4 (8 LDS stores per
// lane, plane and 32-row block instead of 32), the filter tile goes to LDS as f32x4, the cutoff weight is one value per
// lane, the node chain reads / writes h as 16-byte accesses.  Bit-identical (the same MFMA dot products).
// -DTSD_UNIT_TRANS=0 builds the round-4 form (A/B timing: tools/build_variant.sh).
#ifndef TSD_UNIT_TRANS
#define TSD_UNIT_TRANS 1
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Phase trace (variant builds only: tools/build_variant.sh utrace "-DTSD_UNIT_TRACE" kernels_unit.hip; tools/trace_unit.py):
// per workgroup the shader-clock cycles wave 0 spent in each phase, summed over tiles and blocks.
// slots: 0 convert, 1 GEMM nn.0, 2 ssp, 3 GEMM nn.2 (+ prefetch issue), 4 filter tile -> LDS, 5 accumulate, 6 node chain,
// 7 staging, 8 tiles x blocks, 9 total
#ifdef TSD_UNIT_TRACE
__device__ unsigned long long g_unit_trace[4096 * 16];
extern "C" int tsd_debug_unit_trace(void* host_buf) {
    return (int)hipMemcpyFromSymbol(host_buf, HIP_SYMBOL(g_unit_trace), sizeof(g_unit_trace));
}
#define UTRACE_DECL unsigned long long ut_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ut_t = __builtin_amdgcn_s_memtime(); const unsigned long long ut_t0 = ut_t;
#define UTRACE(slot) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ut_acc[slot] += n_ - ut_t; ut_t = n_; } while (0)
#define UTRACE_COUNT(slot) do { ut_acc[slot] += 1; } while (0)
#define UTRACE_FLUSH do { ut_acc[9] = __builtin_amdgcn_s_memtime() - ut_t0; if (threadIdx.x == 0 && m == 0 && u < 4096) { for (int i_ = 0; i_ < 10; ++i_) g_unit_trace[(size_t)u * 16 + i_] = ut_acc[i_]; } } while (0)
#else
#define UTRACE_DECL
#define UTRACE(slot)
#define UTRACE_COUNT(slot)
#define UTRACE_FLUSH
#endif

// A copy of a lane value the compiler must treat as new: the address arithmetic that hangs off it is recomputed where
// it is used (a few VALU instructions per phase) instead of being hoisted out of the block and tile loops as hundreds
// of loop-invariant registers (first build of this kernel: 256 VGPRs + 1 KB of scratch, 214 spill stores in the prologue).
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <int H>
__global__ __launch_bounds__(2 * H) void unit_encoder_kernel(UnitArgs A) {
    static_assert(H == 256, "the unit encoder is built for hidden 256 (8 waves)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDH = ldh_of(H), LDA = H + 4, NT = 2 * H, C4 = H / 4;
    constexpr int NIT = UT * C4 / NT;  // float4 loads per thread and attribute tile (8)
    constexpr int RB16 = UNA / 16, CB16 = 2;
    float* x1s = smem;                               // [UNA][H] fp32
    float* tile = x1s + UNA * H;                     // planes (UT x LDH floats) / fp32 filter tile [UT][LDA]
    float* s_c = tile + UT * LDH;                    // [U_ROWS] cutoff weight per tile row (list mode: row = local pair index)
    uint16_t* s_u = reinterpret_cast<uint16_t*>(s_c + U_ROWS);  // [U_ROWS] list mode: local (i | j << 8) of the pair;
                                                                // block mode: local pair index of the tile row, 0xffff: none
    uint16_t* s_tile = s_u + U_ROWS;                 // [64] block mode: blocks (I * 16 + J) of tile t
    float* s_bias = reinterpret_cast<float*>(s_tile + 64);  // [2][H] transposed form: nn.0 / nn.2 biases of the running block
    constexpr bool UTR = TSD_UNIT_TRANS != 0;
    const Planes pl = planes_at(tile, UT, LDH);
    float* buf = tile;

    int u;       // (1-D grid, checkpoint = id % M: common.hpp wg_item_ckpt; measured neutral for this kernel)
    size_t m;
    wg_item_ckpt(A.M, u, m);
    const int tid = threadIdx.x;
    const int n0 = A.unit_node[u], n1 = A.unit_node[u + 1], na = n1 - n0;
    // The unit partition is the CALLER's (tsd_batch.unit_node): every workgroup checks its own unit -- offsets inside
    // [0, N] and ascending, the first unit starts at 0 and the last ends at N (so the units cover every atom exactly
    // once), at most UNA atoms, whole graphs only (a unit that cuts a graph would index LDS rows outside [0, UNA) with its
    // 8-bit local atom numbers) -- and reports TSD_STATUS_INTERNAL instead of computing on a broken one.
    bool bad = n0 < 0 || n1 > A.N || n1 < n0 || (u == 0 && n0 != 0) || (u == A.num_units - 1 && n1 != A.N) || na > UNA;
    if (!bad && na > 0)
        bad = A.graph_ptr[A.node_graph[n0]] != n0 || A.graph_ptr[A.node_graph[n1 - 1] + 1] != n1;
    int e0 = 0, e1 = 0;
    if (!bad && na > 0) {
        e0 = A.eu.row_ptr[n0];
        e1 = A.eu.row_ptr[n1];
        bad = e1 < e0 || e1 - e0 > UE_MAX;
    }
    if (bad) {
        if (tid == 0 && A.status != nullptr) atomicOr(A.status, TSD_STATUS_INTERNAL);
        return;
    }
    if (na <= 0) return;
    const int ne = e1 - e0;
    const float* Wm = A.W + m * A.w_stride;
    const float* ea = A.ea + m * A.ea_stride + (size_t)e0 * H;
    float* hm = A.h + m * A.nh_stride;
    float amax = 0.0f;
    UTRACE_DECL

    // BLOCK MODE (a unit that is ONE graph of more than 32 atoms): tiles are 8 x 8 blocks (I, J), I <= J, of the atom-pair
    // matrix in the order (0,0) (0,1) .. (0,nb-1) (1,1) .. -- row a * 8 + b of a tile is the pair (I*8 + a, J*8 + b), a pair
    // that is no edge (i >= j on the diagonal, beyond the cutoff, past the graph) is a zero row with cutoff weight 0, so
    // its messages are +-0 and change nothing.  Every row of agg still meets its partners in ascending order (blocks
    // 0 .. K-1 as the J side of tiles (I, K), then the diagonal, then the I side of (K, J)): the same bits as list order.
    // What it buys: a tile's 128 messages are 16 per wave at fixed LDS addresses (wave w adds the 8 partners of row
    // J*8 + w, then of row I*8 + w), where a list tile of a dense graph is one 64-long dependent chain on one row.
    // LIST MODE (everything else): tiles of 64 consecutive pairs of the unit's slice of the undirected list.
    const bool blk = na > 32 && A.node_graph[n0] == A.node_graph[n1 - 1];
    const int nb = (na + 7) >> 3;
    const int ntile = blk ? nb * (nb + 1) / 2 : (ne + UT - 1) / UT;

    // ---- per-unit staging: cutoff weights and end points / pair indices of the tile rows, x1 of block l_begin ----
    if (blk) {
        for (int idx = tid; idx < ntile * UT; idx += NT) {
            int tt = idx >> 6, I = 0;
            while (tt >= nb - I) {
                tt -= nb - I;
                ++I;
            }
            const int J = I + tt, pr = idx & 63;
            int a = pr >> 3, b = pr & 7;
            bool row_ok = true;
            if (I == J) {   // a DIAGONAL tile is 32 rows: the 28 pairs a < b of the block in the order (0,1) .. (0,7) (1,2) ..
                row_ok = pr < 28;
                int q = pr;
                a = 0;
                while (row_ok && q >= 7 - a) {
                    q -= 7 - a;
                    ++a;
                }
                b = a + 1 + q;
            }
            const int i = I * 8 + a, j = J * 8 + b;
            int loc = -1;
            if (row_ok && i < j && j < na) {
                const int uu = A.pair2u[A.pair_ptr[n0 + i] + j - 1];
                if (uu >= e0 && uu < e1) loc = uu - e0;
            }
            s_u[idx] = (uint16_t)(loc < 0 ? 0xffff : loc);
            s_c[idx] = loc < 0 ? 0.0f : cutoff_weight(A.eu.dist[e0 + loc], A.conv_cutoff, A.smooth);
            if (pr == 0) s_tile[idx >> 6] = (uint16_t)(I * 16 + J);
        }
    } else {
        for (int e = tid; e < ne; e += NT) {
            s_c[e] = cutoff_weight(A.eu.dist[e0 + e], A.conv_cutoff, A.smooth);
            s_u[e] = (uint16_t)((A.eu.src[e0 + e] - n0) | ((A.eu.dst[e0 + e] - n0) << 8));
        }
    }
    {
        const float* x_in = (A.l_begin == 0 ? A.x1_0 : A.x1_io) + m * A.nh_stride + (size_t)n0 * H;
        const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
        for (int idx = tid; idx < UNA * C4; idx += NT)  // (rows past the unit: zeros -- block mode multiplies them by +-0)
            *reinterpret_cast<f32x4*>(x1s + idx * 4) = idx < na * C4 ? *reinterpret_cast<const f32x4*>(x_in + idx * 4) : zz;
    }
    __syncthreads();
    const int nrb = (na + 15) >> 4;  // 16-row blocks of the node chain that hold atoms

    // accumulation geometry of a wave -- list mode: 128 columns (2 per lane), rows = class + 4 k; block mode: row w of
    // every 8-atom block, 4 columns per lane;  GEMM geometry: 32 output columns per wave (every phase derives its lane
    // geometry from an opaque copy of the thread index: see opaque())
#define TSD_UNIT_GEOM                                                         \
    const int tq = opaque(tid);                                               \
    const int wave = tq >> 6, lane = tq & 63;                                 \
    const int cls = wave & 3, acol = (wave >> 2) * (H / 2) + lane * 2;        \
    const int hi = lane >> 5, l31 = lane & 31;                                \
    const int col0 = wave * 32, col = col0 + l31;                             \
    const int q = lane >> 4, l15 = lane & 15;                                 \
    (void)cls; (void)acol; (void)hi; (void)l31; (void)col; (void)q; (void)l15; (void)col0;

    for (int l = A.l_begin; l < A.l_end; ++l) {
        const float* Wl = Wm + A.layer0 + (size_t)l * A.layer_stride;
        const float *nn0_w = Wl + A.o_nn0_w, *nn2_w = Wl + A.o_nn2_w;
        // agg of the unit's atoms, 32 registers per lane: agg[k < 8] = row k * 8 + wave, columns 4 lane .. + 3
        // (one array of eight 4-vectors in both modes: a whole-vector access with a uniform dynamic index compiles to
        // s_set_gpr_idx moves; a float[32] with scalar dynamic indices went to scratch memory)
        f32x4 agg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) agg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        float b0 = 0.0f, b2 = 0.0f;  // (loaded here, not in the tile loop: a global load there waits behind the prefetch in vmcnt order)
        if constexpr (UTR) {   // a lane needs 16 channels' biases: both vectors of the block in LDS (the barrier that ends the
                               // first tile's conversion phase is between these stores and the first read)
            if (tid < H) {
                s_bias[tid] = Wl[A.o_nn0_b + tid];
                s_bias[H + tid] = Wl[A.o_nn2_b + tid];
            }
        } else {
            TSD_UNIT_GEOM
            b0 = Wl[A.o_nn0_b + col];
            b2 = Wl[A.o_nn2_b + col];
        }
        f32x4 v[NIT];  // the attribute rows of the next tile (tile row = wave + 8 it, one 1-KiB row per wave-load)
        unsigned vlive = 0;  // bit it: row `it` of v is a pair (else the tile row is zeros)
        auto fetch = [&](int t) {
            TSD_UNIT_GEOM
            vlive = 0;
            if (blk) {
                const unsigned ij = s_tile[t];
                const int nit = (ij >> 4) == (ij & 15u) ? NIT / 2 : NIT;  // (a diagonal tile: 32 rows)
                unsigned uu[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) uu[it] = it < nit ? (unsigned)s_u[t * UT + wave + it * (NT / 64)] : 0xffffu;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    vlive |= (uu[it] != 0xffffu ? 1u : 0u) << it;
                    if (it < nit)
                        v[it] = *reinterpret_cast<const f32x4*>(ea + (size_t)(uu[it] == 0xffffu ? 0u : uu[it]) * H + lane * 4);
                }
            } else {
                const int nr = min(UT, ne - t * UT);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int r = wave + it * (NT / 64);
                    vlive |= (r < nr ? 1u : 0u) << it;
                    v[it] = *reinterpret_cast<const f32x4*>(ea + (size_t)(t * UT + min(r, nr - 1)) * H + lane * 4);
                }
            }
        };
        if (ntile > 0) fetch(0);
        UTRACE(7);
        for (int t = 0; t < ntile; ++t) {
            const int nrows = blk ? UT : min(UT, ne - t * UT);
            const unsigned tij = blk ? (unsigned)s_tile[t] : 1u;
            const int tI = (int)(tij >> 4), tJ = (int)(tij & 15u);  // block mode: the tile's blocks
            const bool diag = blk && tI == tJ;                      // ... a diagonal tile: 32 rows (28 pairs)
            f32x16 accm[2][1], accx[2][1];
            // phases (a) .. (f) on RB 32-row blocks (2: a full tile; 1: a diagonal tile of block mode)
            auto phases = [&](auto rbc) {
                constexpr int RB = decltype(rbc)::value;
                HRing<1, HRING_R> rg;
                {   // (a) attribute tile -> planes (rows that are no pair: zeros).  The rows hold the two f16 planes already
                    // (common.hpp ATTRIBUTE ROWS AS f16 PLANES; range / low-side checks where they were written): lane = 16-byte chunk
                    // of the row, lanes 0-31 the high plane -- one LDS store per row, no conversion
                    TSD_UNIT_GEOM
#pragma unroll
                    for (int it = 0; it < NIT * RB / 2; ++it) {
                        const int r = wave + it * (NT / 64);
                        const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
                        planes_put_chunk<H>(pl, r * LDH, lane, (vlive >> it & 1u) ? v[it] : zz);
                    }
                    hgemm_ring_start<1, H>(rg, nn0_w, H, col0);
#ifdef TSD_UNIT_WFAKE  // (timing experiment, wrong results: every k-step reads the same 2 KB per wave -- weights from L1)
                    rg.step_bytes = 0;
#endif
                }
                __syncthreads();
                UTRACE(0);
                f32x16(&am)[RB][1] = reinterpret_cast<f32x16(&)[RB][1]>(accm);
                f32x16(&ax)[RB][1] = reinterpret_cast<f32x16(&)[RB][1]>(accx);
                {   // (b) GEMM nn.0
                    TSD_UNIT_GEOM
                    hzero(am, ax);
                    hgemm_ring_run<RB, 1, H, false, UTR>(rg, pl, LDH, am, ax);
                    hgemm_ring_start<1, H>(rg, nn2_w, H, col0);
#ifdef TSD_UNIT_WFAKE
                    rg.step_bytes = 0;
#endif
                }
                __syncthreads();
                UTRACE(1);
                // (e) the NEXT tile's attribute rows are requested here: the loads are issued under the VALU work of the
                // epilogue (64 KB through the CU's address path is ~1000 cycles of issue) and have the whole epilogue to
                // come back in -- they are older than every refill of GEMM nn.2's weight ring, whose counted waits
                // (in-order return) would otherwise stall on an HBM round trip; the ring's first three k-steps were
                // issued before them
                if (t + 1 < ntile) fetch(t + 1);
                if constexpr (UTR) {   // (c) shifted softplus -> planes, transposed accumulators: registers 4 g .. 4 g + 3 of a
                    // lane are four consecutive channels of tile row l31: one 8-byte store per plane and group
                    TSD_UNIT_GEOM
                    const int pb = opaque(l31 * LDH + col0 + 4 * hi);
                    const float* bb = s_bias + opaque(col0 + 4 * hi);
                    f32x4 bv[4];  // (all bias reads ahead of the plane stores: LDS accesses of one wave stay in program order)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) bv[g4] = *reinterpret_cast<const f32x4*>(bb + 8 * g4);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            f32x4 y4;
#pragma unroll
                            for (int r = 0; r < 4; r += 2) {   // (pairs: packed fp32 adds / multiplies, as below)
                                const int q0 = 4 * g4 + r;
                                const f32x2 m2 = {am[rb][0][q0], am[rb][0][q0 + 1]}, x2 = {ax[rb][0][q0], ax[rb][0][q0 + 1]};
                                const f32x2 b2v = {bv[g4][r], bv[g4][r + 1]};
                                const f32x2 v = x2 * SPLIT_INV + m2 + b2v;
                                const f32x2 t = {fast_exp(-fabsf(v[0])), fast_exp(-fabsf(v[1]))};
                                const f32x2 t1 = t + 1.0f;
                                const f32x2 lg = {__builtin_amdgcn_logf(t1[0]), __builtin_amdgcn_logf(t1[1])};
                                const f32x2 mx = {fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
                                const f32x2 l2 = lg * 0.69314718055994530942f;
                                const f32x2 y2 = (mx + l2) - 0.69314718055994530942f;
                                y4[r] = y2[0];
                                y4[r + 1] = y2[1];
                            }
                            planes_store4(pl, pb + rb * 32 * LDH + 8 * g4, y4, amax);
                        }
                } else
                {   // (c) shifted softplus -> planes (one base per plane, the rows of a lane at constant offsets from it)
                    TSD_UNIT_GEOM
                    f16* hb = pl.hi + opaque(4 * hi * LDH + col);
                    f16* lb = pl.lo + opaque(4 * hi * LDH + col);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            // two rows at a time: the adds / multiplies of the pair are packed fp32 instructions
                            // (v_pk_fma / v_pk_add / v_pk_mul_f32: the accumulator registers of rows r, r + 1 are adjacent)
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const f32x2 m2 = {am[rb][0][r], am[rb][0][r + 1]}, x2 = {ax[rb][0][r], ax[rb][0][r + 1]};
                            const f32x2 v = x2 * SPLIT_INV + m2 + b0;
                            const f32x2 t = {fast_exp(-fabsf(v[0])), fast_exp(-fabsf(v[1]))};
                            const f32x2 t1 = t + 1.0f;
                            const f32x2 lg = {__builtin_amdgcn_logf(t1[0]), __builtin_amdgcn_logf(t1[1])};
                            const f32x2 mx = {fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
                            const f32x2 l2 = lg * 0.69314718055994530942f;  // (its own statement: not contracted, as in sspf)
                            const f32x2 y2 = (mx + l2) - 0.69314718055994530942f;
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                f16 yh, yl;
                                amax_upd(amax, y2[k]);
                                split1(y2[k], yh, yl);
                                hb[(rb * 32 + ((r + k) & 3) + 8 * ((r + k) >> 2)) * LDH] = yh;
                                lb[(rb * 32 + ((r + k) & 3) + 8 * ((r + k) >> 2)) * LDH] = yl;
                            }
                        }
                }
                __syncthreads();
                UTRACE(2);
                {   // (d) GEMM nn.2
                    hzero(am, ax);
                    hgemm_ring_run<RB, 1, H, false, UTR>(rg, pl, LDH, am, ax);
                }
                __syncthreads();
                UTRACE(3);
                if constexpr (UTR) {   // (f) W = (nn.2 + b) * C as fp32 rows over the planes, transposed accumulators: the cutoff
                    // weight of the lane's row is ONE value, four consecutive channels are one 16-byte store
                    TSD_UNIT_GEOM
                    float* wb = buf + opaque(l31 * LDA + col0 + 4 * hi);
                    const float* bb = s_bias + H + opaque(col0 + 4 * hi);
                    f32x4 bv[4];
                    float cw[RB];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) bv[g4] = *reinterpret_cast<const f32x4*>(bb + 8 * g4);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) cw[rb] = s_c[t * UT + rb * 32 + l31];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            f32x4 w4;
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                w4[r] = (hval(am[rb][0], ax[rb][0], 4 * g4 + r) + bv[g4][r]) * cw[rb];
                            *reinterpret_cast<f32x4*>(wb + rb * 32 * LDA + 8 * g4) = w4;
                        }
                } else
                {   // (f) W = (nn.2 + b) * C as fp32 rows over the planes (the cutoff weights of four consecutive rows by one
                    // 16-byte LDS read, all of a row block's up front: stores to `buf` and loads of `s_c` would otherwise be
                    // kept in program order, one LDS round trip per element)
                    TSD_UNIT_GEOM
                    float* wb = buf + opaque(4 * hi * LDA + col);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        f32x4 cw[4];
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) cw[g4] = *reinterpret_cast<const f32x4*>(s_c + t * UT + rb * 32 + 8 * g4 + 4 * hi);
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            wb[(rb * 32 + (r & 3) + 8 * (r >> 2)) * LDA] = (hval(am[rb][0], ax[rb][0], r) + b2) * cw[r >> 2][r & 3];
                    }
                }
                __syncthreads();
                UTRACE(4);
            };
            if (diag) phases(IntC<1>{});
            else phases(IntC<2>{});
            if (diag) {   // (g) a diagonal tile: row w of the block takes its partners below it (tile rows (a, w), a < w), then above
                TSD_UNIT_GEOM
                const float* wrow = buf + lane * 4;
                const float* xrow = x1s + (tI * 8) * H + lane * 4;
                const int wq = __builtin_amdgcn_readfirstlane(wave);
                f32x4 a4 = agg[tI];
#pragma unroll
                for (int a = 0; a < 7; ++a)
                    if (a < wq) {   // pair (a, w): tile row a (15 - a) / 2 + (w - a - 1)
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + (a * (15 - a) / 2 + (wq - a - 1)) * LDA);
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + a * H);
#pragma unroll
                        for (int c = 0; c < 4; ++c) a4[c] = __fadd_rn(a4[c], __fmul_rn(xv[c], wv[c]));
                    }
                const int off_w = wq * (15 - wq) / 2 - wq - 1;
#pragma unroll
                for (int b = 1; b < 8; ++b)
                    if (b > wq) {   // pair (w, b): tile row w (15 - w) / 2 + (b - w - 1)
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + (off_w + b) * LDA);
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + b * H);
#pragma unroll
                        for (int c = 0; c < 4; ++c) a4[c] = __fadd_rn(a4[c], __fmul_rn(xv[c], wv[c]));
                    }
                agg[tI] = a4;
            } else
            if (blk) {   // (g) block mode, I < J: wave w adds the 8 partners of row J*8 + w (J side), then of row I*8 + w (I side)
                TSD_UNIT_GEOM
                const float* wrow = buf + lane * 4;
                const float* xrow = x1s + lane * 4;
                f32x4 wv[8], xv[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    wv[a] = *reinterpret_cast<const f32x4*>(wrow + (a * 8 + wave) * LDA);
                    xv[a] = *reinterpret_cast<const f32x4*>(xrow + (tI * 8 + a) * H);
                }
                {
                    f32x4 a4 = agg[tJ];
#pragma unroll
                    for (int a = 0; a < 8; ++a)
#pragma unroll
                        for (int s = 0; s < 4; ++s) a4[s] = __fadd_rn(a4[s], __fmul_rn(xv[a][s], wv[a][s]));
                    agg[tJ] = a4;
                }
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    wv[b] = *reinterpret_cast<const f32x4*>(wrow + (wave * 8 + b) * LDA);
                    xv[b] = *reinterpret_cast<const f32x4*>(xrow + (tJ * 8 + b) * H);
                }
                {
                    f32x4 a4 = agg[tI];
#pragma unroll
                    for (int b = 0; b < 8; ++b)
#pragma unroll
                        for (int s = 0; s < 4; ++s) a4[s] = __fadd_rn(a4[s], __fmul_rn(xv[b][s], wv[b][s]));
                    agg[tI] = a4;
                }
            } else {   // (g) list mode: wave w owns the rows r with r % 8 == w (agg[r / 8], 4 columns per lane).  A row's
                // partners below it come from pairs (i, r) -- the row is the pair's j -- and precede, in list order, its
                // partners above it, pairs (r, j).  ROW BY ROW (round 5): for each of its (up to eight) rows the wave takes
                // the row's j-side pairs of the tile in ascending order, then its i-side pairs, into a LOCAL copy of the
                // row's sums -- the loop over the rows is unrolled, so agg[] is indexed statically (the round-4 form walked
                // the tile's pairs and indexed agg[] dynamically per message: s_set_gpr_idx moves around every add, 5.9 k
                // cycles per tile of small molecules, tools/trace_unit.py ens8).  Four pairs per round: their eight 16-byte
                // LDS reads are in flight together, the adds stay in order.  Same order per row: the same bits.
                TSD_UNIT_GEOM
                const unsigned ijv = lane < nrows ? (unsigned)s_u[t * UT + lane] : 0xffffu;  // (0xffff: matches no row)
                const int iv = (int)(ijv & 255u), jv = (int)(ijv >> 8);
                const float* wrow = buf + lane * 4;
                const float* xrow = x1s + lane * 4;
                const int wq = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = wq + 8 * k;
                    const unsigned long long mj = __ballot(jv == r), mi = __ballot(iv == r);
                    if ((mj | mi) != 0ull) {
                        f32x4 a4 = agg[k];
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            const int othv = side == 0 ? iv : jv;
                            unsigned long long msk = side == 0 ? mj : mi;
                            while (msk) {
                                int p[4];
                                bool ok[4];
                                p[0] = (int)__builtin_ctzll(msk);
                                ok[0] = true;
                                msk &= msk - 1ull;
#pragma unroll
                                for (int s4 = 1; s4 < 4; ++s4) {
                                    ok[s4] = msk != 0ull;
                                    p[s4] = ok[s4] ? (int)__builtin_ctzll(msk) : p[s4 - 1];  // (a slot past the end re-reads the last pair)
                                    if (ok[s4]) msk &= msk - 1ull;
                                }
                                f32x4 wv[4], xv[4];
#pragma unroll
                                for (int s4 = 0; s4 < 4; ++s4) {
                                    const int oth = __builtin_amdgcn_readlane(othv, p[s4]);
                                    wv[s4] = *reinterpret_cast<const f32x4*>(wrow + p[s4] * LDA);
                                    xv[s4] = *reinterpret_cast<const f32x4*>(xrow + oth * H);
                                }
#pragma unroll
                                for (int s4 = 0; s4 < 4; ++s4)
                                    if (ok[s4]) {
#pragma unroll
                                        for (int c = 0; c < 4; ++c) a4[c] = __fadd_rn(a4[c], __fmul_rn(xv[s4][c], wv[s4][c]));
                                    }
                            }
                        }
                        agg[k] = a4;
                    }
                }
            }
            __syncthreads();  // (every wave is done with the fp32 tile: the next tile's planes go over it)
            UTRACE(5);
            UTRACE_COUNT(8);
        }

        // ---- node chain of block l on the unit's rows (node_role_h's arithmetic on up to four 16-row blocks) ----
        const bool last = l + 1 == A.L;
        HRing<CB16, HRING16_R> rn;
        f32x4 am[RB16][CB16], ax[RB16][CB16];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        auto zero_n = [&]() {
#pragma unroll
            for (int rb = 0; rb < RB16; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB16; ++cb) am[rb][cb] = ax[rb][cb] = zero4;
        };
        {
            TSD_UNIT_GEOM
            float site_m = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) planes_store4(pl, (k * 8 + wave) * LDH + lane * 4, agg[k], site_m);
            site_close(amax, site_m);
            hgemm16_ring_start<CB16, H>(rn, Wl + A.o_lin2_w, H, col0);
        }
        __syncthreads();
        {
            TSD_UNIT_GEOM
            zero_n();
            hgemm16_ring_run_rb<RB16, CB16, H, UTR>(rn, pl, LDH, am, ax, nrb);
            hgemm16_ring_start<CB16, H>(rn, Wl + A.o_lin_w, H, col0);
        }
        __syncthreads();
        if constexpr (UTR) {   // transposed accumulators: lane = row l15 of the 16-row block, channels 4 q .. 4 q + 3 of the column block
            TSD_UNIT_GEOM
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const int c = col0 + cb * 16 + q * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(Wl + A.o_lin2_b + c);
#pragma unroll
                for (int rb = 0; rb < RB16; ++rb)
                    if (rb < nrb) {
                        f32x4 y4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) y4[r] = sspf(hval4(am[rb][cb], ax[rb][cb], r) + b[r]);
                        planes_store4(pl, (rb * 16 + l15) * LDH + c, y4, amax);
                    }
            }
        } else
        {
            TSD_UNIT_GEOM
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const float b = Wl[A.o_lin2_b + col0 + cb * 16 + l15];
#pragma unroll
                for (int rb = 0; rb < RB16; ++rb)
                    if (rb < nrb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            planes_store1(pl, (rb * 16 + q * 4 + r) * LDH + col0 + cb * 16 + l15,
                                          sspf(hval4(am[rb][cb], ax[rb][cb], r) + b), amax);
                    }
            }
        }
        __syncthreads();
        {
            TSD_UNIT_GEOM
            zero_n();
            hgemm16_ring_run_rb<RB16, CB16, H, UTR>(rn, pl, LDH, am, ax, nrb);
            if (!last) hgemm16_ring_start<CB16, H>(rn, Wl + A.layer_stride + A.o_lin1, H, col0);
        }
        __syncthreads();
        if constexpr (UTR) {
            TSD_UNIT_GEOM
            const float* h_in = (l == 0 ? A.z + m * A.nh_stride : hm) + (size_t)n0 * H;
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const int c = col0 + cb * 16 + q * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(Wl + A.o_lin_b + c);
#pragma unroll
                for (int rb = 0; rb < RB16; ++rb)
                    if (rb < nrb) {
                        const int row = rb * 16 + l15;
                        f32x4 hn = {0.f, 0.f, 0.f, 0.f};
                        if (row < na) {
                            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(h_in + (size_t)row * H + c);
#pragma unroll
                            for (int r = 0; r < 4; ++r) hn[r] = hi4[r] + (hval4(am[rb][cb], ax[rb][cb], r) + b[r]);
                            *reinterpret_cast<f32x4*>(hm + (size_t)(n0 + row) * H + c) = hn;
                        }
                        if (!last) planes_store4(pl, row * LDH + c, hn, amax);
                    }
            }
        } else
        {
            TSD_UNIT_GEOM
            const float* h_in = (l == 0 ? A.z + m * A.nh_stride : hm) + (size_t)n0 * H;
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const int c = col0 + cb * 16 + l15;
                const float b = Wl[A.o_lin_b + c];
#pragma unroll
                for (int rb = 0; rb < RB16; ++rb)
                    if (rb < nrb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = rb * 16 + q * 4 + r;
                            float hn = 0.0f;
                            if (row < na) {
                                hn = h_in[(size_t)row * H + c] + (hval4(am[rb][cb], ax[rb][cb], r) + b);
                                hm[(size_t)(n0 + row) * H + c] = hn;
                            }
                            if (!last) planes_store1(pl, row * LDH + c, hn, amax);
                        }
                    }
            }
        }
        if (last) {
            UTRACE(6);
            break;
        }
        // (the stores of h are younger than the ring of lin1 issued above: they only make its counted waits conservative)
        __syncthreads();
        {
            TSD_UNIT_GEOM
            zero_n();
            hgemm16_ring_run_rb<RB16, CB16, H, UTR>(rn, pl, LDH, am, ax, nrb);
            // x1 of the next block: straight into the LDS copy (every wave is past its reads of the old x1: the
            // barriers of the node chain lie between); row blocks without atoms keep their zeros
            if constexpr (UTR) {
#pragma unroll
                for (int rb = 0; rb < RB16; ++rb)
                    if (rb < nrb) {
#pragma unroll
                        for (int cb = 0; cb < CB16; ++cb) {
                            f32x4 x4;
#pragma unroll
                            for (int r = 0; r < 4; ++r) x4[r] = hval4(am[rb][cb], ax[rb][cb], r);
                            *reinterpret_cast<f32x4*>(x1s + (rb * 16 + l15) * H + col0 + cb * 16 + q * 4) = x4;
                        }
                    }
            } else {
#pragma unroll
            for (int rb = 0; rb < RB16; ++rb)
                if (rb < nrb) {
#pragma unroll
                    for (int cb = 0; cb < CB16; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            x1s[(rb * 16 + q * 4 + r) * H + col0 + cb * 16 + l15] = hval4(am[rb][cb], ax[rb][cb], r);
                }
            }
        }
        __syncthreads();
        UTRACE(6);
    }
#undef TSD_UNIT_GEOM
    if (A.l_end < A.L && A.x1_io != nullptr) {  // a partial run hands x1 of block l_end to the next launch
        float* x_out = A.x1_io + m * A.nh_stride + (size_t)n0 * H;
        for (int idx = tid; idx < na * C4; idx += NT)
            *reinterpret_cast<f32x4*>(x_out + idx * 4) = *reinterpret_cast<const f32x4*>(x1s + idx * 4);
    }
    range_report(amax, A.status);
    UTRACE_FLUSH;
}


size_t unit_encoder_lds(int H) {
    return ((size_t)UNA * H + (size_t)UT * ldh_of(H) + U_ROWS) * 4 + (size_t)U_ROWS * 2 + 64 * 2 + (size_t)2 * H * 4 /* s_bias */;
}

bool unit_encoder_supported(const tsd_model_cfg& c) { return c.hidden == 256; }

int launch_unit_encoder(const tsd_model_cfg& c, const tsd_batch& b, const float* W16, const float* ea, size_t ea_stride,
                        float* h, size_t nh_stride, int l_begin, int l_end, float* x1_io, int32_t* status, hipStream_t st) {
    if (!unit_encoder_supported(c) || b.unit_node == nullptr || b.num_units <= 0) {
        set_error("internal: the fused encoder needs hidden 256 and the batch's unit partition");
        return TSD_ERR_INVALID;
    }
    if (status == nullptr) {  // (the kernel reports a broken unit partition through it: without the word it would return early, silently)
        set_error("internal: the fused encoder needs a status word");
        return TSD_ERR_INVALID;
    }
    const WeightLayout WL = weight_layout(c);
    UnitArgs A{};
    A.L = c.num_convs;
    A.N = b.num_nodes;
    A.num_units = b.num_units;
    A.M = ckpt_grid_m(b.num_models, b.num_units);
    A.unit_node = b.unit_node;
    A.node_graph = b.node_graph;
    A.graph_ptr = b.graph_ptr;
    A.pair_ptr = b.pair_ptr;
    A.pair2u = b.geo.pair2u;
    A.eu = b.geo.enc_u;
    A.W = W16;
    A.w_stride = WL.total;
    A.layer0 = WL.layer0;
    A.layer_stride = WL.layer_stride;
    A.o_nn0_w = WL.L_nn0f_w;  // (the attribute rows hold s1: folded nn.0, common.hpp FOLDED WEIGHTS)
    A.o_nn0_b = WL.L_nn0f_b;
    A.o_nn2_w = WL.L_nn2_w;
    A.o_nn2_b = WL.L_nn2_b;
    A.o_lin1 = WL.L_lin1_w;
    A.o_lin2_w = WL.L_lin2_w;
    A.o_lin2_b = WL.L_lin2_b;
    A.o_lin_w = WL.L_lin_w;
    A.o_lin_b = WL.L_lin_b;
    A.ea = ea;
    A.ea_stride = ea_stride;
    A.z = b.z;
    A.x1_0 = b.x1_0;
    A.h = h;
    A.nh_stride = nh_stride;
    A.conv_cutoff = c.conv_cutoff;
    A.smooth = c.smooth_conv;
    A.l_begin = l_begin;
    A.l_end = l_end;
    A.x1_io = x1_io;
    A.status = status;
    const size_t lds = unit_encoder_lds(c.hidden);
    static DeviceOnce once;
    int r = allow_lds(unit_encoder_kernel<256>, lds, once);
    if (r) return r;
    hipLaunchKernelGGL(unit_encoder_kernel<256>, dim3(b.num_units * b.num_models), dim3(512), lds, st, A);
    TSD_LAUNCH_CHECK("unit_encoder");
    return TSD_OK;
}

}  // namespace tsd

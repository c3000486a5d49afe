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
// Row ownership in the accumulation: wave w = (column half w >> 2, row class w & 3) holds agg[rows = class + 4 k][its 128
// columns, 2 per lane].  A tile's pairs that touch row r are found by ONE ballot over the lanes' (i, j) (lane p holds
// pair p), and walked in ascending p, four at a time (their eight 8-byte LDS reads in flight together; the adds stay in
// order).  No dynamic register index, no atomics.
#include "common.hpp"

namespace tsd {

constexpr int UT = 64;         // undirected pairs per filter tile (two 32-row MFMA blocks)
constexpr int UNA = 64;        // atoms per unit (TSD_UNIT_MAX_NODES)
constexpr int UE_MAX = 2016;   // undirected pairs per unit: one complete 64-atom graph
constexpr int UE_PAD = 2048;

struct UnitArgs {
    int L, N, num_units;
    const int32_t* unit_node;  // [num_units + 1] node offsets
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

// 16-row MFMA GEMM on RB16 row blocks sharing the weight ring (split16.hpp hgemm16_ring_run per row block: the same MFMA
// sequence per output element)
template <int RB16, int CB, int K>
__device__ __forceinline__ void hgemm16_ring_run_rb(HRing<CB, HRING16_R>& r, const Planes& A, int ldh, f32x4 (&accm)[RB16][CB],
                                                    f32x4 (&accx)[RB16][CB]) {
    constexpr int R = HRING16_R, KS = K / 32;
    const int lane = threadIdx.x & 63;
    const int aoff = (lane & 15) * ldh + (lane >> 4) * 8;
    static_for<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int slot = ks % R;
        constexpr int younger = ((ks + R <= KS) ? R : KS - ks) - 1;
        f32x4 ah[RB16], al[RB16];
#pragma unroll
        for (int rb = 0; rb < RB16; ++rb) {
            ah[rb] = *reinterpret_cast<const f32x4*>(A.hi + aoff + rb * 16 * ldh + ks * 32);
            al[rb] = *reinterpret_cast<const f32x4*>(A.lo + aoff + rb * 16 * ldh + ks * 32);
        }
        hring_wait<younger * CB * 2, CB>(r.b[slot]);
#pragma unroll
        for (int rb = 0; rb < RB16; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                accx[rb][cb] = mfma_h16(ah[rb], r.b[slot][cb][1], accx[rb][cb]);
                accm[rb][cb] = mfma_h16(ah[rb], r.b[slot][cb][0], accm[rb][cb]);
                accx[rb][cb] = mfma_h16(al[rb], r.b[slot][cb][0], accx[rb][cb]);
            }
        if constexpr (ks + R < KS)
            hring_issue<CB>(r.b[slot], r.base + (size_t)(ks + R) * r.step_bytes, r.voff, r.plane_bytes, r.cb_bytes);
    });
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// A copy of a lane value the compiler must treat as new: the address arithmetic that hangs off it is recomputed where
// it is used (a few VALU instructions per phase) instead of being hoisted out of the block and tile loops as hundreds
// of loop-invariant registers (first build of this kernel: 256 VGPRs + 1 KB of scratch, 214 spill stores in the prologue).
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <int H>
__global__ __launch_bounds__(2 * H) void unit_encoder_kernel(UnitArgs A) {
    static_assert(H == 256, "the unit encoder is built for hidden 256 (8 waves: 2 column halves x 4 row classes)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDH = ldh_of(H), LDA = H + 4, NT = 2 * H, C4 = H / 4;
    constexpr int NIT = UT * C4 / NT;  // float4 loads per thread and attribute tile (8)
    constexpr int RB16 = UNA / 16, CB16 = 2;
    float* x1s = smem;                               // [UNA][H] fp32
    float* tile = x1s + UNA * H;                     // planes (UT x LDH floats) / fp32 filter tile [UT][LDA]
    float* s_c = tile + UT * LDH;                    // [UE_PAD] cutoff weights of the unit's pairs
    uint16_t* s_ij = reinterpret_cast<uint16_t*>(s_c + UE_PAD);  // [UE_PAD] local (i | j << 8)
    const Planes pl = planes_at(tile, UT, LDH);
    float* buf = tile;

    const int u = blockIdx.x;
    const size_t m = blockIdx.y;
    const int tid = threadIdx.x;
    const int n0 = A.unit_node[u], n1 = A.unit_node[u + 1], na = n1 - n0;
    if (na <= 0) return;
    const int e0 = A.eu.row_ptr[n0], e1 = A.eu.row_ptr[n1], ne = e1 - e0;
    if (na > UNA || ne > UE_MAX || ne < 0) {  // (the host built the units: never expected)
        if (tid == 0) atomicOr(A.status, TSD_STATUS_INTERNAL);
        return;
    }
    const float* Wm = A.W + m * A.w_stride;
    const float* ea = A.ea + m * A.ea_stride + (size_t)e0 * H;
    float* hm = A.h + m * A.nh_stride;
    float amax = 0.0f;

    // ---- per-unit staging: cutoff weights and local end points of the pairs, x1 of block l_begin ----
    for (int e = tid; e < ne; e += NT) {
        s_c[e] = cutoff_weight(A.eu.dist[e0 + e], A.conv_cutoff, A.smooth);
        s_ij[e] = (uint16_t)((A.eu.src[e0 + e] - n0) | ((A.eu.dst[e0 + e] - n0) << 8));
    }
    {
        const float* x_in = (A.l_begin == 0 ? A.x1_0 : A.x1_io) + m * A.nh_stride + (size_t)n0 * H;
        for (int idx = tid; idx < na * C4; idx += NT)
            *reinterpret_cast<f32x4*>(x1s + idx * 4) = *reinterpret_cast<const f32x4*>(x_in + idx * 4);
    }
    const int ntile = (ne + UT - 1) / UT;

    // accumulation geometry of a wave: 128 columns (2 per lane), rows = class + 4 k;  GEMM geometry: 32 output columns
    // per wave (every phase derives its lane geometry from an opaque copy of the thread index: see opaque())
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
        f32x2 agg[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) agg[k] = f32x2{0.0f, 0.0f};
        f32x4 v[NIT];  // the attribute rows of the next tile (row = wave + 8 it, one 1-KiB row per wave-load)
        auto fetch = [&](int t) {
            TSD_UNIT_GEOM
            const int nr = min(UT, ne - t * UT);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int r = wave + it * (NT / 64);
                v[it] = *reinterpret_cast<const f32x4*>(ea + (size_t)(t * UT + min(r, nr - 1)) * H + lane * 4);
            }
        };
        if (ntile > 0) fetch(0);
        for (int t = 0; t < ntile; ++t) {
            const int nrows = min(UT, ne - t * UT);
            HRing<1, HRING_R> rg;
            f32x16 accm[2][1], accx[2][1];
            {   // (a) attribute tile -> planes (rows past the end: zeros)
                TSD_UNIT_GEOM
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int r = wave + it * (NT / 64);
                    const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
                    planes_store4(pl, r * LDH + lane * 4, r < nrows ? v[it] : zz, amax);
                }
                hgemm_ring_start<1, H>(rg, nn0_w, H, col0);
            }
            __syncthreads();
            {   // (b) GEMM nn.0
                TSD_UNIT_GEOM
                hzero(accm, accx);
                hgemm_ring_run<2, 1, H>(rg, pl, LDH, accm, accx);
                hgemm_ring_start<1, H>(rg, nn2_w, H, col0);
            }
            __syncthreads();
            {   // (c) shifted softplus -> planes
                TSD_UNIT_GEOM
                const float b0 = Wl[A.o_nn0_b + col];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        planes_store1(pl, (rb * 32 + acc_row(r, hi)) * LDH + col, sspf(hval(accm[rb][0], accx[rb][0], r) + b0), amax);
            }
            __syncthreads();
            {   // (d) GEMM nn.2
                hzero(accm, accx);
                hgemm_ring_run<2, 1, H>(rg, pl, LDH, accm, accx);
            }
            // (e) the next tile's attribute rows: in flight under the epilogue and the accumulation
            if (t + 1 < ntile) fetch(t + 1);
            __syncthreads();
            {   // (f) W = (nn.2 + b) * C as fp32 rows over the planes
                TSD_UNIT_GEOM
                const float b2 = Wl[A.o_nn2_b + col];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rb * 32 + acc_row(r, hi);
                        buf[row * LDA + col] = (hval(accm[rb][0], accx[rb][0], r) + b2) * s_c[min(t * UT + row, UE_PAD - 1)];
                    }
            }
            __syncthreads();
            {   // (g) messages of both end points into agg, pairs in list order per row
                TSD_UNIT_GEOM
                const unsigned ijv = lane < nrows ? (unsigned)s_ij[t * UT + lane] : 0xffffu;
                const int iv = (int)(ijv & 255u), jv = (int)(ijv >> 8);
                const float* bufc = buf + acol;
                const float* x1c = x1s + acol;
                static_for<0, 16>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int r = cls + 4 * k;
                    unsigned long long msk = __ballot(iv == r || jv == r);
                    while (msk) {
                        int p[4];
                        bool ok[4];
                        p[0] = (int)__builtin_ctzll(msk);
                        ok[0] = true;
                        msk &= msk - 1ull;
#pragma unroll
                        for (int s = 1; s < 4; ++s) {
                            ok[s] = msk != 0ull;
                            p[s] = ok[s] ? (int)__builtin_ctzll(msk) : p[s - 1];  // (a slot past the end re-reads the last pair)
                            if (ok[s]) msk &= msk - 1ull;
                        }
                        f32x2 wv[4], xv[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int ip = __builtin_amdgcn_readlane(iv, p[s]), jp = __builtin_amdgcn_readlane(jv, p[s]);
                            const int other = ip == r ? jp : ip;
                            wv[s] = *reinterpret_cast<const f32x2*>(bufc + p[s] * LDA);
                            xv[s] = *reinterpret_cast<const f32x2*>(x1c + other * H);
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            if (ok[s]) {
                                agg[k][0] = __fadd_rn(agg[k][0], __fmul_rn(xv[s][0], wv[s][0]));
                                agg[k][1] = __fadd_rn(agg[k][1], __fmul_rn(xv[s][1], wv[s][1]));
                            }
                    }
                });
            }
            __syncthreads();  // (every wave is done with the fp32 tile: the next tile's planes go over it)
        }

        // ---- node chain of block l on the unit's rows (node_role_h's arithmetic on four 16-row blocks) ----
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
#pragma unroll
            for (int k = 0; k < 16; ++k) planes_store2(pl, (cls + 4 * k) * LDH + acol, agg[k][0], agg[k][1], amax);
            hgemm16_ring_start<CB16, H>(rn, Wl + A.o_lin2_w, H, col0);
        }
        __syncthreads();
        {
            TSD_UNIT_GEOM
            zero_n();
            hgemm16_ring_run_rb<RB16, CB16, H>(rn, pl, LDH, am, ax);
            hgemm16_ring_start<CB16, H>(rn, Wl + A.o_lin_w, H, col0);
        }
        __syncthreads();
        {
            TSD_UNIT_GEOM
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const float b = Wl[A.o_lin2_b + col0 + cb * 16 + l15];
#pragma unroll
                for (int rb = 0; rb < RB16; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        planes_store1(pl, (rb * 16 + q * 4 + r) * LDH + col0 + cb * 16 + l15,
                                      sspf(hval4(am[rb][cb], ax[rb][cb], r) + b), amax);
            }
        }
        __syncthreads();
        {
            TSD_UNIT_GEOM
            zero_n();
            hgemm16_ring_run_rb<RB16, CB16, H>(rn, pl, LDH, am, ax);
            if (!last) hgemm16_ring_start<CB16, H>(rn, Wl + A.layer_stride + A.o_lin1, H, col0);
        }
        __syncthreads();
        {
            TSD_UNIT_GEOM
            const float* h_in = (l == 0 ? A.z + m * A.nh_stride : hm) + (size_t)n0 * H;
#pragma unroll
            for (int cb = 0; cb < CB16; ++cb) {
                const int c = col0 + cb * 16 + l15;
                const float b = Wl[A.o_lin_b + c];
#pragma unroll
                for (int rb = 0; rb < RB16; ++rb)
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
        if (last) break;
        // (the stores of h are younger than the ring of lin1 issued above: they only make its counted waits conservative)
        __syncthreads();
        {
            TSD_UNIT_GEOM
            zero_n();
            hgemm16_ring_run_rb<RB16, CB16, H>(rn, pl, LDH, am, ax);
            // x1 of the next block: straight into the LDS copy (every wave is past its reads of the old x1: the
            // barriers of the node chain lie between)
#pragma unroll
            for (int rb = 0; rb < RB16; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB16; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        x1s[(rb * 16 + q * 4 + r) * H + col0 + cb * 16 + l15] = hval4(am[rb][cb], ax[rb][cb], r);
        }
        __syncthreads();
    }
#undef TSD_UNIT_GEOM
    if (A.l_end < A.L && A.x1_io != nullptr) {  // a partial run hands x1 of block l_end to the next launch
        float* x_out = A.x1_io + m * A.nh_stride + (size_t)n0 * H;
        for (int idx = tid; idx < na * C4; idx += NT)
            *reinterpret_cast<f32x4*>(x_out + idx * 4) = *reinterpret_cast<const f32x4*>(x1s + idx * 4);
    }
    range_report(amax, A.status);
}

size_t unit_encoder_lds(int H) {
    return ((size_t)UNA * H + (size_t)UT * ldh_of(H) + UE_PAD) * 4 + (size_t)UE_PAD * 2;
}

bool unit_encoder_supported(const tsd_model_cfg& c) { return c.hidden == 256; }

int launch_unit_encoder(const tsd_model_cfg& c, const tsd_batch& b, const float* W16, const float* ea, size_t ea_stride,
                        float* h, size_t nh_stride, int l_begin, int l_end, float* x1_io, int32_t* status, hipStream_t st) {
    if (!unit_encoder_supported(c) || b.unit_node == nullptr || b.num_units <= 0) {
        set_error("internal: the fused encoder needs hidden 256 and the batch's unit partition");
        return TSD_ERR_INVALID;
    }
    const WeightLayout WL = weight_layout(c);
    UnitArgs A{};
    A.L = c.num_convs;
    A.N = b.num_nodes;
    A.num_units = b.num_units;
    A.unit_node = b.unit_node;
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
    hipLaunchKernelGGL(unit_encoder_kernel<256>, dim3(b.num_units, b.num_models), dim3(512), lds, st, A);
    TSD_LAUNCH_CHECK("unit_encoder");
    return TSD_OK;
}

}  // namespace tsd

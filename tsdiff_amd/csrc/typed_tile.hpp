// typed_tile.hpp -- one tile of the type-sorted edge embedding on the f16 MFMA pipes (kernels_typed.hip) as a device
// function: the embedding launch (typed_embed_h_kernel) and the one-launch forward (kernels_combo.hip) both run it.
#pragma once
#include "train_internal.hpp"

#ifndef TSD_TYPED_TRANS
#define TSD_TYPED_TRANS 1  // transposed accumulators in the typed embedding tile (0: the round-4 layout; A/B builds)
#endif

namespace tsd {

struct TypedList {
    int n;  // tiles
    const int32_t *slot, *start, *count, *pair, *ni, *nj;
};
struct TypedEmbedW {
    const float *w0, *b0;    // Linear(1, H) of the distance MLP (packed arena, per checkpoint: + m * wstride)
    const float* bucket;     // bucket arena of checkpoint 0; checkpoint m at + m * bstride
    size_t wstride, bstride;
};

// One tile of the embedding launch on the f16 MFMA pipes (PREC_H2, split16.hpp): `w.bucket` and the block-0 filter weights come from the
// f16-plane arenas (tsd_bucket_weights16 / tsd_pack_weights16), the operand tiles live in LDS as two f16 planes, the
// attribute rows leave as f16 planes (common.hpp ATTRIBUTE ROWS AS f16 PLANES), block 0's filter rows as fp32.
template <int H, bool FUSE0>
__device__ __forceinline__ void typed_embed_tile_h(const TypedEmbedW& w, const TypedList& ta, const TypedList& tb, int bx,
                                                   size_t m, const float* __restrict__ pos,
                                                   const int32_t* __restrict__ pair2u, int P,
                                                   const int32_t* __restrict__ attr_row, float* __restrict__ edge_attr,
                                                   size_t out_stride, const EmbedFuse0& f0, int32_t* range_status,
                                                   float* smem) {
    constexpr int T = TSD_EDGE_TILE;
    constexpr int LDA = H + 4, LDH = ldh_of(H), NT = 2 * H, C4 = H / 4;
    // LDS: 37 KB -- the fp32 staging rows of block 0's filter tile lie OVER the planes (one more barrier per tile), so that
    // THREE workgroups share a CU (<= 80 VGPRs: tools/check_regs.py): round 5 measured 8 checkpoints at batch 100 1.271 ->
    // 1.255 ms/step, BASELINE configs[4] 18.29 -> 18.15, batch 100 unchanged (0.1947 / 0.1946); 64-row tiles (one pass over a
    // bucket's matrix per 64 pairs, two workgroups per CU) measured WORSE at batch 100 (0.195 -> 0.200) and with 8
    // checkpoints (1.271 -> 1.281), equal elsewhere: not kept (docs/NOTEBOOK.md)
    static_assert(LDA <= LDH, "the staging rows lie over the planes");
    const Planes pl = planes_at(smem, T, LDH);
    float* stage = smem;  // fp32 rows [T][LDA] (FUSE0; over the planes)
    float* s_d = smem + T * LDH;
    int* s_row = reinterpret_cast<int*>(s_d + T);
    float* s_cw = s_d + 2 * T;  // CFConv cutoff weight of the row (block-0 filters)
    // transposed accumulators (split16.hpp hgemm_ring_run<..., TRANS>, as kernels_combo.hip::filter_role_h): lane = tile row
    // l31, four runs of four consecutive channels; the three bias vectors of the tile's epilogues in LDS
    constexpr bool TR = TSD_TYPED_TRANS != 0;
    float* s_bias = s_cw + T;   // [3][H]: bucket bias, nn.0 bias, nn.2 bias (TR)
    const bool second = bx >= ta.n;
    const TypedList& tl = second ? tb : ta;
    const int t = second ? bx - ta.n : bx;
    const int PU = P / 2;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = (tid >> 6) * 32, col = col0 + l31;
    const int slot = tl.slot[t], start = tl.start[t], nrows = tl.count[t];
    const float* Wt = w.bucket + m * w.bstride + (size_t)slot * ((size_t)H * H + H);
    const float* bt = Wt + (size_t)H * H;
    const float *w0 = w.w0 + m * w.wstride, *b0 = w.b0 + m * w.wstride;
    edge_attr += m * out_stride;
    float amax = 0.0f;
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
        if constexpr (FUSE0) s_cw[tid] = tid < nrows ? cutoff_weight(sqrtf(d2), f0.conv_cutoff, f0.smooth) : 0.0f;
    }
    __syncthreads();
    {   // (a tile none of whose pairs is an edge at this step: nothing to do)
        int any = 0;
        for (int r = 0; r < nrows; ++r) any |= s_row[r] >= 0;
        if (!any) return;
    }
    HRing<1, HRING_R> rg;
    const float bias_t = TR ? 0.0f : bt[col];
    if constexpr (TR) {
        for (int c = tid; c < H; c += NT) {
            s_bias[c] = bt[c];
            if constexpr (FUSE0) {
                s_bias[H + c] = (f0.nn0_b + m * w.wstride)[c];
                s_bias[2 * H + c] = (f0.nn2_b + m * w.wstride)[c];
            }
        }
    }
    {   // Linear(1,H) + swish: thread = (channel pair, quarter of the tile's rows)
        const int c = (tid % (H / 2)) * 2, r0 = (tid / (H / 2)) * (T / 4);
        const float wa = w0[c], wb = w0[c + 1], ba = b0[c], bb = b0[c + 1];
#pragma unroll
        for (int r = r0; r < r0 + T / 4; ++r) {
            const float d = s_d[r];
            planes_store2(pl, r * LDH + c, swishf(wa * d + ba), swishf(wb * d + bb), amax);
        }
    }
    hgemm_ring_start<1, H>(rg, Wt, H, col0);  // (behind the stage above: its registers are free now)
    __syncthreads();
    f32x16 accm[1][1], accx[1][1];
    hzero(accm, accx);
    hgemm_ring_run<1, 1, H, false, TR>(rg, pl, LDH, accm, accx);
    const bool fuse = FUSE0 && !second;
    const float *nn0_w = f0.nn0_w + m * w.wstride, *nn0_b = f0.nn0_b + m * w.wstride;
    const float *nn2_w = f0.nn2_w + m * w.wstride, *nn2_b = f0.nn2_b + m * w.wstride;
    if constexpr (FUSE0) {
        if (fuse) hgemm_ring_start<1, H>(rg, nn0_w, H, col0);
    }
    __syncthreads();
    // s1 goes to the LDS planes -- the operand of block 0's filter GEMM on this tile, and the FORM the attribute rows are
    // stored in (common.hpp ATTRIBUTE ROWS AS f16 PLANES: every consumer wants the two f16 planes, so the row is converted once, here, with the
    // range and low-side checks of split16.hpp: a lane holds 16 channels of one row = one conversion site)
    if constexpr (TR) {
        const float* bb = s_bias + col0 + 4 * hi;
        f32x4 bn = *reinterpret_cast<const f32x4*>(bb);
        float site_m = 0.0f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 bv = bn;
            if (g4 < 3) bn = *reinterpret_cast<const f32x4*>(bb + 8 * (g4 + 1));
            f32x4 v4;
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = swishf(hval(accm[0][0], accx[0][0], 4 * g4 + r) + bv[r]);
            planes_store4(pl, l31 * LDH + col0 + 8 * g4 + 4 * hi, v4, site_m);
        }
        if (l31 < nrows && s_row[l31] >= 0) site_close(amax, site_m);  // (rows that are no edge now are dropped)
    } else
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, hi);
        planes_store1(pl, row * LDH + col, swishf(hval(accm[0][0], accx[0][0], r) + bias_t), amax);
    }
    __syncthreads();
    // the attribute tile: whole 1-KiB rows -- 512 bytes of the high plane, 512 of the low one -- to the rows of the edges the
    // pairs are at this step
    for (int idx = tid; idx < nrows * C4; idx += NT) {
        const int r = idx / C4, c4 = idx % C4;
        const int row = s_row[r];
        if (row >= 0) {
            const f16* src = (c4 < C4 / 2 ? pl.hi : pl.lo) + r * LDH + (c4 & (C4 / 2 - 1)) * 8;
            store_stream16(edge_attr + (size_t)row * H + c4 * 4, *reinterpret_cast<const f32x4*>(src));
        }
    }
    if constexpr (FUSE0) {
        if (!fuse) {
            range_report(amax, range_status);
            return;
        }
        // filter role of interaction block 0 on this tile (kernels_combo.hip::filter_role_h)
        float* wf = f0.wf + m * f0.wf_stride;
        const float bb0 = TR ? 0.0f : nn0_b[col];
        hzero(accm, accx);
        hgemm_ring_run<1, 1, H, false, TR>(rg, pl, LDH, accm, accx);
        hgemm_ring_start<1, H>(rg, nn2_w, H, col0);
        __syncthreads();
        if constexpr (TR) {
            const float* bb = s_bias + H + col0 + 4 * hi;
            f32x4 bn = *reinterpret_cast<const f32x4*>(bb);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bv = bn;
                if (g4 < 3) bn = *reinterpret_cast<const f32x4*>(bb + 8 * (g4 + 1));
                f32x4 y4;
#pragma unroll
                for (int r = 0; r < 4; ++r) y4[r] = sspf(hval(accm[0][0], accx[0][0], 4 * g4 + r) + bv[r]);
                planes_store4(pl, l31 * LDH + col0 + 8 * g4 + 4 * hi, y4, amax);
            }
        } else
#pragma unroll
        for (int r = 0; r < 16; ++r)
            planes_store1(pl, acc_row(r, hi) * LDH + col, sspf(hval(accm[0][0], accx[0][0], r) + bb0), amax);
        __syncthreads();
        const float bb2 = TR ? 0.0f : nn2_b[col];
        hzero(accm, accx);
        hgemm_ring_run<1, 1, H, false, TR>(rg, pl, LDH, accm, accx);
        __syncthreads();  // (every wave has read its last operand fragment: the planes become the staging rows)
        if constexpr (TR) {
            const float* bb = s_bias + 2 * H + col0 + 4 * hi;
            const float cw = s_cw[l31];
            f32x4 bn = *reinterpret_cast<const f32x4*>(bb);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bv = bn;
                if (g4 < 3) bn = *reinterpret_cast<const f32x4*>(bb + 8 * (g4 + 1));
                f32x4 w4;
#pragma unroll
                for (int r = 0; r < 4; ++r) w4[r] = (hval(accm[0][0], accx[0][0], 4 * g4 + r) + bv[r]) * cw;
                *reinterpret_cast<f32x4*>(stage + l31 * LDA + col0 + 8 * g4 + 4 * hi) = w4;
            }
        } else
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            stage[row * LDA + col] = (hval(accm[0][0], accx[0][0], r) + bb2) * s_cw[row];
        }
        __syncthreads();
        for (int idx = tid; idx < nrows * C4; idx += NT) {
            const int r = idx / C4, c4 = idx % C4;
            const int row = s_row[r];
            if (row >= 0)
                store_stream16(wf + (size_t)row * H + c4 * 4, *reinterpret_cast<const f32x4*>(stage + r * LDA + c4 * 4));
        }
    }
    range_report(amax, range_status);
}


}  // namespace tsd

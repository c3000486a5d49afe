// common.hpp -- shared device/host helpers of libtsdiff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tsdiff_hip.h"

namespace tsd {

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define TSD_HIP(call)                                              \
    do {                                                           \
        int _r = ::tsd::check_hip((call), #call);                  \
        if (_r != TSD_OK) return _r;                               \
    } while (0)

#define TSD_LAUNCH_CHECK(name) TSD_HIP((hipGetLastError()))

#define TSD_REQUIRE(cond, ...)                                     \
    do {                                                           \
        if (!(cond)) {                                             \
            ::tsd::set_error(__VA_ARGS__);                         \
            return TSD_ERR_INVALID;                                \
        }                                                          \
    } while (0)

// ---------------------------------------------------------------------------------------------
// packed weight arena (floats).  Dense matrix W[out][in] is stored as Bp[in/4][out][in%4] so that
// MFMA B-operand lane (kgroup, col) fetches one aligned float4 and a wave fetches 1 KiB contiguous.
// ---------------------------------------------------------------------------------------------
struct WeightLayout {
    size_t bond_emb, emlp_w0, emlp_b0, emlp_w1, emlp_b1, atom_emb, atom_feat;
    size_t ecat_w0, ecat_b0, ecat_w1, ecat_b1;
    size_t layer0, layer_stride;  // per layer: see L_* offsets below
    size_t out_w0, out_b0, out_w1, out_b1, out_w2, out_b2;
    size_t total;
    // offsets inside one layer block
    size_t L_nn0_w, L_nn0_b, L_nn2_w, L_nn2_b, L_lin1_w, L_lin2_w, L_lin2_b, L_lin_w, L_lin_b;
};

inline WeightLayout weight_layout(const tsd_model_cfg& c) {
    WeightLayout L;
    const size_t H = c.hidden, F = c.feat_dim, HH = H * H;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~size_t(3); return r; };  // 16-B aligned
    L.bond_emb = take(100 * H);
    L.emlp_w0 = take(H);
    L.emlp_b0 = take(H);
    L.emlp_w1 = take(HH);
    L.emlp_b1 = take(H);
    L.atom_emb = take(100 * (H / 2));
    L.atom_feat = take((H / 2) * F);
    L.ecat_w0 = take(2 * HH);
    L.ecat_b0 = take(H);
    L.ecat_w1 = take(HH);
    L.ecat_b1 = take(H);
    L.layer0 = o;
    size_t lo = 0;
    auto ltake = [&](size_t n) { size_t r = lo; lo += (n + 3) & ~size_t(3); return r; };
    L.L_nn0_w = ltake(HH);
    L.L_nn0_b = ltake(H);
    L.L_nn2_w = ltake(HH);
    L.L_nn2_b = ltake(H);
    L.L_lin1_w = ltake(HH);
    L.L_lin2_w = ltake(HH);
    L.L_lin2_b = ltake(H);
    L.L_lin_w = ltake(HH);
    L.L_lin_b = ltake(H);
    L.layer_stride = lo;
    o += lo * (size_t)c.num_convs;
    L.out_w0 = take(2 * HH);
    L.out_b0 = take(H);
    L.out_w1 = take(HH / 2);
    L.out_b1 = take(H / 2);
    L.out_w2 = take(H / 2);
    L.out_b2 = take(4);
    L.total = o;
    return L;
}

inline bool hidden_supported(int H) { return H == 64 || H == 128 || H == 256; }

// ---------------------------------------------------------------------------------------------
// device math (fp32, no fast-math)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float swishf(float x) {  // reference utils/activation_functions.py:10-11
    return x / (1.0f + expf(-x));
}
__device__ __forceinline__ float sspf(float x) {  // reference models/encoder/schnet.py:65-71
    // F.softplus(beta=1, threshold=20) - log(2)
    float sp = (x > 20.0f) ? x : log1pf(expf(x));
    return sp - 0.69314718055994530942f;
}

// ---------------------------------------------------------------------------------------------
// MFMA tile GEMM:  acc[T x NOUT] += A[T x K] * W^T   (A in LDS, W packed in global/L2)
//
// 32x32x2 f32 MFMA: A operand lane l holds A[i = l&31][k = l>>5], B operand B[k = l>>5][j = l&31],
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16).
// k is permuted in groups of 8: MFMA step s of k-block kb consumes k = kb*8 + s (lanes 0-31) and
// k = kb*8 + 4 + s (lanes 32-63); A and B use the same permutation so the contraction is unchanged.
// Each lane therefore reads ONE float4 of A (LDS, ds_read_b128) and one float4 of B per column
// block (global_load_dwordx4, a wave reads 1 KiB contiguous) per 4 MFMAs.
//
// RB = T/32 row blocks, CB = column blocks (32 wide) owned by this wave starting at col0.
// ---------------------------------------------------------------------------------------------
template <int RB, int CB, int K>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ ldsA, int lda,
                                          const float* __restrict__ Bp, int nout, int col0,
                                          f32x16 (&acc)[RB][CB]) {
    const int lane = threadIdx.x & 63;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const float* aptr = ldsA + l31 * lda + hi * 4;
    const f32x4* bptr = reinterpret_cast<const f32x4*>(Bp) + (size_t)hi * nout + col0 + l31;
    constexpr int KB = K / 8;
    f32x4 bcur[CB], bnxt[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) bcur[cb] = bptr[cb * 32];
#pragma unroll 2
    for (int kb = 0; kb < KB; ++kb) {
        if (kb + 1 < KB) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) bnxt[cb] = bptr[(size_t)(kb + 1) * 2 * nout + cb * 32];
        }
        f32x4 a[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
            a[rb] = *reinterpret_cast<const f32x4*>(aptr + rb * 32 * lda + kb * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][s], bcur[cb][s], acc[rb][cb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) bcur[cb] = bnxt[cb];
    }
}

template <int RB, int CB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[RB][CB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.0f;
}

// row of accumulator register r for this lane (within a 32-row block)
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

}  // namespace tsd

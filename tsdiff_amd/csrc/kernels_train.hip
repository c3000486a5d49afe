// kernels_train.hip -- primitives of the training step (BASELINE config 4; reference train.py:124-152,
// models/epsnet/condensenc.py:267-328): the dense layers' forward / dgrad / wgrad as fp32-MFMA tile kernels
// (bias, activation, its adjoint, cutoff mask, residual / accumulation in the epilogues), everything
// graph-shaped (segmented aggregation and its two adjoints, pair products, embedding gathers / scatters,
// distance -> Cartesian chain rule) as plain kernels -- no vendor BLAS.  Two hosts sequence them:
// csrc/train_step.hip (the whole step in C++, what get_loss uses) and tsdiff_amd/train_ops.py (one
// torch.autograd.Function node per operation: the differentiable forward() and the cross-check).
//
// All per-edge work runs on the UNDIRECTED lists (see tsd_geometry): a filter row Wf[u] / score s[u] is
// used by both directed edges (i,j) and (j,i), so its gradient is the sum of both directions.
#include <stdlib.h>
#include <algorithm>

#include "train_internal.hpp"
#include "split16.hpp"

namespace tsd {

// ---------------------------------------------------------------------------------------------
// Dense layers of the training step, hand written (no vendor BLAS):
//   forward  Y = X W^T + b      : row tiles of 32 through LDS, W packed on the fly ([in/4][out][in%4]),
//   dgrad    dX = dY W          : the same kernel with W packed the other way ([out/4][in][out%4]),
//   wgrad    dW = dY^T X        : the row-split kernel further down,
// all on the fp32 MFMA (common.hpp::gemm_tile) whenever in/out are multiples of 128/32; the few odd
// shapes of the network (Linear(1,H), Linear(25,H/2), Linear(H/2,1), small test configs) use the plain
// VALU kernels at the end of this block.
// ---------------------------------------------------------------------------------------------
// Bp[k/4][n][k%4] = M[n][k]   (transposed == false: M = W [nout = n][nin = k], forward)
// Bp[k/4][n][k%4] = M[k][n]   (transposed == true : M = W [k = out][n = in],   dgrad)
__global__ void pack_any_kernel(const float* __restrict__ M, float* __restrict__ Bp, int ncols /*n*/, int kdim,
                                int transposed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ncols * kdim) return;
    const int s = idx & 3;
    const int n = (idx >> 2) % ncols;
    const int k = ((idx >> 2) / ncols) * 4 + s;
    Bp[idx] = transposed ? M[(size_t)k * ncols + n] : M[(size_t)n * kdim + k];
}

// Y[rows, nout] = A[rows, K] * Bp (+ bias) (+ R): 4 waves, each CB column blocks of 32; a workgroup owns
// CB*128 columns starting at blockIdx.y * CB*128 (short problems are split over the columns to fill the chip).
// R (nullable, may alias Y): residual / accumulation source with the layout of Y.
// PF: k-blocks per prefetch chunk of B (common.hpp::gemm_tile).  Short problems (fewer workgroups than CUs, one
// column block per wave) are pure latency chains: they use PF = 16, i.e. at most four deep chunks in flight
// two at a time, instead of sixteen shallow ones.
template <int K, int CB, int PF, int NW>
__global__ __launch_bounds__(64 * NW) void linear_mfma_kernel(int rows, int nout, const float* __restrict__ A,
                                                              const float* __restrict__ Bp, LinEpi epi, float* Y) {
    constexpr int LDA = K + 4;
    constexpr int K4 = K / 4;
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int r0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int col0 = blockIdx.y * (CB * 32 * NW) + (tid >> 6) * (CB * 32);
    const int nrows = min(32, rows - r0);
    {   // A tile -> LDS: all loads of a thread in flight together (a guarded load per iteration makes the
        // compiler wait for each one: 8 serial HBM round trips per workgroup); rows past the end are clamped
        constexpr int NIT = 32 * K4 / NT;
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / K4, c4 = idx % K4;
            v[it] = *reinterpret_cast<const f32x4*>(A + (size_t)(r0 + min(r, nrows - 1)) * K + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, r = idx / K4, c4 = idx % K4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(smem + r * LDA + c4 * 4) = r < nrows ? v[it] : z;
        }
    }
    __syncthreads();
    f32x16 acc[1][CB];
    zero_acc(acc);
    gemm_tile<1, CB, K>(smem, LDA, Bp, nout, col0, acc);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int col = col0 + cb * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, hi);
            if (row < nrows) epi_store(epi, acc[0][cb][r], r0 + row, col, (size_t)(r0 + row) * nout + col, Y);
        }
    }
}

template <int K, int CB, int NW, int PF>
static int launch_linear_mfma(int rows, int nout, const float* A, const float* Bp, const LinEpi& epi, float* Y,
                              hipStream_t st) {
    const size_t lds = (size_t)32 * (K + 4) * 4;
    static DeviceOnce once;
    {
        int r = allow_lds(linear_mfma_kernel<K, CB, PF, NW>, lds, once);
        if (r) return r;
    }
    hipLaunchKernelGGL((linear_mfma_kernel<K, CB, PF, NW>), dim3((rows + 31) / 32, nout / (CB * 32 * NW)),
                       dim3(64 * NW), lds, st, rows, nout, A, Bp, epi, Y);
    TSD_LAUNCH_CHECK("linear_mfma");
    return TSD_OK;
}

static bool mfma_shape(int K, int NOUT) { return (K == 128 || K == 256 || K == 512) && (NOUT == 128 || NOUT == 256 || NOUT == 512); }

static int dispatch_linear_mfma(int rows, int K, int NOUT, const float* A, const float* Bp, const LinEpi& epi, float* Y,
                                hipStream_t st) {
    // fewer than one workgroup per CU with full-width tiles: 4 waves x 32 columns, the columns split over
    // grid.y, deep pinned prefetch (a latency chain)
    const bool small = (rows + 31) / 32 < 256;
#define TSD_LM(KK)                                                                                              \
    if (K == KK) {                                                                                              \
        if (small) return launch_linear_mfma<KK, 1, 4, 16>(rows, NOUT, A, Bp, epi, Y, st);                      \
        if (NOUT == 128) return launch_linear_mfma<KK, 1, 4, 4>(rows, NOUT, A, Bp, epi, Y, st);                 \
        if (NOUT == 256) return launch_linear_mfma<KK, 1, 8, 4>(rows, NOUT, A, Bp, epi, Y, st);                 \
        return launch_linear_mfma<KK, 2, 8, 4>(rows, NOUT, A, Bp, epi, Y, st);                                  \
    }
    TSD_LM(128) TSD_LM(256) TSD_LM(512)
#undef TSD_LM
    set_error("internal: no MFMA instance for K=%d N=%d", K, NOUT);
    return TSD_ERR_INVALID;
}

// odd shapes: one thread per output element
// C[r, n] = sum_k A[r, k] * (transposed ? M[k, n] : M[n, k]) (+ bias[n])
__global__ void linear_naive_kernel(int rows, int K, int N, const float* __restrict__ A, const float* __restrict__ M,
                                    int transposed, LinEpi epi, float* C) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)rows * N) return;
    const int r = (int)(t / N), n = (int)(t % N);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(size_t)r * K + k], transposed ? M[(size_t)k * N + n] : M[(size_t)n * K + k], s);
    epi_store(epi, s, r, n, (size_t)t, C);
}
// dW[o, i] = sum_r dY[r, o] * X[r, i]: one workgroup per output element, tree over the rows
__global__ __launch_bounds__(256) void wgrad_naive_kernel(int rows, int in, int out, const float* __restrict__ dY,
                                                          const float* __restrict__ X, float* __restrict__ dW,
                                                          int accumulate) {
    __shared__ float sm[256];
    const int o = blockIdx.x / in, i = blockIdx.x % in;
    float s = 0.0f;
    for (int r = threadIdx.x; r < rows; r += 256) s = fmaf(dY[(size_t)r * out + o], X[(size_t)r * in + i], s);
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) dW[blockIdx.x] = accumulate ? dW[blockIdx.x] + sm[0] : sm[0];
}

// ---------------------------------------------------------------------------------------------
// wgrad: dW[out,in] = dY[rows,out]^T X[rows,in] for the per-edge layers, where rows (26 k at batch 200)
// is the contraction and out/in are 128..512 -- a BLAS call runs these as 4-8 workgroups (300 us each).
// Here: grid (in/128, out/128, S row splits); a workgroup of 4 waves (2x2, each 64 out x 64 in = four
// 32x32 accumulators) streams its row range through LDS in 32-row tiles (register-prefetched), the
// S partial [out,in] blocks are then summed in split order by wgrad_reduce_kernel (deterministic).
// A operand = dY^T: lane (i = out, k = row) reads dYs[row][out] -- consecutive lanes, consecutive words.
// ---------------------------------------------------------------------------------------------
#ifndef TSD_WG_T
#define TSD_WG_T 32
#endif
constexpr int WG_T = TSD_WG_T;      // rows per tile
constexpr int WG_LD = 128 + 4;      // LDS row stride (floats)
// With `bias_part` != NULL the workgroups of in-block 0 also sum their dY tile's columns (the bias gradient):
// partial [split][out], reduced together with the weight partials.
// Workgroup = 8 waves on one 128 x 128 output block: wave (g, c) owns out rows [64 g, 64 g + 64) x in columns
// [32 c, 32 c + 32) -- two waves per SIMD hide each other's LDS / barrier latency (the 4-wave form, one wave per
// SIMD with a 64 x 64 block each, ran at 0.47 of the MFMA peak on the 22 k-row layers).
// A batch of equally shaped problems shares one launch: blockIdx.z = item * S + split (WgradBatch).
constexpr int WG_NT = 512;
struct WgradItem {
    const float *dY, *X;
};
constexpr int WG_BATCH_MAX = 24;
struct WgradBatch {
    WgradItem it[WG_BATCH_MAX];
};
__device__ __forceinline__ void wgrad_body(int rows, int in, int out, int rows_per_split, int split,
                                           const float* __restrict__ dY, const float* __restrict__ X,
                                           float* __restrict__ part, float* __restrict__ bias_part, float* sY,
                                           float* sX) {
    const int in0 = blockIdx.x * 128, out0 = blockIdx.y * 128;
    const int r_begin = split * rows_per_split, r_end = min(rows, r_begin + rows_per_split);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wo = (wave >> 2) * 64, wi = (wave & 3) * 32;  // this wave's 64 x 32 sub-block
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    // each thread stages NQ float4 of dY and NQ float4 of X per tile: (row = idx / 32, col4 = idx % 32)
    constexpr int NQ = WG_T * 32 / WG_NT;
    f32x4 py[NQ], px[NQ];
    auto prefetch = [&](int r0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + q * WG_NT, r = idx >> 5, c4 = idx & 31;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            py[q] = z;
            px[q] = z;
            if (r0 + r < r_end) {
                py[q] = *reinterpret_cast<const f32x4*>(dY + (size_t)(r0 + r) * out + out0 + c4 * 4);
                px[q] = *reinterpret_cast<const f32x4*>(X + (size_t)(r0 + r) * in + in0 + c4 * 4);
            }
        }
    };
    const bool do_bias = bias_part != nullptr && blockIdx.x == 0 && tid < 128;
    float bsum = 0.0f;
    if (r_begin < r_end) prefetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += WG_T) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + q * WG_NT, r = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<f32x4*>(sY + r * WG_LD + c4 * 4) = py[q];
            *reinterpret_cast<f32x4*>(sX + r * WG_LD + c4 * 4) = px[q];
        }
        __syncthreads();
        if (r0 + WG_T < r_end) prefetch(r0 + WG_T);
        if (do_bias) {  // rows past r_end were staged as zeros
#pragma unroll 8
            for (int r = 0; r < WG_T; ++r) bsum += sY[r * WG_LD + tid];
        }
#pragma unroll 8
        for (int kk = 0; kk < WG_T / 2; ++kk) {
            const int row = 2 * kk + hi;
            const float a0 = sY[row * WG_LD + wo + l31], a1 = sY[row * WG_LD + wo + 32 + l31];
            const float b0 = sX[row * WG_LD + wi + l31];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1], 0, 0, 0);
        }
    }
    float* P = part + (size_t)split * out * in;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = out0 + wo + a * 32 + acc_row(r, hi), i = in0 + wi + l31;
            P[(size_t)o * in + i] = acc[a][r];
        }
    if (do_bias) bias_part[(size_t)split * out + out0 + tid] = bsum;
}
__global__ __launch_bounds__(WG_NT) void wgrad_kernel(int rows, int in, int out, int rows_per_split,
                                                      const float* __restrict__ dY, const float* __restrict__ X,
                                                      float* __restrict__ part, float* __restrict__ bias_part) {
    __shared__ __attribute__((aligned(16))) float sY[WG_T * WG_LD];
    __shared__ __attribute__((aligned(16))) float sX[WG_T * WG_LD];
    wgrad_body(rows, in, out, rows_per_split, blockIdx.z, dY, X, part, bias_part, sY, sX);
}
// item k: partials at part + k * S * out * in, bias partials at bias_part + k * S * out (NULL: none)
__global__ __launch_bounds__(WG_NT) void wgrad_batch_kernel(int rows, int in, int out, int rows_per_split, int S,
                                                            WgradBatch b, float* __restrict__ part,
                                                            float* __restrict__ bias_part) {
    __shared__ __attribute__((aligned(16))) float sY[WG_T * WG_LD];
    __shared__ __attribute__((aligned(16))) float sX[WG_T * WG_LD];
    const int item = blockIdx.z / S, split = blockIdx.z - item * S;
    wgrad_body(rows, in, out, rows_per_split, split, b.it[item].dY, b.it[item].X, part + (size_t)item * S * out * in,
               bias_part ? bias_part + (size_t)item * S * out : nullptr, sY, sX);
}
// ---------------------------------------------------------------------------------------------
// The same batched weight gradient on the f16 MFMA pipes (split16.hpp, GRADIENT operands): dW = (s dY)^T X / s with
// s = 2^-e from the running max of the dY tensor that its producer kept (filter_bwd_role_h: amax slots, one per tensor;
// X are saved forward activations, already inside the f16 range when the step's forward raised no flag).
// Both operands are contracted over ROWS, i.e. an MFMA lane needs 8 consecutive rows of one column: a thread loads one
// column of 8 consecutive rows (eight coalesced dword loads per wave row), converts, and writes ONE 16-byte LDS store
// per plane into a column-major tile [plane][128 cols][KT + 8 rows] f16 (column stride 80 B: ds_read/write_b128 conflict
// free).  Workgroup = 4 waves on one 128 x 128 output block, wave (g, c) owns 64 x 64 = 2 x 2 MFMA blocks (two
// accumulator sets: 128 VGPRs), 2 workgroups per CU.  The next tile's 32 loads per thread are in flight under the MFMAs.
// Launch: 1-D grid; the (in/128) x (out/128) blocks of one (item, split) share their dY / X columns pairwise, so they
// are placed on ONE XCD (ids 8 apart) to meet in its L2 -- otherwise every tensor is read from HBM twice, which is what
// bounds this launch (0.6 GB of operands at batch 200 against 40 GFLOP).
// ---------------------------------------------------------------------------------------------
constexpr int WH_KT = 32, WH_LDK = WH_KT + 8, WH_NT = 256;
struct WgradItemH {
    const float *dY, *X;
    const float* amax;  // max |dY| of the tensor as its producer saw it (within a factor 2 below is fine), or NULL: scale 1
};
struct WgradBatchH {
    WgradItemH it[WG_BATCH_MAX];
};
__global__ __launch_bounds__(WH_NT, 2) void wgrad_h2_batch_kernel(int rows, int in, int out, int rows_per_split, int S, int Z,
                                                               WgradBatchH b, float* __restrict__ part,
                                                               float* __restrict__ bias_part) {
    __shared__ __attribute__((aligned(16))) f16 sY[2 * 128 * WH_LDK];
    __shared__ __attribute__((aligned(16))) f16 sX[2 * 128 * WH_LDK];
    __shared__ float s_b[WH_NT];
    const int nbx = in / 128, nby = out / 128, q = nbx * nby;
    const int id = blockIdx.x, slot = id >> 3;
    const int z = (slot / q) * 8 + (id & 7), blk = slot % q;
    if (z >= Z) return;
    const int bx = blk % nbx, by = blk / nbx;
    const int item = z / S, split = z - item * S;
    const float* __restrict__ dY = b.it[item].dY;
    const float* __restrict__ X = b.it[item].X;
    float inv = 1.0f, sc = 1.0f;
    if (b.it[item].amax != nullptr) sc = pow2_scale(*b.it[item].amax, inv);
    part += (size_t)item * S * out * in;
    const int in0 = bx * 128, out0 = by * 128;
    const int r_begin = split * rows_per_split, r_end = min(rows, r_begin + rows_per_split);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wo = (wave >> 1) * 64, wi = (wave & 1) * 64;
    // staging: this thread's column and (wave-uniform) 8-row group 0 / 1 of each 16 rows.  Row bases are scalar, the
    // lane offset is one 32-bit register for all 32 loads of a tile
    const int col = tid & 127, rg = __builtin_amdgcn_readfirstlane(tid >> 7);
    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) accm[a][c][r] = accx[a][c][r] = 0.0f;
    constexpr int NQ = WH_KT / 16;  // 8-row units of this thread per tile and tensor
    float py[NQ][8], px[NQ][8];
    // (buffer loads: scalar resource + scalar row offset + one lane offset register; rows past r_end read as zero)
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, r_end * out * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, r_end * in * 4, 0x00020000);
    const int yoff = (out0 + col) * 4, xoff = (in0 + col) * 4;
    auto prefetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = r0 + u * 16 + rg * 8 + j;
                py[u][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, yoff, r * out * 4, 0));
                px[u][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xoff, r * in * 4, 0));
            }
    };
    const bool do_bias = bias_part != nullptr && bx == 0;
    float bsum = 0.0f;
    if (r_begin < r_end) prefetch(r_begin);
    const int aoff = l31 * WH_LDK + hi * 8;
    for (int r0 = r_begin; r0 < r_end; r0 += WH_KT) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            f16x8 yh, yl, xh, xl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f16 h, l;
                if (do_bias) bsum += py[u][j];
                split1(py[u][j] * sc, h, l);
                yh[j] = h; yl[j] = l;
                split1(px[u][j], h, l);
                xh[j] = h; xl[j] = l;
            }
            const int o = col * WH_LDK + u * 16 + rg * 8;
            *reinterpret_cast<f16x8*>(sY + o) = yh;
            *reinterpret_cast<f16x8*>(sY + 128 * WH_LDK + o) = yl;
            *reinterpret_cast<f16x8*>(sX + o) = xh;
            *reinterpret_cast<f16x8*>(sX + 128 * WH_LDK + o) = xl;
        }
        __syncthreads();
        if (r0 + WH_KT < r_end) prefetch(r0 + WH_KT);
#pragma unroll
        for (int ks = 0; ks < WH_KT / 16; ++ks) {
            f32x4 ah[2], al[2], bh[2], bl[2];
            int ao = aoff;
            asm volatile("" : "+v"(ao));  // (one k-step's fragments at a time: left free, both steps' reads are hoisted)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ah[a] = *reinterpret_cast<const f32x4*>(sY + (wo + a * 32) * WH_LDK + ao + ks * 16);
                al[a] = *reinterpret_cast<const f32x4*>(sY + 128 * WH_LDK + (wo + a * 32) * WH_LDK + ao + ks * 16);
                bh[a] = *reinterpret_cast<const f32x4*>(sX + (wi + a * 32) * WH_LDK + ao + ks * 16);
                bl[a] = *reinterpret_cast<const f32x4*>(sX + 128 * WH_LDK + (wi + a * 32) * WH_LDK + ao + ks * 16);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    accx[a][c] = mfma_h32(ah[a], bl[c], accx[a][c]);
                    accm[a][c] = mfma_h32(ah[a], bh[c], accm[a][c]);
                    accx[a][c] = mfma_h32(al[a], bh[c], accx[a][c]);
                }
        }
    }
    // (buffer stores: one lane offset register, the 64 block / row offsets are scalar)
    float* P = part + (size_t)split * out * in;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(P, 0, out * in * 4, 0x00020000);
    const int wou = __builtin_amdgcn_readfirstlane(out0 + wo), wiu = __builtin_amdgcn_readfirstlane(in0 + wi);
    const int poff = (hi * 4 * in + l31) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = wou + a * 32 + (r >> 2) * 8 + (r & 3), i = wiu + c * 32;  // (acc_row(r, hi) = that + 4 hi)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(hval(accm[a][c], accx[a][c], r) * inv), rp, poff,
                                                      (o * in + i) * 4, 0);
                if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // (eight values at a time leave the accumulator file)
            }
    if (do_bias) {  // the two row groups of a column
        s_b[tid] = bsum;
        __syncthreads();
        if (tid < 128) bias_part[((size_t)item * S + split) * out + out0 + tid] = s_b[tid] + s_b[tid + 128];
    }
}

// sums the S split partials in a fixed order: workgroup = 64 consecutive outputs x 4 split quarters
// (wave q adds splits [q S/4, (q+1) S/4) in order, then ((q0 + q1) + (q2 + q3))).  The weight partials
// [S][nW] and the bias partials [S][nB] (nB may be 0) are reduced by the same launch.
struct WgradOut {
    float *dW, *db;  // db NULL: no bias
};
struct WgradOutBatch {
    WgradOut it[WG_BATCH_MAX];
};
__device__ __forceinline__ void wgrad_reduce_body(int64_t nW, int nB, int S, const float* __restrict__ part,
                                                  const float* __restrict__ bias_part, float* __restrict__ dW,
                                                  float* __restrict__ db, int accumulate);
__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(int64_t nW, int nB, int S,
                                                                 const float* __restrict__ part,
                                                                 const float* __restrict__ bias_part,
                                                                 WgradOutBatch o, int accumulate) {
    const int k = blockIdx.y;
    const bool hb = o.it[k].db != nullptr;
    wgrad_reduce_body(nW, hb ? nB : 0, S, part + (size_t)k * S * nW, bias_part + (size_t)k * S * nB, o.it[k].dW,
                      o.it[k].db, accumulate);
}
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(int64_t nW, int nB, int S, const float* __restrict__ part,
                                                           const float* __restrict__ bias_part,
                                                           float* __restrict__ dW, float* __restrict__ db,
                                                           int accumulate) {
    wgrad_reduce_body(nW, nB, S, part, bias_part, dW, db, accumulate);
}
__device__ __forceinline__ void wgrad_reduce_body(int64_t nW, int nB, int S, const float* __restrict__ part,
                                                  const float* __restrict__ bias_part, float* __restrict__ dW,
                                                  float* __restrict__ db, int accumulate) {
    // Q = blockDim.x / 64 wave groups (4, or 16 for long split lists: the per-lane sum over S / Q partials is a
    // latency chain); group q adds splits [q per, (q+1) per) in order (per = ceil(S / Q): the last groups of a split
    // count that Q does not divide hold fewer, or none), the groups are combined by a fixed pairwise tree
    __shared__ float sm[16][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, Q = blockDim.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const int per = (S + Q - 1) / Q;
    const int k0 = min(S, q * per), k1 = min(S, (q + 1) * per);
    float s = 0.0f;
    if (i < nW) {
        for (int k = k0; k < k1; ++k) s += part[(size_t)k * nW + i];
    } else if (i < nW + nB) {
        for (int k = k0; k < k1; ++k) s += bias_part[(size_t)k * nB + (i - nW)];
    }
    sm[q][lane] = s;
    __syncthreads();
    if (q == 0) {
        float v;
        if (Q == 16) {
            float t[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                t[g] = (sm[4 * g][lane] + sm[4 * g + 1][lane]) + (sm[4 * g + 2][lane] + sm[4 * g + 3][lane]);
            v = (t[0] + t[1]) + (t[2] + t[3]);
        } else {
            v = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
        }
        if (i < nW) dW[i] = accumulate ? dW[i] + v : v;
        else if (i < nW + nB) db[i - nW] = accumulate ? db[i - nW] + v : v;
    }
}
// threads of a reduce launch: 16 wave groups when the split list is long
static inline int reduce_threads(int S) { return S >= 64 ? 1024 : 256; }

// The batched reduce with FOUR consecutive outputs per lane (round 6): 16-byte loads and stores, a quarter of the workgroups.
// Work items [0, nW / 4) are weight quads, [nW / 4, nW / 4 + nB) single bias sums; per element the same additions in the same
// order as wgrad_reduce_body (splits of a wave group in order, the groups by the same fixed tree): bit-identical.
__global__ __launch_bounds__(1024) void wgrad_reduce_batch4_kernel(int64_t nW, int nB, int S, const float* __restrict__ part,
                                                                  const float* __restrict__ bias_part, WgradOutBatch o,
                                                                  int accumulate) {
    __shared__ f32x4 sm[16][64];
    const int k = blockIdx.y;
    part += (size_t)k * S * nW;
    bias_part += (size_t)k * S * nB;
    float* __restrict__ dW = o.it[k].dW;
    float* __restrict__ db = o.it[k].db;
    if (db == nullptr) nB = 0;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6, Q = blockDim.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane, nQ = nW / 4;
    const int per = (S + Q - 1) / Q;
    const int k0 = min(S, q * per), k1 = min(S, (q + 1) * per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < nQ) {
        for (int kk = k0; kk < k1; ++kk) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + (size_t)kk * nW + 4 * i);
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += v[c];
        }
    } else if (i < nQ + nB) {
        for (int kk = k0; kk < k1; ++kk) s[0] += bias_part[(size_t)kk * nB + (i - nQ)];
    }
    sm[q][lane] = s;
    __syncthreads();
    if (q == 0) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (Q == 16) {
                float t[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    t[g] = (sm[4 * g][lane][c] + sm[4 * g + 1][lane][c]) + (sm[4 * g + 2][lane][c] + sm[4 * g + 3][lane][c]);
                v[c] = (t[0] + t[1]) + (t[2] + t[3]);
            } else {
                v[c] = (sm[0][lane][c] + sm[1][lane][c]) + (sm[2][lane][c] + sm[3][lane][c]);
            }
        }
        if (i < nQ) {
            if ((reinterpret_cast<uintptr_t>(dW) & 15) == 0) {
                f32x4* d = reinterpret_cast<f32x4*>(dW + 4 * i);
                if (accumulate) {
                    const f32x4 old = *d;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = old[c] + v[c];
                }
                *d = v;
            } else {  // (a matrix that starts off a 16-byte boundary in the flat gradient: edge_cat.*, behind the 1-float out_b2)
#pragma unroll
                for (int c = 0; c < 4; ++c) dW[4 * i + c] = accumulate ? dW[4 * i + c] + v[c] : v[c];
            }
        } else if (i < nQ + nB) {
            db[i - nQ] = accumulate ? db[i - nQ] + v[0] : v[0];
        }
    }
}

// wgrad of the narrow layers (Linear(1,H), Linear(25,H/2), Linear(H/2,1)): dW[o,i] = sum_r dY[r,o] X[r,i], in <= 32.
// Same two-stage column reduction as the bias gradient with `in` accumulators per thread:
// stage 1: grid (out/64, CS_CHUNKS) -> part[chunk][o*in + i]; stage 2: sums the chunks in order.
// The out == 1 case is the same problem with the roles of dY and X swapped.
constexpr int WS_MAX_IN = 32;
template <int MAXIN>  // 1 (the Linear(1,H) / Linear(H/2,1) layers: no wasted predicated lanes) or WS_MAX_IN
__global__ __launch_bounds__(256) void wgrad_small_kernel(int rows, int in, int out, const float* __restrict__ dY,
                                                          const float* __restrict__ X, float* __restrict__ part,
                                                          float* __restrict__ bias_part /* [chunks][out] or NULL */,
                                                          int bias_of_x /* swapped form (in == 1): [chunks] sums of X */) {
    __shared__ float sm[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + lane;
    const int per = (rows + (int)gridDim.y - 1) / (int)gridDim.y;  // gridDim.y row chunks
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float acc[MAXIN];
    float bsum = 0.0f;  // the bias gradient (column sums of dY) rides along: the same rows are read anyway
#pragma unroll
    for (int i = 0; i < MAXIN; ++i) acc[i] = 0.0f;
    if (o < out)
        for (int r = r0 + w; r < r1; r += 4) {
            const float g = dY[(size_t)r * out + o];
            const float* xr = X + (size_t)r * in;
            bsum += bias_of_x ? xr[0] : g;
#pragma unroll
            for (int i = 0; i < MAXIN; ++i)
                if (i < in) acc[i] = fmaf(g, xr[i], acc[i]);
        }
#pragma unroll
    for (int i = 0; i < MAXIN; ++i) {
        if (i < in) {  // uniform
            __syncthreads();
            sm[w][lane] = acc[i];
            __syncthreads();
            if (w == 0 && o < out)
                part[((size_t)blockIdx.y * out + o) * in + i] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
        }
    }
    if (bias_part) {  // uniform
        __syncthreads();
        sm[w][lane] = bsum;
        __syncthreads();
        const float v = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
        if (bias_of_x) {
            if (w == 0 && o == 0) bias_part[blockIdx.y] = v;
        } else if (w == 0 && o < out) {
            bias_part[(size_t)blockIdx.y * out + o] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// All dense weights of a training step packed by ONE launch (forward layout and transposed dgrad layout):
// the optimizer changes every weight every step, so per-call packing cost 86 launches (0.4 ms) per step.
#define TSD_PACK_MAX 128  // (3.5 KB of kernel arguments per launch; 48 until round 6: a training step packs ~230 items)
struct PackBatch {
    const float* W[TSD_PACK_MAX];
    float* dst[TSD_PACK_MAX];
    int ncols[TSD_PACK_MAX], kdim[TSD_PACK_MAX], transposed[TSD_PACK_MAX];
};
__global__ void pack_batch_kernel(PackBatch b) {
    const int it = blockIdx.y;
    const int ncols = b.ncols[it], kdim = b.kdim[it];
    const float* __restrict__ M = b.W[it];
    float* __restrict__ Bp = b.dst[it];
    if (b.transposed[it] == 2) {  // plain copy of ncols floats (biases, embedding tables)
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < ncols; idx += gridDim.x * blockDim.x) Bp[idx] = M[idx];
        return;
    }
    if (b.transposed[it] >= 3) {  // 3 / 4: the forward / dgrad matrix as the f16 planes of split16.hpp (split16_kernel's layout)
        // one thread = 8 consecutive k of one column: a 32-byte run of a source row (3) or eight coalesced words (4) in,
        // one 16-byte store per plane out
        f16x8* __restrict__ d = reinterpret_cast<f16x8*>(Bp);
        const bool tr = b.transposed[it] == 4;
        for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < ncols * (kdim / 8); u += gridDim.x * blockDim.x) {
            const int col = u % ncols, kg = u / ncols, k0 = kg * 8;
            float a[8];
            if (tr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = M[(size_t)(k0 + j) * ncols + col];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = M[(size_t)col * kdim + k0 + j];  // (a parameter tensor is only 4-byte aligned)
            }
            f16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                h[j] = (f16)a[j];
                l[j] = (f16)((a[j] - (float)h[j]) * SPLIT_SCALE);
            }
            const int ks = kg >> 1, half = kg & 1;
            d[(((size_t)ks * 2 + 0) * 2 + half) * ncols + col] = h;
            d[(((size_t)ks * 2 + 1) * 2 + half) * ncols + col] = l;
        }
        return;
    }
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < ncols * kdim; idx += gridDim.x * blockDim.x) {
        const int s = idx & 3;
        const int n = (idx >> 2) % ncols;
        const int k = ((idx >> 2) / ncols) * 4 + s;
        Bp[idx] = b.transposed[it] ? M[(size_t)k * ncols + n] : M[(size_t)n * kdim + k];
    }
}

int launch_pack_items(int n, const PackItem* items, hipStream_t st) {
    for (int base = 0; base < n; base += TSD_PACK_MAX) {
        PackBatch b;
        const int m = n - base < TSD_PACK_MAX ? n - base : TSD_PACK_MAX;
        int biggest = 0;
        for (int k = 0; k < m; ++k) {
            const PackItem& it = items[base + k];
            b.W[k] = it.src;
            b.dst[k] = it.dst;
            b.transposed[k] = it.mode;
            if (it.mode == 2) {
                b.ncols[k] = it.out;
                b.kdim[k] = 1;
            } else {
                const bool tr = it.mode == 1 || it.mode == 4;
                b.ncols[k] = tr ? it.in : it.out;  // forward: B[k = in][n = out]; dgrad: B[k = out][n = in]
                b.kdim[k] = tr ? it.out : it.in;
            }
            const int sz = it.mode == 2 ? it.out : it.out * it.in;
            biggest = sz > biggest ? sz : biggest;
        }
        const int bx = (biggest + 255) / 256 < 64 ? (biggest + 255) / 256 : 64;
        hipLaunchKernelGGL(pack_batch_kernel, dim3(bx, m), dim3(256), 0, st, b);
    }
    TSD_LAUNCH_CHECK("pack_items");
    return TSD_OK;
}

// ---------------------------------------------------------------------------------------------
// column sums of dY [rows, cols] (bias gradient), deterministic two-stage tree:
// stage 1: grid (cols/64, CS_CHUNKS) workgroups reduce a row chunk each into part[chunk][col];
// stage 2: one thread per column adds the CS_CHUNKS partials in order.
constexpr int CS_CHUNKS = 64;
__global__ __launch_bounds__(256) void colsum_stage1_kernel(int rows, int cols, const float* __restrict__ dy,
                                                            float* __restrict__ part) {
    __shared__ float sm[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    const int per = (rows + CS_CHUNKS - 1) / CS_CHUNKS;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s = 0.0f;
    if (c < cols)
        for (int r = r0 + w; r < r1; r += 4) s += dy[(size_t)r * cols + c];
    sm[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < cols)
        part[(size_t)blockIdx.y * cols + c] =
            (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
__global__ void colsum_stage2_kernel(int cols, const float* __restrict__ part, float* __restrict__ db,
                                     int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.0f;
    for (int k = 0; k < CS_CHUNKS; ++k) s += part[(size_t)k * cols + c];
    db[c] = accumulate ? db[c] + s : s;
}

// activations: kind 0 swish (reference utils/activation_functions.py), 1 shifted softplus (schnet.py:65-71)
__global__ void act_fwd_kernel(int kind, int64_t n, const float* __restrict__ x, float* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    y[i] = act_apply(kind, v);
}
__global__ void act_bwd_kernel(int kind, int64_t n, const float* __restrict__ x, const float* __restrict__ dy,
                               float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    dx[i] = dy[i] * act_deriv(kind, v);
}

// y[r,:] = x[r,:] * emb[idx[r],:]   (edge.py:66-68 d_emb * bond_emb(type))
__global__ void emb_mul_fwd_kernel(int rows, int H, const float* __restrict__ x, const float* __restrict__ emb,
                                   const uint8_t* __restrict__ idx, float* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * H) return;
    const int r = (int)(i / H), c = (int)(i % H);
    y[i] = x[i] * emb[(size_t)idx[r] * H + c];
}
// dx = dy * emb[idx];  demb[idx] += dy * x.  Edge types take ~25 distinct values, so plain atomics from
// every (row, channel) serialise on a few addresses (190 us): each workgroup (64 channels x a chunk of
// rows) first sums per type in LDS, then issues one atomic per (type, channel) it has seen.
constexpr int EMB_ROWS = 100;   // nn.Embedding(100, H)
__global__ __launch_bounds__(256) void emb_mul_bwd_kernel(int rows, int H, int rows_per_wg,
                                                          const float* __restrict__ x,
                                                          const float* __restrict__ emb,
                                                          const uint8_t* __restrict__ idx,
                                                          const float* __restrict__ dy, float* __restrict__ dx,
                                                          float* __restrict__ demb) {
    __shared__ float acc[EMB_ROWS][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < EMB_ROWS * 64; t += 256) acc[t / 64][t % 64] = 0.0f;
    __syncthreads();
    const int r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    if (c < H) {
        for (int r = r0 + w; r < r1; r += 4) {
            const int ty = min((int)idx[r], EMB_ROWS - 1);
            const size_t o = (size_t)r * H + c;
            const float g = dy[o];
            dx[o] = g * emb[(size_t)ty * H + c];
            atomicAdd(&acc[ty][threadIdx.x & 63], g * x[o]);  // LDS atomic, 4 waves share a column
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < EMB_ROWS * 64; t += 256) {
        const int ty = t / 64, cc = blockIdx.x * 64 + (t % 64);
        const float v = acc[ty][t % 64];
        if (v != 0.0f && cc < H) atomicAdd(demb + (size_t)ty * H + cc, v);
    }
}

// y[r,:] = table[idx[r],:]  /  dtable[idx[r],:] += dy[r,:]   (atom_embedding, condensenc.py:193)
__global__ void gather_rows_kernel(int rows, int H, const float* __restrict__ table, const int64_t* __restrict__ idx,
                                   float* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * H) return;
    y[i] = table[(size_t)idx[i / H] * H + (i % H)];
}
__global__ void scatter_rows_add_kernel(int rows, int H, const float* __restrict__ dy, const int64_t* __restrict__ idx,
                                        float* __restrict__ dtable) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * H) return;
    atomicAdd(dtable + (size_t)idx[i / H] * H + (i % H), dy[i]);
}

// x[r,:] *= (dist[r] <= cutoff)    CFConv mask C, forward and backward (schnet.py:97-99)
__global__ void row_mask_kernel(int rows, int H, const float* __restrict__ dist, float cutoff, int smooth,
                                float* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * H) return;
    const float c = cutoff_weight(dist[i / H], cutoff, smooth);
    if (c != 1.0f) x[i] *= c;
}

// adjoint of the aggregation w.r.t. the filter: dWf[u] = dagg[i] * x1[j] + dagg[j] * x1[i], u = {i<j}
// masked != 0: the adjoint of the CFConv cutoff weight is applied as well (dWf *= C(d))
__global__ void aggregate_bwd_filter_kernel(int H, tsd_edges eu, const float* __restrict__ dagg,
                                            const float* __restrict__ x1, float* __restrict__ dWf, int masked,
                                            float cutoff, int smooth) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Eu = *eu.count;
    if (t >= (int64_t)Eu * H) return;
    const int u = (int)(t / H), c = (int)(t % H);
    const int i = eu.src[u], j = eu.dst[u];
    float v = dagg[(size_t)i * H + c] * x1[(size_t)j * H + c] + dagg[(size_t)j * H + c] * x1[(size_t)i * H + c];
    if (masked) v *= cutoff_weight(eu.dist[u], cutoff, smooth);
    dWf[t] = v;
}
int launch_aggregate_bwd_filter(int H, int capacity_u, tsd_edges enc_u, const float* dagg, const float* x1, float* dWf,
                                int masked, float cutoff, int smooth, hipStream_t st) {
    if (capacity_u == 0) return TSD_OK;
    hipLaunchKernelGGL(aggregate_bwd_filter_kernel, dim3((unsigned)(((int64_t)capacity_u * H + 255) / 256)), dim3(256), 0,
                       st, H, enc_u, dagg, x1, dWf, masked, cutoff, smooth);
    TSD_LAUNCH_CHECK("aggregate_bwd_filter");
    return TSD_OK;
}

// p[u,:] = h[i,:] * h[j,:] for the undirected out edges (common.py:226-229 first half of h_pair)
__global__ void pair_product_fwd_kernel(int H, tsd_edges eu, const float* __restrict__ h, float* __restrict__ p) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Eu = *eu.count;
    if (t >= (int64_t)Eu * H) return;
    const int u = (int)(t / H), c = (int)(t % H);
    p[t] = h[(size_t)eu.src[u] * H + c] * h[(size_t)eu.dst[u] * H + c];
}
// dh[i,:] = sum_{e in row i} dp[umap e,:] * h[dst e,:]   (deterministic row gather over the directed CSR)
__global__ void pair_product_bwd_kernel(int N, int H, tsd_edges ed, const float* __restrict__ dp,
                                        const float* __restrict__ h, float* __restrict__ dh) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * H) return;
    const int i = (int)(t / H), c = (int)(t % H);
    float s = 0.0f;
    const int e1 = ed.row_ptr[i + 1];
    for (int e = ed.row_ptr[i]; e < e1; ++e) s += dp[(size_t)ed.umap[e] * H + c] * h[(size_t)ed.dst[e] * H + c];
    dh[t] = s;
}

// node_eq[i] = 2 * sum_{e in row i} ((pos_i - pos_j)/d_e) * s[umap e]     (geometry.py:22-30 on a symmetric s)
__global__ void eq_und_fwd_kernel(int N, tsd_edges ed, const float* __restrict__ pos, const float* __restrict__ s,
                                  float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float ax = 0.f, ay = 0.f, az = 0.f;
    const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
    const int e1 = ed.row_ptr[i + 1];
    for (int e = ed.row_ptr[i]; e < e1; ++e) {
        const int j = ed.dst[e];
        const float w = s[ed.umap[e]] / ed.dist[e];
        ax += (px - pos[3 * j]) * w;
        ay += (py - pos[3 * j + 1]) * w;
        az += (pz - pos[3 * j + 2]) * w;
    }
    out[3 * i] = ax + ax;
    out[3 * i + 1] = ay + ay;
    out[3 * i + 2] = az + az;
}
// ds[u] = 2 * ((pos_i - pos_j)/d_u) . (g_i - g_j)
__global__ void eq_und_bwd_kernel(tsd_edges eu, const float* __restrict__ pos, const float* __restrict__ g,
                                  float* __restrict__ ds) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= *eu.count) return;
    const int i = eu.src[u], j = eu.dst[u];
    const float inv = 2.0f / eu.dist[u];
    ds[u] = inv * ((pos[3 * i] - pos[3 * j]) * (g[3 * i] - g[3 * j]) +
                   (pos[3 * i + 1] - pos[3 * j + 1]) * (g[3 * i + 1] - g[3 * j + 1]) +
                   (pos[3 * i + 2] - pos[3 * j + 2]) * (g[3 * i + 2] - g[3 * j + 2]));
}

// d_gt[u] = |pos0_i - pos0_j|   (get_distance on the clean geometry, condensenc.py:313)
__global__ void pair_distance_kernel(tsd_edges eu, const float* __restrict__ pos, float* __restrict__ d) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= *eu.count) return;
    const int i = eu.src[u], j = eu.dst[u];
    const float dx = pos[3 * i] - pos[3 * j], dy = pos[3 * i + 1] - pos[3 * j + 1], dz = pos[3 * i + 2] - pos[3 * j + 2];
    d[u] = sqrtf(dx * dx + dy * dy + dz * dz);
}

// ---------------------------------------------------------------------------------------------
// legacy dual-encoder pieces (reference models/encoder/gin.py, edge.py:18-41)
// ---------------------------------------------------------------------------------------------
// GINEConv message + aggregation + self term (gin.py:61-73):
//   out[i] = (1 + eps) * x[i] + sum_{e: dst(e) = i} act(x[src(e)] + edge_attr[e])
// Arbitrary directed int64 edge list (the legacy local graph is built by the caller): the self term is
// written first, the messages are added with fp32 atomics.
__global__ void gine_self_kernel(int64_t n, float scale, const float* __restrict__ x, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = scale * x[i];
}
__global__ void gine_message_kernel(int64_t E, int H, int act, const float* __restrict__ x,
                                    const int64_t* __restrict__ ei, const float* __restrict__ edge_attr,
                                    float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * H) return;
    const int64_t e = t / H;
    const int c = (int)(t % H);
    const int64_t j = ei[e], i = ei[E + e];
    float v = x[j * H + c] + edge_attr[t];
    if (act == 1) v = fmaxf(v, 0.0f);
    else if (act == 2) v = sspf(v) + 0.69314718055994530942f;  // softplus
    atomicAdd(out + i * H + c, v);
}

// GaussianSmearingEdgeEncoder (edge.py:18-41, schnet.py:14-23):
//   out[e] = [exp(coeff * (d_e - offset_k)^2), k < K  ||  bond_emb[type_e]]        -> [E, 2K]
__global__ void gaussian_edge_kernel(int64_t E, int K, float coeff, const float* __restrict__ d,
                                     const float* __restrict__ offset, const int64_t* __restrict__ type,
                                     const float* __restrict__ bond_emb, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * 2 * K) return;
    const int64_t e = t / (2 * K);
    const int c = (int)(t % (2 * K));
    if (c < K) {
        const float u = d[e] - offset[c];
        out[t] = expf(coeff * (u * u));
    } else {
        out[t] = bond_emb[type[e] * K + (c - K)];
    }
}


// ---------------------------------------------------------------------------------------------
// Dual-encoder (GeoDiff legacy) network, reference models/epsnet/dualenc.py.
// The local head's GINE message pass over the rows of the symmetric extended edge list, restricted to
// the local edges (type > 0, dualenc.py:1222-1223); edge attributes live once per undirected pair:
//   out[i] = sum_{e in row i, type e > 0} act(x[dst e] + ea_u[umap e])  +  (1 + eps) x[i]
// (gin.py:61-73: the neighbours of target i arrive in ascending source order = the CSR row order).
__device__ __forceinline__ float gine_act(int act, float v) {
    return act == 1 ? fmaxf(v, 0.0f) : act == 2 ? sspf(v) + 0.69314718055994530942f : v;
}
__device__ __forceinline__ float gine_dact(int act, float v) {
    return act == 1 ? (v > 0.0f ? 1.0f : 0.0f) : act == 2 ? __builtin_amdgcn_rcpf(1.0f + fast_exp(-v)) : 1.0f;
}
__global__ void gine_csr_fwd_kernel(int H, int act, float eps, const int32_t* __restrict__ row_ptr,
                                    const int32_t* __restrict__ dst, const int32_t* __restrict__ umap,
                                    const uint8_t* __restrict__ type, const float* __restrict__ ea_u,
                                    const float* __restrict__ x, float* __restrict__ out) {
    const int i = blockIdx.x;
    const int lo = row_ptr[i], hi = row_ptr[i + 1];
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float acc = 0.0f;
        for (int e = lo; e < hi; ++e)
            if (type[e]) acc += gine_act(act, x[(size_t)dst[e] * H + c] + ea_u[(size_t)umap[e] * H + c]);
        out[(size_t)i * H + c] = acc + (1.0f + eps) * x[(size_t)i * H + c];
    }
}
// adjoint w.r.t. x: node i sent act(x_i + ea_ij) to every local neighbour j
//   dx[i] = (1 + eps) dout[i] + sum_{e in row i, local} dout[dst e] * act'(x[i] + ea_u[umap e])
__global__ void gine_csr_bwd_x_kernel(int H, int act, float eps, const int32_t* __restrict__ row_ptr,
                                      const int32_t* __restrict__ dst, const int32_t* __restrict__ umap,
                                      const uint8_t* __restrict__ type, const float* __restrict__ ea_u,
                                      const float* __restrict__ x, const float* __restrict__ dout,
                                      float* __restrict__ dx) {
    const int i = blockIdx.x;
    const int lo = row_ptr[i], hi = row_ptr[i + 1];
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float xi = x[(size_t)i * H + c];
        float acc = 0.0f;
        for (int e = lo; e < hi; ++e)
            if (type[e])
                acc += dout[(size_t)dst[e] * H + c] * gine_dact(act, xi + ea_u[(size_t)umap[e] * H + c]);
        dx[(size_t)i * H + c] = acc + (1.0f + eps) * dout[(size_t)i * H + c];
    }
}
// adjoint w.r.t. the undirected edge attribute: pair u = {i, j} carried the messages i -> j and j -> i
__global__ void gine_csr_bwd_ea_kernel(int Eu, int H, int act, const int32_t* __restrict__ count,
                                       const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                       const uint8_t* __restrict__ type, const float* __restrict__ ea_u,
                                       const float* __restrict__ x, const float* __restrict__ dout,
                                       float* __restrict__ dea_u) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Eu * H) return;
    const int u = (int)(t / H), c = (int)(t % H);
    float g = 0.0f;
    if (u < *count && type[u]) {
        const int i = src[u], j = dst[u];
        const float a = ea_u[t];
        g = dout[(size_t)j * H + c] * gine_dact(act, x[(size_t)i * H + c] + a) +
            dout[(size_t)i * H + c] * gine_dact(act, x[(size_t)j * H + c] + a);
    }
    dea_u[t] = g;
}

// torch.nn.Embedding(max_norm) look-up side effect (schnet.py:151 `Embedding(100, H, max_norm=10.0)`):
// every row that is looked up and whose 2-norm exceeds max_norm is rescaled IN PLACE by
// max_norm / (norm + 1e-7) before the gather (torch embedding_renorm_).  Idempotent per row.
__global__ void emb_mark_kernel(int n, const int64_t* __restrict__ idx, int32_t* __restrict__ used) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) used[idx[i]] = 1;
}
__global__ __launch_bounds__(64) void emb_renorm_kernel(int H, float max_norm, const int32_t* __restrict__ used,
                                                        float* __restrict__ table) {
    const int r = blockIdx.x;
    if (!used[r]) return;
    float s = 0.0f;
    for (int c = threadIdx.x; c < H; c += 64) {
        const float v = table[(size_t)r * H + c];
        s = fmaf(v, v, s);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float norm = sqrtf(s);
    if (norm > max_norm) {
        const float scale = max_norm / (norm + 1e-7f);
        for (int c = threadIdx.x; c < H; c += 64) table[(size_t)r * H + c] *= scale;
    }
}

// eps_pos of the dual-encoder sampler (dualenc.py:826-849):
//   out = clip_norm(eq_local, clip_local) + clip_norm(eq_global, clip_global) * w_global
// clip < 0: no clipping; eq_global NULL: the global term is dropped (sigma >= global_start_sigma)
__global__ void dual_score_kernel(int N, const float* __restrict__ eq_local, const float* __restrict__ eq_global,
                                  float clip_local, float clip_global, float w_global, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float l[3], g[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) l[k] = eq_local[3 * i + k];
    if (clip_local >= 0.0f) {
        const float n = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(l[0], l[0]), __fmul_rn(l[1], l[1])), __fmul_rn(l[2], l[2])));
        const float d = n > clip_local ? clip_local / n : 1.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) l[k] = __fmul_rn(l[k], d);
    }
    if (eq_global) {
#pragma unroll
        for (int k = 0; k < 3; ++k) g[k] = eq_global[3 * i + k];
        if (clip_global >= 0.0f) {
            const float n =
                sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(g[0], g[0]), __fmul_rn(g[1], g[1])), __fmul_rn(g[2], g[2])));
            const float d = n > clip_global ? clip_global / n : 1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] = __fmul_rn(g[k], d);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) l[k] = __fadd_rn(l[k], __fmul_rn(g[k], w_global));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) out[3 * i + k] = l[k];
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

// ---------------------------------------------------------------------------------------------
// dense layer, host side.  Wp / Wp_t: pre-packed weights (pack_batch) or NULL (packed here into scratch).
// R: residual added to the product (may alias Y).  flags: 1 = dX += , 2 = dW / db += .
int linear_fwd_impl(int rows, int in, int out, const float* X, const float* W, const float* Wp, const LinEpi& epi,
                    float* Y, float* scratch, size_t scratch_floats, hipStream_t st) {
    if (rows == 0) return TSD_OK;
    if (mfma_shape(in, out) && (Wp || (scratch && scratch_floats >= (size_t)in * out))) {
        if (!Wp) {
            const int n = in * out;
            hipLaunchKernelGGL(pack_any_kernel, dim3((n + 255) / 256), dim3(256), 0, st, W, scratch, out, in, 0);
            Wp = scratch;
        }
        return dispatch_linear_mfma(rows, in, out, X, Wp, epi, Y, st);
    }
    const int64_t n = (int64_t)rows * out;
    hipLaunchKernelGGL(linear_naive_kernel, dim3(blocks_for(n)), dim3(256), 0, st, rows, in, out, X, W, 0, epi, Y);
    TSD_LAUNCH_CHECK("linear_naive");
    return TSD_OK;
}

int linear_bwd_impl(int rows, int in, int out, const float* X, const float* W, const float* Wp_t, const float* dY,
                    float* dX, float* dW, float* db, int flags, LinEpi epi, float* scratch, size_t scratch_floats,
                    hipStream_t st) {
    // scratch layout: [0, 64*out) bias partials | [.., + in*out) packed W for dgrad | wgrad partials
    const size_t off_pack = 64 * (size_t)out, off_part = off_pack + (size_t)in * out;
    const int accW = (flags & 2) ? 1 : 0;
    if (dX && rows > 0) {  // dX = dY W
        epi.bias = nullptr;
        if (!(flags & 4)) epi.R = (flags & 1) ? dX : nullptr;
        if (mfma_shape(out, in) && (Wp_t || (scratch && scratch_floats >= off_part))) {
            if (!Wp_t) {
                const int n = in * out;
                hipLaunchKernelGGL(pack_any_kernel, dim3((n + 255) / 256), dim3(256), 0, st, W, scratch + off_pack, in,
                                   out, 1);
                Wp_t = scratch + off_pack;
            }
            int r = dispatch_linear_mfma(rows, out, in, dY, Wp_t, epi, dX, st);
            if (r) return r;
        } else {
            const int64_t n = (int64_t)rows * in;
            hipLaunchKernelGGL(linear_naive_kernel, dim3(blocks_for(n)), dim3(256), 0, st, rows, out, in, dY, W, 1, epi,
                               dX);
            TSD_LAUNCH_CHECK("dgrad_naive");
        }
    }
    bool db_done = false;
    if (dW) {  // dW = dY^T X
        const int S = rows >= 4096 ? 64 : (rows >= 1024 ? 32 : (rows >= 256 ? 16 : 4));  // row splits of the MFMA wgrad
        if (rows == 0) {
            if (!accW) TSD_HIP(hipMemsetAsync(dW, 0, (size_t)out * in * sizeof(float), st));
        } else if (scratch && scratch_floats >= off_part + (size_t)S * out * in && out % 128 == 0 && in % 128 == 0) {
            const int per = ((rows + S - 1) / S + WG_T - 1) / WG_T * WG_T;
            float* part = scratch + off_part;
            float* bpart = db ? scratch : nullptr;  // [S][out], S <= 64
            hipLaunchKernelGGL(wgrad_kernel, dim3(in / 128, out / 128, S), dim3(WG_NT), 0, st, rows, in, out, per, dY, X,
                               part, bpart);
            const int64_t n = (int64_t)out * in;
            const int nb = db ? out : 0;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + nb + 63) / 64)), dim3(reduce_threads(S)), 0, st, n, nb, S,
                               part, bpart, dW, db, accW);
            TSD_LAUNCH_CHECK("wgrad");
            db_done = db != nullptr;
        } else if ((in <= WS_MAX_IN || out == 1) && scratch && scratch_floats >= off_part + 64 * (size_t)out * in) {
            // narrow layers; out == 1 is the in == 1 problem with dY and X swapped (same dW memory layout).
            // Every thread walks its row chunk serially (a latency chain): many rows get 256 chunks instead of 64.
            const bool swap = in > WS_MAX_IN;
            const int o2 = swap ? in : out, i2 = swap ? 1 : in;
            const int chunks = (rows >= 2048 && scratch_floats >= off_part + 256 * (size_t)out * in) ? 256 : 64;
            float* part = scratch + off_part;
            // the bias gradient in the same two launches (swapped form: the layer's dY is the kernel's X, one column)
            float* bpart = nullptr;
            const int nb = swap ? 1 : out;
            if (db && scratch_floats >= off_part + (size_t)chunks * ((size_t)out * in + nb)) bpart = part + (size_t)chunks * out * in;
            if (i2 == 1)
                hipLaunchKernelGGL(wgrad_small_kernel<1>, dim3((o2 + 63) / 64, chunks), dim3(256), 0, st, rows, i2, o2,
                                   swap ? X : dY, swap ? dY : X, part, bpart, swap ? 1 : 0);
            else
                hipLaunchKernelGGL(wgrad_small_kernel<WS_MAX_IN>, dim3((o2 + 63) / 64, chunks), dim3(256), 0, st, rows,
                                   i2, o2, swap ? X : dY, swap ? dY : X, part, bpart, 0);
            const int64_t n = (int64_t)o2 * i2;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + (bpart ? nb : 0) + 63) / 64)),
                               dim3(reduce_threads(chunks)), 0, st, n, bpart ? nb : 0, chunks, part, bpart, dW,
                               bpart ? db : (float*)nullptr, accW);
            TSD_LAUNCH_CHECK("wgrad_small");
            db_done = bpart != nullptr;
        } else {
            hipLaunchKernelGGL(wgrad_naive_kernel, dim3(out * in), dim3(256), 0, st, rows, in, out, dY, X, dW, accW);
            TSD_LAUNCH_CHECK("wgrad_naive");
        }
    }
    if (db && !db_done) {
        TSD_REQUIRE(scratch != nullptr && scratch_floats >= off_pack, "tsd_linear_bwd: db needs 64*out scratch floats");
        hipLaunchKernelGGL(colsum_stage1_kernel, dim3((out + 63) / 64, CS_CHUNKS), dim3(256), 0, st, rows, out, dY,
                           scratch);
        hipLaunchKernelGGL(colsum_stage2_kernel, dim3((out + 255) / 256), dim3(256), 0, st, out, scratch, db, accW);
        TSD_LAUNCH_CHECK("colsum");
    }
    return TSD_OK;
}

// dst[i] (+)= sum over the S partials part[s][i] in split order (S % 4 == 0): the reduction stage of every row-split
// kernel of the training step (workgroup = 64 outputs x 4 split quarters)
int launch_split_reduce(int64_t n, int S, const float* part, float* dst, int accumulate, hipStream_t st) {
    if (n == 0) return TSD_OK;
    TSD_REQUIRE(S > 0 && S % 4 == 0, "split reduce: S=%d", S);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(reduce_threads(S)), 0, st, n, 0, S, part,
                       (const float*)nullptr, dst, (float*)nullptr, accumulate);
    TSD_LAUNCH_CHECK("split_reduce");
    return TSD_OK;
}

// dW_k (+)= dY_k^T X_k (and db_k (+)= column sums of dY_k where db_k != NULL) for n equally shaped problems in two
// launches; `part` holds wgrad_batch_scratch_floats(n, rows, in, out) floats.
// Row splits S of a launch of m problems with `blocks` 128 x 128 output blocks each: the m * blocks * S workgroups
// (8 waves, 71 VGPRs: up to three per CU) are all resident at once and a CU's time is its workgroup count, so S is
// taken to fill a whole number k >= 2 of workgroups on each of the 256 CUs from below (14 filter problems: S = 9 ->
// 504 workgroups, two per CU on 252 CUs, where S = 8 left a quarter of the CUs with one and 16 with 3.5 on average;
// 21 node-level problems: S = 6 -> 504, where 8 gave 672 = 2.6 per CU, i.e. three).
int wgrad_batch_splits(int m, int blocks, int rows) {
    const int unit = std::max(1, m * blocks);
    const int cap = std::max(1, std::min(128, rows / 64));
    for (int k = 2; k <= 16; ++k) {
        const int S = 256 * k / unit;
        if (S >= 4) return std::min(S, cap);
    }
    return std::min(4, cap);
}
size_t wgrad_batch_scratch_floats(int n, int rows, int in, int out) {
    const int blocks = (in / 128) * (out / 128);
    size_t worst = 0;
    for (int m = 1; m <= std::min(n, WG_BATCH_MAX); ++m)  // (a launch holds at most WG_BATCH_MAX problems)
        worst = std::max(worst, (size_t)m * wgrad_batch_splits(m, blocks, rows));
    return worst * ((size_t)out * in + out);
}
int launch_wgrad_batch(int n, int rows, int in, int out, const float* const* dY, const float* const* X, float* const* dW,
                       float* const* db, int accumulate, float* part, hipStream_t st, const float* const* amax_h2) {
    if (n == 0) return TSD_OK;
    TSD_REQUIRE(out % 128 == 0 && in % 128 == 0 && rows > 0, "wgrad batch: shape %d x %d, %d rows", out, in, rows);
    const int64_t nW = (int64_t)out * in;
    for (int base = 0; base < n; base += WG_BATCH_MAX) {
        const int m = n - base < WG_BATCH_MAX ? n - base : WG_BATCH_MAX;
        const int S = wgrad_batch_splits(m, (in / 128) * (out / 128), rows);
        const int per = ((rows + S - 1) / S + WG_T - 1) / WG_T * WG_T;
        WgradBatch b;
        WgradOutBatch o;
        for (int k = 0; k < m; ++k) {
            b.it[k] = WgradItem{dY[base + k], X[base + k]};
            o.it[k] = WgradOut{dW[base + k], db ? db[base + k] : nullptr};
        }
        float* bpart = part + (size_t)m * S * nW;
        if (amax_h2 != nullptr) {  // split-f16 form (amax_h2[k]: the dY tensor's running max, or NULL entries: unscaled)
            static_assert(WG_T == WH_KT, "the split sizes are shared with the fp32 form");
            WgradBatchH bh;
            for (int k = 0; k < m; ++k) bh.it[k] = WgradItemH{dY[base + k], X[base + k], amax_h2[base + k]};
            const int Z = m * S, q = (in / 128) * (out / 128);
            hipLaunchKernelGGL(wgrad_h2_batch_kernel, dim3((unsigned)((Z + 7) / 8 * 8 * q)), dim3(WH_NT), 0, st, rows, in, out,
                               per, S, Z, bh, part, bpart);
        } else
        hipLaunchKernelGGL(wgrad_batch_kernel, dim3(in / 128, out / 128, m * S), dim3(WG_NT), 0, st, rows, in, out, per,
                           S, b, part, bpart);
        // (the partials are 16-byte aligned; a dW that is not -- checked per problem in the kernel -- is written by scalar stores)
#ifndef TSD_REDUCE4
#define TSD_REDUCE4 1  // 0 (A/B builds): the one-output-per-lane reduce of rounds 1-5
#endif
        const bool quad = TSD_REDUCE4 != 0 && nW % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
        if (quad)
            hipLaunchKernelGGL(wgrad_reduce_batch4_kernel, dim3((unsigned)((nW / 4 + out + 63) / 64), m), dim3(reduce_threads(S)),
                               0, st, nW, out, S, part, bpart, o, accumulate);
        else
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)((nW + out + 63) / 64), m), dim3(reduce_threads(S)),
                           0, st, nW, out, S, part, bpart, o, accumulate);
    }
    TSD_LAUNCH_CHECK("wgrad_batch");
    return TSD_OK;
}

size_t linear_scratch_floats(int in, int out) { return 64 * (size_t)out + (size_t)in * out + 64 * (size_t)out * in; }

}  // namespace tsd

using namespace tsd;

extern "C" {

int tsd_linear_fwd(int32_t rows, int32_t in, int32_t out, const float* X, const float* W, const float* b, float* Y,
                   float* scratch, size_t scratch_floats, void* stream) {
    LinEpi epi;
    epi.bias = b;
    return linear_fwd_impl(rows, in, out, X, W, nullptr, epi, Y, scratch, scratch_floats, (hipStream_t)stream);
}

int tsd_linear_packable(int32_t in, int32_t out) { return mfma_shape(in, out) ? 1 : 0; }

int tsd_pack_linear_batch(int32_t n, const float* const* W, float* const* dst, const int32_t* out_dim,
                          const int32_t* in_dim, const int32_t* transposed, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += TSD_PACK_MAX) {
        PackBatch b;
        const int m = n - base < TSD_PACK_MAX ? n - base : TSD_PACK_MAX;
        int biggest = 0;
        for (int k = 0; k < m; ++k) {
            const int o = out_dim[base + k], i = in_dim[base + k], t = transposed[base + k];
            TSD_REQUIRE(mfma_shape(i, o), "tsd_pack_linear_batch: layer %d (%d -> %d) has no MFMA instance", base + k, i, o);
            b.W[k] = W[base + k];
            b.dst[k] = dst[base + k];
            b.ncols[k] = t ? i : o;  // forward: B[k = in][n = out]; dgrad: B[k = out][n = in]
            b.kdim[k] = t ? o : i;
            b.transposed[k] = t;
            biggest = o * i > biggest ? o * i : biggest;
        }
        const int bx = (biggest + 255) / 256 < 64 ? (biggest + 255) / 256 : 64;
        hipLaunchKernelGGL(pack_batch_kernel, dim3(bx, m), dim3(256), 0, st, b);
    }
    TSD_LAUNCH_CHECK("pack_batch");
    return TSD_OK;
}

int tsd_linear_fwd_packed(int32_t rows, int32_t in, int32_t out, const float* X, const float* Wp, const float* b,
                          float* Y, void* stream) {
    if (rows == 0) return TSD_OK;
    TSD_REQUIRE(mfma_shape(in, out), "tsd_linear_fwd_packed: no MFMA instance for %d -> %d", in, out);
    LinEpi epi;
    epi.bias = b;
    return dispatch_linear_mfma(rows, in, out, X, Wp, epi, Y, (hipStream_t)stream);
}

int tsd_linear_bwd(int32_t rows, int32_t in, int32_t out, const float* X, const float* W, const float* Wp_t,
                   const float* dY, float* dX, float* dW, float* db, float* scratch, size_t scratch_floats,
                   void* stream) {
    return linear_bwd_impl(rows, in, out, X, W, Wp_t, dY, dX, dW, db, 0, LinEpi(), scratch, scratch_floats,
                           (hipStream_t)stream);
}

int tsd_act_fwd(int32_t kind, int64_t n, const float* x, float* y, void* stream) {
    if (n == 0) return TSD_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, kind, n, x, y);
    TSD_LAUNCH_CHECK("act_fwd");
    return TSD_OK;
}
int tsd_act_bwd(int32_t kind, int64_t n, const float* x, const float* dy, float* dx, void* stream) {
    if (n == 0) return TSD_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, kind, n, x, dy, dx);
    TSD_LAUNCH_CHECK("act_bwd");
    return TSD_OK;
}

int tsd_gine_aggregate(int32_t num_nodes, int64_t num_edges, int32_t H, int32_t activation, float eps,
                       const float* x, const int64_t* edge_index, const float* edge_attr, float* out,
                       void* stream) {
    TSD_REQUIRE(activation >= 0 && activation <= 2, "activation %d (0 none, 1 relu, 2 softplus)", activation);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)num_nodes * H;
    if (n == 0) return TSD_OK;
    hipLaunchKernelGGL(gine_self_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, 1.0f + eps, x, out);
    if (num_edges > 0)
        hipLaunchKernelGGL(gine_message_kernel, dim3(blocks_for(num_edges * H)), dim3(256), 0, st, num_edges, H,
                           activation, x, edge_index, edge_attr, out);
    TSD_LAUNCH_CHECK("gine_aggregate");
    return TSD_OK;
}

int tsd_gine_csr_fwd(int32_t num_nodes, int32_t H, int32_t activation, float eps, tsd_edges enc,
                     const float* ea_u, const float* x, float* out, void* stream) {
    TSD_REQUIRE(activation >= 0 && activation <= 2, "activation %d (0 none, 1 relu, 2 softplus)", activation);
    if (num_nodes == 0) return TSD_OK;
    hipLaunchKernelGGL(gine_csr_fwd_kernel, dim3(num_nodes), dim3(H < 256 ? (H + 63) / 64 * 64 : 256), 0,
                       (hipStream_t)stream, H, activation, eps, enc.row_ptr, enc.dst, enc.umap, enc.type_r, ea_u, x, out);
    TSD_LAUNCH_CHECK("gine_csr_fwd");
    return TSD_OK;
}
int tsd_gine_csr_bwd(int32_t num_nodes, int32_t capacity_u, int32_t H, int32_t activation, float eps, tsd_edges enc,
                     tsd_edges enc_u, const float* ea_u, const float* x, const float* dout, float* dx, float* dea_u,
                     void* stream) {
    TSD_REQUIRE(activation >= 0 && activation <= 2, "activation %d (0 none, 1 relu, 2 softplus)", activation);
    hipStream_t st = (hipStream_t)stream;
    if (dx && num_nodes > 0)
        hipLaunchKernelGGL(gine_csr_bwd_x_kernel, dim3(num_nodes), dim3(H < 256 ? (H + 63) / 64 * 64 : 256), 0, st, H,
                           activation, eps, enc.row_ptr, enc.dst, enc.umap, enc.type_r, ea_u, x, dout, dx);
    if (dea_u && capacity_u > 0)
        hipLaunchKernelGGL(gine_csr_bwd_ea_kernel, dim3(blocks_for((int64_t)capacity_u * H)), dim3(256), 0, st,
                           capacity_u, H, activation, enc_u.count, enc_u.src, enc_u.dst, enc_u.type_r, ea_u, x, dout,
                           dea_u);
    TSD_LAUNCH_CHECK("gine_csr_bwd");
    return TSD_OK;
}
int tsd_embedding_renorm(int32_t num_rows, int32_t H, int32_t n, const int64_t* idx, float max_norm, float* table,
                         int32_t* scratch, void* stream) {
    if (n == 0 || num_rows == 0) return TSD_OK;
    hipStream_t st = (hipStream_t)stream;
    TSD_HIP(hipMemsetAsync(scratch, 0, (size_t)num_rows * sizeof(int32_t), st));
    hipLaunchKernelGGL(emb_mark_kernel, dim3(blocks_for(n)), dim3(256), 0, st, n, idx, scratch);
    hipLaunchKernelGGL(emb_renorm_kernel, dim3(num_rows), dim3(64), 0, st, H, max_norm, scratch, table);
    TSD_LAUNCH_CHECK("embedding_renorm");
    return TSD_OK;
}
int tsd_dual_score(int32_t num_nodes, const float* eq_local, const float* eq_global, float clip_local,
                   float clip_global, float w_global, float* out, void* stream) {
    if (num_nodes == 0) return TSD_OK;
    hipLaunchKernelGGL(dual_score_kernel, dim3(blocks_for(num_nodes)), dim3(256), 0, (hipStream_t)stream, num_nodes,
                       eq_local, eq_global, clip_local, clip_global, w_global, out);
    TSD_LAUNCH_CHECK("dual_score");
    return TSD_OK;
}

int tsd_gaussian_edge_encode(int64_t num_edges, int32_t K, float coeff, const float* d, const float* offset,
                             const int64_t* type, const float* bond_emb, float* out, void* stream) {
    if (num_edges == 0) return TSD_OK;
    hipLaunchKernelGGL(gaussian_edge_kernel, dim3(blocks_for(num_edges * 2 * K)), dim3(256), 0, (hipStream_t)stream,
                       num_edges, K, coeff, d, offset, type, bond_emb, out);
    TSD_LAUNCH_CHECK("gaussian_edge_encode");
    return TSD_OK;
}

int tsd_emb_mul_fwd(int32_t rows, int32_t H, const float* x, const float* emb, const uint8_t* idx, float* y,
                    void* stream) {
    if (rows == 0) return TSD_OK;
    hipLaunchKernelGGL(emb_mul_fwd_kernel, dim3(blocks_for((int64_t)rows * H)), dim3(256), 0, (hipStream_t)stream,
                       rows, H, x, emb, idx, y);
    TSD_LAUNCH_CHECK("emb_mul_fwd");
    return TSD_OK;
}
int tsd_emb_mul_bwd(int32_t rows, int32_t H, const float* x, const float* emb, const uint8_t* idx, const float* dy,
                    float* dx, float* demb, void* stream) {
    if (rows == 0) return TSD_OK;
    const int chunks = (rows + 255) / 256 < 512 ? (rows + 255) / 256 : 512;
    const int per = (rows + chunks - 1) / chunks;
    hipLaunchKernelGGL(emb_mul_bwd_kernel, dim3((H + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, rows, H, per,
                       x, emb, idx, dy, dx, demb);
    TSD_LAUNCH_CHECK("emb_mul_bwd");
    return TSD_OK;
}

int tsd_gather_rows(int32_t rows, int32_t H, const float* table, const int64_t* idx, float* y, void* stream) {
    if (rows == 0) return TSD_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for((int64_t)rows * H)), dim3(256), 0, (hipStream_t)stream,
                       rows, H, table, idx, y);
    TSD_LAUNCH_CHECK("gather_rows");
    return TSD_OK;
}
int tsd_scatter_rows_add(int32_t rows, int32_t H, const float* dy, const int64_t* idx, float* dtable, void* stream) {
    if (rows == 0) return TSD_OK;
    hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(blocks_for((int64_t)rows * H)), dim3(256), 0,
                       (hipStream_t)stream, rows, H, dy, idx, dtable);
    TSD_LAUNCH_CHECK("scatter_rows_add");
    return TSD_OK;
}

int tsd_row_mask(int32_t rows, int32_t H, const float* dist, float cutoff, int32_t smooth, float* x, void* stream) {
    if (rows == 0) return TSD_OK;
    hipLaunchKernelGGL(row_mask_kernel, dim3(blocks_for((int64_t)rows * H)), dim3(256), 0, (hipStream_t)stream, rows,
                       H, dist, cutoff, smooth, x);
    TSD_LAUNCH_CHECK("row_mask");
    return TSD_OK;
}

int tsd_aggregate_bwd_filter(int32_t H, int32_t capacity_u, tsd_edges enc_u, const float* dagg, const float* x1,
                             float* dWf, void* stream) {
    return launch_aggregate_bwd_filter(H, capacity_u, enc_u, dagg, x1, dWf, 0, 0.0f, 0, (hipStream_t)stream);
}

int tsd_pair_product_fwd(int32_t H, int32_t capacity_u, tsd_edges out_u, const float* h, float* p, void* stream) {
    if (capacity_u == 0) return TSD_OK;
    hipLaunchKernelGGL(pair_product_fwd_kernel, dim3(blocks_for((int64_t)capacity_u * H)), dim3(256), 0,
                       (hipStream_t)stream, H, out_u, h, p);
    TSD_LAUNCH_CHECK("pair_product_fwd");
    return TSD_OK;
}
int tsd_pair_product_bwd(int32_t num_nodes, int32_t H, tsd_edges out, const float* dp, const float* h, float* dh,
                         void* stream) {
    if (num_nodes == 0) return TSD_OK;
    hipLaunchKernelGGL(pair_product_bwd_kernel, dim3(blocks_for((int64_t)num_nodes * H)), dim3(256), 0,
                       (hipStream_t)stream, num_nodes, H, out, dp, h, dh);
    TSD_LAUNCH_CHECK("pair_product_bwd");
    return TSD_OK;
}

int tsd_eq_und_fwd(int32_t num_nodes, tsd_edges out, const float* pos, const float* s_u, float* node_eq,
                   void* stream) {
    if (num_nodes == 0) return TSD_OK;
    hipLaunchKernelGGL(eq_und_fwd_kernel, dim3((num_nodes + 127) / 128), dim3(128), 0, (hipStream_t)stream, num_nodes,
                       out, pos, s_u, node_eq);
    TSD_LAUNCH_CHECK("eq_und_fwd");
    return TSD_OK;
}
int tsd_eq_und_bwd(int32_t capacity_u, tsd_edges out_u, const float* pos, const float* g, float* ds_u, void* stream) {
    if (capacity_u == 0) return TSD_OK;
    hipLaunchKernelGGL(eq_und_bwd_kernel, dim3((capacity_u + 255) / 256), dim3(256), 0, (hipStream_t)stream, out_u,
                       pos, g, ds_u);
    TSD_LAUNCH_CHECK("eq_und_bwd");
    return TSD_OK;
}
int tsd_pair_distance(int32_t capacity_u, tsd_edges list_u, const float* pos, float* d, void* stream) {
    if (capacity_u == 0) return TSD_OK;
    hipLaunchKernelGGL(pair_distance_kernel, dim3((capacity_u + 255) / 256), dim3(256), 0, (hipStream_t)stream, list_u,
                       pos, d);
    TSD_LAUNCH_CHECK("pair_distance");
    return TSD_OK;
}

}  // extern "C"

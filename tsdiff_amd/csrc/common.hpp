// common.hpp -- shared device/host helpers of libtsdiff_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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

// One-time set-up per DEVICE (a kernel's dynamic-LDS limit is a per-device function attribute, and one process
// may drive several GPUs): first_on_current_device() is true exactly once per device id for each instance.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool first_on_current_device() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return true;
        const uint64_t bit = 1ull << (d & 63);
        return !(mask.fetch_or(bit) & bit);
    }
};
template <typename K>
static inline int allow_lds(K kernel, size_t bytes, DeviceOnce& once) {
    if (bytes > 48 * 1024 && once.first_on_current_device())
        TSD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return TSD_OK;
}

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

// directed edge e -> index of its undirected pair in the matching *_u list (needs every row's fill to be
// complete).  Stand-alone: edge_umap_kernel (kernels_graph.hip); in the fused forward it rides as an extra role
// of the edge-embedding launch, which needs the undirected lists only (UmapRole, kernels_mlp.hip).
__device__ __forceinline__ void edge_umap_body(const tsd_geometry& g, const int32_t* __restrict__ graph_ptr,
                                               const int32_t* __restrict__ node_graph,
                                               const int32_t* __restrict__ pair_ptr, int P, int e) {
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
struct UmapRole {
    int blocks;  // 0: no umap role
    tsd_geometry g;
    const int32_t *graph_ptr, *node_graph, *pair_ptr;
    int P;
};

// optional third role of the per-block launch (kernels_combo.hip), filled by the forward in api.hip
struct ComboPre {
    int tiles;  // 0: no pre role
    tsd_edges e;
    const float* edge_attr;
    const int32_t* attr_row;  // edge_attr row of out edge e (NULL: e)
    const float *w0b, *b0;    // packed W0 rows k in [H, 2H), bias
    float* out;               // [capacity_u, H]
};

inline bool hidden_supported(int H) { return H == 64 || H == 128 || H == 256; }

// ---------------------------------------------------------------------------------------------
// device math (fp32, no fast-math)
// ---------------------------------------------------------------------------------------------
// Activations use the hardware transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each).
// Absolute error of either function is ~1e-7 for every input, the same class as the fp32 rounding of
// the reference's own softplus(x) - log 2 cancellation; the library versions cost ~100 VALU ops per
// element, which was 30 % of the fused CFConv kernel (profiles/r01_a_kernel_stats.md).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float swishf(float x) {  // reference utils/activation_functions.py:10-11
    return x * __builtin_amdgcn_rcpf(1.0f + fast_exp(-x));  // x * sigmoid(x)
}
__device__ __forceinline__ float sspf(float x) {  // reference models/encoder/schnet.py:65-71
    // F.softplus(beta=1, threshold=20) - log(2) = max(x,0) + log1p(exp(-|x|)) - log 2
    const float t = fast_exp(-fabsf(x));
    const float l = __builtin_amdgcn_logf(1.0f + t) * 0.69314718055994530942f;
    return (fmaxf(x, 0.0f) + l) - 0.69314718055994530942f;
}

// Philox4x32-10 (Salmon et al., SC'11): counter-based generator for the Gaussian draws of the sampling loop
// (reference models/sampler.py:213 torch.randn_like): key = seed, counter = (ctr, 0), four u32 per call.
struct u32x4 { uint32_t x, y, z, w; };
__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint64_t ctr, uint64_t seed) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}
// three standard normals of counter `ctr` (Box-Muller on the four uniforms; the fourth normal is dropped)
__device__ __forceinline__ void philox_normal3(uint64_t ctr, uint64_t seed, float (&z)[3]) {
    const u32x4 r = philox4x32_10(ctr, seed);
    const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1), 24 bits
    const float u1 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(r.w >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float sa, ca, sb, cb;
    sincospif(2.0f * u1, &sa, &ca);
    sincospif(2.0f * u3, &sb, &cb);
    (void)sb;
    z[0] = ra * ca;
    z[1] = ra * sa;
    z[2] = rb * cb;
}

// CFConv cutoff weight C(d), reference models/encoder/schnet.py:92-98
__device__ __forceinline__ float cutoff_weight(float d, float cutoff, int smooth) {
    if (!(d <= cutoff)) return 0.0f;
    if (!smooth) return 1.0f;
    return d >= 0.0f ? 0.5f * (cosf(d * 3.14159265358979323846f / cutoff) + 1.0f) : 0.0f;
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
// B is double-buffered in registers in chunks of PF k-blocks: while chunk c is multiplied (PF*4*RB*CB
// MFMAs = 2048 cycles at RB=1, CB=2), chunk c+1 is in flight from L2 -- one k-block of look-ahead
// (the first version) exposed the L2 latency every iteration (tools/mfma_probe.hip: 84 -> 93 TFLOP/s).
template <int RB, int CB, int K, int PF = 4, bool PIN = false>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ ldsA, int lda,
                                          const float* __restrict__ Bp, int nout, int col0,
                                          f32x16 (&acc)[RB][CB]) {
    const int lane = threadIdx.x & 63;
    const int hi = lane >> 5;
    const int l31 = lane & 31;
    const float* aptr = ldsA + l31 * lda + hi * 4;
    const f32x4* bptr = reinterpret_cast<const f32x4*>(Bp) + (size_t)hi * nout + col0 + l31;
    constexpr int KB = K / 8;
    constexpr int NC = KB / PF;
    static_assert(KB % PF == 0, "K must be a multiple of 8 * PF");
    f32x4 b0[PF][CB], b1[PF][CB];
    auto loadB = [&](f32x4 (&b)[PF][CB], int chunk) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) b[p][cb] = bptr[(size_t)(chunk * PF + p) * 2 * nout + cb * 32];
    };
    auto compute = [&](const f32x4 (&b)[PF][CB], int chunk) {
        f32x4 a[PF][RB];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                a[p][rb] = *reinterpret_cast<const f32x4*>(aptr + rb * 32 * lda + (chunk * PF + p) * 8);
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][rb][s], b[p][cb][s], acc[rb][cb], 0, 0, 0);
    };
    // The machine scheduler sinks every load to just before its first use (one k-block of look-ahead).
    // With >= 2 workgroups per CU the other waves cover that (measured: pinning the pipeline costs 10 % at
    // configs[1] / configs[4] sizes); a launch with fewer workgroups than CUs is a pure latency chain and
    // pins the chunked pipeline with scheduling barriers (PIN).
    loadB(b0, 0);
    for (int c = 0; c < NC; c += 2) {
        if (c + 1 < NC) loadB(b1, c + 1);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        compute(b0, c);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NC) loadB(b0, c + 2);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NC) compute(b1, c + 1);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
}

// 16-row variant on the 16x16x4 f32 MFMA (same rate, half the rows): A lane l holds A[i = l&15][k = l>>4],
// B lane l holds B[k = l>>4][j = l&15], C/D: col = l&15, row = (l>>4)*4 + r, r in [0,4).
// k is permuted in groups of 16: step s of k-block kb consumes k = kb*16 + q*4 + s on lane quarter q, so a
// lane reads one float4 of A and one float4 of B per column block per 4 MFMAs, from the SAME packed
// weight layout ([k/4][out][k%4]) as the 32-row variant.  CB = 16-wide column blocks of this wave.
// PF / PIN as in gemm_tile: the node chain of the per-block launch is a latency chain that shares its CU with
// filter workgroups streaming weights, so it prefetches deep (PF = K/32: the whole B slice of a GEMM in two
// chunks, both in flight from the start) and pins that order.
template <int CB, int K, int PF = 4, bool PIN = false>
__device__ __forceinline__ void gemm_tile16(const float* __restrict__ ldsA, int lda,
                                            const float* __restrict__ Bp, int nout, int col0,
                                            f32x4 (&acc)[CB]) {
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4;
    const int l15 = lane & 15;
    const float* aptr = ldsA + l15 * lda + q * 4;
    const f32x4* bptr = reinterpret_cast<const f32x4*>(Bp) + (size_t)q * nout + col0 + l15;
    constexpr int KB = K / 16;
    constexpr int NC = KB / PF;
    static_assert(KB % PF == 0, "K must be a multiple of 16 * PF");
    f32x4 b0[PF][CB], b1[PF][CB];
    auto loadB = [&](f32x4 (&b)[PF][CB], int chunk) {
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) b[p][cb] = bptr[(size_t)(chunk * PF + p) * 4 * nout + cb * 16];
    };
    auto compute = [&](const f32x4 (&b)[PF][CB], int chunk) {
        f32x4 a[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) a[p] = *reinterpret_cast<const f32x4*>(aptr + (chunk * PF + p) * 16);
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][s], b[p][cb][s], acc[cb], 0, 0, 0);
    };
    loadB(b0, 0);
    for (int c = 0; c < NC; c += 2) {
        if (c + 1 < NC) loadB(b1, c + 1);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        compute(b0, c);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < NC) loadB(b0, c + 2);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NC) compute(b1, c + 1);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
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

// split16.hpp -- the tile GEMM of the inference forward on the f16 MFMA pipes with fp32-class accuracy ("f16x2").
//
// gfx950 runs v_mfma_f32_32x32x16_f16 at 16x the rate of the fp32-input MFMA (MI355X_MICROARCH.md: 2.5 PFLOP/s dense
// against 157 TFLOP/s), and has no TF32-like middle form.  Every fp32 operand is therefore SPLIT into two f16 planes
//     a = a_hi + a_lo * 2^-11,   a_hi = f16(a),   a_lo = f16((a - a_hi) * 2^11)          (22 significant bits)
// and a product a . b is evaluated as THREE f16 MFMAs with fp32 accumulation
//     acc_m += a_hi b_hi ;   acc_x += a_hi b_lo + a_lo b_hi ;   result = acc_m + acc_x * 2^-11
// (the dropped a_lo b_lo term is 2^-22 relative).  The low planes are stored scaled by 2^11, so they never reach the
// f16 subnormal range before the value itself is below ~1e-8: the representation error is max(2^-23 |a|, ~1e-11)
// whatever the magnitude, the accumulation is the MFMA's fp32.  Measured (tools/mfma_probe4.hip, K = 256, softplus-like
// activations): max error against an fp64 GEMM 6.0e-7 of the result scale, an fp32 fma chain 5.0e-7 -- the same class.
// What the split cannot represent is |a| > 65504 (f16 range): every conversion tracks max|a|, a kernel that saw a
// larger value raises TSD_STATUS_RANGE and the host reruns the call on the fp32-MFMA kernels (tsdiff_amd/engine.py).
//
// Weights: tsd_pack_weights16 rewrites every dense matrix of the packed fp32 arena IN THE SAME BYTES (hi + lo planes =
// 4 bytes per element) as  [k/16][plane][k%16 / 8][out][k%8]  f16 -- a lane's B fragment of one k-step (16 k for the
// 32x32x16 MFMA: 8 consecutive k of one output column) is one aligned 16-byte load, a wave reads two 512-byte runs;
// the 16x16x32 MFMA of the 16-row node tiles reads the same image (its four lane quarters take (k-step, half) pairs).
// Biases, embeddings and narrow layers are copied verbatim, so a kernel takes ONE arena pointer in either precision.
// The A operand lives in LDS as two planes [rows][H + 8] f16 (row stride = 4 banks mod 64: ds_read_b128 conflict-free).
//
// The B feed is the counted asm ring of common.hpp (R k-steps in flight per wave).
#pragma once

namespace tsd {

using f16 = _Float16;
using f16x8 = f16 __attribute__((ext_vector_type(8)));
using f16x4 = f16 __attribute__((ext_vector_type(4)));
using f16x2 = f16 __attribute__((ext_vector_type(2)));

constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;
constexpr float F16_MAX = 65504.0f;

constexpr int ldh_of(int H) { return H + 8; }                                  // f16 elements per LDS row of one plane
constexpr size_t planes_bytes(int rows, int H) { return (size_t)rows * ldh_of(H) * 2 * 2; }  // both planes

struct Planes {
    f16* hi;
    f16* lo;
};
__device__ __forceinline__ Planes planes_at(void* smem, int rows, int ldh) {
    f16* p = reinterpret_cast<f16*>(smem);
    return Planes{p, p + rows * ldh};
}

// amax: running max |a| of everything this thread converted (range check).  The max is an asm statement at the point of
// the conversion: left to the compiler it is batched into v_max3 chains long after the stores, which keeps every
// converted value alive across them (+10-15 VGPRs in the tile kernels).
__device__ __forceinline__ void amax_upd(float& amax, float a) {
    asm("v_max_f32 %0, %0, |%1|" : "+v"(amax) : "v"(a));
}
__device__ __forceinline__ void amax_upd2(float& amax, float a, float b) {
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(a), "v"(b));
}
// `a` is made opaque first: with -ffp-contract=on the compiler may fuse a conversion with the multiply that produced its
// input (v_fma_mixlo_f16: f16 of the EXACT product, one rounding) and, worse, does so for ONE of two uses of the same
// conversion -- found in round 5 in the one-launch kernel's pair role: the high plane was stored as f16(fp32(x * s)) (packed
// conversion) while the low plane was computed against f16(x * s) (mixed-precision fma); where the two roundings differ
// (1 element in ~10^3) hi + lo 2^-11 misses the value by an f16 ulp -- 15 of 1791 pair outputs of a test batch were 6e-5 off.
// Behind the empty asm statement both planes derive from the same fp32 register by plain round-to-nearest conversions.
__device__ __forceinline__ void split1(float a, f16& h, f16& l) {
#ifndef TSD_SPLIT_NO_OPAQUE  // (timing experiments only: without the statement the planes can be inconsistent)
    asm("" : "+v"(a));
#endif
    h = (f16)a;
    l = (f16)((a - (float)h) * SPLIT_SCALE);
}
__device__ __forceinline__ void planes_store1(const Planes& p, int off, float a, float& amax) {
    f16 h, l;
    amax_upd(amax, a);
    split1(a, h, l);
    p.hi[off] = h;
    p.lo[off] = l;
}
// two column-adjacent elements (off even): one 4-byte store per plane
__device__ __forceinline__ void planes_store2(const Planes& p, int off, float a0, float a1, float& amax) {
    f16x2 h, l;
    f16 hh, ll;
    amax_upd2(amax, a0, a1);
    split1(a0, hh, ll);
    h[0] = hh; l[0] = ll;
    split1(a1, hh, ll);
    h[1] = hh; l[1] = ll;
    *reinterpret_cast<f16x2*>(p.hi + off) = h;
    *reinterpret_cast<f16x2*>(p.lo + off) = l;
}
// four consecutive elements (off a multiple of 4): one 8-byte store per plane
__device__ __forceinline__ void planes_store4(const Planes& p, int off, const f32x4& a, float& amax) {
    f16x4 h, l;
    amax_upd2(amax, a[0], a[1]);
    amax_upd2(amax, a[2], a[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f16 hh, ll;
        split1(a[i], hh, ll);
        h[i] = hh;
        l[i] = ll;
    }
    *reinterpret_cast<f16x4*>(p.hi + off) = h;
    *reinterpret_cast<f16x4*>(p.lo + off) = l;
}
// A row of H values stored as planes in memory (common.hpp ATTRIBUTE ROWS AS f16 PLANES): chunk c (16 bytes, c < H / 4) of the row holds 8 f16 of
// the high plane (c < H / 8: k = 8 c ..) or of the low plane (k = 8 (c - H / 8) ..).  Copy one chunk into the LDS planes.
template <int H>
__device__ __forceinline__ void planes_put_chunk(const Planes& p, int row_off /* r * ldh */, int c, const f32x4& v) {
    // (one address: hi + row + chunk, the low plane `lo - hi` elements further)
    f16* dst = p.hi + row_off + (c & (H / 8 - 1)) * 8 + (c >= H / 8 ? (int)(p.lo - p.hi) : 0);
    *reinterpret_cast<f32x4*>(dst) = v;
}
// LOW side of the range.  An operand below 2^-14 has a subnormal high plane; its value is then carried by the scaled low
// plane with an ABSOLUTE precision of 2^-36 = 1.5e-11 instead of 22 relative bits, i.e. worse than fp32's 2^-24 relative
// below |a| = 2^-12 = 2.4e-4.  Single such elements do not matter (their absolute error drowns in the sum); a tensor whose
// values are ALL that small does -- measured: attribute rows at 1e-5 followed by 1e5-scale weights put edge_inv 1.4e-5
// from the fp64 evaluation (tests/test_gpu_round4.py).  So every conversion SITE (one tensor of one tile) keeps its own
// max |a| `m` of the elements this thread converts and closes with site_close: the site's max goes into the role's
// running max, and a site whose non-zero values all lie below 2^-12 turns that max into +inf, i.e. the role reports
// TSD_STATUS_RANGE and the host reruns the call on the fp32-MFMA kernels.  Sites are the conversions in which a thread
// holds SEVERAL CHANNELS of several rows (attribute rows, aggregates, h copies, h_i * h_j: 4 consecutive channels x 2-8
// rows per thread): what matters is a ROW of the A operand that is tiny throughout, and a thread that sees nothing but
// tiny values across channels and rows is the cheap witness of it.  The column-wise epilogue sites (one channel per
// thread: ssp / swish outputs) keep the plain running max -- a single dead channel is harmless there (its absolute error
// drowns in the row sums) and must not switch the arithmetic.  Exact zeros -- rows past the end of a list, atoms without
// neighbours -- are not counted.
constexpr float SPLIT_LOW = 2.44140625e-04f;  // 2^-12
__device__ __forceinline__ void site_close(float& amax, float m) {
    amax = fmaxf(amax, (m > 0.0f && m < SPLIT_LOW) ? __builtin_inff() : m);
}
// GRADIENT operands (training step).  Gradients sit wherever the loss scale puts them (1e-3 .. 1e-9 at batch 200), far
// below the band in which the split keeps 22 bits.  A dgrad is row-wise linear, so a gradient tile is converted as
// a * s with s = 2^-e (e = exponent of the tile's or row's max |a|: the scaled max lies in [1, 2)) and the product is
// multiplied by 1 / s = 2^e afterwards -- both exact.  No value can leave the f16 range (the flag is not needed), and an
// element 2^-13 below the tile's max still carries 22 bits.  m == 0 (an all-zero tile): s = 1.
__device__ __forceinline__ float pow2_scale(float m, float& inv) {
    unsigned b = __float_as_uint(m) & 0x7f800000u;
    b = b > 0x7e000000u ? 0x7e000000u : b;  // (|a| >= 2^125: not a gradient; keeps 254 - E a normal exponent)
    inv = b ? __uint_as_float(b) : 1.0f;
    return b ? __uint_as_float(0x7f000000u - b) : 1.0f;
}
// max over aligned groups of 32 lanes (ds_swizzle: no index registers, unlike the ds_bpermute behind __shfl_xor), and over
// the wave (uniform result)
__device__ __forceinline__ float max32(float m) {
#define TSD_SWZ(k) m = fmaxf(m, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(m), ((k) << 10) | 0x1f)))
    TSD_SWZ(16); TSD_SWZ(8); TSD_SWZ(4); TSD_SWZ(2); TSD_SWZ(1);
#undef TSD_SWZ
    return m;
}
__device__ __forceinline__ float max64(float m) {
    m = max32(m);
    return fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0)),
                 __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 32)));
}
// max of non-negative floats (their bit patterns order).  The slot is READ first and the atomic issued only by a caller
// that raises it: thousands of workgroups hitting one L2 line with read-modify-writes serialise there (measured: nine
// atomics per tile on two words made an 85 us kernel out of a 40 us one).
__device__ __forceinline__ void atomic_amax(float* slot, float m) {
    unsigned* p = reinterpret_cast<unsigned*>(slot);
    const unsigned b = __float_as_uint(m);
    if (b > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, b);
}

// the range flag of a workgroup role: any thread that converted a value beyond the f16 range (or a NaN's neighbour inf)
__device__ __forceinline__ void range_report(float amax, int32_t* status) {
    if (status != nullptr && !(amax <= F16_MAX)) atomicOr(status, TSD_STATUS_RANGE);
}

// 16-byte load that is served by L2, never by this CU's L1 (`sc1`): rows that another workgroup stored write-through
// during the same launch (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 loads pair with sc1 stores).  The
// destination counts as written at the statement: consume it behind ld16_wait4 (or any vmcnt(0) that names it).
__device__ __forceinline__ void ld16_sc1(f32x4& v, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld16_wait4(f32x4 (&v)[4]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
}

__device__ __forceinline__ f32x16 mfma_h32(const f32x4& a, const f32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_h16(const f32x4& a, const f32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// a wave-uniform pointer as the compiler can see it (the asm loads take their base through an "s" operand: a base the
// compiler keeps in VGPRs -- e.g. derived from a value it cannot prove uniform -- would be substituted as a VGPR pair)
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// B ring: R k-steps in flight, each CB column blocks x 2 planes of one float4 (8 f16) per lane
template <int CB, int R>
struct HRing {
    f32x4 b[R][CB][2];
    const char* base;  // wave-uniform: packed matrix (+ first k-step of this GEMM)
    unsigned voff;     // lane byte offset inside a k-step
    int step_bytes, plane_bytes, cb_bytes;
};
template <int CB>
__device__ __forceinline__ void hring_issue(f32x4 (&b)[CB][2], const char* __restrict__ sbase, unsigned voff,
                                            int plane_bytes, int cb_bytes) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const char* sb = sbase + (size_t)cb * cb_bytes + (size_t)p * plane_bytes;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[cb][p]) : "v"(voff), "s"(sb) : "memory");
        }
}
template <int N, int CB>
__device__ __forceinline__ void hring_wait(f32x4 (&b)[CB][2]) {
    static_assert(CB == 1 || CB == 2, "");
    if constexpr (CB == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0][0]), "+v"(b[0][1]) : "n"(N) : "memory");
    if constexpr (CB == 2)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]) : "n"(N) : "memory");
}

constexpr int HRING_R = 3;    // k-steps in flight, 32-row GEMMs (CB = 1: 8 fragments = 32 VGPRs)
#ifndef TSD_HRING16_R
#define TSD_HRING16_R 3
#endif
constexpr int HRING16_R = TSD_HRING16_R;  // 32-k steps in flight, 16-row GEMMs (CB = 2: 12 fragments = 48 VGPRs)

// ---- 32-row blocks: v_mfma_f32_32x32x16_f16.  A lane l: row l & 31, k = 8 (l >> 5) .. + 7 of the k-step; B lane l:
// column l & 31, the same k; C/D as the fp32 32x32 MFMA (acc_row).  KS = K / 16 k-steps.
// (the ring depth R is the HRing's: HRING_R by default, a kernel short of registers declares a shallower one)
template <int CB, int K, int R>
__device__ __forceinline__ void hgemm_ring_start(HRing<CB, R>& r, const float* __restrict__ Bp16, int nout, int col0) {
    const int lane = threadIdx.x & 63;
    r.base = uniform_ptr(Bp16);
    r.voff = (unsigned)(((lane >> 5) * nout + col0 + (lane & 31)) * 16);
    r.step_bytes = 64 * nout;   // 2 planes x 2 halves x nout x 16 B
#ifdef TSD_RING_WFAKE  // (timing experiment, WRONG results: every k-step re-reads the first one -- the weights come from L1.
    r.step_bytes = 0;      //  The bound on what ANY weight-stationary scheme could buy; profiles/r06_weight_stationary_bound.md)
#endif
    r.plane_bytes = 32 * nout;
    r.cb_bytes = 32 * 16;
    constexpr int KS = K / 16;
    static_for<0, (R < KS ? R : KS)>([&](auto i) {
        constexpr int I = decltype(i)::value;
        hring_issue<CB>(r.b[I], r.base + (size_t)I * r.step_bytes, r.voff, r.plane_bytes, r.cb_bytes);
    });
}
// PIN: the LDS address of every k-step passes through an asm statement, which orders that step's A reads behind the
// previous step's counted wait.  Left free, the compiler hoists the LDS reads of the last k-steps over the whole unrolled
// loop into registers of their own (+30 VGPRs in the tail of a GEMM: the backward filter chain then loses the second
// workgroup of its CU).
//
// TRANS (round 5): the SAME products with the operand roles swapped -- the weight fragment as the MFMA's A operand, the LDS
// tile fragment as B (both are 8 f16 per lane in the same layout, so the swap is free) -- which leaves the transposed
// result in the accumulators: lane l holds tile ROW l & 31 and the 16 output CHANNELS 8 (r >> 2) + 4 (l >> 5) + (r & 3),
// i.e. four runs of FOUR CONSECUTIVE channels of one row, where the plain form holds 16 rows of one channel.  Epilogues
// then write f16x4 / f32x4 (8 / 4 LDS stores per lane and 32-row block instead of 32 / 16) and per-row factors (cutoff
// weight) are per-lane constants; per-channel biases become 4 x 4 values per lane (from LDS).  Every output element is
// the same chain of MFMA dot products (a . b = b . a exactly; the k order inside the MFMA does not depend on which
// operand is called A), so the transposed and the plain form are bit-identical (tests/test_gpu_round4.py: the fused
// encoder vs the materialising forms, torch.equal).
template <int RB, int CB, int K, bool PIN = false, bool TRANS = false, int R>
__device__ __forceinline__ void hgemm_ring_run(HRing<CB, R>& r, const Planes& A, int ldh, f32x16 (&accm)[RB][CB],
                                               f32x16 (&accx)[RB][CB]) {
    constexpr int KS = K / 16;
    const int lane = threadIdx.x & 63;
    const int aoff = (lane & 31) * ldh + (lane >> 5) * 8;
    static_for<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int slot = ks % R;
        constexpr int younger = ((ks + R <= KS) ? R : KS - ks) - 1;
        f32x4 ah[RB], al[RB];
        int ao = aoff;
        if constexpr (PIN) asm volatile("" : "+v"(ao));
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            ah[rb] = *reinterpret_cast<const f32x4*>(A.hi + ao + rb * 32 * ldh + ks * 16);
            al[rb] = *reinterpret_cast<const f32x4*>(A.lo + ao + rb * 32 * ldh + ks * 16);
        }
        hring_wait<younger * CB * 2, CB>(r.b[slot]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                if constexpr (TRANS) {
                    accx[rb][cb] = mfma_h32(r.b[slot][cb][1], ah[rb], accx[rb][cb]);
                    accm[rb][cb] = mfma_h32(r.b[slot][cb][0], ah[rb], accm[rb][cb]);
#ifndef TSD_MFMA2  // (timing experiment, wrong results: two of the three MFMAs of a product -- tools/energy_probe.sh)
                    accx[rb][cb] = mfma_h32(r.b[slot][cb][0], al[rb], accx[rb][cb]);
#endif
                } else {
                    accx[rb][cb] = mfma_h32(ah[rb], r.b[slot][cb][1], accx[rb][cb]);
                    accm[rb][cb] = mfma_h32(ah[rb], r.b[slot][cb][0], accm[rb][cb]);
                    accx[rb][cb] = mfma_h32(al[rb], r.b[slot][cb][0], accx[rb][cb]);
                }
            }
        if constexpr (ks + R < KS)
            hring_issue<CB>(r.b[slot], r.base + (size_t)(ks + R) * r.step_bytes, r.voff, r.plane_bytes, r.cb_bytes);
        if constexpr (PIN) asm volatile("" : "+v"(accx[RB - 1][CB - 1]));  // (... and the step's MFMAs ahead of the next step's reads)
    });
}
template <int RB, int CB, int K, bool PIN = false, int R = HRING_R, bool TRANS = false>
__device__ __forceinline__ void hgemm_tile(const Planes& A, int ldh, const float* __restrict__ Bp16, int nout, int col0,
                                           f32x16 (&accm)[RB][CB], f32x16 (&accx)[RB][CB]) {
    HRing<CB, R> r;
    hgemm_ring_start<CB, K>(r, Bp16, nout, col0);
    hgemm_ring_run<RB, CB, K, PIN, TRANS>(r, A, ldh, accm, accx);
}
// TRANS accumulators: channel offset (inside the wave's 32-column slice) of accumulator registers 4 g .. 4 g + 3
__device__ __forceinline__ int tacc_col(int g, int hi) { return 8 * g + 4 * hi; }
template <int RB, int CB>
__device__ __forceinline__ void hzero(f32x16 (&accm)[RB][CB], f32x16 (&accx)[RB][CB]) {
    zero_acc(accm);
    zero_acc(accx);
}
__device__ __forceinline__ float hval(const f32x16& m, const f32x16& x, int r) { return fmaf(x[r], SPLIT_INV, m[r]); }

// ---- 16-row blocks: v_mfma_f32_16x16x32_f16 (node tiles).  A lane l: row l & 15, k = 8 (l >> 4) .. + 7 of a 32-k step;
// B lane l: column l & 15, the same k; C/D: col = l & 15, row = (l >> 4) * 4 + r.  One 32-k step = two 16-k steps of the
// packed image: lane quarter q takes k-step 2 s + (q >> 1), half q & 1.  CB = 16-wide column blocks of this wave.
template <int CB, int K, int R = HRING16_R>
__device__ __forceinline__ void hgemm16_ring_start(HRing<CB, R>& r, const float* __restrict__ Bp16, int nout, int col0) {
    const int lane = threadIdx.x & 63, q = lane >> 4;
    r.base = uniform_ptr(Bp16);
    r.voff = (unsigned)((q >> 1) * 64 * nout + ((q & 1) * nout + col0 + (lane & 15)) * 16);
    r.step_bytes = 128 * nout;  // two 16-k steps
#ifdef TSD_RING_WFAKE
    r.step_bytes = 0;
#endif
    r.plane_bytes = 32 * nout;
    r.cb_bytes = 16 * 16;
    constexpr int KS = K / 32;
    static_for<0, (R < KS ? R : KS)>([&](auto i) {
        constexpr int I = decltype(i)::value;
        hring_issue<CB>(r.b[I], r.base + (size_t)I * r.step_bytes, r.voff, r.plane_bytes, r.cb_bytes);
    });
}
// TRANS: as hgemm_ring_run -- lane l then holds tile row l & 15 and the four consecutive channels 4 (l >> 4) + r
template <int CB, int K, bool TRANS = false>
__device__ __forceinline__ void hgemm16_ring_run(HRing<CB, HRING16_R>& r, const Planes& A, int ldh, f32x4 (&accm)[CB],
                                                 f32x4 (&accx)[CB]) {
    constexpr int R = HRING16_R, KS = K / 32;
    const int lane = threadIdx.x & 63;
    const int aoff = (lane & 15) * ldh + (lane >> 4) * 8;
    static_for<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int slot = ks % R;
        constexpr int younger = ((ks + R <= KS) ? R : KS - ks) - 1;
        const f32x4 ah = *reinterpret_cast<const f32x4*>(A.hi + aoff + ks * 32);
        const f32x4 al = *reinterpret_cast<const f32x4*>(A.lo + aoff + ks * 32);
        hring_wait<younger * CB * 2, CB>(r.b[slot]);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if constexpr (TRANS) {
                accx[cb] = mfma_h16(r.b[slot][cb][1], ah, accx[cb]);
                accm[cb] = mfma_h16(r.b[slot][cb][0], ah, accm[cb]);
                accx[cb] = mfma_h16(r.b[slot][cb][0], al, accx[cb]);
            } else {
                accx[cb] = mfma_h16(ah, r.b[slot][cb][1], accx[cb]);
                accm[cb] = mfma_h16(ah, r.b[slot][cb][0], accm[cb]);
                accx[cb] = mfma_h16(al, r.b[slot][cb][0], accx[cb]);
            }
        }
        if constexpr (ks + R < KS)
            hring_issue<CB>(r.b[slot], r.base + (size_t)(ks + R) * r.step_bytes, r.voff, r.plane_bytes, r.cb_bytes);
    });
}
// 16-row MFMA GEMM on RB16 row blocks sharing the weight ring (split16.hpp hgemm16_ring_run per row block: the same MFMA
// sequence per output element)
template <int RB16, int CB, int K, bool TRANS = false, int R = HRING16_R>
__device__ __forceinline__ void hgemm16_ring_run_rb(HRing<CB, R>& r, const Planes& A, int ldh, f32x4 (&accm)[RB16][CB],
                                                    f32x4 (&accx)[RB16][CB], int nrb /* row blocks that hold rows (uniform) */) {
    constexpr int KS = K / 32;
    const int lane = threadIdx.x & 63;
    const int aoff = (lane & 15) * ldh + (lane >> 4) * 8;
    static_for<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        constexpr int slot = ks % R;
        constexpr int younger = ((ks + R <= KS) ? R : KS - ks) - 1;
        f32x4 ah[RB16], al[RB16];
#pragma unroll
        for (int rb = 0; rb < RB16; ++rb)
            if (rb < nrb) {
                ah[rb] = *reinterpret_cast<const f32x4*>(A.hi + aoff + rb * 16 * ldh + ks * 32);
                al[rb] = *reinterpret_cast<const f32x4*>(A.lo + aoff + rb * 16 * ldh + ks * 32);
            }
        hring_wait<younger * CB * 2, CB>(r.b[slot]);
#pragma unroll
        for (int rb = 0; rb < RB16; ++rb)
            if (rb < nrb) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    if constexpr (TRANS) {
                        accx[rb][cb] = mfma_h16(r.b[slot][cb][1], ah[rb], accx[rb][cb]);
                        accm[rb][cb] = mfma_h16(r.b[slot][cb][0], ah[rb], accm[rb][cb]);
                        accx[rb][cb] = mfma_h16(r.b[slot][cb][0], al[rb], accx[rb][cb]);
                    } else {
                        accx[rb][cb] = mfma_h16(ah[rb], r.b[slot][cb][1], accx[rb][cb]);
                        accm[rb][cb] = mfma_h16(ah[rb], r.b[slot][cb][0], accm[rb][cb]);
                        accx[rb][cb] = mfma_h16(al[rb], r.b[slot][cb][0], accx[rb][cb]);
                    }
                }
            }
        if constexpr (ks + R < KS)
            hring_issue<CB>(r.b[slot], r.base + (size_t)(ks + R) * r.step_bytes, r.voff, r.plane_bytes, r.cb_bytes);
    });
}

template <int CB, int K>
__device__ __forceinline__ void hgemm16_tile(const Planes& A, int ldh, const float* __restrict__ Bp16, int nout, int col0,
                                             f32x4 (&accm)[CB], f32x4 (&accx)[CB]) {
    HRing<CB, HRING16_R> r;
    hgemm16_ring_start<CB, K>(r, Bp16, nout, col0);
    hgemm16_ring_run<CB, K>(r, A, ldh, accm, accx);
}
__device__ __forceinline__ float hval4(const f32x4& m, const f32x4& x, int r) { return fmaf(x[r], SPLIT_INV, m[r]); }

}  // namespace tsd

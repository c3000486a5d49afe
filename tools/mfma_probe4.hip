// mfma_probe4.hip -- fp32-accurate tile GEMM on the 16-bit MFMA pipes (16x the fp32 MFMA rate on gfx950) by operand
// splitting, in the shape the product kernels use: 8 waves x 32 columns, RB row blocks of 32 per wave (tile of 32/64 rows,
// K = N = 256), A planes in LDS, B planes packed in L2 and fed by a counted asm ring.
//   bf16 x 3 planes: a = a0 + a1 + a2 (8 bits each), 6 products a0b0 a0b1 a1b0 a1b1 a0b2 a2b0   (error ~ 2^-24, fp32 range)
//   f16  x 2 planes: a = a0 + a1 (11 bits each),     3 products a0b0 a0b1 a1b0                  (error ~ 2^-22, f16 range)
// Reports the fp32-equivalent TFLOP/s (2*M*N*K per GEMM) and the error against an fp64 GEMM next to the plain fp32 one.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe4.hip -o tools/bin/mfma_probe4
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
constexpr int H = 256, KS = H / 16, LDA16 = H + 8;  // 16-bit elements per LDS row (stride = 132 words: b128 reads conflict-free)

template <int V>
struct IntC { static constexpr int value = V; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<I + 1, N>(f);
    }
}

template <bool BF>
__device__ __forceinline__ f32x16 mfma16(const f32x4& a, const f32x4& b, f32x16 acc) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

template <bool BF>
__device__ __forceinline__ uint16_t to16(float x) {
    if constexpr (BF) { __bf16 v = (__bf16)x; return __builtin_bit_cast(uint16_t, v); }
    else { _Float16 v = (_Float16)x; return __builtin_bit_cast(uint16_t, v); }
}
template <bool BF>
__device__ __forceinline__ float from16(uint16_t u) {
    if constexpr (BF) return (float)__builtin_bit_cast(__bf16, u);
    else return (float)__builtin_bit_cast(_Float16, u);
}

// packed B planes: [ks][plane][hi][col][8] 16-bit   (wave fragment: two 512-byte runs)
template <bool BF, int NP>
__global__ void pack_b(const float* __restrict__ W /* [out][in] */, uint16_t* __restrict__ Bp, int nout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one (col, k) per thread
    if (idx >= nout * H) return;
    const int col = idx / H, k = idx % H;
    float r = W[(size_t)col * H + k];
    const int ks = k / 16, hi = (k % 16) / 8, e = k % 8;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint16_t u = to16<BF>(r);
        r -= from16<BF>(u);
        Bp[((((size_t)ks * NP + p) * 2 + hi) * nout + col) * 8 + e] = u;
    }
}

template <bool BF, int NP, int RB, int R, bool REGS_ONLY>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ A, const uint16_t* __restrict__ Bp, int nrep,
                                             float* __restrict__ out, float* __restrict__ result) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    constexpr int ROWS = RB * 32, PLANE = ROWS * LDA16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    // stage + split the A tile (the product kernels do this in their epilogues)
    for (int idx = tid; idx < ROWS * H; idx += 512) {
        const int row = idx / H, k = idx % H;
        float r = A[(size_t)(blockIdx.x % 4) * ROWS * H + idx];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint16_t u = to16<BF>(r);
            r -= from16<BF>(u);
            smem[p * PLANE + row * LDA16 + k] = u;
        }
    }
    __syncthreads();
    const int col0 = wave * 32;
    constexpr int nout = H;
    f32x16 acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
    const uint16_t* aptr = smem + l31 * LDA16 + hi * 8;
    unsigned voff = (unsigned)((hi * nout + col0 + l31) * 16);
    constexpr int KSB = NP * 2 * nout * 16;  // bytes per k-step
    constexpr int PB = 2 * nout * 16;        // bytes per plane inside a k-step
    for (int rep = 0; rep < nrep; ++rep) {
        const char* base = reinterpret_cast<const char*>(Bp) + (size_t)(rep & 1) * KS * KSB;
        f32x4 b[R][NP];
        if (!REGS_ONLY) {
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const char* sb = base + (size_t)i * KSB + p * PB;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[i][p]) : "v"(voff), "s"(sb) : "memory");
                }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[i][p] = *reinterpret_cast<const f32x4*>(base + voff + p * PB);
        }
        static_for<0, KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            constexpr int slot = ks % R;
            const unsigned vo = voff;  // (named use: clang does not capture a variable used only as an asm operand)
            const char* bs = base;
            f32x4 a[RB][NP];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    a[rb][p] = *reinterpret_cast<const f32x4*>(aptr + p * PLANE + rb * 32 * LDA16 + ks * 16);
            if constexpr (!REGS_ONLY) {
                constexpr int younger = ((ks + R <= KS) ? R : KS - ks) - 1;
                // counted wait: `younger` k-steps (NP loads each) were issued after this one
                if constexpr (NP == 3)
                    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b[slot][0]), "+v"(b[slot][1]), "+v"(b[slot][2]) : "n"(younger * NP) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[slot][0]), "+v"(b[slot][1]) : "n"(younger * NP) : "memory");
            }
            // small products first
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if constexpr (NP == 3) {
                    acc[rb] = mfma16<BF>(a[rb][0], b[slot][2], acc[rb]);
                    acc[rb] = mfma16<BF>(a[rb][2], b[slot][0], acc[rb]);
                    acc[rb] = mfma16<BF>(a[rb][1], b[slot][1], acc[rb]);
                }
                acc[rb] = mfma16<BF>(a[rb][0], b[slot][1], acc[rb]);
                acc[rb] = mfma16<BF>(a[rb][1], b[slot][0], acc[rb]);
                acc[rb] = mfma16<BF>(a[rb][0], b[slot][0], acc[rb]);
            }
            if constexpr (!REGS_ONLY && ks + R < KS) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const char* sb = bs + (size_t)(ks + R) * KSB + p * PB;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[slot][p]) : "v"(vo), "s"(sb) : "memory");
                }
            }
        });
    }
    float s = 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[rb][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (result && blockIdx.x == 0 && nrep == 1) {
        // C/D layout of the 32x32 MFMA: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                result[(size_t)(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * H + col0 + l31] = acc[rb][r];
    }
}


// Register-blocked form (round 6): a wave owns RB row blocks x CB column blocks (RB x CB accumulators) and loads RB A fragments
// (LDS) + CB B fragments (ring) per k-step for RB x CB products: with 2 x 2 the operand bytes per MFMA halve against the
// product's 2 x 1.  NW waves per workgroup (NW x CB x 32 = 256 columns), tile of RB x 32 rows, f16 x 2 planes only.
template <int RB, int CB, int NW, int R>
__global__ __launch_bounds__(NW * 64) void probe_blk(const float* __restrict__ A, const uint16_t* __restrict__ Bp, int nrep,
                                                      float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    constexpr int NP = 2, ROWS = RB * 32, PLANE = ROWS * LDA16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    for (int idx = tid; idx < ROWS * H; idx += NW * 64) {
        const int row = idx / H, k = idx % H;
        float r = A[(size_t)(blockIdx.x % 4) * 32 * H + idx % (128 * H)];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint16_t u = to16<false>(r);
            r -= from16<false>(u);
            smem[p * PLANE + row * LDA16 + k] = u;
        }
    }
    __syncthreads();
    const int col0 = wave * CB * 32;
    constexpr int nout = H;
    f32x16 acc[RB][CB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
    const uint16_t* aptr = smem + l31 * LDA16 + hi * 8;
    unsigned voff = (unsigned)((hi * nout + col0 + l31) * 16);
    constexpr int KSB = NP * 2 * nout * 16, PB = 2 * nout * 16;
    for (int rep = 0; rep < nrep; ++rep) {
        const char* base = reinterpret_cast<const char*>(Bp) + (size_t)(rep & 1) * KS * KSB;
        f32x4 b[R][CB][NP];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const char* sb = base + (size_t)i * KSB + p * PB + cb * 512;
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[i][cb][p]) : "v"(voff), "s"(sb) : "memory");
                }
        static_for<0, KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            constexpr int slot = ks % R;
            const unsigned vo = voff;
            const char* bs = base;
            f32x4 a[RB][NP];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    a[rb][p] = *reinterpret_cast<const f32x4*>(aptr + p * PLANE + rb * 32 * LDA16 + ks * 16);
            constexpr int younger = ((ks + R <= KS) ? R : KS - ks) - 1;
            if constexpr (CB == 2)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[slot][0][0]), "+v"(b[slot][0][1]), "+v"(b[slot][1][0]), "+v"(b[slot][1][1]) : "n"(younger * NP * CB) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[slot][0][0]), "+v"(b[slot][0][1]) : "n"(younger * NP * CB) : "memory");
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    acc[rb][cb] = mfma16<false>(a[rb][0], b[slot][cb][1], acc[rb][cb]);
                    acc[rb][cb] = mfma16<false>(a[rb][1], b[slot][cb][0], acc[rb][cb]);
                    acc[rb][cb] = mfma16<false>(a[rb][0], b[slot][cb][0], acc[rb][cb]);
                }
            if constexpr (ks + R < KS) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const char* sb = bs + (size_t)(ks + R) * KSB + p * PB + cb * 512;
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[slot][cb][p]) : "v"(vo), "s"(sb) : "memory");
                    }
            }
        });
    }
    float s = 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[rb][cb][r];
    out[(size_t)blockIdx.x * NW * 64 + tid] = s;
}

template <int RB, int CB, int NW, int R>
static void run_blk(const char* name, int grid, int nrep, const float* A, const uint16_t* Bp, float* out, int wg_per_cu) {
    constexpr int ROWS = RB * 32;
    size_t lds = (size_t)2 * ROWS * LDA16 * 2;
    const size_t want = (size_t)160 * 1024 / wg_per_cu - 512;
    if (lds < want) lds = want;
    if (lds > 160 * 1024) { printf("%-30s skipped (LDS)\n", name); return; }
    auto kern = probe_blk<RB, CB, NW, R>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, A, Bp, nrep, out);
    hipEventRecord(e0, 0);
    const int it = 10;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, A, Bp, nrep, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    const double flop = (double)grid * nrep * ROWS * H * H * 2.0;
    printf("%-22s rows %3d = %d x 32, %d col blocks per wave, %d waves, R %d, wg/cu %d, grid %5d: %8.2f us  %7.1f TFLOP/s(fp32-eq)\n", name,
           ROWS, RB, CB, NW, R, wg_per_cu, grid, ms * 1e3, flop / (ms * 1e-3) / 1e12);
}

static std::vector<float> hA, hW;

template <bool BF, int NP, int RB, int R, bool REGS>
static void run(const char* name, int grid, int nrep, const float* A, const uint16_t* Bp, float* out, int wg_per_cu,
                float* result = nullptr) {
    constexpr int ROWS = RB * 32;
    size_t lds = (size_t)NP * ROWS * LDA16 * 2;
    const size_t want = (size_t)160 * 1024 / wg_per_cu - 512;
    if (lds < want && wg_per_cu < 4) lds = want;
    if (lds > 160 * 1024) { printf("%-30s skipped (LDS)\n", name); return; }
    auto kern = probe<BF, NP, RB, R, REGS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, A, Bp, nrep, out, result);
    hipEventRecord(e0, 0);
    const int it = 10;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, A, Bp, nrep, out, result);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    const double flop = (double)grid * nrep * ROWS * H * H * 2.0;
    printf("%-30s rows %3d R %d wg/cu %d grid %5d nrep %2d : %8.2f us  %7.1f TFLOP/s(fp32-eq)\n", name, ROWS, R, wg_per_cu,
           grid, nrep, ms * 1e3, flop / (ms * 1e-3) / 1e12);
    if (result) {
        std::vector<float> h((size_t)ROWS * H);
        hipMemcpy(h.data(), result, h.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, rmax = 0, e32 = 0;
        for (int i = 0; i < ROWS; ++i)
            for (int j = 0; j < H; ++j) {
                double ref = 0;
                float f = 0.f;
                for (int k = 0; k < H; ++k) {
                    ref += (double)hA[(size_t)i * H + k] * (double)hW[(size_t)j * H + k];
                    f = fmaf(hA[(size_t)i * H + k], hW[(size_t)j * H + k], f);
                }
                emax = fmax(emax, fabs(h[(size_t)i * H + j] - ref));
                e32 = fmax(e32, fabs((double)f - ref));
                rmax = fmax(rmax, fabs(ref));
            }
        printf("    max|err| / max|ref| vs fp64: split %.3e   fp32 fma chain %.3e\n", emax / rmax, e32 / rmax);
    }
}

template <bool BF, int NP>
static void suite(const char* tag, const float* A, const float* W, float* out, float* result) {
    uint16_t* Bp;
    hipMalloc(&Bp, (size_t)2 * H * H * NP * 2);
    for (int m = 0; m < 2; ++m)
        hipLaunchKernelGGL((pack_b<BF, NP>), dim3(H * H / 256), dim3(256), 0, 0, W + (size_t)m * H * H,
                           Bp + (size_t)m * H * H * NP, H);
    hipDeviceSynchronize();
    char name[64];
    // accuracy of one GEMM (block 0, nrep 1)
    snprintf(name, sizeof name, "%s accuracy", tag);
    run<BF, NP, 1, 4, false>(name, 4, 1, A, Bp, out, 1, result);
    for (int wpc : {1, 2}) {
        snprintf(name, sizeof name, "%s regs-only", tag);
        run<BF, NP, 1, 1, true>(name, 2048, 16, A, Bp, out, wpc);
        run<BF, NP, 2, 1, true>(name, 2048, 16, A, Bp, out, wpc);
        snprintf(name, sizeof name, "%s ring", tag);
        run<BF, NP, 1, 4, false>(name, 2048, 16, A, Bp, out, wpc);
        run<BF, NP, 1, 6, false>(name, 2048, 16, A, Bp, out, wpc);
        run<BF, NP, 2, 4, false>(name, 2048, 16, A, Bp, out, wpc);
        run<BF, NP, 2, 6, false>(name, 2048, 16, A, Bp, out, wpc);
        if (NP == 2 || wpc == 1) run<BF, NP, 4, 4, false>(name, 1024, 16, A, Bp, out, wpc);
    }
    // one round of tiles, two GEMMs each (a filter tile's nn.0 / nn.2): latency of a batch-100-sized launch
    snprintf(name, sizeof name, "%s one-round", tag);
    run<BF, NP, 1, 4, false>(name, 408, 2, A, Bp, out, 2);
    run<BF, NP, 2, 4, false>(name, 204, 2, A, Bp, out, 1);
    run<BF, NP, 2, 4, false>(name, 256, 2, A, Bp, out, 1);
    run<BF, NP, 1, 4, false>(name, 256, 2, A, Bp, out, 1);
    run<BF, NP, 1, 4, false>(name, 256, 6, A, Bp, out, 1);
    if (NP == 2 && !BF) {
        // register-blocked forms: operand bytes per MFMA against the product's 2 x 1 (RB x CB)
        for (int wpc : {1, 2}) {
            run_blk<2, 1, 8, 3>("f16x2 blk 2x1 (product)", 2048, 16, A, Bp, out, wpc);
            run_blk<2, 2, 4, 3>("f16x2 blk 2x2", 2048, 16, A, Bp, out, wpc);
            run_blk<2, 2, 4, 2>("f16x2 blk 2x2", 2048, 16, A, Bp, out, wpc);
            run_blk<4, 2, 4, 2>("f16x2 blk 4x2", 1024, 16, A, Bp, out, wpc);
            run_blk<4, 1, 8, 3>("f16x2 blk 4x1", 1024, 16, A, Bp, out, wpc);
        }
    }
    printf("\n");
    hipFree(Bp);
}

int main() {
    float *A, *W, *out, *result;
    hA.resize((size_t)4 * 128 * H);
    hW.resize((size_t)2 * H * H);
    srand(1);
    for (auto& v : hA) {  // shifted-softplus-like activations: mostly small positive, some large
        const float g = (rand() / (float)RAND_MAX) * 6.f - 3.f;
        v = log1pf(expf(g)) - 0.6931472f;
    }
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.125f;
    hipMalloc(&A, hA.size() * 4);
    hipMalloc(&W, hW.size() * 4);
    hipMalloc(&out, (size_t)8192 * 512 * 4);
    hipMalloc(&result, (size_t)128 * H * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    suite<true, 3>("bf16x3", A, W, out, result);
    suite<false, 2>("f16x2", A, W, out, result);
    return 0;
}

// mfma_probe3.hip -- the B-operand feed of the fp32 tile GEMM in the shape the product kernels use it:
// 8 waves x 32 columns, ONE 32x32 accumulator per wave (tile of 32 rows, K = N = 256), A from LDS, B packed in L2.
// Variants of the k-loop (same arithmetic, same result):
//   0  registers only (ceiling of the instruction stream)
//   1  product code today: per-lane 64-bit pointer arithmetic, compiler-scheduled (it keeps 2 loads in flight)
//   2  scalar base + 32-bit lane offset (no VALU address arithmetic between the MFMAs), compiler-scheduled
//   3  explicit ring of R in-flight loads issued by inline asm with counted vmcnt (R = 4 / 8)
//   4  two row blocks per wave (64-row tile): every B fragment feeds two alternating accumulators
//   5  B through a wave-private LDS ring filled by LDS-DMA (global_load_lds_dwordx4), R slots
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe3.hip -o tools/bin/mfma_probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
constexpr int H = 256, LDA = H + 4, KB = H / 8;

__device__ __forceinline__ void mfma4(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
}

template <int VAR, int R>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ A, const float* __restrict__ W, int nrep,
                                             float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = VAR == 4 ? 64 : 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    for (int idx = tid; idx < ROWS * H; idx += 512) smem[(idx / H) * LDA + (idx % H)] = A[idx % (32 * H)];
    __syncthreads();
    const int col0 = wave * 32;
    constexpr int nout = H;
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    const float* aptr = smem + l31 * LDA + hi * 4;
    const f32x4 areg = *reinterpret_cast<const f32x4*>(aptr);
    const f32x4 breg = *reinterpret_cast<const f32x4*>(W + lane * 4);
    const unsigned lane_off = (unsigned)(hi * nout + col0 + l31);  // float4 units
    for (int rep = 0; rep < nrep; ++rep) {
        const float* Wr = W + (size_t)(rep & 1) * H * nout;  // two weight matrices alternate (as nn0 / nn2 do)
        if (VAR == 0) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) mfma4(acc, areg, breg);
        } else if (VAR == 1) {
            const f32x4* bptr = reinterpret_cast<const f32x4*>(Wr) + (size_t)hi * nout + col0 + l31;
            constexpr int PF = 4, NC = KB / PF;
            f32x4 b0[PF], b1[PF];
            auto loadB = [&](f32x4 (&b)[PF], int chunk) {
#pragma unroll
                for (int p = 0; p < PF; ++p) b[p] = bptr[(size_t)(chunk * PF + p) * 2 * nout];
            };
            auto compute = [&](const f32x4 (&b)[PF], int chunk) {
                f32x4 a[PF];
#pragma unroll
                for (int p = 0; p < PF; ++p) a[p] = *reinterpret_cast<const f32x4*>(aptr + (chunk * PF + p) * 8);
#pragma unroll
                for (int p = 0; p < PF; ++p) mfma4(acc, a[p], b[p]);
            };
            loadB(b0, 0);
            for (int c = 0; c < NC; c += 2) {
                if (c + 1 < NC) loadB(b1, c + 1);
                compute(b0, c);
                if (c + 2 < NC) loadB(b0, c + 2);
                if (c + 1 < NC) compute(b1, c + 1);
            }
        } else if (VAR == 2 || VAR == 4) {
            const f32x4* Wq = reinterpret_cast<const f32x4*>(Wr);  // wave-uniform base
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f32x4 b = Wq[(size_t)kb * 2 * nout + lane_off];
                const f32x4 a = *reinterpret_cast<const f32x4*>(aptr + kb * 8);
                if (VAR == 4) {
                    const f32x4 a2 = *reinterpret_cast<const f32x4*>(aptr + 32 * LDA + kb * 8);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[s], b[s], acc2, 0, 0, 0);
                    }
                } else {
                    mfma4(acc, a, b);
                }
            }
        } else if (VAR == 3) {
            // ring of R loads in flight, issued by asm (the compiler neither sinks nor counts them), counted vmcnt
            const unsigned voff = lane_off * 16u;
            f32x4 b[R];
            const char* base = reinterpret_cast<const char*>(Wr);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const char* sb = base + (size_t)i * 2 * nout * 16;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[i]) : "v"(voff), "s"(sb) : "memory");
            }
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(aptr + kb * 8);
                // the oldest outstanding load is k-block kb: everything but the R-1 younger ones must have landed
                const int younger = (kb + R <= KB ? R : KB - kb) - 1;
                if (younger >= 7) asm volatile("s_waitcnt vmcnt(7)" : "+v"(b[kb % R]) :: "memory");
                else if (younger == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(b[kb % R]) :: "memory");
                else if (younger == 5) asm volatile("s_waitcnt vmcnt(5)" : "+v"(b[kb % R]) :: "memory");
                else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(b[kb % R]) :: "memory");
                else if (younger == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(b[kb % R]) :: "memory");
                else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(b[kb % R]) :: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(b[kb % R]) :: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[kb % R]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mfma4(acc, a, b[kb % R]);
                if (kb + R < KB) {
                    const char* sb = base + (size_t)(kb + R) * 2 * nout * 16;
                    // the MFMAs above read b[kb % R] at issue; the reload lands hundreds of cycles later
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[kb % R]) : "v"(voff), "s"(sb) : "memory");
                }
            }
        } else if (VAR == 5) {
            // wave-private LDS ring of R slots (1 KiB each: 64 lanes x float4), filled by LDS-DMA
            float* ring = smem + ROWS * LDA + wave * R * 256;
            const f32x4* Wq = reinterpret_cast<const f32x4*>(Wr);
            auto dma = [&](int kb) {
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(Wq + (size_t)kb * 2 * nout + lane_off),
                    (__attribute__((address_space(3))) void*)(ring + (kb % R) * 256), 16, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < R; ++i) dma(i);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(aptr + kb * 8);
                const int younger = (kb + R <= KB ? R : KB - kb) - 1;
                if (younger >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                else if (younger == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (younger == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (younger == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const f32x4 b = *reinterpret_cast<const volatile f32x4*>(ring + (kb % R) * 256 + lane * 4);
                mfma4(acc, a, b);
                if (kb + R < KB) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slot read before it is refilled
                    dma(kb + R);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

template <int VAR, int R>
static void run(const char* name, int grid, int nrep, const float* A, const float* W, float* out, int wg_per_cu) {
    constexpr int ROWS = VAR == 4 ? 64 : 32;
    size_t lds = (size_t)ROWS * LDA * 4 + (VAR == 5 ? 8 * R * 1024 : 0);
    const size_t want = (size_t)160 * 1024 / wg_per_cu - 512;  // pad so that exactly wg_per_cu workgroups fit a CU
    if (lds < want && wg_per_cu < 4) lds = want;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<VAR, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<VAR, R>), dim3(grid), dim3(512), lds, 0, A, W, nrep, out);
    hipEventRecord(e0, 0);
    const int it = 5;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((probe<VAR, R>), dim3(grid), dim3(512), lds, 0, A, W, nrep, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    const double flop = (double)grid * nrep * ROWS * H * H * 2.0;
    std::vector<float> h(512);
    hipMemcpy(h.data(), out, 512 * 4, hipMemcpyDeviceToHost);
    double cs = 0;
    for (float v : h) cs += v;
    printf("%-34s R %d  wg/cu %d : %8.1f us  %6.1f TFLOP/s   checksum %.6e\n", name, R, wg_per_cu, ms * 1e3,
           flop / (ms * 1e-3) / 1e12, cs);
}

int main() {
    float *A, *W, *out;
    std::vector<float> hA(64 * H), hW(2 * H * H);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    hipMalloc(&A, hA.size() * 4);
    hipMalloc(&W, hW.size() * 4);
    hipMalloc(&out, (size_t)8192 * 512 * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    for (int wpc : {1, 2, 3}) {
        const int grid = 2048, nrep = 16;
        run<0, 1>("regs only", grid, nrep, A, W, out, wpc);
        run<1, 1>("product loop (ptr arith, 2 in flight)", grid, nrep, A, W, out, wpc);
        run<2, 1>("scalar base + lane offset", grid, nrep, A, W, out, wpc);
        run<3, 4>("asm ring, counted vmcnt", grid, nrep, A, W, out, wpc);
        run<3, 8>("asm ring, counted vmcnt", grid, nrep, A, W, out, wpc);
        if (wpc <= 2) run<4, 1>("two row blocks (64-row tile)", grid, nrep, A, W, out, wpc);
        run<5, 4>("B via LDS-DMA ring", grid, nrep, A, W, out, wpc);
        if (wpc <= 2) run<5, 8>("B via LDS-DMA ring", grid, nrep, A, W, out, wpc);
        printf("\n");
    }
    return 0;
}

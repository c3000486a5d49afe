// mfma_probe.hip -- microbenchmark of the tile-GEMM inner loop used by kernels_mlp.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe.hip -o gpurun_out/mfma_probe
// Each workgroup: A tile [T x 256] resident in LDS, NREP back-to-back GEMMs against a packed
// 256x256 weight streamed from L2, accumulators summed into a checksum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));

constexpr int H = 256;

// ---- variant 0: the round-1 loop (prefetch distance one k-block) -------------------------------
template <int RB, int CB, int K>
__device__ __forceinline__ void gemm_v0(const float* __restrict__ ldsA, int lda, const float* __restrict__ Bp,
                                        int nout, int col0, f32x16 (&acc)[RB][CB], int) {
    const int lane = threadIdx.x & 63, hi = lane >> 5, l31 = lane & 31;
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
        for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(aptr + rb * 32 * lda + kb * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][s], bcur[cb][s], acc[rb][cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) bcur[cb] = bnxt[cb];
    }
}

// ---- variant 1/2: chunked double buffer: B for PF k-blocks in flight while PF k-blocks compute ----
template <int RB, int CB, int K, int PF, bool STAGGER>
__device__ __forceinline__ void gemm_v1(const float* __restrict__ ldsA, int lda, const float* __restrict__ Bp,
                                        int nout, int col0, f32x16 (&acc)[RB][CB], int rot) {
    const int lane = threadIdx.x & 63, hi = lane >> 5, l31 = lane & 31;
    const float* aptr = ldsA + l31 * lda + hi * 4;
    const f32x4* bptr = reinterpret_cast<const f32x4*>(Bp) + (size_t)hi * nout + col0 + l31;
    constexpr int KB = K / 8;
    constexpr int NC = KB / PF;  // chunks
    static_assert(KB % PF == 0, "");
    const int c0 = STAGGER ? (rot % NC) : 0;
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
    auto wrap = [&](int c) { int x = c0 + c; return x >= NC ? x - NC : x; };
    loadB(b0, wrap(0));
    for (int c = 0; c < NC; c += 2) {
        if (c + 1 < NC) loadB(b1, wrap(c + 1));
        compute(b0, wrap(c));
        if (c + 2 < NC) loadB(b0, wrap(c + 2));
        if (c + 1 < NC) compute(b1, wrap(c + 1));
    }
}

template <int VAR, int RB>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ A, const float* __restrict__ W, int nrep,
                                             float* __restrict__ out) {
    constexpr int T = 32 * RB, LDA = H + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < T * H; idx += 256) smem[(idx / H) * LDA + (idx % H)] = A[idx];
    __syncthreads();
    const int col0 = (tid >> 6) * 64;
    f32x16 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
    for (int rep = 0; rep < nrep; ++rep) {
        const float* Wr = W + (size_t)(rep & 1) * H * H;
        if (VAR == 0) gemm_v0<RB, 2, H>(smem, LDA, Wr, H, col0, acc, 0);
        if (VAR == 1) gemm_v1<RB, 2, H, 4, false>(smem, LDA, Wr, H, col0, acc, 0);
        if (VAR == 2) gemm_v1<RB, 2, H, 4, true>(smem, LDA, Wr, H, col0, acc, blockIdx.x + rep);
        if (VAR == 3) gemm_v1<RB, 2, H, 8, true>(smem, LDA, Wr, H, col0, acc, blockIdx.x + rep);
        if (VAR == 4) gemm_v1<RB, 2, H, 2, true>(smem, LDA, Wr, H, col0, acc, blockIdx.x + rep);
    }
    float s = 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[rb][cb][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// 8 waves x 32 columns (the layout of the fused per-block launch's filter role)
template <int PF>
__global__ __launch_bounds__(512) void probe8(const float* __restrict__ A, const float* __restrict__ W, int nrep,
                                              float* __restrict__ out) {
    constexpr int T = 32, LDA = H + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < T * H; idx += 512) smem[(idx / H) * LDA + (idx % H)] = A[idx];
    __syncthreads();
    const int col0 = (tid >> 6) * 32;
    f32x16 acc[1][1];
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    for (int rep = 0; rep < nrep; ++rep) {
        const float* Wr = W + (size_t)(rep & 1) * H * H;
        gemm_v1<1, 1, H, PF, false>(smem, LDA, Wr, H, col0, acc, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][0][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

template <int PF>
static void run8(const char* name, int grid, int nrep, const float* A, const float* W, float* out) {
    const size_t lds = (size_t)32 * (H + 4) * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe8<PF>), dim3(grid), dim3(512), lds, 0, A, W, nrep, out);
    hipEventRecord(e0, 0);
    const int it = 5;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((probe8<PF>), dim3(grid), dim3(512), lds, 0, A, W, nrep, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    const double flop = (double)grid * nrep * 32.0 * H * H * 2.0;
    printf("%-34s grid %5d nrep %2d      : %8.1f us  %6.1f TFLOP/s\n", name, grid, nrep, ms * 1e3,
           flop / (ms * 1e-3) / 1e12);
}

template <int VAR, int RB>
static void run(const char* name, int grid, int nrep, const float* A, const float* W, float* out) {
    const size_t lds = (size_t)32 * RB * (H + 4) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<VAR, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<VAR, RB>), dim3(grid), dim3(256), lds, 0, A, W, nrep, out);
    hipEventRecord(e0, 0);
    const int it = 5;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((probe<VAR, RB>), dim3(grid), dim3(256), lds, 0, A, W, nrep, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    const double flop = (double)grid * nrep * 32.0 * RB * H * H * 2.0;
    float h;
    hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
    printf("%-34s grid %5d nrep %2d RB %d : %8.1f us  %6.1f TFLOP/s  (chk %.3e)\n", name, grid, nrep, RB, ms * 1e3,
           flop / (ms * 1e-3) / 1e12, h);
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
    for (int g : {408, 815, 2048}) {
        run<1, 1>("4 waves x 64 cols, chunk4", g, 2, A, W, out);
        run8<4>("8 waves x 32 cols, chunk4", g, 2, A, W, out);
        run8<8>("8 waves x 32 cols, chunk8", g, 2, A, W, out);
        run8<2>("8 waves x 32 cols, chunk2", g, 2, A, W, out);
        printf("\n");
    }
    const int grids[] = {50, 256, 815, 2048, 4096};
    for (int g : grids) {
        run<0, 1>("v0 pf1", g, 2, A, W, out);
        run<4, 1>("v1 chunk2 stagger", g, 2, A, W, out);
        run<1, 1>("v1 chunk4", g, 2, A, W, out);
        run<2, 1>("v1 chunk4 stagger", g, 2, A, W, out);
        run<3, 1>("v1 chunk8 stagger", g, 2, A, W, out);
        run<0, 2>("v0 pf1 T64", g, 2, A, W, out);
        run<1, 2>("v1 chunk4 T64", g, 2, A, W, out);
        run<2, 2>("v1 chunk4 stagger T64", g, 2, A, W, out);
        printf("\n");
    }
    // long-running variant: the MFMA ceiling of this loop when launch/prologue costs vanish
    run<2, 1>("v1 chunk4 stagger, nrep 16", 2048, 16, A, W, out);
    run<2, 2>("v1 chunk4 stagger T64, nrep 16", 2048, 16, A, W, out);
    return 0;
}

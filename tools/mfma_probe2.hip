// mfma_probe2.hip -- where does the fp32 tile GEMM lose its MFMA rate?  Four nested variants of the same loop:
//   0: MFMAs on register operands only                 (the ceiling of the instruction stream)
//   1: + A operand from LDS  (ds_read_b128 per 4 MFMAs)
//   2: + B operand from L2   (global_load_dwordx4 per column block per 4 MFMAs), scheduler free
//   3: like 2 with the chunked pipeline pinned by sched_barrier
//   4: A and B operands both from LDS (B pre-staged once: no global loads in the loop) -- isolates the cost of
//      the global-load return path from the cost of feeding two operands
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_probe2.hip -o gpurun_out/mfma_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
constexpr int H = 256;

template <int VAR, int CB, int PF>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ A, const float* __restrict__ W, int nrep,
                                             float* __restrict__ out) {
    constexpr int LDA = H + 4, K = H, KB = K / 8, NC = KB / PF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    for (int idx = tid; idx < 32 * H; idx += 256) smem[(idx / H) * LDA + (idx % H)] = A[idx];
    if (VAR == 4) {  // stage this workgroup's B once: 4 waves x KB x CB x 64 lanes float4 = K * 4*CB*32 floats
        f32x4* bl = reinterpret_cast<f32x4*>(smem + 32 * LDA);
        const f32x4* src = reinterpret_cast<const f32x4*>(W);
        for (int idx = tid; idx < 4 * 8 * CB * 64; idx += 256) bl[idx] = src[idx];
    }
    __syncthreads();
    const int col0 = (tid >> 6) * (CB * 32);
    const int nout = 4 * CB * 32;
    f32x16 acc[CB];
    for (int cb = 0; cb < CB; ++cb)
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    const float* aptr = smem + l31 * LDA + hi * 4;
    f32x4 areg = *reinterpret_cast<const f32x4*>(aptr);
    f32x4 breg = *reinterpret_cast<const f32x4*>(W + lane * 4);
    for (int rep = 0; rep < nrep; ++rep) {
        const f32x4* bptr = reinterpret_cast<const f32x4*>(W + (size_t)(rep & 1) * H * nout) + (size_t)hi * nout + col0 + l31;
        if (VAR == 4) {
            // B for this wave: [KB][CB][64 lanes] float4 in LDS after the A tile (pre-staged below, outside the loop)
            const f32x4* bl = reinterpret_cast<const f32x4*>(smem + 32 * LDA) + (size_t)(tid >> 6) * 8 * CB * 64 + lane;  // 8 k-blocks staged, reused cyclically
#pragma unroll 4
            for (int kb = 0; kb < KB; ++kb) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(aptr + kb * 8);
                f32x4 b[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) b[cb] = bl[((kb & 7) * CB + cb) * 64];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[cb][s], acc[cb], 0, 0, 0);
            }
        } else if (VAR <= 1) {
#pragma unroll 4
            for (int kb = 0; kb < KB; ++kb) {
                f32x4 a = areg;
                if (VAR == 1) a = *reinterpret_cast<const f32x4*>(aptr + kb * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
                        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], breg[s], acc[cb], 0, 0, 0);
            }
        } else {
            f32x4 b0[PF][CB], b1[PF][CB];
            auto loadB = [&](f32x4 (&b)[PF][CB], int chunk) {
#pragma unroll
                for (int p = 0; p < PF; ++p)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) b[p][cb] = bptr[(size_t)(chunk * PF + p) * 2 * nout + cb * 32];
            };
            auto compute = [&](const f32x4 (&b)[PF][CB], int chunk) {
                f32x4 a[PF];
#pragma unroll
                for (int p = 0; p < PF; ++p) a[p] = *reinterpret_cast<const f32x4*>(aptr + (chunk * PF + p) * 8);
#pragma unroll
                for (int p = 0; p < PF; ++p)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][s], b[p][cb][s], acc[cb], 0, 0, 0);
            };
            loadB(b0, 0);
            for (int c = 0; c < NC; c += 2) {
                if (c + 1 < NC) loadB(b1, c + 1);
                if (VAR == 3) __builtin_amdgcn_sched_barrier(0);
                compute(b0, c);
                if (VAR == 3) __builtin_amdgcn_sched_barrier(0);
                if (c + 2 < NC) loadB(b0, c + 2);
                if (VAR == 3) __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < NC) compute(b1, c + 1);
                if (VAR == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int cb = 0; cb < CB; ++cb)
        for (int r = 0; r < 16; ++r) s += acc[cb][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int VAR, int CB, int PF>
static void run(const char* name, int grid, int nrep, const float* A, const float* W, float* out, int extra_lds) {
    const size_t lds = (size_t)32 * (H + 4) * 4 + extra_lds;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<VAR, CB, PF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<VAR, CB, PF>), dim3(grid), dim3(256), lds, 0, A, W, nrep, out);
    hipEventRecord(e0, 0);
    const int it = 5;
    for (int w = 0; w < it; ++w) hipLaunchKernelGGL((probe<VAR, CB, PF>), dim3(grid), dim3(256), lds, 0, A, W, nrep, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
    const double flop = (double)grid * nrep * 32.0 * H * (4.0 * CB * 32) * 2.0;
    printf("%-28s CB %d PF %2d grid %5d nrep %3d lds+%3dK : %8.1f us  %6.1f TFLOP/s\n", name, CB, PF, grid, nrep,
           extra_lds / 1024, ms * 1e3, flop / (ms * 1e-3) / 1e12);
}

int main() {
    float *A, *W, *out;
    std::vector<float> hA(64 * H), hW(2 * H * 512);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.1f;
    hipMalloc(&A, hA.size() * 4);
    hipMalloc(&W, hW.size() * 4);
    hipMalloc(&out, (size_t)8192 * 256 * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    // occupancy control: extra LDS so that 1 / 2 / 4 workgroups fit a CU (160 KB)
    for (int extra : {100 * 1024, 40 * 1024, 0}) {
        const int grid = 2048, nrep = 16;
        run<0, 2, 4>("regs only", grid, nrep, A, W, out, extra);
        run<1, 2, 4>("A from LDS", grid, nrep, A, W, out, extra);
        run<2, 2, 4>("A LDS + B L2 (free sched)", grid, nrep, A, W, out, extra);
        run<3, 2, 4>("A LDS + B L2 (pinned)", grid, nrep, A, W, out, extra);
        run<3, 2, 8>("A LDS + B L2 (pinned)", grid, nrep, A, W, out, extra);
        run<2, 4, 4>("A LDS + B L2 (free sched)", grid, nrep, A, W, out, extra);
        run<3, 4, 4>("A LDS + B L2 (pinned)", grid, nrep, A, W, out, extra);
        run<0, 4, 4>("regs only", grid, nrep, A, W, out, extra);
        run<0, 1, 4>("regs only", grid, nrep, A, W, out, extra);
        run<2, 1, 4>("A LDS + B L2 (free sched)", grid, nrep, A, W, out, extra);
        run<3, 1, 8>("A LDS + B L2 (pinned)", grid, nrep, A, W, out, extra);
        if (extra <= 40 * 1024) {  // + staged B: 4 waves x 8 k-blocks x CB x 64 lanes x 16 B
            run<4, 1, 4>("A LDS + B LDS", grid, nrep, A, W, out, extra + 32 * 1024);
            run<4, 2, 4>("A LDS + B LDS", grid, nrep, A, W, out, (extra ? 10 * 1024 : 0) + 64 * 1024);
        }
        printf("\n");
    }
    return 0;
}

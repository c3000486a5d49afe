// occupy.hip -- test helper (NOT part of the product): a second tenant that holds workgroup slots of the GPU for a
// given time, so that tests can run the library's forward while another process' kernel owns part of the chip
// (tests/test_gpu_round4.py).  Built on demand by the test: hipcc --offload-arch=gfx950 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void occupy_kernel(unsigned long long ticks /* 100 MHz wall clock */) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) smem[0] = 0.0f;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
}

extern "C" int occupy_launch(int blocks, int threads, int lds_bytes, double seconds, void* stream) {
    if (lds_bytes > 48 * 1024)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes) != hipSuccess)
            return -1;
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream,
                       (unsigned long long)(seconds * 1.0e8));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

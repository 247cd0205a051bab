#include <hip/hip_runtime.h>
__global__ __launch_bounds__(512, 2) void k_hwid(unsigned* out) {
    extern __shared__ float dyn[];
    const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    dyn[threadIdx.x] = id;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
extern "C" int hwid_launch(unsigned* out, int blocks, int lds, void* stream) {
    (void)hipFuncSetAttribute((const void*)k_hwid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_hwid, dim3(blocks), dim3(512), lds, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}

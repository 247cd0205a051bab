#include <hip/hip_runtime.h>
template <int HIREG>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters) {
    extern __shared__ float dyn[];
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (HIREG == 24) asm volatile("" ::: "v23");
    if (HIREG == 32) asm volatile("" ::: "v31");
    if (HIREG == 48) asm volatile("" ::: "v47");
    if (HIREG == 64) asm volatile("" ::: "v63");
    if (HIREG == 80) asm volatile("" ::: "v79");
    if (HIREG == 96) asm volatile("" ::: "v95");
    if (HIREG == 128) asm volatile("" ::: "v127");
    dyn[threadIdx.x & 63] = v;
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = dyn[(threadIdx.x + 1) & 63];
}
#define CASE(N) case N: hipLaunchKernelGGL(k_probe<N>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, out, iters); break;
extern "C" int probe_launch(float* out, int blocks, int lds, int hireg, int iters, void* stream) {
    switch (hireg) { CASE(0) CASE(24) CASE(32) CASE(48) CASE(64) CASE(80) CASE(96) CASE(128) default: return -1; }
    return (int)hipGetLastError();
}

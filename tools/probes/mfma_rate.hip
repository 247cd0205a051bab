// MFMA issue-rate probe: `waves` waves per workgroup (one workgroup per CU), each running `iters` x 64 independent
// v_mfma_f32_16x16x32_bf16 on 16 accumulators (registers only), optionally with 16 ds_read_b128 per 64 MFMAs.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int LDSREADS>
__global__ __launch_bounds__(512) void k_mfma_rate(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(0.001f * (lane + i + e)); b[i][e] = (__bf16)(0.002f * (lane - i + e)); }
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.5f;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (LDSREADS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const bf16x8*>(lds + ((it + q * 4 + r) & 31) * 1024 + lane * 16);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[q * 4 + r * 1 + 0][0] += 0.f,  // keep index math trivial for the compiler
                        acc[(q * 4 + r) & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b[2 * c + t], acc[(q * 4 + r) & 15], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" int mfma_rate_launch(float* out, int blocks, int threads, int iters, int ldsreads, void* stream) {
    if (ldsreads) hipLaunchKernelGGL(k_mfma_rate<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
    else hipLaunchKernelGGL(k_mfma_rate<0>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}

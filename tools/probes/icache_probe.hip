// Does straight-line (fully unrolled) code cost more than a rolled loop of the same work once other kernels have streamed data
// through the L2 in between?  (cold instruction fetch.)  Same 288 MFMAs + 576 FMAs per wave, one 256-thread block per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool ROLLED>
__global__ void k(float* out, int iters) {
    f32x4 acc[8];
    float v[16];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    auto body = [&](int it) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + it, b, acc[i], 0, 0, 0);
            v[2 * i] = fmaf(v[2 * i], 1.0001f, (float)it);
            v[2 * i + 1] = fmaf(v[2 * i + 1], 0.9999f, b);
        }
    };
    if (ROLLED) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) body(it);
    } else {
#pragma unroll
        for (int it = 0; it < 36; ++it) body(it);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[0] = 1;
}
__global__ void trash(float4* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float4 s = {0, 0, 0, 0};
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 t = p[i]; s.x += t.x; s.y += t.y; }
    if (s.x == 1.2345f) p[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 64);
    float4* big; size_t nb = (size_t)64 << 20; hipMalloc(&big, nb); hipMemset(big, 0, nb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rolled = 0; rolled < 2; ++rolled) for (int tr = 0; tr < 2; ++tr) {
        float tot = 0; const int reps = 200;
        for (int w = 0; w < 20; ++w) { if (rolled) hipLaunchKernelGGL(k<true>, dim3(256), dim3(256), 0, 0, d, 36); else hipLaunchKernelGGL(k<false>, dim3(256), dim3(256), 0, 0, d, 36); }
        hipDeviceSynchronize();
        for (int r = 0; r < reps; ++r) {
            if (tr) hipLaunchKernelGGL(trash, dim3(2048), dim3(256), 0, 0, big, nb / 16);
            hipEventRecord(e0);
            if (rolled) hipLaunchKernelGGL(k<true>, dim3(256), dim3(256), 0, 0, d, 36); else hipLaunchKernelGGL(k<false>, dim3(256), dim3(256), 0, 0, d, 36);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
        }
        printf("%s code, %s: %.2f us per launch (event pair included)\n", rolled ? "rolled loop (36 x 8 MFMA)" : "straight-line (288 MFMA unrolled)",
               tr ? "64 MB streamed through L2 before every launch" : "back to back", tot * 1e3 / reps);
    }
    return 0;
}

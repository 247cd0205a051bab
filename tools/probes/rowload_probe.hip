// Cost of the MFMA-fragment load pattern "16 rows x 64 B per wave instruction" vs "1 row x 1 KB" on L2-resident data:
// 256 blocks x 256 threads, every wave issues LOADS dwordx4 loads over a small matrix (row stride = pitch floats).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(const float* __restrict__ W, int pitch, int loads, float* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* p;
    if (MODE == 0) p = W + (size_t)(r + 16 * (w & 1)) * pitch + 4 * g;      // 16 rows x 64 B
    else p = W + (size_t)(w & 1) * pitch + 4 * lane;                         // 1 row x 1 KB
#pragma unroll 8
    for (int i = 0; i < loads; ++i) {
        const int off = MODE == 0 ? 16 * (i % 48) : 256 * (i % 3);
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + off);
        s += v;
    }
    if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[0] = 1;
}
int main() {
    float *W, *d;
    const int rows = 64;
    hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pitch : {768, 3072, 3072 + 32, 16384}) {
        hipMalloc(&W, (size_t)rows * pitch * 4 + 4096); hipMemset(W, 0, (size_t)rows * pitch * 4);
        for (int mode = 0; mode < 2; ++mode) {
            const int loads = 72, reps = 200;
            for (int i = 0; i < 10; ++i) { if (mode) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, W, pitch, loads, d); else hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, W, pitch, loads, d); }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) { if (mode) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, W, pitch, loads, d); else hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, W, pitch, loads, d); }
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("pitch %5d floats, %s: %.2f us per launch (72 dwordx4 loads per wave, 4 waves per CU)\n", pitch,
                   mode ? "1 row x 1 KB per instruction  " : "16 rows x 64 B per instruction", ms * 1e3 / reps);
        }
        hipFree(W);
    }
    return 0;
}

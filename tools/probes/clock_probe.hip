// Shader clock actually delivered to short MFMA kernels (MI355X DVFS): cycles (clock64) vs 100 MHz wall ticks (wall_clock64)
// around a block of f32 MFMAs -- cold single launch, back-to-back launches of a small grid, and of a full grid.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(long long* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    if (s == 12345.678f) out[2] = 1;
}
int main() {
    long long *d, h[3];
    hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 36;  // 288 MFMAs per wave, as one wave of the fused-LN products
    for (int grid : {256, 1024}) for (int threads : {256, 512}) for (int reps : {1, 20, 2000}) {
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("grid %4d x %3d thr, %4d back-to-back: %.2f us per launch; last kernel: %lld cycles for %d MFMAs (%.1f cyc/MFMA), %lld wall ticks -> %.0f MHz shader clock\n",
               grid, threads, reps, ms * 1e3 / reps, h[0], iters * 8, (double)h[0] / (iters * 8), h[1], h[0] / (h[1] / 100.0));
    }
    return 0;
}

// MFMA issue probe (round 5): shader cycles per v_mfma_f32_16x16x32_bf16 on one SIMD with the accumulator / operand pattern of
// k_scores_tile (32 accumulators = 128 registers, A[r] x B[c], hi group of 16 then lo group of 16), registers only.
//   mode 0: plain;  mode 1: s_setprio 1 around every group of 16;  mode 2: a s_barrier after every 64 (the K-step barrier);
//   mode 3: 8 ds_read_b128 between the groups (the fragment reads)
// `threads` = 256: one wave per SIMD, 512: two.  out[block] = cycles of wave 0 for iters x 64 MFMAs.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16x8 __attribute__((may_alias)) bf16x8_ma;
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_mfma_issue(long long* out, float* sink, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63;
    bf16x8 A[4], A2[4], Bh[4], Bl[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            A[i][e] = (__bf16)(seed * (float)((lane * 7 + i * 3 + e * 5) % 17 - 8));
            A2[i][e] = (__bf16)(seed * (float)((lane * 5 + i * 7 + e * 3) % 19 - 9));
            Bh[i][e] = (__bf16)(seed * (float)((lane * 3 + i * 5 + e * 7) % 23 - 11));
            Bl[i][e] = (__bf16)(seed * 0.01f * (float)((lane + i + e) % 13 - 6));
        }
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = seed * (float)(i % 31 - 15);
    __syncthreads();
    f32x4 acc[8][4];
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (MODE == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    (q ? A2 : A)[r] = *reinterpret_cast<const bf16x8_ma*>(lds + ((it * 8 + q * 4 + r) & 63) * 1024 + lane * 16);
                    (q ? Bl : Bh)[r] = *reinterpret_cast<const bf16x8_ma*>(lds + ((it * 8 + q * 4 + r + 32) & 63) * 1024 + lane * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[4 * q + r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((q ? A2 : A)[r], Bh[c], acc[4 * q + r][c], 0, 0, 0);
            if (MODE == 1) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[4 * q + r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((q ? A2 : A)[r], Bl[c], acc[4 * q + r][c], 0, 0, 0);
            if (MODE == 1) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 2) __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 4; ++c) s += acc[r][c][0] + acc[r][c][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
extern "C" int mfma_issue_launch(long long* out, float* sink, int blocks, int threads, int iters, int mode, float seed, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case 0: hipLaunchKernelGGL(k_mfma_issue<0>, dim3(blocks), dim3(threads), 0, st, out, sink, iters, seed); break;
        case 1: hipLaunchKernelGGL(k_mfma_issue<1>, dim3(blocks), dim3(threads), 0, st, out, sink, iters, seed); break;
        case 2: hipLaunchKernelGGL(k_mfma_issue<2>, dim3(blocks), dim3(threads), 0, st, out, sink, iters, seed); break;
        default: hipLaunchKernelGGL(k_mfma_issue<3>, dim3(blocks), dim3(threads), 0, st, out, sink, iters, seed); break;
    }
    return (int)hipGetLastError();
}

// Shared device helpers for the gfx950 kernels of libvlsa_hip.so.  CDNA4 only: 64-lane wavefronts,
// MFMA 16x16x32 bf16, LDS transpose reads.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vlsa_hip.h"

// Experiment switches.  The shipped library reads NO process environment (no global state): every A/B hook of the
// measurement tools goes through VLSA_ENV, which is getenv only in a -DVLSA_EXPERIMENT build (tools/*: VLSA_EXTRA_HIPCC_FLAGS) and a
// null pointer -- the hook's name does not even reach the binary -- otherwise.  tests/test_abi_cpu.py greps the default build.
#ifdef VLSA_EXPERIMENT
#include <cstdlib>
#define VLSA_ENV(name) getenv(name)
#else
#define VLSA_ENV(name) (static_cast<const char*>(nullptr))
#endif

// wave priority of the short tail / preparation kernels that co-run with a persistent streaming kernel
#ifndef VLSA_TAIL_PRIO
#define VLSA_TAIL_PRIO 3
#endif

namespace vlsa {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNormEps = 1e-12f;  // F.normalize eps (model/deepmil.py:187,189)
constexpr int kPStride = 16;        // stride of the per-query m/l arrays

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// Sum / max over the 4 lanes {i, i+16, i+32, i+48} (the four 16-lane rows of a wavefront).
__device__ __forceinline__ float quad_rows_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float quad_rows_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Counter-based dropout mask of the fused attention-score kernels (forward and backward recompute the same bits):
// keep(seed, row, unit) = mix32(seed ^ row * 0x9E3779B1 ^ unit * 0x85EBCA6B) >= thr, thr = p * 2^32 (murmur3 finaliser).
// `unit` = hidden unit + 256 * branch; `row` = the row's index in the launch's score array.
__host__ __device__ __forceinline__ unsigned int dropout_bits(unsigned int seed, unsigned int row, unsigned int unit) {
    unsigned int h = seed ^ (row * 0x9E3779B1u) ^ (unit * 0x85EBCA6Bu);
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__device__ __forceinline__ float load_as_float(const float* p) { return *p; }
__device__ __forceinline__ float load_as_float(const __bf16* p) { return (float)*p; }

// Block-wide sum for 256-thread blocks; `red` is >= 4 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember for which devices of this process it was done
// (normally one process drives one GPU, but a host may also walk over several).
struct DeviceOnce {
    bool done[64] = {};
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

// Opaque prepared-query block layout (see vlsa_prepare_queries).
struct QPrepLayout {
    size_t qeff, qsplit, qhat, qnorm, total;
    __host__ __device__ explicit QPrepLayout(int D) {
        qeff = 0;
        qsplit = qeff + (size_t)16 * D * 4;
        qhat = qsplit + (size_t)3 * 16 * D * 2;
        qnorm = qhat + (size_t)17 * D * 4;
        total = qnorm + 32 * 4;
    }
};

}  // namespace vlsa

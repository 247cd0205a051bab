// Helpers shared by the register-staged MFMA kernels (forward: vlfan_partial.hip, backward: vlfan_backward.hip).
// See the layout notes at the top of k_vlfan_partial_mfma.
#pragma once
#include "vlsa_common.h"

namespace vlsa {

constexpr int kTileRows = 32;
constexpr int kSliceBytes = kTileRows * 256;            // one wave's bf16 slice image
constexpr int kExchWave = 2 * 64 * 16 + 32 * 4;         // per wave: S partials (2 x f32x4 per lane) + 32 row sumsq
constexpr int kExchParity = 4 * kExchWave;
constexpr float kRescaleThreshold = 16.0f;              // log2 units: weights stay <= 2^16

template <bool F32>
constexpr int mfma_lds_bytes() {
    return 4 * kSliceBytes * (F32 ? 2 : 1) + 2 * kExchParity;
}

__device__ __forceinline__ int swz(int row, int byte_off) { return row * 256 + (byte_off ^ ((row & 7) << 5)); }

__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b, float c) {
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c, false);
    c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c, false);
    return c;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// LDS images are written as raw 16-B words and read back as bf16 fragments: every such access goes through
// may_alias types so type-based alias analysis can never reorder a fragment read above the staging write.
typedef u32x4 __attribute__((may_alias)) u32x4_ma;
typedef bf16x8 __attribute__((may_alias)) bf16x8_ma;
typedef bf16x4 __attribute__((may_alias)) bf16x4_ma;
typedef f32x4 __attribute__((may_alias)) f32x4_ma;

// Staging registers for one 32-row tile of this wave's 128-column slice.
//   bf16: 8 x 16 B per lane; load i covers rows 4i..4i+3, lane -> (row 4i + g, 16-B chunk i16)
//   fp32: 16 x 16 B per lane; load i covers rows 2i, 2i+1, lane -> (row 2i + (l >> 5), 4 floats)
// Rows past the shard end are clamped to its last row (valid memory) and masked later.
template <typename XT>
struct StageN { static constexpr int value = sizeof(XT) == 4 ? 16 : 8; };

__device__ __forceinline__ void stage_load(u32x4 (&v)[8], const __bf16* __restrict__ X, int64_t ldx, int64_t r0,
                                           int64_t rlast, int w, int lane) {
    const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int64_t r = r0 + 4 * i + g;
        r = r < rlast ? r : rlast;
#ifdef VLSA_NO_NT
        v[i] = *reinterpret_cast<const u32x4*>(X + r * ldx + w * 128 + i16 * 8);
#else
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(X + r * ldx + w * 128 + i16 * 8));
#endif
    }
}
__device__ __forceinline__ void stage_store(const u32x4 (&v)[8], unsigned char* xs, int lane) {
    const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4_ma*>(xs + swz(4 * i + g, i16 * 16)) = v[i];
}
__device__ __forceinline__ void stage_load(u32x4 (&v)[16], const float* __restrict__ X, int64_t ldx, int64_t r0,
                                           int64_t rlast, int w, int lane) {
    const int hh = lane >> 5, c4 = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int64_t r = r0 + 2 * i + hh;
        r = r < rlast ? r : rlast;
        v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(X + r * ldx + w * 128 + c4 * 4));
    }
}
// split each fp32 into hi + lo bf16 and write the two slice images (hi at xs, lo at xs + kSliceBytes)
__device__ __forceinline__ void stage_store(const u32x4 (&v)[16], unsigned char* xs, int lane) {
    const int hh = lane >> 5, c4 = lane & 31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int bits = v[i][e];  // (bit_cast straight from the vector-element glvalue miscompiles)
            const float f = __uint_as_float(bits);
            hi[e] = (__bf16)f;
            lo[e] = (__bf16)(f - (float)hi[e]);
        }
        const int off = swz(2 * i + hh, c4 * 8);
        *reinterpret_cast<bf16x4_ma*>(xs + off) = hi;
        *reinterpret_cast<bf16x4_ma*>(xs + kSliceBytes + off) = lo;
    }
}


// Row range of workgroup b out of G for N rows, balanced at 16-row granularity.
__device__ __forceinline__ void block_rows(int64_t N, int b, int G, int64_t& rbeg, int64_t& rend) {
    const int64_t units = (N + 15) >> 4;
    const int64_t uq = units / G, ur = units % G;
    const int64_t ubeg = b * uq + (b < ur ? b : ur);
    rbeg = ubeg << 4;
    rend = (ubeg + uq + (b < ur ? 1 : 0)) << 4;
    if (rend > N) rend = N;
    if (rbeg > N) rbeg = N;
}

}  // namespace vlsa

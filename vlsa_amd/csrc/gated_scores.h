// Shared between gated_scores.hip (the score kernels of rounds 1-4) and gated_scores_tile.hip (round 5: both operands through
// LDS-DMA): the packed-weight block layout, the bag table of a batched launch, the activation helpers.
#pragma once
#include "vlsa_common.h"

namespace vlsa {

namespace gs {
constexpr int kHid = 256;
constexpr int kD = 512;
constexpr int kSteps = 16;                        // K steps of 32
constexpr int kHalves = 2;                        // a workgroup covers kHid / kHalves hidden units (of both branches)
}  // namespace gs

struct GatedPrepLayout {
    size_t wpack, ba, bg, w2, c, wtile, total;
    __host__ __device__ explicit GatedPrepLayout(int gated) {
        wpack = 0;
        ba = wpack + (size_t)gs::kHalves * 8 * gs::kSteps * (gated ? 4 : 2) * 1024;
        bg = ba + gs::kHid * 4;
        w2 = bg + gs::kHid * 4;
        c = w2 + gs::kHid * 4;
        // round 5: the LDS image of k_scores_tile (gated_scores_tile.hip), [column half][K step][term][256 columns][64 B]
        wtile = c + 16;
        total = wtile + (size_t)(gated ? 2 : 1) * gs::kSteps * 2 * 256 * 64;
    }
};

// tanh(x) * sigmoid(y) = (1 - u) / ((1 + u)(1 + v)),  u = e^{-2x}, v = e^{-y}: two v_exp_f32 and ONE v_rcp_f32 (1 ulp) per
// value.  The activations are a real cost here (N x 512 of them per bag at quarter rate; an IEEE division would add ~10
// VALU instructions each).  The arguments au = -2 log2(e) x and av = -log2(e) y come straight out of the accumulators (weights
// and biases are pre-scaled); they are clamped from above only: 2^43 * 2^57 keeps (1 + u)(1 + v) finite, where the result
// is saturated in fp32 anyway, and exp2 of a very negative argument is simply 0.
__device__ __forceinline__ float gate_act(float au, float av) {
    const float u = fast_exp2(fminf(au, 43.f)), v = fast_exp2(fminf(av, 57.f));
    return (1.f - u) * __builtin_amdgcn_rcpf((1.f + u) * (1.f + v));
}
__device__ __forceinline__ float tanh_act(float au) {
    const float u = fast_exp2(fminf(au, 43.f));
    return (1.f - u) * __builtin_amdgcn_rcpf(1.f + u);
}
// The same with the clamp as ONE v_med3_f32 (fminf costs a canonicalising v_max_f32 in front of the v_min_f32: accumulators straight
// out of an MFMA are not known to be quiet); a NaN argument comes out as the lower bound's activation instead of NaN.
__device__ __forceinline__ float gate_act3(float au, float av) {
    const float u = fast_exp2(__builtin_amdgcn_fmed3f(au, -1.0e30f, 43.f)), v = fast_exp2(__builtin_amdgcn_fmed3f(av, -1.0e30f, 57.f));
    return (1.f - u) * __builtin_amdgcn_rcpf((1.f + u) * (1.f + v));
}
__device__ __forceinline__ float tanh_act3(float au) {
    const float u = fast_exp2(__builtin_amdgcn_fmed3f(au, -1.0e30f, 43.f));
    return (1.f - u) * __builtin_amdgcn_rcpf(1.f + u);
}
// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane: four full-rate VALU adds
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}

struct GsBag {
    const void* X;
    long long N, ldx;
};
struct GsBatch {
    const GsBag* bags;
    const int* tile_start;      // [B + 1], tile_start[0] = 0
    const long long* a_off;     // [B] offset (floats) of bag b's scores in a_out
    int B;                      // <= 64
    // training-mode dropout of Gated_Attention_Pooling (nn.Dropout behind tanh and behind sigmoid, model/layers.py:94,99):
    // drop_thr = p * 2^32 (0: off), drop_scale = 1 / (1 - p); see dropout_bits()
    unsigned int drop_thr, drop_seed;
    float drop_scale;
    unsigned int row_base;      // first row of this launch inside the bag's score array (a bag may be covered by two launches)
};


// gated_scores_tile.hip (round 5): 128-row x 256-column tiles, both operands through LDS-DMA; bf16 bags
int gs_tile_prepare(const float* Wa, const float* Wg, int gated, unsigned char* prep, hipStream_t st);
int gs_tile_pool_tiles(long long N);
int gs_tile_launch(const void* X, long long N, long long ldx, const unsigned char* prep, int gated, float* a, int n_tiles,
                   int rows_per_tile, const GsBatch& bt, float* ws, float* pooled, hipStream_t st);
}  // namespace vlsa

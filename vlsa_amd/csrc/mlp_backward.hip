// Backward of the two N-sized "linear layer + row-wise non-linearity" pieces of the path:
//
//   (Gated_)Attention_Pooling scores (model/layers.py:103-122,137-153):  a_n = w2 . (tanh(Wa x_n + ba) [* sigmoid(Wg x_n + bg)]) + c
//        given da_n = dL/da_n:  dWa, dba, dWg, dbg, dw2, dc
//   Feat_Projecter (model/layers.py:65-82):  y_n = LayerNorm(W x_n + b) * gamma + beta
//        given dy_n = dL/dy_n:  dW, db, dgamma, dbeta
//
// PyTorch autograd (and round 2 here) does this with the [N, 256] / [N, 512] hidden activations saved in memory, two library
// GEMMs and ~10 elementwise kernels.  Here ONE kernel per layer, no N-sized intermediate: per tile of rows the hidden
// pre-activations are RECOMPUTED on the matrix pipe (same packed bf16 hi + lo weight fragments as the forward kernels), turned
// into the pre-activation gradient dH in registers, and dW += dH^T X is a second MFMA contraction over the rows of the tile.
//
//   * wave = 16 hidden units (one branch) x all 512 input columns: dW accumulator = 32 column tiles = 128 registers per lane.
//     The recomputed H tile comes out of the MFMA as C[row][hidden] with the hidden unit on the lane index and 4 rows per
//     register group -- which IS the A-fragment layout of the second contraction (M = hidden, K = rows) once k-slot (g, j) is
//     mapped to tile row 16 (j >> 2) + 4 g + (j & 3): dH never leaves registers (the trick of the streaming kernels' p X
//     contraction).  Its B fragments X[rows][16 columns] are hardware-transposed LDS reads (ds_read_b64_tr_b16) of the same X
//     tile the recomputation read row-wise.
//   * workgroup = 8 waves = one SLICE of the hidden units: gated 4 tiles x 2 branches (the pair exchanges tanh / sigmoid
//     through LDS: dHa needs s, dHg needs t), tanh-only 8 tiles, projecter 8 output tiles.  Slices x row chunks = 256
//     workgroups (one per CU); a chunk's slices sit on the same XCD (blockIdx -> (slice, chunk) below) so that the X tile they
//     all read comes out of that XCD's L2 after the first miss.
//   * persistent over rows: a workgroup walks its range of row tiles (of all bags of the launch: the weights are shared by
//     the bags of an optimizer step, so dW sums over bags), the next tile's global loads in flight during the dW contraction;
//     one partial per workgroup at the end, reduced in a fixed order by k_mb_reduce (deterministic, no atomics).
//   * X: bf16 rows are consumed exactly (64-row tiles); fp32 rows (the reference's feature format; the projected bag) are
//     split into bf16 hi + lo images on the fly (32-row tiles, 3 product terms).  dH is split hi + lo as well.
// MFMA-bound: per row 2 x (512 x H) recompute + 2 x (512 x H) for dW, both as 2 bf16 terms (H = 512 gated, 256 tanh, 512
// projecter): 2.1 MFLOP per row for the gated scores = 4 x the algorithmic forward.
#include "vlsa_common.h"

namespace vlsa {

typedef bf16x8 __attribute__((may_alias)) bf16x8_mb;
typedef bf16x4 __attribute__((may_alias)) bf16x4_mb;
typedef f32x4 __attribute__((may_alias)) f32x4_mb;
typedef float __attribute__((may_alias)) float_mb;
typedef unsigned int u32x4_mb_t __attribute__((ext_vector_type(4)));
typedef u32x4_mb_t __attribute__((may_alias)) u32x4_mb;

namespace mb {
constexpr int kD = 512;
constexpr int kTileBytes = 65536;            // bf16: 64 rows x 1 KiB; fp32: hi + lo bf16 images of 32 rows
constexpr int kExchOff = kTileBytes;         // gated: tanh / sigmoid exchange (32 KiB); projecter: dy slice [64][132] fp32
constexpr int kExchBytes = 64 * 132 * 4;     // 33,792 B
constexpr int kLds = kExchOff + kExchBytes;
constexpr int kDyPitch = 132;
enum { kTanh = 0, kGated = 1, kLN = 2 };
}  // namespace mb

struct MbBag {
    const void* X;
    long long N, ldx;
};

struct MbArgs {
    const MbBag* bags;          // [B] rows of the layer input
    const MbBag* dy;            // projecter: [B] upstream gradient rows (fp32 [N, 512]); else null
    const int* tile_start;      // [B + 1] first row tile of every bag (tiles of mb rows: 64 bf16 / 32 fp32)
    const long long* row_off;   // [B] offset of bag b's rows in `rowvec`
    const float* rowvec;        // scores: da [sum N]; projecter: stats [sum N][4] = (mean, rstd, c1, c2)
    const unsigned char* wpack; // fragment-packed weights of the forward kernel
    const float* bias_a;        // scores: pre-scaled ba (exp2 domain); projecter: b
    const float* bias_g;        // gated: pre-scaled bg
    const float* vec;           // scores: w2 [256]; projecter: gamma [512]
    float* part_w;              // [C][U][512] partial dW, U = rows of the stacked weight matrix
    float* part_v;              // [C][3][512] partial vectors
    int B, n_tiles, C;
    // gated scores: the training-mode dropout of the forward (vlsa_gated_scores_train), re-evaluated here; drop_thr = 0: off
    unsigned int drop_thr, drop_seed;
    float drop_scale;
};

__device__ __forceinline__ int mb_swz(int row, int byte_off) { return row * 256 + (byte_off ^ ((row & 7) << 5)); }

// MODE: mb::kTanh / kGated / kLN.  XF32: fp32 input rows.
template <int MODE, bool XF32>
__global__ __launch_bounds__(512) void k_mlp_backward(const MbArgs a) {
    using namespace mb;
    constexpr int RT = XF32 ? 2 : 4;             // 16-row tiles per step
    constexpr int ROWS = 16 * RT;
    constexpr int QB = ROWS * 256;               // one column quarter of one bf16 image
    constexpr int IMG = 4 * QB;                  // one bf16 image of the tile (fp32: hi image, lo image behind it)
    constexpr int NSL = MODE == kTanh ? 2 : 4;   // hidden slices
    constexpr int U = MODE == kTanh ? 256 : 512; // rows of the stacked dW
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    // blockIdx -> (slice, chunk): consecutive workgroups go round-robin over the 8 XCDs, so chunk c lives on XCD c % 8 with all
    // of its slices
    const int j = blockIdx.x;
    const int c = (j & 7) + 8 * (j / (8 * NSL));
    const int sl = (j >> 3) % NSL;
    if (c >= a.C) return;
    const int br = MODE == kGated ? (w >> 2) : 0;
    const int unit = MODE == kGated ? 4 * sl + (w & 3) : 8 * sl + w;     // 16-wide tile of hidden units / outputs
    const int urow = (MODE == kGated ? br * 256 : 0) + 16 * unit;        // first row of this wave's block in the stacked dW
    // fragment address of (k step, term): scores: GatedPrepLayout.wpack; projecter: FeatProjLayout.wpack
    const unsigned char* wp;
    int ks_stride;
    if constexpr (MODE == kLN) {
        wp = a.wpack + ((size_t)(unit >> 2) * 16 * 8 + (unit & 3) * 2) * 1024 + lane * 16;
        ks_stride = 8 * 1024;
    } else {
        constexpr int NF = MODE == kGated ? 4 : 2;
        wp = a.wpack + ((size_t)unit * 16 * NF + br * 2) * 1024 + lane * 16;
        ks_stride = NF * 1024;
    }
    const float hb = (MODE == kGated && br == 1) ? a.bias_g[16 * unit + i16] : a.bias_a[16 * unit + i16];
    const float hv = a.vec[16 * unit + i16];      // w2 of the hidden unit / gamma of the output

    const int t0 = (int)((long long)a.n_tiles * c / a.C), t1 = (int)((long long)a.n_tiles * (c + 1) / a.C);

    f32x4 accw[32];
#pragma unroll
    for (int ct = 0; ct < 32; ++ct) accw[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;          // scores: db, dw2, dc;  projecter: db, dgamma, dbeta

    // ---- tile lookup + register staging of a tile's rows -------------------------------------------------------------------
    struct Tile { const unsigned char* x; const float* dy; const float* rv; long long ldx, lddy; int nrows; unsigned int rid; };
    auto find = [&](int t) -> Tile {
        const int ts = lane < a.B ? a.tile_start[lane] : 0x7fffffff;
        const int b = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ts <= t)) - 1;
        const MbBag bag = a.bags[b];
        const long long row0 = (long long)(t - a.tile_start[b]) * ROWS;
        Tile r;
        r.ldx = bag.ldx;
        r.rid = (unsigned int)row0;          // row index inside the bag (the dropout counter of the forward)
        r.nrows = (int)((bag.N - row0) < ROWS ? (bag.N - row0) : ROWS);
        r.x = static_cast<const unsigned char*>(bag.X) + row0 * bag.ldx * (XF32 ? 4 : 2);
        r.rv = a.rowvec + (a.row_off[b] + row0) * (MODE == kLN ? 4 : 1);
        r.dy = nullptr;
        r.lddy = 0;
        if constexpr (MODE == kLN) {
            const MbBag d = a.dy[b];
            r.dy = static_cast<const float*>(d.X) + row0 * d.ldx;
            r.lddy = d.ldx;
        }
        return r;
    };
    u32x4_mb_t st[8];
    u32x4_mb_t sd[MODE == kLN ? (XF32 ? 2 : 4) : 1];
    auto stage_load = [&](const Tile& T) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = tid + 512 * k;
            const int row = XF32 ? (p >> 7) : (p >> 6), ch = XF32 ? (p & 127) : (p & 63);
            st[k] = u32x4_mb_t{0u, 0u, 0u, 0u};
            if (row < T.nrows) st[k] = *reinterpret_cast<const u32x4_mb_t*>(T.x + ((size_t)row * T.ldx * (XF32 ? 4 : 2)) + ch * 16);
        }
        if constexpr (MODE == kLN) {          // this slice's 128 columns of dy: [ROWS][128] fp32
#pragma unroll
            for (int k = 0; k < (XF32 ? 2 : 4); ++k) {
                const int p = tid + 512 * k;
                const int row = p >> 5, ch = p & 31;
                sd[k] = u32x4_mb_t{0u, 0u, 0u, 0u};
                if (row < T.nrows) sd[k] = *reinterpret_cast<const u32x4_mb_t*>(T.dy + (size_t)row * T.lddy + 128 * sl + 4 * ch);
            }
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = tid + 512 * k;
            if constexpr (XF32) {
                const int row = p >> 7, c4 = p & 127;
                bf16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned int bits = st[k][e];
                    const float f = __uint_as_float(bits);
                    hi[e] = (__bf16)f;
                    lo[e] = (__bf16)(f - (float)hi[e]);
                }
                const int off = (c4 >> 5) * QB + mb_swz(row, (c4 & 31) * 8);
                *reinterpret_cast<bf16x4_mb*>(smem + off) = hi;
                *reinterpret_cast<bf16x4_mb*>(smem + IMG + off) = lo;
            } else {
                const int row = p >> 6, ch = p & 63;
                *reinterpret_cast<u32x4_mb*>(smem + (ch >> 4) * QB + mb_swz(row, (ch & 15) * 16)) = st[k];
            }
        }
        if constexpr (MODE == kLN) {
#pragma unroll
            for (int k = 0; k < (XF32 ? 2 : 4); ++k) {
                const int p = tid + 512 * k;
                const int row = p >> 5, ch = p & 31;
                *reinterpret_cast<u32x4_mb*>(smem + kExchOff + (row * kDyPitch + 4 * ch) * 4) = sd[k];
            }
        }
    };

    if (t0 < t1) {
        for (int t = t0; t < t1; ++t) {
            const Tile cur = find(t);
            stage_load(cur);
            stage_store();
            __syncthreads();
            // ---- phase 1: recompute the pre-activations of this wave's 16 units for the tile's rows ----------------------
            f32x4 acch[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acch[rt] = f32x4{hb, hb, hb, hb};
            {
                bf16x8 B0[2], B1[2];
                auto load_b = [&](int ks, bf16x8 (&dst)[2]) {
                    dst[0] = *reinterpret_cast<const bf16x8*>(wp + (size_t)ks * ks_stride);
                    dst[1] = *reinterpret_cast<const bf16x8*>(wp + (size_t)ks * ks_stride + 1024);
                };
                auto kstep = [&](int ks, const bf16x8 (&Bc)[2]) {
                    const int qoff = (ks >> 2) * QB, boff = (ks & 3) * 64 + g * 16;
                    bf16x8 A[RT], AL[XF32 ? RT : 1];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        A[rt] = *reinterpret_cast<const bf16x8_mb*>(smem + qoff + mb_swz(16 * rt + i16, boff));
                        if constexpr (XF32) AL[rt] = *reinterpret_cast<const bf16x8_mb*>(smem + IMG + qoff + mb_swz(16 * rt + i16, boff));
                    }
#pragma unroll
                    for (int term = 0; term < 2; ++term)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acch[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[rt], Bc[term], acch[rt], 0, 0, 0);
                    if constexpr (XF32) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acch[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL[rt], Bc[0], acch[rt], 0, 0, 0);
                    }
                };
                // (a three-deep ring measured no faster -- 146 vs 144 us per 50k gated bag -- and spilled in the projecter variant)
                load_b(0, B0);
#pragma unroll 1
                for (int ks = 0; ks < 16; ks += 2) {      // rolled on purpose: unrolled, the scheduler hoists all 32 weight loads
                    load_b(ks + 1, B1);
                    kstep(ks, B0);
                    if (ks + 2 < 16) load_b(ks + 2, B0);
                    kstep(ks + 1, B1);
                }
            }
            // per-row inputs of this lane's rows (C layout: rows 16 rt + 4 g + r)
            f32x4 rv[RT];             // scores: da of the 4 rows
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (MODE != kLN) {
                    const int row = 16 * rt + 4 * g;
                    if (row + 3 < cur.nrows) {
                        rv[rt] = f32x4{cur.rv[row], cur.rv[row + 1], cur.rv[row + 2], cur.rv[row + 3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) rv[rt][r] = (row + r < cur.nrows) ? cur.rv[row + r] : 0.f;
                    }
                }
            }
            // ---- phase 2a (scores): activations in place of the accumulators; the gated pair exchanges them through LDS --------
            // accumulators are the exp2 arguments: branch a: u = e^{-2x}, tanh = (1 - u) / (1 + u); branch g: v = e^{-y},
            // sigmoid = 1 / (1 + v)   (weights and biases pre-scaled by k_prepare_gated_weights)
            unsigned char* ex = smem + kExchOff + (w & 3) * 8192;
            if constexpr (MODE != kLN) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (br == 0) {
                            const float u = fast_exp2(fminf(acch[rt][r], 43.f));
                            acch[rt][r] = (1.f - u) * __builtin_amdgcn_rcpf(1.f + u);
                        } else {
                            const float v = fast_exp2(fminf(acch[rt][r], 57.f));
                            acch[rt][r] = __builtin_amdgcn_rcpf(1.f + v);
                        }
                    }
                if constexpr (MODE == kGated) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<f32x4_mb*>(ex + br * 4096 + (rt * 64 + lane) * 16) = acch[rt];
                    __syncthreads();
                }
            }
            // ---- per 32 rows: phase 2b: the pre-activation gradient dH of the lane's (rows, unit) entries; phase 3:
            // dW[unit][:] += dH^T X over those rows ---------------------------------------------------------------------------------
#pragma unroll
            for (int q32 = 0; q32 < RT / 2; ++q32) {
                bf16x8 ahi, alo;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int rt = 2 * q32 + h;
                    f32x4 other = f32x4{1.f, 1.f, 1.f, 1.f};
                    if constexpr (MODE == kGated) other = *reinterpret_cast<const f32x4_mb*>(ex + (br ^ 1) * 4096 + (rt * 64 + lane) * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float d;
                        if constexpr (MODE == kLN) {
                            const float_mb* dys = reinterpret_cast<const float_mb*>(smem + kExchOff);
                            const int row = 16 * rt + 4 * g + r;
                            const float dyv = dys[row * kDyPitch + 16 * w + i16];     // 0 for rows past the bag's end
                            f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};                       // (mean, rstd, c1, c2) of the row
                            if (row < cur.nrows) s4 = *reinterpret_cast<const f32x4*>(cur.rv + 4 * row);
                            const float zh = (acch[rt][r] - s4[0]) * s4[1];
                            d = s4[1] * (dyv * hv - s4[2] - zh * s4[3]);
                            v0 += d;
                            v1 += dyv * zh;
                            v2 += dyv;
                        } else {
                            const float m = acch[rt][r], dav = rv[rt][r];
                            const float dfac = br == 0 ? (1.f - m * m) : m * (1.f - m);     // tanh' / sigmoid'
                            float keep = 1.f;
                            if (MODE == kGated && a.drop_thr != 0u) {      // uniform: the forward's dropout masks of both branches
                                const unsigned int row = cur.rid + 16 * rt + 4 * g + r, hu = 16 * unit + i16;
                                const bool ka = dropout_bits(a.drop_seed, row, hu) >= a.drop_thr;
                                const bool kg = dropout_bits(a.drop_seed, row, hu + 256u) >= a.drop_thr;
                                keep = (ka && kg) ? a.drop_scale * a.drop_scale : 0.f;
                            }
                            d = dav * hv * dfac * other[r] * keep;
                            v0 += d;
                            if (br == 0) {
                                v1 += dav * m * other[r] * keep;
                                v2 += dav;
                            }
                        }
                        const __bf16 hi = (__bf16)d;
                        ahi[4 * h + r] = hi;
                        alo[4 * h + r] = (__bf16)(d - (float)hi);
                    }
                }
                const int rr = 32 * q32 + 4 * g + (i16 >> 2);
#pragma unroll
                for (int ct = 0; ct < 32; ++ct) {
                    const int off = (ct >> 3) * QB, c_off = (ct & 7) * 32 + (i16 & 3) * 8;
                    const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + off + mb_swz(rr, c_off)));
                    const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + off + mb_swz(16 + rr, c_off)));
                    const bf16x8 bh = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    accw[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bh, accw[ct], 0, 0, 0);
                    accw[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bh, accw[ct], 0, 0, 0);
                    if constexpr (XF32) {
                        const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + IMG + off + mb_swz(rr, c_off)));
                        const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(smem + IMG + off + mb_swz(16 + rr, c_off)));
                        const bf16x8 bl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                        accw[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bl, accw[ct], 0, 0, 0);
                    }
                }
            }
            __syncthreads();          // every wave is done with the tile (and the exchange area)
        }
    }

    // ---- the workgroup's partial: dW block of this wave, vectors ---------------------------------------------------------------
    float* pw = a.part_w + ((size_t)c * U + urow) * kD;
#pragma unroll
    for (int ct = 0; ct < 32; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[(size_t)(4 * g + r) * kD + 16 * ct + i16] = accw[ct][r];
    v0 = quad_rows_sum(v0);
    v1 = quad_rows_sum(v1);
    v2 = quad_rows_sum(v2);
    float* pv = a.part_v + (size_t)c * 3 * kD;
    if (g == 0) {
        if constexpr (MODE == kLN) {
            pv[16 * unit + i16] = v0;
            pv[kD + 16 * unit + i16] = v1;
            pv[2 * kD + 16 * unit + i16] = v2;
        } else {
            pv[br * 256 + 16 * unit + i16] = v0;                       // db: [ba | bg]
            if (br == 0) pv[kD + 16 * unit + i16] = v1;                // dw2
            if (br == 0 && unit == 0 && i16 == 0) pv[2 * kD] = v2;     // dc = sum of da (every lane of the wave holds it)
        }
    }
}

// out[i] = sum_c part[c][i] in a fixed order: 256 threads = 32 float4 columns x 8 partial groups (thread group q sums the
// partials c = q, q + 8, ...; the 8 group sums are added through LDS in the order of q) -- 8 loads in flight per column
// instead of one dependent chain over all C partials
__global__ __launch_bounds__(256) void k_mb_reduce(const float* __restrict__ part, int C, long long total, float* __restrict__ out) {
    __shared__ f32x4 red[8][32];
    const int col = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long long i = ((long long)blockIdx.x * 32 + col) * 4;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < total)
        for (int c = q; c < C; c += 8) s += *reinterpret_cast<const f32x4*>(part + (size_t)c * total + i);
    red[q][col] = s;
    __syncthreads();
    if (q == 0 && i < total) {
#pragma unroll
        for (int k = 1; k < 8; ++k) s += red[k][col];
        *reinterpret_cast<f32x4*>(out + i) = s;
    }
}

// Per-row LayerNorm-backward scalars of the projecter (one wave per row): c1 = mean_o(dy gamma), c2 = mean_o(dy (y - beta))
// [= mean_o(dy gamma zhat): no division by gamma], written behind the forward's (mean, rstd) into stats[row][2..3].
__global__ __launch_bounds__(256) void k_ln_bwd_rowstats(const float* __restrict__ dy, long long lddy, const float* __restrict__ y,
                                                          long long ldy, long long N, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int col = 4 * lane + 256 * k;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + row * lddy + col);
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + row * ldy + col);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + col);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1 += d[e] * gm[e];
            s2 += d[e] * (v[e] - bt[e]);
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        stats[row * 4 + 2] = s1 * (1.f / 512.f);
        stats[row * 4 + 3] = s2 * (1.f / 512.f);
    }
}

}  // namespace vlsa

using namespace vlsa;

namespace {
struct GatedPrepOffsets {      // mirrors GatedPrepLayout (gated_scores.hip)
    size_t wpack, ba, bg, w2, c;
    explicit GatedPrepOffsets(int gated) {
        wpack = 0;
        ba = wpack + (size_t)2 * 8 * 16 * (gated ? 4 : 2) * 1024;
        bg = ba + 256 * 4;
        w2 = bg + 256 * 4;
        c = w2 + 256 * 4;
    }
};
struct FeatProjOffsets {       // mirrors FeatProjLayout (feat_proj.hip)
    size_t wpack = 0, bias = (size_t)8 * 16 * 8 * 1024, gamma = bias + 512 * 4, beta = gamma + 512 * 4;
};
int chunks_for(int n_tiles, int nsl) {
    int C = 256 / nsl;
    if (n_tiles < C) C = n_tiles;
    return C < 1 ? 1 : C;
}
template <int MODE, bool XF32>
int launch(const MbArgs& a, int nsl, hipStream_t st) {
    static DeviceOnce once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)k_mlp_backward<MODE, XF32>, hipFuncAttributeMaxDynamicSharedMemorySize, mb::kLds);
    const unsigned int grid = 8u * nsl * ((a.C + 7) / 8);
    hipLaunchKernelGGL((k_mlp_backward<MODE, XF32>), dim3(grid), dim3(512), mb::kLds, st, a);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}
}  // namespace

extern "C" int vlsa_mlp_bwd_tile_rows(int x_dtype) { return x_dtype == VLSA_DT_F32 ? 32 : 64; }

// workspace (partials) of one backward launch over n_tiles row tiles
extern "C" size_t vlsa_mlp_bwd_workspace_bytes(int mode, int n_tiles) {
    const int nsl = mode == mb::kTanh ? 2 : 4, U = mode == mb::kTanh ? 256 : 512;
    return (size_t)chunks_for(n_tiles < 1 ? 1 : n_tiles, nsl) * (U + 3) * 512 * sizeof(float);
}

// Backward of the (gated) attention scores of B bags.  bag_desc: device table of vlsa_bag_desc {X, N, ldx}; tile_start [B + 1]
// (device, int32) in tiles of vlsa_mlp_bwd_tile_rows(x_dtype) rows; da: dL/da of all bags' rows, bag b at da + a_off[b] (device
// int64 offsets); prep: the block of vlsa_prepare_gated_weights.  Outputs: dW [gated ? 2 : 1][256][512] (dWa, dWg),
// dvec [3][512]: row 0 = (dba [256] | dbg [256]), row 1 = dw2 [256], dvec[2][0] = dc.  drop_p / seed: the dropout the forward
// ran with (vlsa_gated_scores_train; 0: none).
extern "C" int vlsa_attn_scores_backward(const void* bag_desc, int B, int x_dtype, int D, const void* prep, int gated,
                                         const int* tile_start, int n_tiles, const float* da, const int64_t* a_off, void* ws,
                                         float* dW, float* dvec, float drop_p, unsigned int seed, void* stream) {
    if (!bag_desc || !prep || !tile_start || !da || !a_off || !ws || !dW || !dvec || B < 1 || B > 64 || n_tiles < 1) return VLSA_EINVAL;
    if (D != mb::kD || (x_dtype != VLSA_DT_BF16 && x_dtype != VLSA_DT_F32)) return VLSA_EUNSUPPORTED;
    const int mode = gated ? mb::kGated : mb::kTanh, nsl = gated ? 4 : 2, U = gated ? 512 : 256;
    const GatedPrepOffsets L(gated ? 1 : 0);
    const unsigned char* pp = static_cast<const unsigned char*>(prep);
    MbArgs a{};
    a.bags = static_cast<const MbBag*>(bag_desc);
    a.tile_start = tile_start;
    a.row_off = reinterpret_cast<const long long*>(a_off);
    a.rowvec = da;
    a.wpack = pp + L.wpack;
    a.bias_a = reinterpret_cast<const float*>(pp + L.ba);
    a.bias_g = reinterpret_cast<const float*>(pp + L.bg);
    a.vec = reinterpret_cast<const float*>(pp + L.w2);
    a.B = B;
    a.n_tiles = n_tiles;
    a.C = chunks_for(n_tiles, nsl);
    a.part_w = static_cast<float*>(ws);
    a.part_v = a.part_w + (size_t)a.C * U * 512;
    a.drop_thr = 0u;
    a.drop_scale = 1.f;
    if (gated && drop_p > 0.f) {     // same conversion as vlsa_gated_scores_train
        if (!(drop_p < 1.f)) return VLSA_EINVAL;
        a.drop_thr = (unsigned int)((double)drop_p * 4294967296.0);
        if (a.drop_thr == 0u) a.drop_thr = 1u;
        a.drop_seed = seed;
        a.drop_scale = 1.f / (1.f - drop_p);
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const bool f32 = x_dtype == VLSA_DT_F32;
    if (mode == mb::kGated) rc = f32 ? launch<mb::kGated, true>(a, nsl, st) : launch<mb::kGated, false>(a, nsl, st);
    else rc = f32 ? launch<mb::kTanh, true>(a, nsl, st) : launch<mb::kTanh, false>(a, nsl, st);
    if (rc != VLSA_OK) return rc;
    const long long tw = (long long)U * 512, tv = 3 * 512;
    hipLaunchKernelGGL(k_mb_reduce, dim3((unsigned int)((tw / 4 + 31) / 32)), dim3(256), 0, st, a.part_w, a.C, tw, dW);
    hipLaunchKernelGGL(k_mb_reduce, dim3((unsigned int)((tv / 4 + 31) / 32)), dim3(256), 0, st, a.part_v, a.C, tv, dvec);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// c1, c2 of the projecter's LayerNorm backward for one bag: stats [N][4] holds (mean, rstd) from the training forward
// (vlsa_feat_project_train); columns 2, 3 are written here from dy and the projected rows y.
extern "C" int vlsa_feat_project_rowstats(const float* dy, int64_t lddy, const float* y, int64_t ldy, int64_t N, const void* prep,
                                          float* stats, void* stream) {
    if (!dy || !y || !prep || !stats || N < 1 || lddy < 512 || ldy < 512) return VLSA_EINVAL;
    const FeatProjOffsets L;
    const unsigned char* pp = static_cast<const unsigned char*>(prep);
    hipLaunchKernelGGL(k_ln_bwd_rowstats, dim3((unsigned int)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, (long long)lddy, y,
                       (long long)ldy, (long long)N, reinterpret_cast<const float*>(pp + L.gamma), reinterpret_cast<const float*>(pp + L.beta),
                       stats);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// Backward of Feat_Projecter over B bags.  bag_desc / dy_desc: device tables {ptr, N, ld} of the input rows (bf16 or fp32) and of
// the upstream gradient rows (fp32 [N, 512]); stats: [sum N][4] (mean, rstd, c1, c2), bag b at row row_off[b]; prep: the block of
// vlsa_prepare_featproj.  Outputs: dW [512][512], dvec [3][512] = (db, dgamma, dbeta).
extern "C" int vlsa_feat_project_backward(const void* bag_desc, const void* dy_desc, int B, int x_dtype, const void* prep,
                                          const int* tile_start, int n_tiles, const float* stats, const int64_t* row_off, void* ws,
                                          float* dW, float* dvec, void* stream) {
    if (!bag_desc || !dy_desc || !prep || !tile_start || !stats || !row_off || !ws || !dW || !dvec || B < 1 || B > 64 || n_tiles < 1)
        return VLSA_EINVAL;
    if (x_dtype != VLSA_DT_BF16 && x_dtype != VLSA_DT_F32) return VLSA_EUNSUPPORTED;
    const FeatProjOffsets L;
    const unsigned char* pp = static_cast<const unsigned char*>(prep);
    MbArgs a{};
    a.bags = static_cast<const MbBag*>(bag_desc);
    a.dy = static_cast<const MbBag*>(dy_desc);
    a.tile_start = tile_start;
    a.row_off = reinterpret_cast<const long long*>(row_off);
    a.rowvec = stats;
    a.wpack = pp + L.wpack;
    a.bias_a = reinterpret_cast<const float*>(pp + L.bias);
    a.bias_g = nullptr;
    a.vec = reinterpret_cast<const float*>(pp + L.gamma);
    a.B = B;
    a.n_tiles = n_tiles;
    a.C = chunks_for(n_tiles, 4);
    a.part_w = static_cast<float*>(ws);
    a.part_v = a.part_w + (size_t)a.C * 512 * 512;
    hipStream_t st = (hipStream_t)stream;
    const int rc = x_dtype == VLSA_DT_F32 ? launch<mb::kLN, true>(a, 4, st) : launch<mb::kLN, false>(a, 4, st);
    if (rc != VLSA_OK) return rc;
    const long long tw = 512ll * 512, tv = 3 * 512;
    hipLaunchKernelGGL(k_mb_reduce, dim3((unsigned int)((tw / 4 + 31) / 32)), dim3(256), 0, st, a.part_w, a.C, tw, dW);
    hipLaunchKernelGGL(k_mb_reduce, dim3((unsigned int)((tv / 4 + 31) / 32)), dim3(256), 0, st, a.part_v, a.C, tv, dvec);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// =================================================================================================================================
// dL/dX of the (gated) attention pooling over the N patches -- needed only when the bag itself carries a gradient, i.e. when a
// trainable Feat_Projecter feeds a DeepMIL encoder (model/deepmil.py:267-283):
//     dx_n = dHa_n Wa + dHg_n Wg + A_n dpooled          (dH as above; A_n = softmax weight of the row, dpooled = dL/d pooled)
// One kernel, same building blocks as k_mlp_backward: a workgroup owns a 64-row tile (32 fp32 rows) and ALL hidden units (wave w:
// units [32 w, 32 w + 32) of both branches, so the gate product is wave-local), recomputes the pre-activations, forms dH in
// registers, publishes it to LDS as bf16 hi + lo in the X tile's own (swizzled, row-major) layout -- the X tile is dead by then --
// and contracts it over the hidden units against the un-scaled weights packed [k = hidden][n = column] (k_prepare_attn_dx_weights);
// wave w owns output columns [64 w, 64 w + 64).  3 bf16 terms (dH_hi W_hi + dH_lo W_hi + dH_hi W_lo).
namespace vlsa {
namespace adx {
constexpr int kLds = 2 * 65536;      // X tile (then dH hi) | dH lo
}

// packedT[((w * KS + ks) * 8 + ct * 2 + term) * 1024 + lane * 16 + 2 e] = term of Wcat[32 ks + 8 (lane >> 4) + e][64 w + 16 ct + (lane & 15)],
// Wcat = [Wa; Wg] stacked over the hidden units (256 or 512 rows), KS = rows / 32.  grid = 8 * KS * 8 workgroups of 64 threads.
__global__ __launch_bounds__(64) void k_prepare_attn_dx_weights(const float* __restrict__ Wa, const float* __restrict__ Wg, int gated,
                                                                 unsigned char* __restrict__ out) {
    const int KS = gated ? 16 : 8;
    const int blk = blockIdx.x, lane = threadIdx.x;
    const int f = blk & 7, ks = (blk >> 3) % KS, w = blk / (8 * KS);
    const int term = f & 1, ct = f >> 1;
    const int col = 64 * w + 16 * ct + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kc = k0 + e;
        const float x = kc < 256 ? Wa[(size_t)kc * 512 + col] : Wg[(size_t)(kc - 256) * 512 + col];
        const __bf16 hi = (__bf16)x;
        o[e] = term ? (__bf16)(x - (float)hi) : hi;
    }
    *reinterpret_cast<bf16x8*>(out + (size_t)blk * 1024 + lane * 16) = o;
}

struct AdxArgs {
    const MbBag* bags;          // [B] rows (bf16 or fp32)
    const MbBag* dxs;           // [B] fp32 gradient rows to write
    const int* tile_start;      // [B + 1] tiles of 64 (bf16) / 32 (fp32) rows
    const long long* row_off;   // [B] offset of bag b's rows in da / aw
    const float* da;            // dL/da
    const float* aw;            // softmax weights A_n of the pooling (nullable: no pooling term)
    const float* dpooled;       // [B][512] (nullable with aw)
    const unsigned char* wpack; // forward packing (recompute)
    const unsigned char* wT;    // k_prepare_attn_dx_weights
    const float *bias_a, *bias_g, *w2;
    int B, n_tiles;
    unsigned int drop_thr, drop_seed;
    float drop_scale;
};

template <bool GATED, bool XF32>
__global__ __launch_bounds__(512) void k_attn_scores_dx(const AdxArgs a) {
    constexpr int RT = XF32 ? 2 : 4;
    constexpr int ROWS = 16 * RT;
    constexpr int QB = ROWS * 256;
    constexpr int NB = GATED ? 2 : 1;
    constexpr int NF = GATED ? 4 : 2;
    constexpr int KS3 = GATED ? 16 : 8;            // k steps of the output contraction (hidden units / 32)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int t = blockIdx.x;
    // tile lookup
    const int tsv = lane < a.B ? a.tile_start[lane] : 0x7fffffff;
    const int b = __builtin_popcountll(__builtin_amdgcn_ballot_w64(tsv <= t)) - 1;
    const MbBag bag = a.bags[b], ob = a.dxs[b];
    const long long row0 = (long long)(t - a.tile_start[b]) * ROWS;
    const int nrows = (int)((bag.N - row0) < ROWS ? (bag.N - row0) : ROWS);
    const unsigned char* xsrc = static_cast<const unsigned char*>(bag.X) + row0 * bag.ldx * (XF32 ? 4 : 2);
    const float* dav = a.da + a.row_off[b] + row0;

    // ---- X tile -> LDS (layout of k_mlp_backward) ------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = tid + 512 * k;
        const int row = XF32 ? (p >> 7) : (p >> 6), ch = XF32 ? (p & 127) : (p & 63);
        u32x4_mb_t v = u32x4_mb_t{0u, 0u, 0u, 0u};
        if (row < nrows) v = *reinterpret_cast<const u32x4_mb_t*>(xsrc + ((size_t)row * bag.ldx * (XF32 ? 4 : 2)) + ch * 16);
        if constexpr (XF32) {
            bf16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int bits = v[e];
                const float f = __uint_as_float(bits);
                hi[e] = (__bf16)f;
                lo[e] = (__bf16)(f - (float)hi[e]);
            }
            const int off = (ch >> 5) * QB + mb_swz(row, (ch & 31) * 8);
            *reinterpret_cast<bf16x4_mb*>(smem + off) = hi;
            *reinterpret_cast<bf16x4_mb*>(smem + 65536 + off) = lo;
        } else {
            *reinterpret_cast<u32x4_mb*>(smem + (ch >> 4) * QB + mb_swz(row, (ch & 15) * 16)) = v;
        }
    }
    __syncthreads();

    // ---- phase 1: pre-activations of the wave's 32 hidden units (2 tiles) of both branches --------------------------------------
    f32x4 acch[RT][2][NB];
#pragma unroll
    for (int ht = 0; ht < 2; ++ht)
#pragma unroll
        for (int br = 0; br < NB; ++br) {
            const int h = 16 * (2 * w + ht) + i16;
            const float bb = br == 0 ? a.bias_a[h] : a.bias_g[h];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acch[rt][ht][br] = f32x4{bb, bb, bb, bb};
        }
    {
        const unsigned char* wp = a.wpack + (size_t)(2 * w) * 16 * NF * 1024 + lane * 16;      // tile 2 w; tile 2 w + 1 is 16 NF KB behind
        auto load_b = [&](int ks, bf16x8 (&dst)[2][NB][2]) {
#pragma unroll
            for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                for (int br = 0; br < NB; ++br)
#pragma unroll
                    for (int term = 0; term < 2; ++term)
                        dst[ht][br][term] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)(ht * 16 + ks) * NF + br * 2 + term) * 1024);
        };
        auto kstep = [&](int ks, const bf16x8 (&Bc)[2][NB][2]) {
            const int qoff = (ks >> 2) * QB, boff = (ks & 3) * 64 + g * 16;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 A = *reinterpret_cast<const bf16x8_mb*>(smem + qoff + mb_swz(16 * rt + i16, boff));
                bf16x8 AL = A;
                if constexpr (XF32) AL = *reinterpret_cast<const bf16x8_mb*>(smem + 65536 + qoff + mb_swz(16 * rt + i16, boff));
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
#pragma unroll
                    for (int br = 0; br < NB; ++br) {
                        acch[rt][ht][br] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bc[ht][br][0], acch[rt][ht][br], 0, 0, 0);
                        acch[rt][ht][br] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bc[ht][br][1], acch[rt][ht][br], 0, 0, 0);
                        if constexpr (XF32) acch[rt][ht][br] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL, Bc[ht][br][0], acch[rt][ht][br], 0, 0, 0);
                    }
            }
        };
        bf16x8 B0[2][NB][2], B1[2][NB][2];
        load_b(0, B0);
#pragma unroll 1
        for (int ks = 0; ks < 16; ks += 2) {
            load_b(ks + 1, B1);
            kstep(ks, B0);
            if (ks + 2 < 16) load_b(ks + 2, B0);
            kstep(ks + 1, B1);
        }
    }
    __syncthreads();                     // every wave is done with the X tile: its LDS becomes the dH image

    // ---- phase 2: dH -> LDS (bf16 hi at 0, lo at 64 KiB), element (row, kc = 256 br + hidden) in the X tile's layout ---------------
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const int hu = 16 * (2 * w + ht) + i16;
            const float w2v = a.w2[hu];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + 4 * g + r;
                const float dv = row < nrows ? dav[row] : 0.f;
                const float u = fast_exp2(fminf(acch[rt][ht][0][r], 43.f));
                const float th = (1.f - u) * __builtin_amdgcn_rcpf(1.f + u);
                float sg = 1.f, keep = 1.f;
                if constexpr (GATED) {
                    const float v = fast_exp2(fminf(acch[rt][ht][1][r], 57.f));
                    sg = __builtin_amdgcn_rcpf(1.f + v);
                    if (a.drop_thr != 0u) {
                        const unsigned int rid = (unsigned int)(row0 + row);
                        const bool ka = dropout_bits(a.drop_seed, rid, (unsigned int)hu) >= a.drop_thr;
                        const bool kg = dropout_bits(a.drop_seed, rid, (unsigned int)hu + 256u) >= a.drop_thr;
                        keep = (ka && kg) ? a.drop_scale * a.drop_scale : 0.f;
                    }
                }
                const float base = dv * w2v * keep;
                float d[2];
                d[0] = base * (1.f - th * th) * sg;
                d[1] = base * th * sg * (1.f - sg);
#pragma unroll
                for (int br = 0; br < NB; ++br) {
                    const int kc = 256 * br + hu;
                    const __bf16 hi = (__bf16)d[br];
                    const __bf16 lo = (__bf16)(d[br] - (float)hi);
                    const int off = (kc >> 7) * QB + mb_swz(row, (kc & 127) * 2);
                    *reinterpret_cast<__bf16 __attribute__((may_alias))*>(smem + off) = hi;
                    *reinterpret_cast<__bf16 __attribute__((may_alias))*>(smem + 65536 + off) = lo;
                }
            }
        }
    __syncthreads();

    // ---- phase 3: dX[rows][64 w .. + 63] = dH [rows][hidden] Wcat[hidden][cols] ----------------------------------------------------
    f32x4 acco[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acco[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        const unsigned char* wt = a.wT + (size_t)w * KS3 * 8 * 1024 + lane * 16;
        auto load_t = [&](int ks, bf16x8 (&dst)[4][2]) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int term = 0; term < 2; ++term) dst[ct][term] = *reinterpret_cast<const bf16x8*>(wt + ((size_t)ks * 8 + ct * 2 + term) * 1024);
        };
        auto ostep = [&](int ks, const bf16x8 (&Bc)[4][2]) {
            const int qoff = (ks >> 2) * QB, boff = (ks & 3) * 64 + g * 16;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 Ah = *reinterpret_cast<const bf16x8_mb*>(smem + qoff + mb_swz(16 * rt + i16, boff));
                const bf16x8 Al = *reinterpret_cast<const bf16x8_mb*>(smem + 65536 + qoff + mb_swz(16 * rt + i16, boff));
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    acco[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bc[ct][0], acco[rt][ct], 0, 0, 0);
                    acco[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bc[ct][0], acco[rt][ct], 0, 0, 0);
                    acco[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bc[ct][1], acco[rt][ct], 0, 0, 0);
                }
            }
        };
        bf16x8 T0[4][2], T1[4][2];
        load_t(0, T0);
#pragma unroll 1
        for (int ks = 0; ks < KS3; ks += 2) {
            load_t(ks + 1, T1);
            ostep(ks, T0);
            if (ks + 2 < KS3) load_t(ks + 2, T0);
            ostep(ks + 1, T1);
        }
    }
    // ---- epilogue: + A_n dpooled, store (C layout: lane column i16, rows 4 g + r) ---------------------------------------------------
    float* dst = static_cast<float*>(const_cast<void*>(ob.X)) + row0 * ob.ldx + 64 * w + i16;
    const float* awp = a.aw ? a.aw + a.row_off[b] + row0 : nullptr;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * g + r;
            if (row < nrows) {
                const float an = awp ? awp[row] : 0.f;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    float o = acco[rt][ct][r];
                    if (awp) o = fmaf(an, a.dpooled[(size_t)b * 512 + 64 * w + 16 * ct + i16], o);
                    dst[(size_t)row * ob.ldx + 16 * ct] = o;
                }
            }
        }
}

}  // namespace vlsa

extern "C" size_t vlsa_attn_dx_prep_bytes(int gated) { return (size_t)8 * (gated ? 16 : 8) * 8 * 1024; }

extern "C" int vlsa_prepare_attn_dx_weights(const float* Wa, const float* Wg, int gated, void* prep_t, void* stream) {
    if (!Wa || !prep_t || (gated && !Wg)) return VLSA_EINVAL;
    hipLaunchKernelGGL(k_prepare_attn_dx_weights, dim3(8 * (gated ? 16 : 8) * 8), dim3(64), 0, (hipStream_t)stream, Wa, Wg, gated ? 1 : 0,
                       static_cast<unsigned char*>(prep_t));
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// dL/dX of the (gated) attention pooling for B bags whose rows carry a gradient (see k_attn_scores_dx).  dx_desc: table of the fp32
// gradient rows to write; da / aw: dL/da and the softmax weights A_n of all bags' rows (bag b at a_off[b]); dpooled [B][512] (aw
// and dpooled may both be NULL: scores term only); prep / prep_t: vlsa_prepare_gated_weights / vlsa_prepare_attn_dx_weights.
extern "C" int vlsa_attn_scores_backward_dx(const void* bag_desc, const void* dx_desc, int B, int x_dtype, int D, const void* prep,
                                            const void* prep_t, int gated, const int* tile_start, int n_tiles, const float* da,
                                            const float* aw, const float* dpooled, const int64_t* a_off, float drop_p, unsigned int seed,
                                            void* stream) {
    if (!bag_desc || !dx_desc || !prep || !prep_t || !tile_start || !da || !a_off || B < 1 || B > 64 || n_tiles < 1) return VLSA_EINVAL;
    if ((aw == nullptr) != (dpooled == nullptr)) return VLSA_EINVAL;
    if (D != mb::kD || (x_dtype != VLSA_DT_BF16 && x_dtype != VLSA_DT_F32)) return VLSA_EUNSUPPORTED;
    const GatedPrepOffsets L(gated ? 1 : 0);
    const unsigned char* pp = static_cast<const unsigned char*>(prep);
    AdxArgs a{};
    a.bags = static_cast<const MbBag*>(bag_desc);
    a.dxs = static_cast<const MbBag*>(dx_desc);
    a.tile_start = tile_start;
    a.row_off = reinterpret_cast<const long long*>(a_off);
    a.da = da;
    a.aw = aw;
    a.dpooled = dpooled;
    a.wpack = pp + L.wpack;
    a.wT = static_cast<const unsigned char*>(prep_t);
    a.bias_a = reinterpret_cast<const float*>(pp + L.ba);
    a.bias_g = reinterpret_cast<const float*>(pp + L.bg);
    a.w2 = reinterpret_cast<const float*>(pp + L.w2);
    a.B = B;
    a.n_tiles = n_tiles;
    a.drop_thr = 0u;
    a.drop_scale = 1.f;
    if (gated && drop_p > 0.f) {
        if (!(drop_p < 1.f)) return VLSA_EINVAL;
        a.drop_thr = (unsigned int)((double)drop_p * 4294967296.0);
        if (a.drop_thr == 0u) a.drop_thr = 1u;
        a.drop_seed = seed;
        a.drop_scale = 1.f / (1.f - drop_p);
    }
    hipStream_t st = (hipStream_t)stream;
    static DeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)k_attn_scores_dx<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, adx::kLds);
        (void)hipFuncSetAttribute((const void*)k_attn_scores_dx<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, adx::kLds);
        (void)hipFuncSetAttribute((const void*)k_attn_scores_dx<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, adx::kLds);
        (void)hipFuncSetAttribute((const void*)k_attn_scores_dx<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, adx::kLds);
    }
    const bool f32 = x_dtype == VLSA_DT_F32;
#define VLSA_ADX(G, F) hipLaunchKernelGGL((k_attn_scores_dx<G, F>), dim3(n_tiles), dim3(512), adx::kLds, st, a)
    if (gated) { if (f32) VLSA_ADX(true, true); else VLSA_ADX(true, false); }
    else       { if (f32) VLSA_ADX(false, true); else VLSA_ADX(false, false); }
#undef VLSA_ADX
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// The descriptor tables of ONE bag written on the device from by-value arguments (what vlsa_amd.functional._row_tables uploads for a
// list of bags: [vlsa_bag_desc X | vlsa_bag_desc extra (optional) | row offset 0 | tile_start {0, n_tiles}]): the single-bag backward
// calls of the bag-by-bag training loop would otherwise stage ~60 bytes through pinned memory per call (~50 us of host time).
namespace vlsa {
__global__ void k_fill_one_bag_tables(long long* dst, const void* X, long long N, long long ld, const void* extra, long long extra_ld,
                                      int n_tiles) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int o = 0;
    dst[o++] = (long long)(uintptr_t)X; dst[o++] = N; dst[o++] = ld;
    if (extra) { dst[o++] = (long long)(uintptr_t)extra; dst[o++] = N; dst[o++] = extra_ld; }
    dst[o++] = 0;                                           // row offset of bag 0
    int* ts = reinterpret_cast<int*>(dst + o);
    ts[0] = 0;
    ts[1] = n_tiles;
}
}  // namespace vlsa

extern "C" int vlsa_fill_one_bag_tables(void* dst, const void* X, int64_t N, int64_t ld, const void* extra, int64_t extra_ld,
                                        int tile_rows, void* stream) {
    if (!dst || !X || N < 1 || tile_rows < 1) return VLSA_EINVAL;
    const int n_tiles = (int)((N + tile_rows - 1) / tile_rows);
    hipLaunchKernelGGL(vlsa::k_fill_one_bag_tables, dim3(1), dim3(64), 0, (hipStream_t)stream, static_cast<long long*>(dst), X, (long long)N,
                       (long long)ld, extra, (long long)extra_ld, n_tiles);
    return hipGetLastError() == hipSuccess ? n_tiles : VLSA_ELAUNCH;
}

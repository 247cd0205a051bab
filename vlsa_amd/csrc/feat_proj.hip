// Feat_Projecter over ALL N patches of a bag: Y = LayerNorm(X W^T + b) * gamma + beta, W in [512, 512]
// (reference model/layers.py:65-82, applied by VLFAN / DeepMIL when use_feat_proj=True: model/deepmil.py:176-179,267-268).
// The reference -- and round 1 here -- runs it as a library GEMM that writes [N, 512] fp32 pre-activations plus a LayerNorm
// launch that re-reads them.  Here: ONE kernel, structure of k_gated_scores (gated_scores.hip), the product never leaves
// registers before it is normalised.
//
//   * workgroup = 16 RT rows x ALL 512 output columns, 8 waves; wave w owns output columns [64 w, 64 w + 64) = 4 column tiles
//     x RT row tiles = 16 RT accumulator registers per lane (RT = 8: 128 rows per workgroup, 128 registers).
//   * K loop in 16 steps of 32: the wave's weight fragments of a step (bf16 hi + lo split of the fp32 weights, packed in
//     fragment order once per parameter version: 8 KB per wave and step, L2 resident) go straight into a double-buffered
//     register ring; the step's X chunk (rows x 64 B) is loaded two steps ahead into registers and published to a
//     double-buffered LDS tile (16-B chunks XOR-swizzled for the ds_read_b128 lane groups).  One barrier per step.
//   * bf16 X is consumed exactly (2 MFMA terms: X W_hi + X W_lo); fp32 X (the reference's own feature format) is split into
//     bf16 hi + lo on the fly, both images published, 3 terms (X_hi W_hi + X_hi W_lo + X_lo W_hi; RT = 4: register budget).
//   * epilogue: accumulators start at the bias; row mean, then centred variance (two passes like nn.LayerNorm, biased
//     variance, eps inside the root) by DPP row sums + an LDS reduction over the 8 waves; y = (x - mean) rstd gamma + beta
//     stored fp32 row-major -- the bag the fp32 streaming / scoring kernels then consume.
// Roofline: MFMA-bound, 2 terms x 2 x 512 x 512 = 1.05 MFLOP per patch (the same as the gated attention scores) plus 2 KB
// per patch written.
#include "vlsa_common.h"

#ifndef FP_XBUF
#define FP_XBUF 1
#endif

namespace vlsa {

typedef bf16x8 __attribute__((may_alias)) bf16x8_maf;
typedef float __attribute__((may_alias)) float_maf;
typedef f32x4 __attribute__((may_alias)) f32x4_maf;

namespace fp {
constexpr int kD = 512;                           // input and output width
constexpr int kSteps = 16;                        // K steps of 32
constexpr int kNF = 8;                            // weight fragments per step and wave: 4 column tiles x (hi, lo)
constexpr int kMaxRows = 128;
constexpr int kXBuf = kMaxRows * 64;              // one K step of the tile, one bf16 image: 128 rows x 32 bf16 = 8 KiB
constexpr int kScrOff = 4 * kXBuf;                // 2 buffers x (hi, lo image)
constexpr int kStatOff = kScrOff + 8 * kMaxRows * 4;
constexpr int kLds = kStatOff + kMaxRows * 4;     // 37,376 B
}  // namespace fp

struct FeatProjLayout {
    size_t wpack, bias, gamma, beta, total;
    __host__ __device__ FeatProjLayout() {
        wpack = 0;
        bias = wpack + (size_t)8 * fp::kSteps * fp::kNF * 1024;   // 1 MiB
        gamma = bias + fp::kD * 4;
        beta = gamma + fp::kD * 4;
        total = beta + fp::kD * 4;
    }
};

// packed[((w * 16 + ks) * 8 + f) * 1024 + lane * 16 + 2 e] = term(f & 1) of W[64 w + 16 (f >> 1) + (lane & 15)][32 ks + 8 (lane >> 4) + e]
// grid = 8 * 16 * 8 workgroups of 64 threads.
__global__ __launch_bounds__(64) void k_prepare_featproj(const float* __restrict__ W, const float* __restrict__ b,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          unsigned char* __restrict__ prep) {
    const FeatProjLayout L;
    const int blk = blockIdx.x, lane = threadIdx.x;
    const int f = blk % fp::kNF, ks = (blk / fp::kNF) % fp::kSteps, w = blk / (fp::kNF * fp::kSteps);
    const int term = f & 1, ct = f >> 1;
    const int col = 64 * w + 16 * ct + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = W[(size_t)col * fp::kD + k0 + e];
        const __bf16 hi = (__bf16)x;
        o[e] = term ? (__bf16)(x - (float)hi) : hi;
    }
    *reinterpret_cast<bf16x8*>(prep + L.wpack + (size_t)blk * 1024 + lane * 16) = o;
    if (blk == 0) {
        float* pb = reinterpret_cast<float*>(prep + L.bias);
        float* pg = reinterpret_cast<float*>(prep + L.gamma);
        float* pe = reinterpret_cast<float*>(prep + L.beta);
        for (int i = lane; i < fp::kD; i += 64) {
            pb[i] = b ? b[i] : 0.f;
            pg[i] = gamma ? gamma[i] : 1.f;
            pe[i] = beta ? beta[i] : 0.f;
        }
    }
}

// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane
__device__ __forceinline__ float fp_row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}

template <bool XF32, int RT>
__global__ __launch_bounds__(512) void k_feat_proj(const void* __restrict__ Xv, long long N, long long ldx,
                                                    const unsigned char* __restrict__ prep, float eps, float* __restrict__ Y,
                                                    long long ldy, float* __restrict__ stats) {
    using namespace fp;
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const long long row0 = (long long)blockIdx.x * ROWS;
    const int nrows = (int)((N - row0) < ROWS ? (N - row0) : ROWS);
    const FeatProjLayout L;

    const unsigned char* wp = prep + L.wpack + (size_t)w * kSteps * kNF * 1024 + lane * 16;
    // the thread's share of a step's X chunk: row xr, 16-byte bf16 chunk xc (8 k-values)
    const int xr = tid >> 2, xc = tid & 3;
    const bool xok = xr < nrows;                  // (threads with xr >= ROWS have xr >= nrows as well)
    const __bf16* xsrc = XF32 ? nullptr : static_cast<const __bf16*>(Xv) + (row0 + xr) * ldx + xc * 8;      // + 32 ks
    const float* xsrc32 = XF32 ? static_cast<const float*>(Xv) + (row0 + xr) * ldx + xc * 8 : nullptr;
    // 16-B chunk c of row r is stored at position c ^ f(r), f(r) = (-(r >> 2)) & 3 (see k_gated_scores)
    const int x_dst = xr * 64 + ((xc ^ ((0 - (xr >> 2)) & 3)) << 4);
    const int a_off = i16 * 64 + ((g ^ ((0 - (i16 >> 2)) & 3)) << 4);    // A fragment of row tile rt: + rt * 1024
    // X through range-checked buffer loads: rows past the end of the bag (and threads past the tile) read zeros without a branch, so
    // that the compiler's s_waitcnt counts stay exact (see k_gated_scores)
    constexpr int XESZ = XF32 ? 4 : 2;
    const unsigned long long xbase = reinterpret_cast<unsigned long long>(Xv) + (unsigned long long)row0 * ldx * XESZ;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>((static_cast<unsigned long long>((unsigned)__builtin_amdgcn_readfirstlane((int)(xbase >> 32))) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)xbase)),
        0, __builtin_amdgcn_readfirstlane((int)(((long long)(nrows - 1) * ldx + kD) * XESZ)), 0x00020000);
    const int xvoff = (int)((long long)xr * ldx + xc * 8) * XESZ;
    typedef int i32x4fp __attribute__((ext_vector_type(4)));
    struct XReg { bf16x8 h; f32x4 f[XF32 ? 2 : 1]; };
    auto load_x = [&](int ks) -> XReg {
        XReg r = {};
        if constexpr (FP_XBUF) {
            if constexpr (XF32) {
                r.f[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff, 128 * ks, 0));
                r.f[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff, 128 * ks + 16, 0));
            } else {
                r.h = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff, 64 * ks, 0));
            }
        } else if (xok) {
            if constexpr (XF32) {
                r.f[0] = *reinterpret_cast<const f32x4*>(xsrc32 + 32 * ks);
                r.f[1] = *reinterpret_cast<const f32x4*>(xsrc32 + 32 * ks + 4);
            } else {
                r.h = *reinterpret_cast<const bf16x8*>(xsrc + 32 * ks);
            }
        }
        return r;
    };
    auto load_b = [&](int ks, bf16x8 (&dst)[kNF]) {
#pragma unroll
        for (int f = 0; f < kNF; ++f) dst[f] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(ks * kNF + f) * 1024);
    };

    // accumulators start at the bias of the lane's output column (C layout: lane (j = i16, g) holds rows 4 g + r of column j)
    f32x4 acc[RT][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const float bb = reinterpret_cast<const float*>(prep + L.bias)[64 * w + 16 * ct + i16];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = f32x4{bb, bb, bb, bb};
    }

    bf16x8 B0[kNF], B1[kNF];
    XReg X0, X1;
    X0 = load_x(0);
    load_b(0, B0);
    X1 = load_x(1);

    auto step = [&](int s, const bool body, bf16x8 (&cur)[kNF], bf16x8 (&nxt)[kNF], XReg& xcur) {
        unsigned char* xb = smem + (s & 1) * 2 * kXBuf;     // hi image; fp32 bags: lo image behind it
        if (xr < ROWS) {
            if constexpr (XF32) {
                bf16x8 h, l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = xcur.f[e >> 2][e & 3];
                    const __bf16 a = (__bf16)v;
                    h[e] = a;
                    l[e] = (__bf16)(v - (float)a);
                }
                *reinterpret_cast<bf16x8_maf*>(xb + x_dst) = h;
                *reinterpret_cast<bf16x8_maf*>(xb + kXBuf + x_dst) = l;
            } else {
                *reinterpret_cast<bf16x8_maf*>(xb + x_dst) = xcur.h;
            }
        }
        __syncthreads();                     // X(s) published by every wave; everyone is done reading buffer (s + 1) & 1
        if (body || s + 1 < kSteps) load_b(s + 1, nxt);
        if (body || s + 2 < kSteps) xcur = load_x(s + 2);
#pragma unroll
        for (int q = 0; q < RT / 2; ++q) {   // two row tiles at a time (8 A-fragment registers; groups of four spilled)
            bf16x8 A[2], AL[XF32 ? 2 : 1];
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                A[r2] = *reinterpret_cast<const bf16x8_maf*>(xb + (2 * q + r2) * 1024 + a_off);
                if constexpr (XF32) AL[r2] = *reinterpret_cast<const bf16x8_maf*>(xb + kXBuf + (2 * q + r2) * 1024 + a_off);
            }
            // hi terms of the 8 accumulators of this group, then the lo terms: MFMAs on one accumulator are 8 apart
#pragma unroll
            for (int term = 0; term < 2; ++term)
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc[2 * q + r2][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[r2], cur[2 * ct + term], acc[2 * q + r2][ct], 0, 0, 0);
            if constexpr (XF32) {
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc[2 * q + r2][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL[r2], cur[2 * ct], acc[2 * q + r2][ct], 0, 0, 0);
            }
        }
    };
    constexpr bool PEEL = FP_XBUF;
#pragma unroll 1
    for (int s = 0; s < (PEEL ? kSteps - 2 : kSteps); s += 2) {
        step(s, PEEL, B0, B1, X0);
        step(s + 1, PEEL, B1, B0, X1);
    }
    if constexpr (PEEL) {
        step(kSteps - 2, false, B0, B1, X0);
        step(kSteps - 1, false, B1, B0, X1);
    }

    // ---- epilogue: LayerNorm over the 512 columns of every row (8 waves x 64 columns), two passes
    float_maf* scr = reinterpret_cast<float_maf*>(smem + kScrOff);     // [8 waves][ROWS]
    float_maf* stat = reinterpret_cast<float_maf*>(smem + kStatOff);   // [ROWS]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = fp_row16_sum((acc[rt][0][r] + acc[rt][1][r]) + (acc[rt][2][r] + acc[rt][3][r]));
            if (i16 == 0) scr[w * ROWS + 16 * rt + 4 * g + r] = s;
        }
    __syncthreads();
    float row_mean = 0.f;
    if (tid < ROWS) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) s += scr[ww * ROWS + tid];
        row_mean = s * (1.f / kD);
        stat[tid] = row_mean;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const f32x4 mean = *reinterpret_cast<const f32x4_maf*>(&stat[16 * rt + 4 * g]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] -= mean;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float q = fp_row16_sum((acc[rt][0][r] * acc[rt][0][r] + acc[rt][1][r] * acc[rt][1][r]) +
                                         (acc[rt][2][r] * acc[rt][2][r] + acc[rt][3][r] * acc[rt][3][r]));
            if (i16 == 0) scr[w * ROWS + 16 * rt + 4 * g + r] = q;     // (every wave's reads of scr are behind the barrier above)
        }
    }
    __syncthreads();
    if (tid < ROWS) {
        float q = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) q += scr[ww * ROWS + tid];
        const float rstd = rsqrtf(q * (1.f / kD) + eps);
        stat[tid] = rstd;
        if (stats != nullptr && tid < nrows) {      // training: the LayerNorm statistics the backward kernel needs, [N][4]
            stats[(row0 + tid) * 4] = row_mean;
            stats[(row0 + tid) * 4 + 1] = rstd;
        }
    }
    __syncthreads();
    float gm[4], bt[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        gm[ct] = reinterpret_cast<const float*>(prep + L.gamma)[64 * w + 16 * ct + i16];
        bt[ct] = reinterpret_cast<const float*>(prep + L.beta)[64 * w + 16 * ct + i16];
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const f32x4 rstd = *reinterpret_cast<const f32x4_maf*>(&stat[16 * rt + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * g + r;
            if (row < nrows) {
                float* yr = Y + (row0 + row) * ldy + 64 * w + i16;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) yr[16 * ct] = acc[rt][ct][r] * rstd[r] * gm[ct] + bt[ct];
            }
        }
    }
}

}  // namespace vlsa

using namespace vlsa;

extern "C" size_t vlsa_featproj_prep_bytes(void) { return FeatProjLayout().total; }

extern "C" int vlsa_prepare_featproj(const float* W, const float* b, const float* gamma, const float* beta, int dim_in, int dim_out,
                                     void* prep, void* stream) {
    if (!W || !prep) return VLSA_EINVAL;
    if (dim_in != fp::kD || dim_out != fp::kD) return VLSA_EUNSUPPORTED;
    hipLaunchKernelGGL(k_prepare_featproj, dim3(8 * fp::kSteps * fp::kNF), dim3(64), 0, (hipStream_t)stream, W, b, gamma, beta,
                       static_cast<unsigned char*>(prep));
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

static int feat_project_impl(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, float eps, float* Y,
                             int64_t ldy, float* stats, void* stream) {
    if (!X || !prep || !Y || N < 1 || ldx < D || ldy < D) return VLSA_EINVAL;
    if (D != fp::kD || (x_dtype != VLSA_DT_BF16 && x_dtype != VLSA_DT_F32)) return VLSA_EUNSUPPORTED;
    const bool f32 = x_dtype == VLSA_DT_F32;
    const long long esz = f32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || ((ldx * esz) % 16)) return VLSA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* pp = static_cast<const unsigned char*>(prep);
    // rows per workgroup: every workgroup streams the whole 1 MB of packed weights from L2, so the largest tile (bf16 128, fp32
    // 64 rows) that still leaves >= 120 workgroups, else 32 rows (10k-patch bag: 79 tiles of 128 rows 33.7 us, 157 of 64 27 us,
    // 313 of 32 30-35 us; tools/kbench_featproj.py)
#define VLSA_FP(X32, RT_)                                                                                                  \
    hipLaunchKernelGGL((k_feat_proj<X32, RT_>), dim3((unsigned int)((N + 16 * RT_ - 1) / (16 * RT_))), dim3(512), fp::kLds, st, X, \
                       (long long)N, (long long)ldx, pp, eps, Y, (long long)ldy, stats)
    if (f32) {
        if (N >= 120 * 64) VLSA_FP(true, 4);
        else VLSA_FP(true, 2);
    } else {
        if (N >= 120 * 128) VLSA_FP(false, 8);
        else if (N >= 120 * 64) VLSA_FP(false, 4);
        else VLSA_FP(false, 2);
    }
#undef VLSA_FP
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

extern "C" int vlsa_feat_project(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, float eps, float* Y,
                                 int64_t ldy, void* stream) {
    return feat_project_impl(X, x_dtype, N, ldx, D, prep, eps, Y, ldy, nullptr, stream);
}

// Training forward: as vlsa_feat_project, and stats [N][4] receives (row mean, 1 / sqrt(var + eps)) of the pre-LayerNorm
// activations in columns 0, 1 (columns 2, 3 are filled by vlsa_feat_project_rowstats in the backward pass).
extern "C" int vlsa_feat_project_train(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, float eps, float* Y,
                                       int64_t ldy, float* stats, void* stream) {
    if (!stats) return VLSA_EINVAL;
    return feat_project_impl(X, x_dtype, N, ldx, D, prep, eps, Y, ldy, stats, stream);
}

// Attention scores of the ABMIL-style pooling modules over ALL N patches of a bag:
//     Gated_Attention_Pooling (model/layers.py:85-122):  a_n = w2 . (tanh(Wa x_n + ba) * sigmoid(Wg x_n + bg)) + c
//     Attention_Pooling       (model/layers.py:125-153): a_n = w2 .  tanh(Wa x_n + ba)                       + c
// with Wa, Wg in [256, 512].  This is the one MFMA-bound piece of the path (2 x 512 x 256 FLOP per patch and branch);
// the reference (and the first version here) runs it as two library GEMMs that write [N, 256] fp32 hidden activations to
// memory and re-read them.  Here: ONE kernel, hidden activations never leave registers.
//
//   * workgroup = 128 rows x all 512 hidden units (both branches), 8 waves; wave w owns hidden units [32w, 32w+32) of BOTH
//     branches, so the gate product is wave-local: 16 C tiles (8 row tiles x 2 column tiles) per branch = 128 accumulator
//     registers per lane.
//   * K loop in 16 steps of 32: the wave's weight fragments of a step (bf16 hi + lo split of the fp32 weights, packed in
//     fragment order by k_prepare_gated_weights: 8 KB, contiguous, L2 resident) are loaded straight into registers one step
//     ahead (double-buffered); the wave's 1 KB share of the step's X chunk (128 rows x 64 B) is loaded two steps ahead into
//     registers and published to a double-buffered LDS tile (16-B chunks XOR-swizzled so that the A-fragment ds_read_b128
//     is conflict free).  One barrier per step.  Plain loads only: the compiler counts vmcnt exactly (a variant with
//     LDS-DMA for X and register loads for W showed that the two kinds do NOT retire in one common order).
//   * X is bf16 and consumed exactly; weights are 2-term bf16 splits (rel. 2^-17); fp32 accumulation: 64 MFMAs
//     (16x16x32 bf16) per step and wave, hi and lo terms interleaved so that back-to-back MFMAs never share an accumulator.
//   * epilogue: bias, tanh / sigmoid (v_exp_f32), gate, dot with w2 over the 32 hidden units of the wave (xor shuffles
//     over the 16 column lanes), cross-wave sum through LDS, + c.
// Roofline: nominally MFMA-bound: 2 terms x 2 branches x 2 x 512 x 256 = 1.05 MFLOP per patch -> 52 GFLOP per 50k bag =
// 21 us at 2.5 PFLOP/s dense bf16.  Measured 74 us per 50k bag (28 %; 35 % at N = 400k); a K step costs a flat 2.0 us
// (1.35 us without the gate branch) in every pipeline tried (LDS-DMA ring for the weights, register double-buffering,
// reordered MFMAs, rotated K order), also with only 22 of the 256 CUs busy: ~1600 cycles of per-step fixed cost (barrier,
// X publish, 8 A reads) + ~1600 cycles per (32 KB of weights + 32 MFMAs per wave).  Per step and CU 72 KB come in
// (the 1 MB of packed weights is re-read for every 128-row tile: 8 KB per patch row) = 15 B/clk/CU.
// PMC (N = 400k): MFMA pipe busy 37 % of the kernel, waves parked at waitcnt / barrier 40 %, issue-stalled 35 %, LDS bank
// conflicts 0 (30 % before the lane-group-aware swizzle -- which did not change the time: LDS is not the limit).  The weight
// fragments are prefetched exactly one step ahead (register budget: 128 accumulators + 2 x 32 fragment registers of 256), and
// with the waves in barrier lock step that is not enough to cover an 8 x 8 KB L2 burst per CU.
// Next step (not done): 256 rows x half of the hidden units per workgroup = 6 instead of 9 KB per row, longer K steps.
#include "vlsa_common.h"

namespace vlsa {

typedef __attribute__((address_space(3))) void* lds_void_ptr_g;
typedef bf16x8 __attribute__((may_alias)) bf16x8_mag;
typedef float __attribute__((may_alias)) float_mag;
typedef int i32x4g __attribute__((ext_vector_type(4)));

namespace gs {
constexpr int kRows = 128;
constexpr int kHid = 256;
constexpr int kD = 512;
constexpr int kSteps = 16;
constexpr int kXBuf = kRows * 64;                 // one K step of the tile: 128 rows x 32 bf16
constexpr int kXOff = 0;
constexpr int kScrOff = kXOff + 2 * kXBuf;        // 16 KiB
constexpr int kLds = kScrOff + 8 * kRows * 4;     // + 4 KiB
}  // namespace gs

struct GatedPrepLayout {
    size_t wpack, ba, bg, w2, c, total;
    __host__ __device__ explicit GatedPrepLayout(int gated) {
        wpack = 0;
        ba = wpack + (size_t)8 * gs::kSteps * (gated ? 8 : 4) * 1024;
        bg = ba + gs::kHid * 4;
        w2 = bg + gs::kHid * 4;
        c = w2 + gs::kHid * 4;
        total = c + 16;
    }
};

// packed[((w * 16 + ks) * NF + f) * 1024 + lane * 16 + 2 e] = term(f & 1) of W_br[32 w + 16 ct + (lane & 15)][32 ks + 8 (lane >> 4) + e]
// with f = (br * 2 + ct) * 2 + term.   grid = 8 * 16 * NF workgroups of 64 threads.
__global__ __launch_bounds__(64) void k_prepare_gated_weights(const float* __restrict__ Wa, const float* __restrict__ ba,
                                                               const float* __restrict__ Wg, const float* __restrict__ bg,
                                                               const float* __restrict__ w2, const float* __restrict__ c,
                                                               int gated, unsigned char* __restrict__ prep) {
    const GatedPrepLayout L(gated);
    const int NF = gated ? 8 : 4;
    const int blk = blockIdx.x, lane = threadIdx.x;
    const int f = blk % NF, ks = (blk / NF) % gs::kSteps, w = blk / (NF * gs::kSteps);
    const int term = f & 1, ct = (f >> 1) & 1, br = f >> 2;
    const float* W = br ? Wg : Wa;
    const int h = 32 * w + 16 * ct + (lane & 15), k0 = 32 * ks + 8 * (lane >> 4);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = W[(size_t)h * gs::kD + k0 + e];
        const __bf16 hi = (__bf16)x;
        o[e] = term ? (__bf16)(x - (float)hi) : hi;
    }
    *reinterpret_cast<bf16x8*>(prep + L.wpack + (size_t)blk * 1024 + lane * 16) = o;
    if (blk == 0) {
        float* pba = reinterpret_cast<float*>(prep + L.ba);
        float* pbg = reinterpret_cast<float*>(prep + L.bg);
        float* pw2 = reinterpret_cast<float*>(prep + L.w2);
        for (int i = lane; i < gs::kHid; i += 64) {
            pba[i] = ba ? ba[i] : 0.f;
            pbg[i] = (gated && bg) ? bg[i] : 0.f;
            pw2[i] = w2[i];
        }
        if (lane == 0) reinterpret_cast<float*>(prep + L.c)[0] = c ? c[0] : 0.f;
    }
}

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh x = 1 - 2 / (exp(2x) + 1); exp through v_exp_f32; |x| clamped where the result is +-1 in fp32 anyway
    const float t = fminf(fmaxf(x, -15.f), 15.f);
    const float e2 = fast_exp2(t * (2.f * kLog2e));
    return 1.f - 2.f / (e2 + 1.f);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
    const float t = fminf(fmaxf(x, -80.f), 80.f);
    return 1.f / (1.f + fast_exp2(-t * kLog2e));
}

template <bool GATED>
__global__ __launch_bounds__(512) void k_gated_scores(const __bf16* __restrict__ X, long long N, long long ldx,
                                                       const unsigned char* __restrict__ prep, float* __restrict__ a_out) {
    using namespace gs;
    constexpr int NF = GATED ? 8 : 4;     // weight fragments per step and wave
    constexpr int NB = GATED ? 4 : 2;     // (branch, column tile) pairs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const long long row0 = (long long)blockIdx.x * kRows;
    const int nrows = (int)((N - row0) < kRows ? (N - row0) : kRows);
    const GatedPrepLayout L(GATED ? 1 : 0);

    // plain (compiler-tracked) loads only: weight fragments one step ahead into registers, the wave's 1 KB share of the X
    // chunk two steps ahead into registers and from there into the shared LDS buffer of its step
    const unsigned char* wp = prep + L.wpack + (size_t)w * kSteps * NF * 1024 + lane * 16;
    const int xr = 16 * w + (lane >> 2);                         // this lane's row of the X chunk
    const bool xok = xr < nrows;
    const __bf16* xsrc = X + (row0 + xr) * ldx + (lane & 3) * 8;  // + 32 ks
    // 16-B chunk c of row r is stored at position c ^ f(r), f(r) = (-(r >> 2)) & 3: ds_read_b128 is serviced in the lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), and with this f the 16 lanes of every
    // group hit 16 different 4-bank sets (the plain (r >> 2) & 3 swizzle measured 2-way conflicts: PMC 30 %)
    const int x_dst = xr * 64 + (((lane & 3) ^ ((0 - (xr >> 2)) & 3)) << 4);
    const int a_off = i16 * 64 + ((g ^ ((0 - (i16 >> 2)) & 3)) << 4);    // A fragment of row tile rt: + rt * 1024
    auto load_x = [&](int ks) -> bf16x8 {
        bf16x8 z = {};
        return xok ? *reinterpret_cast<const bf16x8*>(xsrc + 32 * ks) : z;
    };
    auto load_b = [&](int ks, bf16x8 (&dst)[NF]) {
#pragma unroll
        for (int f = 0; f < NF; ++f) dst[f] = *reinterpret_cast<const bf16x8*>(wp + (size_t)(ks * NF + f) * 1024);
    };

    f32x4 acc[8][NB];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[rt][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 B0[NF], B1[NF], X0, X1;
    X0 = load_x(0);
    load_b(0, B0);
    X1 = load_x(1);

    // one K step: publish this step's X share, barrier, start the loads of the next steps, 8 A reads, 64 (32) MFMAs
    auto step = [&](int s, bf16x8 (&cur)[NF], bf16x8 (&nxt)[NF], bf16x8& xcur) {
        unsigned char* xb = smem + kXOff + (s & 1) * kXBuf;
        *reinterpret_cast<bf16x8_mag*>(xb + x_dst) = xcur;
        __syncthreads();                     // X(s) published by every wave; everyone is done reading buffer (s + 1) & 1
        if (s + 1 < kSteps) load_b(s + 1, nxt);
        if (s + 2 < kSteps) xcur = load_x(s + 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 A[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) A[r4] = *reinterpret_cast<const bf16x8_mag*>(xb + (4 * h + r4) * 1024 + a_off);
            // per pair of row tiles: the hi terms of its 2 NB accumulators, then the lo terms -- two MFMAs on the same
            // accumulator are always 2 NB instructions apart (a dependent back-to-back pair stalls for the MFMA latency)
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2)
#pragma unroll
                for (int term = 0; term < 2; ++term)
#pragma unroll
                    for (int r4 = rp; r4 < rp + 2; ++r4)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[4 * h + r4][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[r4], cur[2 * b + term], acc[4 * h + r4][b], 0, 0, 0);
        }
    };
#pragma unroll 1
    for (int s = 0; s < kSteps; s += 2) {
        step(s, B0, B1, X0);
        step(s + 1, B1, B0, X1);
    }

    // ---- epilogue: activations, gate, dot with w2 over this wave's 32 hidden units, then over the 8 waves ----------
    const float* pba = reinterpret_cast<const float*>(prep + L.ba);
    const float* pbg = reinterpret_cast<const float*>(prep + L.bg);
    const float* pw2 = reinterpret_cast<const float*>(prep + L.w2);
    float bav[2], bgv[2], w2v[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int h = 32 * w + 16 * ct + i16;
        bav[ct] = pba[h];
        bgv[ct] = GATED ? pbg[h] : 0.f;
        w2v[ct] = pw2[h];
    }
    float_mag* scr = reinterpret_cast<float_mag*>(smem + kScrOff);
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float e = fast_tanh(acc[rt][ct][r] + bav[ct]);
                if (GATED) e *= fast_sigmoid(acc[rt][2 + ct][r] + bgv[ct]);
                v = fmaf(e, w2v[ct], v);
            }
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (i16 == 0) scr[w * kRows + 16 * rt + 4 * g + r] = v;
        }
    __syncthreads();
    if (tid < kRows && tid < nrows) {
        float s = reinterpret_cast<const float*>(prep + L.c)[0];
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) s += scr[ww * kRows + tid];
        a_out[row0 + tid] = s;
    }
}

}  // namespace vlsa

using namespace vlsa;

extern "C" size_t vlsa_gated_prep_bytes(int gated) { return GatedPrepLayout(gated ? 1 : 0).total; }

extern "C" int vlsa_prepare_gated_weights(const float* Wa, const float* ba, const float* Wg, const float* bg, const float* w2,
                                          const float* c, int dim_in, int dim_hid, int gated, void* prep, void* stream) {
    if (!Wa || !w2 || !prep || (gated && !Wg)) return VLSA_EINVAL;
    if (dim_in != gs::kD || dim_hid != gs::kHid) return VLSA_EUNSUPPORTED;
    const int NF = gated ? 8 : 4;
    hipLaunchKernelGGL(k_prepare_gated_weights, dim3(8 * gs::kSteps * NF), dim3(64), 0, (hipStream_t)stream, Wa, ba, Wg, bg, w2, c,
                       gated ? 1 : 0, static_cast<unsigned char*>(prep));
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

extern "C" int vlsa_gated_scores(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, int gated, float* a,
                                 void* stream) {
    if (!X || !prep || !a || N < 1 || ldx < D) return VLSA_EINVAL;
    if (D != gs::kD || x_dtype != VLSA_DT_BF16) return VLSA_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || ((ldx * 2) % 16) || ldx * 2 * gs::kRows >= (1ll << 31)) return VLSA_EINVAL;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)k_gated_scores<true>, hipFuncAttributeMaxDynamicSharedMemorySize, gs::kLds);
        (void)hipFuncSetAttribute((const void*)k_gated_scores<false>, hipFuncAttributeMaxDynamicSharedMemorySize, gs::kLds);
    }
    const unsigned int tiles = (unsigned int)((N + gs::kRows - 1) / gs::kRows);
    if (gated)
        hipLaunchKernelGGL(k_gated_scores<true>, dim3(tiles), dim3(512), gs::kLds, (hipStream_t)stream, static_cast<const __bf16*>(X),
                           (long long)N, (long long)ldx, static_cast<const unsigned char*>(prep), a);
    else
        hipLaunchKernelGGL(k_gated_scores<false>, dim3(tiles), dim3(512), gs::kLds, (hipStream_t)stream, static_cast<const __bf16*>(X),
                           (long long)N, (long long)ldx, static_cast<const unsigned char*>(prep), a);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// Backward of the VLFAN cross-attention aggregation w.r.t. the queries (X carries no gradient in the
// reference: SURVEY.md 8(a) row a14).  One more streaming pass over X, same skeleton as the forward
// k_vlfan_partial_mfma (vlfan_partial.hip):
//
//   t_pn  = log2-domain score (recomputed),  A_pn = exp2(t_pn - m2_p) / l_p      (saved m2, l)
//   dA_pn = dout_p . x_n                     (a second score-type contraction, "queries" = dout rows)
//   dS_pn = A_pn (dA_pn - delta_p),          delta_p = dout_p . out_p = sum_n A_pn dA_pn
//   de_p  = scale * sum_n dS_pn x_n / max(|x_n|, eps)      (a PV-type contraction with signed weights)
//
// where e_p is the effective query (q^_p - q^_gate).  The chain rule through the normalisation / gating of the
// raw queries is P x D work and stays on the host side (vlsa_amd/functional.py).  Partials are plain sums; they
// are reduced with vlsa_vlfan_merge (pm = 0, pl = 1, normalise = 0).
#include "vlfan_mfma_common.h"

namespace vlsa {

// dsplit[t][p][:] = 3-term bf16 split of dout[p][:] (rows >= P zero); delta[p] = dout_p . out_p
__global__ __launch_bounds__(256) void k_prepare_backward(const float* __restrict__ dout, const float* __restrict__ out,
                                                           int P, int D, __bf16* __restrict__ dsplit,
                                                           float* __restrict__ delta) {
    __shared__ float red[4];
    const int p = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int d = tid; d < D; d += 256) {
        const float x = p < P ? dout[(size_t)p * D + d] : 0.f;
        if (p < P) acc += x * out[(size_t)p * D + d];
        const __bf16 h0 = (__bf16)x;
        const float r1 = x - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const __bf16 h2 = (__bf16)(r1 - (float)h1);
        dsplit[((size_t)0 * 16 + p) * D + d] = h0;
        dsplit[((size_t)1 * 16 + p) * D + d] = h1;
        dsplit[((size_t)2 * 16 + p) * D + d] = h2;
    }
    acc = block_sum_256(acc, red);
    if (tid == 0) delta[p] = acc;
}

constexpr int kBwdExchWave = 4 * 64 * 16 + 32 * 4;  // S and dA partials (2 x f32x4 each per lane) + 32 row sumsq
constexpr int kBwdExchParity = 4 * kBwdExchWave;
template <bool F32>
constexpr int bwd_lds_bytes() {
    return 4 * kSliceBytes * (F32 ? 2 : 1) + 2 * kBwdExchParity;
}

// Several bags per launch (grid (G, B); fp32 bags or P > 12, which the persistent batch kernel of vlfan_backward_batch.hip does
// not take): bags != null -> workgroup (b, bag) works on rows block b of G of bag blockIdx.y, with that bag's upstream
// gradient fragments dsplit + bag * 3 * 16 * D, (m2, l, delta) + bag * 16, and writes partial bag * G + b.
struct BwdBag {
    const void* X;
    long long N, ldx;
};
template <typename XT>
__global__ __launch_bounds__(256, 1) void k_vlfan_backward_mfma(const XT* __restrict__ X, int64_t N, int64_t ldx,
                                                                 const __bf16* __restrict__ qsplit,
                                                                 const __bf16* __restrict__ dsplit, int P,
                                                                 const float* __restrict__ m2, const float* __restrict__ l,
                                                                 const float* __restrict__ delta, float scale,
                                                                 float* __restrict__ pm, float* __restrict__ pl,
                                                                 float* __restrict__ pacc, int G,
                                                                 const BwdBag* __restrict__ bags) {
    constexpr bool F32 = sizeof(XT) == 4;
    constexpr int NX = F32 ? 2 : 1;
    constexpr int D = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    int b = blockIdx.x;
    if (bags != nullptr) {
        const int bag = blockIdx.y;
        const BwdBag d = bags[bag];
        X = static_cast<const XT*>(d.X);
        N = d.N;
        ldx = d.ldx;
        dsplit += (size_t)bag * 3 * 16 * 512;
        m2 += (size_t)bag * kPStride;
        l += (size_t)bag * kPStride;
        delta += (size_t)bag * kPStride;
    }
    int64_t rbeg, rend;
    block_rows(N, b, G, rbeg, rend);
    b += blockIdx.y * G;   // partial slot

    unsigned char* xs = smem + (size_t)w * kSliceBytes * NX;
    unsigned char* exch = smem + (size_t)4 * kSliceBytes * NX;

    bf16x8 qf[3][4], df[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const size_t off = ((size_t)t * 16 + i16) * D + w * 128 + kk * 32 + g * 8;
            qf[t][kk] = *reinterpret_cast<const bf16x8*>(qsplit + off);
            df[t][kk] = *reinterpret_cast<const bf16x8*>(dsplit + off);
        }
    const bool pok = i16 < P;
    const float m2p = pok ? m2[i16] : 0.f;
    const float rlp = pok ? 1.f / l[i16] : 0.f;
    const float dlt = pok ? delta[i16] : 0.f;

    f32x4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 st[StageN<XT>::value];
    const int64_t rlast = rend > rbeg ? rend - 1 : (N > 0 ? N - 1 : 0);
    if (N > 0) stage_load(st, X, ldx, rbeg, rlast, w, lane);
    int par = 0;
    for (int64_t r0 = rbeg; r0 < rend; r0 += kTileRows, par ^= 1) {
        stage_store(st, xs, lane);
        stage_load(st, X, ldx, r0 + kTileRows, rlast, w, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        f32x4 S[2], Dd[2];
        float ss[2];
        {
            bf16x8 xa[2][4], xl[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int off = swz(16 * h + i16, kk * 64 + g * 16);
                    xa[h][kk] = *reinterpret_cast<const bf16x8_ma*>(xs + off);
                    if constexpr (F32) xl[h][kk] = *reinterpret_cast<const bf16x8_ma*>(xs + kSliceBytes + off);
                }
            f32x4 Sa[2], Sb[2], Da[2], Db[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Sa[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                Sb[h] = Sa[h];
                Da[h] = Sa[h];
                Db[h] = Sa[h];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    Sa[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[0][kk], Sa[h], 0, 0, 0);
                    Da[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], df[0][kk], Da[h], 0, 0, 0);
                    Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[1][kk], Sb[h], 0, 0, 0);
                    Db[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], df[1][kk], Db[h], 0, 0, 0);
                    Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[2][kk], Sb[h], 0, 0, 0);
                    Db[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], df[2][kk], Db[h], 0, 0, 0);
                    if constexpr (F32) {
                        Sa[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[h][kk], qf[0][kk], Sa[h], 0, 0, 0);
                        Da[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[h][kk], df[0][kk], Da[h], 0, 0, 0);
                        Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[h][kk], qf[1][kk], Sb[h], 0, 0, 0);
                        Db[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[h][kk], df[1][kk], Db[h], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float a = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    a = dot8(xa[h][kk], xa[h][kk], a);
                    if constexpr (F32) a = dot8(xl[h][kk], xl[h][kk], dot8(xa[h][kk], xl[h][kk], dot8(xa[h][kk], xl[h][kk], a)));
                }
                ss[h] = quad_rows_sum(a);
                S[h] = Sa[h] + Sb[h];
                Dd[h] = Da[h] + Db[h];
            }
        }
        {
            unsigned char* mine = exch + par * kBwdExchParity + w * kBwdExchWave;
            *reinterpret_cast<f32x4_ma*>(mine + (0 * 64 + lane) * 16) = S[0];
            *reinterpret_cast<f32x4_ma*>(mine + (1 * 64 + lane) * 16) = S[1];
            *reinterpret_cast<f32x4_ma*>(mine + (2 * 64 + lane) * 16) = Dd[0];
            *reinterpret_cast<f32x4_ma*>(mine + (3 * 64 + lane) * 16) = Dd[1];
            if (g == 0) {
                typedef float __attribute__((may_alias)) float_ma;
                reinterpret_cast<float_ma*>(mine + 4096)[i16] = ss[0];
                reinterpret_cast<float_ma*>(mine + 4096)[16 + i16] = ss[1];
            }
        }
        __syncthreads();
        f32x4 T[2], DA[2], R2[2];
        {
            f32x4 tv[2][4], dv[2][4], rv[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned char* o = exch + par * kBwdExchParity + ww * kBwdExchWave;
                    tv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + (h * 64 + lane) * 16);
                    dv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + ((2 + h) * 64 + lane) * 16);
                    rv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + 4096 + (16 * h + 4 * g) * 4);
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                T[h] = (tv[h][0] + tv[h][1]) + (tv[h][2] + tv[h][3]);
                DA[h] = (dv[h][0] + dv[h][1]) + (dv[h][2] + dv[h][3]);
                R2[h] = (rv[h][0] + rv[h][1]) + (rv[h][2] + rv[h][3]);
            }
        }
        bf16x8 ahi, alo;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = (r0 + 16 * h + 4 * g + r < rend) && pok;
                const float inv = fminf(__builtin_amdgcn_rsqf(R2[h][r]), 1e12f);
                const float A = fast_exp2(T[h][r] * inv - m2p) * rlp;
                const float u = valid ? A * (DA[h][r] - dlt) * (scale * inv) : 0.f;
                const __bf16 hi = (__bf16)u;
                ahi[4 * h + r] = hi;
                alo[4 * h + r] = (__bf16)(u - (float)hi);
            }
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const int c_off = ct * 32 + (i16 & 3) * 8;
            const int row0 = 4 * g + (i16 >> 2);
            const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + swz(row0, c_off)));
            const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + swz(16 + row0, c_off)));
            const bf16x8 bh = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bh, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bh, acc[ct], 0, 0, 0);
            if constexpr (F32) {
                const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + kSliceBytes + swz(row0, c_off)));
                const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + kSliceBytes + swz(16 + row0, c_off)));
                const bf16x8 bl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bl, acc[ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    if (w == 0 && g == 0 && i16 < P) {
        pm[(size_t)b * kPStride + i16] = 0.f;
        pl[(size_t)b * kPStride + i16] = 1.f;
    }
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 4 * g + r;
            if (p < P) pacc[((size_t)b * P + p) * D + w * 128 + ct * 16 + i16] = acc[ct][r];
        }
}

}  // namespace vlsa

using namespace vlsa;

extern "C" size_t vlsa_bwd_prep_bytes(int D) { return (size_t)3 * 16 * D * 2 + 64; }

extern "C" int vlsa_vlfan_backward(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* qprep, int P,
                                   float coattn_scale, const float* dout, const float* out, const float* m2,
                                   const float* l, void* bwd_prep, float* pm, float* pl, float* pacc, void* stream) {
    if (!qprep || !dout || !out || !m2 || !l || !bwd_prep || !pm || !pl || !pacc || N < 0 || (N > 0 && !X)) return VLSA_EINVAL;
    if (D != 512) return VLSA_EUNSUPPORTED;
    if (P < 1 || P > VLSA_MAX_P || ldx < D) return VLSA_EINVAL;
    if (x_dtype != VLSA_DT_F32 && x_dtype != VLSA_DT_BF16) return VLSA_EINVAL;
    const size_t esz = x_dtype == VLSA_DT_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(X) & 15) != 0 || ((size_t)ldx * esz) % 16 != 0) return VLSA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    __bf16* dsplit = static_cast<__bf16*>(bwd_prep);
    float* delta = reinterpret_cast<float*>(static_cast<unsigned char*>(bwd_prep) + (size_t)3 * 16 * D * 2);
    hipLaunchKernelGGL(k_prepare_backward, dim3(16), dim3(256), 0, s, dout, out, P, D, dsplit, delta);
    const int G = vlsa_num_partials(N);
    const QPrepLayout L(D);
    const __bf16* qsplit = reinterpret_cast<const __bf16*>(static_cast<const unsigned char*>(qprep) + L.qsplit);
    if (x_dtype == VLSA_DT_F32) {
        auto kern = k_vlfan_backward_mfma<float>;
        constexpr int lds = bwd_lds_bytes<true>();
        static DeviceOnce once;
        if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, s, (const float*)X, N, ldx, qsplit, dsplit, P, m2, l, delta,
                           coattn_scale, pm, pl, pacc, G, static_cast<const BwdBag*>(nullptr));
    } else {
        auto kern = k_vlfan_backward_mfma<__bf16>;
        constexpr int lds = bwd_lds_bytes<false>();
        static DeviceOnce once;
        if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, s, (const __bf16*)X, N, ldx, qsplit, dsplit, P, m2, l, delta,
                           coattn_scale, pm, pl, pacc, G, static_cast<const BwdBag*>(nullptr));
    }
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// The per-bag kernel over a table of bags in ONE launch (see BwdBag): dsplit [B][3][16][D] and delta [B][16] as produced by
// k_prepare_backward_batch (vlfan_backward_batch.hip); pm / pl [B * G, 16], pacc [B * G, P, D].
int vlsa_launch_backward_mfma_bags(const void* bag_desc, int B, int x_dtype, const __bf16* qsplit, const __bf16* dsplit, int P,
                                   const float* m2, const float* l, const float* delta, float scale, float* pm, float* pl,
                                   float* pacc, int G, hipStream_t s) {
    const BwdBag* bags = static_cast<const BwdBag*>(bag_desc);
    if (x_dtype == VLSA_DT_F32) {
        auto kern = k_vlfan_backward_mfma<float>;
        constexpr int lds = bwd_lds_bytes<true>();
        static DeviceOnce once;
        if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(G, B), dim3(256), lds, s, static_cast<const float*>(nullptr), (int64_t)0, (int64_t)0, qsplit, dsplit, P,
                           m2, l, delta, scale, pm, pl, pacc, G, bags);
    } else {
        auto kern = k_vlfan_backward_mfma<__bf16>;
        constexpr int lds = bwd_lds_bytes<false>();
        static DeviceOnce once;
        if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(G, B), dim3(256), lds, s, static_cast<const __bf16*>(nullptr), (int64_t)0, (int64_t)0, qsplit, dsplit, P,
                           m2, l, delta, scale, pm, pl, pacc, G, bags);
    }
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// On-device loss tail of the training step: SurvIFMLE (loss/loss_surv.py:127-169) and SurvEMD
// (loss/loss_surv_ext.py:43-109, cdf_loss 13-38) on the [B, K] bag predictions, forward AND gradient in one launch, from
// the raw logits (the handler's softmax converter, runner/vlsa_handler.py:241-258, fused in) or from incidences.
// The reference spends ~40 elementwise launches and a Python loop over the batch (convert_survival_label) on this.
// One thread per sample; K <= 64 values live in registers / scratch -- the work is a few hundred flops per sample.
#include "vlsa_common.h"

namespace vlsa {

// Per-sample K-vectors live in LDS, [vector][k][thread] (conflict free): as dynamically indexed private arrays they sat in scratch
// memory and the launch took 18.6 us for 32 samples x 12 bins (round 4: profiles/r04_step_kernel_stats.csv).  Same serial order per
// sample as before: results are bit-identical.
constexpr int kLossThreads = 32;
struct LossVec {
    float* p;
    __device__ __forceinline__ float& operator[](int k) const { return p[k * kLossThreads]; }
};
__global__ __launch_bounds__(kLossThreads) void k_surv_loss(const float* __restrict__ x, const int64_t* __restrict__ t_,
                                                   const float* __restrict__ e_, int B, int K, int from_logits,
                                                   const float* __restrict__ logit_scale_exp, float alpha, float eps, int p,
                                                   int raw_distance, float w_ifmle, float w_emd,
                                                   float* __restrict__ out_ifmle, float* __restrict__ out_emd,
                                                   float* __restrict__ grad, float* __restrict__ objective, int ls_is_log) {
    // objective != null (round 6, ONE block launched): the block walks the batch in chunks of 32 samples and also writes the scalar
    // objective[0] = mean_i (w_ifmle ifmle_i + w_emd emd_i) -- the handler's calc_objective_loss for 'mean'-reduced losses
    // (runner/vlsa_handler.py:241-258) -- and `grad` comes out scaled by 1 / B (the gradient of that mean): the five elementwise /
    // reduction launches behind the per-sample values were 25 us of a 1.7 ms optimizer step.  ls_is_log: logit_scale_exp points at the
    // RAW (log) logit scale parameter and is exponentiated here (saves the step's `logit_scale.exp()` launch).
    __shared__ float sm[5][VLSA_MAX_K][kLossThreads];
    float obj_acc = 0.f;
    const int n_chunks = objective != nullptr ? (B + kLossThreads - 1) / kLossThreads : 1;
    const float gscale = objective != nullptr ? 1.f / (float)B : 1.f;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
    const int i = (objective != nullptr ? chunk : (int)blockIdx.x) * kLossThreads + threadIdx.x;
    if (i >= B) continue;
    const LossVec inc{&sm[0][0][threadIdx.x]}, g{&sm[1][0][threadIdx.x]};
    const float* xi = x + (size_t)i * K;
    int t = (int)t_[i];
    t = t < 0 ? 0 : (t >= K ? K - 1 : t);
    const float e = e_[i];
    // ---- incidence: softmax over the K bins (utils/func.py:43-44) or the given values
    if (from_logits) {
        float mx = -INFINITY, s = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, xi[k]);
        for (int k = 0; k < K; ++k) { inc[k] = expf(xi[k] - mx); s += inc[k]; }
        for (int k = 0; k < K; ++k) inc[k] /= s;
    } else {
        for (int k = 0; k < K; ++k) inc[k] = xi[k];
    }
    for (int k = 0; k < K; ++k) g[k] = 0.f;

    // ---- SurvIFMLE: (1 - alpha) (censored + uncensored) + alpha uncensored, per sample
    float l_ifmle = 0.f;
    {
        const float c = 1.f - e;
        float cif = 0.f;
        for (int k = 0; k <= t; ++k) cif += inc[k];
        const float a = inc[t], sv = 1.f - cif;
        const float unc = -(1.f - c) * logf(fmaxf(a, eps));
        const float cen = -c * logf(fmaxf(sv, eps));
        l_ifmle = (1.f - alpha) * (cen + unc) + alpha * unc;
        if (w_ifmle != 0.f) {
            if (a >= eps) g[t] += w_ifmle * (-(1.f - c) / a);          // d unc / d inc[t]; clamp passes the gradient at >= eps
            if (sv >= eps) {
                const float gc = w_ifmle * (1.f - alpha) * c / sv;     // d cen / d inc[k], k <= t
                for (int k = 0; k <= t; ++k) g[k] += gc;
            }
        }
    }

    // ---- SurvEMD on y = incidence
    float l_emd = 0.f;
    if (w_emd != 0.f || out_emd != nullptr) {
        const float ls = ls_is_log ? expf(logit_scale_exp[0]) : logit_scale_exp[0];
        const int ei = (int)e;  // e.long() of the reference
        const LossVec pd{&sm[2][0][threadIdx.x]}, td{&sm[3][0][threadIdx.x]}, dp{&sm[4][0][threadIdx.x]};
        float mp = -INFINITY, mt = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const float tg = (k == t ? 1.f : 0.f) + (k > t ? (float)(1 - ei) : 0.f);   // convert_survival_label
            td[k] = (2.f * tg - 1.f) * ls;
            pd[k] = (float)(1 - ei) * ((1.f - tg) * inc[k] + tg * ls) + (float)ei * inc[k];
            dp[k] = (float)(1 - ei) * (1.f - tg) + (float)ei;                               // d pred / d y
            mp = fmaxf(mp, pd[k]);
            mt = fmaxf(mt, td[k]);
        }
        float sp = 0.f, st = 0.f;
        for (int k = 0; k < K; ++k) { pd[k] = expf(pd[k] - mp); sp += pd[k]; td[k] = expf(td[k] - mt); st += td[k]; }
        float cp = 0.f, ct = 0.f, S = 0.f;
        for (int k = 0; k < K; ++k) {
            pd[k] /= sp;
            cp += pd[k];
            ct += td[k] / st;
            const float d = cp - ct;
            td[k] = d;                                   // td now holds the cdf difference
            S += (p == 1) ? fabsf(d) : d * d;
        }
        const bool root = (p == 2 && !raw_distance);
        l_emd = root ? sqrtf(S) : S;
        if (w_emd != 0.f) {
            const float outer = root ? (S > 0.f ? 0.5f / sqrtf(S) : 0.f) : 1.f;
            // r_j = d dist / d pd_j = suffix sum of d S / d d_k
            float r = 0.f, dot = 0.f;
            for (int k = K - 1; k >= 0; --k) {
                const float d = td[k];
                r += (p == 1) ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
                td[k] = r;
                dot += pd[k] * r;
            }
            for (int k = 0; k < K; ++k) g[k] += w_emd * outer * pd[k] * (td[k] - dot) * dp[k];
        }
    }

    if (out_ifmle != nullptr) out_ifmle[i] = l_ifmle;
    if (out_emd != nullptr) out_emd[i] = l_emd;
    obj_acc += w_ifmle * l_ifmle + w_emd * l_emd;
    if (grad != nullptr) {
        float* gi = grad + (size_t)i * K;
        if (from_logits) {  // softmax backward to the raw logits
            float dot = 0.f;
            for (int k = 0; k < K; ++k) dot += inc[k] * g[k];
            for (int k = 0; k < K; ++k) gi[k] = gscale * (inc[k] * (g[k] - dot));
        } else {
            for (int k = 0; k < K; ++k) gi[k] = gscale * g[k];
        }
    }
    }   // chunk
    if (objective != nullptr) {
        // fixed-order sum over the 32 lanes (lanes without a sample hold 0): deterministic
        float s = obj_acc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
        if (threadIdx.x == 0) objective[0] = s / (float)B;
    }
}


// (Round 6, tried and removed: the same arithmetic with the per-sample vectors in registers and every bin loop unrolled to 16 -- 14.1 us
//  against 12.5 us for the LDS loops above inside the training step: one wave running ~20 KB of straight-line code once is bound by
//  instruction fetch, not by the LDS round trips.  Parking the sample's K inputs in LDS with one batch of loads first: no change either,
//  12.7 -> 13.5 us on another box -- the time is the ~500 dependent LDS accesses of the bin loops.)

// K <= 16 bins (the reference's 4 .. 12): a sample per 16-LANE ROW, a bin per lane -- every loop over the bins above becomes a row
// operation through DPP (sum / max by row rotations, the cumulative sums of the EMD term by shifted adds), 32 samples per 512-thread
// workgroup and pass.  Same formulas; the cumulative sums keep the bin order (see row16_prefix), plain sums run in tree order.  Inside the training
// step: 12.7 -> 4.7 us = the floor of a launch (the loops above are ~500 dependent LDS accesses on one wave).
#define VLSA_DPP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
__device__ __forceinline__ float row16_sum(float v) {
    v += VLSA_DPP(v, 0x128); v += VLSA_DPP(v, 0x124); v += VLSA_DPP(v, 0x122); v += VLSA_DPP(v, 0x121);      // row_ror 8, 4, 2, 1
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, VLSA_DPP(v, 0x128)); v = fmaxf(v, VLSA_DPP(v, 0x124)); v = fmaxf(v, VLSA_DPP(v, 0x122)); v = fmaxf(v, VLSA_DPP(v, 0x121));
    return v;
}
// Cumulative sums IN BIN ORDER, as torch.cumsum and its backward add them: c_k = (((v_0 + v_1) + v_2) + ...) + v_k, fifteen dependent
// shifted adds (after step s lane k holds the left-to-right sum of v_{k-s} .. v_k).  The order matters: the reference clamps
// 1 - CIF[t] at eps, and whether a censored sample in the last bin lands on the clamp -- zero gradient or c / sv ~ 1e7 -- is decided by the
// last ulp of that cumsum (a log-step scan moved a 40-step training run off the CPU twin's loss curve by 0.3 % at step 6).
__device__ __forceinline__ float row16_prefix(float v) {      // inclusive, over the lanes <= this one (row_shr:1, zeros shift in)
    float c = v;
#pragma unroll
    for (int s = 0; s < 15; ++s) c = VLSA_DPP(c, 0x111) + v;
    return c;
}
__device__ __forceinline__ float row16_suffix(float v) {      // inclusive, over the lanes >= this one, added from the right (row_shl:1)
    float c = v;
#pragma unroll
    for (int s = 0; s < 15; ++s) c = VLSA_DPP(c, 0x101) + v;
    return c;
}
#undef VLSA_DPP
constexpr int kLossWide = 512, kLossWideSamples = kLossWide / 16;
__global__ __launch_bounds__(kLossWide) void k_surv_loss_rows16(const float* __restrict__ x, const int64_t* __restrict__ t_,
                                                          const float* __restrict__ e_, int B, int K, int from_logits,
                                                          const float* __restrict__ logit_scale_exp, float alpha, float eps, int p,
                                                          int raw_distance, float w_ifmle, float w_emd,
                                                          float* __restrict__ out_ifmle, float* __restrict__ out_emd,
                                                          float* __restrict__ grad, float* __restrict__ objective, int ls_is_log) {
    __shared__ float sval[kLossWideSamples];
    const int tid = threadIdx.x, k = tid & 15, sub = tid >> 4;
    const bool obj = objective != nullptr, kk = k < K;
    const int n_chunks = obj ? (B + kLossWideSamples - 1) / kLossWideSamples : 1;
    const float gscale = obj ? 1.f / (float)B : 1.f;
    const bool need_emd = w_emd != 0.f || out_emd != nullptr;
    float ls = 0.f;
    if (need_emd) ls = ls_is_log ? expf(logit_scale_exp[0]) : logit_scale_exp[0];
    float obj_acc = 0.f;                      // threads 0 .. 31: the samples chunk * 32 + tid, as the one-sample-per-thread kernel adds them
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int i = (obj ? chunk : (int)blockIdx.x) * kLossWideSamples + sub;
        const bool valid = i < B;             // (rows without a sample run on zeros: every lane takes part in the row operations)
        const float xv = (valid && kk) ? x[(size_t)i * K + k] : 0.f;
        int t = valid ? (int)t_[i] : 0;
        t = t < 0 ? 0 : (t >= K ? K - 1 : t);
        const float e = valid ? e_[i] : 0.f;
        float inc;
        if (from_logits) {
            const float mx = row16_max(kk ? xv : -INFINITY);
            const float ex = kk ? expf(xv - mx) : 0.f;
            inc = ex / row16_sum(ex);
        } else {
            inc = kk ? xv : 0.f;
        }
        float g = 0.f;
        float l_ifmle;
        {
            const float c = 1.f - e;
            // (row operations are evaluated by EVERY lane, outside any lane-dependent condition: a lane that sits out does not lend its value)
            const float cum = row16_prefix(inc);
            const float cif = row16_sum(k == t ? cum : 0.f);      // CIF[t]: the cumulative sum in bin order, read at bin t
            const float a = row16_sum(k == t ? inc : 0.f), sv = 1.f - cif;
            const float unc = -(1.f - c) * logf(fmaxf(a, eps));
            const float cen = -c * logf(fmaxf(sv, eps));
            l_ifmle = (1.f - alpha) * (cen + unc) + alpha * unc;
            if (w_ifmle != 0.f) {
                if (a >= eps && k == t) g += w_ifmle * (-(1.f - c) / a);
                if (sv >= eps && k <= t) g += w_ifmle * (1.f - alpha) * c / sv;
            }
        }
        float l_emd = 0.f;
        if (need_emd) {
            const int ei = (int)e;
            const float tg = (k == t ? 1.f : 0.f) + (k > t ? (float)(1 - ei) : 0.f);       // convert_survival_label
            const float tdv = kk ? (2.f * tg - 1.f) * ls : -INFINITY;
            const float pdv = kk ? (float)(1 - ei) * ((1.f - tg) * inc + tg * ls) + (float)ei * inc : -INFINITY;
            const float dpv = (float)(1 - ei) * (1.f - tg) + (float)ei;
            const float mp = row16_max(pdv), mt = row16_max(tdv);
            const float pe = kk ? expf(pdv - mp) : 0.f, te = kk ? expf(tdv - mt) : 0.f;
            const float pn = pe / row16_sum(pe), tn = te / row16_sum(te);
            const float cp = row16_prefix(pn), ct = row16_prefix(tn);
            const float d = kk ? cp - ct : 0.f;
            const float S = row16_sum((p == 1) ? fabsf(d) : d * d);
            const bool root = (p == 2 && !raw_distance);
            l_emd = root ? sqrtf(S) : S;
            if (w_emd != 0.f) {
                const float outer = root ? (S > 0.f ? 0.5f / sqrtf(S) : 0.f) : 1.f;
                const float r = row16_suffix((p == 1) ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d);
                const float dot = row16_sum(pn * r);
                g += w_emd * outer * pn * (r - dot) * dpv;
            }
        }
        if (valid && k == 0) {
            if (out_ifmle != nullptr) out_ifmle[i] = l_ifmle;
            if (out_emd != nullptr) out_emd[i] = l_emd;
        }
        if (grad != nullptr) {
            float gi = gscale * g;
            const float ig = row16_sum(inc * g);
            if (from_logits) gi = gscale * (inc * (g - ig));      // softmax backward to the raw logits
            if (valid && kk) grad[(size_t)i * K + k] = gi;
        }
        if (obj) {
            if (k == 0) sval[sub] = valid ? w_ifmle * l_ifmle + w_emd * l_emd : 0.f;
            __syncthreads();
            if (tid < kLossWideSamples) obj_acc += sval[tid];
            __syncthreads();
        }
    }
    if (obj && tid < 32) {
        float s = obj_acc;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
        if (tid == 0) objective[0] = s / (float)B;
    }
}

}  // namespace vlsa

using namespace vlsa;

extern "C" int vlsa_surv_loss(const float* x, const int64_t* t, const float* e, int B, int K, int from_logits,
                              const float* logit_scale_exp, float alpha, float eps, int p, int raw_distance, float w_ifmle,
                              float w_emd, float* out_ifmle, float* out_emd, float* grad, void* stream) {
    if (!x || !t || !e || B < 1 || K < 1 || K > VLSA_MAX_K) return VLSA_EINVAL;
    if ((w_emd != 0.f || out_emd) && !logit_scale_exp) return VLSA_EINVAL;
    if (p != 1 && p != 2) return VLSA_EUNSUPPORTED;
    if (K <= 16)
        hipLaunchKernelGGL(k_surv_loss_rows16, dim3((B + kLossWideSamples - 1) / kLossWideSamples), dim3(kLossWide), 0, (hipStream_t)stream, x, t, e, B, K,
                           from_logits, logit_scale_exp, alpha, eps, p, raw_distance, w_ifmle, w_emd, out_ifmle, out_emd, grad, (float*)nullptr, 0);
    else
        hipLaunchKernelGGL(k_surv_loss, dim3((B + kLossThreads - 1) / kLossThreads), dim3(kLossThreads), 0, (hipStream_t)stream, x, t, e, B, K, from_logits, logit_scale_exp,
                           alpha, eps, p, raw_distance, w_ifmle, w_emd, out_ifmle, out_emd, grad, (float*)nullptr, 0);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// The handler's whole objective (runner/vlsa_handler.py:241-258 with 'mean'-reduced SurvIFMLE + SurvEMD) from the raw logits in ONE
// launch: objective[0] = mean_i (w_ifmle ifmle_i + w_emd emd_i), grad [B, K] = d objective / d x.  logit_scale: the EXPONENTIATED scale
// (ls_is_log == 0, the reference's `net.get_logit_scale()`) or the raw parameter (ls_is_log != 0).  B <= 4096 (one block walks the batch).
extern "C" int vlsa_surv_objective(const float* x, const int64_t* t, const float* e, int B, int K, int from_logits, const float* logit_scale,
                                   int ls_is_log, float alpha, float eps, int p, int raw_distance, float w_ifmle, float w_emd,
                                   float* objective, float* grad, void* stream) {
    if (!x || !t || !e || !objective || B < 1 || B > 4096 || K < 1 || K > VLSA_MAX_K) return VLSA_EINVAL;
    if (w_emd != 0.f && !logit_scale) return VLSA_EINVAL;
    if (p != 1 && p != 2) return VLSA_EUNSUPPORTED;
    if (K <= 16)
        hipLaunchKernelGGL(k_surv_loss_rows16, dim3(1), dim3(kLossWide), 0, (hipStream_t)stream, x, t, e, B, K, from_logits, logit_scale, alpha, eps, p,
                           raw_distance, w_ifmle, w_emd, (float*)nullptr, (float*)nullptr, grad, objective, ls_is_log);
    else
        hipLaunchKernelGGL(k_surv_loss, dim3(1), dim3(kLossThreads), 0, (hipStream_t)stream, x, t, e, B, K, from_logits, logit_scale, alpha, eps, p,
                           raw_distance, w_ifmle, w_emd, (float*)nullptr, (float*)nullptr, grad, objective, ls_is_log);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

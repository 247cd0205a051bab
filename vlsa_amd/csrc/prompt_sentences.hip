// Sentence assembly of the CoOp prompt learners in embedding space: reference
// model/prompt_learners/rank_prompt_learner.py:100-156 (RankPromptLearner.forward: rank embeddings interpolated from the base
// ranks, context + rank tokens placed behind <sot>, "." and <eot> closing the sentence, pad embeddings behind) and
// plain_prompt_learner.py (one embedding row per rank, no interpolation).  One launch forward, one backward.
//   out[i, p, :] = template[i, p, :]                          for positions without a learnable token
//                = context[(i,) o, :]                         slot p - 1 of sentence i takes source o = order[i, p - 1] < C
//                = sum_b interp[i, b] rank[b, o - C, :]       ... or o >= C (interp NULL: rank[i, o - C, :])
// backward: d context[(i,) o] = sum_i d out[i, pos_i(o)],  d rank[b, t] = sum_i interp[i, b] d out[i, pos_i(C + t)] with the
// inverse map pos (deterministic: one workgroup per gradient row, ranks summed in order).
#include "vlsa_common.h"

namespace vlsa {

// grid (L, R), 256 threads; order: [R, S] int32 (S = C + T slots; -1 = unused slot)
__global__ __launch_bounds__(256) void k_prompt_sentences(const float* __restrict__ templ, const float* __restrict__ context,
                                                           int ctx_per_rank, const float* __restrict__ rank, const float* __restrict__ interp,
                                                           int n_base, const int* __restrict__ order, int R, int L, int S, int C, int T,
                                                           int dim, float* __restrict__ out) {
    const int p = blockIdx.x, i = blockIdx.y, tid = threadIdx.x;
    float* o = out + ((size_t)i * L + p) * dim;
    const int slot = p - 1;
    const int src = (slot >= 0 && slot < S) ? order[i * S + slot] : -1;
    if (src < 0) {
        const float* t = templ + ((size_t)i * L + p) * dim;
        for (int c = tid; c < dim; c += 256) o[c] = t[c];
    } else if (src < C) {
        const float* x = context + ((size_t)(ctx_per_rank ? i : 0) * C + src) * dim;
        for (int c = tid; c < dim; c += 256) o[c] = x[c];
    } else if (interp == nullptr) {
        const float* x = rank + ((size_t)i * T + (src - C)) * dim;
        for (int c = tid; c < dim; c += 256) o[c] = x[c];
    } else {
        for (int c = tid; c < dim; c += 256) {
            float s = 0.f;
            for (int b = 0; b < n_base; ++b) s = fmaf(interp[i * n_base + b], rank[((size_t)b * T + (src - C)) * dim + c], s);
            o[c] = s;
        }
    }
}

// pos: [R, S] int32: position (1-based in the sentence) of source o in sentence i, or -1.
// grid (rows of d context + rows of d rank): block b < n_ctx_rows: context row; else rank row.
__global__ __launch_bounds__(256) void k_prompt_sentences_bwd(const float* __restrict__ dout, const int* __restrict__ pos,
                                                               int ctx_per_rank, const float* __restrict__ interp, int n_base, int n_rank_rows,
                                                               int R, int L, int S, int C, int T, int dim, float* __restrict__ dcontext,
                                                               float* __restrict__ drank) {
    const int tid = threadIdx.x;
    const int n_ctx_rows = (ctx_per_rank ? R : 1) * C;
    int blk = blockIdx.x;
    if (blk < n_ctx_rows) {
        const int i0 = ctx_per_rank ? blk / C : -1, o = blk % C;
        // the sentences' positions first, then their rows 16 at a time (position -> row -> add, one sentence after the other, was a chain
        // of 2 R dependent loads: 14.7 us at R = 12 in the training step); same order of additions
        const int ibeg = i0 < 0 ? 0 : i0, iend = i0 < 0 ? R : i0 + 1;
        for (int c = tid; c < dim; c += 256) {
            float s = 0.f;
            for (int i1 = ibeg; i1 < iend; i1 += 16) {
                int pp[16];
                float dv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pp[u] = i1 + u < iend ? pos[(i1 + u) * S + o] : -1;
#pragma unroll
                for (int u = 0; u < 16; ++u) dv[u] = pp[u] >= 0 ? dout[((size_t)(i1 + u) * L + pp[u]) * dim + c] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (pp[u] >= 0) s += dv[u];
            }
            dcontext[(size_t)blk * dim + c] = s;
        }
        return;
    }
    blk -= n_ctx_rows;                       // rank row (b, t)
    const int b = blk / T, t = blk % T;
    for (int c = tid; c < dim; c += 256) {
        float s = 0.f;
        if (interp == nullptr) {
            const int p = pos[b * S + C + t];
            if (p >= 0) s = dout[((size_t)b * L + p) * dim + c];
        } else {
            for (int i1 = 0; i1 < R; i1 += 16) {
                int pp[16];
                float dv[16], wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    pp[u] = i1 + u < R ? pos[(i1 + u) * S + C + t] : -1;
                    wv[u] = i1 + u < R ? interp[(i1 + u) * n_base + b] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) dv[u] = pp[u] >= 0 ? dout[((size_t)(i1 + u) * L + pp[u]) * dim + c] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (pp[u] >= 0) s = fmaf(wv[u], dv[u], s);
            }
        }
        drank[(size_t)blk * dim + c] = s;
    }
}

}  // namespace vlsa

using namespace vlsa;

extern "C" int vlsa_prompt_sentences(const float* templ, const float* context, int ctx_per_rank, const float* rank, const float* interp,
                                     int n_base, const int* order, int R, int L, int S, int C, int T, int dim, float* out, void* stream) {
    if (!templ || !context || !rank || !order || !out || R < 1 || L < 2 || S != C + T || C < 0 || T < 1 || dim < 1) return VLSA_EINVAL;
    if (interp && n_base < 1) return VLSA_EINVAL;
    hipLaunchKernelGGL(k_prompt_sentences, dim3(L, R), dim3(256), 0, (hipStream_t)stream, templ, context, ctx_per_rank, rank, interp, n_base,
                       order, R, L, S, C, T, dim, out);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

extern "C" int vlsa_prompt_sentences_backward(const float* dout, const int* pos, int ctx_per_rank, const float* interp, int n_base,
                                              int n_rank_rows, int R, int L, int S, int C, int T, int dim, float* dcontext, float* drank,
                                              void* stream) {
    if (!dout || !pos || !dcontext || !drank || R < 1 || L < 2 || S != C + T || T < 1 || dim < 1 || n_rank_rows < 1) return VLSA_EINVAL;
    const int blocks = (ctx_per_rank ? R : 1) * C + n_rank_rows * T;
    hipLaunchKernelGGL(k_prompt_sentences_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, pos, ctx_per_rank, interp, n_base,
                       n_rank_rows, R, L, S, C, T, dim, dcontext, drank);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

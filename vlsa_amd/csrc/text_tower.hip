// Text side of the path: the frozen CoCa text tower that turns the K ordinal rank prompts into the text
// features [K, 512] -- reference model/prompt_encoder.py:267-322 (CONCHPromptEncoder.forward) over
// model/conch/transformer.py:191-247,290-322 (pre-LN blocks: nn.MultiheadAttention + GELU MLP, no dropout).  The reference
// runs it on K x 128 positions for every bag (1.4 s on the CPU); here it runs once per optimizer step / per
// checkpoint on the rows that can reach the pooled CLS token at all:
//
//   * the causal mask lets position t see only positions <= t, and the CLS row (appended last) sees column 0 and column
//     j + 1 for every non-pad token j (build_cls_mask pads its mask on the LEFT: prompt_encoder.py:245-252) -- for a sentence
//     of n tokens that is positions 0..n, i.e. the sentence plus the FIRST pad position, and not itself.  Rows behind that
//     never influence the output, so a prompt costs n + 2 rows instead of 128 (13 for the shipped rank prompts: exact,
//     not an approximation -- tests/test_text_oracle_golden.py::test_rows_behind_the_sentence...).
//   * the rows of all prompts are packed back to back ("compact rows", M = sum (n_s + 2), padded to a multiple of 48);
//     every GEMM of the tower is then a skinny [M, K] x [K, N] product that streams each weight exactly once from HBM.
//
// All contractions run on the f32 matrix pipe (v_mfma_f32_16x16x4_f32: f32 in / f32 accumulate = an fmaf chain): the text
// features are compared with the fp32 reference at 1e-4 after 12 layers and feed logits scaled by exp(logit_scale) ~ 56,
// so a split-bf16 scheme would have to carry 3 terms per operand and buys nothing at this size (the tower is launch- and
// weight-stream-bound: 340 MB of fp32 weights, 96 kernels).
//
// Kernels
//   k_tt_embed            compact rows <- prompts_embedding[seq, pos] (or cls_emb) + positional_embedding[pos]
//   k_tt_gemm_nt<MT,NW>   Y = pro(A) W^T (+ bias) (+ GELU) (+ residual); W [N, K] as nn.Linear stores it.  A workgroup owns
//                         16 MT rows x 32 columns; its NW waves split K and are reduced through LDS (deterministic, no
//                         atomics).  pro = LayerNorm fused into the A-operand load (the wave's whole A slab sits in
//                         registers; row statistics by a two-pass reduction across the waves), or identity.
//   k_tt_gemm_nn<MT,NW>   Y = A W (contraction along W's rows: the input-gradient products of backward, and the final
//                         text projection); a workgroup owns 16 MT rows x 64 columns, 16-byte weight loads.
//   k_tt_attn_fwd/bwd     per (prompt, head) attention over the compact rows: causal for token rows, explicit key list for
//                         the CLS row; <= 128 rows forward, <= 64 rows backward.
//   k_tt_ln_bwd, k_tt_lnf_fwd/bwd, k_tt_scatter   LayerNorm backward (+ residual), ln_final on the CLS rows, d prompts_embedding.
// Backward is w.r.t. the prompt embeddings only: the tower is frozen in every shipped configuration
// (vlsa_txt_encoder_frozen: True, cfg_vlsa_conch.yaml:69; runner/vlsa_handler.py:131).
#include "vlsa_common.h"

namespace vlsa {
namespace tt {

constexpr float kLnEps = 1e-5f;          // nn.LayerNorm default (model/conch/coca_model.py:110: norm_layer = nn.LayerNorm)
constexpr int kHeadDim = 64;

__device__ __forceinline__ float gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tt_embed(float* __restrict__ x, int d, const float* __restrict__ emb, int64_t s_seq,
                                                 int64_t s_tok, const int* __restrict__ row_seq, const int* __restrict__ row_pos,
                                                 const int* __restrict__ row_src, const float* __restrict__ pos_emb,
                                                 const float* __restrict__ cls_emb, int M) {
    const int row = blockIdx.x;
    float* xr = x + (size_t)row * d;
    if (row >= M) {
        for (int c = threadIdx.x * 4; c < d; c += 1024) *reinterpret_cast<f32x4*>(xr + c) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int src = row_src[row];
    const float* e = src >= 0 ? emb + (size_t)row_seq[row] * s_seq + (size_t)src * s_tok : cls_emb;
    const float* pe = pos_emb + (size_t)row_pos[row] * d;
    for (int c = threadIdx.x * 4; c < d; c += 1024) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = e[c + i] + pe[c + i];     // e may be a strided view: scalar loads
        *reinterpret_cast<f32x4*>(xr + c) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Y[m, n] = sum_k pro(A)[m, k] W[n, k]  (+ bias[n]) (gelu) (+ resid[m, n]).
// MFMA v_mfma_f32_16x16x4_f32: first operand lane (i = l & 15, kslot = l >> 4) = A[i][k], second operand lane (j = l & 15,
// kslot) = B[k][j], result lane (j, g = l >> 4) holds D[4 g + v][j], v = 0..3.  One 16-byte load gives a lane the operands of
// FOUR consecutive MFMA steps: k-slot g of step s contracts column 16 jj + 4 g + s -- the same permutation on both operands.
enum { PRO_NONE = 0, PRO_LN = 1 };
enum { EPI_BIAS = 1, EPI_RESID = 2, EPI_GELU = 4 };
constexpr int kSlabMax = 12;   // PRO_LN: (K / NW) / 16 groups of the A slab kept in registers (K <= 768 at NW = 4)

// GT = number of 16-column groups per wave as a compile-time constant (0: runtime).  With GT known every loop below unrolls
// into straight-line code and the compiler's s_waitcnt pass can keep the prefetched loads in flight; with a runtime trip
// count it parks the ring behind `s_waitcnt vmcnt(0)` + register moves at every basic-block edge (measured: 36 us instead
// of 5 us for the fc2 product).  The CONCH sizes are instantiated, anything else takes the runtime path.
template <int MT, int NW, int PRO, int GT>
__global__ __launch_bounds__(NW * 64) void k_tt_gemm_nt(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const float* __restrict__ resid, int ldr,
                                                        float* __restrict__ Y, int ldy, float* __restrict__ Ypre, int N, int K,
                                                        int MG, int xcd_map, int epi, const float* __restrict__ ln_w,
                                                        const float* __restrict__ ln_b) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr int Q = MT * 8;              // accumulator registers per lane
    constexpr int QW = Q / NW;             // ... reduced and stored by each wave
    static_assert(Q % NW == 0, "accumulators must split evenly over the waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    int ntile, mg;
    {
        const int b = blockIdx.x;
        if (xcd_map) {   // the MG workgroups that share a weight tile run on one XCD (block b -> XCD b % 8): W comes from ITS L2
            const int j = b >> 3;
            ntile = (j / MG) * 8 + (b & 7);
            mg = j % MG;
        } else {
            ntile = b / MG;
            mg = b % MG;
        }
    }
    const int n0 = ntile * 32, m0 = mg * (16 * MT);
    const int KW = K / NW, kbeg = w * KW;
    const int G = GT > 0 ? GT : (KW >> 4);
    const float* Ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) Ap[t] = A + (size_t)(m0 + 16 * t + r) * lda + kbeg + 4 * g;
    const float* Wp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) Wp[u] = W + (size_t)(n0 + 16 * u + r) * K + kbeg + 4 * g;
#ifdef VLSA_TT_DEBUG
    if (epi & 512) {    // experiment: every workgroup reads weight tile 0 (stays in L2) instead of its own
#pragma unroll
        for (int u = 0; u < 2; ++u) Wp[u] = W + (size_t)(16 * u + r) * K + kbeg + 4 * g;
    }
    if (epi & 1024) {   // experiment: every workgroup reads A rows 0..15
#pragma unroll
        for (int t = 0; t < MT; ++t) Ap[t] = A + (size_t)r * lda + kbeg + 4 * g;
    }
#endif

    f32x4 acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (PRO == PRO_LN) {
        // ---- LayerNorm fused into the operand load: this wave's [16 MT rows] x [KW columns] slab of A in registers --------
        // the LayerNorm affine parameters go through LDS (first loads issued, so they are the first to land; reading them back
        // later is an LDS access and does not queue behind the weight stream the way a global load would: vmcnt is in-order)
        float* sgam = red + 1024;           // [K] gamma | [K] beta   (K <= 768: 6 KB of the 24 KB reduction buffer)
        f32x4 gld = {0.f, 0.f, 0.f, 0.f}, bld = {0.f, 0.f, 0.f, 0.f};
        if (tid * 4 < K) {
            gld = *reinterpret_cast<const f32x4*>(ln_w + tid * 4);
            bld = *reinterpret_cast<const f32x4*>(ln_b + tid * 4);
        }
        f32x4 slab[MT][kSlabMax];
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
#pragma unroll
                for (int t = 0; t < MT; ++t) slab[t][jj] = *reinterpret_cast<const f32x4*>(Ap[t] + 16 * jj);
            }
        // ... and ALL of this wave's weight groups behind them: the kernel is a latency chain (few waves per CU, every weight
        // read once from HBM), so everything the wave will ever need is put in flight before the first dependent use; the
        // LayerNorm statistics below overlap the weight fetch
        f32x4 wall[kSlabMax][2];
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
#pragma unroll
                for (int u = 0; u < 2; ++u) wall[jj][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 16 * jj);
            }
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler would otherwise sink the weight loads next to their MFMAs)
        if (tid * 4 < K) {
            *reinterpret_cast<f32x4*>(sgam + tid * 4) = gld;
            *reinterpret_cast<f32x4*>(sgam + K + tid * 4) = bld;
        }
        float* st = red;                    // [NW][16 MT] partial row statistics (the reduction buffer is free until the end)
        float mean[MT], rstd[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) s += (slab[t][jj][0] + slab[t][jj][1]) + (slab[t][jj][2] + slab[t][jj][3]);
            s = quad_rows_sum(s);
            if (g == 0) st[w * (16 * MT) + 16 * t + r] = s;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += st[ww * (16 * MT) + 16 * t + r];
            mean[t] = s / (float)K;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float c = slab[t][jj][i] - mean[t];
                        s = fmaf(c, c, s);
                    }
                }
            s = quad_rows_sum(s);
            if (g == 0) st[w * (16 * MT) + 16 * t + r] = s;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += st[ww * (16 * MT) + 16 * t + r];
            rstd[t] = 1.f / sqrtf(s / (float)K + kLnEps);
        }
        __syncthreads();                    // st is handed back to the final reduction
        // normalise the slab in place (affine parameters from LDS); the MFMA loop below then runs on registers only
        const float* gw = sgam + kbeg + 4 * g;
        const float* gb = sgam + K + kbeg + 4 * g;
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
                const f32x4 gam = *reinterpret_cast<const f32x4*>(gw + 16 * jj);
                const f32x4 bet = *reinterpret_cast<const f32x4*>(gb + 16 * jj);
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) slab[t][jj][i] = fmaf((slab[t][jj][i] - mean[t]) * rstd[t], gam[i], bet[i]);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(slab[t][jj][i], wall[jj][u][i], acc[t][u], 0, 0, 0);
            }
    } else {
        // register ring PF groups deep: the loads of group jj + PF are issued when group jj is consumed (vmcnt returns in order)
        constexpr int PF = MT == 1 ? 8 : 6;
        f32x4 ra[PF][MT], rb[PF][2];
#pragma unroll
        for (int sI = 0; sI < PF; ++sI)
            if (sI < G) {
#pragma unroll
                for (int t = 0; t < MT; ++t) ra[sI][t] = *reinterpret_cast<const f32x4*>(Ap[t] + 16 * sI);
#pragma unroll
                for (int u = 0; u < 2; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 16 * sI);
            }
        // keep the machine scheduler from sinking the prefetch loads next to their uses (it minimises register pressure and
        // would leave ~2 groups in flight): nothing moves across these fences
        __builtin_amdgcn_sched_barrier(0);
        auto stage = [&](int sI, int jj) __attribute__((always_inline)) {
            f32x4 a[MT], b[2];
#pragma unroll
            for (int t = 0; t < MT; ++t) a[t] = ra[sI][t];
#pragma unroll
            for (int u = 0; u < 2; ++u) b[u] = rb[sI][u];
            if (jj + PF < G) {
#pragma unroll
                for (int t = 0; t < MT; ++t) ra[sI][t] = *reinterpret_cast<const f32x4*>(Ap[t] + 16 * (jj + PF));
#pragma unroll
                for (int u = 0; u < 2; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 16 * (jj + PF));
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef VLSA_TT_DEBUG
            if (epi & 256) {   // experiment: no MFMAs, keep the loads alive with one add per loaded vector
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t][0] += a[t];
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[0][u] += b[u];
            } else
#endif
            {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][i], b[u][i], acc[t][u], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (GT > 0) {
#pragma unroll
            for (int jj = 0; jj < GT; ++jj) stage(jj % PF, jj);
        } else {
            for (int j0 = 0; j0 < G; j0 += PF) {
#pragma unroll
                for (int sI = 0; sI < PF; ++sI)
                    if (j0 + sI < G) stage(sI, j0 + sI);
            }
        }
    }

    // ---- reduce the NW K-slices through LDS (fixed order), then bias / GELU / residual and store -------------------------
    if constexpr (PRO == PRO_LN) __syncthreads();   // every wave is done reading gamma / beta from the buffer reused below
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[(w * Q + (t * 2 + u) * 4 + v) * 64 + lane] = acc[t][u][v];
    __syncthreads();
#pragma unroll
    for (int qi = 0; qi < QW; ++qi) {
        const int q = w * QW + qi;
        float val = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) val += red[(ww * Q + q) * 64 + lane];
        const int t = q >> 3, u = (q >> 2) & 1, v = q & 3;
        const int row = m0 + 16 * t + 4 * g + v, col = n0 + 16 * u + r;
        if (epi & EPI_BIAS) val += bias[col];
        if (epi & EPI_GELU) {
            if (Ypre) Ypre[(size_t)row * ldy + col] = val;
            val = gelu(val);
        }
        if (epi & EPI_RESID) val += resid[(size_t)row * ldr + col];
        Y[(size_t)row * ldy + col] = val;
    }
}

// Y[m, n] = sum_k A[m, k] W[k, n]  (W row-major [Kc, ldw]); optional epilogue Y *= gelu'(H[m, n]).
// A workgroup owns 16 MT rows x 64 columns: lane (c = l & 15, g) loads W[k][n0 + 4 c .. + 3] with one 16-byte load and feeds
// the four values to four column tiles (tile u holds column n0 + 4 c + u), k = 16 jj + 4 g + s for MFMA step s.
enum { EPN_GELU_BWD = 1 };
template <int MT, int NW, int GT>
__global__ __launch_bounds__(NW * 64) void k_tt_gemm_nn(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                        float* __restrict__ Y, int ldy, int Kc, int MG, int epi,
                                                        const float* __restrict__ H, int ldh) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr int Q = MT * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int ctile = blockIdx.x / MG, mg = blockIdx.x % MG;
    const int n0 = ctile * 64, m0 = mg * (16 * MT);
    const int KW = Kc / NW, kbeg = w * KW;
    const int G = GT > 0 ? GT : (KW >> 4);
    const float* Ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) Ap[t] = A + (size_t)(m0 + 16 * t + c) * lda + kbeg + 4 * g;
    const float* Wp = W + (size_t)(kbeg + 4 * g) * ldw + n0 + 4 * c;

    f32x4 acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PF = MT == 1 ? 6 : 4;     // register ring, as in k_tt_gemm_nt
    f32x4 ra[PF][MT], rb[PF][4];
#pragma unroll
    for (int sI = 0; sI < PF; ++sI)
        if (sI < G) {
#pragma unroll
            for (int t = 0; t < MT; ++t) ra[sI][t] = *reinterpret_cast<const f32x4*>(Ap[t] + 16 * sI);
#pragma unroll
            for (int s = 0; s < 4; ++s) rb[sI][s] = *reinterpret_cast<const f32x4*>(Wp + (size_t)(16 * sI + s) * ldw);
        }
    __builtin_amdgcn_sched_barrier(0);
    auto stage = [&](int sI, int jj) __attribute__((always_inline)) {
        f32x4 a[MT], b[4];
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t] = ra[sI][t];
#pragma unroll
        for (int s = 0; s < 4; ++s) b[s] = rb[sI][s];
        if (jj + PF < G) {
#pragma unroll
            for (int t = 0; t < MT; ++t) ra[sI][t] = *reinterpret_cast<const f32x4*>(Ap[t] + 16 * (jj + PF));
#pragma unroll
            for (int s = 0; s < 4; ++s) rb[sI][s] = *reinterpret_cast<const f32x4*>(Wp + (size_t)(16 * (jj + PF) + s) * ldw);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], b[s][u], acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (GT > 0) {
#pragma unroll
        for (int jj = 0; jj < GT; ++jj) stage(jj % PF, jj);
    } else {
        for (int j0 = 0; j0 < G; j0 += PF) {
#pragma unroll
            for (int sI = 0; sI < PF; ++sI)
                if (j0 + sI < G) stage(sI, j0 + sI);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[(w * Q + (t * 4 + v) * 4 + u) * 64 + lane] = acc[t][u][v];
    __syncthreads();
    // output piece (t, v) of lane (c, g): row m0 + 16 t + 4 g + v, the four columns n0 + 4 c .. + 3 (u = 0..3): one 16-byte store
    for (int pc = w; pc < MT * 4; pc += NW) {
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ww = 0; ww < NW; ++ww)
#pragma unroll
            for (int u = 0; u < 4; ++u) val[u] += red[(ww * Q + pc * 4 + u) * 64 + lane];
        const int t = pc >> 2, v = pc & 3;
        const int row = m0 + 16 * t + 4 * g + v, col = n0 + 4 * c;
        if (epi & EPN_GELU_BWD) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(H + (size_t)row * ldh + col);
#pragma unroll
            for (int u = 0; u < 4; ++u) val[u] *= gelu_grad(h[u]);
        }
        *reinterpret_cast<f32x4*>(Y + (size_t)row * ldy + col) = val;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// attention of one (prompt, head) over the prompt's compact rows.  Token row i sees rows j <= i (causal); the CLS row (last)
// sees the rows flagged in cls_keep (model/prompt_encoder.py:245-252,299-303).
constexpr int kAttnMaxS = 128;
__global__ __launch_bounds__(256) void k_tt_attn_fwd(const float* __restrict__ qkv, int ld, float* __restrict__ out, int ldo,
                                                    const int* __restrict__ seq_row0, const unsigned char* __restrict__ cls_keep,
                                                    int heads, int d) {
    __shared__ float Ks[kAttnMaxS][kHeadDim + 1];
    __shared__ float Vs[kAttnMaxS][kHeadDim + 1];
    __shared__ float Qs[kAttnMaxS][kHeadDim];
    const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
    const int r0 = seq_row0[seq], S = seq_row0[seq + 1] - r0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < S * kHeadDim; e += 256) {   // q, k, v of the prompt in ONE round of global loads
        const int j = e >> 6, c = e & 63;
        const size_t base = (size_t)(r0 + j) * ld + h * kHeadDim + c;
        const float qv = qkv[base], kv = qkv[base + d], vv = qkv[base + 2 * d];
        Qs[j][c] = qv * 0.125f;   // head_dim^-0.5
        Ks[j][c] = kv;
        Vs[j][c] = vv;
    }
    __syncthreads();
    for (int i = w; i < S; i += 4) {
        const float q = Qs[i][lane];
        const bool is_cls = i == S - 1;
        float s[2], p[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int j = lane + 64 * half;
            const int jc = j < S ? j : S - 1;          // every lane computes (no cross-lane reads under a divergent branch)
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < kHeadDim; ++c)
                dot = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(q), c)), Ks[jc][c], dot);
            const bool ok = j < S && (is_cls ? cls_keep[r0 + jc] != 0 : j <= i);
            s[half] = ok ? dot : -INFINITY;
        }
        const float m = wave_max(fmaxf(s[0], s[1]));
        p[0] = s[0] == -INFINITY ? 0.f : __expf(s[0] - m);
        p[1] = s[1] == -INFINITY ? 0.f : __expf(s[1] - m);
        const float inv = 1.f / wave_sum(p[0] + p[1]);
        p[0] *= inv;
        p[1] *= inv;
        float o = 0.f;
        const int S0 = S < 64 ? S : 64;
        for (int j = 0; j < S0; ++j)
            o = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[0]), __builtin_amdgcn_readfirstlane(j))), Vs[j][lane], o);
        for (int j = 64; j < S; ++j)
            o = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[1]), __builtin_amdgcn_readfirstlane(j - 64))), Vs[j][lane], o);
        out[(size_t)(r0 + i) * ldo + h * kHeadDim + lane] = o;
    }
}

constexpr int kAttnBwdMaxS = 64;
__global__ __launch_bounds__(256) void k_tt_attn_bwd(const float* __restrict__ qkv, int ld, const float* __restrict__ dout, int ldo,
                                                    float* __restrict__ dqkv, const int* __restrict__ seq_row0,
                                                    const unsigned char* __restrict__ cls_keep, int heads, int d) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int LD = kHeadDim + 1;
    float* Qs = sm;
    float* Ks = Qs + kAttnBwdMaxS * LD;
    float* Vs = Ks + kAttnBwdMaxS * LD;
    float* Os = Vs + kAttnBwdMaxS * LD;      // dO
    float* Pm = Os + kAttnBwdMaxS * LD;      // softmax weights [i][j]
    float* Dm = Pm + kAttnBwdMaxS * LD;      // d scores (scale folded in) [i][j]
    const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
    const int r0 = seq_row0[seq], S = seq_row0[seq + 1] - r0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int e = tid; e < S * kHeadDim; e += 256) {
        const int j = e >> 6, c = e & 63;
        const size_t base = (size_t)(r0 + j) * ld + h * kHeadDim + c;
        Qs[j * LD + c] = qkv[base];
        Ks[j * LD + c] = qkv[base + d];
        Vs[j * LD + c] = qkv[base + 2 * d];
        Os[j * LD + c] = dout[(size_t)(r0 + j) * ldo + h * kHeadDim + c];
    }
    __syncthreads();
    for (int i = w; i < S; i += 4) {   // lane j: score, weight and their gradients for key j of query row i
        const bool is_cls = i == S - 1;
        const int j = lane;
        float dot = 0.f, dp = 0.f;
        if (j < S) {
#pragma unroll
            for (int c = 0; c < kHeadDim; ++c) {
                dot = fmaf(Qs[i * LD + c], Ks[j * LD + c], dot);
                dp = fmaf(Os[i * LD + c], Vs[j * LD + c], dp);
            }
        }
        const bool ok = j < S && (is_cls ? cls_keep[r0 + j] != 0 : j <= i);
        const float s = ok ? dot * 0.125f : -INFINITY;
        const float m = wave_max(s);
        float p = ok ? __expf(s - m) : 0.f;
        p *= 1.f / wave_sum(p);
        const float delta = wave_sum(p * dp);
        Pm[i * LD + j] = p;
        Dm[i * LD + j] = p * (dp - delta) * 0.125f;
    }
    __syncthreads();
    for (int rr = w; rr < S; rr += 4) {   // lane = feature c of row rr: dQ, dK, dV
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < S; ++j) {
            dq = fmaf(Dm[rr * LD + j], Ks[j * LD + lane], dq);
            dk = fmaf(Dm[j * LD + rr], Qs[j * LD + lane], dk);
            dv = fmaf(Pm[j * LD + rr], Os[j * LD + lane], dv);
        }
        const size_t base = (size_t)(r0 + rr) * ld + h * kHeadDim + lane;
        dqkv[base] = dq;
        dqkv[base + d] = dk;
        dqkv[base + 2 * d] = dv;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm helpers: one wave per row, d <= 1024 (d % 64 == 0): 16 register slots per lane.
constexpr int kLnSlots = 16;
__device__ __forceinline__ void ln_stats(const float (&v)[kLnSlots], int nslot, int d, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) s += v[k];
    mean = wave_sum(s) / (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            const float c = v[k] - mean;
            s2 = fmaf(c, c, s2);
        }
    rstd = 1.f / sqrtf(wave_sum(s2) / (float)d + kLnEps);
}

// dx[row] = dres[row] + LayerNorm'(x[row]; gamma)^T da[row]   (dres nullable)
__global__ __launch_bounds__(256) void k_tt_ln_bwd(const float* __restrict__ da, const float* __restrict__ x,
                                                  const float* __restrict__ gamma, const float* __restrict__ dres,
                                                  float* __restrict__ dx, int d, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nslot = d >> 6;
    float xv[kLnSlots], gv[kLnSlots], rv[kLnSlots];
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {   // every load of the row in one round
            xv[k] = x[(size_t)row * d + lane + 64 * k];
            gv[k] = da[(size_t)row * d + lane + 64 * k] * gamma[lane + 64 * k];
            rv[k] = dres ? dres[(size_t)row * d + lane + 64 * k] : 0.f;
        }
    float mean, rstd;
    ln_stats(xv, nslot, d, mean, rstd);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            xv[k] = (xv[k] - mean) * rstd;
            sg += gv[k];
            sgx = fmaf(gv[k], xv[k], sgx);
        }
    sg = wave_sum(sg) / (float)d;
    sgx = wave_sum(sgx) / (float)d;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) dx[(size_t)row * d + lane + 64 * k] = rv[k] + rstd * (gv[k] - sg - xv[k] * sgx);
}

// pooled[s] = ln_final(x[CLS row of prompt s]); rows n_seq .. n_pad-1 of pooled are zeroed (GEMM padding).
__global__ __launch_bounds__(256) void k_tt_lnf_fwd(const float* __restrict__ x, const int* __restrict__ seq_row0,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ pooled, int d, int n_seq, int n_pad) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_pad) return;
    const int nslot = d >> 6;
    if (s >= n_seq) {
        for (int k = 0; k < nslot; ++k) pooled[(size_t)s * d + lane + 64 * k] = 0.f;
        return;
    }
    const int row = seq_row0[s + 1] - 1;
    float xv[kLnSlots];
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) xv[k] = x[(size_t)row * d + lane + 64 * k];
    float mean, rstd;
    ln_stats(xv, nslot, d, mean, rstd);
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) pooled[(size_t)s * d + lane + 64 * k] = fmaf((xv[k] - mean) * rstd, gamma[lane + 64 * k], beta[lane + 64 * k]);
}

// dx[row] = ln_final backward of dpooled[s] on the CLS row of prompt s, zero on every other row.
__global__ __launch_bounds__(256) void k_tt_lnf_bwd(const float* __restrict__ dpooled, const float* __restrict__ x,
                                                   const int* __restrict__ row_seq, const int* __restrict__ row_src,
                                                   const float* __restrict__ gamma, float* __restrict__ dx, int d, int M, int M_pad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M_pad) return;
    const int nslot = d >> 6;
    if (row >= M || row_src[row] >= 0) {
        for (int k = 0; k < nslot; ++k) dx[(size_t)row * d + lane + 64 * k] = 0.f;
        return;
    }
    const int s = row_seq[row];
    float xv[kLnSlots], gv[kLnSlots];
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            xv[k] = x[(size_t)row * d + lane + 64 * k];
            gv[k] = dpooled[(size_t)s * d + lane + 64 * k] * gamma[lane + 64 * k];
        }
    float mean, rstd;
    ln_stats(xv, nslot, d, mean, rstd);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            xv[k] = (xv[k] - mean) * rstd;
            sg += gv[k];
            sgx = fmaf(gv[k], xv[k], sgx);
        }
    sg = wave_sum(sg) / (float)d;
    sgx = wave_sum(sgx) / (float)d;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) dx[(size_t)row * d + lane + 64 * k] = rstd * (gv[k] - sg - xv[k] * sgx);
}

// d prompts_embedding[seq, src] = dx[row] for the token rows (the CLS rows' gradient belongs to the frozen cls_emb).
__global__ __launch_bounds__(256) void k_tt_scatter(const float* __restrict__ dx, int d, float* __restrict__ demb, int64_t s_seq,
                                                   int64_t s_tok, const int* __restrict__ row_seq, const int* __restrict__ row_src,
                                                   int M) {
    const int row = blockIdx.x;
    if (row >= M) return;
    const int src = row_src[row];
    if (src < 0) return;
    float* o = demb + (size_t)row_seq[row] * s_seq + (size_t)src * s_tok;
    for (int c = threadIdx.x; c < d; c += 256) o[c] = dx[(size_t)row * d + c];
}

}  // namespace tt
}  // namespace vlsa

using namespace vlsa;
using namespace vlsa::tt;

namespace {

struct Shape {
    int d, heads, layers, out_dim, M_pad, M, n_seq, ns_pad, MG3, MG1;
};

bool shape_of(const vlsa_tt_model* m, const vlsa_tt_rows* r, Shape& s) {
    if (!m || !r || !m->layer) return false;
    s.d = m->width;
    s.heads = m->heads;
    s.layers = m->layers;
    s.out_dim = m->out_dim;
    s.M = r->M;
    s.M_pad = r->M_pad;
    s.n_seq = r->n_seq;
    s.ns_pad = (r->n_seq + 47) / 48 * 48;
    s.MG3 = r->M_pad / 48;
    s.MG1 = r->M_pad / 16;
    // width: multiple of 128 (the K splits of every product are whole 16-column groups), <= 768 (fused-LayerNorm slab), 64 per head
    if (s.d < 128 || s.d > 768 || (s.d % 128) || s.heads * kHeadDim != s.d) return false;
    if (s.layers < 1 || s.out_dim < 64 || (s.out_dim % 64) || s.out_dim > 1024) return false;
    if (s.M < 1 || s.M_pad < s.M || (s.M_pad % 48) || s.n_seq < 1 || r->max_len < 2 || r->max_len > kAttnMaxS) return false;
    if (!r->row_seq || !r->row_pos || !r->row_src || !r->seq_row0 || !r->cls_keep) return false;
    if (!m->pos_emb || !m->cls_emb || !m->lnf_w || !m->lnf_b || !m->text_proj) return false;
    return true;
}

// per-layer region of the workspace (floats): x_in [M_pad, d] | qkv [M_pad, 3d] | x_mid [M_pad, d] | h_pre [M_pad, 4d]
inline size_t layer_floats(const Shape& s) { return (size_t)s.M_pad * s.d * 9; }
// shared scratch behind the layer regions: x_final | attn [M_pad, d] | h_act [M_pad, 4d] | pooled [ns_pad, d] |
// backward: dxa, dxb [M_pad, d] | dbig [M_pad, 4d] | dpool [ns_pad, d] | dout_pad [ns_pad, out_dim]
inline size_t scratch_floats(const Shape& s) {
    return (size_t)s.M_pad * s.d * (1 + 1 + 4 + 2 + 4) + (size_t)s.ns_pad * s.d * 2 + (size_t)s.ns_pad * s.out_dim;
}

template <int MT, int NW, int PRO, int GT>
int launch_nt_g(const float* A, int lda, const float* W, const float* bias, const float* resid, int ldr, float* Y, int ldy,
                float* Ypre, int N, int K, int M_pad, int epi, const float* ln_w, const float* ln_b, hipStream_t st) {
    const int MG = M_pad / (16 * MT), NT = N / 32;
    const size_t lds = (size_t)NW * MT * 8 * 64 * sizeof(float);
    hipLaunchKernelGGL((k_tt_gemm_nt<MT, NW, PRO, GT>), dim3(NT * MG), dim3(NW * 64), lds, st, A, lda, W, bias, resid, ldr, Y, ldy,
                       Ypre, N, K, MG, (NT % 8 == 0) ? 1 : 0, epi, ln_w, ln_b);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}
// G1, G2: the group counts of the CONCH-size tower for this product (compile-time specialisations); anything else: runtime G
template <int MT, int NW, int PRO, int G1, int G2 = G1>
int launch_nt(const float* A, int lda, const float* W, const float* bias, const float* resid, int ldr, float* Y, int ldy,
              float* Ypre, int N, int K, int M_pad, int epi, const float* ln_w, const float* ln_b, hipStream_t st) {
    const int G = K / NW / 16;
    if (G == G1) return launch_nt_g<MT, NW, PRO, G1>(A, lda, W, bias, resid, ldr, Y, ldy, Ypre, N, K, M_pad, epi, ln_w, ln_b, st);
    if (G == G2) return launch_nt_g<MT, NW, PRO, G2>(A, lda, W, bias, resid, ldr, Y, ldy, Ypre, N, K, M_pad, epi, ln_w, ln_b, st);
    return launch_nt_g<MT, NW, PRO, 0>(A, lda, W, bias, resid, ldr, Y, ldy, Ypre, N, K, M_pad, epi, ln_w, ln_b, st);
}

template <int MT, int NW, int GT>
int launch_nn_g(const float* A, int lda, const float* W, int ldw, float* Y, int ldy, int Kc, int Nout, int M_pad, int epi,
                const float* H, int ldh, hipStream_t st) {
    const int MG = M_pad / (16 * MT);
    const size_t lds = (size_t)NW * MT * 16 * 64 * sizeof(float);
    static DeviceOnce once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)k_tt_gemm_nn<MT, NW, GT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_tt_gemm_nn<MT, NW, GT>), dim3((Nout / 64) * MG), dim3(NW * 64), lds, st, A, lda, W, ldw, Y, ldy, Kc, MG, epi, H, ldh);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}
template <int MT, int NW, int G1, int G2 = G1>
int launch_nn(const float* A, int lda, const float* W, int ldw, float* Y, int ldy, int Kc, int Nout, int M_pad, int epi,
              const float* H, int ldh, hipStream_t st) {
    const int G = Kc / NW / 16;
    if (G == G1) return launch_nn_g<MT, NW, G1>(A, lda, W, ldw, Y, ldy, Kc, Nout, M_pad, epi, H, ldh, st);
    if (G == G2) return launch_nn_g<MT, NW, G2>(A, lda, W, ldw, Y, ldy, Kc, Nout, M_pad, epi, H, ldh, st);
    return launch_nn_g<MT, NW, 0>(A, lda, W, ldw, Y, ldy, Kc, Nout, M_pad, epi, H, ldh, st);
}

#define TT_TRY(expr)                  \
    do {                              \
        const int rc_ = (expr);       \
        if (rc_ != VLSA_OK) return rc_; \
    } while (0)

}  // namespace

extern "C" size_t vlsa_tt_workspace_bytes(const vlsa_tt_model* m, const vlsa_tt_rows* r, int save_for_backward) {
    Shape s;
    if (!shape_of(m, r, s)) return 0;
    const size_t regions = save_for_backward ? (size_t)s.layers : 1;
    return (regions * layer_floats(s) + scratch_floats(s)) * sizeof(float);
}

#ifdef VLSA_TT_DEBUG
#include <cstdlib>
static int tt_debug_bits() {
    const char* e = getenv("VLSA_TT_DEBUG_BITS");
    return e ? atoi(e) : 0;
}
#define TT_DBG tt_debug_bits()
#else
#define TT_DBG 0
#endif

extern "C" int vlsa_tt_forward(const vlsa_tt_model* m, const vlsa_tt_rows* r, const float* emb, int64_t emb_seq_stride,
                               int64_t emb_tok_stride, void* workspace, int save_for_backward, float* out, void* stream) {
    Shape s;
    if (!shape_of(m, r, s)) return VLSA_EINVAL;
    if (!emb || !workspace || !out) return VLSA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int d = s.d, Mp = s.M_pad;
    float* ws = static_cast<float*>(workspace);
    const size_t LF = layer_floats(s);
    const size_t nreg = save_for_backward ? (size_t)s.layers : 1;
    float* scratch = ws + nreg * LF;
    float* x_final = scratch;
    float* attn = x_final + (size_t)Mp * d;
    float* h_act = attn + (size_t)Mp * d;
    float* pooled = h_act + (size_t)Mp * 4 * d;

    auto region = [&](int layer) { return ws + (save_for_backward ? (size_t)layer : 0) * LF; };
    hipLaunchKernelGGL(k_tt_embed, dim3(Mp), dim3(256), 0, st, region(0), d, emb, emb_seq_stride, emb_tok_stride, r->row_seq,
                       r->row_pos, r->row_src, m->pos_emb, m->cls_emb, s.M);
    if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    for (int L = 0; L < s.layers; ++L) {
        const vlsa_tt_layer& w = m->layer[L];
        float* x_in = region(L);
        float* qkv = x_in + (size_t)Mp * d;
        float* x_mid = qkv + (size_t)Mp * 3 * d;
        float* h_pre = x_mid + (size_t)Mp * d;
        float* x_next = (L + 1 < s.layers) ? (save_for_backward ? region(L + 1) : x_in) : x_final;
        // x_mid = x_in + out_proj(attention(ln_1(x_in)));  x_next = x_mid + c_proj(gelu(c_fc(ln_2(x_mid))))
        TT_TRY((launch_nt<3, 4, PRO_LN, 12>(x_in, d, w.in_w, w.in_b, nullptr, 0, qkv, 3 * d, nullptr, 3 * d, d, Mp, EPI_BIAS, w.ln1_w, w.ln1_b, st)));
        hipLaunchKernelGGL(k_tt_attn_fwd, dim3(s.n_seq * s.heads), dim3(256), 0, st, qkv, 3 * d, attn, d, r->seq_row0, r->cls_keep,
                           s.heads, d);
        if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
        TT_TRY((launch_nt<3, 8, PRO_NONE, 6>(attn, d, w.out_w, w.out_b, x_in, d, x_mid, d, nullptr, d, d, Mp, EPI_BIAS | EPI_RESID, nullptr, nullptr, st)));
        TT_TRY((launch_nt<3, 4, PRO_LN, 12>(x_mid, d, w.fc_w, w.fc_b, nullptr, 0, h_act, 4 * d, save_for_backward ? h_pre : nullptr, 4 * d, d, Mp,
                                        EPI_BIAS | EPI_GELU, w.ln2_w, w.ln2_b, st)));
        TT_TRY((launch_nt<1, 8, PRO_NONE, 24>(h_act, 4 * d, w.proj_w, w.proj_b, x_mid, d, x_next, d, nullptr, d, 4 * d, Mp, EPI_BIAS | EPI_RESID | TT_DBG, nullptr,
                                          nullptr, st)));
    }
    hipLaunchKernelGGL(k_tt_lnf_fwd, dim3((s.ns_pad + 3) / 4), dim3(256), 0, st, x_final, r->seq_row0, m->lnf_w, m->lnf_b, pooled, d,
                       s.n_seq, s.ns_pad);
    if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    // text features = pooled @ text_projection  ([ns_pad, d] x [d, out_dim]); the padded rows land in scratch, then copy out
    float* feat = pooled + (size_t)s.ns_pad * d * 2;   // dout_pad slot doubles as the padded output
    TT_TRY((launch_nn<3, 4, 12>(pooled, d, m->text_proj, s.out_dim, feat, s.out_dim, d, s.out_dim, s.ns_pad, 0, nullptr, 0, st)));
    if (hipMemcpyAsync(out, feat, (size_t)s.n_seq * s.out_dim * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return VLSA_ELAUNCH;
    return VLSA_OK;
}

extern "C" int vlsa_tt_backward(const vlsa_tt_model* m, const vlsa_tt_rows* r, const float* dout, void* workspace, float* demb,
                                int64_t emb_seq_stride, int64_t emb_tok_stride, int64_t demb_floats, void* stream) {
    Shape s;
    if (!shape_of(m, r, s)) return VLSA_EINVAL;
    if (!dout || !workspace || !demb || demb_floats < 0) return VLSA_EINVAL;
    if (r->max_len > kAttnBwdMaxS) return VLSA_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int d = s.d, Mp = s.M_pad;
    float* ws = static_cast<float*>(workspace);
    const size_t LF = layer_floats(s);
    float* scratch = ws + (size_t)s.layers * LF;
    float* x_final = scratch;
    float* attn = x_final + (size_t)Mp * d;          // reused: d(attention output)
    float* h_act = attn + (size_t)Mp * d;            // reused: d(h_pre) [M_pad, 4d]
    float* pooled = h_act + (size_t)Mp * 4 * d;
    float* dpool = pooled + (size_t)s.ns_pad * d;
    float* dout_pad = dpool + (size_t)s.ns_pad * d;
    float* dxa = dout_pad + (size_t)s.ns_pad * s.out_dim;
    float* dxb = dxa + (size_t)Mp * d;
    float* dbig = dxb + (size_t)Mp * d;               // [M_pad, 4d]: d(ln output) / dqkv (3d)
    (void)pooled;

    if (hipMemsetAsync(dout_pad, 0, (size_t)s.ns_pad * s.out_dim * sizeof(float), st) != hipSuccess) return VLSA_ELAUNCH;
    if (hipMemcpyAsync(dout_pad, dout, (size_t)s.n_seq * s.out_dim * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return VLSA_ELAUNCH;
    // d pooled = dout @ text_projection^T: text_projection [d, out_dim] read as W[n = d][k = out_dim]
    if ((s.out_dim / 4) % 16) return VLSA_EUNSUPPORTED;
    TT_TRY((launch_nt<3, 4, PRO_NONE, 8>(dout_pad, s.out_dim, m->text_proj, nullptr, nullptr, 0, dpool, d, nullptr, d, s.out_dim, s.ns_pad, 0, nullptr,
                                      nullptr, st)));
    hipLaunchKernelGGL(k_tt_lnf_bwd, dim3((Mp + 3) / 4), dim3(256), 0, st, dpool, x_final, r->row_seq, r->row_src, m->lnf_w, dxa, d, s.M, Mp);
    if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    static DeviceOnce once;
    const size_t attn_lds = (size_t)6 * kAttnBwdMaxS * (kHeadDim + 1) * sizeof(float);
    if (once.first()) (void)hipFuncSetAttribute((const void*)k_tt_attn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds);
    float* dx = dxa;       // gradient w.r.t. the layer's output
    float* dx2 = dxb;
    for (int L = s.layers - 1; L >= 0; --L) {
        const vlsa_tt_layer& w = m->layer[L];
        float* x_in = ws + (size_t)L * LF;
        float* qkv = x_in + (size_t)Mp * d;
        float* x_mid = qkv + (size_t)Mp * 3 * d;
        float* h_pre = x_mid + (size_t)Mp * d;
        // MLP branch: d h_pre = (dx @ W_proj) * gelu'(h_pre);  d ln_2 out = d h_pre @ W_fc;  dx_mid = dx + ln_2'(.)
        TT_TRY((launch_nn<3, 8, 6>(dx, d, w.proj_w, 4 * d, h_act, 4 * d, d, 4 * d, Mp, EPN_GELU_BWD, h_pre, 4 * d, st)));
        TT_TRY((launch_nn<1, 8, 24, 18>(h_act, 4 * d, w.fc_w, d, dbig, d, 4 * d, d, Mp, 0, nullptr, 0, st)));
        hipLaunchKernelGGL(k_tt_ln_bwd, dim3((Mp + 3) / 4), dim3(256), 0, st, dbig, x_mid, w.ln2_w, dx, dx2, d, Mp);
        if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
        // attention branch: d attn = dx_mid @ W_out;  dqkv = attention'(.);  d ln_1 out = dqkv @ W_in;  dx_in = dx_mid + ln_1'(.)
        TT_TRY((launch_nn<1, 4, 12>(dx2, d, w.out_w, d, attn, d, d, d, Mp, 0, nullptr, 0, st)));
        hipLaunchKernelGGL(k_tt_attn_bwd, dim3(s.n_seq * s.heads), dim3(256), attn_lds, st, qkv, 3 * d, attn, d, dbig, r->seq_row0,
                           r->cls_keep, s.heads, d);
        if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
        // rows M .. M_pad-1 of dqkv are never written by the attention kernel: they hold stale finite numbers that only reach
        // the (discarded) padding rows of the next products
        TT_TRY((launch_nn<1, 8, 24, 18>(dbig, 3 * d, w.in_w, d, h_act, d, 3 * d, d, Mp, 0, nullptr, 0, st)));
        hipLaunchKernelGGL(k_tt_ln_bwd, dim3((Mp + 3) / 4), dim3(256), 0, st, h_act, x_in, w.ln1_w, dx2, dx, d, Mp);
        if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    }
    if (hipMemsetAsync(demb, 0, (size_t)demb_floats * sizeof(float), st) != hipSuccess) return VLSA_ELAUNCH;
    hipLaunchKernelGGL(k_tt_scatter, dim3(s.M), dim3(256), 0, st, dx, d, demb, emb_seq_stride, emb_tok_stride, r->row_seq, r->row_src, s.M);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

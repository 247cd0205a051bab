/*
 * vlsa_hip.h -- C ABI of the MI355X-native VLSA patch-aggregation hot path (libvlsa_hip.so).
 *
 * The upstream reference (liupei101/VLSA) has no FFI / operator registry: its replaceable seam is the
 * duck-typed Python module interface.  These entry points are what a ctypes binding on
 * the reference side would call in place of the torch op sequences cited next to each function
 * (paths relative to the reference root).  INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless said otherwise; all buffers are caller-allocated;
 *   - no allocation, no synchronisation, no global state inside: each call only enqueues kernels (and
 *     hipMemsetAsync nodes) on `stream` (a hipStream_t passed as void*), so calls are re-entrant across
 *     streams and capturable in a hipGraph;
 *   - return value: 0 on success, a negative VLSA_E* code otherwise (nothing was enqueued then);
 *   - matrices are row-major; `ldx` is the row stride of X in ELEMENTS;
 *   - P = number of effective queries (<= VLSA_MAX_P); with the gated query the caller passes nq = P+1
 *     raw queries to vlsa_prepare_queries and the subtraction is folded into the effective queries;
 *   - "log2 domain": scores t = s * log2(e) with s = coattn_scale * cos(q, x); m2 = max t.
 */
#ifndef VLSA_HIP_H
#define VLSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLSA_ABI_VERSION 1
#define VLSA_MAX_P 16      /* queries per bag (reference: 7..12 prototypes (+1 gate), runner/global_cfg.py:1-22) */
#define VLSA_MAX_K 64      /* ordinal rank prompts / classes */
#define VLSA_MAX_D 1024    /* feature dim; the tuned MFMA kernels need D == 512 (CONCH) */

#define VLSA_DT_F32 0
#define VLSA_DT_BF16 1

#define VLSA_OK 0
#define VLSA_EINVAL (-1)      /* bad argument (null pointer, size out of range, misaligned) */
#define VLSA_EUNSUPPORTED (-2)/* valid request the library has no kernel for */
#define VLSA_ELAUNCH (-3)     /* hipLaunch / hipMemsetAsync reported an error */

/* kernel selection for vlsa_vlfan_partial */
#define VLSA_KERNEL_AUTO 0
#define VLSA_KERNEL_GENERIC 1 /* fp32 VALU kernel, any D <= VLSA_MAX_D with D % 8 == 0 */
#define VLSA_KERNEL_MFMA 2    /* split-bf16 MFMA kernel, register-staged, D == 512, fp32 or bf16 rows */
#define VLSA_KERNEL_DMA 3     /* split-bf16 MFMA kernel, LDS-DMA ring, D == 512, bf16 rows (the tuned path) */

/* query pooling over the P aggregated rows (model/deepmil.py:133-150) */
#define VLSA_POOL_MEAN 0
#define VLSA_POOL_MAX 1
#define VLSA_POOL_WEIGHT 2    /* softmax(pool_w[P]) @ out */
#define VLSA_POOL_GIVEN 3     /* `out` already holds ONE pooled row (P == 1): attention poolings done by the caller */

int vlsa_abi_version(void);
const char* vlsa_error_string(int code);

/* Number of per-workgroup partials vlsa_vlfan_partial writes for a shard of N rows (deterministic). */
int vlsa_num_partials(int64_t N);

/* Bytes of the opaque prepared-query block for feature dim D. */
size_t vlsa_qprep_bytes(int D);

/*
 * Replaces: Q = F.normalize(Q.unsqueeze(0), dim=-1) and, with gated_query, the later
 * A_[:, :-1] - A_[:, -1:] (model/deepmil.py:187,192-195) -- folded into effective queries
 * e_p = q^_p - q^_gate, which is exact because the score is linear in the query.
 *   Q      [nq, D] fp32 raw queries (nq = P, or P+1 when gated)
 *   coattn_scale  the reference's exp(coattn_logit_scale) = 100 (model/deepmil.py:120-126,197)
 *   qprep  opaque block of vlsa_qprep_bytes(D): fp32 effective queries, the 3-term bf16 split of
 *          coattn_scale * log2(e) * e_p, fp32 unit queries q^ [nq, D] and raw norms [nq] (the last two are
 *          what backward needs).
 */
int vlsa_prepare_queries(const float* Q, int nq, int D, int gated, float coattn_scale, void* qprep, void* stream);
/* vlsa_prepare_queries + vlsa_normalize_rows(T -> That, tnorm nullable) in ONE launch (both are bag-independent). */
int vlsa_prepare_queries_and_text(const float* Q, int nq, int D, int gated, float coattn_scale, void* qprep,
                                  const float* T, int K, float* That, float* tnorm, void* stream);
/* Accessors into the opaque block (device pointers; for backward and for tests). */
const float* vlsa_qprep_qeff(const void* qprep, int D);   /* [16, D] effective queries, rows >= P zero */
const float* vlsa_qprep_qhat(const void* qprep, int D);   /* [17, D] unit queries */
const float* vlsa_qprep_qnorm(const void* qprep, int D);  /* [17] max(||q||, 1e-12) */

/*
 * Replaces: norm_X = F.normalize(X); A_ = Q @ norm_X^T; A_ *= 100; softmax over N; out = A @ X
 * (model/deepmil.py:189-200), as ONE streaming pass over the shard's rows producing per-workgroup
 * online-softmax partials:
 *   pm   [G, 16]   log2-domain running max per query        (G = vlsa_num_partials(N))
 *   pl   [G, 16]   sum_n exp2(t_pn - pm)
 *   pacc [G, P, D] sum_n exp2(t_pn - pm) x_n
 *   scores (nullable) [P, N] log2-domain scores t_pn, needed only when attention weights are wanted.
 * X: [N, D] rows, dtype VLSA_DT_F32 or VLSA_DT_BF16, row stride ldx elements, 16-byte aligned rows.
 */
int vlsa_vlfan_partial(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* qprep, int P,
                       int kernel, float* pm, float* pl, float* pacc, float* scores, void* stream);

/*
 * Log-sum-exp merge of G partials (from this GPU's workgroups, or all-gathered from the other ranks):
 *   m2 [16], l [16], out [P, D];  normalise != 0 -> out_p = acc_p / l_p (the reference's A @ X row),
 *   normalise == 0 -> out holds the un-normalised merged acc (a compact partial for the RCCL exchange).
 */
int vlsa_vlfan_merge(const float* pm, const float* pl, const float* pacc, int G, int P, int D, int normalise,
                     float* m2, float* l, float* out, void* stream);
/* Same with explicit strides (in floats) between consecutive partials -- lets the all-gathered per-rank records
 * [m2(16) | l(16) | acc(P*D)] of the multi-GPU path be merged in place.  pacc_stride % 4 == 0. */
int vlsa_vlfan_merge_strided(const float* pm, int64_t pm_stride, const float* pl, int64_t pl_stride, const float* pacc,
                             int64_t pacc_stride, int G, int P, int D, int normalise, float* m2, float* l, float* out,
                             void* stream);

/*
 * Backward of the aggregation w.r.t. the EFFECTIVE queries (what torch.autograd does through
 * model/deepmil.py:187-200; X carries no gradient).  Given dout = dLoss/d(out) [P, D] and the forward's out, m2, l:
 * one more streaming pass writes per-workgroup partial SUMS of
 *     de_p = coattn_scale * sum_n A_pn (dout_p . x_n - dout_p . out_p) x_n / max(|x_n|, 1e-12)
 * into pm (= 0), pl (= 1), pacc [G, P, D]; reduce them with vlsa_vlfan_merge(..., normalise = 0).
 * bwd_prep: scratch of vlsa_bwd_prep_bytes(D).  D must be 512 (VLSA_EUNSUPPORTED otherwise).
 */
size_t vlsa_bwd_prep_bytes(int D);
int vlsa_vlfan_backward(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* qprep, int P,
                        float coattn_scale, const float* dout, const float* out, const float* m2, const float* l,
                        void* bwd_prep, float* pm, float* pl, float* pacc, void* stream);

/* A[p, n] = exp2(scores[p, n] - m2[p]) / l[p]   (softmax of model/deepmil.py:198). */
int vlsa_attn_normalise(const float* scores, int P, int64_t N, const float* m2, const float* l, float* A,
                        void* stream);

/* out[r, :] = in[r, :] / max(||in[r, :]||, 1e-12); norms[r] (nullable) = that denominator. (F.normalize) */
int vlsa_normalize_rows(const float* in, int rows, int D, float* out, float* norms, void* stream);

/* Partial merge + head of ONE bag in one call (the tail of vlsa_vlfan_forward_bag; model/deepmil.py:198-204 + model/vlsa.py:188-192):
 * vlsa_vlfan_merge(normalise = 1) + vlsa_head_forward with the same arguments and results.  D == 512, mean / weight pooling and a
 * Linear adapter run as two ticket-free launches (the merge workgroups multiply their merged columns with their slice of W). */
int vlsa_vlfan_merge_head(const float* pm, const float* pl, const float* pacc, int G, int P, int D, int pool_mode,
                          const float* pool_w, const float* W, const float* b, const float* That, int K,
                          const float* logit_scale, void* head_ws, float* m2, float* l, float* out, float* pooled, float* v,
                          float* vhat, float* vnorm, float* logits, float* incidence, void* stream);

/* Bytes of scratch vlsa_head_forward / vlsa_vlfan_merge_head need (a ticket counter + partial adapter vectors).  The caller zeroes it ONCE after allocation; every
 * call leaves it zeroed again, so no per-call memset is needed (one workspace per stream in flight). */
size_t vlsa_head_workspace_bytes(int D);

/*
 * Replaces: forward_query_pooling (mean | max | weight) -> visual_adapter (nn.Linear or Identity) ->
 * F.normalize -> exp(logit_scale) * v^ @ T^^T (model/deepmil.py:203-204, model/vlsa.py:188-192), and
 * optionally the softmax output converter (utils/func.py:43-44).
 *   rows [P, D]; pool_w [P] raw 'weight' pooling parameter (nullable); W [D, D], b [D] (nullable => Identity)
 *   That [K, D] UNIT-norm text features; logit_scale: device pointer to the scalar parameter (pre-exp)
 *   outputs: pooled [D], v [D], vhat [D], vnorm [1], logits [K], incidence [K] (nullable)
 */
int vlsa_head_forward(const float* rows, int P, int D, int pool_mode, const float* pool_w, const float* W,
                      const float* b, const float* That, int K, const float* logit_scale, void* workspace,
                      float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence,
                      void* stream);

/* vlsa_prepare_queries_and_text + vlsa_vlfan_partial + vlsa_vlfan_merge (+ vlsa_attn_normalise when scores and A are given) +
 * vlsa_head_forward in ONE host call: the per-bag forward of model/vlsa.py:181-198 as the reference's handler issues it, one
 * bag at a time (runner/vlsa_handler.py:322-330).  Arguments as in the individual entry points; G = vlsa_num_partials(N).
 * Q == NULL skips the preparation launch: qprep / That must then hold the result of an earlier call with unchanged queries and
 * text features (an evaluation loop prepares them once per checkpoint, not once per bag).  pool_mode < 0 stops after the
 * merge (out [P, D] is the result): for the attention query poolings, followed by vlsa_query_pool_attention + vlsa_head_forward. */
int vlsa_vlfan_forward_bag(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* Q, int nq, int gated,
                           float coattn_scale, const float* T, int K, const float* logit_scale, int pool_mode, const float* pool_w,
                           const float* W, const float* b, int kernel, void* qprep, float* That, float* tnorm, float* pm, float* pl,
                           float* pacc, int G, float* m2, float* l, float* out, float* scores, float* A, void* head_ws,
                           float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence, void* stream);

/* dQ [nq, D] from dE [P, D] = d loss / d (effective unit queries): the chain rule through q^ = q / max(|q|, 1e-12) and, with a gated
 * query, e_p = q^_p - q^_gate (model/deepmil.py:187-193), read off a block of vlsa_prepare_queries.  One launch, nq workgroups. */
int vlsa_query_chain(const float* dE, const void* qprep, int nq, int gated, int D, float* dQ, void* stream);

/* The per-bag backward of the same step -- what autograd runs per bag behind `torch.cat([net(x)[0] for x in bags])` in the
 * reference's training loop (runner/vlsa_handler.py:267-289; model/vlsa.py:181-198, model/deepmil.py:187-204 differentiated) --
 * as ONE host call: vlsa_head_backward_batch (B = 1) -> d rows, dW, db, dT, d logit_scale; vlsa_vlfan_backward + the
 * unnormalised vlsa_vlfan_merge -> d e [P, D]; the chain rule through e_p = q^_p - gated q^_gate, q^ = q / |q| -> dQ [nq, D].
 * Inputs are what vlsa_vlfan_forward_bag (pool_mode = mean) left behind for this bag: qprep / That / tnorm (prepared block),
 * out / m2 / l (aggregation), pooled / vhat / vnorm / logits (head).  g_vhat [D] / g_That [K, D]: gradients into the returned
 * unit features, or NULL.  W NULL = identity adapter (dW, db unused).  Scratch: head_ws (D + 1 floats), bwd_prep
 * (vlsa_bwd_prep_bytes), pm / pl ((G + 1) * 16 floats each), pacc (G * P * D floats), G = vlsa_num_partials(N). */
int vlsa_vlfan_backward_bag(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* qprep, int nq, int gated,
                            float coattn_scale, const float* dlogits, const float* g_vhat, const float* g_That,
                            const float* pooled, const float* vhat, const float* vnorm, const float* That, const float* tnorm,
                            const float* logits, const float* W, const float* logit_scale, const float* out, const float* m2,
                            const float* l, int K, float* head_ws, float* drows, float* dW, float* db, float* dT, float* dls,
                            void* bwd_prep, float* pm, float* pl, float* pacc, int G, float* dE, float* dQ, void* stream);

/* ---- batched forward: B bags per launch ------------------------------------------------------------------- */

/* One bag of a batch (device-resident array of these is passed to vlsa_vlfan_forward_batch). */
typedef struct vlsa_bag_desc {
    const void* X;   /* [N, 512] rows (bf16 or fp32: one dtype per batch), 16-byte aligned */
    int64_t N;       /* patches in this bag (>= 0) */
    int64_t ldx;     /* row stride in elements (>= 512, multiple of 8) */
} vlsa_bag_desc;

/* One fp32 [P, ld] matrix per bag (device-resident array of these): the optional per-bag score / attention-weight
 * outputs of the batched forward.  ptr NULL = this bag wants none.  ptr 16-byte aligned, ld % 4 == 0 and
 * ld >= N rounded up to a multiple of 64 (the streaming kernels store whole 4-row pieces; the pad columns get -inf / 0). */
typedef struct vlsa_rows_desc {
    float* ptr;
    int64_t ld;
} vlsa_rows_desc;

int vlsa_batch_max_bags(void);                                /* B <= this (64): every batched entry point */
/* B <= this (256) for the FORWARD launches (vlsa_vlfan_forward_batch[_attn], vlsa_vlfan_partial_batch*, vlsa_attn_normalise_batch):
 * an evaluation pass over slide-sized bags (the reference's TCGA bags hold 2-12k patches, runner/vlsa_handler.py:315-345 walks them
 * one by one) amortises the launch's fixed latency chain over four times more bags.  Above 64 bags the bags in flight `groups` are
 * raised to >= B / 64 (a workgroup keeps its own <= 64 bags in LDS), which also keeps the workspace at a 64-bag launch's size. */
int vlsa_batch_forward_max_bags(void);
int vlsa_batch_partials_per_bag(int B);                       /* partial records per bag the batch kernel leaves (256/S) */
size_t vlsa_batch_workspace_bytes(int B, int P, int D);       /* zero it ONCE after allocation; calls leave it reusable */

/*
 * B independent bags through the whole per-bag forward (model/vlsa.py:181-198 with cached text features, called once
 * per bag by runner/vlsa_handler.py:267-269,322-330) in THREE launches: one persistent streaming kernel that walks all
 * bags with the LDS-DMA ring running across bag boundaries, a batched merge and a batched incidence head.
 * Queries (qprep) and unit text features (That) are shared by the batch.  D == 512; x_dtype VLSA_DT_BF16 (split-bf16
 * MFMA kernel) or VLSA_DT_F32 (exact f32 MFMA kernel -- the reference's own feature format).
 * Outputs: m2, l [B,16]; out [B,P,D]; pooled, v, vhat [B,D]; vnorm [B]; logits, incidence (nullable) [B,K].
 */
/* Only the persistent streaming kernel of the batch (partials into `workspace`); vlsa_vlfan_forward_batch = this +
 * the batched merge + the batched head. */
int vlsa_vlfan_partial_batch(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P, void* workspace,
                             void* stream);
/* Same with two scheduling knobs:
 *  reserved_cus: compute units (rounded up to a multiple of the bags in flight) left without a persistent workgroup, so
 *    that the tail kernels of the previous batch and communication kernels of another stream (RCCL all-gather of the
 *    multi-GPU path) run concurrently: next to a persistent workgroup only kernels with <= 96 VGPRs and <= 8 KiB LDS
 *    get scheduled;
 *  groups: bags streamed concurrently (power of two <= min(B, 64); 0 = min(B, 8)).  Bag t is streamed by the
 *    workgroups / groups workgroups of group t % groups; fewer workgroups per bag = more rows per workgroup per bag
 *    epilogue and fewer partial records.  vlsa_batch_groups picks it from the bag sizes (HOST array rows_host[B]) by
 *    minimising the modelled duration of the slowest group.
 * Partials per bag = workgroups / groups: vlsa_batch_partials_per_bag_ex. */
int vlsa_batch_groups(const int64_t* rows_host, int B, int reserved_cus);
int vlsa_batch_partials_per_bag_ex(int B, int reserved_cus, int groups);
int vlsa_vlfan_partial_batch_ex(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P, void* workspace,
                                int reserved_cus, int groups, void* stream);
int vlsa_vlfan_forward_batch(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P, int pool_mode,
                             const float* pool_w, const float* W, const float* b, const float* That, int K,
                             const float* logit_scale, void* workspace, float* m2, float* l, float* out, float* pooled,
                             float* v, float* vhat, float* vnorm, float* logits, float* incidence, int reserved_cus,
                             int groups, void* stream);
/*
 * Attention weights out of the batched path -- what `VLFAN.forward(X, ret_with_attn=True)` returns per bag
 * (model/deepmil.py:198,206-215: A = softmax over the patches, [1, P, N]) and utils/model_inference.py:122 consumes:
 *  vlsa_vlfan_partial_batch_scores  = vlsa_vlfan_partial_batch_ex that also stores the log2-domain scores t_pn of every bag
 *                                     whose scores_desc[bag].ptr is non-NULL (48 B per patch at P = 12 on top of the row read);
 *  vlsa_attn_normalise_batch        = A[p, n] = exp2(t[p, n] - m2[bag, p]) / l[bag, p] for all bags in one launch (m2, l: [B, 16],
 *                                     bag-global, i.e. after the merge -- in the multi-GPU path after the all-gather);
 *                                     attn_desc may alias scores_desc (in place); max_N = largest N of the batch (host value);
 *  vlsa_vlfan_forward_batch_attn    = vlsa_vlfan_forward_batch + both of the above (scores_desc / attn_desc both NULL or both set).
 * scores_desc / attn_desc: device arrays of B vlsa_rows_desc.
 */
int vlsa_vlfan_partial_batch_scores(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P, void* workspace,
                                    int reserved_cus, int groups, const void* scores_desc, void* stream);
int vlsa_attn_normalise_batch(const void* bag_desc, int B, int P, int64_t max_N, const void* scores_desc, const float* m2,
                              const float* l, const void* attn_desc, void* stream);
int vlsa_vlfan_forward_batch_attn(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P, int pool_mode,
                                  const float* pool_w, const float* W, const float* b, const float* That, int K,
                                  const float* logit_scale, void* workspace, float* m2, float* l, float* out, float* pooled,
                                  float* v, float* vhat, float* vnorm, float* logits, float* incidence, int reserved_cus,
                                  int groups, const void* scores_desc, const void* attn_desc, int64_t max_N, void* stream);

/*
 * Backward of the aggregation for a BATCH of bags w.r.t. the (shared) effective queries -- one training step of the
 * reference back-propagates 32 bags through the same queries (runner/vlsa_handler.py:260-289, model/deepmil.py:187-200).
 * dout, out: [B, P, D]; m2, l: [B, 16] (the batched forward's outputs).  ONE persistent launch streams all bags and writes
 * G = vlsa_bwd_batch_partials() partial sums of  sum_bags de  into pm (= 0), pl (= 1) [G, 16] and pacc [G, P, D]; reduce them
 * with vlsa_vlfan_merge(..., G, normalise = 0).  bwd_prep: scratch of vlsa_bwd_batch_prep_bytes(B, D).
 * D == 512; bf16 bags with P <= 12, or fp32 bags (any P <= 16) -- VLSA_EUNSUPPORTED otherwise: vlsa_vlfan_backward_bags.
 * groups: bags in flight as in vlsa_vlfan_partial_batch_ex (0 = min(B, 8)).
 */
int vlsa_bwd_batch_partials(void);
size_t vlsa_bwd_batch_prep_bytes(int B, int D);
int vlsa_vlfan_backward_batch(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P,
                              float coattn_scale, const float* dout, const float* out, const float* m2, const float* l,
                              void* bwd_prep, float* pm, float* pl, float* pacc, int groups, void* stream);

/*
 * The same batched backward for what the persistent kernel does not take (fp32 bags -- the reference's own feature format -- or
 * P > 12): the per-bag kernel over the bag table in ONE launch, grid (G, B).  G: row blocks per bag (the largest
 * vlsa_num_partials(N_i) of the batch).  Writes B * G partial sums: pm (= 0), pl (= 1) [B * G, 16], pacc [B * G, P, D]; reduce with
 * vlsa_vlfan_merge(..., B * G, normalise = 0).  bwd_prep: vlsa_bwd_batch_prep_bytes(B, D).
 */
int vlsa_vlfan_backward_bags(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P, float coattn_scale,
                             const float* dout, const float* out, const float* m2, const float* l, void* bwd_prep, float* pm,
                             float* pl, float* pacc, int G, void* stream);

/*
 * Batched log-sum-exp merge with explicit strides (in floats): strides9 (HOST array) = {partial stride of pm, pl, pacc;
 * bag stride of pm, pl, pacc; bag stride of the outputs m2, l, out}.  Used by the multi-GPU batch path to fold the
 * workgroup partials of B bags into B compact records and, after the all-gather, the per-rank records into the result.
 */
int vlsa_vlfan_merge_batch_strided(const float* pm, const float* pl, const float* pacc, int B, int G, int P, int D,
                                   int normalise, const int64_t* strides9, float* m2, float* l, float* out, void* stream);

/*
 * Final merge + head of a batch in three ticket-free launches: log-sum-exp merge of G partial records per bag fused with the
 * query pooling (model/deepmil.py:133-150,203), v = W pooled + b for all bags (W rows kept in registers across 8 bags,
 * model/deepmil.py:204), then normalise + cosine logits + incidence per bag (model/vlsa.py:188-192).  strides9 as in
 * vlsa_vlfan_merge_batch_strided.  pool_mode MEAN / MAX / WEIGHT.  Outputs [B, ...].
 */
int vlsa_vlfan_merge_head_batch_strided(const float* pm, const float* pl, const float* pacc, int B, int G, int P, int D,
                                        const int64_t* strides9, int pool_mode, const float* pool_w, const float* W,
                                        const float* b, const float* That, int K, const float* logit_scale, float* m2,
                                        float* l, float* out, float* pooled, float* v, float* vhat, float* vnorm,
                                        float* logits, float* incidence, void* stream);

/* vlsa_head_forward for B bags in one launch: rows [B,P,D]; counters: B zeroed uint32; outputs [B, ...]. */
int vlsa_head_forward_batch(const float* rows, int B, int P, int D, int pool_mode, const float* pool_w, const float* W,
                            const float* b, const float* That, int K, const float* logit_scale, void* counters,
                            float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence,
                            void* stream);

/*
 * vlsa_normalize_rows(T) + vlsa_head_forward_batch (ticket-free route, B >= 1, pool_mode MEAN / MAX / WEIGHT) with the text normalisation
 * inside the pooling launch: the training step's head (model/vlsa.py:186-192 on the step's own text features, runner/vlsa_handler.py:267-281)
 * in three launches.  T [K, D] raw text features; writes That [K, D], tnorm [K] and the outputs of vlsa_head_forward_batch.
 */
int vlsa_head_forward_batch_text(const float* rows, int B, int P, int D, int pool_mode, const float* pool_w, const float* W,
                                 const float* b, const float* T, int K, const float* logit_scale, float* That, float* tnorm,
                                 float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence, void* stream);

/*
 * Backward of vlsa_head_forward_batch with pool_mode MEAN (the training step's tail: model/deepmil.py:203-204, model/vlsa.py:188-192
 * under autograd) in two launches: dlogits [B, K] (+ optional gradients g_vhat [B, D], g_That [K, D] flowing into the returned unit
 * features) -> drows [B, P, D], dW [D, D], db [D] (W NULL: identity adapter, no dW / db), dT [K, D] (w.r.t. the RAW text features:
 * tnorm [K] = their norms from vlsa_normalize_rows), dls [1] (w.r.t. the pre-exp logit scale).  pooled, vhat, vnorm, logits: the
 * forward's outputs.  workspace: (B * D + B) floats.
 */
int vlsa_head_backward_batch(const float* dlogits, const float* g_vhat, const float* g_That, const float* pooled, const float* vhat,
                             const float* vnorm, const float* That, const float* tnorm, const float* logits, const float* W,
                             const float* logit_scale, int B, int P, int D, int K, float* workspace, float* drows, float* dW,
                             float* db, float* dT, float* dls, void* stream);

/* ---- the other MIL encoders the VLSA wrapper accepts (FeatMIL, DeepMIL) and the zero-shot path ---------- */

/* Partials written by vlsa_scored_pool_partial / scratch rows of vlsa_colmax for N rows. */
int vlsa_pool_num_partials(int64_t N);

/*
 * Replaces: A = softmax(a, dim=N); out = A @ x (model/layers.py:115-116,146-147) and, with scores == NULL,
 * torch.mean(X, dim=1) (model/deepmil.py:57-58,271-272).  One query: pm/pl [G,16] (column 0 used),
 * pacc [G,1,D]; merge with vlsa_vlfan_merge(P = 1).  scores [N] are natural-log-domain raw attention scores.
 */
int vlsa_scored_pool_partial(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* scores,
                             float* pm, float* pl, float* pacc, void* stream);

/* Replaces: torch.max(X, dim=1) (model/deepmil.py:59-60,273-274). partials [G, D] scratch, out [D]. */
int vlsa_colmax(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, float* partials, float* out, void* stream);

/*
 * Replaces the elementwise part of Attention_Pooling / Gated_Attention_Pooling scoring (model/layers.py:110-113,
 * 144): a[n] = w2 . (tanh(H[n] + b1) [* sigmoid(Hg[n] + bg)]) + b2, H = x W1^T (and Hg = x Wg^T) being plain
 * [N,512]x[512,hid] GEMMs done by the caller (rocBLAS).  Hg/bg NULL => ungated.
 */
int vlsa_attn_scores(const float* H, const float* Hg, int64_t N, int hid, const float* b1, const float* bg,
                     const float* w2, const float* b2, float* a, void* stream);

/*
 * The DeepMIL encoder's N-sized part for a BATCH of bags (the reference's evaluation loop is encoder-agnostic,
 * runner/vlsa_handler.py:315-345; model/deepmil.py:261-292 per bag): ONE launch for the raw (gated-)attention scores of all
 * bags (vlsa_gated_scores_batch; weights packed by vlsa_prepare_gated_weights) and ONE for the softmax-weighted row sums
 * (vlsa_scored_pool_partial_batch -> G partials per bag in the P = 1 partial layout, folded by vlsa_vlfan_merge_batch_strided).
 * bag_desc: device table of vlsa_bag_desc; B <= 64; D == 512; bf16 or fp32 bags (one dtype per batch).
 *   tile_start [B + 1] int32 (device): first row tile of every bag for tiles of rows_per_tile rows (multiple of 16, at most the
 *   max_rows of vlsa_gated_scores_tiling); n_tiles = tile_start[B].   a: all bags' scores, bag b at a + a_off[b] (int64, device),
 *   a_floats long.
 * vlsa_gated_scores_tiling: the tile geometry of the score kernel for a (bag dtype, module) pair -- max_rows = rows of its largest
 *   tile, round_tiles = row tiles that fill the 256 CUs once; a batch smaller than one round is best served by the smallest
 *   rows_per_tile that still fits round_tiles.
 * vlsa_gated_scores_big_tile: large bf16 batches of the gated module are better served by the persistent LDS-DMA kernel (both
 *   operands staged through LDS, 256-row x 256-column tiles): *rows = its tile height (0: not for this dtype / module) and
 *   *min_total_rows = the batch size in rows from which to use it; a caller that does builds tile_start for rows_per_tile = *rows
 *   (any multiple of 32 above max_rows and up to 256 selects that kernel in vlsa_gated_scores_batch).
 */
int vlsa_gated_scores_tiling(int x_dtype, int gated, int* max_rows, int* round_tiles);
int vlsa_gated_scores_big_tile(int x_dtype, int gated, int* rows, int64_t* min_total_rows);
/* Scores AND attention pooling (model/layers.py:110-121,144-152: a_n, then sum_n softmax(a)_n x_n) of a batch of bf16 bags in ONE
 * launch of that kernel + the per-bag fold of its per-tile partials (one launch latency chain instead of two; the second pass over a
 * tile's rows and the pooling pass come through the L2 from the MALL / HBM again: profiles/r05_pmc_pool_traffic.json).  Arguments as vlsa_gated_scores_batch with
 * rows_per_tile a multiple of 32 in (max_rows, 256]; ws: n_tiles * 514 floats; pooled [B, 512] fp32; a is written whole. */
int64_t vlsa_gated_scores_pool_ws_floats(int64_t N);      /* ONE bag by pointer (bf16 or fp32; fp32: the same result from the
                                                            * fragment-order score kernel + pooling partials + merge, one host call): ws floats; pooled [512] */
int vlsa_gated_scores_pool(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, int gated, float* a, float* ws,
                           float* pooled, void* stream);
/* ... and DeepMIL's Adapter head (model/deepmil.py:283-286, as vlsa_adapter_head) behind it, same host call: W1 [R, 512], W2 [512, R],
 * R % 4 == 0; ws gets R more floats (the hidden row); logit [512]. */
int vlsa_gated_scores_pool_adapter(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, int gated, float* a,
                                   float* ws, float* pooled, const float* W1, int R, const float* W2, float keep_ratio, float* logit,
                                   void* stream);
int vlsa_gated_scores_pool_batch(const void* bag_desc, int B, int x_dtype, int D, const void* prep, int gated, const int* tile_start,
                                 int n_tiles, int rows_per_tile, float* a, const int64_t* a_off, float* ws, float* pooled,
                                 void* stream);
int vlsa_gated_scores_batch(const void* bag_desc, int B, int x_dtype, int D, const void* prep, int gated, const int* tile_start,
                            int n_tiles, int rows_per_tile, float* a, const int64_t* a_off, int64_t a_floats, void* stream);
int vlsa_scored_pool_partial_batch(const void* bag_desc, int B, int x_dtype, int D, const float* scores, const int64_t* a_off,
                                   int G, float* pm, float* pl, float* pacc, void* stream);

/*
 * Sentence assembly of the CoOp prompt learners in embedding space (reference model/prompt_learners/rank_prompt_learner.py:100-156,
 * plain_prompt_learner.py): out [R, L, dim] = the sentence template (<sot>, ".", <eot>, pad embeddings) with slot s of sentence
 * i (position s + 1) taken from source order[i, s]: < C -> context row (context [C, dim], or [R, C, dim] with ctx_per_rank),
 * >= C -> rank token row, interpolated over the n_base base ranks with interp [R, n_base] (rank [n_base, T, dim]) or, interp NULL,
 * rank [R, T, dim] directly; order [R, C + T] int32, -1 = unused slot.  Backward: pos [R, C + T] int32 = position of every
 * source in its sentence (-1: absent); writes dcontext (shape of context) and drank (shape of rank, n_rank_rows = its first dim).
 */
int vlsa_prompt_sentences(const float* templ, const float* context, int ctx_per_rank, const float* rank, const float* interp,
                          int n_base, const int* order, int R, int L, int S, int C, int T, int dim, float* out, void* stream);
int vlsa_prompt_sentences_backward(const float* dout, const int* pos, int ctx_per_rank, const float* interp, int n_base,
                                   int n_rank_rows, int R, int L, int S, int C, int T, int dim, float* dcontext, float* drank,
                                   void* stream);

/*
 * Exact Shapley values of the P text prototypes for the survival risk sum_k (K - k) softmax_k(logit_scale * mean_{p in S} sim[p, k])
 * with v(empty) = 1 (reference utils/model_inference.py:23-79, an O(P 2^P) host loop there).  sim: [P, K] fp32 decoupled
 * similarities (device); values: workspace of 2^P floats (the coalition values, kept for inspection); shap: [P].  P <= 16.
 */
int vlsa_prototype_shapley(const float* sim, int P, int K, float logit_scale, float* values, float* shap, void* stream);

/*
 * Feat_Projecter over all N patch rows of a bag: Y = LayerNorm(X W^T + b) * gamma + beta, W [512, 512]
 * (reference model/layers.py:65-82, applied by the encoders when use_feat_proj=True: model/deepmil.py:176-179,267-268).
 * vlsa_prepare_featproj packs the weights (bf16 hi + lo split, MFMA fragment order) into `prep`
 * (vlsa_featproj_prep_bytes() bytes) once per parameter version; vlsa_feat_project runs ONE kernel per bag: X bf16 or fp32
 * [N, ldx] (16-byte aligned rows), Y fp32 [N, ldy]; eps = the LayerNorm epsilon.  dim_in = dim_out = D = 512
 * (VLSA_EUNSUPPORTED otherwise: run the two torch modules).  b / gamma / beta may be NULL (0 / 1 / 0).
 */
size_t vlsa_featproj_prep_bytes(void);
int vlsa_prepare_featproj(const float* W, const float* b, const float* gamma, const float* beta, int dim_in, int dim_out,
                          void* prep, void* stream);
int vlsa_feat_project(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, float eps, float* Y,
                      int64_t ldy, void* stream);

/* Replaces forward_query_pooling with query_pooling = 'attention' | 'gated_attention' (model/deepmil.py:101-105,133-150: the
 * ABMIL modules of model/layers.py:85-153 applied to the P aggregated rows) for B bags in two launches: rows [B, P, D] ->
 * pooled [B, D] = softmax_P(a) @ rows, a_p = w2 . (tanh(Wa r_p + ba) [* sigmoid(Wg r_p + bg)]) + c; scores [B, P] (nullable) = a
 * (want_raw: Attention_Pooling's default return) or softmax_P(a) (Gated_Attention_Pooling's).  Wa, Wg [hid, D]; Wg / bg NULL:
 * ungated.  workspace: B * ((hid + 3) / 4) * 16 floats.  Feed pooled to vlsa_head_forward(..., VLSA_POOL_GIVEN). */
int vlsa_query_pool_attention(const float* rows, int B, int P, int D, const float* Wa, const float* ba, const float* Wg,
                              const float* bg, const float* w2, const float* c, int hid, int want_raw, void* workspace,
                              float* pooled, float* scores, void* stream);

/* out[n] = x_n . v  -- the N-sized piece of the attention-pooling backward. */
int vlsa_rowdot(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* v, float* out, void* stream);
/* The whole backward of pooled = softmax_N(a) @ X w.r.t. the raw scores a (model/layers.py:114-116,145-147 under autograd) in one
 * pass over X: da[n] = A_n (x_n . dpooled - pooled . dpooled), A_n = exp2(a_n log2(e) - m2[0]) / l[0] ((m2, l): the forward's
 * log2-domain softmax statistics from vlsa_vlfan_merge).  16-byte aligned rows, D % 8 == 0 (bf16) / % 4 (fp32). */
int vlsa_scored_pool_backward(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* a, const float* m2,
                              const float* l, const float* pooled, const float* dpooled, float* da, void* stream);

/*
 * Replaces: logits.topk(min(k, N), 0).values.mean(0) and logits.mean(0) (logit_pooling, model/deepmil.py:16-37)
 * on class-major scores S [C, N]; out[c] = out_scale * mean of the k largest of S[c, :] (k >= N: plain mean).
 * k <= 32 unless k >= N (VLSA_EUNSUPPORTED otherwise).
 */
int vlsa_topk_mean(const float* S, int C, int64_t N, int k, float out_scale, float* out, void* stream);
/* Same in two stages for long rows (N / 4096 chunks per class in parallel, then a merge of the chunk winners): what the
 * zero-shot path uses.  workspace: vlsa_topk_workspace_bytes(C, N, k), no initialisation needed. */
size_t vlsa_topk_workspace_bytes(int C, int64_t N, int k);
int vlsa_topk_mean_ws(const float* S, int C, int64_t N, int k, float out_scale, void* workspace, float* out, void* stream);

/* The k (<= 32) largest entries of every row of S [C, N], descending, padded with -inf when N < k: vals [C, k].  The per-rank
 * piece of the patch-sharded zero-shot pooling (each rank's local winners are all-gathered and re-selected with
 * vlsa_topk_mean on the [C, G * k] candidates).  workspace: vlsa_topk_workspace_bytes(C, N, k). */
int vlsa_topk_values(const float* S, int C, int64_t N, int k, void* workspace, float* vals, void* stream);

/* The zero-shot pooling of B bags in one launch: scores_desc[bag] = [C, ld] fp32 class scores (vlsa_rows_desc; e.g. the cosines
 * vlsa_vlfan_partial_batch_scores stores when the K text features are passed as queries with coattn_scale = 1 / log2(e)),
 * bag_desc gives N per bag; out [B, C] = exp(*logit_scale) * mean of the min(k, N) largest of each row (k <= 0: plain mean,
 * k <= 32 otherwise; logit_scale NULL: no scaling).  Replaces logit_pooling per bag (model/deepmil.py:16-37, model/vlsa.py:194-196). */
int vlsa_topk_mean_batch(const void* bag_desc, const void* scores_desc, int B, int C, int k, const float* logit_scale, float* out,
                         void* stream);

/* out[n, :] = X[n, :] / max(|X[n, :]|, 1e-12) for ALL N patch rows, fp32 out [N, D] (the image_features the reference's
 * zero-shot forward returns, model/vlsa.py:188-189); bf16 or fp32 rows, 16-byte aligned, D % 8 == 0 (bf16) / % 4 (fp32). */
int vlsa_normalize_many(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, float* out, void* stream);

/*
 * Raw attention scores of the ABMIL-style pooling modules over all N patches of a bag, fused (model/layers.py:85-153):
 *   gated:  a[n] = w2 . (tanh(Wa x_n + ba) * sigmoid(Wg x_n + bg)) + c        Wa, Wg: [dim_hid, dim_in] fp32
 *   plain:  a[n] = w2 .  tanh(Wa x_n + ba)                          + c        (Wg = bg = NULL, gated = 0)
 * vlsa_prepare_gated_weights packs the weights once (bf16 hi + lo split, MFMA-fragment order) into `prep`
 * (vlsa_gated_prep_bytes); vlsa_gated_scores streams the bag once: the [N, dim_hid] hidden activations stay in registers.
 * Feed a[] to vlsa_scored_pool_partial for softmax_N(a) @ X.  bf16 bags (consumed exactly) or fp32 bags (the reference's
 * own feature format; split into bf16 hi + lo on the fly, 1.5x the MFMA work); dim_in == 512, dim_hid == 256
 * (VLSA_EUNSUPPORTED otherwise: the host then uses library GEMMs + vlsa_attn_scores).
 */
size_t vlsa_gated_prep_bytes(int gated);
int vlsa_prepare_gated_weights(const float* Wa, const float* ba, const float* Wg, const float* bg, const float* w2,
                               const float* c, int dim_in, int dim_hid, int gated, void* prep, void* stream);
int vlsa_gated_scores(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, int gated, float* a,
                      void* stream);

/* DeepMIL's Adapter head on the pooled bag vector f [D] (model/deepmil.py:283-286, model/layers.py:50-62):
 * out = keep_ratio * f + (1 - keep_ratio) * relu(W2 relu(W1 f)), W1 [R, D], W2 [D, R] (bias-free); hidden: scratch [R]. */
int vlsa_adapter_head(const float* f, int D, const float* W1, int R, const float* W2, float keep_ratio, float* hidden,
                      float* out, void* stream);

/*
 * Bag ingest (replaces the per-step host concat + blocking H2D of dataset/PatchWSI.py:205-215 / runner/vlsa_handler.py:205):
 * pack N freshly uploaded rows (fp32 -> bf16 round-to-nearest-even, or bf16 copy) into a resident bf16 arena.
 * src [N, lds], dst [N, ldd] device pointers, 16-byte aligned rows, D % 8 == 0.
 */
int vlsa_pack_rows_bf16(const void* src, int src_dtype, int64_t N, int64_t lds, int D, void* dst, int64_t ldd,
                        void* stream);

/*
 * Loss tail of the training step on [B, K] bag predictions, value and gradient in one launch:
 * SurvIFMLE (loss/loss_surv.py:144-169: alpha, eps) and SurvEMD (loss/loss_surv_ext.py:43-109: p in {1, 2}, raw_distance;
 * logit_scale_exp = device pointer to exp(logit_scale), treated as a constant as the reference detaches it).
 * x: raw logits (from_logits = 1: the softmax converter of runner/vlsa_handler.py:245 is fused in) or incidences.
 * t: int64 bin per sample, e: 1 = event, 0 = censored.  out_ifmle / out_emd: per-sample losses [B] (either may be NULL);
 * grad [B, K] = d (w_ifmle * ifmle_i + w_emd * emd_i) / d x[i, :]  (NULL to skip).  K <= 64.
 */
int vlsa_surv_loss(const float* x, const int64_t* t, const float* e, int B, int K, int from_logits,
                   const float* logit_scale_exp, float alpha, float eps, int p, int raw_distance, float w_ifmle,
                   float w_emd, float* out_ifmle, float* out_emd, float* grad, void* stream);
/*
 * The handler's whole objective in ONE launch (runner/vlsa_handler.py:241-258: calc_objective_loss with 'mean'-reduced SurvIFMLE +
 * SurvEMD, loss/loss_surv.py:163-164, loss/loss_surv_ext.py:104-105): objective[0] = mean_i (w_ifmle * ifmle_i + w_emd * emd_i),
 * grad [B, K] = d objective / d x (NULL to skip).  logit_scale: device pointer to exp(logit_scale) as the reference's
 * net.get_logit_scale() hands it over (ls_is_log = 0), or to the raw parameter, exponentiated here (ls_is_log = 1).  B <= 4096.
 */
int vlsa_surv_objective(const float* x, const int64_t* t, const float* e, int B, int K, int from_logits, const float* logit_scale,
                        int ls_is_log, float alpha, float eps, int p, int raw_distance, float w_ifmle, float w_emd,
                        float* objective, float* grad, void* stream);

/*
 * The optimizer of the training step in ONE launch: torch.optim.Adam's update (runner/vlsa_handler.py:283-289 builds
 * optim.Adam(lr, weight_decay) through optim_factory.py:25-60; amsgrad = False, maximize = False, L2 weight decay added to the
 * gradient) over n_tensors fp32 tensors.  tensors: HOST array (read during the call: by-value kernel arguments); hyper: DEVICE float
 * table, {lr, weight_decay} per entry, tensors[i].hyper indexes it (a learning-rate schedule writes the table, captured launches read the
 * new values); state: DEVICE int[2], zeroed by the caller once: state[0] counts the steps taken (bias corrections), state[1] is a
 * ticket that is zero between launches.  More than VLSA_ADAM_MAX_TENSORS tensors go out as several launches; the last advances the count.
 */
#define VLSA_ADAM_MAX_TENSORS 16
typedef struct vlsa_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    int hyper;
} vlsa_adam_tensor;
int vlsa_adam_step(const vlsa_adam_tensor* tensors, int n_tensors, const float* hyper, int* state, double beta1, double beta2, double eps,
                   void* stream);

/* ---- text side: the frozen CoCa text tower on the prompts' compact rows -------------------------- */

/* One pre-LN transformer block (model/conch/transformer.py:191-247): DEVICE pointers to fp32 tensors in the layouts
 * nn.MultiheadAttention / nn.Linear store them ([out_features, in_features]). */
typedef struct vlsa_tt_layer {
    const float *ln1_w, *ln1_b;   /* ln_1            [width] */
    const float *in_w, *in_b;     /* attn.in_proj    [3 width, width], [3 width] */
    const float *out_w, *out_b;   /* attn.out_proj   [width, width], [width] */
    const float *ln2_w, *ln2_b;   /* ln_2            [width] */
    const float *fc_w, *fc_b;     /* mlp.c_fc        [4 width, width], [4 width] */
    const float *proj_w, *proj_b; /* mlp.c_proj      [width, 4 width], [width] */
} vlsa_tt_layer;

/* The tower (HOST struct; `layer` is a HOST array of `layers` entries).  width % 128 == 0, width <= 768, 64 features per head
 * (CONCH: 768 / 12 heads / 12 layers, model/conch/model_configs/conch_ViT-B-16.json), out_dim % 64 == 0. */
typedef struct vlsa_tt_model {
    int width, heads, layers, out_dim, ctx_len;  /* ctx_len = positions incl. the CLS slot (128) */
    const vlsa_tt_layer* layer;
    const float* pos_emb;    /* positional_embedding [ctx_len, width] */
    const float* cls_emb;    /* [width] */
    const float *lnf_w, *lnf_b; /* ln_final */
    const float* text_proj;  /* text_projection [width, out_dim] */
} vlsa_tt_model;

/* The compact rows of a batch of prompts (HOST struct of DEVICE int arrays, built once per prompt-length pattern).
 * A prompt whose CLS token may attend to positions 0..m contributes rows for positions 0..m followed by its CLS row:
 * causal attention makes every later position irrelevant to the pooled output (model/prompt_encoder.py:245-252,299-303).
 *   row_seq[row] prompt index; row_pos[row] position (CLS: ctx_len - 1); row_src[row] token index into prompts_embedding
 *   (-1: the CLS row); seq_row0[n_seq + 1] first row of each prompt; cls_keep[row] 1 if the prompt's CLS row attends to
 *   this row (for the CLS row itself: whether it attends to itself).  M rows, allocated M_pad (multiple of 48);
 *   max_len = most rows of one prompt (<= 128; backward <= 64). */
typedef struct vlsa_tt_rows {
    int n_seq, M, M_pad, max_len;
    const int *row_seq, *row_pos, *row_src, *seq_row0;
    const unsigned char* cls_keep;
    int prefix_len;   /* L > 0: the first L positions carry the SAME embedding in every prompt (<sot> + shared context tokens) and
                         are stored once, as rows 0 .. L-1 (row_seq 0); prompt s then owns rows [seq_row0[s], seq_row0[s + 1]) =
                         its positions L .. + its CLS row, seq_row0[0] == L; max_len counts the L prefix keys.  0: no sharing. */
} vlsa_tt_rows;

/*
 * Replaces CONCHPromptEncoder.forward (model/prompt_encoder.py:267-322): prompts_embedding [n_seq, ctx_len - 1, width]
 * (element [s, t] at emb + s * emb_seq_stride + t * emb_tok_stride, unit inner stride) -> text features out [n_seq, out_dim].
 * workspace: vlsa_tt_workspace_bytes(model, rows, save_for_backward) bytes, zeroed once by the caller; with
 * save_for_backward != 0 it keeps every block's inputs for vlsa_tt_backward, which turns dout [n_seq, out_dim] into d prompts_embedding (demb, same strides;
 * demb_floats = size of the whole demb allocation, zeroed here first).  The tower's own weights get no gradient from this call:
 * frozen in every shipped configuration (cfg_vlsa_conch.yaml:69); vlsa_tt_backward_train below is the training variant.
 */
/* The products read the weights from "tiled" (MFMA-fragment-major) copies: 1 KB per load instruction instead of 16 rows x 64 B.
 * vlsa_tt_pack_weights writes them into `packed` (vlsa_tt_packed_bytes(model, with_backward) bytes; with_backward also packs
 * the transposes the input-gradient products use) -- once per version of the (frozen) weights. */
size_t vlsa_tt_packed_bytes(const vlsa_tt_model* model, int with_backward);
int vlsa_tt_pack_weights(const vlsa_tt_model* model, void* packed, int with_backward, void* stream);
size_t vlsa_tt_workspace_bytes(const vlsa_tt_model* model, const vlsa_tt_rows* rows, int save_for_backward);
/* With VLSA_TT_PERSISTENT or-ed into `save_for_backward` (the library reads no environment; the Python mirror sets the bit when its
 * caller asks for it), CONCH-size towers with <= 112 compact rows run the 12 blocks of vlsa_tt_forward as
 * ONE persistent launch whose stages hand their activations over through in-kernel counters (text_tower.hip:
 * k_tt_forward_persistent; measured slower than the default launch-per-stage path on MI355X, hence opt-in).  Every wait in it is bounded; a time-out (a workgroup that never became resident because a foreign kernel held its
 * CU) is reported in four 32-bit words of the workspace: {code != 0, stage, workgroup, counter value} at this byte offset
 * (-1: this model / row plan takes the launch-per-stage path, which has no in-kernel waits).  The results of a launch whose
 * code is non-zero are void: the caller reads the words back (asynchronously is fine) and must not use them. */
#define VLSA_TT_PERSISTENT 0x100
int64_t vlsa_tt_status_offset(const vlsa_tt_model* model, const vlsa_tt_rows* rows, int save_for_backward);
int vlsa_tt_forward(const vlsa_tt_model* model, const vlsa_tt_rows* rows, const void* packed, const float* emb,
                    int64_t emb_seq_stride, int64_t emb_tok_stride, void* workspace, int save_for_backward, float* out,
                    void* stream);
int vlsa_tt_backward(const vlsa_tt_model* model, const vlsa_tt_rows* rows, const void* packed, const float* dout, void* workspace,
                     float* demb, int64_t emb_seq_stride, int64_t emb_tok_stride, int64_t demb_floats, void* stream);
/*
 * A tower whose OWN parameters train (`vlsa_txt_encoder_frozen: False`, runner/vlsa_handler.py:131 -- model/conch/transformer.py:191-247
 * and model/prompt_encoder.py:267-322 under autograd; off in every shipped configuration): vlsa_tt_forward with
 * save_for_backward == 2 also keeps every block's attention output, and this call produces, next to demb, the gradient of every tower
 * parameter.  `grads` is a vlsa_tt_model whose pointers are the gradient buffers (same shapes as the parameters: layer[i].in_w
 * [3 width, width] ..., pos_emb [ctx_len, width], cls_emb [width], lnf_w / lnf_b [width], text_proj [width, out_dim]); they are WRITTEN,
 * not accumulated (positions of pos_emb no compact row uses get zeros).  `packed` must hold the backward set of the CURRENT weights
 * (re-pack after every optimizer step).  Four dW = dY^T act products per block over the <= 128 compact rows (k_tt_dw, f32 MFMA),
 * bias / LayerNorm parameter gradients as fixed-order column sums: deterministic.
 */
int vlsa_tt_backward_train(const vlsa_tt_model* model, const vlsa_tt_rows* rows, const void* packed, const float* dout, void* workspace,
                           float* demb, int64_t emb_seq_stride, int64_t emb_tok_stride, int64_t demb_floats,
                           const vlsa_tt_model* grads, void* stream);

/* Diagnostics used by the GPU tests to pin hardware-layout assumptions (MFMA fragment / LDS tr-read). */
int vlsa_debug_probe(int which, void* out, size_t out_bytes, void* stream);

/*
 * ---- backward of the N-sized layers around the aggregation --------------------------------------
 * PyTorch autograd through the reference's modules keeps the [N, 256] / [N, 512] hidden activations of every bag alive and
 * runs two library GEMMs + ~10 elementwise kernels per layer; these entry points recompute the activations tile by tile on the
 * matrix pipe and contract dW = dH^T X in the same kernel (vlsa_amd/csrc/mlp_backward.hip).  All take B <= 64 bags per call:
 * the layers' weights are shared by the bags of an optimizer step (runner/vlsa_handler.py:260-289), so dW sums over them.
 *
 * vlsa_attn_scores_backward: Gated_Attention_Pooling / Attention_Pooling scores a_n = w2 . (tanh(Wa x_n + ba) [* sigmoid(Wg x_n
 *   + bg)]) + c (model/layers.py:103-122,137-153) given da = dL/da.  bag_desc: device table of vlsa_bag_desc (bf16 or fp32 rows,
 *   D == 512); tile_start [B + 1] int32 (device): first row tile of every bag in tiles of vlsa_mlp_bwd_tile_rows(x_dtype) rows,
 *   n_tiles = tile_start[B]; da: all bags' rows, bag b at da + a_off[b] (int64, device); prep: the block of
 *   vlsa_prepare_gated_weights; ws: vlsa_mlp_bwd_workspace_bytes(gated, n_tiles) bytes.
 *   Out: dW [gated ? 2 : 1][256][512] (dWa, dWg); dvec [3][512]: row 0 = (dba | dbg), row 1 = dw2 [256], dvec[2][0] = dc.
 * vlsa_feat_project_train / _rowstats / _backward: Feat_Projecter y_n = LayerNorm(W x_n + b) gamma + beta (model/layers.py:65-82).
 *   The training forward also stores (row mean, rstd) in stats [N][4] columns 0, 1; _rowstats fills columns 2, 3 from the upstream
 *   gradient dy [N, 512] and the projected rows y (c1 = mean(dy gamma), c2 = mean(dy (y - beta))); _backward (mode 2 workspace)
 *   takes tables of the input rows and of dy, stats of all bags (bag b at row row_off[b]) and writes dW [512][512] and
 *   dvec [3][512] = (db, dgamma, dbeta).
 * vlsa_vlfan_backward_dx: dL/dX of the cross attention (model/deepmil.py:187-200) for fp32 bags -- what a trainable
 *   Feat_Projecter in front of VLFAN needs (model/deepmil.py:176-179).  dx_desc: table of the fp32 gradient rows to write;
 *   tile_start in 64-row super tiles; qprep of vlsa_prepare_queries; dout / out [B][P][512], m2 / l [B][16] of the forward;
 *   delta_ws: B * 16 floats.
 */
int vlsa_mlp_bwd_tile_rows(int x_dtype);
size_t vlsa_mlp_bwd_workspace_bytes(int mode, int n_tiles);   /* mode: 0 attention, 1 gated attention, 2 Feat_Projecter */
int vlsa_attn_scores_backward(const void* bag_desc, int B, int x_dtype, int D, const void* prep, int gated, const int* tile_start,
                              int n_tiles, const float* da, const int64_t* a_off, void* ws, float* dW, float* dvec, float drop_p,
                              unsigned int seed, void* stream);
/* Training-mode forward of the gated scores: nn.Dropout(drop_p) behind tanh and behind sigmoid (model/layers.py:94,99) with a
 * counter-based mask generator keyed on (seed, row, hidden unit) that vlsa_attn_scores_backward(drop_p, seed) re-evaluates; the
 * masks are Bernoulli(1 - drop_p) bits of a hash, not torch's Philox stream (bit parity with the reference's dropout is
 * impossible either way: SURVEY.md 7.4-8).  drop_p = 0 or gated = 0: identical to vlsa_gated_scores. */
int vlsa_gated_scores_train(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, int gated, float* a,
                            float drop_p, unsigned int seed, void* stream);
int vlsa_feat_project_train(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* prep, float eps, float* Y,
                            int64_t ldy, float* stats, void* stream);
int vlsa_feat_project_rowstats(const float* dy, int64_t lddy, const float* y, int64_t ldy, int64_t N, const void* prep, float* stats,
                               void* stream);
int vlsa_feat_project_backward(const void* bag_desc, const void* dy_desc, int B, int x_dtype, const void* prep, const int* tile_start,
                               int n_tiles, const float* stats, const int64_t* row_off, void* ws, float* dW, float* dvec,
                               void* stream);
/* dL/dX of the (gated) attention pooling over the patches, for bags that carry a gradient (a trainable Feat_Projecter feeding a
 * DeepMIL encoder, model/deepmil.py:267-283): dx_n = dHa_n Wa + dHg_n Wg + A_n dpooled.  prep_t: the un-scaled weights packed
 * [hidden][column] by vlsa_prepare_attn_dx_weights (vlsa_attn_dx_prep_bytes); da / aw: dL/da and the softmax weights of all bags'
 * rows (bag b at a_off[b]); dpooled [B][512]; aw = dpooled = NULL: scores term only.  Tiles as vlsa_attn_scores_backward. */
size_t vlsa_attn_dx_prep_bytes(int gated);
int vlsa_prepare_attn_dx_weights(const float* Wa, const float* Wg, int gated, void* prep_t, void* stream);
int vlsa_attn_scores_backward_dx(const void* bag_desc, const void* dx_desc, int B, int x_dtype, int D, const void* prep, const void* prep_t,
                                 int gated, const int* tile_start, int n_tiles, const float* da, const float* aw, const float* dpooled,
                                 const int64_t* a_off, float drop_p, unsigned int seed, void* stream);
int vlsa_vlfan_backward_dx(const void* bag_desc, const void* dx_desc, int B, int D, const void* qprep, int P, float coattn_scale,
                           const int* tile_start, int n_tiles, const float* dout, const float* out, const float* m2, const float* l,
                           float* delta_ws, void* stream);

/* Device-side descriptor tables of ONE bag for the *_backward entry points above (bag_desc [1], optional second table, row
 * offset [1], tile_start [2]) written from by-value arguments by a one-thread kernel: dst = 64 bytes of device memory.  Returns the
 * number of tiles (> 0) or a negative error code.  Layout: {X, N, ld} {extra, N, extra_ld}? {0} {int32 0, int32 n_tiles}. */
int vlsa_fill_one_bag_tables(void* dst, const void* X, int64_t N, int64_t ld, const void* extra, int64_t extra_ld, int tile_rows,
                             void* stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Peer-write exchange of the patch-sharded multi-GPU path (csrc/xchg.hip).  The reference has no multi-GPU code
 * ("NCCL call sites: none"); the payload is the per-query partial sums of model/deepmil.py:198-200 as compact
 * records [m2(16) | l(16) | acc(P*D)] and the owner's head outputs (model/vlsa.py:188-192).  One process per GPU; every rank
 * exports ONE fine-grained buffer, maps its peers' buffers, and the data path is kernels writing over xGMI + epoch flags.
 *
 * Set-up calls (the ONLY entry points that allocate / synchronise; once per plan, never on the hot path):
 *   vlsa_xchg_alloc   zeroed device buffer of `bytes` (uncached, else fine-grained, else plain: *kind = 2 / 1 / 0) + its 64-byte
 *                     hipIpcMemHandle_t in handle64 (HOST memory).  VLSA_EUNSUPPORTED: the driver refuses IPC for it.
 *   vlsa_xchg_open    map a peer's handle (another process, same or another GPU of the node) -> device address
 *   vlsa_xchg_close / vlsa_xchg_free
 * Data path (enqueue only; the pointer tables are HOST arrays of `world` device addresses, `status` a zeroed device uint32 that
 * collects time-out bits 1 (put gate) | 2 (wait) | 4 (collect); timeout_ticks in 100 MHz wall-clock ticks):
 *   vlsa_xchg_put     block d: [wait gate[d] >= gate_epoch], copy n16[d] x 16 B  src[d] -> dst[d], system fence,
 *                     flag[d] = epoch (and ack[d] = epoch).  ack / gate nullable (table or entry).
 *   vlsa_xchg_wait    until every flag[r] >= epoch
 *   vlsa_xchg_collect block o: [wait flag[o] >= epoch]; owner o's result box (vlsa_xchg_result_floats layout, count[o] bags) ->
 *                     logits [B, K], incidence [B, K] (nullable), vhat [B, D], m2 / l [B, 16] at bag j * world + o (the caller's bag
 *                     order), m2_local / l_local (nullable pair) at start[o] + j (the owner-major order of the local bag table);
 *                     [ack[o] = epoch].  flag / ack tables nullable: the boxes then hold the output of a collective. */
int vlsa_xchg_max_peers(void);
size_t vlsa_xchg_result_floats(int nmax, int K, int D, int64_t* offsets5);
int vlsa_xchg_alloc(size_t bytes, void** ptr, void* handle64, int* kind);
int vlsa_xchg_open(const void* handle64, void** ptr);
int vlsa_xchg_close(void* ptr);
int vlsa_xchg_free(void* ptr);
int vlsa_xchg_put(int world, const void* const* src, void* const* dst, const uint32_t* n16, void* const* flag, void* const* ack,
                  const void* const* gate, uint32_t epoch, uint32_t gate_epoch, int64_t timeout_ticks, void* status, void* stream);
int vlsa_xchg_wait(int world, const void* const* flag, uint32_t epoch, int64_t timeout_ticks, void* status, void* stream);
int vlsa_xchg_collect(int world, const void* const* box, const void* const* flag, void* const* ack, const int* count,
                      const int* start, int nmax, int K, int D, uint32_t epoch, int64_t timeout_ticks, float* logits,
                      float* incidence, float* vhat, float* m2, float* l, float* m2_local, float* l_local, void* status,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLSA_HIP_H */

// Tail of the per-bag forward: partial merge, attention-weight normalisation, row normalisation and the
// bag-level incidence head (query pooling -> visual adapter -> cosine logits against the rank prompts).
// All of it is P x D / D x D sized work; the design goal is few launches and no single-workgroup serial
// stretch longer than a few microseconds (MI355X_MICROARCH.md: kernel boundary ~1.5-1.9 us).
#include "vlsa_common.h"

namespace vlsa {

// ---------------------------------------------------------------------------------------------------
// Merge G partials.  grid = (ceil(D/64), P): workgroup (cc, p) owns 64 columns of query p and reads G
// segments of 256 B.  thread -> (float4 column c4 = tid & 15, partial subset gs = tid >> 4).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vlfan_merge(const float* __restrict__ pm, const float* __restrict__ pl,
                                                      const float* __restrict__ pacc, int G, int P, int D,
                                                      int normalise, float* __restrict__ m2, float* __restrict__ l,
                                                      float* __restrict__ out, int64_t sm, int64_t sl_, int64_t sa) {
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float4 sacc[16][16];
    __shared__ float sl[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = blockIdx.y, c0 = blockIdx.x * 64;
    const int c4 = tid & 15, gs = tid >> 4;
    const int col = c0 + c4 * 4;
    const bool incol = col < D;

    // One round of independent loads per thread (its partials gs, gs+16, ...): the partial maxima, the partial
    // normalisers and the 16-byte accumulator pieces are all in flight together; the exp2 factors follow.
    constexpr int U = 16;  // up to 256 partials per pass
    float mx = -INFINITY;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float lt = 0.f;
    // pass 1: global max over all G partials (each thread scans a strided subset; G is small)
    for (int gI = tid; gI < G; gI += 256) mx = fmaxf(mx, pm[(size_t)gI * sm + p]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int g0 = gs; g0 < G; g0 += 16 * U) {
        float mg[U], lg[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int gI = g0 + 16 * u;
            const bool ok = gI < G;
            mg[u] = ok ? pm[(size_t)gI * sm + p] : -INFINITY;
            lg[u] = ok ? pl[(size_t)gI * sl_ + p] : 0.f;
            v[u] = (ok && incol) ? *reinterpret_cast<const float4*>(pacc + (size_t)gI * sa + (size_t)p * D + col)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float f = (mg[u] == -INFINITY) ? 0.f : fast_exp2(mg[u] - mx);
            lt += lg[u] * f;
            a.x += v[u].x * f; a.y += v[u].y * f; a.z += v[u].z * f; a.w += v[u].w * f;
        }
    }
    sacc[gs][c4] = a;
    if (c4 == 0) sl[gs] = lt;
    __syncthreads();
    if (tid < 16) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        float ls = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 v = sacc[k][tid];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            ls += sl[k];
        }
        if (normalise) {
            const float inv = 1.f / ls;
            s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
        }
        const int cc = c0 + tid * 4;
        if (cc < D) *reinterpret_cast<float4*>(out + (size_t)p * D + cc) = s;
        if (tid == 0 && blockIdx.x == 0) {
            m2[p] = mx;
            l[p] = ls;
        }
    }
}

__global__ __launch_bounds__(256) void k_attn_normalise(const float* __restrict__ scores, int P, int64_t N,
                                                         const float* __restrict__ m2, const float* __restrict__ l,
                                                         float* __restrict__ A) {
    const int p = blockIdx.y;
    const float mp = m2[p], inv = 1.f / l[p];
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256)
        A[(size_t)p * N + n] = fast_exp2(scores[(size_t)p * N + n] - mp) * inv;
}

__global__ __launch_bounds__(256) void k_normalize_rows(const float* __restrict__ in, int D, float* __restrict__ out,
                                                         float* __restrict__ norms) {
    __shared__ float red[4];
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* x = in + (size_t)r * D;
    float ss = 0.f;
    for (int d = tid; d < D; d += 256) ss += x[d] * x[d];
    ss = block_sum_256(ss, red);
    const float nrm = fmaxf(sqrtf(ss), kNormEps);
    for (int d = tid; d < D; d += 256) out[(size_t)r * D + d] = x[d] / nrm;
    if (norms != nullptr && tid == 0) norms[r] = nrm;
}

// ---------------------------------------------------------------------------------------------------
// Incidence head.  grid = NB workgroups; workgroup j computes rows [8j, 8j+8) of v = W pooled + b
// (pooled is recomputed per workgroup from the P x D rows: 24 KB of L2 reads), publishes them, takes a
// ticket; the last arriver normalises v and scores it against the K unit rank-prompt embeddings.
// Hand-off follows the release/acquire recipe of cdna_hip_programming.md Guideline 16.
// ---------------------------------------------------------------------------------------------------
constexpr int kHeadRowsPerBlock = 8;

__device__ __forceinline__ float pooled_col(const float* __restrict__ rows, int P, int D, int c, int pool_mode,
                                            const float* pw) {
    // all P <= 16 row loads are issued together (fully unrolled, predicated): one memory latency, not P of them
    float x[VLSA_MAX_P];
#pragma unroll
    for (int p = 0; p < VLSA_MAX_P; ++p) x[p] = p < P ? rows[(size_t)p * D + c] : 0.f;
    if (pool_mode == VLSA_POOL_MEAN) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < VLSA_MAX_P; ++p) s += x[p];  // same left-to-right order as torch.mean's sum for P <= 16
        return s / (float)P;
    }
    if (pool_mode == VLSA_POOL_MAX) {
        float s = -INFINITY;
#pragma unroll
        for (int p = 0; p < VLSA_MAX_P; ++p) s = p < P ? fmaxf(s, x[p]) : s;
        return s;
    }
    if (pool_mode == VLSA_POOL_WEIGHT) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < VLSA_MAX_P; ++p) s += p < P ? pw[p] * x[p] : 0.f;
        return s;
    }
    return x[0];  // VLSA_POOL_GIVEN
}

__global__ __launch_bounds__(256) void k_head(const float* __restrict__ rows, int P, int D, int pool_mode,
                                               const float* __restrict__ pool_w, const float* __restrict__ W,
                                               const float* __restrict__ bias, const float* __restrict__ That, int K,
                                               const float* __restrict__ logit_scale, unsigned int* counter,
                                               float* pooled, float* v, float* __restrict__ vhat,
                                               float* __restrict__ vnorm, float* __restrict__ logits,
                                               float* __restrict__ incidence, int NB) {
    __shared__ float sp[VLSA_MAX_D];
    __shared__ float spw[VLSA_MAX_P];
    __shared__ float slog[VLSA_MAX_K];
    __shared__ float red[4];
    __shared__ int s_last;
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);  // short kernel that may co-run with a persistent streaming kernel: win issue arbitration
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {   // batched launch: blockIdx.y selects the bag; every per-bag pointer is advanced here
        const int bag = blockIdx.y;
        rows += (size_t)bag * P * D;
        counter += bag;
        pooled += (size_t)bag * D;
        v += (size_t)bag * D;
        vhat += (size_t)bag * D;
        vnorm += bag;
        logits += (size_t)bag * K;
        if (incidence != nullptr) incidence += (size_t)bag * K;
    }

    // W rows of this workgroup first: their loads are in flight while the pooled vector is being formed
    float4 wq[2][VLSA_MAX_D / 256];
    float bj[2] = {0.f, 0.f};
    if (W != nullptr) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int j = blockIdx.x * kHeadRowsPerBlock + wv * 2 + rr;
#pragma unroll
            for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
                const int c = lane * 4 + 256 * i;
                wq[rr][i] = (j < D && c < D) ? *reinterpret_cast<const float4*>(W + (size_t)j * D + c)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (j < D && bias != nullptr) bj[rr] = bias[j];
        }
    }
    if (pool_mode == VLSA_POOL_WEIGHT) {  // softmax over the raw 'weight' parameter (model/deepmil.py:148)
        if (tid == 0) {
            float mx = -INFINITY, s = 0.f;
            for (int p = 0; p < P; ++p) mx = fmaxf(mx, pool_w[p]);
            for (int p = 0; p < P; ++p) { spw[p] = expf(pool_w[p] - mx); s += spw[p]; }
            for (int p = 0; p < P; ++p) spw[p] /= s;
        }
        __syncthreads();
    }
    for (int c = tid; c < D; c += 256) {
        const float pc = pooled_col(rows, P, D, c, pool_mode, spw);
        sp[c] = pc;
        if (blockIdx.x == 0 && pooled != rows) pooled[c] = pc;  // rows == pooled: already pooled by the merge kernel
    }
    __syncthreads();

    // publish v write-through (sc1: relaxed agent-scope atomic stores), so no release fence is needed
    if (W != nullptr) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int j = blockIdx.x * kHeadRowsPerBlock + wv * 2 + rr;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
                const int c = lane * 4 + 256 * i;
                if (c < D) s += wq[rr][i].x * sp[c] + wq[rr][i].y * sp[c + 1] + wq[rr][i].z * sp[c + 2] + wq[rr][i].w * sp[c + 3];
            }
            s = wave_sum(s);
            if (lane == 0 && j < D) __hip_atomic_store(v + j, s + bj[rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        for (int c = tid; c < D; c += 256) __hip_atomic_store(v + c, sp[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (unsigned int)(NB - 1));
    }
    __syncthreads();
    if (!s_last) return;

    // ---- last arriver: v^ = v / max(|v|, eps); logits = exp(ls) * v^ . T^_k; incidence = softmax ----
    // v is read with agent-scope (sc1) loads: L1 is bypassed, so no acquire fence is needed either.
    float ss = 0.f;
    for (int c = tid; c < D; c += 256) {
        const float x = __hip_atomic_load(v + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sp[c] = x;
        ss += x * x;
    }
    ss = block_sum_256(ss, red);
    const float nrm = fmaxf(sqrtf(ss), kNormEps);
    for (int c = tid; c < D; c += 256) {
        const float u = sp[c] / nrm;
        sp[c] = u;
        vhat[c] = u;
    }
    if (tid == 0) vnorm[0] = nrm;
    __syncthreads();
    const float ls = expf(logit_scale[0]);
    for (int k = wv; k < K; k += 4) {
        const float* tk = That + (size_t)k * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += sp[c] * tk[c];
        s = wave_sum(s);
        if (lane == 0) {
            const float lg = ls * s;
            logits[k] = lg;
            slog[k] = lg;
        }
    }
    __syncthreads();
    if (incidence != nullptr && tid == 0) {
        float mx = -INFINITY, s = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, slog[k]);
        for (int k = 0; k < K; ++k) s += expf(slog[k] - mx);
        for (int k = 0; k < K; ++k) incidence[k] = expf(slog[k] - mx) / s;
    }
    if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ticket back to zero
}

// ---------------------------------------------------------------------------------------------------
// Batched head in two ticket-free kernels (the batch path has the pooled vectors from the merge kernel):
//   k_head_linear: v[bag] = W pooled[bag] + b.  grid (D/8, ceil(B/8)): a workgroup keeps 8 rows of W in registers
//                  (2 per wave) and walks 8 bags, whose pooled vectors are all loaded up front (one L2 round trip).
//   k_head_finish: one workgroup per bag: v^ = v / max(|v|, eps); logits = exp(ls) v^ . T^_k; incidence = softmax.
// ---------------------------------------------------------------------------------------------------
constexpr int kHeadBagsPerBlock = 8;

__global__ __launch_bounds__(256) void k_head_linear(const float* __restrict__ pooled, int B, int D,
                                                      const float* __restrict__ W, const float* __restrict__ bias,
                                                      float* __restrict__ v) {
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int bag0 = blockIdx.y * kHeadBagsPerBlock;
    float4 wq[2][VLSA_MAX_D / 256];
    float bj[2] = {0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int j = blockIdx.x * kHeadRowsPerBlock + wv * 2 + rr;
#pragma unroll
        for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
            const int c = lane * 4 + 256 * i;
            wq[rr][i] = (j < D && c < D) ? *reinterpret_cast<const float4*>(W + (size_t)j * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (j < D && bias != nullptr) bj[rr] = bias[j];
    }
    float4 x[kHeadBagsPerBlock][VLSA_MAX_D / 256];
#pragma unroll
    for (int t = 0; t < kHeadBagsPerBlock; ++t)
#pragma unroll
        for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
            const int c = lane * 4 + 256 * i;
            x[t][i] = (bag0 + t < B && c < D) ? *reinterpret_cast<const float4*>(pooled + (size_t)(bag0 + t) * D + c)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
    for (int t = 0; t < kHeadBagsPerBlock; ++t)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < VLSA_MAX_D / 256; ++i)
                s += wq[rr][i].x * x[t][i].x + wq[rr][i].y * x[t][i].y + wq[rr][i].z * x[t][i].z + wq[rr][i].w * x[t][i].w;
            s = wave_sum(s);
            const int j = blockIdx.x * kHeadRowsPerBlock + wv * 2 + rr;
            if (lane == 0 && j < D && bag0 + t < B) v[(size_t)(bag0 + t) * D + j] = s + bj[rr];
        }
}

// v = pooled W^T + b for MANY bags on the f32 matrix pipe (round 5; D == 512): workgroup (bag tile of 16, output tile of 32),
// 8 waves, wave w contracts columns [64 w, 64 w + 64) with v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: no split, no
// rounding beyond fp32) -- A = pooled (M = bag), B = W rows (N = output), one float4 per lane and fragment feeds four k-steps
// (k-slot g of step t <-> column 16 it + 4 g + t, the same map on both sides).  All 12 float4 loads of a wave are in flight
// at once; the eight partial tiles are summed through 8 KiB of LDS in a fixed order.  k_head_linear (VALU, 2 048 workgroups at
// B = 256) took 26.7 us of a 168 us step of 256 slide-sized bags.  <= 8 KiB LDS, <= 96 VGPRs: co-resides with a persistent kernel.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_head_linear_mfma(const float* __restrict__ pooled, int B, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ v) {
    constexpr int D = 512;
    __shared__ __attribute__((aligned(16))) float sred[4][64][8];   // 8 KiB
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int bag0 = blockIdx.x * 16, j0 = blockIdx.y * 32;
    const int bag = min(bag0 + i16, B - 1);
    const float* pa = pooled + (size_t)bag * D + wv * 64 + 4 * g;
    const float* pb0 = W + (size_t)(j0 + i16) * D + wv * 64 + 4 * g;
    const float* pb1 = pb0 + (size_t)16 * D;
    float4 a[4], b0[4], b1[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        a[it] = *reinterpret_cast<const float4*>(pa + 16 * it);
        b0[it] = *reinterpret_cast<const float4*>(pb0 + 16 * it);
        b1[it] = *reinterpret_cast<const float4*>(pb1 + 16 * it);
    }
    f32x4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, b0[it].x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].x, b1[it].x, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, b0[it].y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].y, b1[it].y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, b0[it].z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].z, b1[it].z, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, b0[it].w, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it].w, b1[it].w, c1, 0, 0, 0);
    }
    // fixed-order tree over the 8 column slices: (w, w + 4), then (w, w + 2), then (0, 1)
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wv >= half && wv < 2 * half) {
            float* dst = sred[wv - half][lane];
            *reinterpret_cast<f32x4_t*>(dst) = c0;
            *reinterpret_cast<f32x4_t*>(dst + 4) = c1;
        }
        __syncthreads();
        if (wv < half) {
            const float* src = sred[wv][lane];
            c0 += *reinterpret_cast<const f32x4_t*>(src);
            c1 += *reinterpret_cast<const f32x4_t*>(src + 4);
        }
        __syncthreads();
    }
    if (wv != 0) return;
    const float bj0 = bias != nullptr ? bias[j0 + i16] : 0.f, bj1 = bias != nullptr ? bias[j0 + 16 + i16] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int bg = bag0 + 4 * g + r;      // D fragment: lane (i16, g) holds rows 4 g + r, column i16
        if (bg < B) {
            v[(size_t)bg * D + j0 + i16] = c0[r] + bj0;
            v[(size_t)bg * D + j0 + 16 + i16] = c1[r] + bj1;
        }
    }
}

__global__ __launch_bounds__(256) void k_head_finish(const float* __restrict__ v, int D, const float* __restrict__ That, int K,
                                                      const float* __restrict__ logit_scale, float* __restrict__ vhat,
                                                      float* __restrict__ vnorm, float* __restrict__ logits,
                                                      float* __restrict__ incidence) {
    __shared__ float sp[VLSA_MAX_D];
    __shared__ float slog[VLSA_MAX_K];
    __shared__ float red[4];
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, bag = blockIdx.x;
    v += (size_t)bag * D;
    vhat += (size_t)bag * D;
    logits += (size_t)bag * K;
    // the text rows this wave will need (k = wv, wv + 4, ...: up to 4 rows of D <= 512 here, else read in place) and the logit scale are
    // requested together with v: they do not depend on the norm, and behind it they were one more round of dependent loads per k
    constexpr int kTR = 4, kTC = 8;
    const bool pre = D <= 64 * kTC && K <= 4 * kTR;
    float tr[kTR][kTC];
#pragma unroll
    for (int i = 0; i < kTR; ++i)
#pragma unroll
        for (int q = 0; q < kTC; ++q) {
            const int k = wv + 4 * i, c = lane + 64 * q;
            tr[i][q] = (pre && k < K && c < D) ? That[(size_t)k * D + c] : 0.f;
        }
    const float ls_raw = logit_scale[0];
    float ss = 0.f;
    for (int c = tid; c < D; c += 256) {
        const float x = v[c];
        sp[c] = x;
        ss += x * x;
    }
    ss = block_sum_256(ss, red);
    const float nrm = fmaxf(sqrtf(ss), kNormEps);
    for (int c = tid; c < D; c += 256) {
        const float u = sp[c] / nrm;
        sp[c] = u;
        vhat[c] = u;
    }
    if (tid == 0) vnorm[bag] = nrm;
    __syncthreads();
    const float ls = expf(ls_raw);
    for (int k = wv; k < K; k += 4) {
        const float* tk = That + (size_t)k * D;
        float s = 0.f;
        if (pre) {
            const int i = (k - wv) >> 2;
#pragma unroll
            for (int ii = 0; ii < kTR; ++ii)
                if (ii == i) {
#pragma unroll
                    for (int q = 0; q < kTC; ++q)
                        if (lane + 64 * q < D) s += sp[lane + 64 * q] * tr[ii][q];
                }
        } else {
            for (int c = lane; c < D; c += 64) s += sp[c] * tk[c];
        }
        s = wave_sum(s);
        if (lane == 0) {
            const float lg = ls * s;
            logits[k] = lg;
            slog[k] = lg;
        }
    }
    __syncthreads();
    if (incidence != nullptr && tid == 0) {
        float mx = -INFINITY, s = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, slog[k]);
        for (int k = 0; k < K; ++k) s += expf(slog[k] - mx);
        for (int k = 0; k < K; ++k) incidence[(size_t)bag * K + k] = expf(slog[k] - mx) / s;
    }
}

// ---------------------------------------------------------------------------------------------------
// One slide per call (round 5): the tail as TWO ticket-free launches.  Until round 4 it was k_vlfan_merge (5 us) and
// the ticketed k_head (15.5 us at 50k: W rows -> pooled -> write-through v -> drain -> ticket -> last arriver re-reads
// v -> logits, every step a dependent cross-XCD round trip).  The adapter is linear, so
//     v = W mean_p(out_p) + b = b + sum_p wgt_p sum_cc W[:, cols(cc)] out_p[cols(cc)]
// and the merge workgroup (cc, p), which holds the 64 merged columns of query p anyway, can multiply them with its
// 512 x 64 slice of W (128 KB of coalesced L2 reads, in flight WHILE the partials are being merged).
//   k_vlfan_merge_wpart   grid (8, P) x 256: k_vlfan_merge + vpart[cc][p][j] = W[j, cols] . out_p[cols]   (D == 512)
//   k_head_finish_parts   ONE workgroup x 512: v = b + sum_p wgt_p sum_cc vpart (fixed order), pooled, normalise, K cosine
//                         logits, incidence.
// mean / softmax(weight) query pooling with a Linear adapter; everything else keeps the old route.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vlfan_merge_wpart(const float* __restrict__ pm, const float* __restrict__ pl,
                                                            const float* __restrict__ pacc, int G, int P, float* __restrict__ m2,
                                                            float* __restrict__ l, float* __restrict__ out,
                                                            const float* __restrict__ W, float* __restrict__ vpart) {
    // contiguous partials (pm / pl [G, 16], pacc [G, P, 512]); blockIdx.z = quarter of W's rows: every quarter repeats the (cheap,
    // L2-resident) merge of its (cc, p) piece and multiplies it with 128 rows of W -- 8 float4 of W per thread instead of 32
    constexpr int D = 512;
    constexpr unsigned sm = kPStride, sl_ = kPStride;
    const unsigned sa = (unsigned)P * D;
    const int jq = blockIdx.z;
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float4 sacc[16][16];
    __shared__ float sl[16];
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = blockIdx.y, c0 = blockIdx.x * 64;
    const int c4 = tid & 15, gs = tid >> 4;
    const int col = c0 + c4 * 4;

    // this thread's slice of W: rows jq * 128 + gs * 8 .. + 7, its four columns -- 16 lanes cover 256 contiguous bytes of a row
    float4 wr[8];
    const float* wbase = W + (unsigned)((jq * 128 + gs * 8) * D + col);
#pragma unroll
    for (int r = 0; r < 8; ++r) wr[r] = *reinterpret_cast<const float4*>(wbase + r * D);

    constexpr int U = 16;
    float mx = -INFINITY;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float lt = 0.f;
    for (int gI = tid; gI < G; gI += 256) mx = fmaxf(mx, pm[(unsigned)gI * sm + p]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int g0 = gs; g0 < G; g0 += 16 * U) {
        float mg[U], lg[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // out-of-range partials: the last one again, with weight 0 (no branch between the loads)
            const unsigned gI = (unsigned)min(g0 + 16 * u, G - 1);
            mg[u] = pm[gI * sm + p];
            lg[u] = pl[gI * sl_ + p];
            v[u] = *reinterpret_cast<const float4*>(pacc + gI * sa + (unsigned)(p * D + col));
            if (g0 + 16 * u >= G) mg[u] = -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float f = (mg[u] == -INFINITY) ? 0.f : fast_exp2(mg[u] - mx);
            lt += lg[u] * f;
            a.x += v[u].x * f; a.y += v[u].y * f; a.z += v[u].z * f; a.w += v[u].w * f;
        }
    }
    sacc[gs][c4] = a;
    if (c4 == 0) sl[gs] = lt;
    __syncthreads();
    if (tid < 16) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        float ls = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 v = sacc[k][tid];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            ls += sl[k];
        }
        const float inv = 1.f / ls;
        s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
        if (jq == 0) {
            *reinterpret_cast<float4*>(out + (unsigned)(p * D + c0 + tid * 4)) = s;
            if (tid == 0 && blockIdx.x == 0) {
                m2[p] = mx;
                l[p] = ls;
            }
        }
        sacc[0][tid] = s;   // (row 0 was read by this very thread only: no hazard)
    }
    __syncthreads();
    const float4 o = sacc[0][c4];
    // 8 dot-product pieces per lane, summed over the 16 lanes of a row group by a halving butterfly (4 + 2 + 1 shuffles, then one
    // more): lane c4 ends with the complete sum of row gs * 8 + (c4 >> 1) (both lanes of a pair hold it; the even one stores)
    float d[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = wr[r].x * o.x + wr[r].y * o.y + wr[r].z * o.z + wr[r].w * o.w;
#pragma unroll
    for (int h = 8, n = 8; h >= 2; h >>= 1, n >>= 1) {
        const bool up = (c4 & h) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float keep = up ? d[i + n / 2] : d[i];
            const float send = up ? d[i] : d[i + n / 2];
            d[i] = keep + __shfl_xor(send, h);
        }
    }
    const float tot = d[0] + __shfl_xor(d[0], 1);   // row index: bit3 -> 4, bit2 -> 2, bit1 -> 1
    if ((c4 & 1) == 0) vpart[(unsigned)((blockIdx.x * P + p) * D + jq * 128 + gs * 8 + (c4 >> 1))] = tot;
}

__global__ __launch_bounds__(512) void k_head_finish_parts(const float* __restrict__ vpart, const float* __restrict__ rows, int P,
                                                            int ncc, int pool_mode, const float* __restrict__ pool_w,
                                                            const float* __restrict__ bias, const float* __restrict__ That, int K,
                                                            const float* __restrict__ logit_scale, float* __restrict__ pooled,
                                                            float* __restrict__ v, float* __restrict__ vhat,
                                                            float* __restrict__ vnorm, float* __restrict__ logits,
                                                            float* __restrict__ incidence) {
    constexpr int D = 512;
    __shared__ float sp[D];
    __shared__ float swg[VLSA_MAX_P];
    __shared__ float slog[VLSA_MAX_K];
    __shared__ float red[8];
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // everything this thread needs is in flight at once: its column of the <= 16 x 8 partial vectors and of the P rows
    // (rows p >= P: row P - 1 again, weighted 0 below -- no branch between the loads, so all of them are in flight together)
    float part[VLSA_MAX_P], x[VLSA_MAX_P];
#pragma unroll
    for (int p = 0; p < VLSA_MAX_P; ++p) {
        const unsigned pc = (unsigned)min(p, P - 1);
        float q[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) q[cc] = vpart[((unsigned)cc * P + pc) * D + tid];
        part[p] = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        x[p] = rows[pc * D + tid];
    }
    const float bj = bias != nullptr ? bias[tid] : 0.f;
    if (tid == 0) {
        if (pool_mode == VLSA_POOL_WEIGHT) {   // softmax over the raw 'weight' parameter (model/deepmil.py:148)
            float mx = -INFINITY, s = 0.f;
            for (int p = 0; p < P; ++p) mx = fmaxf(mx, pool_w[p]);
            for (int p = 0; p < P; ++p) { swg[p] = expf(pool_w[p] - mx); s += swg[p]; }
            for (int p = 0; p < P; ++p) swg[p] /= s;
        } else {
            for (int p = 0; p < P; ++p) swg[p] = 1.f;
        }
        for (int p = P; p < VLSA_MAX_P; ++p) swg[p] = 0.f;
    }
    __syncthreads();
    float vs = 0.f, ps = 0.f;
#pragma unroll
    for (int p = 0; p < VLSA_MAX_P; ++p) {
        vs += swg[p] * part[p];
        ps += swg[p] * x[p];
    }
    if (pool_mode == VLSA_POOL_MEAN) {   // left-to-right sum, then / P (pooled_col's order)
        vs /= (float)P;
        ps /= (float)P;
    }
    const float vj = vs + bj;
    pooled[tid] = ps;
    v[tid] = vj;
    float ss = wave_sum(vj * vj);
    if (lane == 0) red[wv] = ss;
    __syncthreads();
    ss = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    const float nrm = fmaxf(sqrtf(ss), kNormEps);
    const float u = vj / nrm;
    sp[tid] = u;
    vhat[tid] = u;
    if (tid == 0) vnorm[0] = nrm;
    __syncthreads();
    const float ls = expf(logit_scale[0]);
    for (int k = wv; k < K; k += 8) {
        const float* tk = That + (size_t)k * D;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < D / 64; ++c) s += sp[lane + 64 * c] * tk[lane + 64 * c];
        s = wave_sum(s);
        if (lane == 0) {
            const float lg = ls * s;
            logits[k] = lg;
            slog[k] = lg;
        }
    }
    if (incidence == nullptr) return;
    __syncthreads();
    if (tid == 0) {
        float mx = -INFINITY, s = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, slog[k]);
        for (int k = 0; k < K; ++k) s += expf(slog[k] - mx);
        for (int k = 0; k < K; ++k) incidence[k] = expf(slog[k] - mx) / s;
    }
}

// ---------------------------------------------------------------------------------------------------
// Hardware-layout probes (used by tests/test_gpu_probe.py)
//   which = 0: MFMA 16x16x32 bf16 fragment map.  A[i][k] = i + 1 (k == probe_k) ... see test.
//   which = 1: ds_read_b64_tr_b16 lane map.
// ---------------------------------------------------------------------------------------------------
__global__ void k_probe_mfma(float* out) {
    // A[i][k] = (i + 1) if k == 5 * (i % 6) else 0 ; B[k][j] = 100 * k + j  -> C[i][j] = (i+1) * (100*k_i + j)
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e;
        a[e] = (__bf16)((k == 5 * (i % 6)) ? (float)(i + 1) : 0.f);
        b[e] = (__bf16)(float)(4 * k + i);  // B[k][j = i]; values < 256 exactly representable in bf16
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

__global__ void k_probe_tr(float* out) {
    // LDS holds a [4 x 16 rows][16 cols] bf16 matrix, row stride 64 B (2 x the payload, to prove the row
    // stride is free), value(row, col) = row * 16 + col (< 1024, but keep < 256: rows 0..15 only used).
    __shared__ __attribute__((aligned(16))) unsigned short lds[16 * 32];
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    for (int idx = lane; idx < 16 * 32; idx += 64) {
        const int row = idx >> 5, col = idx & 31;
        const __bf16 val = (__bf16)(float)((col < 16) ? (row * 16 + col) : 0);
        lds[idx] = __builtin_bit_cast(unsigned short, val);
    }
    __syncthreads();
    // group g reads the 4 x 16 block of rows 4g..4g+3: lane supplies row 4g + (i >> 2), cols 4*(i&3)..+3
    const unsigned short* src = lds + (4 * g + (i >> 2)) * 32 + (i & 3) * 4;
    const bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(src));
#pragma unroll
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = (float)t[r];
}

}  // namespace vlsa

using namespace vlsa;

extern "C" int vlsa_abi_version(void) { return VLSA_ABI_VERSION; }

extern "C" const char* vlsa_error_string(int code) {
    switch (code) {
        case VLSA_OK: return "ok";
        case VLSA_EINVAL: return "invalid argument";
        case VLSA_EUNSUPPORTED: return "unsupported configuration";
        case VLSA_ELAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}

static inline int launch_status() { return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH; }

extern "C" int vlsa_vlfan_merge_strided(const float* pm, int64_t pm_stride, const float* pl, int64_t pl_stride,
                                        const float* pacc, int64_t pacc_stride, int G, int P, int D, int normalise,
                                        float* m2, float* l, float* out, void* stream) {
    if (!pm || !pl || !pacc || !m2 || !l || !out) return VLSA_EINVAL;
    if (G < 1 || P < 1 || P > VLSA_MAX_P || D <= 0 || D > VLSA_MAX_D || (D % 8) != 0) return VLSA_EINVAL;
    if (pm_stride < P || pl_stride < P || pacc_stride < (int64_t)P * D || (pacc_stride % 4) != 0 ||
        (reinterpret_cast<uintptr_t>(pacc) & 15) != 0)
        return VLSA_EINVAL;
    hipLaunchKernelGGL(k_vlfan_merge, dim3((D + 63) / 64, P), dim3(256), 0, (hipStream_t)stream, pm, pl, pacc, G, P, D,
                       normalise, m2, l, out, pm_stride, pl_stride, pacc_stride);
    return launch_status();
}

extern "C" int vlsa_vlfan_merge(const float* pm, const float* pl, const float* pacc, int G, int P, int D,
                                int normalise, float* m2, float* l, float* out, void* stream) {
    return vlsa_vlfan_merge_strided(pm, kPStride, pl, kPStride, pacc, (int64_t)P * D, G, P, D, normalise, m2, l, out, stream);
}

extern "C" int vlsa_attn_normalise(const float* scores, int P, int64_t N, const float* m2, const float* l, float* A,
                                   void* stream) {
    if (!scores || !m2 || !l || !A || P < 1 || P > VLSA_MAX_P || N < 0) return VLSA_EINVAL;
    if (N == 0) return VLSA_OK;
    int64_t nb = (N + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_attn_normalise, dim3((unsigned)nb, P), dim3(256), 0, (hipStream_t)stream, scores, P, N, m2, l, A);
    return launch_status();
}

extern "C" int vlsa_normalize_rows(const float* in, int rows, int D, float* out, float* norms, void* stream) {
    if (!in || !out || rows < 0 || D <= 0) return VLSA_EINVAL;
    if (rows == 0) return VLSA_OK;
    hipLaunchKernelGGL(k_normalize_rows, dim3(rows), dim3(256), 0, (hipStream_t)stream, in, D, out, norms);
    return launch_status();
}

// 256 bytes of ticket counter (k_head) + the [8][VLSA_MAX_P][512] partial adapter vectors of the single-slide tail
constexpr size_t kHeadTicketBytes = 256;
extern "C" size_t vlsa_head_workspace_bytes(int D) { (void)D; return kHeadTicketBytes + (size_t)8 * VLSA_MAX_P * 512 * sizeof(float); }

// Partial merge + query pooling + adapter + cosine logits of ONE bag: the tail of vlsa_vlfan_forward_bag as an entry point of its
// own.  D == 512 with mean / weight pooling and a Linear adapter: k_vlfan_merge_wpart + k_head_finish_parts (two ticket-free
// launches); anything else: vlsa_vlfan_merge + vlsa_head_forward.
extern "C" int vlsa_vlfan_merge_head(const float* pm, const float* pl, const float* pacc, int G, int P, int D, int pool_mode,
                                     const float* pool_w, const float* W, const float* b, const float* That, int K,
                                     const float* logit_scale, void* head_ws, float* m2, float* l, float* out, float* pooled,
                                     float* v, float* vhat, float* vnorm, float* logits, float* incidence, void* stream) {
    if (!pm || !pl || !pacc || !m2 || !l || !out || !That || !logit_scale || !head_ws || !pooled || !v || !vhat || !vnorm || !logits)
        return VLSA_EINVAL;
    if (G < 1 || P < 1 || P > VLSA_MAX_P || K < 1 || K > VLSA_MAX_K || D <= 0 || D > VLSA_MAX_D || (D % 8) != 0) return VLSA_EINVAL;
    if (pool_mode < VLSA_POOL_MEAN || pool_mode > VLSA_POOL_GIVEN) return VLSA_EINVAL;
    if (pool_mode == VLSA_POOL_WEIGHT && !pool_w) return VLSA_EINVAL;
    const bool fused = D == 512 && W != nullptr && (pool_mode == VLSA_POOL_MEAN || pool_mode == VLSA_POOL_WEIGHT) &&
                       (reinterpret_cast<uintptr_t>(pacc) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    if (!fused) {
        const int rc = vlsa_vlfan_merge(pm, pl, pacc, G, P, D, 1, m2, l, out, stream);
        if (rc != VLSA_OK) return rc;
        return vlsa_head_forward(out, P, D, pool_mode, pool_w, W, b, That, K, logit_scale, head_ws, pooled, v, vhat, vnorm, logits,
                                 incidence, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    float* vpart = reinterpret_cast<float*>(static_cast<unsigned char*>(head_ws) + kHeadTicketBytes);
    // (Round 6 measured both launches as ONE -- the merge workgroups publishing write-through and arriving on a two-level counter, the
    // last arriver running the finish on agent-scope loads: 16.6 us against 12.3 us for these two launches, back to back on the same
    // box; the hand-off costs more than the kernel boundary it removes, as it did for the ticketed k_head of rounds 1-4.  Not kept:
    // docs/LAB_NOTEBOOK.md.)
    hipLaunchKernelGGL(k_vlfan_merge_wpart, dim3(8, P, 4), dim3(256), 0, s, pm, pl, pacc, G, P, m2, l, out, W, vpart);
    if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    hipLaunchKernelGGL(k_head_finish_parts, dim3(1), dim3(512), 0, s, vpart, out, P, 8, pool_mode, pool_w, b, That, K, logit_scale,
                       pooled, v, vhat, vnorm, logits, incidence);
    return launch_status();
}

extern "C" int vlsa_head_forward(const float* rows, int P, int D, int pool_mode, const float* pool_w, const float* W,
                                 const float* b, const float* That, int K, const float* logit_scale, void* workspace,
                                 float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence,
                                 void* stream) {
    if (!rows || !That || !logit_scale || !workspace || !pooled || !v || !vhat || !vnorm || !logits) return VLSA_EINVAL;
    if (P < 1 || P > VLSA_MAX_P || K < 1 || K > VLSA_MAX_K || D <= 0 || D > VLSA_MAX_D || (D % 4) != 0) return VLSA_EINVAL;
    if (pool_mode < VLSA_POOL_MEAN || pool_mode > VLSA_POOL_GIVEN) return VLSA_EINVAL;
    if (pool_mode == VLSA_POOL_WEIGHT && !pool_w) return VLSA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int NB = W ? (D + kHeadRowsPerBlock - 1) / kHeadRowsPerBlock : 1;
    hipLaunchKernelGGL(k_head, dim3(NB), dim3(256), 0, s, rows, P, D, pool_mode, pool_w, W, b, That, K, logit_scale,
                       static_cast<unsigned int*>(workspace), pooled, v, vhat, vnorm, logits, incidence, NB);
    return launch_status();
}

int vlsa_launch_head_batch(const float* rows, int B, int P, int D, int pool_mode, const float* pool_w, const float* W,
                           const float* b, const float* That, int K, const float* logit_scale, unsigned int* counters,
                           float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence,
                           hipStream_t s) {
    if (pool_mode < VLSA_POOL_MEAN || pool_mode > VLSA_POOL_GIVEN) return VLSA_EINVAL;
    if (pool_mode == VLSA_POOL_WEIGHT && !pool_w) return VLSA_EINVAL;
    const int NB = W ? (D + kHeadRowsPerBlock - 1) / kHeadRowsPerBlock : 1;
    // one workgroup per (8 rows of W, bag): all bags' latency chains run side by side (a variant that kept W in registers
    // and walked 8 bags per workgroup measured 2x slower: the per-bag chain, not the W traffic, is the cost)
    hipLaunchKernelGGL(k_head, dim3(NB, B), dim3(256), 0, s, rows, P, D, pool_mode, pool_w, W, b, That, K, logit_scale,
                       counters, pooled, v, vhat, vnorm, logits, incidence, NB);
    return launch_status();
}

// Batched head from already pooled vectors (pooled: [B, D]); v = pooled when W == nullptr (Identity head).
int vlsa_launch_head_pooled_batch(const float* pooled, int B, int D, const float* W, const float* b, const float* That, int K,
                                  const float* logit_scale, float* v, float* vhat, float* vnorm, float* logits,
                                  float* incidence, hipStream_t s) {
    const float* vin = pooled;
    if (W != nullptr) {
        if (D == 512 && B >= 16 && (reinterpret_cast<uintptr_t>(pooled) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0)
            hipLaunchKernelGGL(k_head_linear_mfma, dim3((B + 15) / 16, 16), dim3(512), 0, s, pooled, B, W, b, v);   // f32 matrix pipe
        else
            hipLaunchKernelGGL(k_head_linear, dim3((D + kHeadRowsPerBlock - 1) / kHeadRowsPerBlock, (B + kHeadBagsPerBlock - 1) / kHeadBagsPerBlock),
                               dim3(256), 0, s, pooled, B, D, W, b, v);
        vin = v;
    } else {
        if (hipMemcpyAsync(v, pooled, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return VLSA_ELAUNCH;
    }
    hipLaunchKernelGGL(k_head_finish, dim3(B), dim3(256), 0, s, vin, D, That, K, logit_scale, vhat, vnorm, logits, incidence);
    return launch_status();
}

// Batched head from the aggregated rows [B, P, D] without tickets (the training forward, round 4): one pooling launch (a workgroup per
// bag, `pooled_col` as in k_head), then the two launches of the pooled route.  The ticketed k_head over (D / 8) x B workgroups took
// 33.8 us for 32 bags (every workgroup re-pools its bag's rows and runs the drain + ticket chain; profiles/r04_step_kernel_stats.csv).
namespace vlsa {
__global__ __launch_bounds__(256) void k_pool_rows(const float* __restrict__ rows, int P, int D, int pool_mode,
                                                    const float* __restrict__ pool_w, float* __restrict__ pooled, int B,
                                                    const float* __restrict__ T, float* __restrict__ That, float* __restrict__ tnorm) {
    __shared__ float spw[VLSA_MAX_P];
    __shared__ float red[4];
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int tid = threadIdx.x, bag = blockIdx.x;
    if (bag >= B) {      // workgroups behind the bags (vlsa_head_forward_batch_text): T^[r] = T[r] / max(|T[r]|, eps), k_normalize_rows' arithmetic
        const int r = bag - B;
        const float* x = T + (size_t)r * D;
        float ss = 0.f;
        for (int d = tid; d < D; d += 256) ss += x[d] * x[d];
        ss = block_sum_256(ss, red);
        const float nrm = fmaxf(sqrtf(ss), kNormEps);
        for (int d = tid; d < D; d += 256) That[(size_t)r * D + d] = x[d] / nrm;
        if (tnorm != nullptr && tid == 0) tnorm[r] = nrm;
        return;
    }
    if (pool_mode == VLSA_POOL_WEIGHT) {
        if (tid == 0) {
            float mx = -INFINITY, s = 0.f;
            for (int p = 0; p < P; ++p) mx = fmaxf(mx, pool_w[p]);
            for (int p = 0; p < P; ++p) { spw[p] = expf(pool_w[p] - mx); s += spw[p]; }
            for (int p = 0; p < P; ++p) spw[p] /= s;
        }
        __syncthreads();
    }
    for (int c = tid; c < D; c += 256) pooled[(size_t)bag * D + c] = pooled_col(rows + (size_t)bag * P * D, P, D, c, pool_mode, spw);
}
}  // namespace vlsa
int vlsa_launch_head_rows_batch(const float* rows, int B, int P, int D, int pool_mode, const float* pool_w, const float* W,
                                const float* b, const float* That, int K, const float* logit_scale, float* pooled, float* v,
                                float* vhat, float* vnorm, float* logits, float* incidence, hipStream_t s) {
    if (pool_mode < VLSA_POOL_MEAN || pool_mode > VLSA_POOL_WEIGHT) return VLSA_EINVAL;
    if (pool_mode == VLSA_POOL_WEIGHT && !pool_w) return VLSA_EINVAL;
    hipLaunchKernelGGL(vlsa::k_pool_rows, dim3(B), dim3(256), 0, s, rows, P, D, pool_mode, pool_w, pooled, B, (const float*)nullptr,
                       (float*)nullptr, (float*)nullptr);
    if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    return vlsa_launch_head_pooled_batch(pooled, B, D, W, b, That, K, logit_scale, v, vhat, vnorm, logits, incidence, s);
}

// The training step's head from the RAW text features: vlsa_normalize_rows(T) + vlsa_head_forward_batch in three launches instead of
// four -- the K rows of T are normalised by K extra workgroups of the pooling launch (they are independent of the bags' rows, and inside
// the graph-replayed step a launch of its own costs them ~4.8 us).  Writes That [K, D] and tnorm [K] as vlsa_normalize_rows does.
extern "C" int vlsa_head_forward_batch_text(const float* rows, int B, int P, int D, int pool_mode, const float* pool_w, const float* W,
                                            const float* b, const float* T, int K, const float* logit_scale, float* That, float* tnorm,
                                            float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence,
                                            void* stream) {
    if (!rows || !T || !That || !tnorm || !logit_scale || !pooled || !v || !vhat || !vnorm || !logits) return VLSA_EINVAL;
    if (B < 1 || P < 1 || P > VLSA_MAX_P || K < 1 || K > VLSA_MAX_K || D <= 0 || D > VLSA_MAX_D || (D % 4) != 0) return VLSA_EINVAL;
    if (pool_mode < VLSA_POOL_MEAN || pool_mode > VLSA_POOL_WEIGHT || (pool_mode == VLSA_POOL_WEIGHT && !pool_w) || pooled == rows) return VLSA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vlsa::k_pool_rows, dim3(B + K), dim3(256), 0, s, rows, P, D, pool_mode, pool_w, pooled, B, T, That, tnorm);
    if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    return vlsa_launch_head_pooled_batch(pooled, B, D, W, b, That, K, logit_scale, v, vhat, vnorm, logits, incidence, s);
}

extern "C" int vlsa_debug_probe(int which, void* out, size_t out_bytes, void* stream) {
    if (!out || out_bytes < 64 * 4 * sizeof(float)) return VLSA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (which == 0) hipLaunchKernelGGL(k_probe_mfma, dim3(1), dim3(64), 0, s, static_cast<float*>(out));
    else if (which == 1) hipLaunchKernelGGL(k_probe_tr, dim3(1), dim3(64), 0, s, static_cast<float*>(out));
    else return VLSA_EINVAL;
    return launch_status();
}

// The whole per-bag inference forward as ONE host call (model/vlsa.py:181-198 with cached text features): query + text
// preparation, the streaming aggregation, the partial merge, optionally the attention weights, pooling + head.  Same five
// launches as calling the entry points one by one -- what it removes is four Python -> C crossings per bag: the reference's
// handler calls the model one bag at a time (runner/vlsa_handler.py:322-330) and that path is host-bound.
extern "C" int vlsa_vlfan_forward_bag(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* Q, int nq, int gated,
                                      float coattn_scale, const float* T, int K, const float* logit_scale, int pool_mode,
                                      const float* pool_w, const float* W, const float* b, int kernel, void* qprep, float* That,
                                      float* tnorm, float* pm, float* pl, float* pacc, int G, float* m2, float* l, float* out,
                                      float* scores, float* A, void* head_ws, float* pooled, float* v, float* vhat, float* vnorm,
                                      float* logits, float* incidence, void* stream) {
    const int P = gated ? nq - 1 : nq;
    int rc = VLSA_OK;
    if (Q) {   // Q == NULL: qprep / That still hold the result of an earlier call with the same parameters (eval loops)
        rc = vlsa_prepare_queries_and_text(Q, nq, D, gated, coattn_scale, qprep, T, K, That, tnorm, stream);
        if (rc != VLSA_OK) return rc;
    }
    rc = vlsa_vlfan_partial(X, x_dtype, N, ldx, D, qprep, P, kernel, pm, pl, pacc, scores, stream);
    if (rc != VLSA_OK) return rc;
    if (pool_mode < 0) {   // aggregation only: the caller pools the P rows itself (attention poolings) and calls the head
        rc = vlsa_vlfan_merge(pm, pl, pacc, G, P, D, 1, m2, l, out, stream);
        if (rc != VLSA_OK) return rc;
    } else {
        rc = vlsa_vlfan_merge_head(pm, pl, pacc, G, P, D, pool_mode, pool_w, W, b, That, K, logit_scale, head_ws, m2, l, out, pooled, v,
                                   vhat, vnorm, logits, incidence, stream);
        if (rc != VLSA_OK) return rc;
    }
    if (scores && A) return vlsa_attn_normalise(scores, P, N, m2, l, A, stream);
    return VLSA_OK;
}

// ---------------------------------------------------------------------------------------------------
// Backward of the batched head (mean query pooling + Linear / identity adapter + cosine logits) for the training step:
//   pooled = mean_p rows;  v = W pooled + b;  v^ = v / |v|;  T^ = T / |T|;  logits = exp(ls) v^ T^^T     (model/deepmil.py:203-204,
//   model/vlsa.py:188-192).  Two launches instead of ~25 autograd kernels -- the optimizer step is bound by its number of
//   dependent launches.  k_head_bwd_dv (one workgroup per bag): d v^ = exp(ls) dlogits T^ (+ g_vhat), d v = (d v^ - v^ (v^ . d v^)) / |v|,
//   and the bag's share of d ls = sum_k dlogits logits.  k_head_bwd_params: block j < D: dW[j, :] = sum_b dv[b, j] pooled[b, :],
//   db[j] = sum_b dv[b, j]; the next (D / 32) x (B / 8) blocks: d pooled[8 bags, 32 columns] = dv W (identity head: dv), d rows[b, p, :] = d pooled[b] / P;
//   the next K blocks: d T^[k] = exp(ls) sum_b dlogits[b, k] v^[b] (+ g_That), d T = (d T^ - T^ (T^ . d T^)) / |T|; the last: d ls.
namespace vlsa {
__global__ __launch_bounds__(256) void k_head_bwd_dv(const float* __restrict__ dlogits, const float* __restrict__ g_vhat,
                                                      const float* __restrict__ vhat, const float* __restrict__ vnorm,
                                                      const float* __restrict__ That, const float* __restrict__ logits,
                                                      const float* __restrict__ logit_scale, int D, int K, float* __restrict__ dv,
                                                      float* __restrict__ dls_part) {
    __shared__ float red[4];
    __shared__ float sd[VLSA_MAX_K];
    const int bag = blockIdx.x, tid = threadIdx.x;
    const float ls = expf(logit_scale[0]);
    if (tid < K) sd[tid] = dlogits[(size_t)bag * K + tid];
    __syncthreads();
    float dvh[VLSA_MAX_D / 256];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
        const int c = tid + 256 * i;
        float s = 0.f;
        if (c < D) {
            for (int k = 0; k < K; ++k) s = fmaf(sd[k], That[(size_t)k * D + c], s);
            s *= ls;
            if (g_vhat != nullptr) s += g_vhat[(size_t)bag * D + c];
            dot = fmaf(s, vhat[(size_t)bag * D + c], dot);
        }
        dvh[i] = s;
    }
    dot = block_sum_256(dot, red);
    const float inv = 1.f / vnorm[bag];
#pragma unroll
    for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
        const int c = tid + 256 * i;
        if (c < D) dv[(size_t)bag * D + c] = (dvh[i] - vhat[(size_t)bag * D + c] * dot) * inv;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(sd[k], logits[(size_t)bag * K + k], s);
        dls_part[bag] = s;
    }
}

__global__ __launch_bounds__(256) void k_head_bwd_params(const float* __restrict__ dv, const float* __restrict__ pooled,
                                                          const float* __restrict__ W, const float* __restrict__ dlogits,
                                                          const float* __restrict__ g_That, const float* __restrict__ vhat,
                                                          const float* __restrict__ That, const float* __restrict__ tnorm,
                                                          const float* __restrict__ logit_scale, const float* __restrict__ dls_part,
                                                          int B, int P, int D, int K, float* __restrict__ dW, float* __restrict__ db,
                                                          float* __restrict__ drows, float* __restrict__ dT, float* __restrict__ dls) {
    __shared__ float red[4];
    __shared__ float sb[64];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    const int nW = W != nullptr ? D : 0;
    if (blk < nW) {                                   // dW row j, db[j]
        const int j = blk;
        for (int b0 = 0; b0 < B; b0 += 64) {           // dv[:, j] of up to 64 bags at a time
            __syncthreads();
            if (tid < 64) sb[tid] = b0 + tid < B ? dv[(size_t)(b0 + tid) * D + j] : 0.f;
            __syncthreads();
            // 16 loads of `pooled` in flight per round (a load per fma, one after the other, was a chain of 32 L2 round trips: 18 us for
            // 32 bags -- the longest of the small launches of the training step); same order of additions as before
            const int nb = min(64, B - b0);
            for (int c = tid; c < D; c += 256) {
                float s = b0 == 0 ? 0.f : dW[(size_t)j * D + c];
                for (int b1 = 0; b1 < nb; b1 += 16) {
                    float pv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) pv[u] = b1 + u < nb ? pooled[(size_t)(b0 + b1 + u) * D + c] : 0.f;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (b1 + u < nb) s = fmaf(sb[b1 + u], pv[u], s);
                }
                dW[(size_t)j * D + c] = s;
            }
            if (tid == 0) {
                float s = b0 == 0 ? 0.f : db[j];
                for (int b = 0; b < 64 && b0 + b < B; ++b) s += sb[b];
                db[j] = s;
            }
        }
        return;
    }
    blk -= nW;
    // d pooled -> d rows: a workgroup per (32 columns, 8 bags).  d pooled[b, c] = sum_j dv[b, j] W[j, c]: thread (c, js) walks an eighth
    // of the j range, 16 W loads in flight per round (dv staged in LDS, W row pieces coalesced), the eight slices are summed through
    // LDS in a fixed order.  (Until round 4 eight workgroups -- one per 64 columns -- walked ALL bags and the whole of W four times: 67 us
    // of the 32-bag optimizer step, profiles/r04_step_kernel_stats.csv; now (D / 32) x (B / 8) workgroups of 4 rounds each.)
    const int nCB = (D + 31) / 32, nC = nCB * ((B + 7) / 8);
    if (blk < nC) {
        __shared__ float sdv[8][VLSA_MAX_D];
        __shared__ float sred[8][8][32];
        const int cb = blk % nCB, b0 = (blk / nCB) * 8;
        const int cl = tid & 31, js = tid >> 5, c = cb * 32 + cl;
        const float invP = 1.f / (float)P;
        const int jq = (D + 7) / 8;
        for (int e = tid; e < 8 * D; e += 256) {
            const int b = e / D, j = e % D;
            sdv[b][j] = b0 + b < B ? dv[(size_t)(b0 + b) * D + j] : 0.f;
        }
        __syncthreads();
        float acc[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[b] = 0.f;
        if (W != nullptr) {
            const int jbeg = js * jq, jend = min(D, jbeg + jq);
            for (int j0 = jbeg; j0 < jend; j0 += 32) {       // (32 loads in flight per round: two rounds at D = 512)
                float wv[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) wv[u] = (j0 + u < jend && c < D) ? W[(size_t)(j0 + u) * D + c] : 0.f;
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (j0 + u < jend) {
#pragma unroll
                        for (int b = 0; b < 8; ++b) acc[b] = fmaf(sdv[b][j0 + u], wv[u], acc[b]);
                    }
            }
        } else if (js == 0 && c < D) {
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[b] = sdv[b][c];
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) sred[js][b][cl] = acc[b];
        __syncthreads();
        {
            const int b = tid >> 5, col = cb * 32 + cl;       // 256 threads = 8 bags x 32 columns
            if (b0 + b < B && col < D) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) v += sred[q][b][cl];
                v *= invP;
                for (int p = 0; p < P; ++p) drows[((size_t)(b0 + b) * P + p) * D + col] = v;
            }
        }
        return;
    }
    blk -= nC;
    if (blk < K) {                                    // d T[k]
        const int k = blk;
        const float ls = expf(logit_scale[0]);
        float dth[VLSA_MAX_D / 256];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
            const int c = tid + 256 * i;
            float s = 0.f;
            if (c < D) {
                for (int b1 = 0; b1 < B; b1 += 16) {           // (16 loads in flight per round, as above)
                    float dl[16], vh[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        dl[u] = b1 + u < B ? dlogits[(size_t)(b1 + u) * K + k] : 0.f;
                        vh[u] = b1 + u < B ? vhat[(size_t)(b1 + u) * D + c] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (b1 + u < B) s = fmaf(dl[u], vh[u], s);
                }
                s *= ls;
                if (g_That != nullptr) s += g_That[(size_t)k * D + c];
                dot = fmaf(s, That[(size_t)k * D + c], dot);
            }
            dth[i] = s;
        }
        dot = block_sum_256(dot, red);
        const float inv = 1.f / fmaxf(tnorm[k], kNormEps);
#pragma unroll
        for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
            const int c = tid + 256 * i;
            if (c < D) dT[(size_t)k * D + c] = (dth[i] - That[(size_t)k * D + c] * dot) * inv;
        }
        return;
    }
    if (tid == 0) {                                   // d logit_scale
        float s = 0.f;
        for (int b1 = 0; b1 < B; b1 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = b1 + u < B ? dls_part[b1 + u] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (b1 + u < B) s += v[u];
        }
        dls[0] = s;
    }
}
}  // namespace vlsa

/* workspace: (B * D + B) floats.  g_vhat [B, D] / g_That [K, D]: gradients flowing into the returned unit features, or NULL. */
extern "C" int vlsa_head_backward_batch(const float* dlogits, const float* g_vhat, const float* g_That, const float* pooled,
                                        const float* vhat, const float* vnorm, const float* That, const float* tnorm,
                                        const float* logits, const float* W, const float* logit_scale, int B, int P, int D, int K,
                                        float* workspace, float* drows, float* dW, float* db, float* dT, float* dls, void* stream) {
    if (!dlogits || !pooled || !vhat || !vnorm || !That || !tnorm || !logits || !logit_scale || !workspace || !drows || !dT || !dls)
        return VLSA_EINVAL;
    if (W && (!dW || !db)) return VLSA_EINVAL;
    if (B < 1 || P < 1 || P > VLSA_MAX_P || K < 1 || K > VLSA_MAX_K || D < 1 || D > VLSA_MAX_D) return VLSA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* dv = workspace;
    float* dls_part = workspace + (size_t)B * D;
    hipLaunchKernelGGL(vlsa::k_head_bwd_dv, dim3(B), dim3(256), 0, s, dlogits, g_vhat, vhat, vnorm, That, logits, logit_scale, D, K, dv, dls_part);
    hipLaunchKernelGGL(vlsa::k_head_bwd_params, dim3((W ? D : 0) + ((D + 31) / 32) * ((B + 7) / 8) + K + 1), dim3(256), 0, s, dv, pooled, W, dlogits, g_That, vhat, That,
                       tnorm, logit_scale, dls_part, B, P, D, K, dW, db, drows, dT, dls);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Per-bag TRAINING step as the reference's handler issues it (runner/vlsa_handler.py:267-289: one net(X) per bag, the predictions
// concatenated, ONE backward): the forward is vlsa_vlfan_forward_bag with per-call result buffers, the backward of one bag is
// the head backward (B = 1), the streaming backward pass of the aggregation, its reduction and the chain rule back to the raw
// queries -- six launches behind ONE host call (that loop is bound by the host side of its ~64 autograd nodes per step).
namespace vlsa {
// dQ from d e:  e_p = q^_p - gated * q^_gate,  q^ = q / max(|q|, 1e-12)   (model/deepmil.py:187-193).  block = one query row.
__global__ __launch_bounds__(256) void k_query_chain(const float* __restrict__ dE, const float* __restrict__ qhat,
                                                      const float* __restrict__ qnorm, int P, int D, float* __restrict__ dQ) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float d[VLSA_MAX_D / 256];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
        const int c = tid + 256 * i;
        float v = 0.f;
        if (c < D) {
            if (row < P) {
                v = dE[(size_t)row * D + c];
            } else {   // the gate row: every effective query carries - q^_gate
                for (int p = 0; p < P; ++p) v -= dE[(size_t)p * D + c];
            }
            dot += v * qhat[(size_t)row * D + c];
        }
        d[i] = v;
    }
    dot = block_sum_256(dot, red);
    const float inv = 1.f / qnorm[row];
#pragma unroll
    for (int i = 0; i < VLSA_MAX_D / 256; ++i) {
        const int c = tid + 256 * i;
        if (c < D) dQ[(size_t)row * D + c] = (d[i] - qhat[(size_t)row * D + c] * dot) * inv;
    }
}
}  // namespace vlsa

// dQ [nq, D] from dE [P, D] (the gradient w.r.t. the effective unit queries e_p = q^_p - gated q^_gate) and a prepared query block:
// the chain rule through the normalisation (and the gate row) in one launch -- the batched aggregation's backward ran it as five torch
// kernels ([P, 512] mul / sum / mul / sub / div: 25 us of the graph-replayed optimizer step).
extern "C" int vlsa_query_chain(const float* dE, const void* qprep, int nq, int gated, int D, float* dQ, void* stream) {
    if (!dE || !qprep || !dQ || nq < 1 || nq > VLSA_MAX_P + 1 || D < 1 || D > VLSA_MAX_D) return VLSA_EINVAL;
    const int P = gated ? nq - 1 : nq;
    if (P < 1) return VLSA_EINVAL;
    const vlsa::QPrepLayout L(D);
    const unsigned char* qp = static_cast<const unsigned char*>(qprep);
    hipLaunchKernelGGL(vlsa::k_query_chain, dim3(nq), dim3(256), 0, (hipStream_t)stream, dE, reinterpret_cast<const float*>(qp + L.qhat),
                       reinterpret_cast<const float*>(qp + L.qnorm), P, D, dQ);
    return launch_status();
}

extern "C" int vlsa_vlfan_backward_bag(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* qprep, int nq, int gated,
                                       float coattn_scale, const float* dlogits, const float* g_vhat, const float* g_That,
                                       const float* pooled, const float* vhat, const float* vnorm, const float* That,
                                       const float* tnorm, const float* logits, const float* W, const float* logit_scale,
                                       const float* out, const float* m2, const float* l, int K, float* head_ws, float* drows, float* dW,
                                       float* db, float* dT, float* dls, void* bwd_prep, float* pm, float* pl, float* pacc, int G,
                                       float* dE, float* dQ, void* stream) {
    if (!qprep || !dE || !dQ || nq < 1) return VLSA_EINVAL;
    const int P = gated ? nq - 1 : nq;
    if (G != vlsa_num_partials(N)) return VLSA_EINVAL;
    int rc = vlsa_head_backward_batch(dlogits, g_vhat, g_That, pooled, vhat, vnorm, That, tnorm, logits, W, logit_scale, 1, P, D, K,
                                      head_ws, drows, dW, db, dT, dls, stream);
    if (rc != VLSA_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (N > 0) {
        rc = vlsa_vlfan_backward(X, x_dtype, N, ldx, D, qprep, P, coattn_scale, drows, out, m2, l, bwd_prep, pm, pl, pacc, stream);
        if (rc != VLSA_OK) return rc;
        // the unnormalised merge also hands back (max, sum) of the partials' bookkeeping columns: parked behind the partials
        rc = vlsa_vlfan_merge(pm, pl, pacc, G, P, D, 0, pm + (size_t)G * vlsa::kPStride, pl + (size_t)G * vlsa::kPStride, dE, stream);
        if (rc != VLSA_OK) return rc;
    } else {
        if (hipMemsetAsync(dE, 0, (size_t)P * D * sizeof(float), s) != hipSuccess) return VLSA_ELAUNCH;
    }
    const vlsa::QPrepLayout L(D);
    const unsigned char* qp = static_cast<const unsigned char*>(qprep);
    hipLaunchKernelGGL(vlsa::k_query_chain, dim3(nq), dim3(256), 0, s, dE, reinterpret_cast<const float*>(qp + L.qhat),
                       reinterpret_cast<const float*>(qp + L.qnorm), P, D, dQ);
    return launch_status();
}

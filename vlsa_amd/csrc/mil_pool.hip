// N-sized pieces of the other MIL encoders the VLSA wrapper accepts:
//   * scored pooling      out = softmax_N(a) @ X     (Attention_Pooling / Gated_Attention_Pooling over all patches,
//                         model/layers.py:116,147; with a == const it is FeatMIL / DeepMIL mean pooling)
//   * column max          FeatMIL / DeepMIL max pooling (model/deepmil.py:59-60,273-274)
//   * attention scores    a_n = w2 . tanh(h_n + b1) [* sigmoid(hg_n + bg)] + b2 from the hidden projections
//                         (the [N,512]x[512,256] projections themselves are plain GEMMs: rocBLAS via torch)
//   * row dots            x_n . v for the pooling backward
//   * per-class top-k mean of the zero-shot logits (logit_pooling, model/deepmil.py:16-37)
// All of them are HBM-bound streaming reductions: one wave per patch row, 16-byte (bf16) / 8-byte loads per
// lane are not needed here because a row is spread over the 64 lanes with unit stride (fully coalesced).
#include "vlsa_common.h"

namespace vlsa {

__device__ __forceinline__ void rows_of_block(int64_t N, int b, int G, int64_t& rbeg, int64_t& rend) {
    const int64_t q = N / G, r = N % G;
    rbeg = b * q + (b < r ? b : r);
    rend = rbeg + q + (b < r ? 1 : 0);
}

// ---- scored pooling: per-workgroup online-softmax partial with ONE query (P = 1 layout of the VLFAN partials)
template <typename XT, int DPL>
__global__ __launch_bounds__(256) void k_scored_pool_partial(const XT* __restrict__ X, int64_t N, int64_t ldx, int D,
                                                              const float* __restrict__ scores, float* __restrict__ pm,
                                                              float* __restrict__ pl, float* __restrict__ pacc, int G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x;
    int64_t rbeg, rend;
    rows_of_block(N, b, G, rbeg, rend);
    float acc[DPL], M = -INFINITY, l = 0.f;
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = 0.f;
    for (int64_t r = rbeg + w; r < rend; r += 4) {
        const float t = scores != nullptr ? scores[r] * kLog2e : 0.f;
        float x[DPL];
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = lane + 64 * i;
            x[i] = d < D ? load_as_float(X + r * ldx + d) : 0.f;
        }
        if (t > M) {
            const float f = fast_exp2(M - t);
            l *= f;
#pragma unroll
            for (int i = 0; i < DPL; ++i) acc[i] *= f;
            M = t;
        }
        const float wgt = fast_exp2(t - M);
        l += wgt;
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] += wgt * x[i];
    }
    float* sm = reinterpret_cast<float*>(smem);  // [4] M, [4] l, then [4][DPL*64] acc
    float* sl = sm + 4;
    float* sacc = sm + 8;
    constexpr int DW = DPL * 64;
    if (lane == 0) {
        sm[w] = M;
        sl[w] = l;
    }
#pragma unroll
    for (int i = 0; i < DPL; ++i) sacc[w * DW + lane + 64 * i] = acc[i];
    __syncthreads();
    const float mm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float f[4], lt = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = (sm[k] == -INFINITY) ? 0.f : fast_exp2(sm[k] - mm);
        lt += sl[k] * f[k];
    }
    if (tid == 0) {
        pm[(size_t)b * kPStride] = mm;
        pl[(size_t)b * kPStride] = lt;
    }
    for (int d = tid; d < D; d += 256)
        pacc[(size_t)b * D + d] = sacc[d] * f[0] + sacc[DW + d] * f[1] + sacc[2 * DW + d] * f[2] + sacc[3 * DW + d] * f[3];
}

// Fast path of the scored pooling (rows 16-byte aligned, D a multiple of the vector width): a lane owns NC chunks of 16
// bytes of a row (chunk c covers elements [(64 c + lane) VEC, +VEC)), so a wave reads a row with NC fully coalesced
// 16-byte loads; 4 rows per wave are in flight before the (sequential) online-softmax update.  Same outputs as above.
// Several bags per launch: grid (G, B); bags != null: workgroup (g, bag) pools rows_of_block(N_bag, g, G) of bag blockIdx.y with
// the scores at scores + a_off[bag] and writes partial bag * G + g.
struct PoolBag {
    const void* X;
    long long N, ldx;
};
template <typename XT, int NC>
__global__ __launch_bounds__(256) void k_scored_pool_partial_vec(const XT* __restrict__ X, int64_t N, int64_t ldx, int D,
                                                                  const float* __restrict__ scores, float* __restrict__ pm,
                                                                  float* __restrict__ pl, float* __restrict__ pacc, int G,
                                                                  const PoolBag* __restrict__ bags,
                                                                  const long long* __restrict__ a_off) {
    constexpr int VEC = 16 / (int)sizeof(XT);
    constexpr int U = 4;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int b = blockIdx.x;
    if (bags != nullptr) {
        const PoolBag bag = bags[blockIdx.y];
        X = static_cast<const XT*>(bag.X);
        N = bag.N;
        ldx = bag.ldx;
        if (scores != nullptr) scores += a_off[blockIdx.y];
    }
    int64_t rbeg, rend;
    rows_of_block(N, b, G, rbeg, rend);
    b += blockIdx.y * G;   // partial slot
    float acc[NC * VEC], M = -INFINITY, l = 0.f;
#pragma unroll
    for (int i = 0; i < NC * VEC; ++i) acc[i] = 0.f;
    bool live[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) live[c] = (64 * c + lane) * VEC < D;
    for (int64_t r0 = rbeg + U * w; r0 < rend; r0 += 4 * U) {
        u4 raw[U][NC];
        float t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u;
            const bool ok = r < rend;
            t[u] = ok ? (scores != nullptr ? scores[r] * kLog2e : 0.f) : -INFINITY;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                raw[u][c] = (ok && live[c]) ? *reinterpret_cast<const u4*>(X + r * ldx + (64 * c + lane) * VEC) : u4{0u, 0u, 0u, 0u};
        }
        float tm = fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3]));
        if (tm > M) {
            const float f = fast_exp2(M - tm);
            l *= f;
#pragma unroll
            for (int i = 0; i < NC * VEC; ++i) acc[i] *= f;
            M = tm;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float wgt = (t[u] == -INFINITY) ? 0.f : fast_exp2(t[u] - M);
            l += wgt;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if constexpr (sizeof(XT) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned int bits = raw[u][c][e];
                        acc[c * VEC + 2 * e] += wgt * __uint_as_float(bits << 16);
                        acc[c * VEC + 2 * e + 1] += wgt * __uint_as_float(bits & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned int bits = raw[u][c][e];
                        acc[c * VEC + e] += wgt * __uint_as_float(bits);
                    }
                }
            }
        }
    }
    float* sm = reinterpret_cast<float*>(smem);  // [4] M, [4] l, then [4][NC * 64 * VEC] acc
    float* sl = sm + 4;
    float* sacc = sm + 8;
    constexpr int DW = NC * 64 * VEC;
    if (lane == 0) {
        sm[w] = M;
        sl[w] = l;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < VEC; ++e) sacc[w * DW + (64 * c + lane) * VEC + e] = acc[c * VEC + e];
    __syncthreads();
    const float mm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float f[4], lt = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = (sm[k] == -INFINITY) ? 0.f : fast_exp2(sm[k] - mm);
        lt += sl[k] * f[k];
    }
    if (tid == 0) {
        pm[(size_t)b * kPStride] = mm;
        pl[(size_t)b * kPStride] = lt;
    }
    for (int d = tid; d < D; d += 256)
        pacc[(size_t)b * D + d] = sacc[d] * f[0] + sacc[DW + d] * f[1] + sacc[2 * DW + d] * f[2] + sacc[3 * DW + d] * f[3];
}

// ---- column max: per-workgroup partial [G, D], then a tiny reduce
template <typename XT, int DPL>
__global__ __launch_bounds__(256) void k_colmax_partial(const XT* __restrict__ X, int64_t N, int64_t ldx, int D,
                                                         float* __restrict__ part, int G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x;
    int64_t rbeg, rend;
    rows_of_block(N, b, G, rbeg, rend);
    float mx[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) mx[i] = -INFINITY;
    for (int64_t r = rbeg + w; r < rend; r += 4) {
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = lane + 64 * i;
            if (d < D) mx[i] = fmaxf(mx[i], load_as_float(X + r * ldx + d));
        }
    }
    float* s = reinterpret_cast<float*>(smem);
    constexpr int DW = DPL * 64;
#pragma unroll
    for (int i = 0; i < DPL; ++i) s[w * DW + lane + 64 * i] = mx[i];
    __syncthreads();
    for (int d = tid; d < D; d += 256)
        part[(size_t)b * D + d] = fmaxf(fmaxf(s[d], s[DW + d]), fmaxf(s[2 * DW + d], s[3 * DW + d]));
}

// workgroup = 64 columns x 4 partial subsets, 4 independent running maxima per thread (the first version walked all G
// partials in one dependent chain per column: ~50 us at G = 512)
__global__ __launch_bounds__(256) void k_colmax_merge(const float* __restrict__ part, int G, int D, float* __restrict__ out) {
    __shared__ float sm[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + c;
    float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
    if (d < D) {
        int g = q;
        for (; g + 12 < G; g += 16) {
            m0 = fmaxf(m0, part[(size_t)g * D + d]);
            m1 = fmaxf(m1, part[(size_t)(g + 4) * D + d]);
            m2 = fmaxf(m2, part[(size_t)(g + 8) * D + d]);
            m3 = fmaxf(m3, part[(size_t)(g + 12) * D + d]);
        }
        for (; g < G; g += 4) m0 = fmaxf(m0, part[(size_t)g * D + d]);
    }
    sm[q][c] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    __syncthreads();
    if (q == 0 && d < D) out[d] = fmaxf(fmaxf(sm[0][c], sm[1][c]), fmaxf(sm[2][c], sm[3][c]));
}

// 16-byte row pieces -> floats (bf16: 8 values, fp32: 4 values)
typedef unsigned int u32x4_p __attribute__((ext_vector_type(4)));
template <typename XT>
__device__ __forceinline__ void unpack16(const u32x4_p raw, float* v) {
    if constexpr (sizeof(XT) == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int bits = raw[e];
            v[2 * e] = __uint_as_float(bits << 16);
            v[2 * e + 1] = __uint_as_float(bits & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int bits = raw[e];
            v[e] = __uint_as_float(bits);
        }
    }
}

// Fast paths (rows 16-byte aligned, D a multiple of the vector width): a lane owns NC 16-byte chunks of a row (chunk c =
// elements [(64 c + lane) VEC, +VEC)), 4 rows per wave in flight.  Same outputs as the scalar kernels above.
template <typename XT, int NC>
__global__ __launch_bounds__(256) void k_colmax_partial_vec(const XT* __restrict__ X, int64_t N, int64_t ldx, int D,
                                                             float* __restrict__ part, int G) {
    constexpr int VEC = 16 / (int)sizeof(XT);
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x;
    int64_t rbeg, rend;
    rows_of_block(N, b, G, rbeg, rend);
    float mx[NC * VEC];
#pragma unroll
    for (int i = 0; i < NC * VEC; ++i) mx[i] = -INFINITY;
    for (int64_t r0 = rbeg + U * w; r0 < rend; r0 += 4 * U) {
        u32x4_p raw[U][NC];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = r0 + u < rend;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                raw[u][c] = (ok[u] && (64 * c + lane) * VEC < D) ? *reinterpret_cast<const u32x4_p*>(X + (r0 + u) * ldx + (64 * c + lane) * VEC)
                                                                 : u32x4_p{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (ok[u]) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float v[VEC];
                    unpack16<XT>(raw[u][c], v);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) mx[c * VEC + e] = fmaxf(mx[c * VEC + e], v[e]);
                }
            }
    }
    float* s = reinterpret_cast<float*>(smem);
    constexpr int DW = NC * 64 * VEC;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[w * DW + (64 * c + lane) * VEC + e] = mx[c * VEC + e];
    __syncthreads();
    for (int d = tid; d < D; d += 256)
        part[(size_t)b * D + d] = fmaxf(fmaxf(s[d], s[DW + d]), fmaxf(s[2 * DW + d], s[3 * DW + d]));
}

// With `a` given: the whole backward of the softmax-weighted row sum w.r.t. the raw scores in one pass,
//     out[n] = A_n (x_n . v - pooled . v),  A_n = exp2(a_n log2(e) - m2) / l      (v = dL/dpooled)
// instead of the row dots + four [N]-sized torch kernels.
template <typename XT, int NC>
__global__ __launch_bounds__(256) void k_rowdot_vec(const XT* __restrict__ X, int64_t N, int64_t ldx, int D,
                                                     const float* __restrict__ v, float* __restrict__ out,
                                                     const float* __restrict__ a, const float* __restrict__ m2,
                                                     const float* __restrict__ l, const float* __restrict__ pooled) {
    constexpr int VEC = 16 / (int)sizeof(XT);
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    float vv[NC * VEC];
    float delta = 0.f, m2v = 0.f, rl = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int d = (64 * c + lane) * VEC + e;
            vv[c * VEC + e] = d < D ? v[d] : 0.f;
            if (a != nullptr && d < D) delta += pooled[d] * vv[c * VEC + e];
        }
    if (a != nullptr) {
        delta = wave_sum(delta);
        m2v = m2[0];
        rl = 1.f / l[0];
    }
    for (int64_t n0 = wave * U; n0 < N; n0 += nw * U) {
        u32x4_p raw[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                raw[u][c] = (n0 + u < N && (64 * c + lane) * VEC < D) ? *reinterpret_cast<const u32x4_p*>(X + (n0 + u) * ldx + (64 * c + lane) * VEC)
                                                                      : u32x4_p{0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float x[VEC];
                unpack16<XT>(raw[u][c], x);
#pragma unroll
                for (int e = 0; e < VEC; ++e) s += x[e] * vv[c * VEC + e];
            }
            s = wave_sum(s);
            if (lane == 0 && n0 + u < N) {
                if (a != nullptr) s = fast_exp2(a[n0 + u] * kLog2e - m2v) * rl * (s - delta);
                out[n0 + u] = s;
            }
        }
    }
}

// ---- attention scores from the hidden projections: one wave per patch row
__global__ __launch_bounds__(256) void k_attn_scores(const float* __restrict__ H, const float* __restrict__ Hg, int64_t N,
                                                      int hid, const float* __restrict__ b1, const float* __restrict__ bg,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      float* __restrict__ a) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t n = wave; n < N; n += nwaves) {
        float s = 0.f;
        for (int j = lane; j < hid; j += 64) {
            float e = tanhf(H[n * hid + j] + b1[j]);
            if (Hg != nullptr) e *= 1.f / (1.f + expf(-(Hg[n * hid + j] + bg[j])));
            s += w2[j] * e;
        }
        s = wave_sum(s);
        if (lane == 0) a[n] = s + b2[0];
    }
}

// ---- (gated-)attention pooling over the P <= 16 aggregated rows of VLFAN (query_pooling = 'attention' | 'gated_attention',
// model/deepmil.py:101-105,133-150 -> model/layers.py:103-122,137-153), B bags per launch.  Two launches instead of ~10 torch ops on a
// [P, 512] matrix (those paths were host-bound: 236-282 us per bag).
// Stage 1: workgroup (hidden chunk of 4 units, bag): wave w owns hidden unit h = 4 * chunk + w: its rows of Wa (and Wg) in
// registers, the bag's P rows through LDS; part[bag][chunk][p] = sum over the 4 units of w2[h] * tanh(.) (* sigmoid(.)).
__global__ __launch_bounds__(256) void k_qpool_scores(const float* __restrict__ rows, int P, int D, const float* __restrict__ Wa,
                                                       const float* __restrict__ ba, const float* __restrict__ Wg,
                                                       const float* __restrict__ bg, const float* __restrict__ w2, int hid,
                                                       float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float srow[];   // [P][D]
    __shared__ float sred[4][VLSA_MAX_P];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int chunk = blockIdx.x, bag = blockIdx.y, nchunk = gridDim.x;
    const float* r = rows + (size_t)bag * P * D;
    for (int i = tid; i < P * D; i += 256) srow[i] = r[i];
    __syncthreads();
    const int h = 4 * chunk + w;
    float acc[VLSA_MAX_P];
#pragma unroll
    for (int p = 0; p < VLSA_MAX_P; ++p) acc[p] = 0.f;
    if (h < hid) {
        float da[VLSA_MAX_P], dg[VLSA_MAX_P];
#pragma unroll
        for (int p = 0; p < VLSA_MAX_P; ++p) da[p] = dg[p] = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float wa = Wa[(size_t)h * D + c], wg = Wg ? Wg[(size_t)h * D + c] : 0.f;
#pragma unroll
            for (int p = 0; p < VLSA_MAX_P; ++p)
                if (p < P) {
                    const float x = srow[p * D + c];
                    da[p] = fmaf(wa, x, da[p]);
                    dg[p] = fmaf(wg, x, dg[p]);
                }
        }
#pragma unroll
        for (int p = 0; p < VLSA_MAX_P; ++p)
            if (p < P) {
                const float za = wave_sum(da[p]) + ba[h];
                float e = tanhf(za);
                if (Wg) e *= 1.f / (1.f + expf(-(wave_sum(dg[p]) + bg[h])));
                acc[p] = w2[h] * e;
            }
    }
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < VLSA_MAX_P; ++p)
            if (p < P) sred[w][p] = acc[p];
    }
    __syncthreads();
    if (tid < P) part[((size_t)bag * nchunk + chunk) * VLSA_MAX_P + tid] = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
}
// Stage 2: one workgroup per bag: raw[p] = sum of the chunk partials + c; attn = softmax_P(raw); pooled = attn @ rows.
// scores_out [B, P]: the raw scores (want_raw) or the softmax weights -- what the two reference modules hand back by default.
__global__ __launch_bounds__(256) void k_qpool_finish(const float* __restrict__ rows, int P, int D, const float* __restrict__ part,
                                                       int nchunk, const float* __restrict__ c, int want_raw,
                                                       float* __restrict__ pooled, float* __restrict__ scores_out) {
    __shared__ float sa[VLSA_MAX_P];
    const int tid = threadIdx.x, bag = blockIdx.x;
    if (tid < P) {
        float s = 0.f;
        for (int k = 0; k < nchunk; ++k) s += part[((size_t)bag * nchunk + k) * VLSA_MAX_P + tid];
        sa[tid] = s + c[0];
    }
    __syncthreads();
    float mx = -INFINITY, sum = 0.f;
    for (int p = 0; p < P; ++p) mx = fmaxf(mx, sa[p]);
    for (int p = 0; p < P; ++p) sum += expf(sa[p] - mx);
    if (scores_out && tid < P) scores_out[(size_t)bag * P + tid] = want_raw ? sa[tid] : expf(sa[tid] - mx) / sum;
    const float* r = rows + (size_t)bag * P * D;
    for (int col = tid; col < D; col += 256) {
        float o = 0.f;
        for (int p = 0; p < P; ++p) o = fmaf(expf(sa[p] - mx) / sum, r[p * D + col], o);
        pooled[(size_t)bag * D + col] = o;
    }
}

// ---- out[n] = x_n . v  (pooling backward: d a_n = A_n (x_n . dpooled - pooled . dpooled))
template <typename XT>
__global__ __launch_bounds__(256) void k_rowdot(const XT* __restrict__ X, int64_t N, int64_t ldx, int D,
                                                 const float* __restrict__ v, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t n = wave; n < N; n += nwaves) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += load_as_float(X + n * ldx + d) * v[d];
        s = wave_sum(s);
        if (lane == 0) out[n] = s;
    }
}

// ---- per-class top-k mean over N (k <= 32): one workgroup per class.
// Each thread keeps the k largest of its strided subset (sorted insertion in registers), the 256 lists are
// merged by k rounds of workgroup-wide arg-max.  k >= N degenerates to the mean of all N.
constexpr int kTopKMax = 32;
__device__ __forceinline__ void topk_mean_row(const float* __restrict__ s, int64_t N, int k, float scale_log2e_inv,
                                              float* __restrict__ out_elem) {
    __shared__ float sval[256];
    __shared__ int sidx[256];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    if ((int64_t)k >= N) {  // mean over everything
        float a = 0.f;
        for (int64_t n = tid; n < N; n += 256) a += s[n];
        a = block_sum_256(a, red);
        if (tid == 0) *out_elem = a / (float)N * scale_log2e_inv;
        return;
    }
    float top[kTopKMax];
#pragma unroll
    for (int i = 0; i < kTopKMax; ++i) top[i] = -INFINITY;
    for (int64_t n = tid; n < N; n += 256) {
        float v = s[n];
        // sorted insertion (descending); static indices only
#pragma unroll
        for (int i = 0; i < kTopKMax; ++i) {
            if (i < k) {
                const float hi = fmaxf(top[i], v);
                v = fminf(top[i], v);
                top[i] = hi;
            }
        }
    }
    float sum = 0.f;
    int head = 0;  // position of this thread's best remaining candidate
    for (int round = 0; round < k; ++round) {
        float mine = -INFINITY;
#pragma unroll
        for (int i = 0; i < kTopKMax; ++i)
            if (i == head) mine = top[i];
        sval[tid] = mine;
        sidx[tid] = tid;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off && sval[tid + off] > sval[tid]) {
                sval[tid] = sval[tid + off];
                sidx[tid] = sidx[tid + off];
            }
            __syncthreads();
        }
        const int winner = sidx[0];
        sum += sval[0];
        __syncthreads();
        if (tid == winner) ++head;
    }
    if (tid == 0) *out_elem = sum / (float)k * scale_log2e_inv;
}
__global__ __launch_bounds__(256) void k_topk_mean(const float* __restrict__ S, int64_t N, int k, float scale_log2e_inv,
                                                    float* __restrict__ out) {
    const int cls = blockIdx.x;
    topk_mean_row(S + (size_t)cls * N, N, k, scale_log2e_inv, out + cls);
}
// B bags per launch: workgroup (class, bag) pools row `class` of bag's score matrix ([C, ld] fp32, e.g. the per-class cosines
// the batched streaming kernel stored); k is clamped to the bag's N (k >= N: plain mean).  out [B, C] *= exp(*logit_scale).
struct TopkBagDesc {
    const void* X;
    int64_t N;
    int64_t ldx;
};
struct TopkRowsDesc {
    float* ptr;
    int64_t ld;
};
__global__ __launch_bounds__(256) void k_topk_mean_batch(const TopkBagDesc* __restrict__ bags, const TopkRowsDesc* __restrict__ sdesc,
                                                          int C, int k, const float* __restrict__ logit_scale,
                                                          float* __restrict__ out) {
    const int cls = blockIdx.x, bag = blockIdx.y;
    const int64_t N = bags[bag].N;
    const TopkRowsDesc sd = sdesc[bag];
    if (N <= 0 || sd.ptr == nullptr) {
        if (threadIdx.x == 0) out[(size_t)bag * C + cls] = 0.f;
        return;
    }
    const float sc = logit_scale ? __expf(*logit_scale) : 1.f;
    topk_mean_row(sd.ptr + (size_t)cls * sd.ld, N, k, sc, out + (size_t)bag * C + cls);
}

// Stage 1 of the two-stage top-k for long rows: workgroup (chunk b, class) keeps the k largest of ITS contiguous chunk of
// the class's scores (same per-thread sorted insertion + k arg-max rounds) and writes them, descending, to
// part[(cls * G + b) * k ..]; k_topk_mean over the [C, G * k] candidates then finishes.  k >= N: partial sums instead.
__global__ __launch_bounds__(256) void k_topk_partial(const float* __restrict__ S, int64_t N, int k, int G,
                                                       float* __restrict__ part, int select_always = 0) {
    __shared__ float sval[256];
    __shared__ int sidx[256];
    __shared__ float red[4];
    const int tid = threadIdx.x, b = blockIdx.x, cls = blockIdx.y;
    const float* s = S + (size_t)cls * N;
    int64_t n0, n1;
    rows_of_block(N, b, G, n0, n1);
    if ((int64_t)k >= N && !select_always) {   // (select_always: the k winners are wanted even when there are fewer than k candidates: -inf padded)
        float a = 0.f;
        for (int64_t n = n0 + tid; n < n1; n += 256) a += s[n];
        a = block_sum_256(a, red);
        if (tid == 0) part[(size_t)cls * G + b] = a;
        return;
    }
    float top[kTopKMax];
#pragma unroll
    for (int i = 0; i < kTopKMax; ++i) top[i] = -INFINITY;
    for (int64_t n = n0 + tid; n < n1; n += 256) {
        float v = s[n];
#pragma unroll
        for (int i = 0; i < kTopKMax; ++i) {
            if (i < k) {
                const float hi = fmaxf(top[i], v);
                v = fminf(top[i], v);
                top[i] = hi;
            }
        }
    }
    int head = 0;
    for (int round = 0; round < k; ++round) {
        float mine = -INFINITY;
#pragma unroll
        for (int i = 0; i < kTopKMax; ++i)
            if (i == head) mine = top[i];
        sval[tid] = mine;
        sidx[tid] = tid;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off && sval[tid + off] > sval[tid]) {
                sval[tid] = sval[tid + off];
                sidx[tid] = sidx[tid + off];
            }
            __syncthreads();
        }
        const int winner = sidx[0];
        if (tid == 0) part[((size_t)cls * G + b) * k + round] = sval[0];
        __syncthreads();
        if (tid == winner) ++head;
    }
}

// sum of G partial sums per class / N (mean over everything, second stage)
__global__ __launch_bounds__(64) void k_mean_final(const float* __restrict__ part, int G, int64_t N, float scale, float* __restrict__ out) {
    const int cls = blockIdx.x;
    float a = 0.f;
    for (int g = threadIdx.x; g < G; g += 64) a += part[(size_t)cls * G + g];
    a = wave_sum(a);
    if (threadIdx.x == 0) out[cls] = a / (float)N * scale;
}

// F.normalize of MANY rows (the [N, D] patch features the reference's zero-shot forward hands back, model/vlsa.py:188-189):
// one wave per row, 16-byte loads, fp32 out.  (k_normalize_rows in vlfan_tail.hip is the few-rows, one-workgroup-per-row form.)
template <typename XT, int NC>
__global__ __launch_bounds__(256) void k_normalize_many(const XT* __restrict__ X, int64_t N, int64_t ldx, int D,
                                                         float* __restrict__ out) {
    constexpr int VEC = 16 / (int)sizeof(XT);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < N; r += nw) {
        float v[NC * VEC];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e0 = (64 * c + lane) * VEC;
            u4 raw = {0u, 0u, 0u, 0u};
            if (e0 < D) raw = *reinterpret_cast<const u4*>(X + r * ldx + e0);
            if constexpr (sizeof(XT) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned int bits = raw[e];
                    v[c * VEC + 2 * e] = __uint_as_float(bits << 16);
                    v[c * VEC + 2 * e + 1] = __uint_as_float(bits & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned int bits = raw[e];
                    v[c * VEC + e] = __uint_as_float(bits);
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) ss += v[c * VEC + e] * v[c * VEC + e];
        }
        ss = wave_sum(ss);
        const float nrm = fmaxf(sqrtf(ss), kNormEps);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int e0 = (64 * c + lane) * VEC;
            if (e0 < D) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    f32x4 o = {v[c * VEC + 4 * q] / nrm, v[c * VEC + 4 * q + 1] / nrm, v[c * VEC + 4 * q + 2] / nrm,
                               v[c * VEC + 4 * q + 3] / nrm};
                    *reinterpret_cast<f32x4*>(out + r * D + e0 + 4 * q) = o;
                }
            }
        }
    }
}

// ---- DeepMIL's Adapter head on the pooled bag vector (model/deepmil.py:283-286, model/layers.py:50-62):
//      out = keep * f + (1 - keep) * relu(W2 relu(W1 f)),  W1 [R, D], W2 [D, R], bias-free.  One wave per output row.
__global__ __launch_bounds__(256) void k_rows_dot_relu(const float* __restrict__ W, int rows, int cols, const float* __restrict__ x,
                                                        const float* __restrict__ resid, float keep, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* w = W + (size_t)r * cols;
    float s = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(w + c), b = *reinterpret_cast<const float4*>(x + c);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    s = fmaxf(wave_sum(s), 0.f);
    if (lane == 0) out[r] = resid != nullptr ? keep * resid[r] + (1.f - keep) * s : s;
}

}  // namespace vlsa

using namespace vlsa;

static inline int st() { return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH; }

extern "C" int vlsa_pool_num_partials(int64_t N) {
    const int64_t t = (N + 63) / 64;
    return (int)(t < 1 ? 1 : (t > 512 ? 512 : t));
}

template <typename XT>
static int launch_scored(const XT* X, int64_t N, int64_t ldx, int D, const float* scores, float* pm, float* pl, float* pacc,
                         int G, hipStream_t s) {
#define VLSA_SP(DPL)                                                                                                \
    {                                                                                                               \
        const size_t lds = (8 + (size_t)4 * DPL * 64) * sizeof(float);                                              \
        hipLaunchKernelGGL((k_scored_pool_partial<XT, DPL>), dim3(G), dim3(256), lds, s, X, N, ldx, D, scores, pm, pl, pacc, G); \
    }
    constexpr int VEC = 16 / (int)sizeof(XT);
    if ((D % VEC) == 0 && ((ldx * sizeof(XT)) % 16) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        const int NC = (D + 64 * VEC - 1) / (64 * VEC);
#define VLSA_SPV(NCV)                                                                                                   \
    {                                                                                                                   \
        const size_t lds = (8 + (size_t)4 * NCV * 64 * VEC) * sizeof(float);                                            \
        hipLaunchKernelGGL((k_scored_pool_partial_vec<XT, NCV>), dim3(G), dim3(256), lds, s, X, N, ldx, D, scores, pm, pl, pacc, G, \
                           static_cast<const PoolBag*>(nullptr), static_cast<const long long*>(nullptr));                    \
    }
        if (NC == 1) VLSA_SPV(1) else if (NC == 2) VLSA_SPV(2) else if (NC == 3) VLSA_SPV(3) else VLSA_SPV(4)
#undef VLSA_SPV
        return st();
    }
    if (D <= 256) VLSA_SP(4) else if (D <= 512) VLSA_SP(8) else if (D <= 768) VLSA_SP(12) else VLSA_SP(16)
#undef VLSA_SP
    return st();
}

extern "C" int vlsa_scored_pool_partial(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* scores,
                                        float* pm, float* pl, float* pacc, void* stream) {
    if (!X || !pm || !pl || !pacc || N < 1 || D < 1 || D > VLSA_MAX_D || ldx < D) return VLSA_EINVAL;
    const int G = vlsa_pool_num_partials(N);
    if (x_dtype == VLSA_DT_F32) return launch_scored<float>((const float*)X, N, ldx, D, scores, pm, pl, pacc, G, (hipStream_t)stream);
    if (x_dtype == VLSA_DT_BF16) return launch_scored<__bf16>((const __bf16*)X, N, ldx, D, scores, pm, pl, pacc, G, (hipStream_t)stream);
    return VLSA_EINVAL;
}

// B bags (D = 512, 16-byte aligned rows) in one launch: G partials per bag in the P = 1 layout of the VLFAN partials -- pm / pl
// [B * G, 16], pacc [B * G, 512] -- to be folded by vlsa_vlfan_merge_batch_strided (softmax-weighted row sum per bag; scores NULL:
// plain mean).  scores: one buffer, bag b at scores + a_off[b].
extern "C" int vlsa_scored_pool_partial_batch(const void* bag_desc, int B, int x_dtype, int D, const float* scores,
                                              const int64_t* a_off, int G, float* pm, float* pl, float* pacc, void* stream) {
    if (!bag_desc || !pm || !pl || !pacc || B < 1 || G < 1 || (scores && !a_off)) return VLSA_EINVAL;
    if (D != 512 || (x_dtype != VLSA_DT_BF16 && x_dtype != VLSA_DT_F32)) return VLSA_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const PoolBag* bags = static_cast<const PoolBag*>(bag_desc);
    const long long* off = reinterpret_cast<const long long*>(a_off);
    if (x_dtype == VLSA_DT_F32) {
        const size_t lds = (8 + (size_t)4 * 2 * 64 * 4) * sizeof(float);
        hipLaunchKernelGGL((k_scored_pool_partial_vec<float, 2>), dim3(G, B), dim3(256), lds, s, static_cast<const float*>(nullptr), (int64_t)0,
                           (int64_t)0, D, scores, pm, pl, pacc, G, bags, off);
    } else {
        const size_t lds = (8 + (size_t)4 * 1 * 64 * 8) * sizeof(float);
        hipLaunchKernelGGL((k_scored_pool_partial_vec<__bf16, 1>), dim3(G, B), dim3(256), lds, s, static_cast<const __bf16*>(nullptr), (int64_t)0,
                           (int64_t)0, D, scores, pm, pl, pacc, G, bags, off);
    }
    return st();
}

template <typename XT>
static int launch_colmax(const XT* X, int64_t N, int64_t ldx, int D, float* part, float* out, int G, hipStream_t s) {
#define VLSA_CM(DPL)                                                                                               \
    {                                                                                                              \
        const size_t lds = (size_t)4 * DPL * 64 * sizeof(float);                                                   \
        hipLaunchKernelGGL((k_colmax_partial<XT, DPL>), dim3(G), dim3(256), lds, s, X, N, ldx, D, part, G);        \
    }
    constexpr int VEC = 16 / (int)sizeof(XT);
    if ((D % VEC) == 0 && ((ldx * sizeof(XT)) % 16) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        const int NC = (D + 64 * VEC - 1) / (64 * VEC);
#define VLSA_CMV(NCV)                                                                                                  \
    {                                                                                                                  \
        const size_t lds = (size_t)4 * NCV * 64 * VEC * sizeof(float);                                                 \
        hipLaunchKernelGGL((k_colmax_partial_vec<XT, NCV>), dim3(G), dim3(256), lds, s, X, N, ldx, D, part, G);        \
    }
        if (NC == 1) VLSA_CMV(1) else if (NC == 2) VLSA_CMV(2) else if (NC == 3) VLSA_CMV(3) else VLSA_CMV(4)
#undef VLSA_CMV
    } else if (D <= 256) VLSA_CM(4) else if (D <= 512) VLSA_CM(8) else if (D <= 768) VLSA_CM(12) else VLSA_CM(16)
#undef VLSA_CM
    hipLaunchKernelGGL(k_colmax_merge, dim3((D + 63) / 64), dim3(256), 0, s, part, G, D, out);
    return st();
}

extern "C" int vlsa_colmax(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, float* partials, float* out,
                           void* stream) {
    if (!X || !partials || !out || N < 1 || D < 1 || D > VLSA_MAX_D || ldx < D) return VLSA_EINVAL;
    const int G = vlsa_pool_num_partials(N);
    if (x_dtype == VLSA_DT_F32) return launch_colmax<float>((const float*)X, N, ldx, D, partials, out, G, (hipStream_t)stream);
    if (x_dtype == VLSA_DT_BF16) return launch_colmax<__bf16>((const __bf16*)X, N, ldx, D, partials, out, G, (hipStream_t)stream);
    return VLSA_EINVAL;
}

extern "C" int vlsa_attn_scores(const float* H, const float* Hg, int64_t N, int hid, const float* b1, const float* bg,
                                const float* w2, const float* b2, float* a, void* stream) {
    if (!H || !b1 || !w2 || !b2 || !a || N < 1 || hid < 1 || (Hg && !bg)) return VLSA_EINVAL;
    int64_t nb = (N + 3) / 4;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_attn_scores, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, H, Hg, N, hid, b1, bg, w2, b2, a);
    return st();
}

// (Gated_)Attention_Pooling over the P aggregated rows of B bags (VLFAN query pooling): rows [B, P, D] -> pooled [B, D],
// scores [B, P] (raw scores if want_raw -- Attention_Pooling's default return -- else the softmax weights).  Wg / bg NULL: ungated.
// workspace: B * ceil(hid / 4) * 16 floats.
extern "C" int vlsa_query_pool_attention(const float* rows, int B, int P, int D, const float* Wa, const float* ba, const float* Wg,
                                         const float* bg, const float* w2, const float* c, int hid, int want_raw, void* workspace,
                                         float* pooled, float* scores, void* stream) {
    if (!rows || !Wa || !ba || !w2 || !c || !workspace || !pooled || (Wg && !bg)) return VLSA_EINVAL;
    if (B < 1 || P < 1 || P > VLSA_MAX_P || D < 1 || D > VLSA_MAX_D || hid < 1) return VLSA_EINVAL;
    const int nchunk = (hid + 3) / 4;
    hipStream_t s = (hipStream_t)stream;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(k_qpool_scores, dim3(nchunk, B), dim3(256), (size_t)P * D * sizeof(float), s, rows, P, D, Wa, ba, Wg, bg, w2, hid, part);
    hipLaunchKernelGGL(k_qpool_finish, dim3(B), dim3(256), 0, s, rows, P, D, part, nchunk, c, want_raw, pooled, scores);
    return st();
}

static int rowdot_impl(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* v, float* out, const float* a,
                       const float* m2, const float* l, const float* pooled, void* stream) {
    if (!X || !v || !out || N < 1 || D < 1 || ldx < D) return VLSA_EINVAL;
    int64_t nb = (N + 3) / 4;
    if (nb > 2048) nb = 2048;
    if (x_dtype != VLSA_DT_F32 && x_dtype != VLSA_DT_BF16) return VLSA_EINVAL;
    const int esz = x_dtype == VLSA_DT_F32 ? 4 : 2, VEC = 16 / esz;
    hipStream_t s = (hipStream_t)stream;
    if ((D % VEC) == 0 && ((ldx * esz) % 16) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && D <= VLSA_MAX_D) {
        const int NC = (D + 64 * VEC - 1) / (64 * VEC);
        int64_t nbv = (N + 15) / 16;
        if (nbv > 2048) nbv = 2048;
#define VLSA_RD(XT, NCV) hipLaunchKernelGGL((k_rowdot_vec<XT, NCV>), dim3((unsigned)nbv), dim3(256), 0, s, (const XT*)X, N, ldx, D, v, out, a, m2, l, pooled)
        if (x_dtype == VLSA_DT_BF16) { if (NC == 1) VLSA_RD(__bf16, 1); else VLSA_RD(__bf16, 2); }
        else { if (NC == 1) VLSA_RD(float, 1); else if (NC == 2) VLSA_RD(float, 2); else if (NC == 3) VLSA_RD(float, 3); else VLSA_RD(float, 4); }
#undef VLSA_RD
        return st();
    }
    if (a != nullptr) return VLSA_EUNSUPPORTED;
    if (x_dtype == VLSA_DT_F32)
        hipLaunchKernelGGL(k_rowdot<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)X, N, ldx, D, v, out);
    else
        hipLaunchKernelGGL(k_rowdot<__bf16>, dim3((unsigned)nb), dim3(256), 0, s, (const __bf16*)X, N, ldx, D, v, out);
    return st();
}

extern "C" int vlsa_rowdot(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* v, float* out,
                           void* stream) {
    return rowdot_impl(X, x_dtype, N, ldx, D, v, out, nullptr, nullptr, nullptr, nullptr, stream);
}

// Backward of pooled = softmax_N(a) @ X w.r.t. the raw scores (model/layers.py:114-116,145-147 under autograd), one pass over X:
// da[n] = A_n (x_n . dpooled - pooled . dpooled), A_n = exp2(a_n log2(e) - m2[0]) / l[0] with the forward's (m2, l) (log2 domain).
// 16-byte aligned rows, D % 8 == 0 (bf16) / % 4 (fp32), D <= 1024.
extern "C" int vlsa_scored_pool_backward(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const float* a, const float* m2,
                                         const float* l, const float* pooled, const float* dpooled, float* da, void* stream) {
    if (!a || !m2 || !l || !pooled) return VLSA_EINVAL;
    return rowdot_impl(X, x_dtype, N, ldx, D, dpooled, da, a, m2, l, pooled, stream);
}

extern "C" int vlsa_topk_chunks(int64_t N) {
    const int64_t g = (N + 4095) / 4096;
    return (int)(g < 1 ? 1 : (g > 128 ? 128 : g));
}
extern "C" size_t vlsa_topk_workspace_bytes(int C, int64_t N, int k) {
    return (size_t)C * vlsa_topk_chunks(N) * (size_t)(k < 1 ? 1 : (k > kTopKMax ? kTopKMax : k)) * sizeof(float);
}

extern "C" int vlsa_topk_mean_ws(const float* S, int C, int64_t N, int k, float out_scale, void* workspace, float* out,
                                 void* stream) {
    if (!S || !out || !workspace || C < 1 || N < 1 || k < 1) return VLSA_EINVAL;
    if (k > kTopKMax && (int64_t)k < N) return VLSA_EUNSUPPORTED;
    const int G = vlsa_topk_chunks(N);
    hipStream_t s = (hipStream_t)stream;
    float* part = static_cast<float*>(workspace);
    if (G == 1 || ((int64_t)k < N && (int64_t)k * 2 > N / G)) {  // short rows: the single-stage kernel
        hipLaunchKernelGGL(k_topk_mean, dim3(C), dim3(256), 0, s, S, N, k, out_scale, out);
        return st();
    }
    hipLaunchKernelGGL(k_topk_partial, dim3(G, C), dim3(256), 0, s, S, N, k, G, part);
    if ((int64_t)k >= N)
        hipLaunchKernelGGL(k_mean_final, dim3(C), dim3(64), 0, s, part, G, N, out_scale, out);
    else
        hipLaunchKernelGGL(k_topk_mean, dim3(C), dim3(256), 0, s, part, (int64_t)G * k, k, out_scale, out);
    return st();
}

// The k largest entries of every row of S [C, N], descending, -inf padded when N < k: vals [C, k].  The per-rank piece of the
// patch-sharded zero-shot pooling (all-gather local top-k [G, k, K], re-select).  workspace as vlsa_topk_mean_ws.
extern "C" int vlsa_topk_values(const float* S, int C, int64_t N, int k, void* workspace, float* vals, void* stream) {
    if (!S || !vals || !workspace || C < 1 || N < 1 || k < 1) return VLSA_EINVAL;
    if (k > kTopKMax) return VLSA_EUNSUPPORTED;
    const int G = vlsa_topk_chunks(N);
    hipStream_t s = (hipStream_t)stream;
    float* part = static_cast<float*>(workspace);
    if (G == 1) {
        hipLaunchKernelGGL(k_topk_partial, dim3(1, C), dim3(256), 0, s, S, N, k, 1, vals, 1);
        return st();
    }
    hipLaunchKernelGGL(k_topk_partial, dim3(G, C), dim3(256), 0, s, S, N, k, G, part, 1);
    hipLaunchKernelGGL(k_topk_partial, dim3(1, C), dim3(256), 0, s, part, (int64_t)G * k, k, 1, vals, 1);
    return st();
}

// Per-class top-k mean for B bags in one launch (model/deepmil.py:16-37 on the [C, N_i] class scores of every bag):
// out [B, C] = exp(*logit_scale) * mean of the min(k, N_i) largest entries of each row (k <= 0: mean of all N_i).
extern "C" int vlsa_topk_mean_batch(const void* bag_desc, const void* scores_desc, int B, int C, int k, const float* logit_scale,
                                    float* out, void* stream) {
    if (!bag_desc || !scores_desc || !out || B < 1 || C < 1) return VLSA_EINVAL;
    if (k > kTopKMax) return VLSA_EUNSUPPORTED;
    const int kk = k <= 0 ? 0x7fffffff : k;      // mean over all patches
    hipLaunchKernelGGL(k_topk_mean_batch, dim3(C, B), dim3(256), 0, (hipStream_t)stream, static_cast<const TopkBagDesc*>(bag_desc),
                       static_cast<const TopkRowsDesc*>(scores_desc), C, kk, logit_scale, out);
    return st();
}

extern "C" int vlsa_normalize_many(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, float* out, void* stream) {
    if (!X || !out || N < 0 || D < 1 || D > VLSA_MAX_D || ldx < D) return VLSA_EINVAL;
    if (N == 0) return VLSA_OK;
    const int esz = x_dtype == VLSA_DT_F32 ? 4 : 2;
    if ((D % (16 / esz)) || ((ldx * esz) % 16) || (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (D % 4))
        return VLSA_EUNSUPPORTED;
    int64_t nb = (N + 3) / 4;
    if (nb > 256 * 8) nb = 256 * 8;
    hipStream_t s = (hipStream_t)stream;
#define VLSA_NM(XT, NCV) hipLaunchKernelGGL((k_normalize_many<XT, NCV>), dim3((unsigned)nb), dim3(256), 0, s, static_cast<const XT*>(X), N, ldx, D, out)
    if (x_dtype == VLSA_DT_BF16) {
        const int NC = (D + 511) / 512;
        if (NC == 1) VLSA_NM(__bf16, 1); else VLSA_NM(__bf16, 2);
    } else if (x_dtype == VLSA_DT_F32) {
        const int NC = (D + 255) / 256;
        if (NC == 1) VLSA_NM(float, 1); else if (NC == 2) VLSA_NM(float, 2); else if (NC == 3) VLSA_NM(float, 3); else VLSA_NM(float, 4);
    } else {
        return VLSA_EINVAL;
    }
#undef VLSA_NM
    return st();
}

extern "C" int vlsa_adapter_head(const float* f, int D, const float* W1, int R, const float* W2, float keep_ratio, float* hidden,
                                 float* out, void* stream) {
    if (!f || !W1 || !W2 || !hidden || !out || D < 4 || R < 4 || (D % 4) || (R % 4)) return VLSA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_rows_dot_relu, dim3((R + 3) / 4), dim3(256), 0, s, W1, R, D, f, (const float*)nullptr, 0.f, hidden);
    hipLaunchKernelGGL(k_rows_dot_relu, dim3((D + 3) / 4), dim3(256), 0, s, W2, D, R, hidden, f, keep_ratio, out);
    return st();
}

extern "C" int vlsa_topk_mean(const float* S, int C, int64_t N, int k, float out_scale, float* out, void* stream) {
    if (!S || !out || C < 1 || N < 1 || k < 1) return VLSA_EINVAL;
    if (k > kTopKMax && (int64_t)k < N) return VLSA_EUNSUPPORTED;
    hipLaunchKernelGGL(k_topk_mean, dim3(C), dim3(256), 0, (hipStream_t)stream, S, N, k, out_scale, out);
    return st();
}

// ---- exact Shapley values of the P text prototypes for the survival risk  v(S) = sum_k (K - k) softmax_k(ls * mean_{p in S} sim[p, k]),
// v(empty) = 1 (reference utils/model_inference.py:23-79: an O(P 2^P) host loop there).  Two launches: the value of every
// coalition (one thread each), then per prototype the weighted sum of its marginal contributions (one workgroup each).
namespace vlsa {
struct ShapWeights {
    float w[16];   // w[c] = c! (P - c - 1)! / P!
};

__global__ __launch_bounds__(256) void k_shap_values(const float* __restrict__ sim, int P, int K, float ls, float* __restrict__ V) {
    const unsigned int m = blockIdx.x * 256u + threadIdx.x;
    if (m >= (1u << P)) return;
    if (m == 0) {
        V[0] = 1.f;
        return;
    }
    const float inv = 1.f / (float)__builtin_popcount(m);
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {          // two passes over the classes (K <= 64: the [P, K] table sits in L1 / scalar cache)
        float z = 0.f;
        for (int p = 0; p < P; ++p)
            if ((m >> p) & 1u) z += sim[p * K + k];
        mx = fmaxf(mx, ls * (z * inv));
    }
    float den = 0.f, num = 0.f;
    for (int k = 0; k < K; ++k) {
        float z = 0.f;
        for (int p = 0; p < P; ++p)
            if ((m >> p) & 1u) z += sim[p * K + k];
        const float e = __expf(ls * (z * inv) - mx);
        den += e;
        num += e * (float)(K - k);
    }
    V[m] = num / den;
}

__global__ __launch_bounds__(256) void k_shap_reduce(const float* __restrict__ V, int P, const ShapWeights wt, float* __restrict__ shap) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const unsigned int half = 1u << (P - 1), lowmask = (1u << i) - 1u;
    float acc = 0.f;
    for (unsigned int j = tid; j < half; j += 256) {
        const unsigned int m = ((j & ~lowmask) << 1) | (j & lowmask);      // j with a zero bit inserted at position i
        acc += wt.w[__builtin_popcount(m)] * (V[m | (1u << i)] - V[m]);
    }
    acc = block_sum_256(acc, red);
    if (tid == 0) shap[i] = acc;
}
}  // namespace vlsa

extern "C" int vlsa_prototype_shapley(const float* sim, int P, int K, float logit_scale, float* values, float* shap, void* stream) {
    if (!sim || !values || !shap || P < 1 || P > 16 || K < 1 || K > VLSA_MAX_K) return VLSA_EINVAL;
    vlsa::ShapWeights wt;
    double fac[17];
    fac[0] = 1.0;
    for (int i = 1; i <= 16; ++i) fac[i] = fac[i - 1] * i;
    for (int c = 0; c < 16; ++c) wt.w[c] = c < P ? (float)(fac[c] * fac[P - c - 1] / fac[P]) : 0.f;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vlsa::k_shap_values, dim3(((1u << P) + 255u) / 256u), dim3(256), 0, s, sim, P, K, logit_scale, values);
    hipLaunchKernelGGL(vlsa::k_shap_reduce, dim3(P), dim3(256), 0, s, values, P, wt, shap);
    return st();
}

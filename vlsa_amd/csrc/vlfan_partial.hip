// Streaming pass of the language-guided patch aggregation (VLFAN cross-attention) for gfx950.
//
// Replaces the reference's   norm_X = F.normalize(X); A_ = Q^ @ norm_X^T; A = softmax(100*A_); out = A @ X
// (model/deepmil.py:189-200), which reads X three times and writes a normalised copy, by ONE pass over
// X with an online softmax per query.  Output: per-workgroup partials (m2, l, acc).
//
// Two kernels:
//   k_vlfan_partial_mfma     D == 512.  Workgroup = 4 waves; wave w owns columns [128w, 128w+128) of every
//                            row, so X slices are wave-private in LDS and only a [32 x 16] partial-score
//                            tile is exchanged per 32-row tile.  Both contractions run on
//                            v_mfma_f32_16x16x32_bf16 with split-bf16 operands (3-term queries, 2-term
//                            softmax weights, 2-term X when the input is fp32) -- fp32-class accuracy at
//                            bf16 MFMA rate; the kernel is HBM-bound (1 KB / 2 KB per patch row).
//   k_vlfan_partial_generic  any D <= 1024 (D % 8 == 0), fp32 VALU; fallback + on-device cross-check.
#include "vlfan_mfma_common.h"

namespace vlsa {

// ---------------------------------------------------------------------------------------------------
// Query preparation: q^ = q / max(||q||, eps); effective e_p = q^_p - gated * q^_gate; 3-term bf16 split of
// scale * log2(e) * e_p (so MFMA scores come out directly in the log2 domain).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepare_queries(const float* __restrict__ Q, int nq, int D, int gated,
                                                          float scale2, unsigned char* __restrict__ qprep,
                                                          const float* __restrict__ T, float* __restrict__ That,
                                                          float* __restrict__ tnorm) {
    __shared__ float red[4];
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    if (blockIdx.x >= 17) {  // fused text-feature normalisation: F.normalize(text_features) (model/vlsa.py:186)
        const int r = blockIdx.x - 17, tid = threadIdx.x;
        const float* x = T + (size_t)r * D;
        float ss = 0.f;
        for (int d = tid; d < D; d += 256) ss += x[d] * x[d];
        ss = block_sum_256(ss, red);
        const float nrm = fmaxf(sqrtf(ss), kNormEps);
        for (int d = tid; d < D; d += 256) That[(size_t)r * D + d] = x[d] / nrm;
        if (tnorm != nullptr && tid == 0) tnorm[r] = nrm;
        return;
    }
    const QPrepLayout L(D);
    float* qeff = reinterpret_cast<float*>(qprep + L.qeff);
    __bf16* qsplit = reinterpret_cast<__bf16*>(qprep + L.qsplit);
    float* qhat = reinterpret_cast<float*>(qprep + L.qhat);
    float* qnorm = reinterpret_cast<float*>(qprep + L.qnorm);
    const int p = blockIdx.x;  // 0..16
    const int P = gated ? nq - 1 : nq;
    const int tid = threadIdx.x;
    if (p == 16) {  // block 16: unit vector + norm of the gate row only (row index nq-1 when gated)
        if (tid == 0) qnorm[31] = scale2;  // coattn scale * log2(e), read back by the generic kernel
        if (!gated) return;
        const float* q = Q + (size_t)(nq - 1) * D;
        float ss = 0.f;
        for (int d = tid; d < D; d += 256) ss += q[d] * q[d];
        ss = block_sum_256(ss, red);
        const float nrm = fmaxf(sqrtf(ss), kNormEps);
        for (int d = tid; d < D; d += 256) qhat[(size_t)(nq - 1) * D + d] = q[d] / nrm;
        if (tid == 0) qnorm[nq - 1] = nrm;
        return;
    }
    if (p >= P) {  // zero padding rows of the effective queries
        for (int d = tid; d < D; d += 256) {
            qeff[(size_t)p * D + d] = 0.f;
            for (int t = 0; t < 3; ++t) qsplit[((size_t)t * 16 + p) * D + d] = (__bf16)0.f;
        }
        return;
    }
    const float* q = Q + (size_t)p * D;
    float ss = 0.f, sg = 0.f;
    for (int d = tid; d < D; d += 256) ss += q[d] * q[d];
    ss = block_sum_256(ss, red);
    const float nrm = fmaxf(sqrtf(ss), kNormEps);
    float gn = 1.f;
    const float* qg = Q + (size_t)(nq - 1) * D;
    if (gated) {
        for (int d = tid; d < D; d += 256) sg += qg[d] * qg[d];
        sg = block_sum_256(sg, red);
        gn = fmaxf(sqrtf(sg), kNormEps);
    }
    for (int d = tid; d < D; d += 256) {
        const float u = q[d] / nrm;
        qhat[(size_t)p * D + d] = u;
        float e = u;
        if (gated) e = u - qg[d] / gn;
        qeff[(size_t)p * D + d] = e;
        const float es = e * scale2;  // the MFMA kernels contract against scale * log2(e) * e_p
        const __bf16 h0 = (__bf16)es;
        const float r1 = es - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        const __bf16 h2 = (__bf16)r2;
        qsplit[((size_t)0 * 16 + p) * D + d] = h0;
        qsplit[((size_t)1 * 16 + p) * D + d] = h1;
        qsplit[((size_t)2 * 16 + p) * D + d] = h2;
    }
    if (tid == 0) qnorm[p] = nrm;
}

// ---------------------------------------------------------------------------------------------------
// Generic fp32 VALU kernel.  grid = (G, ceil(P/4)); each wave walks rows rbeg+w, rbeg+w+4, ...
// Lane l holds elements d = l + 64 i (i < DPL).  Exact per-row online softmax.
// ---------------------------------------------------------------------------------------------------
template <typename XT, int DPL>
__global__ __launch_bounds__(256) void k_vlfan_partial_generic(const XT* __restrict__ X, int64_t N, int64_t ldx,
                                                                int D, const float* __restrict__ qeff, int P,
                                                                const float* __restrict__ qmeta, float* __restrict__ pm,
                                                                float* __restrict__ pl, float* __restrict__ pacc,
                                                                float* __restrict__ scores, int G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.x, p0 = blockIdx.y * 4;
    const float scale2 = qmeta[31];
    int64_t rbeg, rend;
    block_rows(N, b, G, rbeg, rend);

    float q[4][DPL], acc[4][DPL], M[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        M[j] = -INFINITY;
        l[j] = 0.f;
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = lane + 64 * i;
            q[j][i] = (p0 + j < P && d < D) ? qeff[(size_t)(p0 + j) * D + d] : 0.f;
            acc[j][i] = 0.f;
        }
    }
    for (int64_t r = rbeg + w; r < rend; r += 4) {
        const XT* xr = X + r * ldx;
        float x[DPL];
        float ss = 0.f, dq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < DPL; ++i) {
            const int d = lane + 64 * i;
            x[i] = d < D ? load_as_float(xr + d) : 0.f;
            ss += x[i] * x[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) dq[j] += q[j][i] * x[i];
        }
        ss = wave_sum(ss);
        const float inv = scale2 / fmaxf(sqrtf(ss), kNormEps);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = wave_sum(dq[j]) * inv;
            if (scores != nullptr && lane == 0 && p0 + j < P) scores[(size_t)(p0 + j) * N + r] = t;
            if (t > M[j]) {  // wave-uniform
                const float f = fast_exp2(M[j] - t);
                l[j] *= f;
#pragma unroll
                for (int i = 0; i < DPL; ++i) acc[j][i] *= f;
                M[j] = t;
            }
            const float wgt = fast_exp2(t - M[j]);
            l[j] += wgt;
#pragma unroll
            for (int i = 0; i < DPL; ++i) acc[j][i] += wgt * x[i];
        }
    }
    // combine the 4 waves through LDS
    float* sm = reinterpret_cast<float*>(smem);          // [4 waves][4 q]  M
    float* sl = sm + 16;                                 // [4][4]          l
    float* sacc = sl + 16;                               // [4 waves][4 q][DPL*64]
    constexpr int DW = DPL * 64;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sm[w * 4 + j] = M[j];
            sl[w * 4 + j] = l[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < DPL; ++i) sacc[(size_t)(w * 4 + j) * DW + lane + 64 * i] = acc[j][i];
    __syncthreads();
    for (int j = 0; j < 4; ++j) {
        if (p0 + j >= P) break;
        float mm = fmaxf(fmaxf(sm[j], sm[4 + j]), fmaxf(sm[8 + j], sm[12 + j]));
        float f[4], lt = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[k] = (sm[k * 4 + j] == -INFINITY) ? 0.f : fast_exp2(sm[k * 4 + j] - mm);
            lt += sl[k * 4 + j] * f[k];
        }
        if (tid == 0) {
            pm[(size_t)b * kPStride + p0 + j] = mm;
            pl[(size_t)b * kPStride + p0 + j] = lt;
        }
        for (int d = tid; d < D; d += 256) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) a += sacc[(size_t)(k * 4 + j) * DW + d] * f[k];
            pacc[((size_t)b * P + p0 + j) * D + d] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// MFMA kernel, D == 512.
//
// LDS image of a wave's X slice (32 rows x 128 cols bf16): element (row, col) lives at byte
//     row * 256 + ((col * 2) ^ ((row & 7) << 5))
// i.e. the eight 32-byte blocks of each 256-byte row are XOR-permuted by (row & 7).  With that image
//   * ds_write_b128 staging (16 lanes cover one 256-B row)           -> conflict-free
//   * ds_read_b128 of MFMA A-fragments (16 rows x one 16-B chunk)     -> conflict-free
//   * ds_read_b64_tr_b16 of MFMA B-fragments (8 rows x 32 B per half) -> conflict-free
// MFMA 16x16x32 bf16 fragment maps (lane l, g = l >> 4, i = l & 15):
//   A[i][8g..8g+7], B[8g..8g+7][i], C/D[4g + reg][i].
// Scores use A = X tile (M = patch row), B = query^T (N = query)  => C[n = 4g+reg][p = i].
// The softmax weights therefore land with p on the lane index and 8 patch rows per lane, which IS the
// A-fragment of the second contraction acc[p][c] += W[p][n] X[n][c] once MFMA k-slot (g, j) is mapped to
// tile row 16*(j>>2) + 4g + (j&3); the B-fragment is fetched with two transpose reads per 16-column tile.
// ---------------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256, 2) void k_vlfan_partial_mfma(const XT* __restrict__ X, int64_t N, int64_t ldx,
                                                                const __bf16* __restrict__ qsplit, int P,
                                                                float* __restrict__ pm,
                                                                float* __restrict__ pl, float* __restrict__ pacc,
                                                                float* __restrict__ scores, int G) {
    constexpr bool F32 = sizeof(XT) == 4;
    constexpr int NX = F32 ? 2 : 1;
    constexpr int D = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int b = blockIdx.x;
    int64_t rbeg, rend;
    block_rows(N, b, G, rbeg, rend);

    unsigned char* xs = smem + (size_t)w * kSliceBytes * NX;   // this wave's private slice image(s)
    unsigned char* exch = smem + (size_t)4 * kSliceBytes * NX;  // partial-score exchange, 2 parities

    // query B-fragments for this wave's 128 columns: B[k = c][j = p]; lane holds Q[p=i16][c0 + 32kk + 8g .. +8]
    bf16x8 qf[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            qf[t][kk] = *reinterpret_cast<const bf16x8*>(qsplit + ((size_t)t * 16 + i16) * D + w * 128 + kk * 32 + g * 8);

    f32x4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float M = -INFINITY;  // running reference max of query p = i16 (log2 domain), identical in all 4 waves
    float lsum = 0.f;     // this lane's share of sum_n exp2(t - M) for p = i16

    u32x4 st[StageN<XT>::value];
    const int64_t rlast = rend > rbeg ? rend - 1 : (N > 0 ? N - 1 : 0);
    if (N > 0) stage_load(st, X, ldx, rbeg, rlast, w, lane);
    int par = 0;
    for (int64_t r0 = rbeg; r0 < rend; r0 += kTileRows, par ^= 1) {
        // ---- stage this tile into the wave-private LDS image, then prefetch the next tile --------------
        stage_store(st, xs, lane);
#ifdef VLSA_COND_LOAD
        if (r0 + kTileRows < rend)
#endif
        stage_load(st, X, ldx, r0 + kTileRows, rlast, w, lane);  // unconditional: clamped rows are valid memory
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- contraction 1: partial scores over this wave's 128 columns + partial row sum-of-squares ----
        f32x4 S[2];
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
            // two independent accumulator chains per row group keep the matrix pipe busy
            f32x4 Sa[2], Sb[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Sa[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                Sb[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    Sa[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[0][kk], Sa[h], 0, 0, 0);
                    Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[1][kk], Sb[h], 0, 0, 0);
                    Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[2][kk], Sb[h], 0, 0, 0);
                    if constexpr (F32) {
                        Sa[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[h][kk], qf[0][kk], Sa[h], 0, 0, 0);
                        Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[h][kk], qf[1][kk], Sb[h], 0, 0, 0);
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
                ss[h] = quad_rows_sum(a);  // lanes with the same i16 now hold the wave-partial |x_n|^2, n = 16h + i16
                S[h] = Sa[h] + Sb[h];
            }
        }

        // ---- exchange the partials between the 4 waves (one barrier per tile, parity double-buffered) ----
        {
            unsigned char* mine = exch + par * kExchParity + w * kExchWave;
            *reinterpret_cast<f32x4_ma*>(mine + (0 * 64 + lane) * 16) = S[0];
            *reinterpret_cast<f32x4_ma*>(mine + (1 * 64 + lane) * 16) = S[1];
            if (g == 0) {
                typedef float __attribute__((may_alias)) float_ma;
                reinterpret_cast<float_ma*>(mine + 2048)[i16] = ss[0];
                reinterpret_cast<float_ma*>(mine + 2048)[16 + i16] = ss[1];
            }
        }
        __syncthreads();
        f32x4 T[2], R2[2];
        {
            f32x4 tv[2][4], rv[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const unsigned char* o = exch + par * kExchParity + ww * kExchWave;
                    tv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + (h * 64 + lane) * 16);
                    rv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + 2048 + (16 * h + 4 * g) * 4);
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                T[h] = (tv[h][0] + tv[h][1]) + (tv[h][2] + tv[h][3]);
                R2[h] = (rv[h][0] + rv[h][1]) + (rv[h][2] + rv[h][3]);
            }
        }

        // ---- scores -> softmax weights (log2 domain); lane holds p = i16, rows n = 16h + 4g + reg ---------
        float tmax = -INFINITY;
        bool valid[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = r0 + 16 * h + 4 * g + r;
                valid[h][r] = n < rend;
                T[h][r] *= fminf(__builtin_amdgcn_rsqf(R2[h][r]), 1e12f);  // 1 / max(|x|, 1e-12); scale is in the queries
                if (valid[h][r]) tmax = fmaxf(tmax, T[h][r]);
            }
        if (scores != nullptr && i16 < P && (w & 1) == ((g >> 1) & 1)) {
            const int h = w >> 1;  // wave w stores score group h for half of the lanes: every (h, lane) once
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (valid[h][r]) scores[(size_t)i16 * N + r0 + 16 * h + 4 * g + r] = T[h][r];
        }
        if (__builtin_amdgcn_ballot_w64(tmax > M + kRescaleThreshold) != 0) {  // rare, wave-uniform
            const float newM = fmaxf(M, quad_rows_max(tmax));
            const float f = (M == -INFINITY) ? 0.f : fast_exp2(M - newM);
            lsum *= f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float fr = __shfl(f, 4 * g + r);  // factor of query p = 4g + r (accumulator rows)
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) acc[ct][r] *= fr;
            }
            M = newM;
        }
        bf16x8 ahi, alo;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float wv = valid[h][r] ? fast_exp2(T[h][r] - M) : 0.f;
                lsum += wv;
                const __bf16 hi = (__bf16)wv;
                ahi[4 * h + r] = hi;
                alo[4 * h + r] = (__bf16)(wv - (float)hi);
            }

        // ---- contraction 2: acc[p][c] += W[p][n] X[n][c] over this wave's 8 column tiles ---------------
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

    // ---- epilogue: this workgroup's partial ---------------------------------------------------------
    lsum = quad_rows_sum(lsum);
    if (w == 0 && g == 0 && i16 < P) {
        pm[(size_t)b * kPStride + i16] = M;
        pl[(size_t)b * kPStride + i16] = lsum;
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

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
using namespace vlsa;

static int g_max_partials() {
    static int v = [] {
        const char* e = VLSA_ENV("VLSA_MAX_PARTIALS");
        int x = e ? atoi(e) : 0;
        return x > 0 ? x : 256;  // one workgroup per CU on MI355X (256 CUs)
    }();
    return v;
}

extern "C" int vlsa_num_partials(int64_t N) {
    if (N <= 0) return 1;
    const int64_t tiles = (N + kTileRows - 1) / kTileRows;
    const int gm = g_max_partials();
    return (int)(tiles < gm ? tiles : gm);
}

extern "C" size_t vlsa_qprep_bytes(int D) { return QPrepLayout(D).total; }
extern "C" const float* vlsa_qprep_qeff(const void* qprep, int D) {
    return reinterpret_cast<const float*>(static_cast<const unsigned char*>(qprep) + QPrepLayout(D).qeff);
}
extern "C" const float* vlsa_qprep_qhat(const void* qprep, int D) {
    return reinterpret_cast<const float*>(static_cast<const unsigned char*>(qprep) + QPrepLayout(D).qhat);
}
extern "C" const float* vlsa_qprep_qnorm(const void* qprep, int D) {
    return reinterpret_cast<const float*>(static_cast<const unsigned char*>(qprep) + QPrepLayout(D).qnorm);
}

extern "C" int vlsa_prepare_queries(const float* Q, int nq, int D, int gated, float coattn_scale, void* qprep,
                                    void* stream) {
    if (!Q || !qprep || D <= 0 || D > VLSA_MAX_D || (D % 8) != 0) return VLSA_EINVAL;
    const int P = gated ? nq - 1 : nq;
    if (P < 1 || P > VLSA_MAX_P || !(coattn_scale > 0.f)) return VLSA_EINVAL;
    hipLaunchKernelGGL(k_prepare_queries, dim3(17), dim3(256), 0, (hipStream_t)stream, Q, nq, D, gated,
                       coattn_scale * kLog2e, static_cast<unsigned char*>(qprep), (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

extern "C" int vlsa_prepare_queries_and_text(const float* Q, int nq, int D, int gated, float coattn_scale, void* qprep,
                                             const float* T, int K, float* That, float* tnorm, void* stream) {
    if (!Q || !qprep || !T || !That || D <= 0 || D > VLSA_MAX_D || (D % 8) != 0 || K < 1 || K > VLSA_MAX_K) return VLSA_EINVAL;
    const int P = gated ? nq - 1 : nq;
    if (P < 1 || P > VLSA_MAX_P || !(coattn_scale > 0.f)) return VLSA_EINVAL;
    hipLaunchKernelGGL(k_prepare_queries, dim3(17 + K), dim3(256), 0, (hipStream_t)stream, Q, nq, D, gated,
                       coattn_scale * kLog2e, static_cast<unsigned char*>(qprep), T, That, tnorm);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

template <typename XT>
static int launch_generic(const XT* X, int64_t N, int64_t ldx, int D, const float* qeff, int P, const float* qmeta,
                          float* pm, float* pl, float* pacc, float* scores, int G, hipStream_t s) {
    const dim3 grid(G, (P + 3) / 4), block(256);
#define VLSA_GEN(DPL)                                                                                              \
    {                                                                                                              \
        const size_t lds = (32 + (size_t)16 * DPL * 64) * sizeof(float);                                           \
        auto kern = k_vlfan_partial_generic<XT, DPL>;                                                              \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, X, N, ldx, D, qeff, P, qmeta, pm, pl, pacc, scores, G);     \
    }
    if (D <= 256) VLSA_GEN(4)
    else if (D <= 512) VLSA_GEN(8)
    else if (D <= 768) VLSA_GEN(12)
    else VLSA_GEN(16)
#undef VLSA_GEN
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

template <typename XT>
static int launch_mfma(const XT* X, int64_t N, int64_t ldx, const __bf16* qsplit, int P, float* pm,
                       float* pl, float* pacc, float* scores, int G, hipStream_t s) {
    constexpr bool F32 = sizeof(XT) == 4;
    constexpr int lds = mfma_lds_bytes<F32>();
    auto kern = k_vlfan_partial_mfma<XT>;
    static DeviceOnce attr_once;
    if (lds > 64 * 1024 && attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), lds, s, X, N, ldx, qsplit, P, pm, pl, pacc, scores, G);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

int vlsa_launch_partial_dma(const __bf16* X, int64_t N, int64_t ldx, const __bf16* qsplit_scaled, int P, float* pm,
                            float* pl, float* pacc, float* scores, int G, hipStream_t s);

int vlsa_launch_partial_f32_one(const float* X, int64_t N, int64_t ldx, const float* qeff, const float* qmeta, int P, float* pm,
                                float* pl, float* pacc, int G, hipStream_t s);  // vlfan_batch_f32.hip

extern "C" int vlsa_vlfan_partial(const void* X, int x_dtype, int64_t N, int64_t ldx, int D, const void* qprep,
                                  int P, int kernel, float* pm, float* pl, float* pacc, float* scores,
                                  void* stream) {
    if (!qprep || !pm || !pl || !pacc || N < 0 || (N > 0 && !X)) return VLSA_EINVAL;
    if (D <= 0 || D > VLSA_MAX_D || (D % 8) != 0 || P < 1 || P > VLSA_MAX_P || ldx < D) return VLSA_EINVAL;
    if (x_dtype != VLSA_DT_F32 && x_dtype != VLSA_DT_BF16) return VLSA_EINVAL;
    const size_t esz = x_dtype == VLSA_DT_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(X) & 15) != 0 || ((size_t)ldx * esz) % 16 != 0) return VLSA_EINVAL;
    const int G = vlsa_num_partials(N);
    // the DMA kernel addresses a workgroup's rows through a 32-bit buffer descriptor
    const bool dma_ok = D == 512 && x_dtype == VLSA_DT_BF16 && ((N / G + 64) * ldx * 2) < (int64_t)0x7fff0000;
    const bool auto_kernel = kernel == VLSA_KERNEL_AUTO;
    if (kernel == VLSA_KERNEL_AUTO)
        kernel = (D != 512) ? VLSA_KERNEL_GENERIC : (dma_ok ? VLSA_KERNEL_DMA : VLSA_KERNEL_MFMA);
    if (kernel == VLSA_KERNEL_MFMA && D != 512) return VLSA_EUNSUPPORTED;
    if (kernel == VLSA_KERNEL_DMA && !dma_ok) return VLSA_EUNSUPPORTED;
    if (kernel != VLSA_KERNEL_MFMA && kernel != VLSA_KERNEL_GENERIC && kernel != VLSA_KERNEL_DMA) return VLSA_EINVAL;
    const QPrepLayout L(D);
    const unsigned char* qp = static_cast<const unsigned char*>(qprep);
    hipStream_t s = (hipStream_t)stream;
    const __bf16* qsplit = reinterpret_cast<const __bf16*>(qp + L.qsplit);
    if (auto_kernel && x_dtype == VLSA_DT_F32 && D == 512 && !scores && N > 0 && ((N / G + 64) * ldx * 4) < (int64_t)0x7fff0000)
        // fp32 rows without a score output: the exact-f32 LDS-DMA streaming kernel of the batched path with one bag (vlfan_batch_f32.hip)
        return vlsa_launch_partial_f32_one((const float*)X, N, ldx, reinterpret_cast<const float*>(qp + L.qeff),
                                           reinterpret_cast<const float*>(qp + L.qnorm), P, pm, pl, pacc, G, s);
    if (kernel == VLSA_KERNEL_DMA)
        return vlsa_launch_partial_dma((const __bf16*)X, N, ldx, qsplit, P, pm, pl, pacc, scores, G, s);
    if (kernel == VLSA_KERNEL_MFMA) {
        return x_dtype == VLSA_DT_F32
                   ? launch_mfma<float>((const float*)X, N, ldx, qsplit, P, pm, pl, pacc, scores, G, s)
                   : launch_mfma<__bf16>((const __bf16*)X, N, ldx, qsplit, P, pm, pl, pacc, scores, G, s);
    }
    const float* qeff = reinterpret_cast<const float*>(qp + L.qeff);
    const float* qmeta = reinterpret_cast<const float*>(qp + L.qnorm);
    return x_dtype == VLSA_DT_F32
               ? launch_generic<float>((const float*)X, N, ldx, D, qeff, P, qmeta, pm, pl, pacc, scores, G, s)
               : launch_generic<__bf16>((const __bf16*)X, N, ldx, D, qeff, P, qmeta, pm, pl, pacc, scores, G, s);
}

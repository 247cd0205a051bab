// k_vlfan_partial_dma: the tuned streaming kernel for bf16 bags with D == 512 (CONCH features).
//
// Same math and partial format as k_vlfan_partial_mfma (vlfan_partial.hip) -- see the layout notes there --
// re-structured around what the first profile showed (profiles/r01_*): the v1 kernel was instruction-issue
// bound (47 % of wave cycles issuing, 64-bit address math, IEEE sqrt/div expansions) and had one tile of
// loads in flight per wave.  Here:
//   * X tiles go HBM -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) into a
//     2-slot ring per wave; no staging VGPRs, no ds_write pass, no 64-bit address arithmetic; rows past the
//     workgroup's row range read as zero through the buffer descriptor's bounds check.  The XOR swizzle of
//     the LDS image is applied on the per-lane SOURCE address (LDS-DMA writes lane-linear).
//   * workgroup = 8 waves = 2 row groups x 4 column quarters: two waves per SIMD hide each other's LDS /
//     MFMA / DMA latencies while the grid still produces one partial per CU (256 per bag).
//   * row norms come from the diagonal of X X^T on the matrix pipe (no VALU dot products, no cross-lane
//     reduction); 1/|x| is v_rsq_f32; the coattn scale * log2(e) is folded into the prepared queries.
//   * counted s_waitcnt vmcnt(N) + raw s_barrier keep the next tile's DMA in flight across the exchange.
#include "vlsa_common.h"

namespace vlsa {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef bf16x8 __attribute__((may_alias)) bf16x8_ma;
typedef f32x4 __attribute__((may_alias)) f32x4_ma;
typedef float __attribute__((may_alias)) float_ma;
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef VLSA_TIMING
__device__ long long vlsa_dbg_cycles[32];
#define VLSA_STAMP(k)                                                                      \
    do {                                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0) vlsa_dbg_cycles[k] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define VLSA_STAMP(k) do {} while (0)
#endif

namespace dma {
constexpr int kTile = 32;                       // rows per tile
constexpr int kSlot = kTile * 256;              // 8 KiB: one wave's slice image of a tile
constexpr int kWaveRing = 2 * kSlot;            // 2 slots
constexpr int kRingBytes = 8 * kWaveRing;       // 128 KiB
constexpr int kExchWave = 2048 + 128;           // S partials (2 x f32x4 per lane) + 32 row sums of squares
constexpr int kExchGroup = 4 * kExchWave;
constexpr int kLdsBytes = kRingBytes + 2 * kExchGroup;  // 148,480 B
constexpr float kThr = 16.0f;                   // rescale threshold, log2 units
}  // namespace dma

__device__ __forceinline__ int swz_off(int row, int byte_off) { return row * 256 + (byte_off ^ ((row & 7) << 5)); }

#define VLSA_LDS_BARRIER()                                   \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                        \
        asm volatile("" ::: "memory");                       \
    } while (0)

template <bool WANT_SCORES>
__global__ __launch_bounds__(512, 2) void k_vlfan_partial_dma(const __bf16* __restrict__ X, int64_t N, int64_t ldx,
                                                               const __bf16* __restrict__ qsplit, int P,
                                                               float* __restrict__ pm, float* __restrict__ pl,
                                                               float* __restrict__ pacc, float* __restrict__ scores,
                                                               int uq, int ur) {
    using namespace dma;
    constexpr int D = 512;
    VLSA_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = w >> 2, cw = w & 3;  // row group, column quarter
    const int g = lane >> 4, i16 = lane & 15;
    const int b = blockIdx.x;

    // rows of this workgroup, balanced at 16-row granularity: workgroup b owns uq (+1 if b < ur) 16-row units
    // (uq = units / G, ur = units % G computed on the host: no 64-bit division on the device)
    const int64_t ubeg = (int64_t)b * uq + (b < ur ? b : ur);
    const int64_t rbeg = ubeg << 4;
    int64_t rend = (ubeg + uq + (b < ur ? 1 : 0)) << 4;
    if (rend > N) rend = N;
    const int nrows = rend > rbeg ? (int)(rend - rbeg) : 0;
    const int ntiles = (nrows + kTile - 1) / kTile;
    const int niter = (ntiles + 1) >> 1;  // both row groups run the same number of (lock-step) iterations

    unsigned char* ring = smem + w * kWaveRing;
    unsigned char* exch = smem + kRingBytes + rg * kExchGroup;

    // buffer descriptor over exactly this workgroup's rows: anything past `rend` reads as zero
    const __bf16* xbase = X + rbeg * ldx;
    const unsigned int span = nrows > 0 ? (unsigned int)(((int64_t)(nrows - 1) * ldx + D) * 2) : 0u;
    // raw buffer descriptor, built by hand and pinned to SGPRs: {base_lo, base_hi (stride 0), num_records, flags}
    const uint64_t xaddr = reinterpret_cast<uint64_t>(xbase);
    i32x4 rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)xaddr);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((xaddr >> 32) & 0xffffu));
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)span);
    rsrc[3] = 0x00020000;
    const int ldb = (int)(ldx * 2);  // row pitch in bytes
    // LDS-DMA lands lane l at slot byte 16 l of the 1-KiB piece (rows 4i + (l >> 4)); to realise the swizzled
    // image the lane fetches source chunk (l & 15) ^ ((row & 7) << 1); row & 7 = (l >> 4) + 4 (i & 1).
    const int lr = lane >> 4;
    const int voff_e = lr * ldb + cw * 256 + (((lane & 15) ^ (lr << 1)) << 4);
    const int voff_o = lr * ldb + cw * 256 + (((lane & 15) ^ (lr << 1) ^ 8) << 4);

    // The DMA is issued from inline asm on purpose: hipcc would otherwise order every later ds_read of the ring
    // behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)), which serialises the prefetch; we count it ourselves.
    const unsigned int ring_lds = (unsigned int)(uintptr_t)(lds_void_ptr)ring;  // LDS byte address of this wave's ring
    auto issue_tile = [&](int tile, int slot) {
        const int sbase = tile * kTile * ldb;
        const unsigned int dst = ring_lds + slot * kSlot;
        unsigned int keep;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst + i * 1024), "v"((i & 1) ? voff_o : voff_e), "s"(rsrc), "s"(sbase + i * 4 * ldb)
                : "memory");
        }
    };

    VLSA_STAMP(1);
    if (rg < ntiles) issue_tile(rg, 0);  // first tile goes in flight before anything else touches memory
    // query B-fragments (scale * log2 e already folded in): lane holds Q[p = i16][128 cw + 32 kk + 8 g .. +8]
    bf16x8 qf[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            qf[t][kk] = *reinterpret_cast<const bf16x8*>(qsplit + ((size_t)t * 16 + i16) * D + cw * 128 + kk * 32 + g * 8);

    f32x4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float M = -INFINITY, lsum = 0.f;

    // Retire the query-fragment loads here, in a way hipcc can see (a register use): otherwise its own
    // s_waitcnt vmcnt(0) for them lands inside the loop and drains our in-flight DMA every iteration.
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[t][kk]));
    VLSA_STAMP(2);

    for (int it = 0; it < niter; ++it) {
        const int tile = 2 * it + rg;
        const int slot = it & 1;
        const bool have = tile < ntiles;           // wave-uniform
        const bool have_next = tile + 2 < ntiles;  // wave-uniform
        // every ds_read of the slot we are about to refill was consumed by an MFMA of the previous iteration
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (have_next) {
            issue_tile(tile + 2, slot ^ 1);
            if constexpr (WANT_SCORES)  // the score stores share the vm counter: drain everything (slower path)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else  // this tile's 8 pieces have landed; the next 8 stay in flight
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char* xs = ring + slot * kSlot;
        const int row0 = tile * kTile;  // relative to rbeg
        if (it == 0) VLSA_STAMP(3);

        // ---- contraction 1: partial scores over this wave's 128 columns; |x|^2 from the diagonal of X X^T -----
        f32x4 S[2], Nd[2];
        {
            bf16x8 xa[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    xa[h][kk] = *reinterpret_cast<const bf16x8_ma*>(xs + swz_off(16 * h + i16, kk * 64 + g * 16));
            __builtin_amdgcn_sched_barrier(0);  // keep the 8 fragment reads batched ahead of the MFMA chain
            f32x4 Sa[2], Sb[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Sa[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                Sb[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                Nd[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (have) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        Sa[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[0][kk], Sa[h], 0, 0, 0);
                        Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[1][kk], Sb[h], 0, 0, 0);
                        Nd[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], xa[h][kk], Nd[h], 0, 0, 0);
                        Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[2][kk], Sb[h], 0, 0, 0);
                    }
            }
            S[0] = Sa[0] + Sb[0];
            S[1] = Sa[1] + Sb[1];
        }

        if (it == 0) VLSA_STAMP(4);
        // ---- exchange the partials between the 4 column-quarter waves of this row group ------------------------
        VLSA_LDS_BARRIER();  // all readers of the previous tile's exchange are done
        {
            unsigned char* mine = exch + cw * kExchWave;
            *reinterpret_cast<f32x4_ma*>(mine + (0 * 64 + lane) * 16) = S[0];
            *reinterpret_cast<f32x4_ma*>(mine + (1 * 64 + lane) * 16) = S[1];
            if (g == (i16 >> 2)) {  // this lane's register (i16 & 3) holds the diagonal element (n, n), n = i16
                const int r = i16 & 3;
                const float d0 = r == 0 ? Nd[0][0] : r == 1 ? Nd[0][1] : r == 2 ? Nd[0][2] : Nd[0][3];
                const float d1 = r == 0 ? Nd[1][0] : r == 1 ? Nd[1][1] : r == 2 ? Nd[1][2] : Nd[1][3];
                reinterpret_cast<float_ma*>(mine + 2048)[i16] = d0;
                reinterpret_cast<float_ma*>(mine + 2048)[16 + i16] = d1;
            }
        }
        VLSA_LDS_BARRIER();
        if (it == 0) VLSA_STAMP(5);
        if (have) {
            f32x4 T[2], R2[2];
            {
                f32x4 tv[2][4], rv[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        const unsigned char* o = exch + ww * kExchWave;
                        tv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + (h * 64 + lane) * 16);
                        rv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + 2048 + (16 * h + 4 * g) * 4);
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    T[h] = (tv[h][0] + tv[h][1]) + (tv[h][2] + tv[h][3]);
                    R2[h] = (rv[h][0] + rv[h][1]) + (rv[h][2] + rv[h][3]);
                }
            }

            // ---- scores -> softmax weights (log2 domain); lane holds p = i16, rows n = 16h + 4g + reg ----------
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float inv = fminf(__builtin_amdgcn_rsqf(R2[h][r]), 1e12f);  // 1 / max(|x|, 1e-12)
                    T[h][r] *= inv;
                }
            if (row0 + kTile > nrows) {  // wave-uniform: only the workgroup's last tile is ragged
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row0 + 16 * h + 4 * g + r >= nrows) T[h][r] = -INFINITY;
            }
            if constexpr (WANT_SCORES) {
                if (i16 < P && (cw & 1) == (g >> 1)) {
                    // each (h, lane) pair is stored by exactly one of the 4 waves; static register indices only
                    float* dst = scores + (size_t)i16 * N + rbeg + row0 + 4 * g;
                    if (cw < 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (T[0][r] != -INFINITY) dst[r] = T[0][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (T[1][r] != -INFINITY) dst[16 + r] = T[1][r];
                    }
                }
            }
            const float tmax = fmaxf(fmaxf(fmaxf(T[0][0], T[0][1]), fmaxf(T[0][2], T[0][3])),
                                     fmaxf(fmaxf(T[1][0], T[1][1]), fmaxf(T[1][2], T[1][3])));
            if (__builtin_amdgcn_ballot_w64(tmax > M + kThr) != 0) {  // rare after the first tile; wave-uniform
                const float newM = fmaxf(M, quad_rows_max(tmax));
                const float f = (M == -INFINITY) ? 0.f : fast_exp2(M - newM);
                lsum *= f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float fr = __shfl(f, 4 * g + r);
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
                    const float wv = fast_exp2(T[h][r] - M);  // exp2(-inf) = 0 for masked rows (M is finite here)
                    lsum += wv;
                    const __bf16 hi = (__bf16)wv;
                    ahi[4 * h + r] = hi;
                    alo[4 * h + r] = (__bf16)(wv - (float)hi);
                }

            if (it == 0) VLSA_STAMP(6);
            // ---- contraction 2: acc[p][c] += W[p][n] X[n][c] over this wave's 8 column tiles -------------------
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const int c_off = ct * 32 + (i16 & 3) * 8;
                const int rr = 4 * g + (i16 >> 2);
                const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + swz_off(rr, c_off)));
                const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + swz_off(16 + rr, c_off)));
                const bf16x8 bh = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bh, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bh, acc[ct], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: merge row group 1 into row group 0 through LDS, then write the workgroup's partial ------
    VLSA_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VLSA_LDS_BARRIER();
    lsum = quad_rows_sum(lsum);
    unsigned char* mg = smem + (4 + cw) * kWaveRing;  // row group 1's wave (4 + cw) lends its ring: 16 KiB
    float* ml = reinterpret_cast<float*>(smem + kRingBytes);  // exchange area: [cw][2][16] (M, l) of row group 1
    if (rg == 1) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) *reinterpret_cast<f32x4_ma*>(mg + (ct * 64 + lane) * 16) = acc[ct];
        if (g == 0) {
            reinterpret_cast<float_ma*>(ml)[cw * 32 + i16] = M;
            reinterpret_cast<float_ma*>(ml)[cw * 32 + 16 + i16] = lsum;
        }
    }
    VLSA_LDS_BARRIER();
    VLSA_STAMP(8);
    if (rg == 0) {
        const float M1 = reinterpret_cast<const float_ma*>(ml)[cw * 32 + i16];
        const float l1 = reinterpret_cast<const float_ma*>(ml)[cw * 32 + 16 + i16];
        const float Mn = fmaxf(M, M1);
        const float f0 = (M == -INFINITY) ? 0.f : fast_exp2(M - Mn);
        const float f1 = (M1 == -INFINITY) ? 0.f : fast_exp2(M1 - Mn);
        const float lt = lsum * f0 + l1 * f1;
        if (cw == 0 && g == 0 && i16 < P) {
            pm[(size_t)b * kPStride + i16] = Mn;
            pl[(size_t)b * kPStride + i16] = lt;
        }
        float a0[4], a1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a0[r] = __shfl(f0, 4 * g + r);
            a1[r] = __shfl(f1, 4 * g + r);
        }
        // merged accumulator -> LDS as a [16 p][128 c] fp32 tile (this wave's own ring), then 16-byte row stores
        unsigned char* tp = smem + cw * kWaveRing;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const f32x4 other = *reinterpret_cast<const f32x4_ma*>(mg + (ct * 64 + lane) * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                reinterpret_cast<float_ma*>(tp)[(4 * g + r) * 128 + ct * 16 + i16] = acc[ct][r] * a0[r] + other[r] * a1[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private round trip: in-order LDS, just drain
        float* dstp = pacc + (size_t)b * P * D + cw * 128 + (lane & 31) * 4;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = 2 * k + (lane >> 5);
            const f32x4 v = *reinterpret_cast<const f32x4_ma*>(tp + (p * 128 + (lane & 31) * 4) * 4);
            if (p < P) *reinterpret_cast<f32x4*>(dstp + (size_t)p * D) = v;
        }
    }
    VLSA_STAMP(9);
#ifdef VLSA_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VLSA_STAMP(10);
#endif
}

}  // namespace vlsa

using namespace vlsa;

#ifdef VLSA_TIMING
extern "C" int vlsa_debug_read_cycles(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vlsa::vlsa_dbg_cycles), sizeof(long long) * 32) == hipSuccess ? 0 : -3;
}
#endif

// Called from vlsa_vlfan_partial (vlfan_partial.hip).
int vlsa_launch_partial_dma(const __bf16* X, int64_t N, int64_t ldx, const __bf16* qsplit_scaled, int P, float* pm,
                            float* pl, float* pacc, float* scores, int G, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)k_vlfan_partial_dma<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  dma::kLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_vlfan_partial_dma<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  dma::kLdsBytes);
    }
    const int64_t units = (N + 15) >> 4;
    const int uq = (int)(units / G), ur = (int)(units % G);
    if (scores != nullptr)
        hipLaunchKernelGGL(k_vlfan_partial_dma<true>, dim3(G), dim3(512), dma::kLdsBytes, s, X, N, ldx, qsplit_scaled, P, pm,
                           pl, pacc, scores, uq, ur);
    else
        hipLaunchKernelGGL(k_vlfan_partial_dma<false>, dim3(G), dim3(512), dma::kLdsBytes, s, X, N, ldx, qsplit_scaled, P, pm,
                           pl, pacc, scores, uq, ur);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

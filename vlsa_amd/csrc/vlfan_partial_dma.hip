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
    // Round 6 -- the start of the kernel in the order that keeps HBM busy (profiles/r05_dma_stamps.txt: half of the kernel was ramp):
    //   1. the query B-fragments (L2 hits, 12 x 16 B per lane) go out FIRST, as inline-asm loads: the vm counter retires loads in
    //      order, so behind a DMA tile they would wait for the whole first round of HBM traffic (round 5: +7.9 k cycles), and as
    //      compiler-visible loads hipcc's own `s_waitcnt vmcnt(0)` for them would drain the DMA ring;
    //   2. BOTH ring slots are requested before anything is waited for (round 5 requested the second tile only after the first
    //      had landed: HBM idled between the two rounds);
    //   3. `s_waitcnt vmcnt(<pieces behind the fragments>)` retires the fragments alone.
    // lane holds Q[p = i16][128 cw + 32 kk + 8 g .. +8] of the three split terms (scale * log2 e already folded in)
    bf16x8 qf[3][4];
    {
        const __bf16* q0 = qsplit + (size_t)i16 * D + cw * 128 + g * 8;
        const __bf16* q1 = q0 + (size_t)16 * D;
        const __bf16* q2 = q0 + (size_t)32 * D;
#define VLSA_QLOAD(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr) : "memory")
        VLSA_QLOAD(qf[0][0], q0, 0);   VLSA_QLOAD(qf[0][1], q0, 64);  VLSA_QLOAD(qf[0][2], q0, 128); VLSA_QLOAD(qf[0][3], q0, 192);
        VLSA_QLOAD(qf[1][0], q1, 0);   VLSA_QLOAD(qf[1][1], q1, 64);  VLSA_QLOAD(qf[1][2], q1, 128); VLSA_QLOAD(qf[1][3], q1, 192);
        VLSA_QLOAD(qf[2][0], q2, 0);   VLSA_QLOAD(qf[2][1], q2, 64);  VLSA_QLOAD(qf[2][2], q2, 128); VLSA_QLOAD(qf[2][3], q2, 192);
#undef VLSA_QLOAD
    }
    const bool first = rg < ntiles, second = rg + 2 < ntiles;      // wave-uniform
    if (first) issue_tile(rg, 0);
    // every wave's FIRST tile is queued before anybody's second (the CU serves its waves' requests in issue order, and the first
    // exchange needs all four column quarters of tile 0: without this barrier wave 3's tile 0 sat behind three second tiles --
    // profiles/r06_dma_stamps.txt, first attempt: fragments +6.3 k, wave 0's tile +7.3 k, but the first exchange still at +16 k)
    __builtin_amdgcn_s_barrier();
    if (second) issue_tile(rg + 2, 1);

    f32x4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float M = -INFINITY, lsum = 0.f;

    if (second) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the fragments are defined from here on (a register use hipcc can see, behind the wait)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[t][kk]));
    VLSA_STAMP(2);

    for (int it = 0; it < niter; ++it) {
        const int tile = 2 * it + rg;
        const int slot = it & 1;
        const bool have = tile < ntiles;           // wave-uniform
        const bool have_next = tile + 2 < ntiles;  // wave-uniform: the OTHER slot holds (or is receiving) this wave's next tile
        // This tile's 8 pieces have landed once at most the next tile's 8 are outstanding.  (The score stores of WANT_SCORES share the
        // counter; loads retire in order among themselves, so "<= 8 outstanding" with 8 younger DMA pieces issued still means that
        // every piece of THIS tile is in -- whatever the stores do.)
        if (have_next) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        // the slot just consumed takes the tile after next: every ds_read of it has returned (they fed the MFMAs above)
        if (tile + 4 < ntiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_tile(tile + 4, slot);
        }
    }

    // ---- epilogue: the two row groups merge into ONE partial.  Round 6: BOTH halves of the workgroup work -- wave (rg, cw) parks the
    // four column tiles the OTHER row group will finish in its own (free) ring, takes the partner's four, merges, transposes through
    // LDS and stores 16-byte row pieces: 4 KB per wave instead of 8 KB by half of the waves (round 5: 4.2 k cycles from "loop done" to
    // "stores drained" with row group 1 idle behind the first barrier).  The merged value is the same expression as before --
    // acc(rg 0) * f(rg 0) + acc(rg 1) * f(rg 1) -- whichever wave evaluates it.
    VLSA_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VLSA_LDS_BARRIER();
    lsum = quad_rows_sum(lsum);
    unsigned char* park = smem + w * kWaveRing;                          // this wave's ring: [0, 4 KB) the tiles it hands over
    const unsigned char* theirs = smem + ((rg ^ 1) * 4 + cw) * kWaveRing;   // ... and where the partner (rg ^ 1, cw) parked ours
    float* ml = reinterpret_cast<float*>(smem + kRingBytes);             // exchange area: [rg][cw][2][16] (M, l)
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // rg 0 keeps column tiles 0..3 and parks 4..7; rg 1 the other way round (static register indices)
        const f32x4 v = rg ? acc[k] : acc[4 + k];
        *reinterpret_cast<f32x4_ma*>(park + (k * 64 + lane) * 16) = v;
    }
    if (g == 0) {
        reinterpret_cast<float_ma*>(ml)[(rg * 4 + cw) * 32 + i16] = M;
        reinterpret_cast<float_ma*>(ml)[(rg * 4 + cw) * 32 + 16 + i16] = lsum;
    }
    VLSA_LDS_BARRIER();
    VLSA_STAMP(8);
    {
        const float Mo = reinterpret_cast<const float_ma*>(ml)[((rg ^ 1) * 4 + cw) * 32 + i16];
        const float lo = reinterpret_cast<const float_ma*>(ml)[((rg ^ 1) * 4 + cw) * 32 + 16 + i16];
        const float Mn = fmaxf(M, Mo);
        const float fs = (M == -INFINITY) ? 0.f : fast_exp2(M - Mn);     // this row group's factor
        const float fo = (Mo == -INFINITY) ? 0.f : fast_exp2(Mo - Mn);   // the partner's
        const float f0 = rg ? fo : fs, f1 = rg ? fs : fo;                // factor of row group 0 / of row group 1
        if (rg == 0 && cw == 0 && g == 0 && i16 < P) {
            pm[(size_t)b * kPStride + i16] = Mn;
            pl[(size_t)b * kPStride + i16] = lsum * f0 + lo * f1;
        }
        float a0[4], a1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a0[r] = __shfl(f0, 4 * g + r);
            a1[r] = __shfl(f1, 4 * g + r);
        }
        // merged [16 p][64 c] fp32 tile -> LDS (the second slot of this wave's own ring), then 16-byte row stores
        unsigned char* tp = park + kSlot;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 mine = rg ? acc[4 + k] : acc[k];
            const f32x4 other = *reinterpret_cast<const f32x4_ma*>(theirs + (k * 64 + lane) * 16);
            const f32x4 x0 = rg ? other : mine, x1 = rg ? mine : other;   // row group 0's / row group 1's accumulators
#pragma unroll
            for (int r = 0; r < 4; ++r)
                reinterpret_cast<float_ma*>(tp)[(4 * g + r) * 64 + k * 16 + i16] = x0[r] * a0[r] + x1[r] * a1[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private round trip: in-order LDS, just drain
        float* dstp = pacc + (size_t)b * P * D + cw * 128 + rg * 64 + (lane & 15) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = 4 * k + (lane >> 4);
            const f32x4 v = *reinterpret_cast<const f32x4_ma*>(tp + (p * 64 + (lane & 15) * 4) * 4);
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

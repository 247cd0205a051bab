// fp32 bags (the reference's own storage format: dataset/PatchWSI.py:205-215 returns float32 features) through the
// persistent multi-bag streaming kernel: the score contraction runs EXACTLY in fp32 on v_mfma_f32_16x16x4_f32 (f32 in /
// f32 accumulate, bit-equal to an fmaf chain, 157 TFLOP/s peak -- MI355X_MICROARCH.md), so the attention weights carry no
// split-bf16 approximation of X; the weighted row sum runs on the bf16 pipe from on-the-fly hi + lo splits (2^-17 relative).  Structure = k_vlfan_partial_dma_batch (vlfan_batch.hip); what differs:
//   * tile = 16 rows x 128 fp32 columns per wave (8 KiB slot, same ring); a lock-step iteration covers 32 rows;
//   * LDS-DMA piece = 2 rows x 512 B; the swizzle XORs the 16-byte chunk index with (row & 7) on the SOURCE address;
//   * fragments are single floats: A[i][k] -> lane (i = l & 15, k = l >> 4); ds_read_b32 (2-way conflict on the score
//     reads, conflict-free on the weighted-sum reads);
//   * weighted sum: v_mfma_f32_16x16x16_bf16 with k = tile row: the four softmax weights the lane already holds (p = i16,
//     rows 4g .. 4g + 3) ARE its A fragment, the four floats X[4g .. 4g + 3][16 ct + i16] its B fragment -- both split
//     into bf16 hi + lo in registers (no second LDS image);
//   * row norms on the VALU from the score fragments (32 FMAs + 2 cross-quad adds per tile).
// Matrix-pipe load per 16-row tile per wave: 32 f32 steps (scores, 32 cycles each) + 24 bf16 steps (sum, 16 cycles) = 1408
// cycles; with the sum in f32 as well (round 1) it was 2048 = ~95 % of the pipe at HBM rate, and the kernel sat at 62-64 %.
#include "vlsa_common.h"
#ifndef VLSA_F32_ABL
#define VLSA_F32_ABL 0     // timing-only ablations / schedule variants (tools/f32_ablate.py); 0 = the product
#endif

namespace vlsa {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef f32x4 __attribute__((may_alias)) f32x4_ma;
typedef float __attribute__((may_alias)) float_ma;
typedef int __attribute__((may_alias)) int_ma;
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct BagDesc {
    const void* X;
    int64_t N;
    int64_t ldx;
};
struct RowsDesc {  // one [P, ld] fp32 matrix per bag (mirrors vlsa_rows_desc in vlsa_hip.h)
    float* ptr;
    int64_t ld;
};

namespace bf {
constexpr int kTile = 16;                        // rows per tile
constexpr int kSlot = kTile * 512;               // 8 KiB: 16 rows x 128 fp32 columns
constexpr int kWaveRing = 2 * kSlot;
constexpr int kRingBytes = 8 * kWaveRing;        // 128 KiB
constexpr int kExchWave = 1024 + 64;             // S partials (one f32x4 per lane) + 16 row sums of squares
constexpr int kExchGroup = 4 * kExchWave;
constexpr int kTabOff = kRingBytes + 2 * kExchGroup;
constexpr int kTabInts = 12;                     // 8 stream-descriptor ints + score pointer (lo, hi) + score pitch + pad
constexpr int kMaxLocal = 64;                    // bags per workgroup (LDS table entries); a launch takes up to S x that
constexpr int kMlOff = kTabOff + kMaxLocal * kTabInts * 4;
constexpr int kLdsBytes = kMlOff + 8 * 32 * 4;
constexpr float kThr = 16.0f;
}  // namespace bf

// element (row, col) of a wave's fp32 slice image (16 rows x 128 columns) lives at
// row * 512 + (((col >> 2) ^ row) << 4) + (col & 3) * 4: the 16-B chunk index is XORed with the 4-bit row, so that both the
// score reads (16 rows x one chunk per quarter wave) and the weighted-sum reads (4 rows x 16 words) hit 64 distinct banks
__device__ __forceinline__ int fswz(int row, int col) { return row * 512 + ((((col >> 2) ^ (row & 15))) << 4) + ((col & 3) << 2); }

#define VLSA_FBAR()                                          \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                        \
        asm volatile("" ::: "memory");                       \
    } while (0)

// S = number of workgroup groups: bag t is streamed by the Gb = G / S workgroups of group t % S only, so S bags are in
// flight at once, every workgroup sees S times more rows per bag (fewer bag epilogues, better tile quantisation) and a
// bag leaves Gb instead of G partials behind.
// kScores: see k_vlfan_partial_dma_batch (vlfan_batch.hip) -- optional per-bag store of the normalised log2-domain scores.
template <bool kScores>
__global__ __launch_bounds__(512, 2) void k_vlfan_partial_f32_batch(const BagDesc* __restrict__ bags, int B,
                                                                     const float* __restrict__ qeff, const float* __restrict__ qmeta, int P,
                                                                     float* __restrict__ pm, float* __restrict__ pl,
                                                                     float* __restrict__ pacc, int S,
                                                                     const RowsDesc* __restrict__ sdesc, const BagDesc one) {
    using namespace bf;
    constexpr int D = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = w >> 2, cw = w & 3;
    const int g = lane >> 4, i16 = lane & 15;
    const int Gb = gridDim.x / S;            // workgroups (and partials) per bag
    const int grp = blockIdx.x / Gb, b = blockIdx.x % Gb, G = Gb;

    unsigned char* ring = smem + w * kWaveRing;
    unsigned char* exch = smem + kRingBytes + rg * kExchGroup;
    int_ma* tab = reinterpret_cast<int_ma*>(smem + kTabOff);

    // ---- bag table: the workgroup's own bags grp, grp + S, ... only (local index lb <-> bag grp + lb * S, <= kMaxLocal entries
    // whatever B is, see k_vlfan_partial_dma_batch); thread lb describes this workgroup's rows of its lb-th bag
    const int nloc = grp < B ? (B - grp + S - 1) / S : 0;
    if (tid < nloc) {
        const int bag_id = grp + tid * S;
        const BagDesc d = bags ? bags[bag_id] : one;   // bags == null: ONE bag, described in the kernel arguments (single-slide calls)
        // 32-row units (= one lock-step iteration of the two row groups); the workgroup that gets the remainder
        // unit rotates with the bag index so that the extra iterations even out over the batch
        const unsigned long long units = (unsigned long long)((d.N + 31) >> 5);
        const unsigned int uq = (unsigned int)(units / (unsigned int)G), ur = (unsigned int)(units % (unsigned int)G);
        const unsigned int vb = (unsigned int)((b + tid * 37) % G);  // virtual workgroup index for this bag
        constexpr bool mine = true;
        const unsigned long long ubeg = (unsigned long long)vb * uq + (vb < ur ? vb : ur);
        const long long rbeg = (long long)(ubeg << 5);
        long long rend = (long long)((ubeg + uq + (vb < ur ? 1u : 0u)) << 5);
        if (rend > d.N) rend = d.N;
        const int nrows = (mine && rend > rbeg) ? (int)(rend - rbeg) : 0;
        const unsigned long long addr = reinterpret_cast<unsigned long long>(d.X) + (unsigned long long)rbeg * d.ldx * 4ull;
        int_ma* e = tab + tid * kTabInts;
        if constexpr (kScores) {
            const RowsDesc sd = sdesc[bag_id];
            const unsigned long long sp = sd.ptr ? reinterpret_cast<unsigned long long>(sd.ptr + rbeg) : 0ull;
            e[8] = (int)(unsigned int)sp;
            e[9] = (int)(sp >> 32);
            e[10] = (int)sd.ld;
        }
        e[0] = (int)(unsigned int)addr;
        e[1] = (int)((addr >> 32) & 0xffffu);
        e[2] = nrows > 0 ? (int)(((long long)(nrows - 1) * d.ldx + D) * 4) : 0;  // descriptor span in bytes
        e[3] = (int)(d.ldx * 4);                                                    // row pitch in bytes
        e[4] = nrows;
        e[5] = (nrows + kTile - 1) / kTile;
        e[6] = (int)vb;  // partial slot of this workgroup for this bag
        e[7] = mine ? 1 : 0;
    }
    // query B-fragments, fp32, scale * log2(e) applied here: lane holds e_p[p = i16][128 cw + 4 kk + g], kk = 0..31
    float qf[32];
    {
        const float sc = qmeta[31];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk)  // MFMA step kk = 4 j + r contracts column 16 j + 4 g + r (k-slot g): see the score reads
            qf[kk] = qeff[(size_t)i16 * D + cw * 128 + 16 * (kk >> 2) + 4 * g + (kk & 3)] * sc;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) asm volatile("" : "+v"(qf[kk]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tab_get = [&](int bag, int k) -> int { return __builtin_amdgcn_readfirstlane(tab[bag * kTabInts + k]); };

    const unsigned int ring_lds = (unsigned int)(uintptr_t)(lds_void_ptr)ring;
    const int lr = lane >> 5, lc = lane & 31;
    // LDS-DMA of one 32-row tile of `bag` into ring slot `slot` (see k_vlfan_partial_dma for the layout)
    // descriptor of the bag the DMA currently streams from, cached in SGPRs (reloaded from the table on a bag change)
    int ib = -1, ildb = 0;
    int voff[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // row of piece i within the tile is 2 i + lr
    i32x4 rsrc = {0, 0, 0, 0x00020000};
    auto issue_tile = [&](int bag, int tile, int slot) {
        if (bag != ib) {
            const int4 e = *reinterpret_cast<const int4*>(smem + kTabOff + bag * (kTabInts * 4));
            rsrc[0] = __builtin_amdgcn_readfirstlane(e.x);
            rsrc[1] = __builtin_amdgcn_readfirstlane(e.y);
            rsrc[2] = __builtin_amdgcn_readfirstlane(e.z);
            ildb = __builtin_amdgcn_readfirstlane(e.w);
#pragma unroll
            for (int q = 0; q < 8; ++q) voff[q] = lr * ildb + cw * 512 + ((lc ^ (2 * q + lr)) << 4);
            ib = bag;
        }
        const int ldb = ildb;
        const int sbase = tile * kTile * ldb;
        const unsigned int dst = ring_lds + slot * kSlot;
        unsigned int keep;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst + i * 1024), "v"(voff[i]), "s"(rsrc), "s"(sbase + i * 2 * ldb)
                : "memory");
        }
    };
    // this row group's next own tile after (bag, tile): same bag if it has one, else the first of a later bag
    auto next_of = [&](int bag, int tile, int ntiles_bag, int& nb, int& nt) {
        if (tile + 2 < ntiles_bag) {
            nb = bag;
            nt = tile + 2;
            return;
        }
        nb = bag + 1;
        while (nb < nloc && tab_get(nb, 5) <= rg) ++nb;
        nt = rg;
    };

    int kown = 0;      // own tiles consumed so far by this wave; own tile k lives in ring slot k & 1
    int k0 = 0, k1 = 0;  // tiles consumed so far by row group 0 / 1 (for the epilogue's free-slot bookkeeping)
    {
        int fb = 0;  // first own tile of the whole batch
        while (fb < nloc && tab_get(fb, 5) <= rg) ++fb;
        if (fb < nloc) issue_tile(fb, rg, 0);
    }

    
    for (int bag = 0; bag < nloc; ++bag) {   // `bag` = local index; the batch's bag index is grp + bag * S
        const int nrows = tab_get(bag, 4), ntiles = tab_get(bag, 5);
        const int niter = (ntiles + 1) >> 1;
        f32x4 acc[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        float M = -INFINITY, lsum = 0.f;
        float* srow = nullptr;   // kScores: this lane's query row of the bag's score matrix, at this workgroup's first row
        if constexpr (kScores) {
            const unsigned long long sp = (unsigned long long)(unsigned int)tab_get(bag, 8) |
                                          ((unsigned long long)(unsigned int)tab_get(bag, 9) << 32);
            if (sp != 0 && cw == 0 && i16 < P) srow = reinterpret_cast<float*>(sp) + (size_t)i16 * tab_get(bag, 10) + 4 * g;
        }

        for (int it = 0; it < niter; ++it) {
            const int tile = 2 * it + rg;
            const bool have = tile < ntiles;  // wave-uniform
            const int slot = kown & 1;
            const unsigned char* xs = ring + slot * kSlot;
            const int row0 = tile * kTile;
            f32x4 S = {0.f, 0.f, 0.f, 0.f};
            float ss = 0.f;
            if (have) {
                int nb, nt;
                next_of(bag, tile, ntiles, nb, nt);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads of slot^1's old contents have returned
                if (nb < nloc) {
                    issue_tile(nb, nt, slot ^ 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this tile landed; the next 8 pieces stay in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                // contraction 1 on the f32 matrix pipe: S[n][p] += X[n][c] e_p[c]; A fragment = one float per lane
                // lane (row i16, k-slot g) reads 4 consecutive columns 16 j + 4 g .. + 3 with ONE ds_read_b128 and feeds them to
                // the 4 MFMA steps 4 j .. 4 j + 3 (the query fragments use the same column permutation)
                float xa[32];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4_ma*>(xs + fswz(i16, 16 * j + 4 * g));
                    xa[4 * j] = v[0];
                    xa[4 * j + 1] = v[1];
                    xa[4 * j + 2] = v[2];
                    xa[4 * j + 3] = v[3];
                }
                f32x4 Sb = {0.f, 0.f, 0.f, 0.f};
#if (VLSA_F32_ABL & 4)
                S[0] = xa[0] + xa[31];      // timing only: no score MFMAs
#elif (VLSA_F32_ABL & 32)
                f32x4 Sc = {0.f, 0.f, 0.f, 0.f}, Sd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 32; kk += 4) {
                    S = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kk], qf[kk], S, 0, 0, 0);
                    Sb = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kk + 1], qf[kk + 1], Sb, 0, 0, 0);
                    Sc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kk + 2], qf[kk + 2], Sc, 0, 0, 0);
                    Sd = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kk + 3], qf[kk + 3], Sd, 0, 0, 0);
                }
                S = (S + Sb) + (Sc + Sd);
#else
#pragma unroll
                for (int kk = 0; kk < 32; kk += 2) {
                    S = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kk], qf[kk], S, 0, 0, 0);
                    Sb = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kk + 1], qf[kk + 1], Sb, 0, 0, 0);
                }
                S += Sb;
#endif
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int kk = 0; kk < 32; kk += 2) {
                    s0 = fmaf(xa[kk], xa[kk], s0);
                    s1 = fmaf(xa[kk + 1], xa[kk + 1], s1);
                }
                ss = quad_rows_sum(s0 + s1);  // lanes with the same i16 hold the wave-partial |x_n|^2, n = i16
            }

            VLSA_FBAR();  // readers of the previous exchange are done
            {
                unsigned char* mine = exch + cw * kExchWave;
                *reinterpret_cast<f32x4_ma*>(mine + lane * 16) = S;
                if (g == 0) reinterpret_cast<float_ma*>(mine + 1024)[i16] = ss;
            }
            VLSA_FBAR();
            if (have) {
                f32x4 T, R2;
                {
                    f32x4 tv[4], rv[4];
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        const unsigned char* o = exch + ww * kExchWave;
                        tv[ww] = *reinterpret_cast<const f32x4_ma*>(o + lane * 16);
                        rv[ww] = *reinterpret_cast<const f32x4_ma*>(o + 1024 + 16 * g);
                    }
                    T = (tv[0] + tv[1]) + (tv[2] + tv[3]);
                    R2 = (rv[0] + rv[1]) + (rv[2] + rv[3]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) T[r] *= fminf(__builtin_amdgcn_rsqf(R2[r]), 1e12f);
                if (row0 + kTile > nrows) {  // ragged last tile of this workgroup's range
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row0 + 4 * g + r >= nrows) T[r] = -INFINITY;
                }
                if constexpr (kScores) {  // the four column-quarter waves hold identical scores: wave cw = 0 stores them
                    if (srow != nullptr) *reinterpret_cast<f32x4*>(srow + row0) = T;   // write-back: the partner row group's half of the line follows within the iteration
                }
                const float tmax = fmaxf(fmaxf(T[0], T[1]), fmaxf(T[2], T[3]));
                if (__builtin_amdgcn_ballot_w64(tmax > M + kThr) != 0) {
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
                float wv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    wv[r] = fast_exp2(T[r] - M);
                    lsum += wv[r];
                }
                // contraction 2: acc[p][c] += W[p][n] X[n][c]; A = wv[0..3] (this lane: p = i16, rows 4g .. 4g + 3), B = X[4g + r][16 ct + i16]
                float xb[4][8];
#pragma unroll
                for (int rs = 0; rs < 4; ++rs)
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct) xb[rs][ct] = *reinterpret_cast<const float_ma*>(xs + fswz(4 * g + rs, 16 * ct + i16));
                // the weighted sum on the bf16 pipe: weights and rows as hi + lo bf16 pairs (16 mantissa bits each), three of the
                // four products (2^-17 relative; the scores above -- hence the attention weights -- stay exact fp32).  One
                // 16x16x16 step per column tile and term instead of four f32 16x16x4 steps at twice the cycles: 384 instead of
                // 1024 matrix-pipe cycles per tile -- with both contractions in f32 the pipe was ~95 % busy at HBM rate, i.e. the bound.
                bf16x4 whi, wlo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    whi[r] = (__bf16)wv[r];
                    wlo[r] = (__bf16)(wv[r] - (float)whi[r]);
                }
#if (VLSA_F32_ABL & 2)
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) acc[ct][0] += xb[0][ct] + xb[1][ct] + xb[2][ct] + xb[3][ct];   // timing only: reads stay
#elif (VLSA_F32_ABL & 16)
                bf16x4 xh[8], xl[8];
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        xh[ct][r] = (__bf16)xb[r][ct];
                        xl[ct][r] = (__bf16)(xb[r][ct] - (float)xh[ct][r]);
                    }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi, xh[ct], acc[ct], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wlo, xh[ct], acc[ct], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi, xl[ct], acc[ct], 0, 0, 0);
#else
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    bf16x4 xh, xl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        xh[r] = (__bf16)xb[r][ct];
                        xl[r] = (__bf16)(xb[r][ct] - (float)xh[r]);
                    }
#if (VLSA_F32_ABL & 1)
                    acc[ct][0] += (float)xh[0] + (float)xl[1] + (float)xh[2] + (float)xl[3] + (float)xl[0] + (float)xh[1] + (float)xl[2] + (float)xh[3];   // timing only: splits stay, no MFMAs
#else
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi, xh, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wlo, xh, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi, xl, acc[ct], 0, 0, 0);
#endif
                }
#endif
                ++kown;
            }
            
        }
        k0 += (ntiles + 1) >> 1;
        k1 += ntiles >> 1;

        // ---- bag epilogue.  Waves (0, cw) and (1, cw) each park the half of their accumulators the partner merges in
        // their just-consumed ring slot (the other slot holds the next bag's first tile, already in flight); after ONE
        // barrier wave (rg, cw) merges column tiles [4 rg, 4 rg + 4) of quarter cw from both, transposes them through
        // the other half of its own slot and stores 16-byte row pieces.  A second barrier frees the slots for the ring.
        lsum = quad_rows_sum(lsum);
        const int kmine = rg == 0 ? k0 : k1, kother = rg == 0 ? k1 : k0;
        unsigned char* myslot = ring + (((kmine - 1) & 1) * kSlot);
        const unsigned char* otherslot = smem + ((rg ^ 1) * 4 + cw) * kWaveRing + (((kother - 1) & 1) * kSlot);
        float_ma* mlw = reinterpret_cast<float_ma*>(smem + kMlOff);  // [8 waves][2][16]: (M, l) of every wave
        // first 4 KiB of my free slot: the 4 column tiles the partner wave merges; last 4 KiB: my transpose tile
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_ma*>(myslot + (j * 64 + lane) * 16) = rg == 0 ? acc[4 + j] : acc[j];
        if (g == 0) {
            mlw[w * 32 + i16] = M;
            mlw[w * 32 + 16 + i16] = lsum;
        }
        VLSA_FBAR();
        {
            const int wo = (rg ^ 1) * 4 + cw;
            // one round of LDS reads: both waves' reference maxima for the 4 queries of my accumulator rows, the
            // partner's normaliser, and the partner's 4 parked column tiles
            const f32x4 Mm4 = *reinterpret_cast<const f32x4_ma*>(&mlw[w * 32 + 4 * g]);
            const f32x4 Mo4 = *reinterpret_cast<const f32x4_ma*>(&mlw[wo * 32 + 4 * g]);
            const float Mo = mlw[wo * 32 + i16], lo = mlw[wo * 32 + 16 + i16];
            f32x4 oth[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) oth[j] = *reinterpret_cast<const f32x4_ma*>(otherslot + (j * 64 + lane) * 16);
            float am[4], ao[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float mn = fmaxf(Mm4[r], Mo4[r]);
                am[r] = (Mm4[r] == -INFINITY) ? 0.f : fast_exp2(Mm4[r] - mn);
                ao[r] = (Mo4[r] == -INFINITY) ? 0.f : fast_exp2(Mo4[r] - mn);
            }
            const size_t slotg = (size_t)(grp + bag * S) * G + tab_get(bag, 6);
            if (w == 0 && g == 0 && i16 < P) {
                const float Mn = fmaxf(M, Mo);
                const float fm = (M == -INFINITY) ? 0.f : fast_exp2(M - Mn);
                const float fo = (Mo == -INFINITY) ? 0.f : fast_exp2(Mo - Mn);
                pm[slotg * kPStride + i16] = Mn;
                pl[slotg * kPStride + i16] = lsum * fm + lo * fo;
            }
            float_ma* tp = reinterpret_cast<float_ma*>(myslot + 4096);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 mine = rg == 0 ? acc[j] : acc[4 + j];
#pragma unroll
                for (int r = 0; r < 4; ++r) tp[(4 * g + r) * 64 + j * 16 + i16] = mine[r] * am[r] + oth[j][r] * ao[r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float* dstp = pacc + slotg * P * D + cw * 128 + rg * 64 + (lane & 15) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = 4 * k + (lane >> 4);
                const f32x4 v = *reinterpret_cast<const f32x4_ma*>(reinterpret_cast<unsigned char*>(tp) + (p * 64 + (lane & 15) * 4) * 4);
                if (p < P) *reinterpret_cast<f32x4*>(dstp + (size_t)p * D) = v;
            }
        }
        VLSA_FBAR();  // lent slots and the transpose area are free again
        
    }
}
#ifdef VLSA_TIMING
extern "C" __global__ void k_dummy_batch_dbg() {}
#endif

}  // namespace vlsa

using namespace vlsa;

int vlsa_launch_partial_f32_batch(const void* bag_desc, int B, const float* qeff, const float* qmeta, int P, float* pm,
                                  float* pl, float* pacc, int S, int workgroups, const void* scores_desc, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)k_vlfan_partial_f32_batch<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bf::kLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_vlfan_partial_f32_batch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bf::kLdsBytes);
    }
    if (scores_desc)
        hipLaunchKernelGGL(k_vlfan_partial_f32_batch<true>, dim3(workgroups), dim3(512), bf::kLdsBytes, s,
                           static_cast<const BagDesc*>(bag_desc), B, qeff, qmeta, P, pm, pl, pacc, S,
                           static_cast<const RowsDesc*>(scores_desc), BagDesc{nullptr, 0, 0});
    else
        hipLaunchKernelGGL(k_vlfan_partial_f32_batch<false>, dim3(workgroups), dim3(512), bf::kLdsBytes, s,
                           static_cast<const BagDesc*>(bag_desc), B, qeff, qmeta, P, pm, pl, pacc, S,
                           static_cast<const RowsDesc*>(nullptr), BagDesc{nullptr, 0, 0});
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// ONE fp32 bag through the same kernel (the drop-in's bag-by-bag calls, runner/vlsa_handler.py:322-330): G workgroups, G partials
// in the layout of the single-bag kernels (pm / pl [G, 16], pacc [G, P, 512]); the bag is described in the kernel arguments, so
// no descriptor table has to be uploaded.  No score output (its rows would need the padded pitch of vlsa_rows_desc).
int vlsa_launch_partial_f32_one(const float* X, int64_t N, int64_t ldx, const float* qeff, const float* qmeta, int P, float* pm,
                                float* pl, float* pacc, int G, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)k_vlfan_partial_f32_batch<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bf::kLdsBytes);
    hipLaunchKernelGGL(k_vlfan_partial_f32_batch<false>, dim3(G), dim3(512), bf::kLdsBytes, s, static_cast<const BagDesc*>(nullptr), 1,
                       qeff, qmeta, P, pm, pl, pacc, 1, static_cast<const RowsDesc*>(nullptr), BagDesc{X, N, ldx});
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

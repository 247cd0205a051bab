// Batched (multi-bag) forward: ONE persistent launch of the LDS-DMA streaming kernel over a list of bags, then a
// batched merge and a batched incidence head -- three launches for B slides.
//
// Why: at 50k patches a bag streams in ~9 us but a launch of the single-bag kernel has a fixed ~8 us latency chain
// (descriptor/TLB misses, first tile, epilogue; profiles/README.md).  Here every workgroup walks its row range of bag
// 0, 1, 2, ... with the DMA ring running straight across bag boundaries: the first tile of the next bag is already in
// flight while the current bag's partial is merged and stored, so the fixed cost is paid once per batch and the HBM
// stream never drains.  The evaluation loop of the reference (runner/vlsa_handler.py:315-345) and its 32-bag training
// step (189-289) process independent bags back to back -- exactly this shape.
//
// Per-tile arithmetic, LDS image, exchange and epilogue are those of k_vlfan_partial_dma (vlfan_partial_dma.hip).
#include "vlsa_common.h"
#ifndef VLSA_DMA_NT
#define VLSA_DMA_NT "nt"      // streaming rows: non-temporal (measurement builds may pass -DVLSA_DMA_NT=\"\")
#endif

namespace vlsa {

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef bf16x8 __attribute__((may_alias)) bf16x8_ma;
typedef f32x4 __attribute__((may_alias)) f32x4_ma;
typedef float __attribute__((may_alias)) float_ma;
typedef int __attribute__((may_alias)) int_ma;
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct BagDesc {  // device-side description of one bag (mirrors vlsa_bag_desc in vlsa_hip.h)
    const void* X;
    int64_t N;
    int64_t ldx;
};
struct RowsDesc {  // one [P, ld] fp32 matrix per bag (mirrors vlsa_rows_desc in vlsa_hip.h)
    float* ptr;
    int64_t ld;
};

#ifdef VLSA_TIMING
__device__ long long vlsa_dbg_batch[64];
#define BSTAMP(k)                                                                                            \
    do {                                                                                                     \
        const int k_ = (k);                                                                                  \
        if (blockIdx.x == 3 && threadIdx.x == 0 && k_ < 40) vlsa_dbg_batch[k_] = __builtin_readcyclecounter(); \
        if (blockIdx.x == 3 && threadIdx.x == 0 && (k_ == 1 || k_ == 39)) vlsa_dbg_batch[k_ == 1 ? 62 : 63] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define ISTAMP(k, dep)                                                                                                   \
    do {                                                                                                                 \
        if (kown == 12) {                                                                                                 \
            float sink_ = (dep);                                                                                          \
            asm volatile("v_mov_b32 %0, %0" : "+v"(sink_));                                                              \
            if (blockIdx.x == 3 && threadIdx.x == 0) vlsa_dbg_batch[40 + (k)] = __builtin_readcyclecounter();            \
        }                                                                                                                \
    } while (0)
#else
#define BSTAMP(k) do {} while (0)
#define ISTAMP(k, dep) do {} while (0)
#endif

namespace bt {
constexpr int kTile = 32;
constexpr int kSlot = kTile * 256;
constexpr int kWaveRing = 2 * kSlot;
constexpr int kRingBytes = 8 * kWaveRing;        // 128 KiB
constexpr int kExchWave = 2048 + 128;
constexpr int kExchGroup = 4 * kExchWave;
constexpr int kTabOff = kRingBytes + 2 * kExchGroup;  // bag table: kTabInts ints per bag
constexpr int kTabInts = 12;                          // 8 stream-descriptor ints + score pointer (lo, hi) + score pitch + pad
constexpr int kMaxLocal = 64;                         // bags per WORKGROUP (LDS table entries); a launch takes S x that, <= kMaxBags
constexpr int kMaxBags = 256;
constexpr int kMlOff = kTabOff + kMaxLocal * kTabInts * 4;  // (M, l) hand-off: 8 waves x 32 floats
constexpr int kLdsBytes = kMlOff + 8 * 32 * 4;        // 152,576 B
constexpr float kThr = 16.0f;
}  // namespace bt

__device__ __forceinline__ int bswz(int row, int byte_off) { return row * 256 + (byte_off ^ ((row & 7) << 5)); }

#define VLSA_BAR()                                           \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                        \
        asm volatile("" ::: "memory");                       \
    } while (0)

// S = number of workgroup groups: bag t is streamed by the Gb = G / S workgroups of group t % S only, so S bags are in
// flight at once, every workgroup sees S times more rows per bag (fewer bag epilogues, better tile quantisation) and a
// bag leaves Gb instead of G partials behind.
// kScores: additionally store the normalised log2-domain scores t_pn of every bag whose entry of `sdesc` has a non-null
// pointer ([P, ld] fp32 per bag, ld % 4 == 0, ld >= N rounded up to 64) -- the raw material of the attention weights
// A = softmax_N (model/deepmil.py:198,206-215), which need the bag-global (m, l) and are finished by k_attn_normalise_batch.
template <bool kScores>
__global__ __launch_bounds__(512, 2) void k_vlfan_partial_dma_batch(const BagDesc* __restrict__ bags, int B,
                                                                     const __bf16* __restrict__ qsplit, int P,
                                                                     float* __restrict__ pm, float* __restrict__ pl,
                                                                     float* __restrict__ pacc, int S,
                                                                     const RowsDesc* __restrict__ sdesc) {
    using namespace bt;
    constexpr int D = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = w >> 2, cw = w & 3;
    const int g = lane >> 4, i16 = lane & 15;
#ifdef VLSA_TIMING
    const int xmode = S >> 8;               // timing experiments (results are then wrong)
    S &= 255;
#else
    constexpr int xmode = 0;
#endif
    const int Gb = gridDim.x / S;            // workgroups (and partials) per bag
    const int grp = blockIdx.x / Gb, b = blockIdx.x % Gb, G = Gb;

    unsigned char* ring = smem + w * kWaveRing;
    unsigned char* exch = smem + kRingBytes + rg * kExchGroup;
    int_ma* tab = reinterpret_cast<int_ma*>(smem + kTabOff);

    // ---- bag table: the workgroup streams the bags grp, grp + S, grp + 2 S, ... only, so the LDS table holds THOSE bags
    // (local index lb <-> bag grp + lb * S: at most kMaxLocal entries whatever B is); thread lb describes this workgroup's
    // rows of its lb-th bag
    const int nloc = grp < B ? (B - grp + S - 1) / S : 0;
    if (tid < nloc) {
        const int bag_id = grp + tid * S;
        const BagDesc d = bags[bag_id];
        // 64-row units (= one lock-step iteration of the two row groups); the workgroup that gets the remainder
        // unit rotates with the bag index so that the extra iterations even out over the batch
        const unsigned long long units = (unsigned long long)((d.N + 63) >> 6);
        const unsigned int uq = (unsigned int)(units / (unsigned int)G), ur = (unsigned int)(units % (unsigned int)G);
        const unsigned int vb = (unsigned int)((b + tid * 37) % G);  // virtual workgroup index for this bag
        constexpr bool mine = true;
        const unsigned long long ubeg = (unsigned long long)vb * uq + (vb < ur ? vb : ur);
        const long long rbeg = (long long)(ubeg << 6);
        long long rend = (long long)((ubeg + uq + (vb < ur ? 1u : 0u)) << 6);
        if (rend > d.N) rend = d.N;
        const int nrows = (mine && rend > rbeg) ? (int)(rend - rbeg) : 0;
        const unsigned long long addr = reinterpret_cast<unsigned long long>(d.X) + (unsigned long long)rbeg * d.ldx * 2ull;
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
        e[2] = nrows > 0 ? (int)(((long long)(nrows - 1) * d.ldx + D) * 2) : 0;  // descriptor span in bytes
        e[3] = (int)(d.ldx * 2);                                                    // row pitch in bytes
        e[4] = nrows;
        e[5] = (nrows + kTile - 1) / kTile;
        e[6] = (int)vb;  // partial slot of this workgroup for this bag
        e[7] = mine ? 1 : 0;
    }
    // query B-fragments (scale * log2 e folded in): lane holds Q[p = i16][128 cw + 32 kk + 8 g .. +8]
    bf16x8 qf[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            qf[t][kk] = *reinterpret_cast<const bf16x8*>(qsplit + ((size_t)t * 16 + i16) * D + cw * 128 + kk * 32 + g * 8);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[t][kk]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tab_get = [&](int bag, int k) -> int { return __builtin_amdgcn_readfirstlane(tab[bag * kTabInts + k]); };

    const unsigned int ring_lds = (unsigned int)(uintptr_t)(lds_void_ptr)ring;
    const int lr = lane >> 4;
    const int chunk_e = ((lane & 15) ^ (lr << 1)) << 4, chunk_o = ((lane & 15) ^ (lr << 1) ^ 8) << 4;
    // LDS-DMA of one 32-row tile of `bag` into ring slot `slot` (see k_vlfan_partial_dma for the layout)
    // descriptor of the bag the DMA currently streams from, cached in SGPRs (reloaded from the table on a bag change)
    int ib = -1, ildb = 0, voff_e = 0, voff_o = 0;
    i32x4 rsrc = {0, 0, 0, 0x00020000};
    auto issue_tile = [&](int bag, int tile, int slot) {
        if (bag != ib) {
            const int4 e = *reinterpret_cast<const int4*>(smem + kTabOff + bag * (kTabInts * 4));
            rsrc[0] = __builtin_amdgcn_readfirstlane(e.x);
            rsrc[1] = __builtin_amdgcn_readfirstlane(e.y);
            rsrc[2] = __builtin_amdgcn_readfirstlane(e.z);
            ildb = __builtin_amdgcn_readfirstlane(e.w);
            voff_e = lr * ildb + cw * 256 + chunk_e;
            voff_o = lr * ildb + cw * 256 + chunk_o;
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
                "buffer_load_dwordx4 %2, %3, %4 offen " VLSA_DMA_NT " lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(dst + i * 1024), "v"((i & 1) ? voff_o : voff_e), "s"(rsrc), "s"(sbase + i * 4 * ldb)
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

    int stamp = 0;
    BSTAMP(stamp++);
    for (int bag = 0; bag < nloc; ++bag) {   // `bag` = local index; the batch's bag index is grp + bag * S
        const int nrows = tab_get(bag, 4), ntiles = tab_get(bag, 5);
        const int niter = (ntiles + 1) >> 1;
        f32x4 acc[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        float M = -INFINITY, lsum = 0.f;
        // kScores: the score store is transposed across the wave so that ONE instruction writes whole 128-byte lines: lane l of
        // wave cw (0 / 1) stores rows 4 (l & 7) .. + 3 of the tile for query 8 cw + (l >> 3) -- 8 lanes x 16 B = the 32 rows of a
        // query, contiguous -- as ordinary write-back stores: the lines are complete, so L2 takes them without a fill, and the
        // in-place normalise launch that follows finds part of them still cached.  (Round-2 history: the lane's own f32x4 put
        // the two 64-byte halves of every line into different waves' stores, and nontemporal stores were the lesser evil:
        // 62.9 % of the 1072 B/patch roofline; whole lines + write-back: 65.7-68.5 %.  Deferring the store behind the next tile's
        // DMA -- vmcnt retires in order -- changed nothing.  What remains is traffic, not stalls: scores out, scores in, A out
        // = 144 B/patch really move where the roofline counts 48.)
        float* srow = nullptr;   // this lane's store address for tile row 0 of the workgroup's range
        bool has_scores = false; // wave-uniform: this bag stores scores and this wave is one of the two that do
        if constexpr (kScores) {
            const unsigned long long sp = (unsigned long long)(unsigned int)tab_get(bag, 8) |
                                          ((unsigned long long)(unsigned int)tab_get(bag, 9) << 32);
            const int sp_q = 8 * cw + (lane >> 3);
            has_scores = sp != 0 && cw < 2 && 8 * cw < P;
            if (has_scores && sp_q < P) srow = reinterpret_cast<float*>(sp) + (size_t)sp_q * tab_get(bag, 10) + 4 * (lane & 7);
        }

        for (int it = 0; it < niter; ++it) {
            const int tile = 2 * it + rg;
            const bool have = tile < ntiles;  // wave-uniform
            const int slot = kown & 1;
            const unsigned char* xs = ring + slot * kSlot;
            const int row0 = tile * kTile;
            f32x4 S[2], Nd[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                S[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                Nd[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (have) {
                int nb, nt;
                next_of(bag, tile, ntiles, nb, nt);
                ISTAMP(0, 0.f);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads of slot^1's old contents have returned
                if (nb < nloc) {
                    issue_tile(nb, nt, slot ^ 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this tile landed; the next 8 pieces stay in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                ISTAMP(1, 0.f);
                bf16x8 xa[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        xa[h][kk] = *reinterpret_cast<const bf16x8_ma*>(xs + bswz(16 * h + i16, kk * 64 + g * 16));
                __builtin_amdgcn_sched_barrier(0);
                f32x4 Sb[2];
                Sb[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                Sb[1] = Sb[0];
                if (!(xmode & 1))
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        S[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[0][kk], S[h], 0, 0, 0);
                        Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[1][kk], Sb[h], 0, 0, 0);
                        Nd[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], xa[h][kk], Nd[h], 0, 0, 0);
                        Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[2][kk], Sb[h], 0, 0, 0);
                    }
                S[0] += Sb[0];
                S[1] += Sb[1];
                ISTAMP(2, S[0][0] + S[1][0] + Nd[0][0] + Nd[1][0]);
            }

            if (!(xmode & 2)) VLSA_BAR();  // readers of the previous exchange are done
            if (have) ISTAMP(3, 0.f);
            {
                unsigned char* mine = exch + cw * kExchWave;
                *reinterpret_cast<f32x4_ma*>(mine + (0 * 64 + lane) * 16) = S[0];
                *reinterpret_cast<f32x4_ma*>(mine + (1 * 64 + lane) * 16) = S[1];
                if (g == (i16 >> 2)) {
                    const int r = i16 & 3;
                    const float d0 = r == 0 ? Nd[0][0] : r == 1 ? Nd[0][1] : r == 2 ? Nd[0][2] : Nd[0][3];
                    const float d1 = r == 0 ? Nd[1][0] : r == 1 ? Nd[1][1] : r == 2 ? Nd[1][2] : Nd[1][3];
                    reinterpret_cast<float_ma*>(mine + 2048)[i16] = d0;
                    reinterpret_cast<float_ma*>(mine + 2048)[16 + i16] = d1;
                }
            }
            if (!(xmode & 2)) VLSA_BAR();
            if (have) {
                ISTAMP(4, 0.f);
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
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[h][r] *= fminf(__builtin_amdgcn_rsqf(R2[h][r]), 1e12f);
                if (row0 + kTile > nrows) {  // ragged last tile of this workgroup's range
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row0 + 16 * h + 4 * g + r >= nrows) T[h][r] = -INFINITY;
                }
                if constexpr (kScores) {  // the four column-quarter waves hold identical scores: waves cw = 0 / 1 store half h = cw
                    if (has_scores) {
                        // value of (query q, rows 16 h + 4 gs ..) lives in lane 16 gs + q, register T[h]
                        const int c = lane & 7;
                        const int src = (16 * (c & 3) + ((8 * cw + (lane >> 3)) & 15)) << 2;
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float t0 = T[0][r], t1 = T[1][r];   // (no __builtin_bit_cast on the vector element itself: hipcc 7.0 then reads element 0 for every r)
                            const float a0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(t0)));
                            const float a1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(t1)));
                            v[r] = (c & 4) ? a1 : a0;
                        }
                        if (srow != nullptr) *reinterpret_cast<f32x4*>(srow + row0) = v;
                    }
                }
                ISTAMP(5, T[0][0] + T[1][0]);
                const float tmax = fmaxf(fmaxf(fmaxf(T[0][0], T[0][1]), fmaxf(T[0][2], T[0][3])),
                                         fmaxf(fmaxf(T[1][0], T[1][1]), fmaxf(T[1][2], T[1][3])));
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
                bf16x8 ahi, alo;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float wv = (xmode & 4) ? T[h][r] : fast_exp2(T[h][r] - M);
                        lsum += wv;
                        const __bf16 hi = (__bf16)wv;
                        ahi[4 * h + r] = hi;
                        alo[4 * h + r] = (__bf16)(wv - (float)hi);
                    }
                ISTAMP(6, lsum);
                if (!(xmode & 8))
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    const int c_off = ct * 32 + (i16 & 3) * 8;
                    const int rr = 4 * g + (i16 >> 2);
                    const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + bswz(rr, c_off)));
                    const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + bswz(16 + rr, c_off)));
                    const bf16x8 bh = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bh, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bh, acc[ct], 0, 0, 0);
                }
                ISTAMP(7, acc[0][0] + acc[7][0]);
                ++kown;
            }
            BSTAMP(stamp++);
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
        VLSA_BAR();
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
        VLSA_BAR();  // lent slots and the transpose area are free again
        BSTAMP(stamp++);
    }
}

// ---- batched merge: grid (D/64, P, B); explicit strides (in floats) so it serves both the local fold of workgroup
// partials and the fold of all-gathered per-rank records (see vlsa_vlfan_merge_batch_strided in vlsa_hip.h) ----------
struct MergeStrides {
    int64_t sm, sl, sa;  // between consecutive partials of one bag
    int64_t bm, bl, ba;  // between bags, inputs
    int64_t om, ol, oo;  // between bags, outputs
};

__global__ __launch_bounds__(256, 5) void k_vlfan_merge_batch(const float* __restrict__ pm, const float* __restrict__ pl,
                                                            const float* __restrict__ pacc, int G, int P, int D,
                                                            int normalise, float* __restrict__ m2, float* __restrict__ l,
                                                            float* __restrict__ out, MergeStrides st) {
    // workgroup (cchunk, p, bag): 128 threads x float4 = 512 columns of query p, 2 partial subsets (even / odd g);
    // every thread keeps up to 16 accumulator pieces in flight per pass.
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float4 sacc[128];
    __shared__ float sl;
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);  // short kernel that may co-run with a persistent streaming kernel: win issue arbitration
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = blockIdx.y, c0 = blockIdx.x * 512, bag = blockIdx.z;
    pm += (size_t)bag * st.bm;
    pl += (size_t)bag * st.bl;
    pacc += (size_t)bag * st.ba;
    const int c4 = tid & 127, gs = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform: partial indices stay scalar
    const int col = c0 + c4 * 4;
    const bool incol = col < D;
    constexpr int U = 16;  // <= 96 VGPRs so the kernel can co-reside with a persistent streaming kernel (2 x 208 of 512 VGPRs per SIMD taken)
    float mx = -INFINITY;
    for (int gI = tid; gI < G; gI += 256) mx = fmaxf(mx, pm[(size_t)gI * st.sm + p]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float lt = 0.f;
    for (int g0 = gs; g0 < G; g0 += 2 * U) {
        float mg[U], lg[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int gI = g0 + 2 * u;
            const bool ok = gI < G;
            mg[u] = ok ? pm[(size_t)gI * st.sm + p] : -INFINITY;
            lg[u] = ok ? pl[(size_t)gI * st.sl + p] : 0.f;
            v[u] = (ok && incol) ? *reinterpret_cast<const float4*>(pacc + (size_t)gI * st.sa + (size_t)p * D + col)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float f = (mg[u] == -INFINITY) ? 0.f : fast_exp2(mg[u] - mx);
            lt += lg[u] * f;
            a.x += v[u].x * f; a.y += v[u].y * f; a.z += v[u].z * f; a.w += v[u].w * f;
        }
    }
    if (gs == 1) {
        sacc[c4] = a;
        if (c4 == 0) sl = lt;
    }
    __syncthreads();
    if (gs == 0) {
        const float4 o = sacc[c4];
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        const float ls = lt + sl;
        if (normalise) {
            const float inv = 1.f / ls;
            a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        }
        if (incol) *reinterpret_cast<float4*>(out + (size_t)bag * st.oo + (size_t)p * D + col) = a;
        if (c4 == 0 && blockIdx.x == 0) {
            m2[(size_t)bag * st.om + p] = mx;
            l[(size_t)bag * st.ol + p] = ls;
        }
    }
}

// Merge + query pooling in one pass (the batched forward's tail): workgroup (cc, bag) owns 64 columns of ALL P queries of
// a bag -- thread (gs = tid >> 8, p = (tid >> 4) & 15, c4 = tid & 15) folds every second partial piece of query p, float4
// column c4 -- so the pooled vector (mean / max / softmax(weight) over the queries, model/deepmil.py:133-150) is formed
// right here from LDS instead of every head workgroup re-reading the P x D rows.  The p order of the pooling sum is fixed.
// <= 96 VGPRs: may co-reside with a persistent streaming kernel of another stream.
__global__ __launch_bounds__(512, 4) void k_vlfan_merge_pool_batch(const float* __restrict__ pm, const float* __restrict__ pl,
                                                                    const float* __restrict__ pacc, int G, int P, int D,
                                                                    float* __restrict__ m2, float* __restrict__ l,
                                                                    float* __restrict__ out, MergeStrides st, int pool_mode,
                                                                    const float* __restrict__ pool_w,
                                                                    float* __restrict__ pooled) {
    __shared__ __attribute__((aligned(16))) float4 sacc[VLSA_MAX_P][16];
    __shared__ float slt[VLSA_MAX_P][16];
    __shared__ float spw[VLSA_MAX_P];
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int tid = threadIdx.x;
    const int gs = tid >> 8, p = (tid >> 4) & 15, c4 = tid & 15, bag = blockIdx.y;
    const int col = blockIdx.x * 64 + c4 * 4;
    const bool live = p < P && col < D;
    pm += (size_t)bag * st.bm + p;
    pl += (size_t)bag * st.bl + p;
    pacc += (size_t)bag * st.ba + (size_t)p * D + col;
    if (pool_mode == VLSA_POOL_WEIGHT && tid == 0) {
        float mx = -INFINITY, sum = 0.f;
        for (int q = 0; q < P; ++q) mx = fmaxf(mx, pool_w[q]);
        for (int q = 0; q < P; ++q) { spw[q] = expf(pool_w[q] - mx); sum += spw[q]; }
        for (int q = 0; q < P; ++q) spw[q] /= sum;
    }
    constexpr int U = 8;
    float mx = -INFINITY;
    if (p < P)
        for (int g = 0; g < G; ++g) mx = fmaxf(mx, pm[(size_t)g * st.sm]);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float lt = 0.f;
    for (int g0 = gs; g0 < G; g0 += 2 * U) {
        float mg[U], lg[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = g0 + 2 * u;
            const bool ok = live && g < G;
            mg[u] = ok ? pm[(size_t)g * st.sm] : -INFINITY;
            lg[u] = ok ? pl[(size_t)g * st.sl] : 0.f;
            v[u] = ok ? *reinterpret_cast<const float4*>(pacc + (size_t)g * st.sa) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float f = (mg[u] == -INFINITY) ? 0.f : fast_exp2(mg[u] - mx);
            lt += lg[u] * f;
            a.x += v[u].x * f; a.y += v[u].y * f; a.z += v[u].z * f; a.w += v[u].w * f;
        }
    }
    if (gs == 1) {
        sacc[p][c4] = a;
        slt[p][c4] = lt;
    }
    __syncthreads();
    if (gs == 0) {
        const float4 o = sacc[p][c4];
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        lt += slt[p][c4];
        if (live) {
            const float inv = 1.f / lt;
            a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
            *reinterpret_cast<float4*>(out + (size_t)bag * st.oo + (size_t)p * D + col) = a;
            if (blockIdx.x == 0 && c4 == 0) {
                m2[(size_t)bag * st.om + p] = mx;
                l[(size_t)bag * st.ol + p] = lt;
            }
        }
    }
    if (pooled == nullptr) return;
    __syncthreads();
    if (gs == 0) sacc[p][c4] = a;
    __syncthreads();
    if (tid < 16 && col < D) {
        float4 r;
        if (pool_mode == VLSA_POOL_MAX) {
            r = sacc[0][tid];
            for (int q = 1; q < P; ++q) {
                const float4 o = sacc[q][tid];
                r.x = fmaxf(r.x, o.x); r.y = fmaxf(r.y, o.y); r.z = fmaxf(r.z, o.z); r.w = fmaxf(r.w, o.w);
            }
        } else if (pool_mode == VLSA_POOL_WEIGHT) {
            r = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < P; ++q) {
                const float4 o = sacc[q][tid];
                const float wq = spw[q];
                r.x += wq * o.x; r.y += wq * o.y; r.z += wq * o.z; r.w += wq * o.w;
            }
        } else {  // mean: left-to-right sum, then / P (as pooled_col in vlfan_tail.hip)
            r = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = 0; q < P; ++q) {
                const float4 o = sacc[q][tid];
                r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            const float fp = (float)P;
            r.x /= fp; r.y /= fp; r.z /= fp; r.w /= fp;
        }
        *reinterpret_cast<float4*>(pooled + (size_t)bag * D + col) = r;
    }
}

// Merge + query pooling for FEW partials per bag (G <= 8: wide launches of slide-sized bags run one workgroup per bag, G = 1;
// 64 x 50k bags 8): ONE workgroup per bag, thread = column.  The (m, l) of all G x P partial records go through 1 KB of LDS,
// then every accumulator piece a thread needs (<= 16 queries x 2 partials per round) is in flight at once.  At B = 256, G = 1
// k_vlfan_merge_pool_batch took 13.6 us (2 048 workgroups of 512 threads, a serial pooling loop each); this one 256 workgroups.
// <= 8 KiB LDS, <= 96 VGPRs AND four waves (256 threads, two columns each): co-resides with a persistent streaming kernel of another
// stream -- that kernel's two waves per SIMD leave 96 registers per lane of a SIMD, i.e. room for ONE 88-register wave: as an 8-wave
// workgroup (round 5's first version) this kernel waited for the NEXT launch's streaming kernel to end (kernel trace: it "ran" 154 us),
// and the tails of two launches were paid behind every pair of streaming kernels.
__global__ __launch_bounds__(256) void k_vlfan_merge_pool_small(const float* __restrict__ pm, const float* __restrict__ pl,
                                                                    const float* __restrict__ pacc, int G, int P, int D,
                                                                    float* __restrict__ m2, float* __restrict__ l,
                                                                    float* __restrict__ out, MergeStrides st, int pool_mode,
                                                                    const float* __restrict__ pool_w, float* __restrict__ pooled) {
    __shared__ float sm_[8][VLSA_MAX_P], sl_[8][VLSA_MAX_P];   // per partial: exp2(m_g - m) and l_g
    __shared__ float sinv[VLSA_MAX_P], spw[VLSA_MAX_P];
    __builtin_amdgcn_s_setprio(VLSA_TAIL_PRIO);
    const int tid = threadIdx.x, bag = blockIdx.x;
    pm += (size_t)bag * st.bm;
    pl += (size_t)bag * st.bl;
    pacc += (size_t)bag * st.ba;
    if (tid < 8 * VLSA_MAX_P) {
        const int g = tid >> 4, p = tid & 15;
        const bool ok = g < G && p < P;
        sm_[g][p] = ok ? pm[(size_t)g * st.sm + p] : -INFINITY;
        sl_[g][p] = ok ? pl[(size_t)g * st.sl + p] : 0.f;
    }
    if (pool_mode == VLSA_POOL_WEIGHT && tid == 255) {
        float mx = -INFINITY, sum = 0.f;
        for (int q = 0; q < P; ++q) mx = fmaxf(mx, pool_w[q]);
        for (int q = 0; q < P; ++q) { spw[q] = expf(pool_w[q] - mx); sum += spw[q]; }
        for (int q = 0; q < P; ++q) spw[q] /= sum;
    }
    __syncthreads();
    if (tid < VLSA_MAX_P) {
        const int p = tid;
        float mx = -INFINITY;
#pragma unroll
        for (int g = 0; g < 8; ++g) mx = fmaxf(mx, sm_[g][p]);
        float ls = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float f = sm_[g][p] == -INFINITY ? 0.f : fast_exp2(sm_[g][p] - mx);
            ls += sl_[g][p] * f;
            sm_[g][p] = f;
        }
        sinv[p] = 1.f / ls;
        if (p < P) {
            m2[(size_t)bag * st.om + p] = mx;
            l[(size_t)bag * st.ol + p] = ls;
        }
    }
    __syncthreads();
    for (int c = tid; c < D; c += 256) {
        float acc[VLSA_MAX_P];
#pragma unroll
        for (int p = 0; p < VLSA_MAX_P; ++p) acc[p] = 0.f;
        for (int g = 0; g < G; ++g) {       // one partial record per round: its <= 16 pieces in flight together
            float x[VLSA_MAX_P];
#pragma unroll
            for (int p = 0; p < VLSA_MAX_P; ++p) x[p] = p < P ? pacc[(size_t)g * st.sa + (size_t)p * D + c] : 0.f;
#pragma unroll
            for (int p = 0; p < VLSA_MAX_P; ++p) acc[p] += x[p] * sm_[g][p];
        }
        float r = pool_mode == VLSA_POOL_MAX ? -INFINITY : 0.f;
#pragma unroll
        for (int p = 0; p < VLSA_MAX_P; ++p)
            if (p < P) {
                const float o = acc[p] * sinv[p];
                out[(size_t)bag * st.oo + (size_t)p * D + c] = o;
                if (pool_mode == VLSA_POOL_MAX) r = fmaxf(r, o);
                else if (pool_mode == VLSA_POOL_WEIGHT) r += spw[p] * o;
                else r += o;
            }
        if (pool_mode == VLSA_POOL_MEAN) r /= (float)P;
        if (pooled != nullptr) pooled[(size_t)bag * D + c] = r;
    }
}

// A[p, n] = exp2(t[p, n] - m2[bag, p]) / l[bag, p] for every bag of a batch (softmax over the patches, model/deepmil.py:198),
// from the scores the streaming kernel stored and the bag-global (m2, l) of the merge.  grid (chunks of 1024 patches, P, B);
// float4 per thread; may run in place (A == scores).  Columns N .. ld-1 of a row hold -inf scores -> 0.
__global__ __launch_bounds__(256) void k_attn_normalise_batch(const BagDesc* __restrict__ bags, const RowsDesc* __restrict__ sdesc,
                                                             const RowsDesc* __restrict__ adesc, const float* __restrict__ m2,
                                                             const float* __restrict__ l, int m_stride) {
    const int bag = blockIdx.z, p = blockIdx.y;
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t N = bags[bag].N;
    const RowsDesc sd = sdesc[bag], ad = adesc[bag];
    if (n >= N || sd.ptr == nullptr || ad.ptr == nullptr) return;
    const float m = m2[(size_t)bag * m_stride + p], inv = 1.f / l[(size_t)bag * m_stride + p];
    const f32x4 t = *reinterpret_cast<const f32x4*>(sd.ptr + (size_t)p * sd.ld + n);
    f32x4 a;
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = fast_exp2(t[r] - m) * inv;
    *reinterpret_cast<f32x4*>(ad.ptr + (size_t)p * ad.ld + n) = a;
}

}  // namespace vlsa

using namespace vlsa;

int vlsa_launch_head_batch(const float* rows, int B, int P, int D, int pool_mode, const float* pool_w, const float* W,
                           const float* b, const float* That, int K, const float* logit_scale, unsigned int* counters,
                           float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence,
                           hipStream_t s);  // vlfan_tail.hip
int vlsa_launch_head_pooled_batch(const float* pooled, int B, int D, const float* W, const float* b, const float* That, int K,
                                  const float* logit_scale, float* v, float* vhat, float* vnorm, float* logits,
                                  float* incidence, hipStream_t s);  // vlfan_tail.hip
int vlsa_launch_head_rows_batch(const float* rows, int B, int P, int D, int pool_mode, const float* pool_w, const float* W,
                                const float* b, const float* That, int K, const float* logit_scale, float* pooled, float* v,
                                float* vhat, float* vnorm, float* logits, float* incidence, hipStream_t s);  // vlfan_tail.hip

// merge + pooling of B bags: few partials per bag -> one workgroup per bag, else the (64 columns, bag) workgroups
static int launch_merge_pool(const float* pm, const float* pl, const float* pacc, int B, int G, int P, int D, float* m2, float* l,
                             float* out, const MergeStrides& st, int pool_mode, const float* pool_w, float* pooled, hipStream_t s) {
    if (G <= 8)
        hipLaunchKernelGGL(k_vlfan_merge_pool_small, dim3(B), dim3(256), 0, s, pm, pl, pacc, G, P, D, m2, l, out, st, pool_mode, pool_w,
                           pooled);
    else
        hipLaunchKernelGGL(k_vlfan_merge_pool_batch, dim3((D + 63) / 64, B), dim3(512), 0, s, pm, pl, pacc, G, P, D, m2, l, out, st,
                           pool_mode, pool_w, pooled);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

#ifdef VLSA_TIMING
extern "C" int vlsa_debug_read_batch_cycles(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(vlsa::vlsa_dbg_batch), sizeof(long long) * 64) == hipSuccess ? 0 : -3;
}
#endif

// bags streamed concurrently (each by 256 / S workgroups)
static inline int batch_groups(int B) { return B >= 8 ? 8 : (B >= 4 ? 4 : (B >= 2 ? 2 : 1)); }
// S = bags streamed concurrently (each by workgroups / S workgroups).  Auto: 8 for B >= 8.  Small bags want more: every
// workgroup should see >= ~8 lock-step iterations (512 rows) of a bag per bag epilogue, so the host may ask for S up to 64.
// More than 64 bags per launch (round 4, up to kMaxBags = 256): a workgroup's LDS table holds its own <= 64 bags, so S >= B / 64; that
// also bounds the partial records of a launch (B x 256 / S <= 16 384, the workspace of a 64-bag launch).
static inline int min_groups(int B) {
    int m = 1;
    while (m * bt::kMaxLocal < B) m *= 2;
    return m;
}
static inline int resolve_groups(int B, int groups) {
    int S = groups > 0 ? groups : batch_groups(B);
    int p2 = 1;
    while (p2 * 2 <= S && p2 * 2 <= B && p2 * 2 <= 256) p2 *= 2;  // power of two, <= B, <= 256
    const int m = min_groups(B);
    return p2 < m ? m : p2;
}
// workgroups of the persistent kernels when `reserved_cus` CUs are to stay free for concurrently running tail /
// communication kernels (next to a persistent workgroup only kernels with <= 96 VGPRs / 8 KiB LDS get scheduled)
static inline int batch_workgroups(int S, int reserved_cus) {
    int r = reserved_cus < 0 ? 0 : reserved_cus;
    r = (r + S - 1) / S * S;
    if (r > 256 - S) r = 256 - S;
    return 256 - r;
}
extern "C" int vlsa_batch_partials_per_bag(int B) { return 256 / batch_groups(B); }
extern "C" int vlsa_batch_partials_per_bag_ex(int B, int reserved_cus, int groups) {
    const int S = resolve_groups(B, groups);
    return batch_workgroups(S, reserved_cus) / S;
}
extern "C" int vlsa_batch_groups(const int64_t* rows_host, int B, int reserved_cus) {
    // Pick the number of bags in flight from the bag sizes (HOST array): bag t goes to group t % S and is streamed by
    // Gb = workgroups / S workgroups in 64-row lock-step iterations; a bag epilogue costs about 0.7 iterations.  The launch
    // lasts as long as its slowest group: minimise that.  (Equal 50k bags, B = 32: S = 32 -> 98.7, S = 8 -> 102.8.)
    if (!rows_host || B < 1) return 1;
    int best = 1;
    double best_cost = 1e300;
    for (int S = min_groups(B); S <= 256 && S <= B; S *= 2) {
        const int Gb = batch_workgroups(S, reserved_cus) / S;
        double worst = 0.0;
        for (int g = 0; g < S; ++g) {
            double c = 0.0;
            for (int t = g; t < B; t += S) {
                const int64_t units = (rows_host[t] + 63) / 64;
                c += (double)((units + Gb - 1) / Gb) + 0.7;
            }
            if (c > worst) worst = c;
        }
        if (worst < best_cost * 0.999) {  // ties go to the smaller S (fewer, larger partial merges are not needed)
            best_cost = worst;
            best = S;
        }
    }
    return best;
}

extern "C" int vlsa_batch_max_bags(void) { return 64; }                       // every batched entry point (forward, backward, scores)
extern "C" int vlsa_batch_forward_max_bags(void) { return bt::kMaxBags; }     // the forward streaming launches + their tails

extern "C" size_t vlsa_batch_workspace_bytes(int B, int P, int D) {
    const size_t recs = (size_t)(B < bt::kMaxLocal ? B : bt::kMaxLocal) * 256;   // B x partials per bag, see min_groups
    return (recs * kPStride * 2 + recs * P * D) * sizeof(float) + (size_t)B * 64;
}

int vlsa_launch_partial_f32_batch(const void* bag_desc, int B, const float* qeff, const float* qmeta, int P, float* pm,
                                  float* pl, float* pacc, int S, int workgroups, const void* scores_desc, hipStream_t s);  // vlfan_batch_f32.hip

extern "C" int vlsa_vlfan_partial_batch_scores(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P,
                                               void* workspace, int reserved_cus, int groups, const void* scores_desc,
                                               void* stream) {
    if (!bag_desc || !qprep || !workspace) return VLSA_EINVAL;
    if (B < 1 || B > bt::kMaxBags || P < 1 || P > VLSA_MAX_P) return VLSA_EINVAL;
    if (x_dtype != VLSA_DT_BF16 && x_dtype != VLSA_DT_F32) return VLSA_EINVAL;
    if (D != 512) return VLSA_EUNSUPPORTED;
    const int S = resolve_groups(B, groups);
    const int WG = batch_workgroups(S, reserved_cus);
    const int G = WG / S;  // partials per bag
    float* pm = static_cast<float*>(workspace);
    float* pl = pm + (size_t)B * G * kPStride;
    float* pacc = pl + (size_t)B * G * kPStride;
    static DeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)k_vlfan_partial_dma_batch<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bt::kLdsBytes);
        (void)hipFuncSetAttribute((const void*)k_vlfan_partial_dma_batch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bt::kLdsBytes);
    }
    const QPrepLayout L(D);
    if (x_dtype == VLSA_DT_F32) {
        const unsigned char* qp = static_cast<const unsigned char*>(qprep);
        return vlsa_launch_partial_f32_batch(bag_desc, B, reinterpret_cast<const float*>(qp + L.qeff),
                                             reinterpret_cast<const float*>(qp + L.qnorm), P, pm, pl, pacc, S, WG, scores_desc,
                                             (hipStream_t)stream);
    }
    const __bf16* qsplit = reinterpret_cast<const __bf16*>(static_cast<const unsigned char*>(qprep) + L.qsplit);
#ifdef VLSA_TIMING
    static const int xm = VLSA_ENV("VLSA_EXP") ? atoi(VLSA_ENV("VLSA_EXP")) : 0;
#else
    constexpr int xm = 0;
#endif
    if (scores_desc)
        hipLaunchKernelGGL(k_vlfan_partial_dma_batch<true>, dim3(WG), dim3(512), bt::kLdsBytes, (hipStream_t)stream,
                           static_cast<const BagDesc*>(bag_desc), B, qsplit, P, pm, pl, pacc, S,
                           static_cast<const RowsDesc*>(scores_desc));
    else
        hipLaunchKernelGGL(k_vlfan_partial_dma_batch<false>, dim3(WG), dim3(512), bt::kLdsBytes, (hipStream_t)stream,
                           static_cast<const BagDesc*>(bag_desc), B, qsplit, P, pm, pl, pacc, S | (xm << 8),
                           static_cast<const RowsDesc*>(nullptr));
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

extern "C" int vlsa_vlfan_partial_batch_ex(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P,
                                           void* workspace, int reserved_cus, int groups, void* stream) {
    return vlsa_vlfan_partial_batch_scores(bag_desc, B, x_dtype, D, qprep, P, workspace, reserved_cus, groups, nullptr, stream);
}

extern "C" int vlsa_attn_normalise_batch(const void* bag_desc, int B, int P, int64_t max_N, const void* scores_desc,
                                         const float* m2, const float* l, const void* attn_desc, void* stream) {
    if (!bag_desc || !scores_desc || !attn_desc || !m2 || !l) return VLSA_EINVAL;
    if (B < 1 || B > bt::kMaxBags || P < 1 || P > VLSA_MAX_P || max_N < 0) return VLSA_EINVAL;
    if (max_N == 0) return VLSA_OK;
    const unsigned int chunks = (unsigned int)((max_N + 1023) / 1024);
    hipLaunchKernelGGL(k_attn_normalise_batch, dim3(chunks, P, B), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const BagDesc*>(bag_desc), static_cast<const RowsDesc*>(scores_desc),
                       static_cast<const RowsDesc*>(attn_desc), m2, l, (int)kPStride);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

extern "C" int vlsa_vlfan_partial_batch(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P,
                                        void* workspace, void* stream) {
    return vlsa_vlfan_partial_batch_ex(bag_desc, B, x_dtype, D, qprep, P, workspace, 0, 0, stream);
}

extern "C" int vlsa_vlfan_forward_batch_attn(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P,
                                             int pool_mode, const float* pool_w, const float* W, const float* b,
                                             const float* That, int K, const float* logit_scale, void* workspace, float* m2,
                                             float* l, float* out, float* pooled, float* v, float* vhat, float* vnorm,
                                             float* logits, float* incidence, int reserved_cus, int groups,
                                             const void* scores_desc, const void* attn_desc, int64_t max_N, void* stream) {
    if (!That || !logit_scale || !m2 || !l || !out || !pooled || !v || !vhat || !vnorm || !logits) return VLSA_EINVAL;
    if (K < 1 || K > VLSA_MAX_K) return VLSA_EINVAL;
    if ((scores_desc == nullptr) != (attn_desc == nullptr)) return VLSA_EINVAL;
    if (pool_mode < VLSA_POOL_MEAN || pool_mode > VLSA_POOL_WEIGHT) return VLSA_EINVAL;
    if (pool_mode == VLSA_POOL_WEIGHT && !pool_w) return VLSA_EINVAL;
    const int rc = vlsa_vlfan_partial_batch_scores(bag_desc, B, x_dtype, D, qprep, P, workspace, reserved_cus, groups, scores_desc,
                                                   stream);
    if (rc != VLSA_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int G = vlsa_batch_partials_per_bag_ex(B, reserved_cus, groups);
    float* pm = static_cast<float*>(workspace);
    float* pl = pm + (size_t)B * G * kPStride;
    float* pacc = pl + (size_t)B * G * kPStride;
    unsigned int* counters = reinterpret_cast<unsigned int*>(static_cast<unsigned char*>(workspace) +
                                                             vlsa_batch_workspace_bytes(B, P, D) - (size_t)B * 64);
    const MergeStrides st{kPStride, kPStride, (int64_t)P * D, (int64_t)G * kPStride, (int64_t)G * kPStride,
                          (int64_t)G * P * D, kPStride, kPStride, (int64_t)P * D};
    // merge + pooling in one kernel; the head then starts from the pooled vectors (2 KB instead of P x 2 KB per workgroup)
    if (launch_merge_pool(pm, pl, pacc, B, G, P, D, m2, l, out, st, pool_mode, pool_w, pooled, s) != VLSA_OK) return VLSA_ELAUNCH;
    (void)counters;
    const int rh = vlsa_launch_head_pooled_batch(pooled, B, D, W, b, That, K, logit_scale, v, vhat, vnorm, logits, incidence, s);
    if (rh != VLSA_OK || !scores_desc) return rh;
    return vlsa_attn_normalise_batch(bag_desc, B, P, max_N, scores_desc, m2, l, attn_desc, stream);
}

extern "C" int vlsa_vlfan_forward_batch(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P,
                                        int pool_mode, const float* pool_w, const float* W, const float* b,
                                        const float* That, int K, const float* logit_scale, void* workspace, float* m2,
                                        float* l, float* out, float* pooled, float* v, float* vhat, float* vnorm,
                                        float* logits, float* incidence, int reserved_cus, int groups, void* stream) {
    return vlsa_vlfan_forward_batch_attn(bag_desc, B, x_dtype, D, qprep, P, pool_mode, pool_w, W, b, That, K, logit_scale,
                                         workspace, m2, l, out, pooled, v, vhat, vnorm, logits, incidence, reserved_cus, groups,
                                         nullptr, nullptr, 0, stream);
}

extern "C" int vlsa_vlfan_merge_batch_strided(const float* pm, const float* pl, const float* pacc, int B, int G, int P, int D,
                                              int normalise, const int64_t* strides9, float* m2, float* l, float* out,
                                              void* stream) {
    if (!pm || !pl || !pacc || !strides9 || !m2 || !l || !out) return VLSA_EINVAL;
    if (B < 1 || G < 1 || P < 1 || P > VLSA_MAX_P || D < 4 || (D % 4) != 0 || D > VLSA_MAX_D) return VLSA_EINVAL;
    MergeStrides st{strides9[0], strides9[1], strides9[2], strides9[3], strides9[4], strides9[5], strides9[6], strides9[7], strides9[8]};
    if ((st.sa % 4) || (st.ba % 4) || (st.oo % 4) || (reinterpret_cast<uintptr_t>(pacc) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return VLSA_EINVAL;
    hipLaunchKernelGGL(k_vlfan_merge_batch, dim3((D + 511) / 512, P, B), dim3(256), 0, (hipStream_t)stream, pm, pl, pacc, G, P, D,
                       normalise, m2, l, out, st);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

extern "C" int vlsa_vlfan_merge_head_batch_strided(const float* pm, const float* pl, const float* pacc, int B, int G, int P, int D,
                                                   const int64_t* strides9, int pool_mode, const float* pool_w, const float* W,
                                                   const float* b, const float* That, int K, const float* logit_scale, float* m2,
                                                   float* l, float* out, float* pooled, float* v, float* vhat, float* vnorm,
                                                   float* logits, float* incidence, void* stream) {
    if (!pm || !pl || !pacc || !strides9 || !m2 || !l || !out || !pooled || !v || !vhat || !vnorm || !logits || !That || !logit_scale)
        return VLSA_EINVAL;
    if (B < 1 || G < 1 || P < 1 || P > VLSA_MAX_P || D < 4 || (D % 4) != 0 || D > VLSA_MAX_D || K < 1 || K > VLSA_MAX_K) return VLSA_EINVAL;
    if (pool_mode < VLSA_POOL_MEAN || pool_mode > VLSA_POOL_WEIGHT || (pool_mode == VLSA_POOL_WEIGHT && !pool_w)) return VLSA_EINVAL;
    MergeStrides st{strides9[0], strides9[1], strides9[2], strides9[3], strides9[4], strides9[5], strides9[6], strides9[7], strides9[8]};
    if ((st.sa % 4) || (st.ba % 4) || (st.oo % 4) || (reinterpret_cast<uintptr_t>(pacc) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return VLSA_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (launch_merge_pool(pm, pl, pacc, B, G, P, D, m2, l, out, st, pool_mode, pool_w, pooled, s) != VLSA_OK) return VLSA_ELAUNCH;
    return vlsa_launch_head_pooled_batch(pooled, B, D, W, b, That, K, logit_scale, v, vhat, vnorm, logits, incidence, s);
}

extern "C" int vlsa_head_forward_batch(const float* rows, int B, int P, int D, int pool_mode, const float* pool_w, const float* W,
                                       const float* b, const float* That, int K, const float* logit_scale, void* counters,
                                       float* pooled, float* v, float* vhat, float* vnorm, float* logits, float* incidence,
                                       void* stream) {
    if (!rows || !That || !logit_scale || !counters || !pooled || !v || !vhat || !vnorm || !logits) return VLSA_EINVAL;
    if (B < 1 || P < 1 || P > VLSA_MAX_P || K < 1 || K > VLSA_MAX_K || D <= 0 || D > VLSA_MAX_D || (D % 4) != 0) return VLSA_EINVAL;
    // B > 1, pooling over the P rows: pool once per bag, then the ticket-free pooled route (16 instead of 34 us for 32 bags); the
    // tickets are not touched (they stay zeroed, as the ticketed kernel hands them back)
    if (B > 1 && pool_mode >= VLSA_POOL_MEAN && pool_mode <= VLSA_POOL_WEIGHT && pooled != rows)
        return vlsa_launch_head_rows_batch(rows, B, P, D, pool_mode, pool_w, W, b, That, K, logit_scale, pooled, v, vhat, vnorm, logits,
                                           incidence, (hipStream_t)stream);
    return vlsa_launch_head_batch(rows, B, P, D, pool_mode, pool_w, W, b, That, K, logit_scale, static_cast<unsigned int*>(counters),
                                  pooled, v, vhat, vnorm, logits, incidence, (hipStream_t)stream);
}

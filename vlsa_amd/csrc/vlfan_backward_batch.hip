// Backward of the aggregation for a BATCH of bags, w.r.t. the (shared) effective queries -- the training step of the
// reference back-propagates 32 bags at once (runner/vlsa_handler.py:260-289).  One persistent launch with the structure
// of k_vlfan_partial_dma_batch (vlfan_batch.hip) and the math of k_vlfan_backward_mfma (vlfan_backward.hip):
//     de += scale * sum_bags sum_n A_pn (dout_p . x_n - delta_p) x_n / max(|x_n|, eps)
// Because the queries are shared, the per-bag contributions simply add: the accumulators live in registers across ALL
// bags and every workgroup writes ONE partial at the very end (no per-bag epilogue at all).  Per bag only the upstream
// gradient fragments (3-term bf16 split of dout), m2, 1/l and delta are reloaded.
// Workgroup = FOUR waves = the four column quarters of one 32-row tile, 512 workgroups = two per CU.  (Round 1 ran eight
// waves = two row groups per workgroup: the workgroup-wide barriers of the exchange kept the two waves of every SIMD in lock
// step -- both issue their MFMAs at the same time, then idle together.  Two independent workgroups drift apart and one's
// exchange / softmax phase overlaps the other's MFMA phase: 309 -> 290 us per 32 x 50k bags on the same box.)
// The two partial tiles a wave must share per 32-row tile (scores and dout . x) are exchanged in a compact
// [2 h][4 g][12 p] layout so that a workgroup stays inside 80 KiB of LDS, hence P <= 12 here (the reference's datasets use
// 7..12 prototypes); larger P goes through the per-bag kernel.
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

struct BagDesc {
    const void* X;
    int64_t N;
    int64_t ldx;
};

namespace bb {
constexpr int kTile = 32;
constexpr int kSlot = kTile * 256;
constexpr int kWaveRing = 2 * kSlot;
constexpr int kRingBytes = 4 * kWaveRing;         // 64 KiB: four waves
constexpr int kMaxP = 12;
constexpr int kTileBytes = 2 * 4 * kMaxP * 16;    // one compact [2 h][4 g][12 p] x f32x4 tile = 1536 B
constexpr int kExchWave = 2 * kTileBytes + 128;   // S tile + dA tile + 32 row sums of squares = 3200 B
constexpr int kExchGroup = 4 * kExchWave;
constexpr int kTabOff = kRingBytes + kExchGroup;
constexpr int kMaxBags = 64;
constexpr int kLdsBytes = kTabOff + kMaxBags * 32;  // 80,384 B: two workgroups per CU
}  // namespace bb

__device__ __forceinline__ int wswz(int row, int byte_off) { return row * 256 + (byte_off ^ ((row & 7) << 5)); }

#define VLSA_WBAR()                                          \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                        \
        asm volatile("" ::: "memory");                       \
    } while (0)

// dsplit[bag][t][p][:] = 3-term bf16 split of dout[bag][p][:]; delta[bag][p] = dout . out     grid (16, B)
__global__ __launch_bounds__(256) void k_prepare_backward_batch(const float* __restrict__ dout, const float* __restrict__ out,
                                                                 int P, int D, __bf16* __restrict__ dsplit,
                                                                 float* __restrict__ delta) {
    __shared__ float red[4];
    const int p = blockIdx.x, bag = blockIdx.y, tid = threadIdx.x;
    dout += (size_t)bag * P * D;
    out += (size_t)bag * P * D;
    dsplit += (size_t)bag * 3 * 16 * D;
    float acc = 0.f;
    for (int d = tid; d < D; d += 256) {
        const float x = p < P ? dout[(size_t)p * D + d] : 0.f;
        if (p < P) acc += x * out[(size_t)p * D + d];
        const __bf16 h0 = (__bf16)x;
        const float r1 = x - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        dsplit[((size_t)0 * 16 + p) * D + d] = h0;
        dsplit[((size_t)1 * 16 + p) * D + d] = h1;
        dsplit[((size_t)2 * 16 + p) * D + d] = (__bf16)(r1 - (float)h1);
    }
    acc = block_sum_256(acc, red);
    if (tid == 0) delta[(size_t)bag * kPStride + p] = acc;
}

// S = number of workgroup groups: bag t is streamed by the Gb = G / S workgroups of group t % S only, so S bags are in
// flight at once, every workgroup sees S times more rows per bag (fewer bag epilogues, better tile quantisation) and a
// bag leaves Gb instead of G partials behind.
__global__ __launch_bounds__(256, 2) void k_vlfan_backward_dma_batch(const BagDesc* __restrict__ bags, int B,
                                                                     const __bf16* __restrict__ qsplit,
                                                                     const __bf16* __restrict__ dsplit, int P,
                                                                     const float* __restrict__ m2, const float* __restrict__ l,
                                                                     const float* __restrict__ delta, float scale,
                                                                     float* __restrict__ pm, float* __restrict__ pl,
                                                                     float* __restrict__ pacc, int S) {
    using namespace bb;
    constexpr int D = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = w;
    const int g = lane >> 4, i16 = lane & 15;
    const int Gb = gridDim.x / S;            // workgroups (and partials) per bag
    const int grp = blockIdx.x / Gb, b = blockIdx.x % Gb, G = Gb;

    unsigned char* ring = smem + w * kWaveRing;
    unsigned char* exch = smem + kRingBytes;
    int_ma* tab = reinterpret_cast<int_ma*>(smem + kTabOff);
    const bool pok = i16 < P;
    // compact exchange slot of query p = i16 inside its g block, rotated by 4 g: ds_read_b128 is serviced in the lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... and without the rotation the g and g + 1 parts of a group share banks
    const int xslot = i16 < bb::kMaxP ? (i16 + 4 * g) % bb::kMaxP : (4 * g) % bb::kMaxP;

    // ---- bag table: thread t describes this workgroup's rows of bag t -------------------------------------------
    if (tid < B) {
        const BagDesc d = bags[tid];
        // 64-row units (= one lock-step iteration of the two row groups); the workgroup that gets the remainder
        // unit rotates with the bag index so that the extra iterations even out over the batch
        const unsigned long long units = (unsigned long long)((d.N + 31) >> 5);
        const unsigned int uq = (unsigned int)(units / (unsigned int)G), ur = (unsigned int)(units % (unsigned int)G);
        const unsigned int vb = (unsigned int)((b + (tid / S) * 37) % G);  // virtual workgroup index for this bag
        const bool mine = (tid % S) == grp;
        const unsigned long long ubeg = (unsigned long long)vb * uq + (vb < ur ? vb : ur);
        const long long rbeg = (long long)(ubeg << 5);
        long long rend = (long long)((ubeg + uq + (vb < ur ? 1u : 0u)) << 5);
        if (rend > d.N) rend = d.N;
        const int nrows = (mine && rend > rbeg) ? (int)(rend - rbeg) : 0;
        const unsigned long long addr = reinterpret_cast<unsigned long long>(d.X) + (unsigned long long)rbeg * d.ldx * 2ull;
        int_ma* e = tab + tid * 8;
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

    auto tab_get = [&](int bag, int k) -> int { return __builtin_amdgcn_readfirstlane(tab[bag * 8 + k]); };

    const unsigned int ring_lds = (unsigned int)(uintptr_t)(lds_void_ptr)ring;
    const int lr = lane >> 4;
    const int chunk_e = ((lane & 15) ^ (lr << 1)) << 4, chunk_o = ((lane & 15) ^ (lr << 1) ^ 8) << 4;
    // LDS-DMA of one 32-row tile of `bag` into ring slot `slot` (see k_vlfan_partial_dma for the layout)
    // descriptor of the bag the DMA currently streams from, cached in SGPRs (reloaded from the table on a bag change)
    int ib = -1, ildb = 0, voff_e = 0, voff_o = 0;
    i32x4 rsrc = {0, 0, 0, 0x00020000};
    auto issue_tile = [&](int bag, int tile, int slot) {
        if (bag != ib) {
            const int4 e = *reinterpret_cast<const int4*>(smem + kTabOff + bag * 32);
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
        if (tile + 1 < ntiles_bag) {
            nb = bag;
            nt = tile + 1;
            return;
        }
        nb = bag + 1;
        while (nb < B && tab_get(nb, 5) <= 0) ++nb;
        nt = 0;
    };

    int kown = 0;      // own tiles consumed so far by this wave; own tile k lives in ring slot k & 1
    {
        int fb = 0;  // first own tile of the whole batch
        while (fb < B && tab_get(fb, 5) <= 0) ++fb;
        if (fb < B) issue_tile(fb, 0, 0);
    }

    
    f32x4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int bag = 0; bag < B; ++bag) {
        if (tab_get(bag, 7) == 0) continue;  // another group's bag (workgroup-uniform)
        const int nrows = tab_get(bag, 4), ntiles = tab_get(bag, 5);
        const int niter = ntiles;
        if (niter == 0) continue;
        // per-bag upstream gradient: dout fragments (same layout as the query fragments), m2, 1/l, delta
        bf16x8 df[2][4];  // hi + lo of dout (2^-17 relative: far inside the gradient tolerance; the third term only cost MFMAs)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                df[t][kk] = *reinterpret_cast<const bf16x8*>(dsplit + (((size_t)bag * 3 + t) * 16 + i16) * D + cw * 128 + kk * 32 + g * 8);
        float m2p = pok ? m2[(size_t)bag * kPStride + i16] : 0.f;
        float rlp = pok ? 1.f / l[(size_t)bag * kPStride + i16] : 0.f;
        float dlt = pok ? delta[(size_t)bag * kPStride + i16] : 0.f;
        // retire these loads where hipcc can see it (register uses), not inside the tile loop (cf. the query fragments)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(df[t][kk]));
        asm volatile("" : "+v"(m2p), "+v"(rlp), "+v"(dlt));

        for (int it = 0; it < niter; ++it) {
            const int tile = it;
            constexpr bool have = true;
            const int slot = kown & 1;
            const unsigned char* xs = ring + slot * kSlot;
            const int row0 = tile * kTile;
            f32x4 S[2], Nd[2], Dd[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                S[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                Nd[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                Dd[h] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (have) {
                int nb, nt;
                next_of(bag, tile, ntiles, nb, nt);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads of slot^1's old contents have returned
                if (nb < B) {
                    issue_tile(nb, nt, slot ^ 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this tile landed; the next 8 pieces stay in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                bf16x8 xa[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        xa[h][kk] = *reinterpret_cast<const bf16x8_ma*>(xs + wswz(16 * h + i16, kk * 64 + g * 16));
                __builtin_amdgcn_sched_barrier(0);
                f32x4 Sb[2], Db[2];
                Sb[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                Sb[1] = Sb[0];
                Db[0] = Sb[0];
                Db[1] = Sb[0];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        S[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[0][kk], S[h], 0, 0, 0);
                        Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[1][kk], Sb[h], 0, 0, 0);
                        Nd[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], xa[h][kk], Nd[h], 0, 0, 0);
                        Sb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], qf[2][kk], Sb[h], 0, 0, 0);
                        Dd[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], df[0][kk], Dd[h], 0, 0, 0);
                        Db[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[h][kk], df[1][kk], Db[h], 0, 0, 0);
                    }
                S[0] += Sb[0];
                S[1] += Sb[1];
                Dd[0] += Db[0];
                Dd[1] += Db[1];
            }

            VLSA_WBAR();  // readers of the previous exchange are done
            {
                unsigned char* mine = exch + cw * kExchWave;
                if (i16 < kMaxP) {  // compact tiles: [h][g][p < 12] x f32x4
                    *reinterpret_cast<f32x4_ma*>(mine + ((0 * 4 + g) * kMaxP + xslot) * 16) = S[0];
                    *reinterpret_cast<f32x4_ma*>(mine + ((1 * 4 + g) * kMaxP + xslot) * 16) = S[1];
                    *reinterpret_cast<f32x4_ma*>(mine + kTileBytes + ((0 * 4 + g) * kMaxP + xslot) * 16) = Dd[0];
                    *reinterpret_cast<f32x4_ma*>(mine + kTileBytes + ((1 * 4 + g) * kMaxP + xslot) * 16) = Dd[1];
                }
                if (g == (i16 >> 2)) {
                    const int r = i16 & 3;
                    const float d0 = r == 0 ? Nd[0][0] : r == 1 ? Nd[0][1] : r == 2 ? Nd[0][2] : Nd[0][3];
                    const float d1 = r == 0 ? Nd[1][0] : r == 1 ? Nd[1][1] : r == 2 ? Nd[1][2] : Nd[1][3];
                    reinterpret_cast<float_ma*>(mine + 2 * kTileBytes)[i16] = d0;
                    reinterpret_cast<float_ma*>(mine + 2 * kTileBytes)[16 + i16] = d1;
                }
            }
            VLSA_WBAR();
            if (have) {
                f32x4 T[2], DA[2], R2[2];
                {
                    f32x4 tv[2][4], dv[2][4], rv[2][4];
                    const int pidx = xslot;  // lanes >= 12 read slot of p = 0 and compute nothing useful (masked below)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int ww = 0; ww < 4; ++ww) {
                            const unsigned char* o = exch + ww * kExchWave;
                            tv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + ((h * 4 + g) * kMaxP + pidx) * 16);
                            dv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + kTileBytes + ((h * 4 + g) * kMaxP + pidx) * 16);
                            rv[h][ww] = *reinterpret_cast<const f32x4_ma*>(o + 2 * kTileBytes + (16 * h + 4 * g) * 4);
                        }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        T[h] = (tv[h][0] + tv[h][1]) + (tv[h][2] + tv[h][3]);
                        DA[h] = (dv[h][0] + dv[h][1]) + (dv[h][2] + dv[h][3]);
                        R2[h] = (rv[h][0] + rv[h][1]) + (rv[h][2] + rv[h][3]);
                    }
                }
                bf16x8 ahi, alo;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool valid = pok && (row0 + 16 * h + 4 * g + r < nrows);
                        const float inv = fminf(__builtin_amdgcn_rsqf(R2[h][r]), 1e12f);
                        const float A = fast_exp2(T[h][r] * inv - m2p) * rlp;
                        const float u = valid ? A * (DA[h][r] - dlt) * (scale * inv) : 0.f;
                        const __bf16 hi = (__bf16)u;
                        ahi[4 * h + r] = hi;
                        alo[4 * h + r] = (__bf16)(u - (float)hi);
                    }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    const int c_off = ct * 32 + (i16 & 3) * 8;
                    const int rr = 4 * g + (i16 >> 2);
                    const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + wswz(rr, c_off)));
                    const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xs + wswz(16 + rr, c_off)));
                    const bf16x8 bh = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bh, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bh, acc[ct], 0, 0, 0);
                }
                ++kown;
            }
        }
    }

    // ---- single epilogue for the whole batch: this workgroup's partial (pm = 0, pl = 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const size_t slotg = blockIdx.x;
        if (cw == 0 && g == 0 && i16 < P) {
            pm[slotg * kPStride + i16] = 0.f;
            pl[slotg * kPStride + i16] = 1.f;
        }
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 4 * g + r;
                if (p < P) pacc[(slotg * P + p) * D + cw * 128 + ct * 16 + i16] = acc[ct][r];
            }
    }
}

}  // namespace vlsa

using namespace vlsa;

static inline int bwd_groups(int B) { return B >= 8 ? 8 : (B >= 4 ? 4 : (B >= 2 ? 2 : 1)); }

int vlsa_launch_backward_f32_batch(const void* bag_desc, int B, const __bf16* qsplit, const __bf16* dsplit, int P,
                                   const float* m2, const float* l, const float* delta, float scale, float* pm, float* pl,
                                   float* pacc, int S, hipStream_t s);  // vlfan_backward_batch_f32.hip

extern "C" int vlsa_bwd_batch_partials(void) { return 512; }

extern "C" size_t vlsa_bwd_batch_prep_bytes(int B, int D) { return (size_t)B * 3 * 16 * D * 2 + (size_t)B * kPStride * 4; }

extern "C" int vlsa_vlfan_backward_batch(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P,
                                         float coattn_scale, const float* dout, const float* out, const float* m2,
                                         const float* l, void* bwd_prep, float* pm, float* pl, float* pacc, int groups,
                                         void* stream) {
    if (!bag_desc || !qprep || !dout || !out || !m2 || !l || !bwd_prep || !pm || !pl || !pacc) return VLSA_EINVAL;
    if (B < 1 || B > bb::kMaxBags || P < 1 || P > VLSA_MAX_P) return VLSA_EINVAL;
    if (D != 512 || (x_dtype != VLSA_DT_BF16 && x_dtype != VLSA_DT_F32)) return VLSA_EUNSUPPORTED;
    if (x_dtype == VLSA_DT_BF16 && P > bb::kMaxP) return VLSA_EUNSUPPORTED;   // (fp32 bags: any P <= 16)
    int S = groups > 0 ? groups : bwd_groups(B);  // bags in flight: power of two <= min(B, 64)
    {
        int p2 = 1;
        while (p2 * 2 <= S && p2 * 2 <= B && p2 * 2 <= 64) p2 *= 2;
        S = p2;
    }
    hipStream_t s = (hipStream_t)stream;
    __bf16* dsplit = static_cast<__bf16*>(bwd_prep);
    float* delta = reinterpret_cast<float*>(static_cast<unsigned char*>(bwd_prep) + (size_t)B * 3 * 16 * D * 2);
    hipLaunchKernelGGL(k_prepare_backward_batch, dim3(16, B), dim3(256), 0, s, dout, out, P, D, dsplit, delta);
    const QPrepLayout L(D);
    const __bf16* qsplit = reinterpret_cast<const __bf16*>(static_cast<const unsigned char*>(qprep) + L.qsplit);
    if (x_dtype == VLSA_DT_F32)
        return vlsa_launch_backward_f32_batch(bag_desc, B, qsplit, dsplit, P, m2, l, delta, coattn_scale, pm, pl, pacc, S, s);
    static DeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)k_vlfan_backward_dma_batch, hipFuncAttributeMaxDynamicSharedMemorySize, bb::kLdsBytes);
    hipLaunchKernelGGL(k_vlfan_backward_dma_batch, dim3(512), dim3(256), bb::kLdsBytes, s, static_cast<const BagDesc*>(bag_desc), B,
                       qsplit, dsplit, P, m2, l, delta, coattn_scale, pm, pl, pacc, S);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

int vlsa_launch_backward_mfma_bags(const void* bag_desc, int B, int x_dtype, const __bf16* qsplit, const __bf16* dsplit, int P,
                                   const float* m2, const float* l, const float* delta, float scale, float* pm, float* pl,
                                   float* pacc, int G, hipStream_t s);  // vlfan_backward.hip

/* The same backward for batches the persistent kernel does not take (fp32 bags, P > 12): the per-bag kernel of
 * vlfan_backward.hip over the bag table in ONE launch, grid (G, B) -- G row blocks per bag (the caller passes the largest
 * vlsa_num_partials(N_i) of the batch); B * G partial sums in pm (= 0), pl (= 1) [B * G, 16], pacc [B * G, P, D], to be reduced
 * with vlsa_vlfan_merge(..., B * G, normalise = 0).  bwd_prep: vlsa_bwd_batch_prep_bytes(B, D). */
extern "C" int vlsa_vlfan_backward_bags(const void* bag_desc, int B, int x_dtype, int D, const void* qprep, int P,
                                        float coattn_scale, const float* dout, const float* out, const float* m2, const float* l,
                                        void* bwd_prep, float* pm, float* pl, float* pacc, int G, void* stream) {
    if (!bag_desc || !qprep || !dout || !out || !m2 || !l || !bwd_prep || !pm || !pl || !pacc) return VLSA_EINVAL;
    if (B < 1 || B > bb::kMaxBags || P < 1 || P > VLSA_MAX_P || G < 1) return VLSA_EINVAL;
    if (D != 512 || (x_dtype != VLSA_DT_BF16 && x_dtype != VLSA_DT_F32)) return VLSA_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    __bf16* dsplit = static_cast<__bf16*>(bwd_prep);
    float* delta = reinterpret_cast<float*>(static_cast<unsigned char*>(bwd_prep) + (size_t)B * 3 * 16 * D * 2);
    hipLaunchKernelGGL(k_prepare_backward_batch, dim3(16, B), dim3(256), 0, s, dout, out, P, D, dsplit, delta);
    const QPrepLayout L(D);
    const __bf16* qsplit = reinterpret_cast<const __bf16*>(static_cast<const unsigned char*>(qprep) + L.qsplit);
    return vlsa_launch_backward_mfma_bags(bag_desc, B, x_dtype, qsplit, dsplit, P, m2, l, delta, coattn_scale, pm, pl, pacc, G, s);
}

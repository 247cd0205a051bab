// Backward of the aggregation for a BATCH of fp32 bags (the reference's own feature format: dataset/PatchWSI.py:205-215 -- its
// training loop back-propagates 32 fp32 bags per step, runner/vlsa_handler.py:260-289) w.r.t. the shared effective queries:
//     de += scale * sum_bags sum_n A_pn (dout_p . x_n - delta_p) x_n / max(|x_n|, eps)
// Persistent LDS-DMA streaming kernel: the row layout, ring and swizzle of k_vlfan_partial_f32_batch (vlfan_batch_f32.hip), the
// structure of k_vlfan_backward_dma_batch (vlfan_backward_batch.hip): four-wave workgroups = the four 128-column quarters of a
// 16-row tile, 512 workgroups = two per CU, accumulators kept in registers across ALL bags, one partial per workgroup.
// Matrix-pipe work per 16-row tile and wave -- all of it on the bf16 pipe, from ONE set of registers: the row fragments come in
// as 4 consecutive columns per lane and 16-column group (one ds_read_b128 each, as in the forward kernel); split into bf16 hi +
// lo they ARE the A operand of v_mfma_f32_16x16x32_bf16 under the k map  slot e of lane group g of step s <-> column
// 32 s + 16 (e >> 2) + 4 g + (e & 3), which the query and dout fragments share:
//   * scores  S  = X e^T:    X_hi (e0 + e1 + e2) + X_lo (e0 + e1)       20 steps  (3-term queries as in the bf16 kernels)
//   * dA         = X dout^T: X_hi (d0 + d1) + X_lo d0                   12 steps
//   * de += u X with u = A (dA - delta) scale / |x|: 24 steps of 16x16x16 from in-register hi + lo splits, as the forward's sum.
// 896 matrix-pipe cycles per tile instead of 1600 with the scores on the f32 pipe (measured 755 -> 613 us per 32 x 50k bags):
// unlike the forward -- whose exact f32 scores define the attention weights it hands out -- the backward only needs
// A = exp2(t - m2) / l to the gradient tolerance, and the 2^-17 operand split moves t by ~3e-5.
// No P <= 12 restriction here (the exchange tiles are the full 16 x 16).
#include "vlsa_common.h"

namespace vlsa {

typedef __attribute__((address_space(3))) void* lds_void_ptr_bf;
typedef f32x4 __attribute__((may_alias)) f32x4_mbf;
typedef float __attribute__((may_alias)) float_mbf;
typedef int __attribute__((may_alias)) int_mbf;
typedef int i32x4bf __attribute__((ext_vector_type(4)));
typedef unsigned long long u64bf __attribute__((may_alias));

struct BagDescF {
    const void* X;
    int64_t N;
    int64_t ldx;
};

namespace bbf {
constexpr int kTile = 16;                        // rows per tile
constexpr int kSlot = kTile * 512;               // 8 KiB: 16 rows x 128 fp32 columns
constexpr int kWaveRing = 2 * kSlot;
constexpr int kRingBytes = 4 * kWaveRing;        // 64 KiB: four waves
constexpr int kExchWave = 2 * 1024 + 64;         // S tile + dA tile (one f32x4 per lane each) + 16 row sums of squares
constexpr int kExchBytes = 4 * kExchWave;
constexpr int kTabOff = kRingBytes + kExchBytes;
constexpr int kMaxBags = 64;
constexpr int kLdsBytes = kTabOff + kMaxBags * 32;   // 76,032 B: two workgroups per CU
}  // namespace bbf

// element (row, col) of a wave's fp32 slice image (see vlfan_batch_f32.hip)
__device__ __forceinline__ int fswzb(int row, int col) { return row * 512 + ((((col >> 2) ^ (row & 15))) << 4) + ((col & 3) << 2); }

#define VLSA_FBBAR()                                         \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_s_barrier();                        \
        asm volatile("" ::: "memory");                       \
    } while (0)

__global__ __launch_bounds__(256, 2) void k_vlfan_backward_f32_batch(const BagDescF* __restrict__ bags, int B,
                                                                     const __bf16* __restrict__ qsplit,
                                                                     const __bf16* __restrict__ dsplit, int P,
                                                                     const float* __restrict__ m2, const float* __restrict__ l,
                                                                     const float* __restrict__ delta, float scale,
                                                                     float* __restrict__ pm, float* __restrict__ pl,
                                                                     float* __restrict__ pacc, int S) {
    using namespace bbf;
    constexpr int D = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int cw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int Gb = gridDim.x / S;            // workgroups per bag
    const int grp = blockIdx.x / Gb, b = blockIdx.x % Gb, G = Gb;

    unsigned char* ring = smem + cw * kWaveRing;
    unsigned char* exch = smem + kRingBytes;
    int_mbf* tab = reinterpret_cast<int_mbf*>(smem + kTabOff);
    const bool pok = i16 < P;

    // ---- bag table: thread t describes this workgroup's rows of bag t (16-row units) ------------------------------
    if (tid < B) {
        const BagDescF d = bags[tid];
        const unsigned long long units = (unsigned long long)((d.N + 15) >> 4);
        const unsigned int uq = (unsigned int)(units / (unsigned int)G), ur = (unsigned int)(units % (unsigned int)G);
        const unsigned int vb = (unsigned int)((b + (tid / S) * 37) % G);  // virtual workgroup index for this bag
        const bool mine = (tid % S) == grp;
        const unsigned long long ubeg = (unsigned long long)vb * uq + (vb < ur ? vb : ur);
        const long long rbeg = (long long)(ubeg << 4);
        long long rend = (long long)((ubeg + uq + (vb < ur ? 1u : 0u)) << 4);
        if (rend > d.N) rend = d.N;
        const int nrows = (mine && rend > rbeg) ? (int)(rend - rbeg) : 0;
        const unsigned long long addr = reinterpret_cast<unsigned long long>(d.X) + (unsigned long long)rbeg * d.ldx * 4ull;
        int_mbf* e = tab + tid * 8;
        e[0] = (int)(unsigned int)addr;
        e[1] = (int)((addr >> 32) & 0xffffu);
        e[2] = nrows > 0 ? (int)(((long long)(nrows - 1) * d.ldx + D) * 4) : 0;  // descriptor span in bytes
        e[3] = (int)(d.ldx * 4);                                                    // row pitch in bytes
        e[4] = nrows;
        e[5] = (nrows + kTile - 1) / kTile;
        e[6] = 0;
        e[7] = mine ? 1 : 0;
    }
    // query fragments: the 3-term bf16 split of the effective queries (scale * log2 e folded in), in the column map of the row
    // fragments: k-slot e of lane group g of step s <-> column 128 cw + 32 s + 16 (e >> 2) + 4 g + (e & 3)
    bf16x8 qs[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const __bf16* src = qsplit + ((size_t)t * 16 + i16) * D + cw * 128 + 32 * s4 + 4 * g;
            const bf16x4 lo4 = *reinterpret_cast<const bf16x4*>(src), hi4 = *reinterpret_cast<const bf16x4*>(src + 16);
            qs[t][s4] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) asm volatile("" : "+v"(qs[t][s4]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto tab_get = [&](int bag, int k) -> int { return __builtin_amdgcn_readfirstlane(tab[bag * 8 + k]); };

    const unsigned int ring_lds = (unsigned int)(uintptr_t)(lds_void_ptr_bf)ring;
    const int lr = lane >> 5, lc = lane & 31;
    int ib = -1, ildb = 0;
    int voff[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // row of piece i within the tile is 2 i + lr
    i32x4bf rsrc = {0, 0, 0, 0x00020000};
    auto issue_tile = [&](int bag, int tile, int slot) {
        if (bag != ib) {
            const int4 e = *reinterpret_cast<const int4*>(smem + kTabOff + bag * 32);
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

    int kown = 0;      // tiles consumed so far by this wave; tile k lives in ring slot k & 1
    {
        int fb = 0;
        while (fb < B && tab_get(fb, 5) <= 0) ++fb;
        if (fb < B) issue_tile(fb, 0, 0);
    }

    f32x4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int bag = 0; bag < B; ++bag) {
        if (tab_get(bag, 7) == 0) continue;  // another group's bag (workgroup-uniform)
        const int nrows = tab_get(bag, 4), ntiles = tab_get(bag, 5);
        if (ntiles == 0) continue;
        // per-bag upstream gradient: dout fragments (hi + lo bf16, the column map of the score fragments), m2, 1/l, delta
        bf16x8 df[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const __bf16* src = dsplit + (((size_t)bag * 3 + t) * 16 + i16) * D + cw * 128 + 32 * s4 + 4 * g;
                const bf16x4 lo4 = *reinterpret_cast<const bf16x4*>(src), hi4 = *reinterpret_cast<const bf16x4*>(src + 16);
                df[t][s4] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        float m2p = pok ? m2[(size_t)bag * kPStride + i16] : 0.f;
        float rlp = pok ? 1.f / l[(size_t)bag * kPStride + i16] : 0.f;
        float dlt = pok ? delta[(size_t)bag * kPStride + i16] : 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) asm volatile("" : "+v"(df[t][s4]));
        asm volatile("" : "+v"(m2p), "+v"(rlp), "+v"(dlt));

        for (int tile = 0; tile < ntiles; ++tile) {
            const int slot = kown & 1;
            const unsigned char* xs = ring + slot * kSlot;
            const int row0 = tile * kTile;
            f32x4 Sv = {0.f, 0.f, 0.f, 0.f}, Dv = {0.f, 0.f, 0.f, 0.f};
            float ss = 0.f;
            {
                int nb, nt;
                next_of(bag, tile, ntiles, nb, nt);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads of slot^1's old contents have returned
                if (nb < B) {
                    issue_tile(nb, nt, slot ^ 1);
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this tile landed; the next 8 pieces stay in flight
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                float xa[32];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 v = *reinterpret_cast<const f32x4_mbf*>(xs + fswzb(i16, 16 * j + 4 * g));
                    xa[4 * j] = v[0];
                    xa[4 * j + 1] = v[1];
                    xa[4 * j + 2] = v[2];
                    xa[4 * j + 3] = v[3];
                }
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int kk = 0; kk < 32; kk += 2) {
                    s0 = fmaf(xa[kk], xa[kk], s0);
                    s1 = fmaf(xa[kk + 1], xa[kk + 1], s1);
                }
                ss = quad_rows_sum(s0 + s1);
                // scores and dA on the bf16 pipe from the same registers
                f32x4 Db = {0.f, 0.f, 0.f, 0.f}, Sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    bf16x8 xh, xl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = xa[8 * s4 + e];
                        const __bf16 h = (__bf16)v;
                        xh[e] = h;
                        xl[e] = (__bf16)(v - (float)h);
                    }
                    Sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, qs[0][s4], Sv, 0, 0, 0);
                    Dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, df[0][s4], Dv, 0, 0, 0);
                    Sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, qs[1][s4], Sb, 0, 0, 0);
                    Db = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, df[1][s4], Db, 0, 0, 0);
                    Sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, qs[0][s4], Sb, 0, 0, 0);
                    Db = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, df[0][s4], Db, 0, 0, 0);
                    Sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, qs[2][s4], Sb, 0, 0, 0);
                    Sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, qs[1][s4], Sb, 0, 0, 0);
                }
                Sv += Sb;
                Dv += Db;
            }

            VLSA_FBBAR();  // readers of the previous exchange are done
            {
                unsigned char* mine = exch + cw * kExchWave;
                *reinterpret_cast<f32x4_mbf*>(mine + lane * 16) = Sv;
                *reinterpret_cast<f32x4_mbf*>(mine + 1024 + lane * 16) = Dv;
                if (g == 0) reinterpret_cast<float_mbf*>(mine + 2048)[i16] = ss;
            }
            VLSA_FBBAR();
            {
                f32x4 T, DA, R2;
                {
                    f32x4 tv[4], dv[4], rv[4];
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        const unsigned char* o = exch + ww * kExchWave;
                        tv[ww] = *reinterpret_cast<const f32x4_mbf*>(o + lane * 16);
                        dv[ww] = *reinterpret_cast<const f32x4_mbf*>(o + 1024 + lane * 16);
                        rv[ww] = *reinterpret_cast<const f32x4_mbf*>(o + 2048 + 16 * g);
                    }
                    T = (tv[0] + tv[1]) + (tv[2] + tv[3]);
                    DA = (dv[0] + dv[1]) + (dv[2] + dv[3]);
                    R2 = (rv[0] + rv[1]) + (rv[2] + rv[3]);
                }
                float uv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = pok && (row0 + 4 * g + r < nrows);
                    const float inv = fminf(__builtin_amdgcn_rsqf(R2[r]), 1e12f);
                    const float A = fast_exp2(T[r] * inv - m2p) * rlp;
                    uv[r] = valid ? A * (DA[r] - dlt) * (scale * inv) : 0.f;
                }
                // de[p][c] += u[p][n] X[n][c]: A = u (this lane: p = i16, rows 4g .. 4g + 3), B = X[4g + r][16 ct + i16]
                float xb[4][8];
#pragma unroll
                for (int rs = 0; rs < 4; ++rs)
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct) xb[rs][ct] = *reinterpret_cast<const float_mbf*>(xs + fswzb(4 * g + rs, 16 * ct + i16));
                bf16x4 uhi, ulo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uhi[r] = (__bf16)uv[r];
                    ulo[r] = (__bf16)(uv[r] - (float)uhi[r]);
                }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    bf16x4 xh, xl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        xh[r] = (__bf16)xb[r][ct];
                        xl[r] = (__bf16)(xb[r][ct] - (float)xh[r]);
                    }
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(uhi, xh, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ulo, xh, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(uhi, xl, acc[ct], 0, 0, 0);
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

// fp32 bags of vlsa_vlfan_backward_batch: dsplit / delta as produced by k_prepare_backward_batch; 512 partials
int vlsa_launch_backward_f32_batch(const void* bag_desc, int B, const __bf16* qsplit, const __bf16* dsplit, int P,
                                   const float* m2, const float* l, const float* delta, float scale, float* pm, float* pl,
                                   float* pacc, int S, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)k_vlfan_backward_f32_batch, hipFuncAttributeMaxDynamicSharedMemorySize, bbf::kLdsBytes);
    hipLaunchKernelGGL(k_vlfan_backward_f32_batch, dim3(512), dim3(256), bbf::kLdsBytes, s, static_cast<const BagDescF*>(bag_desc), B,
                       qsplit, dsplit, P, m2, l, delta, scale, pm, pl, pacc, S);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

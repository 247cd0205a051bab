// Round 5: k_scores_tile_p -- the (gated) attention scores of model/layers.py:85-153 for LARGE bf16 bags with BOTH operands staged
// through LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 256-row x 256-column tiles walked by persistent eight-wave workgroups.
//
// What the counters of round 5 said about k_gated_scores (gated_scores.hip; profiles/r05_pmc_scores_gated_393216.json): every
// 64-row workgroup tile streams its 512 KB of weight fragments from the L2 into registers -- 7.6 GB of L2 -> CU requests for 0.4 GB
// of HBM traffic at 393 216 patches, the L2s busy 96 % of the kernel.  Here:
//   * a workgroup tile is up to 256 rows x 256 columns (gated: 128 hidden units of both branches, two workgroups per row tile; ungated:
//     all 256 hidden units, ONE workgroup per row tile, no atomics): 6 KB of L2 -> LDS traffic per patch row instead of 18;
//   * the weights of a K step are a 32 KB block that k_prepare_tile_weights has written in EXACTLY the LDS image (hi term, lo term; a
//     column's 64 bytes with their four 16-byte chunks XOR-swizzled for the ds_read_b128 lane groups): the DMA is a linear copy, whole
//     128-byte lines, no address arithmetic, no staging registers;
//   * the X chunk of a K step (256 rows x 64 B) takes the same route, the swizzle on the per-lane SOURCE address (the destination
//     of an LDS-DMA is lane-linear); rows behind the end of the bag read zeros through the buffer descriptor's range check;
//   * a ring of three 48 KB stages: the DMA of steps s + 1 and s + 2 is in flight while step s is consumed (raw s_barrier +
//     hand-counted vmcnt: hipcc would drain every LDS-DMA before the next ds_read); no operand ever sits in a staging register: a
//     wave's budget is its 128 accumulators + 16 fragments;
//   * wave (wm, wn) owns rows [128 wm, 128 wm + 128) x columns [64 wn, 64 wn + 64) of the tile (8 row tiles x 4 column tiles = 128
//     accumulator registers): 8 A fragments + 8 B fragments from LDS per 64 MFMAs; the gate product is wave-local (column tiles 0,
//     1 = branch a, 2, 3 = branch g of the same 32 hidden units).
// Arithmetic (2-term bf16 weights pre-scaled by the exp2 factors, bf16 X exact, fp32 accumulation from the bias), activations,
// dropout and the tile table of a batched launch are those of k_gated_scores.  What was built and measured on the way to this shape
// (one 128-row tile per four-wave workgroup, two per CU; one 256-row tile per eight-wave workgroup; a ping-pong schedule of the two row
// halves with eight barriers per step; K walked in a per-workgroup rotated order) is in DESIGN.md 4.6b with its numbers.
#include <cstdlib>
#include <type_traits>

#include "gated_scores.h"

// Timing-only ablations (results are WRONG with any bit set; tools/gt_ablate.py builds one library per value): 16 = no activations,
// 32 = no MFMAs (the operands still arrive in registers).  (The one-tile-per-workgroup versions of this kernel had more: no weight /
// no X DMA, no barrier, no fragment reads, X rows out of the L2: their results are in DESIGN.md 4.6b.)
#ifndef VLSA_GT_ABL
#define VLSA_GT_ABL 0
#endif
// -DVLSA_GT_STAMP: wave 0 of workgroup 0 records shader-cycle stamps of its first tiles (tools/gt_stamps.py reads them)
#ifdef VLSA_GT_STAMP
__device__ long long vlsa_gt_stamps[256];
#define VLSA_GT_ST(k)                                                                                                 \
    do {                                                                                                              \
        const int k_ = (k);                                                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0 && k_ < 256) vlsa_gt_stamps[k_] = __builtin_readcyclecounter();          \
    } while (0)
extern "C" int vlsa_debug_gt_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vlsa_gt_stamps), sizeof(long long) * 256) == hipSuccess ? 0 : 1;
}
#else
#define VLSA_GT_ST(k) do {} while (0)
#endif

namespace vlsa {

typedef bf16x8 __attribute__((may_alias)) bf16x8_mat;
typedef float __attribute__((may_alias)) float_mat;
typedef int i32x4t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr_t;

namespace gt {
constexpr int kCols = 256;              // columns (hidden unit, branch) per workgroup
constexpr int kB = 2 * kCols * 64;      // weight block of a K step: hi image + lo image, 32 KiB
constexpr int kA = 256 * 64;            // X chunk of a K step: 256 rows x 64 B
constexpr int kStage = kA + kB;         // 48 KiB
constexpr int kRing = 3 * kStage;       // 147 456 B
constexpr int kLdsP = kRing + 4 * 256 * 4 + 1040 * 4 + 256 * 4 + 64;   // + cross-wave scratch + constants + tile scores
}  // namespace gt

// wtile[((hv * 16 + ks) * 2 + term) * 16384 + col * 64 + slot * 16 + 2 e] = term of s_br W_br[h(col)][32 ks + 8 (slot ^ f(col)) + e],
// f(col) = (-(col >> 2)) & 3; col = 64 wn + 16 ct + i: gated: br = ct >> 1, h = 128 hv + 32 wn + 16 (ct & 1) + i; else h = col.
__global__ __launch_bounds__(256) void k_prepare_tile_weights(const float* __restrict__ Wa, const float* __restrict__ Wg, int gated,
                                                              unsigned char* __restrict__ prep) {
    const GatedPrepLayout L(gated);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int slot = idx & 3, col = (idx >> 2) & 255, term = (idx >> 10) & 1, ks = (idx >> 11) & 15, hv = idx >> 15;
    const int c = slot ^ ((0 - (col >> 2)) & 3);
    const int wn = col >> 6, ct = (col >> 4) & 3, i = col & 15;
    const int br = gated ? (ct >> 1) : 0;
    const int h = gated ? 128 * hv + 32 * wn + 16 * (ct & 1) + i : col;
    const float* W = br ? Wg : Wa;
    const float sc = br ? -kLog2e : -2.f * kLog2e;       // the accumulators are the v_exp_f32 arguments (gate_act / tanh_act)
    const int k0 = 32 * ks + 8 * c;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = W[(size_t)h * gs::kD + k0 + e] * sc;
        const __bf16 hi = (__bf16)x;
        o[e] = term ? (__bf16)(x - (float)hi) : hi;
    }
    *reinterpret_cast<bf16x8*>(prep + L.wtile + (size_t)idx * 16) = o;
}

__device__ __forceinline__ f32x4 gt_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
#if VLSA_GT_ABL & 32
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// (m0 is written without being saved: nothing else in this kernel uses it -- checked in the ISA: the only m0 references are these)
#define VLSA_GT_DMA(dst_m0, voff, rsrc, soff)                                                                          \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"                              \
                 :: "s"(dst_m0), "v"(voff), "s"(rsrc), "s"(soff) : "memory")

template <int N> __device__ __forceinline__ void gt_wait_vm() {
    static_assert(N == 0 || N == 6 || N == 10 || N == 12, "a multiple of the DMA instructions of one stage");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_scores_tile_p.  A CU holds ONE such workgroup (151 KB of LDS), so with one tile per workgroup its launch, its first DMA round trip
// to HBM and its epilogue were paid twelve times per CU at 393 216 patches with nothing to overlap them: the workgroups are
// persistent and walk tiles t = b, b + G, ...:
//   * the DMA ring runs across tile boundaries: the last three half-steps of a tile stage steps 1, 2, 0 of the NEXT tile (the
//     buffers those half-steps release; 16 steps over a ring of three leave every tile starting in buffer 0), so the next tile's
//     operands arrive under the activations of this one;
//   * tiles are 16 u <= 256 rows, chosen by the host so that the rows spread evenly over the workgroups (a 50 000-patch bag:
//     per column half 128 tiles of 208 rows, then 113 of 192, instead of 1.53 rounds of 256-row tiles paid as 2): row half 0
//     owns the first (u + 1) / 2 16-row tiles at LDS rows 0 ..., row half 1 the rest at LDS rows 128 ...; a half's unused row tiles are skipped under
//     wave-uniform branches, their DMA instructions are still issued -- with an offset behind the
//     descriptor's range, which reads zeros without a memory request -- so that every wave counts the same vmcnt;
//   * the cross-wave scratch has its own 4 KB behind the ring (buffer 0 is being refilled during the epilogue).
//
// POOL: scores AND the softmax-weighted row sum of the attention pooling (model/layers.py:117-121,148-152) in the same launch (no
// second launch, no second pass of another kernel's latency chain; the L2s still fetch the bag's rows ~2.9 times from the fabric --
// MALL or HBM, FETCH_SIZE cannot tell -- where score kernel + pooling kernel fetch 2.3 times: profiles/r05_pmc_pool_traffic.json; the
// launch is matrix-pipe-bound, not bound by those 3 TB/s).  A workgroup then needs its rows' COMPLETE scores: for the gated module it walks both column halves of its row tile
// one after the other (two passes of 16 steps over the same rows, the second read of X out of the L2 / MALL; the pass's partial scores
// meet in LDS: no atomics, no zeroed output), then takes max, exp and sum over the tile's scores and accumulates w_n x_n over its rows
// (16-byte row loads: the rows just went through this CU's L2) into ONE partial (m, l, acc[512]) per tile, folded per bag by
// k_pool_fold_tiles.
struct GtPool { float* m; float* l; float* acc; };

template <bool GATED, bool POOL>
__global__ __launch_bounds__(512, 2) void k_scores_tile_p(const void* __restrict__ Xv0, long long N0, long long ldx0,
                                                          const unsigned char* __restrict__ prep, float* __restrict__ a_out0,
                                                          int n_tiles, int u_arg, int tall_rounds, const GsBatch bt, const GtPool pool) {
    using namespace gt;
    constexpr int NW = 8, NDMA = 6;             // LDS-DMA instructions per wave and step
    constexpr bool SEQ = GATED && POOL;         // both column halves in this workgroup, one pass each
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int g = lane >> 4, i16 = lane & 15;
    const GatedPrepLayout L(GATED ? 1 : 0);
    // A tile is `u` 16-row tiles high, row half 0 owning the first (u + 1) / 2 of them and row half 1 the rest (the halves share the
    // SIMDs: what counts is their sum).  (A static instantiation spilled 33-40 registers where this one fits: the uniform branches
    // around the row tiles keep the scheduler from hoisting across them.)  One bag: the tiles of the first `tall_rounds` rounds (a
    // round = one tile per walker) have u = u_arg + 1, the others u_arg: the walkers' row counts differ by at most 16 rows (a
    // 50 000-patch bag: per column half 128 tiles of 208 rows, then 113 of 192 -- 25 units per walker, not 2 x 14).  A batch (tile
    // table): every tile u_arg x 16 rows (the table's rows_per_tile, a multiple of 32).
    // gated: the column halves of a row tile on blocks b and b + 8 (same XCD under the round-robin dispatch; gridDim.x is a
    // multiple of 16): workgroup (hv, k) walks row tiles k, k + G / 2, ...
    const int bid = blockIdx.x;
    const int hv0 = (GATED && !SEQ) ? (bid >> 3) & 1 : 0;                  // the column half of this workgroup (SEQ: of its first pass)
    const int first = (GATED && !SEQ) ? ((bid >> 4) << 3) + (bid & 7) : bid;
    const int stride = (GATED && !SEQ) ? (int)(gridDim.x >> 1) : (int)gridDim.x;     // walkers

    i32x4t wrs;             // both column halves' weight blocks: half hv at hv * 16 * kB
    const unsigned long long waddr = reinterpret_cast<unsigned long long>(prep + L.wtile);
    wrs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)waddr);
    wrs[1] = __builtin_amdgcn_readfirstlane((int)((waddr >> 32) & 0xffffu));
    wrs[2] = (GATED ? 2 : 1) * gs::kSteps * kB;
    wrs[3] = 0x00020000;
    const unsigned int lds0 = (unsigned int)(uintptr_t)(lds_void_ptr_t)smem;

    // a tile's X source, all of it wave-uniform (SGPRs): the descriptor over its rows, the row pitch, where its scores go
    struct Src { i32x4t rs; int ldb; float* a; long long row0; int nrows, n0, n1; const unsigned char* x0; };   // n0, n1: 16-row tiles of row half 0 / 1
    const int xr = lane >> 2;
    const int xchunk = ((lane & 3) ^ ((0 - (xr >> 2)) & 3)) << 4;
    // Who stages what: wave w LDS row blocks w and w + 8 and pieces w, w + 8, w + 16, w + 24 of the step's weight block.  (The SIMD's
    // arbiter serves its OLDER wave first -- tools/probes/mfma_issue.hip --, so the waves of row half 0 reach each step's barrier 500-800
    // cycles before their partners.  Letting the older half do ALL of the staging, 12 instructions per step, measured equal: 336.4 vs
    // 337.9 us at 393 216 patches; its barrier wait shrinks by what its own steps grow.)
    constexpr int NJ = 6, NJA = 2, JW = 8;      // instructions per step, of them X row blocks; piece stride
    const int wi = w;
    auto src_of = [&](int t) -> Src {
        const void* Xv = Xv0;
        long long N = N0, ldx = ldx0;
        float* a = a_out0;
        int tile = t;
        if (bt.bags != nullptr) {
            const int ts = lane < bt.B ? bt.tile_start[lane] : 0x7fffffff;
            const int b = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ts <= t)) - 1;
            const GsBag bag = bt.bags[b];
            Xv = bag.X;
            N = bag.N;
            ldx = bag.ldx;
            a += bt.a_off[b];
            tile -= bt.tile_start[b];
        }
        Src r;
        int u = u_arg;
        if (bt.bags != nullptr || tall_rounds == 0) {
            r.row0 = (long long)tile * (16 * u_arg);
        } else {            // round = tile / walkers (one bag: every walker is present in every full round)
            const int rnd = __builtin_amdgcn_readfirstlane(tile / stride), tall = rnd < tall_rounds ? rnd : tall_rounds;
            u = u_arg + (rnd < tall_rounds ? 1 : 0);
            r.row0 = 16ll * stride * ((long long)rnd * u_arg + tall) + 16ll * (tile - rnd * stride) * u;
        }
        u = __builtin_amdgcn_readfirstlane(u);
        r.n0 = (u + 1) >> 1;
        r.n1 = u >> 1;
        const int rows_pt = 16 * u;
        r.nrows = (int)((N - r.row0) < rows_pt ? (N - r.row0) : rows_pt);
        if (r.nrows < 0) r.nrows = 0;
        r.a = a;
        r.ldb = __builtin_amdgcn_readfirstlane((int)(ldx * 2));
        const unsigned long long xaddr = reinterpret_cast<unsigned long long>(Xv) + (unsigned long long)r.row0 * ldx * 2ull;
        r.x0 = reinterpret_cast<const unsigned char*>(xaddr);
        r.rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)xaddr);
        r.rs[1] = __builtin_amdgcn_readfirstlane((int)((xaddr >> 32) & 0xffffu));
        r.rs[2] = __builtin_amdgcn_readfirstlane(r.nrows > 0 ? (int)(((long long)(r.nrows - 1) * ldx + gs::kD) * 2) : 0);
        r.rs[3] = 0x00020000;
        return r;
    };
    // DMA instruction j of step ks into ring buffer buf: j < NJA: LDS row block b = wi + JW j (= tile rows 16 b ..., or 16 (n0 + b - 8) ...
    // for the second row half; a block behind the tile's height: an EMPTY descriptor -- the instruction is still issued and counted, and
    // reads zeros without a memory request); else piece wi + JW (j - NJA) of the step's weight block
    auto issue_one = [&](const Src& sc, int hvx, int ks, int buf, int j) {
        const unsigned int sa = lds0 + buf * kStage, sb = sa + kA;
        if (j < NJA) {
            const int b = wi + JW * j, hb = b & 7;                   // (uniform)
            const int tb = b < 8 ? b : sc.n0 + hb;
            i32x4t d = sc.rs;
            d[2] = __builtin_amdgcn_readfirstlane(hb < (b < 8 ? sc.n0 : sc.n1) ? sc.rs[2] : 0);
            VLSA_GT_DMA(sa + b * 1024, (16 * tb + xr) * sc.ldb + xchunk, d, ks * 64);
        } else {
            VLSA_GT_DMA(sb + (wi + JW * (j - NJA)) * 1024, wi * 1024 + lane * 16, wrs, (hvx * gs::kSteps + ks) * kB + (j - NJA) * JW * 1024);
        }
    };
    auto issue = [&](const Src& sc, int hvx, int ks, int buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) issue_one(sc, hvx, ks, buf, j);
    };

    int t = first;
    if (t >= n_tiles) return;                   // (uniform)
    [[maybe_unused]] int stk = 0;               // (stamp index; dead code without VLSA_GT_STAMP)
    VLSA_GT_ST(stk++);
    Src cur = src_of(t);
    int hv = hv0;                               // the column half of the current pass
    issue(cur, hv, 0, 0);
    issue(cur, hv, 1, 1);
    issue(cur, hv, 2, 2);

    // the workgroup's constants live in LDS (registers are what this kernel is short of): per column half [256] bias of the tile's
    // columns and [256] w2 of the columns' hidden units; c; behind them the tile's scores [256 LDS rows] (SEQ / POOL)
    constexpr int NH = GATED ? 2 : 4;
    float_mat* cst = reinterpret_cast<float_mat*>(smem + kRing + 4096);
    float_mat* tsc = cst + 1040;
    {
        const int hh = tid >> 8, ct256 = tid & 255;            // 512 threads: column half, column
        const int cn = ct256 >> 6, ct = (ct256 >> 4) & 3, ci = ct256 & 15;
        if (GATED || hh == 0) {
            const int h = GATED ? 128 * hh + 32 * cn + 16 * (ct & 1) + ci : ct256;
            cst[512 * hh + ct256] = reinterpret_cast<const float*>(prep + ((GATED && ct >= 2) ? L.bg : L.ba))[h];
            cst[512 * hh + 256 + ct256] = reinterpret_cast<const float*>(prep + L.w2)[h];
        }
        if (tid == 0) cst[1024] = reinterpret_cast<const float*>(prep + L.c)[0];
    }
    // (the wait for these loads, which hipcc places in front of the LDS stores, and the first tile's wait for its operands are one)

    const int frag = i16 * 64 + ((g ^ ((0 - (i16 >> 2)) & 3)) << 4);
    const int a_frag = 128 * wm * 64 + frag, b_frag = kA + (64 * wn) * 64 + frag;
    auto rd = [&](const unsigned char* p) { return *reinterpret_cast<const bf16x8_mat*>(p); };
    float_mat* scr = reinterpret_cast<float_mat*>(smem + kRing);    // [4 column quarters][256 LDS rows]

    // one iteration = one PASS: 16 K steps of one column half over one row tile (SEQ: two passes per tile)
#pragma unroll 1
    for (; t < n_tiles;) {
        const bool last_pass = !SEQ || hv == 1;         // of this tile (uniform)
        const int t_next = last_pass ? t + stride : t, hv_next = SEQ ? (hv ^ 1) : hv;
        const bool has_next = t_next < n_tiles;         // uniform
        const int nrt = wm ? cur.n1 : cur.n0;           // this wave's 16-row tiles
        Src nxt = cur;
        if (has_next && last_pass) nxt = src_of(t_next);
        // steps 0, 1, 2 of this tile have been issued (by the prologue or by the previous tile's last half-steps, step 0 last)
        VLSA_GT_ST(stk++);
        gt_wait_vm<0>();
        VLSA_GT_ST(stk++);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        VLSA_GT_ST(stk++);
        f32x4 acc[8][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const float bv = cst[512 * hv + 64 * wn + 16 * ct + i16];
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) acc[rt][ct] = f32x4{bv, bv, bv, bv};
        }
        bf16x8 A0[4], A1[4], Bh[4], Bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A0[i] = rd(smem + a_frag + i * 1024);
            Bh[i] = rd(smem + b_frag + i * 1024);
            Bl[i] = rd(smem + b_frag + kCols * 64 + i * 1024);
        }
#pragma unroll
        for (int s = 0; s < gs::kSteps; ++s) {
            const unsigned char* st = smem + (s % 3) * kStage;
            const unsigned char* sn = smem + ((s + 1) % 3) * kStage;
            const bool more = s + 1 < gs::kSteps;
#pragma unroll
            for (int i = 0; i < 4; ++i) A1[i] = rd(st + a_frag + (4 + i) * 1024);        // the second four row tiles of this step
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                if (r4 < nrt) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[r4][ct] = gt_mfma(A0[r4], Bh[ct], acc[r4][ct]);
                }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                if (r4 < nrt) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[r4][ct] = gt_mfma(A0[r4], Bl[ct], acc[r4][ct]);
                }
            __builtin_amdgcn_sched_barrier(0);
            // stage s + 1 has landed (s + 2 stays in flight); this wave's reads of stage s have returned: behind the barrier its
            // buffer takes step s + 3 of this tile, or -- s = 13, 14, 15 release buffers 1, 2, 0 -- steps 1, 2, 0 of the next one
            VLSA_GT_ST(stk++);
            if (more && s + 2 < gs::kSteps) gt_wait_vm<NDMA>(); else gt_wait_vm<0>();
            VLSA_GT_ST(stk++);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            VLSA_GT_ST(stk++);
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) A0[i] = rd(sn + a_frag + i * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            // The six DMA instructions of the step are spread over the 32 MFMAs of this half, one per five or six: a wave issues in
            // order, and eight waves pushing 48 LDS-DMA instructions into the CU's one texture-address unit at once (16 cycles each)
            // stalled every wave's MFMAs behind its own last DMA.
            auto dma = [&](int slot) {          // eight slots per step, six instructions
                if (slot < NJ) {
                    if (s + 3 < gs::kSteps) issue_one(cur, hv, s + 3, s % 3, slot);
                    else if (has_next) issue_one(nxt, hv_next, s % 3, s % 3, slot);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (4 + r4 < nrt) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[4 + r4][ct] = gt_mfma(A1[r4], Bh[ct], acc[4 + r4][ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
                dma(r4);
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Bh[i] = rd(sn + b_frag + i * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (4 + r4 < nrt) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[4 + r4][ct] = gt_mfma(A1[r4], Bl[ct], acc[4 + r4][ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
                dma(4 + r4);
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Bl[i] = rd(sn + b_frag + kCols * 64 + i * 1024);
            }
        }

        // ---- epilogue of the tile (the next tile's first three steps are in flight) ------------------------------------------
        VLSA_GT_ST(stk++);
        const unsigned int rid0 = bt.row_base + (unsigned int)cur.row0 + (unsigned int)(wm ? 16 * cur.n0 : 0);
        float w2v[NH];
#pragma unroll
        for (int j = 0; j < NH; ++j) w2v[j] = cst[512 * hv + 256 + 64 * wn + 16 * j + i16];
        auto tail = [&](auto with_dropout) {
#pragma unroll
            for (int rt = 0; rt < 8; ++rt)
                if (rt < nrt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float ew = 0.f;
#pragma unroll
                        for (int j = 0; j < NH; ++j) {
                            float e = (VLSA_GT_ABL & 16) ? acc[rt][j][r] + acc[rt][j + (GATED ? 2 : 0)][r]
                                      : GATED ? gate_act3(acc[rt][j][r], acc[rt][j + 2][r]) : tanh_act3(acc[rt][j][r]);
                            if constexpr (GATED && decltype(with_dropout)::value) {
                                const unsigned int row = rid0 + 16 * rt + 4 * g + r, h = (unsigned int)(128 * hv + 32 * wn + 16 * j + i16);
                                const bool ka = dropout_bits(bt.drop_seed, row, h) >= bt.drop_thr;
                                const bool kg = dropout_bits(bt.drop_seed, row, h + 256u) >= bt.drop_thr;
                                e = (ka && kg) ? e * bt.drop_scale * bt.drop_scale : 0.f;
                            }
                            ew += e * w2v[j];
                        }
                        const float v = row16_sum(ew);
                        if (i16 == 0) scr[wn * 256 + 128 * wm + 16 * rt + 4 * g + r] = v;
                    }
                }
        };
        if (GATED && bt.drop_thr != 0u) tail(std::true_type{}); else tail(std::false_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid < 256) {
            const int lr = tid & 127;
            const int tr = (tid >> 7) * 16 * cur.n0 + lr;       // LDS row -> row of the tile
            const bool valid = lr < 16 * ((tid >> 7) ? cur.n1 : cur.n0) && tr < cur.nrows;
            const float part = scr[tid] + scr[256 + tid] + scr[512 + tid] + scr[768 + tid];
            if constexpr (SEQ) {            // the halves meet in LDS; the second pass stores the finished score
                const float v = hv == 0 ? cst[1024] + part : tsc[tid] + part;
                tsc[tid] = valid ? v : -INFINITY;
                if (hv == 1 && valid) cur.a[cur.row0 + tr] = v;
            } else if (valid) {
                const float sum = (hv == 0 ? cst[1024] : 0.f) + part;
                if (POOL) tsc[tid] = sum;
                if (GATED) atomicAdd(cur.a + cur.row0 + tr, sum);       // two addends per element on a zeroed array: order-independent
                else cur.a[cur.row0 + tr] = sum;
            } else if (POOL) {
                tsc[tid] = -INFINITY;
            }
        }
        if constexpr (POOL) {
            if (last_pass) {
                // ---- pooling partial of the tile: m = max a_n, w_n = e^(a_n - m), l = sum w_n, acc = sum w_n x_n -------------------
                VLSA_GT_ST(stk++);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                float mx = tsc[lane];
                mx = fmaxf(fmaxf(mx, tsc[64 + lane]), fmaxf(tsc[128 + lane], tsc[192 + lane]));
                mx = wave_max(mx);                                      // (every wave: the same 256 values)
                // wave w accumulates LDS rows 32 w .. 32 w + 31 (one row half holds 128: rows of a wave are consecutive tile rows),
                // lane = columns 8 lane .. 8 lane + 7; eight 16-byte row loads in flight
                float accp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float lsum = 0.f;
                const int lr0 = 32 * (w & 3), trb = (w >> 2) * 16 * cur.n0;   // first LDS row inside the half, tile row of the half's row 0
                const int half_rows = 16 * ((w >> 2) ? cur.n1 : cur.n0);
                const unsigned char* xw = cur.x0 + lane * 16;
#pragma unroll 1
                for (int r8 = 0; r8 < 32; r8 += 8) {
                    bf16x8 xv[8];
                    float wv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int lr = lr0 + r8 + i, tr = trb + lr;
                        const bool ok = lr < half_rows && tr < cur.nrows;       // (uniform)
                        wv[i] = ok ? fast_exp2((tsc[128 * (w >> 2) + lr] - mx) * kLog2e) : 0.f;
                        xv[i] = bf16x8{};
                        if (ok) xv[i] = *reinterpret_cast<const bf16x8*>(xw + (long long)tr * cur.ldb);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        lsum += wv[i];
#pragma unroll
                        for (int e = 0; e < 8; ++e) accp[e] += wv[i] * (float)xv[i][e];
                    }
                }
                // the eight waves' partial sums: four rounds of 128 columns through the 4 KB scratch, fixed order
                VLSA_GT_ST(stk++);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                           // everybody has read tsc / scr
                asm volatile("" ::: "memory");
                if (lane == 0) tsc[w] = lsum;                           // (tsc is dead: reused for the eight l sums)
#pragma unroll 1
                for (int q = 0; q < 4; ++q) {
                    if ((lane >> 4) == q) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) scr[w * 128 + (lane & 15) * 8 + e] = accp[e];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (tid < 128) {
                        float sum = 0.f;
#pragma unroll
                        for (int ww = 0; ww < 8; ++ww) sum += scr[ww * 128 + tid];
                        pool.acc[(long long)t * 512 + 128 * q + tid] = sum;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                if (tid == 0) {
                    float l = 0.f;
#pragma unroll
                    for (int ww = 0; ww < 8; ++ww) l += tsc[ww];
                    pool.m[t] = mx;
                    pool.l[t] = l;
                }
            }
        }
        VLSA_GT_ST(stk++);
        cur = nxt;
        t = t_next;
        hv = hv_next;
    }
}

// Fold of the tiles' pooling partials per bag: pooled[b] = sum_t e^(m_t - m) acc_t / sum_t e^(m_t - m) l_t.  grid (B, 16), 512 threads =
// 16 tile groups x 32 columns: group g folds tiles t0 + g, t0 + g + 16, ... with a running maximum (four tiles in flight), then the 16
// groups are combined through LDS in a fixed order.  (One thread per column walking a bag's ~200 tiles one after the other took longer
// than the score kernel of a single 50 000-patch bag.)
__global__ __launch_bounds__(512) void k_pool_fold_tiles(const float* __restrict__ pm, const float* __restrict__ pl,
                                                         const float* __restrict__ pacc, const int* __restrict__ tile_start,
                                                         int n_tiles_single, float* __restrict__ pooled) {
    __shared__ float sm[16], sl[16], sa[16][32];
    const int b = blockIdx.x, g = threadIdx.x >> 5, ci = threadIdx.x & 31, c = blockIdx.y * 32 + ci;
    const int t0 = tile_start ? tile_start[b] : 0, t1 = tile_start ? tile_start[b + 1] : n_tiles_single;
    float m = -INFINITY, l = 0.f, acc = 0.f;
    for (int t = t0 + g; t < t1; t += 64) {
        float mt[4], lt[4], at[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tt = t + 16 * i;
            const bool ok = tt < t1;
            mt[i] = ok ? pm[tt] : -INFINITY;
            lt[i] = ok ? pl[tt] : 0.f;
            at[i] = ok ? pacc[(long long)tt * 512 + c] : 0.f;
        }
        const float mn = fmaxf(fmaxf(m, fmaxf(mt[0], mt[1])), fmaxf(mt[2], mt[3]));
        const float f = m == -INFINITY ? 0.f : __expf(m - mn);
        l *= f;
        acc *= f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float fi = mt[i] == -INFINITY ? 0.f : __expf(mt[i] - mn);
            l += fi * lt[i];
            acc += fi * at[i];
        }
        m = mn;
    }
    if (ci == 0) { sm[g] = m; sl[g] = l; }
    sa[g][ci] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        float M = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) M = fmaxf(M, sm[k]);
        float L = 0.f, A = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float f = sm[k] == -INFINITY ? 0.f : __expf(sm[k] - M);
            L += f * sl[k];
            A += f * sa[k][threadIdx.x];
        }
        pooled[(long long)b * 512 + c] = A / L;
    }
}

// Called by vlsa_prepare_gated_weights (gated_scores.hip) on the same stream: the LDS image behind the fragment-order pack.
int gs_tile_prepare(const float* Wa, const float* Wg, int gated, unsigned char* prep, hipStream_t st) {
    const int chunks = (gated ? 2 : 1) * gs::kSteps * 2 * 256 * 4;
    hipLaunchKernelGGL(k_prepare_tile_weights, dim3(chunks / 256), dim3(256), 0, st, Wa, Wg, gated ? 1 : 0, prep);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// Tiling of a single-bag launch: U = ceil(N / (16 walkers)) 16-row units per walker in R = ceil(U / 16) rounds (a round = one tile per
// walker, at most 256 rows): the first U % R rounds one unit taller than the others.
struct GtPlan { int n_tiles, u, tall_rounds; };
static GtPlan gs_tile_plan(long long N, int walkers) {
    const long long U = (N + 16ll * walkers - 1) / (16ll * walkers), R = (U + 15) / 16;
    const long long base = U / R, rem = U % R;
    GtPlan p;
    p.u = (int)base;
    p.tall_rounds = (int)rem;
    const long long tall_rows = rem * walkers * 16 * (base + 1);
    if (N <= tall_rows) p.n_tiles = (int)((N + 16 * (base + 1) - 1) / (16 * (base + 1)));
    else p.n_tiles = (int)(rem * walkers + (N - tall_rows + 16 * base - 1) / (16 * base));
    return p;
}

// One bag (bt.bags == nullptr: the tiling is chosen here) or the tile table of a batched launch (rows_per_tile: a multiple of 32,
// <= 256; n_tiles = bt.tile_start[B]).  a is zeroed by the caller for the gated module unless `ws` is given.
// ws != nullptr: scores AND attention pooling in this launch -- ws = n_tiles x 514 floats of per-tile partials (m, l, acc[512];
// one bag: gs_tile_pool_tiles(N) tiles), pooled [B, 512] = the softmax-weighted row sums per bag (k_pool_fold_tiles).
int gs_tile_pool_tiles(long long N) { return gs_tile_plan(N, 256).n_tiles; }

int gs_tile_launch(const void* X, long long N, long long ldx, const unsigned char* prep, int gated, float* a, int n_tiles,
                   int rows_per_tile, const GsBatch& bt, float* ws, float* pooled, hipStream_t st) {
    static DeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)k_scores_tile_p<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::kLdsP);
        (void)hipFuncSetAttribute((const void*)k_scores_tile_p<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::kLdsP);
        (void)hipFuncSetAttribute((const void*)k_scores_tile_p<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::kLdsP);
        (void)hipFuncSetAttribute((const void*)k_scores_tile_p<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::kLdsP);
    }
    const bool pool = ws != nullptr;
    if (pool && pooled == nullptr) return VLSA_EINVAL;
    const int walkers = (gated && !pool) ? 128 : 256;       // (pooling: a workgroup walks both column halves of its row tiles)
    int nrt, tall_rounds = 0;      // (nrt: 16-row tiles per tile)
    if (bt.bags == nullptr) {
        const GtPlan pl = gs_tile_plan(N, walkers);
        n_tiles = pl.n_tiles;
        nrt = pl.u;
        tall_rounds = pl.tall_rounds;
    } else {
        if (rows_per_tile < 32 || rows_per_tile > 256 || (rows_per_tile % 32) || n_tiles < 1) return VLSA_EINVAL;
        nrt = rows_per_tile / 16;
    }
    const int wg = n_tiles < walkers ? n_tiles : walkers;
    const unsigned int grid = (gated && !pool) ? 2u * (unsigned)((wg + 7) / 8 * 8) : (unsigned)wg;
    const GtPool gp{ws, ws ? ws + n_tiles : nullptr, ws ? ws + 2ll * n_tiles : nullptr};
#define VLSA_GTP(G, P) hipLaunchKernelGGL((k_scores_tile_p<G, P>), dim3(grid), dim3(512), gt::kLdsP, st, X, N, ldx, prep, a, n_tiles, nrt, tall_rounds, bt, gp)
    if (gated) { if (pool) VLSA_GTP(true, true); else VLSA_GTP(true, false); }
    else       { if (pool) VLSA_GTP(false, true); else VLSA_GTP(false, false); }
#undef VLSA_GTP
    if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    if (pool) {
        hipLaunchKernelGGL(k_pool_fold_tiles, dim3(bt.bags ? bt.B : 1, 16), dim3(512), 0, st, gp.m, gp.l, gp.acc,
                           bt.bags ? bt.tile_start : (const int*)nullptr, n_tiles, pooled);
        if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    }
    return VLSA_OK;
}

}  // namespace vlsa

// Round 5: k_scores_tile -- the (gated) attention scores of model/layers.py:85-153 for LARGE bf16 bags with BOTH operands staged
// through LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 128-row x 256-column workgroup tiles, two workgroups per CU.
//
// What the counters of round 5 said about k_gated_scores (gated_scores.hip; profiles/r05_pmc_scores_gated_393216.json): every
// 64-row workgroup tile streams its 512 KB of weight fragments from the L2 into registers -- 7.6 GB of L2 -> CU requests for 0.4 GB
// of HBM traffic at 393 216 patches, the L2s busy 96 % of the kernel -- and the loads, LDS traffic and MFMAs of a wave overlap badly.
// Here:
//   * a workgroup tile is 128 rows x 256 columns (gated: 128 hidden units of both branches, two workgroups per row tile; ungated: all 256
//     hidden units, ONE workgroup per row tile, no atomics): 8 KB of weights and 1-2 KB of X per patch row instead of 16 + 2;
//   * the weights of a K step are a 32 KB block that k_prepare_tile_weights has written in EXACTLY the LDS image (hi term, lo term; a
//     column's 64 bytes with their four 16-byte chunks XOR-swizzled for the ds_read_b128 lane groups): the DMA is a linear copy, whole
//     128-byte lines, no address arithmetic, no staging registers;
//   * the X chunk of a K step (128 rows x 64 B) takes the same route, the swizzle on the per-lane SOURCE address (the destination
//     of an LDS-DMA is lane-linear); rows behind the end of the bag read zeros through the buffer descriptor's range check;
//   * two stages of 40 KB: the DMA of step s + 1 is in flight while step s is consumed (raw s_barrier + hand-counted vmcnt: hipcc
//     would drain every LDS-DMA before the next ds_read); no operand ever sits in a staging register, so the wave's budget is its
//     128 accumulators + 16 fragments: 2 workgroups x 4 waves per CU, one wave of each per SIMD -- while one workgroup is in its
//     prologue / epilogue (64 activations per lane: 2 v_exp_f32 + 1 v_rcp_f32 each) the other one owns the matrix pipe;
//   * wave w owns columns [64 w, 64 w + 64) of the tile for all 128 rows (8 row tiles x 4 column tiles = 128 accumulator
//     registers): 8 A fragments + 8 B fragments from LDS per 64 MFMAs; the gate product is wave-local (column tiles 0, 1 = branch
//     a, 2, 3 = branch g of the same 32 hidden units).
// Arithmetic (2-term bf16 weights pre-scaled by the exp2 factors, bf16 X exact, fp32 accumulation from the bias), activations,
// dropout and the tile table of a batched launch are those of k_gated_scores.
#include <cstdlib>
#include <type_traits>

#include "gated_scores.h"

// Timing-only ablations (results are WRONG with any bit set; tools/gt_ablate.py builds one library per value): 1 = fragments read
// from LDS in step 0 only, 2 = no weight DMA after step 1, 4 = no X DMA after step 1, 8 = no per-step barrier, 16 = no activations,
// 32 = no MFMAs (the operands still arrive in registers), 64 = every workgroup streams the FIRST tile's X rows (L2 hits instead of
// HBM).
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
// NW waves = NW / 4 row halves x 4 column quarters: 32 NW rows per tile, NST stages of (2 NW + 32) KiB
constexpr int bm(int NW) { return 32 * NW; }
constexpr int stage(int NW) { return bm(NW) * 64 + kB; }
constexpr int lds(int NW, int NST) { return NST * stage(NW); }
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

// NW = 4: 128-row tiles, two workgroups per CU with NST = 2; NW = 8: 256-row tiles (the weight block of a step is shared by twice the
// rows: 6 KB of L2 -> LDS traffic per patch row instead of 10), one workgroup per CU, NST = 3: two steps of DMA in flight.
// PIPE (NST = 3): the K loop software-pipelined inside the wave -- the barrier sits in the MIDDLE of a step, between its two 32-MFMA
// halves, and every fragment read is issued one half-step before its first use, into registers the MFMAs in front of it have
// just released: the matrix pipe has queued work on both sides of every wait.
template <bool GATED, int NW, int NST, bool PIPE = false>
__global__ __launch_bounds__(64 * NW, 2) void k_scores_tile(const void* __restrict__ Xv, long long N, long long ldx,
                                                            const unsigned char* __restrict__ prep, float* __restrict__ a_out,
                                                            const GsBatch bt) {
    using namespace gt;
    constexpr int kBM = bm(NW), kA = kBM * 64, kStage = stage(NW);
    constexpr int NDMA = 2 + 32 / NW;          // LDS-DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;         // row half, column quarter
    const int g = lane >> 4, i16 = lane & 15;
    const GatedPrepLayout L(GATED ? 1 : 0);
    // gated: blocks b and b + 8 (same XCD under the round-robin dispatch) are the two column halves of a row tile: the second
    // read of its X rows is an L2 hit (speed only: any placement is correct)
    const int bid = blockIdx.x, nfull = (int)(gridDim.x >> 4) << 4;
    const int hv = !GATED ? 0 : bid < nfull ? (bid >> 3) & 1 : bid & 1;
    int tile = !GATED ? bid : bid < nfull ? ((bid >> 4) << 3) + (bid & 7) : bid >> 1;
    if (bt.bags != nullptr) {
        const int ts = lane < bt.B ? bt.tile_start[lane] : 0x7fffffff;
        const int b = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ts <= tile)) - 1;
        const GsBag bag = bt.bags[b];
        Xv = bag.X;
        N = bag.N;
        ldx = bag.ldx;
        a_out += bt.a_off[b];
        tile -= bt.tile_start[b];
    }
    const long long row0 = (long long)tile * kBM;
    const int nrows = (int)((N - row0) < kBM ? (N - row0) : kBM);
    const unsigned int rid0 = bt.row_base + (unsigned int)row0;

    // ---- descriptors and per-lane DMA offsets --------------------------------------------------------------------------
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(Xv) + ((VLSA_GT_ABL & 64) ? 0ull : (unsigned long long)row0 * ldx * 2ull);
    i32x4t xrs, wrs;
    xrs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)xaddr);
    xrs[1] = __builtin_amdgcn_readfirstlane((int)((xaddr >> 32) & 0xffffu));
    xrs[2] = __builtin_amdgcn_readfirstlane((int)(((long long)(nrows - 1) * ldx + gs::kD) * 2));
    xrs[3] = 0x00020000;
    const unsigned long long waddr = reinterpret_cast<unsigned long long>(prep + L.wtile) + (unsigned long long)hv * gs::kSteps * kB;
    wrs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)waddr);
    wrs[1] = __builtin_amdgcn_readfirstlane((int)((waddr >> 32) & 0xffffu));
    wrs[2] = gs::kSteps * kB;
    wrs[3] = 0x00020000;
    // X: a wave instruction lands 16 rows x 64 B; lane l -> row l >> 2, LDS slot l & 3 <- source chunk (l & 3) ^ f(row).  The row
    // goes into the VGPR offset (the range check sees it), the K step into the SGPR offset.  Wave w stages row blocks w and w + NW.
    const int ldb = (int)(ldx * 2);
    const int xr = lane >> 2;
    const int xvo0 = (16 * w + xr) * ldb + (((lane & 3) ^ ((0 - (xr >> 2)) & 3)) << 4);
    const int xvo1 = xvo0 + 16 * NW * ldb;
    // weights: linear copy; wave w stages the 1-KiB pieces w, w + NW, ... of the step's 32 KiB block
    const int wvo = w * 1024 + lane * 16;
    const unsigned int lds0 = (unsigned int)(uintptr_t)(lds_void_ptr_t)smem;
    auto issue = [&](int ks, int buf) {
        const unsigned int sa = lds0 + buf * kStage, sb = sa + kA;
        if (!(VLSA_GT_ABL & 4) || ks < 2) {
            VLSA_GT_DMA(sa + w * 1024, xvo0, xrs, ks * 64);
            VLSA_GT_DMA(sa + (w + NW) * 1024, xvo1, xrs, ks * 64);
        }
        if (!(VLSA_GT_ABL & 2) || ks < 2) {
#pragma unroll
            for (int j = 0; j < 32 / NW; ++j) VLSA_GT_DMA(sb + (w + NW * j) * 1024, wvo, wrs, ks * kB + j * NW * 1024);
        }
    };
    issue(0, 0);
    if constexpr (NST == 3) issue(1, 1);

    // ---- accumulators start at the (pre-scaled) bias of the lane's hidden units ------------------------------------------
    // column tile ct of column quarter wn: gated: branch ct >> 1, hidden unit 128 hv + 32 wn + 16 (ct & 1) + i16; ungated: 64 wn + 16 ct + i16
    constexpr int NH = GATED ? 2 : 4;       // hidden units per lane
    float w2v[NH];
    f32x4 acc[8][4];
    {
        float bias[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int h = GATED ? 128 * hv + 32 * wn + 16 * (ct & 1) + i16 : 64 * wn + 16 * ct + i16;
            bias[ct] = reinterpret_cast<const float*>(prep + ((GATED && ct >= 2) ? L.bg : L.ba))[h];
            if (ct < NH) w2v[ct] = reinterpret_cast<const float*>(prep + L.w2)[h];
        }
        // retire these loads where hipcc can see it (a register use): its own s_waitcnt vmcnt(0) for them would otherwise land
        // inside the K loop and drain the DMA there
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) asm volatile("" : "+v"(bias[ct]));
#pragma unroll
        for (int ct = 0; ct < NH; ++ct) asm volatile("" : "+v"(w2v[ct]));
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = f32x4{bias[ct], bias[ct], bias[ct], bias[ct]};
    }
    float cvk = reinterpret_cast<const float*>(prep + L.c)[0];
    asm volatile("" : "+v"(cvk));

    // fragment addresses: row / column r = 16 t + i16 at r * 64, chunk g at position g ^ f(i16)
    const int frag = i16 * 64 + ((g ^ ((0 - (i16 >> 2)) & 3)) << 4);
    const int a_frag = 128 * wm * 64 + frag, b_frag = kA + (64 * wn) * 64 + frag;

    // ---- K loop: 16 steps of 32 ---------------------------------------------------------------------------------------------
    auto rd = [&](const unsigned char* p) { return *reinterpret_cast<const bf16x8_mat*>(p); };
    constexpr bool kRead = !(VLSA_GT_ABL & 1);      // (ablation 1: the fragments of step 0 stay in the registers)
    if constexpr (PIPE) {
        static_assert(NST == 3, "three stages");
        issue(2, 2);
        gt_wait_vm<2 * NDMA>();                             // stage 0 has landed (1 and 2 in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
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
            for (int i = 0; i < 4; ++i) if (kRead || s == 0) A1[i] = rd(st + a_frag + (4 + i) * 1024);        // rows 64..127 of this step
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[r4][ct] = gt_mfma(A0[r4], Bh[ct], acc[r4][ct]);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[r4][ct] = gt_mfma(A0[r4], Bl[ct], acc[r4][ct]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                // stage s + 1 has landed (s + 2 stays in flight); this wave's reads of stage s have returned: after the barrier its
                // buffer takes the DMA of step s + 3
                if (s + 2 < gs::kSteps) gt_wait_vm<NDMA>(); else gt_wait_vm<0>();
                if (!(VLSA_GT_ABL & 8) || s < 2) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (s + 3 < gs::kSteps) issue(s + 3, s % 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) if (kRead) A0[i] = rd(sn + a_frag + i * 1024);           // rows 0..63 of the next step
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[4 + r4][ct] = gt_mfma(A1[r4], Bh[ct], acc[4 + r4][ct]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (kRead) Bh[i] = rd(sn + b_frag + i * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[4 + r4][ct] = gt_mfma(A1[r4], Bl[ct], acc[4 + r4][ct]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) if (kRead) Bl[i] = rd(sn + b_frag + kCols * 64 + i * 1024);
            }
        }
    } else {
#pragma unroll
    for (int s = 0; s < gs::kSteps; ++s) {
        // this wave's share of step s has landed (with three stages the DMA of step s + 1 stays in flight), and its fragment reads of
        // step s - 1 have returned: the DMA issued behind the barrier overwrites that stage
        if (NST == 3 && s + 1 < gs::kSteps) gt_wait_vm<NDMA>(); else gt_wait_vm<0>();
        if (!(VLSA_GT_ABL & 8) || s < 2) __builtin_amdgcn_s_barrier();   // ... everybody's share; everybody is done with step s - 1
        asm volatile("" ::: "memory");
        if (s + NST - 1 < gs::kSteps) issue(s + NST - 1, (s + NST - 1) % NST);
        const unsigned char* st = smem + (s % NST) * kStage;
        bf16x8 Bh[4], Bl[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            Bh[ct] = rd(st + b_frag + ct * 1024);
            Bl[ct] = rd(st + b_frag + kCols * 64 + ct * 1024);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 A[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) A[r4] = rd(st + a_frag + (4 * q + r4) * 1024);
            // hi terms of the 16 accumulators of this half, then the lo terms: MFMAs on one accumulator are 16 apart
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[4 * q + r4][ct] = gt_mfma(A[r4], Bh[ct], acc[4 * q + r4][ct]);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[4 * q + r4][ct] = gt_mfma(A[r4], Bl[ct], acc[4 * q + r4][ct]);
        }
    }
    }

    // ---- epilogue: activations, gate, w2, sum over the lane row's 16 hidden units, over the 4 column quarters, over the column halves
    __syncthreads();                                        // everybody is done with the last stage: stage 0 becomes the scratch
    float_mat* scr = reinterpret_cast<float_mat*>(smem);    // [4 column quarters][kBM rows]
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float ew = 0.f;
#pragma unroll
            for (int j = 0; j < NH; ++j) {
                float e = (VLSA_GT_ABL & 16) ? acc[rt][j][r] + acc[rt][j + (GATED ? 2 : 0)][r]
                          : GATED ? gate_act(acc[rt][j][r], acc[rt][j + 2][r]) : tanh_act(acc[rt][j][r]);
                if (GATED && bt.drop_thr != 0u) {      // uniform: training-mode dropout behind tanh and behind sigmoid
                    const unsigned int row = rid0 + 128 * wm + 16 * rt + 4 * g + r, h = (unsigned int)(128 * hv + 32 * wn + 16 * j + i16);
                    const bool ka = dropout_bits(bt.drop_seed, row, h) >= bt.drop_thr;
                    const bool kg = dropout_bits(bt.drop_seed, row, h + 256u) >= bt.drop_thr;
                    e = (ka && kg) ? e * bt.drop_scale * bt.drop_scale : 0.f;
                }
                ew += e * w2v[j];
            }
            const float v = row16_sum(ew);
            if (i16 == 0) scr[wn * kBM + 128 * wm + 16 * rt + 4 * g + r] = v;
        }
    __syncthreads();
    if (tid < nrows) {
        const float sum = (hv == 0 ? cvk : 0.f) + scr[tid] + scr[kBM + tid] + scr[2 * kBM + tid] + scr[3 * kBM + tid];
        if (GATED) atomicAdd(a_out + row0 + tid, sum);      // two addends per element on a zeroed array: order-independent
        else a_out[row0 + tid] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_scores_tile_p: the pipelined 8-wave shape above with PERSISTENT workgroups.  Counters of the one-tile-per-workgroup kernel
// (profiles/r05_gt_*.txt): with everything but the MFMAs removed it still ran at 72 % matrix-pipe occupancy -- a CU holds one such
// workgroup (144 KB of LDS), so the launch of every workgroup, its first DMA round trip to HBM and its epilogue were paid twelve
// times per CU at 393 216 patches with nothing to overlap them.  Here a workgroup walks tiles t = b, b + G, ...:
//   * the DMA ring runs across tile boundaries: the last three half-steps of a tile stage steps 1, 2, 0 of the NEXT tile (the
//     buffers those half-steps release; 16 steps over a ring of three leave every tile starting in buffer 0), so the next tile's
//     operands arrive under the activations of this one;
//   * tiles are h = 32 nrt <= 256 rows, one height per launch, chosen by the host so that the tiles spread evenly over the
//     workgroups (a 50 000-patch bag: 2 x 224 tiles of 224 rows on 256 workgroups instead of 1.53 rounds of 256-row tiles paid as 2):
//     row half wm owns tile rows [16 nrt wm, 16 nrt (wm + 1)) at LDS rows 128 wm + ...; row tiles >= nrt are skipped under
//     wave-uniform branches (FULL: nrt = 8, static), their DMA instructions are still issued -- with an offset behind the
//     descriptor's range, which reads zeros without a memory request -- so that every wave counts the same vmcnt;
//   * the cross-wave scratch has its own 4 KB behind the ring (buffer 0 is being refilled during the epilogue).
namespace gt {
constexpr int kRing = 3 * stage(8);     // 147 456 B
constexpr int kLdsP = kRing + 4 * 256 * 4 + 3 * 1024;   // + cross-wave scratch + constants
}  // namespace gt

template <bool GATED, bool FULL>
__global__ __launch_bounds__(512, 2) void k_scores_tile_p(const void* __restrict__ Xv0, long long N0, long long ldx0,
                                                          const unsigned char* __restrict__ prep, float* __restrict__ a_out0,
                                                          int n_tiles, int nrt_arg, const GsBatch bt) {
    using namespace gt;
    constexpr int NW = 8, kA = 256 * 64, kStage = stage(NW), NDMA = 6;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int g = lane >> 4, i16 = lane & 15;
    const GatedPrepLayout L(GATED ? 1 : 0);
    const int nrt = FULL ? 8 : nrt_arg;         // 16-row tiles per row half: tile height 32 nrt
    const int rows_pt = 32 * nrt;
    // gated: the column halves of a row tile on blocks b and b + 8 (same XCD under the round-robin dispatch; gridDim.x is a
    // multiple of 16): workgroup (hv, k) walks row tiles k, k + G / 2, ...
    const int bid = blockIdx.x;
    const int hv = GATED ? (bid >> 3) & 1 : 0;
    const int first = GATED ? ((bid >> 4) << 3) + (bid & 7) : bid;
    const int stride = GATED ? (int)(gridDim.x >> 1) : (int)gridDim.x;

    i32x4t wrs;
    const unsigned long long waddr = reinterpret_cast<unsigned long long>(prep + L.wtile) + (unsigned long long)hv * gs::kSteps * kB;
    wrs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)waddr);
    wrs[1] = __builtin_amdgcn_readfirstlane((int)((waddr >> 32) & 0xffffu));
    wrs[2] = gs::kSteps * kB;
    wrs[3] = 0x00020000;
    const int wvo = w * 1024 + lane * 16;
    const unsigned int lds0 = (unsigned int)(uintptr_t)(lds_void_ptr_t)smem;

    // a tile's X source, all of it wave-uniform (SGPRs): the descriptor over its rows (an EMPTY one for a wave whose row blocks lie
    // behind the tile's height: its DMA instructions are still issued and counted, and read zeros without a memory request), the row
    // pitch, where its scores go.  The lane's row indices (LDS row blocks w and w + 8 = tile rows 16 w ... and 16 (nrt + w) ...) do not
    // depend on the tile.
    struct Src { i32x4t rs; int ldb; float* a; long long row0; int nrows; };
    const int xr = lane >> 2;
    const int xchunk = ((lane & 3) ^ ((0 - (xr >> 2)) & 3)) << 4;
    const int xrow0 = 16 * w + xr, xrow1 = 16 * (nrt + w) + xr;
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
        r.row0 = (long long)tile * rows_pt;
        r.nrows = (int)((N - r.row0) < rows_pt ? (N - r.row0) : rows_pt);
        r.a = a;
        r.ldb = __builtin_amdgcn_readfirstlane((int)(ldx * 2));
        const unsigned long long xaddr = reinterpret_cast<unsigned long long>(Xv) + (unsigned long long)r.row0 * ldx * 2ull;
        r.rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned int)xaddr);
        r.rs[1] = __builtin_amdgcn_readfirstlane((int)((xaddr >> 32) & 0xffffu));
        r.rs[2] = w < nrt ? __builtin_amdgcn_readfirstlane((int)(((long long)(r.nrows - 1) * ldx + gs::kD) * 2)) : 0;
        r.rs[3] = 0x00020000;
        return r;
    };
    // DMA instruction j (0, 1: the wave's two X row blocks; 2..5: its four pieces of the weight block) of step ks into ring buffer buf
    auto issue_one = [&](const Src& sc, int ks, int buf, int j) {
        const unsigned int sa = lds0 + buf * kStage, sb = sa + kA;
        if (j == 0) VLSA_GT_DMA(sa + w * 1024, xrow0 * sc.ldb + xchunk, sc.rs, ks * 64);
        else if (j == 1) VLSA_GT_DMA(sa + (w + NW) * 1024, xrow1 * sc.ldb + xchunk, sc.rs, ks * 64);
        else VLSA_GT_DMA(sb + (w + NW * (j - 2)) * 1024, wvo, wrs, ks * kB + (j - 2) * NW * 1024);
    };
    auto issue = [&](const Src& sc, int ks, int buf) {
#pragma unroll
        for (int j = 0; j < 6; ++j) issue_one(sc, ks, buf, j);
    };

    int t = first;
    if (t >= n_tiles) return;                   // (uniform)
    [[maybe_unused]] int stk = 0;               // (stamp index; dead code without VLSA_GT_STAMP)
    VLSA_GT_ST(stk++);
    Src cur = src_of(t);
    issue(cur, 0, 0);
    issue(cur, 1, 1);
    issue(cur, 2, 2);

    // the workgroup's constants live in LDS (registers are what this kernel is short of): [256] bias of the tile's columns, [256]
    // w2 of the columns' hidden units, c
    constexpr int NH = GATED ? 2 : 4;
    float_mat* cst = reinterpret_cast<float_mat*>(smem + kRing + 4096);
    if (tid < 256) {
        const int cn = tid >> 6, ct = (tid >> 4) & 3, ci = tid & 15;
        const int h = GATED ? 128 * hv + 32 * cn + 16 * (ct & 1) + ci : tid;
        cst[tid] = reinterpret_cast<const float*>(prep + ((GATED && ct >= 2) ? L.bg : L.ba))[h];
        cst[256 + tid] = reinterpret_cast<const float*>(prep + L.w2)[h];
        if (tid == 0) cst[512] = reinterpret_cast<const float*>(prep + L.c)[0];
    }
    // (the wait for these loads, which hipcc places in front of the LDS stores, and the first tile's wait for its operands are one)

    const int frag = i16 * 64 + ((g ^ ((0 - (i16 >> 2)) & 3)) << 4);
    const int a_frag = 128 * wm * 64 + frag, b_frag = kA + (64 * wn) * 64 + frag;
    auto rd = [&](const unsigned char* p) { return *reinterpret_cast<const bf16x8_mat*>(p); };
    float_mat* scr = reinterpret_cast<float_mat*>(smem + kRing);    // [4 column quarters][256 LDS rows]

#pragma unroll 1
    for (; t < n_tiles; t += stride) {
        const bool has_next = t + stride < n_tiles;     // uniform
        Src nxt = cur;
        if (has_next) nxt = src_of(t + stride);
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
            const float bv = cst[64 * wn + 16 * ct + i16];
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
                if (FULL || r4 < nrt) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[r4][ct] = gt_mfma(A0[r4], Bh[ct], acc[r4][ct]);
                }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                if (FULL || r4 < nrt) {
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
            auto dma = [&](int j) {
                if (s + 3 < gs::kSteps) issue_one(cur, s + 3, s % 3, j);
                else if (has_next) issue_one(nxt, s % 3, s % 3, j);
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (FULL || 4 + r4 < nrt) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[4 + r4][ct] = gt_mfma(A1[r4], Bh[ct], acc[4 + r4][ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (r4 < 3) dma(r4);
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Bh[i] = rd(sn + b_frag + i * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (FULL || 4 + r4 < nrt) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[4 + r4][ct] = gt_mfma(A1[r4], Bl[ct], acc[4 + r4][ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (r4 < 3) dma(3 + r4);
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Bl[i] = rd(sn + b_frag + kCols * 64 + i * 1024);
            }
        }

        // ---- epilogue of the tile (the next tile's first three steps are in flight) ------------------------------------------
        VLSA_GT_ST(stk++);
        const unsigned int rid0 = bt.row_base + (unsigned int)cur.row0 + (unsigned int)(16 * nrt * wm);
        float w2v[NH];
#pragma unroll
        for (int j = 0; j < NH; ++j) w2v[j] = cst[256 + 64 * wn + 16 * j + i16];
        auto tail = [&](auto with_dropout) {
#pragma unroll
            for (int rt = 0; rt < 8; ++rt)
                if (FULL || rt < nrt) {
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
            const int half_rows = 16 * nrt, lr = tid & 127;
            const int tr = (tid >> 7) * half_rows + lr;         // LDS row -> row of the tile
            if (lr < half_rows && tr < cur.nrows) {
                const float sum = (hv == 0 ? cst[512] : 0.f) + scr[tid] + scr[256 + tid] + scr[512 + tid] + scr[768 + tid];
                if (GATED) atomicAdd(cur.a + cur.row0 + tr, sum);       // two addends per element on a zeroed array: order-independent
                else cur.a[cur.row0 + tr] = sum;
            }
        }
        VLSA_GT_ST(stk++);
        cur = nxt;
    }
}

// Called by vlsa_prepare_gated_weights (gated_scores.hip) on the same stream: the LDS image behind the fragment-order pack.
int gs_tile_prepare(const float* Wa, const float* Wg, int gated, unsigned char* prep, hipStream_t st) {
    const int chunks = (gated ? 2 : 1) * gs::kSteps * 2 * 256 * 4;
    hipLaunchKernelGGL(k_prepare_tile_weights, dim3(chunks / 256), dim3(256), 0, st, Wa, Wg, gated ? 1 : 0, prep);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// Tile height of a single-bag launch of the persistent kernel: 32 k rows, k <= 8, such that the most loaded workgroup has the
// fewest rows (+ ~24 rows' worth of epilogue per tile): wg row-tile walkers (128 per column half for the gated module, 256 else).
static int gs_tile_pick_rows(long long N, int walkers) {
    long long best = -1;
    int pick = 256;
    for (int k = 8; k >= 1; --k) {
        const long long rpt = 32 * k, tiles = (N + rpt - 1) / rpt, per = (tiles + walkers - 1) / walkers;
        const long long cost = per * (rpt + 24);
        if (best < 0 || cost < best) { best = cost; pick = (int)rpt; }
    }
    return pick;
}

// One bag (bt.bags == nullptr: rows_per_tile is chosen here) or the tile table of a batched launch (rows_per_tile: a multiple of 32,
// <= 256; n_tiles = bt.tile_start[B]).  a is zeroed by the caller for the gated module.
int gs_tile_launch(const void* X, long long N, long long ldx, const unsigned char* prep, int gated, float* a, int n_tiles,
                   int rows_per_tile, const GsBatch& bt, hipStream_t st) {
    static DeviceOnce once;
    if (once.first()) {
        (void)hipFuncSetAttribute((const void*)k_scores_tile_p<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::kLdsP);
        (void)hipFuncSetAttribute((const void*)k_scores_tile_p<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::kLdsP);
        (void)hipFuncSetAttribute((const void*)k_scores_tile_p<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::kLdsP);
        (void)hipFuncSetAttribute((const void*)k_scores_tile_p<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::kLdsP);
        (void)hipFuncSetAttribute((const void*)k_scores_tile<true, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::lds(4, 2));
        (void)hipFuncSetAttribute((const void*)k_scores_tile<false, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::lds(4, 2));
        (void)hipFuncSetAttribute((const void*)k_scores_tile<true, 8, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::lds(8, 3));
        (void)hipFuncSetAttribute((const void*)k_scores_tile<false, 8, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gt::lds(8, 3));
    }
    // (A/B hooks: VLSA_GT_SHAPE = 4: one 128-row tile per four-wave workgroup, two per CU; 8: one 256-row tile per eight-wave workgroup)
    static const int shape = [] { const char* e = getenv("VLSA_GT_SHAPE"); return e ? atoi(e) : 0; }();
    if (bt.bags == nullptr && (shape == 4 || shape == 8)) {
        const int rows = shape == 4 ? 128 : 256;
        const unsigned int grid = (gated ? 2u : 1u) * (unsigned)((N + rows - 1) / rows);
        if (shape == 4) {
            if (gated) hipLaunchKernelGGL((k_scores_tile<true, 4, 2>), dim3(grid), dim3(256), gt::lds(4, 2), st, X, N, ldx, prep, a, bt);
            else hipLaunchKernelGGL((k_scores_tile<false, 4, 2>), dim3(grid), dim3(256), gt::lds(4, 2), st, X, N, ldx, prep, a, bt);
        } else {
            if (gated) hipLaunchKernelGGL((k_scores_tile<true, 8, 3, true>), dim3(grid), dim3(512), gt::lds(8, 3), st, X, N, ldx, prep, a, bt);
            else hipLaunchKernelGGL((k_scores_tile<false, 8, 3, true>), dim3(grid), dim3(512), gt::lds(8, 3), st, X, N, ldx, prep, a, bt);
        }
        return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
    }
    const int walkers = gated ? 128 : 256;
    if (bt.bags == nullptr) {
        rows_per_tile = gs_tile_pick_rows(N, walkers);
        n_tiles = (int)((N + rows_per_tile - 1) / rows_per_tile);
    }
    if (rows_per_tile < 32 || rows_per_tile > 256 || (rows_per_tile % 32) || n_tiles < 1) return VLSA_EINVAL;
    const int wg = n_tiles < walkers ? n_tiles : walkers;
    const unsigned int grid = gated ? 2u * (unsigned)((wg + 7) / 8 * 8) : (unsigned)wg;
    // (the static-height instantiation spills 33-40 registers where the run-time one fits: the uniform branches around the row
    // tiles keep the scheduler from hoisting across them; VLSA_GT_FULL=1 selects it for A/B runs)
    static const bool use_full = [] { const char* e = getenv("VLSA_GT_FULL"); return e && atoi(e) == 1; }();
    const bool full = use_full && rows_per_tile == 256;
    const int nrt = rows_per_tile / 32;
#define VLSA_GTP(G, F) hipLaunchKernelGGL((k_scores_tile_p<G, F>), dim3(grid), dim3(512), gt::kLdsP, st, X, N, ldx, prep, a, n_tiles, nrt, bt)
    if (gated) { if (full) VLSA_GTP(true, true); else VLSA_GTP(true, false); }
    else       { if (full) VLSA_GTP(false, true); else VLSA_GTP(false, false); }
#undef VLSA_GTP
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

}  // namespace vlsa

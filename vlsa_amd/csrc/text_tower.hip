// Text side of the path: the frozen CoCa text tower that turns the K ordinal rank prompts into the text
// features [K, 512] -- reference model/prompt_encoder.py:267-322 (CONCHPromptEncoder.forward) over
// model/conch/transformer.py:191-247,290-322 (pre-LN blocks: nn.MultiheadAttention + GELU MLP, no dropout).  The reference
// runs it on K x 128 positions for every bag (1.4 s on the CPU); here it runs once per optimizer step / per
// checkpoint on the rows that can reach the pooled CLS token at all:
//
//   * the causal mask lets position t see only positions <= t, and the CLS row (appended last) sees column 0 and column
//     j + 1 for every non-pad token j (build_cls_mask pads its mask on the LEFT: prompt_encoder.py:245-252) -- for a sentence
//     of n tokens that is positions 0..n, i.e. the sentence plus the FIRST pad position, and not itself.  Rows behind that
//     never influence the output, so a prompt costs n + 2 rows instead of 128 (13 for the shipped rank prompts: exact,
//     not an approximation -- tests/test_text_oracle_golden.py::test_rows_behind_the_sentence...).
//   * the rows of all prompts are packed back to back ("compact rows", M = sum (n_s + 2), padded to a multiple of 48);
//     every GEMM of the tower is then a skinny [M, K] x [K, N] product that streams each weight exactly once from HBM.
//
// All contractions run on the f32 matrix pipe (v_mfma_f32_16x16x4_f32: f32 in / f32 accumulate = an fmaf chain): the text
// features are compared with the fp32 reference at 1e-4 after 12 layers and feed logits scaled by exp(logit_scale) ~ 56,
// so a split-bf16 scheme would have to carry 3 terms per operand and buys nothing at this size (the tower is launch- and
// weight-stream-bound: 340 MB of fp32 weights, 96 kernels).
//
// Kernels
//   k_tt_embed            compact rows <- prompts_embedding[seq, pos] (or cls_emb) + positional_embedding[pos]
//   k_tt_pack             weights -> "tiled" (MFMA-fragment-major) copies, once per weight version: W for the forward products,
//                         W^T for the input-gradient products (the tower is frozen, so both are constants)
//   k_tt_gemm<MT,NW>      Y = pro(A) W^T (+ bias) (+ GELU | * GELU') (+ residual) on tiled operands: every load instruction
//                         reads 1 KB contiguous.  A workgroup owns 16 MT rows x 32 columns; its NW waves split K and are
//                         reduced through LDS (deterministic, no atomics).  pro = LayerNorm fused into the A-operand load
//                         (the wave's whole A slab sits in registers; row statistics by a two-pass reduction across the
//                         waves), or identity.  The one product kernel serves forward, backward and the text projection.
//                         Round 6: pro = LayerNorm BACKWARD as well (both LayerNorm backward passes of a block ride in front of the
//                         products that consume them), and every product touches the NEXT product's weights -- one dword per
//                         128-byte line, into the memory-side Infinity Cache as it turned out (prefetch_next) -- once its own ring is issued.
//   k_tt_attn_fwd/bwd     per (prompt, head) attention over the compact rows: causal for token rows, explicit key list for
//                         the CLS row; <= 128 rows forward, <= 64 rows backward.  Shared prefix: its keys' dK / dV come from one more
//                         workgroup per (1-2 keys, head) over all query rows, from the row statistics and the output the forward
//                         keeps (<= 128 compact rows), else from a ticketed fixed-order fold of the prompts' shares.
//   k_tt_ln_bwd, k_tt_lnf_fwd/bwd, k_tt_scatter   LayerNorm backward (+ residual; the pass's last one also writes d prompts_embedding),
//                         ln_final on the CLS rows, d prompts_embedding (the general route).
// Backward is w.r.t. the prompt embeddings only: the tower is frozen in every shipped configuration
// (vlsa_txt_encoder_frozen: True, cfg_vlsa_conch.yaml:69; runner/vlsa_handler.py:131).
#include "vlsa_common.h"

#include <cstdlib>

namespace vlsa {
namespace tt {

constexpr float kLnEps = 1e-5f;          // nn.LayerNorm default (model/conch/coca_model.py:110: norm_layer = nn.LayerNorm)
constexpr int kHeadDim = 64;

__device__ __forceinline__ float gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// "Tiled" fp32 matrices.  Every operand of the products below -- the weights (packed once: the tower is frozen) and the
// activations (written that way by their producers) -- is stored as 16 x 16 tiles of 1 KB, tiles row-major over
// (row block, column block); inside a tile element (r, c) sits at float ((c >> 2) * 16 + r) * 4 + (c & 3).  That is the
// register image of one v_mfma_f32_16x16x4_f32 operand GROUP (four consecutive MFMA steps): lane l = (c >> 2) * 16 + r
// fetches its four k-values with ONE 16-byte load at tile + 16 l, so a wave instruction reads 1 KB contiguous.
// Why: the same fragments fetched from row-major storage are 16 rows x 64 B per wave instruction, which the texture
// path serves at ~75 cycles per instruction (tools/probes/rowload_probe.hip: 9.1 us vs 4.1 us for 72 loads per wave) -- the
// first version of these products spent 35 us on 5 us of MFMA work, independent of MFMA, HBM and L2 locality.
__device__ __forceinline__ size_t tiled_index(int r, int c, int C) {
    return ((size_t)(r >> 4) * (C >> 4) + (c >> 4)) * 256 + ((((c & 15) >> 2) * 16 + (r & 15)) << 2) + (c & 3);
}

// out (tiled [R, C]) <- W: transpose == 0: R = rows, C = cols, out(r, c) = W[r, c];  transpose != 0: R = cols, C = rows,
// out(r, c) = W[c, r].  One thread per output float, lane-linear inside a tile.
__global__ __launch_bounds__(256) void k_tt_pack(const float* __restrict__ W, int rows, int cols, int transpose, float* __restrict__ out) {
    const int R = transpose ? cols : rows, C = transpose ? rows : cols;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)R * C) return;
    const size_t tile = idx >> 8;
    const int in = (int)(idx & 255), l = in >> 2, i = in & 3;
    const int r = (int)(tile / (C >> 4)) * 16 + (l & 15), c = (int)(tile % (C >> 4)) * 16 + 4 * (l >> 4) + i;
    out[idx] = transpose ? W[(size_t)c * cols + r] : W[(size_t)r * cols + c];
}

// row-major src [n, C] -> tiled dst [n_pad, C], rows n .. n_pad-1 zero.  grid n_pad / 16 row blocks x C / 16.
__global__ __launch_bounds__(256) void k_tt_tile_rows(const float* __restrict__ src, int n, int C, float* __restrict__ dst) {
    const int rb = blockIdx.x, cb = blockIdx.y, in = threadIdx.x, l = in >> 2, i = in & 3;
    const int r = rb * 16 + (l & 15), c = cb * 16 + 4 * (l >> 4) + i;
    dst[((size_t)rb * gridDim.y + cb) * 256 + in] = r < n ? src[(size_t)r * C + c] : 0.f;
}

// compact rows <- prompts_embedding[seq, src] (or cls_emb) + positional_embedding[pos]; row-major x and its tiled copy xt
__global__ __launch_bounds__(256) void k_tt_embed(float* __restrict__ x, float* __restrict__ xt, int d, const float* __restrict__ emb,
                                                 int64_t s_seq, int64_t s_tok, const int* __restrict__ row_seq,
                                                 const int* __restrict__ row_pos, const int* __restrict__ row_src,
                                                 const float* __restrict__ pos_emb, const float* __restrict__ cls_emb, int M) {
    const int row = blockIdx.x;
    float* xr = x + (size_t)row * d;
    const bool live = row < M;
    const int src = live ? row_src[row] : 0;
    const float* e = !live ? nullptr : (src >= 0 ? emb + (size_t)row_seq[row] * s_seq + (size_t)src * s_tok : cls_emb);
    const float* pe = live ? pos_emb + (size_t)row_pos[row] * d : nullptr;
    for (int c = threadIdx.x; c < d; c += 256) {
        const float v = live ? e[c] + pe[c] : 0.f;
        xr[c] = v;
        xt[tiled_index(row, c, d)] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Y[m, n] = sum_k pro(A)[m, k] W[n, k]  (+ bias[n]) (gelu | * gelu'(H)) (+ resid[m, n]);  A tiled [M_pad, K], W tiled [N, K].
// MFMA v_mfma_f32_16x16x4_f32: first operand lane (i = l & 15, kslot = l >> 4) = A[i][k], second operand lane (j = l & 15,
// kslot) = B[k][j], result lane (j, g = l >> 4) holds D[4 g + v][j], v = 0..3; k-slot g of step s of group jj contracts
// column 16 jj + 4 g + s -- the same permutation on both operands.
// A workgroup owns 16 MT rows x 32 columns; its NW waves split K (each wave streams KW = K / NW columns of both operands:
// 1 KB per load instruction, each weight read exactly once from HBM per row group) and are reduced through LDS in a fixed
// order (deterministic, no atomics).  Outputs: row-major Y and / or tiled Yt (the next product's A operand).
enum { PRO_NONE = 0, PRO_LN = 1, PRO_LNBWD = 2 };
enum { EPI_BIAS = 1, EPI_RESID = 2, EPI_GELU = 4, EPI_GELU_BWD = 8 };
constexpr int kSlabMax = 12;   // PRO_LN: (K / NW) / 16 groups of the A slab kept in registers (K <= 768 at NW = 4)

struct GemmArgs {
    const float* A;        // tiled [M_pad, K]
    const float* W;        // tiled [N, K]
    const float* bias;     // [N] or null
    const float* resid;    // row-major [M_pad, ldr] or null
    float* Y;              // row-major [M_pad, ldy] or null
    float* Yt;             // tiled [M_pad, N] or null
    float* Ypre;           // row-major pre-activation copy (EPI_GELU, training) or null
    const float* H;        // row-major [M_pad, ldh]: EPI_GELU_BWD multiplies by gelu'(H)
    const float *ln_w, *ln_b;
    // PRO_LNBWD (round 6): the A operand is the LayerNorm BACKWARD of this row block, evaluated in the prologue --
    //   dx = R2 + LayerNorm'(X2; ln_w)^T A      (A = d LayerNorm output, X2 = the LayerNorm's input, R2 = the residual path's gradient;
    // all three tiled [M_pad, K]) -- and the workgroups of column tile 0 also store dx (tiled Yt2, row-major Y2 [M_pad, ld2]): the
    // k_tt_ln_bwd4 launch in front of the product is gone (23 of the backward pass's 86 launches at 12 blocks)
    const float *X2, *R2;
    float *Y2, *Yt2;
    int ld2;
    // row statistics (mean, rstd) [M_pad][2] of the LayerNorm a PRO_LN product applies: written by its column tile 0 when stats_out is
    // set (the forward that saves for the backward pass), read by the PRO_LNBWD product of the same LayerNorm (stats_in) instead of two
    // passes over the X2 slab and two barrier rounds
    float* stats_out;
    const float* stats_in;
    // the NEXT product's weights (tiled [N' / 16][K' / 16][256]), pulled towards this product's workgroups' successors (into the
    // Infinity Cache, as measured) while this product's MFMA loop drains its own ring (see prefetch_next): pf_tile_floats = floats of one of ITS column tiles (16 NTW' x K'),
    // pf_tiles_xcd = its whole rounds of 8 column tiles (tile t is read by XCD t % 8: the xcd_map of its launch)
    const float* pfW;
    const float* pf2;      // a second region the next launches open with (saved activations of the forward: the LayerNorm-backward
    int pf2_lines;         // prologues' x, the attention backward's q / k / v): pf2_lines 128-byte lines, touched once by the grid
    int pf_tile_floats, pf_tiles_xcd, pf_magic;     // pf_magic = 65536 / pf_tiles_xcd + 1: n / pf_tiles_xcd = (n * pf_magic) >> 16 for the small n here
    int ldr, ldy, ldh, N, K, MG, xcd_map, epi;
    int M_real;            // host only: rows that carry data (0: all M_pad rows); row groups behind them are not launched
    int M_store;           // > 0: Y has only this many rows (a caller's buffer without padding): rows behind them are not stored
    int lda, a_rows;       // NTW == 1 kernels only: lda > 0 = A is ROW-MAJOR [a_rows, lda] (a caller's buffer: the first product of the backward
                           // pass reads the incoming gradient as it lies); rows behind a_rows - 1 repeat the last one (their results are padding)
};

// Tile choice (measured, K = 12 prompts -> 192 padded rows): the operands reach the MFMAs through the CU's L1 at ~46 B/clk, and an
// f32 MFMA group needs 512 B of fresh operands per 2 x 16 x 16 x 16 FLOP unless fragments are reused, so the products sit on
// the L1 limit, not on the matrix pipe: MT = 1 tiles move 1.5 fragments per MFMA (fc2: 170 MB through the L1s = 6 us for 6 us
// of MFMA work), MT = 3 tiles 0.83; wider workgroups (16 waves, 98 KB reduction buffer) measured 2.3x SLOWER (49 vs 21 us).
// GT = number of 16-column groups per wave as a compile-time constant (0: runtime).  With GT known the loops unroll into
// straight-line code and the s_waitcnt pass keeps the prefetched loads in flight; with a runtime trip count it parks the
// ring behind `s_waitcnt vmcnt(0)` + register moves at every basic-block edge.  The CONCH sizes are instantiated, anything
// else takes the runtime path.
#ifdef VLSA_TT_DEBUG
__device__ long long tt_stamps[16];
#define TT_STAMP(k)                                                                                              \
    do {                                                                                                         \
        if (PRO == PRO_LN && blockIdx.x == 5 && threadIdx.x == 0) tt_stamps[k] = __builtin_readcyclecounter();   \
    } while (0)
#else
#define TT_STAMP(k) do {} while (0)
#endif
// NTW = 16-column tiles per workgroup (2 or 3).  The grid should not exceed the 256 CUs by a fraction: 288 or 384 workgroups
// of equal work run as long as 512 (the CUs that get two share their matrix pipes), so the wide products use 48-column
// tiles: QKV 4 x 48 = 192 workgroups, c_fc 4 x 64 = 256 (measured: 24 -> 14 us per launch).
// Weight prefetch for the launch behind this one.  A product's workgroups all start by waiting for their first weight groups from HBM
// (the tower's 340 MB of weights per pass never stay in the 256 MB Infinity Cache), ~2 us in which nothing else happens; the previous
// launch has bandwidth to spare, so each of its threads touches up to kPfLoads 128-byte lines of the next product's weights -- one dword
// per line.  The lines are dealt so that block b (XCD b % 8) touches the column tiles XCD b % 8 will read in the next launch, but what the
// counters show is that the L2s do NOT keep them across the kernel boundary: FETCH_SIZE per forward pass went from 660 to 985 MB
// (profiles/r06_pmc_text_tower_prefetch.json against r06_pmc_text_tower.json: every product still fetches its own 2.4-9.4 MB and now
// also the next one's) -- the second fetch is served by the memory-side Infinity Cache instead of HBM, and that is the gain
// (forward 583 -> 540 us, 527 with every block on block 0's weights).  Issued once the last ring refill is out (no later load of the wave queues behind them: vmcnt retires in order);
// the values are consumed by an empty asm at the end of the kernel.
constexpr int kPfLoads = 4, kPf2Loads = 2, kPfRegs = kPfLoads + kPf2Loads;
__device__ __forceinline__ void prefetch_next(const GemmArgs& p, int nthreads, float (&pfv)[kPfRegs]) {
    // BRANCH-FREE on purpose: behind a conditional load the compiler's s_waitcnt pass no longer knows how many loads are in flight and
    // parks the rest of the ring behind the prefetch; so every thread always issues kPfLoads loads -- without a target (or out of range)
    // they re-read one line of this product's own weights
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int nj = ((int)gridDim.x - x + 7) >> 3;           // workgroups of this launch on XCD x
    const int nT = p.pf_tiles_xcd;                          // column tiles of the next launch on XCD x: x, x + 8, ...
    const bool on = p.pfW != nullptr && nj >= nT;
    const int jj = (j * p.pf_magic) >> 16, t = j - jj * nT;            // this workgroup's tile (j % nT), its place among the cnt that share it
    const int cnt = ((nj - t + nT - 1) * p.pf_magic) >> 16;
    const int LT = on ? p.pf_tile_floats >> 5 : 1;          // 128-byte lines of a tile
    const float* base = on ? p.pfW + (size_t)(x + 8 * t) * p.pf_tile_floats : p.W;
#pragma unroll
    for (int u = 0; u < kPfLoads; ++u) {
        int o = (jj + u * cnt) * nthreads + (int)threadIdx.x;
        o = o < LT ? o : 0;
        pfv[u] = base[(size_t)o * 32];
    }
    const float* b2 = p.pf2 != nullptr ? p.pf2 : p.W;
    const int n2 = p.pf2 != nullptr ? p.pf2_lines : 1;
#pragma unroll
    for (int u = 0; u < kPf2Loads; ++u) {
        int o = ((int)blockIdx.x + u * (int)gridDim.x) * nthreads + (int)threadIdx.x;      // every line once, by the whole grid (see above: what
        o = o < n2 ? o : 0;                                                                // is warmed is the memory-side cache, not an L2)
        pfv[kPfLoads + u] = b2[(size_t)o * 32];
    }
}

template <int MT, int NW, int PRO, int GT, int NTW = 2>
__global__ __launch_bounds__(NW * 64) void k_tt_gemm(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    constexpr int Q = MT * NTW * 4;        // accumulator registers per lane
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int K = p.K, N = p.N, MG = p.MG, epi = p.epi;
    TT_STAMP(0);
    int ntile, mg;
    {
        const int b = blockIdx.x;
        // the MG workgroups that share a weight tile run on one XCD (block b -> XCD b % 8): W comes from ITS L2.  A column-tile count
        // that is no multiple of 8 maps its whole rounds of 8 tiles that way and spreads the last NT % 8 tiles linearly (round 4: the QKV
        // product, 36 tiles, ran unmapped until the tower's FETCH_SIZE pass showed 52.6 MB per launch for 7.1 MB of weights -- each of a
        // tile's 7 row-group workgroups pulled it into a different XCD's L2, profiles/r04_pmc_text_tower.json; padding to 40 tile slots
        // instead put 35 workgroups on four XCDs' 32 CUs and cost 40 us per pass)
        const int NTf = (N / (16 * NTW)) & ~7;
        if (p.xcd_map && b < NTf * MG) {
            const int j = b >> 3;
            ntile = (j / MG) * 8 + (b & 7);
            mg = j % MG;
        } else if (p.xcd_map) {
            const int rr = b - NTf * MG;
            ntile = NTf + rr / MG;
            mg = rr % MG;
        } else {
            ntile = b / MG;
            mg = b % MG;
        }
    }
    const int n0 = ntile * (16 * NTW), m0 = mg * (16 * MT);
    const int KW = K / NW, kbeg = w * KW, KG = K >> 4;
    const int G = GT > 0 ? GT : (KW >> 4);
    const float* Ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) Ap[t] = p.A + ((size_t)((m0 >> 4) + t) * KG + (kbeg >> 4)) * 256 + lane * 4;
    int astep = 256;       // floats from one 16-column group of the A operand to the next
    if constexpr (NTW == 1) {
        if (p.lda > 0) {   // row-major A: lane (r, g) of a fragment holds A[row r][16 jj + 4 g .. + 3]
            astep = 16;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int row = m0 + 16 * t + r;
                Ap[t] = p.A + (size_t)(row < p.a_rows ? row : p.a_rows - 1) * p.lda + kbeg + 4 * g;
            }
        }
    }
    const float* Wp[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) Wp[u] = p.W + ((size_t)((n0 >> 4) + u) * KG + (kbeg >> 4)) * 256 + lane * 4;

    f32x4 acc[MT][NTW];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NTW; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (compile-time trip counts only; the 8-wave 48-column shape has no four registers to spare: 128 per lane)
    constexpr bool kPf = GT > 0 && !(NW == 8 && NTW == 3);
    float pfv[kPfRegs] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // The epilogue's inputs (bias, gelu' argument, residual) of the accumulator registers this wave will finalise (q = w, w + NW, ...)
    // are fetched NOW: read behind the reduction they are dependent loads that cannot be batched (the stores of one register's
    // results may alias the next one's loads as far as the compiler knows).  Worth ~1 % of the tower: the whole epilogue is 80 of
    // 668 us over the 49 products of a forward pass (VLSA_TT_DEBUG_ALL=8).
    constexpr int QW = (Q + NW - 1) / NW;
    float e_bias[QW], e_h[QW], e_res[QW];
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        const int q = w + i * NW;
        e_bias[i] = 0.f; e_h[i] = 0.f; e_res[i] = 0.f;
        if (q < Q) {
            const int t = q / (NTW * 4), u = (q >> 2) % NTW, v = q & 3;
            const int row = m0 + 16 * t + 4 * g + v, col = n0 + 16 * u + r;
            if (epi & EPI_BIAS) e_bias[i] = p.bias[col];
            if (epi & EPI_GELU_BWD) e_h[i] = p.H[(size_t)row * p.ldh + col];
            if (epi & EPI_RESID) e_res[i] = p.resid[(size_t)row * p.ldr + col];
        }
    }

    constexpr int PF = MT == 1 ? 8 : 6;    // weight (and, without LayerNorm, activation) register ring: groups in flight
#ifdef VLSA_TT_DEBUG
    const int dbg = epi >> 8;
    if (dbg & 4) return;                                    // launch + set-up only
    const int gstep = (dbg & 2) ? 0 : 256;                  // every group re-reads tile 0 (L1 resident)
#define TT_GSTEP gstep
#define TT_NOMFMA (dbg & 1)
#else
#define TT_GSTEP 256
#define TT_NOMFMA 0
#endif
    if constexpr (PRO == PRO_LN) {
        // ---- LayerNorm fused into the operand load: this wave's [16 MT rows] x [KW columns] slab of A in registers ------
        // the affine parameters go through LDS (first loads issued; reading them back later is an LDS access that does not
        // queue behind the weight stream the way a global load would: vmcnt is in-order)
        float* sgam = red + 1024;           // [K] gamma | [K] beta   (K <= 768: 6 KB of the reduction buffer)
        f32x4 gld = {0.f, 0.f, 0.f, 0.f}, bld = {0.f, 0.f, 0.f, 0.f};
        if (tid * 4 < K) {
            gld = *reinterpret_cast<const f32x4*>(p.ln_w + tid * 4);
            bld = *reinterpret_cast<const f32x4*>(p.ln_b + tid * 4);
        }
        f32x4 slab[MT][kSlabMax];
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
#pragma unroll
                for (int t = 0; t < MT; ++t) slab[t][jj] = *reinterpret_cast<const f32x4*>(Ap[t] + TT_GSTEP * jj);
            }
        f32x4 rb[PF][NTW];
#pragma unroll
        for (int sI = 0; sI < PF; ++sI)
            if (sI < G) {
#pragma unroll
                for (int u = 0; u < NTW; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + TT_GSTEP * sI);
            }
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler would otherwise sink the prefetch next to its uses)
        TT_STAMP(1);
        if (tid * 4 < K) {
            *reinterpret_cast<f32x4*>(sgam + tid * 4) = gld;
            *reinterpret_cast<f32x4*>(sgam + K + tid * 4) = bld;
        }
        float* st = red;                    // [NW][16 MT] partial row statistics
        float mean[MT], rstd[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) s += (slab[t][jj][0] + slab[t][jj][1]) + (slab[t][jj][2] + slab[t][jj][3]);
            s = quad_rows_sum(s);
            if (g == 0) st[w * (16 * MT) + 16 * t + r] = s;
        }
        TT_STAMP(2);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += st[ww * (16 * MT) + 16 * t + r];
            mean[t] = s / (float)K;
        }
        TT_STAMP(3);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float c = slab[t][jj][i] - mean[t];
                        s = fmaf(c, c, s);
                    }
                }
            s = quad_rows_sum(s);
            if (g == 0) st[w * (16 * MT) + 16 * t + r] = s;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s += st[ww * (16 * MT) + 16 * t + r];
            rstd[t] = 1.f / sqrtf(s / (float)K + kLnEps);
        }
        TT_STAMP(4);
        if (p.stats_out && ntile == 0 && w == 0 && g == 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t) *reinterpret_cast<float2*>(p.stats_out + 2 * (m0 + 16 * t + r)) = float2{mean[t], rstd[t]};
        }
        // normalise the slab in place (affine parameters from LDS); the MFMA loop below then runs on registers + the ring
        const float* gw = sgam + kbeg + 4 * g;
        const float* gb = sgam + K + kbeg + 4 * g;
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
                const f32x4 gam = *reinterpret_cast<const f32x4*>(gw + 16 * jj);
                const f32x4 bet = *reinterpret_cast<const f32x4*>(gb + 16 * jj);
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) slab[t][jj][i] = fmaf((slab[t][jj][i] - mean[t]) * rstd[t], gam[i], bet[i]);
            }
        __builtin_amdgcn_sched_barrier(0);
        TT_STAMP(5);
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
                const int sI = jj % PF;
                f32x4 b[NTW];
#pragma unroll
                for (int u = 0; u < NTW; ++u) b[u] = rb[sI][u];
                if (jj + PF < G) {
#pragma unroll
                    for (int u = 0; u < NTW; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + TT_GSTEP * (jj + PF));
                }
                if (kPf && jj == (GT > PF ? GT - PF : 0)) prefetch_next(p, NW * 64, pfv);
                __builtin_amdgcn_sched_barrier(0);
                if (TT_NOMFMA) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t][0] += slab[t][jj];
#pragma unroll
                    for (int u = 0; u < NTW; ++u) acc[0][u] += b[u];
                } else
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int u = 0; u < NTW; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(slab[t][jj][i], b[u][i], acc[t][u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        TT_STAMP(6);
        __syncthreads();                    // every wave is done with gamma / beta in the buffer the reduction reuses
    } else if constexpr (PRO == PRO_LNBWD) {
        // ---- LayerNorm BACKWARD fused into the operand load: three [16 MT rows] x [KW columns] slabs in registers (d LN output, the
        // LN's input, the residual gradient), row statistics across the NW waves through LDS (one barrier per pass: each pass has its own
        // scratch), dx left in the first slab = this wave's A operand.  Same MFMA loop and weight ring as PRO_LN behind it.
        float* sgam = red + 1024;           // [K] gamma
        f32x4 gld = {0.f, 0.f, 0.f, 0.f};
        if (tid * 4 < K) gld = *reinterpret_cast<const f32x4*>(p.ln_w + tid * 4);
        f32x4 slab[MT][kSlabMax], sx[MT][kSlabMax], sr[MT][kSlabMax];
        const size_t aoff0 = ((size_t)(m0 >> 4) * KG + (kbeg >> 4)) * 256 + lane * 4;
        const bool have_stats = p.stats_in != nullptr;      // (uniform)
        float2 stin[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
            stin[t] = have_stats ? *reinterpret_cast<const float2*>(p.stats_in + 2 * (m0 + 16 * t + r)) : float2{0.f, 1.f};
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const size_t o = aoff0 + (size_t)t * KG * 256 + (size_t)256 * jj;
                    slab[t][jj] = *reinterpret_cast<const f32x4*>(p.A + o);
                    sx[t][jj] = *reinterpret_cast<const f32x4*>(p.X2 + o);
                    sr[t][jj] = *reinterpret_cast<const f32x4*>(p.R2 + o);
                }
            }
        f32x4 rb[PF][NTW];
#pragma unroll
        for (int sI = 0; sI < PF; ++sI)
            if (sI < G) {
#pragma unroll
                for (int u = 0; u < NTW; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 256 * sI);
            }
        __builtin_amdgcn_sched_barrier(0);
        if (tid * 4 < K) *reinterpret_cast<f32x4*>(sgam + tid * 4) = gld;
        float* st1 = red;                   // [NW][16 MT] per pass
        float* st2 = red + 128 * MT;
        float* st3 = red + 256 * MT;
        float* st4 = red + 384 * MT;
        float mean[MT], rstd[MT];
        if (have_stats) {
#pragma unroll
            for (int t = 0; t < MT; ++t) { mean[t] = stin[t].x; rstd[t] = stin[t].y; }
        } else {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float sm = 0.f;
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) sm += (sx[t][jj][0] + sx[t][jj][1]) + (sx[t][jj][2] + sx[t][jj][3]);
            sm = quad_rows_sum(sm);
            if (g == 0) st1[w * (16 * MT) + 16 * t + r] = sm;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float sm = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sm += st1[ww * (16 * MT) + 16 * t + r];
            mean[t] = sm / (float)K;
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float sm = 0.f;
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float cc = sx[t][jj][i] - mean[t];
                        sm = fmaf(cc, cc, sm);
                    }
                }
            sm = quad_rows_sum(sm);
            if (g == 0) st2[w * (16 * MT) + 16 * t + r] = sm;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float sm = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sm += st2[ww * (16 * MT) + 16 * t + r];
            rstd[t] = 1.f / sqrtf(sm / (float)K + kLnEps);
        }
        }
        if (have_stats) __syncthreads();    // gamma is in LDS
        // x^ in place, g = d LN output * gamma in place; row sums of g and g x^
        const float* gw = sgam + kbeg + 4 * g;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) {
                    const f32x4 gam = *reinterpret_cast<const f32x4*>(gw + 16 * jj);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xh = (sx[t][jj][i] - mean[t]) * rstd[t];
                        const float gv = slab[t][jj][i] * gam[i];
                        sx[t][jj][i] = xh;
                        slab[t][jj][i] = gv;
                        s1 += gv;
                        s2 = fmaf(gv, xh, s2);
                    }
                }
            s1 = quad_rows_sum(s1);
            s2 = quad_rows_sum(s2);
            if (g == 0) {
                st3[w * (16 * MT) + 16 * t + r] = s1;
                st4[w * (16 * MT) + 16 * t + r] = s2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                s1 += st3[ww * (16 * MT) + 16 * t + r];
                s2 += st4[ww * (16 * MT) + 16 * t + r];
            }
            const float sg = s1 / (float)K, sgx = s2 / (float)K;
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        slab[t][jj][i] = sr[t][jj][i] + rstd[t] * (slab[t][jj][i] - sg - sx[t][jj][i] * sgx);
                }
        }
        if (ntile == 0) {                   // one workgroup per row block hands dx on (the residual path of the next LayerNorm backward)
#pragma unroll
            for (int jj = 0; jj < kSlabMax; ++jj)
                if (jj < G) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        if (p.Yt2) *reinterpret_cast<f32x4*>(p.Yt2 + aoff0 + (size_t)t * KG * 256 + (size_t)256 * jj) = slab[t][jj];
                        if (p.Y2)
                            *reinterpret_cast<f32x4*>(p.Y2 + (size_t)(m0 + 16 * t + r) * p.ld2 + kbeg + 16 * jj + 4 * g) = slab[t][jj];
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < kSlabMax; ++jj)
            if (jj < G) {
                const int sI = jj % PF;
                f32x4 b[NTW];
#pragma unroll
                for (int u = 0; u < NTW; ++u) b[u] = rb[sI][u];
                if (jj + PF < G) {
#pragma unroll
                    for (int u = 0; u < NTW; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 256 * (jj + PF));
                }
                if (kPf && jj == (GT > PF ? GT - PF : 0)) prefetch_next(p, NW * 64, pfv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int u = 0; u < NTW; ++u)
                            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(slab[t][jj][i], b[u][i], acc[t][u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        __syncthreads();                    // every wave is done with gamma / the statistics in the buffer the reduction reuses
    } else {
        // register ring PF groups deep: the loads of group jj + PF are issued when group jj is consumed (vmcnt returns in order)
        f32x4 ra[PF][MT], rb[PF][NTW];
#pragma unroll
        for (int sI = 0; sI < PF; ++sI)
            if (sI < G) {
#pragma unroll
                for (int t = 0; t < MT; ++t) ra[sI][t] = *reinterpret_cast<const f32x4*>(Ap[t] + (NTW == 1 ? astep : 256) * sI);
#pragma unroll
                for (int u = 0; u < NTW; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 256 * sI);
            }
        // keep the machine scheduler from sinking the prefetch loads next to their uses (it minimises register pressure and
        // would leave ~2 groups in flight): nothing moves across these fences
        __builtin_amdgcn_sched_barrier(0);
        auto stage = [&](int sI, int jj) __attribute__((always_inline)) {
            f32x4 a[MT], b[NTW];
#pragma unroll
            for (int t = 0; t < MT; ++t) a[t] = ra[sI][t];
#pragma unroll
            for (int u = 0; u < NTW; ++u) b[u] = rb[sI][u];
            if (jj + PF < G) {
#pragma unroll
                for (int t = 0; t < MT; ++t) ra[sI][t] = *reinterpret_cast<const f32x4*>(Ap[t] + (NTW == 1 ? astep : 256) * (jj + PF));
#pragma unroll
                for (int u = 0; u < NTW; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 256 * (jj + PF));
            }
            if (kPf && jj == (GT > PF ? GT - PF : 0)) prefetch_next(p, NW * 64, pfv);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int u = 0; u < NTW; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][i], b[u][i], acc[t][u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (GT > 0) {
#pragma unroll
            for (int jj = 0; jj < GT; ++jj) stage(jj % PF, jj);
        } else {
            for (int j0 = 0; j0 < G; j0 += PF) {
#pragma unroll
                for (int sI = 0; sI < PF; ++sI)
                    if (j0 + sI < G) stage(sI, j0 + sI);
            }
        }
    }

    // ---- reduce the NW K-slices through LDS (fixed order), then bias / activation / residual and store ------------------
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NTW; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[(w * Q + (t * NTW + u) * 4 + v) * 64 + lane] = acc[t][u][v];
    __syncthreads();
    TT_STAMP(7);
#ifdef VLSA_TT_DEBUG
    if (dbg & 8) return;                                    // no epilogue at all
#endif
#pragma unroll
    for (int i = 0; i < QW; ++i) {          // accumulator register q of every lane: summed and stored by wave q % NW
        const int q = w + i * NW;
        if (q >= Q) break;
        float val = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) val += red[(ww * Q + q) * 64 + lane];
        const int t = q / (NTW * 4), u = (q >> 2) % NTW, v = q & 3;
        const int row = m0 + 16 * t + 4 * g + v, col = n0 + 16 * u + r;
        if (epi & EPI_BIAS) val += e_bias[i];
        if (epi & EPI_GELU) {
            if (p.Ypre) p.Ypre[(size_t)row * p.ldy + col] = val;
            val = gelu(val);
        }
        if (epi & EPI_GELU_BWD) val *= gelu_grad(e_h[i]);
        if (epi & EPI_RESID) val += e_res[i];
        if (p.Y && (p.M_store == 0 || row < p.M_store)) p.Y[(size_t)row * p.ldy + col] = val;
        if (p.Yt) p.Yt[tiled_index(row, col, N)] = val;
    }
    if constexpr (kPf) asm volatile("" ::"v"(pfv[0]), "v"(pfv[1]), "v"(pfv[2]), "v"(pfv[3]), "v"(pfv[4]), "v"(pfv[5]));      // the prefetched lines' only consumer
    TT_STAMP(8);
}

// ---------------------------------------------------------------------------------------------------------------
// attention of one (prompt, head) over the prompt's compact rows.  Token row i sees rows j <= i (causal); the CLS row (last)
// sees the rows flagged in cls_keep (model/prompt_encoder.py:245-252,299-303).
// Shared prefix (prefix_len = L > 0): the first L positions of every prompt carry identical embeddings (<sot> + the shared
// context tokens of the rank prompts, model/prompt_learners/rank_prompt_learner.py:116-156 with the rank tokens at the tail), so
// under the causal mask their activations are identical in every layer: they are stored ONCE (rows 0 .. L-1), prompt s owns rows
// [seq_row0[s], seq_row0[s + 1]) = its positions L .. m_s + its CLS row.  Block (s, h) then attends with the keys [prefix rows |
// own rows]; one more block per head (s == n_seq) serves the prefix rows themselves (causal among themselves).
constexpr int kAttnMaxS = 128;
__global__ __launch_bounds__(1024) void k_tt_attn_fwd(const float* __restrict__ qkv, int ld, float* __restrict__ out_t,
                                                    const int* __restrict__ seq_row0, const unsigned char* __restrict__ cls_keep,
                                                    int heads, int d, int n_seq, int L, float* __restrict__ stats) {
    // stats (nullable; the forward that saves for the backward pass): [row][head][2] = (max of the row's scaled scores, 1 / sum of the
    // exponentials) -- with them and the kept output the backward can form the prefix keys' dK / dV without a sum across workgroups
    __shared__ float Ks[kAttnMaxS][kHeadDim + 1];
    __shared__ float Vs[kAttnMaxS][kHeadDim + 1];
    __shared__ float Qs[kAttnMaxS][kHeadDim];
    const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
    const bool pfx_block = seq == n_seq;                       // only launched when L > 0
    const int r0 = pfx_block ? 0 : seq_row0[seq];
    const int S = pfx_block ? L : L + seq_row0[seq + 1] - r0;   // keys: [prefix | own rows]
    const int qbeg = pfx_block ? 0 : L;                        // first query (key index)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nthr = blockDim.x, nwave = nthr >> 6;       // round 4: one wave per query row (16 waves) instead of four rows per wave
    auto grow = [&](int j) { return (pfx_block || j < L) ? j : r0 + (j - L); };     // key / query index -> compact row
    for (int e = tid; e < S * kHeadDim; e += nthr) {   // q, k, v of the prompt in ONE round of global loads
        const int j = e >> 6, c = e & 63;
        const size_t base = (size_t)grow(j) * ld + h * kHeadDim + c;
        const float qv = qkv[base], kv = qkv[base + d], vv = qkv[base + 2 * d];
        Qs[j][c] = qv * 0.125f;   // head_dim^-0.5
        Ks[j][c] = kv;
        Vs[j][c] = vv;
    }
    __syncthreads();
    for (int i = qbeg + w; i < S; i += nwave) {
        const float q = Qs[i][lane];
        const bool is_cls = !pfx_block && i == S - 1;
        float s[2] = {-INFINITY, -INFINITY}, p[2];
        const int halves = S > 64 ? 2 : 1;                 // prompts are short: the second key chunk is rarely needed (uniform)
        for (int half = 0; half < halves; ++half) {
            const int j = lane + 64 * half;
            const int jc = j < S ? j : S - 1;          // every lane computes (no cross-lane reads under a divergent branch)
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < kHeadDim; ++c)
                dot = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(q), c)), Ks[jc][c], dot);
            const bool ok = j < S && (is_cls ? cls_keep[grow(jc)] != 0 : j <= i);
            if (half == 0) s[0] = ok ? dot : -INFINITY; else s[1] = ok ? dot : -INFINITY;
        }
        const float m = wave_max(fmaxf(s[0], s[1]));
        p[0] = s[0] == -INFINITY ? 0.f : __expf(s[0] - m);
        p[1] = s[1] == -INFINITY ? 0.f : __expf(s[1] - m);
        const float inv = 1.f / wave_sum(p[0] + p[1]);
        p[0] *= inv;
        p[1] *= inv;
        if (stats != nullptr && lane == 0) *reinterpret_cast<float2*>(stats + ((size_t)grow(i) * heads + h) * 2) = float2{m, inv};
        float o = 0.f;
        const int S0 = S < 64 ? S : 64;
        for (int j = 0; j < S0; ++j)
            o = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[0]), __builtin_amdgcn_readfirstlane(j))), Vs[j][lane], o);
        for (int j = 64; j < S; ++j)
            o = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[1]), __builtin_amdgcn_readfirstlane(j - 64))), Vs[j][lane], o);
        out_t[tiled_index(grow(i), h * kHeadDim + lane, d)] = o;    // tiled [M_pad, d]: the A operand of the out_proj product
    }
}

// Backward.  With a shared prefix the dK / dV of the prefix rows get a contribution from EVERY block of a head (the n_seq prompts +
// the prefix block): each block publishes its partial [L][2][64] (write-through stores), takes a ticket on the head's counter, and
// the last arriver adds the n_seq + 1 partials in block order (deterministic) into dqkv.  pfx: [(n_seq + 1)][heads][L][128] floats,
// cnt: [heads] zeroed unsigned ints (handed back zeroed).
constexpr int kAttnBwdMaxS = 64;
__global__ __launch_bounds__(1024) void k_tt_attn_bwd(const float* __restrict__ qkv, int ld, const float* __restrict__ dout, int ldo,
                                                    float* __restrict__ dqkv, const int* __restrict__ seq_row0,
                                                    const unsigned char* __restrict__ cls_keep, int heads, int d, int n_seq, int L,
                                                    float* __restrict__ pfx, unsigned int* __restrict__ cnt,
                                                    const float* __restrict__ O_t, const float* __restrict__ stats,
                                                    const int* __restrict__ row_seq, int M, int use_stats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ bool s_last;
#ifdef VLSA_EXPERIMENT
    const int abl = ldo >> 16;      // timing-only ablations (tools/attn_bwd_ablate.sh): 1 = no prefix fold, 2 = no phase 2, 4 = no phase 1, 8 = loads only, 16 / 32 = key workgroups: nothing / loads only
    ldo &= 0xffff;
#else
    constexpr int abl = 0;
#endif
    constexpr int LD = kHeadDim + 1;
    if (use_stats && (int)blockIdx.x >= (n_seq + 1) * heads) {
        // ---- round 6: a workgroup per (prefix key jk, head): dK / dV of the key over ALL M query rows, from the forward's row statistics
        // (m, 1 / l) and its kept output O (delta_i = dO_i . O_i) -- no shares, no ticket, no fold.  M <= 128 rows (host).
        //   A   lane = query row (two halves of 64), wave = (half, slice of the 64 features): partial q.k, dO.v, dO.O      -> LDS
        //   A2  two waves, lane = row: the slices summed in order, weight p = exp(s - m) / l under the row's mask, d score    -> LDS
        //   B   lane = feature, the rows dealt over the waves: dK += d score * q, dV += p * dO                               -> LDS
        //   C   two waves add the waves' sums in wave order and store.
        // (First version: lane = feature throughout, three lane sums per row -- 80 VALU instructions per row on every wave, 3.3 us of
        //  issue time with 3-4 waves per SIMD; profiles/r06_attn_bwd_ablate2.txt.)
        // use_stats = keys per workgroup (1 .. 4: the host keeps the grid within one round of the CUs; the q / dO / O rows are loaded once)
        const int idx = (int)blockIdx.x - (n_seq + 1) * heads, hh = idx % heads, jk0 = (idx / heads) * use_stats;
        const int nk = L - jk0 < use_stats ? L - jk0 : use_stats;
        if (abl & 16) return;
        const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
        constexpr int MR = 128;
        float* Qa = sm;                       // [MR][LD] q, dO, O of the head
        float* Da = Qa + MR * LD;
        float* Oa = Da + MR * LD;
        float* part = Oa + MR * LD;           // A: [8 slices][3][MR];  B / C: the waves' sums [16][128]
        float* st = part + 8 * 3 * MR;        // [MR][2] m, 1 / l
        float* vis = st + 2 * MR;             // [4][MR] the row sees the key
        float* Pd = vis + 4 * MR;             // [MR] p, [MR] d score
        float* kv = Pd + 2 * MR;              // [4][128] k_jk | v_jk
        for (int e = tid; e < M * kHeadDim; e += nthr) {
            const int r = e >> 6, c = e & 63;
            Qa[r * LD + c] = qkv[(size_t)r * ld + hh * kHeadDim + c];
            Da[r * LD + c] = dout[(size_t)r * ldo + hh * kHeadDim + c];
            Oa[r * LD + c] = O_t[tiled_index(r, hh * kHeadDim + c, d)];
        }
        for (int r = tid; r < M; r += nthr) {
            const float2 ml = *reinterpret_cast<const float2*>(stats + ((size_t)r * heads + hh) * 2);
            st[2 * r] = ml.x;
            st[2 * r + 1] = ml.y;
            const bool cls_row = r >= L && r == seq_row0[row_seq[r] + 1] - 1;
            for (int kk = 0; kk < nk; ++kk) {
                const int jk = jk0 + kk;
                // the prefix rows among themselves: causal; a prompt's CLS row sees the flagged rows; its tokens see every prefix key
                const bool ok = r < L ? jk <= r : (cls_row ? cls_keep[jk] != 0 : true);
                vis[kk * MR + r] = ok ? 1.f : 0.f;
            }
        }
        for (int e = tid; e < nk * 128; e += nthr) {
            const int kk = e >> 7, t = e & 127;
            kv[e] = qkv[(size_t)(jk0 + kk) * ld + (t < 64 ? d : 2 * d) + hh * kHeadDim + (t & 63)];
        }
        __syncthreads();
        if (abl & 32) return;
        const int ncs = nwave >> 1 < 8 ? nwave >> 1 : 8;      // feature slices (13 waves: 6)
        for (int kk = 0; kk < nk; ++kk) {
        const int jk = jk0 + kk;
        const float* kvk = kv + kk * 128;
        const float* visk = vis + kk * MR;
        if (kk) __syncthreads();              // (the previous key's phase C has read `part`)
        {   // A
            const int rh = w & 1, cs = w >> 1, r = 64 * rh + lane;
            if (cs < ncs) {
                const int c0 = cs * kHeadDim / ncs, c1 = (cs + 1) * kHeadDim / ncs;
                float sa = 0.f, pa = 0.f, ea = 0.f;
                if (r < M) {
                    for (int c = c0; c < c1; ++c) {
                        const float q = Qa[r * LD + c], g = Da[r * LD + c], o = Oa[r * LD + c];
                        sa = fmaf(q, kvk[c], sa);
                        pa = fmaf(g, kvk[64 + c], pa);
                        ea = fmaf(g, o, ea);
                    }
                }
                part[(cs * 3 + 0) * MR + r] = sa;
                part[(cs * 3 + 1) * MR + r] = pa;
                part[(cs * 3 + 2) * MR + r] = ea;
            }
        }
        __syncthreads();
        if (w < 2) {   // A2
            const int r = 64 * w + lane;
            float sa = 0.f, pa = 0.f, ea = 0.f;
            for (int cs = 0; cs < ncs; ++cs) {
                sa += part[(cs * 3 + 0) * MR + r];
                pa += part[(cs * 3 + 1) * MR + r];
                ea += part[(cs * 3 + 2) * MR + r];
            }
            const bool ok = r < M && visk[r] != 0.f;
            const float pw = ok ? __expf(sa * 0.125f - st[2 * r]) * st[2 * r + 1] : 0.f;
            Pd[r] = pw;
            Pd[MR + r] = pw * (pa - ea) * 0.125f;
        }
        __syncthreads();
        float dk = 0.f, dv = 0.f;
        for (int r = w; r < M; r += nwave) {   // B
            dk = fmaf(Pd[MR + r], Qa[r * LD + lane], dk);
            dv = fmaf(Pd[r], Da[r * LD + lane], dv);
        }
        part[w * 128 + lane] = dk;
        part[w * 128 + 64 + lane] = dv;
        __syncthreads();
        if (w < 2) {   // C
            float t[16];
#pragma unroll
            for (int ww = 0; ww < 16; ++ww) t[ww] = ww < nwave ? part[ww * 128 + 64 * w + lane] : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int ww = 0; ww < 16; ++ww) acc += t[ww];
            dqkv[tiled_index(jk, hh * kHeadDim + lane + (w ? 2 * d : d), 3 * d)] = acc;
        }
        }   // kk
        return;
    }
    float* Qs = sm;
    float* Ks = Qs + kAttnBwdMaxS * LD;
    float* Vs = Ks + kAttnBwdMaxS * LD;
    float* Os = Vs + kAttnBwdMaxS * LD;      // dO
    float* Pm = Os + kAttnBwdMaxS * LD;      // softmax weights [i][j]
    float* Dm = Pm + kAttnBwdMaxS * LD;      // d scores (scale folded in) [i][j]
    const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
    const bool pfx_block = seq == n_seq;
    const int r0 = pfx_block ? 0 : seq_row0[seq];
    const int S = pfx_block ? L : L + seq_row0[seq + 1] - r0;
    const int qbeg = pfx_block ? 0 : L;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nthr = blockDim.x, nwave = nthr >> 6;       // round 4: 16 waves -- a wave per row in both phases
    auto grow = [&](int j) { return (pfx_block || j < L) ? j : r0 + (j - L); };
    for (int e = tid; e < S * kHeadDim; e += nthr) {
        const int j = e >> 6, c = e & 63;
        const size_t base = (size_t)grow(j) * ld + h * kHeadDim + c;
        Qs[j * LD + c] = qkv[base];
        Ks[j * LD + c] = qkv[base + d];
        Vs[j * LD + c] = qkv[base + 2 * d];
        Os[j * LD + c] = dout[(size_t)grow(j) * ldo + h * kHeadDim + c];
    }
    __syncthreads();
    if (abl & 8) return;
    for (int i = (abl & 4) ? S : w; i < S; i += nwave) {   // lane j: score, weight and their gradients for key j of query row i
        const int j = lane;
        if (i < qbeg) {                // prefix rows are queries of the prefix block only
            Pm[i * LD + j] = 0.f;
            Dm[i * LD + j] = 0.f;
            continue;
        }
        const bool is_cls = !pfx_block && i == S - 1;
        float dot = 0.f, dp = 0.f;
        if (j < S) {
#pragma unroll
            for (int c = 0; c < kHeadDim; ++c) {
                dot = fmaf(Qs[i * LD + c], Ks[j * LD + c], dot);
                dp = fmaf(Os[i * LD + c], Vs[j * LD + c], dp);
            }
        }
        const bool ok = j < S && (is_cls ? cls_keep[grow(j < S ? j : 0)] != 0 : j <= i);
        const float s = ok ? dot * 0.125f : -INFINITY;
        const float m = wave_max(s);
        float p = ok ? __expf(s - m) : 0.f;
        p *= 1.f / wave_sum(p);
        const float delta = wave_sum(p * dp);
        Pm[i * LD + j] = p;
        Dm[i * LD + j] = p * (dp - delta) * 0.125f;
    }
    __syncthreads();
    const bool shared_keys = L > 0 && !(abl & 1);
    for (int rr = (abl & 2) ? S : w; rr < S; rr += nwave) {   // lane = feature c of row rr: dQ, dK, dV
        if (use_stats && rr < L && !pfx_block) continue;    // (nothing of a prefix row is this block's to write: its dQ is the prefix block's, its dK / dV the key blocks')
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < S; ++j) {
            dq = fmaf(Dm[rr * LD + j], Ks[j * LD + lane], dq);
            dk = fmaf(Dm[j * LD + rr], Qs[j * LD + lane], dk);
            dv = fmaf(Pm[j * LD + rr], Os[j * LD + lane], dv);
        }
        const int col = h * kHeadDim + lane;       // tiled [M_pad, 3 d]: the A operand of the in_proj^T product
        const bool pfx_row = shared_keys && rr < L;
        if (!pfx_row || pfx_block) dqkv[tiled_index(grow(rr), col, 3 * d)] = dq;
        if (!pfx_row) {
            dqkv[tiled_index(grow(rr), col + d, 3 * d)] = dk;
            dqkv[tiled_index(grow(rr), col + 2 * d, 3 * d)] = dv;
        } else if (!use_stats) {                   // this block's share of a prefix row's dK / dV: published write-through
            float* pp = pfx + (((size_t)seq * heads + h) * L + rr) * 128;
            __hip_atomic_store(pp + lane, dk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 64 + lane, dv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!shared_keys || use_stats) return;
    // (Round 6, timing-only ablations, profiles/r06_attn_bwd_ablate.txt: of this launch's 10.8 us the fold below is 3.3 -- three dependent
    //  trips through the memory fabric: drain of the write-through stores, the ticket, the shares' loads --, phase 2 above 2.1, phase 1
    //  0.7, the loads 4.5 with the launch itself.  Tried and not kept: the shares first and the ticket taken before the rest of phase 2
    //  (the drain then stalls the whole workgroup in the middle: 3.1), phase 2's LDS reads batched (2.1 -> 2.1), phase 1 on
    //  (feature quarter, key) lanes (no change: it is 0.7 us).)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) {
        const unsigned int t = __hip_atomic_fetch_add(cnt + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == (unsigned int)n_seq;                 // n_seq + 1 blocks per head
    }
    __syncthreads();
    if (!s_last) return;
    // (row, dK | dV) pairs over the four waves; the blocks' shares of a pair are loaded 16 at a time (one L2 round trip per 16 blocks,
    // not one per block: the loop is a chain of write-through-visible loads) and added in block order
    // (row, dK | dV) pairs over the waves -- with a wave per row of the prompt (round 4: up to 16 waves) every pair of the L <= 8
    // prefix rows has a wave of its own, so the blocks' shares of all pairs come in with ONE round of L2 round trips; the shares of
    // a pair are loaded 16 at a time and added in block order
    for (int pr = w; pr < 2 * L; pr += nwave) {
        const int rr = pr >> 1, which = pr & 1;
        float acc = 0.f;
        for (int b0 = 0; b0 <= n_seq; b0 += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int b = b0 + u;
                t[u] = b <= n_seq ? __hip_atomic_load(pfx + (((size_t)b * heads + h) * L + rr) * 128 + 64 * which + lane, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT)
                                  : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += t[u];      // fixed order (the padding adds exact zeros)
        }
        dqkv[tiled_index(rr, h * kHeadDim + lane + (which ? 2 * d : d), 3 * d)] = acc;
    }
    if (tid == 0) __hip_atomic_store(cnt + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ticket back to zero
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm helpers: one wave per row, d <= 1024 (d % 64 == 0): 16 register slots per lane.
constexpr int kLnSlots = 16;
__device__ __forceinline__ void ln_stats(const float (&v)[kLnSlots], int nslot, int d, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) s += v[k];
    mean = wave_sum(s) / (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            const float c = v[k] - mean;
            s2 = fmaf(c, c, s2);
        }
    rstd = 1.f / sqrtf(wave_sum(s2) / (float)d + kLnEps);
}

// dx[row] = dres[row] + LayerNorm'(x[row]; gamma)^T da[row]   (dres nullable).  Round 4: 16-byte accesses -- a lane owns the four
// consecutive columns 4 lane + 256 k (contiguous in the row-major arrays AND in a 16 x 16 tile of the tiled copy): 3 loads per array
// and 2 x 3 stores per lane at d = 768 instead of 12 and 24 four-byte ones; d % 256 == 0 (else the 4-byte path below).
constexpr int kLnSlots4 = 4;
// With a scatter target (block 0's ln_1 backward of the frozen tower: the pass's LAST LayerNorm backward) the same launch also produces
// d prompts_embedding, so the memset and k_tt_scatter launches behind it are gone: row r < sc.M with a source token writes its dx to
// demb[row_seq[r], row_src[r], :], and the workgroups behind the LayerNorm rows zero every [seq, token] row of demb (contiguous
// [n, ctx_len, d]) that no compact row writes to (a wave per row: the <= 128 compact rows are searched, 64 per step).
struct ScatterArgs {
    float* demb;            // null: no scatter
    const int* row_seq;
    const int* row_src;
    int64_t s_seq;
    int M, ctx_len, demb_rows, ln_blocks;
};
__global__ __launch_bounds__(256) void k_tt_ln_bwd4(const float* __restrict__ da, const float* __restrict__ x,
                                                   const float* __restrict__ gamma, const float* __restrict__ dres,
                                                   float* __restrict__ dx, float* __restrict__ dxt, int d, int rows, const ScatterArgs sc) {
    const int lane = threadIdx.x & 63;
    const int nslot = d >> 8;
    if (sc.demb != nullptr && (int)blockIdx.x >= sc.ln_blocks) {
        const int R = ((int)blockIdx.x - sc.ln_blocks) * 4 + (threadIdx.x >> 6);
        if (R >= sc.demb_rows) return;
        const int seq = R / sc.ctx_len, tok = R - seq * sc.ctx_len;
        bool hit = false;
        for (int m0 = 0; m0 < sc.M; m0 += 64) {
            const int m = m0 + lane;
            hit = hit || (m < sc.M && sc.row_seq[m] == seq && sc.row_src[m] == tok);
        }
        if (__ballot(hit) != 0ull) return;
        float* o = sc.demb + (size_t)seq * sc.s_seq + (size_t)tok * d;
        for (int k = 0; k < nslot; ++k) *reinterpret_cast<f32x4*>(o + 4 * lane + 256 * k) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 xv[kLnSlots4], gv[kLnSlots4], rv[kLnSlots4];
#pragma unroll
    for (int k = 0; k < kLnSlots4; ++k)
        if (k < nslot) {   // every load of the row in one round
            const int c = 4 * lane + 256 * k;
            xv[k] = *reinterpret_cast<const f32x4*>(x + (size_t)row * d + c);
            gv[k] = *reinterpret_cast<const f32x4*>(da + (size_t)row * d + c) * *reinterpret_cast<const f32x4*>(gamma + c);
            rv[k] = dres ? *reinterpret_cast<const f32x4*>(dres + (size_t)row * d + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots4; ++k)
        if (k < nslot) s += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
    const float mean = wave_sum(s) / (float)d;
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots4; ++k)
        if (k < nslot)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float c = xv[k][i] - mean;
                s2 = fmaf(c, c, s2);
            }
    const float rstd = 1.f / sqrtf(wave_sum(s2) / (float)d + kLnEps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots4; ++k)
        if (k < nslot)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xv[k][i] = (xv[k][i] - mean) * rstd;
                sg += gv[k][i];
                sgx = fmaf(gv[k][i], xv[k][i], sgx);
            }
    sg = wave_sum(sg) / (float)d;
    sgx = wave_sum(sgx) / (float)d;
#pragma unroll
    for (int k = 0; k < kLnSlots4; ++k)
        if (k < nslot) {
            const int c = 4 * lane + 256 * k;
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = rv[k][i] + rstd * (gv[k][i] - sg - xv[k][i] * sgx);
            *reinterpret_cast<f32x4*>(dx + (size_t)row * d + c) = o;
            *reinterpret_cast<f32x4*>(dxt + tiled_index(row, c, d)) = o;      // columns c .. c + 3: one k-slot of the tile, contiguous
            if (sc.demb != nullptr && row < sc.M) {
                const int src = sc.row_src[row];
                if (src >= 0) *reinterpret_cast<f32x4*>(sc.demb + (size_t)sc.row_seq[row] * sc.s_seq + (size_t)src * d + c) = o;
            }
        }
}

// dx[row] = dres[row] + LayerNorm'(x[row]; gamma)^T da[row]   (dres nullable)
__global__ __launch_bounds__(256) void k_tt_ln_bwd(const float* __restrict__ da, const float* __restrict__ x,
                                                  const float* __restrict__ gamma, const float* __restrict__ dres,
                                                  float* __restrict__ dx, float* __restrict__ dxt, int d, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nslot = d >> 6;
    float xv[kLnSlots], gv[kLnSlots], rv[kLnSlots];
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {   // every load of the row in one round
            xv[k] = x[(size_t)row * d + lane + 64 * k];
            gv[k] = da[(size_t)row * d + lane + 64 * k] * gamma[lane + 64 * k];
            rv[k] = dres ? dres[(size_t)row * d + lane + 64 * k] : 0.f;
        }
    float mean, rstd;
    ln_stats(xv, nslot, d, mean, rstd);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            xv[k] = (xv[k] - mean) * rstd;
            sg += gv[k];
            sgx = fmaf(gv[k], xv[k], sgx);
        }
    sg = wave_sum(sg) / (float)d;
    sgx = wave_sum(sgx) / (float)d;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            const float o = rv[k] + rstd * (gv[k] - sg - xv[k] * sgx);
            dx[(size_t)row * d + lane + 64 * k] = o;
            dxt[tiled_index(row, lane + 64 * k, d)] = o;
        }
}

// pooled (TILED [n_pad, d]) [s] = ln_final(x[CLS row of prompt s]); rows n_seq .. n_pad-1 are zeroed (GEMM padding).
__global__ __launch_bounds__(256) void k_tt_lnf_fwd(const float* __restrict__ x, const int* __restrict__ seq_row0,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ pooled, int d, int n_seq, int n_pad) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_pad) return;
    const int nslot = d >> 6;
    if (s >= n_seq) {
        for (int k = 0; k < nslot; ++k) pooled[tiled_index(s, lane + 64 * k, d)] = 0.f;
        return;
    }
    // (the affine parameters are requested together with the row index: behind the statistics they were a third dependent round trip)
    float gv[kLnSlots], bv[kLnSlots];
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            gv[k] = gamma[lane + 64 * k];
            bv[k] = beta[lane + 64 * k];
        }
    const int row = seq_row0[s + 1] - 1;
    float xv[kLnSlots];
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) xv[k] = x[(size_t)row * d + lane + 64 * k];
    float mean, rstd;
    ln_stats(xv, nslot, d, mean, rstd);
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) pooled[tiled_index(s, lane + 64 * k, d)] = fmaf((xv[k] - mean) * rstd, gv[k], bv[k]);
}

// dx[row] = ln_final backward of dpooled[s] on the CLS row of prompt s, zero on every other row.
__global__ __launch_bounds__(256) void k_tt_lnf_bwd(const float* __restrict__ dpooled, const float* __restrict__ x,
                                                   const int* __restrict__ row_seq, const int* __restrict__ row_src,
                                                   const float* __restrict__ gamma, float* __restrict__ dx, float* __restrict__ dxt,
                                                   int d, int M, int M_pad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M_pad) return;
    const int nslot = d >> 6;
    // (one round of requests for everything that does not hang on another load: the row's source flag and prompt, its x, gamma; only
    //  the prompt's d pooled row follows in a second round -- it was flag -> prompt -> rows, three dependent trips)
    const int rsrc = row < M ? row_src[row] : 0, s = row < M ? row_seq[row] : 0;
    float xv[kLnSlots], gv[kLnSlots];
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            xv[k] = x[(size_t)row * d + lane + 64 * k];
            gv[k] = gamma[lane + 64 * k];
        }
    if (row >= M || rsrc >= 0) {
        for (int k = 0; k < nslot; ++k) {
            dx[(size_t)row * d + lane + 64 * k] = 0.f;
            dxt[tiled_index(row, lane + 64 * k, d)] = 0.f;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) gv[k] *= dpooled[(size_t)s * d + lane + 64 * k];
    float mean, rstd;
    ln_stats(xv, nslot, d, mean, rstd);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            xv[k] = (xv[k] - mean) * rstd;
            sg += gv[k];
            sgx = fmaf(gv[k], xv[k], sgx);
        }
    sg = wave_sum(sg) / (float)d;
    sgx = wave_sum(sgx) / (float)d;
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) {
            const float o = rstd * (gv[k] - sg - xv[k] * sgx);
            dx[(size_t)row * d + lane + 64 * k] = o;
            dxt[tiled_index(row, lane + 64 * k, d)] = o;
        }
}

// d prompts_embedding[seq, src] = dx[row] for the token rows (the CLS rows' gradient belongs to the frozen cls_emb).
__global__ __launch_bounds__(256) void k_tt_scatter(const float* __restrict__ dx, int d, float* __restrict__ demb, int64_t s_seq,
                                                   int64_t s_tok, const int* __restrict__ row_seq, const int* __restrict__ row_src,
                                                   int M) {
    const int row = blockIdx.x;
    if (row >= M) return;
    const int src = row_src[row];
    if (src < 0) return;
    float* o = demb + (size_t)row_seq[row] * s_seq + (size_t)src * s_tok;
    for (int c = threadIdx.x; c < d; c += 256) o[c] = dx[(size_t)row * d + c];
}


// ===============================================================================================================
// Weight gradients: a tower whose OWN parameters train (`vlsa_txt_encoder_frozen: False`, runner/vlsa_handler.py:131; off in every
// shipped configuration).  Until round 4 that case ran as torch library GEMMs over the compact rows (14.4 ms forward + backward for
// K = 12 prompts); now the activations the input-gradient pass already has in the workspace feed four more products per block:
//   dW[o, i] = sum_m dY[m, o] act[m, i]      for in_proj / out_proj / c_fc / c_proj (and text_projection),
// i.e. a contraction over the <= 128 compact rows -- the M index is the MFMA k index, so BOTH operands are read as they lie (dY tiled
// [M_pad, N]: every producer of the pass writes a tiled copy; act row-major or tiled), no transposes.  v_mfma_f32_16x16x4_f32 as
// everywhere in the tower (fp32 in, fp32 accumulate: bit-equal to an fmaf chain); k-slot g of step s of row block mb contracts row
// 16 mb + 4 s + g.  Workgroup = 4 waves = 128 outputs x 64 inputs, a wave 32 x 64 (8 accumulator tiles): per step 2 + 4 operand
// dwords for 8 MFMAs.  Rows >= M are padding that may hold anything: both operands are masked to zero there.
// The bias gradient (column sums of dY) falls out of the first operand in the workgroups of input tile 0.
enum { DW_ROWS = 0, DW_TILED = 2 };
struct DwArgs {
    const float* A;   // tiled [M_pad, N]: gradient w.r.t. the product's output
    const float* B;   // the product's input: row-major [.., ldb] (DW_ROWS) or tiled [M_pad, K] (DW_TILED)
    float* dW;        // row-major [N, K]
    float* dbias;     // [N] or null
    int N, K, M, ldb;
};
template <int BSRC>
__global__ __launch_bounds__(256) void k_tt_dw(const DwArgs p) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int o0 = blockIdx.y * 128 + w * 32, j0 = blockIdx.x * 64;
    const int NT = p.N >> 4, KT = p.K >> 4;
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float asum[2] = {0.f, 0.f};
    const int in_tile = ((i >> 2) * 16) * 4 + (i & 3);       // tiled operand: element (row, col 16 c + i) at tile + in_tile + 4 (row & 15)
    const int nmb = (p.M + 15) >> 4;
    // The 24 operand dwords of a row block are loaded one block AHEAD of their MFMAs (two register sets).  Every load is
    // unconditional -- rows >= M read row M - 1 instead and the FIRST operand is zeroed by a select (0 x finite = 0) -- so the loop
    // body is straight-line code: with per-load branches (or `if (mb + 1 < nmb)` around the prefetch) every basic-block edge became an
    // `s_waitcnt vmcnt(0)` and the kernel ran at 25-30 us for 1 us of matrix-pipe work.
    float a[2][4][2], b[2][4][4];
    auto load = [&](int mb, float (&av)[4][2], float (&bv)[4][4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = 16 * mb + 4 * s + g;
            const int mc = m < p.M ? m : p.M - 1, mcb = mc >> 4, mcr = mc & 15;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float v = p.A[((size_t)mcb * NT + ((o0 >> 4) + t)) * 256 + in_tile + 4 * mcr];
                av[s][t] = m < p.M ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                bv[s][u] = BSRC == DW_TILED ? p.B[((size_t)mcb * KT + ((j0 >> 4) + u)) * 256 + in_tile + 4 * mcr]
                                            : p.B[(size_t)mc * p.ldb + j0 + 16 * u + i];
        }
    };
    auto mac = [&](const float (&av)[4][2], const float (&bv)[4][4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                asum[t] += av[s][t];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], bv[s][u], acc[t][u], 0, 0, 0);
            }
        }
    };
    load(0, a[0], b[0]);
    for (int mb = 0; mb < nmb; mb += 2) {       // (a block index past the end loads row M - 1 and contributes zeros)
        load(mb + 1, a[1], b[1]);
        mac(a[0], b[0]);
        load(mb + 2, a[0], b[0]);
        mac(a[1], b[1]);
    }
    // result lane (j = i, g) holds D[out 4 g + v][in j]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) p.dW[(size_t)(o0 + 16 * t + 4 * g + v) * p.K + j0 + 16 * u + i] = acc[t][u][v];
    if (p.dbias && blockIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float s = asum[t];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (g == 0) p.dbias[o0 + 16 * t + i] = s;
        }
    }
}

// y[row] = LayerNorm(x[row]) * gamma + beta (row-major), stats[row] = (mean, rstd): the input of a product whose weight gradient is
// wanted (the forward's fused-LayerNorm products never store it) and the statistics the gamma / beta gradients need.  Wave per row.
__global__ __launch_bounds__(256) void k_tt_ln_rows(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ stats,
                                                   int d, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nslot = d >> 6;
    float xv[kLnSlots];
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot) xv[k] = x[(size_t)row * d + lane + 64 * k];
    float mean, rstd;
    ln_stats(xv, nslot, d, mean, rstd);
#pragma unroll
    for (int k = 0; k < kLnSlots; ++k)
        if (k < nslot && y) y[(size_t)row * d + lane + 64 * k] = fmaf((xv[k] - mean) * rstd, gamma[lane + 64 * k], beta[lane + 64 * k]);
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// y = gelu(h) over the first `rows` rows of a row-major [.., C] matrix: the input of c_proj, which the forward only keeps as h_pre
// (evaluating erf inside the weight-gradient product would repeat it 24 times per element: 37 us per block instead of 13 + 3)
__global__ __launch_bounds__(256) void k_tt_gelu_rows(const float* __restrict__ h, float* __restrict__ y, size_t n4) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(h)[idx];
    reinterpret_cast<f32x4*>(y)[idx] = f32x4{gelu(v[0]), gelu(v[1]), gelu(v[2]), gelu(v[3])};
}

// dgamma[c] = sum_n dy[n, c] xhat[row(n), c], dbeta[c] = sum_n dy[n, c] over n < count; row(n) = n, or the CLS row of prompt n
// (seq_row0[n + 1] - 1) for ln_final.  Workgroup = 32 columns x 8 row groups (row n -> group n % 8), partial sums folded through LDS in
// a fixed order (deterministic).  (The first version -- a thread per column walking all rows -- took 29 us per call.)
__global__ __launch_bounds__(256) void k_tt_ln_param_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ stats, const int* __restrict__ seq_row0, int count,
                                                        int d, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[2][8][32];
    const int cl = threadIdx.x & 31, rgp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float sg = 0.f, sb = 0.f;
    if (c < d)
        for (int n = rgp; n < count; n += 8) {
            const int row = seq_row0 ? seq_row0[n + 1] - 1 : n;
            const float g = dy[(size_t)n * d + c];
            sg = fmaf(g, (x[(size_t)row * d + c] - stats[2 * row]) * stats[2 * row + 1], sg);
            sb += g;
        }
    red[0][rgp][cl] = sg;
    red[1][rgp][cl] = sb;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int which = threadIdx.x >> 5;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[which][k][cl];
        if (c < d) (which ? dbeta : dgamma)[c] = t;
    }
}

// d positional_embedding[p] = sum of dx0 over the compact rows at position p (block p < ctx_len; positions no row uses get zeros),
// d cls_emb = sum over the CLS rows (block ctx_len).  model/prompt_encoder.py:283-292.  The matching rows are listed first (all
// threads test rows in parallel, wave 0 compacts them in row order), then summed in that order.
constexpr int kPosRowsMax = 1024;
__global__ __launch_bounds__(256) void k_tt_pos_cls_bwd(const float* __restrict__ dx, const int* __restrict__ row_pos,
                                                       const int* __restrict__ row_src, int M, int d, int ctx_len,
                                                       float* __restrict__ dpos, float* __restrict__ dcls) {
    __shared__ unsigned char hit[kPosRowsMax];
    __shared__ short rows[kPosRowsMax];
    __shared__ int nrows;
    const int p = blockIdx.x;
    for (int m = threadIdx.x; m < M; m += 256) hit[m] = (p < ctx_len ? row_pos[m] == p : row_src[m] < 0) ? 1 : 0;
    __syncthreads();
    if (threadIdx.x < 64) {
        int n = 0;
        for (int base = 0; base < M; base += 64) {
            const int m = base + threadIdx.x;
            const bool h = m < M && hit[m];
            const unsigned long long mask = __ballot(h);
            if (h) rows[n + __popcll(mask & ((1ull << threadIdx.x) - 1ull))] = (short)m;
            n += __popcll(mask);
        }
        if (threadIdx.x == 0) nrows = n;
    }
    __syncthreads();
    float* out = p < ctx_len ? dpos + (size_t)p * d : dcls;
    const int n = nrows;
    for (int c = threadIdx.x; c < d; c += 256) {
        float s = 0.f;
        for (int k = 0; k < n; ++k) s += dx[(size_t)rows[k] * d + c];
        out[c] = s;
    }
}

// ===============================================================================================================
// Persistent forward (round 4): the 12 blocks of the tower as ONE launch.
//
// Why: a forward pass over K = 12 rank prompts is 101 compact rows = 7 row tiles; every product of a block is 5 - 13 us of which
// ~3 us is the launch boundary and ~1 us the weight loads that could have been in flight before the activations existed
// (profiles/r03_text_kernel_stats.csv: 60 dependent launches = 616 us for 42 us of weight stream).  Here every workgroup walks
// the same static schedule -- per block: QKV product, attention, out-proj, c_fc, c_proj, one task of <= 16 rows x 32 .. 96
// columns per workgroup and stage -- and the stages are ordered by DATAFLOW, not by grid-wide barriers:
//   * every task publishes its outputs write-through (agent-scope `sc1` stores), drains them, and adds 1 to the counter of the
//     row tile(s) it wrote (one counter per block x stage x row tile: <= 36 arrivals each, not 256 on one word);
//   * a task first issues the loads of its WEIGHT fragments (they depend on nothing), then ONE lane polls the counter(s) of the
//     row tile it consumes (relaxed agent-scope loads + s_sleep), a workgroup barrier, then the activation loads -- `sc1` loads,
//     which bypass the CU's L1 (the buffers are rewritten every block; L1 is never refreshed by another CU's stores) and are
//     served memory-side of the per-XCD L2s (MI355X_MICROARCH.md "inter-workgroup visibility": {sc1 stores, sc1 loads} is a valid
//     hand-off at any placement).  Row tile r of block L+1 can start while other tiles still finish block L.
//   * deadlock freedom: a task only waits for tasks of EARLIER stages, every workgroup runs its tasks in stage order, and the
//     grid (<= 256 workgroups of 512 threads) is resident at once; every spin is bounded and a time-out is reported in the
//     workspace's status block (vlsa_tt_status_offset) -- the results of that launch are then void.
// Buffers reused across blocks (x / qkv / attention / hidden activations) are safe by the dependency chain itself; the one
// exception, the prefix rows of `qkv` (read by every prompt's attention task, rewritten by the next block's tile 0), is double
// buffered by block parity.  Supported: width 768 / 12 heads (CONCH), <= 16 blocks, <= 112 compact rows, <= 64 keys per prompt.
// RESULT (round 4): correct on the first run (tests/test_gpu_text_tower.py::test_persistent_forward_equals_...), and SLOWER than the
// launch-per-stage path: 836 vs 616 us -- see persist_supported() and DESIGN.md 4.8; it is therefore opt-in (VLSA_TT_PERSIST=1).
constexpr int kPMaxLayers = 16, kPMaxRT = 8, kPStages = 5;
constexpr int kPAttnMaxS = 64;
constexpr unsigned kPSpinLimit = 1u << 21;      // ~1 s of polling: only a non-resident workgroup (a foreign kernel hogging CUs) gets here

struct PLayerPtrs {
    const float *ln1_w, *ln1_b, *in_b, *out_b, *ln2_w, *ln2_b, *fc_b, *proj_b;
};
struct PArgs {
    float* ws;                 // layer regions
    const float* wset;         // packed (tiled) forward weights
    float *x_final, *xin_t, *xmid_t, *attn_t, *hact_t, *qkv_alt;
    unsigned* ctr;             // [layers][kPStages][kPMaxRT], zeroed before the launch
    unsigned* status;          // [4]: code, stage id, workgroup, spins
    long long* stamps;         // null, or [layers][kPStages][6] shader-clock stamps of workgroup stamp_wg (VLSA_TT_STAMPS=<wg>):
    int stamp_wg;              //   task start | dependency met | operands landed | arithmetic done | stores issued | drained + counted in
    const int* seq_row0;
    const unsigned char* cls_keep;
    size_t LF;                 // floats per layer region
    int d, heads, layers, M_pad, RT, n_seq, L, save;
    int n[kPStages];           // tasks per stage
    PLayerPtrs lp[kPMaxLayers];
};

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ONE lane: spin until *c >= expect (relaxed agent-scope loads).  A time-out -- or one already reported by anybody -- ends the wait.
__device__ __forceinline__ void p_wait_one(const unsigned* c, unsigned expect, unsigned* status, int stage) {
    unsigned spins = 0;
    while (ld_agent(c) < expect) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0u) {
            if (ld_agent(status) != 0u) return;
            if (spins >= kPSpinLimit) {
                if (atomicCAS(status, 0u, 1u) == 0u) {
                    __hip_atomic_store(status + 1, (unsigned)stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(status + 2, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(status + 3, ld_agent(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
        }
    }
}
// what the attention stage writes into row tile t: one task per head for the prefix rows and for every prompt whose own rows
// [seq_row0[s], seq_row0[s + 1]) touch the tile (pattn counts in on every tile of its row range)
struct PAttnRows {
    const int* seq_row0;   // null: the dependency is not the attention stage
    int n_seq, L, heads;
};
__device__ __forceinline__ unsigned p_attn_expect(const PAttnRows& ar, int t) {
    const int lo = t * 16, hi = lo + 16;
    unsigned n = (ar.L > 0 && lo < ar.L) ? 1u : 0u;
    for (int s = 0; s < ar.n_seq; ++s) n += (ar.seq_row0[s] < hi && ar.seq_row0[s + 1] > lo) ? 1u : 0u;
    return n * (unsigned)ar.heads;
}
// dependency of a task: counters ctr[first .. first + count) must each have reached `expect` (count <= 64: lane t of the first
// wave polls counter first + t, so that the round trips of the polls overlap instead of adding up)
__device__ __forceinline__ void p_wait(const unsigned* ctr, int first, int count, unsigned expect, unsigned* status, int stage) {
    if (ctr != nullptr && (int)threadIdx.x < count) p_wait_one(ctr + first + threadIdx.x, expect, status, stage);
    __syncthreads();
}
// end of a task: every wave drains its write-through stores, then ONE lane counts the task in on the row tiles it wrote
__device__ __forceinline__ void p_done(unsigned* ctr, int first, int count) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        for (int t = first; t < first + count; ++t) __hip_atomic_fetch_add(ctr + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PG {
    const float *A, *W, *bias, *resid, *ln_w, *ln_b;
    float *Y, *Yt, *Ypre;
    int ldr, ldy, N, K, RT, M_pad;
    long long* stamps;     // this task's six stamps, or null
};
#define P_STAMP(ptr, k)                                                                    \
    do {                                                                                   \
        if ((ptr) != nullptr && threadIdx.x == 0) (ptr)[k] = __builtin_readcyclecounter(); \
    } while (0)

// One 16-row x (16 NTW)-column tile of Y = pro(A) W^T (+ bias) (gelu) (+ resid): the arithmetic and operand layout of k_tt_gemm
// (MT = 1), eight waves splitting K (G = K / 128 sixteen-column groups each), weights prefetched BEFORE the dependency wait.
template <int PRO, int G, int NTW, int EPI>
__device__ __forceinline__ void pgemm(const PG& p, int b, float* red, const unsigned* dep, unsigned dep_expect, const PAttnRows& ar,
                                      unsigned* done, unsigned* status, int stage) {
    constexpr int NW = 8, Q = NTW * 4, QW = (Q + NW - 1) / NW;
    constexpr int PF = G < 8 ? G : 8;           // register ring: groups in flight (K = 768: the wave's whole K slice)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int K = p.K, N = p.N, RT = p.RT;
    const int NT = N / (16 * NTW), NT8 = NT & ~7;
    int ntile, mg;
    if (b < NT8 * RT) {         // the RT workgroups that share a weight tile on one XCD (workgroup b -> XCD b % 8): W comes from ITS L2
        const int j = b >> 3;
        ntile = (j / RT) * 8 + (b & 7);
        mg = j % RT;
    } else {
        const int bb = b - NT8 * RT;
        ntile = NT8 + bb / RT;
        mg = bb % RT;
    }
    const int n0 = ntile * (16 * NTW), m0 = mg * 16;
    const int KG = K >> 4, kbeg = w * (G * 16);
    P_STAMP(p.stamps, 0);
    const float* Wp[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) Wp[u] = p.W + ((size_t)((n0 >> 4) + u) * KG + (kbeg >> 4)) * 256 + lane * 4;
    // ---- weights first: they depend on nothing ------------------------------------------------------------------
    f32x4 rb[PF][NTW];
#pragma unroll
    for (int sI = 0; sI < PF; ++sI)
#pragma unroll
        for (int u = 0; u < NTW; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 256 * sI);
    float e_bias[QW];
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        const int q = w + i * NW;
        e_bias[i] = (q < Q && (EPI & EPI_BIAS)) ? p.bias[n0 + 16 * ((q >> 2) % NTW) + r] : 0.f;
    }
    f32x4 gld = {0.f, 0.f, 0.f, 0.f}, bld = {0.f, 0.f, 0.f, 0.f};
    if (PRO == PRO_LN && tid * 4 < K) {
        gld = *reinterpret_cast<const f32x4*>(p.ln_w + tid * 4);
        bld = *reinterpret_cast<const f32x4*>(p.ln_b + tid * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the activations of row tile mg exist once all their producers have counted in ----------------------------------
    p_wait(dep, mg, 1, (ar.seq_row0 != nullptr && threadIdx.x == 0) ? p_attn_expect(ar, mg) : dep_expect, status, stage);
    P_STAMP(p.stamps, 1);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((size_t)p.M_pad * K * sizeof(float)), 0x00020000);
    const int a_off = (int)((((size_t)mg * KG + (kbeg >> 4)) * 256 + lane * 4) * sizeof(float));
    auto ldA = [&](int jj) __attribute__((always_inline)) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, a_off + jj * 1024, 0, 16 /* sc1 */));
    };
    constexpr int PFA = G;                      // activations: the wave's whole slab in ONE round of write-through-visible loads
    f32x4 ra[PFA];
#pragma unroll
    for (int sI = 0; sI < PFA; ++sI) ra[sI] = ldA(sI);
    float e_res[QW];
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        const int q = w + i * NW;
        e_res[i] = 0.f;
        if (q < Q && (EPI & EPI_RESID)) e_res[i] = ld_agent(p.resid + (size_t)(m0 + 4 * g + (q & 3)) * p.ldr + n0 + 16 * ((q >> 2) % NTW) + r);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.stamps != nullptr) {      // stamp 2 = this wave's operands have landed (only taken when stamps are on: it drains the loads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P_STAMP(p.stamps, 2);
    }

    if constexpr (PRO == PRO_LN) {
        static_assert(G <= 8, "fused LayerNorm keeps the wave's whole A slab in the ring");
        // LayerNorm over the full row (K columns = the eight waves' slabs): two-pass statistics through LDS, as k_tt_gemm
        float* sgam = red + 1024;
        if (tid * 4 < K) {
            *reinterpret_cast<f32x4*>(sgam + tid * 4) = gld;
            *reinterpret_cast<f32x4*>(sgam + K + tid * 4) = bld;
        }
        float* st = red;                    // [NW][16] partial row statistics
        float s = 0.f;
#pragma unroll
        for (int jj = 0; jj < G; ++jj) s += (ra[jj][0] + ra[jj][1]) + (ra[jj][2] + ra[jj][3]);
        s = quad_rows_sum(s);
        if (g == 0) st[w * 16 + r] = s;
        __syncthreads();
        s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += st[ww * 16 + r];
        const float mean = s / (float)K;
        __syncthreads();
        s = 0.f;
#pragma unroll
        for (int jj = 0; jj < G; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float c = ra[jj][i] - mean;
                s = fmaf(c, c, s);
            }
        s = quad_rows_sum(s);
        if (g == 0) st[w * 16 + r] = s;
        __syncthreads();
        s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += st[ww * 16 + r];
        const float rstd = 1.f / sqrtf(s / (float)K + kLnEps);
        const float* gw = sgam + kbeg + 4 * g;
        const float* gb = sgam + K + kbeg + 4 * g;
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const f32x4 gam = *reinterpret_cast<const f32x4*>(gw + 16 * jj);
            const f32x4 bet = *reinterpret_cast<const f32x4*>(gb + 16 * jj);
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[jj][i] = fmaf((ra[jj][i] - mean) * rstd, gam[i], bet[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < G; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < NTW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[jj][i], rb[jj][u][i], acc[u], 0, 0, 0);
        __syncthreads();                    // every wave is done with gamma / beta in the buffer the reduction reuses
    } else {
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const int sI = jj % PF;
            const f32x4 a = ra[jj];
            f32x4 bq[NTW];
#pragma unroll
            for (int u = 0; u < NTW; ++u) bq[u] = rb[sI][u];
            if (jj + PF < G) {
#pragma unroll
                for (int u = 0; u < NTW; ++u) rb[sI][u] = *reinterpret_cast<const f32x4*>(Wp[u] + 256 * (jj + PF));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < NTW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bq[u][i], acc[u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- reduce the eight K slices through LDS (fixed order), epilogue, write-through stores ------------------------
    P_STAMP(p.stamps, 3);
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) red[(w * Q + u * 4 + v) * 64 + lane] = acc[u][v];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < QW; ++i) {
        const int q = w + i * NW;
        if (q >= Q) break;
        float val = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) val += red[(ww * Q + q) * 64 + lane];
        const int u = (q >> 2) % NTW, v = q & 3;
        const int row = m0 + 4 * g + v, col = n0 + 16 * u + r;
        if (EPI & EPI_BIAS) val += e_bias[i];
        if (EPI & EPI_GELU) {
            if (p.Ypre) st_agent(p.Ypre + (size_t)row * p.ldy + col, val);
            val = gelu(val);
        }
        if (EPI & EPI_RESID) val += e_res[i];
        if (p.Y) st_agent(p.Y + (size_t)row * p.ldy + col, val);
        if (p.Yt) st_agent(p.Yt + tiled_index(row, col, N), val);
    }
    P_STAMP(p.stamps, 4);
    p_done(done, mg, 1);
    P_STAMP(p.stamps, 5);
}

// attention of one (prompt, head) -- or of the shared prefix rows (seq == n_seq) -- as k_tt_attn_fwd, 512 threads, q / k / v read
// with agent-scope loads, output written through; counts in on every row tile it wrote.
__device__ __forceinline__ void pattn(const float* qkv, int ld, float* out_t, const int* __restrict__ seq_row0,
                                      const unsigned char* __restrict__ cls_keep, int heads, int d, int n_seq, int L, int task, float* lds,
                                      const unsigned* dep, int RT, unsigned dep_all, unsigned* done, unsigned* status, long long* stamps) {
    constexpr int LDK = kHeadDim + 1;
    P_STAMP(stamps, 0);
    float* Ks = lds;
    float* Vs = Ks + kPAttnMaxS * LDK;
    float* Qs = Vs + kPAttnMaxS * LDK;
    const int seq = task / heads, h = task % heads;
    const bool pfx_block = seq == n_seq;
    const int r0 = pfx_block ? 0 : seq_row0[seq];
    const int S = pfx_block ? L : L + seq_row0[seq + 1] - r0;
    const int qbeg = pfx_block ? 0 : L;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    auto grow = [&](int j) { return (pfx_block || j < L) ? j : r0 + (j - L); };
    p_wait(dep, 0, RT, dep_all, status, 1);                // the keys live in the prefix tile(s) and in the prompt's own tile(s): all of them
    P_STAMP(stamps, 1);
    for (int e = tid; e < S * kHeadDim; e += 512) {
        const int j = e >> 6, c = e & 63;
        const size_t base = (size_t)grow(j) * ld + h * kHeadDim + c;
        const float qv = ld_agent(qkv + base), kv = ld_agent(qkv + base + d), vv = ld_agent(qkv + base + 2 * d);
        Qs[j * kHeadDim + c] = qv * 0.125f;
        Ks[j * LDK + c] = kv;
        Vs[j * LDK + c] = vv;
    }
    __syncthreads();
    P_STAMP(stamps, 2);
    for (int i = qbeg + w; i < S; i += 8) {
        const float q = Qs[i * kHeadDim + lane];
        const bool is_cls = !pfx_block && i == S - 1;
        const int j = lane, jc = j < S ? j : S - 1;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < kHeadDim; ++c) dot = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(q), c)), Ks[jc * LDK + c], dot);
        const bool ok = j < S && (is_cls ? cls_keep[grow(jc)] != 0 : j <= i);
        const float sc = ok ? dot : -INFINITY;
        const float m = wave_max(sc);
        float pw = sc == -INFINITY ? 0.f : __expf(sc - m);
        pw *= 1.f / wave_sum(pw);
        float o = 0.f;
        for (int jj = 0; jj < S; ++jj)
            o = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pw), __builtin_amdgcn_readfirstlane(jj))), Vs[jj * LDK + lane], o);
        st_agent(out_t + tiled_index(grow(i), h * kHeadDim + lane, d), o);
    }
    const int first_row = grow(qbeg), last_row = grow(S - 1);
    P_STAMP(stamps, 3);
    P_STAMP(stamps, 4);
    p_done(done, first_row >> 4, (last_row >> 4) - (first_row >> 4) + 1);
    P_STAMP(stamps, 5);
}

__global__ __launch_bounds__(512) void k_tt_forward_persistent(const PArgs a) {
    extern __shared__ __attribute__((aligned(16))) float plds[];
    const int wg = blockIdx.x;
    const int d = a.d, Mp = a.M_pad, RT = a.RT;
    const size_t dd = (size_t)d * d;
    for (int L = 0; L < a.layers; ++L) {
        unsigned* c = a.ctr + (size_t)L * kPStages * kPMaxRT;
        const unsigned* prev = L > 0 ? c - kPMaxRT : nullptr;          // c_proj of the block before (block 0: the embedding launch)
        const PLayerPtrs& lp = a.lp[L];
        const PAttnRows no_ar{nullptr, 0, 0, 0}, ar{a.seq_row0, a.n_seq, a.L, a.heads};
        long long* sp = (a.stamps != nullptr && wg == a.stamp_wg) ? a.stamps + (size_t)L * kPStages * 6 : nullptr;
        const float* wl = a.wset + (size_t)L * 12 * dd;
        float* x_in = a.ws + (a.save ? (size_t)L : 0) * a.LF;
        float* qkv = (a.save || !(L & 1)) ? x_in + (size_t)Mp * d : a.qkv_alt;
        float* x_mid = x_in + (size_t)Mp * 4 * d;
        float* h_pre = x_mid + (size_t)Mp * d;
        float* x_next = (L + 1 < a.layers) ? (a.save ? a.ws + (size_t)(L + 1) * a.LF : x_in) : a.x_final;
        if (wg < a.n[0]) {      // qkv = ln_1(x_in) W_in^T + b
            PG p{a.xin_t, wl, lp.in_b, nullptr, lp.ln1_w, lp.ln1_b, qkv, nullptr, nullptr, 0, 3 * d, 3 * d, d, RT, Mp, sp};
            pgemm<PRO_LN, 6, 4, EPI_BIAS>(p, wg, plds, prev, (unsigned)(d / 32), no_ar, c, a.status, 0);
        }
        if (wg < a.n[1])
            pattn(qkv, 3 * d, a.attn_t, a.seq_row0, a.cls_keep, a.heads, d, a.n_seq, a.L, wg, plds, c, RT, (unsigned)(3 * d / 64),
                  c + kPMaxRT, a.status, sp ? sp + 6 : nullptr);
        if (wg < a.n[2]) {      // x_mid = x_in + attn W_out^T + b
            PG p{a.attn_t, wl + 3 * dd, lp.out_b, x_in, nullptr, nullptr, x_mid, a.xmid_t, nullptr, d, d, d, d, RT, Mp, sp ? sp + 12 : nullptr};
            pgemm<PRO_NONE, 6, 2, EPI_BIAS | EPI_RESID>(p, wg, plds, c + kPMaxRT, 0u, ar, c + 2 * kPMaxRT, a.status, 2);
        }
        if (wg < a.n[3]) {      // h = gelu(ln_2(x_mid) W_fc^T + b)
            PG p{a.xmid_t, wl + 4 * dd, lp.fc_b, nullptr, lp.ln2_w, lp.ln2_b, nullptr, a.hact_t, a.save ? h_pre : nullptr, 0, 4 * d, 4 * d, d, RT, Mp, sp ? sp + 18 : nullptr};
            pgemm<PRO_LN, 6, 6, EPI_BIAS | EPI_GELU>(p, wg, plds, c + 2 * kPMaxRT, (unsigned)(d / 32), no_ar, c + 3 * kPMaxRT, a.status, 3);
        }
        if (wg < a.n[4]) {      // x_next = x_mid + h W_proj^T + b
            PG p{a.hact_t, wl + 8 * dd, lp.proj_b, x_mid, nullptr, nullptr, x_next, a.xin_t, nullptr, d, d, d, 4 * d, RT, Mp, sp ? sp + 24 : nullptr};
            pgemm<PRO_NONE, 24, 2, EPI_BIAS | EPI_RESID>(p, wg, plds, c + 3 * kPMaxRT, (unsigned)(4 * d / 96), no_ar, c + 4 * kPMaxRT, a.status, 4);
        }
    }
}

}  // namespace tt
}  // namespace vlsa

using namespace vlsa;
using namespace vlsa::tt;

namespace {

struct Shape {
    int d, heads, layers, out_dim, M_pad, M, n_seq, ns_pad, L;
};

bool shape_of(const vlsa_tt_model* m, const vlsa_tt_rows* r, Shape& s) {
    if (!m || !m->layer) return false;
    s.d = m->width;
    s.heads = m->heads;
    s.layers = m->layers;
    s.out_dim = m->out_dim;
    // width: multiple of 128 (the K splits of every product are whole 16-column groups), <= 768 (fused-LayerNorm slab), 64 per head
    if (s.d < 128 || s.d > 768 || (s.d % 128) || s.heads * kHeadDim != s.d) return false;
    if (s.layers < 1 || s.out_dim < 64 || (s.out_dim % 64) || s.out_dim > 1024) return false;
    if (!m->pos_emb || !m->cls_emb || !m->lnf_w || !m->lnf_b || !m->text_proj) return false;
    if (!r) return true;
    s.M = r->M;
    s.M_pad = r->M_pad;
    s.n_seq = r->n_seq;
    s.ns_pad = (r->n_seq + 47) / 48 * 48;
    s.L = r->prefix_len;
    if (s.L < 0 || s.L >= r->max_len) return false;
    if (s.M < 1 || s.M_pad < s.M || (s.M_pad % 48) || s.n_seq < 1 || r->max_len < 2 || r->max_len > kAttnMaxS) return false;
    if (!r->row_seq || !r->row_pos || !r->row_src || !r->seq_row0 || !r->cls_keep) return false;
    return true;
}

// ---- packed (tiled) weights: [forward set | backward set]; a set = per layer {in_proj, out_proj, c_fc, c_proj} + projection --
inline size_t set_floats(const Shape& s) { return (size_t)s.layers * 12 * s.d * s.d + (size_t)s.d * s.out_dim; }
struct PackedLayer {
    const float *in_w, *out_w, *fc_w, *proj_w;
};
inline PackedLayer packed_layer(const float* set, const Shape& s, int L) {
    const size_t dd = (size_t)s.d * s.d;
    // (measurement aid, -DVLSA_EXPERIMENT builds only: every block reads block 0's weights -- the tower with its weights cache-resident;
    //  results are meaningless)
    static const bool same_w = VLSA_ENV("VLSA_TT_SAMEW") != nullptr;
    if (same_w) L = 0;
    const float* b = set + (size_t)L * 12 * dd;
    return PackedLayer{b, b + 3 * dd, b + 4 * dd, b + 8 * dd};
}
inline const float* packed_proj(const float* set, const Shape& s) { return set + (size_t)s.layers * 12 * s.d * s.d; }

// ---- workspace (floats).  Per-layer region (kept for backward when save != 0, else one region reused):
//      x_in [M_pad, d] | qkv [M_pad, 3d] | x_mid [M_pad, d] | h_pre [M_pad, 4d]     (all row-major)
//      | attn [M_pad, d] tiled: the attention output, kept per block by every saving forward (round 6: the attention backward's prefix-key
//        workgroups need it; with save == 2 also the out_proj weight gradient)
//      | x_in tiled [M_pad, d] | x_mid tiled [M_pad, d]   (round 6: the LayerNorm-backward prologues of the input-gradient products read them)
inline size_t layer_floats(const Shape& s) { return (size_t)s.M_pad * s.d * 12 + (size_t)s.M_pad * 4 + (size_t)s.M_pad * 32; }
inline float* layer_xin_t(float* region, const Shape& s) { return region + (size_t)s.M_pad * s.d * 10; }
inline float* layer_xmid_t(float* region, const Shape& s) { return region + (size_t)s.M_pad * s.d * 11; }
// (mean, rstd) of every row under ln_1 / ln_2 of the block: [M_pad][2] each, kept by the forward for the PRO_LNBWD prologues
inline float* layer_stats1(float* region, const Shape& s) { return region + (size_t)s.M_pad * s.d * 12; }
inline float* layer_stats2(float* region, const Shape& s) { return layer_stats1(region, s) + (size_t)s.M_pad * 2; }
// the attention's row statistics (max, 1 / sum) [M_pad][heads <= 16][2] and -- in the `attn` slot of the region -- its output, kept by every
// saving forward for the attention backward's prefix-key workgroups
inline float* layer_astats(float* region, const Shape& s) { return layer_stats2(region, s) + (size_t)s.M_pad * 2; }
inline float* layer_attn_t(float* region, const Shape& s) { return region + (size_t)s.M_pad * s.d * 9; }
struct Scratch {   // behind the layer regions; *_t = tiled
    float *x_final, *xin_t, *xmid_t, *attn_t, *hact_t, *pooled_t, *feat;                              // forward
    float *dout_t, *dpool, *dxa, *dxa_t, *dxb, *dxb_t, *dh_t, *da, *da_t, *dattn, *dqkv_t;             // backward
    float* pfx;             // shared prefix: per (block, head) partial dK / dV of the prefix rows [(n_seq + 1)][heads][L][128]
    unsigned int* cnt;      // [heads] tickets (zero between launches: the workspace is zeroed once by the caller)
    float *lnout, *lnstats; // weight gradients: LayerNorm output [M_pad, d] row-major and (mean, rstd) per row of the block at hand
    float* qkv_alt;         // persistent forward without saved activations: qkv of the odd blocks (see k_tt_forward_persistent)
    unsigned int* pctr;     // persistent forward: [kPMaxLayers][kPStages][kPMaxRT] task counters | 8 status words
};
constexpr size_t kPCtrWords = (size_t)kPMaxLayers * kPStages * kPMaxRT;
inline size_t pfx_floats(const Shape& s) { return (size_t)(s.n_seq + 1) * s.heads * (s.L > 0 ? s.L : 0) * 128 + 64; }
inline size_t scratch_floats(const Shape& s) {
    return (size_t)s.M_pad * s.d * (1 + 1 + 1 + 1 + 4 + 2 + 2 + 4 + 2 + 1 + 3) + (size_t)s.ns_pad * (2 * s.d + 2 * s.out_dim) + pfx_floats(s)
           + (size_t)s.M_pad * 3 * s.d + kPCtrWords + 8 + 2 * (size_t)kPMaxLayers * kPStages * 6 + 2 + (size_t)s.M_pad * (s.d + 2);
}
inline Scratch scratch_of(float* p, const Shape& s) {
    const size_t md = (size_t)s.M_pad * s.d;
    Scratch c;
    c.x_final = p; p += md;
    c.xin_t = p; p += md;
    c.xmid_t = p; p += md;
    c.attn_t = p; p += md;
    c.hact_t = p; p += 4 * md;
    c.dxa = p; p += md;
    c.dxa_t = p; p += md;
    c.dxb = p; p += md;
    c.dxb_t = p; p += md;
    c.dh_t = p; p += 4 * md;
    c.da = p; p += md;
    c.da_t = p; p += md;
    c.dattn = p; p += md;
    c.dqkv_t = p; p += 3 * md;
    c.pooled_t = p; p += (size_t)s.ns_pad * s.d;
    c.dpool = p; p += (size_t)s.ns_pad * s.d;
    c.feat = p; p += (size_t)s.ns_pad * s.out_dim;
    c.dout_t = p; p += (size_t)s.ns_pad * s.out_dim;
    c.lnout = p; p += (size_t)s.M_pad * s.d;
    c.lnstats = p; p += (size_t)s.M_pad * 2;
    c.cnt = reinterpret_cast<unsigned int*>(p); p += 64;
    c.pfx = p; p += pfx_floats(s) - 64;
    c.qkv_alt = p; p += (size_t)s.M_pad * 3 * s.d;
    c.pctr = reinterpret_cast<unsigned int*>(p);
    return c;
}

#ifdef VLSA_TT_DEBUG
static int tt_debug_bits_all() {
    const char* e = VLSA_ENV("VLSA_TT_DEBUG_ALL");
    return e ? (atoi(e) << 8) : 0;
}
#endif
template <int MT, int NW, int PRO, int GT, int NTW = 2>
int launch_gemm_g(GemmArgs a, int M_pad, hipStream_t st) {
#ifdef VLSA_TT_DEBUG
    a.epi |= tt_debug_bits_all();     // experiments on EVERY product of the tower (results are then wrong: timing only)
#endif
    // row groups: only those that hold real rows (the rows behind M_real are padding nobody reads a result from)
    a.MG = a.M_real > 0 ? (a.M_real + 16 * MT - 1) / (16 * MT) : M_pad / (16 * MT);
    if (a.MG * 16 * MT > M_pad) a.MG = M_pad / (16 * MT);
    const int NT = a.N / (16 * NTW);
    a.xcd_map = NT >= 8 ? 1 : 0;      // XCD-aware block -> tile map (whole rounds of 8 tiles; the rest linear)
    size_t lds = (size_t)NW * MT * NTW * 4 * 64 * sizeof(float);
    if (PRO != PRO_NONE && lds < (size_t)(1024 + 2 * a.K) * sizeof(float)) lds = (size_t)(1024 + 2 * a.K) * sizeof(float);
    if (lds > 64 * 1024) {
        static DeviceOnce once;
        if (once.first()) (void)hipFuncSetAttribute((const void*)k_tt_gemm<MT, NW, PRO, GT, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL((k_tt_gemm<MT, NW, PRO, GT, NTW>), dim3(NT * a.MG), dim3(NW * 64), lds, st, a);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}
// G1, G2: the group counts of the CONCH-size tower for this product (compile-time specialisations); anything else: runtime G
// wide products (N a multiple of 48): 48-column workgroup tiles, see k_tt_gemm
template <int MT, int NW, int PRO, int G1>
int launch_gemm_wide(const GemmArgs& a, int M_pad, hipStream_t st) {
    const int G = a.K / NW / 16;
    if (a.N % 48 == 0) {
        if (G == G1) return launch_gemm_g<MT, NW, PRO, G1, 3>(a, M_pad, st);
        return launch_gemm_g<MT, NW, PRO, 0, 3>(a, M_pad, st);
    }
    if (G == G1) return launch_gemm_g<MT, NW, PRO, G1, 2>(a, M_pad, st);
    return launch_gemm_g<MT, NW, PRO, 0, 2>(a, M_pad, st);
}
// 16-row workgroup tiles (the N = d products): 32-column tiles when the grid then still fits one round of the 256 CUs -- more
// workgroups with a third less weight traffic each (the c_proj product moves the most bytes per workgroup: 18.7 -> ~10 us) --
// else the 48-column choice of launch_gemm_wide.  M = rows that carry data.
template <int NW, int G1>
int launch_gemm_rows16(GemmArgs a, int M, int M_pad, hipStream_t st) {
    a.M_real = M;
    const int G = a.K / NW / 16;
    if (a.N % 32 == 0 && ((M + 15) / 16) * (a.N / 32) <= 256) {
        if (G == G1) return launch_gemm_g<1, NW, PRO_NONE, G1, 2>(a, M_pad, st);
        return launch_gemm_g<1, NW, PRO_NONE, 0, 2>(a, M_pad, st);
    }
    return launch_gemm_wide<1, NW, PRO_NONE, G1>(a, M_pad, st);
}
template <int MT, int NW, int PRO, int G1, int G2>
int launch_gemm_wide2(const GemmArgs& a, int M_pad, hipStream_t st) {
    const int G = a.K / NW / 16;
    if (a.N % 48 == 0) {
        if (G == G1) return launch_gemm_g<MT, NW, PRO, G1, 3>(a, M_pad, st);
        if (G == G2) return launch_gemm_g<MT, NW, PRO, G2, 3>(a, M_pad, st);
        return launch_gemm_g<MT, NW, PRO, 0, 3>(a, M_pad, st);
    }
    if (G == G1) return launch_gemm_g<MT, NW, PRO, G1, 2>(a, M_pad, st);
    if (G == G2) return launch_gemm_g<MT, NW, PRO, G2, 2>(a, M_pad, st);
    return launch_gemm_g<MT, NW, PRO, 0, 2>(a, M_pad, st);
}
template <int MT, int NW, int PRO, int G1, int G2 = G1>
int launch_gemm(const GemmArgs& a, int M_pad, hipStream_t st) {
    const int G = a.K / NW / 16;
    if (G == G1) return launch_gemm_g<MT, NW, PRO, G1>(a, M_pad, st);
    if (G == G2) return launch_gemm_g<MT, NW, PRO, G2>(a, M_pad, st);
    return launch_gemm_g<MT, NW, PRO, 0>(a, M_pad, st);
}
inline GemmArgs gemm_args(const float* A, const float* W, int N, int K) {
    GemmArgs a{};
    a.A = A;
    a.W = W;
    a.N = N;
    a.K = K;
    return a;
}
// this launch also pulls the NEXT product's weights Wn (N' = Nn columns in tiles of 16 ntw, K' = Kn) towards the XCDs that will read them
// (prefetch_next); on: false switches it off (shapes whose launch does not use the tile sizes assumed here)
inline void prefetch_for(GemmArgs& a, bool on, const float* Wn, int Nn, int Kn, int ntw) {
#ifdef VLSA_EXPERIMENT
    static const bool off = VLSA_ENV("VLSA_TT_NOPF") != nullptr;
    if (off) return;
#endif
    const int per_xcd = ((Nn / (16 * ntw)) & ~7) / 8;
    if (!on || !Wn || per_xcd < 1) return;
    a.pfW = Wn;
    a.pf_tile_floats = 16 * ntw * Kn;
    a.pf_tiles_xcd = per_xcd;
    a.pf_magic = 65536 / per_xcd + 1;
}
inline void prefetch_region(GemmArgs& a, const float* ptr, size_t floats) {
#ifdef VLSA_EXPERIMENT
    static const bool off = VLSA_ENV("VLSA_TT_NOPF2") != nullptr || VLSA_ENV("VLSA_TT_NOPF") != nullptr;
    if (off) return;
#endif
    a.pf2 = ptr;
    a.pf2_lines = (int)(floats / 32);
}


void launch_ln_bwd(const float* da, const float* x, const float* gamma, const float* dres, float* dx, float* dxt, int d, int rows,
                   hipStream_t st) {
    if (d % 256 == 0 && d <= 256 * kLnSlots4)
        hipLaunchKernelGGL(k_tt_ln_bwd4, dim3((rows + 3) / 4), dim3(256), 0, st, da, x, gamma, dres, dx, dxt, d, rows, ScatterArgs{});
    else
        hipLaunchKernelGGL(k_tt_ln_bwd, dim3((rows + 3) / 4), dim3(256), 0, st, da, x, gamma, dres, dx, dxt, d, rows);
}

// (measurement aid: the attention backward's timing-only ablation bits ride in the high half of its `ldo` argument, -DVLSA_EXPERIMENT only)
inline int attn_bwd_ldo(int d) {
#ifdef VLSA_EXPERIMENT
    static const int abl = VLSA_ENV("VLSA_TT_ATTN_ABL") ? atoi(VLSA_ENV("VLSA_TT_ATTN_ABL")) : 0;
    return d | (abl << 16);
#else
    return d;
#endif
}
// threads of an attention workgroup: a wave per row of the prompt (up to 16 waves); VLSA_TT_ATTN_THREADS overrides (A/B hook)
int attn_threads(int max_len) {
    if (const char* e = VLSA_ENV("VLSA_TT_ATTN_THREADS")) {
        const int v = atoi(e);
        if (v >= 64 && v <= 1024 && v % 64 == 0) return v;
    }
    int wv = max_len < 4 ? 4 : (max_len > 16 ? 16 : max_len);
    return 64 * wv;
}

// ---- persistent forward: when it applies, and its launch --------------------------------------------------------
bool persist_supported(const Shape& s, const vlsa_tt_rows* r, int flags) {
    // OPT-IN per call (VLSA_TT_PERSISTENT or-ed into save_for_backward; the library itself reads no environment).  Measured on MI355X,
    // K = 12 rank prompts (profiles/r04_bench_text_persist.txt, r04_tt_persist_stamps.txt): 836-872 us against 616 us for the
    // launch-per-stage path -- a stage's in-kernel hand-off (drain of the write-through stores + counter + poll + first round of sc1
    // loads, ~3.5 us) costs MORE than the ~3.2 us kernel boundary it replaces, and the 4-byte sc1 epilogue stores are 2x slower than
    // plain ones; the weight prefetch it buys (task bodies 5.8 vs 8.5 us for QKV) does not make up for it.  Kept, tested, off by default.
    if (!(flags & VLSA_TT_PERSISTENT)) return false;
    const int RT = (s.M + 15) / 16;       // 7 row tiles x 36 QKV column tiles = 252 workgroups: one round of the CUs
    return s.d == 768 && s.heads == 12 && s.layers <= kPMaxLayers && RT <= 7 && RT * 16 <= s.M_pad && r->max_len <= kPAttnMaxS
           && (s.n_seq + (s.L > 0 ? 1 : 0)) * s.heads <= 256;
}

#define TT_TRY(expr)                  \
    do {                              \
        const int rc_ = (expr);       \
        if (rc_ != VLSA_OK) return rc_; \
    } while (0)
#define TT_LAUNCHED()                                              \
    do {                                                           \
        if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;  \
    } while (0)

int pack_one(const float* W, int rows, int cols, int transpose, float* out, hipStream_t st) {
    const size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(k_tt_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, rows, cols, transpose, out);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

}  // namespace

#ifdef VLSA_TT_DEBUG
extern "C" int vlsa_tt_debug_stamps(long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(vlsa::tt::tt_stamps), sizeof(long long) * 16) == hipSuccess ? 0 : -3;
}
#include <cstdlib>
static int tt_debug_bits() {
    const char* e = VLSA_ENV("VLSA_TT_DEBUG_BITS");
    return e ? (atoi(e) << 8) : 0;
}
#define TT_DBG_BITS tt_debug_bits()
#else
#define TT_DBG_BITS 0
#endif

extern "C" size_t vlsa_tt_packed_bytes(const vlsa_tt_model* m, int with_backward) {
    Shape s;
    if (!shape_of(m, nullptr, s)) return 0;
    return set_floats(s) * (with_backward ? 2 : 1) * sizeof(float);
}

extern "C" int vlsa_tt_pack_weights(const vlsa_tt_model* m, void* packed, int with_backward, void* stream) {
    Shape s;
    if (!shape_of(m, nullptr, s) || !packed) return VLSA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int d = s.d;
    const size_t dd = (size_t)d * d;
    float* fwd = static_cast<float*>(packed);
    float* bwd = fwd + set_floats(s);
    for (int L = 0; L < s.layers; ++L) {
        const vlsa_tt_layer& w = m->layer[L];
        if (!w.in_w || !w.out_w || !w.fc_w || !w.proj_w) return VLSA_EINVAL;
        float* f = fwd + (size_t)L * 12 * dd;
        // forward: Y = A W^T with W [N, K] as stored -> tiled [N, K]
        TT_TRY(pack_one(w.in_w, 3 * d, d, 0, f, st));
        TT_TRY(pack_one(w.out_w, d, d, 0, f + 3 * dd, st));
        TT_TRY(pack_one(w.fc_w, 4 * d, d, 0, f + 4 * dd, st));
        TT_TRY(pack_one(w.proj_w, d, 4 * d, 0, f + 8 * dd, st));
        if (with_backward) {   // input gradients: dA = dY W = dY (W^T)^T -> the same product with the tiled TRANSPOSE [K, N]
            float* b = bwd + (size_t)L * 12 * dd;
            TT_TRY(pack_one(w.in_w, 3 * d, d, 1, b, st));
            TT_TRY(pack_one(w.out_w, d, d, 1, b + 3 * dd, st));
            TT_TRY(pack_one(w.fc_w, 4 * d, d, 1, b + 4 * dd, st));
            TT_TRY(pack_one(w.proj_w, d, 4 * d, 1, b + 8 * dd, st));
        }
    }
    // text_projection [d, out_dim]: forward pooled @ P = pooled (P^T)^T -> tiled P^T [out_dim, d]; backward dout @ P^T -> tiled P [d, out_dim]
    TT_TRY(pack_one(m->text_proj, d, s.out_dim, 1, fwd + (size_t)s.layers * 12 * dd, st));
    if (with_backward) TT_TRY(pack_one(m->text_proj, d, s.out_dim, 0, bwd + (size_t)s.layers * 12 * dd, st));
    return VLSA_OK;
}

extern "C" size_t vlsa_tt_workspace_bytes(const vlsa_tt_model* m, const vlsa_tt_rows* r, int save_for_backward) {
    save_for_backward &= 0xff;
    Shape s;
    if (!r || !shape_of(m, r, s)) return 0;
    const size_t regions = save_for_backward ? (size_t)s.layers : 1;
    return (regions * layer_floats(s) + scratch_floats(s)) * sizeof(float);
}

extern "C" int64_t vlsa_tt_status_offset(const vlsa_tt_model* m, const vlsa_tt_rows* r, int save_for_backward) {
    Shape s;
    if (!r || !shape_of(m, r, s)) return -1;
    const int flags = save_for_backward;
    save_for_backward &= 0xff;
    if (save_for_backward == 2 || !persist_supported(s, r, flags)) return -1;         // the launch-per-stage path has no in-kernel waits
    const size_t nreg = save_for_backward ? (size_t)s.layers : 1;
    float* base = nullptr;
    const Scratch c = scratch_of(base + nreg * layer_floats(s), s);
    return (int64_t)((reinterpret_cast<const char*>(c.pctr + kPCtrWords)) - reinterpret_cast<const char*>(base));
}

extern "C" int vlsa_tt_forward(const vlsa_tt_model* m, const vlsa_tt_rows* r, const void* packed, const float* emb,
                               int64_t emb_seq_stride, int64_t emb_tok_stride, void* workspace, int save_for_backward, float* out,
                               void* stream) {
    Shape s;
    if (!r || !shape_of(m, r, s)) return VLSA_EINVAL;
    if (!packed || !emb || !workspace || !out) return VLSA_EINVAL;
    const int tt_flags = save_for_backward;
    save_for_backward &= 0xff;
    hipStream_t st = (hipStream_t)stream;
    const int d = s.d, Mp = s.M_pad;
    float* ws = static_cast<float*>(workspace);
    const size_t LF = layer_floats(s);
    const size_t nreg = save_for_backward ? (size_t)s.layers : 1;
    const Scratch c = scratch_of(ws + nreg * LF, s);
    const float* wset = static_cast<const float*>(packed);
    auto region = [&](int layer) { return ws + (save_for_backward ? (size_t)layer : 0) * LF; };

    const bool keep_attn = save_for_backward == 2;          // a training tower: the attention output of every block stays
    const bool persist = !keep_attn && persist_supported(s, r, tt_flags);
    // tiled copies of a block's input / its middle: scratch, or -- kept for the backward pass -- behind the block's region
    auto xin_t_of = [&](int layer) { return (save_for_backward && !persist) ? layer_xin_t(region(layer), s) : c.xin_t; };
    auto xmid_t_of = [&](int layer) { return (save_for_backward && !persist) ? layer_xmid_t(region(layer), s) : c.xmid_t; };
    // the tile shapes of the few-row (K = 12 prompts) launches below: 16 x 64 in_proj, 16 x 32 out_proj / c_proj, 16 x 96 c_fc -- what the
    // weight prefetch of the launch in front of each assumes
    const int rt = (s.M + 15) / 16;
    const bool few = d / 4 / 16 == 12 && (3 * d) % 64 == 0 && rt * (3 * d / 64) <= 256 && (4 * d) % 96 == 0 && rt * (4 * d / 96) <= 256
                     && d % 32 == 0 && rt * (d / 32) <= 256;
    hipLaunchKernelGGL(k_tt_embed, dim3(Mp), dim3(256), 0, st, region(0), xin_t_of(0), d, emb, emb_seq_stride, emb_tok_stride, r->row_seq,
                       r->row_pos, r->row_src, m->pos_emb, m->cls_emb, s.M);
    TT_LAUNCHED();
    if (persist) {
        PArgs a{};
        a.ws = ws; a.wset = wset;
        a.x_final = c.x_final; a.xin_t = c.xin_t; a.xmid_t = c.xmid_t; a.attn_t = c.attn_t; a.hact_t = c.hact_t; a.qkv_alt = c.qkv_alt;
        a.ctr = c.pctr; a.status = c.pctr + kPCtrWords;
        if (const char* e = VLSA_ENV("VLSA_TT_STAMPS")) {      // measurement aid (tools/tt_persist_stamps.py): stamps behind the status words
            a.stamps = reinterpret_cast<long long*>((reinterpret_cast<uintptr_t>(c.pctr + kPCtrWords + 8) + 7) & ~(uintptr_t)7);
            a.stamp_wg = atoi(e);
        }
        a.seq_row0 = r->seq_row0; a.cls_keep = r->cls_keep;
        a.LF = LF; a.d = d; a.heads = s.heads; a.layers = s.layers; a.M_pad = Mp; a.RT = (s.M + 15) / 16; a.n_seq = s.n_seq; a.L = s.L;
        a.save = save_for_backward ? 1 : 0;
        a.n[0] = a.RT * (3 * d / 64); a.n[1] = (s.n_seq + (s.L > 0 ? 1 : 0)) * s.heads; a.n[2] = a.RT * (d / 32);
        a.n[3] = a.RT * (4 * d / 96); a.n[4] = a.RT * (d / 32);
        int grid = 0;
        for (int i = 0; i < kPStages; ++i) grid = a.n[i] > grid ? a.n[i] : grid;
        for (int L = 0; L < s.layers; ++L) {
            const vlsa_tt_layer& w = m->layer[L];
            a.lp[L] = PLayerPtrs{w.ln1_w, w.ln1_b, w.in_b, w.out_b, w.ln2_w, w.ln2_b, w.fc_b, w.proj_b};
        }
        // every polled word back to zero before the launch (a memset node: replayed first under graph capture too)
        if (hipMemsetAsync(c.pctr, 0, (kPCtrWords + 8) * sizeof(unsigned), st) != hipSuccess) return VLSA_ELAUNCH;
        constexpr size_t lds = 12544 * sizeof(float);      // max(reduction 8 x 24 x 64, attention 3 x 64 x 65, LayerNorm scratch)
        static DeviceOnce once;
        if (once.first()) (void)hipFuncSetAttribute((const void*)k_tt_forward_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_tt_forward_persistent, dim3(grid), dim3(512), lds, st, a);
        TT_LAUNCHED();
        if (save_for_backward) {            // the backward pass reads tiled copies of every block's x_in / x_mid: made here from the rows the
            for (int L = 0; L < s.layers; ++L) {            // persistent launch kept (an opt-in, slower path anyway: 24 small launches)
                float* x_in = region(L);
                float* x_mid = x_in + (size_t)Mp * 4 * d;
                hipLaunchKernelGGL(k_tt_tile_rows, dim3(Mp / 16, d / 16), dim3(256), 0, st, x_in, Mp, d, layer_xin_t(x_in, s));
                hipLaunchKernelGGL(k_tt_tile_rows, dim3(Mp / 16, d / 16), dim3(256), 0, st, x_mid, Mp, d, layer_xmid_t(x_in, s));
                hipLaunchKernelGGL(k_tt_ln_rows, dim3(Mp / 4), dim3(256), 0, st, x_in, m->layer[L].ln1_w, m->layer[L].ln1_b, (float*)nullptr,
                                   layer_stats1(x_in, s), d, Mp);
                hipLaunchKernelGGL(k_tt_ln_rows, dim3(Mp / 4), dim3(256), 0, st, x_mid, m->layer[L].ln2_w, m->layer[L].ln2_b, (float*)nullptr,
                                   layer_stats2(x_in, s), d, Mp);
                // ... and the attention's output + row statistics (the attention once more, from the block's kept q / k / v)
                hipLaunchKernelGGL(k_tt_attn_fwd, dim3((s.n_seq + (s.L > 0 ? 1 : 0)) * s.heads), dim3(attn_threads(r->max_len)), 0, st,
                                   x_in + (size_t)Mp * d, 3 * d, layer_attn_t(x_in, s), r->seq_row0, r->cls_keep, s.heads, d, s.n_seq, s.L,
                                   layer_astats(x_in, s));
            }
            TT_LAUNCHED();
        }
    }
    for (int L = 0; L < (persist ? 0 : s.layers); ++L) {
        const vlsa_tt_layer& w = m->layer[L];
        const PackedLayer pw = packed_layer(wset, s, L);
        float* x_in = region(L);
        float* qkv = x_in + (size_t)Mp * d;
        float* x_mid = qkv + (size_t)Mp * 3 * d;
        float* h_pre = x_mid + (size_t)Mp * d;
        float* x_next = (L + 1 < s.layers) ? (save_for_backward ? region(L + 1) : x_in) : c.x_final;
        float* attn_t = save_for_backward ? layer_attn_t(x_in, s) : c.attn_t;
        // x_mid = x_in + out_proj(attention(ln_1(x_in)));  x_next = x_mid + c_proj(gelu(c_fc(ln_2(x_mid))))
        {
            GemmArgs a = gemm_args(xin_t_of(L), pw.in_w, 3 * d, d);
            a.bias = w.in_b; a.Y = qkv; a.ldy = 3 * d; a.epi = EPI_BIAS | TT_DBG_BITS; a.ln_w = w.ln1_w; a.ln_b = w.ln1_b;
            a.M_real = s.M;
            if (save_for_backward) a.stats_out = layer_stats1(x_in, s);
            prefetch_for(a, few, pw.out_w, d, d, 2);
            // 32-row workgroup tiles when they still fit one round of the CUs (K = 12 prompts: 5 x 48 = 240 workgroups of 2/3 the
            // work instead of 4 x 48 = 192), else 48-row tiles
            const int nt = (3 * d) % 48 == 0 ? (3 * d) / 48 : (3 * d) / 32;
            // few rows (shared prefix: K = 12 prompts are 101 rows = 7 row tiles): whenever 16 x 64 tiles fit ONE round of the CUs
            // (7 x 36 = 252 workgroups) they are taken -- 2/3 of the MFMA work of a 32 x 48 tile per workgroup, and a launch lasts as long
            // as one workgroup (forward 660 -> 620 us, forward + backward 1654 -> 1593 us; until the end of round 3 a second condition that
            // could never hold kept these shapes switched off)
            if ((3 * d) % 64 == 0 && ((s.M + 15) / 16) * (3 * d / 64) <= 256 && d / 4 / 16 == 12)
                TT_TRY((launch_gemm_g<1, 4, PRO_LN, 12, 4>(a, Mp, st)));
            else if (Mp % 32 == 0 && ((s.M + 31) / 32) * nt <= 256) TT_TRY((launch_gemm_wide<2, 4, PRO_LN, 12>(a, Mp, st)));
            else TT_TRY((launch_gemm_wide<3, 4, PRO_LN, 12>(a, Mp, st)));
        }
        hipLaunchKernelGGL(k_tt_attn_fwd, dim3((s.n_seq + (s.L > 0 ? 1 : 0)) * s.heads), dim3(attn_threads(r->max_len)), 0, st, qkv, 3 * d, attn_t, r->seq_row0,
                           r->cls_keep, s.heads, d, s.n_seq, s.L, save_for_backward ? layer_astats(x_in, s) : (float*)nullptr);
        TT_LAUNCHED();
        {
            GemmArgs a = gemm_args(attn_t, pw.out_w, d, d);
            a.bias = w.out_b; a.resid = x_in; a.ldr = d; a.Y = x_mid; a.ldy = d; a.Yt = xmid_t_of(L); a.epi = EPI_BIAS | EPI_RESID;
            prefetch_for(a, few, pw.fc_w, 4 * d, d, 6);
            TT_TRY((launch_gemm_rows16<4, 12>(a, s.M, Mp, st)));
        }
        {
            GemmArgs a = gemm_args(xmid_t_of(L), pw.fc_w, 4 * d, d);
            a.bias = w.fc_b; a.Yt = c.hact_t; a.Ypre = save_for_backward ? h_pre : nullptr; a.ldy = 4 * d; a.epi = EPI_BIAS | EPI_GELU;
            a.ln_w = w.ln2_w; a.ln_b = w.ln2_b;
            a.M_real = s.M;
            if (save_for_backward) a.stats_out = layer_stats2(x_in, s);
            prefetch_for(a, few, pw.proj_w, d, 4 * d, 2);
            // 32 x 64 workgroup tiles when they fit one round (K = 12 prompts: 5 x 48 = 240 workgroups, 160 instead of 192 rows of
            // f32-MFMA work -- this product is matrix-pipe-bound), else 48 x 48
            if ((4 * d) % 96 == 0 && ((s.M + 15) / 16) * (4 * d / 96) <= 256 && d / 4 / 16 == 12)
                TT_TRY((launch_gemm_g<1, 4, PRO_LN, 12, 6>(a, Mp, st)));       // few rows: 16 x 96 tiles (7 x 32 = 224 workgroups)
            else if (Mp % 32 == 0 && (4 * d) % 64 == 0 && ((s.M + 31) / 32) * (4 * d / 64) <= 256 && d / 4 / 16 == 12)
                TT_TRY((launch_gemm_g<2, 4, PRO_LN, 12, 4>(a, Mp, st)));
            else TT_TRY((launch_gemm_wide<3, 4, PRO_LN, 12>(a, Mp, st)));
        }
        {
            GemmArgs a = gemm_args(c.hact_t, pw.proj_w, d, 4 * d);
            a.bias = w.proj_b; a.resid = x_mid; a.ldr = d; a.Y = x_next; a.ldy = d; a.Yt = (L + 1 < s.layers) ? xin_t_of(L + 1) : c.xin_t; a.epi = EPI_BIAS | EPI_RESID;
            if (L + 1 < s.layers) prefetch_for(a, few, packed_layer(wset, s, L + 1).in_w, 3 * d, d, 4);
            TT_TRY((launch_gemm_rows16<8, 24>(a, s.M, Mp, st)));
        }
    }
    hipLaunchKernelGGL(k_tt_lnf_fwd, dim3((s.ns_pad + 3) / 4), dim3(256), 0, st, c.x_final, r->seq_row0, m->lnf_w, m->lnf_b, c.pooled_t, d,
                       s.n_seq, s.ns_pad);
    TT_LAUNCHED();
    {   // text features = pooled @ text_projection
        GemmArgs a = gemm_args(c.pooled_t, packed_proj(wset, s), s.out_dim, d);
        a.Y = c.feat; a.ldy = s.out_dim;
        // 16-row x 16-column tiles over the row groups that hold prompts (K = 12 prompts: 32 workgroups of 49 KB of weights each; the
        // 48 x 32 tiles of the general shape were 16 workgroups of 98 KB and three times the MFMA work: 10.0 -> 5.2 us per launch)
        a.M_real = s.n_seq;
        if (s.out_dim % 16 == 0 && d / 4 / 16 == 12) {
            a.Y = out; a.M_store = s.n_seq;         // straight into the caller's [n_seq, out_dim] (no copy launch behind the product)
            TT_TRY((launch_gemm_g<1, 4, PRO_NONE, 12, 1>(a, s.ns_pad, st)));
            return VLSA_OK;
        }
        TT_TRY((launch_gemm<3, 4, PRO_NONE, 12>(a, s.ns_pad, st)));
    }
    if (hipMemcpyAsync(out, c.feat, (size_t)s.n_seq * s.out_dim * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return VLSA_ELAUNCH;
    return VLSA_OK;
}

namespace {
// dW = dY^T act (+ dbias = column sums of dY) for one product; act row-major (optionally through gelu) or tiled
int launch_dw(int bsrc, const float* dY_t, const float* act, int ldb, float* dW, float* dbias, int N, int K, int M, hipStream_t st) {
    if (!dW) return VLSA_OK;
    if ((N % 128) || (K % 64)) return VLSA_EUNSUPPORTED;
    const DwArgs a{dY_t, act, dW, dbias, N, K, M, ldb};
    const dim3 grid(K / 64, N / 128);
    if (bsrc == DW_TILED) hipLaunchKernelGGL(k_tt_dw<DW_TILED>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_tt_dw<DW_ROWS>, grid, dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

// gr != null: ALSO the gradients of the tower's own parameters, written (not accumulated) into the buffers gr points at -- a
// vlsa_tt_model whose pointers are the gradient tensors, each shaped like the parameter it belongs to; the forward must have run
// with save_for_backward == 2.
int tt_backward(const vlsa_tt_model* m, const vlsa_tt_rows* r, const void* packed, const float* dout, void* workspace, float* demb,
                int64_t emb_seq_stride, int64_t emb_tok_stride, int64_t demb_floats, const vlsa_tt_model* gr, void* stream) {
    Shape s;
    if (!r || !shape_of(m, r, s)) return VLSA_EINVAL;
    if (!packed || !dout || !workspace || !demb || demb_floats < 0) return VLSA_EINVAL;
    if (r->max_len > kAttnBwdMaxS) return VLSA_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int d = s.d, Mp = s.M_pad;
    float* ws = static_cast<float*>(workspace);
    const size_t LF = layer_floats(s);
    const Scratch c = scratch_of(ws + (size_t)s.layers * LF, s);
    const float* bset = static_cast<const float*>(packed) + set_floats(s);

    {   // d pooled = dout @ text_projection^T
        GemmArgs a = gemm_args(c.dout_t, packed_proj(bset, s), d, s.out_dim);
        a.Y = c.dpool; a.ldy = d;
        a.M_real = s.n_seq;
        if (d % 16 == 0 && s.out_dim / 4 / 16 == 8 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0) {
            a.A = dout; a.lda = s.out_dim; a.a_rows = s.n_seq;       // the incoming gradient as it lies (no tiling launch in front)
            TT_TRY((launch_gemm_g<1, 4, PRO_NONE, 8, 1>(a, s.ns_pad, st)));
        } else {
            hipLaunchKernelGGL(k_tt_tile_rows, dim3(s.ns_pad / 16, s.out_dim / 16), dim3(256), 0, st, dout, s.n_seq, s.out_dim, c.dout_t);
            TT_LAUNCHED();
            TT_TRY((launch_gemm<3, 4, PRO_NONE, 8>(a, s.ns_pad, st)));
        }
    }
    hipLaunchKernelGGL(k_tt_lnf_bwd, dim3((Mp + 3) / 4), dim3(256), 0, st, c.dpool, c.x_final, r->row_seq, r->row_src, m->lnf_w, c.dxa,
                       c.dxa_t, d, s.M, Mp);
    TT_LAUNCHED();
    if (gr) {
        if (!gr->layer || !gr->pos_emb || !gr->cls_emb || !gr->lnf_w || !gr->lnf_b || !gr->text_proj || (d % 64) || d > 64 * kLnSlots)
            return VLSA_EINVAL;
        if (s.M > kPosRowsMax) return VLSA_EUNSUPPORTED;
        // text_projection [d, out_dim]: pooled^T dout;  ln_final: gamma / beta from d pooled and the CLS rows of x_final
        TT_TRY(launch_dw(DW_ROWS, c.pooled_t, dout, s.out_dim, (float*)gr->text_proj, nullptr, d, s.out_dim, s.n_seq, st));
        hipLaunchKernelGGL(k_tt_ln_rows, dim3((s.M + 3) / 4), dim3(256), 0, st, c.x_final, m->lnf_w, m->lnf_b, (float*)nullptr, c.lnstats, d, s.M);
        hipLaunchKernelGGL(k_tt_ln_param_bwd, dim3((d + 31) / 32), dim3(256), 0, st, c.dpool, c.x_final, c.lnstats, r->seq_row0, s.n_seq, d,
                           (float*)gr->lnf_w, (float*)gr->lnf_b);
        TT_LAUNCHED();
    }
    static DeviceOnce once;
    bool scattered = false;
    // Round 6: shared prefix, <= 128 compact rows (M; M_pad may be larger), and the extra L x heads workgroups still fit one round of the CUs: the prefix keys'
    // dK / dV come from a workgroup per (key, head) over all query rows (the forward's attention statistics + output) instead of the
    // ticketed fold of per-prompt shares (3.3 of the launch's 10.8 us: profiles/r06_attn_bwd_ablate.txt)
    // keys per key workgroup: the fewest (<= 4) that keep the grid within 256 workgroups
    int use_stats = 0;
    if (s.L > 0 && s.M <= 128 && s.heads <= 16)
        for (int kpb = 1; kpb <= 4 && !use_stats; ++kpb)
            if ((s.n_seq + 1 + (s.L + kpb - 1) / kpb) * s.heads <= 256) use_stats = kpb;
#ifdef VLSA_EXPERIMENT
    if (VLSA_ENV("VLSA_TT_NOSTATS")) use_stats = 0;
    if (VLSA_ENV("VLSA_TT_DEBUG_STATS")) fprintf(stderr, "[tt_backward] use_stats=%d L=%d Mp=%d heads=%d n_seq=%d\n", (int)use_stats, s.L, Mp, s.heads, s.n_seq);
#endif
    const size_t lds_ticket = (size_t)6 * kAttnBwdMaxS * (kHeadDim + 1) * sizeof(float);
    const size_t lds_stats = ((size_t)3 * 128 * (kHeadDim + 1) + 8 * 3 * 128 + 12 * 128) * sizeof(float);
    const size_t attn_lds = use_stats && lds_stats > lds_ticket ? lds_stats : lds_ticket;
    size_t attn_lds_x = attn_lds;
    int attn_grid_x = (s.n_seq + (s.L > 0 ? 1 : 0) + (use_stats ? (s.L + use_stats - 1) / use_stats : 0)) * s.heads;
#ifdef VLSA_EXPERIMENT
    if (const char* e = VLSA_ENV("VLSA_TT_ATTN_LDS")) attn_lds_x = (size_t)atoi(e);      // (timing only)
    if (const char* e = VLSA_ENV("VLSA_TT_ATTN_GRID")) attn_grid_x = atoi(e);
#endif
    const int attn_grid = attn_grid_x;
    if (once.first()) (void)hipFuncSetAttribute((const void*)k_tt_attn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_stats > lds_ticket ? lds_stats : lds_ticket));
    // Round 6: with frozen weights and the CONCH shapes (d = 768: a wave's 192-column slab = 12 groups; few rows: the 16 x 96 / 16 x 32
    // products) both LayerNorm backward passes of a block run as PROLOGUES of the products that consume their result (PRO_LNBWD):
    //   ln_2 backward  -> in front of  d attn = dx_mid W_out          (this block)
    //   ln_1 backward  -> in front of  d h = dx_out W_proj (* gelu')  (the block BELOW, i.e. the next iteration)
    // 5 launches per block instead of 7; only block 0's ln_1 backward stays a launch (its result feeds the scatter).
    const bool fuse = !gr && d / 4 / 16 == kSlabMax && (4 * d) % 96 == 0 && ((s.M + 15) / 16) * (4 * d / 96) <= 256 && d % 32 == 0
                      && ((s.M + 15) / 16) * (d / 32) <= 256;
    for (int L = s.layers - 1; fuse && L >= 0; --L) {
        const vlsa_tt_layer& w = m->layer[L];
        const PackedLayer pw = packed_layer(bset, s, L);
        float* x_in = ws + (size_t)L * LF;
        float* qkv = x_in + (size_t)Mp * d;
        float* x_mid = qkv + (size_t)Mp * 3 * d;
        float* h_pre = x_mid + (size_t)Mp * d;
        const bool top = L == s.layers - 1;
        {   // d h_pre = (dx_out @ W_proj) * gelu'(h_pre);  dx_out = the ln_final gradient (top block) or ln_1 backward of block L + 1
            GemmArgs a = gemm_args(top ? c.dxa_t : c.da_t, pw.proj_w, 4 * d, d);
            a.Yt = c.dh_t; a.H = h_pre; a.ldh = 4 * d; a.epi = EPI_GELU_BWD;
            a.M_real = s.M;
            prefetch_for(a, true, pw.fc_w, d, 4 * d, 2);
            prefetch_region(a, layer_xmid_t(x_in, s), (size_t)Mp * d);      // ln_2 backward's x, two launches ahead
            if (top) {
                TT_TRY((launch_gemm_g<1, 4, PRO_NONE, 12, 6>(a, Mp, st)));
            } else {
                a.X2 = layer_xin_t(ws + (size_t)(L + 1) * LF, s); a.R2 = c.dxb_t; a.ln_w = m->layer[L + 1].ln1_w;
                a.Yt2 = c.dxa_t; a.Y2 = c.dxa; a.ld2 = d;
                a.stats_in = layer_stats1(ws + (size_t)(L + 1) * LF, s);
                TT_TRY((launch_gemm_g<1, 4, PRO_LNBWD, 12, 6>(a, Mp, st)));
            }
        }
        {   // d ln_2 out = d h_pre @ W_fc   (tiled: the next product's prologue reads it)
            GemmArgs a = gemm_args(c.dh_t, pw.fc_w, d, 4 * d);
            a.Yt = c.da_t;
            prefetch_for(a, true, pw.out_w, d, d, 2);
            prefetch_region(a, qkv, (size_t)Mp * 3 * d);                     // the attention backward's q / k / v
            TT_TRY((launch_gemm_rows16<8, 24>(a, s.M, Mp, st)));
        }
        {   // dx_mid = dx_out + ln_2 backward (prologue);  d attn = dx_mid @ W_out
            GemmArgs a = gemm_args(c.da_t, pw.out_w, d, d);
            a.X2 = layer_xmid_t(x_in, s); a.R2 = c.dxa_t; a.ln_w = w.ln2_w; a.Yt2 = c.dxb_t; a.Y2 = c.dxb; a.ld2 = d;
            a.Y = c.dattn; a.ldy = d;
            a.M_real = s.M;
            a.stats_in = layer_stats2(x_in, s);
            prefetch_for(a, true, pw.in_w, d, 3 * d, 2);
            TT_TRY((launch_gemm_g<1, 4, PRO_LNBWD, 12, 2>(a, Mp, st)));
        }
        hipLaunchKernelGGL(k_tt_attn_bwd, dim3(attn_grid), dim3(attn_threads(r->max_len)), attn_lds_x, st, qkv, 3 * d, c.dattn, attn_bwd_ldo(d),
                           c.dqkv_t, r->seq_row0, r->cls_keep, s.heads, d, s.n_seq, s.L, c.pfx, c.cnt, layer_attn_t(x_in, s), layer_astats(x_in, s),
                           r->row_seq, s.M, use_stats);
        TT_LAUNCHED();
        {   // d ln_1 out = dqkv @ W_in
            GemmArgs a = gemm_args(c.dqkv_t, pw.in_w, d, 3 * d);
            a.Yt = c.da_t;
            if (L == 0) { a.Y = c.da; a.ldy = d; }
            if (L > 0) prefetch_for(a, true, packed_layer(bset, s, L - 1).proj_w, 4 * d, d, 6);
            if (L > 0) prefetch_region(a, layer_xin_t(x_in, s), (size_t)Mp * d);    // ln_1 backward's x (the next launch's prologue)
            TT_TRY((launch_gemm_rows16<8, 18>(a, s.M, Mp, st)));
        }
        if (L == 0) {
            // the pass's last LayerNorm backward also writes d prompts_embedding when that is a contiguous [n, ctx_len, d] block
            // (what autograd hands over): no memset, no scatter launch behind it
            scattered = d % 256 == 0 && d <= 256 * kLnSlots4 && emb_tok_stride == d && emb_seq_stride > 0 && emb_seq_stride % d == 0
                        && demb_floats % emb_seq_stride == 0 && (reinterpret_cast<uintptr_t>(demb) & 15) == 0;
            if (scattered) {
                ScatterArgs sc{demb, r->row_seq, r->row_src, emb_seq_stride, s.M, (int)(emb_seq_stride / d), (int)(demb_floats / d), (Mp + 3) / 4};
                hipLaunchKernelGGL(k_tt_ln_bwd4, dim3(sc.ln_blocks + (sc.demb_rows + 3) / 4), dim3(256), 0, st, c.da, x_in, w.ln1_w, c.dxb, c.dxa,
                                   c.dxa_t, d, Mp, sc);
            } else {
                launch_ln_bwd(c.da, x_in, w.ln1_w, c.dxb, c.dxa, c.dxa_t, d, Mp, st);
            }
            TT_LAUNCHED();
        }
    }
    for (int L = s.layers - 1; !fuse && L >= 0; --L) {
        const vlsa_tt_layer& w = m->layer[L];
        const PackedLayer pw = packed_layer(bset, s, L);
        float* x_in = ws + (size_t)L * LF;
        float* qkv = x_in + (size_t)Mp * d;
        float* x_mid = qkv + (size_t)Mp * 3 * d;
        float* h_pre = x_mid + (size_t)Mp * d;
        // (dxa, dxa_t) = gradient w.r.t. the layer's output.  MLP branch:
        {   // d h_pre = (dx @ W_proj) * gelu'(h_pre)
            GemmArgs a = gemm_args(c.dxa_t, pw.proj_w, 4 * d, d);
            a.Yt = c.dh_t; a.H = h_pre; a.ldh = 4 * d; a.epi = EPI_GELU_BWD;
            a.M_real = s.M;
            if ((4 * d) % 96 == 0 && ((s.M + 15) / 16) * (4 * d / 96) <= 256 && d / 4 / 16 == 12)
                TT_TRY((launch_gemm_g<1, 4, PRO_NONE, 12, 6>(a, Mp, st)));
            else if (Mp % 32 == 0 && (4 * d) % 64 == 0 && ((s.M + 31) / 32) * (4 * d / 64) <= 256 && d / 4 / 16 == 12)
                TT_TRY((launch_gemm_g<2, 4, PRO_NONE, 12, 4>(a, Mp, st)));
            else TT_TRY((launch_gemm_wide<3, 4, PRO_NONE, 12>(a, Mp, st)));
        }
        const vlsa_tt_layer* g = gr ? &gr->layer[L] : nullptr;
        const float* attn_t = h_pre + (size_t)Mp * 4 * d;      // (kept by the forward with save == 2)
        if (g) {   // c_proj [d, 4d]: dx_out^T gelu(h_pre);  c_fc [4d, d]: d h_pre^T ln_2(x_mid)
            const size_t n4 = (size_t)s.M * d;               // s.M rows x 4 d columns in float4s; c.hact_t: forward scratch, free here
            hipLaunchKernelGGL(k_tt_gelu_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, h_pre, c.hact_t, n4);
            TT_LAUNCHED();
            TT_TRY(launch_dw(DW_ROWS, c.dxa_t, c.hact_t, 4 * d, (float*)g->proj_w, (float*)g->proj_b, d, 4 * d, s.M, st));
            hipLaunchKernelGGL(k_tt_ln_rows, dim3((s.M + 3) / 4), dim3(256), 0, st, x_mid, w.ln2_w, w.ln2_b, c.lnout, c.lnstats, d, s.M);
            TT_LAUNCHED();
            TT_TRY(launch_dw(DW_ROWS, c.dh_t, c.lnout, d, (float*)g->fc_w, (float*)g->fc_b, 4 * d, d, s.M, st));
        }
        {   // d ln_2 out = d h_pre @ W_fc
            GemmArgs a = gemm_args(c.dh_t, pw.fc_w, d, 4 * d);
            a.Y = c.da; a.ldy = d;
            TT_TRY((launch_gemm_rows16<8, 24>(a, s.M, Mp, st)));
        }
        if (g) {
            hipLaunchKernelGGL(k_tt_ln_param_bwd, dim3((d + 31) / 32), dim3(256), 0, st, c.da, x_mid, c.lnstats, (const int*)nullptr, s.M, d,
                               (float*)g->ln2_w, (float*)g->ln2_b);
            TT_LAUNCHED();
        }
        launch_ln_bwd(c.da, x_mid, w.ln2_w, c.dxa, c.dxb, c.dxb_t, d, Mp, st);
        TT_LAUNCHED();
        if (g) {   // out_proj [d, d]: dx_mid^T attention output
            TT_TRY(launch_dw(DW_TILED, c.dxb_t, attn_t, 0, (float*)g->out_w, (float*)g->out_b, d, d, s.M, st));
        }
        // attention branch
        {   // d attn = dx_mid @ W_out
            GemmArgs a = gemm_args(c.dxb_t, pw.out_w, d, d);
            a.Y = c.dattn; a.ldy = d;
            TT_TRY((launch_gemm_rows16<4, 12>(a, s.M, Mp, st)));
        }
        hipLaunchKernelGGL(k_tt_attn_bwd, dim3(attn_grid), dim3(attn_threads(r->max_len)), attn_lds, st, qkv, 3 * d, c.dattn, d,
                           c.dqkv_t, r->seq_row0, r->cls_keep, s.heads, d, s.n_seq, s.L, c.pfx, c.cnt, layer_attn_t(x_in, s), layer_astats(x_in, s),
                           r->row_seq, s.M, use_stats);
        TT_LAUNCHED();
        {   // d ln_1 out = dqkv @ W_in   (rows M .. M_pad-1 of dqkv_t are never written: they only reach discarded padding rows)
            GemmArgs a = gemm_args(c.dqkv_t, pw.in_w, d, 3 * d);
            a.Y = c.da; a.ldy = d;
            TT_TRY((launch_gemm_rows16<8, 18>(a, s.M, Mp, st)));
        }
        if (g) {   // in_proj [3d, d]: dqkv^T ln_1(x_in);  ln_1 gamma / beta
            hipLaunchKernelGGL(k_tt_ln_rows, dim3((s.M + 3) / 4), dim3(256), 0, st, x_in, w.ln1_w, w.ln1_b, c.lnout, c.lnstats, d, s.M);
            TT_LAUNCHED();
            TT_TRY(launch_dw(DW_ROWS, c.dqkv_t, c.lnout, d, (float*)g->in_w, (float*)g->in_b, 3 * d, d, s.M, st));
            hipLaunchKernelGGL(k_tt_ln_param_bwd, dim3((d + 31) / 32), dim3(256), 0, st, c.da, x_in, c.lnstats, (const int*)nullptr, s.M, d,
                               (float*)g->ln1_w, (float*)g->ln1_b);
            TT_LAUNCHED();
        }
        launch_ln_bwd(c.da, x_in, w.ln1_w, c.dxb, c.dxa, c.dxa_t, d, Mp, st);
        TT_LAUNCHED();
    }
    if (gr) {   // positional_embedding [ctx_len, d] and cls_emb [d] from the gradient of the embedded rows
        hipLaunchKernelGGL(k_tt_pos_cls_bwd, dim3(m->ctx_len + 1), dim3(256), 0, st, c.dxa, r->row_pos, r->row_src, s.M, d, m->ctx_len,
                           (float*)gr->pos_emb, (float*)gr->cls_emb);
        TT_LAUNCHED();
    }
    if (scattered) return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
    if (hipMemsetAsync(demb, 0, (size_t)demb_floats * sizeof(float), st) != hipSuccess) return VLSA_ELAUNCH;
    hipLaunchKernelGGL(k_tt_scatter, dim3(s.M), dim3(256), 0, st, c.dxa, d, demb, emb_seq_stride, emb_tok_stride, r->row_seq, r->row_src, s.M);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}
}  // namespace

extern "C" int vlsa_tt_backward(const vlsa_tt_model* m, const vlsa_tt_rows* r, const void* packed, const float* dout, void* workspace,
                                float* demb, int64_t emb_seq_stride, int64_t emb_tok_stride, int64_t demb_floats, void* stream) {
    return tt_backward(m, r, packed, dout, workspace, demb, emb_seq_stride, emb_tok_stride, demb_floats, nullptr, stream);
}

extern "C" int vlsa_tt_backward_train(const vlsa_tt_model* m, const vlsa_tt_rows* r, const void* packed, const float* dout,
                                      void* workspace, float* demb, int64_t emb_seq_stride, int64_t emb_tok_stride,
                                      int64_t demb_floats, const vlsa_tt_model* grads, void* stream) {
    if (!grads) return VLSA_EINVAL;
    return tt_backward(m, r, packed, dout, workspace, demb, emb_seq_stride, emb_tok_stride, demb_floats, grads, stream);
}

// dL/dX of the VLFAN cross-attention aggregation: needed when the bag itself carries a gradient,
// i.e. when a trainable Feat_Projecter stands in front of the aggregation (use_feat_proj=True is the constructor default of the
// reference's encoders, model/deepmil.py:75,176-179).  Round 2 sent that configuration through torch ops that materialise
// F.normalize(X) and the [P, N] attention matrix; the streaming backward kernels produce only the query gradient.
//
// With s_pn = scale (e_p . x_n) / r_n, r_n = max(|x_n|, eps), A = softmax_n(s), out_p = sum_n A_pn x_n and upstream dout_p:
//     dA_pn = dout_p . x_n          dS_pn = A_pn (dA_pn - delta_p),  delta_p = dout_p . out_p
//     dx_n  = sum_p A_pn dout_p  +  (scale / r_n) sum_p dS_pn e_p  -  (sum_p dS_pn s_pn / r_n^2) x_n
// -- a pure map over the rows (nothing accumulates across rows): one more streaming pass, 2 KB read + 2 KB written per fp32 row.
//
//   * a wave owns 16-row tiles and keeps its tile IN REGISTERS: lane (row = l & 15, g = l >> 4) holds X[row][16 ct + 4 g .. + 3]
//     for the 32 column tiles ct (128 registers) -- which is at once the B operand of the two score-type contractions (exact
//     f32 MFMA 16x16x4, k-slot (step, g) <-> column 16 ct + 4 g + step: any bijection is a valid contraction order), the
//     operand of the -c_n x_n term, and the layout the result comes out in; the next tile's 32 loads are in flight in a second
//     register set while this one is processed (one wave per SIMD, 512-register budget).
//   * scores are computed TRANSPOSED, S^T[p][row] (A operand = the queries), so that the softmax weights land with the row
//     on the lane index and p = 4 g + reg in the registers: exactly the B operand [k = p][n = row] of the output contraction
//     dX^T[c][row] = sum_p dout[p][c] A_pn + e[p][c] u_pn (A operand: one LDS word per lane and step).  No transposes, no
//     exchange between waves.
//   * e_p and the current bag's dout_p live in LDS ([16][512] fp32 each, 16-byte chunks XOR-swizzled by p: conflict-free for
//     the ds_read_b128 of the score steps and the ds_read_b32 of the output steps); a workgroup walks 64-row super tiles of
//     the bag table (its 4 waves = 4 x 16 rows of the same bag) and reloads dout only when the bag changes.
// All products are exact f32 (v_mfma_f32_16x16x4_f32): 512 steps x 32 cycles per 16-row tile = 6.8 us per wave-tile, i.e.
// ~21 us per 50k-patch bag on 1024 waves -- about the HBM time of the 205 MB it moves (26 us at 8 TB/s).
#include "vlsa_common.h"

namespace vlsa {

typedef f32x4 __attribute__((may_alias)) f32x4_dx;
typedef float __attribute__((may_alias)) float_dx;

namespace dx {
constexpr int kD = 512;
constexpr int kQOff = 0;                  // e_p  [16][512] fp32, swizzled
constexpr int kDOff = 16 * kD * 4;        // dout [16][512] fp32, swizzled
constexpr int kLds = 2 * 16 * kD * 4;     // 64 KiB
}  // namespace dx

struct DxBag {
    const void* X;
    long long N, ldx;
};

struct DxArgs {
    const DxBag* bags;          // [B] fp32 rows the aggregation read (the projected bags)
    const DxBag* dxs;           // [B] fp32 gradient rows to write
    const int* tile_start;      // [B + 1] first 64-row super tile of every bag
    const float* qeff;          // [16][512] effective unit queries e_p (rows >= P zero)
    const float* dout;          // [B][P][512] upstream gradient of the aggregated rows
    const float* m2;            // [B][16] log2-domain softmax max
    const float* l;             // [B][16] softmax denominators
    const float* delta;         // [B][16] dout_p . out_p
    float scale;                // coattn scale (100)
    int B, P, n_tiles;
};

// byte offset of 16-byte chunk `ch` of row p in a swizzled [16][512] fp32 block
__device__ __forceinline__ int dx_chunk(int p, int ch) { return p * 2048 + ((ch ^ p) << 4); }

__global__ __launch_bounds__(256) void k_vlfan_dx(const DxArgs a) {
    using namespace dx;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int G = gridDim.x;
    const int s0 = (int)((long long)a.n_tiles * blockIdx.x / G), s1 = (int)((long long)a.n_tiles * (blockIdx.x + 1) / G);
    if (s0 >= s1) return;
    const float sc2 = a.scale * kLog2e;

    // e_p -> LDS once
    for (int i = tid; i < 16 * 128; i += 256) {
        const int p = i >> 7, ch = i & 127;
        *reinterpret_cast<f32x4_dx*>(smem + kQOff + dx_chunk(p, ch)) = *reinterpret_cast<const f32x4*>(a.qeff + p * kD + 4 * ch);
    }

    struct Tile { const float* x; float* dxo; long long ldx, lddx; int nrows, bag; };
    auto find = [&](int s) -> Tile {
        const int ts = lane < a.B ? a.tile_start[lane] : 0x7fffffff;
        const int b = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ts <= s)) - 1;
        const DxBag bag = a.bags[b], o = a.dxs[b];
        const long long row0 = (long long)(s - a.tile_start[b]) * 64 + 16 * w;
        Tile t;
        t.bag = b;
        t.ldx = bag.ldx;
        t.lddx = o.ldx;
        const long long left = bag.N - row0;
        t.nrows = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
        t.x = static_cast<const float*>(bag.X) + row0 * bag.ldx;
        t.dxo = static_cast<float*>(const_cast<void*>(o.X)) + row0 * o.ldx;
        return t;
    };
    // lane (row, g) loads X[row][16 ct + 4 g .. + 3], ct = 0..31; rows past the end re-read the tile's last row (masked later)
    auto load_tile = [&](const Tile& t, f32x4 (&x)[32]) {
        const int row = t.nrows > 0 ? (i16 < t.nrows ? i16 : t.nrows - 1) : 0;
        const float* src = t.x + (size_t)row * t.ldx + 4 * g;
        if (t.nrows > 0) {
#pragma unroll
            for (int ct = 0; ct < 32; ++ct) x[ct] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 16 * ct));
        } else {
#pragma unroll
            for (int ct = 0; ct < 32; ++ct) x[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x4 xa[32], xb[32];
    Tile cur = find(s0);
    load_tile(cur, xa);
    int bag_in_lds = -1;
    float m2p[4], rlp[4], dlt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) m2p[r] = rlp[r] = dlt[r] = 0.f;

    auto process = [&](const Tile& t, f32x4 (&x)[32]) {
        // ---- S^T[p][row] = e_p . x_row and dA^T[p][row] = dout_p . x_row; |x_row|^2 -------------------------------------------
        f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, Dd = f32x4{0.f, 0.f, 0.f, 0.f};
        float ss = 0.f;
#pragma unroll
        for (int ct = 0; ct < 32; ++ct) {
            const f32x4 q4 = *reinterpret_cast<const f32x4_dx*>(smem + kQOff + dx_chunk(i16, 4 * ct + g));
            const f32x4 d4 = *reinterpret_cast<const f32x4_dx*>(smem + kDOff + dx_chunk(i16, 4 * ct + g));
            const f32x4 x4 = x[ct];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                S = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[e], x4[e], S, 0, 0, 0);
                Dd = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[e], x4[e], Dd, 0, 0, 0);
                ss = fmaf(x4[e], x4[e], ss);
            }
        }
        ss = quad_rows_sum(ss);                                        // the 4 lane groups hold the 4 column quarters of the row
        const float rinv = fminf(__builtin_amdgcn_rsqf(ss), 1e12f);   // 1 / max(|x|, 1e-12)
        // ---- weights of the output contraction: lane (row, g), register r <-> p = 4 g + r ---------------------------------------
        float Aw[4], Uw[4], cn = 0.f;
        const bool rok = i16 < t.nrows;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 4 * g + r;
            const float sraw = S[r] * rinv;                            // cos-type score before the scale
            float A = (p < a.P && rok) ? fast_exp2(sraw * sc2 - m2p[r]) * rlp[r] : 0.f;
            const float dS = A * (Dd[r] - dlt[r]);
            Aw[r] = A;
            Uw[r] = dS * a.scale * rinv;
            cn += dS * a.scale * sraw;                                 // dS_pn s_pn
        }
        cn = quad_rows_sum(cn) * rinv * rinv;
        // ---- dX^T[c][row] = sum_p dout[p][c] A_p,row + e[p][c] u_p,row - c_row x[row][c] --------------------------------------
#pragma unroll
        for (int c2 = 0; c2 < 32; c2 += 2) {      // two column tiles x two products = four independent accumulator chains
            f32x4 accA[2], accU[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) accA[k] = accU[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 4 * g + r;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int off = dx_chunk(p, 4 * (c2 + k) + (i16 >> 2)) + (i16 & 3) * 4;
                    const float dv = *reinterpret_cast<const float_dx*>(smem + kDOff + off);
                    const float ev = *reinterpret_cast<const float_dx*>(smem + kQOff + off);
                    accA[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, Aw[r], accA[k], 0, 0, 0);
                    accU[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev, Uw[r], accU[k], 0, 0, 0);
                }
            }
            if (rok) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const f32x4 o = (accA[k] + accU[k]) - cn * x[c2 + k];
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(t.dxo + (size_t)i16 * t.lddx + 16 * (c2 + k) + 4 * g));
                }
            }
        }
    };

    for (int s = s0; s < s1; s += 2) {
        // (two super tiles per iteration so that the two register sets keep static names)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int sc = s + half;
            if (sc >= s1) break;
            Tile nxt = cur;
            const bool more = sc + 1 < s1;
            if (more) nxt = find(sc + 1);
            if (cur.bag != bag_in_lds) {          // workgroup-uniform: the super tile sequence is the same for the 4 waves
                __syncthreads();                  // everyone is done with the previous bag's dout
                const float* dsrc = a.dout + (size_t)cur.bag * a.P * kD;
                for (int i = tid; i < 16 * 128; i += 256) {
                    const int p = i >> 7, ch = i & 127;
                    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p < a.P) v = *reinterpret_cast<const f32x4*>(dsrc + p * kD + 4 * ch);
                    *reinterpret_cast<f32x4_dx*>(smem + kDOff + dx_chunk(p, ch)) = v;
                }
                __syncthreads();
                bag_in_lds = cur.bag;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = 4 * g + r;
                    const bool pok = p < a.P;
                    m2p[r] = pok ? a.m2[cur.bag * kPStride + p] : 0.f;
                    rlp[r] = pok ? 1.f / a.l[cur.bag * kPStride + p] : 0.f;
                    dlt[r] = pok ? a.delta[cur.bag * kPStride + p] : 0.f;
                }
            }
            if (half == 0) {
                if (more) load_tile(nxt, xb);
                process(cur, xa);
            } else {
                if (more) load_tile(nxt, xa);
                process(cur, xb);
            }
            cur = nxt;
        }
    }
}

// delta[b][p] = dout[b][p] . out[b][p]   (grid = B, 16 x 64 threads)
__global__ __launch_bounds__(1024) void k_dx_delta(const float* __restrict__ dout, const float* __restrict__ out, int P,
                                                    float* __restrict__ delta) {
    const int b = blockIdx.x, p = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    if (p < P) {
        const float* d = dout + ((size_t)b * P + p) * 512;
        const float* o = out + ((size_t)b * P + p) * 512;
        for (int i = lane; i < 512; i += 64) s += d[i] * o[i];
    }
    s = wave_sum(s);
    if (lane == 0) delta[b * kPStride + p] = p < P ? s : 0.f;
}

}  // namespace vlsa

using namespace vlsa;

// dL/dX of the cross attention for B bags (fp32 rows, D = 512).  bag_desc / dx_desc: device tables of vlsa_bag_desc (rows read /
// gradient rows written, fp32, 16-byte aligned); tile_start [B + 1] int32 (device): first 64-row super tile of every bag,
// n_tiles = tile_start[B]; qprep: the block of vlsa_prepare_queries (its effective unit queries are read); dout [B][P][512],
// out [B][P][512], m2 / l [B][16] of the forward; delta_ws: [B][16] floats of workspace.
extern "C" int vlsa_vlfan_backward_dx(const void* bag_desc, const void* dx_desc, int B, int D, const void* qprep, int P, float coattn_scale,
                                      const int* tile_start, int n_tiles, const float* dout, const float* out, const float* m2,
                                      const float* l, float* delta_ws, void* stream) {
    if (!bag_desc || !dx_desc || !qprep || !tile_start || !dout || !out || !m2 || !l || !delta_ws || B < 1 || B > 64 || n_tiles < 1)
        return VLSA_EINVAL;
    if (D != dx::kD) return VLSA_EUNSUPPORTED;
    if (P < 1 || P > VLSA_MAX_P) return VLSA_EINVAL;
    static DeviceOnce once;
    if (once.first()) (void)hipFuncSetAttribute((const void*)k_vlfan_dx, hipFuncAttributeMaxDynamicSharedMemorySize, dx::kLds);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dx_delta, dim3(B), dim3(1024), 0, st, dout, out, P, delta_ws);
    DxArgs a{};
    a.bags = static_cast<const DxBag*>(bag_desc);
    a.dxs = static_cast<const DxBag*>(dx_desc);
    a.tile_start = tile_start;
    const QPrepLayout L(D);
    a.qeff = reinterpret_cast<const float*>(static_cast<const unsigned char*>(qprep) + L.qeff);
    a.dout = dout;
    a.m2 = m2;
    a.l = l;
    a.delta = delta_ws;
    a.scale = coattn_scale;
    a.B = B;
    a.P = P;
    a.n_tiles = n_tiles;
    const int G = n_tiles < 256 ? n_tiles : 256;
    hipLaunchKernelGGL(k_vlfan_dx, dim3(G), dim3(256), dx::kLds, st, a);
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

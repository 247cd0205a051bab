// The optimizer of the training step (runner/vlsa_handler.py:283-289 with cfg_vlsa_conch.yaml:111-113: torch.optim.Adam, lr 2e-4, weight
// decay 1e-5 on the >= 2-D parameters, optim_factory.py:25-37) as ONE launch over all parameter tensors.
//
// Why it lives here: inside the hipGraph-replayed step (vlsa_amd/train_step.py) every dependent launch costs ~5 us whatever it does
// (profiles/r06_step_graph_kernel_stats.csv), and torch's fused Adam is four launches of 14.6 us for the step's six small tensors
// (0.28 M elements): 58 us of a 1.72 ms step.  One launch here; the step counter and the per-tensor learning rates live in device memory,
// so a captured launch replays correctly (torch's `capturable=True` keeps its counters on the device for the same reason).
//
// Arithmetic = torch.optim.Adam's single-tensor path (torch/optim/adam.py, amsgrad = False, maximize = False):
//   g += wd p;  m += (g - m) (1 - b1);  v = v b2 + (1 - b2) g g;  step += 1
//   p -= (lr / (1 - b1^step)) m / (sqrt(v) / sqrt(1 - b2^step) + eps)
// with the bias corrections evaluated in double precision (torch evaluates them in Python floats).
#include "vlsa_common.h"

namespace vlsa {

struct AdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    int blk0;        // first workgroup of this tensor
    int hyper;       // index into the device hyper table: {lr, weight_decay} per entry
};
struct AdamArgs {
    AdamTensor t[VLSA_ADAM_MAX_TENSORS];
    int n_tensors, bump;
    float beta1, beta2, eps, omb1, omb2;     // omb = 1 - beta rounded from DOUBLE (torch takes the betas as Python floats: 1 - 0.999 = 0.001, not
    double b1d, b2d;                         // the 9.99987e-4 of the fp32 difference)
};

constexpr int kAdamThreads = 256, kAdamPerThread = 4;

// b^n by squaring in double (n >= 1): at most 2 x 31 multiplications, each within half an ulp of double -- the bias corrections are
// rounded to fp32 afterwards.  (libm's pow is several hundred dependent fp64 instructions: ~1.5 us in front of every update.)
__device__ __forceinline__ double pow_int(double b, int n) {
    double r = 1.0;
    while (n > 0) {
        if (n & 1) r *= b;
        b *= b;
        n >>= 1;
    }
    return r;
}

__global__ __launch_bounds__(kAdamThreads) void k_adam_step(const AdamArgs a, const float* __restrict__ hyper, int* __restrict__ state) {
    // state[0] = steps taken so far, state[1] = arrival ticket of this launch (zero between launches)
    __shared__ int s_step;
    int ti = 0;
#pragma unroll 1
    for (int k = 1; k < a.n_tensors; ++k)
        if ((int)blockIdx.x >= a.t[k].blk0) ti = k;
    const AdamTensor T = a.t[ti];
    // order of the requests = order of the latencies: the step counter (a device-scope load), then this thread's elements, then the ticket --
    // the bias corrections are computed while the elements are in flight
    if (threadIdx.x == 0) s_step = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    const float lr = hyper[2 * T.hyper], wd = hyper[2 * T.hyper + 1];
    const long long base = ((long long)(blockIdx.x - T.blk0) * kAdamThreads + threadIdx.x) * kAdamPerThread;
    float gv[kAdamPerThread], pv[kAdamPerThread], mv[kAdamPerThread], vv[kAdamPerThread];
#pragma unroll
    for (int i = 0; i < kAdamPerThread; ++i) {
        const long long e = base + i;
        const bool ok = e < T.n;
        gv[i] = ok ? T.g[e] : 0.f;
        pv[i] = ok ? T.p[e] : 0.f;
        mv[i] = ok ? T.m[e] : 0.f;
        vv[i] = ok ? T.v[e] : 0.f;
    }
    __syncthreads();
    const int step = s_step;
    // the step counter moves once every workgroup of the launch has READ it -- a workgroup takes its ticket as soon as its read is back
    // (not behind its update), the last arriver writes the counter and zeroes the ticket
    if (a.bump && threadIdx.x == 0) {
        const int tk = __hip_atomic_fetch_add(state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == (int)gridDim.x - 1) {
            __hip_atomic_store(state + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(state, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const double bc1 = 1.0 - pow_int(a.b1d, step);
    const double bc2 = 1.0 - pow_int(a.b2d, step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
#pragma unroll
    for (int i = 0; i < kAdamPerThread; ++i) {
        const long long e = base + i;
        if (e < T.n) {
            float g = gv[i];
            const float p = pv[i];
            if (wd != 0.f) g = g + wd * p;
            float m = mv[i], v = vv[i];
            m = m + (g - m) * a.omb1;
            v = v * a.beta2 + a.omb2 * g * g;
            const float denom = sqrtf(v) / bc2_sqrt + a.eps;
            T.m[e] = m;
            T.v[e] = v;
            T.p[e] = p - step_size * (m / denom);
        }
    }
}

}  // namespace vlsa

using namespace vlsa;

extern "C" int vlsa_adam_step(const vlsa_adam_tensor* tensors, int n_tensors, const float* hyper, int* state, double beta1, double beta2,
                              double eps, void* stream) {
    if (!tensors || n_tensors < 1 || !hyper || !state) return VLSA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int k0 = 0; k0 < n_tensors; k0 += VLSA_ADAM_MAX_TENSORS) {
        AdamArgs a{};
        const int n = n_tensors - k0 < VLSA_ADAM_MAX_TENSORS ? n_tensors - k0 : VLSA_ADAM_MAX_TENSORS;
        int blk = 0;
        for (int i = 0; i < n; ++i) {
            const vlsa_adam_tensor& s = tensors[k0 + i];
            if (!s.param || !s.grad || !s.exp_avg || !s.exp_avg_sq || s.n < 0 || s.hyper < 0) return VLSA_EINVAL;
            a.t[i] = AdamTensor{s.param, s.grad, s.exp_avg, s.exp_avg_sq, (long long)s.n, blk, s.hyper};
            blk += (int)((s.n + kAdamThreads * kAdamPerThread - 1) / (kAdamThreads * kAdamPerThread));
        }
        a.n_tensors = n;
        a.bump = k0 + n >= n_tensors ? 1 : 0;     // the launch of the last chunk advances the step counter
        a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
        a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
        a.b1d = beta1; a.b2d = beta2;
        if (blk == 0) {      // (nothing but empty tensors in this chunk: the counter still moves with the last chunk)
            if (!a.bump) continue;
            blk = 1;
        }
        hipLaunchKernelGGL(k_adam_step, dim3(blk), dim3(kAdamThreads), 0, st, a, hyper, state);
        if (hipGetLastError() != hipSuccess) return VLSA_ELAUNCH;
    }
    return VLSA_OK;
}

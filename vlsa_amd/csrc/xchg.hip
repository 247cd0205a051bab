// Peer-write exchange of the patch-sharded multi-GPU path: one process per GPU, every rank exports ONE
// fine-grained device buffer (inbox + result box + flags) through hipIpc*, maps its peers' buffers once, and from then on the
// data path is kernels only -- no collective library, no host round trip:
//
//   k_xchg_put      block d: (optionally waits for peer d's acknowledgement of the slot's previous use), copies this rank's
//                   piece for peer d straight into d's buffer over xGMI, fences at system scope, raises d's flag for this rank.
//   k_xchg_wait     spins until every peer's flag for this slot has reached the epoch.
//   k_xchg_collect  block o: waits for owner o's result flag, scatters o's results into this rank's [B, ...] outputs
//                   (global bag order), acknowledges to o that the slot may be written again.
//
// Flags are 32-bit epochs (launch counter of the plan), compared as signed differences; every spin is bounded by a wall-clock
// timeout (wall_clock64: constant 100 MHz) and reports through a status word instead of hanging the queue -- bench.py's
// self-test reads it and falls back to the RCCL exchange.  Buffers are hipDeviceMallocUncached / fine-grained: a remote
// write must be visible to a kernel that is already running here (coarse-grained memory is only coherent at kernel boundaries).
// The reference has no counterpart ("NCCL call sites: none"); the payload is the per-query partial sums of
// model/deepmil.py:198-200 as compact records [m2(16) | l(16) | acc(P*D)].
#include <string.h>

#include "vlsa_common.h"

namespace vlsa {

constexpr int kXchgMaxPeers = 16;

struct XchgPut {
    const float* src[kXchgMaxPeers];     // local source of the piece for peer d
    float* dst[kXchgMaxPeers];           // where it goes in peer d's buffer (mapped address)
    unsigned int n16[kXchgMaxPeers];     // 16-byte units of the piece
    unsigned int* flag[kXchgMaxPeers];   // in peer d's buffer: raised to `epoch` after the piece has landed
    unsigned int* ack[kXchgMaxPeers];    // in peer d's buffer, may be null: also raised to `epoch` (acknowledges an EARLIER transfer d -> here)
    const unsigned int* gate[kXchgMaxPeers];  // LOCAL flag, may be null: wait until it has reached `gate_epoch` before writing
};

struct XchgWait {
    const unsigned int* flag[kXchgMaxPeers];  // local flags
};

struct XchgCollect {
    const float* box[kXchgMaxPeers];     // LOCAL result box of owner o: [logits(nmax*K) | incidence(nmax*K) | vhat(nmax*D) | m2(nmax*16) | l(nmax*16)]
    const unsigned int* flag[kXchgMaxPeers];  // local: owner o's results have landed
    unsigned int* ack[kXchgMaxPeers];    // in owner o's buffer: this rank has consumed them
    int count[kXchgMaxPeers];            // bags owner o owns
    int start[kXchgMaxPeers];            // first position of owner o's bags in the owner-major (local) bag order
};

__device__ __forceinline__ unsigned int load_sys(const unsigned int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_sys(unsigned int* p, unsigned int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// thread 0 of the block spins until *flag has reached `epoch` (signed difference: epochs wrap); false on time-out
__device__ __forceinline__ bool spin_until(const unsigned int* flag, unsigned int epoch, long long timeout_ticks) {
    const long long t0 = wall_clock64();
    for (;;) {
        if ((int)(load_sys(flag) - epoch) >= 0) return true;
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout_ticks) return false;
    }
}

__global__ __launch_bounds__(1024) void k_xchg_put(XchgPut a, unsigned int epoch, unsigned int gate_epoch, long long timeout_ticks,
                                                    unsigned int* __restrict__ status) {
    __shared__ int s_ok;
    const int d = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        bool ok = true;
        if (a.gate[d] != nullptr) ok = spin_until(a.gate[d], gate_epoch, timeout_ticks);
        if (!ok) atomicOr(status, 1u);
        // On a gate time-out NOTHING goes out: the owner may not have consumed the previous epoch yet, and overwriting its inbox /
        // result box would make IT broadcast a garbage merge with a clean status word.  No copy and no flag: the peer's own
        // wait on that flag then times out too (bounded), so every rank that could have read stale records carries a non-zero status,
        // and the host side (ShardedVlfanBatchPlan.finish) reduces the status words over the control group before anyone uses a result.
        s_ok = ok ? 1 : 0;
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    if (!s_ok) return;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.src[d]);
    f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(a.dst[d]);
    const unsigned int n = a.n16[d];
    for (unsigned int i = tid; i < n; i += 1024) {
        const f32x4 v = src[i];
        __builtin_nontemporal_store(v, dst + i);
    }
    __threadfence_system();   // every thread: its stores have reached the peer before the flag can be seen
    __syncthreads();
    if (tid == 0) {
        store_sys(a.flag[d], epoch);
        if (a.ack[d] != nullptr) store_sys(a.ack[d], epoch);
    }
}

__global__ __launch_bounds__(64) void k_xchg_wait(XchgWait a, int n, unsigned int epoch, long long timeout_ticks,
                                                   unsigned int* __restrict__ status) {
    const int r = threadIdx.x;
    if (r < n && !spin_until(a.flag[r], epoch, timeout_ticks)) atomicOr(status, 2u);
    __threadfence_system();
}

// results of owner o -> this rank's outputs in GLOBAL bag order: bag b = j * world + o for the j-th bag o owns
__global__ __launch_bounds__(256) void k_xchg_collect(XchgCollect a, int world, int nmax, int K, int D, unsigned int epoch,
                                                       long long timeout_ticks, float* __restrict__ logits,
                                                       float* __restrict__ incidence, float* __restrict__ vhat,
                                                       float* __restrict__ m2, float* __restrict__ l,
                                                       float* __restrict__ m2_loc, float* __restrict__ l_loc,
                                                       unsigned int* __restrict__ status) {
    const int o = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        if (a.flag[o] != nullptr && !spin_until(a.flag[o], epoch, timeout_ticks)) atomicOr(status, 4u);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    const float* box = a.box[o];
    const int n = a.count[o];
    const size_t nK = ((size_t)nmax * K + 3) & ~(size_t)3;   // sections start on 16-byte boundaries (vlsa_xchg_result_floats)
    const float* b_log = box;
    const float* b_inc = b_log + nK;
    const float* b_vh = b_inc + nK;
    const float* b_m2 = b_vh + (size_t)nmax * D;
    const float* b_l = b_m2 + (size_t)nmax * kPStride;
    for (int e = tid; e < n * K; e += 256) {
        const int j = e / K, k = e % K;
        const size_t g = (size_t)(j * world + o) * K + k;
        logits[g] = __builtin_nontemporal_load(b_log + e);
        if (incidence != nullptr) incidence[g] = __builtin_nontemporal_load(b_inc + e);
    }
    for (int e = tid; e < n * D; e += 256) {
        const int j = e / D, c = e % D;
        vhat[(size_t)(j * world + o) * D + c] = __builtin_nontemporal_load(b_vh + e);
    }
    for (int e = tid; e < n * kPStride; e += 256) {
        const int j = e / kPStride, p = e % kPStride;
        const size_t g = (size_t)(j * world + o) * kPStride + p;
        const float vm = __builtin_nontemporal_load(b_m2 + e), vl = __builtin_nontemporal_load(b_l + e);
        m2[g] = vm;
        l[g] = vl;
        if (m2_loc != nullptr) {   // the same in the owner-major order of the local bag table (attention-weight normalisation)
            m2_loc[(size_t)(a.start[o] + j) * kPStride + p] = vm;
            l_loc[(size_t)(a.start[o] + j) * kPStride + p] = vl;
        }
    }
    __syncthreads();
    if (tid == 0 && a.ack[o] != nullptr) store_sys(a.ack[o], epoch);
}

}  // namespace vlsa

using namespace vlsa;

static inline int xchg_status(hipError_t e) { return e == hipSuccess ? VLSA_OK : VLSA_ELAUNCH; }

extern "C" int vlsa_xchg_max_peers(void) { return kXchgMaxPeers; }

// floats of one owner's result box for `nmax` bags: [logits | incidence | vhat | m2 | l], every section 16-byte aligned;
// offsets5 (nullable): start of each section in floats
extern "C" size_t vlsa_xchg_result_floats(int nmax, int K, int D, int64_t* offsets5) {
    const size_t nK = ((size_t)nmax * K + 3) & ~(size_t)3;
    const size_t o[5] = {0, nK, 2 * nK, 2 * nK + (size_t)nmax * D, 2 * nK + (size_t)nmax * D + (size_t)nmax * kPStride};
    if (offsets5)
        for (int i = 0; i < 5; ++i) offsets5[i] = (int64_t)o[i];
    return o[4] + (size_t)nmax * kPStride;
}

// ---- set-up (once per plan; the only entry points of the library that allocate) ------------------------------------------
extern "C" int vlsa_xchg_alloc(size_t bytes, void** ptr, void* handle64, int* kind) {
    if (!ptr || !handle64 || bytes == 0) return VLSA_EINVAL;
    void* p = nullptr;
    int k = 2;   // 2 = uncached, 1 = fine-grained, 0 = plain device memory (coherent at kernel boundaries only: last resort)
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        k = 1;
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            k = 0;
            if (hipMalloc(&p, bytes) != hipSuccess) {
                (void)hipGetLastError();
                return VLSA_ELAUNCH;
            }
        }
    }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p);
        return VLSA_ELAUNCH;
    }
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return VLSA_EUNSUPPORTED;
    }
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
    memcpy(handle64, &h, 64);
    *ptr = p;
    if (kind) *kind = k;
    return VLSA_OK;
}

extern "C" int vlsa_xchg_open(const void* handle64, void** ptr) {
    if (!handle64 || !ptr) return VLSA_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return VLSA_EUNSUPPORTED;
    }
    *ptr = p;
    return VLSA_OK;
}

extern "C" int vlsa_xchg_close(void* ptr) { return ptr ? xchg_status(hipIpcCloseMemHandle(ptr)) : VLSA_EINVAL; }
extern "C" int vlsa_xchg_free(void* ptr) { return ptr ? xchg_status(hipFree(ptr)) : VLSA_EINVAL; }

// ---- data path (enqueue only) ------------------------------------------------------------------------------------------------
// The pointer tables are HOST arrays of `world` entries each (device addresses inside): they travel as kernel arguments.
extern "C" int vlsa_xchg_put(int world, const void* const* src, void* const* dst, const uint32_t* n16, void* const* flag,
                             void* const* ack, const void* const* gate, uint32_t epoch, uint32_t gate_epoch, int64_t timeout_ticks,
                             void* status, void* stream) {
    if (world < 1 || world > kXchgMaxPeers || !src || !dst || !n16 || !flag || !status) return VLSA_EINVAL;
    XchgPut a{};
    for (int d = 0; d < world; ++d) {
        if (!src[d] || !dst[d] || !flag[d]) return VLSA_EINVAL;
        if ((reinterpret_cast<uintptr_t>(src[d]) & 15) || (reinterpret_cast<uintptr_t>(dst[d]) & 15)) return VLSA_EINVAL;
        a.src[d] = static_cast<const float*>(src[d]);
        a.dst[d] = static_cast<float*>(dst[d]);
        a.n16[d] = n16[d];
        a.flag[d] = static_cast<unsigned int*>(flag[d]);
        a.ack[d] = ack ? static_cast<unsigned int*>(ack[d]) : nullptr;
        a.gate[d] = gate ? static_cast<const unsigned int*>(gate[d]) : nullptr;
    }
    hipLaunchKernelGGL(k_xchg_put, dim3(world), dim3(1024), 0, (hipStream_t)stream, a, epoch, gate_epoch, (long long)timeout_ticks,
                       static_cast<unsigned int*>(status));
    return xchg_status(hipGetLastError());
}

extern "C" int vlsa_xchg_wait(int world, const void* const* flag, uint32_t epoch, int64_t timeout_ticks, void* status, void* stream) {
    if (world < 1 || world > kXchgMaxPeers || !flag || !status) return VLSA_EINVAL;
    XchgWait a{};
    for (int r = 0; r < world; ++r) {
        if (!flag[r]) return VLSA_EINVAL;
        a.flag[r] = static_cast<const unsigned int*>(flag[r]);
    }
    hipLaunchKernelGGL(k_xchg_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, a, world, epoch, (long long)timeout_ticks,
                       static_cast<unsigned int*>(status));
    return xchg_status(hipGetLastError());
}

extern "C" int vlsa_xchg_collect(int world, const void* const* box, const void* const* flag, void* const* ack, const int* count,
                                 const int* start, int nmax, int K, int D, uint32_t epoch, int64_t timeout_ticks, float* logits,
                                 float* incidence, float* vhat, float* m2, float* l, float* m2_local, float* l_local, void* status,
                                 void* stream) {
    if (world < 1 || world > kXchgMaxPeers || !box || !count || !logits || !vhat || !m2 || !l || !status) return VLSA_EINVAL;
    if (nmax < 1 || K < 1 || K > VLSA_MAX_K || D < 1 || D > VLSA_MAX_D) return VLSA_EINVAL;
    if ((m2_local == nullptr) != (l_local == nullptr) || (m2_local && !start)) return VLSA_EINVAL;
    XchgCollect a{};
    for (int o = 0; o < world; ++o) {
        if (!box[o] || count[o] < 0 || count[o] > nmax) return VLSA_EINVAL;
        a.box[o] = static_cast<const float*>(box[o]);
        a.flag[o] = flag ? static_cast<const unsigned int*>(flag[o]) : nullptr;
        a.ack[o] = ack ? static_cast<unsigned int*>(ack[o]) : nullptr;
        a.count[o] = count[o];
        a.start[o] = start ? start[o] : 0;
    }
    hipLaunchKernelGGL(k_xchg_collect, dim3(world), dim3(256), 0, (hipStream_t)stream, a, world, nmax, K, D, epoch,
                       (long long)timeout_ticks, logits, incidence, vhat, m2, l, m2_local, l_local, static_cast<unsigned int*>(status));
    return xchg_status(hipGetLastError());
}

// HBM read ceiling of THIS box, for the roofline fraction of the streaming kernel to be read against (profiles/README.md):
//   mode 0  the production kernel's access pattern and machinery with no arithmetic: 256 persistent workgroups x 8 waves, every
//           wave an LDS-DMA ring of two 8 KiB slots (`buffer_load_dwordx4 ... offen nt lds`), a tile = 32 rows x 256 B of a row-major
//           [rows, 1 KiB] matrix (wave cw = column quarter cw, as k_vlfan_partial_dma_batch), counted vmcnt(8), one ds_read per tile;
//   mode 1  the same ring on CONTIGUOUS 8 KiB pieces per wave;
//   mode 2  plain `global_load_dwordx4` with the nontemporal hint, grid-stride, 8 loads in flight per lane, xor-reduced.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/hbm_read_probe.hip -o tools/probes/libhbm_read_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

// wout != nullptr (round 5): waves cw = 0 / 1 of every row group also STORE the production kernel's score lines -- a [12, rows] fp32
// matrix, per 32-row tile one 128-byte line per query (48 B per 1 KiB row = 4.7 % of the read stream), whole lines per instruction --
// so that the machine's own price of that write stream next to the read stream can be read off without any arithmetic.
__global__ __launch_bounds__(512, 2) void k_read_dma(const unsigned char* __restrict__ base, long long rows, int mode, int bar, int hold, int* __restrict__ sink,
                                                      float* __restrict__ wout = nullptr, int wmode = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = w >> 2, cw = w & 3;
    unsigned char* ring = smem + w * 16384;
    const unsigned int ring_lds = (unsigned int)(uintptr_t)(lds_ptr)ring;
    // rows of 1 KiB; workgroup b streams rows [b * per, (b + 1) * per) in 64-row iterations (row group rg takes 32 of them)
    const long long per = (rows / gridDim.x) & ~63ll;
    const long long r0 = (long long)blockIdx.x * per;
    const int niter = (int)(per >> 6);
    const unsigned long long addr = (unsigned long long)(uintptr_t)base + (unsigned long long)r0 * 1024ull;
    i32x4 rsrc = {(int)(unsigned int)addr, (int)((addr >> 32) & 0xffffu), (int)(per * 1024), 0x00020000};
    int voff;
    if (mode == 0) voff = (lane >> 4) * 1024 + cw * 256 + (lane & 15) * 16;      // 4 rows x 256 B per instruction
    else voff = lane * 16;                                                            // 1 KiB contiguous per instruction
    auto issue = [&](int it, int slot) {
        // mode 0: tile = rows [64 it + 32 rg, + 32), piece i = rows 4 i .. 4 i + 3;  mode 1: the wave's own contiguous 8 KiB
        const int sbase = mode == 0 ? (it * 64 + rg * 32) * 1024 : (it * 8 + w) * 8192;
        const int pstep = mode == 0 ? 4 * 1024 : 1024;
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
                : "s"(ring_lds + slot * 8192 + i * 1024), "v"(voff), "s"(rsrc), "s"(sbase + i * pstep)
                : "memory");
        }
    };
    int acc = 0;
    if (niter > 0) issue(0, 0);
    for (int it = 0; it < niter; ++it) {
        const int slot = it & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it + 1 < niter) {
            issue(it + 1, slot ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        acc ^= *reinterpret_cast<const int*>(ring + slot * 8192 + lane * 4);
        if (wout != nullptr && cw < 2) {
            const int q = 8 * cw + (lane >> 3);                       // query row of this lane (cw = 1: queries 8 .. 11 only)
            if (q < 12) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 v = {(float)acc, 1.f, 2.f, 3.f};
                float* dst = wout + (size_t)q * rows + (size_t)(r0 + it * 64 + rg * 32 + 4 * (lane & 7));
                if (wmode == 1) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
                else *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        for (int h = 0; h < (hold & 15); ++h) __builtin_amdgcn_s_sleep(4);   // the slot stays busy for ~256 cycles per unit ("arithmetic")
        if (hold & 16) {      // the product's LDS read volume: the tile twice (row-major fragments + transposed fragments) = 32 x ds_read_b128
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) t += *reinterpret_cast<const f32x4*>(ring + slot * 8192 + ((i * 1024 + lane * 16 + rep * 4096) & 8191));
            acc ^= __float_as_int(t[0] + t[1] + t[2] + t[3]);
        }
        if (hold & 32) {      // the product's matrix-pipe load: 48 x v_mfma_f32_16x16x32_bf16 per wave and iteration, on 6 accumulators
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            bf16x8 a, b2;
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b2[e] = (__bf16)(float)(acc & 3); }
            f32x4 c[6];
            for (int q = 0; q < 6; ++q) c[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int q = 0; q < 6; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2, c[q], 0, 0, 0);
            float r = 0.f;
            for (int q = 0; q < 6; ++q) r += c[q][0];
            acc ^= __float_as_int(r);
        }
        if (bar == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else if (bar == 2) {      // the product's phase structure: work, barrier, exchange write, barrier, exchange read + work
            for (int h = 0; h < 3; ++h) __builtin_amdgcn_s_sleep(4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            *reinterpret_cast<int*>(smem + 8 * 16384 + w * 2048 + lane * 4) = acc;          // (only with the larger LDS allocation)
            for (int h = 0; h < 2; ++h) __builtin_amdgcn_s_sleep(4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int ww = 0; ww < 4; ++ww) acc ^= *reinterpret_cast<const int*>(smem + 8 * 16384 + (rg * 4 + ww) * 2048 + lane * 4);
            for (int h = 0; h < 6; ++h) __builtin_amdgcn_s_sleep(4);
        }
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}


// 4 independent waves per workgroup, each streaming WHOLE rows: tile = 16 rows x 1 KiB (16 KiB slot, two slots per wave), no barriers
__global__ __launch_bounds__(256) void k_read_rows4(const unsigned char* __restrict__ base, long long rows, int* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* ring = smem + w * 32768;
    const unsigned int ring_lds = (unsigned int)(uintptr_t)(lds_ptr)ring;
    const long long per = (rows / gridDim.x) & ~63ll;
    const long long r0 = (long long)blockIdx.x * per;
    const int niter = (int)(per >> 6);                       // 64 rows per workgroup iteration: 16 per wave
    const unsigned long long addr = (unsigned long long)(uintptr_t)base + (unsigned long long)r0 * 1024ull;
    i32x4 rsrc = {(int)(unsigned int)addr, (int)((addr >> 32) & 0xffffu), (int)(per * 1024), 0x00020000};
    const int voff = lane * 16;                              // one instruction = one whole row (1 KiB)
    auto issue = [&](int it, int slot) {
        const int sbase = (it * 64 + w * 16) * 1024;
        unsigned int keep;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %1\n\t"
                "s_nop 0\n\t"
                "buffer_load_dwordx4 %2, %3, %4 offen nt lds\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "s"(ring_lds + slot * 16384 + i * 1024), "v"(voff), "s"(rsrc), "s"(sbase + i * 1024)
                : "memory");
        }
    };
    int acc = 0;
    if (niter > 0) issue(0, 0);
    for (int it = 0; it < niter; ++it) {
        const int slot = it & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it + 1 < niter) {
            issue(it + 1, slot ^ 1);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        acc ^= *reinterpret_cast<const int*>(ring + slot * 16384 + lane * 4);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}

__global__ __launch_bounds__(512) void k_read_plain(const i32x4* __restrict__ p, long long n16, int* __restrict__ sink) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    i32x4 a = {0, 0, 0, 0};
    for (; i + 7 * stride < n16; i += 8 * stride) {
        i32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) a ^= v[u];
    }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x7fffffff) sink[0] = a.x;
}

extern "C" int hbm_read_probe_launch(const void* base, long long bytes, int mode, void* sink, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (mode <= 1 || mode >= 4) {
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute((const void*)k_read_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384 + 8 * 2048);
            once = true;
        }
        hipLaunchKernelGGL(k_read_dma, dim3(256), dim3(512), 8 * 16384 + 8 * 2048, s, static_cast<const unsigned char*>(base), bytes / 1024, mode & 1,
                           (mode >> 2) & 3, mode >> 4, static_cast<int*>(sink));
    } else if (mode == 3) {
        static bool once4 = false;
        if (!once4) {
            (void)hipFuncSetAttribute((const void*)k_read_rows4, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
            once4 = true;
        }
        hipLaunchKernelGGL(k_read_rows4, dim3(256), dim3(256), 4 * 32768, s, static_cast<const unsigned char*>(base), bytes / 1024,
                           static_cast<int*>(sink));
    } else {
        hipLaunchKernelGGL(k_read_plain, dim3(256 * 4), dim3(512), 0, s, static_cast<const i32x4*>(base), bytes / 16, static_cast<int*>(sink));
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// read stream of mode (see hbm_read_probe_launch) + the score-line write stream into wout [12, bytes / 1024] fp32; wmode 1 = nontemporal stores
extern "C" int hbm_rw_probe_launch(const void* base, long long bytes, int mode, void* sink, void* wout, int wmode, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)k_read_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384 + 8 * 2048);
        once = true;
    }
    hipLaunchKernelGGL(k_read_dma, dim3(256), dim3(512), 8 * 16384 + 8 * 2048, s, static_cast<const unsigned char*>(base), bytes / 1024, mode & 1,
                       (mode >> 2) & 3, mode >> 4, static_cast<int*>(sink), static_cast<float*>(wout), wmode);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Bag ingest: the reference loads one fp32 [N, D] feature file per slide, concatenates a patient's
// slides on the host and copies the result to the device per step (dataset/PatchWSI.py:205-215, utils/io.py:30-31,
// runner/vlsa_handler.py:205,324).  Here bags live in HBM for the whole run (288 GB holds thousands of slides as bf16);
// this kernel packs freshly copied rows into the resident arena: fp32 -> bf16 round-to-nearest-even (== torch's
// .to(torch.bfloat16) for finite values and infinities), or a strided bf16 copy.  Pure streaming: 16-byte loads and stores.
#include "vlsa_common.h"

namespace vlsa {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int pack_bf16_rne(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned int, v);
}

// one thread per 8 consecutive elements of a row; D % 8 == 0
__global__ __launch_bounds__(256) void k_pack_rows_f32(const float* __restrict__ src, int64_t N, int64_t lds, int D,
                                                        __bf16* __restrict__ dst, int64_t ldd) {
    const int per_row = D >> 3;
    const int64_t total = N * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row) << 3;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + r * lds + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + r * lds + c + 4);
        u32x4_t o;
        o[0] = pack_bf16_rne(a[0], a[1]);
        o[1] = pack_bf16_rne(a[2], a[3]);
        o[2] = pack_bf16_rne(b[0], b[1]);
        o[3] = pack_bf16_rne(b[2], b[3]);
        *reinterpret_cast<u32x4_t*>(dst + r * ldd + c) = o;
    }
}

__global__ __launch_bounds__(256) void k_pack_rows_bf16(const __bf16* __restrict__ src, int64_t N, int64_t lds, int D,
                                                         __bf16* __restrict__ dst, int64_t ldd) {
    const int per_row = D >> 3;
    const int64_t total = N * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row) << 3;
        *reinterpret_cast<u32x4_t*>(dst + r * ldd + c) = *reinterpret_cast<const u32x4_t*>(src + r * lds + c);
    }
}

}  // namespace vlsa

using namespace vlsa;

extern "C" int vlsa_pack_rows_bf16(const void* src, int src_dtype, int64_t N, int64_t lds, int D, void* dst, int64_t ldd,
                                   void* stream) {
    if (!src || !dst || N < 0 || D < 8 || (D % 8) != 0 || lds < D || ldd < D) return VLSA_EINVAL;
    if (N == 0) return VLSA_OK;
    const int esz = src_dtype == VLSA_DT_F32 ? 4 : 2;
    if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15) || ((lds * esz) % 16) || ((ldd * 2) % 16))
        return VLSA_EINVAL;
    const int64_t total = N * (D >> 3);
    int64_t nb = (total + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    if (src_dtype == VLSA_DT_F32)
        hipLaunchKernelGGL(k_pack_rows_f32, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(src), N, lds, D,
                           static_cast<__bf16*>(dst), ldd);
    else if (src_dtype == VLSA_DT_BF16)
        hipLaunchKernelGGL(k_pack_rows_bf16, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, static_cast<const __bf16*>(src), N, lds,
                           D, static_cast<__bf16*>(dst), ldd);
    else
        return VLSA_EINVAL;
    return hipGetLastError() == hipSuccess ? VLSA_OK : VLSA_ELAUNCH;
}

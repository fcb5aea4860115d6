// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the GDR-Net RoI path.
// Element type T is either float (parity mode, fp32 MFMA) or bf16_t (throughput mode, bf16 MFMA,
// fp32 accumulate).  All activations are NHWC.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#include "../../include/gdrn_hip.h"  // status codes (GDRN_OK, GDRN_ERR_*) and dtype selectors (GDRN_DT_*)

#define GDRN_CHECK_LAUNCH()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return GDRN_ERR_LAUNCH;       \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even, on the gfx950 converter (v_cvt_pk_bf16_f32: one instruction per PAIR; the
// integer-arithmetic version cost ~7 VALU instructions per element in every bf16-writing epilogue)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef float f32x2_cvt_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_cvt_t __attribute__((ext_vector_type(2)));
    const f32x2_cvt_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_cvt_t));
}

__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// 16-byte vector access of VEC elements of T as floats.
template <typename T>
struct Vec16;

template <>
struct Vec16<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ void load(const float* p, float* v) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ static __forceinline__ void unpack(uint4 u, float* v) {
        v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* v) {
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
};

template <>
struct Vec16<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void unpack(uint4 u, float* v) {
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
        v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    __device__ static __forceinline__ uint4 pack(const float* v) {
        return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
    __device__ static __forceinline__ void load(const bf16_t* p, float* v) { unpack(*reinterpret_cast<const uint4*>(p), v); }
    __device__ static __forceinline__ void store(bf16_t* p, const float* v) { *reinterpret_cast<uint4*>(p) = pack(v); }
};

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// sum over the 16 lanes of a DPP row (every lane gets the total): quad xor 1, quad xor 2, half-row mirror, row mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for 256-thread blocks; result valid in every thread. `red` is >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

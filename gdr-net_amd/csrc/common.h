// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the GDR-Net RoI path.
// Element type T is either float (parity mode, fp32 MFMA) or bf16_t = the 16-bit storage type of THIS BUILD (throughput mode, 16-bit
// MFMA operands, fp32 accumulate).  All activations are NHWC.
//
// The 16-bit format is a property of the library build, not of a template parameter: the same sources are compiled twice --
//   libgdrn_hip.so      16-bit type = bfloat16  (dtype code GDRN_DT_BF16; v_mfma_f32_*_bf16, v_cvt_pk_bf16_f32)
//   libgdrn_hip_f16.so  16-bit type = IEEE half (dtype code GDRN_DT_F16, -DGDRN_HALF_F16; v_mfma_f32_*_f16, v_cvt_pk_f16_f32: the
//                       reference's fp16 autocast, main_gdrn.py:53-56,141 / gdrn_evaluator.py:568; same MFMA rate, 11 instead of 8
//                       significand bits) --
// and everything format-specific lives in this header: the storage <-> fp32 conversions (h16lo / h16hi / bf2f / pack_bf2), the MFMA
// builtins (GDRN_MFMA16 / GDRN_MFMA32), their assembler mnemonic (GDRN_MFMA16_ASM), the transposing LDS read (GDRN_TR16) and the
// packed constant 1.0 (GDRN_H16_ONE2).  The names bf16_t / bf16x8_t keep their round-1 spelling in the kernels: read "the build's half".
// GDRN_DT_H16 is the dtype code this build accepts for its 16-bit kernels (the other one is GDRN_ERR_ARG).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bits of the build's 16-bit format
#ifdef GDRN_HALF_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4_t;
#define GDRN_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0)
#define GDRN_MFMA32(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)
#define GDRN_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 gdrn_fp16x4_t;   // the builtin's own element type
typedef __attribute__((address_space(3))) gdrn_fp16x4_t gdrn_lds_fp16x4_t;
#define GDRN_TR16(p_) __builtin_bit_cast(bf16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4f16((gdrn_lds_fp16x4_t*)(p_)))
#define GDRN_H16_ONE2 0x3c003c00u
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
#define GDRN_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b_, c_, 0, 0, 0)
#define GDRN_MFMA32(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)
#define GDRN_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#define GDRN_TR16(p_) __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p_)
#define GDRN_H16_ONE2 0x3f803f80u
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#include "../../include/gdrn_hip.h"  // status codes (GDRN_OK, GDRN_ERR_*) and dtype selectors (GDRN_DT_*)
#ifdef GDRN_HALF_F16
#define GDRN_DT_H16 GDRN_DT_F16
#else
#define GDRN_DT_H16 GDRN_DT_BF16
#endif

// Every kernel launch of the library goes through GDRN_LAUNCH + GDRN_CHECK_LAUNCH.  hipLaunchKernelGGL returns nothing: the launch status is what
// hipGetLastError() says behind it -- and that call ALSO returns (and clears) an error some earlier HIP call of this host thread left behind,
// the host framework's included (a pointer-attribute probe on a host pointer, a failed attribute query of another library, ...): r5's smoke run
// reported "launch failure" for the first launch of a process that way.  The stale error is not this library's: drop it before the launch.
// (the dropped code is kept -- gdrn_tls_stale_hip_error, reported by gdrn_last_hip_error as "stale:<name>" -- instead of vanishing, and
//  every launch starts with a clean slot for its own status, so an old failure cannot be attached to a later, unrelated one)
extern thread_local int gdrn_tls_stale_hip_error;
#define GDRN_LAUNCH(...)                                             \
    do {                                                             \
        const hipError_t stale__ = hipGetLastError();                \
        if (stale__ != hipSuccess) gdrn_tls_stale_hip_error = (int)stale__; \
        gdrn_tls_hip_error = 0;                                      \
        hipLaunchKernelGGL(__VA_ARGS__);                             \
    } while (0)

// (the HIP error code behind the most recent GDRN_ERR_LAUNCH of this host thread, for the caller's error message: gdrn_last_hip_error)
extern thread_local int gdrn_tls_hip_error;
#define GDRN_CHECK_LAUNCH()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) {                             \
            gdrn_tls_hip_error = (int)e__;                   \
            return GDRN_ERR_LAUNCH;                          \
        }                                                    \
    } while (0)

#ifdef GDRN_HALF_F16
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// the low / high 16-bit element of a packed pair as fp32 (v_cvt_f32_f16, the high one through SDWA)
__device__ __forceinline__ float h16lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float h16hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
// fp32 -> fp16 pair, round-to-nearest-even (values beyond 65504 become inf, as torch's .half())
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef float f32x2_cvt_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_cvt_t __attribute__((ext_vector_type(2)));
    const f32x2_cvt_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_cvt_t));
}
#else
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float h16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// fp32 -> bf16, round-to-nearest-even, on the gfx950 converter (v_cvt_pk_bf16_f32: one instruction per PAIR; the
// integer-arithmetic version cost ~7 VALU instructions per element in every bf16-writing epilogue)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef float f32x2_cvt_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_cvt_t __attribute__((ext_vector_type(2)));
    const f32x2_cvt_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_cvt_t));
}
#endif

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
        v[0] = h16lo(u.x); v[1] = h16hi(u.x);
        v[2] = h16lo(u.y); v[3] = h16hi(u.y);
        v[4] = h16lo(u.z); v[5] = h16hi(u.z);
        v[6] = h16lo(u.w); v[7] = h16hi(u.w);
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

// One row of a parameter tensor through the fused Ranger update (lib/torch_utils/solver/ranger.py:100-200): gradient centralisation (row mean),
// RAdam moments, update, optional lookahead; p, g, m, v, slow point at the row.  Shared by the per-tensor and the multi-tensor kernel -- the two
// must stay bit-identical (tests/test_e2e_gpu.py::test_ranger_follows_an_lr_schedule_without_rebuilding_its_table).
// (r6b) rows of whole, 16-byte-aligned float4s -- every conv / fc weight of the path except the stem's -- move as 16-byte vectors, and the row's
// gradient is read ONCE: the first 16 values per thread (rows up to 4096 columns) stay in registers between the centralisation pass and the update
// (it was read twice with 4-byte accesses: 2.6 TB/s over the step's 21 M parameters).
__device__ __forceinline__ void ranger_row(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                           float* __restrict__ slow, int cols, int gc, float lr, float beta1, float beta2, float eps, float wd,
                                           float step_size, int adaptive, int lookahead, float alpha, float grad_scale, float* red) {
    auto one = [&](float gj, float vj, float mj, float pj, float sj, float mean, float& vo, float& mo, float& po, float& so) {
        const float gr = (gj - mean) * grad_scale;  // grad_scale = 1/world: the gradient buffer holds the all-reduced SUM
        const float vv = vj * beta2 + (1.f - beta2) * gr * gr;
        const float mm = mj * beta1 + (1.f - beta1) * gr;
        float pp = pj;
        if (wd != 0.f) pp += -wd * lr * pp;
        if (adaptive) pp += -step_size * lr * mm / (sqrtf(vv) + eps);
        else pp += -step_size * lr * mm;
        float sl = 0.f;
        if (lookahead) { sl = sj + alpha * (pp - sj); pp = sl; }
        vo = vv; mo = mm; po = pp; so = sl;
    };
    if ((cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                             reinterpret_cast<uintptr_t>(slow)) & 15) == 0) {
        const int c4 = cols >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float4 gq[4];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u;
            gq[u] = i < c4 ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (gq[u].x + gq[u].y) + (gq[u].z + gq[u].w);
        }
        float mean = 0.f;
        if (gc) {
            for (int i = threadIdx.x + 1024; i < c4; i += 256) { const float4 q = g4[i]; s += (q.x + q.y) + (q.z + q.w); }
            mean = block_sum_256(s, red) / (float)cols;
        }
        float4* p4 = reinterpret_cast<float4*>(p);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        float4* s4 = reinterpret_cast<float4*>(slow);
        auto upd = [&](int i, float4 gv) {
            const float4 vq = v4[i], mq = m4[i], pq = p4[i];
            float4 sq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lookahead) sq = s4[i];
            float4 vo, mo, po, so;
            one(gv.x, vq.x, mq.x, pq.x, sq.x, mean, vo.x, mo.x, po.x, so.x);
            one(gv.y, vq.y, mq.y, pq.y, sq.y, mean, vo.y, mo.y, po.y, so.y);
            one(gv.z, vq.z, mq.z, pq.z, sq.z, mean, vo.z, mo.z, po.z, so.z);
            one(gv.w, vq.w, mq.w, pq.w, sq.w, mean, vo.w, mo.w, po.w, so.w);
            v4[i] = vo;
            m4[i] = mo;
            if (lookahead) s4[i] = so;
            p4[i] = po;
        };
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u;
            if (i < c4) upd(i, gq[u]);
        }
        for (int i = threadIdx.x + 1024; i < c4; i += 256) upd(i, g4[i]);
        return;
    }
    float mean = 0.f;
    if (gc) {
        float s = 0.f;
        for (int i = threadIdx.x; i < cols; i += 256) s += g[i];
        mean = block_sum_256(s, red) / (float)cols;
    }
    for (int i = threadIdx.x; i < cols; i += 256) {
        float vo, mo, po, so;
        one(g[i], v[i], m[i], p[i], lookahead ? slow[i] : 0.f, mean, vo, mo, po, so);
        v[i] = vo;
        m[i] = mo;
        if (lookahead) slow[i] = so;
        p[i] = po;
    }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Parameter / layout packing, dtype casts and the fused Ranger optimizer step for gfx950.
// Packing turns the reference's parameter tensors (state_dict schema: OIHW conv weights,
// (Cin,Cout,kH,kW) ConvTranspose2d weight, (out,in) Linear weights -- SURVEY.md section 8(b)) into the
// [rows][tap][Cin] operand layout of conv_gemm.hip / conv_wgrad.hip and back (gradients).
// Ranger: lib/torch_utils/solver/ranger.py:100-200.
#include <algorithm>
#include <cstring>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void pack4_kernel(const float* __restrict__ src, T* __restrict__ dst, int A1, int A2, int Tt,
                                                    int B, int A1v, int A2v, int Bv, long long s1, long long s2, long long st,
                                                    long long sb, int flip) {
    const long long total = (long long)A1 * A2 * Tt * B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i % B);
        long long r = i / B;
        const int t = (int)(r % Tt); r /= Tt;
        const int a2 = (int)(r % A2);
        const int a1 = (int)(r / A2);
        float v = 0.f;
        if (a1 < A1v && a2 < A2v && b < Bv) {
            const int tt = flip ? (Tt - 1 - t) : t;
            v = src[a1 * s1 + a2 * s2 + tt * st + b * sb];
        }
        st1<T>(dst + i, v);
    }
}

__global__ __launch_bounds__(256) void unpack4_kernel(const float* __restrict__ packed, float* __restrict__ dst, int A1, int A2,
                                                      int Tt, int B, int A1v, int A2v, int Bv, long long s1, long long s2,
                                                      long long st, long long sb, int flip) {
    const long long total = (long long)A1v * A2v * Tt * Bv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // iterate the destination-friendly order: b fastest is the packed-contiguous order
        const int b = (int)(i % Bv);
        long long r = i / Bv;
        const int t = (int)(r % Tt); r /= Tt;
        const int a2 = (int)(r % A2v);
        const int a1 = (int)(r / A2v);
        const int tt = flip ? (Tt - 1 - t) : t;
        dst[a1 * s1 + a2 * s2 + tt * st + b * sb] = packed[(((long long)a1 * A2 + a2) * Tt + t) * B + b];
    }
}

// conv1.weight (64,3,7,7) -> [64][7 ky][64 = 16 px * 4 ch]; px >= 7 and ch 3 are zero
template <typename T>
__global__ void pack_stem_w_kernel(const float* __restrict__ w, T* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 7 * 64) return;
    const int k = i & 63, ky = (i >> 6) % 7, o = i / (64 * 7);
    const int kx = k >> 2, c = k & 3;
    float v = 0.f;
    if (kx < 7 && c < 3) v = w[((o * 3 + c) * 7 + ky) * 7 + kx];
    st1<T>(dst + i, v);
}

__global__ void unpack_stem_w_kernel(const float* __restrict__ packed, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 3 * 7 * 7) return;
    const int kx = i % 7, ky = (i / 7) % 7, c = (i / 49) % 3, o = i / 147;
    dw[i] = packed[(o * 7 + ky) * 64 + kx * 4 + c];
}

// NCHW fp32 image -> NHWC4 (dtype) placed at (+3,+3) inside a zero [N][Hp][Wp][4] canvas
template <typename T>
__global__ __launch_bounds__(256) void pack_image_kernel(const float* __restrict__ img, T* __restrict__ dst, int N, int H, int W,
                                                         int Hp, int Wp) {
    const long long total = (long long)N * Hp * Wp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        long long r = i / Wp;
        const int yp = (int)(r % Hp);
        const int n = (int)(r / Hp);
        const int y = yp - 3, x = xp - 3;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && x >= 0 && x < W) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = img[(((size_t)n * 3 + c) * H + y) * W + x];
        }
        T* o = dst + i * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) st1<T>(o + c, v[c]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        st1<T>(dst + i, src[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = ld1<T>(src + i);
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, int cs, int c0, int C, float* __restrict__ dst,
                                                           int N, int HW) {
    const long long total = (long long)N * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        long long r = i / HW;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        dst[i] = ld1<T>(src + ((size_t)n * HW + pix) * cs + c0 + c);
    }
}

// one block per row: gradient centralisation (row mean), RAdam moments, update, optional lookahead
__global__ __launch_bounds__(256) void ranger_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, float* __restrict__ slow, int cols, int gc, float lr,
                                                     float beta1, float beta2, float eps, float wd, float step_size, int adaptive,
                                                     int lookahead, float alpha) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * cols;
    ranger_row(p + base, g + base, m + base, v + base, slow + base, cols, gc, lr, beta1, beta2, eps, wd, step_size, adaptive, lookahead, alpha, 1.f, red);
}

inline int ew_grid(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, 256LL * 16)); }

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int gdrn_version(void) { return GDRN_ABI_VERSION; }

thread_local int gdrn_tls_hip_error = 0;
thread_local int gdrn_tls_stale_hip_error = 0;
// returns AND clears the calling thread's launch-error slot (a code is reported once, with the failure it belongs to).  Without a launch error:
// 0, and `name` says "hipSuccess" -- or "stale:<name>" when a launch of this thread found (and dropped) an error some earlier HIP call had left
// behind (GDRN_LAUNCH): not this library's failure, but worth a line in a debug log; that slot is cleared too.
extern "C" int gdrn_last_hip_error(char* name, int cap) {
    const int e = gdrn_tls_hip_error, stale = gdrn_tls_stale_hip_error;
    gdrn_tls_hip_error = 0;
    gdrn_tls_stale_hip_error = 0;
    if (name && cap > 0) {
        int i = 0;
        if (e == 0 && stale != 0) {
            const char* pre = "stale:";
            for (; pre[i] && i < cap - 1; ++i) name[i] = pre[i];
        }
        const char* s = hipGetErrorName((hipError_t)(e ? e : stale));
        for (int k = 0; s && s[k] && i < cap - 1; ++k, ++i) name[i] = s[k];
        name[i] = 0;
    }
    return e;
}
extern "C" int gdrn_half_format(void) { return GDRN_DT_H16; }

extern "C" int gdrn_device_info(int dev, char* name, int* cus, char* arch) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return GDRN_ERR_ARG;
    if (name) { strncpy(name, prop.name, 255); name[255] = 0; }
    if (cus) *cus = prop.multiProcessorCount;
    if (arch) { strncpy(arch, prop.gcnArchName, 255); arch[255] = 0; }
    return GDRN_OK;
}

extern "C" int gdrn_pack4(const float* src, void* dst, int A1, int A2, int T, int B, int A1v, int A2v, int Bv, long long s1,
                          long long s2, long long st, long long sb, int flip, int dtype, void* stream) {
    if (!src || !dst || A1 <= 0 || A2 <= 0 || T <= 0 || B <= 0 || A1v > A1 || A2v > A2 || Bv > B) return GDRN_ERR_ARG;
    const long long n = (long long)A1 * A2 * T * B;
    if (dtype == GDRN_DT_F32)
        GDRN_LAUNCH(pack4_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, ST, src, (float*)dst, A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb, flip);
    else if (dtype == GDRN_DT_H16)
        GDRN_LAUNCH(pack4_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, ST, src, (bf16_t*)dst, A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb, flip);
    else
        return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_unpack4(const float* packed, float* dst, int A1, int A2, int T, int B, int A1v, int A2v, int Bv, long long s1,
                            long long s2, long long st, long long sb, int flip, void* stream) {
    if (!packed || !dst || A1 <= 0 || A2 <= 0 || T <= 0 || B <= 0 || A1v > A1 || A2v > A2 || Bv > B) return GDRN_ERR_ARG;
    const long long n = (long long)A1v * A2v * T * Bv;
    GDRN_LAUNCH(unpack4_kernel, dim3(ew_grid(n)), dim3(256), 0, ST, packed, dst, A1, A2, T, B, A1v, A2v, Bv, s1, s2, st, sb, flip);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_pack_stem_w(const float* w, void* dst, int dtype, void* stream) {
    if (!w || !dst) return GDRN_ERR_ARG;
    const int n = 64 * 7 * 64;
    if (dtype == GDRN_DT_F32) GDRN_LAUNCH(pack_stem_w_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, ST, w, (float*)dst);
    else if (dtype == GDRN_DT_H16) GDRN_LAUNCH(pack_stem_w_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, ST, w, (bf16_t*)dst);
    else return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_unpack_stem_w(const float* packed, float* dw, void* stream) {
    if (!packed || !dw) return GDRN_ERR_ARG;
    GDRN_LAUNCH(unpack_stem_w_kernel, dim3(cdiv(64 * 147, 256)), dim3(256), 0, ST, packed, dw);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_pack_image(const float* img, void* dst, int N, int H, int W, int Hp, int Wp, int dtype, void* stream) {
    if (!img || !dst || N <= 0 || Hp < H + 6 || Wp < W + 6) return GDRN_ERR_ARG;
    const long long n = (long long)N * Hp * Wp;
    if (dtype == GDRN_DT_F32) GDRN_LAUNCH(pack_image_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, ST, img, (float*)dst, N, H, W, Hp, Wp);
    else if (dtype == GDRN_DT_H16) GDRN_LAUNCH(pack_image_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, ST, img, (bf16_t*)dst, N, H, W, Hp, Wp);
    else return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_cast_from_f32(const float* src, void* dst, long long n, int dtype, void* stream) {
    if (!src || !dst || n <= 0) return GDRN_ERR_ARG;
    if (dtype == GDRN_DT_F32) GDRN_LAUNCH(cast_from_f32_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, ST, src, (float*)dst, n);
    else if (dtype == GDRN_DT_H16) GDRN_LAUNCH(cast_from_f32_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, ST, src, (bf16_t*)dst, n);
    else return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_cast_to_f32(const void* src, float* dst, long long n, int dtype, void* stream) {
    if (!src || !dst || n <= 0) return GDRN_ERR_ARG;
    if (dtype == GDRN_DT_F32) GDRN_LAUNCH(cast_to_f32_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, ST, (const float*)src, dst, n);
    else if (dtype == GDRN_DT_H16) GDRN_LAUNCH(cast_to_f32_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, ST, (const bf16_t*)src, dst, n);
    else return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_nhwc_to_nchw_f32(const void* src, int cs, int c0, int C, float* dst, int N, int HW, int dtype, void* stream) {
    if (!src || !dst || N <= 0 || HW <= 0 || C <= 0 || c0 < 0 || c0 + C > cs) return GDRN_ERR_ARG;
    const long long n = (long long)N * C * HW;
    if (dtype == GDRN_DT_F32) GDRN_LAUNCH(nhwc_to_nchw_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, ST, (const float*)src, cs, c0, C, dst, N, HW);
    else if (dtype == GDRN_DT_H16) GDRN_LAUNCH(nhwc_to_nchw_kernel<bf16_t>, dim3(ew_grid(n)), dim3(256), 0, ST, (const bf16_t*)src, cs, c0, C, dst, N, HW);
    else return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_ranger_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* slow, int rows, int cols,
                                int gc, float lr, float beta1, float beta2, float eps, float weight_decay, float step_size,
                                int adaptive, int lookahead, float alpha, void* stream) {
    if (!p || !g || !exp_avg || !exp_avg_sq || !slow || rows <= 0 || cols <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(ranger_kernel, dim3(rows), dim3(256), 0, ST, p, g, exp_avg, exp_avg_sq, slow, cols, gc, lr, beta1, beta2,
                       eps, weight_decay, step_size, adaptive, lookahead, alpha);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

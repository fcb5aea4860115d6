// HBM-bound normalisation / pooling / resampling kernels of the GDR-Net RoI path for gfx950:
// BatchNorm2d (train + eval) with fused ReLU / residual / max-pool, GroupNorm+ReLU, bilinear x2
// upsampling (align_corners), LeakyReLU backward, bias gradients.  NHWC, 16-byte vector accesses,
// fp32 arithmetic, storage type T = float | bf16.  Reference call sites: see include/gdrn_hip.h.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

// ------------------------------------------------------------------------------------------ BN
// Per-tile partial rows [rows][2][C] -> per-channel totals in ONE launch: a workgroup owns 4 consecutive channels (one float4 per
// statistic and row), thread t walks rows t, t+256, ... with all its loads independent (a few L2 round trips even for the
// 2048-8192 rows of the 64x64 maps), fp64 accumulation (the parity mode sits at the fp32 noise floor), shuffle + LDS tree.
// Replaces the bn_reduce_rows + bn_finalize pair (two dependent ~5 us launches per BatchNorm, 43 BatchNorms per step); the
// single-launch variants tried before serialised up to 4096 rows per thread (17 us) or needed an agent-scope fence.
__device__ __forceinline__ void rows_total4(const float* __restrict__ rows, int nrows, int C, int c4, double (&tot)[8], double (*red)[8]) {
    double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int r = threadIdx.x; r < nrows; r += 256) {
        const float4 a = *reinterpret_cast<const float4*>(rows + ((size_t)r * 2 + 0) * C + c4);
        const float4 b = *reinterpret_cast<const float4*>(rows + ((size_t)r * 2 + 1) * C + c4);
        s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
        s[4] += (double)b.x; s[5] += (double)b.y; s[6] += (double)b.z; s[7] += (double)b.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[j] += __shfl_xor(s[j], o, 64);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x >> 6][j] = s[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) tot[j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
}

__global__ __launch_bounds__(256) void bn_finalize_rows_kernel(const float* __restrict__ partial, int rows, int C, double count,
                                                               const float* gamma, const float* beta, float* running_mean,
                                                               float* running_var, long long* nbt, float momentum, float eps,
                                                               float* mean, float* invstd, float* scale, float* shift) {
    __shared__ double red[4][8];
    double tot[8];
    const int c4 = blockIdx.x * 4;
    // the per-channel parameters are fetched before the reduction, not after it (a second dependent memory round trip in a
    // kernel that is nothing but latency)
    float k_g = 0.f, k_b = 0.f, k_rm = 0.f, k_rv = 0.f;
    if (threadIdx.x < 4) {
        k_g = gamma[c4 + threadIdx.x];
        k_b = beta[c4 + threadIdx.x];
        if (running_mean != nullptr) { k_rm = running_mean[c4 + threadIdx.x]; k_rv = running_var[c4 + threadIdx.x]; }
    }
    rows_total4(partial, rows, C, c4, tot, red);
    if (threadIdx.x < 4) {
        const int ch = c4 + threadIdx.x;
        const double s1 = tot[threadIdx.x], s2 = tot[4 + threadIdx.x];
        const double m = s1 / count;
        double var = s2 / count - m * m;
        if (var < 0.0) var = 0.0;
        const double is = 1.0 / sqrt(var + (double)eps);
        mean[ch] = (float)m;
        invstd[ch] = (float)is;
        const float sc = k_g * (float)is;
        scale[ch] = sc;
        shift[ch] = k_b - (float)m * sc;
        if (running_mean != nullptr) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[ch] = (1.f - momentum) * k_rm + momentum * (float)m;
            running_var[ch] = (1.f - momentum) * k_rv + momentum * (float)unb;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
}

// BatchNorm-backward sums (per-tile rows of the fused data-gradient epilogue, or the per-workgroup rows of
// gdrn_bn_bwd_reduce) -> the per-channel coefficients of dx = a*g + (b*x + c) (bn_bwd_apply_kernel's prologue arithmetic),
// dgamma / dbeta: what a consumer that applies the BatchNorm backward while staging its operand needs.
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(const float* __restrict__ rows, int nrows, int C, float inv_n,
                                                          const float* gamma, const float* mean, const float* invstd,
                                                          float* ka, float* kb, float* kc, float* dgamma, float* dbeta) {
    __shared__ double red[4][8];
    double tot[8];
    const int c4 = blockIdx.x * 4;
    float k_g = 0.f, k_mu = 0.f, k_is = 0.f;  // fetched before the reduction (see bn_finalize_rows_kernel)
    if (threadIdx.x < 4) { k_g = gamma[c4 + threadIdx.x]; k_mu = mean[c4 + threadIdx.x]; k_is = invstd[c4 + threadIdx.x]; }
    rows_total4(rows, nrows, C, c4, tot, red);
    if (threadIdx.x < 4) {
        const int ch = c4 + threadIdx.x;
        float m1 = (float)tot[threadIdx.x], m2 = (float)tot[4 + threadIdx.x];
        if (dgamma != nullptr) { dbeta[ch] = m1; dgamma[ch] = m2; }
        m1 *= inv_n;
        m2 *= inv_n;
        const float a = k_g * k_is, b = -a * k_is * m2;
        ka[ch] = a;
        kb[ch] = b;
        kc[ch] = -a * m1 - b * k_mu;
    }
}

__global__ void bn_eval_params_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] / sqrtf(rv[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

// Channel-stationary mapping for the HBM-bound BN passes: thread = (channel vector cv, row lane rl); the per-channel
// constants live in registers and a thread walks rows with a fixed stride -- no per-element index division, no
// per-element scale/shift loads (the first version spent its time in 64-bit modulo and reached only 1.6 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ res,
                                                       T* __restrict__ y, long long npix, int C, int relu, int rows_per_block) {
    constexpr int V = Vec16<T>::VEC;
    const int tpr = C / V, rpp = 256 / tpr;
    const int cv = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    __shared__ float kst[2][512];
    for (int c = threadIdx.x; c < C; c += 256) { kst[0][c] = scale[c]; kst[1][c] = shift[c]; }
    __syncthreads();
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = kst[0][cv * V + j]; sh[j] = kst[1][cv * V + j]; }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(npix, r0 + rows_per_block);
    if (rl >= rpp) return;
    for (long long r = r0 + rl; r < r1; r += rpp) {
        const size_t o = (size_t)r * C + cv * V;
        float v[V], q[V];
        Vec16<T>::load(x + o, v);
        if (res != nullptr) Vec16<T>::load(res + o, q);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float t = __builtin_fmaf(v[j], sc[j], sh[j]);
            if (res != nullptr) t += q[j];
            if (relu) t = fmaxf(t, 0.f);
            v[j] = t;
        }
        Vec16<T>::store(y + o, v);
    }
}

// per-channel sums of g and g*xhat; rows strided over threads, LDS reduction, ONE partial row [2][C] per workgroup (plain
// stores -- gdrn_bn_bwd_coef adds the rows up in fp64: deterministic, no pre-zeroed accumulator, no same-address atomics)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ ym,
                                                            const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ msc,
                                                            const float* __restrict__ msh, long long npix, int C,
                                                            float* __restrict__ rows, int rows_per_block) {
    constexpr int V = Vec16<T>::VEC;
    const int tpr = C / V, rpp = 256 / tpr;
    const int cv = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    // per-channel constants through LDS: one coalesced load round per workgroup instead of 4*V dependent scalar loads
    // per thread (those latency chains cost ~10 us per launch)
    __shared__ float kst[4][512];
    __shared__ float part[4][2 * 512];  // per-wave partial sums, added up in wave order: no float atomics, bit-reproducible
    for (int c = threadIdx.x; c < C; c += 256) {
        kst[0][c] = mean[c];
        kst[1][c] = invstd[c];
        kst[2][c] = msc ? msc[c] : 0.f;
        kst[3][c] = msc ? msh[c] : 1.f;
    }
    for (int i = threadIdx.x; i < 4 * 2 * 512; i += 256) (&part[0][0])[i] = 0.f;
    __syncthreads();
    float s1[V], s2[V], mu[V], is[V], ksc[V], ksh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        s1[j] = 0.f; s2[j] = 0.f;
        mu[j] = kst[0][cv * V + j]; is[j] = kst[1][cv * V + j]; ksc[j] = kst[2][cv * V + j]; ksh[j] = kst[3][cv * V + j];
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(npix, r0 + rows_per_block);
    if (rl < rpp) {
        for (long long r = r0 + rl; r < r1; r += rpp) {
            float g[V], xv[V], yv[V];
            Vec16<T>::load(dy + r * C + cv * V, g);
            Vec16<T>::load(x + r * C + cv * V, xv);
            if (ym != nullptr) Vec16<T>::load(ym + r * C + cv * V, yv);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float gg = g[j];
                if (ym != nullptr && !(yv[j] > 0.f)) gg = 0.f;
                if (!(__builtin_fmaf(xv[j], ksc[j], ksh[j]) > 0.f)) gg = 0.f;  // ReLU mask recomputed from x (no-op without msc)
                s1[j] += gg;
                s2[j] += gg * (xv[j] - mu[j]) * is[j];
            }
        }
    }
    // lanes of a wave that share a channel vector (lane % tpr) are combined with shuffles (fixed tree); every (wave, channel)
    // slot of `part` then has exactly one writer
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = tpr; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < V; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    }
    if (rl < rpp && (tpr >= 64 || lane < tpr)) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            part[wave][cv * V + j] = s1[j];
            part[wave][C + cv * V + j] = s2[j];
        }
    }
    __syncthreads();
    float* dst = rows + (size_t)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ ym,
                                                           const T* __restrict__ x, const float* __restrict__ ca,
                                                           const float* __restrict__ cb, const float* __restrict__ cc,
                                                           const float* __restrict__ msc, const float* __restrict__ msh,
                                                           long long npix, int C, T* __restrict__ dx, T* __restrict__ gout,
                                                           int rows_per_block) {
    constexpr int V = Vec16<T>::VEC;
    // dx = a*g + (b*x + c) with the per-channel (a, b, c) of gdrn_bn_bwd_coef, handed out through LDS
    __shared__ float kst[5][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        kst[0][c] = ca[c];
        kst[1][c] = cb[c];
        kst[2][c] = cc[c];
        kst[3][c] = msc ? msc[c] : 0.f;
        kst[4][c] = msc ? msh[c] : 1.f;
    }
    __syncthreads();
    const int tpr = C / V, rpp = 256 / tpr;
    const int cv = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    float ka[V], kb[V], kc[V], ksc[V], ksh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = cv * V + j;
        ka[j] = kst[0][c]; kb[j] = kst[1][c]; kc[j] = kst[2][c]; ksc[j] = kst[3][c]; ksh[j] = kst[4][c];
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(npix, r0 + rows_per_block);
    if (rl >= rpp) return;
    for (long long r = r0 + rl; r < r1; r += rpp) {
        const size_t off = (size_t)r * C + cv * V;
        float g[V], xv[V], yv[V], o[V];
        Vec16<T>::load(dy + off, g);
        Vec16<T>::load(x + off, xv);
        if (ym != nullptr) Vec16<T>::load(ym + off, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float gg = g[j];
            if (ym != nullptr && !(yv[j] > 0.f)) gg = 0.f;
            if (!(__builtin_fmaf(xv[j], ksc[j], ksh[j]) > 0.f)) gg = 0.f;
            g[j] = gg;
            o[j] = __builtin_fmaf(ka[j], gg, __builtin_fmaf(kb[j], xv[j], kc[j]));
        }
        Vec16<T>::store(dx + off, o);
        if (gout != nullptr) Vec16<T>::store(gout + off, g);
    }
}

// ------------------------------------------------------------------------------------------ stem pool
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ y,
                                                                  unsigned char* __restrict__ idx, int N, int H, int W, int C) {
    constexpr int V = Vec16<T>::VEC;
    __shared__ float sc_s[512], sh_s[512];  // per-channel constants once per workgroup (dependent scalar loads per thread cost ~10 us)
    for (int c = threadIdx.x; c < C; c += 256) { sc_s[c] = scale[c]; sh_s[c] = shift[c]; }
    __syncthreads();
    const int Ho = H / 2, Wo = W / 2, cvn = C / V;
    const unsigned total = (unsigned)((long long)N * Ho * Wo * cvn);  // < 2^31, checked by the host wrapper: 32-bit index math
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvn);
        unsigned t = i / cvn;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float best[V]; int bi[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { best[j] = -INFINITY; bi[j] = 0; }
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                float v[V];
                Vec16<T>::load(x + ((size_t)(n * H + iy) * W + ix) * C + cv * V, v);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float a = fmaxf(v[j] * sc_s[cv * V + j] + sh_s[cv * V + j], 0.f);
                    if (a > best[j]) { best[j] = a; bi[j] = ky * 3 + kx; }
                }
            }
        }
        Vec16<T>::store(y + (size_t)i * V, best);
        unsigned char* ip = idx + (size_t)i * V;
        if (V == 8) {
            uint2 pk;
            pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
            pk.y = bi[4 % V] | (bi[5 % V] << 8) | (bi[6 % V] << 16) | (bi[7 % V] << 24);
            *reinterpret_cast<uint2*>(ip) = pk;
        } else {
            *reinterpret_cast<uint32_t*>(ip) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                          const T* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, T* __restrict__ g, int N, int H,
                                                          int W, int C, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, float* __restrict__ rows) {
    constexpr int V = Vec16<T>::VEC;
    __shared__ float sc_s[512], sh_s[512], mu_s[512], is_s[512];
    for (int c = threadIdx.x; c < C; c += 256) {
        sc_s[c] = scale[c]; sh_s[c] = shift[c];
        mu_s[c] = rows ? mean[c] : 0.f; is_s[c] = rows ? invstd[c] : 0.f;
    }
    __syncthreads();
    // rows != NULL: also the two BatchNorm-backward sums of g (sum g, sum g*xhat per channel) as one partial row per workgroup --
    // the host guarantees a thread's channel vector is the same in every grid-stride iteration (C/V divides 64 and the stride)
    float s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int Ho = H / 2, Wo = W / 2, cvn = C / V;
    const unsigned total = (unsigned)((long long)N * H * W * cvn);  // < 2^31, checked by the host wrapper: 32-bit index math
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvn);
        unsigned t = i / cvn;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        float acc[V], xv[V];
        // windows containing iy: oy with 2*oy-1 <= iy <= 2*oy+1 -- one (even iy) or two (odd iy), likewise in x.  All four candidate windows
        // are loaded up front from clamped addresses (a missing second window re-reads the first: an L1 hit) and masked afterwards: as a
        // nested loop with `continue`s the 1..4 (dy, index) load pairs of a pixel were issued one memory round trip after the other.
        // (the V tap indices of a vector are ONE 4- / 8-byte word, written that way by the forward kernel)
        const int oyA = iy >> 1, oyBr = (iy + 1) >> 1, oxA = ix >> 1, oxBr = (ix + 1) >> 1;
        const bool vBy = oyBr != oyA && oyBr < Ho, vBx = oxBr != oxA && oxBr < Wo;
        const int oyB = min(oyBr, Ho - 1), oxB = min(oxBr, Wo - 1);
        const int kyA = iy - 2 * oyA + 1, kyB = iy - 2 * oyBr + 1, kxA = ix - 2 * oxA + 1, kxB = ix - 2 * oxBr + 1;
        const size_t rA = (size_t)(n * Ho + oyA) * Wo, rB = (size_t)(n * Ho + oyB) * Wo;
        const size_t o4[4] = {(rA + oxA) * C + cv * V, (rA + oxB) * C + cv * V, (rB + oxA) * C + cv * V, (rB + oxB) * C + cv * V};
        const int tap4[4] = {kyA * 3 + kxA, kyA * 3 + kxB, kyB * 3 + kxA, kyB * 3 + kxB};
        const bool ok4[4] = {true, vBx, vBy, vBy && vBx};
        float d4[4][V];
        unsigned long long pk4[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            Vec16<T>::load(dy + o4[w], d4[w]);
            if constexpr (V == 8) pk4[w] = *reinterpret_cast<const unsigned long long*>(idx + o4[w]);
            else pk4[w] = *reinterpret_cast<const unsigned int*>(idx + o4[w]);
        }
        Vec16<T>::load(x + (size_t)i * V, xv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            acc[j] = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)   // (same order of additions as the windows were walked before: oy outer, ox inner)
                acc[j] += (ok4[w] && (int)((pk4[w] >> (8 * j)) & 0xffull) == tap4[w]) ? d4[w][j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; ++j)
            if (!(xv[j] * sc_s[cv * V + j] + sh_s[cv * V + j] > 0.f)) acc[j] = 0.f;
        Vec16<T>::store(g + (size_t)i * V, acc);
        if (rows != nullptr) {
            // the sums see g as the consumer will read it (rounded to the storage type), like a separate gdrn_bn_bwd_reduce pass
            float gr[V];
            Vec16<T>::unpack(Vec16<T>::pack(acc), gr);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                s1[j] += gr[j];
                s2[j] += gr[j] * (xv[j] - mu_s[cv * V + j]) * is_s[cv * V + j];
            }
        }
    }
    if (rows != nullptr) {
        __shared__ float part[4][2 * 512];
        const int cvn2 = C / V, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cvt = threadIdx.x % cvn2;
        for (int o = cvn2; o < 64; o <<= 1) {
#pragma unroll
            for (int j = 0; j < V; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
        }
        if (lane < cvn2) {
#pragma unroll
            for (int j = 0; j < V; ++j) { part[wave][cvt * V + j] = s1[j]; part[wave][C + cvt * V + j] = s2[j]; }
        }
        __syncthreads();
        float* dst = rows + (size_t)blockIdx.x * 2 * C;
        for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
    }
}

// ------------------------------------------------------------------------------------------ upsample
// PyTorch area_pixel_compute_source_index(align_corners=True): src = dst * (in-1)/(out-1) in fp32.
// the bilinear combination with its roundings spelled out (two instantiations of `ly0 * (lx0 * a + lx1 * b) + ly1 * (...)` were contracted into
// different fma patterns by hipcc: fused and unfused launches must give the same bits)
__device__ __forceinline__ float up_lerp(float ly0, float ly1, float lx0, float lx1, float a00, float a01, float a10, float a11) {
#pragma clang fp contract(off)
    // (explicit fma calls are not enough under -ffp-contract=fast: the `contract` flag they carry lets the back end re-pair the products --
    //  measured: rows<false> and rows<true> disagreed on 0.06 % of the elements, every one a bf16 rounding tie.  The products are made opaque.)
    float p0 = lx0 * a00, p1 = lx0 * a10;
    asm volatile("" : "+v"(p0), "+v"(p1));
    float r0 = __builtin_fmaf(lx1, a01, p0), r1 = __builtin_fmaf(lx1, a11, p1);
    asm volatile("" : "+v"(r0), "+v"(r1));
    float q0 = ly0 * r0;
    asm volatile("" : "+v"(q0));
    return __builtin_fmaf(ly1, r1, q0);
}

// BNR (r6): x is the RAW input of a BatchNorm + ReLU and the upsampled tensor is that of relu(scale * x + shift), each source value rounded to
// the storage format first -- exactly what gdrn_bn_apply would have stored and this kernel then read (cdpn_rot_head_region.py:103-123:
// BN -> ReLU -> UpsamplingBilinear2d): one launch and one tensor round trip less per upsampling
template <typename T, bool BNR = false>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                                             const float* __restrict__ scale = nullptr, const float* __restrict__ shift = nullptr) {
    constexpr int V = Vec16<T>::VEC;
    const int Ho = 2 * H, Wo = 2 * W, cvn = C / V;
    const float sh = (float)(H - 1) / (float)(Ho - 1), sw = (float)(W - 1) / (float)(Wo - 1);
    const unsigned total = (unsigned)((long long)N * Ho * Wo * cvn);  // < 2^31, checked by the host wrapper: 32-bit index math
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvn);
        unsigned t = i / cvn;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float fy = sh * oy, fx = sw * ox;
        asm volatile("" : "+v"(fy), "+v"(fx));   // (rounded products: see upsample2x_rows_kernel)
        const int y0 = (int)fy, x0 = (int)fx;
        const int yp = (y0 < H - 1) ? 1 : 0, xp = (x0 < W - 1) ? 1 : 0;
        const float ly1 = fy - y0, ly0 = 1.f - ly1, lx1 = fx - x0, lx0 = 1.f - lx1;
        const T* b = x + ((size_t)(n * H + y0) * W + x0) * C + cv * V;
        float a00[V], a01[V], a10[V], a11[V], o[V];
        Vec16<T>::load(b, a00);
        Vec16<T>::load(b + (size_t)xp * C, a01);
        Vec16<T>::load(b + (size_t)yp * W * C, a10);
        Vec16<T>::load(b + ((size_t)yp * W + xp) * C, a11);
        if constexpr (BNR) {
            float sc[V], sh_[V], t[V];
#pragma unroll
            for (int j = 0; j < V; ++j) { sc[j] = scale[cv * V + j]; sh_[j] = shift[cv * V + j]; }
            float* src[4] = {a00, a01, a10, a11};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int j = 0; j < V; ++j) t[j] = fmaxf(__builtin_fmaf(src[k][j], sc[j], sh_[j]), 0.f);   // bn_apply_kernel's arithmetic
                alignas(16) T q[V];
                Vec16<T>::store(q, t);     // round to the storage format ...
                Vec16<T>::load(q, src[k]); // ... as the stored activation would have been
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = up_lerp(ly0, ly1, lx0, lx1, a00[j], a01[j], a10[j], a11[j]);
        Vec16<T>::store(y + (size_t)i * V, o);
    }
}

// Row form (r6): a workgroup = one output row (n, oy).  The element-wise kernel above spends ~60 of its ~150 VALU instructions per 16-byte output
// on three integer divisions, and its BNR variant transforms every source value four times (once per output pixel that reads it) -- both kernels
// are VALU-bound, not HBM-bound (upsample2x_fwd 27 / 45 us per launch against 17 / 22 us of traffic).  Here the row's source geometry is
// uniform, a thread keeps its channel vector (C / V a power of two that divides 256: shift / mask), and the BNR variant transforms the two source rows
// ONCE into LDS ([2][W][C] in the storage format: exactly the activation gdrn_bn_apply would have stored) before the interpolation reads them.
// Same arithmetic per element as the element-wise kernels: results are bit-identical to theirs.
template <typename T, bool BNR>
__global__ __launch_bounds__(256) void upsample2x_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int lg_cvn,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, float sh, float sw) {
    constexpr int V = Vec16<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char up_smem[];
    const int Ho = 2 * H, Wo = 2 * W, cvn = 1 << lg_cvn;
    const int oy = (int)(blockIdx.x % (unsigned)Ho), n = (int)(blockIdx.x / (unsigned)Ho);
    // (sh, sw = (in - 1) / (out - 1) come from the host: two instantiations of this kernel evaluated the division differently -- one correctly
    //  rounded, one through v_rcp_f32 -- and their outputs disagreed on 0.06 % of the elements, every one a rounding tie of the storage format)
    // (the source coordinate is ROUNDED before its integer part is taken off: hipcc contracted `sh * oy - y0` into one fma in one instantiation of
    //  this kernel and not in the other -- weights one ulp apart, outputs apart on every rounding tie of the storage format.  Opaque products.)
    float fy = sh * oy;
    asm volatile("" : "+v"(fy));
    const int y0 = (int)fy, yp = (y0 < H - 1) ? 1 : 0;
    const float ly1 = fy - y0, ly0 = 1.f - ly1;
    const T* g0 = x + (size_t)(n * H + y0) * W * C;
    const T* g1 = g0 + (size_t)yp * W * C;
    const int cv = threadIdx.x & (cvn - 1);   // (256 % cvn == 0: fixed per thread)
    T* s0 = reinterpret_cast<T*>(up_smem);
    T* s1 = s0 + (size_t)W * C;
    if constexpr (BNR) {
        float sc[V], sf[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { sc[j] = scale[cv * V + j]; sf[j] = shift[cv * V + j]; }
        const int per_row = W << lg_cvn;
        for (int i = threadIdx.x; i < 2 * per_row; i += 256) {
            const int row = i >= per_row ? 1 : 0, j = i - row * per_row;
            float v[V];
            Vec16<T>::load((row ? g1 : g0) + (size_t)j * V, v);
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = fmaxf(__builtin_fmaf(v[k], sc[k], sf[k]), 0.f);   // bn_apply_kernel's arithmetic, one rounding
            Vec16<T>::store((row ? s1 : s0) + (size_t)j * V, v);
        }
        __syncthreads();
    }
    T* yr = y + (size_t)(n * Ho + oy) * Wo * C;
    for (int i = threadIdx.x; i < (Wo << lg_cvn); i += 256) {
        const int ox = i >> lg_cvn;
        float fx = sw * ox;
        asm volatile("" : "+v"(fx));
        const int x0 = (int)fx, xp = (x0 < W - 1) ? 1 : 0;
        const float lx1 = fx - x0, lx0 = 1.f - lx1;
        const int e0 = ((x0 << lg_cvn) + cv) * V, e1 = (((x0 + xp) << lg_cvn) + cv) * V;
        float a00[V], a01[V], a10[V], a11[V], o[V];
        if constexpr (BNR) {
            Vec16<T>::load(s0 + e0, a00); Vec16<T>::load(s0 + e1, a01); Vec16<T>::load(s1 + e0, a10); Vec16<T>::load(s1 + e1, a11);
        } else {
            Vec16<T>::load(g0 + e0, a00); Vec16<T>::load(g0 + e1, a01); Vec16<T>::load(g1 + e0, a10); Vec16<T>::load(g1 + e1, a11);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] = up_lerp(ly0, ly1, lx0, lx1, a00[j], a01[j], a10[j], a11[j]);
        Vec16<T>::store(yr + (size_t)i * V, o);
    }
}

// the row form covers the shape: C / V a power of two that divides 256, the two source rows fit 64 KiB of LDS, one workgroup per output row
static bool upsample_rows_ok(int N, int H, int W, int C, int dtype, int* lg) {
    const int V = dtype == GDRN_DT_H16 ? 8 : 4, cvn = C / V;
    if (C % V || cvn <= 0 || (cvn & (cvn - 1)) || cvn > 256) return false;
    if ((size_t)2 * W * C * (dtype == GDRN_DT_H16 ? 2 : 4) > 64 * 1024) return false;
    if ((long long)N * 2 * H > 0x7fffffffll) return false;
    int l = 0;
    while ((1 << l) < cvn) ++l;
    *lg = l;
    return true;
}

__device__ __forceinline__ float up_weight(int o, int i, int In, float s) {
    float f = s * o;
    asm volatile("" : "+v"(f));   // (rounded product, as the forward kernels take it)
    const int i0 = (int)f;
    const int ip = (i0 < In - 1) ? 1 : 0;
    const float l1 = f - i0;
    float w = 0.f;
    if (i0 == i) w += 1.f - l1;
    if (i0 + ip == i) w += l1;
    return w;
}

// gradient of the 2x bilinear (align_corners) upsampling w.r.t. input pixel (n, iy, ix), channel vector cv: acc[V]
template <typename T>
__device__ __forceinline__ void up_bwd_pixel(const T* __restrict__ dy, int n, int iy, int ix, int cv, int H, int W, int C, float sh, float sw,
                                             float (&acc)[Vec16<T>::VEC]) {
    constexpr int V = Vec16<T>::VEC;
    const int Ho = 2 * H, Wo = 2 * W;
    // outputs whose source coordinate lies in (iy-1, iy+1): a candidate range of at most six per dimension, of which at most FOUR
    // consecutive ones carry a weight (2x, align_corners).  The 4 x 4 block starting at the first weighted row / column is loaded from
    // clamped addresses with zero weights where there is nothing -- all loads independent.  (Walking the candidate range with
    // `continue` on zero weights issued the up to 16 loads of a pixel one memory round trip after the other.)
    const int oy_lo = max(0, (int)floorf((iy - 1) / sh)), ox_lo = max(0, (int)floorf((ix - 1) / sw));
    float wyc[6], wxc[6];
    int fy = 5, fx = 5;
#pragma unroll
    for (int k = 5; k >= 0; --k) {
        wyc[k] = (oy_lo + k < Ho) ? up_weight(oy_lo + k, iy, H, sh) : 0.f;
        wxc[k] = (ox_lo + k < Wo) ? up_weight(ox_lo + k, ix, W, sw) : 0.f;
        if (wyc[k] != 0.f) fy = k;
        if (wxc[k] != 0.f) fx = k;
    }
    float wy4[4], wx4[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        wy4[a] = 0.f;
        wx4[a] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (fy + a == k) wy4[a] = wyc[k];
            if (fx + a == k) wx4[a] = wxc[k];
        }
    }
    const int oy0 = oy_lo + fy, ox0 = ox_lo + fx;
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const size_t row = (size_t)(n * Ho + min(oy0 + a, Ho - 1)) * Wo;
        float d[4][V];
#pragma unroll
        for (int b = 0; b < 4; ++b) Vec16<T>::load(dy + (row + min(ox0 + b, Wo - 1)) * C + cv * V, d[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float w = wy4[a] * wx4[b];
            if (w != 0.f) {   // (same order of additions as before: oy outer, ox inner, zero weights skipped)
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += w * d[b][j];
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C) {
    constexpr int V = Vec16<T>::VEC;
    const int Ho = 2 * H, Wo = 2 * W, cvn = C / V;
    const float sh = (float)(H - 1) / (float)(Ho - 1), sw = (float)(W - 1) / (float)(Wo - 1);
    const unsigned total = (unsigned)((long long)N * H * W * cvn);  // < 2^31, checked by the host wrapper: 32-bit index math
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cv = (int)(i % cvn);
        unsigned t = i / cvn;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        float acc[V];
        up_bwd_pixel<T>(dy, n, iy, ix, cv, H, W, C, sh, sw, acc);
        Vec16<T>::store(dx + (size_t)i * V, acc);
    }
}

// ... and the same with the reduction pass of the BatchNorm(+ReLU) backward that reads the result (bn_bwd_reduce_kernel, affine ReLU mask):
// channel-stationary threads (cv, rl) walk the low-resolution pixels of the workgroup's row range, write dx UNMASKED (what the separate
// kernel stores; the consumer's operand transform masks it) and accumulate sum g / sum g * xhat of the masked, storage-rounded value
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_bnsums_kernel(const T* __restrict__ dy, T* __restrict__ dx, const T* __restrict__ x,
                                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                    const float* __restrict__ msc, const float* __restrict__ msh, int N, int H,
                                                                    int W, int C, float* __restrict__ rows, int rows_per_block) {
    constexpr int V = Vec16<T>::VEC;
    const int tpr = C / V, rpp = 256 / tpr;
    const int cv = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    const float sh = (float)(H - 1) / (float)(2 * H - 1), sw = (float)(W - 1) / (float)(2 * W - 1);
    __shared__ float kst[4][512];
    __shared__ float part[4][2 * 512];
    for (int c = threadIdx.x; c < C; c += 256) { kst[0][c] = mean[c]; kst[1][c] = invstd[c]; kst[2][c] = msc[c]; kst[3][c] = msh[c]; }
    for (int i = threadIdx.x; i < 4 * 2 * 512; i += 256) (&part[0][0])[i] = 0.f;
    __syncthreads();
    float s1[V], s2[V], mu[V], is[V], ksc[V], ksh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        s1[j] = 0.f; s2[j] = 0.f;
        mu[j] = kst[0][cv * V + j]; is[j] = kst[1][cv * V + j]; ksc[j] = kst[2][cv * V + j]; ksh[j] = kst[3][cv * V + j];
    }
    const unsigned npix = (unsigned)N * (unsigned)H * (unsigned)W;   // < 2^31 (host wrapper): 32-bit index arithmetic
    const unsigned r0 = blockIdx.x * (unsigned)rows_per_block;
    const unsigned r1 = min(npix, r0 + (unsigned)rows_per_block);
    if (rl < rpp) {
        for (unsigned r = r0 + rl; r < r1; r += rpp) {
            const unsigned q_ = r / (unsigned)W;
            const int ix = (int)(r - q_ * (unsigned)W), n = (int)(q_ / (unsigned)H), iy = (int)(q_ - (unsigned)n * (unsigned)H);
            float acc[V], g[V], xv[V];
            up_bwd_pixel<T>(dy, n, iy, ix, cv, H, W, C, sh, sw, acc);
            alignas(16) T q[V];
            Vec16<T>::store(q, acc);
            *reinterpret_cast<uint4*>(dx + (size_t)r * C + cv * V) = *reinterpret_cast<const uint4*>(q);
            Vec16<T>::load(q, g);      // the stored (rounded) value: what bn_bwd_reduce_kernel would read back
            Vec16<T>::load(x + (size_t)r * C + cv * V, xv);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float gg = g[j];
                if (!(__builtin_fmaf(xv[j], ksc[j], ksh[j]) > 0.f)) gg = 0.f;
                s1[j] += gg;
                s2[j] += gg * (xv[j] - mu[j]) * is[j];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = tpr; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < V; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    }
    if (rl < rpp && (tpr >= 64 || lane < tpr)) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            part[wave][cv * V + j] = s1[j];
            part[wave][C + cv * V + j] = s2[j];
        }
    }
    __syncthreads();
    float* dst = rows + (size_t)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
}

// ------------------------------------------------------------------------------------------ GroupNorm
// one block per sample; CPG = C/G channels per group (4 at the benchmark configs).
// grid (N, C / CS): a workgroup normalises the CS-channel slab blockIdx.y of sample blockIdx.x (groups never straddle a slab:
// CS is a multiple of C/G).  One workgroup per sample -- the first version -- ran 64 workgroups on 256 CUs.
template <typename T>
__global__ __launch_bounds__(256) void gn_relu_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          float* __restrict__ mean_rstd, int HW, int C, int G, float eps, int CS) {
    constexpr int V = Vec16<T>::VEC;
    __shared__ double sg[2 * 128];
    __shared__ float kgb[2][512];
    __shared__ float pp[2][256 * V];  // per-thread partial sums [row lane][channel]: added up in a fixed order (no atomics -> reproducible)
    const int n = blockIdx.x, cpg = C / G, c0 = blockIdx.y * CS, g0 = c0 / cpg, GS = CS / cpg;
    const int tpr = CS / V, rpp = 256 / tpr;
    const int cv = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    for (int i = threadIdx.x; i < CS; i += 256) { kgb[0][i] = gamma[c0 + i]; kgb[1][i] = beta[c0 + i]; }
    __syncthreads();
    const T* xb = x + (size_t)n * HW * C + c0;
    float s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    // (r6b) a thread's rows -- up to GN_NC of them: the 32 x 32 and smaller maps of the path at 32-channel slabs -- stay in registers between the
    // statistics pass and the normalisation: the slab is read once, and the second pass starts without a memory round trip
    constexpr int GN_NC = 16;
    const bool cached = HW <= GN_NC * rpp;   // (uniform)
    uint4 xc[GN_NC];
    if (cached) {
#pragma unroll
        for (int u = 0; u < GN_NC; ++u) {
            const int r = rl + u * rpp;
            if (r < HW) xc[u] = *reinterpret_cast<const uint4*>(xb + (size_t)r * C + cv * V);
        }
#pragma unroll
        for (int u = 0; u < GN_NC; ++u) {
            const int r = rl + u * rpp;
            if (r < HW) {
                float v[V];
                Vec16<T>::unpack(xc[u], v);
#pragma unroll
                for (int j = 0; j < V; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
            }
        }
    } else {
        for (int r = rl; r < HW; r += rpp) {
            float v[V];
            Vec16<T>::load(xb + (size_t)r * C + cv * V, v);
#pragma unroll
            for (int j = 0; j < V; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        }
    }
    if (rl < rpp) {
#pragma unroll
        for (int j = 0; j < V; ++j) { pp[0][rl * CS + cv * V + j] = s1[j]; pp[1][rl * CS + cv * V + j] = s2[j]; }
    }
    __syncthreads();
    // fixed-order tree: 8 partial sums per (statistic, channel) over the row lanes, then per channel, then per group (fp64)
    __shared__ double pc[2][64 * 8];
    if (CS <= 64) {  // (the 32-channel slabs of the path; wider slabs take the one-level loop below)
        for (int i = threadIdx.x; i < 2 * CS * 8; i += 256) {
            const int st = i / (CS * 8), r8 = (i / CS) & 7, c = i % CS;
            double t = 0.0;
            for (int r = r8; r < rpp; r += 8) t += (double)pp[st][r * CS + c];
            pc[st][r8 * CS + c] = t;
        }
    }
    __syncthreads();
    for (int gi = threadIdx.x; gi < 2 * GS; gi += 256) {
        const int st = gi / GS, gq = gi - st * GS;
        double t = 0.0;
        for (int k = 0; k < cpg; ++k) {
            if (CS <= 64) {
                for (int r8 = 0; r8 < 8; ++r8) t += pc[st][r8 * CS + gq * cpg + k];
            } else {
                for (int r = 0; r < rpp; ++r) t += (double)pp[st][r * CS + gq * cpg + k];
            }
        }
        sg[gi] = t;
    }
    __syncthreads();
    const double cnt = (double)HW * cpg;
    float mu[V], rs[V], gm[V], bt[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int gi = (cv * V + j) / cpg;
        const double m = sg[gi] / cnt;
        const double var = fmax(sg[GS + gi] / cnt - m * m, 0.0);
        mu[j] = (float)m;
        rs[j] = (float)(1.0 / sqrt(var + (double)eps));
        gm[j] = kgb[0][cv * V + j];
        bt[j] = kgb[1][cv * V + j];
    }
    if (threadIdx.x < GS) {
        const double m = sg[threadIdx.x] / cnt;
        const double var = fmax(sg[GS + threadIdx.x] / cnt - m * m, 0.0);
        mean_rstd[((size_t)n * G + g0 + threadIdx.x) * 2 + 0] = (float)m;
        mean_rstd[((size_t)n * G + g0 + threadIdx.x) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    T* yb = y + (size_t)n * HW * C + c0;
    if (cached) {
#pragma unroll
        for (int u = 0; u < GN_NC; ++u) {
            const int r = rl + u * rpp;
            if (r < HW) {
                float v[V];
                Vec16<T>::unpack(xc[u], v);
#pragma unroll
                for (int j = 0; j < V; ++j) v[j] = fmaxf((v[j] - mu[j]) * rs[j] * gm[j] + bt[j], 0.f);
                Vec16<T>::store(yb + (size_t)r * C + cv * V, v);
            }
        }
        return;
    }
    for (int r = rl; r < HW; r += rpp) {
        float v[V];
        Vec16<T>::load(xb + (size_t)r * C + cv * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = fmaxf((v[j] - mu[j]) * rs[j] * gm[j] + bt[j], 0.f);
        Vec16<T>::store(yb + (size_t)r * C + cv * V, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                          const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ mean_rstd, T* __restrict__ dx,
                                                          float* dgamma, float* dbeta, int HW, int C, int G, int CS) {
    constexpr int V = Vec16<T>::VEC;
    __shared__ float sg[2 * 128];   // per group: sum g*gamma, sum g*gamma*xhat
    __shared__ float sc[2 * 512];   // per channel: sum g*xhat, sum g
    __shared__ float kst[3][512];   // per channel: group mean, group rstd, gamma
    __shared__ float pp[2][256 * V];  // per-thread partial sums [row lane][channel], added up in a fixed order: the gradient that
                                      // flows on to the geometric head and the backbone is bit-reproducible (no float atomics)
    const int n = blockIdx.x, cpg = C / G, c0 = blockIdx.y * CS, g0 = c0 / cpg, GS = CS / cpg;
    const int tpr = CS / V, rpp = 256 / tpr;
    const int cv = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    for (int i = threadIdx.x; i < CS; i += 256) {
        const int gi = g0 + i / cpg;
        kst[0][i] = mean_rstd[((size_t)n * G + gi) * 2 + 0];
        kst[1][i] = mean_rstd[((size_t)n * G + gi) * 2 + 1];
        kst[2][i] = gamma[c0 + i];
    }
    __syncthreads();
    const size_t base = (size_t)n * HW * C + c0;
    float mu[V], rs[V], gm[V], a1[V], a2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = cv * V + j;
        mu[j] = kst[0][c];
        rs[j] = kst[1][c];
        gm[j] = kst[2][c];
        a1[j] = 0.f; a2[j] = 0.f;
    }
    // (r6b) as in the forward kernel: the masked gradient (exactly g or 0: it packs losslessly) and x of a thread's rows stay in registers between
    // the reduction pass and the apply pass -- three tensors are read once instead of twice
    constexpr int GN_NC = 16;
    const bool cached = HW <= GN_NC * rpp;   // (uniform)
    uint4 gc[GN_NC], xc[GN_NC];
    if (cached) {
#pragma unroll
        for (int u = 0; u < GN_NC; ++u) {
            const int r = rl + u * rpp;
            if (r < HW) {
                const size_t o = base + (size_t)r * C + cv * V;
                float g[V], yv[V];
                Vec16<T>::load(dy + o, g);
                Vec16<T>::load(y + o, yv);
                xc[u] = *reinterpret_cast<const uint4*>(x + o);
#pragma unroll
                for (int j = 0; j < V; ++j) g[j] = (yv[j] > 0.f) ? g[j] : 0.f;
                gc[u] = Vec16<T>::pack(g);
            }
        }
#pragma unroll
        for (int u = 0; u < GN_NC; ++u) {
            const int r = rl + u * rpp;
            if (r < HW) {
                float g[V], xv[V];
                Vec16<T>::unpack(gc[u], g);
                Vec16<T>::unpack(xc[u], xv);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    a1[j] += g[j];
                    a2[j] += g[j] * (xv[j] - mu[j]) * rs[j];
                }
            }
        }
    } else {
        for (int r = rl; r < HW; r += rpp) {
            float g[V], yv[V], xv[V];
            const size_t o = base + (size_t)r * C + cv * V;
            Vec16<T>::load(dy + o, g);
            Vec16<T>::load(y + o, yv);
            Vec16<T>::load(x + o, xv);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float gg = (yv[j] > 0.f) ? g[j] : 0.f;
                a1[j] += gg;
                a2[j] += gg * (xv[j] - mu[j]) * rs[j];
            }
        }
    }
    if (rl < rpp) {
#pragma unroll
        for (int j = 0; j < V; ++j) { pp[0][rl * CS + cv * V + j] = a2[j]; pp[1][rl * CS + cv * V + j] = a1[j]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * CS; i += 256) {  // per-channel totals over the row lanes
        const int st = i / CS, c = i - st * CS;
        float t = 0.f;
        for (int r = 0; r < rpp; ++r) t += pp[st][r * CS + c];
        sc[i] = t;
    }
    __syncthreads();
    for (int gi = threadIdx.x; gi < 2 * GS; gi += 256) {  // per-group: sum_c gamma_c * (sum g | sum g*xhat)
        const int st = gi / GS, gq = gi - st * GS;
        float t = 0.f;
        for (int k = 0; k < cpg; ++k) t += sc[(st == 0 ? CS : 0) + gq * cpg + k] * kst[2][gq * cpg + k];
        sg[gi] = t;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < CS; c += 256) {
        unsafeAtomicAdd(&dgamma[c0 + c], sc[c]);
        unsafeAtomicAdd(&dbeta[c0 + c], sc[CS + c]);
    }
    const float inv_m = 1.f / ((float)HW * cpg);
    float A[V], B[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int gi = (cv * V + j) / cpg;
        A[j] = sg[gi] * inv_m;
        B[j] = sg[GS + gi] * inv_m;
    }
    if (cached) {
#pragma unroll
        for (int u = 0; u < GN_NC; ++u) {
            const int r = rl + u * rpp;
            if (r < HW) {
                float g[V], xv[V], o[V];
                Vec16<T>::unpack(gc[u], g);
                Vec16<T>::unpack(xc[u], xv);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float xh = (xv[j] - mu[j]) * rs[j];
                    o[j] = rs[j] * (g[j] * gm[j] - A[j] - xh * B[j]);
                }
                Vec16<T>::store(dx + base + (size_t)r * C + cv * V, o);
            }
        }
        return;
    }
    for (int r = rl; r < HW; r += rpp) {
        float g[V], yv[V], xv[V], o[V];
        const size_t off = base + (size_t)r * C + cv * V;
        Vec16<T>::load(dy + off, g);
        Vec16<T>::load(y + off, yv);
        Vec16<T>::load(x + off, xv);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float gg = (yv[j] > 0.f) ? g[j] : 0.f;
            const float xh = (xv[j] - mu[j]) * rs[j];
            o[j] = rs[j] * (gg * gm[j] - A[j] - xh * B[j]);
        }
        Vec16<T>::store(dx + off, o);
    }
}

// ------------------------------------------------------------------------------------------ misc
template <typename T>
__global__ __launch_bounds__(256) void leaky_bwd_kernel(const T* dy, const T* y, T* dx, long long nvec) {
    constexpr int V = Vec16<T>::VEC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float g[V], yv[V];
        Vec16<T>::load(dy + i * V, g);
        Vec16<T>::load(y + i * V, yv);
#pragma unroll
        for (int j = 0; j < V; ++j) g[j] = (yv[j] > 0.f) ? g[j] : 0.1f * g[j];
        Vec16<T>::store(dx + i * V, g);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ dy, int cs, int rows, int C, float* db,
                                                        int rows_per_block) {
    constexpr int V = Vec16<T>::VEC;
    __shared__ float acc[1024];
    const int tpr = cs / V;  // <= 256
    const int rpp = 256 / tpr;
    const int cv = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    for (int i = threadIdx.x; i < cs; i += 256) acc[i] = 0.f;
    __syncthreads();
    float s[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = 0.f;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    if (rl < rpp) {
        for (int r = r0 + rl; r < r1; r += rpp) {
            float v[V];
            Vec16<T>::load(dy + (size_t)r * cs + cv * V, v);
#pragma unroll
            for (int j = 0; j < V; ++j) s[j] += v[j];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) atomicAdd(&acc[cv * V + j], s[j]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) unsafeAtomicAdd(&db[c], acc[c]);
}

// rows per block / block count for the channel-stationary row-walking kernels (~8 blocks per CU)
inline void ew_rows(long long npix, int tpr, int* rpb, int* blocks) {
    const int rpp = 256 / tpr;
    long long r = (npix + 2047) / 2048;
    r = (r + rpp - 1) / rpp * rpp;
    if (r < 4LL * rpp) r = 4LL * rpp;
    *rpb = (int)r;
    *blocks = (int)((npix + r - 1) / r);
}

inline int ew_grid(long long n) { return (int)std::min<long long>((n + 255) / 256, 256LL * 16); }

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)
#define DISPATCH(dtype, CALL_F32, CALL_BF16)                \
    do {                                                    \
        if ((dtype) == GDRN_DT_F32) { CALL_F32; }           \
        else if ((dtype) == GDRN_DT_H16) { CALL_BF16; }    \
        else return GDRN_ERR_ARG;                           \
    } while (0)

extern "C" int gdrn_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, long long* nbt, float momentum, float eps,
                                float* mean, float* invstd, float* scale, float* shift, double* ws, void* stream) {
    (void)ws;  // no workspace any more: one launch whatever the row count
    if (!partial || !gamma || !beta || !mean || !invstd || !scale || !shift || rows <= 0 || C <= 0 || (C & 3)) return GDRN_ERR_ARG;
    GDRN_LAUNCH(bn_finalize_rows_kernel, dim3(C / 4), dim3(256), 0, ST, partial, rows, C, count, gamma, beta, running_mean,
                       running_var, nbt, momentum, eps, mean, invstd, scale, shift);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_bn_bwd_coef(const float* rows, int nrows, int C, long long npix, const float* gamma, const float* mean,
                                const float* invstd, float* a, float* b, float* c, float* dgamma, float* dbeta, void* stream) {
    if (!rows || !gamma || !mean || !invstd || !a || !b || !c || nrows <= 0 || C <= 0 || (C & 3) || npix <= 0) return GDRN_ERR_ARG;
    if ((dgamma != nullptr) != (dbeta != nullptr)) return GDRN_ERR_ARG;
    GDRN_LAUNCH(bn_bwd_coef_kernel, dim3(C / 4), dim3(256), 0, ST, rows, nrows, C, (float)(1.0 / (double)npix), gamma, mean,
                       invstd, a, b, c, dgamma, dbeta);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_bn_eval_params(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                   float* scale, float* shift, void* stream) {
    if (!gamma || !beta || !rm || !rv || !scale || !shift || C <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(bn_eval_params_kernel, dim3(cdiv(C, 256)), dim3(256), 0, ST, gamma, beta, rm, rv, eps, C, scale, shift);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_bn_apply(const void* x, const float* scale, const float* shift, const void* residual, void* y,
                             long long npix, int C, int relu, int dtype, void* stream) {
    if (!x || !scale || !shift || !y || npix <= 0 || C <= 0 || (C % 8)) return GDRN_ERR_ARG;
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    if ((C / V) > 256 || 256 % (C / V)) return GDRN_ERR_SHAPE;
    int rpb, blocks;
    ew_rows(npix, C / V, &rpb, &blocks);
    DISPATCH(dtype,
             GDRN_LAUNCH(bn_apply_kernel<float>, dim3(blocks), dim3(256), 0, ST, (const float*)x, scale,
                                shift, (const float*)residual, (float*)y, npix, C, relu, rpb),
             GDRN_LAUNCH(bn_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, (const bf16_t*)x,
                                scale, shift, (const bf16_t*)residual, (bf16_t*)y, npix, C, relu, rpb));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// grid of the reduce pass: >= 4 rows per thread, at most 1024 workgroups (= partial rows)
static void bwd_reduce_grid(long long npix, int C, int dtype, int* rpb_out, int* blocks_out) {
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    const int rpp = 256 / (C / V);
    int rpb = rpp * 4;  // >= 4 rows per thread, as many workgroups as that allows (small layers were latency-bound at 64)
    long long blocks = (npix + rpb - 1) / rpb;
    if (blocks > 1024) { rpb = (int)(((npix + 1023) / 1024 + rpp - 1) / rpp * rpp); blocks = (npix + rpb - 1) / rpb; }
    *rpb_out = rpb;
    *blocks_out = (int)blocks;
}

extern "C" int gdrn_bn_bwd_reduce_rows(long long npix, int C, int dtype) {
    if (npix <= 0 || C <= 0 || C > 512 || (C % 8)) return GDRN_ERR_ARG;
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    if (256 % (C / V)) return GDRN_ERR_SHAPE;
    int rpb, blocks;
    bwd_reduce_grid(npix, C, dtype, &rpb, &blocks);
    return blocks;
}

extern "C" int gdrn_bn_bwd_reduce(const void* dy, const void* ymask, const void* x, const float* mean, const float* invstd,
                                  const float* mask_scale, const float* mask_shift, long long npix, int C, float* rows, int dtype,
                                  void* stream) {
    if (!dy || !x || !mean || !invstd || !rows || npix <= 0 || C <= 0 || C > 512 || (C % 8)) return GDRN_ERR_ARG;
    if ((mask_scale != nullptr) != (mask_shift != nullptr)) return GDRN_ERR_ARG;
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    if (256 % (C / V)) return GDRN_ERR_SHAPE;
    int rpb, blocks;
    bwd_reduce_grid(npix, C, dtype, &rpb, &blocks);
    DISPATCH(dtype,
             GDRN_LAUNCH(bn_bwd_reduce_kernel<float>, dim3(blocks), dim3(256), 0, ST, (const float*)dy,
                                (const float*)ymask, (const float*)x, mean, invstd, mask_scale, mask_shift, npix, C, rows, rpb),
             GDRN_LAUNCH(bn_bwd_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, (const bf16_t*)dy,
                                (const bf16_t*)ymask, (const bf16_t*)x, mean, invstd, mask_scale, mask_shift, npix, C, rows, rpb));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_bn_bwd_apply(const void* dy, const void* ymask, const void* x, const float* a, const float* b, const float* c,
                                 const float* mask_scale, const float* mask_shift, long long npix, int C, void* dx, void* g_out,
                                 int dtype, void* stream) {
    if (!dy || !x || !a || !b || !c || !dx || npix <= 0 || C <= 0 || C > 512 || (C % 8)) return GDRN_ERR_ARG;
    if ((mask_scale != nullptr) != (mask_shift != nullptr)) return GDRN_ERR_ARG;
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    if ((C / V) > 256 || 256 % (C / V)) return GDRN_ERR_SHAPE;
    int rpb, blocks;
    ew_rows(npix, C / V, &rpb, &blocks);
    DISPATCH(dtype,
             GDRN_LAUNCH(bn_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, ST, (const float*)dy,
                                (const float*)ymask, (const float*)x, a, b, c, mask_scale, mask_shift, npix, C,
                                (float*)dx, (float*)g_out, rpb),
             GDRN_LAUNCH(bn_bwd_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, (const bf16_t*)dy,
                                (const bf16_t*)ymask, (const bf16_t*)x, a, b, c, mask_scale, mask_shift, npix, C,
                                (bf16_t*)dx, (bf16_t*)g_out, rpb));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, unsigned char* idx,
                                        int N, int H, int W, int C, int dtype, void* stream) {
    if (!x || !scale || !shift || !y || !idx || N <= 0 || (H & 1) || (W & 1) || (C % 8)) return GDRN_ERR_ARG;
    if (C > 512 || (long long)N * H * W * C / 4 >= (1ll << 31)) return GDRN_ERR_SHAPE;
    const long long n = (long long)N * (H / 2) * (W / 2) * C;
    DISPATCH(dtype,
             GDRN_LAUNCH(bn_relu_maxpool_fwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, ST, (const float*)x,
                                scale, shift, (float*)y, idx, N, H, W, C),
             GDRN_LAUNCH(bn_relu_maxpool_fwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, ST, (const bf16_t*)x,
                                scale, shift, (bf16_t*)y, idx, N, H, W, C));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_maxpool_bwd_rows(int N, int H, int W, int C, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return GDRN_ERR_ARG;
    const long long n = (long long)N * H * W * C;
    return ew_grid(n / (dtype == GDRN_DT_H16 ? 8 : 4));
}

extern "C" int gdrn_maxpool_bwd(const void* dy, const unsigned char* idx, const void* x, const float* scale,
                                const float* shift, void* g, int N, int H, int W, int C, const float* mean, const float* invstd,
                                float* rows, int dtype, void* stream) {
    if (!dy || !idx || !x || !scale || !shift || !g || N <= 0 || (H & 1) || (W & 1) || (C % 8)) return GDRN_ERR_ARG;
    if (C > 512 || (long long)N * H * W * C / 4 >= (1ll << 31)) return GDRN_ERR_SHAPE;
    if (rows != nullptr) {
        if (!mean || !invstd) return GDRN_ERR_ARG;
        const int cvn = C / (dtype == GDRN_DT_H16 ? 8 : 4);
        if (cvn > 64 || 64 % cvn) return GDRN_ERR_SHAPE;  // a thread keeps one channel vector over its grid-stride iterations
    }
    const long long n = (long long)N * H * W * C;
    DISPATCH(dtype,
             GDRN_LAUNCH(maxpool_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, ST, (const float*)dy, idx,
                                (const float*)x, scale, shift, (float*)g, N, H, W, C, mean, invstd, rows),
             GDRN_LAUNCH(maxpool_bwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, ST, (const bf16_t*)dy, idx,
                                (const bf16_t*)x, scale, shift, (bf16_t*)g, N, H, W, C, mean, invstd, rows));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_upsample2x_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
    if (!x || !y || N <= 0 || H < 2 || W < 2 || (C % 8)) return GDRN_ERR_ARG;
    if ((long long)N * H * W * C >= (1ll << 31)) return GDRN_ERR_SHAPE;  // 4x upsampled element count / 4 per thread, 32-bit index math
    const long long n = (long long)N * 4 * H * W * C;
    int lg = 0;
    const float up_sh = (float)(H - 1) / (float)(2 * H - 1), up_sw = (float)(W - 1) / (float)(2 * W - 1);
    if (upsample_rows_ok(N, H, W, C, dtype, &lg)) {   // one workgroup per output row (see upsample2x_rows_kernel)
        DISPATCH(dtype,
                 GDRN_LAUNCH((upsample2x_rows_kernel<float, false>), dim3(N * 2 * H), dim3(256), 0, ST, (const float*)x, (float*)y, N, H, W, C, lg, (const float*)nullptr, (const float*)nullptr, up_sh, up_sw),
                 GDRN_LAUNCH((upsample2x_rows_kernel<bf16_t, false>), dim3(N * 2 * H), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, lg, (const float*)nullptr, (const float*)nullptr, up_sh, up_sw));
        GDRN_CHECK_LAUNCH();
        return GDRN_OK;
    }
    DISPATCH(dtype,
             GDRN_LAUNCH(upsample2x_fwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, ST, (const float*)x, (float*)y, N, H, W, C),
             GDRN_LAUNCH(upsample2x_fwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, ST, (const bf16_t*)x, (bf16_t*)y, N, H, W, C));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_upsample2x_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, void* stream) {
    if (!dy || !dx || N <= 0 || H < 2 || W < 2 || (C % 8)) return GDRN_ERR_ARG;
    if ((long long)N * H * W * C >= (1ll << 31)) return GDRN_ERR_SHAPE;
    const long long n = (long long)N * H * W * C;
    DISPATCH(dtype,
             GDRN_LAUNCH(upsample2x_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, ST, (const float*)dy, (float*)dx, N, H, W, C),
             GDRN_LAUNCH(upsample2x_bwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, ST, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_bn_relu_upsample2x_fwd(const void* x_raw, const float* scale, const float* shift, void* y, int N, int H, int W, int C, int dtype, void* stream) {
    if (!x_raw || !scale || !shift || !y || N <= 0 || H < 2 || W < 2 || (C % 8)) return GDRN_ERR_ARG;
    if ((long long)N * H * W * C >= (1ll << 31)) return GDRN_ERR_SHAPE;
    const long long n = (long long)N * 4 * H * W * C;
    int lg = 0;
    const float up_sh = (float)(H - 1) / (float)(2 * H - 1), up_sw = (float)(W - 1) / (float)(2 * W - 1);
    if (upsample_rows_ok(N, H, W, C, dtype, &lg)) {
        const size_t lds = (size_t)2 * W * C * (dtype == GDRN_DT_H16 ? 2 : 4);
        DISPATCH(dtype,
                 GDRN_LAUNCH((upsample2x_rows_kernel<float, true>), dim3(N * 2 * H), dim3(256), lds, ST, (const float*)x_raw, (float*)y, N, H, W, C, lg, scale, shift, up_sh, up_sw),
                 GDRN_LAUNCH((upsample2x_rows_kernel<bf16_t, true>), dim3(N * 2 * H), dim3(256), lds, ST, (const bf16_t*)x_raw, (bf16_t*)y, N, H, W, C, lg, scale, shift, up_sh, up_sw));
        GDRN_CHECK_LAUNCH();
        return GDRN_OK;
    }
    DISPATCH(dtype,
             GDRN_LAUNCH((upsample2x_fwd_kernel<float, true>), dim3(ew_grid(n / 4)), dim3(256), 0, ST, (const float*)x_raw, (float*)y, N, H, W, C, scale, shift),
             GDRN_LAUNCH((upsample2x_fwd_kernel<bf16_t, true>), dim3(ew_grid(n / 8)), dim3(256), 0, ST, (const bf16_t*)x_raw, (bf16_t*)y, N, H, W, C, scale, shift));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// upsampling backward + the reduction pass of the BatchNorm(+ReLU) backward that consumes its result (gdrn_bn_bwd_reduce with the affine ReLU
// mask), in one launch: dx (unmasked, as gdrn_upsample2x_bwd writes it) and one partial row [2][C] per workgroup of
// (sum g, sum g * xhat), g = dx where mask_scale * x + mask_shift > 0 -- rows = gdrn_bn_bwd_reduce_rows(N*H*W, C, dtype), for gdrn_bn_bwd_coef
extern "C" int gdrn_upsample2x_bwd_bnsums(const void* dy, void* dx, const void* x_raw, const float* mean, const float* invstd, const float* mask_scale,
                                          const float* mask_shift, int N, int H, int W, int C, float* rows, int dtype, void* stream) {
    if (!dy || !dx || !x_raw || !mean || !invstd || !mask_scale || !mask_shift || !rows || N <= 0 || H < 2 || W < 2 || (C % 8) || C > 512) return GDRN_ERR_ARG;
    if ((long long)N * H * W * C >= (1ll << 31)) return GDRN_ERR_SHAPE;
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    if (256 % (C / V)) return GDRN_ERR_SHAPE;
    int rpb, blocks;
    bwd_reduce_grid((long long)N * H * W, C, dtype, &rpb, &blocks);
    DISPATCH(dtype,
             GDRN_LAUNCH(upsample2x_bwd_bnsums_kernel<float>, dim3(blocks), dim3(256), 0, ST, (const float*)dy, (float*)dx, (const float*)x_raw, mean, invstd,
                         mask_scale, mask_shift, N, H, W, C, rows, rpb),
             GDRN_LAUNCH(upsample2x_bwd_bnsums_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, (const bf16_t*)dy, (bf16_t*)dx, (const bf16_t*)x_raw, mean, invstd,
                         mask_scale, mask_shift, N, H, W, C, rows, rpb));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// channel slab per workgroup for the GroupNorm kernels: 32 channels when groups and 16-byte vectors tile it, else all of C
static int gn_slab(int C, int G, int V) {
    const int cpg = C / G;
    if (C % 32 == 0 && 32 % cpg == 0 && 32 % V == 0) return 32;
    return C;
}

extern "C" int gdrn_gn_relu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int N,
                                int HW, int C, int G, float eps, int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || !mean_rstd || N <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 128 || C > 512 || (C % G) ||
        (C % 8))
        return GDRN_ERR_ARG;
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    const int CS = gn_slab(C, G, V);
    if (256 % (CS / V)) return GDRN_ERR_SHAPE;
    DISPATCH(dtype,
             GDRN_LAUNCH(gn_relu_fwd_kernel<float>, dim3(N, C / CS), dim3(256), 0, ST, (const float*)x, gamma, beta, (float*)y, mean_rstd, HW, C, G, eps, CS),
             GDRN_LAUNCH(gn_relu_fwd_kernel<bf16_t>, dim3(N, C / CS), dim3(256), 0, ST, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean_rstd, HW, C, G, eps, CS));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_gn_relu_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* mean_rstd,
                                void* dx, float* dgamma, float* dbeta, int N, int HW, int C, int G, int dtype, void* stream) {
    if (!dy || !y || !x || !gamma || !mean_rstd || !dx || !dgamma || !dbeta || N <= 0 || HW <= 0 || G <= 0 || G > 128 ||
        C > 512 || (C % G) || (C % 8))
        return GDRN_ERR_ARG;
    const bool prezeroed = (dtype & GDRN_PREZEROED) != 0;
    dtype &= ~GDRN_PREZEROED;
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    const int CS = gn_slab(C, G, V);
    if (256 % (CS / V)) return GDRN_ERR_SHAPE;
    if (!prezeroed) {
        if (hipMemsetAsync(dgamma, 0, C * sizeof(float), ST) != hipSuccess) return GDRN_ERR_LAUNCH;
        if (hipMemsetAsync(dbeta, 0, C * sizeof(float), ST) != hipSuccess) return GDRN_ERR_LAUNCH;
    }
    DISPATCH(dtype,
             GDRN_LAUNCH(gn_relu_bwd_kernel<float>, dim3(N, C / CS), dim3(256), 0, ST, (const float*)dy, (const float*)y,
                                (const float*)x, gamma, mean_rstd, (float*)dx, dgamma, dbeta, HW, C, G, CS),
             GDRN_LAUNCH(gn_relu_bwd_kernel<bf16_t>, dim3(N, C / CS), dim3(256), 0, ST, (const bf16_t*)dy, (const bf16_t*)y,
                                (const bf16_t*)x, gamma, mean_rstd, (bf16_t*)dx, dgamma, dbeta, HW, C, G, CS));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_leaky_bwd(const void* dy, const void* y, void* dx, long long n, int dtype, void* stream) {
    if (!dy || !y || !dx || n <= 0 || (n % 8)) return GDRN_ERR_ARG;
    DISPATCH(dtype,
             GDRN_LAUNCH(leaky_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, ST, (const float*)dy, (const float*)y, (float*)dx, n / 4),
             GDRN_LAUNCH(leaky_bwd_kernel<bf16_t>, dim3(ew_grid(n / 8)), dim3(256), 0, ST, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, n / 8));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_bias_grad(const void* dy, int cs, int rows, int C, float* db, int dtype, void* stream) {
    if (!dy || !db || rows <= 0 || C <= 0 || C > cs || cs > 1024 || (cs % 8)) return GDRN_ERR_ARG;
    const bool prezeroed = (dtype & GDRN_PREZEROED) != 0;
    dtype &= ~GDRN_PREZEROED;
    const int V = dtype == GDRN_DT_H16 ? 8 : 4;
    if ((cs / V) > 256 || 256 % (cs / V)) return GDRN_ERR_SHAPE;
    if (!prezeroed && hipMemsetAsync(db, 0, C * sizeof(float), ST) != hipSuccess) return GDRN_ERR_LAUNCH;
    const int rpp = 256 / (cs / V);
    int rpb = rpp * 16;
    int blocks = cdiv(rows, rpb);
    if (blocks > 1024) { rpb = cdiv(cdiv(rows, 1024), rpp) * rpp; blocks = cdiv(rows, rpb); }
    DISPATCH(dtype,
             GDRN_LAUNCH(bias_grad_kernel<float>, dim3(blocks), dim3(256), 0, ST, (const float*)dy, cs, rows, C, db, rpb),
             GDRN_LAUNCH(bias_grad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, (const bf16_t*)dy, cs, rows, C, db, rpb));
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// Skinny-M linear layer for gfx950 (bf16 operands, fp32 accumulate):  y[m][n] = act(sum_k x[m][k] * w[n][k] + bias[n]),
// M <= 64 (the RoI batch), e.g. Patch-PnP's fc1: 64 x 8192 -> 1024 (conv_pnp_net.py:85-92,152).
//
// As an 8x8 "valid" convolution on the generic gather kernel this layer is one M tile x 8 N tiles = 8 workgroups that
// each stream 2 MB of weights serially (129 us, 8 TFLOP/s).  The layer is bound by reading w once (16.8 MB): split K.
// A workgroup owns 16 output columns and a K range; its four waves take a quarter of that range each and feed both MFMA
// operands straight from global memory (x rows and w rows are K-contiguous, so a lane's 8-element fragment is one
// 16-byte load; no LDS staging, no transposes).  Partial sums meet in an fp32 workspace through atomics; the workgroup
// that draws the last ticket of its column tile applies bias + activation, writes y and leaves the workspace zeroed.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

__global__ __launch_bounds__(256) void linear_splitk_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ y, int M, int K,
                                                            int N, int x_rs, int w_rs, int y_rs, int act, int kper, float* ws,
                                                            unsigned int* tickets) {
    __shared__ float red[4][4][64][4];  // [wave][m fragment][lane][4]
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int n0 = blockIdx.x * 16, S = gridDim.y;
    const int kq = kper >> 2;  // K elements per wave
    const int k_begin = blockIdx.y * kper + wave * kq;
    const bf16_t* wp = w + (size_t)(n0 + r16) * w_rs + k_begin + g * 8;
    const bf16_t* xp[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) xp[f] = x + (size_t)min(f * 16 + r16, M - 1) * x_rs + k_begin + g * 8;
    f32x4_t acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    // four k-steps per trip: all 20 fragment loads are issued before the first MFMA (the loads come straight from
    // L2/HBM; one dependent round trip per k-step made the first version latency-bound).  Rows >= M re-read row M-1
    // (clamped pointers) and are dropped at the store: no per-load condition -- hipcc would branch around each load and
    // wait for it separately.
    for (; k + 128 <= kq; k += 128) {
        uint4 bq[4], aq[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bq[u] = *reinterpret_cast<const uint4*>(wp + k + u * 32);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int u = 0; u < 4; ++u) aq[f][u] = *reinterpret_cast<const uint4*>(xp[f] + k + u * 32);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aq[f][u]), __builtin_bit_cast(bf16x8_t, bq[u]),
                                                                 acc[f], 0, 0, 0);
    }
    for (; k < kq; k += 32) {
        const bf16x8_t b = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp + k));
        uint4 aq[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) aq[f] = *reinterpret_cast<const uint4*>(xp[f] + k);
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aq[f]), b, acc[f], 0, 0, 0);
    }
    // D[i = g*4 + j (row m of the fragment)][col = r16 (column n0 + r16)]
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][f][lane][j] = acc[f][j];
    __syncthreads();
    // 1024 partial outputs (64 rows x 16 columns): 4 per thread, summed over the four waves
    float* wsb = ws + (size_t)n0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;             // (f, lane, j) flattened
        const int f = e >> 8, ln = (e >> 2) & 63, j = e & 3;
        const int m = f * 16 + (ln >> 4) * 4 + j, c = ln & 15;
        if (m < M) {
            const float v = red[0][f][ln][j] + red[1][f][ln][j] + red[2][f][ln][j] + red[3][f][ln][j];
            if (S > 1) unsafeAtomicAdd(wsb + (size_t)m * N + c, v);
            else red[0][f][ln][j] = v;
        }
    }
    if (S > 1) {
        __threadfence();
        __syncthreads();
        if (tid == 0) is_last = (atomicAdd(&tickets[blockIdx.x], 1u) == (unsigned)(S - 1));
        __syncthreads();
        if (!is_last) return;
        __threadfence();
    } else {
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;
        const int m = e >> 4, c = e & 15;         // row-major over the 64 x 16 tile: 32-byte rows of y
        if (m < M) {
            float v;
            if (S > 1) v = __hip_atomic_exchange(wsb + (size_t)m * N + c, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = red[0][m >> 4][((m & 15) >> 2) * 16 + c][m & 3];
            if (bias != nullptr) v += bias[n0 + c];
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = v > 0.f ? v : 0.1f * v;
            y[(size_t)m * y_rs + n0 + c] = f2bf(v);
        }
    }
    if (S > 1 && tid == 0) tickets[blockIdx.x] = 0u;
}

}  // namespace

// ws: M*N floats followed by N/16 uint32 tickets, zeroed ONCE by the caller (left zeroed).  bf16 only.
extern "C" int gdrn_linear_splitk(const void* x, const void* w, const float* bias, void* y, int M, int K, int N, int x_rs, int w_rs,
                                  int y_rs, int act, float* ws, int dtype, void* stream) {
    if (!x || !w || !y || !ws || M <= 0 || M > 64 || K <= 0 || N <= 0) return GDRN_ERR_ARG;
    if (dtype != GDRN_DT_BF16) return GDRN_ERR_SHAPE;
    if ((N % 16) || (K % 128) || (x_rs % 8) || (w_rs % 8) || x_rs < K || w_rs < K || y_rs < N) return GDRN_ERR_SHAPE;
    // K range per workgroup: a multiple of 128 (32 per wave-step x 4 waves); ~1024 workgroups
    const int ntile = N / 16;
    int S = std::max(1, std::min(K / 128, cdiv(1024, ntile)));
    while (S > 1 && (K % (128 * S))) --S;
    const int kper = K / S;
    unsigned int* tickets = reinterpret_cast<unsigned int*>(ws + (size_t)M * N);
    hipLaunchKernelGGL(linear_splitk_kernel, dim3(ntile, S), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const bf16_t*)x,
                       (const bf16_t*)w, bias, (bf16_t*)y, M, K, N, x_rs, w_rs, y_rs, act, kper, ws, tickets);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

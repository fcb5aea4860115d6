// Skinny-M linear layer for gfx950 (bf16 operands, fp32 accumulate):  y[m][n] = act(sum_k x[m][k] * w[n][k] + bias[n]),
// M <= 64 (the RoI batch), e.g. Patch-PnP's fc1: 64 x 8192 -> 1024 (conv_pnp_net.py:85-92,152).
//
// As an 8x8 "valid" convolution on the generic gather kernel this layer is one M tile x 8 N tiles = 8 workgroups that
// each stream 2 MB of weights serially (129 us, 8 TFLOP/s).  The layer is bound by reading w once (16.8 MB): split K.
// A workgroup owns 16 output columns and a K range; its four waves take a quarter of that range each and feed both MFMA
// operands straight from global memory (x rows and w rows are K-contiguous, so a lane's 8-element fragment is one
// 16-byte load; no LDS staging, no transposes).  Partial sums go to per-split slabs of an fp32 workspace; a second tiny
// kernel adds them, applies bias + activation and writes y.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

__global__ __launch_bounds__(256) void linear_splitk_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, int M, int K,
                                                            int N, int x_rs, int w_rs, int kper, float* ws) {
    __shared__ float red[4][4][64][4];  // [wave][m fragment][lane][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    const int n0 = blockIdx.x * 16;
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
                acc[f] = GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, aq[f][u]), __builtin_bit_cast(bf16x8_t, bq[u]),
                                                                 acc[f]);
    }
    for (; k < kq; k += 32) {
        const bf16x8_t b = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wp + k));
        uint4 aq[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) aq[f] = *reinterpret_cast<const uint4*>(xp[f] + k);
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, aq[f]), b, acc[f]);
    }
    // D[i = g*4 + j (row m of the fragment)][col = r16 (column n0 + r16)]
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][f][lane][j] = acc[f][j];
    __syncthreads();
    // 1024 partial outputs (64 rows x 16 columns): 4 per thread, summed over the four waves, stored to this split's slab
    // of the workspace with plain stores.  A second tiny kernel adds the S slabs in a fixed order and applies bias +
    // activation.  (Tried and rejected: fp32 atomics into one slab, and a last-arriver ticket in this kernel -- the
    // agent-scope release/acquire that hand-off needs writes back the XCD's whole L2 per workgroup: 75-150 us.)
    float* dst = ws + (size_t)blockIdx.y * M * N + n0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i;             // (f, lane, j) flattened
        const int f = e >> 8, ln = (e >> 2) & 63, j = e & 3;
        const int m = f * 16 + (ln >> 4) * 4 + j, c = ln & 15;
        if (m < M) dst[(size_t)m * N + c] = red[0][f][ln][j] + red[1][f][ln][j] + red[2][f][ln][j] + red[3][f][ln][j];
    }
}

__global__ __launch_bounds__(256) void linear_finish_kernel(const float* __restrict__ ws, int S, const float* __restrict__ bias,
                                                            bf16_t* __restrict__ y, int M, int N, int y_rs, int act) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    const int m = i / N, n = i - m * N;
    float v = 0.f;
#pragma unroll 4
    for (int sp = 0; sp < S; ++sp) v += ws[(size_t)sp * M * N + i];
    if (bias != nullptr) v += bias[n];
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = v > 0.f ? v : 0.1f * v;
    y[(size_t)m * y_rs + n] = f2bf(v);
}

}  // namespace

// number of K ranges (= slabs [M][N] of ws) gdrn_linear_splitk uses for a K x N layer: a multiple of 128 per range, ~1024 workgroups
extern "C" int gdrn_linear_splits(int K, int N) {
    if (K <= 0 || N <= 0 || (N % 16) || (K % 128)) return GDRN_ERR_SHAPE;
    const int ntile = N / 16;
    int S = std::max(1, std::min(std::min(K / 128, GDRN_LINEAR_MAX_SPLITS), cdiv(1024, ntile)));
    while (S > 1 && (K % (128 * S))) --S;
    return S;
}

// ws: GDRN_LINEAR_MAX_SPLITS * M*N floats (per-split partial slabs; no initialisation needed).  bf16 only.
extern "C" int gdrn_linear_splitk(const void* x, const void* w, const float* bias, void* y, int M, int K, int N, int x_rs, int w_rs,
                                  int y_rs, int act, float* ws, int dtype, void* stream) {
    if (!x || !w || (!y && act != GDRN_LINEAR_NO_FINISH) || !ws || M <= 0 || M > 64 || K <= 0 || N <= 0) return GDRN_ERR_ARG;
    if (dtype != GDRN_DT_H16) return GDRN_ERR_SHAPE;
    if ((N % 16) || (K % 128) || (x_rs % 8) || (w_rs % 8) || x_rs < K || w_rs < K || y_rs < N) return GDRN_ERR_SHAPE;
    // K range per workgroup: a multiple of 128 (32 per wave-step x 4 waves); ~1024 workgroups
    const int ntile = N / 16;
    const int S = gdrn_linear_splits(K, N);
    const int kper = K / S;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GDRN_LAUNCH(linear_splitk_kernel, dim3(ntile, S), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, M, K, N, x_rs, w_rs,
                       kper, ws);
    if (act != GDRN_LINEAR_NO_FINISH)   // (ABI 5) the caller's next kernel adds the gdrn_linear_splits(K, N) slabs itself (gdrn_pose_loss: fc2_ws)
        GDRN_LAUNCH(linear_finish_kernel, dim3(cdiv(M * N, 256)), dim3(256), 0, st, ws, S, bias, (bf16_t*)y, M, N, y_rs, act);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

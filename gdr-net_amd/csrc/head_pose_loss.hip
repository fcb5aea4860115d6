// Head tail (slice / softmax / concat / extent scaling), map losses, pose decode and pose losses of
// the GDR-Net RoI path for gfx950 -- forward and backward, no host synchronisation.
// Reference call sites: GDRN.py:156-169 (glue), conv_pnp_net.py:121-125, GDRN.py:345-471 (losses),
// rot_reps.py:34-49, pose_from_pred_centroid_z.py:52-227, core/utils/utils.py:39-94,208-236,
// pose_utils.py:323-370,430-482, pm_loss.py:82-114, lib/pysixd/misc.py:930-949, pose_error.py:400-425.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

// ------------------------------------------------------------------------------------------ head tail fwd
// 16 lanes per pixel (one DPP row), 4 pixels per wave: lane q holds classes q, q+16, ... -- contiguous 64-byte loads,
// row reductions in 4 DPP adds.  (The first version used one wave per pixel with 6-step cross-lane reductions and ran
// at ~0.5 TB/s.)
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void head_tail_fwd_kernel(const float* __restrict__ head, int hs,
                                                            const float* __restrict__ coord2d,
                                                            const float* __restrict__ extents, T* __restrict__ pnp, int pcs,
                                                            int N, int HW, int nreg) {
    const int q = threadIdx.x & 15;
    const long long M = (long long)N * HW;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
    for (long long m = g0; m < M; m += ng) {
        const int n = (int)(m / HW), pix = (int)(m - (long long)n * HW);
        const float* h = head + m * hs;
        float r[4], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = (q + 16 * j < nreg) ? h[5 + q + 16 * j] : -INFINITY;
            mx = fmaxf(mx, r[j]);
        }
        mx = row16_max(mx);
        float e[4], se = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = (q + 16 * j < nreg) ? expf(r[j] - mx) : 0.f;
            se += e[j];
        }
        se = row16_sum(se);
        T* o = pnp + m * pcs;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q + 16 * j < nreg) st1<T>(o + 5 + q + 16 * j, e[j] / se);
        if (q < 3) st1<T>(o + q, (h[1 + q] - 0.5f) * extents[n * 3 + q]);
        else if (q < 5) st1<T>(o + q, coord2d[((size_t)n * 2 + (q - 3)) * HW + pix]);
        for (int c = 5 + nreg + q; c < pcs; c += 16) st1<T>(o + c, 0.f);
    }
}

// ------------------------------------------------------------------------------------------ head tail fwd, nreg == 64
// The shape the network uses (64 regions, 72-float head rows): lane q of a pixel's 16 lanes holds the FOUR consecutive classes 4q..4q+3
// (one 16-byte load at float 4 + 4q; class 64 rides on lane 15), the softmax lands as one 4-channel store per lane, and in train mode
// (LOSS) the same pass accumulates the map losses of gdrn_map_loss_fwd -- the logits are read once instead of twice and the two
// kernels' ~40 scalar loads / stores per lane become ~8 wide ones.
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = h16lo(t.x); v[1] = h16hi(t.x);
    v[2] = h16lo(t.y); v[3] = h16hi(t.y);
}

template <typename T, bool LOSS>
__global__ __launch_bounds__(256) void head_tail_fwd64_kernel(const float* __restrict__ head, int hs, const float* __restrict__ coord2d,
                                                              const float* __restrict__ extents, T* __restrict__ pnp, int pcs,
                                                              const float* __restrict__ gt_xyz, const float* __restrict__ mvis,
                                                              const float* __restrict__ mtr, const long long* __restrict__ gt_region,
                                                              double* acc, int N, int HW, int write_pad, int acc_rows) {
    __shared__ float red[6][16];
    const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long long M = (long long)N * HW;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long m = g0; m < M; m += ng) {
        const int n = (int)(m / HW), pix = (int)(m - (long long)n * HW);
        const float* h = head + m * hs;
        // EVERY load of the iteration goes out here, by all 16 lanes of the pixel (same address: one request): issued where they are used --
        // behind the softmax, inside the `q == 0` / `q == 14` branches -- the loss inputs cost a second memory round trip per iteration
        // (train variant 65 us against 26 us for the eval variant, r4)
        float r[4], h03[4];
        load4<float>(h + 4 + 4 * q, r);                       // classes 4q .. 4q+3
        load4<float>(h, h03);                                 // mask, x, y, z
        const float h68 = h[68];                              // class 64
        const float c2x = coord2d[((size_t)n * 2 + 0) * HW + pix], c2y = coord2d[((size_t)n * 2 + 1) * HW + pix];
        const float ex[3] = {extents[n * 3 + 0], extents[n * 3 + 1], extents[n * 3 + 2]};
        float mv = 0.f, mt = 0.f, gx[3] = {0.f, 0.f, 0.f};
        long long gr = 0;
        if constexpr (LOSS) {
            mv = mvis[m];
            mt = mtr[m];
            gr = gt_region[m];
#pragma unroll
            for (int c = 0; c < 3; ++c) gx[c] = gt_xyz[((size_t)n * 3 + c) * HW + pix];
        }
        const float r64 = (q == 15) ? h68 : -INFINITY;
        // softmax over classes 1..64 (region[:, 1:])
        float mx = fmaxf(fmaxf(q == 0 ? -INFINITY : r[0], r[1]), fmaxf(fmaxf(r[2], r[3]), r64));
        mx = row16_max(mx);
        float e[4], e64;
        e[0] = (q == 0) ? 0.f : expf(r[0] - mx);
#pragma unroll
        for (int j = 1; j < 4; ++j) e[j] = expf(r[j] - mx);
        e64 = (q == 15) ? expf(r64 - mx) : 0.f;
        const float inv = 1.f / row16_sum(e[0] + e[1] + e[2] + e[3] + e64);
        T* o = pnp + m * pcs;
        float v[4] = {e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
        if (q == 0) v[0] = c2y;                                        // channel 4 = second roi_coord_2d channel
        store4<T>(o + 4 + 4 * q, v);                                   // channels 4 + 4q .. 7 + 4q
        if (q == 15) {
            const float w[4] = {e64 * inv, 0.f, 0.f, 0.f};             // channel 68 = class 64, 69..71 pad
            store4<T>(o + 68, w);
        } else if (q == 14) {
            const float w[4] = {(h03[1] - 0.5f) * ex[0], (h03[2] - 0.5f) * ex[1], (h03[3] - 0.5f) * ex[2], c2x};
            store4<T>(o, w);
        }
        if (write_pad) {
            const float z[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = 72 + 4 * q; c < pcs; c += 64) store4<T>(o + c, z);
        }
        if constexpr (LOSS) {
            // cross entropy over the 65 classes of logits * visib (GDRN.py:392-400)
            const int tgt = (int)(gr * (long long)mv);
            float z[4], zt = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                z[j] = r[j] * mv;
                if (4 * q + j == tgt) zt = z[j];
            }
            const float z64 = (q == 15) ? r64 * mv : -INFINITY;
            if (q == 15 && tgt == 64) zt = z64;
            float mz = fmaxf(fmaxf(z[0], z[1]), fmaxf(fmaxf(z[2], z[3]), z64));
            mz = row16_max(mz);
            float es = expf(z[0] - mz) + expf(z[1] - mz) + expf(z[2] - mz) + expf(z[3] - mz) + (q == 15 ? expf(z64 - mz) : 0.f);
            es = row16_sum(es);
            zt = row16_sum(zt);
            if (q == 0) {
                a[4] += (logf(es) + mz) - zt;
                a[5] += mv;
                a[3] += fabsf(h03[0] - mt);
#pragma unroll
                for (int c = 0; c < 3; ++c) a[c] += fabsf(h03[1 + c] * mv - gx[c] * mv);
            }
        }
    }
    if constexpr (LOSS) {
        if (q == 0)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k][grp] = a[k];
        __syncthreads();
        if (threadIdx.x < 6) {
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) v += (double)red[threadIdx.x][i];
            // acc_rows: this workgroup's partial row (summed in a fixed order by gdrn_map_loss_finalize_rows) instead of 6 atomics per
            // workgroup onto one cache line: 24 k same-line atomics across 8 XCDs cost the kernel 15 us
            if (acc_rows) acc[8 + (size_t)blockIdx.x * 8 + threadIdx.x] = v;
            else unsafeAtomicAdd(&acc[threadIdx.x], v);
        }
    }
}

// ------------------------------------------------------------------------------------------ eval mode: 1x1 output conv + head tail, nreg == 64
// The head's last layer -- nn.Conv2d(256, 69, 1) of cdpn_rot_head_region.py:127-135 -- and the glue behind it (GDRN.py:156-169: slice, softmax
// over region[:, 1:], concat with roi_coord_2d, extent scaling) in ONE pass over the 256-channel activation (r5, inference only: in train mode the
// BatchNorm in front of the conv needs the whole tensor's statistics and the losses want the logits).  As two launches the fp32 logits
// (75 MB at bs = 64) are written by a generic 128 x 128 GEMM tile of which 59 columns are padding and read back by the tail (70 + 28 us).
// Here: the 69 x 256 weights sit in LDS in MFMA fragment order (5 fragments of 16 channels x 8 k-steps = 40 KB, shared by the four waves); a
// wave takes 16 pixels at a time, its B fragments (pixel r16, 8 input channels per k-group) straight from global memory, 40 MFMAs with the bias as
// the accumulator's initial value, then parks the 16 x 80 logit tile in its own 5 KB of LDS and re-reads it in the tail's lane map (16 lanes per
// pixel, four classes per lane): the softmax and the stores are head_tail_fwd64_kernel's.  `head` (nullable) still receives the fp32 logits for
// callers that return the maps (cfg.TEST.USE_PNP, gdrn_correspondences).
constexpr int HCT_PITCH = 84;   // floats per pixel row of the wave's logit tile (80 + 4: consecutive pixels 16 bytes apart in the bank map)
// LOSS (train mode, r5): the same pass also accumulates the map losses as head_tail_fwd64_kernel<T, true> does (GDRN.py:345-400; one partial row per
// workgroup, added in a fixed order by gdrn_map_loss_finalize_rows); `head` is then mandatory -- the backward pass reads the logits.
template <typename T, bool LOSS>
__global__ __launch_bounds__(256) void head_conv_tail64_kernel(const bf16_t* __restrict__ x, int x_cs, const bf16_t* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ coord2d,
                                                               const float* __restrict__ extents, float* __restrict__ head, int hs,
                                                               T* __restrict__ pnp, int pcs, int N, int HW, int ngroups,
                                                               const float* __restrict__ gt_xyz, const float* __restrict__ mvis,
                                                               const float* __restrict__ mtr, const long long* __restrict__ gt_region, double* acc) {
    __shared__ __attribute__((aligned(16))) uint4 wl[5 * 8 * 64];            // [fragment][k-step][lane]: one ds_read_b128 per A operand, conflict-free
    __shared__ __attribute__((aligned(16))) float tile[4][16 * HCT_PITCH];
    __shared__ float red[6][16];
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    for (int i = threadIdx.x; i < 5 * 8 * 64; i += 256) {   // A operand: row = output channel 16f + (l & 15), k = 32 ks + 8 (l >> 4) .. + 7
        const int l = i & 63, ks = (i >> 6) & 7, f = i >> 9;
        const int row = 16 * f + (l & 15);
        wl[i] = row < 69 ? *reinterpret_cast<const uint4*>(w + (size_t)row * 256 + 32 * ks + 8 * (l >> 4)) : make_uint4(0, 0, 0, 0);
    }
    f32x4_t bv[5];   // bias of the lane's result channels 16f + 4g + j
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = 16 * f + 4 * g + j; bv[f][j] = c < 69 ? bias[c] : 0.f; }
    __syncthreads();
    float* tw = tile[wave];
    const int q = lane & 15, sub = lane >> 4;
    const int nw = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    uint4 xq[8], xn[8];
    auto xload = [&](int grp, uint4 (&dst)[8]) {
        const long long m = (long long)grp * 16 + r16;
        const bf16_t* px = x + (size_t)m * x_cs + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) dst[ks] = *reinterpret_cast<const uint4*>(px + 32 * ks);
    };
    if (gw < ngroups) xload(gw, xq);
    for (int grp = gw; grp < ngroups; grp += nw) {
        if (grp + nw < ngroups) xload(grp + nw, xn);   // the next pixel group's operand under this group's MFMAs and tail
        // (r6c) the per-pixel inputs of the tail's four passes (2-D coordinates, extents, and in train mode the loss targets) are requested HERE, in
        // front of the MFMAs, for all four passes at once: inside the passes every one of them was a memory round trip of its own
        const int n = (int)(((long long)grp * 16) / HW);   // (a 16-pixel group never straddles two RoIs: HW % 16 == 0, checked by the host)
        const float ex[3] = {extents[n * 3 + 0], extents[n * 3 + 1], extents[n * 3 + 2]};
        float pc2x[4], pc2y[4], pmv[4], pmt[4], pgx[4][3];
        long long pgr[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const long long m = (long long)grp * 16 + 4 * s4 + sub;
            const int pix = (int)(m - (long long)n * HW);
            pc2x[s4] = coord2d[((size_t)n * 2 + 0) * HW + pix];
            pc2y[s4] = coord2d[((size_t)n * 2 + 1) * HW + pix];
            pmv[s4] = 0.f; pmt[s4] = 0.f; pgr[s4] = 0;
            pgx[s4][0] = pgx[s4][1] = pgx[s4][2] = 0.f;
            if constexpr (LOSS) {
                pmv[s4] = mvis[m];
                pmt[s4] = mtr[m];
                pgr[s4] = gt_region[m];
#pragma unroll
                for (int c = 0; c < 3; ++c) pgx[s4][c] = gt_xyz[((size_t)n * 3 + c) * HW + pix];
            }
        }
        f32x4_t acc[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) acc[f] = bv[f];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int f = 0; f < 5; ++f)
                acc[f] = GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, wl[(f * 8 + ks) * 64 + lane]), __builtin_bit_cast(bf16x8_t, xq[ks]), acc[f]);
        // D[i = 4g + j][col = r16] of fragment f = channel 16f + 4g + j of pixel r16 -> tile[pixel][channel]
#pragma unroll
        for (int f = 0; f < 5; ++f)
            *reinterpret_cast<float4*>(tw + r16 * HCT_PITCH + 16 * f + 4 * g) = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is this wave's own: its LDS operations complete in order
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {   // the tail, four pixels per pass (16 lanes each), as head_tail_fwd64_kernel
            const int pl = 4 * s4 + sub;
            const long long m = (long long)grp * 16 + pl;
            const float* h = tw + pl * HCT_PITCH;
            float r[4], h03[4];
            load4<float>(h + 4 + 4 * q, r);
            load4<float>(h, h03);
            const float h68 = h[68];
            const float c2x = pc2x[s4], c2y = pc2y[s4];
            const float mv = pmv[s4], mt = pmt[s4];
            const float gx[3] = {pgx[s4][0], pgx[s4][1], pgx[s4][2]};
            const long long gr = pgr[s4];
            if (head != nullptr) {   // the fp32 logits, 72 floats per pixel: 16 lanes x 16 bytes + two more
                float* ho = head + m * hs;
                store4<float>(ho + 4 * q, *reinterpret_cast<const float(*)[4]>(h + 4 * q));
                if (q < 2) store4<float>(ho + 64 + 4 * q, *reinterpret_cast<const float(*)[4]>(h + 64 + 4 * q));
            }
            const float r64 = (q == 15) ? h68 : -INFINITY;
            float mx = fmaxf(fmaxf(q == 0 ? -INFINITY : r[0], r[1]), fmaxf(fmaxf(r[2], r[3]), r64));
            mx = row16_max(mx);
            float e[4], e64;
            e[0] = (q == 0) ? 0.f : expf(r[0] - mx);
#pragma unroll
            for (int j = 1; j < 4; ++j) e[j] = expf(r[j] - mx);
            e64 = (q == 15) ? expf(r64 - mx) : 0.f;
            const float inv = 1.f / row16_sum(e[0] + e[1] + e[2] + e[3] + e64);
            T* o = pnp + m * pcs;
            float v[4] = {e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
            if (q == 0) v[0] = c2y;
            store4<T>(o + 4 + 4 * q, v);
            if (q == 15) {
                const float wv[4] = {e64 * inv, 0.f, 0.f, 0.f};
                store4<T>(o + 68, wv);
            } else if (q == 14) {
                const float wv[4] = {(h03[1] - 0.5f) * ex[0], (h03[2] - 0.5f) * ex[1], (h03[3] - 0.5f) * ex[2], c2x};
                store4<T>(o, wv);
            }
            if constexpr (LOSS) {   // cross entropy over the 65 classes of logits * visib + the masked L1 sums (head_tail_fwd64_kernel<T, true>)
                const int tgt = (int)(gr * (long long)mv);
                float z[4], zt = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    z[j] = r[j] * mv;
                    if (4 * q + j == tgt) zt = z[j];
                }
                const float z64 = (q == 15) ? r64 * mv : -INFINITY;
                if (q == 15 && tgt == 64) zt = z64;
                float mz = fmaxf(fmaxf(z[0], z[1]), fmaxf(fmaxf(z[2], z[3]), z64));
                mz = row16_max(mz);
                float es = expf(z[0] - mz) + expf(z[1] - mz) + expf(z[2] - mz) + expf(z[3] - mz) + (q == 15 ? expf(z64 - mz) : 0.f);
                es = row16_sum(es);
                zt = row16_sum(zt);
                if (q == 0) {
                    a[4] += (logf(es) + mz) - zt;
                    a[5] += mv;
                    a[3] += fabsf(h03[0] - mt);
#pragma unroll
                    for (int c = 0; c < 3; ++c) a[c] += fabsf(h03[1 + c] * mv - gx[c] * mv);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tail's reads are done before the next group overwrites the tile
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xq[ks] = xn[ks];
    }
    if constexpr (LOSS) {   // this workgroup's partial row [8 + blockIdx.x * 8 .. + 5]
        if (q == 0)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k][threadIdx.x >> 4] = a[k];
        __syncthreads();
        if (threadIdx.x < 6) {
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) v += (double)red[threadIdx.x][i];
            acc[8 + (size_t)blockIdx.x * 8 + threadIdx.x] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------ 1x1 output conv: data gradient + BatchNorm-backward sums
// The data gradient of the head's nn.Conv2d(256, 69, 1) -- d_act[px][256] = d_logits[px][69] . W -- with the ReLU mask and the two BatchNorm-backward
// sums of the BatchNorm(+ReLU) in front of the conv in its epilogue (what gdrn_conv_gemm's bnb_* epilogue does for this launch; autograd,
// core/gdrn_modeling/engine.py:279).  On the generic 128 x 128 tile this is the step's least efficient launch (K = 69 of a 128-deep reduction, 83 TFLOP/s,
// 111 us for 335 MB).  Here (r5): W^T in LDS in fragment order (16 fragments of 16 channels x 3 k-steps of 32 = 48 KB); a wave takes 16 pixels at a
// time -- B fragments (pixel r16, 8 logit gradients per k-group) straight from global memory, 48 MFMAs -- and turns the 16 x 256 fp32 result through a
// wave-private LDS tile, 64 channels at a time, into the layout in which the tensors stream: a lane owns 8 consecutive channels of one pixel = one
// 16-byte load of the BatchNorm's raw input and one 16-byte store, eight lanes a pixel's 128-byte line.  A lane meets the same 32 channels for every
// pixel it handles, so the BatchNorm-backward sums are 64 lane-local accumulators, reduced once at the end: one partial row [2][256] per workgroup.
constexpr int HOD_PITCH = 68;   // floats per pixel of the wave's 64-channel result tile
template <typename T>
__global__ __launch_bounds__(256, 2) void head_out_dgrad64_kernel(const bf16_t* __restrict__ dy, int dy_cs, const bf16_t* __restrict__ wd, int wd_cs,
                                                               const bf16_t* __restrict__ raw, int raw_cs, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, bf16_t* __restrict__ dx, int dx_cs,
                                                               float* __restrict__ rows, int ngroups) {
    __shared__ __attribute__((aligned(16))) uint4 wl[16 * 3 * 64];          // [fragment][k-step][lane]
    __shared__ __attribute__((aligned(16))) float tab[4][256];              // mean, 1/std, forward scale, forward shift per channel
    __shared__ __attribute__((aligned(16))) float tile[4][16 * HOD_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r16 = lane & 15;
    for (int i = threadIdx.x; i < 16 * 3 * 64; i += 256) {   // A operand: row = input channel 16f + (l & 15), k = logit 32 ks + 8 (l >> 4) .. + 7
        const int l = i & 63, ks = (i >> 6) % 3, f = i / 192;
        wl[i] = *reinterpret_cast<const uint4*>(wd + (size_t)(16 * f + (l & 15)) * wd_cs + 32 * ks + 8 * (l >> 4));
    }
    for (int c = threadIdx.x; c < 256; c += 256) { tab[0][c] = mean[c]; tab[1][c] = invstd[c]; tab[2][c] = scale[c]; tab[3][c] = shift[c]; }
    __syncthreads();
    float* tw = tile[wave];
    const int oct = lane & 7, psub = lane >> 3;   // streaming layout: channel octet of the 64-channel quarter, pixel 0..7 (+ 8 in the second half)
    float t1[4][8], t2[4][8];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int j = 0; j < 8; ++j) { t1[qd][j] = 0.f; t2[qd][j] = 0.f; }
    const int nw = gridDim.x * 4;
    for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += nw) {
        const long long m0 = (long long)grp * 16;
        uint4 yq[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) yq[ks] = *reinterpret_cast<const uint4*>(dy + (size_t)(m0 + r16) * dy_cs + 32 * ks + 8 * g);
        uint4 rq[4][2];   // the raw BatchNorm input of the lane's (pixel, octet) pairs: all eight loads of the group go out before the MFMAs
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                rq[qd][hf] = *reinterpret_cast<const uint4*>(raw + (size_t)(m0 + psub + 8 * hf) * raw_cs + 64 * qd + 8 * oct);
        f32x4_t acc[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int f = 0; f < 16; ++f)
                acc[f] = GDRN_MFMA16(__builtin_bit_cast(bf16x8_t, wl[(f * 3 + ks) * 64 + lane]), __builtin_bit_cast(bf16x8_t, yq[ks]), acc[f]);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {   // channels 64 qd .. 64 qd + 63
#pragma unroll
            for (int f4 = 0; f4 < 4; ++f4) {
                const f32x4_t v = acc[4 * qd + f4];
                *reinterpret_cast<float4*>(tw + r16 * HOD_PITCH + 16 * f4 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int c0 = 64 * qd + 8 * oct;
            float km[8], ki[8], kc[8], kh[8];
#pragma unroll
            for (int j4 = 0; j4 < 2; ++j4) {
                const float4 a0 = *reinterpret_cast<const float4*>(&tab[0][c0 + 4 * j4]), a1 = *reinterpret_cast<const float4*>(&tab[1][c0 + 4 * j4]);
                const float4 a2 = *reinterpret_cast<const float4*>(&tab[2][c0 + 4 * j4]), a3 = *reinterpret_cast<const float4*>(&tab[3][c0 + 4 * j4]);
                km[4 * j4] = a0.x; km[4 * j4 + 1] = a0.y; km[4 * j4 + 2] = a0.z; km[4 * j4 + 3] = a0.w;
                ki[4 * j4] = a1.x; ki[4 * j4 + 1] = a1.y; ki[4 * j4 + 2] = a1.z; ki[4 * j4 + 3] = a1.w;
                kc[4 * j4] = a2.x; kc[4 * j4 + 1] = a2.y; kc[4 * j4 + 2] = a2.z; kc[4 * j4 + 3] = a2.w;
                kh[4 * j4] = a3.x; kh[4 * j4 + 1] = a3.y; kh[4 * j4 + 2] = a3.z; kh[4 * j4 + 3] = a3.w;
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int pl = psub + 8 * hf;
                float gq[8];
                {
                    const float4 u0 = *reinterpret_cast<const float4*>(tw + pl * HOD_PITCH + 8 * oct), u1 = *reinterpret_cast<const float4*>(tw + pl * HOD_PITCH + 8 * oct + 4);
                    gq[0] = u0.x; gq[1] = u0.y; gq[2] = u0.z; gq[3] = u0.w; gq[4] = u1.x; gq[5] = u1.y; gq[6] = u1.z; gq[7] = u1.w;
                }
                float xv[8];
                Vec16<bf16_t>::unpack(rq[qd][hf], xv);
                float ov[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool keep = xv[j] * kc[j] + kh[j] > 0.f;
                    const float gv = keep ? gq[j] : 0.f;
                    ov[j] = gv;
                    t1[qd][j] += gv;
                    t2[qd][j] += gv * (xv[j] - km[j]) * ki[j];
                }
                *reinterpret_cast<uint4*>(dx + (size_t)(m0 + pl) * dx_cs + c0) = Vec16<bf16_t>::pack(ov);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile's reads are done before the next quarter overwrites it
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the lane's 32 channels (quarter qd, octet oct, j) summed over the eight lanes that share the octet (psub = lane >> 3), then over the four waves
    // in a fixed order: one partial row [2][256] per workgroup
    __syncthreads();   // (every wave is done with its tile; the tiles are reused as the reduction buffer: [wave][2][256] floats = 8 KB <= 4 x 4352 B)
    float* red = &tile[0][0];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float u1 = t1[qd][j], u2 = t2[qd][j];
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { u1 += __shfl_xor(u1, o, 64); u2 += __shfl_xor(u2, o, 64); }
            if (psub == 0) {
                red[(wave * 2 + 0) * 256 + 64 * qd + 8 * oct + j] = u1;
                red[(wave * 2 + 1) * 256 + 64 * qd + 8 * oct + j] = u2;
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256)
        rows[(size_t)blockIdx.x * 512 + i] = (red[i] + red[512 + i]) + (red[1024 + i] + red[1536 + i]);
}

// ------------------------------------------------------------------------------------------ map losses fwd
__global__ __launch_bounds__(256) void map_loss_fwd_kernel(const float* __restrict__ head, int hs,
                                                           const float* __restrict__ gt_xyz,
                                                           const float* __restrict__ mvis, const float* __restrict__ mtr,
                                                           const long long* __restrict__ gt_region, int N, int HW, int nreg,
                                                           double* acc, int acc_rows) {
    __shared__ float red[6][16];
    const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long long M = (long long)N * HW;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
    const int ncls = nreg + 1;
    float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // accumulators of lane q == 0 of each pixel group
    for (long long m = g0; m < M; m += ng) {
        const int n = (int)(m / HW), pix = (int)(m - (long long)n * HW);
        const float* h = head + m * hs;
        const float mv = mvis[m];
        // CE over nreg+1 classes of logits region*mv; lane q <-> classes q + 16 j
        const int tgt = (int)(gt_region[m] * (long long)mv);
        float z[5], mx = -INFINITY, zt = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = q + 16 * j;
            z[j] = (k < ncls) ? h[4 + k] * mv : -INFINITY;
            mx = fmaxf(mx, z[j]);
            if (k == tgt) zt = z[j];
        }
        mx = row16_max(mx);
        float e = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (q + 16 * j < ncls) e += expf(z[j] - mx);
        const float se = row16_sum(e);
        zt = row16_sum(zt);
        if (q == 0) {
            a[4] += (logf(se) + mx) - zt;
            a[5] += mv;
            a[3] += fabsf(h[0] - mtr[m]);
#pragma unroll
            for (int c = 0; c < 3; ++c) a[c] += fabsf(h[1 + c] * mv - gt_xyz[((size_t)n * 3 + c) * HW + pix] * mv);
        }
    }
    if (q == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k) red[k][grp] = a[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += (double)red[threadIdx.x][i];
        if (acc_rows) acc[8 + (size_t)blockIdx.x * 8 + threadIdx.x] = v;
        else unsafeAtomicAdd(&acc[threadIdx.x], v);
    }
}

// partial rows acc[8 + 8 r + k] (r < nrows, written by the loss kernels in GDRN_ACC_ROWS mode) -> totals acc[0..7] in a fixed order, then the
// five map losses.  1024 threads: thread t sums column t & 7 over rows (t >> 3) + 128 i with all (<= 32) loads of a batch in flight
// together (256 threads walking their rows one load at a time took 17 us for 4096 rows), 128 such sums per column are added in order.
__global__ __launch_bounds__(1024) void map_loss_finalize_rows_kernel(double* acc, int nrows, double npix, float* losses, const float* pose_rows, int N,
                                                                      const float* w, float* weighted) {
    __shared__ double part[128][8], part2[8][8];
    __shared__ double tot[8];
    const int k = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    double v = 0.0;
    for (int rb = 0; rb < nrows; rb += 128 * 32) {
        double x[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int r = rb + r0 + 128 * i;
            x[i] = (r < nrows) ? acc[8 + (size_t)r * 8 + k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) v += x[i];
    }
    part[r0][k] = (k < 6) ? v : 0.0;
    __syncthreads();
    if (threadIdx.x < 64) {   // column t & 7, eight lanes per column: 16 rows each, then a fixed-order 8-way add through the LDS
        const int c = threadIdx.x & 7, g = threadIdx.x >> 3;
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[g * 16 + i][c];
        part2[g][c] = t;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += part2[i][threadIdx.x];
        tot[threadIdx.x] = t;
        acc[threadIdx.x] = t;
    }
    __syncthreads();
    __shared__ float lv[8];
    if (threadIdx.x == 0) {
        const double den = tot[5] < 1.0 ? 1.0 : tot[5];
        lv[0] = (float)(tot[0] / den);
        lv[1] = (float)(tot[1] / den);
        lv[2] = (float)(tot[2] / den);
        lv[3] = (float)(tot[3] / npix);
        lv[4] = (float)(tot[4] / den);
#pragma unroll
        for (int k = 0; k < 5; ++k) losses[k] = lv[k];
    } else if (threadIdx.x >= 64 && threadIdx.x < 67) {
        // (gdrn_loss_finalize) the three pose losses from gdrn_pose_loss's per-RoI rows, in RoI order (fp32, as the atomics added them -- in a fixed order)
        const int k = threadIdx.x - 64;
        float v = 0.f;
        if (pose_rows != nullptr) {
            for (int n = 0; n < N; ++n) v += pose_rows[n * 4 + k];
            losses[5 + k] = v;
        } else {
            v = losses[5 + k];
        }
        lv[5 + k] = v;
    }
    __syncthreads();
    if (weighted != nullptr && threadIdx.x < 8) weighted[threadIdx.x] = lv[threadIdx.x] * w[threadIdx.x];
}

__global__ void map_loss_finalize_kernel(const double* acc, double npix, float* losses) {
    if (threadIdx.x == 0) {
        const double den = acc[5] < 1.0 ? 1.0 : acc[5];
        losses[0] = (float)(acc[0] / den);
        losses[1] = (float)(acc[1] / den);
        losses[2] = (float)(acc[2] / den);
        losses[3] = (float)(acc[3] / npix);
        losses[4] = (float)(acc[4] / den);
    }
}

// ------------------------------------------------------------------------------------------ head tail bwd
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

template <typename T>
__global__ __launch_bounds__(256) void head_tail_bwd_kernel(const float* __restrict__ head, int hs, const T* __restrict__ pnp,
                                                            const T* __restrict__ dpnp, int pcs,
                                                            const float* __restrict__ extents,
                                                            const float* __restrict__ gt_xyz, const float* __restrict__ mvis,
                                                            const float* __restrict__ mtr,
                                                            const long long* __restrict__ gt_region,
                                                            const double* __restrict__ acc, const float* __restrict__ gw,
                                                            T* __restrict__ dhead, int dcs, int N, int HW, int nreg) {
    const int q = threadIdx.x & 15;  // 16 lanes per pixel, lane q <-> classes q + 16 j
    const long long M = (long long)N * HW;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
    const float inv_den = (float)(1.0 / (acc[5] < 1.0 ? 1.0 : acc[5]));
    const float inv_np = (float)(1.0 / (double)M);
    const float gx[3] = {gw[0], gw[1], gw[2]};
    const float gmask = gw[3], gce = gw[4];
    const int ncls = nreg + 1;
    for (long long m = g0; m < M; m += ng) {
        const int n = (int)(m / HW), pix = (int)(m - (long long)n * HW);
        const float* h = head + m * hs;
        const float mv = mvis[m];
        const int tgt = (int)(gt_region[m] * (long long)mv);
        // CE gradient: mv * (softmax(region*mv)_k - onehot_k) / den
        float z[5], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = q + 16 * j;
            z[j] = (k < ncls) ? h[4 + k] * mv : -INFINITY;
            mx = fmaxf(mx, z[j]);
        }
        mx = row16_max(mx);
        float e[5], se = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            e[j] = (q + 16 * j < ncls) ? expf(z[j] - mx) : 0.f;
            se += e[j];
        }
        se = row16_sum(se);
        // attention softmax chain: region class k (>= 1) <- pnp channel 4 + k
        float sk[5], dk[5], dot = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = q + 16 * j;
            const bool on = dpnp != nullptr && k >= 1 && k <= nreg;
            sk[j] = on ? ld1<T>(pnp + m * pcs + 4 + k) : 0.f;
            dk[j] = on ? ld1<T>(dpnp + m * pcs + 4 + k) : 0.f;
            dot += sk[j] * dk[j];
        }
        dot = row16_sum(dot);
        T* o = dhead + m * dcs;
        const float cs = gce * mv * inv_den;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int k = q + 16 * j;
            if (k < ncls) st1<T>(o + 4 + k, cs * (e[j] / se - (tgt == k ? 1.f : 0.f)) + sk[j] * (dk[j] - dot));
        }
        if (q == 0) {
            st1<T>(o + 0, gmask * inv_np * sgn(h[0] - mtr[m]));
        } else if (q < 4) {
            const int c = q - 1;
            const float gt = gt_xyz[((size_t)n * 3 + c) * HW + pix];
            float d = gx[c] * mv * inv_den * sgn(h[1 + c] * mv - gt * mv);
            if (dpnp != nullptr) d += ld1<T>(dpnp + m * pcs + c) * extents[n * 3 + c];
            st1<T>(o + q, d);
        }
        for (int c = 4 + ncls + q; c < dcs; c += 16) st1<T>(o + c, 0.f);
    }
}

// ------------------------------------------------------------------------------------------ head tail bwd, nreg == 64
// same lane <-> class map as head_tail_fwd64_kernel: 16-byte logit loads, 4-channel loads of the softmax / its gradient, one
// 4-channel store per lane.  write_pad = 0: channels >= 72 of d_head are the caller's (zero-filled once).
template <typename T>
__global__ __launch_bounds__(256) void head_tail_bwd64_kernel(const float* __restrict__ head, int hs, const T* __restrict__ pnp,
                                                              const T* __restrict__ dpnp, int pcs, const float* __restrict__ extents,
                                                              const float* __restrict__ gt_xyz, const float* __restrict__ mvis,
                                                              const float* __restrict__ mtr, const long long* __restrict__ gt_region,
                                                              const double* __restrict__ acc, const float* __restrict__ gw,
                                                              T* __restrict__ dhead, int dcs, int N, int HW, int write_pad) {
    const int q = threadIdx.x & 15;
    const long long M = (long long)N * HW;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
    const float inv_den = (float)(1.0 / (acc[5] < 1.0 ? 1.0 : acc[5]));
    const float inv_np = (float)(1.0 / (double)M);
    const float gx[3] = {gw[0], gw[1], gw[2]};
    const float gmask = gw[3], gce = gw[4];
    for (long long m = g0; m < M; m += ng) {
        const int n = (int)(m / HW), pix = (int)(m - (long long)n * HW);
        const float* h = head + m * hs;
        // every load of the iteration up front, by all 16 lanes of the pixel (see head_tail_fwd64_kernel): the ones that sat in the
        // `q == 14` / `q == 15` branches behind the softmax cost a second memory round trip per iteration
        const float mv = mvis[m];
        const int tgt = (int)(gt_region[m] * (long long)mv);
        float r[4], h03[4];
        load4<float>(h + 4 + 4 * q, r);
        load4<float>(h, h03);                                  // mask, x, y, z
        const float h68 = h[68];
        const float mt = mtr[m];
        float gt3[3], ex[3], d03[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) { gt3[c] = gt_xyz[((size_t)n * 3 + c) * HW + pix]; ex[c] = extents[n * 3 + c]; }
        float sk[4] = {0.f, 0.f, 0.f, 0.f}, dk[4] = {0.f, 0.f, 0.f, 0.f}, s64 = 0.f, d64 = 0.f;
        if (dpnp != nullptr) {   // attention softmax chain: region class k (>= 1) <- pnp channel 4 + k
            load4<T>(pnp + m * pcs + 4 + 4 * q, sk);
            load4<T>(dpnp + m * pcs + 4 + 4 * q, dk);
            load4<T>(dpnp + m * pcs, d03);                     // d/d(x, y, z scaled by the extents), channel 3 unused
            s64 = ld1<T>(pnp + m * pcs + 68);
            d64 = ld1<T>(dpnp + m * pcs + 68);
            if (q == 0) { sk[0] = 0.f; dk[0] = 0.f; }   // channel 4 is roi_coord_2d, not a class
            if (q != 15) { s64 = 0.f; d64 = 0.f; }
        }
        // CE gradient: mv * (softmax(region*mv)_k - onehot_k) / den over the 65 classes
        float z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = r[j] * mv;
        const float z64 = (q == 15) ? h68 * mv : -INFINITY;
        float mz = fmaxf(fmaxf(z[0], z[1]), fmaxf(fmaxf(z[2], z[3]), z64));
        mz = row16_max(mz);
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = expf(z[j] - mz);
        const float e64 = (q == 15) ? expf(z64 - mz) : 0.f;
        const float inv = 1.f / row16_sum(e[0] + e[1] + e[2] + e[3] + e64);
        const float dot = row16_sum(sk[0] * dk[0] + sk[1] * dk[1] + sk[2] * dk[2] + sk[3] * dk[3] + s64 * d64);
        T* o = dhead + m * dcs;
        const float cs = gce * mv * inv_den;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = cs * (e[j] * inv - (tgt == 4 * q + j ? 1.f : 0.f)) + sk[j] * (dk[j] - dot);
        store4<T>(o + 4 + 4 * q, v);
        if (q == 15) {
            const float w[4] = {cs * (e64 * inv - (tgt == 64 ? 1.f : 0.f)) + s64 * (d64 - dot), 0.f, 0.f, 0.f};
            store4<T>(o + 68, w);
        } else if (q == 14) {
            float w[4];
            w[0] = gmask * inv_np * sgn(h03[0] - mt);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float d = gx[c] * mv * inv_den * sgn(h03[1 + c] * mv - gt3[c] * mv);
                if (dpnp != nullptr) d += d03[c] * ex[c];
                w[1 + c] = d;
            }
            store4<T>(o, w);
        }
        if (write_pad) {
            const float zz[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = 72 + 4 * q; c < dcs; c += 64) store4<T>(o + c, zz);
        }
    }
}

// ------------------------------------------------------------------------------------------ pose decode
// Forward-mode dual number for the nine inputs (rot6d[6], t_[3]) of the train-mode decode.  (r6b) One partial per LANE: lane k < 9 of the
// workgroup carries d/d(input k), every lane computes the value part redundantly -- the nine partials of a quantity used to be a 9-loop on one
// thread (20 us of serial arithmetic per RoI in a launch nothing overlaps with).  Same operations per partial, same results bit for bit.
struct D9 {
    float v;
    float d[1];
};
constexpr int DNP = 1;
__device__ __forceinline__ D9 dconst(float c) { D9 r; r.v = c; for (int i = 0; i < DNP; ++i) r.d[i] = 0.f; return r; }
__device__ __forceinline__ D9 dvar(float v, int k, int mine) { D9 r = dconst(v); r.d[0] = (k == mine) ? 1.f : 0.f; return r; }
__device__ __forceinline__ D9 operator+(const D9& a, const D9& b) { D9 r; r.v = a.v + b.v; for (int i = 0; i < DNP; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D9 operator-(const D9& a, const D9& b) { D9 r; r.v = a.v - b.v; for (int i = 0; i < DNP; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D9 operator*(const D9& a, const D9& b) { D9 r; r.v = a.v * b.v; for (int i = 0; i < DNP; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D9 operator*(const D9& a, float s) { D9 r; r.v = a.v * s; for (int i = 0; i < DNP; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ __forceinline__ D9 operator+(const D9& a, float s) { D9 r = a; r.v += s; return r; }
__device__ __forceinline__ D9 operator/(const D9& a, const D9& b) {
    D9 r; const float ib = 1.f / b.v; r.v = a.v * ib;
    for (int i = 0; i < DNP; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
__device__ __forceinline__ D9 dfun(const D9& a, float f, float df) { D9 r; r.v = f; for (int i = 0; i < DNP; ++i) r.d[i] = a.d[i] * df; return r; }
__device__ __forceinline__ D9 dsqrt(const D9& a) { const float s = sqrtf(a.v); return dfun(a, s, s > 0.f ? 0.5f / s : 0.f); }
__device__ __forceinline__ D9 dacos(const D9& a) { const float q = 1.f - a.v * a.v; return dfun(a, acosf(a.v), q > 0.f ? -1.f / sqrtf(q) : 0.f); }
__device__ __forceinline__ D9 dsin(const D9& a) { return dfun(a, sinf(a.v), cosf(a.v)); }
__device__ __forceinline__ D9 dcos(const D9& a) { return dfun(a, cosf(a.v), -sinf(a.v)); }

template <typename S>
__device__ __forceinline__ void cross3(const S* a, const S* b, S* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// F.normalize(v, p=2, dim=1, eps=1e-12): v / max(||v||, eps)
__device__ __forceinline__ void normalize3(const D9* v, D9* o) {
    D9 n = dsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n.v < 1e-12f) n = dconst(1e-12f);
    for (int i = 0; i < 3; ++i) o[i] = v[i] / n;
}

// train-mode decode (differentiable): R_ego[9] row-major, t[3]
__device__ void decode_train(const D9* in, const float* K, const float* ctr, const float* wh, float ratio, D9* R, D9* t) {
    const float eps = 1e-4f;
    D9 x[3], y[3], z[3], zr[3];
    normalize3(in, x);            // x = normalize(a1)
    cross3(x, in + 3, zr);        // z = normalize(x x a2)
    normalize3(zr, z);
    cross3(z, x, y);              // y = z x x ; R_allo = [x y z] as columns
    D9 cx = in[6] * wh[0] + ctr[0];
    D9 cy = in[7] * wh[1] + ctr[1];
    D9 zz = in[8] * ratio;
    t[2] = zz;
    // written as in the reference: z * (cx - px) / fx
    t[0] = (zz * (cx + (-K[2]))) / dconst(K[0]);
    t[1] = (zz * (cy + (-K[5]))) / dconst(K[4]);
    D9 nt = dsqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) + eps;
    D9 ray[3] = {t[0] / nt, t[1] / nt, t[2] / nt};
    D9 angle = dacos(ray[2]);
    // axis = cross((0,0,1), ray) = (-ray_y, ray_x, 0)
    D9 ax[3] = {dconst(0.f) - ray[1], ray[0], dconst(0.f)};
    D9 na = dsqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + eps;
    D9 half = angle * 0.5f;
    D9 sh = dsin(half);
    D9 q[4] = {dcos(half), (ax[0] / na) * sh, (ax[1] / na) * sh, (ax[2] / na) * sh};
    D9 nq = dsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    D9 qw = q[0] / nq, qx = q[1] / nq, qy = q[2] / nq, qz = q[3] / nq;
    D9 X = qx * 2.f, Y = qy * 2.f, Z = qz * 2.f;
    D9 wX = qw * X, wY = qw * Y, wZ = qw * Z, xX = qx * X, xY = qx * Y, xZ = qx * Z, yY = qy * Y, yZ = qy * Z, zZ = qz * Z;
    D9 one = dconst(1.f);
    D9 Q[9] = {one - (yY + zZ), xY - wZ, xZ + wY, xY + wZ, one - (xX + zZ), yZ - wX, xZ - wY, yZ + wX, one - (xX + yY)};
    // R_allo[r][c]: column 0 = x, 1 = y, 2 = z
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const D9* col = (c == 0) ? x : ((c == 1) ? y : z);
            R[r * 3 + c] = Q[r * 3 + 0] * col[0] + Q[r * 3 + 1] * col[1] + Q[r * 3 + 2] * col[2];
        }
}

// test-mode decode (reference: numpy float64 loop, no eps, `if angle > 0`)
__device__ void decode_test(const float* in, const float* K, const float* ctr, const float* wh, float ratio, float* R, float* t) {
    // rot6d -> R_allo in fp32 exactly as train mode (torch ops), then fp64 allo->ego
    float x[3], y[3], z[3], zr[3];
    float n = fmaxf(sqrtf(in[0] * in[0] + in[1] * in[1] + in[2] * in[2]), 1e-12f);
    for (int i = 0; i < 3; ++i) x[i] = in[i] / n;
    cross3(x, in + 3, zr);
    n = fmaxf(sqrtf(zr[0] * zr[0] + zr[1] * zr[1] + zr[2] * zr[2]), 1e-12f);
    for (int i = 0; i < 3; ++i) z[i] = zr[i] / n;
    cross3(z, x, y);
    const float cx = in[6] * wh[0] + ctr[0], cy = in[7] * wh[1] + ctr[1], zz = in[8] * ratio;
    t[0] = zz * (cx - K[2]) / K[0];
    t[1] = zz * (cy - K[5]) / K[4];
    t[2] = zz;
    const double tx = t[0], ty = t[1], tz = t[2];
    const float nt32 = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);  // np.linalg.norm on float32
    const float ray32[3] = {t[0] / nt32, t[1] / nt32, t[2] / nt32};
    (void)tx; (void)ty; (void)tz;
    double c = (double)ray32[2];
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double angle = acos(c);
    double Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (angle > 0.0) {
        double ax = -(double)ray32[1], ay = (double)ray32[0], az = 0.0;
        const double na = sqrt(ax * ax + ay * ay + az * az);
        ax /= na; ay /= na; az /= na;
        const double cs = cos(angle), sn = sin(angle), C = 1.0 - cs;
        Q[0] = ax * ax * C + cs;      Q[1] = ax * ay * C - az * sn; Q[2] = ax * az * C + ay * sn;
        Q[3] = ay * ax * C + az * sn; Q[4] = ay * ay * C + cs;      Q[5] = ay * az * C - ax * sn;
        Q[6] = az * ax * C - ay * sn; Q[7] = az * ay * C + ax * sn; Q[8] = az * az * C + cs;
    }
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) {
            const float* col = (cc == 0) ? x : ((cc == 1) ? y : z);
            R[r * 3 + cc] = (float)(Q[r * 3 + 0] * (double)col[0] + Q[r * 3 + 1] * (double)col[1] + Q[r * 3 + 2] * (double)col[2]);
        }
}

__device__ double rot_err_deg(const double* A, const double* B) {  // pose_error.re: arccos((tr(A B^T)-1)/2)
    double tr = 0.0;
    for (int i = 0; i < 9; ++i) tr += A[i] * B[i];
    if (tr > 3.0) tr = 3.0;
    double c = 0.5 * (tr - 1.0);
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    return acos(c) * 57.29577951308232;
}

__global__ __launch_bounds__(256) void pose_loss_kernel(const gdrn_pose_params p) {
    __shared__ float sR[9], sG[9], sT[3], sW;
    __shared__ float red[10][4];
    __shared__ float sS[10];
    const int n = blockIdx.x, tid = threadIdx.x;
    __shared__ float f2s[256], fcs[12], red9[9][4];
    const float* in = p.fc + (size_t)n * p.fs;
    if (p.fc2_ws != nullptr) {
        // (ABI 5) the tail of Patch-PnP's fully connected stack in this launch (conv_pnp_net.py:152-160): fc2's split-K slabs -> bias + LeakyReLU
        // (gdrn_linear_splitk's finish pass) -> fc_r | fc_t (a 9 x 256 product per RoI) -> p.fc; then the decode below.  One workgroup per RoI,
        // thread c = fc2 channel c.  The three launches this replaces (finish, a one-workgroup GEMM, this kernel) were 20 us of latency.
        float v = 0.f;
        for (int sp = 0; sp < p.fc2_splits; ++sp) v += p.fc2_ws[((size_t)sp * p.N + n) * 256 + tid];
        v += p.fc2_bias[tid];
        v = v > 0.f ? v : 0.1f * v;
        const bf16_t hv = f2bf(v);
        reinterpret_cast<bf16_t*>(p.f2_out)[(size_t)n * 256 + tid] = hv;
        const float a = bf2f(hv);
        const bf16_t* wr = reinterpret_cast<const bf16_t*>(p.w_rt);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float s_ = wave_sum(a * bf2f(wr[k * 256 + tid]));
            if ((tid & 63) == 0) red9[k][tid >> 6] = s_;
        }
        __syncthreads();
        if (tid < 9) {
            const float o = ((red9[tid][0] + red9[tid][1]) + (red9[tid][2] + red9[tid][3])) + p.b_rt[tid];
            fcs[tid] = o;
            p.fc_w[(size_t)n * p.fs + tid] = o;
        }
        __syncthreads();
        in = fcs;
        (void)f2s;
    }
    const float* K = p.cams + n * 9;
    const float* ctr = p.centers + n * 2;
    const float* wh = p.whs + n * 2;
    const float ratio = p.ratios[n];

    D9 Rd[9], td[3];
    if (p.train && tid < 9) {   // lane k: the value parts + the partial derivatives w.r.t. input k
        D9 v[9];
        for (int k = 0; k < 9; ++k) v[k] = dvar(in[k], k, tid);
        decode_train(v, K, ctr, wh, ratio, Rd, td);
    }
    if (tid == 0) {
        float Rf[9], tf[3];
        if (p.train) {
            for (int k = 0; k < 9; ++k) Rf[k] = Rd[k].v;
            for (int k = 0; k < 3; ++k) tf[k] = td[k].v;
        } else {
            decode_test(in, K, ctr, wh, ratio, Rf, tf);
        }
        for (int k = 0; k < 9; ++k) { sR[k] = Rf[k]; p.rot[n * 9 + k] = Rf[k]; }
        for (int k = 0; k < 3; ++k) { sT[k] = tf[k]; p.trans[n * 3 + k] = tf[k]; }
        if (p.gt_rot != nullptr) {
            // closest symmetric ground truth (pose_utils.py:430-482), errors in fp64
            double P[9], G0[9], best[9];
            for (int k = 0; k < 9; ++k) { P[k] = Rf[k]; G0[k] = p.gt_rot[n * 9 + k]; best[k] = G0[k]; }
            double best_err = rot_err_deg(P, G0);
            const double err0 = best_err;
            const int ns = (p.sym != nullptr && p.sym_count != nullptr) ? p.sym_count[n] : 0;
            for (int s = 0; s < ns; ++s) {
                const float* S = p.sym + ((size_t)n * p.Kmax + s) * 9;
                double C[9];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) C[r * 3 + c] = G0[r * 3 + 0] * S[0 * 3 + c] + G0[r * 3 + 1] * S[1 * 3 + c] + G0[r * 3 + 2] * S[2 * 3 + c];
                const double e = rot_err_deg(P, C);
                if (e < best_err) { best_err = e; for (int k = 0; k < 9; ++k) best[k] = C[k]; }
            }
            for (int k = 0; k < 9; ++k) sG[k] = (float)best[k];
            if (p.vis != nullptr) {
                p.vis[n * 2 + 0] = (float)err0;
                float te = 0.f;
                if (p.gt_trans != nullptr)
                    for (int k = 0; k < 3; ++k) { const float d = p.gt_trans[n * 3 + k] - tf[k]; te += d * d; }
                p.vis[n * 2 + 1] = sqrtf(te);
            }
        }
        if (p.extents != nullptr) sW = 1.f / fmaxf(fmaxf(p.extents[n * 3], p.extents[n * 3 + 1]), p.extents[n * 3 + 2]);
    }
    __syncthreads();
    if (!p.train || p.gt_rot == nullptr || p.points == nullptr) return;

    // point-matching loss: sum_p sum_a |w (R p - Rgt p)_a| and S[a][b] = sum_p sign(.)_a w p_b
    float acc[10];
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    const float w = sW;
    for (int i = tid; i < p.npts; i += 256) {
        const float* pt = p.points + ((size_t)n * p.npts + i) * 3;
        const float px = pt[0], py = pt[1], pz = pt[2];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float e = sR[a * 3] * px + sR[a * 3 + 1] * py + sR[a * 3 + 2] * pz;
            const float g = sG[a * 3] * px + sG[a * 3 + 1] * py + sG[a * 3 + 2] * pz;
            const float d = w * e - w * g;
            acc[0] += fabsf(d);
            const float s = sgn(d) * w;
            acc[1 + a * 3 + 0] += s * px;
            acc[1 + a * 3 + 1] += s * py;
            acc[1 + a * 3 + 2] += s * pz;
        }
    }
    for (int k = 0; k < 10; ++k) {
        const float v = wave_sum(acc[k]);
        if ((tid & 63) == 0) red[k][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 10) sS[tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    __syncthreads();
    if (tid == 0) {
        const float inv = 1.f / ((float)p.N * (float)p.npts);  // 3 * mean over (N, npts, 3)
        float lc = 0.f;
        const float* gtr = p.gt_trans_ratio + n * 3;
        lc = fabsf(in[6] - gtr[0]) + fabsf(in[7] - gtr[1]);
        if (p.loss_rows != nullptr) {   // (ABI 5) one row per RoI, added up in RoI order by gdrn_loss_finalize: no memset, no atomics, run-to-run identical
            p.loss_rows[n * 4 + 0] = sS[0] * inv;
            p.loss_rows[n * 4 + 1] = lc / (2.f * p.N);
            p.loss_rows[n * 4 + 2] = fabsf(in[8] - gtr[2]) / (float)p.N;
        } else {
            unsafeAtomicAdd(&p.losses[0], sS[0] * inv);
            unsafeAtomicAdd(&p.losses[1], lc / (2.f * p.N));
            unsafeAtomicAdd(&p.losses[2], fabsf(in[8] - gtr[2]) / (float)p.N);
        }
    }
    // dL/dfc: lane k < 9 holds d(R_ego)/d(input k) and forms its own entry; entries 9 .. fs - 1 are zero
    if (tid < p.fs) {
        const int k = tid;
        const float inv = 1.f / ((float)p.N * (float)p.npts);
        const float* gtr = p.gt_trans_ratio + n * 3;
        float g = 0.f;
        if (k < 9) {
            for (int ab = 0; ab < 9; ++ab) g += sS[1 + ab] * Rd[ab].d[0];
        }
        const float u1 = (k == 6) ? sgn(in[6] - gtr[0]) / (2.f * p.N) : ((k == 7) ? sgn(in[7] - gtr[1]) / (2.f * p.N) : 0.f);   // d loss_centroid / d fc[k]
        const float u2 = (k == 8) ? sgn(in[8] - gtr[2]) / (float)p.N : 0.f;                                                      // d loss_z / d fc[k]
        if (p.dfc_comb != nullptr) {
            // (ABI 5) dL/dfc directly -- the three unit gradients weighted with dL/dloss_k (p.gw: known before the forward pass in the fused train
            // step) and stored at the storage width: the combine + cast launches in front of the backward pass disappear
            bf16_t* dc = reinterpret_cast<bf16_t*>(p.dfc_comb) + (size_t)n * p.fs;
            float d = 0.f;
            if (k < 9) {
                d = p.gw[0] * (g * inv);   // (combine3's order: w0 * d0 + w1 * d1 + w2 * d2)
                if (k == 6 || k == 7) d += p.gw[1] * u1;
                if (k == 8) d += p.gw[2] * u2;
            }
            dc[k] = f2bf(d);
        } else if (p.dfc != nullptr) {
            p.dfc[((size_t)0 * p.N + n) * p.fs + k] = (k < 9) ? g * inv : 0.f;
            p.dfc[((size_t)1 * p.N + n) * p.fs + k] = u1;
            p.dfc[((size_t)2 * p.N + n) * p.fs + k] = u2;
        }
    }
}

__global__ void combine3_kernel(const float* in, const float* w, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = w[0] * in[i] + w[1] * in[n + i] + w[2] * in[2 * n + i];
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

namespace {
// fast path of the head-tail kernels: 64 regions, rows of whole 4-element groups (16-byte head rows, 8-byte bf16 rows)
inline bool ht64_ok(int nreg, int hs, int pcs) { return nreg == 64 && hs >= 72 && (hs & 3) == 0 && pcs >= 72 && (pcs & 3) == 0; }

template <bool LOSS>
int launch_ht_fwd(const float* head, int hs, const float* coord2d, const float* extents, void* pnp_in, int pcs, const float* gt_xyz,
                  const float* mv, const float* mt, const long long* greg, double* acc, int N, int HW, int nreg, int dtype, hipStream_t st) {
    const int dt = dtype & 0xff, write_pad = (dtype & GDRN_PREZEROED) ? 0 : 1, rows = (dtype & GDRN_ACC_ROWS) ? 1 : 0;
    const long long M = (long long)N * HW;
    const int blocks = (int)std::min<long long>((M + 15) / 16, 4096);
    if (ht64_ok(nreg, hs, pcs)) {
        if (dt == GDRN_DT_F32)
            GDRN_LAUNCH((head_tail_fwd64_kernel<float, LOSS>), dim3(blocks), dim3(256), 0, st, head, hs, coord2d, extents, (float*)pnp_in, pcs,
                               gt_xyz, mv, mt, greg, acc, N, HW, write_pad, rows);
        else
            GDRN_LAUNCH((head_tail_fwd64_kernel<bf16_t, LOSS>), dim3(blocks), dim3(256), 0, st, head, hs, coord2d, extents, (bf16_t*)pnp_in, pcs,
                               gt_xyz, mv, mt, greg, acc, N, HW, write_pad, rows);
    } else {
        if (dt == GDRN_DT_F32)
            GDRN_LAUNCH(head_tail_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, head, hs, coord2d, extents, (float*)pnp_in, pcs, N, HW, nreg);
        else
            GDRN_LAUNCH(head_tail_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, head, hs, coord2d, extents, (bf16_t*)pnp_in, pcs, N, HW, nreg);
        if (LOSS) {
            const int lb = (int)std::min<long long>((M + 15) / 16, 2048);
            GDRN_LAUNCH(map_loss_fwd_kernel, dim3(lb), dim3(256), 0, st, head, hs, gt_xyz, mv, mt, greg, N, HW, nreg, acc, rows);
        }
    }
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}
}  // namespace

// partial rows gdrn_head_tail_loss_fwd writes behind acc[0..7] in GDRN_ACC_ROWS mode (= the workgroups of the kernel that carries the sums)
extern "C" int gdrn_head_tail_loss_rows(int N, int HW, int nreg, int hs, int pcs) {
    if (N <= 0 || HW <= 0 || nreg < 1 || nreg > 64) return GDRN_ERR_ARG;
    const long long M = (long long)N * HW;
    return (int)std::min<long long>((M + 15) / 16, ht64_ok(nreg, hs, pcs) ? 4096 : 2048);
}

extern "C" int gdrn_head_tail_fwd(const float* head, int hs, const float* coord2d, const float* extents, void* pnp_in, int pcs,
                                  int N, int HW, int nreg, int dtype, void* stream) {
    const int dt = dtype & 0xff;
    if (!head || !coord2d || !extents || !pnp_in || N <= 0 || HW <= 0 || nreg < 1 || nreg > 64 || hs < nreg + 5 ||
        pcs < nreg + 5 || (dt != GDRN_DT_F32 && dt != GDRN_DT_H16))
        return GDRN_ERR_ARG;
    return launch_ht_fwd<false>(head, hs, coord2d, extents, pnp_in, pcs, nullptr, nullptr, nullptr, nullptr, nullptr, N, HW, nreg, dtype, ST);
}

// eval mode, nreg = 64: the head's 1x1 output conv (256 -> 69, weight rows [>= 69][256] in the 16-bit format, fp32 bias) + gdrn_head_tail_fwd in
// one launch; x: [N*HW][x_cs] 16-bit activations, head (nullable): fp32 logits [N*HW][hs >= 72], pnp_in: [N*HW][pcs >= 72] (pad channels the caller's)
extern "C" int gdrn_head_conv_tail_fwd(const void* x, int x_cs, const void* w, int w_rows, const float* bias, const float* coord2d,
                                       const float* extents, float* head, int hs, void* pnp_in, int pcs, int N, int HW, int nreg, int dtype,
                                       void* stream) {
    const int dt = dtype & 0xff;
    if (!x || !w || !bias || !coord2d || !extents || !pnp_in || N <= 0 || HW <= 0) return GDRN_ERR_ARG;
    if (dt != GDRN_DT_H16 || nreg != 64 || w_rows < 69 || x_cs < 256 || (x_cs & 7) || pcs < 72 || (pcs & 3) || (head && (hs < 72 || (hs & 3)))) return GDRN_ERR_SHAPE;
    const long long M = (long long)N * HW;
    if ((HW & 15) || M * std::max(x_cs, pcs) >= (1ll << 31)) return GDRN_ERR_SHAPE;   // (a 16-pixel group belongs to ONE RoI)
    const int ngroups = (int)(M / 16);
    const int blocks = (int)std::min<long long>((ngroups + 3) / 4, 1024);   // two 62 KB workgroups per CU, two rounds
    GDRN_LAUNCH((head_conv_tail64_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, ST, reinterpret_cast<const bf16_t*>(x), x_cs, reinterpret_cast<const bf16_t*>(w), bias,
                coord2d, extents, head, hs, reinterpret_cast<bf16_t*>(pnp_in), pcs, N, HW, ngroups, nullptr, nullptr, nullptr, nullptr, nullptr);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// data gradient of the 1x1 output conv with the BatchNorm(+ReLU)-backward mask and sums of the layer in front of it: dy = d_logits [N*HW][dy_cs >= 96]
// (16-bit; channels >= 69 zero), wd = the data-gradient operand rows [256][wd_cs >= 96] (row ci = W[:, ci], columns >= 69 zero), raw = that BatchNorm's raw
// input [N*HW][raw_cs], mean / invstd / scale / shift its statistics and forward affine (mask = scale * raw + shift > 0), dx = the masked gradient
// [N*HW][dx_cs], rows = [gdrn_head_out_dgrad_rows(N, HW)][2][256] partial sums for gdrn_bn_bwd_coef
extern "C" int gdrn_head_out_dgrad_rows(int N, int HW) {
    const long long M = (long long)N * HW;
    if (N <= 0 || HW <= 0 || (M & 15)) return GDRN_ERR_ARG;
    return (int)std::min<long long>((M / 16 + 3) / 4, 512);
}

extern "C" int gdrn_head_out_dgrad(const void* dy, int dy_cs, const void* wd, int wd_cs, const void* raw, int raw_cs, const float* mean,
                                   const float* invstd, const float* scale, const float* shift, void* dx, int dx_cs, float* rows, int N, int HW,
                                   int dtype, void* stream) {
    const int dt = dtype & 0xff;
    if (!dy || !wd || !raw || !mean || !invstd || !scale || !shift || !dx || !rows || N <= 0 || HW <= 0) return GDRN_ERR_ARG;
    if (dt != GDRN_DT_H16 || dy_cs < 96 || wd_cs < 96 || raw_cs < 256 || dx_cs < 256 || ((dy_cs | wd_cs | raw_cs | dx_cs) & 7)) return GDRN_ERR_SHAPE;
    const long long M = (long long)N * HW;
    if ((M & 15) || M * std::max(std::max(dy_cs, raw_cs), dx_cs) >= (1ll << 31)) return GDRN_ERR_SHAPE;
    const int ngroups = (int)(M / 16);
    GDRN_LAUNCH(head_out_dgrad64_kernel<bf16_t>, dim3(gdrn_head_out_dgrad_rows(N, HW)), dim3(256), 0, ST, reinterpret_cast<const bf16_t*>(dy), dy_cs,
                reinterpret_cast<const bf16_t*>(wd), wd_cs, reinterpret_cast<const bf16_t*>(raw), raw_cs, mean, invstd, scale, shift,
                reinterpret_cast<bf16_t*>(dx), dx_cs, rows, ngroups);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// train mode: the same + the map-loss sums of gdrn_head_tail_loss_fwd as per-workgroup partial rows: acc = 8 + 8 * gdrn_head_conv_tail_loss_rows(N, HW)
// doubles, finished by gdrn_map_loss_finalize_rows; head (the fp32 logits) is mandatory -- gdrn_head_tail_bwd reads it
extern "C" int gdrn_head_conv_tail_loss_rows(int N, int HW) {
    const long long M = (long long)N * HW;
    if (N <= 0 || HW <= 0 || (M & 15)) return GDRN_ERR_ARG;
    return (int)std::min<long long>((M / 16 + 3) / 4, 1024);
}

extern "C" int gdrn_head_conv_tail_loss_fwd(const void* x, int x_cs, const void* w, int w_rows, const float* bias, const float* coord2d,
                                            const float* extents, float* head, int hs, void* pnp_in, int pcs, const float* gt_xyz,
                                            const float* mask_visib, const float* mask_trunc, const long long* gt_region, double* acc, int N, int HW,
                                            int nreg, int dtype, void* stream) {
    const int dt = dtype & 0xff;
    if (!x || !w || !bias || !coord2d || !extents || !head || !pnp_in || !gt_xyz || !mask_visib || !mask_trunc || !gt_region || !acc || N <= 0 || HW <= 0)
        return GDRN_ERR_ARG;
    if (dt != GDRN_DT_H16 || nreg != 64 || w_rows < 69 || x_cs < 256 || (x_cs & 7) || pcs < 72 || (pcs & 3) || hs < 72 || (hs & 3)) return GDRN_ERR_SHAPE;
    const long long M = (long long)N * HW;
    if ((HW & 15) || M * std::max(x_cs, pcs) >= (1ll << 31)) return GDRN_ERR_SHAPE;   // (a 16-pixel group belongs to ONE RoI)
    const int ngroups = (int)(M / 16);
    const int blocks = gdrn_head_conv_tail_loss_rows(N, HW);
    GDRN_LAUNCH((head_conv_tail64_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, ST, reinterpret_cast<const bf16_t*>(x), x_cs, reinterpret_cast<const bf16_t*>(w), bias,
                coord2d, extents, head, hs, reinterpret_cast<bf16_t*>(pnp_in), pcs, N, HW, ngroups, gt_xyz, mask_visib, mask_trunc, gt_region, acc);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_head_tail_loss_fwd(const float* head, int hs, const float* coord2d, const float* extents, void* pnp_in, int pcs,
                                       const float* gt_xyz, const float* mask_visib, const float* mask_trunc, const long long* gt_region,
                                       double* acc, int N, int HW, int nreg, int dtype, void* stream) {
    const int dt = dtype & 0xff;
    if (!head || !coord2d || !extents || !pnp_in || !gt_xyz || !mask_visib || !mask_trunc || !gt_region || !acc || N <= 0 || HW <= 0 ||
        nreg < 1 || nreg > 64 || hs < nreg + 5 || pcs < nreg + 5 || (dt != GDRN_DT_F32 && dt != GDRN_DT_H16))
        return GDRN_ERR_ARG;
    if (!(dtype & GDRN_ACC_ROWS) && hipMemsetAsync(acc, 0, 8 * sizeof(double), ST) != hipSuccess) return GDRN_ERR_LAUNCH;
    return launch_ht_fwd<true>(head, hs, coord2d, extents, pnp_in, pcs, gt_xyz, mask_visib, mask_trunc, gt_region, acc, N, HW, nreg, dtype, ST);
}

extern "C" int gdrn_map_loss_fwd(const float* head, int hs, const float* gt_xyz, const float* mask_visib, const float* mask_trunc,
                                 const long long* gt_region, int N, int HW, int nreg, double* acc, void* stream) {
    if (!head || !gt_xyz || !mask_visib || !mask_trunc || !gt_region || !acc || N <= 0 || HW <= 0 || nreg < 1 || nreg > 64)
        return GDRN_ERR_ARG;
    if (hipMemsetAsync(acc, 0, 8 * sizeof(double), ST) != hipSuccess) return GDRN_ERR_LAUNCH;
    const long long M = (long long)N * HW;
    const int blocks = (int)std::min<long long>((M + 15) / 16, 2048);
    GDRN_LAUNCH(map_loss_fwd_kernel, dim3(blocks), dim3(256), 0, ST, head, hs, gt_xyz, mask_visib, mask_trunc, gt_region, N, HW, nreg, acc, 0);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_map_loss_finalize(const double* acc, int N, int HW, float* losses, void* stream) {
    if (!acc || !losses) return GDRN_ERR_ARG;
    GDRN_LAUNCH(map_loss_finalize_kernel, dim3(1), dim3(64), 0, ST, acc, (double)N * (double)HW, losses);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_map_loss_finalize_rows(double* acc, int nrows, int N, int HW, float* losses, void* stream) {
    if (!acc || !losses || nrows <= 0 || N <= 0 || HW <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(map_loss_finalize_rows_kernel, dim3(1), dim3(1024), 0, ST, acc, nrows, (double)N * (double)HW, losses, (const float*)nullptr, N,
                (const float*)nullptr, (float*)nullptr);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// (ABI 5) ... the same launch also finishes the pose losses from gdrn_pose_loss's per-RoI rows (pose_rows [N][4], nullable: losses[5..7] are then
// already final) and writes weighted[k] = losses[k] * w[k], k < 8 (both nullable): the loss vector the train step returns, without a launch of its own
extern "C" int gdrn_loss_finalize(double* acc, int nrows, int N, int HW, const float* pose_rows, float* losses, const float* w, float* weighted,
                                  void* stream) {
    if (!acc || !losses || nrows <= 0 || N <= 0 || HW <= 0 || ((w != nullptr) != (weighted != nullptr))) return GDRN_ERR_ARG;
    GDRN_LAUNCH(map_loss_finalize_rows_kernel, dim3(1), dim3(1024), 0, ST, acc, nrows, (double)N * (double)HW, losses, pose_rows, N, w, weighted);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_head_tail_bwd(const float* head, int hs, const void* pnp_in, const void* d_pnp_in, int pcs,
                                  const float* extents, const float* gt_xyz, const float* mask_visib, const float* mask_trunc,
                                  const long long* gt_region, const double* acc, const float* gw, void* d_head, int dcs, int N,
                                  int HW, int nreg, int dtype, void* stream) {
    const int dt = dtype & 0xff, write_pad = (dtype & GDRN_PREZEROED) ? 0 : 1;
    if (!head || !extents || !gt_xyz || !mask_visib || !mask_trunc || !gt_region || !acc || !gw || !d_head || N <= 0 ||
        HW <= 0 || nreg < 1 || nreg > 64 || dcs < nreg + 5 || (dt != GDRN_DT_F32 && dt != GDRN_DT_H16))
        return GDRN_ERR_ARG;
    if (d_pnp_in != nullptr && pnp_in == nullptr) return GDRN_ERR_ARG;
    const long long M = (long long)N * HW;
    const int blocks = (int)std::min<long long>((M + 15) / 16, 4096);
    const bool fast = ht64_ok(nreg, hs, dcs) && (d_pnp_in == nullptr || ht64_ok(nreg, hs, pcs));
    if (dt == GDRN_DT_F32) {
        if (fast)
            GDRN_LAUNCH(head_tail_bwd64_kernel<float>, dim3(blocks), dim3(256), 0, ST, head, hs, (const float*)pnp_in, (const float*)d_pnp_in, pcs,
                               extents, gt_xyz, mask_visib, mask_trunc, gt_region, acc, gw, (float*)d_head, dcs, N, HW, write_pad);
        else
            GDRN_LAUNCH(head_tail_bwd_kernel<float>, dim3(blocks), dim3(256), 0, ST, head, hs, (const float*)pnp_in,
                               (const float*)d_pnp_in, pcs, extents, gt_xyz, mask_visib, mask_trunc, gt_region, acc, gw,
                               (float*)d_head, dcs, N, HW, nreg);
    } else {
        if (fast)
            GDRN_LAUNCH(head_tail_bwd64_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, head, hs, (const bf16_t*)pnp_in, (const bf16_t*)d_pnp_in,
                               pcs, extents, gt_xyz, mask_visib, mask_trunc, gt_region, acc, gw, (bf16_t*)d_head, dcs, N, HW, write_pad);
        else
            GDRN_LAUNCH(head_tail_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, ST, head, hs, (const bf16_t*)pnp_in,
                               (const bf16_t*)d_pnp_in, pcs, extents, gt_xyz, mask_visib, mask_trunc, gt_region, acc, gw,
                               (bf16_t*)d_head, dcs, N, HW, nreg);
    }
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_pose_loss(const gdrn_pose_params* p, void* stream) {
    if (!p || !p->fc || !p->cams || !p->centers || !p->whs || !p->ratios || !p->rot || !p->trans || p->N <= 0 || p->fs < 9 || p->fs > 256)
        return GDRN_ERR_ARG;   // (fs <= 256: one thread per entry of a dL/dfc row)
    if (p->train && (!p->losses || !p->gt_rot || !p->gt_trans_ratio || !p->points || !p->extents || p->npts <= 0))
        return GDRN_ERR_ARG;
    if (p->fc2_ws && (!p->fc2_bias || !p->f2_out || !p->w_rt || !p->b_rt || !p->fc_w || p->fc2_splits <= 0)) return GDRN_ERR_ARG;
    if (p->dfc_comb && (!p->gw || !p->train)) return GDRN_ERR_ARG;
    if (p->losses && !p->loss_rows && hipMemsetAsync(p->losses, 0, 3 * sizeof(float), ST) != hipSuccess) return GDRN_ERR_LAUNCH;
    GDRN_LAUNCH(pose_loss_kernel, dim3(p->N), dim3(256), 0, ST, *p);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_combine3(const float* in, const float* w, float* out, int n, void* stream) {
    if (!in || !w || !out || n <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(combine3_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, in, w, out, n);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

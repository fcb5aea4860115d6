// On-device inference post-processing of the GDR-Net RoI path for gfx950: what the
// reference's evaluator does on the host per instance between the network and cv2's PnP-RANSAC --
//   get_out_coor  (core/gdrn_modeling/engine_utils.py:92-105, L1 branch: concatenate the three coordinate maps),
//   get_out_mask  (engine_utils.py:108-126, L1 branch: per-RoI min-max normalisation, no epsilon),
//   GDRN_Evaluator.get_img_model_points_with_coords2d (gdrn_evaluator.py:89-126: extent de-normalisation, image-size
//   scaling of the 2D coordinates, mask / near-zero selection, row-major compaction of the selected pixels)
// -- for the whole batch in one launch, without the device->host copy of the dense maps and the Python loop.
// fp32 arithmetic in the reference's operation order: results are bit-identical to the numpy / torch-CPU path.
#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

__global__ __launch_bounds__(256) void correspondences_kernel(const float* __restrict__ mask, const float* __restrict__ cx,
                                                              const float* __restrict__ cy, const float* __restrict__ cz, long long ns,
                                                              int ps, const float* __restrict__ coord2d,
                                                              const float* __restrict__ extents, const float* __restrict__ im_hw,
                                                              float mask_thr, int HW, float* __restrict__ out_mask,
                                                              float* __restrict__ out_xyz, float* __restrict__ img_pts,
                                                              float* __restrict__ model_pts, int* __restrict__ counts) {
    __shared__ float red[2][4];
    __shared__ int scan[256];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* mk = mask + (size_t)n * ns;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < HW; i += 256) {
        const float v = mk[(size_t)i * ps];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    mn = -wave_max(-mn);
    if (lane == 0) { red[0][wave] = mn; red[1][wave] = mx; }
    __syncthreads();
    mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    const float den = mx - mn;
    const float e0 = extents[n * 3 + 0], e1 = extents[n * 3 + 1], e2 = extents[n * 3 + 2];
    const float imH = im_hw[n * 2 + 0], imW = im_hw[n * 2 + 1];
    const float t0 = 0.0001f * e0, t1 = 0.0001f * e1, t2 = 0.0001f * e2;
    // each thread owns a run of consecutive pixels so that the compaction keeps the reference's row-major order
    const int per = (HW + 255) / 256;
    const int i0 = tid * per, i1 = min(HW, i0 + per);
    const float* px = cx + (size_t)n * ns;
    const float* py = cy + (size_t)n * ns;
    const float* pz = cz + (size_t)n * ns;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) {
        const float m = (mk[(size_t)i * ps] - mn) / den;
        const float x = px[(size_t)i * ps], y = py[(size_t)i * ps], z = pz[(size_t)i * ps];
        if (out_mask) out_mask[(size_t)n * HW + i] = m;
        if (out_xyz) {
            out_xyz[((size_t)n * 3 + 0) * HW + i] = x;
            out_xyz[((size_t)n * 3 + 1) * HW + i] = y;
            out_xyz[((size_t)n * 3 + 2) * HW + i] = z;
        }
        const float xd = (x - 0.5f) * e0, yd = (y - 0.5f) * e1, zd = (z - 0.5f) * e2;
        cnt += (m > mask_thr && fabsf(xd) > t0 && fabsf(yd) > t1 && fabsf(zd) > t2) ? 1 : 0;
    }
    scan[tid] = cnt;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan
        const int v = (tid >= o) ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    int pos = scan[tid] - cnt;
    if (tid == 255 && counts) counts[n] = scan[255];
    if (img_pts == nullptr || model_pts == nullptr) return;
    const float* c2x = coord2d + ((size_t)n * 2 + 0) * HW;
    const float* c2y = coord2d + ((size_t)n * 2 + 1) * HW;
    for (int i = i0; i < i1; ++i) {
        const float m = (mk[(size_t)i * ps] - mn) / den;
        const float xd = (px[(size_t)i * ps] - 0.5f) * e0, yd = (py[(size_t)i * ps] - 0.5f) * e1, zd = (pz[(size_t)i * ps] - 0.5f) * e2;
        if (m > mask_thr && fabsf(xd) > t0 && fabsf(yd) > t1 && fabsf(zd) > t2) {
            float* ip = img_pts + ((size_t)n * HW + pos) * 2;
            float* mp = model_pts + ((size_t)n * HW + pos) * 3;
            ip[0] = c2x[i] * imW;
            ip[1] = c2y[i] * imH;
            mp[0] = xd;
            mp[1] = yd;
            mp[2] = zd;
            ++pos;
        }
    }
}

}  // namespace

extern "C" int gdrn_correspondences(const float* mask, const float* coor_x, const float* coor_y, const float* coor_z, long long roi_stride,
                                    int pix_stride, const float* coord2d, const float* extents, const float* im_hw, float mask_thr, int N,
                                    int HW, float* out_mask, float* out_xyz, float* img_pts, float* model_pts, int* counts, void* stream) {
    if (!mask || !coor_x || !coor_y || !coor_z || !extents || !im_hw || N <= 0 || HW <= 0 || pix_stride <= 0) return GDRN_ERR_ARG;
    if ((img_pts != nullptr) != (model_pts != nullptr)) return GDRN_ERR_ARG;
    if (img_pts && !coord2d) return GDRN_ERR_ARG;
    GDRN_LAUNCH(correspondences_kernel, dim3(N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mask, coor_x, coor_y, coor_z,
                       roi_stride, pix_stride, coord2d, extents, im_hw, mask_thr, HW, out_mask, out_xyz, img_pts, model_pts, counts);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// GPU RoI cropper / target builder for gfx950: the step immediately before the per-RoI hot
// path.  Restates, for a whole batch of RoIs per launch and with the frames already resident in HBM, what the reference's
// data loader does per instance on the host with cv2 + numpy:
//   crop_resize_by_warp_affine / get_affine_transform        core/utils/data_utils.py:80-137
//   roi_img (bilinear u8, normalize_image) + roi_coord_2d     core/gdrn_modeling/data_loader.py:425-439, 487-498;
//                                                              core/base_data_loader.py:114-118
//   roi_mask_{trunc,visib,obj}, roi_xyz (nearest), xyz / extent + 0.5, xyz_to_region, roi_wh / resize_ratio / trans_ratio
//                                                              data_loader.py:460-545, 617-632; data_utils.py:213-219
// cv2.getAffineTransform / cv2.warpAffine are followed operation by operation (OpenCV 4.x imgwarp.cpp: double LU solve,
// double inversion, AB_BITS = 10 fixed-point walk, 1/32-pixel bilinear table, 15-bit fixed-point weights for u8, fp32
// left-to-right accumulation for float images, BORDER_CONSTANT 0), so integer / index / u8 results are bit-identical to
// the CPU path and fp32 results are too as long as no FMA contraction happens -- hence the pragma below.
// These kernels are HBM / gather bound (a few hundred KB per RoI); there is nothing here for the matrix cores.
#include "common.h"
#include "../../include/gdrn_hip.h"

#pragma clang fp contract(off)

namespace {

constexpr int AB_BITS = 10;
constexpr int INTER_BITS = 5;
constexpr int INTER_TAB = 1 << INTER_BITS;

// cv2.getAffineTransform(src, dst): m = [a b c; d e f] with (u, v) = m * (x, y, 1) through the three point pairs.
__device__ void solve_affine(const float (*src)[2], const float (*dst)[2], double* M) {
    double A[6][6], b[6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) A[i][j] = 0.0;
    for (int i = 0; i < 3; i++) {
        A[i][0] = A[i + 3][3] = (double)src[i][0];
        A[i][1] = A[i + 3][4] = (double)src[i][1];
        A[i][2] = A[i + 3][5] = 1.0;
        b[i] = (double)dst[i][0];
        b[i + 3] = (double)dst[i][1];
    }
    for (int i = 0; i < 6; i++) {
        int k = i;
        for (int j = i + 1; j < 6; j++)
            if (fabs(A[j][i]) > fabs(A[k][i])) k = j;
        if (fabs(A[k][i]) < 2.220446049250313e-16 * 100) {
            for (int j = 0; j < 6; j++) M[j] = 0.0;
            return;
        }
        if (k != i) {
            for (int j = i; j < 6; j++) { const double t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t; }
            const double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        const double d = -1.0 / A[i][i];
        for (int j = i + 1; j < 6; j++) {
            const double alpha = A[j][i] * d;
            for (int kk = i + 1; kk < 6; kk++) A[j][kk] = A[j][kk] + alpha * A[i][kk];
            b[j] = b[j] + alpha * b[i];
        }
    }
    for (int i = 5; i >= 0; i--) {
        double s = b[i];
        for (int kk = i + 1; kk < 6; kk++) s = s - A[i][kk] * b[kk];
        b[i] = s / A[i][i];
    }
    for (int j = 0; j < 6; j++) M[j] = b[j];
}

// get_affine_transform(center, scale, rot = 0, res) followed by warpAffine's inversion: dst pixel -> src position.
__device__ void crop_inverse_map(double cx, double cy, double scale, int res, double* Mi) {
    const float sw = (float)scale;
    const float half = sw * -0.5f;
    float src[3][2], dst[3][2];
    src[0][0] = (float)cx;
    src[0][1] = (float)cy;
    src[1][0] = (float)(cx + 0.0);
    src[1][1] = (float)(cy + (double)half);
    const float dc = (float)(res * 0.5);
    dst[0][0] = dc;
    dst[0][1] = dc;
    dst[1][0] = dc + 0.0f;
    dst[1][1] = dc + (float)(res * -0.5);
    for (int s = 0; s < 2; s++) {
        float(*p)[2] = s ? dst : src;
        const float d0 = p[0][0] - p[1][0], d1 = p[0][1] - p[1][1];
        p[2][0] = p[1][0] + (-d1);
        p[2][1] = p[1][1] + d0;
    }
    double M[6];
    solve_affine(src, dst, M);
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    const double m0 = A11, m1 = M[1] * (-D), m3 = M[3] * (-D), m4 = A22;
    Mi[0] = m0;
    Mi[1] = m1;
    Mi[2] = -m0 * M[2] - m1 * M[5];
    Mi[3] = m3;
    Mi[4] = m4;
    Mi[5] = -m3 * M[2] - m4 * M[5];
}

__global__ __launch_bounds__(64) void roi_affine_kernel(const gdrn_roi_task* __restrict__ tasks, int B, int in_res, int out_res,
                                                        double* __restrict__ minv, float* __restrict__ roi_wh,
                                                        float* __restrict__ resize_ratio, float* __restrict__ trans_ratio) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= B * 2) return;
    const int n = i >> 1, which = i & 1;
    const gdrn_roi_task t = tasks[n];
    crop_inverse_map(t.cx, t.cy, t.scale, which ? out_res : in_res, minv + (size_t)i * 6);
    if (which == 0) {
        const double rr = (double)out_res / t.scale;
        if (roi_wh) { roi_wh[n * 2] = (float)t.bw; roi_wh[n * 2 + 1] = (float)t.bh; }
        if (resize_ratio) resize_ratio[n] = (float)rr;
        if (trans_ratio) {
            trans_ratio[n * 3 + 0] = (float)((t.ox - t.cx) / t.bw);
            trans_ratio[n * 3 + 1] = (float)((t.oy - t.cy) / t.bh);
            trans_ratio[n * 3 + 2] = (float)(t.tz / rr);
        }
    }
}

__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int cv_round(double v) { return (int)rint(v); }

// warpAffine's fixed-point walk for destination pixel (x, y); nearest: integer source pixel.
__device__ __forceinline__ void coords_nearest(const double* M, int x, int y, int& sx, int& sy) {
    const int X0 = cv_round((M[1] * y + M[2]) * 1024.0) + 512, Y0 = cv_round((M[4] * y + M[5]) * 1024.0) + 512;
    const int X = X0 + cv_round(M[0] * x * 1024.0), Y = Y0 + cv_round(M[3] * x * 1024.0);
    sx = sat_short(X >> AB_BITS);
    sy = sat_short(Y >> AB_BITS);
}
__device__ __forceinline__ void coords_linear(const double* M, int x, int y, int& sx, int& sy, int& fx, int& fy) {
    const int X0 = cv_round((M[1] * y + M[2]) * 1024.0) + 16, Y0 = cv_round((M[4] * y + M[5]) * 1024.0) + 16;
    const int X = (X0 + cv_round(M[0] * x * 1024.0)) >> (AB_BITS - INTER_BITS), Y = (Y0 + cv_round(M[3] * x * 1024.0)) >> (AB_BITS - INTER_BITS);
    sx = sat_short(X >> INTER_BITS);
    sy = sat_short(Y >> INTER_BITS);
    fx = X & (INTER_TAB - 1);
    fy = Y & (INTER_TAB - 1);
}

// roi_img: bilinear u8 [H][W][3] -> normalised fp32 [B][3][res][res]
__global__ __launch_bounds__(256) void roi_crop_image_kernel(const gdrn_roi_task* __restrict__ tasks, const double* __restrict__ minv, int res,
                                                             double mean0, double mean1, double mean2, double std0, double std1, double std2,
                                                             float* __restrict__ out) {
    __shared__ double M[6];
    __shared__ const uint8_t* img_s;
    __shared__ int HW[2];
    const int n = blockIdx.y;
    if (threadIdx.x < 6) M[threadIdx.x] = minv[((size_t)n * 2 + 0) * 6 + threadIdx.x];
    if (threadIdx.x == 6) { img_s = tasks[n].image; HW[0] = tasks[n].H; HW[1] = tasks[n].W; }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= res * res) return;
    const int y = p / res, x = p - y * res;
    const uint8_t* img = img_s;
    const int H = HW[0], W = HW[1];
    int sx, sy, fx, fy;
    coords_linear(M, x, y, sx, sy, fx, fy);
    int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }  // short saturation of 32768 and OpenCV's sum fix-up
    const bool x0 = (unsigned)sx < (unsigned)W, x1 = (unsigned)(sx + 1) < (unsigned)W;
    const bool y0 = (unsigned)sy < (unsigned)H, y1 = (unsigned)(sy + 1) < (unsigned)H;
    const int cx0 = min(max(sx, 0), W - 1), cx1 = min(max(sx + 1, 0), W - 1), cy0 = min(max(sy, 0), H - 1), cy1 = min(max(sy + 1, 0), H - 1);
    const uint8_t* r0 = img + (size_t)cy0 * W * 3;
    const uint8_t* r1 = img + (size_t)cy1 * W * 3;
    if (!(x0 && y0)) w00 = 0;
    if (!(x1 && y0)) w01 = 0;
    if (!(x0 && y1)) w10 = 0;
    if (!(x1 && y1)) w11 = 0;
    const double mean[3] = {mean0, mean1, mean2}, sd[3] = {std0, std1, std2};
    float* o = out + (size_t)n * 3 * res * res + p;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int acc = r0[cx0 * 3 + c] * w00 + r0[cx1 * 3 + c] * w01 + r1[cx0 * 3 + c] * w10 + r1[cx1 * 3 + c] * w11;
        const int v = min(max((acc + (1 << 14)) >> 15, 0), 255);
        o[(size_t)c * res * res] = (float)(((double)v - mean[c]) / sd[c]);
    }
}

// roi_coord_2d: bilinear fp32 [H][W][2] -> fp32 [B][2][res][res]
__global__ __launch_bounds__(256) void roi_crop_coord_kernel(const gdrn_roi_task* __restrict__ tasks, const double* __restrict__ minv, int res,
                                                             float* __restrict__ out) {
    __shared__ double M[6];
    __shared__ const float* src_s;
    __shared__ int HW[2];
    const int n = blockIdx.y;
    if (threadIdx.x < 6) M[threadIdx.x] = minv[((size_t)n * 2 + 1) * 6 + threadIdx.x];
    if (threadIdx.x == 6) { src_s = tasks[n].coord2d; HW[0] = tasks[n].H; HW[1] = tasks[n].W; }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= res * res) return;
    const int y = p / res, x = p - y * res;
    const float* src = src_s;
    const int H = HW[0], W = HW[1];
    int sx, sy, fx, fy;
    coords_linear(M, x, y, sx, sy, fx, fy);
    const float tx = (float)fx * (1.0f / INTER_TAB), ty = (float)fy * (1.0f / INTER_TAB);
    const float w00 = (1.0f - ty) * (1.0f - tx), w01 = (1.0f - ty) * tx, w10 = ty * (1.0f - tx), w11 = ty * tx;
    const bool x0 = (unsigned)sx < (unsigned)W, x1 = (unsigned)(sx + 1) < (unsigned)W;
    const bool y0 = (unsigned)sy < (unsigned)H, y1 = (unsigned)(sy + 1) < (unsigned)H;
    const int cx0 = min(max(sx, 0), W - 1), cx1 = min(max(sx + 1, 0), W - 1), cy0 = min(max(sy, 0), H - 1), cy1 = min(max(sy + 1, 0), H - 1);
    const float2 z = make_float2(0.f, 0.f);
    const float2* s2 = reinterpret_cast<const float2*>(src);
    float2 a = s2[(size_t)cy0 * W + cx0], b = s2[(size_t)cy0 * W + cx1], c = s2[(size_t)cy1 * W + cx0], d = s2[(size_t)cy1 * W + cx1];
    if (!(x0 && y0)) a = z;
    if (!(x1 && y0)) b = z;
    if (!(x0 && y1)) c = z;
    if (!(x1 && y1)) d = z;
    float* o = out + (size_t)n * 2 * res * res + p;
    o[0] = ((a.x * w00 + b.x * w01) + c.x * w10) + d.x * w11;
    o[(size_t)res * res] = ((a.y * w00 + b.y * w01) + c.y * w10) + d.y * w11;
}

// train-mode targets at the head's resolution: nearest crops of xyz / masks, region labels, extent normalisation
__global__ __launch_bounds__(256) void roi_targets_kernel(const gdrn_roi_task* __restrict__ tasks, const double* __restrict__ minv, int res,
                                                          const double* __restrict__ fps, int nfps, const float* __restrict__ extents,
                                                          float* __restrict__ roi_xyz, float* __restrict__ m_trunc, float* __restrict__ m_visib,
                                                          float* __restrict__ m_obj, int* __restrict__ region) {
    extern __shared__ double fp[];  // [nfps][3]
    __shared__ double M[6];
    __shared__ gdrn_roi_task tk;
    const int n = blockIdx.y;
    if (threadIdx.x < 6) M[threadIdx.x] = minv[((size_t)n * 2 + 1) * 6 + threadIdx.x];
    if (threadIdx.x == 6) tk = tasks[n];
    __syncthreads();
    if (fps)
        for (int i = threadIdx.x; i < nfps * 3; i += 256) fp[i] = fps[(size_t)tk.cls * nfps * 3 + i];
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= res * res) return;
    const int y = p / res, x = p - y * res;
    int sx, sy;
    coords_nearest(M, x, y, sx, sy);
    const bool in = (unsigned)sx < (unsigned)tk.W && (unsigned)sy < (unsigned)tk.H;
    float v[3] = {0.f, 0.f, 0.f};
    float seg = 0.f, trunc = 1.f;
    if (in) {
        if (sx >= tk.x1 && sx <= tk.x2 && sy >= tk.y1 && sy <= tk.y2) {
            const float* q = tk.xyz_crop + ((size_t)(sy - tk.y1) * (tk.x2 - tk.x1 + 1) + (sx - tk.x1)) * 3;
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2];
        }
        seg = (float)tk.seg[(size_t)sy * tk.W + sx];
        if (tk.trunc) trunc = (float)tk.trunc[(size_t)sy * tk.W + sx];
    }
    const float obj = (v[0] != 0.f || v[1] != 0.f || v[2] != 0.f) ? 1.f : 0.f;
    const float visib = seg * obj;
    const size_t o = (size_t)n * res * res + p;
    m_obj[o] = obj;
    m_visib[o] = visib;
    m_trunc[o] = tk.trunc ? visib * trunc : visib;
    if (region) {
        // first index of min_k sqrt(d2_k), as numpy's argmin over scipy's cdist.  sqrt is monotone, so the running minimum of the
        // distances is sqrt(running minimum of d2): only a candidate with a strictly smaller d2 can lower it, and it takes the
        // label only if its ROUNDED distance is strictly smaller too (two different d2 may round to one distance: first wins).
        int best = 0;
        double bd2 = INFINITY, bd = INFINITY;
        for (int k = 0; k < nfps; k++) {
            const double dx = (double)v[0] - fp[k * 3], dy = (double)v[1] - fp[k * 3 + 1], dz = (double)v[2] - fp[k * 3 + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < bd2) {
                bd2 = d2;
                const double d = sqrt(d2);
                if (d < bd) { bd = d; best = k; }
            }
        }
        region[o] = obj != 0.f ? best + 1 : 0;
    }
    const float* e = extents + (size_t)tk.cls * 3;
    float* xo = roi_xyz + (size_t)n * 3 * res * res + p;
#pragma unroll
    for (int c = 0; c < 3; c++) xo[(size_t)c * res * res] = v[c] / e[c] + 0.5f;
}

}  // namespace

extern "C" int gdrn_roi_affine(const gdrn_roi_task* tasks_dev, int B, int in_res, int out_res, double* minv, float* roi_wh,
                               float* resize_ratio, float* trans_ratio, void* stream) {
    if (!tasks_dev || !minv || B <= 0 || in_res <= 0 || out_res <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(roi_affine_kernel, dim3(cdiv(B * 2, 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), tasks_dev, B, in_res,
                       out_res, minv, roi_wh, resize_ratio, trans_ratio);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_roi_crop_inputs(const gdrn_roi_task* tasks_dev, const double* minv, int B, int in_res, int out_res,
                                    const double* pixel_mean, const double* pixel_std, float* roi_img, float* roi_coord_2d, void* stream) {
    if (!tasks_dev || !minv || B <= 0 || in_res <= 0 || out_res <= 0 || B > 65535) return GDRN_ERR_ARG;
    if (roi_img && (!pixel_mean || !pixel_std)) return GDRN_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (roi_img) {
        GDRN_LAUNCH(roi_crop_image_kernel, dim3(cdiv(in_res * in_res, 256), B), dim3(256), 0, st, tasks_dev, minv, in_res, pixel_mean[0],
                           pixel_mean[1], pixel_mean[2], pixel_std[0], pixel_std[1], pixel_std[2], roi_img);
        GDRN_CHECK_LAUNCH();
    }
    if (roi_coord_2d) {
        GDRN_LAUNCH(roi_crop_coord_kernel, dim3(cdiv(out_res * out_res, 256), B), dim3(256), 0, st, tasks_dev, minv, out_res, roi_coord_2d);
        GDRN_CHECK_LAUNCH();
    }
    return GDRN_OK;
}

extern "C" int gdrn_roi_targets(const gdrn_roi_task* tasks_dev, const double* minv, int B, int out_res, const double* fps_points, int nfps,
                                const float* extents, float* roi_xyz, float* roi_mask_trunc, float* roi_mask_visib, float* roi_mask_obj,
                                int* roi_region, void* stream) {
    if (!tasks_dev || !minv || !extents || !roi_xyz || !roi_mask_trunc || !roi_mask_visib || !roi_mask_obj || B <= 0 || out_res <= 0 || B > 65535)
        return GDRN_ERR_ARG;
    if ((roi_region != nullptr) != (fps_points != nullptr)) return GDRN_ERR_ARG;
    if (roi_region && (nfps <= 0 || nfps > 1024)) return GDRN_ERR_ARG;
    const size_t lds = roi_region ? (size_t)nfps * 3 * sizeof(double) : 0;
    GDRN_LAUNCH(roi_targets_kernel, dim3(cdiv(out_res * out_res, 256), B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), tasks_dev,
                       minv, out_res, fps_points, nfps, extents, roi_xyz, roi_mask_trunc, roi_mask_visib, roi_mask_obj, roi_region);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

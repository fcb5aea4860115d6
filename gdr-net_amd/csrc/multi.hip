// Multi-tensor ("one launch for all 148 parameter tensors") versions of the per-step housekeeping kernels for
// gfx950: operand packing (fp32 masters -> kernel layouts, incl. the fragment-major halo operands), gradient
// unpacking and the fused Ranger step.  The per-tensor versions in pack.hip cost ~370 launches / ~2.8 ms per
// training step at bs=64; the work itself is a few hundred MB of HBM traffic.
// Task tables live in device memory (built once by the host, see engine.py / ranger.py); a workgroup finds its
// task by binary search in a prefix array.
#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

__device__ __forceinline__ int find_task(const int* __restrict__ start, int n, int b) {
    int lo = 0, hi = n;  // start[lo] <= b < start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= b) lo = mid; else hi = mid;
    }
    return lo;
}

constexpr int PACK_CHUNK = 2048;  // elements per workgroup

template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const gdrn_pack_task* __restrict__ tasks, const int* __restrict__ blk_start,
                                                         int ntasks) {
    constexpr int EPS = 128 / (int)sizeof(T), GE = 16 / (int)sizeof(T);
    const int t = find_task(blk_start, ntasks, blockIdx.x);
    const gdrn_pack_task k = tasks[t];
    const long long base = (long long)(blockIdx.x - blk_start[t]) * PACK_CHUNK;
    T* dst = reinterpret_cast<T*>(k.dst);
    for (int e = threadIdx.x; e < PACK_CHUNK; e += 256) {
        const long long i = base + e;
        if (i >= k.n) break;
        int a1, a2, tt, b;
        if (k.frag) {
            // fragment-major destination (conv3x3_halo.hip): granule index ((((cb*9 + tap)*kch + kc)*2 + ks)*64 + lane)
            const int el = (int)(i % GE);
            long long r = i / GE;
            const int lane = (int)(r & 63); r >>= 6;
            const int ks = (int)(r & 1); r >>= 1;
            const int kch = k.B / EPS;
            const int kc = (int)(r % kch); r /= kch;
            tt = (int)(r % 9);
            a1 = (int)(r / 9) * 16 + (lane & 15);
            a2 = 0;
            b = kc * EPS + (ks * 4 + (lane >> 4)) * GE + el;
        } else {
            b = (int)(i % k.B);
            long long r = i / k.B;
            tt = (int)(r % k.T); r /= k.T;
            a2 = (int)(r % k.A2);
            a1 = (int)(r / k.A2);
        }
        float v = 0.f;
        if (a1 < k.A1v && a2 < k.A2v && b < k.Bv) {
            const int ts = k.flip ? (k.T - 1 - tt) : tt;
            v = k.src[a1 * k.s1 + a2 * k.s2 + ts * k.st + b * k.sb];
        }
        st1<T>(dst + i, v);
    }
}

__global__ __launch_bounds__(256) void unpack_multi_kernel(const gdrn_pack_task* __restrict__ tasks, const int* __restrict__ blk_start,
                                                           int ntasks) {
    const int t = find_task(blk_start, ntasks, blockIdx.x);
    const gdrn_pack_task k = tasks[t];  // src = packed fp32 [A1][A2][T][B], dst = parameter-layout gradient
    const long long base = (long long)(blockIdx.x - blk_start[t]) * PACK_CHUNK;
    float* dst = reinterpret_cast<float*>(k.dst);
    for (int e = threadIdx.x; e < PACK_CHUNK; e += 256) {
        const long long i = base + e;  // index over the VALID region [A1v][A2v][T][Bv]
        if (i >= k.n) break;
        const int b = (int)(i % k.Bv);
        long long r = i / k.Bv;
        const int tt = (int)(r % k.T); r /= k.T;
        const int a2 = (int)(r % k.A2v);
        const int a1 = (int)(r / k.A2v);
        const int ts = k.flip ? (k.T - 1 - tt) : tt;
        dst[a1 * k.s1 + a2 * k.s2 + ts * k.st + b * k.sb] = k.src[(((long long)a1 * k.A2 + a2) * k.T + tt) * k.B + b];
    }
}

// one workgroup per (tensor, row): gradient centralisation (row mean), RAdam moments, update, optional lookahead
__global__ __launch_bounds__(256) void ranger_multi_kernel(const gdrn_ranger_task* __restrict__ tasks, const int* __restrict__ row_start,
                                                           int ntasks, float beta1, float beta2, float eps, float wd, float step_size,
                                                           int adaptive, int lookahead, float alpha) {
    __shared__ float red[4];
    const int t = find_task(row_start, ntasks, blockIdx.x);
    const gdrn_ranger_task k = tasks[t];
    const size_t base = (size_t)(blockIdx.x - row_start[t]) * k.cols;
    float mean = 0.f;
    if (k.gc) {
        float s = 0.f;
        for (int i = threadIdx.x; i < k.cols; i += 256) s += k.g[base + i];
        mean = block_sum_256(s, red) / (float)k.cols;
    }
    for (int i = threadIdx.x; i < k.cols; i += 256) {
        const size_t j = base + i;
        const float gr = k.g[j] - mean;
        const float vv = k.v[j] * beta2 + (1.f - beta2) * gr * gr;
        const float mm = k.m[j] * beta1 + (1.f - beta1) * gr;
        k.v[j] = vv;
        k.m[j] = mm;
        float pp = k.p[j];
        if (wd != 0.f) pp += -wd * k.lr * pp;
        if (adaptive) pp += -step_size * k.lr * mm / (sqrtf(vv) + eps);
        else pp += -step_size * k.lr * mm;
        if (lookahead) {
            const float sl = k.slow[j] + alpha * (pp - k.slow[j]);
            k.slow[j] = sl;
            pp = sl;
        }
        k.p[j] = pp;
    }
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int gdrn_pack_chunk(void) { return PACK_CHUNK; }

extern "C" int gdrn_pack_multi(const gdrn_pack_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, int dtype, void* stream) {
    if (!tasks_dev || !blk_start_dev || ntasks <= 0 || nblocks <= 0) return GDRN_ERR_ARG;
    if (dtype == GDRN_DT_F32) hipLaunchKernelGGL(pack_multi_kernel<float>, dim3(nblocks), dim3(256), 0, ST, tasks_dev, blk_start_dev, ntasks);
    else if (dtype == GDRN_DT_BF16) hipLaunchKernelGGL(pack_multi_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, ST, tasks_dev, blk_start_dev, ntasks);
    else return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_unpack_multi(const gdrn_pack_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream) {
    if (!tasks_dev || !blk_start_dev || ntasks <= 0 || nblocks <= 0) return GDRN_ERR_ARG;
    hipLaunchKernelGGL(unpack_multi_kernel, dim3(nblocks), dim3(256), 0, ST, tasks_dev, blk_start_dev, ntasks);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_ranger_multi(const gdrn_ranger_task* tasks_dev, const int* row_start_dev, int ntasks, int total_rows, float beta1,
                                 float beta2, float eps, float weight_decay, float step_size, int adaptive, int lookahead, float alpha,
                                 void* stream) {
    if (!tasks_dev || !row_start_dev || ntasks <= 0 || total_rows <= 0) return GDRN_ERR_ARG;
    hipLaunchKernelGGL(ranger_multi_kernel, dim3(total_rows), dim3(256), 0, ST, tasks_dev, row_start_dev, ntasks, beta1, beta2, eps,
                       weight_decay, step_size, adaptive, lookahead, alpha);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

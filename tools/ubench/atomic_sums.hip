// micro-benchmark (round 4): cost of per-workgroup 64-bit integer atomics into a small [slots][2][C] table -- the question behind moving the
// BatchNorm statistics from per-tile rows + a finalize launch to order-independent fixed-point sums that the consumer kernel reads itself.
// Each workgroup adds one value per (statistic, channel) of its 128-channel tile, as a conv epilogue would (lanes of 4 waves own the channels).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void sums_kernel(unsigned long long* tab, int C, int slots, int spin) {
    // fake main loop so that workgroups do not all arrive at once
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    const int nt = blockIdx.x & 1, mt = blockIdx.x >> 1;            // two 128-channel tiles per pixel tile (C = 256)
    const int slot = mt % slots;
    const int c = nt * 128 + (threadIdx.x & 127), s = threadIdx.x >> 7;
    atomicAdd(tab + ((size_t)slot * 2 + s) * C + c, (unsigned long long)(long long)(v * 1048576.f));
}

__global__ __launch_bounds__(256) void rows_kernel(float* rows, int C, int spin) {
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    const int nt = blockIdx.x & 1, mt = blockIdx.x >> 1;
    const int c = nt * 128 + (threadIdx.x & 127), s = threadIdx.x >> 7;
    rows[((size_t)mt * 2 + s) * C + c] = v;
}

int main() {
    const int C = 256;
    unsigned long long* tab;
    float* rows;
    hipMalloc(&tab, 64 * 2 * C * 8);
    hipMalloc(&rows, (size_t)4096 * 2 * C * 4);
    hipMemset(tab, 0, 64 * 2 * C * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nwg : {256, 512, 2048, 8192}) {
        for (int spin : {0, 2000}) {
            for (int slots : {0, 1, 4, 16}) {
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    for (int i = 0; i < 20; ++i) {
                        if (slots == 0) hipLaunchKernelGGL(rows_kernel, dim3(nwg), dim3(256), 0, 0, rows, C, spin);
                        else hipLaunchKernelGGL(sums_kernel, dim3(nwg), dim3(256), 0, 0, tab, C, slots, spin);
                    }
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep == 2) printf("wg %5d spin %4d %s %2d: %7.2f us per launch\n", nwg, spin, slots ? "atomic slots" : "plain rows  ", slots, ms * 1e3 / 20);
                }
            }
        }
    }
    return 0;
}

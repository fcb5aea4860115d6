// micro-benchmark (bring-up aid): issue rate of v_mfma_f32_32x32x16_bf16 in the access pattern of conv3x3_v3 -- 8 accumulators per wave,
// operands from LDS (ds_read_b128 one sub-step ahead), optional s_barrier per 16 MFMAs, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

template <int MODE>  // 0: MFMA only, registers; 1: + LDS operand reads; 2: + barrier per 16 MFMAs; 3: as 2 with 32x32 replaced by 16x16x32 pairs
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters, const uint4* src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = src[i & 1023];
    __syncthreads();
    f32x16_t acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    uint4 fa[2][4], fb[2][2];
    const unsigned char* ab = smem + lane * 16;
    const unsigned char* bb = smem + 65536 + (lane & 31) * 144 + (lane >> 5) * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[0][a] = *reinterpret_cast<const uint4*>(ab + a * 1024);
#pragma unroll
    for (int b = 0; b < 2; ++b) fb[0][b] = *reinterpret_cast<const uint4*>(bb + b * 5184);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if constexpr (MODE >= 1) {
#pragma unroll
                for (int a = 0; a < 4; ++a) fa[sl ^ 1][a] = *reinterpret_cast<const uint4*>(ab + ((it + sl) & 3) * 8192 + a * 1024);
#pragma unroll
                for (int b = 0; b < 2; ++b) fb[sl ^ 1][b] = *reinterpret_cast<const uint4*>(bb + b * 5184 + ((it + sl) & 3) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[sl][a]), __builtin_bit_cast(bf16x8_t, fb[sl][b]), acc[a * 2 + b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    if (s == 123.456f) out[1] = 1;
    if (lane == 0) out[2 + blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

extern "C" int run(int mode, int threads, int blocks, int iters, unsigned long long* out, const void* src) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t sm = 140 * 1024;
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), sm, 0, out, iters, (const uint4*)src);
    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), sm, 0, out, iters, (const uint4*)src);
    else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), sm, 0, out, iters, (const uint4*)src);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

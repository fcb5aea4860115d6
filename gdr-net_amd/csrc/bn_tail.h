// BatchNorm statistics finished by the producing conv's own epilogue (gdrn_bn_desc, include/gdrn_hip.h).
//
// Every workgroup of the launch adds its per-tile partial sums (the fp32 values the per-tile rows used to hold) to a small table of 64-bit
// FIXED-POINT integers with device-scope atomics -- integer addition is associative, so the totals do not depend on arrival order and the
// step stays bit-reproducible -- and bumps an arrival counter; the workgroup that arrives last converts the totals into the per-channel
// vectors with the arithmetic of gdrn_bn_finalize / gdrn_bn_bwd_coef (norm.hip) and clears table and counter.  Ordering without an
// agent-scope release fence (which would write back the whole L2): the table is only ever touched by device-scope atomics (RMW, load, store);
// a wave waits for its own adds to be acknowledged (s_waitcnt vmcnt(0): gfx9 counts stores and return-less atomics there) before the
// workgroup's single counter RMW, so whoever reads counter == n - 1 knows every other workgroup's adds were performed.
// Measured (tools/ubench/atomic_sums.hip): with 8 slots the adds are free beside any real epilogue (256 .. 8192 workgroups).
#pragma once
#include "common.h"

namespace bn_tail {

constexpr double FIX_FWD = 16777216.0;          // 2^24: forward sums (|x| up to ~1e3 per pixel over 2.6e5 pixels stay below 2^63)
constexpr double FIX_BWD = 1099511627776.0;     // 2^40: backward sums (gradients ~1e-6 .. 1, incl. the fp16 loss scale)

__device__ __forceinline__ void add(const gdrn_bn_desc* d, int slot, int s, int c, float v) {
    const long long q = __double2ll_rn((double)v * (d->kind ? FIX_BWD : FIX_FWD));
    __hip_atomic_fetch_add(d->sums + ((size_t)(slot * 2 + s)) * d->C + c, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// after every wave of the workgroup has issued its add() calls.  True (in all threads) in the workgroup that arrived last.
__device__ __forceinline__ bool arrive(const gdrn_bn_desc* d, unsigned nwg, unsigned* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(d->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_flag = (old == nwg - 1u) ? 1u : 0u;
    }
    __syncthreads();
    return *lds_flag != 0u;
}

__device__ __forceinline__ void totals(const gdrn_bn_desc* d, int c, double& s1, double& s2) {
    long long t1 = 0, t2 = 0;
#pragma unroll
    for (int sl = 0; sl < GDRN_BN_SLOTS; ++sl) {
        long long* p1 = d->sums + ((size_t)(sl * 2 + 0)) * d->C + c;
        long long* p2 = d->sums + ((size_t)(sl * 2 + 1)) * d->C + c;
        t1 += __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2 += __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p1, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p2, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double inv = 1.0 / (d->kind ? FIX_BWD : FIX_FWD);
    s1 = (double)t1 * inv;
    s2 = (double)t2 * inv;
}

// the last workgroup (256 threads): totals -> vectors.  The arithmetic below is bn_finalize_rows_kernel's / bn_bwd_coef_kernel's.
__device__ __forceinline__ void finish(const gdrn_bn_desc* d) {
    for (int ch = threadIdx.x; ch < d->C; ch += 256) {
        double s1, s2;
        totals(d, ch, s1, s2);
        if (d->kind == 0) {
            const double count = d->count;
            const double m = s1 / count;
            double var = s2 / count - m * m;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + (double)d->eps);
            d->mean[ch] = (float)m;
            d->invstd[ch] = (float)is;
            const float sc = d->gamma[ch] * (float)is;
            d->scale[ch] = sc;
            d->shift[ch] = d->beta[ch] - (float)m * sc;
            if (d->running_mean != nullptr) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                d->running_mean[ch] = (1.f - d->momentum) * d->running_mean[ch] + d->momentum * (float)m;
                d->running_var[ch] = (1.f - d->momentum) * d->running_var[ch] + d->momentum * (float)unb;
            }
        } else {
            float m1 = (float)s1, m2 = (float)s2;
            if (d->dgamma != nullptr) { d->dbeta[ch] = m1; d->dgamma[ch] = m2; }
            const float inv_n = (float)(1.0 / d->count);
            m1 *= inv_n;
            m2 *= inv_n;
            const float k_is = d->invstd[ch];
            const float a = d->gamma[ch] * k_is, b = -a * k_is * m2;
            d->ka[ch] = a;
            d->kb[ch] = b;
            d->kc[ch] = -a * m1 - b * d->mean[ch];
        }
    }
    if (threadIdx.x == 0) {
        if (d->kind == 0 && d->nbt != nullptr) *d->nbt += 1;
        __hip_atomic_store(d->counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace bn_tail

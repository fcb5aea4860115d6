// BatchNorm statistics without a launch of their own (gdrn_bn_desc, include/gdrn_hip.h).
//
// PRODUCER (the conv whose output the BatchNorm normalises / the data-gradient conv whose epilogue carries the BatchNorm-backward sums):
// every workgroup adds its per-tile partial sums -- the fp32 values the per-tile rows hold in the other mode -- to a small table of 64-bit
// FIXED-POINT integers, sums[slot = tile % GDRN_BN_SLOTS][2][C], with return-less device-scope atomics and does not wait for them.  Integer
// addition is associative: the totals do not depend on the arrival order, the step stays bit-reproducible.  Measured (tools/ubench/
// atomic_sums.hip): with 8 slots the adds are free beside any real epilogue, 256 .. 8192 workgroups.
// CONSUMER: the kernel boundary makes the table visible; the conv that applies the BatchNorm while it stages its operand (xf modes 1..4)
// turns the totals into its per-channel coefficient vectors itself, in its prologue -- every workgroup for all its input channels, 2 * 8
// loads and a few fp64 operations per channel, the same code in every workgroup, so all of them see identical coefficients -- and workgroup
// 0 also stores the vectors other kernels read later (mean / invstd / scale / shift + running statistics; ka / kb / kc + dgamma / dbeta).
// That removes the bn_finalize / bn_bwd_coef launch between producer and consumer: ~6 us alone, ~12 us beside a weight-gradient launch,
// 86 per step.  Consumers that cannot do it (bn_apply passes, the second-generation kernel, generic kernels) get gdrn_bn_finish: the same
// arithmetic as a one-workgroup-per-256-channels launch over 32 KB instead of 0.25 .. 4 MB of rows.
// The table is cleared by the host's zero_multi launch at the start of a backward pass (engine.py), never by these kernels.
//
// Measured and rejected (round 4): finishing in the PRODUCER's epilogue -- arrival counter, last workgroup computes the vectors.  Every
// device-scope operation with a return value (the counter RMW, the sc1 loads of the totals) is a round trip of several microseconds on this
// part: +14 us per conv, 7.56 -> 8.36 ms per step.
#pragma once
#include "common.h"

namespace bn_sums {

constexpr double FIX_FWD = 16777216.0;          // 2^24: forward sums (|x| up to ~1e3 per pixel over 2.6e5 pixels stay below 2^63)
constexpr double FIX_BWD = 1099511627776.0;     // 2^40: backward sums (gradients ~1e-6 .. 1, incl. the fp16 loss scale)

__device__ __forceinline__ void add(const gdrn_bn_desc* d, int slot, int s, int c, float v) {
    const long long q = __double2ll_rn((double)v * (d->kind ? FIX_BWD : FIX_FWD));
    __hip_atomic_fetch_add(d->sums + ((size_t)(slot * 2 + s)) * d->C + c, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void totals(const gdrn_bn_desc* d, int c, double& s1, double& s2) {
    long long t1 = 0, t2 = 0;
#pragma unroll
    for (int sl = 0; sl < GDRN_BN_SLOTS; ++sl) {
        t1 += d->sums[((size_t)(sl * 2 + 0)) * d->C + c];
        t2 += d->sums[((size_t)(sl * 2 + 1)) * d->C + c];
    }
    const double inv = 1.0 / (d->kind ? FIX_BWD : FIX_FWD);
    s1 = (double)t1 * inv;
    s2 = (double)t2 * inv;
}

// forward (kind 0): y = scale * x + shift of channel ch.  The arithmetic is bn_finalize_rows_kernel's (norm.hip).  store: also write the
// vectors and move the running statistics (exactly one workgroup of one launch per step does that).
__device__ __forceinline__ void coef_fwd(const gdrn_bn_desc* d, int ch, bool store, float& scale, float& shift) {
    double s1, s2;
    totals(d, ch, s1, s2);
    const double count = d->count;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)d->eps);
    scale = d->gamma[ch] * (float)is;
    shift = d->beta[ch] - (float)m * scale;
    if (store) {
        d->mean[ch] = (float)m;
        d->invstd[ch] = (float)is;
        d->scale[ch] = scale;
        d->shift[ch] = shift;
        if (d->running_mean != nullptr) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            d->running_mean[ch] = (1.f - d->momentum) * d->running_mean[ch] + d->momentum * (float)m;
            d->running_var[ch] = (1.f - d->momentum) * d->running_var[ch] + d->momentum * (float)unb;
        }
        if (ch == 0 && d->nbt != nullptr) *d->nbt += 1;
    }
}

// backward (kind 1): dx = a * g + b * x + c of channel ch (bn_bwd_coef_kernel's arithmetic); store: also ka / kb / kc, dgamma / dbeta.
__device__ __forceinline__ void coef_bwd(const gdrn_bn_desc* d, int ch, bool store, float& a, float& b, float& c) {
    double s1, s2;
    totals(d, ch, s1, s2);
    float m1 = (float)s1, m2 = (float)s2;
    if (store && d->dgamma != nullptr) { d->dbeta[ch] = m1; d->dgamma[ch] = m2; }
    const float inv_n = (float)(1.0 / d->count);
    m1 *= inv_n;
    m2 *= inv_n;
    const float k_is = d->invstd[ch];
    a = d->gamma[ch] * k_is;
    b = -a * k_is * m2;
    c = -a * m1 - b * d->mean[ch];
    if (store) { d->ka[ch] = a; d->kb[ch] = b; d->kc[ch] = c; }
}

}  // namespace bn_sums

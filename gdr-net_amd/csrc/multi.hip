// Multi-tensor ("one launch for all 148 parameter tensors") versions of the per-step housekeeping kernels for
// gfx950: operand packing (fp32 masters -> kernel layouts, incl. the fragment-major halo operands), gradient
// unpacking and the fused Ranger step.  The per-tensor versions in pack.hip cost ~370 launches / ~2.8 ms per
// training step at bs=64; the work itself is a few hundred MB of HBM traffic.
// Task tables live in device memory (built once by the host, see engine.py / ranger.py); a workgroup finds its
// task by binary search in a prefix array.
#include <algorithm>

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

// ---- tiled transpose (frag == 3; r6c): a row-major copy whose unit-stride SOURCE index is not the destination's fastest one -- fc1's two operand
// copies and its gradient unpack (8.4 M elements each), the data-gradient operands of the 1x1 / fc layers.  The granule-gather path reads
// such a source 4 bytes per 64-byte sector (b strides of 64 .. 8192 floats).  Here a workgroup moves a 64 (u) x 64 (b) tile through LDS: the
// source is read along u (pad_ = 1 / 2 / 3: u = a1 / a2 / t, the index whose source stride is 1), the destination written along b.
// Block index = ((o1 * O2 + o2) * UT + ut) * BT + bt with (o1, o2) the two other indices in (a1, a2, t) order.
struct TrGeo { int U, Uv, O1, O2, O1v, O2v, UT, BT; };
__host__ __device__ inline TrGeo tr_geo(const gdrn_pack_task& k) {
    TrGeo g;
    const int d[3] = {k.A1, k.A2, k.T}, dv[3] = {k.A1v, k.A2v, k.T};
    const int u = k.pad_ - 1, a = u == 0 ? 1 : 0, b = u == 2 ? 1 : 2;
    g.U = d[u]; g.Uv = dv[u]; g.O1 = d[a]; g.O2 = d[b]; g.O1v = dv[a]; g.O2v = dv[b];
    g.UT = (g.U + 63) / 64; g.BT = (k.B + 63) / 64;
    return g;
}
__host__ inline bool tr_ok(const gdrn_pack_task& k) {
    if (k.frag != 3 || k.pad_ < 1 || k.pad_ > 3 || k.flip || k.scale != nullptr) return false;
    const long long s[3] = {k.s1, k.s2, k.st};
    return s[k.pad_ - 1] == 1 && k.A1 > 0 && k.A2 > 0 && k.T > 0 && k.B > 0;
}
// (a1, a2, t) of the tile's u index `u` and the block's other indices
__device__ __forceinline__ void tr_idx(const gdrn_pack_task& k, int u, int o1, int o2, int& a1, int& a2, int& t) {
    if (k.pad_ == 1) { a1 = u; a2 = o1; t = o2; }
    else if (k.pad_ == 2) { a1 = o1; a2 = u; t = o2; }
    else { a1 = o1; a2 = o2; t = u; }
}
template <typename T, bool UNPACK>
__device__ __forceinline__ void tr_tile(const gdrn_pack_task& k, int blk) {
    __shared__ float tile[64][68];   // [b][u], rows 16-byte aligned
    const TrGeo g = tr_geo(k);
    const int bt = blk % g.BT; blk /= g.BT;
    const int ut = blk % g.UT; blk /= g.UT;
    const int o2 = blk % g.O2, o1 = blk / g.O2;
    const int u0 = ut * 64, b0 = bt * 64;
    // 16-byte accesses along u on the strided side when every address is 16-byte aligned (all non-unit strides and the valid extent multiples of
    // 4 floats), along b on the contiguous side when B is a multiple of 8; the scalar forms otherwise
    const long long sa = (k.pad_ == 1 ? 0 : k.s1) | (k.pad_ == 2 ? 0 : k.s2) | (k.pad_ == 3 ? 0 : k.st) | k.sb;
    const bool v4 = (sa & 3) == 0 && (g.Uv & 3) == 0 && (reinterpret_cast<uintptr_t>(UNPACK ? k.dst : (const void*)k.src) & 15) == 0;
    const bool b8 = (k.B & 7) == 0 && (reinterpret_cast<uintptr_t>(UNPACK ? (const void*)k.src : k.dst) & 15) == 0;
    auto strided = [&](int u, int b) -> long long {
        int a1, a2, t;
        tr_idx(k, u, o1, o2, a1, a2, t);
        return (long long)a1 * k.s1 + (long long)a2 * k.s2 + (long long)t * k.st + (long long)b * k.sb;
    };
    auto dense = [&](int u, int b) -> long long {
        int a1, a2, t;
        tr_idx(k, u, o1, o2, a1, a2, t);
        return (((long long)a1 * k.A2 + a2) * k.T + t) * k.B + b;
    };
    auto valid = [&](int u, int b) -> bool {
        int a1, a2, t;
        tr_idx(k, u, o1, o2, a1, a2, t);
        return u < g.Uv && b < k.Bv && a1 < k.A1v && a2 < k.A2v;
    };
    if constexpr (!UNPACK) {   // strided fp32 source (unit stride along u) -> contiguous [A1][A2][T][B] destination of type T, zero padded
        if (v4) {
            for (int i = threadIdx.x; i < 1024; i += 256) {
                const int u4 = (i & 15) * 4, bb = i >> 4, u = u0 + u4, b = b0 + bb;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid(u, b)) v = *reinterpret_cast<const float4*>(k.src + strided(u, b));
                *reinterpret_cast<float4*>(&tile[bb][u4]) = v;
            }
        } else {
            for (int i = threadIdx.x; i < 4096; i += 256) {
                const int uu = i & 63, bb = i >> 6, u = u0 + uu, b = b0 + bb;
                tile[bb][uu] = valid(u, b) ? k.src[strided(u, b)] : 0.f;
            }
        }
        __syncthreads();
        T* dst = reinterpret_cast<T*>(k.dst);
        if (b8) {
            for (int i = threadIdx.x; i < 512; i += 256) {
                const int b8_ = (i & 7) * 8, uu = i >> 3, u = u0 + uu, b = b0 + b8_;
                if (u < g.U && b < k.B) {
                    float v[8];
#pragma unroll
                    for (int j2 = 0; j2 < 8; ++j2) v[j2] = tile[b8_ + j2][uu];
                    T* d = dst + dense(u, b);
                    if constexpr (sizeof(T) == 2) *reinterpret_cast<uint4*>(d) = Vec16<T>::pack(v);
                    else { Vec16<T>::store(d, v); Vec16<T>::store(d + 4, v + 4); }
                }
            }
        } else {
            for (int i = threadIdx.x; i < 4096; i += 256) {
                const int bb = i & 63, uu = i >> 6, u = u0 + uu, b = b0 + bb;
                if (u < g.U && b < k.B) st1<T>(dst + dense(u, b), tile[bb][uu]);
            }
        }
    } else {                   // contiguous fp32 [A1][A2][T][B] source -> strided fp32 destination (valid region only)
        if (b8) {
            for (int i = threadIdx.x; i < 1024; i += 256) {
                const int b4 = (i & 15) * 4, uu = i >> 4, u = u0 + uu, b = b0 + b4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (u < g.U && b < k.B) v = *reinterpret_cast<const float4*>(k.src + dense(u, b));
                tile[b4][uu] = v.x; tile[b4 + 1][uu] = v.y; tile[b4 + 2][uu] = v.z; tile[b4 + 3][uu] = v.w;
            }
        } else {
            for (int i = threadIdx.x; i < 4096; i += 256) {
                const int bb = i & 63, uu = i >> 6, u = u0 + uu, b = b0 + bb;
                tile[bb][uu] = (u < g.U && b < k.B) ? k.src[dense(u, b)] : 0.f;
            }
        }
        __syncthreads();
        float* dst = reinterpret_cast<float*>(k.dst);
        if (v4) {
            for (int i = threadIdx.x; i < 1024; i += 256) {
                const int u4 = (i & 15) * 4, bb = i >> 4, u = u0 + u4, b = b0 + bb;
                if (valid(u, b)) *reinterpret_cast<float4*>(dst + strided(u, b)) = *reinterpret_cast<const float4*>(&tile[bb][u4]);
            }
        } else {
            for (int i = threadIdx.x; i < 4096; i += 256) {
                const int uu = i & 63, bb = i >> 6, u = u0 + uu, b = b0 + bb;
                if (valid(u, b)) dst[strided(u, b)] = tile[bb][uu];
            }
        }
    }
}

// One thread converts one 16-byte destination granule (GE consecutive b): one index decode per GE elements -- the
// per-element version was bound by its 64-bit divisions, not by HBM.  Needs B % GE == 0 (true for every layout of the
// path: B is a channel stride padded to the 128-byte K stage); other tables take the scalar tail below.
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const gdrn_pack_task* __restrict__ tasks, const int* __restrict__ blk_start,
                                                         int ntasks) {
    constexpr int EPS = 128 / (int)sizeof(T), GE = 16 / (int)sizeof(T);
    const int t = find_task(blk_start, ntasks, blockIdx.x);
    const gdrn_pack_task k = tasks[t];
    T* dst = reinterpret_cast<T*>(k.dst);
    if (k.frag == 3) { tr_tile<T, false>(k, blockIdx.x - blk_start[t]); return; }
    if constexpr (sizeof(T) == 2) {
        if (k.frag) {
            // Fragment-major 3x3 operand, one workgroup per brick of 16 rows x 64 b x 9 taps (= nine 2 KiB blocks of the
            // destination): the source is read along its contiguous axis (runs of 576 or 144 floats), transposed through
            // LDS and written as whole 16-byte granules.  (The granule-gather version read 4-byte words 36 B apart and
            // spent 210 us per step in the texture addresser.)
            __shared__ bf16_t tile[16][9][72];
            const int brick = blockIdx.x - blk_start[t];
            const int kch = k.B >> 6;
            const int cb = brick / kch, kc = brick - cb * kch;
            const int b0 = kc * 64;
            // source row of block cb, fragment row a: gdrn_pack_wfrag's interleave (blocks of a 32-row group alternate in units of 4 rows
            // for operands of more than 64 rows)
            const int FNp = k.A1 <= 64 ? 1 : 2;
            // frag == 2 (gdrn_pack_wfrag32): brick cb = half (cb & 1) of 32-row fragment cb >> 1, fragment row r = half*16 + a
            auto srow = [&](int a) {
                if (k.frag == 2) { const int f = cb >> 1, r = (cb & 1) * 16 + a; return (f >> 1) * 64 + ((r >> 2) & 1) * 32 + (f & 1) * 16 + (r >> 3) * 4 + (r & 3); }
                return (cb / FNp) * 16 * FNp + (a >> 2) * (4 * FNp) + (cb % FNp) * 4 + (a & 3);
            };
            __shared__ float rsc[16];              // optional per-row factor (eval mode: BatchNorm scale folded into the weights)
            if (threadIdx.x < 16) rsc[threadIdx.x] = (k.scale != nullptr && srow((int)threadIdx.x) < k.A1v) ? k.scale[srow((int)threadIdx.x)] : 1.f;
            __syncthreads();
            // whole bricks of 16-byte-aligned operands: four source floats per load (the kernel waits on memory 80 % of its time: bytes in
            // flight per thread are what it lacks)
            const bool full = srow(15) < k.A1v && b0 + 63 < k.Bv && ((size_t)k.src & 15) == 0;
            if (full && k.st == 1 && k.sb == 9 && (k.s1 & 3) == 0) {  // src[a1*s1 + b*9 + t]: rows of 576 contiguous floats
                for (int q = threadIdx.x; q < 16 * 144; q += 256) {
                    const int a = q / 144, r4 = (q - a * 144) * 4;
                    const float4 v4 = *reinterpret_cast<const float4*>(k.src + (long long)srow(a) * k.s1 + (long long)b0 * 9 + r4);
                    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = r4 + e, b = r / 9, tp = r - b * 9;
                        tile[a][k.flip ? 8 - tp : tp][b] = f2bf(vv[e] * rsc[a]);
                    }
                }
            } else if (full && k.st == 1 && k.s1 == 9 && (k.sb & 3) == 0) {  // src[a1*9 + b*sb + t]: per b a run of 144 contiguous floats (rows srow(0..15) contiguous
                                                          // only for the non-interleaved 64-row operands; else per 4-row group of 36 floats)
                for (int q = threadIdx.x; q < 64 * 4 * 9; q += 256) {   // (b, 4-row group, 9 float4 of its 36 floats)
                    const int b = q / 36, r = q - b * 36, grp = r / 9, f4 = r - grp * 9;
                    const float4 v4 = *reinterpret_cast<const float4*>(k.src + (long long)srow(grp * 4) * 9 + (long long)(b0 + b) * k.sb + f4 * 4);
                    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int w = f4 * 4 + e, al = w / 9, tp = w - al * 9, a = grp * 4 + al;
                        tile[a][k.flip ? 8 - tp : tp][b] = f2bf(vv[e] * rsc[a]);
                    }
                }
            } else if (k.st == 1 && k.sb == 9) {          // src[a1*s1 + b*9 + t]
                for (int idx = threadIdx.x; idx < 16 * 576; idx += 256) {
                    const int a = idx / 576, r = idx - a * 576, b = r / 9, tp = r - b * 9;
                    const bool ok = srow(a) < k.A1v && b0 + b < k.Bv;
                    const float v = ok ? k.src[(long long)srow(a) * k.s1 + (long long)(b0 + b) * 9 + tp] * rsc[a] : 0.f;
                    tile[a][k.flip ? 8 - tp : tp][b] = f2bf(v);
                }
            } else if (k.st == 1 && k.s1 == 9) {   // src[a1*9 + b*sb + t]
                for (int idx = threadIdx.x; idx < 64 * 144; idx += 256) {
                    const int b = idx / 144, r = idx - b * 144, a = r / 9, tp = r - a * 9;
                    const bool ok = srow(a) < k.A1v && b0 + b < k.Bv;
                    const float v = ok ? k.src[(long long)srow(a) * 9 + (long long)(b0 + b) * k.sb + tp] * rsc[a] : 0.f;
                    tile[a][k.flip ? 8 - tp : tp][b] = f2bf(v);
                }
            } else {
                for (int idx = threadIdx.x; idx < 16 * 576; idx += 256) {
                    const int a = idx / 576, r = idx - a * 576, b = r / 9, tp = r - b * 9;
                    const bool ok = srow(a) < k.A1v && b0 + b < k.Bv;
                    const float v = ok ? k.src[(long long)srow(a) * k.s1 + (long long)(b0 + b) * k.sb + (long long)tp * k.st] * rsc[a] : 0.f;
                    tile[a][k.flip ? 8 - tp : tp][b] = f2bf(v);
                }
            }
            __syncthreads();
            if (k.frag == 2) {
                // destination block ((kc*9 + tap)*4 + ks)*(A1/32) + f, lane = hh*32 + half*16 + a: row a of this brick, k = (ks*2 + hh)*8 ..+7
                const int nfr = k.A1 >> 5, f = cb >> 1, half = cb & 1;
                for (int gi = threadIdx.x; gi < 9 * 4 * 32; gi += 256) {
                    const int a = gi & 15, hh = (gi >> 4) & 1, ks = (gi >> 5) & 3, tap = gi >> 7;
                    const uint4 v = *reinterpret_cast<const uint4*>(&tile[a][tap][(ks * 2 + hh) * 8]);
                    *reinterpret_cast<uint4*>(dst + ((size_t)((((kc * 9 + tap) * 4 + ks) * nfr + f) * 64 + hh * 32 + half * 16 + a) << 3)) = v;
                }
                return;
            }
            // destination granule ((((cb*9 + tap)*kch + kc)*2 + ks)*64 + lane): row lane&15, b = (ks*4 + (lane>>4))*8 ..+7
            for (int gi = threadIdx.x; gi < 9 * 2 * 64; gi += 256) {
                const int lane = gi & 63, ks = (gi >> 6) & 1, tap = gi >> 7;
                const uint4 v = *reinterpret_cast<const uint4*>(&tile[lane & 15][tap][(ks * 4 + (lane >> 4)) * 8]);
                *reinterpret_cast<uint4*>(dst + ((size_t)((((cb * 9 + tap) * kch + kc) * 2 + ks) * 64 + lane) << 3)) = v;
            }
            return;
        }
    }
    const long long base = (long long)(blockIdx.x - blk_start[t]) * PACK_CHUNK;
    if ((k.B % GE) == 0 && (k.n % GE) == 0) {
        const unsigned B = (unsigned)k.B, Tt = (unsigned)k.T, A2 = (unsigned)k.A2;
        for (int g = threadIdx.x; g < PACK_CHUNK / GE; g += 256) {
            const long long i = base + (long long)g * GE;
            if (i >= k.n) break;
            unsigned a1, a2, tt, b0;
            unsigned r = (unsigned)(i / GE);  // granule index (< 2^31: n < 2^31 * GE)
            if (k.frag) {
                // fragment-major destination (conv3x3_halo.hip): granule ((((cb*9 + tap)*kch + kc)*2 + ks)*64 + lane)
                const unsigned lane = r & 63u; r >>= 6;
                const unsigned ks = r & 1u; r >>= 1;
                const unsigned kch = B / EPS;
                unsigned kc;
                if (k.pad_ > 0) { kc = r & (kch - 1u); r >>= (unsigned)(k.pad_ - 1); }  // kch = 2^(pad_-1): no 32-bit division
                else { kc = r % kch; r /= kch; }
                tt = r % 9u;
                {
                    const unsigned cbq = r / 9u, rr = lane & 15u, fnp = (k.A1 <= 64 || sizeof(T) == 4) ? 1u : 2u;
                    a1 = (cbq / fnp) * 16u * fnp + (rr >> 2) * (4u * fnp) + (cbq % fnp) * 4u + (rr & 3u);
                }
                a2 = 0;
                b0 = kc * EPS + (ks * 4u + (lane >> 4)) * GE;
            } else {
                const unsigned bg = B / GE;
                b0 = (r % bg) * GE; r /= bg;
                tt = r % Tt; r /= Tt;
                a2 = r % A2;
                a1 = r / A2;
            }
            float v[GE];
            const bool ok = (int)a1 < k.A1v && (int)a2 < k.A2v;
            const unsigned ts = k.flip ? (Tt - 1u - tt) : tt;
            const float* sp = k.src + (long long)a1 * k.s1 + (long long)a2 * k.s2 + (long long)ts * k.st;
            const float rs = (k.scale != nullptr && ok) ? k.scale[a1] : 1.f;
#pragma unroll
            for (int el = 0; el < GE; ++el) {
                const int b = (int)b0 + el;
                v[el] = (ok && b < k.Bv) ? sp[(long long)b * k.sb] * rs : 0.f;
            }
            Vec16<T>::store(dst + i, v);
        }
        return;
    }
    for (int e = threadIdx.x; e < PACK_CHUNK; e += 256) {
        const long long i = base + e;
        if (i >= k.n) break;
        int a1, a2, tt, b;
        if (k.frag) {
            const int el = (int)(i % GE);
            long long r = i / GE;
            const int lane = (int)(r & 63); r >>= 6;
            const int ks = (int)(r & 1); r >>= 1;
            const int kch = k.B / EPS;
            const int kc = (int)(r % kch); r /= kch;
            tt = (int)(r % 9);
            {
                const int cbq = (int)(r / 9), rr = lane & 15, fnp = (k.A1 <= 64 || sizeof(T) == 4) ? 1 : 2;
                a1 = (cbq / fnp) * 16 * fnp + (rr >> 2) * (4 * fnp) + (cbq % fnp) * 4 + (rr & 3);
            }
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
            v = k.src[a1 * k.s1 + a2 * k.s2 + ts * k.st + b * k.sb] * (k.scale != nullptr ? k.scale[a1] : 1.f);
        }
        st1<T>(dst + i, v);
    }
}

// zero a list of device regions (16-byte granules) in ONE launch: the small gradient accumulators of a backward pass
constexpr int ZERO_CHUNK = 4096;  // 16-byte granules per workgroup (64 KiB)
__global__ __launch_bounds__(256) void zero_multi_kernel(const gdrn_zero_task* __restrict__ tasks, const int* __restrict__ blk_start, int ntasks) {
    const int t = find_task(blk_start, ntasks, blockIdx.x);
    const gdrn_zero_task k = tasks[t];
    const long long base = (long long)(blockIdx.x - blk_start[t]) * ZERO_CHUNK;
    uint4* dst = reinterpret_cast<uint4*>(k.p);
    for (int i = threadIdx.x; i < ZERO_CHUNK; i += 256) {
        const long long g = base + i;
        if (g >= k.n16) break;
        dst[g] = make_uint4(0u, 0u, 0u, 0u);
    }
}

__global__ __launch_bounds__(256) void unpack_multi_kernel(const gdrn_pack_task* __restrict__ tasks, const int* __restrict__ blk_start,
                                                           int ntasks) {
    const int t = find_task(blk_start, ntasks, blockIdx.x);
    const gdrn_pack_task k = tasks[t];  // src = packed fp32 [A1][A2][T][B], dst = parameter-layout gradient
    if (k.frag == 3) { tr_tile<float, true>(k, blockIdx.x - blk_start[t]); return; }
    const long long base = (long long)(blockIdx.x - blk_start[t]) * PACK_CHUNK;
    float* dst = reinterpret_cast<float*>(k.dst);
    if ((k.Bv & 3) == 0 && (k.B & 3) == 0) {  // 4 consecutive b per thread: one decode, one 16-byte read
        const unsigned bq = (unsigned)k.Bv >> 2, Tt = (unsigned)k.T, A2v = (unsigned)k.A2v;
        for (int g = threadIdx.x; g < PACK_CHUNK / 4; g += 256) {
            const long long i = base + (long long)g * 4;  // index over the VALID region [A1v][A2v][T][Bv]
            if (i >= k.n) break;
            unsigned r = (unsigned)(i >> 2);
            const unsigned b = (r % bq) * 4u; r /= bq;
            const unsigned tt = r % Tt; r /= Tt;
            const unsigned a2 = r % A2v, a1 = r / A2v;
            const unsigned ts = k.flip ? (Tt - 1u - tt) : tt;
            const float4 v = *reinterpret_cast<const float4*>(k.src + (((long long)a1 * k.A2 + a2) * k.T + tt) * k.B + b);
            float* dp = dst + (long long)a1 * k.s1 + (long long)a2 * k.s2 + (long long)ts * k.st + (long long)b * k.sb;
            dp[0] = v.x;
            dp[k.sb] = v.y;
            dp[2 * k.sb] = v.z;
            dp[3 * k.sb] = v.w;
        }
        return;
    }
    for (int e = threadIdx.x; e < PACK_CHUNK; e += 256) {
        const long long i = base + e;
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
                                                           int adaptive, int lookahead, float alpha, float grad_scale,
                                                           const gdrn_loss_scale_state* __restrict__ dyn, int n_sma_threshold, int lookahead_k) {
    __shared__ float red[4];
    if (dyn != nullptr) {
        // fp16 arithmetic mode, dynamic loss scale decided ON THE DEVICE (r6): an overflowed step (gdrn_nonfinite_flag raised dyn->flag) makes
        // the whole update a no-op -- parameters, moments, slow weights untouched, as GradScaler.step skips optimizer.step -- and the step index
        // the RAdam rectification is evaluated at counts APPLIED steps only (ranger.py:154-186 restated on the device, in double as Python's)
        if (dyn->flag != 0) return;
        const int step = dyn->base_step + dyn->applied + 1;
        const double b2t = pow((double)beta2, (double)step), b1t = pow((double)beta1, (double)step);
        const double nmax = 2.0 / (1.0 - (double)beta2) - 1.0;
        const double nsma = nmax - 2.0 * step * b2t / (1.0 - b2t);
        adaptive = nsma > (double)n_sma_threshold;
        step_size = (float)(adaptive ? sqrt((1.0 - b2t) * (nsma - 4.0) / (nmax - 4.0) * (nsma - 2.0) / nsma * nmax / (nmax - 2.0)) / (1.0 - b1t)
                                     : 1.0 / (1.0 - b1t));
        lookahead = (step % lookahead_k) == 0;
        grad_scale *= dyn->inv_scale;
    }
    const int t = find_task(row_start, ntasks, blockIdx.x);
    const gdrn_ranger_task k = tasks[t];
    const size_t base = (size_t)(blockIdx.x - row_start[t]) * k.cols;
    ranger_row(k.p + base, k.g + base, k.m + base, k.v + base, k.slow + base, k.cols, k.gc, k.lr, beta1, beta2, eps, wd, step_size, adaptive, lookahead, alpha,
               grad_scale, red);
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int gdrn_pack_chunk(void) { return PACK_CHUNK; }

// workgroups of a tiled-transpose task (frag = 3, pad_ = 1 / 2 / 3: the index a1 / a2 / t has source stride 1) of gdrn_pack_multi /
// gdrn_unpack_multi; <= 0: the task does not qualify (flip, scale, or the named index is not the unit-stride one)
extern "C" int gdrn_pack_transpose_blocks(const gdrn_pack_task* k) {
    if (!k || !tr_ok(*k)) return 0;
    const TrGeo g = tr_geo(*k);
    const long long nb = (long long)g.O1 * g.O2 * g.UT * g.BT;
    return nb > 0 && nb < (1ll << 30) ? (int)nb : 0;
}

extern "C" int gdrn_pack_multi(const gdrn_pack_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, int dtype, void* stream) {
    if (!tasks_dev || !blk_start_dev || ntasks <= 0 || nblocks <= 0) return GDRN_ERR_ARG;
    if (dtype == GDRN_DT_F32) GDRN_LAUNCH(pack_multi_kernel<float>, dim3(nblocks), dim3(256), 0, ST, tasks_dev, blk_start_dev, ntasks);
    else if (dtype == GDRN_DT_H16) GDRN_LAUNCH(pack_multi_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, ST, tasks_dev, blk_start_dev, ntasks);
    else return GDRN_ERR_ARG;
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_unpack_multi(const gdrn_pack_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream) {
    if (!tasks_dev || !blk_start_dev || ntasks <= 0 || nblocks <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(unpack_multi_kernel, dim3(nblocks), dim3(256), 0, ST, tasks_dev, blk_start_dev, ntasks);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_zero_chunk(void) { return ZERO_CHUNK; }

// inf / NaN detector of the fp16 arithmetic mode's dynamic loss scale (GradScaler's found_inf, main_gdrn.py:53-56): one pass over the flat
// fp32 gradient buffer; a float4 with any all-ones exponent raises *flag (never cleared here: the host clears it after it has seen it)
namespace {
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, long long n, int* __restrict__ flag) {
    const long long n4 = n >> 2;
    unsigned bad = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + 4 * i);
        bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) | ((v.z & 0x7f800000u) == 0x7f800000u) |
               ((v.w & 0x7f800000u) == 0x7f800000u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) bad |= (__float_as_uint(x[(n4 << 2) + threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
}  // namespace

extern "C" int gdrn_nonfinite_flag(const float* x, long long n, int* flag, void* stream) {
    if (!x || !flag || n <= 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return GDRN_ERR_ARG;
    const int grid = (int)std::min<long long>(((n >> 2) + 255) / 256 + 1, 2048);
    GDRN_LAUNCH(nonfinite_flag_kernel, dim3(grid), dim3(256), 0, ST, x, n, flag);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_zero_multi(const gdrn_zero_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream) {
    if (!tasks_dev || !blk_start_dev || ntasks <= 0 || nblocks <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(zero_multi_kernel, dim3(nblocks), dim3(256), 0, ST, tasks_dev, blk_start_dev, ntasks);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_ranger_multi(const gdrn_ranger_task* tasks_dev, const int* row_start_dev, int ntasks, int total_rows, float beta1,
                                 float beta2, float eps, float weight_decay, float step_size, int adaptive, int lookahead, float alpha, float grad_scale,
                                 void* stream) {
    if (!tasks_dev || !row_start_dev || ntasks <= 0 || total_rows <= 0) return GDRN_ERR_ARG;
    GDRN_LAUNCH(ranger_multi_kernel, dim3(total_rows), dim3(256), 0, ST, tasks_dev, row_start_dev, ntasks, beta1, beta2, eps,
                       weight_decay, step_size, adaptive, lookahead, alpha, grad_scale, (const gdrn_loss_scale_state*)nullptr, 0, 1);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_ranger_multi_dyn(const gdrn_ranger_task* tasks_dev, const int* row_start_dev, int ntasks, int total_rows, float beta1,
                                     float beta2, float eps, float weight_decay, int n_sma_threshold, int lookahead_k, float alpha, float grad_scale,
                                     const gdrn_loss_scale_state* dyn, void* stream) {
    if (!tasks_dev || !row_start_dev || !dyn || ntasks <= 0 || total_rows <= 0 || lookahead_k < 1) return GDRN_ERR_ARG;
    GDRN_LAUNCH(ranger_multi_kernel, dim3(total_rows), dim3(256), 0, ST, tasks_dev, row_start_dev, ntasks, beta1, beta2, eps,
                       weight_decay, 0.f, 0, 0, alpha, grad_scale, dyn, n_sma_threshold, lookahead_k);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

namespace {
// the bookkeeping of torch.cuda.amp.GradScaler.update (growth_factor 2, backoff_factor 0.5) on the device, one thread: behind the optimizer
// launches of a step (applied_step = 1) or behind a backward pass whose gradients go to an external optimizer (applied_step = 0)
__global__ void loss_scale_update_kernel(gdrn_loss_scale_state* __restrict__ s, int applied_step) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (s->flag != 0) {
        s->scale = fmaxf(s->scale * 0.5f, 1.f);
        s->good = 0;
        s->skipped += 1;
        s->flag = 0;
        s->last_overflowed = 1;
    } else {
        s->applied += applied_step;
        s->good += 1;
        s->last_overflowed = 0;
        if (s->growth > 0 && s->good >= s->growth && s->scale < 65536.f) {
            s->scale *= 2.f;
            s->good = 0;
        }
    }
    s->inv_scale = 1.f / s->scale;
}

// gradients for an EXTERNAL optimizer (the autograd path, train_step(optimizer=None)): g *= factor / scale, or g = 0 when the step overflowed
// (GradScaler.unscale_ + the skipped step: a zero gradient instead of inf / NaN in .grad)
__global__ __launch_bounds__(256) void unscale_or_zero_kernel(float* __restrict__ g, long long n, float factor, const gdrn_loss_scale_state* __restrict__ s) {
    const float f = s->flag != 0 ? 0.f : factor * s->inv_scale;
    const bool zero = s->flag != 0;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = *reinterpret_cast<float4*>(g + 4 * i);
        v = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
        *reinterpret_cast<float4*>(g + 4 * i) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) g[(n4 << 2) + threadIdx.x] = zero ? 0.f : g[(n4 << 2) + threadIdx.x] * f;
}

// dL/dloss_k of the backward pass = w_k (* w2_k) * loss scale, the scale read from the device state
__global__ void scaled_loss_weights_kernel(const float* __restrict__ w, const float* __restrict__ w2, int n, const gdrn_loss_scale_state* __restrict__ s, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (i < n) out[i] = w[i] * (w2 ? w2[i] : 1.f) * s->scale;
}
}  // namespace

extern "C" int gdrn_loss_scale_update(gdrn_loss_scale_state* state, int applied_step, void* stream) {
    if (!state || (applied_step != 0 && applied_step != 1)) return GDRN_ERR_ARG;
    GDRN_LAUNCH(loss_scale_update_kernel, dim3(1), dim3(64), 0, ST, state, applied_step);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_unscale_or_zero(float* g, long long n, float factor, const gdrn_loss_scale_state* state, void* stream) {
    if (!g || !state || n <= 0 || (reinterpret_cast<uintptr_t>(g) & 15)) return GDRN_ERR_ARG;
    const int grid = (int)std::min<long long>(((n >> 2) + 255) / 256 + 1, 2048);
    GDRN_LAUNCH(unscale_or_zero_kernel, dim3(grid), dim3(256), 0, ST, g, n, factor, state);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

extern "C" int gdrn_scaled_loss_weights(const float* w, const float* w2, int n, const gdrn_loss_scale_state* state, float* out, void* stream) {
    if (!w || !state || !out || n <= 0 || n > 64) return GDRN_ERR_ARG;
    GDRN_LAUNCH(scaled_loss_weights_kernel, dim3(1), dim3(64), 0, ST, w, w2, n, state, out);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

// Weight-gradient of the gather convolution on MFMA for gfx950 (NHWC, fp32 accumulate):
//
//   dWp[co][tap][ci] += sum_{m in split} dY[m][co] * X[pix(m,tap)][ci]
//
// i.e. the autograd backward-weight of every Conv2d / ConvTranspose2d / Linear on the path
// (reference: implicit in `self.backward(losses)`, core/gdrn_modeling/engine.py:279).
// GEMM view: D[co][ci] = A[co][m] * B[m][ci] with the reduction index m (pixels) the *slow* axis of
// both NHWC operands, so both MFMA operands need a transposed fragment: bf16 uses the gfx950 LDS
// transpose read (ds_read_b64_tr_b16), fp32 one ds_read_b32 per lane (mfma 16x16x4 takes one value).
// The pixel range is split over blockIdx.z; partial tiles are accumulated with fp32 hardware atomics
// into the packed fp32 gradient (zeroed by the caller).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdlib>

#include "common.h"
#include "../../include/gdrn_hip.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;

template <typename T, int BCO, int BCI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const gdrn_wgrad_params p) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BKM = (ESZ == 2) ? 64 : 32;  // pixels per stage
    constexpr int PADB = (ESZ == 2) ? 32 : 64;
    constexpr int PA = BCO * ESZ + PADB, PB = BCI * ESZ + PADB;  // LDS row pitch (bytes)
    constexpr int SEGA = BCO * ESZ / 16, SEGB = BCI * ESZ / 16;  // 16B segments per row
    constexpr int LDA = BKM * SEGA / 256, LDB = BKM * SEGB / 256;
    constexpr int RSA = 256 / SEGA, RSB = 256 / SEGB;             // rows covered per pass
    constexpr int WCO = BCO / 2, WCI = BCI / 2;
    constexpr int FA = WCO / 16, FB = WCI / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                     // 2 x [BKM][PA]  (dY tile)
    unsigned char* sB = smem + 2 * BKM * PA;      // 2 x [BKM][PB]  (X tile)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int g = lane >> 4, r16 = lane & 15;

    // 1-D grid = splits x tiles, tile fastest, remapped so that each XCD (private L2; hardware puts
    // workgroup b on XCD b % 8) owns a contiguous run: all (co, tap, ci) tiles of one pixel range then
    // run on the same XCD at the same time and share its dY / X rows through that L2.
    const int KK = p.KH * p.KW;
    const int ncit = p.Cin / BCI;
    const int ncot = (p.Cout + BCO - 1) / BCO;
    const int ntile = ncot * ncit * KK;
    const int nsplit = (int)gridDim.x / ntile;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int split = bid / ntile;
    int tile = bid - split * ntile;
    const int cot = tile % ncot; tile /= ncot;
    const int cit = tile % ncit;
    const int tap = tile / ncit;
    const int co0 = cot * BCO, ci0 = cit * BCI;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;

    // pixel range of this split
    const int nstage_all = (p.M + BKM - 1) / BKM;
    const int per = (nstage_all + nsplit - 1) / nsplit;
    const int s_begin = split * per;
    const int s_end = min(nstage_all, s_begin + per);
    if (s_begin >= s_end) return;

    const int segA = tid % SEGA, rowA = tid / SEGA;
    const int segB = tid % SEGB, rowB = tid / SEGB;
    const char* dyg = reinterpret_cast<const char*>(p.dy);
    const char* xg = reinterpret_cast<const char*>(p.x);
    const int HW = p.Ho * p.Wo;
    // every feature map of the path has power-of-two sides: pixel -> (n, oy, ox) with shifts instead of two integer divisions
    // per gathered row and stage (the PMC pass of round 1 counted 11 VALU per MFMA in this kernel, most of them this index math)
    const bool pow2 = (p.Wo & (p.Wo - 1)) == 0 && (HW & (HW - 1)) == 0;
    const int lgW = 31 - __builtin_clz(p.Wo), lgHW = 31 - __builtin_clz(HW);

    uint4 ra[LDA], rb[LDB];
    auto load_stage = [&](int s) {
        const int m0 = s * BKM;
#pragma unroll
        for (int i = 0; i < LDA; ++i) {
            const int m = m0 + rowA + RSA * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < p.M) v = *reinterpret_cast<const uint4*>(dyg + ((size_t)m * p.dy_cs + co0) * ESZ + segA * 16);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < LDB; ++i) {
            const int m = m0 + rowB + RSB * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < p.M) {
                int n, oy, ox;
                if (pow2) {
                    n = m >> lgHW;
                    const int rem = m & (HW - 1);
                    oy = rem >> lgW;
                    ox = rem & (p.Wo - 1);
                } else {
                    n = m / HW;
                    const int rem = m - n * HW;
                    oy = rem / p.Wo;
                    ox = rem - oy * p.Wo;
                }
                const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                    v = *reinterpret_cast<const uint4*>(xg + ((size_t)((n * p.Hi + iy) * p.Wi + ix) * p.x_cs + ci0) * ESZ + segB * 16);
            }
            rb[i] = v;
        }
    };
    auto write_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LDA; ++i)
            *reinterpret_cast<uint4*>(sA + (size_t)buf * BKM * PA + (rowA + RSA * i) * PA + segA * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < LDB; ++i)
            *reinterpret_cast<uint4*>(sB + (size_t)buf * BKM * PB + (rowB + RSB * i) * PB + segB * 16) = rb[i];
    };

    f32x4_t acc[FA][FB];
#pragma unroll
    for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int b = 0; b < FB; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    load_stage(s_begin);
    write_stage(0);
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) load_stage(s + 1);
        const unsigned char* tA = sA + (size_t)buf * BKM * PA;
        const unsigned char* tB = sB + (size_t)buf * BKM * PB;
        if constexpr (ESZ == 2) {
#pragma unroll
            for (int ks = 0; ks < BKM / 32; ++ks) {
                bf16x8_t fa[FA], fb[FB];
                {
                    // ds_read_b64_tr_b16: lane q of a 16-lane group supplies the address of 4 contiguous
                    // bf16 of row (q>>2), columns (q&3)*4.. of a [4 m][16 ch] block and receives
                    // channel q for those 4 rows.  Lane group g takes reduction rows {4g..4g+3} with the
                    // first read and {16+4g..} with the second (the same map for both operands, so the
                    // k order is consistent): the two groups of a 32-lane half then touch 8 consecutive
                    // rows, which the 32-byte row padding spreads over all 64 banks (conflict-free).
                    const int trow = ks * 32 + g * 4 + (r16 >> 2), tcol = (r16 & 3) * 4;
#pragma unroll
                    for (int a = 0; a < FA; ++a) {
                        const unsigned char* q0 = tA + trow * PA + (wa * WCO + a * 16 + tcol) * 2;
                        bf16x4_t lo = GDRN_TR16((lds_bf16x4_t*)(q0));
                        bf16x4_t hi = GDRN_TR16((lds_bf16x4_t*)(q0 + 16 * PA));
                        fa[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int b = 0; b < FB; ++b) {
                        const unsigned char* q0 = tB + trow * PB + (wb * WCI + b * 16 + tcol) * 2;
                        bf16x4_t lo = GDRN_TR16((lds_bf16x4_t*)(q0));
                        bf16x4_t hi = GDRN_TR16((lds_bf16x4_t*)(q0 + 16 * PB));
                        fb[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
#pragma unroll
                for (int a = 0; a < FA; ++a)
#pragma unroll
                    for (int b = 0; b < FB; ++b)
                        acc[a][b] = GDRN_MFMA16(fa[a], fb[b], acc[a][b]);
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < BKM / 4; ++ks) {
                float fa[FA], fb[FB];
                const int mrow = ks * 4 + g;
#pragma unroll
                for (int a = 0; a < FA; ++a)
                    fa[a] = *reinterpret_cast<const float*>(tA + mrow * PA + (wa * WCO + a * 16 + r16) * 4);
#pragma unroll
                for (int b = 0; b < FB; ++b)
                    fb[b] = *reinterpret_cast<const float*>(tB + mrow * PB + (wb * WCI + b * 16 + r16) * 4);
#pragma unroll
                for (int a = 0; a < FA; ++a)
#pragma unroll
                    for (int b = 0; b < FB; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
        if (s + 1 < s_end) write_stage(buf ^ 1);
        __syncthreads();
    }

    // D[i = g*4+j (co)][col = r16 (ci)]
#pragma unroll
    for (int a = 0; a < FA; ++a) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + wa * WCO + a * 16 + g * 4 + j;
            if (co >= p.Cout) continue;
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int ci = ci0 + wb * WCI + b * 16 + r16;
                unsafeAtomicAdd(p.dw + ((size_t)co * KK + tap) * p.Cin + ci, acc[a][b][j]);
            }
        }
    }
}

template <typename T, int BCO, int BCI>
int launch(const gdrn_wgrad_params& p, hipStream_t st) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BKM = (ESZ == 2) ? 64 : 32;
    constexpr int PADB = (ESZ == 2) ? 32 : 64;
    constexpr size_t smem = 2 * (size_t)BKM * ((BCO * ESZ + PADB) + (BCI * ESZ + PADB));
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<T, BCO, BCI>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GDRN_ERR_LAUNCH;
        attr_set = true;
    }
    const int tiles = cdiv(p.Cout, BCO) * (p.Cin / BCI) * p.KH * p.KW;
    int splits = p.splits;
    if (splits <= 0) {
        const int nst = cdiv(p.M, BKM);
        // every split adds one fp32 atomic per output element: keep >= 8 stages per block; ~2048 workgroups measured best
        // (256: +1.8 % step time, 1024: +0.4 %, 4096: equal) -- these small layers want parallelism more than few atomics
        constexpr int target = 2048;
        splits = max(1, min(nst / 8 > 0 ? nst / 8 : 1, cdiv(target, tiles)));
    }
    GDRN_LAUNCH((conv_wgrad_kernel<T, BCO, BCI>), dim3(tiles * splits), dim3(256), smem, st, p);
    GDRN_CHECK_LAUNCH();
    return GDRN_OK;
}

}  // namespace

extern "C" int gdrn_conv_wgrad(const gdrn_wgrad_params* pp, void* stream) {
    if (!pp || !pp->x || !pp->dy || !pp->dw) return GDRN_ERR_ARG;
    const gdrn_wgrad_params& p = *pp;
    if (p.dtype != GDRN_DT_F32 && p.dtype != GDRN_DT_H16) return GDRN_ERR_ARG;
    const int esz = p.dtype == GDRN_DT_H16 ? 2 : 4;
    if (p.Cin <= 0 || p.Cin % 64 != 0 || p.M <= 0 || p.Cout <= 0) return GDRN_ERR_SHAPE;
    if ((p.x_cs * esz) % 16 != 0 && (p.x_cs * esz) % 8 != 0) return GDRN_ERR_SHAPE;
    const int bco = p.Cout <= 64 ? 64 : 128;
    const int bci = (p.Cin % 128 == 0) ? 128 : 64;
    // dY rows are read in BCO-wide slabs: the caller guarantees dy_cs >= ceil(Cout/BCO)*BCO readable columns
    if (p.dy_cs < cdiv(p.Cout, bco) * bco) return GDRN_ERR_SHAPE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (p.dtype == GDRN_DT_H16) {
        if (bco == 64) return bci == 64 ? launch<bf16_t, 64, 64>(p, st) : launch<bf16_t, 64, 128>(p, st);
        return bci == 64 ? launch<bf16_t, 128, 64>(p, st) : launch<bf16_t, 128, 128>(p, st);
    } else {
        if (bco == 64) return bci == 64 ? launch<float, 64, 64>(p, st) : launch<float, 64, 128>(p, st);
        return bci == 64 ? launch<float, 128, 64>(p, st) : launch<float, 128, 128>(p, st);
    }
}

// Scratch-size queries of the C ABI (host-only code): what a non-Python host needs to allocate before it can call the entry
// points that take a caller-owned workspace -- SURVEY.md section 8(b) `gdrn_workspace_bytes`.  The Python engine sizes the same
// buffers with the same per-op helper queries (gdr-net_amd/engine.py::Plan._build).
#include "common.h"
#include "../../include/gdrn_hip.h"

extern "C" long long gdrn_workspace_bytes(int op, const void* params) {
    if (!params) return GDRN_ERR_ARG;
    switch (op) {
        case GDRN_WS_CONV_STATS: {  // gdrn_conv_params*: p->stats of gdrn_conv_gemm
            const gdrn_conv_params* p = static_cast<const gdrn_conv_params*>(params);
            const int rows = gdrn_conv_stats_rows(p);
            return rows < 0 ? rows : (long long)rows * 2 * p->Cout * (long long)sizeof(float);
        }
        case GDRN_WS_CONV3X3_STATS: {  // gdrn_conv_params*: p->stats / p->bnb_rows of gdrn_conv3x3_halo
            const gdrn_conv_params* p = static_cast<const gdrn_conv_params*>(params);
            const int rows = gdrn_conv3x3_stats_rows(p);
            return rows < 0 ? rows : (long long)rows * 2 * p->Cout * (long long)sizeof(float);
        }
        case GDRN_WS_CONV3X3_WGRAD: {  // gdrn_wgrad_params* (splits = requested count, 0 = automatic): p->ws of gdrn_conv3x3_wgrad[_multi]
            gdrn_wgrad_params q = *static_cast<const gdrn_wgrad_params*>(params);
            float dummy;
            q.ws = &dummy;  // non-null: the workspace-mode split count
            const int splits = gdrn_conv3x3_wgrad_splits(&q);
            return splits < 0 ? splits : (long long)splits * q.Cout * q.Cin * 9 * (long long)sizeof(float);
        }
        case GDRN_WS_STEM_WGRAD: {  // const int* N: ws of gdrn_stem_wgrad
            const int N = *static_cast<const int*>(params);
            return N <= 0 ? GDRN_ERR_ARG : (long long)gdrn_stem_wgrad_parts(N) * 64 * 224 * (long long)sizeof(float);
        }
        case GDRN_WS_STEM_STATS: {  // const int* N: stats of gdrn_stem_conv
            const int N = *static_cast<const int*>(params);
            return N <= 0 ? GDRN_ERR_ARG : (long long)gdrn_stem_stats_rows(N) * 2 * 64 * (long long)sizeof(float);
        }
        case GDRN_WS_LINEAR_SPLITK: {  // const int[2] = {M, N}: ws of gdrn_linear_splitk
            const int* mn = static_cast<const int*>(params);
            return (mn[0] <= 0 || mn[1] <= 0) ? GDRN_ERR_ARG : ((long long)GDRN_LINEAR_MAX_SPLITS * mn[0] * mn[1] + 64) * (long long)sizeof(float);
        }
        case GDRN_WS_BN_BWD_ROWS: {  // const long long[3] = {npix, C, dtype}: rows of gdrn_bn_bwd_reduce
            const long long* a = static_cast<const long long*>(params);
            const int rows = gdrn_bn_bwd_reduce_rows(a[0], (int)a[1], (int)a[2]);
            return rows < 0 ? rows : (long long)rows * 2 * a[1] * (long long)sizeof(float);
        }
        default:
            return GDRN_ERR_ARG;
    }
}

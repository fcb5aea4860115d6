/* Pure-C client of include/gdrn_hip.h (tests/test_host_cpu.py::test_c_client_compiles_links_and_runs): the header is C, its enums
 * and structs are what the comments say, and the library answers the host-only queries without a GPU. */
#include <stdio.h>
#include <string.h>

#include "gdrn_hip.h"

int main(void) {
    gdrn_conv_params p;
    gdrn_wgrad_params w;
    long long rows[3] = {64LL * 64 * 64, 256, GDRN_DT_BF16};
    memset(&p, 0, sizeof p);
    memset(&w, 0, sizeof w);
    p.Hi = p.Wi = p.Ho = p.Wo = 64;
    p.Cin = p.Cout = p.x_cs = p.y_cs = 256;
    p.KH = p.KW = 3;
    p.stride = p.pad = 1;
    p.M = 64 * 64 * 64;
    p.w_rows = 256;
    p.dtype = GDRN_DT_BF16;
    p.xf_mode = 1;
    if (gdrn_version() != GDRN_ABI_VERSION) return 1;
    if (gdrn_conv3x3_wfrag(&p) != 2) return 2;            /* 256-channel tile of the second-generation kernel */
    p.w_frag = 2;
    if (gdrn_conv3x3_stats_rows(&p) != 64 * 4 * 4) return 3;  /* one row per 16x16 pixel tile */
    if (gdrn_workspace_bytes(GDRN_WS_CONV3X3_STATS, &p) != (long long)64 * 16 * 2 * 256 * 4) return 4;
    if (gdrn_workspace_bytes(GDRN_WS_BN_BWD_ROWS, rows) <= 0) return 5;
    if (gdrn_conv3x3_halo(NULL, NULL) != GDRN_ERR_ARG) return 6;
    if (GDRN_OK != 0 || GDRN_ERR_SHAPE != -2 || GDRN_ERR_LAUNCH != -3 || GDRN_DT_F32 != 0) return 7;
    printf("gdrn C client ok: abi %d, conv params %zu bytes, wgrad params %zu bytes\n", gdrn_version(), sizeof p, sizeof w);
    return 0;
}

/* A plain-C consumer of libpnpmri.so: proves the boundary is a C ABI (no C++/torch types).
 * Built with gcc against include/pnp_mri.h by tests/test_abi_cpu.py (argument/error paths, no GPU)
 * and by tests/test_gpu_parity.py (a tiny ADMM_L1 run whose checksum is compared with the
 * Python binding's).   usage: abi_consumer [run]                                              */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pnp_mri.h"

int main(int argc, char** argv) {
    pnp_ctx* ctx = NULL;
    int n = -1;
    if (pnp_abi_version() != PNP_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 10; }
    if (pnp_device_count(&n) != PNP_OK || n < 0) return 11;
    if (pnp_ctx_create(0, 300, 256, 1, &ctx) != PNP_E_ARG || ctx != NULL) return 12;
    if (strstr(pnp_last_error(), "256 or 512") == NULL) return 13;
    if (pnp_init_state(NULL) != PNP_E_ARG) return 14;
    if (pnp_ctx_destroy(NULL) != PNP_OK) return 15;
    printf("devices %d\n", n);
    if (argc < 2 || strcmp(argv[1], "run") != 0) return 0;

    /* 2 slices: y = k-space of a centred box image sampled on every other row, 5 L1 iterations */
    const int B = 2, H = 256, W = 256, N = H * W;
    float* img = (float*)calloc((size_t)B * N, sizeof(float));
    float* noise = (float*)calloc((size_t)2 * N, sizeof(float));       /* one shared [H][W] complex array */
    uint8_t* mask = (uint8_t*)calloc((size_t)N, 1);
    float* x = (float*)malloc((size_t)B * N * sizeof(float));
    for (int b = 0; b < B; ++b)
        for (int r = 96; r < 160; ++r)
            for (int c = 64 + 16 * b; c < 192; ++c) img[(size_t)b * N + r * W + c] = 0.5f + 0.25f * b;
    for (int r = 0; r < H; ++r)
        if (r % 2 == 0 || r < 8 || r > H - 8)
            for (int c = 0; c < W; ++c) mask[r * W + c] = 1;
    int rc = pnp_ctx_create(0, H, W, B, &ctx);
    if (rc) { fprintf(stderr, "create: %s\n", pnp_last_error()); return 20; }
    rc = pnp_synthesize_problem(ctx, img, noise, 0, mask, NULL, B, 1, 0);
    if (!rc) rc = pnp_init_state(ctx);
    if (!rc) rc = pnp_admm_l1_run(ctx, 5, 0.1, 0.015);
    if (!rc) rc = pnp_download_x(ctx, x, 0);
    if (rc) { fprintf(stderr, "run: %s\n", pnp_last_error()); return 21; }
    double s = 0.0;
    for (size_t i = 0; i < (size_t)B * N; ++i) s += x[i];
    printf("path %s checksum %.9e\n", pnp_path_name(ctx), s);
    pnp_ctx_destroy(ctx);
    free(img); free(noise); free(mask); free(x);
    return 0;
}

/*
 * examples/c_host.c -- a plain-C host of libumpa_hip.so: no Python, no PyTorch, only include/umpa_hip.h.
 *
 * Builds a small synthetic stack (sample = 0.8 * reference shifted by (+1 row, -1 column)), matches it with
 * the dark-field model and prints how many pixels recovered that shift.  This is what a C/C++/cgo/JNI host
 * of the reference's native layer (UMPA/lib/Model.h) would do through the C ABI.
 *
 *   gcc -O2 -Iinclude examples/c_host.c -o examples/c_host -Lumpa_amd -lumpa_hip -Wl,-rpath,$PWD/umpa_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "umpa_hip.h"

static double noise(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (double)(*s >> 8) / 16777216.0 - 0.5; }

int main(void)
{
    enum { K = 4, H = 96, W = 128, NW = 3, MS = 4, PAD = NW + MS, S = 2 * NW + 1 };
    const int N0 = H - 2 * PAD, N1 = W - 2 * PAD, dy = 1, dx = -1;
    double *ref[K], *sam[K], win[S * S], wsum = 0.0;
    int dims[2 * K], pos[2 * K];
    unsigned seed = 12345u;

    for (int k = 0; k < K; k++) {
        ref[k] = malloc(sizeof(double) * H * W);
        sam[k] = malloc(sizeof(double) * H * W);
        double *tmp = malloc(sizeof(double) * H * W);
        for (int n = 0; n < H * W; n++) tmp[n] = noise(&seed);
        for (int pass = 0; pass < 2; pass++) {            /* 5x5 box blur, twice: a smooth speckle-like pattern */
            for (int i = 0; i < H; i++)
                for (int j = 0; j < W; j++) {
                    double a = 0.0;
                    for (int p = -2; p <= 2; p++)
                        for (int q = -2; q <= 2; q++) a += tmp[((i + p + H) % H) * W + (j + q + W) % W];
                    ref[k][i * W + j] = a / 25.0;
                }
            for (int n = 0; n < H * W; n++) tmp[n] = ref[k][n];
        }
        for (int n = 0; n < H * W; n++) ref[k][n] = 1.0 + 8.0 * tmp[n];
        for (int i = 0; i < H; i++)                       /* sam[i][j] = 0.8 * ref[i+dy][j+dx] */
            for (int j = 0; j < W; j++) sam[k][i * W + j] = 0.8 * ref[k][((i + dy + H) % H) * W + (j + dx + W) % W];
        free(tmp);
        dims[2 * k] = H; dims[2 * k + 1] = W; pos[2 * k] = pos[2 * k + 1] = 0;
    }
    for (int a = 0; a < S; a++)                           /* outer(hamming, hamming) / sum, model.pyx:691-696 */
        for (int b = 0; b < S; b++) {
            win[a * S + b] = (0.54 - 0.46 * cos(2 * M_PI * a / (S - 1))) * (0.54 - 0.46 * cos(2 * M_PI * b / (S - 1)));
            wsum += win[a * S + b];
        }
    for (int n = 0; n < S * S; n++) win[n] /= wsum;

    if (umpa_hip_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
    umpa_hip_model *m = umpa_hip_create(UMPA_HIP_KIND_DF, K, dims, sam, ref, NULL, pos, NW, win, MS, PAD, 0, 0);
    if (!m) { fprintf(stderr, "create: %s\n", umpa_hip_last_error()); return 1; }
    umpa_hip_set_subpx(m, 0);                             /* report the integer minimum */

    double *values = calloc((size_t)N0 * N1 * 5, sizeof(double));
    int *err = calloc((size_t)N0 * N1, sizeof(int));
    int rc = umpa_hip_match_region(m, 0, 1, N0, 0, 1, N1, values, 5, NULL, err, NULL, 0.0, NULL, NULL, NULL, 0, NULL);
    if (rc < 0) { fprintf(stderr, "match: %s\n", umpa_hip_last_error()); return 1; }

    int good = 0, inner = 0;
    double Tsum = 0.0;
    for (int i = 8; i < N0 - 8; i++)
        for (int j = 8; j < N1 - 8; j++) {
            const double *v = values + ((size_t)i * N1 + j) * 5;     /* [f, T, dx, dy, df] */
            inner++;
            if (err[i * N1 + j] == 1 && v[2] == dx && v[3] == dy) { good++; Tsum += v[1]; }
        }
    printf("c_host: %d of %d inner pixels recovered (dy,dx)=(%d,%d); mean T = %.6f; path = %d\n",
           good, inner, dy, dx, good ? Tsum / good : 0.0, umpa_hip_last_path(m));
    umpa_hip_destroy(m);
    return (good >= inner - inner / 500 && fabs(Tsum / good - 0.8) < 1e-2) ? 0 : 3;   /* a stray local minimum is allowed */
}

/*
 * oracle/umpa_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, fp64, CPU restatement of the UMPA per-pixel matching path, written
 * from the behavioural description in SURVEY.md section 8(a); every function cites the
 * reference lines it follows (paths relative to /root/reference/UMPA).  It is the
 * checker for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product library (umpa_amd/csrc) never links,
 * calls or falls back to anything in this directory.
 *
 * Parity status: PINNED.  The reference's own tests pin no numbers,
 * so the oracle is pinned against (a) golden vectors frozen from the unmodified
 * reference imported in the build container (tests/golden/, generator
 * tests/golden/make_golden.py) and (b) oracle/_ref/libumpa_ref.so, the reference
 * C++ core compiled in place (oracle/ref_shim.cpp), see tests/test_oracle_*.py.
 *
 * Build: see oracle/Makefile  ->  oracle/libumpa_oracle.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ST_OK        1
#define ST_BOUND     2
#define ST_DIM       4
#define ST_POSITIVE  8

#define KIND_NODF     0
#define KIND_DF       1
#define KIND_DFKERNEL 2

#define BLUR_HALF 8                               /* lib/Model.h:7  KERNEL_WINDOW_SIZE */
#define BLUR_SIDE (2 * BLUR_HALF + 1)
static int CALL_CAP = 500;                        /* lib/Optim.cpp:14 MAX_CALLS; UMPA_CALL_CAP in the environment lowers it (tests) */

typedef struct {
    int kind, Na, Nw, max_shift, padding;
    int subpx;                                    /* -1 spline (default), 0 none, 1 paraboloid */
    int ref_mode;                                 /* 0: window fixed on sample ('sam'), 1: on reference */
    int has_mask;
    int *dim, *pos;                               /* [Na][2] */
    const double **sam, **ref, **mask;
    double *win;                                  /* (2Nw+1)^2 */
} oracle_model;

/* the "args" payload the minimiser carries around (lib/Model.h:29-71) */
typedef struct { double t, v; } fit_t;

/* ------------------------------------------------------------------ helpers */

/* lib/Utils.cpp:125-130 : weight of a pixel pair from the two mask values */
static inline double pair_weight(double a, double b) { return a * b / (a + b + 1e-8); }

/* lib/Model.cpp:285-288 (and :430-433, :716-719, ...): does frame k contribute at (i,j)? */
static inline int frame_covers(const oracle_model *m, int k, int i, int j)
{
    int li = i - m->pos[2 * k], lj = j - m->pos[2 * k + 1];
    return li - m->padding >= 0 && li + m->padding <= m->dim[2 * k] &&
           lj - m->padding >= 0 && lj + m->padding <= m->dim[2 * k + 1];
}

/* lib/Utils.cpp:85-97 (unweighted) and :103-117 (mask-weighted) blur of one ref pixel */
static double blur_at(const double *img, const double *wgt, int W, int i, int j, const double *kern)
{
    double acc = 0.0, wsum = 0.0;
    for (int a = -BLUR_HALF; a <= BLUR_HALF; a++) {
        const double *row = img + (size_t)(i + a) * W + j;
        const double *krow = kern + (a + BLUR_HALF) * BLUR_SIDE + BLUR_HALF;
        if (!wgt) {
            for (int b = -BLUR_HALF; b <= BLUR_HALF; b++) acc += krow[b] * row[b];
        } else {
            const double *wrow = wgt + (size_t)(i + a) * W + j;
            for (int b = -BLUR_HALF; b <= BLUR_HALF; b++) {
                acc += krow[b] * row[b] * wrow[b];
                wsum += krow[b] * wrow[b];
            }
        }
    }
    return wgt ? acc / wsum : acc;
}

/* lib/Model.cpp:88-117 : normalised exp(-a i^2 - b i j - c j^2) on a 17x17 grid */
static void build_blur_kernel(double a, double b, double c, double *kern)
{
    double norm = 0.0;
    for (int i = -BLUR_HALF; i <= BLUR_HALF; i++)
        for (int j = -BLUR_HALF; j <= BLUR_HALF; j++) {
            double g = exp(-a * i * i - b * i * j - c * j * j);     /* lib/Utils.cpp:46-50 */
            kern[(i + BLUR_HALF) * BLUR_SIDE + j + BLUR_HALF] = g;
            norm += g;
        }
    for (int n = 0; n < BLUR_SIDE * BLUR_SIDE; n++) kern[n] /= norm;
}

/* ------------------------------------------------------------------ cost
 * One evaluation of the windowed multi-frame least-squares cost at pixel (i,j) and
 * integer shift (si = rows, sj = columns), analytically minimised over the
 * transmission (and dark-field).  Follows lib/Model.cpp:359-509 (NoDF),
 * :631-862 (DF), :997-1151 (DFKernel).  On a bound error nothing is written.
 */
static int eval_cost(const oracle_model *m, int i, int j, int si, int sj,
                     const double *kern, double *cost, fit_t *fit)
{
    const int ms = m->max_shift, Nw = m->Nw, S = 2 * Nw + 1;
    /* guards, lib/Model.cpp:372-399 / :654-681 / :1011-1038 (note the asymmetric flags) */
    if (si <= -ms || si >= ms) return ST_BOUND;
    if (sj <= -ms) return ST_BOUND | ST_DIM;
    if (sj >= ms) return ST_BOUND | ST_DIM | ST_POSITIVE;

    /* window centres, lib/Model.cpp:408-421 / :688-701 */
    int ri = i, rj = j, qi = i, qj = j;                 /* r*: reference, q*: sample */
    if (m->ref_mode) { qi -= si; qj -= sj; } else { ri += si; rj += sj; }

    double t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
    double wt = m->has_mask ? 0.0 : (double)m->Na;      /* :425/:711 vs :463/:777 */

    for (int k = 0; k < m->Na; k++) {
        if (!frame_covers(m, k, i, j)) continue;
        const int W = m->dim[2 * k + 1];
        const int pi = m->pos[2 * k], pj = m->pos[2 * k + 1];
        const double *R = m->ref[k] + (size_t)(ri - pi - Nw) * W + (rj - pj - Nw);
        const double *Q = m->sam[k] + (size_t)(qi - pi - Nw) * W + (qj - pj - Nw);
        const double *MR = m->has_mask ? m->mask[k] + (size_t)(ri - pi - Nw) * W + (rj - pj - Nw) : NULL;
        const double *MQ = m->has_mask ? m->mask[k] + (size_t)(qi - pi - Nw) * W + (qj - pj - Nw) : NULL;

        double mean = 0.0;
        if (m->kind == KIND_DF) {                       /* pass 1, :723-739 / :789-808 (never mask-weighted) */
            double den = 0.0;
            for (int a = 0; a < S; a++)
                for (int b = 0; b < S; b++) {
                    double w = m->win[a * S + b];
                    mean += w * R[(size_t)a * W + b];
                    den += w;
                }
            mean /= den;
        }

        double s2 = 0, s4 = 0, s6 = 0;                  /* per-frame partial sums */
        for (int a = 0; a < S; a++)
            for (int b = 0; b < S; b++) {
                double w = m->win[a * S + b];
                double q = Q[(size_t)a * W + b];
                double r;
                if (m->kind == KIND_DFKERNEL)           /* :1088-1090 / :1130-1132 */
                    r = blur_at(m->ref[k], m->has_mask ? m->mask[k] : NULL, W,
                                ri - pi - Nw + a, rj - pj - Nw + b, kern);
                else
                    r = R[(size_t)a * W + b];
                if (m->has_mask) {                      /* :488-495 / :830-840 / :1127-1138 */
                    w *= pair_weight(MR[(size_t)a * W + b], MQ[(size_t)a * W + b]);
                    wt += w;
                }
                t1 += w * q * q;
                t3 += w * r * r;
                t5 += w * r * q;
                s2 += w; s4 += w * q; s6 += w * r;
            }
        if (m->kind == KIND_DF) {                       /* :770-772 / :843-845 */
            t2 += m->has_mask ? mean * mean * s2 : mean * mean;
            t4 += mean * s4;
            t6 += mean * s6;
        }
    }

    if (m->kind == KIND_DF) {                           /* :849-858 */
        double det = t2 * t3 - t6 * t6;
        double K = (t2 * t5 - t4 * t6) / det;
        double beta = (t3 * t4 - t5 * t6) / det;
        fit->t = beta + K;
        fit->v = K / fit->t;
        *cost = (t1 + beta * beta * t2 + K * K * t3 - 2 * beta * t4 - 2 * K * t5 + 2 * beta * K * t6) / wt;
    } else {                                            /* :502-505 / :1144-1147 */
        fit->t = t5 / t3;
        *cost = (t1 - t5 * fit->t) / wt;
    }
    return ST_OK;
}

/* ------------------------------------------------------------------ sub-pixel fits */

/* Uniform cubic B-spline basis, control points at -1,0,1,2: B[m][p] is the
 * coefficient of t^p of basis function m, times 6 (cf. utils.py:151-154). */
static const double BSPL[4][4] = {{1, -3, 3, -1}, {4, 0, -6, 3}, {1, 3, 3, -3}, {0, 0, 0, 1}};

/* lib/Optim.cpp:41-130 : minimum of the bicubic B-spline through a[4*row+col] by
 * unclamped Newton-Raphson (<= 21 steps, stop when |step|^2 < 1e-8); pos[0] is the
 * row coordinate.  Returns the spline value there.  c[4q+p] multiplies x^p y^q. */
double umpaor_spmin(const double *a, double *pos)
{
    double c[16];
    for (int q = 0; q < 4; q++)
        for (int p = 0; p < 4; p++) {
            double acc = 0.0;
            for (int r = 0; r < 4; r++)
                for (int s = 0; s < 4; s++) acc += a[4 * r + s] * BSPL[r][p] * BSPL[s][q];
            c[4 * q + p] = acc;
        }
    double x = pos[0], y = pos[1];
    double g[4], g1[4], g2[4];
    for (int it = 0; it <= 20; it++) {
        for (int q = 0; q < 4; q++) {
            const double *cq = c + 4 * q;
            g1[q] = cq[1] + x * (2 * cq[2] + x * 3 * cq[3]);
            g2[q] = 2 * cq[2] + 6 * cq[3] * x;
            g[q] = cq[0] + x * (cq[1] + x * (cq[2] + x * cq[3]));
        }
        double fx = g1[0] + y * (g1[1] + y * (g1[2] + y * g1[3]));
        double fxx = g2[0] + y * (g2[1] + y * (g2[2] + y * g2[3]));
        double fy = g[1] + y * (2 * g[2] + y * 3 * g[3]);
        double fxy = g1[1] + y * (2 * g1[2] + y * 3 * g1[3]);
        double fyy = 2 * g[2] + 6 * g[3] * y;
        double det = fxx * fyy - fxy * fxy;            /* no definiteness / zero check, :103 */
        double dx = (fxy * fy - fyy * fx) / det;
        double dy = (fxy * fx - fxx * fy) / det;
        x += dx; y += dy;
        if (dx * dx + dy * dy < 1e-8) break;
    }
    pos[0] = x; pos[1] = y;
    for (int q = 0; q < 4; q++) {
        const double *cq = c + 4 * q;
        g[q] = cq[0] + x * (cq[1] + x * (cq[2] + x * cq[3]));
    }
    return (g[0] + y * (g[1] + y * (g[2] + y * g[3]))) / 36.0;
}

/* lib/Optim.cpp:155-185 : least-squares paraboloid c0 + gi*i + gj*j + cii*i^2 + cij*i*j + cjj*j^2
 * through the 16 values on {-1,0,1,2}^2.  The reference tabulates 400 * pinv(design) with
 * integer entries (:169-174); here it is derived once from the design matrix.  The vertex
 * formula pairs (cii, gj) and (cjj, gi) exactly as the reference does (:180-181). */
static double QUAD[6][16];
static int quad_ready = 0;
static void quad_init(void)
{
    double A[16][6], N[6][12];
    const int g[4] = {-1, 0, 1, 2};
    for (int r = 0; r < 4; r++)
        for (int s = 0; s < 4; s++) {
            double *row = A[4 * r + s];
            double i = g[r], j = g[s];
            row[0] = 1; row[1] = i; row[2] = j; row[3] = i * i; row[4] = i * j; row[5] = j * j;
        }
    for (int p = 0; p < 6; p++)
        for (int q = 0; q < 6; q++) {
            N[p][q] = 0; N[p][6 + q] = (p == q);
            for (int n = 0; n < 16; n++) N[p][q] += A[n][p] * A[n][q];
        }
    for (int p = 0; p < 6; p++) {                       /* Gauss-Jordan with partial pivoting */
        int best = p;
        for (int r = p + 1; r < 6; r++) if (fabs(N[r][p]) > fabs(N[best][p])) best = r;
        for (int q = 0; q < 12; q++) { double t = N[p][q]; N[p][q] = N[best][q]; N[best][q] = t; }
        double piv = N[p][p];
        for (int q = 0; q < 12; q++) N[p][q] /= piv;
        for (int r = 0; r < 6; r++) if (r != p) {
            double f = N[r][p];
            for (int q = 0; q < 12; q++) N[r][q] -= f * N[p][q];
        }
    }
    for (int p = 0; p < 6; p++)
        for (int n = 0; n < 16; n++) {
            double acc = 0;
            for (int q = 0; q < 6; q++) acc += N[p][6 + q] * A[n][q];
            QUAD[p][n] = nearbyint(400.0 * acc);        /* exact integers, :169-174 */
        }
    quad_ready = 1;
}

double umpaor_spmin_quad(const double *a, double *pos)
{
    if (!quad_ready) quad_init();
    double p[6];
    for (int n = 0; n < 6; n++) {
        double acc = 0;
        for (int k = 0; k < 16; k++) acc += QUAD[n][k] * a[k];
        p[n] = acc;
    }
    double det = 4 * p[3] * p[5] - p[4] * p[4];
    pos[0] = -(2 * p[3] * p[2] - p[4] * p[1]) / det;
    pos[1] = -(2 * p[5] * p[1] - p[4] * p[2]) / det;
    return (p[0] + 0.5 * (p[2] * pos[0] + p[1] * pos[1])) / 400.0;
}

/* ------------------------------------------------------------------ minimiser
 * Greedy axis-alternating integer descent with a 5x5 memo, hard restart, 4x4
 * gather and sub-pixel refinement: lib/Optim.cpp:233-479.  `memo` (25) and
 * `nb` (16) are the debug arrays d / a of lib/Optim.h:15-21; unknown memo
 * cells hold -1.  `live` plays *args, `kept` plays args_copy.
 */
#define CENTRE 12
static const double TIE = 1e-8;                        /* lib/Optim.cpp:243 */

static void memo_shift(double *d, int axis, int dir)
{
    /* axis 0: columns, axis 1: rows; dir +1 moves the centre to larger index (:436-470) */
    double n[25];
    for (int r = 0; r < 5; r++)
        for (int c = 0; c < 5; c++) {
            int sr = r + (axis ? dir : 0), sc = c + (axis ? 0 : dir);
            n[5 * r + c] = (sr < 0 || sr > 4 || sc < 0 || sc > 4) ? -1.0 : d[5 * sr + sc];
        }
    memcpy(d, n, sizeof(n));
}

static int minimise(const oracle_model *m, int i, int j, const double *kern,
                    double *out, double *uv, fit_t *live, double *memo, double *nb, int *ncalls)
{
    int ci = (int)round(uv[0]), cj = (int)round(uv[1]);         /* :258-259 */
    int n = 0, st, axis = 0, found[2] = {0, 0};
    int moves = 0, n_at_move = 0;                               /* centre moves since the last cost call, see MOVE_CAP below */
    fit_t kept;
    for (int q = 0; q < 25; q++) memo[q] = -1.0;                /* :252 */

    st = eval_cost(m, i, j, ci, cj, kern, &memo[CENTRE], live); /* :262-265 */
    *ncalls = ++n;
    if (!(st & ST_OK)) return st;
    kept = *live;

    while (n < CALL_CAP) {                                      /* :267 */
    rescan:;
        const int lo = axis ? CENTRE - 5 : CENTRE - 1;          /* south / west  (:271-284) */
        const int hi = axis ? CENTRE + 5 : CENTRE + 1;          /* north / east  (:304-317) */
        int lo_up, hi_up;

        if (memo[lo] < -0.5) {                                  /* :287-301 */
            st = eval_cost(m, i, j, ci - (axis ? 1 : 0), cj - (axis ? 0 : 1), kern, &memo[lo], live);
            *ncalls = ++n;
            if (!(st & ST_OK)) return st;
            lo_up = memo[lo] > memo[CENTRE] + TIE;
            if (!lo_up) kept = *live;
        } else
            lo_up = memo[lo] > memo[CENTRE] + TIE;

        if (memo[hi] < -0.5) {                                  /* :320-332 */
            st = eval_cost(m, i, j, ci + (axis ? 1 : 0), cj + (axis ? 0 : 1), kern, &memo[hi], live);
            *ncalls = ++n;
            if (!(st & ST_OK)) return st;
            hi_up = memo[hi] > memo[CENTRE] - TIE;
            if (!hi_up) kept = *live;
        } else
            hi_up = memo[hi] > memo[CENTRE] - TIE;

        if (lo_up && hi_up) {                                   /* :334-418 */
            found[axis] = memo[lo] < memo[hi] ? -1 : 1;
            if (!found[1 - axis]) { axis = 1 - axis; continue; }

            /* quadrant for the 4x4 neighbourhood, :344-345 */
            const int ip = memo[CENTRE + 5] < memo[CENTRE - 5] ? 1 : 0;
            const int jp = memo[CENTRE + 1] < memo[CENTRE - 1] ? 1 : 0;
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) {
                    double *cell = &memo[5 * (ip + r) + jp + c];
                    if (*cell < -0.9) {                         /* :353-378 */
                        int ei = ci + ip + r - 2, ej = cj + jp + c - 2;
                        st = eval_cost(m, i, j, ei, ej, kern, &nb[4 * r + c], live);
                        *ncalls = ++n;
                        if (!(st & ST_OK)) return st;
                        *cell = nb[4 * r + c];
                        if (nb[4 * r + c] < memo[CENTRE]) {     /* missed a lower value: hard restart */
                            double vnew = nb[4 * r + c];
                            ci = ei; cj = ej;
                            for (int q = 0; q < 25; q++) memo[q] = -1.0;
                            memo[CENTRE] = vnew;
                            *live = kept;                       /* stale on purpose, :373 */
                            found[0] = found[1] = 0;
                            goto rescan;                        /* skips the call-cap test, :376 */
                        }
                    } else
                        nb[4 * r + c] = *cell;
                }
            *live = kept;                                       /* :386 */
            uv[0] = 1.0 - ip; uv[1] = 1.0 - jp;                 /* :395-396 */
            if (m->subpx == 0) *out = uv[0];                    /* :399 */
            else if (m->subpx == 1) *out = umpaor_spmin_quad(nb, uv);
            else *out = umpaor_spmin(nb, uv);
            uv[0] += ci + ip - 1.0;                             /* :407-408 */
            uv[1] += cj + jp - 1.0;
            return st;
        }

        uv[0] = ci; uv[1] = cj; *out = memo[CENTRE];            /* best so far, :421-423 */
        if (!hi_up && !lo_up) lo_up = memo[hi] < memo[lo];      /* both lower: go to the lower one */
        const int dir = lo_up ? 1 : -1;                         /* :431-474 */
        if (axis) ci += dir; else cj += dir;
        memo_shift(memo, axis, dir);
        found[1 - axis] = 0;
        /* NOT in the reference, which never returns from this: with a NaN cost in the neighbourhood (a masked window without
           a valid pixel pair) the centre can step between two memoised cells for ever, no call is made and the call cap is
           never reached.  64 moves in a row without a cost call end the walk like the cap does (umpa_walk.h: UMPA_MOVE_CAP,
           the same rule); walks on finite costs never get there. */
        if (n != n_at_move) { n_at_move = n; moves = 0; }
        if (++moves > 64) return st & ~ST_OK;
    }
    return st & ~ST_OK;                                         /* :477 */
}

/* ------------------------------------------------------------------ C interface */

void *umpaor_create(int kind, int Na, const int *dims, double *const *sam, double *const *ref,
                    double *const *mask, const int *pos, int Nw, const double *win,
                    int max_shift, int padding)
{
    if (getenv("UMPA_CALL_CAP")) CALL_CAP = atoi(getenv("UMPA_CALL_CAP")) > 0 ? atoi(getenv("UMPA_CALL_CAP")) : 1; else CALL_CAP = 500;
    oracle_model *m = (oracle_model *)calloc(1, sizeof(*m));
    m->kind = kind; m->Na = Na; m->Nw = Nw; m->max_shift = max_shift; m->padding = padding;
    m->subpx = -1; m->ref_mode = 0; m->has_mask = mask != NULL;   /* lib/Model.cpp:214-215 */
    m->dim = (int *)malloc(2 * Na * sizeof(int));
    m->pos = (int *)malloc(2 * Na * sizeof(int));
    memcpy(m->dim, dims, 2 * Na * sizeof(int));
    memcpy(m->pos, pos, 2 * Na * sizeof(int));
    m->sam = (const double **)malloc(Na * sizeof(double *));
    m->ref = (const double **)malloc(Na * sizeof(double *));
    m->mask = (const double **)malloc(Na * sizeof(double *));
    for (int k = 0; k < Na; k++) { m->sam[k] = sam[k]; m->ref[k] = ref[k]; m->mask[k] = mask ? mask[k] : NULL; }
    int S = 2 * Nw + 1;
    m->win = (double *)malloc(S * S * sizeof(double));
    memcpy(m->win, win, S * S * sizeof(double));
    return m;
}

void umpaor_destroy(void *p)
{
    oracle_model *m = (oracle_model *)p;
    free(m->dim); free(m->pos); free(m->sam); free(m->ref); free(m->mask); free(m->win); free(m);
}

int umpaor_set_window(void *p, const double *win, int Nw)       /* lib/Model.cpp:239-246 */
{
    oracle_model *m = (oracle_model *)p;
    if (Nw < 0) return -1;
    int S = 2 * Nw + 1;
    free(m->win);
    m->win = (double *)malloc(S * S * sizeof(double));
    memcpy(m->win, win, S * S * sizeof(double));
    m->Nw = Nw;
    return 0;
}

void umpaor_set_subpx(void *p, int mode) { ((oracle_model *)p)->subpx = mode; }
void umpaor_set_reference_shift(void *p, int v) { ((oracle_model *)p)->ref_mode = v; }

/* lib/Model.cpp:273-314 */
int umpaor_coverage(void *p, double *out, int i, int j)
{
    const oracle_model *m = (const oracle_model *)p;
    double c = 0.0;
    for (int k = 0; k < m->Na; k++) {
        if (!frame_covers(m, k, i, j)) continue;
        c += m->has_mask ? m->mask[k][(size_t)(i - m->pos[2 * k]) * m->dim[2 * k + 1] + (j - m->pos[2 * k + 1])] : 1.0;
    }
    *out = c;
    return ST_OK;
}

/* lib/Model.cpp:533-542, :887-897, :1181-1192; values = [cost, T, (v | a, b, c in)] */
int umpaor_cost(void *p, int i, int j, int si, int sj, double *values)
{
    const oracle_model *m = (const oracle_model *)p;
    double kern[BLUR_SIDE * BLUR_SIDE];
    fit_t fit = {0.0, 0.0};
    if (m->kind == KIND_DFKERNEL) build_blur_kernel(values[2], values[3], values[4], kern);
    int st = eval_cost(m, i, j, si, sj, kern, &values[0], &fit);
    values[1] = fit.t;
    if (m->kind == KIND_DF) values[2] = fit.v;
    return st;
}

/* lib/Model.cpp:562-578, :923-940, :1222-1238; values = [f, T, dx(col), dy(row), (df | a,b,c in)] */
static int min_pixel(const oracle_model *m, int i, int j, double *values, double *uv,
                     double *memo, double *nb, int *ncalls)
{
    double kern[BLUR_SIDE * BLUR_SIDE];
    double D = 0.0;                                   /* uninitialised in the reference */
    fit_t fit = {0.0, 0.0};
    if (m->kind == KIND_DFKERNEL) build_blur_kernel(values[4], values[5], values[6], kern);
    int st = minimise(m, i, j, kern, &D, uv, &fit, memo, nb, ncalls);
    values[0] = D;
    values[1] = fit.t;
    values[2] = uv[1];
    values[3] = uv[0];
    if (m->kind == KIND_DF) values[4] = fit.v;
    return st;
}

int umpaor_min(void *p, int i, int j, double *values, double *uv, double *dbg_d, double *dbg_a, int *ncalls)
{
    double memo[25], nb[16];
    int n = 0;
    memset(nb, 0, sizeof(nb));
    int st = min_pixel((const oracle_model *)p, i, j, values, uv, memo, nb, &n);
    if (dbg_d) memcpy(dbg_d, memo, sizeof(memo));
    if (dbg_a) memcpy(dbg_a, nb, sizeof(nb));
    if (ncalls) *ncalls = n;
    return st;
}

/* The pixel loop of model.pyx:476-492 (padding added to the coordinates as there). */
void umpaor_match_region(void *p, int start0, int step0, int N0, int start1, int step1, int N1,
                         double *values, int nparam, double *uv, int *err,
                         const double *covermap, double cover_threshold,
                         double *dbg_d, double *dbg_a, int *dbg_ncalls, int num_threads)
{
    const oracle_model *m = (const oracle_model *)p;
    const int off = m->padding;
    if (!quad_ready) quad_init();
#pragma omp parallel for schedule(dynamic) num_threads(num_threads)
    for (int xi = 0; xi < N0; xi++) {
        double memo[25], nb[16];
        for (int xj = 0; xj < N1; xj++) {
            size_t px = (size_t)xi * N1 + xj;
            if (covermap && covermap[px] < cover_threshold) continue;
            int n = 0;
            memset(nb, 0, sizeof(nb));
            int st = min_pixel(m, off + start0 + step0 * xi, off + start1 + step1 * xj,
                               &values[px * nparam], &uv[px * 2], memo, nb, &n);
            err[px] = st & ST_OK;
            if (dbg_d) memcpy(&dbg_d[px * 25], memo, sizeof(memo));
            if (dbg_a) memcpy(&dbg_a[px * 16], nb, sizeof(nb));
            if (dbg_ncalls) dbg_ncalls[px] = n;
        }
    }
}

/* First-touch copy for the CPU baseline of bench.py: the pages of `dst` (fresh, untouched memory) are written by
 * the same static thread partition that a row-parallel match reads them with, so on a multi-socket host each
 * thread's rows live on its own NUMA node instead of all on the node of the thread that generated the data. */
void umpaor_parallel_copy(double *dst, const double *src, size_t n, int num_threads)
{
    const size_t chunk = 512;                      /* one page */
    const size_t nchunk = (n + chunk - 1) / chunk;
#pragma omp parallel for schedule(static) num_threads(num_threads)
    for (size_t c = 0; c < nchunk; c++) {
        size_t a = c * chunk, b = a + chunk < n ? a + chunk : n;
        memcpy(dst + a, src + a, (b - a) * sizeof(double));
    }
}

int umpaor_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// umpa_walk.h -- per-pixel minimiser (integer descent + sub-pixel fit) for gfx950.
//
// The reference runs one blocking C++ routine per pixel (UMPA/lib/Optim.cpp:233-479) that
// calls the cost function wherever it needs a value.  On the GPU the cost evaluation is the
// expensive, wave-convergent part, so the walk is written here as a *resumable* machine: it
// posts the next shift it needs (`req_i`, `req_j`), the caller evaluates (or looks up) the
// cost for all active lanes together, and `walk_feed` consumes the value and advances to the
// next request.  The sequence of requests, the tie rules, the memo handling, the hard
// restart with stale fit parameters and every early exit are those of the reference; the
// trajectory (err, Ncalls, integer minimum) is bit-identical by construction.
#pragma once
#include <hip/hip_runtime.h>

#define UMPA_ST_OK        1
#define UMPA_ST_BOUND     2
#define UMPA_ST_DIM       4
#define UMPA_ST_POSITIVE  8

#define UMPA_CALL_CAP 500          // Optim.cpp:14 (the default of ModelDev::call_cap)
#define UMPA_TIE      1e-8         // Optim.cpp:243
// The reference's loop (Optim.cpp:267-474) only tests its call cap, and moves between memoised cells cost no call: with a
// NaN cost in the neighbourhood (a masked window without a single valid pixel pair: 0/0) the centre can step back and forth
// between two known cells for ever -- the reference hangs there.  A GPU kernel must not: this many moves in a row without a
// cost call end the walk like the call cap does (status not ok).  oracle/umpa_oracle.c carries the same rule; no walk on
// finite costs comes near it (a 5x5 memo holds at most four known cells in a row).
#define UMPA_MOVE_CAP 64

namespace umpa {

// Pointers that reach a kernel inside a by-value struct, or are loaded from a descriptor table, are
// "generic" to the compiler and turn into flat_load / flat_store, which also tick the LDS counter
// (lgkmcnt) and so serialise against ds_read / ds_write.  Every such pointer is device memory here:
// say so, and get global_load / global_store.
#define UMPA_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const UMPA_GLOBAL T* gp(const T* p) { return (const UMPA_GLOBAL T*)p; }
template <class T>
__device__ __forceinline__ UMPA_GLOBAL T* gpw(T* p) { return (UMPA_GLOBAL T*)p; }

// 1/x by v_rcp_f64 and two Newton steps (full double precision for the finite, non-zero arguments met here)
// instead of the IEEE division sequence (scale, rcp, four FMAs, fmas, fixup)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

struct Fit { double t, v; };       // the CostArgs payload (Model.h:29-52)

enum Phase { PH_CENTRE = 0, PH_LO = 1, PH_HI = 2, PH_GATHER = 3, PH_FIT = 4, PH_DONE = 5 };

// The 5x5 memo of the reference (Optim.cpp:252: the costs known around the current centre, -1 = unknown) is kept
// here as two things:
//   * `Walk::known`, a 25-bit mask (bit 5r+c: cell (r,c) of the neighbourhood, centre (2,2)): what the reference
//     tests with `d[..] < 0`.  Moving the centre shifts the mask (cells that scroll in are unknown), a hard restart
//     clears it -- a few integer instructions instead of the 25 loads and stores of the reference's array shuffle;
//   * the values, in LDS, one column per lane ([slot][lane]: the walk indexes them with run-time cell numbers, which
//     would push a per-lane array into scratch memory), addressed as a TORUS: the cost of the absolute shift (x, y)
//     lives in slot 5 (x mod 5) + (y mod 5).  The 5x5 neighbourhood spans five consecutive values per axis, so this is
//     a bijection for any centre and nothing ever moves; slots are never initialised (the mask says what is valid).
template <int STRIDE>
struct LdsMemo {
    double* p;
    __device__ __forceinline__ double& operator[](int q) const { return p[q * STRIDE]; }
};

struct Walk {
    int ci, cj;        // integer centre of the 5x5 neighbourhood
    int req_i, req_j;  // shift whose cost is wanted next
    int axis;          // 0: columns (west/east), 1: rows
    int found0, found1;
    int n;             // cost calls so far (minimizer_debug::Ncalls)
    int phase;
    int g;             // gather: the cell 0..15 being asked for
    unsigned need;     // gather: cells of the 4x4 neighbourhood still unknown (bit g)
    int ip, jp;        // quadrant of the 4x4 neighbourhood
    double c0;         // cost at the centre
    int status;
    Fit live, kept;    // *args and args_copy of Optim.cpp:249,265
    double out, uv0, uv1;
    unsigned known;    // bit 5r+c: cell (r,c) has been evaluated since it scrolled in
    int bi, bj;        // (ci - 2) mod 5, (cj - 2) mod 5: torus row / column of cell (0,0)
};

__device__ __forceinline__ int wrap5(int x) { return x >= 5 ? x - 5 : x; }                  // x in [0, 9]
__device__ __forceinline__ int mod5(int x) { const int r = x % 5; return r < 0 ? r + 5 : r; }
// LDS slot of cell (r, c), 0 <= r, c <= 4
__device__ __forceinline__ int walk_slot(const Walk& w, int r, int c) { return 5 * wrap5(w.bi + r) + wrap5(w.bj + c); }

// minimizer_debug::d as the reference leaves it: the known costs of the 5x5 neighbourhood, -1 elsewhere
template <class Memo>
__device__ __forceinline__ double walk_memo_cell(const Walk& w, Memo memo, int q)
{
    return (w.known >> q) & 1u ? memo[walk_slot(w, q / 5, q % 5)] : -1.0;
}

// ---------------------------------------------------------------- sub-pixel fits

// Optim.cpp:41-130.  a[4*row+col] are control points of a uniform bicubic B-spline on
// {-1,0,1,2}^2; Newton-Raphson on its gradient, <= 21 steps, unclamped.  c[4q+p] is the
// coefficient of x^p y^q (x = row coordinate), times 36.
__device__ inline double spmin(const double* a, double& px, double& py)
{
    // B[m][p]: coefficient of t^p of cubic B-spline basis function m, times 6
    const double B[4][4] = {{1, -3, 3, -1}, {4, 0, -6, 3}, {1, 3, 3, -3}, {0, 0, 0, 1}};
    double c[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // row-combine first: e[p][s] = sum_r a[r][s] B[r][p]; then c[q][p] = sum_s e[p][s] B[s][q]
#pragma unroll
        for (int p = 0; p < 4; p++) {
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                double e = 0.0;
#pragma unroll
                for (int r = 0; r < 4; r++) e += a[4 * r + s] * B[r][p];
                acc += e * B[s][q];
            }
            c[4 * q + p] = acc;
        }
    }
    double x = px, y = py;
    double g[4], g1[4], g2[4];
    for (int it = 0; it <= 20; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double c0 = c[4 * q], c1 = c[4 * q + 1], c2 = c[4 * q + 2], c3 = c[4 * q + 3];
            g[q] = c0 + x * (c1 + x * (c2 + x * c3));
            g1[q] = c1 + x * (2 * c2 + x * 3 * c3);
            g2[q] = 2 * c2 + 6 * c3 * x;
        }
        const double fx = g1[0] + y * (g1[1] + y * (g1[2] + y * g1[3]));
        const double fxx = g2[0] + y * (g2[1] + y * (g2[2] + y * g2[3]));
        const double fy = g[1] + y * (2 * g[2] + y * 3 * g[3]);
        const double fxy = g1[1] + y * (2 * g1[2] + y * 3 * g1[3]);
        const double fyy = 2 * g[2] + 6 * g[3] * y;
        const double det = fxx * fyy - fxy * fxy;
        const double dx = (fxy * fy - fyy * fx) / det;
        const double dy = (fxy * fx - fxx * fy) / det;
        x += dx;
        y += dy;
        if (dx * dx + dy * dy < 1e-8) break;
    }
    px = x;
    py = y;
#pragma unroll
    for (int q = 0; q < 4; q++) g[q] = c[4 * q] + x * (c[4 * q + 1] + x * (c[4 * q + 2] + x * c[4 * q + 3]));
    return (g[0] + y * (g[1] + y * (g[2] + y * g[3]))) / 36.0;
}

// Optim.cpp:155-185.  Least-squares paraboloid c0 + gi i + gj j + cii i^2 + cij ij + cjj j^2 on
// {-1,0,1,2}^2: the integer table is 400*pinv(design) (derivation: oracle/umpa_oracle.c
// quad_init and tests/test_oracle_golden.py).  The vertex pairs (cii,gj) and (cjj,gi) as the
// reference does (Optim.cpp:180-181).
__device__ inline double spmin_quad(const double* a, double& px, double& py)
{
    // 400*pinv(design), design columns [1, i, j, i^2, ij, j^2] on i,j in {-1,0,1,2}:
    //   constant row C0 and row-gradient row GI are tabulated (4x4, [row][col]); the
    //   column-gradient row is GI transposed; the quadratic rows have closed forms
    //   25*(i^2-i-1), 4*(2i-1)(2j-1), 25*(j^2-j-1).
    const int C0[16] = {14, 48, 32, -34, 48, 86, 74, 12, 32, 74, 66, 8, -34, 12, 8, -46};
    const int GI[16] = {-73, -61, -49, -37, 9, 13, 17, 21, 41, 37, 33, 29, 23, 11, -1, -13};
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0;
#pragma unroll
    for (int n = 0; n < 16; n++) {
        const int r = n >> 2, s = n & 3;
        const int i = r - 1, j = s - 1;
        const double v = a[n];
        p0 += C0[n] * v;
        p1 += GI[n] * v;
        p2 += GI[4 * s + r] * v;
        p3 += (25 * (i * i - i - 1)) * v;
        p4 += (4 * (2 * i - 1) * (2 * j - 1)) * v;
        p5 += (25 * (j * j - j - 1)) * v;
    }
    const double det = 4 * p3 * p5 - p4 * p4;
    px = -(2 * p3 * p2 - p4 * p1) / det;
    py = -(2 * p5 * p1 - p4 * p2) / det;
    return (p0 + 0.5 * (p2 * px + p1 * py)) / 400.0;
}

// ---------------------------------------------------------------- the walk

template <class Memo>
__device__ inline void walk_begin(Walk& w, Memo memo, double u0, double u1)
{
    w.ci = (int)round(u0);                                      // Optim.cpp:258-259
    w.cj = (int)round(u1);
    w.bi = mod5(w.ci - 2);
    w.bj = mod5(w.cj - 2);
    w.known = 0;                                                // Optim.cpp:252
    w.req_i = w.ci;
    w.req_j = w.cj;
    w.axis = 0;
    w.found0 = w.found1 = 0;
    w.n = 0;
    w.phase = PH_CENTRE;
    w.g = 0;
    w.need = 0;
    w.ip = w.jp = 0;
    w.c0 = 0.0;
    w.status = 0;
    w.live.t = w.live.v = 0.0;                                  // Model.cpp:68,74
    w.kept = w.live;
    w.out = 0.0;                                                // `D` is uninitialised in the reference (Model.cpp:566,927)
    w.uv0 = u0;
    w.uv1 = u1;
}

// Deliver the result of the pending request (status `st`, cost `val`, fit parameters `fit`)
// and advance to the next request, to PH_FIT (4x4 neighbourhood complete: walk_finish does the
// sub-pixel fit once for the whole wave) or to PH_DONE (failed).
//
// Written for a SIMD machine: the value is stored into the memo first, and the search for the next
// request then reads the two neighbours of the centre along the current axis back from the memo,
// whether they were just evaluated or memoised -- the reference's "fresh" (Optim.cpp:294,325) and
// "memoised" (:300,331) comparisons are the same expression on the same numbers, so only the
// snapshot `args_copy = *args` (:296,327) depends on which one it was.  The centre value rides in a
// register (`c0`): after a move it is the neighbour that was just compared.
template <class Memo>
__device__ inline void walk_feed(Walk& w, Memo memo, int st, double val, Fit fit, int call_cap = UMPA_CALL_CAP)
{
    w.n++;                                                      // Ncalls counts failed calls too (Optim.cpp:263-264)
    if (!(st & UMPA_ST_OK)) {                                   // any failed call returns at once, outputs as they stand
        w.status = st;
        w.phase = PH_DONE;
        return;
    }
    w.live = fit;
    // The call cap is tested in one place only, the `while (Ncalls < MAX_CALLS)` header (Optim.cpp:267): after the
    // centre evaluation and after every move.  A probe of the low or the high neighbour that reaches the cap still
    // finishes its iteration (the other probe, then the move or the gather); `goto start` skips the test as well.
    bool cap_test = w.phase == PH_CENTRE;
    bool scan = false;
    int moves = 0;                                              // centre moves since the last cost call (UMPA_MOVE_CAP)
    if (w.phase == PH_GATHER) {                                 // Optim.cpp:353-378
        const int rr = w.ip + (w.g >> 2), cc = w.jp + (w.g & 3);   // the `a` entry is this cell of the neighbourhood
        memo[walk_slot(w, rr, cc)] = val;
        if (val < w.c0) {                                       // missed a lower value: hard restart, centred there
            const int di = rr - 2, dj = cc - 2;
            w.ci += di;
            w.cj += dj;
            int x = w.bi + di, y = w.bj + dj;                   // in [-2, 6]
            x += x < 0 ? 5 : 0; y += y < 0 ? 5 : 0;
            w.bi = wrap5(x); w.bj = wrap5(y);
            w.known = 1u << 12;                                 // only the new centre (its value sits in its slot already)
            w.c0 = val;
            w.live = w.kept;                                    // stale on purpose (Optim.cpp:373)
            w.found0 = w.found1 = 0;
        } else {
            w.known |= 1u << (5 * rr + cc);
            w.need &= w.need - 1;
            scan = true;
        }
    } else {
        bool keep;
        int r, c;
        if (w.phase == PH_CENTRE) {                             // Optim.cpp:262-265
            r = 2; c = 2;
            w.c0 = val;
            keep = true;
        } else if (w.phase == PH_LO) {                          // Optim.cpp:287-297
            r = w.axis ? 1 : 2; c = w.axis ? 2 : 1;
            keep = !(val > w.c0 + UMPA_TIE);
        } else {                                                // PH_HI, Optim.cpp:320-328
            r = w.axis ? 3 : 2; c = w.axis ? 2 : 3;
            keep = !(val > w.c0 - UMPA_TIE);
        }
        memo[walk_slot(w, r, c)] = val;
        w.known |= 1u << (5 * r + c);
        if (keep) w.kept = w.live;
    }

    while (!scan) {
        if (cap_test && w.n >= call_cap) {
            w.status = st & ~UMPA_ST_OK;                        // Optim.cpp:477
            w.phase = PH_DONE;
            return;
        }
        cap_test = true;
        const int lr = w.axis ? 1 : 2, lc = w.axis ? 2 : 1, hr = w.axis ? 3 : 2, hc = w.axis ? 2 : 3;
        if (!((w.known >> (5 * lr + lc)) & 1u)) {
            w.phase = PH_LO;
            w.req_i = w.ci - (w.axis ? 1 : 0);
            w.req_j = w.cj - (w.axis ? 0 : 1);
            return;
        }
        if (!((w.known >> (5 * hr + hc)) & 1u)) {
            w.phase = PH_HI;
            w.req_i = w.ci + (w.axis ? 1 : 0);
            w.req_j = w.cj + (w.axis ? 0 : 1);
            return;
        }
        const double dlo = memo[walk_slot(w, lr, lc)], dhi = memo[walk_slot(w, hr, hc)];
        const bool lo_up = dlo > w.c0 + UMPA_TIE;               // Optim.cpp:294,300
        const bool hi_up = dhi > w.c0 - UMPA_TIE;               // Optim.cpp:325,331
        if (lo_up && hi_up) {                                   // Optim.cpp:334-417
            const int f = dlo < dhi ? -1 : 1;
            if (w.axis) w.found1 = f; else w.found0 = f;
            const int other = w.axis ? w.found0 : w.found1;
            if (!other) {
                w.axis ^= 1;
                continue;
            }
            // both axes have a minimum here, so the four neighbours of the centre are known
            w.ip = memo[walk_slot(w, 3, 2)] < memo[walk_slot(w, 1, 2)] ? 1 : 0;     // Optim.cpp:344-345
            w.jp = memo[walk_slot(w, 2, 3)] < memo[walk_slot(w, 2, 1)] ? 1 : 0;
            // which cells of the 4x4 neighbourhood are still unknown: rows ip .. ip+3, columns jp .. jp+3 of the mask
            const unsigned sub = w.known >> (5 * w.ip + w.jp);
            const unsigned have = (sub & 0xFu) | ((sub >> 5) & 0xFu) << 4 | ((sub >> 10) & 0xFu) << 8 | ((sub >> 15) & 0xFu) << 12;
            w.need = ~have & 0xFFFFu;
            scan = true;
        } else {                                                // Optim.cpp:420-474
            w.uv0 = w.ci;
            w.uv1 = w.cj;
            w.out = w.c0;
            bool up = lo_up;
            if (!hi_up && !lo_up) up = dhi < dlo;
            const int dir = up ? 1 : -1;
            // the centre moves by `dir` along `axis` (Optim.cpp:436-470): cells that scroll in are unknown
            if (w.axis) {
                w.ci += dir;
                w.bi = up ? wrap5(w.bi + 1) : (w.bi == 0 ? 4 : w.bi - 1);
                w.known = up ? (w.known >> 5) : ((w.known << 5) & 0x1FFFFFFu);
                w.found0 = 0;
            } else {
                w.cj += dir;
                w.bj = up ? wrap5(w.bj + 1) : (w.bj == 0 ? 4 : w.bj - 1);
                w.known = up ? ((w.known & ~0x0108421u) >> 1) : ((w.known & ~0x1084210u) << 1);
                w.found1 = 0;
            }
            w.c0 = up ? dhi : dlo;                              // the new centre is the neighbour stepped onto
            if (++moves > UMPA_MOVE_CAP) {                      // (never on finite costs; see UMPA_MOVE_CAP)
                w.status = st & ~UMPA_ST_OK;
                w.phase = PH_DONE;
                return;
            }
        }
    }
    // fill the 4x4 neighbourhood, asking for what is missing in the reference's order (Optim.cpp:353-362)
    if (w.need) {
        w.g = __ffs(w.need) - 1;
        w.phase = PH_GATHER;
        w.req_i = w.ci + w.ip + (w.g >> 2) - 2;
        w.req_j = w.cj + w.jp + (w.g & 3) - 2;
        return;
    }
    w.live = w.kept;                                            // Optim.cpp:386
    w.status = st;
    w.phase = PH_FIT;
}

// The request that will follow the pending one -- where that is known before the pending one's cost is (the high neighbour
// after the low one, the next cell of the gather), or a guess (a walk that just stepped along an axis often steps again).
// A caller may evaluate it beside the pending request (two independent chains of loads instead of one) and deliver it
// with walk_feed if, after the pending result has been delivered, the walk does ask for exactly this shift; otherwise the
// value is dropped.  The walk itself is untouched: which costs it sees, in which order, and Ncalls stay the reference's.
#ifndef UMPA_WALK_GUESS
#define UMPA_WALK_GUESS 1
#endif
__device__ inline bool walk_speculate(const Walk& w, int& si, int& sj)
{
    const int di = w.axis ? 1 : 0, dj = w.axis ? 0 : 1;
    if (w.phase == PH_CENTRE) {                                 // then the low neighbour (nothing else is known at the start)
        si = w.ci - di; sj = w.cj - dj;
        return true;
    }
    if (w.phase == PH_LO) {
        const int hr = w.axis ? 3 : 2, hc = w.axis ? 2 : 3;
        if (!((w.known >> (5 * hr + hc)) & 1u)) { si = w.ci + di; sj = w.cj + dj; }   // the high neighbour follows in any case
        else if (UMPA_WALK_GUESS) { si = w.req_i - di; sj = w.req_j - dj; }   // guess: the walk steps down and asks for the cell beyond
        else return false;
        return true;
    }
    if (w.phase == PH_HI) {                                     // guess: the walk steps up and asks for the cell beyond
        if (!UMPA_WALK_GUESS) return false;                     // (choosing the side by the low neighbour's cost: measured slower)
        si = w.req_i + di; sj = w.req_j + dj;
        return true;
    }
    if (w.phase == PH_GATHER) {
        const unsigned rest = w.need & (w.need - 1);
        if (!rest) return false;
        const int g = __ffs(rest) - 1;
        si = w.ci + w.ip + (g >> 2) - 2; sj = w.cj + w.jp + (g & 3) - 2;
        return true;
    }
    return false;
}

// Sub-pixel refinement for the lanes whose walk completed (Optim.cpp:386-410).  `nb` receives the
// 4x4 neighbourhood (minimizer_debug::a); it is the (ip,jp) sub-block of the memo.
template <class Memo>
__device__ inline void walk_finish(Walk& w, Memo memo, int subpx, double* nb)
{
    if (w.phase != PH_FIT) {
#pragma unroll
        for (int g = 0; g < 16; g++) nb[g] = 0.0;
        return;
    }
#pragma unroll
    for (int g = 0; g < 16; g++) nb[g] = memo[walk_slot(w, w.ip + (g >> 2), w.jp + (g & 3))];
    double x = 1.0 - w.ip, y = 1.0 - w.jp;                      // Optim.cpp:395-396
    if (subpx == 0) w.out = x;                                  // Optim.cpp:399
    else if (subpx == 1) w.out = spmin_quad(nb, x, y);
    else w.out = spmin(nb, x, y);
    w.uv0 = x + (w.ci + w.ip - 1.0);                            // Optim.cpp:407-408
    w.uv1 = y + (w.cj + w.jp - 1.0);
    w.phase = PH_DONE;
}

} // namespace umpa

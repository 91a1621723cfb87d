// umpa_direct.h -- the general ("direct") matching kernel: one lane per output pixel, the
// windowed multi-frame cost evaluated by an explicit window x frame sum straight from the
// frames in HBM/L2, exactly the arithmetic of the reference cost functions
// (UMPA/lib/Model.cpp:359-509 NoDF, :631-862 DF), including masks, per-frame positions
// (sample stepping), unequal frame shapes and both coordinate conventions.
//
// It is the path for everything the tiled fast path (umpa_tiled.h) does not cover, and the
// first-principles cross-check for it.  All 64 lanes of a wave evaluate their cost together
// (the window/frame loops have wave-uniform trip counts); only the cheap walk bookkeeping
// diverges.  Window weights and frame descriptors are wave-uniform and come in through
// scalar loads.
#pragma once
#include "umpa_walk.h"

namespace umpa {

struct FrameDesc {
    const double* sam;
    const double* ref;
    const double* mask;   // NULL when the model has no masks
    int H, W;             // frame shape            (ModelBase::dim, Model.h:86)
    int pi, pj;           // frame position >= 0    (ModelBase::pos, Model.h:87)
};

struct ModelDev {
    const FrameDesc* frames;   // [Na], device
    const double* win;         // (2Nw+1)^2, device (ModelBase::win, Model.h:88)
    double win_sum;            // sum of win in row-major order (the `denom` of Model.cpp:724-739)
    int Na, Nw, ms, padding;
    int Nwt;                   // the frame count the cost of a model without masks is divided by (Model.cpp:425, :711): the
                               // model's, also where `frames` lists only the frames that contribute in a sub-rectangle
    int subpx, ref_mode;
    int call_cap;              // MAX_CALLS of Optim.cpp:14 (500; UMPA_CALL_CAP in the environment of the creating process
                               // lowers it for the tests that pin the behaviour at the cap)
};

struct RegionArgs {
    int org0, step0, N0;       // pixel (xi,xj) sits at frame coords (org0+step0*xi, org1+step1*xj)
    int org1, step1, N1;
    int pitch;                 // tiled path: pixels per row of the OUTPUT arrays (N1, or the full region's N1 when this is one
                               // rectangle of it written in place); the general kernels take dense arrays (pitch = N1)
    double* values; int nparam;
    size_t v_px, v_k;          // element (pixel px, parameter k) of `values` sits at px*v_px + k*v_k (interleaved: nparam,1; planar: 1,N0*N1)
    double* uv;                // may be NULL: start at (0,0), result not stored
    int* err;
    const double* cover; double thr;
    double* dbg_d; double* dbg_a; int* dbg_n;
    double* kern; size_t kern_stride;   // DFKernel only: per-pixel 17x17 blur kernels, [tap][pixel]
    int row_base;                       // first region row covered by `kern` (row chunking of DFKernel launches)
    double* blur; int blur_F;           // DFKernel only: per-pixel blurred reference footprints, [frame][F][F][pixel] (stride = kern_stride)
    int blur_ready;                     // DFKernel only: kernels and footprints have been filled by blur_tiles_kernel
};

// descriptor k, read through the CONSTANT address space: the table is written before any kernel runs and k is
// wave-uniform, so this is an s_load into SGPRs.  Read as ordinary global memory it becomes a vector load followed
// by s_waitcnt vmcnt(0) -- which also waits for every frame load (and table store) still in flight.
// (member-wise: a struct copy across address spaces does not compile in the host pass)
#define UMPA_CONSTANT __attribute__((address_space(4)))
__device__ __forceinline__ FrameDesc load_frame(const FrameDesc* frames, int k)
{
    const UMPA_CONSTANT FrameDesc* g = (const UMPA_CONSTANT FrameDesc*)frames + k;
    FrameDesc f;
    f.sam = g->sam; f.ref = g->ref; f.mask = g->mask;
    f.H = g->H; f.W = g->W; f.pi = g->pi; f.pj = g->pj;
    return f;
}

__device__ __forceinline__ double pair_weight(double a, double b)   // Utils.cpp:125-130
{
    return a * b * fast_rcp(a + b + 1e-8);       // one division per window element and frame: the reciprocal is what the masked kernels spend their time on
}

#define UMPA_BLUR_HALF 8                                   // Model.h:7 KERNEL_WINDOW_SIZE
#define UMPA_BLUR_SIDE (2 * UMPA_BLUR_HALF + 1)
#define UMPA_BLUR_TAPS (UMPA_BLUR_SIDE * UMPA_BLUR_SIDE)

// CostArgsDFKernel constructor (Model.cpp:88-117): normalised exp(-a i^2 - b i j - c j^2) on [-8,8]^2,
// written to this pixel's column of the scratch array.
__device__ inline void build_blur_kernel(UMPA_GLOBAL double* kern, size_t stride, double a, double b, double c)
{
    double norm = 0.0;
    for (int i = -UMPA_BLUR_HALF; i <= UMPA_BLUR_HALF; i++)
        for (int j = -UMPA_BLUR_HALF; j <= UMPA_BLUR_HALF; j++) {
            const double g = exp(-a * i * i - b * i * j - c * j * j);            // Utils.cpp:46-50
            kern[(size_t)((i + UMPA_BLUR_HALF) * UMPA_BLUR_SIDE + j + UMPA_BLUR_HALF) * stride] = g;
            norm += g;
        }
    for (int n = 0; n < UMPA_BLUR_TAPS; n++) kern[(size_t)n * stride] /= norm;
}

// convolve / weighted_convolve (Utils.cpp:85-117) of one reference pixel with this pixel's kernel
template <bool MASK>
__device__ __forceinline__ double blur_at(const UMPA_GLOBAL double* img, const UMPA_GLOBAL double* wgt, int W, int i, int j,
                                          const UMPA_GLOBAL double* kern, size_t stride)
{
    double acc = 0.0, wsum = 0.0;
    for (int a = -UMPA_BLUR_HALF; a <= UMPA_BLUR_HALF; a++) {
        const size_t row = (size_t)(i + a) * W + j;
        for (int b = -UMPA_BLUR_HALF; b <= UMPA_BLUR_HALF; b++) {
            const double kv = kern[(size_t)((a + UMPA_BLUR_HALF) * UMPA_BLUR_SIDE + b + UMPA_BLUR_HALF) * stride];
            if (MASK) {
                const double wv = wgt[row + b];
                acc += kv * img[row + b] * wv;
                wsum += kv * wv;
            } else
                acc += kv * img[row + b];
        }
    }
    return MASK ? acc / wsum : acc;
}

// The kernel-dark-field model blurs the reference with the PIXEL's kernel (Model.cpp:88-117, :997-1151): the blurred
// values a pixel's evaluations read all lie in one footprint around it -- its window, widened by max_shift - 1 where the
// reference window moves -- and do not depend on the shift.  The reference (and round 1 here) recomputes the 289-tap
// blur of every window element at every evaluation, ~18 times per pixel; this fills the footprint once.  Per footprint
// row and chunk of 8 columns: for each kernel row, 24 image values and 17 kernel values feed 8 x 17 FMAs (0.3 loads per
// FMA instead of 2), every output summing its taps in blur_at's order (kernel rows outer, columns inner), so the
// values are those of blur_at bit for bit.
#define UMPA_BLUR_CHUNK 8
template <bool MASK>
__device__ inline void blur_footprint(const ModelDev& m, int i, int j, const UMPA_GLOBAL double* kern, size_t stride,
                                      UMPA_GLOBAL double* blur, int F, int halo)
{
    const int pad = m.padding;
    for (int k = 0; k < m.Na; k++) {
        const FrameDesc f = load_frame(m.frames, k);
        const int li = i - f.pi, lj = j - f.pj;
        if (li - pad < 0 || li + pad > f.H || lj - pad < 0 || lj + pad > f.W) continue;     // frame does not contribute here
        UMPA_GLOBAL double* out = blur + (size_t)k * F * F * stride;
        const UMPA_GLOBAL double* img = gp(f.ref);
        const UMPA_GLOBAL double* wgt = gp(f.mask);
        for (int fy = 0; fy < F; fy++) {
            for (int fx0 = 0; fx0 < F; fx0 += UMPA_BLUR_CHUNK) {
                double acc[UMPA_BLUR_CHUNK], wacc[UMPA_BLUR_CHUNK];
#pragma unroll
                for (int o = 0; o < UMPA_BLUR_CHUNK; o++) { acc[o] = 0.0; wacc[o] = 0.0; }
                for (int a = -UMPA_BLUR_HALF; a <= UMPA_BLUR_HALF; a++) {
                    const size_t row = (size_t)(li - halo + fy + a) * f.W;
                    const int c0 = lj - halo + fx0 - UMPA_BLUR_HALF;
                    double v[UMPA_BLUR_CHUNK + UMPA_BLUR_SIDE - 1], wv[MASK ? UMPA_BLUR_CHUNK + UMPA_BLUR_SIDE - 1 : 1], kv[UMPA_BLUR_SIDE];
#pragma unroll
                    for (int t = 0; t < UMPA_BLUR_CHUNK + UMPA_BLUR_SIDE - 1; t++) {
                        const int c = min(c0 + t, f.W - 1);                      // past the footprint's last column: unused outputs
                        v[t] = img[row + c];
                        if (MASK) wv[t] = wgt[row + c];
                    }
#pragma unroll
                    for (int b = 0; b < UMPA_BLUR_SIDE; b++) kv[b] = kern[(size_t)((a + UMPA_BLUR_HALF) * UMPA_BLUR_SIDE + b) * stride];
#pragma unroll
                    for (int o = 0; o < UMPA_BLUR_CHUNK; o++)
#pragma unroll
                        for (int b = 0; b < UMPA_BLUR_SIDE; b++) {
                            if (MASK) { acc[o] += kv[b] * v[o + b] * wv[o + b]; wacc[o] += kv[b] * wv[o + b]; }
                            else acc[o] += kv[b] * v[o + b];
                        }
                }
#pragma unroll
                for (int o = 0; o < UMPA_BLUR_CHUNK; o++)
                    if (fx0 + o < F) out[(size_t)(fy * F + fx0 + o) * stride] = MASK ? acc[o] / wacc[o] : acc[o];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// blur_tiles: blur_footprint for a whole 64 x 4 pixel box at a time ("ideal for LDS tiling").
//
// The work of the kernel-dark-field model is this blur: 289 taps x F x F footprint values x Na frames per PIXEL
// (every pixel has its own kernel, Model.cpp:88-117), 70 k FMAs per pixel on the E_dfkernel-type stack, ~1 M with
// BASELINE config C2's parameters.  blur_footprint fed them from global memory, 3.3 FMAs per load.  Here the workgroup
// stages the reference patch of its box (box + halo + 8 on every side; the mask patch beside it) in LDS once per frame;
// a lane then blurs its footprint in blocks of 4 rows x 8 columns: per kernel row its 17 taps (its own column of the
// scratch array, coalesced across the wave) and 4 x 24 patch values out of LDS feed 4 x 8 x 17 FMAs -- 32 FMAs per
// global load, 5.7 per LDS read.  Every output still sums its taps in blur_at's order (kernel rows outer, columns
// inner), so the footprints -- and every map -- are those of blur_footprint bit for bit.
// ------------------------------------------------------------------------------------------------
#define UMPA_BLURT_BX 64
#define UMPA_BLURT_BY 4
#define UMPA_BLURT_CH 4               // footprint rows per register block
#define UMPA_BLURT_CW 8               // footprint columns per register block

struct BlurTileGeom { int halo, F, PR, PC; };     // patch: PR rows of PC doubles (PC includes UMPA_BLURT_CW columns of slack)

inline BlurTileGeom blur_tile_geometry(int halo, int step0, int step1)
{
    BlurTileGeom g;
    g.halo = halo; g.F = 2 * halo + 1;
    const int reach = halo + UMPA_BLUR_HALF;
    g.PR = (UMPA_BLURT_BY - 1) * step0 + 1 + 2 * reach + UMPA_BLURT_CH;          // + slack for the last row block
    g.PC = ((UMPA_BLURT_BX - 1) * step1 + 1 + 2 * reach + UMPA_BLURT_CW) | 1;    // odd pitch
    return g;
}

template <bool MASK>
__global__ void __launch_bounds__(UMPA_BLURT_BX * UMPA_BLURT_BY, 2)
blur_tiles_kernel(ModelDev m, RegionArgs A, BlurTileGeom G)
{
    extern __shared__ __attribute__((aligned(16))) char blur_raw[];
    double* pimg = reinterpret_cast<double*>(blur_raw);
    double* pwgt = pimg + G.PR * G.PC;                                 // MASK only
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * UMPA_BLURT_BX + tx;
    const int xj = blockIdx.x * UMPA_BLURT_BX + tx, xi = blockIdx.y * UMPA_BLURT_BY + ty;
    const bool valid = xi < A.N0 && xj < A.N1;
    const size_t px = (size_t)min(xi, A.N0 - 1) * A.N1 + min(xj, A.N1 - 1);
    const bool wanted = valid && !(A.cover && gp(A.cover)[px] < A.thr);          // model.pyx:480-481
    const int i = A.org0 + A.step0 * xi, j = A.org1 + A.step1 * xj;
    const int gi0 = A.org0 + A.step0 * (blockIdx.y * UMPA_BLURT_BY), gj0 = A.org1 + A.step1 * (blockIdx.x * UMPA_BLURT_BX);   // box origin
    const int reach = G.halo + UMPA_BLUR_HALF, pad = m.padding, F = G.F;
    const size_t stride = A.kern_stride;
    UMPA_GLOBAL double* kern = gpw(A.kern) + ((size_t)(min(xi, A.N0 - 1) - A.row_base) * A.N1 + min(xj, A.N1 - 1));
    UMPA_GLOBAL double* blur = gpw(A.blur) + ((size_t)(min(xi, A.N0 - 1) - A.row_base) * A.N1 + min(xj, A.N1 - 1));
    if (wanted) {                                                      // Model.cpp:1228-1229: kernel from values[4..6]
        const UMPA_GLOBAL double* v = gp(A.values) + px * A.v_px;
        build_blur_kernel(kern, stride, v[4 * A.v_k], v[5 * A.v_k], v[6 * A.v_k]);
    }
    for (int k = 0; k < m.Na; k++) {
        const FrameDesc f = load_frame(m.frames, k);
        __syncthreads();                                               // the previous frame's patch has been read
        // ---- stage the patch: image rows gi0 - reach .. of frame k (frame coordinates = image - position), clamped at
        // the frame edges (what is clamped is never read by a pixel the frame contributes to)
        for (int q = tid; q < G.PR * G.PC; q += UMPA_BLURT_BX * UMPA_BLURT_BY) {
            const int r = q / G.PC, c = q - r * G.PC;
            const size_t src = (size_t)min(max(gi0 - f.pi - reach + r, 0), f.H - 1) * f.W + min(max(gj0 - f.pj - reach + c, 0), f.W - 1);
            pimg[q] = gp(f.ref)[src];
            if (MASK) pwgt[q] = gp(f.mask)[src];
        }
        __syncthreads();
        const int li = i - f.pi, lj = j - f.pj;
        if (!wanted || li - pad < 0 || li + pad > f.H || lj - pad < 0 || lj + pad > f.W) continue;   // frame does not contribute here
        UMPA_GLOBAL double* out = blur + (size_t)k * F * F * stride;
        // this pixel's footprint origin inside the patch: footprint (fy, fx), tap (a, b) reads patch (r0 + fy + a, c0 + fx + b)
        const int r0 = ty * A.step0, c0 = tx * A.step1;
        for (int fy0 = 0; fy0 < F; fy0 += UMPA_BLURT_CH) {
            for (int fx0 = 0; fx0 < F; fx0 += UMPA_BLURT_CW) {
                double acc[UMPA_BLURT_CH][UMPA_BLURT_CW], wacc[MASK ? UMPA_BLURT_CH : 1][UMPA_BLURT_CW];
#pragma unroll
                for (int y = 0; y < UMPA_BLURT_CH; y++)
#pragma unroll
                    for (int o = 0; o < UMPA_BLURT_CW; o++) { acc[y][o] = 0.0; if (MASK) wacc[y][o] = 0.0; }
                for (int a = 0; a < UMPA_BLUR_SIDE; a++) {
                    double kv[UMPA_BLUR_SIDE];
#pragma unroll
                    for (int b = 0; b < UMPA_BLUR_SIDE; b++) kv[b] = kern[(size_t)(a * UMPA_BLUR_SIDE + b) * stride];
#pragma unroll
                    for (int y = 0; y < UMPA_BLURT_CH; y++) {
                        const double* row = pimg + (r0 + fy0 + y + a) * G.PC + c0 + fx0;
                        const double* wrow = pwgt + (r0 + fy0 + y + a) * G.PC + c0 + fx0;
                        double v[UMPA_BLURT_CW + UMPA_BLUR_SIDE - 1], wv[MASK ? UMPA_BLURT_CW + UMPA_BLUR_SIDE - 1 : 1];
#pragma unroll
                        for (int t = 0; t < UMPA_BLURT_CW + UMPA_BLUR_SIDE - 1; t++) { v[t] = row[t]; if (MASK) wv[t] = wrow[t]; }
#pragma unroll
                        for (int o = 0; o < UMPA_BLURT_CW; o++)
#pragma unroll
                            for (int b = 0; b < UMPA_BLUR_SIDE; b++) {
                                if (MASK) { acc[y][o] += kv[b] * v[o + b] * wv[o + b]; wacc[y][o] += kv[b] * wv[o + b]; }
                                else acc[y][o] += kv[b] * v[o + b];
                            }
                    }
                }
#pragma unroll
                for (int y = 0; y < UMPA_BLURT_CH; y++)
#pragma unroll
                    for (int o = 0; o < UMPA_BLURT_CW; o++)
                        if (fy0 + y < F && fx0 + o < F)
                            out[(size_t)((fy0 + y) * F + fx0 + o) * stride] = MASK ? acc[y][o] / wacc[y][o] : acc[y][o];
            }
        }
    }
}

// One cost evaluation at pixel (i,j), shift (si rows, sj cols).  KIND: 0 NoDF, 1 DF, 2 DFKernel
// (NoDF arithmetic on a reference blurred on the fly, Model.cpp:997-1151).
// NWC > 0: the window half-width is a compile-time constant (the column loop unrolls, all loads of a window row are
// in flight together); NWC == 0: any window.
template <int KIND, bool MASK, int NWC = 0>
__device__ __forceinline__ int eval_direct(const ModelDev& m, int i, int j, int si, int sj,
                                           double& cost, Fit& fit,
                                           const UMPA_GLOBAL double* kern = nullptr, size_t kstride = 0,
                                           const UMPA_GLOBAL double* blur = nullptr, int blur_F = 0)
{
    const int ms = m.ms;
    // Model.cpp:372-399 / :654-681 (flags are asymmetric in the reference; kept)
    if (si <= -ms || si >= ms) return UMPA_ST_BOUND;
    if (sj <= -ms) return UMPA_ST_BOUND | UMPA_ST_DIM;
    if (sj >= ms) return UMPA_ST_BOUND | UMPA_ST_DIM | UMPA_ST_POSITIVE;

    int ri = i, rj = j, qi = i, qj = j;                  // Model.cpp:408-421 / :688-701
    if (m.ref_mode) { qi -= si; qj -= sj; } else { ri += si; rj += sj; }

    const int Nw = NWC > 0 ? NWC : m.Nw, S = 2 * Nw + 1, pad = m.padding;
    double t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
    double wt = MASK ? 0.0 : (double)m.Na;               // Model.cpp:425,:711 / :463,:777

    for (int k = 0; k < m.Na; k++) {
        const FrameDesc f = load_frame(m.frames, k);
        const int li = i - f.pi, lj = j - f.pj;          // Model.cpp:430-433 / :716-719
        if (li - pad < 0 || li + pad > f.H || lj - pad < 0 || lj + pad > f.W) continue;
        const size_t ro = (size_t)(ri - f.pi - Nw) * f.W + (rj - f.pj - Nw);
        const size_t qo = (size_t)(qi - f.pi - Nw) * f.W + (qj - f.pj - Nw);
        const UMPA_GLOBAL double* __restrict__ R = gp(f.ref) + ro;
        const UMPA_GLOBAL double* __restrict__ Q = gp(f.sam) + qo;
        const UMPA_GLOBAL double* __restrict__ MR = MASK ? gp(f.mask) + ro : nullptr;
        const UMPA_GLOBAL double* __restrict__ MQ = MASK ? gp(f.mask) + qo : nullptr;

        double s2 = 0, s4 = 0, s6 = 0, sm = 0;
        // one window element: the arithmetic and the order of Model.cpp:746-772 (and :813-841 with masks)
        auto term = [&](double w, double r, double q, double mr, double mq) {
            if (MASK) {
                if (KIND == 1) sm += w * r;              // the ref mean is never mask-weighted (Model.cpp:804)
                w *= pair_weight(mr, mq);
                wt += w;
                s2 += w;
            }
            const double wq = w * q, wr = w * r;
            t1 += wq * q;
            t3 += wr * r;
            t5 += wr * q;
            if (KIND == 1) { s4 += wq; s6 += wr; }
        };
        typedef double pair_t __attribute__((ext_vector_type(2), aligned(8)));
        for (int a = 0; a < S; a++) {
            const UMPA_GLOBAL double* wrow = gp(m.win) + a * S;
            const size_t off = (size_t)a * f.W;
            int b = 0;
            if (KIND != 2) {
                // two columns per step: 16-byte loads (the L1/TA path is half-rate for 8-byte ones); the terms are
                // still accumulated in the reference's order
                auto pair_step = [&](int c) {
                    const pair_t r2 = *reinterpret_cast<const UMPA_GLOBAL pair_t*>(R + off + c);
                    const pair_t q2 = *reinterpret_cast<const UMPA_GLOBAL pair_t*>(Q + off + c);
                    pair_t mr2 = {0.0, 0.0}, mq2 = {0.0, 0.0};
                    if (MASK) {
                        mr2 = *reinterpret_cast<const UMPA_GLOBAL pair_t*>(MR + off + c);
                        mq2 = *reinterpret_cast<const UMPA_GLOBAL pair_t*>(MQ + off + c);
                    }
                    term(wrow[c], r2[0], q2[0], mr2[0], mq2[0]);
                    term(wrow[c + 1], r2[1], q2[1], mr2[1], mq2[1]);
                };
                if constexpr (NWC > 0) {
#pragma unroll
                    for (int c = 0; c < 2 * NWC; c += 2) pair_step(c);
                    b = 2 * NWC;
                } else {
                    for (; b + 1 < S; b += 2) pair_step(b);
                }
            }
            if (KIND == 2 && blur && NWC > 0) {
                // a window row of the pixel's blurred footprint (blur_tiles / blur_footprint: rows / columns relative to
                // (i, j) - halo): all its loads issued before the first term, the terms summed in the reference's order
                const int halo = (blur_F - 1) >> 1;
                const UMPA_GLOBAL double* brow = blur + ((size_t)k * blur_F * blur_F + (size_t)(ri - i + halo - Nw + a) * blur_F + (rj - j + halo - Nw)) * kstride;
                double rv[2 * NWC + 1], qv[2 * NWC + 1], mrv[MASK ? 2 * NWC + 1 : 1], mqv[MASK ? 2 * NWC + 1 : 1];
#pragma unroll
                for (int c = 0; c < 2 * NWC + 1; c++) {
                    rv[c] = brow[(size_t)c * kstride];
                    qv[c] = Q[off + c];
                    if (MASK) { mrv[c] = MR[off + c]; mqv[c] = MQ[off + c]; }
                }
#pragma unroll
                for (int c = 0; c < 2 * NWC + 1; c++) term(wrow[c], rv[c], qv[c], MASK ? mrv[c] : 0.0, MASK ? mqv[c] : 0.0);
                b = S;
            }
            for (; b < S; b++) {
                double r;
                if (KIND == 2 && blur) {                 // the pixel's blurred footprint (blur_footprint): rows / columns relative to (i, j) - halo
                    const int halo = (blur_F - 1) >> 1;
                    r = blur[((size_t)k * blur_F * blur_F + (size_t)(ri - i + halo - Nw + a) * blur_F + (rj - j + halo - Nw + b)) * kstride];
                } else if (KIND == 2) r = blur_at<MASK>(gp(f.ref), gp(f.mask), f.W, ri - f.pi - Nw + a, rj - f.pj - Nw + b, kern, kstride);
                else r = R[off + b];
                term(wrow[b], r, Q[off + b], MASK ? MR[off + b] : 0.0, MASK ? MQ[off + b] : 0.0);
            }
        }
        if (KIND == 1) {                                 // Model.cpp:739,:770-772 / :808,:843-845
            const double mean = (MASK ? sm : s6) / m.win_sum;
            t2 += MASK ? mean * mean * s2 : mean * mean;
            t4 += mean * s4;
            t6 += mean * s6;
        }
    }

    if (KIND == 1) {                                     // Model.cpp:849-858
        const double det = t2 * t3 - t6 * t6;
        const double K = (t2 * t5 - t4 * t6) / det;
        const double beta = (t3 * t4 - t5 * t6) / det;
        fit.t = beta + K;
        fit.v = K / fit.t;
        cost = (t1 + beta * beta * t2 + K * K * t3 - 2 * beta * t4 - 2 * K * t5 + 2 * beta * K * t6) / wt;
    } else {                                             // Model.cpp:502-505
        fit.t = t5 / t3;
        fit.v = 0.0;
        cost = (t1 - t5 * fit.t) / wt;
    }
    return UMPA_ST_OK;
}

// Store one pixel's results the way Model*::min + the Cython loop do
// (Model.cpp:573-577 / :934-938; model.pyx:487-491).
template <class Memo>
__device__ __forceinline__ void store_pixel(const RegionArgs& A, size_t px, int kind, const Walk& w,
                                            Memo memo, const double* nb)
{
    UMPA_GLOBAL double* v = gpw(A.values) + px * A.v_px;
    v[0] = w.out;
    v[A.v_k] = w.live.t;
    v[2 * A.v_k] = w.uv1;
    v[3 * A.v_k] = w.uv0;
    if (kind == 1) v[4 * A.v_k] = w.live.v;
    if (A.uv) { gpw(A.uv)[2 * px] = w.uv0; gpw(A.uv)[2 * px + 1] = w.uv1; }
    gpw(A.err)[px] = w.status & UMPA_ST_OK;
    if (A.dbg_n) gpw(A.dbg_n)[px] = w.n;
    if (A.dbg_d) for (int q = 0; q < 25; q++) gpw(A.dbg_d)[px * 25 + q] = walk_memo_cell(w, memo, q);
    if (A.dbg_a) for (int q = 0; q < 16; q++) gpw(A.dbg_a)[px * 16 + q] = nb[q];
}

// blockIdx -> tile remap so that each XCD (blocks b, b+8, ... share one) works on a
// contiguous band of tile rows and keeps the halo rows of neighbouring tiles in its own L2.
__device__ __forceinline__ int xcd_band_remap(int lin, int total)
{
    const int per = (total + 7) >> 3;
    return (lin & 7) * per + (lin >> 3);
}

#ifndef UMPA_DIRECT_BX
#define UMPA_DIRECT_BX 64
#define UMPA_DIRECT_BY 4
#endif

#define UMPA_WALK_THREADS (UMPA_DIRECT_BX * UMPA_DIRECT_BY)

template <int KIND, bool MASK, int NWC>
__global__ void __launch_bounds__(UMPA_WALK_THREADS, 3)
match_direct_kernel(ModelDev m, RegionArgs A, int nbx, int nby)
{
    __shared__ double memo_lds[25 * UMPA_WALK_THREADS];
    const int lin = xcd_band_remap(blockIdx.x, nbx * nby);
    if (lin >= nbx * nby) return;
    const int bx = lin % nbx, by = lin / nbx;
    const int xj = bx * UMPA_DIRECT_BX + threadIdx.x;
    const int xi = by * UMPA_DIRECT_BY + threadIdx.y;
    if (xi >= A.N0 || xj >= A.N1) return;
    const size_t px = (size_t)xi * A.N1 + xj;
    if (A.cover && gp(A.cover)[px] < A.thr) return;      // model.pyx:480-481: skipped pixels keep their zeros

    const int i = A.org0 + A.step0 * xi, j = A.org1 + A.step1 * xj;
    const LdsMemo<UMPA_WALK_THREADS> memo = {memo_lds + threadIdx.y * UMPA_DIRECT_BX + threadIdx.x};
    UMPA_GLOBAL double* kern = nullptr;
    if (KIND == 2) {                                     // Model.cpp:1228-1229: kernel from values[4..6]
        kern = gpw(A.kern) + ((size_t)(xi - A.row_base) * A.N1 + xj);
        if (!A.blur_ready) {
            const UMPA_GLOBAL double* v = gp(A.values) + px * A.v_px;
            build_blur_kernel(kern, A.kern_stride, v[4 * A.v_k], v[5 * A.v_k], v[6 * A.v_k]);
        }
    }
    UMPA_GLOBAL double* blur = nullptr;
    if (KIND == 2 && A.blur) {                           // the blurred reference this pixel can ever read, once
        blur = gpw(A.blur) + ((size_t)(xi - A.row_base) * A.N1 + xj);
        if (!A.blur_ready) blur_footprint<MASK>(m, i, j, kern, A.kern_stride, blur, A.blur_F, (A.blur_F - 1) >> 1);
    }
    Walk w;
    walk_begin(w, memo, A.uv ? gp(A.uv)[2 * px] : 0.0, A.uv ? gp(A.uv)[2 * px + 1] : 0.0);
    while (w.phase < PH_FIT) {
        double c = 0.0;
        Fit fit = w.live;
        const int st = eval_direct<KIND, MASK, NWC>(m, i, j, w.req_i, w.req_j, c, fit, kern, A.kern_stride, blur, A.blur_F);
        walk_feed(w, memo, st, c, fit, m.call_cap);
    }
    double nb[16];
    walk_finish(w, memo, m.subpx, nb);
    store_pixel(A, px, KIND, w, memo, nb);
}

// ------------------------------------------------------------------------------------------------
// The general path with the windows served from LDS ("staged" direct kernel).
//
// match_direct reads every window element of every evaluation through the L1/TA path, which is what bounds it
// (2.7 TB per C2 match at 51 of 64 B/clk/CU).  All windows a workgroup of BX x BY pixels can ever ask for lie in a
// fixed footprint: its pixel box widened by Nw for the stack whose window does not move and by Nw + max_shift - 1
// for the one that does.  Here the workgroup runs the walk in lockstep: for every round of requests it goes through
// the frames, stages the footprint of frame k (coalesced loads, once for all 256 lanes), and every lane sums its
// own two windows out of LDS -- the same terms in the same order as eval_direct, so the results are identical.
// Masks, per-frame positions and shapes, both coordinate conventions and steps work as there; the kernel-dark-field
// model and footprints that do not fit LDS (large steps) stay on match_direct.
// ------------------------------------------------------------------------------------------------
// (round 4, late: a box of 16 x 16 pixels instead of 64 x 4 -- its footprint is 36 x 36 instead of 24 x 84 elements per frame at
//  C2's window and search range, and a wave is a block of 16 x 4 pixels: C2's stack on this kernel 58.9 -> 52.4 ms, the 2 x 2
//  sample-stepping stack 58.2 -> 50.9; match_direct keeps 64 x 4: at coarse steps and for the kernel-dark-field model the square
//  box loses 8-20 %, profiles/r04_replay_blocks.txt)
#ifndef UMPA_STAGED_BX
#define UMPA_STAGED_BX 16
#define UMPA_STAGED_BY 16
#endif
#define UMPA_STAGED_THREADS (UMPA_STAGED_BX * UMPA_STAGED_BY)
#define UMPA_STAGED_TAIL 24          // fewer walking lanes than this (of 256): they finish without staging
#ifndef UMPA_STAGED_NR
#define UMPA_STAGED_NR 3             // a footprint has at most NR * BY = 48 rows ...
#define UMPA_STAGED_NC 3             // ... and NC * BX = 48 columns (staged_geometry sends anything larger to match_direct)
#endif

struct StagedGeom {
    int hq, hr, hm;                  // halo of the sample / reference / mask footprint around the pixel box
    int rowsQ, colsQ, rowsR, colsR, rowsM, colsM;
    int offR, offM, offW;            // LDS offsets (doubles) of the reference and mask footprints (sample at 0) and of the window
};

template <int KIND, bool MASK, int NWC>
__global__ void __launch_bounds__(UMPA_STAGED_THREADS)
match_staged_kernel(ModelDev m, RegionArgs A, int nbx, int nby, StagedGeom G)
{
    __shared__ double memo_lds[25 * UMPA_STAGED_THREADS];
    extern __shared__ __attribute__((aligned(16))) char staged_raw[];
    double* fq = reinterpret_cast<double*>(staged_raw);
    double* fr = fq + G.offR;
    double* fm = fq + G.offM;
    double* wl = fq + G.offW;                                        // the window: every lane reads the same weight (LDS broadcast)

    const int lin = xcd_band_remap(blockIdx.x, nbx * nby);
    if (lin >= nbx * nby) return;                                    // whole workgroup
    const int bx = lin % nbx, by = lin / nbx;
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int xj = bx * UMPA_STAGED_BX + tx, xi = by * UMPA_STAGED_BY + ty;
    const size_t px = (size_t)xi * A.N1 + xj;
    bool active = xi < A.N0 && xj < A.N1;
    if (active && A.cover && gp(A.cover)[px] < A.thr) active = false;   // model.pyx:480-481: skipped pixels keep their zeros
    const int i = A.org0 + A.step0 * xi, j = A.org1 + A.step1 * xj;
    const int gi0 = A.org0 + A.step0 * (by * UMPA_STAGED_BY), gj0 = A.org1 + A.step1 * (bx * UMPA_STAGED_BX);   // box origin
    const LdsMemo<UMPA_STAGED_THREADS> memo = {memo_lds + ty * UMPA_STAGED_BX + tx};
    const int Nw = NWC > 0 ? NWC : m.Nw, S = 2 * Nw + 1, pad = m.padding, ms = m.ms;

    for (int q = ty * UMPA_STAGED_BX + tx; q < S * S; q += UMPA_STAGED_THREADS) wl[q] = gp(m.win)[q];

    Walk w;
    walk_begin(w, memo, (active && A.uv) ? gp(A.uv)[2 * px] : 0.0, (active && A.uv) ? gp(A.uv)[2 * px + 1] : 0.0);
    if (!active) w.phase = PH_DONE;

    // Staged rounds while a fair share of the workgroup is still walking; the few long walks left over (a round costs
    // the same for 1 lane as for 256) finish on their own through eval_direct -- same terms, same order, same numbers.
    while (__syncthreads_count(w.phase < PH_FIT) >= UMPA_STAGED_TAIL) {
        const bool ev = w.phase < PH_FIT;
        const int si = w.req_i, sj = w.req_j;
        int st = UMPA_ST_OK;                                         // Model.cpp:372-399 / :654-681
        if (si <= -ms || si >= ms) st = UMPA_ST_BOUND;
        else if (sj <= -ms) st = UMPA_ST_BOUND | UMPA_ST_DIM;
        else if (sj >= ms) st = UMPA_ST_BOUND | UMPA_ST_DIM | UMPA_ST_POSITIVE;
        const bool sum = ev && st == UMPA_ST_OK;
        int ri = i, rj = j, qi = i, qj = j;                          // Model.cpp:408-421 / :688-701
        if (m.ref_mode) { qi -= si; qj -= sj; } else { ri += si; rj += sj; }
        // window origins inside the staged footprints
        const int q0 = (qi - Nw - (gi0 - G.hq)) * G.colsQ + (qj - Nw - (gj0 - G.hq));
        const int r0 = (ri - Nw - (gi0 - G.hr)) * G.colsR + (rj - Nw - (gj0 - G.hr));
        const int mq0 = (qi - Nw - (gi0 - G.hm)) * G.colsM + (qj - Nw - (gj0 - G.hm));
        const int mr0 = (ri - Nw - (gi0 - G.hm)) * G.colsM + (rj - Nw - (gj0 - G.hm));
        double t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
        double wt = MASK ? 0.0 : (double)m.Na;                       // Model.cpp:425,:711 / :463,:777

        // ---- footprints of frame k: loaded into registers one frame ahead (the loads fly while frame k-1 is summed),
        // written to LDS when that is done.  Frame coordinates = image coordinates - position, clamped at the frame
        // edges: what is clamped is never read by a lane the frame contributes to.  Thread (tx, ty) owns the elements
        // (ty + 4 n, tx + 64 n').
        double vq[UMPA_STAGED_NR][UMPA_STAGED_NC], vr[UMPA_STAGED_NR][UMPA_STAGED_NC], vm[MASK ? UMPA_STAGED_NR : 1][UMPA_STAGED_NC];
        auto fetch = [&](int k) {
            const FrameDesc f = load_frame(m.frames, k);
            const int fi0 = gi0 - f.pi, fj0 = gj0 - f.pj;
#pragma unroll
            for (int nr = 0; nr < UMPA_STAGED_NR; nr++) {
                const int r = ty + nr * UMPA_STAGED_BY;
                const size_t rowq = (size_t)min(max(fi0 - G.hq + r, 0), f.H - 1) * f.W;
                const size_t rowr = (size_t)min(max(fi0 - G.hr + r, 0), f.H - 1) * f.W;
                const size_t rowm = (size_t)min(max(fi0 - G.hm + r, 0), f.H - 1) * f.W;
#pragma unroll
                for (int nc = 0; nc < UMPA_STAGED_NC; nc++) {
                    const int c = tx + nc * UMPA_STAGED_BX;
                    vq[nr][nc] = (r < G.rowsQ && c < G.colsQ) ? gp(f.sam)[rowq + min(max(fj0 - G.hq + c, 0), f.W - 1)] : 0.0;
                    vr[nr][nc] = (r < G.rowsR && c < G.colsR) ? gp(f.ref)[rowr + min(max(fj0 - G.hr + c, 0), f.W - 1)] : 0.0;
                    if (MASK) vm[nr][nc] = (r < G.rowsM && c < G.colsM) ? gp(f.mask)[rowm + min(max(fj0 - G.hm + c, 0), f.W - 1)] : 0.0;
                }
            }
        };
        auto deposit = [&]() {
#pragma unroll
            for (int nr = 0; nr < UMPA_STAGED_NR; nr++) {
                const int r = ty + nr * UMPA_STAGED_BY;
#pragma unroll
                for (int nc = 0; nc < UMPA_STAGED_NC; nc++) {
                    const int c = tx + nc * UMPA_STAGED_BX;
                    if (r < G.rowsQ && c < G.colsQ) fq[r * G.colsQ + c] = vq[nr][nc];
                    if (r < G.rowsR && c < G.colsR) fr[r * G.colsR + c] = vr[nr][nc];
                    if (MASK && r < G.rowsM && c < G.colsM) fm[r * G.colsM + c] = vm[nr][nc];
                }
            }
        };
        fetch(0);
        for (int k = 0; k < m.Na; k++) {
            const FrameDesc f = load_frame(m.frames, k);
            deposit();
            if (k + 1 < m.Na) fetch(k + 1);
            __syncthreads();
            const int li = i - f.pi, lj = j - f.pj;                  // Model.cpp:430-433 / :716-719
            if (sum && !(li - pad < 0 || li + pad > f.H || lj - pad < 0 || lj + pad > f.W)) {
                const double* __restrict__ Q = fq + q0;
                const double* __restrict__ R = fr + r0;
                const double* __restrict__ MQ = fm + mq0;
                const double* __restrict__ MR = fm + mr0;
                double s2 = 0, s4 = 0, s6 = 0, sm = 0;
                // one window element: the arithmetic and the order of Model.cpp:746-772 (and :813-841 with masks)
                auto term = [&](double wv, double r, double q, double mr, double mq) {
                    if (MASK) {
                        if (KIND == 1) sm += wv * r;                 // the ref mean is never mask-weighted (Model.cpp:804)
                        wv *= pair_weight(mr, mq);
                        wt += wv;
                        s2 += wv;
                    }
                    const double wq = wv * q, wr = wv * r;
                    t1 += wq * q;
                    t3 += wr * r;
                    t5 += wr * q;
                    if (KIND == 1) { s4 += wq; s6 += wr; }
                };
                // a window row at a time: all its LDS reads are issued before the first term is summed (an in-order
                // wave would otherwise pay the LDS latency once per element)
                constexpr int CH = NWC > 0 ? 2 * NWC + 1 : 8;
                for (int a = 0; a < S; a++) {
                    const double* wrow = wl + a * S;
                    const int oq = a * G.colsQ, orr = a * G.colsR, om = a * G.colsM;
                    for (int b0 = 0; b0 < S; b0 += CH) {
                        double wv[CH], rv[CH], qv[CH], mrv[CH], mqv[CH];
#pragma unroll
                        for (int b = 0; b < CH; b++) {
                            const bool in = NWC > 0 || b0 + b < S;
                            wv[b] = in ? wrow[b0 + b] : 0.0;
                            rv[b] = in ? R[orr + b0 + b] : 0.0;
                            qv[b] = in ? Q[oq + b0 + b] : 0.0;
                            mrv[b] = (MASK && in) ? MR[om + b0 + b] : 0.0;
                            mqv[b] = (MASK && in) ? MQ[om + b0 + b] : 0.0;
                        }
#pragma unroll
                        for (int b = 0; b < CH; b++)
                            if (NWC > 0 || b0 + b < S) term(wv[b], rv[b], qv[b], mrv[b], mqv[b]);
                    }
                }
                if (KIND == 1) {                                     // Model.cpp:739,:770-772 / :808,:843-845
                    const double mean = (MASK ? sm : s6) / m.win_sum;
                    t2 += MASK ? mean * mean * s2 : mean * mean;
                    t4 += mean * s4;
                    t6 += mean * s6;
                }
            }
            __syncthreads();                                         // footprints free for the next frame
        }
        if (ev) {
            double cost = 0.0;
            Fit fit = w.live;
            if (st == UMPA_ST_OK) {
                if (KIND == 1) {                                     // Model.cpp:849-858
                    const double det = t2 * t3 - t6 * t6;
                    const double K = (t2 * t5 - t4 * t6) / det;
                    const double beta = (t3 * t4 - t5 * t6) / det;
                    fit.t = beta + K;
                    fit.v = K / fit.t;
                    cost = (t1 + beta * beta * t2 + K * K * t3 - 2 * beta * t4 - 2 * K * t5 + 2 * beta * K * t6) / wt;
                } else {                                             // Model.cpp:502-505
                    fit.t = t5 / t3;
                    fit.v = 0.0;
                    cost = (t1 - t5 * fit.t) / wt;
                }
            }
            walk_feed(w, memo, st, cost, fit, m.call_cap);
        }
    }
    while (w.phase < PH_FIT) {
        double c = 0.0;
        Fit fit = w.live;
        const int st = eval_direct<KIND, MASK, NWC>(m, i, j, w.req_i, w.req_j, c, fit);
        walk_feed(w, memo, st, c, fit, m.call_cap);
    }
    if (!active) return;
    double nb[16];
    walk_finish(w, memo, m.subpx, nb);
    store_pixel(A, px, KIND, w, memo, nb);
}

// Model*::cost_interface for one pixel (Model.cpp:533-542, :887-897): out = [cost, T, df, status]
template <int KIND, bool MASK>
__global__ void cost_one_kernel(ModelDev m, int i, int j, int si, int sj, double* out, double ka, double kb, double kc)
{
    double c = 0.0;
    Fit fit = {0.0, 0.0};
    UMPA_GLOBAL double* kern = gpw(out) + 8;                  // DFKernel: 289 doubles of scratch behind the results
    if (KIND == 2) build_blur_kernel(kern, 1, ka, kb, kc);    // Model.cpp:1187-1188
    const int st = eval_direct<KIND, MASK>(m, i, j, si, sj, c, fit, kern, 1);
    out[0] = c; out[1] = fit.t; out[2] = fit.v; out[3] = (double)st;
}

// ModelBase::coverage over a region (Model.cpp:273-314 inside the loop of model.pyx:524-528)
__global__ void coverage_kernel(ModelDev m, int org0, int step0, int N0, int org1, int step1, int N1,
                                int has_mask, double* out)
{
    const int xj = blockIdx.x * blockDim.x + threadIdx.x;
    const int xi = blockIdx.y * blockDim.y + threadIdx.y;
    if (xi >= N0 || xj >= N1) return;
    const int i = org0 + step0 * xi, j = org1 + step1 * xj, pad = m.padding;
    double c = 0.0;
    for (int k = 0; k < m.Na; k++) {
        const FrameDesc f = load_frame(m.frames, k);
        const int li = i - f.pi, lj = j - f.pj;
        if (li - pad < 0 || li + pad > f.H || lj - pad < 0 || lj + pad > f.W) continue;
        c += has_mask ? gp(f.mask)[(size_t)li * f.W + lj] : 1.0;
    }
    gpw(out)[(size_t)xi * N1 + xj] = c;
}

// Flat-field correction fused into the upload of a projection (UMPA/umpa_multi.py:144: sam = (proj - dark) / flat[refnum]):
// `raw` is the detector frame as it came over PCIe (float64, float32 or uint16 counts), `out` the model's float64
// sample frame.  dark / flat may be NULL (plain conversion).
template <class T>
__global__ void __launch_bounds__(256)
flat_correct_kernel(const T* __restrict__ raw, const double* __restrict__ dark, const double* __restrict__ flat,
                    double* __restrict__ out, size_t n)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    double v = (double)gp(raw)[q];
    if (dark) v -= gp(dark)[q];
    if (flat) v /= gp(flat)[q];
    gpw(out)[q] = v;
}

__global__ void spfit_kernel(const double* a, double* io, int quad)
{
    double x = io[0], y = io[1];
    io[2] = quad ? spmin_quad(a, x, y) : spmin(a, x, y);
    io[0] = x; io[1] = y;
}


// ------------------------------------------------------------------------------------------------
// bad-pixel repair of a result map (UMPA/align.py:661-732), the epilogue of align.UMPA_normal / UMPA_nobias.
// `bad` marks the pixels of the ORIGINAL image outside [lo, hi]; one pass replaces each of them by the
// median of its 2*ndims neighbours in `src` (edges reflect: -1 -> 1, n -> n-2), everything else is copied.
// All reads of a pass come from `src` (numpy gathers all neighbours before it writes).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
badpix_mark_kernel(const double* __restrict__ img, unsigned char* __restrict__ bad, size_t n, double lo, double hi)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const double v = gp(img)[q];
    gpw(bad)[q] = (v < lo || v > hi) ? 1 : 0;
}

__device__ __forceinline__ void sort2(double& a, double& b) { const double lo = fmin(a, b), hi = fmax(a, b); a = lo; b = hi; }

__global__ void __launch_bounds__(256)
badpix_pass_kernel(const double* __restrict__ src, double* __restrict__ dst, const unsigned char* __restrict__ bad,
                   size_t nimg, int H, int W, int ndims)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t plane = (size_t)H * W;
    if (q >= nimg * plane) return;
    double v = gp(src)[q];
    if (gp(bad)[q]) {
        const size_t base = q / plane * plane;
        const int r = (int)((q - base) / W), c = (int)((q - base) % W);
        const int cl = c == 0 ? 1 : c - 1, ch = c + 1 == W ? W - 2 : c + 1;
        double a = gp(src)[base + (size_t)r * W + cl], b = gp(src)[base + (size_t)r * W + ch];
        if (ndims == 1) {
            v = (a + b) / 2.0;                                        // median of two = their mean
        } else {
            const int rl = r == 0 ? 1 : r - 1, rh = r + 1 == H ? H - 2 : r + 1;
            double cc = gp(src)[base + (size_t)rl * W + c], d = gp(src)[base + (size_t)rh * W + c];
            const bool nan = a != a || b != b || cc != cc || d != d;  // numpy.median: NaN if any
            sort2(a, b); sort2(cc, d); sort2(a, cc); sort2(b, d);     // a = min, d = max
            v = nan ? __builtin_nan("") : (b + cc) / 2.0;             // the two middle values, any order
        }
    }
    gpw(dst)[q] = v;
}

} // namespace umpa

// umpa_tiled.h -- the tiled fast path for the common case (no masks, all frames at position
// (0,0) with one shape, unit step, separable window): three kernels per row chunk.
//
// The reference evaluates the windowed cost lazily, ~18 times per pixel, each time as an
// explicit (2Nw+1)^2 x K sum (Model.cpp:709-774).  Results parity does not need that
// operation order, and the structure of the cost allows far less arithmetic:
//
//   * the window is an outer product h (x) h of 1-D Hamming vectors (model.pyx:692), so every
//     windowed sum W[.] is a separable filter;
//   * in the terms of Model.cpp:763-772 only  t5(p,u) = sum_k W[s_k r_k(.+u)](p)  couples the
//     pixel p and the shift u spatially;  t1 and W[s_k] depend on the sample-window position
//     only, t3, t2, t6 and the per-frame windowed reference mean on the reference-window
//     position only, and  t4(p,u) = sum_k mean_k(x_ref) W[s_k](x_sam)  is a K-term dot product.
//
//   prep_maps   : per frame separable window sums -> maps SamSq (t1), RefSq (t3) and per-frame
//                 WS_k = W[s_k], MR_k = W[r_k]/sum(w) (t2 and t6 are sums over k of MR_k^2).  HBM-bound.
//   corr_volume : the exhaustive table t5[u][p] for all (2 max_shift - 1)^2 integer shifts of a
//                 row chunk: product planes accumulated over frames in registers, then the two
//                 1-D filters through LDS.  K + 2(2Nw+1) FMAs per (p,u) before halo overhead.
//                 This is the dominant kernel (fp64 FMA / LDS bound).
//   replay_walk : one lane per pixel replays the reference's walk (umpa_walk.h) with every
//                 cost evaluation replaced by a table lookup + the closed-form solve of
//                 Model.cpp:849-858, so err / Ncalls / integer minimum stay bit-identical.
//
// LDS tiles are stored transposed ([column][row], row stride odd) so that lanes that walk
// along rows read and write consecutive 8-byte words (no bank conflicts for ds_read_b64).
#pragma once
#include "umpa_direct.h"
#include "umpa_corr.h"
#include "umpa_march.h"
#include "umpa_masked.h"
#include <mutex>
#include <vector>
#include <functional>

namespace umpa {

struct Maps {                        // prep_maps outputs, each a full H x W plane (borders unused)
    double* SamSq;                   // sum_k W[s_k^2]                      -> t1
    double* RefSq;                   // sum_k W[r_k^2]                      -> t3
    // Per-frame maps (DF), frames interleaved in PAIRS: element (k, x) of W[s_k] is WS[((k/2) * plane + x) * 2 + k%2]
    // (map_at), (K+1)/2 pair planes of 2 * plane doubles.  replay_walk reads two frames of a pixel with one 16-byte
    // load: 5 + 2 load instructions per evaluation at K = 10 instead of 10 + 2 (its time follows that count).
    double* WS;                      // W[s_k]
    double* MR;                      // mean_k = W[r_k]/sum(w)
    int H, W;                        // plane size = extent of the image the frames tile (max over frames of position + shape)
    // Frames of one shape at different positions (sample stepping): rows / columns of the image that lie inside EVERY
    // frame (patch reads are clamped to this box; inside the fully covered region nothing is clamped), and the frames'
    // common width.  Frame k's pixel (row, col) of the image is element (row - pi_k) * Wf + (col - pj_k) of its array.
    int br0, br1, bc0, bc1, Wf;
};

typedef double map_pair_t __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ size_t map_at(int k, size_t x, size_t plane) { return ((size_t)(k >> 1) * plane + x) * 2 + (k & 1); }

// ------------------------------------------------------------------------------------------------
// prep_maps
// ------------------------------------------------------------------------------------------------
template <int NW>
struct PrepCfg {
    static constexpr int T = UMPA_TILE, S = 2 * NW + 1, Q = T + 2 * NW, QP = Q | 1, NT = 256, CB = 8;
    static constexpr int RAW = Q * QP;                    // one transposed raw patch [c][r]
    static constexpr int HPL = T * QP;                    // one H-filtered plane [c < T][r < Q]
    // the sample and the reference patch of a frame go through the same LDS one after the other: one raw patch and
    // its two H-filtered planes (values, squares) -- 36 KB at Nw = 5, 41 KB at Nw = 7: three to four workgroups per CU
    // (with both stacks resident the C3 window needed 83 KB, one workgroup per CU, and ran at a third of the HBM rate)
    static constexpr size_t LDS = (size_t)(RAW + 2 * HPL) * sizeof(double);
};

template <int KIND, int NW>
// three workgroups per CU where the window leaves the registers for it (C2: 0.42 -> 0.375 ms; at Nw = 7 the 168-register
// cap spills and loses: C3 3.9 -> 4.3 ms)
__global__ void __launch_bounds__(256, NW <= 5 ? 3 : 2)
prep_maps_kernel(ModelDev m, Maps M, Sep1D sep, int ntx, int nty, int sides, int tx0, int ty0)
{
    // sides: bit 0 = the sample-side maps (SamSq, WS_k), bit 1 = the reference-side maps (RefSq, MR_k).
    // A model whose reference stack has not changed since the last match only recomputes the sample side.
    const bool do_ref = (sides & 2) != 0;
    using C = PrepCfg<NW>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* raw = reinterpret_cast<double*>(smem_raw);
    double* hpl = raw + C::RAW;                            // 2 planes: H-filtered values, H-filtered squares

    const int lin = xcd_band_remap(blockIdx.x, ntx * nty);            // (tiles to the XCDs in turn instead: C2 0.374 -> 0.364 ms, but 0.70-0.75 -> 0.81 on a slab of C4)
    if (lin >= ntx * nty) return;
    const int tx = tx0 + lin % ntx, ty = ty0 + lin / ntx;              // (tx0, ty0): first tile of the rectangle the match needs
    // outputs live on x in [NW, H-NW) x [NW, W-NW); this tile's first output pixel:
    const int r0 = NW + ty * C::T, c0 = NW + tx * C::T;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)M.H * M.W;

    // V-stage ownership: item = (sq, rb, c), c fastest: 2 planes x 4 row blocks x 32 columns = 256 items, one per
    // thread and stack (wave-uniform sq).  acc[st]: sums over the frames for stack st (0 sample, 1 reference).
    const int vc = tid & 31, vrb = (tid >> 5) & 3, vsq = tid >> 7;
    // one register set for both roles: the threads of the squares (vsq = 1) sum over the frames in it, the threads of
    // the values (vsq = 0) park the even frame of a pair in it
    double acc[2][C::CB];
#pragma unroll
    for (int o = 0; o < C::CB; o++) acc[0][o] = acc[1][o] = 0.0;

    // staging slots of this thread: compile-time count, so the frame k+1 can wait in registers while frame k is filtered
    constexpr int NS = (C::Q * C::Q + C::NT - 1) / C::NT;
    int s_lds[NS], s_g[NS];
    {
#pragma unroll
        for (int n = 0; n < NS; n++) {
            const int it = tid + n * C::NT, c = it % C::Q, r = it / C::Q;
            s_lds[n] = it < C::Q * C::Q ? c * C::QP + r : -1;
            s_g[n] = min(max(r0 - NW + r, M.br0), M.br1) * M.Wf + min(max(c0 - NW + c, M.bc0), M.bc1);
        }
    }
    // one patch waits in registers while the one before it is filtered: sample k, reference k, sample k+1, ...
    double pp[NS];
    auto fetch = [&](int k, int st) {
        const FrameDesc f = load_frame(m.frames, k);
        const long shift = (long)f.pi * M.Wf + f.pj;                  // image coordinates -> this frame's array
        const UMPA_GLOBAL double* src = gp(st ? f.ref : f.sam);
#pragma unroll
        for (int n = 0; n < NS; n++) pp[n] = src[s_g[n] - shift];
    };
    fetch(0, 0);
    for (int k = 0; k < m.Na; k++) {
#pragma unroll
        for (int st = 0; st < 2; st++) {                              // 0: sample patch, 1: reference patch
            if (st == 1 && !do_ref) break;
            __syncthreads();                                          // every reader of the LDS region is done
            // stage the raw patch, transposed; the global reads (issued one patch ahead) are coalesced along columns
#pragma unroll
            for (int n = 0; n < NS; n++)
                if (s_lds[n] >= 0) raw[s_lds[n]] = pp[n];
            if (st == 0 && do_ref) fetch(k, 1);
            else if (k + 1 < m.Na) fetch(k + 1, 0);
            __syncthreads();
            // H stage (along columns): items (cb, r), r fastest: 4 x Q <= 256
            if (tid < 4 * C::Q) {
                const int r = tid % C::Q, cb = tid / C::Q;
                const double* src = raw + (cb * C::CB) * C::QP + r;
                double lin_[C::CB], sq_[C::CB];
#pragma unroll
                for (int o = 0; o < C::CB; o++) { lin_[o] = 0.0; sq_[o] = 0.0; }
#pragma unroll
                for (int t = 0; t < C::CB + C::S - 1; t++) {
                    const double v = src[t * C::QP];
                    const double v2 = v * v;
#pragma unroll
                    for (int o = 0; o < C::CB; o++) {
                        const int tap = t - o;
                        if (tap >= 0 && tap < C::S) { lin_[o] = fma(sep.hc[tap], v, lin_[o]); sq_[o] = fma(sep.hc[tap], v2, sq_[o]); }
                    }
                }
                double* dl = hpl + (cb * C::CB) * C::QP + r;
                double* dq = dl + C::HPL;
#pragma unroll
                for (int o = 0; o < C::CB; o++) { dl[o * C::QP] = lin_[o]; dq[o * C::QP] = sq_[o]; }
            }
            __syncthreads();
            // V stage (along rows): vsq = 0 the windowed values (W[s_k] / W[r_k]), vsq = 1 the windowed squares
            if (!(KIND == 0 && vsq == 0)) {                           // NoDF needs only the squares
                double out[C::CB];
                fir_block<NW, C::CB>(hpl + vsq * C::HPL + vc * C::QP + vrb * C::CB, 1, sep.hr, out);
                const int gc = c0 + vc;
                // the per-frame maps are stored as frame PAIRS (Maps): an even frame waits in registers for its odd
                // partner and the two go out as one 16-byte store; the last frame of an odd count goes out alone
                const bool odd = (k & 1) != 0, last = k + 1 == m.Na;
                UMPA_GLOBAL double* dstmap = gpw(st == 0 ? M.WS : M.MR);
#pragma unroll
                for (int o = 0; o < C::CB; o++) {
                    const int gr = r0 + vrb * C::CB + o;
                    const bool inside = gr < M.H - NW && gc < M.W - NW;
                    if (vsq == 1) acc[st][o] += out[o];               // t1 / t3
                    else {
                        const double v = st == 0 ? out[o] : out[o] / m.win_sum;                  // Model.cpp:739
                        if (odd) {
                            if (inside) {
                                map_pair_t pv; pv[0] = acc[st][o]; pv[1] = v;
                                *reinterpret_cast<UMPA_GLOBAL map_pair_t*>(dstmap + map_at(k - 1, (size_t)gr * M.W + gc, plane)) = pv;
                            }
                        } else if (last) {
                            if (inside) dstmap[map_at(k, (size_t)gr * M.W + gc, plane)] = v;
                        } else acc[st][o] = v;
                    }
                }
            }
        }
    }
    const int gc = c0 + vc;
#pragma unroll
    for (int o = 0; o < C::CB; o++) {
        const int gr = r0 + vrb * C::CB + o;
        if (gr >= M.H - NW || gc >= M.W - NW) continue;
        const size_t g = (size_t)gr * M.W + gc;
        if (vsq == 1) {
            gpw(M.SamSq)[g] = acc[0][o];
            if (do_ref) gpw(M.RefSq)[g] = acc[1][o];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// replay_walk
// ------------------------------------------------------------------------------------------------
struct ReplayArgs {
    const double* table;      // [(2ms-1)^2][drows][N1d]: the DENSE (unit-step) grid under the requested region
    size_t slot_stride;       // drows * N1d
    int drow0, N1d;           // first dense row held by the table, doubles per table row (the dense row length, padded: CorrArgs::pitch)
    int strip_w, tw, drows;   // strip_w > 0: the table is corr_march's, blocked by column strips of strip_w dense columns:
                              // [strip][drows][(2ms-1)^2][tw]; slot_stride = tw
    int row0, rows;           // OUTPUT rows [row0, row0+rows) whose dense rows the table holds
    int bw_log2;              // the plain kernel (no on-demand stages): a wave walks a block of 2^bw_log2 x 2^(6 - bw_log2) pixels.
                              // Round 4, late: 16 x 4 on corr_volume's table (C2: replay_walk 1.17 -> 0.91 ms; 32 x 2: 0.96, 8 x 8: 1.25),
                              // 32 x 2 on corr_march's strip-blocked one (C3: 8.34 -> 7.78 ms; 16 x 4: 13.8) -- the walks of a block
                              // share a smaller window of the maps and the table than those of 64 pixels in a row
    int ablate;               // diagnostics only (UMPA_HIP_ABLATE_REPLAY): 1 = 18 fixed lookups instead of the walk, 2 = no sub-pixel fit
};

#define UMPA_KFIX 16          // frames whose fixed-window map value is kept in registers by replay_walk (any frame count)
#define UMPA_KTEMPL 24        // largest frame count with its own replay_walk instantiation (all of them in registers)

// One "cost evaluation" of the walk: table lookup + maps + the closed-form solve of Model.cpp:849-858.
// `fixed[k]` holds, for k < UMPA_KFIX, the per-frame map at the window that does not move with the shift
// (W[s_k](p) in 'sam' mode, mean_k(p) in 'ref' mode); `movmap` is the other per-frame map.
// load base[byte_off / 8]: a wave-uniform base with a 32-bit per-lane byte offset is the
// `global_load v, v_off, s[base:base+1]` addressing mode -- no 64-bit address arithmetic on the VALU
__device__ __forceinline__ double ld_off(const UMPA_GLOBAL double* base, unsigned byte_off)
{
    return *reinterpret_cast<const UMPA_GLOBAL double*>(reinterpret_cast<const UMPA_GLOBAL char*>(base) + byte_off);
}
__device__ __forceinline__ map_pair_t ld_pair_off(const UMPA_GLOBAL double* base, unsigned byte_off)    // 16-byte aligned
{
    return *reinterpret_cast<const UMPA_GLOBAL map_pair_t*>(reinterpret_cast<const UMPA_GLOBAL char*>(base) + byte_off);
}

// What an evaluation needs of the pixel itself, loaded or summed once before the walk: in 'sam' mode the sample window
// does not move (t1), in 'ref' mode the reference window does not (t3, t2, t6).
struct PixConst { double t1, t3, t2, t6; };

// t2 = sum_k mean_k^2 and t6 = sum_k mean_k W[r_k] (Model.cpp:770-772) from the per-frame means, which an evaluation
// has in registers anyway: W[r_k] = mean_k * sum(w) by the definition of the mean (Model.cpp:739), so t6 = sum(w) t2.
// (Two map planes and two cache-line streams per evaluation less than reading them back.)

// NA > 0: the number of frames is a compile-time constant (<= UMPA_KTEMPL) and every map plane is addressable
// with 32-bit byte offsets: straight-line code, no per-frame tests.  NA == 0: any frame count.
template <int KIND, int NA>
__device__ __forceinline__ int eval_lookup(const ModelDev& m, const Maps& M, const ReplayArgs& R, int ref_mode,
                                           int i, int j, size_t tpx, int si, int sj,
                                           const double* fixed, const PixConst& pc, double& cost, Fit& fit)
{
    // (no implicit contraction: a * b + c stays two roundings unless written as fma().  Left to the compiler, the inlined
    //  copies of this arithmetic -- here, in lookup_solve, in every kernel variant -- may be fused differently from one another:
    //  seen in round 4 when a restructured replay_walk answered T an ulp away from the on-demand kernel's on 27 k pixels)
#pragma clang fp contract(off)
    const int ms = m.ms;
    if (si <= -ms || si >= ms) return UMPA_ST_BOUND;
    if (sj <= -ms) return UMPA_ST_BOUND | UMPA_ST_DIM;
    if (sj >= ms) return UMPA_ST_BOUND | UMPA_ST_DIM | UMPA_ST_POSITIVE;
    const int UJ = 2 * ms - 1;
    const unsigned slot = (unsigned)((si + ms - 1) * UJ + (sj + ms - 1));
    // slot_stride < 2^32 (tiled_match bounds the row chunk): one 32x32->64-bit multiply-add
    // (a non-temporal load here is slower: 1.25 -> 1.32 ms on C2)
    const double t5 = gp(R.table)[(size_t)slot * (unsigned)R.slot_stride + tpx];
    // window positions (Model.cpp:688-701)
    const size_t xs = ref_mode ? (size_t)(i - si) * M.W + (j - sj) : (size_t)i * M.W + j;
    const size_t xr = ref_mode ? (size_t)i * M.W + j : (size_t)(i + si) * M.W + (j + sj);
    const double rwt = 1.0 / (double)m.Nwt;                         // wave-uniform: scalar
    double t1 = pc.t1, t3 = pc.t3;
    if (ref_mode) t1 = NA > 0 ? ld_off(gp(M.SamSq), (unsigned)xs * 8u) : gp(M.SamSq)[xs];
    else t3 = NA > 0 ? ld_off(gp(M.RefSq), (unsigned)xr * 8u) : gp(M.RefSq)[xr];
    if (KIND == 1) {
        double t2 = 0.0, t4 = 0.0;
        const size_t plane = (size_t)M.H * M.W;
        if (NA > 0) {
            const unsigned bm = (unsigned)(ref_mode ? xs : xr) * 16u;
            const UMPA_GLOBAL double* __restrict__ mov = gp(ref_mode ? M.WS : M.MR);
            map_pair_t mp[NA > 0 ? (NA + 1) / 2 : 1];
#pragma unroll
            for (int q = 0; q < (NA + 1) / 2; q++) mp[q] = ld_pair_off(mov + (size_t)q * 2 * plane, bm);   // all loads in flight together
#pragma unroll
            for (int k = 0; k < NA; k++) { const double a = mp[k >> 1][k & 1]; t4 = fma(a, fixed[k], t4); t2 = fma(a, a, t2); }
        } else {
            const UMPA_GLOBAL double* __restrict__ mov = gp(ref_mode ? M.WS : M.MR);
            const size_t xm = ref_mode ? xs : xr;
            double mv[UMPA_KFIX];
#pragma unroll
            for (int k = 0; k < UMPA_KFIX; k++) mv[k] = k < m.Na ? mov[map_at(k, xm, plane)] : 0.0;
#pragma unroll
            for (int k = 0; k < UMPA_KFIX; k++) if (k < m.Na) { t4 = fma(mv[k], fixed[k], t4); t2 = fma(mv[k], mv[k], t2); }
            if (m.Na > UMPA_KFIX) {
                const UMPA_GLOBAL double* __restrict__ fx = gp(ref_mode ? M.MR : M.WS);
                const size_t xf = ref_mode ? xr : xs;
                for (int k = UMPA_KFIX; k < m.Na; k++) { const double a = mov[map_at(k, xm, plane)]; t4 = fma(a, fx[map_at(k, xf, plane)], t4); t2 = fma(a, a, t2); }
            }
        }
        double t6;
        if (ref_mode) { t2 = pc.t2; t6 = pc.t6; }                   // the moving maps were the sample's: the means are the pixel's own
        else t6 = m.win_sum * t2;
        // Model.cpp:849-858 with one reciprocal instead of the reference's three divisions by the same
        // determinant and the division by wt (1-ulp level differences; the bar is 1e-5).  The dark-field
        // value v = K/T is only needed for the pixel's final answer: `fit.v` carries K, replay_walk divides once.
        const double rdet = fast_rcp(t2 * t3 - t6 * t6);
        const double K = (t2 * t5 - t4 * t6) * rdet;
        const double beta = (t3 * t4 - t5 * t6) * rdet;
        fit.t = beta + K;
        fit.v = K;
        cost = (t1 + beta * beta * t2 + K * K * t3 - 2 * beta * t4 - 2 * K * t5 + 2 * beta * K * t6) * rwt;
    } else {
        fit.t = t5 / t3;                                            // Model.cpp:502-505
        fit.v = 0.0;
        cost = (t1 - t5 * fit.t) * rwt;
    }
    return UMPA_ST_OK;
}

// The same evaluation in three pieces for the frame-count templates (NA > 0), so that a caller can have the loads of TWO
// evaluations in flight before it waits for either (replay_walk's speculative second lookup): status, loads, solve.
// lookup_solve repeats eval_lookup's arithmetic expression by expression: the numbers are the same.
__device__ __forceinline__ int lookup_status(int ms, int si, int sj)
{
    if (si <= -ms || si >= ms) return UMPA_ST_BOUND;
    if (sj <= -ms) return UMPA_ST_BOUND | UMPA_ST_DIM;
    if (sj >= ms) return UMPA_ST_BOUND | UMPA_ST_DIM | UMPA_ST_POSITIVE;
    return UMPA_ST_OK;
}

template <int NA>
struct LookupRaw {
    double t5, sq;                                                  // table entry; SamSq ('ref' mode) or RefSq at the moving window
    map_pair_t mp[NA > 0 ? (NA + 1) / 2 : 1];                       // the moving per-frame maps (dark-field model)
};

template <int KIND, int NA>
__device__ __forceinline__ void lookup_load(const ModelDev& m, const Maps& M, const ReplayArgs& R, int ref_mode,
                                            int i, int j, size_t tpx, int si, int sj, LookupRaw<NA>& raw)   // |si|, |sj| < ms
{
    const int ms = m.ms, UJ = 2 * ms - 1;
    const unsigned slot = (unsigned)((si + ms - 1) * UJ + (sj + ms - 1));
    raw.t5 = gp(R.table)[(size_t)slot * (unsigned)R.slot_stride + tpx];
    const size_t xs = ref_mode ? (size_t)(i - si) * M.W + (j - sj) : (size_t)i * M.W + j;
    const size_t xr = ref_mode ? (size_t)i * M.W + j : (size_t)(i + si) * M.W + (j + sj);
    raw.sq = ref_mode ? ld_off(gp(M.SamSq), (unsigned)xs * 8u) : ld_off(gp(M.RefSq), (unsigned)xr * 8u);
    if (KIND == 1) {
        const size_t plane = (size_t)M.H * M.W;
        const unsigned bm = (unsigned)(ref_mode ? xs : xr) * 16u;
        const UMPA_GLOBAL double* __restrict__ mov = gp(ref_mode ? M.WS : M.MR);
#pragma unroll
        for (int q = 0; q < (NA + 1) / 2; q++) raw.mp[q] = ld_pair_off(mov + (size_t)q * 2 * plane, bm);
    }
}

template <int KIND, int NA>
__device__ __forceinline__ void lookup_solve(const ModelDev& m, int ref_mode, const LookupRaw<NA>& raw,
                                             const double* fixed, const PixConst& pc, double& cost, Fit& fit)
{
#pragma clang fp contract(off)                                      // (as eval_lookup: the same roundings in every inlined copy)
    const double t5 = raw.t5;
    const double rwt = 1.0 / (double)m.Nwt;
    double t1 = pc.t1, t3 = pc.t3;
    if (ref_mode) t1 = raw.sq; else t3 = raw.sq;
    if (KIND == 1) {
        double t2 = 0.0, t4 = 0.0;
#pragma unroll
        for (int k = 0; k < NA; k++) { const double a = raw.mp[k >> 1][k & 1]; t4 = fma(a, fixed[k], t4); t2 = fma(a, a, t2); }
        double t6;
        if (ref_mode) { t2 = pc.t2; t6 = pc.t6; }
        else t6 = m.win_sum * t2;
        const double rdet = fast_rcp(t2 * t3 - t6 * t6);
        const double K = (t2 * t5 - t4 * t6) * rdet;
        const double beta = (t3 * t4 - t5 * t6) * rdet;
        fit.t = beta + K;
        fit.v = K;
        cost = (t1 + beta * beta * t2 + K * K * t3 - 2 * beta * t4 - 2 * K * t5 + 2 * beta * K * t6) * rwt;
    } else {
        fit.t = t5 / t3;
        fit.v = 0.0;
        cost = (t1 - t5 * fit.t) * rwt;
    }
}

// Workgroup = UMPA_REPLAY_ROWS waves, each on a block of 64 pixels (ReplayArgs::bw_log2: 16 x 4, or 32 x 2 on corr_march's table; the
// lattice and parked-pixel modes of the on-demand stages: 64 list entries).  A workgroup keeps its LDS until its slowest wave has
// finished, so small workgroups refill the CU sooner (the walk lengths differ): one wave per workgroup (two: C2 0.93 against 0.91 ms).
#ifndef UMPA_REPLAY_ROWS
#define UMPA_REPLAY_ROWS 1
#endif
#define UMPA_REPLAY_THREADS (64 * UMPA_REPLAY_ROWS)
#ifndef UMPA_REPLAY_SPECULATE
#define UMPA_REPLAY_SPECULATE 1
#endif
// two evaluations in flight hold two sets of per-frame map values: beyond this many frames the registers run out (12: spills)
// (20 frames, C3: 8.7 -> 20.6 ms with the second lookup)
#define UMPA_REPLAY_SPECULATE_NA 11

// OD = false: the plain kernel (every table plane is there): the on-demand bookkeeping compiles away (it costs 20 VGPRs and
// 0.1 ms on C2 otherwise)
template <int KIND, int NA, bool OD>
__global__ void __launch_bounds__(UMPA_REPLAY_THREADS, 3)
replay_walk_kernel(ModelDev m, Maps M, ReplayArgs R, RegionArgs A, OdArgs od_in)
{
    OdArgs od = od_in;
    if (!OD) od.mode = 0;
    __shared__ double memo_lds[25 * UMPA_REPLAY_THREADS];
    const LdsMemo<UMPA_REPLAY_THREADS> memo = {memo_lds + threadIdx.y * 64 + threadIdx.x};
    // od.mode (umpa_ondemand.h): 0 every pixel, every table plane is there; 1 the pixels of the seed tiles (a compact grid;
    // which passes they read is recorded); 2 the other pixels (a walk that needs a missing pass parks its pixel); 3 a
    // persistent grid over the parked pixels' list.  No lane leaves before the end: od_record_visited shuffles across the wave.
    const int nparked = od.mode == 3 ? __builtin_amdgcn_readfirstlane(gp(od.cnt_in)[OD_C_PX]) : 0;
  for (int q = (blockIdx.y * gridDim.x + blockIdx.x) * UMPA_REPLAY_THREADS + threadIdx.y * 64 + threadIdx.x, first = 1;;
       q += gridDim.x * gridDim.y * UMPA_REPLAY_THREADS, first = 0) {
    int xi, xj;
    bool live;
    if (od.mode == 3) {
        if (__builtin_amdgcn_readfirstlane(q & ~63) >= nparked) break;   // whole wave
        live = q < nparked;
        const int pxq = live ? gp(od.px_in)[q] : 0;
        xi = pxq / A.N1; xj = pxq - xi * A.N1;
    } else if (od.mode == 1) {
        if (!first) break;
        live = od_seed_pixel(od, R.drow0, A.step0, A.step1, xi, xj);
        live = live && xi >= R.row0 && xi < R.row0 + R.rows && xj < A.N1;
    } else {
        if (!first) break;
        if (OD && od.sub > 1) {                                      // the sample lattice (a compact grid, 64 lattice pixels of a row per wave)
            xj = (blockIdx.x * 64 + threadIdx.x) * od.sub + (od.sub >> 1);
            xi = (blockIdx.y * UMPA_REPLAY_ROWS + threadIdx.y) * od.sub + (od.sub >> 1);
        } else {
            // a wave = a block of 2^bw_log2 x 2^(6 - bw_log2) pixels (ReplayArgs): the window of the maps and of the table that
            // its walks read is (rows + 2 ms) x (columns + 2 ms) positions -- 288 for 16 x 4 against 648 for 64 x 1 at C2.
            // (blocks onto XCD-contiguous bands of block rows -- xcd_band_remap -- is slower: C2 0.90 -> 0.98 ms, C3 7.7 -> 9.4; so are
            //  XCD-contiguous strips of block columns: 0.91 -> 1.08; see the grid's width at the launch)
            const int bwl = R.bw_log2;
            xj = (blockIdx.x << bwl) + (threadIdx.x & ((1 << bwl) - 1));
            xi = ((blockIdx.y * UMPA_REPLAY_ROWS + threadIdx.y) << (6 - bwl)) + (threadIdx.x >> bwl);
        }
        xi += R.row0;
        live = xi < R.row0 + R.rows && xj < A.N1;
    }
    const size_t px = (size_t)xi * A.pitch + xj;                     // in the output arrays
    size_t tpx = (size_t)(xi * A.step0 - R.drow0) * R.N1d + (size_t)xj * A.step1;
    if (R.strip_w > 0) {                                             // corr_march's strip-blocked table (umpa_march.h)
        const int dc = xj * A.step1, strip = dc / R.strip_w, UJr = 2 * m.ms - 1;
        tpx = (((size_t)strip * R.drows + (size_t)(xi * A.step0 - R.drow0)) * (size_t)(UJr * UJr)) * R.tw + (size_t)(dc - strip * R.strip_w);
    }
    if (live && A.cover && gp(A.cover)[px] < A.thr) live = false;
    OdLane L;
    if (live && !od_begin(od, xi * A.step0 - R.drow0, xj * A.step1, L)) live = false;
    if (live) {
    const int i = A.org0 + A.step0 * xi, j = A.org1 + A.step1 * xj;
    constexpr int NFIX = NA > 0 ? NA : UMPA_KFIX;
    double fixed[NFIX];
    if (KIND == 1) {
        const size_t plane = (size_t)M.H * M.W, x0 = (size_t)i * M.W + j;
        const UMPA_GLOBAL double* __restrict__ fx = gp(m.ref_mode ? M.MR : M.WS);
        if constexpr (NA > 0) {
#pragma unroll
            for (int qq = 0; qq < (NA + 1) / 2; qq++) {
                const map_pair_t v = *reinterpret_cast<const UMPA_GLOBAL map_pair_t*>(fx + ((size_t)qq * plane + x0) * 2);
                fixed[2 * qq] = v[0];
                if (2 * qq + 1 < NFIX) fixed[2 * qq + 1] = v[1];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NFIX; k++) fixed[k] = k < m.Na ? fx[map_at(k, x0, plane)] : 0.0;
        }
    }
    PixConst pc = {0.0, 0.0, 0.0, 0.0};
    {
        const size_t x0 = (size_t)i * M.W + j;
        if (m.ref_mode) {
            pc.t3 = gp(M.RefSq)[x0];
            if (KIND == 1) {                                        // the means at the pixel: `fixed` for k < NFIX, the planes beyond
#pragma unroll
                for (int k = 0; k < NFIX; k++) if (NA > 0 || k < m.Na) pc.t2 = fma(fixed[k], fixed[k], pc.t2);
                for (int k = NFIX; k < m.Na; k++) { const double a = gp(M.MR)[map_at(k, x0, (size_t)M.H * M.W)]; pc.t2 = fma(a, a, pc.t2); }
                pc.t6 = m.win_sum * pc.t2;
            }
        } else pc.t1 = gp(M.SamSq)[x0];
    }
    Walk w;
    walk_begin(w, memo, A.uv ? gp(A.uv)[2 * px] : 0.0, A.uv ? gp(A.uv)[2 * px + 1] : 0.0);
    if (R.ablate & 1) {                                             // diagnostics: the lookups without the walk
        double csum = 0.0;
        Fit fit = w.live;
        for (int n = 0; n < 18; n++) {
            double c = 0.0;
            eval_lookup<KIND, NA>(m, M, R, m.ref_mode, i, j, tpx, (n % 5) - 2, (n / 5) - 2, fixed, pc, c, fit);
            csum += c;
        }
        w.out = csum; w.live = fit; w.phase = PH_DONE; w.status = 1;
    }
    const int sigma = m.ref_mode ? -1 : 1;
    while (w.phase < PH_FIT) {
        if (!od_check(od, L, m.ms, sigma, w.req_i, w.req_j)) break;  // the plane is not there: this pixel is parked
        double c = 0.0;
        Fit fit = w.live;
        if constexpr (!OD && NA > 0 && NA <= UMPA_REPLAY_SPECULATE_NA && UMPA_REPLAY_SPECULATE) {
            // the walk is a chain of dependent lookups, each a round trip to the table: the request that will follow is
            // looked up beside the pending one (walk_speculate) and delivered if the walk then asks for it.  Straight-line
            // code: lanes without a (valid) second request repeat the first one's addresses, a request outside the search
            // range loads shift (0, 0) and is answered by its status alone.
            int si = w.req_i, sj = w.req_j;
            const bool spec = walk_speculate(w, si, sj);
            const int st = lookup_status(m.ms, w.req_i, w.req_j), st2 = lookup_status(m.ms, si, sj);
            const bool ok1 = st == UMPA_ST_OK, ok2 = st2 == UMPA_ST_OK;
            LookupRaw<NA> r1, r2;
            lookup_load<KIND, NA>(m, M, R, m.ref_mode, i, j, tpx, ok1 ? w.req_i : 0, ok1 ? w.req_j : 0, r1);
            lookup_load<KIND, NA>(m, M, R, m.ref_mode, i, j, tpx, ok2 ? si : 0, ok2 ? sj : 0, r2);
            double c2 = 0.0;
            Fit fit2 = w.live;
            lookup_solve<KIND, NA>(m, m.ref_mode, r1, fixed, pc, c, fit);
            lookup_solve<KIND, NA>(m, m.ref_mode, r2, fixed, pc, c2, fit2);
            walk_feed(w, memo, st, c, fit, m.call_cap);
            if (spec && w.phase < PH_FIT && w.req_i == si && w.req_j == sj) walk_feed(w, memo, st2, c2, fit2, m.call_cap);
        } else {
            const int st = eval_lookup<KIND, NA>(m, M, R, m.ref_mode, i, j, tpx, w.req_i, w.req_j, fixed, pc, c, fit);
            walk_feed(w, memo, st, c, fit, m.call_cap);
        }
    }
    if (L.miss) od_park(od, L, xi * A.N1 + xj);
    else if (!(OD && od.sub > 1)) {                                   // (the sample lattice's walks only predict: A.uv is read AND written)
        double nb[16];
        walk_finish(w, memo, (R.ablate & 2) ? 0 : m.subpx, nb);
        if (KIND == 1 && (w.live.t != 0.0 || w.live.v != 0.0))      // eval_lookup left K in the v slot; (0,0) = never evaluated
            w.live.v = w.live.v / w.live.t;                         // Model.cpp:854
        store_pixel(A, px, KIND, w, memo, nb);
    }
    }
    if (od.mode == 1) od_record_visited(od, L, live);
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct TiledState {
    double* maps = nullptr;   size_t maps_cap = 0;
    double* table = nullptr;  size_t table_cap = 0;
    bool table_limited = false;   // the budget's worth of table could not be allocated once: keep to the capacity we have
    Sep1D sep;
    bool separable = false;
    int sep_nw = -1;
    // the reference-side maps in `maps` are current for this (kind, frame count, plane size): set by tiled_match,
    // cleared by whoever changes the reference frames or the window
    bool ref_maps_ok = false;
    int ref_kind = -1, ref_K = 0;
    size_t ref_plane = 0;
    int ref_rect[4] = {0, 0, 0, 0};   // tiles (tx0, tx1, ty0, ty1) of the maps that were computed (PrepRect)
    // corr_march (umpa_march.h): the frames' 32-bit byte offsets from the two stacks' base addresses, [2][K] on the device
    unsigned* march_off = nullptr;  int march_off_cap = 0;
    std::vector<unsigned> march_off_host;
    // on-demand passes (umpa_ondemand.h): device scratch, and page-locked slots the counters of a timed match land in
    void* od_buf = nullptr;   size_t od_cap = 0;
    int* od_host = nullptr;   int od_slot = 0;
    // the last match: passes computed / passes of the exhaustive table, parked pixels, missed tiles (umpa_hip_last_stats)
    const int* stat_slots[64];  int stat_n = 0;
    double stat_total_passes = 0.0;
};
#define UMPA_OD_SLOTS 256             // ring of counter slots (UMPA_OD_NCNT ints each) in TiledState::od_host

// one entry per timed launch of a tiled match; the events come from the model's pool (`get`), so nothing is
// created per launch and a match split into any number of row chunks is recorded completely
struct TiledTimers {
    // fma: FMAs of the launch; or, where that depends on what the device decided (on-demand passes), `fma_per` per pass and
    // `counts` -> the counters of the chunk, copied to page-locked memory behind its last kernel (counts[OD_C_DONE] = passes computed)
    struct Entry { int name; hipEvent_t t0, t1; double fma; const int* counts; double fma_per; };
    std::vector<Entry> entries;
    std::function<hipEvent_t()> get;
};

inline bool tiled_supported(int Nw, int ms, int Na)
{
    return Nw >= 1 && Nw <= UMPA_MAX_NW && ms >= 1 && Na >= 1;
}

// factor win[a][b] = hr[a]*hc[b]; false if the window is not an outer product
inline bool tiled_factor_window(TiledState& st, const double* win, int Nw)
{
    const int S = 2 * Nw + 1;
    st.separable = false;
    st.ref_maps_ok = false;                                           // every map is a filter with this window
    st.sep_nw = Nw;
    if (Nw > UMPA_MAX_NW) return false;
    const double pivot = win[Nw * S + Nw];
    if (!(pivot > 0.0)) return false;
    double mx = 0.0;
    for (int a = 0; a < S; a++) { st.sep.hr[a] = win[a * S + Nw] / pivot; st.sep.hc[a] = win[Nw * S + a]; }
    for (int n = 0; n < S * S; n++) mx = fmax(mx, fabs(win[n]));
    for (int a = 0; a < S; a++)
        for (int b = 0; b < S; b++)
            if (fabs(st.sep.hr[a] * st.sep.hc[b] - win[a * S + b]) > 4e-16 * mx) return false;
    st.separable = true;
    return true;
}

inline void tiled_release(TiledState& st)
{
    if (st.maps) (void)hipFree(st.maps);
    if (st.table) (void)hipFree(st.table);
    if (st.od_buf) (void)hipFree(st.od_buf);
    if (st.od_host) (void)hipHostFree(st.od_host);
    if (st.march_off) (void)hipFree(st.march_off);
    st.march_off = nullptr; st.march_off_cap = 0; st.march_off_host.clear();
    st.maps = st.table = nullptr;
    st.od_buf = nullptr; st.od_host = nullptr; st.od_cap = 0;
    st.maps_cap = st.table_cap = 0;
    st.ref_maps_ok = false;
}

inline int pick_ub(int UJ)
{
    // batch width over the column shifts: fewest batches (each batch re-stages the frames), then least padding
    int best = 5, best_nb = 1 << 30, best_waste = 1 << 30;
    const int cand[4] = {9, 8, 7, 5};
    for (int q = 0; q < 4; q++) {
        const int ub = cand[q], nb = (UJ + ub - 1) / ub, w = nb * ub - UJ;
        if (nb < best_nb || (nb == best_nb && w < best_waste)) { best = ub; best_nb = nb; best_waste = w; }
    }
    return best;
}

inline int tiled_corr_shape()
{
    const char* e = getenv("UMPA_HIP_CORR_SHAPE");                    // tuning override, see launch_corr_shape; 0 = automatic
    return e ? atoi(e) : 0;
}

// serialises the one-time per-device kernel attribute calls (two host threads may match on one model type)
inline std::mutex& tiled_attr_mutex()
{
    static std::mutex mu;
    return mu;
}

inline OdCorr od_corr_args(const OdArgs& od)
{
    OdCorr c;
    c.done = od.done; c.items = od.items; c.nitems = od.cnt_out ? od.cnt_out + OD_C_ITEMS : nullptr;
    c.ndone = od.cnt0 ? od.cnt0 + OD_C_DONE : nullptr; c.mode = od.mode;
    c.r0 = od.r0; c.c0 = od.c0;
    c.nsx = od_seed_count(od.ntx, od.c0);
    c.nseed = c.nsx * od_seed_count(od.nty, od.r0);
    return c;
}

struct CorrLaunch {
    OdArgs od;                // od.mode 0: every pass (static grid); 3: seed tiles (compact static grid); 2: queue over od.items
    bool dry;                 // only report the geometry
    int tc, ub, nrow, nbatch, npass, ntx, nty;   // out: tile columns, column / row offsets per pass, ... of the shape that was picked
    double fma_per_pass;      // out: fp64 FMAs one (tile, pass) executes (roofline accounting)
};

inline int device_cu_count()
{
    static int n[64] = {};
    int devid = 0;
    (void)hipGetDevice(&devid);
    if (!n[devid & 63]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || v <= 0) v = 256;
        n[devid & 63] = v;
    }
    return n[devid & 63];
}

template <int NW, int UB, int TC, int NTG, int UI, int WPC, int NF, int RO = 1>
inline hipError_t launch_corr(const ModelDev& dev, CorrArgs A, const Sep1D& sep, hipStream_t s, CorrLaunch& L)
{
    using C = CorrCfg<NW, UB, TC, NTG, UI, WPC, NF, RO>;
    A.ntx = (A.N1 + TC - 1) / TC;
    A.nty = (A.rows + C::TR - 1) / C::TR;
    const int UJ = 2 * dev.ms - 1, nbatch = (UJ + UB - 1) / UB, npass = ((UJ + C::NROW - 1) / C::NROW) * nbatch;
    L.tc = TC; L.ub = UB; L.nrow = C::NROW; L.nbatch = nbatch; L.npass = npass; L.ntx = A.ntx; L.nty = A.nty;
    // fp64 FMAs of one (tile, pass): the products of the active threads over all frames, the column filter on QR rows
    // and the row filter on the tile, for every plane of the pass (roofline accounting)
    L.fma_per_pass = (double)C::QR * C::NQB * C::QB * UB * C::NROW * dev.Na +
                     (double)C::NPL * C::QR * TC * C::S + (double)C::NPL * C::TR * TC * C::S;
    if (L.dry) return hipSuccess;
    static bool attr_set[64] = {};                                    // the attribute is per device
    int devid = 0;
    (void)hipGetDevice(&devid);
    {
        std::lock_guard<std::mutex> lock(tiled_attr_mutex());
        if (!attr_set[devid & 63]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_volume_kernel<NW, UB, TC, NTG, UI, WPC, NF, RO>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
            if (e != hipSuccess) return e;
            attr_set[devid & 63] = true;
        }
    }
    const OdCorr oc = od_corr_args(L.od);
    if (L.od.mode == 2) {                                             // persistent grid over the work list: one workgroup per slot of the chip
        static bool qattr_set[64] = {};
        {
            std::lock_guard<std::mutex> lock(tiled_attr_mutex());
            if (!qattr_set[devid & 63]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_volume_queue_kernel<NW, UB, TC, NTG, UI, WPC, NF, RO>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
                if (e != hipSuccess) return e;
                qattr_set[devid & 63] = true;
            }
        }
        const int grid = ((device_cu_count() * WPC + 7) / 8) * 8;
        hipLaunchKernelGGL((corr_volume_queue_kernel<NW, UB, TC, NTG, UI, WPC, NF, RO>), dim3(grid), dim3(C::NT), C::LDS, s, dev, A, sep, oc);
    } else {
        const int nt = L.od.mode == 3 ? oc.nseed : A.ntx * A.nty;
        const int grid = 8 * ((nt + 7) / 8) * npass;
        if (nt > 0) hipLaunchKernelGGL((corr_volume_kernel<NW, UB, TC, NTG, UI, WPC, NF, RO>), dim3(grid), dim3(C::NT), C::LDS, s, dev, A, sep, oc);
    }
    return hipGetLastError();
}

// Workgroup shapes, first that fits (UMPA_HIP_CORR_SHAPE picks one by number):
//   id: tile columns, threads per group, groups (row offsets per pass), workgroups per CU, flush rounds.
// Measured on C2 (ms per corr_volume, round 2; the card's clock state moves all of them by +-5 %):
//   32x32 tiles / 256 threads (8-wide register blocks) / 2 per CU / two flush rounds, two-slot ring with the next
//   frame issued right after the products: 1.5-1.6 (the default for windows up to 11; three flush rounds +2 %, five
//   +7 %); with the issue at the head of the step 1.81-1.91 (three rounds; two 1.93, five or nine 1.88-1.91);
//   32x16 tiles / 256 threads / 2 per CU (four-slot ring) 1.95-2.18; 3 per CU with two flush rounds 2.05-2.18;
//   32x32 with 320 or 384 threads at 2 per CU 2.99-3.46 (168-register cap, spills); two or three row offsets per pass
//   sharing one staging (512 / 768 threads, one workgroup per CU) 1.68 / 3.5-4.6 (2.16-2.33 before the non-temporal
//   stores and the flush re-layout); 24-column tiles with a three-frame ring 1.94-2.04.
// C3 (window 15; 32x32 / 256 does not hold the halo rows): 24-column tiles / 256 threads / 2 per CU / two flush rounds,
// two-slot ring with the early issue 36.0; 32x32 / 512 threads / 1 per CU (four-slot ring, DMA instructions between the
// product rows) 37.0-38.2 with two flush rounds, 38.2-40.6 with one, 41.8 with three; the early issue on that ring
// 42-44; 384 threads 46.1-51.6.
// Per workgroup and pass (s_memtime, 32x32 / 256 / 2 per CU, issue at the head): frame loop 37 k cycles, flush 21 k
// (planes to LDS 3.5 k, column filter 5.2 k, write back 2.5 k, row filter + stores 9.6 k).
// (round 3: 32x32 / 512 threads (4 columns per product thread) / TWO per CU, two or three flush rounds: 1.72-1.73 against 1.70 for
// shape 1 on the same box -- twice the waves change nothing: the kernel is not short of waves)
// Shape 5 (round 3): 32x32 tiles / 512 threads / ONE per CU, every product thread accumulates THREE row offsets from one
// staging (RO = 3, 4 columns x UB offsets x 3 = up to 108 accumulators of the 128 a thread can hold), flush in rounds of three
// planes (as many rounds as column offsets).  A third of the L2 -> LDS traffic per plane: C2 1.64-1.72 -> 1.31-1.45 (same
// boxes).  Taken for stacks of at least 7 frames and windows up to 13 wide.
// Shape 6: the same on 24-column tiles for wider windows (C3: 34.4 -> 28.9 ms).
#define UMPA_CORR_SHAPES(X) X(5, 32, 512, 1, 1, 9, 3) X(6, 24, 512, 1, 1, 9, 3) X(1, 32, 256, 1, 2, 2, 1) X(2, 24, 256, 1, 2, 2, 1) X(3, 16, 256, 1, 2, 1, 1) X(4, 32, 512, 1, 1, 2, 1)
template <int NW, int UB>
inline hipError_t launch_corr_shape(const ModelDev& dev, const CorrArgs& A, const Sep1D& sep, hipStream_t s, CorrLaunch& L)
{
    const int want = tiled_corr_shape();
    // three row offsets per pass: enough frames for the staging it saves to outweigh its longer flush (5 frames, C5: 3.87
    // against 3.80 ms per projection with the two-per-CU shape).  An idle third of the last pass (2 max_shift - 1 not a
    // multiple of three) costs less than it saves (tools/shape_rate.py, C2's stack, shape 1 -> 5: max_shift 3: 0.69 -> 0.60 ms,
    // 4: 1.22 -> 1.10, 6: 3.27 -> 2.76, 7: 3.87 -> 3.33; window 7, max_shift 4: 1.12 -> 1.10; window 15, max_shift 6,
    // shape 2 -> 6: 4.43 -> 4.10)
    const bool ro3_pays = dev.Na >= 7;
    // (two row offsets per pass, 384 threads with 6 columns each -- the window of C3 does not fit the 512 x 4-column shape:
    //  C3 35.8 -> 34.3 ms; not kept, 4 % do not pay for another set of instantiations)
    const bool ro2_pays = false;
#define UMPA_TRY_SHAPE(id, TC, NTG, UI, WPC, NF, RO)                                                    \
    if constexpr (CorrCfg<NW, UB, TC, NTG, UI, WPC, (RO > 1 ? UB : NF), RO>::OK && (id != 6 || NW >= 7)) {   \
        if ((want == 0 && (RO == 1 || (RO == 3 && ro3_pays) || (RO == 2 && ro2_pays))) || want == id)       \
            return launch_corr<NW, UB, TC, NTG, UI, WPC, (RO > 1 ? UB : NF), RO>(dev, A, sep, s, L);          \
    }
    UMPA_CORR_SHAPES(UMPA_TRY_SHAPE)
#undef UMPA_TRY_SHAPE
    return launch_corr<NW, UB, 32, 512, 1, 1, 2>(dev, A, sep, s, L);
}

template <int NW>
inline hipError_t launch_corr_nw(int ub, const ModelDev& dev, const CorrArgs& A, const Sep1D& sep, hipStream_t s, CorrLaunch& L)
{
    if (ub == 9) {
        if constexpr (CorrCfg<NW, 9, 32, 512, 1, 1, 2>::OK) return launch_corr_shape<NW, 9>(dev, A, sep, s, L);
        ub = 8;
    }
    if (ub == 8) {
        if constexpr (CorrCfg<NW, 8, 32, 512, 1, 1, 2>::OK) return launch_corr_shape<NW, 8>(dev, A, sep, s, L);
        ub = 7;
    }
    if (ub == 7) {
        if constexpr (CorrCfg<NW, 7, 32, 512, 1, 1, 2>::OK) return launch_corr_shape<NW, 7>(dev, A, sep, s, L);
        ub = 5;
    }
    static_assert(CorrCfg<NW, 5, 32, 512, 1, 1, 2>::OK, "UB=5 must always fit");
    return launch_corr_shape<NW, 5>(dev, A, sep, s, L);
}

// ------------------------------------------------------------------------------------------------
// corr_march (umpa_march.h): the table kernel for wide windows
// ------------------------------------------------------------------------------------------------
// rows / columns of the image inside every frame, the frames' common width (tiled_match)
struct FrameBox { int r0, r1, c0, c1, Wf; int slack; };   // slack: 1 if 8 readable bytes follow every frame row's last column

// corr_volume's tiles carry a halo of 2 Nw rows and columns and restage it for every pass: at Nw = 7, max_shift = 8 (BASELINE
// config C3) it moves 147 GB from L2 into LDS per match -- which at the rate a CU's vector-memory path sustains IS its
// 28.9 ms.  corr_march stages 40 GB for the same table: 24.7 ms, untuned (tools/microbench/march_dev.hip).  At Nw = 5 the
// two are level (1.37 against 1.28 ms on C2) and corr_volume stays.
struct MarchPlan {
    bool ok;
    int nuy, npass, nstrips, wo, tw, nxb, nt;
    int npa, npb, da, db;
    unsigned a_slot, b_slot;
    size_t lds;
};
#define UMPA_MARCH_NT 768
#define UMPA_MARCH_NPT 4
#define UMPA_MARCH_LA 2

inline bool march_wanted(int Nw)
{
    const char* e = getenv("UMPA_HIP_MARCH");                         // 0: never, 1: wherever it is instantiated (Nw 6, 7)
    if (e) return atoi(e) != 0 && (Nw == 6 || Nw == 7);
    return Nw == 6 || Nw == 7;
}

inline MarchPlan march_plan(int Nw, int UJ, int K, int N1d)
{
    MarchPlan P;
    memset(&P, 0, sizeof(P));
    if (!(Nw == 6 || Nw == 7) || UJ > 33 || UJ < 3) return P;
    P.nxb = UJ <= 17 ? 4 : 8;
    const int nbe = 16 + P.nxb;
    P.wo = (63 - 2 * Nw) / 4 * 4;
    P.tw = (P.wo % 16 == 0) ? P.wo : 64;
    P.nstrips = (N1d + P.wo - 1) / P.wo;
    P.nt = UMPA_MARCH_NT;
    P.npa = (K + 1) / 2;
    P.npb = (K * 2 * nbe + 63) / 64;
    if (P.npa + P.npb > UMPA_MARCH_NPT * (P.nt / 64)) return P;       // more staging instructions per step than the waves hold offsets for
    P.a_slot = (unsigned)P.npa * 1024u;
    P.b_slot = (unsigned)P.npb * 1024u;
    for (int nuy = std::min(UJ, (P.nt / 16) / UJ); nuy >= 1; nuy--) {  // as many row offsets per pass as the waves and the LDS hold
        const size_t lds = (size_t)(UMPA_MARCH_LA + 1) * P.a_slot + (size_t)(nuy + UMPA_MARCH_LA) * P.b_slot;
        if (lds <= UMPA_LDS_BUDGET) { P.nuy = nuy; P.lds = lds; break; }
    }
    if (P.nuy < 1) return P;
    P.npass = (UJ + P.nuy - 1) / P.nuy;
    P.da = UMPA_MARCH_LA + 1; P.db = P.nuy + UMPA_MARCH_LA;
    P.ok = true;
    return P;
}

// the frames' byte offsets from the lowest frame address of each stack (positions folded in): false if a stack does not fit
// 32-bit offsets up to the last image row the kernel may address
inline bool march_offsets(const FrameDesc* hf, int K, int sigma, const FrameBox& box, std::vector<unsigned>& off,
                          const char*& baseA, const char*& baseB)
{
    off.assign(2 * (size_t)K, 0u);
    for (int st = 0; st < 2; st++) {
        const bool sam = (st == 0) == (sigma > 0);                    // A = the stack whose window does not move (umpa_corr.h)
        intptr_t lo = INTPTR_MAX, hi = INTPTR_MIN;
        for (int k = 0; k < K; k++) {
            const intptr_t p = (intptr_t)(sam ? hf[k].sam : hf[k].ref) - ((intptr_t)hf[k].pi * box.Wf + hf[k].pj) * 8;
            lo = std::min(lo, p); hi = std::max(hi, p);
        }
        const uint64_t span = (uint64_t)(hi - lo) + ((uint64_t)box.r1 + 2) * (uint64_t)box.Wf * 8u;
        if (span >= ((uint64_t)1 << 32) - 4096) return false;
        for (int k = 0; k < K; k++)
            off[(size_t)st * K + k] = (unsigned)((intptr_t)(sam ? hf[k].sam : hf[k].ref) - ((intptr_t)hf[k].pi * box.Wf + hf[k].pj) * 8 - lo);
        (st == 0 ? baseA : baseB) = (const char*)lo;
    }
    return true;
}

template <int NW, int NXB>
inline hipError_t launch_march_inst(const ModelDev& dev, const MarchArgs& A, const Sep1D& sep, const MarchPlan& P, hipStream_t s)
{
    auto kern = corr_march_kernel<NW, NXB, UMPA_MARCH_NPT, UMPA_MARCH_LA, UMPA_MARCH_NT, 3>;
    static bool attr_set[64] = {};
    int devid = 0;
    (void)hipGetDevice(&devid);
    {
        std::lock_guard<std::mutex> lock(tiled_attr_mutex());
        if (!attr_set[devid & 63]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, UMPA_LDS_BUDGET);
            if (e != hipSuccess) return e;
            attr_set[devid & 63] = true;
        }
    }
    const int nitems = A.nstrips * A.nbands;
    // (work list: a slot for every unit there could be; the workgroups past the list's end leave at once)
    const int grid = A.items ? 8 * ((nitems * A.npass + 7) / 8) : 8 * ((nitems + 7) / 8) * A.npass;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(UMPA_MARCH_NT), P.lds, s, dev, A, sep);
    return hipGetLastError();
}

// the fp64 FMAs of an exhaustive launch with A's strips and bands (roofline accounting)
inline void march_fma(const ModelDev& dev, const MarchArgs& A, double* fma)
{
    const int Nw = dev.Nw, UJ = 2 * dev.ms - 1, S = 2 * Nw + 1;
    double f = 0.0;
    for (int b = 0; b < A.nbands; b++) {
        const int rows = std::min(A.rows, (b + 1) * A.band_rows) - b * A.band_rows;
        if (rows <= 0) continue;
        const double steps = rows + 2 * Nw;
        f += (double)A.nstrips * UJ * UJ * 64.0 * (steps * (dev.Na + S) + (double)rows * S);
    }
    *fma = f;
}

// one row chunk of the table; `fma`: the fp64 FMAs the launch executes (roofline accounting)
inline hipError_t launch_march(const ModelDev& dev, MarchArgs A, const Sep1D& sep, const MarchPlan& P, hipStream_t s, double* fma)
{
    const int Nw = dev.Nw, UJ = 2 * dev.ms - 1, S = 2 * Nw + 1;
    // bands: one workgroup per CU; as many bands as fill whole rounds of the chip, each paying 2 Nw rows of lead-in
    const int ncu = device_cu_count();
    int best_nb = 1; double best_cost = 1e300;
    for (int nb = 1; nb <= 64 && nb <= A.rows; nb++) {
        const double rounds = std::ceil((double)A.nstrips * A.npass * nb / ncu);
        const double cost = rounds * (std::ceil((double)A.rows / nb) + 2 * Nw + 8);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_nb = nb; }
    }
    { const char* e = getenv("UMPA_HIP_MARCH_BANDS"); if (e && atoi(e) > 0) best_nb = atoi(e); }
    if (A.nbands <= 0) { A.nbands = best_nb; A.band_rows = (A.rows + best_nb - 1) / best_nb; }   // (on-demand passes: the bands are the caller's tiles)
    if (fma) march_fma(dev, A, fma);
    if (Nw == 6) return P.nxb == 4 ? launch_march_inst<6, 4>(dev, A, sep, P, s) : launch_march_inst<6, 8>(dev, A, sep, P, s);
    return P.nxb == 4 ? launch_march_inst<7, 4>(dev, A, sep, P, s) : launch_march_inst<7, 8>(dev, A, sep, P, s);
}

// The tiles of the maps a region needs: its pixels, seen from the image, and max_shift around them (a map is read at the
// pixel and at the pixel + shift).  Matches of a part of the image -- a ROI, one rectangle of a sample-stepping stack -- then
// pay for their part of the maps only.
struct PrepRect { int tx0, tx1, ty0, ty1; };
template <int NW>
inline PrepRect prep_rect(const Maps& M, const RegionArgs& A, int ms)
{
    const int T = PrepCfg<NW>::T;
    const int ntx = (M.W - 2 * NW + T - 1) / T, nty = (M.H - 2 * NW + T - 1) / T;
    auto lo = [&](int x) { const int q = (x - ms - NW) / T; return x - ms - NW < 0 ? 0 : q; };
    auto hi = [&](int x, int n) { const int q = (x + ms - NW) / T + 1; return q > n ? n : (q < 0 ? 0 : q); };
    PrepRect R;
    R.ty0 = std::min(lo(A.org0), nty); R.ty1 = std::max(R.ty0, hi(A.org0 + A.step0 * (A.N0 - 1), nty));
    R.tx0 = std::min(lo(A.org1), ntx); R.tx1 = std::max(R.tx0, hi(A.org1 + A.step1 * (A.N1 - 1), ntx));
    return R;
}

template <int KIND, int NW>
inline hipError_t launch_prep(const ModelDev& dev, const Maps& M, const Sep1D& sep, int sides, hipStream_t s, const PrepRect& PR)
{
    using C = PrepCfg<NW>;
    static bool attr_set[64] = {};                                    // the attribute is per device
    int devid = 0;
    (void)hipGetDevice(&devid);
    {
        std::lock_guard<std::mutex> lock(tiled_attr_mutex());
        if (!attr_set[devid & 63]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_maps_kernel<KIND, NW>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
            if (e != hipSuccess) return e;
            attr_set[devid & 63] = true;
        }
    }
    const int ntx = PR.tx1 - PR.tx0, nty = PR.ty1 - PR.ty0;
    const int total = ntx * nty, grid = ((total + 7) / 8) * 8;
    if (total > 0) hipLaunchKernelGGL((prep_maps_kernel<KIND, NW>), dim3(grid), dim3(C::NT), C::LDS, s, dev, M, sep, ntx, nty, sides, PR.tx0, PR.ty0);
    return hipGetLastError();
}


// ---- dispatch over the compile-time window half-width
#define UMPA_NW_SWITCH(nw, CALL)                \
    switch (nw) {                               \
    case 1: { constexpr int NWC = 1; CALL; } break; \
    case 2: { constexpr int NWC = 2; CALL; } break; \
    case 3: { constexpr int NWC = 3; CALL; } break; \
    case 4: { constexpr int NWC = 4; CALL; } break; \
    case 5: { constexpr int NWC = 5; CALL; } break; \
    case 6: { constexpr int NWC = 6; CALL; } break; \
    case 7: { constexpr int NWC = 7; CALL; } break; \
    case 8: { constexpr int NWC = 8; CALL; } break; \
    default: break;                             \
    }

// ------------------------------------------------------------------------------------------------
// on-demand passes (umpa_ondemand.h): scratch and the launch sequence shared by the plain and the masked path
// ------------------------------------------------------------------------------------------------
// Measured on one MI355X (round 3): where a pass is expensive and the replay cheap -- models with masks:
// dark-field C2 + mask 41.8 -> 25.2 ms, plain 11.5 -> 8.2 ms -- leaving out ~40 % of the passes pays; on the plain
// path the serial small launches (seed tiles, repair rounds: ~0.5 ms on C2, ~7 ms on C3) cost what the table kernel
// saves (C2 3.44 -> 3.46 ms, C3 48.0 -> 46.7 ms), so it is off there unless UMPA_HIP_ONDEMAND=1 asks for it.
inline bool od_enabled(int ntiles, int npass, bool masked)
{
    const char* e = getenv("UMPA_HIP_ONDEMAND");                      // 0: never, 1: whenever a tile has 2..64 passes (read per match: the tests switch it)
    if (npass > 64 || npass < 2) return false;
    if (e && atoi(e) == 0) return false;
    if (e && atoi(e) == 1) return ntiles >= 2;
    return masked && ntiles >= 6;                                     // a few tiles at least: one of them is a seed tile
}

// device scratch for `ntiles` tiles of `npass` passes and `npx` region pixels; fills the pointers of `od`
#define UMPA_OD_NCNT (8 * OD_STAGES)                                 // counters of a chunk: int[8] per stage
struct OdBuffers { int* tiles[2]; int* px[2]; int* counters; size_t zero_bytes; };

inline int od_reserve(TiledState& st, int ntiles, int npass, size_t npx, OdArgs& od, OdBuffers& B)
{
    const size_t nt = (size_t)ntiles;
    B.zero_bytes = nt * 8 * 2 + nt * 4 + UMPA_OD_NCNT * 4;            // done, visited, tile_flag, counters: cleared per chunk
    const size_t need = B.zero_bytes + 2 * nt * 4 + nt * npass * 4 + 2 * npx * 4 + 64;
    if (st.od_cap < need) {
        if (st.od_buf) (void)hipFree(st.od_buf);
        st.od_buf = nullptr; st.od_cap = 0;
        if (hipMalloc(&st.od_buf, need) != hipSuccess) return -3;
        st.od_cap = need;
    }
    if (!st.od_host && hipHostMalloc((void**)&st.od_host, UMPA_OD_SLOTS * UMPA_OD_NCNT * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        st.od_host = nullptr;
        return -3;
    }
    char* b = (char*)st.od_buf;
    od.done = (unsigned long long*)b;
    od.visited = od.done + nt;
    od.tile_flag = (int*)(od.visited + nt);
    B.counters = od.tile_flag + nt;
    B.tiles[0] = B.counters + UMPA_OD_NCNT; B.tiles[1] = B.tiles[0] + nt;
    od.items = B.tiles[1] + nt;
    B.px[0] = od.items + nt * npass; B.px[1] = B.px[0] + npx;
    od.cnt0 = B.counters;
    return 0;
}

// One row chunk with on-demand passes.  corr(od) launches the table kernel (od.mode 3: static grid over the seed tiles;
// 2: persistent grid over od.items), replay(od) the walk kernel in od.mode 1 / 2 / 3.  Both return a hipError_t.
template <class Corr, class Replay>
inline hipError_t od_run_chunk(OdArgs od, const OdBuffers& B, hipStream_t s, Corr corr, Replay replay)
{
    hipError_t e = hipMemsetAsync(od.done, 0, B.zero_bytes, s);
    if (e != hipSuccess) return e;
    const int ntiles = od.ntx * od.nty, lb = (ntiles + 255) / 256;
    od.r0 = std::min(1, od.nty - 1); od.c0 = std::min(1, od.ntx - 1);
    { const char* pe = getenv("UMPA_HIP_OD_PRED"); od.nearest = pe && pe[0] == '0' ? 0 : 1; }   // tuning: 0 = union over the seed tiles around
    auto stage = [&](int r) { return B.counters + 8 * r; };
    // 1. seed tiles: every pass (a compact grid over them); their pixels record what they read
    od.cnt_in = nullptr; od.cnt_out = stage(0);
    od.tile_in = nullptr; od.tile_out = B.tiles[0]; od.px_in = nullptr; od.px_out = B.px[0];
    od.mode = 3; if ((e = corr(od)) != hipSuccess) return e;
    od.mode = 1; if ((e = replay(od)) != hipSuccess) return e;
    // 2. + 3. the other tiles: what the seed tiles around them visited; a walk that needs more parks its pixel
    hipLaunchKernelGGL(od_list_kernel, dim3(lb), dim3(256), 0, s, od, 1);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    od.mode = 2; if ((e = corr(od)) != hipSuccess) return e;
    od.mode = 2; if ((e = replay(od)) != hipSuccess) return e;
    // 4. repair rounds: what the parked pixels asked for (the last round: everything the missed tiles lack), then
    // the parked pixels again from the start
    for (int r = 1; r <= OD_ROUNDS; r++) {
        od.cnt_in = stage(r - 1); od.cnt_out = stage(r);
        od.tile_in = B.tiles[(r - 1) & 1]; od.tile_out = B.tiles[r & 1];
        od.px_in = B.px[(r - 1) & 1]; od.px_out = B.px[r & 1];
        hipLaunchKernelGGL(od_list_kernel, dim3(lb), dim3(256), 0, s, od, r == OD_ROUNDS ? 3 : 2);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        od.mode = 2; if ((e = corr(od)) != hipSuccess) return e;
        od.mode = 3; if ((e = replay(od)) != hipSuccess) return e;
    }
    return hipSuccess;
}

// The same without seed tiles: every walk starts at shift (0, 0), so the prediction can be the walks themselves on a lattice of
// sample pixels.  For every table kernel; corr_march's unit of work is (strip, band, pass) -- a "tile" is a strip x band, a pass
// NUY row offsets x every column offset.
//   0. the passes around shift (0, 0) (od.central), every tile;
//   1. the pixels of a lattice (every `sub`-th in both directions) walk; one that needs a pass that is not there parks and asks
//      for it (od_park); OD_SAMPLE_ROUNDS rounds of  list -> table kernel over the list -> parked pixels again,  then what
//      the last round still asked for.  Their cost is 1 / sub^2 of a replay each; what they leave behind is `done`;
//      (these walks store nothing: the start shifts A.uv are read AND written by a walk that ends);
//   2. every pixel walks (the lattice's again: they are 1 / sub^2 of the pixels); OD_ROUNDS repair rounds as in od_run_chunk,
//      the last one computing every pass the tiles with parked pixels lack.
template <class Corr, class Replay>
inline hipError_t od_run_chunk_lattice(OdArgs od, const OdBuffers& B, hipStream_t s, Corr corr, Replay replay, int sub)
{
    hipError_t e = hipMemsetAsync(od.done, 0, B.zero_bytes, s);
    if (e != hipSuccess) return e;
    const int ntiles = od.ntx * od.nty, lb = (ntiles + 255) / 256;
    od.r0 = od.c0 = -1;                                               // no seed tiles
    od.nearest = 1;
    auto stage = [&](int r) { return B.counters + 8 * r; };
    int k = 0;                                                        // the stage whose counters were written last
    auto round = [&](int what, bool walk) {
        od.cnt_in = stage(k); od.cnt_out = stage(k + 1);
        od.tile_in = B.tiles[k & 1]; od.tile_out = B.tiles[(k + 1) & 1];
        od.px_in = B.px[k & 1]; od.px_out = B.px[(k + 1) & 1];
        hipLaunchKernelGGL(od_list_kernel, dim3(lb), dim3(256), 0, s, od, what);
        hipError_t re = hipGetLastError();
        if (re != hipSuccess) return re;
        od.mode = 2; if ((re = corr(od)) != hipSuccess) return re;
        if (walk) { od.mode = 3; if ((re = replay(od)) != hipSuccess) return re; }
        k++;
        return hipSuccess;
    };
    od.cnt_in = nullptr; od.cnt_out = stage(0);
    od.tile_in = nullptr; od.tile_out = B.tiles[0]; od.px_in = nullptr; od.px_out = B.px[0];
    hipLaunchKernelGGL(od_list_kernel, dim3(lb), dim3(256), 0, s, od, 4);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    od.mode = 2; if ((e = corr(od)) != hipSuccess) return e;
    if (sub > 1) {
        od.sub = sub; od.mode = 2; if ((e = replay(od)) != hipSuccess) return e;
        for (int r = 1; r <= OD_SAMPLE_ROUNDS; r++) if ((e = round(2, true)) != hipSuccess) return e;
        if ((e = round(2, false)) != hipSuccess) return e;            // (also clears the flags and requests of the last round's tiles)
        od.sub = 0;
        k++;                                                          // a fresh stage for the walks of every pixel
        od.cnt_in = nullptr; od.cnt_out = stage(k);
        od.tile_in = nullptr; od.tile_out = B.tiles[k & 1]; od.px_in = nullptr; od.px_out = B.px[k & 1];
    }
    od.mode = 2; if ((e = replay(od)) != hipSuccess) return e;
    for (int r = 1; r <= OD_ROUNDS; r++) if ((e = round(r == OD_ROUNDS ? 3 : 2, true)) != hipSuccess) return e;
    return hipSuccess;
}

// the passes that hold the row and column offsets -1 .. +1 and the one further that every walk's 4 x 4 gather reaches
// (Optim.cpp:41-130: rows / columns min - 1 .. min + 2 of the SHIFT; a pass is counted in offsets = sigma * shift)
inline unsigned long long od_central_passes(int ms, int sigma, int nrow, int nbatch, int ub, int UJ)
{
    unsigned long long mask = 0;
    const int lo = sigma > 0 ? -1 : -2, hi = sigma > 0 ? 2 : 1;
    for (int oi = lo; oi <= hi; oi++)
        for (int oj = lo; oj <= hi; oj++) {
            const int ri = oi + ms - 1, rj = oj + ms - 1;
            if (ri < 0 || ri >= UJ || rj < 0 || rj >= UJ) continue;
            mask |= 1ull << ((ri / nrow) * nbatch + rj / ub);
        }
    return mask;
}

// which prediction the on-demand stages of corr_volume / corr_masked use: UMPA_HIP_OD_PRED = lattice | seed (default) | 0 (seed,
// union over the seed tiles around); corr_march's is always the lattice
inline bool od_lattice_wanted()
{
    const char* e = getenv("UMPA_HIP_OD_PRED");
    return e && e[0] == 'l';                                          // (measured: the seed tiles predict corr_volume's and corr_masked's 32 x 32 tiles better)
}
inline int od_alone_wanted()                                         // 1: parked pixels ask for the missed pass alone (od_park)
{
    const char* e = getenv("UMPA_HIP_OD_ALONE");
    return e ? atoi(e) : 1;
}
inline int od_lattice_sub(int fallback)
{
    const char* e = getenv("UMPA_HIP_OD_SUB");
    return e ? atoi(e) : fallback;
}

inline size_t tiled_table_budget()
{
    const char* e = getenv("UMPA_HIP_TABLE_MB");
    long mb = e ? atol(e) : 16384;                                    // 288 GB of HBM: C3's 30 GB table in two chunks (4 GiB: 50.2 -> 49.3 ms)
    if (mb < 16) mb = 16;
    return (size_t)mb << 20;
}

// One match of a region on the tiled path.  Returns 0, -3 (allocation) or a positive hipError_t.
// `reuse_ref_maps`: the caller vouches that the reference frames are those of the previous call (it owns them).
// `piece_rows` > 0: row chunks of at most that many dense rows (a multiple of 32) even where the table budget would
// allow more (the host-array entry point downloads the rows of chunk c while chunk c+1 is being matched, the multi-GPU
// leg sends them to rank 0); `on_rows(xi_lo, xi_hi)` is called after the kernels of a chunk have been enqueued.

inline int tiled_match(TiledState& st, const ModelDev& dev, int kind, int H, int W, const FrameBox& box, const RegionArgs& A,
                       hipStream_t s, TiledTimers* tt, bool reuse_ref_maps,
                       int piece_rows = 0, const std::function<void(int, int)>& on_rows = nullptr,
                       const FrameDesc* host_frames = nullptr)      // a host copy of dev.frames (corr_march's staging offsets)
{
    const int K = dev.Na, Nw = dev.Nw, ms = dev.ms, UJ = 2 * ms - 1;
    const size_t plane = (size_t)H * W;
    const size_t KP = ((size_t)K + 1) / 2;                             // pair planes per stack (Maps)
    const size_t nmaps = kind == 1 ? 2 + 4 * KP : 2;
    if (st.maps_cap < nmaps * plane) {
        if (st.maps) (void)hipFree(st.maps);
        st.maps = nullptr; st.maps_cap = 0;
        st.ref_maps_ok = false;
        if (hipMalloc((void**)&st.maps, nmaps * plane * sizeof(double)) != hipSuccess) return -3;
        st.maps_cap = nmaps * plane;
    }
    if (st.ref_kind != kind || st.ref_K != K || st.ref_plane != plane) st.ref_maps_ok = false;
    Maps M;
    M.H = H; M.W = W;
    M.br0 = box.r0; M.br1 = box.r1; M.bc0 = box.c0; M.bc1 = box.c1; M.Wf = box.Wf;
    M.SamSq = st.maps; M.RefSq = st.maps + plane;
    M.WS = kind == 1 ? st.maps + 2 * plane : nullptr;
    M.MR = kind == 1 ? st.maps + (2 + 2 * KP) * plane : nullptr;

    // The table lives on the dense (unit-step) grid under the region: with step > 1 every step-th entry is used
    // (tiled_applicable only sends small steps here).  Rows per chunk: the shift table of one chunk stays within
    // the budget (whole tiles).
    const int N0d = A.step0 * (A.N0 - 1) + 1, N1d = A.step1 * (A.N1 - 1) + 1;
    // table rows are padded to whole 256-byte tile rows: every (tile, row) of a plane is then a run of whole, aligned
    // 128-byte lines (C2: 2028 -> 2048 doubles; UMPA_HIP_TABLE_ALIGN=1 packs the rows as round 3 did)
    static const int table_align = getenv("UMPA_HIP_TABLE_ALIGN") ? std::max(1, atoi(getenv("UMPA_HIP_TABLE_ALIGN"))) : 32;
    int N1p = (N1d + table_align - 1) / table_align * table_align;
    // wide windows: the table comes from corr_march (strip-blocked: [strip][rows][shift][tw], one "row" = nstrips * UJ^2 * tw doubles)
    MarchPlan MP;
    memset(&MP, 0, sizeof(MP));
    const char* march_baseA = nullptr; const char* march_baseB = nullptr;
    {
        if (host_frames && march_wanted(Nw)) {
            MP = march_plan(Nw, UJ, K, N1d);
            if (MP.ok && !march_offsets(host_frames, K, dev.ref_mode ? -1 : 1, box, st.march_off_host, march_baseA, march_baseB)) MP.ok = false;
        }
    }
    if (MP.ok) N1p = MP.nstrips * MP.tw;                               // doubles per dense row and shift
    // on-demand passes of corr_march (od_run_chunk_march): UMPA_HIP_ONDEMAND=1 on, =0 off
    bool march_od = false;
    { const char* od_env = getenv("UMPA_HIP_ONDEMAND"); if (od_env) march_od = atoi(od_env) != 0; }
    march_od = march_od && MP.ok && MP.npass >= 3 && MP.npass <= 64 && !getenv("UMPA_HIP_ABLATE_MARCH");
    const size_t row_bytes = (size_t)UJ * UJ * N1p * sizeof(double);
    long rows_chunk = (long)(tiled_table_budget() / row_bytes) / UMPA_TILE * UMPA_TILE;
    if (rows_chunk < UMPA_TILE) rows_chunk = UMPA_TILE;
    if (rows_chunk > N0d) rows_chunk = ((long)N0d + UMPA_TILE - 1) / UMPA_TILE * UMPA_TILE;
    if (piece_rows > 0) {
        const long want = ((long)piece_rows + UMPA_TILE - 1) / UMPA_TILE * UMPA_TILE;
        if (want < rows_chunk) rows_chunk = want;
    }
    {   // eval_lookup multiplies the slot number by a 32-bit slot stride (rows_chunk * N1d)
        const long cap = (long)(0xffffffffull / (size_t)N1p) / UMPA_TILE * UMPA_TILE;
        if (cap < UMPA_TILE) return (int)hipErrorInvalidValue;
        if (rows_chunk > cap) rows_chunk = cap;
    }
    if (st.table_limited && st.table_cap > 0) {                        // an earlier allocation of the full budget failed: live with what we got
        const long fit = (long)(st.table_cap / ((size_t)UJ * UJ * N1p)) / UMPA_TILE * UMPA_TILE;
        if (fit >= UMPA_TILE && fit < rows_chunk) rows_chunk = fit;
    }
    size_t table_need = (size_t)UJ * UJ * rows_chunk * N1p;
    if (st.table_cap < table_need) {
        if (st.table) (void)hipFree(st.table);
        st.table = nullptr; st.table_cap = 0;
        // a card with less free memory than the budget assumes: smaller row chunks rather than no tiled path
        while (hipMalloc((void**)&st.table, table_need * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            st.table = nullptr;
            if (rows_chunk <= UMPA_TILE) return -3;
            st.table_limited = true;
            rows_chunk = std::max<long>(UMPA_TILE, rows_chunk / 2 / UMPA_TILE * UMPA_TILE);
            table_need = (size_t)UJ * UJ * rows_chunk * N1p;
        }
        st.table_cap = table_need;
    }

    bool timing_open = false;
    auto tic = [&](int name) {
        timing_open = false;
        if (!tt) return;
        TiledTimers::Entry en = {name, tt->get(), tt->get(), 0.0, nullptr, 0.0};
        if (!en.t0 || !en.t1) return;
        (void)hipEventRecord(en.t0, s);
        tt->entries.push_back(en);
        timing_open = true;
    };
    auto toc = [&](double fma = 0.0) {
        if (!timing_open) return;
        tt->entries.back().fma = fma;
        (void)hipEventRecord(tt->entries.back().t1, s);
        timing_open = false;
    };

    hipError_t e = hipErrorInvalidValue;
    tic(2);
    PrepRect PRc = {0, 0, 0, 0};
    UMPA_NW_SWITCH(Nw, (PRc = prep_rect<NWC>(M, A, ms)))
    const bool covered = st.ref_rect[0] <= PRc.tx0 && st.ref_rect[1] >= PRc.tx1 && st.ref_rect[2] <= PRc.ty0 && st.ref_rect[3] >= PRc.ty1;
    const int sides = (reuse_ref_maps && st.ref_maps_ok && covered) ? 1 : 3;
    st.ref_maps_ok = false;
    if (kind == 1) { UMPA_NW_SWITCH(Nw, (e = launch_prep<1, NWC>(dev, M, st.sep, sides, s, PRc))) }
    else { UMPA_NW_SWITCH(Nw, (e = launch_prep<0, NWC>(dev, M, st.sep, sides, s, PRc))) }
    toc();
    if (e != hipSuccess) return (int)e;
    st.ref_maps_ok = true; st.ref_kind = kind; st.ref_K = K; st.ref_plane = plane;
    st.ref_rect[0] = PRc.tx0; st.ref_rect[1] = PRc.tx1; st.ref_rect[2] = PRc.ty0; st.ref_rect[3] = PRc.ty1;

    int ub = pick_ub(UJ);
    { const char* e = getenv("UMPA_HIP_UB"); if (e) ub = atoi(e); }      // tuning override: 9, 8, 7 or 5
    st.stat_n = 0; st.stat_total_passes = 0.0;
    if (MP.ok) {                                                       // the frames' staging offsets (stable storage: the state's own vector)
        if (st.march_off_cap < 2 * K) {
            if (st.march_off) (void)hipFree(st.march_off);
            st.march_off = nullptr; st.march_off_cap = 0;
            if (hipMalloc((void**)&st.march_off, 2 * (size_t)K * sizeof(unsigned)) != hipSuccess) return -3;
            st.march_off_cap = 2 * K;
        }
        if ((e = hipMemcpyAsync(st.march_off, st.march_off_host.data(), 2 * (size_t)K * sizeof(unsigned), hipMemcpyHostToDevice, s)) != hipSuccess) return (int)e;
    }
    for (int drow0 = 0; drow0 < N0d; drow0 += (int)rows_chunk) {
        const int drows = (int)((N0d - drow0 < rows_chunk) ? N0d - drow0 : rows_chunk);
        CorrArgs CA;
        CA.table = st.table; CA.slot_stride = (size_t)drows * N1p; CA.pitch = N1p;
        CA.org0 = A.org0; CA.org1 = A.org1; CA.row0 = drow0; CA.rows = drows; CA.N1 = N1d;
        CA.sigma = dev.ref_mode ? -1 : 1;
        CA.br0 = box.r0; CA.br1 = box.r1; CA.bc0 = box.c0; CA.bc1 = box.c1 + box.slack; CA.Wf = box.Wf;   // (pairs may start on column c1 then)
        { const char* ab = getenv("UMPA_HIP_ABLATE"); CA.ablate = ab ? atoi(ab) : 0; }
        CA.ntx = CA.nty = 0;                                          // set by launch_corr for the tile shape it picks
        // output rows whose dense row lies in [drow0, drow0 + drows)
        const int xi_lo = (drow0 + A.step0 - 1) / A.step0;
        const int xi_hi = std::min(A.N0, (drow0 + drows - 1) / A.step0 + 1);
        ReplayArgs R;
        R.table = st.table; R.slot_stride = CA.slot_stride; R.drow0 = drow0; R.N1d = N1p;
        R.strip_w = 0; R.tw = 0; R.drows = drows;
        if (MP.ok) { R.strip_w = MP.wo; R.tw = MP.tw; R.slot_stride = (size_t)MP.tw; }
        {
            const char* be = getenv("UMPA_HIP_REPLAY_BW");           // tuning: 64, 32, 16, 8, 4
            const int bw = be ? atoi(be) : (MP.ok ? 32 : 16);
            R.bw_log2 = bw >= 64 ? 6 : bw >= 32 ? 5 : bw >= 16 ? 4 : bw >= 8 ? 3 : 2;
        }
        R.row0 = xi_lo; R.rows = std::max(0, xi_hi - xi_lo);
        { const char* ab = getenv("UMPA_HIP_ABLATE_REPLAY"); R.ablate = ab ? atoi(ab) : 0; }
        // frame count as a template constant where the map planes are 32-bit addressable (eval_lookup)
        const bool small = (size_t)M.H * M.W * 2 * sizeof(double) < ((size_t)1 << 32);   // a pair plane

        CorrLaunch CL;
        memset(&CL, 0, sizeof(CL));
        CL.dry = true;                                                // which shape, how many passes?
        e = hipErrorInvalidValue;
        UMPA_NW_SWITCH(Nw, (e = launch_corr_nw<NWC>(ub, dev, CA, st.sep, s, CL)))
        if (e != hipSuccess) return (int)e;
        CL.dry = false;
        const int ntiles = CL.ntx * CL.nty;
        const int* counts = nullptr;
        double march_fma_full = 0.0;

        auto corr = [&](const OdArgs& od) {
            CL.od = od;
            hipError_t ce = hipErrorInvalidValue;
            if (MP.ok) {                                              // corr_march (od.mode is 0 here: the on-demand stages are corr_volume's)
                MarchArgs MA;
                memset(&MA, 0, sizeof(MA));
                MA.table = st.table; MA.tw = MP.tw;
                MA.org0 = A.org0; MA.org1 = A.org1; MA.row0 = drow0; MA.rows = drows; MA.N1 = N1d;
                MA.sigma = CA.sigma;
                MA.br0 = box.r0; MA.br1 = box.r1; MA.bc0 = box.c0; MA.bc1 = box.c1 + box.slack; MA.Wf = box.Wf;
                MA.nstrips = MP.nstrips; MA.npass = MP.npass; MA.nuy = MP.nuy;
                MA.npa = MP.npa; MA.npb = MP.npb; MA.a_slot = MP.a_slot; MA.b_slot = MP.b_slot; MA.da = MP.da; MA.db = MP.db;
                MA.baseA = march_baseA; MA.baseB = march_baseB; MA.frame_off = st.march_off;
                { const char* ab = getenv("UMPA_HIP_ABLATE_MARCH"); MA.ablate = ab ? atoi(ab) : 0; }
                if (od.done) {                                        // on-demand passes: the caller's bands, a part of the units
                    MA.nbands = od.nty; MA.band_rows = od.tr;
                    MA.done = od.done; MA.ndone = od.cnt0 ? od.cnt0 + OD_C_DONE : nullptr;
                    if (od.mode == 2) { MA.items = od.items; MA.nitems = od.cnt_out + OD_C_ITEMS; }
                }
                double fma = 0.0;
                tic(9);
                ce = launch_march(dev, MA, st.sep, MP, s, &fma);
                if (!od.done) march_fma_full = fma;
                toc(od.done ? 0.0 : fma);                             // (on-demand: counted from the device's tally, below)
                return ce;
            }
            tic(3);
            UMPA_NW_SWITCH(Nw, (ce = launch_corr_nw<NWC>(ub, dev, CA, st.sep, s, CL)))
            toc(od.mode ? 0.0 : CL.fma_per_pass * ntiles * CL.npass);
            return ce;
        };
        auto replay = [&](const OdArgs& od) {
            if (R.rows <= 0) return hipSuccess;
            dim3 blk(64, UMPA_REPLAY_ROWS), grd((A.N1 + 63) / 64, (R.rows + UMPA_REPLAY_ROWS - 1) / UMPA_REPLAY_ROWS);
            if (od.mode == 0 || od.mode == 2) {                       // (every pixel: blocks of 2^bw_log2 x 2^(6 - bw_log2) pixels per wave)
                const int bw = 1 << R.bw_log2, bh = UMPA_REPLAY_ROWS * (64 >> R.bw_log2);
                grd = dim3((A.N1 + bw - 1) / bw, (R.rows + bh - 1) / bh);
                // Workgroups go to the XCDs in turn: with a multiple of 8 blocks per grid row a block and the ones above and below it,
                // which read much the same rows of the maps, meet in ONE XCD's L2.  Whether that is good depends on how much they share
                // (profiles/r04_replay_blocks.txt): at C2 (28 KB of map window per block) eight L2s serve them faster than one -- 0.967
                // against 0.916 ms --, at C3 (130 KB per block: 20 frames, +-7 shifts) the lines fetched eight times over are what
                // counts -- 7.8 against 13.2 ms.  So: a multiple of 8 for large windows, never one for small ones (idle blocks on the right).
                {
                    const size_t window = (size_t)(bh + 2 * ms) * (bw + 2 * ms) * (((size_t)K + 1) / 2 * 16 + 16);
                    const char* ge = getenv("UMPA_HIP_REPLAY_GRIDX");                 // tuning: 1 never a multiple of 8, 2 always
                    const int gx = ge ? atoi(ge) : (kind == 1 && window > 65536 ? 2 : 1);
                    if (gx == 1 && (grd.x & 7) == 0) grd.x += 1;
                    if (gx == 2) grd.x = (grd.x + 7) / 8 * 8;
                }
            }
            if (od.mode == 3) grd = dim3(2 * device_cu_count(), 1);   // queue over the parked pixels
            if (od.mode == 2 && od.sub > 1)                           // corr_march's sample lattice
                grd = dim3(((A.N1 + od.sub - 1) / od.sub + 63) / 64, ((R.rows + od.sub - 1) / od.sub + UMPA_REPLAY_ROWS - 1) / UMPA_REPLAY_ROWS);
            if (od.mode == 1) { blk = dim3(64, 1); grd = dim3((32 * od.tc + 63) / 64, od_seed_count(od.ntx, od.c0) * od_seed_count(od.nty, od.r0)); if (!grd.y) return hipSuccess; }
            static const int pad_lds = getenv("UMPA_HIP_REPLAY_PAD_LDS") ? atoi(getenv("UMPA_HIP_REPLAY_PAD_LDS")) : 0;   // diagnostics: occupancy
            tic(4);
#define UMPA_REPLAY_NA(n) case n: if (od.mode) hipLaunchKernelGGL((replay_walk_kernel<1, n, true>), grd, blk, pad_lds, s, dev, M, R, A, od); \
                           else hipLaunchKernelGGL((replay_walk_kernel<1, n, false>), grd, blk, pad_lds, s, dev, M, R, A, od); break;
            if (kind == 1 && small && dev.Na <= UMPA_KTEMPL) {
                switch (dev.Na) {
                    UMPA_REPLAY_NA(1) UMPA_REPLAY_NA(2) UMPA_REPLAY_NA(3) UMPA_REPLAY_NA(4) UMPA_REPLAY_NA(5) UMPA_REPLAY_NA(6)
                    UMPA_REPLAY_NA(7) UMPA_REPLAY_NA(8) UMPA_REPLAY_NA(9) UMPA_REPLAY_NA(10) UMPA_REPLAY_NA(11) UMPA_REPLAY_NA(12)
                    UMPA_REPLAY_NA(13) UMPA_REPLAY_NA(14) UMPA_REPLAY_NA(15) UMPA_REPLAY_NA(16) UMPA_REPLAY_NA(17) UMPA_REPLAY_NA(18)
                    UMPA_REPLAY_NA(19) UMPA_REPLAY_NA(20) UMPA_REPLAY_NA(21) UMPA_REPLAY_NA(22) UMPA_REPLAY_NA(23) UMPA_REPLAY_NA(24)
                }
            } else if (kind == 1) {
                if (od.mode) hipLaunchKernelGGL((replay_walk_kernel<1, 0, true>), grd, blk, pad_lds, s, dev, M, R, A, od);
                else hipLaunchKernelGGL((replay_walk_kernel<1, 0, false>), grd, blk, pad_lds, s, dev, M, R, A, od);
            } else if (od.mode) hipLaunchKernelGGL((replay_walk_kernel<0, 0, true>), grd, blk, pad_lds, s, dev, M, R, A, od);
            else hipLaunchKernelGGL((replay_walk_kernel<0, 0, false>), grd, blk, pad_lds, s, dev, M, R, A, od);
#undef UMPA_REPLAY_NA
            toc();
            return hipGetLastError();
        };

        OdArgs od;
        memset(&od, 0, sizeof(od));
        od.tc = CL.tc; od.ub = CL.ub; od.nbatch = CL.nbatch; od.npass = CL.npass; od.ntx = CL.ntx; od.nty = CL.nty;
        od.ub_inv = (65536 + CL.ub - 1) / CL.ub; od.nrow_inv = (65536 + CL.nrow - 1) / CL.nrow;
        od.tr = UMPA_TILE;
        OdBuffers OB;
        int march_tiles = 0;
        if (march_od) {                                               // tiles = strips x bands of about UMPA_HIP_MARCH_OD_ROWS rows
            const char* bt = getenv("UMPA_HIP_MARCH_OD_ROWS");      // (read per match: the tests set it)
            const int band_target = bt ? std::max(32, atoi(bt)) : 512;
            const int nb = std::max(1, (drows + band_target / 2) / band_target);
            od.tc = MP.wo; od.tr = (drows + nb - 1) / nb; od.ntx = MP.nstrips; od.nty = nb;
            od.npass = MP.npass; od.nbatch = 1; od.ub = 64; od.ub_inv = 1024; od.nrow_inv = (65536 + MP.nuy - 1) / MP.nuy;
            od.alone = 1;
            od.central = od_central_passes(ms, dev.ref_mode ? -1 : 1, MP.nuy, 1, 64, UJ);
            march_tiles = od.ntx * od.nty;
        }
        if (march_od && march_tiles >= 2) {
            if (od_reserve(st, march_tiles, MP.npass, (size_t)A.N0 * A.N1, od, OB)) return -3;
            if ((e = od_run_chunk_lattice(od, OB, s, corr, replay, od_lattice_sub(8))) != hipSuccess) return (int)e;
            int* slot = st.od_host + UMPA_OD_NCNT * (st.od_slot++ % UMPA_OD_SLOTS);
            if ((e = hipMemcpyAsync(slot, OB.counters, UMPA_OD_NCNT * sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return (int)e;
            counts = slot;
            if (st.stat_n < 64) st.stat_slots[st.stat_n++] = slot;
            if (getenv("UMPA_HIP_OD_DEBUG")) {                        // diagnostics: the stages' counters (a host wait)
                (void)hipStreamSynchronize(s);
                fprintf(stderr, "march on-demand: %d tiles (%d strips x %d bands of %d rows) x %d passes\n", march_tiles, od.ntx, od.nty, od.tr, MP.npass);
                for (int r = 0; r < OD_STAGES; r++)
                    fprintf(stderr, "  stage %2d: tiles listed %6d  pixels parked %8d  units listed %6d  units done (stage 0: all) %6d\n",
                            r, slot[8 * r + OD_C_TILES], slot[8 * r + OD_C_PX], slot[8 * r + OD_C_ITEMS], slot[8 * r + OD_C_DONE]);
            }
            {   // the FMAs of one unit: an exhaustive launch's over its units (the bands are equal but for the last)
                MarchArgs MF;
                memset(&MF, 0, sizeof(MF));
                MF.rows = drows; MF.nstrips = MP.nstrips; MF.npass = MP.npass; MF.nbands = od.nty; MF.band_rows = od.tr;
                double f = 0.0;
                march_fma(dev, MF, &f);
                march_fma_full = f / ((double)march_tiles * MP.npass);
            }
            if (tt) for (auto it = tt->entries.rbegin(); it != tt->entries.rend(); ++it)
                if (it->name == 9) { it->counts = counts; it->fma_per = march_fma_full; break; }
            st.stat_total_passes += (double)march_tiles * MP.npass - (double)ntiles * CL.npass;   // (the sum below adds ntiles * CL.npass)
        } else if (!MP.ok && od_enabled(ntiles, CL.npass, false) && !CA.ablate) {
            if (od_reserve(st, ntiles, CL.npass, (size_t)A.N0 * A.N1, od, OB)) return -3;
            if (od_lattice_wanted()) {
                od.alone = od_alone_wanted();
                od.central = od_central_passes(ms, dev.ref_mode ? -1 : 1, CL.nrow, CL.nbatch, CL.ub, UJ);
                e = od_run_chunk_lattice(od, OB, s, corr, replay, od_lattice_sub(4));
            } else e = od_run_chunk(od, OB, s, corr, replay);
            if (e != hipSuccess) return (int)e;
            // the counters of this chunk, for the FMA count of a timed match and for umpa_hip_last_stats
            int* slot = st.od_host + UMPA_OD_NCNT * (st.od_slot++ % UMPA_OD_SLOTS);
            if ((e = hipMemcpyAsync(slot, OB.counters, UMPA_OD_NCNT * sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return (int)e;
            counts = slot;
            if (st.stat_n < 64) st.stat_slots[st.stat_n++] = slot;
            if (tt) for (auto it = tt->entries.rbegin(); it != tt->entries.rend(); ++it)
                if (it->name == 3) { it->counts = counts; it->fma_per = CL.fma_per_pass; break; }   // all FMAs of the chunk on its last table launch
        } else {
            od.mode = 0;
            if ((e = corr(od)) != hipSuccess) return (int)e;
            if ((e = replay(od)) != hipSuccess) return (int)e;
        }
        st.stat_total_passes += (double)ntiles * CL.npass;
        if (xi_hi > xi_lo && on_rows) on_rows(xi_lo, xi_hi);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// models with masks: corr_masked + replay_cost (umpa_masked.h)
// ------------------------------------------------------------------------------------------------
// column offsets per pass: as many as the registers hold accumulated planes for (DF also keeps the filter threads' sums)
template <int KIND, int NW>
constexpr int masked_ub() { return KIND == 1 ? (NW <= 6 ? 3 : 2) : (NW <= 6 ? 5 : 3); }

// Tile width of the masked table kernel.  MaskCfg also describes 32 x 16 tiles with two 256-thread workgroups per CU
// (TC_ = 16: a single buffer for the means, 6 columns per product thread); measured on C2 + mask, dark-field: 24.1 against
// 21.5 ms for one 512-thread workgroup on 32 x 32 tiles -- 21 % more halo in the products, a quarter more staging per pixel
// and 32 spilled registers outweigh what two independent workgroups overlap.  Not instantiated.
template <int KIND, int NW>
constexpr int masked_tc_narrow() { return 32; }

template <int NW>
constexpr bool masked_cfg_ok() { return MaskCfg<0, NW, masked_ub<0, NW>()>::OK && MaskCfg<1, NW, masked_ub<1, NW>()>::OK; }

inline bool masked_supported(int Nw)
{
    bool ok = false;
    UMPA_NW_SWITCH(Nw, (ok = masked_cfg_ok<NWC>()))
    return ok;
}

template <int KIND, int NW, int TCT>
inline hipError_t launch_masked_tc(const ModelDev& dev, MaskedArgs A, const Sep1D& sep, hipStream_t s, CorrLaunch& L)
{
    constexpr int UBM = masked_ub<KIND, NW>();
    using C = MaskCfg<KIND, NW, UBM, TCT>;
    if constexpr (!C::OK) return hipErrorInvalidValue;
    else {
        A.ntx = (A.N1 + C::TC - 1) / C::TC;
        A.nty = (A.rows + C::TR - 1) / C::TR;
        const int UJ = 2 * dev.ms - 1, nbatch = (UJ + UBM - 1) / UBM, npass = UJ * nbatch;
        L.tc = C::TC; L.ub = UBM; L.nrow = 1; L.nbatch = nbatch; L.npass = npass; L.ntx = A.ntx; L.nty = A.nty;
        // fp64 issue slots (FMA, multiply or add each counted once) of one (tile, pass) with UBM column offsets: per frame
        // and column offset the products of the active threads (pair weight 6 + 7) and, DF, three planes through both
        // filters + the fold into t2, t4, t6, wt; at the end NPL planes through both filters and the solve
        const double filt = (double)C::QR * C::TC * C::S + (double)C::TR * C::TC * C::S;
        const double prod = (double)C::QR * C::NQB * C::QB * 13.0;
        const double per_shift_frame = prod + (KIND == 1 ? 3.0 * filt + 2.0 * C::TR * C::TC * 4.0 : 0.0);
        const double per_shift_end = C::NPL * filt + 20.0 * C::TR * C::TC;
        L.fma_per_pass = (per_shift_frame * dev.Na + per_shift_end) * (double)UJ / nbatch;    // (UJ real column offsets over nbatch passes)
        if (L.dry) return hipSuccess;
        static bool attr_set[64] = {};
        int devid = 0;
        (void)hipGetDevice(&devid);
        {
            std::lock_guard<std::mutex> lock(tiled_attr_mutex());
            if (!attr_set[devid & 63]) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_masked_kernel<KIND, NW, UBM, TCT>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_masked_queue_kernel<KIND, NW, UBM, TCT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
                if (e != hipSuccess) return e;
                attr_set[devid & 63] = true;
            }
        }
        const OdCorr oc = od_corr_args(L.od);
        if (L.od.mode == 2) {
            const int grid = ((device_cu_count() * C::WPC + 7) / 8) * 8;       // persistent: one workgroup per slot of the chip
            hipLaunchKernelGGL((corr_masked_queue_kernel<KIND, NW, UBM, TCT>), dim3(grid), dim3(C::NT), C::LDS, s, dev, A, sep, oc);
        } else {
            const int nt = L.od.mode == 3 ? oc.nseed : A.ntx * A.nty;
            const int grid = 8 * ((nt + 7) / 8) * npass;
            if (nt > 0) hipLaunchKernelGGL((corr_masked_kernel<KIND, NW, UBM, TCT>), dim3(grid), dim3(C::NT), C::LDS, s, dev, A, sep, oc);
        }
        return hipGetLastError();
    }
}

template <int KIND, int NW>
inline hipError_t launch_masked(const ModelDev& dev, const MaskedArgs& A, const Sep1D& sep, hipStream_t s, CorrLaunch& L)
{
    constexpr int TCN = masked_tc_narrow<KIND, NW>();
    return launch_masked_tc<KIND, NW, TCN>(dev, A, sep, s, L);
}

// One match of a region of a masked model.  Returns 0, -3 (allocation) or a positive hipError_t.
inline int tiled_match_masked(TiledState& st, const ModelDev& dev, int kind, int H, int W, const FrameBox& box, const RegionArgs& A,
                              hipStream_t s, TiledTimers* tt, bool reuse_ref_maps, bool binary_masks,
                              int piece_rows = 0, const std::function<void(int, int)>& on_rows = nullptr)
{
    const int K = dev.Na, Nw = dev.Nw, ms = dev.ms, UJ = 2 * ms - 1, NV = kind == 1 ? 3 : 2;
    const size_t plane = (size_t)H * W;
    const size_t KP = ((size_t)K + 1) / 2;
    Maps M;
    M.H = H; M.W = W;
    M.br0 = box.r0; M.br1 = box.r1; M.bc0 = box.c0; M.bc1 = box.c1; M.Wf = box.Wf;
    M.SamSq = M.RefSq = M.WS = M.MR = nullptr;
    if (kind == 1) {                                                   // the un-weighted reference means (Model.cpp:804-808): prep_maps' MR planes
        const size_t nmaps = 2 + 4 * KP;
        if (st.maps_cap < nmaps * plane) {
            if (st.maps) (void)hipFree(st.maps);
            st.maps = nullptr; st.maps_cap = 0;
            st.ref_maps_ok = false;
            if (hipMalloc((void**)&st.maps, nmaps * plane * sizeof(double)) != hipSuccess) return -3;
            st.maps_cap = nmaps * plane;
        }
        if (st.ref_kind != kind || st.ref_K != K || st.ref_plane != plane) st.ref_maps_ok = false;
        M.SamSq = st.maps; M.RefSq = st.maps + plane;
        M.WS = st.maps + 2 * plane;
        M.MR = st.maps + (2 + 2 * KP) * plane;
    }
    const int N0d = A.step0 * (A.N0 - 1) + 1, N1d = A.step1 * (A.N1 - 1) + 1;
    const size_t row_bytes = (size_t)NV * UJ * UJ * N1d * sizeof(double);
    long rows_chunk = (long)(tiled_table_budget() / row_bytes) / UMPA_TILE * UMPA_TILE;
    if (rows_chunk < UMPA_TILE) rows_chunk = UMPA_TILE;
    if (rows_chunk > N0d) rows_chunk = ((long)N0d + UMPA_TILE - 1) / UMPA_TILE * UMPA_TILE;
    if (piece_rows > 0) {
        const long want = ((long)piece_rows + UMPA_TILE - 1) / UMPA_TILE * UMPA_TILE;
        if (want < rows_chunk) rows_chunk = want;
    }
    if (st.table_limited && st.table_cap > 0) {
        const long fit = (long)(st.table_cap / ((size_t)NV * UJ * UJ * N1d)) / UMPA_TILE * UMPA_TILE;
        if (fit >= UMPA_TILE && fit < rows_chunk) rows_chunk = fit;
    }
    size_t table_need = (size_t)NV * UJ * UJ * rows_chunk * N1d;
    if (st.table_cap < table_need) {
        if (st.table) (void)hipFree(st.table);
        st.table = nullptr; st.table_cap = 0;
        while (hipMalloc((void**)&st.table, table_need * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            st.table = nullptr;
            if (rows_chunk <= UMPA_TILE) return -3;
            st.table_limited = true;
            rows_chunk = std::max<long>(UMPA_TILE, rows_chunk / 2 / UMPA_TILE * UMPA_TILE);
            table_need = (size_t)NV * UJ * UJ * rows_chunk * N1d;
        }
        st.table_cap = table_need;
    }

    bool timing_open = false;
    auto tic = [&](int name) {
        timing_open = false;
        if (!tt) return;
        TiledTimers::Entry en = {name, tt->get(), tt->get(), 0.0, nullptr, 0.0};
        if (!en.t0 || !en.t1) return;
        (void)hipEventRecord(en.t0, s);
        tt->entries.push_back(en);
        timing_open = true;
    };
    auto toc = [&](double fma = 0.0) {
        if (!timing_open) return;
        tt->entries.back().fma = fma;
        (void)hipEventRecord(tt->entries.back().t1, s);
        timing_open = false;
    };

    hipError_t e = hipErrorInvalidValue;
    if (kind == 1) {
        PrepRect PRc = {0, 0, 0, 0};
        UMPA_NW_SWITCH(Nw, (PRc = prep_rect<NWC>(M, A, ms)))
        const bool covered = st.ref_rect[0] <= PRc.tx0 && st.ref_rect[1] >= PRc.tx1 && st.ref_rect[2] <= PRc.ty0 && st.ref_rect[3] >= PRc.ty1;
        const int sides = (reuse_ref_maps && st.ref_maps_ok && covered) ? 0 : 3;   // only the reference means are read: nothing to redo for a new sample stack
        st.ref_maps_ok = false;
        if (sides) {
            tic(2);
            UMPA_NW_SWITCH(Nw, (e = launch_prep<1, NWC>(dev, M, st.sep, sides, s, PRc)))
            toc();
            if (e != hipSuccess) return (int)e;
            st.ref_rect[0] = PRc.tx0; st.ref_rect[1] = PRc.tx1; st.ref_rect[2] = PRc.ty0; st.ref_rect[3] = PRc.ty1;
        }
        st.ref_maps_ok = true; st.ref_kind = kind; st.ref_K = K; st.ref_plane = plane;
    }
    st.stat_n = 0; st.stat_total_passes = 0.0;
    for (int drow0 = 0; drow0 < N0d; drow0 += (int)rows_chunk) {
        const int drows = (int)((N0d - drow0 < rows_chunk) ? N0d - drow0 : rows_chunk);
        MaskedArgs MA;
        MA.table = st.table; MA.slot_stride = (size_t)drows * N1d;
        MA.MR = M.MR; MA.H = H; MA.W = W;
        MA.org0 = A.org0; MA.org1 = A.org1; MA.row0 = drow0; MA.rows = drows; MA.N1 = N1d;
        MA.sigma = dev.ref_mode ? -1 : 1;
        MA.br0 = box.r0; MA.br1 = box.r1; MA.bc0 = box.c0; MA.bc1 = box.c1; MA.Wf = box.Wf;
        MA.ntx = MA.nty = 0;
        MA.binary = binary_masks ? 1 : 0;
        { const char* ab = getenv("UMPA_HIP_ABLATE_MASKED"); MA.ablate = ab ? atoi(ab) : 0; }
        const int xi_lo = (drow0 + A.step0 - 1) / A.step0;
        const int xi_hi = std::min(A.N0, (drow0 + drows - 1) / A.step0 + 1);
        const int rrows = std::max(0, xi_hi - xi_lo);

        CorrLaunch CL;
        memset(&CL, 0, sizeof(CL));
        CL.dry = true;
        e = hipErrorInvalidValue;
        if (kind == 1) { UMPA_NW_SWITCH(Nw, (e = launch_masked<1, NWC>(dev, MA, st.sep, s, CL))) }
        else { UMPA_NW_SWITCH(Nw, (e = launch_masked<0, NWC>(dev, MA, st.sep, s, CL))) }
        if (e != hipSuccess) return (int)e;
        CL.dry = false;
        const int ntiles = CL.ntx * CL.nty;

        auto corr = [&](const OdArgs& od) {
            CL.od = od;
            hipError_t ce = hipErrorInvalidValue;
            tic(6);
            if (kind == 1) { UMPA_NW_SWITCH(Nw, (ce = launch_masked<1, NWC>(dev, MA, st.sep, s, CL))) }
            else { UMPA_NW_SWITCH(Nw, (ce = launch_masked<0, NWC>(dev, MA, st.sep, s, CL))) }
            toc(od.mode ? 0.0 : CL.fma_per_pass * ntiles * CL.npass);
            return ce;
        };
        auto replay = [&](const OdArgs& od) {
            if (rrows <= 0) return hipSuccess;
            dim3 blk(64), grd((A.N1 + 15) / 16, (rrows + 3) / 4);           // blocks of 16 x 4 pixels (modes 0 and 2)
            if (od.mode == 3) grd = dim3(2 * device_cu_count(), 1);
            if (od.mode == 2 && od.sub > 1) grd = dim3(((A.N1 + od.sub - 1) / od.sub + 63) / 64, (rrows + od.sub - 1) / od.sub);   // the sample lattice
            if (od.mode == 1) { grd = dim3((32 * od.tc + 63) / 64, od_seed_count(od.ntx, od.c0) * od_seed_count(od.nty, od.r0)); if (!grd.y) return hipSuccess; }
            tic(7);
            if (kind == 1) hipLaunchKernelGGL((replay_cost_kernel<1>), grd, blk, 0, s, dev, (const double*)st.table, MA.slot_stride, drow0, N1d, xi_lo, rrows, A, od);
            else hipLaunchKernelGGL((replay_cost_kernel<0>), grd, blk, 0, s, dev, (const double*)st.table, MA.slot_stride, drow0, N1d, xi_lo, rrows, A, od);
            toc();
            return hipGetLastError();
        };

        OdArgs od;
        memset(&od, 0, sizeof(od));
        od.tc = CL.tc; od.ub = CL.ub; od.nbatch = CL.nbatch; od.npass = CL.npass; od.ntx = CL.ntx; od.nty = CL.nty;
        od.ub_inv = (65536 + CL.ub - 1) / CL.ub; od.nrow_inv = (65536 + CL.nrow - 1) / CL.nrow;
        od.tr = UMPA_TILE;
        OdBuffers OB;
        if (od_enabled(ntiles, CL.npass, true) && !MA.ablate) {
            if (od_reserve(st, ntiles, CL.npass, (size_t)A.N0 * A.N1, od, OB)) return -3;
            if (od_lattice_wanted()) {
                od.alone = od_alone_wanted();
                od.central = od_central_passes(ms, dev.ref_mode ? -1 : 1, CL.nrow, CL.nbatch, CL.ub, UJ);
                e = od_run_chunk_lattice(od, OB, s, corr, replay, od_lattice_sub(4));
            } else e = od_run_chunk(od, OB, s, corr, replay);
            if (e != hipSuccess) return (int)e;
            int* slot = st.od_host + UMPA_OD_NCNT * (st.od_slot++ % UMPA_OD_SLOTS);
            if ((e = hipMemcpyAsync(slot, OB.counters, UMPA_OD_NCNT * sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return (int)e;
            if (st.stat_n < 64) st.stat_slots[st.stat_n++] = slot;
            if (tt) for (auto it = tt->entries.rbegin(); it != tt->entries.rend(); ++it)
                if (it->name == 6) { it->counts = slot; it->fma_per = CL.fma_per_pass; break; }
        } else {
            od.mode = 0;
            if ((e = corr(od)) != hipSuccess) return (int)e;
            if ((e = replay(od)) != hipSuccess) return (int)e;
        }
        st.stat_total_passes += (double)ntiles * CL.npass;
        if (xi_hi > xi_lo && on_rows) on_rows(xi_lo, xi_hi);
    }
    return 0;
}

} // namespace umpa

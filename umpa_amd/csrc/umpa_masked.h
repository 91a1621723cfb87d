// umpa_masked.h -- the tiled path for models with masks (Model.cpp:461-499 NoDF, :775-847 DF; combine_weights,
// Utils.cpp:125-130): the exhaustive table of FINISHED costs for all (2 max_shift - 1)^2 integer shifts.
//
// With masks every window element carries the pair weight  w_k(q,u) = pw(M_k(x_ref), M_k(x_sam)),
// pw(a,b) = a b / (a + b + 1e-8), so none of the six sums of the cost depends on one window position alone:
//   t1 = sum_k W[w s^2]   t3 = sum_k W[w r^2]   t5 = sum_k W[w s r]   wt = sum_k W[w]          (W = the window filter)
// and for the dark-field model, whose reference mean is NOT mask-weighted (Model.cpp:804-808),
//   t2 = sum_k mean_k^2 W[w]   t4 = sum_k mean_k W[w s]   t6 = sum_k mean_k W[w r]              (:843-845)
// i.e. three filtered planes per frame and shift, folded into t2, t4, t6 right after the filter.
//
// corr_masked: one workgroup (512 threads, one per CU) = one (tile, pass); a pass is one row offset and UB column
// offsets.  Per frame the sample, reference and mask patches are staged by LDS-DMA.  Product threads own a patch row
// and QB columns: the pair weight once per (element, shift, frame); the planes w s^2, w r^2, w s r accumulated over
// the frames in registers.
//   NoDF: w is accumulated too; two frame slots, the DMA of frame k+1 flies while frame k is multiplied.
//   DF:   w, w s, w r of THIS frame go through the separable filter (planes -> LDS transposed, column filter in place,
//         row filter); the 512 row-filter items are (plane, 8 rows, column) for the three planes plus a fourth set on
//         plane w; their threads keep  sum_k mean_k^2 W[w], sum_k mean_k W[w s], sum_k mean_k W[w r], sum_k W[w]
//         for the pixels they own.  The means of two frames (they are stored as frame pairs) ride in with the even
//         frame's DMA.  One frame slot: the DMA of frame k+1 flies during the last filter round of frame k.
// After the last frame the accumulated planes are filtered, the sums meet in LDS and the closed-form solve
// (Model.cpp:849-858 / :502-505) writes (cost, T, v) to the table.  replay_cost then replays the walk on finished costs.
//
// Work per (pixel, shift, frame), DF: ~13 issue slots of products (x1.8 halo) + 3 planes x (11 x 1.31 + 11) filter
// FMAs + 4: ~100, against ~22 x (2Nw+1)^2 = 2662 per EVALUATION of the general kernel's explicit sum.
#pragma once
#include "umpa_corr.h"

namespace umpa {

struct MaskedArgs {
    double* table;            // [(2ms-1)^2][NV][rows][N1], NV = 2 (cost, T) or 3 (cost, T, v)
    size_t slot_stride;       // rows * N1
    const double* MR;         // per-frame reference means, frame-pair planes (Maps, umpa_tiled.h); DF only
    int H, W;                 // size of a map plane
    int org0, org1;           // frame coordinates of output pixel (0,0) of the REGION
    int row0, rows;           // this launch covers region rows [row0, row0+rows)
    int N1;
    int sigma;                // +1 'sam' mode: A = sample at p, B = reference at p+u; -1 'ref' mode: A = reference at p, B = sample at p-u
    int ntx, nty;
    int br0, br1, bc0, bc1, Wf;
    int binary;               // every mask value is 0 or 1 (checked on the device when the model is created): pw(a, b) = a b / (2 + 1e-8)
    int ablate;               // diagnostics only (UMPA_HIP_ABLATE_MASKED): 1 means = 1, 2 weight = mask product, 4 no column filter, 8 no row filter
};

// TC = 32: one 512-thread workgroup per CU; TC = 16: 256 threads on 32 x 16 tiles, two workgroups per CU (their barrier-
// separated phases overlap each other; 21 % more halo in the products, a quarter more staging per pixel)
template <int KIND, int NW, int UB, int TC_ = UMPA_TILE>
struct MaskCfg {
    static constexpr int TC = TC_, NT = TC_ == 32 ? 512 : 256, WPC = TC_ == 32 ? 1 : 2, TR = UMPA_TILE, S = 2 * NW + 1, NPX = TR * TC;
    static constexpr int NV = KIND == 1 ? 3 : 2;
    static constexpr int QR = TR + 2 * NW, QC = TC + 2 * NW, QP = QR | 1;
    static constexpr int PPL = QC * QP + ((QR - QC * QP) % 32 + 32) % 32;       // as CorrCfg::PPL
    static constexpr int QB = (QR * ((QC + 3) / 4) <= NT) ? 4 : 6;              // columns per product thread (even)
    static constexpr int NQB = (QC + QB - 1) / QB;
    static constexpr int NBP = (UB - 1) / 2 + QB / 2 + 1;                       // B column pairs a thread may read: (u >> 1) + t, t <= QB/2
    static constexpr int BW = (QC + UB) & ~1;
    static constexpr int PA = (QC / 2) | 1, PB = (BW / 2) | 1;                  // 16-byte pieces per image row: odd
    // one staged frame: images A, mask under A, B, mask under B, in piece order
    // (each image starts at a multiple of 64 pieces: a wave-instruction of the LDS-DMA reads one array only and its base is a
    // scalar, as in corr_volume)
    static constexpr int IMG_A = (QR * PA + 63) & ~63, IMG_B = (QR * PB + 63) & ~63;
    static constexpr int OFF_MA = IMG_A, OFF_B = 2 * IMG_A, OFF_MB = 2 * IMG_A + IMG_B;
    static constexpr int NPIECE = 2 * IMG_A + IMG_B + QR * PB;
    static constexpr int NPT = (NPIECE + NT - 1) / NT;
    static constexpr int OVER = 2 * (NQB * (QB / 2) - (QB / 2) + NBP) - 2 * PB;  // the last block's reads past its image row
    static constexpr int SLOT_IMG = NPIECE * 2 + (OVER > 0 ? OVER : 0);
    static constexpr int SLOT_DMA = NPT * NT * 2;
    static constexpr int SLOT = SLOT_IMG > SLOT_DMA ? SLOT_IMG : SLOT_DMA;      // doubles per frame slot
    static constexpr int NSUM = KIND == 1 ? 7 : 4;                              // sums that meet in LDS for the solve
    static constexpr int NPL = KIND == 1 ? 3 : 4;                               // planes per filter round
    // NoDF: the planes and the sums are only needed after the last frame and lie over the frame slots
    static constexpr int NSLOT = (KIND != 1 && 2 * SLOT * 8 <= (UMPA_LDS_BUDGET / WPC)) ? 2 : 1;
    static constexpr int RING = NSLOT * SLOT > NSUM * NPX ? NSLOT * SLOT : NSUM * NPX;
    static constexpr int END_NODF = NSUM * NPX + NPL * PPL;
    // DF: the means of a frame pair under the pixels of the tile, widened by the column offsets of the pass:
    // [TR][MW] pairs, two buffers (the pair after this one arrives while this one is read)
    static constexpr int MW = TC + UB - 1, MUP = TR * MW * 2;
    static constexpr int NMU = (TR * MW + NT - 1) / NT;                         // DMA instructions per thread and frame pair
    static constexpr int MU_DMA = NMU * NT * 2;                                 // doubles one buffer must hold (every lane writes)
    static constexpr int LDS_DF_NOMU = RING + NPL * PPL;
    static constexpr int LDSB = (UMPA_LDS_BUDGET / WPC) & ~15;                  // what one workgroup may use
    // buffers for the means: 2 (the next pair arrives while this one is read), 1 (it arrives at the head of the even frame
    // and is waited for before the first row filter), 0 (read from global memory)
    static constexpr int MUBUFS = KIND != 1 ? 0 : (size_t)(LDS_DF_NOMU + 2 * MU_DMA) * 8 <= (size_t)LDSB ? 2 : (size_t)(LDS_DF_NOMU + MU_DMA) * 8 <= (size_t)LDSB ? 1 : 0;
    static constexpr bool MULDS = MUBUFS > 0;
    static constexpr int LDS_DOUBLES = KIND == 1 ? LDS_DF_NOMU + MUBUFS * MU_DMA : (RING > END_NODF ? RING : END_NODF);
    static constexpr size_t LDS = (size_t)LDS_DOUBLES * sizeof(double);
    static constexpr int CB = 8;
    static constexpr int HITEMS = NPL * (TC / CB) * QR, HROUNDS = (HITEMS + NT - 1) / NT;
    static constexpr int VITEMS = 4 * (TR / CB) * TC;                           // = NT: one item per thread (DF: the fourth set is plane w again)
    static constexpr bool OK = QR * NQB <= NT && LDS <= (size_t)LDSB && VITEMS == NT && NPT <= 16;
};

// all values of a mask stack 0 or 1?  (*flag starts at 1)
__global__ void __launch_bounds__(256)
mask_binary_kernel(const double* __restrict__ mask, size_t n, int* flag)
{
    bool ok = true;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256) {
        const double v = gp(mask)[q];
        ok = ok && (v == 0.0 || v == 1.0);
    }
    if (!ok) *flag = 0;
}

// combine_weights (Utils.cpp:125-130) with v_rcp_f64 and ONE Newton step (relative error ~1e-14; the weight enters
// every sum linearly and identically, so this is a 1e-14 perturbation of the window, not of a cancellation)
__device__ __forceinline__ double pair_weight_fast(double a, double b)
{
    const double d = a + b + 1e-8;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    return a * b * r;
}

// fir_block (umpa_corr.h) with the inputs consumed as they arrive: fewer values live (corr_masked is short of registers)
template <int NW, int CB>
__device__ __forceinline__ void fir_stream(const double* __restrict__ in, int stride, const double* h, double* out)
{
    constexpr int S = 2 * NW + 1;
#pragma unroll
    for (int o = 0; o < CB; o++) out[o] = 0.0;
#pragma unroll
    for (int t = 0; t < CB + S - 1; t++) {
        const double v = in[t * stride];
#pragma unroll
        for (int o = 0; o < CB; o++) {
            const int tap = t - o;
            if (tap >= 0 && tap < S) out[o] = fma(h[tap], v, out[o]);
        }
    }
}

// the finished costs of one (tile, pass) into the table
template <int KIND, int NW, int UB, int TCT>
__device__ __forceinline__ void corr_masked_tile(const ModelDev& m, const MaskedArgs& A, const Sep1D& sep, double* ring,
                                                 const int lin, const int pass, const int tid)
{
    using C = MaskCfg<KIND, NW, UB, TCT>;
    constexpr int NT = C::NT, QB = C::QB, TC = C::TC, TR = C::TR, NPX = C::NPX;
    double* sums = ring;                                                // after the last frame: [NSUM][NPX]
    double* planes = KIND == 1 ? ring + C::RING : ring + C::NSUM * NPX;  // NoDF: over the slots, after the last frame
    double* mubuf = ring + C::LDS_DF_NOMU;                              // DF with MULDS: 2 x MU_DMA
    const int ms = m.ms, UJ = 2 * ms - 1;
    const int nbatch = (UJ + UB - 1) / UB;
  {
    const int tx = lin % A.ntx, ty = lin / A.ntx;
    const int prow0 = A.row0 + ty * TR, pcol0 = tx * TC;
    const int fr0 = A.org0 + prow0 - NW, fc0 = A.org1 + pcol0 - NW;
    const int oi0 = pass / nbatch - (ms - 1), oj0 = (pass % nbatch) * UB - (ms - 1);
    const int nu = min(UB, ms - oj0);                                   // column offsets oj0 .. oj0 + nu - 1 are real

    // ---- LDS-DMA pieces of this thread (see corr_volume): 16 bytes = two adjacent columns of one frame row
    unsigned src_off[C::NPT];
#pragma unroll
    for (int n = 0; n < C::NPT; n++) {
        const int p = tid + n * NT;
        int img, q;
        if (p < C::OFF_MA) { img = 0; q = min(p, C::QR * C::PA - 1); }
        else if (p < C::OFF_B) { img = 1; q = min(p - C::OFF_MA, C::QR * C::PA - 1); }
        else if (p < C::OFF_MB) { img = 2; q = min(p - C::OFF_B, C::QR * C::PB - 1); }
        else { img = 3; q = min(p - C::OFF_MB, C::QR * C::PB - 1); }
        int r, c;
        if (img < 2) { r = q / C::PA; c = 2 * (q % C::PA); }
        else { r = q / C::PB + oi0; c = 2 * (q % C::PB) + oj0; }
        const int gr = min(max(fr0 + r, A.br0), A.br1), gc = min(max(fc0 + c, A.bc0), A.bc1 - 1);
        src_off[n] = (unsigned)(gr * A.Wf + gc) * 8u;
    }
    const unsigned wave_piece0 = (unsigned)__builtin_amdgcn_readfirstlane(tid & ~63);
    unsigned src_sel = 0;                                               // 2 bits per instruction of this wave: 0 A, 1 mask under A, 2 B, 3 mask under B
#pragma unroll
    for (int n = 0; n < C::NPT; n++) {
        const unsigned p0 = wave_piece0 + n * NT;
        src_sel |= (p0 < (unsigned)C::OFF_MA ? 0u : p0 < (unsigned)C::OFF_B ? 1u : p0 < (unsigned)C::OFF_MB ? 2u : 3u) << (2 * n);
    }
    // DF: pieces of the means of a frame pair: piece p = pixel (p / MW, p % MW) of the tile rows x (tile columns widened by the
    // column offsets); the reference window sits at p + u in 'sam' mode, at p in 'ref' mode (Model.cpp:688-701)
    const int mdi = A.sigma > 0 ? oi0 : 0, mdj = A.sigma > 0 ? oj0 : 0;
    unsigned mu_off[C::MULDS ? C::NMU : 1];
    if (C::MULDS) {
#pragma unroll
        for (int n = 0; n < C::NMU; n++) {
            const int p = min(tid + n * NT, TR * C::MW - 1);
            const int r = p / C::MW, c = p % C::MW;
            const int row = min(A.org0 + min(prow0 + r, A.row0 + A.rows - 1) + mdi, A.H - 1);
            const int col = min(A.org1 + pcol0 + c + mdj, A.W - 1);
            mu_off[n] = (unsigned)(row * A.W + col) * 16u;
        }
    }
    const size_t mplane = (size_t)A.H * A.W;
    auto issue_frame = [&](int k) {
        const FrameDesc fd = load_frame(m.frames, k);
        const long shift = ((long)fd.pi * A.Wf + fd.pj) * 8;
        const UMPA_GLOBAL char* gA = (const UMPA_GLOBAL char*)gp(A.sigma > 0 ? fd.sam : fd.ref) - shift;
        const UMPA_GLOBAL char* gB = (const UMPA_GLOBAL char*)gp(A.sigma > 0 ? fd.ref : fd.sam) - shift;
        const UMPA_GLOBAL char* gM = (const UMPA_GLOBAL char*)gp(fd.mask) - shift;
        UMPA_LDS_AS char* slot = (UMPA_LDS_AS char*)(ring + (k % C::NSLOT) * C::SLOT);
#pragma unroll
        for (int n = 0; n < C::NPT; n++) {
            const unsigned sel = (src_sel >> (2 * n)) & 3u;                                     // (wave-uniform)
            lds_dma16(sel == 0 ? gA : sel == 2 ? gB : gM, src_off[n], slot + (size_t)(wave_piece0 + n * NT) * 16);
        }
        // (two buffers for the means: they ride with the even frame; one buffer: see the head of the frame loop)
    };
    auto issue_means = [&](int k) {                                     // the means of frames k and k+1 (k even)
        {
            const UMPA_GLOBAL char* gmu = (const UMPA_GLOBAL char*)gp(A.MR) + (size_t)(k >> 1) * mplane * 16;
            UMPA_LDS_AS char* dst = (UMPA_LDS_AS char*)(mubuf + (C::MUBUFS == 2 ? ((k >> 1) & 1) : 0) * C::MU_DMA);
#pragma unroll
            for (int n = 0; n < C::NMU; n++) {
                // (the source through a local: with the subscript inside the builtin's argument list clang drops the kernel's
                // host stub without a diagnostic, ROCm 7.2)
                const unsigned off = mu_off[n];
                lds_dma16(gmu, off, dst + (size_t)(wave_piece0 + n * NT) * 16);
            }
        }
    };

    // product-stage ownership: (qb, r), r fastest
    const int pr = tid % C::QR, pqb = tid / C::QR;
    const bool pactive = pqb < C::NQB;
    typedef double pair_t __attribute__((ext_vector_type(2)));
    double PAA[QB][UB], PBB[QB][UB], PAB[QB][UB], PW[KIND == 1 ? 1 : QB][UB];
#pragma unroll
    for (int t = 0; t < QB; t++)
#pragma unroll
        for (int u = 0; u < UB; u++) { PAA[t][u] = PBB[t][u] = PAB[t][u] = 0.0; if (KIND != 1) PW[t][u] = 0.0; }

    // filter-stage ownership (row filter): item (plane, row block, column), column fastest; fixed for the whole pass.
    // DF: items of set 3 work on plane 0 (w) again and keep the plain sum (wt).
    const int vc = tid % TC, vrb = (tid / TC) % (TR / C::CB), vset = tid / (TC * (TR / C::CB));
    const int vpl = (KIND == 1 && vset == 3) ? 0 : vset;
    double vacc[KIND == 1 ? UB : 1][C::CB];
#pragma unroll
    for (int u = 0; u < (KIND == 1 ? UB : 1); u++)
#pragma unroll
        for (int o = 0; o < C::CB; o++) vacc[u][o] = 0.0;

    // One filter round on the NPL planes in `planes` (already written, transposed [column][row]), in two halves: the
    // column filter in place, then the row filter, whose 8 outputs of this thread's item come back in `out`.
    auto filter_cols = [&]() {
        __syncthreads();                                                // plane writes done
        if (A.ablate & 4) return;
        double hres[C::HROUNDS][C::CB];
#pragma unroll
        for (int rd = 0; rd < C::HROUNDS; rd++) {
            const int it = tid + rd * NT;
            if (it < C::HITEMS) {
                const int r = it % C::QR, rest = it / C::QR, pl = rest % C::NPL, cb = rest / C::NPL;
                fir_stream<NW, C::CB>(planes + pl * C::PPL + (cb * C::CB) * C::QP + r, C::QP, sep.hc, hres[rd]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int rd = 0; rd < C::HROUNDS; rd++) {
            const int it = tid + rd * NT;
            if (it < C::HITEMS) {
                const int r = it % C::QR, rest = it / C::QR, pl = rest % C::NPL, cb = rest / C::NPL;
                double* dst = planes + pl * C::PPL + (cb * C::CB) * C::QP + r;
#pragma unroll
                for (int o = 0; o < C::CB; o++) dst[o * C::QP] = hres[rd][o];
            }
        }
        __syncthreads();
    };
    auto filter_rows = [&](double* out) {
        if (A.ablate & 8) {
#pragma unroll
            for (int o = 0; o < C::CB; o++) out[o] = planes[vpl * C::PPL + vc * C::QP + vrb * C::CB + o];
            return;
        }
        fir_stream<NW, C::CB>(planes + vpl * C::PPL + vc * C::QP + vrb * C::CB, 1, sep.hr, out);
    };

    const int vcol = min(pcol0 + vc, A.N1 - 1);                         // (direct loads of the means, where they do not ride in LDS)

    const int K = m.Na;
    issue_frame(0);
    if (C::MUBUFS == 2) issue_means(0);
    for (int k = 0; k < K; k++) {
        if (C::NSLOT == 2 && k + 1 < K) {
            issue_frame(k + 1);                                         // its slot was last read in frame k-1: the barrier below is behind us
            wait_vmcnt<C::NPT>();                                       // frame k has landed, frame k+1 may still be on its way
        } else wait_vmcnt<0>();
        lds_barrier();                                                  // frame k is in for everyone (and frame k-1 has been read for the last time)
        if (C::MUBUFS == 1 && (k & 1) == 0) issue_means(k);             // one buffer for the means: free now, needed at the first row filter
        const pair_t* img = reinterpret_cast<const pair_t*>(ring + (k % C::NSLOT) * C::SLOT);
        const pair_t* la = img + pr * C::PA + pqb * (QB / 2);
        const pair_t* lb = img + C::OFF_B + pr * C::PB + pqb * (QB / 2);
        // the pair weight and the products of column offset u (operands out of the slot each time: registers are what this
        // kernel is short of); DF hands back this frame's w, w a, w b
        auto products = [&](int u, double* w_, double* wa_, double* wb_) {
            pair_t av[QB / 2], mav[QB / 2], bv[QB / 2 + 1], mbv[QB / 2 + 1];
#pragma unroll
            for (int t = 0; t < QB / 2; t++) { av[t] = la[t]; mav[t] = la[C::OFF_MA + t]; }
#pragma unroll
            for (int t = 0; t < QB / 2 + 1; t++) { bv[t] = lb[(u >> 1) + t]; mbv[t] = lb[C::OFF_MB - C::OFF_B + (u >> 1) + t]; }
#pragma unroll
            for (int t = 0; t < QB; t++) {
                const int tb = t + (u & 1);
                const double a = av[t >> 1][t & 1], ma = mav[t >> 1][t & 1];
                const double b = bv[tb >> 1][tb & 1], mb = mbv[tb >> 1][tb & 1];
                // 0/1 masks (the usual bad-pixel masks): a b / (a + b + 1e-8) is 0 or 1 / (2 + 1e-8) -- one multiply instead of
                // a reciprocal and a Newton step, the same number to the last bits
                const double w = A.binary ? ma * mb * (1.0 / (2.0 + 1e-8)) : (A.ablate & 2) ? ma * mb : pair_weight_fast(mb, ma);
                const double wa = w * a, wb = w * b;
                PAA[t][u] = fma(wa, a, PAA[t][u]);
                PBB[t][u] = fma(wb, b, PBB[t][u]);
                PAB[t][u] = fma(wa, b, PAB[t][u]);
                if (KIND == 1) { w_[t] = w; wa_[t] = wa; wb_[t] = wb; }
                else PW[t][u] += w;
            }
        };
        if (KIND != 1) {
            if (pactive) {
#pragma unroll
                for (int u = 0; u < UB; u++) products(u, nullptr, nullptr, nullptr);
            }
            lds_barrier();                                              // the slot has been read by everyone
            if (C::NSLOT == 1 && k + 1 < K) issue_frame(k + 1);
            continue;
        }
        // ---- DF: one filter round per column offset on this frame's w, w a, w b
        const pair_t* mup = reinterpret_cast<const pair_t*>(mubuf + (C::MUBUFS == 2 ? ((k >> 1) & 1) : 0) * C::MU_DMA);
#pragma unroll
        for (int u = 0; u < UB; u++) {
            if (u < nu) {                                               // (uniform)
                double w_[QB], wa_[QB], wb_[QB];
                if (pactive) products(u, w_, wa_, wb_);
                if (u == nu - 1) {                                      // last read of the slot: frame k+1 flies during this round
                    lds_barrier();
                    if (k + 1 < K) { issue_frame(k + 1); if (C::MUBUFS == 2 && ((k + 1) & 1) == 0) issue_means(k + 1); }
                } else if (u > 0) __syncthreads();                      // the previous round's row filter has read the planes
                if (pactive) {
#pragma unroll
                    for (int t = 0; t < QB; t++) {
                        const int c = pqb * QB + t;
                        if (c < C::QC) {
                            planes[0 * C::PPL + c * C::QP + pr] = w_[t];
                            planes[1 * C::PPL + c * C::QP + pr] = wa_[t];
                            planes[2 * C::PPL + c * C::QP + pr] = wb_[t];
                        }
                    }
                }
                filter_cols();
                if (C::MUBUFS == 1 && u == 0 && (k & 1) == 0) {             // the means issued at the head of this frame have landed for everyone
                    wait_vmcnt<0>();
                    __syncthreads();
                }
                // the means of this item's pixels at the reference window (Model.cpp:808)
                double mu[C::CB];
                if (A.ablate & 1) {
#pragma unroll
                    for (int o = 0; o < C::CB; o++) mu[o] = 1.0;
                } else if (C::MULDS) {
#pragma unroll
                    for (int o = 0; o < C::CB; o++) mu[o] = mup[(vrb * C::CB + o) * C::MW + vc + (A.sigma > 0 ? u : 0)][k & 1];
                } else {
#pragma unroll
                    for (int o = 0; o < C::CB; o++) {
                        const int row = min(prow0 + vrb * C::CB + o, A.row0 + A.rows - 1);
                        const size_t x = (size_t)(A.org0 + row + mdi) * A.W + (A.org1 + vcol + mdj + (A.sigma > 0 ? u : 0));
                        mu[o] = gp(A.MR)[((size_t)(k >> 1) * mplane + x) * 2 + (k & 1)];
                    }
                }
                double out[C::CB];
                filter_rows(out);
#pragma unroll
                for (int o = 0; o < C::CB; o++) {
                    const double f = vset == 0 ? mu[o] * mu[o] : vset == 3 ? 1.0 : mu[o];
                    vacc[u][o] = fma(f, out[o], vacc[u][o]);
                }
            }
        }
        __syncthreads();                                                // planes free for the next frame's first round
    }

    // ---- all frames are in: filter the accumulated planes of each column offset, let the sums meet in LDS, solve, store
#pragma unroll
    for (int u = 0; u < UB; u++) {
        if (u < nu) {
            __syncthreads();                                            // planes and sums of the previous round consumed
            if (pactive) {
#pragma unroll
                for (int t = 0; t < QB; t++) {
                    const int c = pqb * QB + t;
                    if (c < C::QC) {
                        planes[0 * C::PPL + c * C::QP + pr] = PAA[t][u];
                        planes[1 * C::PPL + c * C::QP + pr] = PBB[t][u];
                        planes[2 * C::PPL + c * C::QP + pr] = PAB[t][u];
                        if (KIND != 1) planes[3 * C::PPL + c * C::QP + pr] = PW[t][u];
                    }
                }
            }
            filter_cols();
            if (KIND != 1 || vset < 3) {
                double out[C::CB];
                filter_rows(out);
#pragma unroll
                for (int o = 0; o < C::CB; o++) sums[vset * NPX + (vrb * C::CB + o) * TC + vc] = out[o];   // 0: W[w a a], 1: W[w b b], 2: W[w a b], (3: W[w])
            }
            if (KIND == 1) {
#pragma unroll
                for (int o = 0; o < C::CB; o++) sums[(3 + vset) * NPX + (vrb * C::CB + o) * TC + vc] = vacc[u][o];   // 3: t2, 4: sum mean W[w a], 5: sum mean W[w b], 6: wt
            }
            __syncthreads();
            const int ui = A.sigma * oi0, uj = A.sigma * (oj0 + u);
            const size_t slot = (size_t)(ui + ms - 1) * UJ + (uj + ms - 1);
            for (int px = tid; px < NPX; px += NT) {
                const int r = px / TC, c = px % TC;
                const int row = prow0 + r, col = pcol0 + c;
                if (row >= A.row0 + A.rows || col >= A.N1) continue;
                const double aa = sums[px], bb = sums[NPX + px], t5 = sums[2 * NPX + px];
                const double t1 = A.sigma > 0 ? aa : bb, t3 = A.sigma > 0 ? bb : aa;   // s = A in 'sam' mode, B in 'ref' mode
                UMPA_GLOBAL double* dst = gpw(A.table) + slot * C::NV * A.slot_stride + (size_t)(row - A.row0) * A.N1 + col;
                if (KIND == 1) {
                    const double t2 = sums[3 * NPX + px], ma = sums[4 * NPX + px], mb = sums[5 * NPX + px], wt = sums[6 * NPX + px];
                    const double t4 = A.sigma > 0 ? ma : mb, t6 = A.sigma > 0 ? mb : ma;
                    const double det = t2 * t3 - t6 * t6;               // Model.cpp:849-858
                    const double Kc = (t2 * t5 - t4 * t6) / det;
                    const double beta = (t3 * t4 - t5 * t6) / det;
                    const double T = beta + Kc;
                    dst[0] = (t1 + beta * beta * t2 + Kc * Kc * t3 - 2 * beta * t4 - 2 * Kc * t5 + 2 * beta * Kc * t6) / wt;
                    dst[A.slot_stride] = T;
                    dst[2 * A.slot_stride] = Kc / T;
                } else {
                    const double wt = sums[3 * NPX + px];
                    const double T = t5 / t3;                           // Model.cpp:502-505
                    dst[0] = (t1 - t5 * T) / wt;
                    dst[A.slot_stride] = T;
                }
            }
        }
    }
  }
}

// one workgroup = one (tile, pass) of a static grid (od_static_item, umpa_corr.h)
template <int KIND, int NW, int UB, int TCT>
__global__ void __launch_bounds__((MaskCfg<KIND, NW, UB, TCT>::NT), 2)
corr_masked_kernel(ModelDev m, MaskedArgs A, Sep1D sep, OdCorr od)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int UJ = 2 * m.ms - 1, npass = UJ * ((UJ + UB - 1) / UB);
    int lin, pass;
    if (!od_static_item(od, A.ntx, A.ntx * A.nty, npass, lin, pass)) return;
    corr_masked_tile<KIND, NW, UB, TCT>(m, A, sep, reinterpret_cast<double*>(smem_raw), lin, pass, threadIdx.x);
    od_mark_done(od, lin, pass);
}

// the same over a work list, by a persistent grid (the repair step of umpa_ondemand.h)
template <int KIND, int NW, int UB, int TCT>
__global__ void __launch_bounds__((MaskCfg<KIND, NW, UB, TCT>::NT), 2)
corr_masked_queue_kernel(ModelDev m, MaskedArgs A, Sep1D sep, OdCorr od)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nitems = __builtin_amdgcn_readfirstlane(*gp(od.nitems));
    for (int j = 0;; j++) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                                   // (what follows from it is not to be hoisted out of the loop)
        const int idx = od_item_index(nitems, blockIdx.x, gridDim.x, j);
        if (idx < 0) break;
        const int it = __builtin_amdgcn_readfirstlane(gp(od.items)[idx]);
        if (j) __syncthreads();                                         // the previous item's last LDS reads
        corr_masked_tile<KIND, NW, UB, TCT>(m, A, sep, reinterpret_cast<double*>(smem_raw), it >> 8, it & 255, tid);
        od_mark_done(od, it >> 8, it & 255);
    }
}

// ------------------------------------------------------------------------------------------------
// replay_cost: the walk on a table of finished costs (one entry = cost, T and, for DF, v)
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(64, 3)
replay_cost_kernel(ModelDev m, const double* table, size_t slot_stride, int drow0, int N1d, int row0, int rows, RegionArgs A, OdArgs od)
{
    __shared__ double memo_lds[25 * 64];
    constexpr int NV = KIND == 1 ? 3 : 2;
    const LdsMemo<64> memo = {memo_lds + threadIdx.x};
    const int nparked = od.mode == 3 ? __builtin_amdgcn_readfirstlane(gp(od.cnt_in)[OD_C_PX]) : 0;       // od.mode: as replay_walk (umpa_tiled.h)
  for (int q = (blockIdx.y * gridDim.x + blockIdx.x) * 64 + threadIdx.x, first = 1;; q += gridDim.x * gridDim.y * 64, first = 0) {
    int xi, xj;
    bool live;
    if (od.mode == 3) {
        if (__builtin_amdgcn_readfirstlane(q & ~63) >= nparked) break;
        live = q < nparked;
        const int pxq = live ? gp(od.px_in)[q] : 0;
        xi = pxq / A.N1; xj = pxq - xi * A.N1;
    } else if (od.mode == 1) {
        if (!first) break;
        live = od_seed_pixel(od, drow0, A.step0, A.step1, xi, xj);
        live = live && xi >= row0 && xi < row0 + rows && xj < A.N1;
    } else {
        if (!first) break;
        xj = blockIdx.x * 64 + threadIdx.x;
        xi = blockIdx.y;
        if (od.sub > 1) { xj = xj * od.sub + (od.sub >> 1); xi = xi * od.sub + (od.sub >> 1); }   // the sample lattice (od_run_chunk_lattice)
        else {                                                          // a wave = a block of 16 x 4 pixels (as replay_walk, ReplayArgs::bw_log2)
            xj = (blockIdx.x << 4) + (threadIdx.x & 15);
            xi = (blockIdx.y << 2) + (threadIdx.x >> 4);
        }
        xi += row0;
        live = xi < row0 + rows && xj < A.N1;
    }
    const size_t px = (size_t)xi * A.pitch + xj;                        // in the output arrays
    const size_t tpx = (size_t)(xi * A.step0 - drow0) * N1d + (size_t)xj * A.step1;
    if (live && A.cover && gp(A.cover)[px] < A.thr) live = false;
    OdLane L;
    if (live && !od_begin(od, xi * A.step0 - drow0, xj * A.step1, L)) live = false;
    if (live) {
        Walk w;
        walk_begin(w, memo, A.uv ? gp(A.uv)[2 * px] : 0.0, A.uv ? gp(A.uv)[2 * px + 1] : 0.0);
        const int ms = m.ms, UJ = 2 * ms - 1, sigma = m.ref_mode ? -1 : 1;
        while (w.phase < PH_FIT) {
            double c = 0.0;
            Fit fit = w.live;
            const int si = w.req_i, sj = w.req_j;
            int st = UMPA_ST_OK;
            if (si <= -ms || si >= ms) st = UMPA_ST_BOUND;
            else if (sj <= -ms) st = UMPA_ST_BOUND | UMPA_ST_DIM;
            else if (sj >= ms) st = UMPA_ST_BOUND | UMPA_ST_DIM | UMPA_ST_POSITIVE;
            else {
                if (!od_check(od, L, ms, sigma, si, sj)) break;         // the plane is not there: this pixel is parked
                const size_t slot = (size_t)((si + ms - 1) * UJ + (sj + ms - 1));
                const UMPA_GLOBAL double* e = gp(table) + slot * NV * slot_stride + tpx;
                c = e[0];
                fit.t = e[slot_stride];
                fit.v = KIND == 1 ? e[2 * slot_stride] : 0.0;
            }
            walk_feed(w, memo, st, c, fit, m.call_cap);
        }
        if (L.miss) od_park(od, L, xi * A.N1 + xj);
        else if (od.sub <= 1) {                                         // (the sample lattice's walks only predict: A.uv is read AND written)
            double nb[16];
            walk_finish(w, memo, m.subpx, nb);
            store_pixel(A, px, KIND, w, memo, nb);
        }
    }
    if (od.mode == 1) od_record_visited(od, L, live);
  }
}

} // namespace umpa

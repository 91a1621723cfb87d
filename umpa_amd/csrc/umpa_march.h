// umpa_march.h -- corr_march, the table kernel of the tiled path for wide windows (round 4): the numbers of corr_volume,
//   t5[u][p] = sum_k W[ s_k(.) r_k(.+u) ](p)                              (Model.cpp:763-772, window = hr (x) hc, model.pyx:691-696)
// by MARCHING instead of tiling.  A workgroup owns a column strip of the region and a band of its rows and walks down the
// patch rows of the band one row per step; a pass is NUY consecutive row offsets x every column offset of the search range.
//
//   * 16 lanes (one DPP row) x 4 columns = the 64 patch columns of ONE plane (shift); the compute waves of a workgroup hold
//     NUY * (2 ms - 1) planes.  Per step a thread forms its 4 products summed over the frames (operands out of rings of staged
//     rows: A by four ds_read_b64, B by two ds_read_b128 per frame), feeds them into the window's ROW filter kept as a ring of 2 Nw + 1 running sums in registers
//     (out[r] += hr[t] * P[r + t]: the sum an output row is waiting for grows by one tap per step), and runs the row that
//     completes through the COLUMN filter with its neighbours' values fetched by DPP row shifts (lane l + n of the same 16).
//     Nothing of the two filters goes through LDS: no plane writes, no transposes, no flush, one barrier per step.
//   * No vertical halo: a patch row is staged and multiplied once per pass, not once per 32-row tile; the horizontal halo is
//     64 / WO, WO = 48 at Nw = 6 and 7.
//   * Staging (LDS-DMA, 1 KiB per wave-instruction) is spread over ALL waves: instruction n of a step belongs to wave n mod NWV,
//     the rows LA steps ahead.  A wave's table stores count in the same in-order vmcnt as its staging: the counted wait at
//     the head of a step knows how many of either were issued since (exact counts, not a drain).  (One dedicated staging
//     wave was tried: an LDS-DMA instruction costs its wave about 140 cycles to issue, C3's 23 per step bound the kernel.)
//   * LDS images are laid out for the reads: a row of A or B is [frame][even pairs | odd pairs] (16-byte column pairs), so
//     that the 16 lanes of a plane read 256 contiguous bytes per instruction.  A plane at an ODD column shift works one
//     column to the left (columns 4l-1 .. 4l+2): its B reads are then the 16-byte aligned ones of the even shift below,
//     its A columns come as four ds_read_b64, and the finished row moves back by one lane-column on its way out (DPP).
//     The two DPP rows that share the lane groups of an LDS instruction are always the planes (2m, 2m+1) of one row offset
//     (or the last, even column offset of two row offsets): they read the same B addresses on complementary lanes and A
//     columns on complementary halves of the 16-byte slots -- no bank conflicts (SQ_LDS_BANK_CONFLICT = 0).
//   * The table is blocked by strips, [strip][row][shift][TW columns], TW * 8 bytes a multiple of 128 where the window
//     allows: a plane's row is a run of whole, aligned lines (tools/microbench/table_store_rate.hip: 54-column runs at the
//     region's row pitch 1.3-2.9 TB/s, aligned runs 5.3 TB/s; plain stores: the L2 puts the two halves of a 64-byte piece
//     together, non-temporal ones send them out separately).  replay_walk addresses either layout (ReplayArgs::strip_w).
//
// Where it stands (tools/microbench/march_dev.hip, one MI355X): C3 (4096 x 4096, 20 frames, Nw = 7, max_shift = 8) 24.7-25.8 ms
// for the whole table against corr_volume's 28.9: LDS 45 % busy, VALU about 59 %, waves waiting 47 % of their cycles -- one
// workgroup of 12 waves per CU (168 VGPRs: the ring of a 15-wide window alone takes 112) leaves nothing to run beside a
// barrier.  C2 (Nw = 5, two workgroups of 7 waves per CU): 1.29-1.46 ms against corr_volume's 1.28 -- level, so windows up
// to 11 pixels stay on corr_volume.  What both kernels run into is the CU's vector-memory path: staged bytes plus stored
// bytes leave at about 19 GB/s per CU (4.9 TB/s over the chip) whichever kernel issues them.
//
// Sums are formed in another order than corr_volume's (frames first, then rows, then columns); parity is on results.
#pragma once
#include "umpa_corr.h"

namespace umpa {

struct MarchArgs {
    double* table;            // [strip][rows][(2ms-1)^2][tw]
    int tw;                   // columns of a table block (64, or WO where WO * 8 is a multiple of 128)
    int org0, org1;           // frame coordinates of output pixel (0,0) of the REGION
    int row0, rows;           // this launch covers dense region rows [row0, row0+rows)
    int N1;
    int sigma;                // +1: B = ref sits at p+u ('sam' mode); -1: B = sam sits at p-u ('ref' mode)
    int br0, br1, bc0, bc1, Wf;   // image rows / columns inside every frame, the frames' common width
    int nstrips, nbands, band_rows, npass, nuy;
    int npa, npb;             // LDS-DMA instructions per step: A pieces (64 per frame: whole instructions), B pieces (36+ per frame)
    unsigned a_slot, b_slot;  // bytes of one A / B row slot (b_slot a multiple of 256)
    int da, db;               // ring depths (rows): LA + 1, nuy + LA
    const char* baseA;        // byte address the A / B stack's frame offsets count from
    const char* baseB;
    const unsigned* frame_off;   // [2][Na]: byte offset of frame k's image origin (position folded in) from baseA / baseB
    int ablate;               // diagnostics: 1 no DMA, 2 no frame loop, 4 no filters, 8 no stores
    // which (strip, band, pass) units the launch computes (on-demand passes, umpa_ondemand.h: a unit = a "tile" lin = band *
    // nstrips + strip and a pass):
    //   items == nullptr: the static grid, every (strip, band) x every pass;
    //   items != nullptr: the work list ((lin << 8) | pass), *nitems entries; the grid has a slot for every unit there could be
    //   and the workgroups past the list's end leave at once.
    // done / ndone (or null): the unit's bit is set in done[lin] and *ndone counted when its planes are written (read by later launches).
    const int* items;
    const int* nitems;
    unsigned long long* done;
    int* ndone;
};

template <int NW, int NXB>
struct MarchCfg {
    static constexpr int S = 2 * NW + 1, PW = 64;
    static constexpr int WO = (63 - 2 * NW) / 4 * 4;      // output columns of a strip (a plane at an odd shift has 63 patch columns)
    static constexpr int NBE = 16 + NXB;                  // entries of one B array: column shifts up to 4 * NXB
    static constexpr int APF = 32, BPF = 2 * NBE;         // pieces per frame and row: A (E, O), B (E, O)
    static constexpr int NNB = (2 * NW + 3) / 4;          // neighbour lanes the column filter reaches
    static constexpr int TW = WO;                         // columns of a table block
};

template <int N>
__device__ __forceinline__ double dpp_row_shl(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + N, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + N, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void wait_vmcnt_rt(int n)     // (wave-uniform n)
{
#define UMPA_W(i) case i: wait_vmcnt<i>(); break;
    switch (n) {
    UMPA_W(0) UMPA_W(1) UMPA_W(2) UMPA_W(3) UMPA_W(4) UMPA_W(5) UMPA_W(6) UMPA_W(7) UMPA_W(8) UMPA_W(9) UMPA_W(10) UMPA_W(11)
    UMPA_W(12) UMPA_W(13) UMPA_W(14) UMPA_W(15) UMPA_W(16) UMPA_W(17) UMPA_W(18) UMPA_W(19) UMPA_W(20) UMPA_W(21) UMPA_W(22)
    UMPA_W(23) UMPA_W(24) UMPA_W(25) UMPA_W(26) UMPA_W(27) UMPA_W(28) UMPA_W(29) UMPA_W(30) UMPA_W(31) UMPA_W(32) UMPA_W(33)
    UMPA_W(34) UMPA_W(35) UMPA_W(36) UMPA_W(37) UMPA_W(38) UMPA_W(39) UMPA_W(40) UMPA_W(41) UMPA_W(42) UMPA_W(43) UMPA_W(44)
    UMPA_W(45) UMPA_W(46) UMPA_W(47) UMPA_W(48)
    default: wait_vmcnt<0>(); break;
    }
#undef UMPA_W
}

// NT threads, every wave computes and stages.  NPT: upper bound of the LDS-DMA instructions of one wave per step.
// SPB: steps per workgroup barrier.  1: a barrier per step, rows LA steps ahead (ring depths da = LA + 1, db = nuy + LA).
// 2 and more: the steps come in intervals of SPB; at the head of an interval the rows of the whole interval have landed (they
// were issued at the head of the one before), one barrier, the rows of the next interval are issued; inside an interval the
// waves run free, up to SPB steps apart (da = 2 SPB, db = nuy - 1 + 2 SPB).
template <int NW, int NXB, int NPT, int LA, int NT, int WPS, int SPB = 1>
__global__ void __launch_bounds__(NT, WPS)
corr_march_kernel(ModelDev m, MarchArgs A, Sep1D sep)
{
    using C = MarchCfg<NW, NXB>;
    constexpr int S = C::S, NWV = NT / 64;
    typedef double pair_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    UMPA_LDS_AS char* const lds = (UMPA_LDS_AS char*)smem_raw;

    const int tid = threadIdx.x;
    const int ms = m.ms, UJ = 2 * ms - 1, K = m.Na;
    // ---- (strip, band, pass) of this workgroup: the passes of one (strip, band) on consecutive slots of one XCD
    int item, pass;
    if (A.items) {                                                    // the list cut into 8 contiguous ranges, one per XCD
        const int n = __builtin_amdgcn_readfirstlane(gp(A.nitems)[0]), per_xcd = (n + 7) >> 3;
        const int seq = blockIdx.x >> 3, idx = (blockIdx.x & 7) * per_xcd + seq;
        if (seq >= per_xcd || idx >= n) return;
        const int it = __builtin_amdgcn_readfirstlane(gp(A.items)[idx]);
        item = it >> 8; pass = it & 255;
    } else {
        const int nitems = A.nstrips * A.nbands, per_xcd = (nitems + 7) >> 3;
        const int seq = blockIdx.x >> 3;
        item = (blockIdx.x & 7) * per_xcd + seq / A.npass; pass = seq % A.npass;   // (items to the XCDs in turn instead: no difference, C3 26.6 / 26.3 ms)
        if (seq / A.npass >= per_xcd || item >= nitems) return;
    }
    const int strip = item % A.nstrips, band = item / A.nstrips;
    const int r_lo = band * A.band_rows, r_hi = min(A.rows, r_lo + A.band_rows);   // rows of this launch's chunk
    if (r_lo >= r_hi) return;
    const int nsteps = r_hi - r_lo + 2 * NW;
    const int c_lo = strip * C::WO;                                   // first output column of the strip
    const int oilo = pass * A.nuy - (ms - 1);                         // first row offset of the pass
    const int fr0 = A.org0 + A.row0 + r_lo - NW, fc0 = A.org1 + c_lo - NW;   // frame coords of patch (row 0, column 0)
    const unsigned ldsA = 0, ldsB = A.da * A.a_slot;

    // ---- staging: instruction n of a step (64 pieces of 16 bytes; A: 32 pieces per frame, two frames per instruction; then
    // B: 2 NBE pieces per frame) belongs to wave n mod NWV
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int npd = A.npa + A.npb;
    unsigned src_off[NPT];
    int cnt_w = 0;                                                    // instructions of this wave per step (wave-uniform)
#pragma unroll
    for (int i = 0; i < NPT; i++) {
        const int n = wv + i * NWV;
        if (n < npd) cnt_w = i + 1;
        int k, col, st;
        if (n < A.npa) {                                              // frames 2n, 2n+1: E (columns 4j, 4j+1), O (4j+2, 4j+3)
            k = min(2 * n + (lane >> 5), K - 1);
            const int j = lane & 31;
            col = fc0 + (j < 16 ? 4 * j : 4 * (j - 16) + 2);
            st = 0;
        } else {
            const int qq = min((n - A.npa) * 64 + lane, K * C::BPF - 1);
            k = qq / C::BPF;
            const int j = qq - k * C::BPF;
            col = fc0 - (ms - 1) + (j < C::NBE ? 4 * j : 4 * (j - C::NBE) + 2);
            st = 1;
        }
        const int gc = min(max(col, A.bc0), A.bc1 - 1);
        src_off[i] = gp(A.frame_off)[st * K + k] + (unsigned)gc * 8u;
    }
    const bool dma = !(A.ablate & 1);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A.baseA, (short)0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)A.baseB, (short)0, -1, 0x00020000);
    // stage A patch row ya into slot_a (if doA) and B patch row rho (counted from the first row offset of the pass) into slot_b
    auto issue_rows = [&](int ya, int rho, int slot_a, int slot_b, bool doA) {
        if (!dma) return;
        const int gra = min(max(fr0 + ya, A.br0), A.br1), grb = min(max(fr0 + oilo + rho, A.br0), A.br1);
        const unsigned soA = (unsigned)gra * (unsigned)A.Wf * 8u, soB = (unsigned)grb * (unsigned)A.Wf * 8u;
        const unsigned dA = ldsA + (unsigned)slot_a * A.a_slot, dB = ldsB + (unsigned)slot_b * A.b_slot;
#pragma unroll
        for (int i = 0; i < NPT; i++) {
            if (i < cnt_w) {
                const int n = wv + i * NWV;
                if (n < A.npa) {
                    if (doA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (UMPA_LDS_AS void*)(lds + dA + (unsigned)n * 1024u), 16, (int)src_off[i], (int)soA, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (UMPA_LDS_AS void*)(lds + dB + (unsigned)(n - A.npa) * 1024u), 16, (int)src_off[i], (int)soB, 0, 0);
                }
            }
        }
    };
    // prologue: B rows 0 .. nuy-2 alone, then the rows of the first LA steps
    // (ring depths da = LA + 1, db = nuy + LA: the rows staged at step y go into the slots step y - 1 has just given up)
    for (int i = 0; i + 1 < A.nuy; i++) issue_rows(0, i, 0, i, false);
#pragma unroll
    for (int s = 0; s < (SPB > 1 ? SPB : LA); s++) issue_rows(s, s + A.nuy - 1, s, s + A.nuy - 1, true);

    // ---- DPP row r -> plane: the first nuy * (UJ - 1) rows are (row offset, column offsets 0 .. UJ-2), pairs (2m, 2m+1) on rows
    // (2i, 2i+1); then the last column offset (even) of every row offset.  Rows past the pass idle.
    const int l = tid & 15, rw = tid >> 4;
    const int nfirst = A.nuy * (UJ - 1);
    const int uyl = rw < nfirst ? rw / (UJ - 1) : min(rw - nfirst, A.nuy - 1);
    const int ux = rw < nfirst ? rw - uyl * (UJ - 1) : UJ - 1;
    const int oi = oilo + uyl, oj = ux - (ms - 1);
    const bool plane_ok = rw < A.nuy * UJ && oi <= ms - 1;
    const bool odd = ux & 1;                                          // works on columns 4l-1 .. 4l+2: A', result moved back by one column
    const int mh = ux >> 1;
    // B pieces (2l + mh) and (2l + mh + 1): E[j] holds piece 2j, O[j] piece 2j + 1
    const int bp1 = (mh & 1) ? C::NBE + l + (mh - 1) / 2 : l + mh / 2;
    const int bp2 = (mh & 1) ? l + (mh + 1) / 2 : C::NBE + l + mh / 2;
    // A columns of this lane, one ds_read_b64 each out of [E: 16 pairs | O: 16 pairs] (256 bytes each): an even plane reads
    // (E.lo, E.hi, O.lo, O.hi)[l], an odd one (O[l-1].hi, E[l].lo, E[l].hi, O[l].lo) -- in every instruction the two planes of a
    // pair sit on complementary halves of the 16-byte slots, i.e. on different banks
    // (three address registers: columns 1 and 3 differ by the same 8 bytes between the two kinds of plane)
    const unsigned a_o0 = odd ? (l ? 256u + (l - 1) * 16u + 8u : 0u) : l * 16u;
    const int a_o13 = odd ? (int)l * 16 - 8 : (int)l * 16;            // column 1 at +8, column 3 at +264
    const unsigned a_o2 = odd ? l * 16u + 8u : 256u + l * 16u;
    const unsigned b_off1 = bp1 * 16, b_off2 = bp2 * 16;

    // ring[t]: the sum of the output row that started t steps ago, taps 0 .. t done (t = 0 .. 2 Nw - 1)
    double ring[S - 1][4];
#pragma unroll
    for (int j = 0; j < S - 1; j++)
#pragma unroll
        for (int c = 0; c < 4; c++) ring[j][c] = 0.0;

    int ia = 0, ib = 0;                                               // y mod da, y mod db (scalars)
    const int UU = UJ * UJ;
    const int ui = A.sigma * oi, uj = A.sigma * oj;
    const int slot = plane_ok ? (ui + ms - 1) * UJ + (uj + ms - 1) : 0;
    // table block of this strip: [row][shift][tw]; a scalar base per row + this lane's 32-bit offset (global_store with an SGPR base)
    UMPA_GLOBAL char* const tbase = (UMPA_GLOBAL char*)(gpw(A.table) + ((size_t)strip * A.rows + r_lo) * UU * A.tw);
    const unsigned lane_off = (unsigned)(slot * A.tw + 4 * l) * 8u;
    const size_t row_pitch = (size_t)UU * A.tw * 8;
    const bool st_ok = plane_ok && 4 * l < A.tw && !(A.ablate & 8);
    // the table stores of this wave per step that completes a row: they count in vmcnt like the LDS-DMA
    const int st_w = __builtin_amdgcn_ballot_w64(st_ok) != 0 ? 2 : 0;

    // (Tried with SPB = 2 and not kept: the upper half of the waves one phase behind the lower half -- filters and stores of step
    //  y - 1 before the products of step y -- so that one wave's LDS-bound frame loop would meet another's VALU-bound filters:
    //  C3 23.7 -> 25.1 ms, C2 1.43 -> 1.54.  SPB = 2 itself: C3 24.1 -> 23.7 ms for 24 KB more LDS; the library runs SPB = 1.)
    double p[4] = {0.0, 0.0, 0.0, 0.0};
    // the row filter takes the products of step ys; the row that completes goes through the column filter and out
    auto filter_and_store = [&](int ys) {
            if (A.ablate & 4) {
#pragma unroll
                for (int c = 0; c < 4; c++) ring[0][c] += p[c];
                return;
            }
            // ---- row filter: out[r] = sum_t hr[t] P[r + t].  The row that started 2 Nw steps ago completes; every other sum moves
            // one place up the ring as it takes its next tap (v_fma with the destination beside the addend: no copies)
            double v[4 + 4 * C::NNB];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                v[c] = fma(sep.hr[S - 1], p[c], ring[S - 2][c]);
#pragma unroll
                for (int t = S - 2; t >= 1; t--) ring[t][c] = fma(sep.hr[t], p[c], ring[t - 1][c]);
                ring[0][c] = sep.hr[0] * p[c];
            }
            if (ys < 2 * NW) return;                                      // (wave-uniform) nothing completes yet
            // ---- column filter of the completed row: the 2 Nw values to the right come from lanes l+1 .. l+NNB
#pragma unroll
            for (int c = 0; c < 4; c++) {
                v[4 + c] = dpp_row_shl<1>(v[c]);
                if (C::NNB >= 2 && 8 + c < 4 + 2 * NW) v[8 + c] = dpp_row_shl<2>(v[c]);
                if (C::NNB >= 3 && 12 + c < 4 + 2 * NW) v[12 + c] = dpp_row_shl<3>(v[c]);
                if (C::NNB >= 4 && 16 + c < 4 + 2 * NW) v[16 + c] = dpp_row_shl<4>(v[c]);
            }
            double o[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double acc = sep.hc[0] * v[c];
#pragma unroll
                for (int t = 1; t < S; t++) acc = fma(sep.hc[t], v[c + t], acc);
                o[c] = acc;
            }
            // a plane at an odd shift holds columns 4l-1 .. 4l+2: its column 4l+3 is the next lane's first value
            {
                const double nx = dpp_row_shl<1>(o[0]);
                const double o0 = odd ? o[1] : o[0], o1 = odd ? o[2] : o[1], o2 = odd ? o[3] : o[2], o3 = odd ? nx : o[3];
                o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
            }
            // ---- table row r_lo + y - 2 Nw
            // (plain stores: each instruction writes half of every 64 bytes, the L2 puts the lines together; non-temporal stores of
            //  this shape go out as partial lines -- tools/microbench/table_store_rate.hip: 2.5 against 4.2-6.8 TB/s)
            if (st_ok) {
                UMPA_GLOBAL char* dst = tbase + (size_t)(ys - 2 * NW) * row_pitch;     // (wave-uniform)
                pair_t v0, v1;
                v0[0] = o[0]; v0[1] = o[1]; v1[0] = o[2]; v1[1] = o[3];
                *reinterpret_cast<UMPA_GLOBAL table_pair_t*>(dst + lane_off) = v0;
                *reinterpret_cast<UMPA_GLOBAL table_pair_t*>(dst + lane_off + 16) = v1;
            }
    };
    for (int y = 0; y < nsteps; y++) {
        // the rows of step y were issued LA steps ago; since then this wave has issued the rows of LA - 1 further steps (none
        // for steps past the end) and the table stores of those of the steps y - LA .. y - 1 that completed a row
        if constexpr (SPB == 1) {
            const int later = min(LA - 1, max(0, nsteps - 1 - y));
            wait_vmcnt_rt(later * cnt_w + st_w * max(0, min(LA, y - 2 * NW)));
            lds_barrier();                                            // everyone's pieces of step y are in; step y - 1 has been read
            const int ja = ia == 0 ? A.da - 1 : ia - 1, jb = ib == 0 ? A.db - 1 : ib - 1;
            if (y + LA < nsteps) issue_rows(y + LA, y + LA + A.nuy - 1, ja, jb, true);
        } else if (y % SPB == 0) {
            // head of an interval: its rows were issued at the head of the one before; since then this wave has only issued the
            // table stores of that interval's steps that completed a row
            wait_vmcnt_rt(st_w * max(0, min(SPB, y - 2 * NW)));
            lds_barrier();                                            // the interval's rows are in; the previous interval has been read
#pragma unroll
            for (int q = 0; q < SPB; q++) {
                const int yn = y + SPB + q;
                if (yn < nsteps) {
                    int ja = ia + SPB + q; ja -= ja >= A.da ? A.da : 0;
                    int jb = ib + SPB + q + A.nuy - 1; jb -= jb >= A.db ? A.db : 0; jb -= jb >= A.db ? A.db : 0;
                    issue_rows(yn, yn + A.nuy - 1, ja, jb, true);
                }
            }
        }
        // ---- products of patch row y, summed over the frames
        p[0] = p[1] = p[2] = p[3] = 0.0;
        if (!(A.ablate & 2)) {
            int ibl = ib + uyl;
            if (ibl >= A.db) ibl -= A.db;
            UMPA_LDS_AS const char* pa = lds + ldsA + (unsigned)ia * A.a_slot;
            UMPA_LDS_AS const char* pb = lds + ldsB + (unsigned)ibl * A.b_slot;
            unsigned q0 = (unsigned)(size_t)pa + a_o0, q1 = (unsigned)((int)(size_t)pa + a_o13), q2 = (unsigned)(size_t)pa + a_o2;   // LDS byte addresses
            unsigned r1 = (unsigned)(size_t)pb + b_off1, r2 = (unsigned)(size_t)pb + b_off2;
            // One frame: six LDS reads in the order they are used -- columns 0, 1 of A, their B pair, columns 2, 3, their B pair --
            // and two counted waits, so that the first two FMAs run while the second half is still on its way.  (Everything is
            // written out: left to itself the compiler pairs the A reads into ds_read2_b64 / ds_read2st64_b64, which run at half
            // the LDS rate -- C2: frame loop 0.7 -> 1.5 ms.)
            auto frame = [&](auto FA, auto FB) {
                constexpr int fa = decltype(FA)::value, fb = decltype(FB)::value;
                double a0, a1, a2, a3;
                pair_t b0, b1;
                asm volatile("ds_read_b64 %0, %6 offset:%11\n\tds_read_b64 %1, %7 offset:%12\n\tds_read_b128 %4, %9 offset:%14\n\t"
                             "ds_read_b64 %2, %8 offset:%11\n\tds_read_b64 %3, %7 offset:%13\n\tds_read_b128 %5, %10 offset:%14"
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1)
                             : "v"(q0), "v"(q1), "v"(q2), "v"(r1), "v"(r2), "n"(fa), "n"(fa + 8), "n"(fa + 264), "n"(fb) : "memory");
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a0), "+v"(a1), "+v"(b0) :: "memory");
                p[0] = fma(a0, b0[0], p[0]); p[1] = fma(a1, b0[1], p[1]);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a2), "+v"(a3), "+v"(b1) :: "memory");
                p[2] = fma(a2, b1[0], p[2]); p[3] = fma(a3, b1[1], p[3]);
            };
            using I0 = std::integral_constant<int, 0>;
            int k = 0;
            for (; k + 4 <= K; k += 4) {                              // (frame offsets as immediates: one address update per four frames)
                frame(I0{}, I0{});
                frame(std::integral_constant<int, C::APF * 16>{}, std::integral_constant<int, C::BPF * 16>{});
                frame(std::integral_constant<int, 2 * C::APF * 16>{}, std::integral_constant<int, 2 * C::BPF * 16>{});
                frame(std::integral_constant<int, 3 * C::APF * 16>{}, std::integral_constant<int, 3 * C::BPF * 16>{});
                q0 += 4 * C::APF * 16; q1 += 4 * C::APF * 16; q2 += 4 * C::APF * 16;
                r1 += 4 * C::BPF * 16; r2 += 4 * C::BPF * 16;
            }
            for (; k < K; k++) {
                frame(I0{}, I0{});
                q0 += C::APF * 16; q1 += C::APF * 16; q2 += C::APF * 16;
                r1 += C::BPF * 16; r2 += C::BPF * 16;
            }
        }
        ia = ia + 1 == A.da ? 0 : ia + 1;
        ib = ib + 1 == A.db ? 0 : ib + 1;
        filter_and_store(y);
    }
    wait_vmcnt<0>();
    if (A.done && tid == 0) {
        atomicOr(A.done + item, 1ull << pass);
        if (A.ndone) atomicAdd(A.ndone, 1);
    }
    if (A.ablate & 4) {
        if (plane_ok && tid == 0) *(UMPA_GLOBAL double*)tbase = ring[0][0] + ring[0][1] + ring[0][2] + ring[0][3];
    }
}

} // namespace umpa

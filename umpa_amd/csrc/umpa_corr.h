// umpa_corr.h -- corr_volume, the dominant kernel of the tiled path: the exhaustive table
//   t5[u][p] = sum_k W[ s_k(.) r_k(.+u) ](p)          (the only term of Model.cpp:763-772 that couples
// pixel and shift spatially) for all (2 max_shift - 1)^2 integer shifts u of a row chunk.
//
// One workgroup = one (tile, pass): a tile is TR x TC output pixels, a pass is UI consecutive row offsets and
// a batch of UB column offsets.  Per frame k the workgroup needs the q-region (tile + window halo) of the
// fixed-window stack A and the same region, widened by the offsets of the pass, of the moving stack B.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): the frames go from L2 straight into LDS, no VGPRs, no
// ds_write.  A wave-instruction writes 64 consecutive 16-byte pieces, the SOURCE address is per lane, so
// the LDS image is a plain row-major patch [row][column pair] filled in piece order, in a ring of NSLOT
// frame slots with counted s_waitcnt vmcnt (no drain in the loop).  Two-slot rings (the 2-workgroups-per-CU
// shapes): frame k+2 is issued into the slot of frame k as soon as everyone has multiplied frame k, so two
// frames travel at a time; deeper rings: frame k+NSLOT-1 is issued at the head of step k, its DMA
// instructions spread between the product rows.  Product threads own one patch row and QB consecutive
// columns and read them as column pairs (ds_read_b128): with an odd number of pieces per image row the 16
// lanes of a b128 group (consecutive rows) hit 16 different 16-byte slots -- conflict free.
//
// After the last frame the product planes go to LDS in NF rounds (transposed, [column][row], odd row pitch,
// plane pitch = QR mod 32), get the window's column filter in place and its row filter -- one column per
// lane, adjacent lanes swapping rows by DPP -- on the way to the table (16-byte non-temporal stores).
#pragma once
#include "umpa_direct.h"
#include "umpa_ondemand.h"
#include <type_traits>

namespace umpa {

#define UMPA_TILE 32
#define UMPA_MAX_NW 8
#define UMPA_LDS_BUDGET (160 * 1024)

struct Sep1D {                       // the two 1-D factors of the window, win[a][b] = hr[a]*hc[b]
    double hr[2 * UMPA_MAX_NW + 1];
    double hc[2 * UMPA_MAX_NW + 1];
};

// Reads in[t*stride], t < CB + 2NW, and returns the CB filtered values out[o] = sum_tap h[tap] in[o+tap].
// All inputs are fetched before the first FMA so the LDS latency is paid once per item, not once per read.
template <int NW, int CB>
__device__ __forceinline__ void fir_block(const double* __restrict__ in, int stride, const double* h, double* out)
{
    constexpr int S = 2 * NW + 1;
    double v[CB + S - 1];
#pragma unroll
    for (int t = 0; t < CB + S - 1; t++) v[t] = in[t * stride];
#pragma unroll
    for (int o = 0; o < CB; o++) {
        double acc = 0.0;
#pragma unroll
        for (int tap = 0; tap < S; tap++) acc = fma(h[tap], v[o + tap], acc);
        out[o] = acc;
    }
}

// The same sums in the same order, one input at a time: 8 running outputs instead of CB + 2NW inputs in registers.
template <int NW, int CB>
__device__ __forceinline__ void fir_running(const double* __restrict__ in, int stride, const double* h, double* out)
{
    constexpr int S = 2 * NW + 1;
#pragma unroll
    for (int o = 0; o < CB; o++) out[o] = 0.0;
#pragma unroll
    for (int t = 0; t < CB + S - 1; t++) {
        const double v = in[t * stride];
#pragma unroll
        for (int o = 0; o < CB; o++)
            if (t - o >= 0 && t - o < S) out[o] = fma(h[t - o], v, out[o]);
    }
}

struct CorrArgs {
    double* table;            // [(2ms-1)^2][rows][N1]
    size_t slot_stride;       // rows * N1
    int org0, org1;           // frame coordinates of output pixel (0,0) of the REGION
    int row0, rows;           // this launch covers region rows [row0, row0+rows)
    int N1;
    int pitch;                // doubles per table row: N1 rounded up to whole 256-byte tile rows (every tile's store instructions then
                              // write whole, aligned 128-byte lines: tools/microbench/table_store_rate.hip), columns N1 .. pitch-1 padding
    int sigma;                // +1: B = ref sits at p+u ('sam' mode); -1: B = sam sits at p-u ('ref' mode)
    int ntx, nty;
    int br0, br1, bc0, bc1, Wf;   // rows / columns of the image inside every frame, the frames' common width (Maps, umpa_tiled.h)
    int ablate;               // diagnostics only (UMPA_HIP_ABLATE): 1 no global loads, 4 no products, 8 no filters, 16 no table stores
};

#define UMPA_LDS_AS __attribute__((address_space(3)))
// 1: on two-slot rings the next frame is issued right after the products (see the frame loop); 2: on every ring
// (measured: deeper rings lose the DMA/FMA interleave and pay the second barrier, C3 37.1 -> 42-44 ms)
#ifndef UMPA_CORR_EARLY_ISSUE
#define UMPA_CORR_EARLY_ISSUE 1
#endif

// Tile = TR x TC output pixels (TR = 32 rows).  A workgroup is UI groups of NTG threads: group g works on row
// offset oi0 + g, all groups share the staged frames (the A image, and a B image of QR + UI - 1 rows), so a pass
// produces UI * UB planes from one staging.  WPC workgroups share a CU, so one may use LDSB = 160 KiB / WPC of
// LDS; the planes are filtered and stored in NF rounds of PR planes (fewer planes in LDS at a time).
// RO > 1: every product thread accumulates RO consecutive row offsets (all reading ONE staged B image of QR + RO - 1 rows):
// a pass produces RO * UB planes from one staging, a third of the L2 -> LDS traffic per plane at RO = 3.  The accumulators
// then fill the register file (4 x 9 x 3 doubles per thread on 32x32 tiles), so the flush goes in rounds of few planes with
// the column filter writing to a second LDS area (nothing is held in registers across a barrier).
template <int NW, int UB, int TC, int NTG, int UI, int WPC, int NF, int RO = 1>
struct CorrCfg {
    static constexpr int NT = NTG * UI;
    static constexpr int NROW = UI * RO;                  // row offsets per pass
    static constexpr int TR = UMPA_TILE, S = 2 * NW + 1;
    static constexpr int LDSB = (UMPA_LDS_BUDGET / WPC) & ~15;
    static constexpr int NPL = NROW * UB;                 // planes per pass
    static constexpr int PR = (NPL + NF - 1) / NF;        // planes per flush round
    static constexpr int WPS = (NT / 64 * WPC + 3) / 4;   // waves per SIMD this shape is built for
    static constexpr int QR = TR + 2 * NW, QC = TC + 2 * NW;  // q-region (tile + window halo): rows, columns
    static constexpr int QRB = QR + NROW - 1;             // rows of the B image
    static constexpr int QP = QR | 1;                     // odd pitch of the transposed product planes ([column][row])
    // one product plane; padded so that PPL = QR (mod 32): the column filter's lanes run over the rows of one
    // (plane, column block) after the other, plane fastest, and with this pitch the LDS bank sequence continues
    // across the seam between two runs (a 32-lane group never meets a bank twice)
    static constexpr int PPL = QC * QP + ((QR - QC * QP) % 32 + 32) % 32;
    // product blocking: a thread owns one q row and QB consecutive columns (QB even: column pairs)
    static constexpr int QB = (QR * ((QC + 3) / 4) <= NTG) ? 4 : (QR * ((QC + 5) / 6) <= NTG) ? 6 : 8;
    static constexpr int NQB = (QC + QB - 1) / QB;        // column blocks
    static constexpr int NBP = (QB + UB) / 2;             // B column pairs a thread reads: ceil((QB+UB-1)/2)
    static constexpr int BW = (QC + UB) & ~1;             // B columns staged, even
    // 16-byte pieces per image row: odd (16 consecutive rows of one column block then sit in 16 different 16-byte slots), and
    // -- where the window allows it -- QR * pitch = QB / 2 (mod 16): the lanes of a ds_read_b128 group that straddle two column
    // blocks (rows wrap from QR - 1 to 0, the piece index steps by QB / 2) then continue the slot sequence.  Measured
    // (tools/microbench/lds_b128_pitch.hip): 42 rows at pitch 21: no conflict cycles, at pitch 25: 40 % of the LDS-active cycles.
    static constexpr int pitch_for(int least)
    {
        for (int p = least | 1; p < (least | 1) + 16; p += 2)
            if ((QR * p) % 16 == (QB / 2) % 16) return p;
        return least | 1;                                 // (even half-window: no such pitch)
    }
#ifndef UMPA_CORR_PB_MIN
#define UMPA_CORR_PB_MIN 1
#endif
    // Round 4: the three-row-offset shapes stage the B image at its LEAST odd pitch (25 pieces instead of pitch_for's 29 on C2):
    // 8 % fewer staged bytes and four LDS-DMA instructions per thread and frame instead of five.  The kernel is bound by the
    // bytes through the CU's vector-memory path (DESIGN.md 4.3 "round 4"), not by the LDS bank conflicts the wider pitch avoids:
    // corr_volume 1.277 / 1.282 -> 1.254 / 1.264 ms on one box, back to back (tools/ab_libs.sh; -DUMPA_CORR_PB_MIN=0 restores 29).
    static constexpr int PA = pitch_for(QC / 2), PB = (UMPA_CORR_PB_MIN && RO == 3) ? ((BW / 2) | 1) : pitch_for(BW / 2);
    // pieces of one staged frame: A image, then B image from a multiple of 64 pieces on -- a wave-instruction (64 pieces)
    // then reads one stack only, and its base address is a scalar (SGPR base + 32-bit lane offset)
    static constexpr int APIECES = (QR * PA + 63) & ~63;
    static constexpr int NPIECE = APIECES + QRB * PB;
    static constexpr int NPT = (NPIECE + NT - 1) / NT;    // LDS-DMA instructions per thread and frame
    static constexpr int SLOT = NPT * NT * 2;             // doubles per frame slot (every lane of every instruction writes)
    // the last block's threads read past the end of their image row, on the last row past the image:
    // (every thread multiplies, also those past the last column block -- no divergence in the frame loop; what they
    // read is never stored)
    static constexpr int NQBX = (NTG + QR - 1) / QR;
    static constexpr int OVER = 2 * (NQBX * (QB / 2) - (QB / 2) + NBP) - 2 * PB;
    static constexpr int TAIL = NPIECE * 2 + (OVER > 0 ? OVER : 0);             // doubles of the last slot that are touched
    static constexpr int NSLOT_FIT = (LDSB / 8 - (TAIL > SLOT ? TAIL - SLOT : 0)) / SLOT;
    static constexpr int NSLOT = NSLOT_FIT > 4 ? 4 : NSLOT_FIT;               // ring depth: NSLOT-1 frames in flight
    static constexpr int RING = NSLOT * SLOT + (TAIL > SLOT ? TAIL - SLOT : 0);
    // RO > 1: the column-filtered planes live beside the raw ones, [column][row] with the even row pitch HQP = 2 (mod 4): the
    // row filter's lanes (one column each) read 16-byte aligned row pairs (ds_read_b128), 16 lanes of a group in 16
    // different 16-byte slots of the 256-byte bank row
    static constexpr int HQP = QR + ((6 - QR % 4) % 4);
    // plane pitch = QR (mod 32), as PPL: the column filter's lanes run over the rows of one (plane, column block) after the
    // other and write their results at this pitch -- the bank sequence continues across the seam (32x32 tiles at window 11:
    // TC * HQP = 0 mod 32 put the lanes after the seam on the banks of the lanes before it)
    static constexpr int HPL = TC * HQP + ((QR - TC * HQP) % 32 + 32) % 32;
    static constexpr int FLUSH = PR * PPL + (RO > 1 ? PR * HPL : 0);
    static constexpr int LDS_DOUBLES = RING > FLUSH ? RING : FLUSH;
    static constexpr size_t LDS = (size_t)LDS_DOUBLES * sizeof(double);
    static constexpr int CB = 8;                          // outputs per filter item
    static constexpr int HITEMS = PR * (TC / CB) * QR, HROUNDS = (HITEMS + NT - 1) / NT;
    static constexpr bool OK = NSLOT >= 2 && FLUSH * 8 <= LDSB && QR * NQB <= NTG && (NSLOT - 1) * NPT < 64 &&
                               LDS_DOUBLES * 8 <= LDSB && NTG % 64 == 0 && NT <= 1024 && WPS <= 8 && (RO == 1 || UI == 1) &&
                               QB * UB * RO <= (QB == 4 ? 112 : 100);
};

// 16-byte store of two table entries, non-temporal (`nt`): the 2.7 GB streaming out no longer push the frame patches
// the other passes of the tile are about to re-read out of L2 (C2 1.94 -> 1.84 ms; an `sc1` store made no difference)
typedef double table_pair_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_table16(UMPA_GLOBAL double* dst, table_pair_t v)
{
    __builtin_nontemporal_store(v, reinterpret_cast<UMPA_GLOBAL table_pair_t*>(dst));
}

// the value of the lane whose number differs in bit 0 (DPP quad_perm [1,0,3,2]: no LDS crossbar, unlike __shfl_xor)
__device__ __forceinline__ double swap_adjacent_lanes(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// workgroup barrier that does NOT drain the vector-memory counter (LDS-DMA of later frames stays in flight);
// the compiler must not move LDS accesses across it
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One LDS-DMA wave-instruction: lane l copies the 16 bytes at base + off (its own offset) to dst + 16 l.  `base` is
// wave-uniform: buffer addressing (SGPR descriptor + 32-bit lane offset) needs no 64-bit address per lane.
__device__ __forceinline__ void lds_dma16(const UMPA_GLOBAL char* base, unsigned off, UMPA_LDS_AS char* dst)
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, -1, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (UMPA_LDS_AS void*)dst, 16, (int)off, 0, 0, 0);
}

// the planes of one (tile, pass) into the table
template <int NW, int UB, int TC, int NTG, int UI, int WPC, int NF, int RO>
__device__ __forceinline__ void corr_volume_tile(const ModelDev& m, const CorrArgs& A, const Sep1D& sep, double* lds,
                                                 const int lin, const int pass, const int tid)
{
    using C = CorrCfg<NW, UB, TC, NTG, UI, WPC, NF, RO>;
    constexpr int NT = C::NT;
    const int ms = m.ms, UJ = 2 * ms - 1;
    const int nbatch = (UJ + UB - 1) / UB;
  do {
    const int tx = lin % A.ntx, ty = lin / A.ntx;
    const int grp = UI > 1 ? __builtin_amdgcn_readfirstlane(tid / NTG) : 0, gtid = tid - grp * NTG;
    const int prow0 = A.row0 + ty * C::TR, pcol0 = tx * TC;           // first output pixel of the tile (region coords)
    const int fr0 = A.org0 + prow0 - NW, fc0 = A.org1 + pcol0 - NW;   // frame coords of the q-region origin
    const int oi0 = (pass / nbatch) * C::NROW - (ms - 1), oj0 = (pass % nbatch) * UB - (ms - 1);

    // ---- LDS-DMA slots of this thread: piece p = tid + n*NT of the frame image (A rows, then B rows); its source
    // is 16 bytes = two adjacent columns of one frame row.  Pieces past the image (padding of the last instruction,
    // the pad piece of an image row) read a clamped address and land in bytes nobody uses.
    unsigned src_off[C::NPT];                                         // byte offset inside a frame
#pragma unroll
    for (int n = 0; n < C::NPT; n++) {
        const int p = tid + n * NT;
        int r, c;
        if (p < C::APIECES) { r = p / C::PA; c = 2 * (p % C::PA); }
        else {
            const int q = min(p - C::APIECES, C::QRB * C::PB - 1);
            r = q / C::PB + oi0; c = 2 * (q % C::PB) + oj0;
        }
        const int gr = min(max(fr0 + r, A.br0), A.br1), gc = min(max(fc0 + c, A.bc0), A.bc1 - 1);
        src_off[n] = (unsigned)(gr * A.Wf + gc) * 8u;
    }
    const unsigned wave_piece0 = (unsigned)__builtin_amdgcn_readfirstlane(tid & ~63);   // first piece of this wave's instruction 0
    unsigned src_isB = 0;                                             // bit n: instruction n of this wave reads the B stack (wave-uniform)
#pragma unroll
    for (int n = 0; n < C::NPT; n++)
        if (wave_piece0 + n * NT >= (unsigned)C::APIECES) src_isB |= 1u << n;

    auto issue_frame = [&](int k) {                                   // frame k -> ring slot k % NSLOT
        const FrameDesc fd = load_frame(m.frames, k);
        const long shift = ((long)fd.pi * A.Wf + fd.pj) * 8;          // image coordinates -> this frame's array, bytes
        const UMPA_GLOBAL char* gA = (const UMPA_GLOBAL char*)gp(A.sigma > 0 ? fd.sam : fd.ref) - shift;
        const UMPA_GLOBAL char* gB = (const UMPA_GLOBAL char*)gp(A.sigma > 0 ? fd.ref : fd.sam) - shift;
        UMPA_LDS_AS char* slot = (UMPA_LDS_AS char*)(lds + (k % C::NSLOT) * C::SLOT);
#pragma unroll
        for (int n = 0; n < C::NPT; n++)
            lds_dma16((src_isB >> n) & 1u ? gB : gA, src_off[n], slot + (size_t)(wave_piece0 + n * NT) * 16);
    };

    // product-stage ownership inside the group: (qb, r), r fastest; a group whose row offset is past the search
    // range (the last pass when UI does not divide 2 ms - 1) only helps with the staging and the filters
    const int pr = gtid % C::QR, pqb = gtid / C::QR;
    const bool gvalid = oi0 + grp * RO <= ms - 1;
    const bool pactive = pqb < C::NQB && gvalid;                       // owns accumulators that reach the table
    typedef double pair_t __attribute__((ext_vector_type(2)));
    double acc[RO][C::QB][UB];
#pragma unroll
    for (int ro = 0; ro < RO; ro++)
#pragma unroll
        for (int t = 0; t < C::QB; t++)
#pragma unroll
            for (int u = 0; u < UB; u++) acc[ro][t][u] = 0.0;

    const int K = m.Na;
    constexpr int D = C::NSLOT - 1;                                   // frames in flight behind the one in use
    if (!(A.ablate & 1)) {
#pragma unroll
        for (int f = 0; f < D; f++)
            if (f < K) issue_frame(f);
    }
    // One frame step.  ISSUE: frame k+D exists and its DMA instructions go out BETWEEN the product rows of frame k, one
    // per QB-th of the FMAs (straight-line code, pinned with sched_barrier): a wave that issues them all at once sits in
    // the issue queue of the memory pipeline instead of multiplying.
    const bool dma = !(A.ablate & 1), prod = gvalid && !(A.ablate & 4);     // (wave-uniform)
    auto frame_step = [&](int k, auto ISSUE, auto DRAIN) {
        // frames k .. k+D-1 have been issued, in order: frame k has landed when at most (D-1)*NPT are outstanding
        if constexpr (decltype(DRAIN)::value) wait_vmcnt<0>(); else wait_vmcnt<(D - 1) * C::NPT>();
        lds_barrier();                                                // everyone's pieces of frame k are in; slot (k-1)%NSLOT is free
        constexpr bool issue = decltype(ISSUE)::value;
        const FrameDesc fdn = load_frame(m.frames, issue ? k + D : k);
        const long shiftn = ((long)fdn.pi * A.Wf + fdn.pj) * 8;
        const UMPA_GLOBAL char* gAn = (const UMPA_GLOBAL char*)gp(A.sigma > 0 ? fdn.sam : fdn.ref) - shiftn;
        const UMPA_GLOBAL char* gBn = (const UMPA_GLOBAL char*)gp(A.sigma > 0 ? fdn.ref : fdn.sam) - shiftn;
        UMPA_LDS_AS char* slotn = (UMPA_LDS_AS char*)(lds + ((k + D) % C::NSLOT) * C::SLOT);
        auto issue_piece = [&](int n) {
            lds_dma16((src_isB >> n) & 1u ? gBn : gAn, src_off[n], slotn + (size_t)(wave_piece0 + n * NT) * 16);
        };
        constexpr int PER = (C::NPT + C::QB - 1) / C::QB;            // DMA instructions per product row
        if (prod) {
            const pair_t* img = reinterpret_cast<const pair_t*>(lds + (k % C::NSLOT) * C::SLOT);
            const pair_t* la = img + pr * C::PA + pqb * (C::QB / 2);
            const pair_t* lb = img + C::APIECES + (pr + grp * RO) * C::PB + pqb * (C::QB / 2);
            pair_t av[C::QB / 2], bv0[C::NBP];
#pragma unroll
            for (int t = 0; t < C::QB / 2; t++) av[t] = la[t];
            if constexpr (RO > 1) {
                // the accumulators fill the register file: the B row comes in two halves, each with the FMAs that need
                // nothing else (column sums t + u below / from HB), so that only half a row has to be in registers
                constexpr int HP = C::NBP / 2, HB = 2 * HP;           // pairs / B columns of the first half
#pragma unroll
                for (int ro = 0; ro < RO; ro++) {
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        constexpr int SL = 2 * RO;                    // slots the DMA instructions are spread over
                        if constexpr (issue) {
#pragma unroll
                            for (int n = 0; n < C::NPT; n++)
                                if (n % SL == ro * 2 + half) issue_piece(n);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        pair_t bv[C::NBP];
#pragma unroll
                        for (int t = 0; t < C::NBP; t++)
                            if ((t < HP) == (half == 0)) bv[t] = lb[ro * C::PB + t];
#pragma unroll
                        for (int t = 0; t < C::QB; t++)
#pragma unroll
                            for (int u = 0; u < UB; u++)
                                if ((t + u < HB) == (half == 0))
                                    acc[ro][t][u] = fma(av[t >> 1][t & 1], bv[(t + u) >> 1][(t + u) & 1], acc[ro][t][u]);
                        if constexpr (issue) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < C::NBP; t++) bv0[t] = lb[t];
#pragma unroll
                for (int t = 0; t < C::QB; t++) {
                    if constexpr (issue) {
#pragma unroll
                        for (int q = 0; q < PER; q++)
                            if (t * PER + q < C::NPT) issue_piece(t * PER + q);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int u = 0; u < UB; u++) acc[0][t][u] = fma(av[t >> 1][t & 1], bv0[(t + u) >> 1][(t + u) & 1], acc[0][t][u]);
                    if constexpr (issue) __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else if constexpr (issue) {
#pragma unroll
            for (int n = 0; n < C::NPT; n++) issue_piece(n);
        }
    };
    if constexpr (UMPA_CORR_EARLY_ISSUE >= (C::NSLOT == 2 ? 1 : 2)) {
        // Frame k+NSLOT goes into the slot of frame k as soon as everyone has multiplied frame k -- a second barrier per
        // frame, but its DMA then runs beside those of the frames before it instead of starting only when frame k+1 has
        // landed (NSLOT frames in flight out of NSLOT slots; with the issue at the head of the step, NSLOT-1).
        if (dma) {
#pragma unroll
            for (int f = D; f < C::NSLOT; f++)
                if (f < K) issue_frame(f);                            // frames 0 .. D-1 went out above
        }
        auto products = [&](int k) {
            if (!prod) return;
            const pair_t* img = reinterpret_cast<const pair_t*>(lds + (k % C::NSLOT) * C::SLOT);
            const pair_t* la = img + pr * C::PA + pqb * (C::QB / 2);
            const pair_t* lb = img + C::APIECES + (pr + grp * RO) * C::PB + pqb * (C::QB / 2);
            pair_t av[C::QB / 2];
#pragma unroll
            for (int t = 0; t < C::QB / 2; t++) av[t] = la[t];
#pragma unroll
            for (int ro = 0; ro < RO; ro++) {
                pair_t bv[C::NBP];
#pragma unroll
                for (int t = 0; t < C::NBP; t++) bv[t] = lb[ro * C::PB + t];
#pragma unroll
                for (int t = 0; t < C::QB; t++)
#pragma unroll
                    for (int u = 0; u < UB; u++) acc[ro][t][u] = fma(av[t >> 1][t & 1], bv[(t + u) >> 1][(t + u) & 1], acc[ro][t][u]);
            }
        };
        for (int k = 0; k < K; k++) {
            // frames k+1 .. min(k+NSLOT-1, K-1) may still be on their way
            const int behind = dma ? min(C::NSLOT - 1, K - 1 - k) : 0;
            if (behind >= 3 && C::NSLOT >= 4) wait_vmcnt<3 * C::NPT>();
            else if (behind == 2 && C::NSLOT >= 3) wait_vmcnt<2 * C::NPT>();
            else if (behind == 1) wait_vmcnt<C::NPT>();
            else wait_vmcnt<0>();
            lds_barrier();
            products(k);
            if (k + C::NSLOT < K && dma) {
                lds_barrier();                                        // slot k % NSLOT has been read by everyone
                issue_frame(k + C::NSLOT);
            }
        }
    } else {
        int k = 0;
        if (dma) for (; k + D < K; k++) frame_step(k, std::true_type{}, std::false_type{});
        // the last D frames (all frames without DMA): nothing left to issue; once fewer than D frames are outstanding
        // the counted wait would be too lax, so these steps drain
        for (; k < K; k++) {
            if (k + D - 1 < K && dma) frame_step(k, std::false_type{}, std::false_type{});
            else frame_step(k, std::false_type{}, std::true_type{});
        }
    }
    if (A.ablate & 8) break;
    // ---- all frames of this pass are in.  Plane P = g*UB + u of the pass; round f puts planes [f*PR, (f+1)*PR) into
    // LDS (transposed, [column][row]), filters them along the columns in place and along the rows on the way out.
    const int nu = min(UB, ms - oj0);                                 // column offsets oj0 .. oj0+nu-1 are real
    constexpr int PR = C::PR, VITEMS2 = PR * (C::TR / C::CB) * TC, VROUNDS2 = (VITEMS2 + NT - 1) / NT;
    const bool vec_ok = (A.pitch & 1) == 0;
#pragma unroll
    for (int f = 0; f < NF; f++) {
        // frames (or the previous round's planes) consumed.  RO > 1: the raw planes were last read before the barrier in
        // front of the previous round's row filter, so only the first round has to wait here
        if (RO == 1 || f == 0) __syncthreads();
        if (pactive) {
#pragma unroll
            for (int ro = 0; ro < RO; ro++) {
                const int l0 = (grp * RO + ro) * UB - f * PR;         // round-local index of this row offset's plane u = 0
#pragma unroll
                for (int t = 0; t < C::QB; t++) {
                    const int c = pqb * C::QB + t;
                    if (c < C::QC) {
#pragma unroll
                        for (int u = 0; u < UB; u++)
                            if (l0 + u >= 0 && l0 + u < PR) lds[(l0 + u) * C::PPL + c * C::QP + pr] = acc[ro][t][u];
                    }
                }
            }
        }
        __syncthreads();
        const double* vsrc = lds;                                     // what the row filter reads
        int vpitch = C::PPL;
        if constexpr (RO > 1) {
            // H stage (along columns) into the second area
            double* hout = lds + PR * C::PPL;
#pragma unroll
            for (int rd = 0; rd < C::HROUNDS; rd++) {
                const int it = tid + rd * NT;
                if (it < C::HITEMS) {
                    const int r = it % C::QR, rest = it / C::QR, u = rest % PR, cb = rest / PR;
                    double hres[C::CB];
                    fir_running<NW, C::CB>(lds + u * C::PPL + (cb * C::CB) * C::QP + r, C::QP, sep.hc, hres);
                    double* dst = hout + u * C::HPL + (cb * C::CB) * C::HQP + r;
#pragma unroll
                    for (int o = 0; o < C::CB; o++) dst[o * C::HQP] = hres[o];
                }
            }
            __syncthreads();
            // V stage (along rows) and store, one item per thread and round: (plane of the round, row block, column) with the
            // plane wave-uniform -- the table slot's base address is a scalar, the lane's offset inside a slot is the same in
            // every round.  Column pairs swap half of their rows as below.  Tiles that reach over the region's edge, and odd
            // N1, take the general code.
            // (a plane takes 32 lanes per row block, also on narrower tiles, where the lanes past the last column idle)
            static_assert(PR * (C::TR / C::CB) * 32 <= NT && TC <= 32 && TC % 2 == 0, "one item per thread, plane per two waves");
            const bool inside = prow0 + C::TR <= A.row0 + A.rows && pcol0 + TC <= A.pitch && vec_ok && !(A.ablate & 16);
            if (inside) {
                const int pl = __builtin_amdgcn_readfirstlane(tid / (32 * (C::TR / C::CB)));
                const int P = f * PR + pl, g = P / UB, u = P - g * UB;
                const int c = tid % 32, rb = (tid / 32) % (C::TR / C::CB), odd = c & 1;
                if (pl < PR && P < C::NPL && u < nu && oi0 + g <= ms - 1 && c < TC) {
                    double out[C::CB];
                    fir_running<NW, C::CB>(reinterpret_cast<const double*>(__builtin_assume_aligned(hout + pl * C::HPL + c * C::HQP + rb * C::CB, 16)),
                                           1, sep.hr, out);
                    const int ui = A.sigma * (oi0 + g), uj = A.sigma * (oj0 + u);
                    const size_t slot = (size_t)(ui + ms - 1) * UJ + (uj + ms - 1);
                    UMPA_GLOBAL double* sbase = gpw(A.table) + slot * A.slot_stride + (size_t)(prow0 - A.row0) * A.pitch + pcol0;   // scalar
                    const unsigned voff = (unsigned)((rb * C::CB + odd) * A.pitch + c - odd) * 8u;     // this lane's bytes from there
#pragma unroll
                    for (int h = 0; h < C::CB / 2; h++) {
                        const double give = odd ? out[2 * h] : out[2 * h + 1];
                        const double got = swap_adjacent_lanes(give);
                        pair_t v2; v2[0] = odd ? got : out[2 * h]; v2[1] = odd ? out[2 * h + 1] : got;
                        UMPA_GLOBAL char* rowp = reinterpret_cast<UMPA_GLOBAL char*>(sbase + (size_t)(2 * h) * A.pitch);
                        __builtin_nontemporal_store(v2, reinterpret_cast<UMPA_GLOBAL table_pair_t*>(rowp + voff));
                    }
                }
                continue;                                             // next round
            }
            vsrc = hout; vpitch = C::HPL;
        } else {
            // H stage (along columns), results kept in registers, then written in place
            double hres[C::HROUNDS][C::CB];
#pragma unroll
            for (int rd = 0; rd < C::HROUNDS; rd++) {
                const int it = tid + rd * NT;
                if (it < C::HITEMS) {
                    const int r = it % C::QR, rest = it / C::QR, u = rest % PR, cb = rest / PR;
                    fir_block<NW, C::CB>(lds + u * C::PPL + (cb * C::CB) * C::QP + r, C::QP, sep.hc, hres[rd]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int rd = 0; rd < C::HROUNDS; rd++) {
                const int it = tid + rd * NT;
                if (it < C::HITEMS) {
                    const int r = it % C::QR, rest = it / C::QR, u = rest % PR, cb = rest / PR;
                    double* dst = lds + u * C::PPL + (cb * C::CB) * C::QP + r;
#pragma unroll
                    for (int o = 0; o < C::CB; o++) dst[o * C::QP] = hres[rd][o];
                }
            }
            __syncthreads();
        }
        // V stage (along rows) and store: items (plane, rb, column), column fastest: the lanes of a 32-lane group read
        // 32 columns at the odd pitch QP (16 columns x two row blocks 16 rows apart on 16-column tiles): no bank is met
        // twice.  Two lanes with adjacent columns then swap half of their rows (DPP) so that each writes 16-byte table
        // entries (twice the rate of 8-byte stores): the even column's lane the even rows, the odd one the odd rows.
        // Odd N1 breaks the 16-byte alignment of the table rows: then every lane stores its own column.
#pragma unroll
        for (int rd = 0; rd < VROUNDS2; rd++) {
            const int it = tid + rd * NT;
            if (it < VITEMS2) {
                const int c = it % TC, rest = it / TC;
                const int rbq = rest % (C::TR / C::CB), pl = rest / (C::TR / C::CB);
                const int rb = TC == 16 ? ((rbq & 1) << 1) | (rbq >> 1) : rbq;     // 16-column tiles: row blocks 0,2,1,3
                const int P = f * PR + pl, g = P / UB, u = P - g * UB;
                if (P < C::NPL && u < nu && oi0 + g <= ms - 1) {
                    double out[C::CB];
                    if constexpr (RO > 1) fir_running<NW, C::CB>(vsrc + pl * vpitch + c * C::HQP + rb * C::CB, 1, sep.hr, out);
                    else fir_block<NW, C::CB>(vsrc + pl * vpitch + c * C::QP + rb * C::CB, 1, sep.hr, out);
                    const int ui = A.sigma * (oi0 + g), uj = A.sigma * (oj0 + u);
                    const size_t slot = (size_t)(ui + ms - 1) * UJ + (uj + ms - 1);
                    const int col = pcol0 + c;
                    UMPA_GLOBAL double* dst = gpw(A.table) + slot * A.slot_stride + (size_t)(prow0 + rb * C::CB - A.row0) * A.pitch + col;
                    if (vec_ok) {
                        const int odd = c & 1;
#pragma unroll
                        for (int h = 0; h < C::CB / 2; h++) {
                            const double mine = odd ? out[2 * h + 1] : out[2 * h];
                            const double give = odd ? out[2 * h] : out[2 * h + 1];
                            const double got = swap_adjacent_lanes(give);           // the neighbour column's value of MY row
                            const int o = 2 * h + odd, row = prow0 + rb * C::CB + o;
                            if (row < A.row0 + A.rows && col < A.pitch && !(A.ablate & 16)) {     // (ablation 16: everything but the table stores)
                                pair_t v2; v2[0] = odd ? got : mine; v2[1] = odd ? mine : got;
                                store_table16(dst + (size_t)o * A.pitch - odd, v2);
                            }
                        }
                    } else {
#pragma unroll
                        for (int o = 0; o < C::CB; o++)
                            if (prow0 + rb * C::CB + o < A.row0 + A.rows && col < A.N1) dst[(size_t)o * A.pitch] = out[o];
                    }
                }
            }
        }
    }
  } while (0);
}

// (tile, pass) of a workgroup of a static grid: the passes of a tile on consecutive slots of ONE XCD (blocks b, b+8, ...
// share an XCD), so that their re-reads of the tile's patches are served by that XCD's L2; the TILES go to the XCDs in turn
// (round 4, late: until then every XCD had a contiguous band of tiles -- neighbouring tiles on eight L2s instead of one: C2
// corr_volume 1.28 -> 1.23 ms, one rank's slab of C4 2.62 -> 2.46; the passes of a tile on different XCDs as well: 1.37).
// Placement only affects speed.
// od.mode 0: every pass of every tile; 3: a compact grid over the seed tiles (umpa_ondemand.h).
__device__ __forceinline__ bool od_static_item(const OdCorr& od, int ntx, int ntiles, int npass, int& lin, int& pass)
{
    const int nt = od.mode == 3 ? od.nseed : ntiles, tiles_per_xcd = (nt + 7) >> 3;
    const int seq = blockIdx.x >> 3;                                  // position in this XCD's queue
    const int t = (seq / npass) * 8 + (blockIdx.x & 7);               // tile 8 q + x on XCD x
    pass = seq % npass;
    if (seq / npass >= tiles_per_xcd || t >= nt) return false;
    lin = od.mode == 3 ? (od.r0 + (t / od.nsx) * OD_SP) * ntx + od.c0 + (t % od.nsx) * OD_SP : t;
    return true;
}

// the planes of this (tile, pass) are on their way to the table: valid for every later kernel
__device__ __forceinline__ void od_mark_done(const OdCorr& od, int lin, int pass)
{
    if (od.mode != 0 && threadIdx.x == 0) { atomicOr(od.done + lin, 1ull << pass); atomicAdd(od.ndone, 1); }
}

// One workgroup = one (tile, pass): a pass is UI consecutive row offsets and one batch of UB column offsets.
template <int NW, int UB, int TC, int NTG, int UI, int WPC, int NF, int RO>
__global__ void __launch_bounds__(NTG * UI, (NTG * UI / 64 * WPC + 3) / 4)
corr_volume_kernel(ModelDev m, CorrArgs A, Sep1D sep, OdCorr od)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int UJ = 2 * m.ms - 1, npass = ((UJ + UI * RO - 1) / (UI * RO)) * ((UJ + UB - 1) / UB);
    int lin, pass;
    if (!od_static_item(od, A.ntx, A.ntx * A.nty, npass, lin, pass)) return;
    corr_volume_tile<NW, UB, TC, NTG, UI, WPC, NF, RO>(m, A, sep, reinterpret_cast<double*>(smem_raw), lin, pass, threadIdx.x);
    od_mark_done(od, lin, pass);
}

// The same over a work list of (tile, pass) pairs, by a persistent grid (the predicted passes and the repair rounds of
// umpa_ondemand.h).  The body is a call in a loop: written out inside the loop it needed more registers than there are.
template <int NW, int UB, int TC, int NTG, int UI, int WPC, int NF, int RO>
__global__ void __launch_bounds__(NTG * UI, (NTG * UI / 64 * WPC + 3) / 4)
corr_volume_queue_kernel(ModelDev m, CorrArgs A, Sep1D sep, OdCorr od)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nitems = __builtin_amdgcn_readfirstlane(*gp(od.nitems));           // (wave-uniform: SGPRs)
    for (int j = 0;; j++) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                                 // (what follows from it is not to be hoisted out of the loop)
        const int idx = od_item_index(nitems, blockIdx.x, gridDim.x, j);
        if (idx < 0) break;
        const int it = __builtin_amdgcn_readfirstlane(gp(od.items)[idx]);
        if (j) __syncthreads();                                       // the previous item's last LDS reads
        corr_volume_tile<NW, UB, TC, NTG, UI, WPC, NF, RO>(m, A, sep, reinterpret_cast<double*>(smem_raw), it >> 8, it & 255, tid);
        od_mark_done(od, it >> 8, it & 255);
    }
}

} // namespace umpa

// umpa_hip.hip -- libumpa_hip.so: C ABI (include/umpa_hip.h) + kernel launches.  gfx950 only.
//
// There is no CPU fallback in this library: every entry point needs a HIP device and fails
// loudly (negative code + umpa_hip_last_error()) without one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <new>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <mutex>
#include <functional>

#include "../../include/umpa_hip.h"
#include "umpa_walk.h"
#include "umpa_direct.h"
#include "umpa_tiled.h"

using namespace umpa;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr, code)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return fail(code, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevBuf {                       // grow-only device scratch
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return -1; }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct TimedLaunch { int name; hipEvent_t t0, t1; double fma; const int* counts; double fma_per; };

} // namespace

struct umpa_hip_model {
    int kind = 0, Na = 0, Nw = 0, ms = 0, padding = 0, subpx = -1, ref_mode = 0, call_cap = UMPA_CALL_CAP;
    int device = 0;
    bool has_mask = false, owns_frames = true;
    bool mask_binary = false;                      // every mask value is 0 or 1 (corr_masked's cheap pair weight)
    std::vector<int> dims, pos;
    std::vector<double*> d_sam, d_ref, d_mask;     // device frame pointers
    void* d_frames_blob = nullptr;                 // one allocation holding all owned frames
    FrameDesc* d_desc = nullptr;
    // sample stepping: descriptor lists of frame subsets (run_match), a ring in page-locked and device memory
    struct SubList { std::vector<FrameDesc> host; int slot; };
    std::map<unsigned, SubList> sub_lists;    // by frame bit mask: what the device slot holds
    FrameDesc* d_sub = nullptr;
    double* d_win = nullptr;
    std::vector<double> win;
    double win_sum = 0.0;
    hipStream_t stream = nullptr;                  // used by the host-I/O entry points
    hipStream_t copy_stream = nullptr;             // downloads of finished row chunks, behind the compute stream
    // double-buffered sample stack (umpa_hip_stage_sample): the next projection is uploaded + flat-corrected on
    // `up_stream` into the back buffer while the current one is matched; a match with F_USE_STAGED swaps them
    hipStream_t up_stream = nullptr;
    void* d_back_blob = nullptr;                   // back sample frames (float64), same layout as the front ones
    void* d_raw_blob = nullptr; size_t raw_cap = 0;
    std::vector<double*> d_sam_back;
    hipEvent_t ev_staged = nullptr;
    hipEvent_t ev_read_done[2] = {nullptr, nullptr};   // after the last match that read sample buffer 0 (front at creation) / 1
    bool read_once[2] = {false, false};
    bool staged = false;
    // UMPA_HIP_F_ASYNC matches in flight (at most two: each owns one of the two device output sets), oldest first:
    // the events of its row pieces (recycled by umpa_hip_wait) and `done`, recorded behind its last download
    struct PendingMatch { std::vector<hipEvent_t> events; hipEvent_t done; };
    std::vector<PendingMatch> pending;
    int out_set = 0;                               // device output set of the next host-array match
    umpa_hip_rows_fn rows_cb = nullptr;            // umpa_hip_set_rows_callback: told after the kernels of a row piece are enqueued
    void* rows_user = nullptr;
    int rows_piece_rows = 0;
    FrameDesc* h_desc = nullptr;                   // pinned host copy of the descriptor table (source of the stream-ordered update)
    // device copies of a host-array match's arrays, two sets: the maps of match p travel to the host while match p + 1
    // (the step-scan pipeline, umpa_amd/farm.py) already writes the other set
    DevBuf b_values[2], b_uv[2], b_err[2], b_cover[2], b_dd[2], b_da[2], b_dn[2];
    DevBuf b_covout, b_small, b_kern;
    DevBuf t_values, t_uv, t_err, t_cover, t_dd, t_da, t_dn;   // dense outputs of a sub-rectangle (sample-stepping split)
    TiledState tiled;                              // scratch of the tiled fast path
    int last_path = 0;
    bool timing = false;
    std::vector<TimedLaunch> launches;
    std::vector<hipEvent_t> event_pool;
    std::vector<std::string> tnames;
    std::vector<double> tms;
    std::vector<int> tcount;
    std::vector<double> tfma;

    ModelDev dev() const
    {
        ModelDev d;
        d.frames = d_desc; d.win = d_win; d.win_sum = win_sum;
        d.Na = Na; d.Nwt = Na; d.Nw = Nw; d.ms = ms; d.padding = padding; d.subpx = subpx; d.ref_mode = ref_mode; d.call_cap = call_cap;
        return d;
    }
};

namespace {

const char* const KERNEL_NAMES[] = {"match_direct", "coverage", "prep_maps", "corr_volume", "replay_walk", "match_staged", "corr_masked", "replay_cost", "blur_tiles", "corr_march"};
enum { KN_DIRECT = 0, KN_COVER = 1, KN_PREP = 2, KN_CORR = 3, KN_REPLAY = 4, KN_STAGED = 5, KN_MASKED = 6, KN_REPLAY_COST = 7, KN_BLUR = 8, KN_MARCH = 9 };

hipEvent_t get_event(umpa_hip_model* m)
{
    if (!m->event_pool.empty()) { hipEvent_t e = m->event_pool.back(); m->event_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct ScopedTimer {                   // brackets a launch with events when timing is enabled
    umpa_hip_model* m; hipStream_t s; TimedLaunch tl; bool on;
    ScopedTimer(umpa_hip_model* m_, hipStream_t s_, int name) : m(m_), s(s_), on(m_->timing)
    {
        if (!on) return;
        tl.name = name; tl.t0 = get_event(m); tl.t1 = get_event(m); tl.fma = 0.0; tl.counts = nullptr; tl.fma_per = 0.0;
        if (!tl.t0 || !tl.t1) { on = false; return; }
        (void)hipEventRecord(tl.t0, s);
    }
    ~ScopedTimer()
    {
        if (!on) return;
        (void)hipEventRecord(tl.t1, s);
        m->launches.push_back(tl);
    }
};

int upload_win(umpa_hip_model* m, const double* win, int Nw)
{
    const int S = 2 * Nw + 1;
    m->win.assign(win, win + (size_t)S * S);
    double s = 0.0;
    for (int n = 0; n < S * S; n++) s += win[n];     // row-major order, as Model.cpp:725-736
    m->win_sum = s;
    if (m->d_win) (void)hipFree(m->d_win);
    m->d_win = nullptr;
    HIP_TRY(hipMalloc((void**)&m->d_win, (size_t)S * S * sizeof(double)), UMPA_HIP_E_NOMEM);
    HIP_TRY(hipMemcpy(m->d_win, win, (size_t)S * S * sizeof(double), hipMemcpyHostToDevice), UMPA_HIP_E_DEVICE);
    m->Nw = Nw;
    tiled_factor_window(m->tiled, win, Nw);
    return 0;
}

int check_region(const umpa_hip_model* m, int start0, int step0, int N0, int start1, int step1, int N1)
{
    if (N0 <= 0 || N1 <= 0 || step0 <= 0 || step1 <= 0 || start0 < 0 || start1 < 0)
        return fail(UMPA_HIP_E_ARG, "bad region start/step/N (%d,%d,%d ; %d,%d,%d)", start0, step0, N0, start1, step1, N1);
    // extent = max(pos+shape) - 2*padding  (model.pyx:531-549)
    int e0 = 0, e1 = 0;
    for (int k = 0; k < m->Na; k++) {
        e0 = std::max(e0, m->pos[2 * k] + m->dims[2 * k]);
        e1 = std::max(e1, m->pos[2 * k + 1] + m->dims[2 * k + 1]);
    }
    e0 -= 2 * m->padding; e1 -= 2 * m->padding;
    if (start0 + (long)step0 * (N0 - 1) >= e0 || start1 + (long)step1 * (N1 - 1) >= e1)
        return fail(UMPA_HIP_E_ARG, "region exceeds the reconstructible extent %d x %d", e0, e1);
    return 0;
}

template <int KIND, bool MASK, int NWC>
void launch_direct_nw(umpa_hip_model* m, const RegionArgs& A, hipStream_t s)
{
    const int nbx = (A.N1 + UMPA_DIRECT_BX - 1) / UMPA_DIRECT_BX;
    const int nby = (A.N0 + UMPA_DIRECT_BY - 1) / UMPA_DIRECT_BY;
    const int total = nbx * nby;
    const int grid = ((total + 7) / 8) * 8;          // multiple of 8 so the XCD band remap covers every tile
    ScopedTimer t(m, s, KN_DIRECT);
    hipLaunchKernelGGL((match_direct_kernel<KIND, MASK, NWC>), dim3(grid), dim3(UMPA_DIRECT_BX, UMPA_DIRECT_BY), 0, s,
                       m->dev(), A, nbx, nby);
}

// the window half-widths of practical use get their own instantiation (unrolled window rows); the kernel-dark-field
// model and anything else run the generic one
template <int KIND, bool MASK>
void launch_direct(umpa_hip_model* m, const RegionArgs& A, hipStream_t s)
{
    {                                                                 // (the kernel-dark-field model too: its evaluations read the pixel's
        switch (m->Nw) {                                                // blurred footprint a window row at a time, all loads of a row in flight)
        case 1: return launch_direct_nw<KIND, MASK, 1>(m, A, s);
        case 2: return launch_direct_nw<KIND, MASK, 2>(m, A, s);
        case 3: return launch_direct_nw<KIND, MASK, 3>(m, A, s);
        case 4: return launch_direct_nw<KIND, MASK, 4>(m, A, s);
        case 5: return launch_direct_nw<KIND, MASK, 5>(m, A, s);
        case 6: return launch_direct_nw<KIND, MASK, 6>(m, A, s);
        case 7: return launch_direct_nw<KIND, MASK, 7>(m, A, s);
        case 8: return launch_direct_nw<KIND, MASK, 8>(m, A, s);
        default: break;
        }
    }
    launch_direct_nw<KIND, MASK, 0>(m, A, s);
}

// Footprints of the staged general kernel for this region; false if they do not fit beside the walk memo
bool staged_geometry(const umpa_hip_model* m, const RegionArgs& A, StagedGeom& G, size_t& lds_bytes)
{
    const int boxr = (UMPA_STAGED_BY - 1) * A.step0 + 1, boxc = (UMPA_STAGED_BX - 1) * A.step1 + 1;
    const int wide = m->Nw + std::max(m->ms - 1, 0);
    G.hq = m->ref_mode ? wide : m->Nw;                                // the sample window moves in 'ref' mode (Model.cpp:688-701)
    G.hr = m->ref_mode ? m->Nw : wide;
    G.hm = wide;
    G.rowsQ = boxr + 2 * G.hq; G.colsQ = boxc + 2 * G.hq;
    G.rowsR = boxr + 2 * G.hr; G.colsR = boxc + 2 * G.hr;
    G.rowsM = m->has_mask ? boxr + 2 * G.hm : 0; G.colsM = m->has_mask ? boxc + 2 * G.hm : 0;
    G.offR = (G.rowsQ * G.colsQ + 1) & ~1;
    G.offM = (G.offR + G.rowsR * G.colsR + 1) & ~1;
    G.offW = (G.offM + G.rowsM * G.colsM + 1) & ~1;
    const int S = 2 * m->Nw + 1;
    lds_bytes = (size_t)(G.offW + S * S + 2) * sizeof(double);
    const size_t memo = (size_t)25 * UMPA_STAGED_THREADS * sizeof(double);
    const int maxr = UMPA_STAGED_NR * UMPA_STAGED_BY, maxc = UMPA_STAGED_NC * UMPA_STAGED_BX;
    if (std::max(G.rowsQ, std::max(G.rowsR, G.rowsM)) > maxr || std::max(G.colsQ, std::max(G.colsR, G.colsM)) > maxc) return false;
    return lds_bytes + memo <= (size_t)UMPA_LDS_BUDGET - 1024;
}

template <int KIND, bool MASK, int NWC>
hipError_t launch_staged_nw(umpa_hip_model* m, const RegionArgs& A, const StagedGeom& G, size_t lds_bytes, hipStream_t s)
{
    static bool attr_set[64] = {};
    int devid = 0;
    (void)hipGetDevice(&devid);
    {
        std::lock_guard<std::mutex> lock(tiled_attr_mutex());
        if (!attr_set[devid & 63]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&match_staged_kernel<KIND, MASK, NWC>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(UMPA_LDS_BUDGET - 25 * UMPA_STAGED_THREADS * sizeof(double) - 1024));
            if (e != hipSuccess) return e;
            attr_set[devid & 63] = true;
        }
    }
    const int nbx = (A.N1 + UMPA_STAGED_BX - 1) / UMPA_STAGED_BX, nby = (A.N0 + UMPA_STAGED_BY - 1) / UMPA_STAGED_BY;
    const int grid = ((nbx * nby + 7) / 8) * 8;
    ScopedTimer t(m, s, KN_STAGED);
    hipLaunchKernelGGL((match_staged_kernel<KIND, MASK, NWC>), dim3(grid), dim3(UMPA_STAGED_BX, UMPA_STAGED_BY), lds_bytes, s,
                       m->dev(), A, nbx, nby, G);
    return hipGetLastError();
}

template <int KIND, bool MASK>
hipError_t launch_staged(umpa_hip_model* m, const RegionArgs& A, const StagedGeom& G, size_t lds_bytes, hipStream_t s)
{
    switch (m->Nw) {
    case 1: return launch_staged_nw<KIND, MASK, 1>(m, A, G, lds_bytes, s);
    case 2: return launch_staged_nw<KIND, MASK, 2>(m, A, G, lds_bytes, s);
    case 3: return launch_staged_nw<KIND, MASK, 3>(m, A, G, lds_bytes, s);
    case 4: return launch_staged_nw<KIND, MASK, 4>(m, A, G, lds_bytes, s);
    case 5: return launch_staged_nw<KIND, MASK, 5>(m, A, G, lds_bytes, s);
    case 6: return launch_staged_nw<KIND, MASK, 6>(m, A, G, lds_bytes, s);
    case 7: return launch_staged_nw<KIND, MASK, 7>(m, A, G, lds_bytes, s);
    case 8: return launch_staged_nw<KIND, MASK, 8>(m, A, G, lds_bytes, s);
    default: return launch_staged_nw<KIND, MASK, 0>(m, A, G, lds_bytes, s);
    }
}

int run_direct(umpa_hip_model* m, const RegionArgs& A0, hipStream_t s, int flags = 0)
{
    RegionArgs A = A0;
    A.kern = nullptr; A.kern_stride = 0; A.row_base = 0; A.blur = nullptr; A.blur_F = 0; A.blur_ready = 0;
    // the staged kernel (windows out of LDS) for regions of more than a few pixels whose footprints fit; the plain one
    // (windows through L1) for the kernel-dark-field model, single pixels, large steps, or on request
    static const bool no_staged = getenv("UMPA_HIP_NO_STAGED") != nullptr;
    StagedGeom G;
    size_t lds_bytes = 0;
    // (with masks the staged kernel is the slower one -- three footprints leave room for one workgroup per CU and the
    // weights make the summation VALU-bound: 225 against 152 ms on C2 -- so masked models stay on match_direct)
    if (m->kind != UMPA_HIP_KIND_DFKERNEL && !m->has_mask && !no_staged && !(flags & UMPA_HIP_F_FORCE_PLAIN_DIRECT) &&
        (size_t)A.N0 * A.N1 >= 64 && A.N1 >= 8 && staged_geometry(m, A, G, lds_bytes)) {      // (a column or two: one lane per workgroup row would work)
        hipError_t e;
        // (no masked instantiation: masked models never come here -- above -- and run on the tiled path, umpa_masked.h)
        if (m->kind == UMPA_HIP_KIND_NODF) e = launch_staged<0, false>(m, A, G, lds_bytes, s);
        else e = launch_staged<1, false>(m, A, G, lds_bytes, s);
        if (e != hipSuccess) return fail(UMPA_HIP_E_LAUNCH, "staged kernel: %s", hipGetErrorString(e));
        m->last_path = 3;
        return 0;
    }
    if (m->kind == UMPA_HIP_KIND_DFKERNEL) {
        // every pixel carries its own 17x17 blur kernel (Model.cpp:88-117: 289 doubles) and, unless UMPA_HIP_DFK_NO_REUSE
        // is set, the reference blurred with it over the footprint its evaluations read (blur_footprint: Na * F * F
        // doubles): the region is matched in row chunks whose scratch stays below 2 GiB
        static const bool no_reuse = getenv("UMPA_HIP_DFK_NO_REUSE") != nullptr;
        const int halo = m->Nw + (m->ref_mode ? 0 : std::max(m->ms - 1, 0));    // the reference window moves in 'sam' mode only
        const int F = 2 * halo + 1;
        const size_t per_px = (size_t)UMPA_BLUR_TAPS + (no_reuse ? 0 : (size_t)m->Na * F * F);
        const size_t per_row = per_px * A.N1 * sizeof(double);
        int rows_chunk = (int)std::max<size_t>(1, ((size_t)2 << 30) / per_row);
        rows_chunk = std::min(rows_chunk, A.N0);
        if (m->b_kern.reserve(per_row * rows_chunk)) return fail(UMPA_HIP_E_NOMEM, "blur-kernel scratch");
        for (int row0 = 0; row0 < A0.N0; row0 += rows_chunk) {
            RegionArgs C = A0;
            const size_t px0 = (size_t)row0 * A0.N1;
            C.org0 = A0.org0 + A0.step0 * row0;
            C.N0 = std::min(rows_chunk, A0.N0 - row0);
            C.values = A0.values + px0 * A0.v_px;
            C.uv = A0.uv ? A0.uv + 2 * px0 : nullptr;
            C.err = A0.err + px0;
            C.cover = A0.cover ? A0.cover + px0 : nullptr;
            C.dbg_d = A0.dbg_d ? A0.dbg_d + 25 * px0 : nullptr;
            C.dbg_a = A0.dbg_a ? A0.dbg_a + 16 * px0 : nullptr;
            C.dbg_n = A0.dbg_n ? A0.dbg_n + px0 : nullptr;
            C.kern = (double*)m->b_kern.p; C.kern_stride = (size_t)C.N0 * C.N1; C.row_base = 0;
            C.blur = no_reuse ? nullptr : C.kern + (size_t)UMPA_BLUR_TAPS * C.kern_stride;
            C.blur_F = F;
            C.blur_ready = 0;
            // the footprints of a 64 x 4 pixel box at a time, reference patch staged in LDS (blur_tiles_kernel); where the
            // patch does not fit (large steps) or on request (UMPA_HIP_DFK_NO_TILES) each lane fills its own as before
            static const bool no_tiles = getenv("UMPA_HIP_DFK_NO_TILES") != nullptr;
            const BlurTileGeom BG = blur_tile_geometry(halo, C.step0, C.step1);
            const size_t blur_lds = (size_t)BG.PR * BG.PC * sizeof(double) * (m->has_mask ? 2 : 1);
            if (!no_reuse && !no_tiles && blur_lds <= (size_t)UMPA_LDS_BUDGET / 2) {
                static bool battr[2][64] = {};
                {
                    std::lock_guard<std::mutex> lock(tiled_attr_mutex());
                    if (!battr[m->has_mask][m->device & 63]) {
                        hipError_t be = m->has_mask
                            ? hipFuncSetAttribute(reinterpret_cast<const void*>(&blur_tiles_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, UMPA_LDS_BUDGET / 2)
                            : hipFuncSetAttribute(reinterpret_cast<const void*>(&blur_tiles_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, UMPA_LDS_BUDGET / 2);
                        if (be != hipSuccess) return fail(UMPA_HIP_E_LAUNCH, "blur_tiles attribute: %s", hipGetErrorString(be));
                        battr[m->has_mask][m->device & 63] = true;
                    }
                }
                dim3 bgrid((C.N1 + UMPA_BLURT_BX - 1) / UMPA_BLURT_BX, (C.N0 + UMPA_BLURT_BY - 1) / UMPA_BLURT_BY), bblk(UMPA_BLURT_BX, UMPA_BLURT_BY);
                {
                    ScopedTimer t(m, s, KN_BLUR);
                    if (m->has_mask) hipLaunchKernelGGL((blur_tiles_kernel<true>), bgrid, bblk, blur_lds, s, m->dev(), C, BG);
                    else hipLaunchKernelGGL((blur_tiles_kernel<false>), bgrid, bblk, blur_lds, s, m->dev(), C, BG);
                }
                HIP_TRY(hipGetLastError(), UMPA_HIP_E_LAUNCH);
                C.blur_ready = 1;
            }
            if (m->has_mask) launch_direct<2, true>(m, C, s); else launch_direct<2, false>(m, C, s);
            HIP_TRY(hipGetLastError(), UMPA_HIP_E_LAUNCH);
        }
        m->last_path = 1;
        return 0;
    }
    if (m->kind == UMPA_HIP_KIND_NODF) { if (m->has_mask) launch_direct<0, true>(m, A, s); else launch_direct<0, false>(m, A, s); }
    else { if (m->has_mask) launch_direct<1, true>(m, A, s); else launch_direct<1, false>(m, A, s); }
    HIP_TRY(hipGetLastError(), UMPA_HIP_E_LAUNCH);
    m->last_path = 1;
    return 0;
}

// Can the tiled fast path take this model and region?  (see umpa_tiled.h for what it covers; frame positions are
// dealt with by run_match)
bool tiled_applicable(const umpa_hip_model* m, const RegionArgs& A, bool forced = false)
{
    if (m->kind == UMPA_HIP_KIND_DFKERNEL) return false;
    if (m->has_mask && !masked_supported(m->Nw)) return false;        // corr_masked + replay_cost (umpa_masked.h)
    for (int k = 0; k < m->Na; k++)
        if (m->dims[2 * k] != m->dims[0] || m->dims[2 * k + 1] != m->dims[1]) return false;
    // corr_volume / corr_masked address a frame with 32-bit byte offsets in IMAGE coordinates at the frames' pitch (LDS-DMA
    // source = (frame - its position) + per-lane offset, the offset up to the image's last row: ADVICE round 3)
    {
        size_t Himg = 0, Wimg = 0;
        for (int k = 0; k < m->Na; k++) {
            Himg = std::max(Himg, (size_t)m->pos[2 * k] + m->dims[2 * k]); Wimg = std::max(Wimg, (size_t)m->pos[2 * k + 1] + m->dims[2 * k + 1]);
        }
        if ((Himg + 1) * (size_t)m->dims[1] * sizeof(double) >= ((size_t)1 << 32)) return false;
        // corr_masked stages the means from 16-byte pair planes of the IMAGE
        if (m->has_mask && m->kind == UMPA_HIP_KIND_DF && Himg * Wimg * 16 >= ((size_t)1 << 32)) return false;
    }
    // One frame, 3x3 windows and masks: a window with as many valid pixels as the model has parameters is fitted EXACTLY,
    // every cost there is rounding noise around zero and the walk follows the noise -- i.e. the order of the sums.  The
    // general kernel sums in the reference's order (its walks are the oracle's, test_exact_fit_windows_follow_the_reference_order);
    // the table-based path does not.
    if (m->has_mask && m->Na * (2 * m->Nw + 1) * (2 * m->Nw + 1) <= 9 && !forced) return false;
    // stepped regions: the tiled kernels still compute the dense grid, the direct kernel only the requested pixels at
    // about 20 costs each.  Measured on 2048^2 x 10 frames, Nw 5, max_shift 5 (tools/step_rate.py): step 3: 3.5 vs 9.8 ms,
    // 4: 2.5 vs 7.6, 5: 2.4 vs 6.8, 6: 2.4 vs 5.0 -- the dense grid wins up to about 64 dense pixels per requested
    // one there.  The table grows with the number of shifts, the direct walk does not: the limit scales with 81 / planes.
    // Masked models cost 7x (NoDF 2x) more per dense pixel on the tiled path and about the same on the direct one.
    {
        static const char* se = getenv("UMPA_HIP_TILED_MAX_STEP2");   // tuning override
        const int planes = (2 * m->ms - 1) * (2 * m->ms - 1);
        const int base = !m->has_mask ? 64 : m->kind == UMPA_HIP_KIND_NODF ? 16 : 5;
        int lim = planes <= 81 ? base : base * 81 / planes;
        if (lim < 1) lim = 1;
        if (se) lim = atoi(se);
        if (forced && lim < 81) lim = 81;                              // UMPA_HIP_F_FORCE_TILED: the caller's choice, not the economics
        if (A.step0 * A.step1 > lim) return false;
    }
    // the exhaustive table has (2 max_shift - 1)^2 planes: beyond a few hundred shifts the lazy evaluation of the
    // direct kernel (about 20 costs per pixel) is less work than filling it
    if ((2 * m->ms - 1) * (2 * m->ms - 1) > 1089) return false;
    // the region must keep every window inside the frames even for the partial tiles' halo reads: guaranteed
    // by check_region + clamped staging.  Separable window required (always true for the Hamming window).
    if (!m->tiled.separable || m->tiled.sep_nw != m->Nw) return false;
    return tiled_supported(m->Nw, m->ms, m->Na);
}

// Frames of one shape at different positions (sample stepping, Model.cpp:428-433 / :716-719): the image they tile,
// the box of image rows / columns that lie inside every frame, and the rectangle [a0,b0) x [a1,b1) of region pixels
// that every frame contributes to (coverage = Na there, so those pixels see exactly the all-frames sums the tiled
// kernels form).
struct StepGeom { bool any_pos; int Himg, Wimg; FrameBox box; int a0, b0, a1, b1; };

// corr_volume stages 16-byte column pairs: a pair that starts on a frame's last column reads 8 bytes past the end of the
// row.  Frames the library owns are laid out back to back with 16 spare bytes at the end of the allocation, so those bytes
// are readable (and never used); borrowed device frames give no such promise.  (corr_masked keeps its own clamp.)
bool pair_slack(const umpa_hip_model* m) { return m->owns_frames && !m->has_mask; }

StepGeom step_geometry(const umpa_hip_model* m, const RegionArgs& A)
{
    StepGeom g;
    g.any_pos = false;
    int r0 = 0, r1 = 1 << 30, c0 = 0, c1 = 1 << 30;
    g.Himg = g.Wimg = 0;
    for (int k = 0; k < m->Na; k++) {
        const int pi = m->pos[2 * k], pj = m->pos[2 * k + 1], H = m->dims[2 * k], W = m->dims[2 * k + 1];
        if (pi || pj) g.any_pos = true;
        r0 = std::max(r0, pi); r1 = std::min(r1, pi + H - 1);
        c0 = std::max(c0, pj); c1 = std::min(c1, pj + W - 1);
        g.Himg = std::max(g.Himg, pi + H); g.Wimg = std::max(g.Wimg, pj + W);
    }
    g.box.r0 = r0; g.box.r1 = r1; g.box.c0 = c0; g.box.c1 = c1; g.box.Wf = m->dims[1];
    // a frame contributes at image pixel (i, j) iff i - pi - pad >= 0 and i - pi + pad <= H (same for j)
    const int pad = m->padding;
    // (the last contributing column, j = c1 + 1 - pad, would make corr_volume read the very last column of a frame as the
    // first half of a 16-byte column pair: it is left to the border strip)
    // -- unless 8 readable bytes follow every frame (pair_slack: frames the library owns, models without masks)
    g.box.slack = pair_slack(m) ? 1 : 0;
    const int imin = r0 + pad, imax = r1 + 1 - pad, jmin = c0 + pad, jmax = c1 - pad + g.box.slack;
    auto lo = [](int vmin, int org, int step, int N) { int q = vmin - org; q = q <= 0 ? 0 : (q + step - 1) / step; return std::min(q, N); };
    auto hi = [](int vmax, int org, int step, int N) { int q = vmax - org; q = q < 0 ? 0 : q / step + 1; return std::min(q, N); };
    g.a0 = lo(imin, A.org0, A.step0, A.N0); g.b0 = std::max(g.a0, hi(imax, A.org0, A.step0, A.N0));
    g.a1 = lo(jmin, A.org1, A.step1, A.N1); g.b1 = std::max(g.a1, hi(jmax, A.org1, A.step1, A.N1));
    return g;
}

// `sub`: the frames that contribute in this sub-rectangle of a sample-stepping stack (a descriptor list of its own, the box
// of image rows / columns inside every one of them); NULL = all frames
struct FrameSubset { const FrameDesc* frames; int n; FrameBox box; const FrameDesc* host; };   // host: the same list in host memory

int run_tiled(umpa_hip_model* m, const RegionArgs& A, const StepGeom& g, int flags, hipStream_t s,
              int piece_rows, const std::function<void(int, int)>& on_rows, const FrameSubset* sub = nullptr)
{
    TiledTimers tt;
    tt.get = [m]() { return get_event(m); };
    ModelDev dev = m->dev();
    if (sub) { dev.frames = sub->frames; dev.Na = sub->n; m->tiled.ref_maps_ok = false; }      // (Nwt stays the model's frame count)
    const bool reuse = !sub && (flags & UMPA_HIP_F_REUSE_REF_MAPS) != 0;
    std::vector<FrameDesc> hf;                                        // a host copy of the frame list (corr_march's staging offsets)
    if (!sub) {
        hf.resize(m->Na);
        for (int k = 0; k < m->Na; k++) {
            memset(&hf[k], 0, sizeof(FrameDesc));
            hf[k].sam = m->d_sam[k]; hf[k].ref = m->d_ref[k]; hf[k].mask = m->d_mask[k];
            hf[k].H = m->dims[2 * k]; hf[k].W = m->dims[2 * k + 1]; hf[k].pi = m->pos[2 * k]; hf[k].pj = m->pos[2 * k + 1];
        }
    }
    int rc = m->has_mask
        ? tiled_match_masked(m->tiled, dev, m->kind, g.Himg, g.Wimg, sub ? sub->box : g.box, A, s,
                             m->timing ? &tt : nullptr, reuse, m->mask_binary, piece_rows, on_rows)
        : tiled_match(m->tiled, dev, m->kind, g.Himg, g.Wimg, sub ? sub->box : g.box, A, s,
                      m->timing ? &tt : nullptr, reuse, piece_rows, on_rows, sub ? sub->host : hf.data());
    if (sub) m->tiled.ref_maps_ok = false;                            // the maps now hold a subset's planes
    if (rc == -3) return fail(UMPA_HIP_E_NOMEM, "tiled path: device scratch allocation failed");
    if (rc != 0) return fail(UMPA_HIP_E_LAUNCH, "tiled path: launch failed (%d): %s", rc, hipGetErrorString(hipGetLastError()));
    for (const auto& en : tt.entries) { TimedLaunch tl; tl.name = en.name; tl.t0 = en.t0; tl.t1 = en.t1; tl.fma = en.fma; tl.counts = en.counts; tl.fma_per = en.fma_per; m->launches.push_back(tl); }
    return 0;
}

// copy a block of pixels (rows x cols, `epp` bytes per pixel) between two row-major pixel arrays on the device
hipError_t copy_block(void* dst, int dN1, int dr, int dc, const void* src, int sN1, int sr, int sc,
                      int rows, int cols, size_t epp, hipStream_t s)
{
    if (rows <= 0 || cols <= 0) return hipSuccess;
    return hipMemcpy2DAsync((char*)dst + ((size_t)dr * dN1 + dc) * epp, (size_t)dN1 * epp,
                            (const char*)src + ((size_t)sr * sN1 + sc) * epp, (size_t)sN1 * epp,
                            (size_t)cols * epp, (size_t)rows, hipMemcpyDeviceToDevice, s);
}

// One sub-rectangle [r0,r1) x [c0,c1) of the region, matched into dense scratch arrays and copied back: `tiled` sends
// it down the tiled path (every frame contributes everywhere in it), otherwise the general kernels take it.
int run_block(umpa_hip_model* m, const RegionArgs& A, const StepGeom& g, int r0, int r1, int c0, int c1, bool tiled,
              int flags, hipStream_t s, const FrameSubset* sub = nullptr)
{
    if (tiled) {
        // the tiled kernels write this rectangle of the full arrays in place (row pitch = the full region's): nothing is
        // copied, pixels below the coverage threshold keep what they had
        RegionArgs B = A;
        const size_t off = (size_t)r0 * A.pitch + c0;
        B.org0 = A.org0 + A.step0 * r0; B.N0 = r1 - r0;
        B.org1 = A.org1 + A.step1 * c0; B.N1 = c1 - c0;
        if (B.N0 <= 0 || B.N1 <= 0) return 0;
        B.values = A.values + off * A.v_px; B.err = A.err + off;
        if (A.uv) B.uv = A.uv + 2 * off;
        if (A.cover) B.cover = A.cover + off;
        if (A.dbg_d) B.dbg_d = A.dbg_d + 25 * off;
        if (A.dbg_a) B.dbg_a = A.dbg_a + 16 * off;
        if (A.dbg_n) B.dbg_n = A.dbg_n + off;
        return run_tiled(m, B, g, flags, s, 0, nullptr, sub);
    }
    const bool keep_in = true;                                        // pixels the kernels may leave untouched: coverage threshold
    const int rows = r1 - r0, cols = c1 - c0;
    if (rows <= 0 || cols <= 0) return 0;
    const size_t n = (size_t)rows * cols, nfull = (size_t)A.N0 * A.N1;
    const bool planar = A.v_px == 1 && A.nparam > 1;
    const int np = A.nparam;
    if (m->t_values.reserve(n * np * sizeof(double)) || m->t_err.reserve(n * sizeof(int))) return fail(UMPA_HIP_E_NOMEM, "block scratch");
    RegionArgs B = A;
    B.org0 = A.org0 + A.step0 * r0; B.N0 = rows;
    B.org1 = A.org1 + A.step1 * c0; B.N1 = cols; B.pitch = cols;
    B.values = (double*)m->t_values.p; B.err = (int*)m->t_err.p;
    B.v_px = planar ? 1 : (size_t)np; B.v_k = planar ? n : 1;
    B.uv = nullptr; B.cover = nullptr; B.dbg_d = nullptr; B.dbg_a = nullptr; B.dbg_n = nullptr;
    hipError_t e = hipSuccess;
    auto in = [&](void* t, const void* full, size_t epp) { if (e == hipSuccess) e = copy_block(t, cols, 0, 0, full, A.N1, r0, c0, rows, cols, epp, s); };
    auto out = [&](void* full, const void* t, size_t epp) { if (e == hipSuccess) e = copy_block(full, A.N1, r0, c0, t, cols, 0, 0, rows, cols, epp, s); };
    // what the kernels may leave untouched (pixels below the coverage threshold) must come back as it was
    if (keep_in) {
        if (planar) for (int k = 0; k < np; k++) in(B.values + (size_t)k * n, A.values + (size_t)k * nfull, sizeof(double));
        else in(B.values, A.values, np * sizeof(double));
        in(B.err, A.err, sizeof(int));
    }
    if (A.uv) { if (m->t_uv.reserve(n * 2 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "block scratch"); B.uv = (double*)m->t_uv.p; in(B.uv, A.uv, 2 * sizeof(double)); }
    if (A.cover && keep_in) { if (m->t_cover.reserve(n * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "block scratch"); in(m->t_cover.p, A.cover, sizeof(double)); B.cover = (const double*)m->t_cover.p; }
    if (A.dbg_d) { if (m->t_dd.reserve(n * 25 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "block scratch"); B.dbg_d = (double*)m->t_dd.p; if (keep_in) in(B.dbg_d, A.dbg_d, 25 * sizeof(double)); }
    if (A.dbg_a) { if (m->t_da.reserve(n * 16 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "block scratch"); B.dbg_a = (double*)m->t_da.p; if (keep_in) in(B.dbg_a, A.dbg_a, 16 * sizeof(double)); }
    if (A.dbg_n) { if (m->t_dn.reserve(n * sizeof(int))) return fail(UMPA_HIP_E_NOMEM, "block scratch"); B.dbg_n = (int*)m->t_dn.p; if (keep_in) in(B.dbg_n, A.dbg_n, sizeof(int)); }
    if (e != hipSuccess) return fail(UMPA_HIP_E_DEVICE, "block gather: %s", hipGetErrorString(e));
    if (int rc = tiled ? run_tiled(m, B, g, flags, s, 0, nullptr, sub) : run_direct(m, B, s, flags)) return rc;
    if (planar) for (int k = 0; k < np; k++) out(A.values + (size_t)k * nfull, B.values + (size_t)k * n, sizeof(double));
    else out(A.values, B.values, np * sizeof(double));
    out(A.err, B.err, sizeof(int));
    if (A.uv) out(A.uv, B.uv, 2 * sizeof(double));
    if (A.dbg_d) out(A.dbg_d, B.dbg_d, 25 * sizeof(double));
    if (A.dbg_a) out(A.dbg_a, B.dbg_a, 16 * sizeof(double));
    if (A.dbg_n) out(A.dbg_n, B.dbg_n, sizeof(int));
    if (e != hipSuccess) return fail(UMPA_HIP_E_DEVICE, "block scatter: %s", hipGetErrorString(e));
    return 0;
}

// Sample stepping: a frame contributes at a pixel iff the pixel lies `padding` inside it (Model.cpp:428-433, :716-719), so
// the region falls into a grid of rectangles inside each of which the SAME frames contribute -- a plain all-frames problem
// on that subset (without masks the cost is still divided by the model's frame count, ModelDev::Nwt; with masks by the sum
// of the pair weights, which only the contributing frames add to).  Every rectangle
// large enough goes down the tiled path with its subset's descriptor list; the general kernels keep the slivers: the last
// contributing column of a frame (corr_volume would read the frame's very last column as the first half of a 16-byte pair),
// rectangles below UMPA_STEP_CELL_MIN pixels and rectangles no frame contributes to.
#define UMPA_STEP_CELL_MIN 1024
#define UMPA_SUB_SLOTS 256
int run_stepping_cells(umpa_hip_model* m, const RegionArgs& A, const StepGeom& g, int flags, hipStream_t s)
{
    const int K = m->Na, pad = m->padding;
    auto lo = [](int vmin, int org, int step, int N) { int q = vmin - org; q = q <= 0 ? 0 : (q + step - 1) / step; return std::min(q, N); };
    auto hi = [](int vmax, int org, int step, int N) { int q = vmax - org; q = q < 0 ? 0 : q / step + 1; return std::min(q, N); };
    // per frame, in region pixel indices: rows [ra, rb) and columns [ca, cb) it contributes to; [ca, cs) = without the last column
    std::vector<int> ra(K), rb(K), ca(K), cb(K), cs(K), rcut{0, A.N0}, ccut{0, A.N1};
    for (int k = 0; k < K; k++) {
        const int pi = m->pos[2 * k], pj = m->pos[2 * k + 1], H = m->dims[2 * k], W = m->dims[2 * k + 1];
        ra[k] = lo(pi + pad, A.org0, A.step0, A.N0); rb[k] = std::max(ra[k], hi(pi + H - pad, A.org0, A.step0, A.N0));
        ca[k] = lo(pj + pad, A.org1, A.step1, A.N1); cb[k] = std::max(ca[k], hi(pj + W - pad, A.org1, A.step1, A.N1));
        cs[k] = pair_slack(m) ? cb[k] : std::max(ca[k], std::min(cb[k], hi(pj + W - pad - 1, A.org1, A.step1, A.N1)));
        rcut.push_back(ra[k]); rcut.push_back(rb[k]);
        ccut.push_back(ca[k]); ccut.push_back(cb[k]); ccut.push_back(cs[k]);
    }
    auto uniq = [](std::vector<int>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    uniq(rcut); uniq(ccut);
    if (!m->d_sub) HIP_TRY(hipMalloc((void**)&m->d_sub, (size_t)UMPA_SUB_SLOTS * K * sizeof(FrameDesc)), UMPA_HIP_E_NOMEM);
    for (size_t x = 0; x + 1 < ccut.size(); x++) {
        const int c0 = ccut[x], c1 = ccut[x + 1];
        bool sliver = false;                                          // some frame's last contributing column(s)
        for (int k = 0; k < K; k++) if (c0 >= cs[k] && c1 <= cb[k] && cs[k] < cb[k]) sliver = true;
        if (sliver) {                                                 // full height, general kernels (they sort out the frames per pixel)
            if (int rc = run_block(m, A, g, 0, A.N0, c0, c1, false, flags, s)) return rc;
            continue;
        }
        for (size_t y = 0; y + 1 < rcut.size(); y++) {
            const int r0 = rcut[y], r1 = rcut[y + 1];
            std::vector<int> S;
            for (int k = 0; k < K; k++) if (r0 >= ra[k] && r1 <= rb[k] && c0 >= ca[k] && c1 <= cs[k]) S.push_back(k);
            unsigned mask = 0;
            for (int k : S) mask |= 1u << k;
            const bool known = m->sub_lists.count(mask) != 0;
            const bool tile = !S.empty() && (size_t)(r1 - r0) * (c1 - c0) >= UMPA_STEP_CELL_MIN &&
                              ((int)S.size() == K || known || (int)m->sub_lists.size() < UMPA_SUB_SLOTS);
            if (!tile) { if (int rc = run_block(m, A, g, r0, r1, c0, c1, false, flags, s)) return rc; continue; }
            if ((int)S.size() == K) { if (int rc = run_block(m, A, g, r0, r1, c0, c1, true, flags, s)) return rc; continue; }   // every frame
            // this subset's descriptor list (a device slot per subset, uploaded when the frames' addresses have changed) and its box
            std::vector<FrameDesc> h(S.size());
            FrameSubset sub;
            sub.n = (int)S.size();
            int br0 = 0, br1 = 1 << 30, bc0 = 0, bc1 = 1 << 30;
            for (int q = 0; q < sub.n; q++) {
                const int k = S[q];
                memset(&h[q], 0, sizeof(FrameDesc));
                h[q].sam = m->d_sam[k]; h[q].ref = m->d_ref[k]; h[q].mask = m->d_mask[k];
                h[q].H = m->dims[2 * k]; h[q].W = m->dims[2 * k + 1]; h[q].pi = m->pos[2 * k]; h[q].pj = m->pos[2 * k + 1];
                br0 = std::max(br0, h[q].pi); br1 = std::min(br1, h[q].pi + h[q].H - 1);
                bc0 = std::max(bc0, h[q].pj); bc1 = std::min(bc1, h[q].pj + h[q].W - 1);
            }
            sub.box.r0 = br0; sub.box.r1 = br1; sub.box.c0 = bc0; sub.box.c1 = bc1; sub.box.Wf = m->dims[1]; sub.box.slack = g.box.slack;
            umpa_hip_model::SubList& L = m->sub_lists[mask];
            if (!known) L.slot = (int)m->sub_lists.size() - 1;
            sub.frames = m->d_sub + (size_t)L.slot * K;
            sub.host = nullptr;
            if (L.host.size() != h.size() || memcmp(L.host.data(), h.data(), h.size() * sizeof(FrameDesc)) != 0) {
                L.host = h;                                          // the copy's source outlives the enqueue: the map's own storage
                HIP_TRY(hipMemcpyAsync((void*)sub.frames, L.host.data(), L.host.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, s), UMPA_HIP_E_DEVICE);
            }
            sub.host = L.host.data();
            if (int rc = run_block(m, A, g, r0, r1, c0, c1, true, flags, s, &sub)) return rc;
        }
    }
    return 0;
}

int run_match(umpa_hip_model* m, const RegionArgs& A, int flags, hipStream_t s,
              int piece_rows = 0, const std::function<void(int, int)>& on_rows = nullptr)
{
    const bool can_tile = tiled_applicable(m, A, (flags & UMPA_HIP_F_FORCE_TILED) != 0);
    const StepGeom g = can_tile ? step_geometry(m, A) : StepGeom();
    const bool whole = can_tile && (!g.any_pos || (g.a0 == 0 && g.b0 == A.N0 && g.a1 == 0 && g.b1 == A.N1));
    // sample stepping: the tiled path takes the rectangle every frame contributes to, the general kernels the border
    // strips around it -- worth the split while the rectangle is most of the region
    // (every rectangle with a constant set of contributing frames can go there, run_stepping_cells)
    const bool cells = can_tile && !whole && m->Na <= 32;
    const bool split = cells || (can_tile && !whole && (size_t)(g.b0 - g.a0) * (g.b1 - g.a1) * 2 >= (size_t)A.N0 * A.N1);
    if ((flags & UMPA_HIP_F_FORCE_TILED) && !(whole || split))
        return fail(UMPA_HIP_E_UNSUPPORTED, "tiled path does not cover this model/region");
    if (whole && !(flags & UMPA_HIP_F_FORCE_DIRECT)) {
        if (int rc = run_tiled(m, A, g, flags, s, piece_rows, on_rows)) return rc;
        m->last_path = 2;
        return 0;
    }
    if (split && !(flags & UMPA_HIP_F_FORCE_DIRECT)) {
        if (cells) {
            if (int rc = run_stepping_cells(m, A, g, flags, s)) return rc;
        } else {
            if (int rc = run_block(m, A, g, g.a0, g.b0, g.a1, g.b1, true, flags, s)) return rc;           // centre: tiled
            if (int rc = run_block(m, A, g, 0, g.a0, 0, A.N1, false, flags, s)) return rc;                 // top strip
            if (int rc = run_block(m, A, g, g.b0, A.N0, 0, A.N1, false, flags, s)) return rc;              // bottom strip
            if (int rc = run_block(m, A, g, g.a0, g.b0, 0, g.a1, false, flags, s)) return rc;              // left strip
            if (int rc = run_block(m, A, g, g.a0, g.b0, g.b1, A.N1, false, flags, s)) return rc;           // right strip
        }
        if (on_rows) on_rows(0, A.N0);
        m->last_path = 4;
        return 0;
    }
    const int rc = run_direct(m, A, s, flags);
    if (rc == 0 && on_rows) on_rows(0, A.N0);
    return rc;
}

// F_USE_STAGED: the stack uploaded by umpa_hip_stage_sample becomes the sample stack, in stream order on `s`
int adopt_staged(umpa_hip_model* m, int flags, hipStream_t s)
{
    if (!(flags & UMPA_HIP_F_USE_STAGED)) return 0;
    if (!m->staged) return fail(UMPA_HIP_E_ARG, "UMPA_HIP_F_USE_STAGED without a staged sample stack");
    HIP_TRY(hipStreamWaitEvent(s, m->ev_staged, 0), UMPA_HIP_E_DEVICE);
    std::swap(m->d_sam, m->d_sam_back);
    // the descriptor table is rewritten in stream order: matches already enqueued on `s` still see the old one
    if (!m->h_desc) HIP_TRY(hipHostMalloc((void**)&m->h_desc, 2 * m->Na * sizeof(FrameDesc), hipHostMallocDefault), UMPA_HIP_E_NOMEM);
    static_assert(sizeof(FrameDesc) % 8 == 0, "descriptor layout");
    FrameDesc* h = m->h_desc + (m->d_sam[0] == (double*)m->d_back_blob ? m->Na : 0);   // one slot per buffer parity
    for (int k = 0; k < m->Na; k++) {
        h[k].sam = m->d_sam[k]; h[k].ref = m->d_ref[k]; h[k].mask = m->d_mask[k];
        h[k].H = m->dims[2 * k]; h[k].W = m->dims[2 * k + 1]; h[k].pi = m->pos[2 * k]; h[k].pj = m->pos[2 * k + 1];
    }
    HIP_TRY(hipMemcpyAsync(m->d_desc, h, m->Na * sizeof(FrameDesc), hipMemcpyHostToDevice, s), UMPA_HIP_E_DEVICE);
    m->staged = false;
    return 0;
}

// which of the two sample buffers is the front one (the stack the kernels read)
int front_buffer(const umpa_hip_model* m) { return (m->d_back_blob && m->d_sam[0] == (double*)m->d_back_blob) ? 1 : 0; }

void mark_matched(umpa_hip_model* m, hipStream_t s)
{
    if (!m->owns_frames) return;
    const int q = front_buffer(m);
    if (!m->ev_read_done[q] && hipEventCreateWithFlags(&m->ev_read_done[q], hipEventDisableTiming) != hipSuccess) { m->ev_read_done[q] = nullptr; return; }
    (void)hipEventRecord(m->ev_read_done[q], s);
    m->read_once[q] = true;
}

} // namespace

extern "C" {

int umpa_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* umpa_hip_last_error(void) { return g_err.c_str(); }
const char* umpa_hip_version(void) { return "umpa_hip 0.1 (gfx950)"; }

umpa_hip_model* umpa_hip_create(int kind, int Na, const int* dims, double* const* sam, double* const* ref,
                                double* const* mask, const int* pos, int Nw, const double* win,
                                int max_shift, int padding, int device, int flags)
{
    if (kind != UMPA_HIP_KIND_NODF && kind != UMPA_HIP_KIND_DF && kind != UMPA_HIP_KIND_DFKERNEL) {
        fail(UMPA_HIP_E_ARG, "unknown model kind %d", kind);
        return nullptr;
    }
    // the kernel-dark-field model blurs the reference on the fly: its reads reach UMPA_BLUR_HALF pixels further (safe_crop, model.pyx:904)
    const int reach = kind == UMPA_HIP_KIND_DFKERNEL ? UMPA_BLUR_HALF : 0;
    if (Na <= 0 || !dims || !sam || !ref || !pos || !win || Nw < 0 || max_shift < 0 || padding < Nw + max_shift + reach) {
        fail(UMPA_HIP_E_ARG, "bad arguments to umpa_hip_create (Na=%d Nw=%d max_shift=%d padding=%d, needs padding >= Nw + max_shift + %d)",
             Na, Nw, max_shift, padding, reach);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fail(UMPA_HIP_E_DEVICE, "no HIP device available (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { fail(UMPA_HIP_E_ARG, "device %d out of range (%d devices)", device, ndev); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(UMPA_HIP_E_DEVICE, "hipSetDevice(%d) failed", device); return nullptr; }

    umpa_hip_model* m = new (std::nothrow) umpa_hip_model;
    if (!m) { fail(UMPA_HIP_E_NOMEM, "out of host memory"); return nullptr; }
    m->kind = kind; m->Na = Na; m->ms = max_shift; m->padding = padding; m->device = device;
    if (const char* cap = getenv("UMPA_CALL_CAP")) m->call_cap = std::max(1, atoi(cap));
    m->has_mask = mask != nullptr;
    m->owns_frames = !(flags & UMPA_HIP_F_DEVICE_FRAMES);
    m->dims.assign(dims, dims + 2 * Na);
    m->pos.assign(pos, pos + 2 * Na);
    for (int k = 0; k < Na; k++) {
        if (dims[2 * k] <= 0 || dims[2 * k + 1] <= 0 || pos[2 * k] < 0 || pos[2 * k + 1] < 0) {
            fail(UMPA_HIP_E_ARG, "frame %d: bad shape or negative position", k);
            delete m;
            return nullptr;
        }
    }
    bool ok = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking) == hipSuccess;

    m->d_sam.resize(Na); m->d_ref.resize(Na); m->d_mask.assign(Na, nullptr);
    if (ok && m->owns_frames) {
        size_t total = 0;
        for (int k = 0; k < Na; k++) total += (size_t)dims[2 * k] * dims[2 * k + 1];
        const size_t nstacks = m->has_mask ? 3 : 2;
        ok = hipMalloc(&m->d_frames_blob, total * nstacks * sizeof(double) + 16) == hipSuccess;     // (+16: pair_slack below)
        if (!ok) fail(UMPA_HIP_E_NOMEM, "cannot allocate %zu bytes for the frame stacks", total * nstacks * sizeof(double));
        double* base = (double*)m->d_frames_blob;
        size_t off = 0;
        for (int k = 0; ok && k < Na; k++) {
            const size_t n = (size_t)dims[2 * k] * dims[2 * k + 1];
            m->d_sam[k] = base + off;
            m->d_ref[k] = base + total + off;
            ok = hipMemcpyAsync(m->d_sam[k], sam[k], n * sizeof(double), hipMemcpyHostToDevice, m->stream) == hipSuccess &&
                 hipMemcpyAsync(m->d_ref[k], ref[k], n * sizeof(double), hipMemcpyHostToDevice, m->stream) == hipSuccess;
            if (ok && m->has_mask) {
                m->d_mask[k] = base + 2 * total + off;
                ok = hipMemcpyAsync(m->d_mask[k], mask[k], n * sizeof(double), hipMemcpyHostToDevice, m->stream) == hipSuccess;
            }
            off += n;
        }
        if (ok) ok = hipStreamSynchronize(m->stream) == hipSuccess;
        if (!ok && g_err.empty()) fail(UMPA_HIP_E_DEVICE, "frame upload failed: %s", hipGetErrorString(hipGetLastError()));
    } else if (ok) {
        for (int k = 0; k < Na; k++) { m->d_sam[k] = sam[k]; m->d_ref[k] = ref[k]; if (m->has_mask) m->d_mask[k] = mask[k]; }
    }
    if (ok) {
        std::vector<FrameDesc> desc(Na);
        for (int k = 0; k < Na; k++) {
            desc[k].sam = m->d_sam[k]; desc[k].ref = m->d_ref[k]; desc[k].mask = m->d_mask[k];
            desc[k].H = dims[2 * k]; desc[k].W = dims[2 * k + 1]; desc[k].pi = pos[2 * k]; desc[k].pj = pos[2 * k + 1];
        }
        ok = hipMalloc((void**)&m->d_desc, Na * sizeof(FrameDesc)) == hipSuccess &&
             hipMemcpy(m->d_desc, desc.data(), Na * sizeof(FrameDesc), hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) fail(UMPA_HIP_E_NOMEM, "cannot allocate the frame descriptor table");
    }
    if (ok) ok = upload_win(m, win, Nw) == 0;
    if (ok && m->has_mask && m->owns_frames && !getenv("UMPA_HIP_NO_BINARY_MASKS")) {
        // 0/1 masks let corr_masked form the pair weight with one multiply; owned frames only (borrowed ones may change)
        int* d_flag = nullptr;
        int h_flag = 1;
        size_t total = 0;
        for (int k = 0; k < Na; k++) total += (size_t)dims[2 * k] * dims[2 * k + 1];
        if (hipMalloc((void**)&d_flag, sizeof(int)) == hipSuccess) {
            if (hipMemcpy(d_flag, &h_flag, sizeof(int), hipMemcpyHostToDevice) == hipSuccess) {
                hipLaunchKernelGGL(umpa::mask_binary_kernel, dim3(2048), dim3(256), 0, m->stream, (const double*)m->d_mask[0], total, d_flag);
                if (hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
                    hipStreamSynchronize(m->stream) != hipSuccess) h_flag = 0;
                m->mask_binary = h_flag == 1;
            }
            (void)hipFree(d_flag);
        }
    }
    if (!ok) { umpa_hip_destroy(m); return nullptr; }
    return m;
}

int umpa_hip_update_frames(umpa_hip_model* m, double* const* sam, double* const* ref)
{
    if (!m) return fail(UMPA_HIP_E_ARG, "null model");
    if (!m->owns_frames) return fail(UMPA_HIP_E_ARG, "the model borrows device frames; update them in place instead");
    HIP_TRY(hipSetDevice(m->device), UMPA_HIP_E_DEVICE);
    if (ref) m->tiled.ref_maps_ok = false;
    for (int k = 0; k < m->Na; k++) {
        const size_t n = (size_t)m->dims[2 * k] * m->dims[2 * k + 1] * sizeof(double);
        if (sam) HIP_TRY(hipMemcpyAsync(m->d_sam[k], sam[k], n, hipMemcpyHostToDevice, m->stream), UMPA_HIP_E_DEVICE);
        if (ref) HIP_TRY(hipMemcpyAsync(m->d_ref[k], ref[k], n, hipMemcpyHostToDevice, m->stream), UMPA_HIP_E_DEVICE);
    }
    HIP_TRY(hipStreamSynchronize(m->stream), UMPA_HIP_E_DEVICE);
    return 0;
}

int umpa_hip_stage_sample(umpa_hip_model* m, const void* const* raw, int raw_dtype,
                          const double* const* dark, const double* const* flat)
{
    if (!m || !raw) return fail(UMPA_HIP_E_ARG, "null argument");
    if (!m->owns_frames) return fail(UMPA_HIP_E_ARG, "the model borrows device frames: write the next stack there yourself");
    if (raw_dtype < 0 || raw_dtype > 2) return fail(UMPA_HIP_E_ARG, "raw_dtype %d: 0 float64, 1 float32, 2 uint16", raw_dtype);
    HIP_TRY(hipSetDevice(m->device), UMPA_HIP_E_DEVICE);
    const size_t esz = raw_dtype == 0 ? 8 : raw_dtype == 1 ? 4 : 2;
    size_t total = 0;
    for (int k = 0; k < m->Na; k++) total += (size_t)m->dims[2 * k] * m->dims[2 * k + 1];
    if (!m->up_stream) HIP_TRY(hipStreamCreateWithFlags(&m->up_stream, hipStreamNonBlocking), UMPA_HIP_E_DEVICE);
    if (!m->ev_staged) HIP_TRY(hipEventCreateWithFlags(&m->ev_staged, hipEventDisableTiming), UMPA_HIP_E_DEVICE);
    if (!m->d_back_blob) {
        HIP_TRY(hipMalloc(&m->d_back_blob, total * sizeof(double) + 16), UMPA_HIP_E_NOMEM);
        m->d_sam_back.resize(m->Na);
        size_t off = 0;
        for (int k = 0; k < m->Na; k++) { m->d_sam_back[k] = (double*)m->d_back_blob + off; off += (size_t)m->dims[2 * k] * m->dims[2 * k + 1]; }
    }
    const bool convert = raw_dtype != 0 || dark || flat;
    if (convert && m->raw_cap < total * esz) {
        if (m->d_raw_blob) (void)hipFree(m->d_raw_blob);
        m->d_raw_blob = nullptr; m->raw_cap = 0;
        HIP_TRY(hipMalloc(&m->d_raw_blob, total * esz), UMPA_HIP_E_NOMEM);
        m->raw_cap = total * esz;
    }
    hipStream_t us = m->up_stream;
    // the back buffer was the sample stack until the last swap: wait for the last match that read it -- not for the one
    // that is running on the front buffer now, which is the point of the double buffer
    const int back = 1 - front_buffer(m);
    if (m->read_once[back]) HIP_TRY(hipStreamWaitEvent(us, m->ev_read_done[back], 0), UMPA_HIP_E_DEVICE);
    size_t off = 0;
    for (int k = 0; k < m->Na; k++) {
        const size_t n = (size_t)m->dims[2 * k] * m->dims[2 * k + 1];
        double* out = m->d_sam_back[k];
        if (!convert) {
            HIP_TRY(hipMemcpyAsync(out, raw[k], n * 8, hipMemcpyHostToDevice, us), UMPA_HIP_E_DEVICE);
        } else {
            char* dr = (char*)m->d_raw_blob + off * esz;
            HIP_TRY(hipMemcpyAsync(dr, raw[k], n * esz, hipMemcpyHostToDevice, us), UMPA_HIP_E_DEVICE);
            const double* dk = dark ? dark[k] : nullptr;
            const double* fl = flat ? flat[k] : nullptr;
            const unsigned grid = (unsigned)((n + 255) / 256);
            if (raw_dtype == 0) hipLaunchKernelGGL((umpa::flat_correct_kernel<double>), dim3(grid), dim3(256), 0, us, (const double*)dr, dk, fl, out, n);
            else if (raw_dtype == 1) hipLaunchKernelGGL((umpa::flat_correct_kernel<float>), dim3(grid), dim3(256), 0, us, (const float*)dr, dk, fl, out, n);
            else hipLaunchKernelGGL((umpa::flat_correct_kernel<unsigned short>), dim3(grid), dim3(256), 0, us, (const unsigned short*)dr, dk, fl, out, n);
            HIP_TRY(hipGetLastError(), UMPA_HIP_E_LAUNCH);
        }
        off += n;
    }
    HIP_TRY(hipEventRecord(m->ev_staged, us), UMPA_HIP_E_DEVICE);
    m->staged = true;
    return 0;
}

int umpa_hip_set_rows_callback(umpa_hip_model* m, umpa_hip_rows_fn fn, void* user, int piece_rows)
{
    if (!m) return fail(UMPA_HIP_E_ARG, "null model");
    m->rows_cb = fn; m->rows_user = user; m->rows_piece_rows = piece_rows > 0 ? piece_rows : 0;
    return 0;
}

int umpa_hip_wait(umpa_hip_model* m)
{
    if (!m) return fail(UMPA_HIP_E_ARG, "null model");
    HIP_TRY(hipSetDevice(m->device), UMPA_HIP_E_DEVICE);
    if (!m->pending.empty()) {                                        // the OLDEST asynchronous match: kernels and downloads
        umpa_hip_model::PendingMatch pm = m->pending.front();
        m->pending.erase(m->pending.begin());
        const hipError_t de = pm.done ? hipEventSynchronize(pm.done) : hipStreamSynchronize(m->copy_stream);
        for (auto e : pm.events) m->event_pool.push_back(e);
        if (pm.done) m->event_pool.push_back(pm.done);
        if (de != hipSuccess) return fail(UMPA_HIP_E_LAUNCH, "asynchronous match (kernels or download of the result maps): %s", hipGetErrorString(de));
        return UMPA_HIP_ST_OK;
    }
    const hipError_t se = hipStreamSynchronize(m->stream);
    const hipError_t ce = hipStreamSynchronize(m->copy_stream);
    if (se != hipSuccess) return fail(UMPA_HIP_E_LAUNCH, "match kernels: %s", hipGetErrorString(se));
    if (ce != hipSuccess) return fail(UMPA_HIP_E_DEVICE, "download of the result maps: %s", hipGetErrorString(ce));
    return UMPA_HIP_ST_OK;
}

int umpa_hip_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) return fail(UMPA_HIP_E_ARG, "null argument");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault), UMPA_HIP_E_NOMEM);
    return 0;
}

int umpa_hip_host_unregister(void* p)
{
    if (!p) return 0;
    HIP_TRY(hipHostUnregister(p), UMPA_HIP_E_DEVICE);
    return 0;
}

void umpa_hip_destroy(umpa_hip_model* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->copy_stream) (void)hipStreamSynchronize(m->copy_stream);
    for (auto& l : m->launches) { (void)hipEventDestroy(l.t0); (void)hipEventDestroy(l.t1); }
    for (auto e : m->event_pool) (void)hipEventDestroy(e);
    tiled_release(m->tiled);
    for (int q = 0; q < 2; q++) {
        m->b_values[q].release(); m->b_uv[q].release(); m->b_err[q].release(); m->b_cover[q].release();
        m->b_dd[q].release(); m->b_da[q].release(); m->b_dn[q].release();
    }
    m->b_covout.release(); m->b_small.release(); m->b_kern.release();
    m->t_values.release(); m->t_uv.release(); m->t_err.release(); m->t_cover.release(); m->t_dd.release(); m->t_da.release(); m->t_dn.release();
    if (m->d_desc) (void)hipFree(m->d_desc);
    if (m->d_sub) (void)hipFree(m->d_sub);
    if (m->d_win) (void)hipFree(m->d_win);
    if (m->d_frames_blob) (void)hipFree(m->d_frames_blob);
    if (m->d_back_blob) (void)hipFree(m->d_back_blob);
    if (m->d_raw_blob) (void)hipFree(m->d_raw_blob);
    if (m->h_desc) (void)hipHostFree(m->h_desc);
    if (m->ev_staged) (void)hipEventDestroy(m->ev_staged);
    for (int q = 0; q < 2; q++) if (m->ev_read_done[q]) (void)hipEventDestroy(m->ev_read_done[q]);
    for (auto& pm : m->pending) { for (auto e : pm.events) (void)hipEventDestroy(e); if (pm.done) (void)hipEventDestroy(pm.done); }
    if (m->up_stream) { (void)hipStreamSynchronize(m->up_stream); (void)hipStreamDestroy(m->up_stream); }
    if (m->stream) (void)hipStreamDestroy(m->stream);
    if (m->copy_stream) (void)hipStreamDestroy(m->copy_stream);
    delete m;
}

int umpa_hip_set_window(umpa_hip_model* m, const double* win, int Nw)
{
    if (!m || !win) return fail(UMPA_HIP_E_ARG, "null argument");
    if (Nw < 0) return fail(UMPA_HIP_E_ARG, "Nw must be non-negative.");           // Model.cpp:242
    // The reference does not re-derive the padding when the window grows (model.pyx:702-704) and
    // then reads outside the frames.  Refuse that instead of faulting the GPU.
    const int reach = m->kind == UMPA_HIP_KIND_DFKERNEL ? UMPA_BLUR_HALF : 0;
    if (Nw + m->ms + reach > m->padding)
        return fail(UMPA_HIP_E_ARG, "Nw=%d with max_shift=%d (+%d) exceeds the padding %d fixed at construction", Nw, m->ms, reach, m->padding);
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    return upload_win(m, win, Nw);
}

int umpa_hip_set_subpx(umpa_hip_model* m, int mode)
{
    if (!m) return fail(UMPA_HIP_E_ARG, "null model");
    m->subpx = mode;
    return 0;
}

int umpa_hip_set_reference_shift(umpa_hip_model* m, int v)
{
    if (!m) return fail(UMPA_HIP_E_ARG, "null model");
    m->ref_mode = v ? 1 : 0;
    return 0;
}

int umpa_hip_coverage_region(umpa_hip_model* m, int start0, int step0, int N0, int start1, int step1, int N1, double* out)
{
    if (!m || !out) return fail(UMPA_HIP_E_ARG, "null argument");
    if (int rc = check_region(m, start0, step0, N0, start1, step1, N1)) return rc;
    HIP_TRY(hipSetDevice(m->device), UMPA_HIP_E_DEVICE);
    const size_t n = (size_t)N0 * N1;
    if (m->b_covout.reserve(n * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "coverage buffer");
    dim3 blk(64, 4), grd((N1 + 63) / 64, (N0 + 3) / 4);
    {
        ScopedTimer t(m, m->stream, KN_COVER);
        hipLaunchKernelGGL(coverage_kernel, grd, blk, 0, m->stream, m->dev(), m->padding + start0, step0, N0,
                           m->padding + start1, step1, N1, (int)m->has_mask, (double*)m->b_covout.p);
    }
    HIP_TRY(hipGetLastError(), UMPA_HIP_E_LAUNCH);
    HIP_TRY(hipMemcpyAsync(out, m->b_covout.p, n * sizeof(double), hipMemcpyDeviceToHost, m->stream), UMPA_HIP_E_DEVICE);
    HIP_TRY(hipStreamSynchronize(m->stream), UMPA_HIP_E_DEVICE);
    return UMPA_HIP_ST_OK;
}

int umpa_hip_coverage(umpa_hip_model* m, double* out, int i, int j)
{
    if (!m || !out) return fail(UMPA_HIP_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(m->device), UMPA_HIP_E_DEVICE);
    if (m->b_small.reserve(64 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "scratch");
    hipLaunchKernelGGL(coverage_kernel, dim3(1), dim3(1), 0, m->stream, m->dev(), i, 1, 1, j, 1, 1, (int)m->has_mask, (double*)m->b_small.p);
    HIP_TRY(hipGetLastError(), UMPA_HIP_E_LAUNCH);
    HIP_TRY(hipMemcpyAsync(out, m->b_small.p, sizeof(double), hipMemcpyDeviceToHost, m->stream), UMPA_HIP_E_DEVICE);
    HIP_TRY(hipStreamSynchronize(m->stream), UMPA_HIP_E_DEVICE);
    return UMPA_HIP_ST_OK;
}

// Single-pixel calls accept any (i, j): a frame contributes only if the pixel lies `padding` inside it
// (the coverage test inside the kernels), which keeps every window read within that frame.
int umpa_hip_cost(umpa_hip_model* m, int i, int j, int si, int sj, double* values)
{
    if (!m || !values) return fail(UMPA_HIP_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(m->device), UMPA_HIP_E_DEVICE);
    if (m->b_small.reserve(512 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "scratch");
    double* d = (double*)m->b_small.p;
    const ModelDev dv = m->dev();
    const double ka = m->kind == 2 ? values[2] : 0.0, kb = m->kind == 2 ? values[3] : 0.0, kc = m->kind == 2 ? values[4] : 0.0;
#define UMPA_COST_LAUNCH(K, M) hipLaunchKernelGGL((cost_one_kernel<K, M>), 1, 1, 0, m->stream, dv, i, j, si, sj, d, ka, kb, kc)
    if (m->kind == 0) { if (m->has_mask) UMPA_COST_LAUNCH(0, true); else UMPA_COST_LAUNCH(0, false); }
    else if (m->kind == 1) { if (m->has_mask) UMPA_COST_LAUNCH(1, true); else UMPA_COST_LAUNCH(1, false); }
    else { if (m->has_mask) UMPA_COST_LAUNCH(2, true); else UMPA_COST_LAUNCH(2, false); }
#undef UMPA_COST_LAUNCH
    HIP_TRY(hipGetLastError(), UMPA_HIP_E_LAUNCH);
    double h[4];
    HIP_TRY(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, m->stream), UMPA_HIP_E_DEVICE);
    HIP_TRY(hipStreamSynchronize(m->stream), UMPA_HIP_E_DEVICE);
    const int st = (int)h[3];
    if (st & UMPA_HIP_ST_OK) {                 // on a bound error the reference leaves values[0] untouched
        values[0] = h[0];
        values[1] = h[1];
        if (m->kind == 1) values[2] = h[2];
    } else {
        values[1] = 0.0;                       // args.t / args.v as constructed (Model.cpp:538-541, :892-895)
        if (m->kind == 1) values[2] = 0.0;
    }
    return st;
}

int umpa_hip_match_region(umpa_hip_model* m, int start0, int step0, int N0, int start1, int step1, int N1,
                          double* values, int nparam, double* uv, int* err,
                          const double* covermap, double cover_threshold,
                          double* dbg_d, double* dbg_a, int* dbg_ncalls, int flags, void* stream)
{
    if (!m || !values || !err) return fail(UMPA_HIP_E_ARG, "null argument");
    if (nparam < (m->kind == 1 ? 5 : m->kind == 2 ? 7 : 4)) return fail(UMPA_HIP_E_ARG, "nparam=%d too small for this model", nparam);
    if (int rc = check_region(m, start0, step0, N0, start1, step1, N1)) return rc;
    HIP_TRY(hipSetDevice(m->device), UMPA_HIP_E_DEVICE);
    const size_t n = (size_t)N0 * N1;

    RegionArgs A;
    A.org0 = m->padding + start0; A.step0 = step0; A.N0 = N0;       // model.pyx:482-483
    A.org1 = m->padding + start1; A.step1 = step1; A.N1 = N1; A.pitch = N1;
    A.nparam = nparam; A.thr = cover_threshold; A.kern = nullptr; A.kern_stride = 0; A.row_base = 0; A.blur = nullptr; A.blur_F = 0; A.blur_ready = 0;
    const bool planar = (flags & UMPA_HIP_F_PLANAR) != 0;
    A.v_px = planar ? 1 : (size_t)nparam; A.v_k = planar ? n : 1;

    if (flags & UMPA_HIP_F_DEVICE_IO) {
        A.values = values; A.uv = uv; A.err = err; A.cover = covermap;
        A.dbg_d = dbg_d; A.dbg_a = dbg_a; A.dbg_n = dbg_ncalls;
        if (int rc = adopt_staged(m, flags, (hipStream_t)stream)) return rc;
        if (m->rows_cb) {
            umpa_hip_rows_fn cb = m->rows_cb;
            void* user = m->rows_user;
            if (int rc = run_match(m, A, flags, (hipStream_t)stream, m->rows_piece_rows, [cb, user](int lo, int hi) { cb(lo, hi, user); })) return rc;
        } else if (int rc = run_match(m, A, flags, (hipStream_t)stream)) return rc;
        mark_matched(m, (hipStream_t)stream);
        return UMPA_HIP_ST_OK;
    }

    hipStream_t s = m->stream;
    if ((flags & UMPA_HIP_F_ASYNC) && m->pending.size() >= 2)
        return fail(UMPA_HIP_E_ARG, "two asynchronous matches are in flight already: umpa_hip_wait() for the older one first");
    if (!(flags & UMPA_HIP_F_ASYNC) && !m->pending.empty())
        return fail(UMPA_HIP_E_ARG, "an asynchronous match is in flight: umpa_hip_wait() first");
    const int os = m->out_set;                                        // this match's device output set
    m->out_set ^= 1;
    if (int rc = adopt_staged(m, flags, s)) return rc;
    if (m->b_values[os].reserve(n * nparam * sizeof(double)) || m->b_err[os].reserve(n * sizeof(int)))
        return fail(UMPA_HIP_E_NOMEM, "output buffers (%zu pixels)", n);
    // values and err start from the caller's arrays (zeros in the reference, model.pyx:455,468).  They only matter
    // where the kernel reads them (DFKernel's a,b,c) or leaves them alone (pixels skipped by the coverage test):
    // otherwise every element is overwritten and the upload is skipped.
    if (m->kind == UMPA_HIP_KIND_DFKERNEL || covermap) {
        HIP_TRY(hipMemcpyAsync(m->b_values[os].p, values, n * nparam * sizeof(double), hipMemcpyHostToDevice, s), UMPA_HIP_E_DEVICE);
        HIP_TRY(hipMemcpyAsync(m->b_err[os].p, err, n * sizeof(int), hipMemcpyHostToDevice, s), UMPA_HIP_E_DEVICE);
    }
    A.values = (double*)m->b_values[os].p; A.err = (int*)m->b_err[os].p;
    A.uv = nullptr; A.cover = nullptr; A.dbg_d = nullptr; A.dbg_a = nullptr; A.dbg_n = nullptr;
    if (uv) {
        if (m->b_uv[os].reserve(n * 2 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "uv buffer");
        HIP_TRY(hipMemcpyAsync(m->b_uv[os].p, uv, n * 2 * sizeof(double), hipMemcpyHostToDevice, s), UMPA_HIP_E_DEVICE);
        A.uv = (double*)m->b_uv[os].p;
    }
    if (covermap) {
        if (m->b_cover[os].reserve(n * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "coverage buffer");
        HIP_TRY(hipMemcpyAsync(m->b_cover[os].p, covermap, n * sizeof(double), hipMemcpyHostToDevice, s), UMPA_HIP_E_DEVICE);
        A.cover = (const double*)m->b_cover[os].p;
    }
    // The debug maps are written for every pixel the kernels visit: they only need clearing where the coverage test
    // may skip pixels (1.4 GB of memset per C2 match otherwise).
    const bool clear_dbg = covermap != nullptr;
    if (dbg_d) { if (m->b_dd[os].reserve(n * 25 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "debug_d buffer");
                 if (clear_dbg) HIP_TRY(hipMemsetAsync(m->b_dd[os].p, 0, n * 25 * sizeof(double), s), UMPA_HIP_E_DEVICE);
                 A.dbg_d = (double*)m->b_dd[os].p; }
    if (dbg_a) { if (m->b_da[os].reserve(n * 16 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "debug_a buffer");
                 if (clear_dbg) HIP_TRY(hipMemsetAsync(m->b_da[os].p, 0, n * 16 * sizeof(double), s), UMPA_HIP_E_DEVICE);
                 A.dbg_a = (double*)m->b_da[os].p; }
    if (dbg_ncalls) { if (m->b_dn[os].reserve(n * sizeof(int))) return fail(UMPA_HIP_E_NOMEM, "debug_Ncalls buffer");
                      if (clear_dbg) HIP_TRY(hipMemsetAsync(m->b_dn[os].p, 0, n * sizeof(int), s), UMPA_HIP_E_DEVICE);
                      A.dbg_n = (int*)m->b_dn[os].p; }

    // The region is matched in a few row chunks; all kernels are enqueued first (each chunk leaves an event on the
    // compute stream), then the rows of chunk c travel to the host on the copy stream while chunk c+1 is still
    // being matched.  Into pinned host memory (umpa_hip_host_alloc) that is a DMA at PCIe rate; into pageable memory
    // the runtime stages it, still overlapped with the compute.
    struct Piece { int lo, hi; hipEvent_t done; };
    std::vector<Piece> piece_list;
    auto on_rows = [&](int lo, int hi) {
        Piece p = {lo, hi, get_event(m)};
        if (p.done) (void)hipEventRecord(p.done, s);
        piece_list.push_back(p);
    };
    int pieces = n >= ((size_t)1 << 21) ? 8 : n >= ((size_t)1 << 19) ? 4 : 1;
    // an asynchronous match overlaps its download with the NEXT match (two in flight): fewer, larger pieces are cheaper there
    // (step-scan series of 2048^2 x 5 frames: 8 pieces 4.09, 4: 3.86, 2: 3.76, 1: 3.79 ms per projection)
    if ((flags & UMPA_HIP_F_ASYNC) && pieces > 2) pieces = 2;
    { static const char* pe = getenv("UMPA_HIP_PIECES"); if (pe && atoi(pe) > 0) pieces = atoi(pe); }   // tuning override
    const int N0d = step0 * (N0 - 1) + 1;
    const int piece_rows = pieces > 1 ? std::max(4 * UMPA_TILE, (N0d + pieces - 1) / pieces) : 0;
    if (int rc = run_match(m, A, flags, s, piece_rows, on_rows)) return rc;
    mark_matched(m, s);

    hipStream_t cs = m->copy_stream;
    hipError_t ce = hipSuccess;
    auto down = [&](void* host, const void* dev, size_t elem_bytes, size_t per_px, int lo, int hi) {
        if (ce != hipSuccess || !host) return;
        const size_t off = (size_t)lo * N1 * per_px * elem_bytes, bytes = (size_t)(hi - lo) * N1 * per_px * elem_bytes;
        ce = hipMemcpyAsync((char*)host + off, (const char*)dev + off, bytes, hipMemcpyDeviceToHost, cs);
    };
    for (const Piece& p : piece_list) {
        if (p.done) { if (ce == hipSuccess) ce = hipStreamWaitEvent(cs, p.done, 0); }
        else if (ce == hipSuccess) ce = hipStreamSynchronize(s);
        if (planar) for (int k = 0; k < nparam; k++) down(values + (size_t)k * n, A.values + (size_t)k * n, sizeof(double), 1, p.lo, p.hi);
        else down(values, A.values, sizeof(double), nparam, p.lo, p.hi);
        down(err, A.err, sizeof(int), 1, p.lo, p.hi);
        down(uv, A.uv, sizeof(double), 2, p.lo, p.hi);
        down(dbg_d, A.dbg_d, sizeof(double), 25, p.lo, p.hi);
        down(dbg_a, A.dbg_a, sizeof(double), 16, p.lo, p.hi);
        down(dbg_ncalls, A.dbg_n, sizeof(int), 1, p.lo, p.hi);
    }
    if ((flags & UMPA_HIP_F_ASYNC) && ce == hipSuccess) {          // the caller collects the result with umpa_hip_wait
        umpa_hip_model::PendingMatch pm;
        for (const Piece& p : piece_list) if (p.done) pm.events.push_back(p.done);
        pm.done = get_event(m);
        if (pm.done) (void)hipEventRecord(pm.done, cs);              // behind the last download of this match
        m->pending.push_back(pm);
        return UMPA_HIP_ST_OK;
    }
    if (ce == hipSuccess) ce = hipStreamSynchronize(cs);
    const hipError_t se = hipStreamSynchronize(s);
    for (const Piece& p : piece_list) if (p.done) m->event_pool.push_back(p.done);
    if (ce != hipSuccess) return fail(UMPA_HIP_E_DEVICE, "download of the result maps: %s", hipGetErrorString(ce));
    if (se != hipSuccess) return fail(UMPA_HIP_E_LAUNCH, "match kernels: %s", hipGetErrorString(se));
    return UMPA_HIP_ST_OK;
}

int umpa_hip_min(umpa_hip_model* m, int i, int j, double* values, double* uv, double* dbg_d, double* dbg_a, int* ncalls)
{
    if (!m || !values) return fail(UMPA_HIP_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(m->device), UMPA_HIP_E_DEVICE);
    const int np = m->kind == 1 ? 5 : m->kind == 2 ? 7 : 4;
    if (m->b_small.reserve(512 * sizeof(double))) return fail(UMPA_HIP_E_NOMEM, "scratch");
    double* d = (double*)m->b_small.p;                 // [0..7) values, [8..9] uv, [10] err(int), [11] ncalls(int), [12..37) d, [37..53) a
    double h[53];
    memset(h, 0, sizeof(h));
    for (int q = 0; q < np; q++) h[q] = values[q];
    if (uv) { h[8] = uv[0]; h[9] = uv[1]; }
    hipStream_t s = m->stream;
    HIP_TRY(hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, s), UMPA_HIP_E_DEVICE);
    RegionArgs A;
    A.org0 = i; A.step0 = 1; A.N0 = 1; A.org1 = j; A.step1 = 1; A.N1 = 1; A.pitch = 1;      // Model::min takes absolute coordinates
    A.values = d; A.nparam = np; A.v_px = np; A.v_k = 1; A.uv = d + 8; A.err = (int*)(d + 10); A.cover = nullptr; A.thr = 0.0;
    A.dbg_n = (int*)(d + 11); A.dbg_d = d + 12; A.dbg_a = d + 37; A.kern = nullptr; A.kern_stride = 0; A.row_base = 0; A.blur = nullptr; A.blur_F = 0; A.blur_ready = 0;
    if (int rc = run_direct(m, A, s)) return rc;
    HIP_TRY(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s), UMPA_HIP_E_DEVICE);
    HIP_TRY(hipStreamSynchronize(s), UMPA_HIP_E_DEVICE);
    for (int q = 0; q < np; q++) values[q] = h[q];
    if (uv) { uv[0] = h[8]; uv[1] = h[9]; }
    int e, nc;
    memcpy(&e, &h[10], sizeof(int));
    memcpy(&nc, &h[11], sizeof(int));
    if (ncalls) *ncalls = nc;
    if (dbg_d) memcpy(dbg_d, h + 12, 25 * sizeof(double));
    if (dbg_a) memcpy(dbg_a, h + 37, 16 * sizeof(double));
    return e;          // only the ok bit survives the region kernel, as in the Cython loop (model.pyx:487)
}

static int spfit(int device, const double* a16, double* pos2, double* value, int quad)
{
    if (!a16 || !pos2 || !value) return fail(UMPA_HIP_E_ARG, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(UMPA_HIP_E_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device), UMPA_HIP_E_DEVICE);
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 19 * sizeof(double)), UMPA_HIP_E_NOMEM);
    double h[19];
    memcpy(h, a16, 16 * sizeof(double));
    h[16] = pos2[0]; h[17] = pos2[1]; h[18] = 0.0;
    hipError_t e = hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(spfit_kernel, 1, 1, 0, 0, d, d + 16, quad); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(UMPA_HIP_E_LAUNCH, "spfit: %s", hipGetErrorString(e));
    pos2[0] = h[16]; pos2[1] = h[17]; *value = h[18];
    return UMPA_HIP_ST_OK;
}

int umpa_hip_spmin(int device, const double* a16, double* pos2, double* value) { return spfit(device, a16, pos2, value, 0); }
int umpa_hip_spmin_quad(int device, const double* a16, double* pos2, double* value) { return spfit(device, a16, pos2, value, 1); }

int umpa_hip_correct_bad_pixels(const double* in, double* out, long nimg, int H, int W, int ndims,
                                double lo, double hi, int iterations, int device, int flags, void* stream)
{
    if (!in || !out) return fail(UMPA_HIP_E_ARG, "null argument");
    if (nimg < 1 || H < 1 || W < 2 || (ndims == 2 && H < 2) || (ndims != 1 && ndims != 2) || iterations < 0)
        return fail(UMPA_HIP_E_ARG, "correct_bad_pixels: bad shape %ld x %d x %d, ndims %d, iterations %d", nimg, H, W, ndims, iterations);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(UMPA_HIP_E_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device), UMPA_HIP_E_DEVICE);
    hipStream_t s = (hipStream_t)stream;
    const bool dev_io = (flags & UMPA_HIP_F_DEVICE_IO) != 0;
    const size_t n = (size_t)nimg * H * W, bytes = n * sizeof(double);
    // scratch: [a][b] ping-pong images + the mask; with device I/O `out` is one of the two images
    double *a = nullptr, *b = nullptr;
    unsigned char* bad = nullptr;
    void* blob = nullptr;
    const size_t need = (dev_io ? bytes : 2 * bytes) + n;
    if (hipMalloc(&blob, need) != hipSuccess) return fail(UMPA_HIP_E_NOMEM, "correct_bad_pixels: %zu bytes of scratch", need);
    hipError_t e = hipSuccess;
    if (dev_io) {
        a = out; b = (double*)blob; bad = (unsigned char*)blob + bytes;
        if (a != in) e = hipMemcpyAsync(a, in, bytes, hipMemcpyDeviceToDevice, s);
    } else {
        a = (double*)blob; b = a + n; bad = (unsigned char*)blob + 2 * bytes;
        e = hipMemcpyAsync(a, in, bytes, hipMemcpyHostToDevice, s);
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (e == hipSuccess) { hipLaunchKernelGGL(umpa::badpix_mark_kernel, dim3(grid), dim3(256), 0, s, a, bad, n, lo, hi); e = hipGetLastError(); }
    double *src = a, *dst = b;
    for (int it = 0; it < iterations && e == hipSuccess; it++) {
        hipLaunchKernelGGL(umpa::badpix_pass_kernel, dim3(grid), dim3(256), 0, s, src, dst, bad, (size_t)nimg, H, W, ndims);
        e = hipGetLastError();
        std::swap(src, dst);
    }
    // the result is in `src`
    if (e == hipSuccess) {
        if (dev_io) { if (src != out) e = hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToDevice, s); }
        else e = hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);                 // the scratch is freed below
    (void)hipFree(blob);
    if (e != hipSuccess) return fail(UMPA_HIP_E_LAUNCH, "correct_bad_pixels: %s", hipGetErrorString(e));
    return 0;
}

/* ---- pinned host memory for result maps (and inputs): a small pool, because pinning is slow (page-locking) and a
 * caller that matches in a loop asks for the same sizes again and again */
namespace {
struct HostBlock { void* p; size_t cap; };
std::mutex g_host_mu;
std::vector<HostBlock> g_host_free;                 // blocks handed back, kept pinned
std::vector<HostBlock> g_host_live;
size_t g_host_cached = 0;
const size_t HOST_CACHE_LIMIT = (size_t)8 << 30;
}

void* umpa_hip_host_alloc(size_t bytes)
{
    if (bytes == 0) bytes = 1;
    std::lock_guard<std::mutex> lock(g_host_mu);
    size_t best = g_host_free.size();
    for (size_t q = 0; q < g_host_free.size(); q++)
        if (g_host_free[q].cap >= bytes && g_host_free[q].cap <= bytes + bytes / 4 + 4096 &&
            (best == g_host_free.size() || g_host_free[q].cap < g_host_free[best].cap)) best = q;
    HostBlock b;
    if (best < g_host_free.size()) {
        b = g_host_free[best];
        g_host_free.erase(g_host_free.begin() + best);
        g_host_cached -= b.cap;
    } else {
        b.cap = (bytes + 4095) & ~(size_t)4095;
        if (hipHostMalloc(&b.p, b.cap, hipHostMallocDefault) != hipSuccess) {
            fail(UMPA_HIP_E_NOMEM, "cannot pin %zu bytes of host memory", b.cap);
            return nullptr;
        }
    }
    g_host_live.push_back(b);
    return b.p;
}

void umpa_hip_host_free(void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_host_mu);
    for (size_t q = 0; q < g_host_live.size(); q++)
        if (g_host_live[q].p == p) {
            HostBlock b = g_host_live[q];
            g_host_live.erase(g_host_live.begin() + q);
            if (g_host_cached + b.cap <= HOST_CACHE_LIMIT) { g_host_free.push_back(b); g_host_cached += b.cap; }
            else (void)hipHostFree(b.p);
            return;
        }
}

void umpa_hip_host_trim(void)
{
    std::lock_guard<std::mutex> lock(g_host_mu);
    for (auto& b : g_host_free) (void)hipHostFree(b.p);
    g_host_free.clear();
    g_host_cached = 0;
}

int umpa_hip_timing_enable(umpa_hip_model* m, int on)
{
    if (!m) return fail(UMPA_HIP_E_ARG, "null model");
    m->timing = on != 0;
    return 0;
}

/* Synchronise all recorded launches, fold them into per-kernel totals and return the number of distinct kernels. */
int umpa_hip_timing_collect(umpa_hip_model* m)
{
    if (!m) return fail(UMPA_HIP_E_ARG, "null model");
    (void)hipSetDevice(m->device);
    m->tnames.clear(); m->tms.clear(); m->tcount.clear(); m->tfma.clear();
    for (auto& l : m->launches) {
        float ms = 0.f;
        if (hipEventSynchronize(l.t1) == hipSuccess && hipEventElapsedTime(&ms, l.t0, l.t1) == hipSuccess) {
            const char* nm = KERNEL_NAMES[l.name];
            size_t q = 0;
            for (; q < m->tnames.size(); q++) if (m->tnames[q] == nm) break;
            if (q == m->tnames.size()) { m->tnames.push_back(nm); m->tms.push_back(0.0); m->tcount.push_back(0); m->tfma.push_back(0.0); }
            // (the event wait above is behind the copy of the chunk's counters in stream order: counts[3] = passes computed)
            m->tms[q] += ms; m->tcount[q] += 1; m->tfma[q] += l.fma + (l.counts ? l.fma_per * l.counts[3] : 0.0);
        }
        m->event_pool.push_back(l.t0); m->event_pool.push_back(l.t1);
    }
    m->launches.clear();
    return (int)m->tnames.size();
}

int umpa_hip_timing_read(umpa_hip_model* m, int index, const char** name, double* total_ms, int* launches)
{
    if (!m || index < 0 || index >= (int)m->tnames.size()) return fail(UMPA_HIP_E_ARG, "bad timing index");
    if (name) *name = m->tnames[index].c_str();
    if (total_ms) *total_ms = m->tms[index];
    if (launches) *launches = m->tcount[index];
    return 0;
}

int umpa_hip_timing_fma(umpa_hip_model* m, int index, double* fma)
{
    if (!m || !fma || index < 0 || index >= (int)m->tfma.size()) return fail(UMPA_HIP_E_ARG, "bad timing index");
    *fma = m->tfma[index];
    return 0;
}

int umpa_hip_last_path(umpa_hip_model* m) { return m ? m->last_path : 0; }

int umpa_hip_last_stats(umpa_hip_model* m, double* out4)
{
    if (!m || !out4) return fail(UMPA_HIP_E_ARG, "null argument");
    (void)hipSetDevice(m->device);
    // the counters were copied behind the match's kernels on the match's stream; the caller has synchronised it (or the
    // match was a host-array call, which synchronises itself): a device-wide wait makes this safe for either
    HIP_TRY(hipDeviceSynchronize(), UMPA_HIP_E_DEVICE);
    const TiledState& st = m->tiled;
    double done = 0.0, parked = 0.0, missed = 0.0;
    for (int q = 0; q < st.stat_n; q++) {
        done += st.stat_slots[q][OD_C_DONE];
        missed += st.stat_slots[q][OD_C_TILES];                          // tiles whose prediction fell short (stage 0)
        for (int r = 0; r + 1 < OD_STAGES; r++) parked += st.stat_slots[q][8 * r + OD_C_PX];   // walks started again, all rounds
    }
    out4[0] = st.stat_n ? done : st.stat_total_passes;
    out4[1] = st.stat_total_passes;
    out4[2] = parked;
    out4[3] = missed;
    return 0;
}

} // extern "C"

// umpa_ondemand.h -- only the passes of the shift table that some walk reads are computed.
//
// The exhaustive table has (2 max_shift - 1)^2 planes, but a pixel's walk (Optim.cpp:233-479) visits ~18 shifts and
// the pixels of a 32 x 32 tile together a compact blob of them: on BASELINE config C2 about 5-6 of the 9 row offsets,
// on C3 (15 x 15 shifts) well under half.  A table plane is produced per (tile, pass) -- a pass is one row offset and
// one batch of column offsets, exactly the unit of work of a corr_volume / corr_masked workgroup -- so whole
// workgroups can be left out.  Which ones is PREDICTED, and every prediction error is repaired, so the maps are those
// of the exhaustive table bit for bit:
//
//   1. seed tiles (every OD_SP-th tile in both directions): all passes are computed (a compact static grid), their
//      pixels are replayed, and every lane records the passes it read (visited[tile], one bit per pass);
//   2. predict: a tile between seed tiles needs what the seed tiles around it visited; those (tile, pass) pairs go onto a
//      work list that a persistent grid of table workgroups goes through;
//   3. the other pixels are replayed; a walk that asks for a pass that is not there stops: its pixel goes onto the parked
//      list, its tile onto the missed list, and the pass (with its neighbours in the row-offset direction) into the
//      tile's request mask;
//   4. repair rounds (persistent grids that read the list lengths on the device: no host round trip): the requested
//      passes are computed, the parked pixels' walks run again from the start and may park again; the last round
//      computes every pass the missed tiles still lack, so nothing can be left over.
// A table entry a walk is allowed to read is the number the exhaustive table holds; the rest is unavailable, never wrong.
//
// Per tile: done / visited are 64-bit masks (on-demand is used while a tile has at most 64 passes).  Every stage has
// its own counters (cleared once per row chunk), so nothing is reset between the launches of a chunk.
#pragma once
#include "umpa_walk.h"

namespace umpa {

#define OD_SP 4                       // seed-tile spacing
#define OD_ROUNDS 3                   // repair rounds; the last one computes everything that is still missing
#define OD_SAMPLE_ROUNDS 3            // corr_march's schedule (od_run_chunk_march): rounds of the sample lattice
#define OD_STAGES 12                  // counter blocks of a chunk (>= OD_ROUNDS + 1 and >= OD_SAMPLE_ROUNDS + OD_ROUNDS + 3)
// counters: int[8] per stage, stage 0 = steps 1-3, stage r = repair round r.
#define OD_C_TILES  0                 // missed tiles listed by the replay of this stage
#define OD_C_PX     1                 // pixels parked by the replay of this stage
#define OD_C_ITEMS  2                 // work items of this stage's table launch
#define OD_C_DONE   3                 // (stage 0 only) passes computed by all stages

struct OdArgs {
    unsigned long long* done;         // [ntiles] passes whose table planes are valid for the tile
    unsigned long long* visited;      // [ntiles] seed tiles: passes their pixels read; other tiles: passes their parked pixels ask for
    int* tile_flag;                   // [ntiles] 1 while the tile is on the missed list being written
    int* tile_in;  int* tile_out;     // missed tiles: read by the list kernel / written by the replay
    int* px_in;    int* px_out;       // parked pixels (region pixel numbers): read by the queue replay / written by the replay
    int* items;                       // work list of the table kernels in queue mode: (tile << 8) | pass
    int* cnt_in;   int* cnt_out;      // counters of the previous stage (lengths of tile_in, px_in) / of this stage
    int* cnt0;                        // stage 0's counters (OD_C_DONE)
    int mode;                         // table kernels: 0 every pass of every tile (static grid), 3 every pass of the seed tiles (compact static
                                      // grid), 2 persistent grid over `items`
                                      // replay kernels: 0 no on-demand, 1 seed-tile pixels (compact grid; record visited), 2 the other pixels
                                      // (park on a miss), 3 persistent grid over px_in (park on a miss)
    int tc, ub, nbatch, npass, ntx, nty;   // pass / tile geometry of the table kernel
    int tr;                           // rows of a tile: 32 (corr_volume, corr_masked) or corr_march's band height (a tile = strip x band)
    int sub;                          // replay mode 2: > 1 = only the pixels of a lattice, every sub-th in both directions (corr_march's
                                      // sample stage); 0, 1: every pixel
    unsigned long long central;       // od_run_chunk_lattice: the passes every tile gets before any walk (those around shift (0, 0))
    int alone;                        // 1: a parked pixel asks for the pass it missed alone, 0: and for its neighbours in the row-offset direction
    int r0, c0;                       // seed tiles: ty % OD_SP == r0 and tx % OD_SP == c0
    int ub_inv;                       // ceil(2^16 / ub): x / ub = (x * ub_inv) >> 16 for 0 <= x < 70
    int nrow_inv;                     // the same for the row offsets per pass
    int nearest;                      // prediction: 1 what the nearest seed tile visited, 0 the union over the (up to four) seed tiles around
};

// what the table kernels need of it (their SGPRs are nearly all taken by the window taps)
struct OdCorr {
    unsigned long long* done;
    const int* items;
    const int* nitems;                // -> length of `items`
    int* ndone;                       // -> passes computed
    int mode;                         // static-grid kernels: 0 every pass of every tile, 3 every pass of the seed tiles; queue kernels: 2
    int nsx, nseed, r0, c0;           // seed tiles: nsx per tile row, nseed in all, first at (r0, c0), OD_SP apart
};

__host__ __device__ inline int od_seed_count(int n, int first) { return n > first ? (n - first + OD_SP - 1) / OD_SP : 0; }   // seed rows (columns) among n

__device__ __forceinline__ bool od_is_seed(const OdArgs& od, int ty, int tx)
{
    return ty % OD_SP == od.r0 && tx % OD_SP == od.c0;
}

// the pass that holds table slot (si, sj): table kernels fill slot sigma * (offset), offsets counted from -(ms - 1).
// (24-bit multiplies: this runs once per cost evaluation of every walk, and 32-bit integer multiplies are quarter rate)
__device__ __forceinline__ int od_pass_of(const OdArgs& od, int ms, int sigma, int si, int sj)
{
    const int oi = (sigma > 0 ? si : -si) + ms - 1, oj = (sigma > 0 ? sj : -sj) + ms - 1;
    return __mul24(__mul24(oi, od.nrow_inv) >> 16, od.nbatch) + (__mul24(oj, od.ub_inv) >> 16);
}

// the passes in `mask` of tile `lin` onto the work list (the passes of a tile stay together: they share their patches in L2)
__device__ __forceinline__ void od_append(const OdArgs& od, int lin, unsigned long long mask)
{
    if (!mask) return;
    int at = atomicAdd(od.cnt_out + OD_C_ITEMS, __popcll(mask));
    while (mask) {
        const int p = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        gpw(od.items)[at++] = (lin << 8) | p;
    }
}

// what = 4: the central passes of every tile (od_run_chunk_lattice).
// what = 1, step 2: the predicted passes of the tiles that are not seed tiles, onto the work list.
// what = 2, a repair round: the passes the missed tiles' parked pixels asked for; what = 3, the last round: every pass they lack.
__global__ void __launch_bounds__(256)
od_list_kernel(OdArgs od, int what)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long all = od.npass >= 64 ? ~0ull : (1ull << od.npass) - 1;
    if (what == 4) {                                                  // od_run_chunk_lattice, stage 0: the central passes of every tile
        if (q < od.ntx * od.nty) od_append(od, q, od.central & all);
        return;
    }
    if (what >= 2) {
        if (q >= gp(od.cnt_in)[OD_C_TILES]) return;
        const int lin = gp(od.tile_in)[q];
        const unsigned long long lack = all & ~gp(od.done)[lin];
        od_append(od, lin, what == 3 ? lack : (gp(od.visited)[lin] & lack));
        gpw(od.visited)[lin] = 0;
        gpw(od.tile_flag)[lin] = 0;
        return;
    }
    if (q >= od.ntx * od.nty) return;
    const int ty = q / od.ntx, tx = q % od.ntx;
    if (od_is_seed(od, ty, tx)) return;
    unsigned long long nd = 0;
    // the seed rows around ty: the largest sy <= ty with sy % OD_SP == r0, and the next one
    const int sy0 = ty - ((ty - od.r0) % OD_SP + OD_SP) % OD_SP, sx0 = tx - ((tx - od.c0) % OD_SP + OD_SP) % OD_SP;
    int best = 1 << 30;
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            const int sy = sy0 + a * OD_SP, sx = sx0 + b * OD_SP;
            if (sy < 0 || sy >= od.nty || sx < 0 || sx >= od.ntx) continue;
            const unsigned long long v = gp(od.visited)[sy * od.ntx + sx];
            if (!od.nearest) nd |= v;
            else {
                const int d = max(abs(sy - ty), abs(sx - tx)) * 8 + abs(sy - ty) + abs(sx - tx);
                if (d < best) { best = d; nd = v; }
            }
        }
    od_append(od, q, nd & all);
}

// The work item of workgroup `b` of `g` in its j-th round: the list is cut into 8 contiguous ranges, one per XCD
// (blocks b, b + 8, ... share an XCD), so that the passes of a tile, neighbours on the list, run on one XCD at the same time.
__device__ __forceinline__ int od_item_index(int nitems, int b, int g, int j)
{
    const int x = b & 7, g8 = g >> 3;
    const long lo = (long)nitems * x / 8, hi = (long)nitems * (x + 1) / 8;
    const long idx = lo + (b >> 3) + (long)j * g8;
    return idx < hi ? (int)idx : -1;
}

// mode 1 of the replay kernels: a compact grid over the seed tiles, blockIdx.y = seed tile, blockIdx.x * 64 + lane = position
// inside the tile (row-major, od.tc columns).  False where the position is no pixel of the region (stepped regions use
// every step-th dense position).
__device__ __forceinline__ bool od_seed_pixel(const OdArgs& od, int drow0, int step0, int step1, int& xi, int& xj)
{
    const int nsx = od_seed_count(od.ntx, od.c0);
    const int t = blockIdx.y, idx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dr = (od.r0 + (t / nsx) * OD_SP) * 32 + idx / od.tc + drow0;      // dense row / column of the region
    const int dc = (od.c0 + (t % nsx) * OD_SP) * od.tc + idx % od.tc;
    xi = dr / step0; xj = dc / step1;
    return idx < 32 * od.tc && xi * step0 == dr && xj * step1 == dc;
}

// ---- per-lane bookkeeping of the replay kernels
struct OdLane {
    unsigned long long avail;         // passes whose planes are valid for this pixel's tile
    unsigned long long vis;           // passes this lane has read
    int lin, miss_pass;
    bool miss;
};

// false: this pixel is not this launch's business (seed-tile pixels belong to mode 1, the others to mode 2)
__device__ __forceinline__ bool od_begin(const OdArgs& od, int dense_row, int dense_col, OdLane& L)
{
    L.avail = ~0ull; L.vis = 0; L.lin = 0; L.miss = false; L.miss_pass = 0;
    if (od.mode == 0) return true;
    const int ty = od.tr == 32 ? dense_row >> 5 : dense_row / od.tr, tx = dense_col / od.tc;          // (once per pixel)
    L.lin = ty * od.ntx + tx;
    if (od.mode == 2 && od_is_seed(od, ty, tx)) return false;
    L.avail = gp(od.done)[L.lin];
    return true;
}

// May the walk read table slot (si, sj)?  Out-of-range shifts pass (the evaluation reports the bound error itself).
__device__ __forceinline__ bool od_check(const OdArgs& od, OdLane& L, int ms, int sigma, int si, int sj)
{
    if (od.mode == 0 || si <= -ms || si >= ms || sj <= -ms || sj >= ms) return true;
    const int pass = od_pass_of(od, ms, sigma, si, sj);
    const unsigned half = pass < 32 ? (unsigned)L.avail : (unsigned)(L.avail >> 32);      // (32-bit shifts: 64-bit ones are slow)
    if (!((half >> (pass & 31)) & 1u)) { L.miss = true; L.miss_pass = pass; return false; }
    if (od.mode == 1) L.vis |= 1ull << pass;                          // only the seed tiles' pixels keep track
    return true;
}

// mode 1: the passes the lanes of this wave read, into visited[tile].  Called by ALL 64 lanes of the wave (lanes
// without a pixel pass live = false); a wave of the compact grid lies inside one seed tile.
__device__ __forceinline__ void od_record_visited(const OdArgs& od, const OdLane& L, bool live)
{
    unsigned lo = live ? (unsigned)L.vis : 0u, hi = live ? (unsigned)(L.vis >> 32) : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo |= __shfl_xor(lo, off);
        hi |= __shfl_xor(hi, off);
    }
    const unsigned long long who = __ballot(live);
    if (live && (int)(threadIdx.x & 63) == __ffsll((long long)who) - 1 && (lo | hi))
        atomicOr(od.visited + L.lin, ((unsigned long long)hi << 32) | lo);
}

// modes 2, 3: the walk asked for a pass that is not there: the pixel waits for the next repair round, which computes that
// pass and the same column batch one row offset up and down (a walk moves on by single steps, and its 4 x 4 gather
// reaches one or two rows further)
__device__ __forceinline__ void od_park(const OdArgs& od, const OdLane& L, int px)
{
    gpw(od.px_out)[atomicAdd(od.cnt_out + OD_C_PX, 1)] = px;
    unsigned long long want = 1ull << L.miss_pass;
    if (od.alone == 0) {                                             // (corr_march's passes are several row offsets deep already)
        if (L.miss_pass >= od.nbatch) want |= 1ull << (L.miss_pass - od.nbatch);
        if (L.miss_pass + od.nbatch < od.npass) want |= 1ull << (L.miss_pass + od.nbatch);
    }
    atomicOr(od.visited + L.lin, want);
    if (atomicExch(od.tile_flag + L.lin, 1) == 0) gpw(od.tile_out)[atomicAdd(od.cnt_out + OD_C_TILES, 1)] = L.lin;
}

} // namespace umpa

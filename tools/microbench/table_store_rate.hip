// How fast can the shift table be WRITTEN?  corr_march's store pattern and alternatives, stores only:
// 456 workgroups x 448 threads, every "step" a workgroup writes one row segment (54 columns = 432 bytes) of each of its
// 27 planes; 517 steps (BASELINE config C2's geometry: 2028 x 2028 x 81 shifts x 8 bytes = 2.66 GB).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/table_store_rate.hip -o _exp/table_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double pair_t __attribute__((ext_vector_type(2)));
#define GLOBAL __attribute__((address_space(1)))

// MODE 0: lane l of a 16-lane row stores columns 4l..4l+3 as two 16-byte stores (pairs interleaved at 32 bytes)
// MODE 1: the same bytes, each instruction a contiguous run (16 lanes x 16 bytes, then 11 x 16)
// MODE 2: MODE 0 with plain (not non-temporal) stores
// MODE 3: a wave writes ONE plane per instruction: 4 rows x 256 bytes ... (4 steps buffered) -- 27 lanes of 16 bytes per row
// MODE 4: upper bound: every workgroup streams a contiguous chunk of the table
// MODE 5 / 6: the table blocked by strips, dense: [strip][row][shift][52 columns]; lane l (< 13) stores its two pairs
//   (5: non-temporal, 6: plain)
// MODE 7 / 8: the same table, every instruction writes whole 64-byte pieces: lane 4i+j stores pair 8i+j, then 8i+4+j
//   (7: non-temporal, 8: plain)
template <int MODE>
__global__ void __launch_bounds__(448) store_kernel(double* table, size_t slot_stride, size_t row_pitch, int N1, int nstrips, int band_rows, int rows, int wo, int align)
{
    const int tid = threadIdx.x, l = tid & 15, pl = tid >> 4;
    const int nitems = nstrips * 4, per_xcd = (nitems + 7) >> 3;
    const int seq = blockIdx.x >> 3, item = (blockIdx.x & 7) * per_xcd + seq / 3, pass = seq % 3;
    if (seq / 3 >= per_xcd || item >= nitems) return;
    const int strip = item % nstrips, band = item / nstrips;
    const int r_lo = band * band_rows, r_hi = min(rows, r_lo + band_rows);
    const int slot = pass * 27 + pl;
    const bool ok = pl < 27;
    const int c_lo = strip * (align ? 64 : wo);
    const int c_end = align ? c_lo + 48 : min(N1, c_lo + wo);
    if (MODE == 4) {
        // contiguous: the workgroup's share of the whole table
        const size_t total = (size_t)81 * rows * N1, share = total / gridDim.x / 2 * 2;
        GLOBAL pair_t* p = (GLOBAL pair_t*)(table + (size_t)blockIdx.x * share);
        pair_t v; v[0] = tid; v[1] = 1.0;
        for (size_t i = tid; i < share / 2; i += 448) __builtin_nontemporal_store(v, p + i);
        return;
    }
    if (MODE >= 5) {
        const int TWc = wo;                                            // columns of a table block (52 dense, 56, 64)
        GLOBAL double* b5 = (GLOBAL double*)table + (((size_t)strip * rows + r_lo) * 81 + slot) * TWc;
        pair_t v; v[0] = tid; v[1] = blockIdx.x;
        const int i4 = l >> 2, j4 = l & 3;
        for (int r = 0; r < r_hi - r_lo; r++) {
            GLOBAL double* dst = b5 + (size_t)r * 81 * TWc;
            if (MODE == 5 || MODE == 6) {
                if (ok && 4 * l < TWc) {
                    if (MODE == 5) { __builtin_nontemporal_store(v, (GLOBAL pair_t*)(dst + 4 * l)); __builtin_nontemporal_store(v, (GLOBAL pair_t*)(dst + 4 * l + 2)); }
                    else { *(GLOBAL pair_t*)(dst + 4 * l) = v; *(GLOBAL pair_t*)(dst + 4 * l + 2) = v; }
                }
            } else {
                const int p1 = 8 * i4 + j4, p2 = p1 + 4;
                if (ok && 2 * p1 < TWc) { if (MODE == 7) __builtin_nontemporal_store(v, (GLOBAL pair_t*)(dst + 2 * p1)); else *(GLOBAL pair_t*)(dst + 2 * p1) = v; }
                if (ok && 2 * p2 < TWc) { if (MODE == 7) __builtin_nontemporal_store(v, (GLOBAL pair_t*)(dst + 2 * p2)); else *(GLOBAL pair_t*)(dst + 2 * p2) = v; }
            }
        }
        return;
    }
    GLOBAL double* base = (GLOBAL double*)table + (size_t)slot * slot_stride + (size_t)r_lo * row_pitch + c_lo;
    pair_t v; v[0] = tid; v[1] = blockIdx.x;
    for (int r = 0; r < r_hi - r_lo; r++) {
        GLOBAL double* dst = base + (size_t)r * row_pitch;
        if (MODE == 0 || MODE == 2) {
            const int c = c_lo + 4 * l;
            if (ok && c + 1 < c_end) { if (MODE == 0) __builtin_nontemporal_store(v, (GLOBAL pair_t*)(dst + 4 * l)); else *(GLOBAL pair_t*)(dst + 4 * l) = v; }
            if (ok && c + 3 < c_end) { if (MODE == 0) __builtin_nontemporal_store(v, (GLOBAL pair_t*)(dst + 4 * l + 2)); else *(GLOBAL pair_t*)(dst + 4 * l + 2) = v; }
        } else if (MODE == 1) {
            if (ok && c_lo + 2 * l + 1 < c_end) __builtin_nontemporal_store(v, (GLOBAL pair_t*)(dst + 2 * l));
            if (ok && c_lo + 32 + 2 * l + 1 < c_end) __builtin_nontemporal_store(v, (GLOBAL pair_t*)(dst + 32 + 2 * l));
        } else if (MODE == 3) {
            // every 4th step: wave w stores 4 rows of each of its 4 planes, one plane per instruction pair:
            // lanes 0..26 the 27 pairs of row r-3 ... (rows on lane groups of 32: two instructions per plane)
            if ((r & 3) == 3) {
                const int wl = tid & 63, wv = tid >> 6;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int sl = pass * 27 + wv * 4 + q;
                    if (wv * 4 + q < 27) {
                        GLOBAL double* b2 = (GLOBAL double*)table + (size_t)sl * slot_stride + (size_t)(r_lo + r - 3) * row_pitch + c_lo;
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int rr = 2 * h + (wl >> 5), pc = wl & 31;
                            if (c_lo + 2 * pc + 1 < c_end) __builtin_nontemporal_store(v, (GLOBAL pair_t*)(b2 + (size_t)rr * row_pitch + 2 * pc));
                        }
                    }
                }
            }
        }
    }
}

template <int MODE>
void run(const char* name, double* table, size_t ss, size_t rp, int N1, int rows, int wo, int align)
{
    const int nstrips = MODE >= 5 ? 39 : align ? 38 : (N1 + wo - 1) / wo, band_rows = (rows + 3) / 4;
    const int grid = 8 * ((nstrips * 4 + 7) / 8) * 3;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    store_kernel<MODE><<<grid, 448>>>(table, ss, rp, N1, nstrips, band_rows, rows, wo, align);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int i = 0; i < 10; i++) {
        hipEventRecord(e0);
        store_kernel<MODE><<<grid, 448>>>(table, ss, rp, N1, nstrips, band_rows, rows, wo, align);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = MODE >= 5 ? 81.0 * rows * 39 * wo * 8 : align ? 81.0 * rows * 38 * 384 : 81.0 * rows * N1 * 8;
    printf("%-34s %.3f ms  %.2f TB/s\n", name, best, bytes / best * 1e-9);
}

int main(int argc, char** argv)
{
    const int rows = 2028, N1 = 2028, wo = 54;
    double* table;
    const size_t n = (size_t)81 * rows * 39 * 64 + 4096;           // the largest layout below: 39 strips of 64 columns
    if (hipMalloc(&table, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(table, 0, n * 8);
    for (int layout = 0; layout < 2; layout++) {
        for (int align = 0; align < 2; align++) {
            const int pitch = align ? 2432 : N1;
            const size_t ss = layout ? pitch : (size_t)rows * pitch, rp = layout ? (size_t)81 * pitch : pitch;
            printf("-- layout %s, %s\n", layout ? "[row][shift][col]" : "[shift][row][col]", align ? "48-column runs, 512-byte aligned" : "54-column runs, unaligned");
            run<0>("pairs interleaved (corr_march)", table, ss, rp, N1, rows, wo, align);
            run<1>("contiguous per 16 lanes", table, ss, rp, N1, rows, wo, align);
            run<2>("pairs interleaved, plain stores", table, ss, rp, N1, rows, wo, align);
            run<3>("one plane per instruction, 4 rows", table, ss, rp, N1, rows, wo, align);
        }
    }
    run<4>("contiguous stream (upper bound)", table, 0, 0, N1, rows, wo, 0);
    printf("-- layout [strip][row][shift][52], dense (39 strips)\n");
    run<5>("pairs interleaved, non-temporal", table, 0, 0, N1, rows, 52, 0);
    run<6>("pairs interleaved, plain", table, 0, 0, N1, rows, 52, 0);
    run<7>("64-byte pieces, non-temporal", table, 0, 0, N1, rows, 52, 0);
    run<8>("64-byte pieces, plain", table, 0, 0, N1, rows, 52, 0);
    for (int tw = 56; tw <= 64; tw += 8) {
        printf("-- layout [strip][row][shift][%d]\n", tw);
        run<5>("pairs interleaved, non-temporal", table, 0, 0, N1, rows, tw, 0);
        run<6>("pairs interleaved, plain", table, 0, 0, N1, rows, tw, 0);
        run<7>("64-byte pieces, non-temporal", table, 0, 0, N1, rows, tw, 0);
        run<8>("64-byte pieces, plain", table, 0, 0, N1, rows, tw, 0);
    }
    return 0;
}

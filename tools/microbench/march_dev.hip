// Development harness of corr_march (umpa_amd/csrc/umpa_march.h): the kernel alone on random stacks of BASELINE config C2's
// (or C3's) geometry, checked against a plain CPU sum on sampled (pixel, shift) pairs and on whole rows at the region's edges.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I umpa_amd/csrc -I tools/microbench tools/microbench/march_dev.hip -o _exp/march_dev
//   gpurun_out/march_dev [c2|c3|small] [nbands] [ablate]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include "umpa_march.h"

using namespace umpa;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NW, int NXB, int NPD, int LA, int NT, int WPS, int SPB = 1>
static float run(const ModelDev& dev, const MarchArgs& A, const Sep1D& sep, size_t lds, int grid, int reps)
{
    auto kern = corr_march_kernel<NW, NXB, NPD, LA, NT, WPS, SPB>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, dev, A, sep);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int r = 0; r < (getenv("REPS") ? atoi(getenv("REPS")) : reps); r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, dev, A, sep);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms); sum += ms;
    }
    printf("corr_march: best %.3f ms, mean %.3f ms over %d launches (grid %d x %d threads, %zu B LDS)\n", best, sum / reps, reps, grid, NT, lds);
    return best;
}

int main(int argc, char** argv)
{
    const char* cfg = argc > 1 ? argv[1] : "c2";
    int H = 2048, W = 2048, K = 10, Nw = 5, ms = 5;
    if (!strcmp(cfg, "c3")) { H = W = 4096; K = 20; Nw = 7; ms = 8; }
    if (!strcmp(cfg, "small")) { H = 200; W = 333; K = 3; Nw = 5; ms = 5; }
    if (!strcmp(cfg, "c5")) { K = 5; }
    int nbands_arg = argc > 2 ? atoi(argv[2]) : 0;
    const int ablate = argc > 3 ? atoi(argv[3]) : 0;
    const int sigma = argc > 4 ? atoi(argv[4]) : 1;
    const int P = Nw + ms, N0 = H - 2 * P, N1 = W - 2 * P, UJ = 2 * ms - 1;
    printf("config %s: %dx%d, K=%d, Nw=%d, ms=%d, region %dx%d, %d shifts\n", cfg, H, W, K, Nw, ms, N0, N1, UJ * UJ);

    // stacks: one allocation per stack, random values
    const size_t plane = (size_t)H * W;
    std::vector<double> hs(plane * K), hr(plane * K);
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.5, 1.5);
    for (auto& x : hs) x = U(rng);
    for (auto& x : hr) x = U(rng);
    double *dsam, *dref;
    CK(hipMalloc(&dsam, plane * K * 8 + 64)); CK(hipMalloc(&dref, plane * K * 8 + 64));
    CK(hipMemcpy(dsam, hs.data(), plane * K * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dref, hr.data(), plane * K * 8, hipMemcpyHostToDevice));
    std::vector<FrameDesc> fd(K);
    for (int k = 0; k < K; k++) { fd[k].sam = dsam + k * plane; fd[k].ref = dref + k * plane; fd[k].mask = nullptr; fd[k].H = H; fd[k].W = W; fd[k].pi = fd[k].pj = 0; }
    FrameDesc* dfd;
    CK(hipMalloc(&dfd, K * sizeof(FrameDesc)));
    CK(hipMemcpy(dfd, fd.data(), K * sizeof(FrameDesc), hipMemcpyHostToDevice));

    ModelDev dev;
    memset(&dev, 0, sizeof(dev));
    dev.frames = dfd; dev.Na = K; dev.Nw = Nw; dev.ms = ms; dev.padding = P; dev.Nwt = K;
    Sep1D sep;
    memset(&sep, 0, sizeof(sep));
    {
        const int S = 2 * Nw + 1;
        double tot = 0;
        std::vector<double> h(S);
        for (int i = 0; i < S; i++) { h[i] = 0.54 - 0.46 * cos(2 * M_PI * i / (S - 1)); tot += h[i]; }
        for (int i = 0; i < S; i++) { sep.hr[i] = h[i] / tot; sep.hc[i] = h[i] / tot; }
    }

    const int NXB = (UJ - 1 + 3) / 4, NBE = 16 + NXB, WO = (63 - 2 * Nw) / 4 * 4, TW = getenv("TW64") ? 64 : WO;
    const int nstrips = (N1 + WO - 1) / WO;
    const size_t table_n = (size_t)nstrips * N0 * UJ * UJ * TW;
    double* table;
    CK(hipMalloc(&table, table_n * 8));
    CK(hipMemset(table, 0xff, table_n * 8));

    MarchArgs A;
    memset(&A, 0, sizeof(A));
    A.table = table; A.tw = TW;
    A.org0 = P; A.org1 = P; A.row0 = 0; A.rows = N0; A.N1 = N1; A.sigma = sigma;
    A.br0 = 0; A.br1 = H - 1; A.bc0 = 0; A.bc1 = W; A.Wf = W;      // (the allocation has 64 spare bytes: a pair may start on the last column)
    int nuy = Nw <= 5 ? 3 : 2;
    if (getenv("NUY")) nuy = atoi(getenv("NUY"));
    const int ncw = (nuy * UJ + 3) / 4, NT = ncw * 64;
    A.nuy = nuy; A.npass = (UJ + nuy - 1) / nuy;
    A.nstrips = nstrips;
    const int slots = 256 * (NT <= 512 ? 2 : 1);
    int nbands = nbands_arg > 0 ? nbands_arg : std::max(1, slots / (A.nstrips * A.npass));
    A.nbands = nbands; A.band_rows = (N0 + nbands - 1) / nbands;
    A.npa = (K + 1) / 2; A.npb = (K * 2 * NBE + 63) / 64;
    A.a_slot = A.npa * 1024;
    A.b_slot = A.npb * 1024;
    const int LA = getenv("LA") ? atoi(getenv("LA")) : 2;
    const int SPB = getenv("SPB") ? atoi(getenv("SPB")) : 1;
    A.da = SPB > 1 ? 2 * SPB : LA + 1; A.db = SPB > 1 ? nuy - 1 + 2 * SPB : nuy + LA;
    const bool sam_is_A = sigma > 0;
    A.baseA = (const char*)(sam_is_A ? dsam : dref); A.baseB = (const char*)(sam_is_A ? dref : dsam);
    std::vector<unsigned> foff(2 * K);
    for (int k = 0; k < K; k++) { foff[k] = (unsigned)(k * plane * 8); foff[K + k] = (unsigned)(k * plane * 8); }
    unsigned* dfoff;
    CK(hipMalloc(&dfoff, 2 * K * 4));
    CK(hipMemcpy(dfoff, foff.data(), 2 * K * 4, hipMemcpyHostToDevice));
    A.frame_off = dfoff;
    A.ablate = ablate;
    const size_t lds = (size_t)A.da * A.a_slot + (size_t)A.db * A.b_slot;
    const int nitems = A.nstrips * A.nbands, grid = 8 * ((nitems + 7) / 8) * A.npass;
    const int npd = A.npa + A.npb, npt = (npd + ncw - 1) / ncw;
    printf("strips %d (WO %d, TW %d), bands %d (%d rows), passes %d (nuy %d), NT %d, DMA instructions/step %d (%d per wave), LDS %zu\n", A.nstrips, WO, TW, A.nbands, A.band_rows, A.npass, nuy, NT, npd, npt, lds);

    float best = 0;
    const int reps = 20;
    if (Nw == 5 && NT == 448 && npt <= 2 && LA == 2 && SPB == 1) best = run<5, 2, 2, 2, 448, 4>(dev, A, sep, lds, grid, reps);
    else if (Nw == 5 && NT == 448 && npt <= 2 && LA == 3) best = run<5, 2, 2, 3, 448, 4>(dev, A, sep, lds, grid, reps);
    else if (Nw == 5 && NT == 768 && npt <= 1 && LA == 2) best = run<5, 2, 1, 2, 768, 3>(dev, A, sep, lds, grid, reps);
    else if (Nw == 5 && NT == 576 && npt <= 2 && LA == 2) best = run<5, 2, 2, 2, 576, 3>(dev, A, sep, lds, grid, reps);
    else if (Nw == 7 && NT == 512 && npt <= 3 && LA == 2) best = run<7, 4, 3, 2, 512, 2>(dev, A, sep, lds, grid, reps);
    else if (Nw == 7 && NT == 768 && npt <= 2 && LA == 2 && SPB == 2) best = run<7, 4, 2, 2, 768, 3, 2>(dev, A, sep, lds, grid, reps);
    else if (Nw == 7 && NT == 768 && npt <= 2 && LA == 2 && SPB == 3) best = run<7, 4, 2, 2, 768, 3, 3>(dev, A, sep, lds, grid, reps);
    else if (Nw == 5 && NT == 448 && npt <= 2 && LA == 2 && SPB == 2) best = run<5, 2, 2, 2, 448, 4, 2>(dev, A, sep, lds, grid, reps);
    else if (Nw == 7 && NT == 768 && npt <= 2 && LA == 2) best = run<7, 4, 2, 2, 768, 3>(dev, A, sep, lds, grid, reps);
    else if (Nw == 7 && NT == 768 && npt <= 2 && LA == 3) best = run<7, 4, 2, 3, 768, 3>(dev, A, sep, lds, grid, reps);
    else { printf("no instantiation for Nw=%d NT=%d npt=%d\n", Nw, NT, npt); return 1; }
    const double useful = (double)(K + 2 * (2 * Nw + 1)) * UJ * UJ * N0 * N1;
    printf("useful FMA %.3f G -> %.1f TFLOP/s (%.1f %% of 78.6); %.0f Mpx/s for this kernel alone\n", useful * 1e-9, 2 * useful / best * 1e-9,
           2 * useful / best * 1e-9 / 78.6 * 100, (double)N0 * N1 / best * 1e-3);
    if (ablate) return 0;

    // ---- check
    std::vector<double> ht(table_n);
    CK(hipMemcpy(ht.data(), table, table_n * 8, hipMemcpyDeviceToHost));
    const std::vector<double>& SA = sam_is_A ? hs : hr;
    const std::vector<double>& SB = sam_is_A ? hr : hs;
    auto cpu = [&](int i, int j, int oi, int oj) {
        double acc = 0;
        const int S = 2 * Nw + 1;
        for (int k = 0; k < K; k++)
            for (int a = 0; a < S; a++)
                for (int b = 0; b < S; b++) {
                    const size_t ra = (size_t)(P + i - Nw + a), ca = (size_t)(P + j - Nw + b);
                    acc += sep.hr[a] * sep.hc[b] * SA[k * plane + ra * W + ca] * SB[k * plane + (ra + oi) * W + (ca + oj)];
                }
        return acc;
    };
    double worst = 0; long bad = 0, n = 0;
    auto check = [&](int i, int j, int oi, int oj) {
        const size_t slot = (size_t)(sigma * oi + ms - 1) * UJ + (sigma * oj + ms - 1);
        const double got = ht[(((size_t)(j / WO) * N0 + i) * UJ * UJ + slot) * TW + j % WO], want = cpu(i, j, oi, oj);
        const double rel = fabs(got - want) / fabs(want);
        if (!(rel < 1e-12)) { if (bad < 10) printf("  MISMATCH px (%d,%d) shift (%d,%d): got %.15g want %.15g\n", i, j, oi, oj, got, want); bad++; }
        if (rel > worst) worst = rel;
        n++;
    };
    std::uniform_int_distribution<int> Ri(0, N0 - 1), Rj(0, N1 - 1), Ru(-(ms - 1), ms - 1);
    for (int t = 0; t < 20000; t++) check(Ri(rng), Rj(rng), Ru(rng), Ru(rng));
    const int rows[] = {0, 1, N0 - 1, std::min(N0 - 1, A.band_rows - 1), std::min(N0 - 1, A.band_rows), std::min(N0 - 1, A.band_rows + 1)};
    for (int r : rows)
        for (int j = 0; j < N1; j++) { check(r, j, Ru(rng), Ru(rng)); }
    const int cols[] = {0, 1, WO - 1, WO, WO + 1, N1 - 1, N1 - 2};
    for (int c : cols)
        for (int i = 0; i < N0; i += 7) check(i, c, Ru(rng), Ru(rng));
    // every shift at a few pixels
    for (int oi = -(ms - 1); oi <= ms - 1; oi++)
        for (int oj = -(ms - 1); oj <= ms - 1; oj++) { check(N0 / 2, N1 / 2, oi, oj); check(3, N1 - 3, oi, oj); }
    printf("check: %ld samples, %ld mismatches, worst relative difference %.2e -> %s\n", n, bad, worst, bad ? "FAIL" : "ok");
    return bad ? 2 : 0;
}

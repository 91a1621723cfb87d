// What the chip sustains on the two instruction kinds corr_volume is made of: fp64 FMAs (independent accumulators) and LDS
// reads, alone and together, at the occupancy of the kernel (512 threads per CU).  hipcc --offload-arch=gfx950 -O3 ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 1: FMAs, 2: LDS reads (b128), 3: both interleaved
__global__ void __launch_bounds__(512) rate_kernel(double* out, int iters, double seed)
{
    __shared__ __attribute__((aligned(16))) double sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) sm[i] = seed * i;
    __syncthreads();
    double acc[32];
#pragma unroll
    for (int j = 0; j < 32; j++) acc[j] = seed + j;
    typedef double pair_t __attribute__((ext_vector_type(2)));
    const pair_t* p = reinterpret_cast<const pair_t*>(sm) + threadIdx.x;
    pair_t s = {0.0, 0.0};
    double a = seed * 1.0001, b = seed * 0.9999;
    for (int it = 0; it < iters; it++) {
        if (MODE & 2) {
#pragma unroll
            for (int j = 0; j < 8; j++) { pair_t v = p[(j * 512 + it * 7) & 2047]; s += v; }
        }
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < 32; j++) acc[j] = fma(acc[j], a, b);
        }
    }
    double t = s[0] + s[1];
#pragma unroll
    for (int j = 0; j < 32; j++) t += acc[j];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

template <int MODE>
void run(const char* name, double* d, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<256, 512>>>(d, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<MODE><<<256, 512>>>(d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fma = (MODE & 1) ? 256.0 * 512 * iters * 128 : 0, lds = (MODE & 2) ? 256.0 * 512 * iters * 8 * 16 : 0;
    printf("%-12s %.3f ms  %.1f TFLOP/s fp64  %.1f TB/s LDS\n", name, ms, 2 * fma / ms * 1e-9, lds / ms * 1e-9);
}

int main()
{
    double* d;
    hipMalloc(&d, 256 * 512 * sizeof(double));
    run<1>("fma", d, 20000);
    run<2>("lds", d, 20000);
    run<3>("fma+lds", d, 20000);
    run<1>("fma", d, 20000);
    return 0;
}

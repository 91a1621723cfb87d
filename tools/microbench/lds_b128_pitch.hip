// Bank conflicts of the product stage's read pattern: 512 threads, thread (pr = tid % ROWS, pqb = tid / ROWS) reads the
// 16-byte pieces pr * PITCH + 2 * pqb + t of an LDS image (ds_read_b128), for several row pitches (in 16-byte pieces) and
// row counts.  Run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE and compare the kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double pair_t __attribute__((ext_vector_type(2)));

template <int PITCH, int ROWS>
__global__ void __launch_bounds__(512) read_kernel(double* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char sm[];
    pair_t* img = reinterpret_cast<pair_t*>(sm);
    for (int i = threadIdx.x; i < 4096; i += 512) { pair_t v = {1.0 * i, 2.0 * i}; img[i] = v; }
    __syncthreads();
    const int pr = threadIdx.x % ROWS, pqb = threadIdx.x / ROWS;
    const pair_t* p = img + pr * PITCH + 2 * pqb;
    pair_t s = {0.0, 0.0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 8; t++) { pair_t v = p[(t + it) & 7]; s += v; }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1];
}

template <int PITCH, int ROWS>
void run(double* d)
{
    hipLaunchKernelGGL((read_kernel<PITCH, ROWS>), dim3(256), dim3(512), 65536, 0, d, 2000);
    hipDeviceSynchronize();
    printf("pitch %d rows %d done\n", PITCH, ROWS);
}

int main()
{
    double* d;
    hipMalloc(&d, 256 * 512 * sizeof(double));
    run<21, 42>(d); run<25, 42>(d); run<23, 42>(d); run<19, 42>(d); run<17, 42>(d); run<27, 42>(d);
    run<21, 64>(d); run<25, 64>(d); run<23, 46>(d); run<19, 46>(d);
    return 0;
}

// micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on this GPU (the FP64 matrix peak the roofline divides by)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, int iters) {
    d4 acc[NACC];
    for (int a = 0; a < NACC; a++) acc[a] = d4{0, 0, 0, 0};
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    double s = 0;
    for (int a = 0; a < NACC; a++) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 20000;
    double* d; hipMalloc(&d, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * (double)iters * NACC * blocks * (threads / 64);
    printf("acc=%d waves/SIMD=%d: %.2f TFLOP/s FP64 MFMA  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NACC, waves_per_simd,
           flops / ms / 1e9, (ms * 1e-3 * 2.4e9) / ((double)iters * NACC * waves_per_simd));
    hipFree(d);
}
int main() { run<1>(1); run<2>(1); run<4>(1); run<4>(2); run<8>(2); run<4>(4); return 0; }

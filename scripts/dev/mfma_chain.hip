// micro-benchmark: latency of dependent v_mfma_f64_16x16x4_f64 chains as they appear in the Riccati sweeps
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
// variant 0: 7 MFMAs chained through srcC only; 1: result of a 3-chain feeds operand B of a 4-chain (forward sweep shape);
// 2: same as 1 plus the result is read by a VALU op in between
template <int V>
__global__ void k(double* out, int iters, unsigned long long* cyc) {
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 + threadIdx.x * 1e-4;
    d4 x = {b, b, b, 0.0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        d4 c = {a, 0, 0, 0};
        if (V == 0) {
            d4 acc = c;
            for (int j = 0; j < 7; j++) acc = mfma(a, b, acc);
            x = acc;
        } else {
            d4 v = c;
            v = mfma(a, x[0], v); v = mfma(a, x[1], v); v = mfma(a, x[2], v);
            double v0 = v[0];
            if (V == 2) v0 = v0 * 1.0000001 + 1e-9;
            d4 y = {b, b, b, 0};
            y = mfma(a, x[0], y); y = mfma(a, x[1], y); y = mfma(a, x[2], y); y = mfma(a, v0, y);
            x = y; x[3] = 0.0;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V> void run() {
    double* d; unsigned long long* c; hipMalloc(&d, 8 * 64 * 1024); hipMalloc(&c, 8);
    hipLaunchKernelGGL(k<V>, dim3(1024), dim3(64), 0, 0, d, 2000, c);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("variant %d: %.1f cycles per iteration (7 MFMAs) = %.1f per MFMA\n", V, h / 2000.0, h / 14000.0);
    hipFree(d); hipFree(c);
}
int main() { run<0>(); run<1>(); run<2>(); return 0; }

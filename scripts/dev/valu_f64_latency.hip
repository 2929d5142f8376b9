// dev tool: issue vs. dependent-issue cost of FP64 VALU ops on gfx950, one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void chain(double* out, long long* cyc, double a, double b) {
    double x[CH];
    for (int k = 0; k < CH; k++) x[k] = threadIdx.x * 1e-3 + k;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 256; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int k = 0; k < CH; k++) x[k] = __builtin_fma(x[k], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int k = 0; k < CH; k++) s += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
void run(const char* name) {
    double* o; long long* c;
    hipMalloc(&o, 1024 * 64 * 8); hipMalloc(&c, 1024 * 8);
    chain<CH><<<1024, 64>>>(o, c, 0.999, 1e-3);
    hipDeviceSynchronize();
    long long h[1024]; hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    printf("%s: %d independent chains: %.2f cycles per v_fma_f64\n", name, CH, (double)h[512] / (256.0 * 16 * CH));
}
int main() { run<1>("fma"); run<2>("fma"); run<3>("fma"); run<4>("fma"); run<8>("fma"); return 0; }

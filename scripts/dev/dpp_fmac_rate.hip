#include <hip/hip_runtime.h>
#include <cstdio>
template <int K> __device__ __forceinline__ void fm(double& acc, double s, double a) { asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(a), "n"(K)); }
__device__ __forceinline__ void fn(double& acc, double s, double a) { asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(s), "v"(a)); }
template <int MODE>
__global__ void k(double* o, long long* cyc, double a) {
    double acc[16], b[16];
    for (int j = 0; j < 16; j++) { acc[j] = 0; b[j] = threadIdx.x + j; }
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 256; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) { if (MODE == 0) fn(acc[j], b[j], a); else fm<3>(acc[j], b[j], a); }
#pragma unroll
        for (int j = 0; j < 16; j++) { if (MODE == 0) fn(acc[j], b[(j + 1) & 15], a); else fm<7>(acc[j], b[(j + 1) & 15], a); }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int j = 0; j < 16; j++) s += acc[j];
    o[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* n) {
    double* o; long long* c; (void)hipMalloc(&o, 1024 * 512); (void)hipMalloc(&c, 8192);
    k<MODE><<<1024, 64>>>(o, c, 0.5); (void)hipDeviceSynchronize();
    long long h[1024]; (void)hipMemcpy(h, c, 8192, hipMemcpyDeviceToHost);
    printf("%s: %.2f cycles per instruction (16 independent accumulators)\n", n, (double)h[512] / (256.0 * 32));
}
int main() { run<0>("v_fmac_f64_e32"); run<1>("v_fmac_f64_dpp row_newbcast"); return 0; }

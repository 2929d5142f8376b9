// dev tool: can one wave overlap its own FP64 MFMA (64-cycle pipe occupancy) with independent FP64 VALU work?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NV, int NM>
__global__ void k(double* out, long long* cyc, double a, double b) {
    double x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 1e-3 + j;
    d4 c[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double p = threadIdx.x * 0.01, q = 1.0 - threadIdx.x * 0.02;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 512; it++) {
#pragma unroll
        for (int m = 0; m < NM; m++) c[m & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(p, q, c[m & 1], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; v++) x[v & 7] = __builtin_fma(x[v & 7], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s + c[0][0] + c[0][1] + c[0][2] + c[0][3] + c[1][0] + c[1][3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// same with 32-bit integer VALU work (v_bfi-like selects / address arithmetic)
template <int NV, int NM>
__global__ void k32(double* out, long long* cyc, unsigned a, unsigned b) {
    unsigned x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 7 + j;
    d4 c[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double p = threadIdx.x * 0.01, q = 1.0 - threadIdx.x * 0.02;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 512; it++) {
#pragma unroll
        for (int m = 0; m < NM; m++) c[m & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(p, q, c[m & 1], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; v++) x[v & 7] = (x[v & 7] & a) | (b & ~a) ^ x[(v + 1) & 7];
    }
    long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
    for (int j = 0; j < 8; j++) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s + c[0][0] + c[0][1] + c[0][2] + c[0][3] + c[1][0] + c[1][3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NV, int NM>
void run32() {
    double* o; long long* c;
    (void)hipMalloc(&o, 1024 * 64 * 8); (void)hipMalloc(&c, 1024 * 8);
    k32<NV, NM><<<1024, 64>>>(o, c, 0x0f0f0f0fu, 0x12345678u);
    (void)hipDeviceSynchronize();
    long long h[1024]; (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    printf("%d MFMA + %2d 32-bit VALU statements (2-3 instr each) per trip: %.1f cycles per trip\n", NM, NV, (double)h[512] / 512.0);
}
template <int NV, int NM>
void run() {
    double* o; long long* c;
    (void)hipMalloc(&o, 1024 * 64 * 8); (void)hipMalloc(&c, 1024 * 8);
    k<NV, NM><<<1024, 64>>>(o, c, 0.999, 1e-3);
    (void)hipDeviceSynchronize();
    long long h[1024]; (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    printf("%d MFMA (2 alternating accumulators) + %2d independent v_fma_f64 per trip: %.1f cycles per trip\n", NM, NV, (double)h[512] / 512.0);
}
int main() {
    run<0, 1>(); run<8, 1>(); run<12, 1>(); run<16, 1>(); run<24, 1>(); run<32, 1>();
    run<0, 2>(); run<16, 2>(); run<28, 2>(); run<40, 2>(); run<16, 0>(); run<32, 0>();
    run32<8, 0>(); run32<16, 0>(); run32<8, 1>(); run32<16, 1>(); run32<8, 2>(); run32<16, 2>();
    return 0;
}

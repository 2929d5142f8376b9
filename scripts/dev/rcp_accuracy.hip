// dev: accuracy of v_rcp_f64 alone and after one / two Newton steps (the pivot recursion of the Riccati sweeps uses 1/d)
//   hipcc --offload-arch=gfx950 -O2 scripts/dev/rcp_accuracy.hip -o /tmp/rcp_accuracy && /tmp/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* d, double* o0, double* o1, double* o2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    double y = __builtin_amdgcn_rcp(x);
    o0[i] = y;
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    o1[i] = y;
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    o2[i] = y;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> h(n), r0(n), r1(n), r2(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (s >> 11) * (1.0 / 9007199254740992.0);
        h[i] = std::ldexp(1.0 + u, (int)(s % 120) - 60);   // 2^-60 .. 2^60
    }
    double *d, *o0, *o1, *o2;
    hipMalloc(&d, n * 8); hipMalloc(&o0, n * 8); hipMalloc(&o1, n * 8); hipMalloc(&o2, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o0, o1, o2, n);
    hipMemcpy(r0.data(), o0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), o1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / (long double)h[i];
        m0 = std::fmax(m0, (double)fabsl((r0[i] - t) / t)); m1 = std::fmax(m1, (double)fabsl((r1[i] - t) / t)); m2 = std::fmax(m2, (double)fabsl((r2[i] - t) / t));
    }
    std::printf("max relative error: v_rcp_f64 %.3e, + 1 Newton step %.3e, + 2 steps %.3e (eps = 2.2e-16)\n", m0, m1, m2);
    return 0;
}

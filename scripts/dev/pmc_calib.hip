// calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 with THIS project's access pattern (8-byte per lane,
// wave-contiguous 512 B): streams a known number of bytes (MI355X_MICROARCH.md "HBM": calibrate before trusting absolutes)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read8(const double* __restrict__ a, double* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456) out[0] = s;
}
__global__ void calib_write8(double* __restrict__ a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}
int main() {
    const size_t n = (size_t)1 << 27;  // 1 GiB of doubles: larger than the 256 MiB Infinity Cache
    double *a, *o;
    if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&o, 8) != hipSuccess) return 1;
    hipMemset(a, 0, n * 8);
    hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, a, n);
    hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, a, o, n);
    hipDeviceSynchronize();
    printf("calib bytes %zu\n", n * 8);
    return 0;
}

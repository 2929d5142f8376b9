// traj_kernel.hip -- the step immediately BEFORE the hot path, on the device: reference windows
// yref[N+1][16] cut out of a resident trajectory table with the reference's end-padding rule, and analytic per-instance
// candidate windows (BASELINE config 4) so that tens of thousands of references never cross PCIe.
//   window rule:   /root/reference/bluerov2_path/src/bluerov2_path.cpp:79-118 (read_N_pub), bluerov2_dobmpc/src/bluerov2_dob.cpp:218-265
//   12 vs 16 cols: bluerov2_dobmpc/src/ctrller/mpc.cpp:242-262 (CTRL node copies the 12 state columns only)
//   generators:    /root/reference/bluerov2_path/config/traj/circle.py:22-56, lemniscate.py:18-39
// Pure data movement / a handful of FP64 ops per element: HBM-bound, 128 B written per (instance, node).
#include <hip/hip_runtime.h>

#include "nmpc_device.hpp"

namespace brov {

// one thread per output element; lines == nullptr -> every instance uses `line0` (or a single shared window when B == 1)
__global__ void window_kernel(const double* __restrict__ traj, int rows, const int* __restrict__ lines, int line0, int B, int N,
                              int ncols, double* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)(N + 1) * 16;
    if (t >= per * B) return;
    const int b = (int)(t / per);
    const int rem = (int)(t - (size_t)b * per);
    const int i = rem >> 4, c = rem & 15;
    const int line = lines ? lines[b] : line0;
    int row = line + i;
    if (row > rows - 1 || line >= rows) row = rows - 1;  // past the end: the last row, repeated
    if (row < 0) row = 0;
    out[t] = (c < ncols) ? traj[(size_t)row * 16 + c] : 0.0;
}

// kind 0: lemniscate (p0 = amp, p1 = frq), kind 1: circle (p0 = r, p1 = v); node i at t0 + i dt
__global__ void candidates_kernel(int kind, const double* __restrict__ p0, const double* __restrict__ p1,
                                  const double* __restrict__ phase, double t0, double dt, int B, int N, double* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * (N + 1)) return;
    const int b = t / (N + 1), i = t - b * (N + 1);
    const double tt = t0 + i * dt;
    double y[16];
#pragma unroll
    for (int c = 0; c < 16; c++) y[c] = 0.0;
    double s, co;
    if (kind == 0) {
        const double amp = p0[b], f = p1[b], a = tt * f + phase[b];
        sincos(a, &s, &co);
        y[0] = amp * co;
        y[1] = amp * s * co;
        y[2] = -20.0;
        y[6] = -amp * f * s;
        y[7] = amp * f * cos(2.0 * a);
    } else {
        const double r = p0[b], v = p1[b], a = tt * v / r + phase[b];
        sincos(a, &s, &co);
        y[0] = -r * co;
        y[1] = -r * s;
        y[2] = -20.0;
        y[5] = a - 0.5 * 3.14159265358979323846;
        y[6] = v;
        y[7] = v * cos(dt * v / r);  // circle.py:45-46: a scalar assigned to the whole column
        y[14] = 57.5;
    }
    double* o = out + (size_t)t * 16;
#pragma unroll
    for (int c = 0; c < 16; c++) o[c] = y[c];
}

void launch_window(const double* traj, int rows, const int* lines, int line0, int B, int N, int ncols, double* out, hipStream_t st) {
    const size_t tot = (size_t)B * (N + 1) * 16;
    hipLaunchKernelGGL(window_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, traj, rows, lines, line0, B, N, ncols, out);
}
void launch_candidates(int kind, const double* p0, const double* p1, const double* phase, double t0, double dt, int B, int N,
                       double* out, hipStream_t st) {
    const int tot = B * (N + 1);
    hipLaunchKernelGGL(candidates_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, kind, p0, p1, phase, t0, dt, B, N, out);
}

}  // namespace brov

// ---- the step immediately AFTER the hot path: plant update for closed-loop Monte-Carlo roll-outs ------
// x0 <- ERK4(x0, u0, p_plant, dt): the same 12-state model the OCP uses (bluerov2.py:103-137) integrated over one control
// period with the first optimal input of the last solve, per-instance TRUE parameters (disturbance draw, model mismatch).
// One lane per instance; ~600 FP64 ops, 240 B in / 96 B out.
#include "bluerov2_model.hpp"
namespace brov {

__global__ __launch_bounds__(128) void plant_kernel(double* __restrict__ x0, const brov_result* __restrict__ res, const double* __restrict__ pplant,
                             const double* __restrict__ prp, int rp_stride, int B, double dt, int substeps, double* __restrict__ xlog,
                             double* __restrict__ ulog) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x[NX], u[NU], k[NX], xs[NX], acc[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) x[j] = x0[(size_t)b * NX + j];
#pragma unroll
    for (int j = 0; j < NU; j++) u[j] = res[b].u0[j];
    const ModelPar m = make_par(pplant + (size_t)b * NP);
    Wrench w = make_wrench(u);
    if (prp) { w.k3 = prp[(size_t)b * rp_stride]; w.k4 = prp[(size_t)b * rp_stride + 1]; }   // 6-disturbance variant
    const double h = dt / substeps;
    StagePoint sp;
    for (int s = 0; s < substeps; s++) {
        model_f(x, w, m, k, sp);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] = x[j] + (h / 6.0) * k[j]; xs[j] = x[j] + 0.5 * h * k[j]; }
        model_f(xs, w, m, k, sp);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * k[j]; xs[j] = x[j] + 0.5 * h * k[j]; }
        model_f(xs, w, m, k, sp);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * k[j]; xs[j] = x[j] + h * k[j]; }
        model_f(xs, w, m, k, sp);
#pragma unroll
        for (int j = 0; j < NX; j++) x[j] = acc[j] + (h / 6.0) * k[j];
    }
#pragma unroll
    for (int j = 0; j < NX; j++) x0[(size_t)b * NX + j] = x[j];
    if (xlog) {
#pragma unroll
        for (int j = 0; j < NX; j++) xlog[(size_t)b * NX + j] = x[j];
    }
    if (ulog) {
#pragma unroll
        for (int j = 0; j < NU; j++) ulog[(size_t)b * NU + j] = u[j];
    }
}

void launch_plant(double* x0, const brov_result* res, const double* pplant, const double* prp, int rp_stride, int B, double dt, int substeps,
                  double* xlog, double* ulog, hipStream_t st) {
    hipLaunchKernelGGL(plant_kernel, dim3((B + 127) / 128), dim3(128), 0, st, x0, res, pplant, prp, rp_stride, B, dt, substeps, xlog, ulog);
}

}  // namespace brov

// ekf_kernel.hip -- batched EKF disturbance observer and its C ABI (include/bluerov2_nmpc.h,
// brov_ekf_*).  B independent copies of the reference's 18-state filter BLUEROV2_DOB::EKF()
// (/root/reference/bluerov2_dobmpc/src/bluerov2_dob.cpp:495-545): forward-difference Jacobians of the RK4 map (:730-744, RK4
// with the k2/3 stage quirk :621-634, process model :637-702) and of the measurement model (:705-727, :747-762), Kalman gain
// through the explicit inverse of the innovation covariance, Joseph-form covariance update, world-frame disturbance, and the
// hand-over to the NMPC parameters p[0..3] (:334-337).
//
// Three kernels.  ekf_update_kernel_sp (default since round 6): one filter per 16-lane DPP row, right-hand rows of the products
// broadcast out of registers with v_fmac_f64_dpp row_newbcast, only the non-zero pattern of the finite-difference Jacobians
// multiplied, 16 evaluations of the RK4 map in one pass, two LDS buffers, two waves per SIMD -- see the comment in front of it.
// ekf_update_kernel_dpp (rounds 2-5, BROV_EKF_VARIANT=1): the same mapping, dense.  ekf_update_kernel (first version,
// BROV_EKF_VARIANT=0): 19 lanes per filter, right-hand rows broadcast through LDS.  The older two are kept for A/B measurements.
// Common to all: 18 does not fit the 16-wide FP64 MFMA tile (a 32x32 padding would waste 3/4 of the issue slots, and FP64
// MFMA has the same flop rate as FP64 VALU on this part), so the filter runs on the VALU with one lane per matrix row; the
// lane that perturbs state r in the finite differences ends up holding column r of the Jacobian, i.e. row r of its
// transpose, and the products are arranged so that this is the form they need.
// HBM traffic per tick and filter: P in and out (2 x 2592 B), x (2 x 144 B), inputs 192 B, outputs 84 B = 5.7 KB against
// ~0.13 MFLOP: compute-bound on FP64 VALU issue.
//
// LDS-broadcast kernel (ekf_update_kernel):
//   * 19 lanes per filter, 3 filters per wavefront (57 of 64 lanes): lane r < 18 evaluates the map perturbed in state r,
//     lane 18 the unperturbed one -- the two finite-difference Jacobians cost one RK4 / one h() evaluation of wall time each;
//   * lane i < 18 then owns row i of every matrix; a product C = A B is "row i of C += A[i][k] * (row k of B)" with the
//     right-hand rows broadcast out of LDS (all 18 lanes read the same 144 bytes) and the accumulator row in registers;
//   * four 18x18 LDS buffers per filter (10.4 KB) + a few vectors: 33 KB per wave, 4 waves per CU.  Bound by LDS bandwidth.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "bluerov2_model.hpp"
#include "bluerov2_nmpc.h"

namespace brov {

constexpr int EN = 18;
constexpr int kGrp = 19;      // lanes per filter
constexpr int kPerWave = 3;   // filters per wavefront
constexpr int kMat = EN * EN;
constexpr int kVecs = 5 * EN + 2 * 40;               // xp, y, ye, xnew, spare | two pivot-row buffers
constexpr int kLdsPerFilter = 4 * kMat + kVecs;      // doubles
typedef __attribute__((address_space(3))) double elds;
typedef double ed2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) ed2 elds2;

struct EkfConst {
    double dt, mass, bo, mzg, iy_iz, iz_ix, iy_ix, R, d, inv_d, inv_cc, inv_rc;
    double Dl[6], Dnl[6], Md[6], iMd[6], K[36], Q[EN];
};
struct EkfArgs {
    EkfConst c;
    int B;
    double* x;            // [B][18]
    double* P;            // [B][18][18]
    const double* thrust; // [B][6]
    const double* y12;    // [B][12]
    const double* acc;    // [B][6]
    double* wf;           // [B][6]
    double* mp;           // [B][4]
    int* status;          // [B]
};

// first 12 components of the process model (the disturbance states 12..17 have no dynamics), bluerov2_dob.cpp:637-702
__device__ __forceinline__ void ekf_f12(const EkfConst& c, const double (&x)[EN], const double (&tau)[6], double (&xd)[12]) {
    double sph, cph, sth, cth, sps, cps;
    sincos_pio2(x[3], &sph, &cph);
    sincos_pio2(x[4], &sth, &cth);
    sincos_pio2(x[5], &sps, &cps);
    const double m = c.mass;
    xd[0] = (cps * cth) * x[6] + (-sps * cph + cps * sth * sph) * x[7] + (sps * sph + cps * cph * sth) * x[8];
    xd[1] = (sps * cth) * x[6] + (cps * cph + sph * sth * sps) * x[7] + (-cps * sph + sth * sps * cph) * x[8];
    xd[2] = (-sth) * x[6] + (cth * sph) * x[7] + (cth * cph) * x[8];
    const double icth = 1.0 / cth;   // one division per evaluation; the quotients of the reference differ by <= 1 ulp
    xd[3] = x[9] + (sps * sth * icth) * x[10] + cph * sth * icth * x[11];
    xd[4] = cph * x[10] + sph * x[11];
    xd[5] = (sph * icth) * x[10] + (cph * icth) * x[11];
    xd[6] = c.iMd[0] * (tau[0] + m * x[11] * x[7] - m * x[10] * x[8] - c.bo * sth + x[12] + c.Dl[0] * x[6] + c.Dnl[0] * fabs(x[6]) * x[6]);
    xd[7] = c.iMd[1] * (tau[1] - m * x[11] * x[6] + m * x[9] * x[8] + c.bo * cth * sph + x[13] + c.Dl[1] * x[7] + c.Dnl[1] * fabs(x[7]) * x[7]);
    xd[8] = c.iMd[2] * (tau[2] + m * x[10] * x[6] - m * x[9] * x[7] + c.bo * cth * cph + x[14] + c.Dl[2] * x[8] + c.Dnl[2] * fabs(x[8]) * x[8]);
    xd[9] = c.iMd[3] * (tau[3] + c.iy_iz * x[10] * x[11] - c.mzg * cth * sph + x[15] + c.Dl[3] * x[9] + c.Dnl[3] * fabs(x[9]) * x[9]);
    xd[10] = c.iMd[4] * (tau[4] + c.iz_ix * x[9] * x[11] - c.mzg * sth + x[16] + c.Dl[4] * x[10] + c.Dnl[4] * fabs(x[10]) * x[10]);
    xd[11] = c.iMd[5] * (tau[5] - c.iy_ix * x[9] * x[10] + x[17] + c.Dl[5] * x[11] + c.Dnl[5] * fabs(x[11]) * x[11]);
}

// RK4 of bluerov2_dob.cpp:621-634: classical weights, third stage at x + k2/3 (sic)
__device__ __forceinline__ void ekf_rk4(const EkfConst& c, const double (&x)[EN], const double (&tau)[6], double (&xn)[EN]) {
    double k1[12], k2[12], k3[12], k4[12], xs[EN];
#pragma unroll
    for (int i = 12; i < EN; i++) { xs[i] = x[i]; xn[i] = x[i]; }
    ekf_f12(c, x, tau, k1);
#pragma unroll
    for (int i = 0; i < 12; i++) { k1[i] *= c.dt; xs[i] = x[i] + k1[i] * 0.5; }
    ekf_f12(c, xs, tau, k2);
#pragma unroll
    for (int i = 0; i < 12; i++) { k2[i] *= c.dt; xs[i] = x[i] + k2[i] * (1.0 / 3.0); }
    ekf_f12(c, xs, tau, k3);
#pragma unroll
    for (int i = 0; i < 12; i++) { k3[i] *= c.dt; xs[i] = x[i] + k3[i]; }
    ekf_f12(c, xs, tau, k4);
#pragma unroll
    for (int i = 0; i < 12; i++) { k4[i] *= c.dt; xn[i] = x[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) * (1.0 / 6.0); }
}

// measurement model, bluerov2_dob.cpp:705-727
__device__ __forceinline__ void ekf_h(const EkfConst& c, const double (&x)[EN], const double (&a)[6], double (&y)[EN]) {
    double sph, cph, sth, cth;
    sincos_pio2(x[3], &sph, &cph);
    sincos_pio2(x[4], &sth, &cth);
    const double m = c.mass;
#pragma unroll
    for (int i = 0; i < 12; i++) y[i] = x[i];
    y[12] = c.Md[0] * a[0] - m * x[11] * x[7] + m * x[10] * x[8] + c.bo * sth - x[12] - c.Dl[0] * x[6] - c.Dnl[0] * fabs(x[6]) * x[6];
    y[13] = c.Md[1] * a[1] + m * x[11] * x[6] - m * x[9] * x[8] - c.bo * cth * sph - x[13] - c.Dl[1] * x[7] - c.Dnl[1] * fabs(x[7]) * x[7];
    y[14] = c.Md[2] * a[2] - m * x[10] * x[6] + m * x[9] * x[7] - c.bo * cth * cph - x[14] - c.Dl[2] * x[8] - c.Dnl[2] * fabs(x[8]) * x[8];
    y[15] = c.Md[3] * a[3] - c.iy_iz * x[10] * x[11] + c.mzg * cth * sph - x[15] - c.Dl[3] * x[9] - c.Dnl[3] * fabs(x[9]) * x[9];
    y[16] = c.Md[4] * a[4] - c.iz_ix * x[9] * x[11] + c.mzg * sth - x[16] - c.Dl[4] * x[10] - c.Dnl[4] * fabs(x[10]) * x[10];
    y[17] = c.Md[5] * a[5] + c.iy_ix * x[9] * x[10] - x[17] - c.Dl[5] * x[11] - c.Dnl[5] * fabs(x[11]) * x[11];
}

// row <- LDS (contiguous, 16-byte aligned: the row stride is 144 bytes)
__device__ __forceinline__ void load_row(const elds* p, double (&v)[EN]) {
#pragma unroll
    for (int j = 0; j < EN / 2; j++) { const ed2 t = ((const elds2*)p)[j]; v[2 * j] = t.x; v[2 * j + 1] = t.y; }
}
__device__ __forceinline__ void store_row(elds* p, const double (&v)[EN]) {
#pragma unroll
    for (int j = 0; j < EN / 2; j++) ((elds2*)p)[j] = ed2{v[2 * j], v[2 * j + 1]};
}
// acc[:] += sum_k a(k) * B(k,:) with a(k) = a0[k * a_sk] (lane-specific) and row k of B either contiguous at b0 + 18 k
// (B stored row-major) or strided at b0[j * 18 + k] (B^T stored row-major).  The 18 lanes of a filter read the same B row.
template <bool BT>
__device__ __forceinline__ void fetch_brow(const elds* b0, int k, double (&br)[EN]) {
    if (BT) {
#pragma unroll
        for (int j = 0; j < EN; j++) br[j] = b0[j * EN + k];
    } else {
        load_row(b0 + k * EN, br);
    }
}
// The next right-hand row is requested before the 18 FMAs of the current one, so that the LDS pipe and the VALU overlap
// (there is one wave per SIMD: nothing else hides the LDS round trip).
template <bool BT>
__device__ __forceinline__ void row_gemm(const elds* a0, int a_sk, const elds* b0, double (&acc)[EN]) {
    double br[EN], bn[EN];
    double a = a0[0], an;
    fetch_brow<BT>(b0, 0, br);
#pragma unroll 2
    for (int k = 0; k < EN; k++) {
        const int kn = k + 1 < EN ? k + 1 : EN - 1;
        an = a0[kn * a_sk];
        fetch_brow<BT>(b0, kn, bn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < EN; j++) acc[j] = fma(a, br[j], acc[j]);
        __builtin_amdgcn_sched_barrier(0);
        a = an;
#pragma unroll
        for (int j = 0; j < EN; j++) br[j] = bn[j];
    }
}

__global__ __launch_bounds__(64) void ekf_update_kernel(EkfArgs A) {
    extern __shared__ __attribute__((aligned(16))) double esm[];
    const EkfConst& c = A.c;
    const int lane = threadIdx.x;
    const int g0 = lane / kGrp;
    const bool spare = g0 >= kPerWave;             // lanes 57..63: shadow the base-evaluation lane of the third filter
    const int g = spare ? kPerWave - 1 : g0;
    const int r = spare ? EN : lane - g * kGrp;    // 0..17: row / perturbed state, 18: unperturbed evaluation
    const int inst0 = blockIdx.x * kPerWave + g;
    const bool live = inst0 < A.B;                 // the tail block recomputes the last filter and drops the result
    const int inst = live ? inst0 : A.B - 1;
    const bool row = r < EN;
    const bool base = (r == EN) && !spare;
    const int ri = row ? r : 0;                    // safe row index for the base lanes (their matrix work is discarded)

    elds* sm = (elds*)esm + g * kLdsPerFilter;
    elds* bufA = sm;              // F^T, later H^T, later V
    elds* bufB = bufA + kMat;     // G, later U, later Kal
    elds* bufC = bufB + kMat;     // P, later P_pred
    elds* bufD = bufC + kMat;     // S^-1, later J
    elds* v_xp = bufD + kMat;     // predicted state
    elds* v_y = v_xp + EN;        // measurement
    elds* v_ye = v_y + EN;        // innovation
    elds* v_xn = v_ye + EN;       // corrected state
    elds* piv = v_xn + 2 * EN;    // 2 x 40: normalised pivot rows of the Gauss-Jordan sweep (+ reciprocal pivot)

    // ---- inputs: every lane of the group keeps x, tau, acc (wave-broadcast loads)
    double x[EN], tau[6], ac[6];
    {
        const double* xg = A.x + (size_t)inst * EN;
        const double* tg = A.thrust + (size_t)inst * 6;
        const double* ag = A.acc + (size_t)inst * 6;
#pragma unroll
        for (int j = 0; j < EN; j++) x[j] = xg[j];
        double th[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { th[j] = tg[j]; ac[j] = ag[j]; }
#pragma unroll
        for (int i = 0; i < 6; i++) {   // tau = K * meas_u (bluerov2_dob.cpp:499-500)
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) s += c.K[i * 6 + j] * th[j];
            tau[i] = s;
        }
        if (base) {
            const double* yg = A.y12 + (size_t)inst * 12;
#pragma unroll
            for (int j = 0; j < 12; j++) v_y[j] = yg[j];
#pragma unroll
            for (int j = 0; j < 6; j++) v_y[12 + j] = tau[j];
        }
        // P row i -> LDS (left operand of the first product)
        if (row) {
            const double* pg = A.P + (size_t)inst * kMat + (size_t)r * EN;
            double pr[EN];
#pragma unroll
            for (int j = 0; j < EN; j++) pr[j] = pg[j];
            store_row(bufC + r * EN, pr);
        }
    }

    // ---- F = d RK4 / d x by forward differences (column r in lane r), x_pred
    double col[EN];
    {
        double xl[EN], xn[EN];
#pragma unroll
        for (int j = 0; j < EN; j++) xl[j] = x[j] + ((j == r) ? c.d : 0.0);
        ekf_rk4(c, xl, tau, xn);
        if (base) store_row(v_xp, xn);
        __syncthreads();
        double f0[EN];
        load_row(v_xp, f0);
#pragma unroll
        for (int j = 0; j < EN; j++) col[j] = (xn[j] - f0[j]) * c.inv_d;
        if (row) store_row(bufA + r * EN, col);   // row r of F^T
#pragma unroll
        for (int j = 0; j < EN; j++) x[j] = f0[j];  // from here on x = x_pred
    }
    __syncthreads();

    double acc[EN];
    // G = P F^T  (left: P row i in bufC, right: F^T row-major in bufA)
#pragma unroll
    for (int j = 0; j < EN; j++) acc[j] = 0.0;
    row_gemm<false>(bufC + ri * EN, 1, bufA, acc);
    if (row) store_row(bufB + r * EN, acc);
    __syncthreads();
    // P_pred = F G + Q  (left: F[i][k] = F^T[k][i])
#pragma unroll
    for (int j = 0; j < EN; j++) acc[j] = (j == r) ? c.Q[j] : 0.0;
    row_gemm<false>(bufA + ri, EN, bufB, acc);
    if (row) store_row(bufC + r * EN, acc);      // P (staged) is dead: every lane read only its own row
    // ---- H = d h / d x at x_pred by forward differences, innovation
    {
        double xl[EN], yv[EN];
#pragma unroll
        for (int j = 0; j < EN; j++) xl[j] = x[j] + ((j == r) ? c.d : 0.0);
        ekf_h(c, xl, ac, yv);
        __syncthreads();                           // all reads of F^T (bufA) are done
        if (base) {
            double ym[EN], ye[EN];
            load_row(v_y, ym);
#pragma unroll
            for (int j = 0; j < EN; j++) ye[j] = ym[j] - yv[j];
            store_row(v_ye, ye);
            store_row(v_xn, yv);                   // y_pred, parked in the x_new slot until the gain exists
        }
        __syncthreads();
        double y0[EN];
        load_row(v_xn, y0);
#pragma unroll
        for (int j = 0; j < EN; j++) col[j] = (yv[j] - y0[j]) * c.inv_d;
        if (row) store_row(bufA + r * EN, col);    // row r of H^T
    }
    __syncthreads();
    // U = P_pred H^T  (left: P_pred row i in bufC, right: H^T row-major)
#pragma unroll
    for (int j = 0; j < EN; j++) acc[j] = 0.0;
    row_gemm<false>(bufC + ri * EN, 1, bufA, acc);
    if (row) store_row(bufB + r * EN, acc);      // G is dead
    __syncthreads();
    // S = H U + R  (left: H[i][k] = H^T[k][i])
    double s[EN], t[EN];
#pragma unroll
    for (int j = 0; j < EN; j++) s[j] = (j == r) ? c.R : 0.0;
    row_gemm<false>(bufA + ri, EN, bufB, s);
    // ---- S^-1 by Gauss-Jordan without pivoting (S = H P H^T + R I is symmetric positive definite); lane i holds row i of
    // [S | I].  At step k only columns k.. of the left block and 0..k of the right block are non-trivial.  The pivot lane
    // publishes its row as it is (nothing in front of the LDS write); every lane forms the reciprocal pivot itself.
#pragma unroll
    for (int j = 0; j < EN; j++) t[j] = (j == r) ? 1.0 : 0.0;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < EN; k++) {
        elds* pb = piv + (k & 1) * 40;
        if (r == k) {
#pragma unroll
            for (int j = k; j < EN; j++) pb[j] = s[j];
#pragma unroll
            for (int j = 0; j < k; j++) pb[EN + j] = t[j];
        }
        __syncthreads();
        const double pk = pb[k];
        ok = ok && (pk > 0.0) && (pk < 1e300);
        double ip = __builtin_amdgcn_rcp(pk);   // v_rcp_f64 + 2 Newton steps (~1 ulp)
        double e1 = fma(-pk, ip, 1.0);
        ip = fma(ip, e1, ip);
        e1 = fma(-pk, ip, 1.0);
        ip = fma(ip, e1, ip);
        const bool me = (r == k);
        // row_i -= (s_ik / p) * row_k for i != k;  row_k *= 1/p  (written as row_k -= (1 - 1/p) row_k: one code path)
        const double f = me ? 1.0 - ip : s[k] * ip;
#pragma unroll
        for (int j = k + 1; j < EN; j++) s[j] = fma(-f, pb[j], s[j]);
#pragma unroll
        for (int j = 0; j < k; j++) t[j] = fma(-f, pb[EN + j], t[j]);
        t[k] = me ? ip : -f;                      // column k of the right block was e_k
    }
    if (row) store_row(bufD + r * EN, t);
    __syncthreads();
    // Kal = U S^-1
    double kal[EN];
#pragma unroll
    for (int j = 0; j < EN; j++) kal[j] = 0.0;
    row_gemm<false>(bufB + ri * EN, 1, bufD, kal);
    if (!ok) {   // innovation covariance not positive definite (or NaN): keep the prediction, P := P_pred
#pragma unroll
        for (int j = 0; j < EN; j++) kal[j] = 0.0;
    }
    // x_new = x_pred + Kal (y - y_pred)
    {
        double ye[EN];
        load_row(v_ye, ye);
        double dxi = 0.0;
#pragma unroll
        for (int j = 0; j < EN; j++) dxi = fma(kal[j], ye[j], dxi);
        double xi = 0.0;
#pragma unroll
        for (int j = 0; j < EN; j++) xi = (j == r) ? x[j] : xi;
        xi += dxi;
        __syncthreads();                           // y_pred (parked in v_xn) has been consumed by every lane
        if (row) {
            v_xn[r] = xi;
            if (live) A.x[(size_t)inst * EN + r] = xi;
            store_row(bufB + r * EN, kal);         // U row i was read by lane i only
        }
    }
    __syncthreads();
    // J = I - Kal H  (right: H[k][j] = H^T[j][k], strided)
#pragma unroll
    for (int j = 0; j < EN; j++) acc[j] = 0.0;
    row_gemm<true>(bufB + ri * EN, 1, bufA, acc);
#pragma unroll
    for (int j = 0; j < EN; j++) acc[j] = ((j == r) ? 1.0 : 0.0) - (ok ? acc[j] : 0.0);   // !ok: J = I exactly (H may hold NaN)
    if (row) store_row(bufD + r * EN, acc);      // S^-1 is dead (consumed before the barrier above)
    __syncthreads();
    // V = J P_pred
#pragma unroll
    for (int j = 0; j < EN; j++) acc[j] = 0.0;
    row_gemm<false>(bufD + ri * EN, 1, bufC, acc);
    if (row) store_row(bufA + r * EN, acc);      // H^T is dead
    __syncthreads();
    // P_new = V J^T + R Kal Kal^T   (Joseph form, bluerov2_dob.cpp:537)
    double pn[EN], kk[EN];
#pragma unroll
    for (int j = 0; j < EN; j++) { pn[j] = 0.0; kk[j] = 0.0; }
    row_gemm<true>(bufA + ri * EN, 1, bufD, pn);
    row_gemm<true>(bufB + ri * EN, 1, bufB, kk);
    if (row && live) {
        double* pg = A.P + (size_t)inst * kMat + (size_t)r * EN;
#pragma unroll
        for (int j = 0; j < EN; j++) pg[j] = fma(c.R, kk[j], pn[j]);
    }
    // ---- outputs: world-frame disturbance with the MEASURED attitude (:540-545), NMPC parameters (:334-337)
    if (base && live) {
        double xn[EN], ym[EN];
        load_row(v_xn, xn);
        load_row(v_y, ym);
        double sph, cph, sth, cth, sps, cps;
        sincos_pio2(ym[3], &sph, &cph);
        sincos_pio2(ym[4], &sth, &cth);
        sincos_pio2(ym[5], &sps, &cps);
        double* w = A.wf + (size_t)inst * 6;
        w[0] = (cps * cth) * xn[12] + (-sps * cph + cps * sth * sph) * xn[13] + (sps * sph + cps * cph * sth) * xn[14];
        w[1] = (sps * cth) * xn[12] + (cps * cph + sph * sth * sps) * xn[13] + (-cps * sph + sth * sps * cph) * xn[14];
        w[2] = (-sth) * xn[12] + (cth * sph) * xn[13] + (cth * cph) * xn[14];
        w[3] = xn[15] + (sps * sth / cth) * xn[16] + cph * sth / cth * xn[17];
        w[4] = cph * xn[16] + sph * xn[17];
        w[5] = (sph / cth) * xn[16] + (cph / cth) * xn[17];
        double* mp = A.mp + (size_t)inst * 4;
        mp[0] = xn[12] * c.inv_cc;
        mp[1] = xn[13] * c.inv_cc;
        mp[2] = xn[14] * c.inv_rc;
        mp[3] = xn[17] * c.inv_rc;
        bool fin = true;
#pragma unroll
        for (int j = 0; j < EN; j++) fin = fin && (fabs(xn[j]) < 1e300);
        A.status[inst] = !ok ? 1 : (fin ? 0 : 2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// DPP variant (dense; default until round 5, BROV_EKF_VARIANT=1 since).  The LDS-broadcast kernel above is bound by LDS bandwidth: every lane streams the whole right-hand
// matrix of every product out of LDS (0.5 MB per update against 128 B/clk per CU).  Here a filter lives in ONE 16-lane DPP
// row (4 filters per wave) and the right-hand rows are broadcast straight out of the registers of the lane that owns them:
//     v_fmac_f64_dpp acc, B_row_reg, a   row_newbcast:k      (acc += a * (B_row_reg of lane k of my 16-lane row))
// -- the only DPP control gfx950 offers for 64-bit operands, and exactly the one a row-times-matrix product needs.
// 18 = 16 + 2: lane l owns row l ("primary"), lanes 0 and 1 additionally own rows 16 and 17 ("secondary" register set, zero
// in the other lanes); a product is 18 x 18 fmacs for each set.  The left operand's row elements come out of LDS (or a
// register with a static index), transposes are strided LDS reads, three 18x18 LDS buffers per filter.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kDppFilters = 4;
constexpr int kDppLds = 3 * kMat + 4 * EN;       // doubles per filter: buffers A, B, C + x_new, x_pred, innovation, measurement
constexpr int kSpLds = 2 * kMat + 4 * EN;        // ekf_update_kernel_sp: two buffers (left operands that are a lane's own rows come out of registers)

struct Rows { double p[EN], s[EN]; };   // row layout: p = row l of the matrix, s = row 16 + l (lanes 0, 1), else 0

template <int K>
__device__ __forceinline__ void fmac_bc(double& acc, double src, double a) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(a), "n"(K));
}
// value of `src` in lane K of this lane's 16-lane row (s_nop: a VALU write must be 2 wait states ahead of a DPP read, and the
// compiler does not see through inline assembly)
template <int K>
__device__ __forceinline__ double bcast(double src) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(K));
    return r;
}
template <int... Ks, class F>
__device__ __forceinline__ void for_k(std::integer_sequence<int, Ks...>, F f) { (f(std::integral_constant<int, Ks>{}), ...); }

// A register that a DPP instruction reads must have been written at least two wait states earlier (the hardware does not interlock
// that case), and the compiler does not see through inline assembly: it is free to sink the instruction that PRODUCES a row element
// (a select, the last multiply of a finite difference) down to just in front of the asm statement that reads it through DPP -- met in
// round 6, where a changed operand form moved the finite-difference arithmetic of F^T into the first product and the product read
// stale registers.  pin_rows() is an empty volatile asm with every element as an in-out operand: volatile asm statements keep their
// order, so everything that produces the rows is issued before it and the `s_nop 1` behind it; the opaque result cannot be
// rematerialised later either.  No instruction is emitted for it.
__device__ __forceinline__ void pin_rows(const Rows& Rc) {
    Rows& R = const_cast<Rows&>(Rc);
#pragma unroll
    for (int j = 0; j < EN; j++) asm volatile("" : "+v"(R.p[j]), "+v"(R.s[j]));
    asm volatile("s_nop 1");
}

// C += A B.  B in row layout (registers); ap(k) / as(k) deliver A[l][k] and A[16+l][k] of this lane (as(k) = 0 for l >= 2).
// The A elements of a block of k's are requested (LDS) before the fmacs of the previous block are issued.
// CORNER: only columns 16, 17 of the secondary rows are accumulated -- for a symmetric result the rest of rows 16, 17 is
// columns 16, 17 of the primary rows.
template <int K0, int NK, bool SECB, bool CORNER>
__device__ __forceinline__ void gemm_block(Rows& C, const Rows& B, const double (&a0)[NK], const double (&a1)[NK]) {
    for_k(std::make_integer_sequence<int, NK>{}, [&](auto kc) {
        constexpr int KK = decltype(kc)::value;
        constexpr int K = K0 + KK;            // lane that owns the B row
#pragma unroll
        for (int j = 0; j < EN; j++) {
            fmac_bc<K>(C.p[j], SECB ? B.s[j] : B.p[j], a0[KK]);
            if (!CORNER || j >= 16) fmac_bc<K>(C.s[j], SECB ? B.s[j] : B.p[j], a1[KK]);
        }
    });
}
template <bool CORNER = false, class AP, class AS>
__device__ __forceinline__ void gemm_dpp(Rows& C, const Rows& B, AP ap, AS as) {
    double p0[6], s0[6], p1[6], s1[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { p0[k] = ap(k); s0[k] = as(k); }
#pragma unroll
    for (int k = 0; k < 6; k++) { p1[k] = ap(6 + k); s1[k] = as(6 + k); }
    pin_rows(B);
    gemm_block<0, 6, false, CORNER>(C, B, p0, s0);
#pragma unroll
    for (int k = 0; k < 4; k++) { p0[k] = ap(12 + k); s0[k] = as(12 + k); }
    p0[4] = ap(16); s0[4] = as(16); p0[5] = ap(17); s0[5] = as(17);
    gemm_block<6, 6, false, CORNER>(C, B, p1, s1);
    {
        const double q0[4] = {p0[0], p0[1], p0[2], p0[3]}, q1[4] = {s0[0], s0[1], s0[2], s0[3]};
        gemm_block<12, 4, false, CORNER>(C, B, q0, q1);
        const double r0[2] = {p0[4], p0[5]}, r1[2] = {s0[4], s0[5]};
        gemm_block<0, 2, true, CORNER>(C, B, r0, r1);
    }
}
// Symmetric result computed with CORNER = true: rows 16, 17 (secondary set of lanes 0, 1), columns 0..15, are columns 16, 17 of
// the primary rows -- lane 0 / 1 gathers them with 2 x 16 row broadcasts instead of 2 x 16 x 18 fmacs issued for two lanes.
// ZS = false: lanes >= 2 receive lane 1's values instead of zeros (their secondary set is never read: every DPP read of a secondary row
// names lane 0 or 1, and only lanes 0, 1 store theirs) -- one select per element instead of two
template <bool ZS = true>
__device__ __forceinline__ void fill_secondary_from_symmetry(Rows& C, int l) {
    for_k(std::make_integer_sequence<int, 16>{}, [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const double t16 = bcast<J>(C.p[16]), t17 = bcast<J>(C.p[17]);
        C.s[J] = (l == 0) ? t16 : ((ZS && l != 1) ? 0.0 : t17);
    });
}
// C += A B for a SYMMETRIC right operand B, secondary rows of C done the cheap way: C[16+r][j] = sum_k A[16+r][k] B[k][j] and
// B[k][j] = B[j][k] is element k of the row lane j owns, so lane j computes C[16][j] and C[17][j] with 2 x 18 fmacs on its own
// registers (a16(k), a17(k): rows 16, 17 of A, identical for the 16 lanes of a filter), then lanes 0 / 1 gather their rows with
// row broadcasts.  Columns 16, 17 of those rows use B's secondary rows (lanes 0, 1).  init16 / init17: C[16][j], C[17][j] to add to.
template <bool ZS = true, class AP, class A16, class A17>
__device__ __forceinline__ void gemm_dpp_symB(Rows& C, const Rows& B, AP ap, A16 a16, A17 a17, int l) {
    gemm_dpp<true>(C, B, ap, [&](int) { return 0.0; });     // primary rows; the secondary set is rebuilt below
    double c16 = 0.0, c17 = 0.0, d16[2] = {0.0, 0.0}, d17[2] = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < EN; k++) {
        const double x16 = a16(k), x17 = a17(k);
        c16 = fma(x16, B.p[k], c16);          // lane j < 16: column j of rows 16 / 17
        c17 = fma(x17, B.p[k], c17);
        // columns 16, 17: B[k][16 + m] = B[16 + m][k] = B.s[k] of lane m
        d16[0] = fma(x16, B.s[k], d16[0]);    // valid in lanes 0 (column 16) and 1 (column 17)
        d17[0] = fma(x17, B.s[k], d17[0]);
    }
    for_k(std::make_integer_sequence<int, 16>{}, [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const double t16 = bcast<J>(c16), t17 = bcast<J>(c17);
        C.s[J] = (l == 0) ? t16 : ((ZS && l != 1) ? 0.0 : t17);
    });
    {   // corner: lane 0 needs C[16][16] (its own d16), C[16][17] (lane 1's d16); lane 1 needs C[17][16] (lane 0's d17), C[17][17]
        const double e16_0 = bcast<0>(d16[0]), e16_1 = bcast<1>(d16[0]), e17_0 = bcast<0>(d17[0]), e17_1 = bcast<1>(d17[0]);
        C.s[16] = (l == 0) ? e16_0 : ((ZS && l != 1) ? 0.0 : e17_0);
        C.s[17] = (l == 0) ? e16_1 : ((ZS && l != 1) ? 0.0 : e17_1);
    }
    (void)d16[1]; (void)d17[1];
}
__device__ __forceinline__ void zero_rows(Rows& R) {
#pragma unroll
    for (int j = 0; j < EN; j++) { R.p[j] = 0.0; R.s[j] = 0.0; }
}
// row layout <-> LDS (row-major 18 x 18): lane l stores / loads row l, lanes 0, 1 also rows 16, 17
__device__ __forceinline__ void store_rows(elds* m, const Rows& R, int l) {
    store_row(m + l * EN, R.p);
    if (l < 2) store_row(m + (16 + l) * EN, R.s);
}
template <bool ZS = true>
__device__ __forceinline__ void load_rows(const elds* m, Rows& R, int l) {
    load_row(m + l * EN, R.p);
    double t[EN];
    load_row(m + (16 + (l < 2 ? l : 0)) * EN, t);
#pragma unroll
    for (int j = 0; j < EN; j++) R.s[j] = (!ZS || l < 2) ? t[j] : 0.0;
}
// rows of the TRANSPOSE of the matrix stored row-major at m (strided reads)
template <bool ZS = true>
__device__ __forceinline__ void load_rows_t(const elds* m, Rows& R, int l) {
#pragma unroll
    for (int j = 0; j < EN; j++) { R.p[j] = m[j * EN + l]; const double t = m[j * EN + 16 + (l < 2 ? l : 0)]; R.s[j] = (!ZS || l < 2) ? t : 0.0; }
}

__global__ __launch_bounds__(64, 1) void ekf_update_kernel_dpp(EkfArgs A) {
    extern __shared__ __attribute__((aligned(16))) double esm[];
    const EkfConst& c = A.c;
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    const int inst0 = blockIdx.x * kDppFilters + g;
    const bool live = inst0 < A.B;
    const int inst = live ? inst0 : A.B - 1;
    const bool sec = l < 2;              // owns a secondary row
    const int ls = sec ? l : 0;

    elds* sm = (elds*)esm + g * kDppLds;
    elds* bufA = sm;
    elds* bufB = bufA + kMat;
    elds* bufC = bufB + kMat;
    elds* v_xn = bufC + kMat;            // [18] corrected state (gathered for the output lane)
    elds* v_xp = v_xn + EN;              // [18] predicted state
    elds* v_ye = v_xp + EN;              // [18] innovation
    elds* v_ym = v_ye + EN;              // [18] measurement

    // ---- inputs: every lane keeps x, tau, acc, y of its filter
    double x[EN], tau[6], ac[6], ym[EN];
    {
        const double* xg = A.x + (size_t)inst * EN;
        const double* tg = A.thrust + (size_t)inst * 6;
        const double* ag = A.acc + (size_t)inst * 6;
        const double* yg = A.y12 + (size_t)inst * 12;
#pragma unroll
        for (int j = 0; j < EN; j++) x[j] = xg[j];
        double th[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { th[j] = tg[j]; ac[j] = ag[j]; }
#pragma unroll
        for (int i = 0; i < 6; i++) {   // tau = K * meas_u (bluerov2_dob.cpp:499-500)
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) t += c.K[i * 6 + j] * th[j];
            tau[i] = t;
        }
#pragma unroll
        for (int j = 0; j < 12; j++) ym[j] = yg[j];
#pragma unroll
        for (int j = 0; j < 6; j++) ym[12 + j] = tau[j];
        store_row(v_ym, ym);   // identical values from the 16 lanes of the filter
        // P -> bufA (left operand of the first product): the filter's 2592 contiguous bytes in 16-byte pieces, 16 lanes wide
        const ed2* pg = (const ed2*)(A.P + (size_t)inst * kMat);
        ed2 pv[11];
#pragma unroll
        for (int m = 0; m < 11; m++) { const int q = m * 16 + l; pv[m] = pg[q < kMat / 2 ? q : 0]; }
#pragma unroll
        for (int m = 0; m < 11; m++) { const int q = m * 16 + l; if (q < kMat / 2) ((elds2*)bufA)[q] = pv[m]; }
    }

    // ---- F^T by forward differences of the RK4 map: lane l perturbs state l; lanes 0, 1 also states 16, 17; lane 2's second
    // evaluation is the unperturbed one and is broadcast
    Rows Ft;
    {
        double xl[EN], xa[EN], xb[EN];
#pragma unroll
        for (int j = 0; j < EN; j++) xl[j] = x[j] + ((sec && j == 16 + l) ? c.d : 0.0);
        ekf_rk4(c, xl, tau, xa);            // lanes 0, 1: perturbed in 16 / 17; all other lanes: unperturbed
#pragma unroll
        for (int j = 0; j < EN; j++) xl[j] = x[j] + ((j == l) ? c.d : 0.0);
        ekf_rk4(c, xl, tau, xb);
        double f0[EN];
#pragma unroll
        for (int j = 0; j < EN; j++) f0[j] = bcast<2>(xa[j]);
#pragma unroll
        for (int j = 0; j < EN; j++) {
            Ft.p[j] = (xb[j] - f0[j]) * c.inv_d;
            Ft.s[j] = sec ? (xa[j] - f0[j]) * c.inv_d : 0.0;
            x[j] = f0[j];                   // from here on x = x_pred
        }
        store_row(v_xp, f0);
    }
    __syncthreads();
    // G = P F^T
    Rows G;
    zero_rows(G);
    gemm_dpp(G, Ft, [&](int k) { return bufA[l * EN + k]; }, [&](int k) { const double t = bufA[(16 + ls) * EN + k]; return sec ? t : 0.0; });
    store_rows(bufC, Ft, l);                // F^T row-major: F[l][k] = bufC[k][l]
    __syncthreads();
    // P_pred = F G + Q
    Rows Pq;
#pragma unroll
    for (int j = 0; j < EN; j++) { Pq.p[j] = (j == l) ? c.Q[j] : 0.0; Pq.s[j] = (sec && j == 16 + l) ? c.Q[j] : 0.0; }
    gemm_dpp<true>(Pq, G, [&](int k) { return bufC[k * EN + l]; }, [&](int k) { const double t = bufC[k * EN + 16 + ls]; return sec ? t : 0.0; });
    fill_secondary_from_symmetry(Pq, l);    // P_pred is symmetric: only the 2x2 corner of rows 16, 17 was accumulated
    store_rows(bufB, Pq, l);                // P_pred stays in bufB (right operand of V = J P_pred)
    // ---- H^T by forward differences of h at x_pred, innovation
    Rows Ht;
    {
        double xl[EN], y0[EN], ya[EN], yb[EN], ye[EN], ym2[EN];
        load_row(v_ym, ym2);
        ekf_h(c, x, ac, y0);
#pragma unroll
        for (int j = 0; j < EN; j++) xl[j] = x[j] + ((sec && j == 16 + l) ? c.d : 0.0);
        ekf_h(c, xl, ac, ya);
#pragma unroll
        for (int j = 0; j < EN; j++) xl[j] = x[j] + ((j == l) ? c.d : 0.0);
        ekf_h(c, xl, ac, yb);
#pragma unroll
        for (int j = 0; j < EN; j++) {
            Ht.p[j] = (yb[j] - y0[j]) * c.inv_d;
            Ht.s[j] = sec ? (ya[j] - y0[j]) * c.inv_d : 0.0;
            ye[j] = ym2[j] - y0[j];
        }
        store_row(v_ye, ye);
    }
    __syncthreads();                        // reads of bufC (F^T) are done
    store_rows(bufC, Ht, l);                // H^T row-major: H[l][k] = bufC[k][l]
    __syncthreads();
    // W = H P_pred  (= (P_pred H^T)^T; right operand: P_pred, still in registers)
    Rows W;
    zero_rows(W);
    gemm_dpp_symB(W, Pq, [&](int k) { return bufC[k * EN + l]; }, [&](int k) { return bufC[k * EN + 16]; }, [&](int k) { return bufC[k * EN + 17]; }, l);
    store_rows(bufA, W, l);                 // P is dead
    __syncthreads();
    // S = W H^T + R
    Rows S;
#pragma unroll
    for (int j = 0; j < EN; j++) { S.p[j] = (j == l) ? c.R : 0.0; S.s[j] = (sec && j == 16 + l) ? c.R : 0.0; }
    gemm_dpp<true>(S, Ht, [&](int k) { return bufA[l * EN + k]; }, [&](int k) { const double t = bufA[(16 + ls) * EN + k]; return sec ? t : 0.0; });
    fill_secondary_from_symmetry(S, l);     // S is symmetric
    // ---- K^T = S^-1 W by Gauss-Jordan on [S | W] without pivoting (S is SPD); the pivot row is broadcast with DPP.
    // Secondary rows of lanes >= 2 are zero and stay zero (their factor is 0).
    Rows T;
    load_rows(bufA, T, l);
    bool ok = true;
    pin_rows(S); pin_rows(T);                 // (fill_secondary_from_symmetry / load_rows end in plain selects)
    for_k(std::make_integer_sequence<int, EN>{}, [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        constexpr bool PS = K >= 16;          // pivot row in the secondary set (lanes 0, 1)
        constexpr int KL = PS ? K - 16 : K;   // lane that owns the pivot row
        const double pk = PS ? bcast<KL>(S.s[K]) : bcast<KL>(S.p[K]);
        ok = ok && (pk > 0.0) && (pk < 1e300);
        double ip = __builtin_amdgcn_rcp(pk);
        double e1 = fma(-pk, ip, 1.0);
        ip = fma(ip, e1, ip);
        e1 = fma(-pk, ip, 1.0);
        ip = fma(ip, e1, ip);
        // row_i -= (s_ik / p) row_k for i != k;  row_k *= 1/p, written as row_k -= (1 - 1/p) row_k.  The register set that does
        // NOT hold the pivot row is updated first: both updates read the pivot row, and the second one overwrites it.
        const double nfp = (!PS && l == KL) ? ip - 1.0 : -(S.p[K] * ip);
        const double nfs = (PS && l == KL) ? ip - 1.0 : -(S.s[K] * ip);
#pragma unroll
        for (int j = K + 1; j < EN; j++) {
            if (PS) { fmac_bc<KL>(S.p[j], S.s[j], nfp); fmac_bc<KL>(S.s[j], S.s[j], nfs); }
            else { fmac_bc<KL>(S.s[j], S.p[j], nfs); fmac_bc<KL>(S.p[j], S.p[j], nfp); }
        }
#pragma unroll
        for (int j = 0; j < EN; j++) {
            if (PS) { fmac_bc<KL>(T.p[j], T.s[j], nfp); fmac_bc<KL>(T.s[j], T.s[j], nfs); }
            else { fmac_bc<KL>(T.s[j], T.p[j], nfs); fmac_bc<KL>(T.p[j], T.p[j], nfp); }
        }
    });
    if (!ok) zero_rows(T);                  // innovation covariance not positive definite / NaN: keep the prediction
    __syncthreads();                        // reads of bufA (W) are done
    store_rows(bufA, T, l);                 // K^T row-major: Kal[l][k] = bufA[k][l]
    __syncthreads();
    // x_new = x_pred + Kal (y - y_pred)
    {
        double dp = 0.0, ds = 0.0, ye[EN];
        load_row(v_ye, ye);
#pragma unroll
        for (int k = 0; k < EN; k++) { dp = fma(bufA[k * EN + l], ye[k], dp); ds = fma(bufA[k * EN + 16 + ls], ye[k], ds); }
        const double xp_ = v_xp[l] + dp, xs_ = v_xp[16 + ls] + ds;
        v_xn[l] = xp_;
        if (sec) v_xn[16 + l] = xs_;
        if (live) {
            A.x[(size_t)inst * EN + l] = xp_;
            if (sec) A.x[(size_t)inst * EN + 16 + l] = xs_;
        }
    }
    // J = I - Kal H   (right operand: H in row layout = columns of H^T)
    Rows H;
    load_rows_t(bufC, H, l);
    Rows J;
    zero_rows(J);
    gemm_dpp(J, H, [&](int k) { return -bufA[k * EN + l]; }, [&](int k) { const double t = -bufA[k * EN + 16 + ls]; return sec ? t : 0.0; });
#pragma unroll
    for (int j = 0; j < EN; j++) {
        if (!ok) { J.p[j] = 0.0; J.s[j] = 0.0; }   // H may hold NaN: J = I exactly
        J.p[j] += (j == l) ? 1.0 : 0.0;
        J.s[j] += (sec && j == 16 + l) ? 1.0 : 0.0;
    }
    __syncthreads();                        // reads of bufC (H^T) are done
    store_rows(bufC, J, l);                 // J row-major
    __syncthreads();
    // V = J P_pred
    Rows V;
    {
        Rows Pr;
        load_rows(bufB, Pr, l);
        zero_rows(V);
        gemm_dpp_symB(V, Pr, [&](int k) { return bufC[l * EN + k]; }, [&](int k) { return bufC[16 * EN + k]; }, [&](int k) { return bufC[17 * EN + k]; }, l);
    }
    __syncthreads();                        // reads of bufB (P_pred) are done
    store_rows(bufB, V, l);
    __syncthreads();
    // P_new = V J^T + R Kal Kal^T   (Joseph form, bluerov2_dob.cpp:537)
    Rows Pn;
    zero_rows(Pn);
    {
        Rows Jt;
        load_rows_t(bufC, Jt, l);
        gemm_dpp<true>(Pn, Jt, [&](int k) { return bufB[l * EN + k]; }, [&](int k) { const double t = bufB[(16 + ls) * EN + k]; return sec ? t : 0.0; });
        Rows Kt;
        load_rows(bufA, Kt, l);
        gemm_dpp<true>(Pn, Kt, [&](int k) { return c.R * bufA[k * EN + l]; }, [&](int k) { const double t = c.R * bufA[k * EN + 16 + ls]; return sec ? t : 0.0; });
    }
    if (live) {   // P_new is symmetric: rows 16, 17 are columns 16, 17 of the primary rows (+ the 2x2 corner from lanes 0, 1)
        double* pg = A.P + (size_t)inst * kMat;
#pragma unroll
        for (int j = 0; j < EN; j++) pg[l * EN + j] = Pn.p[j];
        pg[16 * EN + l] = Pn.p[16];
        pg[17 * EN + l] = Pn.p[17];
        if (sec) { pg[(16 + l) * EN + 16] = Pn.s[16]; pg[(16 + l) * EN + 17] = Pn.s[17]; }
    }
    // ---- outputs: world-frame disturbance with the MEASURED attitude (:540-545), NMPC parameters (:334-337)
    __syncthreads();
    if (l == 0 && live) {
        double xn[EN];
        load_row(v_xn, xn);
        double sph, cph, sth, cth, sps, cps;
        sincos_pio2(v_ym[3], &sph, &cph);
        sincos_pio2(v_ym[4], &sth, &cth);
        sincos_pio2(v_ym[5], &sps, &cps);
        double* w = A.wf + (size_t)inst * 6;
        w[0] = (cps * cth) * xn[12] + (-sps * cph + cps * sth * sph) * xn[13] + (sps * sph + cps * cph * sth) * xn[14];
        w[1] = (sps * cth) * xn[12] + (cps * cph + sph * sth * sps) * xn[13] + (-cps * sph + sth * sps * cph) * xn[14];
        w[2] = (-sth) * xn[12] + (cth * sph) * xn[13] + (cth * cph) * xn[14];
        w[3] = xn[15] + (sps * sth / cth) * xn[16] + cph * sth / cth * xn[17];
        w[4] = cph * xn[16] + sph * xn[17];
        w[5] = (sph / cth) * xn[16] + (cph / cth) * xn[17];
        double* mp = A.mp + (size_t)inst * 4;
        mp[0] = xn[12] * c.inv_cc;
        mp[1] = xn[13] * c.inv_cc;
        mp[2] = xn[14] * c.inv_rc;
        mp[3] = xn[17] * c.inv_rc;
        bool fin = true;
#pragma unroll
        for (int j = 0; j < EN; j++) fin = fin && (fabs(xn[j]) < 1e300);
        A.status[inst] = !ok ? 1 : (fin ? 0 : 2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Structured variant of the DPP kernel (default since round 6; ekf_update_kernel_dpp stays behind BROV_EKF_VARIANT=1).
// Same mapping (one filter per 16-lane DPP row, rows of the right operand broadcast out of registers), same arithmetic in
// the same order -- what it leaves out are operations whose operand is an EXACT zero of a finite-difference Jacobian, and
// function evaluations whose result is known without them:
//   * a component of the RK4 map / of h that does not depend on the perturbed state comes out bit-identical in the perturbed
//     and the unperturbed evaluation (same instruction stream, same inputs), so its forward difference is exactly 0.
//     F = d(RK4)/dx: nothing depends on the position (bluerov2_dob.cpp:637-702 read x(3..17) only), so columns 0..2 are
//     e_r ((x_r + d) + inc_r - (x_r + inc_r)) / d with the increment of the UNPERTURBED evaluation; rows 12..17 (disturbances: no
//     dynamics) are e_r ((x_r + d) - x_r) / d.  H = dh/dx (:705-727): rows 0..11 are the states themselves, rows 12..17 depend on
//     phi, theta, the six velocities and their own disturbance.
//   * so the 19 evaluations of the RK4 map are 16: lane 0 evaluates the unperturbed map, lanes 1 / 2 the ones perturbed in
//     states 16 / 17, lanes 3..15 their own state, and columns 0..2 follow from lane 0's increment -- ONE pass of ekf_rk4 per
//     wave instead of two (the second one ran for three useful lanes of sixteen).  The same for h (one pass instead of three).
//   * the products that have F, H or their transposes as an operand issue only the multiply-adds of the non-zero pattern:
//     G = P F' 378 (648), P_pred = F G 292 (360), W = H P_pred 234 (468), S = W H' 84 (360), J = I - Kal H 132 (648) DPP
//     multiply-adds per wave; Gauss-Jordan, V = J P_pred and the Joseph form are dense and unchanged.
//   * left operands that are a lane's own rows (W in S = W H', V in P_new = V J') come out of the lane's registers, W is eliminated in
//     place, P and P_pred share a buffer and so do F', H' and K': TWO 18 x 18 LDS buffers per filter instead of three -- 22.5 KB per
//     wave, seven waves per CU -- and the leaner products need 224 registers: two waves per SIMD (the dense kernel: 264, one).
//   * the secondary register set (rows 16, 17) of lanes 2..15 holds don't-care values here (the dense kernel keeps it zero with two
//     selects per element): every DPP read of a secondary row names lane 0 or 1, and only those lanes store theirs.
//   * every product pins its broadcast rows first (pin_rows): see the hazard note there and scripts/check_dpp_hazard.py.
// Skipping a multiply-add whose product is an exact zero leaves the accumulator as it is (up to the sign of a zero); what is NOT
// bit-identical to the dense kernel is the code the compiler makes of the RK4 map here (one evaluation + the position increments
// instead of two evaluations), and a last-bit difference there is a 1e-10 relative difference in F.  The two kernels agree like any
// two FP64 evaluations of the filter do (tests/test_gpu_ekf.py: 1e-6 after a tick, against each other and against the oracle).
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr bool ekf_h21col(int k) { return k == 3 || k == 4 || (k >= 6 && k < 12); }   // columns of H's rows 12..17 (besides their own)
__host__ __device__ constexpr int ekf_h21idx(int k) { return k < 5 ? k - 3 : k - 4; }                           // 3, 4, 6..11 -> 0..7
struct NzFt { static constexpr bool at(int k, int j) { return k < 3 ? j == k : (j < 12 || j == k); } };      // F'[k][j] = F[j][k]
struct NzHt { static constexpr bool at(int k, int j) { return j == k || (j >= 12 && ekf_h21col(k)); } };     // H'[k][j] = H[j][k]

// both halves of a double through 32-bit DPP moves (the 64-bit DPP move knows row_newbcast only); lanes whose source is outside
// the 16-lane row receive 0
template <int CTRL>
__device__ __forceinline__ double dpp32_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ekf_rk4 and, for the position components, the result the same evaluation would have with x_i + d in place of x_i (the
// stages do not read the position): the forward difference of columns 0..2 without evaluating them
__device__ __forceinline__ void ekf_rk4_fd(const EkfConst& c, const double (&x)[EN], const double (&tau)[6], double (&xn)[EN],
                                           double (&xnp)[3]) {
    double k1[12], k2[12], k3[12], k4[12], xs[EN];
#pragma unroll
    for (int i = 12; i < EN; i++) { xs[i] = x[i]; xn[i] = x[i]; }
    ekf_f12(c, x, tau, k1);
#pragma unroll
    for (int i = 0; i < 12; i++) { k1[i] *= c.dt; xs[i] = x[i] + k1[i] * 0.5; }
    ekf_f12(c, xs, tau, k2);
#pragma unroll
    for (int i = 0; i < 12; i++) { k2[i] *= c.dt; xs[i] = x[i] + k2[i] * (1.0 / 3.0); }
    ekf_f12(c, xs, tau, k3);
#pragma unroll
    for (int i = 0; i < 12; i++) { k3[i] *= c.dt; xs[i] = x[i] + k3[i]; }
    ekf_f12(c, xs, tau, k4);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        k4[i] *= c.dt;
        xn[i] = x[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) * (1.0 / 6.0);
        if (i < 3) xnp[i] = (x[i] + c.d) + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) * (1.0 / 6.0);
    }
}

// C += A B over the non-zero pattern NZ of B (NZ::at(row of B, column)); otherwise gemm_block / gemm_dpp
template <class NZ, int K0, int NK, bool SECB, bool CORNER>
__device__ __forceinline__ void gemm_block_sp(Rows& C, const Rows& B, const double (&a0)[NK], const double (&a1)[NK]) {
    for_k(std::make_integer_sequence<int, NK>{}, [&](auto kc) {
        constexpr int KK = decltype(kc)::value;
        constexpr int K = K0 + KK;                 // lane that owns the B row
        constexpr int ROW = SECB ? 16 + K : K;     // the row's index in B
        for_k(std::make_integer_sequence<int, EN>{}, [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if constexpr (NZ::at(ROW, J)) {
                fmac_bc<K>(C.p[J], SECB ? B.s[J] : B.p[J], a0[KK]);
                if constexpr (!CORNER || J >= 16) fmac_bc<K>(C.s[J], SECB ? B.s[J] : B.p[J], a1[KK]);
            }
        });
    });
}
template <class NZ, bool CORNER = false, class AP, class AS>
__device__ __forceinline__ void gemm_dpp_sp(Rows& C, const Rows& B, AP ap, AS as) {
    double p0[6], s0[6], p1[6], s1[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { p0[k] = ap(k); s0[k] = as(k); }
#pragma unroll
    for (int k = 0; k < 6; k++) { p1[k] = ap(6 + k); s1[k] = as(6 + k); }
    pin_rows(B);
    gemm_block_sp<NZ, 0, 6, false, CORNER>(C, B, p0, s0);
#pragma unroll
    for (int k = 0; k < 4; k++) { p0[k] = ap(12 + k); s0[k] = as(12 + k); }
    p0[4] = ap(16); s0[4] = as(16); p0[5] = ap(17); s0[5] = as(17);
    gemm_block_sp<NZ, 6, 6, false, CORNER>(C, B, p1, s1);
    {
        const double q0[4] = {p0[0], p0[1], p0[2], p0[3]}, q1[4] = {s0[0], s0[1], s0[2], s0[3]};
        gemm_block_sp<NZ, 12, 4, false, CORNER>(C, B, q0, q1);
        const double r0[2] = {p0[4], p0[5]}, r1[2] = {s0[4], s0[5]};
        gemm_block_sp<NZ, 0, 2, true, CORNER>(C, B, r0, r1);
    }
}

__global__ __launch_bounds__(64, 2) void ekf_update_kernel_sp(EkfArgs A) {
    extern __shared__ __attribute__((aligned(16))) double esm[];
    const EkfConst& c = A.c;
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    const int inst0 = blockIdx.x * kDppFilters + g;
    const bool live = inst0 < A.B;
    const int inst = live ? inst0 : A.B - 1;
    const bool sec = l < 2;              // owns a secondary row.  The secondary register set of the other lanes holds don't-care values in this
    const int ls = sec ? l : 0;          // kernel (a copy of row 16's arithmetic): nothing reads it, and zeroing it costs two selects per element
    // finite-difference roles: lane 0 evaluates the unperturbed map, lanes 1 / 2 perturb states 16 / 17, lanes 3..15 their own state
    const int pert = l >= 3 ? l : (l == 0 ? -1 : 15 + l);
    const double idp = l >= 3 ? c.inv_d : 0.0;   // 1 / d where this lane's evaluation is a column of its primary row set
    const double ids = sec ? c.inv_d : 0.0;      // ... where the evaluation of lane l + 1 is this lane's secondary row

    elds* sm = (elds*)esm + g * kSpLds;
    elds* bufX = sm;                     // P -> P_pred -> J
    elds* bufY = bufX + kMat;            // F^T -> H^T -> K^T
    elds* v_xn = bufY + kMat;            // [18] corrected state (gathered for the output lane)
    elds* v_xp = v_xn + EN;              // [18] predicted state
    elds* v_ye = v_xp + EN;              // [18] innovation
    elds* v_ym = v_ye + EN;              // [18] measurement

    // ---- inputs: every lane keeps x, tau, acc of its filter
    double x[EN], tau[6], ac[6];
    {
        const double* xg = A.x + (size_t)inst * EN;
        const double* tg = A.thrust + (size_t)inst * 6;
        const double* ag = A.acc + (size_t)inst * 6;
        const double* yg = A.y12 + (size_t)inst * 12;
#pragma unroll
        for (int j = 0; j < EN; j++) x[j] = xg[j];
        double th[6], ym[EN];
#pragma unroll
        for (int j = 0; j < 6; j++) { th[j] = tg[j]; ac[j] = ag[j]; }
#pragma unroll
        for (int i = 0; i < 6; i++) {   // tau = K * meas_u (bluerov2_dob.cpp:499-500)
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) t += c.K[i * 6 + j] * th[j];
            tau[i] = t;
        }
#pragma unroll
        for (int j = 0; j < 12; j++) ym[j] = yg[j];
#pragma unroll
        for (int j = 0; j < 6; j++) ym[12 + j] = tau[j];
        store_row(v_ym, ym);   // identical values from the 16 lanes of the filter
        // P -> bufX (left operand of the first product): the filter's 2592 contiguous bytes in 16-byte pieces, 16 lanes wide
        const ed2* pg = (const ed2*)(A.P + (size_t)inst * kMat);
        ed2 pv[11];
#pragma unroll
        for (int m = 0; m < 11; m++) { const int q = m * 16 + l; pv[m] = pg[q < kMat / 2 ? q : 0]; }
#pragma unroll
        for (int m = 0; m < 11; m++) { const int q = m * 16 + l; if (q < kMat / 2) ((elds2*)bufX)[q] = pv[m]; }
    }

    // ---- F^T by forward differences of the RK4 map, one evaluation per lane
    Rows Ft;
    {
        double xl[EN], xb[EN], xnp[3];
#pragma unroll
        for (int j = 0; j < EN; j++) xl[j] = x[j] + ((j == pert) ? c.d : 0.0);
        ekf_rk4_fd(c, xl, tau, xb, xnp);
        double f0[EN];
#pragma unroll
        for (int j = 0; j < EN; j++) f0[j] = bcast<0>(xb[j]);
        const double q0 = bcast<0>(xnp[0]), q1 = bcast<0>(xnp[1]), q2 = bcast<0>(xnp[2]);
        const double fd[3] = {(q0 - f0[0]) * c.inv_d, (q1 - f0[1]) * c.inv_d, (q2 - f0[2]) * c.inv_d};
#pragma unroll
        for (int j = 0; j < EN; j++) {
            const double v = (xb[j] - f0[j]) * idp;
            Ft.p[j] = (j < 3 && l == j) ? fd[j < 3 ? j : 0] : v;
            Ft.s[j] = (dpp32_f64<0x101>(xb[j]) - f0[j]) * ids;   // row_shl:1 -- lane 0 <- lane 1 (state 16), lane 1 <- lane 2 (state 17)
            x[j] = f0[j];                   // from here on x = x_pred
        }
        store_row(v_xp, f0);
    }
    __syncthreads();
    // G = P F^T
    Rows G;
    zero_rows(G);
    gemm_dpp_sp<NzFt>(G, Ft, [&](int k) { return bufX[l * EN + k]; }, [&](int k) { return bufX[(16 + ls) * EN + k]; });
    store_rows(bufY, Ft, l);                // F^T row-major: F[l][k] = bufY[k][l]
    __syncthreads();                        // (also: the reads of P are done, bufX is free)
    // P_pred = F G + Q.  F[l][k] is zero for k < 3 except on the diagonal (own-lane multiply-add), rows 16 / 17 of F are e_16 / e_17
    Rows Pq;
#pragma unroll
    for (int j = 0; j < EN; j++) { Pq.p[j] = (j == l) ? c.Q[j] : 0.0; Pq.s[j] = (sec && j == 16 + l) ? c.Q[j] : 0.0; }
    {
        double a[15];
#pragma unroll
        for (int k = 0; k < 15; k++) a[k] = bufY[(3 + k) * EN + l];
        const double fdl = bufY[l * EN + l], a16 = bufY[16 * EN + 16 + ls], a17 = bufY[17 * EN + 16 + ls];
        const double fd3 = (l < 3) ? fdl : 0.0, s16 = a16, s17 = a17;
#pragma unroll
        for (int j = 0; j < EN; j++) Pq.p[j] = fma(fd3, G.p[j], Pq.p[j]);
        pin_rows(G);
        for_k(std::make_integer_sequence<int, 13>{}, [&](auto kc) {
            constexpr int K = 3 + decltype(kc)::value;
#pragma unroll
            for (int j = 0; j < EN; j++) fmac_bc<K>(Pq.p[j], G.p[j], a[K - 3]);
        });
#pragma unroll
        for (int j = 0; j < EN; j++) fmac_bc<0>(Pq.p[j], G.s[j], a[13]);
        fmac_bc<0>(Pq.s[16], G.s[16], s16); fmac_bc<0>(Pq.s[17], G.s[17], s16);
#pragma unroll
        for (int j = 0; j < EN; j++) fmac_bc<1>(Pq.p[j], G.s[j], a[14]);
        fmac_bc<1>(Pq.s[16], G.s[16], s17); fmac_bc<1>(Pq.s[17], G.s[17], s17);
    }
    fill_secondary_from_symmetry<false>(Pq, l);    // P_pred is symmetric: only the 2x2 corner of rows 16, 17 was accumulated
    store_rows(bufX, Pq, l);                // P_pred stays in bufX (right operand of V = J P_pred)
    // ---- H^T by forward differences of h at x_pred (one evaluation per lane), innovation
    Rows Ht;
    {
        double xl[EN], yb[EN], ye[EN], ym2[EN];
        load_row(v_ym, ym2);
#pragma unroll
        for (int j = 0; j < EN; j++) xl[j] = x[j] + ((j == pert) ? c.d : 0.0);
        ekf_h(c, xl, ac, yb);
#pragma unroll
        for (int j = 0; j < EN; j++) {
            const double y0 = bcast<0>(yb[j]);
            const double v = (yb[j] - y0) * idp;
            // columns 0..2: h passes the position through, (x_j + d) - x_j
            Ht.p[j] = (j < 3 && l == j) ? ((x[j] + c.d) - y0) * c.inv_d : v;
            Ht.s[j] = (dpp32_f64<0x101>(yb[j]) - y0) * ids;
            ye[j] = ym2[j] - y0;
        }
        store_row(v_ye, ye);
    }
    __syncthreads();                        // reads of bufY (F^T) are done
    store_rows(bufY, Ht, l);                // H^T row-major: H[l][k] = bufY[k][l]
    __syncthreads();
    // W = H P_pred.  Rows 0..11 of H are diagonal; rows 12..17 have the eight columns of ekf_h21col and their own
    Rows W;
    zero_rows(W);
    {
        const bool hi = l >= 12;
        double a[8], h16[8], h17[8];
        for_k(std::make_integer_sequence<int, 12>{}, [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (ekf_h21col(K)) {
                constexpr int Q = ekf_h21idx(K);
                const double t = bufY[K * EN + l];
                a[Q] = hi ? t : 0.0; h16[Q] = bufY[K * EN + 16]; h17[Q] = bufY[K * EN + 17];
            }
        });
        const double hd = bufY[l * EN + l], hd16 = bufY[16 * EN + 16], hd17 = bufY[17 * EN + 17];
        pin_rows(Pq);
        for_k(std::make_integer_sequence<int, 12>{}, [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (ekf_h21col(K)) {
#pragma unroll
                for (int j = 0; j < EN; j++) fmac_bc<K>(W.p[j], Pq.p[j], a[ekf_h21idx(K)]);
            }
        });
#pragma unroll
        for (int j = 0; j < EN; j++) W.p[j] = fma(hd, Pq.p[j], W.p[j]);
        // rows 16, 17 through the symmetry of P_pred, as gemm_dpp_symB does it: lane j computes W[16][j], W[17][j] from its own row
        double c16 = 0.0, c17 = 0.0, d16 = 0.0, d17 = 0.0;
        for_k(std::make_integer_sequence<int, 12>{}, [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (ekf_h21col(K)) {
                constexpr int Q = ekf_h21idx(K);
                c16 = fma(h16[Q], Pq.p[K], c16); c17 = fma(h17[Q], Pq.p[K], c17);
                d16 = fma(h16[Q], Pq.s[K], d16); d17 = fma(h17[Q], Pq.s[K], d17);
            }
        });
        c16 = fma(hd16, Pq.p[16], c16); d16 = fma(hd16, Pq.s[16], d16);
        c17 = fma(hd17, Pq.p[17], c17); d17 = fma(hd17, Pq.s[17], d17);
        for_k(std::make_integer_sequence<int, 16>{}, [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            const double t16 = bcast<J>(c16), t17 = bcast<J>(c17);
            W.s[J] = (l == 0) ? t16 : t17;
        });
        const double e16_0 = bcast<0>(d16), e16_1 = bcast<1>(d16), e17_0 = bcast<0>(d17), e17_1 = bcast<1>(d17);
        W.s[16] = (l == 0) ? e16_0 : e17_0;
        W.s[17] = (l == 0) ? e16_1 : e17_1;
    }
    // S = W H^T + R.  The left operand's elements W[l][k], W[16 + l][k] are this lane's own rows: W never goes through LDS
    Rows S;
#pragma unroll
    for (int j = 0; j < EN; j++) { S.p[j] = (j == l) ? c.R : 0.0; S.s[j] = (sec && j == 16 + l) ? c.R : 0.0; }
    gemm_dpp_sp<NzHt, true>(S, Ht, [&](int k) { return W.p[k]; }, [&](int k) { return W.s[k]; });
    fill_secondary_from_symmetry<false>(S, l);     // S is symmetric
    // ---- K^T = S^-1 W by Gauss-Jordan on [S | W] without pivoting (S is SPD); the pivot row is broadcast with DPP.
    Rows& T = W;                            // Gauss-Jordan on [S | W] in place
    bool ok = true;
    pin_rows(S); pin_rows(T);                 // (fill_secondary_from_symmetry / load_rows end in plain selects)
    for_k(std::make_integer_sequence<int, EN>{}, [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        constexpr bool PS = K >= 16;          // pivot row in the secondary set (lanes 0, 1)
        constexpr int KL = PS ? K - 16 : K;   // lane that owns the pivot row
        const double pk = PS ? bcast<KL>(S.s[K]) : bcast<KL>(S.p[K]);
        ok = ok && (pk > 0.0) && (pk < 1e300);
        double ip = __builtin_amdgcn_rcp(pk);
        double e1 = fma(-pk, ip, 1.0);
        ip = fma(ip, e1, ip);
        e1 = fma(-pk, ip, 1.0);
        ip = fma(ip, e1, ip);
        const double nfp = (!PS && l == KL) ? ip - 1.0 : -(S.p[K] * ip);
        const double nfs = (PS && l == KL) ? ip - 1.0 : -(S.s[K] * ip);
#pragma unroll
        for (int j = K + 1; j < EN; j++) {
            if (PS) { fmac_bc<KL>(S.p[j], S.s[j], nfp); fmac_bc<KL>(S.s[j], S.s[j], nfs); }
            else { fmac_bc<KL>(S.s[j], S.p[j], nfs); fmac_bc<KL>(S.p[j], S.p[j], nfp); }
        }
#pragma unroll
        for (int j = 0; j < EN; j++) {
            if (PS) { fmac_bc<KL>(T.p[j], T.s[j], nfp); fmac_bc<KL>(T.s[j], T.s[j], nfs); }
            else { fmac_bc<KL>(T.s[j], T.p[j], nfs); fmac_bc<KL>(T.p[j], T.p[j], nfp); }
        }
    });
    if (!ok) zero_rows(T);                  // innovation covariance not positive definite / NaN: keep the prediction
    // the entries of H that J = I - Kal H needs (right operand in row layout: lane k owns row k -- its diagonal entry and, for rows
    // 12..17, the eight entries of ekf_h21col), out of bufY before K^T takes its place
    double hp[8], hs[8];
    for_k(std::make_integer_sequence<int, 12>{}, [&](auto jc) {
        constexpr int JJ = decltype(jc)::value;
        if constexpr (ekf_h21col(JJ)) { hp[ekf_h21idx(JJ)] = bufY[JJ * EN + l]; hs[ekf_h21idx(JJ)] = bufY[JJ * EN + 16 + ls]; }
    });
    const double hdJ = bufY[l * EN + l], hsdJ = bufY[(16 + ls) * EN + 16 + ls];
    __syncthreads();                        // reads of bufY (H^T) are done
    store_rows(bufY, T, l);                 // K^T row-major: Kal[l][k] = bufY[k][l]
    __syncthreads();
    // x_new = x_pred + Kal (y - y_pred)
    {
        double dp = 0.0, ds = 0.0, ye[EN];
        load_row(v_ye, ye);
#pragma unroll
        for (int k = 0; k < EN; k++) { dp = fma(bufY[k * EN + l], ye[k], dp); ds = fma(bufY[k * EN + 16 + ls], ye[k], ds); }
        const double xp_ = v_xp[l] + dp, xs_ = v_xp[16 + ls] + ds;
        v_xn[l] = xp_;
        if (sec) v_xn[16 + l] = xs_;
        if (live) {
            A.x[(size_t)inst * EN + l] = xp_;
            if (sec) A.x[(size_t)inst * EN + 16 + l] = xs_;
        }
    }
    // J = I - Kal H
    Rows J;
    zero_rows(J);
    {
        const double hd = hdJ, hsd = hsdJ;
        double a0[EN], a1[EN];
#pragma unroll
        for (int k = 0; k < EN; k++) { a0[k] = -bufY[k * EN + l]; a1[k] = -bufY[k * EN + 16 + ls]; }
        asm volatile("s_nop 1");
        for_k(std::make_integer_sequence<int, 12>{}, [&](auto kc) {
            constexpr int K = decltype(kc)::value;
            fmac_bc<K>(J.p[K], hd, a0[K]); fmac_bc<K>(J.s[K], hd, a1[K]);
        });
        for_k(std::make_integer_sequence<int, 4>{}, [&](auto kc) {
            constexpr int K = 12 + decltype(kc)::value;
            for_k(std::make_integer_sequence<int, 12>{}, [&](auto jc) {
                constexpr int JJ = decltype(jc)::value;
                if constexpr (ekf_h21col(JJ)) { fmac_bc<K>(J.p[JJ], hp[ekf_h21idx(JJ)], a0[K]); fmac_bc<K>(J.s[JJ], hp[ekf_h21idx(JJ)], a1[K]); }
            });
            fmac_bc<K>(J.p[K], hd, a0[K]); fmac_bc<K>(J.s[K], hd, a1[K]);
        });
        for_k(std::make_integer_sequence<int, 2>{}, [&](auto kc) {
            constexpr int KL = decltype(kc)::value;
            constexpr int K = 16 + KL;
            for_k(std::make_integer_sequence<int, 12>{}, [&](auto jc) {
                constexpr int JJ = decltype(jc)::value;
                if constexpr (ekf_h21col(JJ)) { fmac_bc<KL>(J.p[JJ], hs[ekf_h21idx(JJ)], a0[K]); fmac_bc<KL>(J.s[JJ], hs[ekf_h21idx(JJ)], a1[K]); }
            });
            fmac_bc<KL>(J.p[K], hsd, a0[K]); fmac_bc<KL>(J.s[K], hsd, a1[K]);
        });
    }
#pragma unroll
    for (int j = 0; j < EN; j++) {
        if (!ok) { J.p[j] = 0.0; J.s[j] = 0.0; }   // H may hold NaN: J = I exactly
        J.p[j] += (j == l) ? 1.0 : 0.0;
        J.s[j] += (sec && j == 16 + l) ? 1.0 : 0.0;
    }
    // V = J P_pred: P_pred out of bufX into registers (right operand), J takes its place (left operand, and J^T below)
    Rows V;
    {
        Rows Pr;
        load_rows<false>(bufX, Pr, l);
        __syncthreads();                    // reads of bufX (P_pred) are done
        store_rows(bufX, J, l);             // J row-major
        __syncthreads();
        zero_rows(V);
        gemm_dpp_symB<false>(V, Pr, [&](int k) { return bufX[l * EN + k]; }, [&](int k) { return bufX[16 * EN + k]; }, [&](int k) { return bufX[17 * EN + k]; }, l);
    }
    // P_new = V J^T + R Kal Kal^T   (Joseph form, bluerov2_dob.cpp:537).  Left operand of the first product: V's own rows (registers)
    Rows Pn;
    zero_rows(Pn);
    {
        Rows Jt;
        load_rows_t<false>(bufX, Jt, l);
        gemm_dpp<true>(Pn, Jt, [&](int k) { return V.p[k]; }, [&](int k) { return V.s[k]; });
        Rows Kt;
        load_rows<false>(bufY, Kt, l);
        gemm_dpp<true>(Pn, Kt, [&](int k) { return c.R * bufY[k * EN + l]; }, [&](int k) { return c.R * bufY[k * EN + 16 + ls]; });
    }
    if (live) {   // P_new is symmetric: rows 16, 17 are columns 16, 17 of the primary rows (+ the 2x2 corner from lanes 0, 1)
        double* pg = A.P + (size_t)inst * kMat;
#pragma unroll
        for (int j = 0; j < EN; j++) pg[l * EN + j] = Pn.p[j];
        pg[16 * EN + l] = Pn.p[16];
        pg[17 * EN + l] = Pn.p[17];
        if (sec) { pg[(16 + l) * EN + 16] = Pn.s[16]; pg[(16 + l) * EN + 17] = Pn.s[17]; }
    }
    // ---- outputs: world-frame disturbance with the MEASURED attitude (:540-545), NMPC parameters (:334-337)
    __syncthreads();
    if (l == 0 && live) {
        double xn[EN];
        load_row(v_xn, xn);
        double sph, cph, sth, cth, sps, cps;
        sincos_pio2(v_ym[3], &sph, &cph);
        sincos_pio2(v_ym[4], &sth, &cth);
        sincos_pio2(v_ym[5], &sps, &cps);
        double* w = A.wf + (size_t)inst * 6;
        w[0] = (cps * cth) * xn[12] + (-sps * cph + cps * sth * sph) * xn[13] + (sps * sph + cps * cph * sth) * xn[14];
        w[1] = (sps * cth) * xn[12] + (cps * cph + sph * sth * sps) * xn[13] + (-cps * sph + sth * sps * cph) * xn[14];
        w[2] = (-sth) * xn[12] + (cth * sph) * xn[13] + (cth * cph) * xn[14];
        w[3] = xn[15] + (sps * sth / cth) * xn[16] + cph * sth / cth * xn[17];
        w[4] = cph * xn[16] + sph * xn[17];
        w[5] = (sph / cth) * xn[16] + (cph / cth) * xn[17];
        double* mp = A.mp + (size_t)inst * 4;
        mp[0] = xn[12] * c.inv_cc;
        mp[1] = xn[13] * c.inv_cc;
        mp[2] = xn[14] * c.inv_rc;
        mp[3] = xn[17] * c.inv_rc;
        bool fin = true;
#pragma unroll
        for (int j = 0; j < EN; j++) fin = fin && (fabs(xn[j]) < 1e300);
        A.status[inst] = !ok ? 1 : (fin ? 0 : 2);
    }
}

// measurement assembly for the on-device DOB-MPC loop: y12 = plant state, thrust = allocation of u0 (bluerov2_dob.cpp:390-395)
// with the OCP model's rotor constant, i.e. exactly the thruster vector brov_plant_step applies; acc = (v - v_prev) / dt
// (:148-153); one lane per filter
__global__ void ekf_inputs_from_solver_kernel(int B, double dt, double inv_rc, const double* __restrict__ x0,
                                              const brov_result* __restrict__ res, double* __restrict__ vprev,
                                              double* __restrict__ thrust, double* __restrict__ y12, double* __restrict__ acc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* xs = x0 + (size_t)b * 12;
#pragma unroll
    for (int j = 0; j < 12; j++) y12[(size_t)b * 12 + j] = xs[j];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double v = xs[6 + j];
        acc[(size_t)b * 6 + j] = (v - vprev[(size_t)b * 6 + j]) / dt;
        vprev[(size_t)b * 6 + j] = v;
    }
    const double u0 = res[b].u0[0], u1 = res[b].u0[1], u2 = res[b].u0[2], u3 = res[b].u0[3];
    double* t = thrust + (size_t)b * 6;
    t[0] = (-u0 + u1 + u3) * inv_rc;
    t[1] = (-u0 - u1 - u3) * inv_rc;
    t[2] = (u0 + u1 - u3) * inv_rc;
    t[3] = (u0 - u1 + u3) * inv_rc;
    t[4] = (-u2) * inv_rc;
    t[5] = (-u2) * inv_rc;
}

// p[0..3] of every stage of instance b := estimate of instance b (bluerov2_dob.cpp:332-337)
// 6-disturbance variant (rp != nullptr): the roll / pitch estimates esti_x(15), esti_x(16) too, scaled like the z / yaw channels
__global__ void ekf_apply_kernel(int B, int stages, const double* __restrict__ mp, double* __restrict__ par, const double* __restrict__ xest,
                                 double inv_rc, double* __restrict__ rp) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B * stages) return;
    const int b = k / stages;
    double* p = par + (size_t)k * 16;
#pragma unroll
    for (int j = 0; j < 4; j++) p[j] = mp[(size_t)b * 4 + j];
    if (rp) { rp[(size_t)k * 2] = xest[(size_t)b * 18 + 15] * inv_rc; rp[(size_t)k * 2 + 1] = xest[(size_t)b * 18 + 16] * inv_rc; }
}

}  // namespace brov

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
using namespace brov;

static thread_local std::string g_ekf_err;
#define EKFCHK(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            g_ekf_err = std::string(#call) + ": " + hipGetErrorString(e_);                                    \
            (void)hipGetLastError();                                                                              \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorInsufficientDriver) \
                       ? BROV_ERR_NO_DEVICE                                                                   \
                       : BROV_ERR_HIP;                                                                        \
        }                                                                                                     \
    } while (0)

struct brov_ekf {
    int device = 0, B = 0;
    brov_ekf_params par{};
    EkfConst c{};
    double *x = nullptr, *P = nullptr, *thrust = nullptr, *y12 = nullptr, *acc = nullptr, *wf = nullptr, *mp = nullptr,
           *vprev = nullptr;
    int* status = nullptr;
    hipStream_t last_stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ev_valid = false;
    int variant = 2;   // 2: structured DPP kernel (default), 1: dense DPP row-broadcast kernel, 0: LDS-broadcast kernel (BROV_EKF_VARIANT, kept for A/B runs)
    std::vector<void*> allocs;
};

extern "C" const char* brov_ekf_last_error(void) { return g_ekf_err.c_str(); }

extern "C" void brov_ekf_default_params(brov_ekf_params* p) {
    // bluerov2_dob.h:171-183,208; bluerov2_dob.cpp:52-62
    static const double am[6] = {1.7182, 0, 5.468, 0, 1.2481, 0.4006};
    static const double dl[6] = {-11.7391, -20, -31.8678, -25, -44.9085, -5};
    static const double dnl[6] = {-18.18, -21.66, -36.99, -1.55, -1.55, -1.55};
    static const double K[36] = {
        0.7071067811847433,   0.7071067811847433,    -0.7071067811919605, -0.7071067811919605,  0.0,                   0.0,
        0.7071067811883519,   -0.7071067811883519,   0.7071067811811348,  -0.7071067811811348,  0.0,                   0.0,
        0,                    0,                     0,                   0,                    1,                     1,
        0.051265241636155506, -0.05126524163615552,  0.05126524163563227, -0.05126524163563227, -0.11050000000000001,  0.11050000000000003,
        -0.05126524163589389, -0.051265241635893896, 0.05126524163641713, 0.05126524163641713,  -0.002499999999974481, -0.002499999999974481,
        0.16652364696949604,  -0.16652364696949604,  -0.17500892834341342, 0.17500892834341342, 0.0,                   0.0};
    std::memset(p, 0, sizeof(*p));
    p->dt = 0.05;
    p->mass = 11.26; p->Ix = 0.3; p->Iy = 0.63; p->Iz = 0.58; p->ZG = 0.02; p->g = 9.81; p->bouyancy = 0.661618;
    std::memcpy(p->added_mass, am, sizeof am);
    std::memcpy(p->Dl, dl, sizeof dl);
    std::memcpy(p->Dnl, dnl, sizeof dnl);
    std::memcpy(p->K, K, sizeof K);
    for (int i = 0; i < 6; i++) p->Q[i] = std::pow(p->dt, 4) / 4;
    for (int i = 6; i < 18; i++) p->Q[i] = std::pow(p->dt, 2);
    p->R = std::pow(p->dt, 4) / 4;
    p->fd_step = 1e-6;
    p->compensate_coef = 0.032546960744430276;
    p->rotor_constant = 0.026546960744430276;
}

// diagonal of M and of M^-1 (bluerov2_dob.cpp:41-47); 6x6 Gauss-Jordan with partial pivoting on the host, once
static void derive_mass(const brov_ekf_params& p, double* Md, double* iMd) {
    double a[6][12];
    std::memset(a, 0, sizeof a);
    const double mv[6] = {p.mass + p.added_mass[0], p.mass + p.added_mass[1], p.mass + p.added_mass[2],
                          p.Ix + p.added_mass[3],   p.Iy + p.added_mass[4],   p.Iz + p.added_mass[5]};
    for (int i = 0; i < 6; i++) { a[i][i] = mv[i]; a[i][6 + i] = 1.0; Md[i] = mv[i]; }
    a[0][4] = p.mass * p.ZG; a[1][3] = -p.mass * p.ZG; a[3][1] = -p.mass * p.ZG; a[4][0] = p.mass * p.ZG;
    for (int k = 0; k < 6; k++) {
        int q = k;
        for (int i = k + 1; i < 6; i++)
            if (std::fabs(a[i][k]) > std::fabs(a[q][k])) q = i;
        if (q != k)
            for (int j = 0; j < 12; j++) std::swap(a[k][j], a[q][j]);
        const double ip = 1.0 / a[k][k];
        for (int j = 0; j < 12; j++) a[k][j] *= ip;
        for (int i = 0; i < 6; i++) {
            if (i == k) continue;
            const double f = a[i][k];
            for (int j = 0; j < 12; j++) a[i][j] -= f * a[k][j];
        }
    }
    for (int i = 0; i < 6; i++) iMd[i] = a[i][6 + i];
}

static void make_const(const brov_ekf_params& p, EkfConst& c) {
    c.dt = p.dt; c.mass = p.mass; c.bo = p.bouyancy; c.mzg = p.mass * p.ZG * p.g;
    c.iy_iz = p.Iy - p.Iz; c.iz_ix = p.Iz - p.Ix; c.iy_ix = p.Iy - p.Ix;
    c.R = p.R; c.d = p.fd_step; c.inv_d = 1.0 / p.fd_step; c.inv_cc = 1.0 / p.compensate_coef; c.inv_rc = 1.0 / p.rotor_constant;
    for (int i = 0; i < 6; i++) { c.Dl[i] = p.Dl[i]; c.Dnl[i] = p.Dnl[i]; }
    for (int i = 0; i < 36; i++) c.K[i] = p.K[i];
    for (int i = 0; i < 18; i++) c.Q[i] = p.Q[i];
    derive_mass(p, c.Md, c.iMd);
}

template <typename T>
static int ekf_alloc(brov_ekf* e, T** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(T)) != hipSuccess) { g_ekf_err = "hipMalloc failed"; return BROV_ERR_ALLOC; }
    e->allocs.push_back(q);
    *p = (T*)q;
    return BROV_OK;
}

extern "C" void brov_ekf_destroy(brov_ekf* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (void* q : e->allocs) (void)hipFree(q);
    for (auto& ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    delete e;
}

extern "C" int brov_ekf_reset(brov_ekf* e, const double* x0, const double* P0) {
    if (!e) return BROV_ERR_ARG;
    EKFCHK(hipSetDevice(e->device));
    static const double xr[18] = {0, 0, -20, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 6, 6, 0, 0, 0};  // bluerov2_dob.cpp:64
    std::vector<double> hx((size_t)e->B * 18), hp((size_t)e->B * 324, 0.0);
    for (int b = 0; b < e->B; b++) {
        std::memcpy(&hx[(size_t)b * 18], x0 ? x0 : xr, sizeof xr);
        if (P0) std::memcpy(&hp[(size_t)b * 324], P0, 324 * sizeof(double));
        else for (int i = 0; i < 18; i++) hp[(size_t)b * 324 + i * 19] = 1.0;
    }
    EKFCHK(hipMemcpy(e->x, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice));
    EKFCHK(hipMemcpy(e->P, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice));
    EKFCHK(hipMemset(e->vprev, 0, (size_t)e->B * 6 * sizeof(double)));
    return BROV_OK;
}

extern "C" int brov_ekf_create(brov_ekf** out, int device, int B, const brov_ekf_params* p) {
    if (!out || B <= 0) { g_ekf_err = "brov_ekf_create: bad arguments"; return BROV_ERR_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_ekf_err = "brov_ekf_create: no usable HIP device (the observer has no CPU path)";
        return BROV_ERR_NO_DEVICE;
    }
    EKFCHK(hipSetDevice(device));
    brov_ekf* e = new brov_ekf();
    e->device = device; e->B = B;
    if (const char* v = std::getenv("BROV_EKF_VARIANT")) { const int k = std::atoi(v); e->variant = (k >= 0 && k <= 2) ? k : 2; }
    if (p) e->par = *p; else brov_ekf_default_params(&e->par);
    make_const(e->par, e->c);
    int rc = BROV_OK;
    if ((rc = ekf_alloc(e, &e->x, (size_t)B * 18)) || (rc = ekf_alloc(e, &e->P, (size_t)B * 324)) ||
        (rc = ekf_alloc(e, &e->thrust, (size_t)B * 6)) || (rc = ekf_alloc(e, &e->y12, (size_t)B * 12)) ||
        (rc = ekf_alloc(e, &e->acc, (size_t)B * 6)) || (rc = ekf_alloc(e, &e->wf, (size_t)B * 6)) ||
        (rc = ekf_alloc(e, &e->mp, (size_t)B * 4)) || (rc = ekf_alloc(e, &e->vprev, (size_t)B * 6)) ||
        (rc = ekf_alloc(e, &e->status, (size_t)B))) {
        brov_ekf_destroy(e);
        return rc;
    }
    if (hipMemset(e->wf, 0, (size_t)B * 6 * sizeof(double)) != hipSuccess || hipMemset(e->mp, 0, (size_t)B * 4 * sizeof(double)) != hipSuccess ||
        hipMemset(e->status, 0, (size_t)B * sizeof(int)) != hipSuccess || hipEventCreate(&e->ev[0]) != hipSuccess ||
        hipEventCreate(&e->ev[1]) != hipSuccess) {
        g_ekf_err = "brov_ekf_create: device initialisation failed";
        brov_ekf_destroy(e);
        return BROV_ERR_HIP;
    }
    rc = brov_ekf_reset(e, nullptr, nullptr);
    if (rc) { brov_ekf_destroy(e); return rc; }
    // (no hipFuncSetAttribute: the three kernels ask for 35 / 33 / 23 KB of dynamic LDS per block, below the 64 KB a launch may have without it)
    static_assert(kPerWave * kLdsPerFilter * sizeof(double) <= 64 * 1024 && kDppFilters * kDppLds * sizeof(double) <= 64 * 1024, "dynamic LDS");
    *out = e;
    return BROV_OK;
}

extern "C" int brov_ekf_batch(const brov_ekf* e) { return e ? e->B : 0; }

extern "C" int brov_ekf_set_state_host(brov_ekf* e, const double* x, const double* P) {
    if (!e) return BROV_ERR_ARG;
    EKFCHK(hipSetDevice(e->device));
    if (x) EKFCHK(hipMemcpy(e->x, x, (size_t)e->B * 18 * sizeof(double), hipMemcpyHostToDevice));
    if (P) EKFCHK(hipMemcpy(e->P, P, (size_t)e->B * 324 * sizeof(double), hipMemcpyHostToDevice));
    return BROV_OK;
}

extern "C" int brov_ekf_get_state_host(brov_ekf* e, double* x, double* P) {
    if (!e) return BROV_ERR_ARG;
    EKFCHK(hipSetDevice(e->device));
    EKFCHK(hipStreamSynchronize(e->last_stream));
    if (x) EKFCHK(hipMemcpy(x, e->x, (size_t)e->B * 18 * sizeof(double), hipMemcpyDeviceToHost));
    if (P) EKFCHK(hipMemcpy(P, e->P, (size_t)e->B * 324 * sizeof(double), hipMemcpyDeviceToHost));
    return BROV_OK;
}

static int launch_update(brov_ekf* e, const double* thrust, const double* y12, const double* acc, hipStream_t st) {
    EkfArgs a;
    a.c = e->c; a.B = e->B; a.x = e->x; a.P = e->P; a.thrust = thrust; a.y12 = y12; a.acc = acc; a.wf = e->wf; a.mp = e->mp;
    a.status = e->status;
    EKFCHK(hipEventRecord(e->ev[0], st));
    if (e->variant == 0) {
        const int blocks = (e->B + kPerWave - 1) / kPerWave;
        hipLaunchKernelGGL(ekf_update_kernel, dim3(blocks), dim3(64), kPerWave * kLdsPerFilter * sizeof(double), st, a);
    } else {
        const int blocks = (e->B + kDppFilters - 1) / kDppFilters;
        if (e->variant == 1) hipLaunchKernelGGL(ekf_update_kernel_dpp, dim3(blocks), dim3(64), kDppFilters * kDppLds * sizeof(double), st, a);
        else hipLaunchKernelGGL(ekf_update_kernel_sp, dim3(blocks), dim3(64), kDppFilters * kSpLds * sizeof(double), st, a);
    }
    EKFCHK(hipGetLastError());
    EKFCHK(hipEventRecord(e->ev[1], st));
    e->ev_valid = true;
    e->last_stream = st;
    return BROV_OK;
}

extern "C" int brov_ekf_update_device(brov_ekf* e, const double* thrust, const double* y12, const double* acc, void* stream) {
    if (!e || !thrust || !y12 || !acc) return BROV_ERR_ARG;
    EKFCHK(hipSetDevice(e->device));
    return launch_update(e, thrust, y12, acc, (hipStream_t)stream);
}

extern "C" int brov_ekf_update_host(brov_ekf* e, const double* thrust, const double* y12, const double* acc, void* stream) {
    if (!e || !thrust || !y12 || !acc) return BROV_ERR_ARG;
    EKFCHK(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    EKFCHK(hipMemcpyAsync(e->thrust, thrust, (size_t)e->B * 6 * sizeof(double), hipMemcpyHostToDevice, st));
    EKFCHK(hipMemcpyAsync(e->y12, y12, (size_t)e->B * 12 * sizeof(double), hipMemcpyHostToDevice, st));
    EKFCHK(hipMemcpyAsync(e->acc, acc, (size_t)e->B * 6 * sizeof(double), hipMemcpyHostToDevice, st));
    return launch_update(e, e->thrust, e->y12, e->acc, st);
}

extern "C" int brov_ekf_get_outputs_host(brov_ekf* e, double* wf, double* mpc_p, int* status) {
    if (!e) return BROV_ERR_ARG;
    EKFCHK(hipSetDevice(e->device));
    EKFCHK(hipStreamSynchronize(e->last_stream));
    if (wf) EKFCHK(hipMemcpy(wf, e->wf, (size_t)e->B * 6 * sizeof(double), hipMemcpyDeviceToHost));
    if (mpc_p) EKFCHK(hipMemcpy(mpc_p, e->mp, (size_t)e->B * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (status) EKFCHK(hipMemcpy(status, e->status, (size_t)e->B * sizeof(int), hipMemcpyDeviceToHost));
    return BROV_OK;
}

extern "C" const double* brov_ekf_x_device(const brov_ekf* e) { return e ? e->x : nullptr; }
extern "C" const double* brov_ekf_P_device(const brov_ekf* e) { return e ? e->P : nullptr; }
extern "C" const double* brov_ekf_mpc_p_device(const brov_ekf* e) { return e ? e->mp : nullptr; }

extern "C" int brov_ekf_update_from_solver(brov_ekf* e, brov_solver* s, void* stream) {
    if (!e || !s || brov_batch(s) != e->B) { g_ekf_err = "brov_ekf_update_from_solver: batch sizes differ"; return BROV_ERR_ARG; }
    EKFCHK(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    if (brov_order_stream(s, stream) != BROV_OK) { g_ekf_err = "brov_ekf_update_from_solver: could not order behind the solver's last stream"; return BROV_ERR_HIP; }
    hipLaunchKernelGGL(ekf_inputs_from_solver_kernel, dim3((e->B + 255) / 256), dim3(256), 0, st, e->B, e->c.dt, 1.0 / kRotor,
                       (const double*)brov_x0_device(s), brov_results_device(s), e->vprev, e->thrust, e->y12, e->acc);
    EKFCHK(hipGetLastError());
    return launch_update(e, e->thrust, e->y12, e->acc, st);
}

extern "C" int brov_ekf_apply_to_solver(brov_ekf* e, brov_solver* s, void* stream) {
    if (!e || !s || brov_batch(s) != e->B) { g_ekf_err = "brov_ekf_apply_to_solver: batch sizes differ"; return BROV_ERR_ARG; }
    EKFCHK(hipSetDevice(e->device));
    brov_opts o;
    if (brov_get_opts(s, &o) != BROV_OK) return BROV_ERR_ARG;
    const int stages = o.N + 1;
    const long long n = (long long)e->B * stages;
    if (brov_order_stream(s, stream) != BROV_OK) { g_ekf_err = "brov_ekf_apply_to_solver: could not order behind the solver's last stream"; return BROV_ERR_HIP; }
    hipLaunchKernelGGL(ekf_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->B, stages,
                       (const double*)e->mp, brov_params_device(s), (const double*)e->x, e->c.inv_rc, brov_rp_disturbance_device(s));
    EKFCHK(hipGetLastError());
    e->last_stream = (hipStream_t)stream;
    return BROV_OK;
}

extern "C" int brov_ekf_last_update_seconds(brov_ekf* e, double* seconds) {
    if (!e || !seconds || !e->ev_valid) return BROV_ERR_ARG;
    EKFCHK(hipSetDevice(e->device));
    EKFCHK(hipEventSynchronize(e->ev[1]));
    float ms = 0.f;
    EKFCHK(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    *seconds = ms * 1e-3;
    return BROV_OK;
}

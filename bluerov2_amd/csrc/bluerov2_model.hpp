// bluerov2_model.hpp -- device-side BlueROV2 model for gfx950 (FP64).
//
// Hand-derived f(x,u,p) and the action of its Jacobian on a direction (A_c s), exploiting the 48/144 sparsity of
// df/dx, the constancy of df/du and the rotation-matrix structure of the kinematic rows.  Follows the OCP definition
// /root/reference/bluerov2_dobmpc/scripts/bluerov2.py:77-137 (incl. the sin(psi) term of dphi, :133) and the
// derivative conventions of the generated c_generated_code/bluerov2_model/bluerov2_expl_vde_forw.c
// (d|v|v/dv = sign(v) v + |v| = 2|v|, zero at v = 0, :65).  Not a translation of the CasADi code: no common file
// structure, ~130 flops per Jacobian-vector product instead of a 4.5k-statement dense VDE.
#pragma once
#include <hip/hip_runtime.h>

namespace brov {

constexpr int NX = 12, NU = 4, NP = 16, NY = 16;

// bluerov2.py:77-84
constexpr double kMass = 11.26, kIx = 0.3, kIy = 0.63, kIz = 0.58, kZG = 0.02, kG = 9.81, kBouy = 0.66;
constexpr double kRotor = 0.026546960744430276;
constexpr double kMzg = kMass * kZG * kG;

// parameter-derived constants of one stage (p = [dist4 | added mass4 | linear damping4 | quadratic damping4])
struct ModelPar {
    double dx, dy, dz, dn;      // disturbances
    double imx, imy, imz, imn;  // 1/(m+Xa), 1/(m+Ya), 1/(m+Za), 1/(Iz+Na)
    double lx, ly, lz, ln;      // linear damping
    double qx, qy, qz, qn;      // quadratic damping
};

// 1/d for a positive, normal d: v_rcp_f64 seed + 2 Newton steps (~1 ulp) -- an IEEE division is ~35 dependent instructions
__device__ __forceinline__ double rcp_nr(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    return fma(y, e, y);
}

__device__ __forceinline__ ModelPar make_par(const double* __restrict__ p) {
    ModelPar m;
    m.dx = p[0]; m.dy = p[1]; m.dz = p[2]; m.dn = p[3];
    m.imx = rcp_nr(kMass + p[4]); m.imy = rcp_nr(kMass + p[5]); m.imz = rcp_nr(kMass + p[6]);
    m.imn = rcp_nr(kIz + p[7]);
    m.lx = p[8]; m.ly = p[9]; m.lz = p[10]; m.ln = p[11];
    m.qx = p[12]; m.qy = p[13]; m.qz = p[14]; m.qn = p[15];
    return m;
}

// generalised forces of the 6 thrusters for the 4 wrench commands (bluerov2.py:95-121), constant over an RK step
// k3 / k4: roll / pitch moments.  The propulsion matrix gives the thrusters none (K rows 3, 4 are zero, bluerov2.py:95-100); they carry
// the roll / pitch DISTURBANCE moments of the 6-disturbance model variant (the symbols bluerov2.py:37-38 keeps commented out,
// entering dp, dq the way the other four disturbances enter their rows, :123-128) and are 0 in the shipped np = 16 model.
struct Wrench { double k0, k1, k2, k5, k3, k4; };

__device__ __forceinline__ Wrench make_wrench(const double* __restrict__ u) {
    const double ir = 1.0 / kRotor;
    const double t0 = (-u[0] + u[1] + u[3]) * ir, t1 = (-u[0] - u[1] - u[3]) * ir;
    const double t2 = (u[0] + u[1] - u[3]) * ir, t3 = (u[0] - u[1] + u[3]) * ir;
    const double t4 = -u[2] * ir;
    Wrench w;
    w.k0 = 0.707 * t0 + 0.707 * t1 - 0.707 * t2 - 0.707 * t3;
    w.k1 = 0.707 * t0 - 0.707 * t1 + 0.707 * t2 - 0.707 * t3;
    w.k2 = t4 + t4;
    w.k5 = 0.167 * t0 - 0.167 * t1 - 0.175 * t2 + 0.175 * t3;
    w.k3 = 0.0; w.k4 = 0.0;
    return w;
}

// sin and cos together, branch-free: 3-term Cody-Waite reduction by pi/2 (FMA) + the classic degree-13/14 minimax kernels
// on [-pi/4, pi/4].  Error < 1 ulp for |a| < ~1e6 rad (yaw_sum of the reference grows by 2 pi per lap, i.e. a few
// hundred rad at most); NaN/Inf propagate.  ocml's sincos() costs ~3x the registers (Payne-Hanek path) and pushed the
// linearisation kernel into scratch.
__device__ __forceinline__ void sincos_pio2(double a, double* sn, double* cs) {
    const double n = rint(a * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632679489655800e+00, a);
    r = fma(-n, 6.12323399573676603587e-17, r);
    r = fma(-n, -1.49738490485916983e-33, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)n & 3;
    const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// what one RK stage point contributes to the Jacobian: 6 trig values, 1/cos(theta), body velocities and rates
struct StagePoint {
    double sph, cph, sth, cth, sps, cps, icth;
    double vu, vv, vw, wp, wq, wr;
};

// xdot = f(x,u,p); also returns the stage point record
__device__ __forceinline__ void model_f(const double (&x)[NX], const Wrench& w, const ModelPar& m, double (&f)[NX],
                                        StagePoint& sp) {
    sincos_pio2(x[3], &sp.sph, &sp.cph);
    sincos_pio2(x[4], &sp.sth, &sp.cth);
    sincos_pio2(x[5], &sp.sps, &sp.cps);
    sp.icth = 1.0 / sp.cth;
    sp.vu = x[6]; sp.vv = x[7]; sp.vw = x[8]; sp.wp = x[9]; sp.wq = x[10]; sp.wr = x[11];
    const double r00 = sp.cps * sp.cth, r01 = sp.cps * sp.sth * sp.sph - sp.sps * sp.cph,
                 r02 = sp.sps * sp.sph + sp.cps * sp.cph * sp.sth;
    const double r10 = sp.sps * sp.cth, r11 = sp.cps * sp.cph + sp.sph * sp.sth * sp.sps,
                 r12 = sp.sth * sp.sps * sp.cph - sp.cps * sp.sph;
    const double r21 = sp.cth * sp.sph, r22 = sp.cth * sp.cph;
    f[0] = r00 * sp.vu + r01 * sp.vv + r02 * sp.vw;
    f[1] = r10 * sp.vu + r11 * sp.vv + r12 * sp.vw;
    f[2] = -sp.sth * sp.vu + r21 * sp.vv + r22 * sp.vw;
    const double tth = sp.sth * sp.icth;
    f[3] = sp.wp + sp.sps * tth * sp.wq + sp.cph * tth * sp.wr;  // sin(psi): reference quirk, bluerov2.py:133
    f[4] = sp.cph * sp.wq + sp.sph * sp.wr;
    f[5] = (sp.sph * sp.wq + sp.cph * sp.wr) * sp.icth;
    f[6] = (w.k0 - kBouy * sp.sth + m.dx + m.lx * sp.vu + m.qx * fabs(sp.vu) * sp.vu) * m.imx;
    f[7] = (w.k1 + kBouy * r21 + m.dy + m.ly * sp.vv + m.qy * fabs(sp.vv) * sp.vv) * m.imy;
    f[8] = (w.k2 + kBouy * r22 + m.dz + m.lz * sp.vw + m.qz * fabs(sp.vw) * sp.vw) * m.imz;
    f[9] = (w.k3 + (kIy - kIz) * sp.wq * sp.wr - kMzg * r21) * (1.0 / kIx);
    f[10] = (w.k4 + (kIz - kIx) * sp.wp * sp.wr - kMzg * sp.sth) * (1.0 / kIy);
    f[11] = (w.k5 - (kIy - kIx) * sp.wp * sp.wq + m.dn + m.ln * sp.wr + m.qn * fabs(sp.wr) * sp.wr) * m.imn;
}

}  // namespace brov

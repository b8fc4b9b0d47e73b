// tsff_device.h -- device-side building blocks of the form-factor kernels (gfx950 / CDNA4 only).
//
// Everything is float64.  The per-point physics follows FormFactor.__call__
// (reference core/physics/form_factor.py:182-298, equations restated in DESIGN.md section 3);
// the reverse sweep is a hand-written adjoint of exactly that arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/tsff.h"

#ifndef TSFF_BRANCHFREE
#define TSFF_BRANCHFREE 1  // table lookups without divergent branches (one scheduling region per point)
#endif

namespace tsff {

constexpr int kThreads = 256;    // 4 wavefronts of 64
constexpr int kStrip = 4;        // consecutive wavelength samples owned by one thread
constexpr int kNP_MAX = TSFF_NP(TSFF_MAX_ION);

// physical constants (form_factor.py:123-125, 207-209)
constexpr double kC = 2.99792458e10;
constexpr double kMe = 510.9896 / (kC * kC);
constexpr double kMp = kMe * 1836.1;
constexpr double kRe = 2.8179e-13;
constexpr double kPi = 3.14159265358979323846;
constexpr double kEsq = kMe * kC * kC * kRe;
constexpr double kC0sq = 4.0 * kPi * kEsq / kMe;  // C0^2, omega_pe^2 = C0^2 * ne
constexpr double kOmgLnum = 2.0 * kPi * 1e7 * kC;
constexpr double kInvSqrt2Pi = 0.39894228040143267794;
constexpr double kSqrt2 = 1.41421356237309504880;

// xi2 grid of the Z' and W tables (form_factor.py:138): arange(-8.2, 8.2, 0.01)
constexpr double kXi2_0 = -8.2;
constexpr double kXi2_h = 0.01;
constexpr double kXi2_ih = 100.0;
constexpr int kNXi2 = TSFF_NXI2;
constexpr int kNXi1 = TSFF_NXI1;

// ------------------------------------------------------------------------------------------
// fast float64 reciprocal / square root for operands in the normal range (no denormal or
// overflow rescaling, which the library forms spend ~40 % of their instructions on): the hardware
// seed (v_rcp_f64 / v_rsq_f64) plus Newton / Goldschmidt refinement.  Measured on gfx950 over 4M log-uniform
// arguments in [1e-30, 1e30] (scripts/seed_precision.hip): the seeds are good to 2^-24.4 / 2^-24.2; ONE step gives
// 2.2e-15 (1/x), 4.1e-15 (1/sqrt) and a correctly rounded sqrt (with the final residual correction); a second step
// reaches the last bit and costs 2 (3) more dependent FMAs on the critical path of every point.  The path's parity
// budget is 1e-9 (north_star: 1e-5), so one step is what is used.
// ------------------------------------------------------------------------------------------
#ifndef TSFF_NEWTON
#define TSFF_NEWTON 1  // refinement steps after the hardware seed (see the table above)
#endif
__device__ __forceinline__ double frcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
#if TSFF_NEWTON > 1
  e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
#endif
  return y;
}

// s = sqrt(x), is = 1/sqrt(x)
__device__ __forceinline__ void fsqrt2(double x, double& s, double& is) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
#if TSFF_NEWTON > 1
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
#endif
  const double d = __builtin_fma(-g, g, x);
  s = __builtin_fma(d, h, g);
  is = h + h;
}

// A value every lane of the wavefront holds identically (lineout scalars, per-angle constants):
// move it to SGPRs so it stops occupying a VGPR pair in every lane.
__device__ __forceinline__ double uni(double x) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------
// wavefront / workgroup reductions (64-wide wavefronts, xor-shuffle butterflies)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the workgroup; result valid in every thread.  scratch: >= 4 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[w] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// max with lowest-index tie break; result valid in every thread.  scratch: >= 8 doubles.
__device__ __forceinline__ void block_argmax(double& v, int& idx, double* scratch) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(idx, o, 64);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { scratch[w] = v; scratch[4 + w] = (double)idx; }
  __syncthreads();
  v = scratch[0]; idx = (int)scratch[4];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    double ov = scratch[k]; int oi = (int)scratch[4 + k];
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}

// ------------------------------------------------------------------------------------------
// table lookups (tables live in LDS)
// ------------------------------------------------------------------------------------------
struct Tables {
  const double2* zp;   // [1640] (Re Z', Im Z') on xi2
  const double* W;     // [1640] Re(chi_e) table on xi2
  const double2* ht;   // [nvx]  (ln fe, node slope) on vx            (k_fe_prepare)
  const double2* hc;   // [2*(nvx-1)] cubic coefficients per interval: (f0, m0), (c2, c3)   (spectrum kernels)
  const double2* hcm;  // the same for d(ln fe)/dm (gradient w.r.t. the DLM order), or nullptr
  const double* Wm;    // [1640] dW/dm, or nullptr
  double* Wb;          // [1640] adjoint of W            (gradient w.r.t. the distribution function itself, GM == 2)
  double* Hy;          // [nvx]  adjoint of the ln fe node values
  double* Hs;          // [nvx]  adjoint of the ln fe node slopes
  double vx0, dv, idv, vxlast;
  double u0, utop;     // -vx0 / dv and (nvx - 1)(1 - 2^-52): hermite_lookup_c's position in cell units and its clamp
  const double* etab;  // [64] 2^(j/64) in LDS (fexp_t), or nullptr: polynomial exp
  int nvx;
};
__device__ __forceinline__ void tables_set_grid(Tables& T, double vx0, double dv, int nvx) {
  T.vx0 = vx0; T.dv = dv; T.idv = 1.0 / dv; T.vxlast = vx0 + (nvx - 1) * dv; T.nvx = nvx;
  T.u0 = -vx0 * T.idv; T.utop = (double)(nvx - 1) * (1.0 - 1.1102230246251565e-16);
}

// Gradient w.r.t. the tabulated distribution function (GM == 2): every point scatters the adjoint of its two table
// lookups into the table adjoints in LDS.  Neighbouring points of one thread fall into the same table interval most of
// the time, so a thread keeps run-length accumulators and only issues LDS atomics when its interval changes.
struct FeAcc {
  int iw, ih;
  double w0, w1, y0, y1, s0, s1;
};
__device__ __forceinline__ void fe_acc_init(FeAcc& a) { a.iw = a.ih = -1; a.w0 = a.w1 = a.y0 = a.y1 = a.s0 = a.s1 = 0.0; }
__device__ __forceinline__ void fe_flush_w(FeAcc& a, double* Wb) {
  if (a.iw >= 0) { atomicAdd(&Wb[a.iw], a.w0); atomicAdd(&Wb[a.iw + 1], a.w1); }
  a.w0 = a.w1 = 0.0;
}
__device__ __forceinline__ void fe_flush_h(FeAcc& a, double* Hy, double* Hs) {
  if (a.ih >= 0) {
    atomicAdd(&Hy[a.ih], a.y0); atomicAdd(&Hy[a.ih + 1], a.y1);
    atomicAdd(&Hs[a.ih], a.s0); atomicAdd(&Hs[a.ih + 1], a.s1);
  }
  a.y0 = a.y1 = a.s0 = a.s1 = 0.0;
}
// adjoint of w_lookup w.r.t. the table: wb = adjoint of the interpolated value
__device__ __forceinline__ void fe_add_w(FeAcc& a, double* Wb, double xe, double wb) {
  const double xlast = kXi2_0 + (kNXi2 - 1) * kXi2_h;
  const double u = (xe - kXi2_0) * kXi2_ih;
  int i = (int)u;
  i = i < 0 ? 0 : (i > kNXi2 - 2 ? kNXi2 - 2 : i);
  double t = (xe - (kXi2_0 + i * kXi2_h)) * kXi2_ih;
  t = xe < kXi2_0 ? 0.0 : (xe > xlast ? 1.0 : t);  // clamped lookups read an end node
  if (i != a.iw) { fe_flush_w(a, Wb); a.iw = i; }
  a.w0 += wb * (1.0 - t);
  a.w1 += wb * t;
}
// adjoint of hermite_lookup w.r.t. the node values and slopes: hb = adjoint of H = ln f_e(x)
__device__ __forceinline__ void fe_add_h(FeAcc& a, const Tables& T, double x, double hb) {
  if (x < T.vx0 || x > T.vxlast) return;  // constant -50 outside the grid
  const double u = (x - T.vx0) * T.idv;
  int i = (int)u;
  i = i < 0 ? 0 : (i > T.nvx - 2 ? T.nvx - 2 : i);
  const double t = (x - (T.vx0 + i * T.dv)) * T.idv;
  if (i != a.ih) { fe_flush_h(a, T.Hy, T.Hs); a.ih = i; }
  const double t2 = t * t, t3 = t2 * t;
  a.y0 += hb * (2.0 * t3 - 3.0 * t2 + 1.0);
  a.y1 += hb * (3.0 * t2 - 2.0 * t3);
  a.s0 += hb * (t3 - 2.0 * t2 + t) * T.dv;
  a.s1 += hb * (t3 - t2) * T.dv;
}

// jnp.interp(xie, xi2, W): clamps to the end values outside the table (form_factor.py:270)
__device__ __forceinline__ void w_lookup(const double* W, double xe, double& w, double& dw) {
#pragma clang fp contract(off)   // (every rounding that reaches a spectrum is written out: the same bits from every kernel)
  // position in cell units, clamped into the table: outside it the clamped cell reproduces the end value (t = 0 at the left
  // end, t = 1 - 2^-52 at the right end) -- floor / subtract instead of convert-back / multiply-add, and ONE comparison
  // (clamped or not) for the slope (round 2: 6 instructions where the index arithmetic took 10)
  constexpr double kTop = (double)(kNXi2 - 1) * (1.0 - 1.1102230246251565e-16);
  const double u = __builtin_fma(xe, kXi2_ih, -kXi2_0 * kXi2_ih);
  const double uc = fmin(fmax(u, 0.0), kTop);
  const double fl = __builtin_floor(uc);
  const double t = uc - fl;
  const int i = (int)fl;
  const double a = W[i], b = W[i + 1];
  const double d = b - a;
  w = __builtin_fma(t, d, a);
  dw = (u == uc) ? d * kXi2_ih : 0.0;
}

// cubic coefficients of interval i of the Hermite interpolant in t = (x - vx_i)/dv:
// H = f0 + t (m0 + t (c2 + t c3))
__device__ __forceinline__ void hermite_coeffs(double2 a, double2 b, double dv, double2& c01, double2& c23) {
  const double f0 = a.x, f1 = b.x, m0 = a.y * dv, m1 = b.y * dv;
  c01 = make_double2(f0, m0);
  c23 = make_double2(-3.0 * f0 + 3.0 * f1 - 2.0 * m0 - m1, 2.0 * f0 - 2.0 * f1 + m0 + m1);
}

// the same lookup as hermite_lookup() below from the per-interval coefficient table
__device__ __forceinline__ void hermite_lookup_c(const Tables& T, double x, double& H, double& dH) {
#pragma clang fp contract(off)
  // (same index arithmetic as w_lookup; T.u0 = -vx0 / dv, T.utop = (nvx - 1)(1 - 2^-52): a position that the clamp moves
  //  lies outside the grid, where the interpolant is the constant -50)
  const double u = __builtin_fma(x, T.idv, T.u0);
  const double uc = fmin(fmax(u, 0.0), T.utop);
  const double fl = __builtin_floor(uc);
  const double t = uc - fl;
  const int i = (int)fl;
  const double2 c01 = T.hc[2 * i], c23 = T.hc[2 * i + 1];
  const bool out = u != uc;
  const double Hi = __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, c23.y, c23.x), c01.y), c01.x);
  const double dHi = __builtin_fma(t, __builtin_fma(3.0 * t, c23.y, 2.0 * c23.x), c01.y) * T.idv;
  H = out ? -50.0 : Hi;
  dH = out ? 0.0 : dHi;
}

// interpax.interp1d(x, vx, ln fe, method="cubic", extrap=[-50,-50])  (form_factor.py:256,263)
// returns H = ln f(x) and dH/dx
__device__ __forceinline__ void hermite_lookup(const Tables& T, double x, double& H, double& dH) {
  double u = (x - T.vx0) * T.idv;
  int i = (int)u;
  i = i < 0 ? 0 : (i > T.nvx - 2 ? T.nvx - 2 : i);
  const double t = (x - (T.vx0 + i * T.dv)) * T.idv;
  const double2 a = T.ht[i], b = T.ht[i + 1];
  const double f0 = a.x, f1 = b.x, m0 = a.y * T.dv, m1 = b.y * T.dv;
  const double c2 = -3.0 * f0 + 3.0 * f1 - 2.0 * m0 - m1;
  const double c3 = 2.0 * f0 - 2.0 * f1 + m0 + m1;
  if (x < T.vx0 || x > T.vxlast) { H = -50.0; dH = 0.0; }
  else {
    H = f0 + t * (m0 + t * (c2 + t * c3));
    dH = (m0 + t * (2.0 * c2 + 3.0 * t * c3)) * T.idv;
  }
}

// ------------------------------------------------------------------------------------------
// lineout-level scalars for one gradient point (form_factor.py:182-253)
// ------------------------------------------------------------------------------------------
template <int NI>
struct Phys {  // physical parameters of one lineout (after ThomsonParams.__call__)
  double Te, ne, m, lam, amp1, amp2, amp3, neg, teg, ud, Va;
  double Ti[NI], Z[NI], A[NI], fr[NI];
  double fsum;  // sum of the un-normalised fractions (ts_params.py:559-562)
};

template <int NI>
struct LineS {
  double wpe2, wL, kL, ivTe, a_e, pref, Ud, Vd;
  double i2wL;  // 2 / wL (derived; carries no adjoint of its own)
  double m;     // adjoint accumulator of the DLM order (through the ln f_e and W tables); unused in the forward
  double ixi[NI], a_i[NI], cs[NI];
  double hai[NI];   // -a_i / 2 (derived: chi_i = sum hai ik^2 Z'; carries no adjoint of its own)
};

// gradient-point factor: linspace(1 - v/200, 1 + v/200, G)[g] = 1 + v*cg   (form_factor.py:182-195)
__device__ __forceinline__ double grad_coef(int g, int G) {
  return G == 1 ? -1.0 / 200.0 : -1.0 / 200.0 + (double)g / (100.0 * (double)(G - 1));
}

template <int NI>
__device__ __forceinline__ void make_lines(const Phys<NI>& p, double lam_shift, int g, int G, LineS<NI>& L) {
  const double cg = grad_coef(g, G);
  const double ne_g = 1.0e20 * p.ne * (1.0 + p.neg * cg);
  const double Te_g = p.Te * (1.0 + p.teg * cg);
  L.wL = kOmgLnum / (p.lam + lam_shift);
  L.i2wL = 2.0 / L.wL;
  L.m = 0.0;
  L.wpe2 = kC0sq * ne_g;
  L.kL = sqrt(L.wL * L.wL - L.wpe2) / kC;
  L.ivTe = 1.0 / sqrt(Te_g / kMe);
  L.a_e = L.wpe2 * L.ivTe * L.ivTe;
  L.pref = kRe * kRe * ne_g / (2.0 * kPi * kC);
  L.Ud = p.ud * 1e6;
  L.Vd = p.Va * 1e6;
  double Zbar = 0.0;
#pragma unroll
  for (int s = 0; s < NI; ++s) Zbar += p.Z[s] * p.fr[s];
#pragma unroll
  for (int s = 0; s < NI; ++s) {
    const double Ms = p.A[s] * kMp;
    const double vTi = sqrt(p.Ti[s] / Ms);
    L.ixi[s] = 1.0 / (kSqrt2 * vTi);
    L.a_i[s] = kC0sq * kMe * p.Z[s] * p.Z[s] * p.fr[s] * ne_g / (Zbar * p.Ti[s]);
    L.cs[s] = p.fr[s] * p.Z[s] * p.Z[s] / (Zbar * vTi);
    L.hai[s] = -0.5 * L.a_i[s];
  }
}

// the lineout scalars are identical in every lane: keep them in SGPRs
template <int NI>
__device__ __forceinline__ void make_lines_uniform(const Phys<NI>& p, double lam_shift, int g, int G, LineS<NI>& L) {
  make_lines<NI>(p, lam_shift, g, G, L);
  L.wpe2 = uni(L.wpe2); L.wL = uni(L.wL); L.i2wL = uni(L.i2wL); L.kL = uni(L.kL); L.ivTe = uni(L.ivTe);
  L.a_e = uni(L.a_e); L.pref = uni(L.pref); L.Ud = uni(L.Ud); L.Vd = uni(L.Vd);
#pragma unroll
  for (int s = 0; s < NI; ++s) { L.ixi[s] = uni(L.ixi[s]); L.a_i[s] = uni(L.a_i[s]); L.cs[s] = uni(L.cs[s]); L.hai[s] = uni(L.hai[s]); }
}

// adjoint of make_lines: LB holds dL/d(LineS fields) summed over the points of gradient point g;
// accumulates into pb[slot] (adjoint w.r.t. the PHYSICAL, renormalised parameters).
template <int NI>
__device__ __forceinline__ void make_lines_adjoint(const Phys<NI>& p, double lam_shift, int g, int G,
                                                   const LineS<NI>& L, const LineS<NI>& LB, double* pb) {
  const double cg = grad_coef(g, G);
  const double gfn = 1.0 + p.neg * cg, gft = 1.0 + p.teg * cg;
  const double ne_g = 1.0e20 * p.ne * gfn;
  const double Te_g = p.Te * gft;
  double wpe2b = LB.wpe2, wLb = LB.wL, ivTeb = LB.ivTe;
  double ne_gb = LB.pref * (kRe * kRe / (2.0 * kPi * kC));
  // a_e = wpe2 * ivTe^2
  wpe2b += LB.a_e * L.ivTe * L.ivTe;
  ivTeb += LB.a_e * 2.0 * L.wpe2 * L.ivTe;
  // kL = sqrt(wL^2 - wpe2)/c
  wLb += LB.kL * L.wL / (kC * kC * L.kL);
  wpe2b += LB.kL * (-0.5 / (kC * kC * L.kL));
  double Zbar = 0.0;
#pragma unroll
  for (int s = 0; s < NI; ++s) Zbar += p.Z[s] * p.fr[s];
  double Zbarb = 0.0;
#pragma unroll
  for (int s = 0; s < NI; ++s) {
    const double Ms = p.A[s] * kMp;
    const double vTi = sqrt(p.Ti[s] / Ms);
    const double Z2 = p.Z[s] * p.Z[s];
    double Tib = 0.0, Zb = 0.0, frb = 0.0;
    // a_i = C0^2 me Z^2 fr ne_g / (Zbar Ti)
    const double ai_over_fr = kC0sq * kMe * Z2 * ne_g / (Zbar * p.Ti[s]);
    Zb += LB.a_i[s] * 2.0 * L.a_i[s] / p.Z[s];
    frb += LB.a_i[s] * ai_over_fr;
    ne_gb += LB.a_i[s] * L.a_i[s] / ne_g;
    Zbarb -= LB.a_i[s] * L.a_i[s] / Zbar;
    Tib -= LB.a_i[s] * L.a_i[s] / p.Ti[s];
    // ixi = 1/(sqrt2 vTi)
    Tib += LB.ixi[s] * (-0.5 * L.ixi[s] / p.Ti[s]);
    // cs = fr Z^2/(Zbar vTi)
    Tib += LB.cs[s] * (-0.5 * L.cs[s] / p.Ti[s]);
    Zb += LB.cs[s] * 2.0 * L.cs[s] / p.Z[s];
    frb += LB.cs[s] * Z2 / (Zbar * vTi);
    Zbarb -= LB.cs[s] * L.cs[s] / Zbar;
    pb[TSFF_P_ION0 + 4 * s + TSFF_ION_TI] += Tib;
    pb[TSFF_P_ION0 + 4 * s + TSFF_ION_Z] += Zb;
    pb[TSFF_P_ION0 + 4 * s + TSFF_ION_FRACT] += frb;
  }
#pragma unroll
  for (int s = 0; s < NI; ++s) {
    pb[TSFF_P_ION0 + 4 * s + TSFF_ION_Z] += Zbarb * p.fr[s];
    pb[TSFF_P_ION0 + 4 * s + TSFF_ION_FRACT] += Zbarb * p.Z[s];
  }
  // ivTe = sqrt(me/Te_g)
  const double Te_gb = ivTeb * (-0.5 * L.ivTe / Te_g);
  ne_gb += wpe2b * kC0sq;
  const double lam0 = p.lam + lam_shift;
  pb[TSFF_P_LAM] += wLb * (-L.wL / lam0);
  pb[TSFF_P_NE] += ne_gb * 1.0e20 * gfn;
  pb[TSFF_P_NE_GRADIENT] += ne_gb * 1.0e20 * p.ne * cg;
  pb[TSFF_P_TE] += Te_gb * gft;
  pb[TSFF_P_TE_GRADIENT] += Te_gb * p.Te * cg;
  pb[TSFF_P_UD] += LB.Ud * 1e6;
  pb[TSFF_P_VA] += LB.Vd * 1e6;
  pb[TSFF_P_M] += LB.m;
}

// ------------------------------------------------------------------------------------------
// per-point physics
// ------------------------------------------------------------------------------------------

// exp(x) for x <= ~1 (arguments here are ln f_e in [-50, 2] and -xi_i^2 <= 0): Cody-Waite reduction by
// ln 2 and a Taylor polynomial on |r| <= ln2/2; no overflow handling, underflow to 0 through v_ldexp_f64.
// TSFF_FEXP: 1 degree 13 Horner (truncation 4e-18, ~1 ulp), 2 degree 11 Horner (6.4e-15), 3 degree 11 Estrin.
#ifndef TSFF_FEXP
#define TSFF_FEXP 1
#endif
// CLAMP = false: the caller guarantees a finite argument (ln f_e from the table or -50): no guard against -inf
template <bool CLAMP = true>
__device__ __forceinline__ double fexp(double x) {
#if !TSFF_FEXP
  return exp(x);
#endif
  if (CLAMP) x = fmax(x, -800.0);
  const double n = __builtin_rint(x * 1.4426950408889634074);   // (not contractible: rint takes the product)
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
#if TSFF_FEXP == 1
  double p = 1.6059043836821613e-10;                 // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);     // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);    // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);    // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06);   // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);     // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);    // 1/7!
  p = __builtin_fma(p, r, 1.388888888888889e-03);    // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);    // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02);   // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01);   // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
#elif TSFF_FEXP == 2
  double p = 2.505210838544172e-08;                  // 1/11!  (truncation |r|^12 / 12! <= 6.4e-15 on |r| <= ln2/2)
  p = __builtin_fma(p, r, 2.755731922398589e-07);    // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06);   // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);     // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);    // 1/7!
  p = __builtin_fma(p, r, 1.388888888888889e-03);    // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);    // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02);   // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01);   // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
#else
  // degree 11, Estrin's scheme: 11 FMAs + 3 multiplies in a dependency chain of depth 4 instead of 12
  const double r2 = r * r;
  const double a0 = __builtin_fma(r, 1.0, 1.0);
  const double a1 = __builtin_fma(r, 1.6666666666666666e-01, 0.5);
  const double a2 = __builtin_fma(r, 8.333333333333333e-03, 4.1666666666666664e-02);
  const double a3 = __builtin_fma(r, 1.984126984126984e-04, 1.388888888888889e-03);
  const double a4 = __builtin_fma(r, 2.7557319223985893e-06, 2.48015873015873e-05);
  const double a5 = __builtin_fma(r, 2.505210838544172e-08, 2.755731922398589e-07);
  const double r4 = r2 * r2;
  const double b0 = __builtin_fma(a1, r2, a0);
  const double b1 = __builtin_fma(a3, r2, a2);
  const double b2 = __builtin_fma(a5, r2, a4);
  const double r8 = r4 * r4;
  const double d0 = __builtin_fma(b1, r4, b0);
  const double p = __builtin_fma(b2, r8, d0);
#endif
  return ldexp(p, (int)n);
}

// exp(x) for x <= ~1 with a 64-entry table: x = (64 n' + j) ln2/64 + r, |r| <= ln2/128, exp(x) = 2^n' 2^(j/64) exp(r) with a
// degree-5 Taylor polynomial (truncation r^6/720 <= 3.5e-17) -- 15 VALU instructions and one LDS read where the
// degree-13 polynomial of fexp takes 20 (two exps per (lambda, theta) point: ln f_e -> f_e and exp(-xi_i^2)).  tab[j] = 2^(j/64)
// correctly rounded (host); total error ~1.5 ulp.
constexpr int kNExpTab = 64;
template <bool CLAMP = true>
__device__ __forceinline__ double fexp_t(double x, const double* __restrict__ tab) {
  if (CLAMP) x = fmax(x, -800.0);
  const double n = __builtin_rint(x * 92.332482616893656908);          // 64 / ln 2
  double r = __builtin_fma(n, -0.01083042469326756, x);           // ln2/64, high part (trailing zeros: n * hi exact)
  r = __builtin_fma(n, -2.9815858269852933e-12, r);              // low part
  double p = 8.333333333333333e-03;                                    // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02);
  p = __builtin_fma(p, r, 1.6666666666666666e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const int ni = (int)n;
  return ldexp(p * tab[ni & (kNExpTab - 1)], ni >> 6);
}

struct Base {  // quantities needed at point j AND as the right neighbour of point j-1
  double ks, k2, ik, wd, xe, F, dH;   // k = k2*ik, v_ph = wd*ik
};

// k_s = sqrt(w_s^2 - wpe^2)/c (form_factor.py:218; angle independent)
__device__ __forceinline__ double ks_eval(double ws, double wpe2) {
  double s, is;
  fsqrt2(__builtin_fma(ws, ws, -wpe2), s, is);
  return s * (1.0 / kC);
}

// k^2 = k_s^2 + k_L^2 - 2 k_s k_L cos(theta) (form_factor.py:220), in the one rounding sequence every kernel uses
template <int NI>
__device__ __forceinline__ double base_k2(double ks, double ct, const LineS<NI>& L) {
#pragma clang fp contract(off)
  const double c1 = (2.0 * L.kL) * ct, c0 = L.kL * L.kL;   // wavefront-uniform: hoisted out of the strip loops
  return __builtin_fma(ks, ks - c1, c0);
}

template <int NI>
__device__ __forceinline__ void base_eval(double ws, double ks, double ct, const LineS<NI>& L, const Tables& T,
                                          Base& b) {
#pragma clang fp contract(off)
  b.ks = ks;
  b.k2 = base_k2<NI>(ks, ct, L);
  double k;
  fsqrt2(b.k2, k, b.ik);
  b.wd = __builtin_fma(-k, L.Vd, ws - L.wL);               // :216, 222-223
  b.xe = __builtin_fma(b.wd, b.ik, -L.Ud) * L.ivTe;        // :253
  double H;
  hermite_lookup_c(T, b.xe, H, b.dH);
  b.F = T.etab ? fexp_t<false>(H, T.etab) : fexp<false>(H);   // :256  (T.etab is set or not per kernel: folded at compile time)
}

// ion terms of one species at normalised phase velocity xi (form_factor.py:243-249, 277-280):
// Z'(xi) = jnp.interp(xi, xi2, Zp, left=xi**-2 / 0, right=xi**-2 / 0) with its slope, and
// gs = exp(-xi^2)/sqrt(2 pi).
// The table is held for xi >= 0 only (kNZh = 821 nodes 0, 0.01, ..., 8.2): the shipped rdWT / idWT data are exactly even
// (Re Z') and odd (Im Z') -- tsff_create verifies it -- so node i of the full table is node |i - 820| of the half table with
// the sign of i - 820 on the odd part.  The table's own domain stays the reference's asymmetric [-8.2, 8.19] (arange): outside
// it the asymptote applies.  13 KB of LDS per workgroup less than the full table.
// ZH = false: the full 1640-node table (no index mirror, no sign handling: about 1 % fewer instructions per point); the
// spectrum kernels use it whenever the LDS budget does not need the 13 KB back (launch_spectrum decides).  Both forms
// interpolate between the same two node values with the same weight in the same order: a spectrum does not depend on which
// one the launch plan picked (test_one_sweep_kernel_random_geometry caught an earlier half-table form that interpolated from
// the other end for xi < 0: 1e-16 per lookup, visible in the last bits when two plans of one deck differed in ZH).
constexpr int kNZh = kNXi2 / 2 + 1;
template <bool ZH = true>
__device__ __forceinline__ void ion_terms(const double2* zp, double xi, double& zr, double& zi, double& dzr,
                                          double& dzi, double& gs, const double* etab = nullptr) {
#pragma clang fp contract(off)
  gs = (etab ? fexp_t(-(xi * xi), etab) : fexp(-(xi * xi))) * kInvSqrt2Pi;
  bool outf = false;
  if (ZH) {
    // the cell and the weight exactly as the full table takes them (below), then the two nodes through the mirror: node i of the
    // full table, xi2[i] = -8.2 + 0.01 i, is node |i - 820| of the half table with the sign of i - 820 on the odd part.  The
    // interpolation runs over the same two values in the same order: the same bits as ZH = false.
    constexpr double kTop = (double)(kNXi2 - 1) * (1.0 - 1.1102230246251565e-16);
    constexpr int kMid = kNZh - 1;   // full-table index of xi = 0
    const double u = __builtin_fma(xi, kXi2_ih, -kXi2_0 * kXi2_ih);
    const double uc = fmin(fmax(u, 0.0), kTop);
    const double fl = __builtin_floor(uc);
    const double t = uc - fl;
    const int i = (int)fl, da = i - kMid, db = da + 1;
    double2 a = zp[da < 0 ? -da : da], b = zp[db < 0 ? -db : db];
    const int sx = da & (int)0x80000000;   // (node i + 1 = 0 when da = -1: its odd part is an exact zero, the sign is immaterial)
    a.y = __hiloint2double(__double2hiint(a.y) ^ sx, __double2loint(a.y));
    b.y = __hiloint2double(__double2hiint(b.y) ^ sx, __double2loint(b.y));
    const double dr = b.x - a.x, di = b.y - a.y;
    zr = __builtin_fma(t, dr, a.x); zi = __builtin_fma(t, di, a.y); dzr = dr * kXi2_ih; dzi = di * kXi2_ih;
    outf = u != uc;
  } else {
    constexpr double kTop = (double)(kNXi2 - 1) * (1.0 - 1.1102230246251565e-16);
    const double u = __builtin_fma(xi, kXi2_ih, -kXi2_0 * kXi2_ih);
    const double uc = fmin(fmax(u, 0.0), kTop);
    const double fl = __builtin_floor(uc);
    const double t = uc - fl;
    const int i = (int)fl;
    const double2 a = zp[i], b = zp[i + 1];
    const double dr = b.x - a.x, di = b.y - a.y;
    zr = __builtin_fma(t, dr, a.x); zi = __builtin_fma(t, di, a.y); dzr = dr * kXi2_ih; dzi = di * kXi2_ih;
    outf = u != uc;   // (outside the table <=> the clamp moved the position)
  }
  const bool out = outf;
  // straight-line form (selects instead of branches: one scheduling region per point; wavefront-uniform shortcuts for the
  // far EPW window and for the in-table IAW window were both measured slower)
  const double i2 = frcp(xi * xi);
  const double di2 = -2.0 * i2 * i2 * xi;
  zr = out ? i2 : zr; zi = out ? 0.0 : zi; dzr = out ? di2 : dzr; dzi = out ? 0.0 : dzi;
}

// P(lambda_j, theta_a) of one gradient point (form_factor.py:247-296): the forward algebra shared by every kernel that
// evaluates a point (forward sweep, reverse sweep, one-sweep kernel).  The roundings that reach the spectrum are pinned by
// explicit FMAs, so that a spectrum comes out bit-identical from tsff_forward and from tsff_loss_grad whatever kernel runs.
template <int NI>
struct PointF {
  double ik2, ike2, pike, vph, cim, gsum, Wl, dW, idx, D, cer, cei, opc, er, ei, ieps2, ce2, ci2, N, t1, S, dop;
  double xi[NI], zr[NI], zi[NI], dzr[NI], dzi[NI], hk[NI], gs[NI];   // hk = -a_i / (2 k^2)
};
// FAR = true: the caller guarantees |xi_i| > kXiFar for every species at this point.  Then the ion terms are the asymptote
// Z' = xi^-2 + 0 i (outside the table, :247) and exp(-xi^2) underflows to exactly 0, so Im chi_i, the ion numerator gsum and
// everything multiplied by them vanish IDENTICALLY -- the pruned algebra below gives the same bits as the general one (adding
// +-0, multiplying an exact 0), at about 50 instructions per point less (no exp, no table lookup, no selects).  Whole
// wavefronts of the electron feature qualify: every sample away from the laser line (see far_wavefront()).
constexpr double kXiFar = 28.0;   // exp(-28^2) = 2^-1131 < the smallest denormal; 28 > 8.2, the edge of the Z' table
template <int NI, bool ZH, bool FAR = false>
__device__ __forceinline__ void point_core(const Base& b, const Base& bn, bool has_next, const LineS<NI>& L, const Tables& T,
                                           PointF<NI>& p) {
#pragma clang fp contract(off)
  p.ik2 = b.ik * b.ik;
  p.ike2 = L.a_e * p.ik2;
  p.pike = kPi * p.ike2;
  p.vph = b.wd * b.ik;
  double opc = 1.0, cim = 0.0, gsum = 0.0;   // opc = 1 + Re chi_i
#pragma unroll
  for (int s = 0; s < NI; ++s) {
    p.xi[s] = p.vph * L.ixi[s];                              // :243
    const double hk = L.hai[s] * p.ik2;
    p.hk[s] = hk;
    if (FAR) {
      const double i2 = frcp(p.xi[s] * p.xi[s]);             // (the very expressions of ion_terms' asymptote branch)
      p.zr[s] = i2; p.dzr[s] = -2.0 * i2 * i2 * p.xi[s];
      p.zi[s] = 0.0; p.dzi[s] = 0.0; p.gs[s] = 0.0;
      opc = __builtin_fma(hk, i2, opc);
    } else {
      ion_terms<ZH>(T.zp, p.xi[s], p.zr[s], p.zi[s], p.dzr[s], p.dzi[s], p.gs[s], T.etab);
      opc = __builtin_fma(hk, p.zr[s], opc);                 // :249
      cim = s == 0 ? hk * p.zi[s] : __builtin_fma(hk, p.zi[s], cim);
      gsum = s == 0 ? L.cs[s] * p.gs[s] : __builtin_fma(L.cs[s], p.gs[s], gsum);   // :277-280
    }
  }
  p.cim = cim; p.gsum = gsum;
  w_lookup(T.W, b.xe, p.Wl, p.dW);
  p.idx = has_next ? frcp(bn.xe - b.xe) : 0.0;
  p.D = has_next ? (bn.F - b.F) * p.idx : 0.0;               // :258-259
  p.cer = -p.ike2 * p.Wl;                                    // :270-271
  p.cei = p.pike * p.D;                                      // :261
  p.opc = opc;
  p.er = p.opc + p.cer;
  p.ei = FAR ? p.cei : p.cei + cim;                          // :274   (cim = -0: x + -0 = x)
  const double eps2 = __builtin_fma(p.er, p.er, p.ei * p.ei);
  p.ieps2 = frcp(eps2);
  if (FAR) {
    p.ce2 = 0.0;                                             // (only ever multiplied by gsum = 0 or its adjoint)
    p.ci2 = p.opc * p.opc;                                   // fma(opc, opc, +0) = round(opc^2)
    p.N = (p.ci2 * b.F) * L.ivTe;                            // fma(+0, ce2, X) = X
  } else {
    p.ce2 = __builtin_fma(p.cer, p.cer, p.cei * p.cei);
    p.ci2 = __builtin_fma(p.opc, p.opc, cim * cim);
    p.N = __builtin_fma(gsum, p.ce2, (p.ci2 * b.F) * L.ivTe);  // :282-288
  }
  p.t1 = b.ik * p.ieps2;
  p.S = p.N * p.t1;
  p.dop = __builtin_fma(b.wd, L.i2wL, 1.0);                  // :291
}
// point_forward_sd: S (1 + 2 w/w_L) -- everything but the factor pref ws^2, which k_spectrum applies once per angle
// (pref) and once per wavelength sample (ws^2) instead of once per point
template <int NI, bool ZH = true, bool FAR = false>
__device__ __forceinline__ double point_forward_sd(const Base& b, const Base& bn, bool has_next, const LineS<NI>& L,
                                                   const Tables& T) {
  PointF<NI> p;
  point_core<NI, ZH, FAR>(b, bn, has_next, L, T, p);
  return p.S * p.dop;
}

// Is |xi_i| > kXiFar (with 2 % margin) guaranteed at every point of the samples [jw0, jw1] (all angles, all species)?
//   |xi_i| = |w - k V| / k * ixi >= (min |w| - kmax |V|) / kmax * ixi,   kmax = k_s,max + k_L  (triangle inequality),
// w = ws - wL of one sign over the range (ws and k_s fall monotonically with the sample index).  Evaluated with
// wavefront-uniform arguments; the result is made a scalar.
template <int NI>
__device__ __forceinline__ bool far_range_w(double w0, double w1, const LineS<NI>& L);
template <int NI>
__device__ __forceinline__ bool far_range(const double* __restrict__ omgs, int jw0, int jw1, const LineS<NI>& L) {
  return far_range_w<NI>(omgs[jw0], omgs[jw1], L);
}
// the same with the frequencies of the two ends in hand
template <int NI>
__device__ __forceinline__ bool far_range_w(double w0, double w1, const LineS<NI>& L) {
  const double d0 = w0 - L.wL, d1 = w1 - L.wL;
  const double kmax = ks_eval(w0, L.wpe2) + L.kL;
  const double dmin = fmin(fabs(d0), fabs(d1)) - kmax * fabs(L.Vd);
  bool ok = (d0 > 0.0) == (d1 > 0.0) && d0 != 0.0 && d1 != 0.0;
#pragma unroll
  for (int s = 0; s < NI; ++s) ok = ok && dmin * L.ixi[s] > (kXiFar * 1.02) * kmax;
  return __builtin_amdgcn_readfirstlane((int)ok) != 0;
}
template <int NI>
__device__ __forceinline__ double point_forward(double ws, const Base& b, const Base& bn, bool has_next,
                                                const LineS<NI>& L, const Tables& T) {
  return point_forward_sd<NI>(b, bn, has_next, L, T) * L.pref * ws * ws;   // :291-294
}

struct BaseAdj {  // adjoints flowing into base quantities of a point
  double k2, ik, wd, xe, F;
};

// reverse of point_forward: given the seed, produce adjoints of this point's base quantities (ba),
// of the right neighbour's (xe, F) (xen, Fn), and accumulate lineout-scalar adjoints into LB.
// GM: 0 plasma parameters only, 1 + DLM order m (tangent tables), 2 + the distribution-function tables themselves
// PQ = Pbar pref ws^2 (the seed times the factor point_forward_sd leaves out)
template <int NI, int GM = 0, bool ZH = true, bool FAR = false>
__device__ __forceinline__ void point_reverse(const Base& b, const Base& bn, bool has_next,
                                              const LineS<NI>& L, const Tables& T, double PQ,
                                              BaseAdj& ba, double& xen, double& Fn, LineS<NI>& LB, FeAcc& fa) {
  // ---- recompute forward ----
  PointF<NI> pf;
  point_core<NI, ZH, FAR>(b, bn, has_next, L, T, pf);
  const double ik2 = pf.ik2, ike2 = pf.ike2, vph = pf.vph, gsum = pf.gsum, Wl = pf.Wl, dW = pf.dW, idx = pf.idx, D = pf.D;
  const double cer = pf.cer, cei = pf.cei, opc = pf.opc, cim = pf.cim, er = pf.er, ei = pf.ei, ieps2 = pf.ieps2, ce2 = pf.ce2, ci2 = pf.ci2;
  const double N = pf.N, t1 = pf.t1, S = pf.S, dop = pf.dop;
  const double* xi = pf.xi; const double* zr = pf.zr; const double* zi = pf.zi; const double* dzr = pf.dzr; const double* dzi = pf.dzi;
  const double* hk = pf.hk; const double* gs = pf.gs;
  // ---- reverse ----
  // Accumulators with a DEFERRED wavefront-uniform factor (applied once per gradient point by lines_adjoint_finalize,
  // instead of once per point): LB.pref holds sum Sb S (x 1/pref), LB.i2wL holds sum PSQ wd (-> LB.wL, x -i2wL^2/2),
  // LB.a_i[s] holds sum (creb zr + cimb zi) ik2 (x -1/2).
  const double Sb = PQ * dop;
  const double PSQ = PQ * S;
  LB.i2wL += PSQ * b.wd;
  LB.pref += Sb * S;
  const double Nb = Sb * t1;
  const double ikb0 = Sb * N * ieps2;          // first part of the adjoint of 1/k
  const double e2 = -2.0 * (ikb0 * t1);        // 2 x adjoint of |eps|^2
  const double NbI = Nb * L.ivTe;
  const double ci2b2 = 2.0 * (NbI * b.F);       // 2 x adjoint of |1 + chi_i|^2
  ba.F = NbI * ci2;
  LB.ivTe += (Nb * ci2) * b.F;
  const double erb = e2 * er, eib = e2 * ei;
  double cerb = erb, ceib = eib, cimb = eib;
  if (!FAR) {   // (FAR: the ion numerator gsum and Im chi_i vanish identically, see point_core)
    const double ce2b2 = 2.0 * (Nb * gsum);     // 2 x adjoint of |chi_e|^2
    cerb = erb + ce2b2 * cer; ceib = eib + ce2b2 * cei;
    cimb = eib + ci2b2 * cim;
  }
  const double creb = erb + ci2b2 * opc;
  const double cp = ceib * kPi;
  const double ike2b = cp * D - cerb * Wl;
  const double Wlb = -cerb * ike2;  // adjoint of the interpolated W
  ba.xe = Wlb * dW;
  if (GM == 1) {  // W depends on the DLM order through the table itself
    double Wml, dWm;
    w_lookup(T.Wm, b.xe, Wml, dWm);
    LB.m += Wlb * Wml;
  }
  if (GM == 2) fe_add_w(fa, T.Wb, b.xe, Wlb);
  // D = (Fn - F) * idx
  Fn = (cp * ike2) * idx;
  ba.F -= Fn;
  xen = -Fn * D;
  ba.xe -= xen;
  double k2acc = 0.0, vphb = 0.0;
  const double u = FAR ? 0.0 : Nb * ce2;
#pragma unroll
  for (int s = 0; s < NI; ++s) {
    double w, xib;
    if (FAR) {   // gs = Im Z' = d Im Z' = 0
      w = creb * zr[s];
      xib = hk[s] * (creb * dzr[s]);
    } else {
      const double v = u * gs[s];
      LB.cs[s] += v;
      w = creb * zr[s] + cimb * zi[s];
      xib = -2.0 * (v * xi[s] * L.cs[s]) + hk[s] * (creb * dzr[s] + cimb * dzi[s]);
    }
    const double wk = w * ik2;
    LB.a_i[s] += wk;
    k2acc += wk * hk[s];
    vphb += xib * L.ixi[s];
    LB.ixi[s] += xib * vph;
  }
  const double we = ike2b * ik2;
  LB.a_e += we;
  ba.k2 = -k2acc - we * ike2;
  // v_ph = wd * ik
  ba.wd = PSQ * L.i2wL + vphb * b.ik;
  ba.ik = ikb0 + vphb * b.wd;
}

// the deferred factors of point_reverse (see there), applied to a thread's (or a reduced) accumulator set
template <int NI>
__device__ __forceinline__ void lines_adjoint_finalize(const LineS<NI>& L, LineS<NI>& LB) {
  LB.pref *= 1.0 / L.pref;
  LB.wL -= LB.i2wL * (0.5 * L.i2wL * L.i2wL);
  LB.i2wL = 0.0;
#pragma unroll
  for (int s = 0; s < NI; ++s) LB.a_i[s] *= -0.5;
}

// reverse of base_eval
template <int NI, int GM = 0>
__device__ __forceinline__ void base_reverse(double ct, const Base& b, const LineS<NI>& L, const Tables& T,
                                             const BaseAdj& ba, LineS<NI>& LB, FeAcc& fa) {
  const double Hb = ba.F * b.F;  // adjoint of H = ln f_e(xi_e)
  if (GM == 2) fe_add_h(fa, T, b.xe, Hb);
  if (GM == 1) {  // d ln f_e(xi_e)/dm: Hermite interpolant of the tangent table (zero outside the vx grid)
    Tables Tm = T;
    Tm.hc = T.hcm;
    double Hm, dHm;
    hermite_lookup_c(Tm, b.xe, Hm, dHm);
    LB.m += (b.xe < T.vx0 || b.xe > T.vxlast) ? 0.0 : Hb * Hm;
  }
  const double xeb = ba.xe + Hb * b.dH;
  const double vph = b.wd * b.ik, k = b.k2 * b.ik;
  // xe = (vph - Ud) * ivTe
  const double vphb = xeb * L.ivTe;
  LB.Ud -= vphb;
  LB.ivTe += xeb * (vph - L.Ud);
  // vph = wd * ik
  const double wdb = ba.wd + vphb * b.ik;
  const double ikb = ba.ik + vphb * b.wd;
  // wd = ws - wL - k Vd
  LB.wL -= wdb;
  LB.Vd -= wdb * k;
  // k = sqrt(k2), ik = 1/k:  kb = -wdb Vd - ikb ik^2,  k2b = ba.k2 + kb ik / 2
  const double kb = wdb * L.Vd + ikb * (b.ik * b.ik);
  const double k22 = 2.0 * ba.k2 - kb * b.ik;     // 2 x adjoint of k^2
  const double ksb = k22 * (b.ks - L.kL * ct);
  LB.kL += k22 * (L.kL - b.ks * ct);
  LB.wpe2 -= ksb * (0.5 / (kC * kC)) * frcp(b.ks);
}

// physical parameters staged once per workgroup in LDS: ph[slot] (after Ti tying and fraction
// renormalisation), ph[NP] = sum of the un-normalised fractions
template <int NI>
__device__ __forceinline__ void phys_from_lds(const double* ph, Phys<NI>& p) {
  p.Te = ph[TSFF_P_TE]; p.ne = ph[TSFF_P_NE]; p.m = ph[TSFF_P_M]; p.lam = ph[TSFF_P_LAM];
  p.amp1 = ph[TSFF_P_AMP1]; p.amp2 = ph[TSFF_P_AMP2]; p.amp3 = ph[TSFF_P_AMP3];
  p.neg = ph[TSFF_P_NE_GRADIENT]; p.teg = ph[TSFF_P_TE_GRADIENT]; p.ud = ph[TSFF_P_UD]; p.Va = ph[TSFF_P_VA];
#pragma unroll
  for (int s = 0; s < NI; ++s) {
    const int o = TSFF_P_ION0 + 4 * s;
    p.Ti[s] = ph[o + TSFF_ION_TI]; p.Z[s] = ph[o + TSFF_ION_Z]; p.A[s] = ph[o + TSFF_ION_A]; p.fr[s] = ph[o + TSFF_ION_FRACT];
  }
  p.fsum = ph[TSFF_NP(NI)];
}

template <int NI>
__device__ __forceinline__ void zero_lines(LineS<NI>& L) {
  L.wpe2 = L.wL = L.kL = L.ivTe = L.a_e = L.pref = L.Ud = L.Vd = L.i2wL = L.m = 0.0;
#pragma unroll
  for (int s = 0; s < NI; ++s) L.ixi[s] = L.a_i[s] = L.cs[s] = L.hai[s] = 0.0;
}

// ------------------------------------------------------------------------------------------
// parameter transform  (ts_params.py:329-350, 543-603)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double sigmoid(double x) { return 1.0 / (1.0 + exp(-x)); }

// sg (optional): sigmoid(x[s]) of the slots that have one, already evaluated by the caller (k_fused_finish needs them again
// for the derivative of the activation)
template <int NI>
__device__ __forceinline__ void load_phys(const double* __restrict__ x, const double* __restrict__ scale,
                                          const double* __restrict__ shift, const uint8_t* __restrict__ sig,
                                          const uint8_t* ti_same, bool activate, Phys<NI>& p, const double* sg = nullptr) {
  auto tr = [&](int s) {
    const double v = x[s];
    if (!activate) return v;
    return (sig[s] ? (sg ? sg[s] : sigmoid(v)) : v) * scale[s] + shift[s];
  };
  p.Te = tr(TSFF_P_TE); p.ne = tr(TSFF_P_NE); p.m = tr(TSFF_P_M); p.lam = tr(TSFF_P_LAM);
  p.amp1 = tr(TSFF_P_AMP1); p.amp2 = tr(TSFF_P_AMP2); p.amp3 = tr(TSFF_P_AMP3);
  p.neg = tr(TSFF_P_NE_GRADIENT); p.teg = tr(TSFF_P_TE_GRADIENT);
  p.ud = tr(TSFF_P_UD); p.Va = tr(TSFF_P_VA);
  double fsum = 0.0;
#pragma unroll
  for (int s = 0; s < NI; ++s) {
    const int o = TSFF_P_ION0 + 4 * s;
    p.Ti[s] = tr(o + TSFF_ION_TI); p.Z[s] = tr(o + TSFF_ION_Z);
    p.A[s] = tr(o + TSFF_ION_A); p.fr[s] = tr(o + TSFF_ION_FRACT);
    if (s > 0 && ti_same[s]) p.Ti[s] = p.Ti[0];
    fsum += p.fr[s];
  }
  p.fsum = fsum;
  if (activate) {
#pragma unroll
    for (int s = 0; s < NI; ++s) p.fr[s] /= fsum;
  }
}

}  // namespace tsff

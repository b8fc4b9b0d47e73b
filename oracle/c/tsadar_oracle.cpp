// TEST INFRASTRUCTURE ONLY -- not part of the product.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
// bench.py may load this library; tsadar_amd/ never does.
//
// C++ restatement of the reference's 1-D hot path (ergodicio/tsadar @ 2025-08-24) for a whole batch of lineouts on the
// host cores (OpenMP over lineouts): parameter transform, FormFactor.__call__, FitModel, instrument response, masked
// loss -- and its gradient by FORWARD-mode dual numbers (one tangent per trainable leaf).  It shares no code with the HIP
// path (whose gradient is a hand-written reverse sweep) nor with the NumPy / torch oracle (vectorised restatement,
// reverse-mode autograd); it is pinned by tests/test_oracle_c.py to the NumPy oracle, which is itself pinned to the
// reference's golden vector tests/test_forward/ThryE-1d.npy.
//
// Every block cites the reference lines it follows (paths relative to /root/reference/tsadar).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---- constants (core/physics/form_factor.py:123-125, 207-209) ----
constexpr double kC = 2.99792458e10;
constexpr double kPi = 3.14159265358979323846;
const double kMe = 510.9896 / (kC * kC);
const double kMp = kMe * 1836.1;
constexpr double kRe = 2.8179e-13;
const double kEsq = kMe * kC * kC * kRe;
const double kC0 = std::sqrt(4 * kPi * kEsq / kMe);
constexpr int kNXi1 = 1024, kNXi2 = 1640;
constexpr double kXiMax = 8.2, kXiH = 0.01;

// ---- forward-mode dual number with N tangents ----
template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  Dual(double x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
};
template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; const double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double a, const Dual<N>& b) { return b + a; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double a, const Dual<N>& b) { return -b + a; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> inline Dual<N> operator*(double a, const Dual<N>& b) { return b * a; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N> sqrt(const Dual<N>& a) { Dual<N> r; r.v = std::sqrt(a.v); const double s = 0.5 / r.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> inline Dual<N> exp(const Dual<N>& a) { Dual<N> r; r.v = std::exp(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v; return r; }
template <int N> inline Dual<N> log(const Dual<N>& a) { Dual<N> r; r.v = std::log(a.v); const double s = 1.0 / a.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
template <int N> inline Dual<N> fabs(const Dual<N>& a) { return a.v < 0.0 ? -a : a; }
template <int N> inline Dual<N> logcosh(const Dual<N>& a) { Dual<N> r; r.v = std::log(std::cosh(a.v)); const double s = std::tanh(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r; }
inline double logcosh(double a) { return std::log(std::cosh(a)); }
template <int N> inline double val(const Dual<N>& a) { return a.v; }
inline double val(double a) { return a; }
using std::exp; using std::fabs; using std::log; using std::sqrt;

}  // namespace

extern "C" {

// parameter slots: the order of include/tsff.h (shared with the test helpers only for convenience)
enum { P_TE = 0, P_NE, P_M, P_LAM, P_AMP1, P_AMP2, P_AMP3, P_NEG, P_TEG, P_UD, P_VA, P_ION0 };
enum { ION_TI = 0, ION_Z, ION_A, ION_FRACT };

struct orc_deck {
  double lamrangE[2], lamrangI[2];
  int32_t npts, load_ele, load_ion;
  double ele_lam_shift;
  int32_t n_angles;
  const double* sa_deg;
  const double* sa_w;
  int32_t G, n_ion, nvx;
  const double* fe;   // [nvx] distribution function shared by all lineouts
  const double* zr;   // [1640] Re Z' on xi2 (form_factor.py:33-44,139)
  const double* zi;   // [1640] Im Z'
  double sigma_e, sigma_i;          // PhysParams.widIRF.spect_stddev_ele / _ion
  int32_t filt_on;
  double filt_od, filt_w, filt_c;   // iawfilter = [on, OD, width, centre]
  const double* scale;              // [NP]
  const double* shift;              // [NP]
  const uint8_t* sig;               // [NP] 1: sigmoid-activated leaf
  const uint8_t* ti_same;           // [n_ion]
  int32_t loss_method;              // 0 l2, 1 l1, 2 log-cosh, 3 poisson
  int32_t fit_iaw, fit_blue, fit_red;
  double blue_min, blue_max, red_min, red_max, iaw_min, iaw_cf_min, iaw_cf_max, iaw_max;
};

}  // extern "C"

namespace {

struct Static {  // per-deck tables
  std::vector<double> xi1, xi2, vx, lnfe, slope, W;
  double dv;
};

// interpax.interp1d(x, vx, ln fe, method="cubic", extrap=[-50, -50]) (form_factor.py:256, 263): C1 cubic Hermite with
// node slopes = mean of the adjacent secants (one-sided at the ends), constant outside the grid
template <class T>
T hermite(const Static& S, const T& x) {
  const int n = (int)S.vx.size();
  const double xv = val(x);
  if (xv < S.vx[0] || xv > S.vx[n - 1]) return T(-50.0);
  int i = (int)(std::upper_bound(S.vx.begin(), S.vx.end(), xv) - S.vx.begin());
  i = std::min(std::max(i, 1), n - 1);
  const double dx = S.vx[i] - S.vx[i - 1];
  const T t = (x - S.vx[i - 1]) / dx;
  const double f0 = S.lnfe[i - 1], f1 = S.lnfe[i], m0 = S.slope[i - 1] * dx, m1 = S.slope[i] * dx;
  const double c2 = -3 * f0 + 3 * f1 - 2 * m0 - m1, c3 = 2 * f0 - 2 * f1 + m0 + m1;
  return f0 + t * (m0 + t * (c2 + t * c3));
}

// jnp.interp(x, xi2, tab, left, right) (form_factor.py:247-248, 270)
template <class T>
T interp_xi2(const Static& S, const double* tab, const T& x, bool& below, bool& above) {
  const double xv = val(x);
  below = xv < S.xi2[0];
  above = xv > S.xi2[kNXi2 - 1];
  int i = (int)(std::upper_bound(S.xi2.begin(), S.xi2.end(), xv) - S.xi2.begin());
  i = std::min(std::max(i, 1), kNXi2 - 1);
  const double dx = S.xi2[i] - S.xi2[i - 1];
  return tab[i - 1] + ((x - S.xi2[i - 1]) / dx) * (tab[i] - tab[i - 1]);
}

void build_static(const orc_deck& D, Static& S) {
  // form_factor.py:137-138
  S.xi1.resize(kNXi1);
  const double a = kXiMax + std::sqrt(2.0) / 1024.0;
  for (int i = 0; i < kNXi1; ++i) S.xi1[i] = i == kNXi1 - 1 ? a : -a + i * ((2 * a) / (kNXi1 - 1));  // np.linspace
  S.xi2.resize(kNXi2);
  for (int i = 0; i < kNXi2; ++i) S.xi2[i] = -kXiMax + kXiH * i;  // jnp.arange(-8.2, 8.2, 0.01)
  // base.py:149-151
  const int n = D.nvx;
  S.dv = 12.0 / n;
  S.vx.resize(n);
  {
    const double v0 = -6.0 + S.dv / 2, v1 = 6.0 - S.dv / 2, step = (v1 - v0) / (n - 1);
    for (int i = 0; i < n; ++i) S.vx[i] = i == n - 1 ? v1 : v0 + i * step;
  }
  S.lnfe.resize(n);
  S.slope.resize(n);
  for (int i = 0; i < n; ++i) S.lnfe[i] = std::log(D.fe[i]);
  for (int i = 0; i < n; ++i) {
    const double dl = i > 0 ? (S.lnfe[i] - S.lnfe[i - 1]) / (S.vx[i] - S.vx[i - 1]) : 0.0;
    const double dr = i < n - 1 ? (S.lnfe[i + 1] - S.lnfe[i]) / (S.vx[i + 1] - S.vx[i]) : 0.0;
    S.slope[i] = i == 0 ? dr : (i == n - 1 ? dl : 0.5 * (dl + dr));
  }
  // Re(chi_e) table (form_factor.py:263-268; ratintn.py:4-52)
  std::vector<double> ratmod(kNXi1), ratdf(kNXi1);
  for (int i = 0; i < kNXi1; ++i) ratmod[i] = std::exp(hermite<double>(S, S.xi1[i]));
  const double h = S.xi1[1] - S.xi1[0];
  for (int i = 0; i < kNXi1; ++i) {  // jnp.gradient
    if (i == 0) ratdf[i] = (ratmod[1] - ratmod[0]) / h;
    else if (i == kNXi1 - 1) ratdf[i] = (ratmod[i] - ratmod[i - 1]) / h;
    else ratdf[i] = (ratmod[i + 1] - ratmod[i - 1]) / (2 * h);
  }
  S.W.assign(kNXi2, 0.0);
  for (int q = 0; q < kNXi2; ++q) {
    double acc = 0.0;
    for (int i = 0; i < kNXi1 - 2; ++i) {  // ratintn.py:21, 41-44: N - 2 intervals
      const double g0 = S.xi1[i] - S.xi2[q], g1 = S.xi1[i + 1] - S.xi2[q];
      const double fdif = ratdf[i + 1] - ratdf[i], gdif = g1 - g0, fav = 0.5 * (ratdf[i + 1] + ratdf[i]), gav = 0.5 * (g1 + g0);
      const double tmp = fav * gdif - gav * fdif;
      double r;
      if (std::fabs(gdif) < 1.0e-4 * std::fabs(gav)) r = fav / gav + tmp * gdif / (12.0 * gav * gav * gav);
      else r = fdif / gdif + tmp * std::log(std::fabs((gav + 0.5 * gdif) / (gav - 0.5 * gdif))) / (gdif * gdif);
      acc += r * (S.xi1[i + 1] - S.xi1[i]);
    }
    S.W[q] = acc;
  }
}

template <class T>
struct Phys {
  T Te, ne, lam, amp1, amp2, amp3, neg, teg, ud, Va;
  T Ti[4], Z[4], fr[4];
  double A[4];
};

// ThomsonParams.__call__ (core/modules/ts_params.py:329-350, 543-603)
template <class T>
void physical(const orc_deck& D, const T* x, Phys<T>& p) {
  auto act = [&](int s) -> T {
    T u = x[s];
    if (D.sig[s]) u = 1.0 / (1.0 + exp(-u));
    return u * D.scale[s] + D.shift[s];
  };
  p.Te = act(P_TE); p.ne = act(P_NE); p.lam = act(P_LAM); p.amp1 = act(P_AMP1); p.amp2 = act(P_AMP2); p.amp3 = act(P_AMP3);
  p.neg = act(P_NEG); p.teg = act(P_TEG); p.ud = act(P_UD); p.Va = act(P_VA);
  T fsum = 0.0;
  for (int s = 0; s < D.n_ion; ++s) {
    const int o = P_ION0 + 4 * s;
    p.Ti[s] = (s > 0 && D.ti_same[s]) ? p.Ti[0] : act(o + ION_TI);
    p.Z[s] = act(o + ION_Z);
    p.A[s] = val(x[o + ION_A]);
    p.fr[s] = act(o + ION_FRACT);
    fsum = fsum + p.fr[s];
  }
  for (int s = 0; s < D.n_ion; ++s) p.fr[s] = p.fr[s] / fsum;
}

// FormFactor.__call__ + FitModel (form_factor.py:163-298, generate_spectra.py:139-220): modl[npts], lam_nm[npts]
template <class T>
void model(const orc_deck& D, const Static& S, const Phys<T>& p, int feature, std::vector<T>& modl, std::vector<double>& lam_nm) {
  const int npts = D.npts, NA = D.n_angles, G = D.G, NI = D.n_ion;
  const double* rng = feature ? D.lamrangI : D.lamrangE;
  const double lam_shift = feature ? 0.0 : D.ele_lam_shift;
  std::vector<double> omgs(npts);
  lam_nm.resize(npts);
  for (int j = 0; j < npts; ++j) {
    const double lj = j == npts - 1 ? rng[1] : rng[0] + j * ((rng[1] - rng[0]) / (npts - 1));  // np.linspace (:132)
    omgs[j] = 2e7 * kPi * kC / lj;                   // :134-135
    lam_nm[j] = (2 * kPi * kC / omgs[j]) * 1e7;      // :293, generate_spectra.py:163,191
  }
  modl.assign(npts, T(0.0));
  std::vector<T> xie(npts), F(npts), kk(npts), od(npts);
  const T omgL = (2 * kPi * 1e7 * kC) / (p.lam + lam_shift);  // :212
  T Zbar = 0.0;
  for (int s = 0; s < NI; ++s) Zbar = Zbar + p.Z[s] * p.fr[s];
  for (int g = 0; g < G; ++g) {
    const double cg = G == 1 ? -1.0 / 200.0 : -1.0 / 200.0 + (double)g / (100.0 * (G - 1));  // linspace(1 - v/200, 1 + v/200, G)
    const T ne = 1.0e20 * p.ne * (1.0 + p.neg * cg);                                          // :182-195
    const T Te = p.Te * (1.0 + p.teg * cg);
    const T omgpe = kC0 * sqrt(ne);
    const T kL = sqrt(omgL * omgL - omgpe * omgpe) / kC;
    const T vTe = sqrt(Te / kMe);
    T vTi[4], omgpi[4];
    for (int s = 0; s < NI; ++s) {
      const double Mi = p.A[s] * kMp;
      const T ni = p.fr[s] * ne / Zbar;
      omgpi[s] = kC0 * p.Z[s] * sqrt(ni * kMe / Mi);
      vTi[s] = sqrt(p.Ti[s] / Mi);
    }
    for (int a = 0; a < NA; ++a) {
      const double ct = std::cos(D.sa_deg[a] * kPi / 180.0);
      for (int j = 0; j < npts; ++j) {
        const T ks = sqrt(omgs[j] * omgs[j] - omgpe * omgpe) / kC;             // :218
        const T k = sqrt(ks * ks + kL * kL - 2.0 * ks * kL * ct);               // :220
        const T omgdop = (omgs[j] - omgL) - k * (p.Va * 1e6);                   // :216, 222-223
        kk[j] = k;
        od[j] = omgdop;
        xie[j] = omgdop / (k * vTe) - (p.ud * 1e6) / vTe;                       // :253
        F[j] = exp(hermite<T>(S, xie[j]));                                      // :256
      }
      for (int j = 0; j < npts; ++j) {
        const T k = kk[j], omgdop = od[j];
        const T klde = (vTe / omgpe) * k;                                       // :227-228
        const T iklde2 = 1.0 / (klde * klde);
        T cre = 0.0, cim = 0.0, gsum = 0.0;
        for (int s = 0; s < NI; ++s) {
          const T kldi = (vTi[s] / omgpi[s]) * k;                               // :239
          const T xii = (1.0 / (std::sqrt(2.0) * vTi[s])) * (omgdop / k);       // :243
          bool lo, hi;
          T zr = interp_xi2<T>(S, D.zr, xii, lo, hi);
          T zi = interp_xi2<T>(S, D.zi, xii, lo, hi);
          if (lo || hi) { zr = 1.0 / (xii * xii); zi = T(0.0); }                // :247-248
          const T c = -0.5 / (kldi * kldi);                                     // :249
          cre = cre + c * zr;
          cim = cim + c * zi;
          gsum = gsum + (p.fr[s] * p.Z[s] * p.Z[s] / Zbar / vTi[s]) * exp(-(xii * xii)) / std::sqrt(2 * kPi);  // :277-280
        }
        const T df = j < npts - 1 ? (F[j + 1] - F[j]) / (xie[j + 1] - xie[j]) : T(0.0);  // :258-259
        const T cei = kPi * iklde2 * df;                                        // :261
        bool lo, hi;
        T Wl = interp_xi2<T>(S, S.W.data(), xie[j], lo, hi);                    // :270 (clamps to the end values)
        if (lo) Wl = T(S.W[0]);
        if (hi) Wl = T(S.W[kNXi2 - 1]);
        const T cer = -1.0 * iklde2 * Wl;                                       // :271
        const T er = 1.0 + cer + cre, ei = cei + cim;                           // :274
        const T eps2 = er * er + ei * ei;
        const T ce2 = cer * cer + cei * cei;
        const T ci2 = (1.0 + cre) * (1.0 + cre) + cim * cim;
        const T Sv = (gsum * ce2 + ci2 * F[j] / vTe) / (k * eps2);              // :282-288
        const T PsOmg = Sv * (1.0 + 2.0 * omgdop / omgL) * (kRe * kRe) * ne;    // :291
        const double lam_cm = 2 * kPi * kC / omgs[j];
        const T PsLam = PsOmg * (2 * kPi * kC / (lam_cm * lam_cm));             // :294
        modl[j] = modl[j] + PsLam * (D.sa_w[a] / (double)G);                    // generate_spectra.py:164-165, 193-197
      }
    }
  }
  if (feature == 0 && D.filt_on) {                                              // generate_spectra.py:210-216
    const double fb = D.filt_c - D.filt_w / 2, fr = D.filt_c + D.filt_w / 2;
    if (D.lamrangE[0] < fr && D.lamrangE[1] > fb) {
      const double m = std::pow(10.0, -D.filt_od);
      for (int j = 0; j < npts; ++j)
        if (fb < lam_nm[j] && fr > lam_nm[j]) modl[j] = modl[j] * m;
    }
  }
}

template <class T>
int argmax(const std::vector<T>& v) {
  int k = 0;
  for (int i = 1; i < (int)v.size(); ++i)
    if (val(v[i]) > val(v[k])) k = i;
  return k;
}

// add_electron_IRF / add_ion_IRF (core/physics/irf.py:50-132, norm == 0): Gaussian on the full axis, "same" convolution,
// rescale to the unconvolved maximum, points_per_pixel bin average
template <class T>
void irf(const std::vector<T>& x, const std::vector<double>& lam, double sigma, int nbins, std::vector<T>& yb, std::vector<double>& lamb) {
  const int n = (int)x.size(), ppp = n / nbins;
  const double lo = *std::min_element(lam.begin(), lam.end()), hi = *std::max_element(lam.begin(), lam.end());
  const double origin = (hi + lo) / 2.0;
  std::vector<double> g(n);
  for (int i = 0; i < n; ++i) g[i] = (1.0 / (sigma * std::sqrt(2.0 * kPi))) * std::exp(-((lam[i] - origin) * (lam[i] - origin)) / (2.0 * sigma * sigma));
  int g0 = 0, g1 = n - 1;
  while (g0 < n && g[g0] == 0.0) ++g0;          // (taps that are exactly zero contribute exactly nothing)
  while (g1 >= 0 && g[g1] == 0.0) --g1;
  const int c = (n - 1) / 2;                    // np.convolve(x, g, "same"): y[j] = sum_k x[k] g[j + c - k]
  std::vector<T> y(n);
  for (int j = 0; j < n; ++j) {
    T acc = 0.0;
    const int k0 = std::max(0, j + c - g1), k1 = std::min(n - 1, j + c - g0);
    for (int k = k0; k <= k1; ++k) acc = acc + x[k] * g[j + c - k];
    y[j] = acc;
  }
  const T sc = x[argmax(x)] / y[argmax(y)];     // irf.py:73, 115
  yb.assign(nbins, T(0.0));
  lamb.assign(nbins, 0.0);
  for (int p = 0; p < nbins; ++p) {
    T acc = 0.0;
    double la = 0.0;
    for (int q = 0; q < ppp; ++q) { acc = acc + y[p * ppp + q] * sc; la += lam[p * ppp + q]; }
    yb[p] = acc / (double)ppp;
    lamb[p] = la / ppp;
  }
}

template <class T>
T functional(int method, double d, const T& t) {  // inverse/loss_function.py:386-418 (1/uncert of l1, l2 lives in the weights)
  const T r = d - t;
  if (method == 0) return r * r;
  if (method == 1) return fabs(r);
  if (method == 2) return logcosh(r);
  return t - d * log(t);
}

// one lineout: spectra, masked sums; T carries the tangents
template <class T>
void lineout(const orc_deck& D, const Static& S, const T* x, const double* e_data, const double* i_data, double e_amp, double i_amp,
             const double* noise_e, const double* noise_i, T sums[3], double* ThryE, double* ThryI) {
  Phys<T> p;
  physical<T>(D, x, p);
  sums[0] = sums[1] = sums[2] = T(0.0);
  std::vector<T> modl, yb;
  std::vector<double> lam, lamb;
  const int NB = 1024;
  if (D.load_ion) {
    model<T>(D, S, p, 1, modl, lam);
    irf<T>(modl, lam, D.sigma_i, NB, yb, lamb);
    const T mx = yb[argmax(yb)];
    for (int b = 0; b < NB; ++b) {
      const T t = p.amp3 * i_amp * yb[b] / mx + (noise_i ? noise_i[b] : 0.0);   // irf.py:79-81, thomson_diagnostic.py:140
      if (ThryI) ThryI[b] = val(t);
      const bool m = (lamb[b] > D.iaw_min && lamb[b] < D.iaw_cf_min) || (lamb[b] > D.iaw_cf_max && lamb[b] < D.iaw_max);
      if (D.fit_iaw && m && i_data) sums[0] = sums[0] + functional<T>(D.loss_method, i_data[b], t);   // loss_function.py:224-237
    }
  }
  if (D.load_ele) {
    model<T>(D, S, p, 0, modl, lam);
    irf<T>(modl, lam, D.sigma_e, NB, yb, lamb);
    const T mx = yb[argmax(yb)];
    for (int b = 0; b < NB; ++b) {
      const T amp = lamb[b] < val(p.lam) ? p.amp1 : p.amp2;                     // irf.py:126-130
      const T t = e_amp * yb[b] / mx * amp + (noise_e ? noise_e[b] : 0.0);
      if (ThryE) ThryE[b] = val(t);
      if (e_data && D.fit_blue && lamb[b] > D.blue_min && lamb[b] < D.blue_max) sums[1] = sums[1] + functional<T>(D.loss_method, e_data[b], t);
      if (e_data && D.fit_red && lamb[b] > D.red_min && lamb[b] < D.red_max) sums[2] = sums[2] + functional<T>(D.loss_method, e_data[b], t);
    }
  }
}

template <int N>
void run(const orc_deck& D, const Static& S, const double* X, const double* e_data, const double* i_data, const double* e_amps,
         const double* i_amps, const double* noise_e, const double* noise_i, int B, const double* w, const uint8_t* gmask,
         int nthreads, double* sums, double* grad, double* ThryE, double* ThryI) {
  const int NP = P_ION0 + 4 * D.n_ion;
  std::vector<int> act;
  for (int s = 0; s < NP; ++s)
    if (gmask && gmask[s]) act.push_back(s);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : omp_get_max_threads())
#endif
  for (int b = 0; b < B; ++b) {
    const double* xb = X + (size_t)b * NP;
    const double* ed = e_data ? e_data + (size_t)b * 1024 : nullptr;
    const double* id = i_data ? i_data + (size_t)b * 1024 : nullptr;
    const double* ne_ = noise_e ? noise_e + (size_t)b * 1024 : nullptr;
    const double* ni_ = noise_i ? noise_i + (size_t)b * 1024 : nullptr;
    double* te = ThryE ? ThryE + (size_t)b * 1024 : nullptr;
    double* ti = ThryI ? ThryI + (size_t)b * 1024 : nullptr;
    if (N == 0) {
      std::vector<double> x(xb, xb + NP);
      double s3[3];
      lineout<double>(D, S, x.data(), ed, id, e_amps ? e_amps[b] : 0.0, i_amps ? i_amps[b] : 0.0, ne_, ni_, s3, te, ti);
      for (int k = 0; k < 3; ++k) sums[(size_t)b * 3 + k] = s3[k];
    } else {
      constexpr int M = N > 0 ? N : 1;
      std::vector<Dual<M>> x(NP);
      for (int s = 0; s < NP; ++s) x[s] = Dual<M>(xb[s]);
      for (int a = 0; a < (int)act.size(); ++a) x[act[a]].d[a] = 1.0;
      Dual<M> s3[3];
      lineout<Dual<M>>(D, S, x.data(), ed, id, e_amps ? e_amps[b] : 0.0, i_amps ? i_amps[b] : 0.0, ne_, ni_, s3, te, ti);
      for (int k = 0; k < 3; ++k) sums[(size_t)b * 3 + k] = s3[k].v;
      for (int s = 0; s < NP; ++s) grad[(size_t)b * NP + s] = 0.0;
      for (int a = 0; a < (int)act.size(); ++a)
        grad[(size_t)b * NP + act[a]] = w[0] * s3[0].d[a] + w[1] * s3[1].d[a] + w[2] * s3[2].d[a];
    }
  }
}

}  // namespace

extern "C" {

// Re(chi_e) table of one distribution function (form_factor.py:263-268) -- for the table test
int orc_chi_table(const orc_deck* D, double* W) {
  Static S;
  build_static(*D, S);
  std::memcpy(W, S.W.data(), kNXi2 * sizeof(double));
  return 0;
}

// X [B][NP] normalised leaves; data/noise [B][1024] (noise may be NULL); w[3] weights of the three masked sums
// (Engine.loss_weights semantics: 1/uncert, 1/N and the 1/2 of blue+red folded in); gmask [NP] trainable leaves (NULL:
// forward only).  Out: sums [B][3] un-weighted masked sums; grad [B][NP] = d(w . sums_b)/dX_b; ThryE/ThryI [B][1024] (may be NULL).
int orc_loss_grad(const orc_deck* D, const double* X, const double* e_data, const double* i_data, const double* e_amps,
                  const double* i_amps, const double* noise_e, const double* noise_i, int32_t B, const double* w,
                  const uint8_t* gmask, int32_t nthreads, double* sums, double* grad, double* ThryE, double* ThryI) {
  if (!D || !X || !sums || B < 1 || D->n_ion < 1 || D->n_ion > 4 || D->npts % 1024) return -1;
  Static S;
  build_static(*D, S);
  const int NP = P_ION0 + 4 * D->n_ion;
  int nact = 0;
  if (gmask && grad)
    for (int s = 0; s < NP; ++s) nact += gmask[s] ? 1 : 0;
  if (nact == 0) run<0>(*D, S, X, e_data, i_data, e_amps, i_amps, noise_e, noise_i, B, w, nullptr, nthreads, sums, grad, ThryE, ThryI);
  else if (nact <= 6) run<6>(*D, S, X, e_data, i_data, e_amps, i_amps, noise_e, noise_i, B, w, gmask, nthreads, sums, grad, ThryE, ThryI);
  else if (nact <= 12) run<12>(*D, S, X, e_data, i_data, e_amps, i_amps, noise_e, noise_i, B, w, gmask, nthreads, sums, grad, ThryE, ThryI);
  else run<27>(*D, S, X, e_data, i_data, e_amps, i_amps, noise_e, noise_i, B, w, gmask, nthreads, sums, grad, ThryE, ThryI);
  return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"

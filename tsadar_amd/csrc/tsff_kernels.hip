// tsff_kernels.hip -- HIP kernels of the Thomson-scattering form-factor engine (gfx950 only).
//
// Kernel map (reference rows are those of SURVEY.md section 8a; DESIGN.md section 4 has the details)
//   k_fe_prepare   a2, a6(table part), a8 : f_e -> ln f_e Hermite table, ratmod, ratdf, W[1640] (ratintn)
//   k_spectrum     a1, a4-a15            : one workgroup per lineout (256 threads per feature); forward sweep
//                                          over (lambda x theta), IRF convolution, binning, normalisation,
//                                          loss partials and -- MODE 1 -- the hand-written adjoint incl. the
//                                          chain rule to the normalised leaves
//   k_loss_reduce  a14                   : deterministic reduction of the masked loss sums
//   k_form_factor  a4-a10                : raw FormFactor.__call__ output (known-answer tests)
#include "tsff_device.h"

namespace tsff {

// static, per-handle device configuration (passed by value)
struct KStatic {
  int npts, ppp, n_angles, G, nvx, NP;
  int shared_fe;  // 1: every lineout uses table slot 0
  int loss_method;
  int load[2];
  double lam_shift[2];
  const double* omgs[2];     // [npts] scattered-frequency axis (form_factor.py:134)
  const double* lam_bin[2];  // [1024] binned wavelength axis in nm (irf.py:75,125)
  const double* filt;        // [npts] EPW multiplier (iawfilter) or nullptr
  const double* cos_sa;      // [n_angles]
  const double* sa_rad;      // [n_angles] scattering angles in radians (2-D path)
  const double* w_sa;        // [n_angles]
  const double2* zp;         // [1640]
  const double* xi1;         // [1024]
  const double* xi2;         // [1640]
  const double* taps[2];
  int ntaps[2];              // length of the bin-averaged IRF taps hb
  int toff[2];               // ybin[p] = sum_s hb[s] x[p * ppp + toff + s]
  // phase-layout convolution (points_per_pixel = 1, 256 threads per feature; see "IRF convolution" in k_spectrum):
  const double* ptaps[2];    // hb zero-padded by 3 + rounding on both sides: ptaps[3 + s] = hb[s]
  int cf_i0[2], cf_na[2], cf_a0[2];  // forward: first padded-tap index, tap groups of 4, first slot offset (u0 / 4)
  int ca_i0[2], ca_na[2], ca_a0[2];  // adjoint: first (descending) padded-tap index, groups, slot offset
  int hs;                    // halo of a phase array in slots of 4 samples
  int halo;                  // zero padding on both sides of the LDS spectrum buffers (>= every |tap offset|)
  int halo_bins;             // the same for the per-bin adjoint buffer
  const uint8_t* mask[2];    // [1024]
  const double* p_scale;
  const double* p_shift;
  const uint8_t* p_sig;
  const double* dlm_table;   // [nvx][31]
  const double* lg;          // [1640][1024] log-ratio table of the rationally-centred integration (ratintn.py:49), zero padded
  uint8_t ti_same[TSFF_MAX_ION];
  double vx0, dv;
};

// per-call pointers (device)
struct KCall {
  const double* params;  // [B][NP]
  const double2* ht;     // [slots][nvx]
  const double* W;       // [slots][1640]
  const double2* htm;    // [slots][nvx]  d(ln fe, slope)/dm   (WITH_M)
  const double* Wm;      // [slots][1640] dW/dm                (WITH_M)
  double* Wb_out;        // [B][1640] adjoint of the W table       (gradient w.r.t. f_e, GM == 2)
  double* Hy_out;        // [B][nvx]  adjoint of the ln fe node values
  double* Hs_out;        // [B][nvx]  adjoint of the ln fe node slopes
  const double* amps[2];
  const double* noise[2];
  const double* data[2];
  double* thry[2];
  double* sqdev[2];
  double* lpart;         // [B][3]
  double* gpart;         // [2][B][NP] per-feature gradient parts (interleaved plan)
  double wts[3];
  int B;
  int denom_mode;        // MODE 1 only: 0 constant denominators (folded into wts), 2: |data| + 1e-10 per sample
};

// ------------------------------------------------------------------------------------------
// k_fe_prepare: distribution function -> Hermite table of ln fe and the Re(chi_e) table W
// grid (slots, qsplit), 256 threads.   reference: form_factor.py:263-268, ratintn.py:4-52,
// base.py:277-294 (DLM), interpax _approx_df (node slopes)
// ------------------------------------------------------------------------------------------
template <int NI>
__global__ __launch_bounds__(kThreads) void k_fe_prepare(KStatic S, const double* __restrict__ fe_in, int fe_mode,
                                                         const double* __restrict__ params, double2* __restrict__ ht_out,
                                                         double* __restrict__ W_out, double* __restrict__ fe_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* lnfe = reinterpret_cast<double*>(smem);  // [nvx]
  double* slope = lnfe + S.nvx;                    // [nvx]
  double* ratmod = slope + S.nvx;                  // [1024]
  double* ratdf = ratmod + kNXi1;                  // [1024]
  double* xi1 = ratdf + kNXi1;                     // [1024]
  double* red = xi1 + kNXi1;                       // [8]
  double2* ht = reinterpret_cast<double2*>(red + 8);  // [nvx]

  const int slot = blockIdx.x, tid = threadIdx.x;
  const int nvx = S.nvx;

  // ---- fe on the vx grid ----
  if (fe_mode == TSFF_FE_DLM) {
    // DLM1V.__call__ (base.py:277-294): interp in m over m_ax = linspace(2,5,31), then /sum/dv
    Phys<NI> p;
    load_phys<NI>(params + (size_t)slot * S.NP, S.p_scale, S.p_shift, S.p_sig, S.ti_same, true, p);
    const double m = p.m;
    double u = (m - 2.0) * 10.0;
    int k = (int)u;
    k = k < 0 ? 0 : (k > TSFF_DLM_NM - 2 ? TSFF_DLM_NM - 2 : k);
    double t = (m - (2.0 + 0.1 * k)) * 10.0;
    t = m < 2.0 ? 0.0 : (m > 5.0 ? 1.0 : t);
    double part = 0.0;
    for (int i = tid; i < nvx; i += kThreads) {
      const double a = S.dlm_table[i * TSFF_DLM_NM + k], b = S.dlm_table[i * TSFF_DLM_NM + k + 1];
      const double f = a + t * (b - a);
      lnfe[i] = f;
      part += f;
    }
    const double tot = block_sum(part, red);
    for (int i = tid; i < nvx; i += kThreads) {
      const double f = lnfe[i] / tot / S.dv;
      if (fe_out && blockIdx.y == 0) fe_out[(size_t)slot * nvx + i] = f;
      lnfe[i] = log(f);
    }
  } else {
    for (int i = tid; i < nvx; i += kThreads) lnfe[i] = log(fe_in[(size_t)slot * nvx + i]);
  }
  for (int i = tid; i < kNXi1; i += kThreads) xi1[i] = S.xi1[i];
  __syncthreads();
  // ---- node slopes: mean of adjacent secants, one-sided at the ends ----
  for (int i = tid; i < nvx; i += kThreads) {
    const double dl = i > 0 ? (lnfe[i] - lnfe[i - 1]) / S.dv : 0.0;
    const double dr = i < nvx - 1 ? (lnfe[i + 1] - lnfe[i]) / S.dv : 0.0;
    const double s = i == 0 ? dr : (i == nvx - 1 ? dl : 0.5 * (dl + dr));
    slope[i] = s;
    ht[i] = make_double2(lnfe[i], s);
    if (blockIdx.y == 0) ht_out[(size_t)slot * nvx + i] = make_double2(lnfe[i], s);
  }
  __syncthreads();
  Tables T;
  T.zp = nullptr; T.W = nullptr; T.ht = ht; T.nvx = nvx;
  T.vx0 = S.vx0; T.dv = S.dv; T.idv = 1.0 / S.dv; T.vxlast = S.vx0 + (nvx - 1) * S.dv;
  // ---- ratmod = exp(H(xi1)) (form_factor.py:263) ----
  for (int i = tid; i < kNXi1; i += kThreads) {
    double H, dH;
    hermite_lookup(T, xi1[i], H, dH);
    ratmod[i] = exp(H);
  }
  __syncthreads();
  // ---- ratdf = gradient(ratmod, dxi1) (form_factor.py:264) ----
  const double h1 = xi1[1] - xi1[0];
  for (int i = tid; i < kNXi1; i += kThreads) {
    double g;
    if (i == 0) g = (ratmod[1] - ratmod[0]) / h1;
    else if (i == kNXi1 - 1) g = (ratmod[kNXi1 - 1] - ratmod[kNXi1 - 2]) / h1;
    else g = (ratmod[i + 1] - ratmod[i - 1]) / (2.0 * h1);
    ratdf[i] = g;
  }
  __syncthreads();
  // ---- W[q] = ratintn(ratdf, xi1 - xi2[q], xi1): one wavefront per q, lanes stride the 1022
  //      intervals, xor-shuffle reduction (ratintn.py:21, 41-52; the last interval is dropped) ----
  const int lane = tid & 63, wave = tid >> 6;
  const int qper = (kNXi2 + gridDim.y - 1) / gridDim.y;
  const int q0 = blockIdx.y * qper, q1 = min(kNXi2, q0 + qper);
  for (int q = q0 + wave; q < q1; q += kThreads / 64) {
    const double x2 = S.xi2[q];
    double acc = 0.0;
    for (int i = lane; i < kNXi1 - 2; i += 64) {
      const double f0 = ratdf[i], f1 = ratdf[i + 1];
      const double g0 = xi1[i] - x2, g1 = xi1[i + 1] - x2;
      const double fdif = f1 - f0, gdif = g1 - g0;
      const double fav = 0.5 * (f1 + f0), gav = 0.5 * (g1 + g0);
      const double tmp = fav * gdif - gav * fdif;
      double r;
      if (fabs(gdif) < 1.0e-4 * fabs(gav)) r = fav / gav + tmp * gdif / (12.0 * gav * gav * gav);
      else r = fdif / gdif + tmp * log(fabs((gav + 0.5 * gdif) / (gav - 0.5 * gdif))) / (gdif * gdif);
      acc += r * (xi1[i + 1] - xi1[i]);
    }
    acc = wave_sum(acc);
    if (lane == 0) W_out[(size_t)slot * kNXi2 + q] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// Per-lineout distribution functions (fe_mode DLM / PER_LINEOUT).  The Re(chi_e) table is linear in
// ratdf: with h_i = xi1[i+1]-xi1[i], s_i = fdif_i/h_i, A_i = fav_i - xi1mid_i s_i,
//   W[q] = sum_i fdif_i + sum_i Lg[q][i] (A_i + xi2[q] s_i),   Lg[q][i] = log|(gav+gdif/2)/(gav-gdif/2)|
// (ratintn.py:41-52 with gdif = zdif; the small-gdif branch never triggers on this grid).  Lg is a
// constant 1640x1022 table, so the 1.68 M logarithms per f_e of k_fe_prepare become two matrix-vector
// products per lineout, done for the whole batch by k_wgemm.  The derivative with respect to the DLM
// order m rides along as a second pair of vectors (tangent of f_e -> ln f_e -> ratmod -> ratdf).
//
// k_fe_vectors: grid B, 256 threads.  Outputs ht/htm [B][nvx], X [B][4][1024] = (A, s, dA/dm, ds/dm) zero
// padded, cst [B][2] = (sum fdif, d/dm).
// ------------------------------------------------------------------------------------------
template <int NI>
__global__ __launch_bounds__(kThreads) void k_fe_vectors(KStatic S, const double* __restrict__ fe_in, int fe_mode,
                                                         const double* __restrict__ params, double2* __restrict__ ht_out,
                                                         double2* __restrict__ htm_out, double* __restrict__ X,
                                                         double* __restrict__ cst) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* ht = reinterpret_cast<double2*>(smem);         // [nvx]
  double2* htm = ht + S.nvx;                              // [nvx]
  double2* hc = htm + S.nvx;                              // [2 nvx]
  double2* hcm = hc + 2 * S.nvx;                          // [2 nvx]
  double* lnfe = reinterpret_cast<double*>(hcm + 2 * S.nvx);  // [nvx]
  double* dln = lnfe + S.nvx;                             // [nvx]
  double* rat = dln + S.nvx;                              // [1024] ratmod -> ratdf
  double* ratm = rat + kNXi1;                             // [1024]
  double* rdf = ratm + kNXi1;                             // [1024]
  double* rdfm = rdf + kNXi1;                             // [1024]
  double* red = rdfm + kNXi1;                             // [8]
  const int b = blockIdx.x, tid = threadIdx.x, nvx = S.nvx;

  if (fe_mode == TSFF_FE_DLM) {
    Phys<NI> p;
    load_phys<NI>(params + (size_t)b * S.NP, S.p_scale, S.p_shift, S.p_sig, S.ti_same, true, p);
    const double m = p.m;
    double u = (m - 2.0) * 10.0;
    int k = (int)u;
    k = k < 0 ? 0 : (k > TSFF_DLM_NM - 2 ? TSFF_DLM_NM - 2 : k);
    double t = (m - (2.0 + 0.1 * k)) * 10.0;
    const bool inside = m >= 2.0 && m <= 5.0;   // jnp.interp clamps outside the m axis: zero slope there
    t = m < 2.0 ? 0.0 : (m > 5.0 ? 1.0 : t);
    double part = 0.0, dpart = 0.0;
    for (int i = tid; i < nvx; i += kThreads) {
      const double a = S.dlm_table[i * TSFF_DLM_NM + k], c = S.dlm_table[i * TSFF_DLM_NM + k + 1];
      const double f = a + t * (c - a);
      const double df = inside ? (c - a) * 10.0 : 0.0;
      lnfe[i] = f;
      dln[i] = df;
      part += f;
      dpart += df;
    }
    const double tot = block_sum(part, red);
    const double dtot = block_sum(dpart, red);
    for (int i = tid; i < nvx; i += kThreads) {
      const double f = lnfe[i];
      dln[i] = dln[i] / f - dtot / tot;               // d ln fe / dm
      lnfe[i] = log(f / tot / S.dv);                  // base.py:293
    }
  } else {
    for (int i = tid; i < nvx; i += kThreads) { lnfe[i] = log(fe_in[(size_t)b * nvx + i]); dln[i] = 0.0; }
  }
  __syncthreads();
  for (int i = tid; i < nvx; i += kThreads) {
    const double dl = i > 0 ? (lnfe[i] - lnfe[i - 1]) / S.dv : 0.0, dr = i < nvx - 1 ? (lnfe[i + 1] - lnfe[i]) / S.dv : 0.0;
    const double ml = i > 0 ? (dln[i] - dln[i - 1]) / S.dv : 0.0, mr = i < nvx - 1 ? (dln[i + 1] - dln[i]) / S.dv : 0.0;
    const double sl = i == 0 ? dr : (i == nvx - 1 ? dl : 0.5 * (dl + dr));
    const double sm = i == 0 ? mr : (i == nvx - 1 ? ml : 0.5 * (ml + mr));
    ht[i] = make_double2(lnfe[i], sl);
    htm[i] = make_double2(dln[i], sm);
    ht_out[(size_t)b * nvx + i] = ht[i];
    htm_out[(size_t)b * nvx + i] = htm[i];
  }
  __syncthreads();
  for (int i = tid; i < nvx - 1; i += kThreads) {
    hermite_coeffs(ht[i], ht[i + 1], S.dv, hc[2 * i], hc[2 * i + 1]);
    hermite_coeffs(htm[i], htm[i + 1], S.dv, hcm[2 * i], hcm[2 * i + 1]);
  }
  __syncthreads();
  Tables T;
  T.zp = nullptr; T.W = nullptr; T.ht = ht; T.hc = hc; T.nvx = nvx;
  T.vx0 = S.vx0; T.dv = S.dv; T.idv = 1.0 / S.dv; T.vxlast = S.vx0 + (nvx - 1) * S.dv;
  Tables Tm = T;
  Tm.hc = hcm;
  for (int i = tid; i < kNXi1; i += kThreads) {
    const double x = S.xi1[i];
    double H, dH, Hm, dHm;
    hermite_lookup_c(T, x, H, dH);
    hermite_lookup_c(Tm, x, Hm, dHm);
    const bool out = x < T.vx0 || x > T.vxlast;
    const double r = exp(H);
    rat[i] = r;
    ratm[i] = out ? 0.0 : r * Hm;
  }
  __syncthreads();
  const double h1 = S.xi1[1] - S.xi1[0];
  for (int i = tid; i < kNXi1; i += kThreads) {
    double g, gm;
    if (i == 0) { g = (rat[1] - rat[0]) / h1; gm = (ratm[1] - ratm[0]) / h1; }
    else if (i == kNXi1 - 1) { g = (rat[i] - rat[i - 1]) / h1; gm = (ratm[i] - ratm[i - 1]) / h1; }
    else { g = (rat[i + 1] - rat[i - 1]) / (2.0 * h1); gm = (ratm[i + 1] - ratm[i - 1]) / (2.0 * h1); }
    rdf[i] = g;
    rdfm[i] = gm;
  }
  __syncthreads();
  double c0 = 0.0, c1 = 0.0;
  double* Xb = X + (size_t)b * 4 * kNXi1;
  for (int i = tid; i < kNXi1; i += kThreads) {
    double A = 0.0, Bs = 0.0, Am = 0.0, Bm = 0.0;
    if (i < kNXi1 - 2) {
      const double x0 = S.xi1[i], x1 = S.xi1[i + 1];
      const double ih = 1.0 / (x1 - x0), mid = 0.5 * (x1 + x0);
      const double fd = rdf[i + 1] - rdf[i], fa = 0.5 * (rdf[i + 1] + rdf[i]);
      const double fdm = rdfm[i + 1] - rdfm[i], fam = 0.5 * (rdfm[i + 1] + rdfm[i]);
      Bs = fd * ih; A = fa - mid * Bs;
      Bm = fdm * ih; Am = fam - mid * Bm;
      c0 += fd; c1 += fdm;
    }
    Xb[i] = A; Xb[kNXi1 + i] = Bs; Xb[2 * kNXi1 + i] = Am; Xb[3 * kNXi1 + i] = Bm;
  }
  c0 = block_sum(c0, red);
  c1 = block_sum(c1, red);
  if (tid == 0) { cst[2 * b] = c0; cst[2 * b + 1] = c1; }
}

// k_wgemm: W[b][q] = c0_b + sum_i Lg[q][i] (A_b[i] + xi2[q] s_b[i]) and the same for d/dm: the GEMM
// C[v][q] = sum_k X[v][k] Lg[q][k] (v = 4 vectors per lineout, M = 4B, N = 1640, K = 1024) on the FP64 matrix
// cores (v_mfma_f64_16x16x4_f64).  256 threads = 2 x 2 wavefronts, 64 x 64 per wavefront (4 x 4 MFMA tiles, 64
// accumulator doubles per lane), 128 (vectors = 32 lineouts) x 128 (q) per workgroup, K in chunks of 16 staged
// K-major in LDS with the next chunk prefetched into registers.  Inside every 16-row MFMA tile the rows are ordered
// (component, lineout): a lane's four accumulators (rows (lane>>4) + 4 reg) are then (A, s, dA/dm, ds/dm)·Lg of ONE
// lineout and 16 consecutive q sit on 16 consecutive lanes, so the epilogue needs no shuffle and writes 128-B runs.
constexpr int kGM = 128, kGN = 128, kGK = 16, kGP = 132;  // kGP: LDS row pitch in doubles
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
// NC = 4: (A, s, dA/dm, ds/dm), 4 lineouts per 16-row tile.  NC = 2: no tangents (explicit f_e tables), 8 lineouts per
// tile, rows (component, lineout): a lane then holds (A, A', s, s') of lineouts fk and fk + 4.
template <int NC>
__global__ __launch_bounds__(kThreads, 2) void k_wgemm(const double* __restrict__ Lg, const double* __restrict__ X,
                                                    const double* __restrict__ cst, const double* __restrict__ xi2,
                                                    int B, double* __restrict__ W, double* __restrict__ Wm) {
  __shared__ double Xs[2][kGK][kGP];  // double-buffered: one barrier per K chunk
  __shared__ double Ls[2][kGK][kGP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroup L runs on XCD L % 8; the 13 q-tiles that share one X tile (and every Lg tile) are
  // given to the same XCD so that X is fetched into one L2 only.  Grid = 8 * ceil(nM / 8) * nQ workgroups.
  constexpr int nQ = (kNXi2 + kGN - 1) / kGN;
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int mtile = xcd + 8 * (jx / nQ), qtile = jx % nQ;
  constexpr int LPT = 16 / NC;  // lineouts per 16-row MFMA tile
  if (mtile * (kGM / NC) >= B) return;
  const int q0 = qtile * kGN, b0 = mtile * (kGM / NC);
  // staging: thread -> LDS row lr, kGK/2 consecutive k starting at lk
  const int lr = tid >> 1, lk = (tid & 1) * (kGK / 2);
  constexpr int kPF = kGK / 4;  // double2 loads per thread and operand
  const int sb = b0 + (lr >> 4) * LPT + (lr & (LPT - 1)), sc = (lr & 15) / LPT;  // lineout / component of X row lr
  const double* __restrict__ xrow = X + ((size_t)min(sb, B - 1) * 4 + sc) * kNXi1 + lk;
  const double* __restrict__ lrow = Lg + (size_t)min(q0 + lr, kNXi2 - 1) * kNXi1 + lk;
  mfma_d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
  double2 px[kPF], pl[kPF];
#pragma unroll
  for (int j = 0; j < kPF; ++j) {
    px[j] = *reinterpret_cast<const double2*>(xrow + 2 * j);
    pl[j] = *reinterpret_cast<const double2*>(lrow + 2 * j);
  }
#pragma unroll
  for (int j = 0; j < kPF; ++j) {
    Xs[0][lk + 2 * j][lr] = px[j].x; Xs[0][lk + 2 * j + 1][lr] = px[j].y;
    Ls[0][lk + 2 * j][lr] = pl[j].x; Ls[0][lk + 2 * j + 1][lr] = pl[j].y;
  }
  __syncthreads();
  const int fr = lane & 15, fk = lane >> 4;
  int cur = 0;
  for (int k0 = 0; k0 < kNXi1; k0 += kGK, cur ^= 1) {
    const bool more = k0 + kGK < kNXi1;
    if (more) {
#pragma unroll
      for (int j = 0; j < kPF; ++j) {
        px[j] = *reinterpret_cast<const double2*>(xrow + k0 + kGK + 2 * j);
        pl[j] = *reinterpret_cast<const double2*>(lrow + k0 + kGK + 2 * j);
      }
    }
#pragma unroll
    for (int ks = 0; ks < kGK; ks += 4) {
      double a[4], c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = Xs[cur][ks + fk][wm * 64 + i * 16 + fr];
        c[i] = Ls[cur][ks + fk][wn * 64 + i * 16 + fr];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], c[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < kPF; ++j) {
        Xs[cur ^ 1][lk + 2 * j][lr] = px[j].x; Xs[cur ^ 1][lk + 2 * j + 1][lr] = px[j].y;
        Ls[cur ^ 1][lk + 2 * j][lr] = pl[j].x; Ls[cur ^ 1][lk + 2 * j + 1][lr] = pl[j].y;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int u = 0; u < (NC == 4 ? 1 : 2); ++u) {
      const int bb = b0 + (wm * 4 + i) * LPT + fk + 4 * u;
      if (bb >= B) continue;
      const double c0 = cst[2 * bb], c1 = cst[2 * bb + 1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = q0 + wn * 64 + j * 16 + fr;
        if (q < kNXi2) {
          const double x2 = xi2[q];
          if (NC == 4) {
            W[(size_t)bb * kNXi2 + q] = c0 + acc[i][j][0] + x2 * acc[i][j][1];
            Wm[(size_t)bb * kNXi2 + q] = c1 + acc[i][j][2] + x2 * acc[i][j][3];
          } else {
            W[(size_t)bb * kNXi2 + q] = c0 + acc[i][j][u] + x2 * acc[i][j][2 + u];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Gradient w.r.t. the distribution function itself (fe_mode PER_LINEOUT, SURVEY 8f-1 "free-form f_e").
// k_spectrum<.., GM = 2> leaves per lineout the adjoints of its tables: Wb[1640] (Re chi_e table) and Hy/Hs[nvx]
// (ln fe node values / slopes).  The W table is W = c0 + Lg (A + xi2 s), so its adjoint is the transposed GEMM
//   Y[b][0][i] = sum_q Wb[b][q] Lg[q][i]            (adjoint of A)
//   Y[b][1][i] = sum_q xi2[q] Wb[b][q] Lg[q][i]     (adjoint of s)
// done by k_wgemm_t on the FP64 matrix cores (M = 2B vectors, N = 1024, K = 1640), and k_fe_adjoint walks
// k_fe_vectors backwards (X construction, central differences, exp, Hermite evaluation at xi1, node slopes, ln).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 2) void k_wgemm_t(const double* __restrict__ Lg, const double* __restrict__ Wb,
                                                         const double* __restrict__ xi2, int B, double* __restrict__ Y) {
  __shared__ double Vs[kGK][kGP];   // [k][vector]
  __shared__ double Ls[kGK][kGP];   // [k][i]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int v0 = blockIdx.y * kGM, i0 = blockIdx.x * kGN;
  // staging of V: thread -> vector row lr, 8 consecutive q starting at lk; of Lg: row (q) sk, 8 consecutive i at sn
  const int lr = tid >> 1, lk = (tid & 1) * 8;
  const int vb = (v0 + lr) >> 1, vc = (v0 + lr) & 1;
  const double* __restrict__ vrow = Wb + (size_t)min(vb, B - 1) * kNXi2 + lk;
  const int sk = tid >> 4, sn = (tid & 15) * 8;
  const double* __restrict__ lrow = Lg + (size_t)sk * kNXi1 + i0 + sn;
  mfma_d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
  const int fr = lane & 15, fk = lane >> 4;
  for (int k0 = 0; k0 < kNXi2; k0 += kGK) {
    double pv[8];
    double2 pl[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = k0 + lk + j;
      const double w = q < kNXi2 ? vrow[k0 + j] : 0.0;
      pv[j] = vc ? w * (q < kNXi2 ? xi2[q] : 0.0) : w;
    }
    const bool lok = k0 + sk < kNXi2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pl[j] = lok ? *reinterpret_cast<const double2*>(lrow + (size_t)k0 * kNXi1 + 2 * j) : make_double2(0.0, 0.0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) Vs[lk + j][lr] = pv[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<double2*>(&Ls[sk][sn + 2 * j]) = pl[j];
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < kGK; ks += 4) {
      double a[4], c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = Vs[ks + fk][wm * 64 + i * 16 + fr];
        c[i] = Ls[ks + fk][wn * 64 + i * 16 + fr];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], c[j], acc[i][j], 0, 0, 0);
    }
  }
  // C/D layout: col = lane & 15 (i), row = (lane >> 4) + 4 reg (vector inside the 16-row tile)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int v = v0 + wm * 64 + i * 16 + fk + 4 * r;
      if (v >= 2 * B) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) Y[(size_t)v * kNXi1 + i0 + wn * 64 + j * 16 + fr] = acc[i][j][r];
    }
}

// k_fe_adjoint: grid B, 256 threads.  In: ht [B][nvx] (ln fe, slope) of the forward pass, Y [B][2][1024], Wb [B][1640]
// (for the adjoint of c0 = sum_q Wb), Hy/Hs [B][nvx] from k_spectrum.  Out: dfe [B][nvx] = d loss / d f_e.
__global__ __launch_bounds__(kThreads) void k_fe_adjoint(KStatic S, const double2* __restrict__ ht_in,
                                                         const double* __restrict__ Y, const double* __restrict__ Wb,
                                                         const double* __restrict__ Hy_in, const double* __restrict__ Hs_in,
                                                         double* __restrict__ dfe) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* ht = reinterpret_cast<double2*>(smem);   // [nvx]
  double2* hc = ht + S.nvx;                         // [2 nvx]
  double* rat = reinterpret_cast<double*>(hc + 2 * S.nvx);  // [1024]
  double* fdb = rat + kNXi1;                        // [1024] adjoint of fdif, later of rat
  double* fab = fdb + kNXi1;                        // [1024] adjoint of fav
  double* rdb = fab + kNXi1;                        // [1024] adjoint of ratdf
  double* yb = rdb + kNXi1;                         // [nvx]
  double* sb = yb + S.nvx;                          // [nvx]
  double* red = sb + S.nvx;                         // [8]
  const int b = blockIdx.x, tid = threadIdx.x, nvx = S.nvx;
  for (int i = tid; i < nvx; i += kThreads) {
    ht[i] = ht_in[(size_t)b * nvx + i];
    yb[i] = Hy_in[(size_t)b * nvx + i];
    sb[i] = Hs_in[(size_t)b * nvx + i];
  }
  double part = 0.0;
  for (int q = tid; q < kNXi2; q += kThreads) part += Wb[(size_t)b * kNXi2 + q];
  const double c0b = block_sum(part, red);  // (contains the barrier that publishes ht)
  for (int i = tid; i < nvx - 1; i += kThreads) hermite_coeffs(ht[i], ht[i + 1], S.dv, hc[2 * i], hc[2 * i + 1]);
  __syncthreads();
  Tables T;
  T.zp = nullptr; T.W = nullptr; T.ht = ht; T.hc = hc; T.hcm = nullptr; T.Wm = nullptr; T.nvx = nvx;
  T.Wb = nullptr; T.Hy = yb; T.Hs = sb;
  T.vx0 = S.vx0; T.dv = S.dv; T.idv = 1.0 / S.dv; T.vxlast = S.vx0 + (nvx - 1) * S.dv;
  const double* Ab = Y + (size_t)b * 2 * kNXi1;
  const double* Sb = Ab + kNXi1;
  for (int i = tid; i < kNXi1; i += kThreads) {
    double H, dH;
    hermite_lookup_c(T, S.xi1[i], H, dH);
    rat[i] = exp(H);
    // X construction (k_fe_vectors): Bs = fd ih, A = fa - mid Bs, c0 += fd   for i < 1022
    double fd = 0.0, fa = 0.0;
    if (i < kNXi1 - 2) {
      const double x0 = S.xi1[i], x1 = S.xi1[i + 1];
      const double ih = 1.0 / (x1 - x0), mid = 0.5 * (x1 + x0);
      fa = Ab[i];
      fd = (Sb[i] - mid * fa) * ih + c0b;
    }
    fdb[i] = fd;
    fab[i] = fa;
  }
  __syncthreads();
  // fd_i = rdf[i+1] - rdf[i], fa_i = (rdf[i+1] + rdf[i]) / 2
  for (int j = tid; j < kNXi1; j += kThreads) {
    double v = 0.0;
    if (j < kNXi1 - 2) v += -fdb[j] + 0.5 * fab[j];
    if (j >= 1 && j <= kNXi1 - 2) v += fdb[j - 1] + 0.5 * fab[j - 1];
    rdb[j] = v;
  }
  __syncthreads();
  // rdf = gradient(rat, h1): central inside, one-sided at the ends
  const double ih1 = 1.0 / (S.xi1[1] - S.xi1[0]);
  const int n = kNXi1;
  for (int j = tid; j < n; j += kThreads) {
    double v = 0.0;
    if (j + 1 >= 1 && j + 1 <= n - 2) v -= 0.5 * ih1 * rdb[j + 1];
    if (j - 1 >= 1 && j - 1 <= n - 2) v += 0.5 * ih1 * rdb[j - 1];
    if (j == 1) v += ih1 * rdb[0];
    if (j == 0) v -= ih1 * rdb[0];
    if (j == n - 1) v += ih1 * rdb[n - 1];
    if (j == n - 2) v -= ih1 * rdb[n - 1];
    fdb[j] = v * rat[j];  // adjoint of H(xi1_j)  (rat = exp H)
  }
  __syncthreads();
  {  // scatter to the nodes: each thread walks consecutive xi1 points, so the run-length accumulators rarely flush
    FeAcc fa;
    fe_acc_init(fa);
    const int per = (kNXi1 + kThreads - 1) / kThreads;
    for (int j = tid * per; j < min((tid + 1) * per, kNXi1); ++j) fe_add_h(fa, T, S.xi1[j], fdb[j]);
    fe_flush_h(fa, yb, sb);
  }
  __syncthreads();
  // node slopes: mean of the adjacent secants, one-sided at the ends (k_fe_vectors)
  for (int j = tid; j < nvx; j += kThreads) {
    double v = yb[j];
    const double idv = 1.0 / S.dv;
    if (j >= 1) v += sb[j - 1] * (j - 1 == 0 ? idv : 0.5 * idv);        // s_{j-1} reads y_j with +
    if (j + 1 <= nvx - 1) v -= sb[j + 1] * (j + 1 == nvx - 1 ? idv : 0.5 * idv);  // s_{j+1} reads y_j with -
    if (j == 0) v -= sb[0] * idv;
    if (j == nvx - 1) v += sb[nvx - 1] * idv;
    dfe[(size_t)b * nvx + j] = v * exp(-ht[j].x);  // y = ln fe
  }
}

// ------------------------------------------------------------------------------------------
// shared helpers of the spectrum kernels
// ------------------------------------------------------------------------------------------
constexpr int kHalf = 256;  // threads per feature when a k_spectrum workgroup evaluates both features (TPF)
#ifndef TSFF_QUNROLL
#define TSFF_QUNROLL 1  // unroll factor of the strip loop (points interleaved per thread)
#endif
#define TSFF_PRAGMA(x) _Pragma(#x)
#define TSFF_UNROLL(n) TSFF_PRAGMA(unroll n)
#ifndef TSFF_OCC
#define TSFF_OCC 2  // wavefronts per SIMD the register allocator must leave room for (one 512-thread workgroup per CU)
#endif

struct Smem {
  double2* zp;    // [1640]
  double2* ht;    // [nvx]
  double* W;      // [1640]
  double* x;      // [nfeat][halo + npts + halo]   model spectrum of each feature, later its adjoint
  double* yb;     // [nfeat][halo_bins + 1024 + halo_bins]   adjoint of the binned spectrum (MODE 1)
  double* taps;   // [ntaps[0] + ntaps[1]] bin-averaged IRF taps of both features
  double2* hc;    // [2*(nvx-1)] Hermite coefficients per interval
  double2* hcm;   // the same for d ln fe / dm
  double* Wm;     // [1640] dW/dm
  double* ksc;    // [nfeat][npts + 1] k_s cache of the current gradient point
  double* phys;   // [kNP_MAX + 1] physical parameters of this lineout
  double* cosa;   // [n_angles]
  double* wsa;    // [n_angles]
  double* red;    // [8 * kNP_MAX + 64]
};

// one spectrum / per-bin adjoint buffer: linear with a zero halo, or four phase arrays (sample j -> phase j & 3, slot
// j >> 2) with a halo of hs slots each -- sized for the larger of the two layouts
__host__ __device__ inline size_t xbuf_doubles(const KStatic& S) {
  const size_t lin = (size_t)S.npts + 2 * (size_t)S.halo, ph = (size_t)S.npts + 8 * (size_t)S.hs;
  return (lin > ph ? lin : ph) + 2;
}
__host__ __device__ inline size_t ybuf_doubles(const KStatic& S) {
  const size_t lin = (size_t)TSFF_NBINS + 2 * (size_t)S.halo_bins, ph = (size_t)TSFF_NBINS + 8 * (size_t)S.hs;
  return (lin > ph ? lin : ph) + 2;
}

// LDS budget (in doubles) of one k_spectrum / k_form_factor workgroup holding `nfeat` features;
// with_m: tangent tables of the DLM order; with_ks: k_s cache
__host__ __device__ inline size_t smem_doubles(const KStatic& S, int nfeat, bool with_m, bool with_ks) {
  size_t n = 2 * (size_t)(kNXi2 + S.nvx) + kNXi2 + 4 * (size_t)S.nvx;                    // zp, ht, W, hc
  if (with_m) n += 4 * (size_t)S.nvx + kNXi2;                                               // hcm, Wm
  n += (size_t)nfeat * xbuf_doubles(S);                                                     // spectrum buffers
  n += (size_t)nfeat * ybuf_doubles(S);                                                     // per-bin adjoint buffers
  if (with_ks) n += (size_t)nfeat * ((size_t)S.npts + 2);                                   // k_s cache
  n += S.ntaps[0] + S.ntaps[1] + 2 * (size_t)S.n_angles + 11 * kNP_MAX + 66;                // taps, angles, phys, scratch
  return n;
}

__device__ __forceinline__ Smem carve(unsigned char* smem, const KStatic& S, int nfeat, bool with_m, bool with_ks) {
  Smem m;
  m.zp = reinterpret_cast<double2*>(smem);
  m.ht = m.zp + kNXi2;
  m.W = reinterpret_cast<double*>(m.ht + S.nvx);
  double* p = m.W + kNXi2;
  // one buffer per feature holds the model spectrum x and later its adjoint, another the adjoint of the binned
  // spectrum; zero halos on both sides so the convolutions need no bounds checks
  m.x = p; p += (size_t)nfeat * xbuf_doubles(S);
  m.yb = p; p += (size_t)nfeat * ybuf_doubles(S);
  m.hc = reinterpret_cast<double2*>(p); p += 4 * (size_t)S.nvx;
  m.hcm = nullptr; m.Wm = nullptr;
  if (with_m) { m.hcm = reinterpret_cast<double2*>(p); p += 4 * (size_t)S.nvx; m.Wm = p; p += kNXi2; }
  m.ksc = nullptr;
  if (with_ks) { m.ksc = p; p += (size_t)nfeat * (S.npts + 2); }
  m.taps = p; p += S.ntaps[0] + S.ntaps[1];
  m.phys = p; p += kNP_MAX + 2;
  m.cosa = p; p += S.n_angles;
  m.wsa = p; p += S.n_angles;
  m.red = p;
  return m;
}

__device__ __forceinline__ void load_tables(const Smem& m, const KStatic& S, const KCall& K, int slot, bool with_taps,
                                            Tables& T) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < kNXi2; i += nthr) {
    m.zp[i] = S.zp[i];
    m.W[i] = K.W[(size_t)slot * kNXi2 + i];
  }
  for (int i = tid; i < S.nvx; i += nthr) m.ht[i] = K.ht[(size_t)slot * S.nvx + i];
  for (int i = tid; i < S.nvx - 1; i += nthr) {
    double2 c01, c23;
    hermite_coeffs(K.ht[(size_t)slot * S.nvx + i], K.ht[(size_t)slot * S.nvx + i + 1], S.dv, c01, c23);
    m.hc[2 * i] = c01;
    m.hc[2 * i + 1] = c23;
  }
  if (m.hcm && K.htm) {  // tangent tables of the DLM order
    for (int i = tid; i < S.nvx - 1; i += nthr) {
      double2 c01, c23;
      hermite_coeffs(K.htm[(size_t)slot * S.nvx + i], K.htm[(size_t)slot * S.nvx + i + 1], S.dv, c01, c23);
      m.hcm[2 * i] = c01;
      m.hcm[2 * i + 1] = c23;
    }
    for (int i = tid; i < kNXi2; i += nthr) m.Wm[i] = K.Wm[(size_t)slot * kNXi2 + i];
  }
  if (with_taps) {
    for (int i = tid; i < S.ntaps[0]; i += nthr) m.taps[i] = S.taps[0][i];
    for (int i = tid; i < S.ntaps[1]; i += nthr) m.taps[S.ntaps[0] + i] = S.taps[1][i];
  }
  for (int i = tid; i < S.n_angles; i += nthr) { m.cosa[i] = S.cos_sa[i]; m.wsa[i] = S.w_sa[i]; }
  T.zp = m.zp; T.W = m.W; T.ht = m.ht; T.hc = m.hc; T.hcm = m.hcm; T.Wm = m.Wm; T.nvx = S.nvx;
  T.Wb = nullptr; T.Hy = nullptr; T.Hs = nullptr;
  T.vx0 = S.vx0; T.dv = S.dv; T.idv = 1.0 / S.dv; T.vxlast = S.vx0 + (S.nvx - 1) * S.dv;
}

// physical parameters of lineout `xpar` -> LDS (one activation per thread, then Ti tying and fraction
// renormalisation by one thread; ts_params.py:329-350, 543-563).  Contains workgroup barriers.
template <int NI>
__device__ __forceinline__ void stage_phys(const KStatic& S, const double* __restrict__ xpar, double* ph) {
  constexpr int NPk = TSFF_NP(NI);
  const int tid = threadIdx.x;
  if (tid < NPk) {
    const double v = xpar[tid];
    ph[tid] = (S.p_sig[tid] ? sigmoid(v) : v) * S.p_scale[tid] + S.p_shift[tid];
  }
  __syncthreads();
  if (tid == 0) {
    double fsum = 0.0;
#pragma unroll
    for (int s = 0; s < NI; ++s) {
      const int o = TSFF_P_ION0 + 4 * s;
      if (s > 0 && S.ti_same[s]) ph[o + TSFF_ION_TI] = ph[TSFF_P_ION0 + TSFF_ION_TI];
      fsum += ph[o + TSFF_ION_FRACT];
    }
#pragma unroll
    for (int s = 0; s < NI; ++s) ph[TSFF_P_ION0 + 4 * s + TSFF_ION_FRACT] /= fsum;
    ph[NPk] = fsum;
  }
  __syncthreads();
}

// reductions over the NW wavefronts of one feature group; every thread of the WORKGROUP must call them
// (they contain workgroup barriers).  scratch: 32 doubles.
template <int NW>
__device__ __forceinline__ double half_sum(double v, double* scratch, int half, int hw, int lane) {
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[half * NW + hw] = v;
  __syncthreads();
  const double* r = scratch + half * NW;
  double s = (r[0] + r[1]) + (r[2] + r[3]);
  if (NW == 8) s += (r[4] + r[5]) + (r[6] + r[7]);
  return s;
}

template <int NW>
__device__ __forceinline__ void half_argmax(double& v, int& idx, double* scratch, int half, int hw, int lane) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  __syncthreads();
  if (lane == 0) { scratch[half * 2 * NW + hw] = v; scratch[half * 2 * NW + NW + hw] = (double)idx; }
  __syncthreads();
  const double* r = scratch + half * 2 * NW;
  v = r[0]; idx = (int)r[NW];
#pragma unroll
  for (int k = 1; k < NW; ++k) {
    const double ov = r[k]; const int oi = (int)r[NW + k];
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}

// loss functional e(d, t) and de/dt (loss_function.py:386-418); the 1/uncert of l1/l2 is folded
// into the weights by the host (constant denominators) or applied here (theory denominator).
__device__ __forceinline__ void loss_point(int method, double d, double t, double& e, double& det) {
  const double r = d - t;
  if (method == TSFF_LOSS_L2) { e = r * r; det = -2.0 * r; }
  else if (method == TSFF_LOSS_L1) { e = fabs(r); det = r > 0.0 ? -1.0 : (r < 0.0 ? 1.0 : 0.0); }
  else if (method == TSFF_LOSS_LOGCOSH) { e = log(cosh(r)); det = -tanh(r); }
  else { e = t - d * log(t); det = 1.0 - d / t; }
}

// ------------------------------------------------------------------------------------------
// k_spectrum: one workgroup per lineout; threads [0,256) evaluate the first loaded feature, threads
// [256,512) the second (EPW, IAW), sharing the LDS copies of the Z', W and ln f_e tables.  A deck with
// one feature launches 256-thread workgroups.
//   MODE 0: ThryE/ThryI                              (ThomsonScatteringDiagnostic.__call__)
//   MODE 1: + masked loss sums + adjoint -> grad      (LossFunction.vg_loss)
//   MODE 2: + per-lineout sums, theory denominator, sqdev arrays (LossFunction.array_loss)
// ------------------------------------------------------------------------------------------
template <int NI, int MODE, int GM = 0, int TPF = kHalf>
__global__ __launch_bounds__(2 * kHalf, TSFF_OCC) void k_spectrum(KStatic S, KCall K, int f0, int nfeat, int flags,
                                                           const uint8_t* __restrict__ gmask, double* __restrict__ grad) {
  // flags bit 0: add to grad instead of overwriting it (second launch of a feature-split call);
  //       bit 1: k_s cache present in LDS;
  //       bit 2: interleaved plan -- grid 2B, workgroup f B + b evaluates feature f of lineout b and leaves its part of
  //              the gradient in K.gpart[f][b][:]; k_loss_reduce adds the two parts
  const bool accumulate = flags & 1, use_ks = flags & 2, interleaved = flags & 4;
  const int f_il = interleaved ? (int)(blockIdx.x >= (unsigned)K.B) : 0;  // first B workgroups: feature 0, next B: feature 1
  const int b = interleaved ? (int)blockIdx.x - f_il * K.B : (int)blockIdx.x, tid = threadIdx.x;
  // TPF threads per feature: 256 when the workgroup holds both features, 512 for a one-feature workgroup
  constexpr int NW = TPF / 64;          // wavefronts per feature
  constexpr int BPT = TSFF_NBINS / TPF;  // output bins per thread
  const int half = tid / TPF, ht = tid % TPF, lane = tid & 63, hw = ht >> 6;
  // feature of this group: wavefront-uniform, kept in an SGPR so that everything indexed by it (tap counts, axis
  // pointers, the padded taps of the convolution) is read through scalar loads
  const int f = __builtin_amdgcn_readfirstlane(interleaved ? f_il : f0 + half);
  extern __shared__ __align__(16) unsigned char smem[];
  const Smem m = carve(smem, S, nfeat, GM != 0, use_ks);
  Tables T;
  load_tables(m, S, K, S.shared_fe ? 0 : b, true, T);

  const double* __restrict__ xpar = K.params + (size_t)b * S.NP;
  stage_phys<NI>(S, xpar, m.phys);
  const double lam_shift = S.lam_shift[f];
  const double* __restrict__ omgs = S.omgs[f];
  const int npts = S.npts, ppp = S.ppp, G = S.G, NA = S.n_angles, nstrips = S.npts / kStrip;
  const double invG = 1.0 / (double)G;
  // (pointer arithmetic on the LDS base, not a runtime-indexed pointer array: keeps ds_* addressing)
  // Layout of the spectrum buffer x (and of the per-bin adjoint buffer yb).  Linear: x[H + j] with a zero halo.  Phase
  // layout (points_per_pixel = 1, 256 threads per feature): sample j lives at x[(j & 3) Ls + (j >> 2) + hs], four
  // arrays of Ls = npts/4 + 2 hs slots.  A thread owns the four CONSECUTIVE samples 4 ht .. 4 ht + 3 (its strip of the
  // sweeps, its bins of the convolution), and every access of a wavefront is to 64 consecutive doubles of one phase
  // array (no bank conflicts), which lets the convolution slide a register window over adjacent samples (below).
  const bool ph = (TPF == kHalf) && ppp == 1;
  const int H = S.halo, Hb = S.halo_bins, hs = S.hs, Ls = npts / 4 + 2 * hs;
  double* __restrict__ xs = m.x + half * xbuf_doubles(S);
  double* __restrict__ ybs = m.yb + half * ybuf_doubles(S);
  auto XA = [&](int j) { return ph ? (j & 3) * Ls + (j >> 2) + hs : H + j; };     // sample j of x
  auto YA = [&](int pbin) { return ph ? (pbin & 3) * Ls + (pbin >> 2) + hs : Hb + pbin; };  // bin p of yb
  auto PB = [&](int r) { return ph ? 4 * ht + r : ht + TPF * r; };                // r-th bin of this thread
  for (int i = ht; i < (int)xbuf_doubles(S); i += TPF) xs[i] = 0.0;
  for (int i = ht; i < (int)ybuf_doubles(S); i += TPF) ybs[i] = 0.0;
  __syncthreads();

  // ================= forward sweep over (gradient point, lambda strip, angle) =================
  // each thread owns strips of kStrip consecutive samples; the right neighbour's (xi_e, F) needed by the
  // finite difference along lambda (form_factor.py:258) is evaluated by the owner of the strip.
  double* __restrict__ ksc = use_ks ? m.ksc + half * (S.npts + 2) : nullptr;  // k_s(lambda) of the current gradient point
  for (int g = 0; g < G; ++g) {
    LineS<NI> L;
    {
      Phys<NI> p;  // re-read from LDS where needed instead of being kept live across the sweeps
      phys_from_lds<NI>(m.phys, p);
      make_lines_uniform<NI>(p, lam_shift, g, G, L);
    }
    if (use_ks) {
      if (g > 0) __syncthreads();
      for (int i = ht; i < npts; i += TPF) ksc[i] = ks_eval(omgs[i], L.wpe2);  // angle independent (form_factor.py:218)
      __syncthreads();
    }
    for (int st = ht; st < nstrips; st += TPF) {
      const int j0 = kStrip * st;
      // the frequency axis is read from global memory (L1/L2): the strip's first two samples once per
      // chunk, the others one iteration ahead of their use so the load latency hides behind a whole point
      const double ws_first = omgs[j0], ws_second = omgs[min(j0 + 1, npts - 1)];
      for (int a = 0; a < NA; ++a) {
        const double ct = uni(m.cosa[a]), wa = uni(m.wsa[a] * invG * L.pref);   // (pref: see point_forward_sd)
        double ws = ws_first, wnext = ws_second;
        Base b0;
        base_eval<NI>(ws, use_ks ? ksc[j0] : ks_eval(ws, L.wpe2), ct, L, T, b0);
        TSFF_UNROLL(TSFF_QUNROLL)
        for (int q = 0; q < kStrip; ++q) {
          const int j = j0 + q;
          const bool has_next = (j + 1) < npts;
          const double wsn = wnext;
          wnext = omgs[min(j + 2, npts - 1)];
          Base b1;
          base_eval<NI>(wsn, use_ks ? ksc[min(j + 1, npts - 1)] : ks_eval(wsn, L.wpe2), ct, L, T, b1);
          xs[XA(j)] += wa * point_forward_sd<NI>(b0, b1, has_next, L, T);
          b0 = b1;
          ws = wsn;
        }
      }
    }
  }
  {  // the factor ws^2 of every sample (left out of the sweep) and the notch filter of the electron feature
    const bool filt = f == TSFF_FEATURE_ELE && S.filt;
    for (int st = ht; st < nstrips; st += TPF) {
      const int j0 = kStrip * st;
#pragma unroll
      for (int q = 0; q < kStrip; ++q) {
        const double w = omgs[j0 + q];
        xs[XA(j0 + q)] *= filt ? w * w * S.filt[j0 + q] : w * w;
      }
    }
  }
  __syncthreads();

  // ================= IRF convolution ("same"), bin average, normalisation =================
  // the host folds the bin average into the taps: hb[s] = (1/ppp) sum_jj g[jj - s + nt - 1], so that
  // ybin[p] = sum_s hb[s] x[p ppp + toff + s]  (irf.py:72-74 / 114,124 in one pass; zero halo -> no bounds checks)
  const int nh = S.ntaps[f], toff = S.toff[f];
  const double* __restrict__ taps = m.taps + (f == TSFF_FEATURE_ELE ? 0 : S.ntaps[0]);
  double ybin[BPT];
#pragma unroll
  for (int r = 0; r < BPT; ++r) ybin[r] = 0.0;
  if (ph) {
    // y[4 ht + r] = sum_u g(u) x[4 ht + r + u]: the taps are walked in groups of four (u = 4 a + c); a group needs the
    // seven samples V[k] = x[4 (ht + a) + k], of which three carry over from the previous group -- four LDS reads
    // (conflict-free, one per phase array) and four scalar tap loads per sixteen FMAs, against five LDS reads per four
    // FMAs of the linear form: the convolution stops being LDS-bandwidth bound.
    // (constant address space: the padded taps are read-only for the whole launch, so the wavefront-uniform reads
    //  below become scalar loads and cost neither LDS bandwidth nor vector-memory instructions)
    typedef const double __attribute__((address_space(4))) cdouble;
    cdouble* pt = (cdouble*)(S.ptaps[f] + S.cf_i0[f]);
    const double* __restrict__ X0 = xs, * __restrict__ X1 = xs + Ls, * __restrict__ X2 = xs + 2 * Ls, * __restrict__ X3 = xs + 3 * Ls;
    int sl = ht + S.cf_a0[f] + hs;
    double V0 = X0[sl], V1 = X1[sl], V2 = X2[sl];
    const int na = S.cf_na[f];
    for (int a = 0; a < na; ++a, ++sl) {
      const double V3 = X3[sl], V4 = X0[sl + 1], V5 = X1[sl + 1], V6 = X2[sl + 1];
      const double g0 = pt[4 * a], g1 = pt[4 * a + 1], g2 = pt[4 * a + 2], g3 = pt[4 * a + 3];
      // (index % BPT: the branch is dead, but must compile, in the 512-threads-per-feature instantiation)
      ybin[0] += g0 * V0 + g1 * V1 + g2 * V2 + g3 * V3;
      ybin[1 % BPT] += g0 * V1 + g1 * V2 + g2 * V3 + g3 * V4;
      ybin[2 % BPT] += g0 * V2 + g1 * V3 + g2 * V4 + g3 * V5;
      ybin[3 % BPT] += g0 * V3 + g1 * V4 + g2 * V5 + g3 * V6;
      V0 = V4; V1 = V5; V2 = V6;
    }
  } else {
    const double* __restrict__ x0 = xs + H + toff + ht * ppp;
    const int rs = TPF * ppp;
#pragma unroll 4
    for (int t = 0; t < nh; ++t) {  // one tap read feeds all the thread's bins
      const double g = taps[t];
#pragma unroll
      for (int r = 0; r < BPT; ++r) ybin[r] += g * x0[r * rs + t];
    }
  }
  double M = ybin[0];
  int pstar = PB(0);
#pragma unroll
  for (int r = 1; r < BPT; ++r)
    if (ybin[r] > M) { M = ybin[r]; pstar = PB(r); }
  half_argmax<NW>(M, pstar, m.red, half, hw, lane);
  const double invM = 1.0 / M;
  const double amps = K.amps[f][b];
  double p_lam, p_amp1, p_amp2, p_amp3;
  {
    p_lam = uni(m.phys[TSFF_P_LAM]); p_amp1 = uni(m.phys[TSFF_P_AMP1]);
    p_amp2 = uni(m.phys[TSFF_P_AMP2]); p_amp3 = uni(m.phys[TSFF_P_AMP3]);
  }
  const double* __restrict__ lamb = S.lam_bin[f];
  double Tb[BPT];  // dLoss/dT (MODE 1)
  double Ap[BPT];  // amplitude factor of bin p
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int r = 0; r < BPT; ++r) {
    const int pb = PB(r);
    double A;
    if (f == TSFF_FEATURE_ELE) A = amps * (lamb[pb] < p_lam ? p_amp1 : p_amp2);  // irf.py:126-130
    else A = amps * p_amp3;                                                      // irf.py:76
    Ap[r] = A;
    double t = A * ybin[r] * invM;
    if (K.noise[f]) t += K.noise[f][(size_t)b * TSFF_NBINS + pb];             // thomson_diagnostic.py:139-140
    if (K.thry[f]) K.thry[f][(size_t)b * TSFF_NBINS + pb] = t;
    Tb[r] = 0.0;
    if (MODE >= 1) {
      const double d = K.data[f][(size_t)b * TSFF_NBINS + pb];
      const uint8_t mk = S.mask[f][pb];
      double e, det;
      loss_point(S.loss_method, d, t, e, det);
      if (MODE == 2) {
        if (S.loss_method == TSFF_LOSS_L2 || S.loss_method == TSFF_LOSS_L1) e /= t;  // loss_function.py:320-321
        double sq = 0.0;
        if (mk & 1) { s0 += e; sq += e; }
        if (mk & 2) { s1 += e; sq += e; }
        if (K.sqdev[f]) K.sqdev[f][(size_t)b * TSFF_NBINS + pb] = sq;
      } else {
        if (K.denom_mode == 2 && (S.loss_method == TSFF_LOSS_L2 || S.loss_method == TSFF_LOSS_L1)) {
          const double iden = 1.0 / (fabs(d) + 1e-10);  // loss_function.py:183 (_loss_for_hess_fn_)
          e *= iden;
          det *= iden;
        }
        double w = 0.0;
        if (mk & 1) { s0 += e; w += (f == TSFF_FEATURE_ELE ? K.wts[1] : K.wts[0]); }
        if (mk & 2) { s1 += e; w += K.wts[2]; }
        Tb[r] = w != 0.0 ? det * w : 0.0;  // (samples outside every fit range may hold anything, NaN included)
      }
    }
  }
  if (MODE == 0) return;
  s0 = half_sum<NW>(s0, m.red, half, hw, lane);
  s1 = half_sum<NW>(s1, m.red, half, hw, lane);
  if (ht == 0) {
    if (f == TSFF_FEATURE_ELE) { K.lpart[(size_t)b * 3 + 1] = s0; K.lpart[(size_t)b * 3 + 2] = s1; }
    else K.lpart[(size_t)b * 3 + 0] = s0;
  }
  if (MODE == 2) return;

  // ================= adjoint of normalisation + binning =================
  // T_p = A_p ybin_p / M (+ noise), M = max_p ybin_p attained at pstar
  double sn = 0.0, a1b = 0.0, a2b = 0.0;
#pragma unroll
  for (int r = 0; r < BPT; ++r) {
    const int pb = PB(r);
    const double u = Tb[r] * ybin[r] * invM;  // dL/dA_p
    sn += u * Ap[r];
    if (f == TSFF_FEATURE_ELE) { if (lamb[pb] < p_lam) a1b += u * amps; else a2b += u * amps; }
    else a1b += u * amps;
  }
  sn = half_sum<NW>(sn, m.red, half, hw, lane);
  a1b = half_sum<NW>(a1b, m.red, half, hw, lane);
  a2b = half_sum<NW>(a2b, m.red, half, hw, lane);
#pragma unroll
  for (int r = 0; r < BPT; ++r) {
    const int pb = PB(r);
    double yb = Tb[r] * Ap[r] * invM;
    if (pb == pstar) yb -= sn * invM;
    ybs[YA(pb)] = yb;
  }
  __syncthreads();
  // ================= adjoint of convolution + binning: xbar_i = filt_i sum_p ybar_p hb[i - p ppp - toff] =================
  if (ph) {
    // xbar[4 ht + r] = sum_u g'(u) ybar[4 ht + r + u], g'(u) = hb[-toff - u]: the same sliding window, taps read backwards
    typedef const double __attribute__((address_space(4))) cdouble;
    cdouble* pt = (cdouble*)(S.ptaps[f] + S.ca_i0[f]);
    const double* __restrict__ Y0 = ybs, * __restrict__ Y1 = ybs + Ls, * __restrict__ Y2 = ybs + 2 * Ls, * __restrict__ Y3 = ybs + 3 * Ls;
    int sl = ht + S.ca_a0[f] + hs;
    double V0 = Y0[sl], V1 = Y1[sl], V2 = Y2[sl];
    double sx[4] = {0.0, 0.0, 0.0, 0.0};
    const int na = S.ca_na[f];
    for (int a = 0; a < na; ++a, ++sl) {
      const double V3 = Y3[sl], V4 = Y0[sl + 1], V5 = Y1[sl + 1], V6 = Y2[sl + 1];
      const double g0 = pt[-4 * a], g1 = pt[-4 * a - 1], g2 = pt[-4 * a - 2], g3 = pt[-4 * a - 3];
      sx[0] += g0 * V0 + g1 * V1 + g2 * V2 + g3 * V3;
      sx[1] += g0 * V1 + g1 * V2 + g2 * V3 + g3 * V4;
      sx[2] += g0 * V2 + g1 * V3 + g2 * V4 + g3 * V5;
      sx[3] += g0 * V3 + g1 * V4 + g2 * V5 + g3 * V6;
      V0 = V4; V1 = V5; V2 = V6;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * ht + r;
      double v = sx[r] * (omgs[i] * omgs[i]);   // (the seed of the reverse sweep carries ws^2, see point_reverse)
      if (f == TSFF_FEATURE_ELE && S.filt) v *= S.filt[i];
      xs[XA(i)] = v * invG;
    }
  } else if (ppp == 1) {
    double sx[BPT];
#pragma unroll
    for (int r = 0; r < BPT; ++r) sx[r] = 0.0;
    const double* __restrict__ y0 = ybs + Hb - toff + ht;
#pragma unroll 4
    for (int t = 0; t < nh; ++t) {
      const double g = taps[t];
#pragma unroll
      for (int r = 0; r < BPT; ++r) sx[r] += g * y0[r * TPF - t];
    }
#pragma unroll
    for (int r = 0; r < BPT; ++r) {
      const int i = ht + TPF * r;
      double v = sx[r] * (omgs[i] * omgs[i]);
      if (f == TSFF_FEATURE_ELE && S.filt) v *= S.filt[i];
      xs[H + i] = v * invG;
    }
  } else {
    for (int i = ht; i < npts; i += TPF) {
      const int q = i - toff;           // >= 0: toff = -dmax <= 0
      int pb = q / ppp;
      double sv = 0.0;
      for (int t = q - pb * ppp; t < nh; t += ppp, --pb) sv += taps[t] * ybs[Hb + pb];
      sv *= omgs[i] * omgs[i];
      if (f == TSFF_FEATURE_ELE && S.filt) sv *= S.filt[i];
      xs[H + i] = sv * invG;
    }
  }
  __syncthreads();

  // ================= reverse sweep: recompute each point, accumulate lineout-scalar adjoints =================
  constexpr int NPk = TSFF_NP(NI);
  constexpr int NLB = 9 + 3 * NI;               // adjoint-carrying fields of LineS
  double* gsum = m.red + 8 * kNP_MAX;           // [2][NPk] physical-parameter adjoints of the two features
  if (ht < NPk) gsum[half * NPk + ht] = 0.0;
  const int wv = tid >> 6;
  FeAcc fa;
  fe_acc_init(fa);
  if (GM == 2) {  // table adjoints live where the tangent tables of GM == 1 would (same LDS budget)
    T.Wb = m.Wm; T.Hy = reinterpret_cast<double*>(m.hcm); T.Hs = T.Hy + S.nvx;
    for (int i = tid; i < kNXi2; i += blockDim.x) T.Wb[i] = 0.0;
    for (int i = tid; i < 2 * S.nvx; i += blockDim.x) T.Hy[i] = 0.0;
    __syncthreads();
  }
  for (int g = 0; g < G; ++g) {
    LineS<NI> L, LB;
    {
      Phys<NI> p;
      phys_from_lds<NI>(m.phys, p);
      make_lines_uniform<NI>(p, lam_shift, g, G, L);
    }
    zero_lines<NI>(LB);
    if (use_ks && G > 1) {  // (with one gradient point the cache of the forward sweep is still valid)
      __syncthreads();
      for (int i = ht; i < npts; i += TPF) ksc[i] = ks_eval(omgs[i], L.wpe2);
      __syncthreads();
    }
    for (int st = ht; st < nstrips; st += TPF) {
      const int j0 = kStrip * st;
      const double ws_first = omgs[j0], ws_second = omgs[min(j0 + 1, npts - 1)];
      for (int a = 0; a < NA; ++a) {
        const double ct = uni(m.cosa[a]), wa = uni(m.wsa[a] * L.pref);
        double ws = ws_first, wnext = ws_second;
        Base b0;
        base_eval<NI>(ws, use_ks ? ksc[j0] : ks_eval(ws, L.wpe2), ct, L, T, b0);
        double cxe = 0.0, cF = 0.0;
        TSFF_UNROLL(TSFF_QUNROLL)
        for (int q = 0; q < kStrip; ++q) {
          const int j = j0 + q;
          const bool has_next = (j + 1) < npts;
          const double wsn = wnext;
          wnext = omgs[min(j + 2, npts - 1)];
          Base b1;
          base_eval<NI>(wsn, use_ks ? ksc[min(j + 1, npts - 1)] : ks_eval(wsn, L.wpe2), ct, L, T, b1);
          BaseAdj ba;
          double xen, Fn;
          point_reverse<NI, GM>(b0, b1, has_next, L, T, xs[XA(j)] * wa, ba, xen, Fn, LB, fa);
          ba.xe += cxe; ba.F += cF;
          base_reverse<NI, GM>(ct, b0, L, T, ba, LB, fa);
          cxe = xen; cF = Fn;
          b0 = b1;
          ws = wsn;
        }
        if (j0 + kStrip < npts) {  // the strip's right neighbour receives the D-coupling of the last point
          BaseAdj ba;
          ba.k2 = ba.ik = ba.wd = 0.0; ba.xe = cxe; ba.F = cF;
          base_reverse<NI, GM>(ct, b0, L, T, ba, LB, fa);
        }
      }
    }
    // ---- reduce the lineout-scalar adjoints over the feature's 4 wavefronts; one thread per feature
    //      chains them to the physical parameters (make_lines_adjoint) ----
    {
      lines_adjoint_finalize<NI>(L, LB);   // deferred wavefront-uniform factors of point_reverse
      double lb[NLB];
      lb[0] = LB.wpe2; lb[1] = LB.wL; lb[2] = LB.kL; lb[3] = LB.ivTe; lb[4] = LB.a_e; lb[5] = LB.pref; lb[6] = LB.Ud; lb[7] = LB.Vd;
#pragma unroll
      for (int s = 0; s < NI; ++s) { lb[8 + 3 * s] = LB.ixi[s]; lb[9 + 3 * s] = LB.a_i[s]; lb[10 + 3 * s] = LB.cs[s]; }
      lb[NLB - 1] = LB.m;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NLB; ++k) {
        const double v = wave_sum(lb[k]);
        if (lane == 0) m.red[wv * NLB + k] = v;
      }
      __syncthreads();
      if (ht == 0) {
        const double* r = m.red + (half * NW) * NLB;
#pragma unroll
        for (int k = 0; k < NLB; ++k) {
          double v = (r[k] + r[NLB + k]) + (r[2 * NLB + k] + r[3 * NLB + k]);
          if (NW == 8) v += (r[4 * NLB + k] + r[5 * NLB + k]) + (r[6 * NLB + k] + r[7 * NLB + k]);
          lb[k] = v;
        }
        LB.wpe2 = lb[0]; LB.wL = lb[1]; LB.kL = lb[2]; LB.ivTe = lb[3]; LB.a_e = lb[4]; LB.pref = lb[5]; LB.Ud = lb[6]; LB.Vd = lb[7];
#pragma unroll
        for (int s = 0; s < NI; ++s) { LB.ixi[s] = lb[8 + 3 * s]; LB.a_i[s] = lb[9 + 3 * s]; LB.cs[s] = lb[10 + 3 * s]; }
        LB.m = lb[NLB - 1];
        Phys<NI> p;
        phys_from_lds<NI>(m.phys, p);
        double pbar[NPk];
#pragma unroll
        for (int s = 0; s < NPk; ++s) pbar[s] = 0.0;
        make_lines_adjoint<NI>(p, lam_shift, g, G, L, LB, pbar);
#pragma unroll
        for (int s = 0; s < NPk; ++s) gsum[half * NPk + s] += pbar[s];
      }
    }
  }
  if (GM == 2) {  // table adjoints of this lineout -> global (k_fe_adjoint chains them to f_e)
    fe_flush_w(fa, T.Wb);
    fe_flush_h(fa, T.Hy, T.Hs);
    __syncthreads();
    for (int i = tid; i < kNXi2; i += blockDim.x) {
      double* o = K.Wb_out + (size_t)b * kNXi2 + i;
      *o = accumulate ? *o + T.Wb[i] : T.Wb[i];
    }
    for (int i = tid; i < S.nvx; i += blockDim.x) {
      double* oy = K.Hy_out + (size_t)b * S.nvx + i;
      double* os = K.Hs_out + (size_t)b * S.nvx + i;
      *oy = accumulate ? *oy + T.Hy[i] : T.Hy[i];
      *os = accumulate ? *os + T.Hs[i] : T.Hs[i];
    }
  }
  // amplitudes (irf.py:76,126-130)
  if (ht == 0) {
    if (f == TSFF_FEATURE_ELE) { gsum[half * NPk + TSFF_P_AMP1] += a1b; gsum[half * NPk + TSFF_P_AMP2] += a2b; }
    else gsum[half * NPk + TSFF_P_AMP3] += a1b;
  }
  __syncthreads();
  // ---- feature sum and the chain rule to the normalised leaves (Ti tying, fraction renormalisation,
  //      activation; ts_params.py:329-350, 543-563) ----
  if (tid == 0) {
    if (nfeat > 1)
      for (int s = 0; s < NPk; ++s) gsum[s] += gsum[NPk + s];
    Phys<NI> p;
    phys_from_lds<NI>(m.phys, p);
#pragma unroll
    for (int s = 1; s < NI; ++s)
      if (S.ti_same[s]) {
        gsum[TSFF_P_ION0 + TSFF_ION_TI] += gsum[TSFF_P_ION0 + 4 * s + TSFF_ION_TI];
        gsum[TSFF_P_ION0 + 4 * s + TSFF_ION_TI] = 0.0;
      }
    double dot = 0.0;
#pragma unroll
    for (int s = 0; s < NI; ++s) dot += gsum[TSFF_P_ION0 + 4 * s + TSFF_ION_FRACT] * p.fr[s];
#pragma unroll
    for (int s = 0; s < NI; ++s) {
      const int o = TSFF_P_ION0 + 4 * s + TSFF_ION_FRACT;
      gsum[o] = (gsum[o] - dot) / p.fsum;
      gsum[TSFF_P_ION0 + 4 * s + TSFF_ION_A] = 0.0;
    }
    if (GM != 1) gsum[TSFF_P_M] = 0.0;
  }
  __syncthreads();
  if (tid < NPk) {
    const double xv = xpar[tid];
    double v = gsum[tid] * S.p_scale[tid];
    if (S.p_sig[tid]) { const double sg = sigmoid(xv); v *= sg * (1.0 - sg); }
    v = gmask[tid] ? v : 0.0;
    if (interleaved) K.gpart[((size_t)f * K.B + b) * NPk + tid] = v;  // summed over the two features by k_loss_reduce
    else grad[(size_t)b * NPk + tid] = accumulate ? grad[(size_t)b * NPk + tid] + v : v;
  }
}

// deterministic reduction of lpart[B][3] -> out[3] by workgroup 0 in a fixed order; with gpart (interleaved plan) every
// workgroup also adds the two per-feature gradient parts: grad[i] = gpart[0][i] + gpart[1][i], i < n
__global__ __launch_bounds__(kThreads) void k_loss_reduce(const double* __restrict__ lpart, int B, double* __restrict__ out,
                                                          const double* __restrict__ gpart, long n, double* __restrict__ grad) {
  __shared__ double red[8];
  if (gpart) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) grad[i] = gpart[i] + gpart[n + i];
  }
  if (blockIdx.x != 0) return;
  double a[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < B; b += kThreads) {
    a[0] += lpart[(size_t)b * 3 + 0]; a[1] += lpart[(size_t)b * 3 + 1]; a[2] += lpart[(size_t)b * 3 + 2];
  }
  for (int k = 0; k < 3; ++k) {
    const double v = block_sum(a[k], red);
    if (threadIdx.x == 0) out[k] = v;
  }
}

// ------------------------------------------------------------------------------------------
// k_form_factor: raw FormFactor.__call__ (form_factor.py:163-298) -> P[b][g][j][a], physical
// parameters in, no instrument chain.  One workgroup per lineout; any npts.
// ------------------------------------------------------------------------------------------
template <int NI>
__global__ __launch_bounds__(kThreads) void k_form_factor(KStatic S, KCall K, int f, const double* __restrict__ omgs,
                                                          int npts, double* __restrict__ P) {
  const int b = blockIdx.x, tid = threadIdx.x;
  extern __shared__ __align__(16) unsigned char smem[];
  const Smem m = carve(smem, S, 1, false, false);
  Tables T;
  load_tables(m, S, K, S.shared_fe ? 0 : b, false, T);
  Phys<NI> p;
  load_phys<NI>(K.params + (size_t)b * S.NP, S.p_scale, S.p_shift, S.p_sig, S.ti_same, false, p);
  __syncthreads();
  const int G = S.G, NA = S.n_angles;
  const int nstrips = (npts + kStrip - 1) / kStrip;
  for (int st = tid; st < nstrips; st += kThreads) {
    const int j0 = st * kStrip;
    double ws[kStrip + 1];
#pragma unroll
    for (int q = 0; q <= kStrip; ++q) ws[q] = omgs[min(j0 + q, npts - 1)];
    for (int g = 0; g < G; ++g) {
      LineS<NI> L;
      make_lines<NI>(p, S.lam_shift[f], g, G, L);
      double ksv[kStrip + 1];
#pragma unroll
      for (int q = 0; q <= kStrip; ++q) ksv[q] = ks_eval(ws[q], L.wpe2);
      for (int a = blockIdx.y; a < NA; a += gridDim.y) {   // (few lineouts, many angles: the angles are spread over blockIdx.y)
        const double ct = m.cosa[a];
        Base b0;
        base_eval<NI>(ws[0], ksv[0], ct, L, T, b0);
#pragma unroll
        for (int q = 0; q < kStrip; ++q) {
          const int j = j0 + q;
          const bool has_next = (j + 1) < npts;
          Base b1;
          base_eval<NI>(ws[q + 1], ksv[q + 1], ct, L, T, b1);
          if (j < npts) P[(((size_t)b * G + g) * npts + j) * NA + a] = point_forward<NI>(ws[q], b0, b1, has_next, L, T);
          b0 = b1;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_form_factor_adj: reverse of k_form_factor for an arbitrary seed Pbar[b][g][j][a] (the angular instrument chain hands
// one back per (wavelength, angle) point, unlike the fit path whose seed is a spectrum adjoint times the angle
// weights).  The reverse sweep of k_spectrum with the same device functions: every point is recomputed and reversed,
// the lineout-scalar adjoints are reduced per workgroup and added to LBacc[b][g][:] (k_ff_lines_adj finishes the chain to
// the physical parameters), and with GM == 2 the adjoints of the two distribution-function tables are gathered in LDS
// and added to Wb_out / Hy_out / Hs_out (k_wgemm_t and k_fe_adjoint chain them to f_e, as for tsff_loss_grad_fe).
// Grid (B, angle chunks).
// ------------------------------------------------------------------------------------------
template <int NI, int GM>
__global__ __launch_bounds__(kThreads) void k_form_factor_adj(KStatic S, KCall K, int f, const double* __restrict__ omgs,
                                                              int npts, const double* __restrict__ Pbar,
                                                              double* __restrict__ LBacc) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  extern __shared__ __align__(16) unsigned char smem[];
  const Smem m = carve(smem, S, 1, GM != 0, false);
  Tables T;
  load_tables(m, S, K, S.shared_fe ? 0 : b, false, T);
  Phys<NI> p;
  load_phys<NI>(K.params + (size_t)b * S.NP, S.p_scale, S.p_shift, S.p_sig, S.ti_same, false, p);
  if (GM == 2) {
    T.Wb = m.Wm; T.Hy = reinterpret_cast<double*>(m.hcm); T.Hs = T.Hy + S.nvx;
    for (int i = tid; i < kNXi2; i += kThreads) T.Wb[i] = 0.0;
    for (int i = tid; i < 2 * S.nvx; i += kThreads) T.Hy[i] = 0.0;
  }
  __syncthreads();
  constexpr int NLB = 9 + 3 * NI;
  const int G = S.G, NA = S.n_angles;
  const int nstrips = (npts + kStrip - 1) / kStrip;
  FeAcc fa;
  fe_acc_init(fa);
  for (int g = 0; g < G; ++g) {
    LineS<NI> L, LB;
    make_lines<NI>(p, S.lam_shift[f], g, G, L);
    zero_lines<NI>(LB);
    for (int st = tid; st < nstrips; st += kThreads) {
      const int j0 = st * kStrip;
      double ws[kStrip + 1], ksv[kStrip + 1];
#pragma unroll
      for (int q = 0; q <= kStrip; ++q) { ws[q] = omgs[min(j0 + q, npts - 1)]; ksv[q] = ks_eval(ws[q], L.wpe2); }
      for (int a = blockIdx.y; a < NA; a += gridDim.y) {
        const double ct = m.cosa[a];
        Base b0;
        base_eval<NI>(ws[0], ksv[0], ct, L, T, b0);
        double cxe = 0.0, cF = 0.0;
#pragma unroll
        for (int q = 0; q < kStrip; ++q) {
          const int j = j0 + q;
          const bool has_next = (j + 1) < npts;
          Base b1;
          base_eval<NI>(ws[q + 1], ksv[q + 1], ct, L, T, b1);
          if (j < npts) {
            BaseAdj ba;
            double xen, Fn;
            point_reverse<NI, GM>(b0, b1, has_next, L, T,
                                  Pbar[(((size_t)b * G + g) * npts + j) * NA + a] * (L.pref * ws[q] * ws[q]), ba, xen, Fn, LB, fa);
            ba.xe += cxe; ba.F += cF;
            base_reverse<NI, GM>(ct, b0, L, T, ba, LB, fa);
            cxe = xen; cF = Fn;
          }
          b0 = b1;
        }
        if (j0 + kStrip < npts) {  // the strip's right neighbour receives the D-coupling of the last point
          BaseAdj ba;
          ba.k2 = ba.ik = ba.wd = 0.0; ba.xe = cxe; ba.F = cF;
          base_reverse<NI, GM>(ct, b0, L, T, ba, LB, fa);
        }
      }
    }
    lines_adjoint_finalize<NI>(L, LB);   // deferred wavefront-uniform factors of point_reverse
    double lb[NLB];
    lb[0] = LB.wpe2; lb[1] = LB.wL; lb[2] = LB.kL; lb[3] = LB.ivTe; lb[4] = LB.a_e; lb[5] = LB.pref; lb[6] = LB.Ud; lb[7] = LB.Vd;
#pragma unroll
    for (int s = 0; s < NI; ++s) { lb[8 + 3 * s] = LB.ixi[s]; lb[9 + 3 * s] = LB.a_i[s]; lb[10 + 3 * s] = LB.cs[s]; }
    lb[NLB - 1] = LB.m;
#pragma unroll
    for (int k = 0; k < NLB; ++k) {
      const double v = wave_sum(lb[k]);
      if (lane == 0) atomicAdd(LBacc + ((size_t)b * G + g) * NLB + k, v);
    }
  }
  if (GM == 2) {
    fe_flush_w(fa, T.Wb);
    fe_flush_h(fa, T.Hy, T.Hs);
    __syncthreads();
    for (int i = tid; i < kNXi2; i += kThreads)
      if (T.Wb[i] != 0.0) atomicAdd(K.Wb_out + (size_t)b * kNXi2 + i, T.Wb[i]);
    for (int i = tid; i < S.nvx; i += kThreads) {
      if (T.Hy[i] != 0.0) atomicAdd(K.Hy_out + (size_t)b * S.nvx + i, T.Hy[i]);
      if (T.Hs[i] != 0.0) atomicAdd(K.Hs_out + (size_t)b * S.nvx + i, T.Hs[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_form_factor_2d: FormFactor.calc_in_2D (form_factor.py:449-587) for a 2-D distribution function
// fe2d[nv][nv] (first index = v_x): one workgroup per (lineout, gradient point, wavelength, angle).
//   1. point scalars: vector k = k_s - k_L, omega_d, xi_i, chi_i, vector xi_e -> (|xi_e|, beta)   (:515-558)
//   2. rotate(fe2d, beta) + column sum (:300-324, 371): thread iy evaluates the bicubic interpolant along the
//      rotated line {R_beta (vx[ix], vx[iy])}, ix = 0..nv-1.  interpax's "cubic" 2-D interpolant (bicubic Hermite
//      patch, derivative estimates = mean of adjacent secants, one-sided at the edges, extrap=True) is the tensor
//      product of the 1-D Hermite interpolants, i.e. a 4 x 4 stencil with separable weights (Catmull-Rom in the
//      interior, modified in the first/last cell).
//   3. gradient of the projection, the two linear interpolations at |xi_e| and the rationally-centred integral
//      over the nv-2 intervals (:372-387), spectrum assembly (:560-585).
// ------------------------------------------------------------------------------------------
// interior-cell (Catmull-Rom) weights of the four nodes c-1..c+2
__device__ __forceinline__ void catmull_rom(double t, double w[4]) {
  const double t2 = t * t, t3 = t2 * t;
  const double h10 = t3 - 2.0 * t2 + t, h11 = t3 - t2;
  w[0] = -0.5 * h10;
  w[1] = (2.0 * t3 - 3.0 * t2 + 1.0) - 0.5 * h11;
  w[2] = (-2.0 * t3 + 3.0 * t2) + 0.5 * h10;
  w[3] = 0.5 * h11;
}

// One bicubic sample at (xq, yq) of a table that carries one GHOST row / column on every side, filled by linear
// extrapolation (f[-1] = 2 f[0] - f[1], f[n] = 2 f[n-1] - f[n-2]; corners by both).  With those ghosts the Catmull-Rom
// weights of an interior cell reproduce interpax's edge cells exactly -- its one-sided node slope f[1] - f[0] IS the
// central slope (f[1] - f[-1]) / 2 of the extended sequence -- and its extrapolation (edge-cell polynomial continued
// outside the grid, extrap=True) is the same formula with the cell index clamped and t left free.  Every sample then
// takes one straight-line path: 4 x 4 consecutive entries, no index clamps, no divergence between lanes.
// Fp points at the ghost corner; pitch = row pitch in doubles of the padded table.
// (u, v): the sample position in CELL units, u = (xq - vx[0]) / dv.  Along a rotated line the callers advance it as
// u = u0 + ix cos(beta), v = v0 + ix sin(beta) -- one FMA per axis instead of rotating, shifting and scaling every sample.
__device__ __forceinline__ void cell_of(double u, int nv, int& c, double& t) {
  c = (int)floor(u);
  c = c < 0 ? 0 : (c > nv - 2 ? nv - 2 : c);
  t = u - (double)c;   // free outside the grid: the edge-cell polynomial continued (extrap=True)
}
__device__ __forceinline__ double bicubic_sample(const double* __restrict__ Fp, int nv, int pitch, double u, double v) {
  int cx, cy;
  double tx, ty, wx[4], wy[4];
  cell_of(u, nv, cx, tx);
  cell_of(v, nv, cy, ty);
  catmull_rom(tx, wx);
  catmull_rom(ty, wy);
  const double* __restrict__ q0 = Fp + (size_t)cx * pitch + cy;  // padded index of node (cx - 1, cy - 1)
  double r = 0.0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double* __restrict__ row = q0 + (size_t)m * pitch;
    r += wx[m] * (wy[0] * row[0] + wy[1] * row[1] + wy[2] * row[2] + wy[3] * row[3]);
  }
  return r;
}
// two-stage form of the sampler for tables read through L1/L2: the 16 stencil values of the NEXT sample are requested
// before the current one is contracted, so that a wavefront keeps loads in flight while it computes (203 -> 190 ms at
// 256^2; with the table in LDS the same change loses 8 %: there the LDS pipe, not its latency, is the limit)
__device__ __forceinline__ void bicubic_fetch(const double* __restrict__ Fp, int pitch, int cx, int cy, double V[16]) {
  const double* __restrict__ q0 = Fp + (size_t)cx * pitch + cy;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double* __restrict__ row = q0 + (size_t)m * pitch;
    V[4 * m] = row[0]; V[4 * m + 1] = row[1]; V[4 * m + 2] = row[2]; V[4 * m + 3] = row[3];
  }
}
__device__ __forceinline__ double bicubic_dot(const double V[16], double tx, double ty) {
  double wx[4], wy[4];
  catmull_rom(tx, wx);
  catmull_rom(ty, wy);
  double r = 0.0;
#pragma unroll
  for (int m = 0; m < 4; ++m) r += wx[m] * (wy[0] * V[4 * m] + wy[1] * V[4 * m + 1] + wy[2] * V[4 * m + 2] + wy[3] * V[4 * m + 3]);
  return r;
}
// cell coordinates of sample (ix, iy) of the grid rotated by (cb, sb): u = ix cb + u0, v = ix sb + v0
__device__ __forceinline__ void line_origin(double cb, double sb, double y, double vx0, double idv, double& u0, double& v0c) {
  u0 = (vx0 * cb - y * sb - vx0) * idv;
  v0c = (vx0 * sb + y * cb - vx0) * idv;
}

// ghost cells of a padded table P[(nv + 2)][pitch] whose interior [1..nv][1..nv] is filled: rows first, then columns
// (which also makes the corners).  Called by all threads of a workgroup (LDS) or of a grid (global copy).
__device__ __forceinline__ void ghost_rows(double* __restrict__ P, int nv, int pitch, int tid, int nthr) {
  for (int c = tid; c < nv; c += nthr) {
    P[c + 1] = 2.0 * P[pitch + c + 1] - P[2 * pitch + c + 1];
    P[(size_t)(nv + 1) * pitch + c + 1] = 2.0 * P[(size_t)nv * pitch + c + 1] - P[(size_t)(nv - 1) * pitch + c + 1];
  }
}
__device__ __forceinline__ void ghost_cols(double* __restrict__ P, int nv, int pitch, int tid, int nthr) {
  for (int r = tid; r < nv + 2; r += nthr) {
    double* row = P + (size_t)r * pitch;
    row[0] = 2.0 * row[1] - row[2];
    row[nv + 1] = 2.0 * row[nv] - row[nv - 1];
  }
}

// padded copy in global memory for tables that do not fit LDS: one workgroup per table
__global__ __launch_bounds__(kThreads) void k_pad2d(const double* __restrict__ F, int nv, double* __restrict__ P) {
  const int pitch = nv + 2;
  const double* __restrict__ Fb = F + (size_t)blockIdx.x * nv * nv;
  double* __restrict__ Pb = P + (size_t)blockIdx.x * (nv + 2) * pitch;
  for (int i = threadIdx.x; i < nv * nv; i += kThreads) Pb[(size_t)(i / nv + 1) * pitch + (i % nv + 1)] = Fb[i];
  __syncthreads();
  ghost_rows(Pb, nv, pitch, threadIdx.x, kThreads);
  __syncthreads();
  ghost_cols(Pb, nv, pitch, threadIdx.x, kThreads);
}

// LDS: true -> the nv x nv table is staged once per (persistent) workgroup in LDS (nv <= 128: 128 KB); false -> the
// table is read through L1/L2 (any nv).  A workgroup is kG2 = 4 groups of 256 threads that work on four different
// points at once and share the table: 4 wavefronts per SIMD hide the latency of the 16 table reads per sample (one
// 256-thread workgroup per CU, all the 128 KB table allows, leaves one wavefront per SIMD waiting on LDS).  The
// per-point scalars are computed by one thread per group and passed through LDS so that the sampling loop stays
// within the 128 registers of a 1024-thread workgroup.  One table per launch (the host loops over lineouts when every
// lineout has its own).
#ifndef TSFF_2D_GROUPS_LDS
#define TSFF_2D_GROUPS_LDS 4
#endif
#ifndef TSFF_2D_GROUPS_L2
#define TSFF_2D_GROUPS_L2 1
#endif
constexpr int kSc2 = 40;  // doubles of per-group scalar scratch
// Row pitch of the padded (nv + 2)^2 table.  In LDS it is made odd: with an even (worse: power-of-two) pitch the rows
// start in the same banks and the lanes of a wavefront (neighbouring points of a rotated line) collide whenever the line
// runs along the first table axis.
__host__ __device__ inline int pitch2d(int nv, bool lds) {
  if (!lds) return nv + 2;
  // LDS bank model: the 32 lanes of a half wavefront read one double each; doubles a != b collide when a = b (mod 32).
  // The lanes are neighbouring samples of a rotated line, i.e. a digital straight line of cells, and with
  // bank = (pitch cx + cy) mod 32 the passes per read depend on pitch mod 32 (simulated over all directions / measured
  // by SQ_LDS_BANK_CONFLICT): 1 -> 1.49, 2 -> 1.78 / 1.8, 3 -> 1.97 / 1.96, growing to 2.4 at 10.  1 where the LDS
  // budget allows it (nv <= 96), else 2.
  const int r = nv <= 96 ? 1 : 2;
  return nv + 2 + ((r - (nv + 2) % 32) + 32) % 32;
}
// per-group scratch: f1, d1 [nv], part [nparts][nv] (nparts = 4, 2, 1 for nv <= 64, 128, larger), red [8], scalars
__host__ __device__ inline size_t group2d_doubles(int nv) {
  const int nparts = nv <= 64 ? 4 : (nv <= 128 ? 2 : 1);
  return (2 + (size_t)nparts) * nv + 8 + kSc2;
}
__host__ __device__ inline size_t smem2d_doubles(int nv, bool lds, int ng) {
  return (size_t)ng * group2d_doubles(nv) + (lds ? (size_t)(nv + 2) * pitch2d(nv, true) : 0);
}
template <int NI, bool LDS, int kG2>
__global__ __launch_bounds__(kG2 * kThreads) void k_form_factor_2d(KStatic S, const double* __restrict__ phys,
                                                                   const double* __restrict__ Fg, int nv,
                                                                   double ud_ang, double va_ang, int f, long pbegin,
                                                                   long pend, double* __restrict__ P) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int grp = threadIdx.x >> 8, gt = threadIdx.x & (kThreads - 1);
  // thread -> (column iy, part of the ix range): nparts = 256 / nvp with nvp = nv rounded up to 64, 128 or 256
  const int nvp = nv <= 64 ? 64 : (nv <= 128 ? 128 : 256);
  const int nparts = nv <= 256 ? kThreads / nvp : 1;
  double* gbase = reinterpret_cast<double*>(smem) + (size_t)grp * group2d_doubles(nv);
  double* f1 = gbase;              // [nv] projected distribution
  double* d1 = f1 + nv;            // [nv] its gradient
  double* part = d1 + nv;          // [nparts][nv] partial column sums
  double* red = part + (nv <= 64 ? 4 : (nv <= 128 ? 2 : 1)) * nv;  // [8]
  double* sc = red + 8;            // [kSc2] point scalars
  double* Fl = reinterpret_cast<double*>(smem) + (size_t)kG2 * group2d_doubles(nv);  // padded table (LDS variant)
  const int NA = S.n_angles, G = S.G, npts = S.npts;
  const double dv = 12.0 / nv, v0 = -6.0 + 0.5 * dv, idv = 1.0 / dv;  // base.py:333-335
  const int pitch = pitch2d(nv, LDS);
  if (LDS) {  // Fg: the plain nv x nv table; the ghost cells are made here
    for (int i = threadIdx.x; i < nv * nv; i += kG2 * kThreads) Fl[(i / nv + 1) * pitch + (i % nv + 1)] = Fg[i];
    __syncthreads();
    ghost_rows(Fl, nv, pitch, threadIdx.x, kG2 * kThreads);
    __syncthreads();
    ghost_cols(Fl, nv, pitch, threadIdx.x, kG2 * kThreads);
  }
  const double* __restrict__ F = LDS ? Fl : Fg;  // (not LDS: Fg is the padded copy made by k_pad2d)
  const long stride = (long)gridDim.x * kG2;
  for (long base = pbegin + (long)blockIdx.x * kG2; base < pend; base += stride) {
    const long pid = base + grp;
    const bool active = pid < pend;  // every group runs the same barrier sequence; idle groups skip the work
    __syncthreads();
    if (active && gt == 0) {
      const int a = (int)(pid % NA), j = (int)((pid / NA) % npts), g = (int)((pid / ((long)NA * npts)) % G);
      const int b = (int)(pid / ((long)NA * npts * G));
      Phys<NI> p;
      load_phys<NI>(phys + (size_t)b * S.NP, S.p_scale, S.p_shift, S.p_sig, S.ti_same, false, p);
      LineS<NI> L;
      make_lines<NI>(p, S.lam_shift[f], g, G, L);
      // ---- point scalars (form_factor.py:515-558) ----
      const double ws = S.omgs[f][j], th = S.sa_rad[a];
      const double ks = ks_eval(ws, L.wpe2);
      const double kx = cos(th) * ks - L.kL, ky = sin(th) * ks;
      const double k2 = kx * kx + ky * ky, k = sqrt(k2);
      const double Vx = L.Vd * cos(va_ang), Vy = L.Vd * sin(va_ang);
      const double Ux = L.Ud * cos(ud_ang), Uy = L.Ud * sin(ud_ang);
      const double wd = (ws - L.wL) - (kx * Vx + ky * Vy);
      const double aa = wd / k2;
      const double xex = (aa * kx - Ux) * L.ivTe, xey = (aa * ky - Uy) * L.ivTe;
      const double beta = atan(xey / xex) + (xex >= 0.0 ? 0.0 : kPi);   // heaviside(x, 1) = 1 at x == 0
      sc[0] = cos(beta); sc[1] = sin(beta); sc[2] = sqrt(xex * xex + xey * xey);
      sc[3] = k2; sc[4] = k; sc[5] = wd; sc[6] = ws; sc[7] = L.a_e; sc[8] = L.ivTe; sc[9] = L.wL; sc[10] = L.pref;
#pragma unroll
      for (int s = 0; s < NI; ++s) { sc[12 + 3 * s] = L.ixi[s]; sc[13 + 3 * s] = L.a_i[s]; sc[14 + 3 * s] = L.cs[s]; }
    }
    __syncthreads();
    const double cb = sc[0], sb = sc[1], xmag = sc[2];
    // ---- rotate + project (:300-324, 371) ----
    if (active) {
      if (nv <= 256) {
        const int iy = gt % nvp, pt = gt / nvp;
        if (iy < nv) {
          const int ix0 = (nv * pt) / nparts, ix1 = (nv * (pt + 1)) / nparts;
          const double y = v0 + iy * dv;
          double acc = 0.0, ul, vl, xi_d = (double)ix0;
          line_origin(cb, sb, y, v0, idv, ul, vl);
          if (LDS) {
            for (int ix = ix0; ix < ix1; ++ix, xi_d += 1.0)
              acc += bicubic_sample(F, nv, pitch, __builtin_fma(xi_d, cb, ul), __builtin_fma(xi_d, sb, vl));
          } else {
            int cx, cy;
            double tx, ty, V[16];
            cell_of(__builtin_fma(xi_d, cb, ul), nv, cx, tx);
            cell_of(__builtin_fma(xi_d, sb, vl), nv, cy, ty);
            bicubic_fetch(F, pitch, cx, cy, V);
            for (int ix = ix0; ix < ix1; ++ix) {
              xi_d += 1.0;   // (the fetch after the last sample repeats a valid position: no branch in the loop)
              const double xn = ix + 1 < ix1 ? xi_d : xi_d - 1.0;
              int ncx, ncy;
              double ntx, nty, N[16];
              cell_of(__builtin_fma(xn, cb, ul), nv, ncx, ntx);
              cell_of(__builtin_fma(xn, sb, vl), nv, ncy, nty);
              bicubic_fetch(F, pitch, ncx, ncy, N);
              __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the arithmetic on the previous sample
              acc += bicubic_dot(V, tx, ty);
#pragma unroll
              for (int k = 0; k < 16; ++k) V[k] = N[k];
              tx = ntx; ty = nty;
            }
          }
          part[pt * nv + iy] = acc;
        }
      } else {
        for (int iy = gt; iy < nv; iy += kThreads) {
          const double y = v0 + iy * dv;
          double acc = 0.0, ul, vl, xi_d = 0.0;
          line_origin(cb, sb, y, v0, idv, ul, vl);
          for (int ix = 0; ix < nv; ++ix, xi_d += 1.0)
            acc += bicubic_sample(F, nv, pitch, __builtin_fma(xi_d, cb, ul), __builtin_fma(xi_d, sb, vl));
          part[iy] = acc;
        }
      }
    }
    __syncthreads();
    if (active) {
      for (int i = gt; i < nv; i += kThreads) {
        double sacc = 0.0;
        for (int q = 0; q < nparts; ++q) sacc += part[q * nv + i];
        f1[i] = sacc * dv;
      }
    }
    __syncthreads();
    if (active) {
      for (int i = gt; i < nv; i += kThreads) {
        double gd;
        if (i == 0) gd = (f1[1] - f1[0]) * idv;
        else if (i == nv - 1) gd = (f1[nv - 1] - f1[nv - 2]) * idv;
        else gd = (f1[i + 1] - f1[i - 1]) * (0.5 * idv);
        d1[i] = gd;
      }
    }
    __syncthreads();
    // ---- ratintn(df, vx - |xi_e|, vx): nv - 2 intervals (:372-387) ----
    double psum = 0.0;
    if (active) {
      for (int i = gt; i < nv - 2; i += kThreads) {
        const double f0 = d1[i], f1v = d1[i + 1];
        const double g0 = (v0 + i * dv) - xmag, g1 = (v0 + (i + 1) * dv) - xmag;
        const double fdif = f1v - f0, gdif = g1 - g0, fav = 0.5 * (f1v + f0), gav = 0.5 * (g1 + g0);
        const double tmp = fav * gdif - gav * fdif;
        double r;
        if (fabs(gdif) < 1.0e-4 * fabs(gav)) r = fav / gav + tmp * gdif / (12.0 * gav * gav * gav);
        else r = fdif / gdif + tmp * log(fabs((gav + 0.5 * gdif) / (gav - 0.5 * gdif))) / (gdif * gdif);
        psum += r * dv;
      }
    }
    psum = wave_sum(psum);
    if ((gt & 63) == 0) red[gt >> 6] = psum;
    __syncthreads();
    if (active && gt == 0) {
      const double R = (red[0] + red[1]) + (red[2] + red[3]);
      const double k2 = sc[3], k = sc[4], wd = sc[5], ws = sc[6], a_e = sc[7], ivTe = sc[8], wL = sc[9], pref = sc[10];
      // jnp.interp(|xi_e|, vx, .): clamps to the end values
      double u = (xmag - v0) * idv;
      int i = (int)u;
      i = i < 0 ? 0 : (i > nv - 2 ? nv - 2 : i);
      double t = (xmag - (v0 + i * dv)) * idv;
      t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
      const double fe_vphi = f1[i] + t * (f1[i + 1] - f1[i]);
      const double dfe = d1[i] + t * (d1[i + 1] - d1[i]);
      const double ike2 = a_e / k2;
      const double cer = -ike2 * R, cei = kPi * ike2 * dfe;
      double cre = 0.0, cim = 0.0, gsum = 0.0;
      const double vph = wd / k;
#pragma unroll
      for (int s = 0; s < NI; ++s) {
        const double xi = vph * sc[12 + 3 * s];
        double zr, zi, dzr, dzi, gs;
        ion_terms(S.zp, xi, zr, zi, dzr, dzi, gs);   // (one thread per point: the Z' table is read from global memory)
        const double iki2 = sc[13 + 3 * s] / k2;
        cre -= 0.5 * iki2 * zr;
        cim -= 0.5 * iki2 * zi;
        gsum += sc[14 + 3 * s] * gs;
      }
      const double er = 1.0 + cer + cre, ei = cei + cim;
      const double eps2 = er * er + ei * ei, ce2 = cer * cer + cei * cei;
      const double ci2 = (1.0 + cre) * (1.0 + cre) + cim * cim;
      const double Sv = (gsum * ce2 + ci2 * fe_vphi * ivTe) / (k * eps2);
      P[pid] = Sv * (1.0 + 2.0 * wd / wL) * pref * ws * ws;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Adjoint of k_form_factor_2d.  Same workgroup structure; per point, after the forward projection f1:
//   one thread reverses the spectrum assembly (-> adjoints of R, fe(v_phi), dfe and of the point scalars),
//   the group reverses ratintn, the two interpolations and the finite-difference gradient (-> f1bar[iy]),
//   a second sampling sweep with derivative weights gives the adjoint of the rotation (cos beta, sin beta),
//   one thread finishes the chain to the lineout scalars (LineS fields) and adds them to LBacc[b][g][:].
// f1bar is also written to global memory for the table adjoint (k_ff2d_table_adj): the scatter into the table needs the
// sample weights only, not the table, so it runs as a separate pass whose LDS holds the table ADJOINT.
// ------------------------------------------------------------------------------------------
constexpr int kNLB2 = 8;  // + 3 per ion: wpe2, wL, kL, ivTe, a_e, pref, Ud, Vd | ixi, a_i, cs

// value and both first derivatives of one bicubic sample (ghost-padded table, see bicubic_sample)
__device__ __forceinline__ void bicubic_sample_grad(const double* __restrict__ Fp, int nv, int pitch, double idv, double u,
                                                    double v, double& Sx, double& Sy) {
  int cx, cy;
  double tx, ty;
  cell_of(u, nv, cx, tx);
  cell_of(v, nv, cy, ty);
  double wx[4], wy[4], dx[4], dy[4];
  catmull_rom(tx, wx);
  catmull_rom(ty, wy);
  dx[0] = -0.5 * (3.0 * tx * tx - 4.0 * tx + 1.0); dx[1] = 4.5 * tx * tx - 5.0 * tx;
  dx[2] = -4.5 * tx * tx + 4.0 * tx + 0.5;         dx[3] = 1.5 * tx * tx - tx;
  dy[0] = -0.5 * (3.0 * ty * ty - 4.0 * ty + 1.0); dy[1] = 4.5 * ty * ty - 5.0 * ty;
  dy[2] = -4.5 * ty * ty + 4.0 * ty + 0.5;         dy[3] = 1.5 * ty * ty - ty;
  const double* __restrict__ q0 = Fp + (size_t)cx * pitch + cy;
  double sx = 0.0, sy = 0.0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double* __restrict__ row = q0 + (size_t)m * pitch;
    const double r = wy[0] * row[0] + wy[1] * row[1] + wy[2] * row[2] + wy[3] * row[3];
    const double rd = dy[0] * row[0] + dy[1] * row[1] + dy[2] * row[2] + dy[3] * row[3];
    sx += dx[m] * r;
    sy += wx[m] * rd;
  }
  Sx = sx * idv;   // d / d xq = (1 / dv) d / du
  Sy = sy * idv;
}

template <int NI, bool LDS, int kG2>
__global__ __launch_bounds__(kG2 * kThreads) void k_form_factor_2d_adj(KStatic S, const double* __restrict__ phys,
                                                                       const double* __restrict__ Fg, int nv, double ud_ang,
                                                                       double va_ang, int f, long pbegin, long pend,
                                                                       const double* __restrict__ Pbar,
                                                                       double* __restrict__ LBacc, double* __restrict__ f1bar_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int grp = threadIdx.x >> 8, gt = threadIdx.x & (kThreads - 1);
  const int nvp = nv <= 64 ? 64 : (nv <= 128 ? 128 : 256);
  const int nparts = nv <= 256 ? kThreads / nvp : 1;
  constexpr int kScA = 64;
  const size_t gsz = (2 + (size_t)(nv <= 64 ? 4 : (nv <= 128 ? 2 : 1))) * nv + 2 * (size_t)nv + 16 + kScA;
  double* gbase = reinterpret_cast<double*>(smem) + (size_t)grp * gsz;
  double* f1 = gbase;              // [nv] projected distribution, later f1bar
  double* d1 = f1 + nv;            // [nv] its gradient, later d1bar
  double* part = d1 + nv;          // [nparts][nv] partial sums (two values per part in the derivative sweep: reused)
  double* c0 = part + (nv <= 64 ? 4 : (nv <= 128 ? 2 : 1)) * nv;  // [nv] per-interval adjoint to d1[i]
  double* c1 = c0 + nv;            // [nv] per-interval adjoint to d1[i + 1]
  double* red = c1 + nv;           // [16]
  double* sc = red + 16;           // [kScA] point scalars and adjoints
  double* Fl = reinterpret_cast<double*>(smem) + (size_t)kG2 * gsz;
  const int NA = S.n_angles, G = S.G, npts = S.npts;
  constexpr int NLB = kNLB2 + 3 * NI;
  const double dv = 12.0 / nv, v0 = -6.0 + 0.5 * dv, idv = 1.0 / dv;
  const int pitch = pitch2d(nv, LDS);
  if (LDS) {
    for (int i = threadIdx.x; i < nv * nv; i += kG2 * kThreads) Fl[(i / nv + 1) * pitch + (i % nv + 1)] = Fg[i];
    __syncthreads();
    ghost_rows(Fl, nv, pitch, threadIdx.x, kG2 * kThreads);
    __syncthreads();
    ghost_cols(Fl, nv, pitch, threadIdx.x, kG2 * kThreads);
  }
  const double* __restrict__ F = LDS ? Fl : Fg;
  const long stride = (long)gridDim.x * kG2;
  for (long base = pbegin + (long)blockIdx.x * kG2; base < pend; base += stride) {
    const long pid = base + grp;
    const bool active = pid < pend;
    const int a = (int)(pid % NA), j = (int)((pid / NA) % npts), g = (int)((pid / ((long)NA * npts)) % G);
    const int b = (int)(pid / ((long)NA * npts * G));
    __syncthreads();
    if (active && gt == 0) {
      Phys<NI> p;
      load_phys<NI>(phys + (size_t)b * S.NP, S.p_scale, S.p_shift, S.p_sig, S.ti_same, false, p);
      LineS<NI> L;
      make_lines<NI>(p, S.lam_shift[f], g, G, L);
      const double ws = S.omgs[f][j], th = S.sa_rad[a];
      const double ks = ks_eval(ws, L.wpe2);
      const double ct = cos(th), st = sin(th);
      const double kx = ct * ks - L.kL, ky = st * ks;
      const double k2 = kx * kx + ky * ky, k = sqrt(k2);
      const double Vx = L.Vd * cos(va_ang), Vy = L.Vd * sin(va_ang);
      const double Ux = L.Ud * cos(ud_ang), Uy = L.Ud * sin(ud_ang);
      const double wd = (ws - L.wL) - (kx * Vx + ky * Vy);
      const double aa = wd / k2;
      const double xex = (aa * kx - Ux) * L.ivTe, xey = (aa * ky - Uy) * L.ivTe;
      const double xmag = sqrt(xex * xex + xey * xey);
      sc[0] = xex / xmag; sc[1] = xey / xmag; sc[2] = xmag;   // cos / sin of beta (atan + heaviside of :552-558)
      sc[3] = k2; sc[4] = k; sc[5] = wd; sc[6] = ws; sc[7] = L.a_e; sc[8] = L.ivTe; sc[9] = L.wL; sc[10] = L.pref;
      sc[11] = kx; sc[12] = ky; sc[13] = aa; sc[14] = xex; sc[15] = xey; sc[16] = ks; sc[17] = Vx; sc[18] = Vy;
      sc[19] = Ux; sc[20] = Uy; sc[21] = ct; sc[22] = st;
#pragma unroll
      for (int s = 0; s < NI; ++s) { sc[24 + 3 * s] = L.ixi[s]; sc[25 + 3 * s] = L.a_i[s]; sc[26 + 3 * s] = L.cs[s]; }
    }
    __syncthreads();
    const double cb = sc[0], sb = sc[1], xmag = sc[2];
    // ---- forward projection ----
    if (active) {
      if (nv <= 256) {
        const int iy = gt % nvp, pt = gt / nvp;
        if (iy < nv) {
          const int ix0 = (nv * pt) / nparts, ix1 = (nv * (pt + 1)) / nparts;
          const double y = v0 + iy * dv;
          double acc = 0.0, ul, vl, xi_d = (double)ix0;
          line_origin(cb, sb, y, v0, idv, ul, vl);
          for (int ix = ix0; ix < ix1; ++ix, xi_d += 1.0)
            acc += bicubic_sample(F, nv, pitch, __builtin_fma(xi_d, cb, ul), __builtin_fma(xi_d, sb, vl));
          part[pt * nv + iy] = acc;
        }
      } else {
        for (int iy = gt; iy < nv; iy += kThreads) {
          const double y = v0 + iy * dv;
          double acc = 0.0, ul, vl, xi_d = 0.0;
          line_origin(cb, sb, y, v0, idv, ul, vl);
          for (int ix = 0; ix < nv; ++ix, xi_d += 1.0)
            acc += bicubic_sample(F, nv, pitch, __builtin_fma(xi_d, cb, ul), __builtin_fma(xi_d, sb, vl));
          part[iy] = acc;
        }
      }
    }
    __syncthreads();
    if (active)
      for (int i = gt; i < nv; i += kThreads) {
        double sacc = 0.0;
        for (int q = 0; q < nparts; ++q) sacc += part[q * nv + i];
        f1[i] = sacc * dv;
      }
    __syncthreads();
    if (active)
      for (int i = gt; i < nv; i += kThreads) {
        double gd;
        if (i == 0) gd = (f1[1] - f1[0]) * idv;
        else if (i == nv - 1) gd = (f1[nv - 1] - f1[nv - 2]) * idv;
        else gd = (f1[i + 1] - f1[i - 1]) * (0.5 * idv);
        d1[i] = gd;
      }
    __syncthreads();
    // ---- ratintn forward (value) and its partial derivatives per interval ----
    double psum = 0.0;
    if (active)
      for (int i = gt; i < nv - 2; i += kThreads) {
        const double f0 = d1[i], f1v = d1[i + 1];
        const double g0 = (v0 + i * dv) - xmag, g1 = (v0 + (i + 1) * dv) - xmag;
        const double fdif = f1v - f0, gdif = g1 - g0, fav = 0.5 * (f1v + f0), gav = 0.5 * (g1 + g0);
        const double tmp = fav * gdif - gav * fdif;
        double r, rfd, rfa, rga;   // r and dr/d(fdif, fav, gav)
        if (fabs(gdif) < 1.0e-4 * fabs(gav)) {
          const double ig = 1.0 / gav, c = gdif / (12.0 * gav * gav * gav);
          r = fav * ig + tmp * c;
          rfd = -gav * c; rfa = ig + gdif * c;
          rga = -fav * ig * ig - fdif * c - 3.0 * tmp * c * ig;
        } else {
          const double lg = log(fabs((gav + 0.5 * gdif) / (gav - 0.5 * gdif))), ig2 = 1.0 / (gdif * gdif);
          r = fdif / gdif + tmp * lg * ig2;
          rfd = 1.0 / gdif - gav * lg * ig2;
          rfa = gdif * lg * ig2;
          rga = (-fdif * lg + tmp * (1.0 / (gav + 0.5 * gdif) - 1.0 / (gav - 0.5 * gdif))) * ig2;
        }
        psum += r * dv;
        c0[i] = (-rfd + 0.5 * rfa) * dv;   // d(sum r dv)/d d1[i]
        c1[i] = (rfd + 0.5 * rfa) * dv;    // d(sum r dv)/d d1[i + 1]
        part[i] = -rga * dv;               // d(sum r dv)/d xmag   (gav = mid - xmag)
      }
    psum = wave_sum(psum);
    if ((gt & 63) == 0) red[gt >> 6] = psum;
    __syncthreads();
    // ---- one thread: forward assembly and its reverse ----
    if (active && gt == 0) {
      const double R = (red[0] + red[1]) + (red[2] + red[3]);
      const double k2 = sc[3], k = sc[4], wd = sc[5], ws = sc[6], a_e = sc[7], ivTe = sc[8], wL = sc[9], pref = sc[10];
      double u = (xmag - v0) * idv;
      int i = (int)u;
      i = i < 0 ? 0 : (i > nv - 2 ? nv - 2 : i);
      double t = (xmag - (v0 + i * dv)) * idv;
      const bool tin = t >= 0.0 && t <= 1.0;
      t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
      const double fe_vphi = f1[i] + t * (f1[i + 1] - f1[i]);
      const double dfe = d1[i] + t * (d1[i + 1] - d1[i]);
      const double ike2 = a_e / k2;
      const double cer = -ike2 * R, cei = kPi * ike2 * dfe;
      double cre = 0.0, cim = 0.0, gsum = 0.0;
      const double vph = wd / k;
      double xi[NI], zr[NI], zi[NI], dzr[NI], dzi[NI], gs[NI], iki2[NI];
#pragma unroll
      for (int s = 0; s < NI; ++s) {
        xi[s] = vph * sc[24 + 3 * s];
        ion_terms(S.zp, xi[s], zr[s], zi[s], dzr[s], dzi[s], gs[s]);
        iki2[s] = sc[25 + 3 * s] / k2;
        cre -= 0.5 * iki2[s] * zr[s];
        cim -= 0.5 * iki2[s] * zi[s];
        gsum += sc[26 + 3 * s] * gs[s];
      }
      const double er = 1.0 + cer + cre, ei = cei + cim;
      const double eps2 = er * er + ei * ei, ce2 = cer * cer + cei * cei;
      const double ci2 = (1.0 + cre) * (1.0 + cre) + cim * cim;
      const double N = gsum * ce2 + ci2 * fe_vphi * ivTe;
      const double Sv = N / (k * eps2);
      const double dop = 1.0 + 2.0 * wd / wL, Q = pref * ws * ws;
      // reverse
      const double Pb = Pbar[pid];
      const double Svb = Pb * dop * Q;
      double wdb = Pb * Sv * Q * 2.0 / wL;
      double wLb = -Pb * Sv * Q * 2.0 * wd / (wL * wL);
      const double prefb = Pb * Sv * dop * ws * ws;
      const double Nb = Svb / (k * eps2);
      double kb = -Svb * N / (k * k * eps2);
      const double eps2b = -Svb * N / (k * eps2 * eps2);
      const double gsumb = Nb * ce2, ce2b = Nb * gsum, ci2b = Nb * fe_vphi * ivTe;
      const double fevb = Nb * ci2 * ivTe;
      double ivTeb = Nb * ci2 * fe_vphi;
      const double erb = 2.0 * er * eps2b, eib = 2.0 * ei * eps2b;
      const double cerb = erb + 2.0 * cer * ce2b, ceib = eib + 2.0 * cei * ce2b;
      const double creb = erb + 2.0 * (1.0 + cre) * ci2b, cimb = eib + 2.0 * cim * ci2b;
      double ike2b = -cerb * R + ceib * kPi * dfe;
      const double Rb = -cerb * ike2, dfeb = ceib * kPi * ike2;
      const double a_eb = ike2b / k2;
      double k2b = -ike2b * a_e / (k2 * k2);
      double vphb = 0.0;
      double* ob = sc + 44;  // adjoints of the ion scalars: [ixi, a_i, cs] per species
#pragma unroll
      for (int s = 0; s < NI; ++s) {
        const double iki2b = -0.5 * (creb * zr[s] + cimb * zi[s]);
        const double zrb = -0.5 * iki2[s] * creb, zib = -0.5 * iki2[s] * cimb;
        const double gsb = gsumb * sc[26 + 3 * s];
        const double xib = zrb * dzr[s] + zib * dzi[s] + gsb * gs[s] * (-2.0 * xi[s]);
        ob[3 * s + 2] = gsumb * gs[s];            // cs
        ob[3 * s + 1] = iki2b / k2;               // a_i
        k2b -= iki2b * sc[25 + 3 * s] / (k2 * k2);
        vphb += xib * sc[24 + 3 * s];
        ob[3 * s] = xib * vph;                    // ixi
      }
      wdb += vphb / k;
      kb -= vphb * wd / (k * k);
      k2b += kb / (2.0 * k);
      // the two interpolations at |xi_e| (jnp.interp: zero slope where clamped)
      double xmagb = tin ? (fevb * (f1[i + 1] - f1[i]) + dfeb * (d1[i + 1] - d1[i])) * idv : 0.0;
      sc[36] = Rb; sc[37] = fevb; sc[38] = dfeb; sc[39] = (double)i; sc[40] = t;
      sc[41] = xmagb; sc[42] = wdb; sc[43] = wLb; sc[56] = prefb; sc[57] = ivTeb; sc[58] = a_eb; sc[59] = k2b;
    }
    __syncthreads();
    // ---- group: d1bar, xmagbar (ratintn part), f1bar ----
    double xms = 0.0;
    if (active) {
      const double Rb = sc[36], dfeb = sc[38], t = sc[40];
      const int il = (int)sc[39];
      for (int i = gt; i < nv - 2; i += kThreads) xms += part[i] * Rb;
      for (int i = gt; i < nv; i += kThreads) {
        double v = 0.0;
        if (i < nv - 2) v += c0[i] * Rb;
        if (i >= 1 && i - 1 < nv - 2) v += c1[i - 1] * Rb;
        if (i == il) v += dfeb * (1.0 - t);
        if (i == il + 1) v += dfeb * t;
        c0[i] = v;   // d1bar, in place: a thread reads c0 only at its own index here, c1 is not written in this loop
      }
    }
    xms = wave_sum(xms);
    if ((gt & 63) == 0) red[8 + (gt >> 6)] = xms;
    __syncthreads();
    if (active) {
      const double fevb = sc[37], t = sc[40];
      const int il = (int)sc[39];
      for (int i = gt; i < nv; i += kThreads) {   // gradient stencil transposed + the fe(v_phi) interpolation
        double v = 0.0;
        if (i + 1 <= nv - 2 && i + 1 >= 1) v -= 0.5 * idv * c0[i + 1];
        if (i - 1 >= 1 && i - 1 <= nv - 2) v += 0.5 * idv * c0[i - 1];
        if (i == 1) v += idv * c0[0];
        if (i == 0) v -= idv * c0[0];
        if (i == nv - 1) v += idv * c0[nv - 1];
        if (i == nv - 2) v -= idv * c0[nv - 1];
        if (i == il) v += fevb * (1.0 - t);
        if (i == il + 1) v += fevb * t;
        c1[i] = v;   // f1bar
      }
    }
    __syncthreads();
    if (active && f1bar_out)
    {
      for (int i = gt; i < nv; i += kThreads) f1bar_out[(size_t)(pid - pbegin) * (nv + 2) + i] = c1[i];
      if (gt == 0) { f1bar_out[(size_t)(pid - pbegin) * (nv + 2) + nv] = cb; f1bar_out[(size_t)(pid - pbegin) * (nv + 2) + nv + 1] = sb; }
    }
    // ---- derivative sampling sweep: adjoint of (cos beta, sin beta) ----
    double acb = 0.0, asb = 0.0;
    if (active) {
      if (nv <= 256) {
        const int iy = gt % nvp, pt = gt / nvp;
        if (iy < nv) {
          const int ix0 = (nv * pt) / nparts, ix1 = (nv * (pt + 1)) / nparts;
          const double y = v0 + iy * dv;
          double s1 = 0.0, s2 = 0.0, ul, vl;
          line_origin(cb, sb, y, v0, idv, ul, vl);
          for (int ix = ix0; ix < ix1; ++ix) {
            const double x = v0 + ix * dv;
            double Sx, Sy;
            bicubic_sample_grad(F, nv, pitch, idv, __builtin_fma((double)ix, cb, ul), __builtin_fma((double)ix, sb, vl), Sx, Sy);
            s1 += Sx * x + Sy * y;
            s2 += -Sx * y + Sy * x;
          }
          acb = s1 * c1[iy] * dv;
          asb = s2 * c1[iy] * dv;
        }
      } else {
        for (int iy = gt; iy < nv; iy += kThreads) {
          const double y = v0 + iy * dv;
          double s1 = 0.0, s2 = 0.0, ul, vl;
          line_origin(cb, sb, y, v0, idv, ul, vl);
          for (int ix = 0; ix < nv; ++ix) {
            const double x = v0 + ix * dv;
            double Sx, Sy;
            bicubic_sample_grad(F, nv, pitch, idv, __builtin_fma((double)ix, cb, ul), __builtin_fma((double)ix, sb, vl), Sx, Sy);
            s1 += Sx * x + Sy * y;
            s2 += -Sx * y + Sy * x;
          }
          acb += s1 * c1[iy] * dv;
          asb += s2 * c1[iy] * dv;
        }
      }
    }
    acb = wave_sum(acb);
    asb = wave_sum(asb);
    if ((gt & 63) == 0) { red[gt >> 6] = acb; red[4 + (gt >> 6)] = asb; }
    __syncthreads();
    // ---- one thread: chain to the lineout scalars ----
    if (active && gt == 0) {
      const double cbb = (red[0] + red[1]) + (red[2] + red[3]), sbb = (red[4] + red[5]) + (red[6] + red[7]);
      double xmagb = sc[41] + (red[8] + red[9]) + (red[10] + red[11]);
      double wdb = sc[42], wLb = sc[43], ivTeb = sc[57], k2b = sc[59];
      const double prefb = sc[56], a_eb = sc[58];
      const double k2 = sc[3], wd = sc[5], ivTe = sc[8], kx = sc[11], ky = sc[12], aa = sc[13], xex = sc[14], xey = sc[15];
      const double ks = sc[16], Vx = sc[17], Vy = sc[18], Ux = sc[19], Uy = sc[20], ct = sc[21], st = sc[22];
      // cb = xex / xmag, sb = xey / xmag, xmag = |(xex, xey)|
      const double xm = sc[2];
      const double xmt = xmagb - (cbb * xex + sbb * xey) / (xm * xm);
      const double xexb = cbb / xm + xmt * xex / xm, xeyb = sbb / xm + xmt * xey / xm;
      // xex = (aa kx - Ux) ivTe
      double aab = (xexb * kx + xeyb * ky) * ivTe;
      double kxb = xexb * aa * ivTe, kyb = xeyb * aa * ivTe;
      const double Uxb = -xexb * ivTe, Uyb = -xeyb * ivTe;
      ivTeb += xexb * (aa * kx - Ux) + xeyb * (aa * ky - Uy);
      // aa = wd / k2
      wdb += aab / k2;
      k2b -= aab * wd / (k2 * k2);
      // wd = ws - wL - (kx Vx + ky Vy)
      wLb -= wdb;
      kxb -= wdb * Vx; kyb -= wdb * Vy;
      const double Vxb = -wdb * kx, Vyb = -wdb * ky;
      // k2 = kx^2 + ky^2
      kxb += 2.0 * kx * k2b; kyb += 2.0 * ky * k2b;
      // kx = ct ks - kL, ky = st ks
      const double ksb = ct * kxb + st * kyb;
      const double kLb = -kxb;
      const double wpe2b = -ksb / (2.0 * kC * kC * ks);
      const double Vdb = Vxb * cos(va_ang) + Vyb * sin(va_ang);
      const double Udb = Uxb * cos(ud_ang) + Uyb * sin(ud_ang);
      double* o = LBacc + ((size_t)b * G + g) * NLB;
      atomicAdd(o + 0, wpe2b); atomicAdd(o + 1, wLb); atomicAdd(o + 2, kLb); atomicAdd(o + 3, ivTeb);
      atomicAdd(o + 4, a_eb); atomicAdd(o + 5, prefb); atomicAdd(o + 6, Udb); atomicAdd(o + 7, Vdb);
#pragma unroll
      for (int s = 0; s < NI; ++s) {
        atomicAdd(o + 8 + 3 * s, sc[44 + 3 * s]); atomicAdd(o + 9 + 3 * s, sc[45 + 3 * s]); atomicAdd(o + 10 + 3 * s, sc[46 + 3 * s]);
      }
    }
  }
}

// Table adjoint: Fbar[cell entries] += f1bar[point][iy] dv wx[m] wy[n] for every sample of every point.  The scatter needs
// the sample weights only, so the LDS of this pass holds the padded table ADJOINT (LDS atomics, ds_add_f64); each
// persistent workgroup adds its partial table to the global one at the end.  Tables larger than LDS are cut into tiles
// of kTile2 x kTile2 cells (blockIdx.y = tile; a tile of cells touches (kTile2 + 3)^2 padded entries): the workgroups
// of a tile walk ALL points but visit only the samples whose cell lies in their tile -- per column iy that is one
// contiguous run of ix, found from the intersection of the rotated line with the tile rectangle (taken two samples
// wide on each side and then decided per sample by the very floor() every tile evaluates, so that each sample lands in
// exactly one tile).
constexpr int kTile2 = 128;
// indices ix with lo <= c * ix + e < hi (lo_inf / hi_inf: that side is open), widened by two
__device__ __forceinline__ void line_range(double c, double e, double lo, double hi, bool lo_inf, bool hi_inf, int& a, int& b) {
  if (fabs(c) < 1.0e-9) return;   // (almost) parallel to the tile edge: the per-sample test decides
  double x0 = lo_inf ? -1.0e9 : (lo - e) / c, x1 = hi_inf ? 1.0e9 : (hi - e) / c;
  if (c < 0.0) { const double t = x0; x0 = x1; x1 = t; if (lo_inf) x1 = 1.0e9; if (hi_inf) x0 = -1.0e9; }
  x0 = fmin(fmax(x0, -1.0e9), 1.0e9);
  x1 = fmin(fmax(x1, -1.0e9), 1.0e9);
  const int ia = (int)floor(x0) - 2, ib = (int)ceil(x1) + 2;
  a = a > ia ? a : ia;
  b = b < ib ? b : ib;
}
__global__ __launch_bounds__(4 * kThreads) void k_ff2d_table_adj(int nv, const double* __restrict__ f1bar, long npoint,
                                                                  double* __restrict__ Fbar_pad) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* T = reinterpret_cast<double*>(smem);
  const int ncell = nv - 1, ntx = (ncell + kTile2 - 1) / kTile2;
  const int cx0 = (blockIdx.y / ntx) * kTile2, cy0 = (blockIdx.y % ntx) * kTile2;
  const int cx1 = cx0 + kTile2 < ncell ? cx0 + kTile2 : ncell, cy1 = cy0 + kTile2 < ncell ? cy0 + kTile2 : ncell;
  const int trow = cx1 - cx0 + 3, tcol = cy1 - cy0 + 3, pitch = tcol | 1;
  const bool whole = ntx == 1;
  for (int i = threadIdx.x; i < trow * pitch; i += blockDim.x) T[i] = 0.0;
  __syncthreads();
  const double dv = 12.0 / nv, v0 = -6.0 + 0.5 * dv, idv = 1.0 / dv;
  const int grp = threadIdx.x >> 8, gt = threadIdx.x & (kThreads - 1);
  const int nvp = nv <= 64 ? 64 : (nv <= 128 ? 128 : 256);
  const int nparts = nv <= 256 ? kThreads / nvp : 1;
  for (long pid = (long)blockIdx.x * 4 + grp; pid < npoint; pid += (long)gridDim.x * 4) {
    const double* fb = f1bar + (size_t)pid * (nv + 2);
    const double cb = fb[nv], sb = fb[nv + 1];
    for (int iy = nv <= 256 ? gt % nvp : gt; iy < nv; iy += kThreads) {
      const int pt = nv <= 256 ? gt / nvp : 0;
      int ix0 = (nv * pt) / nparts, ix1 = (nv * (pt + 1)) / nparts;
      const double y = v0 + iy * dv, val = fb[iy] * dv;
      if (!whole) {
        // cell coordinate of sample ix along each table axis: u = c * ix + e
        line_range(cb, (cb * v0 - y * sb - v0) * idv, (double)cx0, (double)cx1, cx0 == 0, cx1 == ncell, ix0, ix1);
        line_range(sb, (sb * v0 + y * cb - v0) * idv, (double)cy0, (double)cy1, cy0 == 0, cy1 == ncell, ix0, ix1);
      }
      double ul, vl;
      line_origin(cb, sb, y, v0, idv, ul, vl);
      for (int ix = ix0; ix < ix1; ++ix) {
        int cx, cy;
        double tx, ty;
        cell_of(__builtin_fma((double)ix, cb, ul), nv, cx, tx);
        cell_of(__builtin_fma((double)ix, sb, vl), nv, cy, ty);
        if (!whole && (cx < cx0 || cx >= cx1 || cy < cy0 || cy >= cy1)) continue;
        double wx[4], wy[4];
        catmull_rom(tx, wx);
        catmull_rom(ty, wy);
        double* q0 = T + (size_t)(cx - cx0) * pitch + (cy - cy0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const double wm = val * wx[m];
#pragma unroll
          for (int n = 0; n < 4; ++n) atomicAdd(q0 + (size_t)m * pitch + n, wm * wy[n]);
        }
      }
      if (nv <= 256) break;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < trow * tcol; i += blockDim.x) {
    const int r = i / tcol, c = i % tcol;
    const double v = T[r * pitch + c];
    if (v != 0.0) atomicAdd(Fbar_pad + (size_t)(cx0 + r) * (nv + 2) + (cy0 + c), v);
  }
}

// adjoint of the ghost cells (ghost_cols was applied last, so it is reversed first), then the interior is copied out
__global__ __launch_bounds__(kThreads) void k_ff2d_fold_ghosts(int nv, double* __restrict__ P, double* __restrict__ out) {
  const int pitch = nv + 2;
  for (int r = threadIdx.x; r < nv + 2; r += kThreads) {
    double* row = P + (size_t)r * pitch;
    row[1] += 2.0 * row[0]; row[2] -= row[0];
    row[nv] += 2.0 * row[nv + 1]; row[nv - 1] -= row[nv + 1];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < nv; c += kThreads) {
    P[pitch + c + 1] += 2.0 * P[c + 1]; P[2 * pitch + c + 1] -= P[c + 1];
    P[(size_t)nv * pitch + c + 1] += 2.0 * P[(size_t)(nv + 1) * pitch + c + 1];
    P[(size_t)(nv - 1) * pitch + c + 1] -= P[(size_t)(nv + 1) * pitch + c + 1];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nv * nv; i += kThreads) out[i] = P[(size_t)(i / nv + 1) * pitch + (i % nv + 1)];
}

// LBacc[b][g][NLB] -> gphys[b][NP]: make_lines_adjoint per gradient point, summed (one thread per lineout)
template <int NI>
__global__ void k_ff2d_lines_adj(KStatic S, const double* __restrict__ phys, int f, int B, const double* __restrict__ LBacc,
                                 double* __restrict__ gphys, int with_m) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  constexpr int NPk = TSFF_NP(NI);
  const int NLB = kNLB2 + 3 * NI + (with_m ? 1 : 0);   // (the 1-D path carries one more slot: the DLM-order tangent)
  Phys<NI> p;
  load_phys<NI>(phys + (size_t)b * S.NP, S.p_scale, S.p_shift, S.p_sig, S.ti_same, false, p);
  double pb[NPk];
#pragma unroll
  for (int s = 0; s < NPk; ++s) pb[s] = 0.0;
  for (int g = 0; g < S.G; ++g) {
    LineS<NI> L, LB;
    make_lines<NI>(p, S.lam_shift[f], g, S.G, L);
    zero_lines<NI>(LB);
    const double* o = LBacc + ((size_t)b * S.G + g) * NLB;
    LB.wpe2 = o[0]; LB.wL = o[1]; LB.kL = o[2]; LB.ivTe = o[3]; LB.a_e = o[4]; LB.pref = o[5]; LB.Ud = o[6]; LB.Vd = o[7];
#pragma unroll
    for (int s = 0; s < NI; ++s) { LB.ixi[s] = o[8 + 3 * s]; LB.a_i[s] = o[9 + 3 * s]; LB.cs[s] = o[10 + 3 * s]; }
    if (with_m) LB.m = o[NLB - 1];
    make_lines_adjoint<NI>(p, S.lam_shift[f], g, S.G, L, LB, pb);
  }
#pragma unroll
  for (int s = 0; s < NPk; ++s) gphys[(size_t)b * NPk + s] = pb[s];
}

// ------------------------------------------------------------------------------------------
// Angular (ARTS) instrument chain: FitModel.electron_spectrum for spectype "angular_full"
// (generate_spectra.py:193-216), add_ATS_IRF (irf.py:5-47), reduce_ATS_to_resunit
// (thomson_diagnostic.py:78-107).  Five small kernels over [n_px x npts] images; none of them is hot.
// ------------------------------------------------------------------------------------------
// M[r][j] = filt[j] * sum_a Wt[r][a] * mean_g P[g][j][a]
__global__ void k_ats_weights(const double* __restrict__ P, const double* __restrict__ Wt, const double* __restrict__ filt,
                              int G, int npts, int NA, int npx, double* __restrict__ M) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (j >= npts) return;
  double acc = 0.0;
  for (int a = 0; a < NA; ++a) {
    const double w = Wt[(size_t)r * NA + a];
    if (w == 0.0) continue;
    double m = 0.0;
    for (int g = 0; g < G; ++g) m += P[((size_t)g * npts + j) * NA + a];
    acc += w * (m / (double)G);
  }
  M[(size_t)r * npts + j] = filt ? acc * filt[j] : acc;
}

// "same" convolution along the angular-pixel axis (dim 0) or the wavelength axis (dim 1):
// y[i] = sum_s taps[s] x[i + off + s], zero outside
__global__ void k_ats_conv(const double* __restrict__ X, const double* __restrict__ taps, int nt, int off, int along_rows,
                           int npx, int npts, double* __restrict__ Y) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (j >= npts) return;
  double acc = 0.0;
  if (along_rows) {
    for (int s = 0; s < nt; ++s) {
      const int rr = r + off + s;
      if (rr >= 0 && rr < npx) acc += taps[s] * X[(size_t)rr * npts + j];
    }
  } else {
    for (int s = 0; s < nt; ++s) {
      const int jj = j + off + s;
      if (jj >= 0 && jj < npts) acc += taps[s] * X[(size_t)r * npts + jj];
    }
  }
  Y[(size_t)r * npts + j] = acc;
}

// per angular pixel: Y[r][:] *= max_j M[r][:] / max_j Y[r][:]   (irf.py:38)
__global__ __launch_bounds__(kThreads) void k_ats_rownorm(const double* __restrict__ M, double* __restrict__ Y, int npts) {
  __shared__ double red[8];
  const int r = blockIdx.x;
  double mm = -1e300, my = -1e300;
  for (int j = threadIdx.x; j < npts; j += kThreads) {
    mm = fmax(mm, M[(size_t)r * npts + j]);
    my = fmax(my, Y[(size_t)r * npts + j]);
  }
  int dummy = 0;
  block_argmax(mm, dummy, red);
  dummy = 0;
  block_argmax(my, dummy, red);
  const double sc = mm / my;
  for (int j = threadIdx.x; j < npts; j += kThreads) Y[(size_t)r * npts + j] *= sc;
}

// resolution-unit reduction + amplitude scaling (thomson_diagnostic.py:93-106): one workgroup per output row
__global__ __launch_bounds__(kThreads) void k_ats_resunit(const double* __restrict__ Y, const double* __restrict__ lam_nm,
                                                          int npts, int lam_step, int ang_step, int row_start,
                                                          const double* __restrict__ e_amps, double lam, double amp1, double amp2,
                                                          double* __restrict__ out) {
  __shared__ double red[8];
  const int R = blockIdx.x, nJ = npts / lam_step;
  const int r0 = (row_start + R) * ang_step;
  double mx = -1e300;
  for (int J = threadIdx.x; J < nJ; J += kThreads) {
    double acc = 0.0;
    for (int dr = 0; dr < ang_step; ++dr)
      for (int dj = 0; dj < lam_step; ++dj) acc += Y[(size_t)(r0 + dr) * npts + J * lam_step + dj];
    acc /= (double)(ang_step * lam_step);
    out[(size_t)R * nJ + J] = acc;
    mx = fmax(mx, acc);
  }
  int dummy = 0;
  block_argmax(mx, dummy, red);
  for (int J = threadIdx.x; J < nJ; J += kThreads) {
    double lb = 0.0;
    for (int dj = 0; dj < lam_step; ++dj) lb += lam_nm[J * lam_step + dj];
    lb /= (double)lam_step;
    out[(size_t)R * nJ + J] = e_amps[R] * out[(size_t)R * nJ + J] / mx * (lb < lam ? amp1 : amp2);
  }
}

// ---- adjoint of the ARTS instrument chain (reverse of the five kernels above) ----
// per angular pixel r: maxima of the unconvolved (M) and convolved (Bm, unscaled) rows and where they sit
__global__ __launch_bounds__(kThreads) void k_ats_rowstats(const double* __restrict__ M, const double* __restrict__ Bm, int npts,
                                                           double* __restrict__ stats /*[npx][4]: mM, jM, mB, jB*/) {
  __shared__ double red[8];
  const int r = blockIdx.x;
  double mm = -1e300, mb = -1e300;
  int jm = 0, jb = 0;
  for (int j = threadIdx.x; j < npts; j += kThreads) {
    const double a = M[(size_t)r * npts + j], c = Bm[(size_t)r * npts + j];
    if (a > mm) { mm = a; jm = j; }
    if (c > mb) { mb = c; jb = j; }
  }
  block_argmax(mm, jm, red);
  block_argmax(mb, jb, red);
  if (threadIdx.x == 0) { stats[4 * r] = mm; stats[4 * r + 1] = jm; stats[4 * r + 2] = mb; stats[4 * r + 3] = jb; }
}

// reverse of k_ats_resunit (+ the scaling of k_ats_rownorm): one workgroup per output row R.  Ebar [rows][nJ] ->
// Cbar[r][j] for the rows / columns of R's resolution units (C = Bm * sc_r); amp adjoints accumulated per row.
__global__ __launch_bounds__(kThreads) void k_ats_resunit_adj(const double* __restrict__ Bm, const double* __restrict__ stats,
                                                              const double* __restrict__ lam_nm, int npts, int lam_step,
                                                              int ang_step, int row_start, const double* __restrict__ e_amps,
                                                              double lam, double amp1, double amp2,
                                                              const double* __restrict__ Ebar, double* __restrict__ Cbar,
                                                              double* __restrict__ ampbar /*[rows][2]*/) {
  __shared__ double red[8];
  __shared__ double Dl[TSFF_NBINS];
  const int R = blockIdx.x, nJ = npts / lam_step;
  const int r0 = (row_start + R) * ang_step;
  const double inv = 1.0 / (double)(ang_step * lam_step);
  double mx = -1e300;
  int js = 0;
  for (int J = threadIdx.x; J < nJ; J += kThreads) {
    double acc = 0.0;
    for (int dr = 0; dr < ang_step; ++dr) {
      const double sc = stats[4 * (r0 + dr)] / stats[4 * (r0 + dr) + 2];
      for (int dj = 0; dj < lam_step; ++dj) acc += Bm[(size_t)(r0 + dr) * npts + J * lam_step + dj] * sc;
    }
    acc *= inv;
    Dl[J] = acc;
    if (acc > mx) { mx = acc; js = J; }
  }
  block_argmax(mx, js, red);
  double su = 0.0, a1 = 0.0, a2 = 0.0;
  const double ea = e_amps[R];
  for (int J = threadIdx.x; J < nJ; J += kThreads) {
    double lb = 0.0;
    for (int dj = 0; dj < lam_step; ++dj) lb += lam_nm[J * lam_step + dj];
    lb /= (double)lam_step;
    const bool blue = lb < lam;
    const double eb = Ebar[(size_t)R * nJ + J];
    const double base = eb * ea * Dl[J] / mx;      // d out / d amp
    if (blue) a1 += base; else a2 += base;
    su += eb * ea * (blue ? amp1 : amp2) * Dl[J] / mx;   // sum_J u_J D_J, u_J = Ebar ea amp / mx
  }
  su = block_sum(su, red);
  a1 = block_sum(a1, red);
  a2 = block_sum(a2, red);
  if (threadIdx.x == 0) { ampbar[2 * R] = a1; ampbar[2 * R + 1] = a2; }
  for (int J = threadIdx.x; J < nJ; J += kThreads) {
    double lb = 0.0;
    for (int dj = 0; dj < lam_step; ++dj) lb += lam_nm[J * lam_step + dj];
    lb /= (double)lam_step;
    double db = Ebar[(size_t)R * nJ + J] * ea * (lb < lam ? amp1 : amp2) / mx;
    if (J == js) db -= su / mx;
    db *= inv;
    for (int dr = 0; dr < ang_step; ++dr)
      for (int dj = 0; dj < lam_step; ++dj) Cbar[(size_t)(r0 + dr) * npts + J * lam_step + dj] = db;
  }
}

// reverse of k_ats_rownorm: C = Bm * sc, sc = mM / mB.  In place: Cbar -> Bmbar; Mbar gets the one-hot of d sc / d mM
__global__ __launch_bounds__(kThreads) void k_ats_rownorm_adj(const double* __restrict__ Bm, const double* __restrict__ stats,
                                                              int npts, double* __restrict__ Cbar, double* __restrict__ Mbar) {
  __shared__ double red[8];
  const int r = blockIdx.x;
  const double mM = stats[4 * r], mB = stats[4 * r + 2];
  const int jM = (int)stats[4 * r + 1], jB = (int)stats[4 * r + 3];
  double scb = 0.0;
  for (int j = threadIdx.x; j < npts; j += kThreads) scb += Cbar[(size_t)r * npts + j] * Bm[(size_t)r * npts + j];
  scb = block_sum(scb, red);
  const double sc = mM / mB;
  for (int j = threadIdx.x; j < npts; j += kThreads) {
    double v = Cbar[(size_t)r * npts + j] * sc;
    if (j == jB) v -= scb * mM / (mB * mB);
    Cbar[(size_t)r * npts + j] = v;
    Mbar[(size_t)r * npts + j] = j == jM ? scb / mB : 0.0;
  }
}

// reverse of k_ats_conv: Xbar[i] (+)= sum_s taps[s] Ybar[i - off - s]
__global__ void k_ats_conv_adj(const double* __restrict__ Yb, const double* __restrict__ taps, int nt, int off, int along_rows,
                               int npx, int npts, int accumulate, double* __restrict__ Xb) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (j >= npts) return;
  double acc = 0.0;
  if (along_rows) {
    for (int s = 0; s < nt; ++s) {
      const int rr = r - off - s;
      if (rr >= 0 && rr < npx) acc += taps[s] * Yb[(size_t)rr * npts + j];
    }
  } else {
    for (int s = 0; s < nt; ++s) {
      const int jj = j - off - s;
      if (jj >= 0 && jj < npts) acc += taps[s] * Yb[(size_t)r * npts + jj];
    }
  }
  double* o = Xb + (size_t)r * npts + j;
  *o = accumulate ? *o + acc : acc;
}

// reverse of k_ats_weights: Pbar[g][j][a] = filt[j] / G * sum_r Wt[r][a] Mbar[r][j]
__global__ void k_ats_weights_adj(const double* __restrict__ Mbar, const double* __restrict__ Wt, const double* __restrict__ filt,
                                  int G, int npts, int NA, int npx, double* __restrict__ Pbar) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (a >= NA) return;
  double acc = 0.0;
  for (int r = 0; r < npx; ++r) {
    const double w = Wt[(size_t)r * NA + a];
    if (w != 0.0) acc += w * Mbar[(size_t)r * npts + j];
  }
  acc *= (filt ? filt[j] : 1.0) / (double)G;
  for (int g = 0; g < G; ++g) Pbar[((size_t)g * npts + j) * NA + a] = acc;
}

// ------------------------------------------------------------------------------------------
// k_fma_peak: micro-benchmark of the FP64 vector FMA rate (the roof this path is bound by): 16 independent
// accumulators per lane, `iters` x 16 fused multiply-adds, enough wavefronts to fill every SIMD.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_fma_peak(double* __restrict__ out, int iters, double a, double b) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (double)(threadIdx.x + i) * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[(size_t)blockIdx.x * kThreads + threadIdx.x] = s;
}

// the same for the FP64 matrix cores: 8 independent 16x16 accumulator tiles per wavefront, `iters` x 8 v_mfma_f64_16x16x4_f64
__global__ __launch_bounds__(kThreads) void k_mfma_peak(double* __restrict__ out, int iters, double a, double b) {
  mfma_d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
  const double x = a + 1e-9 * threadIdx.x, y = b + 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * kThreads + threadIdx.x] = s;
}

}  // namespace tsff

#include "tsff_api.inc"

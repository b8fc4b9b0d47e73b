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
constexpr int kTapPad = 16;   // zero padding of the taps for points_per_pixel > 1 (>= the largest ppp of k_spectrum_rows' bin-wise adjoint)
struct KStatic {
  int npts, ppp, n_angles, G, nvx, NP;
  int shared_fe;  // 1: every lineout uses table slot 0
  int loss_method;
  int load[2];
  int raw[2];   // 1: no instrument response for this feature (irf.py:82-86, spect_stddev_ion == 0): Thry = modl + noise
  double lam_shift[2];
  const double* omgs[2];     // [npts] scattered-frequency axis (form_factor.py:134)
  const double* lam_bin[2];  // [1024] binned wavelength axis in nm (irf.py:75,125)
  const double* filt;        // [npts] EPW multiplier (iawfilter) or nullptr
  const double* cos_sa;      // [n_angles]
  const double* sa_rad;      // [n_angles] scattering angles in radians (2-D path)
  const double* w_sa;        // [n_angles]
  const double2* zp;         // [kNZh] Z' table for xi >= 0 (ion_terms)
  const double2* zpf;        // [1640] the full table (spectrum kernels, when LDS allows)
  const double* xi1;         // [1024]
  const double* xi2;         // [1640]
  const double* etab;        // [64] 2^(j/64), the table of fexp_t
  const double* taps[2];
  int ntaps[2];              // length of the bin-averaged IRF taps hb
  int toff[2];               // ybin[p] = sum_s hb[s] x[p * ppp + toff + s]
  // phase-layout convolution (points_per_pixel = 1, 256 threads per feature; see "IRF convolution" in k_spectrum):
  const double* ptaps[2];    // points_per_pixel 1: hb zero-padded by 3 + rounding on both sides, ptaps[3 + s] = hb[s]; else by kTapPad, ptaps[kTapPad + s] = hb[s]
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
  double* lbrec;         // [items][4 wavefronts][kLBRec] lineout-scalar adjoints of k_spectrum_fused (finished by k_fused_finish)
  double wts[3];
  int B;
  int denom_mode;        // MODE 1 only: 0 constant denominators (folded into wts), 2: |data| + 1e-10 per sample
  // k_spectrum_fused launched per column block of the batch (the pipelined DLM plan, launch_fused): this launch covers lineouts
  // [b0, b0 + B) of a batch of Btot; every per-lineout pointer above stays batch-global.  One launch over the whole batch: b0 = 0, Btot = B.
  int b0, Btot;
};

#include "k_tables.inc"
#include "k_spectrum.inc"
constexpr int kLineRec = 24;     // doubles per item of k_fused_prep's record: 9 + 4 n_ion lineout scalars + lam, amp1, amp2, amp3
constexpr int kLBRec = 20;       // doubles per (item, wavefront) record of k_spectrum_fused: 9 + 3 n_ion sums + the two amplitude adjoints
constexpr int kFusedMaxIon = 2;  // k_spectrum_fused is instantiated for n_ion <= 2 (4 x (7 + 3 n_ion) register accumulators per thread)
#include "k_spectrum_fused.inc"
#include "k_forward.inc"
#include "k_spectrum_rows.inc"
#include "k_form_factor.inc"
#include "k_form_factor_2d.inc"
#include "k_ats.inc"
#include "k_peak.inc"

}  // namespace tsff

#ifndef TSFF_NO_API   // (scratch translation units that instantiate single kernels for a look at their assembly: scripts/isa_one.sh)
#include "tsff_api.inc"
#endif

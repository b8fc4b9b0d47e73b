/*
 * tsff.h -- C ABI of libtsff.so, the MI355X (gfx950) Thomson-scattering form-factor engine.
 *
 * Drop-in boundary for ONE hot path of ergodicio/tsadar: the S(k,w) forward model and its
 * parameter gradient over a batch of lineouts.  Citations are file:line in the reference tree.
 *
 *   reference interface                                            replaced by
 *   -------------------------------------------------------------  ---------------------------
 *   FormFactor.__init__            core/physics/form_factor.py:120-161      tsff_create
 *   FormFactor.__call__            core/physics/form_factor.py:163-298      tsff_form_factor
 *   ratintn / ratcen               core/physics/ratintn.py:4-52             tsff_chi_table
 *   DLM1V.__call__                 core/modules/distribution_functions/base.py:277-294
 *                                                                           (fe_mode TSFF_FE_DLM)
 *   ThomsonParams.__call__         core/modules/ts_params.py:583-603        (inside every call)
 *   ThomsonScatteringDiagnostic.__call__  core/thomson_diagnostic.py:109-142  tsff_forward
 *     FitModel.ion/electron_spectrum core/physics/generate_spectra.py:139-220
 *     add_ion_IRF / add_electron_IRF core/physics/irf.py:50-132
 *   LossFunction.vg_loss / __loss__ inverse/loss_function.py:128-168,364-373  tsff_loss_grad
 *   LossFunction.array_loss (post_loss) inverse/loss_function.py:375-384      tsff_array_loss
 *   FormFactor.calc_in_2D / rotate / calc_chi_vals  core/physics/form_factor.py:300-587   tsff_form_factor_2d(_range)
 *   angular_full electron_spectrum, add_ATS_IRF, reduce_ATS_to_resunit
 *       generate_spectra.py:193-216, irf.py:5-47, thomson_diagnostic.py:78-107           tsff_ats_setup / tsff_ats_spectrum
 *   jax reverse mode through the above (equinox.filter_value_and_grad, loss_function.py:108)
 *       tsff_loss_grad(_fe), tsff_form_factor_grad, tsff_form_factor_2d_grad, tsff_ats_adjoint
 *
 * Conventions
 *   - every array pointer in a *call* is a DEVICE pointer (hipMalloc'ed or a torch CUDA tensor);
 *     every pointer inside tsff_config is a HOST pointer and is copied by tsff_create;
 *   - all arithmetic and all arrays are float64, row-major;
 *   - the caller owns every buffer; the library never frees caller memory;
 *   - calls are asynchronous on the handle's stream (tsff_set_stream); the caller synchronises;
 *   - return value 0 = success, negative = error, text via tsff_last_error();
 *   - a handle is bound to the device that was current at tsff_create; handles are not thread-safe,
 *     the library is re-entrant across handles.
 *   - no CPU fallback exists: without a HIP device every entry point fails.
 */
#ifndef TSFF_H
#define TSFF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSFF_ABI_VERSION 8
/* error codes (every entry point returns 0 or one of these; text via tsff_last_error):
 * -1 bad argument, -2 unsupported configuration / option, -3 not a differentiable leaf, -5 HIP runtime error,
 * -22 stale or foreign token of saved projection records (tsff_form_factor_2d_grad),
 * TSFF_ERR_LDS: the configuration needs more LDS than a CU has (wide instrument functions at several points per pixel:
 * spectrum + halo + taps) -- the caller may retry with fewer IRF taps. */
#define TSFF_ERR_LDS (-7)

/* ---- parameter slots of one lineout: params[b][TSFF_NP(n_ion)] (normalised leaves of the
 * reference's ThomsonParams pytree, core/modules/ts_params.py:49-60,395-420,253-262) ---------- */
enum {
  TSFF_P_TE = 0,
  TSFF_P_NE = 1,
  TSFF_P_M = 2, /* DLM super-Gaussian order (base.py:252-262); ignored unless fe_mode == DLM */
  TSFF_P_LAM = 3,
  TSFF_P_AMP1 = 4,
  TSFF_P_AMP2 = 5,
  TSFF_P_AMP3 = 6,
  TSFF_P_NE_GRADIENT = 7,
  TSFF_P_TE_GRADIENT = 8,
  TSFF_P_UD = 9,
  TSFF_P_VA = 10,
  TSFF_P_ION0 = 11 /* then per ion species s: +4s+0 Ti, +1 Z, +2 A, +3 fract */
};
#define TSFF_ION_TI 0
#define TSFF_ION_Z 1
#define TSFF_ION_A 2
#define TSFF_ION_FRACT 3
#define TSFF_NP(n_ion) (TSFF_P_ION0 + 4 * (n_ion))
#define TSFF_MAX_ION 4
#define TSFF_MAX_ANGLES 1024
#define TSFF_NBINS 1024 /* irf.py:74,124: reshape(1024, -1) */
#define TSFF_NXI2 1640  /* form_factor.py:138 */
#define TSFF_NXI1 1024  /* form_factor.py:137 */
#define TSFF_DLM_NM 31  /* base.py:270 */

enum { TSFF_FE_SHARED = 0, TSFF_FE_PER_LINEOUT = 1, TSFF_FE_DLM = 2 };
enum { TSFF_LOSS_L2 = 0, TSFF_LOSS_L1 = 1, TSFF_LOSS_LOGCOSH = 2, TSFF_LOSS_POISSON = 3 };
enum { TSFF_FEATURE_ELE = 0, TSFF_FEATURE_ION = 1 };

typedef struct tsff_config {
  int32_t abi_version; /* = TSFF_ABI_VERSION */

  /* wavelength grids (form_factor.py:132-135; cfg other.lamrangE/lamrangI/npts) */
  double lamrangE[2];
  double lamrangI[2];
  int32_t npts;         /* = 1024 * points_per_pixel */
  int32_t load_ele;     /* other.extraoptions.load_ele_spec */
  int32_t load_ion;     /* other.extraoptions.load_ion_spec */
  double ele_lam_shift; /* data.ele_lam_shift (form_factor.py:196) */

  /* scattering angles and the row `weights[0]` (generate_spectra.py:165,197) */
  int32_t n_angles;
  const double *sa_deg;     /* [n_angles] */
  const double *sa_weights; /* [n_angles] */

  int32_t num_grad_points; /* parameters.general.*_gradient.num_grad_points */
  int32_t n_ion;

  /* electron distribution function */
  int32_t nvx;
  int32_t fe_mode;
  const double *fe_shared; /* [nvx], TSFF_FE_SHARED */
  const double *dlm_table; /* [nvx][TSFF_DLM_NM] = f_vx_m of base.py:272, TSFF_FE_DLM */

  /* static grids/tables of FormFactor.__init__ (form_factor.py:137-139) */
  const double *xi1;       /* [TSFF_NXI1] */
  const double *xi2;       /* [TSFF_NXI2] */
  const double *zprime_re; /* [TSFF_NXI2] rdWT.txt interpolated on xi2 */
  const double *zprime_im; /* [TSFF_NXI2] */
  /* log-ratio table of ratintn.py:49 on the (xi1, xi2) grids, Lg[q][i] = log|(gav + gdif/2)/(gav - gdif/2)|,
   * [TSFF_NXI2][TSFF_NXI1] with the two unused columns (i >= 1022) zero.  Required unless fe_mode == SHARED:
   * per-lineout W tables are then two matrix-vector products with this constant table. */
  const double *lg_table;

  /* instrument response + binning (irf.py:50-132) as ONE set of bin-averaged taps per feature:
   *   ybin[p] = sum_{s < n_taps} taps[s] * x[p * ppp + tap_off + s],   ppp = npts / 1024,
   * i.e. taps = (1/ppp) * (Gaussian of irf.py:66-72 / 110-114, in the reference's "same" alignment) convolved
   * with a ppp-wide box (irf.py:74,124); see DESIGN.md section 3 and engine.binned_taps(). */
  int32_t n_taps_ele;
  int32_t tap_off_ele;
  const double *taps_ele;
  int32_t n_taps_ion; /* 0 <=> spect_stddev_ion == 0 (irf.py:82-86): ThryI = modlI + noise_i, not convolved, not normalised, no
                         amplitudes (needs npts == 1024, as in the reference where [npts] must match the [1024] data) */
  int32_t tap_off_ion;
  const double *taps_ion;
  int32_t norm; /* other.PhysParams.norm; only 0 is implemented */

  /* iawfilter (generate_spectra.py:210-216) as a per-sample multiplier on the EPW grid */
  const double *ele_filter; /* [npts] or NULL (= all ones) */

  /* parameter transform (ts_params.py:329-350): phys = act(x) * scale + shift */
  const double *p_scale;        /* [NP] */
  const double *p_shift;        /* [NP] */
  const uint8_t *p_sigmoid;     /* [NP] 1 = sigmoid activation, 0 = identity */
  uint8_t ti_same[TSFF_MAX_ION]; /* ion-k.Ti.same (ts_params.py:557-558) */

  /* loss (loss_function.py:190-267, 386-418) */
  int32_t loss_method;
  const uint8_t *mask_ele; /* [1024] bit0: blue range fitted, bit1: red range fitted */
  const uint8_t *mask_ion; /* [1024] bit0: IAW range fitted */
} tsff_config;

typedef struct tsff_handle tsff_handle;

/* lifecycle ------------------------------------------------------------------------------- */
int tsff_create(const tsff_config *cfg, tsff_handle **out);
void tsff_destroy(tsff_handle *h);
const char *tsff_last_error(const tsff_handle *h); /* h == NULL: last tsff_create failure */
int tsff_abi_version(void);
int tsff_set_stream(tsff_handle *h, void *hip_stream);
/* options.  TSFF_OPT_DENOM_MODE: denominators of the l1/l2 functionals in tsff_loss_grad -- 0 (default):
 * constants folded into `weights` (LossFunction.__loss__, loss_function.py:364-373); 2: |data| + 1e-10 per sample
 * (LossFunction._loss_for_hess_fn_, loss_function.py:173-188, the loss whose Hessian gives the fit uncertainties).
 * TSFF_OPT_LAUNCH_PLAN: bit mask, 0 (default) automatic -- with two loaded features one launch of 2B 256-thread workgroups, one
 * per (lineout, feature), when two of them fit a CU; tsff_loss_grad runs the one-sweep kernel (forward value and Jacobian rows
 * of every wavelength sample in one pass over the points) where it applies: one gradient point, n_ion <= 2, no gradient w.r.t.
 * the tabulated f_e; with one point per pixel the rows stay in registers, with more (the reference's default decks: 5) they go
 * through a scratch array in device memory that the handle allocates on first use: B x loaded features x (8 + 3 n_ion) x npts
 * doubles, 1 GB at B = 1024 and 5 points per pixel (above 16 GB the two-sweep kernel is used instead).  Bit 0: never interleave (both features in one 512-thread workgroup) and never spread the
 * rounds of a lineout over several workgroups (the small-batch form of the several-points-per-pixel kernel); bit 1: always the
 * two-sweep kernel (and, for tsff_forward at several points per pixel, k_spectrum instead of the forward form of the rounds kernel); bit 2: never three forward-only workgroups per CU (tsff_forward runs three 256-thread workgroups per CU when
 * the batch is large enough to need them); bit 3: the one-sweep kernel evaluates every base point itself instead of taking its pair's
 * right neighbour from the next lane; bit 5 (32): the per-lineout W tables by the 128 x 128-tiled GEMM instead of the 128 x 144 one (bit 4 is
 * unused and refused).  The spectra are identical either way; the gradient differs by rounding.
 * TSFF_OPT_DLM_BLOCKS (experimental, default off): tsff_loss_grad(_packed) with fe_mode TSFF_FE_DLM and the DLM order among the
 * differentiated leaves builds the per-lineout tables (k_fe_vectors + the FP64-MFMA GEMM) in n column blocks of the batch on a second
 * stream owned by the handle (non-blocking, higher priority) while the one-sweep kernel works on the previous block on the handle's
 * stream; the call still returns with everything ordered on the handle's stream (fork / join by events).  0 / 1: off (one stream),
 * n: n blocks (rounded to multiples of 256 lineouts).  Same bits either way.  On gfx950 it is slower than one stream (the FP64 matrix
 * and vector instructions share one datapath: DESIGN.md section 4.2), which is why it is off. */
enum { TSFF_OPT_DENOM_MODE = 1, TSFF_OPT_LAUNCH_PLAN = 2, TSFF_OPT_DLM_BLOCKS = 3 };
int tsff_set_option(tsff_handle *h, int32_t key, int32_t value);
/* make sure the workspace holds B lineouts (calls grow it lazily; not inside graph capture) */
int tsff_reserve(tsff_handle *h, int32_t B);

/* wavelength axes (HOST pointers): the binned axes lamAxisE/lamAxisI [1024] in nm that the
 * reference returns per lineout (irf.py:75,125) -- they do not depend on the parameters. */
int tsff_get_axes(const tsff_handle *h, double *lamE, double *lamI);

/* Re(chi_e) table W[1640] and the Hermite table of ln fe for `n` distribution functions
 * fe[n][nvx] (form_factor.py:263-268 + ratintn.py).  W: [n][1640]. */
int tsff_chi_table(tsff_handle *h, const double *fe, int32_t n, double *W);

/* FormFactor.__call__ for B lineouts without the instrument chain: P[B][G][npts][n_angles]
 * (W/m^... arbitrary units of the reference).  `phys` holds PHYSICAL parameters [B][NP]
 * (no activation); fe is [B][nvx] (PER_LINEOUT), ignored otherwise. */
int tsff_form_factor(tsff_handle *h, int32_t feature, const double *phys, const double *fe,
                     int32_t B, double *P);

/* Adjoint of tsff_form_factor for an arbitrary seed Pbar = d loss / d P (device, [B][G][npts][n_angles]):
 *   grad_phys [B][NP] (device): d loss / d PHYSICAL parameters (the DLM order m carries none here: it acts through fe),
 *   grad_fe   [B][nvx] (device, or NULL): d loss / d fe[b][i] through both uses of the distribution function (Hermite
 *     ln f_e lookup and the ratintn table W), fe_mode PER_LINEOUT only.
 * The angular (ARTS) instrument chain hands back one adjoint per (wavelength, angle) point (tsff_ats_adjoint); this is
 * what reverse-mode JAX does for angular decks with a 1-D distribution function (inverse/loops.py:167-275). */
int tsff_form_factor_grad(tsff_handle *h, int32_t feature, const double *phys, const double *fe, int32_t B,
                          const double *Pbar, double *grad_phys, double *grad_fe);

/* FormFactor.calc_in_2D (core/physics/form_factor.py:449-587, with rotate :300-324 and calc_chi_vals :349-388)
 * for a 2-D electron distribution fe2d[nv][nv] on the grid linspace(-6 + dv/2, 6 - dv/2, nv) (first index = v_x;
 * one table shared by all lineouts when shared_fe != 0, else [B][nv][nv]): P[B][G][npts][n_angles].  `phys` holds
 * PHYSICAL parameters [B][NP]; ud_angle / va_angle are the drift and flow directions in degrees
 * (parameters.general.ud.angle / Va.angle).  Its adjoint is tsff_form_factor_2d_grad.  Parity with the reference is unpinned for this entry:
 * its golden vectors are not part of the reference source tree. */
int tsff_form_factor_2d(tsff_handle *h, int32_t feature, const double *phys, const double *fe2d, int32_t nv,
                        int32_t shared_fe, double ud_angle_deg, double va_angle_deg, int32_t B, double *P);
/* The same for the points [point_begin, point_end) of the flat (lineout, gradient point, wavelength, angle) list only;
 * the rest of P is left untouched (point_end < 0: to the end).  This is the unit of work the reference shards across
 * devices (parallel_calc_all_chi_vals, form_factor.py:431-447): each rank evaluates its range, the ranges are gathered. */
int tsff_form_factor_2d_range(tsff_handle *h, int32_t feature, const double *phys, const double *fe2d, int32_t nv,
                              int32_t shared_fe, double ud_angle_deg, double va_angle_deg, int32_t B,
                              int64_t point_begin, int64_t point_end, double *P);

/* tsff_form_factor_2d_range for ONE shared table (nv <= 256) that also keeps, in the handle, the projection record of
 * every point of the range (the projected distribution and the derivative sums of its rotation: proj2d_doubles(nv)
 * doubles per point): the adjoint that follows in a fit step (tsff_form_factor_2d_grad with saved_token = *token, same
 * feature / table size / point range, same phys and fe2d) then does no sampling of its own.  *token (HOST pointer, never 0 on
 * success) names this generation of records: a per-handle call counter mixed with the buffers, angles and range they were made
 * from.  Only the token of the LAST save is valid, and any later 2-D forward on the handle invalidates it.  Precondition the
 * token cannot check: the CONTENTS of phys / fe2d must not change between the save and the adjoint that presents its token. */
int tsff_form_factor_2d_save(tsff_handle *h, int32_t feature, const double *phys, const double *fe2d, int32_t nv,
                             double ud_angle_deg, double va_angle_deg, int32_t B, int64_t point_begin,
                             int64_t point_end, double *P, uint64_t *token);

/* Adjoint of tsff_form_factor_2d for ONE shared table (the 2-D path is never batched in the reference): given
 * Pbar = d loss / d P (device, [B][G][npts][n_angles]) ->
 *   grad_phys [B][NP] (device): d loss / d PHYSICAL parameters (Te, ne, lam, ne_gradient, Te_gradient, ud, Va, Ti, Z;
 *     amplitudes and A carry none here), and, when grad_fe2d != NULL,
 *   grad_fe2d [nv][nv] (device): d loss / d fe2d[i][j] -- the table adjoint of the rotate-and-project step (bicubic
 *     weights scattered by LDS atomics, ghost cells folded back).
 * [point_begin, point_end) (point_end < 0: to the end): the contributions of that slice of the flat point list only --
 * both adjoints are sums over points, so the ranks of a node each take a slice and all-reduce the two outputs.
 * saved_token != 0: the projections come from the records of the tsff_form_factor_2d_save that returned this token (see
 * there); a stale or foreign token is refused with -22 (EINVAL), records made from other buffers / angles / range with -2.
 * Replaces what JAX reverse mode gives the reference for angular fits (inverse/loops.py:167-275). */
int tsff_form_factor_2d_grad(tsff_handle *h, int32_t feature, const double *phys, const double *fe2d, int32_t nv,
                             double ud_angle_deg, double va_angle_deg, int32_t B, int64_t point_begin,
                             int64_t point_end, uint64_t saved_token, const double *Pbar, double *grad_phys,
                             double *grad_fe2d);

/* Angular (ARTS) instrument chain for one image P[G][npts][n_angles] (device; from tsff_form_factor_2d or, for a 1-D
 * distribution function, tsff_form_factor): FitModel.electron_spectrum "angular_full" branch
 * (core/physics/generate_spectra.py:193-216: weight-matrix product, iawfilter), add_ATS_IRF (core/physics/irf.py:5-47,
 * norm == 0), reduce_ATS_to_resunit (core/thomson_diagnostic.py:78-107).  Output ThryE (device)
 * [row_end - row_start][npts / lam_step].  Every pointer of tsff_ats_config is a HOST pointer, copied by tsff_ats_setup.
 * Convolution taps are in the flipped "same" form y[i] = sum_s taps[s] x[i + tap_off + s] (engine.binned_taps with ppp = 1). */
typedef struct tsff_ats_config {
  int32_t n_px;           /* rows of the weight matrix = angular pixels (1024) */
  const double *weights;  /* [n_px][n_angles] */
  int32_t n_taps_ang, tap_off_ang;
  const double *taps_ang; /* Gaussian over sas["angAxis"], ang_FWHM_ele / 2.3548 */
  int32_t n_taps_lam, tap_off_lam;
  const double *taps_lam; /* Gaussian over the wavelength axis, spect_FWHM_ele / 2.3548 */
  int32_t lam_step, ang_step; /* thomson_diagnostic.py:93-94 */
  int32_t row_start, row_end; /* data.lineouts.start / end (rows of the reduced image) */
  const double *lam_axis;     /* [npts] wavelength axis in nm */
} tsff_ats_config;
int tsff_ats_setup(tsff_handle *h, const tsff_ats_config *cfg);
int tsff_ats_spectrum(tsff_handle *h, const double *P, const double *e_amps /* device [rows] */, double lam, double amp1, double amp2, double *ThryE);
/* Reverse of tsff_ats_spectrum: given Ebar = d loss / d ThryE (device, [rows][npts / lam_step]) -> Pbar = d loss / d P
 * (device, [G][npts][n_angles]) and amp_bar[2] = d loss / d (amp1, amp2) (host).  `lam` enters the model only through
 * the blue / red split of the wavelength axis and carries no gradient, like in the reference's jnp.where. */
int tsff_ats_adjoint(tsff_handle *h, const double *P, const double *e_amps, double lam, double amp1, double amp2,
                     const double *Ebar, double *Pbar, double *amp_bar);

/* ThomsonScatteringDiagnostic.__call__: ThryE/ThryI [B][1024].  noise_* may be NULL (= 0).
 * params: normalised leaves [B][NP]; fe: [B][nvx] when fe_mode == PER_LINEOUT else NULL. */
int tsff_forward(tsff_handle *h, const double *params, const double *fe, const double *e_amps,
                 const double *i_amps, const double *noise_e, const double *noise_i, int32_t B,
                 double *ThryE, double *ThryI);

/* LossFunction.vg_loss: value and gradient.
 *   weights[3] (HOST): w_iaw, w_blue, w_red -- the factor each masked sum carries in the total
 *   loss, i.e. ion_loss_scale/(N_iaw*i_norm^2), c/(N_blue*e_norm^2), c/(N_red*e_norm^2) with N_* the
 *   number of fitted samples over the WHOLE (global, all ranks) batch and c = 1/2 when both EPW
 *   ranges are fitted (loss_function.py:262-264, 335-338).
 *   loss_terms[3] (device): un-weighted masked sums  S_iaw, S_blue, S_red  over these B lineouts;
 *   total loss = sum_k weights[k] * S_k (summed over ranks).
 *   grad (device) [B][NP]: d(total loss)/d(params[b][slot]) for slots with grad_mask[slot] != 0
 *   (HOST uint8 [NP]), 0 elsewhere.  The DLM order slot (TSFF_P_M) is differentiable when
 *   fe_mode == TSFF_FE_DLM (through the ln f_e and W tables).  ThryE/ThryI may be NULL. */
int tsff_loss_grad(tsff_handle *h, const double *params, const double *fe, const double *e_data,
                   const double *i_data, const double *e_amps, const double *i_amps,
                   const double *noise_e, const double *noise_i, int32_t B,
                   const double *weights, const uint8_t *grad_mask, double *loss_terms,
                   double *grad, double *ThryE, double *ThryI);

/* tsff_loss_grad delivering the gradient the way the optimiser consumes it -- the flat vector of ravel_pytree that
 * scipy L-BFGS-B iterates on (inverse/loops.py:40-54: trainable leaves outermost, lineouts innermost) over the GLOBAL
 * batch of a lineout-sharded fit -- written straight into the buffer of the step's one all-reduce:
 *   packed (device) [3 + n_active * B_global] = [S_iaw, S_blue, S_red | g[k][b]],  k < n_active, b < B_global;
 *   active_slots (HOST) [n_active]: the parameter slot of each row k (the ravel order of the trainable leaves);
 *   this call fills the loss sums of ITS B lineouts and the columns [b_offset, b_offset + B) of every row, and writes
 *   zero to every other column: summed over the ranks (in place, no memset between steps) the buffer is the full
 *   loss and gradient on every rank; with one rank (B_global == B, b_offset == 0) it is what LossFunction.vg_loss
 *   returns after one device-to-host copy.  weights / grad_mask as for tsff_loss_grad. */
int tsff_loss_grad_packed(tsff_handle *h, const double *params, const double *fe, const double *e_data,
                          const double *i_data, const double *e_amps, const double *i_amps, const double *noise_e,
                          const double *noise_i, int32_t B, const double *weights, const uint8_t *grad_mask,
                          const int32_t *active_slots, int32_t n_active, int64_t B_global, int64_t b_offset,
                          double *packed, double *ThryE, double *ThryI);

/* The same plus the gradient w.r.t. the tabulated distribution function itself: grad_fe [B][nvx] (device) =
 * d loss / d fe[b][i], for fe_mode == TSFF_FE_PER_LINEOUT.  This is what equinox.filter_value_and_grad returns for the
 * leaves of a free-form distribution (Arbitrary1V.fval, core/modules/distribution_functions/base.py:157-204, filter
 * spec :462-471) before the generator's own chain rule, which stays on the host.  The adjoint runs through both uses of
 * f_e: the Hermite interpolant of ln f_e at the phase velocity (form_factor.py:256) and the Re(chi_e) table
 * (form_factor.py:263-268, ratintn.py) via one transposed GEMM with the constant log-ratio table. */
int tsff_loss_grad_fe(tsff_handle *h, const double *params, const double *fe, const double *e_data, const double *i_data,
                      const double *e_amps, const double *i_amps, const double *noise_e, const double *noise_i, int32_t B,
                      const double *weights, const uint8_t *grad_mask, double *loss_terms, double *grad, double *grad_fe,
                      double *ThryE, double *ThryI);

/* The packed buffer of a free-form f_e fit step from the outputs of tsff_loss_grad_fe (all device pointers):
 * packed = [loss_terms[0..3) | rows x B_global], rows = the n_active scalar leaves (grad[b][active_slots[k]]) followed by the
 * nvx values d loss / d fe[b][i] -- the ravel order of the optimiser's flat vector before the generator's chain rule (which
 * stays on the host) --, this rank's columns [b_offset, b_offset + B) filled and every other column ZERO, so that the buffer
 * is all-reduced in place like the one of tsff_loss_grad_packed (inverse/loops.py:40-54; form_factor.py:431-447 for the sharding). */
int tsff_pack_fe_rows(tsff_handle *h, const double *loss_terms, const double *grad, const double *grad_fe, int32_t B,
                      const int32_t *active_slots, int32_t n_active, int64_t B_global, int64_t b_offset, double *packed);

/* LossFunction.array_loss: per-lineout masked sums with the theory spectrum as denominator
 * ((d-t)^2/t, loss_function.py:320-321).  sums [B][3] = S_iaw, S_blue, S_red per lineout;
 * sqdev_e / sqdev_i [B][1024] (may be NULL) = nan_to_num'ed error arrays (:238,250,264). */
int tsff_array_loss(tsff_handle *h, const double *params, const double *fe, const double *e_data,
                    const double *i_data, const double *e_amps, const double *i_amps,
                    const double *noise_e, const double *noise_i, int32_t B, double *sums,
                    double *sqdev_e, double *sqdev_i, double *ThryE, double *ThryI);

/* HIP-event timing of the main kernel of tsff_forward / tsff_loss_grad(_packed, _fe) / tsff_array_loss (k_spectrum or
 * k_spectrum_fused) and of tsff_form_factor_2d(_range, _save) (k_form_factor_2d) on the handle's stream.  tsff_enable_timing(h, ring) keeps one event pair per
 * launch in a ring of `ring` entries (0 disables); tsff_kernel_times returns the durations [ms] of
 * the most recent min(ring, launches, max_n) launches, oldest first, and synchronises on them. */
int tsff_enable_timing(tsff_handle *h, int32_t ring);
int tsff_kernel_times(tsff_handle *h, float *ms, int32_t max_n, int32_t *n_out);

/* micro-benchmark: sustained FP64 vector FMA rate of this device in TFLOP/s (the roof the path is bound by;
 * AMD's datasheet figure for MI355X is 78.6).  Synchronous. */
int tsff_fp64_fma_peak(tsff_handle *h, double *tflops);
/* the same on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), the roof of k_wgemm */
int tsff_fp64_mfma_peak(tsff_handle *h, double *tflops);
/* micro-benchmark: sustained read rate of the vector L1 (every wavefront re-reads a 16 KB window with 16-byte loads) in
 * TB/s over the whole device -- the roof of the 2-D sampler when its table is read through L1/L2 (nv > 128) */
int tsff_l1_read_peak(tsff_handle *h, double *tbps);

#ifdef __cplusplus
}
#endif
#endif /* TSFF_H */

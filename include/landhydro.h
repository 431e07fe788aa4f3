/*
 * landhydro.h -- C ABI of liblandhydro_hip.so: the MI355X (gfx950) batched
 * soil-column tendency path behind LandHydrology.jl's SoilModel API.
 *
 * What it replaces.  LandHydrology.jl has no FFI; its seam is the Julia closure
 *     rhs!(dY, Y, Ya, t) -> dY
 * returned by make_rhs(model::SoilModel) (src/SoilModel/right_hand_side.jl:33-44)
 * and handed to DiffEqBase.ODEProblem (src/Simulations/simulation.jl:58-63).
 * A Julia shim builds the same closure on top of the calls
 * below with ccall.  Every entry point names the reference interface it stands
 * in for (file:line under the reference tree).
 *
 * Conventions
 *   - plain C types only; all entry points return 0 on success or a negative
 *     LH_E* code, never throw; lh_last_error() gives the message (the Julia shim
 *     turns it into error(msg), matching the reference's ArgumentError/error paths).
 *   - a context owns ncols independent columns of nlev cells on ONE device; one
 *     process per GPU (multi-GPU = block partition of columns over processes).
 *   - states are device-resident and owned by the library; host pointers passed
 *     to upload/download are borrowed for the call only.
 *   - cells are indexed bottom -> top (test/SoilModel/coupled.jl:198); host
 *     arrays are addressed a[col*col_stride + lev*lev_stride] (element strides),
 *     so both parent(field)-style level-fastest columns and column-fastest
 *     planes are accepted.
 *   - parameters cross the ABI as double and are rounded to the working type
 *     exactly where the Julia code applies FT(...).
 *   - calls enqueue on the context's HIP stream; lh_download, lh_stable_dt,
 *     lh_get_status and lh_synchronize wait for it.  A context is single-owner
 *     (one host thread at a time), like the reference closure which mutates Ya.
 */
#ifndef LANDHYDRO_H
#define LANDHYDRO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LH_VERSION_MAJOR 0
#define LH_VERSION_MINOR 1

/* status codes */
enum {
    LH_OK = 0,
    LH_EINVAL = -1,    /* bad argument (ArgumentError in the reference) */
    LH_ENODEVICE = -2, /* no usable HIP device / HIP runtime failure */
    LH_ENOMEM = -3,
    LH_EMODEL = -4, /* model / boundary-condition combination with no reference method */
    LH_ESTATE = -5  /* state handle lacks a variable this call needs */
};

/* working type FT of SoilModel{FT} (src/SoilModel/models.jl:90-135) */
enum { LH_F32 = 0, LH_F64 = 1 };

/* which make_rhs method (right_hand_side.jl): :118-186, :192-263, :269-369 */
enum {
    LH_MODEL_RICHARDS = 0, /* PrescribedTemperatureModel + SoilHydrologyModel */
    LH_MODEL_HEAT = 1,     /* SoilEnergyModel + PrescribedHydrologyModel      */
    LH_MODEL_COUPLED = 2   /* SoilEnergyModel + SoilHydrologyModel            */
};

/* AbstractBC subtypes, src/SoilModel/boundary_conditions.jl:19-77 */
enum { LH_BC_NONE = 0, LH_BC_FLUX = 1, LH_BC_DIRICHLET = 2, LH_BC_FREE_DRAINAGE = 3 };
/* domain.boundary_tags, src/Domains/domain.jl:31 */
enum { LH_FACE_BOTTOM = 0, LH_FACE_TOP = 1 };
/* SoilComponentBC fields, boundary_conditions.jl:95-101 */
enum { LH_COMP_ENERGY = 0, LH_COMP_HYDROLOGY = 1 };
/* AbstractConductivityFactor, SoilWaterParameterizations.jl:38-65 */
enum { LH_FACTOR_NONE = 0, LH_FACTOR_ON = 1 };

/* variables of Y / dY / Ya (src/SoilModel/initial_conditions.jl:14-17, 85-89) */
enum {
    LH_VAR_VARTHETA_L = 0, /* augmented liquid fraction (Y.soil, or Ya.soil for HEAT) */
    LH_VAR_THETA_I = 1,    /* ice fraction                                             */
    LH_VAR_RHOE_INT = 2,   /* volumetric internal energy                               */
    LH_VAR_T = 3,          /* Ya.soil.T (RICHARDS); diagnostics: temperature           */
    LH_NVARS = 4
};
/* masks for lh_state_create */
#define LH_MASK(var) (1u << (var))
/* diagnostic states reuse the slots: 0 = K, 1 = psi, 2 = kappa, 3 = T */
enum { LH_DIAG_K = 0, LH_DIAG_PSI = 1, LH_DIAG_KAPPA = 2, LH_DIAG_T = 3 };

/* transcendental policy: FAST = the gfx950 log2/exp2 kernels of the product path;
 * LIBM = ocml pow/exp (<= 1 ulp), kept for parity debugging */
enum { LH_MATH_FAST = 0, LH_MATH_LIBM = 1 };

/* per-column parameter ids for lh_set_percol_param (BASELINE config 5) */
enum {
    LH_PC_VG_N = 0,
    LH_PC_VG_ALPHA = 1,
    LH_PC_VG_THETA_R = 2,
    LH_PC_VG_KSAT = 3,
    LH_PC_NU = 4,
    LH_PC_S_S = 5,
    LH_PC_COUNT = 6
};

typedef struct lh_ctx lh_ctx;     /* one SoilModel on one device */
typedef struct lh_state lh_state; /* one FieldVector (Y, dY or Ya) */

/* Column{FT}(zlim, nelements) (src/Domains/domain.jl:12-33) x ncols, plus the
 * SoilModel type parameters that select the method. */
typedef struct {
    int64_t ncols;  /* independent columns owned by this context (>= 1)       */
    int32_t nlev;   /* Column.nelements (>= 1)                                */
    int32_t dtype;  /* LH_F32 | LH_F64                                        */
    double zmin;    /* Column.zlim[1]; zmin < zmax (domain.jl:30)             */
    double zmax;    /* Column.zlim[2]                                         */
    int32_t model;  /* LH_MODEL_*                                             */
    int32_t device; /* HIP device ordinal; -1 = current device                */
    void* stream;   /* hipStream_t to enqueue on; NULL = library-owned stream */
} lh_config;

/* CLIMAParameters constants read by src/SoilModel/SoilHeatParameterizations.jl:12-13
 * (Planet: rho_cloud_liq, rho_cloud_ice, cp_l, cp_i, T_0, LH_f0; Microphysics: K_therm). */
typedef struct {
    double rho_liq, rho_ice, cp_l, cp_i, T_0, LH_f0, K_therm;
} lh_earth_params;

/* SoilParams{FT}, src/SoilModel/parameters.jl:11-43 (physics fields) */
typedef struct {
    double nu, S_s, nu_ss_gravel, nu_ss_om, nu_ss_quartz, rho_c_ds, kappa_solid, rho_p,
        kappa_sat_unfrozen, kappa_sat_frozen, a, b, kappa_dry_parameter;
} lh_soil_params;

/* vanGenuchten{FT}, SoilWaterParameterizations.jl:150-169 (m = 1 - 1/n is derived) */
typedef struct {
    double n, alpha, theta_r, Ksat;
} lh_vg_params;

/* PrescribedAtmosForcing{FT} (boundary_conditions.jl:119-132: the first six fields), the
 * roughness lengths it reads from SoilParams (parameters.jl:38-41), and the CLIMAParameters
 * constants compute_turbulent_surface_fluxes (:553-620) consumes on top of lh_earth_params
 * (Planet: R_v, R_d, grav, cp_d, cp_v, LH_v0, T_triple, press_triple; SubgridScale:
 * von_karman_const).  Inputs, never kernel literals. */
typedef struct {
    double u_atm, theta_atm, z_atm, theta_scale, rho_a_sfc, q_atm;
    double z_0m, z_0s;
    double R_v, R_d, grav, cp_d, cp_v, LH_v0, T_triple, press_triple, von_karman;
} lh_atmos_forcing;

/* ---- lifetime ------------------------------------------------------------ */

/* SoilModel(FT; domain, energy_model, hydrology_model, ...) (models.jl:115-135).
 * Starts with the reference defaults: loam vanGenuchten, default SoilParams,
 * NoEffect factors, NoBC on every face/component; earth parameters must be set
 * before lh_rhs when the model has an energy component. */
int lh_create(lh_ctx** out, const lh_config* cfg);
int lh_destroy(lh_ctx* ctx);
/* message of the last failing call on ctx (ctx == NULL: last lh_create failure) */
const char* lh_last_error(const lh_ctx* ctx);
int lh_version(void);

/* ---- parameters ----------------------------------------------------------- */

int lh_set_earth_params(lh_ctx*, const lh_earth_params*);   /* SoilModel.earth_param_set */
int lh_set_soil_params(lh_ctx*, const lh_soil_params*);     /* SoilModel.soil_param_set  */
int lh_set_vg_params(lh_ctx*, const lh_vg_params*);         /* hydrology.hydraulic_model */
/* per-column override of one parameter (host array of ncols doubles); NULL
 * restores the scalar.  Build extension: the reference has one column. */
int lh_set_percol_param(lh_ctx*, int32_t param_id, const double* host_values);
/* SoilHydrologyModel.viscosity_factor / .impedance_factor (models.jl:28-33) */
int lh_set_conductivity_factors(lh_ctx*, int32_t viscosity_kind, double gamma, double T_ref,
                                int32_t impedance_kind, double Omega);
/* SoilColumnBC(top = SoilComponentBC(energy=, hydrology=), bottom = ...)
 * (boundary_conditions.jl:95-161).  value = VerticalFlux.flux, or
 * Dirichlet.state_value(t) evaluated by the host shim for the current time;
 * percol_values (ncols doubles) overrides value per column when not NULL. */
int lh_set_bc(lh_ctx*, int32_t face, int32_t component, int32_t kind, double value,
              const double* percol_values);
/* SoilColumnBC(top = PrescribedAtmosForcing{FT}(...), bottom = ...) (boundary_conditions.jl:
 * 119-161, 516-533): the TOP face of every column is driven by Monin-Obukhov surface fluxes
 * computed on the device from the top cell's (vartheta_l, theta_i, T) before every tendency
 * evaluation (compute_turbulent_surface_fluxes, :553-620; kernel in csrc/lh_atmos.hpp); the top
 * entries of lh_set_bc are ignored while it is set.  Only LH_MODEL_COUPLED has a method
 * (SoilEnergyModel + SoilHydrologyModel; LH_EMODEL otherwise, like the reference's MethodError),
 * and only the top face (:523-528).  forcing == NULL removes it.  percol: NULL, or [3][ncols]
 * doubles overriding u_atm, theta_atm, q_atm per column (build extension).  Where the
 * Monin-Obukhov system has no root the fluxes are NaN and bit 1 of lh_get_status is set.
 * PARITY UNPINNED beyond the reference's equilibrium invariant
 * (test/SoilModel/test_prescribed_atmos_bc.jl:75-79): SurfaceFluxes.jl / Thermodynamics.jl are
 * not part of the reference tree. */
int lh_set_atmos_forcing(lh_ctx*, const lh_atmos_forcing* forcing, const double* percol);
/* compute_turbulent_surface_fluxes.(energy, hydrology, model, vartheta_l, theta_i, T) for n
 * top-cell states given as host arrays of doubles (as test_prescribed_atmos_bc.jl:92-100 calls
 * it): heat_flux[n] and water_flux[n] (volume flux, positive upward) in doubles.  Scalar soil
 * and forcing parameters of the context; evaluated on the device.  Synchronises. */
int lh_atmos_surface_fluxes(lh_ctx*, int64_t n, const double* vartheta_l, const double* theta_i,
                            const double* T, double* heat_flux, double* water_flux);
/* 0 (default): bottom-face hydrology Dirichlet flux exactly as the reference
 * writes it (boundary_conditions.jl:395-398); 1: physically consistent sign. */
int lh_set_bottom_sign_consistent(lh_ctx*, int32_t flag);
/* LH_MATH_* (default LH_MATH_FAST; the environment variable LH_MATH=libm
 * selects LH_MATH_LIBM at lh_create) */
int lh_set_math_mode(lh_ctx*, int32_t mode);
/* launch-shape override for measurements, e.g. "block=128" (threads per
 * workgroup: 64, 128, 192 or 256); "" restores the defaults.  The environment
 * variable LH_TUNE is read the same way at lh_create.  Results do not depend
 * on it. */
int lh_set_tuning(lh_ctx*, const char* spec);

/* ---- states ---------------------------------------------------------------- */

/* Fields.FieldVector analogue.  var_mask selects the planes to allocate
 * (LH_MASK(...)); 0 = the prognostic set of the model
 * (initial_conditions.jl:85-89). */
int lh_state_create(lh_ctx*, uint32_t var_mask, lh_state** out);
int lh_state_destroy(lh_ctx*, lh_state*);
/* host -> device / device -> host for one variable, FT elements */
int lh_upload(lh_ctx*, lh_state*, int32_t var, const void* host, int64_t lev_stride,
              int64_t col_stride);
int lh_download(lh_ctx*, const lh_state*, int32_t var, void* host, int64_t lev_stride,
                int64_t col_stride);
/* one level of one variable, FT[ncols] (level 0 = bottom cell, nlev-1 = top cell):
 * what interior_values(X, face, cs) hands to a host-evaluated boundary condition
 * (boundary_conditions.jl:174-186, 516-533) -- ncols values instead of a plane */
int lh_download_level(lh_ctx*, const lh_state*, int32_t var, int32_t level, void* host);
/* A LEVEL-UNIFORM variable: host = FT[nlev], one value per level (bottom first), the same in every
 * column -- what Ya.soil.T .= T_profile.(zc, t) / theta_l_profile / theta_i_profile produce
 * (make_update_aux, right_hand_side.jl:54-81: functions of z and t only) and what an initial
 * condition f(z) gives.  nlev numbers cross PCIe instead of a plane, asynchronously (pinned staging;
 * the call does not wait).  The column kernels take a level-uniform prescribed field of Ya (T of the
 * Richards viscosity factor, vartheta_l and theta_i of the heat-only model) from LDS beside z and
 * read no plane for it (8 B per cell less on the viscosity configuration); every other use
 * (download, a prognostic variable, diagnostics, device pointers) first broadcasts the profile into
 * the plane on the device.  Results are bitwise those of uploading the broadcast plane. */
int lh_upload_profile(lh_ctx*, lh_state*, int32_t var, const void* host_nlev);
int lh_state_fill(lh_ctx*, lh_state*, int32_t var, double value);
int lh_state_copy(lh_ctx*, lh_state* dst, const lh_state* src);
/* zero-copy access: device pointer of a plane and its element strides
 * (column-fastest planes: col_stride == 1). */
int lh_state_device_ptr(lh_ctx*, const lh_state*, int32_t var, void** dptr,
                        int64_t* lev_stride, int64_t* col_stride);
/* While a plane's pointer is out, the caller may write through it at any time, so the library assumes
 * nothing about that plane: its contents are re-read at every launch (never treated as a plane of known
 * zeros, whatever lh_state_fill / lh_state_copy / lh_rhs wrote into it since), and the identically zero
 * d theta_i of a tendency state is re-stored at every lh_rhs.  lh_state_release_ptr ends that: the
 * caller promises not to write through pointers obtained before the call (var = -1: every plane of
 * the state).  Pointers stay valid addresses until the state is destroyed or moved (lh_tune_placement). */
int lh_state_release_ptr(lh_ctx*, lh_state*, int32_t var);
/* coordinates(cs) (right_hand_side.jl:7-8): cell-centre z, nlev doubles */
int lh_coordinates(const lh_ctx*, double* zc_host);

/* ---- the hot path ----------------------------------------------------------- */

/* rhs!(dY, Y, Ya, t) (right_hand_side.jl:37-42).  Ya may be NULL when the model
 * reads no auxiliary field (COUPLED; RICHARDS with NoEffect viscosity).  The
 * prescribed-profile update of make_update_aux (:54-96) is the host shim's job:
 * it uploads Ya when its closures depend on t. */
int lh_rhs(lh_ctx*, double t, const lh_state* Y, const lh_state* Ya, lh_state* dY);

/* lh_rhs plus, from the same pass over the state, this rank's stable-step bound
 * (the rule of lh_stable_dt) left in device memory (one FT value): the input of
 * the RCCL min all-reduce costs no second sweep over the columns.  With a
 * communicator attached (lh_comm_init) that all-reduce is enqueued right behind
 * the launch and the value is the global minimum.  The fused bound accumulates the
 * face diffusivities in Float32 whatever the working type (a safety estimate under
 * a Courant factor): it agrees with lh_stable_dt to 1e-6 relative, and -- a maximum
 * being exact -- does not depend on how the columns are dealt to waves or ranks. */
int lh_rhs_stable_dt(lh_ctx*, double t, const lh_state* Y, const lh_state* Ya, lh_state* dY,
                     double courant, void* dt_device_ft);

/* boundary_fluxes(X, bc::SoilComponentBC, face, model, cs, t) (boundary_conditions.jl:470-489; with a
 * PrescribedAtmosForcing at the top: :516-533) for every column: the pair (f_rhoe_int, f_vartheta_l)
 * of ONE face -- LH_FACE_BOTTOM or LH_FACE_TOP -- from the state of the cell next to it, positive in
 * +z: exactly the two SetValue fluxes the tendency launch uses (same device functions, same bits), for
 * callers that budget water and energy.  Boundary values are those of the last lh_set_bc (the host
 * shim evaluates Dirichlet closures at t first, as for lh_rhs).  f_energy / f_water: ncols doubles
 * each (either may be NULL); NaN where the component has no boundary condition (NoBC: `nothing` in
 * the reference).  Synchronises. */
int lh_boundary_fluxes(lh_ctx*, const lh_state* Y, const lh_state* Ya, double t, int32_t face,
                       double* f_energy, double* f_water);

/* centre fields K, psi, kappa, T of the same pointwise stage
 * (right_hand_side.jl:156-167, 291-314) into a 4-plane state (LH_DIAG_*) */
int lh_diagnostics(lh_ctx*, const lh_state* Y, const lh_state* Ya, lh_state* out);

/* solve(prob, SSPRK33(), dt = dt) (simulation.jl:58-87; coupled.jl:94): nsteps
 * fixed-dt steps, Y advanced in place on the device.  bc_stage_values is NULL or
 * [nsteps][3][2][2] doubles (step, stage, face, component): Dirichlet/flux
 * values at the stage times t, t+dt, t+dt/2. */
int lh_step_ssprk33(lh_ctx*, lh_state* Y, const lh_state* Ya, double t, double dt,
                    int64_t nsteps, const double* bc_stage_values);

/* Which engine lh_step_ssprk33 runs a call of nsteps steps with (no counterpart in the reference;
 * for logs, benches and tests -- results do not depend on it: the engines are bitwise equal):
 * LH_ENGINE_FUSED_STAGES = three fused-stage launches per step (rhs_kernel MODE 1-3),
 * LH_ENGINE_COLUMN_STEPPER = ONE launch of the persistent column stepper for the whole call (state in
 * registers).  per_stage_boundary_values: the call would pass bc_stage_values != NULL.
 * Returns the engine (>= 0) or a negative LH_E* code. */
enum { LH_ENGINE_FUSED_STAGES = 0, LH_ENGINE_COLUMN_STEPPER = 1 };
int lh_step_engine(const lh_ctx*, int64_t nsteps, int32_t per_stage_boundary_values);

/* One stage of that step (OrdinaryDiffEq SSPRK33, Shu-Osher form), for hosts that must refresh
 * Ya between the stage evaluations: the reference's rhs! calls update_aux_en!/update_aux_hydr!
 * with the STAGE time at every evaluation (right_hand_side.jl:37-42, 54-81), so a prescribed
 * T_profile(z, t) / theta_l_profile(z, t) that really depends on t has to be re-evaluated by
 * the host shim and uploaded three times per step:
 *   stage 1 (time t):        U = Y + dt f(Y; Ya)
 *   stage 2 (time t + dt):   U = (3 Y + U + dt f(U; Ya)) / 4
 *   stage 3 (time t + dt/2): Y = (Y + 2 U + 2 dt f(U; Ya)) / 3
 * U: a state with the model's prognostic planes (its theta_i plane is not used: theta_i is read
 * from Y and never changes).  bc_values: NULL or [2][2] doubles (face, component) for this
 * stage.  Three calls are bitwise one lh_step_ssprk33 step with the same Ya. */
int lh_ssprk33_stage(lh_ctx*, int32_t stage, lh_state* Y, lh_state* U, const lh_state* Ya, double dt,
                     const double* bc_values);

/* Placement tuning -- no counterpart in the reference (host arrays have no such
 * effect).  The speed of the column launch on MI355X depends on where in HBM the
 * planes it streams together sit relative to each other (a few discrete rates,
 * up to ~10 % apart, reproducible for a given set of plane slots, also seen by a plain
 * device copy; cause not identified).  This call times the real launch on the
 * data in Y with the WRITTEN planes in up to max_candidates (0 = default 6) different slot
 * sets -- all written planes together, then each on its own -- and keeps the fastest:
 *   dY != NULL: the tendency launch of lh_rhs / lh_rhs_stable_dt writing dY;
 *   dY == NULL: the fused SSPRK33 stages of lh_step_ssprk33* writing the context's
 *               internal stage state (the trial stages run with dt = 0).
 * With LH_PLACE_MOVE_INPUT in flags the planes of Y are then tried in other slots
 * the same way (contents copied).  The values in Y never change; the contents of dY
 * are unspecified afterwards; device pointers obtained earlier from
 * lh_state_device_ptr for a moved state (dY; Y with LH_PLACE_MOVE_INPUT) are
 * stale.  Results of later launches do not depend on the placement.  One-off cost:
 * ~12 launches per candidate and (max_candidates-1) temporary copies of the moved
 * planes, never more than 25 % of the free device memory (LH_TUNE place_mem=PCT).
 * ms_before/ms_after (optional): launch time with the original and the
 * chosen placement.  Synchronises.  Always explicit: no other entry point tunes. */
#define LH_PLACE_MOVE_INPUT 1u
int lh_tune_placement(lh_ctx*, lh_state* Y, const lh_state* Ya, lh_state* dY, int max_candidates,
                      uint32_t flags, float* ms_before, float* ms_after);

/* One SSPRK33 step whose dt is read from DEVICE memory (one FT value, e.g. the
 * output of lh_stable_dt_device after an RCCL min all-reduce): adaptive stepping
 * across ranks without a host round trip.  bc_stage_values: NULL or [3][2][2]. */
int lh_step_ssprk33_device_dt(lh_ctx*, lh_state* Y, const lh_state* Ya, double t,
                              const void* dt_device_ft, const double* bc_stage_values);

/* nsteps ADAPTIVE SSPRK33 steps with nothing leaving the device (build-defined: the reference
 * steps with a fixed user dt, simulation.jl:34-70).  Per step, all on the context's stream:
 *   f(Y) and the stable-step bound of Y in ONE launch (as lh_rhs_stable_dt; the RCCL min over
 *   ranks follows when a communicator is attached); dt = min(bound, dt_max) (dt_max <= 0: no
 *   cap), *elapsed += dt; then stages 2 and 3 -- stage 2 takes (Y, f(Y)) and forms
 *   U1 = Y + dt f(Y) in registers, so f(Y) is evaluated once, not twice: three evaluations of f
 *   per step instead of the four of lh_rhs_stable_dt + lh_step_ssprk33_device_dt, and bitwise
 *   their result.
 * Boundary values are the constants of lh_set_bc (a time-dependent Dirichlet closure needs the
 * stage times on the host: use the per-step calls).  dt_device_ft: one FT in device memory,
 * holds the last step's dt afterwards; elapsed_device_ft: NULL or one FT in device memory that
 * accumulates the simulated time (the caller zeroes it, on a stream ordered before this call: the
 * context's own stream is non-blocking).  A step whose bound is not a positive finite number is
 * taken with dt = 0 and sets bit 2 of lh_get_status.  Does not synchronise. */
int lh_step_ssprk33_adaptive(lh_ctx*, lh_state* Y, const lh_state* Ya, double t, double courant,
                             double dt_max, int64_t nsteps, void* dt_device_ft, void* elapsed_device_ft);

/* Build-defined stable step (the reference uses a fixed user dt):
 * courant*dz^2 / max over owned faces of the face diffusivities
 * ((K_lo+K_hi)/2 * max dpsi/dvl, (kappa_lo+kappa_hi)/2 / min rho_c_s; boundary
 * cells and Dirichlet faces with their own coefficients).
 * _device leaves the FT result in device memory (for an RCCL min all-reduce
 * across ranks without a host round trip); the host form synchronises. */
int lh_stable_dt(lh_ctx*, const lh_state* Y, const lh_state* Ya, double courant,
                 double* dt_host);
int lh_stable_dt_device(lh_ctx*, const lh_state* Y, const lh_state* Ya, double courant,
                        void* dt_device_ft);

/* ---- multi-GPU: block partition + the one collective ------------- */

/* No counterpart in the reference (one column, one process, fixed user dt:
 * src/Simulations/simulation.jl:34-70).  Columns are independent, so an ensemble
 * of ncols_global columns is block-partitioned over ranks (one process and one
 * context per GPU) with no halo and no data-path exchange.  The only collective
 * of the path is the minimum of the ranks' stable-step bounds: ONE FT value,
 * ncclAllReduce(count = 1, ncclMin) over RCCL/xGMI on the context's stream. */

/* columns [*lo, *hi) owned by `rank` of `nranks` (the first ncols_global % nranks
 * ranks own one more); per-column parameter and BC arrays partition the same way */
int lh_block_range(int64_t ncols_global, int32_t rank, int32_t nranks, int64_t* lo, int64_t* hi);

/* ncclGetUniqueId: fills LH_COMM_ID_BYTES bytes on ONE rank; the host ships them
 * to the others (MPI_Bcast, a file, torch.distributed ...) before lh_comm_init */
#define LH_COMM_ID_BYTES 128
int lh_comm_unique_id(void* id_out);
/* ncclCommInitRank on the context's device (collective: every rank calls it).
 * While a communicator is attached, lh_rhs_stable_dt, lh_stable_dt_device and
 * lh_stable_dt deliver the GLOBAL minimum over all ranks (the all-reduce is
 * enqueued on the context's stream behind the local reduction: no host round
 * trip); everything else stays rank-local. */
int lh_comm_init(lh_ctx*, int32_t rank, int32_t nranks, const void* unique_id);
int lh_comm_destroy(lh_ctx*);
/* rank / nranks of the attached communicator (0 / 1 when there is none) */
int lh_comm_info(const lh_ctx*, int32_t* rank, int32_t* nranks);
/* in-place min all-reduce of one FT value in device memory over the attached
 * communicator (no-op without one): for hosts that form their own step bound */
int lh_allreduce_min(lh_ctx*, void* value_device_ft);

/* ---- status / timing -------------------------------------------------------- */

/* bit 0: a non-finite tendency was produced since the last call (the reference
 * would have raised DomainError from `^`); bit 1: the Monin-Obukhov system of the
 * prescribed-atmosphere BC had no root in some column; bit 2: a step of
 * lh_step_ssprk33_adaptive found no positive finite step bound (no positive diffusivity anywhere and
 * no dt_max, or a NaN) and was taken with dt = 0; synchronises and clears. */
int lh_get_status(lh_ctx*, uint32_t* flags);
int lh_synchronize(lh_ctx*);
/* Streaming ceiling of the column launch on a given set of planes (measurement aid, no
 * counterpart in the reference): a kernel with rhs_kernel's access pattern and no arithmetic
 * reads the planes of `in` selected by read_mask and writes the planes of `out` selected by
 * write_mask (LH_MASK bits), `reps` launches (<= 0: 20); *ms_per_launch = their average
 * duration by HIP events.  The selected planes of `out` hold unspecified values afterwards.
 * Synchronises. */
int lh_stream_probe(lh_ctx*, const lh_state* in, uint32_t read_mask, lh_state* out, uint32_t write_mask,
                    int reps, float* ms_per_launch);
/* HIP-event stopwatch on the context's stream */
int lh_timer_start(lh_ctx*);
int lh_timer_stop(lh_ctx*, float* elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* LANDHYDRO_H */

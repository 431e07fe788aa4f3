/*
 * lh_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the LandHydrology.jl SoilModel tendency path
 * (make_rhs / rhs!(dY, Y, Ya, t)) for Float32 and Float64.  It exists to CHECK
 * the HIP product path; nothing in the product (landhydrology.jl_amd/) may
 * include, link, call or execute it.  Allowed users: tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py.
 *
 * Parity status: the reference is Julia and cannot be run in the build
 * container (no julia binary, no depot).  The restatement is pinned by the
 * reference's own known-answer tests (tests/test_oracle_pins.py: K1..K7, K9 of
 * SURVEY.md section 8c); no output of the running reference exists to compare
 * against.
 *
 * All citations are file:line under the reference tree (src/... or test/...).
 *
 * Conventions
 *   cells i = 0..n-1 bottom -> top (test/SoilModel/coupled.jl:198),
 *   faces k = 0..n, dz = (zmax-zmin)/n (src/Domains/domain.jl:64).
 *   Parameters cross this interface as double and are rounded to the working
 *   type FT exactly where the Julia code applies FT(...).
 */
#ifndef LH_ORACLE_H
#define LH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { LHO_MODEL_RICHARDS = 0, LHO_MODEL_HEAT = 1, LHO_MODEL_COUPLED = 2 };
enum { LHO_BC_NONE = 0, LHO_BC_FLUX = 1, LHO_BC_DIRICHLET = 2, LHO_BC_FREE_DRAINAGE = 3 };
/* (PrescribedAtmosForcing is a property of the whole top face: lho_model.atmos_on) */
enum { LHO_FACE_BOTTOM = 0, LHO_FACE_TOP = 1 };
enum { LHO_COMP_ENERGY = 0, LHO_COMP_HYDROLOGY = 1 };
enum { LHO_FACTOR_NONE = 0, LHO_FACTOR_ON = 1 };

/* CLIMAParameters.Planet / Atmos.Microphysics constants consumed by
 * src/SoilModel/SoilHeatParameterizations.jl:12-13.  Inputs, never literals. */
typedef struct {
    double rho_liq, rho_ice, cp_l, cp_i, T_0, LH_f0, K_therm;
} lho_earth_params;

/* src/SoilModel/parameters.jl:11-43 (the 13 physics fields). */
typedef struct {
    double nu, S_s, nu_ss_gravel, nu_ss_om, nu_ss_quartz, rho_c_ds, kappa_solid,
        rho_p, kappa_sat_unfrozen, kappa_sat_frozen, a, b, kappa_dry_parameter;
} lho_soil_params;

/* src/SoilModel/SoilWaterParameterizations.jl:150-169; m = 1 - 1/n is derived. */
typedef struct {
    double n, alpha, theta_r, Ksat;
} lho_vg_params;

/* SoilWaterParameterizations.jl:46-65 */
typedef struct {
    int32_t viscosity_kind; /* LHO_FACTOR_* : NoEffect | TemperatureDependentViscosity */
    int32_t impedance_kind; /* LHO_FACTOR_* : NoEffect | IceImpedance */
    double gamma, T_ref, Omega;
} lho_cond_factors;

/* PrescribedAtmosForcing{FT} (src/SoilModel/boundary_conditions.jl:119-132), the roughness
 * lengths it reads from SoilParams (parameters.jl:38-41), and the CLIMAParameters constants
 * compute_turbulent_surface_fluxes (:553-620) and the two un-vendored packages it calls
 * (SurfaceFluxes 0.1, Thermodynamics 0.5) consume.  PARITY UNPINNED beyond the reference's
 * equilibrium invariant (test_prescribed_atmos_bc.jl:75-79): the Monin-Obukhov and saturation
 * formulas are restated from their published forms, not from sources under the reference tree
 * (SURVEY.md Appendix B). */
typedef struct {
    double u_atm, theta_atm, z_atm, theta_scale, rho_a_sfc, q_atm; /* the six fields, :119-132 */
    double z_0m, z_0s;                                             /* SoilParams */
    double R_v, R_d, grav, cp_d, cp_v, LH_v0, T_triple, press_triple; /* CLIMAParameters.Planet */
    double von_karman;                                             /* SubgridScale.von_karman_const */
} lho_atmos_forcing;

typedef struct {
    int32_t kind; /* LHO_BC_* */
    int32_t pad_;
    double value; /* flux (FLUX) or state value at this time (DIRICHLET) */
} lho_bc;

typedef struct {
    int32_t model; /* LHO_MODEL_* */
    int32_t nlev;
    double zmin, zmax;
    lho_earth_params earth;
    lho_soil_params soil;
    lho_vg_params vg;
    lho_cond_factors cf;
    lho_bc bc[2][2]; /* [face][component] */
    /* 0 = reference formula as written (boundary_conditions.jl:395-398, the
     * gravity term keeps the top-face sign at the bottom); 1 = physically
     * consistent sign. */
    int32_t consistent_bottom_sign;
    /* 1: the top face is a PrescribedAtmosForcing (bc[TOP][*] are then ignored); only the
     * coupled model has a method (boundary_conditions.jl:553-560) */
    int32_t atmos_on;
    lho_atmos_forcing atmos;
} lho_model;

/* Optional per-column overrides (NULL = use the scalar in lho_model). */
typedef struct {
    const double *vg_n, *vg_alpha, *vg_theta_r, *vg_Ksat, *nu, *S_s;
    const double *bc_value[2][2]; /* [face][component], per column */
    const double *atm_u, *atm_theta, *atm_q; /* per-column u_atm, theta_atm, q_atm */
} lho_percol;

/* ---- grid (domain.jl:58-69; coupled.jl:198) -------------------------- */
void lho_grid_f64(double zmin, double zmax, int n, double* zc, double* zf);
void lho_grid_f32(double zmin, double zmax, int n, float* zc, float* zf);

/* ---- scalar closures, Float64 ---------------------------------------- */
double lho_volumetric_liquid_fraction_f64(double vl, double nu_eff);
double lho_effective_saturation_f64(double porosity, double vl, double theta_r);
double lho_matric_potential_f64(const lho_vg_params*, double S);
double lho_inverse_matric_potential_f64(const lho_vg_params*, double psi);
double lho_pressure_head_f64(const lho_vg_params*, double vl, double nu_eff, double S_s);
double lho_hydraulic_conductivity_f64(const lho_vg_params*, double S, double visc, double imp);
double lho_viscosity_factor_f64(const lho_cond_factors*, double T);
double lho_impedance_factor_f64(const lho_cond_factors*, double f_i);
double lho_hydrostatic_profile_f64(const lho_vg_params*, double z, double z_interface,
                                   double nu, double S_s);
double lho_temperature_from_rhoe_int_f64(double rhoe, double ti, double rho_c_s,
                                         const lho_earth_params*);
double lho_volumetric_heat_capacity_f64(double tl, double ti, double rho_c_ds,
                                        const lho_earth_params*);
double lho_volumetric_internal_energy_f64(double ti, double rho_c_s, double T,
                                          const lho_earth_params*);
double lho_saturated_thermal_conductivity_f64(double tl, double ti, double k_unf, double k_fr);
double lho_relative_saturation_f64(double tl, double ti, double porosity);
double lho_kersten_number_f64(double ti, double S_r, const lho_soil_params*);
double lho_thermal_conductivity_f64(double k_dry, double K_e, double k_sat);
double lho_volumetric_internal_energy_liq_f64(double T, const lho_earth_params*);
double lho_k_solid_f64(double nu_om, double nu_q, double k_q, double k_min, double k_om);
double lho_ksat_frozen_f64(double k_solid, double porosity, double k_ice);
double lho_ksat_unfrozen_f64(double k_solid, double porosity, double k_l);
double lho_k_dry_f64(const lho_earth_params*, const lho_soil_params*);

/* ---- scalar closures, Float32 ---------------------------------------- */
float lho_volumetric_liquid_fraction_f32(float vl, float nu_eff);
float lho_effective_saturation_f32(float porosity, float vl, float theta_r);
float lho_matric_potential_f32(const lho_vg_params*, float S);
float lho_inverse_matric_potential_f32(const lho_vg_params*, float psi);
float lho_pressure_head_f32(const lho_vg_params*, float vl, float nu_eff, float S_s);
float lho_hydraulic_conductivity_f32(const lho_vg_params*, float S, float visc, float imp);
float lho_viscosity_factor_f32(const lho_cond_factors*, float T);
float lho_impedance_factor_f32(const lho_cond_factors*, float f_i);
float lho_hydrostatic_profile_f32(const lho_vg_params*, float z, float z_interface, float nu,
                                  float S_s);
float lho_temperature_from_rhoe_int_f32(float rhoe, float ti, float rho_c_s,
                                        const lho_earth_params*);
float lho_volumetric_heat_capacity_f32(float tl, float ti, float rho_c_ds,
                                       const lho_earth_params*);
float lho_volumetric_internal_energy_f32(float ti, float rho_c_s, float T,
                                         const lho_earth_params*);
float lho_saturated_thermal_conductivity_f32(float tl, float ti, float k_unf, float k_fr);
float lho_relative_saturation_f32(float tl, float ti, float porosity);
float lho_kersten_number_f32(float ti, float S_r, const lho_soil_params*);
float lho_thermal_conductivity_f32(float k_dry, float K_e, float k_sat);
float lho_volumetric_internal_energy_liq_f32(float T, const lho_earth_params*);
float lho_k_solid_f32(float nu_om, float nu_q, float k_q, float k_min, float k_om);
float lho_ksat_frozen_f32(float k_solid, float porosity, float k_ice);
float lho_ksat_unfrozen_f32(float k_solid, float porosity, float k_l);
float lho_k_dry_f32(const lho_earth_params*, const lho_soil_params*);

/* compute_turbulent_surface_fluxes(energy, hydrology, model, vartheta_l, theta_i, T)
 * (boundary_conditions.jl:553-620) for the top-cell state of one column: the heat flux and the
 * water volume flux through the surface (positive upward).  Returns 0, 1 when the
 * Monin-Obukhov system has no root (bulk Richardson number past the critical one: fluxes NaN). */
int lho_turbulent_surface_fluxes_f64(const lho_model*, double vl, double ti, double T,
                                     double* heat_flux, double* water_flux);
int lho_turbulent_surface_fluxes_f32(const lho_model*, float vl, float ti, float T,
                                     float* heat_flux, float* water_flux);

/* ---- batched tendency: rhs!(dY, Y, Ya, t) -----------------------------
 * Arrays are addressed as a[c*col_stride + i*lev_stride] (element strides).
 * Richards: state vl, ti ; aux T (may be NULL unless viscosity is ON)
 *           -> d_vl, d_ti            (right_hand_side.jl:118-186)
 * Heat    : state rhoe ; aux vl, ti  -> d_rhoe            (:192-263)
 * Coupled : state vl, ti, rhoe       -> d_vl, d_ti, d_rhoe (:269-369)
 * Unused pointers may be NULL.  Returns 0, or a negative code for an invalid
 * model / boundary-condition combination (the Julia code would raise).
 * nthreads <= 1 runs serially; > 1 uses OpenMP over columns when built with it.
 */
int lho_rhs_f64(const lho_model*, const lho_percol*, int64_t ncols, const double* vl,
                const double* ti, const double* rhoe, const double* T_aux, double* d_vl,
                double* d_ti, double* d_rhoe, int64_t lev_stride, int64_t col_stride,
                int nthreads);
int lho_rhs_f32(const lho_model*, const lho_percol*, int64_t ncols, const float* vl,
                const float* ti, const float* rhoe, const float* T_aux, float* d_vl,
                float* d_ti, float* d_rhoe, int64_t lev_stride, int64_t col_stride,
                int nthreads);

/* Diagnostic: centre fields K, psi, T, kappa (NULL = skip) of one batch. */
int lho_diagnostics_f64(const lho_model*, const lho_percol*, int64_t ncols, const double* vl,
                        const double* ti, const double* rhoe, const double* T_aux, double* K,
                        double* psi, double* T, double* kappa, int64_t lev_stride,
                        int64_t col_stride);
int lho_diagnostics_f32(const lho_model*, const lho_percol*, int64_t ncols, const float* vl,
                        const float* ti, const float* rhoe, const float* T_aux, float* K,
                        float* psi, float* T, float* kappa, int64_t lev_stride,
                        int64_t col_stride);

/* boundary_fluxes(X, bc, face, model, cs, t) (boundary_conditions.jl:470-489, :516-533) per column:
 * the SetValue flux pair (f_rhoe_int, f_vartheta_l) of one face, [ncols] each; NaN for a component
 * without a boundary condition. */
int lho_boundary_fluxes_f64(const lho_model*, const lho_percol*, int64_t ncols, const double* vl,
                            const double* ti, const double* rhoe, const double* T_aux,
                            int64_t lev_stride, int64_t col_stride, int face, double* f_e, double* f_w);
int lho_boundary_fluxes_f32(const lho_model*, const lho_percol*, int64_t ncols, const float* vl,
                            const float* ti, const float* rhoe, const float* T_aux,
                            int64_t lev_stride, int64_t col_stride, int face, float* f_e, float* f_w);

/* ---- SSPRK33, fixed dt (OrdinaryDiffEq SSPRK33 as driven by
 * src/Simulations/simulation.jl:58-70; Shu-Osher form, stage times
 * t, t+dt, t+dt/2).  State arrays are advanced in place.  bc_stage_values is
 * NULL (boundary values constant in time) or [nsteps][3][2][2] doubles
 * (step, stage, face, component) replacing lho_model.bc[][].value per stage.
 * For LHO_MODEL_HEAT vl/ti are the prescribed (aux) fields and stay fixed;
 * for LHO_MODEL_RICHARDS T_aux stays fixed.
 */
int lho_ssprk33_f64(const lho_model*, const lho_percol*, int64_t ncols, double* vl, double* ti,
                    double* rhoe, const double* T_aux, int64_t lev_stride, int64_t col_stride,
                    double t0, double dt, int64_t nsteps, const double* bc_stage_values,
                    int nthreads);
int lho_ssprk33_f32(const lho_model*, const lho_percol*, int64_t ncols, float* vl, float* ti,
                    float* rhoe, const float* T_aux, int64_t lev_stride, int64_t col_stride,
                    double t0, double dt, int64_t nsteps, const double* bc_stage_values,
                    int nthreads);

/* Build-defined stable-dt bound (no reference counterpart):
 * courant*dz^2 / max over faces of the face diffusivities (see the .inc).
 * Returns the min over the batch. */
double lho_stable_dt_f64(const lho_model*, const lho_percol*, int64_t ncols, const double* vl,
                         const double* ti, const double* rhoe, const double* T_aux,
                         int64_t lev_stride, int64_t col_stride, double courant);
double lho_stable_dt_f32(const lho_model*, const lho_percol*, int64_t ncols, const float* vl,
                         const float* ti, const float* rhoe, const float* T_aux,
                         int64_t lev_stride, int64_t col_stride, double courant);

int lho_openmp_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* LH_ORACLE_H */

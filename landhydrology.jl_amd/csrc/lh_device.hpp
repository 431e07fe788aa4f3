// lh_device.hpp -- kernel-argument structs shared by the host API (lh_api.hip)
// and the gfx950 kernels (lh_kernels.hip).  Product code: nothing here touches
// the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lh {

enum { MODEL_RICHARDS = 0, MODEL_HEAT = 1, MODEL_COUPLED = 2 };
enum { BC_NONE = 0, BC_FLUX = 1, BC_DIRICHLET = 2, BC_FREE_DRAINAGE = 3 };
enum { FACE_BOTTOM = 0, FACE_TOP = 1 };
enum { COMP_ENERGY = 0, COMP_HYDROLOGY = 1 };
enum { PC_VG_N = 0, PC_VG_ALPHA, PC_VG_THETA_R, PC_VG_KSAT, PC_NU, PC_S_S, PC_COUNT };

// per-column constants derived from vanGenuchten / SoilParams; one set per lane
// (uniform ensembles: computed once on the host, see DevParams::uc)
template <typename FT>
struct ColC {
    FT nu, S_s, theta_r, theta_lim; // theta_lim = theta_r + eps(FT)
    FT n, inv_n, m, inv_m;          // m = 1 - 1/n (SoilWaterParameterizations.jl:167)
    FT alpha_pnn;                   // alpha^(-n)
    FT Ksat;
    FT cgw;                         // cg2 * Ksat: the water flux constant of the column kernels (K is carried without Ksat)
    FT k_dry;                       // SoilHeatParameterizations.jl:280-294
    FT inv_por, inv_S_s, inv_nu;    // 1/(nu - theta_r), 1/S_s, 1/nu
    FT log2_alpha;                  // log2(alpha), for the log-domain psi
    FT l2_por;                      // log2(nu - theta_r) BY THE DEVICE'S OWN log2 (finish_colc): ice lanes
    // exponent multipliers of the log-domain closures, pre-scaled by the exp2 unit
    // of the math policy (MathFast<double>::EXP2_SCALE = 2048, else 1)
    FT e_inv_m, e_m, e_inv_n, e_log2_alpha, e_one; // S*/m, S*m, S*/n, S*log2 alpha, S
    // every power of the water closures stays a normal number for every S >= eps / (nu - theta_r)
    // (set_fast_vg): m >= LH_VG_FAST_MIN_M and moderate alpha, porosity.  Clay-like columns below that
    // (n < 1.075) take the v_ldexp form of 2^(.), which saturates like the reference's pows do.
    int32_t vg_fast;
    int32_t pad_;
};
// smallest m = 1 - 1/n for which S^(1/m) of the driest representable cell stays a normal number
#define LH_VG_FAST_MIN_M 0.07

// Everything a launch needs, already rounded to the working type FT the way the
// Julia constructors round (FT(x)); passed by value as the kernel argument.
template <typename FT>
struct DevParams {
    int64_t ncols;   // columns owned by this context
    int64_t stride;  // elements between consecutive levels of a plane (>= ncols, multiple of 64)
    int32_t nlev;
    int32_t model;
    FT dz;        // (zmax - zmin) / nlev            (domain.jl:64)
    FT inv_dz;    // 1 / dz
    FT half_inv_dz; // (1/2) / dz: the arithmetic-mean factor of InterpolateC2F folded into GradientC2F (exact)
    FT cg2;       // (1/2) / dz^2 = half_inv_dz * inv_dz: the factor of a centre difference that makes a flux a TENDENCY
                  // (host-computed: a product of two kernel arguments would live in VGPRs -- there is no scalar f64 multiply)
    FT half_dz;   // boundary centre-to-face distance (boundary_conditions.jl:196-208)
    const FT* zc; // device array [nlev], coordinates(cs)

    // vanGenuchten{FT} (SoilWaterParameterizations.jl:150-169)
    FT vg_n, vg_alpha, vg_theta_r, vg_Ksat;
    // SoilParams{FT} (parameters.jl:11-43)
    FT nu, S_s, rho_c_ds, kappa_solid, rho_p, kappa_sat_unfrozen, kappa_sat_frozen,
        kappa_dry_parameter, nu_ss_om, b;
    FT kersten_exp_unfrozen; // (1 + nu_om - a*nu_q - nu_g)/2  (SoilHeatParameterizations.jl:165)
    FT kersten_exp_frozen;   // 1 + nu_om                      (:171)
    FT one_minus_om;         // 1 - nu_om                      (:169)
    FT neg_b_log2e_sc;       // -b log2(e) x the exp2 unit of the production math: exp(-b S_r) as one 2^(.)
    // log2 of the saturated conductivities, for kappa_su^(tl/tw) kappa_sf^(ti/tw) as one 2^(.)
    FT l2_kappa_sat_unfrozen, l2_kappa_sat_frozen;
    // CLIMAParameters constants as SoilHeatParameterizations.jl forms them
    FT rho_i, rhocp_l, rhocp_i, T_ref, LH_f0, k_air;
    // conductivity factors (SoilWaterParameterizations.jl:46-126)
    int32_t viscosity_kind, impedance_kind;
    FT gamma, T_ref_visc, Omega;

    ColC<FT> uc; // column constants of the scalar parameters (host-computed)

    // per-column overrides, device arrays [ncols] or nullptr
    const FT* pc[PC_COUNT];
    // boundary conditions [face][component]
    int32_t bc_kind[2][2];
    FT bc_value[2][2];
    const FT* bc_pc[2][2];
    int32_t consistent_bottom_sign;

    uint32_t* status; // device word; bit 0 = non-finite tendency seen
    const double* math_tab; // device copy of the log2/exp2 tables (lh_fastmath.hpp)
    void* dt_out;           // MODE 4: FT-sized device word receiving the stable-step minimum
    int32_t cs_cpb;         // persistent column stepper: columns per workgroup (0 = launcher's choice)
    int32_t seg_len;        // > 0: level-segmented launch, levels per segment (small ensembles)
    int32_t xcd_remap;      // workgroup -> column-block map that gives each XCD one contiguous column range
    // LEVEL-UNIFORM prescribed fields of Ya (lh_upload_profile): FT[nlev] device arrays or nullptr, indexed
    // like Planes::v -- the column kernels stage them in LDS beside z and read no plane for them
    // (Ya.soil.T .= T_profile.(zc, t) is a function of z and t only: right_hand_side.jl:54-62)
    const FT* aux_prof[4];
    int32_t vg_fast_all;    // every column has ColC::vg_fast (host decision: uniform parameters + the ranges of the per-column arrays)
};

// PrescribedAtmosForcing{FT} (boundary_conditions.jl:119-132) plus every constant
// compute_turbulent_surface_fluxes (:553-620) reads, rounded to FT (lh_atmos.hpp)
template <typename FT>
struct AtmosParams {
    FT u_atm, theta_atm, z_atm, theta_scale, rho_a_sfc, q_atm; // PrescribedAtmosForcing{FT}
    FT z_0m, z_0s;                                             // SoilParams.z_0m, z_0s
    FT R_v, R_d, grav, cp_d, cp_v, T_triple, press_triple, von_karman;
    FT cp_l, T_0, rho_liq;   // from the earth parameters of the context
    double cp_v_d, LH_v0_d;  // FT(cp_v (T - T_ref) + LH_v0): formed in Float64, rounded once (:614-615)
    const FT* pc_u;          // per-column u_atm / theta_atm / q_atm or nullptr
    const FT* pc_theta;
    const FT* pc_q;
};

// one FieldVector on the device: up to four planes [nlev][stride]
template <typename FT>
struct Planes {
    FT* v[4];
};

} // namespace lh

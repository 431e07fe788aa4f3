"""Host-side setup helpers: the pointwise functions LandHydrology.jl EXPORTS from
SoilWaterParameterizations / SoilHeatParameterizations and its users call to build initial
conditions, boundary values and expected profiles (e.g. `hydrostatic_profile` in
test/SoilModel/richards_equation.jl:52, experiments/SoilModel/surface_fluxes.jl:107;
`volumetric_heat_capacity` / `volumetric_internal_energy` in every coupled test).

These are NOT the hot path: the tendency evaluates its closures on the GPU
(csrc/lh_closures.hpp).  They are plain numpy, element-wise over scalars or arrays, in the
working type of their arguments (FT(...) where the Julia source rounds), so that a user of the
host mirror can write the reference's own setups without touching test infrastructure.

Reference lines are under src/SoilModel/.
"""
from __future__ import annotations

import numpy as np

__all__ = ["volumetric_liquid_fraction", "effective_saturation", "matric_potential",
           "inverse_matric_potential", "pressure_head", "hydraulic_conductivity",
           "hydrostatic_profile", "viscosity_factor", "impedance_factor",
           "temperature_from_ρe_int", "temperature_from_rhoe_int", "volumetric_heat_capacity",
           "volumetric_internal_energy", "volumetric_internal_energy_liq",
           "saturated_thermal_conductivity", "relative_saturation", "kersten_number",
           "thermal_conductivity", "k_solid", "ksat_frozen", "ksat_unfrozen", "k_dry"]


def _ft(*xs):
    """Working type: Float32 only if every array/numpy argument is Float32."""
    dts = [np.asarray(x).dtype for x in xs if isinstance(x, (np.ndarray, np.generic))]
    if dts and all(d == np.float32 for d in dts):
        return np.float32
    return np.float64


def _hm(hm, FT):
    n = np.asarray(hm.n, dtype=FT)
    return n, FT(1) - FT(1) / n, np.asarray(hm.alpha, dtype=FT), np.asarray(hm.theta_r, dtype=FT), \
        np.asarray(hm.Ksat, dtype=FT)


# ------------------------------------------------ SoilWaterParameterizations.jl

def volumetric_liquid_fraction(vartheta_l, nu_eff):
    """:181-188"""
    FT = _ft(vartheta_l, nu_eff)
    vl, ne = np.asarray(vartheta_l, FT), np.asarray(nu_eff, FT)
    return np.where(vl < ne, vl, ne).astype(FT)[()]


def effective_saturation(porosity, vartheta_l, theta_r):
    """:213-217; may exceed 1"""
    FT = _ft(porosity, vartheta_l, theta_r)
    vl, thr, por = (np.asarray(a, FT) for a in (vartheta_l, theta_r, porosity))
    safe = np.maximum(vl, thr + np.finfo(FT).eps)
    return ((safe - thr) / (por - thr)).astype(FT)[()]


def matric_potential(hm, S):
    """:196-200"""
    FT = _ft(S)
    n, m, alpha, _, _ = _hm(hm, FT)
    S = np.asarray(S, FT)
    return (-((S ** (-FT(1) / m) - FT(1)) * alpha ** (-n)) ** (FT(1) / n)).astype(FT)[()]


def inverse_matric_potential(hm, psi):
    """:253-258; a positive potential is an error"""
    FT = _ft(psi)
    psi = np.asarray(psi, FT)
    if np.any(psi > 0):
        raise ValueError("Matric potential is positive")
    n, m, alpha, _, _ = _hm(hm, FT)
    return ((FT(1) + (alpha * np.abs(psi)) ** n) ** (-m)).astype(FT)[()]


def pressure_head(hm, vartheta_l, nu_eff, S_s):
    """:229-242"""
    FT = _ft(vartheta_l, nu_eff, S_s)
    vl, ne, ss = (np.asarray(a, FT) for a in (vartheta_l, nu_eff, S_s))
    _, _, _, thr, _ = _hm(hm, FT)
    S = np.asarray(effective_saturation(ne, vl, thr), FT)
    with np.errstate(invalid="ignore", divide="ignore"):
        unsat = np.asarray(matric_potential(hm, np.minimum(S, FT(1))), FT)
    return np.where(S <= 1, unsat, (vl - ne) / ss).astype(FT)[()]


def hydraulic_conductivity(hm, S, viscosity_f=1.0, impedance_f=1.0):
    """:269-282"""
    FT = _ft(S)
    _, m, _, _, Ksat = _hm(hm, FT)
    S = np.asarray(S, FT)
    with np.errstate(invalid="ignore"):
        Sc = np.minimum(S, FT(1))
        K = np.where(S < 1, np.sqrt(Sc) * (FT(1) - (FT(1) - Sc ** (FT(1) / m)) ** m) ** FT(2), FT(1))
    return (K * Ksat * np.asarray(viscosity_f, FT) * np.asarray(impedance_f, FT)).astype(FT)[()]


def hydrostatic_profile(hm, z, z_interface, nu, S_s):
    """:290-306 -- hydrostatic augmented liquid fraction with the saturated/unsaturated boundary
    at `z_interface` (z_∇)."""
    FT = _ft(z, z_interface, nu, S_s)
    n, m, alpha, thr, _ = _hm(hm, FT)
    z, zi, nu, ss = (np.asarray(a, FT) for a in (z, z_interface, nu, S_s))
    with np.errstate(invalid="ignore"):
        S = (FT(1) + (alpha * np.maximum(z - zi, FT(0))) ** n) ** (-m)
    return np.where(z > zi, S * (nu - thr) + thr, -ss * (z - zi) + nu).astype(FT)[()]


def viscosity_factor(vm, T=None):
    """:104-126: NoEffect -> 1; TemperatureDependentViscosity -> exp(γ (T - T_ref))"""
    if not hasattr(vm, "gamma"):
        return 1.0
    FT = _ft(T)
    return np.exp(FT(vm.gamma) * (np.asarray(T, FT) - FT(vm.T_ref))).astype(FT)[()]


def impedance_factor(imp, f_i=None):
    """:76-93: NoEffect -> 1; IceImpedance -> FT(10^(-Ω f_i))"""
    if not hasattr(imp, "Omega"):
        return 1.0
    FT = _ft(f_i)
    return np.asarray(10.0 ** (-float(imp.Omega) * np.asarray(f_i, np.float64)), FT)[()]


# ------------------------------------------------- SoilHeatParameterizations.jl

def _earth(ps, FT):
    rho_i, rho_l = FT(ps.rho_cloud_ice), FT(ps.rho_cloud_liq)
    return dict(rho_i=rho_i, rho_l=rho_l, T_ref=FT(ps.T_0), LH_f0=FT(ps.LH_f0),
                rhocp_l=FT(ps.cp_l * float(rho_l)), rhocp_i=FT(ps.cp_i * float(rho_i)))


def temperature_from_rhoe_int(rhoe_int, theta_i, rho_c_s, param_set):
    """:42-53"""
    FT = _ft(rhoe_int, theta_i, rho_c_s)
    e = _earth(param_set, FT)
    re, ti, rcs = (np.asarray(a, FT) for a in (rhoe_int, theta_i, rho_c_s))
    return (e["T_ref"] + (re + ti * e["rho_i"] * e["LH_f0"]) / rcs).astype(FT)[()]


temperature_from_ρe_int = temperature_from_rhoe_int


def volumetric_heat_capacity(theta_l, theta_i, rho_c_ds, param_set):
    """:65-79"""
    FT = _ft(theta_l, theta_i, rho_c_ds)
    e = _earth(param_set, FT)
    tl, ti, ds = (np.asarray(a, FT) for a in (theta_l, theta_i, rho_c_ds))
    return (ds + tl * e["rhocp_l"] + ti * e["rhocp_i"]).astype(FT)[()]


def volumetric_internal_energy(theta_i, rho_c_s, T, param_set):
    """:91-102"""
    FT = _ft(theta_i, rho_c_s, T)
    e = _earth(param_set, FT)
    ti, rcs, T = (np.asarray(a, FT) for a in (theta_i, rho_c_s, T))
    return (rcs * (T - e["T_ref"]) - ti * e["rho_i"] * e["LH_f0"]).astype(FT)[()]


def volumetric_internal_energy_liq(T, param_set):
    """:198-207"""
    FT = _ft(T)
    e = _earth(param_set, FT)
    return (e["rhocp_l"] * (np.asarray(T, FT) - e["T_ref"])).astype(FT)[()]


def saturated_thermal_conductivity(theta_l, theta_i, kappa_sat_unfrozen, kappa_sat_frozen):
    """:114-128"""
    FT = _ft(theta_l, theta_i, kappa_sat_unfrozen, kappa_sat_frozen)
    tl, ti = np.asarray(theta_l, FT), np.asarray(theta_i, FT)
    tw = tl + ti
    with np.errstate(invalid="ignore", divide="ignore"):
        k = FT(kappa_sat_unfrozen) ** (tl / tw) * FT(kappa_sat_frozen) ** (ti / tw)
    return np.where(tw < np.finfo(FT).eps, FT(0), k).astype(FT)[()]


def relative_saturation(theta_l, theta_i, porosity):
    """:139-142"""
    FT = _ft(theta_l, theta_i, porosity)
    return ((np.asarray(theta_l, FT) + np.asarray(theta_i, FT)) / np.asarray(porosity, FT)).astype(FT)[()]


def kersten_number(theta_i, S_r, soil_params):
    """:152-174"""
    FT = _ft(theta_i, S_r)
    sp = soil_params
    ti, Sr = np.asarray(theta_i, FT), np.asarray(S_r, FT)
    om, q, g, a, b = (FT(getattr(sp, k)) for k in ("nu_ss_om", "nu_ss_quartz", "nu_ss_gravel", "a", "b"))
    with np.errstate(invalid="ignore"):
        unfrozen = Sr ** ((FT(1) + om - a * q - g) / FT(2)) * \
            ((FT(1) + np.exp(-b * Sr)) ** FT(-3) - ((FT(1) - Sr) / FT(2)) ** FT(3)) ** (FT(1) - om)
        frozen = Sr ** (FT(1) + om)
    return np.where(ti < np.finfo(FT).eps, unfrozen, frozen).astype(FT)[()]


def thermal_conductivity(kappa_dry, K_e, kappa_sat):
    """:185-188"""
    FT = _ft(kappa_dry, K_e, kappa_sat)
    kd, Ke, ks = (np.asarray(a, FT) for a in (kappa_dry, K_e, kappa_sat))
    return (Ke * ks + (FT(1) - Ke) * kd).astype(FT)[()]


def k_solid(nu_ss_om, nu_ss_quartz, kappa_quartz, kappa_minerals, kappa_om):
    """:223-233"""
    FT = _ft(nu_ss_om, nu_ss_quartz, kappa_quartz, kappa_minerals, kappa_om)
    om, q = FT(nu_ss_om), FT(nu_ss_quartz)
    return FT(FT(kappa_om) ** om * FT(kappa_quartz) ** q * FT(kappa_minerals) ** (FT(1) - om - q))


def ksat_frozen(kappa_solid, porosity, kappa_ice):
    """:245-247"""
    FT = _ft(kappa_solid, porosity, kappa_ice)
    return FT(FT(kappa_solid) ** (FT(1) - FT(porosity)) * FT(kappa_ice) ** FT(porosity))


def ksat_unfrozen(kappa_solid, porosity, kappa_l):
    """:258-260"""
    FT = _ft(kappa_solid, porosity, kappa_l)
    return FT(FT(kappa_solid) ** (FT(1) - FT(porosity)) * FT(kappa_l) ** FT(porosity))


def k_dry(param_set, soil_params):
    """:280-294 with ρb_ss :268-270"""
    sp = soil_params
    FT = np.dtype(getattr(sp, "FT", np.float64)).type
    kdp, por, rho_p, ks = (FT(getattr(sp, k)) for k in ("kappa_dry_parameter", "nu", "rho_p", "kappa_solid"))
    k_air = FT(param_set.K_therm)
    rho_b = (FT(1) - por) * rho_p
    return FT(((kdp * ks - k_air) * rho_b + k_air * rho_p) / (rho_p - (FT(1) - kdp) * rho_b))

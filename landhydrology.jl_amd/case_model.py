"""Plain-number description of a soil model for the synthetic workloads.

Everything SoilModel(...) holds (src/SoilModel/models.jl:90-135 of the reference) as plain
Python numbers: what `workloads.make_case` builds its cases from and what every runner -- the HIP
path through the C ABI (`workloads.GpuModel`: bench.py, tools/, tests/) and, in the tests only, the
CPU oracle -- is configured from.  Part of the package so that bench.py depends on nothing under
tests/; imports nothing but the standard library.
"""
from __future__ import annotations

from dataclasses import dataclass, field

MODEL_RICHARDS, MODEL_HEAT, MODEL_COUPLED = 0, 1, 2
BC_NONE, BC_FLUX, BC_DIRICHLET, BC_FREE_DRAINAGE, BC_ATMOS = 0, 1, 2, 3, 4
FACE_BOTTOM, FACE_TOP = 0, 1
COMP_ENERGY, COMP_HYDROLOGY = 0, 1


@dataclass
class EarthParams:
    """CLIMAParameters 0.1 values read by SoilHeatParameterizations.jl:12-13 (K_therm
    and T_0 are pinned by the reference's tests, the others are unpinned and therefore inputs
    everywhere)."""
    rho_liq: float = 1000.0
    rho_ice: float = 916.7
    cp_l: float = 4181.0
    cp_i: float = 2100.0
    T_0: float = 273.16
    LH_f0: float = 2.8344e6 - 2.5008e6
    K_therm: float = 2.4e-2


@dataclass
class SoilParams:
    """src/SoilModel/parameters.jl:11-43 defaults (loam)."""
    nu: float = 0.43
    S_s: float = 1e-3
    nu_ss_gravel: float = 0.0
    nu_ss_om: float = 0.0
    nu_ss_quartz: float = 0.41
    rho_c_ds: float = 2700.0
    kappa_solid: float = 3.97
    rho_p: float = 2700.0
    kappa_sat_unfrozen: float = 1.72
    kappa_sat_frozen: float = 3.13
    a: float = 0.24
    b: float = 18.1
    kappa_dry_parameter: float = 0.053
    z_0m: float = 0.001     # roughness lengths, only read by the prescribed-atmosphere BC
    z_0s: float = 0.001


@dataclass
class VGParams:
    """SoilWaterParameterizations.jl:161-166 defaults (loam)."""
    n: float = 1.56
    alpha: float = 3.6
    theta_r: float = 0.0
    Ksat: float = 2.9e-7


@dataclass
class CondFactors:
    """SoilWaterParameterizations.jl:46-65"""
    viscosity_kind: int = 0
    impedance_kind: int = 0
    gamma: float = 2.64e-2
    T_ref: float = 288.0
    Omega: float = 7.0


@dataclass
class AtmosForcing:
    """PrescribedAtmosForcing{FT} (boundary_conditions.jl:119-132) plus the constants its
    flux computation reads from CLIMAParameters (Planet: R_v, grav, R_d, cp_d, cp_v, LH_v0,
    T_triple, press_triple; SubgridScale: von_karman_const), CLIMAParameters 0.1 values."""
    u_atm: float = 0.34
    theta_atm: float = 299.0
    z_atm: float = 0.05
    theta_scale: float = 299.0
    rho_a_sfc: float = 1.17
    q_atm: float = 0.015
    R_v: float = 8.3144598 / 18.01528e-3
    R_d: float = 8.3144598 / 28.97e-3
    grav: float = 9.81
    cp_d: float = (8.3144598 / 28.97e-3) / (2.0 / 7.0)
    cp_v: float = 1859.0
    LH_v0: float = 2.5008e6
    T_triple: float = 273.16
    press_triple: float = 611.657
    von_karman: float = 0.4


def default_earth(**kw) -> EarthParams:
    return EarthParams(**kw)


def default_soil(**kw) -> SoilParams:
    return SoilParams(**kw)


def default_vg(**kw) -> VGParams:
    return VGParams(**kw)


def default_cf(viscosity=False, impedance=False, gamma=2.64e-2, T_ref=288.0, Omega=7.0):
    return CondFactors(int(viscosity), int(impedance), gamma, T_ref, Omega)


@dataclass
class CaseModel:
    model: int
    nlev: int
    zmin: float
    zmax: float
    earth: EarthParams = field(default_factory=default_earth)
    soil: SoilParams = field(default_factory=default_soil)
    vg: VGParams = field(default_factory=default_vg)
    cf: CondFactors = field(default_factory=default_cf)
    # bc[(face, comp)] = (kind, value)
    bc: dict = field(default_factory=dict)
    consistent_bottom_sign: bool = False
    # per-column overrides: name -> float64 array [ncols]; bc values via percol_bc[(face, comp)]
    percol: dict = field(default_factory=dict)
    percol_bc: dict = field(default_factory=dict)
    # PrescribedAtmosForcing at the top face (then bc has no top entries), or None
    atmos: AtmosForcing = None
    # per-column overrides of u_atm / theta_atm / q_atm: name -> float64 array [ncols]
    percol_atmos: dict = field(default_factory=dict)

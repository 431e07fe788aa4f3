"""Pins the CPU oracle against every known-answer test the reference holds for
the hot path.  The reference is Julia and cannot
run here; these are its own closed forms, grids and conservation results.
Citations are file:line under the reference tree."""
import dataclasses
import math

import numpy as np
import pytest

import case_model as M
import oracle_py as O

F32, F64 = np.float32, np.float64
EARTH = M.default_earth()


def call(name, dtype, *args):
    return O.fn(name, dtype)(*[O.as_c(a) for a in args])


# ---------------------------------------------------------------- K1 (Float32)
# test/SoilModel/test_water_parameterizations.jl:1-63
class TestWaterParams:
    FT = F32
    vg = M.default_vg(theta_r=float(np.float32(0.2)))
    nu = np.float32(0.4)
    S_s = np.float32(1e-2)

    def hm(self):
        n = F32(self.vg.n)
        alpha = F32(self.vg.alpha)
        m = F32(1) - F32(1) / n
        return n, alpha, m, F32(self.vg.Ksat)

    def test_effective_saturation(self):  # :10-13
        th = np.array([0.3, 0.4, 0.5], dtype=F32)
        S = [call("lho_effective_saturation", F32, self.nu, t, F32(0.2)) for t in th]
        assert np.allclose(S, [0.5, 1.0, 1.5], rtol=math.sqrt(np.finfo(F32).eps))

    def test_matric_potential_and_inverse(self):  # :16-21
        n, alpha, m, _ = self.hm()
        S = np.array([0.5, 1.0], dtype=F32)
        S0 = F32(call("lho_effective_saturation", F32, self.nu, F32(0.3), F32(0.2)))
        va = -F32(F32(F32(S0 ** (-F32(1) / m)) - F32(1)) * F32(alpha ** (-n))) ** (F32(1) / n)
        psi = [call("lho_matric_potential", F32, self.vg, s) for s in S]
        assert psi[0] == pytest.approx(float(va), rel=1e-5)
        assert psi[1] == 0.0
        inv = [call("lho_inverse_matric_potential", F32, self.vg, F32(p)) for p in psi]
        assert np.allclose(inv, S, rtol=math.sqrt(np.finfo(F32).eps))
        # :19 inverse of a positive potential is an error (NaN in the oracle)
        assert math.isnan(call("lho_inverse_matric_potential", F32, self.vg, F32(1)))

    def test_pressure_head(self):  # :24-26
        th = np.array([0.3, 0.4, 0.5], dtype=F32)
        p = [call("lho_pressure_head", F32, self.vg, t, self.nu, self.S_s) for t in th]
        psi = [call("lho_matric_potential", F32, self.vg, F32(s)) for s in (0.5, 1.0)]
        assert np.allclose(p, psi + [10.0], rtol=math.sqrt(np.finfo(F32).eps), atol=0)

    def test_hydraulic_conductivity(self):  # :30-36
        n, alpha, m, Ksat = self.hm()
        S = np.array([0.5, 1.0, 1.5], dtype=F32)
        k = [call("lho_hydraulic_conductivity", F32, self.vg, s, F32(1), F32(1)) for s in S]
        S0 = F32(0.5)
        inner = F32(1) - F32(F32(1) - F32(S0 ** (F32(1) / m))) ** m
        va = F32(np.sqrt(S0)) * F32(inner) ** F32(2) * Ksat
        assert np.allclose(k, [va, Ksat, Ksat], rtol=1e-5)

    def test_factors(self):  # :40-46
        cf = M.default_cf(viscosity=True, impedance=True)
        assert call("lho_impedance_factor", F32, cf, F32(1.0)) == pytest.approx(1e-7, rel=1e-6)
        T = np.array([278.0, 288.0, 298.0], dtype=F32)
        got = [call("lho_viscosity_factor", F32, cf, t) for t in T]
        want = np.exp(F32(2.64e-2) * (T - F32(288.0)))
        assert np.allclose(got, want, rtol=1e-6)
        none = M.default_cf()
        assert call("lho_impedance_factor", F32, none, F32(0.3)) == 1.0
        assert call("lho_viscosity_factor", F32, none, F32(300)) == 1.0

    def test_hydrostatic_profile_gives_uniform_head(self):  # :49-54 (K9)
        z = np.arange(-1.0, 0.0 + 1e-9, 0.1).astype(F32)
        th = [call("lho_hydrostatic_profile", F32, self.vg, zz, F32(-0.5), self.nu, self.S_s)
              for zz in z]
        psi = [call("lho_pressure_head", F32, self.vg, F32(t), self.nu, self.S_s) for t in th]
        h = np.array(psi, dtype=F32) + z
        assert np.std(h, ddof=1) < 1e-6

    def test_volumetric_liquid_fraction(self):  # :57-58
        got = [call("lho_volumetric_liquid_fraction", F32, F32(v), F32(0.5))
               for v in (0.25, 0.5, 0.75)]
        assert got == [0.25, 0.5, 0.5]


# ---------------------------------------------------------------- K2 (Float64)
# test/SoilModel/test_heat_parameterizations.jl:22-80 -- the reference asserts
# `==`; here +,-,*,/ forms are compared bitwise and pow forms to 2 ulp (libm vs
# numpy pow).
class TestHeatParams:
    rho_l, rho_i = EARTH.rho_liq, EARTH.rho_ice
    rhocp_l = EARTH.cp_l * EARTH.rho_liq
    rhocp_i = EARTH.cp_i * EARTH.rho_ice
    T_ref, LH_f0, k_air = EARTH.T_0, EARTH.LH_f0, EARTH.K_therm

    def test_temperature(self):  # :22-23
        got = call("lho_temperature_from_rhoe_int", F64, 5.4e7, 0.05, 2.1415e6, EARTH)
        assert got == self.T_ref + (5.4e7 + 0.05 * self.rho_i * self.LH_f0) / 2.1415e6

    def test_heat_capacity(self):  # :25-26
        got = call("lho_volumetric_heat_capacity", F64, 0.25, 0.05, 1e6, EARTH)
        assert got == 1e6 + 0.25 * self.rhocp_l + 0.05 * self.rhocp_i

    def test_internal_energy(self):  # :28-29
        got = call("lho_volumetric_internal_energy", F64, 0.05, 2.1415e6, 300.0, EARTH)
        assert got == 2.1415e6 * (300.0 - self.T_ref) - 0.05 * self.rho_i * self.LH_f0

    def test_ksat(self):  # :31-34
        got = call("lho_saturated_thermal_conductivity", F64, 0.25, 0.05, 0.57, 2.29)
        want = 0.57 ** (0.25 / (0.05 + 0.25)) * 2.29 ** (0.05 / (0.05 + 0.25))
        assert got == pytest.approx(want, rel=4e-16)
        assert call("lho_saturated_thermal_conductivity", F64, 0.0, 0.0, 0.57, 2.29) == 0.0

    def test_relative_saturation(self):  # :36
        assert call("lho_relative_saturation", F64, 0.25, 0.05, 0.4) == (0.25 + 0.05) / 0.4

    SP = M.default_soil(nu=0.2, S_s=1e-3, nu_ss_om=0.1, nu_ss_gravel=0.1, nu_ss_quartz=0.1,
                        rho_c_ds=0.0, kappa_solid=0.1, rho_p=1.0, kappa_sat_unfrozen=0.0,
                        kappa_sat_frozen=0.0)

    def test_kersten(self):  # :39-62
        got = call("lho_kersten_number", F64, 0.0, 0.75, self.SP)
        want = (0.75 ** ((1.0 + 0.1 - 0.24 * 0.1 - 0.1) / 2.0) *
                ((1.0 + math.exp(-18.1 * 0.75)) ** (-3.0) - ((1.0 - 0.75) / 2.0) ** 3.0)
                ** (1.0 - 0.1))
        assert got == pytest.approx(want, rel=4e-16)
        got = call("lho_kersten_number", F64, 0.05, 0.75, self.SP)
        assert got == pytest.approx(0.75 ** (1.0 + 0.1), rel=4e-16)

    def test_thermal_conductivity(self):  # :64-65
        got = call("lho_thermal_conductivity", F64, 1.5, 0.7287, 0.7187)
        assert got == 0.7287 * 0.7187 + (1.0 - 0.7287) * 1.5

    def test_internal_energy_liq(self):  # :67-68
        got = call("lho_volumetric_internal_energy_liq", F64, 300.0, EARTH)
        assert got == self.rhocp_l * (300.0 - self.T_ref)

    def test_k_solid_ksat(self):  # :70-77
        assert call("lho_k_solid", F64, 0.5, 0.25, 2.0, 3.0, 2.0) == pytest.approx(
            2.0 ** 0.5 * 2.0 ** 0.25 * 3.0 ** 0.25, rel=4e-16)
        assert call("lho_ksat_frozen", F64, 0.5, 0.1, 0.4) == pytest.approx(
            0.5 ** 0.9 * 0.4 ** 0.1, rel=4e-16)
        assert call("lho_ksat_unfrozen", F64, 0.5, 0.1, 0.4) == pytest.approx(
            0.5 ** 0.9 * 0.4 ** 0.1, rel=4e-16)

    def test_k_dry(self):  # :79-81
        got = call("lho_k_dry", F64, EARTH, self.SP)
        k = self.k_air
        want = ((0.053 * 0.1 - k) * 0.8 + k * 1.0) / (1.0 - (1.0 - 0.053) * 0.8)
        assert got == want

    def test_K_therm_is_pinned_by_heat_test(self):
        # heat_test_interface.jl:7 sets rho_c_ds = 0.43314518988433487 so that
        # kappa/rho_c_s == 1 for dry soil; that equals k_dry only for
        # K_therm = 0.024.
        sp = M.default_soil(nu=0.495, nu_ss_gravel=0.1, nu_ss_om=0.1, nu_ss_quartz=0.1,
                            rho_c_ds=0.43314518988433487, kappa_solid=8.0,
                            kappa_sat_unfrozen=0.57, kappa_sat_frozen=2.29)
        assert call("lho_k_dry", F64, EARTH, sp) == 0.43314518988433487


# ---------------------------------------------------------------- K3 grid
def test_grid_matches_coupled_jl_198():
    zc, zf = O.grid(-2.0, 0.0, 20)
    want = np.array([(-195 + 10 * i) / 100 for i in range(20)])   # -1.95:0.1:-0.05
    assert np.allclose(zc, want, rtol=0, atol=4.5e-16)
    assert zf[0] == -2.0 and zf[-1] == 0.0
    zc32, _ = O.grid(-2.0, 0.0, 20, dtype=F32)
    assert zc32.dtype == F32 and np.allclose(zc32, zc, atol=1e-7)


# ---------------------------------------------------------------- shared setup
def coupled_soil():
    """Parameters of test/SoilModel/coupled.jl:3-32 (both testsets)."""
    nu = 0.5
    k_solid = call("lho_k_solid", F64, 0.0, 0.92, 7.7, 2.5, 0.25)
    k_fr = call("lho_ksat_frozen", F64, k_solid, nu, 2.29)
    k_unf = call("lho_ksat_unfrozen", F64, k_solid, nu, 0.57)
    sp = M.default_soil(nu=nu, S_s=1e-3, nu_ss_gravel=0.0, nu_ss_om=0.0, nu_ss_quartz=0.92,
                        rho_c_ds=(1 - nu) * 1.926e06, kappa_solid=k_solid,
                        kappa_sat_unfrozen=k_unf, kappa_sat_frozen=k_fr)
    vg = M.default_vg(n=2.0, alpha=2.6, Ksat=0.0443 / 3600 / 100, theta_r=0.0)
    return sp, vg


def zero_flux_bcs(energy=True, hydrology=True):
    bc = {}
    for f in (M.FACE_BOTTOM, M.FACE_TOP):
        if energy:
            bc[(f, M.COMP_ENERGY)] = (M.BC_FLUX, 0.0)
        if hydrology:
            bc[(f, M.COMP_HYDROLOGY)] = (M.BC_FLUX, 0.0)
    return bc


# ---------------------------------------------------------------- K4
def test_single_rhs_eval_default_ic():
    """test/SoilModel/coupled.jl:123-234 ("test default ic")."""
    sp, vg = coupled_soil()
    n = 20
    om = M.CaseModel(M.MODEL_COUPLED, n, -2.0, 0.0, soil=sp, vg=vg, bc=zero_flux_bcs())
    # default_initial_conditions, models.jl:147-162
    vl = np.full((1, n), 0.5 * sp.nu)
    ti = np.zeros((1, n))
    T0 = 273.16
    rho_c_s = call("lho_volumetric_heat_capacity", F64, 0.25, 0.0, sp.rho_c_ds, EARTH)
    rhoe = np.full((1, n), call("lho_volumetric_internal_energy", F64, 0.0, rho_c_s, T0, EARTH))
    assert np.allclose(vl, 0.25) and T0 == EARTH.T_0          # :199-201
    d = O.rhs(om, vl, ti, rhoe)
    assert np.allclose(d["ti"], 0.0, atol=0)                   # :221
    assert np.allclose(d["rhoe"], 0.0, atol=1e-12)             # :222 (approx zeros)
    S = call("lho_effective_saturation", F64, sp.nu, 0.25, 0.0)
    K = call("lho_hydraulic_conductivity", F64, vg, S, 1.0, 1.0)
    flux = np.zeros(n + 1) - K
    flux[0] = flux[-1] = 0.0
    minus_div = -(flux[1:] - flux[:-1]) / 0.1
    assert np.sum(d["vl"][0] - minus_div) < np.finfo(F64).eps  # :234 as written
    assert np.allclose(d["vl"][0], minus_div, rtol=1e-12, atol=1e-20)  # and elementwise


# ---------------------------------------------------------------- K5
@pytest.mark.slow
def test_heat_analytic():
    """test/SoilModel/heat_test_interface.jl:1-100: heat-only, Dirichlet T=0 top,
    T = 5 cos(2 pi t) bottom, n=60 on (0,1), dt=1e-4, t_f=2, MSE < 1e-6."""
    sp = M.default_soil(nu=0.495, nu_ss_gravel=0.1, nu_ss_om=0.1, nu_ss_quartz=0.1,
                        rho_c_ds=0.43314518988433487, kappa_solid=8.0, kappa_sat_unfrozen=0.57,
                        kappa_sat_frozen=2.29)
    n, dt, tf = 60, 1e-4, 2.0
    A, omega = 5.0, 2 * math.pi / 1.0
    bc = {(M.FACE_TOP, M.COMP_ENERGY): (M.BC_DIRICHLET, 0.0),
          (M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_DIRICHLET, A)}
    om = M.CaseModel(M.MODEL_HEAT, n, 0.0, 1.0, soil=sp, bc=bc)
    vl = np.zeros((1, n))   # PrescribedHydrologyModel defaults, models.jl:73-78
    ti = np.zeros((1, n))
    rho_c_s = call("lho_volumetric_heat_capacity", F64, 0.0, 0.0, sp.rho_c_ds, EARTH)
    rhoe = np.full((1, n), call("lho_volumetric_internal_energy", F64, 0.0, rho_c_s, 0.0, EARTH))
    nsteps = int(round(tf / dt))
    # stage times of SSPRK33: t, t+dt, t+dt/2
    t = dt * np.arange(nsteps)
    ts = np.stack([t, t + dt, t + dt / 2], axis=1)
    bcv = np.zeros((nsteps, 3, 2, 2))
    bcv[:, :, M.FACE_BOTTOM, M.COMP_ENERGY] = A * np.cos(omega * ts)
    O.ssprk33(om, dt, nsteps, vl=vl, ti=ti, rhoe=rhoe, bc_stage_values=bcv)
    z, _ = O.grid(0.0, 1.0, n)
    s = math.sqrt(omega / 2) * (1 + 1j)
    num = np.exp(s * (1 - z)) - np.exp(-s * (1 - z))
    den = np.exp(s) - np.exp(-s)
    analytic = np.real(num * A * np.exp(1j * omega * tf) / den)
    Tfinal = np.array([call("lho_temperature_from_rhoe_int", F64, r, 0.0, rho_c_s, EARTH)
                       for r in rhoe[0]])
    mse = np.mean((analytic - Tfinal) ** 2)
    assert mse < 1e-6                                           # :99


def expected_equilibrium(z, z_interface, nu, S_s=1e-3, alpha=2.6, n=2.0, m=0.5):
    return np.where(z < z_interface, -S_s * (z - z_interface) + nu,
                    nu * (1 + (alpha * np.maximum(z - z_interface, 0.0)) ** n) ** (-m))


# ---------------------------------------------------------------- K6
@pytest.mark.slow
def test_coupled_variably_saturated_equilibrium():
    """test/SoilModel/coupled.jl:1-120: n=20 on (-2,0), zero-flux BCs, 32 days at
    dt=20 s; vl -> hydrostatic profile with interface -0.3, T -> 284 K."""
    sp, vg = coupled_soil()
    n, dt = 20, 20.0
    nsteps = int(60 * 60 * 24 * 32 / dt)
    om = M.CaseModel(M.MODEL_COUPLED, n, -2.0, 0.0, soil=sp, vg=vg, bc=zero_flux_bcs())
    z, _ = O.grid(-2.0, 0.0, n)
    vl = np.full((1, n), 0.495)
    ti = np.zeros((1, n))
    T = 289.0 + 5.0 * z
    rho_c_s = call("lho_volumetric_heat_capacity", F64, 0.495, 0.0, sp.rho_c_ds, EARTH)
    rhoe = np.array([[call("lho_volumetric_internal_energy", F64, 0.0, rho_c_s, t, EARTH)
                      for t in T]])
    mass0, e0 = vl.sum(), rhoe.sum()
    O.ssprk33(om, dt, nsteps, vl=vl, ti=ti, rhoe=rhoe)
    # the reference's assertions exactly as written (:117-118)
    assert math.sqrt(np.mean(vl[0] - expected_equilibrium(z, -0.3, 0.5)) ** 2.0) < 1e-3
    rcs = np.array([call("lho_volumetric_heat_capacity", F64, v, 0.0, sp.rho_c_ds, EARTH)
                    for v in vl[0]])
    temp = np.array([call("lho_temperature_from_rhoe_int", F64, r, 0.0, c, EARTH)
                     for r, c in zip(rhoe[0], rcs)])
    assert math.sqrt(np.mean(temp - 284.0) ** 2.0) < 1e-3
    # zero-flux BCs + one flux per face => discrete conservation
    assert abs(vl.sum() - mass0) < 1e-11 * abs(mass0)
    assert abs(rhoe.sum() - e0) < 1e-11 * abs(e0)
    # stronger than the reference's signed-mean check: per-cell RMS against the
    # continuum profile (coarse 0.1 m grid => a few 1e-4), and a uniform total head
    assert np.sqrt(np.mean((vl[0] - expected_equilibrium(z, -0.2936, 0.5)) ** 2)) < 5e-4
    psi = np.array([call("lho_pressure_head", F64, vg, v, sp.nu, sp.S_s) for v in vl[0]])
    assert np.std(psi + z) < 1e-3
    assert np.all(ti == 0.0)


# ---------------------------------------------------------------- K7
@pytest.mark.slow
def test_richards_variably_saturated_equilibrium():
    """test/SoilModel/richards_equation.jl:1-95: n=50 on (-10,0), 36 days at
    dt=100 s, interface -0.56."""
    sp = M.default_soil(nu=0.495, S_s=1e-3)
    vg = M.default_vg(n=2.0, alpha=2.6, Ksat=0.0443 / 3600 / 100, theta_r=0.0)
    n, dt = 50, 100.0
    nsteps = int(60 * 60 * 24 * 36 / dt)
    om = M.CaseModel(M.MODEL_RICHARDS, n, -10.0, 0.0, soil=sp, vg=vg,
                       bc=zero_flux_bcs(energy=False))
    z, _ = O.grid(-10.0, 0.0, n)
    vl = np.full((1, n), 0.494)
    ti = np.zeros((1, n))
    mass0 = vl.sum()
    O.ssprk33(om, dt, nsteps, vl=vl, ti=ti)
    assert math.sqrt(np.mean(vl[0] - expected_equilibrium(z, -0.56, 0.495)) ** 2.0) < 1e-4  # :94
    assert abs(vl.sum() - mass0) < 1e-11 * mass0
    # stronger than the signed mean: the discrete equilibrium has a uniform head
    psi = np.array([call("lho_pressure_head", F64, vg, v, sp.nu, sp.S_s) for v in vl[0]])
    assert np.std(psi + z) < 5e-3
    assert np.sqrt(np.mean((vl[0] - expected_equilibrium(z, -0.5606, 0.495)) ** 2)) < 2e-3


# ---------------------------------------------------------------- K8 (setup only)
def test_sand_infiltration_setup_runs():
    """test/SoilModel/richards_equation.jl:98-173: Dirichlet top + FreeDrainage
    bottom.  Its comparison data is downloaded in the reference (:175-185) and is
    not available offline, so only qualitative properties are checked: the front
    moves down, vl stays within [IC, top value], flux leaves through the bottom."""
    sp = M.default_soil(nu=0.287, S_s=1e-3)
    vg = M.default_vg(n=3.96, alpha=2.7, Ksat=34 / 3600 / 100, theta_r=0.075)
    n, dt = 150, 0.25
    bc = {(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.267),
          (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FREE_DRAINAGE, 0.0)}
    om = M.CaseModel(M.MODEL_RICHARDS, n, -1.5, 0.0, soil=sp, vg=vg, bc=bc)
    vl = np.full((1, n), 0.1)
    ti = np.zeros((1, n))
    O.ssprk33(om, dt, 2400, vl=vl, ti=ti)   # 10 minutes of the 48
    assert np.all(np.isfinite(vl))
    assert vl[0, -1] > 0.25 and vl[0, 0] < 0.1001
    assert np.all(np.diff(vl[0]) >= -1e-9)        # monotone wetting front
    assert vl.max() <= 0.267 + 1e-9 and vl.min() >= 0.1 - 1e-6


# ---------------------------------------------------------------- K9 invariants
def test_zero_interior_flux_for_hydrostatic_uniform_T():
    sp, vg = coupled_soil()
    n = 40
    om = M.CaseModel(M.MODEL_COUPLED, n, -2.0, 0.0, soil=sp, vg=vg, bc=zero_flux_bcs())
    z, _ = O.grid(-2.0, 0.0, n)
    vl = np.array([[call("lho_hydrostatic_profile", F64, vg, zz, -0.7, sp.nu, sp.S_s)
                    for zz in z]])
    ti = np.zeros((1, n))
    rcs = [call("lho_volumetric_heat_capacity", F64, min(v, sp.nu), 0.0, sp.rho_c_ds, EARTH)
           for v in vl[0]]
    rhoe = np.array([[call("lho_volumetric_internal_energy", F64, 0.0, c, 285.0, EARTH)
                      for c in rcs]])
    d = O.rhs(om, vl, ti, rhoe)
    Ksat = vg.Ksat
    assert np.max(np.abs(d["vl"])) < 1e-9 * Ksat / 0.05 * 1e3
    assert np.all(d["ti"] == 0.0)


def test_invalid_combinations_are_errors():
    sp, vg = coupled_soil()
    n = 8
    vl = np.full((1, n), 0.3)
    ti = np.zeros((1, n))
    # a dynamic component with NoBC has no flux to SetValue -> error
    om = M.CaseModel(M.MODEL_RICHARDS, n, -1.0, 0.0, soil=sp, vg=vg, bc={})
    with pytest.raises(ValueError):
        O.rhs(om, vl, ti)
    # zlim[1] < zlim[2] assertion, domain.jl:30
    om = M.CaseModel(M.MODEL_RICHARDS, n, 0.0, -1.0, soil=sp, vg=vg,
                       bc=zero_flux_bcs(energy=False))
    with pytest.raises(ValueError):
        O.rhs(om, vl, ti)


def test_bottom_dirichlet_sign_quirk_and_flag():
    """boundary_conditions.jl:395-398: at the bottom the reference flips the whole
    expression, gravity term included.  Default = as written; flag = consistent."""
    sp = M.default_soil()
    vg = M.default_vg()
    n = 64
    bc = {(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.35),
          (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.20)}
    vl = np.full((1, n), 0.20)
    ti = np.zeros((1, n))
    om = M.CaseModel(M.MODEL_RICHARDS, n, -1.28, 0.0, soil=sp, vg=vg, bc=bc)
    d_ref = O.rhs(om, vl, ti)["vl"][0]
    om.consistent_bottom_sign = True
    d_fix = O.rhs(om, vl, ti)["vl"][0]
    S = call("lho_effective_saturation", F64, sp.nu, 0.20, 0.0)
    K = call("lho_hydraulic_conductivity", F64, vg, S, 1.0, 1.0)
    dz = 1.28 / 64
    # psi_f == psi_c at the bottom: as written F_bot = +K, consistent F_bot = -K;
    # the interior face above carries -K.
    assert d_ref[0] == pytest.approx(-((-K) - (+K)) / dz, rel=1e-9)
    assert d_fix[0] == pytest.approx(0.0, abs=1e-9 * K / dz)
    assert np.allclose(d_ref[1:], d_fix[1:], rtol=0, atol=0)


def test_input_generation_helpers_match_the_oracle():
    """tests/parity_cases.py generates inputs (for the parity tests and for bench.py)
    without executing the oracle: its numpy grid and its literal soil constants must be
    bitwise what the oracle's functions give."""
    import parity_cases as pc
    for (zmin, zmax, n) in [(-1.28, 0.0, 64), (-2.56, 0.0, 128), (-2.0, 0.0, 20), (0.0, 1.0, 60),
                            (-3.0, -0.5, 37), (-10.0, 0.0, 50), (-0.1, 0.0, 1), (-2.4, 0.0, 48)]:
        for dt in (np.float64, np.float32):
            zc, zf = pc.grid_np(zmin, zmax, n, dt)
            wc, wf = O.grid(zmin, zmax, n, dt)
            np.testing.assert_array_equal(zc, wc)
            np.testing.assert_array_equal(zf, wf)
    f = lambda name, *a: O.fn(name, np.float64)(*a)
    k_solid = f("lho_k_solid", 0.0, 0.92, 7.7, 2.5, 0.25)
    assert k_solid == pc.COUPLED_K_SOLID
    assert f("lho_ksat_unfrozen", k_solid, 0.5, 0.57) == pc.COUPLED_KSAT_UNFROZEN
    assert f("lho_ksat_frozen", k_solid, 0.5, 2.29) == pc.COUPLED_KSAT_FROZEN


# ------------------------------------------------ K10: prescribed-atmosphere BC (8f-4)
def _atmos_reference_model(q_atm=None, T_surf=299.0):
    """test/SoilModel/test_prescribed_atmos_bc.jl:9-57"""
    atm = M.AtmosForcing(u_atm=0.34, theta_atm=T_surf, z_atm=0.05, theta_scale=T_surf, rho_a_sfc=1.17,
                         q_atm=0.0 if q_atm is None else q_atm)
    if q_atm is None:       # q_vap_saturation_generic(param_set, T_surf, rho_a_sfc, Liquid()) (:28)
        e = M.default_earth()
        dcp = atm.cp_v - e.cp_l
        p = atm.press_triple * (T_surf / atm.T_triple) ** (dcp / atm.R_v) * math.exp(
            (atm.LH_v0 - dcp * e.T_0) / atm.R_v * (1 / atm.T_triple - 1 / T_surf))
        atm.q_atm = p / (atm.rho_a_sfc * atm.R_v * T_surf)
    return M.CaseModel(M.MODEL_COUPLED, 10, -0.55, 0.0, soil=M.default_soil(nu=0.55, rho_c_ds=1.0),
                       vg=M.default_vg(n=1.68, alpha=5.0, Ksat=0.0, theta_r=0.084),
                       bc={(M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_FLUX, 0.0),
                           (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FLUX, 0.0)}, atmos=atm)


def test_k10_prescribed_atmosphere_equilibrium_invariant():
    """test_prescribed_atmos_bc.jl:59-79: saturated soil and saturated air at the same
    temperature exchange nothing: `sum(parent(dY)) == 0.0`.  The ONLY reference-held answer for
    this boundary condition (SurfaceFluxes.jl / Thermodynamics.jl are not in the reference tree:
    everything else about it is parity-unpinned)."""
    om = _atmos_reference_model()
    n = 10
    e, sp = om.earth, om.soil
    rho_c_s = call("lho_volumetric_heat_capacity", F64, sp.nu, 0.0, sp.rho_c_ds, e)
    rhoe = call("lho_volumetric_internal_energy", F64, 0.0, rho_c_s, 299.0, e)
    vl = np.full((1, n), sp.nu)
    d = O.rhs(om, vl, np.zeros((1, n)), np.full((1, n), rhoe))
    assert sum(float(np.sum(v)) for v in d.values()) == 0.0
    assert not any(np.any(v) for v in d.values())


def test_k10_surface_flux_properties():
    """test_prescribed_atmos_bc.jl:81-159 as far as it needs no un-vendored package: oversaturated
    == saturated (:155), neutral => tstar == 0 => no sensible heat (:148-151), plus the physics
    any correct evaluation must show (signs, monotonicity in the wind, no root past the critical
    Richardson number)."""
    om = _atmos_reference_model()
    nu = 0.55
    h, w, st = O.turbulent_surface_fluxes(om, [nu, nu + 1e-3, nu - 1e-3, nu], [0, 0, 0, 0.1],
                                          [299.0, 299.0, 289.5, 289.5])
    assert not st.any()
    assert h[0] == 0.0 and w[0] == 0.0 and (h[0], w[0]) == (h[1], w[1])
    assert h[2] < 0 and h[3] < 0           # warm air, cold soil: heat flows down (negative z-flux)
    dry = dataclasses.replace(om, atmos=dataclasses.replace(om.atmos, q_atm=0.005))
    h2, w2, _ = O.turbulent_surface_fluxes(dry, [0.4], [0.0], [299.0])
    assert w2[0] > 0 and h2[0] > 0         # dry air over moist soil at equal temperature: evaporation
    winds = [O.turbulent_surface_fluxes(dataclasses.replace(dry, atmos=dataclasses.replace(dry.atmos, u_atm=u)),
                                        [0.4], [0.0], [305.0])[1][0] for u in (0.2, 1.0, 5.0)]
    assert winds[0] < winds[1] < winds[2]  # more wind, more evaporation
    assert O.turbulent_surface_fluxes(dry, [0.4], [0.0], [250.0])[2][0] == 1       # no root
    # Float32 runs the same code (the reference itself cannot: min(S, 1.0) promotes to Float64)
    h32, w32, _ = O.turbulent_surface_fluxes(dry, [0.4], [0.0], [305.0], dtype=F32)
    h64, w64, _ = O.turbulent_surface_fluxes(dry, [0.4], [0.0], [305.0])
    assert h32[0] == pytest.approx(h64[0], rel=1e-4) and w32[0] == pytest.approx(w64[0], rel=1e-4)


def test_k10_error_paths():
    """boundary_conditions.jl:553-560: compute_turbulent_surface_fluxes has a method for
    SoilEnergyModel + SoilHydrologyModel only (test_prescribed_atmos_bc.jl:161-183)."""
    om = _atmos_reference_model()
    n = 10
    vl, ti = np.full((1, n), 0.3), np.zeros((1, n))
    with pytest.raises(ValueError):
        O.rhs(dataclasses.replace(om, model=M.MODEL_RICHARDS), vl, ti)
    with pytest.raises(ValueError):
        O.rhs(dataclasses.replace(om, model=M.MODEL_HEAT), vl, ti, np.full((1, n), 1e7))

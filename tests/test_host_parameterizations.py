"""The host mirror's exported setup helpers (landhydrology.jl_amd/parameterizations.py) against
the CPU oracle's restatement of the same reference functions, and the reference's own
known answers for them (test/SoilModel/test_water_parameterizations.jl,
test_heat_parameterizations.jl).  CPU only: these helpers are numpy (setup, not the hot path)."""
import math

import numpy as np
import pytest

import __graft_entry__ as g
import case_model as M
import oracle_py as O

pkg = g.load_package()
P = pkg.parameterizations


def _hm(FT, **kw):
    d = dict(n=1.43, alpha=2.6, Ksat=1e-6, theta_r=0.067)
    d.update(kw)
    return pkg.vanGenuchten(FT, **d), M.default_vg(**d)


@pytest.mark.parametrize("FT", [np.float32, np.float64])
def test_water_helpers_match_the_oracle(FT):
    hm, vg = _hm(FT)
    tol = 16 * np.finfo(FT).eps
    call = lambda name, *a: O.fn(name, FT)(*[O.as_c(x) for x in a])
    S = np.linspace(0.02, 0.999, 57).astype(FT)
    want = np.array([call("lho_matric_potential", vg, FT(s)) for s in S])
    assert np.allclose(P.matric_potential(hm, S), want, rtol=tol * 8, atol=0)
    psi = -np.logspace(-3, 2, 40).astype(FT)
    want = np.array([call("lho_inverse_matric_potential", vg, FT(p)) for p in psi])
    assert np.allclose(P.inverse_matric_potential(hm, psi), want, rtol=tol, atol=0)
    with pytest.raises(ValueError):                      # test_water_parameterizations.jl:19
        P.inverse_matric_potential(hm, FT(1.0))
    want = np.array([call("lho_hydraulic_conductivity", vg, FT(s), FT(1), FT(1)) for s in S])
    got = P.hydraulic_conductivity(hm, S)
    # K_r = sqrt(S) (1 - (1 - S^(1/m))^m)^2 cancels for small S: two correct evaluations differ by
    # eps * 2 / (1 - (1 - S^(1/m))^m) relatively (the tolerance model of tests/parity_cases.py)
    m = 1.0 - 1.0 / 1.43
    inner = 1.0 - (1.0 - S.astype(np.float64) ** (1.0 / m)) ** m
    assert np.all(np.abs(got - want) <= tol * (4.0 + 2.0 / inner) * np.abs(want)) and got.dtype == FT
    assert P.hydraulic_conductivity(hm, FT(1.5)) == FT(hm.Ksat)           # :30-36: K_r = 1 above saturation
    nu, S_s = FT(0.47), FT(1e-3)
    vl = np.linspace(0.07, 0.55, 33).astype(FT)
    want = np.array([call("lho_pressure_head", vg, FT(v), nu, S_s) for v in vl])
    assert np.allclose(P.pressure_head(hm, vl, nu, S_s), want, rtol=tol * 8, atol=1e-30)
    z = np.linspace(-2.0, 0.0, 41).astype(FT)
    want = np.array([call("lho_hydrostatic_profile", vg, FT(zz), FT(-0.8), nu, S_s) for zz in z])
    got = P.hydrostatic_profile(hm, z, FT(-0.8), nu, S_s)
    assert np.allclose(got, want, rtol=tol, atol=0) and got.dtype == FT
    # effective_saturation may exceed 1 and clamps at theta_r + eps (:10-13)
    assert np.allclose(P.effective_saturation(FT(0.4), np.array([0.3, 0.4, 0.5], FT), FT(0.2)),
                       [0.5, 1.0, 1.5], rtol=math.sqrt(np.finfo(FT).eps))
    assert P.effective_saturation(FT(0.4), FT(0.1), FT(0.2)) > 0
    assert P.volumetric_liquid_fraction(FT(0.5), FT(0.4)) == FT(0.4)
    assert P.volumetric_liquid_fraction(FT(0.3), FT(0.4)) == FT(0.3)
    # conductivity factors (:39-58 of the reference's test)
    visc = pkg.TemperatureDependentViscosity(FT)
    assert P.viscosity_factor(pkg.NoEffect(FT)) == 1.0
    assert np.isclose(P.viscosity_factor(visc, FT(288.0)), 1.0)
    cf = M.default_cf(viscosity=True, impedance=True)
    assert np.isclose(P.viscosity_factor(visc, FT(300.0)), call("lho_viscosity_factor", cf, FT(300.0)), rtol=tol)
    assert np.isclose(P.impedance_factor(pkg.IceImpedance(FT), FT(0.3)), call("lho_impedance_factor", cf, FT(0.3)), rtol=tol)
    assert P.impedance_factor(pkg.NoEffect(FT)) == 1.0


@pytest.mark.parametrize("FT", [np.float32, np.float64])
def test_heat_helpers_match_the_oracle(FT):
    tol = 16 * np.finfo(FT).eps
    ps = pkg.EarthParameterSet()
    earth = M.default_earth()
    call = lambda name, *a: O.fn(name, FT)(*[O.as_c(x) for x in a])
    tl, ti, ds = FT(0.2), FT(0.05), FT(2.1e6)
    rcs = P.volumetric_heat_capacity(tl, ti, ds, ps)
    assert np.isclose(rcs, call("lho_volumetric_heat_capacity", tl, ti, ds, earth), rtol=tol)
    T = FT(281.5)
    re = P.volumetric_internal_energy(ti, rcs, T, ps)
    assert np.isclose(re, call("lho_volumetric_internal_energy", ti, FT(rcs), T, earth), rtol=tol)
    assert np.isclose(P.temperature_from_rhoe_int(re, ti, rcs, ps), T, rtol=tol)     # round trip
    assert np.isclose(P.volumetric_internal_energy_liq(T, ps),
                      call("lho_volumetric_internal_energy_liq", T, earth), rtol=tol)
    assert np.isclose(P.saturated_thermal_conductivity(tl, ti, FT(1.7), FT(3.1)),
                      call("lho_saturated_thermal_conductivity", tl, ti, FT(1.7), FT(3.1)), rtol=tol)
    assert P.saturated_thermal_conductivity(FT(0), FT(0), FT(1.7), FT(3.1)) == 0
    assert np.isclose(P.relative_saturation(tl, ti, FT(0.5)), call("lho_relative_saturation", tl, ti, FT(0.5)), rtol=tol)
    sp = pkg.SoilParams(FT, ν=0.5, ν_ss_om=0.1, ν_ss_quartz=0.3, ν_ss_gravel=0.05)
    osp = M.default_soil(nu=0.5, nu_ss_om=0.1, nu_ss_quartz=0.3, nu_ss_gravel=0.05)
    for tii in (FT(0.0), FT(0.05)):
        assert np.isclose(P.kersten_number(tii, FT(0.6), sp), call("lho_kersten_number", tii, FT(0.6), osp), rtol=tol * 4)
    assert np.isclose(P.thermal_conductivity(FT(0.3), FT(0.4), FT(1.9)),
                      call("lho_thermal_conductivity", FT(0.3), FT(0.4), FT(1.9)), rtol=tol)
    ks = P.k_solid(FT(0.1), FT(0.3), FT(7.7), FT(2.5), FT(0.25))
    assert np.isclose(ks, call("lho_k_solid", FT(0.1), FT(0.3), FT(7.7), FT(2.5), FT(0.25)), rtol=tol)
    assert np.isclose(P.ksat_frozen(ks, FT(0.5), FT(2.29)), call("lho_ksat_frozen", FT(ks), FT(0.5), FT(2.29)), rtol=tol)
    assert np.isclose(P.ksat_unfrozen(ks, FT(0.5), FT(0.57)), call("lho_ksat_unfrozen", FT(ks), FT(0.5), FT(0.57)), rtol=tol)
    assert np.isclose(P.k_dry(ps, sp), call("lho_k_dry", earth, osp), rtol=tol)


def test_reference_known_answers():
    """The literals the reference pins: heat_test_interface.jl:7 (rho_c_ds equals k_dry for
    kappa_air = 0.024) and coupled.jl:16-22 (k_solid, ksat_*)."""
    FT = np.float64
    import parity_cases as pc
    ks = P.k_solid(FT(0.0), FT(0.92), FT(7.7), FT(2.5), FT(0.25))
    assert ks == pytest.approx(pc.COUPLED_K_SOLID, rel=1e-15)
    assert P.ksat_unfrozen(ks, FT(0.5), FT(0.57)) == pytest.approx(pc.COUPLED_KSAT_UNFROZEN, rel=1e-15)
    assert P.ksat_frozen(ks, FT(0.5), FT(2.29)) == pytest.approx(pc.COUPLED_KSAT_FROZEN, rel=1e-15)
    sp = pkg.SoilParams(FT, ν=0.495, ν_ss_gravel=0.1, ν_ss_om=0.1, ν_ss_quartz=0.1, κ_solid=8.0)
    assert P.k_dry(pkg.EarthParameterSet(), sp) == pytest.approx(0.43314518988433487, rel=1e-14)

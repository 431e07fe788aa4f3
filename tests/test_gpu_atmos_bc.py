"""the prescribed-atmosphere top boundary condition (PrescribedAtmosForcing,
src/SoilModel/boundary_conditions.jl:119-132, 516-533, 553-620) on the device.

What pins it:
  * the reference's own invariant -- saturated soil, soil and air at the same temperature,
    q_atm = q_sat  =>  sum(dY) == 0 exactly (test/SoilModel/test_prescribed_atmos_bc.jl:1-79);
    "oversaturated gives the same fluxes as saturated" (:155) and the error paths (:161-194);
  * everything else is the device against the CPU oracle's restatement of the same published
    formulas (SurfaceFluxes.jl 0.1 / Thermodynamics.jl 0.5 are not under the reference tree):
    PARITY UNPINNED beyond the invariant, and said so in DESIGN.md / SURVEY.md Appendix B.
"""
import dataclasses

import numpy as np
import pytest

import case_model as M
import parity_cases as pc

pytestmark = pytest.mark.gpu
O = pc.O


@pytest.fixture(scope="module")
def lh():
    import __graft_entry__ as g
    return g.load_package()


def _q_sat(atm, earth, T, rho):
    dcp = atm.cp_v - earth.cp_l
    p = atm.press_triple * (T / atm.T_triple) ** (dcp / atm.R_v) * np.exp(
        (atm.LH_v0 - dcp * earth.T_0) / atm.R_v * (1 / atm.T_triple - 1 / T))
    return p / (rho * atm.R_v * T)


def _reference_test_model(lh, FT=np.float64, q_atm=None):
    """test_prescribed_atmos_bc.jl:9-57"""
    ps = lh.EarthParameterSet()
    nu = 0.55
    msp = lh.SoilParams(FT, ν=nu, ρc_ds=1.0)
    domain = lh.Column(FT, zlim=(-0.55, 0.0), nelements=10)
    hm = lh.vanGenuchten(FT, n=1.68, α=5.0, Ksat=0.0, θr=0.084)
    T_surf, rho_a = 299.0, 1.17
    if q_atm is None:
        q_atm = _q_sat(M.AtmosForcing(), M.default_earth(), T_surf, rho_a)     # q_vap_saturation_generic
    surface_bc = lh.PrescribedAtmosForcing(FT, u_atm=0.34, θ_atm=T_surf, z_atm=0.05, θ_scale=T_surf,
                                           ρ_a_sfc=rho_a, q_atm=q_atm)
    bc = lh.SoilColumnBC(top=surface_bc, bottom=lh.SoilComponentBC(energy=lh.VerticalFlux(0.0),
                                                                   hydrology=lh.VerticalFlux(0.0)))
    model = lh.SoilModel(FT, domain=domain, energy_model=lh.SoilEnergyModel(),
                         hydrology_model=lh.SoilHydrologyModel(FT, hydraulic_model=hm),
                         boundary_conditions=bc, soil_param_set=msp, earth_param_set=ps)
    return model, nu, T_surf, q_atm


def _oracle_model(q_atm, n=10, **atm_kw):
    atm = M.AtmosForcing(u_atm=0.34, theta_atm=299.0, z_atm=0.05, theta_scale=299.0, rho_a_sfc=1.17,
                         q_atm=q_atm, **atm_kw)
    return M.CaseModel(M.MODEL_COUPLED, n, -0.55, 0.0, soil=M.default_soil(nu=0.55, rho_c_ds=1.0),
                       vg=M.default_vg(n=1.68, alpha=5.0, Ksat=0.0, theta_r=0.084),
                       bc={(M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_FLUX, 0.0),
                           (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FLUX, 0.0)}, atmos=atm)


def test_reference_equilibrium_invariant(lh):
    """test_prescribed_atmos_bc.jl:59-79: saturated soil at the air's temperature under saturated
    air: sum(parent(dY)) == 0.0 -- exactly."""
    model, nu, T_surf, _ = _reference_test_model(lh)

    def ic(z, m):
        rho_c_s = lh.volumetric_heat_capacity(m.soil_param_set.nu, 0.0, m.soil_param_set.rho_c_ds, m.earth_param_set)
        return {"ϑ_l": m.soil_param_set.nu + 0.0 * z, "θ_i": 0.0 * z,
                "ρe_int": lh.volumetric_internal_energy(0.0, rho_c_s, 299.0, m.earth_param_set) + 0.0 * z}

    Y, Ya = lh.initialize_states(model, ic, 0.0)
    dY = Y.similar()
    lh.make_rhs(model)(dY, Y, Ya, 0.0)
    total = sum(float(np.sum(dY.get(n))) for n in dY.names)
    assert total == 0.0
    for n in dY.names:
        assert not np.any(dY.get(n))
    model.close()


def test_reference_flux_checks_and_error_paths(lh):
    """test_prescribed_atmos_bc.jl:81-194 -- the four surface states of the reference's test."""
    model, nu, T_surf, q_atm = _reference_test_model(lh)
    vl = np.array([nu, nu + 1e-3, nu - 1e-3, nu])
    ti = np.array([0.0, 0.0, 0.0, 0.1])
    T = np.array([T_surf, T_surf, 289.5, 289.5])
    h, w = lh.compute_turbulent_surface_fluxes(model.energy_model, model.hydrology_model, model, vl, ti, T)
    assert h[0] == 0.0 and w[0] == 0.0                       # neutral and saturated: no flux at all
    assert (h[0], w[0]) == (h[1], w[1])                      # :155 oversaturated == saturated
    assert h.dtype == np.float64 and w.dtype == np.float64   # :158-159 typing
    assert h[2] < 0 and h[3] < 0                             # warm air over cold soil heats the soil (flux is +z)
    # the oracle's restatement agrees on all four
    oh, ow, st = O.turbulent_surface_fluxes(_oracle_model(q_atm), vl, ti, T)
    assert not st.any()
    assert np.allclose(h, oh, rtol=1e-11, atol=0) and np.allclose(w, ow, rtol=1e-11, atol=0)
    one = lh.compute_turbulent_surface_fluxes(model.energy_model, model.hydrology_model, model, vl[2], ti[2], T[2])
    assert isinstance(one[0], np.float64) and one[0] == h[2] and one[1] == w[2]
    # :161-183 no method for other component models
    for em, hm in ((lh.PrescribedTemperatureModel(), lh.PrescribedHydrologyModel()),
                   (lh.SoilEnergyModel(), lh.PrescribedHydrologyModel()),
                   (lh.PrescribedTemperatureModel(), lh.SoilHydrologyModel(np.float64))):
        with pytest.raises(Exception):
            lh.compute_turbulent_surface_fluxes(em, hm, model, vl[0], ti[0], T[0])
    # :186-193 only valid at the top
    with pytest.raises(Exception, match="only valid at the top"):
        lh.boundary_fluxes((vl[0], ti[0], T[0]), model.boundary_conditions.top, "bottom", model)
    got = lh.boundary_fluxes((vl[2], ti[2], T[2]), model.boundary_conditions.top, "top", model)
    assert got["fρe_int"] == h[2] and got["fϑ_l"] == w[2]
    model.close()
    # the C ABI refuses a prescribed atmosphere on a model without both components
    F = lh._ffi
    m2 = lh.SoilModel(np.float64, domain=lh.Column(np.float64, zlim=(-1.0, 0.0), nelements=8),
                      energy_model=lh.PrescribedTemperatureModel(),
                      hydrology_model=lh.SoilHydrologyModel(np.float64),
                      boundary_conditions=lh.SoilColumnBC(top=model.boundary_conditions.top,
                                                          bottom=lh.SoilComponentBC(hydrology=lh.VerticalFlux(0.0))),
                      earth_param_set=lh.EarthParameterSet())
    with pytest.raises(F.ModelError):
        m2._backend()
    m2.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_surface_fluxes_match_the_oracle_over_a_state_grid(dtype):
    """Unstable, neutral and (sub-critically) stable surfaces, dry to saturated soil, with and
    without ice: the device's solve of the Monin-Obukhov system against the oracle's."""
    om = _oracle_model(0.012, n=10)
    rng = np.random.default_rng(7)
    n = 4000
    vl = rng.uniform(0.09, 0.60, n)
    ti = np.where(rng.uniform(size=n) < 0.3, rng.uniform(0.0, 0.1, n), 0.0)
    T = rng.uniform(294.0, 320.0, n)
    T[:50] = 299.0                                           # exactly neutral
    case = pc.Case("atm_grid", om, dtype, 1, vl=np.zeros((1, 10), dtype), ti=np.zeros((1, 10), dtype),
                   rhoe=np.zeros((1, 10), dtype))
    import ctypes as C
    with pc.GpuModel(case) as g:
        h, w = np.empty(n), np.empty(n)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        g.F.check(g.L.lh_atmos_surface_fluxes(g.ctx, n, dp(vl), dp(ti), dp(T), dp(h), dp(w)), g.ctx)
        assert g.status() == 0
    oh, ow, st = O.turbulent_surface_fluxes(om, vl.astype(dtype), ti.astype(dtype), T.astype(dtype), dtype=dtype)
    assert not st.any()
    rtol = 1e-10 if dtype == np.float64 else 2e-3
    scale_h, scale_w = np.max(np.abs(oh)), np.max(np.abs(ow))
    assert np.all(np.abs(h - oh) <= rtol * (np.abs(oh) + 1e-3 * scale_h))
    assert np.all(np.abs(w - ow) <= rtol * (np.abs(ow) + 1e-3 * scale_w))
    # neutral columns still evaporate; stable and unstable surfaces are both in the sample
    assert np.all(w[:50] != 0) and (T < 299.0).any() and (T > 299.0).any()


def test_no_root_is_flagged_not_silent():
    """Strongly stable (cold soil under warm air, weak wind): past the critical bulk Richardson
    number the similarity system has no solution; the fluxes are NaN and status bit 1 is set."""
    om = _oracle_model(0.012)
    case = pc.Case("atm_noroot", om, np.float64, 1, vl=np.zeros((1, 10)), ti=np.zeros((1, 10)), rhoe=np.zeros((1, 10)))
    import ctypes as C
    vl, ti, T = np.array([0.3]), np.array([0.0]), np.array([270.0])
    assert O.turbulent_surface_fluxes(om, vl, ti, T)[2][0] == 1
    with pc.GpuModel(case) as g:
        h, w = np.empty(1), np.empty(1)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        g.F.check(g.L.lh_atmos_surface_fluxes(g.ctx, 1, dp(vl), dp(ti), dp(T), dp(h), dp(w)), g.ctx)
        assert np.isnan(h[0]) and np.isnan(w[0])
        assert g.status() & 2


def _atmos_case(dtype, N=300, n=24, percol=False):
    """An ensemble under a prescribed atmosphere: surface_fluxes.jl's soil (experiments/SoilModel/
    surface_fluxes.jl:25-62), varied initial moisture and temperature per column."""
    nu = 0.55
    sp = dataclasses.replace(pc.coupled_soil()[0], nu=nu, nu_ss_quartz=0.4, rho_c_ds=(1 - nu) * 1.926e06)
    vg = M.default_vg(n=1.68, alpha=5.0, Ksat=1.31 / 100 / 3600 / 1000, theta_r=0.084)
    atm = M.AtmosForcing(u_atm=0.34, theta_atm=299.0, z_atm=0.05, theta_scale=299.0, rho_a_sfc=1.17, q_atm=0.015)
    c = np.arange(N)
    pa = {}
    if percol:
        pa = {"u_atm": 0.2 + 3.0 * pc.uhash(c, 31, n), "theta_atm": 297.0 + 4.0 * pc.uhash(c, 32, n),
              "q_atm": 0.006 + 0.012 * pc.uhash(c, 33, n)}
    om = M.CaseModel(M.MODEL_COUPLED, n, -0.55, 0.0, soil=sp, vg=vg,
                     bc={(M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_FLUX, 0.0),
                         (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FLUX, 0.0)}, atmos=atm, percol_atmos=pa)
    zc, _ = pc.grid_np(-0.55, 0.0, n)
    vl = 0.25 + 0.25 * pc.uhash(c, 34, n)[:, None] + 0.05 * np.sin(9.0 * zc)[None, :]
    T = 296.0 + 6.0 * pc.uhash(c, 35, n)[:, None] + 2.0 * zc[None, :]
    e = om.earth
    rho_c_s = sp.rho_c_ds + vl * (e.cp_l * e.rho_liq)
    rhoe = rho_c_s * (T - e.T_0)
    return pc.Case("atmos_ens", om, dtype, N, vl=vl.astype(dtype), ti=np.zeros((N, n), dtype), rhoe=rhoe.astype(dtype))


def _assert_close_with_atmos_top(case, got, want, Cw):
    """The tolerance model of the interior and the bottom face (parity_cases) plus, in the top
    cell, the surface fluxes' own agreement bound (the device and the oracle converge to the same
    root: 1e-10 relative in Float64, 2e-3 in Float32 where the iteration stops at eps32)."""
    bc = dict(case.om.bc)
    bc[(M.FACE_TOP, M.COMP_ENERGY)] = bc[(M.FACE_TOP, M.COMP_HYDROLOGY)] = (M.BC_FLUX, 0.0)
    om_flux = dataclasses.replace(case.om, atmos=None, percol_atmos={}, bc=bc)
    tol = pc.tendency_tolerance(dataclasses.replace(case, om=om_flux), Cw)
    n = case.om.nlev
    dz = (case.om.zmax - case.om.zmin) / n
    rel = 1e-10 if case.dtype == np.float64 else 2e-3
    eps = float(np.finfo(case.dtype).eps) * Cw
    for k, top_scale in (("vl", 5e-8), ("rhoe", 700.0)):
        g, wv = got[k].astype(np.float64), want[k].astype(np.float64)
        allowed = tol[k] + eps * np.abs(wv)
        allowed[:, -1] += rel * (np.abs(wv[:, -1]) + top_scale / dz)
        assert np.all(np.isfinite(g))
        bad = np.abs(g - wv) > allowed
        assert not bad.any(), (k, int(bad.sum()), float(np.max(np.abs(g - wv) / allowed)))
    assert not np.any(got["ti"])


@pytest.mark.parametrize("dtype,percol", [(np.float64, False), (np.float64, True), (np.float32, False)])
def test_tendency_with_prescribed_atmosphere_matches_the_oracle(dtype, percol):
    case = _atmos_case(dtype, percol=percol)
    got = pc.run_gpu_rhs(case)
    want = pc.run_oracle_rhs(case)
    _assert_close_with_atmos_top(case, got, want, 4.0 if dtype == np.float64 else 16.0)
    # the surface really exchanges heat and water with the air (the test is not vacuous)
    assert np.max(np.abs(want["vl"][:, -1])) > 0 and np.max(np.abs(want["rhoe"][:, -1])) > 0


def test_stepping_under_a_prescribed_atmosphere_matches_the_oracle():
    """theta(z,t) and rho e_int(z,t) after 40 SSPRK33 steps: the surface fluxes are re-evaluated
    from the stage state before every stage launch, as the reference's rhs! does."""
    import test_gpu_stepper as ts
    case = _atmos_case(np.float64, N=64, percol=True)
    dt = 20.0
    got = ts.gpu_steps(case, dt, 40)
    want = ts.cpu_steps(case, dt, 40)
    for k in ("vl", "rhoe"):
        scale = np.max(np.abs(want[k]))
        assert np.max(np.abs(got[k] - want[k])) <= 1e-9 * scale, k
        assert np.max(np.abs(want[k] - getattr(case, "vl" if k == "vl" else "rhoe"))) > 1e-6 * scale

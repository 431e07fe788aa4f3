"""Level-uniform variables (lh_upload_profile): what the reference's make_update_aux writes into Ya --
`Ya.soil.T .= T_profile.(zc, t)`, `theta_l_profile`, `theta_i_profile` (right_hand_side.jl:54-81) -- is a
function of z and t only, i.e. ONE value per level for every column.  Uploaded as nlev numbers, read
by the column kernels from LDS, never a plane: everything here is a BITWISE comparison against
uploading the broadcast plane, for the tendency, the fused stages and the persistent stepper, plus
the time-dependent case (the profile refreshed at every stage time, as the reference's rhs! does)."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

import case_model as M
import parity_cases as pc

pytestmark = pytest.mark.gpu
O = pc.O


def _profile_case(kind, dtype, n=None):
    """A case whose prescribed aux fields are level-uniform; returns (case, {var: profile})."""
    if kind == "richards_viscosity":
        base = pc.make_case("richards_viscosity_f64" if n is None else f"richards_viscosity_f64")
        om = base.om
        zc, _ = pc.grid_np(om.zmin, om.zmax, om.nlev)
        T = (281.0 + 9.0 * np.sin(0.7 * zc)).astype(dtype)
        case = dataclasses.replace(base, dtype=dtype, vl=base.vl.astype(dtype), ti=base.ti.astype(dtype),
                                   T_aux=np.repeat(T[None, :], base.ncols, axis=0))
        return case, {"T": T}
    base = pc.make_case("heat_dirichlet_f64" if dtype == np.float64 else "heat_dirichlet_f32")
    om = base.om
    zc, _ = pc.grid_np(om.zmin, om.zmax, om.nlev)
    vl = (0.12 + 0.3 * (0.5 + 0.5 * np.sin(5.0 * zc))).astype(dtype)
    ti = np.where(zc > 0.6, 0.04, 0.0).astype(dtype) if kind == "heat_ice" else np.zeros(om.nlev, dtype)
    case = dataclasses.replace(base, vl=np.repeat(vl[None, :], base.ncols, axis=0),
                               ti=np.repeat(ti[None, :], base.ncols, axis=0))
    return case, {"vl": vl, "ti": ti}


def _states(g, case, profiles, as_profile):
    """Y, Ya with the aux fields uploaded as planes or as per-level profiles."""
    F = g.F
    Y, Ya = g.prognostic_and_aux()
    if as_profile:
        var = {"T": F.LH_VAR_T, "vl": F.LH_VAR_VARTHETA_L, "ti": F.LH_VAR_THETA_I}
        for k, v in profiles.items():
            v = np.ascontiguousarray(v, dtype=case.dtype)
            F.check(g.L.lh_upload_profile(g.ctx, Ya, var[k], v.ctypes.data), g.ctx)
    return Y, Ya


@pytest.mark.parametrize("kind,dtype", [("richards_viscosity", np.float64), ("heat", np.float64), ("heat", np.float32),
                                         ("heat_ice", np.float64), ("heat_ice", np.float32)])
def test_profile_is_bitwise_the_broadcast_plane(kind, dtype):
    case, prof = _profile_case(kind, dtype)
    res = []
    for as_profile in (False, True):
        out = {}
        with pc.GpuModel(case) as g:
            F = g.F
            Y, Ya = _states(g, case, prof, as_profile)
            dY = g.state(0)
            g.rhs(Y, Ya, dY)
            out.update({"d" + k: v for k, v in g.tendencies(dY).items()})
            import torch
            tdt = torch.zeros(1, device="cuda", dtype=torch.float64 if dtype == np.float64 else torch.float32)
            F.check(g.L.lh_rhs_stable_dt(g.ctx, 0.0, Y, Ya, dY, 0.5, tdt.data_ptr()), g.ctx)
            F.check(g.L.lh_synchronize(g.ctx), g.ctx)
            out["dt"] = np.array([tdt.item()])
            dt = 0.2 * float(tdt.item())
            for tune in (b"persist=0", b"persist=2"):
                F.check(g.L.lh_set_tuning(g.ctx, tune), g.ctx)
                Y2 = g.state(0)
                F.check(g.L.lh_state_copy(g.ctx, Y2, Y), g.ctx)
                F.check(g.L.lh_step_ssprk33(g.ctx, Y2, Ya, 0.0, dt, 5, None), g.ctx)
                var = F.LH_VAR_RHOE_INT if case.om.model == M.MODEL_HEAT else F.LH_VAR_VARTHETA_L
                out[tune.decode()] = g.download(Y2, var)
            # the profile is still what a download of the aux plane shows
            if as_profile:
                for k, v in prof.items():
                    vid = {"T": F.LH_VAR_T, "vl": F.LH_VAR_VARTHETA_L, "ti": F.LH_VAR_THETA_I}[k]
                    assert np.array_equal(g.download(Ya, vid), np.repeat(np.asarray(v, dtype)[None, :], case.ncols, axis=0))
            assert g.status() == 0
        res.append(out)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k], equal_nan=True), (kind, k)
    assert np.array_equal(res[0]["persist=0"], res[0]["persist=2"])
    pc.assert_tendencies_close(case, {k[1:]: v for k, v in res[1].items() if k.startswith("d") and k != "dt"},
                               pc.run_oracle_rhs(case))


def test_a_prognostic_profile_becomes_a_plane_on_the_device():
    """An initial condition f(z): nlev numbers uploaded, the plane broadcast on the device."""
    case = pc.make_case("c2_richards_f64", ncols=333)
    prof = np.ascontiguousarray(case.vl[7])
    uni = dataclasses.replace(case, vl=np.repeat(prof[None, :], case.ncols, axis=0))
    with pc.GpuModel(uni) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.rhs(Y, Ya, dY)
        a = g.tendencies(dY)
        F.check(g.L.lh_state_fill(g.ctx, Y, F.LH_VAR_VARTHETA_L, 0.123), g.ctx)
        F.check(g.L.lh_upload_profile(g.ctx, Y, F.LH_VAR_VARTHETA_L, prof.ctypes.data), g.ctx)
        g.rhs(Y, Ya, dY)
        b = g.tendencies(dY)
        assert np.array_equal(g.download(Y, F.LH_VAR_VARTHETA_L), uni.vl)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
        # an all-zero profile is a zero fill (the no-ice kernels then do not even read the plane)
        z = np.zeros(case.om.nlev)
        F.check(g.L.lh_upload_profile(g.ctx, Y, F.LH_VAR_THETA_I, z.ctypes.data), g.ctx)
        g.rhs(Y, Ya, dY)
        c = g.tendencies(dY)
        for k in a:
            assert np.array_equal(a[k], c[k]), k


def test_time_dependent_profile_is_refreshed_at_every_stage_time():
    """The reference's rhs! evaluates T_profile at the stage time of EVERY evaluation
    (right_hand_side.jl:37-42).  A profile that holds for five steps and then jumps must be seen by
    run() even though the first steps show no time dependence (the host mirror probes every stage
    time of a chunk); the result equals stepping stage by stage with the plane uploaded by hand."""
    lh = pc._pkg()
    FT = np.float64
    n, N = 40, 96
    dt = 600.0

    def T_profile(z, t):
        return (280.0 if t < 5 * dt else 296.0) + 2.0 * z

    def build(Tp):
        hm = lh.vanGenuchten(FT, n=2.0, α=2.6, Ksat=0.0443 / 3600 / 100, θr=0.0)
        dom = lh.Column(FT, zlim=(-2.0, 0.0), nelements=n, ncolumns=N)
        bc = lh.SoilColumnBC(top=lh.SoilComponentBC(hydrology=lh.VerticalFlux(-2e-8)),
                             bottom=lh.SoilComponentBC(hydrology=lh.FreeDrainage()))
        model = lh.SoilModel(FT, domain=dom, energy_model=lh.PrescribedTemperatureModel(T_profile=Tp),
                             hydrology_model=lh.SoilHydrologyModel(
                                 FT, hydraulic_model=hm, viscosity_factor=lh.TemperatureDependentViscosity(FT)),
                             boundary_conditions=bc, soil_param_set=lh.SoilParams(FT, ν=0.5),
                             earth_param_set=lh.EarthParameterSet())
        Y, Ya = lh.initialize_states(model, lambda z, m: {"ϑ_l": 0.25 + 0.1 * np.sin(3.0 * z), "θ_i": 0.0 * z}, 0.0)
        return model, Y, Ya

    model, Y, Ya = build(T_profile)
    sim = lh.Simulation(model, lh.SSPRK33(), Y_init=Y, dt=dt, tspan=(0.0, 12 * dt), Ya_init=Ya)
    lh.run(sim)
    got = np.array(sim.integrator.u.soil.ϑ_l)
    # by hand: every stage with the profile of its own time, uploaded as a full plane
    model2, Y2, Ya2 = build(T_profile)
    be = model2._backend()
    F = lh._ffi
    L = F.lib()
    U = Y2.similar()
    z = np.asarray(Ya2.zc)
    t = 0.0
    be.set_bcs(model2, 0.0)
    for _ in range(12):
        for stage, ts in ((1, t), (2, t + dt), (3, t + dt / 2)):
            plane = np.ascontiguousarray(np.repeat(np.asarray(T_profile(z, ts), FT)[None, :], N, axis=0))
            F.check(L.lh_upload(be.ctx, Ya2.handle, F.LH_VAR_T, plane.ctypes.data, 1, n), be.ctx)
            F.check(L.lh_ssprk33_stage(be.ctx, stage, Y2.handle, U.handle, Ya2.handle, dt, None), be.ctx)
        t += dt
    want = np.array(Y2.soil.ϑ_l)
    assert np.array_equal(got, want)
    # and the jump matters: frozen at its first value the run ends elsewhere
    model3, Y3, Ya3 = build(lambda z, t: 280.0 + 2.0 * z)
    sim3 = lh.Simulation(model3, lh.SSPRK33(), Y_init=Y3, dt=dt, tspan=(0.0, 12 * dt), Ya_init=Ya3)
    lh.run(sim3)
    assert np.max(np.abs(np.array(sim3.integrator.u.soil.ϑ_l) - got)) > 1e-9
    for m in (model, model2, model3):
        m.close()

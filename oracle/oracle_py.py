"""ctypes binding of the CPU oracle (oracle/liblh_oracle.so).

TEST INFRASTRUCTURE ONLY.  Import this from tests/, from
``__graft_entry__.smoke()`` and from the ``cpu_baseline`` leg of ``bench.py``
-- never from the product package ``landhydrology.jl_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblh_oracle.so")

MODEL_RICHARDS, MODEL_HEAT, MODEL_COUPLED = 0, 1, 2
BC_NONE, BC_FLUX, BC_DIRICHLET, BC_FREE_DRAINAGE = 0, 1, 2, 3
FACE_BOTTOM, FACE_TOP = 0, 1
COMP_ENERGY, COMP_HYDROLOGY = 0, 1


class EarthParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("rho_liq", "rho_ice", "cp_l", "cp_i", "T_0", "LH_f0", "K_therm")]


class SoilParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("nu", "S_s", "nu_ss_gravel", "nu_ss_om", "nu_ss_quartz", "rho_c_ds",
                 "kappa_solid", "rho_p", "kappa_sat_unfrozen", "kappa_sat_frozen", "a", "b",
                 "kappa_dry_parameter")]


class VGParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("n", "alpha", "theta_r", "Ksat")]


class CondFactors(C.Structure):
    _fields_ = [("viscosity_kind", C.c_int32), ("impedance_kind", C.c_int32),
                ("gamma", C.c_double), ("T_ref", C.c_double), ("Omega", C.c_double)]


class BC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad_", C.c_int32), ("value", C.c_double)]


class AtmosForcing(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("u_atm", "theta_atm", "z_atm", "theta_scale", "rho_a_sfc", "q_atm", "z_0m", "z_0s",
                 "R_v", "R_d", "grav", "cp_d", "cp_v", "LH_v0", "T_triple", "press_triple",
                 "von_karman")]


class Model(C.Structure):
    _fields_ = [("model", C.c_int32), ("nlev", C.c_int32), ("zmin", C.c_double),
                ("zmax", C.c_double), ("earth", EarthParams), ("soil", SoilParams),
                ("vg", VGParams), ("cf", CondFactors), ("bc", (BC * 2) * 2),
                ("consistent_bottom_sign", C.c_int32), ("atmos_on", C.c_int32),
                ("atmos", AtmosForcing)]


_DP = C.POINTER(C.c_double)


class PerCol(C.Structure):
    _fields_ = [("vg_n", _DP), ("vg_alpha", _DP), ("vg_theta_r", _DP), ("vg_Ksat", _DP),
                ("nu", _DP), ("S_s", _DP), ("bc_value", (_DP * 2) * 2),
                ("atm_u", _DP), ("atm_theta", _DP), ("atm_q", _DP)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("lh_oracle.c", "lh_oracle_impl.inc", "lh_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "liblh_oracle.so"], check=True,
                   stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    i64 = C.c_int64
    for sfx, ft in (("f64", C.c_double), ("f32", C.c_float)):
        P = C.POINTER(ft)
        vgp, ep, sp, cfp = (C.POINTER(VGParams), C.POINTER(EarthParams), C.POINTER(SoilParams),
                            C.POINTER(CondFactors))

        def sig(name, res, args):
            f = getattr(L, f"{name}_{sfx}")
            f.restype, f.argtypes = res, args

        sig("lho_grid", None, [C.c_double, C.c_double, C.c_int, P, P])
        sig("lho_volumetric_liquid_fraction", ft, [ft, ft])
        sig("lho_effective_saturation", ft, [ft, ft, ft])
        sig("lho_matric_potential", ft, [vgp, ft])
        sig("lho_inverse_matric_potential", ft, [vgp, ft])
        sig("lho_pressure_head", ft, [vgp, ft, ft, ft])
        sig("lho_hydraulic_conductivity", ft, [vgp, ft, ft, ft])
        sig("lho_viscosity_factor", ft, [cfp, ft])
        sig("lho_impedance_factor", ft, [cfp, ft])
        sig("lho_hydrostatic_profile", ft, [vgp, ft, ft, ft, ft])
        sig("lho_temperature_from_rhoe_int", ft, [ft, ft, ft, ep])
        sig("lho_volumetric_heat_capacity", ft, [ft, ft, ft, ep])
        sig("lho_volumetric_internal_energy", ft, [ft, ft, ft, ep])
        sig("lho_saturated_thermal_conductivity", ft, [ft, ft, ft, ft])
        sig("lho_relative_saturation", ft, [ft, ft, ft])
        sig("lho_kersten_number", ft, [ft, ft, sp])
        sig("lho_thermal_conductivity", ft, [ft, ft, ft])
        sig("lho_volumetric_internal_energy_liq", ft, [ft, ep])
        sig("lho_k_solid", ft, [ft, ft, ft, ft, ft])
        sig("lho_ksat_frozen", ft, [ft, ft, ft])
        sig("lho_ksat_unfrozen", ft, [ft, ft, ft])
        sig("lho_k_dry", ft, [ep, sp])
        mp, pcp = C.POINTER(Model), C.POINTER(PerCol)
        sig("lho_rhs", C.c_int, [mp, pcp, i64, P, P, P, P, P, P, P, i64, i64, C.c_int])
        sig("lho_diagnostics", C.c_int, [mp, pcp, i64, P, P, P, P, P, P, P, P, i64, i64])
        sig("lho_ssprk33", C.c_int, [mp, pcp, i64, P, P, P, P, i64, i64, C.c_double, C.c_double,
                                     i64, _DP, C.c_int])
        sig("lho_stable_dt", C.c_double, [mp, pcp, i64, P, P, P, P, i64, i64, C.c_double])
        sig("lho_boundary_fluxes", C.c_int, [mp, pcp, i64, P, P, P, P, i64, i64, C.c_int, P, P])
        sig("lho_turbulent_surface_fluxes", C.c_int, [mp, ft, ft, ft, P, P])
    L.lho_openmp_max_threads.restype = C.c_int


def _sfx(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise TypeError(f"oracle supports float32/float64, not {dtype}")


def fn(name: str, dtype):
    return getattr(lib(), f"{name}_{_sfx(dtype)}")


def _fill(struct_cls, obj):
    """ctypes image of a plain parameter object (any object with the struct's field names)."""
    return struct_cls(**{n: getattr(obj, n) for n, _ in struct_cls._fields_ if not n.startswith("pad")})


def as_c(obj):
    """The ctypes struct of a plain parameter object (matched by class name), by reference
    semantics ctypes applies to POINTER arguments; anything else passes through."""
    cls = {"EarthParams": EarthParams, "SoilParams": SoilParams, "VGParams": VGParams,
           "CondFactors": CondFactors}.get(type(obj).__name__)
    if cls is None or isinstance(obj, C.Structure):
        return obj
    return _fill(cls, obj)


def c_model(om) -> Model:
    """The C image of a model description: any object with the attributes of
    tests/case_model.CaseModel (model, nlev, zmin, zmax, earth, soil, vg, cf, bc,
    consistent_bottom_sign)."""
    m = Model()
    m.model, m.nlev, m.zmin, m.zmax = om.model, om.nlev, om.zmin, om.zmax
    m.earth, m.soil = _fill(EarthParams, om.earth), _fill(SoilParams, om.soil)
    m.vg, m.cf = _fill(VGParams, om.vg), _fill(CondFactors, om.cf)
    for f in range(2):
        for k in range(2):
            kind, val = om.bc.get((f, k), (BC_NONE, 0.0))
            m.bc[f][k].kind = kind
            m.bc[f][k].value = val
    m.consistent_bottom_sign = int(om.consistent_bottom_sign)
    atm = getattr(om, "atmos", None)
    m.atmos_on = int(atm is not None)
    if atm is not None:
        vals = {n: getattr(atm, n) for n, _ in AtmosForcing._fields_ if hasattr(atm, n)}
        vals["z_0m"], vals["z_0s"] = om.soil.z_0m, om.soil.z_0s      # SoilParams roughness lengths
        m.atmos = AtmosForcing(**vals)
    return m


def c_percol(om):
    if not om.percol and not om.percol_bc and not getattr(om, "percol_atmos", None):
        return None, []
    keep = []
    pc = PerCol()
    for name in ("vg_n", "vg_alpha", "vg_theta_r", "vg_Ksat", "nu", "S_s"):
        if name in om.percol:
            a = np.ascontiguousarray(om.percol[name], dtype=np.float64)
            keep.append(a)
            setattr(pc, name, a.ctypes.data_as(_DP))
    for (f, k), v in om.percol_bc.items():
        a = np.ascontiguousarray(v, dtype=np.float64)
        keep.append(a)
        pc.bc_value[f][k] = a.ctypes.data_as(_DP)
    for name, v in (getattr(om, "percol_atmos", None) or {}).items():     # u_atm / theta_atm / q_atm
        a = np.ascontiguousarray(v, dtype=np.float64)
        keep.append(a)
        setattr(pc, {"u_atm": "atm_u", "theta_atm": "atm_theta", "q_atm": "atm_q"}[name],
                a.ctypes.data_as(_DP))
    return pc, keep


def _ptr(a: Optional[np.ndarray], ft):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ft))


def _ft(dtype):
    return C.c_double if np.dtype(dtype) == np.float64 else C.c_float


def _strides(a: np.ndarray):
    """Arrays are [ncols, nlev] (level-fastest, like parent(field)) or any
    2-D strided view; returns element strides (lev, col)."""
    assert a.ndim == 2
    it = a.itemsize
    # numpy reports arbitrary strides for length-1 axes: normalise them
    ls = a.strides[1] // it if a.shape[1] > 1 else 1
    cs = a.strides[0] // it if a.shape[0] > 1 else max(1, a.shape[1] * ls)
    return ls, cs


def grid(zmin, zmax, n, dtype=np.float64):
    zc = np.empty(n, dtype=dtype)
    zf = np.empty(n + 1, dtype=dtype)
    ft = _ft(dtype)
    fn("lho_grid", dtype)(zmin, zmax, n, _ptr(zc, ft), _ptr(zf, ft))
    return zc, zf


def rhs(om, vl=None, ti=None, rhoe=None, T_aux=None, nthreads=1):
    """rhs!(dY, Y, Ya, t) for a batch.  Inputs are [ncols, nlev] arrays (any
    strides, all the same).  Returns dict of tendencies."""
    ref = vl if vl is not None else rhoe
    dtype = ref.dtype
    ft = _ft(dtype)
    ncols = ref.shape[0]
    ls, cs = _strides(ref)
    for a in (vl, ti, rhoe, T_aux):
        if a is not None:
            assert a.dtype == dtype and _strides(a) == (ls, cs) and a.shape == ref.shape
    out = {}
    d_vl = d_ti = d_re = None
    if om.model != MODEL_HEAT:
        d_vl = np.empty_like(ref)
        d_ti = np.empty_like(ref)
        assert _strides(d_vl) == (ls, cs)
        out["vl"], out["ti"] = d_vl, d_ti
    if om.model != MODEL_RICHARDS:
        d_re = np.empty_like(ref)
        out["rhoe"] = d_re
    m = c_model(om)
    pc, keep = c_percol(om)
    rc = fn("lho_rhs", dtype)(C.byref(m), C.byref(pc) if pc is not None else None, ncols,
                              _ptr(vl, ft), _ptr(ti, ft), _ptr(rhoe, ft), _ptr(T_aux, ft),
                              _ptr(d_vl, ft), _ptr(d_ti, ft), _ptr(d_re, ft), ls, cs, nthreads)
    del keep
    if rc:
        raise ValueError(f"oracle rhs: invalid model/boundary combination (code {rc})")
    return out


def diagnostics(om, vl=None, ti=None, rhoe=None, T_aux=None):
    ref = vl if vl is not None else rhoe
    dtype = ref.dtype
    ft = _ft(dtype)
    ls, cs = _strides(ref)
    K, psi, T, kap = (np.empty_like(ref) for _ in range(4))
    m = c_model(om)
    pc, keep = c_percol(om)
    rc = fn("lho_diagnostics", dtype)(C.byref(m), C.byref(pc) if pc is not None else None,
                                      ref.shape[0], _ptr(vl, ft), _ptr(ti, ft), _ptr(rhoe, ft),
                                      _ptr(T_aux, ft), _ptr(K, ft), _ptr(psi, ft), _ptr(T, ft),
                                      _ptr(kap, ft), ls, cs)
    del keep
    if rc:
        raise ValueError(f"oracle diagnostics failed (code {rc})")
    return dict(K=K, psi=psi, T=T, kappa=kap)


def boundary_fluxes(om, face, vl, ti, rhoe=None, T_aux=None):
    """boundary_fluxes(X, bc, face, model, cs, t) per column: (f_rhoe_int[ncols], f_vartheta_l[ncols])
    of one face (NaN where the component has no boundary condition).  HEAT: vl, ti are the
    prescribed (aux) fields."""
    dtype = vl.dtype
    ft = _ft(dtype)
    ls, cs = _strides(vl)
    fe, fw = np.empty(vl.shape[0], dtype), np.empty(vl.shape[0], dtype)
    m = c_model(om)
    pc, keep = c_percol(om)
    rc = fn("lho_boundary_fluxes", dtype)(C.byref(m), C.byref(pc) if pc is not None else None, vl.shape[0],
                                          _ptr(vl, ft), _ptr(ti, ft), _ptr(rhoe, ft), _ptr(T_aux, ft), ls, cs,
                                          int(face), _ptr(fe, ft), _ptr(fw, ft))
    del keep
    if rc:
        raise ValueError(f"oracle boundary_fluxes failed (code {rc})")
    return fe, fw


def ssprk33(om, dt, nsteps, vl=None, ti=None, rhoe=None, T_aux=None, t0=0.0,
            bc_stage_values=None, nthreads=1):
    """Advance the state arrays IN PLACE by nsteps fixed-dt SSPRK33 steps."""
    ref = vl if vl is not None else rhoe
    dtype = ref.dtype
    ft = _ft(dtype)
    ls, cs = _strides(ref)
    bcv = None
    if bc_stage_values is not None:
        bcv = np.ascontiguousarray(bc_stage_values, dtype=np.float64)
        assert bcv.shape == (nsteps, 3, 2, 2)
    m = c_model(om)
    pc, keep = c_percol(om)
    rc = fn("lho_ssprk33", dtype)(C.byref(m), C.byref(pc) if pc is not None else None,
                                  ref.shape[0], _ptr(vl, ft), _ptr(ti, ft), _ptr(rhoe, ft),
                                  _ptr(T_aux, ft), ls, cs, float(t0), float(dt), int(nsteps),
                                  _ptr(bcv, C.c_double), nthreads)
    del keep
    if rc:
        raise ValueError(f"oracle ssprk33 failed (code {rc})")


def stable_dt(om, vl, ti, rhoe=None, courant=0.5, T_aux=None):
    dtype = vl.dtype
    ft = _ft(dtype)
    ls, cs = _strides(vl)
    m = c_model(om)
    pc, keep = c_percol(om)
    r = fn("lho_stable_dt", dtype)(C.byref(m), C.byref(pc) if pc is not None else None,
                                   vl.shape[0], _ptr(vl, ft), _ptr(ti, ft), _ptr(rhoe, ft),
                                   _ptr(T_aux, ft), ls, cs, float(courant))
    del keep
    return r


def turbulent_surface_fluxes(om, vl, ti, T, dtype=np.float64):
    """compute_turbulent_surface_fluxes (boundary_conditions.jl:553-620) for arrays of top-cell
    states -> (heat_flux, water_flux, status) arrays; status 1 = no Monin-Obukhov root."""
    ft = _ft(dtype)
    m = c_model(om)
    vl, ti, T = (np.atleast_1d(np.asarray(a, dtype=dtype)) for a in (vl, ti, T))
    h, w = np.empty_like(vl), np.empty_like(vl)
    st = np.zeros(vl.shape, dtype=np.int32)
    f = fn("lho_turbulent_surface_fluxes", dtype)
    hh, ww = ft(), ft()
    for i in range(vl.size):
        st.flat[i] = f(C.byref(m), ft(vl.flat[i]), ft(ti.flat[i]), ft(T.flat[i]), C.byref(hh), C.byref(ww))
        h.flat[i], w.flat[i] = hh.value, ww.value
    return h, w, st


def max_threads() -> int:
    return lib().lho_openmp_max_threads()

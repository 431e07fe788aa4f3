"""Synthetic workloads of the batched soil-column path and a driver for the HIP
library THROUGH THE C ABI (include/landhydro.h): what bench.py times, what tools/ measures and what
the parity tests compare with the CPU oracle.  Product-side measurement code: nothing here imports
the oracle (tests/parity_cases.py adds the oracle runner and the tolerance model on top).

Inputs are generated identically on every rank from a counter-based hash,
u(c, i) = splitmix64(seed xor (c n + i)) / 2^64, seed 0x4C48_5944_524F.
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import _ffi
from . import case_model as M

SEED = 0x4C485944524F  # "LHYDRO"


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uhash(c: np.ndarray, i, n: int, seed: int = SEED) -> np.ndarray:
    """u(c, i) = splitmix64(seed xor (c*n + i)) / 2^64 in [0, 1)."""
    with np.errstate(over="ignore"):
        k = c.astype(np.uint64) * np.uint64(n) + np.uint64(i)
        h = splitmix64(np.uint64(seed) ^ k)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


@dataclass
class Case:
    name: str
    om: M.CaseModel
    dtype: type
    ncols: int
    # host arrays [ncols, nlev] (level-fastest, like parent(field)); None = unused
    vl: Optional[np.ndarray] = None
    ti: Optional[np.ndarray] = None
    rhoe: Optional[np.ndarray] = None
    T_aux: Optional[np.ndarray] = None
    col_offset: int = 0
    # the prescribed fields of Ya are level-uniform (what the reference's profile closures of (z, t)
    # give): uploaded as nlev numbers with lh_upload_profile instead of planes
    aux_profile: bool = False


# k_solid(nu_om = 0, nu_q = 0.92, kappa_quartz = 7.7, kappa_minerals = 2.5, kappa_om = 0.25) and
# ksat_unfrozen / ksat_frozen(k_solid, nu = 0.5, kappa_l = 0.57 / kappa_ice = 2.29) of
# test/SoilModel/coupled.jl:16-22, as literals: input generation (also bench.py's) executes
# nothing of the oracle.  tests/test_oracle_pins.py checks them against the oracle's functions.
COUPLED_K_SOLID = 7.037309762302548
COUPLED_KSAT_UNFROZEN = 2.0028146605496104
COUPLED_KSAT_FROZEN = 4.014403985110721


def coupled_soil():
    """test/SoilModel/coupled.jl:3-32."""
    nu = 0.5
    sp = M.default_soil(nu=nu, S_s=1e-3, nu_ss_gravel=0.0, nu_ss_om=0.0, nu_ss_quartz=0.92,
                        rho_c_ds=(1 - nu) * 1.926e06, kappa_solid=COUPLED_K_SOLID,
                        kappa_sat_unfrozen=COUPLED_KSAT_UNFROZEN,
                        kappa_sat_frozen=COUPLED_KSAT_FROZEN)
    vg = M.default_vg(n=2.0, alpha=2.6, Ksat=0.0443 / 3600 / 100, theta_r=0.0)
    return sp, vg


def grid_np(zmin, zmax, n, dtype=np.float64):
    """The uniform mesh of domain.jl:58-69 in numpy, for input generation: faces = the
    correctly rounded zmin + k L / n (extended precision, as the oracle's lho_grid), centres =
    face midpoints in FT, bottom first.  Bitwise lho_grid (tests/test_oracle_pins.py)."""
    ft = np.dtype(dtype).type
    lo, hi = ft(zmin), ft(zmax)
    k = np.arange(1, n + 1).astype(np.longdouble)
    x = np.longdouble(lo) + (np.longdouble(hi) - np.longdouble(lo)) * k / np.longdouble(n)
    zf = np.empty(n + 1, dtype=dtype)
    zf[0] = lo
    zf[1:] = x.astype(dtype)
    zf[n] = hi
    zc = ((zf[:-1] + zf[1:]) / ft(2)).astype(dtype)
    return zc, zf


def _flux_bcs(energy=None, hydrology=None):
    bc = {}
    for f in (M.FACE_BOTTOM, M.FACE_TOP):
        if energy is not None:
            bc[(f, M.COMP_ENERGY)] = (M.BC_FLUX, energy)
        if hydrology is not None:
            bc[(f, M.COMP_HYDROLOGY)] = (M.BC_FLUX, hydrology)
    return bc


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def wetting_front(ncols, n, zmin, zmax, nu, col_offset=0):
    """C2: vl(c,i) = nu (0.35 + 0.5 sigma((z_i - z_f(c))/0.1)),
    z_f(c) = zmin + (0.2 + 0.6 u(c,0)) L."""
    zc, _ = grid_np(zmin, zmax, n)
    c = np.arange(col_offset, col_offset + ncols)
    L = zmax - zmin
    zf = zmin + (0.2 + 0.6 * uhash(c, 0, n)) * L
    return nu * (0.35 + 0.5 * sigmoid((zc[None, :] - zf[:, None]) / 0.1))


def make_case(name: str, ncols: Optional[int] = None, col_offset: int = 0, _nlev: Optional[int] = None) -> Case:
    """The BASELINE configs at test sizes plus edge cases."""
    f64, f32 = np.float64, np.float32
    # "<family>_nNNN": the family's case with NNN levels (tall and ragged columns: 65..128 levels are
    # one wavefront with two cells per lane in the persistent stepper, more than 128 one thread per cell)
    if name.endswith("_icesorted"):   # the family's case with ice-free columns first (ice_sorted_order)
        base = make_case(name[:-len("_icesorted")], ncols, col_offset, _nlev=_nlev)
        return reorder_columns(base, ice_sorted_order(base.ti))
    # boundary-condition variants of a family (which closures a Dirichlet face state needs, and whether
    # they are constants of a stepping call: face_state_is_static in lh_closures.hpp):
    #   _hyddir  top face: the hydrology Dirichlet value stays, the energy component becomes a flux
    #            (a viscosity factor then sees the moving T of the boundary cell)
    #   _endir   top face: the energy Dirichlet value stays, the hydrology component becomes a flux
    #            (kappa of the face state then sees the moving vartheta_l of the boundary cell)
    #   _pcdir   every Dirichlet value becomes a per-column array
    for suffix in ("_hyddir", "_endir", "_pcdir"):
        if name.endswith(suffix):
            import copy
            case = make_case(name[:-len(suffix)], ncols, col_offset, _nlev=_nlev)
            om = copy.deepcopy(case.om)
            top_e, top_h = (M.FACE_TOP, M.COMP_ENERGY), (M.FACE_TOP, M.COMP_HYDROLOGY)
            if suffix == "_hyddir" and top_e in om.bc:
                om.bc[top_e] = (M.BC_FLUX, 0.05)
            elif suffix == "_endir" and top_h in om.bc:
                om.bc[top_h] = (M.BC_FLUX, -1e-8)
            elif suffix == "_pcdir":
                cols = np.arange(case.ncols) + col_offset
                for k, (kind, v) in om.bc.items():
                    if kind == M.BC_DIRICHLET:
                        om.percol_bc[k] = v * (1.0 + 0.05 * (uhash(cols, 77 + 2 * k[0] + k[1], 1) - 0.5))
            case.om, case.name = om, name
            return case
    nlev_override = None
    mm_ = re.match(r"^(.*)_n(\d+)$", name)
    if mm_:
        name, nlev_override = mm_.group(1), int(mm_.group(2))
        case = make_case(name, ncols, col_offset, _nlev=nlev_override)
        case.name = f"{name}_n{nlev_override}"
        return case
    if name == "c1_dirichlet_f64":
        # C1: 1 column, n=64, zlim=(-1.28,0), loam, Dirichlet 0.35 top / 0.20 bottom
        n, N = 64, ncols or 1
        bc = {(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.35),
              (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.20)}
        om = M.CaseModel(M.MODEL_RICHARDS, n, -1.28, 0.0, bc=bc)
        return Case(name, om, f64, N, vl=np.full((N, n), 0.20), ti=np.zeros((N, n)))
    if name in ("c2_richards_f64", "c2_richards_f32", "c4_richards_f64_128"):
        n = 128 if name.startswith("c4") else 64
        zmin = -2.56 if n == 128 else -1.28
        if _nlev:
            n, zmin = _nlev, -0.02 * _nlev
        N = ncols or (1000 if not _nlev else 300)
        dt = f32 if name.endswith("f32") else f64
        om = M.CaseModel(M.MODEL_RICHARDS, n, zmin, 0.0, bc=_flux_bcs(hydrology=0.0))
        vl = wetting_front(N, n, zmin, 0.0, om.soil.nu, col_offset).astype(dt)
        return Case(name, om, dt, N, vl=vl, ti=np.zeros((N, n), dt), col_offset=col_offset)
    if name in ("c3_coupled_f32", "coupled_f64_small", "c3_coupled_f64"):
        n = 64
        N = ncols or (256 if name == "coupled_f64_small" else 1000)
        dt = f32 if name.endswith("f32") else f64
        sp, vg = coupled_soil()
        zmin, zmax = -1.28, 0.0
        om = M.CaseModel(M.MODEL_COUPLED, n, zmin, zmax, soil=sp, vg=vg,
                           bc=_flux_bcs(energy=0.0, hydrology=0.0))
        vl = wetting_front(N, n, zmin, zmax, sp.nu, col_offset)
        zc, _ = grid_np(zmin, zmax, n)
        c = np.arange(col_offset, col_offset + N)
        T = 284.0 + 5.0 * zc[None, :] / (zmax - zmin) + 2.0 * (uhash(c, 1, n)[:, None] - 0.5)
        e = om.earth
        rho_c_s = sp.rho_c_ds + vl * (e.cp_l * e.rho_liq)
        rhoe = rho_c_s * (T - e.T_0)
        return Case(name, om, dt, N, vl=vl.astype(dt), ti=np.zeros((N, n), dt),
                    rhoe=rhoe.astype(dt), col_offset=col_offset)
    if name == "c5_percol_f64":
        # C5: per-column vG/porosity, top flux -0.5 Ksat_c, free drainage bottom
        n, N = 128, ncols or 1000
        c = np.arange(col_offset, col_offset + N)
        vg_n = 1.4 + 2.6 * uhash(c, 2, n)
        alpha = 1.5 + 6.0 * uhash(c, 3, n)
        Ksat = 10.0 ** (-7.0 + 3.0 * uhash(c, 4, n))
        theta_r = 0.08 * uhash(c, 5, n)
        nu = 0.3 + 0.25 * uhash(c, 6, n)
        bc = {(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_FLUX, 0.0),
              (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FREE_DRAINAGE, 0.0)}
        om = M.CaseModel(M.MODEL_RICHARDS, n, -2.56, 0.0, bc=bc,
                           percol=dict(vg_n=vg_n, vg_alpha=alpha, vg_Ksat=Ksat, vg_theta_r=theta_r,
                                       nu=nu),
                           percol_bc={(M.FACE_TOP, M.COMP_HYDROLOGY): -0.5 * Ksat})
        vl = np.repeat((theta_r + 0.25 * (nu - theta_r))[:, None], n, axis=1)
        # a wetter band near the top so fluxes are not all tiny
        zc, _ = grid_np(-2.56, 0.0, n)
        vl = vl + (0.5 * (nu - theta_r))[:, None] * sigmoid((zc[None, :] + 0.4) / 0.1)
        return Case(name, om, f64, N, vl=vl, ti=np.zeros((N, n)), col_offset=col_offset)
    if name in ("heat_dirichlet_f64", "heat_dirichlet_f32"):
        n, N = 60, ncols or 300
        dt = f32 if name.endswith("f32") else f64
        sp, _ = coupled_soil()
        bc = {(M.FACE_TOP, M.COMP_ENERGY): (M.BC_DIRICHLET, 280.0),
              (M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_DIRICHLET, 290.0)}
        om = M.CaseModel(M.MODEL_HEAT, n, 0.0, 1.0, soil=sp, bc=bc)
        c = np.arange(N)
        vl = 0.1 + 0.35 * uhash(c[:, None], np.arange(n)[None, :] + 7, 1000)
        ti = np.where(uhash(c, 8, n)[:, None] < 0.3, 0.05 * uhash(c[:, None], np.arange(n)[None, :] + 99, 1000), 0.0)
        zc, _ = grid_np(0.0, 1.0, n)
        T = 285.0 + 3.0 * np.sin(6.0 * zc)[None, :] + uhash(c, 9, n)[:, None]
        e = om.earth
        tl = np.minimum(vl, sp.nu - ti)
        rho_c_s = sp.rho_c_ds + tl * (e.cp_l * e.rho_liq) + ti * (e.cp_i * e.rho_ice)
        rhoe = rho_c_s * (T - e.T_0) - ti * e.rho_ice * e.LH_f0
        return Case(name, om, dt, N, vl=vl.astype(dt), ti=ti.astype(dt), rhoe=rhoe.astype(dt))
    if name in ("mixed_factors_f64", "mixed_factors_f32"):
        # saturated cells, ice, both conductivity factors, Dirichlet top + free drainage,
        # coupled model: exercises every branch of the closures
        n, N = 37, ncols or 515   # ragged sizes on purpose
        dt = f32 if name.endswith("f32") else f64
        sp, vg = coupled_soil()
        bc = {(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.47),
              (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FREE_DRAINAGE, 0.0),
              (M.FACE_TOP, M.COMP_ENERGY): (M.BC_DIRICHLET, 275.0),
              (M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_FLUX, 0.05)}
        om = M.CaseModel(M.MODEL_COUPLED, n, -3.0, -0.5, soil=sp, vg=vg, bc=bc,
                           cf=M.default_cf(viscosity=True, impedance=True))
        c = np.arange(N)[:, None]
        lev = np.arange(n)[None, :]
        ti = np.where(uhash(c, 11, 1) < 0.5, 0.12 * uhash(c, lev + 13, 1000), 0.0)
        vl = 0.08 + 0.47 * uhash(c, lev + 300, 1000)       # up to 0.55 > nu: saturated cells
        vl = np.where(uhash(c, lev + 700, 1000) < 0.02, 1e-9, vl)  # nearly dry cells
        T = 270.0 + 12.0 * uhash(c, lev + 500, 1000)
        e = om.earth
        tl = np.minimum(vl, sp.nu - ti)
        rho_c_s = sp.rho_c_ds + tl * (e.cp_l * e.rho_liq) + ti * (e.cp_i * e.rho_ice)
        rhoe = rho_c_s * (T - e.T_0) - ti * e.rho_ice * e.LH_f0
        return Case(name, om, dt, N, vl=vl.astype(dt), ti=ti.astype(dt), rhoe=rhoe.astype(dt))
    if name in ("mixed_smooth_f64", "mixed_smooth_f32"):
        # steppable variant of the above: smooth fields with ice lenses, saturated
        # zones, both conductivity factors, Dirichlet top / free drainage bottom
        n, N = _nlev or 48, ncols or 200
        dt = f32 if name.endswith("f32") else f64
        sp, vg = coupled_soil()
        bc = {(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.42),
              (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FREE_DRAINAGE, 0.0),
              (M.FACE_TOP, M.COMP_ENERGY): (M.BC_DIRICHLET, 276.0),
              (M.FACE_BOTTOM, M.COMP_ENERGY): (M.BC_FLUX, 0.05)}
        om = M.CaseModel(M.MODEL_COUPLED, n, -2.4, 0.0, soil=sp, vg=vg, bc=bc,
                           cf=M.default_cf(viscosity=True, impedance=True))
        c = np.arange(N)
        zc, _ = grid_np(-2.4, 0.0, n)
        ph = 6.28 * uhash(c, 21, n)[:, None]
        ti = np.where(uhash(c, 22, n)[:, None] < 0.5,
                      0.06 * np.maximum(0.0, np.sin(3.0 * zc[None, :] + ph)), 0.0)
        vl = 0.30 + 0.14 * np.sin(2.0 * zc[None, :] + ph) + 0.09 * uhash(c, 23, n)[:, None]
        T = 278.0 + 6.0 * np.cos(1.5 * zc[None, :] + ph)
        e = om.earth
        tl = np.minimum(vl, sp.nu - ti)
        rho_c_s = sp.rho_c_ds + tl * (e.cp_l * e.rho_liq) + ti * (e.cp_i * e.rho_ice)
        rhoe = rho_c_s * (T - e.T_0) - ti * e.rho_ice * e.LH_f0
        return Case(name, om, dt, N, vl=vl.astype(dt), ti=ti.astype(dt), rhoe=rhoe.astype(dt))
    if name == "richards_viscosity_profile_f64":
        # the same model with the prescribed temperature as the reference has it: T_profile(z, t), one
        # value per level for every column (right_hand_side.jl:54-62), handed over as a profile
        base = make_case("richards_viscosity_f64", ncols, col_offset)
        zc, _ = grid_np(base.om.zmin, base.om.zmax, base.om.nlev)
        T = 284.0 + 10.0 * np.sin(0.55 * zc)
        return Case(name, base.om, np.float64, base.ncols, vl=base.vl, ti=base.ti,
                    T_aux=np.repeat(T[None, :], base.ncols, axis=0), aux_profile=True)
    if name == "richards_viscosity_f64":
        n, N = 50, ncols or 130
        om = M.CaseModel(M.MODEL_RICHARDS, n, -10.0, 0.0,
                           bc={(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_FLUX, -1e-7),
                               (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.40)},
                           cf=M.default_cf(viscosity=True))
        c = np.arange(N)[:, None]
        lev = np.arange(n)[None, :]
        vl = 0.1 + 0.3 * uhash(c, lev, 1000)
        T = 275.0 + 25.0 * uhash(c, lev + 50, 1000)
        return Case(name, om, np.float64, N, vl=vl, ti=np.zeros((N, n)), T_aux=T)
    if name == "single_cell_f64":
        # n = 1: both faces are boundary faces
        N = ncols or 70
        om = M.CaseModel(M.MODEL_RICHARDS, 1, -0.1, 0.0,
                           bc={(M.FACE_TOP, M.COMP_HYDROLOGY): (M.BC_DIRICHLET, 0.3),
                               (M.FACE_BOTTOM, M.COMP_HYDROLOGY): (M.BC_FREE_DRAINAGE, 0.0)})
        vl = (0.1 + 0.3 * uhash(np.arange(N), 0, 1))[:, None]
        return Case(name, om, np.float64, N, vl=vl, ti=np.zeros((N, 1)))
    raise KeyError(name)


# ------------------------------------------------------------ oracle runner

def ice_sorted_order(theta_i: np.ndarray) -> np.ndarray:
    """order[k] = the column that should sit at position k so that ice-free columns come first and
    columns with any ice last (a stable partition).  theta_i never changes (its tendency is identically
    zero: right_hand_side.jl:182, :359), so the pattern is static: with the columns laid out in this
    order a wavefront (64 consecutive columns) is either ice-free -- K and psi then share one power
    chain, water_closures_log -- or icy, and at most one wave per ensemble is mixed.  Columns are
    independent: any order gives the same per-column results bit for bit."""
    icy = np.any(np.asarray(theta_i) != 0, axis=1)
    return np.argsort(icy, kind="stable")


def reorder_columns(case: Case, order: np.ndarray) -> Case:
    """The same ensemble with its columns in another order (every per-column array permuted)."""
    import dataclasses
    take = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a)[order])
    om = case.om
    om2 = dataclasses.replace(om, percol={k: take(v) for k, v in om.percol.items()},
                              percol_bc={k: take(v) for k, v in om.percol_bc.items()})
    if getattr(om, "percol_atmos", None):
        om2 = dataclasses.replace(om2, percol_atmos={k: take(v) for k, v in om.percol_atmos.items()})
    return dataclasses.replace(case, name=case.name + "_sorted", om=om2, vl=take(case.vl), ti=take(case.ti),
                               rhoe=take(case.rhoe), T_aux=take(case.T_aux))


class GpuModel:
    """A context on the HIP library configured from an OracleModel description,
    using only C-ABI calls (include/landhydro.h)."""

    def __init__(self, case: Case, math_mode: Optional[int] = None, stream=None):
        F = _ffi
        self.F, self.L = F, F.lib()
        om = case.om
        self.case = case
        cfg = F.lh_config(case.ncols, om.nlev, F.dtype_code(case.dtype), om.zmin, om.zmax,
                          om.model, -1, stream)
        self.ctx = C.c_void_p()
        F.check(self.L.lh_create(C.byref(self.ctx), C.byref(cfg)), None)
        L, ctx = self.L, self.ctx
        e, s, v, cf = om.earth, om.soil, om.vg, om.cf
        F.check(L.lh_set_earth_params(ctx, C.byref(F.lh_earth_params(
            e.rho_liq, e.rho_ice, e.cp_l, e.cp_i, e.T_0, e.LH_f0, e.K_therm))), ctx)
        F.check(L.lh_set_soil_params(ctx, C.byref(F.lh_soil_params(
            *[getattr(s, n[0]) for n in F.lh_soil_params._fields_]))), ctx)
        F.check(L.lh_set_vg_params(ctx, C.byref(F.lh_vg_params(v.n, v.alpha, v.theta_r, v.Ksat))),
                ctx)
        F.check(L.lh_set_conductivity_factors(ctx, cf.viscosity_kind, cf.gamma, cf.T_ref,
                                              cf.impedance_kind, cf.Omega), ctx)
        for key, arr in om.percol.items():
            a = np.ascontiguousarray(arr, dtype=np.float64)
            F.check(L.lh_set_percol_param(ctx, F.LH_PC[key], a.ctypes.data_as(
                C.POINTER(C.c_double))), ctx)
        self.set_bcs(om)
        if getattr(om, "atmos", None) is not None:      # PrescribedAtmosForcing at the top face
            a = om.atmos
            f = F.lh_atmos_forcing(a.u_atm, a.theta_atm, a.z_atm, a.theta_scale, a.rho_a_sfc, a.q_atm,
                                   om.soil.z_0m, om.soil.z_0s, a.R_v, a.R_d, a.grav, a.cp_d, a.cp_v,
                                   a.LH_v0, a.T_triple, a.press_triple, a.von_karman)
            pc = None
            if om.percol_atmos:
                pc = np.ascontiguousarray(np.stack([
                    np.asarray(om.percol_atmos.get(k, np.full(case.ncols, getattr(a, k))), dtype=np.float64)
                    for k in ("u_atm", "theta_atm", "q_atm")]))
            F.check(L.lh_set_atmos_forcing(ctx, C.byref(f),
                                           pc.ctypes.data_as(C.POINTER(C.c_double)) if pc is not None else None), ctx)
        F.check(L.lh_set_bottom_sign_consistent(ctx, int(om.consistent_bottom_sign)), ctx)
        if math_mode is not None:
            F.check(L.lh_set_math_mode(ctx, math_mode), ctx)
        self._states = []

    def set_bcs(self, om):
        F, L, ctx = self.F, self.L, self.ctx
        for f in range(2):
            for k in range(2):
                kind, val = om.bc.get((f, k), (F.LH_BC_NONE, 0.0))
                pc = om.percol_bc.get((f, k))
                p = None
                if pc is not None:
                    pc = np.ascontiguousarray(pc, dtype=np.float64)
                    p = pc.ctypes.data_as(C.POINTER(C.c_double))
                F.check(L.lh_set_bc(ctx, f, k, kind, float(val), p), ctx)

    def state(self, mask=0, **fields):
        h = C.c_void_p()
        self.F.check(self.L.lh_state_create(self.ctx, mask, C.byref(h)), self.ctx)
        self._states.append(h)
        for var, a in fields.items():
            self.upload(h, var, a)
        return h

    @staticmethod
    def _strides(a):
        """Element strides (lev, col); numpy reports arbitrary strides for
        length-1 axes, so normalise those."""
        it = a.itemsize
        ls = a.strides[1] // it if a.shape[1] > 1 else 1
        cs = a.strides[0] // it if a.shape[0] > 1 else max(1, a.shape[1] * max(ls, 1))
        return ls, cs

    def upload(self, h, var, a):
        a = np.asarray(a)
        assert a.dtype == self.case.dtype and a.ndim == 2
        if not a.any():
            # an all-zero field is a fill: the library then KNOWS the plane is zero and neither
            # reads a zero theta_i plane nor re-stores d theta_i = 0 (the host mirror does the same)
            self.F.check(self.L.lh_state_fill(self.ctx, h, var, 0.0), self.ctx)
            return
        ls, cs = self._strides(a)
        self.F.check(self.L.lh_upload(self.ctx, h, var, a.ctypes.data, ls, cs), self.ctx)

    def download(self, h, var, out=None):
        om = self.case.om
        if out is None:
            out = np.empty((self.case.ncols, om.nlev), dtype=self.case.dtype)
        ls, cs = self._strides(out)
        self.F.check(self.L.lh_download(self.ctx, h, var, out.ctypes.data, ls, cs), self.ctx)
        return out

    def prognostic_and_aux(self):
        """(Y, Ya) states uploaded from the case arrays."""
        F, c = self.F, self.case
        m = c.om.model
        if m == F.LH_MODEL_HEAT:
            Y = self.state(0)
            self.upload(Y, F.LH_VAR_RHOE_INT, c.rhoe)
            Ya = self.state(0b0011)
            if c.aux_profile:
                self.upload_profile(Ya, F.LH_VAR_VARTHETA_L, c.vl[0])
                self.upload_profile(Ya, F.LH_VAR_THETA_I, c.ti[0])
            else:
                self.upload(Ya, F.LH_VAR_VARTHETA_L, c.vl)
                self.upload(Ya, F.LH_VAR_THETA_I, c.ti)
            return Y, Ya
        Y = self.state(0)
        self.upload(Y, F.LH_VAR_VARTHETA_L, c.vl)
        self.upload(Y, F.LH_VAR_THETA_I, c.ti)
        if m == F.LH_MODEL_COUPLED:
            self.upload(Y, F.LH_VAR_RHOE_INT, c.rhoe)
        Ya = None
        if c.T_aux is not None:
            Ya = self.state(0b1000)
            if c.aux_profile:
                self.upload_profile(Ya, F.LH_VAR_T, c.T_aux[0])
            else:
                self.upload(Ya, F.LH_VAR_T, c.T_aux)
        return Y, Ya

    def upload_profile(self, st, var, values):
        """One value per level for every column (lh_upload_profile)."""
        a = np.ascontiguousarray(values, dtype=self.case.dtype)
        assert a.shape == (self.case.om.nlev,)
        self.F.check(self.L.lh_upload_profile(self.ctx, st, var, a.ctypes.data), self.ctx)

    def rhs(self, Y, Ya, dY, t=0.0):
        self.F.check(self.L.lh_rhs(self.ctx, t, Y, Ya, dY), self.ctx)

    def tendencies(self, dY):
        F, m = self.F, self.case.om.model
        out = {}
        if m != F.LH_MODEL_HEAT:
            out["vl"] = self.download(dY, F.LH_VAR_VARTHETA_L)
            out["ti"] = self.download(dY, F.LH_VAR_THETA_I)
        if m != F.LH_MODEL_RICHARDS:
            out["rhoe"] = self.download(dY, F.LH_VAR_RHOE_INT)
        return out

    def status(self) -> int:
        f = C.c_uint32()
        self.F.check(self.L.lh_get_status(self.ctx, C.byref(f)), self.ctx)
        return f.value

    def close(self):
        if self.ctx:
            self.L.lh_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def run_gpu_rhs(case: Case, math_mode: Optional[int] = None):
    with GpuModel(case, math_mode) as g:
        Y, Ya = g.prognostic_and_aux()
        dY = g.state(0)
        g.rhs(Y, Ya, dY)
        out = g.tendencies(dY)
        assert g.status() == 0, "non-finite tendency flagged"
        return out


def run_gpu_diagnostics(case: Case, math_mode: Optional[int] = None):
    with GpuModel(case, math_mode) as g:
        F = g.F
        Y, Ya = g.prognostic_and_aux()
        D = g.state(0b1111)
        F.check(g.L.lh_diagnostics(g.ctx, Y, Ya, D), g.ctx)
        return dict(K=g.download(D, F.LH_DIAG_K), psi=g.download(D, F.LH_DIAG_PSI),
                    kappa=g.download(D, F.LH_DIAG_KAPPA), T=g.download(D, F.LH_DIAG_T))


# ------------------------------------------------------------ tolerance model

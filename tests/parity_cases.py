"""Shared parity machinery: synthetic cases, a runner for the HIP
path THROUGH THE C ABI, a runner for the CPU oracle, and the tolerance model.

Test infrastructure, never imported by the product.  The oracle is imported lazily by the
checker legs only (see `O`).

Tolerance model (why not one global rtol): a tendency is a difference of face
fluxes, each a product of a conductivity and a head gradient, so agreement
between two correct implementations is limited by (a) the conditioning of the
van Genuchten closures -- K has 1 - (1 - S^(1/m))^m, psi has S^(-1/m) - 1, both
cancelling -- and (b) the cancellation psi_i - psi_(i-1) + dz.  `tendency_tolerance`
propagates C*eps(FT) through exactly those expressions; C is stated per test.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import sys
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import case_model as M  # noqa: E402  (plain descriptions; imports nothing from oracle/)


class _LazyOracle:
    """`O.rhs`, `O.ssprk33`, ... -- oracle/oracle_py.py, imported on first use only: building
    a case or running the HIP path never loads (or even imports) the oracle; the checker legs
    (run_oracle_rhs, the tolerance model) do."""

    def __getattr__(self, name):
        import oracle_py
        return getattr(oracle_py, name)


O = _LazyOracle()

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

def run_oracle_rhs(case: Case, nthreads: int = 1):
    return O.rhs(case.om, case.vl, case.ti, case.rhoe, case.T_aux, nthreads=nthreads)


# --------------------------------------------------------------- GPU runner

def _pkg():
    import __graft_entry__ as g
    return g.load_package()


class GpuModel:
    """A context on the HIP library configured from an OracleModel description,
    using only C-ABI calls (include/landhydro.h)."""

    def __init__(self, case: Case, math_mode: Optional[int] = None, stream=None):
        F = _pkg()._ffi
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
            self.upload(Ya, F.LH_VAR_T, c.T_aux)
        return Y, Ya

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

def _percol(case: Case, key, scalar):
    v = case.om.percol.get(key)
    return np.full(case.ncols, scalar) if v is None else np.asarray(v, dtype=np.float64)


def closure_tolerances(case: Case, diag, Cw: float):
    """Absolute tolerances for K, psi, T, kappa between two correct
    implementations in precision eps(FT): C*eps propagated through the
    cancelling sub-expressions of the closures."""
    om = case.om
    eps = float(np.finfo(case.dtype).eps) * Cw
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    K, psi, T, kap = f8(diag["K"]), f8(diag["psi"]), f8(diag["T"]), f8(diag["kappa"])
    vl, ti = f8(case.vl), f8(case.ti)
    nu = _percol(case, "nu", om.soil.nu)[:, None]
    thr = _percol(case, "vg_theta_r", om.vg.theta_r)[:, None]
    n = _percol(case, "vg_n", om.vg.n)[:, None]
    Ksat = _percol(case, "vg_Ksat", om.vg.Ksat)[:, None]
    m = 1.0 - 1.0 / n
    tiny = float(np.finfo(case.dtype).eps)
    vls = np.maximum(vl, thr + tiny)
    S = (vls - thr) / (nu - thr)
    Se = (vls - thr) / (nu - ti - thr)
    with np.errstate(all="ignore"):
        t = np.where(S < 1, S ** (1.0 / m), 1.0)
        inner = np.where(S < 1, 1.0 - (1.0 - t) ** m, 1.0)
        condK = 1.0 + 2.0 / np.maximum(inner, 1e-300) + np.abs(np.log(np.maximum(S, 1e-300))) / m
        # the inner cancellation w = 1 - S^(1/m): an eps-rounding of t = S^(1/m) (which the
        # reference's own correctly rounded pow has) moves w by eps t / w relatively, and
        # K ~ (1 - w^m)^2 by 2 m w^m / inner of that -- large next to saturation (t -> 1, w -> 0)
        w = np.where(S < 1, 1.0 - t, 1.0)
        condK = condK + np.where(S < 1, 2.0 * m * t * np.maximum(w, 1e-300) ** (m - 1.0)
                                 / np.maximum(inner, 1e-300), 0.0)
        # and S itself is computed (a difference and a quotient: one to two roundings in any
        # implementation): the same chain amplifies ITS relative error by another 1/m
        # (d ln t / d ln S), which matters for small m (clay-like n < 1.3) next to saturation
        condK = condK + np.where(S < 1, 2.0 * t * np.maximum(w, 1e-300) ** (m - 1.0)
                                 / np.maximum(inner, 1e-300), 0.0)
        u = np.where(Se < 1, Se ** (-1.0 / m) - 1.0, 0.0)
        condpsi = np.where(Se < 1, 1.0 + (u + 1.0) / (n * np.maximum(u, 1e-300)) +
                           np.abs(np.log(np.maximum(Se, 1e-300))) / (m * n), 2.0)
        # Se itself is a computed quantity (a difference and a quotient, or their logarithms: one
        # to two roundings in any implementation), and psi amplifies its relative error by
        # |d ln psi / d ln Se| = (u + 1) / (u n m) -- 1/m times the term above, which only covers a
        # rounding of S^(-1/m); it matters for small m (clay-like n < 1.3) next to saturation
        condpsi = condpsi + np.where(Se < 1, (u + 1.0) / (n * m * np.maximum(u, 1e-300)), 0.0)
    # conductivity factors: exp(gamma (T - T_ref)) and 10^(-Omega f_i) are powers too
    cf = om.cf
    if cf.viscosity_kind:
        condK = condK + np.abs(cf.gamma * (T - cf.T_ref))
    if cf.impedance_kind:
        with np.errstate(all="ignore"):
            tl = np.minimum(vl, nu - ti)
            f_i = np.where(tl + ti != 0, ti / (tl + ti), 0.0)
        condK = condK + cf.Omega * np.abs(f_i) * np.log(10.0)
    absK = eps * np.abs(K) * np.minimum(condK, 1e12) + 1e-300
    # psi near Se -> 1 from below is O(eps^(1/n)); give it that floor
    alpha = _percol(case, "vg_alpha", om.vg.alpha)[:, None]
    floor = (eps * 4) ** (1.0 / n) / alpha * (np.abs(Se - 1.0) < 4 * eps)
    abspsi = eps * np.abs(psi) * np.minimum(condpsi, 1e12) + floor + 1e-300
    absT = eps * (np.abs(T) + np.abs(T - om.earth.T_0)) * 2.0
    abskap = eps * np.abs(kap) * 8.0
    return dict(K=absK, psi=abspsi, T=absT, kappa=abskap)


def tendency_tolerance(case: Case, Cw: float = 16.0):
    """Per-cell absolute tolerance for each tendency (dict like the outputs)."""
    om = case.om
    diag = O.diagnostics(om, case.vl, case.ti, case.rhoe, case.T_aux)
    tol = closure_tolerances(case, diag, Cw)
    eps = float(np.finfo(case.dtype).eps) * Cw
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    K, psi, T, kap = f8(diag["K"]), f8(diag["psi"]), f8(diag["T"]), f8(diag["kappa"])
    n = om.nlev
    dz = (om.zmax - om.zmin) / n
    zc, _ = O.grid(om.zmin, om.zmax, n)
    zc = zc[None, :]
    water = om.model != M.MODEL_HEAT
    heat = om.model != M.MODEL_RICHARDS
    e = om.earth
    rhocp_l = e.cp_l * e.rho_liq
    E = rhocp_l * (T - e.T_0) * K
    absE = rhocp_l * (tol["T"] * np.abs(K) + np.abs(T - e.T_0) * tol["K"])
    abs_h = tol["psi"] + eps * np.abs(zc)
    N = case.ncols
    dFw = np.zeros((N, n + 1))
    dFe = np.zeros((N, n + 1))
    if n > 1:
        g = (psi[:, 1:] - psi[:, :-1] + (zc[:, 1:] - zc[:, :-1])) / dz
        Kb = 0.5 * (K[:, 1:] + K[:, :-1])
        if water:
            dFw[:, 1:n] = (0.5 * (tol["K"][:, 1:] + tol["K"][:, :-1]) * np.abs(g)
                           + Kb * (abs_h[:, 1:] + abs_h[:, :-1]) / dz + eps * np.abs(Kb * g))
        if heat:
            gT = (T[:, 1:] - T[:, :-1]) / dz
            kb = 0.5 * (kap[:, 1:] + kap[:, :-1])
            dFe[:, 1:n] = (0.5 * (tol["kappa"][:, 1:] + tol["kappa"][:, :-1]) * np.abs(gT)
                           + kb * (tol["T"][:, 1:] + tol["T"][:, :-1]) / dz + eps * np.abs(kb * gT))
            if water:
                Eb = 0.5 * (E[:, 1:] + E[:, :-1])
                dFe[:, 1:n] += (0.5 * (absE[:, 1:] + absE[:, :-1]) * np.abs(g)
                                + np.abs(Eb) * (abs_h[:, 1:] + abs_h[:, :-1]) / dz
                                + eps * np.abs(Eb * g))
    # boundary faces: evaluate the face closures with the oracle
    for face, ci, k in ((M.FACE_BOTTOM, 0, 0), (M.FACE_TOP, n - 1, n)):
        kh, vh = om.bc.get((face, M.COMP_HYDROLOGY), (M.BC_NONE, 0.0))
        ke, ve = om.bc.get((face, M.COMP_ENERGY), (M.BC_NONE, 0.0))
        pch = om.percol_bc.get((face, M.COMP_HYDROLOGY))
        vh = np.asarray(pch if pch is not None else vh, dtype=np.float64) + np.zeros(N)
        dzb = dz / 2
        vl_f = case.vl.copy()
        re_f = None if case.rhoe is None else case.rhoe.copy()
        if water and kh == M.BC_DIRICHLET:
            vl_f[:, ci] = vh.astype(case.dtype)
        fc = Case("face", om, case.dtype, N, vl=vl_f, ti=case.ti, rhoe=re_f, T_aux=case.T_aux)
        dgf = O.diagnostics(om, vl_f, case.ti, case.rhoe, case.T_aux)
        tf = closure_tolerances(fc, dgf, Cw)
        Kf, psif, kapf = f8(dgf["K"])[:, ci], f8(dgf["psi"])[:, ci], f8(dgf["kappa"])[:, ci]
        if water:
            if kh == M.BC_FLUX:
                dFw[:, k] = eps * np.abs(vh)
            elif kh == M.BC_FREE_DRAINAGE:
                dFw[:, k] = tol["K"][:, ci]
            elif kh == M.BC_DIRICHLET:
                gb = (psif - psi[:, ci] + dzb) / dzb
                gb2 = (psif - psi[:, ci] - dzb) / dzb
                gmax = np.maximum(np.abs(gb), np.abs(gb2))
                dFw[:, k] = (tf["K"][:, ci] * gmax + Kf * (tf["psi"][:, ci] + tol["psi"][:, ci]) / dzb
                             + eps * Kf * gmax)
        if heat:
            if ke == M.BC_FLUX:
                dFe[:, k] = eps * abs(ve)
            elif ke == M.BC_DIRICHLET:
                gTb = (ve - T[:, ci]) / dzb
                dFe[:, k] = (tf["kappa"][:, ci] * np.abs(gTb) + kapf * (tol["T"][:, ci] + eps * abs(ve)) / dzb
                             + eps * np.abs(kapf * gTb))
    out = {}
    if water:
        out["vl"] = (dFw[:, 1:] + dFw[:, :-1]) / dz
        out["ti"] = np.zeros((N, n))
    if heat:
        out["rhoe"] = (dFe[:, 1:] + dFe[:, :-1]) / dz
    return out


def assert_tendencies_close(case: Case, got, want, Cw: float = 16.0, label=""):
    tol = tendency_tolerance(case, Cw)
    for k in want:
        g = np.asarray(got[k], dtype=np.float64)
        w = np.asarray(want[k], dtype=np.float64)
        assert g.shape == w.shape, (k, g.shape, w.shape)
        assert np.all(np.isfinite(g)), f"{case.name}:{k} non-finite"
        if k == "ti":
            assert np.all(g == 0.0), "d theta_i must be identically zero"
            continue
        err = np.abs(g - w)
        allowed = tol[k] + Cw * float(np.finfo(case.dtype).eps) * np.abs(w)
        bad = err > allowed
        if bad.any():
            idx = np.unravel_index(np.argmax(err / allowed), err.shape)
            raise AssertionError(
                f"{case.name}{label}:{k}: {bad.sum()} of {bad.size} cells outside tolerance; worst at "
                f"{idx}: got {g[idx]:.17g} want {w[idx]:.17g} err {err[idx]:.3g} allowed "
                f"{allowed[idx]:.3g}")


def error_summary(case: Case, got, want, Cw: float = 16.0):
    """max(err/allowed) per tendency -- how much of the tolerance is used."""
    tol = tendency_tolerance(case, Cw)
    out = {}
    for k in want:
        if k == "ti":
            continue
        g = np.asarray(got[k], dtype=np.float64)
        w = np.asarray(want[k], dtype=np.float64)
        allowed = tol[k] + Cw * float(np.finfo(case.dtype).eps) * np.abs(w)
        out[k] = float(np.max(np.abs(g - w) / allowed))
    return out

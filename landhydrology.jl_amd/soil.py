"""Host-side mirror of the LandHydrology.jl SoilModel API for the hot path.

Julia is not available in the build image, so the host side above the C ABI is
written in Python with the reference's names, argument meaning and error
behaviour.  The Julia shim a maintainer would add is
`julia/LandHydrologyHIP.jl` (text only); both are thin: all
arithmetic happens in liblandhydro_hip.so on the GPU.

Reference files mirrored (under the reference tree):
  src/Domains/domain.jl                 Column, make_function_space
  src/SoilModel/parameters.jl           SoilParams
  src/SoilModel/SoilWaterParameterizations.jl  vanGenuchten, NoEffect, ...
  src/SoilModel/models.jl               SoilModel, SoilHydrologyModel, ...
  src/SoilModel/boundary_conditions.jl  NoBC, VerticalFlux, Dirichlet, FreeDrainage, ...
  src/SoilModel/initial_conditions.jl   initialize_states
  src/SoilModel/right_hand_side.jl      make_rhs, make_update_aux, coordinates
  src/Simulations/simulation.jl         Simulation, step!, run!

Build extension: `Column(..., ncolumns=N)` is an ensemble of N independent
columns (the reference has exactly one); every field is then [ncolumns, nelements].
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np

from . import _ffi as F
from .parameterizations import *  # noqa: F401,F403  (the exported SoilWater/SoilHeat helpers)

Float32, Float64 = np.float32, np.float64

# ------------------------------------------------------------------ Domains


class Column:
    """Column{FT}(zlim, nelements) -- src/Domains/domain.jl:12-33."""

    def __init__(self, FT=Float64, *, zlim, nelements, ncolumns: int = 1):
        self.FT = np.dtype(FT).type
        F.dtype_code(self.FT)
        zlim = (float(zlim[0]), float(zlim[1]))
        if not zlim[0] < zlim[1]:                       # @assert zlim[1] < zlim[2], domain.jl:30
            raise AssertionError("zlim[1] < zlim[2]")
        self.zlim = (self.FT(zlim[0]), self.FT(zlim[1]))
        self.nelements = int(nelements)
        self.ncolumns = int(ncolumns)
        if self.nelements < 1 or self.ncolumns < 1:
            raise ValueError("nelements and ncolumns must be >= 1")
        self.boundary_tags = ("bottom", "top")          # domain.jl:31

    def ndims(self):                                    # domain.jl:35
        return 1

    def length(self):                                   # domain.jl:37
        return self.zlim[1] - self.zlim[0]

    def size(self):                                     # domain.jl:39
        return self.length()

    def __repr__(self):                                 # Base.show, domain.jl:41-49
        return "[%0.1f, %0.1f]" % (self.zlim[0], self.zlim[1])


def make_function_space(domain: Column):
    """Centre and face z of the uniform column mesh (domain.jl:58-69).  The
    rounding of the centres is the library's (lh_coordinates)."""
    n = domain.nelements
    lo, hi = np.longdouble(domain.zlim[0]), np.longdouble(domain.zlim[1])
    zf = np.array([lo + (hi - lo) * k / n for k in range(n + 1)], dtype=np.longdouble)
    zf = zf.astype(domain.FT)
    zf[-1] = domain.zlim[1]
    zc = ((zf[:-1] + zf[1:]) / domain.FT(2)).astype(domain.FT)
    return zc, zf


# --------------------------------------------------------------- parameters


@dataclass
class EarthParameterSet:
    """The CLIMAParameters constants the soil model reads
    (SoilHeatParameterizations.jl:12-13), CLIMAParameters 0.1 values.  They are
    ABI inputs, never kernel literals."""
    rho_cloud_liq: float = 1e3
    rho_cloud_ice: float = 916.7
    cp_l: float = 4181.0
    cp_i: float = 2100.0
    T_0: float = 273.16
    LH_f0: float = 2.8344e6 - 2.5008e6   # LH_s0 - LH_v0
    K_therm: float = 2.4e-2
    # read by the prescribed-atmosphere BC only (boundary_conditions.jl:553-620 and the
    # SurfaceFluxes / Thermodynamics calls it makes): Planet.R_v, R_d, grav, cp_d, cp_v, LH_v0,
    # T_triple, press_triple; SubgridScale.von_karman_const
    R_v: float = 8.3144598 / 18.01528e-3
    R_d: float = 8.3144598 / 28.97e-3
    grav: float = 9.81
    cp_d: float = (8.3144598 / 28.97e-3) / (2.0 / 7.0)
    cp_v: float = 1859.0
    LH_v0: float = 2.5008e6
    T_triple: float = 273.16
    press_triple: float = 611.657
    von_karman_const: float = 0.4


_SOIL_DEFAULTS = dict(nu=0.43, S_s=1e-3, nu_ss_gravel=0.0, nu_ss_om=0.0, nu_ss_quartz=0.41,
                      rho_c_ds=2700.0, kappa_solid=3.97, rho_p=2700.0, kappa_sat_unfrozen=1.72,
                      kappa_sat_frozen=3.13, a=0.24, b=18.1, kappa_dry_parameter=0.053,
                      z_0m=0.001, z_0s=0.001)
# the reference's field names (parameters.jl:11-43) -> ASCII
_SOIL_ALIASES = {"ν": "nu", "ν_ss_gravel": "nu_ss_gravel", "ν_ss_om": "nu_ss_om",
                 "ν_ss_quartz": "nu_ss_quartz", "ρc_ds": "rho_c_ds", "κ_solid": "kappa_solid",
                 "ρp": "rho_p", "κ_sat_unfrozen": "kappa_sat_unfrozen",
                 "κ_sat_frozen": "kappa_sat_frozen", "κ_dry_parameter": "kappa_dry_parameter"}


class SoilParams:
    """SoilParams{FT}(; ν, S_s, ...) -- parameters.jl:11-43 (loam defaults).
    `nu` and `S_s` may be arrays of ncolumns values (per-column soils)."""

    def __init__(self, FT=Float64, **kw):
        self.FT = np.dtype(FT).type
        vals = dict(_SOIL_DEFAULTS)
        for k, v in kw.items():
            k = _SOIL_ALIASES.get(k, k)
            if k not in vals:
                raise TypeError(f"SoilParams has no field {k!r}")
            vals[k] = v
        self.__dict__.update(vals)

    # reference spellings
    ν = property(lambda s: s.nu)
    ρc_ds = property(lambda s: s.rho_c_ds)


class vanGenuchten:
    """vanGenuchten{FT}(; n, α, Ksat, θr) -- SoilWaterParameterizations.jl:150-169.
    Any field may be an array of ncolumns values (BASELINE config 5)."""

    def __init__(self, FT=Float64, *, n=1.56, alpha=None, Ksat=2.9e-7, theta_r=None, **kw):
        self.FT = np.dtype(FT).type
        alpha = kw.pop("α", alpha)
        theta_r = kw.pop("θr", theta_r)
        if kw:
            raise TypeError(f"vanGenuchten has no field(s) {sorted(kw)}")
        self.n = n
        self.alpha = 3.6 if alpha is None else alpha
        self.theta_r = 0.0 if theta_r is None else theta_r
        self.Ksat = Ksat

    @property
    def m(self):
        n = np.asarray(self.n, dtype=self.FT)
        return self.FT(1) - self.FT(1) / n

    α = property(lambda s: s.alpha)
    θr = property(lambda s: s.theta_r)


class NoEffect:
    """SoilWaterParameterizations.jl:38"""

    def __init__(self, FT=Float64):
        self.FT = FT


class TemperatureDependentViscosity:
    """SoilWaterParameterizations.jl:46-52"""

    def __init__(self, FT=Float64, *, gamma=2.64e-2, T_ref=288.0, **kw):
        self.FT = FT
        self.gamma = kw.pop("γ", gamma)
        self.T_ref = T_ref
        if kw:
            raise TypeError(f"unknown field(s) {sorted(kw)}")


class IceImpedance:
    """SoilWaterParameterizations.jl:62-65"""

    def __init__(self, FT=Float64, *, Omega=7.0, **kw):
        self.FT = FT
        self.Omega = kw.pop("Ω", Omega)
        if kw:
            raise TypeError(f"unknown field(s) {sorted(kw)}")


# ------------------------------------------------------------ component models


class SoilEnergyModel:
    """models.jl:17"""


class SoilHydrologyModel:
    """models.jl:28-33"""

    def __init__(self, FT=Float64, *, hydraulic_model=None, viscosity_factor=None,
                 impedance_factor=None):
        self.FT = np.dtype(FT).type
        self.hydraulic_model = hydraulic_model if hydraulic_model is not None else vanGenuchten(FT)
        self.viscosity_factor = viscosity_factor if viscosity_factor is not None else NoEffect(FT)
        self.impedance_factor = impedance_factor if impedance_factor is not None else NoEffect(FT)
        if not isinstance(self.viscosity_factor, (NoEffect, TemperatureDependentViscosity)):
            raise TypeError("viscosity_factor must be NoEffect or TemperatureDependentViscosity")
        if not isinstance(self.impedance_factor, (NoEffect, IceImpedance)):
            raise TypeError("impedance_factor must be NoEffect or IceImpedance")


class PrescribedTemperatureModel:
    """models.jl:51-54; default profile 288 K."""

    def __init__(self, T_profile: Optional[Callable] = None):
        self.is_default = T_profile is None
        self.T_profile = T_profile if T_profile is not None else (lambda z, t: 288.0 + 0.0 * z)


class PrescribedHydrologyModel:
    """models.jl:73-78; defaults: totally dry soil."""

    def __init__(self, vartheta_l_profile: Optional[Callable] = None,
                 theta_i_profile: Optional[Callable] = None, **kw):
        vartheta_l_profile = kw.pop("ϑ_l_profile", kw.pop("θ_l_profile", vartheta_l_profile))
        theta_i_profile = kw.pop("θ_i_profile", theta_i_profile)
        if kw:
            raise TypeError(f"unknown field(s) {sorted(kw)}")
        self.vartheta_l_profile = vartheta_l_profile or (lambda z, t: 0.0 * z)
        self.theta_i_profile = theta_i_profile or (lambda z, t: 0.0 * z)


# --------------------------------------------------------- boundary conditions


class NoBC:
    """boundary_conditions.jl:27"""


class VerticalFlux:
    """boundary_conditions.jl:43-46; positive = aligned with z-hat.  A scalar, or
    an array of ncolumns values."""

    def __init__(self, flux):
        self.flux = flux


class Dirichlet:
    """boundary_conditions.jl:61-64; state_value is t -> value (scalar or
    per-column array); a number is accepted as a constant."""

    def __init__(self, state_value):
        self.state_value = state_value if callable(state_value) else (lambda t, v=state_value: v)


class FreeDrainage:
    """boundary_conditions.jl:77"""


class SoilComponentBC:
    """boundary_conditions.jl:95-101"""

    def __init__(self, *, energy=None, hydrology=None):
        self.energy = energy if energy is not None else NoBC()
        self.hydrology = hydrology if hydrology is not None else NoBC()


class PrescribedAtmosForcing:
    """PrescribedAtmosForcing{FT}(; u_atm, θ_atm, z_atm, θ_scale, ρ_a_sfc, q_atm)
    (boundary_conditions.jl:119-132): the atmospheric state that drives the TOP face of a coupled
    water + heat soil model through Monin-Obukhov surface fluxes (:553-620), evaluated on the
    device from the top cell's state.  `u_atm`, `θ_atm`, `q_atm` may be arrays of ncolumns values
    (build extension).  Parity beyond the reference's equilibrium invariant is unpinned: the
    SurfaceFluxes / Thermodynamics formulas are restated from their published forms."""

    _FIELDS = ("u_atm", "theta_atm", "z_atm", "theta_scale", "rho_a_sfc", "q_atm")
    _ALIASES = {"θ_atm": "theta_atm", "θ_scale": "theta_scale", "ρ_a_sfc": "rho_a_sfc"}

    def __init__(self, FT=Float64, **kw):
        self.FT = np.dtype(FT).type
        vals = {}
        for k, v in kw.items():
            k = self._ALIASES.get(k, k)
            if k not in self._FIELDS:
                raise TypeError(f"PrescribedAtmosForcing has no field {k!r}")
            vals[k] = v
        missing = [k for k in self._FIELDS if k not in vals]
        if missing:                                  # Base.@kwdef without defaults: UndefKeywordError
            raise TypeError(f"PrescribedAtmosForcing: keyword argument(s) {missing} not assigned")
        self.__dict__.update(vals)

    θ_atm = property(lambda s: s.theta_atm)
    θ_scale = property(lambda s: s.theta_scale)
    ρ_a_sfc = property(lambda s: s.rho_a_sfc)


class SoilColumnBC:
    """boundary_conditions.jl:144-161"""

    def __init__(self, *, top=None, bottom=None):
        self.top = top if top is not None else SoilComponentBC()
        self.bottom = bottom if bottom is not None else SoilComponentBC()
        # SoilColumnBC{TBC <: Union{SoilComponentBC, PrescribedAtmosForcing}, BBC <: SoilComponentBC}
        if not isinstance(self.bottom, SoilComponentBC):
            raise TypeError("bottom must be a SoilComponentBC (a prescribed atmosphere is only valid at the top)")
        if not isinstance(self.top, (SoilComponentBC, PrescribedAtmosForcing)):
            raise TypeError("top must be a SoilComponentBC or a PrescribedAtmosForcing")


_BC_KIND = {NoBC: F.LH_BC_NONE, VerticalFlux: F.LH_BC_FLUX, Dirichlet: F.LH_BC_DIRICHLET,
            FreeDrainage: F.LH_BC_FREE_DRAINAGE}

# ------------------------------------------------------------------- states

_VAR = {"ϑ_l": 0, "θ_l": 0, "vartheta_l": 0, "θ_i": 1, "theta_i": 1, "ρe_int": 2, "rhoe_int": 2,
        "T": 3}
_NAMES = ["ϑ_l", "θ_i", "ρe_int", "T"]


class _Namespace:
    """`Y.soil` -- attribute access downloads the field as [ncolumns, nelements]."""

    def __init__(self, fv):
        object.__setattr__(self, "_fv", fv)

    def __getattr__(self, name):
        if name not in _VAR:
            raise AttributeError(name)
        return self._fv.get(name)

    def __setattr__(self, name, value):
        self._fv.set(name, value)


class FieldVector:
    """Device-resident analogue of Fields.FieldVector(; soil = (...)): one
    lh_state.  `fv.soil.ϑ_l` downloads, `fv.soil.ϑ_l = a` uploads (host arrays are
    level-fastest [ncolumns, nelements] like parent(field); a 1-D array of
    nelements is broadcast over columns)."""

    def __init__(self, model: "SoilModel", mask: int, zc=None):
        self.model = model
        self._be = model._backend()
        self.mask = mask
        h = C.c_void_p()
        F.check(F.lib().lh_state_create(self._be.ctx, mask, C.byref(h)), self._be.ctx)
        self.handle = h
        self.zc = zc
        self.soil = _Namespace(self)
        setattr(self, model.name, self.soil)

    def __del__(self):
        try:
            if self.handle and self._be.ctx:
                F.lib().lh_state_destroy(self._be.ctx, self.handle)
        except Exception:
            pass

    @property
    def names(self):
        return [n for i, n in enumerate(_NAMES) if self.mask & (1 << i)]

    def get(self, name) -> np.ndarray:
        d = self.model.domain
        out = np.empty((d.ncolumns, d.nelements), dtype=d.FT)
        F.check(F.lib().lh_download(self._be.ctx, self.handle, _VAR[name], out.ctypes.data, 1,
                                    d.nelements), self._be.ctx)
        return out

    def get_level(self, name, level) -> np.ndarray:
        """One level of one variable for every column ([ncolumns]; level -1 = the top
        cell): interior_values(X, :top, cs) of a host-evaluated boundary condition."""
        d = self.model.domain
        lev = int(level) % d.nelements if -d.nelements <= int(level) < d.nelements else int(level)
        out = np.empty(d.ncolumns, dtype=d.FT)
        F.check(F.lib().lh_download_level(self._be.ctx, self.handle, _VAR[name], lev, out.ctypes.data),
                self._be.ctx)
        return out

    def set(self, name, value):
        d = self.model.domain
        a = np.asarray(value, dtype=d.FT)
        if a.ndim == 0 or not a.any():
            # a constant (or all-zero) field is a fill; the library then knows a zero plane is
            # zero: a launch neither reads a zero θ_i plane nor re-stores dθ_i = 0
            F.check(F.lib().lh_state_fill(self._be.ctx, self.handle, _VAR[name],
                                          float(a) if a.ndim == 0 else 0.0), self._be.ctx)
            return
        if a.ndim == 1:
            # one value per level, the same in every column (what the reference's profile closures
            # and an initial condition f(z) give): nelements numbers cross PCIe, not a plane
            if a.shape != (d.nelements,):
                raise ValueError(f"expected shape {(d.nelements,)} for a per-level profile, got {a.shape}")
            a = np.ascontiguousarray(a)
            F.check(F.lib().lh_upload_profile(self._be.ctx, self.handle, _VAR[name], a.ctypes.data), self._be.ctx)
            return
        if a.shape != (d.ncolumns, d.nelements):
            raise ValueError(f"expected shape {(d.ncolumns, d.nelements)}, got {a.shape}")
        a = np.ascontiguousarray(a)
        F.check(F.lib().lh_upload(self._be.ctx, self.handle, _VAR[name], a.ctypes.data, 1,
                                  d.nelements), self._be.ctx)

    def similar(self) -> "FieldVector":
        return FieldVector(self.model, self.mask, self.zc)

    def copy(self) -> "FieldVector":
        o = self.similar()
        F.check(F.lib().lh_state_copy(self._be.ctx, o.handle, self.handle), self._be.ctx)
        return o

    def device_ptr(self, name):
        """Device pointer and element strides of one plane.  Until release_ptr the library assumes
        nothing about that plane (the holder may write through the pointer at any time)."""
        p, ls, cs = C.c_void_p(), C.c_int64(), C.c_int64()
        F.check(F.lib().lh_state_device_ptr(self._be.ctx, self.handle, _VAR[name], C.byref(p),
                                            C.byref(ls), C.byref(cs)), self._be.ctx)
        return p.value, ls.value, cs.value

    def release_ptr(self, name=None):
        """The holder of device_ptr pointers promises not to write through them any more
        (name None: every plane of this state)."""
        F.check(F.lib().lh_state_release_ptr(self._be.ctx, self.handle, -1 if name is None else _VAR[name]),
                self._be.ctx)


# -------------------------------------------------------------------- model


class _Backend:
    """One lh_ctx: the device-side image of a SoilModel."""

    def device_index(self) -> int:
        return self._device

    def __init__(self, model: "SoilModel", stream=None, device=-1):
        d = model.domain
        L = F.lib()
        em, hm = model.energy_model, model.hydrology_model
        if isinstance(em, SoilEnergyModel) and isinstance(hm, SoilHydrologyModel):
            kind = F.LH_MODEL_COUPLED
        elif isinstance(em, SoilEnergyModel) and isinstance(hm, PrescribedHydrologyModel):
            kind = F.LH_MODEL_HEAT
        elif isinstance(em, PrescribedTemperatureModel) and isinstance(hm, SoilHydrologyModel):
            kind = F.LH_MODEL_RICHARDS
        else:
            kind = None   # prescribed/prescribed: empty RHS (right_hand_side.jl:103-112)
        self.kind = kind
        self.ctx = None
        self._keep = []
        self._device = int(device)
        if kind is None:
            return
        if self._device < 0:   # the current device, by number (buffers handed to the library live there)
            import torch
            self._device = int(torch.cuda.current_device()) if torch.cuda.is_available() else 0
            device = self._device
        cfg = F.lh_config(d.ncolumns, d.nelements, F.dtype_code(d.FT), float(d.zlim[0]),
                          float(d.zlim[1]), kind, device, stream)
        ctx = C.c_void_p()
        F.check(L.lh_create(C.byref(ctx), C.byref(cfg)), None)
        self.ctx = ctx
        ep = model.earth_param_set
        if ep is not None and kind != F.LH_MODEL_RICHARDS:
            e = F.lh_earth_params(ep.rho_cloud_liq, ep.rho_cloud_ice, ep.cp_l, ep.cp_i, ep.T_0,
                                  ep.LH_f0, ep.K_therm)
            F.check(L.lh_set_earth_params(ctx, C.byref(e)), ctx)
        sp = model.soil_param_set
        scal = {}
        for k in F.lh_soil_params._fields_:
            v = getattr(sp, k[0])
            scal[k[0]] = self._scalar_or_percol(k[0], v, d)
        s = F.lh_soil_params(**scal)
        F.check(L.lh_set_soil_params(ctx, C.byref(s)), ctx)
        if isinstance(hm, SoilHydrologyModel):
            vg = hm.hydraulic_model
            v = F.lh_vg_params(self._scalar_or_percol("vg_n", vg.n, d),
                               self._scalar_or_percol("vg_alpha", vg.alpha, d),
                               self._scalar_or_percol("vg_theta_r", vg.theta_r, d),
                               self._scalar_or_percol("vg_Ksat", vg.Ksat, d))
            F.check(L.lh_set_vg_params(ctx, C.byref(v)), ctx)
            vf, imf = hm.viscosity_factor, hm.impedance_factor
            vk = isinstance(vf, TemperatureDependentViscosity)
            ik = isinstance(imf, IceImpedance)
            F.check(L.lh_set_conductivity_factors(
                ctx, int(vk), vf.gamma if vk else 2.64e-2, vf.T_ref if vk else 288.0, int(ik),
                imf.Omega if ik else 7.0), ctx)
        self.time_dependent_bc = False
        self.set_bcs(model, 0.0)

    def _scalar_or_percol(self, key, v, d):
        a = np.asarray(v, dtype=np.float64)
        if a.ndim == 0:
            return float(a)
        if key not in F.LH_PC:
            raise ValueError(f"{key} cannot vary per column")
        if a.shape != (d.ncolumns,):
            raise ValueError(f"{key}: expected {d.ncolumns} per-column values, got shape {a.shape}")
        a = np.ascontiguousarray(a)
        F.check(F.lib().lh_set_percol_param(self.ctx, F.LH_PC[key],
                                            a.ctypes.data_as(C.POINTER(C.c_double))), self.ctx)
        return float(a[0])

    def bc_values(self, model, t):
        """Evaluate the BC closures at time t -> {(face, comp): (kind, value)}."""
        out = {}
        for face, tag in ((F.LH_FACE_BOTTOM, "bottom"), (F.LH_FACE_TOP, "top")):
            fbc = getattr(model.boundary_conditions, tag)
            if isinstance(fbc, PrescribedAtmosForcing):      # set through lh_set_atmos_forcing
                continue
            for comp, cname in ((F.LH_COMP_ENERGY, "energy"), (F.LH_COMP_HYDROLOGY, "hydrology")):
                bc = getattr(fbc, cname)
                kind = _BC_KIND.get(type(bc))
                if kind is None:
                    raise TypeError(f"unsupported boundary condition {type(bc).__name__}")
                val = 0.0
                if kind == F.LH_BC_FLUX:
                    val = bc.flux
                elif kind == F.LH_BC_DIRICHLET:
                    val = bc.state_value(t)
                out[(face, comp)] = (kind, val)
        return out

    def set_atmos(self, model):
        """lh_set_atmos_forcing from SoilColumnBC.top (or its removal)."""
        L = F.lib()
        d = model.domain
        top = model.boundary_conditions.top
        if not isinstance(top, PrescribedAtmosForcing):
            if getattr(self, "_atmos_set", False):
                F.check(L.lh_set_atmos_forcing(self.ctx, None, None), self.ctx)
                self._atmos_set = False
            return
        ep, sp = model.earth_param_set, model.soil_param_set
        scal, percol = {}, np.zeros((3, d.ncolumns))
        any_pc = False
        for k, name in enumerate(("u_atm", "theta_atm", "q_atm")):
            a = np.asarray(getattr(top, name), dtype=np.float64)
            if a.ndim == 0:
                scal[name] = float(a)
                percol[k] = float(a)
            else:
                if a.shape != (d.ncolumns,):
                    raise ValueError(f"{name}: expected {d.ncolumns} per-column values")
                scal[name] = float(a[0])
                percol[k] = a
                any_pc = True
        f = F.lh_atmos_forcing(scal["u_atm"], scal["theta_atm"], float(top.z_atm), float(top.theta_scale),
                               float(top.rho_a_sfc), scal["q_atm"], float(sp.z_0m), float(sp.z_0s),
                               ep.R_v, ep.R_d, ep.grav, ep.cp_d, ep.cp_v, ep.LH_v0, ep.T_triple,
                               ep.press_triple, ep.von_karman_const)
        percol = np.ascontiguousarray(percol)
        F.check(L.lh_set_atmos_forcing(self.ctx, C.byref(f),
                                       percol.ctypes.data_as(C.POINTER(C.c_double)) if any_pc else None),
                self.ctx)
        self._atmos_set = True

    def set_bcs(self, model, t):
        if model.boundary_conditions is None:
            return
        L = F.lib()
        d = model.domain
        self.set_atmos(model)
        for (face, comp), (kind, val) in self.bc_values(model, t).items():
            a = np.asarray(val, dtype=np.float64)
            if a.ndim == 0:
                F.check(L.lh_set_bc(self.ctx, face, comp, kind, float(a), None), self.ctx)
            else:
                if a.shape != (d.ncolumns,):
                    raise ValueError("per-column boundary values must have ncolumns entries")
                a = np.ascontiguousarray(a)
                F.check(L.lh_set_bc(self.ctx, face, comp, kind, float(a[0]),
                                    a.ctypes.data_as(C.POINTER(C.c_double))), self.ctx)

    def close(self):
        if self.ctx:
            F.lib().lh_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SoilModel:
    """SoilModel(FT; domain, energy_model, hydrology_model, boundary_conditions,
    soil_param_set, earth_param_set, name) -- models.jl:90-135."""

    def __init__(self, FT=Float64, *, domain, energy_model, hydrology_model, boundary_conditions,
                 soil_param_set=None, earth_param_set, name="soil", stream=None, device=-1):
        self.FT = np.dtype(FT).type
        if np.dtype(domain.FT) != np.dtype(self.FT):
            raise TypeError("domain FT differs from the model FT")   # AbstractVerticalDomain{FT}
        self.domain = domain
        self.energy_model = energy_model
        self.hydrology_model = hydrology_model
        self.boundary_conditions = boundary_conditions
        self.soil_param_set = soil_param_set if soil_param_set is not None else SoilParams(FT)
        self.earth_param_set = earth_param_set
        self.name = name
        self._stream, self._device = stream, device
        self._be = None

    def _backend(self) -> _Backend:
        if self._be is None:
            self._be = _Backend(self, self._stream, self._device)
        return self._be

    def close(self):
        if self._be is not None:
            self._be.close()


def coordinates(model: SoilModel) -> np.ndarray:
    """coordinates(cs) (right_hand_side.jl:7-8): cell-centre z as the library
    holds them."""
    be = model._backend()
    n = model.domain.nelements
    if be.ctx is None:
        return make_function_space(model.domain)[0]
    z = np.empty(n, dtype=np.float64)
    F.check(F.lib().lh_coordinates(be.ctx, z.ctypes.data_as(C.POINTER(C.c_double))), be.ctx)
    return z.astype(model.domain.FT)


def _prognostic_mask(kind):
    return {F.LH_MODEL_RICHARDS: 0b0011, F.LH_MODEL_HEAT: 0b0100, F.LH_MODEL_COUPLED: 0b0111}[kind]


def _aux_mask(kind, model):
    """Planes of Ya the DEVICE reads: (ϑ_l, θ_i) for a prescribed hydrology; T for
    a prescribed temperature only when a viscosity factor consumes it
    (right_hand_side.jl:160).  Otherwise Ya stays a host object."""
    if kind == F.LH_MODEL_HEAT:
        return 0b0011
    if kind == F.LH_MODEL_RICHARDS and isinstance(model.hydrology_model.viscosity_factor,
                                                  TemperatureDependentViscosity):
        return 0b1000
    return 0


class _HostAux:
    """Ya for models whose aux fields never reach the device (or prescribed /
    prescribed models): plain host arrays with the same attribute access."""

    def __init__(self, zc, fields):
        self.zc = zc
        self.soil = type("soil", (), {})()
        for k, v in fields.items():
            setattr(self.soil, k, v)


def make_update_aux(component_model, model: Optional[SoilModel] = None):
    """make_update_aux (right_hand_side.jl:54-96): returns update_aux!(Ya, t),
    which overwrites the prescribed fields of Ya with the closures at time t."""
    if isinstance(component_model, PrescribedTemperatureModel):
        def update_aux(Ya, t):
            z = np.asarray(Ya.zc)
            Ya.soil.T = np.asarray(component_model.T_profile(z, t) + 0.0 * z)
            return Ya
        return update_aux
    if isinstance(component_model, PrescribedHydrologyModel):
        def update_aux(Ya, t):
            z = np.asarray(Ya.zc)
            Ya.soil.θ_l = np.asarray(component_model.vartheta_l_profile(z, t) + 0.0 * z)
            Ya.soil.θ_i = np.asarray(component_model.theta_i_profile(z, t) + 0.0 * z)
            return Ya
        return update_aux

    def update_aux(Ya, t):      # models that add no auxiliary variables (:91-96)
        return None
    return update_aux


def initialize_states(model: SoilModel, f: Callable, t0):
    """initialize_states(model, f, t0) -> (Y, Ya) (initial_conditions.jl:101-107).
    `f(z, model)` returns a dict (NamedTuple) of the prognostic variables at the
    centre coordinate z; it is called once with the vector of nelements centres
    (scalars, [nelements] or [ncolumns, nelements] values are accepted), or per
    level if it cannot take arrays."""
    be = model._backend()
    d = model.domain
    zc = coordinates(model)
    if be.kind is None:
        fields = {}
        if isinstance(model.energy_model, PrescribedTemperatureModel):
            fields["T"] = np.asarray(model.energy_model.T_profile(zc, t0) + 0.0 * zc)
        if isinstance(model.hydrology_model, PrescribedHydrologyModel):
            fields["θ_l"] = np.asarray(model.hydrology_model.vartheta_l_profile(zc, t0) + 0 * zc)
            fields["θ_i"] = np.asarray(model.hydrology_model.theta_i_profile(zc, t0) + 0 * zc)
        return {}, _HostAux(zc, fields)
    try:
        ic = f(zc, model)
        ic = {k: np.asarray(v) for k, v in ic.items()}
    except (TypeError, ValueError):
        per = [f(d.FT(z), model) for z in zc]
        ic = {k: np.array([p[k] for p in per]) for k in per[0]}
    Y = FieldVector(model, _prognostic_mask(be.kind), zc)
    want = set(Y.names)
    got = {_NAMES[_VAR[k]] for k in ic}
    if got != want:
        raise KeyError(f"initial condition must provide {sorted(want)}, got {sorted(ic)}")
    for k, v in ic.items():
        if v.ndim == 0:
            v = np.full(d.nelements, v)
        Y.set(k, v)
    am = _aux_mask(be.kind, model)
    Ya = FieldVector(model, am, zc) if am else _HostAux(zc, {})
    make_update_aux(model.energy_model)(Ya, t0)
    make_update_aux(model.hydrology_model)(Ya, t0)
    return Y, Ya


def default_initial_conditions(model: SoilModel):
    """models.jl:147-166: isothermal soil at T0 = 273.16, no ice, ϑ_l = ν/2; only
    for SoilEnergyModel + SoilHydrologyModel."""
    if not (isinstance(model.energy_model, SoilEnergyModel)
            and isinstance(model.hydrology_model, SoilHydrologyModel)):
        raise RuntimeError("No default IC exist for this type of soil model.")
    FT = model.FT
    ep, sp = model.earth_param_set, model.soil_param_set

    def ic(z, m):
        T = FT(273.16)
        theta_i = FT(0.0)
        theta_l = FT(0.5) * FT(sp.nu)
        # volumetric_heat_capacity / volumetric_internal_energy with FT rounding
        rho_i, rho_l = FT(ep.rho_cloud_ice), FT(ep.rho_cloud_liq)
        rhocp_i = FT(ep.cp_i * float(rho_i))
        rhocp_l = FT(ep.cp_l * float(rho_l))
        rho_c_s = FT(sp.rho_c_ds) + theta_l * rhocp_l + theta_i * rhocp_i
        rhoe = rho_c_s * (T - FT(ep.T_0)) - theta_i * rho_i * FT(ep.LH_f0)
        return {"ϑ_l": theta_l, "θ_i": theta_i, "ρe_int": rhoe}

    return initialize_states(model, ic, FT(0.0))


# Placement tuning (lh_tune_placement) is explicit: call `tune_placement(model, Y, Ya, dY)` once
# per (Y, dY) pair of a large ensemble.  LH_PLACEMENT_TUNE=1 opts in to the host mirror doing it
# by itself on the first rhs! / step of ensembles whose planes reach this size -- never moving
# the input state (device pointers a user holds for Y stay valid).
PLACEMENT_TUNE_MIN_PLANE_BYTES = 32 << 20


def _placement_tuning_wanted(model: SoilModel) -> bool:
    d = model.domain
    if os.environ.get("LH_PLACEMENT_TUNE", "0") != "1":
        return False
    return d.ncolumns * d.nelements * np.dtype(d.FT).itemsize >= PLACEMENT_TUNE_MIN_PLANE_BYTES


def tune_placement(model: SoilModel, Y: "FieldVector", Ya=None, dY: Optional["FieldVector"] = None,
                   max_candidates: int = 0, move_input: bool = False):
    """Build extension (no counterpart in the reference): let the library choose, by
    timing the real launch, where in HBM the state written by rhs! (dY given) or the
    SSPRK33 stage state (dY=None) lives -- lh_tune_placement.  Returns the launch
    time in ms before and after.  Results of later calls do not depend on it."""
    be = model._backend()
    ya = Ya.handle if isinstance(Ya, FieldVector) else None
    be.set_bcs(model, 0.0)
    b, a = C.c_float(), C.c_float()
    F.check(F.lib().lh_tune_placement(be.ctx, Y.handle, ya, dY.handle if dY is not None else None,
                                      int(max_candidates), F.LH_PLACE_MOVE_INPUT if move_input else 0,
                                      C.byref(b), C.byref(a)), be.ctx)
    return b.value, a.value


def make_rhs(model: SoilModel):
    """make_rhs(model) -> rhs!(dY, Y, Ya, t) (right_hand_side.jl:33-44).  In place
    on dY, returns dY.  Everything numerical happens in lh_rhs on the GPU."""
    be = model._backend()
    if be.kind is None:
        def rhs_empty(dY, Y, Ya, t):   # right_hand_side.jl:103-112
            make_update_aux(model.energy_model)(Ya, t)
            make_update_aux(model.hydrology_model)(Ya, t)
            return dY
        return rhs_empty
    update_en = make_update_aux(model.energy_model)
    update_hy = make_update_aux(model.hydrology_model)
    L = F.lib()
    needs_aux = bool(_aux_mask(be.kind, model))
    placed = set() if _placement_tuning_wanted(model) else None   # (Y, dY) pairs already placed

    def rhs(dY, Y, Ya, t):
        if needs_aux:
            update_en(Ya, t)
            update_hy(Ya, t)
        be.set_bcs(model, t)
        ya = Ya.handle if isinstance(Ya, FieldVector) else None
        if placed is not None and (id(Y), id(dY)) not in placed:
            placed.add((id(Y), id(dY)))
            F.check(L.lh_tune_placement(be.ctx, Y.handle, ya, dY.handle, 0, 0, None, None), be.ctx)
        F.check(L.lh_rhs(be.ctx, float(t), Y.handle, ya, dY.handle), be.ctx)
        return dY

    return rhs


def compute_turbulent_surface_fluxes(energy, hydrology, model: SoilModel, vartheta_l, theta_i, T):
    """compute_turbulent_surface_fluxes(energy, hydrology, model, ϑ_l, θ_i, T)
    (boundary_conditions.jl:553-620) -> (heat_flux, Ẽ): the surface heat flux and water volume
    flux the prescribed atmosphere of `model.boundary_conditions.top` drives for a top-cell state
    (scalars or arrays), evaluated on the device (lh_atmos_surface_fluxes).  Only
    SoilEnergyModel + SoilHydrologyModel has a method (test_prescribed_atmos_bc.jl:161-183)."""
    if not (isinstance(energy, SoilEnergyModel) and isinstance(hydrology, SoilHydrologyModel)):
        raise TypeError("no method compute_turbulent_surface_fluxes for "
                        f"({type(energy).__name__}, {type(hydrology).__name__})")
    be = model._backend()
    be.set_bcs(model, 0.0)
    vl, ti, T = np.broadcast_arrays(*(np.asarray(a, dtype=np.float64) for a in (vartheta_l, theta_i, T)))
    shape = vl.shape
    vl, ti, T = (np.ascontiguousarray(a.reshape(-1)) for a in (vl, ti, T))
    heat, water = np.empty_like(vl), np.empty_like(vl)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    F.check(F.lib().lh_atmos_surface_fluxes(be.ctx, vl.size, dp(vl), dp(ti), dp(T), dp(heat), dp(water)), be.ctx)
    FT = model.FT
    if shape == ():
        return FT(heat[0]), FT(water[0])
    return heat.reshape(shape).astype(FT), water.reshape(shape).astype(FT)


def boundary_fluxes(X, bc, face, model: SoilModel, cs=None, t=0.0):
    """boundary_fluxes(X, bc, face, model, cs, t) (boundary_conditions.jl:470-489, :516-533): the named
    pair (fρe_int, fϑ_l) of one face -- the two SetValue fluxes of the tendency -- for every column.

    * `bc` a PrescribedAtmosForcing and `X = (ϑ_l, θ_i, T)` host values of the top cell: the
      surface fluxes of those states (compute_turbulent_surface_fluxes), as the reference's tests
      call it.
    * `bc` a SoilComponentBC (the model's own `boundary_conditions.top` / `.bottom`) and `X` the
      device state -- `Y`, or `(Y, Ya)` when the model reads prescribed fields: evaluated on the
      device by the function the tendency kernel uses (lh_boundary_fluxes), [ncolumns] each; a
      component without a boundary condition gives NaN (`nothing` in the reference)."""
    tag = face.lstrip(":") if isinstance(face, str) else face
    if tag not in ("top", "bottom"):
        raise ValueError("Expected :top or :bottom")                 # boundary_conditions.jl:188
    if isinstance(bc, PrescribedAtmosForcing) and not isinstance(X, (FieldVector, tuple)) or (
            isinstance(bc, PrescribedAtmosForcing) and isinstance(X, tuple) and not isinstance(X[0], FieldVector)):
        if tag != "top":
            raise RuntimeError("Prescribed atmosphere-driven boundary conditions are only valid at the "
                               "top of the soil column.")            # :523-528
        vl, ti, T = X
        h, w = compute_turbulent_surface_fluxes(model.energy_model, model.hydrology_model, model, vl, ti, T)
        return {"fρe_int": h, "fϑ_l": w}
    if isinstance(bc, PrescribedAtmosForcing) and tag != "top":
        raise RuntimeError("Prescribed atmosphere-driven boundary conditions are only valid at the "
                           "top of the soil column.")
    if bc is not getattr(model.boundary_conditions, tag):
        raise ValueError("boundary_fluxes on the device evaluates the model's own boundary condition of that face: "
                         f"pass model.boundary_conditions.{tag}")
    Y, Ya = X if isinstance(X, tuple) else (X, None)
    be = model._backend()
    be.set_bcs(model, t)
    d = model.domain
    fe, fw = np.empty(d.ncolumns), np.empty(d.ncolumns)
    ya = Ya.handle if isinstance(Ya, FieldVector) else None
    F.check(F.lib().lh_boundary_fluxes(be.ctx, Y.handle, ya, float(t), F.LH_FACE_TOP if tag == "top" else F.LH_FACE_BOTTOM,
                                       fe.ctypes.data_as(C.POINTER(C.c_double)), fw.ctypes.data_as(C.POINTER(C.c_double))),
            be.ctx)
    return {"fρe_int": fe.astype(d.FT), "fϑ_l": fw.astype(d.FT)}


def stable_dt(model: SoilModel, Y: "FieldVector", Ya=None, courant: float = 0.5) -> float:
    """Build extension (the reference steps with a fixed user dt): the largest
    step the explicit diffusive bound allows on this rank's columns,
    courant * dz^2 / max face diffusivity (lh_stable_dt).  Across ranks take the
    minimum with `partition.global_min_dt`."""
    be = model._backend()
    out = C.c_double()
    ya = Ya.handle if isinstance(Ya, FieldVector) else None
    F.check(F.lib().lh_stable_dt(be.ctx, Y.handle, ya, float(courant), C.byref(out)), be.ctx)
    return out.value


def step_adaptive(model: SoilModel, Y: "FieldVector", Ya=None, t: float = 0.0, courant: float = 0.5,
                  nsteps: int = 1, dt_max: float = 0.0, profiles_constant: bool = False):
    """Build extension: `nsteps` adaptive SSPRK33 steps of `Y` with nothing leaving the device
    (lh_step_ssprk33_adaptive: per step the tendency and the stable-step bound in one launch, the
    min over ranks when a communicator is attached, stages 2 and 3 -- three evaluations of the
    right-hand side per step).  Boundary values are those of time `t` (constant over the call: a
    time-dependent Dirichlet closure needs the stage times, i.e. `Simulation` with a fixed dt).
    The stage times of an adaptive step are not known in advance, so prescribed profiles the device
    reads (T with a viscosity factor, the water fields of a heat-only model) cannot be probed for
    time dependence: the caller states `profiles_constant=True` (they are then taken at time `t`).
    Returns (simulated time advanced, last dt)."""
    if _time_dependent(model):
        raise ValueError("step_adaptive needs boundary values that do not depend on time")
    if _device_reads_aux(model, Ya):
        if not profiles_constant:
            raise ValueError("step_adaptive: the device reads prescribed profiles whose time dependence cannot be "
                             "probed at unknown stage times; pass profiles_constant=True if they do not depend on t")
        make_update_aux(model.energy_model)(Ya, t)
        make_update_aux(model.hydrology_model)(Ya, t)
    be = model._backend()
    L = F.lib()
    ya = Ya.handle if isinstance(Ya, FieldVector) else None
    be.set_bcs(model, t)
    import torch
    dtype = torch.float64 if np.dtype(model.domain.FT) == np.float64 else torch.float32
    # [dt, elapsed] on the CONTEXT's device; the zero fill runs on torch's stream, the library on its
    # own (non-blocking) one: the fill must have completed before the first `*elapsed += dt`
    dev = torch.device("cuda", be.device_index())
    buf = torch.zeros(2, device=dev, dtype=dtype)
    torch.cuda.synchronize(dev)
    F.check(L.lh_step_ssprk33_adaptive(be.ctx, Y.handle, ya, float(t), float(courant), float(dt_max), int(nsteps),
                                       C.c_void_p(buf.data_ptr()),
                                       C.c_void_p(buf.data_ptr() + buf.element_size())), be.ctx)
    F.check(L.lh_synchronize(be.ctx), be.ctx)
    dt, elapsed = (float(x) for x in buf.cpu())
    return elapsed, dt


# --------------------------------------------------------------- Simulations


class SSPRK33:
    """OrdinaryDiffEq.SSPRK33 marker (the only stepper the reference's tests use)."""


class _Solution:
    def __init__(self):
        self.t = []
        self.u = []


class _Integrator:
    def __init__(self, model, Y, Ya, t0, tf, dt, saveat):
        self.model, self.u, self.p = model, Y, Ya
        self.t, self.tf, self.dt = float(t0), float(tf), float(dt)
        self.saveat = saveat
        self.sol = _Solution()
        self._nsteps_done = 0
        self._save()

    def _save(self):
        self.sol.t.append(self.t)
        self.sol.u.append({n: self.u.get(n) for n in self.u.names})


class Simulation:
    """Simulation(model, method; Y_init, dt, tspan, Ya_init, saveat, ...)
    (simulation.jl:34-73).  Stepping is lh_step_ssprk33: the whole SSPRK33 step
    runs on the device; Dirichlet closures are evaluated on the host at the stage
    times t, t+dt, t+dt/2 and passed as numbers."""

    def __init__(self, model, method, *, Y_init, dt, tspan, Ya_init, callbacks=None, saveat=None,
                 **kwargs):
        if not isinstance(method, SSPRK33):
            raise NotImplementedError("only SSPRK33 is provided on the device")
        if Y_init is None:
            # simulation.jl:50 references an undefined variable here:
            # the reference cannot run this branch either.
            raise NameError("soil_model not defined")
        self.model = model
        self.callbacks = callbacks
        self.integrator = _Integrator(model, Y_init, Ya_init, tspan[0], tspan[1], dt, saveat)


def _time_dependent(model):
    bcs = model.boundary_conditions
    for tag in ("top", "bottom"):
        for c in ("energy", "hydrology"):
            if isinstance(getattr(getattr(bcs, tag), c), Dirichlet):
                return True
    return False


def _device_reads_aux(model, Ya) -> bool:
    """Does the device read a prescribed profile of Ya (T with a viscosity factor; the water fields of
    a heat-only model)?"""
    return bool(_aux_mask(model._backend().kind, model)) and isinstance(Ya, FieldVector)


_PROBE_STEPS = 4096   # stage times probed (and stepped) at a time: 3 closure calls per step


def _aux_constant_over(model, Ya, t, dt, nsteps) -> bool:
    """Are the prescribed profiles the DEVICE reads the same at EVERY stage time of the next `nsteps`
    steps (t + k dt, t + k dt + dt, t + k dt + dt/2)?  The reference's rhs! re-evaluates them at every
    stage time (right_hand_side.jl:37-42); the closures are opaque callables, so every one of those
    times is probed on the model's own centre coordinates -- a profile that starts to change later
    in the chunk (piecewise in t, hold-then-ramp) is seen."""
    if not _device_reads_aux(model, Ya):
        return True
    z = np.asarray(Ya.zc)
    fns = []
    if isinstance(model.energy_model, PrescribedTemperatureModel):
        fns.append(model.energy_model.T_profile)
    if isinstance(model.hydrology_model, PrescribedHydrologyModel):
        fns += [model.hydrology_model.vartheta_l_profile, model.hydrology_model.theta_i_profile]
    for f in fns:
        a0 = np.asarray(f(z, t) + 0.0 * z)
        for k in range(int(nsteps)):
            tk = t + k * dt
            for tt in ((tk,) if k else ()) + (tk + dt, tk + dt / 2):
                if not np.array_equal(a0, np.asarray(f(z, tt) + 0.0 * z)):
                    return False
    return True


def _advance_refreshing_aux(sim: Simulation, nsteps: int):
    """SSPRK33 steps for prescribed profiles that depend on time: the reference's rhs! re-evaluates
    them at EVERY stage time (update_aux_en!/update_aux_hydr!, right_hand_side.jl:37-42), so each
    stage is its own launch (lh_ssprk33_stage) with Ya refreshed in between."""
    it, model = sim.integrator, sim.model
    be = model._backend()
    L = F.lib()
    update_en = make_update_aux(model.energy_model)
    update_hy = make_update_aux(model.hydrology_model)
    if getattr(it, "_stage_state", None) is None:
        it._stage_state = it.u.similar()
    U = it._stage_state
    for _ in range(nsteps):
        for stage, ts in ((1, it.t), (2, it.t + it.dt), (3, it.t + it.dt / 2)):
            update_en(it.p, ts)
            update_hy(it.p, ts)
            vals = np.zeros((2, 2))
            for (f, c), (kind, v) in be.bc_values(model, ts).items():
                if np.ndim(v) == 0:
                    vals[f, c] = float(v)
            be.set_bcs(model, ts)       # kinds and per-column values
            F.check(L.lh_ssprk33_stage(be.ctx, stage, it.u.handle, U.handle, it.p.handle, it.dt,
                                       vals.ctypes.data_as(C.POINTER(C.c_double))), be.ctx)
        it._nsteps_done += 1
        it.t = it.t + it.dt


def _advance(sim: Simulation, nsteps: int):
    it = sim.integrator
    model = sim.model
    be = model._backend()
    L = F.lib()
    ya = it.p.handle if isinstance(it.p, FieldVector) else None
    if nsteps <= 0:
        return
    if _device_reads_aux(model, it.p):
        # prescribed profiles on the device: every stage time of the chunk is probed; only a chunk
        # over which they do not change runs with Ya frozen (the persistent stepper), any other
        # refreshes Ya at every stage like the reference's rhs! does
        if nsteps > _PROBE_STEPS:
            done = 0
            while done < nsteps:
                k = min(_PROBE_STEPS, nsteps - done)
                _advance(sim, k)
                done += k
            return
        if not _aux_constant_over(model, it.p, it.t, it.dt, nsteps):
            return _advance_refreshing_aux(sim, nsteps)
    if _aux_mask(be.kind, model) and isinstance(it.p, FieldVector):
        # constant-in-time profiles: still the values of THIS time (a user may have edited Ya)
        make_update_aux(model.energy_model)(it.p, it.t)
        make_update_aux(model.hydrology_model)(it.p, it.t)
    bcv = None
    if _time_dependent(model):
        t = it.t + it.dt * np.arange(nsteps)
        vals = np.zeros((nsteps, 3, 2, 2))
        base = be.bc_values(model, it.t)
        for (f, c), (kind, v) in base.items():
            if np.ndim(v) == 0:
                vals[:, :, f, c] = float(v)
        for si, off in enumerate((0.0, it.dt, it.dt / 2)):
            for k in range(nsteps):
                for (f, c), (kind, v) in be.bc_values(model, t[k] + off).items():
                    if kind == F.LH_BC_DIRICHLET and np.ndim(v) == 0:
                        vals[k, si, f, c] = float(v)
        bcv = np.ascontiguousarray(vals)
    be.set_bcs(model, it.t)
    if not getattr(it, "_placed", False) and _placement_tuning_wanted(model):
        it._placed = True
        F.check(L.lh_tune_placement(be.ctx, it.u.handle, ya, None, 0, 0, None, None), be.ctx)
    F.check(L.lh_step_ssprk33(be.ctx, it.u.handle, ya, it.t, it.dt, nsteps,
                              bcv.ctypes.data_as(C.POINTER(C.c_double)) if bcv is not None
                              else None), be.ctx)
    it._nsteps_done += nsteps
    it.t = it.t + nsteps * it.dt


def step(sim: Simulation):
    """step!(simulation) (simulation.jl:79-80)."""
    _advance(sim, 1)
    return None


def run(sim: Simulation):
    """run!(simulation) (simulation.jl:86-87): integrate to tspan[2], saving
    every `saveat` like DiffEq's saveat."""
    it = sim.integrator
    total = int(round((it.tf - it.t) / it.dt))
    chunk = total
    if it.saveat:
        chunk = max(1, int(round(it.saveat / it.dt)))
    done = 0
    while done < total:
        n = min(chunk, total - done)
        _advance(sim, n)
        done += n
        it._save()
    return it.sol

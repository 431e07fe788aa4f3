"""landhydrology.jl_amd -- MI355X-native batched soil-column tendency path behind
the LandHydrology.jl SoilModel API.

The directory name contains a dot, so it is loaded through
`__graft_entry__.load_package()` (or tests/conftest.py) under the module name
`landhydrology_jl_amd`.  Compute lives in lib/liblandhydro_hip.so (HIP, gfx950);
this package is the host-side mirror of the reference's Julia interface.
"""
from . import _ffi, case_model, parameterizations, partition, workloads
from ._ffi import LandHydroError, ModelError
from .parameterizations import *  # noqa: F401,F403
from .soil import (Column, Dirichlet, EarthParameterSet, FieldVector, Float32, Float64,
                   FreeDrainage, IceImpedance, NoBC, NoEffect, PrescribedAtmosForcing,
                   PrescribedHydrologyModel, PrescribedTemperatureModel, Simulation,
                   SoilColumnBC, SoilComponentBC, SoilEnergyModel, SoilHydrologyModel, SoilModel,
                   SoilParams, SSPRK33, TemperatureDependentViscosity, VerticalFlux, boundary_fluxes,
                   compute_turbulent_surface_fluxes, coordinates,
                   default_initial_conditions, initialize_states, make_function_space, make_rhs,
                   make_update_aux, run, stable_dt, step, step_adaptive, tune_placement, vanGenuchten)

__all__ = [n for n in dir() if not n.startswith("_")]

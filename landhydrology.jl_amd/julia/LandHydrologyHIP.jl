# LandHydrologyHIP.jl -- the reference-side binding a LandHydrology.jl maintainer
# would add to route `make_rhs(model)` / `Simulation` through liblandhydro_hip.so.
#
# STATUS: text only.  Julia is not installed in the build image, so this file has
# never been executed; it documents the ccall surface one-to-one with
# include/landhydro.h.  The tested host binding is the Python mirror
# (landhydrology.jl_amd/soil.py), which makes exactly these calls through ctypes.
#
# Usage sketch (single column, exactly the reference's API):
#
#   using LandHydrology, LandHydrologyHIP
#   rhs! = make_rhs(soil_model, HIPBackend())          # same 4-argument closure
#   prob = ODEProblem(rhs!, Y, (t0, tf), Ya)            # Y, Ya: host FieldVectors
#
# Ensemble use keeps the state on the device:
#
#   ens  = ColumnEnsemble(soil_model, ncolumns)          # lh_create + parameters
#   Y    = upload(ens, ϑ_l = A1, θ_i = A2)              # [nelements, ncolumns] arrays
#   step_ssprk33!(ens, Y, nothing, t, dt, nsteps)
#
# One process per GPU, columns block-partitioned (block_range), global adaptive dt:
#
#   attach_comm!(ens, rank, nranks, id)                  # id = comm_unique_id() on rank 0, broadcast
#   rhs_stable_dt!(ens, dY, Y, nothing, t, 0.5, dt_dev)  # dt_dev now holds the GLOBAL minimum
#   step_ssprk33_device_dt!(ens, Y, nothing, t, dt_dev)
#
module LandHydrologyHIP

using LandHydrology
using LandHydrology.SoilInterface
using LandHydrology.SoilInterface: SoilModel, SoilEnergyModel, SoilHydrologyModel,
    PrescribedTemperatureModel, PrescribedHydrologyModel, NoBC, VerticalFlux, Dirichlet,
    FreeDrainage, SoilComponentBC, PrescribedAtmosForcing
using LandHydrology.SoilInterface.SoilWaterParameterizations:
    NoEffect, TemperatureDependentViscosity, IceImpedance
using CLIMAParameters.Planet: ρ_cloud_liq, ρ_cloud_ice, cp_l, cp_i, T_0, LH_f0, R_v, R_d, grav, cp_d,
    cp_v, LH_v0, T_triple, press_triple
using CLIMAParameters.Atmos.Microphysics: K_therm
using CLIMAParameters.SubgridScale: von_karman_const
import LandHydrology.SoilInterface: make_rhs

export HIPBackend, ColumnEnsemble, upload, download, device_rhs!, step_ssprk33!, stable_dt,
    rhs_stable_dt!, step_ssprk33_device_dt!, block_range, comm_unique_id, attach_comm!, detach_comm!,
    step_engine

const lib = get(ENV, "LANDHYDRO_HIP_LIB", "liblandhydro_hip.so")

# ---- mirrors of the C structs (include/landhydro.h) -------------------------
struct lh_config
    ncols::Int64
    nlev::Int32
    dtype::Int32        # LH_F32 = 0, LH_F64 = 1
    zmin::Float64
    zmax::Float64
    model::Int32        # RICHARDS = 0, HEAT = 1, COUPLED = 2
    device::Int32
    stream::Ptr{Cvoid}
end
struct lh_earth_params
    rho_liq::Float64; rho_ice::Float64; cp_l::Float64; cp_i::Float64
    T_0::Float64; LH_f0::Float64; K_therm::Float64
end
struct lh_soil_params
    nu::Float64; S_s::Float64; nu_ss_gravel::Float64; nu_ss_om::Float64; nu_ss_quartz::Float64
    rho_c_ds::Float64; kappa_solid::Float64; rho_p::Float64; kappa_sat_unfrozen::Float64
    kappa_sat_frozen::Float64; a::Float64; b::Float64; kappa_dry_parameter::Float64
end
struct lh_vg_params
    n::Float64; alpha::Float64; theta_r::Float64; Ksat::Float64
end
struct lh_atmos_forcing          # PrescribedAtmosForcing + z_0m, z_0s + the constants of :553-620
    u_atm::Float64; theta_atm::Float64; z_atm::Float64; theta_scale::Float64; rho_a_sfc::Float64
    q_atm::Float64; z_0m::Float64; z_0s::Float64
    R_v::Float64; R_d::Float64; grav::Float64; cp_d::Float64; cp_v::Float64; LH_v0::Float64
    T_triple::Float64; press_triple::Float64; von_karman::Float64
end

const LH_BC_NONE, LH_BC_FLUX, LH_BC_DIRICHLET, LH_BC_FREE_DRAINAGE = Int32(0), Int32(1), Int32(2), Int32(3)
const LH_FACE_BOTTOM, LH_FACE_TOP = Int32(0), Int32(1)
const LH_COMP_ENERGY, LH_COMP_HYDROLOGY = Int32(0), Int32(1)
const LH_VAR = (ϑ_l = Int32(0), θ_i = Int32(1), ρe_int = Int32(2), T = Int32(3))

struct HIPBackend
    device::Int32
end
HIPBackend() = HIPBackend(Int32(-1))

# every entry point returns a status; non-zero becomes error(msg) like the
# reference's ArgumentError / error(...) paths (boundary_conditions.jl:188,524)
function check(ctx, rc)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:lh_last_error, lib), Cstring, (Ptr{Cvoid},), ctx))
    error("landhydro_hip [$rc]: $msg")
end

model_kind(::PrescribedTemperatureModel, ::SoilHydrologyModel) = Int32(0)
model_kind(::SoilEnergyModel, ::PrescribedHydrologyModel) = Int32(1)
model_kind(::SoilEnergyModel, ::SoilHydrologyModel) = Int32(2)

bc_kind(::NoBC) = LH_BC_NONE
bc_kind(::VerticalFlux) = LH_BC_FLUX
bc_kind(::Dirichlet) = LH_BC_DIRICHLET
bc_kind(::FreeDrainage) = LH_BC_FREE_DRAINAGE
bc_value(bc::VerticalFlux, t) = Float64(bc.flux)
bc_value(bc::Dirichlet, t) = Float64(bc.state_value(t))
bc_value(::Any, t) = 0.0

"""
    ColumnEnsemble(model::SoilModel{FT}, ncolumns; backend = HIPBackend())

`ncolumns` independent copies of `model.domain` on one device (lh_create +
parameter setters).  Everything `SoilModel` holds maps to one setter call.
"""
mutable struct ColumnEnsemble{FT, M <: SoilModel}
    ctx::Ptr{Cvoid}
    model::M
    ncolumns::Int
end

function ColumnEnsemble(model::SoilModel{FT}, ncolumns::Integer; backend = HIPBackend()) where {FT}
    d = model.domain
    cfg = Ref(lh_config(ncolumns, d.nelements, FT == Float64 ? 1 : 0, d.zlim[1], d.zlim[2],
                        model_kind(model.energy_model, model.hydrology_model), backend.device, C_NULL))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:lh_create, lib), Cint, (Ptr{Ptr{Cvoid}}, Ptr{lh_config}), ctx, cfg)
    rc == 0 || check(C_NULL, rc)
    ens = ColumnEnsemble{FT, typeof(model)}(ctx[], model, ncolumns)
    # lh_destroy frees every state of the context too.  Julia runs finalizers in no particular
    # order, so the handle is nulled here and a DeviceState finalized later skips lh_state_destroy
    # (its `ens` field keeps this object alive until then).
    finalizer(ens) do e
        if e.ctx != C_NULL
            ccall((:lh_destroy, lib), Cint, (Ptr{Cvoid},), e.ctx)
            e.ctx = C_NULL
        end
    end
    ps = model.earth_param_set
    if ps !== nothing && !(model.energy_model isa PrescribedTemperatureModel)
        ep = Ref(lh_earth_params(ρ_cloud_liq(ps), ρ_cloud_ice(ps), cp_l(ps), cp_i(ps), T_0(ps),
                                 LH_f0(ps), K_therm(ps)))
        check(ens.ctx, ccall((:lh_set_earth_params, lib), Cint, (Ptr{Cvoid}, Ptr{lh_earth_params}), ens.ctx, ep))
    end
    sp = model.soil_param_set
    s = Ref(lh_soil_params(sp.ν, sp.S_s, sp.ν_ss_gravel, sp.ν_ss_om, sp.ν_ss_quartz, sp.ρc_ds,
                           sp.κ_solid, sp.ρp, sp.κ_sat_unfrozen, sp.κ_sat_frozen, sp.a, sp.b,
                           sp.κ_dry_parameter))
    check(ens.ctx, ccall((:lh_set_soil_params, lib), Cint, (Ptr{Cvoid}, Ptr{lh_soil_params}), ens.ctx, s))
    hy = model.hydrology_model
    if hy isa SoilHydrologyModel
        hm = hy.hydraulic_model
        v = Ref(lh_vg_params(hm.n, hm.α, hm.θr, hm.Ksat))
        check(ens.ctx, ccall((:lh_set_vg_params, lib), Cint, (Ptr{Cvoid}, Ptr{lh_vg_params}), ens.ctx, v))
        vf, imf = hy.viscosity_factor, hy.impedance_factor
        vk = vf isa TemperatureDependentViscosity
        ik = imf isa IceImpedance
        check(ens.ctx, ccall((:lh_set_conductivity_factors, lib), Cint,
                             (Ptr{Cvoid}, Int32, Float64, Float64, Int32, Float64), ens.ctx,
                             Int32(vk), vk ? vf.γ : 2.64e-2, vk ? vf.T_ref : 288.0, Int32(ik),
                             ik ? imf.Ω : 7.0))
    end
    set_bcs!(ens, 0.0)
    return ens
end

# Dirichlet.state_value and friends are Julia closures of t
# (boundary_conditions.jl:61-64): evaluate on the host, pass numbers.
function set_bcs!(ens::ColumnEnsemble, t)
    bcs = ens.model.boundary_conditions
    for (face, fbc) in ((LH_FACE_BOTTOM, bcs.bottom), (LH_FACE_TOP, bcs.top))
        if fbc isa PrescribedAtmosForcing
            # boundary_conditions.jl:523-528: only valid at the top (SoilColumnBC's type parameters
            # already exclude the bottom).  The surface fluxes are computed on the device from the
            # top cell before every tendency evaluation (lh_set_atmos_forcing, csrc/lh_atmos.hpp).
            face == LH_FACE_TOP || error("Prescribed atmosphere-driven boundary conditions are only valid at the top of the soil column.")
            ps, sp = ens.model.earth_param_set, ens.model.soil_param_set
            f = Ref(lh_atmos_forcing(fbc.u_atm, fbc.θ_atm, fbc.z_atm, fbc.θ_scale, fbc.ρ_a_sfc, fbc.q_atm,
                                     sp.z_0m, sp.z_0s, R_v(ps), R_d(ps), grav(ps), cp_d(ps), cp_v(ps),
                                     LH_v0(ps), T_triple(ps), press_triple(ps), von_karman_const(ps)))
            check(ens.ctx, ccall((:lh_set_atmos_forcing, lib), Cint,
                                 (Ptr{Cvoid}, Ptr{lh_atmos_forcing}, Ptr{Float64}), ens.ctx, f, C_NULL))
            continue
        end
        for (comp, bc) in ((LH_COMP_ENERGY, fbc.energy), (LH_COMP_HYDROLOGY, fbc.hydrology))
            check(ens.ctx, ccall((:lh_set_bc, lib), Cint,
                                 (Ptr{Cvoid}, Int32, Int32, Int32, Float64, Ptr{Float64}),
                                 ens.ctx, face, comp, bc_kind(bc), bc_value(bc, t), C_NULL))
        end
    end
end

mutable struct DeviceState
    ens::ColumnEnsemble
    handle::Ptr{Cvoid}
end

function new_state(ens::ColumnEnsemble, mask::UInt32 = UInt32(0))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ens.ctx, ccall((:lh_state_create, lib), Cint, (Ptr{Cvoid}, UInt32, Ptr{Ptr{Cvoid}}), ens.ctx, mask, h))
    st = DeviceState(ens, h[])
    finalizer(st) do s
        # the context may already be gone (and with it this state): nothing left to free then
        if s.ens.ctx != C_NULL && s.handle != C_NULL
            ccall((:lh_state_destroy, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), s.ens.ctx, s.handle)
        end
        s.handle = C_NULL
    end
    return st
end

# A is [nelements, ncolumns], column-major: level-fastest per column, exactly
# `parent(field)` of the reference when ncolumns == 1 (coupled.jl:198-200).
function upload!(st::DeviceState, var::Symbol, A::AbstractMatrix{FT}) where {FT}
    if iszero(A)
        # an all-zero field is a fill: the library then KNOWS the plane is zero and neither reads a
        # zero θ_i plane nor re-stores dθ_i = 0
        check(st.ens.ctx, ccall((:lh_state_fill, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float64),
                                st.ens.ctx, st.handle, LH_VAR[var], 0.0))
        return
    end
    check(st.ens.ctx, ccall((:lh_upload, lib), Cint,
                            (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Int64),
                            st.ens.ctx, st.handle, LH_VAR[var], A, 1, size(A, 1)))
end
# one value per level, the same in every column (Ya.soil.T .= T_profile.(zc, t) is a function of z and t
# only, right_hand_side.jl:54-62): nelements numbers cross PCIe, and the column kernels read a
# level-uniform prescribed field from LDS instead of a plane
function upload!(st::DeviceState, var::Symbol, v::AbstractVector{FT}) where {FT}
    length(v) == st.ens.model.domain.nelements || error("a per-level profile has nelements values")
    check(st.ens.ctx, ccall((:lh_upload_profile, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}),
                            st.ens.ctx, st.handle, LH_VAR[var], v))
end
function download!(A::AbstractMatrix{FT}, st::DeviceState, var::Symbol) where {FT}
    check(st.ens.ctx, ccall((:lh_download, lib), Cint,
                            (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int64, Int64),
                            st.ens.ctx, st.handle, LH_VAR[var], A, 1, size(A, 1)))
    return A
end
function upload(ens::ColumnEnsemble; fields...)
    st = new_state(ens)
    for (k, A) in fields
        upload!(st, k, A)
    end
    return st
end
"interior_values(X, face, cs) for every column: one level of one variable (1 = bottom, nelements = top)"
function download_level(st::DeviceState, var::Symbol, level::Integer, ::Type{FT}) where {FT}
    out = Vector{FT}(undef, st.ens.ncolumns)
    check(st.ens.ctx, ccall((:lh_download_level, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}),
                            st.ens.ctx, st.handle, LH_VAR[var], level - 1, out))
    return out
end
download(st::DeviceState, var::Symbol, ::Type{FT}) where {FT} =
    download!(Matrix{FT}(undef, st.ens.model.domain.nelements, st.ens.ncolumns), st, var)

"rhs!(dY, Y, Ya, t) on device states (right_hand_side.jl:37-42)"
function device_rhs!(ens::ColumnEnsemble, dY::DeviceState, Y::DeviceState, Ya, t)
    set_bcs!(ens, t)
    ya = Ya === nothing ? C_NULL : Ya.handle
    check(ens.ctx, ccall((:lh_rhs, lib), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                         ens.ctx, t, Y.handle, ya, dY.handle))
    return dY
end

"""
    boundary_fluxes(ens, Y, Ya, face, t) -> (fρe_int = ..., fϑ_l = ...)

`boundary_fluxes(X, bc::SoilComponentBC, face, model, cs, t)` (boundary_conditions.jl:470-489) of every
column for the device state: the two SetValue fluxes of the tendency launch (same device functions),
`ncolumns` values each; `NaN` where the component has no boundary condition (`nothing` there).
"""
function boundary_fluxes(ens::ColumnEnsemble, Y::DeviceState, Ya, face::Symbol, t)
    face in (:top, :bottom) || throw(ArgumentError("Expected :top or :bottom"))
    set_bcs!(ens, t)
    ya = Ya === nothing ? C_NULL : Ya.handle
    fe = Vector{Float64}(undef, ens.ncolumns)
    fw = Vector{Float64}(undef, ens.ncolumns)
    check(ens.ctx, ccall((:lh_boundary_fluxes, lib), Cint,
                         (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Int32, Ptr{Float64}, Ptr{Float64}),
                         ens.ctx, Y.handle, ya, Float64(t), face === :top ? Int32(1) : Int32(0), fe, fw))
    return (fρe_int = fe, fϑ_l = fw)
end

# Planes of Ya the DEVICE reads (the prescribed fields of make_update_aux, right_hand_side.jl:54-81):
# (ϑ_l, θ_i) for a prescribed hydrology; T for a prescribed temperature only when a viscosity
# factor consumes it (:160).  Mask bits as LH_MASK(var) of include/landhydro.h.
function aux_mask(model::SoilModel)
    model.hydrology_model isa PrescribedHydrologyModel && return UInt32(0b0011)
    if model.energy_model isa PrescribedTemperatureModel &&
       model.hydrology_model.viscosity_factor isa TemperatureDependentViscosity
        return UInt32(0b1000)
    end
    return UInt32(0)
end

# the prescribed profiles at time t on the centre coordinates zc (a Vector), as [nelements, ncolumns]
# per-level profiles uploaded into the aux state Yad (every column the same profile: the reference's
# closures are functions of (z, t) only)
function upload_aux!(ens::ColumnEnsemble, Yad::DeviceState, zc::AbstractVector{FT}, t) where {FT}
    m = ens.model
    rep(v) = collect(FT, v)   # a per-level profile: upload!(::DeviceState, ::Symbol, ::AbstractVector)
    if m.hydrology_model isa PrescribedHydrologyModel
        upload!(Yad, :ϑ_l, rep(m.hydrology_model.ϑ_l_profile.(zc, t)))
        upload!(Yad, :θ_i, rep(m.hydrology_model.θ_i_profile.(zc, t)))
    else
        upload!(Yad, :T, rep(m.energy_model.T_profile.(zc, t)))
    end
    return Yad
end

coordinates(ens::ColumnEnsemble) = begin
    zc = Vector{Float64}(undef, ens.model.domain.nelements)
    check(ens.ctx, ccall((:lh_coordinates, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), ens.ctx, zc))
    zc
end

bc_stage_values(ens::ColumnEnsemble, ts) = begin
    bcs = ens.model.boundary_conditions
    vals = Array{Float64}(undef, 2, 2)                 # (component, face): C order [face][component]
    fill!(vals, 0.0)
    for (f, fbc) in enumerate((bcs.bottom, bcs.top))
        fbc isa SoilComponentBC || continue            # a prescribed atmosphere has no stage values
        for (c, bc) in enumerate((fbc.energy, fbc.hydrology))
            vals[c, f] = bc_value(bc, ts)
        end
    end
    vals
end

"compute_turbulent_surface_fluxes.(energy, hydrology, model, ϑ_l, θ_i, T) on the device (boundary_conditions.jl:553-620)"
function turbulent_surface_fluxes(ens::ColumnEnsemble, ϑ_l::Vector{Float64}, θ_i::Vector{Float64}, T::Vector{Float64})
    n = length(ϑ_l)
    heat, water = Vector{Float64}(undef, n), Vector{Float64}(undef, n)
    check(ens.ctx, ccall((:lh_atmos_surface_fluxes, lib), Cint,
                         (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                         ens.ctx, n, ϑ_l, θ_i, T, heat, water))
    return heat, water
end

"""
    step_ssprk33!(ens, Y, Ya, t, dt, nsteps; aux_depends_on_time = false)

`solve(prob, SSPRK33(), dt = dt)` for `nsteps` steps, state stays on the device
(simulation.jl:58-87).  `Ya` is `nothing` or the aux DeviceState (see `aux_mask`).  With
`aux_depends_on_time` the prescribed profiles are re-evaluated and uploaded at every STAGE time,
as the reference's rhs! does (right_hand_side.jl:37-42), one lh_ssprk33_stage launch per stage;
otherwise all steps run in one lh_step_ssprk33 call with Ya as it is.
"""
function step_ssprk33!(ens::ColumnEnsemble{FT}, Y::DeviceState, Ya, t, dt, nsteps; aux_depends_on_time = false) where {FT}
    ya = Ya === nothing ? C_NULL : Ya.handle
    if aux_depends_on_time && Ya !== nothing
        zc = FT.(coordinates(ens))
        U = new_state(ens)
        for s in 1:nsteps, (stage, off) in enumerate((0.0, dt, dt / 2))
            ts = t + (s - 1) * dt + off
            upload_aux!(ens, Ya, zc, ts)
            set_bcs!(ens, ts)
            check(ens.ctx, ccall((:lh_ssprk33_stage, lib), Cint,
                                 (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Float64}),
                                 ens.ctx, Int32(stage), Y.handle, U.handle, ya, dt, bc_stage_values(ens, ts)))
        end
        return Y
    end
    # Dirichlet closures at the stage times t, t+dt, t+dt/2 -> [nsteps][3][2][2] doubles in C order
    vals = Array{Float64}(undef, 2, 2, 3, nsteps)      # (component, face, stage, step): column-major = C order reversed
    for s in 1:nsteps, (k, off) in enumerate((0.0, dt, dt / 2))
        vals[:, :, k, s] = bc_stage_values(ens, t + (s - 1) * dt + off)
    end
    set_bcs!(ens, t)
    # boundary values that do not change over the call go in through set_bcs! alone (NULL here): the
    # library then evaluates the closures of a Dirichlet face state once per call, not once per stage
    base = bc_stage_values(ens, t)
    constant = all(vals[:, :, k, s] == base for k in 1:3, s in 1:nsteps)
    check(ens.ctx, ccall((:lh_step_ssprk33, lib), Cint,
                         (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int64, Ptr{Float64}),
                         ens.ctx, Y.handle, ya, t, dt, nsteps, constant ? C_NULL : vals))
    return Y
end

"""
    step_engine(ens, nsteps; per_stage_boundary_values = false)

`:column_stepper` or `:fused_stages`: the engine `lh_step_ssprk33` runs such a call with (the
results do not depend on it).
"""
function step_engine(ens::ColumnEnsemble, nsteps::Integer; per_stage_boundary_values::Bool = false)
    e = ccall((:lh_step_engine, lib), Cint, (Ptr{Cvoid}, Int64, Int32), ens.ctx, Int64(nsteps),
              Int32(per_stage_boundary_values))
    e < 0 && check(ens.ctx, e)
    return e == 1 ? :column_stepper : :fused_stages
end

"rhs! plus this rank's stable-step bound (one FT value at the device pointer `dt_dev`) in one launch"
function rhs_stable_dt!(ens::ColumnEnsemble, dY::DeviceState, Y::DeviceState, Ya, t, courant, dt_dev::Ptr{Cvoid})
    set_bcs!(ens, t)
    ya = Ya === nothing ? C_NULL : Ya.handle
    check(ens.ctx, ccall((:lh_rhs_stable_dt, lib), Cint,
                         (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}),
                         ens.ctx, t, Y.handle, ya, dY.handle, courant, dt_dev))
    return dY
end

"one SSPRK33 step with dt read from device memory (after the min all-reduce of `dt_dev`)"
function step_ssprk33_device_dt!(ens::ColumnEnsemble, Y::DeviceState, Ya, t, dt_dev::Ptr{Cvoid})
    set_bcs!(ens, t)
    ya = Ya === nothing ? C_NULL : Ya.handle
    check(ens.ctx, ccall((:lh_step_ssprk33_device_dt, lib), Cint,
                         (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Ptr{Float64}),
                         ens.ctx, Y.handle, ya, t, dt_dev, C_NULL))
    return Y
end

"""
    step_ssprk33_adaptive!(ens, Y, Ya, t, courant, nsteps, dt_dev; dt_max = 0.0, elapsed_dev = C_NULL)

`nsteps` adaptive SSPRK33 steps with nothing leaving the device (lh_step_ssprk33_adaptive): per step
the tendency and the stable-step bound of `Y` in one launch (the RCCL min over ranks follows when a
communicator is attached), then stages 2 and 3 -- three evaluations of the right-hand side per step,
bitwise the result of `rhs_stable_dt!` + `step_ssprk33_device_dt!`.  Constant boundary values only.
`dt_dev` / `elapsed_dev`: one FT each in device memory (last dt; accumulated simulated time).
"""
function step_ssprk33_adaptive!(ens::ColumnEnsemble, Y::DeviceState, Ya, t, courant, nsteps, dt_dev::Ptr{Cvoid};
                                dt_max = 0.0, elapsed_dev::Ptr{Cvoid} = C_NULL)
    set_bcs!(ens, t)
    ya = Ya === nothing ? C_NULL : Ya.handle
    check(ens.ctx, ccall((:lh_step_ssprk33_adaptive, lib), Cint,
                         (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Int64, Ptr{Cvoid}, Ptr{Cvoid}),
                         ens.ctx, Y.handle, ya, t, courant, dt_max, nsteps, dt_dev, elapsed_dev))
    return Y
end

"""
    tune_placement!(ens, Y, Ya, dY = nothing; max_candidates = 0, move_input = true)

Let the library place the state written by `rhs!` (`dY` given) or the SSPRK33 stage state
(`dY === nothing`) in HBM by timing the real launch (lh_tune_placement).  Returns
`(ms_before, ms_after)`.  Worth one call per (Y, dY) pair for ensembles with planes >= 32 MiB.
"""
function tune_placement!(ens::ColumnEnsemble, Y::DeviceState, Ya, dY = nothing; max_candidates = 0, move_input = true)
    set_bcs!(ens, 0.0)
    b, a = Ref{Cfloat}(0), Ref{Cfloat}(0)
    ya = Ya === nothing ? C_NULL : Ya.handle
    check(ens.ctx, ccall((:lh_tune_placement, lib), Cint,
                         (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, UInt32, Ptr{Cfloat}, Ptr{Cfloat}),
                         ens.ctx, Y.handle, ya, dY === nothing ? C_NULL : dY.handle, max_candidates,
                         move_input ? UInt32(1) : UInt32(0), b, a))
    return b[], a[]
end

function stable_dt(ens::ColumnEnsemble, Y::DeviceState, Ya = nothing; courant = 0.5)
    out = Ref{Float64}(0.0)
    ya = Ya === nothing ? C_NULL : Ya.handle
    check(ens.ctx, ccall((:lh_stable_dt, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ptr{Float64}),
                         ens.ctx, Y.handle, ya, courant, out))
    return out[]
end

"""
    make_rhs(model::SoilModel, ::HIPBackend)

Drop-in for `make_rhs(model)` (right_hand_side.jl:33-44): returns a closure with
the same `rhs!(dY, Y, Ya, t) -> dY` signature working on the reference's host
`FieldVector`s (one column).  Each call uploads Y (and the prescribed fields of
Ya), runs lh_rhs and downloads dY, so it is a correctness bridge for existing
scripts; throughput comes from `ColumnEnsemble` + `step_ssprk33!`, where the state
never leaves the device.
"""
function make_rhs(model::SoilModel{FT}, backend::HIPBackend) where {FT}
    ens = ColumnEnsemble(model, 1; backend = backend)
    update_aux_en! = LandHydrology.SoilInterface.make_update_aux(model.energy_model)
    update_aux_hydr! = LandHydrology.SoilInterface.make_update_aux(model.hydrology_model)
    Yd, dYd = new_state(ens), new_state(ens)
    am = aux_mask(model)
    Yad = am == 0 ? nothing : new_state(ens, am)
    col(field) = reshape(parent(field), :, 1)          # one column, level-fastest (coupled.jl:198-200)
    # (the closure has its own name: an inner `rhs!` would shadow nothing here, but a 4-argument
    # local named like the 5-argument device method invites exactly that mistake)
    function rhs_host!(dY, Y, Ya, t)
        update_aux_en!(Ya, t)                          # right_hand_side.jl:38-39
        update_aux_hydr!(Ya, t)
        ys = getproperty(Y, model.name)
        for k in propertynames(ys)
            upload!(Yd, k, col(getproperty(ys, k)))
        end
        if Yad !== nothing                             # the prescribed fields the device reads
            ya = getproperty(Ya, model.name)
            if model.hydrology_model isa PrescribedHydrologyModel
                upload!(Yad, :ϑ_l, col(ya.ϑ_l))
                upload!(Yad, :θ_i, col(ya.θ_i))
            else
                upload!(Yad, :T, col(ya.T))
            end
        end
        device_rhs!(ens, dYd, Yd, Yad, t)
        ds = getproperty(dY, model.name)
        for k in propertynames(ds)
            download!(col(getproperty(ds, k)), dYd, k)
        end
        return dY                                      # :41
    end
    return rhs_host!
end

# ---- multi-GPU: block partition + the one collective (include/landhydro.h, SURVEY 8e) --------

"columns [lo, hi) (0-based, half-open) owned by `rank` of `nranks`"
function block_range(ncols_global::Integer, rank::Integer, nranks::Integer)
    lo, hi = Ref{Int64}(0), Ref{Int64}(0)
    rc = ccall((:lh_block_range, lib), Cint, (Int64, Int32, Int32, Ptr{Int64}, Ptr{Int64}),
               ncols_global, rank, nranks, lo, hi)
    rc == 0 || error("lh_block_range: bad arguments")
    return lo[], hi[]
end

"ncclGetUniqueId: 128 bytes drawn on ONE rank; ship them to the others (e.g. MPI.Bcast!) before attach_comm!"
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = ccall((:lh_comm_unique_id, lib), Cint, (Ptr{Cvoid},), id)
    rc == 0 || check(C_NULL, rc)
    return id
end

"""
    attach_comm!(ens, rank, nranks, id)

One process per GPU, each with its ColumnEnsemble over its block of columns: after this
collective call `rhs_stable_dt!` and `stable_dt` deliver the GLOBAL minimum of the ranks' step
bounds -- the library enqueues ncclAllReduce(count = 1, min) over RCCL/xGMI on its own stream.
With MPI.jl: `id = rank == 0 ? comm_unique_id() : Vector{UInt8}(undef, 128); MPI.Bcast!(id, 0, comm)`.
"""
function attach_comm!(ens::ColumnEnsemble, rank::Integer, nranks::Integer, id::Vector{UInt8})
    length(id) == 128 || error("the communicator id has 128 bytes")
    check(ens.ctx, ccall((:lh_comm_init, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}),
                         ens.ctx, rank, nranks, id))
    return ens
end
detach_comm!(ens::ColumnEnsemble) =
    check(ens.ctx, ccall((:lh_comm_destroy, lib), Cint, (Ptr{Cvoid},), ens.ctx))

end # module

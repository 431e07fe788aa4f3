// lh_api.hip -- the C ABI of liblandhydro_hip.so (include/landhydro.h).
// Host logic only: parameter rounding, validation, state management, launches.
// There is no CPU fallback: without a HIP device lh_create fails loudly.
#include "../../include/landhydro.h"
#include "lh_launch.hpp"
#include "lh_fastmath.hpp"
#include "lh_closures.hpp"

#include <rccl/rccl.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace lh;

// default address stagger between consecutively allocated planes (bytes)
#ifndef LH_PLANE_STAGGER
#define LH_PLANE_STAGGER 0 /* no consistent gain measured (profiles/round1_tune_plane_stagger.txt): placement noise is +-5 % */
#endif

namespace {

thread_local std::string g_create_error;

struct HostParams {
    lh_earth_params earth{};
    bool earth_set = false;
    lh_soil_params soil{};
    lh_vg_params vg{};
    int32_t viscosity_kind = LH_FACTOR_NONE, impedance_kind = LH_FACTOR_NONE;
    double gamma = 2.64e-2, T_ref_visc = 288.0, Omega = 7.0; // SoilWaterParameterizations.jl:49-64
    int32_t bc_kind[2][2] = {{0, 0}, {0, 0}};
    double bc_value[2][2] = {{0, 0}, {0, 0}};
    int32_t consistent_bottom_sign = 0;
    bool atmos_on = false;
    lh_atmos_forcing atmos{};
};

} // namespace

struct lh_state {
    lh_ctx* ctx;
    uint32_t mask;
    // Planes the library KNOWS to be all (+)zeros: set by creation (memset), a zero fill and by
    // launches that store zeros; cleared by anything else that may write (upload, fill, copy of
    // a non-zero plane, a launch writing the plane, handing out the device pointer).  A launch
    // neither reads a theta_i plane known to be zero (rhs_kernel NOICE) nor re-stores the
    // identically zero d theta_i into a plane that already holds zeros (no kernel stores it).
    uint32_t zero_mask;
    // Planes whose device pointer was handed out (lh_state_device_ptr) and not released
    // (lh_state_release_ptr): the caller may write through the pointer at any time, so the library
    // assumes NOTHING about their contents -- the zero bit is never set for them (a zero fill, a copy
    // of a zero plane or the d theta_i = 0 clear of lh_rhs still write the zeros, every time).
    uint32_t exposed_mask;
    // LEVEL-UNIFORM variables (lh_upload_profile): profile[var] holds FT[nlev] on the device and IS the
    // variable; stale_mask marks planes that have not received those values (the column kernels read a
    // prescribed profile of Ya from LDS and never need them; everything else materialises first)
    uint32_t profile_mask, stale_mask;
    void* profile[LH_NVARS];
    void* plane[LH_NVARS]; // what kernels address (a slot of a context arena)
};

// Planes are carved from a few large device allocations (8 plane slots each)
// instead of one hipMalloc per plane: states that are streamed together then sit
// next to each other in one contiguous range, and state creation costs no
// allocator round trip.
struct PlaneArena {
    char* base = nullptr;
    size_t slot_bytes = 0;
    int nslots = 0;
    std::vector<char> used;
};

struct lh_ctx {
    lh_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t stride = 0; // plane row length (elements), multiple of 64
    size_t esize = 8;
    HostParams hp;
    void* d_zc = nullptr;              // FT[nlev]
    void* d_pc[LH_PC_COUNT] = {};      // FT[ncols] or null
    double pc_lo[LH_PC_COUNT] = {}, pc_hi[LH_PC_COUNT] = {}; // range of each per-column array (NaN if it holds a NaN)
    void* d_bc_pc[2][2] = {};          // FT[ncols] or null
    uint32_t* d_status = nullptr;
    void* d_dt = nullptr;              // FT scratch for lh_stable_dt
    void* d_atm_pc[3] = {};            // per-column u_atm / theta_atm / q_atm (FT[ncols]) or null
    void* d_atm_flux[2] = {};          // the top-face fluxes of the prescribed atmosphere: heat, water (FT[ncols])
    double* d_math_tab = nullptr;      // log2/exp2 tables of MathFast<double>
    char* h_ring = nullptr;            // pinned staging ring of lh_upload_profile (asynchronous small uploads)
    size_t ring_bytes = 0, ring_pos = 0;
    lh_state* scratch_u1 = nullptr;    // SSPRK33 stage state
    lh_state* scratch_u2 = nullptr;    // second stage state (level-segmented launches cannot update U1 in place)
    lh_state* scratch_k1 = nullptr;    // f(Y) of lh_step_ssprk33_adaptive
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int math = MATH_FAST;
    Tune tune;
    unsigned plane_counter = 0;
    std::vector<PlaneArena> arenas;
    std::vector<double> zc_host;
    std::string err;
    std::vector<lh_state*> states;
    // the one collective of the path: min all-reduce of the stable-step bound
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_nranks = 1;
};

namespace {

// roctx range around one call of the path (rhs, stage, step, allreduce): what a rocprofv3
// --marker-trace of a multi-GPU run is read by; a few tens of nanoseconds without a tool attached
struct Range {
    explicit Range(const char* name) { roctxRangePushA(name); }
    ~Range() { roctxRangePop(); }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

int fail(lh_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define LH_HIP(ctx, call)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? LH_ENOMEM : LH_ENODEVICE,           \
                        "%s failed: %s", #call, hipGetErrorString(e_));                      \
    } while (0)

// In-place min all-reduce of one FT value over the attached communicator, enqueued on the
// context's stream behind whatever produced the value (no host round trip).
int allreduce_min(lh_ctx* c, void* value_device_ft) {
    if (!c->comm) return LH_OK;
    Range r_("lh:allreduce_min");
    const ncclResult_t r = ncclAllReduce(value_device_ft, value_device_ft, 1,
                                         c->cfg.dtype == LH_F64 ? ncclDouble : ncclFloat, ncclMin, c->comm, c->stream);
    if (r != ncclSuccess) return fail(c, LH_ENODEVICE, "ncclAllReduce(min) failed: %s", ncclGetErrorString(r));
    return LH_OK;
}

// Uniform mesh of domain.jl:58-69: faces are the correctly rounded
// zmin + k L / n (Julia ranges step in twice precision), centres are face
// midpoints in FT, bottom first (coupled.jl:198).
template <typename FT>
void make_grid(double zmin, double zmax, int n, std::vector<FT>& zc) {
    zc.resize(n);
    FT lo = FT(zmin), hi = FT(zmax), prev = lo;
    for (int k = 1; k <= n; ++k) {
        long double x = (long double)lo + ((long double)hi - (long double)lo) * k / n;
        FT f = (k == n) ? hi : FT(x);
        zc[k - 1] = (prev + f) / FT(2);
        prev = f;
    }
}

template <typename FT> FT host_pow(FT x, FT y);
template <> double host_pow<double>(double x, double y) { return std::pow(x, y); }
template <> float host_pow<float>(float x, float y) { return powf(x, y); }

// Build the kernel argument, rounding to FT where the Julia constructors do.
// Levels per segment of the level-segmented launch (rhs_kernel, CFG::SEG), 0 = one lane
// marches the whole column.  Below ~2.6e5 lanes the launch time is the latency of one
// lane's march, so the column is cut into as many segments as it takes to fill the chip,
// each at least 4 levels long (every segment evaluates two extra cells).
int segment_length(const lh_ctx* c) {
    const int nlev = c->cfg.nlev;
    if (c->tune.seg < 0) return 0;
    if (c->tune.seg > 0) return c->tune.seg < nlev ? c->tune.seg : 0;
    const int64_t lanes = c->cfg.dtype == LH_F64 ? c->cfg.ncols : (c->cfg.ncols + 1) / 2;
    const int64_t target = 262144; // 256 CUs x 4 SIMDs x 64 lanes x 4 waves
    if (lanes * 2 > target || nlev < 16) return 0;
    int64_t nseg = (target + lanes - 1) / lanes;
    if (nseg > nlev / 4) nseg = nlev / 4;
    if (nseg < 2) return 0;
    const int len = int((nlev + nseg - 1) / nseg);
    return len < nlev ? len : 0;
}

bool vg_fast_all(const lh_ctx* c);

template <typename FT>
DevParams<FT> make_params(const lh_ctx* c) {
    const HostParams& h = c->hp;
    DevParams<FT> P;
    memset(&P, 0, sizeof P);
    P.ncols = c->cfg.ncols;
    P.stride = c->stride;
    P.nlev = c->cfg.nlev;
    P.model = c->cfg.model;
    P.dz = (FT(c->cfg.zmax) - FT(c->cfg.zmin)) / FT(c->cfg.nlev);
    P.inv_dz = FT(1) / P.dz;
    P.half_inv_dz = FT(0.5) * P.inv_dz;
    P.cg2 = P.half_inv_dz * P.inv_dz;
    P.half_dz = P.dz / FT(2);
    P.zc = static_cast<const FT*>(c->d_zc);
    P.vg_n = FT(h.vg.n);
    P.vg_alpha = FT(h.vg.alpha);
    P.vg_theta_r = FT(h.vg.theta_r);
    P.vg_Ksat = FT(h.vg.Ksat);
    P.nu = FT(h.soil.nu);
    P.S_s = FT(h.soil.S_s);
    P.rho_c_ds = FT(h.soil.rho_c_ds);
    P.kappa_solid = FT(h.soil.kappa_solid);
    P.rho_p = FT(h.soil.rho_p);
    P.kappa_sat_unfrozen = FT(h.soil.kappa_sat_unfrozen);
    P.kappa_sat_frozen = FT(h.soil.kappa_sat_frozen);
    P.kappa_dry_parameter = FT(h.soil.kappa_dry_parameter);
    P.nu_ss_om = FT(h.soil.nu_ss_om);
    P.b = FT(h.soil.b);
    const FT om = FT(h.soil.nu_ss_om), q = FT(h.soil.nu_ss_quartz), g = FT(h.soil.nu_ss_gravel),
             a = FT(h.soil.a);
    P.kersten_exp_unfrozen = (FT(1) + om - a * q - g) / FT(2); // SoilHeatParameterizations.jl:165
    P.kersten_exp_frozen = FT(1) + om;                         // :171
    P.one_minus_om = FT(1) - om;                               // :169
    P.neg_b_log2e_sc = FT(-double(P.b) * 1.4426950408889634 * (sizeof(FT) == 8 ? double(EXP_TAB_N) : 1.0));
    P.l2_kappa_sat_unfrozen = FT(std::log2(double(P.kappa_sat_unfrozen)));
    P.l2_kappa_sat_frozen = FT(std::log2(double(P.kappa_sat_frozen)));
    // FT(cp_l(param_set) * _rho_l): Float64 product of the constant and the
    // already rounded density (SoilHeatParameterizations.jl:71-75)
    P.rho_i = FT(h.earth.rho_ice);
    const FT rho_l = FT(h.earth.rho_liq);
    P.rhocp_i = FT(h.earth.cp_i * double(P.rho_i));
    P.rhocp_l = FT(h.earth.cp_l * double(rho_l));
    P.T_ref = FT(h.earth.T_0);
    P.LH_f0 = FT(h.earth.LH_f0);
    P.k_air = FT(h.earth.K_therm);
    P.viscosity_kind = h.viscosity_kind;
    P.impedance_kind = h.impedance_kind;
    P.gamma = FT(h.gamma);
    P.T_ref_visc = FT(h.T_ref_visc);
    P.Omega = FT(h.Omega);
    // uniform column constants, host libm
    ColC<FT>& u = P.uc;
    u.nu = P.nu;
    u.S_s = P.S_s;
    u.theta_r = P.vg_theta_r;
    u.theta_lim = u.theta_r + Limits<FT>::eps();
    u.n = P.vg_n;
    u.inv_n = FT(1) / u.n;
    u.m = FT(1) - FT(1) / u.n;
    u.inv_m = FT(1) / u.m;
    u.alpha_pnn = host_pow<FT>(P.vg_alpha, -u.n);
    u.Ksat = P.vg_Ksat;
    u.cgw = P.cg2 * u.Ksat;
    u.inv_por = FT(1) / (u.nu - u.theta_r);
    u.inv_S_s = FT(1) / u.S_s;
    u.inv_nu = FT(1) / u.nu;
    u.log2_alpha = FT(std::log2(double(P.vg_alpha)));
    set_fast_vg(u, P.vg_alpha);
    u.l2_por = FT(0); // filled on the device by finish_colc (the device math policy's own log2)
    {   // exponent multipliers in the exp2 unit of the production math (lh_fastmath.hpp)
        const FT sc = sizeof(FT) == 8 ? FT(MathFast<double>::EXP2_SCALE) : FT(1);
        u.e_one = sc;
        u.e_inv_m = sc * u.inv_m;
        u.e_m = sc * u.m;
        u.e_inv_n = u.inv_n;
        u.e_log2_alpha = sc * u.log2_alpha;
    }
    if (!(u.nu > u.theta_r)) u.Ksat = u.cgw = u.inv_S_s = u.log2_alpha = u.e_log2_alpha = u.alpha_pnn = FT(NAN);
    {
        FT rho_b = (FT(1) - u.nu) * P.rho_p;
        FT num = (P.kappa_dry_parameter * P.kappa_solid - P.k_air) * rho_b + P.k_air * P.rho_p;
        FT den = P.rho_p - (FT(1) - P.kappa_dry_parameter) * rho_b;
        u.k_dry = num / den;
    }
    for (int i = 0; i < LH_PC_COUNT; ++i) P.pc[i] = static_cast<const FT*>(c->d_pc[i]);
    for (int f = 0; f < 2; ++f)
        for (int k = 0; k < 2; ++k) {
            P.bc_kind[f][k] = h.bc_kind[f][k];
            P.bc_value[f][k] = FT(h.bc_value[f][k]);
            P.bc_pc[f][k] = static_cast<const FT*>(c->d_bc_pc[f][k]);
        }
    if (h.atmos_on) // the top face is the prescribed atmosphere: a (per-column) flux once evaluated
        P.bc_kind[LH_FACE_TOP][LH_COMP_ENERGY] = P.bc_kind[LH_FACE_TOP][LH_COMP_HYDROLOGY] = LH_BC_FLUX;
    P.consistent_bottom_sign = h.consistent_bottom_sign;
    P.status = c->d_status;
    P.math_tab = c->d_math_tab;
    P.dt_out = nullptr;
    P.xcd_remap = 0;
    P.seg_len = segment_length(c);
    P.cs_cpb = c->tune.cpb;
    P.vg_fast_all = (c->tune.vgfast != 0 && vg_fast_all(c)) ? 1 : 0;
    return P;
}

template <typename FT>
AtmosParams<FT> make_atmos_params(const lh_ctx* c) {
    const lh_atmos_forcing& a = c->hp.atmos;
    AtmosParams<FT> A;
    memset(&A, 0, sizeof A);
    A.u_atm = FT(a.u_atm);
    A.theta_atm = FT(a.theta_atm);
    A.z_atm = FT(a.z_atm);
    A.theta_scale = FT(a.theta_scale);
    A.rho_a_sfc = FT(a.rho_a_sfc);
    A.q_atm = FT(a.q_atm);
    A.z_0m = FT(a.z_0m);
    A.z_0s = FT(a.z_0s);
    A.R_v = FT(a.R_v);
    A.R_d = FT(a.R_d);
    A.grav = FT(a.grav);
    A.cp_d = FT(a.cp_d);
    A.cp_v = FT(a.cp_v);
    A.T_triple = FT(a.T_triple);
    A.press_triple = FT(a.press_triple);
    A.von_karman = FT(a.von_karman);
    A.cp_l = FT(c->hp.earth.cp_l);
    A.T_0 = FT(c->hp.earth.T_0);
    A.rho_liq = FT(c->hp.earth.rho_liq);
    A.cp_v_d = a.cp_v;
    A.LH_v0_d = a.LH_v0;
    A.pc_u = static_cast<const FT*>(c->d_atm_pc[0]);
    A.pc_theta = static_cast<const FT*>(c->d_atm_pc[1]);
    A.pc_q = static_cast<const FT*>(c->d_atm_pc[2]);
    return A;
}

// "block=128,pf=2,nt=1,cpl=2": launch-shape overrides (lh_launch.hpp, Tune)
void parse_tune(Tune& tu, const char* t) {
    tu = Tune();
    int v;
    const char* q;
    if ((q = strstr(t, "cpb=")) && sscanf(q + 4, "%d", &v) == 1 && v >= 0 && v <= 16) tu.cpb = v;
    if ((q = strstr(t, "persist=")) && sscanf(q + 8, "%d", &v) == 1 && v >= 0 && v <= 2) tu.persist = v;
    if ((q = strstr(t, "graph=")) && sscanf(q + 6, "%d", &v) == 1 && (v == 0 || v == 1)) tu.graph = v;
    if ((q = strstr(t, "seg=")) && sscanf(q + 4, "%d", &v) == 1 && v >= -1 && v <= 4096) tu.seg = v;
    if ((q = strstr(t, "place_mem=")) && sscanf(q + 10, "%d", &v) == 1 && v >= 1 && v <= 90) tu.place_mem = v;
    if ((q = strstr(t, "zero=")) && sscanf(q + 5, "%d", &v) == 1 && (v == 0 || v == 1)) tu.zero = v;
    if ((q = strstr(t, "xcd=")) && sscanf(q + 4, "%d", &v) == 1 && (v == 0 || v == 1)) tu.xcd = v;
    if ((q = strstr(t, "vgfast=")) && sscanf(q + 7, "%d", &v) == 1 && (v == 0 || v == 1)) tu.vgfast = v;
    if ((q = strstr(t, "block=")) && sscanf(q + 6, "%d", &v) == 1 && v >= 64 && v <= 1024 && v % 64 == 0) tu.block = v;
    if ((q = strstr(t, "cpl=")) && sscanf(q + 4, "%d", &v) == 1) tu.cpl = v;
    if ((q = strstr(t, "pf=")) && sscanf(q + 3, "%d", &v) == 1) tu.pf = v;
    if ((q = strstr(t, "nt=")) && sscanf(q + 3, "%d", &v) == 1) tu.nt = v;
    if ((q = strstr(t, "arena=")) && sscanf(q + 6, "%d", &v) == 1 && v >= 1 && v <= 64) tu.arena = v;
    if ((q = strstr(t, "rowpad=")) && sscanf(q + 7, "%d", &v) == 1 && v >= 0 && v % 2 == 0) tu.rowpad = v;
    for (q = strstr(t, "pad="); q; q = strstr(q + 4, "pad=")) // not the tail of "rowpad="
        if ((q == t || q[-1] == ',' || q[-1] == ' ') && sscanf(q + 4, "%d", &v) == 1 && v >= 0 && v % 256 == 0) tu.pad = v;
}

// Whether EVERY column may take the integer-exponent 2^(.) in its water closures (ColC::vg_fast): decided
// on the host from the scalar parameters and the ranges of the per-column arrays, conservatively (an
// interval test; every comparison is false for a NaN), because the choice selects the kernel
// instantiation (rhs_kernel VGF).
bool vg_fast_all(const lh_ctx* c) {
    const HostParams& h = c->hp;
    auto lo = [&](int id, double scalar) { return c->d_pc[id] ? c->pc_lo[id] : scalar; };
    auto hi = [&](int id, double scalar) { return c->d_pc[id] ? c->pc_hi[id] : scalar; };
    const double n_lo = lo(LH_PC_VG_N, h.vg.n), n_hi = hi(LH_PC_VG_N, h.vg.n);
    const double a_lo = lo(LH_PC_VG_ALPHA, h.vg.alpha), a_hi = hi(LH_PC_VG_ALPHA, h.vg.alpha);
    const double t_lo = lo(LH_PC_VG_THETA_R, h.vg.theta_r), t_hi = hi(LH_PC_VG_THETA_R, h.vg.theta_r);
    const double nu_lo = lo(LH_PC_NU, h.soil.nu), nu_hi = hi(LH_PC_NU, h.soil.nu);
    const double m_lo = 1.0 - 1.0 / n_lo; // (Float32 rounding of the parameters moves m by 1e-7: the margin below covers it)
    return n_lo > 1.0 && n_hi < 1e30 && m_lo >= LH_VG_FAST_MIN_M * 1.001 && a_lo > 1.01e-20 && a_hi < 0.99e20 &&
           nu_lo - t_hi >= 1.01e-5 && nu_hi - t_lo <= 1.0;
}

bool any_percol(const lh_ctx* c) {
    for (int i = 0; i < LH_PC_COUNT; ++i)
        if (c->d_pc[i]) return true;
    return false;
}

bool model_water(int m) { return m != LH_MODEL_HEAT; }
bool model_heat(int m) { return m != LH_MODEL_RICHARDS; }

uint32_t prognostic_mask(int model) {
    switch (model) {
        case LH_MODEL_RICHARDS: return LH_MASK(LH_VAR_VARTHETA_L) | LH_MASK(LH_VAR_THETA_I);
        case LH_MODEL_HEAT: return LH_MASK(LH_VAR_RHOE_INT);
        default: return LH_MASK(LH_VAR_VARTHETA_L) | LH_MASK(LH_VAR_THETA_I) | LH_MASK(LH_VAR_RHOE_INT);
    }
}

uint32_t aux_mask(const lh_ctx* c) {
    if (c->cfg.model == LH_MODEL_HEAT) return LH_MASK(LH_VAR_VARTHETA_L) | LH_MASK(LH_VAR_THETA_I);
    if (c->cfg.model == LH_MODEL_RICHARDS && c->hp.viscosity_kind != LH_FACTOR_NONE)
        return LH_MASK(LH_VAR_T);
    return 0;
}

// The combinations for which the reference has a method; anything else raises
// there (MethodError / SetValue(nothing)) and is LH_EMODEL here.
int validate_model(lh_ctx* c) {
    const HostParams& h = c->hp;
    const int m = c->cfg.model;
    if (h.atmos_on && m != LH_MODEL_COUPLED) // compute_turbulent_surface_fluxes has one method (:553-560)
        return fail(c, LH_EMODEL, "PrescribedAtmosForcing needs SoilEnergyModel + SoilHydrologyModel (no method for this model)");
    for (int f = 0; f < 2; ++f) {
        if (f == LH_FACE_TOP && h.atmos_on) continue; // the whole top face is the prescribed atmosphere
        const int ke = h.bc_kind[f][LH_COMP_ENERGY], kh = h.bc_kind[f][LH_COMP_HYDROLOGY];
        const char* fn = f == LH_FACE_TOP ? "top" : "bottom";
        if (ke == LH_BC_FREE_DRAINAGE)
            return fail(c, LH_EMODEL, "FreeDrainage is a hydrology boundary condition (%s energy)", fn);
        if (model_heat(m)) {
            if (ke == LH_BC_NONE)
                return fail(c, LH_EMODEL, "SoilEnergyModel needs an energy boundary condition at the %s face (got NoBC)", fn);
        } else if (ke == LH_BC_DIRICHLET) {
            return fail(c, LH_EMODEL, "Dirichlet energy BC has no method for PrescribedTemperatureModel (%s)", fn);
        }
        if (model_water(m)) {
            if (kh == LH_BC_NONE)
                return fail(c, LH_EMODEL, "SoilHydrologyModel needs a hydrology boundary condition at the %s face (got NoBC)", fn);
        } else if (kh == LH_BC_DIRICHLET || kh == LH_BC_FREE_DRAINAGE) {
            return fail(c, LH_EMODEL, "hydrology BC of this kind has no method for PrescribedHydrologyModel (%s)", fn);
        }
    }
    if (model_heat(m) && !h.earth_set)
        return fail(c, LH_EINVAL, "earth parameters (lh_set_earth_params) are required by the energy model");
    return LH_OK;
}

int check_state(lh_ctx* c, const lh_state* s, uint32_t need, const char* what) {
    if (need == 0) return LH_OK;
    if (!s) return fail(c, LH_ESTATE, "%s state is NULL but the model reads it", what);
    if (s->ctx != c) return fail(c, LH_EINVAL, "%s state belongs to another context", what);
    if ((s->mask & need) != need)
        return fail(c, LH_ESTATE, "%s state lacks a required variable (has mask 0x%x, needs 0x%x)", what, s->mask, need);
    return LH_OK;
}

template <typename FT>
Planes<FT> planes_of(const lh_state* s) {
    Planes<FT> p;
    for (int i = 0; i < LH_NVARS; ++i) p.v[i] = s ? static_cast<FT*>(s->plane[i]) : nullptr;
    return p;
}

// the zero bit of a plane the library has just filled with zeros (never for an exposed plane)
void mark_zero(lh_state* s, int var) {
    if (!(s->exposed_mask & (1u << var))) s->zero_mask |= 1u << var;
}
// a plane is about to be (or may have been) overwritten with arbitrary values
void mark_written(lh_state* s, uint32_t vars) {
    s->zero_mask &= ~vars;
    s->profile_mask &= ~vars;
    s->stale_mask &= ~vars;
}

// Give the planes of `vars` the values of their level-uniform profiles (one broadcast launch per
// stale plane); the profile stays valid.  Every consumer that addresses planes calls this first.
int materialize(lh_ctx* c, const lh_state* cs, uint32_t vars) {
    lh_state* s = const_cast<lh_state*>(cs);
    if (!s) return LH_OK;
    const uint32_t todo = s->stale_mask & s->profile_mask & vars;
    for (int i = 0; i < LH_NVARS && todo; ++i)
        if (todo >> i & 1u) {
            if (c->cfg.dtype == LH_F64)
                launch_broadcast_profile<double>(static_cast<double*>(s->plane[i]), static_cast<const double*>(s->profile[i]),
                                                 c->cfg.ncols, c->stride, c->cfg.nlev, c->stream);
            else
                launch_broadcast_profile<float>(static_cast<float*>(s->plane[i]), static_cast<const float*>(s->profile[i]),
                                                c->cfg.ncols, c->stride, c->cfg.nlev, c->stream);
            LH_HIP(c, hipGetLastError());
        }
    s->stale_mask &= ~todo;
    return LH_OK;
}

// the variables of Ya a column launch takes from a level-uniform profile instead of a plane
uint32_t aux_profile_vars(const lh_ctx* c, const lh_state* aux) {
    if (!aux) return 0;
    uint32_t m = 0;
    if (c->cfg.model == LH_MODEL_HEAT) m = LH_MASK(LH_VAR_VARTHETA_L) | LH_MASK(LH_VAR_THETA_I);
    else if (c->cfg.model == LH_MODEL_RICHARDS) m = LH_MASK(LH_VAR_T);
    return m & aux->profile_mask;
}
template <typename FT>
void set_aux_profiles(const lh_ctx* c, const lh_state* aux, DevParams<FT>& P) {
    const uint32_t m = aux_profile_vars(c, aux);
    for (int i = 0; i < LH_NVARS; ++i) P.aux_prof[i] = (m >> i & 1u) ? static_cast<const FT*>(aux->profile[i]) : nullptr;
}

template <typename FT>
int do_rhs(lh_ctx* c, const lh_state* in, const lh_state* aux, const lh_state* base, lh_state* out,
           double dt, int mode, const double* bc_override, const void* dt_device = nullptr,
           void* dt_out = nullptr, bool unsegmented = false) {
    static const char* const range_names[6] = {"lh:rhs", "lh:stage1", "lh:stage2", "lh:stage3", "lh:rhs_stable_dt", "lh:stage2_from_k1"};
    Range r_(range_names[mode >= 0 && mode < 6 ? mode : 0]);
    DevParams<FT> P = make_params<FT>(c);
    if (unsegmented) P.seg_len = 0; // an in-place stage must not be level-segmented
    P.dt_out = dt_out;
    {   // level-uniform variables: Ya's prescribed profiles go to the kernel as they are, the rest become planes
        int rc;
        if ((rc = materialize(c, in, ~0u)) || (rc = materialize(c, base, ~0u)) || (rc = materialize(c, out, ~0u)) ||
            (rc = materialize(c, aux, ~aux_profile_vars(c, aux))))
            return rc;
        set_aux_profiles<FT>(c, aux, P);
    }
    P.xcd_remap = c->tune.xcd;
    if (bc_override)
        for (int f = 0; f < 2; ++f)
            for (int k = 0; k < 2; ++k) P.bc_value[f][k] = FT(bc_override[f * 2 + k]);
    const bool factors = c->hp.viscosity_kind != LH_FACTOR_NONE || c->hp.impedance_kind != LH_FACTOR_NONE;
    const bool tend = mode == 0 || mode == 4;
    const uint32_t ti_bit = LH_MASK(LH_VAR_THETA_I);
    if (c->hp.atmos_on) {
        // boundary_fluxes(X, bc::PrescribedAtmosForcing, :top, ...) (:516-533): the surface fluxes of
        // the state this evaluation reads, from its top cells, into the per-column flux arrays the
        // column kernel takes as VerticalFlux values at the top face
        const size_t top = size_t(c->cfg.nlev - 1) * size_t(c->stride);
        const lh_state* ti_state = tend ? in : base; // fused stages keep theta_i in the base state
        const FT* vl_top = static_cast<const FT*>(in->plane[LH_VAR_VARTHETA_L]) + top;
        const FT* ti_top = static_cast<const FT*>(ti_state->plane[LH_VAR_THETA_I]) + top;
        const FT* re_top = static_cast<const FT*>(in->plane[LH_VAR_RHOE_INT]) + top;
        launch_atmos_flux<FT>(P, make_atmos_params<FT>(c), c->cfg.ncols, true, any_percol(c), vl_top, ti_top, re_top,
                              static_cast<FT*>(c->d_atm_flux[0]), static_cast<FT*>(c->d_atm_flux[1]), c->stream);
        P.bc_kind[LH_FACE_TOP][LH_COMP_ENERGY] = P.bc_kind[LH_FACE_TOP][LH_COMP_HYDROLOGY] = LH_BC_FLUX;
        P.bc_pc[LH_FACE_TOP][LH_COMP_ENERGY] = static_cast<const FT*>(c->d_atm_flux[0]);
        P.bc_pc[LH_FACE_TOP][LH_COMP_HYDROLOGY] = static_cast<const FT*>(c->d_atm_flux[1]);
    }
    // the state theta_i is read from: Ya (HEAT), Y (tendency), the step's base state (fused stages)
    const lh_state* ti_src = c->cfg.model == LH_MODEL_HEAT ? aux : (tend ? in : base);
    const bool noice = c->tune.zero != 0 && ti_src && (ti_src->zero_mask & ti_bit);
    const bool water = model_water(c->cfg.model);
    // d theta_i = 0 (right_hand_side.jl:182, :359): no kernel stores it -- the tendency state's
    // theta_i plane is cleared here unless it is known to hold zeros already (normally once per
    // state; LH_TUNE zero=0 clears it at every launch, the traffic of a kernel that stores it)
    if (tend && water && (c->tune.zero == 0 || !(out->zero_mask & ti_bit))) {
        const size_t bytes = size_t(c->cfg.nlev) * size_t(c->stride) * c->esize;
        LH_HIP(c, hipMemsetAsync(out->plane[LH_VAR_THETA_I], 0, bytes, c->stream));
        mark_written(out, ti_bit);
        mark_zero(out, LH_VAR_THETA_I);
    }
    launch_rhs<FT>(P, planes_of<FT>(in), planes_of<FT>(aux), planes_of<FT>(base), planes_of<FT>(out),
                   FT(dt), static_cast<const FT*>(dt_device), mode, factors, any_percol(c), noice, c->math, c->tune, c->stream);
    LH_HIP(c, hipGetLastError());
    // what the launch wrote: vartheta_l / rhoe_int values
    if (water) mark_written(out, LH_MASK(LH_VAR_VARTHETA_L));
    if (model_heat(c->cfg.model)) mark_written(out, LH_MASK(LH_VAR_RHOE_INT));
    return LH_OK;
}

constexpr int ARENA_SLOTS = 8;

void* plane_alloc(lh_ctx* c, size_t bytes) {
    const size_t pad = c->tune.pad >= 0 ? size_t(c->tune.pad) : LH_PLANE_STAGGER;
    // slot pitch: plane size rounded to 2 MiB, plus the stagger (LH_TUNE pad=, a multiple of 256 B)
    const size_t slot = (((bytes + (size_t(2) << 20) - 1) >> 21) << 21) + pad;
    for (auto& a : c->arenas)
        if (a.slot_bytes == slot)
            for (int k = 0; k < a.nslots; ++k)
                if (!a.used[k]) {
                    a.used[k] = 1;
                    return a.base + size_t(k) * slot;
                }
    PlaneArena a;
    a.slot_bytes = slot;
    a.nslots = c->tune.arena > 0 ? c->tune.arena : ARENA_SLOTS;
    // small ensembles: one slot per arena would waste nothing, but keep it uniform
    if (hipMalloc(reinterpret_cast<void**>(&a.base), slot * size_t(a.nslots)) != hipSuccess) {
        (void)hipGetLastError();
        a.nslots = 1; // memory is tight: fall back to a single-plane arena
        if (hipMalloc(reinterpret_cast<void**>(&a.base), slot) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    a.used.assign(a.nslots, 0);
    a.used[0] = 1;
    c->arenas.push_back(a);
    return c->arenas.back().base;
}

void plane_free(lh_ctx* c, void* p) {
    for (size_t i = 0; i < c->arenas.size(); ++i) {
        PlaneArena& a = c->arenas[i];
        char* q = static_cast<char*>(p);
        if (q >= a.base && q < a.base + a.slot_bytes * size_t(a.nslots)) {
            a.used[(q - a.base) / a.slot_bytes] = 0;
            bool any = false;
            for (char u : a.used) any = any || u;
            if (!any) { // release an arena once its last plane is gone
                (void)hipFree(a.base);
                c->arenas.erase(c->arenas.begin() + i);
            }
            return;
        }
    }
}

int state_alloc(lh_ctx* c, uint32_t mask, lh_state** out) {
    lh_state* s = new (std::nothrow) lh_state();
    if (!s) return fail(c, LH_ENOMEM, "out of host memory");
    s->ctx = c;
    s->mask = mask;
    s->zero_mask = mask; // every plane is cleared below
    s->exposed_mask = s->profile_mask = s->stale_mask = 0;
    for (int i = 0; i < LH_NVARS; ++i) s->profile[i] = nullptr;
    const size_t bytes = size_t(c->cfg.nlev) * size_t(c->stride) * c->esize;
    for (int i = 0; i < LH_NVARS; ++i) s->plane[i] = nullptr;
    for (int i = 0; i < LH_NVARS; ++i) {
        if (mask & (1u << i)) {
            s->plane[i] = plane_alloc(c, bytes);
            if (!s->plane[i]) {
                for (int j = 0; j < i; ++j)
                    if (s->plane[j]) plane_free(c, s->plane[j]);
                delete s;
                return fail(c, LH_ENOMEM, "device allocation of a %zu-byte plane failed", bytes);
            }
            hipError_t e = hipMemsetAsync(s->plane[i], 0, bytes, c->stream);
            if (e != hipSuccess) {
                for (int j = 0; j <= i; ++j)
                    if (s->plane[j]) plane_free(c, s->plane[j]);
                delete s;
                return fail(c, LH_ENODEVICE, "hipMemsetAsync failed: %s", hipGetErrorString(e));
            }
        }
    }
    c->states.push_back(s);
    *out = s;
    return LH_OK;
}

void state_free(lh_ctx* c, lh_state* s) {
    for (int i = 0; i < LH_NVARS; ++i) {
        if (s->plane[i]) plane_free(c, s->plane[i]);
        if (s->profile[i]) (void)hipFree(s->profile[i]);
    }
    for (size_t i = 0; i < c->states.size(); ++i)
        if (c->states[i] == s) {
            c->states.erase(c->states.begin() + i);
            break;
        }
    delete s;
}

// The persistent column stepper keeps the state in registers over all steps of a call: 0.56 ms per
// step on 1e6 x 64 Float64 columns against 0.82 ms for three fused-stage launches, and 2-3 us per
// step for a single column.  Getting the level-fastest registers from and to the column-fastest
// planes costs about one step per call at that size, so large ensembles take it from 3 steps per call
// on; small ones (<= 2^20 threads) always.  Not for columns with more levels than a workgroup has threads.
// LH_TUNE=persist=0: never (fused-stage launches), persist=2: always.
// A Dirichlet face needs the closures of the face state.  rhs_kernel evaluates them for 64 columns
// at once; in the stepper the one lane next to the face does, with its whole wave waiting: a second
// closure pass per stage.  The wave stepper (<= 128 levels) evaluates them ONCE per call when the
// boundary values are constant over the call and nothing else the face state reads moves (the rule
// of face_state_is_static, lh_closures.hpp, restated here); otherwise large ensembles with a
// Dirichlet face stay with the fused-stage launches.
static bool dirichlet_faces_are_hoisted(const lh_ctx* c, bool bcv) {
    const bool water = c->cfg.model != LH_MODEL_HEAT, heat = c->cfg.model != LH_MODEL_RICHARDS;
    bool any = false, all_static = true;
    for (int f = 0; f < 2; ++f) {
        const int kh = c->hp.bc_kind[f][LH_COMP_HYDROLOGY], ke = c->hp.bc_kind[f][LH_COMP_ENERGY];
        const bool needs_w = water && kh == LH_BC_DIRICHLET, needs_k = heat && ke == LH_BC_DIRICHLET;
        any = any || needs_w || needs_k;
        if (needs_w && c->hp.viscosity_kind != LH_FACTOR_NONE && heat && ke != LH_BC_DIRICHLET) all_static = false;
        if (needs_k && water && kh != LH_BC_DIRICHLET) all_static = false;
    }
    if (!any) return true;
    // (measured, 1e6 columns, ms per step, stepper vs fused stages: Richards with a viscosity factor
    // 0.80 vs 1.05; the coupled model with both conductivity factors and ice, where the closures and not
    // the planes' traffic bound either engine, 0.749 vs 0.753 in Float32 and 2.03 vs 1.97 in Float64)
    const bool factors = c->hp.viscosity_kind != LH_FACTOR_NONE || c->hp.impedance_kind != LH_FACTOR_NONE;
    if (c->cfg.model == LH_MODEL_COUPLED && factors && c->cfg.dtype == LH_F64) return false;
    return all_static && !bcv && c->cfg.nlev <= 128;
}

bool use_column_stepper(const lh_ctx* c, int64_t nsteps, bool bcv = false) {
    if (c->tune.persist == 0 || c->cfg.nlev > 1024) return false;
    if (c->hp.atmos_on) return false; // the surface fluxes are re-evaluated before every stage launch
    if (c->tune.persist == 2) return true;
    const int64_t threads = c->cfg.ncols * int64_t((c->cfg.nlev + 63) / 64 * 64);
    if (threads <= (int64_t(1) << 20)) return true;
    if (!dirichlet_faces_are_hoisted(c, bcv)) return false;
    return nsteps >= 3;
}

int run_column_stepper(lh_ctx* c, lh_state* Y, const lh_state* Ya, double dt, const void* dt_device,
                       int64_t nsteps, const double* bcv) {
    if (nsteps <= 0) return LH_OK;
    Range r_("lh:column_stepper");
    void* d_bcv = nullptr;
    if (bcv) { // [nsteps][3][2][2] doubles -> FT on the device
        const size_t nv = size_t(nsteps) * 12;
        std::vector<char> tmp(nv * c->esize);
        for (size_t k = 0; k < nv; ++k) {
            if (c->cfg.dtype == LH_F64) reinterpret_cast<double*>(tmp.data())[k] = bcv[k];
            else reinterpret_cast<float*>(tmp.data())[k] = float(bcv[k]);
        }
        LH_HIP(c, hipMalloc(&d_bcv, nv * c->esize));
        hipError_t e = hipMemcpyAsync(d_bcv, tmp.data(), nv * c->esize, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream); // tmp is pageable host memory
        if (e != hipSuccess) {
            (void)hipFree(d_bcv);
            return fail(c, LH_ENODEVICE, "boundary-value upload failed: %s", hipGetErrorString(e));
        }
    }
    const bool factors = c->hp.viscosity_kind != LH_FACTOR_NONE || c->hp.impedance_kind != LH_FACTOR_NONE;
    const lh_state* ti_src = c->cfg.model == LH_MODEL_HEAT ? Ya : Y;
    const bool noice = c->tune.zero != 0 && !factors && ti_src && (ti_src->zero_mask & LH_MASK(LH_VAR_THETA_I));
    {
        int rc;
        if ((rc = materialize(c, Y, ~0u)) || (rc = materialize(c, Ya, ~aux_profile_vars(c, Ya)))) return rc;
    }
    if (c->cfg.dtype == LH_F64) {
        DevParams<double> P = make_params<double>(c);
        set_aux_profiles<double>(c, Ya, P);
        launch_column_stepper<double>(P, planes_of<double>(Y), planes_of<double>(Ya), dt,
                                      static_cast<const double*>(dt_device), nsteps,
                                      static_cast<const double*>(d_bcv), factors, any_percol(c), noice, c->stream);
    } else {
        DevParams<float> P = make_params<float>(c);
        set_aux_profiles<float>(c, Ya, P);
        launch_column_stepper<float>(P, planes_of<float>(Y), planes_of<float>(Ya), float(dt),
                                     static_cast<const float*>(dt_device), nsteps,
                                     static_cast<const float*>(d_bcv), factors, any_percol(c), noice, c->stream);
    }
    mark_written(Y, LH_MASK(LH_VAR_VARTHETA_L) | LH_MASK(LH_VAR_RHOE_INT));
    hipError_t e = hipGetLastError();
    if (d_bcv) { // the launch reads it: wait before releasing
        const hipError_t e2 = hipStreamSynchronize(c->stream);
        (void)hipFree(d_bcv);
        if (e == hipSuccess) e = e2;
    }
    if (e != hipSuccess) return fail(c, LH_ENODEVICE, "column stepper launch failed: %s", hipGetErrorString(e));
    return LH_OK;
}

// A level-segmented stage reads cells its neighbour segments write: it needs a target
// other than its input, i.e. a second stage state.
int second_stage_state(lh_ctx* c, lh_state** U2) {
    if (segment_length(c) <= 0) return LH_OK; // in place
    if (!c->scratch_u2) {
        const uint32_t pm = prognostic_mask(c->cfg.model);
        int rc = state_alloc(c, pm & ~LH_MASK(LH_VAR_THETA_I), &c->scratch_u2);
        if (rc) return rc;
    }
    *U2 = c->scratch_u2;
    return LH_OK;
}

int upload_percol(lh_ctx* c, void** slot, const double* host) {
    if (*slot) {
        LH_HIP(c, hipStreamSynchronize(c->stream));
        (void)hipFree(*slot);
        *slot = nullptr;
    }
    if (!host) return LH_OK;
    const int64_t n = c->cfg.ncols;
    double* tmp = nullptr;
    LH_HIP(c, hipMalloc(&tmp, size_t(n) * sizeof(double)));
    hipError_t e = hipMalloc(slot, size_t(n) * c->esize);
    if (e != hipSuccess) {
        (void)hipFree(tmp);
        return fail(c, LH_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    LH_HIP(c, hipMemcpyAsync(tmp, host, size_t(n) * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (c->cfg.dtype == LH_F64) launch_convert<double>(static_cast<double*>(*slot), tmp, n, c->stream);
    else launch_convert<float>(static_cast<float*>(*slot), tmp, n, c->stream);
    LH_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(tmp);
    return LH_OK;
}

} // namespace

// Placement tuning.  On MI355X the speed of the column kernel depends on WHERE
// the planes it streams together sit relative to each other in HBM: the same
// launch on the same data runs at one of a few discrete rates (e.g. 0.339 /
// 0.378 / 0.393 ms on the 1e6 x 64 Float64 Richards case) depending on which
// arena slots hold the written state, reproducibly for a given set of slots and
// with no usable rule in the virtual addresses (profiles/round1_placement_notes.txt).
// So the library measures: it times the real launch with the written state in
// up to `max_candidates` different slot sets and keeps the fastest.
namespace {

// keep_contents: every candidate receives a copy of the target's planes (an INPUT
// of the launch is being moved); otherwise candidates start zeroed (an output).
// plane_mask: which planes of the target move (the others stay where they are).
template <typename RUN>
int tune_state_planes(lh_ctx* c, lh_state* target, uint32_t plane_mask, int max_candidates,
                      bool keep_contents, RUN&& run, float* ms_before, float* ms_after,
                      float goal_ms = 0.0f) {
    struct Cand {
        void* plane[LH_NVARS];
        float ms;
    };
    const size_t bytes = size_t(c->cfg.nlev) * size_t(c->stride) * c->esize;
    int nplanes = 0;
    for (int i = 0; i < LH_NVARS; ++i) nplanes += (target->plane[i] != nullptr) && (plane_mask >> i & 1u);
    if (!nplanes) return LH_OK;
    int K = max_candidates > 0 ? max_candidates : 6;
    if (K > 64) K = 64;
    // every candidate stays allocated until the choice is made: keep well inside free memory
    size_t free_b = 0, total_b = 0;
    LH_HIP(c, hipMemGetInfo(&free_b, &total_b));
    const size_t per_cand = (bytes + (size_t(4) << 20)) * size_t(nplanes);
    const size_t arena_b = (bytes + (size_t(4) << 20)) * ARENA_SLOTS;
    // transient memory of the search: at most LH_TUNE place_mem=PCT percent of the free device
    // memory (default 25)
    const size_t budget = free_b / 100 * size_t(c->tune.place_mem > 0 ? c->tune.place_mem : 25);
    while (K > 1 && size_t(K - 1) * per_cand + 2 * arena_b > budget) --K;

    // the trial launches must not leave their own mark in the status word
    uint32_t status_saved = 0;
    LH_HIP(c, hipMemcpyAsync(&status_saved, c->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    LH_HIP(c, hipStreamSynchronize(c->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    LH_HIP(c, hipEventCreate(&e0));
    {
        const hipError_t ee = hipEventCreate(&e1);
        if (ee != hipSuccess) {
            (void)hipEventDestroy(e0);
            return fail(c, LH_ENODEVICE, "hipEventCreate failed: %s", hipGetErrorString(ee));
        }
    }
    int rc = LH_OK;
    auto timed = [&](float& ms) -> int {
        constexpr int REPS = 5;
        for (int r = 0; r < 2 && !rc; ++r) rc = run();
        if (rc) return rc;
        if (hipEventRecord(e0, c->stream) != hipSuccess) return LH_ENODEVICE;
        for (int r = 0; r < REPS && !rc; ++r) rc = run();
        if (rc) return rc;
        if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
            return LH_ENODEVICE;
        ms /= REPS;
        return LH_OK;
    };
    std::vector<Cand> cands;
    Cand cur;
    for (int i = 0; i < LH_NVARS; ++i) cur.plane[i] = target->plane[i];
    cur.ms = 0;
    cands.push_back(cur);
    for (int r = 0; r < 20 && !rc; ++r) rc = run(); // clocks up before anything is compared
    for (int k = 1; k < K && !rc; ++k) {
        Cand nc;
        bool ok = true;
        for (int i = 0; i < LH_NVARS; ++i) {
            nc.plane[i] = nullptr;
            if (!(plane_mask >> i & 1u)) {
                nc.plane[i] = cur.plane[i]; // shared with every candidate, never freed here
                continue;
            }
            if (ok && target->plane[i]) {
                nc.plane[i] = plane_alloc(c, bytes);
                if (!nc.plane[i]) ok = false;
                else if ((keep_contents ? hipMemcpyAsync(nc.plane[i], cands[0].plane[i], bytes,
                                                         hipMemcpyDeviceToDevice, c->stream)
                                        : hipMemsetAsync(nc.plane[i], 0, bytes, c->stream)) != hipSuccess)
                    ok = false;
            }
        }
        if (!ok) { // out of memory: work with the candidates there are
            (void)hipGetLastError();
            for (int i = 0; i < LH_NVARS; ++i)
                if (nc.plane[i] && (plane_mask >> i & 1u)) plane_free(c, nc.plane[i]);
            break;
        }
        nc.ms = 0;
        cands.push_back(nc);
        if (goal_ms > 0) { // a long search stops as soon as a candidate reaches the goal
            for (int i = 0; i < LH_NVARS; ++i) target->plane[i] = nc.plane[i];
            float ms = 0;
            if ((rc = timed(ms))) break;
            if (ms <= goal_ms) break;
        }
    }
    // two interleaved passes, the minimum counts
    for (int pass = 0; pass < 2 && !rc; ++pass)
        for (auto& cd : cands) {
            for (int i = 0; i < LH_NVARS; ++i) target->plane[i] = cd.plane[i];
            float ms = 0;
            if ((rc = timed(ms))) break;
            cd.ms = (pass == 0 || ms < cd.ms) ? ms : cd.ms;
        }
    size_t best = 0;
    if (!rc)
        for (size_t k = 1; k < cands.size(); ++k)
            if (cands[k].ms < cands[best].ms * 0.985f) best = k; // move only for a real gain
    (void)hipStreamSynchronize(c->stream);
    for (size_t k = 0; k < cands.size(); ++k) {
        if (k == best) continue;
        for (int i = 0; i < LH_NVARS; ++i)
            if (cands[k].plane[i] && (plane_mask >> i & 1u)) plane_free(c, cands[k].plane[i]);
    }
    for (int i = 0; i < LH_NVARS; ++i) target->plane[i] = cands[best].plane[i];
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipMemcpyAsync(c->d_status, &status_saved, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
    (void)hipStreamSynchronize(c->stream);
    if (rc) return rc;
    if (ms_before) *ms_before = cands[0].ms;
    if (ms_after) *ms_after = cands[best].ms;
    return LH_OK;
}

} // namespace

namespace {

template <typename FT>
int boundary_fluxes_impl(lh_ctx* c, const lh_state* Y, const lh_state* Ya, int32_t face, double* f_energy, double* f_water) {
    const int64_t n = c->cfg.ncols;
    FT* d_out = nullptr;
    LH_HIP(c, hipMalloc(reinterpret_cast<void**>(&d_out), size_t(n) * 2 * sizeof(FT)));
    DevParams<FT> P = make_params<FT>(c);
    if (c->hp.atmos_on && face == LH_FACE_TOP) {
        // boundary_fluxes(X, bc::PrescribedAtmosForcing, :top, ...) (:516-533), as do_rhs prepares it
        const size_t top = size_t(c->cfg.nlev - 1) * size_t(c->stride);
        launch_atmos_flux<FT>(P, make_atmos_params<FT>(c), n, true, any_percol(c),
                              static_cast<const FT*>(Y->plane[LH_VAR_VARTHETA_L]) + top,
                              static_cast<const FT*>(Y->plane[LH_VAR_THETA_I]) + top,
                              static_cast<const FT*>(Y->plane[LH_VAR_RHOE_INT]) + top,
                              static_cast<FT*>(c->d_atm_flux[0]), static_cast<FT*>(c->d_atm_flux[1]), c->stream);
        P.bc_kind[LH_FACE_TOP][LH_COMP_ENERGY] = P.bc_kind[LH_FACE_TOP][LH_COMP_HYDROLOGY] = LH_BC_FLUX;
        P.bc_pc[LH_FACE_TOP][LH_COMP_ENERGY] = static_cast<const FT*>(c->d_atm_flux[0]);
        P.bc_pc[LH_FACE_TOP][LH_COMP_HYDROLOGY] = static_cast<const FT*>(c->d_atm_flux[1]);
    }
    const bool factors = c->hp.viscosity_kind != LH_FACTOR_NONE || c->hp.impedance_kind != LH_FACTOR_NONE;
    launch_boundary_fluxes<FT>(P, planes_of<FT>(Y), planes_of<FT>(Ya), face, d_out, d_out + n, factors, any_percol(c),
                               c->math, c->stream);
    hipError_t e = hipGetLastError();
    std::vector<FT> h(size_t(n) * 2);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, h.size() * sizeof(FT), hipMemcpyDeviceToHost, c->stream);
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, LH_ENODEVICE, "lh_boundary_fluxes failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    for (int64_t i = 0; i < n; ++i) {
        if (f_energy) f_energy[i] = double(h[size_t(i)]);
        if (f_water) f_water[i] = double(h[size_t(n + i)]);
    }
    return LH_OK;
}

} // namespace

// ============================================================== C ABI

extern "C" {

int lh_version(void) { return LH_VERSION_MAJOR * 100 + LH_VERSION_MINOR; }

const char* lh_last_error(const lh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int lh_create(lh_ctx** out, const lh_config* cfg) {
    if (!out || !cfg) return fail(nullptr, LH_EINVAL, "lh_create: NULL argument");
    *out = nullptr;
    if (cfg->ncols < 1 || cfg->nlev < 1) return fail(nullptr, LH_EINVAL, "lh_create: ncols and nlev must be >= 1");
    if (cfg->ncols > (int64_t(1) << 28)) return fail(nullptr, LH_EINVAL, "lh_create: more than 2^28 columns per context is not supported (a plane row must stay below 4 GiB); partition the ensemble");
    if (cfg->nlev > 4096) return fail(nullptr, LH_EINVAL, "lh_create: nlev > 4096 is not supported (level coordinates are staged in LDS)");
    if (!(cfg->zmin < cfg->zmax)) return fail(nullptr, LH_EINVAL, "lh_create: zlim[1] < zlim[2] required (domain.jl:30)");
    if (cfg->dtype != LH_F32 && cfg->dtype != LH_F64) return fail(nullptr, LH_EINVAL, "lh_create: dtype must be LH_F32 or LH_F64");
    if (cfg->model < LH_MODEL_RICHARDS || cfg->model > LH_MODEL_COUPLED) return fail(nullptr, LH_EINVAL, "lh_create: unknown model");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, LH_ENODEVICE, "lh_create: no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    int dev = cfg->device;
    if (dev < 0) {
        e = hipGetDevice(&dev);
        if (e != hipSuccess) return fail(nullptr, LH_ENODEVICE, "hipGetDevice failed: %s", hipGetErrorString(e));
    }
    if (dev >= ndev) return fail(nullptr, LH_EINVAL, "lh_create: device %d out of range (%d devices)", dev, ndev);
    e = hipSetDevice(dev);
    if (e != hipSuccess) return fail(nullptr, LH_ENODEVICE, "hipSetDevice(%d) failed: %s", dev, hipGetErrorString(e));

    lh_ctx* c = new (std::nothrow) lh_ctx();
    if (!c) return fail(nullptr, LH_ENOMEM, "out of host memory");
    c->cfg = *cfg;
    c->device = dev;
    c->esize = cfg->dtype == LH_F64 ? 8 : 4;
    // Plane row length: whole 512-B units, and an ODD number of them.  The column
    // kernel has every level of a plane in flight at once (waves drift apart), so
    // a level stride with a large power-of-two factor piles those streams onto
    // the same HBM channels/banks: 2^20 columns ran at 59.5 % of peak with the
    // natural 8-MiB stride and at 65-66 % with one extra unit (profiles/
    // round1_placement_notes.txt).
    {
        const int64_t unit = 512 / int64_t(c->esize); // elements per 512 B
        int64_t units = (cfg->ncols + unit - 1) / unit;
        if (units % 2 == 0) ++units;
        c->stride = units * unit;
    }
    // reference defaults: loam vanGenuchten, default SoilParams
    c->hp.vg = lh_vg_params{1.56, 3.6, 0.0, 2.9e-7};
    c->hp.soil = lh_soil_params{0.43, 1e-3, 0.0, 0.0, 0.41, 2700.0, 3.97, 2700.0, 1.72, 3.13, 0.24, 18.1, 0.053};
    if (const char* m = getenv("LH_MATH"))
        if (!strcmp(m, "libm")) c->math = MATH_LIBM;
    if (const char* t = getenv("LH_TUNE")) parse_tune(c->tune, t);
    if (c->tune.rowpad >= 0) c->stride += c->tune.rowpad; // explicit row padding (elements)

#define CREATE_HIP(call)                                                                           \
    do {                                                                                           \
        hipError_t e2_ = (call);                                                                   \
        if (e2_ != hipSuccess) {                                                                   \
            int rc_ = fail(nullptr, e2_ == hipErrorOutOfMemory ? LH_ENOMEM : LH_ENODEVICE,         \
                           "lh_create: %s failed: %s", #call, hipGetErrorString(e2_));             \
            lh_destroy(c);                                                                         \
            return rc_;                                                                            \
        }                                                                                          \
    } while (0)

    if (cfg->stream) {
        c->stream = static_cast<hipStream_t>(cfg->stream);
    } else {
        CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    CREATE_HIP(hipEventCreate(&c->ev0));
    CREATE_HIP(hipEventCreate(&c->ev1));
    CREATE_HIP(hipMalloc(&c->d_zc, size_t(cfg->nlev) * c->esize));
    CREATE_HIP(hipMalloc(&c->d_status, sizeof(uint32_t)));
    CREATE_HIP(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream));
    CREATE_HIP(hipMalloc(&c->d_dt, 8));
    {
        // tables of lh_fastmath.hpp, built in long double and rounded once
        std::vector<double> tab(MATH_TAB_DOUBLES);
        for (int i = 0; i < LOG_TAB_N; ++i) {
            long double cc = 0.5L * (1.0L + (i + 0.5L) / LOG_TAB_N);
            if (i == 0) cc = 0.5L;               // log2(1) == 0 exactly, and
            if (i == LOG_TAB_N - 1) cc = 1.0L;   // nothing lost next to 1
            tab[2 * i] = double(1.0L / cc);
            tab[2 * i + 1] = double(log2l(cc));
        }
        for (int j = 0; j < EXP_TAB_N; ++j) {
            // 2^(j/2048), stored with (j << 9) subtracted from its high word: the kernels put the
            // binary exponent in place by ONE integer addition of (k << 9), k = 2048 e + j
            // (MathFast<double>::exp2_scaled_ins); the low word is the value's own
            const double v = double(exp2l((long double)j / EXP_TAB_N));
            uint64_t bits;
            memcpy(&bits, &v, 8);
            bits -= uint64_t(uint32_t(j) << 9) << 32;
            memcpy(&tab[2 * LOG_TAB_N + j], &bits, 8);
        }
        CREATE_HIP(hipMalloc(&c->d_math_tab, tab.size() * sizeof(double)));
        CREATE_HIP(hipMemcpy(c->d_math_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    c->zc_host.resize(cfg->nlev);
    if (cfg->dtype == LH_F64) {
        std::vector<double> z;
        make_grid<double>(cfg->zmin, cfg->zmax, cfg->nlev, z);
        for (int i = 0; i < cfg->nlev; ++i) c->zc_host[i] = z[i];
        CREATE_HIP(hipMemcpy(c->d_zc, z.data(), z.size() * 8, hipMemcpyHostToDevice));
    } else {
        std::vector<float> z;
        make_grid<float>(cfg->zmin, cfg->zmax, cfg->nlev, z);
        for (int i = 0; i < cfg->nlev; ++i) c->zc_host[i] = z[i];
        CREATE_HIP(hipMemcpy(c->d_zc, z.data(), z.size() * 4, hipMemcpyHostToDevice));
    }
#undef CREATE_HIP
    *out = c;
    return LH_OK;
}

int lh_destroy(lh_ctx* c) {
    if (!c) return LH_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) {
        (void)ncclCommDestroy(c->comm);
        c->comm = nullptr;
    }
    while (!c->states.empty()) state_free(c, c->states.back());
    for (auto& a : c->arenas) (void)hipFree(a.base); // none should be left
    c->arenas.clear();
    for (int i = 0; i < LH_PC_COUNT; ++i)
        if (c->d_pc[i]) (void)hipFree(c->d_pc[i]);
    for (int f = 0; f < 2; ++f)
        for (int k = 0; k < 2; ++k)
            if (c->d_bc_pc[f][k]) (void)hipFree(c->d_bc_pc[f][k]);
    if (c->d_zc) (void)hipFree(c->d_zc);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_dt) (void)hipFree(c->d_dt);
    for (int k = 0; k < 3; ++k)
        if (c->d_atm_pc[k]) (void)hipFree(c->d_atm_pc[k]);
    for (int k = 0; k < 2; ++k)
        if (c->d_atm_flux[k]) (void)hipFree(c->d_atm_flux[k]);
    if (c->d_math_tab) (void)hipFree(c->d_math_tab);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return LH_OK;
}

int lh_set_earth_params(lh_ctx* c, const lh_earth_params* p) {
    if (!c || !p) return fail(c, LH_EINVAL, "lh_set_earth_params: NULL argument");
    c->hp.earth = *p;
    c->hp.earth_set = true;
    return LH_OK;
}

int lh_set_soil_params(lh_ctx* c, const lh_soil_params* p) {
    if (!c || !p) return fail(c, LH_EINVAL, "lh_set_soil_params: NULL argument");
    c->hp.soil = *p;
    return LH_OK;
}

int lh_set_vg_params(lh_ctx* c, const lh_vg_params* p) {
    if (!c || !p) return fail(c, LH_EINVAL, "lh_set_vg_params: NULL argument");
    c->hp.vg = *p;
    return LH_OK;
}

int lh_set_percol_param(lh_ctx* c, int32_t id, const double* host) {
    if (!c) return LH_EINVAL;
    if (id < 0 || id >= LH_PC_COUNT) return fail(c, LH_EINVAL, "lh_set_percol_param: unknown parameter id %d", id);
    (void)hipSetDevice(c->device);
    if (host) { // the array's range decides which closure form the ensemble may take (vg_fast_all)
        double lo = host[0], hi = host[0];
        bool nan = false;
        for (int64_t k = 0; k < c->cfg.ncols; ++k) {
            const double v = host[k];
            nan = nan || v != v;
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        c->pc_lo[id] = nan ? NAN : lo;
        c->pc_hi[id] = nan ? NAN : hi;
    }
    return upload_percol(c, &c->d_pc[id], host);
}

int lh_set_conductivity_factors(lh_ctx* c, int32_t vk, double gamma, double T_ref, int32_t ik, double Omega) {
    if (!c) return LH_EINVAL;
    if ((vk != LH_FACTOR_NONE && vk != LH_FACTOR_ON) || (ik != LH_FACTOR_NONE && ik != LH_FACTOR_ON))
        return fail(c, LH_EINVAL, "lh_set_conductivity_factors: kind must be LH_FACTOR_NONE or LH_FACTOR_ON");
    c->hp.viscosity_kind = vk;
    c->hp.impedance_kind = ik;
    c->hp.gamma = gamma;
    c->hp.T_ref_visc = T_ref;
    c->hp.Omega = Omega;
    return LH_OK;
}

int lh_set_bc(lh_ctx* c, int32_t face, int32_t comp, int32_t kind, double value, const double* percol) {
    if (!c) return LH_EINVAL;
    if (face != LH_FACE_BOTTOM && face != LH_FACE_TOP)
        return fail(c, LH_EINVAL, "Expected :top or :bottom"); // boundary_conditions.jl:188
    if (comp != LH_COMP_ENERGY && comp != LH_COMP_HYDROLOGY) return fail(c, LH_EINVAL, "lh_set_bc: unknown component %d", comp);
    if (kind < LH_BC_NONE || kind > LH_BC_FREE_DRAINAGE) return fail(c, LH_EINVAL, "lh_set_bc: unknown kind %d", kind);
    c->hp.bc_kind[face][comp] = kind;
    c->hp.bc_value[face][comp] = value;
    (void)hipSetDevice(c->device);
    return upload_percol(c, &c->d_bc_pc[face][comp], percol);
}

int lh_set_atmos_forcing(lh_ctx* c, const lh_atmos_forcing* f, const double* percol) {
    if (!c) return LH_EINVAL;
    (void)hipSetDevice(c->device);
    if (!f) { // back to the component boundary conditions of lh_set_bc
        c->hp.atmos_on = false;
        for (int k = 0; k < 3; ++k) {
            int rc = upload_percol(c, &c->d_atm_pc[k], nullptr);
            if (rc) return rc;
        }
        return LH_OK;
    }
    if (c->cfg.model != LH_MODEL_COUPLED)
        return fail(c, LH_EMODEL, "PrescribedAtmosForcing needs SoilEnergyModel + SoilHydrologyModel (no method for this model)");
    if (!(f->z_atm > 0) || !(f->z_0m > 0) || !(f->z_0s > 0) || !(f->rho_a_sfc > 0) || !(f->theta_scale > 0) ||
        !(f->R_v > 0) || !(f->von_karman > 0) || !(f->T_triple > 0))
        return fail(c, LH_EINVAL, "lh_set_atmos_forcing: z_atm, z_0m, z_0s, rho_a_sfc, theta_scale, R_v, von_karman, T_triple must be > 0");
    for (int k = 0; k < 2; ++k)
        if (!c->d_atm_flux[k]) LH_HIP(c, hipMalloc(&c->d_atm_flux[k], size_t(c->cfg.ncols) * c->esize));
    for (int k = 0; k < 3; ++k) {
        int rc = upload_percol(c, &c->d_atm_pc[k], percol ? percol + size_t(k) * size_t(c->cfg.ncols) : nullptr);
        if (rc) return rc;
    }
    c->hp.atmos = *f;
    c->hp.atmos_on = true;
    return LH_OK;
}

int lh_atmos_surface_fluxes(lh_ctx* c, int64_t n, const double* vl, const double* ti, const double* T,
                            double* heat, double* water) {
    if (!c || !vl || !ti || !T || !heat || !water) return fail(c, LH_EINVAL, "lh_atmos_surface_fluxes: NULL argument");
    if (n < 0) return fail(c, LH_EINVAL, "lh_atmos_surface_fluxes: n < 0");
    if (!c->hp.atmos_on) return fail(c, LH_EMODEL, "no PrescribedAtmosForcing is set on this model (lh_set_atmos_forcing)");
    int rc = validate_model(c);
    if (rc) return rc;
    if (n == 0) return LH_OK;
    (void)hipSetDevice(c->device);
    const size_t es = c->esize;
    double* d_in = nullptr;   // 3 n doubles in, converted to FT; 2 n FT out
    char* d_ft = nullptr;
    LH_HIP(c, hipMalloc(&d_in, size_t(n) * 3 * sizeof(double)));
    hipError_t e = hipMalloc(&d_ft, size_t(n) * 5 * es);
    if (e != hipSuccess) {
        (void)hipFree(d_in);
        return fail(c, LH_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    auto cleanup = [&]() {
        (void)hipFree(d_in);
        (void)hipFree(d_ft);
    };
    const double* src[3] = {vl, ti, T};
    for (int k = 0; k < 3 && e == hipSuccess; ++k)
        e = hipMemcpyAsync(d_in + size_t(k) * n, src[k], size_t(n) * sizeof(double), hipMemcpyHostToDevice, c->stream);
    std::vector<char> out(size_t(n) * 2 * es);
    if (e == hipSuccess) {
        if (c->cfg.dtype == LH_F64) {
            double* f = reinterpret_cast<double*>(d_ft);
            launch_convert<double>(f, d_in, 3 * n, c->stream);
            launch_atmos_flux<double>(make_params<double>(c), make_atmos_params<double>(c), n, false, false, f, f + n, f + 2 * n,
                                      f + 3 * n, f + 4 * n, c->stream);
        } else {
            float* f = reinterpret_cast<float*>(d_ft);
            launch_convert<float>(f, d_in, 3 * n, c->stream);
            launch_atmos_flux<float>(make_params<float>(c), make_atmos_params<float>(c), n, false, false, f, f + n, f + 2 * n,
                                     f + 3 * n, f + 4 * n, c->stream);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out.data(), d_ft + size_t(n) * 3 * es, size_t(n) * 2 * es, hipMemcpyDeviceToHost, c->stream);
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, LH_ENODEVICE, "lh_atmos_surface_fluxes failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    for (int64_t i = 0; i < n; ++i) {
        if (c->cfg.dtype == LH_F64) {
            heat[i] = reinterpret_cast<const double*>(out.data())[i];
            water[i] = reinterpret_cast<const double*>(out.data())[n + i];
        } else {
            heat[i] = reinterpret_cast<const float*>(out.data())[i];
            water[i] = reinterpret_cast<const float*>(out.data())[n + i];
        }
    }
    return LH_OK;
}

int lh_set_bottom_sign_consistent(lh_ctx* c, int32_t flag) {
    if (!c) return LH_EINVAL;
    c->hp.consistent_bottom_sign = flag ? 1 : 0;
    return LH_OK;
}

int lh_set_tuning(lh_ctx* c, const char* spec) {
    if (!c || !spec) return LH_EINVAL;
    const int arena = c->tune.arena, pad = c->tune.pad, rowpad = c->tune.rowpad; // fixed at lh_create
    parse_tune(c->tune, spec);
    c->tune.arena = arena;
    c->tune.pad = pad;
    c->tune.rowpad = rowpad;
    return LH_OK;
}

int lh_set_math_mode(lh_ctx* c, int32_t mode) {
    if (!c) return LH_EINVAL;
    if (mode != LH_MATH_FAST && mode != LH_MATH_LIBM) return fail(c, LH_EINVAL, "lh_set_math_mode: unknown mode %d", mode);
    c->math = mode == LH_MATH_LIBM ? MATH_LIBM : MATH_FAST;
    return LH_OK;
}

int lh_state_create(lh_ctx* c, uint32_t mask, lh_state** out) {
    if (!c || !out) return fail(c, LH_EINVAL, "lh_state_create: NULL argument");
    if (mask == 0) mask = prognostic_mask(c->cfg.model);
    if (mask >= (1u << LH_NVARS)) return fail(c, LH_EINVAL, "lh_state_create: bad variable mask 0x%x", mask);
    (void)hipSetDevice(c->device);
    return state_alloc(c, mask, out);
}

int lh_state_destroy(lh_ctx* c, lh_state* s) {
    if (!c || !s) return LH_OK;
    if (s->ctx != c) return fail(c, LH_EINVAL, "lh_state_destroy: state belongs to another context");
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->scratch_u1 == s) c->scratch_u1 = nullptr;
    if (c->scratch_u2 == s) c->scratch_u2 = nullptr;
    if (c->scratch_k1 == s) c->scratch_k1 = nullptr;
    state_free(c, s);
    return LH_OK;
}

static int transfer(lh_ctx* c, lh_state* s, int32_t var, void* host, int64_t ls, int64_t cs, bool upload) {
    if (!c || !s || !host) return fail(c, LH_EINVAL, "lh_upload/lh_download: NULL argument");
    if (s->ctx != c) return fail(c, LH_EINVAL, "state belongs to another context");
    if (var < 0 || var >= LH_NVARS || !(s->mask & (1u << var))) return fail(c, LH_ESTATE, "state has no variable %d", var);
    if (ls < 1 || cs < 1) return fail(c, LH_EINVAL, "strides must be >= 1");
    (void)hipSetDevice(c->device);
    if (upload) mark_written(s, 1u << var);
    else {
        int rc = materialize(c, s, 1u << var);
        if (rc) return rc;
    }
    const int64_t ncols = c->cfg.ncols;
    const int nlev = c->cfg.nlev;
    const size_t es = c->esize;
    char* plane = static_cast<char*>(s->plane[var]);
    if (cs == 1 && ls >= ncols) { // already column-fastest: one 2-D copy
        if (upload)
            LH_HIP(c, hipMemcpy2DAsync(plane, size_t(c->stride) * es, host, size_t(ls) * es, size_t(ncols) * es, nlev, hipMemcpyHostToDevice, c->stream));
        else
            LH_HIP(c, hipMemcpy2DAsync(host, size_t(ls) * es, plane, size_t(c->stride) * es, size_t(ncols) * es, nlev, hipMemcpyDeviceToHost, c->stream));
        LH_HIP(c, hipStreamSynchronize(c->stream));
        return LH_OK;
    }
    const int64_t span = (ncols - 1) * cs + int64_t(nlev - 1) * ls + 1;
    void* tmp = nullptr;
    LH_HIP(c, hipMalloc(&tmp, size_t(span) * es));
    hipError_t e = hipSuccess;
    if (upload) e = hipMemcpyAsync(tmp, host, size_t(span) * es, hipMemcpyHostToDevice, c->stream);
    else if (span != ncols * int64_t(nlev)) // gaps in the user layout keep their contents
        e = hipMemcpyAsync(tmp, host, size_t(span) * es, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        if (c->cfg.dtype == LH_F64)
            launch_strided_copy<double>(reinterpret_cast<double*>(plane), c->stride, static_cast<double*>(tmp), ls, cs, ncols, nlev, upload, c->stream);
        else
            launch_strided_copy<float>(reinterpret_cast<float*>(plane), c->stride, static_cast<float*>(tmp), ls, cs, ncols, nlev, upload, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !upload) e = hipMemcpyAsync(host, tmp, size_t(span) * es, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, LH_ENODEVICE, "transfer failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(c, LH_ENODEVICE, "transfer failed: %s", hipGetErrorString(e2));
    return LH_OK;
}

int lh_upload(lh_ctx* c, lh_state* s, int32_t var, const void* host, int64_t ls, int64_t cs) {
    return transfer(c, s, var, const_cast<void*>(host), ls, cs, true);
}

int lh_download(lh_ctx* c, const lh_state* s, int32_t var, void* host, int64_t ls, int64_t cs) {
    return transfer(c, const_cast<lh_state*>(s), var, host, ls, cs, false);
}

int lh_download_level(lh_ctx* c, const lh_state* s, int32_t var, int32_t level, void* host) {
    if (!c || !s || !host) return fail(c, LH_EINVAL, "lh_download_level: NULL argument");
    if (s->ctx != c) return fail(c, LH_EINVAL, "state belongs to another context");
    if (var < 0 || var >= LH_NVARS || !(s->mask & (1u << var))) return fail(c, LH_ESTATE, "state has no variable %d", var);
    if (level < 0 || level >= c->cfg.nlev) return fail(c, LH_EINVAL, "lh_download_level: level %d outside [0, %d)", level, c->cfg.nlev);
    (void)hipSetDevice(c->device);
    {
        int rc = materialize(c, s, 1u << var);
        if (rc) return rc;
    }
    // a level of a plane is one contiguous row of ncols elements
    const char* row = static_cast<const char*>(s->plane[var]) + size_t(level) * size_t(c->stride) * c->esize;
    LH_HIP(c, hipMemcpyAsync(host, row, size_t(c->cfg.ncols) * c->esize, hipMemcpyDeviceToHost, c->stream));
    LH_HIP(c, hipStreamSynchronize(c->stream));
    return LH_OK;
}

int lh_state_fill(lh_ctx* c, lh_state* s, int32_t var, double value) {
    if (!c || !s) return fail(c, LH_EINVAL, "lh_state_fill: NULL argument");
    if (s->ctx != c) return fail(c, LH_EINVAL, "state belongs to another context");
    if (var < 0 || var >= LH_NVARS || !(s->mask & (1u << var))) return fail(c, LH_ESTATE, "state has no variable %d", var);
    (void)hipSetDevice(c->device);
    const int64_t n = int64_t(c->cfg.nlev) * c->stride;
    if (c->cfg.dtype == LH_F64) launch_fill<double>(static_cast<double*>(s->plane[var]), n, value, c->stream);
    else launch_fill<float>(static_cast<float*>(s->plane[var]), n, float(value), c->stream);
    LH_HIP(c, hipGetLastError());
    mark_written(s, 1u << var);
    if (value == 0.0 && !std::signbit(value)) mark_zero(s, var);
    return LH_OK;
}

int lh_state_copy(lh_ctx* c, lh_state* dst, const lh_state* src) {
    if (!c || !dst || !src) return fail(c, LH_EINVAL, "lh_state_copy: NULL argument");
    if (dst->ctx != c || src->ctx != c) return fail(c, LH_EINVAL, "state belongs to another context");
    if ((dst->mask & src->mask) != src->mask) return fail(c, LH_ESTATE, "lh_state_copy: destination lacks planes of the source");
    (void)hipSetDevice(c->device);
    const size_t bytes = size_t(c->cfg.nlev) * size_t(c->stride) * c->esize;
    int rc = materialize(c, src, ~0u); // (the copy carries planes, not profiles)
    if (rc) return rc;
    for (int i = 0; i < LH_NVARS; ++i)
        if (src->mask & (1u << i)) {
            LH_HIP(c, hipMemcpyAsync(dst->plane[i], src->plane[i], bytes, hipMemcpyDeviceToDevice, c->stream));
            mark_written(dst, 1u << i);
            if (src->zero_mask & (1u << i)) mark_zero(dst, i);
        }
    return LH_OK;
}

int lh_state_device_ptr(lh_ctx* c, const lh_state* s, int32_t var, void** dptr, int64_t* ls, int64_t* cs) {
    if (!c || !s || !dptr) return fail(c, LH_EINVAL, "lh_state_device_ptr: NULL argument");
    if (var < 0 || var >= LH_NVARS || !(s->mask & (1u << var))) return fail(c, LH_ESTATE, "state has no variable %d", var);
    (void)hipSetDevice(c->device);
    {
        int rc = materialize(c, s, 1u << var);
        if (rc) return rc;
    }
    *dptr = s->plane[var];
    // the caller may write through the pointer, now or later: nothing is assumed about this plane
    // until lh_state_release_ptr
    mark_written(const_cast<lh_state*>(s), 1u << var);
    const_cast<lh_state*>(s)->exposed_mask |= 1u << var;
    if (ls) *ls = c->stride;
    if (cs) *cs = 1;
    return LH_OK;
}

int lh_state_release_ptr(lh_ctx* c, lh_state* s, int32_t var) {
    if (!c || !s) return fail(c, LH_EINVAL, "lh_state_release_ptr: NULL argument");
    if (s->ctx != c) return fail(c, LH_EINVAL, "state belongs to another context");
    if (var < -1 || var >= LH_NVARS) return fail(c, LH_EINVAL, "lh_state_release_ptr: bad variable %d", var);
    s->exposed_mask &= var < 0 ? 0u : ~(1u << var);
    return LH_OK;
}

int lh_upload_profile(lh_ctx* c, lh_state* s, int32_t var, const void* host) {
    if (!c || !s || !host) return fail(c, LH_EINVAL, "lh_upload_profile: NULL argument");
    if (s->ctx != c) return fail(c, LH_EINVAL, "state belongs to another context");
    if (var < 0 || var >= LH_NVARS || !(s->mask & (1u << var))) return fail(c, LH_ESTATE, "state has no variable %d", var);
    (void)hipSetDevice(c->device);
    const int nlev = c->cfg.nlev;
    const size_t bytes = size_t(nlev) * c->esize;
    {   // an all-(+0) profile is a zero fill: the plane is then KNOWN to be zero (no ice: not even read)
        bool allzero = true;
        const unsigned char* b = static_cast<const unsigned char*>(host);
        for (size_t k = 0; k < bytes && allzero; ++k) allzero = b[k] == 0;
        if (allzero) return lh_state_fill(c, s, var, 0.0);
    }
    if (!s->profile[var]) LH_HIP(c, hipMalloc(&s->profile[var], bytes));
    // through a pinned staging ring, so that the copy is asynchronous and the call returns at once
    // (a time-dependent prescribed profile is refreshed at every stage: three uploads per step)
    if (!c->h_ring) {
        c->ring_bytes = size_t(1) << 18;
        if (c->ring_bytes < 4 * bytes) c->ring_bytes = 4 * bytes;
        LH_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_ring), c->ring_bytes, hipHostMallocDefault));
        c->ring_pos = 0;
    }
    const size_t need = (bytes + 255) & ~size_t(255);
    if (c->ring_pos + need > c->ring_bytes) { // wrap: every earlier copy out of the ring must have completed
        LH_HIP(c, hipStreamSynchronize(c->stream));
        c->ring_pos = 0;
    }
    memcpy(c->h_ring + c->ring_pos, host, bytes);
    LH_HIP(c, hipMemcpyAsync(s->profile[var], c->h_ring + c->ring_pos, bytes, hipMemcpyHostToDevice, c->stream));
    c->ring_pos += need;
    mark_written(s, 1u << var);
    s->profile_mask |= 1u << var;
    s->stale_mask |= 1u << var;
    return LH_OK;
}

int lh_coordinates(const lh_ctx* c, double* zc) {
    if (!c || !zc) return LH_EINVAL;
    memcpy(zc, c->zc_host.data(), c->zc_host.size() * sizeof(double));
    return LH_OK;
}

int lh_rhs(lh_ctx* c, double t, const lh_state* Y, const lh_state* Ya, lh_state* dY) {
    (void)t; // boundary/aux closures of t are evaluated by the host shim
    if (!c) return LH_EINVAL;
    int rc = validate_model(c);
    if (rc) return rc;
    const uint32_t pm = prognostic_mask(c->cfg.model);
    if ((rc = check_state(c, Y, pm, "Y"))) return rc;
    if ((rc = check_state(c, dY, pm, "dY"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    (void)hipSetDevice(c->device);
    return c->cfg.dtype == LH_F64 ? do_rhs<double>(c, Y, Ya, nullptr, dY, 0.0, 0, nullptr)
                                  : do_rhs<float>(c, Y, Ya, nullptr, dY, 0.0, 0, nullptr);
}

int lh_rhs_stable_dt(lh_ctx* c, double t, const lh_state* Y, const lh_state* Ya, lh_state* dY,
                     double courant, void* dt_device_ft) {
    (void)t;
    if (!c || !dt_device_ft) return fail(c, LH_EINVAL, "lh_rhs_stable_dt: NULL argument");
    if (!(courant > 0)) return fail(c, LH_EINVAL, "lh_rhs_stable_dt: courant must be > 0");
    int rc = validate_model(c);
    if (rc) return rc;
    const uint32_t pm = prognostic_mask(c->cfg.model);
    if ((rc = check_state(c, Y, pm, "Y"))) return rc;
    if ((rc = check_state(c, dY, pm, "dY"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    (void)hipSetDevice(c->device);
    rc = c->cfg.dtype == LH_F64
             ? do_rhs<double>(c, Y, Ya, nullptr, dY, courant, 4, nullptr, nullptr, dt_device_ft)
             : do_rhs<float>(c, Y, Ya, nullptr, dY, courant, 4, nullptr, nullptr, dt_device_ft);
    if (rc) return rc;
    return allreduce_min(c, dt_device_ft); // the global minimum when a communicator is attached
}

int lh_boundary_fluxes(lh_ctx* c, const lh_state* Y, const lh_state* Ya, double t, int32_t face, double* f_energy,
                       double* f_water) {
    (void)t; // boundary values of time t are set by the host shim (lh_set_bc) before the call, as for lh_rhs
    if (!c) return LH_EINVAL;
    if (face != LH_FACE_BOTTOM && face != LH_FACE_TOP) return fail(c, LH_EINVAL, "Expected :top or :bottom"); // boundary_conditions.jl:188
    if (!f_energy && !f_water) return fail(c, LH_EINVAL, "lh_boundary_fluxes: both outputs are NULL");
    int rc = validate_model(c);
    if (rc) return rc;
    if ((rc = check_state(c, Y, prognostic_mask(c->cfg.model), "Y"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    (void)hipSetDevice(c->device);
    if ((rc = materialize(c, Y, ~0u)) || (rc = materialize(c, Ya, ~0u))) return rc;
    Range r_("lh:boundary_fluxes");
    return c->cfg.dtype == LH_F64 ? boundary_fluxes_impl<double>(c, Y, Ya, face, f_energy, f_water)
                                  : boundary_fluxes_impl<float>(c, Y, Ya, face, f_energy, f_water);
}

int lh_diagnostics(lh_ctx* c, const lh_state* Y, const lh_state* Ya, lh_state* out) {
    if (!c) return LH_EINVAL;
    int rc;
    if (model_heat(c->cfg.model) && !c->hp.earth_set)
        return fail(c, LH_EINVAL, "earth parameters (lh_set_earth_params) are required by the energy model");
    if ((rc = check_state(c, Y, prognostic_mask(c->cfg.model), "Y"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    if ((rc = check_state(c, out, 0xFu, "diagnostic"))) return rc;
    (void)hipSetDevice(c->device);
    if ((rc = materialize(c, Y, ~0u)) || (rc = materialize(c, Ya, ~0u))) return rc;
    if (c->cfg.dtype == LH_F64) {
        DevParams<double> P = make_params<double>(c);
        launch_diag<double>(P, planes_of<double>(Y), planes_of<double>(Ya), planes_of<double>(out), any_percol(c), c->math, c->stream);
    } else {
        DevParams<float> P = make_params<float>(c);
        launch_diag<float>(P, planes_of<float>(Y), planes_of<float>(Ya), planes_of<float>(out), any_percol(c), c->math, c->stream);
    }
    LH_HIP(c, hipGetLastError());
    mark_written(out, ~0u);
    return LH_OK;
}

int lh_step_engine(const lh_ctx* c, int64_t nsteps, int32_t per_stage_boundary_values) {
    if (!c || nsteps < 0) return LH_EINVAL;
    return use_column_stepper(c, nsteps, per_stage_boundary_values != 0) ? LH_ENGINE_COLUMN_STEPPER : LH_ENGINE_FUSED_STAGES;
}

int lh_step_ssprk33(lh_ctx* c, lh_state* Y, const lh_state* Ya, double t, double dt, int64_t nsteps,
                    const double* bcv) {
    (void)t;
    if (!c) return LH_EINVAL;
    if (nsteps < 0 || !(dt > 0)) return fail(c, LH_EINVAL, "lh_step_ssprk33: need nsteps >= 0 and dt > 0");
    Range r_("lh:step_ssprk33");
    int rc = validate_model(c);
    if (rc) return rc;
    const uint32_t pm = prognostic_mask(c->cfg.model);
    if ((rc = check_state(c, Y, pm, "Y"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    (void)hipSetDevice(c->device);
    // Default: all nsteps in ONE launch of the persistent column stepper (state in registers,
    // no plane traffic between stages or steps; column_stepper_kernel)
    if (use_column_stepper(c, nsteps, bcv != nullptr)) return run_column_stepper(c, Y, Ya, dt, nullptr, nsteps, bcv);
    // the stage state carries no theta_i plane: the fused stages read theta_i from Y
    if (!c->scratch_u1 && (rc = state_alloc(c, pm & ~LH_MASK(LH_VAR_THETA_I), &c->scratch_u1))) return rc;
    lh_state* U1 = c->scratch_u1;
    lh_state* U2 = U1;
    if ((rc = second_stage_state(c, &U2))) return rc;
    auto one_step = [&](const double* bc3) -> int {
        for (int stage = 0; stage < 3; ++stage) {
            const double* ov = bc3 ? bc3 + stage * 4 : nullptr;
            // stage 1: U1 = Y + dt f(Y); 2: U2 = (3Y + U1 + dt f(U1))/4; 3: Y = (Y + 2U2 + 2dt f(U2))/3
            // (U2 is U1 itself unless the launch is level-segmented)
            const lh_state* in = stage == 0 ? Y : (stage == 1 ? U1 : U2);
            lh_state* out = stage == 2 ? Y : (stage == 1 ? U2 : U1);
            int r = c->cfg.dtype == LH_F64 ? do_rhs<double>(c, in, Ya, Y, out, dt, stage + 1, ov)
                                           : do_rhs<float>(c, in, Ya, Y, out, dt, stage + 1, ov);
            if (r) return r;
        }
        return LH_OK;
    };
    int64_t done = 0;
    // Small ensembles with constant boundary values: the launches themselves are the cost of a
    // step (a few microseconds of device work each), so a block of steps is captured once into
    // a hipGraph and replayed.  Anything that goes wrong with the capture falls back to plain
    // launches of whatever is left.
    constexpr int GRAPH_STEPS = 16;
    if (!bcv && c->tune.graph != 0 && segment_length(c) > 0 && nsteps >= 4 * GRAPH_STEPS) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            for (int k = 0; k < GRAPH_STEPS && !rc; ++k) rc = one_step(nullptr);
            ok = hipStreamEndCapture(c->stream, &graph) == hipSuccess && graph && !rc;
            if (rc) { // a launch was refused while capturing: nothing has run yet
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                return rc;
            }
        }
        if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (ok)
            for (; done + GRAPH_STEPS <= nsteps; done += GRAPH_STEPS)
                if (hipGraphLaunch(exec, c->stream) != hipSuccess) {
                    ok = false;
                    break;
                }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (!ok) {
            const hipError_t e = hipGetLastError();
            if (done > 0 && done < nsteps && e != hipSuccess)
                return fail(c, LH_ENODEVICE, "hipGraphLaunch failed after %lld steps: %s", (long long)done, hipGetErrorString(e));
        }
    }
    for (int64_t s = done; s < nsteps; ++s)
        if ((rc = one_step(bcv ? bcv + s * 12 : nullptr))) return rc;
    return LH_OK;
}

int lh_ssprk33_stage(lh_ctx* c, int32_t stage, lh_state* Y, lh_state* U, const lh_state* Ya, double dt,
                     const double* bc_values) {
    if (!c) return LH_EINVAL;
    if (stage < 1 || stage > 3) return fail(c, LH_EINVAL, "lh_ssprk33_stage: stage must be 1, 2 or 3");
    if (!(dt > 0)) return fail(c, LH_EINVAL, "lh_ssprk33_stage: need dt > 0");
    int rc = validate_model(c);
    if (rc) return rc;
    const uint32_t pm = prognostic_mask(c->cfg.model);
    if ((rc = check_state(c, Y, pm, "Y"))) return rc;
    if ((rc = check_state(c, U, pm & ~LH_MASK(LH_VAR_THETA_I), "U"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    if (U == Y) return fail(c, LH_EINVAL, "lh_ssprk33_stage: U must not be Y");
    (void)hipSetDevice(c->device);
    // stage 1: U = Y + dt f(Y); 2: U = (3Y + U + dt f(U))/4; 3: Y = (Y + 2U + 2dt f(U))/3
    const lh_state* in = stage == 1 ? Y : U;
    lh_state* out = stage == 3 ? Y : U;
    return c->cfg.dtype == LH_F64 ? do_rhs<double>(c, in, Ya, Y, out, dt, stage, bc_values, nullptr, nullptr, true)
                                  : do_rhs<float>(c, in, Ya, Y, out, dt, stage, bc_values, nullptr, nullptr, true);
}

int lh_step_ssprk33_device_dt(lh_ctx* c, lh_state* Y, const lh_state* Ya, double t,
                              const void* dt_device_ft, const double* bcv) {
    (void)t;
    if (!c || !dt_device_ft) return fail(c, LH_EINVAL, "lh_step_ssprk33_device_dt: NULL argument");
    int rc = validate_model(c);
    if (rc) return rc;
    const uint32_t pm = prognostic_mask(c->cfg.model);
    if ((rc = check_state(c, Y, pm, "Y"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    (void)hipSetDevice(c->device);
    if (use_column_stepper(c, 1, bcv != nullptr)) return run_column_stepper(c, Y, Ya, 0.0, dt_device_ft, 1, bcv);
    if (!c->scratch_u1 && (rc = state_alloc(c, pm & ~LH_MASK(LH_VAR_THETA_I), &c->scratch_u1))) return rc;
    lh_state* U1 = c->scratch_u1;
    lh_state* U2 = U1;
    if ((rc = second_stage_state(c, &U2))) return rc;
    for (int stage = 0; stage < 3; ++stage) {
        const double* ov = bcv ? bcv + stage * 4 : nullptr;
        const lh_state* in = stage == 0 ? Y : (stage == 1 ? U1 : U2);
        lh_state* out = stage == 2 ? Y : (stage == 1 ? U2 : U1);
        rc = c->cfg.dtype == LH_F64 ? do_rhs<double>(c, in, Ya, Y, out, 0.0, stage + 1, ov, dt_device_ft)
                                    : do_rhs<float>(c, in, Ya, Y, out, 0.0, stage + 1, ov, dt_device_ft);
        if (rc) return rc;
    }
    return LH_OK;
}

int lh_step_ssprk33_adaptive(lh_ctx* c, lh_state* Y, const lh_state* Ya, double t, double courant, double dt_max,
                             int64_t nsteps, void* dt_device_ft, void* elapsed_device_ft) {
    (void)t;
    if (!c || !dt_device_ft) return fail(c, LH_EINVAL, "lh_step_ssprk33_adaptive: NULL argument");
    if (nsteps < 0 || !(courant > 0)) return fail(c, LH_EINVAL, "lh_step_ssprk33_adaptive: need nsteps >= 0 and courant > 0");
    Range r_("lh:step_ssprk33_adaptive");
    int rc = validate_model(c);
    if (rc) return rc;
    const uint32_t pm = prognostic_mask(c->cfg.model);
    if ((rc = check_state(c, Y, pm, "Y"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    (void)hipSetDevice(c->device);
    if (!c->scratch_k1 && (rc = state_alloc(c, pm, &c->scratch_k1))) return rc;
    if (!c->scratch_u1 && (rc = state_alloc(c, pm & ~LH_MASK(LH_VAR_THETA_I), &c->scratch_u1))) return rc;
    lh_state* K1 = c->scratch_k1;
    lh_state* U1 = c->scratch_u1;
    const bool f64 = c->cfg.dtype == LH_F64;
    // With a prescribed atmosphere the surface fluxes of a stage come from the stage state's top
    // cells, which MODE 5 never stores: that model takes the four-launch sequence.
    const bool three = !c->hp.atmos_on;
    lh_state* U2 = U1;
    if (!three && (rc = second_stage_state(c, &U2))) return rc;
    for (int64_t s = 0; s < nsteps; ++s) {
        // f(Y) and the step bound of Y in one launch; the global minimum with a communicator
        rc = f64 ? do_rhs<double>(c, Y, Ya, nullptr, K1, courant, 4, nullptr, nullptr, dt_device_ft)
                 : do_rhs<float>(c, Y, Ya, nullptr, K1, courant, 4, nullptr, nullptr, dt_device_ft);
        if (rc) return rc;
        if ((rc = allreduce_min(c, dt_device_ft))) return rc;
        if (f64) launch_dt_prepare<double>(static_cast<double*>(dt_device_ft), dt_max, static_cast<double*>(elapsed_device_ft), c->d_status, c->stream);
        else launch_dt_prepare<float>(static_cast<float*>(dt_device_ft), float(dt_max), static_cast<float*>(elapsed_device_ft), c->d_status, c->stream);
        if (three) {
            // stage 2 from (Y, k1): U1 = Y + dt k1 formed in registers; then stage 3
            rc = f64 ? do_rhs<double>(c, K1, Ya, Y, U1, 0.0, 5, nullptr, dt_device_ft)
                     : do_rhs<float>(c, K1, Ya, Y, U1, 0.0, 5, nullptr, dt_device_ft);
            if (rc) return rc;
            rc = f64 ? do_rhs<double>(c, U1, Ya, Y, Y, 0.0, 3, nullptr, dt_device_ft)
                     : do_rhs<float>(c, U1, Ya, Y, Y, 0.0, 3, nullptr, dt_device_ft);
            if (rc) return rc;
        } else {
            for (int stage = 0; stage < 3; ++stage) {
                const lh_state* in = stage == 0 ? Y : (stage == 1 ? U1 : U2);
                lh_state* out = stage == 2 ? Y : (stage == 1 ? U2 : U1);
                rc = f64 ? do_rhs<double>(c, in, Ya, Y, out, 0.0, stage + 1, nullptr, dt_device_ft)
                         : do_rhs<float>(c, in, Ya, Y, out, 0.0, stage + 1, nullptr, dt_device_ft);
                if (rc) return rc;
            }
        }
    }
    LH_HIP(c, hipGetLastError());
    return LH_OK;
}

int lh_stable_dt_device(lh_ctx* c, const lh_state* Y, const lh_state* Ya, double courant, void* d_out) {
    if (!c || !d_out) return fail(c, LH_EINVAL, "lh_stable_dt_device: NULL argument");
    int rc;
    if (model_heat(c->cfg.model) && !c->hp.earth_set)
        return fail(c, LH_EINVAL, "earth parameters (lh_set_earth_params) are required by the energy model");
    if ((rc = check_state(c, Y, prognostic_mask(c->cfg.model), "Y"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    (void)hipSetDevice(c->device);
    if ((rc = materialize(c, Y, ~0u)) || (rc = materialize(c, Ya, ~0u))) return rc;
    if (c->cfg.dtype == LH_F64) {
        DevParams<double> P = make_params<double>(c);
        launch_stable_dt<double>(P, planes_of<double>(Y), planes_of<double>(Ya), courant, d_out, any_percol(c), c->stream);
    } else {
        DevParams<float> P = make_params<float>(c);
        launch_stable_dt<float>(P, planes_of<float>(Y), planes_of<float>(Ya), float(courant), d_out, any_percol(c), c->stream);
    }
    LH_HIP(c, hipGetLastError());
    return allreduce_min(c, d_out); // the global minimum when a communicator is attached
}

int lh_stable_dt(lh_ctx* c, const lh_state* Y, const lh_state* Ya, double courant, double* dt_host) {
    if (!c || !dt_host) return fail(c, LH_EINVAL, "lh_stable_dt: NULL argument");
    int rc = lh_stable_dt_device(c, Y, Ya, courant, c->d_dt);
    if (rc) return rc;
    if (c->cfg.dtype == LH_F64) {
        double v;
        LH_HIP(c, hipMemcpyAsync(&v, c->d_dt, 8, hipMemcpyDeviceToHost, c->stream));
        LH_HIP(c, hipStreamSynchronize(c->stream));
        *dt_host = v;
    } else {
        float v;
        LH_HIP(c, hipMemcpyAsync(&v, c->d_dt, 4, hipMemcpyDeviceToHost, c->stream));
        LH_HIP(c, hipStreamSynchronize(c->stream));
        *dt_host = v;
    }
    return LH_OK;
}

int lh_get_status(lh_ctx* c, uint32_t* flags) {
    if (!c || !flags) return LH_EINVAL;
    (void)hipSetDevice(c->device);
    LH_HIP(c, hipMemcpyAsync(flags, c->d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    LH_HIP(c, hipMemsetAsync(c->d_status, 0, sizeof(uint32_t), c->stream));
    LH_HIP(c, hipStreamSynchronize(c->stream));
    return LH_OK;
}

int lh_synchronize(lh_ctx* c) {
    if (!c) return LH_EINVAL;
    (void)hipSetDevice(c->device);
    LH_HIP(c, hipStreamSynchronize(c->stream));
    return LH_OK;
}

int lh_tune_placement(lh_ctx* c, lh_state* Y, const lh_state* Ya, lh_state* dY, int max_candidates,
                      uint32_t flags, float* ms_before, float* ms_after) {
    if (!c) return LH_EINVAL;
    int rc = validate_model(c);
    if (rc) return rc;
    const uint32_t pm = prognostic_mask(c->cfg.model);
    if ((rc = check_state(c, Y, pm, "Y"))) return rc;
    if ((rc = check_state(c, Ya, aux_mask(c), "Ya"))) return rc;
    if (dY && (rc = check_state(c, dY, pm, "dY"))) return rc;
    if (dY == Y) return fail(c, LH_EINVAL, "lh_tune_placement: dY must not be Y");
    if (flags & ~uint32_t(LH_PLACE_MOVE_INPUT)) return fail(c, LH_EINVAL, "lh_tune_placement: unknown flag bits");
    (void)hipSetDevice(c->device);
    if ((rc = materialize(c, Y, ~0u)) || (rc = materialize(c, Ya, ~0u)) || (rc = materialize(c, dY, ~0u))) return rc;
    const bool f64 = c->cfg.dtype == LH_F64;
    lh_state* written = dY;
    if (!dY && (use_column_stepper(c, 1) || segment_length(c) > 0)) { // no stage state in HBM, or a cache-resident one
        if (ms_before) *ms_before = 0;
        if (ms_after) *ms_after = 0;
        return LH_OK;
    }
    if (!dY) { // the stage state of the fused SSPRK33 stepper
        if (!c->scratch_u1 && (rc = state_alloc(c, pm & ~LH_MASK(LH_VAR_THETA_I), &c->scratch_u1))) return rc;
        written = c->scratch_u1;
    }
    auto run = [&]() -> int {
        if (dY) // the tendency launch of lh_rhs / lh_rhs_stable_dt
            return f64 ? do_rhs<double>(c, Y, Ya, nullptr, dY, 0.0, 0, nullptr)
                       : do_rhs<float>(c, Y, Ya, nullptr, dY, 0.0, 0, nullptr);
        // stages 1 and 2 with dt = 0 stream the same planes as a step and leave Y untouched
        // (U1 = Y + 0 f(Y); U1 = (3Y + U1 + 0 f(U1))/4)
        lh_state* U1 = c->scratch_u1;
        int r = f64 ? do_rhs<double>(c, Y, Ya, Y, U1, 0.0, 1, nullptr)
                    : do_rhs<float>(c, Y, Ya, Y, U1, 0.0, 1, nullptr);
        if (r) return r;
        return f64 ? do_rhs<double>(c, U1, Ya, Y, U1, 0.0, 2, nullptr)
                   : do_rhs<float>(c, U1, Ya, Y, U1, 0.0, 2, nullptr);
    };
    // The written planes matter most (two write streams in an unlucky relative position cost
    // ~12 %; reads hardly care): first the written state as a whole, then each of its planes
    // on its own against the others, then -- if allowed -- the read state.
    // (a tendency state's theta_i plane is never written -- d theta_i = 0 is kept by the zero bits --
    // so it does not take part)
    float b0 = 0, a0 = 0, b1 = 0, a1 = 0;
    uint32_t wmask = 0;
    for (int i = 0; i < LH_NVARS; ++i)
        if (written->plane[i] && !(dY && i == LH_VAR_THETA_I)) wmask |= 1u << i;
    if ((rc = tune_state_planes(c, written, wmask, max_candidates, false, run, &b0, &a0))) return rc;
    int nwritten = 0;
    for (int i = 0; i < LH_NVARS; ++i) nwritten += (wmask >> i) & 1u;
    for (int i = 0; i < LH_NVARS && nwritten > 1; ++i)
        if (wmask >> i & 1u) {
            if ((rc = tune_state_planes(c, written, 1u << i, max_candidates, false, run, &b1, &a1))) return rc;
            if (a1 < a0) a0 = a1; // b1 re-measures the placement a0 was measured on
        }
    if (flags & LH_PLACE_MOVE_INPUT) {
        if ((rc = tune_state_planes(c, Y, ~0u, max_candidates, true, run, &b1, &a1))) return rc;
        if (a1 < a0) a0 = a1;
    }
    if (ms_before) *ms_before = b0;
    if (ms_after) *ms_after = a0;
    return LH_OK;
}

int lh_block_range(int64_t ncols_global, int32_t rank, int32_t nranks, int64_t* lo, int64_t* hi) {
    if (!lo || !hi || nranks < 1 || rank < 0 || rank >= nranks || ncols_global < nranks) return LH_EINVAL;
    const int64_t q = ncols_global / nranks, r = ncols_global % nranks;
    *lo = rank * q + (rank < r ? rank : r);
    *hi = *lo + q + (rank < r ? 1 : 0);
    return LH_OK;
}

int lh_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == LH_COMM_ID_BYTES, "LH_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    if (!id_out) return fail(nullptr, LH_EINVAL, "lh_comm_unique_id: NULL argument");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, LH_ENODEVICE, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
    memcpy(id_out, &id, sizeof id);
    return LH_OK;
}

int lh_comm_init(lh_ctx* c, int32_t rank, int32_t nranks, const void* unique_id) {
    if (!c || !unique_id) return fail(c, LH_EINVAL, "lh_comm_init: NULL argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, LH_EINVAL, "lh_comm_init: rank %d outside [0, %d)", rank, nranks);
    if (c->comm) return fail(c, LH_EINVAL, "lh_comm_init: a communicator is already attached (lh_comm_destroy first)");
    (void)hipSetDevice(c->device);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return fail(c, LH_ENODEVICE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, ncclGetErrorString(r));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return LH_OK;
}

int lh_comm_destroy(lh_ctx* c) {
    if (!c) return LH_EINVAL;
    if (!c->comm) return LH_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    const ncclResult_t r = ncclCommDestroy(c->comm);
    c->comm = nullptr;
    c->comm_rank = 0;
    c->comm_nranks = 1;
    if (r != ncclSuccess) return fail(c, LH_ENODEVICE, "ncclCommDestroy failed: %s", ncclGetErrorString(r));
    return LH_OK;
}

int lh_comm_info(const lh_ctx* c, int32_t* rank, int32_t* nranks) {
    if (!c) return LH_EINVAL;
    if (rank) *rank = c->comm ? c->comm_rank : 0;
    if (nranks) *nranks = c->comm ? c->comm_nranks : 1;
    return LH_OK;
}

int lh_allreduce_min(lh_ctx* c, void* value_device_ft) {
    if (!c || !value_device_ft) return fail(c, LH_EINVAL, "lh_allreduce_min: NULL argument");
    (void)hipSetDevice(c->device);
    return allreduce_min(c, value_device_ft);
}

int lh_stream_probe(lh_ctx* c, const lh_state* in, uint32_t read_mask, lh_state* out, uint32_t write_mask,
                    int reps, float* ms_per_launch) {
    if (!c || !in || !out || !ms_per_launch) return fail(c, LH_EINVAL, "lh_stream_probe: NULL argument");
    if (in->ctx != c || out->ctx != c) return fail(c, LH_EINVAL, "state belongs to another context");
    if ((in->mask & read_mask) != read_mask || (out->mask & write_mask) != write_mask)
        return fail(c, LH_ESTATE, "lh_stream_probe: a selected plane does not exist");
    if (reps < 1) reps = 20;
    (void)hipSetDevice(c->device);
    {
        int rc;
        if ((rc = materialize(c, in, read_mask)) || (rc = materialize(c, out, write_mask))) return rc;
    }
    // the selected planes, packed to the front
    void *rp[4] = {nullptr, nullptr, nullptr, nullptr}, *wp[4] = {nullptr, nullptr, nullptr, nullptr};
    int nr = 0, nw = 0;
    for (int i = 0; i < LH_NVARS; ++i) {
        if (read_mask >> i & 1u) rp[nr++] = in->plane[i];
        if (write_mask >> i & 1u) wp[nw++] = out->plane[i];
    }
    const double touched = double(c->cfg.nlev) * double(c->stride) * double(c->esize) * (nr + nw);
    const bool nt = c->tune.nt >= 0 ? c->tune.nt != 0 : touched > 192.0 * 1024 * 1024; // as launch_rhs_model
    auto go = [&]() {
        if (c->cfg.dtype == LH_F64) {
            Planes<double> pi, po;
            for (int k = 0; k < 4; ++k) pi.v[k] = static_cast<double*>(rp[k]), po.v[k] = static_cast<double*>(wp[k]);
            launch_stream_probe<double>(c->cfg.ncols, c->stride, c->cfg.nlev, c->tune.xcd, pi, nr, po, nw, nt, c->stream);
        } else {
            Planes<float> pi, po;
            for (int k = 0; k < 4; ++k) pi.v[k] = static_cast<float*>(rp[k]), po.v[k] = static_cast<float*>(wp[k]);
            launch_stream_probe<float>(c->cfg.ncols, c->stride, c->cfg.nlev, c->tune.xcd, pi, nr, po, nw, nt, c->stream);
        }
    };
    mark_written(out, write_mask);
    for (int r = 0; r < 3; ++r) go();
    LH_HIP(c, hipGetLastError());
    LH_HIP(c, hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; ++r) go();
    LH_HIP(c, hipEventRecord(c->ev1, c->stream));
    LH_HIP(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    LH_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *ms_per_launch = ms / float(reps);
    return LH_OK;
}

int lh_timer_start(lh_ctx* c) {
    if (!c) return LH_EINVAL;
    (void)hipSetDevice(c->device);
    LH_HIP(c, hipEventRecord(c->ev0, c->stream));
    return LH_OK;
}

int lh_timer_stop(lh_ctx* c, float* ms) {
    if (!c || !ms) return LH_EINVAL;
    (void)hipSetDevice(c->device);
    LH_HIP(c, hipEventRecord(c->ev1, c->stream));
    LH_HIP(c, hipEventSynchronize(c->ev1));
    LH_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return LH_OK;
}

} // extern "C"

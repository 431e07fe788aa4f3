// lh_launch.hpp -- host-callable launchers implemented in lh_kernels.hip.
#pragma once
#include "lh_device.hpp"

namespace lh {

enum { MATH_FAST = 0, MATH_LIBM = 1 };

template <int CPL_, int PF_, bool NT_, bool SEG_ = false> struct KCfg;
// production launch shape per working type (chosen by measurement)
template <typename FT> struct DefaultCfg;
#ifndef LH_PF64
#define LH_PF64 2 // levels in flight ahead of the one computed (Float64): -2..3 % against 1 (profiles/round1_tune_prefetch.txt)
#endif
template <> struct DefaultCfg<double> { using type = KCfg<1, LH_PF64, false>; };
template <> struct DefaultCfg<float> { using type = KCfg<2, 1, false>; };

// run-time launch overrides (LH_TUNE environment variable; tuning builds only
// honour cpl/pf/nt, every build honours block)
struct Tune {
    int block = 0, cpl = 0, pf = 0, nt = -1;
    int pad = -1; // plane address stagger in bytes (state allocation)
    int arena = 0; // plane slots per device allocation (0 = default)
    int rowpad = -1; // extra elements per plane row (-1 = library default)
    int seg = 0;     // levels per segment of the level-segmented launch (0 = automatic, -1 = never)
    int cpb = 0;     // persistent column stepper: columns per workgroup (0 = default)
    int persist = 1; // persistent column stepper (0: fused-stage launches instead)
    int graph = 1;   // replay blocks of fused SSPRK33 steps of small ensembles as a hipGraph (0: plain launches)
    int xcd = 1;     // XCD-contiguous workgroup map (0: workgroup b = column block b)
    int place_mem = 0;  // lh_tune_placement: transient memory bound, percent of free memory (0 = 25)
    int zero = 1;    // use the states' known-zero plane bits (0: always read theta_i and store d theta_i = 0)
    int vgfast = 1;  // Float64: integer-exponent 2^(.) in the water closures when every column allows it (0: v_ldexp form always)
};

// mode 0: tendency into `out`; 4: tendency + step bound; 1..3, 5: fused SSPRK33 stages (see rhs_kernel)
template <typename FT>
void launch_rhs(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                const Planes<FT>& base, const Planes<FT>& out, FT dt, const FT* dt_dev, int mode,
                bool factors, bool percol, bool noice, int math, const Tune& tune, hipStream_t s);
// nsteps fused SSPRK33 steps of a small ensemble in ONE launch (workgroup = column, thread = cell)
template <typename FT>
void launch_column_stepper(const DevParams<FT>& P, const Planes<FT>& Y, const Planes<FT>& aux, FT dt,
                           const FT* dt_dev, int64_t nsteps, const FT* bcv, bool factors, bool percol,
                           bool noice, hipStream_t s);
template <typename FT>
void launch_diag(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                 const Planes<FT>& out, bool percol, int math, hipStream_t s);
// boundary_fluxes of one face for every column (lh_boundary_fluxes): FT[ncols] each
template <typename FT>
void launch_boundary_fluxes(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux, int face, FT* out_e,
                            FT* out_w, bool factors, bool percol, int math, hipStream_t s);
template <typename FT>
void launch_stable_dt(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                      FT courant, void* out_ft, bool percol, hipStream_t s);
template <typename FT>
void launch_strided_copy(FT* plane, int64_t stride, FT* user, int64_t ls, int64_t cs,
                         int64_t ncols, int nlev, bool to_plane, hipStream_t s);
// the column launch's access pattern without arithmetic: nr planes of `in` read, nw planes of `out` written
template <typename FT>
void launch_stream_probe(int64_t ncols, int64_t stride, int nlev, int xcd_remap, const Planes<FT>& in, int nr,
                         const Planes<FT>& out, int nw, bool nt, hipStream_t s);
// compute_turbulent_surface_fluxes for n top-cell states (lh_atmos.hpp)
template <typename FT>
void launch_atmos_flux(const DevParams<FT>& P, const AtmosParams<FT>& A, int64_t n, bool from_state, bool percol,
                       const FT* vl, const FT* ti, const FT* third, FT* out_heat, FT* out_water, hipStream_t s);
template <typename FT>
void launch_fill(FT* p, int64_t n, FT v, hipStream_t s);
// plane[lev][col] = prof[lev]: a level-uniform variable made a plane (lh_upload_profile)
template <typename FT>
void launch_broadcast_profile(FT* plane, const FT* prof, int64_t ncols, int64_t stride, int nlev, hipStream_t s);
// one thread: dt = min(dt, dt_max), elapsed += dt (lh_step_ssprk33_adaptive)
template <typename FT>
void launch_dt_prepare(FT* dt, FT dt_max, FT* elapsed, uint32_t* status, hipStream_t s);
template <typename FT>
void launch_convert(FT* dst, const double* src, int64_t n, hipStream_t s);

} // namespace lh

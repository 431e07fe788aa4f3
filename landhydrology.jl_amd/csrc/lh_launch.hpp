// lh_launch.hpp -- host-callable launchers implemented in lh_kernels.hip.
#pragma once
#include "lh_device.hpp"

namespace lh {

enum { MATH_FAST = 0, MATH_LIBM = 1 };

// columns per lane: 16 B per lane and level for both working types
template <typename FT> struct LH_CPL;
template <> struct LH_CPL<double> { static constexpr int value = 1; };
template <> struct LH_CPL<float> { static constexpr int value = 2; };

// mode 0: tendency into `out`; 1..3: fused SSPRK33 stage (see rhs_kernel)
template <typename FT>
void launch_rhs(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                const Planes<FT>& base, const Planes<FT>& out, FT dt, int mode, bool factors,
                bool percol, int math, hipStream_t s);
template <typename FT>
void launch_diag(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                 const Planes<FT>& out, bool percol, int math, hipStream_t s);
template <typename FT>
void launch_stable_dt(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                      FT courant, void* out_ft, bool percol, hipStream_t s);
template <typename FT>
void launch_strided_copy(FT* plane, int64_t stride, FT* user, int64_t ls, int64_t cs,
                         int64_t ncols, int nlev, bool to_plane, hipStream_t s);
template <typename FT>
void launch_fill(FT* p, int64_t n, FT v, hipStream_t s);
template <typename FT>
void launch_convert(FT* dst, const double* src, int64_t n, hipStream_t s);

} // namespace lh

// lh_fastmath.hpp -- math policies for the soil closures on gfx950.
//
//  MathLibm<FT>  ocml pow/exp/log: <= 1 ulp, the reference-faithful policy used
//                for parity debugging and for once-per-column constants.
//  MathFast<FT>  the production policy (see the second half of this file).
#pragma once
#include <hip/hip_runtime.h>

namespace lh {

template <typename FT> struct Limits;
template <> struct Limits<double> {
    __host__ __device__ static constexpr double eps() { return 2.220446049250313e-16; } // eps(Float64)
};
template <> struct Limits<float> {
    __host__ __device__ static constexpr float eps() { return 1.1920928955078125e-07f; } // eps(Float32)
};

template <typename FT> struct MathLibm;

template <> struct MathLibm<double> {
    static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double pow_neg3(double x) { return ::pow(x, -3.0); }
};
template <> struct MathLibm<float> {
    static __device__ __forceinline__ float pow(float x, float y) { return ::powf(x, y); }
    static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
    static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return 1.0f / x; }
    static __device__ __forceinline__ float pow_neg3(float x) { return ::powf(x, -3.0f); }
};

// Production policy: replaced below once the custom kernels are validated.
template <typename FT> struct MathFast : MathLibm<FT> {};

} // namespace lh

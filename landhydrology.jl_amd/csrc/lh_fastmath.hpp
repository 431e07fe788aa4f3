// lh_fastmath.hpp -- math policies for the soil closures on gfx950.
//
//  MathLibm<FT>  ocml pow/exp: <= 1 ulp, the reference-faithful policy used for
//                parity debugging (LH_MATH_LIBM) and once-per-column constants.
//  MathFast<FT>  the production policy.
//     double: pow(x, y) = exp2(y * log2 x) with LDS-resident tables --
//        log2: x = m 2^e (v_frexp), 2048 intervals on m in [0.5, 1) give
//              (1/c, log2 c) from LDS, r = fma(m, 1/c, -1) with |r| <= 2^-12 and a
//              degree-4 polynomial; c = 0.5 and c = 1 at the two ends so
//              log2(1) == 0 exactly and arguments next to 1 keep their relative accuracy
//              (3 ulp there, where log2 itself amplifies the argument's rounding 4096-fold);
//        exp2: t = k/2048 + r, 2^(j/2048) from LDS, degree-3 polynomial, v_ldexp.
//        11 VALU instructions per log2 or exp2 against several hundred for ocml's
//        correctly-rounded pow: CDNA4 has no f64 transcendental unit, and once the
//        launch streams only the planes it must (known-zero planes) the f64 closures --
//        not HBM -- set the kernel's speed: every polynomial term is launch time.  The
//        tables take 48 KiB of LDS per workgroup, so the Float64 kernels run 1024- or
//        512-thread workgroups (2 or 3 per CU).
//        Error: |rel| <= (2.5 |y log2 x| ln 2 + 2) 2^-53, i.e. the rounding of
//        the exponent product; the parity tolerance model accounts for it.
//     float: v_log_f32 / v_exp_f32 (hardware, ~1 ulp each).
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

// ---------------------------------------------------------------- tables
constexpr int LOG_TAB_N = 2048; // entries of (1/c, log2 c)
constexpr int LOG_TAB_BITS = 11;
constexpr int EXP_TAB_N = 2048; // entries of 2^(j/2048)
constexpr int EXP_TAB_BITS = 11;
constexpr int MATH_TAB_DOUBLES = 2 * LOG_TAB_N + EXP_TAB_N;

// Shared-memory image of the tables: [0, 2 LOG_TAB_N) log pairs, then EXP_TAB_N exp2 values.
struct MathTables {
    const double* log_tab; // LDS, pairs
    const double* exp_tab; // LDS
};

// ------------------------------------------------------------------ libm
template <typename FT> struct MathLibm;

template <> struct MathLibm<double> {
    static constexpr bool uses_tables = false;
    static constexpr bool is_production = false;
    __device__ __forceinline__ explicit MathLibm(const MathTables&) {}
    __device__ __forceinline__ MathLibm() {}
    static __device__ __forceinline__ double pow(double x, double y) { return ::pow(x, y); }
    static constexpr double EXP2_SCALE = 1.0;
    static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
    static __device__ __forceinline__ double exp2(double x) { return ::exp2(x); }
    static __device__ __forceinline__ double exp2_scaled(double x) { return ::exp2(x); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double sqrt_mul(double x, double f) { return ::sqrt(x) * f; }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double pow_neg3(double x) { return ::pow(x, -3.0); }
};
template <> struct MathLibm<float> {
    static constexpr bool uses_tables = false;
    static constexpr bool is_production = false;
    __device__ __forceinline__ explicit MathLibm(const MathTables&) {}
    __device__ __forceinline__ MathLibm() {}
    static __device__ __forceinline__ float pow(float x, float y) { return ::powf(x, y); }
    static constexpr float EXP2_SCALE = 1.0f;
    static __device__ __forceinline__ float log2(float x) { return ::log2f(x); }
    static __device__ __forceinline__ float exp2(float x) { return ::exp2f(x); }
    static __device__ __forceinline__ float exp2_scaled(float x) { return ::exp2f(x); }
    static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
    static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float sqrt_mul(float x, float f) { return ::sqrtf(x) * f; }
    static __device__ __forceinline__ float rcp(float x) { return 1.0f / x; }
    static __device__ __forceinline__ float pow_neg3(float x) { return ::powf(x, -3.0f); }
};

// ------------------------------------------------------------------ fast
template <typename FT> struct MathFast;

// (Horner steps stay plain __builtin_fma: forcing the coefficients into SGPRs with
// inline asm removes the v_mov_b64 copies hipcc makes, but it pads every asm
// statement with s_nop hazards -- 28 per cell against 8 saved moves.)
// A constant the compiler must keep in a VGPR pair for the kernel's lifetime
// instead of re-materialising it with v_mov_b64 (as costly as an FMA on gfx950)
// in front of every use: the empty asm makes its value opaque.
__device__ __forceinline__ double vgpr_resident(double k) {
    asm volatile("" : "+v"(k));
    return k;
}

template <> struct MathFast<double> {
    static constexpr bool uses_tables = true;
    static constexpr bool is_production = true;
    // exponents are handed to exp2 in units of 1/2048 (x2048): the argument reduction
    // is then three additions, with no multiply by 2048 and no v_mov/v_fmac pairs
    static constexpr double EXP2_SCALE = double(EXP_TAB_N);
    MathTables tb;
    // the polynomial coefficients the first Horner steps need next to a second constant, and the
    // rounding shifter, VGPR-resident: only one operand of a VOP3 may come from SGPRs/literals
    double c3v, q2v, shiftv;
    __device__ __forceinline__ explicit MathFast(const MathTables& t)
        : tb(t), c3v(vgpr_resident(0.48089834696298783)),
          q2v(vgpr_resident(5.727446255423176e-08)),
          shiftv(vgpr_resident(6755399441055744.0)) {}

    // log2 of a positive, finite, normal x
    __device__ __forceinline__ double log2_core(double x) const {
        const double m = __builtin_amdgcn_frexp_mant(x); // [0.5, 1)
        const int e = __builtin_amdgcn_frexp_exp(x);     // x = m 2^e
        const unsigned hi = (unsigned)__double2hiint(m);
        // byte offset of the 16-byte entry: the top LOG_TAB_BITS fraction bits (hi[19 .. 20-BITS])
        const unsigned off = (hi >> (20 - LOG_TAB_BITS - 4)) & ((LOG_TAB_N - 1) << 4);
        const double* ent = reinterpret_cast<const double*>(reinterpret_cast<const char*>(tb.log_tab) + off);
        const double invc = ent[0];
        const double l2c = ent[1];
        const double r = __builtin_fma(m, invc, -1.0);   // |r| <= 2^-12 (2^-11 in the first interval)
        // log2(1 + r) = r (c1 + r (c2 + r (c3 + r c4))),  c_k = (-1)^(k+1) / (k ln 2)
        double p = __builtin_fma(r, -0.36067376022224085, c3v); // c4 r + c3
        p = __builtin_fma(p, r, -0.72134752044448170);   // c2
        p = __builtin_fma(p, r, 1.4426950408889634);     // c1 = 1/ln2
        return __builtin_fma(r, p, (double)e + l2c);
    }

    // The exp2 table is stored BIASED: entry j holds 2^(j/2048) with (j << 9) subtracted from its
    // high word, so that adding (k << 9) = (e << 20) + (j << 9) for k = 2048 e + j puts the binary
    // exponent e in place with ONE 32-bit v_lshl_add_u32 (exp2_scaled_ins); the general form below
    // adds back (j << 9) = (byte offset << 6) and scales with v_ldexp_f64.
    // 2^(u/2048) for |u| < 2^50 (v_ldexp saturates to 0 / inf far inside that; the
    // closures' exponents are bounded by ~53 n/(n-1)); NaN in, NaN out.
    // k = rint(u) by the shifter trick (the low word of u + 1.5*2^52 is k), u - k in
    // [-1/2, 1/2], 2^(j/2048) from LDS, degree-3 polynomial in (u - k) with the 1/2048^k
    // folded into the coefficients (near-minimax: |error| < 9e-18 of the result).
    __device__ __forceinline__ double exp2_scaled(double u) const {
        const double sh = u + shiftv;
        const int k = __double2loint(sh);
        const double r = u - (sh - shiftv);                      // exact
        const unsigned off = ((unsigned)k << 3) & ((EXP_TAB_N - 1) << 3);
        const double tb_ = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tb.exp_tab) + off);
        const double tj = __hiloint2double(__double2hiint(tb_) + (int)(off << 6), __double2loint(tb_));
        const int e = k >> EXP_TAB_BITS;
        // 2^(r/2048) - 1 = r (q1 + r (q2 + r q3))
        double p = __builtin_fma(r, 6.461528679825916e-12, q2v);
        p = __builtin_fma(p, r, 0.0003384507717577858);
        const double res = __builtin_fma(tj, r * p, tj);
        return __builtin_amdgcn_ldexp(res, e);
    }
    // The same value when the caller GUARANTEES a normal result (|u / 2048| < 1021, u not NaN --
    // the exponent is inserted by integer addition: no saturation, no NaN propagation through it;
    // a NaN u still gives NaN because r = u - k is NaN): seven f64 and three 32-bit instructions.
    __device__ __forceinline__ double exp2_scaled_ins(double u) const {
        const double sh = u + shiftv;
        const int k = __double2loint(sh);
        const double r = u - (sh - shiftv);                      // exact
        const unsigned off = ((unsigned)k << 3) & ((EXP_TAB_N - 1) << 3);
        const double tb_ = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tb.exp_tab) + off);
        const double T = __hiloint2double(__double2hiint(tb_) + (int)((unsigned)k << 9), __double2loint(tb_)); // 2^(j/2048) 2^e
        double p = __builtin_fma(r, 6.461528679825916e-12, q2v);
        p = __builtin_fma(p, r, 0.0003384507717577858);
        return __builtin_fma(T, r * p, T);
    }
    // 2^(x y / 2048) under exp2_scaled_ins's contract: the product enters the rounding shifter and
    // the reduced argument through FMAs, i.e. unrounded and without an instruction of its own
    __device__ __forceinline__ double exp2_prod_ins(double x, double y) const {
        const double sh = __builtin_fma(x, y, shiftv);
        const int k = __double2loint(sh);
        const double r = __builtin_fma(x, y, -(sh - shiftv));
        const unsigned off = ((unsigned)k << 3) & ((EXP_TAB_N - 1) << 3);
        const double tb_ = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tb.exp_tab) + off);
        const double T = __hiloint2double(__double2hiint(tb_) + (int)((unsigned)k << 9), __double2loint(tb_));
        double p = __builtin_fma(r, 6.461528679825916e-12, q2v);
        p = __builtin_fma(p, r, 0.0003384507717577858);
        return __builtin_fma(T, r * p, T);
    }
    __device__ __forceinline__ double exp2_core(double t) const { return exp2_scaled(t * EXP2_SCALE); }

    // x^y with libm's results for the special bases the closures can produce:
    // x == 0 (0 or inf by the sign of y), x < 0 or NaN (NaN; the reference raises
    // DomainError there), x == inf.
    __device__ __forceinline__ double log2(double x) const { return log2_core(x); }
    __device__ __forceinline__ double exp2(double t) const { return exp2_core(t); }
    __device__ __forceinline__ double pow(double x, double y) const {
        const double xs = (x > 0.0 && x < __builtin_inf()) ? x : 1.0;
        double t = y * log2_core(xs);
        t = __builtin_fmin(__builtin_fmax(t, -1100.0), 1100.0);
        double res = exp2_core(t);
        if (x == 0.0) res = (y > 0.0) ? 0.0 : __builtin_inf();
        if (x == __builtin_inf()) res = (y > 0.0) ? __builtin_inf() : 0.0;
        if (!(x >= 0.0) || y != y) res = __builtin_nan("");
        return res;
    }
    __device__ __forceinline__ double exp(double x) const {
        double res = exp2_scaled(x * (1.4426950408889634 * EXP2_SCALE));
        return (x != x) ? x : res;
    }
    // sqrt of a positive normal x: v_rsq_f64 seed, one Newton step on g = sqrt(x) and a
    // final residual correction (no range scaling: the closures take sqrt(S),
    // S in [eps, 1)).  h = 1/(2 sqrt x) stays at the seed's accuracy: it only scales the
    // residual d, which is already below 2^-40 of g.
    static __device__ __forceinline__ double sqrt(double x) {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y;
        const double h = 0.5 * y;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        const double d = __builtin_fma(-g, g, x);
        return __builtin_fma(d, h, g);
    }
    // f sqrt(x) for a positive normal x: with y = rsq(x) (relative error < 2^-22), g = x y and
    // e = 1 - g y, sqrt(x) = g (1 + e/2 + 3 e^2/8 + O(e^3)); the correction is applied to the
    // product f g directly -- six instructions behind the seed, one less than sqrt() and a product
    static __device__ __forceinline__ double sqrt_mul(double x, double f) {
        const double y = __builtin_amdgcn_rsq(x);
        const double g = x * y;
        const double e = __builtin_fma(-g, y, 1.0);
        const double q = __builtin_fma(e, 0.375, 0.5);
        const double fg = f * g;
        return __builtin_fma(fg, q * e, fg);
    }
    // 1/x by v_rcp_f64 and two Newton steps: < 1 ulp for normal x, no
    // div_scale/div_fixup (the closures only take reciprocals of normal values)
    static __device__ __forceinline__ double rcp(double x) {
        double y = __builtin_amdgcn_rcp(x);
        double e = __builtin_fma(-x, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-x, y, 1.0);
        return __builtin_fma(y, e, y);
    }
    static __device__ __forceinline__ double pow_neg3(double x) {
        const double y = rcp(x);
        return y * y * y;
    }
};

template <> struct MathFast<float> {
    static constexpr bool uses_tables = false;
    static constexpr bool is_production = true;
    __device__ __forceinline__ explicit MathFast(const MathTables&) {}
    // v_log_f32 / v_exp_f32: log2(0) = -inf, log2(<0) = NaN, so the special bases
    // fall out of the hardware semantics (x = 0: y * -inf = -/+inf -> 0 / inf).
    static __device__ __forceinline__ float pow(float x, float y) {
        float res = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
        if (x == 1.0f || y == 0.0f) res = 1.0f; // keep 1^y and x^0 exact
        return res;
    }
    static constexpr float EXP2_SCALE = 1.0f;
    static __device__ __forceinline__ float log2(float x) { return __builtin_amdgcn_logf(x); }
    static __device__ __forceinline__ float exp2(float t) { return __builtin_amdgcn_exp2f(t); }
    static __device__ __forceinline__ float exp2_scaled(float t) { return __builtin_amdgcn_exp2f(t); }
    static __device__ __forceinline__ float exp(float x) {
        return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
    }
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float sqrt_mul(float x, float f) { return __builtin_amdgcn_sqrtf(x) * f; }
    // v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE division sequence
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float pow_neg3(float x) {
        const float y = __builtin_amdgcn_rcpf(x);
        return y * y * y;
    }
};

// fused multiply-add in the working type (plain __builtin_fma is the double one)
__device__ __forceinline__ double fma_ft(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_ft(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

} // namespace lh

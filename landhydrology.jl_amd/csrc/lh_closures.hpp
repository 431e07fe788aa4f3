// lh_closures.hpp -- pointwise soil closures on the device (gfx950).
//
// The formulas are those of src/SoilModel/SoilWaterParameterizations.jl and
// src/SoilModel/SoilHeatParameterizations.jl in the reference, evaluated per
// lane (lane = soil column).  `M` is the math policy: MathLibm (ocml pow/exp,
// <= 1 ulp) or MathFast (lh_fastmath.hpp, table-free log2/exp2 kernels tuned
// for gfx950).
#pragma once
#include "lh_device.hpp"
#include "lh_fastmath.hpp"

namespace lh {

// exponent multipliers in the exp2 unit of the math policy (see ColC)
template <typename FT>
__host__ __device__ inline void set_scaled_exponents(ColC<FT>& c, FT scale) {
    c.e_one = scale;
    c.e_inv_m = scale * c.inv_m;
    c.e_m = scale * c.m;
    c.e_inv_n = c.inv_n; // applied to an already scaled difference
    c.e_log2_alpha = scale * c.log2_alpha;
}

// ColC::vg_fast: every power the water closures form for this column stays a normal Float64 number,
// so 2^(.) may put its exponent in place by integer addition (MathFast<double>::exp2_scaled_ins):
// |log2 S| <= 52 + |log2(nu - theta_r)| <= 69, S^(1/m) >= 2^(-69/0.07) = 2^-986, and psi's exponent
// (log2 w - log2 S / m) / n - log2 alpha <= 986 / 1.075 + 67 < 1023.
template <typename FT>
__host__ __device__ inline void set_fast_vg(ColC<FT>& c, FT alpha) {
    const FT por = c.nu - c.theta_r;
    // (every comparison false for NaN parameters: those columns take the v_ldexp form too)
    c.vg_fast = (c.m >= FT(LH_VG_FAST_MIN_M) && c.m < FT(1) && por >= FT(1e-5) && por <= FT(1) &&
                 alpha > FT(1e-20) && alpha < FT(1e20)) ? 1 : 0;
    c.pad_ = 0;
}

// A non-positive saturation (nu <= theta_r) makes `^` raise DomainError in the
// reference; here the column's K and psi become NaN (and the status flag is set)
template <typename FT>
__host__ __device__ inline void poison_invalid(ColC<FT>& c) {
    if (!(c.nu > c.theta_r)) {
        c.Ksat = FT(NAN);
        c.cgw = FT(NAN);
        c.inv_S_s = FT(NAN);
        c.log2_alpha = FT(NAN);
        c.e_log2_alpha = FT(NAN);
        c.alpha_pnn = FT(NAN);
    }
}

template <typename FT, typename M>
__device__ __forceinline__ ColC<FT> make_colc(const DevParams<FT>& P, int64_t col, bool percol) {
    if (!percol) return P.uc;
    ColC<FT> c;
    FT n = P.vg_n, alpha = P.vg_alpha;
    c.theta_r = P.vg_theta_r;
    c.Ksat = P.vg_Ksat;
    c.nu = P.nu;
    c.S_s = P.S_s;
    {
        if (P.pc[PC_VG_N]) n = P.pc[PC_VG_N][col];
        if (P.pc[PC_VG_ALPHA]) alpha = P.pc[PC_VG_ALPHA][col];
        if (P.pc[PC_VG_THETA_R]) c.theta_r = P.pc[PC_VG_THETA_R][col];
        if (P.pc[PC_VG_KSAT]) c.Ksat = P.pc[PC_VG_KSAT][col];
        if (P.pc[PC_NU]) c.nu = P.pc[PC_NU][col];
        if (P.pc[PC_S_S]) c.S_s = P.pc[PC_S_S][col];
    }
    c.n = n;
    c.inv_n = FT(1) / n;
    c.m = FT(1) - FT(1) / n; // vanGenuchten constructor, SoilWaterParameterizations.jl:167
    c.inv_m = FT(1) / c.m;
    c.alpha_pnn = MathLibm<FT>::pow(alpha, -n); // once per column
    c.theta_lim = c.theta_r + Limits<FT>::eps();
    c.inv_por = FT(1) / (c.nu - c.theta_r);
    c.inv_S_s = FT(1) / c.S_s;
    c.inv_nu = FT(1) / c.nu;
    c.log2_alpha = MathLibm<FT>::log2(alpha);
    c.cgw = P.cg2 * c.Ksat;
    set_fast_vg(c, alpha);
    set_scaled_exponents(c, FT(M::EXP2_SCALE));
    poison_invalid(c);
    // k_dry, SoilHeatParameterizations.jl:268-270, 280-294
    FT rho_b = (FT(1) - c.nu) * P.rho_p;
    FT num = (P.kappa_dry_parameter * P.kappa_solid - P.k_air) * rho_b + P.k_air * P.rho_p;
    FT den = P.rho_p - (FT(1) - P.kappa_dry_parameter) * rho_b;
    c.k_dry = num / den;
    return c;
}

// volumetric_liquid_fraction, SoilWaterParameterizations.jl:180-187
template <typename FT>
__device__ __forceinline__ FT liquid_fraction(FT vl, FT nu_eff) {
    return (vl < nu_eff) ? vl : nu_eff;
}

// K and psi of one cell: right_hand_side.jl:156-167 / :308-313 with
// effective_saturation (:212-216), hydraulic_conductivity (:268-281),
// pressure_head (:228-241), matric_potential (:195-199), the conductivity
// factors (:76-126).  K uses the true porosity nu, psi uses nu_eff = nu - ti.
template <typename FT, typename M, bool FACTORS, bool WANT_PSI = true>
__device__ __forceinline__ void water_closures_pow(const M& mm, const DevParams<FT>& P,
                                               const ColC<FT>& c, FT vl, FT ti, FT T, FT& K,
                                               FT& psi) {
    const FT nu_eff = c.nu - ti;
    const FT vls = !(vl <= c.theta_lim) ? vl : c.theta_lim; // max(vl, theta_r + eps), NaN kept
    const FT num = vls - c.theta_r;
    const FT S = num * c.inv_por; // / (nu - theta_r), reciprocal formed once per column
    // when ti == 0, nu_eff == nu bitwise and the two saturations coincide
    const FT Se = (nu_eff == c.nu) ? S : num * mm.rcp(nu_eff - c.theta_r);

    FT Kr;
    FT t_S = FT(0); // S^(1/m), reused by psi when Se == S
    if (S < FT(1)) {
        t_S = mm.pow(S, c.inv_m);
        FT inner = FT(1) - mm.pow(FT(1) - t_S, c.m);
        Kr = mm.sqrt(S) * (inner * inner); // (.)^FT(2)
    } else {
        Kr = FT(1);
    }
    K = Kr * c.Ksat;
    if (FACTORS) {
        FT visc = FT(1), imp = FT(1);
        if (P.viscosity_kind) visc = mm.exp(P.gamma * (T - P.T_ref_visc));
        if (P.impedance_kind) {
            FT tl = liquid_fraction(vl, nu_eff);
            FT f_i = ti / (tl + ti);
            // FT(10.0^(-Omega*f_i)): Float64 power, rounded to FT (:89-93)
            imp = FT(MathLibm<double>::pow(10.0, double(-P.Omega * f_i)));
        }
        K = K * visc * imp;
    }
    if (WANT_PSI) {
        if (Se <= FT(1)) {
            // S^(-1/m): share the power with K when the saturations coincide
            FT r = (Se == S && S < FT(1)) ? mm.rcp(t_S) : mm.pow(Se, -c.inv_m);
            psi = -mm.pow((r - FT(1)) * c.alpha_pnn, c.inv_n);
        } else {
            psi = (vl - nu_eff) * c.inv_S_s;
        }
    }
}

// The same closures in the log2 domain -- the production form.
//   L = log2 S,  t = S^(1/m) = 2^(L/m),  w = 1 - t,  Lw = log2 w,
//   K_r = sqrt(S) (1 - 2^(m Lw))^2,
//   S^(-1/m) - 1 = w / t  =>  psi = -2^((Lw - L/m)/n - log2 alpha)
// i.e. two logarithms and three exponentials per cell instead of four pows and
// a reciprocal, and no second cancellation in S^(-1/m) - 1.  With ice (Se != S)
// psi takes its own log2(Se), 2^(.), log2(1 - .).
// WANT_DPSI additionally returns dpsi/dvl for the stable-step bound: with
// u = S^(-1/m) - 1 = w/t the van Genuchten slope |psi| (u+1)/(n m u Se (nu_eff - theta_r))
// collapses to |psi| / (n m w (vl_safe - theta_r)); 1/S_s when saturated.  It is returned TIMES
// n m (a column constant the caller divides out once per column) and in Float32 whatever FT is
// (slope32): the bound is a safety estimate under a Courant factor <= 1/2, seven digits are
// plenty, and a Float64 reciprocal alone costs nine issue slots per cell.
// NOICE: the ice plane of the state is known to be all zeros (lh_state zero bits): ti is the
// literal 0, nu_eff == nu, and the separate psi saturation never exists -- the same numbers as
// the general path produces for ti == 0, with the ice code compiled out.
// -psi / q in Float32 (v_cvt_f32_f64 x2, v_rcp_f32, v_mul_f32), kept finite:
// a bone-dry cell of a clay-like soil has |psi| beyond the Float32 range while its K underflows to 0,
// and 0 x Inf would poison the column's maximum (the value is non-negative: an integer minimum of the
// bit patterns clamps Inf -- and a NaN -- to FLT_MAX)
template <typename FT>
__device__ __forceinline__ float slope32(FT npsi, FT q) { // npsi = -psi
    const float s = float(npsi) * __builtin_amdgcn_rcpf(float(q));
    return __builtin_bit_cast(float, __builtin_elementwise_min(__builtin_bit_cast(int, s), 0x7f7fffff));
}

// log2(nu - theta_r) by the math policy's own log2: with it log2 Se = log2 S + (l2_por -
// log2(nu_eff - theta_r)) is EXACTLY log2 S for a lane without ice (nu_eff == nu bitwise), so the
// per-lane psi chain of a wave with ice reproduces the shared chain bit for bit on its ice-free lanes
template <typename FT, typename M>
__device__ __forceinline__ void finish_colc(const M& mm, ColC<FT>& c) {
    const FT por = c.nu - c.theta_r;
    c.l2_por = (M::is_production && por > FT(0)) ? mm.log2(por) : FT(0);
}

// h_hi - h_lo of two vertically adjacent cell centres (h = psi + z, the hydraulic head of
// right_hand_side.jl:141 / :333) from the NEGATED potentials the closures hand out (NEGPSI): the grid
// is uniform (make_grid), so z_hi - z_lo is the constant dz -- the heads themselves are never formed,
// and |psi| >> z does not cost the gravity term its low bits
template <typename FT>
__device__ __forceinline__ FT head_difference(FT npsi_hi, FT npsi_lo, FT dz) {
    return (npsi_lo - npsi_hi) + dz; // (-a) - (-b) = b - a exactly
}

// every active lane of the wave has the predicate (one s_and/s_cmp on the ballot)
__device__ __forceinline__ bool wave_all(bool pred) {
    return __builtin_amdgcn_ballot_w64(!pred) == 0ull;
}

// RELK: K is returned WITHOUT the factor Ksat (relative conductivity times the conductivity factors);
// the column kernels fold Ksat into the per-column flux constant instead of multiplying every cell.
// BRANCHY: the saturated cells are a divergent branch around the power chain (few live registers:
// the Float64 coupled kernels are register-bound) instead of every lane running the chain and a
// wave-level repair (fewest instructions: the Richards kernels are issue-bound).  Same values.
// NEGPSI: `psi` receives -psi (>= 0 where unsaturated).  Every branch below forms -psi anyway (it is
// the 2^(.) itself), and a caller that only differences psi (head_difference) or negates once per
// column (boundary faces) then never spends an instruction per cell on the sign.
template <typename FT, typename M, bool FACTORS, bool WANT_PSI = true, bool WANT_DPSI = false, bool NOICE = false, bool RELK = false,
          bool BRANCHY = false, bool NEGPSI = false>
__device__ __forceinline__ void water_closures_log(const M& mm, const DevParams<FT>& P,
                                                   const ColC<FT>& c, FT vl, FT ti, FT T, FT& K,
                                                   FT& psi, float* dpsi = nullptr, bool vgfast = false) {
    constexpr FT SC = FT(M::EXP2_SCALE);
    const FT nu_eff = NOICE ? c.nu : c.nu - ti;
    // max(vl, theta_r + eps), NaN kept.  Float64: a bone-dry cell is rare, so the clamp is a real
    // (wave-level) branch -- one compare per cell on the hot path instead of a compare and two
    // selects; the empty asm keeps the compiler from turning it back into selects.
    // (the clamped lanes' difference (theta_r + eps) - theta_r is the column constant the reference's
    // own subtraction gives: nothing but `num` depends on the clamped value)
    FT num;
    if constexpr (M::uses_tables) {
        num = vl - c.theta_r;
        if (__builtin_amdgcn_ballot_w64(vl <= c.theta_lim) != 0ull) {
            num = (vl <= c.theta_lim) ? c.theta_lim - c.theta_r : num;
            asm volatile("" : "+v"(num));
        }
    } else {
        const FT vls = !(vl <= c.theta_lim) ? vl : c.theta_lim;
        num = vls - c.theta_r;
    }
    const FT S = num * c.inv_por;
    const bool same = NOICE || (nu_eff == c.nu); // no ice: the two saturations coincide bitwise
    const bool unsat = S < FT(1);
    // psi's own saturation Se = num / (nu_eff - theta_r) is never formed: Se < 1 <=> num < por_e,
    // and log2 Se = log2 S + (log2(nu - theta_r) - log2 por_e) (one log2 instead of a reciprocal,
    // a product and a log2)
    const FT por_e = nu_eff - c.theta_r;
    const bool unsat_e = (same & unsat) | (!same & (num < por_e)); // (mask logic: no selects)
    // ONE decision per wave: without any ice lane K and psi share t = S^(1/m) and w = 1 - t; with
    // one, every lane runs psi from its own saturation -- a lane-level choice would make a mixed
    // wave run both chains one after the other.  Per-lane results do not depend on the choice
    // (see finish_colc), so they do not depend on which cells share a wave either.
    const bool shared = NOICE || wave_all(same);
    // (Float64 production math) vgfast: the host found every column's powers safely inside the normal
    // range (ColC::vg_fast), so each 2^(.) puts its exponent in place by integer addition
    // (exp2_scaled_ins) -- bitwise the v_ldexp form whenever that one's result is a normal number, so
    // the two psi chains below still agree bit for bit on a lane without ice.
    bool ins = false;
    if constexpr (M::uses_tables) ins = vgfast;

    // exponents are formed in the exp2 unit of the policy (c.e_* carry the scale)
    auto ex2 = [&](FT u) -> FT {
        if constexpr (M::uses_tables) {
            if (ins) return mm.exp2_scaled_ins(u);
        }
        return mm.exp2_scaled(u);
    };
    auto ex2_prod = [&](FT x, FT y) -> FT { // 2^(x y): the product goes into the reduction unrounded
        if constexpr (M::uses_tables) {
            if (ins) return mm.exp2_prod_ins(x, y);
        }
        return mm.exp2_scaled(x * y);
    };
    FT L, Kb;
    FT npsi = FT(0); // -psi
    if constexpr (BRANCHY) {
        L = FT(0); // log2 S of an unsaturated cell (0 otherwise: K_r = 1, and sqrt(S) = 2^0 below)
        if (unsat) {
            L = mm.log2(S);
            const FT a = L * c.e_inv_m;                    // scale * log2 S^(1/m)
            const FT w = FT(1) - ex2(a);
            const FT Lw = mm.log2(w);
            const FT inner = FT(1) - ex2_prod(Lw, c.e_m);
            Kb = FACTORS ? inner * inner : mm.sqrt_mul(S, inner * inner);
            if (!RELK) Kb = Kb * c.Ksat;
            if (WANT_PSI && shared) {
                npsi = ex2(fma_ft(fma_ft(Lw, c.e_one, -a), c.e_inv_n, -c.e_log2_alpha));
                if (WANT_DPSI) *dpsi = slope32<FT>(npsi, w * num);
            }
        } else {
            // (saturated cells are rare: the empty asm keeps this a real branch -- otherwise the
            // compiler evaluates it for every cell and selects)
            Kb = RELK ? FT(1) : c.Ksat; // K_r = 1
            if (WANT_PSI && shared) {
                FT vin = vl; // (opaque INPUT: nothing of the saturated evaluation can move above the branch)
                asm volatile("" : "+v"(vin));
                npsi = (S == FT(1)) ? FT(0) : (nu_eff - vin) * c.inv_S_s; // -((vl - nu_eff)/S_s), exactly
                if (WANT_DPSI) *dpsi = float(c.n * c.m * c.inv_S_s);
            }
        }
    } else {
        // EVERY lane runs the unsaturated chain (saturated cells are rare: their lanes compute values
        // nobody uses -- table offsets are masked, nothing traps) and a wave with a saturated cell
        // repairs those lanes in a real branch below: no divergent-branch bookkeeping per cell.
        L = mm.log2(S);                                // log2 S
        const FT a = L * c.e_inv_m;                    // scale * log2 S^(1/m)
        const FT w = FT(1) - ex2(a);
        const FT Lw = mm.log2(w);
        const FT inner = FT(1) - ex2_prod(Lw, c.e_m);
        Kb = FACTORS ? inner * inner : mm.sqrt_mul(S, inner * inner); // K without the conductivity factors (FACTORS: without sqrt(S) too)
        if (!RELK) Kb = Kb * c.Ksat;
        if (WANT_PSI && shared) {
            npsi = ex2(fma_ft(fma_ft(Lw, c.e_one, -a), c.e_inv_n, -c.e_log2_alpha));
            if (WANT_DPSI) *dpsi = slope32<FT>(npsi, w * num);
        }
        if (__builtin_amdgcn_ballot_w64(!unsat) != 0ull) { // (a NaN saturation lands here too, as in the reference's `S < 1 ? ... : ...`)
            FT vin = vl; // (opaque: the compiler must not turn this block into per-cell selects)
            asm volatile("" : "+v"(vin));
            L = unsat ? L : FT(0);   // K_r = 1, and sqrt(S) = 2^0 in the factor product below
            Kb = unsat ? Kb : (RELK ? FT(1) : c.Ksat);
            if (WANT_PSI && shared) {
                const FT nps = (S == FT(1)) ? FT(0) : (nu_eff - vin) * c.inv_S_s; // -(vl - nu_eff)/S_s, exactly
                npsi = unsat ? npsi : nps;
                if (WANT_DPSI) *dpsi = unsat ? *dpsi : float(c.n * c.m * c.inv_S_s);
            }
        }
    }
    if (!NOICE && WANT_PSI && !shared) { // ice somewhere in the wave: psi from every lane's own saturation
        if (unsat_e) {
            // (unsat_e without unsat needs theta_i < 0: out of contract, L is 0 there)
            const FT Le = L + (c.l2_por - mm.log2(por_e)); // log2 Se; == L bitwise for a lane without ice
            const FT ae = Le * c.e_inv_m;
            const FT we = FT(1) - ex2(ae);
            const FT Lwe = mm.log2(we);
            npsi = ex2(fma_ft(fma_ft(Lwe, c.e_one, -ae), c.e_inv_n, -c.e_log2_alpha));
            if (WANT_DPSI) *dpsi = slope32<FT>(npsi, we * num);
        } else {
            const bool one = (same & (S == FT(1))) | (!same & (num == por_e));
            FT vin = vl;
            asm volatile("" : "+v"(vin));
            FT nps = one ? FT(0) : (nu_eff - vin) * c.inv_S_s;
            if (por_e < FT(0)) nps = FT(NAN); // Se < 0: `^` raises DomainError in the reference
            npsi = nps;
            if (WANT_DPSI) *dpsi = float(c.n * c.m * c.inv_S_s);
        }
    }
    if (FACTORS) {
        // exp(gamma (T - T_ref)) 10^(-Omega f_i) sqrt(S) as ONE 2^(.): the three exponents add
        FT ex = FT(0);
        if (P.viscosity_kind) ex = (P.gamma * (T - P.T_ref_visc)) * FT(1.4426950408889634);
        if (P.impedance_kind) {
            const FT tl = liquid_fraction(vl, nu_eff);
            const FT f_i = ti * mm.rcp(tl + ti); // 0/0 = NaN as in the reference
            ex = fma_ft(-P.Omega * f_i, FT(3.3219280948873623), ex);
        }
        K = Kb * mm.exp2_scaled(fma_ft(L, SC * FT(0.5), ex * SC));
    } else {
        K = Kb;
    }
    if (WANT_PSI) psi = NEGPSI ? npsi : -npsi;
    // (nu <= theta_r, a DomainError in the reference, is poisoned per column in make_colc)
}

template <typename FT, typename M, bool FACTORS, bool WANT_PSI = true, bool WANT_DPSI = false, bool NOICE = false, bool RELK = false,
          bool BRANCHY = false, bool NEGPSI = false>
__device__ __forceinline__ void water_closures(const M& mm, const DevParams<FT>& P,
                                               const ColC<FT>& c, FT vl, FT ti, FT T, FT& K,
                                               FT& psi, float* dpsi = nullptr, bool vgfast = false) {
    static_assert(!RELK || M::is_production, "the relative-conductivity form exists for the production math only");
    if (M::is_production) {
        water_closures_log<FT, M, FACTORS, WANT_PSI, WANT_DPSI, NOICE, RELK, BRANCHY, NEGPSI>(mm, P, c, vl, ti, T, K, psi, dpsi, vgfast);
    } else {
        water_closures_pow<FT, M, FACTORS, WANT_PSI>(mm, P, c, vl, ti, T, K, psi);
        if (WANT_DPSI) { // as the oracle writes it
            const FT nu_eff = c.nu - ti;
            const FT vls = !(vl <= c.theta_lim) ? vl : c.theta_lim;
            const FT Se = (vls - c.theta_r) / (nu_eff - c.theta_r);
            const FT u = mm.pow(Se, -c.inv_m) - FT(1);
            if (Se <= FT(1) && u > FT(0))
                *dpsi = float(fabs(psi) * (u + FT(1)) / (u * Se * (nu_eff - c.theta_r)));
            else
                *dpsi = float(c.n * c.m / c.S_s);
        }
        if (NEGPSI && WANT_PSI) psi = -psi;
    }
}

// T, kappa (and rho_c_s) of one cell: right_hand_side.jl:291-305 with
// volumetric_heat_capacity (:65-79), temperature_from_rhoe_int (:42-53),
// relative_saturation (:139-142), kersten_number (:152-174),
// saturated_thermal_conductivity (:114-128), thermal_conductivity (:185-188).
// The production form of kappa: Kersten number and saturated conductivity in the
// log2 domain.  One log2(S_r) serves the frozen and the unfrozen exponent (a wave
// with both kinds of lanes no longer evaluates two pows), and
// kappa_su^(tl/tw) kappa_sf^(ti/tw) = 2^((tl log2 kappa_su + ti log2 kappa_sf)/tw)
// is one exp2 instead of two pows.
template <typename FT, typename M, bool NOICE = false>
__device__ __forceinline__ FT kappa_closure_log(const M& mm, const DevParams<FT>& P,
                                                const ColC<FT>& c, FT vl, FT ti) {
    constexpr FT SC = FT(M::EXP2_SCALE);
    const FT nu_eff = NOICE ? c.nu : c.nu - ti;
    const FT tl = liquid_fraction(vl, nu_eff);
    const FT tw = NOICE ? tl : tl + ti;
    const FT S_r = tw * c.inv_nu;
    const bool unfrozen = NOICE || ti < Limits<FT>::eps();
    // S_r^e: 0 for S_r == 0 (e > 0), NaN for S_r < 0 or NaN (DomainError in the reference).
    // The hardware Float32 log/exp give exactly that (log2 0 = -inf, 2^-inf = 0); the table
    // log2 of the Float64 policy needs a positive normal argument.
    const FT e_sel = unfrozen ? P.kersten_exp_unfrozen * SC : P.kersten_exp_frozen * SC;
    FT K_e;
    if constexpr (M::uses_tables) {
        K_e = mm.exp2_scaled(mm.log2(S_r > FT(0) ? S_r : FT(1)) * e_sel);
        K_e = (S_r > FT(0)) ? K_e : ((S_r == FT(0)) ? FT(0) : FT(NAN));
    } else {
        K_e = mm.exp2_scaled(mm.log2(S_r) * e_sel);
    }
    if (unfrozen) {
        const FT e = mm.exp2_scaled(S_r * P.neg_b_log2e_sc); // exp(-b S_r): -b log2(e) x the exp2 unit, one product
        const FT a = mm.pow_neg3(FT(1) + e);    // (1 + exp(-b S_r))^(-3)
        const FT h = (FT(1) - S_r) * FT(0.5);
        const FT d = a - h * h * h;             // ((1 - S_r)/2)^3
        // (.)^(1 - nu_om): the exponent is exactly 1 for soils without organic matter.  A real
        // (uniform) branch: the Float32 compiler otherwise evaluates the power for every cell and
        // selects -- two transcendentals and four more instructions per cell for nothing
        FT dp = d;
        if (P.one_minus_om != FT(1)) {
            dp = mm.pow(d, P.one_minus_om);
            asm volatile("" : "+v"(dp));
        }
        K_e = K_e * dp;
    }
    FT k_sat = P.kappa_sat_unfrozen;            // tl/tw == 1 exactly when ti == 0
    if (!NOICE && ti != FT(0)) {
        const FT itw = mm.rcp(tw);
        k_sat = mm.exp2_scaled(((tl * (P.l2_kappa_sat_unfrozen * SC) + ti * (P.l2_kappa_sat_frozen * SC)) * itw));
    }
    k_sat = (tw < Limits<FT>::eps()) ? FT(0) : k_sat;
    return K_e * k_sat + (FT(1) - K_e) * c.k_dry;
}

template <typename FT, typename M, bool NOICE = false>
__device__ __forceinline__ FT kappa_closure(const M& mm, const DevParams<FT>& P,
                                            const ColC<FT>& c, FT vl, FT ti) {
    if constexpr (M::is_production) return kappa_closure_log<FT, M, NOICE>(mm, P, c, vl, ti);
    const FT nu_eff = c.nu - ti;
    const FT tl = liquid_fraction(vl, nu_eff);
    const FT tw = tl + ti;
    // production: reciprocals formed once per column / by the fast rcp
    const FT S_r = M::is_production ? tw * c.inv_nu : tw / c.nu;
    FT K_e;
    if (ti < Limits<FT>::eps()) {
        FT e = mm.exp(-P.b * S_r);
        FT a = mm.pow_neg3(FT(1) + e);          // (1 + exp(-b S_r))^(-3)
        FT h = M::is_production ? (FT(1) - S_r) * FT(0.5) : (FT(1) - S_r) / FT(2);
        FT d = a - h * h * h;                   // ((1 - S_r)/2)^3
        // (.)^(1 - nu_om): the exponent is exactly 1 for soils without organic matter
        FT dp = (M::is_production && P.one_minus_om == FT(1)) ? d : mm.pow(d, P.one_minus_om);
        K_e = mm.pow(S_r, P.kersten_exp_unfrozen) * dp;
    } else {
        K_e = mm.pow(S_r, P.kersten_exp_frozen);
    }
    FT k_sat;
    if (tw < Limits<FT>::eps()) {
        k_sat = FT(0);
    } else if (ti == FT(0)) {
        // kappa_unf^(tl/tw) * kappa_fr^0 with tl/tw == 1 exactly
        k_sat = P.kappa_sat_unfrozen;
    } else {
        const FT itw = M::is_production ? mm.rcp(tw) : FT(1) / tw;
        k_sat = mm.pow(P.kappa_sat_unfrozen, tl * itw) * mm.pow(P.kappa_sat_frozen, ti * itw);
    }
    return K_e * k_sat + (FT(1) - K_e) * c.k_dry;
}

template <typename FT, typename M, bool NOICE = false>
__device__ __forceinline__ FT temperature_closure(const M& mm, const DevParams<FT>& P,
                                                  const ColC<FT>& c, FT vl, FT ti, FT rhoe,
                                                  FT& rho_c_s) {
    const FT nu_eff = NOICE ? c.nu : c.nu - ti;
    const FT tl = liquid_fraction(vl, nu_eff);
    // (x + 0*y == x bitwise for the finite positive x here; rhoe + 0 == rhoe up to the sign of
    // a zero, which the division below does not see)
    rho_c_s = NOICE ? P.rho_c_ds + tl * P.rhocp_l : P.rho_c_ds + tl * P.rhocp_l + ti * P.rhocp_i;
    const FT num = NOICE ? rhoe : rhoe + ti * P.rho_i * P.LH_f0;
    return P.T_ref + (M::is_production ? num * mm.rcp(rho_c_s) : num / rho_c_s);
}

// boundary_fluxes for one face of one column: boundary_conditions.jl:470-489
// with :218-288 (centre/face pairs, Dirichlet overwrite) and the vertical_flux
// methods :295-444.  (vl_c, ti_c, T_c, K_c, psi_c) are the centre values next
// to the face; K_c/psi_c are the values the interior stage already computed
// (the reference recomputes them on a 2-element array: same numbers).
// Two parts: face_state evaluates the closures of the FACE state of a Dirichlet component (the only
// expensive part: a full water and/or kappa closure), boundary_fluxes_from assembles the two fluxes
// from it.  The persistent steppers evaluate the first part once per call when nothing it reads
// changes during the call (face_state_is_static).
// kinds and values of one face's boundary conditions.  `face` may be a per-lane value (the steppers'
// bottom and top lanes go through the boundary code together): selects between the two faces' uniform
// entries, not an indexed read of the by-value parameter block (which would put it into scratch memory).
template <typename FT>
struct FaceBC {
    int ke, kh; // kind of the energy / hydrology component
    FT ve, vh;  // value (per-column array element where one is set)
};
template <typename FT>
__device__ __forceinline__ FaceBC<FT> face_bc(const DevParams<FT>& P, int face, int64_t col) {
    const bool b = (face == FACE_BOTTOM);
    FaceBC<FT> r;
    r.ke = b ? P.bc_kind[FACE_BOTTOM][COMP_ENERGY] : P.bc_kind[FACE_TOP][COMP_ENERGY];
    r.kh = b ? P.bc_kind[FACE_BOTTOM][COMP_HYDROLOGY] : P.bc_kind[FACE_TOP][COMP_HYDROLOGY];
    r.ve = b ? P.bc_value[FACE_BOTTOM][COMP_ENERGY] : P.bc_value[FACE_TOP][COMP_ENERGY];
    r.vh = b ? P.bc_value[FACE_BOTTOM][COMP_HYDROLOGY] : P.bc_value[FACE_TOP][COMP_HYDROLOGY];
    const FT* pe = b ? P.bc_pc[FACE_BOTTOM][COMP_ENERGY] : P.bc_pc[FACE_TOP][COMP_ENERGY];
    const FT* ph = b ? P.bc_pc[FACE_BOTTOM][COMP_HYDROLOGY] : P.bc_pc[FACE_TOP][COMP_HYDROLOGY];
    if (pe) r.ve = pe[col];
    if (ph) r.vh = ph[col];
    return r;
}

template <typename FT>
struct FaceState {
    FT K, psi, kap, T; // K(face state), psi(face state), kappa(face state), T of the face state
};

template <typename FT, typename M, int MODEL, bool FACTORS, bool NOICE = false>
__device__ __forceinline__ FaceState<FT> face_state(const M& mm, const DevParams<FT>& P, const ColC<FT>& c,
                                                    int face, int64_t col, FT vl_c, FT ti_c, FT T_c,
                                                    bool vgfast = false) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    const FaceBC<FT> bc = face_bc(P, face, col);
    const int ke = bc.ke, kh = bc.kh;
    const FT ve = bc.ve, vh = bc.vh;
    FaceState<FT> fs;
    fs.K = fs.psi = fs.kap = FT(0);
    FT vl_f = vl_c; // face := centre (:218-228)
    fs.T = T_c;
    if (HEAT && ke == BC_DIRICHLET) fs.T = ve;
    if (WATER && kh == BC_DIRICHLET) vl_f = vh;
    if (HEAT && ke == BC_DIRICHLET) fs.kap = kappa_closure<FT, M, NOICE>(mm, P, c, vl_f, ti_c); // :416-444
    if (WATER && kh == BC_DIRICHLET)                                                            // :371-401
        water_closures<FT, M, FACTORS, true, false, NOICE>(mm, P, c, vl_f, ti_c, fs.T, fs.K, fs.psi, nullptr, vgfast);
    return fs;
}

// what face_state reads besides the boundary values: the centre's vartheta_l (when the hydrology
// component is not Dirichlet, for kappa), its theta_i (never changes) and its T (through the viscosity
// factor, when the energy component is not Dirichlet).  True when none of that changes while a model
// steps with constant boundary values -- the face state's closures are then constants of the call.
template <typename FT, int MODEL, bool FACTORS>
__device__ __forceinline__ bool face_state_is_static(const DevParams<FT>& P, int face) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    const bool bot = (face == FACE_BOTTOM);
    const int ke = bot ? P.bc_kind[FACE_BOTTOM][COMP_ENERGY] : P.bc_kind[FACE_TOP][COMP_ENERGY];
    const int kh = bot ? P.bc_kind[FACE_BOTTOM][COMP_HYDROLOGY] : P.bc_kind[FACE_TOP][COMP_HYDROLOGY];
    const bool needs_w = WATER && kh == BC_DIRICHLET, needs_k = HEAT && ke == BC_DIRICHLET;
    bool ok = true;
    if (needs_w && FACTORS && P.viscosity_kind && HEAT && ke != BC_DIRICHLET) ok = false; // T_f = T_c moves
    if (needs_k && WATER && kh != BC_DIRICHLET) ok = false;                                // vl_f = vl_c moves
    return ok;
}

// the T of the face state (the value face_state puts into FaceState::T)
template <typename FT, int MODEL>
__device__ __forceinline__ FT face_state_T(const DevParams<FT>& P, int face, int64_t col, FT T_c) {
    if (MODEL == MODEL_RICHARDS) return T_c;
    const FaceBC<FT> bc = face_bc(P, face, col);
    return bc.ke == BC_DIRICHLET ? bc.ve : T_c;
}

template <typename FT, int MODEL>
__device__ __forceinline__ void boundary_fluxes_from(const DevParams<FT>& P, const FaceState<FT>& fs, int face,
                                                     int64_t col, FT T_c, FT K_c, FT psi_c, FT& f_e, FT& f_w) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    const FaceBC<FT> bc = face_bc(P, face, col);
    const int ke = bc.ke, kh = bc.kh;
    const FT ve = bc.ve, vh = bc.vh;
    // (x / (dz/2) as x * (2 (1/dz)): the reciprocal is a constant of the grid, and a Float64 division is
    // a dozen instructions the steppers' boundary lanes would issue every stage with their waves waiting)
    const FT dzb = P.half_dz, inv_dzb = FT(2) * P.inv_dz;
    const FT sgn = (face == FACE_BOTTOM) ? FT(-1) : FT(1);
    f_e = FT(0);
    f_w = FT(0);
    if (HEAT) {
        if (ke == BC_FLUX) f_e = ve;
        else if (ke == BC_DIRICHLET) f_e = sgn * ((-fs.kap * (fs.T - T_c)) * inv_dzb); // :416-444
    }
    if (WATER) {
        if (kh == BC_FLUX) {
            f_w = vh;
        } else if (kh == BC_FREE_DRAINAGE) { // :328-356
            f_w = -K_c;
        } else if (kh == BC_DIRICHLET) { // :371-401
            if (face == FACE_BOTTOM && P.consistent_bottom_sign)
                f_w = (fs.K * (fs.psi - psi_c - dzb)) * inv_dzb;
            else
                f_w = sgn * ((-fs.K * (fs.psi - psi_c + dzb)) * inv_dzb);
        }
    }
}

template <typename FT, typename M, int MODEL, bool FACTORS, bool NOICE = false>
__device__ __forceinline__ void boundary_fluxes(const M& mm, const DevParams<FT>& P,
                                                const ColC<FT>& c, int face, int64_t col, FT vl_c,
                                                FT ti_c, FT T_c, FT K_c, FT psi_c, FT& f_e,
                                                FT& f_w, FT* K_face = nullptr,
                                                FT* kappa_face = nullptr, bool vgfast = false) {
    const FaceState<FT> fs = face_state<FT, M, MODEL, FACTORS, NOICE>(mm, P, c, face, col, vl_c, ti_c, T_c, vgfast);
    const bool bot = (face == FACE_BOTTOM);
    const int kh = bot ? P.bc_kind[FACE_BOTTOM][COMP_HYDROLOGY] : P.bc_kind[FACE_TOP][COMP_HYDROLOGY];
    const int ke = bot ? P.bc_kind[FACE_BOTTOM][COMP_ENERGY] : P.bc_kind[FACE_TOP][COMP_ENERGY];
    if (K_face && MODEL != MODEL_HEAT && kh == BC_DIRICHLET) *K_face = fs.K;
    if (kappa_face && MODEL != MODEL_RICHARDS && ke == BC_DIRICHLET) *kappa_face = fs.kap;
    boundary_fluxes_from<FT, MODEL>(P, fs, face, col, T_c, K_c, psi_c, f_e, f_w);
}

} // namespace lh

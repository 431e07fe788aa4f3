// lh_kernels_impl.hpp -- gfx950 (MI355X, CDNA4) kernels of the batched soil-column
// tendency path.  No MFMA: this is a bandwidth/VALU-bound vertical stencil.
//
// Data layout: every variable is a plane [nlev][stride] with the
// column index fastest, so a wavefront reads 64*CPL consecutive columns of one
// level in one fully coalesced instruction (CPL = 1 for Float64, 2 for Float32:
// 8 B per lane either way).  One lane owns CPL whole columns
// and marches bottom -> top, carrying K, -psi (T, kappa, rho_e_l K) of the previous
// cell and the previous face flux in registers: each face flux is computed once
// and differenced, which keeps the discrete mass/energy conservation the
// reference's equilibrium tests rely on (coupled.jl:117, richards_equation.jl:94).
#pragma once
#include "lh_closures.hpp"
#include "lh_launch.hpp"
#include "lh_atmos.hpp"
#include <type_traits>

namespace lh {

// minimum waves per SIMD the register allocator must leave room for in the
// column kernel (64 VGPRs): the kernel hides HBM latency by occupancy
#ifndef LH_RHS_WAVES_PER_SIMD
#define LH_RHS_WAVES_PER_SIMD 8
#endif
#ifndef LH_RHS_F64_WAVES
#define LH_RHS_F64_WAVES 6
#endif

// waves/SIMD the Float32 coupled tendency + step-bound launch is compiled for (86 VGPRs unconstrained)
#ifndef LH_F32C_DT_WAVES
#define LH_F32C_DT_WAVES 5
#endif
// ... and whether that launch keeps its column constants in VGPRs (as the plain tendency launch
// does) or in SGPRs at LH_F32C_DT_WAVES_SGPR waves/SIMD
// (measured on C3, fused-dt launch: VGPR constants at 5 waves 0.215 ms; SGPR constants at 8 waves
// with 12 B of scratch 0.224, at 7 waves 0.222)
#ifndef LH_F32C_DT_VGPRCONST
#define LH_F32C_DT_VGPRCONST 1
#endif
#ifndef LH_F32C_DT_WAVES_SGPR
#define LH_F32C_DT_WAVES_SGPR 8
#endif

// waves/SIMD of the per-column Richards tendency + step-bound launch (C5); measured: 6 waves (72 VGPRs)
// 0.511 ms, 8 waves (64 VGPRs + 20 B of scratch) 0.537 ms
#ifndef LH_PERCOL_DT_WAVES
#define LH_PERCOL_DT_WAVES 6
#endif

// The plain production kernels are held to 64 VGPRs (8 waves/SIMD); the
// per-column / conductivity-factor variants and the libm debug policy keep what
// they need (a bound there only produces scratch spills).
// the coupled tendency launches (MODE 0 and 4; Float32 without factors, Float64 always) keep
// their column constants in VGPRs
template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename M, int MODE>
constexpr bool f32_coupled_vgpr_constants() {
    if (MODEL != MODEL_COUPLED || PERCOL || !M::is_production || !(MODE == 0 || MODE == 4)) return false;
    if (sizeof(FT) == 4 && MODE == 4) return LH_F32C_DT_VGPRCONST != 0;
    // Float64: without conductivity factors only.  (Round 2 pinned the factors kernels too, +2.5 % at 127
    // VGPRs + 12 B of scratch; with the round-3 closures the pinned tendency kernel needs 154 VGPRs --
    // 3 waves, 0.61 ms on f3c64 -- or spills 76 B at 4 waves, 0.67 ms; unpinned it fits 119 VGPRs at 4
    // waves without scratch and takes 0.58 ms, v_readlane reloads of spilled uniforms and all.)
    return !FACTORS;
}

template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename M, bool NOICE>
constexpr bool heat_vgpr_constants() {
    return sizeof(FT) == 8 && MODEL != MODEL_RICHARDS && M::is_production && !NOICE;
}

template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename M, int PF, int MODE, bool NOICE = false, int CPL = 1>
constexpr int rhs_waves_per_simd() {
    if (sizeof(FT) == 4 && CPL >= 4) return 4; // (tuning builds: four Float32 columns per lane, 16-byte accesses)
    // Float64 heat kernels with conductivity factors: 127 VGPRs (4 waves/SIMD, 512-thread workgroups)
    // with the uniform constants left to the SGPR file and its spills beats 156 VGPRs with the
    // constants VGPR-resident at 3 waves (f3c64 tendency 0.530 vs 0.554 ms); the step-bound launch
    // would spill to scratch under that bound and keeps the other arrangement
    // (Richards with a conductivity factor: 66 VGPRs unconstrained = 6 waves; bounded to 64 for 8 waves it
    // spills SGPRs to lanes and 12 B to scratch: 0.324 vs 0.307 ms on f3v64 -- left unconstrained)
#ifndef LH_F64_FACTORS_STAGE_WAVES4
#define LH_F64_FACTORS_STAGE_WAVES4 0
#endif
#ifndef LH_F64_FACTORS_TEND_WAVES
#define LH_F64_FACTORS_TEND_WAVES 4
#endif
    // (round 3: with its column constants pinned in VGPRs the tendency kernel needs 154; they are left to
    // the SGPR file now, see f32_coupled_vgpr_constants)
    if (M::is_production && FACTORS && sizeof(FT) == 8 && MODEL != MODEL_RICHARDS && !PERCOL &&
        (MODE == 0 || (LH_F64_FACTORS_STAGE_WAVES4 && MODE != 4))) return LH_F64_FACTORS_TEND_WAVES;
    if (!M::is_production || FACTORS) return 1;
    // Float64 Richards without ice fits 64 VGPRs (8 waves) in every mode, but asked for only 6 waves (<= 80 VGPRs,
    // 512-thread workgroups, 3 per CU) the max-ILP scheduler uses the room: tendency -1 %, with the step bound -2 %,
    // C5 -1.6 %, the Dirichlet ensemble C1 -4 %, fused stages equal (round 3, same-process A/B).  Float32 keeps 8
    // (its fused stages lose 5 % at 6).
    if (MODEL == MODEL_RICHARDS && NOICE && sizeof(FT) == 8 && !PERCOL) return LH_RHS_F64_WAVES;
    if (MODEL == MODEL_RICHARDS && NOICE && PERCOL) return MODE == 4 ? LH_PERCOL_DT_WAVES : (sizeof(FT) == 8 ? LH_RHS_F64_WAVES : LH_RHS_WAVES_PER_SIMD); // 62 VGPRs (72 with the dt bound)
    if (PERCOL) return 1;
    if (MODEL == MODEL_RICHARDS && NOICE) return LH_RHS_WAVES_PER_SIMD; // no ice ring: fits 64 VGPRs in every mode
    if (MODEL == MODEL_RICHARDS) return PF > 1 ? 7 : LH_RHS_WAVES_PER_SIMD; // a deeper ring costs registers
    // (the Float32 coupled tendency + step bound needs ~90 VGPRs: held to 72 it spills 18 values
    // per level to scratch and runs 0.34 instead of 0.2x ms on 1e6 x 64)
    if (f32_coupled_vgpr_constants<FT, MODEL, FACTORS, PERCOL, M, MODE>()) return sizeof(FT) == 4 ? (MODE == 4 ? LH_F32C_DT_WAVES : 7) : 1;
    if (sizeof(FT) == 4) return (MODE == 4 && MODEL == MODEL_COUPLED) ? LH_F32C_DT_WAVES_SGPR : LH_RHS_WAVES_PER_SIMD; // coupled/heat Float32
    return 1;
}

// Threads per workgroup of the column kernel.  The Float64 production math stages 48 KiB of
// log2/exp2 tables per workgroup (lh_fastmath.hpp), so a CU holds 3 such workgroups at most:
// kernels that fit 64 VGPRs run 1024-thread workgroups (2 per CU = 8 waves/SIMD), the others
// 512-thread ones (3 per CU = 6 waves/SIMD).  Everything else (Float32: hardware log/exp, the
// libm debug policy) keeps 256.
template <typename M, int WAVES>
constexpr int rhs_max_threads() {
    return M::uses_tables ? (WAVES >= 8 ? 1024 : 512) : 256;
}
// the waves-per-SIMD the compiler is asked to leave room for, given that workgroup size
template <typename M, int WAVES>
constexpr int rhs_min_waves() {
    if (!M::uses_tables) return WAVES;
    return WAVES >= 8 ? 8 : (WAVES >= 6 ? 6 : (WAVES >= 4 ? 4 : (WAVES >= 2 ? 2 : 1)));
}

// trips of the level loop (PF levels each) the compiler is told to unroll (0: its own choice)
#ifndef LH_RHS_OUTER_UNROLL
#define LH_RHS_OUTER_UNROLL 0
#endif
constexpr int RHS_OUTER_UNROLL = LH_RHS_OUTER_UNROLL;

// ----------------------------------------------------------------- helpers

// native clang vectors (the nontemporal builtins do not take HIP_vector_type)
template <typename FT, int N> struct Vec;
template <> struct Vec<double, 1> { using type = double; };
template <> struct Vec<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec<float, 1> { using type = float; };
template <> struct Vec<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec<double, 4> { typedef double type __attribute__((ext_vector_type(4))); };

template <typename FT, int N, bool NT = false>
__device__ __forceinline__ void vload(const FT* p, FT (&out)[N]) {
    using V = typename Vec<FT, N>::type;
    V v = NT ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p))
             : *reinterpret_cast<const V*>(p);
    const FT* e = reinterpret_cast<const FT*>(&v);
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = e[j];
}
template <typename FT, int N, bool NT = false>
__device__ __forceinline__ void vstore(FT* p, const FT (&in)[N]) {
    using V = typename Vec<FT, N>::type;
    V v;
    FT* e = reinterpret_cast<FT*>(&v);
#pragma unroll
    for (int j = 0; j < N; ++j) e[j] = in[j];
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
    else *reinterpret_cast<V*>(p) = v;
}

// Row access through a buffer descriptor: a uniform row pointer (SGPRs) plus a
// 32-bit lane byte offset -- no 64-bit vector address arithmetic, and loads or
// stores past the end of the row are dropped by the hardware range check.
// (hipcc otherwise turns `row + lane` into per-lane 64-bit induction variables.)
template <typename FT, int N, bool NT>
__device__ __forceinline__ void bload(const FT* row, unsigned row_bytes, unsigned lane_byte, FT (&out)[N]) {
    static_assert(sizeof(FT) * N == 8, "one 8-byte access per lane");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<FT*>(row), 0, row_bytes, 0x00020000);
    u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_byte, 0, NT ? 2 : 0);
    __builtin_memcpy(out, &v, 8);
}
template <typename FT, int N, bool NT>
__device__ __forceinline__ void bstore(FT* row, unsigned row_bytes, unsigned lane_byte, const FT (&in)[N]) {
    static_assert(sizeof(FT) * N == 8, "one 8-byte access per lane");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, row_bytes, 0x00020000);
    u32x2 v;
    __builtin_memcpy(&v, in, 8);
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, lane_byte, 0, NT ? 2 : 0);
}

// launch shape of the column kernel: columns per lane, levels kept in flight
// ahead of the one being computed, nontemporal global access
// SEG: level-segmented launch for small ensembles (see rhs_kernel)
template <int CPL_, int PF_, bool NT_, bool SEG_>
struct KCfg {
    static constexpr int CPL = CPL_, PF = PF_;
    static constexpr bool NT = NT_, SEG = SEG_;
};

// Stage the log2/exp2 tables of MathFast<double> in LDS (48 KiB per workgroup);
// every thread of the block must call this before any thread leaves.
template <typename M>
__device__ __forceinline__ MathTables stage_math_tables(const double* gtab, double* lds) {
    MathTables t;
    t.log_tab = lds;
    t.exp_tab = lds + 2 * LOG_TAB_N;
    if (M::uses_tables) {
        for (int i = threadIdx.x; i < MATH_TAB_DOUBLES; i += blockDim.x) lds[i] = gtab[i];
        __syncthreads();
    }
    return t;
}

// positive IEEE values order like their bit patterns: global minima are integer atomicMin
template <typename FT> struct Bits;
template <> struct Bits<double> { using type = unsigned long long; };
template <> struct Bits<float> { using type = unsigned int; };

__device__ __forceinline__ double fmax_ft(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float fmax_ft(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fmin_ft(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float fmin_ft(float a, float b) { return __builtin_fminf(a, b); }

// max of two NON-NEGATIVE floats as a signed-integer max of their bit patterns (they order alike;
// -0.0 and negative values lose, a NaN wins and is dropped with its lane at the end): one
// v_max_i32 / v_max3_i32, where fmaxf costs a canonicalising v_max_f32 per operand on top
__device__ __forceinline__ float max_nonneg(float a, float b) {
    return __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b)));
}

// Orders one wave's LDS traffic around a hand-off between its lanes: the LDS operations of a wave
// execute in order, so no instruction is needed -- the wave barrier pins the compiler's schedule and
// the two wavefront-scope fences (no code on gfx950) keep it from moving a load of another lane's
// word above the store that produced it.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename FT>
__device__ __forceinline__ bool finite(FT x) {
    return x - x == FT(0);
}

// ------------------------------------------------------------- rhs kernel
//
// rhs!(dY, Y, Ya, t): right_hand_side.jl:118-186 (RICHARDS), :192-263 (HEAT),
// :269-369 (COUPLED).  Stencils: InterpolateC2F = mean,
// GradientC2F = difference/dz, DivergenceF2C = flux difference/dz with the two
// boundary faces replaced by the boundary fluxes (SetValue).
//
// MODE 0: write the tendency dY (its theta_i plane is not touched: d theta_i = 0, see below).
// MODE 4: MODE 0 plus the local stable-step bound of the same state (the rule of
//         stable_dt_kernel, from the K, dpsi/dvl, kappa, rho_c_s this pass has in
//         registers anyway, accumulated in Float32: a Courant-type bound needs seven
//         digits, and the maximum of non-negative floats is an integer maximum of their
//         bits): one division and one integer atomicMin per wave into P.dt_out.
// MODE 1..3: fused SSPRK33 stage s (OrdinaryDiffEq SSPRK33, Shu-Osher form):
//   1: U1 = Y + dt f(Y)              (in = Y,  out = U1)
//   2: U1 = (3 Y + U1 + dt f(U1))/4  (in = U1, base = Y, out = U1)
//   3: Y  = (Y + 2 U1 + 2 dt f(U1))/3 (in = U1, base = Y, out = Y)
// MODE 5: stage 2 of a step whose first stage was NOT stored: in = k1 = f(Y) (a tendency, e.g. the
//         one lh_rhs_stable_dt left with the step bound), base = Y.  U1 = Y + dt k1 is formed in
//         registers with MODE 1's expression and used exactly like MODE 2's stored U1:
//         out = (3 Y + U1 + dt f(U1))/4.  An adaptive step is then MODE 4 + MODE 5 + MODE 3: three
//         evaluations of f instead of four, bitwise the four-launch sequence.
// In the fused stages theta_i is read from BASE (= Y) and never written: its
// tendency is identically zero (right_hand_side.jl:182, :359), so every stage
// value of theta_i equals Y's.
//
// CFG::SEG (small ensembles): a column no longer belongs to one lane.  With fewer
// columns than the chip has lanes the launch is bound by the LATENCY of one lane's
// bottom-to-top march (35 us for 64 Float64 levels, whatever the column count up to
// ~1e5), so blockIdx.y splits every column into segments of P.seg_len levels.  A lane
// computes the closures of its own cells plus the one cell below and the one above its
// segment (the face fluxes at the segment ends are evaluated by both neighbours from
// the same inputs in the same order: bitwise equal, conservation is untouched) and
// emits only its own cells.  Results are bitwise those of the unsegmented launch.
//
// Addressing: a uniform row pointer per plane (SGPRs, advanced by `stride` per
// level) plus one 32-bit lane offset, so the loop carries no vector address
// arithmetic.  The level coordinate z_i comes from LDS (staged once per block):
// as a global load it would sit in the vector-memory queue behind the next
// level's prefetch and force a full vmcnt(0) drain every level.
// NOICE: the theta_i plane the launch would read is known to be all zeros (the zero bits of
// lh_state): it is not read, ti is the literal 0 and the ice branches are compiled out -- the
// same numbers the general kernel produces for ti == 0.
// d theta_i = 0 (right_hand_side.jl:182, :359) is never stored by any mode: the host side keeps
// the theta_i plane of a tendency state zero (cleared once, tracked by the state's zero bits).
// VGF (Float64 production math): every column of the context has m = 1 - 1/n >= LH_VG_FAST_MIN_M, so the
// water closures put the exponents of their 2^(.) in place by integer addition (water_closures_log); the
// host decides (DevParams::vg_fast_all), clay-like ensembles run the VGF = false instantiation.
template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename CFG, typename M, int MODE, bool NOICE = false, bool VGF = true>
__global__ void __launch_bounds__((rhs_max_threads<M, rhs_waves_per_simd<FT, MODEL, FACTORS, PERCOL, M, CFG::PF, MODE, NOICE, CFG::CPL>()>()),
                                  (rhs_min_waves<M, rhs_waves_per_simd<FT, MODEL, FACTORS, PERCOL, M, CFG::PF, MODE, NOICE, CFG::CPL>()>()))
rhs_kernel(const DevParams<FT> P0, const Planes<FT> IN, const Planes<FT> AUX, const Planes<FT> BASE,
           const Planes<FT> OUT, const FT dt_value, const FT* __restrict__ dt_device) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    constexpr bool TEND_MODE = (MODE == 0 || MODE == 4); // (the fused stages: measured slower with it)
    // The Float64 heat kernels with conductivity factors or ice run at 3 waves/SIMD whatever they
    // do (130 VGPRs) and need more uniform constants than there are SGPRs: the compiler spills the
    // excess to VGPR lanes and reloads them with v_readlane in front of every use (20 VALU
    // instructions per cell).  The constants of the heat closures held in VGPRs instead remove that.
    DevParams<FT> P = P0;
    if constexpr (heat_vgpr_constants<FT, MODEL, FACTORS, PERCOL, M, NOICE>() && TEND_MODE && (!FACTORS || PERCOL || MODE == 4)) {
        auto vr = [](FT& x) { asm volatile("" : "+v"(x)); };
        vr(P.rho_c_ds); vr(P.rhocp_l); vr(P.rhocp_i); vr(P.T_ref); vr(P.kappa_sat_unfrozen);
        vr(P.l2_kappa_sat_unfrozen); vr(P.l2_kappa_sat_frozen); vr(P.kersten_exp_unfrozen);
        vr(P.kersten_exp_frozen); vr(P.neg_b_log2e_sc);
        if (MODE != 4) { // (the step-bound bookkeeping of MODE 4 needs the registers: 175 VGPRs = 2 waves/SIMD otherwise)
            vr(P.rho_i); vr(P.LH_f0); vr(P.inv_dz); vr(P.half_inv_dz);
            if (FACTORS) { vr(P.gamma); vr(P.T_ref_visc); vr(P.Omega); }
        }
    }
#ifndef LH_F64_FACTORS_PARTIAL_PIN
#define LH_F64_FACTORS_PARTIAL_PIN 3
#endif
    // The Float64 tendency with conductivity factors stays at 4 waves per SIMD (128 registers) with the
    // uniforms in the SGPR file and its spills (see rhs_waves_per_simd) -- but the three constants every
    // cell's T and rho_e_l K use fit beside them: 52 -> fewer v_readlane per two cells, f3c64 0.560 ->
    // 0.546 ms (same-process A/B; five or eight pinned: no further gain)
#ifndef LH_F64_FACTORS_PARTIAL_PIN_STAGES
#define LH_F64_FACTORS_PARTIAL_PIN_STAGES 0 // (the fused stages carry more state: the same pinning costs them a wave, 0.64 -> 0.81 ms)
#endif
    if constexpr (LH_F64_FACTORS_PARTIAL_PIN > 0 && heat_vgpr_constants<FT, MODEL, FACTORS, PERCOL, M, NOICE>() &&
                  (MODE == 0 || (LH_F64_FACTORS_PARTIAL_PIN_STAGES && MODE != 4)) &&
                  FACTORS && !PERCOL) {
        auto vr = [](FT& x) { asm volatile("" : "+v"(x)); };
        vr(P.rho_c_ds); vr(P.rhocp_l); vr(P.T_ref);
        if (LH_F64_FACTORS_PARTIAL_PIN >= 5) { vr(P.rhocp_i); vr(P.rho_i); }
        if (LH_F64_FACTORS_PARTIAL_PIN >= 8) { vr(P.LH_f0); vr(P.gamma); vr(P.Omega); }
    }
    constexpr int CPL = CFG::CPL, PF = CFG::PF;
    constexpr bool NT = CFG::NT;
    constexpr bool TEND = (MODE == 0 || MODE == 4); // writes a tendency (not a stage state)
    constexpr bool WANT_DT = (MODE == 4);
    // the step size either comes by value or is read from device memory, where a
    // preceding stable-dt reduction / RCCL min all-reduce left it (no host sync)
    const FT dt = (!TEND && dt_device) ? *dt_device : dt_value; // MODE 4: dt_value = courant
    __shared__ double s_tab[M::uses_tables ? MATH_TAB_DOUBLES : 2];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int n = P.nlev;
    // MODE 4: one LDS word per thread for the wave-level maximum of the face diffusivities; lanes
    // past the last column leave the neutral 0 there
    float* s_red = reinterpret_cast<float*>(s_dyn);
    if (WANT_DT) s_red[threadIdx.x] = 0.0f;
    __shared__ unsigned s_waves_done; // MODE 4: waves of this workgroup that have left their maximum in s_red
    if (WANT_DT && threadIdx.x == 0) s_waves_done = 0u;
    // level-uniform prescribed fields of Ya (DevParams::aux_prof), staged behind the reduction
    // words: T for the Richards viscosity factor, vartheta_l and theta_i for the heat-only model
    constexpr bool MAY_PROF = (MODEL == MODEL_HEAT) || (MODEL == MODEL_RICHARDS && FACTORS);
    FT* s_pf = reinterpret_cast<FT*>(s_dyn + (WANT_DT ? ((size_t(blockDim.x) * sizeof(float) + 15) & ~size_t(15)) : 0));
    const FT* pf_T = (MAY_PROF && MODEL == MODEL_RICHARDS) ? P.aux_prof[3] : nullptr;
    const FT* pf_vl = (MAY_PROF && MODEL == MODEL_HEAT) ? P.aux_prof[0] : nullptr;
    const FT* pf_ti = (MAY_PROF && MODEL == MODEL_HEAT) ? P.aux_prof[1] : nullptr;
    if constexpr (MAY_PROF) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (pf_T) s_pf[i] = pf_T[i];
            if (pf_vl) s_pf[i] = pf_vl[i];
            if (pf_ti) s_pf[n + i] = pf_ti[i];
        }
    }
    const M mm(stage_math_tables<M>(P.math_tab, s_tab));
    if (!M::uses_tables) __syncthreads();

    // Workgroups are dealt round-robin to the 8 XCDs.  With xcd_remap, workgroup b handles
    // column block (b % 8) * (nblocks / 8) + b / 8: each XCD streams one contiguous eighth of
    // every plane row instead of every eighth 2-KiB piece.
    unsigned blk = blockIdx.x;
    if (P.xcd_remap) {
        const unsigned per = gridDim.x >> 3;
        if (blk < (per << 3)) blk = (blk & 7u) * per + (blk >> 3);
    }
    const int64_t col0 = (int64_t(blk) * blockDim.x + threadIdx.x) * CPL;
    if (col0 >= P.ncols) return;
    const int64_t stride = P.stride;
    // cells this lane emits: [i_lo, i_hi); cells whose closures it evaluates: [i_first, i_end)
    constexpr bool SEG = CFG::SEG;
    const int i_lo = SEG ? int(blockIdx.y) * P.seg_len : 0;
    const int i_hi = SEG ? (i_lo + P.seg_len < n ? i_lo + P.seg_len : n) : n;
    const int i_first = (SEG && i_lo > 0) ? i_lo - 1 : 0;
    const int i_end = (SEG && i_hi < n) ? i_hi + 1 : n;
    // 32-bit BYTE offset of this lane inside a plane row (lh_create bounds a row to < 4 GiB)
    const unsigned lane_byte = (unsigned)col0 * (unsigned)sizeof(FT);
    const unsigned row_bytes = (unsigned)(stride * (int64_t)sizeof(FT));
    constexpr bool BUF = (sizeof(FT) * CPL == 8); // production shapes: one 8-byte access per lane
    auto rload = [&](const FT* row, FT (&out)[CPL]) {
        if constexpr (BUF) bload<FT, CPL, NT>(row, row_bytes, lane_byte, out);
        else vload<FT, CPL, NT>(row + col0, out);
    };
    auto rstore = [&](FT* row, const FT (&in)[CPL]) {
        if constexpr (BUF) bstore<FT, CPL, NT>(row, row_bytes, lane_byte, in);
        else vstore<FT, CPL, NT>(row + col0, in);
    };

    // uniform row pointers (level 0); HEAT reads the prescribed water fields from
    // Ya (right_hand_side.jl:200-201); fused stages read theta_i from BASE
    const int64_t in0 = SEG ? stride * i_first : 0, out0 = SEG ? stride * i_lo : 0;
    constexpr bool FROM_K1 = (MODE == 5); // the stage state is Y + dt k1, formed on the fly
    const FT* r_vl = (MODEL == MODEL_HEAT ? AUX.v[0] : (FROM_K1 ? BASE.v[0] : IN.v[0])) + in0;
    const FT* r_ti = NOICE ? nullptr : (MODEL == MODEL_HEAT ? AUX.v[1] : (TEND ? IN.v[1] : BASE.v[1])) + in0;
    const FT* r_re = HEAT ? (FROM_K1 ? BASE.v[2] : IN.v[2]) + in0 : nullptr;
    const FT* r_k1v = (FROM_K1 && WATER) ? IN.v[0] + in0 : nullptr; // k1 = f(Y)
    const FT* r_k1e = (FROM_K1 && HEAT) ? IN.v[2] + in0 : nullptr;
    const bool need_Taux = (MODEL == MODEL_RICHARDS) && FACTORS && P.viscosity_kind;
    const FT* r_Ta = need_Taux ? AUX.v[3] + in0 : nullptr;
    const FT* b_vl = ((MODE == 2 || MODE == 3) && WATER) ? BASE.v[0] + out0 : nullptr;
    const FT* b_re = ((MODE == 2 || MODE == 3) && HEAT) ? BASE.v[2] + out0 : nullptr;
    FT* o_vl = WATER ? OUT.v[0] + out0 : nullptr;
    FT* o_re = HEAT ? OUT.v[2] + out0 : nullptr;

    ColC<FT> c[CPL];
    int64_t colj[CPL]; // column index clamped into [0, ncols): pad lanes reuse the last column
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        colj[j] = col0 + j < P.ncols ? col0 + j : P.ncols - 1;
        c[j] = make_colc<FT, M>(P, colj[j], PERCOL);
        if (WATER && !NOICE) finish_colc<FT, M>(mm, c[j]);
        // (psi's exponent is an fma of two column constants: one of them has to sit in a VGPR anyway --
        // pinned here, it is not re-materialised from its SGPR pair in front of every use)
        if constexpr (M::uses_tables && WATER && !PERCOL && !HEAT) asm volatile("" : "+v"(c[j].e_log2_alpha));
    }
    // Float32 coupled tendency: with every uniform constant in SGPRs the kernel spills 45 of them
    // to VGPR lanes (28 v_readlane per two cells).  Thirteen column constants held in VGPRs
    // instead (and 72 rather than 64 VGPRs: 7 waves/SIMD) remove every spill: +1..2.5 %.
    if constexpr (f32_coupled_vgpr_constants<FT, MODEL, FACTORS, PERCOL, M, MODE>()) {
        auto vr = [](FT& x) { asm volatile("" : "+v"(x)); };
        ColC<FT>& q = c[0];
        vr(q.nu); vr(q.theta_lim); vr(q.theta_r); vr(q.inv_por); vr(q.e_inv_m); vr(q.e_m); vr(q.e_one);
        vr(q.e_inv_n); vr(q.e_log2_alpha); vr(q.Ksat); vr(q.inv_S_s); vr(q.inv_nu); vr(q.k_dry);
        if (!NOICE) vr(q.l2_por);
#pragma unroll
        for (int j = 1; j < CPL; ++j) c[j] = c[0];
    }

    constexpr bool vgf = VGF && M::uses_tables; // integer-exponent 2^(.) (a host decision)
    // Fluxes are carried in units of the TENDENCY: the arithmetic-mean factor 1/2 of InterpolateC2F,
    // the 1/dz of GradientC2F and the 1/dz of DivergenceF2C are one constant cg = (1/2)/dz^2 applied to
    // the centre difference, and (production math) the closures return K WITHOUT Ksat, which joins cg in
    // the per-column constant of the water flux: two multiplications per cell less than
    // F = -(K_lo + K_hi) (dh (1/2)/dz), -(F_hi - F_lo)/dz, a rounding-level regrouping.  Boundary
    // fluxes (physical units, boundary_fluxes) are scaled by 1/dz once per column.
    constexpr bool RELK = M::is_production;
    const FT cgT = P.cg2;
    FT cgw[CPL], Ksc[CPL]; // water flux constant; the factor that makes a closure K a true conductivity
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        Ksc[j] = RELK ? c[j].Ksat : FT(1);
        cgw[j] = RELK ? c[j].cgw : cgT;
    }

    FT vl[CPL], ti[CPL], re[CPL], Ta[CPL];         // current cell inputs
    FT vl_n[PF][CPL], ti_n[PF][CPL], re_n[PF][CPL], Ta_n[PF][CPL]; // PF levels in flight
    FT kv_n[PF][CPL], ke_n[PF][CPL];               // MODE 5: k1 of those levels
    FT vl_p[CPL], re_p[CPL];                       // previous cell inputs (fused stages)
    FT yv[CPL], ye[CPL], yv_p[CPL], ye_p[CPL];     // MODE 5: Y of the current / previous cell (the stage's base)
    FT K_p[CPL], psi_p[CPL], T_p[CPL], kap_p[CPL], E_p[CPL];
    FT Fw_lo[CPL], Fe_lo[CPL];
    FT nf_acc = FT(0); // += 0 * tendency: becomes NaN once any tendency is non-finite
    // MODE 4: stable-step bookkeeping, in Float32 whatever FT is (see slope32): n m d psi / d vl
    // and 1 / rho_c_s of the previous cell, and the lane's maxima of TWICE the water (x n m) and
    // heat face diffusivities (the column constant n m and the exact factor 1/2 of the mean
    // coefficients are divided out once per column / wave at the end)
    float dpsi_p[CPL], ircs_p[CPL], DmaxW[CPL], DmaxT[CPL];
    float DmaxWb[CPL]; // the Dirichlet-face terms, formed with TRUE conductivities (DmaxW: relative to Ksc)

#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        K_p[j] = psi_p[j] = T_p[j] = kap_p[j] = E_p[j] = Fw_lo[j] = Fe_lo[j] = FT(0);
        vl_p[j] = re_p[j] = FT(0);
        yv[j] = ye[j] = yv_p[j] = ye_p[j] = FT(0);
        vl[j] = ti[j] = re[j] = FT(0);
        Ta[j] = FT(288); // PrescribedTemperatureModel default (models.jl:53)
        dpsi_p[j] = DmaxW[j] = DmaxWb[j] = DmaxT[j] = ircs_p[j] = 0.0f;
    }
    // loads the level the row pointers currently address into ring slot `slot`,
    // then moves the pointers one level up
    int lev_f = i_first; // the level fetch() addresses
    auto fetch = [&](int slot) {
        if (MAY_PROF && pf_vl) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) vl_n[slot][j] = s_pf[lev_f];
        } else rload(r_vl, vl_n[slot]);
        if (!NOICE) {
            if (MAY_PROF && pf_ti) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) ti_n[slot][j] = s_pf[n + lev_f];
            } else rload(r_ti, ti_n[slot]);
        }
        if (HEAT) rload(r_re, re_n[slot]);
        if (need_Taux) {
            if (MAY_PROF && pf_T) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) Ta_n[slot][j] = s_pf[lev_f];
            } else rload(r_Ta, Ta_n[slot]);
        }
        ++lev_f;
        if (FROM_K1 && WATER) { rload(r_k1v, kv_n[slot]); r_k1v += stride; }
        if (FROM_K1 && HEAT) { rload(r_k1e, ke_n[slot]); r_k1e += stride; }
        r_vl += stride;
        if (!NOICE) r_ti += stride;
        if (HEAT) r_re += stride;
        if (need_Taux) r_Ta += stride;
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            vl_n[k][j] = ti_n[k][j] = re_n[k][j] = kv_n[k][j] = ke_n[k][j] = FT(0);
            Ta_n[k][j] = FT(288);
        }
        if (i_first + k < i_end) fetch(k);
    }

    // emit the result of the cell the OUT/BASE row pointers address
    auto emit = [&](const FT (&Fw_hi)[CPL], const FT (&Fe_hi)[CPL], const FT (&u_vl)[CPL],
                    const FT (&u_re)[CPL], const FT (&y_vl)[CPL], const FT (&y_re)[CPL]) {
        FT dvl[CPL], dre[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            dvl[j] = WATER ? Fw_lo[j] - Fw_hi[j] : FT(0); // -(F_hi - F_lo), fluxes in tendency units
            dre[j] = HEAT ? Fe_lo[j] - Fe_hi[j] : FT(0);
            if (CPL == 1 || col0 + j < P.ncols) {
                if (WATER) nf_acc = fma_ft(dvl[j], FT(0), nf_acc);
                if (HEAT) nf_acc = fma_ft(dre[j], FT(0), nf_acc);
            }
        }
        if (TEND) {
            if (WATER) rstore(o_vl, dvl); // (d theta_i = 0: the plane is kept zero by the host side)
            if (HEAT) rstore(o_re, dre);
        } else {
            auto stage = [&](const FT* brow, FT* orow, const FT (&u)[CPL], const FT (&k)[CPL], const FT (&yb)[CPL]) {
                FT b[CPL], r[CPL];
                if (MODE == 2 || MODE == 3) rload(brow, b);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    if (FROM_K1) b[j] = yb[j]; // (the base was read with the state: no second load)
                    if (MODE == 1)
                        r[j] = u[j] + dt * k[j];
                    else if (MODE == 2 || MODE == 5)
                        r[j] = (FT(3) * b[j] + u[j] + dt * k[j]) * FT(0.25);
                    else {
                        // s / 3 as s*(1/3) plus one residual correction: a bare multiply by
                        // the rounded 1/3 biases every step by 5.5e-17 and the total mass
                        // drifts (1.6e-11 after 138 240 steps); this form is unbiased
                        const FT sum = b[j] + FT(2) * u[j] + FT(2) * dt * k[j];
                        const FT q = sum * FT(1.0 / 3.0);
                        r[j] = fma_ft(fma_ft(FT(-3), q, sum), FT(1.0 / 3.0), q);
                    }
                }
                rstore(orow, r);
            };
            if (WATER) stage(b_vl, o_vl, u_vl, dvl, y_vl);
            if (HEAT) stage(b_re, o_re, u_re, dre, y_re);
        }
        if (WATER) {
            o_vl += stride;
            if (MODE == 2 || MODE == 3) b_vl += stride;
        }
        if (HEAT) {
            o_re += stride;
            if (MODE == 2 || MODE == 3) b_re += stride;
        }
    };

#if LH_RHS_OUTER_UNROLL > 0
#pragma unroll RHS_OUTER_UNROLL
#endif
    for (int i0 = i_first; i0 < i_end; i0 += PF) {
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int i = i0 + k;
        if (PF > 1 && i >= i_end) break;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            vl[j] = vl_n[k][j];
            ti[j] = NOICE ? FT(0) : ti_n[k][j];
            re[j] = re_n[k][j];
            Ta[j] = Ta_n[k][j];
            if (FROM_K1) { // U1 = Y + dt k1, MODE 1's expression
                yv[j] = vl[j];
                ye[j] = re[j];
                if (WATER) vl[j] = yv[j] + dt * kv_n[k][j];
                if (HEAT) re[j] = ye[j] + dt * ke_n[k][j];
            }
        }
        // keep PF levels in flight ahead of the one computed.  (Refilling the slot only AFTER the cell's
        // closures, so that the load lands in the registers they free and the ring needs no moves,
        // was measured: -1 instruction per cell but the prefetch distance shrinks by a cell -- C2 +2 %,
        // the HBM-bound Float32 C3 +5 % slower; with 4 levels in flight -1 % at twice the code.)
        if (i + PF < i_end) fetch(k);
        FT K[CPL], psi[CPL], T[CPL], kap[CPL], E[CPL], rcs[CPL];
        float dpsi[CPL], ircs[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            T[j] = Ta[j];
            kap[j] = FT(0);
            K[j] = psi[j] = E[j] = FT(0);
            dpsi[j] = ircs[j] = 0.0f;
            rcs[j] = FT(1);
            if (HEAT) {
                T[j] = temperature_closure<FT, M, NOICE>(mm, P, c[j], vl[j], ti[j], re[j], rcs[j]);
                kap[j] = kappa_closure<FT, M, NOICE>(mm, P, c[j], vl[j], ti[j]);
                if (WANT_DT) ircs[j] = float(mm.rcp(rcs[j])); // (the reciprocal temperature_closure formed)
            }
            if (WATER) {
                // (psi[], psi_p[] hold -psi: see head_difference)
                water_closures<FT, M, FACTORS, true, WANT_DT, NOICE, RELK, HEAT, true>(mm, P, c[j], vl[j], ti[j], T[j], K[j],
                                                                                       psi[j], &dpsi[j], vgf);
                if (HEAT) E[j] = (P.rhocp_l * (T[j] - P.T_ref)) * K[j]; // rho_e_int_l * K (:364)
            }
            if (WANT_DT && i > i_first) { // the rule of stable_dt_kernel per interior face (x 2)
                if (WATER) DmaxW[j] = max_nonneg(DmaxW[j], float(K_p[j] + K[j]) * max_nonneg(dpsi_p[j], dpsi[j]));
                if (HEAT) DmaxT[j] = max_nonneg(DmaxT[j], float(kap_p[j] + kap[j]) * max_nonneg(ircs_p[j], ircs[j]));
            }
        }
        if (i == 0) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                FT K_f = FT(0), kap_f = FT(0);
                const FT K_c = K[j] * Ksc[j]; // the true conductivity of the boundary cell
                boundary_fluxes<FT, M, MODEL, FACTORS, NOICE>(mm, P, c[j], FACE_BOTTOM, colj[j], vl[j], ti[j],
                                                              T[j], K_c, -psi[j], Fe_lo[j], Fw_lo[j], &K_f, &kap_f, vgf);
                Fe_lo[j] = Fe_lo[j] * P.inv_dz;
                Fw_lo[j] = Fw_lo[j] * P.inv_dz;
                if (WANT_DT) { // the bottom cell's own coefficients; Dirichlet faces: half a cell away, face-state coefficients
                    DmaxW[j] = max_nonneg(DmaxW[j], 2.0f * float(K[j]) * dpsi[j]);
                    DmaxWb[j] = max_nonneg(DmaxWb[j], 4.0f * float(fmax_ft(K_f, K_f > FT(0) ? K_c : FT(0))) * dpsi[j]);
                    if (HEAT) {
                        DmaxT[j] = max_nonneg(DmaxT[j], 2.0f * float(kap[j]) * ircs[j]);
                        DmaxT[j] = max_nonneg(DmaxT[j], 4.0f * float(fmax_ft(kap_f, kap_f > FT(0) ? kap[j] : FT(0))) * ircs[j]);
                    }
                }
            }
        } else if (!SEG || i > i_first) { // (the cell below a segment only primes the *_p values)
            FT Fw[CPL], Fe[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                Fw[j] = Fe[j] = FT(0);
                FT gh = FT(0);
                // -1/2 (a_lo + a_hi) (x_hi - x_lo)/dz /dz with the three constants (and Ksat) folded
                // into the gradient's factor (see cgw above)
                if (WATER) {
                    gh = head_difference(psi[j], psi_p[j], P.dz) * cgw[j];
                    Fw[j] = -(K_p[j] + K[j]) * gh;
                }
                if (HEAT) {
                    FT gT = (T[j] - T_p[j]) * cgT;
                    Fe[j] = -(kap_p[j] + kap[j]) * gT;
                    if (WATER) Fe[j] = Fe[j] - (E_p[j] + E[j]) * gh;
                }
            }
            if (!SEG || i > i_lo) emit(Fw, Fe, vl_p, re_p, yv_p, ye_p); // cell i-1 belongs to this segment
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                Fw_lo[j] = Fw[j];
                Fe_lo[j] = Fe[j];
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            vl_p[j] = vl[j];
            re_p[j] = re[j];
            if (FROM_K1) {
                yv_p[j] = yv[j];
                ye_p[j] = ye[j];
            }
            K_p[j] = K[j];
            psi_p[j] = psi[j];
            T_p[j] = T[j];
            kap_p[j] = kap[j];
            E_p[j] = E[j];
            if (WANT_DT) {
                dpsi_p[j] = dpsi[j];
                ircs_p[j] = ircs[j];
            }
        }
      }
    }
    if (!SEG || i_hi == n) { // the top face of the column
        FT Fw[CPL], Fe[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            FT K_f = FT(0), kap_f = FT(0);
            const FT K_c = K_p[j] * Ksc[j]; // the true conductivity of the boundary cell
            boundary_fluxes<FT, M, MODEL, FACTORS, NOICE>(mm, P, c[j], FACE_TOP, colj[j], vl[j], ti[j], T_p[j],
                                                          K_c, -psi_p[j], Fe[j], Fw[j], &K_f, &kap_f, vgf);
            Fe[j] = Fe[j] * P.inv_dz;
            Fw[j] = Fw[j] * P.inv_dz;
            if (WANT_DT) { // the top cell's own coefficients, and the Dirichlet face's
                DmaxW[j] = max_nonneg(DmaxW[j], 2.0f * float(K_p[j]) * dpsi_p[j]);
                DmaxWb[j] = max_nonneg(DmaxWb[j], 4.0f * float(fmax_ft(K_f, K_f > FT(0) ? K_c : FT(0))) * dpsi_p[j]);
                if (HEAT) {
                    DmaxT[j] = max_nonneg(DmaxT[j], 2.0f * float(kap_p[j]) * ircs_p[j]);
                    DmaxT[j] = max_nonneg(DmaxT[j], 4.0f * float(fmax_ft(kap_f, kap_f > FT(0) ? kap_p[j] : FT(0))) * ircs_p[j]);
                }
            }
        }
        emit(Fw, Fe, vl_p, re_p, yv_p, ye_p);
    }
    if (nf_acc != nf_acc) atomicOr(P.status, 1u);
    if (WANT_DT) { // dt = courant dz^2 / (max D over the wave's columns), one atomicMin per wave
        using U = typename Bits<FT>::type;
        float dmax = 0.0f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) { // (a column whose maximum is NaN is dropped: it is flagged through P.status)
            const float Dj = max_nonneg(max_nonneg(DmaxW[j] * float(Ksc[j]), DmaxWb[j]) * float(FT(1) / (c[j].n * c[j].m)), DmaxT[j]);
            if ((CPL == 1 || col0 + j < P.ncols) && Dj == Dj) dmax = max_nonneg(dmax, Dj);
        }
        // Wave maximum by a binary tree through LDS (x -> fl(c/x) is monotone, so the minimum of
        // the lanes' quotients IS the quotient of the maximum: one division per wave; a maximum
        // is exact, so the result does not depend on how columns are dealt to waves or ranks).
        // Lanes past the last column returned early -- a cross-lane shuffle could read their dead
        // registers; their LDS words hold the neutral 0 from the prologue.  The lanes still here
        // are a prefix of the wave (col0 grows with the lane), so lane l < off always has its
        // partner's word to read.  One wave = one 64-word segment; the LDS operations of a wave
        // execute in order, so the wave barriers only pin the compiler's ordering.
        float* seg = s_red + (threadIdx.x & ~63u);
        const unsigned l = threadIdx.x & 63u;
        seg[l] = dmax;
#pragma unroll
        for (unsigned off = 32; off > 0; off >>= 1) {
            wave_sync();
            if (l < off) {
                dmax = max_nonneg(dmax, seg[l + off]);
                seg[l] = dmax;
            }
        }
        // One candidate per WORKGROUP: the last of its waves to get here (an LDS counter; no barrier, nobody waits)
        // folds the others' maxima in -- seg[0] of every wave, written before that wave's count, in order.
        if (l == 0) {
            const int64_t first_lane = int64_t(blk) * blockDim.x;             // lanes (of CPL columns) before this workgroup
            const int64_t lanes_left = (P.ncols + CPL - 1) / CPL - first_lane;  // > 0: the grid covers the columns
            const unsigned nwaves = blockDim.x >> 6;
            const unsigned nw = lanes_left >= int64_t(blockDim.x) ? nwaves : unsigned((lanes_left + 63) >> 6); // waves not gone at the top
            if (atomicAdd(&s_waves_done, 1u) == nw - 1u) {
                for (unsigned w = 0; w < nwaves; ++w) dmax = max_nonneg(dmax, s_red[w << 6]);
                if (dmax > 0.0f) {
                    const FT best = (FT(2) * dt * P.dz * P.dz) / FT(dmax); // dmax = twice the diffusivity
                    U b;
                    __builtin_memcpy(&b, &best, sizeof(FT));
                    // (most bounds are above the minimum already there: a plain read first -- the atomic only
                    // when it would change the word; a stale read can only cause a redundant atomic)
                    U* word = reinterpret_cast<U*>(P.dt_out);
                    if (b < __atomic_load_n(word, __ATOMIC_RELAXED)) atomicMin(word, b);
                }
            }
        }
    }
}

// --------------------------------------------- persistent column stepper
//
// Ensembles of a few columns (the reference's own shape is ONE column): even
// level-segmented, a fused SSPRK33 step is three dependent launches of ~6 us each.
// Here one workgroup owns one column and one thread one CELL, the state lives in
// registers, and the whole loop over steps and stages runs inside one launch: per
// stage every thread evaluates the closures of its cell, publishes K, -psi (T, kappa,
// rho_e_l K) in LDS, and forms the flux of the face below and the face above it from
// its neighbours' values -- each interior face is evaluated by both adjacent threads
// with the expression of rhs_kernel (lower cell first), so the step is bitwise the
// fused-stage one and conservation is untouched.  Two barriers per stage.
// bcv: NULL or [nsteps][3][2][2] FT boundary values (step, stage, face, component).
// This thread-per-cell form serves columns of MORE than 128 levels; up to 128 levels a column is one
// wavefront with one or two adjacent cells per lane and no workgroup barrier at all
// (column_stepper_wave_kernel below).
// exchange arrays per column / plane tiles of the initial fetch (the two share the dynamic LDS)
constexpr int CS_FACE_WORDS = 8; // column_stepper_wave_kernel: 2 faces x (K, psi, kappa) of a Dirichlet face state, padded
template <int MODEL> constexpr int cs_exchange_arrays() {
    return MODEL == MODEL_COUPLED ? 5 : 2;
}
// one-wave columns also publish the flux of the face BELOW each cell (water and/or heat), so that the
// face above is read, not evaluated a second time
template <int MODEL> constexpr int cs_flux_arrays() {
    return MODEL == MODEL_COUPLED ? 2 : 1;
}
static inline int cs_fetch_tiles(int model, bool noice, bool need_Taux) {
    if (model == MODEL_RICHARDS) return 1 + (noice ? 0 : 1) + (need_Taux ? 2 : 0); // T sits in tile 3
    return 3; // vl, ti, rhoe (HEAT reads the first two from Ya)
}

template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename M, bool NOICE = false, bool VGF = true>
__global__ void __launch_bounds__(1024)
column_stepper_kernel(const DevParams<FT> P0, const Planes<FT> Y, const Planes<FT> AUX, const FT dt_value,
                      const FT* __restrict__ dt_device, const int64_t nsteps, const FT* __restrict__ bcv) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    __shared__ double s_tab[M::uses_tables ? MATH_TAB_DOUBLES : 2];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int n = P0.nlev;
    // a workgroup holds blockDim.x / tpc columns of tpc = roundup(n, 64) threads each (they
    // share the staged math tables; the barriers couple them, which costs nothing)
    const int tpc = (n + 63) & ~63;
    const int slot = int(threadIdx.x) / tpc;
    const int cpb = int(blockDim.x) / tpc;
    const FT dt = dt_device ? *dt_device : dt_value;
    // exchange arrays of this column: (K, -psi) for the water, (T, kappa) for the heat, rho_e_l K for
    // both -- only what the model needs (LDS per workgroup sets how many workgroups a CU holds)
    constexpr int NARR = cs_exchange_arrays<MODEL>();
    FT* sK = reinterpret_cast<FT*>(s_dyn) + size_t(slot) * NARR * n;
    FT* sh = sK + n;
    FT* sT = WATER ? sh + n : sK;
    FT* sKap = sT + n;
    FT* sE = sKap + n;
    const M mm(stage_math_tables<M>(P0.math_tab, s_tab));
    if (!M::uses_tables) __syncthreads();
    DevParams<FT> P = P0; // boundary values change per stage
    const int i = int(threadIdx.x) - slot * tpc;
    // XCD-contiguous workgroup map (as rhs_kernel): neighbouring workgroups share the 128-byte
    // lines their 16..64-byte pieces of a plane row lie in, so they must share an L2
    unsigned blk = blockIdx.x;
    if (P.xcd_remap) {
        const unsigned per = gridDim.x >> 3;
        if (blk < (per << 3)) blk = (blk & 7u) * per + (blk >> 3);
    }
    const int64_t col_raw = int64_t(blk) * cpb + slot;
    const bool cell = i < n && col_raw < P.ncols;
    const int ic = i < n ? i : n - 1;
    const int64_t col = col_raw < P.ncols ? col_raw : P.ncols - 1; // spare slots shadow the last column
    ColC<FT> c = make_colc<FT, M>(P, col, PERCOL);
    if (WATER && !NOICE) finish_colc<FT, M>(mm, c);
    constexpr bool vgf = VGF && M::uses_tables; // (as rhs_kernel)
    // fluxes in tendency units, K without Ksat: rhs_kernel's constants and expressions, to the letter
    constexpr bool RELK = M::is_production;
    const FT cgT = P.cg2;
    const FT Ksc = RELK ? c.Ksat : FT(1);
    const FT cgw = RELK ? c.cgw : cgT;
    const bool need_Taux = (MODEL == MODEL_RICHARDS) && FACTORS && P.viscosity_kind;
    // Planes are column-fastest, threads here are level-fastest: go through LDS tiles so that
    // global memory sees the cpb adjacent columns of a level as one contiguous piece.  All
    // planes are requested before the one barrier (a block is short-lived when nsteps is
    // small: serialised round trips to HBM would dominate it).
    FT* tiles = reinterpret_cast<FT*>(s_dyn); // tiles [cpb][n] in the exchange arrays' space (tile k <= 3)
    const int64_t col_first = int64_t(blk) * cpb;
    const int tile_n = n * cpb;
    auto request = [&](const FT* plane, int k) {
        for (int e = threadIdx.x; e < tile_n; e += blockDim.x) {
            const int lev = e / cpb, cs = e - lev * cpb;
            if (col_first + cs < P.ncols) tiles[k * tile_n + cs * n + lev] = plane[int64_t(lev) * P.stride + col_first + cs];
        }
    };
    auto fetched = [&](int k) -> FT { return (col_raw < P.ncols) ? tiles[k * tile_n + slot * n + ic] : FT(0); };
    // HEAT reads the prescribed water fields from Ya (right_hand_side.jl:200-201); level-uniform
    // prescribed fields (DevParams::aux_prof) come from their [nlev] arrays, not from planes
    const FT* pf_vl = MODEL == MODEL_HEAT ? P.aux_prof[0] : nullptr;
    const FT* pf_ti = MODEL == MODEL_HEAT ? P.aux_prof[1] : nullptr;
    const FT* pf_T = need_Taux ? P.aux_prof[3] : nullptr;
    if (!pf_vl) request(MODEL == MODEL_HEAT ? AUX.v[0] : Y.v[0], 0);
    if (!NOICE && !pf_ti) request(MODEL == MODEL_HEAT ? AUX.v[1] : Y.v[1], 1);
    if (HEAT) request(Y.v[2], 2);
    if (need_Taux && !pf_T) request(AUX.v[3], 3);
    __syncthreads();
    FT y_vl = pf_vl ? pf_vl[ic] : fetched(0);
    const FT ti = NOICE ? FT(0) : (pf_ti ? pf_ti[ic] : fetched(1)); // NOICE: the theta_i plane is known to be all zeros
    FT y_re = HEAT ? fetched(2) : FT(0);
    const FT Ta = need_Taux ? (pf_T ? pf_T[ic] : fetched(3)) : FT(288);
    __syncthreads();
    FT nf_acc = FT(0);
    for (int64_t s = 0; s < nsteps; ++s) {
        FT u_vl = y_vl, u_re = y_re; // the stage state
#pragma unroll 1
        for (int stage = 0; stage < 3; ++stage) {
            if (bcv) {
                const FT* b = bcv + (s * 3 + stage) * 4;
                P.bc_value[0][0] = b[0];
                P.bc_value[0][1] = b[1];
                P.bc_value[1][0] = b[2];
                P.bc_value[1][1] = b[3];
            }
            FT T = Ta, kap = FT(0), K = FT(0), psi = FT(0), E = FT(0), rcs = FT(1);
            if (HEAT) {
                T = temperature_closure<FT, M, NOICE>(mm, P, c, u_vl, ti, u_re, rcs);
                kap = kappa_closure<FT, M, NOICE>(mm, P, c, u_vl, ti);
            }
            if (WATER) {
                water_closures<FT, M, FACTORS, true, false, NOICE, RELK, HEAT, true>(mm, P, c, u_vl, ti, T, K, psi, nullptr, vgf); // psi: -psi
                if (HEAT) E = (P.rhocp_l * (T - P.T_ref)) * K; // rho_e_int_l * K (:364)
            }
            if (i < n) {
                if (WATER) { sK[i] = K; sh[i] = psi; }
                if (HEAT) { sT[i] = T; sKap[i] = kap; }
                if (HEAT && WATER) sE[i] = E;
            }
            __syncthreads();
            FT Fw_lo = FT(0), Fe_lo = FT(0), Fw_hi = FT(0), Fe_hi = FT(0);
            // boundary faces: the bottom thread and the top thread go through boundary_fluxes
            // TOGETHER (one divergent pass, not two, when both faces need closures)
            FT Fw_b = FT(0), Fe_b = FT(0);
            const bool at_bottom = (i == 0), at_top = (i == n - 1);
            if (i < n && (at_bottom || at_top)) {
                boundary_fluxes<FT, M, MODEL, FACTORS, NOICE>(mm, P, c, at_bottom ? FACE_BOTTOM : FACE_TOP, col, u_vl, ti,
                                                              T, K * Ksc, -psi, Fe_b, Fw_b, nullptr, nullptr, vgf);
                Fe_b = Fe_b * P.inv_dz;
                Fw_b = Fw_b * P.inv_dz;
            }
            if (i < n) {
                if (at_bottom) {
                    Fw_lo = Fw_b;
                    Fe_lo = Fe_b;
                } else {
                    FT gh = FT(0);
                    if (WATER) { // (as rhs_kernel)
                        gh = head_difference(psi, sh[i - 1], P.dz) * cgw;
                        Fw_lo = -(sK[i - 1] + K) * gh;
                    }
                    if (HEAT) {
                        const FT gT = (T - sT[i - 1]) * cgT;
                        Fe_lo = -(sKap[i - 1] + kap) * gT;
                        if (WATER) Fe_lo = Fe_lo - (sE[i - 1] + E) * gh;
                    }
                }
                if (at_top) {
                    if (at_bottom) { // a one-cell column: the same thread owns both faces
                        boundary_fluxes<FT, M, MODEL, FACTORS, NOICE>(mm, P, c, FACE_TOP, col, u_vl, ti, T, K * Ksc, -psi, Fe_hi, Fw_hi, nullptr, nullptr, vgf);
                        Fe_hi = Fe_hi * P.inv_dz;
                        Fw_hi = Fw_hi * P.inv_dz;
                    } else {
                        Fw_hi = Fw_b;
                        Fe_hi = Fe_b;
                    }
                } else {
                    FT gh = FT(0);
                    if (WATER) {
                        gh = head_difference(sh[i + 1], psi, P.dz) * cgw;
                        Fw_hi = -(K + sK[i + 1]) * gh;
                    }
                    if (HEAT) {
                        const FT gT = (sT[i + 1] - T) * cgT;
                        Fe_hi = -(kap + sKap[i + 1]) * gT;
                        if (WATER) Fe_hi = Fe_hi - (E + sE[i + 1]) * gh;
                    }
                }
            }
            const FT dvl = WATER ? Fw_lo - Fw_hi : FT(0);
            const FT dre = HEAT ? Fe_lo - Fe_hi : FT(0);
            if (cell) {
                if (WATER) nf_acc = fma_ft(dvl, FT(0), nf_acc);
                if (HEAT) nf_acc = fma_ft(dre, FT(0), nf_acc);
            }
            // the stage updates of rhs_kernel MODE 1..3 (b = Y, u = stage state)
            auto upd = [&](FT b, FT u, FT k) -> FT {
                if (stage == 0) return u + dt * k;
                if (stage == 1) return (FT(3) * b + u + dt * k) * FT(0.25);
                const FT sum = b + FT(2) * u + FT(2) * dt * k;
                const FT q = sum * FT(1.0 / 3.0);
                return fma_ft(fma_ft(FT(-3), q, sum), FT(1.0 / 3.0), q);
            };
            if (WATER) u_vl = upd(y_vl, u_vl, dvl);
            if (HEAT) u_re = upd(y_re, u_re, dre);
            __syncthreads(); // neighbours have read this stage's LDS values
        }
        y_vl = u_vl;
        y_re = u_re;
    }
    __syncthreads(); // the tiles overlay other columns' exchange arrays
    if (i < n) {
        if (WATER) tiles[slot * n + i] = y_vl;
        if (HEAT) tiles[tile_n + slot * n + i] = y_re;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < tile_n; e += blockDim.x) {
        const int lev = e / cpb, cs = e - lev * cpb;
        if (col_first + cs < P.ncols) {
            const int64_t g = int64_t(lev) * P.stride + col_first + cs;
            if (WATER) Y.v[0][g] = tiles[cs * n + lev];
            if (HEAT) Y.v[2][g] = tiles[tile_n + cs * n + lev];
        }
    }
    if (cell && nf_acc != nf_acc) atomicOr(P.status, 1u);
}

// ----------------------------------- persistent column stepper, one wave per column
//
// column_stepper_kernel's one-wave form for columns of up to 64 CW levels: lane l of the column's
// wave owns the CW ADJACENT cells CW l .. CW l + CW - 1, so the faces between them are evaluated in
// the lane, and only the lane's top cell travels through LDS (to lane l + 1) and the flux of the face
// below its bottom cell back (to lane l - 1): every interior face is evaluated ONCE, with
// rhs_kernel's expression (lower cell first), no workgroup barrier anywhere in the time loop (LDS
// operations of one wave execute in order) -- bitwise the fused stages.  CW = 1: columns of <= 64
// levels; CW = 2: the 128-level configurations, which the thread-per-cell form runs with two waves
// per column, two workgroup barriers per stage and every interior face evaluated twice.
// waves per SIMD the compiler is asked to leave room for (wave_stepper_columns picks the workgroup size
// from the register count it ends up with): the Float64 Richards instantiations sit within a register
// or two of the 80-register step (6 waves), so they are held to it
// A value every lane of the wave holds identically (the per-column constants of the one column a wave of
// the wave stepper owns), moved to the scalar register file word by word: the vector registers it
// occupied are free again, and it is used as a uniform operand from then on.
template <typename T>
__device__ __forceinline__ T wave_uniform(const T& x) {
    static_assert(sizeof(T) % 4 == 0, "32-bit words");
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &x, sizeof(T));
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; ++k) w[k] = __builtin_amdgcn_readfirstlane(w[k]);
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}

template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename M, int CW, bool NOICE>
constexpr int cs_wave_min_waves() {
    if (!M::uses_tables || MODEL != MODEL_RICHARDS) return 4; // (1024 threads = 4 waves per SIMD)
    // (per-column parameters are scalar registers here, wave_uniform: the same budgets hold)
    if (CW == 1) return 6; // (the ice-free kernel needs 61 of its own accord; held to 64 it runs 4 % slower)
    return (NOICE && !FACTORS) ? 6 : 4;
}
template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename M, int CW, bool NOICE = false, bool VGF = true>
__global__ void __launch_bounds__(1024, (cs_wave_min_waves<FT, MODEL, FACTORS, PERCOL, M, CW, NOICE>()))
column_stepper_wave_kernel(const DevParams<FT> P0, const Planes<FT> Y, const Planes<FT> AUX, const FT dt_value,
                           const FT* __restrict__ dt_device, const int64_t nsteps, const FT* __restrict__ bcv) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    __shared__ double s_tab[M::uses_tables ? MATH_TAB_DOUBLES : 2];
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int n = P0.nlev; // <= 64 CW (the launcher's choice of CW)
    const int slot = int(threadIdx.x) >> 6; // column of this workgroup
    const int l = int(threadIdx.x) & 63;    // lane = CW adjacent cells
    const int cpb = int(blockDim.x) >> 6;
    const FT dt = dt_device ? *dt_device : dt_value;
    // per column: the top cell of every lane -- (K, -psi), (T, kappa), rho_e_l K as the model needs --
    // and the flux(es) of the face below every lane's bottom cell; 64 words each
    // (+ CS_FACE_WORDS per column: K, psi, kappa of the face state of a Dirichlet face, bottom and top)
    constexpr int NEX = cs_exchange_arrays<MODEL>();
    constexpr int NARR = NEX + cs_flux_arrays<MODEL>();
    constexpr size_t COLW = size_t(NARR) * 64 + CS_FACE_WORDS;
    FT* sK = reinterpret_cast<FT*>(s_dyn) + size_t(slot) * COLW;
    FT* sh = sK + 64;
    FT* sT = WATER ? sh + 64 : sK;
    FT* sKap = sT + 64;
    FT* sE = sKap + 64;
    FT* sFw = sK + size_t(NEX) * 64;
    FT* sFe = WATER ? sFw + 64 : sFw;
    FT* sFace = sK + size_t(NARR) * 64;
    const M mm(stage_math_tables<M>(P0.math_tab, s_tab));
    if (!M::uses_tables) __syncthreads();
    DevParams<FT> P = P0; // boundary values change per stage
    unsigned blk = blockIdx.x; // XCD-contiguous workgroup map (as rhs_kernel)
    if (P.xcd_remap) {
        const unsigned per = gridDim.x >> 3;
        if (blk < (per << 3)) blk = (blk & 7u) * per + (blk >> 3);
    }
    const int64_t col_raw = int64_t(blk) * cpb + slot;
    const int64_t col = col_raw < P.ncols ? col_raw : P.ncols - 1; // spare slots shadow the last column
    ColC<FT> c = make_colc<FT, M>(P, col, PERCOL);
    if (WATER && !NOICE) finish_colc<FT, M>(mm, c);
    // one wave = one column: its per-column constants are uniform (rhs_kernel, one LANE per column,
    // has to keep them in vector registers)
    if constexpr (PERCOL) c = wave_uniform(c);
    constexpr bool vgf = VGF && M::uses_tables;
    constexpr bool RELK = M::is_production; // fluxes in tendency units, K without Ksat: as rhs_kernel, to the letter
    const FT cgT = P.cg2;
    const FT Ksc = RELK ? c.Ksat : FT(1);
    const FT cgw = RELK ? c.cgw : cgT;
    const bool need_Taux = (MODEL == MODEL_RICHARDS) && FACTORS && P.viscosity_kind;
    // planes are column-fastest, lanes here level-fastest: through LDS tiles [cpb][n] (column_stepper_kernel)
    FT* tiles = reinterpret_cast<FT*>(s_dyn);
    const int64_t col_first = int64_t(blk) * cpb;
    const int tile_n = n * cpb;
    auto request = [&](const FT* plane, int k) {
        for (int e = threadIdx.x; e < tile_n; e += blockDim.x) {
            const int lev = e / cpb, cs = e - lev * cpb;
            if (col_first + cs < P.ncols) tiles[k * tile_n + cs * n + lev] = plane[int64_t(lev) * P.stride + col_first + cs];
        }
    };
    // (level-uniform prescribed fields of Ya come from their [nlev] arrays: DevParams::aux_prof)
    const FT* pf_vl = MODEL == MODEL_HEAT ? P.aux_prof[0] : nullptr;
    const FT* pf_ti = MODEL == MODEL_HEAT ? P.aux_prof[1] : nullptr;
    const FT* pf_T = need_Taux ? P.aux_prof[3] : nullptr;
    if (!pf_vl) request(MODEL == MODEL_HEAT ? AUX.v[0] : Y.v[0], 0);
    if (!NOICE && !pf_ti) request(MODEL == MODEL_HEAT ? AUX.v[1] : Y.v[1], 1);
    if (HEAT) request(Y.v[2], 2);
    if (need_Taux && !pf_T) request(AUX.v[3], 3);
    __syncthreads();
    FT y_vl[CW], y_re[CW], ti[CW], Ta[CW];
    bool act[CW]; // the cell exists (and the column does)
#pragma unroll
    for (int q = 0; q < CW; ++q) {
        const int i = CW * l + q;
        const int ic = i < n ? i : n - 1;
        act[q] = i < n && col_raw < P.ncols;
        const bool have = col_raw < P.ncols;
        y_vl[q] = pf_vl ? pf_vl[ic] : (have ? tiles[slot * n + ic] : FT(0));
        ti[q] = NOICE ? FT(0) : (pf_ti ? pf_ti[ic] : (have ? tiles[tile_n + slot * n + ic] : FT(0))); // NOICE: the theta_i plane is known to be all zeros
        y_re[q] = (HEAT && have) ? tiles[2 * tile_n + slot * n + ic] : FT(0);
        Ta[q] = need_Taux ? (pf_T ? pf_T[ic] : (have ? tiles[3 * tile_n + slot * n + ic] : FT(288))) : FT(288);
    }
    __syncthreads(); // the tiles overlay other columns' exchange arrays
    const int lt = (n - 1) / CW, qt = (n - 1) - lt * CW; // lane and slot of the top cell
    const bool at_bottom = (l == 0), at_top = (l == lt);
    // Dirichlet faces: the closures of the FACE state -- a whole extra closure pass by one lane while
    // its wave waits -- read the boundary value, the boundary cell's theta_i and (some models) its
    // vartheta_l or T.  With constant boundary values and none of those moving (face_state_is_static)
    // they are constants of the call: evaluated once here, kept in LDS, and a stage only assembles
    // the two fluxes from them (boundary_fluxes_from) -- the same numbers as boundary_fluxes.
    const bool hoist_b = !bcv && face_state_is_static<FT, MODEL, FACTORS>(P, FACE_BOTTOM);
    const bool hoist_t = !bcv && face_state_is_static<FT, MODEL, FACTORS>(P, FACE_TOP);
    auto hoist_face = [&](int face, int q_cell) {
        FT bv = y_vl[0], bti = ti[0], bT = Ta[0];
#pragma unroll
        for (int q = 1; q < CW; ++q)
            if (q == q_cell) { bv = y_vl[q]; bti = ti[q]; bT = Ta[q]; }
        const FaceState<FT> fs = face_state<FT, M, MODEL, FACTORS, NOICE>(mm, P, c, face, col, bv, bti, bT, vgf);
        FT* dst = sFace + (face == FACE_BOTTOM ? 0 : 3);
        dst[0] = fs.K;
        dst[1] = fs.psi;
        dst[2] = fs.kap;
    };
    if ((at_bottom && hoist_b) || (at_top && hoist_t && !at_bottom)) // (one divergent pass for both lanes)
        hoist_face(at_bottom ? FACE_BOTTOM : FACE_TOP, at_bottom ? 0 : qt);
    if (at_bottom && at_top && hoist_t) hoist_face(FACE_TOP, qt); // a column of <= CW cells
    wave_sync();
    FT nf_acc = FT(0);
    for (int64_t s = 0; s < nsteps; ++s) {
        FT u_vl[CW], u_re[CW]; // the stage state
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            u_vl[q] = y_vl[q];
            u_re[q] = y_re[q];
        }
#pragma unroll 1
        for (int stage = 0; stage < 3; ++stage) {
            if (bcv) {
                const FT* b = bcv + (s * 3 + stage) * 4;
                P.bc_value[0][0] = b[0];
                P.bc_value[0][1] = b[1];
                P.bc_value[1][0] = b[2];
                P.bc_value[1][1] = b[3];
            }
            FT T[CW], kap[CW], K[CW], psi[CW], E[CW];
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                FT rcs = FT(1);
                T[q] = Ta[q];
                kap[q] = K[q] = psi[q] = E[q] = FT(0);
                if (HEAT) {
                    T[q] = temperature_closure<FT, M, NOICE>(mm, P, c, u_vl[q], ti[q], u_re[q], rcs);
                    kap[q] = kappa_closure<FT, M, NOICE>(mm, P, c, u_vl[q], ti[q]);
                }
                if (WATER) {
                    water_closures<FT, M, FACTORS, true, false, NOICE, RELK, HEAT, true>(mm, P, c, u_vl[q], ti[q], T[q], K[q], psi[q], nullptr, vgf); // psi: -psi
                    if (HEAT) E[q] = (P.rhocp_l * (T[q] - P.T_ref)) * K[q]; // rho_e_int_l * K (:364)
                }
            }
            // the lane's top cell, for the lane above
            if (WATER) { sK[l] = K[CW - 1]; sh[l] = psi[CW - 1]; }
            if (HEAT) { sT[l] = T[CW - 1]; sKap[l] = kap[CW - 1]; }
            if (HEAT && WATER) sE[l] = E[CW - 1];
            wave_sync();
            // boundary faces: the lane with the bottom cell and the lane with the top cell go through
            // boundary_fluxes TOGETHER (one divergent pass, not two, when both faces need closures)
            FT Fw_b = FT(0), Fe_b = FT(0), Fw_t = FT(0), Fe_t = FT(0);
            if (at_bottom || at_top) {
                // (the top cell's slot qt is uniform: a chain of selects over the unrolled slots)
                FT bv = u_vl[0], bti = ti[0], bT = T[0], bK = K[0], bpsi = psi[0];
                if (!at_bottom) {
#pragma unroll
                    for (int q = 1; q < CW; ++q)
                        if (q == qt) { bv = u_vl[q]; bti = ti[q]; bT = T[q]; bK = K[q]; bpsi = psi[q]; }
                }
                auto face_fluxes = [&](int face, bool hoisted, FT& fe, FT& fw) {
                    FaceState<FT> fs;
                    if (hoisted) {
                        const FT* src = sFace + (face == FACE_BOTTOM ? 0 : 3);
                        fs.K = src[0];
                        fs.psi = src[1];
                        fs.kap = src[2];
                        fs.T = face_state_T<FT, MODEL>(P, face, col, bT);
                    } else {
                        fs = face_state<FT, M, MODEL, FACTORS, NOICE>(mm, P, c, face, col, bv, bti, bT, vgf);
                    }
                    boundary_fluxes_from<FT, MODEL>(P, fs, face, col, bT, bK * Ksc, -bpsi, fe, fw);
                    fe = fe * P.inv_dz;
                    fw = fw * P.inv_dz;
                };
                FT fe, fw;
                face_fluxes(at_bottom ? FACE_BOTTOM : FACE_TOP, at_bottom ? hoist_b : hoist_t, fe, fw);
                if (at_bottom) { Fe_b = fe; Fw_b = fw; }
                else { Fe_t = fe; Fw_t = fw; }
                if (at_bottom && at_top) { // a column of <= CW cells: the same lane owns both faces
#pragma unroll
                    for (int q = 0; q < CW; ++q)
                        if (q == qt) { bv = u_vl[q]; bti = ti[q]; bT = T[q]; bK = K[q]; bpsi = psi[q]; }
                    face_fluxes(FACE_TOP, hoist_t, Fe_t, Fw_t);
                }
            }
            // F[q]: the face below the lane's cell q; F[CW]: the face above its top cell
            FT Fw[CW + 1], Fe[CW + 1];
#pragma unroll
            for (int q = 0; q <= CW; ++q) Fw[q] = Fe[q] = FT(0);
            if (at_bottom) {
                Fw[0] = Fw_b;
                Fe[0] = Fe_b;
            } else { // (as rhs_kernel: lower cell first)
                FT gh = FT(0);
                if (WATER) {
                    gh = head_difference(psi[0], sh[l - 1], P.dz) * cgw;
                    Fw[0] = -(sK[l - 1] + K[0]) * gh;
                }
                if (HEAT) {
                    const FT gT = (T[0] - sT[l - 1]) * cgT;
                    Fe[0] = -(sKap[l - 1] + kap[0]) * gT;
                    if (WATER) Fe[0] = Fe[0] - (sE[l - 1] + E[0]) * gh;
                }
            }
#pragma unroll
            for (int q = 1; q < CW; ++q) {
                FT gh = FT(0);
                if (WATER) {
                    gh = head_difference(psi[q], psi[q - 1], P.dz) * cgw;
                    Fw[q] = -(K[q - 1] + K[q]) * gh;
                }
                if (HEAT) {
                    const FT gT = (T[q] - T[q - 1]) * cgT;
                    Fe[q] = -(kap[q - 1] + kap[q]) * gT;
                    if (WATER) Fe[q] = Fe[q] - (E[q - 1] + E[q]) * gh;
                }
            }
            // the face above the lane's top cell is the face below the next lane's bottom cell
            if (WATER) sFw[l] = Fw[0];
            if (HEAT) sFe[l] = Fe[0];
            wave_sync();
            if (l < 63) {
                if (WATER) Fw[CW] = sFw[l + 1];
                if (HEAT) Fe[CW] = sFe[l + 1];
            }
            if (at_top) { // the face above the column's top cell (slot qt) is the boundary
#pragma unroll
                for (int q = 0; q < CW; ++q)
                    if (q == qt) { Fw[q + 1] = Fw_t; Fe[q + 1] = Fe_t; }
            }
            // the stage updates of rhs_kernel MODE 1..3 (b = Y, u = stage state)
            auto upd = [&](FT b, FT u, FT k) -> FT {
                if (stage == 0) return u + dt * k;
                if (stage == 1) return (FT(3) * b + u + dt * k) * FT(0.25);
                const FT sum = b + FT(2) * u + FT(2) * dt * k;
                const FT qq = sum * FT(1.0 / 3.0);
                return fma_ft(fma_ft(FT(-3), qq, sum), FT(1.0 / 3.0), qq);
            };
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const FT dvl = WATER ? Fw[q] - Fw[q + 1] : FT(0);
                const FT dre = HEAT ? Fe[q] - Fe[q + 1] : FT(0);
                if (act[q]) {
                    if (WATER) nf_acc = fma_ft(dvl, FT(0), nf_acc);
                    if (HEAT) nf_acc = fma_ft(dre, FT(0), nf_acc);
                }
                if (WATER) u_vl[q] = upd(y_vl[q], u_vl[q], dvl);
                if (HEAT) u_re[q] = upd(y_re[q], u_re[q], dre);
            }
            wave_sync(); // the neighbours have read this stage's LDS values
        }
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            y_vl[q] = u_vl[q];
            y_re[q] = u_re[q];
        }
    }
    __syncthreads(); // the tiles overlay other columns' exchange arrays
#pragma unroll
    for (int q = 0; q < CW; ++q) {
        const int i = CW * l + q;
        if (i < n) {
            if (WATER) tiles[slot * n + i] = y_vl[q];
            if (HEAT) tiles[tile_n + slot * n + i] = y_re[q];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < tile_n; e += blockDim.x) {
        const int lev = e / cpb, cs = e - lev * cpb;
        if (col_first + cs < P.ncols) {
            const int64_t g = int64_t(lev) * P.stride + col_first + cs;
            if (WATER) Y.v[0][g] = tiles[cs * n + lev];
            if (HEAT) Y.v[2][g] = tiles[tile_n + cs * n + lev];
        }
    }
    if (col_raw < P.ncols && nf_acc != nf_acc) atomicOr(P.status, 1u);
}

// --------------------------------------------------------- diagnostics
// K, psi, kappa, T of every cell (the pointwise stage only).
template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename M>
__global__ void __launch_bounds__(256)
diag_kernel(const DevParams<FT> P, const Planes<FT> IN, const Planes<FT> AUX, const Planes<FT> OUT) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    __shared__ double s_tab[M::uses_tables ? MATH_TAB_DOUBLES : 1];
    const M mm(stage_math_tables<M>(P.math_tab, s_tab));
    const int64_t col = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (col >= P.ncols) return;
    const FT* p_vl = (MODEL == MODEL_HEAT ? AUX.v[0] : IN.v[0]) + col;
    const FT* p_ti = (MODEL == MODEL_HEAT ? AUX.v[1] : IN.v[1]) + col;
    const bool need_Taux = (MODEL == MODEL_RICHARDS) && FACTORS && P.viscosity_kind;
    ColC<FT> c = make_colc<FT, M>(P, col, PERCOL);
    if (WATER) finish_colc<FT, M>(mm, c);
    const bool vgf = M::uses_tables && P.vg_fast_all != 0; // (kernel-argument constant: a scalar branch)
    for (int i = 0; i < P.nlev; ++i) {
        const int64_t o = int64_t(i) * P.stride;
        FT vl = p_vl[o], ti = p_ti[o];
        FT T = need_Taux ? AUX.v[3][o + col] : FT(288), kap = FT(0), K = FT(0), psi = FT(0);
        if (HEAT) {
            FT rcs;
            T = temperature_closure<FT, M>(mm, P, c, vl, ti, IN.v[2][o + col], rcs);
            kap = kappa_closure<FT, M>(mm, P, c, vl, ti);
        }
        if (WATER) water_closures<FT, M, FACTORS>(mm, P, c, vl, ti, T, K, psi, nullptr, vgf);
        OUT.v[0][o + col] = K;
        OUT.v[1][o + col] = psi;
        OUT.v[2][o + col] = kap;
        OUT.v[3][o + col] = T;
    }
}

// ------------------------------------------------------ boundary fluxes
// boundary_fluxes(X, bc, face, model, cs, t) (boundary_conditions.jl:470-489; the prescribed
// atmosphere :516-533 arrives as per-column flux values) for ONE face of every column: the pair
// (f_rhoe_int, f_vartheta_l) the tendency launch puts into its SetValue divergence, from the same
// device functions, the same closure instantiation (FACTORS, math policy, K without Ksat times Ksat)
// and therefore the same bits.  A component without a boundary condition (NoBC) gives NaN.
template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename M>
__global__ void __launch_bounds__(256)
boundary_flux_kernel(const DevParams<FT> P, const Planes<FT> IN, const Planes<FT> AUX, const int face, FT* out_e, FT* out_w) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    __shared__ double s_tab[M::uses_tables ? MATH_TAB_DOUBLES : 1];
    const M mm(stage_math_tables<M>(P.math_tab, s_tab));
    const int64_t col = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (col >= P.ncols) return;
    // interior_values (:174-190): the cell next to the face
    const int64_t o = int64_t(face == FACE_BOTTOM ? 0 : P.nlev - 1) * P.stride + col;
    ColC<FT> c = make_colc<FT, M>(P, col, PERCOL);
    if (WATER) finish_colc<FT, M>(mm, c);
    const bool vgf = M::uses_tables && P.vg_fast_all != 0;
    const bool need_Taux = (MODEL == MODEL_RICHARDS) && FACTORS && P.viscosity_kind;
    const FT vl = (MODEL == MODEL_HEAT ? AUX.v[0] : IN.v[0])[o];
    const FT ti = (MODEL == MODEL_HEAT ? AUX.v[1] : IN.v[1])[o];
    FT T = need_Taux ? AUX.v[3][o] : FT(288), K = FT(0), psi = FT(0), rcs = FT(1);
    if (HEAT) T = temperature_closure<FT, M>(mm, P, c, vl, ti, IN.v[2][o], rcs);
    constexpr bool RELK = M::is_production;
    if (WATER) {
        water_closures<FT, M, FACTORS, true, false, false, RELK>(mm, P, c, vl, ti, T, K, psi, nullptr, vgf);
        if (RELK) K = K * c.Ksat; // (as rhs_kernel hands the boundary cell's conductivity over)
    }
    FT fe, fw;
    boundary_fluxes<FT, M, MODEL, FACTORS, false>(mm, P, c, face, col, vl, ti, T, K, psi, fe, fw, nullptr, nullptr, vgf);
    if (P.bc_kind[face][COMP_ENERGY] == BC_NONE) fe = FT(NAN);
    if (P.bc_kind[face][COMP_HYDROLOGY] == BC_NONE) fw = FT(NAN);
    out_e[col] = fe;
    out_w[col] = fw;
}

// ---------------------------------------------------------- stable dt
// min over cells of courant*dz^2/max(K dpsi/dvl, kappa/rho_c_s) (build-defined;
// the reference steps with a fixed user dt, simulation.jl:34-70).  Positive IEEE
// values order like their bit patterns, so the global min is an integer atomicMin.

template <typename FT, int MODEL, bool PERCOL, typename M>
__global__ void __launch_bounds__(256)
stable_dt_kernel(const DevParams<FT> P, const Planes<FT> IN, const Planes<FT> AUX, const FT courant,
                 typename Bits<FT>::type* out_bits) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    using U = typename Bits<FT>::type;
    __shared__ double s_tab[M::uses_tables ? MATH_TAB_DOUBLES : 1];
    const M mm(stage_math_tables<M>(P.math_tab, s_tab));
    const int64_t col = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    FT best = FT(INFINITY);
    if (col < P.ncols) {
        const FT* p_vl = (MODEL == MODEL_HEAT ? AUX.v[0] : IN.v[0]) + col;
        const FT* p_ti = (MODEL == MODEL_HEAT ? AUX.v[1] : IN.v[1]) + col;
        ColC<FT> c = make_colc<FT, M>(P, col, PERCOL);
        if (WATER) finish_colc<FT, M>(mm, c);
        const FT cdz2 = courant * P.dz * P.dz;
        const int n = P.nlev;
        FT K_p = FT(0), dpsi_p = FT(0), kap_p = FT(0), rcs_p = FT(1);
        for (int i = 0; i < n; ++i) {
            const int64_t o = int64_t(i) * P.stride;
            FT vl = p_vl[o], ti = p_ti[o];
            FT K = FT(0), dpsi = FT(0), kap = FT(0), rcs = FT(1);
            FT Tc = FT(288);
            if (MODEL == MODEL_RICHARDS && P.viscosity_kind) Tc = AUX.v[3][o + col];
            if (HEAT) Tc = temperature_closure<FT, M>(mm, P, c, vl, ti, IN.v[2][o + col], rcs);
            if (WATER) {
                FT psi;
                water_closures<FT, M, true>(mm, P, c, vl, ti, Tc, K, psi);
                const FT nu_eff = c.nu - ti;
                const FT vls = !(vl <= c.theta_lim) ? vl : c.theta_lim;
                const FT Se = (vls - c.theta_r) / (nu_eff - c.theta_r);
                const FT u = mm.pow(Se, -c.inv_m) - FT(1);
                if (Se <= FT(1) && u > FT(0))
                    dpsi = fabs(psi) * (u + FT(1)) / (c.n * c.m * u * Se * (nu_eff - c.theta_r));
                else
                    dpsi = FT(1) / c.S_s;
            }
            if (HEAT) kap = kappa_closure<FT, M>(mm, P, c, vl, ti);
            FT D = FT(0);
            if (i == 0 || i == n - 1) { // boundary cells: their own coefficients
                D = K * dpsi;
                if (HEAT && kap / rcs > D) D = kap / rcs;
                // Dirichlet faces sit half a cell away and use the face state's coefficients
                for (int face = 0; face < 2; ++face) {
                    if ((face == FACE_BOTTOM) != (i == 0) && n > 1) continue;
                    const int kh = P.bc_kind[face][COMP_HYDROLOGY], ke = P.bc_kind[face][COMP_ENERGY];
                    FT vh = P.bc_value[face][COMP_HYDROLOGY];
                    if (P.bc_pc[face][COMP_HYDROLOGY]) vh = P.bc_pc[face][COMP_HYDROLOGY][col];
                    const FT vl_f = (WATER && kh == BC_DIRICHLET) ? vh : vl;
                    FT ve = P.bc_value[face][COMP_ENERGY];
                    if (P.bc_pc[face][COMP_ENERGY]) ve = P.bc_pc[face][COMP_ENERGY][col];
                    const FT T_f = (HEAT && ke == BC_DIRICHLET) ? ve : Tc; // as boundary_fluxes
                    if (WATER && kh == BC_DIRICHLET) {
                        FT K_f, psi_f;
                        water_closures<FT, M, true, false>(mm, P, c, vl_f, ti, T_f, K_f, psi_f);
                        FT Db = FT(2) * (K_f > K ? K_f : K) * dpsi;
                        if (Db > D) D = Db;
                    }
                    if (HEAT && ke == BC_DIRICHLET) {
                        FT kf = kappa_closure<FT, M>(mm, P, c, vl_f, ti);
                        FT Db = FT(2) * (kf > kap ? kf : kap) / rcs;
                        if (Db > D) D = Db;
                    }
                }
            }
            if (i > 0) { // interior face: arithmetic-mean coefficients as in the stencil
                FT Dw = (K_p + K) * FT(0.5) * (dpsi_p > dpsi ? dpsi_p : dpsi);
                FT DT = HEAT ? (kap_p + kap) * FT(0.5) / (rcs_p < rcs ? rcs_p : rcs) : FT(0);
                if (Dw > D) D = Dw;
                if (DT > D) D = DT;
            }
            if (D > FT(0)) {
                FT dtc = cdz2 / D;
                if (dtc < best) best = dtc;
            }
            K_p = K;
            dpsi_p = dpsi;
            kap_p = kap;
            rcs_p = rcs;
        }
    }
    // wave64 reduction, then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        FT other = __shfl_down(best, off, 64);
        if (other < best) best = other;
    }
    if ((threadIdx.x & 63) == 0 && best < FT(INFINITY)) {
        U b;
        __builtin_memcpy(&b, &best, sizeof(FT));
        if (b < __atomic_load_n(out_bits, __ATOMIC_RELAXED)) atomicMin(out_bits, b); // (as rhs_kernel MODE 4: a read first)
    }
}

// ------------------------------------------------ layout conversion kernels
// host/user layout a[col*cs + lev*ls]  <->  plane [lev][stride] (column-fastest)
template <typename FT, bool TO_PLANE>
__global__ void __launch_bounds__(256)
strided_copy_kernel(FT* plane, int64_t stride, FT* user, int64_t ls, int64_t cs, int64_t ncols,
                    int nlev) {
    // 64x64 tile through LDS so both sides stay coalesced for the level-fastest
    // case (ls == 1); correct for any strides.
    __shared__ FT tile[64][65];
    const int64_t c0 = int64_t(blockIdx.x) * 64;
    const int l0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
    if (TO_PLANE) {
        for (int r = ty; r < 64; r += 4) { // r = column within tile, tx = level
            int64_t c = c0 + r;
            int l = l0 + tx;
            if (c < ncols && l < nlev) tile[r][tx] = user[c * cs + int64_t(l) * ls];
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) { // r = level within tile, tx = column
            int64_t c = c0 + tx;
            int l = l0 + r;
            if (c < ncols && l < nlev) plane[int64_t(l) * stride + c] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 64; r += 4) {
            int64_t c = c0 + tx;
            int l = l0 + r;
            if (c < ncols && l < nlev) tile[tx][r] = plane[int64_t(l) * stride + c];
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {
            int64_t c = c0 + r;
            int l = l0 + tx;
            if (c < ncols && l < nlev) user[c * cs + int64_t(l) * ls] = tile[r][tx];
        }
    }
}

// ------------------------------------------------------- streaming probe
// The access pattern of rhs_kernel with the arithmetic taken out: one lane per CPL columns
// marches bottom -> top, reads one row of each of NR planes per level (PF levels in flight),
// and stores their sum into each of NW planes -- same buffer-descriptor row access, same
// nontemporal policy, same workgroup -> column-block map.  Its rate is the ceiling the column
// launch can reach on THIS set of planes (lh_stream_probe).
template <typename FT, int CPL, int PF, bool NT>
__global__ void __launch_bounds__(256, 8)
stream_probe_kernel(const int64_t ncols, const int64_t stride, const int nlev, const int xcd_remap,
                    const Planes<FT> IN, const int nr, const Planes<FT> OUT, const int nw) {
    static_assert(sizeof(FT) * CPL == 8, "one 8-byte access per lane");
    unsigned blk = blockIdx.x;
    if (xcd_remap) {
        const unsigned per = gridDim.x >> 3;
        if (blk < (per << 3)) blk = (blk & 7u) * per + (blk >> 3);
    }
    const int64_t col0 = (int64_t(blk) * blockDim.x + threadIdx.x) * CPL;
    if (col0 >= ncols) return;
    const unsigned lane_byte = (unsigned)col0 * (unsigned)sizeof(FT);
    const unsigned row_bytes = (unsigned)(stride * (int64_t)sizeof(FT));
    const FT* r[4];
    FT* w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = IN.v[k];
        w[k] = OUT.v[k];
    }
    FT ring[PF][4][CPL];
#pragma unroll
    for (int q = 0; q < PF; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < CPL; ++j) ring[q][k][j] = FT(0);
    auto fetch = [&](int slot) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nr) {
                bload<FT, CPL, NT>(r[k], row_bytes, lane_byte, ring[slot][k]);
                r[k] += stride;
            }
    };
#pragma unroll
    for (int q = 0; q < PF; ++q)
        if (q < nlev) fetch(q);
    for (int i0 = 0; i0 < nlev; i0 += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int i = i0 + q;
            if (PF > 1 && i >= nlev) break;
            FT x[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) x[j] = (ring[q][0][j] + ring[q][1][j]) + (ring[q][2][j] + ring[q][3][j]);
            if (i + PF < nlev) fetch(q);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nw) {
                    bstore<FT, CPL, NT>(w[k], row_bytes, lane_byte, x);
                    w[k] += stride;
                }
        }
    }
}

template <typename FT>
__global__ void __launch_bounds__(256) fill_kernel(FT* p, int64_t n, FT v) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t step = int64_t(gridDim.x) * blockDim.x;
    for (; i < n; i += step) p[i] = v;
}

template <typename FT>
__global__ void __launch_bounds__(256) broadcast_profile_kernel(FT* plane, const FT* __restrict__ prof, int64_t ncols, int64_t stride) {
    const int64_t col = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (col < ncols) plane[int64_t(blockIdx.y) * stride + col] = prof[blockIdx.y];
}

template <typename FT>
__global__ void __launch_bounds__(256) convert_kernel(FT* dst, const double* src, int64_t n) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = FT(src[i]);
}

// one thread: dt = min(dt, dt_max) (dt_max <= 0: no cap), elapsed += dt.  A bound that is not a
// positive finite number (no positive diffusivity anywhere and no cap: +inf; a NaN) would turn the
// state into NaNs: the step is then taken with dt = 0 (the state stays) and bit 2 of the status is set.
template <typename FT>
__global__ void dt_prepare_kernel(FT* dt, FT dt_max, FT* elapsed, uint32_t* status) {
    FT d = *dt;
    if (dt_max > FT(0) && !(d <= dt_max)) d = dt_max;
    if (!(d > FT(0)) || d - d != FT(0)) {
        d = FT(0);
        atomicOr(status, 4u);
    }
    *dt = d;
    if (elapsed) *elapsed += d;
}

template <typename FT>
__global__ void init_bits_kernel(typename Bits<FT>::type* p) {
    FT inf = FT(INFINITY);
    __builtin_memcpy(p, &inf, sizeof(FT));
}

// ------------------------------------------------------------- launchers

static inline dim3 grid_for(int64_t work, int block) {
    return dim3((unsigned)((work + block - 1) / block));
}

template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename CFG, typename M, bool NOICE, bool VGF = true>
static void launch_rhs_mode(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                            const Planes<FT>& base, const Planes<FT>& out, FT dt, const FT* dt_dev,
                            int mode, int block_req, hipStream_t s) {
    // workgroup size: the kernel's own (see rhs_max_threads) unless LH_TUNE block= asks for less
    const int mode_k = M::is_production ? mode : 0;
    int kmax = 256;
#define LH_KMAX(MD) case MD: kmax = rhs_max_threads<M, rhs_waves_per_simd<FT, MODEL, FACTORS, PERCOL, M, CFG::PF, MD, NOICE && M::is_production, CFG::CPL>()>(); break;
    switch (mode_k) { LH_KMAX(0) LH_KMAX(1) LH_KMAX(2) LH_KMAX(3) LH_KMAX(4) LH_KMAX(5) }
#undef LH_KMAX
    const int block = (block_req > 0 && block_req <= kmax) ? block_req : kmax;
    const int64_t lanes = (P.ncols + CFG::CPL - 1) / CFG::CPL;
    dim3 g = grid_for(lanes, block), b(block);
    if (CFG::SEG) g.y = (unsigned)((P.nlev + P.seg_len - 1) / P.seg_len);
    // dynamic LDS: one word per thread for the mode-4 reduction
    // (+ two level arrays for level-uniform prescribed fields of Ya where the kernel can take them)
    constexpr bool MAY_PROF = (MODEL == MODEL_HEAT) || (MODEL == MODEL_RICHARDS && FACTORS);
    const unsigned dyn = (unsigned)((mode == 4 ? (((size_t)block * sizeof(float) + 15) & ~(size_t)15) : 0) +
                                    (MAY_PROF ? 2 * (size_t)P.nlev * sizeof(FT) : 0));
    if constexpr (!M::is_production) { // MathLibm: tendency only (the other modes are never instantiated)
        hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 0, false>), g, b, dyn, s, P, in, aux, base, out, dt, dt_dev);
    } else {
        switch (mode) {
            case 0: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 0, NOICE, VGF>), g, b, dyn, s, P, in, aux, base, out, dt, dt_dev); break;
            case 1: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 1, NOICE, VGF>), g, b, dyn, s, P, in, aux, base, out, dt, dt_dev); break;
            case 2: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 2, NOICE, VGF>), g, b, dyn, s, P, in, aux, base, out, dt, dt_dev); break;
            case 3: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 3, NOICE, VGF>), g, b, dyn, s, P, in, aux, base, out, dt, dt_dev); break;
            case 5: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 5, NOICE, VGF>), g, b, dyn, s, P, in, aux, base, out, dt, dt_dev); break;
            default: // 4: tendency + stable-step bound; the minimum starts at +inf
                hipLaunchKernelGGL((init_bits_kernel<FT>), dim3(1), dim3(1), 0, s, reinterpret_cast<typename Bits<FT>::type*>(P.dt_out));
                hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 4, NOICE, VGF>), g, b, dyn, s, P, in, aux, base, out, dt, dt_dev);
                break;
        }
    }
}

template <typename FT, int MODEL, typename M>
static void launch_rhs_model(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                             const Planes<FT>& base, const Planes<FT>& out, FT dt, const FT* dt_dev,
                             int mode, bool factors, bool percol, bool noice, const Tune& tune, hipStream_t s) {
    using CFG = typename DefaultCfg<FT>::type;
    const int block = tune.block; // 0: the kernel's own workgroup size
#ifdef LH_TUNING_VARIANTS
    // tuning builds: alternative columns-per-lane / prefetch depth for the plain tendency kernels
    if (!factors && !percol && mode == 0 && M::is_production && (tune.cpl > 0 || tune.pf > 0)) {
        const bool ntv = tune.nt >= 0 ? tune.nt != 0 : true;
        const int cplv = tune.cpl > 0 ? tune.cpl : CFG::CPL, pfv = tune.pf > 0 ? tune.pf : CFG::PF;
#define LH_TRY(C, F, N)                                                                            \
    if (cplv == C && pfv == F && ntv == N) {                                                       \
        if (noice) launch_rhs_mode<FT, MODEL, false, false, KCfg<C, F, N>, M, true>(P, in, aux, base, out, dt, dt_dev, 0, block, s); \
        else launch_rhs_mode<FT, MODEL, false, false, KCfg<C, F, N>, M, false>(P, in, aux, base, out, dt, dt_dev, 0, block, s); \
        return;                                                                                    \
    }
        LH_TRY(1, 1, true) LH_TRY(1, 2, true) LH_TRY(1, 3, true) LH_TRY(1, 4, true)
        LH_TRY(2, 1, true) LH_TRY(2, 2, true) LH_TRY(1, 1, false) LH_TRY(1, 2, false) LH_TRY(4, 1, true) LH_TRY(4, 2, true)
#undef LH_TRY
    }
#endif
    // Nontemporal access when the launch streams more than the 256 MiB Infinity
    // Cache can hold (measured +4 % on 1e6 x 64 columns); plain access for small
    // ensembles whose planes stay cache-resident between launches.
    const int nplanes = (MODEL == MODEL_COUPLED) ? 6 : (MODEL == MODEL_RICHARDS ? 4 : 4);
    const double touched = double(P.nlev) * double(P.stride) * sizeof(FT) * nplanes;
    const bool nt = tune.nt >= 0 ? tune.nt != 0 : touched > 192.0 * 1024 * 1024;
    using CFGN = KCfg<CFG::CPL, CFG::PF, true>;
    using CFGP = KCfg<CFG::CPL, CFG::PF, false>;
    using CFGS = KCfg<CFG::CPL, CFG::PF, false, true>; // level-segmented (small ensembles: cache-resident)
    const bool seg = M::is_production && P.seg_len > 0 && P.seg_len < P.nlev;
    // the no-ice kernels exist for the production math without conductivity factors
    const bool ni = noice && M::is_production && !factors;
    // clay-like ensembles (some column with m < LH_VG_FAST_MIN_M): the log-domain closures, one launch
    // shape (plain access, unsegmented) -- the Float64 water kernels only
    constexpr bool has_robust = M::uses_tables && MODEL != MODEL_HEAT;
    const bool robust = has_robust && P.vg_fast_all == 0;
#define LH_GO3(F, PC, NI)                                                                             \
    do {                                                                                              \
        if constexpr (has_robust) {                                                                   \
            if (robust) {                                                                             \
                launch_rhs_mode<FT, MODEL, F, PC, CFGP, M, NI, false>(P, in, aux, base, out, dt, dt_dev, mode, block, s); \
                break;                                                                                \
            }                                                                                         \
        }                                                                                             \
        if (seg) launch_rhs_mode<FT, MODEL, F, PC, CFGS, M, NI>(P, in, aux, base, out, dt, dt_dev, mode, block, s); \
        else if (nt) launch_rhs_mode<FT, MODEL, F, PC, CFGN, M, NI>(P, in, aux, base, out, dt, dt_dev, mode, block, s); \
        else launch_rhs_mode<FT, MODEL, F, PC, CFGP, M, NI>(P, in, aux, base, out, dt, dt_dev, mode, block, s);   \
    } while (0)
    if (factors) {
        if (percol) LH_GO3(true, true, false);
        else LH_GO3(true, false, false);
    } else if (ni) {
        if (percol) LH_GO3(false, true, true);
        else LH_GO3(false, false, true);
    } else {
        if (percol) LH_GO3(false, true, false);
        else LH_GO3(false, false, false);
    }
#undef LH_GO3
}

// All rhs_kernel variants of one model.  Each (FT, MODEL) pair is instantiated in its own
// translation unit (lh_kernels_<ft>_<model>.hip) so the build runs eight compilers side by side.
template <typename FT, int MODEL>
void launch_rhs_for_model(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                          const Planes<FT>& base, const Planes<FT>& out, FT dt, const FT* dt_dev,
                          int mode, bool factors, bool percol, bool noice, int math, const Tune& tune,
                          hipStream_t s) {
    // MathLibm is a parity-debugging policy for the tendency itself (mode 0);
    // the fused SSPRK33 stages and mode 4 always run the production math.
    if (math == MATH_LIBM && mode == 0)
        launch_rhs_model<FT, MODEL, MathLibm<FT>>(P, in, aux, base, out, dt, dt_dev, mode, factors, percol, false, tune, s);
    else
        launch_rhs_model<FT, MODEL, MathFast<FT>>(P, in, aux, base, out, dt, dt_dev, mode, factors, percol, noice, tune, s);
}

// Columns (= waves) per workgroup of column_stepper_wave_kernel.  Whole multiples of 4, so the four
// SIMDs of a CU carry the same number of waves (measured on 1e6 x 64 Float64 columns, ms per step:
// 8 columns 0.60, 9..11 0.72..0.85, 12 0.56, 13 0.85, 14 0.79, 16 0.71), and among 4 / 8 / 12 the
// count that keeps most waves resident under the instantiation's register count and the LDS of a
// CU (48 KiB of math tables per Float64 workgroup + the exchange arrays of its columns): 12 for
// Float64 Richards without ice (63 / 77 VGPRs, 6 waves per SIMD), 8 for the per-column-parameter
// and ice kernels (94+ VGPRs: a second 768-thread workgroup would not fit), 4 for Float32 (8 for
// short calls).  Small ensembles take 4 to spread over the CUs.
template <auto Kernel>
unsigned wave_stepper_columns(int64_t ncols, size_t dyn_col, int64_t nsteps) {
    static const hipFuncAttributes attr = [] {
        hipFuncAttributes a{};
        if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(Kernel)) != hipSuccess) a.numRegs = 0;
        return a;
    }();
    if (ncols < 4096 || attr.numRegs <= 0) return 4u;
    const unsigned regs = (unsigned(attr.numRegs) + 7u) & ~7u;
    const unsigned waves_cu = 4u * (512u / regs < 8u ? 512u / regs : 8u);
    unsigned best = 4u, best_resident = 0u;
    for (unsigned c = 4u; c <= 12u; c += 4u) {
        const size_t lds = attr.sharedSizeBytes + c * dyn_col;
        const unsigned by_lds = unsigned((size_t(160) << 10) / (lds ? lds : 1));
        const unsigned by_regs = waves_cu / c;
        const unsigned resident = c * (by_lds < by_regs ? by_lds : by_regs);
        // equal residency (Float32: no tables): 4 columns step fastest, 8 halve the row pieces of the
        // tile I/O -- T(n) = 1.26 + 0.422 n against 0.89 + 0.439 n ms on C3 (tools/stepper_call_cost.py):
        // the wider workgroup wins calls of up to 20 steps
        if (resident > best_resident || (resident == best_resident && c == 8u && nsteps <= 20)) best = c, best_resident = resident;
    }
    return best;
}

template <typename FT, int MODEL>
void launch_column_stepper_for_model(const DevParams<FT>& P, const Planes<FT>& Y, const Planes<FT>& aux,
                                     FT dt, const FT* dt_dev, int64_t nsteps, const FT* bcv, bool factors,
                                     bool percol, bool noice, hipStream_t s) {
    // One wave per column with 1 or 2 adjacent cells per lane (column_stepper_wave_kernel) for columns
    // of up to 128 levels; one thread per cell with workgroup barriers beyond that.
    const int cw = (P.nlev + 63) / 64;
    const bool wave = cw <= 2;
    const unsigned tpc = wave ? 64u : (unsigned)((P.nlev + 63) / 64 * 64);
    // columns per workgroup of the thread-per-cell kernel: 256 threads, 512 for large Float64
    // ensembles (the Float64 math tables take 48 KiB of LDS per workgroup whatever its size);
    // the wave kernel picks its own from the instantiation's registers (wave_stepper_columns)
    unsigned cpb = 256u / tpc ? 256u / tpc : 1u;
    if (sizeof(FT) == 8 && P.ncols >= 4096) cpb = 512u / tpc ? 512u / tpc : 1u;
    const bool need_Taux = (MODEL == MODEL_RICHARDS) && factors && P.viscosity_kind;
    const int tiles = cs_fetch_tiles(MODEL, noice && !factors, need_Taux);
    // dynamic LDS: the plane tiles of the initial fetch and the exchange arrays share it
    const int narr = cs_exchange_arrays<MODEL>() + (wave ? cs_flux_arrays<MODEL>() : 0);
    const size_t ex_words = wave ? (size_t)narr * 64 + CS_FACE_WORDS : (size_t)narr * (size_t)P.nlev;
    const size_t tile_words = (size_t)tiles * (size_t)P.nlev;
    const size_t dyn_col = (ex_words > tile_words ? ex_words : tile_words) * sizeof(FT);
    using M = MathFast<FT>;
    constexpr bool has_robust = M::uses_tables && MODEL != MODEL_HEAT; // (as launch_rhs_model)
    const bool robust = has_robust && P.vg_fast_all == 0;
#define LH_CS_GO(F, PC, NI, VG)                                                                                                        \
    do {                                                                                                                               \
        if (wave && cw == 1) {                                                                                                         \
            cpb = wave_stepper_columns<column_stepper_wave_kernel<FT, MODEL, F, PC, M, 1, NI, VG>>(P.ncols, dyn_col, nsteps);                  \
        } else if (wave) {                                                                                                             \
            cpb = wave_stepper_columns<column_stepper_wave_kernel<FT, MODEL, F, PC, M, 2, NI, VG>>(P.ncols, dyn_col, nsteps);                  \
        }                                                                                                                              \
        if (P.cs_cpb > 0 && (unsigned)P.cs_cpb * tpc <= 1024u) cpb = (unsigned)P.cs_cpb;                                               \
        const dim3 g((unsigned)((P.ncols + cpb - 1) / cpb)), b(tpc * cpb);                                                             \
        const unsigned dyn = (unsigned)(cpb * dyn_col);                                                                                \
        if (wave && cw == 1) hipLaunchKernelGGL((column_stepper_wave_kernel<FT, MODEL, F, PC, M, 1, NI, VG>), g, b, dyn, s, P, Y, aux, dt, dt_dev, nsteps, bcv); \
        else if (wave) hipLaunchKernelGGL((column_stepper_wave_kernel<FT, MODEL, F, PC, M, 2, NI, VG>), g, b, dyn, s, P, Y, aux, dt, dt_dev, nsteps, bcv);      \
        else hipLaunchKernelGGL((column_stepper_kernel<FT, MODEL, F, PC, M, NI, VG>), g, b, dyn, s, P, Y, aux, dt, dt_dev, nsteps, bcv);                \
    } while (0)
#define LH_CS(F, PC, NI)                                    \
    do {                                                    \
        if constexpr (has_robust) {                         \
            if (robust) {                                   \
                LH_CS_GO(F, PC, NI, false);                 \
                break;                                      \
            }                                               \
        }                                                   \
        LH_CS_GO(F, PC, NI, true);                          \
    } while (0)
    if (factors) {
        if (percol) LH_CS(true, true, false);
        else LH_CS(true, false, false);
    } else if (noice) {
        if (percol) LH_CS(false, true, true);
        else LH_CS(false, false, true);
    } else {
        if (percol) LH_CS(false, true, false);
        else LH_CS(false, false, false);
    }
#undef LH_CS_GO
#undef LH_CS
}

#define LH_CS_MODEL_ARGS(FT) \
    const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&, FT, const FT*, int64_t, const FT*, bool, bool, bool, hipStream_t
#define LH_RHS_MODEL_ARGS(FT)                                                                        \
    const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&, const Planes<FT>&, const Planes<FT>&, \
        FT, const FT*, int, bool, bool, bool, int, const Tune&, hipStream_t
#ifndef LH_TU_MODEL // the common translation unit only dispatches
extern template void launch_rhs_for_model<double, MODEL_RICHARDS>(LH_RHS_MODEL_ARGS(double));
extern template void launch_rhs_for_model<double, MODEL_HEAT>(LH_RHS_MODEL_ARGS(double));
extern template void launch_rhs_for_model<double, MODEL_COUPLED>(LH_RHS_MODEL_ARGS(double));
extern template void launch_rhs_for_model<float, MODEL_RICHARDS>(LH_RHS_MODEL_ARGS(float));
extern template void launch_rhs_for_model<float, MODEL_HEAT>(LH_RHS_MODEL_ARGS(float));
extern template void launch_rhs_for_model<float, MODEL_COUPLED>(LH_RHS_MODEL_ARGS(float));
extern template void launch_column_stepper_for_model<double, MODEL_RICHARDS>(LH_CS_MODEL_ARGS(double));
extern template void launch_column_stepper_for_model<double, MODEL_HEAT>(LH_CS_MODEL_ARGS(double));
extern template void launch_column_stepper_for_model<double, MODEL_COUPLED>(LH_CS_MODEL_ARGS(double));
extern template void launch_column_stepper_for_model<float, MODEL_RICHARDS>(LH_CS_MODEL_ARGS(float));
extern template void launch_column_stepper_for_model<float, MODEL_HEAT>(LH_CS_MODEL_ARGS(float));
extern template void launch_column_stepper_for_model<float, MODEL_COUPLED>(LH_CS_MODEL_ARGS(float));
#endif
#define LH_INSTANTIATE_MODEL(FT, MODEL)                                    \
    template void launch_rhs_for_model<FT, MODEL>(LH_RHS_MODEL_ARGS(FT)); \
    template void launch_column_stepper_for_model<FT, MODEL>(LH_CS_MODEL_ARGS(FT));

template <typename FT>
void launch_rhs(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                const Planes<FT>& base, const Planes<FT>& out, FT dt, const FT* dt_dev, int mode,
                bool factors, bool percol, bool noice, int math, const Tune& tune, hipStream_t s) {
    switch (P.model) {
        case MODEL_RICHARDS: launch_rhs_for_model<FT, MODEL_RICHARDS>(P, in, aux, base, out, dt, dt_dev, mode, factors, percol, noice, math, tune, s); break;
        case MODEL_HEAT: launch_rhs_for_model<FT, MODEL_HEAT>(P, in, aux, base, out, dt, dt_dev, mode, factors, percol, noice, math, tune, s); break;
        default: launch_rhs_for_model<FT, MODEL_COUPLED>(P, in, aux, base, out, dt, dt_dev, mode, factors, percol, noice, math, tune, s); break;
    }
}

template <typename FT>
void launch_column_stepper(const DevParams<FT>& P, const Planes<FT>& Y, const Planes<FT>& aux, FT dt,
                           const FT* dt_dev, int64_t nsteps, const FT* bcv, bool factors, bool percol,
                           bool noice, hipStream_t s) {
    switch (P.model) {
        case MODEL_RICHARDS: launch_column_stepper_for_model<FT, MODEL_RICHARDS>(P, Y, aux, dt, dt_dev, nsteps, bcv, factors, percol, noice, s); break;
        case MODEL_HEAT: launch_column_stepper_for_model<FT, MODEL_HEAT>(P, Y, aux, dt, dt_dev, nsteps, bcv, factors, percol, noice, s); break;
        default: launch_column_stepper_for_model<FT, MODEL_COUPLED>(P, Y, aux, dt, dt_dev, nsteps, bcv, factors, percol, noice, s); break;
    }
}

template <typename FT>
void launch_diag(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                 const Planes<FT>& out, bool percol, int math, hipStream_t s) {
    dim3 g = grid_for(P.ncols, 256), b(256);
#define LH_DIAG(MODEL, MATH)                                                                           \
    do {                                                                                               \
        if (percol) hipLaunchKernelGGL((diag_kernel<FT, MODEL, true, true, MATH>), g, b, 0, s, P, in, aux, out);  \
        else hipLaunchKernelGGL((diag_kernel<FT, MODEL, true, false, MATH>), g, b, 0, s, P, in, aux, out);        \
    } while (0)
#define LH_DIAG_M(MATH)                                          \
    switch (P.model) {                                           \
        case MODEL_RICHARDS: LH_DIAG(MODEL_RICHARDS, MATH); break; \
        case MODEL_HEAT: LH_DIAG(MODEL_HEAT, MATH); break;         \
        default: LH_DIAG(MODEL_COUPLED, MATH); break;              \
    }
    if (math == MATH_LIBM) {
        LH_DIAG_M(MathLibm<FT>)
    } else {
        LH_DIAG_M(MathFast<FT>)
    }
#undef LH_DIAG_M
#undef LH_DIAG
}

template <typename FT>
void launch_boundary_fluxes(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux, int face, FT* out_e,
                            FT* out_w, bool factors, bool percol, int math, hipStream_t s) {
    dim3 g = grid_for(P.ncols, 256), b(256);
#define LH_BF(MODEL, F, PC, MATH) hipLaunchKernelGGL((boundary_flux_kernel<FT, MODEL, F, PC, MATH>), g, b, 0, s, P, in, aux, face, out_e, out_w)
#define LH_BF_M(MODEL, MATH)                                \
    do {                                                    \
        if (factors) {                                      \
            if (percol) LH_BF(MODEL, true, true, MATH);     \
            else LH_BF(MODEL, true, false, MATH);           \
        } else {                                            \
            if (percol) LH_BF(MODEL, false, true, MATH);    \
            else LH_BF(MODEL, false, false, MATH);          \
        }                                                   \
    } while (0)
#define LH_BF_MODEL(MATH)                                              \
    switch (P.model) {                                                 \
        case MODEL_RICHARDS: LH_BF_M(MODEL_RICHARDS, MATH); break;     \
        case MODEL_HEAT: LH_BF_M(MODEL_HEAT, MATH); break;             \
        default: LH_BF_M(MODEL_COUPLED, MATH); break;                  \
    }
    if (math == MATH_LIBM) {
        LH_BF_MODEL(MathLibm<FT>)
    } else {
        LH_BF_MODEL(MathFast<FT>)
    }
#undef LH_BF_MODEL
#undef LH_BF_M
#undef LH_BF
}

template <typename FT>
void launch_stable_dt(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                      FT courant, void* out_ft, bool percol, hipStream_t s) {
    using U = typename Bits<FT>::type;
    U* out = reinterpret_cast<U*>(out_ft);
    hipLaunchKernelGGL((init_bits_kernel<FT>), dim3(1), dim3(1), 0, s, out);
    dim3 g = grid_for(P.ncols, 256), b(256);
#define LH_SDT(MODEL)                                                                                       \
    do {                                                                                                    \
        if (percol) hipLaunchKernelGGL((stable_dt_kernel<FT, MODEL, true, MathFast<FT>>), g, b, 0, s, P, in, aux, courant, out);  \
        else hipLaunchKernelGGL((stable_dt_kernel<FT, MODEL, false, MathFast<FT>>), g, b, 0, s, P, in, aux, courant, out);        \
    } while (0)
    switch (P.model) {
        case MODEL_RICHARDS: LH_SDT(MODEL_RICHARDS); break;
        case MODEL_HEAT: LH_SDT(MODEL_HEAT); break;
        default: LH_SDT(MODEL_COUPLED); break;
    }
#undef LH_SDT
}

template <typename FT>
void launch_strided_copy(FT* plane, int64_t stride, FT* user, int64_t ls, int64_t cs,
                         int64_t ncols, int nlev, bool to_plane, hipStream_t s) {
    dim3 g((unsigned)((ncols + 63) / 64), (unsigned)((nlev + 63) / 64)), b(256);
    if (to_plane)
        hipLaunchKernelGGL((strided_copy_kernel<FT, true>), g, b, 0, s, plane, stride, user, ls, cs, ncols, nlev);
    else
        hipLaunchKernelGGL((strided_copy_kernel<FT, false>), g, b, 0, s, plane, stride, user, ls, cs, ncols, nlev);
}

template <typename FT>
void launch_stream_probe(int64_t ncols, int64_t stride, int nlev, int xcd_remap, const Planes<FT>& in, int nr,
                         const Planes<FT>& out, int nw, bool nt, hipStream_t s) {
    using CFG = typename DefaultCfg<FT>::type;
    constexpr int CPL = CFG::CPL, PF = CFG::PF;
    const int64_t lanes = (ncols + CPL - 1) / CPL;
    dim3 g = grid_for(lanes, 256), b(256);
    if (nt) hipLaunchKernelGGL((stream_probe_kernel<FT, CPL, PF, true>), g, b, 0, s, ncols, stride, nlev, xcd_remap, in, nr, out, nw);
    else hipLaunchKernelGGL((stream_probe_kernel<FT, CPL, PF, false>), g, b, 0, s, ncols, stride, nlev, xcd_remap, in, nr, out, nw);
}

template <typename FT>
void launch_dt_prepare(FT* dt, FT dt_max, FT* elapsed, uint32_t* status, hipStream_t s) {
    hipLaunchKernelGGL((dt_prepare_kernel<FT>), dim3(1), dim3(1), 0, s, dt, dt_max, elapsed, status);
}

template <typename FT>
void launch_fill(FT* p, int64_t n, FT v, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((fill_kernel<FT>), dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
}

template <typename FT>
void launch_broadcast_profile(FT* plane, const FT* prof, int64_t ncols, int64_t stride, int nlev, hipStream_t s) {
    dim3 g((unsigned)((ncols + 255) / 256), (unsigned)nlev);
    hipLaunchKernelGGL((broadcast_profile_kernel<FT>), g, dim3(256), 0, s, plane, prof, ncols, stride);
}

template <typename FT>
void launch_convert(FT* dst, const double* src, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((convert_kernel<FT>), grid_for(n, 256), dim3(256), 0, s, dst, src, n);
}

// explicit instantiation for one working type per translation unit
#define LH_INSTANTIATE(FT)                                                                            \
    template void launch_rhs<FT>(const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&,          \
                                 const Planes<FT>&, const Planes<FT>&, FT, const FT*, int, bool, bool, \
                                 bool, int, const Tune&, hipStream_t);                                                      \
    template void launch_column_stepper<FT>(const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&, \
                                            FT, const FT*, int64_t, const FT*, bool, bool, bool, hipStream_t); \
    template void launch_diag<FT>(const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&,         \
                                  const Planes<FT>&, bool, int, hipStream_t);                         \
    template void launch_stable_dt<FT>(const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&,    \
                                       FT, void*, bool, hipStream_t);                                 \
    template void launch_boundary_fluxes<FT>(const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&, int, FT*, FT*, \
                                             bool, bool, int, hipStream_t);                           \
    template void launch_strided_copy<FT>(FT*, int64_t, FT*, int64_t, int64_t, int64_t, int, bool,    \
                                          hipStream_t);                                               \
    template void launch_stream_probe<FT>(int64_t, int64_t, int, int, const Planes<FT>&, int,         \
                                          const Planes<FT>&, int, bool, hipStream_t);                 \
    template void launch_atmos_flux<FT>(const DevParams<FT>&, const AtmosParams<FT>&, int64_t, bool,  \
                                        bool, const FT*, const FT*, const FT*, FT*, FT*, hipStream_t); \
    template void launch_fill<FT>(FT*, int64_t, FT, hipStream_t);                                     \
    template void launch_broadcast_profile<FT>(FT*, const FT*, int64_t, int64_t, int, hipStream_t);   \
    template void launch_dt_prepare<FT>(FT*, FT, FT*, uint32_t*, hipStream_t);                                   \
    template void launch_convert<FT>(FT*, const double*, int64_t, hipStream_t);

} // namespace lh

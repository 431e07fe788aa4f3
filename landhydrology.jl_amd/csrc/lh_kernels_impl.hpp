// lh_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched soil-column
// tendency path.  No MFMA: this is a bandwidth/VALU-bound vertical stencil.
//
// Data layout: every variable is a plane [nlev][stride] with the
// column index fastest, so a wavefront reads 64*CPL consecutive columns of one
// level in one fully coalesced instruction.  One lane owns CPL whole columns
// and marches bottom -> top, carrying K, h (T, kappa, rho_e_l K) of the previous
// cell and the previous face flux in registers: each face flux is computed once
// and differenced, which keeps the discrete mass/energy conservation the
// reference's equilibrium tests rely on (coupled.jl:117, richards_equation.jl:94).
#pragma once
#include "lh_closures.hpp"
#include "lh_launch.hpp"

namespace lh {

// ----------------------------------------------------------------- helpers

// native clang vectors (the nontemporal builtins do not take HIP_vector_type)
template <typename FT, int N> struct Vec;
template <> struct Vec<double, 1> { using type = double; };
template <> struct Vec<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec<float, 1> { using type = float; };
template <> struct Vec<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };

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

// launch shape of the column kernel: columns per lane, levels kept in flight
// ahead of the one being computed, nontemporal global access
template <int CPL_, int PF_, bool NT_>
struct KCfg {
    static constexpr int CPL = CPL_, PF = PF_;
    static constexpr bool NT = NT_;
};

// Stage the log2/exp2 tables of MathFast<double> in LDS (5 KiB per workgroup);
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
// MODE 0: write the tendency dY.
// MODE 1..3: fused SSPRK33 stage s (OrdinaryDiffEq SSPRK33, Shu-Osher form):
//   1: U1 = Y + dt f(Y)            (in = Y,  out = U1)
//   2: U1 = (3 Y + U1 + dt f(U1))/4 (in = U1, base = Y, out = U1)
//   3: Y  = (Y + 2 U1 + 2 dt f(U1))/3 (in = U1, base = Y, out = Y)
// with in/base/out planes handed over by the launcher.
template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename CFG, typename M, int MODE>
__global__ void __launch_bounds__(256)
rhs_kernel(const DevParams<FT> P, const Planes<FT> IN, const Planes<FT> AUX, const Planes<FT> BASE,
           const Planes<FT> OUT, const FT dt) {
    constexpr bool WATER = (MODEL != MODEL_HEAT);
    constexpr bool HEAT = (MODEL != MODEL_RICHARDS);
    constexpr int CPL = CFG::CPL, PF = CFG::PF;
    constexpr bool NT = CFG::NT;
    __shared__ double s_tab[M::uses_tables ? MATH_TAB_DOUBLES : 1];
    const M mm(stage_math_tables<M>(P.math_tab, s_tab));
    const int64_t col0 = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) * CPL;
    if (col0 >= P.ncols) return;
    const int64_t stride = P.stride;
    const int n = P.nlev;

    // inputs: HEAT reads the prescribed water fields from Ya (right_hand_side.jl:200-201)
    const FT* p_vl = (MODEL == MODEL_HEAT ? AUX.v[0] : IN.v[0]) + col0;
    const FT* p_ti = (MODEL == MODEL_HEAT ? AUX.v[1] : IN.v[1]) + col0;
    const FT* p_re = HEAT ? IN.v[2] + col0 : nullptr;
    const bool need_Taux = (MODEL == MODEL_RICHARDS) && FACTORS && P.viscosity_kind;
    const FT* p_Ta = need_Taux ? AUX.v[3] + col0 : nullptr;

    ColC<FT> c[CPL];
    int64_t colj[CPL]; // column index clamped into [0, ncols): pad lanes reuse the last column
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        colj[j] = col0 + j < P.ncols ? col0 + j : P.ncols - 1;
        c[j] = make_colc<FT, M>(P, colj[j], PERCOL);
    }

    FT vl[CPL], ti[CPL], re[CPL], Ta[CPL];       // current cell inputs
    FT vl_n[PF][CPL], ti_n[PF][CPL], re_n[PF][CPL], Ta_n[PF][CPL]; // levels in flight
    FT vl_p[CPL], ti_p[CPL], re_p[CPL];          // previous cell inputs (fused stages)
    FT K_p[CPL], h_p[CPL], psi_p[CPL], T_p[CPL], kap_p[CPL], E_p[CPL];
    FT Fw_lo[CPL], Fe_lo[CPL];
    bool bad = false;

#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        K_p[j] = h_p[j] = psi_p[j] = T_p[j] = kap_p[j] = E_p[j] = Fw_lo[j] = Fe_lo[j] = FT(0);
        vl_p[j] = ti_p[j] = re_p[j] = FT(0);
        vl[j] = ti[j] = re[j] = Ta[j] = FT(0);
    }
    auto fetch = [&](int lev, int slot) {
        const int64_t o = int64_t(lev) * stride;
        vload<FT, CPL, NT>(p_vl + o, vl_n[slot]);
        vload<FT, CPL, NT>(p_ti + o, ti_n[slot]);
        if (HEAT) vload<FT, CPL, NT>(p_re + o, re_n[slot]);
        if (need_Taux) vload<FT, CPL, NT>(p_Ta + o, Ta_n[slot]);
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            vl_n[k][j] = ti_n[k][j] = re_n[k][j] = FT(0);
            Ta_n[k][j] = FT(288); // PrescribedTemperatureModel default (models.jl:53)
        }
        if (k < n) fetch(k, k);
    }

    // emit the result for cell `lev` given its two face fluxes
    // (u_vl, u_ti, u_re) are the cell's own input values, kept in registers
    auto emit = [&](int lev, const FT (&Fw_hi)[CPL], const FT (&Fe_hi)[CPL], const FT (&u_vl)[CPL],
                    const FT (&u_ti)[CPL], const FT (&u_re)[CPL]) {
        const int64_t o = int64_t(lev) * stride + col0;
        FT dvl[CPL], dre[CPL], zero[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            dvl[j] = WATER ? -((Fw_hi[j] - Fw_lo[j]) * P.inv_dz) : FT(0);
            dre[j] = HEAT ? -((Fe_hi[j] - Fe_lo[j]) * P.inv_dz) : FT(0);
            zero[j] = FT(0);
            if (col0 + j < P.ncols) bad = bad || !finite(dvl[j]) || !finite(dre[j]);
        }
        if (MODE == 0) {
            if (WATER) {
                vstore<FT, CPL, NT>(OUT.v[0] + o, dvl);
                vstore<FT, CPL, NT>(OUT.v[1] + o, zero); // d theta_i = 0 (:182, :359)
            }
            if (HEAT) vstore<FT, CPL, NT>(OUT.v[2] + o, dre);
        } else {
            // fused SSPRK33 stage; theta_i has a zero tendency and is carried unchanged
            auto stage = [&](int var, const FT (&u)[CPL], const FT (&k)[CPL]) {
                FT b[CPL], r[CPL];
                if (MODE != 1) vload<FT, CPL, NT>(BASE.v[var] + o, b);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    if (MODE == 1)
                        r[j] = u[j] + dt * k[j];
                    else if (MODE == 2)
                        r[j] = (FT(3) * b[j] + u[j] + dt * k[j]) * FT(0.25);
                    else
                        r[j] = (b[j] + FT(2) * u[j] + FT(2) * dt * k[j]) * FT(1.0 / 3.0);
                }
                vstore<FT, CPL, false>(OUT.v[var] + o, r);
            };
            if (WATER) {
                stage(0, u_vl, dvl);
                stage(1, u_ti, zero);
            }
            if (HEAT) stage(2, u_re, dre);
        }
    };

    for (int i0 = 0; i0 < n; i0 += PF) {
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        const int i = i0 + k;
        if (PF > 1 && i >= n) break;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            vl[j] = vl_n[k][j];
            ti[j] = ti_n[k][j];
            re[j] = re_n[k][j];
            Ta[j] = Ta_n[k][j];
        }
        if (i + PF < n) fetch(i + PF, k); // keep PF levels in flight ahead of the compute
        const FT z = P.zc[i];
        FT K[CPL], h[CPL], psi[CPL], T[CPL], kap[CPL], E[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            T[j] = Ta[j];
            kap[j] = FT(0);
            K[j] = psi[j] = h[j] = E[j] = FT(0);
            if (HEAT) {
                FT rcs;
                T[j] = temperature_closure<FT>(P, c[j], vl[j], ti[j], re[j], rcs);
                kap[j] = kappa_closure<FT, M>(mm, P, c[j], vl[j], ti[j]);
            }
            if (WATER) {
                water_closures<FT, M, FACTORS>(mm, P, c[j], vl[j], ti[j], T[j], K[j], psi[j]);
                h[j] = psi[j] + z;
                if (HEAT) E[j] = (P.rhocp_l * (T[j] - P.T_ref)) * K[j]; // rho_e_int_l * K (:364)
            }
        }
        if (i == 0) {
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                boundary_fluxes<FT, M, MODEL, FACTORS>(mm, P, c[j], FACE_BOTTOM, colj[j], vl[j], ti[j],
                                                       T[j], K[j], psi[j], Fe_lo[j], Fw_lo[j]);
        } else {
            FT Fw[CPL], Fe[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                Fw[j] = Fe[j] = FT(0);
                FT gh = FT(0);
                if (WATER) {
                    gh = (h[j] - h_p[j]) * P.inv_dz;
                    Fw[j] = -((K_p[j] + K[j]) * FT(0.5)) * gh;
                }
                if (HEAT) {
                    FT gT = (T[j] - T_p[j]) * P.inv_dz;
                    Fe[j] = -((kap_p[j] + kap[j]) * FT(0.5)) * gT;
                    if (WATER) Fe[j] = Fe[j] - ((E_p[j] + E[j]) * FT(0.5)) * gh;
                }
            }
            emit(i - 1, Fw, Fe, vl_p, ti_p, re_p);
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                Fw_lo[j] = Fw[j];
                Fe_lo[j] = Fe[j];
            }
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            vl_p[j] = vl[j];
            ti_p[j] = ti[j];
            re_p[j] = re[j];
            K_p[j] = K[j];
            h_p[j] = h[j];
            psi_p[j] = psi[j];
            T_p[j] = T[j];
            kap_p[j] = kap[j];
            E_p[j] = E[j];
        }
      }
    }
    {
        FT Fw[CPL], Fe[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            boundary_fluxes<FT, M, MODEL, FACTORS>(mm, P, c[j], FACE_TOP, colj[j], vl[j], ti[j], T_p[j],
                                                   K_p[j], psi_p[j], Fe[j], Fw[j]);
        emit(n - 1, Fw, Fe, vl_p, ti_p, re_p);
    }
    if (bad) atomicOr(P.status, 1u);
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
    const ColC<FT> c = make_colc<FT, M>(P, col, PERCOL);
    for (int i = 0; i < P.nlev; ++i) {
        const int64_t o = int64_t(i) * P.stride;
        FT vl = p_vl[o], ti = p_ti[o];
        FT T = need_Taux ? AUX.v[3][o + col] : FT(288), kap = FT(0), K = FT(0), psi = FT(0);
        if (HEAT) {
            FT rcs;
            T = temperature_closure<FT>(P, c, vl, ti, IN.v[2][o + col], rcs);
            kap = kappa_closure<FT, M>(mm, P, c, vl, ti);
        }
        if (WATER) water_closures<FT, M, FACTORS>(mm, P, c, vl, ti, T, K, psi);
        OUT.v[0][o + col] = K;
        OUT.v[1][o + col] = psi;
        OUT.v[2][o + col] = kap;
        OUT.v[3][o + col] = T;
    }
}

// ---------------------------------------------------------- stable dt
// min over cells of courant*dz^2/max(K dpsi/dvl, kappa/rho_c_s) (build-defined;
// the reference steps with a fixed user dt, simulation.jl:34-70).  Positive IEEE
// values order like their bit patterns, so the global min is an integer atomicMin.
template <typename FT> struct Bits;
template <> struct Bits<double> { using type = unsigned long long; };
template <> struct Bits<float> { using type = unsigned int; };

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
        const ColC<FT> c = make_colc<FT, M>(P, col, PERCOL);
        const FT cdz2 = courant * P.dz * P.dz;
        const int n = P.nlev;
        FT K_p = FT(0), dpsi_p = FT(0), kap_p = FT(0), rcs_p = FT(1);
        for (int i = 0; i < n; ++i) {
            const int64_t o = int64_t(i) * P.stride;
            FT vl = p_vl[o], ti = p_ti[o];
            FT K = FT(0), dpsi = FT(0), kap = FT(0), rcs = FT(1);
            FT Tc = FT(288);
            if (MODEL == MODEL_RICHARDS && P.viscosity_kind) Tc = AUX.v[3][o + col];
            if (HEAT) Tc = temperature_closure<FT>(P, c, vl, ti, IN.v[2][o + col], rcs);
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
                    if (WATER && kh == BC_DIRICHLET) {
                        FT K_f, psi_f;
                        water_closures<FT, M, true, false>(mm, P, c, vl_f, ti, FT(288), K_f, psi_f);
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
        atomicMin(out_bits, b);
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

template <typename FT>
__global__ void __launch_bounds__(256) fill_kernel(FT* p, int64_t n, FT v) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t step = int64_t(gridDim.x) * blockDim.x;
    for (; i < n; i += step) p[i] = v;
}

template <typename FT>
__global__ void __launch_bounds__(256) convert_kernel(FT* dst, const double* src, int64_t n) {
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = FT(src[i]);
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

template <typename FT, int MODEL, bool FACTORS, bool PERCOL, typename CFG, typename M>
static void launch_rhs_mode(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                            const Planes<FT>& base, const Planes<FT>& out, FT dt, int mode,
                            int block, hipStream_t s) {
    const int64_t lanes = (P.ncols + CFG::CPL - 1) / CFG::CPL;
    dim3 g = grid_for(lanes, block), b(block);
    if (!M::is_production) { // MathLibm: tendency only
        hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 0>), g, b, 0, s, P, in, aux, base, out, dt);
        return;
    }
    switch (mode) {
        case 0: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 0>), g, b, 0, s, P, in, aux, base, out, dt); break;
        case 1: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 1>), g, b, 0, s, P, in, aux, base, out, dt); break;
        case 2: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 2>), g, b, 0, s, P, in, aux, base, out, dt); break;
        default: hipLaunchKernelGGL((rhs_kernel<FT, MODEL, FACTORS, PERCOL, CFG, M, 3>), g, b, 0, s, P, in, aux, base, out, dt); break;
    }
}

template <typename FT, int MODEL, typename M>
static void launch_rhs_model(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                             const Planes<FT>& base, const Planes<FT>& out, FT dt, int mode,
                             bool factors, bool percol, const Tune& tune, hipStream_t s) {
    using CFG = typename DefaultCfg<FT>::type;
    const int block = tune.block > 0 ? tune.block : 256;
#ifdef LH_TUNING_VARIANTS
    // tuning builds: alternative launch shapes for the plain Richards tendency
    if (MODEL == MODEL_RICHARDS && !factors && !percol && mode == 0 && M::is_production &&
        (tune.cpl > 0 || tune.pf > 0 || tune.nt >= 0)) {
        const int cpl = tune.cpl > 0 ? tune.cpl : CFG::CPL, pf = tune.pf > 0 ? tune.pf : CFG::PF;
        const bool nt = tune.nt >= 0 ? tune.nt != 0 : CFG::NT;
#define LH_TRY(C, F, N)                                                                              \
    if (cpl == C && pf == F && nt == N) {                                                            \
        launch_rhs_mode<FT, MODEL_RICHARDS, false, false, KCfg<C, F, N>, M>(P, in, aux, base, out, dt, 0, block, s); \
        return;                                                                                      \
    }
        LH_TRY(1, 1, false) LH_TRY(1, 2, false) LH_TRY(1, 3, false) LH_TRY(1, 4, false)
        LH_TRY(1, 1, true) LH_TRY(1, 2, true) LH_TRY(1, 4, true)
        LH_TRY(2, 1, false) LH_TRY(2, 2, false) LH_TRY(2, 2, true) LH_TRY(2, 1, true)
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
#define LH_GO(F, PC)                                                                                  \
    do {                                                                                              \
        if (nt) launch_rhs_mode<FT, MODEL, F, PC, CFGN, M>(P, in, aux, base, out, dt, mode, block, s); \
        else launch_rhs_mode<FT, MODEL, F, PC, CFGP, M>(P, in, aux, base, out, dt, mode, block, s);   \
    } while (0)
    if (factors) {
        if (percol) LH_GO(true, true);
        else LH_GO(true, false);
    } else {
        if (percol) LH_GO(false, true);
        else LH_GO(false, false);
    }
#undef LH_GO
}

template <typename FT>
void launch_rhs(const DevParams<FT>& P, const Planes<FT>& in, const Planes<FT>& aux,
                const Planes<FT>& base, const Planes<FT>& out, FT dt, int mode, bool factors,
                bool percol, int math, const Tune& tune, hipStream_t s) {
#define LH_DISPATCH_MODEL(MATH)                                                                 \
    switch (P.model) {                                                                          \
        case MODEL_RICHARDS: launch_rhs_model<FT, MODEL_RICHARDS, MATH>(P, in, aux, base, out, dt, mode, factors, percol, tune, s); break; \
        case MODEL_HEAT: launch_rhs_model<FT, MODEL_HEAT, MATH>(P, in, aux, base, out, dt, mode, factors, percol, tune, s); break;         \
        default: launch_rhs_model<FT, MODEL_COUPLED, MATH>(P, in, aux, base, out, dt, mode, factors, percol, tune, s); break;              \
    }
    // MathLibm is a parity-debugging policy for the tendency itself (mode 0);
    // the fused SSPRK33 stages always run the production math.
    if (math == MATH_LIBM && mode == 0) {
        LH_DISPATCH_MODEL(MathLibm<FT>)
    } else {
        LH_DISPATCH_MODEL(MathFast<FT>)
    }
#undef LH_DISPATCH_MODEL
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
void launch_fill(FT* p, int64_t n, FT v, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((fill_kernel<FT>), dim3((unsigned)blocks), dim3(256), 0, s, p, n, v);
}

template <typename FT>
void launch_convert(FT* dst, const double* src, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL((convert_kernel<FT>), grid_for(n, 256), dim3(256), 0, s, dst, src, n);
}

// explicit instantiation for one working type per translation unit
#define LH_INSTANTIATE(FT)                                                                            \
    template void launch_rhs<FT>(const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&,          \
                                 const Planes<FT>&, const Planes<FT>&, FT, int, bool, bool, int,      \
                                 const Tune&, hipStream_t);                                                      \
    template void launch_diag<FT>(const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&,         \
                                  const Planes<FT>&, bool, int, hipStream_t);                         \
    template void launch_stable_dt<FT>(const DevParams<FT>&, const Planes<FT>&, const Planes<FT>&,    \
                                       FT, void*, bool, hipStream_t);                                 \
    template void launch_strided_copy<FT>(FT*, int64_t, FT*, int64_t, int64_t, int64_t, int, bool,    \
                                          hipStream_t);                                               \
    template void launch_fill<FT>(FT*, int64_t, FT, hipStream_t);                                     \
    template void launch_convert<FT>(FT*, const double*, int64_t, hipStream_t);

} // namespace lh

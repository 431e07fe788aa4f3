/*
 * lh_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see lh_oracle.h).
 * Instantiates lh_oracle_impl.inc for Float64 and Float32.
 */
#include "lh_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LHO_CAT_(a, b) a##b
#define LHO_CAT(a, b) LHO_CAT_(a, b)

/* ---- Float64 ---- */
#define FT double
#define SFX(name) LHO_CAT(name, _f64)
#define FT_EPS DBL_EPSILON /* eps(Float64) */
#define FT_POW pow
#define FT_EXP exp
#define FT_SQRT sqrt
#define FT_FABS fabs
#define FT_LOG log
#define FT_ATAN atan
#include "lh_oracle_impl.inc"
#undef FT
#undef SFX
#undef FT_EPS
#undef FT_POW
#undef FT_EXP
#undef FT_SQRT
#undef FT_FABS
#undef FT_LOG
#undef FT_ATAN

/* ---- Float32 ---- */
#define FT float
#define SFX(name) LHO_CAT(name, _f32)
#define FT_EPS FLT_EPSILON /* eps(Float32) */
#define FT_POW powf
#define FT_EXP expf
#define FT_SQRT sqrtf
#define FT_FABS fabsf
#define FT_LOG logf
#define FT_ATAN atanf
#include "lh_oracle_impl.inc"
#undef FT
#undef SFX
#undef FT_EPS
#undef FT_POW
#undef FT_EXP
#undef FT_SQRT
#undef FT_FABS
#undef FT_LOG
#undef FT_ATAN

int lho_openmp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

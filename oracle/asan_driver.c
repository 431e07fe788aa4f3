#include "lh_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
/* ASan/UBSan smoke of the oracle: all three models, both types, per-column
 * overrides, SSPRK33 with stage values. */
int main(void) {
    lho_model m; memset(&m, 0, sizeof m);
    m.nlev = 37; m.zmin = -3.0; m.zmax = -0.5;
    m.earth = (lho_earth_params){1000, 916.7, 4181, 2100, 273.16, 333600, 0.024};
    m.soil = (lho_soil_params){0.5, 1e-3, 0, 0, 0.92, 963000, 5.0, 2700, 1.5, 2.9, 0.24, 18.1, 0.053};
    m.vg = (lho_vg_params){2.0, 2.6, 0.0, 1.2e-7};
    m.cf = (lho_cond_factors){1, 1, 2.64e-2, 288.0, 7.0};
    int N = 33, n = m.nlev;
    double *vl = malloc(sizeof(double)*N*n), *ti = malloc(sizeof(double)*N*n), *re = malloc(sizeof(double)*N*n);
    double *d1 = malloc(sizeof(double)*N*n), *d2 = malloc(sizeof(double)*N*n), *d3 = malloc(sizeof(double)*N*n);
    float *vlf = malloc(sizeof(float)*N*n), *tif = malloc(sizeof(float)*N*n), *ref = malloc(sizeof(float)*N*n);
    float *f1 = malloc(sizeof(float)*N*n), *f2 = malloc(sizeof(float)*N*n), *f3 = malloc(sizeof(float)*N*n);
    double *pcn = malloc(sizeof(double)*N), *pcb = malloc(sizeof(double)*N);
    for (int i = 0; i < N*n; ++i) { vl[i] = 0.1 + 0.4*((i*37)%101)/101.0; ti[i] = (i%7==0)?0.05:0.0; re[i] = 2.0e6*(280.0 + (i%13) - 273.16);
        vlf[i] = (float)vl[i]; tif[i] = (float)ti[i]; ref[i] = (float)re[i]; }
    for (int c = 0; c < N; ++c) { pcn[c] = 1.4 + 0.05*c; pcb[c] = -1e-8*c; }
    lho_percol pc; memset(&pc, 0, sizeof pc); pc.vg_n = pcn; pc.bc_value[1][1] = pcb;
    int rc = 0;
    for (int model = 0; model < 3; ++model) {
        m.model = model;
        for (int f = 0; f < 2; ++f) for (int k = 0; k < 2; ++k) { m.bc[f][k].kind = LHO_BC_FLUX; m.bc[f][k].value = 0.0; }
        if (model != LHO_MODEL_HEAT) { m.bc[1][1].kind = LHO_BC_DIRICHLET; m.bc[1][1].value = 0.45; m.bc[0][1].kind = LHO_BC_FREE_DRAINAGE; }
        if (model != LHO_MODEL_RICHARDS) { m.bc[1][0].kind = LHO_BC_DIRICHLET; m.bc[1][0].value = 279.0; }
        rc |= lho_rhs_f64(&m, &pc, N, vl, ti, re, re, d1, d2, d3, 1, n, 2);
        rc |= lho_rhs_f32(&m, &pc, N, vlf, tif, ref, ref, f1, f2, f3, 1, n, 2);
        rc |= lho_rhs_f64(&m, NULL, N, vl, ti, re, re, d1, d2, d3, N, 1, 1);   /* column-fastest strides */
        double dt = lho_stable_dt_f64(&m, &pc, N, vl, ti, re, re, 1, n, 0.1);
        double bcv[2*3*4]; for (int i = 0; i < 24; ++i) bcv[i] = (i%4==3)?0.45:((i%4==2)?279.0:0.0);
        rc |= lho_ssprk33_f64(&m, &pc, N, vl, ti, re, re, 1, n, 0.0, isfinite(dt) ? dt : 1.0, 2, bcv, 2);
        printf("model %d rc %d dt %g\n", model, rc, dt);
    }
    double zc[64], zf[65]; lho_grid_f64(-1.28, 0, 64, zc, zf);
    return rc != 0;
}

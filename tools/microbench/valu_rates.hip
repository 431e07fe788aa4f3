// valu_rates.hip -- measure issue cost (cycles per wave64 instruction per SIMD) of the
// f64/f32 VALU instructions the soil closures use, on the device it runs on.
// Build: hipcc -O2 --offload-arch=gfx950 valu_rates.hip -o valu_rates ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP 64   // instructions per loop body per chain set
#define ITERS 256

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* cyc, double seed) {
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001, c = 1e-9;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#define R8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#define R8F(S) S(f0) S(f1) S(f2) S(f3) S(f4) S(f5) S(f6) S(f7)
#define X8(B) B B B B B B B B
        if (OP == 0) {
#define S(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            X8(R8(S))
#undef S
        } else if (OP == 1) {
#define S(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
            X8(R8(S))
#undef S
        } else if (OP == 2) {
#define S(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
            X8(R8(S))
#undef S
        } else if (OP == 3) {
#define S(x) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(i0));
            X8(R8(S))
#undef S
        } else if (OP == 4) {
#define S(x) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x));
            X8(R8(S))
#undef S
        } else if (OP == 5) {
#define S(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
            X8(R8(S))
#undef S
        } else if (OP == 6) {
#define S(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
            X8(R8(S))
#undef S
        } else if (OP == 7) {
#define S(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
            X8(R8(S))
#undef S
        } else if (OP == 8) {
#define S(x) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i1) : "v"(x));
            X8(R8(S))
#undef S
        } else if (OP == 9) {
#define S(x) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x) : "v"(i0));
            X8(R8(S))
#undef S
        } else if (OP == 10) {
#define S(x) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i1) : "v"(x));
            X8(R8(S))
#undef S
        } else if (OP == 11) {
#define S(x) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(b));
            X8(R8(S))
#undef S
        } else if (OP == 12) {
#define S(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i1) : "v"(i0));
            X8(R8(S))
#undef S
        } else if (OP == 13) {
#define S(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(f1), "v"(f2));
            X8(S(f0) S(f3) S(f4) S(f5) S(f6) S(f7) S(f0) S(f3))
#undef S
        } else if (OP == 14) {
#define S(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            X8(R8F(S))
#undef S
        } else if (OP == 15) {
#define S(x) asm volatile("v_log_f32 %0, %0" : "+v"(x));
            X8(R8F(S))
#undef S
        } else if (OP == 16) {
#define S(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
            X8(R8(S))
#undef S
        } else if (OP == 17) {
#define S(x) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(x), "v"(b) : "vcc");
            X8(R8(S))
#undef S
        } else if (OP == 18) {
#define S(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(i0));
            X8(R8(S))
#undef S
        } else if (OP == 19) {
#define S(x) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
            X8(R8(S))
#undef S
        } else if (OP == 20) {
#define S(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
            X8(R8F(S))
#undef S
        } else if (OP == 21) {
#define S(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
            X8(R8F(S))
#undef S
        } else if (OP == 22) {
#define S(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            X8(R8(S))
#undef S
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0 + i1 + i2 + i3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int blocks_per_cu) {
    int ncu = 256;
    int blocks = ncu * blocks_per_cu;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, cyc, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, cyc, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    double ninst = (double)ITERS * REP;            // per wave
    int waves_per_simd = blocks_per_cu;            // 256 threads = 4 waves = 1 per SIMD per block
    // s_memtime ticks at 100 MHz-ish constant clock? report both: ticks/inst and wall-derived
    double wall_cycles_24 = ms * 1e-3 * 2.4e9;
    printf("%-22s waves/SIMD=%d  s_memtime ticks/inst/wave=%.3f  -> per SIMD %.3f | wall: %.3f cyc@2.4GHz per inst per SIMD (ms=%.4f)\n",
           name, waves_per_simd, avg / ninst, avg / ninst / waves_per_simd, wall_cycles_24 / (ninst * waves_per_simd), ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int bpc : {1, 4}) {
        run<0>("v_fma_f64", bpc); run<1>("v_mul_f64", bpc); run<2>("v_add_f64", bpc); run<3>("v_ldexp_f64", bpc);
        run<4>("v_frexp_mant_f64", bpc); run<5>("v_rndne_f64", bpc); run<6>("v_rcp_f64", bpc); run<7>("v_rsq_f64", bpc);
        run<8>("v_cvt_i32_f64", bpc); run<9>("v_cvt_f64_i32", bpc); run<10>("v_frexp_exp_i32_f64", bpc);
        run<11>("v_mov_b64", bpc); run<12>("v_cndmask_b32", bpc); run<13>("v_fma_f32", bpc); run<14>("v_exp_f32", bpc);
        run<15>("v_log_f32", bpc); run<16>("v_max_f64", bpc); run<17>("v_cmp_lt_f64", bpc); run<18>("v_add_u32", bpc);
        run<19>("v_sqrt_f64", bpc); run<20>("v_rcp_f32", bpc); run<21>("v_sqrt_f32", bpc); run<22>("v_pk_fma_f32", bpc);
    }
    return 0;
}

// issue_rates.hip -- cycles per wave64 VALU instruction per SIMD at FULL occupancy (8 waves/SIMD),
// every instruction in an independent chain (round 1's valu_rates.hip measured one wave per SIMD
// and several of its chains were serially dependent, so its 32-bit figures were latencies).
// One workgroup of 1024 threads per CU slot, 2 per CU: each SIMD holds 8 waves running the same
// straight-line body; the clock is read with s_memtime around the loop by every wave and the
// per-SIMD cost is (ticks of one wave) / (instructions per wave x 8 waves).
// Build: hipcc -O2 --offload-arch=gfx950 issue_rates.hip -o issue_rates ; run: ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 128

// 16 independent accumulators per kind keep every instruction independent of the previous 15
#define D16(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7) S(a8) S(a9) S(a10) S(a11) S(a12) S(a13) S(a14) S(a15)
#define I16(S) S(i0) S(i1) S(i2) S(i3) S(i4) S(i5) S(i6) S(i7) S(i8) S(i9) S(i10) S(i11) S(i12) S(i13) S(i14) S(i15)

template <int OP>
__global__ void __launch_bounds__(1024, 2) k(double* out, unsigned long long* cyc, double seed, int iseed) {
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
           a7 = a0 + 7, a8 = a0 + 8, a9 = a0 + 9, a10 = a0 + 10, a11 = a0 + 11, a12 = a0 + 12, a13 = a0 + 13,
           a14 = a0 + 14, a15 = a0 + 15;
    int i0 = threadIdx.x + iseed, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7,
        i8 = i0 + 8, i9 = i0 + 9, i10 = i0 + 10, i11 = i0 + 11, i12 = i0 + 12, i13 = i0 + 13, i14 = i0 + 14, i15 = i0 + 15;
    double b = 1.0000001, c = 1e-9;
    int m = 0x7ff0;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(m));
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) {
#define S(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            D16(S) D16(S)
#undef S
        } else if (OP == 1) {
#define S(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(m));
            I16(S) I16(S)
#undef S
        } else if (OP == 2) {
#define S(x) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(m));
            I16(S) I16(S)
#undef S
        } else if (OP == 3) { // half f64 fma, half 32-bit and: do the costs add?
#define S(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define T(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(m));
            D16(S) I16(T)
#undef S
#undef T
        } else if (OP == 4) {
#define S(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m) : );
            I16(S) I16(S)
#undef S
        } else if (OP == 5) {
#define S(x) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(x), "v"(b) : "vcc");
            D16(S) D16(S)
#undef S
        } else if (OP == 6) {
#define S(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
            D16(S) D16(S)
#undef S
        } else if (OP == 7) {
#define S(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
            D16(S) D16(S)
#undef S
        } else if (OP == 8) {
#define S(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
            D16(S) D16(S)
#undef S
        } else if (OP == 9) {
#define S(x) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(i0));
            D16(S) D16(S)
#undef S
        } else if (OP == 10) {
#define S(x) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x));
            D16(S) D16(S)
#undef S
        } else if (OP == 11) { // v_frexp_exp_i32_f64 then v_cvt_f64_i32 (the log2 pair)
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i1) : "v"(a0));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i2) : "v"(a1));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i3) : "v"(a2));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i4) : "v"(a3));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i5) : "v"(a4));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i6) : "v"(a5));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i7) : "v"(a6));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i8) : "v"(a7));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i9) : "v"(a8));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i10) : "v"(a9));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i11) : "v"(a10));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i12) : "v"(a11));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i13) : "v"(a12));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i14) : "v"(a13));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i15) : "v"(a14));
            asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i0) : "v"(a15));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a0) : "v"(i1));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a1) : "v"(i2));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a2) : "v"(i3));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a3) : "v"(i4));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a4) : "v"(i5));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a5) : "v"(i6));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a6) : "v"(i7));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a7) : "v"(i8));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a8) : "v"(i9));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a9) : "v"(i10));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a10) : "v"(i11));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a11) : "v"(i12));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a12) : "v"(i13));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a13) : "v"(i14));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a14) : "v"(i15));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a15) : "v"(i0));
        } else if (OP == 12) {
#define S(x) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(b));
            D16(S) D16(S)
#undef S
        } else if (OP == 13) {
#define S(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
            D16(S) D16(S)
#undef S
        } else if (OP == 14) {
#define S(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
            D16(S) D16(S)
#undef S
        } else if (OP == 15) { // f64 fma with an SGPR-pair constant operand (as the polynomial steps have)
            double sc = 0.333;
            asm volatile("" : "+s"(sc));
#define S(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "s"(sc));
            D16(S) D16(S)
#undef S
        } else if (OP == 16) {
#define S(x) asm volatile("v_bfe_u32 %0, %0, 5, 11" : "+v"(x));
            I16(S) I16(S)
#undef S
        } else if (OP == 17) { // three quarters f64 fma, one quarter 32-bit
#define S(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define T(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(m));
            D16(S) T(i0) T(i1) T(i2) T(i3) T(i4) T(i5) T(i6) T(i7) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#undef S
#undef T
        } else if (OP == 18) { // select through an SGPR-pair mask (VOP3 form)
            unsigned long long msk = 0x5555555555555555ull;
            asm volatile("" : "+s"(msk));
#define S(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(m), "s"(msk));
            I16(S) I16(S)
#undef S
        } else if (OP == 19) { // the realistic pair: compare into vcc, select on it
#define S(x, y) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(x) : "v"(y), "v"(b), "v"(m) : "vcc");
            S(i0, a0) S(i1, a1) S(i2, a2) S(i3, a3) S(i4, a4) S(i5, a5) S(i6, a6) S(i7, a7)
            S(i8, a8) S(i9, a9) S(i10, a10) S(i11, a11) S(i12, a12) S(i13, a13) S(i14, a14) S(i15, a15)
#undef S
        } else if (OP == 20) { // select with dst != src
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i0) : "v"(i1), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i2) : "v"(i3), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i4) : "v"(i5), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i6) : "v"(i7), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i8) : "v"(i9), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i10) : "v"(i11), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i12) : "v"(i13), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i14) : "v"(i15), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i1) : "v"(i0), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i3) : "v"(i2), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i5) : "v"(i4), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i7) : "v"(i6), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i9) : "v"(i8), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i11) : "v"(i10), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i13) : "v"(i12), "v"(m));
            asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(i15) : "v"(i14), "v"(m));
        } else if (OP == 21) {
#define S(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(m));
            I16(S) I16(S)
#undef S
        } else if (OP == 22) {
#define S(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(m));
            I16(S) I16(S)
#undef S
        } else if (OP == 23) { // f64 add with a literal-constant operand
#define S(x) asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(x));
            D16(S) D16(S)
#undef S
        } else if (OP == 24) { // one select per four f64 fma (the density of the Float64 column kernel's clamp)
#define S(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define T(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(m));
            S(a0) S(a1) S(a2) S(a3) T(i0) S(a4) S(a5) S(a6) S(a7) T(i1) S(a8) S(a9) S(a10) S(a11) T(i2) S(a12) S(a13) S(a14) S(a15) T(i3)
            S(a0) S(a1) S(a2) S(a3) T(i4) S(a4) S(a5) S(a6) S(a7) T(i5) S(a8) S(a9) S(a10) T(i6) T(i7)
#undef S
#undef T
        } else if (OP == 25) {
#define S(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            I16(S) I16(S)
#undef S
        } else if (OP == 26) {
#define S(x) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(i0) : "v"(x));
            D16(S) D16(S)
#undef S
        } else if (OP == 27) { // DPP move
#define S(x) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
            I16(S) I16(S)
#undef S
        } else if (OP == 28) {
#define S(x) asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(x) : "s20");
            I16(S) I16(S)
#undef S
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15 +
                                                 i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + i8 + i9 + i10 + i11 + i12 + i13 + i14 + i15;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, int per_iter = 32) {
    const int blocks = 256 * 2;
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 1024);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<OP><<<blocks, 1024>>>(out, cyc, 1.5, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) k<OP><<<blocks, 1024>>>(out, cyc, 1.5, 3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<unsigned long long> h(blocks * 16);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 16, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= h.size();
    const double ninst = (double)ITERS * per_iter; // per wave
    // 8 waves share a SIMD: cycles per instruction per SIMD = wave ticks / (8 x instructions of one wave)
    printf("%-34s ticks/inst/SIMD=%.3f   wall: %.3f ns per inst per SIMD (launch %.4f ms)  -> clock if tick=cycle: %.2f GHz\n", name,
           avg / ninst / 8.0, ms * 1e6 / (ninst * 8.0), ms, (avg / ninst / 8.0) / (ms * 1e6 / (ninst * 8.0)));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fma_f64");
    run<13>("v_add_f64");
    run<14>("v_mul_f64");
    run<15>("v_fma_f64 (SGPR const)");
    run<1>("v_and_b32");
    run<2>("v_lshl_add_u32");
    run<16>("v_bfe_u32");
    run<3>("16 v_fma_f64 + 16 v_and_b32");
    run<17>("24 v_fma_f64 + 8 v_and_b32");
    run<4>("v_cndmask_b32");
    run<5>("v_cmp_lt_f64");
    run<6>("v_max_f64");
    run<7>("v_rcp_f64");
    run<8>("v_rsq_f64");
    run<9>("v_ldexp_f64");
    run<10>("v_frexp_mant_f64");
    run<11>("16 frexp_exp_i32_f64 + 16 cvt_f64_i32");
    run<12>("v_mov_b64");
    run<18>("v_cndmask_b32_e64 (SGPR mask)");
    run<19>("16 x (v_cmp_lt_f64 vcc + v_cndmask_b32)");
    run<20>("v_cndmask_b32 dst != src", 16);
    run<21>("v_max_f32");
    run<22>("v_fma_f32");
    run<23>("v_add_f64 (inline constant)");
    run<24>("27 v_fma_f64 + 8 v_cndmask_b32", 35);
    run<25>("v_exp_f32");
    run<26>("v_cvt_f32_f64");
    run<27>("v_mov_b32_dpp row_shr:1");
    run<28>("v_readfirstlane_b32");
    return 0;
}

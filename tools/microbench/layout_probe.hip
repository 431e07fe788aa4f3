// layout_probe.hip -- streaming rate of the column kernel's access pattern under two
// plane layouts, with a tunable amount of dependent arithmetic per level:
//   level-major  a[lev*stride + col]            (rows 8 MB apart for 1e6 columns)
//   tile-major   a[(tile*nlev + lev)*64 + lane] (a wave's 64 levels are one 32-KiB block)
// 2 planes read + 2 planes written per level, one lane per column, nontemporal.
// Build: hipcc -O3 --offload-arch=gfx950 layout_probe.hip -o layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool TILE, int WORK>
__global__ void __launch_bounds__(256) probe(const double* __restrict__ in0, const double* __restrict__ in1,
                                              double* __restrict__ out0, double* __restrict__ out1,
                                              long ncols, long stride, int nlev) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    const long base = TILE ? (col >> 6) * (long)nlev * 64 + (col & 63) : col;
    const long step = TILE ? 64 : stride;
    double prev = 0.0;
    double a = __builtin_nontemporal_load(in0 + base), b = __builtin_nontemporal_load(in1 + base);
    for (int i = 0; i < nlev; ++i) {
        const long o = base + (long)i * step;
        double an = 0, bn = 0;
        if (i + 1 < nlev) { an = __builtin_nontemporal_load(in0 + o + step); bn = __builtin_nontemporal_load(in1 + o + step); }
        double x = a + b;
#pragma unroll
        for (int k = 0; k < WORK; ++k) x = __builtin_fma(x, 0.999999, 1e-9 * k);   // dependent chain
        __builtin_nontemporal_store(x - prev, out0 + o);
        __builtin_nontemporal_store(0.0, out1 + o);
        prev = x; a = an; b = bn;
    }
}

// level-pair interleaved planes: a[(lev/2) * 2*stride + col*2 + (lev&1)]: one 16-byte access
// per lane brings two consecutive levels of its column (1 KiB per wave instruction)
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int WORK>
__global__ void __launch_bounds__(256) probe_pair(const double* __restrict__ in0, const double* __restrict__ in1,
                                                   double* __restrict__ out0, double* __restrict__ out1,
                                                   long ncols, long stride, int nlev) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    const long base = col * 2, step = 2 * stride;
    double prev = 0.0;
    dbl2 a = __builtin_nontemporal_load((const dbl2*)(in0 + base)), b = __builtin_nontemporal_load((const dbl2*)(in1 + base));
    for (int i = 0; i < nlev; i += 2) {
        const long o = base + (long)(i / 2) * step;
        dbl2 an = {0, 0}, bn = {0, 0};
        if (i + 2 < nlev) { an = __builtin_nontemporal_load((const dbl2*)(in0 + o + step)); bn = __builtin_nontemporal_load((const dbl2*)(in1 + o + step)); }
        dbl2 r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double x = a[h] + b[h];
#pragma unroll
            for (int k = 0; k < WORK; ++k) x = __builtin_fma(x, 0.999999, 1e-9 * k);
            r[h] = x - prev;
            prev = x;
        }
        __builtin_nontemporal_store(r, (dbl2*)(out0 + o));
        dbl2 z = {0, 0};
        __builtin_nontemporal_store(z, (dbl2*)(out1 + o));
        a = an; b = bn;
    }
}
template <int WORK>
float run_pair(double* d[4], long ncols, long stride, int nlev, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g((unsigned)((ncols + 255) / 256)), b(256);
    for (int i = 0; i < 3; ++i) probe_pair<WORK><<<g, b>>>(d[0], d[1], d[2], d[3], ncols, stride, nlev);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) probe_pair<WORK><<<g, b>>>(d[0], d[1], d[2], d[3], ncols, stride, nlev);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <bool TILE, int WORK>
float run(double* d[4], long ncols, long stride, int nlev, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g((unsigned)((ncols + 255) / 256)), b(256);
    for (int i = 0; i < 3; ++i) probe<TILE, WORK><<<g, b>>>(d[0], d[1], d[2], d[3], ncols, stride, nlev);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) probe<TILE, WORK><<<g, b>>>(d[0], d[1], d[2], d[3], ncols, stride, nlev);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

// plain device copy, 16 B per lane, contiguous: the "measured copy ceiling" for the same bytes
__global__ void __launch_bounds__(256) copy16(const dbl2* __restrict__ in, dbl2* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    for (; i < n; i += step) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

int main(int argc, char** argv) {
    const long ncols = argc > 1 ? atol(argv[1]) : 1000000;
    const int nlev = argc > 2 ? atoi(argv[2]) : 64;
    const long stride = (ncols + 63) / 64 * 64;
    const size_t bytes = (size_t)nlev * stride * sizeof(double);
    double* d[4];
    for (int i = 0; i < 4; ++i) { hipMalloc(&d[i], bytes); hipMemset(d[i], 0, bytes); }
    const double gb = 4.0 * ncols * nlev * 8 / 1e9;
    for (int round = 0; round < 2; ++round) {
#define LINE(T, W) { float ms = run<T, W>(d, ncols, stride, nlev, 20); \
        printf("round %d  %-11s work=%3d  %.4f ms  %7.1f GB/s  %.1f%% of 8 TB/s\n", round, T ? "tile-major" : "level-major", W, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 80.0); }
        LINE(false, 0) LINE(true, 0) LINE(false, 40) LINE(true, 40)
#define PLINE(W) { float ms = run_pair<W>(d, ncols, stride, nlev, 20); \
        printf("round %d  %-11s work=%3d  %.4f ms  %7.1f GB/s  %.1f%% of 8 TB/s\n", round, "level-pair", W, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 80.0); }
        PLINE(0) PLINE(40)
        {   // copy planes 0,1 -> 2,3 as two contiguous 16-B/lane copies (same 2 R + 2 W bytes)
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const long n2 = (long)nlev * stride / 2;
            for (int w = 0; w < 2; ++w) { copy16<<<8192, 256>>>((const dbl2*)d[0], (dbl2*)d[2], n2); copy16<<<8192, 256>>>((const dbl2*)d[1], (dbl2*)d[3], n2); }
            hipEventRecord(e0);
            for (int r = 0; r < 20; ++r) { copy16<<<8192, 256>>>((const dbl2*)d[0], (dbl2*)d[2], n2); copy16<<<8192, 256>>>((const dbl2*)d[1], (dbl2*)d[3], n2); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("round %d  %-11s           %.4f ms  %7.1f GB/s  %.1f%% of 8 TB/s\n", round, "plain copy", ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 80.0);
        }
    }
    return 0;
}

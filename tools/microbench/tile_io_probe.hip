// tile_io_probe.hip -- what does the persistent column stepper's prologue/epilogue cost?
// Workgroup = CPB columns x 64 levels (one thread per cell), planes column-fastest [nlev][stride].
// Variants: load 2 planes through LDS tiles (L), store 1 plane (S), both (LS), with an optional
// dependent-FMA "compute" phase between them (work = number of FMAs).
// Build: hipcc -O3 --offload-arch=gfx950 tile_io_probe.hip -o tile_io_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CPB, bool LOAD, bool STORE>
__global__ void __launch_bounds__(64 * CPB) probe(const double* __restrict__ a, const double* __restrict__ b,
                                                   double* __restrict__ out, long ncols, long stride, int work) {
    constexpr int n = 64;
    __shared__ double tiles[2 * CPB * n];
    const int slot = threadIdx.x / n, i = threadIdx.x % n;
    const long col_first = (long)blockIdx.x * CPB;
    double x = 1.0, y = 2.0;
    if (LOAD) {
        for (int e = threadIdx.x; e < n * CPB; e += blockDim.x) {
            const int lev = e / CPB, cs = e - lev * CPB;
            if (col_first + cs < ncols) {
                tiles[cs * n + lev] = a[(long)lev * stride + col_first + cs];
                tiles[CPB * n + cs * n + lev] = b[(long)lev * stride + col_first + cs];
            }
        }
        __syncthreads();
        x = tiles[slot * n + i];
        y = tiles[CPB * n + slot * n + i];
        __syncthreads();
    }
    for (int k = 0; k < work; ++k) x = __builtin_fma(x, 0.999999, y * 1e-9);
    if (STORE) {
        tiles[slot * n + i] = x;
        __syncthreads();
        for (int e = threadIdx.x; e < n * CPB; e += blockDim.x) {
            const int lev = e / CPB, cs = e - lev * CPB;
            if (col_first + cs < ncols) out[(long)lev * stride + col_first + cs] = tiles[cs * n + lev];
        }
    } else if (x == 12345.678) out[0] = x;
}

template <int CPB, bool L, bool S>
void run(const char* name, double* d[3], long ncols, long stride, int work) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g((unsigned)((ncols + CPB - 1) / CPB)), b(64 * CPB);
    for (int i = 0; i < 3; ++i) probe<CPB, L, S><<<g, b>>>(d[0], d[1], d[2], ncols, stride, work);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) probe<CPB, L, S><<<g, b>>>(d[0], d[1], d[2], ncols, stride, work);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double gb = (double)ncols * 64 * 8 * ((L ? 2 : 0) + (S ? 1 : 0)) / 1e9;
    printf("  cpb=%2d %-4s work=%4d  %.4f ms  %7.1f GB/s\n", CPB, name, work, ms, gb / (ms * 1e-3));
}

int main() {
    const long ncols = 1000000, stride = 1000064; const int nlev = 64;
    const size_t bytes = (size_t)nlev * stride * 8;
    double* d[3];
    for (int i = 0; i < 3; ++i) { hipMalloc(&d[i], bytes); hipMemset(d[i], 0, bytes); }
    for (int work : {0, 400}) {
        run<4, true, false>("L", d, ncols, stride, work);
        run<4, false, true>("S", d, ncols, stride, work);
        run<4, true, true>("LS", d, ncols, stride, work);
        run<8, true, true>("LS", d, ncols, stride, work);
        run<16, true, true>("LS", d, ncols, stride, work);
        run<1, true, true>("LS", d, ncols, stride, work);
        run<4, false, false>("-", d, ncols, stride, work);
    }
    return 0;
}

// write_probe.hip -- is the fast/slow placement effect a property of single buffers or of
// buffer pairs?  N buffers of 512 MB (own hipMalloc each); measures
//   (1) streaming WRITE of each buffer alone,  (2) streaming READ of each buffer alone,
//   (3) lock-step write of the pair (i, j) for a few i and all j (one 8-B store per lane to each
//       buffer per step, like the two written planes of the Richards tendency launch).
// Build: hipcc -O3 --offload-arch=gfx950 write_probe.hip -o write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) fill1(double* __restrict__ a, long ncols, long stride, int nlev) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    for (int i = 0; i < nlev; ++i) __builtin_nontemporal_store((double)i, a + (long)i * stride + col);
}
__global__ void __launch_bounds__(256) read1(const double* __restrict__ a, double* sink, long ncols, long stride, int nlev) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    double s = 0;
    for (int i = 0; i < nlev; ++i) s += __builtin_nontemporal_load(a + (long)i * stride + col);
    if (s == 12345.678) *sink = s;
}
__global__ void __launch_bounds__(256) fill2(double* __restrict__ a, double* __restrict__ b, long ncols, long stride, int nlev) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    for (int i = 0; i < nlev; ++i) {
        __builtin_nontemporal_store((double)i, a + (long)i * stride + col);
        __builtin_nontemporal_store(0.0, b + (long)i * stride + col);
    }
}

template <typename L>
float timeit(L&& launch, int reps = 10) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 24;
    const long ncols = 1000000; const int nlev = 64;
    const long stride = 1000064;
    const size_t bytes = (size_t)nlev * stride * sizeof(double);
    std::vector<double*> d(N);
    for (int i = 0; i < N; ++i) { if (hipMalloc(&d[i], bytes) != hipSuccess) { printf("alloc %d failed\n", i); return 1; } hipMemset(d[i], 0, bytes); }
    double* sink; hipMalloc(&sink, 8);
    dim3 g((unsigned)((ncols + 255) / 256)), b(256);
    const double gb = (double)ncols * nlev * 8 / 1e9;
    for (int w = 0; w < 30; ++w) fill1<<<g, b>>>(d[0], ncols, stride, nlev);
    printf("buffer VA (MiB rel. to lowest), write-alone GB/s, read-alone GB/s\n");
    size_t lo = ~size_t(0); for (int i = 0; i < N; ++i) lo = (size_t)d[i] < lo ? (size_t)d[i] : lo;
    for (int i = 0; i < N; ++i) {
        float w = timeit([&] { fill1<<<g, b>>>(d[i], ncols, stride, nlev); });
        float r = timeit([&] { read1<<<g, b>>>(d[i], sink, ncols, stride, nlev); });
        printf("  buf %2d  +%6zu MiB   write %7.1f   read %7.1f\n", i, ((size_t)d[i] - lo) >> 20, gb / (w * 1e-3), gb / (r * 1e-3));
    }
    printf("pair write (GB/s of the two stores), rows = first buffer, columns = second buffer\n");
    const int firsts[4] = {0, N / 3, 2 * N / 3, N - 1};
    for (int fi = 0; fi < 4; ++fi) {
        const int i = firsts[fi];
        printf("  buf %2d:", i);
        for (int j = 0; j < N; ++j) {
            if (j == i) { printf("    -- "); continue; }
            float t = timeit([&] { fill2<<<g, b>>>(d[i], d[j], ncols, stride, nlev); }, 6);
            printf(" %6.0f", 2 * gb / (t * 1e-3));
        }
        printf("\n");
    }
    return 0;
}

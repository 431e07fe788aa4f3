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

// one write stream whose columns are split between two buffers: the lower half of the blocks
// writes its columns into a, the upper half into b (same bytes as fill1)
__global__ void __launch_bounds__(256) fill_split(double* __restrict__ a, double* __restrict__ b, long ncols, long stride, int nlev) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    double* p = (blockIdx.x < gridDim.x / 2) ? a : b;
    for (int i = 0; i < nlev; ++i) __builtin_nontemporal_store((double)i, p + (long)i * stride + col);
}
// a fused-stage-like stream: three planes read, one written (optionally column-split over two buffers)
__global__ void __launch_bounds__(256) stage31(const double* __restrict__ r0, const double* __restrict__ r1, const double* __restrict__ r2,
                                               double* __restrict__ wa, double* __restrict__ wb, long ncols, long stride, int nlev) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    double* p = (blockIdx.x < gridDim.x / 2) ? wa : wb;
    for (int i = 0; i < nlev; ++i) {
        const long o = (long)i * stride + col;
        const double x = __builtin_nontemporal_load(r0 + o) + __builtin_nontemporal_load(r1 + o) + __builtin_nontemporal_load(r2 + o);
        __builtin_nontemporal_store(x, p + o);
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
    if (N <= 32)
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
    // column-split experiments: a = buffer 0, b = the first buffer that pairs fast with it
    float r0b[256]; int bsel = -1; float best = 0;
    for (int j = 1; j < N; ++j) { float t = timeit([&] { fill2<<<g, b>>>(d[0], d[j], ncols, stride, nlev); }, 6); r0b[j] = 2 * gb / (t * 1e-3); if (r0b[j] > best) { best = r0b[j]; bsel = j; } }
    int same = -1; for (int j = 1; j < N; ++j) if (j != bsel && r0b[j] < 0.9f * best) { same = j; break; }
    if (bsel > 0 && same > 0) {
        printf("a = buf 0, b = buf %d (pair %.0f GB/s), s = buf %d (same kind as a, pair %.0f)\n", bsel, best, same, r0b[same]);
        float t;
        t = timeit([&] { fill1<<<g, b>>>(d[0], ncols, stride, nlev); });              printf("  write a alone            %7.1f GB/s\n", gb / (t * 1e-3));
        t = timeit([&] { fill1<<<g, b>>>(d[bsel], ncols, stride, nlev); });           printf("  write b alone            %7.1f GB/s\n", gb / (t * 1e-3));
        t = timeit([&] { fill_split<<<g, b>>>(d[0], d[bsel], ncols, stride, nlev); }); printf("  write split a|b          %7.1f GB/s\n", gb / (t * 1e-3));
        t = timeit([&] { fill_split<<<g, b>>>(d[0], d[same], ncols, stride, nlev); }); printf("  write split a|s          %7.1f GB/s\n", gb / (t * 1e-3));
        int r[3], k = 0; for (int j = 1; j < N && k < 3; ++j) if (j != bsel && j != same) r[k++] = j;
        t = timeit([&] { stage31<<<g, b>>>(d[r[0]], d[r[1]], d[r[2]], d[0], d[0], ncols, stride, nlev); });    printf("  3 reads + write a        %7.1f GB/s (4 planes)\n", 4 * gb / (t * 1e-3));
        t = timeit([&] { stage31<<<g, b>>>(d[r[0]], d[r[1]], d[r[2]], d[bsel], d[bsel], ncols, stride, nlev); }); printf("  3 reads + write b        %7.1f GB/s\n", 4 * gb / (t * 1e-3));
        t = timeit([&] { stage31<<<g, b>>>(d[r[0]], d[r[1]], d[r[2]], d[0], d[bsel], ncols, stride, nlev); }); printf("  3 reads + write a|b      %7.1f GB/s\n", 4 * gb / (t * 1e-3));
        t = timeit([&] { stage31<<<g, b>>>(d[r[0]], d[r[1]], d[bsel], d[0], d[0], ncols, stride, nlev); });    printf("  reads (s,s,b) + write a  %7.1f GB/s\n", 4 * gb / (t * 1e-3));
        t = timeit([&] { stage31<<<g, b>>>(d[0], d[r[1]], d[r[2]], d[0], d[0], ncols, stride, nlev); });       printf("  in place: reads (a,s,s) + write a %7.1f GB/s\n", 4 * gb / (t * 1e-3));
    }
    return 0;
}

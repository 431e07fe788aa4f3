// class_map.hip -- map the "write class" of device memory at fine granularity.
// One large allocation; every WIN-MiB window is written in lock-step with a fixed reference
// window (fill2 of write_probe.hip): pairs in the same class run ~5.5 TB/s, pairs in different
// classes ~6.9 TB/s.  Prints one character per window: '.' same class as window 0, '#' other.
// usage: class_map [total_GiB=24] [win_MiB=128]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) fill2(double* __restrict__ a, double* __restrict__ b, long ncols, long stride, int nlev) {
    const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    for (int i = 0; i < nlev; ++i) {
        __builtin_nontemporal_store((double)i, a + (long)i * stride + col);
        __builtin_nontemporal_store(0.0, b + (long)i * stride + col);
    }
}
template <typename L>
float timeit(L&& launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / reps;
}
int main(int argc, char** argv) {
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 24) << 30;
    const size_t win = (size_t)(argc > 2 ? atoi(argv[2]) : 128) << 20;
    char* base;
    if (hipMalloc(&base, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(base, 0, total);
    const long ncols = 1 << 20;                  // 8 MiB rows
    const long stride = ncols;
    const int nlev = (int)(win / (ncols * 8));
    const int nwin = (int)(total / win);
    dim3 g((unsigned)(ncols / 256)), b(256);
    const double gb = 2.0 * ncols * nlev * 8 / 1e9;
    auto W = [&](int w) { return reinterpret_cast<double*>(base + (size_t)w * win); };
    for (int i = 0; i < 50; ++i) fill2<<<g, b>>>(W(0), W(1), ncols, stride, nlev);
    std::vector<float> r0(nwin), r1(nwin);
    // reference 0 = window 0; reference 1 = the first window that pairs fast with window 0
    for (int w = 1; w < nwin; ++w) r0[w] = gb / (timeit([&] { fill2<<<g, b>>>(W(0), W(w), ncols, stride, nlev); }, 6) * 1e-3);
    float lo = 1e9, hi = 0; for (int w = 1; w < nwin; ++w) { lo = r0[w] < lo ? r0[w] : lo; hi = r0[w] > hi ? r0[w] : hi; }
    printf("window %zu MiB, %d windows, pair-with-window-0 rate min %.0f max %.0f GB/s\n", win >> 20, nwin, lo, hi);
    const float thr = 0.5f * (lo + hi);
    int other = -1;
    for (int w = 1; w < nwin && other < 0; ++w) if (r0[w] > thr) other = w;
    printf("vs window 0   : ");
    for (int w = 0; w < nwin; ++w) putchar(w == 0 ? '0' : (r0[w] > thr ? '#' : '.'));
    printf("\n");
    if (other > 0) {
        for (int w = 0; w < nwin; ++w) if (w != other) r1[w] = gb / (timeit([&] { fill2<<<g, b>>>(W(other), W(w), ncols, stride, nlev); }, 6) * 1e-3);
        printf("vs window %-4d: ", other);
        for (int w = 0; w < nwin; ++w) putchar(w == other ? '0' : (r1[w] > thr ? '#' : '.'));
        printf("\n");
    }
    printf("rates vs window 0 (GB/s):");
    for (int w = 1; w < nwin; ++w) printf("%s%.0f", (w % 16 == 1) ? "\n  " : " ", r0[w]);
    printf("\n");
    return 0;
}

// gather_probe.hip -- what does it cost when the lanes of a wave read columns PERMUTED inside a
// window of W columns instead of 64 consecutive ones (rhs_kernel's access pattern otherwise: one lane
// per column, bottom -> top, 8 B per lane per level, NR planes read and NW written)?  The question
// behind "sort ice-free and icy columns into separate waves by an index list" (round-3 verdict item 3).
// Build: hipcc -O3 --offload-arch=gfx950 gather_probe.hip -o gather_probe ; run: ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NR, int NW>
__global__ void __launch_bounds__(256, 8)
k(const double* __restrict__ in, double* __restrict__ out, long ncols, long stride, int nlev, int W, int P) {
    const long g = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= ncols) return;
    long col = g;
    if (W > 1) {
        const long w = g / W, i = g % W;
        col = w * W + (i * P) % W; // a permutation of the window (P odd, W a power of two)
        if (col >= ncols) col = g;
    }
    const size_t plane = size_t(nlev) * stride;
    for (int i = 0; i < nlev; ++i) {
        double x = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) x += __builtin_nontemporal_load(in + r * plane + size_t(i) * stride + col);
#pragma unroll
        for (int q = 0; q < NW; ++q) __builtin_nontemporal_store(x, out + q * plane + size_t(i) * stride + col);
    }
}

template <int NR, int NW>
void run(long ncols, int nlev) {
    const long stride = ((ncols + 63) / 64 * 64) + 64;
    double *in, *out;
    hipMalloc(&in, sizeof(double) * NR * nlev * stride);
    hipMalloc(&out, sizeof(double) * NW * nlev * stride);
    hipMemset(in, 0, sizeof(double) * NR * nlev * stride);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = int((ncols + 255) / 256);
    printf("%d planes read, %d written, %ld columns x %d levels (%.2f GB per launch)\n", NR, NW, ncols, nlev,
           double(NR + NW) * nlev * ncols * 8 / 1e9);
    for (int W : {1, 128, 256, 512, 1024, 4096}) {
        for (int w = 0; w < 3; ++w) k<NR, NW><<<blocks, 256>>>(in, out, ncols, stride, nlev, W, 37);
        hipEventRecord(e0);
        const int reps = 10;
        for (int w = 0; w < reps; ++w) k<NR, NW><<<blocks, 256>>>(in, out, ncols, stride, nlev, W, 37);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        printf("  window %5d: %.4f ms  %.0f GB/s\n", W, ms, double(NR + NW) * nlev * ncols * 8 / (ms * 1e-3) / 1e9);
    }
    hipFree(in);
    hipFree(out);
}

int main() {
    run<3, 2>(1000000, 48);   // the Float64 coupled tendency (f3c64)
    run<1, 1>(1000000, 64);   // C2
    return 0;
}

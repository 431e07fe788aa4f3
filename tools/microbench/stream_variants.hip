// stream_variants.hip -- why does a naive one-lane-per-column loop (gather_probe, W = 1) stream at
// 6.0-6.1 TB/s when lh_stream_probe reaches 5.3?  Variants of the same traffic (C2: one plane read, one
// written, 1e6 columns x 64 levels, 8 B per lane per level):
//   alloc: separate hipMallocs for the two planes | one allocation, planes back to back (the arena)
//   data:  zeros | random
//   stride: ncols + 64 | odd multiple of 512 B (lh_create's rule)
// Build: hipcc -O3 --offload-arch=gfx950 stream_variants.hip -o stream_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256, 8)
k(const double* __restrict__ in, double* __restrict__ out, long ncols, long stride, int nlev) {
    const long col = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    for (int i = 0; i < nlev; ++i) {
        const double x = __builtin_nontemporal_load(in + size_t(i) * stride + col);
        __builtin_nontemporal_store(x + 1.0, out + size_t(i) * stride + col);
    }
}

int main() {
    const long ncols = 1000000;
    const int nlev = 64;
    for (int oddstride = 0; oddstride < 2; ++oddstride)
        for (int onealloc = 0; onealloc < 2; ++onealloc)
            for (int randdata = 0; randdata < 2; ++randdata) {
                long stride = ((ncols + 63) / 64 * 64) + 64;
                if (oddstride) {
                    long units = (ncols + 63) / 64;
                    if (units % 2 == 0) ++units;
                    stride = units * 64;
                }
                const size_t plane = size_t(nlev) * stride;
                const size_t slot = ((plane * 8 + (size_t(2) << 20) - 1) >> 21) << 21; // 2-MiB slots as the arena
                double *in, *out, *base = nullptr;
                if (onealloc) {
                    hipMalloc(&base, slot * 8);
                    in = base;
                    out = reinterpret_cast<double*>(reinterpret_cast<char*>(base) + 2 * slot); // (slot 2: as Y.vl, Y.ti, dY.vl)
                } else {
                    hipMalloc(&in, plane * 8);
                    hipMalloc(&out, plane * 8);
                }
                std::vector<double> h(plane);
                for (size_t q = 0; q < plane; ++q) h[q] = randdata ? 0.1 + 0.3 * (double)rand() / RAND_MAX : 0.0;
                hipMemcpy(in, h.data(), plane * 8, hipMemcpyHostToDevice);
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                const int blocks = int((ncols + 255) / 256);
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    for (int w = 0; w < 5; ++w) k<<<blocks, 256>>>(in, out, ncols, stride, nlev);
                    hipEventRecord(e0);
                    for (int w = 0; w < 20; ++w) k<<<blocks, 256>>>(in, out, ncols, stride, nlev);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    ms /= 20;
                    if (ms < best) best = ms;
                }
                printf("stride %s  %s  data %s: %.4f ms  %.0f GB/s\n", oddstride ? "odd x512B" : "ncols+64 ", onealloc ? "one allocation " : "two allocations",
                       randdata ? "random" : "zeros ", best, 2.0 * nlev * ncols * 8 / (best * 1e-3) / 1e9);
                if (onealloc) hipFree(base);
                else { hipFree(in); hipFree(out); }
            }
    return 0;
}

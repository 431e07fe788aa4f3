// probe_vs_naive.hip -- the library's stream_probe_kernel (buffer-descriptor rows, prefetch ring, XCD map)
// against a naive one-lane-per-column loop on the SAME two buffers (C2 traffic: one plane read, one written).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../landhydrology.jl_amd/csrc probe_vs_naive.hip -o probe_vs_naive
#define LH_TU_MODEL
#include "lh_kernels_impl.hpp"
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256, 8)
naive(const double* __restrict__ in, double* __restrict__ out, long ncols, long stride, int nlev) {
    const long col = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    for (int i = 0; i < nlev; ++i) {
        const double x = __builtin_nontemporal_load(in + size_t(i) * stride + col);
        __builtin_nontemporal_store(x + 1.0, out + size_t(i) * stride + col);
    }
}
// the naive loop with the workgroup -> column-block map of rhs_kernel
__global__ void __launch_bounds__(256, 8)
naive_xcd(const double* __restrict__ in, double* __restrict__ out, long ncols, long stride, int nlev) {
    unsigned blk = blockIdx.x;
    const unsigned per = gridDim.x >> 3;
    if (blk < (per << 3)) blk = (blk & 7u) * per + (blk >> 3);
    const long col = long(blk) * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    for (int i = 0; i < nlev; ++i) {
        const double x = __builtin_nontemporal_load(in + size_t(i) * stride + col);
        __builtin_nontemporal_store(x + 1.0, out + size_t(i) * stride + col);
    }
}

template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        for (int w = 0; w < 5; ++w) f();
        hipEventRecord(e0);
        for (int w = 0; w < 20; ++w) f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    return best;
}

int main() {
    const long ncols = 1000000;
    for (int nlev : {64, 128}) {
        long units = (ncols + 63) / 64;
        if (units % 2 == 0) ++units;
        const long stride = units * 64;
        const size_t plane = size_t(nlev) * stride;
        double *in, *out;
        hipMalloc(&in, plane * 8);
        hipMalloc(&out, plane * 8);
        hipMemset(in, 0, plane * 8);
        const int blocks = int((ncols + 255) / 256);
        lh::Planes<double> pi{}, po{};
        pi.v[0] = in;
        po.v[0] = out;
        const double gb = 2.0 * nlev * ncols * 8 / 1e9;
        auto rep = [&](const char* name, float ms) { printf("nlev %3d  %-44s %.4f ms  %.0f GB/s\n", nlev, name, ms, gb / (ms * 1e-3)); };
        rep("naive loop", timeit([&] { naive<<<blocks, 256>>>(in, out, ncols, stride, nlev); }));
        rep("naive loop + XCD-contiguous block map", timeit([&] { naive_xcd<<<blocks, 256>>>(in, out, ncols, stride, nlev); }));
        rep("stream_probe_kernel PF=1 nt, identity map", timeit([&] { hipLaunchKernelGGL((lh::stream_probe_kernel<double, 1, 1, true>), dim3(blocks), dim3(256), 0, 0, ncols, stride, nlev, 0, pi, 1, po, 1); }));
        rep("stream_probe_kernel PF=2 nt, identity map", timeit([&] { hipLaunchKernelGGL((lh::stream_probe_kernel<double, 1, 2, true>), dim3(blocks), dim3(256), 0, 0, ncols, stride, nlev, 0, pi, 1, po, 1); }));
        rep("stream_probe_kernel PF=2 nt, XCD map (library)", timeit([&] { hipLaunchKernelGGL((lh::stream_probe_kernel<double, 1, 2, true>), dim3(blocks), dim3(256), 0, 0, ncols, stride, nlev, 1, pi, 1, po, 1); }));
        rep("stream_probe_kernel PF=4 nt, identity map", timeit([&] { hipLaunchKernelGGL((lh::stream_probe_kernel<double, 1, 4, true>), dim3(blocks), dim3(256), 0, 0, ncols, stride, nlev, 0, pi, 1, po, 1); }));
        rep("stream_probe_kernel PF=2 plain, identity map", timeit([&] { hipLaunchKernelGGL((lh::stream_probe_kernel<double, 1, 2, false>), dim3(blocks), dim3(256), 0, 0, ncols, stride, nlev, 0, pi, 1, po, 1); }));
        hipFree(in);
        hipFree(out);
    }
    return 0;
}

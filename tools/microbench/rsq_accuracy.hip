// Accuracy of the hardware seeds v_rsq_f64 / v_rcp_f64 on gfx950 (how many Newton terms sqrt_mul needs):
// max and rms of |y sqrt(x) - 1| over 4M logarithmically spread x in [2^-60, 2^4].
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void seeds(const double* x, double* rs, double* rc, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rs[i] = __builtin_amdgcn_rsq(x[i]);
    rc[i] = __builtin_amdgcn_rcp(x[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), rs(n), rc(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = double(s >> 11) / 9007199254740992.0;
        x[i] = std::exp2(-60.0 + 64.0 * u);
    }
    double *dx, *drs, *drc;
    hipMalloc(&dx, n * 8); hipMalloc(&drs, n * 8); hipMalloc(&drc, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    seeds<<<n / 256, 256>>>(dx, drs, drc, n);
    hipMemcpy(rs.data(), drs, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(rc.data(), drc, n * 8, hipMemcpyDeviceToHost);
    long double mrs = 0, mrc = 0, srs = 0, src = 0;
    for (int i = 0; i < n; ++i) {
        const long double ers = fabsl((long double)rs[i] * sqrtl((long double)x[i]) - 1.0L);
        const long double erc = fabsl((long double)rc[i] * (long double)x[i] - 1.0L);
        if (ers > mrs) mrs = ers;
        if (erc > mrc) mrc = erc;
        srs += ers * ers; src += erc * erc;
    }
    printf("v_rsq_f64: max rel err %.3Le = 2^%.2Lf, rms %.3Le\n", mrs, log2l(mrs), sqrtl(srs / n));
    printf("v_rcp_f64: max rel err %.3Le = 2^%.2Lf, rms %.3Le\n", mrc, log2l(mrc), sqrtl(src / n));
    return 0;
}

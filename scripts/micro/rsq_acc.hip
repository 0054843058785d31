// Accuracy of v_rsq_f64 / v_rcp_f64 and of one / two Newton refinements (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rsq(v);
    o[i] = r;
    double e = fma(-(v * r), r, 1.0);
    double r1 = fma(0.5 * r, e, r);
    o[n + i] = r1;
    e = fma(-(v * r1), r1, 1.0);
    o[2 * n + i] = fma(0.5 * r1, e, r1);
    double c = __builtin_amdgcn_rcp(v);
    o[3 * n + i] = c;
    o[4 * n + i] = fma(fma(-v, c, 1.0), c, c);
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(5 * n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::exp((u - 0.5) * 60.0);
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 5 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
    const char *nm[5] = {"rsq raw", "rsq 1 NR", "rsq 2 NR", "rcp raw", "rcp 1 NR"};
    for (int k = 0; k < 5; ++k) {
        double m = 0;
        for (int i = 0; i < n; ++i) {
            long double ref = k < 3 ? 1.0L / sqrtl((long double)x[i]) : 1.0L / (long double)x[i];
            double e = (double)fabsl(((long double)o[k * n + i] - ref) / ref);
            if (e > m) m = e;
        }
        printf("%s max rel err %.3e\n", nm[k], m);
    }
    return 0;
}

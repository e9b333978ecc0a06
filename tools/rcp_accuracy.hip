// Relative error of v_rcp_f64 alone and after one and two Newton steps, over random doubles of many magnitudes.
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/rcp_accuracy tools/rcp_accuracy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-v, r, 1.0), r, r);
    r2[i] = r;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    srand(7);
    for (int i = 0; i < n; i++) {
        const double m = 1.0 + rand() / (double)RAND_MAX;
        x[i] = ldexp(m, rand() % 200 - 100) * ((rand() & 1) ? 1 : -1);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        const long double t = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl(((long double)a[i] - t) / t)); e1 = fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
        e2 = fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
    }
    printf("max relative error: v_rcp_f64 %.3e   + 1 Newton step %.3e   + 2 Newton steps %.3e   (2^-53 = %.3e)\n", e0, e1, e2, ldexp(1.0, -53));
    return 0;
}

// accuracy of v_rcp_f64 with 0 / 1 / 2 Newton steps against 1.0/x (correctly rounded), in ulps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    r0[i] = r;
    double e = fma(-d, r, 1.0); r = fma(r, e, r);
    r1[i] = r;
    e = fma(-d, r, 1.0); r = fma(r, e, r);
    r2[i] = r;
}
static double ulps(double a, double b) { long long x, y; std::memcpy(&x, &a, 8); std::memcpy(&y, &b, 8); return (double)llabs(x - y); }
int main() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); h[i] = ldexp(1.0 + u, (int)(s % 80) - 40); }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0; long n1 = 0, n2 = 0;
    for (int i = 0; i < n; i++) { double t = 1.0 / h[i]; double u0 = ulps(a[i], t), u1 = ulps(b[i], t), u2 = ulps(c[i], t); m0 = fmax(m0, u0); m1 = fmax(m1, u1); m2 = fmax(m2, u2); n1 += u1 > 0; n2 += u2 > 0; }
    printf("max ulp error: raw v_rcp_f64 %.0f | +1 Newton %.0f (%ld of %d not exact) | +2 Newton %.0f (%ld not exact)\n", m0, m1, n1, n, m2, n2);
    return 0;
}

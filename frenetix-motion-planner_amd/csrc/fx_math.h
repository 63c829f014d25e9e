// fx_math.h -- small FP64 elementary functions for the evaluation kernel.
//
// The OCML versions of atan / cos / tan / sincos carry Payne-Hanek large-argument paths and cost the kernel
// ~115 VGPRs of pressure (235 -> 120 without them), i.e. half its occupancy.  The arguments here are bounded
// (|theta| <= a few pi; atan takes d' = d_dot/s_dot), so the classic fdlibm kernels with a two-constant
// FMA Cody-Waite reduction are enough: measured max error vs NumPy < 1 ulp for atan and < 2 ulp for
// sin/cos on |x| <= 64 (tests/test_hip_math.py) -- the same order as the libm-to-libm differences between
// the reference (NumPy SIMD kernels) and the CPU oracle (glibc).
#pragma once
#include <hip/hip_runtime.h>

namespace fxm {

// atan(x): fdlibm s_atan.c argument reduction (5 intervals) made branch-free with selects + one division.
__device__ __forceinline__ double atan(double x) {
    const double ax = fabs(x);
    double num = ax, den = 1.0, hi = 0.0, lo = 0.0;
    if (ax >= 0.4375) {
        if (ax < 0.6875) { num = 2.0 * ax - 1.0; den = 2.0 + ax; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
        else if (ax < 1.1875) { num = ax - 1.0; den = ax + 1.0; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
        else if (ax < 2.4375) { num = ax - 1.5; den = 1.0 + 1.5 * ax; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
        else { num = -1.0; den = ax; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
    }
    const double xr = num / den;
    const double z = xr * xr, w = z * z;
    const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02),
                                                    6.66107313738753120669e-02), 9.09088713343650656196e-02),
                                     1.42857142725034663711e-01), 3.33333333333329318027e-01);
    const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02),
                                             -7.69187620504482999495e-02), -1.11111104054623557880e-01),
                              -1.99999999998764832476e-01);
    const double r = hi - ((xr * (s1 + s2) - lo) - xr);
    return copysign(r, x);
}

// atan(x) for |x| < 7/16: the id = -1 branch of fdlibm (no reduction, no division); identical to atan() there
__device__ __forceinline__ double atan_small(double x) {
    const double z = x * x, w = z * z;
    const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02),
                                                    6.66107313738753120669e-02), 9.09088713343650656196e-02),
                                     1.42857142725034663711e-01), 3.33333333333329318027e-01);
    const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02),
                                             -7.69187620504482999495e-02), -1.11111104054623557880e-01),
                              -1.99999999998764832476e-01);
    return x - x * (s1 + s2);
}

// sin and cos of x for |x| up to ~1e6 (two-constant Cody-Waite with FMA; fdlibm k_sin / k_cos kernels)
__device__ __forceinline__ void sincos(double x, double *sn, double *cs) {
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.5707963267948966, x);
    r = fma(-n, 6.123233995736766e-17, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                       2.75573137070700676789e-06), -1.98412698298579493134e-04),
                          8.33333333332248946124e-03);
    const double s = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                                -2.75573143513906633035e-07), 2.48015872894767294178e-05),
                                     -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * pc);
    const int q = (int)n & 3;
    const double s_out = (q & 1) ? c : s, c_out = (q & 1) ? s : c;
    *sn = (q & 2) ? -s_out : s_out;
    *cs = ((q + 1) & 2) ? -c_out : c_out;
}

}  // namespace fxm

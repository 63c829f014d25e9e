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

// Constants of the rarely taken paths -- the interval constants of atan's argument reduction and everything sincos needs --
// either as literals (TAB = false) or fetched from constant memory with scalar loads where they are used (TAB = true).  As
// literals the compiler materialises them in front of the walk loop and keeps them in registers across it (loop-invariant code
// motion): 23 doubles = 46 VGPRs, which is what made the three-waves-per-SIMD walk with the obstacle stage spill at 168
// registers.  With TAB the pointer is made opaque inside the path that uses it, so the loads can neither be folded back into
// literals nor hoisted out of it; the price is a scalar-cache round trip whenever such a path runs, which the latency-bound
// small grids (two waves per SIMD, registers to spare) should not pay -- they take the literals.
static __device__ __constant__ double fx_ktab[36] = {
    4.63647609000806093515e-01, 2.26987774529616870924e-17,   //  0: atan(0.5)  hi, lo
    7.85398163397448278999e-01, 3.06161699786838301793e-17,   //  2: atan(1.0)
    9.82793723247329054082e-01, 1.39033110312309984516e-17,   //  4: atan(1.5)
    1.57079632679489655800e+00, 6.12323399573676603587e-17,   //  6: atan(inf)
    6.36619772367581382433e-01, 1.5707963267948966, 6.123233995736766e-17,   // 8: 2/pi, pi/2 hi, pi/2 lo
    1.58969099521155010221e-10, -2.50507602534068634195e-08, 2.75573137070700676789e-06, -1.98412698298579493134e-04,
    8.33333333332248946124e-03, -1.66666666666666324348e-01,   // 11: k_sin S6 .. S1
    -1.13596475577881948265e-11, 2.08757232129817482790e-09, -2.75573143513906633035e-07, 2.48015872894767294178e-05,
    -1.38888888888741095749e-03, 4.16666666666666019037e-02,   // 17: k_cos C6 .. C1
    0.0,
    // 24: atan's odd / even polynomials (fdlibm aT[10], aT[8], .. aT[0] | aT[9], aT[7], .. aT[1]); FX_ATAN_K doubles
    1.62858201153657823623e-02, 4.97687799461593236017e-02, 6.66107313738753120669e-02, 9.09088713343650656196e-02,
    1.42857142725034663711e-01, 3.33333333333329318027e-01,
    -3.65315727442169155270e-02, -5.83357013379057348645e-02, -7.69187620504482999495e-02, -1.11111104054623557880e-01,
    -1.99999999998764832476e-01, 0.0};
#define FX_ATAN_K 12   // doubles of the polynomial block (a kernel may keep a copy in LDS: atan_small_tab)
#define FX_ATAN_K0 24  // its first entry in fx_ktab
typedef const __attribute__((address_space(4))) double *ktab_ptr;
__device__ __forceinline__ ktab_ptr ktab_here() {
    ktab_ptr p = (ktab_ptr)fx_ktab;
    asm volatile("" : "+s"(p));
    return p;
}
// constant k: literal or table entry
#define FX_K(k, lit) (TAB ? kt[k] : (lit))

// atan(x): fdlibm s_atan.c argument reduction (5 intervals) made branch-free with selects + one division.
template <bool TAB = false>
__device__ __forceinline__ double atan(double x) {
    const ktab_ptr kt = TAB ? ktab_here() : nullptr;
    const double ax = fabs(x);
    double num = ax, den = 1.0, hi = 0.0, lo = 0.0;
    if (ax >= 0.4375) {
        if (ax < 0.6875) { num = 2.0 * ax - 1.0; den = 2.0 + ax; hi = FX_K(0, 4.63647609000806093515e-01); lo = FX_K(1, 2.26987774529616870924e-17); }
        else if (ax < 1.1875) { num = ax - 1.0; den = ax + 1.0; hi = FX_K(2, 7.85398163397448278999e-01); lo = FX_K(3, 3.06161699786838301793e-17); }
        else if (ax < 2.4375) { num = ax - 1.5; den = 1.0 + 1.5 * ax; hi = FX_K(4, 9.82793723247329054082e-01); lo = FX_K(5, 1.39033110312309984516e-17); }
        else { num = -1.0; den = ax; hi = FX_K(6, 1.57079632679489655800e+00); lo = FX_K(7, 6.12323399573676603587e-17); }
    }
    const double xr = num / den;
    const double z = xr * xr, w = z * z;
    const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, FX_K(24, 1.62858201153657823623e-02), FX_K(25, 4.97687799461593236017e-02)),
                                                    FX_K(26, 6.66107313738753120669e-02)), FX_K(27, 9.09088713343650656196e-02)),
                                     FX_K(28, 1.42857142725034663711e-01)), FX_K(29, 3.33333333333329318027e-01));
    const double s2 = w * fma(w, fma(w, fma(w, fma(w, FX_K(30, -3.65315727442169155270e-02), FX_K(31, -5.83357013379057348645e-02)),
                                             FX_K(32, -7.69187620504482999495e-02)), FX_K(33, -1.11111104054623557880e-01)),
                              FX_K(34, -1.99999999998764832476e-01));
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

// atan_small with the eleven coefficients read from a table in LDS (kl[0 .. 10] = fx_ktab[FX_ATAN_K0 ..]) each time it runs:
// a loaded coefficient lands in a fresh register that the next v_fmac_f64 accumulates into, where a coefficient kept in a
// register across the walk loop has to be copied first (v_mov_b64 + v_fmac_f64: the Horner addend is not the accumulator) --
// ten copies per step less and 22 VGPRs free.  The address is made opaque so that the loads stay where the values are used.
typedef const __attribute__((address_space(3))) double *lds_cptr;
__device__ __forceinline__ double atan_small_tab(double x, lds_cptr kl) {
    asm volatile("" : "+v"(kl));
    const double z = x * x, w = z * z;
    const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, kl[0], kl[1]), kl[2]), kl[3]), kl[4]), kl[5]);
    const double s2 = w * fma(w, fma(w, fma(w, fma(w, kl[6], kl[7]), kl[8]), kl[9]), kl[10]);
    return x - x * (s1 + s2);
}

// sin and cos of x for |x| up to ~1e6 (two-constant Cody-Waite with FMA; fdlibm k_sin / k_cos kernels)
template <bool TAB = false>
__device__ __forceinline__ void sincos(double x, double *sn, double *cs) {
    const ktab_ptr kt = TAB ? ktab_here() : nullptr;
    const double n = rint(x * FX_K(8, 6.36619772367581382433e-01));
    double r = fma(-n, FX_K(9, 1.5707963267948966), x);
    r = fma(-n, FX_K(10, 6.123233995736766e-17), r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, FX_K(11, 1.58969099521155010221e-10), FX_K(12, -2.50507602534068634195e-08)),
                                       FX_K(13, 2.75573137070700676789e-06)), FX_K(14, -1.98412698298579493134e-04)),
                          FX_K(15, 8.33333333332248946124e-03));
    const double s = fma(z * r, fma(z, ps, FX_K(16, -1.66666666666666324348e-01)), r);
    const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, FX_K(17, -1.13596475577881948265e-11), FX_K(18, 2.08757232129817482790e-09)),
                                                FX_K(19, -2.75573143513906633035e-07)), FX_K(20, 2.48015872894767294178e-05)),
                                     FX_K(21, -1.38888888888741095749e-03)), FX_K(22, 4.16666666666666019037e-02));
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * pc);
    const int q = (int)n & 3;
    const double s_out = (q & 1) ? c : s, c_out = (q & 1) ? s : c;
    *sn = (q & 2) ? -s_out : s_out;
    *cs = ((q + 1) & 2) ? -c_out : c_out;
}

#undef FX_K

}  // namespace fxm

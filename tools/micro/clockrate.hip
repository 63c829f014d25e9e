// Shader clock during short kernels (s_memtime against the 100 MHz wall clock) and the issue cadence of ONE wave per SIMD against four:
// shader cycles per instruction and wave for dependent / independent v_fma_f64, v_add_f64, v_fma_f32 and v_mad_u32.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(unsigned long long *out, double *sink, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5, d = 0.25, e = 0.125, f = 0.3, g = 0.7, h = 0.9;
    float fa = threadIdx.x * 1e-3f, fb = 1.000001f, fc = 0.5f, fd = 0.25f, fe = 0.125f, ff = 0.3f, fg = 0.7f, fh = 0.9f;
    unsigned ia = threadIdx.x, ib = 3, ic = 5, id = 7, ie = 11, ig = 13, ih = 17, ii = 19;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
        if (OP == 1) { a = fma(a, b, c); d = fma(d, b, c); e = fma(e, b, c); f = fma(f, b, c); g = fma(g, b, c); h = fma(h, b, c); a = fma(a, b, d); e = fma(e, b, f); }
        if (OP == 2) { a = a + c; d = d + c; e = e + c; f = f + c; g = g + c; h = h + c; a = a + d; e = e + f; }
        if (OP == 3) { fa = fmaf(fa, fb, fc); fd = fmaf(fd, fb, fc); fe = fmaf(fe, fb, fc); ff = fmaf(ff, fb, fc); fg = fmaf(fg, fb, fc); fh = fmaf(fh, fb, fc); fa = fmaf(fa, fb, fd); fe = fmaf(fe, fb, ff); }
        if (OP == 4) { ia = ia * ib + ic; id = id * ib + ic; ie = ie * ib + ic; ig = ig * ib + ic; ih = ih * ib + ic; ii = ii * ib + ic; ia = ia * ib + id; ie = ie * ib + ig; }
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = w1 - w0; out[2 * blockIdx.x + 1] = c1 - c0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h + fa + fd + fe + ff + fg + fh + (double)(ia + id + ie + ig + ih + ii);
}
template <int OP>
void run(const char *name, unsigned long long *o, double *s) {
    for (int blocks : {1024, 2048, 4096, 8192}) {   // 1, 2, 4, 8 waves per SIMD (64-lane workgroups on 1 024 SIMDs)
        const int iters = 4000;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, o, s, iters);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, o, s, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long h[2 * 8192];
        (void)hipMemcpy(h, o, 16 * blocks, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int b = 0; b < blocks; b++) { wall += h[2 * b] * 0.01; cyc += (double)h[2 * b + 1]; }
        cyc /= blocks; wall /= blocks;
        const double n = 8.0 * iters;
        printf("%-22s %d wave(s)/SIMD: per wave %6.2f cycles = %5.2f ns per instruction (mean of all waves), clock %4.0f MHz; whole kernel %7.1f us = %5.2f ns per wave-instruction and SIMD\n",
               name, blocks / 1024, cyc / n, wall * 1e3 / n, cyc / wall, ms * 1e3, ms * 1e6 / (n * blocks / 1024.0));
    }
}
int main() {
    unsigned long long *o; double *s;
    (void)hipMalloc(&o, 16 * 8192); (void)hipMalloc(&s, 8 * 8192 * 64);
    run<0>("v_fma_f64 dependent", o, s);
    run<1>("v_fma_f64 independent", o, s);
    run<2>("v_add_f64 independent", o, s);
    run<3>("v_fma_f32 independent", o, s);
    run<4>("v_mad_u32 independent", o, s);
    return 0;
}

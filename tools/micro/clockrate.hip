// shader clock during short kernels: s_memtime (clock64) against the 100 MHz wall clock, and the issue rate of dependent / independent FP64 FMAs
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *out, double *sink, int iters, int mode) {
    double a = threadIdx.x * 1e-3, b = 1.000001, c = 0.5, d = 0.25, e = 0.125, f = 0.3, g = 0.7, h = 0.9;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (mode == 0) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
        else { a = fma(a, b, c); d = fma(d, b, c); e = fma(e, b, c); f = fma(f, b, c); g = fma(g, b, c); h = fma(h, b, c); a = fma(a, b, d); e = fma(e, b, f); }
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = w1 - w0; out[2 * blockIdx.x + 1] = c1 - c0; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h;
}
int main() {
    unsigned long long *o; double *s;
    hipMalloc(&o, 16 * 4096); hipMalloc(&s, 8 * 4096 * 256);
    for (int mode = 0; mode < 2; mode++)
        for (int blocks : {1, 256, 1024, 4096})
            for (int iters : {500, 20000}) {
                for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, o, s, iters, mode);
                hipDeviceSynchronize();
                unsigned long long h[2];
                hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
                printf("mode %d blocks %4d iters %5d: wall %.2f us, clock64 %llu ticks -> %.0f MHz; %.2f clock64 ticks per FMA, %.2f ns per FMA\n", mode, blocks, iters,
                       h[0] * 0.01, h[1], h[1] / (h[0] * 0.01), (double)h[1] / (8.0 * iters), h[0] * 10.0 / (8.0 * iters));
            }
    return 0;
}

// FP64 issue-rate microbenchmark: wave-instructions per SIMD-cycle for v_fma_f64, v_rcp_f64, v_rsq_f64, v_mul_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int OP>
__global__ void k(double* out, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-9, b = a * 1.0001, c = a * 1.0002, d = a * 1.0003;
    double e = a * 1.0004, f = a * 1.0005, g = a * 1.0006, h = a * 1.0007;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { a = fma(a, 0.999999, 1e-9); b = fma(b, 0.999999, 1e-9); c = fma(c, 0.999999, 1e-9); d = fma(d, 0.999999, 1e-9);
                       e = fma(e, 0.999999, 1e-9); f = fma(f, 0.999999, 1e-9); g = fma(g, 0.999999, 1e-9); h = fma(h, 0.999999, 1e-9); }
        if (OP == 1) { a = __builtin_amdgcn_rcp(a); b = __builtin_amdgcn_rcp(b); c = __builtin_amdgcn_rcp(c); d = __builtin_amdgcn_rcp(d);
                       e = __builtin_amdgcn_rcp(e); f = __builtin_amdgcn_rcp(f); g = __builtin_amdgcn_rcp(g); h = __builtin_amdgcn_rcp(h); }
        if (OP == 2) { a = __builtin_amdgcn_rsq(a); b = __builtin_amdgcn_rsq(b); c = __builtin_amdgcn_rsq(c); d = __builtin_amdgcn_rsq(d);
                       e = __builtin_amdgcn_rsq(e); f = __builtin_amdgcn_rsq(f); g = __builtin_amdgcn_rsq(g); h = __builtin_amdgcn_rsq(h); }
        if (OP == 3) { a = a * 0.999999; b = b * 0.999999; c = c * 0.999999; d = d * 0.999999; e = e * 0.999999; f = f * 0.999999; g = g * 0.999999; h = h * 0.999999; }
        if (OP == 4) { float x = (float)a, y = (float)b; x = __builtin_amdgcn_rcpf(x); y = __builtin_amdgcn_rcpf(y); a = x; b = y;
                       float z = (float)c, w = (float)d; z = __builtin_amdgcn_rcpf(z); w = __builtin_amdgcn_rcpf(w); c = z; d = w; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
int main() {
    double* d; CK(hipMalloc(&d, 8 * 1024 * 1024 * 8));
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 4096, blocks = 256 * 8, thr = 256;  // 8 WGs x 4 waves per CU = 8 waves per SIMD
    const char* names[] = {"v_fma_f64", "v_rcp_f64", "v_rsq_f64", "v_mul_f64", "cvt+v_rcp_f32+cvt (4 chains)"};
    for (int op = 0; op < 5; op++) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(s);
            if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(thr), 0, 0, d, iters, 1.5);
            if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(thr), 0, 0, d, iters, 1.5);
            if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(thr), 0, 0, d, iters, 1.5);
            if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(thr), 0, 0, d, iters, 1.5);
            if (op == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(thr), 0, 0, d, iters, 1.5);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e); if (ms < best) best = ms;
        }
        const double n_per = op == 4 ? 4.0 : 8.0;
        const double wave_instr = (double)blocks * (thr / 64) * iters * n_per;
        const double per_simd = wave_instr / 1024.0;
        printf("%-32s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles @2.4 GHz)\n", names[op], best,
               best * 1e6 / per_simd, best * 1e6 / per_simd * 2.4);
    }
    return 0;
}

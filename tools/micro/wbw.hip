// write-bandwidth microbenchmark: how fast can MI355X absorb the TrajectoryBundle store pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// same pattern as the walk: lane = candidate, loop over steps, 14 plane stores of 8 B each per step
__global__ void k_planes(double* __restrict__ out, long ld, int S, long C) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= C) return;
    double v = (double)g;
    for (int i = 0; i < S; i++) {
        double* row = out + (long)i * ld + g;
        long ps = (long)S * ld;
#pragma unroll
        for (int p = 0; p < 14; p++) row[p * ps] = v + p;
        v += 1.0;
    }
}
// same, with a dependent FP64 chain of `work` FMAs per step in front of the stores (do stores overlap compute?)
template <bool STORE>
__global__ void k_planes_work(double* __restrict__ out, long ld, int S, long C, int work, double seed) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= C) return;
    double v = (double)g * 1e-9 + seed;
    double acc = 0.0;
    for (int i = 0; i < S; i++) {
        double x = v, y = v * 0.5, z = v * 0.25, w = v * 0.125;   // 4 independent chains
        for (int k = 0; k < work; k++) { x = fma(x, 0.999999, 1e-9); y = fma(y, 0.999998, 2e-9); z = fma(z, 0.999997, 3e-9); w = fma(w, 0.999996, 4e-9); }
        v = (x + y) + (z + w);
        if (STORE) {
            double* row = out + (long)i * ld + g;
            long ps = (long)S * ld;
#pragma unroll
            for (int p = 0; p < 14; p++) row[p * ps] = v + p;
        } else acc += v;
    }
    if (!STORE && acc == 12345.678) out[g] = acc;
}
// nontemporal variant of the plane pattern
__global__ void k_planes_nt(double* __restrict__ out, long ld, int S, long C) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= C) return;
    double v = (double)g;
    for (int i = 0; i < S; i++) {
        double* row = out + (long)i * ld + g;
        long ps = (long)S * ld;
#pragma unroll
        for (int p = 0; p < 14; p++) __builtin_nontemporal_store(v + p, row + p * ps);
        v += 1.0;
    }
}
template <bool NT>
__global__ void k_planes_work2(double* __restrict__ out, long ld, int S, long C, int work, double seed) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= C) return;
    double v = (double)g * 1e-9 + seed;
    for (int i = 0; i < S; i++) {
        double x = v, y = v * 0.5, z = v * 0.25, w = v * 0.125;
        for (int k = 0; k < work; k++) { x = fma(x, 0.999999, 1e-9); y = fma(y, 0.999998, 2e-9); z = fma(z, 0.999997, 3e-9); w = fma(w, 0.999996, 4e-9); }
        v = (x + y) + (z + w);
        double* row = out + (long)i * ld + g;
        long ps = (long)S * ld;
#pragma unroll
        for (int p = 0; p < 14; p++) { if (NT) __builtin_nontemporal_store(v + p, row + p * ps); else row[p * ps] = v + p; }
    }
}
// horizon split over G adjacent lanes: lane = (candidate, part), part walks steps [part*CH, (part+1)*CH)
template <int G>
__global__ void k_planes_split(double* __restrict__ out, long ld, int S, long C) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long g = t / G; int part = (int)(t % G);
    if (g >= C) return;
    const int CH = (S + G - 1) / G;
    const int i0 = part * CH, i1 = min(S, i0 + CH);
    double v = (double)g;
    for (int i = i0; i < i1; i++) {
        double* row = out + (long)i * ld + g;
        long ps = (long)S * ld;
#pragma unroll
        for (int p = 0; p < 14; p++) row[p * ps] = v + p;
        v += 1.0;
    }
}
// tile layout: out[tile][step][plane][64]: every wave streams one contiguous 14*S*512 B region
__global__ void k_tile(double* __restrict__ out, int S, long C) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= C) return;
    const long tile = g >> 6; const int lane = (int)(g & 63);
    double v = (double)g;
    double* base = out + tile * (long)S * 14 * 64 + lane;
    for (int i = 0; i < S; i++) {
        double* row = base + (long)i * 14 * 64;
#pragma unroll
        for (int p = 0; p < 14; p++) row[p * 64] = v + p;
        v += 1.0;
    }
}
// tile layout, plane-major inside the tile: out[tile][plane][step][64]
__global__ void k_tile_pm(double* __restrict__ out, int S, long C) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= C) return;
    const long tile = g >> 6; const int lane = (int)(g & 63);
    double v = (double)g;
    double* base = out + tile * (long)S * 14 * 64 + lane;
    for (int i = 0; i < S; i++) {
        double* row = base + (long)i * 64;
#pragma unroll
        for (int p = 0; p < 14; p++) row[(long)p * S * 64] = v + p;
        v += 1.0;
    }
}
// tile layout with the horizon split over 2 waves of a workgroup (wave w of a pair walks steps [w*CH, (w+1)*CH))
__global__ void k_tile_ws2(double* __restrict__ out, int S, long C) {
    const int CPB = blockDim.x / 2;
    const int part = threadIdx.x / CPB, cl = threadIdx.x - part * CPB;
    long g = (long)blockIdx.x * CPB + cl;
    if (g >= C) return;
    const long tile = g >> 6; const int lane = (int)(g & 63);
    const int CH = (S + 1) / 2, i0 = part * CH, i1 = min(S, i0 + CH);
    double v = (double)g;
    double* base = out + tile * (long)S * 14 * 64 + lane;
    for (int i = i0; i < i1; i++) {
        double* row = base + (long)i * 14 * 64;
#pragma unroll
        for (int p = 0; p < 14; p++) row[p * 64] = v + p;
        v += 1.0;
    }
}
__global__ void k_planes_ws2(double* __restrict__ out, long ld, int S, long C) {
    const int CPB = blockDim.x / 2;
    const int part = threadIdx.x / CPB, cl = threadIdx.x - part * CPB;
    long g = (long)blockIdx.x * CPB + cl;
    if (g >= C) return;
    const int CH = (S + 1) / 2, i0 = part * CH, i1 = min(S, i0 + CH);
    double v = (double)g;
    for (int i = i0; i < i1; i++) {
        double* row = out + (long)i * ld + g;
        long ps = (long)S * ld;
#pragma unroll
        for (int p = 0; p < 14; p++) row[p * ps] = v + p;
        v += 1.0;
    }
}
// plane pattern with write-through stores (agent / system scope): nothing dirty left in L2 at kernel end
template <int SCOPE>
__global__ void k_planes_wt(double* __restrict__ out, long ld, int S, long C) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= C) return;
    double v = (double)g;
    for (int i = 0; i < S; i++) {
        double* row = out + (long)i * ld + g;
        long ps = (long)S * ld;
#pragma unroll
        for (int p = 0; p < 14; p++) __hip_atomic_store(row + p * ps, v + p, __ATOMIC_RELAXED, SCOPE);
        v += 1.0;
    }
}
// streaming: each lane writes 16 B, consecutive, grid-stride
__global__ void k_stream16(double2* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = make_double2(1.0, 2.0);
}
__global__ void k_stream8(double* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = 1.0;
}
int main() {
    const long C = 50388, ld = 50432; const int S = 31;
    const long n = 14L * S * ld;
    double* d; CK(hipMalloc(&d, n * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        float best = 1e9, sum = 0;
        for (int i = 0; i < 20; i++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; sum += ms; }
        printf("%-28s best %.1f us  avg %.1f us  -> %.2f TB/s (best)\n", name, best * 1e3, sum / 20 * 1e3, n * 8 / (best * 1e-3) / 1e12);
    };
    timeit("plane pattern (256 thr)", [&] { hipLaunchKernelGGL(k_planes, dim3((C + 255) / 256), dim3(256), 0, 0, d, ld, S, C); });
    timeit("tile [t][i][p][64] (256 thr)", [&] { hipLaunchKernelGGL(k_tile, dim3((C + 255) / 256), dim3(256), 0, 0, d, S, C); });
    timeit("tile [t][i][p][64] (64 thr)", [&] { hipLaunchKernelGGL(k_tile, dim3((C + 63) / 64), dim3(64), 0, 0, d, S, C); });
    timeit("tile [t][p][i][64] (256 thr)", [&] { hipLaunchKernelGGL(k_tile_pm, dim3((C + 255) / 256), dim3(256), 0, 0, d, S, C); });
    timeit("tile wave-split 2 (256 thr)", [&] { hipLaunchKernelGGL(k_tile_ws2, dim3((C + 127) / 128), dim3(256), 0, 0, d, S, C); });
    timeit("planes wave-split 2 (256 thr)", [&] { hipLaunchKernelGGL(k_planes_ws2, dim3((C + 127) / 128), dim3(256), 0, 0, d, ld, S, C); });
    timeit("planes, agent-scope stores", [&] { hipLaunchKernelGGL(k_planes_wt<__HIP_MEMORY_SCOPE_AGENT>, dim3((C + 255) / 256), dim3(256), 0, 0, d, ld, S, C); });
    timeit("planes, system-scope stores", [&] { hipLaunchKernelGGL(k_planes_wt<__HIP_MEMORY_SCOPE_SYSTEM>, dim3((C + 255) / 256), dim3(256), 0, 0, d, ld, S, C); });
    timeit("plane pattern (64 thr)", [&] { hipLaunchKernelGGL(k_planes, dim3((C + 63) / 64), dim3(64), 0, 0, d, ld, S, C); });
    timeit("plane pattern NT (256 thr)", [&] { hipLaunchKernelGGL(k_planes_nt, dim3((C + 255) / 256), dim3(256), 0, 0, d, ld, S, C); });
    timeit("work=20 c+s plain", [&] { hipLaunchKernelGGL(k_planes_work2<false>, dim3((C + 255) / 256), dim3(256), 0, 0, d, ld, S, C, 20, 1.0); });
    timeit("work=20 c+s NT", [&] { hipLaunchKernelGGL(k_planes_work2<true>, dim3((C + 255) / 256), dim3(256), 0, 0, d, ld, S, C, 20, 1.0); });
    timeit("split G=2", [&] { hipLaunchKernelGGL(k_planes_split<2>, dim3((C * 2 + 255) / 256), dim3(256), 0, 0, d, ld, S, C); });
    timeit("split G=4", [&] { hipLaunchKernelGGL(k_planes_split<4>, dim3((C * 4 + 255) / 256), dim3(256), 0, 0, d, ld, S, C); });
    timeit("split G=8", [&] { hipLaunchKernelGGL(k_planes_split<8>, dim3((C * 8 + 255) / 256), dim3(256), 0, 0, d, ld, S, C); });
    for (int work : {25}) {
        char nm[64];
        snprintf(nm, 64, "work=%d compute only", work);
        timeit(nm, [&] { hipLaunchKernelGGL(k_planes_work<false>, dim3((C + 255) / 256), dim3(256), 0, 0, d, ld, S, C, work, 1.0); });
        snprintf(nm, 64, "work=%d compute+stores", work);
        timeit(nm, [&] { hipLaunchKernelGGL(k_planes_work<true>, dim3((C + 255) / 256), dim3(256), 0, 0, d, ld, S, C, work, 1.0); });
        snprintf(nm, 64, "work=%d c+s 128thr", work);
        timeit(nm, [&] { hipLaunchKernelGGL(k_planes_work<true>, dim3((C + 127) / 128), dim3(128), 0, 0, d, ld, S, C, work, 1.0); });
    }
    timeit("stream 16B x 2048 blocks", [&] { hipLaunchKernelGGL(k_stream16, dim3(2048), dim3(256), 0, 0, (double2*)d, n / 2); });
    timeit("stream 8B x 2048 blocks", [&] { hipLaunchKernelGGL(k_stream8, dim3(2048), dim3(256), 0, 0, d, n); });
    timeit("stream 8B x 197 blocks", [&] { hipLaunchKernelGGL(k_stream8, dim3(197), dim3(256), 0, 0, d, n); });
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, n * 8, 0); });
    return 0;
}

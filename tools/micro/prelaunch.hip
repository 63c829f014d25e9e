// A kernel launched BEFORE its inputs exist: it waits on a flag in device memory that the host writes (posted write through the
// BAR) behind the inputs.  Time from "host starts writing the 20 KB table" to "result visible in pinned host memory", against
// write-then-launch (bar_write.hip).  The waiting kernel gives up after 50 ms by the wall clock.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void consumer(const double *t, int n, const unsigned long long *go, unsigned long long want, double *out, unsigned long long *seq) {
    if (go) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
            if (wall_clock64() - t0 > 5000000ULL) { if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = -1.0; __hip_atomic_store(seq, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } return; }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    double a = 0; for (int i = threadIdx.x; i < n; i += blockDim.x) a += t[i];
    __shared__ double sh[256]; sh[threadIdx.x] = a; __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) { double r = 0; for (int i = 0; i < 256; i++) r += sh[i]; out[0] = r;
        __hip_atomic_store(seq, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
int main() {
    const int N = 2560, WG = 158;
    double *tab, *out_h; unsigned long long *go, *seq;
    hipMalloc((void **)&tab, N * 8); hipMalloc((void **)&go, 64);
    hipHostMalloc((void **)&out_h, 64, hipHostMallocMapped); hipHostMalloc((void **)&seq, 64, hipHostMallocMapped);
    *seq = 0; *go = 0; __builtin_ia32_sfence();
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int mode = 0; mode < 2; mode++) {
        std::vector<double> ts; int wrong = 0;
        for (unsigned long long it = 1 + 1000 * mode; it <= 300 + 1000 * mode; it++) {
            if (mode == 1) {   // launched first; 20 us of "host work" before the inputs are written
                hipLaunchKernelGGL(consumer, WG, 256, 0, st, tab, N, go, it, out_h, seq);
                auto w = std::chrono::steady_clock::now();
                while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w).count() < 20.0) { }
            }
            auto a = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) tab[i] = (double)it;
            __builtin_ia32_sfence();
            if (mode == 0) hipLaunchKernelGGL(consumer, WG, 256, 0, st, tab, N, (const unsigned long long *)nullptr, it, out_h, seq);
            else { *go = it; __builtin_ia32_sfence(); }
            while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) != it) { }
            auto b = std::chrono::steady_clock::now();
            if (out_h[0] != (double)it * N) wrong++;
            ts.push_back(std::chrono::duration<double, std::micro>(b - a).count());
            hipStreamSynchronize(st);
        }
        std::sort(ts.begin(), ts.end());
        printf("%s: inputs written -> result seen p50 %.2f us (min %.2f), wrong results %d\n", mode == 0 ? "write, then launch      " : "launched before, go flag", ts[ts.size() / 2], ts[0], wrong);
    }
    return 0;
}

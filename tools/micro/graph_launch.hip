// Three dependent kernels shaped like a config-3 step (394 x 256 lanes ~40 us | 788 x 640 lanes ~25 us | 32 x 256 lanes, publishes a
// sequence word to pinned host memory) as three stream launches against ONE hipGraphLaunch of the captured triple: wall time from
// the first API call to the host seeing the word.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256, 2) void k1(double *buf, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    for (int i = 0; i < iters; i++) a = fma(a, b, 1e-9);
    buf[(size_t)blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(640, 1) void k2(const double *buf, double *buf2, int n_src, int iters) {
    const int src = (int)(((size_t)blockIdx.x * 256 + (threadIdx.x & 255)) % ((size_t)n_src * 256));
    double a = buf[src], b = 1.0000001;
    for (int i = 0; i < iters; i++) a = fma(a, b, 1e-9);
    if ((threadIdx.x & 255) == threadIdx.x) buf2[(size_t)blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ void k3(const double *buf2, unsigned long long *dev_seq, unsigned long long *tickets, unsigned long long *seq_h, double *out_h) {
    __shared__ double sh[256];
    sh[threadIdx.x] = buf2[(size_t)blockIdx.x * 256 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = atomicAdd(tickets, 1ULL);
        if ((t + 1) % gridDim.x == 0) {   // the last workgroup of this launch publishes
            const unsigned long long s = (t + 1) / gridDim.x;
            out_h[0] = sh[0];
            __hip_atomic_store(seq_h, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    (void)dev_seq;
}
int main() {
    const int N1 = 394, N2 = 788, N3 = 32, it1 = 2100, it2 = 500;
    double *buf, *buf2, *out_h; unsigned long long *tickets, *seq_h;
    CHECK(hipMalloc((void **)&buf, (size_t)N1 * 256 * 8)); CHECK(hipMalloc((void **)&buf2, (size_t)N2 * 256 * 8)); CHECK(hipMalloc((void **)&tickets, 64));
    CHECK(hipMemset(tickets, 0, 64));
    CHECK(hipHostMalloc((void **)&out_h, 64, hipHostMallocMapped)); CHECK(hipHostMalloc((void **)&seq_h, 64, hipHostMallocMapped));
    *seq_h = 0;
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(k1, N1, 256, 0, st, buf, it1);
    hipLaunchKernelGGL(k2, N2, 640, 0, st, buf, buf2, N1, it2);
    hipLaunchKernelGGL(k3, N3, 256, 0, st, buf2, tickets, tickets, seq_h, out_h);
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    unsigned long long seq = 0;
    for (int round = 0; round < 2; round++)
        for (int mode = 0; mode < 2; mode++) {
            std::vector<double> ts;
            for (int rep = 0; rep < 300; rep++) {
                seq++;
                auto a = std::chrono::steady_clock::now();
                if (mode == 0) {
                    hipLaunchKernelGGL(k1, N1, 256, 0, st, buf, it1);
                    hipLaunchKernelGGL(k2, N2, 640, 0, st, buf, buf2, N1, it2);
                    hipLaunchKernelGGL(k3, N3, 256, 0, st, buf2, tickets, tickets, seq_h, out_h);
                } else CHECK(hipGraphLaunch(exec, st));
                while (__atomic_load_n(seq_h, __ATOMIC_ACQUIRE) != seq) {
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() > 2.0) { printf("timeout mode %d\n", mode); return 2; }
                }
                auto b = std::chrono::steady_clock::now();
                ts.push_back(std::chrono::duration<double, std::micro>(b - a).count());
            }
            std::sort(ts.begin(), ts.end());
            printf("%-28s p50 %7.2f us  min %7.2f  p90 %7.2f\n", mode == 0 ? "three stream launches" : "one hipGraphLaunch", ts[ts.size() / 2], ts[0], ts[ts.size() * 9 / 10]);
            fflush(stdout);
        }
    return 0;
}

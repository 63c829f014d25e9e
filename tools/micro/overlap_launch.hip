// Can a dependent kernel B be resident and past its prologue BEFORE its producer A has finished?  A = 394 workgroups x 256 lanes
// busy for ~40 us, raises `done` per workgroup; B = 788 workgroups that need A's output.
//   mode 0  stream order (B launched behind A on the same stream)
//   mode 1  B with hipExtAnyOrderLaunch on the same stream, B's workgroups wait for `done` themselves
//   mode 2  B on a second stream behind hipStreamWaitValue64(`dispatched` >= seq) -- A's LAST workgroup raises `dispatched` at entry,
//           so every workgroup of A is resident before one of B's is placed -- and B's workgroups wait for `done` themselves
// Every device-side wait gives up after 20 ms by the wall clock and reports it.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void producer(double *buf, int iters, unsigned long long *dispatched, unsigned long long *done, unsigned long long seq) {
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(dispatched, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    for (int i = 0; i < iters; i++) a = fma(a, b, 1e-9);
    __hip_atomic_store(buf + (size_t)blockIdx.x * 256 + threadIdx.x, a + (double)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned long long s_last;
    if (threadIdx.x == 0) { s_last = __hip_atomic_fetch_add(done, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicMax(done + 48, wall_clock64()); atomicMin(done + 80, wall_clock64()); }
    __syncthreads();
    // the last workgroup to arrive raises 64 release flags on 64 cache lines (the waiting workgroups poll those, not the counter)
    if (s_last == (unsigned long long)gridDim.x * seq - 1 && threadIdx.x < 64)
        __hip_atomic_store(done + 128 + 16 * threadIdx.x, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(640, 1) void consumer(const double *buf, int n_src, int iters, const unsigned long long *done, unsigned long long want,
                                                    unsigned long long *timeouts, double *out_h, unsigned long long *seq_h, unsigned long long *fin,
                                                    unsigned long long seq, int self_wait, unsigned long long fin_last) {
    __shared__ double sh[64];
    if (threadIdx.x == 0) atomicMin(const_cast<unsigned long long *>(done) + 64, wall_clock64());
    if (threadIdx.x < 64) sh[threadIdx.x] = threadIdx.x;   // "prologue"
    if (self_wait) {
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            const unsigned long long *rel = done + 128 + 16 * (blockIdx.x & 63);
            while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
                if (wall_clock64() - t0 > 2000000ULL) { atomicAdd(timeouts, 1ULL); break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __syncthreads();
    }
    const int src = (int)(((size_t)blockIdx.x * 256 + (threadIdx.x & 255)) % ((size_t)n_src * 256));
    double a = __hip_atomic_load(buf + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = 1.0000001;
    for (int i = 0; i < iters; i++) a = fma(a, b, 1e-9);
    if (a == 12345.678) sh[0] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = __hip_atomic_fetch_add(fin, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == fin_last) {
            out_h[0] = a - (double)seq;
            __hip_atomic_store(seq_h, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
int main() {
    const int NA = 394, NB = 788;
    double *buf, *out_h; unsigned long long *flags, *seq_h;
    CHECK(hipMalloc((void **)&buf, (size_t)NA * 256 * 8)); CHECK(hipMalloc((void **)&flags, 65536));
    CHECK(hipMemset(flags, 0, 65536));
    CHECK(hipHostMalloc((void **)&out_h, 64, hipHostMallocMapped)); CHECK(hipHostMalloc((void **)&seq_h, 64, hipHostMallocMapped));
    *seq_h = 0;
    unsigned long long *dispatched = flags, *done = flags + 16, *timeouts = flags + 32, *fin = flags + 48;
    hipStream_t s1, s2; CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CHECK(hipDeviceSynchronize());
    // calibrate: iterations for ~40 us (A) and ~25 us (B)
    int itA = 2100, itB = 500;
    unsigned long long b_runs = 0;
    unsigned long long seq = 0;
    for (int mode = -2; mode < 3; mode++) {
        std::vector<double> ts, gaps, adur;
        for (int rep = 0; rep < 200; rep++) {
            seq++;
            { unsigned long long z[3] = {0ULL, ~0ULL, ~0ULL}; CHECK(hipMemcpy(done + 48, &z[0], 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(done + 64, &z[1], 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(done + 80, &z[2], 8, hipMemcpyHostToDevice)); }
            auto a = std::chrono::steady_clock::now();
            const unsigned long long want = (unsigned long long)NA * seq;
            if (mode != -2) b_runs++;
            const unsigned long long fin_last = (unsigned long long)NB * b_runs - 1;
            if (mode == -2) {   // A alone
                hipLaunchKernelGGL(producer, NA, 256, 0, s1, buf, itA, dispatched, done, seq);
                CHECK(hipStreamSynchronize(s1));
            } else if (mode == -1) {   // B alone
                hipLaunchKernelGGL(producer, 1, 64, 0, s1, buf, 1, dispatched, done, seq);   // keeps `done`'s arithmetic out of the way
                CHECK(hipMemsetAsync(done, 0, 8, s1));
                hipLaunchKernelGGL(consumer, NB, 640, 0, s1, buf, NA, itB, done, 0ULL, timeouts, out_h, seq_h, fin, seq, 0, fin_last);
                CHECK(hipStreamSynchronize(s1));
                unsigned long long z = (unsigned long long)NA * seq; CHECK(hipMemcpy(done, &z, 8, hipMemcpyHostToDevice));
            } else {
                hipLaunchKernelGGL(producer, NA, 256, 0, s1, buf, itA, dispatched, done, seq);
                if (mode == 0) hipLaunchKernelGGL(consumer, NB, 640, 0, s1, buf, NA, itB, done, want, timeouts, out_h, seq_h, fin, seq, 0, fin_last);
                if (mode == 1) hipExtLaunchKernelGGL(consumer, NB, 640, 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, buf, NA, itB, done, want, timeouts, out_h, seq_h, fin, seq, 1, fin_last);
                if (mode == 2) {
                    CHECK(hipStreamWaitValue64(s2, dispatched, seq, hipStreamWaitValueGte, ~0ULL));
                    hipLaunchKernelGGL(consumer, NB, 640, 0, s2, buf, NA, itB, done, want, timeouts, out_h, seq_h, fin, seq, 1, fin_last);
                }
                while (__atomic_load_n(seq_h, __ATOMIC_ACQUIRE) != seq) {
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() > 2.0) { printf("host timeout mode %d\n", mode); return 2; }
                }
            }
            auto b = std::chrono::steady_clock::now();
            ts.push_back(std::chrono::duration<double, std::micro>(b - a).count());
            CHECK(hipStreamSynchronize(s1)); CHECK(hipStreamSynchronize(s2));
            if (mode >= 0) { unsigned long long ae, bs, a1; CHECK(hipMemcpy(&ae, done + 48, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&bs, done + 64, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&a1, done + 80, 8, hipMemcpyDeviceToHost));
                gaps.push_back(((double)bs - (double)ae) * 0.01); adur.push_back(((double)ae - (double)a1) * 0.01); }
        }
        std::sort(ts.begin(), ts.end());
        unsigned long long to = 0; CHECK(hipMemcpy(&to, timeouts, 8, hipMemcpyDeviceToHost));
        const char *names[5] = {"A alone (launch + sync)", "B alone (launch + sync)", "stream order A -> B", "B any-order, waits itself", "B on stream 2 behind WaitValue64, waits itself"};
        printf("%-48s p50 %7.2f us  min %7.2f  p90 %7.2f   device-side timeouts so far %llu\n", names[mode + 2], ts[ts.size() / 2], ts[0], ts[ts.size() * 9 / 10], to);
        if (!gaps.empty()) { std::sort(gaps.begin(), gaps.end()); std::sort(adur.begin(), adur.end()); printf("      first B workgroup enters %.2f us after the last A workgroup ends (p50; negative: before); A first end -> last end %.2f us\n", gaps[gaps.size() / 2], adur[adur.size() / 2]); }
        fflush(stdout);
    }
    return 0;
}

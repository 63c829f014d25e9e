// Host -> device-memory write bandwidth through the BAR by size (memcpy from a cached host buffer + sfence), against the
// staging kernel's pull of the same bytes from pinned host memory (launch + completion poll).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
__global__ void stage_kernel(const uint4 *src, uint4 *dst, int n, unsigned long long *seq, unsigned long long s) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[i];
    if (i == 0) __hip_atomic_store(seq, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (not a completion proof: timing only)
}
int main() {
    const size_t MAX = 1 << 20;
    char *dev = nullptr, *pinned = nullptr, *src = (char *)aligned_alloc(64, MAX); unsigned long long *seq = nullptr;
    hipMalloc((void **)&dev, MAX); hipHostMalloc((void **)&pinned, MAX, hipHostMallocMapped); hipHostMalloc((void **)&seq, 64, hipHostMallocMapped);
    memset(src, 1, MAX); memset(pinned, 2, MAX); *seq = 0;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned long long it = 0;
    for (size_t sz : {1024, 4096, 16384, 32768, 65536, 131072, 262144, 1048576}) {
        std::vector<double> t1, t2;
        for (int r = 0; r < 200; r++) {
            auto a = std::chrono::steady_clock::now();
            memcpy(dev, src, sz); __builtin_ia32_sfence();
            auto b = std::chrono::steady_clock::now();
            t1.push_back(std::chrono::duration<double, std::micro>(b - a).count());
            ++it;
            a = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(stage_kernel, (int)((sz / 16 + 255) / 256), 256, 0, st, (const uint4 *)pinned, (uint4 *)dev, (int)(sz / 16), seq, it);
            b = std::chrono::steady_clock::now();
            hipStreamSynchronize(st);
            auto c = std::chrono::steady_clock::now();
            t2.push_back(std::chrono::duration<double, std::micro>(b - a).count());
            (void)c;
        }
        std::sort(t1.begin(), t1.end()); std::sort(t2.begin(), t2.end());
        printf("%8zu B: host memcpy into device memory p50 %7.2f us (%.1f GB/s)   staging-kernel launch call p50 %5.2f us\n", sz, t1[100], sz / t1[100] * 1e-3, t2[100]);
    }
    return 0;
}

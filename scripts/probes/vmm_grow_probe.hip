// can a reserved address range be backed chunk by chunk WHILE a kernel uses the part already mapped?
// The kernel polls a limit in pinned host memory and touches every 4 KB page below it; a host thread creates + maps 1 GB at a time.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void walker(char *base, const unsigned long long *limit, unsigned long long target, unsigned long long *seen, unsigned long long *polls) {
    // one workgroup per 1 GB: waits until its GB is mapped, then writes one byte per page of it
    const unsigned long long lo = (unsigned long long)blockIdx.x << 30, hi = lo + (1ull << 30);
    unsigned long long lim = 0, n = 0;
    while (true) {
        lim = __hip_atomic_load(limit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ++n;
        if (lim >= hi || n > (1ull << 24)) break;
        __builtin_amdgcn_s_sleep(127);
    }
    if (lim >= hi)
        for (unsigned long long o = lo + (unsigned long long)threadIdx.x * 4096; o < hi; o += (unsigned long long)blockDim.x * 4096) base[o] = (char)(blockIdx.x + 1);
    if (threadIdx.x == 0) { atomicMax(seen, lim); atomicAdd(polls, n); }
}

int main() {
    CK(hipSetDevice(0));
    CK(hipFree(0));
    const size_t chunk = 1ull << 30, total = 48 * chunk;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, 1 << 21, nullptr, 0));
    unsigned long long *limit = nullptr;
    CK(hipHostMalloc((void **)&limit, 64, hipHostMallocMapped));
    *limit = 0;
    unsigned long long *d_seen = nullptr;
    CK(hipMalloc((void **)&d_seen, 16));
    CK(hipMemset(d_seen, 0, 16));
    std::vector<hipMemGenericAllocationHandle_t> hs;
    std::atomic<int> failed{0};
    double t0 = now();
    walker<<<(unsigned)(total / chunk), 256>>>((char *)va, limit, total, d_seen, d_seen + 1);
    std::thread mapper([&] {
        (void)hipSetDevice(0);
        for (size_t off = 0; off < total; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { failed = 1; break; }
            if (hipMemMap((char *)va + off, chunk, 0, h, 0) != hipSuccess) { failed = 2; break; }
            if (hipMemSetAccess((char *)va + off, chunk, &acc, 1) != hipSuccess) { failed = 3; break; }
            hs.push_back(h);
            __atomic_store_n(limit, (unsigned long long)(off + chunk), __ATOMIC_RELEASE);
        }
    });
    mapper.join();
    double t1 = now();
    CK(hipDeviceSynchronize());
    double t2 = now();
    unsigned long long h_seen[2] = {0, 0};
    CK(hipMemcpy(h_seen, d_seen, 16, hipMemcpyDeviceToHost));
    int bad = 0;
    for (size_t g = 0; g < total / chunk; ++g) {
        char c = 0;
        CK(hipMemcpy(&c, (char *)va + g * chunk + 4096 * 777, 1, hipMemcpyDeviceToHost));
        bad += c != (char)(g + 1);
    }
    printf("mapped %zu GB in %.3f s while the kernel ran (failed=%d); kernel done %.3f s later; limit seen %.1f GB, %llu polls; %d of %zu GB wrong\n",
           total >> 30, t1 - t0, failed.load(), t2 - t1, h_seen[0] / 1073741824.0, h_seen[1], bad, total >> 30);
    return 0;
}

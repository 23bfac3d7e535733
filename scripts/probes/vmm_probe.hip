// how long does device memory take to become usable?  hipMalloc vs reserve + create + map (virtual memory management API).
// hipcc --offload-arch=gfx950 -O2 scripts/probes/vmm_probe.hip -o /tmp/vmm_probe && /tmp/vmm_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
int main() {
    CK(hipSetDevice(0));
    CK(hipFree(0));
    for (size_t gb : {1, 8, 12, 16, 24, 32, 64}) {
        void *p = nullptr;
        double t = now();
        CK(hipMalloc(&p, gb << 30));
        double t1 = now();
        touch<<<(unsigned)(((gb << 30) / 4096 + 255) / 256), 256>>>((char *)p, gb << 30, 4096);
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipFree(p));
        printf("hipMalloc %2zu GB: %.4f s (%.2f ms/GB), first touch %.4f s, free %.4f s\n", gb, t1 - t, (t1 - t) * 1e3 / gb, t2 - t1, now() - t2);
    }
    {   // many separate 8 GB allocations
        std::vector<void *> ps;
        double t = now();
        for (int i = 0; i < 20; ++i) { void *p = nullptr; CK(hipMalloc(&p, 8ull << 30)); ps.push_back(p); }
        printf("20 x hipMalloc 8 GB: %.4f s\n", now() - t);
        t = now();
        for (void *p : ps) CK(hipFree(p));
        printf("20 x hipFree: %.4f s\n", now() - t);
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    const size_t total = 160ull << 30, chunk = 1ull << 30;
    void *va = nullptr;
    double t = now();
    CK(hipMemAddressReserve(&va, total, gran, nullptr, 0));
    printf("reserve 160 GB: %.6f s\n", now() - t);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    double tc = 0, tm = 0, ta = 0;
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        double a = now();
        CK(hipMemCreate(&h, chunk, &prop, 0));
        double b = now();
        CK(hipMemMap((char *)va + off, chunk, 0, h, 0));
        double c = now();
        CK(hipMemSetAccess((char *)va + off, chunk, &acc, 1));
        double d = now();
        tc += b - a; tm += c - b; ta += d - c;
        hs.push_back(h);
    }
    printf("160 x 1 GB: create %.4f s, map %.4f s, set access %.4f s (%.3f ms/GB in all)\n", tc, tm, ta, (tc + tm + ta) * 1e3 / 160);
    t = now();
    touch<<<(unsigned)((total / 4096 + 255) / 256), 256>>>((char *)va, total, 4096);
    CK(hipDeviceSynchronize());
    printf("first touch of the mapped range: %.4f s\n", now() - t);
    {   // the bytes are really there: fill, read back one byte per GB
        CK(hipMemset(va, 0x5a, total));
        CK(hipDeviceSynchronize());
        int bad = 0;
        for (size_t off = 0; off < total; off += chunk) { unsigned char c = 0; CK(hipMemcpy(&c, (char *)va + off + 12345, 1, hipMemcpyDeviceToHost)); bad += c != 0x5a; }
        size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
        printf("fill + check: %d bad of %zu; free memory now %.1f of %.1f GB\n", bad, total / chunk, fr / 1e9, tot / 1e9);
    }
    t = now();
    for (size_t i = 0; i < hs.size(); ++i) { CK(hipMemUnmap((char *)va + i * chunk, chunk)); CK(hipMemRelease(hs[i])); }
    CK(hipMemAddressFree(va, total));
    printf("unmap + release + free the range: %.4f s\n", now() - t);
    {   void *p = nullptr; t = now(); CK(hipMalloc(&p, 64ull << 30)); printf("hipMalloc 64 GB again: %.4f s\n", now() - t); CK(hipFree(p)); }
    // map while a kernel runs on the part already mapped?
    return 0;
}

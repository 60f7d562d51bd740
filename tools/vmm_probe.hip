// GPU box: what the HIP virtual-memory calls accept (granularity, growth in steps, access after growth)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void touch(unsigned long long *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = i; }
static hipMemAllocationProp prop{};
static hipMemAccessDesc acc{};
// map chunks of the given sizes back to back from offset `start`; returns how many succeeded
static void trial(const char *name, size_t start, const std::vector<size_t> &sizes, bool touch_all) {
    void *base = nullptr; const size_t res = (size_t)64 << 30;
    if (hipMemAddressReserve(&base, res, 0, nullptr, 0) != hipSuccess) { printf("%s: reserve failed\n", name); return; }
    size_t off = start, ok = 0; std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> hs; const char *why = "";
    for (size_t st : sizes) {
        hipMemGenericAllocationHandle_t h; hipError_t e;
        if ((e = hipMemCreate(&h, st, &prop, 0)) != hipSuccess) { why = "create"; (void)hipGetLastError(); break; }
        if ((e = hipMemMap((char *)base + off, st, 0, h, 0)) != hipSuccess) { why = "map"; (void)hipGetLastError(); (void)hipMemRelease(h); break; }
        if ((e = hipMemSetAccess((char *)base + off, st, &acc, 1)) != hipSuccess) { why = "setaccess"; (void)hipGetLastError(); (void)hipMemUnmap((char *)base + off, st); (void)hipMemRelease(h); break; }
        hs.push_back({h, st}); off += st; ++ok;
    }
    bool touched = true;
    if (touch_all && off > start) {
        touch<<<(unsigned)(((off - start) / 8 + 255) / 256), 256>>>((unsigned long long *)((char *)base + start), (off - start) / 8);
        touched = hipDeviceSynchronize() == hipSuccess;
    }
    printf("%-40s %zu of %zu chunks ok (stopped at offset %zu MiB: %s) kernel over the range: %s\n", name, ok, sizes.size(), off >> 20, why, touched ? "ok" : "FAILED");
    size_t o = start;
    for (auto &p : hs) { (void)hipMemUnmap((char *)base + o, p.second); (void)hipMemRelease(p.first); o += p.second; }
    (void)hipMemAddressFree(base, res);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t M = (size_t)1 << 20, G = (size_t)1 << 30;
    trial("2 MiB x 600", 0, std::vector<size_t>(600, 2 * M), true);
    trial("64 MiB x 40", 0, std::vector<size_t>(40, 64 * M), true);
    trial("1 GiB x 8", 0, std::vector<size_t>(8, G), true);
    trial("1 GiB then 66 MiB then 1 GiB", 0, {G, 66 * M, G}, true);
    trial("2,4,8,..512 MiB (aligned to own size?)", 2 * M, {2 * M, 4 * M, 8 * M, 16 * M, 32 * M, 64 * M, 128 * M, 256 * M, 512 * M}, true);
    trial("642 MiB then 66 MiB", 0, {642 * M, 66 * M}, true);
    trial("6 MiB x 50", 0, std::vector<size_t>(50, 6 * M), true);
    trial("8 GiB one handle", 0, {8 * G}, true);
    trial("4 KiB, 4 KiB, 128 KiB", 0, {4096, 4096, 131072}, false);
    trial("128 KiB at 128 KiB", 131072, {131072, 131072}, false);
    return 0;
}

// GPU box: is a virtual range that was unmapped safe to map again? (smx_trim gives chunks back and the arena maps the same addresses
// later.) Map N chunks, fill them by a kernel, unmap + release them, let hipMalloc take the physical memory and fill THAT with another
// pattern, map fresh chunks at the same addresses, write a third pattern through them, and check both sides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void fill(unsigned long long *p, size_t n, unsigned long long v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + i;
}
__global__ void check(const unsigned long long *p, size_t n, unsigned long long v, unsigned long long *bad) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != v + i;
    if (c) atomicAdd(bad, c);
}
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t gran = (size_t)512 << 20, nch = argc > 1 ? (size_t)atoll(argv[1]) : 64;  // 32 GiB by default
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc{};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, nch * gran, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
    std::vector<hipMemGenericAllocationHandle_t> h(nch);
    auto map_all = [&]() {
        for (size_t i = 0; i < nch; ++i) {
            if (hipMemCreate(&h[i], gran, &prop, 0) != hipSuccess) { printf("create %zu failed\n", i); return false; }
            if (hipMemMap((char *)base + i * gran, gran, 0, h[i], 0) != hipSuccess) { printf("map %zu failed\n", i); return false; }
            if (hipMemSetAccess((char *)base + i * gran, gran, &acc, 1) != hipSuccess) { printf("access %zu failed\n", i); return false; }
        }
        return true;
    };
    unsigned long long *bad;
    (void)hipMalloc(&bad, 8);
    (void)hipMemset(bad, 0, 8);
    const size_t n = nch * gran / 8;
    if (!map_all()) return 1;
    fill<<<4096, 256>>>((unsigned long long *)base, n, 0x1111000000000000ull);
    printf("first mapping filled: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    for (size_t i = 0; i < nch; ++i) { (void)hipMemUnmap((char *)base + i * gran, gran); (void)hipMemRelease(h[i]); }
    // somebody else takes the physical memory
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const size_t other_bytes = free_b > (size_t)24 << 30 ? free_b - ((size_t)16 << 30) - nch * gran / 2 : nch * gran;
    unsigned long long *other = nullptr;
    if (hipMalloc(&other, other_bytes) != hipSuccess) { printf("hipMalloc of %zu MiB failed\n", other_bytes >> 20); return 1; }
    fill<<<4096, 256>>>(other, other_bytes / 8, 0x2222000000000000ull);
    printf("hipMalloc of %zu MiB filled: %s\n", other_bytes >> 20, hipGetErrorString(hipDeviceSynchronize()));
    // the same addresses again, backed by new physical memory (what is left: half of the range)
    const size_t nch2 = nch / 2, n2 = nch2 * gran / 8;
    for (size_t i = 0; i < nch2; ++i) {
        if (hipMemCreate(&h[i], gran, &prop, 0) != hipSuccess || hipMemMap((char *)base + i * gran, gran, 0, h[i], 0) != hipSuccess ||
            hipMemSetAccess((char *)base + i * gran, gran, &acc, 1) != hipSuccess) { printf("second mapping of chunk %zu failed\n", i); return 1; }
    }
    fill<<<4096, 256>>>((unsigned long long *)base, n2, 0x3333000000000000ull);
    printf("second mapping filled: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    unsigned long long hb = 0;
    check<<<4096, 256>>>(other, other_bytes / 8, 0x2222000000000000ull, bad);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    printf("words of the hipMalloc buffer damaged by writes through the re-mapped range: %llu\n", hb);
    (void)hipMemset(bad, 0, 8);
    check<<<4096, 256>>>((unsigned long long *)base, n2, 0x3333000000000000ull, bad);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    printf("words of the re-mapped range that do not read back: %llu\n", hb);
    // the runtime's copies on the re-mapped range (the library reads counters back with hipMemcpy and takes caller memory with D2D copies)
    const size_t probe_words = (size_t)1 << 20;
    std::vector<unsigned long long> host(probe_words);
    size_t wrong_d2h = 0, wrong_kernel_after_h2d = 0, wrong_kernel_after_d2d = 0;
    for (size_t c = 0; c < nch2; c += (nch2 > 8 ? nch2 / 8 : 1)) {
        unsigned long long *at = (unsigned long long *)((char *)base + c * gran);
        const size_t w0 = c * gran / 8;
        hipError_t e = hipMemcpy(host.data(), at, probe_words * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) printf("D2H from chunk %zu: %s\n", c, hipGetErrorString(e));
        for (size_t i = 0; i < probe_words; ++i) wrong_d2h += host[i] != 0x3333000000000000ull + w0 + i;
        for (size_t i = 0; i < probe_words; ++i) host[i] = 0x4444000000000000ull + i;
        e = hipMemcpy(at, host.data(), probe_words * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) printf("H2D into chunk %zu: %s\n", c, hipGetErrorString(e));
        (void)hipMemset(bad, 0, 8);
        check<<<256, 256>>>(at, probe_words, 0x4444000000000000ull, bad);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        wrong_kernel_after_h2d += hb;
        e = hipMemcpy(at, other, probe_words * 8, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) printf("D2D into chunk %zu: %s\n", c, hipGetErrorString(e));
        (void)hipMemset(bad, 0, 8);
        check<<<256, 256>>>(at, probe_words, 0x2222000000000000ull, bad);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        wrong_kernel_after_d2d += hb;
    }
    printf("hipMemcpy on the re-mapped range: D2H wrong words %zu, H2D not seen by a kernel %zu, D2D not seen by a kernel %zu\n", wrong_d2h, wrong_kernel_after_h2d,
           wrong_kernel_after_d2d);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    {   // hipMemset / hipMemsetAsync on the re-mapped range (the library zeroes its counters this way), small and large, checked by a kernel
        size_t wrong_small = 0, wrong_large = 0, wrong_async = 0;
        for (size_t c = 0; c < nch2; c += (nch2 > 8 ? nch2 / 8 : 1)) {
            unsigned long long *at = (unsigned long long *)((char *)base + c * gran) + 4096;
            fill<<<256, 256>>>(at, probe_words, 0x7777000000000000ull);
            (void)hipDeviceSynchronize();
            (void)hipMemset(at, 0, 64);  // small: 8 words
            (void)hipMemset(bad, 0, 8);
            check<<<1, 64>>>(at, 8, 0ull - 0, bad);  // expects word i == i: only word 0 matches a zero fill, so count by hand below
            (void)hipDeviceSynchronize();
            unsigned long long h8[8];
            (void)hipMemcpy(h8, at, 64, hipMemcpyDeviceToHost);
            for (int i = 0; i < 8; ++i) wrong_small += h8[i] != 0;
            (void)hipMemset(at + 1024, 0xFF, (probe_words - 1024) * 8);  // large
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(host.data(), at + 1024, (probe_words - 1024) * 8, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < probe_words - 1024; ++i) wrong_large += host[i] != ~0ull;
            fill<<<256, 256, 0, st>>>(at, probe_words, 0x8888000000000000ull);
            (void)hipMemsetAsync(at, 0, 4096, st);
            (void)hipMemsetAsync(at + 8192, 0xFF, 1 << 20, st);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(host.data(), at, probe_words * 8, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < 512; ++i) wrong_async += host[i] != 0;
            for (size_t i = 8192; i < 8192 + (1 << 17); ++i) wrong_async += host[i] != ~0ull;
            for (size_t i = 512; i < 8192; ++i) wrong_async += host[i] != 0x8888000000000000ull + i;
        }
        printf("hipMemset on the re-mapped range: small fill wrong words %zu, large fill wrong words %zu, async fills after a kernel wrong words %zu\n", wrong_small,
               wrong_large, wrong_async);
    }
    unsigned long long *pinned = nullptr;
    (void)hipHostMalloc(&pinned, probe_words * 8, 0);
    size_t wrong_async = 0;
    fill<<<256, 256, 0, st>>>((unsigned long long *)base, probe_words, 0x5555000000000000ull);
    (void)hipMemcpyAsync(pinned, base, probe_words * 8, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    for (size_t i = 0; i < probe_words; ++i) wrong_async += pinned[i] != 0x5555000000000000ull + i;
    printf("hipMemcpyAsync D2H (pinned, non-blocking stream) after a kernel on the same stream: wrong words %zu\n", wrong_async);
    return 0;
}

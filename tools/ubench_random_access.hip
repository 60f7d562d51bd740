// tools/ubench_random_access.hip — what bounds the lookups that leave their chunk (k_pm_remote, k_pm_walk_len, k_pm_walk_write: DESIGN §6/§7)?
// Independent random reads over a working set of W GB: G groups/s by bytes per group (8 B per lane, or 16 B per lane with 1/2/4/8 lanes on one
// aligned 16/32/64/128-B block), by working-set size and by how the memory was obtained (hipMalloc, or the arena's way: reserved address range
// backed by physical chunks of C MiB). Under rocprofv3 --pmc FETCH_SIZE the same run calibrates the counter for these access shapes (the guide's x2
// correction is for wide streaming reads only).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_random_access.hip -o tools/ab/ubench_random_access
//   ubench_random_access <malloc|vmm> <chunk MiB> <W GiB> [only_shape]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

constexpr int ITER = 32, UNR = 4;

// LPG lanes share one aligned block of 16 * LPG bytes; LPG == 0: every lane reads 8 bytes of its own
template <int LPG>
__global__ void __launch_bounds__(256) k_rand(const char *base, uint64_t nblocks /* of 16 * max(LPG, 1) bytes (8 for LPG 0) */, uint64_t seed, unsigned long long *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t grp = LPG > 1 ? tid / LPG : tid;
    const unsigned sub = LPG > 1 ? (unsigned)(tid % LPG) : 0;
    unsigned long long acc = 0;
    for (int it = 0; it < ITER; it += UNR) {
        uint64_t v[UNR][2];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const uint64_t b = mix(grp * ITER + it + u + seed) % nblocks;
            if (LPG == 0) {
                v[u][0] = *(const uint64_t *)(base + b * 8);
                v[u][1] = 0;
            } else {
                const ulonglong2 t = *(const ulonglong2 *)(base + (b * LPG + sub) * 16);
                v[u][0] = t.x;
                v[u][1] = t.y;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u][0] ^ v[u][1];
    }
    if (acc == 0x1234567887654321ull) *sink = acc;
}

template <int LPG>
static void run(const char *name, const char *base, uint64_t W, unsigned long long *sink) {
    const uint64_t blk = LPG == 0 ? 8 : 16 * (uint64_t)LPG;
    const uint64_t nblocks = W / blk;
    const unsigned grid = 256 * 64;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_rand<LPG>, dim3(grid), dim3(256), 0, 0, base, nblocks, 1ull, sink);
    CK(hipDeviceSynchronize());
    const int reps = 3;
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_rand<LPG>, dim3(grid), dim3(256), 0, 0, base, nblocks, 1000003ull * (r + 2), sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double groups = (double)grid * 256 / (LPG > 1 ? LPG : 1) * ITER * reps;
    const double gps = groups / (ms * 1e-3) / 1e9;
    printf("  %-28s %7.2f G groups/s  = %6.0f GB/s of asked bytes, %6.0f GB/s of 64-B sectors, %6.0f GB/s of 128-B lines  (%.2f ms / launch)\n", name, gps, gps * blk,
           gps * (blk > 64 ? blk : 64), gps * 128, ms / reps);
    fflush(stdout);
}

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <malloc|vmm> <chunk MiB> <W GiB> [shape 0|1|2|4|8]\n", argv[0]);
        return 2;
    }
    const bool vmm = !strcmp(argv[1], "vmm");
    const size_t chunk = (size_t)atoll(argv[2]) << 20;
    const double wg = atof(argv[3]);
    const int only = argc > 4 ? atoi(argv[4]) : -1;
    size_t W = (size_t)(wg * (double)(1ull << 30));
    char *base = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    if (vmm) {
        W = (W + chunk - 1) / chunk * chunk;
        void *p = nullptr;
        CK(hipMemAddressReserve(&p, W, 0, nullptr, 0));
        base = (char *)p;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        for (size_t o = 0; o < W; o += chunk) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap(base + o, chunk, 0, h, 0));
            CK(hipMemSetAccess(base + o, chunk, &acc, 1));
            hs.push_back(h);
        }
    } else {
        CK(hipMalloc((void **)&base, W));
    }
    CK(hipMemset(base, 1, W));
    unsigned long long *sink;
    CK(hipMalloc((void **)&sink, 8));
    CK(hipDeviceSynchronize());
    printf("%s chunk %zu MiB, working set %.2f GiB, base %p\n", vmm ? "vmm" : "hipMalloc", chunk >> 20, (double)W / (double)(1ull << 30), (void *)base);
    if (only < 0 || only == 0) run<0>("8 B per lane", base, W, sink);
    if (only < 0 || only == 1) run<1>("16 B per lane", base, W, sink);
    if (only < 0 || only == 2) run<2>("32 B by 2 lanes", base, W, sink);
    if (only < 0 || only == 4) run<4>("64 B by 4 lanes", base, W, sink);
    if (only < 0 || only == 8) run<8>("128 B by 8 lanes", base, W, sink);
    return 0;
}

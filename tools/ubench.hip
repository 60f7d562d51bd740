// tools/ubench.hip — micro-benchmarks that size the partition design on MI355X.
//   copy      : 16 B/lane streaming copy (HBM roofline reference)
//   atomic    : returning device-scope atomicAdd on random counters (N counters), ops/s
//   runs      : scattered writes of R consecutive 16-B records to random 16-B-aligned run starts
//   st4       : scattered 4-B plain stores / loads (per-(tile,bin) offset traffic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__device__ __forceinline__ uint64_t mix(uint64_t x){ x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n){
  for (size_t i = (size_t)blockIdx.x*blockDim.x+threadIdx.x; i<n; i += (size_t)gridDim.x*blockDim.x) b[i]=a[i];
}
__global__ void k_atomic(unsigned long long* c, uint64_t ncounters, int per, unsigned long long* sink){
  uint64_t t = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x; unsigned long long acc=0;
  for (int i=0;i<per;++i){ uint64_t a = mix(t*per+i) % ncounters; acc += atomicAdd(&c[a], 1ull); }
  if (acc==0xdeadbeef) *sink=acc;
}
__global__ void k_atomic_noret(unsigned long long* c, uint64_t ncounters, int per){
  uint64_t t = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x;
  for (int i=0;i<per;++i){ uint64_t a = mix(t*per+i) % ncounters; atomicAdd(&c[a], 1ull); }
}
// each group of R lanes writes R consecutive 16-B records at a random run start
__global__ void k_runs(uint4* out, size_t nrec, int R, int per){
  uint64_t t = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x;
  uint64_t grp = t / R; int lane = t % R;
  for (int i=0;i<per;++i){
    uint64_t s = (mix(grp*per+i) % (nrec / R)) * R;
    out[s+lane] = make_uint4((uint32_t)t, i, 0, 0);
  }
}
__global__ void k_st4(uint32_t* out, size_t n, int per){
  uint64_t t = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x;
  for (int i=0;i<per;++i){ out[mix(t*per+i) % n] = (uint32_t)t; }
}
__global__ void k_ld4(const uint32_t* in, size_t n, int per, uint32_t* sink){
  uint64_t t = (uint64_t)blockIdx.x*blockDim.x+threadIdx.x; uint32_t acc=0;
  for (int i=0;i<per;++i){ acc += in[mix(t*per+i) % n]; }
  if (acc==0xdeadbeef) *sink=acc;
}
template<typename F> float timeit(F f, int reps=3){
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize();
  hipEventRecord(a); for(int i=0;i<reps;++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b); return ms/reps;
}
int main(){
  size_t bytes = (size_t)4<<30; uint4 *A,*B; CK(hipMalloc(&A,bytes)); CK(hipMalloc(&B,bytes)); CK(hipMemset(A,1,bytes)); CK(hipMemset(B,0,bytes));
  unsigned long long* sink; CK(hipMalloc(&sink,8));
  { float ms = timeit([&]{ hipLaunchKernelGGL(k_copy,dim3(8192),dim3(256),0,0,A,B,bytes/16); });
    printf("copy 4GiB: %.3f ms  -> %.2f TB/s (r+w)\n", ms, 2.0*bytes/ms/1e9); }
  for (uint64_t nc : {256ull, 4096ull, 65536ull, 1048576ull, 16777216ull}){
    unsigned long long* c; CK(hipMalloc(&c, nc*8)); CK(hipMemset(c,0,nc*8));
    int per=16; size_t threads=(size_t)1<<24;
    float ms = timeit([&]{ hipLaunchKernelGGL(k_atomic,dim3(threads/256),dim3(256),0,0,c,nc,per,sink); });
    float ms2 = timeit([&]{ hipLaunchKernelGGL(k_atomic_noret,dim3(threads/256),dim3(256),0,0,c,nc,per); });
    printf("atomicAdd u64 random over %9llu counters: returning %.2f Gop/s, no-return %.2f Gop/s\n", (unsigned long long)nc, threads*per/ms/1e6, threads*per/ms2/1e6);
    hipFree(c);
  }
  for (int R : {1,2,4,8,16,32,64}){
    int per=8; size_t threads=(size_t)1<<25;
    float ms = timeit([&]{ hipLaunchKernelGGL(k_runs,dim3(threads/256),dim3(256),0,0,B,bytes/16,R,per); });
    printf("scatter runs of %2d x16B (%4d B) into 4 GiB: %.2f TB/s payload\n", R, R*16, 16.0*threads*per/ms/1e9);
  }
  for (size_t n : {(size_t)1<<20, (size_t)1<<24, (size_t)1<<28}){
    int per=16; size_t threads=(size_t)1<<24;
    float ms = timeit([&]{ hipLaunchKernelGGL(k_st4,dim3(threads/256),dim3(256),0,0,(uint32_t*)B,n,per); });
    float ms2 = timeit([&]{ hipLaunchKernelGGL(k_ld4,dim3(threads/256),dim3(256),0,0,(const uint32_t*)B,n,per,(uint32_t*)sink); });
    printf("random 4B over %10zu words: store %.2f Gop/s, load %.2f Gop/s\n", n, threads*per/ms/1e6, threads*per/ms2/1e6);
  }
  return 0;
}

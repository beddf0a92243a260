// Where does global_load_lds put a lane's data?  One wave, sizes 4 and 12: prints the LDS stride per lane it observes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
__global__ void k(const unsigned* src, unsigned* out) {
  __shared__ unsigned buf[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned* p = src + 7 * threadIdx.x;  // lane l's record starts at dword 7 l
  if (threadIdx.x != 5)                        // one lane masked off
    __builtin_amdgcn_global_load_lds((gptr)p, (lptr)buf, 12, 0, 0);
  __builtin_amdgcn_global_load_lds((gptr)(p + 3), (lptr)(buf + 512), 4, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = buf[i];
}
int main() {
  std::vector<unsigned> src(7 * 64), out(1024);
  for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned)i;
  unsigned *ds, *dout;
  hipMalloc(&ds, src.size() * 4); hipMalloc(&dout, 4096);
  hipMemcpy(ds, src.data(), src.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, dout);
  hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
  printf("size 12:"); for (int i = 0; i < 24; ++i) printf(" %x", out[i]); printf("\n");
  printf("size 4 :"); for (int i = 512; i < 524; ++i) printf(" %x", out[i]); printf("\n");
  int ok12 = 1, ok4 = 1;
  for (int l = 0; l < 64; ++l) {
    for (int j = 0; j < 3; ++j) { const unsigned want = l == 5 ? 0xdeadbeefu : (unsigned)(7 * l + j); if (out[3 * l + j] != want) ok12 = 0; }
    if (out[512 + l] != (unsigned)(7 * l + 3)) ok4 = 0;
  }
  printf("lane l's 12 bytes at base + 12 l (masked lane untouched): %s; lane l's 4 bytes at base + 4 l: %s\n", ok12 ? "yes" : "NO", ok4 ? "yes" : "NO");
  return ok12 && ok4 ? 0 : 1;
}

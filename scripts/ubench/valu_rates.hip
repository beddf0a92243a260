// Issue rate of a few vector instructions on gfx950, relative to each other: every kernel runs the same loop of
// 16 independent instructions of one kind per trip, 8 waves per SIMD, and the host times it.
// build: hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, DECL, BODY, SINK)                                                  \
  __global__ void __launch_bounds__(256) NAME(double* out, int iters, float seed) {     \
    DECL;                                                                               \
    for (int it = 0; it < iters; ++it) { REP16(BODY) }                                  \
    double acc = 0; SINK;                                                               \
    if (acc == 12345.678) out[threadIdx.x] = acc;                                       \
  }

#define D_DECL double a[16], b = seed, c = seed * 0.5; for (int k = 0; k < 16; ++k) a[k] = seed + k
#define D_SINK for (int k = 0; k < 16; ++k) acc += a[k]
#define F_DECL float a[16], b = seed, c = seed * 0.5f; for (int k = 0; k < 16; ++k) a[k] = seed + k
#define F_SINK for (int k = 0; k < 16; ++k) acc += a[k]

#define B_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define B_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define B_ADD64(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define B_RCP64(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
#define B_RSQ64(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[k]));
#define B_SQRT64(k) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[k]));
#define B_MAX64(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define B_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define B_MIN32(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define B_MAX3_32(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
#define B_RCP32(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define B_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b));

KERNEL(k_fma64, D_DECL, B_FMA64, D_SINK)
KERNEL(k_mul64, D_DECL, B_MUL64, D_SINK)
KERNEL(k_add64, D_DECL, B_ADD64, D_SINK)
KERNEL(k_max64, D_DECL, B_MAX64, D_SINK)
KERNEL(k_rcp64, D_DECL, B_RCP64, D_SINK)
KERNEL(k_rsq64, D_DECL, B_RSQ64, D_SINK)
KERNEL(k_sqrt64, D_DECL, B_SQRT64, D_SINK)
KERNEL(k_fma32, F_DECL, B_FMA32, F_SINK)
KERNEL(k_min32, F_DECL, B_MIN32, F_SINK)
KERNEL(k_max3_32, F_DECL, B_MAX3_32, F_SINK)
KERNEL(k_rcp32, F_DECL, B_RCP32, F_SINK)
KERNEL(k_mov32, F_DECL, B_MOV, F_SINK)

// conversions and compares / selects: mixed operand types
__global__ void __launch_bounds__(256) k_cvt_f64_f32(double* out, int iters, float seed) {
  double a[16]; float f[16];
  for (int k = 0; k < 16; ++k) f[k] = seed + k;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f[k]));
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_cvt_f32_f64(double* out, int iters, float seed) {
  double a[16]; float f[16];
  for (int k = 0; k < 16; ++k) a[k] = seed + k;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(a[k]));
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += f[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_cmp_f64(double* out, int iters, float seed) {
  double a[16]; double b = seed;
  for (int k = 0; k < 16; ++k) a[k] = seed + k;
  unsigned long long m = 0;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc + m;
}
__global__ void __launch_bounds__(256) k_cndmask(double* out, int iters, float seed) {
  float a[16]; float b = seed, c = seed * 2;
  for (int k = 0; k < 16; ++k) a[k] = seed + k;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[k]) : "v"(b), "v"(c) : );
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_pk_fma32(double* out, int iters, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a[16]; f2 b = {seed, seed}, c = {seed * 0.5f, seed};
  for (int k = 0; k < 16; ++k) a[k] = f2{seed + k, seed - k};
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k].x + a[k].y;
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_readlane(double* out, int iters, float seed) {
  int a = (int)seed + threadIdx.x; int s = 0;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a));
    REP16(B)
#undef B
  }
  if (s == 123456) out[threadIdx.x] = s;
}


__global__ void __launch_bounds__(256) k_cndmask_sgpr(double* out, int iters, float seed) {
  float a[16]; float b = seed, c = seed * 2;
  for (int k = 0; k < 16; ++k) a[k] = seed + k;
  const unsigned lo = __builtin_amdgcn_readfirstlane(0x55555555 + iters), hi = __builtin_amdgcn_readfirstlane(0x33333333 + iters);
  const unsigned long long m = ((unsigned long long)hi << 32) | lo;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[k]) : "v"(b), "v"(c), "s"(m));
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_cmp_cndmask(double* out, int iters, float seed) {
  float a[16]; float b = seed, c = seed * 2;
  for (int k = 0; k < 16; ++k) a[k] = seed + k;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[k]) : "v"(b), "v"(c) : "vcc");
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_fma32_sgpr(double* out, int iters, float seed) {
  float a[16]; float c = seed * 0.5f;
  for (int k = 0; k < 16; ++k) a[k] = seed + k;
  const int si = __builtin_amdgcn_readfirstlane(__float_as_int(seed));
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[k]) : "s"(si), "v"(c));
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_min3_32(double* out, int iters, float seed) {
  float a[16]; float b = seed, c = seed * 0.5f;
  for (int k = 0; k < 16; ++k) a[k] = seed + k;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_add32(double* out, int iters, float seed) {
  float a[16]; float b = seed;
  for (int k = 0; k < 16; ++k) a[k] = seed + k;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_and32(double* out, int iters, float seed) {
  int a[16]; int b = (int)seed;
  for (int k = 0; k < 16; ++k) a[k] = (int)seed + k;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
    REP16(B)
#undef B
  }
  double acc = 0; for (int k = 0; k < 16; ++k) acc += a[k];
  if (acc == 12345.678) out[threadIdx.x] = acc;
}
template <class K> void run(const char* name, K kern, double* out) {
  const int iters = 4000, blocks = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10, 1.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  // instructions per SIMD = iters * 16 * 8 waves; cycles per instruction per SIMD at 2.4 GHz
  const double per_simd = (double)iters * 16 * 8;
  printf("%-16s %8.3f ms   %6.2f cycles per wave instruction at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}


// dependent chains: every instruction reads the result of the one before it
__global__ void __launch_bounds__(256) k_chain_fma64(double* out, int iters, float seed) {
  double a = seed, b = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    REP16(B)
#undef B
  }
  if (a == 12345.678) out[threadIdx.x] = a;
}
__global__ void __launch_bounds__(256) k_chain_fma32(double* out, int iters, float seed) {
  float a = seed, b = 1.0000001f, c = 1e-9f;
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    REP16(B)
#undef B
  }
  if (a == 12345.678f) out[threadIdx.x] = a;
}
__global__ void __launch_bounds__(256) k_chain_mix(double* out, int iters, float seed) {  // vector, scalar, vector, scalar ...
  double a = seed, b = 1.0000001, c = 1e-9;
  int s = __builtin_amdgcn_readfirstlane(iters);
  for (int it = 0; it < iters; ++it) {
#define B(k) asm volatile("v_fma_f64 %0, %0, %2, %3\n s_add_i32 %1, %1, 3" : "+v"(a), "+s"(s) : "v"(b), "v"(c) : "scc");
    REP16(B)
#undef B
  }
  if (a == 12345.678) out[threadIdx.x] = a + s;
}
template <class K> void run_waves(const char* name, K kern, double* out, int per_instr) {
  for (int wps : {1, 2, 4, 5, 8}) {
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10, 1.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %d waves per SIMD: %6.2f cycles per instruction of one wave, %6.2f per instruction of the SIMD (2.4 GHz)\n", name, wps,
           ms * 1e-3 * 2.4e9 / ((double)iters * 16 * per_instr), ms * 1e-3 * 2.4e9 / ((double)iters * 16 * per_instr * wps));
  }
}

int main() {
  double* out; hipMalloc(&out, 4096);
  run("v_fma_f64", k_fma64, out); run("v_mul_f64", k_mul64, out); run("v_add_f64", k_add64, out); run("v_max_f64", k_max64, out);
  run("v_rcp_f64", k_rcp64, out); run("v_rsq_f64", k_rsq64, out); run("v_sqrt_f64", k_sqrt64, out);
  run("v_cvt_f64_f32", k_cvt_f64_f32, out); run("v_cvt_f32_f64", k_cvt_f32_f64, out); run("v_cmp_lt_f64", k_cmp_f64, out);
  run("v_cndmask_b32", k_cndmask, out); run("v_fma_f32", k_fma32, out); run("v_pk_fma_f32", k_pk_fma32, out);
  run("v_min_f32", k_min32, out); run("v_max3_f32", k_max3_32, out); run("v_rcp_f32", k_rcp32, out); run("v_mov_b32", k_mov32, out);
  run("v_readlane_b32", k_readlane, out);
  run("v_cndmask e64 sgpr", k_cndmask_sgpr, out); run("v_cmp+v_cndmask", k_cmp_cndmask, out); run("v_fma_f32 sgpr op", k_fma32_sgpr, out);
  run("v_min3_f32", k_min3_32, out); run("v_add_f32", k_add32, out); run("v_and_b32", k_and32, out);
  run_waves("chain v_fma_f64", k_chain_fma64, out, 1); run_waves("chain v_fma_f32", k_chain_fma32, out, 1);
  run_waves("chain fma64+s_add", k_chain_mix, out, 2);
  hipFree(out);
  return 0;
}

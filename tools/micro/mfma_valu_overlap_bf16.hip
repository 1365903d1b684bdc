// Micro-benchmark (GPU box): does the VALU work of one wave run under the MFMAs of the OTHER wave
// on its SIMD?  512 threads, one block per CU (2 waves per SIMD), per step: four
// v_mfma_f32_32x32x16_bf16 (BF16=1) or v_mfma_f32_16x16x4_f32 followed by NV plain VALU operations (4 cycles each),
// as the staging pieces of the Winograd kernels sit behind a stage's MFMAs.
//   lockstep : every wave runs NV VALU ops in every step
//   stagger  : waves 0-3 run 2 NV ops in even steps, waves 4-7 (the SIMD partners) in odd steps
//   spread   : the NV ops are spread between the four MFMAs (NV/4 after each)
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N>
__device__ __forceinline__ void valu(float (&x)[4], float a, float b) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 3]) : "v"(a), "v"(b));
}

template <int NV, int MODE>
__global__ __launch_bounds__(512, 2) void kb(float* out, int iters) {
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a = 1.0f + 0.001f * lane, b = 0.5f;
  bf16x8 av, bv;
#pragma unroll
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b + i); }
  float x[4] = {a, b, a + b, a - b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[(s & 1) * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[(s & 1) * 4 + q], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          valu<NV / 4>(x, a, b);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[(s & 1) * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[(s & 1) * 4 + q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        valu<NV>(x, a, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float sm = x[0] + x[1] + x[2] + x[3];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sm += acc[t][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sm;
}

template <int NV, int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  f32x4 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  float a = 1.0f + 0.001f * lane, b = 0.5f;
  float x[4] = {a, b, a + b, a - b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (MODE == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[s + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[s + h], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu<NV / 4>(x, a, b);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[s + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[s + h], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (MODE == 0) valu<NV>(x, a, b);
          if (MODE == 1) { if (grp == h) valu<2 * NV>(x, a, b); }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float s = x[0] + x[1] + x[2] + x[3];
#pragma unroll
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int nv, int iters) {
  float* out;
  hipMalloc(&out, (size_t)256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 100 << 10, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps = 16.0 * iters;
    if (rep == 2)
      printf("%-28s NV=%3d  %7.3f ms  %6.1f ns/step  (MFMA alone: 2 waves x 4 x 32 cyc = 256 cyc = 106.7 ns at 2.4 GHz)\n",
             name, nv, ms, ms * 1e6 / steps);
  }
  hipFree(out);
}

#define RUN(NV)                                                      \
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<NV, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10); \
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<NV, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10); \
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<NV, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10); \
  run("lockstep", k<NV, 0>, NV, 4000); run("stagger", k<NV, 1>, NV, 4000); run("spread", k<NV, 2>, NV, 4000);

#define RUNB(NV) run("bf16 32x32x16 lockstep", kb<NV, 0>, NV, 4000); run("bf16 32x32x16 spread", kb<NV, 2>, NV, 4000);
int main() {
  printf("bf16: 2 waves x 4 x v_mfma_f32_32x32x16_bf16 per step\n");
  RUNB(0) RUNB(8) RUNB(16) RUNB(32) RUNB(64)
  return 0;
  return 0;
}

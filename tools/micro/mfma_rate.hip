// Micro-benchmark (GPU box): sustained v_mfma_f32_32x32x2_f32 rate per CU under the occupancy
// shapes the convolution kernels use.  Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
//   variant 0: 256 threads, 1 block/CU  (1 wave per SIMD), 9 independent accumulators
//   variant 1: 256 threads, 2 blocks/CU (2 waves per SIMD, independent blocks)
//   variant 2: 512 threads, 1 block/CU  (2 waves per SIMD, one block)
//   variant 3: as 1 + 10 ds_read_b32 per 9 MFMAs (operands re-read from LDS each step)
//   variant 4: as 0 + 10 ds_read_b32 per 9 MFMAs
//   variant 5: as 1, one __syncthreads() per 16 steps (144 MFMAs)
//   variant 6: as 2, one __syncthreads() per 16 steps
//   variant 7: 256 threads 1 block/CU, 18 accumulators (two 9-tap sets), 512-VGPR budget
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS, int SYNC>
__device__ __forceinline__ void body(float* out, int iters, float* smem) {
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a[NACC + 1];
#pragma unroll
  for (int t = 0; t <= NACC; ++t) a[t] = 1.0f + 0.001f * (lane + t);
  if (LDS) {
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) smem[i] = 1.0f + 1e-4f * i;
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
      const float* P = smem + ((it & 15) * 64 + lane);
#pragma unroll
      for (int t = 0; t <= NACC; ++t) a[t] = P[t * 1024 % 7168];
    }
#pragma unroll
    for (int t = 0; t < NACC; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], a[NACC], acc[t], 0, 0, 0);
    if (SYNC && (it % SYNC) == SYNC - 1) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NT, int WPE, int NACC, bool LDS, int SYNC>
__global__ __launch_bounds__(NT, WPE) void k(float* out, int iters) {
  extern __shared__ float smem[];
  body<NACC, LDS, SYNC>(out, iters, smem);
}

template <typename K>
static void run(const char* name, K kern, int nt, int blocks_per_cu, size_t lds, int nacc, int iters) {
  float* out;
  hipMalloc(&out, (size_t)256 * blocks_per_cu * nt * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * blocks_per_cu), dim3(nt), lds, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 2 * nacc * (double)iters * (nt / 64) * 256.0 * blocks_per_cu;
    if (rep == 2) printf("%-60s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop / ms * 1e-9);
  }
  hipFree(out);
}

int main() {
  const int it = 20000;
  // LDS hog of 96 KB forces one block per CU; 64 KB allows two
  run("v0 256thr 1blk/CU (1 wave/SIMD) 9 acc", k<256, 1, 9, false, 0>, 256, 1, 96 << 10, 9, it);
  run("v1 256thr 2blk/CU (2 waves/SIMD) 9 acc", k<256, 2, 9, false, 0>, 256, 2, 64 << 10, 9, it);
  run("v2 512thr 1blk/CU (2 waves/SIMD) 9 acc", k<512, 2, 9, false, 0>, 512, 1, 96 << 10, 9, it);
  run("v3 256thr 2blk/CU + 10 ds_read/9 mfma", k<256, 2, 9, true, 0>, 256, 2, 64 << 10, 9, it);
  run("v4 256thr 1blk/CU + 10 ds_read/9 mfma", k<256, 1, 9, true, 0>, 256, 1, 96 << 10, 9, it);
  run("v5 256thr 2blk/CU, barrier per 16 steps", k<256, 2, 9, false, 16>, 256, 2, 64 << 10, 9, it);
  run("v6 512thr 1blk/CU, barrier per 16 steps", k<512, 2, 9, false, 16>, 512, 1, 96 << 10, 9, it);
  run("v7 256thr 1blk/CU 18 acc (512 VGPR)", k<256, 1, 18, false, 0>, 256, 1, 96 << 10, 18, it / 2);
  run("v8 256thr 1blk/CU 18 acc + ds_read", k<256, 1, 18, true, 0>, 256, 1, 96 << 10, 18, it / 2);
  run("v9 512thr 1blk/CU + ds_read + barrier/16", k<512, 2, 9, true, 16>, 512, 1, 96 << 10, 9, it);
  run("v10 256thr 2blk/CU + ds_read + barrier/16", k<256, 2, 9, true, 16>, 256, 2, 64 << 10, 9, it);
  return 0;
}

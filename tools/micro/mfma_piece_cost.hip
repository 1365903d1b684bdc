// Micro-benchmark (GPU box): what a staging "piece" placed behind a stage's four
// v_mfma_f32_16x16x4_f32 costs in matrix time, 512 threads = 2 waves per SIMD, one block per CU.
//   piece 0: nothing            piece 1: 16 VALU (v_fma_f32)      piece 2: 16 VALU as 8 v_pk_fma_f32
//   piece 3: 4 ds_read_b64 + s_waitcnt lgkmcnt(0)                 piece 4: 4 ds_write_b32
//   piece 5: 2 ds_write2st64_b32   piece 6: 24 SALU   piece 7: 1 buffer_load_dwordx4 (waited a step later)
//   piece 8: 3 ds_read_b64 issued BEFORE the MFMAs, waited after them (the fragment reads)
// lockstep: every wave in every step; stagger: waves 0-3 in even steps, their SIMD partners 4-7 in odd steps
// Build: hipcc -O3 --offload-arch=gfx950 mfma_piece_cost.hip -o mfma_piece_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int P>
__device__ __forceinline__ void piece(float (&x)[8], f32x2 (&r)[4], unsigned lds, const float* g, f32x4& ld) {
  const float a = x[6], b = x[7];
  if (P == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 3]) : "v"(a), "v"(b));
  }
  if (P == 2) {
    f32x2 u = {x[0], x[1]}, v = {x[2], x[3]}, ab = {a, b};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(u) : "v"(ab));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(ab));
    }
    x[0] = u[0]; x[1] = u[1]; x[2] = v[0]; x[3] = v[1];
  }
  if (P == 3) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:272\n\t"
                 "ds_read_b64 %3, %4 offset:280\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "v"(lds) : "memory");
  }
  if (P == 4) {
    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:1440\n\tds_write_b32 %0, %3 offset:2880\n\t"
                 "ds_write_b32 %0, %4 offset:4320" :: "v"(lds), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]) : "memory");
  }
  if (P == 5) {
    asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:8\n\tds_write2st64_b32 %0, %3, %4 offset0:16 offset1:24"
                 :: "v"(lds), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]) : "memory");
  }
  if (P == 6) {
    int s = __builtin_amdgcn_readfirstlane((int)lds);
#pragma unroll
    for (int i = 0; i < 24; ++i) asm volatile("s_add_i32 %0, %0, 3" : "+s"(s));
    asm volatile("" :: "s"(s));
  }
}

template <int P, int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const float* g, int iters) {
  extern __shared__ float smem[];
  f32x4 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  float a = 1.0f + 0.001f * lane, b = 0.5f;
  float x[8] = {a, b, a + b, a - b, 0, 0, a, b};
  f32x2 r[4] = {};
  f32x2 fr[3] = {{a, b}, {a, b}, {a, b}};
  f32x4 ld = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < 16384; i += 512) smem[i] = 1.f;
  __syncthreads();
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem + threadIdx.x * 8;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, 1 << 24, 0x00020000);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (P == 8) {
          asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:2048\n\tds_read_b64 %2, %3 offset:2560"
                       : "=v"(fr[0]), "=v"(fr[1]), "=v"(fr[2]) : "v"(lds) : "memory");
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[s + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[s + h], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (P == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]));
        if (P == 7) {
          if (MODE == 0 || grp == h) {
            x[4] += ld[0];
            ld = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (threadIdx.x + 512 * blockIdx.x) * 16, (it & 63) * 1024, 0));
          }
        } else if (P != 8) {
          if (MODE == 0) piece<P>(x, r, lds, g, ld);
          if (MODE == 1) { if (grp == h) { piece<P>(x, r, lds, g, ld); piece<P>(x, r, lds, g, ld); } }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = x[0] + x[1] + x[2] + x[3] + x[4] + ld[0] + fr[0][0] + fr[1][0] + fr[2][1];
#pragma unroll
  for (int t = 0; t < 4; ++t) s += r[t][0] + r[t][1];
#pragma unroll
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int iters, float* out, float* g) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 100 << 10, 0, out, g, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("%-44s %6.1f ns/step\n", name, ms * 1e6 / (16.0 * iters));
  }
}

int main() {
  float *out, *g;
  hipMalloc(&out, (size_t)256 * 512 * 4);
  hipMalloc(&g, 64 << 20);
  hipMemset(g, 0, 64 << 20);
  const int it = 3000;
  printf("MFMA alone: 2 waves x 4 x 32 cycles = 256 cycles per step = 106.7 ns at 2.4 GHz\n");
#define BOTH(P, txt) run(txt " lockstep", k<P, 0>, it, out, g); run(txt " stagger", k<P, 1>, it, out, g);
  run("0 nothing", k<0, 0>, it, out, g);
  BOTH(1, "1 16 v_fma_f32")
  BOTH(2, "2 8 v_pk_fma_f32")
  BOTH(3, "3 4 ds_read_b64 + wait")
  BOTH(4, "4 4 ds_write_b32")
  BOTH(5, "5 2 ds_write2st64_b32")
  BOTH(6, "6 24 s_add")
  BOTH(7, "7 buffer_load_dwordx4 used a step later")
  run("8 3 ds_read_b64 before, wait after the MFMAs", k<8, 0>, it, out, g);
  return 0;
}

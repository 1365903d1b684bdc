// common.h — shared device/host helpers for the gfx950 kernels of libunet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/unet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- error reporting -------------------------------------------------------
void unet_set_error(const char* fmt, ...);

#define UNET_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      unet_set_error(__VA_ARGS__);     \
      return UNET_E_INVALID;           \
    }                                  \
  } while (0)

#define UNET_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      unet_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return UNET_E_LAUNCH;                                                  \
    }                                                                        \
  } while (0)

#define UNET_HIP_CALL(expr)                                                    \
  do {                                                                         \
    hipError_t e__ = (expr);                                                   \
    if (e__ != hipSuccess) {                                                   \
      unet_set_error("%s failed: %s", #expr, hipGetErrorString(e__));          \
      return UNET_E_LAUNCH;                                                    \
    }                                                                          \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device setting: remember it per (kernel
// instantiation, device).  `done` is a function-local static of the launcher; setting the
// attribute twice from two threads is harmless.
#include <atomic>
static inline hipError_t unet_set_max_dyn_lds(const void* kern, size_t bytes,
                                              std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_relaxed);
  return e;
}
#define UNET_SET_DYN_LDS(kern, bytes)                                                   \
  do {                                                                                  \
    static std::atomic<unsigned long long> done__{0};                                   \
    UNET_HIP_CALL(unet_set_max_dyn_lds(reinterpret_cast<const void*>(kern), bytes, done__)); \
  } while (0)

// ---- library-internal entry points shared between translation units (instnorm.hip) ----------
int unet_in_finalize_tiles(const void* partial, int tiles, int px_per_tile, const float* gamma,
                           const float* beta, float eps, const float* mask, float* mean,
                           float* rstd, float* alpha, float* beta2, int N, int HW, int C,
                           hipStream_t stream, void* grp_scratch = nullptr);
// bytes of `grp_scratch` (group sums of the two-level statistics finalize)
static inline size_t unet_in_finalize_scratch_bytes(int N, int C) {
  return (size_t)N * 16 * C * 2 * sizeof(float);
}
int unet_in_stats_masked(const float* y, const float* gamma, const float* beta, float eps,
                         const float* mask, float* mean, float* rstd, float* alpha, float* beta2,
                         void* workspace, size_t workspace_bytes, int N, int HW, int C,
                         hipStream_t stream, int y_is_bf16 = 0);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div64(long long a, long long b) { return (a + b - 1) / b; }

// ---- device helpers --------------------------------------------------------
#ifdef __HIPCC__

// Blocks b and b+8 share an XCD (round-robin dispatch over the 8 XCDs); remap so
// each XCD walks a contiguous range of tile ids (private L2 reuse of halo rows
// and weight panels).  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = orig & 7, idx = orig >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): the three bf16 terms
// of the split-bf16 ("bf16x3") matrix-core path; |x - h - m - l| <= 2^-26 |x|.
__device__ __forceinline__ void split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
  f32x4 r1, r2;
#pragma unroll
  for (int i = 0; i < 4; ++i) { h[i] = (__bf16)v[i]; r1[i] = v[i] - (float)h[i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) { m[i] = (__bf16)r1[i]; r2[i] = r1[i] - (float)m[i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) l[i] = (__bf16)r2[i];
}

// ---- storage-type helpers -------------------------------------------------------------------
// Activations live in HBM as fp32 or (mixed-precision mode, BASELINE config 4) as bf16; all
// arithmetic on them is fp32.  ld4 / st4 move four consecutive channels, buf_ld4 does the same
// through a buffer descriptor (`off` in ELEMENTS, `oob` = 0 or 0x80000000 to force the
// out-of-range zero).
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) {
  const bf16x4 h = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ void st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, const f32x4 v) {
  bf16x4 h;
  h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
  *reinterpret_cast<bf16x4*>(p) = h;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }
template <typename T>
__device__ __forceinline__ f32x4 buf_ld4(const __amdgpu_buffer_rsrc_t rs, unsigned off,
                                         unsigned oob) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (off * 4u) | oob, 0, 0));
  } else {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (off * 2u) | oob, 0, 0);
    // bf16 -> fp32 is a 16-bit shift
    return f32x4{__builtin_bit_cast(float, v[0] << 16), __builtin_bit_cast(float, v[0] & 0xffff0000),
                 __builtin_bit_cast(float, v[1] << 16), __builtin_bit_cast(float, v[1] & 0xffff0000)};
  }
}

// the two halves of buf_ld4<__bf16>: the raw 8 bytes (two registers - a loader that keeps many
// slots in flight holds half the registers until it converts) and the widening
typedef int i32x2r __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i32x2r buf_ld4_raw16(const __amdgpu_buffer_rsrc_t rs, unsigned off,
                                                unsigned oob) {
  return __builtin_amdgcn_raw_buffer_load_b64(rs, (off * 2u) | oob, 0, 0);
}
__device__ __forceinline__ f32x4 widen16(const i32x2r v) {
  return f32x4{__builtin_bit_cast(float, v[0] << 16), __builtin_bit_cast(float, v[0] & 0xffff0000),
               __builtin_bit_cast(float, v[1] << 16), __builtin_bit_cast(float, v[1] & 0xffff0000)};
}

// Bilinear blend of a 2 x 2 neighbourhood with the fused multiply-adds SPELLED OUT (the compiler's
// own contraction differs from kernel to kernel): the bf16 up-sampling kernel and the up-sampling
// loader of conv_patch_b16_kernel both call this, which is what makes the two bit-identical.
__device__ __forceinline__ f32x4 blend2x2(const f32x4 p00, const f32x4 p01, const f32x4 p10,
                                          const f32x4 p11, float wx0, float wx1, float wy0,
                                          float wy1) {
  f32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t0 = __builtin_fmaf(p01[k], wx1, p00[k] * wx0);
    const float t1 = __builtin_fmaf(p11[k], wx1, p10[k] * wx0);
    v[k] = __builtin_fmaf(t1, wy1, t0 * wy0);
  }
  return v;
}

// Chan/Welford merge of (count, mean, M2) pairs.
__device__ __forceinline__ void wf_merge(float& n, float& mean, float& m2, float nb, float mb,
                                         float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb;
  const float d = mb - mean;
  const float f = nb / nt;
  mean += d * f;
  m2 += m2b + d * d * n * f;
  n = nt;
}

#endif  // __HIPCC__

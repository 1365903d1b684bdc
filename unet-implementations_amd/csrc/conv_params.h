// conv_params.h — the gather-GEMM launch descriptor shared by the convolution kernels
// (conv_igemm.hip: fp32 gather-GEMM, row-fused, stride-2 dgrad, stem; conv_patch.hip: the
// patch-staged kernels; conv_lowp.hip: bf16 and split-bf16 gather-GEMMs) and the entry points
// those translation units offer each other.
#pragma once
#include "common.h"

namespace unet_conv {

struct IgemmParams {
  const float* src0;
  const float* src1;
  int C0, C1;        // channels of the two (virtually concatenated) sources
  const float* w;    // packed weights [tap][n][k]: w[tap*tap_stride + (n_off + n)*Ktot + k]
  int tap_stride;
  int n_off;
  unsigned src0_bytes, src1_bytes, w_bytes;  // buffer-descriptor ranges (each < 2 GiB)
  const float* bias; // [Ncols] or nullptr
  float* out;
  int ldo;           // channel count of the output tensor
  int accumulate;
  int N, Hin, Win;   // source spatial size
  int Hl, Wl;        // logical grid
  int Hout, Wout;    // output tensor spatial size
  int sin;           // input coordinate = a*sin + off
  int sout, py, px;  // output coordinate = a*sout + py
  int ntaps;
  // tap table packed 8 bits per tap (4 taps per word): bits 0-1 = offy+1, 2-3 = offx+1,
  // 4-7 = weight tap index.  Lives in SGPRs: no scalar-memory load per K step.
  unsigned tapw[3];
  int Ncols;
  // split-bf16 path: the weights as three bf16 planes, each laid out like `w`
  const __bf16* w3;
  int w3_plane;       // elements per plane
  unsigned w3_bytes;
  // Fused layer pipeline (kernels instantiated with ACT / STATS; ignored by the others).
  // ACT: source s holds the RAW output of the producing convolution and the loader applies
  // that layer's InstanceNorm + LeakyReLU + channel dropout while staging it:
  //   a = lrelu(v * act_alpha[n][c] + act_beta[n][c])   (coefficients already times the
  //   dropout mask; zero padding stays 0).  A null pointer = the source is used as stored.
  const float* act0_alpha; const float* act0_beta;   // [N][C0]
  const float* act1_alpha; const float* act1_beta;   // [N][C1]
  float slope;
  // STATS: per-tile (mean, M2) of the output (bias included) for the InstanceNorm that
  // follows: stats[(n * stats_tiles + tile_in_image) * Ncols + col]; every tile holds the
  // same number of pixels of ONE image.
  float2* stats;
  int stats_tiles;
  // "channel taps" (gather-GEMM only): tap t reads the source channels [t*tap_cstride + c) of a
  // pixel whose channel pitch is src0_pitch (0 = C0): the nine D_tap tensors of the
  // low-resolution data gradient of conv3x3(upsample2x(.)) stored as 9*Cout channels per pixel.
  int tap_cstride;
  int src0_pitch;
  // BSTATS (data-gradient instantiations): the output g = dL/da_l of this launch is FINAL for
  // layer l, so the epilogue also emits the two reductions of that layer's InstanceNorm +
  // LeakyReLU + dropout backward (what in_bwd_reduce_kernel would read g and y again for):
  //   S1 = sum gz,  S2 = sum gz * xhat,   gz = g * mask * lrelu'(z),  z = y * A + B0,
  // per (image, tile, column) into bs_partial[(n * bs_tiles + tile) * Ncols + col].
  const float* bs_y;       // raw conv output y_l, same shape as `out`
  const float* bs_mean;    // [N][Ncols]
  const float* bs_rstd;    // [N][Ncols]
  const float* bs_gamma;   // [Ncols]
  const float* bs_beta;    // [Ncols]
  const float* bs_mask;    // [N][Ncols] or nullptr
  float2* bs_partial;
  int bs_tiles;            // tiles per image
  int bs_tile0;            // first tile index of this launch (per-class stride-2 launches)
};

// The kernels address their operands through buffer descriptors (free zero padding), whose
// range is 2 GiB.  Images are independent along N, so the host dispatchers split a batch whose
// tensor exceeds the range into chunks of this many images (0: one image is already too big).
long long& chunk_limit_bytes();   // misc.hip; 2^31 - 1 unless a test lowered it
inline int batch_chunk(int N, long long per_image_bytes) {
  const long long lim = chunk_limit_bytes();
  const long long n = lim / (per_image_bytes > 0 ? per_image_bytes : 1);
  return (int)(n < N ? n : N);
}

inline void set_tap(IgemmParams& p, int t, int oy, int ox, int wt) {
  const unsigned e = (unsigned)(oy + 1) | ((unsigned)(ox + 1) << 2) | ((unsigned)wt << 4);
  p.tapw[t >> 2] |= e << ((t & 3) * 8);
}

// Epilogue of one 32x32 accumulator block: o[r] = destination of register r (nullptr = out of
// range).  With `accumulate` all 16 old values are loaded before the first add, so the reads
// overlap (a per-element load/add/store chain costs one HBM round trip per register).
#ifdef __HIPCC__
template <typename TO>
__device__ __forceinline__ void store_block16(TO* const (&o)[16], const f32x16& acc, float bv,
                                              int accumulate) {
  if (accumulate) {
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = o[r] ? ld1(o[r]) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (o[r]) st1(o[r], acc[r] + bv + old[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (o[r]) st1(o[r], acc[r] + bv);
  }
}
// The same with 32-bit element offsets from `base` (kNoOut = out of range): half the address
// registers of the pointer form (a tensor stays below 2^29 elements: batch_chunk above).
constexpr unsigned kNoOut = 0xffffffffu;
template <typename TO>
__device__ __forceinline__ void store_block16_off(TO* base, const unsigned (&o)[16],
                                                  const f32x16& acc, float bv, int accumulate) {
  if (accumulate) {
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = o[r] != kNoOut ? ld1(base + o[r]) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (o[r] != kNoOut) st1(base + o[r], acc[r] + bv + old[r]);
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (o[r] != kNoOut) st1(base + o[r], acc[r] + bv);
  }
}
#endif  // __HIPCC__

#ifdef __HIPCC__
// Consumer-side activation of four channels: InstanceNorm (folded into alpha/beta) + LeakyReLU
// with 0 <= slope <= 1 (so lrelu(z) = max(z, slope*z): no compare / select pair), written as
// whole-vector expressions so the multiplies and the fma become v_pk_*_f32.  okf = 1 inside the
// image, 0 for a zero-padding slot (the padding is applied AFTER the activation).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 act4f(const f32x4 v, const f32x4 al, const f32x4 be,
                                       float slope, float okf) {
  const f32x4 z = (v * al + be) * okf;
  const f32x4 sz = z * slope;
  return __builtin_elementwise_max(z, sz);
}
__device__ __forceinline__ f32x4 act4(const f32x4 v, const f32x4 al, const f32x4 be, float slope,
                                      bool ok) {
  return act4f(v, al, be, slope, ok ? 1.f : 0.f);
}

// merge two (mean, M2) summaries of `cnt` samples each into the first
__device__ __forceinline__ void wf_merge_eq(float& mean, float& m2, float mb, float m2b,
                                            float cnt) {
  const float d = mb - mean;
  mean += 0.5f * d;
  m2 += m2b + d * d * (0.5f * cnt);
}

// (mean, M2) of the TM x 16 values val(m, r) (accumulator + bias) a lane holds for one output
// column, merged with the partner lane (lane ^ 32 holds the other 16 rows of each 32-row
// block): the result summarises the wave's 32*TM pixels of that column.
template <int TM, typename F>
__device__ __forceinline__ float2 wave_col_stats(F&& val) {
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += val(m, r);
  float mean = s * (1.f / (16 * TM));
  float m2 = 0.f;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = val(m, r) - mean;
      m2 = fmaf(d, d, m2);
    }
  const float mb = __shfl_xor(mean, 32, 64), qb = __shfl_xor(m2, 32, 64);
  wf_merge_eq(mean, m2, mb, qb, 16.f * TM);
  return float2{mean, m2};
}

// Per-lane part of the BSTATS epilogue for one output column: `g(m, r)` = final gradient
// values this lane holds, `yv(m, r)` = the raw conv output at the same positions.  Returns the
// wave's (S1, S2) for the column (both 32-lane halves summed).
struct BwdCoef { float A, B0, mu, rs, mk; };
__device__ __forceinline__ BwdCoef bwd_coef(const IgemmParams& p, int n, int col) {
  const size_t i = (size_t)n * p.Ncols + col;
  BwdCoef c;
  c.mu = p.bs_mean[i];
  c.rs = p.bs_rstd[i];
  c.A = p.bs_gamma[col] * c.rs;
  c.B0 = p.bs_beta[col] - c.mu * c.A;      // same expression as the forward / in_bwd kernels
  c.mk = p.bs_mask ? p.bs_mask[i] : 1.f;
  return c;
}
template <int TM, typename G, typename Y>
__device__ __forceinline__ float2 wave_bwd_stats(const BwdCoef c, float slope, G&& g, Y&& yv) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float y = yv(m, r);
      const float z = fmaf(y, c.A, c.B0);
      const float gz = g(m, r) * c.mk * (z > 0.f ? 1.f : slope);
      s1 += gz;
      s2 = fmaf(gz, (y - c.mu) * c.rs, s2);
    }
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  return float2{s1, s2};
}
// block-level sum of the waves' (S1, S2) per column (fixed order); thread c < BN gets the result
template <int BN, int WAVES_M>
__device__ __forceinline__ bool block_col_sums(float2* red, float2& out) {
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= BN) return false;
  float a = red[c].x, b = red[c].y;
#pragma unroll
  for (int k = 1; k < WAVES_M; ++k) { a += red[k * BN + c].x; b += red[k * BN + c].y; }
  out = float2{a, b};
  return true;
}

// Block-level finish of the statistics epilogue: `mine` = this wave's summary (32*TM pixels) of
// column `col_local` (0..BN-1); waves_m waves cover different pixels of the same columns.  red
// = LDS scratch of waves_m * BN float2 that no other wave is still reading.  Thread c < BN
// merges the waves in fixed order and returns true with the tile summary in `out`.
template <int BN, int WAVES_M>
__device__ __forceinline__ bool block_col_stats(float2* red, int wave_m, int col_local, bool writer,
                                                float2 mine, float cnt_wave, float2& out) {
  if (writer) red[wave_m * BN + col_local] = mine;
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= BN) return false;
  float mean = red[c].x, m2 = red[c].y;
  if (WAVES_M == 2) {
    wf_merge_eq(mean, m2, red[BN + c].x, red[BN + c].y, cnt_wave);
  } else if (WAVES_M == 4) {
    float mean1 = red[2 * BN + c].x, m21 = red[2 * BN + c].y;
    wf_merge_eq(mean, m2, red[BN + c].x, red[BN + c].y, cnt_wave);
    wf_merge_eq(mean1, m21, red[3 * BN + c].x, red[3 * BN + c].y, cnt_wave);
    wf_merge_eq(mean, m2, mean1, m21, 2.f * cnt_wave);
  }
  out = float2{mean, m2};
  return true;
}
#endif  // __HIPCC__

// ---- dispatchers implemented in the other translation units -------------------------------
bool patch_f32_applicable(const IgemmParams& p);              // conv_patch.hip
int launch_patch_f32_auto(const IgemmParams& p, hipStream_t stream,    // returns 1 if no tile fits
                          int* stats_px = nullptr, int* bs_px = nullptr);
int launch_patch_up_auto(const IgemmParams& p, hipStream_t stream, int* stats_px);
int& c32_winograd_flag();                                      // unet_set_c32_winograd, conv_c32.hip
bool c32_applicable(const IgemmParams& p);                    // 32 -> 32 channels, conv_c32.hip
int launch_c32(const IgemmParams& p, int fused, hipStream_t stream, int* tile_px);
bool wino_up32_applicable(const IgemmParams& p);              // (64 up + 32) -> 32, conv_c32.hip
int launch_wino_up32(const IgemmParams& p, hipStream_t stream);
bool patch_s2_applicable(const IgemmParams& p);               // stride-2 forward, conv_patch.hip
int launch_patch_s2_auto(const IgemmParams& p, hipStream_t stream, int* stats_px);
int launch_patch_s2_b16_auto(const IgemmParams& p, hipStream_t stream, int* stats_px);   // bf16 tensors
int launch_dgrad_s2_patch_auto(const IgemmParams& p, hipStream_t stream, int* bs_tiles_out);
int launch_patch_b16_up_auto(const IgemmParams& p, hipStream_t stream, int* stats_px);   // conv_patch.hip
int launch_dgrad_s2_patch_b16_auto(const IgemmParams& p, hipStream_t stream, int* bs_tiles_out);   // bf16 tensors
int launch_patch_split_auto(const IgemmParams& p, hipStream_t stream);
int launch_patch_b16_auto(const IgemmParams& p, hipStream_t stream, int* stats_px,   // bf16 tensors
                          int* bs_px = nullptr);
int launch_patch_split_fused_auto(const IgemmParams& p, hipStream_t stream, int* stats_px,
                                  int* bs_px);
bool patch_split_applicable(const IgemmParams& p);
int dispatch_igemm(const IgemmParams& p, hipStream_t stream,           // conv_igemm.hip
                   int* stats_px = nullptr, int* bs_px = nullptr);
int dispatch_igemm_bf16(const IgemmParams& p, hipStream_t stream);     // conv_lowp.hip
int dispatch_igemm_b16(const IgemmParams& p, hipStream_t stream, int* stats_px,   // bf16 storage
                       int* bs_px = nullptr);
int dispatch_igemm_split(const IgemmParams& p, hipStream_t stream);    // conv_lowp.hip

}  // namespace unet_conv

// head_loss.hip — 1x1 segmentation head (NHWC activations -> NCHW logits) and
// SimpleLoss (dynamic inverse-frequency weighted CE + soft Dice, ignore_index)
// forward + closed-form gradient.  All HBM-bound; per-(n,class) reductions go
// through per-block slabs and a single-block finalize in double (fixed order).
//
// Replaces segmentation_output (Our_UNet/models/unet.py:374-381,:430) and
// SimpleLoss.forward + autograd (Our_UNet/models/losses.py:24-121).
#include "conv_params.h"

namespace {
using unet_conv::act4;

constexpr int HT = 256;  // pixels per head tile
// ... of the backward kernel, whose tile lives in registers: 64 pixels = 92 VGPRs, five waves per
// SIMD.  At 256 pixels (234 VGPRs, two waves) it ran at 2.1 TB/s: 192 -> 136 us on fp32 tensors,
// 161 -> 108 us on bf16 (tools/bench_head.py; 128 pixels: 178 VGPRs, no gain; 32: 164 / 119 us).
#ifndef UNET_HEAD_BWD_TILE
#define UNET_HEAD_BWD_TILE 64
#endif
constexpr int HB = UNET_HEAD_BWD_TILE;

// ------------------------------------------------------------------ head forward
// 8 lanes per pixel, 4 channels per lane: a wave's load instruction reads 8 whole pixels
// (1 KB contiguous), the 32-channel dot products are finished with three xor-shuffles inside
// each 8-lane group; no LDS.  Lane 0 of a group writes the pixel's K logits (NCHW planes).
template <typename TS>
__global__ __launch_bounds__(256) void head_fwd_kernel(const TS* __restrict__ a,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ b,
                                                       float* __restrict__ logits, long long M,
                                                       int HW, int K,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ beta,
                                                       float slope) {
  // alpha != nullptr (fused layer pipeline): `a` is the raw output of the last convolution
  // and its InstanceNorm + LeakyReLU + dropout is applied on load
  const int tid = threadIdx.x;
  const int seg = tid & 7;                       // channels 4*seg .. 4*seg+3
  f32x4 wk[4];
  float bk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    wk[k] = k < K ? *reinterpret_cast<const f32x4*>(w + k * 32 + seg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    bk[k] = (k < K && b) ? b[k] : 0.f;
  }
  const long long m0 = (long long)blockIdx.x * HT;
  // all loads of the tile first (rows past M are clamped and masked: no branch, the eight pixel
  // loads are in flight together); one coefficient row and one image index for the whole tile
  // where a tile never straddles images (no 64-bit division per pixel) - as head_bwd_kernel
  const bool one_image = HW % HT == 0;
  const long long n_tile = m0 / HW;
  f32x4 al = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
  if (alpha && one_image) {
    const size_t o = (size_t)n_tile * 32 + seg * 4;
    al = *reinterpret_cast<const f32x4*>(alpha + o);
    be = *reinterpret_cast<const f32x4*>(beta + o);
  }
  f32x4 av[HT / 32];
#pragma unroll
  for (int it = 0; it < HT / 32; ++it) {          // 32 pixels per pass of the 256 threads
    const long long m = m0 + it * 32 + (tid >> 3);
    const long long mc = m < M ? m : M - 1;
    av[it] = ld4(a + (size_t)mc * 32 + seg * 4);
  }
#pragma unroll
  for (int it = 0; it < HT / 32; ++it) {
    const long long m = m0 + it * 32 + (tid >> 3);
    const long long mc = m < M ? m : M - 1;
    const long long n = one_image ? n_tile : mc / HW;
    f32x4 v = av[it];
    if (alpha) {   // uniform
      if (!one_image) {
        al = *reinterpret_cast<const f32x4*>(alpha + (size_t)n * 32 + seg * 4);
        be = *reinterpret_cast<const f32x4*>(beta + (size_t)n * 32 + seg * 4);
      }
      v = act4(v, al, be, slope, true);
    }
    float acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // same accumulation order inside a lane as the scalar loop: c = 4*seg .. 4*seg+3
      float s = v[0] * wk[k][0];
      s = fmaf(v[1], wk[k][1], s);
      s = fmaf(v[2], wk[k][2], s);
      s = fmaf(v[3], wk[k][3], s);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      acc[k] = s + bk[k];
    }
    if (seg == 0 && m < M) {
      const long long pp = m - n * HW;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < K) logits[((size_t)n * K + k) * HW + pp] = acc[k];
    }
  }
}

// ------------------------------------------------------------------ head backward
// InstanceNorm-backward reductions of the layer in front of the head (unet_bwd_stats)
struct HeadBs {
  const float* mean; const float* rstd; const float* gamma; const float* beta; const float* mask;
  float2* partial;          // nullptr: no reductions
  int tiles_per_block;
};
// partial[block][K*32 + K]: dw then db
#ifndef UNET_HEAD_BWD_OCC
#define UNET_HEAD_BWD_OCC 1
#endif
template <typename TS>
__global__ __launch_bounds__(256, UNET_HEAD_BWD_OCC) void head_bwd_kernel(const TS* __restrict__ a,
                                                       const float* __restrict__ dl,
                                                       const float* __restrict__ w,
                                                       TS* __restrict__ da,
                                                       float* __restrict__ partial, long long M,
                                                       int HW, int K, long long tiles,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ beta,
                                                       float slope, const HeadBs bs) {
  // 8 lanes per pixel, 4 channels per lane (as head_fwd_kernel): a and da move as whole pixels
  // (1 KB per wave instruction), the K logit gradients of a pixel are broadcast loads; every
  // lane keeps its own dw[k][4 channels] partial sums over the pixels it sees, merged per block
  // through LDS in fixed order.
  __shared__ float red[32][4 * 32 + 4];
  const int tid = threadIdx.x;
  const int seg = tid & 7, grp = tid >> 3;
  f32x4 wk[4], dwacc[4];
  float dbacc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    wk[k] = k < K ? *reinterpret_cast<const f32x4*>(w + k * 32 + seg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    dwacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    dbacc[k] = 0.f;
  }
  const bool one_image = HW % HB == 0;   // a tile never straddles images
  // bs.partial != nullptr (round 4; the launcher checks that a workgroup's tiles - a contiguous
  // range then - lie in ONE image): da is the final gradient of the layer that produced `a`, so
  // the two reductions of its InstanceNorm + LeakyReLU + dropout backward,
  //   S1 = sum gz,  S2 = sum gz * xhat,   gz = da * mask * (z > 0 ? 1 : slope),
  // are formed here from values the kernel already holds (same expressions as the convolution
  // epilogues, conv_params.h) - no pass over (da, y) by in_bwd_reduce_kernel.
  const bool with_bs = bs.partial != nullptr;
  f32x4 cA = {0.f, 0.f, 0.f, 0.f}, cB = cA, cM = cA, cMu = cA, cRs = cA, s1 = cA, s2 = cA;
  const long long t_first = with_bs ? (long long)blockIdx.x * bs.tiles_per_block : blockIdx.x;
  const long long t_step = with_bs ? 1 : gridDim.x;
  const long long t_last = with_bs ? min(t_first + bs.tiles_per_block, tiles) : tiles;
  if (with_bs) {
    const size_t o = (size_t)(t_first * HB / HW) * 32 + seg * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(bs.gamma + seg * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(bs.beta + seg * 4);
    cMu = *reinterpret_cast<const f32x4*>(bs.mean + o);
    cRs = *reinterpret_cast<const f32x4*>(bs.rstd + o);
    cA = g * cRs;
    cB = b - cMu * cA;
    cM = bs.mask ? *reinterpret_cast<const f32x4*>(bs.mask + o) : f32x4{1.f, 1.f, 1.f, 1.f};
  }
  for (long long t = t_first; t < t_last; t += t_step) {
    const long long m0 = t * HB;
    // all loads of the tile first (rows past M are clamped and masked: no branch, so the eight
    // pixel loads and 8 x K gradient loads are in flight together)
    f32x4 av[HB / 32], yv[HB / 32];
    float dv[HB / 32][4];
    f32x4 al = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (alpha && one_image) {   // uniform: one coefficient row for the whole tile
      const size_t o = (size_t)(m0 / HW) * 32 + seg * 4;
      al = *reinterpret_cast<const f32x4*>(alpha + o);
      be = *reinterpret_cast<const f32x4*>(beta + o);
    }
#pragma unroll
    for (int it = 0; it < HB / 32; ++it) {
      const long long m = m0 + it * 32 + grp;
      const long long mc = m < M ? m : M - 1;
      av[it] = ld4(a + (size_t)mc * 32 + seg * 4);
      yv[it] = av[it];
      const long long n = one_image ? m0 / HW : mc / HW, pp = mc - n * HW;
      if (alpha) {   // uniform: the operand is a raw convolution output, activated on load
        if (!one_image) {
          al = *reinterpret_cast<const f32x4*>(alpha + (size_t)n * 32 + seg * 4);
          be = *reinterpret_cast<const f32x4*>(beta + (size_t)n * 32 + seg * 4);
        }
        av[it] = act4(av[it], al, be, slope, true);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        dv[it][k] = (k < K && m < M) ? dl[((size_t)n * K + k) * HW + pp] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < HB / 32; ++it) {
      const long long m = m0 + it * 32 + grp;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < K) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            v[jj] = fmaf(dv[it][k], wk[k][jj], v[jj]);
            dwacc[k][jj] = fmaf(dv[it][k], av[it][jj], dwacc[k][jj]);
          }
          dbacc[k] += dv[it][k];
        }
      if (m < M) st4(da + (size_t)m * 32 + seg * 4, v);
      if (with_bs) {   // uniform (rows past M carry dv = 0, so v = 0 and gz = 0)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float z = fmaf(yv[it][jj], cA[jj], cB[jj]);
          const float gz = v[jj] * cM[jj] * (z > 0.f ? 1.f : slope);   // (fp32 v, as the conv epilogues)
          s1[jj] += gz;
          s2[jj] = fmaf(gz, (yv[it][jj] - cMu[jj]) * cRs[jj], s2[jj]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) red[grp][k * 32 + seg * 4 + jj] = dwacc[k][jj];
    if (seg == 0) red[grp][128 + k] = dbacc[k];
  }
  __syncthreads();
  if (tid < K * 32 + K) {
    const int col = tid < K * 32 ? tid : 128 + (tid - K * 32);
    float sm = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) sm += red[g][col];
    partial[(size_t)blockIdx.x * (K * 32 + K) + tid] = sm;
  }
  if (with_bs) {   // uniform: the 32 pixel groups of the workgroup merged in fixed order
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      red[grp][seg * 4 + jj] = s1[jj];
      red[grp][32 + seg * 4 + jj] = s2[jj];
    }
    __syncthreads();
    if (tid < 64) {
      float sm = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) sm += red[g][tid];
      // partial[(image * tiles_per_image + tile) * 32 + c] = (S1, S2): workgroup b IS tile b
      reinterpret_cast<float*>(bs.partial)[((size_t)blockIdx.x * 32 + (tid & 31)) * 2 + (tid >> 5)] = sm;
    }
  }
}

// one block per output column (K*32 weights then K biases): 256 threads stride the slabs,
// fixed-order LDS tree
__global__ __launch_bounds__(256) void head_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                float* __restrict__ dw,
                                                                float* __restrict__ db,
                                                                int nblocks, int K) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  const int cols = K * 32 + K;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(size_t)b * cols + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (i < K * 32) {
      if (dw) dw[i] = red[0];
    } else if (db) {
      db[i - K * 32] = red[0];
    }
  }
}

// ------------------------------------------------------------------ loss
constexpr int LQ = 13;  // cnt[3], nll[3], I[3], P[3], valid
constexpr int LOSS_BLOCKS = 128;  // per image

struct LossCoef {  // written by finalize, read by the gradient kernel
  float w[3];
  float inv_wsum;   // w_ce / sum_i w[t_i]
  float wdice;
};

__device__ __forceinline__ void softmax3(float z0, float z1, float z2, float& p0, float& p1,
                                         float& p2, float& lse) {
  const float m = fmaxf(z0, fmaxf(z1, z2));
  const float e0 = expf(z0 - m), e1 = expf(z1 - m), e2 = expf(z2 - m);
  const float s = e0 + e1 + e2;
  const float inv = 1.f / s;
  p0 = e0 * inv; p1 = e1 * inv; p2 = e2 * inv;
  lse = m + logf(s);
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ logits,
                                                          const long long* __restrict__ target,
                                                          float* __restrict__ partial, int HW,
                                                          int ignore_index) {
  __shared__ float red[4][LQ];
  const int n = blockIdx.y;
  const float* z = logits + (size_t)n * 3 * HW;
  const long long* tg = target + (size_t)n * HW;
  float q[LQ];
#pragma unroll
  for (int i = 0; i < LQ; ++i) q[i] = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const long long t = tg[p];
    const bool valid = (t != (long long)ignore_index);
    float p0, p1, p2, lse;
    const float z0 = z[p], z1 = z[HW + p], z2 = z[2 * HW + p];
    softmax3(z0, z1, z2, p0, p1, p2, lse);
    if (valid) {
      q[12] += 1.f;
      q[9] += p0; q[10] += p1; q[11] += p2;
      if (t == 0) { q[0] += 1.f; q[3] += lse - z0; q[6] += p0; }
      else if (t == 1) { q[1] += 1.f; q[4] += lse - z1; q[7] += p1; }
      else if (t == 2) { q[2] += 1.f; q[5] += lse - z2; q[8] += p2; }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < LQ; ++i) {
    const float s = wave_sum(q[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < LQ) {
    const int i = threadIdx.x;
    partial[((size_t)n * gridDim.x + blockIdx.x) * LQ + i] =
        (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

// Sum of one column over the slabs, in slab order, in double: 32 loads in flight, then their adds
// (the order - and so the value - of the plain loop, which waited for one L2 round trip per slab:
// 27 us per step for the 128 slabs of a 512 x 512 image).
__device__ __forceinline__ double loss_column_sum(const float* __restrict__ col, int nblocks) {
  double s = 0.0;
  int b = 0;
  for (; b + 32 <= nblocks; b += 32) {
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = col[(size_t)(b + j) * LQ];
#pragma unroll
    for (int j = 0; j < 32; ++j) s += (double)v[j];
  }
  for (; b < nblocks; ++b) s += (double)col[(size_t)b * LQ];
  return s;
}

// single block: thread i < N*LQ sums its column over the slabs (double), then thread 0
// evaluates the loss and the gradient coefficients.
__global__ __launch_bounds__(256) void loss_finalize_kernel(
    const float* __restrict__ partial, int N, int nblocks, float smooth, float w_dice, float w_ce,
    int dynamic_weights, const float* __restrict__ class_weights, float grad_scale,
    float* __restrict__ loss_out, LossCoef* __restrict__ coef, float* __restrict__ dice_ab) {
  extern __shared__ double sums[];  // [N][LQ]
  for (int i = threadIdx.x; i < N * LQ; i += blockDim.x) {
    const int n = i / LQ, k = i - n * LQ;
    sums[i] = loss_column_sum(partial + (size_t)n * nblocks * LQ + k, nblocks);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double cnt[3] = {0, 0, 0}, nll[3] = {0, 0, 0}, total = 0;
  for (int n = 0; n < N; ++n) {
    for (int c = 0; c < 3; ++c) {
      cnt[c] += sums[n * LQ + c];
      nll[c] += sums[n * LQ + 3 + c];
    }
    total += sums[n * LQ + 12];
  }
  float w[3];
  if (dynamic_weights) {
    // Our_UNet/models/losses.py:44-60: count 0 -> 1, w = total/count, renormalised to sum 3
    float cw[3], ws = 0.f;
    for (int c = 0; c < 3; ++c) {
      const float cp = cnt[c] == 0 ? 1.f : (float)cnt[c];
      cw[c] = (float)total / cp;
      ws += cw[c];
    }
    for (int c = 0; c < 3; ++c) w[c] = cw[c] * (3.f / ws);
  } else {
    for (int c = 0; c < 3; ++c) w[c] = class_weights ? class_weights[c] : 1.f;
  }
  double num = 0, den = 0;
  for (int c = 0; c < 3; ++c) {
    num += (double)w[c] * nll[c];
    den += (double)w[c] * cnt[c];
  }
  const float ce = (float)(num / den);
  double dice_sum = 0;
  for (int c = 0; c < 3; ++c) {
    double dmean = 0;
    for (int n = 0; n < N; ++n) {
      const double I = sums[n * LQ + 6 + c], P = sums[n * LQ + 9 + c], T = sums[n * LQ + c];
      const double U = P + T;
      dmean += (2.0 * I + smooth) / (U + smooth);
      // d(dice loss)/dp_c at pixel i of image n = A*[t_i == c] + B (times valid mask)
      const double k = 1.0 / (3.0 * N);
      dice_ab[(n * 3 + c) * 2 + 0] = (float)(-2.0 * k / (U + smooth));
      dice_ab[(n * 3 + c) * 2 + 1] = (float)((2.0 * I + smooth) * k / ((U + smooth) * (U + smooth)));
    }
    dice_sum += 1.0 - dmean / N;
  }
  const float dice = (float)(dice_sum / 3.0);
  loss_out[0] = w_ce * ce + w_dice * dice;
  loss_out[1] = ce;
  loss_out[2] = dice;
  for (int c = 0; c < 3; ++c) {
    loss_out[3 + c] = w[c];
    coef->w[c] = w[c];
  }
  coef->inv_wsum = (float)((double)w_ce * grad_scale / den);
  coef->wdice = w_dice * grad_scale;
}

// ---- sharded batch (data parallel, "global-exact"): the loss of the concatenated batch ------
// Phase 1 leaves this shard's per-image sums in the workspace and its contribution to the
// batch-wide statistics in stats[10] = {cnt[3], nll[3], valid, sum_n dice_n[3]}; the host sums
// stats over the shards (an all-reduce of 80 bytes); phase 2 turns the global statistics into
// the loss value (identical on every shard) and this shard's gradient coefficients.
constexpr int LSTATS = 10;

__global__ __launch_bounds__(256) void loss_shard_stats_kernel(const float* __restrict__ partial,
                                                               int N, int nblocks, float smooth,
                                                               double* __restrict__ sums,
                                                               double* __restrict__ stats) {
  for (int i = threadIdx.x; i < N * LQ; i += blockDim.x) {
    const int n = i / LQ, k = i - n * LQ;
    sums[i] = loss_column_sum(partial + (size_t)n * nblocks * LQ + k, nblocks);
  }
  __syncthreads();   // global writes of this block are visible to it after the barrier
  if (threadIdx.x >= LSTATS) return;
  const int k = threadIdx.x;
  double s = 0.0;
  if (k < 7) {
    const int col = k < 6 ? k : 12;
    for (int n = 0; n < N; ++n) s += sums[n * LQ + col];
  } else {
    const int c = k - 7;
    for (int n = 0; n < N; ++n) {
      const double I = sums[n * LQ + 6 + c], U = sums[n * LQ + 9 + c] + sums[n * LQ + c];
      s += (2.0 * I + smooth) / (U + smooth);
    }
  }
  stats[k] = s;
}

__global__ void loss_shard_apply_kernel(const double* __restrict__ sums, int N,
                                        const double* __restrict__ g, int n_global, float smooth,
                                        float w_dice, float w_ce, int dynamic_weights,
                                        const float* __restrict__ class_weights, float grad_scale,
                                        float* __restrict__ loss_out, LossCoef* __restrict__ coef,
                                        float* __restrict__ dice_ab) {
  const double k = 1.0 / (3.0 * n_global);
  for (int i = threadIdx.x; i < N * 3; i += blockDim.x) {
    const int n = i / 3, c = i - n * 3;
    const double I = sums[n * LQ + 6 + c], U = sums[n * LQ + 9 + c] + sums[n * LQ + c];
    dice_ab[i * 2 + 0] = (float)(-2.0 * k / (U + smooth));
    dice_ab[i * 2 + 1] = (float)((2.0 * I + smooth) * k / ((U + smooth) * (U + smooth)));
  }
  if (threadIdx.x != 0) return;
  const double total = g[6];
  float w[3];
  if (dynamic_weights) {
    float cw[3], ws = 0.f;
    for (int c = 0; c < 3; ++c) {
      const float cp = g[c] == 0 ? 1.f : (float)g[c];
      cw[c] = (float)total / cp;
      ws += cw[c];
    }
    for (int c = 0; c < 3; ++c) w[c] = cw[c] * (3.f / ws);
  } else {
    for (int c = 0; c < 3; ++c) w[c] = class_weights ? class_weights[c] : 1.f;
  }
  double num = 0, den = 0, dice_sum = 0;
  for (int c = 0; c < 3; ++c) {
    num += (double)w[c] * g[3 + c];
    den += (double)w[c] * g[c];
    dice_sum += 1.0 - g[7 + c] / n_global;
  }
  const float ce = (float)(num / den);
  const float dice = (float)(dice_sum / 3.0);
  loss_out[0] = w_ce * ce + w_dice * dice;
  loss_out[1] = ce;
  loss_out[2] = dice;
  for (int c = 0; c < 3; ++c) {
    loss_out[3 + c] = w[c];
    coef->w[c] = w[c];
  }
  coef->inv_wsum = (float)((double)w_ce * grad_scale / den);
  coef->wdice = w_dice * grad_scale;
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ logits,
                                                        const long long* __restrict__ target,
                                                        const LossCoef* __restrict__ coef,
                                                        const float* __restrict__ dice_ab,
                                                        float* __restrict__ dlogits, int HW,
                                                        int ignore_index,
                                                        const float* __restrict__ upstream) {
  const int n = blockIdx.y;
  const float* z = logits + (size_t)n * 3 * HW;
  float* dz = dlogits + (size_t)n * 3 * HW;
  // dL/d(loss) handed down by autograd (a device scalar; 1 for loss.backward()): applied to the
  // finished gradient, i.e. the same rounding as a separate `dlogits *= g` pass
  const float up = upstream ? upstream[0] : 1.f;
  const long long* tg = target + (size_t)n * HW;
  const float w0 = coef->w[0], w1 = coef->w[1], w2 = coef->w[2];
  const float iw = coef->inv_wsum, wd = coef->wdice;
  float A[3], B[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    A[c] = dice_ab[(n * 3 + c) * 2 + 0];
    B[c] = dice_ab[(n * 3 + c) * 2 + 1];
  }
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const long long t = tg[p];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (t != (long long)ignore_index) {
      float p0, p1, p2, lse;
      softmax3(z[p], z[HW + p], z[2 * HW + p], p0, p1, p2, lse);
      const float wt = (t == 0) ? w0 : (t == 1) ? w1 : (t == 2) ? w2 : 0.f;
      const float k = wt * iw;
      // weighted CE: w_t (p - onehot) / sum w
      g0 = k * (p0 - (t == 0 ? 1.f : 0.f));
      g1 = k * (p1 - (t == 1 ? 1.f : 0.f));
      g2 = k * (p2 - (t == 2 ? 1.f : 0.f));
      // dice through the softmax Jacobian: p_k (a_k - sum_c a_c p_c)
      const float a0 = B[0] + (t == 0 ? A[0] : 0.f);
      const float a1 = B[1] + (t == 1 ? A[1] : 0.f);
      const float a2 = B[2] + (t == 2 ? A[2] : 0.f);
      const float dot = a0 * p0 + a1 * p1 + a2 * p2;
      g0 += wd * p0 * (a0 - dot);
      g1 += wd * p1 * (a1 - dot);
      g2 += wd * p2 * (a2 - dot);
    }
    dz[p] = g0 * up; dz[HW + p] = g1 * up; dz[2 * HW + p] = g2 * up;
  }
}

// ------------------------------------------------------------------ validation metrics
// argmax over the 3 class planes (lowest index wins ties, like torch.argmax) and, per class,
// the exact integer counts the reference's validate() turns into Dice scores
// (Our_UNet/src/train.py:556-577): counts[c] = {intersection, predicted, labelled}, pixels
// labelled ignore_index excluded.
__global__ __launch_bounds__(256) void argmax_counts_kernel(const float* __restrict__ logits,
                                                            const long long* __restrict__ target,
                                                            unsigned char* __restrict__ preds,
                                                            unsigned long long* __restrict__ counts,
                                                            int HW, int ignore_index) {
  __shared__ unsigned int red[4][9];
  const int n = blockIdx.y;
  const float* z = logits + (size_t)n * 3 * HW;
  const long long* tg = target + (size_t)n * HW;
  unsigned int q[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) q[i] = 0;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const float z0 = z[p], z1 = z[HW + p], z2 = z[2 * HW + p];
    int am = 0;
    float best = z0;
    if (z1 > best) { best = z1; am = 1; }
    if (z2 > best) { best = z2; am = 2; }
    if (preds) preds[(size_t)n * HW + p] = (unsigned char)am;
    if (!target) continue;           // prediction only (inference)
    const long long t = tg[p];
    if (t != (long long)ignore_index) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const unsigned int pc = (am == c), mc = (t == c);
        q[c * 3 + 0] += pc & mc;
        q[c * 3 + 1] += pc;
        q[c * 3 + 2] += mc;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    unsigned int v = q[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (target && threadIdx.x < 9) {
    const unsigned int s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                           red[3][threadIdx.x];
    if (s) atomicAdd(&counts[threadIdx.x], (unsigned long long)s);   // integer: order-independent
  }
}

struct LossWs {
  float* partial; LossCoef* coef; float* dice_ab; double* sums;
};
size_t loss_ws_layout(int N, LossWs* out, char* base) {
  size_t off = 0;
  const size_t partial_b = align_up((size_t)N * LOSS_BLOCKS * LQ * sizeof(float), 256);
  const size_t coef_b = align_up(sizeof(LossCoef), 256);
  const size_t ab_b = align_up((size_t)N * 3 * 2 * sizeof(float), 256);
  if (out) {
    out->partial = reinterpret_cast<float*>(base + off);
    out->coef = reinterpret_cast<LossCoef*>(base + off + partial_b);
    out->dice_ab = reinterpret_cast<float*>(base + off + partial_b + coef_b);
    out->sums = reinterpret_cast<double*>(base + off + partial_b + coef_b + ab_b);
  }
  off = partial_b + coef_b + ab_b + align_up((size_t)N * LQ * sizeof(double), 256);
  return off;
}

#ifndef UNET_HEAD_BWD_BLOCKS
#define UNET_HEAD_BWD_BLOCKS 1024
#endif
int head_bwd_blocks(long long tiles) { return (int)(tiles < UNET_HEAD_BWD_BLOCKS ? tiles : UNET_HEAD_BWD_BLOCKS); }

}  // namespace

extern "C" int unet_head1x1_fwd(const float* a, const float* w, const float* b, float* logits,
                                int N, int HW, int C, int K, unet_stream_t stream) {
  UNET_REQUIRE(a && w && logits, "head1x1_fwd: null pointer");
  UNET_REQUIRE(C == 32 && K >= 1 && K <= 4 && N > 0 && HW > 0,
               "head1x1_fwd: needs C == 32, K <= 4 (got C=%d K=%d)", C, K);
  const long long M = (long long)N * HW;
  hipLaunchKernelGGL(head_fwd_kernel<float>, dim3((unsigned)ceil_div64(M, HT)), dim3(256), 0,
                     (hipStream_t)stream, a, w, b, logits, M, HW, K, (const float*)nullptr,
                     (const float*)nullptr, 0.f);
  UNET_CHECK_LAUNCH("head_fwd");
  return UNET_OK;
}

extern "C" int unet_head1x1_in_fwd(const unet_act_src* x, float slope, const float* w,
                                   const float* b, float* logits, int N, int HW, int K,
                                   unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && w && logits, "head1x1_in_fwd: null pointer");
  UNET_REQUIRE(x->C == 32 && K >= 1 && K <= 4 && N > 0 && HW > 0 && (!x->alpha || x->beta),
               "head1x1_in_fwd: needs C == 32, K <= 4 (got C=%d K=%d)", x->C, K);
  const long long M = (long long)N * HW;
  hipLaunchKernelGGL(head_fwd_kernel<float>, dim3((unsigned)ceil_div64(M, HT)), dim3(256), 0,
                     (hipStream_t)stream, x->x, w, b, logits, M, HW, K, x->alpha, x->beta, slope);
  UNET_CHECK_LAUNCH("head_fwd");
  return UNET_OK;
}

extern "C" int unet_head1x1_in_fwd_b16(const unet_act_src* x, float slope, const float* w,
                                       const float* b, float* logits, int N, int HW, int K,
                                       unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && w && logits, "head1x1_in_fwd_b16: null pointer");
  UNET_REQUIRE(x->C == 32 && K >= 1 && K <= 4 && N > 0 && HW > 0 && (!x->alpha || x->beta),
               "head1x1_in_fwd_b16: needs C == 32, K <= 4 (got C=%d K=%d)", x->C, K);
  const long long M = (long long)N * HW;
  hipLaunchKernelGGL(head_fwd_kernel<__bf16>, dim3((unsigned)ceil_div64(M, HT)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const __bf16*>(x->x), w, b, logits, M,
                     HW, K, x->alpha, x->beta, slope);
  UNET_CHECK_LAUNCH("head_fwd(b16)");
  return UNET_OK;
}

extern "C" size_t unet_head1x1_bwd_workspace_bytes(int N, int HW, int C, int K) {
  if (N <= 0 || HW <= 0) return 0;
  const long long tiles = ceil_div64((long long)N * HW, HB);
  return (size_t)head_bwd_blocks(tiles) * (K * 32 + K) * sizeof(float);
}

static int head1x1_bwd_impl(const float* a, const float* dlogits, const float* w, float* da,
                            float* dw, float* db, void* workspace, size_t workspace_bytes, int N,
                            int HW, int C, int K, const float* alpha, const float* beta,
                            float slope, unet_stream_t stream, int b16 = 0,
                            unet_bwd_stats* bs = nullptr);

extern "C" int unet_head1x1_bwd(const float* a, const float* dlogits, const float* w, float* da,
                                float* dw, float* db, void* workspace, size_t workspace_bytes,
                                int N, int HW, int C, int K, unet_stream_t stream) {
  return head1x1_bwd_impl(a, dlogits, w, da, dw, db, workspace, workspace_bytes, N, HW, C, K,
                          nullptr, nullptr, 0.f, stream);
}

extern "C" int unet_head1x1_in_bwd(const unet_act_src* x, float slope, const float* dlogits,
                                   const float* w, float* da, float* dw, float* db,
                                   void* workspace, size_t workspace_bytes, int N, int HW, int K,
                                   unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && (!x->alpha || x->beta), "head1x1_in_bwd: null source");
  return head1x1_bwd_impl(x->x, dlogits, w, da, dw, db, workspace, workspace_bytes, N, HW, x->C, K,
                          x->alpha, x->beta, slope, stream);
}

// ... with the reductions of the InstanceNorm + LeakyReLU + dropout backward of the layer whose
// raw output x->x is (bs->y == x->x): da is that layer's final dL/da, so the kernel that writes
// it also sums gz and gz * xhat per workgroup (bs->tiles_out summaries per image; 0 = the shape
// does not split evenly: run unet_instnorm_lrelu_drop_bwd as usual)
extern "C" int unet_head1x1_in_bwd_bs(const unet_act_src* x, float slope, const float* dlogits,
                                      const float* w, float* da, float* dw, float* db,
                                      void* workspace, size_t workspace_bytes, int N, int HW,
                                      int K, unet_bwd_stats* bs, unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && (!x->alpha || x->beta), "head1x1_in_bwd_bs: null source");
  return head1x1_bwd_impl(x->x, dlogits, w, da, dw, db, workspace, workspace_bytes, N, HW, x->C, K,
                          x->alpha, x->beta, slope, stream, 0, bs);
}

extern "C" int unet_head1x1_in_bwd_bs_b16(const unet_act_src* x, float slope, const float* dlogits,
                                          const float* w, uint16_t* da, float* dw, float* db,
                                          void* workspace, size_t workspace_bytes, int N, int HW,
                                          int K, unet_bwd_stats* bs, unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && (!x->alpha || x->beta), "head1x1_in_bwd_bs_b16: null source");
  return head1x1_bwd_impl(x->x, dlogits, w, reinterpret_cast<float*>(da), dw, db, workspace,
                          workspace_bytes, N, HW, x->C, K, x->alpha, x->beta, slope, stream, 1, bs);
}

// x and da are bf16 tensors (mixed-precision pipeline); logits gradient and dw / db stay fp32
extern "C" int unet_head1x1_in_bwd_b16(const unet_act_src* x, float slope, const float* dlogits,
                                       const float* w, uint16_t* da, float* dw, float* db,
                                       void* workspace, size_t workspace_bytes, int N, int HW,
                                       int K, unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && (!x->alpha || x->beta), "head1x1_in_bwd_b16: null source");
  return head1x1_bwd_impl(x->x, dlogits, w, reinterpret_cast<float*>(da), dw, db, workspace,
                          workspace_bytes, N, HW, x->C, K, x->alpha, x->beta, slope, stream, 1);
}

static int head1x1_bwd_impl(const float* a, const float* dlogits, const float* w, float* da,
                            float* dw, float* db, void* workspace, size_t workspace_bytes, int N,
                            int HW, int C, int K, const float* alpha, const float* beta,
                            float slope, unet_stream_t stream, int b16, unet_bwd_stats* bs) {
  UNET_REQUIRE(a && dlogits && w && da && workspace, "head1x1_bwd: null pointer");
  UNET_REQUIRE(C == 32 && K >= 1 && K <= 4 && N > 0 && HW > 0,
               "head1x1_bwd: needs C == 32, K <= 4 (got C=%d K=%d)", C, K);
  if (workspace_bytes < unet_head1x1_bwd_workspace_bytes(N, HW, C, K)) {
    unet_set_error("head1x1_bwd: workspace too small");
    return UNET_E_WORKSPACE;
  }
  const long long M = (long long)N * HW;
  const long long tiles = ceil_div64(M, HB);
  const int blocks = head_bwd_blocks(tiles);
  float* partial = reinterpret_cast<float*>(workspace);
  // reductions of the InstanceNorm backward of the layer in front (bs): a workgroup then takes a
  // contiguous range of tiles, which must lie in one image and divide it evenly - one summary
  // "tile" of tiles_per_block * HB pixels per workgroup
  HeadBs hb{};
  if (bs) bs->tiles_out = 0;
  if (bs && alpha && bs->y == a && bs->mean && bs->rstd && bs->gamma && bs->beta && bs->partial &&
      M % HB == 0) {
    const long long tpb = ceil_div64(tiles, blocks);
    const long long px = tpb * HB;
    if (tiles % tpb == 0 && tiles / tpb == blocks && HW % px == 0 &&
        bs->partial_bytes >= (size_t)blocks * 32 * sizeof(float2)) {
      hb.mean = bs->mean; hb.rstd = bs->rstd; hb.gamma = bs->gamma; hb.beta = bs->beta;
      hb.mask = bs->mask; hb.partial = reinterpret_cast<float2*>(bs->partial);
      hb.tiles_per_block = (int)tpb;
      bs->tiles_out = (int)(HW / px);
      UNET_REQUIRE(bs->slope == slope, "head1x1_in_bwd_bs: bs->slope differs from slope");
    }
  }
  if (b16)
    hipLaunchKernelGGL(head_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(a), dlogits, w,
                       reinterpret_cast<__bf16*>(da), partial, M, HW, K, tiles, alpha, beta, slope,
                       hb);
  else
    hipLaunchKernelGGL(head_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a,
                       dlogits, w, da, partial, M, HW, K, tiles, alpha, beta, slope, hb);
  UNET_CHECK_LAUNCH("head_bwd");
  hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(K * 32 + K), dim3(256), 0, (hipStream_t)stream,
                     partial, dw, db, blocks, K);
  UNET_CHECK_LAUNCH("head_bwd_finalize");
  return UNET_OK;
}

extern "C" size_t unet_dice_wce_loss_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return loss_ws_layout(N, nullptr, nullptr);
}

extern "C" int unet_dice_wce_loss_fwd_bwd(const float* logits, const int64_t* target,
                                          float* loss_out, float* dlogits, void* workspace,
                                          size_t workspace_bytes, int N, int H, int W,
                                          float smooth, float w_dice, float w_ce, int ignore_index,
                                          int dynamic_weights, const float* class_weights,
                                          float grad_scale, unet_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  UNET_REQUIRE(logits && target && loss_out && workspace, "dice_wce_loss: null pointer");
  UNET_REQUIRE(N > 0 && N <= 1024 && H > 0 && W > 0, "dice_wce_loss: bad shape");
  if (workspace_bytes < loss_ws_layout(N, nullptr, nullptr)) {
    unet_set_error("dice_wce_loss: workspace too small");
    return UNET_E_WORKSPACE;
  }
  LossWs ws;
  loss_ws_layout(N, &ws, reinterpret_cast<char*>(workspace));
  const int HW = H * W;
  int blocks = ceil_div(HW, 256 * 8);
  if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(blocks, N), dim3(256), 0, stream, logits,
                     reinterpret_cast<const long long*>(target), ws.partial, HW, ignore_index);
  UNET_CHECK_LAUNCH("loss_reduce");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), (size_t)N * LQ * sizeof(double),
                     stream, ws.partial, N, blocks, smooth, w_dice, w_ce, dynamic_weights,
                     class_weights, grad_scale, loss_out, ws.coef, ws.dice_ab);
  UNET_CHECK_LAUNCH("loss_finalize");
  if (dlogits) {
    int gblocks = ceil_div(HW, 256 * 4);
    if (gblocks < 1) gblocks = 1;
    hipLaunchKernelGGL(loss_grad_kernel, dim3(gblocks, N), dim3(256), 0, stream, logits,
                       reinterpret_cast<const long long*>(target), ws.coef, ws.dice_ab, dlogits, HW,
                       ignore_index, nullptr);
    UNET_CHECK_LAUNCH("loss_grad");
  }
  return UNET_OK;
}

extern "C" int unet_dice_wce_loss_grad(const float* logits, const int64_t* target,
                                       const void* workspace, size_t workspace_bytes,
                                       const float* upstream, float* dlogits, int N, int H, int W,
                                       int ignore_index, unet_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  UNET_REQUIRE(logits && target && workspace && dlogits, "dice_wce_loss_grad: null pointer");
  UNET_REQUIRE(N > 0 && N <= 1024 && H > 0 && W > 0, "dice_wce_loss_grad: bad shape");
  if (workspace_bytes < loss_ws_layout(N, nullptr, nullptr)) {
    unet_set_error("dice_wce_loss_grad: workspace too small");
    return UNET_E_WORKSPACE;
  }
  LossWs ws;
  loss_ws_layout(N, &ws, reinterpret_cast<char*>(const_cast<void*>(workspace)));
  const int HW = H * W;
  int gblocks = ceil_div(HW, 256 * 4);
  if (gblocks < 1) gblocks = 1;
  hipLaunchKernelGGL(loss_grad_kernel, dim3(gblocks, N), dim3(256), 0, stream, logits,
                     reinterpret_cast<const long long*>(target), ws.coef, ws.dice_ab, dlogits, HW,
                     ignore_index, upstream);
  UNET_CHECK_LAUNCH("loss_grad");
  return UNET_OK;
}

extern "C" int unet_argmax_dice_counts(const float* logits_nchw, const int64_t* target,
                                       uint8_t* preds, uint64_t* counts, int N, int H, int W,
                                       int ignore_index, unet_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  UNET_REQUIRE(logits_nchw && ((target && counts) || (!target && preds)),
               "argmax_dice_counts: needs target + counts, or preds alone");
  UNET_REQUIRE(N > 0 && H > 0 && W > 0, "argmax_dice_counts: bad shape");
  const int HW = H * W;
  if (counts) UNET_HIP_CALL(hipMemsetAsync(counts, 0, 9 * sizeof(uint64_t), stream));
  int blocks = ceil_div(HW, 256 * 4);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(argmax_counts_kernel, dim3(blocks, N), dim3(256), 0, stream, logits_nchw,
                     reinterpret_cast<const long long*>(target), preds,
                     reinterpret_cast<unsigned long long*>(counts), HW, ignore_index);
  UNET_CHECK_LAUNCH("argmax_counts");
  return UNET_OK;
}

namespace {
int loss_blocks_for(int HW) {
  int blocks = ceil_div(HW, 256 * 8);
  if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
  return blocks < 1 ? 1 : blocks;
}
}  // namespace

extern "C" int unet_dice_wce_loss_shard_stats(const float* logits, const int64_t* target,
                                              double* stats, void* workspace,
                                              size_t workspace_bytes, int N, int H, int W,
                                              float smooth, int ignore_index,
                                              unet_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  UNET_REQUIRE(logits && target && stats && workspace, "loss_shard_stats: null pointer");
  UNET_REQUIRE(N > 0 && N <= 1024 && H > 0 && W > 0, "loss_shard_stats: bad shape");
  if (workspace_bytes < loss_ws_layout(N, nullptr, nullptr)) {
    unet_set_error("loss_shard_stats: workspace too small");
    return UNET_E_WORKSPACE;
  }
  LossWs ws;
  loss_ws_layout(N, &ws, reinterpret_cast<char*>(workspace));
  const int HW = H * W, blocks = loss_blocks_for(HW);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(blocks, N), dim3(256), 0, stream, logits,
                     reinterpret_cast<const long long*>(target), ws.partial, HW, ignore_index);
  UNET_CHECK_LAUNCH("loss_reduce");
  hipLaunchKernelGGL(loss_shard_stats_kernel, dim3(1), dim3(256), 0, stream, ws.partial, N, blocks,
                     smooth, ws.sums, stats);
  UNET_CHECK_LAUNCH("loss_shard_stats");
  return UNET_OK;
}

extern "C" int unet_dice_wce_loss_shard_apply(const float* logits, const int64_t* target,
                                              const double* global_stats, int N_global,
                                              float* loss_out, float* dlogits, void* workspace,
                                              size_t workspace_bytes, int N, int H, int W,
                                              float smooth, float w_dice, float w_ce,
                                              int ignore_index, int dynamic_weights,
                                              const float* class_weights, float grad_scale,
                                              unet_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  UNET_REQUIRE(logits && target && global_stats && loss_out && workspace,
               "loss_shard_apply: null pointer");
  UNET_REQUIRE(N > 0 && N <= 1024 && N_global >= N && H > 0 && W > 0,
               "loss_shard_apply: bad shape");
  if (workspace_bytes < loss_ws_layout(N, nullptr, nullptr)) {
    unet_set_error("loss_shard_apply: workspace too small");
    return UNET_E_WORKSPACE;
  }
  LossWs ws;
  loss_ws_layout(N, &ws, reinterpret_cast<char*>(workspace));
  const int HW = H * W;
  hipLaunchKernelGGL(loss_shard_apply_kernel, dim3(1), dim3(256), 0, stream, ws.sums, N,
                     global_stats, N_global, smooth, w_dice, w_ce, dynamic_weights, class_weights,
                     grad_scale, loss_out, ws.coef, ws.dice_ab);
  UNET_CHECK_LAUNCH("loss_shard_apply");
  if (dlogits) {
    int gblocks = ceil_div(HW, 256 * 4);
    if (gblocks < 1) gblocks = 1;
    hipLaunchKernelGGL(loss_grad_kernel, dim3(gblocks, N), dim3(256), 0, stream, logits,
                       reinterpret_cast<const long long*>(target), ws.coef, ws.dice_ab, dlogits, HW,
                       ignore_index, nullptr);
    UNET_CHECK_LAUNCH("loss_grad");
  }
  return UNET_OK;
}

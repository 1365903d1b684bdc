// conv_c32.hip — the 32 -> 32 channel 3x3 stride-1 convolutions of the full-resolution stages
// (forward of enc0.3 / dec4.3, and the data gradients that produce 32 channels:
// Our_UNet/models/unet.py:106-115 at 512x512) on the fp32 matrix cores.
//
// K = 9 * 32 and 32 output columns leave one 32x32 accumulator block per wave and only 144
// MFMAs per 32 output pixels, so the per-tile overheads (barriers, load latency, epilogue)
// decide the rate.  This kernel removes them structurally:
//   * persistent workgroup of EIGHT waves (512 threads, one per CU, two waves per SIMD) walking
//     8 x 32-pixel tiles; wave w owns tile row w;
//   * the whole weight tensor lives in REGISTERS: lane (li, lh) keeps its B fragments of all
//     nine taps (9 x 4 f32x4 = 144 VGPRs), so the K loop issues no weight traffic at all;
//   * the (8+2) x 34 x 32-channel input patch is staged once per tile and serves all nine taps;
//     two LDS patch buffers: the next tile's patch is loaded at the start of a tile, rides in
//     registers through the tile's 144 MFMAs per wave (>= 3.8 us: HBM latency is covered) and
//     is written to the other buffer after them - ONE barrier per tile;
//   * statistics (forward) and the next layer's InstanceNorm-backward sums (data gradient):
//     every wave leaves its row's summary in LDS and 32 lanes merge the eight rows (fixed
//     order) AFTER the tile's barrier, so the 256-pixel summaries cost no barrier of their own.
// FUSED = the fused-layer forward (activation on load, bias, statistics); otherwise the data
// gradient form (optional accumulate, optional BSTATS epilogue).
#include "conv_params.h"
#include "lds_asm.h"
#include <stdlib.h>

#include <utility>

namespace unet_conv {
namespace {

template <int B, int... I, typename F>
__device__ __forceinline__ void for_range_c_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, typename F>
__device__ __forceinline__ void for_range_c(F&& f) {
  for_range_c_impl<B>(std::make_integer_sequence<int, (E > B ? E - B : 0)>{}, f);
}

constexpr int C32_TH = 8, C32_TW = 32, C32_PW = C32_TW + 2, C32_PH = C32_TH + 2;
constexpr int C32_LDA = 36;
constexpr int C32_PPIX = C32_PH * C32_PW;          // 340 patch pixels
constexpr int C32_SLOTS = C32_PPIX * 8;            // f32x4 slots
constexpr int C32_PASSES = (C32_SLOTS + 511) / 512;
constexpr size_t C32_LDS = 2 * (size_t)C32_PPIX * C32_LDA * sizeof(float) +
                           2 * 8 * 32 * sizeof(float2) +   // per-wave summaries of two tiles
                           8 * 1024 * sizeof(float);       // BSTATS: the raw outputs under a tile

template <bool FUSED>
__global__ __launch_bounds__(512, 1) void conv_c32_kernel(const IgemmParams p, int ntiles) {
  constexpr int LDA = C32_LDA, PW = C32_PW, P_PASSES = C32_PASSES;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // tile row of this wave
  const int li = lane & 31, lh = lane >> 5;
  const int H = p.Hin, W = p.Win;
  const int tiles_x = W / C32_TW, tiles_y = H / C32_TH;

  // persistent walk: the workgroups of one XCD (blockIdx % 8) cover a contiguous eighth of the
  // tile sequence (halo rows stay in that XCD's L2)
  const int G = gridDim.x;
  int t_first, t_stride, t_end;
  if ((ntiles & 7) == 0 && (G & 7) == 0) {
    const int per = ntiles >> 3, xcd = blockIdx.x & 7;
    t_first = xcd * per + (blockIdx.x >> 3);
    t_stride = G >> 3;
    t_end = (xcd + 1) * per;
  } else {
    t_first = blockIdx.x; t_stride = G; t_end = ntiles;
  }
  if (t_first >= t_end) return;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);

  // ---- weights: this lane's B fragments of every tap, for the whole kernel ----
  f32x4 wb[9][4];
  int tap_off[9];   // patch offset of tap t (floats), uniform
  for_range_c<0, 9>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int oy = (int)(e & 3u) - 1, ox = (int)((e >> 2) & 3u) - 1;
    const int wt = (int)(e >> 4);
    tap_off[t] = (oy * PW + ox) * LDA;
    const float* wp = p.w + (size_t)wt * p.tap_stride + (size_t)(p.n_off + li) * 32 + 4 * lh;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) wb[t][kg] = *reinterpret_cast<const f32x4*>(wp + kg * 8);
  });

  // ---- patch slots of this thread (constant across tiles) ----
  int pp_rel[P_PASSES], pp_lds[P_PASSES], pp_rc[P_PASSES];
#pragma unroll
  for (int i = 0; i < P_PASSES; ++i) {
    const int slot = tid + 512 * i;
    const bool valid = slot < C32_SLOTS;
    const int pix = valid ? slot >> 3 : 0, seg = slot & 7;
    const int prow = pix / PW, pcol = pix - prow * PW;
    pp_rel[i] = ((prow * W + pcol) * 32 + seg * 4) * 4;   // bytes from the patch origin
    pp_lds[i] = pix * LDA + seg * 4;
    pp_rc[i] = valid ? (prow | (pcol << 8)) : (1 << 20);  // invalid: fails both range checks
  }

  f32x4 pr[P_PASSES];
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  float cs = 1.f;
  unsigned okm = 0;
  auto tile_pos = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % tiles_x;
    const int r = tile / tiles_x;
    const int ty = r % tiles_y;
    n = r / tiles_y; y0 = ty * C32_TH; x0 = tx * C32_TW;
  };
  auto load_patch = [&](int tile) {
    int n, y0, x0;
    tile_pos(tile, n, y0, x0);
    const int base = ((n * H + y0 - 1) * W + x0 - 1) * 128;   // bytes; may be negative at borders
    okm = 0;
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      const int prow = pp_rc[i] & 0xff, pcol = pp_rc[i] >> 8;
      const bool ok = (unsigned)(y0 - 1 + prow) < (unsigned)H && (unsigned)(x0 - 1 + pcol) < (unsigned)W;
      okm |= (ok ? 1u : 0u) << i;
      const unsigned off = ok ? (unsigned)(base + pp_rel[i]) : 0x80000000u;
      pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
    }
    if (FUSED) {
      if (p.act0_alpha) {   // uniform
        const size_t o = (size_t)n * 32 + (tid & 7) * 4;
        ca = *reinterpret_cast<const f32x4*>(p.act0_alpha + o);
        cb = *reinterpret_cast<const f32x4*>(p.act0_beta + o);
        cs = p.slope;
      }
    }
  };
  auto store_patch = [&](int buf) {
    float* Pb = smem + buf * (C32_PPIX * LDA);
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      if (FUSED) pr[i] = act4(pr[i], ca, cb, cs, (okm >> i) & 1u);
      if (512 * (i + 1) <= C32_SLOTS || tid + 512 * i < C32_SLOTS)
        *reinterpret_cast<f32x4*>(Pb + pp_lds[i]) = pr[i];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // row summaries of the tile in patch buffer b: red[b][wave][column]
  float2* red = reinterpret_cast<float2*>(smem + 2 * C32_PPIX * LDA);
  const bool summaries = FUSED ? p.stats != nullptr : p.bs_partial != nullptr;   // uniform
  // BSTATS: the 32 x 32 raw outputs y under this wave's row (4 KB, contiguous in HBM) go
  // global -> LDS by DMA at the start of the tile (no registers: none are free), land under the
  // tile's MFMAs and are read back in the epilogue
  float* ylds = smem + 2 * C32_PPIX * LDA + 2 * 8 * 32 * 2 + wave * 1024;
  const bool y_dma = !FUSED && p.bs_partial != nullptr && p.ldo == 32;   // uniform

  load_patch(t_first);
  store_patch(0);
  __syncthreads();

  const int a_lane = ((wave + 1) * PW + li + 1) * LDA + 4 * lh;
  int buf = 0;
  for (int tile = t_first; tile < t_end; tile += t_stride, buf ^= 1) {
    const int nxt = tile + t_stride;
    const bool more = nxt < t_end;
    if (more) load_patch(nxt);   // uniform
    if (y_dma) {
      int n_, y0_, x0_;
      tile_pos(tile, n_, y0_, x0_);
      const float* ysrc = p.bs_y + (((size_t)n_ * H + y0_ + wave) * W + x0_) * 32 + lane * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)   // LDS destination = uniform base + lane * 16 B
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(ysrc + i * 256),
            (__attribute__((address_space(3))) void*)(ylds + i * 256), 16, 0, 0);
    }
    int n, y0, x0;
    tile_pos(tile, n, y0, x0);
    const int y = y0 + wave;
    const size_t pix0o = ((size_t)n * H + y) * W + x0 + 4 * lh;   // first output pixel of this lane

    // ---- 9 taps x 4 k-groups x 4 MFMAs from the staged patch; A fragments one group ahead ----
    const float* P = smem + buf * (C32_PPIX * LDA) + a_lane;
    f32x4 a[2];
    a[0] = *reinterpret_cast<const f32x4*>(P + tap_off[0]);
    for_range_c<0, 36>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int t = g / 4, kg = g % 4;
      constexpr int cur = g & 1, nx = cur ^ 1;
      if constexpr (g + 1 < 36) {
        constexpr int t1 = (g + 1) / 4, kg1 = (g + 1) % 4;
        a[nx] = *reinterpret_cast<const f32x4*>(P + tap_off[t1] + kg1 * 8);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][r], wb[t][kg][r], acc, 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue of this wave's row: D row (pixel) (reg&3) + 8*(reg>>2) + 4*lh, column li ----
    float* o = p.out + pix0o * p.ldo + li;
    if (FUSED) {
      const float bv = p.bias ? p.bias[li] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += bv;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[r];
      if (p.stats) {   // uniform
        const float2 mine = wave_col_stats<1>([&](int, int r) { return acc[r]; });
        if (lh == 0) red[(buf * 8 + wave) * 32 + li] = mine;
      }
    } else {
      if (p.accumulate) {   // uniform: all 16 reads in flight before the first add
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += old[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[r];
      if (p.bs_partial) {   // uniform: reductions of the next backward stage (IgemmParams)
        const BwdCoef cf = bwd_coef(p, n, li);
        float2 mine;
        if (y_dma) {   // uniform
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA issued at the tile start
          const float* yl = ylds + 4 * lh * 32 + li;
          mine = wave_bwd_stats<1>(
              cf, p.slope, [&](int, int r) { return acc[r]; },
              [&](int, int r) { return yl[((r & 3) + 8 * (r >> 2)) * 32]; });
        } else {
          const float* yb = p.bs_y + pix0o * p.ldo + li;
          mine = wave_bwd_stats<1>(
              cf, p.slope, [&](int, int r) { return acc[r]; },
              [&](int, int r) { return yb[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo]; });
        }
        if (lh == 0) red[(buf * 8 + wave) * 32 + li] = mine;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);

    if (more) store_patch(buf ^ 1);   // every wave left that buffer before the previous barrier
    __syncthreads();
    if (summaries && tid < 32) {   // red[buf] is rewritten two tiles (two barriers) from now
      const float2* rr = red + buf * 8 * 32 + tid;
      const size_t dst = ((size_t)n * (tiles_x * tiles_y) + (y0 / C32_TH) * tiles_x + (x0 >> 5)) * 32 + tid;
      if (FUSED) {
        float mean[4], m2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          mean[k] = rr[(2 * k) * 32].x; m2[k] = rr[(2 * k) * 32].y;
          wf_merge_eq(mean[k], m2[k], rr[(2 * k + 1) * 32].x, rr[(2 * k + 1) * 32].y, 32.f);
        }
        wf_merge_eq(mean[0], m2[0], mean[1], m2[1], 64.f);
        wf_merge_eq(mean[2], m2[2], mean[3], m2[3], 64.f);
        wf_merge_eq(mean[0], m2[0], mean[2], m2[2], 128.f);
        p.stats[dst] = float2{mean[0], m2[0]};
      } else {
        float a = rr[0].x, b = rr[0].y;
#pragma unroll
        for (int k = 1; k < 8; ++k) { a += rr[k * 32].x; b += rr[k * 32].y; }
        p.bs_partial[dst] = float2{a, b};
      }
    }
  }
}


// ---------------------------------------------------------------------------
// The same layers in the Winograd F(2x2, 3x3) form: 16 element-wise 32 x 32 products per
// 2 x 2 output tile instead of 36 multiply-adds per output - 4/9 of the matrix-core work.
//
// K = 32 is ONE Winograd "chunk", so there is no K loop to pipeline: the kernel is a sequence
// of phases per unit of work, separated by barriers.
//   * unit = 4 x 16 pixels = 2 x 8 Winograd tiles, its activated (4+2) x 18 x 32 patch in LDS
//     (the next unit's patch rides in registers through the unit); a workgroup of FOUR waves
//     walks whole 8 x 32-pixel tiles, four units each, so the statistics / BSTATS tiles stay
//     256 pixels; two workgroups per CU (70 KB of LDS each);
//   * the transformed weights U = G g G^T live in REGISTERS and are built in the prologue from
//     the packed 3x3 weights the direct kernel reads (no extra weight form, no packing launch):
//     wave w keeps row w of the 4 x 4 xi grid for all 32 output columns: 64 VGPRs;
//   * per unit: every thread transforms one (tile, channel pair) V = B^T d B from the patch into
//     LDS; 64 MFMAs (16x16x4) per wave: M[xi] = V[xi] U[xi] for the wave's four xi, every
//     element of V read once; the column pass of A^T M A in registers (the wave holds a whole
//     row of xi); the row pass across the four waves through a 16 KB LDS exchange, after which
//     lane (column, half) owns 8 output pixels of one column - the epilogue layout of the direct
//     kernel: bias, statistics / BSTATS (accumulated per unit), 128-byte store segments.
// Measured (bs 8, 512 x 512; tools/bench_c32.py): 340 -> 240-270 us per launch.  The phases do
// NOT overlap, and cannot be made to: an eight-wave version (one workgroup per CU, half-tile
// units) took 128 us of products + 111 us of everything else, strictly in sequence; this one,
// whose two workgroups per CU are free to drift apart, takes the same time with or without a
// forced phase offset between them, and one workgroup per CU alone reaches 89 % of the rate of
// two - the same behaviour as in DESIGN.md section 3g (an instruction of either wave of a SIMD
// takes its issue time away from the fp32 matrix pipe), so what counts is the instruction and
// LDS-byte count per unit, not the schedule.
// ---------------------------------------------------------------------------
constexpr int W32_VP = 36;                         // V row pitch in floats (conflict-free b64 fragment reads)

typedef float f32x2w __attribute__((ext_vector_type(2)));

// G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
__device__ __forceinline__ float wino_g(int i, int u) {
  if (i == 0) return u == 0 ? 1.f : 0.f;
  if (i == 3) return u == 2 ? 1.f : 0.f;
  if (i == 1) return 0.5f;
  return u == 1 ? -0.5f : 0.5f;
}

constexpr int Q_PW = 18, Q_PH = 6, Q_PPIX = Q_PW * Q_PH;   // 108 patch pixels
constexpr int Q_SLOTS = Q_PPIX * 8;                        // f32x4 slots
constexpr int Q_PASSES = (Q_SLOTS + 255) / 256;            // 4
constexpr int Q_V = 16 * 16 * W32_VP;                      // [xi][tile slot 16][channel 32 (+4)]
constexpr int Q_Z = 4 * 2 * 4 * 128;                       // [xi row][b][r][column half][lane]
constexpr size_t Q_LDS = ((size_t)Q_PPIX * C32_LDA + Q_V + Q_Z) * sizeof(float) + 4 * 32 * sizeof(float2);

template <bool FUSED>
__global__ __launch_bounds__(256, 2) void conv_wino32q_kernel(const IgemmParams p, int ntiles) {
  constexpr int LDA = C32_LDA, PW = Q_PW, P_PASSES = Q_PASSES;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Pb = smem;
  float* const Vs = smem + Q_PPIX * LDA;
  float* const Zs = Vs + Q_V;
  float2* const red = reinterpret_cast<float2*>(Zs + Q_Z);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row of the xi grid
  const int li = lane & 31, lh = lane >> 5;
  const int H = p.Hin, W = p.Win;
  const int tiles_x = W / C32_TW, tiles_y = H / C32_TH;

  const int G = gridDim.x;
  int t_first, t_stride, t_end;
  if ((ntiles & 7) == 0 && (G & 7) == 0) {
    const int per = ntiles >> 3, xcd = blockIdx.x & 7;
    t_first = xcd * per + (blockIdx.x >> 3);
    t_stride = G >> 3;
    t_end = (xcd + 1) * per;
  } else {
    t_first = blockIdx.x; t_stride = G; t_end = ntiles;
  }
  if (t_first >= t_end) return;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);

  // ---- U = G g G^T, row `wave` of the xi grid, both column halves ----
  const int fn = lane & 15, fk = lane >> 4;
  float ub[4][2][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int k = 0; k < 8; ++k) ub[j][nb][k] = 0.f;
  for_range_c<0, 9>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int u = (int)(e & 3u), v = (int)((e >> 2) & 3u);   // = offset + 1: g[u][v]
    const int wt = (int)(e >> 4);
    const float gu = wino_g(wave, u);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float* wp = p.w + (size_t)wt * p.tap_stride + (size_t)(p.n_off + 16 * nb + fn) * 32 + 2 * fk;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x2w w2 = *reinterpret_cast<const f32x2w*>(wp + 8 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float c = gu * wino_g(j, v);
          ub[j][nb][2 * q] = fmaf(c, w2[0], ub[j][nb][2 * q]);
          ub[j][nb][2 * q + 1] = fmaf(c, w2[1], ub[j][nb][2 * q + 1]);
        }
      }
    }
  });

  // ---- patch slots of this thread (constant across units) ----
  int pp_rel[P_PASSES], pp_lds[P_PASSES], pp_rc[P_PASSES];
#pragma unroll
  for (int i = 0; i < P_PASSES; ++i) {
    const int slot = tid + 256 * i;
    const bool valid = slot < Q_SLOTS;
    const int pix = valid ? slot >> 3 : 0, seg = slot & 7;
    const int prow = pix / PW, pcol = pix - prow * PW;
    pp_rel[i] = ((prow * W + pcol) * 32 + seg * 4) * 4;
    pp_lds[i] = pix * LDA + seg * 4;
    pp_rc[i] = valid ? (prow | (pcol << 8)) : (1 << 20);
  }
  f32x4 pr[P_PASSES];
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  float cs = 1.f;
  unsigned okm = 0;
  int n_coef = -1;     // image whose activation / BSTATS coefficients are in the registers
  auto tile_pos = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % tiles_x;
    const int r = tile / tiles_x;
    const int ty = r % tiles_y;
    n = r / tiles_y; y0 = ty * C32_TH; x0 = tx * C32_TW;
  };
  // unit u of a tile: rows 4 (u >> 1) .., columns 16 (u & 1) ..
  auto load_patch = [&](int n, int yu, int xu) {
    const int base = ((n * H + yu - 1) * W + xu - 1) * 128;
    okm = 0;
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      const int prow = pp_rc[i] & 0xff, pcol = pp_rc[i] >> 8;
      const bool ok = (unsigned)(yu - 1 + prow) < (unsigned)H && (unsigned)(xu - 1 + pcol) < (unsigned)W;
      okm |= (ok ? 1u : 0u) << i;
      const unsigned off = ok ? (unsigned)(base + pp_rel[i]) : 0x80000000u;
      pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      if (FUSED) pr[i] = act4(pr[i], ca, cb, cs, (okm >> i) & 1u);
      if (256 * (i + 1) <= Q_SLOTS || tid + 256 * i < Q_SLOTS)
        *reinterpret_cast<f32x4*>(Pb + pp_lds[i]) = pr[i];
    }
  };
  // activation coefficients of image n (the patch in the registers is stored with them)
  auto load_act = [&](int n) {
    if (FUSED) {
      if (p.act0_alpha) {   // uniform
        const size_t o = (size_t)n * 32 + (tid & 7) * 4;
        ca = *reinterpret_cast<const f32x4*>(p.act0_alpha + o);
        cb = *reinterpret_cast<const f32x4*>(p.act0_beta + o);
        cs = p.slope;
      }
    }
  };

  // ---- input transform: thread -> (channel pair cp, tile slot tt = 4 wave + (lane >> 4)) ----
  // slot tt sits at tile row (tt & 3) >> 1, tile column 4 (tt & 1) + (tt >> 2): the 16-lane
  // groups of a half wave read pixels 8 apart (conflict-free 8-byte reads)
  const int cp = tid & 15, tt = tid >> 4;
  const int t_ty = (tt & 3) >> 1, t_tx = 4 * (tt & 1) + (tt >> 2);
  const unsigned t_srca = lds_addr(Pb + ((2 * t_ty) * PW + 2 * t_tx) * LDA + 2 * cp);
  float* const t_dst = Vs + tt * W32_VP + 2 * cp;
  auto transform = [&]() {
    f32x2v d[4][4];
    for_range_c<0, 16>([&](auto ic) {
      constexpr int r = decltype(ic)::value / 4, c = decltype(ic)::value % 4;
      d[r][c] = lds_rd64<((r * PW + c) * LDA) * 4>(t_srca);
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_wait<0>(d[r][0], d[r][1], d[r][2], d[r][3]);
    f32x2v t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      t[0][c] = d[0][c] - d[2][c];
      t[1][c] = d[1][c] + d[2][c];
      t[2][c] = d[2][c] - d[1][c];
      t[3][c] = d[1][c] - d[3][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float* dst = t_dst + (4 * a) * (16 * W32_VP);
      *reinterpret_cast<f32x2v*>(dst) = t[a][0] - t[a][2];
      *reinterpret_cast<f32x2v*>(dst + 16 * W32_VP) = t[a][1] + t[a][2];
      *reinterpret_cast<f32x2v*>(dst + 2 * 16 * W32_VP) = t[a][2] - t[a][1];
      *reinterpret_cast<f32x2v*>(dst + 3 * 16 * W32_VP) = t[a][1] - t[a][3];
    }
  };

  const unsigned a_addr = lds_addr(Vs + (4 * wave * 16 + fn) * W32_VP + 2 * fk);   // + j * 16 * VP + 8 q
  // Exchange Z[xi row][b][r][tile group fk][column c32 ^ 16 (fk & 1)] (c32 = 16 nb + fn): the
  // readers (column li of tile group 2 lh + e) then read 32 consecutive floats per half wave -
  // conflict-free; the first layout ([nb][fk][fn]) put columns li and li + 16 on the same bank
  // (2-way on every read).  The writers' half waves (fk = 0, 1) land on the two halves of the
  // banks thanks to the XOR - conflict-free as well.
  float* const z_dst0 = Zs + (wave * 2) * 512 + fk * 32 + (fn + 16 * (fk & 1));         // nb = 0
  float* const z_dst1 = Zs + (wave * 2) * 512 + fk * 32 + (fn + 16 * (1 - (fk & 1)));   // nb = 1; + b * 512 + r * 128
  // row pass + epilogue: wave -> slots with r = wave; lane (li, lh) -> column li, fk = 2 lh + e
  const float* const z_src0 = Zs + wave * 128 + 64 * lh + li;                 // e = 0; + (i * 2 + b) * 512
  const float* const z_src1 = Zs + wave * 128 + 64 * lh + 32 + (li ^ 16);     // e = 1
  const int o_ty = wave >> 1, o_tx0 = 4 * (wave & 1) + 2 * lh;
  const bool summaries = FUSED ? p.stats != nullptr : p.bs_partial != nullptr;   // uniform
  const float bv = (FUSED && p.bias) ? p.bias[li] : 0.f;
  BwdCoef cf{};

  {
    int n, y0, x0;
    tile_pos(t_first, n, y0, x0);
    load_patch(n, y0, x0);
    load_act(n);
    n_coef = n;
    if (!FUSED && p.bs_partial) cf = bwd_coef(p, n, li);
    store_patch();
  }
  __syncthreads();

  for (int tile = t_first; tile < t_end; tile += t_stride) {
    const int nxt = tile + t_stride;
    const bool more = nxt < t_end;
    int n, y0, x0;
    tile_pos(tile, n, y0, x0);
    // running (mean, M2) of the four units (FUSED) / (S1, S2) (BSTATS) of this lane's column
    float um[4], uq[4];
    float s1 = 0.f, s2 = 0.f;
    for_range_c<0, 4>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int yu = y0 + 4 * (u >> 1), xu = x0 + 16 * (u & 1);
      // the next unit's patch: into the registers now, to LDS after this unit's epilogue
      bool have_next = true;
      int nn = n;
      if constexpr (u < 3) {
        load_patch(n, y0 + 4 * ((u + 1) >> 1), x0 + 16 * ((u + 1) & 1));
      } else {
        have_next = more;
        if (more) {   // uniform
          int ny, nx;
          tile_pos(nxt, nn, ny, nx);
          load_patch(nn, ny, nx);
        }
      }
      float yv[8], oldv[8];
      if (!FUSED) {
        if (p.bs_partial || p.accumulate) {   // uniform
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int e = k >> 2, a = (k >> 1) & 1, b = k & 1;
            const size_t pix = ((size_t)n * H + yu + 2 * o_ty + a) * W + xu + 2 * (o_tx0 + e) + b;
            if (p.bs_partial) yv[k] = p.bs_y[pix * p.ldo + li];
            if (p.accumulate) oldv[k] = p.out[pix * p.ldo + li];
          }
        }
      }
      transform();
      __syncthreads();
      f32x4 acc[4][2];
      f32x2w af[2][4];
      // hand-issued ds_read_b64 (lds_asm.h): left to the compiler, the reads of q and q + 1 fuse
      // into ds_read2_b64, which the LDS serves in 16-lane groups over 32 banks - rows fn and
      // fn + 8 of the 36-float pitch then collide (2-way) and the instruction takes 8 array
      // cycles where two plain reads take 4 (the 0.27 conflict cycles per active LDS cycle of
      // the round-3 PMC table)
      auto frag = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        for_range_c<0, 4>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          af[q & 1][j] = lds_rd64<(j * 16 * W32_VP + 8 * q) * 4>(a_addr);
        });
      };
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[j][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      frag(std::integral_constant<int, 0>{});
      for_range_c<0, 4>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q + 1 < 4) frag(std::integral_constant<int, q + 1>{});
        lds_wait<(q + 1 < 4 ? 4 : 0)>(af[q & 1][0], af[q & 1][1], af[q & 1][2], af[q & 1][3]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q & 1][j][e], ub[j][nb][2 * q + e],
                                                                acc[j][nb], 0, 0, 0);
      });
      // column pass of A^T M A over this wave's row of xi
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r];
          float* const zd = nb == 0 ? z_dst0 : z_dst1;
          zd[r * 128] = m0 + (m1 + m2);
          zd[512 + r * 128] = (m1 - m2) - m3;
        }
      // the next unit's patch (in the registers since the top of this unit): its buffer was last
      // read by this unit's transform, one barrier back, and is next read behind the one below
      if (have_next) {   // uniform
        if (nn != n_coef) {   // uniform: a new image - its coefficients (rare: once per image)
          load_act(nn);
          n_coef = nn;
        }
        store_patch();
      }
      __syncthreads();
      // row pass: Y[a][b] = sum_i A^T[a][i] Z[i][b]
      float ov[8];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float z[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int b = 0; b < 2; ++b) z[i][b] = (e == 0 ? z_src0 : z_src1)[(i * 2 + b) * 512];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          ov[4 * e + b] = z[0][b] + (z[1][b] + z[2][b]);
          ov[4 * e + 2 + b] = (z[1][b] - z[2][b]) - z[3][b];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = k >> 2, a = (k >> 1) & 1, b = k & 1;
        const size_t pix = ((size_t)n * H + yu + 2 * o_ty + a) * W + xu + 2 * (o_tx0 + e) + b;
        if (FUSED) ov[k] += bv;
        else if (p.accumulate) ov[k] += oldv[k];   // uniform
        p.out[pix * p.ldo + li] = ov[k];
      }
      if (FUSED) {
        if (p.stats) {   // uniform: (mean, M2) of this unit's 8 values
          float sm = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) sm += ov[k];
          const float mean = sm * 0.125f;
          float m2 = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) { const float dd = ov[k] - mean; m2 = fmaf(dd, dd, m2); }
          um[u] = mean; uq[u] = m2;
        }
      } else if (p.bs_partial) {   // uniform
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float z = fmaf(yv[k], cf.A, cf.B0);
          const float gz = ov[k] * cf.mk * (z > 0.f ? 1.f : p.slope);
          s1 += gz;
          s2 = fmaf(gz, (yv[k] - cf.mu) * cf.rs, s2);
        }
      }
      if constexpr (u == 3) {
        if (FUSED) {
          if (p.stats) {   // uniform
            wf_merge_eq(um[0], uq[0], um[1], uq[1], 8.f);
            wf_merge_eq(um[2], uq[2], um[3], uq[3], 8.f);
            wf_merge_eq(um[0], uq[0], um[2], uq[2], 16.f);
            const float mb = __shfl_xor(um[0], 32, 64), qb = __shfl_xor(uq[0], 32, 64);
            wf_merge_eq(um[0], uq[0], mb, qb, 32.f);
            if (lh == 0) red[wave * 32 + li] = float2{um[0], uq[0]};
          }
        } else if (p.bs_partial) {   // uniform
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (lh == 0) red[wave * 32 + li] = float2{s1, s2};
        }
      }
      if constexpr (u == 3) {
        // (the BSTATS coefficients of the next tile's image; the statistics above used this one's)
        if (!FUSED && p.bs_partial && have_next && nn != n) cf = bwd_coef(p, nn, li);   // uniform
        __syncthreads();   // the summaries in `red`; units 0-2 need no third barrier
      }
    });
    if (summaries && tid < 32) {   // red is rewritten eight barriers from now
      const float2* rr = red + tid;
      const size_t dst = ((size_t)n * (tiles_x * tiles_y) + (y0 / C32_TH) * tiles_x + (x0 >> 5)) * 32 + tid;
      if (FUSED) {
        float mean[2], m2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          mean[k] = rr[(2 * k) * 32].x; m2[k] = rr[(2 * k) * 32].y;
          wf_merge_eq(mean[k], m2[k], rr[(2 * k + 1) * 32].x, rr[(2 * k + 1) * 32].y, 64.f);
        }
        wf_merge_eq(mean[0], m2[0], mean[1], m2[1], 128.f);
        p.stats[dst] = float2{mean[0], m2[0]};
      } else {
        float a = rr[0].x, b = rr[0].y;
#pragma unroll
        for (int k = 1; k < 4; ++k) { a += rr[k * 32].x; b += rr[k * 32].y; }
        p.bs_partial[dst] = float2{a, b};
      }
    }
  }
}


// ---------------------------------------------------------------------------
// conv3x3(cat(upsample2x(act(low)), act(skip))) -> 32 channels at full resolution (the first
// convolution of the last decoder stage: 64 low-resolution + 32 skip channels; 982 us in the
// direct patch kernel, the largest launch of the step) by the same recipe: K = 96 is THREE
// register-resident chunks of 32 input channels.  One workgroup of FOUR waves per CU, so that a
// wave owns 512 registers: wave w = row w of the xi grid keeps U of all three chunks for all 32
// output columns (192 registers) and ONE accumulator set; per 4 x 16-pixel unit the chunks pass through the same patch and V
// buffers (stage -> transform -> products, two barriers each), then one exchange + epilogue.
// An up-sampled chunk never exists at full resolution: bilinear up-sampling (align_corners =
// False at exactly 2x: weights 0.25 / 0.75 by the parity of the pixel, indices clamped at the
// border) is linear, so it is folded into the input transform - the loader stages the ACTIVATED
// 4 x 10 low-resolution pixels under the unit and the transform reads a 3 x 3 window of them
// (see transform_up; the zero padding of the convolution applies to the up-sampled tensor: the
// interpolated rows / columns outside the image are zero).  (A first version blended the four
// taps per patch pixel in the loader: 16 loads and ~400 VALU instructions per thread and chunk,
// 1062 us - hardly better than the direct kernel.)
// ---------------------------------------------------------------------------
constexpr int WU_PASSES = Q_PASSES;                          // 4 passes of 256 threads
constexpr int WU_LSLOTS = 4 * 10 * 8;                        // f32x4 slots of the low-resolution patch
constexpr size_t WU_LDS = Q_LDS + 4 * 10 * C32_LDA * sizeof(float);

__global__ __launch_bounds__(256, 1) void conv_wino_up32_kernel(const IgemmParams p, int ntiles) {
  constexpr int LDA = C32_LDA, PW = Q_PW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Pb = smem;
  float* const Vs = smem + Q_PPIX * LDA;
  float* const Zs = Vs + Q_V;
  float2* const red = reinterpret_cast<float2*>(Zs + Q_Z);
  float* const Lb = reinterpret_cast<float*>(red + 4 * 32);   // low-resolution patch: 4 x 10 pixels

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row of the xi grid
  const int li = lane & 31, lh = lane >> 5;
  const int H = p.Hin, W = p.Win, hl = H >> 1, wl = W >> 1;
  const int tiles_x = W / C32_TW, tiles_y = H / C32_TH;
  const int G = gridDim.x;
  int t_first, t_stride, t_end;
  if ((ntiles & 7) == 0 && (G & 7) == 0) {
    const int per = ntiles >> 3, xcd = blockIdx.x & 7;
    t_first = xcd * per + (blockIdx.x >> 3);
    t_stride = G >> 3;
    t_end = (xcd + 1) * per;
  } else {
    t_first = blockIdx.x; t_stride = G; t_end = ntiles;
  }
  if (t_first >= t_end) return;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1), 0, (int)p.src1_bytes, 0x00020000);

  // ---- U = G g G^T of the three chunks: row `wave` of the xi grid, both column halves ----
  const int fn = lane & 15, fk = lane >> 4;
  float ub[3][4][2][8];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int k = 0; k < 8; ++k) ub[c][j][nb][k] = 0.f;
  // (one tap at a time: unrolled, the compiler hoists all 216 weight loads above the arithmetic)
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int u = (int)(e & 3u), v = (int)((e >> 2) & 3u);
    const int wt = (int)(e >> 4);
    const float gu = wino_g(wave, u);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float* wp = p.w + (size_t)wt * p.tap_stride + (size_t)(p.n_off + 16 * nb + fn) * 96 + 2 * fk;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2w w2 = *reinterpret_cast<const f32x2w*>(wp + 32 * c + 8 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float cf = gu * wino_g(j, v);
            ub[c][j][nb][2 * q] = fmaf(cf, w2[0], ub[c][j][nb][2 * q]);
            ub[c][j][nb][2 * q + 1] = fmaf(cf, w2[1], ub[c][j][nb][2 * q + 1]);
          }
        }
    }
  }

  // ---- staging slots of this thread: skip chunk = the 6 x 18 patch (4 passes), up-sampled chunk
  //      = the 4 x 10 LOW-resolution pixels under it (2 passes); channel quad seg ----
  int pp_lds[WU_PASSES], pp_rc[WU_PASSES];
#pragma unroll
  for (int i = 0; i < WU_PASSES; ++i) {
    const int slot = tid + 256 * i;
    const bool valid = slot < Q_SLOTS;
    const int pix = valid ? slot >> 3 : 0;
    const int prow = pix / PW, pcol = pix - prow * PW;
    pp_lds[i] = pix * LDA + (slot & 7) * 4;
    pp_rc[i] = valid ? (prow | (pcol << 8)) : (1 << 20);
  }
  int lp_lds[2], lp_rc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = tid + 256 * i;
    const bool valid = slot < WU_LSLOTS;
    const int pix = valid ? slot >> 3 : 0;
    const int lr = pix / 10, lc = pix - lr * 10;
    lp_lds[i] = pix * LDA + (slot & 7) * 4;
    lp_rc[i] = valid ? (lr | (lc << 8)) : (1 << 20);
  }
  const int seg4 = (tid & 7) * 4;
  f32x4 pr[WU_PASSES];
  f32x4 ca, cb;
  unsigned okm = 0;
  auto tile_pos = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % tiles_x;
    const int r = tile / tiles_x;
    const int ty = r % tiles_y;
    n = r / tiles_y; y0 = ty * C32_TH; x0 = tx * C32_TW;
  };
  // chunk c of the unit at (n, yu, xu): loads into the registers
  auto load_chunk = [&](int c, int n, int yu, int xu) {   // c uniform
    okm = 0;
    if (c < 2) {   // low-resolution rows yu / 2 - 1 .. + 2, columns xu / 2 - 1 .. + 8
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int Y = (yu >> 1) - 1 + (lp_rc[i] & 0xff), X = (xu >> 1) - 1 + (lp_rc[i] >> 8);
        const bool ok = (unsigned)Y < (unsigned)hl && (unsigned)X < (unsigned)wl;
        okm |= (ok ? 1u : 0u) << i;
        const unsigned off = (unsigned)((((n * hl + Y) * wl + X) * 64 + 32 * c + seg4) * 4);
        pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, ok ? off : 0x80000000u, 0, 0));
      }
      const size_t o = (size_t)n * 64 + 32 * c + seg4;
      ca = *reinterpret_cast<const f32x4*>(p.act0_alpha + o);
      cb = *reinterpret_cast<const f32x4*>(p.act0_beta + o);
    } else {
#pragma unroll
      for (int i = 0; i < WU_PASSES; ++i) {
        const int Y = yu - 1 + (pp_rc[i] & 0xff), X = xu - 1 + (pp_rc[i] >> 8);
        const bool ok = (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
        okm |= (ok ? 1u : 0u) << i;
        const unsigned off = (unsigned)((((n * H + Y) * W + X) * 32 + seg4) * 4);
        pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, ok ? off : 0x80000000u, 0, 0));
      }
      const size_t o = (size_t)n * 32 + seg4;
      ca = *reinterpret_cast<const f32x4*>(p.act1_alpha + o);
      cb = *reinterpret_cast<const f32x4*>(p.act1_beta + o);
    }
  };
  auto store_chunk = [&](int c) {   // c uniform: activate, registers -> LDS
    if (c < 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (tid + 256 * i < WU_LSLOTS)
          *reinterpret_cast<f32x4*>(Lb + lp_lds[i]) = act4(pr[i], ca, cb, p.slope, (okm >> i) & 1u);
    } else {
#pragma unroll
      for (int i = 0; i < WU_PASSES; ++i)
        if (256 * (i + 1) <= Q_SLOTS || tid + 256 * i < Q_SLOTS)
          *reinterpret_cast<f32x4*>(Pb + pp_lds[i]) = act4(pr[i], ca, cb, p.slope, (okm >> i) & 1u);
    }
  };

  // ---- input transform, products, exchange: the mappings of conv_wino32q_kernel ----
  const int cp = tid & 15, tt = tid >> 4;
  const int t_ty = (tt & 3) >> 1, t_tx = 4 * (tt & 1) + (tt >> 2);
  const unsigned t_srca = lds_addr(Pb + ((2 * t_ty) * PW + 2 * t_tx) * LDA + 2 * cp);
  float* const t_dst = Vs + tt * W32_VP + 2 * cp;
  auto transform = [&]() {
    f32x2v d[4][4];
    for_range_c<0, 16>([&](auto ic) {
      constexpr int r = decltype(ic)::value / 4, c = decltype(ic)::value % 4;
      d[r][c] = lds_rd64<((r * PW + c) * LDA) * 4>(t_srca);
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_wait<0>(d[r][0], d[r][1], d[r][2], d[r][3]);
    f32x2v t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      t[0][c] = d[0][c] - d[2][c];
      t[1][c] = d[1][c] + d[2][c];
      t[2][c] = d[2][c] - d[1][c];
      t[3][c] = d[1][c] - d[3][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float* dst = t_dst + (4 * a) * (16 * W32_VP);
      *reinterpret_cast<f32x2v*>(dst) = t[a][0] - t[a][2];
      *reinterpret_cast<f32x2v*>(dst + 16 * W32_VP) = t[a][1] + t[a][2];
      *reinterpret_cast<f32x2v*>(dst + 2 * 16 * W32_VP) = t[a][2] - t[a][1];
      *reinterpret_cast<f32x2v*>(dst + 3 * 16 * W32_VP) = t[a][1] - t[a][3];
    }
  };
  // The same for an up-sampled chunk, WITHOUT building the up-sampled patch: d = Ry L Rx^T with L
  // the 3 x 3 low-resolution window under the tile (rows R - 1 .. R + 1 for output rows 2R, 2R+1)
  // and Ry / Rx the interpolation of its four input rows / columns 2R - 1 .. 2R + 2:
  //   2R-1: 0.75 L0 + 0.25 L1 (zero when outside the image: the convolution's padding)
  //   2R  : 0.25 L0 + 0.75 L1 (L1 alone in the first row: the index clamps)
  //   2R+1: 0.75 L1 + 0.25 L2 (L1 alone in the last row)
  //   2R+2: 0.25 L1 + 0.75 L2 (zero when outside the image)
  // so V = B^T (Ry L Rx^T) B costs 9 reads instead of 16 and no blending in the loader.
  const unsigned l_srca = lds_addr(Lb + (t_ty * 10 + t_tx) * LDA + 2 * cp);
  auto transform_up = [&](int yu, int xu) {
    const int R = (yu >> 1) + t_ty, C = (xu >> 1) + t_tx;
    const bool fr = R == 0, lr = R == hl - 1, fc = C == 0, lc = C == wl - 1;
    // coefficient pairs of the four interpolated rows / columns
    const float ra0 = fr ? 0.f : 0.75f, rb0 = fr ? 0.f : 0.25f, ra1 = fr ? 0.f : 0.25f, rb1 = fr ? 1.f : 0.75f;
    const float ra2 = lr ? 1.f : 0.75f, rb2 = lr ? 0.f : 0.25f, ra3 = lr ? 0.f : 0.25f, rb3 = lr ? 0.f : 0.75f;
    const float ka0 = fc ? 0.f : 0.75f, kb0 = fc ? 0.f : 0.25f, ka1 = fc ? 0.f : 0.25f, kb1 = fc ? 1.f : 0.75f;
    const float ka2 = lc ? 1.f : 0.75f, kb2 = lc ? 0.f : 0.25f, ka3 = lc ? 0.f : 0.25f, kb3 = lc ? 0.f : 0.75f;
    f32x2v L[3][3];
    for_range_c<0, 9>([&](auto ic) {
      constexpr int r = decltype(ic)::value / 3, c = decltype(ic)::value % 3;
      L[r][c] = lds_rd64<((r * 10 + c) * LDA) * 4>(l_srca);
    });
#pragma unroll
    for (int r = 0; r < 3; ++r) lds_wait<0>(L[r][0], L[r][1], L[r][2]);
    f32x2v t[4][3];   // B^T Ry L
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const f32x2v u0 = L[0][c] * ra0 + L[1][c] * rb0, u1 = L[0][c] * ra1 + L[1][c] * rb1;
      const f32x2v u2 = L[1][c] * ra2 + L[2][c] * rb2, u3 = L[1][c] * ra3 + L[2][c] * rb3;
      t[0][c] = u0 - u2;
      t[1][c] = u1 + u2;
      t[2][c] = u2 - u1;
      t[3][c] = u1 - u3;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x2v w0 = t[a][0] * ka0 + t[a][1] * kb0, w1 = t[a][0] * ka1 + t[a][1] * kb1;
      const f32x2v w2 = t[a][1] * ka2 + t[a][2] * kb2, w3 = t[a][1] * ka3 + t[a][2] * kb3;
      float* dst = t_dst + (4 * a) * (16 * W32_VP);
      *reinterpret_cast<f32x2v*>(dst) = w0 - w2;
      *reinterpret_cast<f32x2v*>(dst + 16 * W32_VP) = w1 + w2;
      *reinterpret_cast<f32x2v*>(dst + 2 * 16 * W32_VP) = w2 - w1;
      *reinterpret_cast<f32x2v*>(dst + 3 * 16 * W32_VP) = w1 - w3;
    }
  };
  const unsigned a_addr = lds_addr(Vs + (4 * wave * 16 + fn) * W32_VP + 2 * fk);
  // (the exchange layout of conv_wino32q_kernel: conflict-free on both sides)
  float* const z_dst0 = Zs + (wave * 2) * 512 + fk * 32 + (fn + 16 * (fk & 1));
  float* const z_dst1 = Zs + (wave * 2) * 512 + fk * 32 + (fn + 16 * (1 - (fk & 1)));
  const float* const z_src0 = Zs + wave * 128 + 64 * lh + li;
  const float* const z_src1 = Zs + wave * 128 + 64 * lh + 32 + (li ^ 16);
  const int o_ty = wave >> 1, o_tx0 = 4 * (wave & 1) + 2 * lh;
  const float bv = p.bias ? p.bias[li] : 0.f;

  {
    int n, y0, x0;
    tile_pos(t_first, n, y0, x0);
    load_chunk(0, n, y0, x0);
    store_chunk(0);
  }
  __syncthreads();

  for (int tile = t_first; tile < t_end; tile += t_stride) {
    const int nxt = tile + t_stride;
    const bool more = nxt < t_end;
    int n, y0, x0;
    tile_pos(tile, n, y0, x0);
    // (a runtime loop over the four units: unrolled, the body's twelve chunk steps cost registers)
    float pm = 0.f, pq = 0.f, hm = 0.f, hq = 0.f;   // running (mean, M2): pair of units, half tile
#pragma unroll 1
    for (int u = 0; u < 4; ++u) {
      const int yu = y0 + 4 * (u >> 1), xu = x0 + 16 * (u & 1);
      f32x4 acc[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[j][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      for_range_c<0, 3>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        // the next chunk (of this unit, of the next unit, of the next tile): its loads go out now
        bool have_next = true;
        if constexpr (c < 2) {
          load_chunk(c + 1, n, yu, xu);
        } else if (u < 3) {   // uniform
          load_chunk(0, n, y0 + 4 * ((u + 1) >> 1), x0 + 16 * ((u + 1) & 1));
        } else {
          have_next = more;
          if (more) {   // uniform
            int nn, ny, nx;
            tile_pos(nxt, nn, ny, nx);
            load_chunk(0, nn, ny, nx);
          }
        }
        if constexpr (c < 2) transform_up(yu, xu); else transform();
        __syncthreads();
        f32x2w af[2][4];
        auto frag = [&](auto qc) {   // hand-issued ds_read_b64: see conv_wino32q_kernel
          constexpr int q = decltype(qc)::value;
          for_range_c<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            af[q & 1][j] = lds_rd64<(j * 16 * W32_VP + 8 * q) * 4>(a_addr);
          });
        };
        frag(std::integral_constant<int, 0>{});
        for_range_c<0, 4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          if constexpr (q + 1 < 4) frag(std::integral_constant<int, q + 1>{});
          lds_wait<(q + 1 < 4 ? 4 : 0)>(af[q & 1][0], af[q & 1][1], af[q & 1][2], af[q & 1][3]);
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int nb = 0; nb < 2; ++nb)
                acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q & 1][j][e], ub[c][j][nb][2 * q + e],
                                                                  acc[j][nb], 0, 0, 0);
        });
        if constexpr (c == 2) {   // column pass of A^T M A over this wave's row of xi
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float m0 = acc[0][nb][r], m1 = acc[1][nb][r], m2 = acc[2][nb][r], m3 = acc[3][nb][r];
              float* const zd = nb == 0 ? z_dst0 : z_dst1;
              zd[r * 128] = m0 + (m1 + m2);
              zd[512 + r * 128] = (m1 - m2) - m3;
            }
        }
        if (have_next) store_chunk(c < 2 ? c + 1 : 0);   // uniform
        __syncthreads();
      });
      // row pass: Y[a][b] = sum_i A^T[a][i] Z[i][b]
      float ov[8];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float z[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int b = 0; b < 2; ++b) z[i][b] = (e == 0 ? z_src0 : z_src1)[(i * 2 + b) * 512];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          ov[4 * e + b] = z[0][b] + (z[1][b] + z[2][b]) + bv;
          ov[4 * e + 2 + b] = (z[1][b] - z[2][b]) - z[3][b] + bv;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e = k >> 2, a = (k >> 1) & 1, b = k & 1;
        const size_t pix = ((size_t)n * H + yu + 2 * o_ty + a) * W + xu + 2 * (o_tx0 + e) + b;
        p.out[pix * p.ldo + li] = ov[k];
      }
      if (p.stats) {   // uniform: (mean, M2) of this unit's 8 values
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sm += ov[k];
        const float mean = sm * 0.125f;
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float dd = ov[k] - mean; m2 = fmaf(dd, dd, m2); }
        // equal-count merges: units (0, 1) and (2, 3) into pairs, then the two pairs
        if ((u & 1) == 0) { pm = mean; pq = m2; }   // uniform
        else {
          wf_merge_eq(pm, pq, mean, m2, 8.f);
          if (u == 1) { hm = pm; hq = pq; }
          else wf_merge_eq(hm, hq, pm, pq, 16.f);
        }
      }
    }
    if (p.stats) {   // uniform
      const float mb = __shfl_xor(hm, 32, 64), qb = __shfl_xor(hq, 32, 64);
      wf_merge_eq(hm, hq, mb, qb, 32.f);
      if (lh == 0) red[wave * 32 + li] = float2{hm, hq};
      __syncthreads();
      if (tid < 32) {
        const float2* rr = red + tid;
        const size_t dst = ((size_t)n * (tiles_x * tiles_y) + (y0 / C32_TH) * tiles_x + (x0 >> 5)) * 32 + tid;
        float mean[2], m2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          mean[k] = rr[(2 * k) * 32].x; m2[k] = rr[(2 * k) * 32].y;
          wf_merge_eq(mean[k], m2[k], rr[(2 * k + 1) * 32].x, rr[(2 * k + 1) * 32].y, 64.f);
        }
        wf_merge_eq(mean[0], m2[0], mean[1], m2[1], 128.f);
        p.stats[dst] = float2{mean[0], m2[0]};
      }
    }
  }
}

}  // namespace

// choice between the two forms of these layers (unet_set_c32_winograd), per calling thread: the
// host code sets it for its own calls, so two models on two threads (or autograd's backward
// thread) cannot change each other's dispatch
int& c32_winograd_flag() {
  static thread_local int on = 1;
  return on;
}

// 3x3 stride-1, 32 input and 32 output channels, image tiles as 8 x 32 pixels
bool c32_applicable(const IgemmParams& p) {
  return p.ntaps == 9 && p.tap_cstride == 0 && p.src0_pitch == 0 && p.sout == 1 &&
         p.Hl == p.Hin && p.Wl == p.Win && p.Hl == p.Hout && p.Wl == p.Wout && p.Hin % 8 == 0 &&
         p.Win % 32 == 0 && p.C0 == 32 && p.C1 == 0 && p.Ncols == 32 &&
         (long long)p.N * p.Hin * p.Win * 128 < (1LL << 31);
}

// fused != 0: the fused-layer forward (activation on load, statistics into p.stats when set);
// else the data gradient (p.accumulate, BSTATS epilogue when p.bs_partial is set).  *tile_px
// receives the pixels per statistics / reduction tile (256) or 0 when none was emitted.
int launch_c32(const IgemmParams& p0, int fused, hipStream_t stream, int* tile_px) {
  IgemmParams p = p0;
  const int tiles256 = p.Hin * p.Win / 256;
  if (tile_px) *tile_px = 0;
  if (fused) {
    p.bs_partial = nullptr;
    if (p.stats && tile_px) { p.stats_tiles = tiles256; *tile_px = 256; }
    else p.stats = nullptr;
  } else {
    p.stats = nullptr;
    if (p.bs_partial && tile_px) { p.bs_tiles = tiles256; p.bs_tile0 = 0; *tile_px = 256; }
    else p.bs_partial = nullptr;
  }
  const int ntiles = p.N * (p.Hin / C32_TH) * (p.Win / C32_TW);
  const int grid = ntiles < 256 ? ntiles : 256;
  // Winograd form: when the launch fills the chip with its two workgroups per CU (the same kind
  // of rule as for the wider layers' Winograd kernels), or always on request (tests)
  if (c32_winograd_flag() == 2 || (c32_winograd_flag() == 1 && ntiles >= 512)) {
    const int grid2 = ntiles < 512 ? ntiles : 512;
    if (fused) {
      auto kern = conv_wino32q_kernel<true>;
      UNET_SET_DYN_LDS(kern, Q_LDS);
      hipLaunchKernelGGL(kern, dim3((unsigned)grid2), dim3(256), Q_LDS, stream, p, ntiles);
    } else {
      auto kern = conv_wino32q_kernel<false>;
      UNET_SET_DYN_LDS(kern, Q_LDS);
      hipLaunchKernelGGL(kern, dim3((unsigned)grid2), dim3(256), Q_LDS, stream, p, ntiles);
    }
    UNET_CHECK_LAUNCH("conv_wino32q");
    return UNET_OK;
  }
  if (fused) {
    auto kern = conv_c32_kernel<true>;
    UNET_SET_DYN_LDS(kern, C32_LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C32_LDS, stream, p, ntiles);
  } else {
    auto kern = conv_c32_kernel<false>;
    UNET_SET_DYN_LDS(kern, C32_LDS);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), C32_LDS, stream, p, ntiles);
  }
  UNET_CHECK_LAUNCH("conv_c32");
  return UNET_OK;
}

}  // namespace unet_conv

// 1 (default): the 32 -> 32 channel layers run the Winograd F(2x2, 3x3) kernel when the launch
// has at least 512 tiles of 8 x 32 pixels (two workgroups per CU), 2: always, 0: never (the
// direct kernel).  Returns the previous setting.
extern "C" int unet_set_c32_winograd(int on) {
  const int prev = unet_conv::c32_winograd_flag();
  unet_conv::c32_winograd_flag() = on < 0 ? 0 : (on > 2 ? 2 : on);
  return prev;
}

extern "C" int unet_conv_c32_is_winograd(int N, int H, int W, int Cin, int Cout, int stride) {
  const int f = unet_conv::c32_winograd_flag();
  if (!(f && stride == 1 && Cin == 32 && Cout == 32 && N > 0 && H > 0 && W > 0 && H % 8 == 0 &&
        W % 32 == 0 && (long long)N * H * W * 128 < (1LL << 31)))
    return 0;
  return f == 2 || (long long)N * (H / 8) * (W / 32) >= 512;
}

namespace unet_conv {
// (64 up-sampled + 32 skip) -> 32 channels, both sources activated on load, whole 8 x 32 tiles,
// a tile for every CU: the Winograd form of launch_patch_up<32, ...> (conv_patch.hip)
bool wino_up32_applicable(const IgemmParams& p) {
  const int f = c32_winograd_flag();
  if (!(f && p.C0 == 64 && p.C1 == 32 && p.Ncols == 32 && p.ntaps == 9 && p.Hin % 8 == 0 &&
        p.Win % 32 == 0 && p.act0_alpha && p.act1_alpha && p.ldo == 32 && !p.accumulate &&
        (long long)p.N * p.Hin * p.Win * 128 < (1LL << 31)))
    return false;
  return f == 2 || (long long)p.N * (p.Hin / 8) * (p.Win / 32) >= 256;
}
int launch_wino_up32(const IgemmParams& p, hipStream_t stream) {
  const int ntiles = p.N * (p.Hin / C32_TH) * (p.Win / C32_TW);
  const int grid = ntiles < 256 ? ntiles : 256;
  auto kern = conv_wino_up32_kernel;
  UNET_SET_DYN_LDS(kern, WU_LDS);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), WU_LDS, stream, p, ntiles);
  UNET_CHECK_LAUNCH("conv_wino_up32");
  return UNET_OK;
}
}  // namespace unet_conv

// conv_patch.hip — patch-staged 3x3 stride-1 convolution (forward and data gradient) on the
// matrix cores: the input patch of a TH x 32-pixel output tile is staged into LDS once per
// channel chunk and serves all nine taps.  fp32 form (v_mfma_f32_32x32x2_f32: the dominant
// kernel of the train step) and split-bf16 form (three bf16 planes, six
// v_mfma_f32_32x32x16_bf16 products per multiply: fp32-class accuracy at 2.67x the matrix rate).
//
// Replaces nn.Conv2d forward / aten::convolution_backward(data) of
// Our_UNet/models/unet.py:106-115 for every stride-1 layer whose image tiles as TH x 32.
#include "conv_params.h"
#include <stdlib.h>

#include <utility>

namespace unet_conv {
namespace {

// compile-time loop: f(integral_constant<int, I>) for I in [B, E)
template <int B, int... I, typename F>
__device__ __forceinline__ void for_range_p_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, B + I>{}), ...);
}
template <int B, int E, typename F>
__device__ __forceinline__ void for_range_p(F&& f) {
  for_range_p_impl<B>(std::make_integer_sequence<int, (E > B ? E - B : 0)>{}, f);
}

// ---------------------------------------------------------------------------
// Patch-staged split-bf16 kernel: stride-1 3x3 convolution (forward and data gradient) whose
// image tiles as 4 rows x 32 columns.  Per 16-channel chunk the (4+2) x (32+2) input patch is
// split into its three bf16 planes ONCE and kept in LDS for all nine taps (the gather-GEMM
// above re-stages and re-splits the A tile for every tap: 9x the loads, VALU and LDS writes);
// a tap is a constant offset into the patch, an MFMA A-fragment is one 32-pixel patch row.
// Weights arrive pre-split (unet_pack_conv3x3_weights_bf16x3) and go global -> LDS as raw bits,
// double-buffered per tap.  LDS: patch 3 x 204 x 48 B + weights 2 x 3 x BN x 48 B (66 KB at
// BN = 128: two workgroups per CU).
// ---------------------------------------------------------------------------
// ACT / STATS / BSTATS: the fused layer pipeline on this kernel, exactly as in
// conv_patch_f32_kernel below (activation on load BEFORE the split, statistics of the output,
// reductions of the next backward stage) - the split mode of the fused pipeline.
template <int BN, int WM, int WN, int TH, bool ACT = false, bool STATS = false, bool BSTATS = false>
__global__ __launch_bounds__(256, 2) void conv_patch_split_kernel(const IgemmParams p) {
  constexpr int LDA = 24;                    // bf16 per LDS row: 16 + 8 pad (48 B)
  constexpr int TW = 32, PW = TW + 2;
  constexpr int PPIX = (TH + 2) * PW;        // patch pixels (204 for 4 rows, 340 for 8)
  constexpr int P_PLANE = PPIX * LDA;
  constexpr int P_SLOTS = PPIX * 4;          // f32x4 slots: 16 channels per pixel
  constexpr int P_PASSES = (P_SLOTS + 255) / 256;
  constexpr int B_SLOTS = BN * 6, B_PASSES = (B_SLOTS + 255) / 256;
  constexpr int B_TILE = BN * LDA;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((TH * 32 / WM) * (BN / WN) == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];
  __bf16* Ps = smem_h;                       // [plane][pixel][LDA]
  __bf16* Bs = smem_h + 3 * P_PLANE;         // [buf][plane][BN][LDA]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow0 = (wave / WAVES_N) * TM, wn0 = (wave % WAVES_N) * WN;

  const int H = p.Hin, W = p.Win;
  const int tiles_n = p.Ncols / BN, tiles_x = W / TW, tiles_y = H / TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const int Ktot = p.C0 + p.C1;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(p.w3), 0, (int)p.w3_bytes, 0x00020000);

  // patch slots (slots past P_SLOTS alias an earlier slot: same bytes to the same place)
  int pp_lin[P_PASSES], pp_lds[P_PASSES];
  unsigned pp_oob[P_PASSES];
#pragma unroll
  for (int i = 0; i < P_PASSES; ++i) {
    const int slot = (tid + 256 * i) % P_SLOTS;
    const int pix = slot >> 2, seg = slot & 3;
    const int prow = pix / PW, pcol = pix - prow * PW;
    const int iy = y0 - 1 + prow, ix = x0 - 1 + pcol;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    pp_lin[i] = ok ? ((n * H + iy) * W + ix) * 4 : 0;   // x channel count = byte offset
    pp_oob[i] = (ok ? 0u : 0x80000000u) | (unsigned)(seg * 16);   // + this slot's 4 channels
    pp_lds[i] = pix * LDA + seg * 4;
  }
  unsigned wslot_off[B_PASSES];
  int wslot_lds[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int slot = (tid + 256 * j) % B_SLOTS;
    const int pl = slot / (2 * BN), rem = slot - pl * 2 * BN;
    // 16 consecutive lanes = 8 rows x 2 halves, row fastest: the 8 lanes of one ds_write_b128
    // group hit 8 different rows (conflict-free at the 48-B row stride) and lanes k, k+8
    // read the two halves of one 32-B global segment
    const int row = (rem >> 4) * 8 + (rem & 7), half = (rem >> 3) & 1;
    wslot_off[j] = (unsigned)(pl * p.w3_plane + (p.n_off + n0 + row) * Ktot + 8 * half) * 2u;
    wslot_lds[j] = pl * B_TILE + row * LDA + 8 * half;
  }

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  f32x4 pr[P_PASSES];
  i32x4 rb[B_PASSES];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int nb = 0; nb < TN; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;

  // ACT: coefficients of this thread's four channels (slot & 3 is the same for every pass)
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  float cs = 1.f;
  auto load_patch = [&](int chunk) {
    const int c = chunk * 16;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? p.C0 : p.C1;
    const unsigned cbytes = (unsigned)(first ? c : c - p.C0) * 4u;
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      // pp_lin = 4 * pixel index, so pp_lin * Cs = byte offset of the pixel's channel 0
      const unsigned off = ((unsigned)(pp_lin[i] * Cs) + cbytes) + pp_oob[i];
      pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    if (ACT) {
      const float* al = first ? p.act0_alpha : p.act1_alpha;
      const float* be = first ? p.act0_beta : p.act1_beta;
      if (al) {   // uniform
        const size_t o = (size_t)n * Cs + (first ? c : c - p.C0) + (tid & 3) * 4;
        ca = *reinterpret_cast<const f32x4*>(al + o);
        cb = *reinterpret_cast<const f32x4*>(be + o);
        cs = p.slope;
      } else {    // plain source: z = v, slope 1 = identity
        ca = f32x4{1.f, 1.f, 1.f, 1.f};
        cb = f32x4{0.f, 0.f, 0.f, 0.f};
        cs = 1.f;
      }
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      if (ACT) pr[i] = act4(pr[i], ca, cb, cs, (pp_oob[i] >> 31) == 0u);
      bf16x4 h, m, l;
      split3(pr[i], h, m, l);
      __bf16* d = Ps + pp_lds[i];
      *reinterpret_cast<bf16x4*>(d) = h;
      *reinterpret_cast<bf16x4*>(d + P_PLANE) = m;
      *reinterpret_cast<bf16x4*>(d + 2 * P_PLANE) = l;
    }
  };
  auto load_b = [&](int t, int chunk) {
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const int wt = (int)(((tw >> ((t & 3) * 8)) & 0xffu) >> 4);
    const unsigned woff = (unsigned)(wt * p.tap_stride + chunk * 16) * 2u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wslot_off[j] + woff, 0, 0);
  };
  auto store_b = [&](int buf) {
    __bf16* Bb = Bs + buf * 3 * B_TILE;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<i32x4*>(Bb + wslot_lds[j]) = rb[j];
  };

  const int chunks = Ktot / 16;
  const int steps = chunks * 9;
  load_patch(0);
  load_b(0, 0);
  store_patch();
  store_b(0);
  __syncthreads();

  const int a_lane = ((wrow0 + 1) * PW + li + 1) * LDA + 8 * lh;
  const int b_lane = (wn0 + li) * LDA + 8 * lh;
  auto tap_off = [&](int t) {   // patch offset of tap t (elements)
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int oy = (int)(e & 3u) - 1, ox = (int)((e >> 2) & 3u) - 1;
    return (oy * PW + ox) * LDA;
  };
  bf16x8 a[3][TM], b[3][TN];
  auto read_a = [&](bf16x8 (&q)[3][TM], int t) {
    const __bf16* Ab = Ps + a_lane + tap_off(t);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int m = 0; m < TM; ++m)
        q[pl][m] = *reinterpret_cast<const bf16x8*>(Ab + pl * P_PLANE + m * PW * LDA);
  };
  int t = 0, chunk = 0;
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    // next step's weights; next chunk's patch rides in registers through the nine taps
    const int t1 = (t == 8) ? 0 : t + 1;
    const int chunk1 = (t == 8) ? chunk + 1 : chunk;
    const bool more = s + 1 < steps;
    load_b(more ? t1 : t, more ? chunk1 : chunk);
    if (t == 0) load_patch(chunk + 1 < chunks ? chunk + 1 : chunk);

    const __bf16* Bb = Bs + buf * 3 * B_TILE + b_lane;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int nb = 0; nb < TN; ++nb)
        b[pl][nb] = *reinterpret_cast<const bf16x8*>(Bb + pl * B_TILE + nb * 32 * LDA);
    read_a(a, t);   // (prefetching the next tap's A fragments under the MFMAs gained nothing)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int nb = 0; nb < TN; ++nb) {
        f32x16 c = acc[m][nb];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][m], b[0][nb], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[1][nb], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[2][nb], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[0][nb], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[1][nb], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[0][nb], c, 0, 0, 0);
        acc[m][nb] = c;
      }
    store_b(buf ^ 1);
    if (t == 8) {            // every wave is done with this chunk's patch
      __syncthreads();
      store_patch();
    }
    __syncthreads();
    t = t1;
    chunk = chunk1;
  }

  // ---- epilogue: D row (= pixel column) (reg&3) + 8*(reg>>2) + 4*lh, D column (= channel) li
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int col = n0 + wn0 + nb * 32 + li;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      float* o = p.out + (((size_t)n * H + (y0 + wrow0 + m)) * W + x0 + 4 * lh) * p.ldo + col;
      if (p.accumulate) {        // uniform: all 16 reads in flight before the first add
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nb][r] += bv + old[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nb][r] += bv;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[m][nb][r];
    }
  }
  constexpr int WAVES_M = 4 / WAVES_N;
  static_assert(WAVES_M * BN * 8 <= 3 * P_PLANE * 2, "reduction scratch fits in the patch area");
  if (BSTATS && p.bs_partial) {   // uniform: reductions of the NEXT backward stage (IgemmParams)
    float2* red = reinterpret_cast<float2*>(Ps);   // the K loop ended on a barrier
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const int col = n0 + wn0 + nb * 32 + li;
      const BwdCoef cf = bwd_coef(p, n, col);
      const float* yb = p.bs_y + (((size_t)n * H + (y0 + wrow0)) * W + x0 + 4 * lh) * p.ldo + col;
      const float2 mine = wave_bwd_stats<TM>(
          cf, p.slope, [&](int m, int r) { return acc[m][nb][r]; },
          [&](int m, int r) { return yb[((size_t)m * W + (r & 3) + 8 * (r >> 2)) * p.ldo]; });
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = mine;
    }
    float2 out;
    if (block_col_sums<BN, WAVES_M>(red, out))
      p.bs_partial[((size_t)n * p.bs_tiles + p.bs_tile0 + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
  if (STATS && p.stats) {   // uniform; the patch area is free scratch
    float2* red = reinterpret_cast<float2*>(Ps);
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const float2 mine = wave_col_stats<TM>([&](int m, int r) { return acc[m][nb][r]; });
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = mine;
    }
    float2 out;
    if (block_col_stats<BN, WAVES_M>(red, 0, 0, false, float2{0.f, 0.f}, 32.f * TM, out))
      p.stats[((size_t)n * p.stats_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
}

// ---------------------------------------------------------------------------
// Patch-staged fp32 kernel: the layout idea of conv_patch_split_kernel on the fp32 matrix
// cores (v_mfma_f32_32x32x2_f32).  Stride-1 3x3 convolution (forward / data gradient) over an
// image that tiles as TH x 32 pixels.  Per 32-channel chunk the (TH+2) x 34 input patch is
// staged ONCE and serves all nine taps (the gather-GEMM re-stages its A tile per tap: 9x the
// global loads and LDS writes); only the [BN][32] weight panel is re-staged per tap, double
// buffered.  One K step = one tap = 16 k-pairs x TM x TN MFMAs per wave between barriers.
// LDS: patch (TH+2)*34 x 144 B + weights 2 x BN x 144 B (66 KB at TH 4 / BN 128: two
// workgroups per CU).
// ACT: the sources hold raw convolution outputs and the previous layer's InstanceNorm +
// LeakyReLU + dropout is applied while the patch goes from registers to LDS (once per element
// per chunk; zero padding stays zero).  STATS: the epilogue emits the tile's per-column
// (mean, M2) for the InstanceNorm that follows this convolution.
// ---------------------------------------------------------------------------
template <int BN, int WM, int WN, int TH, bool ACT = false, bool STATS = false, bool BSTATS = false>
__global__ __launch_bounds__(256, 2) void conv_patch_f32_kernel(const IgemmParams p) {
  constexpr int BK = 32, LDA = BK + 4;
  constexpr int TW = 32, PW = TW + 2;
  constexpr int PPIX = (TH + 2) * PW;
  constexpr int P_SLOTS = PPIX * 8;          // f32x4 slots: 32 channels per pixel
  constexpr int P_PASSES = (P_SLOTS + 255) / 256;
  constexpr int B_SLOTS = BN * 8, B_PASSES = (B_SLOTS + 255) / 256;
  constexpr int B_TILE = BN * LDA;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((TH * 32 / WM) * (BN / WN) == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ps = smem;                          // [pixel][LDA]
  float* Bs = smem + PPIX * LDA;             // [buf][BN][LDA]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow0 = (wave / WAVES_N) * TM, wn0 = (wave % WAVES_N) * WN;

  const int H = p.Hin, W = p.Win;
  const int tiles_n = p.Ncols / BN, tiles_x = W / TW, tiles_y = H / TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const int Ktot = p.C0 + p.C1;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // patch slots (slots past P_SLOTS alias an earlier slot: same bytes to the same place)
  int pp_lin[P_PASSES], pp_lds[P_PASSES];
  unsigned pp_oob[P_PASSES];
#pragma unroll
  for (int i = 0; i < P_PASSES; ++i) {
    const int slot = (tid + 256 * i) % P_SLOTS;
    const int pix = slot >> 3, seg = slot & 7;
    const int prow = pix / PW, pcol = pix - prow * PW;
    const int iy = y0 - 1 + prow, ix = x0 - 1 + pcol;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    pp_lin[i] = ok ? ((n * H + iy) * W + ix) * 4 : 0;   // x channel count = byte offset
    pp_oob[i] = (ok ? 0u : 0x80000000u) | (unsigned)(seg * 16);
    pp_lds[i] = pix * LDA + seg * 4;
  }
  unsigned wslot_off[B_PASSES];
  int wslot_lds[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int slot = (tid + 256 * j) % B_SLOTS;
    const int row = slot >> 3, seg = slot & 7;
    wslot_off[j] = (unsigned)((p.n_off + n0 + row) * Ktot + seg * 4) * 4u;
    wslot_lds[j] = row * LDA + seg * 4;
  }

  f32x4 pr[P_PASSES], rb[B_PASSES];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int nb = 0; nb < TN; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;

  // ACT: coefficients of this thread's four channels (slot & 7 is the same for every pass)
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  float cs = 1.f;
  auto load_patch = [&](int chunk) {
    const int c = chunk * BK;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? p.C0 : p.C1;
    const unsigned cbytes = (unsigned)(first ? c : c - p.C0) * 4u;
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      const unsigned off = ((unsigned)(pp_lin[i] * Cs) + cbytes) + pp_oob[i];
      pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    if (ACT) {
      const float* al = first ? p.act0_alpha : p.act1_alpha;
      const float* be = first ? p.act0_beta : p.act1_beta;
      if (al) {   // uniform
        const size_t o = (size_t)n * Cs + (first ? c : c - p.C0) + (tid & 7) * 4;
        ca = *reinterpret_cast<const f32x4*>(al + o);
        cb = *reinterpret_cast<const f32x4*>(be + o);
        cs = p.slope;
      } else {    // plain source: z = v, slope 1 = identity
        ca = f32x4{1.f, 1.f, 1.f, 1.f};
        cb = f32x4{0.f, 0.f, 0.f, 0.f};
        cs = 1.f;
      }
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      if (ACT) pr[i] = act4(pr[i], ca, cb, cs, (pp_oob[i] >> 31) == 0u);
      *reinterpret_cast<f32x4*>(Ps + pp_lds[i]) = pr[i];
    }
  };
  auto load_b = [&](int t, int chunk) {
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const int wt = (int)(((tw >> ((t & 3) * 8)) & 0xffu) >> 4);
    const unsigned woff = (unsigned)(wt * p.tap_stride + chunk * BK) * 4u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rsw, wslot_off[j] + woff, 0, 0));
  };
  auto store_b = [&](int buf) {
    float* Bb = Bs + buf * B_TILE;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bb + wslot_lds[j]) = rb[j];
  };

  const int chunks = Ktot / BK;
  load_patch(0);
  load_b(0, 0);
  store_patch();
  store_b(0);
  __syncthreads();

  // fragment addresses: lane (li, lh) reads 4 consecutive k of its row at k offset 4*lh;
  // MFMA r of a k-group of 8 multiplies k = 8*kk + 4*lh + r on both operands
  const int a_lane = ((wrow0 + 1) * PW + li + 1) * LDA + 4 * lh;
  const int b_lane = (wn0 + li) * LDA + 4 * lh;
  // Tap 0 of a chunk is PEELED from the rolled loop over taps 1..8: vmcnt is in-order, so with
  // one rolled tap body the wait for a tap's four weight loads had to assume the worst path and
  // also drained the next chunk's patch loads at tap 0, 0.85 us after their issue.  In the
  // peeled copy the compiler counts exactly (tap 0 waits for its weights only; the patch loads
  // land under taps 0..1).
  auto tap_step = [&](auto first_tag, int t, int chunk, int chunk_n) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int buf = (chunk + t) & 1;   // step = 9 * chunk + t
    const int t1 = (t == 8) ? 0 : t + 1;
    load_b(t1, t == 8 ? chunk_n : chunk);
    if constexpr (FIRST) load_patch(chunk_n);
    __builtin_amdgcn_sched_barrier(0);   // the loads stay ahead of the tap's MFMAs

    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int oy = (int)(e & 3u) - 1, ox = (int)((e >> 2) & 3u) - 1;
    const float* Ab = Ps + a_lane + (oy * PW + ox) * LDA;
    const float* Bb = Bs + buf * B_TILE + b_lane;
    f32x4 a[2][TM], b[2][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(Ab + m * PW * LDA);
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) b[0][nb] = *reinterpret_cast<const f32x4*>(Bb + nb * 32 * LDA);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 8) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
          a[nxt][m] = *reinterpret_cast<const f32x4*>(Ab + m * PW * LDA + (kk + 1) * 8);
#pragma unroll
        for (int nb = 0; nb < TN; ++nb)
          b[nxt][nb] = *reinterpret_cast<const f32x4*>(Bb + nb * 32 * LDA + (kk + 1) * 8);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int nb = 0; nb < TN; ++nb)
            acc[m][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][m][r], b[cur][nb][r],
                                                              acc[m][nb], 0, 0, 0);
    }
    store_b(buf ^ 1);
  };
  for (int chunk = 0; chunk < chunks; ++chunk) {
    const int chunk_n = chunk + 1 < chunks ? chunk + 1 : chunk;   // the last chunk re-stages itself
    tap_step(std::true_type{}, 0, chunk, chunk_n);
    __syncthreads();
#pragma nounroll
    for (int t = 1; t < 9; ++t) {
      tap_step(std::false_type{}, t, chunk, chunk_n);
      if (t == 8) {            // every wave is done with this chunk's patch
        __syncthreads();
        store_patch();
      }
      __syncthreads();
    }
  }

#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int col = n0 + wn0 + nb * 32 + li;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      float* o = p.out + (((size_t)n * H + (y0 + wrow0 + m)) * W + x0 + 4 * lh) * p.ldo + col;
      if (p.accumulate) {        // uniform: all 16 reads in flight before the first add
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nb][r] += bv + old[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nb][r] += bv;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[m][nb][r];
    }
  }
  if (BSTATS && p.bs_partial) {   // uniform: reductions of the NEXT backward stage (IgemmParams)
    constexpr int WAVES_M = 4 / WAVES_N;
    float2* red = reinterpret_cast<float2*>(Ps);   // the K loop ended on a barrier
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const int col = n0 + wn0 + nb * 32 + li;
      const BwdCoef cf = bwd_coef(p, n, col);
      const float* yb = p.bs_y + (((size_t)n * H + (y0 + wrow0)) * W + x0 + 4 * lh) * p.ldo + col;
      const float2 mine = wave_bwd_stats<TM>(
          cf, p.slope, [&](int m, int r) { return acc[m][nb][r]; },
          [&](int m, int r) {
            return yb[((size_t)m * W + (r & 3) + 8 * (r >> 2)) * p.ldo];
          });
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = mine;
    }
    float2 out;
    if (block_col_sums<BN, WAVES_M>(red, out))
      p.bs_partial[((size_t)n * p.bs_tiles + p.bs_tile0 + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
  if (STATS && p.stats) {   // uniform
    // the K loop ended on a barrier: the patch area is free scratch
    constexpr int WAVES_M = 4 / WAVES_N;
    float2* red = reinterpret_cast<float2*>(Ps);
    static_assert(WAVES_M * BN * 2 <= PPIX * LDA, "stats scratch fits in the patch area");
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const float2 mine = wave_col_stats<TM>([&](int m, int r) { return acc[m][nb][r]; });  // acc already holds + bias
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = mine;
    }
    float2 out;
    if (block_col_stats<BN, WAVES_M>(red, 0, 0, false, float2{0.f, 0.f}, 32.f * TM, out))
      p.stats[((size_t)n * p.stats_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
}

// ---------------------------------------------------------------------------
// Patch-staged kernel of the mixed-precision pipeline (BASELINE config 4): bf16 layer tensors
// in HBM, v_mfma_f32_32x32x16_bf16, fp32 accumulation; stride-1 3x3 forward (ACT: activation on
// load in fp32 before the rounding to bf16; STATS: statistics of the fp32 accumulators) and data
// gradient.  Geometry of conv_patch_f32_kernel: the (TH+2) x 34 patch of a 32-channel chunk is
// staged once (bf16, rows of 40 elements = 80 B: conflict-free ds_read_b128 fragments) and
// serves all nine taps.  At the bf16 matrix rate a tap is only 2 x TM x TN MFMAs of 32 cycles,
// so a K step covers a whole kernel ROW (three taps, three weight panels staged together): 24
// MFMAs per wave between barriers at 128 columns.  Weights stay fp32 in HBM (master copy) and
// are rounded while staged.  LDS 16 KB patch + 61 KB weights at 128 columns (two per CU).
// ---------------------------------------------------------------------------
// UP (fused forward of a decoder stage's first convolution, round 4): source 0 is the
// LOW-resolution tensor [N][H/2][W/2][C0]; per 32-channel chunk of it the (TH/2 + 2) x 18
// low-resolution pixels under the tile's patch are activated (fp32) and staged into an fp32 LDS
// scratch, and every patch pixel is blended from its 2 x 2 neighbours in PyTorch's operation
// order and rounded to bf16 ONCE - the value the materialised form (upsample2x_fwd_b16x8_kernel
// + this kernel on its output) stages, so the two are bit-identical and the up-sampled tensor
// (268 MB at 512 x 512) is neither written nor read.  As conv_patch_up_kernel (fp32).
#ifdef B16_STAMPS
__device__ unsigned long long g_stamps[256 * 16];   // (timing experiments only: tools/stamps_b16.py)
#endif
// SD = 2 (round 4, fused forward only): the stride-2 first convolution of an encoder stage.  The
// patch is the (2 TH + 1) x 65 INPUT pixels under a TH x 32 tile of output pixels; an LDS patch
// row holds the 33 even input columns first and the 32 odd ones behind them (as
// conv_patch_s2_kernel), so the fragment of tap kx for output columns 0..31 is 32 consecutive
// slots (kx = 0: even 0.., kx = 1: odd 0.., kx = 2: even 1..) and fragment rows of consecutive
// output rows are two patch rows apart.  Everything else - K steps of three taps, panels,
// statistics, the output tile through LDS - is the stride-1 code.
template <int BN, int WM, int WN, int TH, bool ACT = false, bool STATS = false, bool BSTATS = false,
          bool WB = false, bool UP = false, int SD = 1>
// (32-accumulator tiles without the up-sampling loader - and, when fused, with the pre-rounded
// weight panels - stay under 170 registers and 48 KB of LDS: three workgroups per CU - the short K loops of the 32- and 64-channel layers expose one
// HBM latency per chunk, and the third workgroup is what covers it)
__global__ __launch_bounds__(256, ((WM / 32) * (WN / 32) <= 2 && !UP && (WB || !ACT) && SD == 1) ? 3 : 2)
void conv_patch_b16_kernel(const IgemmParams p) {
  static_assert(!UP || (ACT && STATS && !BSTATS), "UP is a fused-forward loader");
  static_assert(SD == 1 || (SD == 2 && ACT && STATS && !BSTATS && !UP), "stride 2: the fused forward");
#ifdef B16_STAMPS
  int stamp_i = 0;
#define STAMP() do { if (threadIdx.x == 0 && blockIdx.x >= B16_STAMPS && blockIdx.x < B16_STAMPS + 256 && stamp_i < 16) \
      g_stamps[(blockIdx.x - B16_STAMPS) * 16 + stamp_i++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP() do { } while (0)
#endif
  STAMP();
  constexpr int BK = 32, LDA = BK + 8;       // bf16 elements per LDS row
  constexpr int TW = 32, PW = SD == 2 ? 2 * TW + 1 : TW + 2;
  constexpr int ODD0 = TW + 1;               // SD = 2: first odd-column slot of a patch row
  constexpr int PPIX = (SD == 2 ? 2 * TH + 1 : TH + 2) * PW;
  constexpr int LW = TW / 2 + 2, LPIX = (TH / 2 + 2) * LW;   // UP: low-resolution scratch pixels
  constexpr int L_PASSES = (LPIX * 8 + 255) / 256, LLD = 36;  // fp32 scratch rows of 36 floats
  constexpr int P_SLOTS = PPIX * 8;          // 4-channel slots: 32 channels per pixel
  constexpr int P_PASSES = (P_SLOTS + 255) / 256;
  // weight slots of a K step (three taps x BN rows x 32 channels): 4 fp32 channels per slot, or
  // (WB: the weights pre-rounded to bf16, p.w3) 8 bf16 channels - half the slots, no conversion
  constexpr int B_SEGS = WB ? 4 : 8;
  constexpr int B_SLOTS = 3 * BN * B_SEGS, B_PASSES = (B_SLOTS + 255) / 256;
  constexpr int B_TILE = 3 * BN * LDA;       // three taps of one kernel row
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((TH * 32 / WM) * (BN / WN) == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];
  __bf16* Ps = smem_h;                       // [pixel][LDA]
  __bf16* Bs = smem_h + PPIX * LDA;          // [buf][tap in row][BN][LDA]
  float* Ls = reinterpret_cast<float*>(Bs + 2 * B_TILE);   // UP: [low pixel][LLD] activated fp32

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow0 = (wave / WAVES_N) * TM, wn0 = (wave % WAVES_N) * WN;

  const int H = p.Hin, W = p.Win;            // input image (the bounds of the patch)
  const int Ho = SD == 2 ? p.Hl : H, Wo = SD == 2 ? p.Wl : W;   // output image (the tiles)
  const int tiles_n = p.Ncols / BN, tiles_x = Wo / TW, tiles_y = Ho / TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const int Ktot = p.C0 + p.C1;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = WB ? __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(p.w3), 0, (int)p.w3_bytes, 0x00020000)
                                        : __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // Per patch slot ONE register: the pixel index, bit 31 set for a zero-padding slot.  The LDS
  // address is affine in the pass (32 pixels a pass); in the last pass the threads past the patch
  // repeat their pass-0 slot (the same bytes to the same place) instead of wrapping around, which
  // keeps it so.  (Three arrays of P_PASSES registers here were what held the fused forms of the
  // 32-accumulator tiles at two workgroups per CU.)
  constexpr int LAST = P_PASSES - 1;
  const bool last_ok = tid + 256 * LAST < P_SLOTS;
  auto slot_of = [&](int i) __attribute__((always_inline)) {
    return (i == LAST && !last_ok) ? tid : tid + 256 * i;
  };
  int pp_lin[P_PASSES];
#pragma unroll
  for (int i = 0; i < P_PASSES; ++i) {
    const int pix = slot_of(i) >> 3;
    const int prow = pix / PW, pidx = pix - prow * PW;
    const int pcol = SD == 2 ? (pidx < ODD0 ? 2 * pidx : 2 * (pidx - ODD0) + 1) : pidx;
    const int iy = SD * y0 - 1 + prow, ix = SD * x0 - 1 + pcol;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    pp_lin[i] = ok ? (n * H + iy) * W + ix : (int)0x80000000;
  }
  auto pp_lds0_of = [&](int t) __attribute__((always_inline)) { return (t >> 3) * LDA + (t & 7) * 4; };
  int pp_lds0 = pp_lds0_of(tid);   // (store_patch derives it again from an opaque thread index)
  auto pp_lds = [&](int i) __attribute__((always_inline)) {
    return (i == LAST && !last_ok) ? pp_lds0 : pp_lds0 + i * 32 * LDA;
  };
  auto pp_oob = [&](int i) __attribute__((always_inline)) {
    return (unsigned)pp_lin[i] & 0x80000000u;
  };
  const int seg4 = (tid & 7) * 4;            // this thread's four channels within a chunk
  // UP: low-resolution scratch slots (clamped pixel index of the source; a pass = 32 pixels) and,
  // per patch slot, its top-left scratch pixel (bits 4..) + the parities that pick the 0.75 / 0.25
  // weights (bits 0, 1)
  int lp_lin[UP ? L_PASSES : 1];
  if constexpr (UP) {
    const int h = H >> 1, w = W >> 1;
#pragma unroll
    for (int j = 0; j < L_PASSES; ++j) {
      const int lpix = (tid >> 3) + 32 * j;
      const int lp = lpix < LPIX ? lpix : LPIX - 1;
      const int lr = lp / LW, lc = lp - lr * LW;
      int gy = (y0 >> 1) - 1 + lr, gx = (x0 >> 1) - 1 + lc;
      gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
      gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
      lp_lin[j] = (n * h + gy) * w + gx;
    }
  }
  // weight slot of pass j = tid + 256 j: (tap in row, column row, segment).  BN * B_SEGS (slots
  // of one tap) and 256 are powers of two, so the slot of pass j is the slot of pass 0 moved by
  // compile-time amounts - three registers instead of three per pass (slots past the panel, in
  // the last pass, load tap 2 again and are not stored):
  constexpr int PER = BN * B_SEGS;
  static_assert((PER & (PER - 1)) == 0 && (PER >= 256 || 256 % PER == 0), "power-of-two tap panels");
  // (derived again from an opaque copy of the thread index where they are used: registers that
  // would otherwise be held across the MFMAs)
  auto tid_again = [&]() __attribute__((always_inline)) { int t = tid; asm volatile("" : "+v"(t)); return t; };
  auto ws_tap0_of = [&](int t) __attribute__((always_inline)) { return PER >= 256 ? 0 : t / PER; };   // tap of pass j: + (256 j) / PER
  auto wslot_off0_of = [&](int t) __attribute__((always_inline)) {
    const int rem0 = t % PER;
    return (unsigned)((p.n_off + n0 + rem0 / B_SEGS) * Ktot + (rem0 % B_SEGS) * (32 / B_SEGS)) * (WB ? 2u : 4u);
  };
  auto wslot_lds0_of = [&](int t) __attribute__((always_inline)) {
    const int rem0 = t % PER;
    return (ws_tap0_of(t) * BN + rem0 / B_SEGS) * LDA + (rem0 % B_SEGS) * (32 / B_SEGS);
  };
  auto ws_tapj = [](int j) { return (256 * j) / PER; };           // pass j: tap + this, column row + ws_rowj
  auto ws_rowj = [](int j) { return ((256 * j) % PER) / B_SEGS; };

  // WB: two register sets - the panel of step s + 2 is loaded while that of s + 1 waits for its
  // LDS stage (a full step of flight; with one set the loads of s + 1 were stored at the end of
  // the step that issued them)
  constexpr int B_SETS = WB ? 2 : 1;
  i32x2r pr[P_PASSES];      // the patch slots in flight, raw (8 bytes = 4 bf16: two registers)
  f32x4 rb[B_SETS][B_PASSES];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int nb = 0; nb < TN; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;

  auto tap_of = [&](int t, int& oy, int& ox, int& wt) __attribute__((always_inline)) {
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    oy = (int)(e & 3u) - 1; ox = (int)((e >> 2) & 3u) - 1; wt = (int)(e >> 4);
  };
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  float cs = 1.f;
  static_assert(L_PASSES <= P_PASSES, "the low-resolution slots travel in pr[]");
  bool up_chunk = false;           // UP: the chunk in the registers comes from source 0 (uniform)
  // `dead`: the loads of a chunk that does not exist (issued all the same, out of range = no memory
  // access, so that every path through a K step issues the SAME NUMBER of loads: hipcc then counts
  // vmcnt exactly; with a conditional load anywhere in the loop it falls back to vmcnt(0) at every
  // LDS store of a weight panel, which exposed a full load latency per step - round 4 finding)
  // Byte offsets of the patch slots in the CURRENT source (pixel * channels + this thread's four
  // channels, bit 31 = padding), rebuilt when the source changes - once per tile at most; a
  // chunk's loads then cost no VALU instruction at all: the chunk offset rides in the SCALAR
  // offset of the buffer load (raw buffers: not part of the range check, so padding slots stay
  // out of range).  (Phase stamps: a K step that issued 11 loads cost 0.6-1.0 us more than one
  // that did not - six address instructions per load.)
  unsigned pp_off[P_PASSES];
  int pp_src = -1;
  auto load_patch = [&](int chunk, bool dead = false) __attribute__((always_inline)) {
    const unsigned kill = dead ? 0x80000000u : 0u;
    const int c = chunk * BK;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? p.C0 : p.C1;
    const int cc = (first ? c : c - p.C0) + seg4;
    if (pp_src != (first ? 0 : 1)) {   // uniform; no load inside
      pp_src = first ? 0 : 1;
#pragma unroll
      for (int i = 0; i < P_PASSES; ++i)
        pp_off[i] = ((unsigned)((pp_lin[i] & 0x7fffffff) * Cs + seg4) * 2u) | pp_oob(i);
    }
    const int soff = (first ? c : c - p.C0) * 2;       // bytes, scalar
    if constexpr (UP) {
      up_chunk = first;
      if (first) {   // uniform: the low-resolution pixels under the patch (clamped, all valid)
#pragma unroll
        for (int j = 0; j < L_PASSES; ++j)
          pr[j] = buf_ld4_raw16(rs0, (unsigned)(lp_lin[j] * Cs + cc), kill);
#pragma unroll
        for (int j = L_PASSES; j < P_PASSES; ++j)   // (as many loads as the other branch: see `dead`)
          pr[j] = buf_ld4_raw16(rs0, 0u, 0x80000000u);
      } else {
#pragma unroll
        for (int i = 0; i < P_PASSES; ++i)
          pr[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, pp_off[i] | kill, soff, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < P_PASSES; ++i)
        pr[i] = __builtin_amdgcn_raw_buffer_load_b64(rs, pp_off[i] | kill, soff, 0);
    }
    if (ACT) {
      const float* al = first ? p.act0_alpha : p.act1_alpha;
      const float* be = first ? p.act0_beta : p.act1_beta;
      // (always two loads - a plain source reads the head of the weights instead - see `dead`)
      const size_t o = (size_t)n * Cs + cc;
      ca = *reinterpret_cast<const f32x4*>(al ? al + o : p.w);
      cb = *reinterpret_cast<const f32x4*>(al ? be + o : p.w);
      cs = p.slope;
      if (!al) {  // uniform; plain source: z = v, slope 1 = identity
        ca = f32x4{1.f, 1.f, 1.f, 1.f};
        cb = f32x4{0.f, 0.f, 0.f, 0.f};
        cs = 1.f;
      }
    }
  };
  auto to_bf16 = [](const f32x4 v) {
    bf16x4 h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    return h;
  };
  // UP: activate the low-resolution slots (fp32) into the scratch; a barrier later every patch
  // slot blends its 2 x 2 neighbours (PyTorch's order, as upsample2x_fwd_b16x8_kernel) and the
  // result is rounded to bf16 once; zero padding after the blend
  auto store_low = [&]() __attribute__((always_inline)) {
    if constexpr (UP) {
#pragma unroll
      for (int j = 0; j < L_PASSES; ++j) {
        const int lpix = (tid >> 3) + 32 * j;
        if (lpix < LPIX)
          *reinterpret_cast<f32x4*>(Ls + lpix * LLD + seg4) = act4(widen16(pr[j]), ca, cb, cs, true);
      }
    }
  };
  // The blend, by 2 x 2 BLOCKS of patch pixels: patch rows 2b, 2b + 1 (an odd and the next even
  // image row) take the same two low-resolution rows, with weights (0.75, 0.25) and (0.25, 0.75) -
  // image row 0: (0, 1), PyTorch clamps the source coordinate - and columns alike, so a block
  // shares its four scratch pixels: 4 reads, 4 row blends and 4 column blends for four outputs
  // (one pixel at a time it was 16 reads and 12 blends), the same fused multiply-adds in the same
  // order as blend2x2 / upsample2x_fwd_b16x8_kernel: bit-identical.  Zero padding after the blend.
  auto blend_patch = [&]() __attribute__((always_inline)) {
    if constexpr (UP) {
      constexpr int BW = PW / 2, BH = (TH + 2) / 2;
      constexpr int U_PASSES = (BH * BW * 8 + 255) / 256;
      int tb = tid;               // (opaque: nothing of this is carried across the K loop)
      asm volatile("" : "+v"(tb));
      const int sg4 = (tb & 7) * 4;
#pragma unroll
      for (int i = 0; i < U_PASSES; ++i) {
        const int bq = (tb >> 3) + 32 * i;
        if (bq < BH * BW) {
          const int brow = bq / BW, bcol = bq - brow * BW;
          const float* L = Ls + (brow * LW + bcol) * LLD + sg4;
          const f32x4 p00 = *reinterpret_cast<const f32x4*>(L);
          const f32x4 p01 = *reinterpret_cast<const f32x4*>(L + LLD);
          const f32x4 p10 = *reinterpret_cast<const f32x4*>(L + LW * LLD);
          const f32x4 p11 = *reinterpret_cast<const f32x4*>(L + LW * LLD + LLD);
          const int iy0 = y0 - 1 + 2 * brow, ix0 = x0 - 1 + 2 * bcol;   // image pixel of (dy, dx) = (0, 0)
          const float wxb1 = ix0 + 1 == 0 ? 1.f : 0.75f, wxb0 = 1.f - wxb1;   // dx = 1 (even image column)
          const float wyb1 = iy0 + 1 == 0 ? 1.f : 0.75f, wyb0 = 1.f - wyb1;   // dy = 1 (even image row)
          f32x4 t[2][2];          // [low row][dx]
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            t[0][0][e] = __builtin_fmaf(p01[e], 0.25f, p00[e] * 0.75f);
            t[1][0][e] = __builtin_fmaf(p11[e], 0.25f, p10[e] * 0.75f);
            t[0][1][e] = __builtin_fmaf(p01[e], wxb1, p00[e] * wxb0);
            t[1][1][e] = __builtin_fmaf(p11[e], wxb1, p10[e] * wxb0);
          }
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
              const float wy1 = dy ? wyb1 : 0.25f, wy0 = dy ? wyb0 : 0.75f;
              const bool ok = (unsigned)(iy0 + dy) < (unsigned)H && (unsigned)(ix0 + dx) < (unsigned)W;
              const float okf = ok ? 1.f : 0.f;
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(t[1][dx][e], wy1, t[0][dx][e] * wy0) * okf;
              *reinterpret_cast<bf16x4*>(Ps + ((2 * brow + dy) * PW + 2 * bcol + dx) * LDA + sg4) = to_bf16(v);
            }
        }
        asm volatile("" ::: "memory");   // one pass (four ds_read_b128) at a time
      }
    }
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
    { int t = tid; asm volatile("" : "+v"(t)); pp_lds0 = pp_lds0_of(t); }
    if constexpr (UP) {
      if (up_chunk) { blend_patch(); return; }   // uniform
    }
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      if (ACT) {
        // (opaque: the padding factor of every slot is loop-invariant and would be hoisted into
        // P_PASSES registers - pairs, for the packed multiplies - held across the whole K loop)
        unsigned oob = pp_oob(i);
        asm volatile("" : "+v"(oob));
        const f32x4 v = act4(widen16(pr[i]), ca, cb, cs, oob == 0u);
        *reinterpret_cast<bf16x4*>(Ps + pp_lds(i)) = to_bf16(v);
        asm volatile("" ::: "memory");   // one slot's temporaries at a time
      } else {   // a plain bf16 operand goes to LDS as it came (bf16 -> fp32 -> bf16 is the identity)
        *reinterpret_cast<i32x2r*>(Ps + pp_lds(i)) = pr[i];
      }
    }
  };
  auto load_b = [&](int row, int chunk, auto setc) __attribute__((always_inline)) {   // the three taps 3*row .. 3*row+2
    constexpr int SET = decltype(setc)::value;
    int oy, ox, wt0, wt1, wt2;
    tap_of(3 * row, oy, ox, wt0);
    tap_of(3 * row + 1, oy, ox, wt1);
    tap_of(3 * row + 2, oy, ox, wt2);
    const int t_ = tid_again();
    const int ws_tap0 = ws_tap0_of(t_);
    const unsigned wslot_off0 = wslot_off0_of(t_);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const int tap = ws_tap0 + ws_tapj(j);
      const int wt = tap == 0 ? wt0 : (tap == 1 ? wt1 : wt2);
      const unsigned woff = (unsigned)(wt * p.tap_stride + ws_rowj(j) * Ktot + chunk * BK) * (WB ? 2u : 4u);
      rb[SET][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 rsw, wslot_off0 + woff, 0, 0));
    }
  };
  auto store_b = [&](int buf, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    __bf16* Bb = Bs + buf * B_TILE;
    const int wslot_lds0 = wslot_lds0_of(tid_again());
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      if (256 * (j + 1) <= B_SLOTS || tid + 256 * j < B_SLOTS) {
        const int lds = wslot_lds0 + (ws_tapj(j) * BN + ws_rowj(j)) * LDA;
        if constexpr (WB) *reinterpret_cast<f32x4*>(Bb + lds) = rb[SET][j];   // 8 bf16
        else *reinterpret_cast<bf16x4*>(Bb + lds) = to_bf16(rb[SET][j]);
      }
  };

  const int chunks = Ktot / BK;
  const int steps = chunks * 3;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, B_SETS - 1>;
  // the bias of this lane's columns, fetched here (phase stamps: read at the top of the epilogue,
  // its latency was ~0.8 of the 13 us of a 64-channel tile); one load whether there is a bias or not
  float bias_v[TN];
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const float bv = (p.bias ? p.bias : p.w)[n0 + wn0 + nb * 32 + li];
    bias_v[nb] = p.bias ? bv : 0.f;
  }
  load_patch(0);
  load_b(0, 0, S0{});
  if constexpr (B_SETS == 2) load_b(steps > 1 ? 1 : 0, 0, S1{});   // (chunks >= 1: step 1 = row 1 of chunk 0)
  if constexpr (UP) {      // chunk 0 is a low-resolution one (C0 >= 32): scratch, barrier, blend
    store_low();
    __syncthreads();
  }
  STAMP();          // loads issued
  store_patch();
  store_b(0, S0{});
  __syncthreads();
  STAMP();          // first stage in LDS

  // lane (li, lh) reads the 8 consecutive k = 16*kk + 8*lh .. +7 of its row
  // (SD = 2: output pixel (wrow0 + m, li), tap (ky, kx) -> patch row 2 (wrow0 + m) + ky, column
  // slot li + {0, ODD0, 1}[kx])
  const int a_lane = SD == 2 ? (2 * wrow0 * PW + li) * LDA + 8 * lh
                             : ((wrow0 + 1) * PW + li + 1) * LDA + 8 * lh;
  auto tap_lds = [&](int oy, int ox) __attribute__((always_inline)) {   // (oy, ox) = (ky - 1, kx - 1)
    if constexpr (SD == 2) return ((oy + 1) * PW + (ox == 0 ? ODD0 : (ox + 1) >> 1)) * LDA;
    else return (oy * PW + ox) * LDA;
  };
  const int b_lane = (wn0 + li) * LDA + 8 * lh;
  // One K step (three taps of kernel row ROW) with EVERYTHING that issues a load fixed at compile
  // time: the row, the register set that receives this step's panel load (PH; the other set holds
  // the panel of the next step), an unconditional patch load at row 0 (`dead` in the last chunk).
  // Two sets alternate with period two and the rows with period three, so the loop body is two
  // chunks (six steps), with a three-step tail for an odd chunk count.
  auto step = [&](auto rowc, auto phc, int chunk) __attribute__((always_inline)) {
    constexpr int ROW = decltype(rowc)::value, PH = decltype(phc)::value;
    using LOADSET = std::integral_constant<int, B_SETS == 2 ? PH : 0>;
    using STORESET = std::integral_constant<int, B_SETS == 2 ? 1 - PH : 0>;
    constexpr int DIST = B_SETS == 2 ? 2 : 1;          // panel of step s + DIST
    const int buf = (chunk + ROW) & 1;                 // s = 3 chunk + ROW
    {
      constexpr int rown = (ROW + DIST) % 3;
      const int chunkn = chunk + (ROW + DIST) / 3;
      load_b(rown, chunkn < chunks ? chunkn : chunks - 1, LOADSET{});   // (past the end: any panel)
    }
    if constexpr (ROW == 0) load_patch(chunk + 1 < chunks ? chunk + 1 : chunk, chunk + 1 >= chunks);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int oy, ox, wt;
      tap_of(3 * ROW + j, oy, ox, wt);
      const __bf16* Ab = Ps + a_lane + tap_lds(oy, ox);
      const __bf16* Bb = Bs + buf * B_TILE + j * BN * LDA + b_lane;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
          a[m] = *reinterpret_cast<const bf16x8*>(Ab + m * SD * PW * LDA + kk * 16);
#pragma unroll
        for (int nb = 0; nb < TN; ++nb)
          b[nb] = *reinterpret_cast<const bf16x8*>(Bb + nb * 32 * LDA + kk * 16);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int nb = 0; nb < TN; ++nb)
            acc[m][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[nb], acc[m][nb], 0, 0, 0);
      }
    }
    store_b(buf ^ 1, STORESET{});
    if constexpr (UP && ROW == 1) {   // the next chunk's low-resolution pixels: a barrier ahead of the blend
      if (up_chunk && chunk + 1 < chunks) store_low();   // uniform
    }
    if constexpr (ROW == 2) {
      if (chunk + 1 < chunks) {   // uniform; every wave is done with this chunk's patch
        __syncthreads();
        store_patch();
      }
    }
    __syncthreads();
    STAMP();
  };
  using R0 = std::integral_constant<int, 0>;
  using R1 = std::integral_constant<int, 1>;
  using R2 = std::integral_constant<int, 2>;
  if constexpr (B_SETS == 2) {
    int c = 0;
    for (; c + 2 <= chunks; c += 2) {
      step(R0{}, R0{}, c); step(R1{}, R1{}, c); step(R2{}, R0{}, c);
      step(R0{}, R1{}, c + 1); step(R1{}, R0{}, c + 1); step(R2{}, R1{}, c + 1);
    }
    if (c < chunks) { step(R0{}, R0{}, c); step(R1{}, R1{}, c); step(R2{}, R0{}, c); }
  } else {
    for (int c = 0; c < chunks; ++c) { step(R0{}, R0{}, c); step(R1{}, R0{}, c); step(R2{}, R0{}, c); }
  }

  __bf16* outp = reinterpret_cast<__bf16*>(p.out);
  // Output tile through LDS (round 4): a lane of the 32 x 32 accumulator block holds ONE channel
  // of 16 pixels, so the direct form is 16 two-byte stores per block (64-byte runs, the texture
  // path's worst case).  The tile is rounded into [pixel][BN + 8] bf16 over the dead patch /
  // weight stages and leaves as 16 bytes per lane, BN * 2 bytes contiguous per pixel.
  constexpr int OLD = BN + 8;
  static_assert(TH * 32 * OLD <= PPIX * LDA + 2 * B_TILE, "the output tile fits in the dead stages");
  // (UP: the launcher checks it - the scalar form next to the loader's state spills)
  const bool wide = UP || (!p.accumulate && (p.ldo & 7) == 0 &&
                           (reinterpret_cast<uintptr_t>(outp) & 15) == 0);   // uniform
  if (wide) {
    __bf16* Os = smem_h;   // (the barrier that ends the last K step has passed)
    // (the lane's tile coordinates are derived again from an opaque copy of the thread index:
    // kept across the K loop they are what tips the 64-column loader form into scratch)
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    const int o_lane = (((te >> 6) / WAVES_N) * TM * 32 + 4 * ((te >> 5) & 1)) * OLD +
                       ((te >> 6) % WAVES_N) * WN + (te & 31);
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const float bv = bias_v[nb];
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[m][nb][r] += bv;
          Os[o_lane + (m * 32 + (r & 3) + 8 * (r >> 2)) * OLD + nb * 32] = (__bf16)acc[m][nb][r];
        }
    }
    STAMP();               // tile rounded into LDS
    __syncthreads();
    STAMP();
    constexpr int SEGS = BN / 8, O_SLOTS = TH * 32 * SEGS;
    static_assert(O_SLOTS % 256 == 0, "whole passes");
#pragma unroll
    for (int i = 0; i < O_SLOTS / 256; ++i) {
      const int slot = te + 256 * i;
      const int pix = slot / SEGS, seg = slot - pix * SEGS;
      const f32x4 v = *reinterpret_cast<const f32x4*>(Os + pix * OLD + seg * 8);
      *reinterpret_cast<f32x4*>(outp + (((size_t)n * Ho + (y0 + (pix >> 5))) * Wo + x0 + (pix & 31)) * p.ldo +
                                n0 + seg * 8) = v;
    }
    STAMP();               // stores issued
    __syncthreads();       // the statistics epilogues reuse the stages
    STAMP();               // output stored
  } else if constexpr (!UP)
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int col = n0 + wn0 + nb * 32 + li;
    const float bv = bias_v[nb];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      __bf16* o = outp + (((size_t)n * Ho + (y0 + wrow0 + m)) * Wo + x0 + 4 * lh) * p.ldo + col;
      if (p.accumulate) {        // uniform: all 16 reads in flight before the first add
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = (float)o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nb][r] += bv + old[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nb][r] += bv;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = (__bf16)acc[m][nb][r];
    }
  }
  if (BSTATS && p.bs_partial) {   // uniform: reductions of the NEXT backward stage, from the fp32
    constexpr int WAVES_M = 4 / WAVES_N;   // accumulators and the bf16 raw outputs of that layer
    float2* red = reinterpret_cast<float2*>(Ps);
    const __bf16* ybase = reinterpret_cast<const __bf16*>(p.bs_y);
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const int col = n0 + wn0 + nb * 32 + li;
      const BwdCoef cf = bwd_coef(p, n, col);
      const __bf16* yb = ybase + (((size_t)n * H + (y0 + wrow0)) * W + x0 + 4 * lh) * p.ldo + col;
      const float2 mine = wave_bwd_stats<TM>(
          cf, p.slope, [&](int m, int r) { return acc[m][nb][r]; },
          [&](int m, int r) { return (float)yb[((size_t)m * W + (r & 3) + 8 * (r >> 2)) * p.ldo]; });
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = mine;
    }
    float2 out;
    if (block_col_sums<BN, WAVES_M>(red, out))
      p.bs_partial[((size_t)n * p.bs_tiles + p.bs_tile0 + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
  if (STATS && p.stats) {   // uniform; statistics of the fp32 accumulators (before the rounding)
    constexpr int WAVES_M = 4 / WAVES_N;
    float2* red = reinterpret_cast<float2*>(Ps);
    static_assert(WAVES_M * BN * 8 <= PPIX * LDA * 2, "stats scratch fits in the patch area");
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const float2 mine = wave_col_stats<TM>([&](int m, int r) { return acc[m][nb][r]; });
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = mine;
    }
    float2 out;
    if (block_col_stats<BN, WAVES_M>(red, 0, 0, false, float2{0.f, 0.f}, 32.f * TM, out))
      p.stats[((size_t)n * p.stats_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
  STAMP();
#undef STAMP
}

// ---------------------------------------------------------------------------
// Stride-2 3x3 forward convolution (the first convolution of encoder stages 1..4,
// Our_UNet/models/unet.py:106-115 with stride 2), patch-staged: the (2 TH + 1) x 65 input
// pixels under a TH x 32 tile of OUTPUT pixels are staged once per 16-channel chunk and serve
// all nine taps (the gather-GEMM re-loads and re-activates its A tile per tap).  An LDS patch
// row holds the 33 even input columns first and the 32 odd ones behind them, so the fragment
// of tap kx for output columns 0..31 is 32 CONSECUTIVE slots (kx = 0: even 0.., kx = 1: odd
// 0.., kx = 2: even 1..) and reads as conflict-free as the stride-1 patch.  16-channel chunks
// keep patch + weight panels at 67 KB (two workgroups per CU at BN = 128).  Always the fused
// layer form: activation on load (plain sources pass null coefficients), statistics epilogue.
// ---------------------------------------------------------------------------
template <int BN, int WM, int WN, int TH>
__global__ __launch_bounds__(256, 2) void conv_patch_s2_kernel(const IgemmParams p) {
  constexpr int BK = 16, LDA = BK + 4, SEG = BK / 4;
  constexpr int TW = 32, PWC = 2 * TW + 1, PH = 2 * TH + 1;
  constexpr int ODD0 = TW + 1;               // first odd-column slot of a patch row
  constexpr int PPIX = PH * PWC;
  constexpr int P_SLOTS = PPIX * SEG;        // f32x4 slots: 16 channels per pixel
  constexpr int P_PASSES = (P_SLOTS + 255) / 256;
  constexpr int B_SLOTS = BN * SEG, B_PASSES = (B_SLOTS + 255) / 256;
  constexpr int B_TILE = BN * LDA;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((TH * 32 / WM) * (BN / WN) == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ps = smem;                          // [patch row][even cols | odd cols][LDA]
  float* Bs = smem + PPIX * LDA;             // [buf][BN][LDA]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow0 = (wave / WAVES_N) * TM, wn0 = (wave % WAVES_N) * WN;

  const int H = p.Hin, W = p.Win, Ho = p.Hl, Wo = p.Wl;
  const int tiles_n = p.Ncols / BN, tiles_x = Wo / TW, tiles_y = Ho / TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;   // output coordinates
  const int Ktot = p.C0 + p.C1;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // patch slots (slots past P_SLOTS alias an earlier slot: same bytes to the same place)
  int pp_lin[P_PASSES], pp_lds[P_PASSES];
  unsigned pp_oob[P_PASSES];
#pragma unroll
  for (int i = 0; i < P_PASSES; ++i) {
    const int slot = (tid + 256 * i) % P_SLOTS;
    // slots walk the LDS order, except that the two pixels of an 8-lane group (the unit a
    // ds_write_b128 is served in, over 32 banks) lie FOUR rows of 80 bytes apart: rows p and
    // p + 1 overlap in four banks (the 0.31 conflict cycles per active LDS cycle of the round-2/3
    // PMC tables), rows p and p + 4 are 320 B = 16 banks apart and do not.  The 64-B global
    // segments of neighbouring lanes then lie a few pixels apart, which costs nothing (a
    // segment is the unit)
    const int seg = slot % SEG;
    int pix = slot / SEG;
    if (pix < PPIX / 8 * 8) pix = (pix & ~7) + ((pix & 7) >> 1) + 4 * (pix & 1);
    const int prow = pix / PWC, idx = pix - prow * PWC;
    const int pcol = idx < ODD0 ? 2 * idx : 2 * (idx - ODD0) + 1;
    const int iy = 2 * y0 - 1 + prow, ix = 2 * x0 - 1 + pcol;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    pp_lin[i] = ok ? ((n * H + iy) * W + ix) * 4 : 0;   // x channel count = byte offset
    pp_oob[i] = (ok ? 0u : 0x80000000u) | (unsigned)(seg * 16);
    pp_lds[i] = pix * LDA + seg * 4;
  }
  unsigned wslot_off[B_PASSES];
  int wslot_lds[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int slot = (tid + 256 * j) % B_SLOTS;
    const int seg = slot % SEG;
    int row = slot / SEG;
    row = (row & ~7) + ((row & 7) >> 1) + 4 * (row & 1);   // (as the patch slots: BN % 8 == 0)
    wslot_off[j] = (unsigned)((p.n_off + n0 + row) * Ktot + seg * 4) * 4u;
    wslot_lds[j] = row * LDA + seg * 4;
  }

  f32x4 pr[P_PASSES], rb[B_PASSES];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int nb = 0; nb < TN; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;

  // coefficients of this thread's four channels (slot % SEG is the same for every pass)
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  float cs = 1.f;
  // `dead`: the loads of a chunk that does not exist are issued all the same, out of range (no
  // memory access), so that every path through the tap steps issues the same number of loads and
  // hipcc counts vmcnt exactly (round 4: with the taps rolled around a conditional patch load it
  // waited vmcnt(0) in the middle of EVERY tap, draining the next chunk's patch 1 us after its
  // issue - what the peeled tap 0 of conv_patch_f32_kernel had already cured there)
  auto load_patch = [&](int chunk, bool dead = false) __attribute__((always_inline)) {
    const unsigned kill = dead ? 0x80000000u : 0u;
    const int c = chunk * BK;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? p.C0 : p.C1;
    const unsigned cbytes = (unsigned)(first ? c : c - p.C0) * 4u;
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      const unsigned off = (((unsigned)(pp_lin[i] * Cs) + cbytes) + pp_oob[i]) | kill;
      pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    const float* al = first ? p.act0_alpha : p.act1_alpha;
    const float* be = first ? p.act0_beta : p.act1_beta;
    {   // (always two loads - a plain source reads the head of the weights instead)
      const size_t o = (size_t)n * Cs + (first ? c : c - p.C0) + (tid % SEG) * 4;
      ca = *reinterpret_cast<const f32x4*>(al ? al + o : p.w);
      cb = *reinterpret_cast<const f32x4*>(al ? be + o : p.w);
      cs = p.slope;
      if (!al) {  // uniform; plain source: z = v, slope 1 = identity
        ca = f32x4{1.f, 1.f, 1.f, 1.f};
        cb = f32x4{0.f, 0.f, 0.f, 0.f};
        cs = 1.f;
      }
    }
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      pr[i] = act4(pr[i], ca, cb, cs, (pp_oob[i] >> 31) == 0u);
      *reinterpret_cast<f32x4*>(Ps + pp_lds[i]) = pr[i];
    }
  };
  auto load_b = [&](int t, int chunk) __attribute__((always_inline)) {
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const int wt = (int)(((tw >> ((t & 3) * 8)) & 0xffu) >> 4);
    const unsigned woff = (unsigned)(wt * p.tap_stride + chunk * BK) * 4u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rsw, wslot_off[j] + woff, 0, 0));
  };
  auto store_b = [&](int buf) __attribute__((always_inline)) {
    float* Bb = Bs + buf * B_TILE;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bb + wslot_lds[j]) = rb[j];
  };

  const int chunks = Ktot / BK;
  load_patch(0);
  load_b(0, 0);
  store_patch();
  store_b(0);
  __syncthreads();

  // output pixel (wrow0 + m, li), tap (ky, kx) -> patch row 2 (wrow0 + m) + ky, column slot
  // li + {0, ODD0, 1}[kx]; lane (li, lh) reads 4 consecutive k at k offset 4*lh
  const int a_lane = (2 * wrow0 * PWC + li) * LDA + 4 * lh;
  const int b_lane = (wn0 + li) * LDA + 4 * lh;
  // Tap 0 of a chunk (the step that also issues the next chunk's patch loads) is PEELED from
  // the rolled loop over taps 1..8, as in conv_patch_f32_kernel: every step then issues a fixed
  // number of loads and the waits are exact (the panel of the next tap only).
  auto tap_step = [&](auto first_tag, int t, int chunk, int chunk_n, bool dead) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int buf = (chunk + t) & 1;   // step = 9 * chunk + t
    const int t1 = (t == 8) ? 0 : t + 1;
    load_b(t1, t == 8 ? chunk_n : chunk);
    if constexpr (FIRST) load_patch(chunk_n, dead);

    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int ky = (int)(e & 3u), kx = (int)((e >> 2) & 3u);
    const float* Ab = Ps + a_lane + (ky * PWC + (kx == 1 ? ODD0 : (kx >> 1))) * LDA;
    const float* Bb = Bs + buf * B_TILE + b_lane;
    f32x4 a[2][TM], b[2][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(Ab + m * 2 * PWC * LDA);
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) b[0][nb] = *reinterpret_cast<const f32x4*>(Bb + nb * 32 * LDA);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 8) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
          a[nxt][m] = *reinterpret_cast<const f32x4*>(Ab + m * 2 * PWC * LDA + (kk + 1) * 8);
#pragma unroll
        for (int nb = 0; nb < TN; ++nb)
          b[nxt][nb] = *reinterpret_cast<const f32x4*>(Bb + nb * 32 * LDA + (kk + 1) * 8);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int nb = 0; nb < TN; ++nb)
            acc[m][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][m][r], b[cur][nb][r],
                                                              acc[m][nb], 0, 0, 0);
    }
    store_b(buf ^ 1);
  };
  for (int chunk = 0; chunk < chunks; ++chunk) {
    const bool dead = chunk + 1 >= chunks;           // (the last chunk stages nothing)
    const int chunk_n = dead ? chunk : chunk + 1;
    tap_step(std::true_type{}, 0, chunk, chunk_n, dead);
    __syncthreads();
#pragma nounroll
    for (int t = 1; t < 9; ++t) {
      tap_step(std::false_type{}, t, chunk, chunk_n, dead);
      if (t == 8 && !dead) {   // every wave is done with this chunk's patch
        __syncthreads();
        store_patch();
      }
      __syncthreads();
    }
  }

#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int col = n0 + wn0 + nb * 32 + li;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      float* o = p.out + (((size_t)n * Ho + (y0 + wrow0 + m)) * Wo + x0 + 4 * lh) * p.ldo + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nb][r] += bv;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[m][nb][r];
    }
  }
  if (p.stats) {   // uniform; the K loop ended on a barrier: the patch area is free scratch
    constexpr int WAVES_M = 4 / WAVES_N;
    float2* red = reinterpret_cast<float2*>(Ps);
    static_assert(WAVES_M * BN * 2 <= PPIX * LDA, "stats scratch fits in the patch area");
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const float2 mine = wave_col_stats<TM>([&](int m, int r) { return acc[m][nb][r]; });
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = mine;
    }
    float2 out;
    if (block_col_stats<BN, WAVES_M>(red, 0, 0, false, float2{0.f, 0.f}, 32.f * TM, out))
      p.stats[((size_t)n * p.stats_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
}

// ---------------------------------------------------------------------------
// Stride-2 data gradient, patch-staged (aten::convolution_backward(data) of the stride-2
// convolutions, Our_UNet/models/unet.py:106-115).  Position (a, b) of the dy grid produces the
// four dx pixels (2a+py, 2b+px), class = py*2+px; tap (ky, kx) feeds the class with
// py = (ky+1)&1, px = (kx+1)&1 from dy[a + (py+1-ky)/2][b + (px+1-kx)/2], i.e. a dy shift in
// {0,1}^2.  The (TH+1) x 33 dy pixels under a TH x 32 tile of positions are staged once per
// 32-channel chunk and serve all nine taps; per tap only the [BN][32] weight panel is
// re-staged (double buffered) and its 16 k-pairs x TM x TN MFMAs accumulate into the tap's
// class (4 x TM x TN accumulator blocks = 128 VGPRs per wave).  The nine taps are unrolled:
// the class of a step is a compile-time index.  Output tile = 2 TH x 64 dx pixels x BN channels.
// ---------------------------------------------------------------------------
struct S2PTap { int ky, kx, oy, ox, cls; };
__device__ constexpr S2PTap kS2PTaps[9] = {{1, 1, 0, 0, 0}, {1, 2, 0, 0, 1}, {2, 1, 0, 0, 2},
                                           {2, 2, 0, 0, 3}, {1, 0, 0, 1, 1}, {2, 0, 0, 1, 3},
                                           {0, 1, 1, 0, 2}, {0, 2, 1, 0, 3}, {0, 0, 1, 1, 3}};

template <int BN, int WM, int WN, int TH>
__global__ __launch_bounds__(256, 2) void conv_dgrad_s2_patch_kernel(const IgemmParams p) {
  constexpr int BK = 32, LDA = BK + 4;
  constexpr int TW = 32, PW = TW + 1, PH = TH + 1;
  constexpr int PPIX = PH * PW;
  constexpr int P_SLOTS = PPIX * 8;
  constexpr int P_PASSES = (P_SLOTS + 255) / 256;
  constexpr int B_SLOTS = BN * 8, B_PASSES = (B_SLOTS + 255) / 256;
  constexpr int B_TILE = BN * LDA;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN, WAVES_M = 4 / WAVES_N;
  static_assert((TH * 32 / WM) * (BN / WN) == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ps = smem;                          // [dy pixel][LDA]
  float* Bs = smem + PPIX * LDA;             // [buf][BN][LDA]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow0 = (wave / WAVES_N) * TM, wn0 = (wave % WAVES_N) * WN;

  const int Hl = p.Hl, Wl = p.Wl;            // dy grid
  const int tiles_n = p.Ncols / BN, tiles_x = Wl / TW, tiles_y = Hl / TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const int Ktot = p.C0;                     // Cout of the forward convolution

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  int pp_lin[P_PASSES], pp_lds[P_PASSES];
  unsigned pp_oob[P_PASSES];
#pragma unroll
  for (int i = 0; i < P_PASSES; ++i) {
    const int slot = (tid + 256 * i) % P_SLOTS;
    const int pix = slot >> 3, seg = slot & 7;
    const int prow = pix / PW, pcol = pix - prow * PW;
    const int iy = y0 + prow, ix = x0 + pcol;
    const bool ok = iy < Hl && ix < Wl;
    pp_lin[i] = ok ? ((n * Hl + iy) * Wl + ix) * 4 : 0;   // x channel count = byte offset
    pp_oob[i] = (ok ? 0u : 0x80000000u) | (unsigned)(seg * 16);
    pp_lds[i] = pix * LDA + seg * 4;
  }
  unsigned wslot_off[B_PASSES];
  int wslot_lds[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int slot = (tid + 256 * j) % B_SLOTS;
    const int row = slot >> 3, seg = slot & 7;
    wslot_off[j] = (unsigned)((p.n_off + n0 + row) * Ktot + seg * 4) * 4u;
    wslot_lds[j] = row * LDA + seg * 4;
  }

  f32x4 pr[P_PASSES], rb[B_PASSES];
  f32x16 acc[4][TM][TN];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int nb = 0; nb < TN; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][m][nb][r] = 0.f;

  auto load_patch = [&](int chunk) {
    const unsigned cbytes = (unsigned)(chunk * BK) * 4u;
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      const unsigned off = ((unsigned)(pp_lin[i] * Ktot) + cbytes) + pp_oob[i];
      pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) *reinterpret_cast<f32x4*>(Ps + pp_lds[i]) = pr[i];
  };
  auto load_b = [&](int wt, int chunk) {
    const unsigned woff = (unsigned)(wt * p.tap_stride + chunk * BK) * 4u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rsw, wslot_off[j] + woff, 0, 0));
  };
  auto store_b = [&](int buf) {
    float* Bb = Bs + buf * B_TILE;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bb + wslot_lds[j]) = rb[j];
  };

  const int chunks = Ktot / BK;
  load_patch(0);
  load_b(kS2PTaps[0].ky * 3 + kS2PTaps[0].kx, 0);
  store_patch();
  store_b(0);
  __syncthreads();

  const int a_lane = (wrow0 * PW + li) * LDA + 4 * lh;
  const int b_lane = (wn0 + li) * LDA + 4 * lh;
  for (int chunk = 0; chunk < chunks; ++chunk) {
    // (the last chunk re-stages itself; `dead` out-of-range loads and no second store measured
    // 0.5 % SLOWER on the step, A/B on one box - tools/ab_libs.sh)
    const int chunk_n = chunk + 1 < chunks ? chunk + 1 : chunk;
    for_range_p<0, 9>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr S2PTap T = kS2PTaps[t];
      constexpr S2PTap T1 = kS2PTaps[t == 8 ? 0 : t + 1];
      const int buf = (chunk + t) & 1;   // step = 9 * chunk + t
      load_b(T1.ky * 3 + T1.kx, t == 8 ? chunk_n : chunk);
      if (t == 0) load_patch(chunk_n);

      const float* Ab = Ps + a_lane + (T.oy * PW + T.ox) * LDA;
      const float* Bb = Bs + buf * B_TILE + b_lane;
      f32x4 a[2][TM], b[2][TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(Ab + m * PW * LDA);
#pragma unroll
      for (int nb = 0; nb < TN; ++nb) b[0][nb] = *reinterpret_cast<const f32x4*>(Bb + nb * 32 * LDA);
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 8) {
#pragma unroll
          for (int m = 0; m < TM; ++m)
            a[nxt][m] = *reinterpret_cast<const f32x4*>(Ab + m * PW * LDA + (kk + 1) * 8);
#pragma unroll
          for (int nb = 0; nb < TN; ++nb)
            b[nxt][nb] = *reinterpret_cast<const f32x4*>(Bb + nb * 32 * LDA + (kk + 1) * 8);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int nb = 0; nb < TN; ++nb)
              acc[T.cls][m][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  a[cur][m][r], b[cur][nb][r], acc[T.cls][m][nb], 0, 0, 0);
      }
      store_b(buf ^ 1);
      if (t == 8) {            // every wave is done with this chunk's patch
        __syncthreads();
        store_patch();
      }
      __syncthreads();
    });
  }

  // ---- epilogue: position (y0 + wrow0 + m, x0 + row), row = (reg&3) + 8*(reg>>2) + 4*lh;
  // class c -> dx pixel (2a + c/2, 2b + c%2), column li
  float s1[TN], s2[TN];
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) { s1[nb] = 0.f; s2[nb] = 0.f; }
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int col = n0 + wn0 + nb * 32 + li;
    BwdCoef cf{};
    if (p.bs_partial) cf = bwd_coef(p, n, col);   // uniform
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t pix = ((size_t)n * p.Hout + 2 * (y0 + wrow0 + m) + (c >> 1)) * p.Wout +
                           2 * (x0 + 4 * lh) + (c & 1);
        float* o = p.out + pix * p.ldo + col;
        if (p.accumulate) {        // uniform: all 16 reads in flight before the first add
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = o[(size_t)(2 * ((r & 3) + 8 * (r >> 2))) * p.ldo];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][m][nb][r] += old[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[(size_t)(2 * ((r & 3) + 8 * (r >> 2))) * p.ldo] = acc[c][m][nb][r];
        if (p.bs_partial) {   // uniform: sums of the next backward stage over the final values
          const float* yb = p.bs_y + pix * p.ldo + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float y = yb[(size_t)(2 * ((r & 3) + 8 * (r >> 2))) * p.ldo];
            const float z = fmaf(y, cf.A, cf.B0);
            const float gz = acc[c][m][nb][r] * cf.mk * (z > 0.f ? 1.f : p.slope);
            s1[nb] += gz;
            s2[nb] = fmaf(gz, (y - cf.mu) * cf.rs, s2[nb]);
          }
        }
      }
  }
  if (p.bs_partial) {   // uniform; the K loop ended on a barrier: the patch area is free scratch
    float2* red = reinterpret_cast<float2*>(Ps);
    static_assert(WAVES_M * BN * 2 <= PPIX * LDA, "reduction scratch fits in the patch area");
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const float a = s1[nb] + __shfl_xor(s1[nb], 32, 64);
      const float b = s2[nb] + __shfl_xor(s2[nb], 32, 64);
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = float2{a, b};
    }
    float2 out;
    if (block_col_sums<BN, WAVES_M>(red, out))
      p.bs_partial[((size_t)n * p.bs_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
}

// ---------------------------------------------------------------------------
// The stride-2 data gradient of the mixed-precision pipeline (bf16 dy / dx, bf16 matrix cores):
// geometry of conv_dgrad_s2_patch_kernel (dy patch of TH+1 rows x 33 columns staged once per
// 32-channel chunk, nine taps into four parity-class accumulators), LDS rows of 40 bf16, and -
// as a tap is only 2 x TM x TN MFMAs at the bf16 rate - three taps (three weight panels staged
// together) per barrier.  Replaces sixteen per-class gather-GEMM launches per step.
// ---------------------------------------------------------------------------
template <int BN, int WM, int WN, int TH, bool WB = false>
__global__ __launch_bounds__(256, 2) void conv_dgrad_s2_patch_b16_kernel(const IgemmParams p) {
  constexpr int BK = 32, LDA = BK + 8;
  constexpr int TW = 32, PW = TW + 1, PH = TH + 1;
  constexpr int PPIX = PH * PW;
  constexpr int P_SLOTS = PPIX * 8;          // 4-channel slots
  constexpr int P_PASSES = (P_SLOTS + 255) / 256;
  // WB (round 4): the weights pre-rounded to bf16 (p.w3): 8-channel slots, no conversion, and TWO
  // panel register sets - the panel of step s + 2 is in flight while that of s + 1 waits for its
  // LDS stage (with one set every step waited for the panel it had just asked for: matrix pipes
  // 6-9 % busy, waves waiting 61-72 % of their cycles)
  constexpr int B_SEGS = WB ? 4 : 8;
  constexpr int B_SETS = WB ? 2 : 1;
  constexpr int B_SLOTS = 3 * BN * B_SEGS, B_PASSES = (B_SLOTS + 255) / 256;
  constexpr int B_TILE = 3 * BN * LDA;       // three taps per step
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((TH * 32 / WM) * (BN / WN) == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];
  __bf16* Ps = smem_h;                       // [dy pixel][LDA]
  __bf16* Bs = smem_h + PPIX * LDA;          // [buf][tap in step][BN][LDA]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow0 = (wave / WAVES_N) * TM, wn0 = (wave % WAVES_N) * WN;

  const int Hl = p.Hl, Wl = p.Wl;            // dy grid
  const int tiles_n = p.Ncols / BN, tiles_x = Wl / TW, tiles_y = Hl / TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const int Ktot = p.C0;                     // Cout of the forward convolution

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = WB ? __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(p.w3), 0, (int)p.w3_bytes, 0x00020000)
                                        : __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // per slot ONE register: the linear dy pixel, or -1 for a slot outside the grid / past the
  // patch (its buffer load returns 0 and nothing is written); the LDS slot of pass i is
  // pass 0's + 32 pixels * i (256 threads = 32 pixels x 8 segments)
  int pp_lin[P_PASSES];
#pragma unroll
  for (int i = 0; i < P_PASSES; ++i) {
    const int slot = tid + 256 * i;
    const int pix = slot >> 3;
    const int prow = pix / PW, pcol = pix - prow * PW;
    const int iy = y0 + prow, ix = x0 + pcol;
    const bool ok = slot < P_SLOTS && iy < Hl && ix < Wl;
    pp_lin[i] = ok ? (n * Hl + iy) * Wl + ix : -1;
  }
  const int seg4 = (tid & 7) * 4;
  const int pp_lds0 = (tid >> 3) * LDA + seg4;
  unsigned wslot_off[B_PASSES];
  int wslot_lds[B_PASSES], wslot_tap[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int slot = (tid + 256 * j) % B_SLOTS;
    const int tr = slot / (BN * B_SEGS), rem = slot - tr * BN * B_SEGS;
    const int row = rem / B_SEGS, seg = rem % B_SEGS;
    wslot_tap[j] = tr;
    wslot_off[j] = (unsigned)((p.n_off + n0 + row) * Ktot + seg * (32 / B_SEGS)) * (WB ? 2u : 4u);
    wslot_lds[j] = (tr * BN + row) * LDA + seg * (32 / B_SEGS);
  }

  i32x2r pr[P_PASSES];           // raw bf16 (a plain operand goes to LDS as it came)
  f32x4 rb[B_SETS][B_PASSES];
  f32x16 acc[4][TM][TN];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int nb = 0; nb < TN; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][m][nb][r] = 0.f;

  auto to_bf16 = [](const f32x4 v) {
    bf16x4 h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    return h;
  };
  // (`dead`: out-of-range loads for the chunk after the last: a fixed load count per step)
  auto load_patch = [&](int chunk, bool dead = false) {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i)
      pr[i] = buf_ld4_raw16(rs0, (unsigned)(pp_lin[i] * Ktot + chunk * BK + seg4),
                            (pp_lin[i] < 0 || dead) ? 0x80000000u : 0u);
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i)
      if (256 * (i + 1) <= P_SLOTS || tid + 256 * i < P_SLOTS)
        *reinterpret_cast<i32x2r*>(Ps + pp_lds0 + i * 32 * LDA) = pr[i];
  };
  auto load_b = [&](auto step_tag, int chunk, auto setc) {   // the three taps 3*step .. 3*step+2
    constexpr int st = decltype(step_tag)::value;
    constexpr int SET = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const int wt = wslot_tap[j] == 0 ? kS2PTaps[3 * st].ky * 3 + kS2PTaps[3 * st].kx
                   : wslot_tap[j] == 1 ? kS2PTaps[3 * st + 1].ky * 3 + kS2PTaps[3 * st + 1].kx
                                       : kS2PTaps[3 * st + 2].ky * 3 + kS2PTaps[3 * st + 2].kx;
      const unsigned woff = (unsigned)(wt * p.tap_stride + chunk * BK) * (WB ? 2u : 4u);
      rb[SET][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 rsw, wslot_off[j] + woff, 0, 0));
    }
  };
  auto store_b = [&](int buf, auto setc) {
    constexpr int SET = decltype(setc)::value;
    __bf16* Bb = Bs + buf * B_TILE;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      if (256 * (j + 1) <= B_SLOTS || tid + 256 * j < B_SLOTS) {
        if constexpr (WB) *reinterpret_cast<f32x4*>(Bb + wslot_lds[j]) = rb[SET][j];   // 8 bf16
        else *reinterpret_cast<bf16x4*>(Bb + wslot_lds[j]) = to_bf16(rb[SET][j]);
      }
  };

  const int chunks = Ktot / BK;
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  load_patch(0);
  load_b(C0{}, 0, C0{});
  if constexpr (B_SETS == 2) load_b(C1{}, 0, C1{});
  store_patch();
  store_b(0, C0{});
  __syncthreads();

  const int a_lane = (wrow0 * PW + li) * LDA + 8 * lh;
  const int b_lane = (wn0 + li) * LDA + 8 * lh;
  // one K step (three taps, their output classes compile-time values); PH = s & 1 = the LDS stage
  // it reads and the register set that receives its panel load.  Two sets alternate with period
  // two, the steps of a chunk with period three: the loop body is two chunks (six steps) with a
  // three-step tail, every load unconditional (`dead` patch loads in the last chunk).
  auto dstep = [&](auto sc, auto phc, int chunk) __attribute__((always_inline)) {
    constexpr int st = decltype(sc)::value, PH = decltype(phc)::value;
    using LOADSET = std::integral_constant<int, B_SETS == 2 ? PH : 0>;
    using STORESET = std::integral_constant<int, B_SETS == 2 ? 1 - PH : 0>;
    constexpr int DIST = B_SETS == 2 ? 2 : 1;
    const bool more_chunks = chunk + 1 < chunks;
    const int buf = B_SETS == 2 ? PH : ((chunk + st) & 1);   // step = 3 * chunk + st
    {
      constexpr int stn = (st + DIST) % 3;
      const int chunkn = chunk + (st + DIST) / 3;
      load_b(std::integral_constant<int, stn>{}, chunkn < chunks ? chunkn : chunks - 1, LOADSET{});
    }
    if constexpr (st == 0) load_patch(more_chunks ? chunk + 1 : chunk, !more_chunks);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      constexpr S2PTap T0 = kS2PTaps[3 * st], T1 = kS2PTaps[3 * st + 1], T2 = kS2PTaps[3 * st + 2];
      const S2PTap T = j == 0 ? T0 : (j == 1 ? T1 : T2);
      const __bf16* Ab = Ps + a_lane + (T.oy * PW + T.ox) * LDA;
      const __bf16* Bb = Bs + buf * B_TILE + j * BN * LDA + b_lane;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
          a[m] = *reinterpret_cast<const bf16x8*>(Ab + m * PW * LDA + kk * 16);
#pragma unroll
        for (int nb = 0; nb < TN; ++nb)
          b[nb] = *reinterpret_cast<const bf16x8*>(Bb + nb * 32 * LDA + kk * 16);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int nb = 0; nb < TN; ++nb) {
            // T.cls is a compile-time value once j is unrolled
            if (T.cls == 0) acc[0][m][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[nb], acc[0][m][nb], 0, 0, 0);
            else if (T.cls == 1) acc[1][m][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[nb], acc[1][m][nb], 0, 0, 0);
            else if (T.cls == 2) acc[2][m][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[nb], acc[2][m][nb], 0, 0, 0);
            else acc[3][m][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[nb], acc[3][m][nb], 0, 0, 0);
          }
      }
    }
    store_b(buf ^ 1, STORESET{});
    if constexpr (st == 2) {
      if (more_chunks) {   // uniform; every wave is done with this chunk's patch
        __syncthreads();
        store_patch();
      }
    }
    __syncthreads();
  };
  if constexpr (B_SETS == 2) {
    int c = 0;
    for (; c + 2 <= chunks; c += 2) {
      dstep(C0{}, C0{}, c); dstep(C1{}, C1{}, c); dstep(C2{}, C0{}, c);
      dstep(C0{}, C1{}, c + 1); dstep(C1{}, C0{}, c + 1); dstep(C2{}, C1{}, c + 1);
    }
    if (c < chunks) { dstep(C0{}, C0{}, c); dstep(C1{}, C1{}, c); dstep(C2{}, C0{}, c); }
  } else {
    for (int c = 0; c < chunks; ++c) { dstep(C0{}, C0{}, c); dstep(C1{}, C0{}, c); dstep(C2{}, C0{}, c); }
  }

  // ---- epilogue: class c -> dx pixel (2a + c/2, 2b + c%2), column li; BSTATS (uniform
  // p.bs_partial) as in conv_dgrad_s2_patch_kernel: the sums of the next backward stage from the
  // fp32 accumulators and that layer's stored (bf16) raw outputs ----
  __bf16* outp = reinterpret_cast<__bf16*>(p.out);
  const __bf16* ybase = reinterpret_cast<const __bf16*>(p.bs_y);
  float s1[TN], s2[TN];
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) { s1[nb] = 0.f; s2[nb] = 0.f; }
#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int col = n0 + wn0 + nb * 32 + li;
    BwdCoef cf{};
    if (p.bs_partial) cf = bwd_coef(p, n, col);   // uniform
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t pix = ((size_t)n * p.Hout + 2 * (y0 + wrow0 + m) + (c >> 1)) * p.Wout +
                           2 * (x0 + 4 * lh) + (c & 1);
        __bf16* o = outp + pix * p.ldo + col;
        if (p.accumulate) {        // uniform
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            old[r] = (float)o[(size_t)(2 * ((r & 3) + 8 * (r >> 2))) * p.ldo];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][m][nb][r] += old[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[(size_t)(2 * ((r & 3) + 8 * (r >> 2))) * p.ldo] = (__bf16)acc[c][m][nb][r];
        if (p.bs_partial) {   // uniform
          // (two halves of eight values, fenced: with all sixteen y loads of a block in flight on
          // top of the 128 accumulator registers the 64-column instantiation spilled)
          const __bf16* yb = ybase + pix * p.ldo + col;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int r = 8 * h + q;
              const float y = (float)yb[(size_t)(2 * ((r & 3) + 8 * (r >> 2))) * p.ldo];
              const float z = fmaf(y, cf.A, cf.B0);
              const float gz = acc[c][m][nb][r] * cf.mk * (z > 0.f ? 1.f : p.slope);
              s1[nb] += gz;
              s2[nb] = fmaf(gz, (y - cf.mu) * cf.rs, s2[nb]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  }
  if (p.bs_partial) {   // uniform; the K loop ended on a barrier: the patch area is free scratch
    constexpr int WAVES_M = 4 / WAVES_N;
    float2* red = reinterpret_cast<float2*>(Ps);
    static_assert(WAVES_M * BN * 8 <= PPIX * LDA * 2, "reduction scratch fits in the patch area");
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const float a = s1[nb] + __shfl_xor(s1[nb], 32, 64);
      const float b = s2[nb] + __shfl_xor(s2[nb], 32, 64);
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = float2{a, b};
    }
    float2 out;
    if (block_col_sums<BN, WAVES_M>(red, out))
      p.bs_partial[((size_t)n * p.bs_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
}

// ---------------------------------------------------------------------------
// First convolution of a decoder stage: y = conv3x3(cat(upsample2x(act(low)), act(skip))) with the
// bilinear up-sampling done INSIDE the patch loader (Our_UNet/models/unet.py:215-231: the
// reference materialises both the up-sampled tensor and the concatenation).  Source 0 is the
// LOW-resolution raw tensor [N][H/2][W/2][C0]; per 32-channel chunk of it the (TH/2+2) x 18
// low-resolution pixels under the tile's patch are activated and staged into a small LDS
// scratch (coordinates clamped to the image: the align_corners=False stencil at exact 2x is
// {0.75, 0.25} with edge clamping), and every high-resolution patch pixel is blended from its
// 2 x 2 low-resolution neighbours with PyTorch's operation order
//   wy0 * (wx0 * p00 + wx1 * p01) + wy1 * (wx0 * p10 + wx1 * p11)
// (parity of the pixel picks 0.75 / 0.25); zero padding is applied after the blend.
// The nine taps are unrolled so the staging work of the NEXT chunk is spread behind this
// chunk's MFMAs: tap 0 issues the loads, tap 2 activates + writes the low-resolution scratch,
// taps 3..7 blend, tap 8 writes the patch.  Chunks of source 1 (the skip tensor) are staged as
// in conv_patch_f32_kernel.  Always the fused-layer form (activation on load, statistics).
// ---------------------------------------------------------------------------
template <int BN, int WM, int WN, int TH>
__global__ __launch_bounds__(256, 2) void conv_patch_up_kernel(const IgemmParams p) {
  constexpr int BK = 32, LDA = BK + 4;
  constexpr int TW = 32, PW = TW + 2;
  constexpr int PPIX = (TH + 2) * PW;
  constexpr int P_SLOTS = PPIX * 8;
  constexpr int P_PASSES = (P_SLOTS + 255) / 256;
  constexpr int B_SLOTS = BN * 8, B_PASSES = (B_SLOTS + 255) / 256;
  constexpr int B_TILE = BN * LDA;
  constexpr int LH = TH / 2 + 2, LW = TW / 2 + 2, LPIX = LH * LW;
  constexpr int L_SLOTS = LPIX * 8, L_PASSES = (L_SLOTS + 255) / 256;
  constexpr int BPER = (P_PASSES + 4) / 5;     // patch slots blended per tap (taps 3..7)
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((TH * 32 / WM) * (BN / WN) == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ps = smem;                          // [pixel][LDA]
  float* Bs = smem + PPIX * LDA;             // [buf][BN][LDA]
  float* Ls = Bs + 2 * B_TILE;               // [low pixel][LDA]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow0 = (wave / WAVES_N) * TM, wn0 = (wave % WAVES_N) * WN;
  const int seg = tid & 7;                   // this thread's 4 channels of a chunk (every pass)

  const int H = p.Hin, W = p.Win, h = H >> 1, w = W >> 1;
  const int tiles_n = p.Ncols / BN, tiles_x = W / TW, tiles_y = H / TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const int Ktot = p.C0 + p.C1;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // Slot geometry is DERIVED per use from two registers (a pass advances 32 pixels: 256 threads
  // x 4 channels of a 32-channel pixel) instead of held in per-slot arrays: this kernel carries
  // the patch, the low-resolution scratch values and the weight panel in registers at once.
  const int pix0 = tid >> 3;                 // patch pixel of pass 0 (pass i: + 32 i)
  // (the empty asm hides the value from loop-invariant code motion, which would otherwise
  // rebuild the per-slot arrays in registers and spill)
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto patch_slot = [&](int i, int& prow, int& pcol) {   // false: past the patch (last pass)
    const int pix = opaque(pix0) + 32 * i;
    prow = pix / PW;
    pcol = pix - prow * PW;
    return pix < PPIX;
  };
  auto low_slot = [&](int j, int& lpix) {    // clamped global pixel index (x4) of scratch slot j
    lpix = opaque(pix0) + 32 * j;
    const int lp = lpix < LPIX ? lpix : LPIX - 1;
    const int lr = lp / LW, lc = lp - lr * LW;
    int gy = (y0 >> 1) - 1 + lr, gx = (x0 >> 1) - 1 + lc;
    gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
    gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
    return ((n * h + gy) * w + gx) * 4;
  };
  const unsigned wslot_off0 = (unsigned)((p.n_off + n0 + pix0) * Ktot + seg * 4) * 4u;
  const int wslot_lds0 = pix0 * LDA + seg * 4;
  static_assert(BN % 32 == 0, "weight rows advance 32 per pass");

  f32x4 pr[P_PASSES], rb[B_PASSES], lr4[L_PASSES];
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  float cs = 1.f;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int nb = 0; nb < TN; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nb][r] = 0.f;

  auto load_coef = [&](const float* al, const float* be, int Cs, int cch) {
    if (al) {   // uniform
      const size_t o = (size_t)n * Cs + cch + seg * 4;
      ca = *reinterpret_cast<const f32x4*>(al + o);
      cb = *reinterpret_cast<const f32x4*>(be + o);
      cs = p.slope;
    } else {    // plain source: z = v, slope 1 = identity
      ca = f32x4{1.f, 1.f, 1.f, 1.f};
      cb = f32x4{0.f, 0.f, 0.f, 0.f};
      cs = 1.f;
    }
  };
  // next chunk from source 0 (low resolution) / source 1 (skip tensor, as conv_patch_f32_kernel)
  auto load_low = [&](int chunk) {
    const unsigned cbytes = (unsigned)(chunk * BK + seg * 4) * 4u;
#pragma unroll
    for (int j = 0; j < L_PASSES; ++j) {
      int lpix;
      const int lin = low_slot(j, lpix);
      lr4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                             rs0, (unsigned)(lin * p.C0) + cbytes, 0, 0));
    }
    load_coef(p.act0_alpha, p.act0_beta, p.C0, chunk * BK);
  };
  auto load_skip = [&](int chunk) {
    const int c = chunk * BK - p.C0;
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      int prow, pcol;
      const bool in = patch_slot(i, prow, pcol);
      const int iy = y0 - 1 + prow, ix = x0 - 1 + pcol;
      const bool ok = in && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const unsigned off = (unsigned)(((n * H + iy) * W + ix) * 4 * p.C1) +
                           (unsigned)(c + seg * 4) * 4u;
      pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rs1, ok ? off : 0x80000000u, 0, 0));
    }
    load_coef(p.act1_alpha, p.act1_beta, p.C1, c);
  };
  auto store_low = [&]() {     // activation of the low-resolution pixels, then the scratch
#pragma unroll
    for (int j = 0; j < L_PASSES; ++j) {
      const int lpix = pix0 + 32 * j;
      if (lpix < LPIX)
        *reinterpret_cast<f32x4*>(Ls + lpix * LDA + seg * 4) = act4f(lr4[j], ca, cb, cs, 1.f);
    }
  };
  auto slot_ok = [&](int i) {   // 1 inside the image, 0 for a zero-padding (or surplus) slot
    int prow, pcol;
    const bool in = patch_slot(i, prow, pcol);
    const int iy = y0 - 1 + prow, ix = x0 - 1 + pcol;
    return (in && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? 1.f : 0.f;
  };
  auto blend_slot = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    int prow, pcol;
    patch_slot(i, prow, pcol);
    const float* L = Ls + (((prow >> 1) * LW + (pcol >> 1)) * LDA + seg * 4);
    const f32x4 p00 = *reinterpret_cast<const f32x4*>(L);
    const f32x4 p01 = *reinterpret_cast<const f32x4*>(L + LDA);
    const f32x4 p10 = *reinterpret_cast<const f32x4*>(L + LW * LDA);
    const f32x4 p11 = *reinterpret_cast<const f32x4*>(L + LW * LDA + LDA);
    // odd patch row = even image row 2k: taps (k-1, k) weigh (0.25, 0.75); even patch row =
    // odd image row: (0.75, 0.25); columns alike (y0, x0 are even)
    const float wy1 = (prow & 1) ? 0.75f : 0.25f, wy0 = 1.f - wy1;
    const float wx1 = (pcol & 1) ? 0.75f : 0.25f, wx0 = 1.f - wx1;
    pr[i] = ((p00 * wx0 + p01 * wx1) * wy0 + (p10 * wx0 + p11 * wx1) * wy1) * slot_ok(i);
  };
  auto act_skip_slot = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    pr[i] = act4f(pr[i], ca, cb, cs, slot_ok(i));
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
      const int pix = pix0 + 32 * i;
      if (pix < PPIX) *reinterpret_cast<f32x4*>(Ps + pix * LDA + seg * 4) = pr[i];
    }
  };
  auto load_b = [&](int t, int chunk) {
    const unsigned woff = wslot_off0 + (unsigned)(t * p.tap_stride + chunk * BK) * 4u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rsw, woff + (unsigned)(32 * j * Ktot) * 4u, 0, 0));
  };
  auto store_b = [&](int buf) {
    float* Bb = Bs + buf * B_TILE + wslot_lds0;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bb + 32 * j * LDA) = rb[j];
  };

  const int chunks = Ktot / BK;
  const int up_chunks = p.C0 / BK;           // >= 1: chunk 0 always comes from source 0
  // ---- prologue: chunk 0 staged start to finish
  load_low(0);
  load_b(0, 0);
  store_low();
  store_b(0);
  __syncthreads();
  for_range_p<0, P_PASSES>(blend_slot);
  store_patch();
  __syncthreads();

  const int a_lane = ((wrow0 + 1) * PW + li + 1) * LDA + 4 * lh;
  const int b_lane = (wn0 + li) * LDA + 4 * lh;
  for (int chunk = 0; chunk < chunks; ++chunk) {
    const int nxt = chunk + 1 < chunks ? chunk + 1 : chunk;   // the last chunk re-stages itself
    const bool nxt_up = nxt < up_chunks;                       // uniform
    for_range_p<0, 9>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const int buf = (chunk + t) & 1;                         // 9 steps per chunk
      load_b(t == 8 ? 0 : t + 1, t == 8 ? nxt : chunk);
      if constexpr (t == 0) {
        if (nxt_up) load_low(nxt);
        else load_skip(nxt);
      }
      constexpr int oy = t / 3 - 1, ox = t % 3 - 1;
      const float* Ab = Ps + a_lane + (oy * PW + ox) * LDA;
      const float* Bb = Bs + buf * B_TILE + b_lane;
      f32x4 a[2][TM], b[2][TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(Ab + m * PW * LDA);
#pragma unroll
      for (int nb = 0; nb < TN; ++nb) b[0][nb] = *reinterpret_cast<const f32x4*>(Bb + nb * 32 * LDA);
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        const int cur = kk & 1, nx = cur ^ 1;
        if (kk + 1 < BK / 8) {
#pragma unroll
          for (int m = 0; m < TM; ++m)
            a[nx][m] = *reinterpret_cast<const f32x4*>(Ab + m * PW * LDA + (kk + 1) * 8);
#pragma unroll
          for (int nb = 0; nb < TN; ++nb)
            b[nx][nb] = *reinterpret_cast<const f32x4*>(Bb + nb * 32 * LDA + (kk + 1) * 8);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int nb = 0; nb < TN; ++nb)
              acc[m][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][m][r], b[cur][nb][r],
                                                                acc[m][nb], 0, 0, 0);
        // the next chunk's staging work, one piece per tap behind the first k-group's MFMAs
        if (kk == 0) {
          if constexpr (t == 2) {
            if (nxt_up) store_low();
          }
          if constexpr (t >= 3 && t <= 7) {
            constexpr int s0 = (t - 3) * BPER;
            constexpr int s1 = s0 + BPER < P_PASSES ? s0 + BPER : P_PASSES;
            if (nxt_up) for_range_p<s0, s1>(blend_slot);
            else for_range_p<s0, s1>(act_skip_slot);
          }
        }
      }
      store_b(buf ^ 1);
      if constexpr (t == 8) {    // every wave is done with this chunk's patch
        __syncthreads();
        store_patch();
      }
      __syncthreads();
    });
  }

#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int col = n0 + wn0 + nb * 32 + li;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      float* o = p.out + (((size_t)n * H + (y0 + wrow0 + m)) * W + x0 + 4 * lh) * p.ldo + col;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[m][nb][r] + bv;
    }
  }
  if (p.stats) {   // uniform; the K loop ended on a barrier: the patch area is free scratch
    constexpr int WAVES_M = 4 / WAVES_N;
    float2* red = reinterpret_cast<float2*>(Ps);
    static_assert(WAVES_M * BN * 2 <= PPIX * LDA, "stats scratch fits in the patch area");
#pragma unroll
    for (int nb = 0; nb < TN; ++nb) {
      const int col = n0 + wn0 + nb * 32 + li;
      const float bv = p.bias ? p.bias[col] : 0.f;
      const float2 mine = wave_col_stats<TM>([&](int m, int r) { return acc[m][nb][r] + bv; });
      if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + nb * 32 + li] = mine;
    }
    float2 out;
    if (block_col_stats<BN, WAVES_M>(red, 0, 0, false, float2{0.f, 0.f}, 32.f * TM, out))
      p.stats[((size_t)n * p.stats_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
}

template <int BN, int WM, int WN, int TH>
int launch_patch_up(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((TH + 2) * 34) * 36 + 2 * (size_t)BN * 36 +
                          (size_t)((TH / 2 + 2) * 18) * 36) * sizeof(float);
  auto kern = conv_patch_up_kernel<BN, WM, WN, TH>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long tiles = (long long)p.N * (p.Hin / TH) * (p.Win / 32) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_patch_up");
  return UNET_OK;
}

template <int BN, int WM, int WN, int TH, bool ACT = false, bool STATS = false, bool BSTATS = false>
int launch_patch_split(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds =
      (3 * (size_t)((TH + 2) * 34) * 24 + 2 * 3 * (size_t)BN * 24) * sizeof(__bf16);
  auto kern = conv_patch_split_kernel<BN, WM, WN, TH, ACT, STATS, BSTATS>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long tiles = (long long)p.N * (p.Hin / TH) * (p.Win / 32) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_patch_split");
  return UNET_OK;
}

template <int BN, int WM, int WN, int TH, bool ACT = false, bool STATS = false, bool BSTATS = false>
int launch_patch_f32(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((TH + 2) * 34) * 36 + 2 * (size_t)BN * 36) * sizeof(float);
  auto kern = conv_patch_f32_kernel<BN, WM, WN, TH, ACT, STATS, BSTATS>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long tiles = (long long)p.N * (p.Hin / TH) * (p.Win / 32) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_patch_f32");
  return UNET_OK;
}

template <int BN, int WM, int WN, int TH>
int launch_patch_s2(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((2 * TH + 1) * 65) * 20 + 2 * (size_t)BN * 20) * sizeof(float);
  auto kern = conv_patch_s2_kernel<BN, WM, WN, TH>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long tiles = (long long)p.N * (p.Hl / TH) * (p.Wl / 32) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_patch_s2");
  return UNET_OK;
}

template <int BN, int WM, int WN, int TH>
int launch_dgrad_s2_patch(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((TH + 1) * 33) * 36 + 2 * (size_t)BN * 36) * sizeof(float);
  auto kern = conv_dgrad_s2_patch_kernel<BN, WM, WN, TH>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long tiles = (long long)p.N * (p.Hl / TH) * (p.Wl / 32) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_dgrad_s2_patch");
  return UNET_OK;
}

}  // namespace

template <int BN, int WM, int WN, int TH>
int launch_dgrad_s2_patch_b16(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((TH + 1) * 33) * 40 + 2 * 3 * (size_t)BN * 40) * sizeof(__bf16);
  const long long tiles = (long long)p.N * (p.Hl / TH) * (p.Wl / 32) * (p.Ncols / BN);
  static const bool one_set = [] { const char* e = getenv("UNET_B16_S2_DGRAD_WB"); return e && e[0] == '0'; }();
  if (p.w3 && !one_set) {   // the weights pre-rounded to bf16: two panel sets
    auto kern = conv_dgrad_s2_patch_b16_kernel<BN, WM, WN, TH, true>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  } else {
    auto kern = conv_dgrad_s2_patch_b16_kernel<BN, WM, WN, TH, false>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_dgrad_s2_patch_b16");
  return UNET_OK;
}

// The same for the mixed-precision pipeline (bf16 dy / dx / bs_y); BSTATS as the fp32 form.
int launch_dgrad_s2_patch_b16_auto(const IgemmParams& p0, hipStream_t stream, int* bs_tiles_out) {
  IgemmParams p = p0;
  if (bs_tiles_out) *bs_tiles_out = 0;
  if (p.Wl % 32 != 0 || p.C0 % 32 != 0 || p.Hout != 2 * p.Hl || p.Wout != 2 * p.Wl) return 1;
  const long long pos = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  p.bs_tile0 = 0;
  if (!bs_tiles_out) p.bs_partial = nullptr;
  if (nc % 64 == 0 && p.Hl % 4 == 0 && pos / 128 * (nc / 64) >= 256) {
    p.bs_tiles = p.Hl * p.Wl / 128;
    if (bs_tiles_out) *bs_tiles_out = p.bs_partial ? p.bs_tiles : 0;
    return launch_dgrad_s2_patch_b16<64, 32, 64, 4>(p, stream);
  }
  if (nc == 32 && p.Hl % 8 == 0 && pos / 256 >= 256) {
    p.bs_tiles = p.Hl * p.Wl / 256;
    if (bs_tiles_out) *bs_tiles_out = p.bs_partial ? p.bs_tiles : 0;
    return launch_dgrad_s2_patch_b16<32, 64, 32, 8>(p, stream);
  }
  return 1;
}

// Stride-2 data gradient on the patch-staged kernel: p describes the dy grid (Hl x Wl, K = C0,
// fp32) and dx = (2 Hl) x (2 Wl) x Ncols.  Returns 1 when the shape does not tile or too few
// tiles would run (the caller keeps its gather-GEMM forms).  With p.bs_partial set the BSTATS
// epilogue runs and *bs_tiles_out receives the reduction tiles per image.
int launch_dgrad_s2_patch_auto(const IgemmParams& p0, hipStream_t stream, int* bs_tiles_out) {
  IgemmParams p = p0;
  if (p.Wl % 32 != 0 || p.C0 % 32 != 0 || p.Hout != 2 * p.Hl || p.Wout != 2 * p.Wl) return 1;
  const long long pos = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  p.bs_tile0 = 0;
  if (nc % 64 == 0 && p.Hl % 4 == 0 && pos / 128 * (nc / 64) >= 256) {
    p.bs_tiles = p.Hl * p.Wl / 128;
    if (bs_tiles_out) *bs_tiles_out = p.bs_partial ? p.bs_tiles : 0;
    return launch_dgrad_s2_patch<64, 32, 64, 4>(p, stream);
  }
  if (nc == 32 && p.Hl % 8 == 0 && pos / 256 >= 256) {
    p.bs_tiles = p.Hl * p.Wl / 256;
    if (bs_tiles_out) *bs_tiles_out = p.bs_partial ? p.bs_tiles : 0;
    return launch_dgrad_s2_patch<32, 64, 32, 8>(p, stream);
  }
  return 1;
}

// stride-2 3x3 forward whose OUTPUT tiles as 4 x 32 pixels, 16-channel chunks, standard taps
bool patch_s2_applicable(const IgemmParams& p) {
  IgemmParams std_taps{};
  for (int t = 0; t < 9; ++t) set_tap(std_taps, t, t / 3 - 1, t % 3 - 1, t);
  return p.ntaps == 9 && p.tap_cstride == 0 && p.src0_pitch == 0 && p.sin == 2 &&
         p.sout == 1 && p.Hin == 2 * p.Hl && p.Win == 2 * p.Wl && p.Hl == p.Hout &&
         p.Wl == p.Wout && p.Hl % 4 == 0 && p.Wl % 32 == 0 && p.C0 % 16 == 0 && p.C1 % 16 == 0 &&
         !p.accumulate && p.tapw[0] == std_taps.tapw[0] && p.tapw[1] == std_taps.tapw[1] &&
         p.tapw[2] == std_taps.tapw[2];
}

// fused forward with the up-sampling in the loader (source 0 = the low-resolution tensor)
template <int BN, int WM, int WN, int TH>
int launch_patch_b16_up_t(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((TH + 2) * 34) * 40 + 2 * 3 * (size_t)BN * 40) * sizeof(__bf16) +
                         (size_t)((TH / 2 + 2) * 18) * 36 * sizeof(float);
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  const long long tiles = (long long)p.N * (p.Hin / TH) * (p.Win / 32) * (p.Ncols / BN);
  // (only the pre-rounded weight panels, p.w3: the fp32-panel form of the 64-column tile spills)
  auto kern = conv_patch_b16_kernel<BN, WM, WN, TH, true, true, false, true, true>;
  UNET_SET_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_patch_b16(up)");
  return UNET_OK;
}

// y = conv3x3(cat(upsample2x(act(low)), act(skip))) on bf16 tensors: p.src0 = low
// [N][Hin/2][Win/2][C0], p.src1 = skip [N][Hin][Win][C1], both activated on load.  Returns 1 when
// no tile shape fits (the caller materialises the up-sampled tensor).
int launch_patch_b16_up_auto(const IgemmParams& p0, hipStream_t stream, int* stats_px) {
  IgemmParams p = p0;
  if (p.Hin % 4 || p.Win % 32 || p.C0 % 32 || p.C1 % 32 || p.C0 < 32 || !p.act0_alpha ||
      (p.C1 && !p.act1_alpha) || !p.w3 || p.accumulate || (p.ldo & 7) ||
      (reinterpret_cast<uintptr_t>(p.out) & 15))
    return 1;
  const long long M = (long long)p.N * p.Hin * p.Win, mt = M / 128;
  const int nc = p.Ncols;
  p.bs_partial = nullptr;
  if (nc % 64 == 0 && p.Hin % 8 == 0 && (M / 256) * (nc / 64) >= 512) {
    *stats_px = p.stats ? 256 : 0; p.stats_tiles = p.Hin * p.Win / 256;
    return launch_patch_b16_up_t<64, 64, 64, 8>(p, stream);
  }
  if (nc % 64 == 0 && mt * (nc / 64) >= 256) {
    *stats_px = p.stats ? 128 : 0; p.stats_tiles = p.Hin * p.Win / 128;
    return launch_patch_b16_up_t<64, 64, 32, 4>(p, stream);
  }
  if (nc == 32 && p.Hin % 8 == 0 && (M / 256) >= 256) {
    *stats_px = p.stats ? 256 : 0; p.stats_tiles = p.Hin * p.Win / 256;
    return launch_patch_b16_up_t<32, 64, 32, 8>(p, stream);
  }
  return 1;
}

template <int BN, int WM, int WN, int TH, bool ACT, bool STATS, bool BSTATS = false>
int launch_patch_b16_t(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((TH + 2) * 34) * 40 + 2 * 3 * (size_t)BN * 40) * sizeof(__bf16);
  const long long tiles = (long long)p.N * (p.Hin / TH) * (p.Win / 32) * (p.Ncols / BN);
  if (p.w3) {     // the weights pre-rounded to bf16: panels staged without conversion
    auto kern = conv_patch_b16_kernel<BN, WM, WN, TH, ACT, STATS, BSTATS, true>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  } else if constexpr (TH == 8 && BN == 64) {
    // (the 8-row 64-column tile exists for the pre-rounded panels only - launch_patch_b16_auto
    // does not pick it without them: its fused form with fp32 panels spills)
    unet_set_error("conv_patch_b16: the 8 x 32-pixel tile needs the bf16 weight plane");
    return UNET_E_INVALID;
  } else {
    auto kern = conv_patch_b16_kernel<BN, WM, WN, TH, ACT, STATS, BSTATS, false>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_patch_b16");
  return UNET_OK;
}

// Mixed-precision pipeline (bf16 tensors): stride-1 3x3 whose image tiles as 4 x 32 pixels.
// stats_px != nullptr = fused forward (activation on load + statistics), else data gradient.
// Returns 1 when the shape does not qualify (the caller keeps the bf16 gather-GEMM).
int launch_patch_b16_auto(const IgemmParams& p0, hipStream_t stream, int* stats_px, int* bs_px) {
  if (bs_px) *bs_px = 0;
  if (!patch_f32_applicable(p0) || p0.src0_pitch) return 1;
  IgemmParams p = p0;
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  const long long mt = M / 128;
  const bool fused = stats_px != nullptr;
  // 8 x 32-pixel tiles of 64 columns where they still give two workgroups per CU: half the
  // weight-panel traffic (L2 -> LDS) per output of the 4 x 32 tiles; measured -2..-16 % per
  // launch on the 64..256-channel layers, -0.13 ms per step (profiles/r04_bf16_experiments.txt)
  const bool th8 = p.w3 && nc % 64 == 0 && p.Hin % 8 == 0 && (M / 256) * (nc / 64) >= 512;
  if (!fused && bs_px && p.bs_partial) {   // data gradient with the BSTATS epilogue
    p.bs_tile0 = 0;
    if (th8) {
      *bs_px = 256; p.bs_tiles = p.Hin * p.Win / 256;
      return launch_patch_b16_t<64, 64, 64, 8, false, false, true>(p, stream);
    }
    if (nc % 128 == 0 && mt * (nc / 128) >= 256) {
      *bs_px = 128; p.bs_tiles = p.Hin * p.Win / 128;
      return launch_patch_b16_t<128, 64, 64, 4, false, false, true>(p, stream);
    }
    if (nc % 64 == 0 && mt * (nc / 64) >= 256) {
      *bs_px = 128; p.bs_tiles = p.Hin * p.Win / 128;
      return launch_patch_b16_t<64, 64, 32, 4, false, false, true>(p, stream);
    }
    if (nc == 32 && p.Hin % 8 == 0 && (M / 256) >= 256) {
      *bs_px = 256; p.bs_tiles = p.Hin * p.Win / 256;
      return launch_patch_b16_t<32, 64, 32, 8, false, false, true>(p, stream);
    }
    return 1;
  }
  p.bs_partial = nullptr;
  if (th8) {
    if (!fused) return launch_patch_b16_t<64, 64, 64, 8, false, false>(p, stream);
    *stats_px = p.stats ? 256 : 0; p.stats_tiles = p.Hin * p.Win / 256;
    return launch_patch_b16_t<64, 64, 64, 8, true, true>(p, stream);
  }
  if (nc % 128 == 0 && mt * (nc / 128) >= 256) {
    if (!fused) return launch_patch_b16_t<128, 64, 64, 4, false, false>(p, stream);
    *stats_px = p.stats ? 128 : 0; p.stats_tiles = p.Hin * p.Win / 128;
    return launch_patch_b16_t<128, 64, 64, 4, true, true>(p, stream);
  }
  if (nc % 64 == 0 && mt * (nc / 64) >= 256) {
    if (!fused) return launch_patch_b16_t<64, 64, 32, 4, false, false>(p, stream);
    *stats_px = p.stats ? 128 : 0; p.stats_tiles = p.Hin * p.Win / 128;
    return launch_patch_b16_t<64, 64, 32, 4, true, true>(p, stream);
  }
  if (nc == 32 && p.Hin % 8 == 0 && (M / 256) >= 256) {
    if (!fused) return launch_patch_b16_t<32, 64, 32, 8, false, false>(p, stream);
    *stats_px = p.stats ? 256 : 0; p.stats_tiles = p.Hin * p.Win / 256;
    return launch_patch_b16_t<32, 64, 32, 8, true, true>(p, stream);
  }
  return 1;
}

// Fused-layer stride-2 forward on the patch-staged kernel; returns 1 when no tile shape fills
// the chip (the caller falls back to the gather-GEMM).  *stats_px = pixels per statistics tile.
// Stride-2 fused forward on bf16 tensors (round 4): conv_patch_b16_kernel<.., SD = 2>.  Returns 1
// when the shape does not tile or fill the chip (the caller keeps the bf16 gather-GEMM).
template <int BN, int WM, int WN, int TH>
int launch_patch_b16_s2_t(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((2 * TH + 1) * 65) * 40 + 2 * 3 * (size_t)BN * 40) * sizeof(__bf16);
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  const long long tiles = (long long)p.N * (p.Hl / TH) * (p.Wl / 32) * (p.Ncols / BN);
  auto kern = conv_patch_b16_kernel<BN, WM, WN, TH, true, true, false, true, false, 2>;
  UNET_SET_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_patch_b16(stride 2)");
  return UNET_OK;
}
int launch_patch_s2_b16_auto(const IgemmParams& p0, hipStream_t stream, int* stats_px) {
  static const bool off = [] { const char* e = getenv("UNET_B16_S2_PATCH"); return e && e[0] == '0'; }();
  IgemmParams p = p0;
  if (off || !p.w3 || !patch_s2_applicable(p) || p.C0 % 32 || p.C1 % 32 || (p.ldo & 7) ||
      (reinterpret_cast<uintptr_t>(p.out) & 15))
    return 1;
  const long long mt = (long long)p.N * p.Hl * p.Wl / 128;
  const int nc = p.Ncols;
  if (nc % 64 == 0 && mt * (nc / 64) >= 512) {
    *stats_px = p.stats ? 128 : 0;
    p.stats_tiles = p.Hl * p.Wl / 128;
    p.bs_partial = nullptr;
    return launch_patch_b16_s2_t<64, 64, 32, 4>(p, stream);
  }
  return 1;
}

int launch_patch_s2_auto(const IgemmParams& p0, hipStream_t stream, int* stats_px) {
  IgemmParams p = p0;
  const long long mt = (long long)p.N * p.Hl * p.Wl / 128;
  const int nc = p.Ncols;
  *stats_px = p.stats ? 128 : 0;
  p.stats_tiles = p.Hl * p.Wl / 128;
  if (nc % 128 == 0 && mt * (nc / 128) >= 512) return launch_patch_s2<128, 64, 64, 4>(p, stream);
  if (nc % 64 == 0 && mt * (nc / 64) >= 512) return launch_patch_s2<64, 64, 32, 4>(p, stream);
  *stats_px = 0;
  return 1;
}

// stride-1 3x3 over an image that tiles as 4 x 32 pixels, 16-channel chunks
bool patch_split_applicable(const IgemmParams& p) {
  return p.ntaps == 9 && p.sin == 1 && p.sout == 1 && p.Hl == p.Hin && p.Wl == p.Win &&
         p.Hl == p.Hout && p.Wl == p.Wout && p.Hin % 4 == 0 && p.Win % 32 == 0 &&
         p.C0 % 16 == 0 && p.C1 % 16 == 0;
}

// the same with 32-channel chunks (fp32 form)
bool patch_f32_applicable(const IgemmParams& p) {
  return p.ntaps == 9 && p.tap_cstride == 0 && p.sin == 1 && p.sout == 1 &&
         p.Hl == p.Hin && p.Wl == p.Win && p.Hl == p.Hout && p.Wl == p.Wout && p.Hin % 4 == 0 &&
         p.Win % 32 == 0 && p.C0 % 32 == 0 && p.C1 % 32 == 0;
}

// fp32: measured on the net's layers +5..19 % over the gather-GEMM at 64 and 128 columns, +8 %
// at 32 columns when K > 32 (K = 32 stays on the row-fused kernel).  Returns 1 when no tile
// shape fits (too few tiles): the caller falls back to the gather-GEMM.
// stats_px != nullptr selects the fused-layer instantiation (activation on load, statistics
// epilogue into p.stats) and receives the number of pixels per statistics tile.
// bs_px != nullptr (data gradient whose output is final for a layer): the BSTATS epilogue
// (p.bs_*) runs and *bs_px receives the pixels per reduction tile.
int launch_patch_f32_auto(const IgemmParams& p0, hipStream_t stream, int* stats_px, int* bs_px) {
  IgemmParams p = p0;
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  const long long mt = M / 128;
  const bool fused = stats_px != nullptr;
  if (!fused && bs_px && p.bs_partial) {
    p.bs_tile0 = 0;
    if (nc % 128 == 0 && mt * (nc / 128) >= 512) {
      *bs_px = 128; p.bs_tiles = p.Hin * p.Win / 128;
      return launch_patch_f32<128, 64, 64, 4, false, false, true>(p, stream);
    }
    if (nc % 64 == 0 && mt * (nc / 64) >= 512) {
      *bs_px = 128; p.bs_tiles = p.Hin * p.Win / 128;
      return launch_patch_f32<64, 64, 32, 4, false, false, true>(p, stream);
    }
    if (nc == 32 && p.Hin % 8 == 0 && (M / 256) >= 512) {
      *bs_px = 256; p.bs_tiles = p.Hin * p.Win / 256;
      return launch_patch_f32<32, 64, 32, 8, false, false, true>(p, stream);
    }
    return 1;
  }
  if (nc % 128 == 0 && mt * (nc / 128) >= 512) {
    if (!fused) return launch_patch_f32<128, 64, 64, 4>(p, stream);
    *stats_px = p.stats ? 128 : 0; p.stats_tiles = p.Hin * p.Win / 128;
    return launch_patch_f32<128, 64, 64, 4, true, true>(p, stream);
  }
  if (nc % 64 == 0 && mt * (nc / 64) >= 512) {
    if (!fused) return launch_patch_f32<64, 64, 32, 4>(p, stream);
    *stats_px = p.stats ? 128 : 0; p.stats_tiles = p.Hin * p.Win / 128;
    return launch_patch_f32<64, 64, 32, 4, true, true>(p, stream);
  }
  if (nc == 32 && p.Hin % 8 == 0 && (M / 256) >= 512) {
    if (!fused) return launch_patch_f32<32, 64, 32, 8>(p, stream);
    *stats_px = p.stats ? 256 : 0; p.stats_tiles = p.Hin * p.Win / 256;
    return launch_patch_f32<32, 64, 32, 8, true, true>(p, stream);
  }
  return 1;
}

// conv3x3(cat(upsample2x(act(src0 at half resolution)), act(src1))): the up-sampling in the
// loader.  Same tile choice as launch_patch_f32_auto; returns 1 when no tile fits (the caller
// then materialises the up-sampled operand).
int launch_patch_up_auto(const IgemmParams& p0, hipStream_t stream, int* stats_px) {
  IgemmParams p = p0;
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  const long long mt = M / 128;
  if (p.Hin % 4 != 0 || p.Win % 32 != 0 || p.C0 % 32 != 0 || p.C1 % 32 != 0 || p.C0 < 32) return 1;
  if (nc % 128 == 0 && mt * (nc / 128) >= 512) {
    *stats_px = p.stats ? 128 : 0; p.stats_tiles = p.Hin * p.Win / 128;
    return launch_patch_up<128, 64, 64, 4>(p, stream);
  }
  if (nc % 64 == 0 && mt * (nc / 64) >= 512) {
    *stats_px = p.stats ? 128 : 0; p.stats_tiles = p.Hin * p.Win / 128;
    return launch_patch_up<64, 64, 32, 4>(p, stream);
  }
  if (wino_up32_applicable(p)) {   // conv_c32.hip: K = 96 as three register-resident Winograd chunks
    *stats_px = p.stats ? 256 : 0; p.stats_tiles = p.Hin * p.Win / 256;
    return launch_wino_up32(p, stream);
  }
  if (nc == 32 && p.Hin % 8 == 0 && (M / 256) >= 512) {
    *stats_px = p.stats ? 256 : 0; p.stats_tiles = p.Hin * p.Win / 256;
    return launch_patch_up<32, 64, 32, 8>(p, stream);
  }
  return 1;
}

// The split mode of the fused layer pipeline: stats_px != nullptr = fused forward (activation
// on load, statistics epilogue), bs_px != nullptr = data gradient with the BSTATS epilogue.
// Returns 1 when no tile shape fills the chip (the caller runs the fp32 fused kernel instead).
int launch_patch_split_fused_auto(const IgemmParams& p0, hipStream_t stream, int* stats_px,
                                  int* bs_px) {
  IgemmParams p = p0;
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  const bool fwd = stats_px != nullptr;
  const bool tall = p.Hin % 8 == 0;
  int px = 0;
  int cls = 0;   // 1: <128,64,64,4>  2: <64,128,32,8>  3: <64,64,32,4>  4: <32,64,32,8>
  if (nc % 128 == 0 && (M / 128) * (nc / 128) >= 512) { cls = 1; px = 128; }
  else if (nc % 64 == 0 && tall && (M / 256) * (nc / 64) >= 512) { cls = 2; px = 256; }
  else if (nc % 64 == 0 && (M / 128) * (nc / 64) >= 256) { cls = 3; px = 128; }
  else if (nc == 32 && tall && (M / 256) >= 512) { cls = 4; px = 256; }
  if (!cls) return 1;
  if (fwd) {
    *stats_px = p.stats ? px : 0;
    p.stats_tiles = p.Hin * p.Win / px;
    p.bs_partial = nullptr;
    switch (cls) {
      case 1: return launch_patch_split<128, 64, 64, 4, true, true, false>(p, stream);
      case 2: return launch_patch_split<64, 128, 32, 8, true, true, false>(p, stream);
      case 3: return launch_patch_split<64, 64, 32, 4, true, true, false>(p, stream);
      default: return launch_patch_split<32, 64, 32, 8, true, true, false>(p, stream);
    }
  }
  p.stats = nullptr;
  if (bs_px && p.bs_partial) { *bs_px = px; p.bs_tiles = p.Hin * p.Win / px; p.bs_tile0 = 0; }
  else { if (bs_px) *bs_px = 0; p.bs_partial = nullptr; }
  switch (cls) {
    case 1: return launch_patch_split<128, 64, 64, 4, false, false, true>(p, stream);
    case 2: return launch_patch_split<64, 128, 32, 8, false, false, true>(p, stream);
    case 3: return launch_patch_split<64, 64, 32, 4, false, false, true>(p, stream);
    default: return launch_patch_split<32, 64, 32, 8, false, false, true>(p, stream);
  }
}

int launch_patch_split_auto(const IgemmParams& p, hipStream_t stream) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  const long long mt = M / 128;
  if (nc % 128 == 0 && mt * (nc / 128) >= 512)
    return launch_patch_split<128, 64, 64, 4>(p, stream);
  // narrow outputs: taller tiles (two or four patch rows per wave) cut the LDS reads per MFMA
  const bool tall = p.Hin % 8 == 0;
  if (nc % 64 == 0) {
    if (tall && (M / 256) * (nc / 64) >= 512) return launch_patch_split<64, 128, 32, 8>(p, stream);
    return launch_patch_split<64, 64, 32, 4>(p, stream);
  }
  if (tall && (M / 256) * (nc / 32) >= 512) return launch_patch_split<32, 64, 32, 8>(p, stream);
  return launch_patch_split<32, 32, 32, 4>(p, stream);
}

}  // namespace unet_conv

#ifdef B16_STAMPS
extern "C" int unet_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(unet_conv::g_stamps), sizeof(unsigned long long) * 256 * 16);
}
#endif

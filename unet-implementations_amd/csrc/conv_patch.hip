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

namespace unet_conv {
namespace {

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
template <int BN, int WM, int WN, int TH>
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
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < P_PASSES; ++i) {
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
        for (int r = 0; r < 16; ++r)
          o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[m][nb][r] + bv + old[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[m][nb][r] + bv;
      }
    }
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
template <int BN, int WM, int WN, int TH, bool ACT = false, bool STATS = false>
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
  const int steps = chunks * 9;
  load_patch(0);
  load_b(0, 0);
  store_patch();
  store_b(0);
  __syncthreads();

  // fragment addresses: lane (li, lh) reads 4 consecutive k of its row at k offset 4*lh;
  // MFMA r of a k-group of 8 multiplies k = 8*kk + 4*lh + r on both operands
  const int a_lane = ((wrow0 + 1) * PW + li + 1) * LDA + 4 * lh;
  const int b_lane = (wn0 + li) * LDA + 4 * lh;
  int t = 0, chunk = 0;
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    const int t1 = (t == 8) ? 0 : t + 1;
    const int chunk1 = (t == 8) ? chunk + 1 : chunk;
    const bool more = s + 1 < steps;
    load_b(more ? t1 : t, more ? chunk1 : chunk);
    if (t == 0) load_patch(chunk + 1 < chunks ? chunk + 1 : chunk);

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
    if (t == 8) {            // every wave is done with this chunk's patch
      __syncthreads();
      store_patch();
    }
    __syncthreads();
    t = t1;
    chunk = chunk1;
  }

#pragma unroll
  for (int nb = 0; nb < TN; ++nb) {
    const int col = n0 + wn0 + nb * 32 + li;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      float* o = p.out + (((size_t)n * H + (y0 + wrow0 + m)) * W + x0 + 4 * lh) * p.ldo + col;
      if (p.accumulate) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[m][nb][r] + bv + old[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo] = acc[m][nb][r] + bv;
      }
    }
  }
  if (STATS && p.stats) {   // uniform
    // the K loop ended on a barrier: the patch area is free scratch
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
int launch_patch_split(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds =
      (3 * (size_t)((TH + 2) * 34) * 24 + 2 * 3 * (size_t)BN * 24) * sizeof(__bf16);
  auto kern = conv_patch_split_kernel<BN, WM, WN, TH>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long tiles = (long long)p.N * (p.Hin / TH) * (p.Win / 32) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_patch_split");
  return UNET_OK;
}

template <int BN, int WM, int WN, int TH, bool ACT = false, bool STATS = false>
int launch_patch_f32(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = ((size_t)((TH + 2) * 34) * 36 + 2 * (size_t)BN * 36) * sizeof(float);
  auto kern = conv_patch_f32_kernel<BN, WM, WN, TH, ACT, STATS>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long tiles = (long long)p.N * (p.Hin / TH) * (p.Win / 32) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_patch_f32");
  return UNET_OK;
}

}  // namespace

// stride-1 3x3 over an image that tiles as 4 x 32 pixels, 16-channel chunks
bool patch_split_applicable(const IgemmParams& p) {
  static const int off = getenv("UNET_NO_PATCH") ? 1 : 0;
  return !off && p.ntaps == 9 && p.sin == 1 && p.sout == 1 && p.Hl == p.Hin && p.Wl == p.Win &&
         p.Hl == p.Hout && p.Wl == p.Wout && p.Hin % 4 == 0 && p.Win % 32 == 0 &&
         p.C0 % 16 == 0 && p.C1 % 16 == 0;
}

// the same with 32-channel chunks (fp32 form)
bool patch_f32_applicable(const IgemmParams& p) {
  static const int off = getenv("UNET_NO_PATCH") ? 1 : 0;
  return !off && p.ntaps == 9 && p.tap_cstride == 0 && p.sin == 1 && p.sout == 1 &&
         p.Hl == p.Hin && p.Wl == p.Win && p.Hl == p.Hout && p.Wl == p.Wout && p.Hin % 4 == 0 &&
         p.Win % 32 == 0 && p.C0 % 32 == 0 && p.C1 % 32 == 0;
}

// fp32: measured on the net's layers +5..19 % over the gather-GEMM at 64 and 128 columns, +8 %
// at 32 columns when K > 32 (K = 32 stays on the row-fused kernel).  Returns 1 when no tile
// shape fits (too few tiles): the caller falls back to the gather-GEMM.
// stats_px != nullptr selects the fused-layer instantiation (activation on load, statistics
// epilogue into p.stats) and receives the number of pixels per statistics tile.
int launch_patch_f32_auto(const IgemmParams& p0, hipStream_t stream, int* stats_px) {
  IgemmParams p = p0;
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  const long long mt = M / 128;
  const bool fused = stats_px != nullptr;
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

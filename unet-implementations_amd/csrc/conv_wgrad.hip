// conv_wgrad.hip — 3x3 convolution weight gradient on the fp32 matrix cores.
//
// dW[tap][ci][co] = sum over output pixels p of x[p @ tap][ci] * dy[p][co].
// One workgroup = 4 waves; each wave owns one 32x32 (ci,co) sub-block for all nine taps; the
// block owns a (CI_T x CO_T) channel tile and a contiguous range of output-row segments of
// S pixels.  Per
// segment the 3-row input halo patch [3][(S-1)*stride+3][CI_T] and the dy
// segment [S][CO_T] are register-staged into double-buffered LDS once and
// shared by all 9 taps (each input element is fetched ~1.1x instead of 9x).
// MFMA operands: A = patch^T (rows = ci, k = pixel pair), B = dy (k = pixel
// pair, cols = co); both are plain conflict-free ds_read_b32 (lanes run along
// the contiguous channel axis).  Partial sums per pixel-range split are written
// as slabs and reduced (fixed order => run-to-run deterministic) by
// wgrad_reduce_kernel, which also scatters into the OIHW gradient.
//
// Replaces the weight-gradient half of aten::convolution_backward reached from
// loss.backward() (Our_UNet/src/train.py:663).
#include "conv_params.h"
#include <vector>
#include "lds_asm.h"
#include <utility>

namespace {
using unet_conv::act4;
using unet_conv::act4f;
template <int I> using template_ic = std::integral_constant<int, I>;
// compile-time loop: f(integral_constant<int, I>) for I in [B, E)
template <int B, int... I, typename F>
__device__ __forceinline__ void for_range_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(template_ic<B + I>{}), ...);
}
template <int B, int E, typename F>
__device__ __forceinline__ void for_range(F&& f) {
  for_range_impl<B>(std::make_integer_sequence<int, (E > B ? E - B : 0)>{}, f);
}

struct WgradParams {
  const float* x;   // [N][H][W][Cx]
  const float* dy;  // [N][Ho][Wo][Cout]
  float* partial;   // [split][9][Cx][Cout]
  int Cx, Cout;
  int N, H, W, Ho, Wo;
  int segs_per_row, total_segs, segs_per_block, split;
  int ci_tiles, co_tiles;
  unsigned x_bytes, dy_bytes;  // buffer-descriptor ranges (< 2 GiB each)
  // fused layer pipeline (ACT instantiations): x holds the RAW output of the producing
  // convolution; a = lrelu(x * alpha[n][c] + beta[n][c], slope) is applied while the patch is
  // staged (zero padding stays zero)
  const float* alpha;  // [N][Cx]
  const float* beta;   // [N][Cx]
  float slope;
  int b16;             // x and dy are bf16 tensors (mixed-precision pipeline)
  // conv_wgrad_wino32_kernel<.., DZ>: `dy` holds g = dL/da (gradient w.r.t. the layer's ACTIVATED
  // output); the kernel forms dz = dL/dy of the layer's InstanceNorm + LeakyReLU + dropout
  // backward while it loads its dy tiles - dz = (z > 0 ? P : P slope) g + (Q y + R),
  // z = y a1 + b1, coefficient planes [5][N][C] from unet_instnorm_bwd_coefs - uses it, and
  // WRITES it to dz_out (may alias dy: every pixel is read once) for the data gradient.
  const float* dz_y;      // raw conv output y of the layer, [N][H][W][Cout]
  const float* dz_coef;   // [5][N][Cout]: a1, b1, P, Q, R
  float* dz_out;
  const float2* dz_sums;  // [N][Cout] (S1, S2)
  const float* dz_gamma;  // [Cout]
  const float* dz_rstd;   // [N][Cout]
  float* dz_dgamma; float* dz_dbeta; float* dz_dbias;   // [Cout] each (may be null)
  float dz_slope;
};

// Epilogue shared by the weight-gradient kernels.  A wave holds nine 32x32 accumulator blocks
// (taps t = 0..8) of sub-block sb for its pixel part pp.  The NPP parts of a sub-block are
// MERGED THROUGH LDS (three taps per round, parts added in part order: deterministic) and the
// sums leave the CU as 16-byte stores - ONE slab per workgroup: partial[sp][9][Cx][Cout]
// (NPP == 1: the round trip through LDS only widens the stores).  `smem` must hold
// 3 * NW * 4 KB and be free (the main loops end on a barrier).
// acc_of(integral_constant<t>) returns the block of tap t.
template <int NSB, int NPP, int TJ, int NT, typename ACC>
__device__ __forceinline__ void wgrad_epilogue(float* smem, const WgradParams& p, int sp, int ci0,
                                               int co0, int sb, int pp, ACC&& acc_of) {
  const int tid = threadIdx.x, lane = tid & 63;
  {
    constexpr int TPR = 3;
    const f32x4* M4 = reinterpret_cast<const f32x4*>(smem);
    for_range<0, 3>([&](auto uc) {          // round u = taps 3u .. 3u+2
      constexpr int u = decltype(uc)::value;
      for_range<0, TPR>([&](auto vc) {
        constexpr int v = decltype(vc)::value;
        const f32x16& a = acc_of(template_ic<u * 3 + v>{});
        float* dst = smem + ((v * NPP + pp) * NSB + sb) * 1024 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64] = a[r];
      });
      __syncthreads();
      for (int i = tid; i < TPR * NSB * 256; i += NT) {
        const int v = i / (NSB * 256), rem = i - v * (NSB * 256);
        const int s = rem >> 8, e4 = rem & 255;
        f32x4 sum = M4[((v * NPP) * NSB + s) * 256 + e4];
#pragma unroll
        for (int q = 1; q < NPP; ++q) sum += M4[((v * NPP + q) * NSB + s) * 256 + e4];
        // element e = 4*e4 of a 32x32 block: register r = e >> 6, lane = e & 63
        const int r = e4 >> 4, l0 = (e4 & 15) * 4;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l0 >> 5);
        const int swi = s / TJ, swj = s - swi * TJ;
        float* out = p.partial +
            ((size_t)(sp * 9 + u * 3 + v) * p.Cx + ci0 + swi * 32 + row) * p.Cout +
            co0 + swj * 32 + (l0 & 31);
        *reinterpret_cast<f32x4*>(out) = sum;
      }
      __syncthreads();
    });
  }
}
constexpr size_t kWgradMergeLds4 = (size_t)3 * 4 * 1024 * sizeof(float);   // 4-wave workgroup
constexpr size_t kWgradMergeLds8 = (size_t)3 * 8 * 1024 * sizeof(float);   // 8-wave workgroup

// Workgroup = NW waves: 4 (one per SIMD, two workgroups per CU) or 8 (two per SIMD, ONE
// workgroup per CU).  The (CI_T x CO_T) tile has NSB = (CI_T/32)*(CO_T/32) 32x32 sub-blocks;
// wave w owns sub-block w % NSB for ALL nine taps (9 accumulator blocks = 144 VGPRs) and the
// pixel pairs q = w / NSB (mod NPP = NW / NSB) of each segment.
// The NPP parts of a sub-block are merged through LDS before they leave the CU
// (wgrad_epilogue): one slab per workgroup.
template <int CI_T, int CO_T, int S, int STRIDE, bool ACT = false, typename TX = float,
          typename TD = float, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int TI = CI_T / 32, TJ = CO_T / 32;
  constexpr int NSB = TI * TJ, NPP = NW / NSB;
  static_assert(NSB == 1 || NSB == 2 || NSB == 4, "tile must have 1, 2 or 4 sub-blocks");
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int NT = 64 * NW;
  constexpr int PW = (S - 1) * STRIDE + 3;
  constexpr int NP4 = 3 * PW * CI_T / 4;   // float4 slots of the patch
  constexpr int ND4 = S * CO_T / 4;        // float4 slots of the dy segment
  constexpr int PATCH = NP4 * 4;           // floats
  constexpr int STAGE = PATCH + ND4 * 4;
  // loader slots: the first NLP slots of every thread are patch slots, the next NLD dy slots
  // (slot kind is a compile-time property: no per-slot select, no branch)
  constexpr int NLP = (NP4 + NT - 1) / NT, NLD = (ND4 + NT - 1) / NT;
  constexpr int NQ = (S / 2) / NPP;        // pixel pairs per wave per segment
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int sb = wave % NSB, pp = wave / NSB;
  const int wi = sb / TJ, wj = sb - wi * TJ;

  int bid = blockIdx.x;
  const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
  const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
  const int sp = bid;
  const int ci0 = ci_t * CI_T, co0 = co_t * CO_T;
  const int g_begin = sp * p.segs_per_block;
  const int g_end = min(g_begin + p.segs_per_block, p.total_segs);

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

  // patch slot: (row-1, col-1, channel); invalid slots (idx >= NP4) get row = -2^20 so the
  // bounds test fails and the buffer load returns 0
  // (row, column) of a slot packed into one register: row in the high half (invalid: 0x4000)
  int p_rc[NLP];
  const int p_ch = ci0 + (tid % (CI_T / 4)) * 4;   // NT % (CI_T/4) == 0: the same for every k
#pragma unroll
  for (int k = 0; k < NLP; ++k) {
    const int idx = tid + NT * k;
    const int pix = idx / (CI_T / 4);
    const int prow = pix / PW, pcol = pix - prow * PW;
    p_rc[k] = ((idx < NP4 ? prow : 0x4000) << 16) | pcol;
  }
  int d_p[NLD], d_ch[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int d = tid + NT * k;
    const int dpix = d / (CO_T / 4), seg = d - dpix * (CO_T / 4);
    d_p[k] = d < ND4 ? dpix : (1 << 20);
    d_ch[k] = co0 + seg * 4;
  }

  f32x4 rp[NLP], rd[NLD];
  // ACT: coefficients of this thread's four channels (the channel group of a patch slot is
  // tid % (CI_T/4) for every k because CI_T/4 divides the block size), in-image flags
  static_assert(NT % (CI_T / 4) == 0, "one channel group per thread");
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  unsigned okm = 0;

  // Uniform cursor of the segment being LOADED: image ln, output row loy, first output column lx0.
  int ln, loy, lx0;
  auto set_cursor = [&](int g) {
    const int xs = g % p.segs_per_row;
    const int r = g / p.segs_per_row;
    loy = r % p.Ho;
    ln = r / p.Ho;
    lx0 = xs * S;
  };
  auto advance_cursor = [&]() {   // next segment in (n, row, column) order: a few scalar ops
    lx0 += S;
    if (lx0 >= p.segs_per_row * S) {
      lx0 = 0;
      if (++loy == p.Ho) { loy = 0; ++ln; }
    }
  };
  auto load_coef = [&]() {
    if (ACT) {
      const size_t o = (size_t)ln * p.Cx + p_ch;
      ca = *reinterpret_cast<const f32x4*>(p.alpha + o);
      cb = *reinterpret_cast<const f32x4*>(p.beta + o);
      okm = 0;
    }
  };
  // one loader slot of the cursor's segment: slots [0, NLP) = patch, [NLP, NLP+NLD) = dy
  auto issue_slot = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k < NLP) {
      const int iy = loy * STRIDE - 1 + (p_rc[k] >> 16), ix = lx0 * STRIDE - 1 + (p_rc[k] & 0xffff);
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      if (ACT) okm |= (ok ? 1u : 0u) << k;
      rp[k] = buf_ld4<TX>(rsx, (unsigned)(((ln * p.H + iy) * p.W + ix) * p.Cx + p_ch),
                          ok ? 0u : 0x80000000u);
    } else {
      constexpr int j = k - NLP;
      const int ox = lx0 + d_p[j];
      const bool ok = ox < p.Wo;
      rd[j] = buf_ld4<TD>(rsd, (unsigned)(((ln * p.Ho + loy) * p.Wo + ox) * p.Cout + d_ch[j]),
                          ok ? 0u : 0x80000000u);
    }
  };
  f32x16 acc[3][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;

  // Schedule of one stage (= one segment, NQ pixel-pair steps of 9 MFMAs per wave).  Everything
  // that is not an MFMA is placed by hand behind a step's MFMAs and fenced there, so it issues
  // in the shadow of the matrix pipe instead of at the loop top / bottom:
  //   steps [0, LQ):        the next segment's buffer loads (address arithmetic + issue)
  //   steps [NQ-HALF, NQ):  per loaded slot the activation (ACT) and its LDS write into the idle
  //                         stage - the writes drain at ~80 B/clk/CU, so bunched at the end of
  //                         the segment they would hold every wave at the barrier
  //   after the last step:  the barrier.
  constexpr int NLS = NLP + NLD;
  constexpr int LQ = NQ >= 4 ? NQ / 4 : 1;
  constexpr int LPER = (NLS + LQ - 1) / LQ;
  constexpr int HALF = NQ >= 2 ? NQ / 2 : 1;
  constexpr int APER = (NLS + HALF - 1) / HALF;
  auto put_slot = [&](auto kc, float* base) {   // activate (patch slots) and write one slot
    constexpr int k = decltype(kc)::value;
    if constexpr (k < NLP) {
      if (ACT) rp[k] = act4(rp[k], ca, cb, p.slope, (okm >> k) & 1u);
      if (NT * (k + 1) <= NP4 || tid + NT * k < NP4)
        *reinterpret_cast<f32x4*>(base + 4 * (tid + NT * k)) = rp[k];
    } else {
      constexpr int j = k - NLP;
      if (NT * (j + 1) <= ND4 || tid + NT * j < ND4)
        *reinterpret_cast<f32x4*>(base + PATCH + 4 * (tid + NT * j)) = rd[j];
    }
  };

  if (g_begin < g_end) {
    set_cursor(g_begin);
    load_coef();
    for_range<0, NLS>(issue_slot);
    for_range<0, NLS>([&](auto kc) { put_slot(kc, smem); });
    __syncthreads();
    for (int g = g_begin; g < g_end; ++g) {
      const int buf = (g - g_begin) & 1;
      // always stage (the last iteration re-stages the final segment into the idle buffer):
      // branch-free, so the compute part of a stage is one scheduling region
      if (g + 1 < g_end) advance_cursor();
      load_coef();
      const float* P = smem + buf * STAGE + wi * 32 + li;
      const float* D = smem + buf * STAGE + PATCH + wj * 32 + li;
      float* nxt_stage = smem + (buf ^ 1) * STAGE;
      float a[2][3][3], b[2];
      {
        const int xx = 2 * pp + lh;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int v = 0; v < 3; ++v) a[0][u][v] = P[(u * PW + xx * STRIDE + v) * CI_T];
        b[0] = D[xx * CO_T];
      }
      for_range<0, NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int cur = q & 1, nxt = cur ^ 1;
        if constexpr (q + 1 < NQ) {
          const int xx = 2 * (pp + NPP * (q + 1)) + lh;
#pragma unroll
          for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) a[nxt][u][v] = P[(u * PW + xx * STRIDE + v) * CI_T];
          b[nxt] = D[xx * CO_T];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int v = 0; v < 3; ++v)
            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][u][v], b[cur], acc[u][v], 0, 0, 0);
        if constexpr (q < LQ)
          for_range<q * LPER, ((q + 1) * LPER < NLS ? (q + 1) * LPER : NLS)>(issue_slot);
        if constexpr (q >= NQ - HALF) {
          constexpr int s0 = (q - (NQ - HALF)) * APER;
          for_range<s0, (s0 + APER < NLS ? s0 + APER : NLS)>(
              [&](auto kc) { put_slot(kc, nxt_stage); });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      __syncthreads();
    }
  }

  wgrad_epilogue<NSB, NPP, TJ, NT>(smem, p, sp, ci0, co0, sb, pp, [&](auto tc) -> const f32x16& {
    constexpr int t = decltype(tc)::value;
    return acc[t / 3][t % 3];
  });
}

// ---------------------------------------------------------------------------
// Weight gradient of conv3x3(upsample2x(a)) with respect to the up-sampled operand, at LOW
// resolution (misc.hip upsample2x_bwd_taps_kernel has the derivation):
//   dW[tap][ci][co] = sum over low-resolution pixels q of act(x)[q][ci] * D[q][tap*Cout + co].
// Nine GEMMs that share the A operand: the blocking of conv_wgrad_kernel with the roles of
// the operands swapped - per pixel pair ONE A fragment and nine B fragments (there: nine
// shifted A fragments and one B fragment) - and no halo: a segment is S consecutive pixels of
// the flattened [N*h*w] pixel list.  Same slabs / reductions as conv_wgrad_kernel.
// WgradParams: x = the low-resolution operand [Q][Cx], dy = D [Q][9*Cout], N*H*W = Q.
// ---------------------------------------------------------------------------
// NW = 8: one 8-wave workgroup per CU, pixel parts merged through LDS (see conv_wgrad_kernel).
template <int CI_T, int CO_T, int S, bool ACT, typename TX = float, typename TD = float,
          int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void conv_wgrad_taps_kernel(const WgradParams p) {
  constexpr int TI = CI_T / 32, TJ = CO_T / 32;
  constexpr int NSB = TI * TJ, NPP = NW / NSB;
  static_assert(NSB == 1 || NSB == 2 || NSB == 4, "tile must have 1, 2 or 4 sub-blocks");
  constexpr int NT = 64 * NW;
  constexpr int BW = 9 * CO_T;             // floats per pixel of the staged D tile
  constexpr int NA4 = S * CI_T / 4;        // float4 slots of the A segment
  constexpr int NB4 = S * BW / 4;          // float4 slots of the D segment
  constexpr int ATILE = NA4 * 4;
  constexpr int STAGE = ATILE + NB4 * 4;
  constexpr int NLA = (NA4 + NT - 1) / NT, NLB = (NB4 + NT - 1) / NT;
  static_assert(NA4 % NT == 0 || NLA == 1, "A slots");
  static_assert(NB4 % NT == 0, "D slots fill whole passes");
  constexpr int NQ = (S / 2) / NPP;        // pixel pairs per wave per segment
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int sb = wave % NSB, pp = wave / NSB;
  const int wi = sb / TJ, wj = sb - wi * TJ;

  // the tiles of one pixel range run on ONE XCD (they share its D / A segments through L2)
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
  const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
  const int sp = bid;
  const int ci0 = ci_t * CI_T, co0 = co_t * CO_T;
  const int g_begin = sp * p.segs_per_block;
  const int g_end = min(g_begin + p.segs_per_block, p.total_segs);
  const int HW = p.H * p.W;
  const int Q = p.N * HW;

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

  // A slot: pixel a_px[k] of the segment, channels a_ch..+3; D slot: pixel, float offset in BW
  // D slots carry ONE register: the element offset from the segment's first pixel.  A pixel past
  // the end of the tensor lands beyond the buffer descriptor's range and reads 0 by itself.
  int a_px[NLA];
  unsigned b_off[NLB];
  const int a_ch = ci0 + (tid % (CI_T / 4)) * 4;
#pragma unroll
  for (int k = 0; k < NLA; ++k) a_px[k] = (tid + NT * k) / (CI_T / 4);
#pragma unroll
  for (int k = 0; k < NLB; ++k) {
    const int idx = tid + NT * k;
    const int px = idx / (BW / 4);
    const int f = (idx - px * (BW / 4)) * 4;           // float offset within the 9*CO_T row
    const int t = f / CO_T;
    b_off[k] = (unsigned)(px * (9 * p.Cout) + t * p.Cout + co0 + (f - t * CO_T));
  }

  f32x4 ra[NLA], rb[NLB];
  f32x4 ca[NLA], cb[NLA];
  unsigned okm = 0;
  int lq0 = 0;   // first pixel of the segment being loaded
  auto issue_slot = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k < NLA) {
      const int q = lq0 + a_px[k];
      const bool ok = q < Q && (NA4 % NT == 0 || tid < NA4);
      ra[k] = buf_ld4<TX>(rsx, (unsigned)(q * p.Cx + a_ch), ok ? 0u : 0x80000000u);
      if (ACT) {
        const int n = ok ? q / HW : 0;
        ca[k] = *reinterpret_cast<const f32x4*>(p.alpha + (size_t)n * p.Cx + a_ch);
        cb[k] = *reinterpret_cast<const f32x4*>(p.beta + (size_t)n * p.Cx + a_ch);
        okm |= (ok ? 1u : 0u) << k;
      }
    } else {
      constexpr int j = k - NLA;
      rb[j] = buf_ld4<TD>(rsd, (unsigned)(lq0 * (9 * p.Cout)) + b_off[j], 0u);
    }
  };
  auto act_slot = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if (ACT) ra[k] = act4(ra[k], ca[k], cb[k], p.slope, (okm >> k) & 1u);
  };
  auto store_stage = [&](int buf) {
    float* base = smem + buf * STAGE;
#pragma unroll
    for (int k = 0; k < NLA; ++k)
      if (NA4 % NT == 0 || tid < NA4) *reinterpret_cast<f32x4*>(base + 4 * (tid + NT * k)) = ra[k];
#pragma unroll
    for (int k = 0; k < NLB; ++k)
      *reinterpret_cast<f32x4*>(base + ATILE + 4 * (tid + NT * k)) = rb[k];
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // stage schedule as in conv_wgrad_kernel: loads behind the first steps' MFMAs, the
  // activation behind the last steps', LDS writes + barrier at the end
  constexpr int NLS = NLA + NLB;
  constexpr int LQ = NQ >= 4 ? NQ / 2 : 1;
  constexpr int LPER = (NLS + LQ - 1) / LQ;

  if (g_begin < g_end) {
    lq0 = g_begin * S;
    for_range<0, NLS>(issue_slot);
    for_range<0, NLA>(act_slot);
    store_stage(0);
    __syncthreads();
    for (int g = g_begin; g < g_end; ++g) {
      const int buf = (g - g_begin) & 1;
      if (g + 1 < g_end) lq0 += S;   // the last iteration re-stages the final segment
      okm = 0;
      const float* A = smem + buf * STAGE + wi * 32 + li;
      const float* B = smem + buf * STAGE + ATILE + wj * 32 + li;
      float a[2], b[2][9];
      {
        const int xx = 2 * pp + lh;
        a[0] = A[xx * CI_T];
#pragma unroll
        for (int t = 0; t < 9; ++t) b[0][t] = B[xx * BW + t * CO_T];
      }
      for_range<0, NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int cur = q & 1, nxt = cur ^ 1;
        if constexpr (q + 1 < NQ) {
          const int xx = 2 * (pp + NPP * (q + 1)) + lh;
          a[nxt] = A[xx * CI_T];
#pragma unroll
          for (int t = 0; t < 9; ++t) b[nxt][t] = B[xx * BW + t * CO_T];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur], b[cur][t], acc[t], 0, 0, 0);
        if constexpr (q < LQ)
          for_range<q * LPER, ((q + 1) * LPER < NLS ? (q + 1) * LPER : NLS)>(issue_slot);
        if constexpr (ACT && q == NQ - 1) for_range<0, NLA>(act_slot);
        __builtin_amdgcn_sched_barrier(0);
      });
      store_stage(buf ^ 1);
      __syncthreads();
    }
  }

  wgrad_epilogue<NSB, NPP, TJ, NT>(smem, p, sp, ci0, co0, sb, pp, [&](auto tc) -> const f32x16& {
    return acc[decltype(tc)::value];
  });
}

// ---------------------------------------------------------------------------
// conv_wgrad_taps_kernel on the bf16 matrix cores (mixed-precision pipeline: x and D are bf16
// tensors): dW[tap][ci][co] = sum_q act(x)[q][ci] * D[q][tap*Cout + co] with
// v_mfma_f32_32x32x16_bf16.  Operands are staged as [32-channel sub-tile][pixel][32] (64-byte
// rows) and read with the transposing LDS read, as in conv_wgrad_bf16_kernel below: per
// 16-pixel k-group one A fragment and nine D fragments.  D is copied raw (16 bytes = 8 channels
// per slot: no bf16 -> fp32 -> bf16 round trip); x is activated in fp32 and rounded.
// 64 x 64 tiles: wave w owns sub-block w for all nine taps (one pixel part); 32 x 32 tiles
// (round 4: the 32-column layer of the last decoder stage, 266 us on the fp32 matrix cores):
// four pixel parts of a 64-pixel segment, merged in the epilogue.
// ---------------------------------------------------------------------------
template <int CI_T, int CO_T, int S, bool ACT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_taps_b16_kernel(const WgradParams p) {
  constexpr int TI = CI_T / 32, TJ = CO_T / 32;
  constexpr int NSB = TI * TJ, NPP = 4 / NSB;
  static_assert(NSB == 1 || NSB == 2 || NSB == 4, "tile must have 1, 2 or 4 sub-blocks");
  constexpr int NT = 256;
  constexpr int SUB = S * 32;                  // elements of one 32-channel sub-tile
  constexpr int ATILE = TI * SUB, DTILE = 9 * TJ * SUB, STAGE = ATILE + DTILE;
  constexpr int NA = S * CI_T / 4;             // 4-channel slots of the x segment
  constexpr int ND = S * 9 * CO_T / 8;         // 8-channel (16-byte) slots of the D segment
  constexpr int NLA = (NA + NT - 1) / NT, NLD = (ND + NT - 1) / NT;
  constexpr int NG = (S / 16) / NPP;           // 16-pixel k-groups per wave per segment
  static_assert((S / 16) % NPP == 0 && NG >= 1, "segment must split into whole k-groups");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sb = wave % NSB, pp = wave / NSB;
  const int wi = sb / TJ, wj = sb - wi * TJ;

  int bid = xcd_remap(blockIdx.x, gridDim.x);   // the tiles of one pixel range share D in one L2
  const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
  const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
  const int sp = bid;
  const int ci0 = ci_t * CI_T, co0 = co_t * CO_T;
  const int g_begin = sp * p.segs_per_block;
  const int g_end = min(g_begin + p.segs_per_block, p.total_segs);
  const int HW = p.H * p.W;
  const int Q = p.N * HW;

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

  // x slot k: pixel a_px of the segment, channels a_ch .. +3 -> LDS [sub][pixel][32]
  // D slot k: byte offset from the segment's first pixel (a pixel past the end of the tensor is
  //           beyond the descriptor's range and reads 0 by itself) -> LDS [tap][sub][pixel][32]
  int a_px[NLA], a_lds[NLA];
  const int a_seg = tid % (CI_T / 4);
  const int a_ch = ci0 + a_seg * 4;
#pragma unroll
  for (int k = 0; k < NLA; ++k) {
    a_px[k] = (tid + NT * k) / (CI_T / 4);
    a_lds[k] = (a_seg >> 3) * SUB + a_px[k] * 32 + (a_seg & 7) * 4;
  }
  unsigned d_off[NLD];
  int d_lds[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int idx = tid + NT * k;
    const int px = idx / (9 * CO_T / 8), rem = idx - px * (9 * CO_T / 8);
    const int t = rem / (CO_T / 8), c8 = rem - t * (CO_T / 8);
    d_off[k] = (unsigned)(px * (9 * p.Cout) + t * p.Cout + co0 + 8 * c8) * 2u;
    d_lds[k] = ATILE + (t * TJ + (c8 >> 2)) * SUB + px * 32 + (c8 & 3) * 8;
  }

  // two register sets (round 4): the segment after next is in flight while the next one waits
  // for its LDS stage (with one set a step - nine MFMAs - waited for the loads it had just issued:
  // matrix pipes 10 % busy)
  // (the 32 x 32 tile's D segment is nine 16-byte slots a thread: one set there)
  constexpr int NS = NLD <= 5 ? 2 : 1;
  f32x4 ra[NS][NLA], rdv[NS][NLD];
  f32x4 ca[NS][NLA], cb[NS][NLA];
  unsigned okm[NS] = {};
  int lq0 = 0;   // first pixel of the segment being loaded
  auto load_stage = [&](auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    okm[SET] = 0;
#pragma unroll
    for (int k = 0; k < NLA; ++k) {
      const int q = lq0 + a_px[k];
      const bool ok = q < Q && (NA % NT == 0 || tid + NT * k < NA);
      ra[SET][k] = buf_ld4<__bf16>(rsx, (unsigned)(q * p.Cx + a_ch), ok ? 0u : 0x80000000u);
      if (ACT) {
        const int n = ok ? q / HW : 0;
        ca[SET][k] = *reinterpret_cast<const f32x4*>(p.alpha + (size_t)n * p.Cx + a_ch);
        cb[SET][k] = *reinterpret_cast<const f32x4*>(p.beta + (size_t)n * p.Cx + a_ch);
        okm[SET] |= (ok ? 1u : 0u) << k;
      }
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      rdv[SET][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rsd, d_off[k] + (unsigned)(lq0 * (9 * p.Cout)) * 2u, 0, 0));
  };
  auto store_stage = [&](int buf, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    __bf16* base = smem_h + buf * STAGE;
#pragma unroll
    for (int k = 0; k < NLA; ++k) {
      f32x4 v = ra[SET][k];
      if (ACT) v = act4(v, ca[SET][k], cb[SET][k], p.slope, (okm[SET] >> k) & 1u);
      bf16x4 h;
      h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
      if (NA % NT == 0 || tid + NT * k < NA) *reinterpret_cast<bf16x4*>(base + a_lds[k]) = h;
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (NT * (k + 1) <= ND || tid + NT * k < ND)
        *reinterpret_cast<f32x4*>(base + d_lds[k]) = rdv[SET][k];
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (g_begin < g_end) {
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NS - 1>;
    lq0 = g_begin * S;
    load_stage(S0{});
    if constexpr (NS == 2) {
      if (g_begin + 1 < g_end) lq0 += S;   // (a range of one segment loads it twice: never stored)
      load_stage(S1{});
    }
    store_stage(0, S0{});
    __syncthreads();
    const int tg = lane >> 4, th = tg >> 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const int tcol = 16 * (tg & 1) + 4 * tp;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    auto frag = [&](const __bf16* q) {   // rows r0..r0+3 and r0+4..r0+7 of this lane's group
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * 32));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // step g on stage (g - g_begin) & 1; LOADSET receives segment g + 2 (past the end: the last
    // one again - every path issues the same loads), STORESET holds segment g + 1
    auto seg_step = [&](int g, auto loadc, auto storec) __attribute__((always_inline)) {
      const int buf = (g - g_begin) & 1;
      if (g + NS < g_end) lq0 += S;
      load_stage(loadc);
      const __bf16* A = smem_h + buf * STAGE + wi * SUB;
      const __bf16* D = smem_h + buf * STAGE + ATILE + wj * SUB;
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        const int r0 = 16 * (pp + NPP * gq) + 8 * th + tq;
        const bf16x8 a = frag(A + r0 * 32 + tcol);
#pragma unroll
        for (int t = 0; t < 9; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              a, frag(D + t * TJ * SUB + r0 * 32 + tcol), acc[t], 0, 0, 0);
      }
      store_stage(buf ^ 1, storec);
      __syncthreads();
    };
    if constexpr (NS == 2) {
      for (int g = g_begin; g < g_end; g += 2) {
        seg_step(g, S0{}, S1{});
        if (g + 1 < g_end) seg_step(g + 1, S1{}, S0{});
      }
    } else {
      for (int g = g_begin; g < g_end; ++g) seg_step(g, S0{}, S0{});
    }
  }

  wgrad_epilogue<NSB, NPP, TJ, NT>(reinterpret_cast<float*>(smem_h), p, sp, ci0, co0, sb, pp,
                                   [&](auto tc) -> const f32x16& {
    return acc[decltype(tc)::value];
  });
}

// ---------------------------------------------------------------------------
// bf16 mixed-precision weight gradient (stride 1): same blocking, slabs and loader as
// conv_wgrad_kernel; operands are rounded to bf16 while staged into LDS as
// [32-channel sub-tile][pixel][32] (64-B rows) and read as MFMA fragments with the gfx950
// transposing LDS read ds_read_b64_tr_b16 (the reduction axis = pixels is the ROW axis of
// both tiles): per 16-pixel k-group 2 reads per operand, v_mfma_f32_32x32x16_bf16, fp32 sums.
// Lane l: g = l>>4, h = g>>1, q = (l&15)>>2, p = l&3 supplies row 8h + 4*half + q, columns
// 16*(g&1) + 4p and receives column l&31 of those four rows = operand element k = 8h+4*half+q'.
// ---------------------------------------------------------------------------
//
// NPL = 3 is the split-bf16 ("bf16x3") form: every operand is staged as three bf16 planes
// (common.h split3) and each tap accumulates the six products of weight >= 2^-16, which
// reproduces the fp32 product to one fp32 rounding (see conv_igemm_split_kernel).
// SB = single LDS stage (two barriers per segment) where the double-buffered planes would leave
// one workgroup per CU (32x32 tile, 64-pixel segments, three planes: 50 KB instead of 100 KB).
// STRIDE = 2 (mixed-precision pipeline only): the patch holds (S-1)*2+3 input columns per row
// and the A fragment rows are two patch pixels apart (the transposing read takes a row address
// per lane, so the stride costs nothing).
template <int CI_T, int CO_T, int S, int NPL, bool SB = false, typename TX = float,
          typename TD = float, bool ACT = false, int STRIDE = 1>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const WgradParams p) {
  constexpr int TI = CI_T / 32, TJ = CO_T / 32;
  constexpr int NSB = TI * TJ, NPP = 4 / NSB;
  static_assert(NSB == 1 || NSB == 2 || NSB == 4, "tile must have 1, 2 or 4 sub-blocks");
  constexpr int NT = 256;
  constexpr int PW = (S - 1) * STRIDE + 3;
  constexpr int NP4 = 3 * PW * CI_T / 4;   // float4 slots of the patch
  constexpr int ND4 = S * CO_T / 4;        // float4 slots of the dy segment
  constexpr int PATCH = NP4 * 4;           // elements
  constexpr int PLANE = PATCH + ND4 * 4;   // one bf16 plane of a stage
  constexpr int STAGE = PLANE * NPL;
  // loader slots: the first NLP slots of every thread are patch slots, the next NLD dy slots
  // (slot kind is a compile-time property: no per-slot select, no branch)
  constexpr int NLP = (NP4 + NT - 1) / NT, NLD = (ND4 + NT - 1) / NT;
  constexpr int NG = (S / 16) / NPP;       // 16-pixel k-groups per wave per segment
  static_assert((S / 16) % NPP == 0 && NG >= 1, "segment must split into whole k-groups");
  static_assert(NPL == 1 || NPL == 3, "one bf16 plane or the three of the split form");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];
  constexpr int PSUB = 3 * PW * 32, DSUB = S * 32;   // elements per 32-channel sub-tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sb = wave % NSB, pp = wave / NSB;
  const int wi = sb / TJ, wj = sb - wi * TJ;

  int bid = blockIdx.x;
  const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
  const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
  const int sp = bid;
  const int ci0 = ci_t * CI_T, co0 = co_t * CO_T;
  const int g_begin = sp * p.segs_per_block;
  const int g_end = min(g_begin + p.segs_per_block, p.total_segs);

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

  // patch slot: (row-1, col-1, channel); invalid slots (idx >= NP4) get row = -2^20 so the
  // bounds test fails and the buffer load returns 0
  int p_r[NLP], p_c[NLP], p_ch[NLP];
#pragma unroll
  for (int k = 0; k < NLP; ++k) {
    const int idx = tid + NT * k;
    const int pix = idx / (CI_T / 4), seg = idx - pix * (CI_T / 4);
    const int prow = pix / PW, pcol = pix - prow * PW;
    p_r[k] = idx < NP4 ? prow - 1 : -(1 << 20);
    p_c[k] = pcol - 1;
    p_ch[k] = ci0 + seg * 4;
  }
  int d_p[NLD], d_ch[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int d = tid + NT * k;
    const int dpix = d / (CO_T / 4), seg = d - dpix * (CO_T / 4);
    d_p[k] = d < ND4 ? dpix : (1 << 20);
    d_ch[k] = co0 + seg * 4;
  }

  f32x4 rp[NLP], rd[NLD];
  // ACT: coefficients of this thread's four channels (one channel group per thread), flags
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  unsigned okm = 0;
  auto load_stage = [&](int g) {
    const int xs = g % p.segs_per_row;
    const int r = g / p.segs_per_row;
    const int oy = r % p.Ho;
    const int n = r / p.Ho;
    const int x0 = xs * S;
    if (ACT) {
      const size_t o = (size_t)n * p.Cx + p_ch[0];
      ca = *reinterpret_cast<const f32x4*>(p.alpha + o);
      cb = *reinterpret_cast<const f32x4*>(p.beta + o);
      okm = 0;
    }
#pragma unroll
    for (int k = 0; k < NLP; ++k) {
      const int iy = oy * STRIDE + p_r[k], ix = x0 * STRIDE + p_c[k];
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      if (ACT) okm |= (ok ? 1u : 0u) << k;
      rp[k] = buf_ld4<TX>(rsx, (unsigned)(((n * p.H + iy) * p.W + ix) * p.Cx + p_ch[k]),
                          ok ? 0u : 0x80000000u);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int ox = x0 + d_p[k];
      const bool ok = ox < p.Wo;
      rd[k] = buf_ld4<TD>(rsd, (unsigned)(((n * p.Ho + oy) * p.Wo + ox) * p.Cout + d_ch[k]),
                          ok ? 0u : 0x80000000u);
    }
  };
  auto to_bf16 = [](const f32x4 v) {
    bf16x4 h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    return h;
  };
  // LDS element offset of patch slot idx: [plane][sub][pixel][32]; seg -> channel 4*seg
  auto put = [&](__bf16* dst, const f32x4 v) {
    if (NPL == 1) {
      *reinterpret_cast<bf16x4*>(dst) = to_bf16(v);
    } else {
      bf16x4 h, m, l;
      split3(v, h, m, l);
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + PLANE) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * PLANE) = l;
    }
  };
  auto store_stage = [&](int buf) {
    __bf16* base = smem_h + buf * STAGE;
#pragma unroll
    for (int k = 0; k < NLP; ++k) {
      const int idx = tid + NT * k;
      const int pix = idx / (CI_T / 4), seg = idx - pix * (CI_T / 4);
      if (ACT) rp[k] = act4(rp[k], ca, cb, p.slope, (okm >> k) & 1u);
      if (NT * (k + 1) <= NP4 || idx < NP4)
        put(base + (seg >> 3) * PSUB + pix * 32 + (seg & 7) * 4, rp[k]);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int d = tid + NT * k;
      const int dpix = d / (CO_T / 4), seg = d - dpix * (CO_T / 4);
      if (NT * (k + 1) <= ND4 || d < ND4)
        put(base + PATCH + (seg >> 3) * DSUB + dpix * 32 + (seg & 7) * 4, rd[k]);
    }
  };

  f32x16 acc[3][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;

  if (g_begin < g_end) {
    load_stage(g_begin);
    store_stage(0);
    __syncthreads();
    for (int g = g_begin; g < g_end; ++g) {
      const int buf = SB ? 0 : (g - g_begin) & 1;
      // always stage (the last iteration re-stages the final segment into the idle buffer):
      // branch-free, so the compute part of a stage is one scheduling region
      load_stage(min(g + 1, g_end - 1));
      const __bf16* P = smem_h + buf * STAGE + wi * PSUB;
      const __bf16* D = smem_h + buf * STAGE + PATCH + wj * DSUB;
      const int tg = lane >> 4, th = tg >> 1, tq = (lane & 15) >> 2, tp = lane & 3;
      const int tcol = 16 * (tg & 1) + 4 * tp;
      typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        const int xx0 = 16 * (pp + NPP * gq);          // first pixel of this k-group
        const int r0 = xx0 + 8 * th + tq;              // this lane's row for half 0 (+4 for half 1)
        // rows r0..r0+3 and r0+4..r0+7 of this lane's group (`step` elements between rows + 4)
        auto frag = [&](const __bf16* q, int step = 4 * 32) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + step));
          return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        bf16x8 b[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[pl] = frag(D + (pl < NPL ? pl : 0) * PLANE + r0 * 32 + tcol);
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            const __bf16* pa = P + (u * PW + STRIDE * r0 + v) * 32 + tcol;
            constexpr int AST = 4 * STRIDE * 32;
            if (NPL == 1) {
              acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(pa, AST), b[0], acc[u][v], 0, 0, 0);
            } else {
              const bf16x8 a0 = frag(pa, AST), a1 = frag(pa + PLANE, AST),
                           a2 = frag(pa + 2 * PLANE, AST);
              f32x16 c = acc[u][v];
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[0], c, 0, 0, 0);
              acc[u][v] = c;
            }
          }
      }
      if (SB) __syncthreads();   // every wave has read the only stage
      store_stage(SB ? 0 : buf ^ 1);
      __syncthreads();
    }
  }

  wgrad_epilogue<NSB, NPP, TJ, NT>(reinterpret_cast<float*>(smem_h), p, sp, ci0, co0, sb, pp,
                                   [&](auto tc) -> const f32x16& {
    constexpr int t = decltype(tc)::value;
    return acc[t / 3][t % 3];
  });
}


// ---------------------------------------------------------------------------
// Mixed-precision weight gradient, stride 1, bf16 tensors: the ROW-RING form (round 4).
//
// conv_wgrad_bf16_kernel stages, per segment of S pixels, the three input rows under it and
// prefetches ONE segment ahead: a segment is 18 MFMAs per wave (0.25 us) while a load takes
// 1-2 us to come back under load, so every segment waits for its operands (PMC r04: matrix
// cores busy 16 %, waves waiting 56 % of their cycles), and every input row is fetched and
// activated three times.  Here a workgroup walks DOWN a column strip (S output pixels wide):
//   * the input rows live in a four-slot ring in LDS ([sub-tile][slot][S + 2 pixels][32] bf16,
//     slot = row & 3): a step reads rows oy-1 .. oy+1 and fills the slot of row oy+2 - each row
//     is loaded, activated and rounded ONCE (a third of the loads and of the activation VALU);
//   * the operands of a step are one input row and one dy row, 16 bytes per lane and load:
//     12 registers a step, so the loads of DEPTH = 2 steps are in flight (the old form held
//     36 registers for one step; DEPTH = 4 measured no better - with each row fetched once the
//     kernel runs at ~75 % of what its HBM bytes, slabs included, allow);
//   * the activation coefficients depend on (image, channel) only: read once per workgroup.
// Blocking, MFMA fragments (ds_read_b64_tr_b16), slabs and the merge epilogue are those of
// conv_wgrad_bf16_kernel.  g in [0, total_segs) enumerates (image, strip, output row) with the
// row fastest; a workgroup's range lies inside one strip (the plan makes segs_per_block divide
// Ho and a multiple of DEPTH).
// ---------------------------------------------------------------------------
// RG = 2: eight waves, TWO output rows a step - waves 0-3 take the even row of the pair, waves
// 4-7 the odd one (each half loads "its" input and dy row and runs the four-wave code on it; the
// ring is shared: 8 slots); the two halves are merged with the pixel parts in the epilogue, so
// a workgroup still leaves ONE slab: 256 slabs per layer instead of 512 - the slabs are a third
// of the kernel's HBM traffic at 64 x 64 tiles and all of the reduction kernel's.
// STRIDE = 2: an output row needs input rows 2 oy - 1 .. 2 oy + 1, of which 2 oy and 2 oy + 1
// are new: a bundle is TWO input rows (S * 2 + 1 pixels wide) and one dy row, the ring has
// 8 RG slots, and the A fragment rows are two patch pixels apart (the transposing read takes a
// row address per lane).
template <int CI_T, int CO_T, int S, bool ACT, int DEPTH, int RG, int STRIDE = 1>
__global__ __launch_bounds__(256 * RG, RG == 2 ? 1 : 2) void conv_wgrad_b16_ring_kernel(const WgradParams p) {
  constexpr int TI = CI_T / 32, TJ = CO_T / 32;
  constexpr int NSB = TI * TJ, NPP = 4 / NSB, NT = 256;   // (NT: one row group)
  constexpr int PW = (S - 1) * STRIDE + 3;
  constexpr int NR = STRIDE;                               // new input rows per output row
  constexpr int NG = (S / 16) / NPP;
  static_assert((S / 16) % NPP == 0 && NG >= 1, "segment must split into whole k-groups");
  constexpr int XSEG = CI_T / 8, DSEG = CO_T / 8;          // 16-byte slots per pixel
  constexpr int XS8 = NR * PW * XSEG, DS8 = S * DSEG;      // slots of a bundle's input rows / of a dy row
  constexpr int NLX = (XS8 + NT - 1) / NT, NLD = (DS8 + NT - 1) / NT;
  static_assert(NT % XSEG == 0 && NT % DSEG == 0, "a thread keeps its channel group");
  constexpr int RR = 4 * RG * STRIDE;                      // ring slots (rows)
  constexpr int RSUB = RR * PW * 32;                       // ring elements of a 32-channel sub-tile
  constexpr int RING = TI * RSUB;
  constexpr int DSUB = S * 32, DBUF = TJ * DSUB;
  static_assert((size_t)(RING + 2 * RG * DBUF) * 2 <= (RG == 2 ? kWgradMergeLds8 : kWgradMergeLds4),
                "ring + dy stages fit the merge space");
  static_assert(STRIDE == 1 || STRIDE == 2, "stride 1 or 2");
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];
  __bf16* Dst = smem_h + RING;
  typedef int i32x4r __attribute__((ext_vector_type(4)));

  const int lg = threadIdx.x >> 8;                         // row group (0 when RG == 1)
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int sb = wave % NSB, pp = wave / NSB;
  const int wi = sb / TJ, wj = sb - wi * TJ;

  int bid = blockIdx.x;
  const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
  const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
  const int sp = bid;
  const int ci0 = ci_t * CI_T, co0 = co_t * CO_T;
  const int g_begin = sp * p.segs_per_block;
  const int strip = g_begin / p.Ho;
  const int oy0 = g_begin - strip * p.Ho;
  const int rows = p.segs_per_block;                       // (divides Ho: the strip is not left)
  const int n = strip / p.segs_per_row;
  const int x0 = (strip - n * p.segs_per_row) * S;         // first OUTPUT column of the strip

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

  // input-row slot k: pixel (tid + 256 k) / XSEG of the S + 2 under the strip, channels
  // ci0 + 8 seg .. + 7; byte offset of the slot inside image row 0 and its LDS place in a ring row
  const int xseg = tid % XSEG, dseg = tid % DSEG;
  int x_off[NLX], x_lds[NLX];
  unsigned x_okc = 0;                                      // bit k: the slot's column is inside the image
  unsigned x_row1 = 0;                                     // bit k: the slot belongs to the bundle's second row
#pragma unroll
  for (int k = 0; k < NLX; ++k) {
    const int slot = tid + NT * k;
    const int r2 = (slot / XSEG) / PW;                     // 0 (.. NR - 1)
    const int pix = slot / XSEG - r2 * PW;
    const int ix = x0 * STRIDE - 1 + pix;
    const bool ok = slot < XS8 && (unsigned)ix < (unsigned)p.W;
    x_okc |= (ok ? 1u : 0u) << k;
    x_row1 |= (r2 ? 1u : 0u) << k;
    x_off[k] = ((r2 * p.W + ix) * p.Cx + ci0 + xseg * 8) * 2;
    x_lds[k] = (xseg >> 2) * RSUB + (r2 * PW + pix) * 32 + (xseg & 3) * 8;
  }
  int d_off[NLD], d_lds[NLD];
  unsigned d_okc = 0;
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int slot = tid + NT * k;
    const int dpix = slot / DSEG;
    const bool ok = slot < DS8 && x0 + dpix < p.Wo;
    d_okc |= (ok ? 1u : 0u) << k;
    d_off[k] = ((x0 + dpix) * p.Cout + co0 + dseg * 8) * 2;
    d_lds[k] = (dseg >> 2) * DSUB + dpix * 32 + (dseg & 3) * 8;
  }
  f32x4 ca[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}};
  f32x4 cb[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (ACT) {
    const size_t o = (size_t)n * p.Cx + ci0 + xseg * 8;
    ca[0] = *reinterpret_cast<const f32x4*>(p.alpha + o);
    ca[1] = *reinterpret_cast<const f32x4*>(p.alpha + o + 4);
    cb[0] = *reinterpret_cast<const f32x4*>(p.beta + o);
    cb[1] = *reinterpret_cast<const f32x4*>(p.beta + o + 4);
  }

  i32x4r rx[DEPTH][NLX], rd[DEPTH][NLD];
  unsigned rok[DEPTH];                                     // uniform: bit r: row iy + r of the set is inside the image
  // bundle t of the walk: the new bottom row(s) of step t - rows iy .. iy + NR - 1 - and dy row oy0 + t
  auto load_x = [&](auto setc, int iy) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    const unsigned ok0 = (unsigned)iy < (unsigned)p.H ? 1u : 0u;
    const unsigned ok1 = (NR == 2 && (unsigned)(iy + 1) < (unsigned)p.H) ? 2u : 0u;
    rok[SET] = ok0 | ok1;
    const int rowbase = (n * p.H + iy) * p.W * p.Cx * 2;
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      const bool rowok = (rok[SET] >> ((x_row1 >> k) & 1u)) & 1u;
      const bool ok = rowok && ((x_okc >> k) & 1u);
      rx[SET][k] = __builtin_amdgcn_raw_buffer_load_b128(
          rsx, ok ? (unsigned)(rowbase + x_off[k]) : 0x80000000u, 0, 0);
    }
  };
  auto load_d = [&](auto setc, int oy) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    const bool rowok = oy < p.Ho;
    const int rowbase = (n * p.Ho + oy) * p.Wo * p.Cout * 2;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const bool ok = rowok && ((d_okc >> k) & 1u);
      rd[SET][k] = __builtin_amdgcn_raw_buffer_load_b128(
          rsd, ok ? (unsigned)(rowbase + d_off[k]) : 0x80000000u, 0, 0);
    }
  };
  auto store_x = [&](auto setc, int iy) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    __bf16* row = smem_h + (iy & (RR - 1)) * (PW * 32);
#pragma unroll
    for (int k = 0; k < NLX; ++k) {
      if (NT * (k + 1) <= XS8 || tid + NT * k < XS8) {
        if (ACT) {
          const bool ok = ((rok[SET] >> ((x_row1 >> k) & 1u)) & 1u) && ((x_okc >> k) & 1u);   // zero padding stays zero
          const i32x4r q = rx[SET][k];
          const f32x4 lo = act4(widen16(i32x2r{q[0], q[1]}), ca[0], cb[0], p.slope, ok);
          const f32x4 hi = act4(widen16(i32x2r{q[2], q[3]}), ca[1], cb[1], p.slope, ok);
          bf16x8 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) { h[e] = (__bf16)lo[e]; h[4 + e] = (__bf16)hi[e]; }
          *reinterpret_cast<bf16x8*>(row + x_lds[k]) = h;
        } else {
          *reinterpret_cast<i32x4r*>(row + x_lds[k]) = rx[SET][k];
        }
      }
    }
  };
  auto store_d = [&](auto setc, int buf) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    __bf16* base = Dst + buf * DBUF;
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      if (NT * (k + 1) <= DS8 || tid + NT * k < DS8)
        *reinterpret_cast<i32x4r*>(base + d_lds[k]) = rd[SET][k];
  };

  f32x16 acc[3][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;

  using C0 = std::integral_constant<int, 0>;
  // bundle t of a row group: input row oy0 + RG t + 1 + lg (the new bottom rows of step t) and
  // dy row oy0 + RG t + lg.  prologue: rows oy0 - 1 and oy0, then bundle 0; bundles 1 .. DEPTH-1
  // stay in flight
  // (first new input row of output row oy: oy + 1 at stride 1, 2 oy at stride 2)
  auto new_row = [](int oy) { return STRIDE == 1 ? oy + 1 : 2 * oy; };
  if constexpr (STRIDE == 1) {
    for (int r = lg; r < 2; r += RG) {     // uniform per row group
      load_x(C0{}, oy0 - 1 + r);
      store_x(C0{}, oy0 - 1 + r);
    }
  } else {                                 // row 2 oy0 - 1 = the second row of bundle "-1"
    load_x(C0{}, new_row(oy0 - RG + lg));  // (the group of output row oy0 - 1 brings it; the
    store_x(C0{}, new_row(oy0 - RG + lg)); //  other rows of that round land in slots nothing reads)
  }
  load_x(C0{}, new_row(oy0 + lg));
  load_d(C0{}, oy0 + lg);
  store_x(C0{}, new_row(oy0 + lg));
  store_d(C0{}, lg);
  for_range<1, DEPTH>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    load_x(jc, new_row(oy0 + RG * j + lg));
    load_d(jc, oy0 + RG * j + lg);
  });
  __syncthreads();

  const int tg = lane >> 4, th = tg >> 1, tq = (lane & 15) >> 2, tp = lane & 3;
  const int tcol = 16 * (tg & 1) + 4 * tp;
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  auto frag = [&](const __bf16* q) __attribute__((always_inline)) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * 32));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto frag_a = [&](const __bf16* q) __attribute__((always_inline)) {   // rows STRIDE pixels apart
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * STRIDE * 32));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  const int steps = rows / RG;
  for (int t0 = 0; t0 < steps; t0 += DEPTH) {
    for_range<0, DEPTH>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int JN = (j + 1) % DEPTH;
      const int t = t0 + j;
      const int oy = oy0 + RG * t + lg;                    // this row group's output row
      // set j is free (bundle t went to LDS at the end of step t - 1): bundle t + DEPTH
      // (past the workgroup's range the loads are out of range on purpose: they return zeros
      // that are stored into slots nothing reads - the waits stay countable)
      const bool more = t + DEPTH < steps;
      load_x(jc, more ? new_row(oy + RG * DEPTH) : -4);
      load_d(jc, more ? oy + RG * DEPTH : p.Ho);
      const __bf16* P = smem_h + wi * RSUB;
      const __bf16* D = Dst + ((t & 1) * RG + lg) * DBUF + wj * DSUB;
#pragma unroll
      for (int gq = 0; gq < NG; ++gq) {
        const int r0 = 16 * (pp + NPP * gq) + 8 * th + tq;
        const bf16x8 b = frag(D + r0 * 32 + tcol);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const __bf16* pr = P + (((STRIDE * oy - 1 + u) & (RR - 1)) * PW + STRIDE * r0) * 32 + tcol;
#pragma unroll
          for (int v = 0; v < 3; ++v)
            acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_a(pr + v * 32), b, acc[u][v], 0, 0, 0);
        }
      }
      // bundle t + 1 (set JN, loaded DEPTH - 1 steps ago): rows into the slots no wave reads
      store_x(std::integral_constant<int, JN>{}, new_row(oy + RG));
      store_d(std::integral_constant<int, JN>{}, ((t + 1) & 1) * RG + lg);
      __syncthreads();
    });
  }

  wgrad_epilogue<NSB, NPP * RG, TJ, NT * RG>(reinterpret_cast<float*>(smem_h), p, sp, ci0, co0, sb,
                                             lg * NPP + pp,
                                   [&](auto tc) -> const f32x16& {
    constexpr int t = decltype(tc)::value;
    return acc[t / 3][t % 3];
  });
}

// ---------------------------------------------------------------------------
// Reduction of the per-workgroup slabs into the OIHW gradient.  Every weight-gradient kernel
// leaves `nslab` partial gradients [tap][ci][co]; they are summed in a FIXED order (no float
// atomics: run-to-run determinism) in up to three stages - chunks of 16 slabs while more than 16
// remain (SLAB jobs), then the final sum + scatter (the other kinds).
//
// All of it runs through ONE kernel that takes a TABLE of jobs in its kernel arguments
// (wgrad_reduce_batched_kernel).  Called one job at a time it is what rounds 1-3 launched per
// weight gradient (2-3 launches of 4-15 us each, 44 per train step); with deferral on
// (unet_wgrad_defer_begin) the entry points only QUEUE their jobs and unet_wgrad_defer_flush
// launches the queued stages of every layer together: 2-3 launches per flush, the jobs of all
// layers running side by side.  The arithmetic and its order are those of the per-call form:
// bit-identical results.
// ---------------------------------------------------------------------------
constexpr int kSlabChunk = 16;
enum ReduceKind { RK_SLAB = 0, RK_CENTER, RK_GENERIC, RK_WIDE8, RK_WIDE16, RK_TAP, RK_STEM };
struct ReduceJob {
  const float* src;
  float* dst;
  int nslab;                          // slabs to sum (SLAB: of the whole input)
  int Cx, Cout, ci_off, Cin_total;    // SLAB: Cx = E / 4 (float4 elements of a slab)
  int kind;
  int gx, gy;                         // grid of the job: gx * gy * gz blocks
  int block_begin;                    // first block of the job in the batched launch
};
constexpr int kReduceJobs = 56;       // 56 x 56 B + 8 = 3144 B of kernel arguments (limit 4 KB)
struct ReduceTable {
  int n, pad;
  ReduceJob j[kReduceJobs];
};

// out[c][e] = sum of slabs [16c, 16c+16) of in[.][e] (fixed order)
__device__ __forceinline__ void reduce_slab_body(const ReduceJob& J, int bx, int c) {
  const long long E4 = J.Cx;
  const long long e = (long long)bx * 256 + threadIdx.x;
  if (e >= E4) return;
  const int s0 = c * kSlabChunk;
  const int s1 = min(s0 + kSlabChunk, J.nslab);
  const f32x4* src = reinterpret_cast<const f32x4*>(J.src) + e;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = s0; k < s1; ++k) acc += src[(size_t)k * E4];
  reinterpret_cast<f32x4*>(J.dst)[(size_t)c * E4 + e] = acc;
}

// 1x1 weight gradient = centre tap of the 3x3 slabs: dw[co][ci_off+ci] = sum_s partial[s][4][ci][co]
__device__ __forceinline__ void reduce_center_body(const ReduceJob& J, float* sm, int bx, int by) {
  float (*tile)[33] = reinterpret_cast<float (*)[33]>(sm);   // [32][33]
  const int Cx = J.Cx, Cout = J.Cout;
  const int co0 = bx * 32, ci0 = by * 32;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const size_t slab = (size_t)9 * Cx * Cout;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ci = r + 8 * k;
    const float* src = J.src + ((size_t)4 * Cx + ci0 + ci) * Cout + co0 + c;
    float s = 0.f;
    for (int q = 0; q < J.nslab; ++q) s += src[(size_t)q * slab];
    tile[ci][c] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int co = r + 8 * k;
    J.dst[(size_t)(co0 + co) * J.Cin_total + J.ci_off + ci0 + c] = tile[c][co];
  }
}

// dw_oihw[co][ci_off+ci][tap] = sum_s partial[s][tap][ci][co]
// tile: 32 co x 8 ci per block, all 9 taps; LDS transpose so both sides coalesce-ish
__device__ __forceinline__ void reduce_generic_body(const ReduceJob& J, float* sm, int bx, int by) {
  float (*tile)[8][33] = reinterpret_cast<float (*)[8][33]>(sm);   // [9][8][33]
  const int Cx = J.Cx, Cout = J.Cout;
  const int co0 = bx * 32, ci0 = by * 8;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;  // r in 0..7
  const size_t slab = (size_t)9 * Cx * Cout;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float* src = J.src + ((size_t)t * Cx + ci0 + r) * Cout + co0 + c;
    float s = 0.f;
    for (int k = 0; k < J.nslab; ++k) s += src[(size_t)k * slab];
    tile[t][r][c] = s;
  }
  __syncthreads();
  // write: for each co (32) a run of 8 ci * 9 taps = 72 contiguous floats
  for (int i = threadIdx.x; i < 32 * 72; i += 256) {
    const int co = i / 72, rem = i - co * 72;
    const int ci = rem / 9, t = rem - ci * 9;
    J.dst[((size_t)(co0 + co) * J.Cin_total + J.ci_off + ci0 + ci) * 9 + t] = tile[t][ci][co];
  }
}

// Wide form for the large gradients (Cx * Cout > 128 * 128): a block sums 64 co x CI ci x 9 taps
// with 16-byte loads along co (256-byte row segments instead of 128) and writes runs of CI * 9
// contiguous floats per output channel.
template <int CI>
__device__ __forceinline__ void reduce_wide_body(const ReduceJob& J, float* sm, int bx, int by) {
  float (*tile)[65] = reinterpret_cast<float (*)[65]>(sm);   // [9 * CI][65]
  const int Cx = J.Cx, Cout = J.Cout;
  const int co0 = bx * 64, ci0 = by * CI;
  const int c4 = threadIdx.x & 15, r = threadIdx.x >> 4;   // 16 rows x 16 float4 per pass
  const size_t slab = (size_t)9 * Cx * Cout;
  for (int rr = r; rr < 9 * CI; rr += 16) {
    const int t = rr / CI, ci = rr - t * CI;
    const float* src = J.src + ((size_t)t * Cx + ci0 + ci) * Cout + co0 + c4 * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(src);
    for (int k = 1; k < J.nslab; ++k) s += *reinterpret_cast<const f32x4*>(src + (size_t)k * slab);
    tile[rr][c4 * 4 + 0] = s[0]; tile[rr][c4 * 4 + 1] = s[1];
    tile[rr][c4 * 4 + 2] = s[2]; tile[rr][c4 * 4 + 3] = s[3];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * CI * 9; i += 256) {
    const int co = i / (CI * 9), rem = i - co * (CI * 9);
    const int ci = rem / 9, t = rem - ci * 9;
    J.dst[((size_t)(co0 + co) * J.Cin_total + J.ci_off + ci0 + ci) * 9 + t] = tile[t * CI + ci][co];
  }
}

// The same with one tap per block (gz = 9): 9x the workgroups for the layers whose gradient is
// small (32 / 64 channels: 4 / 16 blocks of the forms above, each thread walking split x 9
// dependent slab rows, would run for tens of microseconds on a near-idle chip).
__device__ __forceinline__ void reduce_tap_body(const ReduceJob& J, float* sm, int bx, int by, int t) {
  float (*tile)[33] = reinterpret_cast<float (*)[33]>(sm);   // [8][33]
  const int Cx = J.Cx, Cout = J.Cout;
  const int co0 = bx * 32, ci0 = by * 8;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;  // r in 0..7
  const size_t slab = (size_t)9 * Cx * Cout;
  const float* src = J.src + ((size_t)t * Cx + ci0 + r) * Cout + co0 + c;
  float s = 0.f;
  for (int k = 0; k < J.nslab; ++k) s += src[(size_t)k * slab];
  tile[r][c] = s;
  __syncthreads();
  // thread (co = tid >> 3, ci = tid & 7)
  const int co = threadIdx.x >> 3, ci = threadIdx.x & 7;
  J.dst[((size_t)(co0 + co) * J.Cin_total + J.ci_off + ci0 + ci) * 9 + t] = tile[ci][co];
}

// stem reduce: dw_oihw[co][ci][tap] (Cin_total = 3) = sum_b partial[b][tap*3+ci][co]
__device__ __forceinline__ void reduce_stem_body(const ReduceJob& J, int bx) {
  const int Cout = J.Cout;
  const int i = bx * 256 + threadIdx.x;
  if (i >= 27 * Cout) return;
  const int k = i / Cout, co = i - k * Cout;
  float s = 0.f;
  for (int b = 0; b < J.nslab; ++b) s += J.src[((size_t)b * 27 + k) * Cout + co];
  const int t = k / 3, ci = k - t * 3;
  J.dst[((size_t)co * 3 + ci) * 9 + t] = s;
}

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const ReduceTable tab) {
  __shared__ float sm[9 * 16 * 65];
  // the job of this block: the table is sorted by block_begin (uniform scalar search)
  int k = 0;
  for (int q = 1; q < tab.n; ++q)
    if (tab.j[q].block_begin <= (int)blockIdx.x) k = q;
  const ReduceJob& J = tab.j[k];
  const int b = (int)blockIdx.x - J.block_begin;
  const int bx = b % J.gx, rest = b / J.gx;
  const int by = rest % J.gy, bz = rest / J.gy;
  switch (J.kind) {   // uniform
    case RK_SLAB: reduce_slab_body(J, bx, by); break;
    case RK_CENTER: reduce_center_body(J, sm, bx, by); break;
    case RK_GENERIC: reduce_generic_body(J, sm, bx, by); break;
    case RK_WIDE8: reduce_wide_body<8>(J, sm, bx, by); break;
    case RK_WIDE16: reduce_wide_body<16>(J, sm, bx, by); break;
    case RK_TAP: reduce_tap_body(J, sm, bx, by, bz); break;
    default: reduce_stem_body(J, bx); break;
  }
}

// ---------------------------------------------------------------------------
// RGB stem weight gradient (Cx = 3): im2col rows (K = 27 -> 32) gathered into
// LDS; out[k][co] accumulates over pixels; one 32x32 MFMA block per co tile.
// ---------------------------------------------------------------------------
constexpr int SW_PIX = 128;  // pixels per stage
constexpr int SW_LDK = 32;

template <typename TD>
__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(const float* __restrict__ x,
                                                              const TD* __restrict__ dy,
                                                              float* __restrict__ partial, int N,
                                                              int H, int W, int Cout,
                                                              int stages_per_block,
                                                              long long total_stages) {
  __shared__ float A[SW_PIX * SW_LDK];  // [pix][k]
  __shared__ float D[SW_PIX * 32];      // [pix][co]
  __shared__ float R[4][32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int HW = H * W;
  const long long M = (long long)N * HW;
  const int co0 = blockIdx.y * 32;
  const long long st_begin = (long long)blockIdx.x * stages_per_block;
  const long long st_end = min(st_begin + (long long)stages_per_block, total_stages);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (long long st = st_begin; st < st_end; ++st) {
    const long long m0 = st * SW_PIX;
    for (int it = tid; it < SW_PIX * 9; it += 256) {
      const int pix = it / 9, t = it - pix * 9;
      const long long m = m0 + pix;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (m < M) {
        const int n = (int)(m / HW);
        const int r = (int)(m - (long long)n * HW);
        const int yy = r / W + t / 3 - 1, xx = r % W + t % 3 - 1;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
          const float* s = x + ((size_t)n * HW + (size_t)yy * W + xx) * 3;
          v0 = s[0]; v1 = s[1]; v2 = s[2];
        }
      }
      float* d = A + pix * SW_LDK + t * 3;
      d[0] = v0; d[1] = v1; d[2] = v2;
    }
    for (int i = tid; i < SW_PIX * 5; i += 256) A[(i / 5) * SW_LDK + 27 + i % 5] = 0.f;
    for (int i = tid; i < SW_PIX * 8; i += 256) {
      const int pix = i >> 3, seg = i & 7;
      const long long m = m0 + pix;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m < M) v = ld4(dy + (size_t)m * Cout + co0 + seg * 4);
      *reinterpret_cast<f32x4*>(D + pix * 32 + seg * 4) = v;
    }
    __syncthreads();
    // wave handles pixels [wave*32, wave*32+32)
#pragma unroll 4
    for (int q = 0; q < 32; q += 2) {
      const int pix = wave * 32 + q + lh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[pix * SW_LDK + li], D[pix * 32 + li], acc, 0, 0,
                                                 0);
    }
    __syncthreads();
  }
  // cross-wave reduce in fixed order
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    R[wave][row * 32 + li] = acc[r];
  }
  __syncthreads();
  for (int i = tid; i < 27 * 32; i += 256) {
    const float s = (R[0][i] + R[1][i]) + (R[2][i] + R[3][i]);
    const int k = i >> 5, c = i & 31;
    partial[((size_t)blockIdx.x * 27 + k) * Cout + co0 + c] = s;
  }
}

// Row form of the stem weight gradient for images whose width is a multiple of the 128-pixel
// stage: a stage lies in ONE image row, so instead of gathering 27 im2col values per pixel
// (scalar loads and index divisions per element) the three input rows y-1, y, y+1 are copied
// raw (coalesced) as R[ky][3*(col+1) + ci] and the im2col element k = 9*ky + (3*kx + ci) of
// pixel px is simply R[ky][3*px + (3*kx + ci)]: lane k of the A operand reads one word at a
// lane-constant offset plus 3*pixel.  Row pitch 393 = 9 mod 32 puts the 27 offsets on 27
// different banks.  Same stages, partial layout and reductions as conv_stem_wgrad_kernel.
constexpr int SWR_PITCH = 393;

// T = unsigned char: the dataset's uint8 image, normalised on load (see unet_stem_u8_fwd)
struct StemNormW { float mean[3], std[3]; };
__device__ __forceinline__ float stem_px(const float* x, size_t i, int, const StemNormW&) {
  return x[i];
}
__device__ __forceinline__ float stem_px(const unsigned char* x, size_t i, int c,
                                         const StemNormW& nm) {
  return ((float)x[i] / 255.0f - nm.mean[c]) / nm.std[c];
}

template <typename T, typename TD = float>
__global__ __launch_bounds__(256) void conv_stem_wgrad_rows_kernel(
    const T* __restrict__ x, const TD* __restrict__ dy, float* __restrict__ partial, int N,
    int H, int W, int Cout, int stages_per_block, long long total_stages, const StemNormW nm) {
  __shared__ float Rw[3 * SWR_PITCH];
  __shared__ float D[SW_PIX * 32];      // [pix][co]
  __shared__ float R[4][32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int HW = H * W;
  const int co0 = blockIdx.y * 32;
  const long long st_begin = (long long)blockIdx.x * stages_per_block;
  const long long st_end = min(st_begin + (long long)stages_per_block, total_stages);
  // A-operand lane: row k = li of the 32 x 2 block; k >= 27 contributes zero
  const int kk = li < 27 ? li : 0;
  const int a_off = (kk / 9) * SWR_PITCH + (kk % 9) + 3 * lh;
  const float a_on = li < 27 ? 1.f : 0.f;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // The operands of stage st + 1 are fetched into registers while the matrix cores work on stage
  // st (round 4; the loop used to load, wait, compute: 101 us for the 512 x 512 x 8 stem on bf16
  // tensors against 32 us of HBM time).
  static_assert(SW_PIX * 8 == 4 * 256, "four 4-channel segments of dy per thread and stage");
  constexpr int RV = (3 * 390 + 255) / 256;     // raw row words per thread
  float rreg[RV];
  f32x4 dreg[4];
  auto fetch = [&](long long st) __attribute__((always_inline)) {
    const long long m0 = st * SW_PIX;
    const int n = (int)(m0 / HW);
    const int rem = (int)(m0 - (long long)n * HW);
    const int yy = rem / W, x0 = rem - yy * W;
    // raw rows: word j of row ky = x[n][yy+ky-1][x0-1 + j/3][j%3], j < 3*130
#pragma unroll
    for (int t = 0; t < RV; ++t) {
      const int i = tid + 256 * t;
      const int ky = i / 390, j = i - ky * 390;
      const int iy = yy + ky - 1, ix = x0 - 1 + j / 3;
      float v = 0.f;
      if (i < 3 * 390 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        v = stem_px(x, ((size_t)n * HW + (size_t)iy * W + x0 - 1) * 3 + j, j % 3, nm);
      rreg[t] = v;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = tid + 256 * t;
      const int pix = i >> 3, seg = i & 7;
      dreg[t] = ld4(dy + (size_t)(m0 + pix) * Cout + co0 + seg * 4);
    }
  };
  auto stage_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < RV; ++t) {
      const int i = tid + 256 * t;
      const int ky = i / 390, j = i - ky * 390;
      if (i < 3 * 390) Rw[ky * SWR_PITCH + j] = rreg[t];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = tid + 256 * t;
      *reinterpret_cast<f32x4*>(D + (i >> 3) * 32 + (i & 7) * 4) = dreg[t];
    }
  };
  if (st_begin < st_end) {
    fetch(st_begin);
    stage_lds();
  }
  __syncthreads();
  for (long long st = st_begin; st < st_end; ++st) {
    const bool more = st + 1 < st_end;     // uniform
    if (more) fetch(st + 1);
    // wave handles pixels [wave*32, wave*32+32): 16 pixel pairs
#pragma unroll 4
    for (int q = 0; q < 32; q += 2) {
      const int px = wave * 32 + q;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Rw[a_off + 3 * px] * a_on,
                                                 D[(px + lh) * 32 + li], acc, 0, 0, 0);
    }
    __syncthreads();
    if (more) stage_lds();
    __syncthreads();
  }
  // cross-wave reduce in fixed order
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    R[wave][row * 32 + li] = acc[r];
  }
  __syncthreads();
  for (int i = tid; i < 27 * 32; i += 256) {
    const float sm = (R[0][i] + R[1][i]) + (R[2][i] + R[3][i]);
    const int k = i >> 5, c = i & 31;
    partial[((size_t)blockIdx.x * 27 + k) * Cout + co0 + c] = sm;
  }
}

// ---- host side of the reductions: launch at once, or queue until unet_wgrad_defer_flush ----
struct ReduceQueue {
  bool on = false;
  std::vector<std::vector<ReduceJob>> stage;   // stage k of every queued weight gradient
  int pending = 0;
};
ReduceQueue& reduce_queue() {
  static thread_local ReduceQueue q;
  return q;
}
int launch_reduce_jobs(const ReduceJob* jobs, int n, hipStream_t stream) {
  for (int i = 0; i < n; i += kReduceJobs) {
    ReduceTable t{};
    t.n = n - i < kReduceJobs ? n - i : kReduceJobs;
    long long blocks = 0;
    for (int k = 0; k < t.n; ++k) {
      t.j[k] = jobs[i + k];
      t.j[k].block_begin = (int)blocks;
      blocks += (long long)t.j[k].gx * t.j[k].gy * (t.j[k].kind == RK_TAP ? 9 : 1);
    }
    UNET_REQUIRE(blocks > 0 && blocks < (1LL << 31), "wgrad_reduce: bad grid");
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, t);
    UNET_CHECK_LAUNCH("wgrad_reduce_batched");
  }
  return UNET_OK;
}
int emit_reduce(int stage, const ReduceJob& j, hipStream_t stream) {
  ReduceQueue& q = reduce_queue();
  if (!q.on) return launch_reduce_jobs(&j, 1, stream);
  if ((int)q.stage.size() <= stage) q.stage.resize(stage + 1);
  q.stage[stage].push_back(j);
  ++q.pending;
  return UNET_OK;
}
int flush_reduce_queue(hipStream_t stream) {
  ReduceQueue& q = reduce_queue();
  int rc = UNET_OK;
  for (auto& st : q.stage) {
    if (!st.empty() && rc == UNET_OK) rc = launch_reduce_jobs(st.data(), (int)st.size(), stream);
    st.clear();
  }
  q.pending = 0;
  return rc;
}
// The reduction of ONE weight gradient: `nslab` slabs of E floats at `ws` (scratch for the
// chunked stages right behind them) summed into dw.  stem: slabs [27][Cout] -> dw[co][3][9].
int emit_wgrad_reduction(float* ws, int nslab, size_t E, float* dw, int Cx, int Cout, int ci_off,
                         int Cin_total, bool center_only, bool stem, hipStream_t stream) {
  const float* cur = ws;
  float* ping = ws + (size_t)nslab * E;
  float* pong = ping + (size_t)ceil_div(nslab, kSlabChunk) * E;
  int stage = 0;
  while (nslab > 16) {
    const int chunks = ceil_div(nslab, kSlabChunk);
    ReduceJob j{};
    j.src = cur; j.dst = ping; j.nslab = nslab; j.Cx = (int)(E / 4); j.kind = RK_SLAB;
    j.gx = (int)ceil_div64((long long)(E / 4), 256); j.gy = chunks;
    const int rc = emit_reduce(stage++, j, stream);
    if (rc != UNET_OK) return rc;
    cur = ping;
    float* t = ping; ping = pong; pong = t;
    nslab = chunks;
  }
  ReduceJob j{};
  j.src = cur; j.dst = dw; j.nslab = nslab; j.Cx = Cx; j.Cout = Cout; j.ci_off = ci_off;
  j.Cin_total = Cin_total;
  if (stem) {
    j.kind = RK_STEM; j.gx = ceil_div(27 * Cout, 256); j.gy = 1;
  } else if (center_only) {
    j.kind = RK_CENTER; j.gx = Cout / 32; j.gy = Cx / 32;
  } else if ((long long)Cx * Cout <= 128 * 128) {   // few output tiles: one tap per block
    j.kind = RK_TAP; j.gx = Cout / 32; j.gy = Cx / 8;
  } else if (Cout % 64 == 0 && Cx % 16 == 0) {
    if ((long long)Cx * Cout >= 512 * 512) { j.kind = RK_WIDE16; j.gx = Cout / 64; j.gy = Cx / 16; }
    else { j.kind = RK_WIDE8; j.gx = Cout / 64; j.gy = Cx / 8; }
  } else {
    j.kind = RK_GENERIC; j.gx = Cout / 32; j.gy = Cx / 8;
  }
  return emit_reduce(stage, j, stream);
}

struct WgradPlan {
  int npp;   // pixel-pair parts per segment (waves that share a sub-block)
  int sps;   // slabs per pixel split: 1 (the parts of a sub-block are merged in LDS)
  int nw;    // waves per workgroup: 4, or 8 (fp32 tensors, one workgroup per CU, LDS merge)
  int ci_t, co_t, S, split, segs_per_row, total_segs, segs_per_block;
  int rg;    // conv_wgrad_b16_ring_kernel: row groups of a workgroup (2 = eight waves), 0 = not planned for it
  size_t ws_floats;
  bool stem;
  int stem_blocks, stem_spb;
  long long stem_stages;
};

int stem_grid(long long stages) { return (int)(stages < 1024 ? stages : 1024); }

int wgrad_tiles(int Cx, int Cout) {
  int ci_t = (Cx % 64 == 0) ? 64 : 32;
  const int co_t = (Cout % 64 == 0) ? 64 : 32;
  if (ci_t == 64 && co_t == 32) ci_t = 32;
  return (Cx / ci_t) * (Cout / co_t);
}

// wide: the 8-wave kernel (fp32 tensors on the fp32 matrix cores)
// (2: measured against 4 per layer - equal or 5-8 % faster, and 24 registers fewer)
constexpr int kRingDepth = 2;   // steps whose operands are in flight in conv_wgrad_b16_ring_kernel
WgradPlan make_plan(int N, int H, int W, int Cx, int Cout, int stride, int prec = 0,
                    bool wide = false, bool ring8 = false, bool ring_s2 = false) {
  WgradPlan pl{};
  pl.nw = 4;
  if (Cx == 3) {
    pl.stem = true;
    const long long M = (long long)N * H * W;
    pl.stem_stages = ceil_div64(M, SW_PIX);
    pl.stem_blocks = stem_grid(pl.stem_stages);
    pl.stem_spb = (int)ceil_div64(pl.stem_stages, pl.stem_blocks);
    pl.stem_blocks = (int)ceil_div64(pl.stem_stages, pl.stem_spb);
    pl.ws_floats = (size_t)pl.stem_blocks * 27 * Cout +
                   2 * (size_t)ceil_div(pl.stem_blocks, kSlabChunk) * 27 * Cout;
    return pl;
  }
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  pl.ci_t = (Cx % 64 == 0) ? 64 : 32;
  pl.co_t = (Cout % 64 == 0) ? 64 : 32;
  if (pl.ci_t == 64 && pl.co_t == 32) pl.ci_t = 32;  // instantiated: 32x32, 32x64, 64x64
  const int nsb = (pl.ci_t / 32) * (pl.co_t / 32);
  if (wide && prec == 0) {
    // one 8-wave workgroup per CU; a segment as long as the image row and 133 KB of stages
    // allow (16 pixel pairs per wave per barrier on 64x64 tiles at S = 64)
    pl.nw = 8;
    pl.npp = 8 / nsb;
    pl.sps = 1;
    const int smax = nsb == 1 ? (stride == 1 ? 128 : 32)
                   : nsb == 2 ? 64
                              : (stride == 1 ? 64 : 32);
    pl.S = 16;
    while (pl.S < smax && pl.S < Wo) pl.S *= 2;
  } else {
    pl.npp = 4 / nsb;  // pixel-pair parts per segment, each with a slab of its own
    pl.sps = 1;
    // segment length: enough pixel pairs per wave per stage, within the LDS budget
    if (Wo <= 16) pl.S = 16;
    else if (pl.ci_t == 64) pl.S = (stride == 2 || prec == 3) ? 16 : 32;  // bf16x3: 3 planes in LDS
    else if (pl.co_t == 64) pl.S = 32;                       // 32x64 tile
    else pl.S = (stride == 1 && Wo >= 64) ? 64 : 32;         // 32x32 tile
    // (tried for the bf16 operands, whose stages are half as large: segments of 64 / 128 pixels
    // on the 64 x 64 / 32 x 32 tiles - the bf16 weight-gradient group went from 2.40 to 2.74 ms
    // per step, profiles/r04_bf16_experiments.txt: kept at 32 / 64)
  }
  pl.segs_per_row = ceil_div(Wo, pl.S);
  pl.total_segs = N * Ho * pl.segs_per_row;
  const int tiles = (Cx / pl.ci_t) * (Cout / pl.co_t);
  // aim for two 4-wave workgroups / one 8-wave workgroup per CU, at least 4 segments per block
  int split = ceil_div(pl.nw == 8 ? 256 : 512, tiles);
  const int max_split = ceil_div(pl.total_segs, 4);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  pl.segs_per_block = ceil_div(pl.total_segs, split);
  if (((prec == 1 && stride == 1) || (ring_s2 && stride == 2)) && pl.nw == 4) {
    // the row-ring kernel (bf16 tensors) walks down a column strip: a workgroup's range must
    // lie inside one strip and be a whole number of prefetch rounds.  Harmless to the segment
    // kernels, which enumerate the same g differently.  ring8: its eight-wave form - one
    // workgroup per CU, two rows a step, half the slabs - where the rows allow it.
    auto rows_for = [&](int target_blocks, int rg) {
      int sp = ceil_div(target_blocks, tiles);
      if (sp > max_split) sp = max_split;
      if (sp < 1) sp = 1;
      int r = ceil_div(pl.total_segs, sp);
      const int q = kRingDepth * rg;
      while (r < Ho && (Ho % r || r % q)) ++r;
      return (r <= Ho && Ho % r == 0 && r % q == 0) ? r : 0;
    };
    const int r2 = ring8 ? rows_for(256, 2) : 0;
    const int r1 = rows_for(512, 1);
    if (r2) { pl.segs_per_block = r2; pl.rg = 2; }
    else if (r1) { pl.segs_per_block = r1; pl.rg = 1; }
  }
  pl.split = ceil_div(pl.total_segs, pl.segs_per_block);
  // slabs + ping-pong room for the staged reduction (each stage shrinks 16x)
  const size_t E = (size_t)9 * Cx * Cout;
  const int slabs = pl.split * pl.sps;
  pl.ws_floats = (size_t)slabs * E + 2 * (size_t)ceil_div(slabs, kSlabChunk) * E;
  return pl;
}

// 8-wave form: LDS = the two stages, at least the 96 KB the merge epilogue uses
template <int CI_T, int CO_T, int S, int STRIDE>
int launch_wgrad8(const WgradParams& p, hipStream_t stream) {
  constexpr int PW = (S - 1) * STRIDE + 3;
  constexpr size_t stages = 2 * (size_t)(3 * PW * CI_T + S * CO_T) * sizeof(float);
  constexpr size_t lds = stages > kWgradMergeLds8 ? stages : kWgradMergeLds8;
  static_assert(lds <= 160 * 1024, "stages exceed the CU's LDS");
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  if (p.alpha) {
    auto kern = conv_wgrad_kernel<CI_T, CO_T, S, STRIDE, true, float, float, 8>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
  } else {
    auto kern = conv_wgrad_kernel<CI_T, CO_T, S, STRIDE, false, float, float, 8>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_wgrad(8 waves)");
  return UNET_OK;
}

template <int CI_T, int CO_T, int STRIDE>
int launch_wgrad8_s(const WgradParams& p, int S, hipStream_t stream) {
  constexpr int NSB = (CI_T / 32) * (CO_T / 32);
  constexpr int SMAX = NSB == 1 ? (STRIDE == 1 ? 128 : 32) : NSB == 2 ? 64 : (STRIDE == 1 ? 64 : 32);
  if constexpr (SMAX >= 128) if (S == 128) return launch_wgrad8<CI_T, CO_T, 128, STRIDE>(p, stream);
  if constexpr (SMAX >= 64) if (S == 64) return launch_wgrad8<CI_T, CO_T, 64, STRIDE>(p, stream);
  if (S == 32) return launch_wgrad8<CI_T, CO_T, 32, STRIDE>(p, stream);
  if (S == 16) return launch_wgrad8<CI_T, CO_T, 16, STRIDE>(p, stream);
  unet_set_error("conv_wgrad(8 waves): no instantiation for S=%d", S);
  return UNET_E_INVALID;
}

// ---------------------------------------------------------------------------
// Winograd F(3x3, 2x2) weight gradient of a stride-1 3x3 convolution on the fp32 matrix cores:
// per 2x2 tile of dy (e) and its 4x4 input window (d)
//     dW_tile = A^T [ (G e G^T) .* (B^T d B) ] A,
// 16 multiplies instead of the 36 of the direct form (2.25x fewer MFMA FLOPs); the sum over
// tiles and images is linear, so it is taken in the transform domain: 16 independent GEMMs
//     M_xi[ci][co] = sum over tiles of V_xi[tile][ci] * E_xi[tile][co]
// with K = tiles, and A^T M A is applied once per workgroup in the epilogue.
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1]   (F(3,2): the last row differs from F(2,3))
//   G   = [1 0; 1/2 1/2; 1/2 -1/2; 0 1],   A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1].
// Workgroup = 8 waves (one per CU) owning a 64 ci x 64 co tile for all 16 xi (wave w: 16 ci x
// 32 co, v_mfma_f32_16x16x4_f32, 128 accumulator VGPRs) and a contiguous range of K chunks; a
// chunk = 8 horizontally adjacent tiles (16 x 2 output pixels).  Per chunk the 4 x 18 pixel
// window of x (activated on load in the fused pipeline) and the 2 x 16 pixels of dy are staged
// raw ([pixel][64 channels], 16-byte stores), then every thread transforms one (channel, tile
// pair, half of the xi rows) of each operand - channel = lane, so all LDS traffic of the
// transforms is conflict-free - into V / E stored as [xi][tile pair][channel][2] (bit 5 of the
// offset XOR-flipped for odd pairs: conflict-free 8-byte fragment reads).  Slabs, reductions
// and the OIHW scatter are those of conv_wgrad_kernel.
// ---------------------------------------------------------------------------
constexpr int WW_XP = 4 * 18, WW_DP = 2 * 16;          // staged pixels of x / dy per chunk
constexpr int WW_BUF = 16 * 512;                        // floats of one V / E stage
constexpr size_t WW_LDS = (size_t)(4 * WW_BUF + (WW_XP + WW_DP) * 64) * sizeof(float);

// The K loop is written for instruction count (lds_asm.h): loop-invariant lane offsets + scalar
// chunk offsets for every global load, the chunk cursor (n, tile row, chunk column) advanced
// with scalar compares, transforms as single v_add / v_sub (hipcc's v_pk_add_f32 forms cost
// twice as much and a v_mov per operand pair), the 1/2 factors of G e G^T deferred to the
// epilogue (exact: powers of two), LDS reads issued ahead of a stage's MFMAs and consumed behind
// them, every LDS offset an immediate (the chunk body is instantiated per stage parity and per
// transform half of the wave).
template <bool ACT>
__global__ __launch_bounds__(512, 2) void conv_wgrad_wino_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                       // [buf][xi][pair 4][ci 64][2]
  float* Es = smem + 2 * WW_BUF;          // [buf][xi][pair 4][co 64][2]
  float* Rx = smem + 4 * WW_BUF;          // [pixel 72][ci 64]
  float* Rd = Rx + WW_XP * 64;            // [pixel 32][co 64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
  const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
  const int sp = bid;
  const int ci0 = ci_t * 64, co0 = co_t * 64;
  const int g_begin = sp * p.segs_per_block;
  const int g_end = min(g_begin + p.segs_per_block, p.total_segs);
  const int cw = p.W >> 4, th = p.H >> 1;      // chunks per tile row, tile rows per image

  // descriptors: x starts one image row + one pixel early, so the scalar offset of a chunk
  // (its window starts at row 2 tr - 1, column 16 cc - 1) is never negative
  const int xshift = (p.W + 1) * p.Cx;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x) - xshift, 0, (int)(p.x_bytes + 4u * (unsigned)xshift), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

  // ---- raw slots: x 72 pixels x 16 channel groups = 1152 (thread: tid, +512, +1024 < 1152),
  //      dy 32 x 16 = 512 (one per thread); the channel group is tid & 15 for every slot.
  //      Per slot: the lane's byte offset inside a chunk window and the border flags of its
  //      window position (top row, bottom row, left column, right column, no such slot) ----
  const int grp4 = (tid & 15) * 4;
  unsigned voffx[3], fx[3];
  float* xdst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int pix = (tid + 512 * i) >> 4;                     // 0..95 (>= 72: no slot)
    const int prow = pix / 18, pcol = pix - prow * 18;
    voffx[i] = (unsigned)((prow * p.W + pcol) * p.Cx + ci0 + grp4) * 4u;
    fx[i] = (prow == 0 ? 1u : 0u) | (prow == 3 ? 2u : 0u) | (pcol == 0 ? 4u : 0u) |
            (pcol == 17 ? 8u : 0u) | (pix >= WW_XP ? 16u : 0u);
    xdst[i] = Rx + (pix < WW_XP ? pix : 0) * 64 + grp4;
  }
  const bool slot2 = tid + 1024 < WW_XP * 16;
  const int d_pixl = tid >> 4;                                      // 0..31
  const unsigned voffd = (unsigned)(((d_pixl >> 4) * p.W + (d_pixl & 15)) * p.Cout + co0 + grp4) * 4u;
  float* const ddst = Rd + d_pixl * 64 + grp4;
  f32x4 rx[3], rd;
  float okf[3] = {1.f, 1.f, 1.f};
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  // chunk cursor of the loader (scalar): chunk column, tile row, image
  int l_cc, l_tr, l_n, l_g = g_begin;
  {
    const int r = g_begin / cw;
    l_cc = g_begin - r * cw;
    l_n = r / th;
    l_tr = r - l_n * th;
  }
  auto load_raw = [&]() {     // loads chunk l_g, then advances the cursor (the tail re-loads the last chunk)
    const unsigned m = (l_tr == 0 ? 1u : 0u) | (l_tr == th - 1 ? 2u : 0u) | (l_cc == 0 ? 4u : 0u) |
                       (l_cc == cw - 1 ? 8u : 0u) | 16u;
    const int pixbase = (l_n * p.H + 2 * l_tr) * p.W + 16 * l_cc;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool ok = (fx[i] & m) == 0u;
      okf[i] = ok ? 1.f : 0.f;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rsx, ok ? voffx[i] : 0x80000000u, pixbase * p.Cx * 4, 0));
    }
    rd = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsd, voffd,
                                                                          pixbase * p.Cout * 4, 0));
    if (ACT) {   // the loaded coefficients are only looked at when the patch is stored
      const size_t o = (size_t)l_n * p.Cx + ci0;
      // (compiler-tracked loads on purpose: the values live across the loop's back edge, where
      // the register allocator copies them between the chunk bodies - an untracked asm load
      // still in flight at such a copy is copied stale and lands in a register that has been
      // given to something else.  Seen as memory faults when two processes shared the GPU.)
      ca = *reinterpret_cast<const f32x4*>(p.alpha + o + grp4);
      cb = *reinterpret_cast<const f32x4*>(p.beta + o + grp4);
    }
    if (l_g + 1 < g_end) {     // uniform
      ++l_g;
      if (++l_cc == cw) { l_cc = 0; if (++l_tr == th) { l_tr = 0; ++l_n; } }
    }
  };
  auto store_x = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    f32x4 v = rx[i];
    if (ACT) v = act4f(v, ca, cb, p.slope, okf[i]);
    if (i < 2 || slot2) *reinterpret_cast<f32x4*>(xdst[i]) = v;
  };
  auto store_d = [&]() { *reinterpret_cast<f32x4*>(ddst) = rd; };

  // ---- transforms: thread -> (channel = lane, tile pair = wave & 3, xi rows {2h, 2h+1}) ----
  const int t_pair = wave & 3;
  const int t_half = __builtin_amdgcn_readfirstlane(wave >> 2);
  const unsigned t_xa = lds_addr(Rx + (4 * t_pair) * 64 + lane);   // window columns 4 pair .. +5
  const unsigned t_da = lds_addr(Rd + (4 * t_pair) * 64 + lane);   // dy columns 4 pair .. +3
  const int t_off = t_pair * 128 + ((lane * 2) ^ (32 * (t_pair & 1)));
  float* const t_vdst = Vs + t_off;
  float* const t_edst = Es + t_off;
  f32x2v tq[6];      // what a transform piece read ahead of its stage's MFMAs
  // V = B^T d B, xi row a, for the two tiles of the pair: tv_rd reads the two window rows the
  // row combines over the pair's 6 columns (one ds_read2st64_b32 per column), tv_wr combines
  // them, runs the column pass per tile and writes four 8-byte (tile pair) values
  auto tv_rd = [&](auto ac) {
    constexpr int a = decltype(ac)::value;
    constexpr int r0 = a == 0 ? 0 : (a == 1 ? 1 : (a == 2 ? 2 : 3));
    constexpr int r1 = a == 0 ? 2 : (a == 1 ? 2 : 1);
    tq[0] = lds_rd2st64<r0 * 18 + 0, r1 * 18 + 0>(t_xa);
    tq[1] = lds_rd2st64<r0 * 18 + 1, r1 * 18 + 1>(t_xa);
    tq[2] = lds_rd2st64<r0 * 18 + 2, r1 * 18 + 2>(t_xa);
    tq[3] = lds_rd2st64<r0 * 18 + 3, r1 * 18 + 3>(t_xa);
    tq[4] = lds_rd2st64<r0 * 18 + 4, r1 * 18 + 4>(t_xa);
    tq[5] = lds_rd2st64<r0 * 18 + 5, r1 * 18 + 5>(t_xa);
  };
  auto tv_wr = [&](auto ac, auto bc) {
    constexpr int a = decltype(ac)::value;
    constexpr int B = decltype(bc)::value;
    float c[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) c[j] = a == 1 ? vadd(tq[j][0], tq[j][1]) : vsub(tq[j][0], tq[j][1]);
    float* dst = t_vdst + B * WW_BUF + (4 * a) * 512;
    *reinterpret_cast<f32x2v*>(dst) = f32x2v{vsub(c[0], c[2]), vsub(c[2], c[4])};
    *reinterpret_cast<f32x2v*>(dst + 512) = f32x2v{vadd(c[1], c[2]), vadd(c[3], c[4])};
    *reinterpret_cast<f32x2v*>(dst + 1024) = f32x2v{vsub(c[2], c[1]), vsub(c[4], c[3])};
    *reinterpret_cast<f32x2v*>(dst + 1536) = f32x2v{vsub(c[3], c[1]), vsub(c[5], c[3])};
  };
  // E = G e G^T WITHOUT its factors 1/2 (rows / columns 1 and 2 carry one each: the epilogue
  // scales M_xi instead - exact), xi row a, for the two tiles of the pair (4 dy columns)
  auto te_rd = [&](auto ac) {
    constexpr int a = decltype(ac)::value;
    if constexpr (a == 0) {            // e0 only
      tq[0] = lds_rd2st64<0, 1>(t_da);
      tq[1] = lds_rd2st64<2, 3>(t_da);
    } else if constexpr (a == 3) {     // e1 only
      tq[0] = lds_rd2st64<16, 17>(t_da);
      tq[1] = lds_rd2st64<18, 19>(t_da);
    } else {                           // (e0, e1) per column
      tq[0] = lds_rd2st64<0, 16>(t_da);
      tq[1] = lds_rd2st64<1, 17>(t_da);
      tq[2] = lds_rd2st64<2, 18>(t_da);
      tq[3] = lds_rd2st64<3, 19>(t_da);
    }
  };
  auto te_wr = [&](auto ac, auto bc) {
    constexpr int a = decltype(ac)::value;
    constexpr int B = decltype(bc)::value;
    float gq[4];
    if constexpr (a == 0 || a == 3) {
      gq[0] = tq[0][0]; gq[1] = tq[0][1]; gq[2] = tq[1][0]; gq[3] = tq[1][1];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) gq[j] = a == 1 ? vadd(tq[j][0], tq[j][1]) : vsub(tq[j][0], tq[j][1]);
    }
    float* dst = t_edst + B * WW_BUF + (4 * a) * 512;
    *reinterpret_cast<f32x2v*>(dst) = f32x2v{gq[0], gq[2]};
    *reinterpret_cast<f32x2v*>(dst + 512) = f32x2v{vadd(gq[0], gq[1]), vadd(gq[2], gq[3])};
    *reinterpret_cast<f32x2v*>(dst + 1024) = f32x2v{vsub(gq[0], gq[1]), vsub(gq[2], gq[3])};
    *reinterpret_cast<f32x2v*>(dst + 1536) = f32x2v{gq[1], gq[3]};
  };
  // the four transform pieces of a thread: its two xi rows (2 half, 2 half + 1) of both operands
  auto t_reads = [&](auto kc, auto hc) -> int {      // returns nothing useful; NR below
    constexpr int k = decltype(kc)::value, h = decltype(hc)::value;
    if constexpr (k < 2) tv_rd(template_ic<2 * h + k>{}); else te_rd(template_ic<2 * h + k - 2>{});
    return 0;
  };
  auto t_writes = [&](auto kc, auto hc, auto bc) {
    constexpr int k = decltype(kc)::value, h = decltype(hc)::value;
    if constexpr (k < 2) tv_wr(template_ic<2 * h + k>{}, bc); else te_wr(template_ic<2 * h + k - 2>{}, bc);
  };

  // ---- MFMA fragments: wave -> ci rows 16 tg .. +15, co columns 32 nh .. +31 ----
  const int tg = wave & 3, nh = wave >> 2;
  const int fm = lane & 15, fk = lane >> 4;
  const unsigned va = lds_addr(Vs + fk * 128 + (((16 * tg + fm) * 2) ^ (32 * (fk & 1))));
  const unsigned eb0 = lds_addr(Es + fk * 128 + (((32 * nh + fm) * 2) ^ (32 * (fk & 1))));
  const unsigned eb1 = lds_addr(Es + fk * 128 + (((32 * nh + 16 + fm) * 2) ^ (32 * (fk & 1))));
  f32x4 acc[16][2];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // two fragment sets: the reads of xi + 1 are issued ahead of the MFMAs of xi
  f32x2v fa[2], fb0[2], fb1[2];
  auto frag = [&](auto bc, auto xc) {
    constexpr int x = decltype(xc)::value, B = decltype(bc)::value, sl = x & 1;
    fa[sl] = lds_rd64<(B * WW_BUF + x * 512) * 4>(va);
    fb0[sl] = lds_rd64<(B * WW_BUF + x * 512) * 4>(eb0);
    fb1[sl] = lds_rd64<(B * WW_BUF + x * 512) * 4>(eb1);
  };
  auto mm = [&](auto xc) {
    constexpr int x = decltype(xc)::value, sl = x & 1;
    acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][0], fb0[sl][0], acc[x][0], 0, 0, 0);
    acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][0], fb1[sl][0], acc[x][1], 0, 0, 0);
    acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][1], fb0[sl][1], acc[x][0], 0, 0, 0);
    acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][1], fb1[sl][1], acc[x][1], 0, 0, 0);
  };
  // reads a transform piece issues ahead of its stage (piece k of half h)
  auto n_reads = [](int k, int h) { return k < 2 ? 6 : ((2 * h + k - 2) % 3 == 0 ? 2 : 4); };

  // One chunk on stage parity B by a wave of transform half HF: stage B is complete at the first
  // barrier; xi 0..7 carry the raw stores of chunk g + 1 (loaded an iteration ago) and the loads
  // of chunk g + 2; after the second barrier (raw chunk g + 1 complete) xi 8..11 carry its
  // transform into stage B ^ 1.
  auto chunk_body = [&](auto bc, auto hc) {
    constexpr int B = decltype(bc)::value, HF = decltype(hc)::value;
    using NB = template_ic<B ^ 1>;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    frag(bc, template_ic<0>{});
    for_range<0, 16>([&](auto xc) {
      constexpr int x = decltype(xc)::value;
      constexpr bool T = x >= 8 && x < 12;
      constexpr int NR = T ? n_reads(T ? x - 8 : 0, HF) : 0;
      if constexpr (x == 8) {      // raw chunk g + 1 complete (LDS only; the fragments of xi 8 in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if constexpr (T) t_reads(template_ic<(T ? x - 8 : 0)>{}, hc);
      if constexpr (x + 1 < 16) frag(bc, template_ic<x + 1>{});
      lds_wait<(x + 1 < 16 ? 3 : 0) + NR>(fa[x & 1], fb0[x & 1], fb1[x & 1]);
      __builtin_amdgcn_sched_barrier(0);
      mm(xc);
      if constexpr (T) {
        lds_wait6<3>(tq[0], tq[1], tq[2], tq[3], tq[4], tq[5]);   // (only fragment reads in flight)
        t_writes(template_ic<(T ? x - 8 : 0)>{}, hc, NB{});
      }
      if constexpr (x < 3) store_x(template_ic<(x < 3 ? x : 0)>{});   // chunk g + 1
      if constexpr (x == 3) store_d();
      if constexpr (x == 4) load_raw();
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto k_loop = [&](auto hc) {
    // prologue: chunk 0 transformed into stage 0, chunk 1 in the registers
    load_raw();
    for_range<0, 3>(store_x);
    store_d();
    load_raw();
    __syncthreads();
    for_range<0, 4>([&](auto kc) {
      t_reads(kc, hc);
      lds_wait6<0>(tq[0], tq[1], tq[2], tq[3], tq[4], tq[5]);
      t_writes(kc, hc, template_ic<0>{});
    });
    for (int g = g_begin; g < g_end; g += 2) {
      chunk_body(template_ic<0>{}, hc);
      if (g + 1 < g_end) chunk_body(template_ic<1>{}, hc);
    }
  };
  if (g_begin < g_end) {
    if (t_half == 0) k_loop(template_ic<0>{}); else k_loop(template_ic<1>{});   // (scalar branch)
  }

  // ---- epilogue: dW = A^T M A per (ci, co), register-local; lane holds ci = 16 tg + 4 fk + r,
  //      co = 32 nh + 16 b + fm.  M_xi first gets the factors of G e G^T the transform left out:
  //      1/2 per xi row / column 1 or 2 ----
#pragma unroll
  for (int x = 0; x < 16; ++x) {
    const float sc = (((x >> 2) == 1 || (x >> 2) == 2) ? 0.5f : 1.f) * (((x & 3) == 1 || (x & 3) == 2) ? 0.5f : 1.f);
    if (sc != 1.f) {
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[x][b] *= sc;
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[0][j] = acc[0 + j][b][r] + acc[4 + j][b][r] + acc[8 + j][b][r];
        s[1][j] = acc[4 + j][b][r] - acc[8 + j][b][r];
        s[2][j] = acc[4 + j][b][r] + acc[8 + j][b][r] + acc[12 + j][b][r];
      }
      const int row = ci0 + 16 * tg + 4 * fk + r, col = co0 + 32 * nh + 16 * b + fm;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        float* out = p.partial + ((size_t)(sp * 9 + u * 3) * p.Cx + row) * p.Cout + col;
        const size_t ts = (size_t)p.Cx * p.Cout;
        out[0] = s[u][0] + s[u][1] + s[u][2];
        out[ts] = s[u][1] - s[u][2];
        out[2 * ts] = s[u][1] + s[u][2] + s[u][3];
      }
    }
}

// ---------------------------------------------------------------------------
// Winograd F(3x3, 2x2) weight gradient of the 32 -> 32 channel layers at full resolution (the
// direct kernel: 349 us each, three per step), by the recipe of conv_wino32q_kernel
// (conv_c32.hip): the (ci, co) tile IS the layer, so one persistent workgroup per CU keeps all
// 16 xi accumulators (wave (i, h): row i of the xi grid, output-channel half h: 32 VGPRs) over
// its whole walk and transforms BOTH operands on chip per unit of 4 x 16 pixels (16 tiles = four
// k steps): waves 0-3 take V = B^T d B of the activated x patch in LDS, waves 4-7 E' = G' e G'^T
// of the 2 x 2 dy tiles straight from global memory (G' = G without its halves, which the
// epilogue applies: exact).  No exchange and no store per unit - two barriers - and ONE
// A^T M A + slab per workgroup at the end.
// ---------------------------------------------------------------------------
constexpr int WQ_PW = 18, WQ_PPIX = 108, WQ_LDA = 36, WQ_SLOTS = WQ_PPIX * 8;
constexpr int WQ_PASSES = (WQ_SLOTS + 511) / 512;          // 2
constexpr int WQ_VP = 32;                                   // row pitch of V / E (floats): see the products
constexpr int WQ_V = 16 * 16 * WQ_VP;                       // [xi][tile slot 16][channel 32]
constexpr size_t WQ_LDS = ((size_t)WQ_PPIX * WQ_LDA + 2 * WQ_V) * sizeof(float);
static_assert(16 * 1024 <= 2 * WQ_V, "the epilogue exchange lives in the V / E stages");

template <bool ACT, bool DZ>
__global__ __launch_bounds__(512, 1) void conv_wgrad_wino32_kernel(const WgradParams p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Pb = smem;
  float* const Vs = smem + WQ_PPIX * WQ_LDA;
  float* const Es = Vs + WQ_V;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  const int tiles_x = W / 32, tiles_y = H / 8;
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
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  if (DZ && blockIdx.x == 0) {   // parameter gradients of the layer's norm / bias (N x 32 sums)
    const float hw = (float)(H * W), inv = 1.f / hw;
    if (tid < 32) {
      float dg = 0.f, db = 0.f, dbi = 0.f;
      for (int q = 0; q < p.N; ++q) {
        const float2 v = p.dz_sums[(size_t)q * 32 + tid];
        db += v.x;
        dg += v.y;
        dbi += p.dz_gamma[tid] * p.dz_rstd[(size_t)q * 32 + tid] * (v.x - hw * (v.x * inv));
      }
      if (p.dz_dgamma) p.dz_dgamma[tid] = dg;
      if (p.dz_dbeta) p.dz_dbeta[tid] = db;
      if (p.dz_dbias) p.dz_dbias[tid] = dbi;
    }
  }

  // ---- patch slots of this thread ----
  int pp_rel[WQ_PASSES], pp_lds[WQ_PASSES], pp_rc[WQ_PASSES];
#pragma unroll
  for (int i = 0; i < WQ_PASSES; ++i) {
    const int slot = tid + 512 * i;
    const bool valid = slot < WQ_SLOTS;
    const int pix = valid ? slot >> 3 : 0, seg = slot & 7;
    const int prow = pix / WQ_PW, pcol = pix - prow * WQ_PW;
    pp_rel[i] = ((prow * W + pcol) * 32 + seg * 4) * 4;
    pp_lds[i] = pix * WQ_LDA + seg * 4;
    pp_rc[i] = valid ? (prow | (pcol << 8)) : (1 << 20);
  }
  f32x4 pr[WQ_PASSES];
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};
  unsigned okm = 0;
  int n_coef = -1;
  auto tile_pos = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % tiles_x;
    const int r = tile / tiles_x;
    const int ty = r % tiles_y;
    n = r / tiles_y; y0 = ty * 8; x0 = tx * 32;
  };
  // ---- operand roles: waves 0-3 transform x, waves 4-7 dy; slot tt = 4 (wave & 3) + (lane >> 4)
  //      sits at tile row (tt & 3) >> 1, tile column 4 (tt & 1) + (tt >> 2) ----
  const bool xside = wave < 4;   // uniform
  const int cp = tid & 15, tt = (tid >> 4) & 15;
  const int t_ty = (tt & 3) >> 1, t_tx = 4 * (tt & 1) + (tt >> 2);
  const unsigned t_srca = lds_addr(Pb + ((2 * t_ty) * WQ_PW + 2 * t_tx) * WQ_LDA + 2 * cp);
  float* const t_dst = (xside ? Vs : Es) + tt * WQ_VP + 2 * cp;
  f32x2v en[4];   // dy side: the next unit's 2 x 2 tile (channel pair cp)
  f32x2v yn[4];   // DZ: the same of the layer's raw output y
  f32x2v cz[5];   // DZ: a1, b1, P, Q, R of image n_cz for this channel pair
  int n_cz = -1;
  size_t en_off = 0;   // element offset of the tile in the registers (DZ: where dz goes)
  auto load_unit = [&](int n, int yu, int xu) {   // patch of x (all threads) + dy tile (waves 4-7)
    const int base = ((n * H + yu - 1) * W + xu - 1) * 128;
    okm = 0;
#pragma unroll
    for (int i = 0; i < WQ_PASSES; ++i) {
      const int prow = pp_rc[i] & 0xff, pcol = pp_rc[i] >> 8;
      const bool ok = (unsigned)(yu - 1 + prow) < (unsigned)H && (unsigned)(xu - 1 + pcol) < (unsigned)W;
      okm |= (ok ? 1u : 0u) << i;
      const unsigned off = ok ? (unsigned)(base + pp_rel[i]) : 0x80000000u;
      pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0));
    }
    if (!xside) {   // uniform
      const size_t o = (((size_t)n * H + yu + 2 * t_ty) * W + xu + 2 * t_tx) * 32 + 2 * cp;
      const float* d = p.dy + o;
      en[0] = *reinterpret_cast<const f32x2v*>(d);
      en[1] = *reinterpret_cast<const f32x2v*>(d + 32);
      en[2] = *reinterpret_cast<const f32x2v*>(d + (size_t)W * 32);
      en[3] = *reinterpret_cast<const f32x2v*>(d + (size_t)W * 32 + 32);
      if (DZ) {
        const float* yy = p.dz_y + o;
        yn[0] = *reinterpret_cast<const f32x2v*>(yy);
        yn[1] = *reinterpret_cast<const f32x2v*>(yy + 32);
        yn[2] = *reinterpret_cast<const f32x2v*>(yy + (size_t)W * 32);
        yn[3] = *reinterpret_cast<const f32x2v*>(yy + (size_t)W * 32 + 32);
        en_off = o;
      }
    }
  };
  // DZ: dz of the tile in (e, y) with the coefficients of image n; written out, returned in e
  auto apply_dz = [&](f32x2v (&e)[4], const f32x2v (&y)[4], size_t off, int n) {
    if (n != n_cz) {   // uniform: once per image
      const size_t plane = (size_t)p.N * 32;
#pragma unroll
      for (int k = 0; k < 5; ++k)
        cz[k] = *reinterpret_cast<const f32x2v*>(p.dz_coef + k * plane + (size_t)n * 32 + 2 * cp);
      n_cz = n;
    }
    const f32x2v ps = cz[2] * p.dz_slope;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2v z = y[k] * cz[0] + cz[1];
      f32x2v sel;
      sel[0] = z[0] > 0.f ? cz[2][0] : ps[0];
      sel[1] = z[1] > 0.f ? cz[2][1] : ps[1];
      e[k] = sel * e[k] + (cz[3] * y[k] + cz[4]);
    }
    float* o = p.dz_out + off;
    *reinterpret_cast<f32x2v*>(o) = e[0];
    *reinterpret_cast<f32x2v*>(o + 32) = e[1];
    *reinterpret_cast<f32x2v*>(o + (size_t)W * 32) = e[2];
    *reinterpret_cast<f32x2v*>(o + (size_t)W * 32 + 32) = e[3];
  };
  auto load_act = [&](int n) {
    if (ACT) {
      const size_t o = (size_t)n * 32 + (tid & 7) * 4;
      ca = *reinterpret_cast<const f32x4*>(p.alpha + o);
      cb = *reinterpret_cast<const f32x4*>(p.beta + o);
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < WQ_PASSES; ++i) {
      if (ACT) pr[i] = act4(pr[i], ca, cb, p.slope, (okm >> i) & 1u);
      if (512 * (i + 1) <= WQ_SLOTS || tid + 512 * i < WQ_SLOTS)
        *reinterpret_cast<f32x4*>(Pb + pp_lds[i]) = pr[i];
    }
  };
  auto transform_x = [&]() {   // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 -1 0 1]
    f32x2v d[4][4];
    for_range<0, 16>([&](auto ic) {
      constexpr int r = decltype(ic)::value / 4, c = decltype(ic)::value % 4;
      d[r][c] = lds_rd64<((r * WQ_PW + c) * WQ_LDA) * 4>(t_srca);
    });
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_wait<0>(d[r][0], d[r][1], d[r][2], d[r][3]);
    f32x2v t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      t[0][c] = d[0][c] - d[2][c];
      t[1][c] = d[1][c] + d[2][c];
      t[2][c] = d[2][c] - d[1][c];
      t[3][c] = d[3][c] - d[1][c];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float* dst = t_dst + (4 * a) * (16 * WQ_VP);
      *reinterpret_cast<f32x2v*>(dst) = t[a][0] - t[a][2];
      *reinterpret_cast<f32x2v*>(dst + 16 * WQ_VP) = t[a][1] + t[a][2];
      *reinterpret_cast<f32x2v*>(dst + 2 * 16 * WQ_VP) = t[a][2] - t[a][1];
      *reinterpret_cast<f32x2v*>(dst + 3 * 16 * WQ_VP) = t[a][3] - t[a][1];
    }
  };
  auto transform_dy = [&](const f32x2v (&e)[4]) {   // E' = G' e G'^T,  G' = [1 0; 1 1; 1 -1; 0 1]
    f32x2v t[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      t[0][b] = e[b];
      t[1][b] = e[b] + e[2 + b];
      t[2][b] = e[b] - e[2 + b];
      t[3][b] = e[2 + b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float* dst = t_dst + (4 * a) * (16 * WQ_VP);
      *reinterpret_cast<f32x2v*>(dst) = t[a][0];
      *reinterpret_cast<f32x2v*>(dst + 16 * WQ_VP) = t[a][0] + t[a][1];
      *reinterpret_cast<f32x2v*>(dst + 2 * 16 * WQ_VP) = t[a][0] - t[a][1];
      *reinterpret_cast<f32x2v*>(dst + 3 * 16 * WQ_VP) = t[a][1];
    }
  };

  // ---- products: wave w owns xi = 2 w and 2 w + 1 for the whole 32 x 32 (ci, co) block:
  //      M[ci][co] += sum over tile slots k = 4 q + fk of V[k][ci] E[k][co].
  // Both fragments are 8-byte reads of a CHANNEL PAIR (lane fn: channels 2 fn, 2 fn + 1 of tile
  // slot fk), so one A read and one B read feed FOUR MFMAs (ci parity x co parity): 16 ds_read_b64
  // per unit and wave where the first version (wave = xi row x co half, 4-byte fragments) issued
  // 48 ds_read_b32, two-way bank-conflicted at the 36-float pitch (rows fk and fk + 1 share 12 of
  // their 16 banks: the 0.33 conflict cycles per active LDS cycle of the round-3 PMC table).  A
  // pitch of 32 floats puts the two tile slots of a 32-lane group on the two halves of the 64
  // banks: conflict-free; the transforms' 8-byte stores (16-lane groups, 32 consecutive floats)
  // are conflict-free at any pitch.  Hand-issued (lds_asm.h): hipcc would fuse the reads of two
  // steps into ds_read2_b64, which the LDS serves at half the rate. ----
  const int fn = lane & 15, fk = lane >> 4;
  const unsigned a_addr = lds_addr(Vs + (2 * wave * 16 + fk) * WQ_VP + 2 * fn);   // + (jj * 16 + 4 q) * VP
  const unsigned b_addr = lds_addr(Es + (2 * wave * 16 + fk) * WQ_VP + 2 * fn);
  f32x4 acc[2][2][2];   // [xi 2 w + jj][ci parity][co parity]; rows 4 fk + r <-> ci = 2 (4 fk + r) + parity
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int d = 0; d < 2; ++d) acc[j][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (t_first < t_end) {
    {
      int n, y0, x0;
      tile_pos(t_first, n, y0, x0);
      load_unit(n, y0, x0);
      load_act(n);
      n_coef = n;
      store_patch();
    }
    __syncthreads();
    for (int tile = t_first; tile < t_end; tile += t_stride) {
      const int nxt = tile + t_stride;
      const bool more = nxt < t_end;
      int n, y0, x0;
      tile_pos(tile, n, y0, x0);
      for_range<0, 4>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        f32x2v ec[4], yc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { ec[k] = en[k]; yc[k] = yn[k]; }   // this unit's dy tile (loaded a unit ago)
        const size_t ec_off = en_off;
        bool have_next = true;
        int nn = n;
        if constexpr (u < 3) {
          load_unit(n, y0 + 4 * ((u + 1) >> 1), x0 + 16 * ((u + 1) & 1));
        } else {
          have_next = more;
          if (more) {   // uniform
            int ny, nx;
            tile_pos(nxt, nn, ny, nx);
            load_unit(nn, ny, nx);
          }
        }
        if (xside) transform_x();   // uniform
        else {
          if (DZ) apply_dz(ec, yc, ec_off, n);
          transform_dy(ec);
        }
        __syncthreads();
        {
          f32x2v fa[2], fb[2];
          auto rd = [&](auto sc) {      // step s = 2 q + jj
            constexpr int st = decltype(sc)::value;
            constexpr int off = (((st & 1) * 16 + 4 * (st >> 1)) * WQ_VP) * 4;
            fa[st & 1] = lds_rd64<off>(a_addr);
            fb[st & 1] = lds_rd64<off>(b_addr);
          };
          rd(std::integral_constant<int, 0>{});
          for_range<0, 8>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            constexpr int jj = st & 1, sl = st & 1;
            if constexpr (st + 1 < 8) rd(std::integral_constant<int, st + 1>{});
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[sl]), "+v"(fb[sl]) : "n"(st + 1 < 8 ? 2 : 0));
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
              for (int d = 0; d < 2; ++d)
                acc[jj][c][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][c], fb[sl][d],
                                                                     acc[jj][c][d], 0, 0, 0);
          });
        }
        if (have_next) {   // uniform
          if (nn != n_coef) { load_act(nn); n_coef = nn; }   // uniform, once per image
          store_patch();
        }
        __syncthreads();
      });
    }
  }

  // ---- epilogue: the halves of G (xi row / column 1 or 2), then dW = A^T M A with
  //      A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1]: every M[xi] goes to LDS once ([xi][ci][co], 64 KB in
  //      the V / E stages - every wave is past its last fragment read behind the unit's second
  //      barrier) and each thread combines the 16 values of two (ci, co) positions ----
  float* const X = Vs;   // [xi 16][ci 32][co 32]
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int xi = 2 * wave + jj, gi = xi >> 2, gj = xi & 3;
    const float sc = ((gi == 1 || gi == 2) ? 0.5f : 1.f) * ((gj == 1 || gj == 2) ? 0.5f : 1.f);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* o = X + (size_t)xi * 1024 + (2 * (4 * fk + r) + c) * 32 + 2 * fn;
        *reinterpret_cast<f32x2v*>(o) = f32x2v{acc[jj][c][0][r] * sc, acc[jj][c][1][r] * sc};
      }
  }
  __syncthreads();
  float* const slab = p.partial + (size_t)blockIdx.x * 9 * 1024;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int rc = tid + 512 * h;
    float t[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float m0 = X[(4 * i + 0) * 1024 + rc], m1 = X[(4 * i + 1) * 1024 + rc];
      const float m2 = X[(4 * i + 2) * 1024 + rc], m3 = X[(4 * i + 3) * 1024 + rc];
      t[i][0] = m0 + m1 + m2;
      t[i][1] = m1 - m2;
      t[i][2] = m1 + m2 + m3;
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      slab[(0 * 3 + v) * 1024 + rc] = t[0][v] + t[1][v] + t[2][v];
      slab[(1 * 3 + v) * 1024 + rc] = t[1][v] - t[2][v];
      slab[(2 * 3 + v) * 1024 + rc] = t[1][v] + t[2][v] + t[3][v];
    }
  }
}

// shapes it takes: the layer is ONE 32 x 32 channel tile, stride 1, whole 8 x 32-pixel tiles,
// enough of them for one workgroup per CU
// (unet_set_c32_winograd: 1 = when there is a tile for every CU, 2 = always, 0 = never)
bool wgrad_wino32_ok(int N, int H, int W, int Cx, int Cout, int stride) {
  const int f = unet_conv::c32_winograd_flag();
  if (!(f && stride == 1 && Cx == 32 && Cout == 32 && H % 8 == 0 && W % 32 == 0 &&
        (long long)N * H * W * 128 < (1LL << 31)))
    return false;
  return f == 2 || (long long)N * (H / 8) * (W / 32) >= 256;
}
WgradPlan make_plan_wino32(int N, int H, int W) {
  WgradPlan pl{};
  const long long t = (long long)N * (H / 8) * (W / 32);
  pl.nw = 8; pl.ci_t = pl.co_t = 32; pl.npp = 1; pl.sps = 1; pl.S = 16;
  pl.split = (int)(t < 256 ? t : 256);
  pl.total_segs = (int)t; pl.segs_per_block = 1; pl.segs_per_row = W / 32;
  const size_t E = (size_t)9 * 32 * 32;
  pl.ws_floats = (size_t)pl.split * E + 2 * (size_t)ceil_div(pl.split, kSlabChunk) * E;
  return pl;
}
int launch_wgrad_wino32(const WgradParams& p, hipStream_t stream) {
  const int ntiles = p.N * (p.H / 8) * (p.W / 32);
  const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);
  if (p.dz_y) {   // (the fused pipeline: the x operand is always activated on load)
    auto kern = conv_wgrad_wino32_kernel<true, true>;
    UNET_SET_DYN_LDS(kern, WQ_LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), WQ_LDS, stream, p, ntiles);
  } else if (p.alpha) {
    auto kern = conv_wgrad_wino32_kernel<true, false>;
    UNET_SET_DYN_LDS(kern, WQ_LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), WQ_LDS, stream, p, ntiles);
  } else {
    auto kern = conv_wgrad_wino32_kernel<false, false>;
    UNET_SET_DYN_LDS(kern, WQ_LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), WQ_LDS, stream, p, ntiles);
  }
  UNET_CHECK_LAUNCH("conv_wgrad_wino32");
  return UNET_OK;
}

int launch_wgrad_wino(const WgradParams& p, hipStream_t stream) {
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  if (p.alpha) {
    auto kern = conv_wgrad_wino_kernel<true>;
    UNET_SET_DYN_LDS(kern, WW_LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), WW_LDS, stream, p);
  } else {
    auto kern = conv_wgrad_wino_kernel<false>;
    UNET_SET_DYN_LDS(kern, WW_LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), WW_LDS, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_wgrad_wino");
  return UNET_OK;
}

// shapes the Winograd weight gradient tiles: stride 1, 64-wide channel tiles, an image that
// splits into chunks of 8 tiles (16 x 2 pixels), enough chunks to split over workgroups
bool wgrad_wino_ok(int N, int H, int W, int Cx, int Cout, int stride) {
  if (stride != 1 || Cx % 64 || Cout % 64 || H % 2 || W % 16) return false;
  const long long chunks = (long long)N * (H / 2) * (W / 16);
  return chunks >= 32 && chunks < (1LL << 30);   // >= 8 chunks for each of >= 4 pixel splits
}

// plan: K chunks (`segments`) split over ~256 / tiles workgroups per channel tile
WgradPlan make_plan_wino(int N, int H, int W, int Cx, int Cout) {
  WgradPlan pl{};
  pl.nw = 8; pl.ci_t = pl.co_t = 64; pl.npp = 2; pl.sps = 1; pl.S = 16;
  pl.segs_per_row = W / 16;
  pl.total_segs = N * (H / 2) * (W / 16);
  const int tiles = (Cx / 64) * (Cout / 64);
  int split = ceil_div(256, tiles);
  const int max_split = ceil_div(pl.total_segs, 8);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  pl.segs_per_block = ceil_div(pl.total_segs, split);
  pl.split = ceil_div(pl.total_segs, pl.segs_per_block);
  const size_t E = (size_t)9 * Cx * Cout;
  pl.ws_floats = (size_t)pl.split * E + 2 * (size_t)ceil_div(pl.split, kSlabChunk) * E;
  return pl;
}

template <int CI_T, int CO_T, int S, int STRIDE, bool ACT, typename TS>
int launch_wgrad_t(const WgradParams& p, hipStream_t stream);

template <int CI_T, int CO_T, int S, int STRIDE>
int launch_wgrad(const WgradParams& p, hipStream_t stream) {
  if (p.b16) {
    if (p.alpha) return launch_wgrad_t<CI_T, CO_T, S, STRIDE, true, __bf16>(p, stream);
    return launch_wgrad_t<CI_T, CO_T, S, STRIDE, false, __bf16>(p, stream);
  }
  if (p.alpha) return launch_wgrad_t<CI_T, CO_T, S, STRIDE, true, float>(p, stream);
  return launch_wgrad_t<CI_T, CO_T, S, STRIDE, false, float>(p, stream);
}

template <int CI_T, int CO_T, int S, int STRIDE, bool ACT, typename TS>
int launch_wgrad_t(const WgradParams& p, hipStream_t stream) {
  constexpr int PW = (S - 1) * STRIDE + 3;
  constexpr size_t stages = 2 * (size_t)(3 * PW * CI_T + S * CO_T) * sizeof(float);
  constexpr bool merge = true;
  constexpr size_t lds = (merge && stages < kWgradMergeLds4) ? kWgradMergeLds4 : stages;
  constexpr int NT = 256;
  auto kern = conv_wgrad_kernel<CI_T, CO_T, S, STRIDE, ACT, TS, TS>;
  UNET_SET_DYN_LDS(kern, lds);
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_wgrad");
  return UNET_OK;
}

template <int CI_T, int CO_T, int S, int NPL = 1, bool SB = false>
int launch_wgrad_bf16(const WgradParams& p, hipStream_t stream) {
  constexpr int PW = S + 2;
  constexpr size_t stages = (SB ? 1 : 2) * NPL * (size_t)(3 * PW * CI_T + S * CO_T) * sizeof(__bf16);
  constexpr bool merge = true;
  constexpr size_t lds = (merge && stages < kWgradMergeLds4) ? kWgradMergeLds4 : stages;
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  if constexpr (NPL == 1 && !SB) {
    if (p.b16) {   // bf16 tensors in HBM, operand activated on load when p.alpha is set
      if (p.alpha) {
        auto kern = conv_wgrad_bf16_kernel<CI_T, CO_T, S, 1, false, __bf16, __bf16, true>;
        UNET_SET_DYN_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
      } else {
        auto kern = conv_wgrad_bf16_kernel<CI_T, CO_T, S, 1, false, __bf16, __bf16, false>;
        UNET_SET_DYN_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
      }
      UNET_CHECK_LAUNCH("conv_wgrad_bf16(b16)");
      return UNET_OK;
    }
  }
  if constexpr (NPL == 3) {
    if (p.alpha) {   // split mode of the fused pipeline: activation on load, then the split
      auto kern = conv_wgrad_bf16_kernel<CI_T, CO_T, S, 3, SB, float, float, true>;
      UNET_SET_DYN_LDS(kern, lds);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
      UNET_CHECK_LAUNCH("conv_wgrad_bf16x3(act)");
      return UNET_OK;
    }
  }
  auto kern = conv_wgrad_bf16_kernel<CI_T, CO_T, S, NPL, SB>;
  UNET_SET_DYN_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_wgrad_bf16");
  return UNET_OK;
}

// mixed-precision pipeline, stride 1, bf16 tensors: the row-ring form (plan: make_plan)
bool wgrad_ring_ok(const WgradParams& p, int rg) {
  return rg >= 1 && p.b16 &&
         p.segs_per_block % (kRingDepth * rg) == 0 && p.Ho % p.segs_per_block == 0 &&
         p.total_segs % p.segs_per_block == 0 && p.Cx % 8 == 0 && p.Cout % 8 == 0;
}
template <int CI_T, int CO_T, int S, int RG, int STRIDE>
int launch_wgrad_b16_ring_t(const WgradParams& p, hipStream_t stream) {
  constexpr size_t lds = RG == 2 ? kWgradMergeLds8 : kWgradMergeLds4;
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  if (p.alpha) {
    auto kern = conv_wgrad_b16_ring_kernel<CI_T, CO_T, S, true, kRingDepth, RG, STRIDE>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * RG), lds, stream, p);
  } else {
    auto kern = conv_wgrad_b16_ring_kernel<CI_T, CO_T, S, false, kRingDepth, RG, STRIDE>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * RG), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_wgrad_b16_ring");
  return UNET_OK;
}
template <int CI_T, int CO_T, int S, int STRIDE = 1>
int launch_wgrad_b16_ring(const WgradParams& p, int rg, hipStream_t stream) {
  return rg == 2 ? launch_wgrad_b16_ring_t<CI_T, CO_T, S, 2, STRIDE>(p, stream)
                 : launch_wgrad_b16_ring_t<CI_T, CO_T, S, 1, STRIDE>(p, stream);
}

// mixed-precision pipeline, stride 2: bf16 tensors, bf16 matrix cores
template <int CI_T, int CO_T, int S>
int launch_wgrad_b16_s2(const WgradParams& p, hipStream_t stream) {
  constexpr int PW = (S - 1) * 2 + 3;
  constexpr size_t stages = 2 * (size_t)(3 * PW * CI_T + S * CO_T) * sizeof(__bf16);
  constexpr bool merge = true;
  constexpr size_t lds = (merge && stages < kWgradMergeLds4) ? kWgradMergeLds4 : stages;
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  if (p.alpha) {
    auto kern = conv_wgrad_bf16_kernel<CI_T, CO_T, S, 1, false, __bf16, __bf16, true, 2>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  } else {
    auto kern = conv_wgrad_bf16_kernel<CI_T, CO_T, S, 1, false, __bf16, __bf16, false, 2>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_wgrad_bf16(b16, stride 2)");
  return UNET_OK;
}

// the 8-wave form (fp32 tensors)
template <int CI_T, int CO_T, int S>
int launch_wgrad_taps8(const WgradParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)S * (CI_T + 9 * CO_T) * sizeof(float);
  static_assert(lds <= 160 * 1024 && lds >= 96 * 1024, "8-wave stages / merge space");
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  if (p.alpha) {
    auto kern = conv_wgrad_taps_kernel<CI_T, CO_T, S, true, float, float, 8>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
  } else {
    auto kern = conv_wgrad_taps_kernel<CI_T, CO_T, S, false, float, float, 8>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_wgrad_taps(8 waves)");
  return UNET_OK;
}

// bf16 tensors, bf16 matrix cores (64 x 64 tiles)
template <int CI_T, int CO_T, int S>
int launch_wgrad_taps_b16(const WgradParams& p, hipStream_t stream) {
  constexpr size_t stages = 2 * (size_t)S * (CI_T + 9 * CO_T) * sizeof(__bf16);
  constexpr size_t lds = stages > kWgradMergeLds4 ? stages : kWgradMergeLds4;
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  if (p.alpha) {
    auto kern = conv_wgrad_taps_b16_kernel<CI_T, CO_T, S, true>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  } else {
    auto kern = conv_wgrad_taps_b16_kernel<CI_T, CO_T, S, false>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_wgrad_taps_b16");
  return UNET_OK;
}

template <int CI_T, int CO_T, int S>
int launch_wgrad_taps(const WgradParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)S * (CI_T + 9 * CO_T) * sizeof(float);
  const unsigned grid = (unsigned)(p.split * p.ci_tiles * p.co_tiles);
  if (p.b16) {   // bf16 tensors in HBM (fp32 matrix cores: 1/4 of the FLOPs already)
    if (p.alpha) {
      auto kern = conv_wgrad_taps_kernel<CI_T, CO_T, S, true, __bf16, __bf16>;
      UNET_SET_DYN_LDS(kern, lds);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    } else {
      auto kern = conv_wgrad_taps_kernel<CI_T, CO_T, S, false, __bf16, __bf16>;
      UNET_SET_DYN_LDS(kern, lds);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    }
  } else if (p.alpha) {
    auto kern = conv_wgrad_taps_kernel<CI_T, CO_T, S, true>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  } else {
    auto kern = conv_wgrad_taps_kernel<CI_T, CO_T, S, false>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_wgrad_taps");
  return UNET_OK;
}

// plan of the low-resolution tap GEMM: Q pixels, (Cx x Cout) channel tiles
// wide: the 8-wave kernel (fp32 tensors)
WgradPlan make_plan_taps(long long Q, int Cx, int Cout, bool wide, bool b16 = false) {
  WgradPlan pl{};
  pl.ci_t = (Cx % 64 == 0) ? 64 : 32;
  pl.co_t = (Cout % 64 == 0) ? 64 : 32;
  if (pl.ci_t == 64 && pl.co_t == 32) pl.ci_t = 32;  // instantiated: 32x32, 32x64, 64x64
  const int nsb = (pl.ci_t / 32) * (pl.co_t / 32);
  pl.nw = wide ? 8 : 4;
  pl.npp = pl.nw / nsb;
  pl.sps = 1;
  pl.S = (nsb == 1 ? 32 : 16) * (wide ? 2 : 1);
  if (b16 && nsb == 1) pl.S = 64;   // conv_wgrad_taps_b16_kernel: four k-groups, one per wave
  pl.segs_per_row = 0;
  pl.total_segs = (int)ceil_div64(Q, pl.S);
  const int tiles = (Cx / pl.ci_t) * (Cout / pl.co_t);
  int split = ceil_div(wide ? 256 : 512, tiles);
  const int max_split = ceil_div(pl.total_segs, 4);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  pl.segs_per_block = ceil_div(pl.total_segs, split);
  pl.split = ceil_div(pl.total_segs, pl.segs_per_block);
  const size_t E = (size_t)9 * Cx * Cout;
  const int slabs = pl.split * pl.sps;
  pl.ws_floats = (size_t)slabs * E + 2 * (size_t)ceil_div(slabs, kSlabChunk) * E;
  return pl;
}

// UNET_WGRAD_RING=0: the segment kernels instead of the row-ring form (A/B measurements)
bool ring_off() {
  static const bool off = [] { const char* e = getenv("UNET_WGRAD_RING"); return e && e[0] == '0'; }();
  return off;
}
// kernel instantiation for a plan (tile, segment length, stride, operand mode)
int launch_wgrad_plan(const WgradParams& p, const WgradPlan& pl, int stride, int prec,
                      hipStream_t stream) {
  if (pl.nw == 8) {
    if (pl.ci_t == 32 && pl.co_t == 32)
      return stride == 1 ? launch_wgrad8_s<32, 32, 1>(p, pl.S, stream)
                         : launch_wgrad8_s<32, 32, 2>(p, pl.S, stream);
    if (pl.ci_t == 32)
      return stride == 1 ? launch_wgrad8_s<32, 64, 1>(p, pl.S, stream)
                         : launch_wgrad8_s<32, 64, 2>(p, pl.S, stream);
    return stride == 1 ? launch_wgrad8_s<64, 64, 1>(p, pl.S, stream)
                       : launch_wgrad8_s<64, 64, 2>(p, pl.S, stream);
  }
  // bf16 operands: stride 1 and a segment that splits into whole 16-pixel k-groups per wave
  const bool use_bf16 = prec != 0 && stride == 1 && (pl.S / 16) % pl.npp == 0 && pl.S >= 16;
  if (use_bf16 && prec == 3) {
    if (pl.ci_t == 32 && pl.co_t == 32) return launch_wgrad_bf16<32, 32, 64, 3, true>(p, stream);
    if (pl.ci_t == 32) return launch_wgrad_bf16<32, 64, 32, 3>(p, stream);
    return launch_wgrad_bf16<64, 64, 16, 3>(p, stream);
  }
  if (p.b16 && prec == 1 && stride == 2 && (pl.S / 16) % pl.npp == 0 && wgrad_ring_ok(p, pl.rg) &&
      !ring_off()) {   // the row-ring form with two input rows a step
    if (pl.ci_t == 64 && pl.S == 16) return launch_wgrad_b16_ring<64, 64, 16, 2>(p, pl.rg, stream);
    if (pl.ci_t == 32 && pl.co_t == 64 && pl.S == 32) return launch_wgrad_b16_ring<32, 64, 32, 2>(p, pl.rg, stream);
  }
  if (p.b16 && prec == 1 && stride == 2 && (pl.S / 16) % pl.npp == 0) {
    if (pl.ci_t == 64 && pl.S == 16) return launch_wgrad_b16_s2<64, 64, 16>(p, stream);
    if (pl.ci_t == 32 && pl.co_t == 64 && pl.S == 32) return launch_wgrad_b16_s2<32, 64, 32>(p, stream);
  }
  if (use_bf16 && prec == 1 && wgrad_ring_ok(p, pl.rg) && !ring_off()) {
    if (pl.ci_t == 32 && pl.co_t == 32 && pl.S == 64) return launch_wgrad_b16_ring<32, 32, 64>(p, pl.rg, stream);
    if (pl.ci_t == 32 && pl.co_t == 64 && pl.S == 32) return launch_wgrad_b16_ring<32, 64, 32>(p, pl.rg, stream);
    if (pl.ci_t == 64 && pl.S == 32) return launch_wgrad_b16_ring<64, 64, 32>(p, pl.rg, stream);
    if (pl.ci_t == 64 && pl.S == 16) return launch_wgrad_b16_ring<64, 64, 16>(p, pl.rg, stream);
  }
  if (use_bf16 && prec == 1) {
    if (pl.ci_t == 32 && pl.co_t == 32) return launch_wgrad_bf16<32, 32, 64>(p, stream);
    if (pl.ci_t == 32) return launch_wgrad_bf16<32, 64, 32>(p, stream);
    return pl.S == 32 ? launch_wgrad_bf16<64, 64, 32>(p, stream)
                      : launch_wgrad_bf16<64, 64, 16>(p, stream);
  }
  if (pl.ci_t == 32 && pl.co_t == 32) {
    if (stride == 1)
      return pl.S == 64 ? launch_wgrad<32, 32, 64, 1>(p, stream)
           : pl.S == 32 ? launch_wgrad<32, 32, 32, 1>(p, stream)
                        : launch_wgrad<32, 32, 16, 1>(p, stream);
    return pl.S == 32 ? launch_wgrad<32, 32, 32, 2>(p, stream)
                      : launch_wgrad<32, 32, 16, 2>(p, stream);
  }
  if (pl.ci_t == 32 && pl.co_t == 64) {
    if (stride == 1)
      return pl.S == 32 ? launch_wgrad<32, 64, 32, 1>(p, stream)
                        : launch_wgrad<32, 64, 16, 1>(p, stream);
    return pl.S == 32 ? launch_wgrad<32, 64, 32, 2>(p, stream)
                      : launch_wgrad<32, 64, 16, 2>(p, stream);
  }
  if (stride == 1)
    return pl.S == 32 ? launch_wgrad<64, 64, 32, 1>(p, stream)
                      : launch_wgrad<64, 64, 16, 1>(p, stream);
  return launch_wgrad<64, 64, 16, 2>(p, stream);
}

// x and dy are addressed through 2 GiB buffer descriptors: a larger batch is processed in
// chunks of `nmax` images whose slabs are reduced together.
int wgrad_batch_chunk(int N, int H, int W, int Cx, int Cout, int stride) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long long a = (long long)H * W * Cx * 4, b = (long long)Ho * Wo * Cout * 4;
  return unet_conv::batch_chunk(N, a > b ? a : b);
}

size_t wgrad_ws_floats(int N, int H, int W, int Cx, int Cout, int stride, int prec,
                       bool wide = false) {
  if (Cx == 3) return make_plan(N, H, W, Cx, Cout, stride, prec).ws_floats;
  const int nmax = wgrad_batch_chunk(N, H, W, Cx, Cout, stride);
  if (nmax < 1) return 0;
  const size_t E = (size_t)9 * Cx * Cout;
  size_t slabs = 0;
  for (int nb = 0; nb < N; nb += nmax) {
    const WgradPlan pl = make_plan(N - nb < nmax ? N - nb : nmax, H, W, Cx, Cout, stride, prec,
                                   wide);
    slabs += (size_t)pl.split * pl.sps;
  }
  return slabs * E + 2 * (size_t)ceil_div((int)slabs, kSlabChunk) * E;
}

// db[c] = sum over pixels of dy[.][c]: 64 row chunks per 32-channel group, then the chunk sums
// (fixed order => deterministic).  out[chunk][C] when chunks > 1.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy,
                                                        float* __restrict__ out, long long M,
                                                        int C, long long rows_per_chunk) {
  __shared__ float red[8][33];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  const int r = threadIdx.x >> 5;
  const long long m0 = (long long)blockIdx.y * rows_per_chunk;
  const long long m1 = m0 + rows_per_chunk < M ? m0 + rows_per_chunk : M;
  float s = 0.f;
  for (long long m = m0 + r; m < m1; m += 8) s += dy[(size_t)m * C + c];
  red[r][threadIdx.x & 31] = s;
  __syncthreads();
  if (r == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
    out[(size_t)blockIdx.y * C + c] = t;
  }
}

}  // namespace

extern "C" size_t unet_conv3x3_bwd_weight_workspace_bytes(int N, int H, int W, int Cx, int Cout,
                                                          int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cx <= 0 || Cout <= 0) return 0;
  // one size for every operand mode (the bf16x3 plan uses shorter segments on 64x64 tiles)
  const size_t a = wgrad_ws_floats(N, H, W, Cx, Cout, stride, 0);
  const size_t b = wgrad_ws_floats(N, H, W, Cx, Cout, stride, 3);
  const size_t c = wgrad_ws_floats(N, H, W, Cx, Cout, stride, 0, true);
  const size_t e = wgrad_ws_floats(N, H, W, Cx, Cout, stride, 1);   // bf16: longer segments
  size_t m = a > b ? a : b;
  if (c > m) m = c;
  if (e > m) m = e;
  if (Cx != 3 && wgrad_wino_ok(N, H, W, Cx, Cout, stride)) {
    const size_t d = make_plan_wino(N, H, W, Cx, Cout).ws_floats;
    if (d > m) m = d;
  }
  if (stride == 1 && Cx == 32 && Cout == 32 && H % 8 == 0 && W % 32 == 0) {   // (whatever the switch says)
    const size_t d = make_plan_wino32(N, H, W).ws_floats;
    if (d > m) m = d;
  }
  return m * sizeof(float);
}

// 1 when the fp32 entry points (unet_conv3x3_bwd_weight, unet_conv_in_bwd_weight with ksize 3)
// run this shape on the Winograd F(3x3,2x2) kernel (16/36 of the direct matrix FLOPs)
extern "C" int unet_conv3x3_bwd_weight_is_winograd(int N, int H, int W, int Cx, int Cout,
                                                   int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cx <= 3 || Cout <= 0) return 0;
  return ((wgrad_wino_ok(N, H, W, Cx, Cout, stride) || wgrad_wino32_ok(N, H, W, Cx, Cout, stride)) &&
          wgrad_batch_chunk(N, H, W, Cx, Cout, stride) >= N) ? 1 : 0;
}

static int conv_bwd_weight_impl(const float* x, int Cx, const float* dy, float* dw_oihw,
                                int ci_offset, int Cin_total, float* db, void* workspace,
                                size_t workspace_bytes, int N, int H, int W, int Cout, int stride,
                                bool center_only, hipStream_t stream, int prec = 0,
                                const float* act_alpha = nullptr, const float* act_beta = nullptr,
                                float slope = 0.f, const unsigned char* x_u8 = nullptr,
                                const float* u8_mean_std = nullptr, int b16 = 0,
                                const WgradParams* dz = nullptr) {
  // b16: x (except the RGB image) and dy are bf16 tensors; prec is then 1 (bf16 matrix cores
  // for the stride-1 layers, fp32 matrix cores on bf16 storage for the rest)
  const long long es = b16 ? 2 : 4;
  UNET_REQUIRE((x || x_u8) && dy && dw_oihw && workspace, "conv3x3_bwd_weight: null pointer");
  UNET_REQUIRE(stride == 1 || stride == 2, "conv3x3_bwd_weight: stride %d unsupported", stride);
  UNET_REQUIRE(Cout > 0 && Cout % 32 == 0, "conv3x3_bwd_weight: Cout %d not a multiple of 32", Cout);
  UNET_REQUIRE(Cx == 3 || (Cx > 0 && Cx % 32 == 0), "conv3x3_bwd_weight: Cx %d unsupported", Cx);
  UNET_REQUIRE(ci_offset >= 0 && ci_offset + Cx <= Cin_total, "conv3x3_bwd_weight: bad ci slice");
  const int pprec = (prec == 3 && stride == 1) ? 3 : ((prec == 1 && stride == 1) ? 1 : 0);
  // fp32 tensors on the fp32 matrix cores: the 8-wave kernel (parts merged in LDS)
  // fp32 tensors on the fp32 matrix cores: with >= 4 channel tiles the 8-wave kernel (one
  // workgroup per CU, <= 64 pixel splits, 16 pixel pairs per barrier); the 32 / 64-channel layers
  // keep two independent 4-wave workgroups per CU (measured 4-6 % faster there: their 128..256
  // pixel splits leave the 8-wave form few MFMAs per barrier and long reductions)
  const bool wide = prec == 0 && !b16 && Cx != 3 && wgrad_tiles(Cx, Cout) >= 4;
  // fp32 tensors, stride-1 3x3 layers with 64-wide channel tiles: the Winograd F(3x3,2x2) form
  const bool wino = prec == 0 && !b16 && !center_only && Cx != 3 &&
                    wgrad_wino_ok(N, H, W, Cx, Cout, stride) &&
                    wgrad_batch_chunk(N, H, W, Cx, Cout, stride) >= N;
  // ... and the 32 -> 32 channel layers their own (one workgroup per CU keeps the whole M)
  const bool wino32 = prec == 0 && !b16 && !center_only && wgrad_wino32_ok(N, H, W, Cx, Cout, stride) &&
                      wgrad_batch_chunk(N, H, W, Cx, Cout, stride) >= N;
  // bf16 tensors, stride 1: the eight-wave row-ring kernel where the plan finds rows for it
  static const bool ring8_off = [] { const char* e = getenv("UNET_WGRAD_RING8"); return e && e[0] == '0'; }();
  // (64 x 64 channel tiles only: measured per layer - 32 x 32 tiles, enc0 / dec4 at 512 x 512,
  // lose 20 % in the eight-wave form, the 64- and 128-channel layers gain 15 %, deeper ones +-0)
  const bool ring_s2 = b16 && prec == 1 && stride == 2;
  const bool ring8 = b16 && (pprec == 1 || ring_s2) && !ring8_off && Cx % 64 == 0 && Cout % 64 == 0;
  const WgradPlan pl = wino32 ? make_plan_wino32(N, H, W)
                       : wino ? make_plan_wino(N, H, W, Cx, Cout)
                              : make_plan(N, H, W, Cx, Cout, stride, pprec, wide, ring8, ring_s2);
  const size_t need = ((wino || wino32) ? pl.ws_floats
                            : wgrad_ws_floats(N, H, W, Cx, Cout, stride, pprec, wide)) * sizeof(float);
  if (workspace_bytes < need || need == 0) {
    unet_set_error("conv3x3_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return UNET_E_WORKSPACE;
  }
  float* ws = reinterpret_cast<float*>(workspace);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (pl.stem) {
    UNET_REQUIRE(stride == 1 && Cin_total == 3 && ci_offset == 0, "conv3x3_bwd_weight: stem shape");
    dim3 grid(pl.stem_blocks, Cout / 32);
    if (x_u8) {
      UNET_REQUIRE(W % SW_PIX == 0, "stem_u8_bwd_weight: needs W %% %d == 0", SW_PIX);
      StemNormW nm;
      for (int c = 0; c < 3; ++c) { nm.mean[c] = u8_mean_std[c]; nm.std[c] = u8_mean_std[3 + c]; }
      if (b16)
        hipLaunchKernelGGL((conv_stem_wgrad_rows_kernel<unsigned char, __bf16>), grid, dim3(256), 0,
                           stream, x_u8, reinterpret_cast<const __bf16*>(dy), ws, N, H, W, Cout,
                           pl.stem_spb, pl.stem_stages, nm);
      else
        hipLaunchKernelGGL((conv_stem_wgrad_rows_kernel<unsigned char, float>), grid, dim3(256), 0,
                           stream, x_u8, dy, ws, N, H, W, Cout, pl.stem_spb, pl.stem_stages, nm);
    } else if (W % SW_PIX == 0) {  // a 128-pixel stage never straddles image rows: raw-row form
      if (b16)
        hipLaunchKernelGGL((conv_stem_wgrad_rows_kernel<float, __bf16>), grid, dim3(256), 0, stream,
                           x, reinterpret_cast<const __bf16*>(dy), ws, N, H, W, Cout, pl.stem_spb,
                           pl.stem_stages, StemNormW{});
      else
        hipLaunchKernelGGL((conv_stem_wgrad_rows_kernel<float, float>), grid, dim3(256), 0, stream,
                           x, dy, ws, N, H, W, Cout, pl.stem_spb, pl.stem_stages, StemNormW{});
    } else if (b16)
      hipLaunchKernelGGL(conv_stem_wgrad_kernel<__bf16>, grid, dim3(256), 0, stream, x,
                         reinterpret_cast<const __bf16*>(dy), ws, N, H, W, Cout, pl.stem_spb,
                         pl.stem_stages);
    else
      hipLaunchKernelGGL(conv_stem_wgrad_kernel<float>, grid, dim3(256), 0, stream, x, dy, ws, N, H,
                         W, Cout, pl.stem_spb, pl.stem_stages);
    UNET_CHECK_LAUNCH("conv_stem_wgrad");
    {
      const int rc = emit_wgrad_reduction(ws, pl.stem_blocks, (size_t)27 * Cout, dw_oihw, 3, Cout, 0, 3,
                                          false, true, stream);
      if (rc != UNET_OK) return rc;
    }
  } else {
    UNET_REQUIRE(!act_alpha || (act_beta && (prec == 0 || prec == 3 || b16)),
                 "conv_bwd_weight: activation on load needs the fp32, split or bf16-storage path");
    const size_t E = (size_t)9 * Cx * Cout;
    const int nmax = wgrad_batch_chunk(N, H, W, Cx, Cout, stride);
    int nslab = 0;
    for (int nb = 0; nb < N; nb += nmax) {   // one pass unless a tensor exceeds 2 GiB
      const int nc = N - nb < nmax ? N - nb : nmax;
      const WgradPlan pc = (wino || wino32) ? pl : make_plan(nc, H, W, Cx, Cout, stride, pprec, wide, ring8, ring_s2);
      WgradParams p{};
      p.x = reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) +
                                           (size_t)nb * H * W * Cx * es);
      p.dy = reinterpret_cast<const float*>(reinterpret_cast<const char*>(dy) +
                                            (size_t)nb * Ho * Wo * Cout * es);
      p.b16 = b16;
      p.partial = ws + (size_t)nslab * E; p.Cx = Cx; p.Cout = Cout;
      p.N = nc; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo;
      p.segs_per_row = pc.segs_per_row; p.total_segs = pc.total_segs;
      p.segs_per_block = pc.segs_per_block; p.split = pc.split;
      p.ci_tiles = Cx / pc.ci_t; p.co_tiles = Cout / pc.co_t;
      p.x_bytes = (unsigned)((long long)nc * H * W * Cx * es);
      p.dy_bytes = (unsigned)((long long)nc * Ho * Wo * Cout * es);
      p.alpha = act_alpha ? act_alpha + (size_t)nb * Cx : nullptr;
      p.beta = act_alpha ? act_beta + (size_t)nb * Cx : nullptr;
      p.slope = slope;
      if (dz) {   // the InstanceNorm backward applied by the dy side of the Winograd kernel
        UNET_REQUIRE(wino32 && act_alpha, "conv_in_bwd_weight_dz: shape not taken by conv_wgrad_wino32_kernel");
        p.dz_y = dz->dz_y; p.dz_coef = dz->dz_coef; p.dz_out = dz->dz_out; p.dz_sums = dz->dz_sums;
        p.dz_gamma = dz->dz_gamma; p.dz_rstd = dz->dz_rstd; p.dz_dgamma = dz->dz_dgamma;
        p.dz_dbeta = dz->dz_dbeta; p.dz_dbias = dz->dz_dbias; p.dz_slope = dz->dz_slope;
      }
      const int rc = wino32 ? launch_wgrad_wino32(p, stream)
                     : wino ? launch_wgrad_wino(p, stream) : launch_wgrad_plan(p, pc, stride, prec, stream);
      if (rc != UNET_OK) return rc;
      nslab += pc.split * pc.sps;
    }
    {
      const int rc = emit_wgrad_reduction(ws, nslab, E, dw_oihw, Cx, Cout, ci_offset, Cin_total,
                                          center_only, false, stream);
      if (rc != UNET_OK) return rc;
    }
  }
  UNET_REQUIRE(!(db && b16), "conv_bwd_weight: db is not produced on the bf16-storage path");
  if (db) {
    UNET_REQUIRE(!reduce_queue().on, "conv_bwd_weight: the bias gradient reuses the slab workspace "
                                     "and cannot be combined with deferred reductions");
    // the slab workspace is free again at this point of the stream; reuse its head as scratch
    const long long M = (long long)N * Ho * Wo;
    const int chunks = M >= 4096 ? 64 : 1;
    const long long rpc = ceil_div64(M, chunks);
    if (chunks == 1) {
      hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout / 32, 1), dim3(256), 0, stream, dy, db, M,
                         Cout, rpc);
    } else {
      hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout / 32, chunks), dim3(256), 0, stream, dy, ws, M,
                         Cout, rpc);
      UNET_CHECK_LAUNCH("bias_grad(stage)");
      hipLaunchKernelGGL(bias_grad_kernel, dim3(Cout / 32, 1), dim3(256), 0, stream, ws, db,
                         (long long)chunks, Cout, (long long)chunks);
    }
    UNET_CHECK_LAUNCH("bias_grad");
  }
  return UNET_OK;
}

extern "C" int unet_conv3x3_bwd_weight(const float* x, int Cx, const float* dy, float* dw_oihw,
                                       int ci_offset, int Cin_total, float* db, void* workspace,
                                       size_t workspace_bytes, int N, int H, int W, int Cout,
                                       int stride, unet_stream_t stream) {
  return conv_bwd_weight_impl(x, Cx, dy, dw_oihw, ci_offset, Cin_total, db, workspace,
                              workspace_bytes, N, H, W, Cout, stride, false, (hipStream_t)stream);
}

// 1x1 weight gradient dw[Cout][Cin_total] (columns ci_offset .. +Cx).  Runs the 3x3 kernel and
// keeps its centre tap (the fusion layer sits at 1/32 resolution: 2,048 pixels at bs 8, so the
// 9x matrix work is ~0.2 ms); workspace as unet_conv3x3_bwd_weight_workspace_bytes(.., 1).
extern "C" int unet_conv1x1_bwd_weight(const float* x, int Cx, const float* dy, float* dw,
                                       int ci_offset, int Cin_total, void* workspace,
                                       size_t workspace_bytes, int N, int H, int W, int Cout,
                                       unet_stream_t stream) {
  UNET_REQUIRE(Cx % 32 == 0, "conv1x1_bwd_weight: Cx %d must be a multiple of 32", Cx);
  return conv_bwd_weight_impl(x, Cx, dy, dw, ci_offset, Cin_total, nullptr, workspace,
                              workspace_bytes, N, H, W, Cout, 1, true, (hipStream_t)stream);
}

extern "C" int unet_conv3x3_bwd_weight_bf16(const float* x, int Cx, const float* dy,
                                            float* dw_oihw, int ci_offset, int Cin_total,
                                            float* db, void* workspace, size_t workspace_bytes,
                                            int N, int H, int W, int Cout, int stride,
                                            unet_stream_t stream) {
  return conv_bwd_weight_impl(x, Cx, dy, dw_oihw, ci_offset, Cin_total, db, workspace,
                              workspace_bytes, N, H, W, Cout, stride, false, (hipStream_t)stream,
                              1);
}

extern "C" int unet_conv3x3_bwd_weight_bf16x3(const float* x, int Cx, const float* dy,
                                              float* dw_oihw, int ci_offset, int Cin_total,
                                              float* db, void* workspace, size_t workspace_bytes,
                                              int N, int H, int W, int Cout, int stride,
                                              unet_stream_t stream) {
  return conv_bwd_weight_impl(x, Cx, dy, dw_oihw, ci_offset, Cin_total, db, workspace,
                              workspace_bytes, N, H, W, Cout, stride, false, (hipStream_t)stream,
                              3);
}

extern "C" int unet_conv_in_bwd_weight(const unet_act_src* x, float slope, const float* dy,
                                       float* dw_oihw, int ci_offset, int Cin_total, int ksize,
                                       int stride, void* workspace, size_t workspace_bytes, int N,
                                       int H, int W, int Cout, unet_stream_t stream) {
  UNET_REQUIRE(x && x->x, "conv_in_bwd_weight: null source");
  UNET_REQUIRE(ksize == 3 || (ksize == 1 && stride == 1),
               "conv_in_bwd_weight: kernel %d / stride %d unsupported", ksize, stride);
  UNET_REQUIRE(x->C == 3 ? !x->alpha : x->C % 32 == 0,
               "conv_in_bwd_weight: Cx %d unsupported (the RGB image is a plain operand)", x->C);
  return conv_bwd_weight_impl(x->x, x->C, dy, dw_oihw, ci_offset, Cin_total, nullptr, workspace,
                              workspace_bytes, N, H, W, Cout, stride, ksize == 1,
                              (hipStream_t)stream, 0, x->alpha, x->beta, slope);
}

// unet_conv_in_bwd_weight of a 32 -> 32 channel layer with the layer's InstanceNorm + LeakyReLU +
// dropout backward applied ON LOAD: `g` = dL/da (w.r.t. the activated output), y / coef5 / sums as
// for unet_conv3x3_bwd_data_dz_wino (unet_instnorm_bwd_coefs).  Writes dz = dL/dy to dz_out (may
// alias g) for the layer's data gradient, dgamma / dbeta / dbias, and the weight gradient.
// Shapes: unet_conv_in_bwd_weight_dz_supported.
extern "C" int unet_conv_in_bwd_weight_dz_supported(int N, int H, int W, int Cx, int Cout) {
  return (N > 0 && H > 0 && W > 0 && wgrad_wino32_ok(N, H, W, Cx, Cout, 1) &&
          wgrad_batch_chunk(N, H, W, Cx, Cout, 1) >= N) ? 1 : 0;
}
extern "C" int unet_conv_in_bwd_weight_dz(const unet_act_src* x, float slope, const float* g,
                                          const float* y, const float* coef5, const float* sums,
                                          const float* gamma, const float* rstd, float dz_slope,
                                          float* dz_out, float* dgamma, float* dbeta, float* dbias,
                                          float* dw_oihw, int ci_offset, int Cin_total,
                                          void* workspace, size_t workspace_bytes, int N, int H,
                                          int W, int Cout, unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && x->alpha && x->beta && g && y && coef5 && sums && gamma && rstd && dz_out,
               "conv_in_bwd_weight_dz: null pointer");
  UNET_REQUIRE(unet_conv_in_bwd_weight_dz_supported(N, H, W, x->C, Cout),
               "conv_in_bwd_weight_dz: shape N=%d %dx%d %d->%d not supported", N, H, W, x->C, Cout);
  WgradParams dz{};
  dz.dz_y = y; dz.dz_coef = coef5; dz.dz_out = dz_out;
  dz.dz_sums = reinterpret_cast<const float2*>(sums);
  dz.dz_gamma = gamma; dz.dz_rstd = rstd; dz.dz_slope = dz_slope;
  dz.dz_dgamma = dgamma; dz.dz_dbeta = dbeta; dz.dz_dbias = dbias;
  return conv_bwd_weight_impl(x->x, x->C, g, dw_oihw, ci_offset, Cin_total, nullptr, workspace,
                              workspace_bytes, N, H, W, Cout, 1, false, (hipStream_t)stream, 0,
                              x->alpha, x->beta, slope, nullptr, nullptr, 0, &dz);
}

// The same in the split-bf16 operand mode (fp32 tensors, fp32-class accuracy): stride-1 3x3
// layers whose segments split into whole 16-pixel groups run the three-plane kernel with the
// activation applied before the split, other shapes the fp32 kernels.
extern "C" int unet_conv_in_bwd_weight_bf16x3(const unet_act_src* x, float slope, const float* dy,
                                              float* dw_oihw, int ci_offset, int Cin_total,
                                              int ksize, int stride, void* workspace,
                                              size_t workspace_bytes, int N, int H, int W, int Cout,
                                              unet_stream_t stream) {
  UNET_REQUIRE(x && x->x, "conv_in_bwd_weight_bf16x3: null source");
  UNET_REQUIRE(ksize == 3 || (ksize == 1 && stride == 1),
               "conv_in_bwd_weight_bf16x3: kernel %d / stride %d unsupported", ksize, stride);
  UNET_REQUIRE(x->C == 3 ? !x->alpha : x->C % 32 == 0,
               "conv_in_bwd_weight_bf16x3: Cx %d unsupported (the RGB image is a plain operand)",
               x->C);
  const int prec = (ksize == 3 && x->C != 3) ? 3 : 0;
  return conv_bwd_weight_impl(x->x, x->C, dy, dw_oihw, ci_offset, Cin_total, nullptr, workspace,
                              workspace_bytes, N, H, W, Cout, stride, ksize == 1,
                              (hipStream_t)stream, prec, x->alpha, x->beta, slope);
}

// ---- conv3x3(upsample2x(a)): weight gradient w.r.t. the up-sampled operand at low resolution
namespace {
int up_wgrad_chunk(int N, int h, int w, int Cx, int Cout) {
  const long long a = (long long)h * w * Cx * 4, b = (long long)h * w * 9 * Cout * 4;
  return unet_conv::batch_chunk(N, a > b ? a : b);
}
size_t up_wgrad_ws_floats(int N, int h, int w, int Cx, int Cout, bool wide) {
  const int nmax = up_wgrad_chunk(N, h, w, Cx, Cout);
  if (nmax < 1) return 0;
  const size_t E = (size_t)9 * Cx * Cout;
  size_t slabs = 0;
  for (int nb = 0; nb < N; nb += nmax) {
    const WgradPlan pl =
        make_plan_taps((long long)(N - nb < nmax ? N - nb : nmax) * h * w, Cx, Cout, wide);
    slabs += (size_t)pl.split * pl.sps;
  }
  return slabs * E + 2 * (size_t)ceil_div((int)slabs, kSlabChunk) * E;
}
}  // namespace

extern "C" size_t unet_conv3x3_up_bwd_weight_workspace_bytes(int N, int h, int w, int Cx, int Cout) {
  if (N <= 0 || h <= 0 || w <= 0 || Cx <= 0 || Cout <= 0) return 0;
  const size_t a = up_wgrad_ws_floats(N, h, w, Cx, Cout, false);
  const size_t b = up_wgrad_ws_floats(N, h, w, Cx, Cout, true);
  return (a > b ? a : b) * sizeof(float);
}

static int up_bwd_weight_impl(const unet_act_src* x, float slope, const float* D, float* dw_oihw,
                              int ci_offset, int Cin_total, void* workspace,
                              size_t workspace_bytes, int N, int h, int w, int Cout,
                              hipStream_t stream, int b16);

extern "C" int unet_conv3x3_up_bwd_weight(const unet_act_src* x, float slope, const float* D,
                                          float* dw_oihw, int ci_offset, int Cin_total,
                                          void* workspace, size_t workspace_bytes, int N, int h,
                                          int w, int Cout, unet_stream_t stream) {
  return up_bwd_weight_impl(x, slope, D, dw_oihw, ci_offset, Cin_total, workspace, workspace_bytes,
                            N, h, w, Cout, (hipStream_t)stream, 0);
}

// x (activated on load) and D are bf16 tensors
extern "C" int unet_conv3x3_up_bwd_weight_b16(const unet_act_src* x, float slope, const uint16_t* D,
                                              float* dw_oihw, int ci_offset, int Cin_total,
                                              void* workspace, size_t workspace_bytes, int N,
                                              int h, int w, int Cout, unet_stream_t stream) {
  return up_bwd_weight_impl(x, slope, reinterpret_cast<const float*>(D), dw_oihw, ci_offset,
                            Cin_total, workspace, workspace_bytes, N, h, w, Cout,
                            (hipStream_t)stream, 1);
}

static int up_bwd_weight_impl(const unet_act_src* x, float slope, const float* D, float* dw_oihw,
                              int ci_offset, int Cin_total, void* workspace,
                              size_t workspace_bytes, int N, int h, int w, int Cout,
                              hipStream_t stream, int b16) {
  const long long es = b16 ? 2 : 4;
  UNET_REQUIRE(x && x->x && D && dw_oihw && workspace, "conv3x3_up_bwd_weight: null pointer");
  const int Cx = x->C;
  UNET_REQUIRE(Cx > 0 && Cx % 32 == 0 && Cout > 0 && Cout % 32 == 0 && N > 0 && h > 0 && w > 0,
               "conv3x3_up_bwd_weight: bad shape Cx=%d Cout=%d", Cx, Cout);
  UNET_REQUIRE(ci_offset >= 0 && ci_offset + Cx <= Cin_total, "conv3x3_up_bwd_weight: bad ci slice");
  UNET_REQUIRE(!x->alpha || x->beta, "conv3x3_up_bwd_weight: alpha without beta");
  const int nmax = up_wgrad_chunk(N, h, w, Cx, Cout);
  UNET_REQUIRE(nmax >= 1, "conv3x3_up_bwd_weight: one image exceeds the 2 GiB buffer-descriptor range");
  const bool wide = !b16 && wgrad_tiles(Cx, Cout) >= 4;   // as in conv_bwd_weight_impl
  const size_t need = up_wgrad_ws_floats(N, h, w, Cx, Cout, wide) * sizeof(float);
  if (workspace_bytes < need) {
    unet_set_error("conv3x3_up_bwd_weight: workspace %zu < %zu bytes", workspace_bytes, need);
    return UNET_E_WORKSPACE;
  }
  float* ws = reinterpret_cast<float*>(workspace);
  const size_t E = (size_t)9 * Cx * Cout;
  int nslab = 0;
  for (int nb = 0; nb < N; nb += nmax) {   // one pass unless a tensor exceeds 2 GiB
    const int nc = N - nb < nmax ? N - nb : nmax;
    const long long Q = (long long)nc * h * w;
    const WgradPlan pl = make_plan_taps(Q, Cx, Cout, wide, b16);
    WgradParams p{};
    p.x = reinterpret_cast<const float*>(reinterpret_cast<const char*>(x->x) +
                                         (size_t)nb * h * w * Cx * es);
    p.dy = reinterpret_cast<const float*>(reinterpret_cast<const char*>(D) +
                                          (size_t)nb * h * w * 9 * Cout * es);
    p.b16 = b16;
    p.partial = ws + (size_t)nslab * E; p.Cx = Cx; p.Cout = Cout;
    p.N = nc; p.H = h; p.W = w; p.Ho = h; p.Wo = w;
    p.segs_per_row = 0; p.total_segs = pl.total_segs;
    p.segs_per_block = pl.segs_per_block; p.split = pl.split;
    p.ci_tiles = Cx / pl.ci_t; p.co_tiles = Cout / pl.co_t;
    p.x_bytes = (unsigned)(Q * Cx * es);
    p.dy_bytes = (unsigned)(Q * 9 * Cout * es);
    p.alpha = x->alpha ? x->alpha + (size_t)nb * Cx : nullptr;
    p.beta = x->alpha ? x->beta + (size_t)nb * Cx : nullptr;
    p.slope = slope;
    int rc;
    if (wide) {
      if (pl.ci_t == 64) rc = launch_wgrad_taps8<64, 64, 32>(p, stream);
      else if (pl.co_t == 64) rc = launch_wgrad_taps8<32, 64, 32>(p, stream);
      else rc = launch_wgrad_taps8<32, 32, 64>(p, stream);
    } else {
      if (pl.ci_t == 64 && b16) rc = launch_wgrad_taps_b16<64, 64, 16>(p, stream);
      else if (pl.ci_t == 32 && pl.co_t == 32 && b16) rc = launch_wgrad_taps_b16<32, 32, 64>(p, stream);
      else if (pl.ci_t == 64) rc = launch_wgrad_taps<64, 64, 16>(p, stream);
      else if (pl.co_t == 64) rc = launch_wgrad_taps<32, 64, 16>(p, stream);
      else rc = launch_wgrad_taps<32, 32, 32>(p, stream);
    }
    if (rc != UNET_OK) return rc;
    nslab += pl.split * pl.sps;
  }
  return emit_wgrad_reduction(ws, nslab, E, dw_oihw, Cx, Cout, ci_offset, Cin_total, false, false,
                              stream);
}

// Mixed-precision pipeline: x (activated on load) and dy are bf16 tensors; stride-1 layers on the
// bf16 matrix cores, stride-2 layers and 16-pixel segments on the fp32 matrix cores; dw fp32.
// The RGB image (x->C == 3) stays fp32.
extern "C" int unet_conv_in_bwd_weight_b16(const unet_act_src* x, float slope, const uint16_t* dy,
                                           float* dw_oihw, int ci_offset, int Cin_total, int ksize,
                                           int stride, void* workspace, size_t workspace_bytes,
                                           int N, int H, int W, int Cout, unet_stream_t stream) {
  UNET_REQUIRE(x && x->x, "conv_in_bwd_weight_b16: null source");
  UNET_REQUIRE(ksize == 3 || (ksize == 1 && stride == 1),
               "conv_in_bwd_weight_b16: kernel %d / stride %d unsupported", ksize, stride);
  UNET_REQUIRE(x->C == 3 ? !x->alpha : x->C % 32 == 0,
               "conv_in_bwd_weight_b16: Cx %d unsupported (the RGB image is a plain operand)", x->C);
  return conv_bwd_weight_impl(x->x, x->C, reinterpret_cast<const float*>(dy), dw_oihw, ci_offset,
                              Cin_total, nullptr, workspace, workspace_bytes, N, H, W, Cout, stride,
                              ksize == 1, (hipStream_t)stream, 1, x->alpha, x->beta, slope, nullptr,
                              nullptr, 1);
}

// Weight gradient of the RGB stem from the uint8 image (normalised on load, W % 128 == 0);
// workspace as unet_conv3x3_bwd_weight_workspace_bytes(N, H, W, 3, Cout, 1).
extern "C" int unet_stem_u8_bwd_weight(const uint8_t* image_hwc, const float* mean3,
                                       const float* std3, const float* dy, float* dw_oihw,
                                       void* workspace, size_t workspace_bytes, int N, int H,
                                       int W, int Cout, unet_stream_t stream) {
  UNET_REQUIRE(image_hwc && mean3 && std3, "stem_u8_bwd_weight: null pointer");
  const float ms[6] = {mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]};
  return conv_bwd_weight_impl(nullptr, 3, dy, dw_oihw, 0, 3, nullptr, workspace, workspace_bytes,
                              N, H, W, Cout, 1, false, (hipStream_t)stream, 0, nullptr, nullptr,
                              0.f, image_hwc, ms);
}

// ---- deferred reductions (include/unet_hip.h) ----
extern "C" int unet_wgrad_defer_begin(void) {
  ReduceQueue& q = reduce_queue();
  UNET_REQUIRE(!q.on || q.pending == 0, "wgrad_defer_begin: reductions still pending (flush first)");
  q.on = true;
  return UNET_OK;
}
extern "C" int unet_wgrad_defer_pending(void) { return reduce_queue().pending; }
extern "C" int unet_wgrad_defer_flush(unet_stream_t stream) {
  return flush_reduce_queue((hipStream_t)stream);
}
extern "C" int unet_wgrad_defer_end(unet_stream_t stream) {
  const int rc = flush_reduce_queue((hipStream_t)stream);
  reduce_queue().on = false;
  return rc;
}

// conv_igemm.hip — 3x3 convolution forward and data-gradient as one NHWC
// implicit-GEMM kernel on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), its row-fused and
// stride-2-dgrad variants, the RGB stem, and the C entry points of all convolution paths
// (patch-staged kernels: conv_patch.hip; bf16 / split-bf16 gather-GEMMs: conv_lowp.hip).
//
// GEMM view: M = N*Hl*Wl logical positions, K = taps*Cin,
// Ncols = output channels.  One "gather-GEMM" kernel serves
//   forward  stride 1/2 : input pixel (a*s + ky-1, b*s + kx-1), output (a, b)
//   dgrad    stride 1   : input pixel (a + 1-ky,  b + 1-kx),    output (a, b)
//   dgrad    stride 2   : four launches, one per output parity class (py,px):
//                         input (a + oy, b + ox) over that class's tap subset,
//                         output (2a+py, 2b+px)
// through a per-launch tap table.  The K loop walks (channel chunk of 32) x
// (tap); A rows (one pixel's 32 contiguous channels = one 128-B line) and the
// [32][BN] weight panel are register-staged into double-buffered LDS.
//
// Replaces nn.Conv2d forward / aten::convolution_backward(data) of
// Our_UNet/models/unet.py:106-115 (reference is NCHW via oneDNN/cuDNN).
#include "conv_params.h"
#include <stdlib.h>

namespace unet_conv {
namespace {


// FUSED (the fused layer pipeline, see IgemmParams): the A rows are raw convolution outputs and
// the producing layer's InstanceNorm + LeakyReLU + dropout is applied between the buffer load
// and the LDS write (per tap: a pixel is re-activated for each tap that stages it); with
// p.stats the epilogue emits the tile's per-column (mean, M2); the statistics need every tile
// inside ONE image (Hl*Wl % BM == 0, checked by the dispatcher), the activation does not (each
// staged row carries its own image's coefficients).
//
// KG > 1 (deep layers: M = N*16*16 rows leave one 64x64 tile per CU, and a lone 4-wave block
// cannot hide the load latency of its K loop): KG groups of four waves share the tile, group g
// runs K steps [g, g+1) * KS / KG through its own pair of LDS stages, and the partial
// accumulators are summed through LDS in group order before the common epilogue.
template <int BM, int BN, int WM, int WN, int BK, bool FUSED = false, int KG = 1>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 2 : 1) void conv_igemm_kernel(const IgemmParams p) {
  constexpr int LDA = BK + 4;  // 144-B (80-B) rows: conflict-free ds_read_b128 across 16 rows
  constexpr int SEGS = BK / 4;         // 16-B segments per tile row
  constexpr int ROWS = 256 / SEGS;     // tile rows covered by one loader pass
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int A_PASSES = BM / ROWS;
  constexpr int B_PASSES = BN / ROWS;
  constexpr int A_TILE = BM * LDA, B_TILE = BN * LDA;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // K group: uniform per wave, kept in an SGPR so the K-step bookkeeping stays scalar
  const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  const bool lead = KG == 1 || grp == 0;
  float* As = smem + grp * 2 * (A_TILE + B_TILE);
  float* Bs = As + 2 * A_TILE;

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

  const int tiles_n = p.Ncols / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int HlWl = p.Hl * p.Wl;
  const int M = p.N * HlWl;
  const int Ktot = p.C0 + p.C1;

  // ---- loaders: thread -> (row lrow + 32*i, 16-B segment lseg) for both tiles ----
  const int lrow = tid / SEGS, lseg = tid % SEGS;
  int a_nb[A_PASSES], a_iy[A_PASSES], a_ix[A_PASSES];
  int a_img[FUSED ? A_PASSES : 1];   // FUSED: image of each staged row (its coefficient row)
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + lrow + ROWS * i;
    if (m < M) {
      const int n = m / HlWl;
      const int r = m - n * HlWl;
      const int a = r / p.Wl;
      const int b = r - a * p.Wl;
      a_nb[i] = n * p.Hin * p.Win;
      a_iy[i] = a * p.sin;
      a_ix[i] = b * p.sin;
      if (FUSED) a_img[i] = n;
    } else {
      a_nb[i] = 0;
      a_iy[i] = -(1 << 24);
      a_ix[i] = 0;
      if (FUSED) a_img[i] = 0;
    }
  }
  // Buffer descriptors: out-of-range lanes (zero padding, rows past M) get an offset beyond
  // num_records and read 0 with no branch, so the K loop is one basic block.
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  // weight rows of this thread: ((n_off + n0 + lrow + 32*j) * Ktot + lseg*4) floats
  const unsigned wrow_off = (unsigned)((p.n_off + n0 + lrow) * Ktot + lseg * 4) * 4u;

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  f32x4 ra[A_PASSES], rb[B_PASSES];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int KS = p.ntaps * (Ktot / BK) / KG;   // K steps of this group (the launcher checks % KG)

  // FUSED: coefficients of this thread's four channels per staged row, in-image flags
  constexpr int CP = FUSED ? A_PASSES : 1;
  f32x4 ca[CP], cb[CP];
  float cs = 1.f;
  unsigned okm = 0;
  auto load_tiles = [&](int t, int chunk) {
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int oy = (int)(e & 3u) - 1, ox = (int)((e >> 2) & 3u) - 1;
    const int wt = (int)(e >> 4);
    const int c = chunk * BK;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? (p.src0_pitch ? p.src0_pitch : p.C0) : p.C1;
    const int coff = (first ? c : c - p.C0) + lseg * 4 + wt * p.tap_cstride;
    if (FUSED) {
      const float* al = first ? p.act0_alpha : p.act1_alpha;
      const float* be = first ? p.act0_beta : p.act1_beta;
      if (al) {   // uniform
#pragma unroll
        for (int i = 0; i < CP; ++i) {
          ca[i] = *reinterpret_cast<const f32x4*>(al + (size_t)a_img[i] * Cs + coff);
          cb[i] = *reinterpret_cast<const f32x4*>(be + (size_t)a_img[i] * Cs + coff);
        }
        cs = p.slope;
      } else {    // plain source: z = v, slope 1 = identity
#pragma unroll
        for (int i = 0; i < CP; ++i) {
          ca[i] = f32x4{1.f, 1.f, 1.f, 1.f};
          cb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        cs = 1.f;
      }
      okm = 0;
    }
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int iy = a_iy[i] + oy, ix = a_ix[i] + ox;
      const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      if (FUSED) okm |= (ok ? 1u : 0u) << i;
      // invalid lanes get bit 31 set: beyond num_records (< 2 GiB), the load returns 0
      const unsigned off = ((unsigned)((a_nb[i] + iy * p.Win + ix) * Cs + coff) * 4u) |
                           (ok ? 0u : 0x80000000u);
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      ra[i] = __builtin_bit_cast(f32x4, v);
    }
    const unsigned woff = wrow_off + (unsigned)(wt * p.tap_stride + c) * 4u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsw, woff + (unsigned)(ROWS * j * Ktot) * 4u, 0, 0);
      rb[j] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto store_tiles = [&](int buf) {
    float* Ab = As + buf * A_TILE + lrow * LDA + lseg * 4;
    float* Bb = Bs + buf * B_TILE + lrow * LDA + lseg * 4;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      if (FUSED) ra[i] = act4(ra[i], ca[i], cb[i], cs, (okm >> i) & 1u);
      *reinterpret_cast<f32x4*>(Ab + ROWS * i * LDA) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bb + ROWS * j * LDA) = rb[j];
  };

  int t_next = 0, chunk_next = 0;
  if (KG > 1) {   // tap-fastest order: step = chunk * ntaps + t
    chunk_next = grp * KS / p.ntaps;
    t_next = grp * KS - chunk_next * p.ntaps;
  }
  auto advance = [&](bool on) {  // branch-free: keeps the K step a single basic block
    const int tn = t_next + 1;
    const bool wrap = tn == p.ntaps;
    t_next = on ? (wrap ? 0 : tn) : t_next;
    chunk_next = on ? chunk_next + (wrap ? 1 : 0) : chunk_next;
  };

  load_tiles(t_next, chunk_next);
  advance(KS > 1);
  store_tiles(0);
  __syncthreads();

  // fragment addresses: lane (li, lh) reads 4 consecutive k at row li, k offset 4*lh.
  // MFMA r of a k-group of 8 then multiplies k = kb + 4*lh + r on both operands.
  const int frag_off = li * LDA + 4 * lh;
  for (int ks = 0; ks < KS; ++ks) {
    const int buf = ks & 1;
    // Always stage a tile (the last iteration re-stages the final one into the idle buffer):
    // no branch, so the whole K step is one scheduling region.
    load_tiles(t_next, chunk_next);
    advance(ks + 2 < KS);
    const float* Ab = As + buf * A_TILE + wm0 * LDA + frag_off;
    const float* Bb = Bs + buf * B_TILE + wn0 * LDA + frag_off;
    f32x4 a[2][TM], b[2][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(Ab + m * 32 * LDA);
#pragma unroll
    for (int n = 0; n < TN; ++n) b[0][n] = *reinterpret_cast<const f32x4*>(Bb + n * 32 * LDA);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 8) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
          a[nxt][m] = *reinterpret_cast<const f32x4*>(Ab + m * 32 * LDA + (kk + 1) * 8);
#pragma unroll
        for (int n = 0; n < TN; ++n)
          b[nxt][n] = *reinterpret_cast<const f32x4*>(Bb + n * 32 * LDA + (kk + 1) * 8);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][m][r], b[cur][n][r], acc[m][n],
                                                             0, 0, 0);
    }
    // FUSED: the activation arithmetic on the staged rows must not drift above the MFMAs (it
    // would wait for this step's buffer loads before the matrix work that is meant to hide them)
    if (!FUSED) store_tiles(buf ^ 1);
    // Pin the software pipeline (hipcc otherwise sinks every read to just before its first
    // use): fragment reads run one k-group ahead of the MFMAs, the next tile's buffer loads
    // issue behind the first MFMA group, the LDS writes of the staged tile come last.
    if (BK / 8 > 1) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
    else __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, A_PASSES + B_PASSES, 0);
#pragma unroll
    for (int kk = 1; kk < BK / 8; ++kk) {
      if (kk + 1 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
    }
    if (FUSED) {
      __builtin_amdgcn_sched_barrier(0);
      store_tiles(buf ^ 1);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x200, A_PASSES + B_PASSES, 0);
    }
    __syncthreads();
  }

  // the K loop ended on a barrier: every LDS stage is free scratch from here on
  float* scratch = smem;
  if (KG > 1) {   // partial sums of groups 1.. in [group][wave][register][lane] order
    constexpr int PART = 256 * 16 * TM * TN;
    static_assert((KG - 1) * PART + 2 * (BM / WM) * BN <= KG * 2 * (A_TILE + B_TILE),
                  "partial sums + reduction scratch fit in the stages");
    if (!lead) {
      float* dst = smem + (grp - 1) * PART + wave * 64 * 16 * TM * TN + lane;
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((m * TN + n) * 16 + r) * 64] = acc[m][n][r];
    }
    __syncthreads();
    if (lead) {
#pragma unroll
      for (int g = 1; g < KG; ++g) {
        const float* src = smem + (g - 1) * PART + wave * 64 * 16 * TM * TN + lane;
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] += src[((m * TN + n) * 16 + r) * 64];
      }
    }
    scratch = smem + (KG - 1) * PART;
  }

  // ---- epilogue: D row = (reg&3) + 8*(reg>>2) + 4*lh, column = li ----
  // BSTATS (uniform `bs`): the output is FINAL for the layer behind it, so each stored block is
  // followed by the read of the matching raw outputs y (same element offsets) and the lane's
  // share of that layer's InstanceNorm-backward sums (IgemmParams); needs every tile inside one
  // image and all rows valid (dispatcher check)
  const bool direct = (p.sout == 1 && p.Hl == p.Hout && p.Wl == p.Wout);
  const bool bs = !FUSED && p.bs_partial;
  constexpr int WAVES_M_ = BM / WM;
  float2* bred = reinterpret_cast<float2*>(scratch);
  const int bimg = m0 / HlWl;
#pragma unroll
  for (int n = 0; n < TN; ++n) {
    const int col = n0 + wn0 + n * 32 + li;
    const float bv = p.bias ? p.bias[col] : 0.f;
    BwdCoef cf{};
    if (bs) cf = bwd_coef(p, bimg, col);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      unsigned o[16];   // element offsets into p.out (the tensor stays below 2^29 elements)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int mg = lead ? m0 + row : M;   // the other K groups store nothing
        unsigned opix = (unsigned)mg;
        if (!direct) {
          const int nn = mg / HlWl;
          const int rr = mg - nn * HlWl;
          const int a = rr / p.Wl;
          const int b = rr - a * p.Wl;
          opix = (unsigned)((nn * p.Hout + (a * p.sout + p.py)) * p.Wout + (b * p.sout + p.px));
        }
        o[r] = mg < M ? opix * (unsigned)p.ldo + (unsigned)col : kNoOut;
      }
      if (bs) {
        if (p.accumulate) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] += bv + (o[r] != kNoOut ? p.out[o[r]] : 0.f);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] += bv;
        }
        float yv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          yv[r] = 0.f;
          if (o[r] != kNoOut) {
            p.out[o[r]] = acc[m][n][r];
            yv[r] = p.bs_y[o[r]];
          }
        }
        if (lead) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {   // the order of wave_bwd_stats (conv_params.h)
            const float z = fmaf(yv[r], cf.A, cf.B0);
            const float gz = acc[m][n][r] * cf.mk * (z > 0.f ? 1.f : p.slope);
            s1 += gz;
            s2 = fmaf(gz, (yv[r] - cf.mu) * cf.rs, s2);
          }
        }
      } else {
        store_block16_off(p.out, o, acc[m][n], bv, p.accumulate);
      }
    }
    if (bs) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lh == 0 && lead) bred[(wave / WAVES_N) * BN + wn0 + n * 32 + li] = float2{s1, s2};
    }
  }
  if (bs) {
    float2 out;
    if (block_col_sums<BN, WAVES_M_>(bred, out))
      p.bs_partial[((size_t)bimg * p.bs_tiles + p.bs_tile0 + (m0 - bimg * HlWl) / BM) * p.Ncols +
                   n0 + tid] = out;
  }
  if (FUSED && p.stats) {   // uniform; the K loop ended on a barrier: the A tiles are free scratch
    constexpr int WAVES_M = BM / WM;
    float2* red = reinterpret_cast<float2*>(scratch);
    static_assert(WAVES_M * BN * 2 <= 2 * A_TILE, "stats scratch fits in the A tiles");
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = n0 + wn0 + n * 32 + li;
      const float bv = p.bias ? p.bias[col] : 0.f;
      const float2 mine = wave_col_stats<TM>([&](int m, int r) { return acc[m][n][r] + bv; });
      if (lh == 0 && lead) red[(wave / WAVES_N) * BN + wn0 + n * 32 + li] = mine;
    }
    float2 out;
    if (block_col_stats<BN, WAVES_M>(red, 0, 0, false, float2{0.f, 0.f}, 32.f * TM, out))
    {
      const int img = m0 / HlWl;   // the whole tile lies in this image (dispatcher check)
      p.stats[((size_t)img * p.stats_tiles + (m0 - img * HlWl) / BM) * p.Ncols + n0 + tid] = out;
    }
  }
}

template <int BM, int BN, int WM, int WN, int BK = 32, bool FUSED = false, int KG = 1>
int launch_igemm(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDA = BK + 4;
  constexpr size_t lds = KG * 2 * (size_t)(BM + BN) * LDA * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS stages of every K group fit one CU");
  auto kern = conv_igemm_kernel<BM, BN, WM, WN, BK, FUSED, KG>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const long long tiles = ceil_div64(M, BM) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256 * KG), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_igemm");
  return UNET_OK;
}


// ---------------------------------------------------------------------------
// Row-fused variant for stride-1 layers whose logical row length is a multiple of BM
// (the full-resolution, narrow layers where A traffic, not MFMA, limits the generic
// kernel).  One K step = (32-channel chunk, kernel row ky): the A tile is the BM+2
// consecutive pixels x0-1 .. x0+BM of ONE image row, staged once and consumed by the three
// taps kx through row shifts 0/1/2 of the fragment reads; three [BN][32] weight tiles ride
// along.  3x less A traffic and 3x fewer barriers per MFMA than the per-tap K loop.
// ---------------------------------------------------------------------------
// FUSED: activation on load + statistics epilogue of the fused layer pipeline (IgemmParams).
template <int BM, int BN, int WM, int WN, bool PW, bool FUSED = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_rf_kernel(const IgemmParams p, int ntiles) {
  // PW (persistent weights): Ktot == 32, so all nine [BN][32] weight tiles are loaded into LDS
  // once per workgroup and only the A rows stream through the double buffer.
  constexpr int BK = 32, LDA = BK + 4;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int AR = BM + 2;                       // staged pixels per K step
  constexpr int A_PASSES = (AR + 31) / 32;
  constexpr int B_PASSES = PW ? 0 : 3 * BN / 32;
  constexpr int A_TILE = AR * LDA, B_TILE = 3 * BN * LDA;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_TILE;   // PW: [3 ky][3 shifts][BN][LDA], else 2 x [3 shifts][BN][LDA]
  // FUSED: scratch of the statistics epilogue behind the pipeline buffers
  float2* red = reinterpret_cast<float2*>(Bs + (PW ? 3 : 2) * B_TILE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const int Ktot = p.C0 + p.C1;
  const int HW = p.Hin * p.Win;
  const int flip = p.sin;  // reused: 0 = forward (dy = ky-1), 1 = data gradient (dy = 1-ky)

  // Persistent workgroup: tiles are walked so that the workgroups of one XCD (blockIdx % 8)
  // cover one contiguous eighth of the image rows (halo rows stay in that XCD's L2); the
  // K-step pipeline runs across tile boundaries, so the short K loop of a 32-channel layer
  // (3 steps) never drains.  Requires Ncols == BN (one column tile).
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

  const int lrow = tid >> 3, lseg = tid & 7;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // B slot j: row rr = lrow + 32*j of the [3][BN] panel -> (shift s, column n)
  constexpr int BP = B_PASSES > 0 ? B_PASSES : 1;
  int b_s[BP], b_off[BP];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) {
    const int rr = lrow + 32 * j;
    const int sft = rr / BN, n = rr - sft * BN;
    b_s[j] = flip ? 2 - sft : sft;                     // kx of that shift
    b_off[j] = ((p.n_off + n) * Ktot + lseg * 4) * 4;  // bytes within a tap
  }
  if (PW) {  // one-time fill of the persistent weight panel [ky][shift][n][k]
    for (int rr = lrow; rr < 9 * BN; rr += 32) {
      const int ky = rr / (3 * BN), r2 = rr - ky * 3 * BN;
      const int sft = r2 / BN, n = r2 - sft * BN;
      const int kx = flip ? 2 - sft : sft;
      const unsigned off = (unsigned)(((ky * 3 + kx) * p.tap_stride + (p.n_off + n) * Ktot + lseg * 4)) * 4u;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0));
      *reinterpret_cast<f32x4*>(Bs + rr * LDA + lseg * 4) = v;
    }
  }

  f32x4 ra[A_PASSES], rb[BP];
  f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = {0.f, 0.f, 0.f, 0.f};   // FUSED: this thread's channels
  float cs = 1.f;
  unsigned okm = 0;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int KS = 3 * (Ktot / BK);

  auto load_tiles = [&](int tile, int ky, int chunk) {
    const int m0 = tile * BM;             // the whole tile lies in one image row
    const int img = m0 / HW;
    const int rem = m0 - img * HW;
    const int y0 = rem / p.Win, x0 = rem - y0 * p.Win;
    const int c = chunk * BK;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? p.C0 : p.C1;
    const int coff = (first ? c : c - p.C0) + lseg * 4;
    const int y = y0 + (flip ? 1 - ky : ky - 1);
    const bool yok = (unsigned)y < (unsigned)p.Hin;
    const int rowbase = (img * p.Hin + y) * p.Win;
    if (FUSED) {
      const float* al = first ? p.act0_alpha : p.act1_alpha;
      const float* be = first ? p.act0_beta : p.act1_beta;
      if (al) {   // uniform
        ca = *reinterpret_cast<const f32x4*>(al + (size_t)img * Cs + coff);
        cb = *reinterpret_cast<const f32x4*>(be + (size_t)img * Cs + coff);
        cs = p.slope;
      } else {    // plain source: z = v, slope 1 = identity
        ca = f32x4{1.f, 1.f, 1.f, 1.f};
        cb = f32x4{0.f, 0.f, 0.f, 0.f};
        cs = 1.f;
      }
      okm = 0;
    }
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int r = lrow + 32 * i;
      const int x = x0 - 1 + r;
      const bool ok = yok && r < AR && (unsigned)x < (unsigned)p.Win;
      if (FUSED) okm |= (ok ? 1u : 0u) << i;
      const unsigned off = ((unsigned)((rowbase + x) * Cs + coff) * 4u) | (ok ? 0u : 0x80000000u);
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const unsigned off = (unsigned)b_off[j] + (unsigned)((ky * 3 + b_s[j]) * p.tap_stride + c) * 4u;
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0));
    }
  };
  auto store_tiles = [&](int buf) {
    float* Ab = As + buf * A_TILE + lrow * LDA + lseg * 4;
    float* Bb = Bs + buf * B_TILE + lrow * LDA + lseg * 4;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      if (FUSED) ra[i] = act4(ra[i], ca, cb, cs, (okm >> i) & 1u);
      if (32 * (i + 1) <= AR || lrow + 32 * i < AR) *reinterpret_cast<f32x4*>(Ab + 32 * i * LDA) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bb + 32 * j * LDA) = rb[j];
  };

  // load cursor (tile, ky, chunk); it stops advancing on the last step of the last tile
  int l_tile = t_first, l_ky = 0, l_chunk = 0;
  auto advance = [&]() {
    int ky = l_ky + 1, ch = l_chunk, tl = l_tile;
    if (ky == 3) { ky = 0; ch += 1; }
    if (ch == Ktot / BK) { ch = 0; tl += t_stride; }
    const bool on = tl < t_end;
    l_ky = on ? ky : l_ky; l_chunk = on ? ch : l_chunk; l_tile = on ? tl : l_tile;
  };

  load_tiles(l_tile, l_ky, l_chunk);
  advance();
  store_tiles(0);
  __syncthreads();

  const int frag_off = li * LDA + 4 * lh;
  int step = 0;
  for (int tile = t_first; tile < t_end; tile += t_stride) {
    for (int ks = 0; ks < KS; ++ks, ++step) {
      const int buf = step & 1;
      load_tiles(l_tile, l_ky, l_chunk);
      advance();
      const float* Ab = As + buf * A_TILE + wm0 * LDA + frag_off;
      const float* Bb = Bs + (PW ? ks : buf) * B_TILE + wn0 * LDA + frag_off;
      constexpr int NG = 3 * (BK / 8);  // (shift, k-group) steps
      f32x4 a[2][TM], b[2][TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(Ab + m * 32 * LDA);
#pragma unroll
      for (int n = 0; n < TN; ++n) b[0][n] = *reinterpret_cast<const f32x4*>(Bb + n * 32 * LDA);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < NG) {
          const int sft = (g + 1) / (BK / 8), kk = (g + 1) % (BK / 8);
#pragma unroll
          for (int m = 0; m < TM; ++m)
            a[nxt][m] = *reinterpret_cast<const f32x4*>(Ab + (m * 32 + sft) * LDA + kk * 8);
#pragma unroll
          for (int n = 0; n < TN; ++n)
            b[nxt][n] = *reinterpret_cast<const f32x4*>(Bb + (sft * BN + n * 32) * LDA + kk * 8);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][m][r], b[cur][n][r],
                                                               acc[m][n], 0, 0, 0);
      }
      // pinned pipeline: reads one group ahead, next step's loads behind the first group
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, A_PASSES + B_PASSES, 0);
#pragma unroll
      for (int g = 1; g < NG; ++g) {
        if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
      }
      // FUSED: keep the activation arithmetic (and its wait for this step's loads) behind the MFMAs
      if (FUSED) __builtin_amdgcn_sched_barrier(0);
      store_tiles(buf ^ 1);
      __syncthreads();
    }
    // tile epilogue (stores drain while the next tile's K steps run)
    const int m0 = tile * BM;
    if (FUSED && p.stats) {   // uniform.  `red` is rewritten one tile (>= 3 barriers) later
      constexpr int WAVES_M = BM / WM;
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const float bv = p.bias ? p.bias[wn0 + n * 32 + li] : 0.f;
        const float2 mine = wave_col_stats<TM>([&](int m, int r) { return acc[m][n][r] + bv; });
        if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + n * 32 + li] = mine;
      }
      float2 out;
      if (block_col_stats<BN, WAVES_M>(red, 0, 0, false, float2{0.f, 0.f}, 32.f * TM, out)) {
        const int img = m0 / HW;
        p.stats[((size_t)img * p.stats_tiles + (m0 - img * HW) / BM) * p.Ncols + tid] = out;
      }
    }
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = wn0 + n * 32 + li;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        float* o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          o[r] = p.out + (size_t)(m0 + row) * p.ldo + col;
        }
        if (!FUSED && p.bs_partial) {   // uniform: final values kept for the reductions below
          if (p.accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] += bv + *o[r];
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] += bv;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) *o[r] = acc[m][n][r];
        } else {
          store_block16(o, acc[m][n], bv, p.accumulate);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        }
      }
    }
    if (!FUSED && p.bs_partial) {   // uniform.  `red` is rewritten one tile (>= 3 barriers) later
      constexpr int WAVES_M = BM / WM;
      const int img = m0 / HW;
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int col = wn0 + n * 32 + li;
        const BwdCoef cf = bwd_coef(p, img, col);
        const float2 mine = wave_bwd_stats<TM>(
            cf, p.slope, [&](int m, int r) { return acc[m][n][r]; },
            [&](int m, int r) {
              const int row = wm0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
              return p.bs_y[(size_t)(m0 + row) * p.ldo + col];
            });
        if (lh == 0) red[(wave / WAVES_N) * BN + wn0 + n * 32 + li] = mine;
      }
      float2 out;
      if (block_col_sums<BN, WAVES_M>(red, out))
        p.bs_partial[((size_t)img * p.bs_tiles + (m0 - img * HW) / BM) * p.Ncols + tid] = out;
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    }
  }
}

template <int BM, int BN, int WM, int WN, bool PW, bool FUSED = false>
int launch_igemm_rf(const IgemmParams& p, int flip, hipStream_t stream, int* stats_px = nullptr) {
  constexpr int LDA = 36;
  constexpr size_t lds = (PW ? (size_t)(2 * (BM + 2) + 9 * BN) * LDA * sizeof(float)
                             : 2 * (size_t)((BM + 2) + 3 * BN) * LDA * sizeof(float)) +
                         (size_t)(BM / WM) * BN * sizeof(float2);   // statistics scratch
  auto kern = conv_igemm_rf_kernel<BM, BN, WM, WN, PW, FUSED>;
  UNET_SET_DYN_LDS(kern, lds);
  IgemmParams q = p;
  q.sin = flip;
  if (FUSED && stats_px) {
    if (q.stats) { *stats_px = BM; q.stats_tiles = p.Hin * p.Win / BM; }
    else *stats_px = 0;
  }
  if (!FUSED) {   // data gradient: stats_px doubles as the BSTATS tile report
    if (stats_px && q.bs_partial) { *stats_px = BM; q.bs_tiles = p.Hin * p.Win / BM; q.bs_tile0 = 0; }
    else q.bs_partial = nullptr;
  }
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int ntiles = (int)(M / BM);
  const int resident = 256 * (int)((160 * 1024) / lds);   // CUs x workgroups that fit in LDS
  const int grid = ntiles < resident ? ntiles : resident;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, q, ntiles);
  UNET_CHECK_LAUNCH("conv_igemm_rf");
  return UNET_OK;
}

// stride-1 3x3 with rows that tile exactly: use the row-fused kernel for narrow outputs
bool rf_applicable(const IgemmParams& p) {
  return p.ntaps == 9 && p.sout == 1 && p.Hl == p.Hin && p.Wl == p.Win &&
         p.Hl == p.Hout && p.Wl == p.Wout && p.Wl % 128 == 0 && p.Ncols == 32;
}


// ---------------------------------------------------------------------------
// Stride-2 data gradient in ONE launch.  Logical position (a,b) of the dy grid produces the
// four dx pixels (2a+py, 2b+px).  The nine taps fall into four dy shifts (oy,ox) in {0,1}^2:
//   (0,0): taps (1,1)->class 0, (1,2)->1, (2,1)->2, (2,2)->3      class = py*2+px
//   (0,1): taps (1,0)->1, (2,0)->3      (1,0): taps (0,1)->2, (0,2)->3      (1,1): tap (0,0)->3
// so a K step = (32-channel chunk of dy, shift): ONE staged dy tile feeds 4/2/2/1 weight
// tiles, each accumulating into its class's accumulator block (4 x 16 VGPRs per wave).  The
// four shifts are unrolled so every step has a compile-time tap list (no per-step branch).
// Block = 4 waves stacked along M (128 logical positions) x 32 dx channels.
// ---------------------------------------------------------------------------
struct S2Tap { int ky, kx, cls; };
template <int SH> struct S2Shift;
template <> struct S2Shift<0> { static constexpr int n = 4, oy = 0, ox = 0;
  static constexpr S2Tap t[4] = {{1, 1, 0}, {1, 2, 1}, {2, 1, 2}, {2, 2, 3}}; };
template <> struct S2Shift<1> { static constexpr int n = 2, oy = 0, ox = 1;
  static constexpr S2Tap t[4] = {{1, 0, 1}, {2, 0, 3}, {0, 0, 0}, {0, 0, 0}}; };
template <> struct S2Shift<2> { static constexpr int n = 2, oy = 1, ox = 0;
  static constexpr S2Tap t[4] = {{0, 1, 2}, {0, 2, 3}, {0, 0, 0}, {0, 0, 0}}; };
template <> struct S2Shift<3> { static constexpr int n = 1, oy = 1, ox = 1;
  static constexpr S2Tap t[4] = {{0, 0, 3}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; };

__global__ __launch_bounds__(256, 2) void conv_dgrad_s2_kernel(const IgemmParams p) {
  constexpr int BM = 128, BN = 32, BK = 32, LDA = BK + 4;
  constexpr int A_TILE = BM * LDA, B_TILE = 4 * BN * LDA;
  constexpr int STAGE = A_TILE + B_TILE;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm0 = wave * 32;
  const int tiles_n = p.Ncols / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int HlWl = p.Hl * p.Wl;   // dy grid
  const int M = p.N * HlWl;
  const int Ktot = p.C0;          // Cout of the forward conv

  const int lrow = tid >> 3, lseg = tid & 7;
  int a_nb[4], a_y[4], a_x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < M) {
      const int n = m / HlWl;
      const int r = m - n * HlWl;
      a_y[i] = r / p.Wl;
      a_x[i] = r - a_y[i] * p.Wl;
      a_nb[i] = n * HlWl;
    } else {
      a_nb[i] = 0; a_y[i] = -(1 << 24); a_x[i] = 0;
    }
  }
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  const unsigned wrow_off = (unsigned)((p.n_off + n0 + lrow) * Ktot + lseg * 4) * 4u;

  f32x4 ra[4], rb[4];
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  const int nchunks = Ktot / BK;

  auto load_step = [&](auto shift_tag, int chunk) {
    using SHT = decltype(shift_tag);
    const int c = chunk * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = a_y[i] + SHT::oy, x = a_x[i] + SHT::ox;
      const bool ok = (unsigned)y < (unsigned)p.Hl && (unsigned)x < (unsigned)p.Wl;
      const unsigned off = ((unsigned)((a_nb[i] + y * p.Wl + x) * Ktot + c + lseg * 4) * 4u) |
                           (ok ? 0u : 0x80000000u);
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < SHT::n; ++j) {
      const unsigned off = wrow_off + (unsigned)((SHT::t[j].ky * 3 + SHT::t[j].kx) * p.tap_stride + c) * 4u;
      rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0));
    }
  };
  auto store_step = [&](auto shift_tag, int buf) {
    using SHT = decltype(shift_tag);
    float* Ab = smem + buf * STAGE + lrow * LDA + lseg * 4;
    float* Bb = smem + buf * STAGE + A_TILE + lrow * LDA + lseg * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Ab + 32 * i * LDA) = ra[i];
#pragma unroll
    for (int j = 0; j < SHT::n; ++j) *reinterpret_cast<f32x4*>(Bb + 32 * j * LDA) = rb[j];
  };
  const int frag_off = li * LDA + 4 * lh;
  auto compute_step = [&](auto shift_tag, int buf) {
    using SHT = decltype(shift_tag);
    const float* Ab = smem + buf * STAGE + wm0 * LDA + frag_off;
    const float* Bb = smem + buf * STAGE + A_TILE + frag_off;
    f32x4 a[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a[kk] = *reinterpret_cast<const f32x4*>(Ab + kk * 8);
#pragma unroll
    for (int j = 0; j < SHT::n; ++j) {
      f32x4 b[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) b[kk] = *reinterpret_cast<const f32x4*>(Bb + 32 * j * LDA + kk * 8);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[SHT::t[j].cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][r], b[kk][r],
                                                                    acc[SHT::t[j].cls], 0, 0, 0);
    }
    // pinned pipeline: A fragments + first tap's B fragments, then each tap's 16 MFMAs with the
    // next tap's reads ahead of them; the next step's buffer loads follow the first MFMA group
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int j = 0; j < SHT::n; ++j) {
      if (j + 1 < SHT::n) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      if (j == 0) __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
    }
  };

  // software pipeline over the flattened (chunk, shift) sequence, two LDS stages
  load_step(S2Shift<0>{}, 0);
  store_step(S2Shift<0>{}, 0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int nc = min(chunk + 1, nchunks - 1);   // last chunk re-stages itself (harmless)
    load_step(S2Shift<1>{}, chunk);
    compute_step(S2Shift<0>{}, 0);
    store_step(S2Shift<1>{}, 1);
    __syncthreads();
    load_step(S2Shift<2>{}, chunk);
    compute_step(S2Shift<1>{}, 1);
    store_step(S2Shift<2>{}, 0);
    __syncthreads();
    load_step(S2Shift<3>{}, chunk);
    compute_step(S2Shift<2>{}, 0);
    store_step(S2Shift<3>{}, 1);
    __syncthreads();
    load_step(S2Shift<0>{}, nc);
    compute_step(S2Shift<3>{}, 1);
    store_step(S2Shift<0>{}, 0);
    __syncthreads();
  }

  // epilogue: logical row -> (n, a, b); class c -> dx pixel (2a + c/2, 2b + c%2)
  const int col = n0 + li;
  size_t base[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int mg = m0 + row;
    const int nn = mg / HlWl;
    const int rr = mg - nn * HlWl;
    const int a = rr / p.Wl;
    const int b = rr - a * p.Wl;
    base[r] = mg < M ? ((size_t)nn * p.Hout + 2 * a) * p.Wout + 2 * b : ~(size_t)0;
  }
  float s1 = 0.f, s2 = 0.f;
  BwdCoef cf{};
  if (p.bs_partial) cf = bwd_coef(p, m0 / HlWl, col);   // uniform; tiles lie inside one image
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float* o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      o[r] = base[r] != ~(size_t)0
                 ? p.out + (base[r] + (size_t)(c >> 1) * p.Wout + (c & 1)) * p.ldo + col
                 : nullptr;
    if (p.bs_partial) {   // uniform: final values (all rows valid here) + the next stage's sums
      if (p.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] += *o[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) *o[r] = acc[c][r];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float y = p.bs_y[o[r] - p.out];
        const float z = fmaf(y, cf.A, cf.B0);
        const float gz = acc[c][r] * cf.mk * (z > 0.f ? 1.f : p.slope);
        s1 += gz;
        s2 = fmaf(gz, (y - cf.mu) * cf.rs, s2);
      }
    } else {
      store_block16(o, acc[c], 0.f, p.accumulate);
    }
  }
  if (p.bs_partial) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    float2* red = reinterpret_cast<float2*>(smem);   // the pipeline ended on a barrier
    if (lh == 0) red[wave * BN + li] = float2{s1, s2};
    float2 out;
    if (block_col_sums<BN, 4>(red, out)) {
      const int img = m0 / HlWl;
      p.bs_partial[((size_t)img * p.bs_tiles + (m0 - img * HlWl) / BM) * p.Ncols + n0 + tid] = out;
    }
  }
}

int launch_dgrad_s2(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)(128 + 4 * 32) * 36 * sizeof(float);
  UNET_SET_DYN_LDS(conv_dgrad_s2_kernel, lds);
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const long long tiles = ceil_div64(M, 128) * (p.Ncols / 32);
  hipLaunchKernelGGL(conv_dgrad_s2_kernel, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_dgrad_s2");
  return UNET_OK;
}

}  // namespace

// stats_px != nullptr: the fused-layer call (activation on load; statistics epilogue into
// p.stats when every tile lies inside one image, reported as *stats_px = pixels per statistics
// tile, else *stats_px = 0 and the caller runs the stand-alone statistics kernel).
template <int BM, int BN, int WM, int WN, int KG = 1>
static int launch_igemm_fused(IgemmParams p, hipStream_t stream, int* stats_px) {
  const int HlWl = p.Hl * p.Wl;
  const bool direct = p.sout == 1 && p.Hl == p.Hout && p.Wl == p.Wout;
  if (direct && p.stats && HlWl % BM == 0) { *stats_px = BM; p.stats_tiles = HlWl / BM; }
  else { *stats_px = 0; p.stats = nullptr; }
  return launch_igemm<BM, BN, WM, WN, 32, true, KG>(p, stream);
}

// Deep layers (at most one 64x64 tile per CU): number of K groups per block (1 = no split).
static int deep_k_groups(const IgemmParams& p) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  if (p.Ncols % 64 != 0 || ceil_div64(M, 64) * (p.Ncols / 64) > 256) return 1;
  const int ks = p.ntaps * ((p.C0 + p.C1) / 32);
  int kg = 4;
  while (kg > 1 && (ks % kg != 0 || ks / kg < 4)) kg >>= 1;
  return kg;
}

// plain gather-GEMM launch; with bs_px (data gradient whose output is final for a layer) the
// BSTATS epilogue runs where every tile lies inside one image, else *bs_px = 0
template <int BM, int BN, int WM, int WN, int KG = 1>
static int launch_igemm_bs(IgemmParams p, hipStream_t stream, int* bs_px) {
  if (bs_px) {
    const int HlWl = p.Hl * p.Wl;
    if (p.bs_partial && HlWl % BM == 0) { *bs_px = BM; p.bs_tiles = HlWl / BM * (p.sout * p.sout); }
    else { *bs_px = 0; p.bs_partial = nullptr; }
  } else {
    p.bs_partial = nullptr;
  }
  return launch_igemm<BM, BN, WM, WN, 32, false, KG>(p, stream);
}

static int launch_igemm_deep(IgemmParams p, hipStream_t stream, int* bs_px) {
  const int kg = deep_k_groups(p);
  return kg == 4   ? launch_igemm_bs<64, 64, 32, 32, 4>(p, stream, bs_px)
         : kg == 2 ? launch_igemm_bs<64, 64, 32, 32, 2>(p, stream, bs_px)
                   : launch_igemm_bs<64, 64, 32, 32>(p, stream, bs_px);
}

int dispatch_igemm(const IgemmParams& p, hipStream_t stream, int* stats_px, int* bs_px) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  if (patch_f32_applicable(p)) {   // conv_patch.hip; 1 = no tile shape fits this launch
    const int rc = launch_patch_f32_auto(p, stream, stats_px, bs_px);
    if (rc != 1) return rc;
  }
  if (stats_px) {
    if (patch_s2_applicable(p)) {   // conv_patch.hip; 1 = too few tiles
      const int rc = launch_patch_s2_auto(p, stream, stats_px);
      if (rc != 1) return rc;
    }
    if (nc % 128 == 0 && ceil_div64(M, 128) * (nc / 128) >= 256)
      return launch_igemm_fused<128, 128, 64, 64>(p, stream, stats_px);
    if (nc % 64 == 0 && ceil_div64(M, 128) * (nc / 64) >= 256)
      return launch_igemm_fused<128, 64, 64, 32>(p, stream, stats_px);
    if (nc % 64 == 0 && M <= 128 * 256) {
      const int kg = deep_k_groups(p);
      return kg == 4   ? launch_igemm_fused<64, 64, 32, 32, 4>(p, stream, stats_px)
             : kg == 2 ? launch_igemm_fused<64, 64, 32, 32, 2>(p, stream, stats_px)
                       : launch_igemm_fused<64, 64, 32, 32>(p, stream, stats_px);
    }
    return launch_igemm_fused<128, 32, 32, 32>(p, stream, stats_px);
  }
  // low-resolution data gradient of the up-sampled operand ("channel taps": a plain GEMM over
  // K = 9 Cout): 64 x 64 tiles at four workgroups per CU where the larger tiles leave the chip
  // half filled or the layer has 64 columns (tools/bench_lowres_b16.py --fp32: 199 -> 188 us at
  // 8192 x 512 x 2304, 242 -> 226 at 524288 x 64 x 288; the other three shapes +-0)
  if (p.tap_cstride != 0 && nc % 64 == 0 && M % 64 == 0 && (M / 64) * (nc / 64) >= 512 &&
      (nc == 64 || ceil_div64(M, 128) * (nc / 128) < 512))
    return launch_igemm_bs<64, 64, 32, 32>(p, stream, bs_px);
  if (bs_px) {   // per-class launches of one stride-2 gradient pass bs_tile0 themselves
    if (nc % 128 == 0 && ceil_div64(M, 128) * (nc / 128) >= 256)
      return launch_igemm_bs<128, 128, 64, 64>(p, stream, bs_px);
    if (nc % 64 == 0 && ceil_div64(M, 128) * (nc / 64) >= 256)
      return launch_igemm_bs<128, 64, 64, 32>(p, stream, bs_px);
    if (nc % 64 == 0 && M <= 128 * 256) return launch_igemm_deep(p, stream, bs_px);
    return launch_igemm_bs<128, 32, 32, 32>(p, stream, bs_px);
  }
  // Largest tile that still yields >= 256 workgroups (one per CU); otherwise the
  // small 64x64 tile.
  if (nc % 128 == 0 && ceil_div64(M, 128) * (nc / 128) >= 256)
    return launch_igemm<128, 128, 64, 64>(p, stream);
  if (nc % 64 == 0 && ceil_div64(M, 128) * (nc / 64) >= 256)
    return launch_igemm<128, 64, 64, 32>(p, stream);
  if (nc % 64 == 0 && M <= 128 * 256) return launch_igemm_deep(p, stream, nullptr);
  return launch_igemm<128, 32, 32, 32>(p, stream);
}

namespace {

// ---------------------------------------------------------------------------
// RGB stem (Cin = 3): the whole K = 27 im2col row is gathered into LDS and one
// K step of 28 feeds the matrix cores.  HBM-bound (writes 32 channels per pixel).
// ---------------------------------------------------------------------------
constexpr int STEM_PIX = 256;  // pixels per block
constexpr int STEM_LDK = 29;   // odd row stride: conflict-free column reads

template <typename TO>
__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ wf,
                                                            const float* __restrict__ bias,
                                                            TO* __restrict__ y, int N, int H,
                                                            int W, int Cout) {
  __shared__ float A[STEM_PIX * STEM_LDK];
  __shared__ float B[28 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int HW = H * W;
  const long long M = (long long)N * HW;
  const long long m0 = (long long)blockIdx.x * STEM_PIX;
  const int co0 = blockIdx.y * 32;

  for (int i = tid; i < 28 * 32; i += 256) {
    const int k = i >> 5, c = i & 31;  // k = tap*3 + ci; wf is [tap][co][ci]
    B[i] = (k < 27) ? wf[((k / 3) * Cout + co0 + c) * 3 + (k % 3)] : 0.f;
  }
  // im2col gather: item = (pixel, tap) -> 3 channels
  for (int it = tid; it < STEM_PIX * 9; it += 256) {
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
    float* d = A + pix * STEM_LDK + t * 3;
    d[0] = v0; d[1] = v1; d[2] = v2;
  }
  for (int pix = tid; pix < STEM_PIX; pix += 256) A[pix * STEM_LDK + 27] = 0.f;
  __syncthreads();

  // each wave: 64 pixels = 2 row blocks of 32
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
  for (int k = 0; k < 28; k += 2) {
    const float b = B[(k + lh) * 32 + li];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float a = A[(wave * 64 + m * 32 + li) * STEM_LDK + k + lh];
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
    }
  }
  const float bv = bias ? bias[co0 + li] : 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const long long mg = m0 + row;
      if (mg < M) st1(y + (size_t)mg * Cout + co0 + li, acc[m][r] + bv);
    }
}

// Row form of the stem for images whose width is a multiple of 128: a block's 128 pixels lie in
// one image row, so the three input rows are copied raw (coalesced) as R[ky][3*(col+1) + ci]
// and the im2col element k = 9*ky + (3*kx + ci) of pixel px is R[ky][3*px + 3*kx + ci] - no
// per-element gather.  Lanes run along pixels at a stride of 3 words (conflict-free).
constexpr int STEM_ROW_PIX = 128;
constexpr int STEM_ROW_PITCH = 393;

// Input pixel of the RGB stem: fp32 as stored, or uint8 normalised on load with the dataset's
// expression ((v / 255) - mean[c]) / std[c] (Our_UNet/src/train.py:303-308) - the fp32 image is
// then never written to HBM (12x less input traffic).
struct StemNorm { float mean[3], std[3]; };
__device__ __forceinline__ float stem_pixel(const float* x, size_t i, int, const StemNorm&) {
  return x[i];
}
__device__ __forceinline__ float stem_pixel(const unsigned char* x, size_t i, int c,
                                            const StemNorm& nm) {
  return ((float)x[i] / 255.0f - nm.mean[c]) / nm.std[c];
}

template <typename T, typename TO = float>
__global__ __launch_bounds__(256) void conv_stem_fwd_rows_kernel(const T* __restrict__ x,
                                                                 const float* __restrict__ wf,
                                                                 const float* __restrict__ bias,
                                                                 TO* __restrict__ y, int N,
                                                                 int H, int W, int Cout,
                                                                 float2* __restrict__ stats,
                                                                 const StemNorm nm) {
  __shared__ float Rw[3 * STEM_ROW_PITCH];
  __shared__ float B[28 * 32];
  __shared__ float2 red[4 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int HW = H * W;
  const long long m0 = (long long)blockIdx.x * STEM_ROW_PIX;
  const int co0 = blockIdx.y * 32;
  const int n = (int)(m0 / HW);
  const int rem = (int)(m0 - (long long)n * HW);
  const int yy = rem / W, x0 = rem - yy * W;
  for (int i = tid; i < 28 * 32; i += 256) {
    const int k = i >> 5, c = i & 31;  // k = tap*3 + ci; wf is [tap][co][ci]
    B[i] = (k < 27) ? wf[((k / 3) * Cout + co0 + c) * 3 + (k % 3)] : 0.f;
  }
  for (int i = tid; i < 3 * 390; i += 256) {
    const int ky = i / 390, j = i - ky * 390;
    const int iy = yy + ky - 1, ix = x0 - 1 + j / 3;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
      v = stem_pixel(x, ((size_t)n * HW + (size_t)iy * W + x0 - 1) * 3 + j, j % 3, nm);
    Rw[ky * STEM_ROW_PITCH + j] = v;
  }
  __syncthreads();
  const int px = wave * 32 + li;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int k2 = 0; k2 < 28; k2 += 2) {
    const int k = k2 + lh;
    const int kc = k < 27 ? k : 26;          // row 27 of B is zero: any finite a will do
    const float a = Rw[(kc / 9) * STEM_ROW_PITCH + (kc % 9) + 3 * px];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, B[k * 32 + li], acc, 0, 0, 0);
  }
  const float bv = bias ? bias[co0 + li] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    st1(y + (size_t)(m0 + row) * Cout + co0 + li, acc[r] + bv);
  }
  if (stats) {   // fused layer pipeline: (mean, M2) of this 128-pixel tile per output column
    const float2 mine = wave_col_stats<1>([&](int, int r) { return acc[r] + bv; });
    if (lh == 0) red[wave * 32 + li] = mine;
    float2 out;
    if (block_col_stats<32, 4>(red, 0, 0, false, float2{0.f, 0.f}, 32.f, out))
      stats[((size_t)n * (HW / STEM_ROW_PIX) + rem / STEM_ROW_PIX) * Cout + co0 + tid] = out;
  }
}

// Walking form of the row stem (round 4): a workgroup walks R rows down its 128-pixel column
// strip.  The weight fragments live in registers for the whole walk, the input rows in a ring of
// eight LDS rows, one new row a step instead of three.  Row yy + 3 is fetched (unconditional
// buffer loads: rows and columns outside the image read zero through the descriptor) during step
// yy and written to LDS at the end of step yy + 1, behind that step's output stores: the counted
// wait then covers the loads only and leaves the 16 stores of the step in flight - the memory
// counter is in order, so a load issued behind the stores of the step before it cannot be waited
// for without them (the first version of this kernel did: 72 us against 97 for the rows kernel
// above, which pays a weight gather, three row loads and their latency per 128 pixels).
// Same MFMA sequence and the same 128-pixel statistics tiles as the rows kernel: identical bits.
template <typename T, typename TO, int R>
__global__ __launch_bounds__(256) void conv_stem_fwd_walk_kernel(const T* __restrict__ x,
                                                                 const float* __restrict__ wf,
                                                                 const float* __restrict__ bias,
                                                                 TO* __restrict__ y, int N,
                                                                 int H, int W, int Cout,
                                                                 float2* __restrict__ stats,
                                                                 const StemNorm nm) {
  static_assert(R % 2 == 0, "two register sets alternate");
  __shared__ float Rw[8 * STEM_ROW_PITCH];
  __shared__ float2 red[4 * 32];
  // output staging, one 32-pixel x 32-channel tile per wave: the accumulator layout gives a lane
  // 16 scattered 2- or 4-byte values, the tile leaves as 16-byte stores (1 KB contiguous per
  // instruction at Cout = 32) - 2 / 4 store instructions a step instead of 16.  Row pitch: the
  // row + 16 bytes (bf16), + 32 (fp32: the rows of the two half-waves, 4 apart, 32 banks apart).
  constexpr int OROW = 32 * (int)sizeof(TO);
  constexpr int OPITCH = OROW + (sizeof(TO) == 2 ? 16 : 32);
  __shared__ __attribute__((aligned(16))) char Os[4 * 32 * OPITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int HW = H * W;
  const int strips = W / STEM_ROW_PIX, groups = H / R;
  int b = blockIdx.x;
  const int strip = b % strips; b /= strips;
  const int grp = b % groups;
  const int n = b / groups;
  const int x0 = strip * STEM_ROW_PIX, y0 = grp * R, co0 = blockIdx.y * 32;
  // B operand of this lane, k = 2 * k2 + lh (k = tap*3 + ci; wf is [tap][co][ci]; k = 27: zero)
  float breg[14];
  int kof[14];      // A operand: LDS offset inside its row, and the row (ky) it comes from
  int kyv[14];
  const int px = wave * 32 + li;
#pragma unroll
  for (int k2 = 0; k2 < 14; ++k2) {
    const int k = 2 * k2 + lh;
    breg[k2] = k < 27 ? wf[((k / 3) * Cout + co0 + li) * 3 + (k % 3)] : 0.f;
    const int kc = k < 27 ? k : 26;          // (its weight is zero: any finite a will do)
    kyv[k2] = kc / 9;
    kof[k2] = (kc % 9) + 3 * px;
  }
  // loader: a row of the strip is 390 consecutive values from pixel x0 - 1 on; lanes outside
  // the image (or past the 390) carry the kill bit: beyond num_records, the load returns 0
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(x), 0, (int)((long long)N * HW * 3 * (long long)sizeof(T)), 0x00020000);
  const int e0 = tid, e1 = tid + 256;
  const bool has1 = e1 < 390;
  const unsigned kill0 = (unsigned)(x0 - 1 + e0 / 3) < (unsigned)W ? 0u : 0x80000000u;
  const unsigned kill1 = (has1 && (unsigned)(x0 - 1 + e1 / 3) < (unsigned)W) ? 0u : 0x80000000u;
  const int c0 = e0 % 3, c1 = e1 % 3;
  auto fetch = [&](unsigned off) __attribute__((always_inline)) {
    if constexpr (sizeof(T) == 4) {
      return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, off, 0, 0));
    } else {
      return (float)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rsx, off, 0, 0);
    }
  };
  auto load_row = [&](int iy, float& v0, float& v1) __attribute__((always_inline)) {
    const unsigned rk = (unsigned)iy < (unsigned)H ? 0u : 0x80000000u;      // uniform
    const unsigned base = (unsigned)(((n * H + iy) * W + x0 - 1) * 3) * (unsigned)sizeof(T);
    v0 = fetch((base + (unsigned)e0 * (unsigned)sizeof(T)) | kill0 | rk);
    v1 = fetch((base + (unsigned)e1 * (unsigned)sizeof(T)) | kill1 | rk);
  };
  auto put_row = [&](int iy, float v0, float v1) __attribute__((always_inline)) {
    if constexpr (sizeof(T) == 1) {   // uint8: normalise; the zero padding stays zero
      const bool in = (unsigned)iy < (unsigned)H;
      v0 = (in && !kill0) ? (v0 / 255.0f - nm.mean[c0]) / nm.std[c0] : 0.f;
      v1 = (in && !kill1) ? (v1 / 255.0f - nm.mean[c1]) / nm.std[c1] : 0.f;
    }
    float* d = Rw + ((iy + 1) & 7) * STEM_ROW_PITCH;
    d[e0] = v0;
    if (has1) d[e1] = v1;
  };
#pragma unroll
  for (int r = -1; r <= 2; ++r) {
    float v0, v1;
    load_row(y0 + r, v0, v1);
    put_row(y0 + r, v0, v1);
  }
  __syncthreads();
  const float bv = bias ? bias[co0 + li] : 0.f;
  float na[2], nb[2];     // rows in flight: [0] fetched in even steps, [1] in odd ones
  na[1] = nb[1] = 0.f;    // (nothing to write at the end of step 0: row y0 + 2 is in LDS)
  auto step = [&](int s, auto parc) __attribute__((always_inline)) {
    constexpr int PAR = decltype(parc)::value;
    const int yy = y0 + s;
    // row yy + 3, needed by step s + 2 (past the walk: a dead load, the wait stays counted)
    load_row(s + 2 < R ? yy + 3 : -1, na[PAR], nb[PAR]);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < 14; ++k2) {
      const float a = Rw[((yy + kyv[k2]) & 7) * STEM_ROW_PITCH + kof[k2]];   // row yy - 1 + ky
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, breg[k2], acc, 0, 0, 0);
    }
    const size_t m0 = (size_t)n * HW + (size_t)yy * W + x0;
    {
      char* ow = Os + wave * 32 * OPITCH;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        *reinterpret_cast<TO*>(ow + row * OPITCH + li * (int)sizeof(TO)) = (TO)(acc[r] + bv);
      }
      __builtin_amdgcn_wave_barrier();   // (one wave: its LDS accesses complete in order)
      constexpr int SEGS = OROW / 16;            // 16-byte segments per pixel: 4 (bf16) / 8 (fp32)
      constexpr int RPP = 64 / SEGS;             // pixels per store instruction
      typedef int i32x4s __attribute__((ext_vector_type(4)));
      const int seg = lane % SEGS, r0 = lane / SEGS;
      char* yb = reinterpret_cast<char*>(y + (m0 + wave * 32) * Cout + co0) + seg * 16;
#pragma unroll
      for (int q = 0; q < 32 / RPP; ++q) {
        const int row = q * RPP + r0;
        const i32x4s v = *reinterpret_cast<const i32x4s*>(ow + row * OPITCH + seg * 16);
        *reinterpret_cast<i32x4s*>(yb + (size_t)row * Cout * sizeof(TO)) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // row yy + 2 (fetched during step s - 1) into its slot: nobody reads that slot before the
    // barrier below (this step reads rows yy - 1 .. yy + 1, the slot held row yy - 6)
    __builtin_amdgcn_sched_barrier(0);   // (behind the stores: hipcc hoists it to the top otherwise)
    if (s > 0 && s + 1 < R) put_row(yy + 2, na[PAR ^ 1], nb[PAR ^ 1]);
    if (stats) {   // (mean, M2) of this 128-pixel tile per output column
      const float2 mine = wave_col_stats<1>([&](int, int r) { return acc[r] + bv; });
      if (lh == 0) red[wave * 32 + li] = mine;
      float2 out;
      if (block_col_stats<32, 4>(red, 0, 0, false, float2{0.f, 0.f}, 32.f, out))
        stats[((size_t)n * (HW / STEM_ROW_PIX) + (size_t)(yy * W + x0) / STEM_ROW_PIX) * Cout + co0 +
              tid] = out;
    }
    __syncthreads();
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // fully unrolled: across a loop back edge hipcc cannot count the stores between a fetch and
  // its use and waits for (nearly) everything - the wait this kernel is built to avoid
#pragma unroll
  for (int s = 0; s < R; s += 2) {
    step(s, P0{});
    step(s + 1, P1{});
  }
}

constexpr int STEM_WALK_ROWS = 8;
template <typename T, typename TO>
void launch_stem_fwd_rows(const T* x, const float* wf, const float* bias, TO* y, int N, int H, int W,
                          int Cout, float2* stats, const StemNorm& nm, hipStream_t stream) {
  static const bool walk_off = [] { const char* e = getenv("UNET_STEM_WALK"); return e && e[0] == '0'; }();
  if (!walk_off && H % STEM_WALK_ROWS == 0 &&
      (long long)N * H * W * 3 * (long long)sizeof(T) < (1LL << 31)) {   // (buffer descriptor)
    dim3 grid((unsigned)((long long)N * (H / STEM_WALK_ROWS) * (W / STEM_ROW_PIX)), Cout / 32);
    hipLaunchKernelGGL((conv_stem_fwd_walk_kernel<T, TO, STEM_WALK_ROWS>), grid, dim3(256), 0, stream,
                       x, wf, bias, y, N, H, W, Cout, stats, nm);
  } else {
    dim3 grid((unsigned)((long long)N * H * W / STEM_ROW_PIX), Cout / 32);
    hipLaunchKernelGGL((conv_stem_fwd_rows_kernel<T, TO>), grid, dim3(256), 0, stream, x, wf, bias, y,
                       N, H, W, Cout, stats, nm);
  }
}

void fill_fwd_taps(IgemmParams& p, int stride) {
  p.ntaps = 9;
  p.tapw[0] = p.tapw[1] = p.tapw[2] = 0;
  for (int t = 0; t < 9; ++t) set_tap(p, t, t / 3 - 1, t % 3 - 1, t);
  p.sin = stride;
  p.sout = 1;
  p.py = p.px = 0;
}

}  // namespace
}  // namespace unet_conv

using namespace unet_conv;

static int conv3x3_fwd_impl(const float* x0, int C0, const float* x1, int C1, const float* wf,
                            const float* bias, float* y, int N, int H, int W, int Cout,
                            int stride, int prec, hipStream_t stream,
                            const uint16_t* wf3 = nullptr) {
  UNET_REQUIRE(x0 && wf && y, "conv3x3_fwd: null pointer");
  UNET_REQUIRE(prec != 3 || wf3, "conv3x3_fwd_bf16x3: the pre-split weight planes are null");
  UNET_REQUIRE(stride == 1 || stride == 2, "conv3x3_fwd: stride %d unsupported", stride);
  UNET_REQUIRE(N > 0 && H > 0 && W > 0, "conv3x3_fwd: bad shape");
  UNET_REQUIRE(Cout > 0 && Cout % 32 == 0, "conv3x3_fwd: Cout %d must be a multiple of 32", Cout);
  if (C0 == 3) {
    UNET_REQUIRE(C1 == 0 && stride == 1, "conv3x3_fwd: RGB stem is stride-1, single source");
    const long long M = (long long)N * H * W;
    if (W % STEM_ROW_PIX == 0) {
      launch_stem_fwd_rows<float, float>(x0, wf, bias, y, N, H, W, Cout, nullptr, StemNorm{}, stream);
    } else {
      dim3 grid((unsigned)ceil_div64(M, STEM_PIX), Cout / 32);
      hipLaunchKernelGGL(conv_stem_fwd_kernel<float>, grid, dim3(256), 0, stream, x0, wf, bias, y,
                         N, H, W, Cout);
    }
    UNET_CHECK_LAUNCH("conv_stem_fwd");
    return UNET_OK;
  }
  UNET_REQUIRE(C0 > 0 && C0 % 32 == 0 && C1 >= 0 && C1 % 32 == 0,
               "conv3x3_fwd: channel counts (%d,%d) must be multiples of 32", C0, C1);
  UNET_REQUIRE(C1 == 0 || x1, "conv3x3_fwd: x1 is null with C1=%d", C1);
  UNET_REQUIRE((long long)9 * Cout * (C0 + C1) * 4 < (1LL << 31), "conv3x3_fwd: weights exceed 2 GiB");
  {  // batch chunks keep every source inside the 2 GiB buffer-descriptor range
    const int nmax = batch_chunk(N, (long long)H * W * (C0 > C1 ? C0 : C1) * 4);
    UNET_REQUIRE(nmax >= 1, "conv3x3_fwd: one image exceeds the 2 GiB buffer-descriptor range");
    if (nmax < N) {
      const size_t Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
      for (int nb = 0; nb < N; nb += nmax) {
        const int n = N - nb < nmax ? N - nb : nmax;
        const int rc = conv3x3_fwd_impl(x0 + (size_t)nb * H * W * C0,  C0,
                                        x1 ? x1 + (size_t)nb * H * W * C1 : nullptr, C1, wf, bias,
                                        y + (size_t)nb * Ho * Wo * Cout, n, H, W, Cout, stride, prec,
                                        stream, wf3);
        if (rc != UNET_OK) return rc;
      }
      return UNET_OK;
    }
  }
  IgemmParams p{};
  p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1;
  p.w = wf; p.tap_stride = Cout * (C0 + C1); p.n_off = 0; p.bias = bias;
  p.src0_bytes = (unsigned)((long long)N * H * W * C0 * 4);
  p.src1_bytes = (unsigned)((long long)N * H * W * C1 * 4);
  p.w_bytes = (unsigned)((long long)9 * Cout * (C0 + C1) * 4);
  p.out = y; p.ldo = Cout; p.accumulate = 0;
  p.N = N; p.Hin = H; p.Win = W;
  p.Hl = p.Hout = (H - 1) / stride + 1;
  p.Wl = p.Wout = (W - 1) / stride + 1;
  p.Ncols = Cout;
  p.w3 = reinterpret_cast<const __bf16*>(wf3);
  p.w3_plane = 9 * Cout * (C0 + C1);
  p.w3_bytes = (unsigned)((long long)3 * p.w3_plane * 2);
  fill_fwd_taps(p, stride);
  if (prec == 1) return dispatch_igemm_bf16(p, stream);
  if (prec == 3) return dispatch_igemm_split(p, stream);
  // K = 32: row-fused kernel with the weights resident in LDS; wider K: the patch kernel
  if (stride == 1 && rf_applicable(p) &&
      (C0 + C1 == 32 || !patch_f32_applicable(p) || p.Hin % 8 != 0))
    return (C0 + C1 == 32) ? launch_igemm_rf<128, 32, 32, 32, true>(p, 0, stream)
                           : launch_igemm_rf<128, 32, 32, 32, false>(p, 0, stream);
  return dispatch_igemm(p, stream);
}

extern "C" int unet_conv3x3_fwd(const float* x0, int C0, const float* x1, int C1, const float* wf,
                                const float* bias, float* y, int N, int H, int W, int Cout,
                                int stride, unet_stream_t stream) {
  return conv3x3_fwd_impl(x0, C0, x1, C1, wf, bias, y, N, H, W, Cout, stride, 0,
                          (hipStream_t)stream);
}

extern "C" int unet_conv3x3_fwd_bf16(const float* x0, int C0, const float* x1, int C1,
                                     const float* wf, const float* bias, float* y, int N, int H,
                                     int W, int Cout, int stride, unet_stream_t stream) {
  if (C0 == 3)   // the RGB stem is HBM-bound: it stays on the fp32 path
    return conv3x3_fwd_impl(x0, C0, x1, C1, wf, bias, y, N, H, W, Cout, stride, 0,
                            (hipStream_t)stream);
  return conv3x3_fwd_impl(x0, C0, x1, C1, wf, bias, y, N, H, W, Cout, stride, 1,
                          (hipStream_t)stream);
}

extern "C" int unet_conv3x3_fwd_bf16x3(const float* x0, int C0, const float* x1, int C1,
                                       const float* wf, const uint16_t* wf3, const float* bias,
                                       float* y, int N, int H, int W, int Cout, int stride,
                                       unet_stream_t stream) {
  // the RGB stem (K = 27, HBM-bound) stays on the fp32 matrix-core path
  return conv3x3_fwd_impl(x0, C0, x1, C1, wf, bias, y, N, H, W, Cout, stride, C0 == 3 ? 0 : 3,
                          (hipStream_t)stream, wf3);
}

static int conv3x3_bwd_data_impl(const float* dy, const float* wd, int Cin_total, int ci_offset,
                                 float* dx, int N, int H, int W, int Cout, int Ccols, int stride,
                                 int accumulate, int prec, hipStream_t stream,
                                 const uint16_t* wd3 = nullptr, int b16 = 0,
                                 unet_bwd_stats* bs = nullptr) {
  // b16: dy and dx are bf16 tensors (mixed-precision pipeline; prec is then 1)
  const long long es = b16 ? 2 : 4;
  if (bs) bs->tiles_out = 0;
  // split mode with reductions (fused pipeline): only the stride-1 patch shapes have a split
  // kernel with that epilogue; everything else runs the fp32 kernels
  if (prec == 3 && bs && stride == 2) prec = 0;
  // (b16: the stride-1 patch kernel and the gather-GEMM have the epilogue; the stride-2 patch
  // kernel reports 0 tiles)
  const bool use_bs = bs && (prec == 0 || prec == 3 || b16) && bs->y && bs->mean &&
                      bs->rstd && bs->gamma &&
                      bs->beta && bs->partial &&
                      bs->partial_bytes >= (size_t)N * ceil_div(H * W, 64) * Ccols * sizeof(float2);
  UNET_REQUIRE(dy && wd && dx, "conv3x3_bwd_data: null pointer");
  UNET_REQUIRE(prec != 3 || wd3, "conv3x3_bwd_data_bf16x3: the pre-split weight planes are null");
  UNET_REQUIRE(stride == 1 || stride == 2, "conv3x3_bwd_data: stride %d unsupported", stride);
  UNET_REQUIRE(Cout > 0 && Cout % 32 == 0 && Ccols > 0 && Ccols % 32 == 0 && ci_offset >= 0 &&
                   ci_offset + Ccols <= Cin_total,
               "conv3x3_bwd_data: bad channel counts Cout=%d Ccols=%d slice %d of %d", Cout, Ccols,
               ci_offset, Cin_total);
  UNET_REQUIRE(stride == 1 || (H % 2 == 0 && W % 2 == 0),
               "conv3x3_bwd_data: stride 2 needs even H, W");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  UNET_REQUIRE((long long)9 * Cout * Cin_total * 4 < (1LL << 31),
               "conv3x3_bwd_data: weights exceed 2 GiB");
  {  // batch chunks keep dy inside the 2 GiB buffer-descriptor range
    const int nmax = batch_chunk(N, (long long)Ho * Wo * Cout * es);
    UNET_REQUIRE(nmax >= 1, "conv3x3_bwd_data: one image exceeds the 2 GiB buffer-descriptor range");
    if (nmax < N) {   // chunked batches have no single tile layout: no BSTATS epilogue
      for (int nb = 0; nb < N; nb += nmax) {
        const int n = N - nb < nmax ? N - nb : nmax;
        const int rc = conv3x3_bwd_data_impl(
            reinterpret_cast<const float*>(reinterpret_cast<const char*>(dy) +
                                           (size_t)nb * Ho * Wo * Cout * es),
            wd, Cin_total, ci_offset,
            reinterpret_cast<float*>(reinterpret_cast<char*>(dx) + (size_t)nb * H * W * Ccols * es),
            n, H, W, Cout, Ccols, stride, accumulate, prec, stream, wd3, b16);
        if (rc != UNET_OK) return rc;
      }
      return UNET_OK;
    }
  }
  IgemmParams p{};
  p.src0 = dy; p.src1 = nullptr; p.C0 = Cout; p.C1 = 0;
  p.w = wd; p.tap_stride = Cin_total * Cout; p.n_off = ci_offset; p.bias = nullptr;
  p.src0_bytes = (unsigned)((long long)N * Ho * Wo * Cout * es);
  p.src1_bytes = 0;
  p.w_bytes = (unsigned)((long long)9 * Cout * Cin_total * 4);
  p.out = dx; p.ldo = Ccols; p.accumulate = accumulate;
  p.N = N; p.Hin = Ho; p.Win = Wo;
  p.Hout = H; p.Wout = W;
  p.Ncols = Ccols;
  p.w3 = reinterpret_cast<const __bf16*>(wd3);
  p.w3_plane = 9 * Cout * Cin_total;
  p.w3_bytes = (unsigned)((long long)(b16 ? 1 : 3) * p.w3_plane * 2);
  p.sin = 1;
  int bs_px = 0;
  if (use_bs) {
    p.bs_y = bs->y; p.bs_mean = bs->mean; p.bs_rstd = bs->rstd; p.bs_gamma = bs->gamma;
    p.bs_beta = bs->beta; p.bs_mask = bs->mask; p.slope = bs->slope;
    p.bs_partial = reinterpret_cast<float2*>(bs->partial);
  }
  if (stride == 1) {
    p.Hl = H; p.Wl = W; p.sout = 1; p.py = p.px = 0;
    p.ntaps = 9;
    p.tapw[0] = p.tapw[1] = p.tapw[2] = 0;
    for (int t = 0; t < 9; ++t) set_tap(p, t, 1 - t / 3, 1 - t % 3, t);
    if (b16) {
      const int rc = dispatch_igemm_b16(p, stream, nullptr, use_bs ? &bs_px : nullptr);
      if (rc == UNET_OK && use_bs && bs_px > 0) bs->tiles_out = H * W / bs_px;
      return rc;
    }
    if (prec == 1) return dispatch_igemm_bf16(p, stream);
    if (prec == 3 && !bs) return dispatch_igemm_split(p, stream);
    int rc = 1;
    if (prec == 3 && patch_split_applicable(p))   // fused pipeline, split mode (else: fp32 below)
      rc = launch_patch_split_fused_auto(p, stream, nullptr, use_bs ? &bs_px : nullptr);
    if (rc != 1) {
    } else
    if (c32_applicable(p))   // conv_c32.hip: 32 -> 32 channels
      rc = launch_c32(p, 0, stream, use_bs ? &bs_px : nullptr);
    else if (rf_applicable(p) && (Cout == 32 || !patch_f32_applicable(p) || p.Hin % 8 != 0))
      rc = (Cout == 32) ? launch_igemm_rf<128, 32, 32, 32, true>(p, 1, stream, &bs_px)
                        : launch_igemm_rf<128, 32, 32, 32, false>(p, 1, stream, &bs_px);
    else
      rc = dispatch_igemm(p, stream, nullptr, use_bs ? &bs_px : nullptr);
    if (rc == UNET_OK && use_bs && bs_px > 0) bs->tiles_out = H * W / bs_px;
    return rc;
  }
  // stride 2: dx[2a+py][2b+px] = sum over ky with (py+1-ky) even of dy[a + (py+1-ky)/2][..]
  p.Hl = H / 2; p.Wl = W / 2; p.sout = 2;
  {
    // one launch for all four parity classes when there are enough tiles to fill the chip
    const long long tiles = ceil_div64((long long)N * p.Hl * p.Wl, 128) * (Ccols / 32);
    if (prec == 0 && !b16) {   // conv_patch.hip: the dy patch staged once per chunk
      IgemmParams q = p;
      if (!use_bs) q.bs_partial = nullptr;
      int bt = 0;
      const int rc = launch_dgrad_s2_patch_auto(q, stream, &bt);
      if (rc != 1) {
        if (rc == UNET_OK && use_bs && bt > 0) bs->tiles_out = bt;
        return rc;
      }
    }
    if (b16) {   // mixed-precision pipeline: one launch instead of four per class
      IgemmParams q = p;
      if (!use_bs) q.bs_partial = nullptr;
      int bt = 0;
      const int rc = launch_dgrad_s2_patch_b16_auto(q, stream, use_bs ? &bt : nullptr);
      if (rc != 1) {
        if (rc == UNET_OK && use_bs && bt > 0) bs->tiles_out = bt;
        return rc;
      }
      p.bs_partial = nullptr;   // (the per-class gather-GEMM launches below: no epilogue)
    }
    if (prec != 1 && !b16 && tiles >= 512) {
      p.py = p.px = 0; p.ntaps = 9;
      if (use_bs && (p.Hl * p.Wl) % 128 == 0) p.bs_tiles = p.Hl * p.Wl / 128;
      else p.bs_partial = nullptr;
      const int rc = launch_dgrad_s2(p, stream);
      if (rc == UNET_OK && p.bs_partial) bs->tiles_out = p.bs_tiles;   // 512 output pixels each
      return rc;
    }
  }
  int class_tiles = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      p.py = py; p.px = px;
      p.tapw[0] = p.tapw[1] = p.tapw[2] = 0;
      int nt = 0;
      for (int ky = 0; ky < 3; ++ky) {
        if ((py + 1 - ky) & 1) continue;
        for (int kx = 0; kx < 3; ++kx) {
          if ((px + 1 - kx) & 1) continue;
          set_tap(p, nt, (py + 1 - ky) / 2, (px + 1 - kx) / 2, ky * 3 + kx);
          ++nt;
        }
      }
      p.ntaps = nt;
      int bpx = 0;
      p.bs_tile0 = (py * 2 + px) * (class_tiles > 0 ? class_tiles : 0);
      int rc = b16 ? dispatch_igemm_b16(p, stream, nullptr)
                   : prec == 1 ? dispatch_igemm_bf16(p, stream)
                   : (prec == 3 ? dispatch_igemm_split(p, stream)
                                : dispatch_igemm(p, stream, nullptr, use_bs ? &bpx : nullptr));
      if (rc != UNET_OK) return rc;
      if (use_bs) {
        if (bpx == 0) { p.bs_partial = nullptr; class_tiles = -1; }   // no epilogue: give up on it
        else if (class_tiles == 0) class_tiles = p.Hl * p.Wl / bpx;
      }
    }
  if (use_bs && class_tiles > 0 && p.bs_partial) bs->tiles_out = 4 * class_tiles;
  return UNET_OK;
}

extern "C" int unet_conv3x3_bwd_data(const float* dy, const float* wd, int Cin_total,
                                     int ci_offset, float* dx, int N, int H, int W, int Cout,
                                     int Ccols, int stride, int accumulate,
                                     unet_stream_t stream) {
  return conv3x3_bwd_data_impl(dy, wd, Cin_total, ci_offset, dx, N, H, W, Cout, Ccols, stride,
                               accumulate, 0, (hipStream_t)stream);
}

extern "C" int unet_conv3x3_bwd_data_bf16(const float* dy, const float* wd, int Cin_total,
                                          int ci_offset, float* dx, int N, int H, int W, int Cout,
                                          int Ccols, int stride, int accumulate,
                                          unet_stream_t stream) {
  return conv3x3_bwd_data_impl(dy, wd, Cin_total, ci_offset, dx, N, H, W, Cout, Ccols, stride,
                               accumulate, 1, (hipStream_t)stream);
}

extern "C" int unet_conv3x3_bwd_data_bs(const float* dy, const float* wd, int Cin_total,
                                        int ci_offset, float* dx, int N, int H, int W, int Cout,
                                        int Ccols, int stride, int accumulate, unet_bwd_stats* bs,
                                        unet_stream_t stream) {
  return conv3x3_bwd_data_impl(dy, wd, Cin_total, ci_offset, dx, N, H, W, Cout, Ccols, stride,
                               accumulate, 0, (hipStream_t)stream, nullptr, 0, bs);
}

// Mixed-precision pipeline with the BSTATS epilogue: dy, dx and bs->y are bf16 tensors
extern "C" int unet_conv3x3_bwd_data_bs_b16(const uint16_t* dy, const float* wd, int Cin_total,
                                            int ci_offset, uint16_t* dx, int N, int H, int W,
                                            int Cout, int Ccols, int stride, int accumulate,
                                            unet_bwd_stats* bs, unet_stream_t stream) {
  return conv3x3_bwd_data_impl(reinterpret_cast<const float*>(dy), wd, Cin_total, ci_offset,
                               reinterpret_cast<float*>(dx), N, H, W, Cout, Ccols, stride,
                               accumulate, 1, (hipStream_t)stream, nullptr, 1, bs);
}

// ... with the weights also pre-rounded to bf16 (wdb = [9][Cin_total][Cout] bf16: plane 0 of
// the pack's wd3; may be null), as unet_conv_in_fwd_b16_wb
extern "C" int unet_conv3x3_bwd_data_bs_b16_wb(const uint16_t* dy, const float* wd,
                                               const uint16_t* wdb, int Cin_total, int ci_offset,
                                               uint16_t* dx, int N, int H, int W, int Cout,
                                               int Ccols, int stride, int accumulate,
                                               unet_bwd_stats* bs, unet_stream_t stream) {
  return conv3x3_bwd_data_impl(reinterpret_cast<const float*>(dy), wd, Cin_total, ci_offset,
                               reinterpret_cast<float*>(dx), N, H, W, Cout, Ccols, stride,
                               accumulate, 1, (hipStream_t)stream, wdb, 1, bs);
}

// Split-bf16 mode of the fused pipeline (wd3 = pre-split planes, data-gradient layout); bs may
// be null (no reductions wanted: the plain split kernels run)
extern "C" int unet_conv3x3_bwd_data_bs_bf16x3(const float* dy, const float* wd,
                                               const uint16_t* wd3, int Cin_total, int ci_offset,
                                               float* dx, int N, int H, int W, int Cout, int Ccols,
                                               int stride, int accumulate, unet_bwd_stats* bs,
                                               unet_stream_t stream) {
  return conv3x3_bwd_data_impl(dy, wd, Cin_total, ci_offset, dx, N, H, W, Cout, Ccols, stride,
                               accumulate, 3, (hipStream_t)stream, wd3, 0, bs);
}

// dy and dx are bf16 tensors, bf16 matrix cores (mixed-precision pipeline)
extern "C" int unet_conv3x3_bwd_data_b16(const uint16_t* dy, const float* wd, int Cin_total,
                                         int ci_offset, uint16_t* dx, int N, int H, int W, int Cout,
                                         int Ccols, int stride, int accumulate,
                                         unet_stream_t stream) {
  return conv3x3_bwd_data_impl(reinterpret_cast<const float*>(dy), wd, Cin_total, ci_offset,
                               reinterpret_cast<float*>(dx), N, H, W, Cout, Ccols, stride,
                               accumulate, 1, (hipStream_t)stream, nullptr, 1);
}

extern "C" int unet_conv3x3_bwd_data_bf16x3(const float* dy, const float* wd,
                                            const uint16_t* wd3, int Cin_total, int ci_offset,
                                            float* dx, int N, int H, int W, int Cout, int Ccols,
                                            int stride, int accumulate, unet_stream_t stream) {
  return conv3x3_bwd_data_impl(dy, wd, Cin_total, ci_offset, dx, N, H, W, Cout, Ccols, stride,
                               accumulate, 3, (hipStream_t)stream, wd3);
}

// ---------------------------------------------------------------------------
// 1x1 convolution (the CLIP fusion layer of CLIP_UNet/models/unet.py:356-362): the same
// gather-GEMM with a one-entry tap table.  w is [Cout][Cin] (the OIHW tensor itself),
// wT is [Cin_total][Cout] for the data gradient.
// ---------------------------------------------------------------------------
extern "C" int unet_conv1x1_fwd(const float* x0, int C0, const float* x1, int C1, const float* w,
                                const float* bias, float* y, int N, int H, int W, int Cout,
                                unet_stream_t stream) {
  UNET_REQUIRE(x0 && w && y, "conv1x1_fwd: null pointer");
  UNET_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 32 == 0 && C0 > 0 && C0 % 32 == 0 &&
                   C1 >= 0 && C1 % 32 == 0 && (C1 == 0 || x1),
               "conv1x1_fwd: channel counts (%d,%d)->%d must be multiples of 32", C0, C1, Cout);
  {
    const int nmax = batch_chunk(N, (long long)H * W * (C0 > C1 ? C0 : C1) * 4);
    UNET_REQUIRE(nmax >= 1, "conv1x1_fwd: one image exceeds the 2 GiB buffer-descriptor range");
    if (nmax < N) {
      for (int nb = 0; nb < N; nb += nmax) {
        const int n = N - nb < nmax ? N - nb : nmax;
        const int rc = unet_conv1x1_fwd(x0 + (size_t)nb * H * W * C0, C0,
                                        x1 ? x1 + (size_t)nb * H * W * C1 : nullptr, C1, w, bias,
                                        y + (size_t)nb * H * W * Cout, n, H, W, Cout, stream);
        if (rc != UNET_OK) return rc;
      }
      return UNET_OK;
    }
  }
  IgemmParams p{};
  p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1;
  p.w = w; p.tap_stride = Cout * (C0 + C1); p.n_off = 0; p.bias = bias;
  p.src0_bytes = (unsigned)((long long)N * H * W * C0 * 4);
  p.src1_bytes = (unsigned)((long long)N * H * W * C1 * 4);
  p.w_bytes = (unsigned)((long long)Cout * (C0 + C1) * 4);
  p.out = y; p.ldo = Cout; p.accumulate = 0;
  p.N = N; p.Hin = p.Hl = p.Hout = H; p.Win = p.Wl = p.Wout = W;
  p.Ncols = Cout; p.sin = 1; p.sout = 1; p.py = p.px = 0;
  p.ntaps = 1; p.tapw[0] = p.tapw[1] = p.tapw[2] = 0;
  set_tap(p, 0, 0, 0, 0);
  return dispatch_igemm(p, (hipStream_t)stream);
}

extern "C" int unet_conv1x1_bwd_data(const float* dy, const float* wT, int Cin_total,
                                     int ci_offset, float* dx, int N, int H, int W, int Cout,
                                     int Ccols, int accumulate, unet_stream_t stream) {
  UNET_REQUIRE(dy && wT && dx, "conv1x1_bwd_data: null pointer");
  UNET_REQUIRE(Cout > 0 && Cout % 32 == 0 && Ccols > 0 && Ccols % 32 == 0 && ci_offset >= 0 &&
                   ci_offset + Ccols <= Cin_total, "conv1x1_bwd_data: bad channel slice");
  {
    const int nmax = batch_chunk(N, (long long)H * W * Cout * 4);
    UNET_REQUIRE(nmax >= 1, "conv1x1_bwd_data: one image exceeds the 2 GiB buffer-descriptor range");
    if (nmax < N) {
      for (int nb = 0; nb < N; nb += nmax) {
        const int n = N - nb < nmax ? N - nb : nmax;
        const int rc = unet_conv1x1_bwd_data(dy + (size_t)nb * H * W * Cout, wT, Cin_total,
                                             ci_offset, dx + (size_t)nb * H * W * Ccols, n, H, W,
                                             Cout, Ccols, accumulate, stream);
        if (rc != UNET_OK) return rc;
      }
      return UNET_OK;
    }
  }
  IgemmParams p{};
  p.src0 = dy; p.src1 = nullptr; p.C0 = Cout; p.C1 = 0;
  p.w = wT; p.tap_stride = Cin_total * Cout; p.n_off = ci_offset; p.bias = nullptr;
  p.src0_bytes = (unsigned)((long long)N * H * W * Cout * 4);
  p.src1_bytes = 0;
  p.w_bytes = (unsigned)((long long)Cout * Cin_total * 4);
  p.out = dx; p.ldo = Ccols; p.accumulate = accumulate;
  p.N = N; p.Hin = p.Hl = p.Hout = H; p.Win = p.Wl = p.Wout = W;
  p.Ncols = Ccols; p.sin = 1; p.sout = 1; p.py = p.px = 0;
  p.ntaps = 1; p.tapw[0] = p.tapw[1] = p.tapw[2] = 0;
  set_tap(p, 0, 0, 0, 0);
  return dispatch_igemm(p, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// Fused layer forward (include/unet_hip.h "fused layer pipeline"): convolution with the
// producing layers' InstanceNorm + LeakyReLU + dropout applied to the operands on load and the
// InstanceNorm statistics of the output emitted from the epilogue.
// Replaces Conv2d + InstanceNorm2d of ConvBlock (Our_UNet/models/unet.py:101-134).
// ---------------------------------------------------------------------------
namespace {
// statistics tiles hold at least 64 pixels (the smallest gather-GEMM tile)
size_t stats_partial_bytes(int N, int HoWo, int Cout) {
  return align_up((size_t)N * (size_t)ceil_div(HoWo, 64) * Cout * sizeof(float2), 256);
}
}  // namespace

extern "C" size_t unet_conv_in_fwd_workspace_bytes(int N, int H, int W, int Cout, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || stride < 1) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const size_t a = stats_partial_bytes(N, Ho * Wo, Cout) +
                   align_up(unet_in_finalize_scratch_bytes(N, Cout), 256);
  const size_t b = unet_instnorm_workspace_bytes(N, Ho * Wo, Cout);
  return a > b ? a : b;
}

// b16: the layer tensors (sources other than the RGB image, and y) are bf16 in HBM and the
// contraction runs on the bf16 matrix cores (mixed-precision pipeline); else fp32 / fp32 MFMA.
// w3 != nullptr: the split-bf16 mode - shapes the split patch kernel tiles run there (fp32
// tensors, three-term operands, fp32-class accuracy), everything else on the fp32 kernels.
static int conv_in_fwd_impl(const unet_act_src* s0, const unet_act_src* s1, float slope,
                            const float* w, const float* bias, int ksize, int stride, float* y,
                            void* workspace, size_t workspace_bytes, int* stats_px_out, int N,
                            int H, int W, int Cout, hipStream_t stream, int b16,
                            const uint16_t* w3 = nullptr) {
  const size_t es = b16 ? 2 : 4;   // bytes per activation element
  UNET_REQUIRE(s0 && s0->x && w && y && workspace && stats_px_out, "conv_in_fwd: null pointer");
  UNET_REQUIRE(ksize == 3 || ksize == 1, "conv_in_fwd: kernel size %d unsupported", ksize);
  UNET_REQUIRE(stride == 1 || (stride == 2 && ksize == 3), "conv_in_fwd: stride %d unsupported",
               stride);
  UNET_REQUIRE(N > 0 && H > 0 && W > 0, "conv_in_fwd: bad shape");
  UNET_REQUIRE(Cout > 0 && Cout % 32 == 0, "conv_in_fwd: Cout %d must be a multiple of 32", Cout);
  const int C0 = s0->C, C1 = s1 ? s1->C : 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (workspace_bytes < unet_conv_in_fwd_workspace_bytes(N, H, W, Cout, stride)) {
    unet_set_error("conv_in_fwd: workspace too small");
    return UNET_E_WORKSPACE;
  }
  int stats_px = 0;
  if (C0 == 3) {   // RGB stem: the image is a plain operand
    UNET_REQUIRE(C1 == 0 && stride == 1 && ksize == 3 && !s0->alpha,
                 "conv_in_fwd: the RGB stem is a plain stride-1 3x3 single source");
    const long long M = (long long)N * H * W;
    __bf16* yh = reinterpret_cast<__bf16*>(y);
    if (W % STEM_ROW_PIX == 0) {
      if (b16)
        launch_stem_fwd_rows<float, __bf16>(s0->x, w, bias, yh, N, H, W, Cout,
                                            reinterpret_cast<float2*>(workspace), StemNorm{}, stream);
      else
        launch_stem_fwd_rows<float, float>(s0->x, w, bias, y, N, H, W, Cout,
                                           reinterpret_cast<float2*>(workspace), StemNorm{}, stream);
      stats_px = STEM_ROW_PIX;
    } else {
      dim3 grid((unsigned)ceil_div64(M, STEM_PIX), Cout / 32);
      if (b16)
        hipLaunchKernelGGL(conv_stem_fwd_kernel<__bf16>, grid, dim3(256), 0, stream, s0->x, w, bias,
                           yh, N, H, W, Cout);
      else
        hipLaunchKernelGGL(conv_stem_fwd_kernel<float>, grid, dim3(256), 0, stream, s0->x, w, bias,
                           y, N, H, W, Cout);
    }
    UNET_CHECK_LAUNCH("conv_stem_fwd");
  } else {
    UNET_REQUIRE(C0 > 0 && C0 % 32 == 0 && C1 >= 0 && C1 % 32 == 0,
                 "conv_in_fwd: channel counts (%d,%d) must be multiples of 32", C0, C1);
    UNET_REQUIRE(C1 == 0 || s1->x, "conv_in_fwd: second source is null with C1=%d", C1);
    UNET_REQUIRE((!s0->alpha || s0->beta) && (!s1 || !s1->alpha || s1->beta),
                 "conv_in_fwd: alpha without beta");
    const int Cin = C0 + C1;
    const int taps = ksize * ksize;
    UNET_REQUIRE((long long)taps * Cout * Cin * 4 < (1LL << 31), "conv_in_fwd: weights exceed 2 GiB");
    // batch chunks keep each source inside the 2 GiB buffer-descriptor range
    const int nmax = batch_chunk(N, (long long)H * W * (C0 > C1 ? C0 : C1) * (long long)es);
    UNET_REQUIRE(nmax >= 1, "conv_in_fwd: one image exceeds the 2 GiB buffer-descriptor range");
    const bool chunked = nmax < N;
    for (int nb = 0; nb < N; nb += nmax) {
      const int n = (N - nb) < nmax ? (N - nb) : nmax;
      IgemmParams p{};
      p.src0 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(s0->x) +
                                              (size_t)nb * H * W * C0 * es);
      p.src1 = s1 ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(s1->x) +
                                                   (size_t)nb * H * W * C1 * es)
                  : nullptr;
      p.C0 = C0; p.C1 = C1;
      p.act0_alpha = s0->alpha ? s0->alpha + (size_t)nb * C0 : nullptr;
      p.act0_beta = s0->alpha ? s0->beta + (size_t)nb * C0 : nullptr;
      p.act1_alpha = (s1 && s1->alpha) ? s1->alpha + (size_t)nb * C1 : nullptr;
      p.act1_beta = (s1 && s1->alpha) ? s1->beta + (size_t)nb * C1 : nullptr;
      p.slope = slope;
      p.w = w; p.tap_stride = Cout * Cin; p.n_off = 0; p.bias = bias;
      p.src0_bytes = (unsigned)((long long)n * H * W * C0 * (long long)es);
      p.src1_bytes = (unsigned)((long long)n * H * W * C1 * (long long)es);
      p.w_bytes = (unsigned)((long long)taps * Cout * Cin * 4);
      p.out = reinterpret_cast<float*>(reinterpret_cast<char*>(y) +
                                       (size_t)nb * Ho * Wo * Cout * es);
      p.ldo = Cout; p.accumulate = 0;
      p.N = n; p.Hin = H; p.Win = W;
      p.Hl = p.Hout = Ho; p.Wl = p.Wout = Wo;
      p.Ncols = Cout;
      // the statistics epilogue needs one tile layout for the whole batch: chunked calls use
      // the stand-alone statistics pass instead
      p.stats = chunked ? nullptr : reinterpret_cast<float2*>(workspace);
      int px = 0;
      int rc;
      if (ksize == 1) {
        p.sin = 1; p.sout = 1; p.py = p.px = 0;
        p.ntaps = 1; p.tapw[0] = p.tapw[1] = p.tapw[2] = 0;
        set_tap(p, 0, 0, 0, 0);
        rc = b16 ? dispatch_igemm_b16(p, stream, &px) : dispatch_igemm(p, stream, &px);
      } else if (b16) {
        fill_fwd_taps(p, stride);
        if (w3) {   // plane 0 of the pre-split weights = the bf16-rounded weights (patch kernel)
          p.w3 = reinterpret_cast<const __bf16*>(w3);
          p.w3_plane = 9 * Cout * Cin;
          p.w3_bytes = (unsigned)((long long)p.w3_plane * 2);
        }
        rc = dispatch_igemm_b16(p, stream, &px);
      } else {
        fill_fwd_taps(p, stride);
        rc = 1;
        if (w3 && stride == 1 && patch_split_applicable(p)) {   // split-bf16 patch kernel
          p.w3 = reinterpret_cast<const __bf16*>(w3);
          p.w3_plane = 9 * Cout * Cin;
          p.w3_bytes = (unsigned)((long long)3 * p.w3_plane * 2);
          rc = launch_patch_split_fused_auto(p, stream, &px, nullptr);
        }
        if (rc != 1) {
        } else
        // K = 32: row-fused kernel with the weights resident in LDS; wider K: the patch kernel
        if (stride == 1 && c32_applicable(p))   // conv_c32.hip: 32 -> 32 channels
          rc = launch_c32(p, 1, stream, &px);
        else if (stride == 1 && rf_applicable(p) &&
            (Cin == 32 || !patch_f32_applicable(p) || p.Hin % 8 != 0))
          rc = (Cin == 32) ? launch_igemm_rf<128, 32, 32, 32, true, true>(p, 0, stream, &px)
                           : launch_igemm_rf<128, 32, 32, 32, false, true>(p, 0, stream, &px);
        else
          rc = dispatch_igemm(p, stream, &px);
      }
      if (rc != UNET_OK) return rc;
      stats_px = px;
    }
  }
  *stats_px_out = stats_px;
  return UNET_OK;
}

extern "C" int unet_conv_in_fwd(const unet_act_src* s0, const unet_act_src* s1, float slope,
                                const float* w, const float* bias, int ksize, int stride, float* y,
                                void* workspace, size_t workspace_bytes, int* stats_px_out, int N,
                                int H, int W, int Cout, unet_stream_t stream) {
  return conv_in_fwd_impl(s0, s1, slope, w, bias, ksize, stride, y, workspace, workspace_bytes,
                          stats_px_out, N, H, W, Cout, (hipStream_t)stream, 0);
}

// Split-bf16 mode of the fused pipeline: fp32 tensors; w3 = the pre-split weight planes of
// unet_pack_conv3x3_weights_bf16x3 (forward layout).  Same arguments otherwise.
extern "C" int unet_conv_in_fwd_bf16x3(const unet_act_src* s0, const unet_act_src* s1, float slope,
                                       const float* w, const uint16_t* w3, const float* bias,
                                       int ksize, int stride, float* y, void* workspace,
                                       size_t workspace_bytes, int* stats_px_out, int N, int H,
                                       int W, int Cout, unet_stream_t stream) {
  UNET_REQUIRE(w3 || ksize != 3, "conv_in_fwd_bf16x3: the pre-split weight planes are null");
  return conv_in_fwd_impl(s0, s1, slope, w, bias, ksize, stride, y, workspace, workspace_bytes,
                          stats_px_out, N, H, W, Cout, (hipStream_t)stream, 0, w3);
}

// Mixed-precision pipeline (BASELINE config 4): the sources (except the fp32 RGB image) and y
// are bf16 tensors in HBM, the operands are contracted on the bf16 matrix cores with fp32
// accumulation, the statistics come from the fp32 accumulators.  w / bias stay fp32 (master).
extern "C" int unet_conv_in_fwd_b16(const unet_act_src* s0, const unet_act_src* s1, float slope,
                                    const float* w, const float* bias, int ksize, int stride,
                                    uint16_t* y, void* workspace, size_t workspace_bytes,
                                    int* stats_px_out, int N, int H, int W, int Cout,
                                    unet_stream_t stream) {
  return conv_in_fwd_impl(s0, s1, slope, w, bias, ksize, stride, reinterpret_cast<float*>(y),
                          workspace, workspace_bytes, stats_px_out, N, H, W, Cout,
                          (hipStream_t)stream, 1);
}

// The same with the weights also given pre-rounded to bf16 (wb = [9][Cout][Cin] bf16: plane 0 of
// unet_pack_conv3x3_weights_batched's wf3, refreshed once per step): the patch kernel then
// stages its weight panels without the fp32 -> bf16 conversion and at half the L2 traffic.
// Results are bit-identical (the kernel rounds the same way).  wb may be null.
extern "C" int unet_conv_in_fwd_b16_wb(const unet_act_src* s0, const unet_act_src* s1,
                                       float slope, const float* w, const uint16_t* wb,
                                       const float* bias, int ksize, int stride, uint16_t* y,
                                       void* workspace, size_t workspace_bytes, int* stats_px_out,
                                       int N, int H, int W, int Cout, unet_stream_t stream) {
  return conv_in_fwd_impl(s0, s1, slope, w, bias, ksize, stride, reinterpret_cast<float*>(y),
                          workspace, workspace_bytes, stats_px_out, N, H, W, Cout,
                          (hipStream_t)stream, 1, ksize == 3 ? wb : nullptr);
}

static int conv_in_stats_finalize_impl(const float* y, void* workspace, size_t workspace_bytes,
                                       int stats_px, const float* gamma, const float* beta,
                                       float eps, const float* mask, float* mean, float* rstd,
                                       float* alpha_out, float* beta_out, int N, int HoWo, int Cout,
                                       unet_stream_t stream, int b16);

extern "C" int unet_conv_in_stats_finalize(const float* y, void* workspace, size_t workspace_bytes,
                                           int stats_px, const float* gamma, const float* beta,
                                           float eps, const float* mask, float* mean, float* rstd,
                                           float* alpha_out, float* beta_out, int N, int HoWo,
                                           int Cout, unet_stream_t stream) {
  return conv_in_stats_finalize_impl(y, workspace, workspace_bytes, stats_px, gamma, beta, eps,
                                     mask, mean, rstd, alpha_out, beta_out, N, HoWo, Cout, stream,
                                     0);
}

extern "C" int unet_conv_in_stats_finalize_b16(const uint16_t* y, void* workspace,
                                               size_t workspace_bytes, int stats_px,
                                               const float* gamma, const float* beta, float eps,
                                               const float* mask, float* mean, float* rstd,
                                               float* alpha_out, float* beta_out, int N, int HoWo,
                                               int Cout, unet_stream_t stream) {
  return conv_in_stats_finalize_impl(reinterpret_cast<const float*>(y), workspace, workspace_bytes,
                                     stats_px, gamma, beta, eps, mask, mean, rstd, alpha_out,
                                     beta_out, N, HoWo, Cout, stream, 1);
}

static int conv_in_stats_finalize_impl(const float* y, void* workspace, size_t workspace_bytes,
                                       int stats_px, const float* gamma, const float* beta,
                                       float eps, const float* mask, float* mean, float* rstd,
                                       float* alpha_out, float* beta_out, int N, int HoWo, int Cout,
                                       unet_stream_t stream, int b16) {
  UNET_REQUIRE(y && workspace && mean && rstd, "conv_in_stats_finalize: null pointer");
  UNET_REQUIRE(stats_px >= 0 && (stats_px == 0 || HoWo % stats_px == 0),
               "conv_in_stats_finalize: %d-pixel tiles do not cover %d pixels", stats_px, HoWo);
  if (stats_px > 0)   // the group scratch sits behind the summaries in the workspace
    return unet_in_finalize_tiles(workspace, HoWo / stats_px, stats_px, gamma, beta, eps, mask,
                                  mean, rstd, alpha_out, beta_out, N, HoWo, Cout,
                                  (hipStream_t)stream,
                                  reinterpret_cast<char*>(workspace) +
                                      stats_partial_bytes(N, HoWo, Cout));
  return unet_in_stats_masked(y, gamma, beta, eps, mask, mean, rstd, alpha_out, beta_out, workspace,
                              workspace_bytes, N, HoWo, Cout, (hipStream_t)stream, b16);
}

// ---------------------------------------------------------------------------
// Data gradient of conv3x3(upsample2x(a)) with respect to a, at LOW resolution:
//   g[q][ci] = sum_tap sum_co D[q][tap*Cout + co] * wd[tap][ci][co]
// (misc.hip upsample2x_bwd_taps_kernel has the derivation): a 1x1 gather-GEMM over the low-
// resolution pixels with K = 9*Cout, i.e. a quarter of the FLOPs of the 3x3 data gradient on the
// up-sampled grid, and it lands directly on the low-resolution tensor (no upsample2x_bwd pass).
// ---------------------------------------------------------------------------
static int conv3x3_up_bwd_data_impl(const float* D, const float* wd, int Cin_total, int ci_offset,
                                    float* g, int N, int h, int w, int Cout, int Ccols,
                                    int accumulate, unet_stream_t stream, int b16,
                                    unet_bwd_stats* bs = nullptr, const uint16_t* wdb = nullptr);

extern "C" int unet_conv3x3_up_bwd_data_bs(const float* D, const float* wd, int Cin_total,
                                           int ci_offset, float* g, int N, int h, int w, int Cout,
                                           int Ccols, int accumulate, unet_bwd_stats* bs,
                                           unet_stream_t stream) {
  return conv3x3_up_bwd_data_impl(D, wd, Cin_total, ci_offset, g, N, h, w, Cout, Ccols, accumulate,
                                  stream, 0, bs);
}

extern "C" int unet_conv3x3_up_bwd_data(const float* D, const float* wd, int Cin_total,
                                        int ci_offset, float* g, int N, int h, int w, int Cout,
                                        int Ccols, int accumulate, unet_stream_t stream) {
  return conv3x3_up_bwd_data_impl(D, wd, Cin_total, ci_offset, g, N, h, w, Cout, Ccols, accumulate,
                                  stream, 0);
}

extern "C" int unet_conv3x3_up_bwd_data_b16(const uint16_t* D, const float* wd, int Cin_total,
                                            int ci_offset, uint16_t* g, int N, int h, int w,
                                            int Cout, int Ccols, int accumulate,
                                            unet_stream_t stream) {
  return conv3x3_up_bwd_data_impl(reinterpret_cast<const float*>(D), wd, Cin_total, ci_offset,
                                  reinterpret_cast<float*>(g), N, h, w, Cout, Ccols, accumulate,
                                  stream, 1);
}

// wdb = the data-gradient weights pre-rounded to bf16 ([9][Cin_total][Cout], plane 0 of the
// pack's wd3): a plain bf16 GEMM over the 9 * Cout contiguous values of a D row.  bs may be null.
extern "C" int unet_conv3x3_up_bwd_data_bs_b16_wb(const uint16_t* D, const float* wd,
                                                  const uint16_t* wdb, int Cin_total,
                                                  int ci_offset, uint16_t* g, int N, int h, int w,
                                                  int Cout, int Ccols, int accumulate,
                                                  unet_bwd_stats* bs, unet_stream_t stream) {
  return conv3x3_up_bwd_data_impl(reinterpret_cast<const float*>(D), wd, Cin_total, ci_offset,
                                  reinterpret_cast<float*>(g), N, h, w, Cout, Ccols, accumulate,
                                  stream, 1, bs, wdb);
}

// the same with the BSTATS epilogue: g is final for the layer described by bs (bs->y bf16)
extern "C" int unet_conv3x3_up_bwd_data_bs_b16(const uint16_t* D, const float* wd, int Cin_total,
                                               int ci_offset, uint16_t* g, int N, int h, int w,
                                               int Cout, int Ccols, int accumulate,
                                               unet_bwd_stats* bs, unet_stream_t stream) {
  return conv3x3_up_bwd_data_impl(reinterpret_cast<const float*>(D), wd, Cin_total, ci_offset,
                                  reinterpret_cast<float*>(g), N, h, w, Cout, Ccols, accumulate,
                                  stream, 1, bs);
}

static int conv3x3_up_bwd_data_impl(const float* D, const float* wd, int Cin_total, int ci_offset,
                                    float* g, int N, int h, int w, int Cout, int Ccols,
                                    int accumulate, unet_stream_t stream, int b16,
                                    unet_bwd_stats* bs, const uint16_t* wdb) {
  const long long es = b16 ? 2 : 4;
  if (bs) bs->tiles_out = 0;
  const bool use_bs = bs && bs->y && bs->mean && bs->rstd && bs->gamma && bs->beta &&
                      bs->partial &&
                      bs->partial_bytes >= (size_t)N * ceil_div(h * w, 64) * Ccols * sizeof(float2);
  UNET_REQUIRE(D && wd && g, "conv3x3_up_bwd_data: null pointer");
  UNET_REQUIRE(Cout > 0 && Cout % 32 == 0 && Ccols > 0 && Ccols % 32 == 0 && ci_offset >= 0 &&
                   ci_offset + Ccols <= Cin_total && N > 0 && h > 0 && w > 0,
               "conv3x3_up_bwd_data: bad shape Cout=%d Ccols=%d slice %d of %d", Cout, Ccols,
               ci_offset, Cin_total);
  UNET_REQUIRE((long long)9 * Cout * Cin_total * 4 < (1LL << 31),
               "conv3x3_up_bwd_data: weights exceed 2 GiB");
  {
    const int nmax = batch_chunk(N, (long long)h * w * 9 * Cout * es);
    UNET_REQUIRE(nmax >= 1, "conv3x3_up_bwd_data: one image exceeds the 2 GiB buffer-descriptor range");
    if (nmax < N) {
      for (int nb = 0; nb < N; nb += nmax) {
        const int n = N - nb < nmax ? N - nb : nmax;
        const int rc = conv3x3_up_bwd_data_impl(
            reinterpret_cast<const float*>(reinterpret_cast<const char*>(D) +
                                           (size_t)nb * h * w * 9 * Cout * es),
            wd, Cin_total, ci_offset,
            reinterpret_cast<float*>(reinterpret_cast<char*>(g) + (size_t)nb * h * w * Ccols * es),
            n, h, w, Cout, Ccols, accumulate, stream, b16, nullptr, wdb);
        if (rc != UNET_OK) return rc;
      }
      return UNET_OK;
    }
  }
  IgemmParams p{};
  p.src0 = D; p.src1 = nullptr; p.C0 = Cout; p.C1 = 0;
  if (b16 && wdb) {   // the weights pre-rounded to bf16: the plain-GEMM form (conv_lowp.hip)
    p.w3 = reinterpret_cast<const __bf16*>(wdb);
    p.w3_plane = 9 * Cout * Cin_total;
    p.w3_bytes = (unsigned)((long long)p.w3_plane * 2);
  }
  p.src0_pitch = 9 * Cout; p.tap_cstride = Cout;
  p.w = wd; p.tap_stride = Cin_total * Cout; p.n_off = ci_offset; p.bias = nullptr;
  p.src0_bytes = (unsigned)((long long)N * h * w * 9 * Cout * es);
  p.src1_bytes = 0;
  p.w_bytes = (unsigned)((long long)9 * Cout * Cin_total * 4);
  p.out = g; p.ldo = Ccols; p.accumulate = accumulate;
  p.N = N; p.Hin = p.Hl = p.Hout = h; p.Win = p.Wl = p.Wout = w;
  p.Ncols = Ccols; p.sin = 1; p.sout = 1; p.py = p.px = 0;
  p.ntaps = 9; p.tapw[0] = p.tapw[1] = p.tapw[2] = 0;
  for (int t = 0; t < 9; ++t) set_tap(p, t, 0, 0, t);
  int bs_px = 0;
  if (use_bs) {
    p.bs_y = bs->y; p.bs_mean = bs->mean; p.bs_rstd = bs->rstd; p.bs_gamma = bs->gamma;
    p.bs_beta = bs->beta; p.bs_mask = bs->mask; p.slope = bs->slope;
    p.bs_partial = reinterpret_cast<float2*>(bs->partial);
  }
  const int rc = b16 ? dispatch_igemm_b16(p, (hipStream_t)stream, nullptr, use_bs ? &bs_px : nullptr)
                     : dispatch_igemm(p, (hipStream_t)stream, nullptr, use_bs ? &bs_px : nullptr);
  if (rc == UNET_OK && use_bs && bs_px > 0) bs->tiles_out = h * w / bs_px;
  return rc;
}

// unet_conv_up_in_fwd on bf16 tensors (mixed-precision pipeline): low [N][H/2][W/2][C0] and
// skip [N][H][W][C1] are bf16 and activated on load, y [N][H][W][Cout] is bf16; wf fp32 packed
// weights, w3 (may be null) their bf16-rounded plane.  1 from _supported = a tile shape exists.
extern "C" int unet_conv_up_in_fwd_b16_supported(int N, int H, int W, int C0, int C1, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || H % 4 || W % 32 || C0 < 32 || C0 % 32 || C1 < 0 || C1 % 32 ||
      Cout % 32)
    return 0;
  const long long M = (long long)N * H * W, mt = M / 128;
  if ((long long)H * W * (C0 > 4 * C1 ? C0 / 4 : C1) * 2 * N >= (1LL << 31)) return 0;
  if (Cout % 64 == 0 && mt * (Cout / 64) >= 256) return 1;
  if (Cout == 32 && H % 8 == 0 && M / 256 >= 256) return 1;
  return 0;
}

extern "C" int unet_conv_up_in_fwd_b16(const unet_act_src* low, const unet_act_src* skip,
                                       float slope, const float* wf, const uint16_t* w3,
                                       const float* bias, uint16_t* y, void* workspace,
                                       size_t workspace_bytes, int* stats_px_out, int N, int H,
                                       int W, int Cout, unet_stream_t stream) {
  UNET_REQUIRE(low && low->x && wf && w3 && y && workspace && stats_px_out,
               "conv_up_in_fwd_b16: null pointer (w3, the bf16 weight plane of unet_pack_w_batched, "
               "is required here)");
  const int C0 = low->C, C1 = skip ? skip->C : 0;
  UNET_REQUIRE(C1 == 0 || skip->x, "conv_up_in_fwd_b16: skip source is null");
  UNET_REQUIRE(low->alpha && low->beta && (!skip || (skip->alpha && skip->beta)),
               "conv_up_in_fwd_b16: the loader takes activated sources (alpha / beta set)");
  UNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && unet_conv_up_in_fwd_b16_supported(N, H, W, C0, C1, Cout),
               "conv_up_in_fwd_b16: shape N=%d %dx%d (%d+%d)->%d has no fused-upsample tile (query "
               "unet_conv_up_in_fwd_b16_supported and fall back to unet_upsample2x_in_fwd_b16)",
               N, H, W, C0, C1, Cout);
  if (workspace_bytes < unet_conv_in_fwd_workspace_bytes(N, H, W, Cout, 1)) {
    unet_set_error("conv_up_in_fwd_b16: workspace too small");
    return UNET_E_WORKSPACE;
  }
  IgemmParams p{};
  p.src0 = low->x; p.src1 = skip ? skip->x : nullptr; p.C0 = C0; p.C1 = C1;
  p.act0_alpha = low->alpha; p.act0_beta = low->beta;
  p.act1_alpha = skip ? skip->alpha : nullptr;
  p.act1_beta = skip ? skip->beta : nullptr;
  p.slope = slope;
  p.w = wf; p.tap_stride = Cout * (C0 + C1); p.n_off = 0; p.bias = bias;
  if (w3) {
    p.w3 = reinterpret_cast<const __bf16*>(w3);
    p.w3_plane = 9 * Cout * (C0 + C1);
    p.w3_bytes = (unsigned)((long long)p.w3_plane * 2);
  }
  p.src0_bytes = (unsigned)((long long)N * (H / 2) * (W / 2) * C0 * 2);
  p.src1_bytes = (unsigned)((long long)N * H * W * C1 * 2);
  p.w_bytes = (unsigned)((long long)9 * Cout * (C0 + C1) * 4);
  p.out = reinterpret_cast<float*>(y); p.ldo = Cout; p.accumulate = 0;
  p.N = N; p.Hin = H; p.Win = W; p.Hl = p.Hout = H; p.Wl = p.Wout = W;
  p.Ncols = Cout;
  p.stats = reinterpret_cast<float2*>(workspace);
  fill_fwd_taps(p, 1);
  int px = 0;
  const int rc = launch_patch_b16_up_auto(p, (hipStream_t)stream, &px);
  UNET_REQUIRE(rc != 1, "conv_up_in_fwd_b16: no tile fits");
  if (rc != UNET_OK) return rc;
  *stats_px_out = px;
  return UNET_OK;
}

// RGB stem straight from the dataset's uint8 HWC image: normalisation fused into the loader
// (Our_UNet/src/train.py:303-308 + the first Conv2d of encoder_stages.0), statistics epilogue as
// unet_conv_in_fwd.  Needs W % 128 == 0 (the raw-row form); other widths: unet_preprocess_u8.
extern "C" int unet_stem_u8_fwd(const uint8_t* image_hwc, const float* mean3, const float* std3,
                                const float* wf, const float* bias, float* y, void* workspace,
                                size_t workspace_bytes, int* stats_px_out, int N, int H, int W,
                                int Cout, unet_stream_t stream) {
  UNET_REQUIRE(image_hwc && mean3 && std3 && wf && y && workspace && stats_px_out,
               "stem_u8_fwd: null pointer");
  UNET_REQUIRE(N > 0 && H > 0 && W > 0 && W % STEM_ROW_PIX == 0 && Cout > 0 && Cout % 32 == 0,
               "stem_u8_fwd: needs W %% %d == 0 and Cout %% 32 == 0 (got W=%d Cout=%d)",
               STEM_ROW_PIX, W, Cout);
  if (workspace_bytes < unet_conv_in_fwd_workspace_bytes(N, H, W, Cout, 1)) {
    unet_set_error("stem_u8_fwd: workspace too small");
    return UNET_E_WORKSPACE;
  }
  StemNorm nm;
  for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.std[c] = std3[c]; }
  launch_stem_fwd_rows<unsigned char, float>(image_hwc, wf, bias, y, N, H, W, Cout,
                                             reinterpret_cast<float2*>(workspace), nm,
                                             (hipStream_t)stream);
  UNET_CHECK_LAUNCH("conv_stem_fwd(u8)");
  *stats_px_out = STEM_ROW_PIX;
  return UNET_OK;
}

// ---------------------------------------------------------------------------
// y = conv3x3(cat(upsample2x(act(low)), act(skip))) + bias with the bilinear up-sampling done in
// the patch loader (conv_patch.hip conv_patch_up_kernel): UpBlock.forward without the up-sampled
// tensor and without the concatenation (Our_UNet/models/unet.py:215-231).  low->x is
// [N][H/2][W/2][C0], skip->x [N][H][W][C1]; workspace / *stats_px_out as unet_conv_in_fwd.
// ---------------------------------------------------------------------------
extern "C" int unet_conv_up_in_fwd_supported(int N, int H, int W, int C0, int C1, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || H % 4 || W % 32 || C0 < 32 || C0 % 32 || C1 < 0 || C1 % 32 ||
      Cout % 32)
    return 0;
  const long long M = (long long)N * H * W, mt = M / 128;
  if ((long long)H * W * (C0 > 4 * C1 ? C0 / 4 : C1) * 4 * N >= (1LL << 31)) return 0;
  if (Cout % 128 == 0 && mt * (Cout / 128) >= 512) return 1;
  if (Cout % 64 == 0 && mt * (Cout / 64) >= 512) return 1;
  if (Cout == 32 && H % 8 == 0 && M / 256 >= 512) return 1;
  // (the Winograd kernel of the (64 + 32) -> 32 layer forced for every shape: tests)
  if (Cout == 32 && C0 == 64 && C1 == 32 && H % 8 == 0 && c32_winograd_flag() == 2) return 1;
  return 0;
}

// 1 when unet_conv_up_in_fwd runs this shape (both sources activated on load) on the Winograd
// kernel of csrc/conv_c32.hip (conv_wino_up32_kernel): 16/36 of the direct matrix FLOPs
extern "C" int unet_conv_up_c32_is_winograd(int N, int H, int W, int C0, int C1, int Cout) {
  const int f = c32_winograd_flag();
  if (!(f && N > 0 && C0 == 64 && C1 == 32 && Cout == 32 && H > 0 && W > 0 && H % 8 == 0 &&
        W % 32 == 0 && (long long)N * H * W * 128 < (1LL << 31)))
    return 0;
  return f == 2 || (long long)N * (H / 8) * (W / 32) >= 256;
}

extern "C" int unet_conv_up_in_fwd(const unet_act_src* low, const unet_act_src* skip, float slope,
                                   const float* wf, const float* bias, float* y, void* workspace,
                                   size_t workspace_bytes, int* stats_px_out, int N, int H, int W,
                                   int Cout, unet_stream_t stream) {
  UNET_REQUIRE(low && low->x && wf && y && workspace && stats_px_out,
               "conv_up_in_fwd: null pointer");
  const int C0 = low->C, C1 = skip ? skip->C : 0;
  UNET_REQUIRE(C1 == 0 || skip->x, "conv_up_in_fwd: skip source is null");
  UNET_REQUIRE((!low->alpha || low->beta) && (!skip || !skip->alpha || skip->beta),
               "conv_up_in_fwd: alpha without beta");
  UNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && unet_conv_up_in_fwd_supported(N, H, W, C0, C1, Cout),
               "conv_up_in_fwd: shape N=%d %dx%d (%d+%d)->%d has no fused-upsample tile (query "
               "unet_conv_up_in_fwd_supported and fall back to unet_upsample2x_in_fwd)",
               N, H, W, C0, C1, Cout);
  if (workspace_bytes < unet_conv_in_fwd_workspace_bytes(N, H, W, Cout, 1)) {
    unet_set_error("conv_up_in_fwd: workspace too small");
    return UNET_E_WORKSPACE;
  }
  IgemmParams p{};
  p.src0 = low->x; p.src1 = skip ? skip->x : nullptr; p.C0 = C0; p.C1 = C1;
  p.act0_alpha = low->alpha; p.act0_beta = low->alpha ? low->beta : nullptr;
  p.act1_alpha = skip ? skip->alpha : nullptr;
  p.act1_beta = (skip && skip->alpha) ? skip->beta : nullptr;
  p.slope = slope;
  p.w = wf; p.tap_stride = Cout * (C0 + C1); p.n_off = 0; p.bias = bias;
  p.src0_bytes = (unsigned)((long long)N * (H / 2) * (W / 2) * C0 * 4);
  p.src1_bytes = (unsigned)((long long)N * H * W * C1 * 4);
  p.w_bytes = (unsigned)((long long)9 * Cout * (C0 + C1) * 4);
  p.out = y; p.ldo = Cout; p.accumulate = 0;
  p.N = N; p.Hin = H; p.Win = W; p.Hl = p.Hout = H; p.Wl = p.Wout = W;
  p.Ncols = Cout;
  p.stats = reinterpret_cast<float2*>(workspace);
  fill_fwd_taps(p, 1);
  int px = 0;
  const int rc = launch_patch_up_auto(p, (hipStream_t)stream, &px);
  UNET_REQUIRE(rc != 1, "conv_up_in_fwd: no tile fits");
  if (rc != UNET_OK) return rc;
  *stats_px_out = px;
  return UNET_OK;
}

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


template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const IgemmParams p) {
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
  float* As = smem;
  float* Bs = smem + 2 * A_TILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    } else {
      a_nb[i] = 0;
      a_iy[i] = -(1 << 24);
      a_ix[i] = 0;
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

  const int KS = p.ntaps * (Ktot / BK);

  auto load_tiles = [&](int t, int chunk) {
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int oy = (int)(e & 3u) - 1, ox = (int)((e >> 2) & 3u) - 1;
    const int wt = (int)(e >> 4);
    const int c = chunk * BK;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? p.C0 : p.C1;
    const int coff = (first ? c : c - p.C0) + lseg * 4;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int iy = a_iy[i] + oy, ix = a_ix[i] + ox;
      const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
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
    for (int i = 0; i < A_PASSES; ++i) *reinterpret_cast<f32x4*>(Ab + ROWS * i * LDA) = ra[i];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bb + ROWS * j * LDA) = rb[j];
  };

  int t_next = 0, chunk_next = 0;
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
#ifdef UNET_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
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
#ifdef UNET_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    store_tiles(buf ^ 1);
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
    __builtin_amdgcn_sched_group_barrier(0x200, A_PASSES + B_PASSES, 0);
    __syncthreads();
  }

  // ---- epilogue: D row = (reg&3) + 8*(reg>>2) + 4*lh, column = li ----
  const bool direct = (p.sout == 1 && p.Hl == p.Hout && p.Wl == p.Wout);
#pragma unroll
  for (int n = 0; n < TN; ++n) {
    const int col = n0 + wn0 + n * 32 + li;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      float* o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int mg = m0 + row;
        size_t opix = (size_t)mg;
        if (!direct) {
          const int nn = mg / HlWl;
          const int rr = mg - nn * HlWl;
          const int a = rr / p.Wl;
          const int b = rr - a * p.Wl;
          opix = ((size_t)nn * p.Hout + (a * p.sout + p.py)) * p.Wout + (b * p.sout + p.px);
        }
        o[r] = mg < M ? p.out + opix * p.ldo + col : nullptr;
      }
      store_block16(o, acc[m][n], bv, p.accumulate);
    }
  }
}

template <int BM, int BN, int WM, int WN, int BK = 32>
int launch_igemm(const IgemmParams& p, hipStream_t stream) {
  constexpr int LDA = BK + 4;
  constexpr size_t lds = 2 * (size_t)(BM + BN) * LDA * sizeof(float);
  static bool attr_set = false;
  auto kern = conv_igemm_kernel<BM, BN, WM, WN, BK>;
  if (!attr_set) {
    UNET_HIP_CALL(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const long long tiles = ceil_div64(M, BM) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
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
template <int BM, int BN, int WM, int WN, bool PW>
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
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int r = lrow + 32 * i;
      const int x = x0 - 1 + r;
      const bool ok = yok && r < AR && (unsigned)x < (unsigned)p.Win;
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
    for (int i = 0; i < A_PASSES; ++i)
      if (32 * (i + 1) <= AR || lrow + 32 * i < AR) *reinterpret_cast<f32x4*>(Ab + 32 * i * LDA) = ra[i];
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
      store_tiles(buf ^ 1);
      __syncthreads();
    }
    // tile epilogue (stores drain while the next tile's K steps run)
    const int m0 = tile * BM;
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
        store_block16(o, acc[m][n], bv, p.accumulate);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, bool PW>
int launch_igemm_rf(const IgemmParams& p, int flip, hipStream_t stream) {
  constexpr int LDA = 36;
  constexpr size_t lds = PW ? (size_t)(2 * (BM + 2) + 9 * BN) * LDA * sizeof(float)
                            : 2 * (size_t)((BM + 2) + 3 * BN) * LDA * sizeof(float);
  static bool attr_set = false;
  auto kern = conv_igemm_rf_kernel<BM, BN, WM, WN, PW>;
  if (!attr_set) {
    UNET_HIP_CALL(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  IgemmParams q = p;
  q.sin = flip;
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
  static const int off = getenv("UNET_NO_ROWFUSE") ? 1 : 0;
  return !off && p.ntaps == 9 && p.sout == 1 && p.Hl == p.Hin && p.Wl == p.Win &&
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
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float* o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      o[r] = base[r] != ~(size_t)0
                 ? p.out + (base[r] + (size_t)(c >> 1) * p.Wout + (c & 1)) * p.ldo + col
                 : nullptr;
    store_block16(o, acc[c], 0.f, p.accumulate);
  }
}

int launch_dgrad_s2(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)(128 + 4 * 32) * 36 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    UNET_HIP_CALL(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dgrad_s2_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const long long tiles = ceil_div64(M, 128) * (p.Ncols / 32);
  hipLaunchKernelGGL(conv_dgrad_s2_kernel, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_dgrad_s2");
  return UNET_OK;
}

}  // namespace

int dispatch_igemm(const IgemmParams& p, hipStream_t stream) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  if (patch_f32_applicable(p)) {   // conv_patch.hip; 1 = no tile shape fits this launch
    const int rc = launch_patch_f32_auto(p, stream);
    if (rc != 1) return rc;
  }
  // Largest tile that still yields >= 256 workgroups (one per CU); otherwise the
  // small 64x64 tile.
  static const int bk16 = getenv("UNET_IGEMM_BK16") ? 1 : 0;
  if (nc % 128 == 0 && ceil_div64(M, 128) * (nc / 128) >= 256)
    return bk16 ? launch_igemm<128, 128, 64, 64, 16>(p, stream)
                : launch_igemm<128, 128, 64, 64>(p, stream);
  if (nc % 64 == 0 && ceil_div64(M, 128) * (nc / 64) >= 256)
    return launch_igemm<128, 64, 64, 32>(p, stream);
  if (nc % 64 == 0 && M <= 128 * 256) return launch_igemm<64, 64, 32, 32>(p, stream);
  return launch_igemm<128, 32, 32, 32>(p, stream);
}

namespace {

// ---------------------------------------------------------------------------
// RGB stem (Cin = 3): the whole K = 27 im2col row is gathered into LDS and one
// K step of 28 feeds the matrix cores.  HBM-bound (writes 32 channels per pixel).
// ---------------------------------------------------------------------------
constexpr int STEM_PIX = 256;  // pixels per block
constexpr int STEM_LDK = 29;   // odd row stride: conflict-free column reads

__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ wf,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ y, int N, int H,
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
      if (mg < M) y[(size_t)mg * Cout + co0 + li] = acc[m][r] + bv;
    }
}

// Row form of the stem for images whose width is a multiple of 128: a block's 128 pixels lie in
// one image row, so the three input rows are copied raw (coalesced) as R[ky][3*(col+1) + ci]
// and the im2col element k = 9*ky + (3*kx + ci) of pixel px is R[ky][3*px + 3*kx + ci] - no
// per-element gather.  Lanes run along pixels at a stride of 3 words (conflict-free).
constexpr int STEM_ROW_PIX = 128;
constexpr int STEM_ROW_PITCH = 393;

__global__ __launch_bounds__(256) void conv_stem_fwd_rows_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ wf,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ y, int N,
                                                                 int H, int W, int Cout) {
  __shared__ float Rw[3 * STEM_ROW_PITCH];
  __shared__ float B[28 * 32];
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
      v = x[((size_t)n * HW + (size_t)iy * W + x0 - 1) * 3 + j];
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
    y[(size_t)(m0 + row) * Cout + co0 + li] = acc[r] + bv;
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
      dim3 grid((unsigned)(M / STEM_ROW_PIX), Cout / 32);
      hipLaunchKernelGGL(conv_stem_fwd_rows_kernel, grid, dim3(256), 0, stream, x0, wf, bias, y, N,
                         H, W, Cout);
    } else {
      dim3 grid((unsigned)ceil_div64(M, STEM_PIX), Cout / 32);
      hipLaunchKernelGGL(conv_stem_fwd_kernel, grid, dim3(256), 0, stream, x0, wf, bias, y, N, H,
                         W, Cout);
    }
    UNET_CHECK_LAUNCH("conv_stem_fwd");
    return UNET_OK;
  }
  UNET_REQUIRE(C0 > 0 && C0 % 32 == 0 && C1 >= 0 && C1 % 32 == 0,
               "conv3x3_fwd: channel counts (%d,%d) must be multiples of 32", C0, C1);
  UNET_REQUIRE(C1 == 0 || x1, "conv3x3_fwd: x1 is null with C1=%d", C1);
  IgemmParams p{};
  p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1;
  p.w = wf; p.tap_stride = Cout * (C0 + C1); p.n_off = 0; p.bias = bias;
  UNET_REQUIRE((long long)N * H * W * (C0 > C1 ? C0 : C1) * 4 < (1LL << 31) &&
                   (long long)9 * Cout * (C0 + C1) * 4 < (1LL << 31),
               "conv3x3_fwd: tensor exceeds the 2 GiB buffer-descriptor range");
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
                                 const uint16_t* wd3 = nullptr) {
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
  IgemmParams p{};
  p.src0 = dy; p.src1 = nullptr; p.C0 = Cout; p.C1 = 0;
  p.w = wd; p.tap_stride = Cin_total * Cout; p.n_off = ci_offset; p.bias = nullptr;
  UNET_REQUIRE((long long)N * Ho * Wo * Cout * 4 < (1LL << 31) &&
                   (long long)9 * Cout * Cin_total * 4 < (1LL << 31),
               "conv3x3_bwd_data: tensor exceeds the 2 GiB buffer-descriptor range");
  p.src0_bytes = (unsigned)((long long)N * Ho * Wo * Cout * 4);
  p.src1_bytes = 0;
  p.w_bytes = (unsigned)((long long)9 * Cout * Cin_total * 4);
  p.out = dx; p.ldo = Ccols; p.accumulate = accumulate;
  p.N = N; p.Hin = Ho; p.Win = Wo;
  p.Hout = H; p.Wout = W;
  p.Ncols = Ccols;
  p.w3 = reinterpret_cast<const __bf16*>(wd3);
  p.w3_plane = 9 * Cout * Cin_total;
  p.w3_bytes = (unsigned)((long long)3 * p.w3_plane * 2);
  p.sin = 1;
  if (stride == 1) {
    p.Hl = H; p.Wl = W; p.sout = 1; p.py = p.px = 0;
    p.ntaps = 9;
    p.tapw[0] = p.tapw[1] = p.tapw[2] = 0;
    for (int t = 0; t < 9; ++t) set_tap(p, t, 1 - t / 3, 1 - t % 3, t);
    if (prec == 1) return dispatch_igemm_bf16(p, stream);
    if (prec == 3) return dispatch_igemm_split(p, stream);
    if (rf_applicable(p) && (Cout == 32 || !patch_f32_applicable(p) || p.Hin % 8 != 0))
      return (Cout == 32) ? launch_igemm_rf<128, 32, 32, 32, true>(p, 1, stream)
                          : launch_igemm_rf<128, 32, 32, 32, false>(p, 1, stream);
    return dispatch_igemm(p, stream);
  }
  // stride 2: dx[2a+py][2b+px] = sum over ky with (py+1-ky) even of dy[a + (py+1-ky)/2][..]
  p.Hl = H / 2; p.Wl = W / 2; p.sout = 2;
  {
    static const int per_class = getenv("UNET_S2_PER_CLASS") ? 1 : 0;
    // one launch for all four parity classes when there are enough tiles to fill the chip
    const long long tiles = ceil_div64((long long)N * p.Hl * p.Wl, 128) * (Ccols / 32);
    if (prec != 1 && !per_class && tiles >= 512) { p.py = p.px = 0; p.ntaps = 9; return launch_dgrad_s2(p, stream); }
  }
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
      int rc = prec == 1 ? dispatch_igemm_bf16(p, stream)
                         : (prec == 3 ? dispatch_igemm_split(p, stream) : dispatch_igemm(p, stream));
      if (rc != UNET_OK) return rc;
    }
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
  IgemmParams p{};
  p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1;
  p.w = w; p.tap_stride = Cout * (C0 + C1); p.n_off = 0; p.bias = bias;
  UNET_REQUIRE((long long)N * H * W * (C0 > C1 ? C0 : C1) * 4 < (1LL << 31),
               "conv1x1_fwd: tensor exceeds the 2 GiB buffer-descriptor range");
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
  IgemmParams p{};
  p.src0 = dy; p.src1 = nullptr; p.C0 = Cout; p.C1 = 0;
  p.w = wT; p.tap_stride = Cin_total * Cout; p.n_off = ci_offset; p.bias = nullptr;
  UNET_REQUIRE((long long)N * H * W * Cout * 4 < (1LL << 31),
               "conv1x1_bwd_data: tensor exceeds the 2 GiB buffer-descriptor range");
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

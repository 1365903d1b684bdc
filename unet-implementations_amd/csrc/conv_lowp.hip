// conv_lowp.hip — the gather-GEMM of conv_igemm.hip on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation): bf16 mixed precision (BASELINE config 4) and the
// split-bf16 form used where the patch-staged kernel does not apply (stride-2 forward, per-class
// stride-2 data gradient).
#include "conv_params.h"
#include <stdlib.h>

namespace unet_conv {
namespace {

// ---------------------------------------------------------------------------
// bf16 mixed-precision variant (BASELINE config 4 / SURVEY a15): tensors stay fp32 in HBM,
// the loader rounds both operands to bf16 (v_cvt_pk_bf16_f32) while staging them into LDS
// and the contraction runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (16x the fp32
// matrix rate, so the kernel turns load/L2-bound).  Same gather-GEMM, tap table, tiles and
// epilogue as conv_igemm_kernel; LDS rows hold 32 bf16 + 8 pad (80 B: conflict-free b128).
// ---------------------------------------------------------------------------
// TS / TO: storage type of the source / output activations (float, or __bf16 in the real
// mixed-precision pipeline where the layer tensors live in HBM as bf16).  FUSED: activation on
// load + statistics epilogue of the fused layer pipeline (IgemmParams), as conv_igemm_kernel.
// KG > 1 (the 1/32-resolution layers, M = N*16*16 rows: one 64x64 tile per CU, and a lone 4-wave
// block cannot hide the load latency of its 144-step K loop - round 3 measured this kernel at
// 0.66 ms over those launches against 0.31 ms for its fp32 sibling, which has K groups): KG
// groups of four waves share the tile, group g runs K steps [g, g+1) * KS / KG through its own
// pair of LDS stages, the partial accumulators are summed through LDS in group order before
// the common epilogue (deterministic) - as conv_igemm_kernel (conv_igemm.hip).
// !FUSED with p.bs_partial: the BSTATS epilogue (the output is FINAL for the layer behind it:
// per-tile sums of that layer's InstanceNorm backward, from the fp32 accumulators and the stored
// raw outputs y) - as conv_igemm_kernel / conv_patch_b16_kernel.
// DENSE (round 4; the low-resolution data gradient of the up-sampled operand, "channel taps":
// A row m = the 9 * C0 contiguous bf16 of D[m][tap][c], every row valid, no geometry): the
// contraction is a plain GEMM over K = 9 * C0, so the K step is 64 wide - a whole 128-byte line
// of every A row per step instead of half of one (the other half came back eight steps later,
// from L2 or further), 16-byte loads on both operands, the weights from their bf16 plane (p.w3,
// no conversion, half the L2 traffic), half the barriers.  A K step may straddle taps (C0 = 32)
// and the last one may be ragged (9 * 32 = 4.5 steps): lanes past K read zeros.
template <int BM, int BN, int WM, int WN, typename TS = float, typename TO = float,
          bool FUSED = false, int KG = 1, int MODE = 0>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 2 : 1) void conv_igemm_bf16_kernel(const IgemmParams p) {
  // MODE 1 (WIDE): the gather form on the same 64-wide K steps (C0, C1 multiples of 64): raw
  // 16-byte loads of 8 bf16 channels of a tap, weights from the bf16 plane - the data gradients
  // of the 1/32-resolution stage.  MODE 2: DENSE.
  constexpr bool DENSE = MODE == 2, WIDE = MODE != 0;
  static_assert(!WIDE || sizeof(TS) == 2, "WIDE / DENSE: bf16 rows");
  static_assert(!DENSE || !FUSED, "DENSE: no activation");
  constexpr int BK = WIDE ? 64 : 32;
  constexpr int LDA = BK + 8;  // bf16 elements per LDS row
  constexpr int EPS = WIDE ? 8 : 4;   // elements per loader segment
  constexpr int SEGS = BK / EPS;       // loader segments per tile row
  constexpr int ROWS = 256 / SEGS;     // tile rows covered by one loader pass
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int A_PASSES = BM / ROWS;
  constexpr int B_PASSES = BN / ROWS;
  constexpr int A_TILE = BM * LDA, B_TILE = BN * LDA;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];
  // K group: uniform per wave, kept in an SGPR so the K-step bookkeeping stays scalar
  const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  const bool lead = KG == 1 || grp == 0;
  __bf16* As = smem_h + grp * 2 * (A_TILE + B_TILE);
  __bf16* Bs = As + 2 * A_TILE;

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
  // DENSE: the bf16 weight plane; A rows by byte offset (pitch = 9 * C0 elements)
  const __amdgpu_buffer_rsrc_t rsw3 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(WIDE ? p.w3 : nullptr), 0, WIDE ? (int)p.w3_bytes : 0, 0x00020000);
  const int Kd = 9 * p.C0;                          // DENSE: the GEMM's K
  const int c0_log2 = __builtin_ctz((unsigned)p.C0);   // (launcher: C0 is a power of two)
  unsigned a_off[DENSE ? A_PASSES : 1];
  if constexpr (DENSE) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      a_off[i] = (unsigned)((m0 + lrow + ROWS * i) * Kd + lseg * 8) * 2u;
  }

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  // NSET = 2 (the data-gradient forms, round 4): the tiles of step k + 2 are in flight while
  // those of k + 1 wait for their LDS stage; with one set every step waited for the loads it had
  // just issued - 8 MFMAs a step against an L2 round trip.  (The fused forward keeps one set: its
  // per-row activation coefficients would double with the tiles.)
  constexpr int NSET = FUSED ? 1 : 2;
  f32x4 ra[NSET][A_PASSES], rb[NSET][B_PASSES];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // K steps of this group (the launcher checks % KG)
  const int KS = DENSE ? ((Kd + BK - 1) / BK) / KG : p.ntaps * (Ktot / BK) / KG;
  constexpr int CP = FUSED ? A_PASSES : 1;
  constexpr int CV = WIDE ? 2 : 1;     // coefficient quads per staged segment (8 / 4 channels)
  f32x4 ca[CP][CV], cb[CP][CV];
  float cs = 1.f;
  unsigned okm = 0;

  auto load_tiles = [&](int t, int chunk, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    if constexpr (DENSE) {   // `chunk` = the K step (t unused): k = chunk * 64 + lseg * 8 ..+7
      const int k = chunk * BK + lseg * 8;
      const unsigned kill = k < Kd ? 0u : 0x80000000u;
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i)
        ra[SET][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs0, a_off[i] | kill, (unsigned)(chunk * BK) * 2u, 0));
      const int tap = k >> c0_log2;
      const unsigned woff =
          (unsigned)(tap * p.tap_stride + (p.n_off + n0 + lrow) * Ktot + (k - (tap << c0_log2))) * 2u | kill;
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j)
        rb[SET][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsw3, woff + (unsigned)(ROWS * j * Ktot) * 2u, 0, 0));
      return;
    }
    const unsigned tw = (t < 4) ? p.tapw[0] : (t < 8 ? p.tapw[1] : p.tapw[2]);
    const unsigned e = (tw >> ((t & 3) * 8)) & 0xffu;
    const int oy = (int)(e & 3u) - 1, ox = (int)((e >> 2) & 3u) - 1;
    const int wt = (int)(e >> 4);
    const int c = chunk * BK;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? (p.src0_pitch ? p.src0_pitch : p.C0) : p.C1;
    const int cch = (first ? c : c - p.C0) + lseg * EPS;   // channel within the K slice
    const int coff = cch + wt * p.tap_cstride;
    if (FUSED) {
      const float* al = first ? p.act0_alpha : p.act1_alpha;
      const float* be = first ? p.act0_beta : p.act1_beta;
      const int Cc = first ? p.C0 : p.C1;
      if (al) {   // uniform
#pragma unroll
        for (int i = 0; i < CP; ++i)
#pragma unroll
          for (int v = 0; v < CV; ++v) {
            ca[i][v] = *reinterpret_cast<const f32x4*>(al + (size_t)a_img[i] * Cc + cch + 4 * v);
            cb[i][v] = *reinterpret_cast<const f32x4*>(be + (size_t)a_img[i] * Cc + cch + 4 * v);
          }
        cs = p.slope;
      } else {    // plain source: z = v, slope 1 = identity
#pragma unroll
        for (int i = 0; i < CP; ++i)
#pragma unroll
          for (int v = 0; v < CV; ++v) {
            ca[i][v] = f32x4{1.f, 1.f, 1.f, 1.f};
            cb[i][v] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        cs = 1.f;
      }
      okm = 0;
    }
    if constexpr (WIDE) {   // raw bf16 bits: 8 channels of the tap per lane, both operands
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        const int iy = a_iy[i] + oy, ix = a_ix[i] + ox;
        const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        if (FUSED) okm |= (ok ? 1u : 0u) << i;
        ra[SET][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rs, (unsigned)((a_nb[i] + iy * p.Win + ix) * Cs + coff) * 2u | (ok ? 0u : 0x80000000u), 0, 0));
      }
      const unsigned woff =
          (unsigned)(wt * p.tap_stride + (p.n_off + n0 + lrow) * Ktot + c + lseg * 8) * 2u;
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j)
        rb[SET][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsw3, woff + (unsigned)(ROWS * j * Ktot) * 2u, 0, 0));
      return;
    }
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int iy = a_iy[i] + oy, ix = a_ix[i] + ox;
      const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      if (FUSED) okm |= (ok ? 1u : 0u) << i;
      // invalid lanes get bit 31 set: beyond num_records (< 2 GiB), the load returns 0
      ra[SET][i] = buf_ld4<TS>(rs, (unsigned)((a_nb[i] + iy * p.Win + ix) * Cs + coff),
                               ok ? 0u : 0x80000000u);
    }
    const unsigned woff = wrow_off + (unsigned)(wt * p.tap_stride + c) * 4u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsw, woff + (unsigned)(ROWS * j * Ktot) * 4u, 0, 0);
      rb[SET][j] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto to_bf16 = [](const f32x4 v) {
    bf16x4 h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    return h;
  };
  auto store_tiles = [&](int buf, auto setc) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    __bf16* Ab = As + buf * A_TILE + lrow * LDA + lseg * EPS;
    __bf16* Bb = Bs + buf * B_TILE + lrow * LDA + lseg * EPS;
    if constexpr (WIDE) {   // raw bf16 bits, 16 bytes a lane
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        if constexpr (FUSED) {   // unpack, activate (fp32, as the 32-wide form), round again
          const i32x4 w = __builtin_bit_cast(i32x4, ra[SET][i]);
          auto lo = [](int x) { return __builtin_bit_cast(float, x << 16); };
          auto hi = [](int x) { return __builtin_bit_cast(float, x & (int)0xffff0000); };
          const bool ok = (okm >> i) & 1u;
          const f32x4 v0 = act4(f32x4{lo(w[0]), hi(w[0]), lo(w[1]), hi(w[1])}, ca[i][0], cb[i][0], cs, ok);
          const f32x4 v1 = act4(f32x4{lo(w[2]), hi(w[2]), lo(w[3]), hi(w[3])}, ca[i][1], cb[i][1], cs, ok);
          *reinterpret_cast<bf16x4*>(Ab + ROWS * i * LDA) = to_bf16(v0);
          *reinterpret_cast<bf16x4*>(Ab + ROWS * i * LDA + 4) = to_bf16(v1);
        } else {
          *reinterpret_cast<f32x4*>(Ab + ROWS * i * LDA) = ra[SET][i];
        }
      }
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<f32x4*>(Bb + ROWS * j * LDA) = rb[SET][j];
      return;
    }
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      if (FUSED) ra[SET][i] = act4(ra[SET][i], ca[i][0], cb[i][0], cs, (okm >> i) & 1u);
      *reinterpret_cast<bf16x4*>(Ab + ROWS * i * LDA) = to_bf16(ra[SET][i]);
    }
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<bf16x4*>(Bb + ROWS * j * LDA) = to_bf16(rb[SET][j]);
  };

  int t_next = 0, chunk_next = 0;
  if (KG > 1) {   // tap-fastest order: step = chunk * ntaps + t
    chunk_next = grp * KS / p.ntaps;
    t_next = grp * KS - chunk_next * p.ntaps;
  }
  if constexpr (DENSE) { t_next = 0; chunk_next = grp * KS; }   // chunk_next = the K step
  auto advance = [&](bool on) {  // branch-free: keeps the K step a single basic block
    if constexpr (DENSE) { chunk_next += on ? 1 : 0; return; }
    const int tn = t_next + 1;
    const bool wrap = tn == p.ntaps;
    t_next = on ? (wrap ? 0 : tn) : t_next;
    chunk_next = on ? chunk_next + (wrap ? 1 : 0) : chunk_next;
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, NSET - 1>;
  load_tiles(t_next, chunk_next, S0{});
  advance(KS > 1);
  if constexpr (NSET == 2) {   // step 1's tiles (KS == 1: step 0's again, never stored)
    load_tiles(t_next, chunk_next, S1{});
    advance(KS > 2);
  }
  store_tiles(0, S0{});
  __syncthreads();

  // fragment addresses: lane (li, lh) reads the 8 consecutive k = 16*kk + 8*lh .. +7 of row li
  // (the natural A/B operand map of v_mfma_f32_32x32x16_bf16)
  const int frag_off = li * LDA + 8 * lh;
  // one K step; LOADSET receives the tiles of step ks + NSET, STORESET holds those of ks + 1
  auto k_step = [&](int ks, auto loadc, auto storec) __attribute__((always_inline)) {
    const int buf = ks & 1;
    load_tiles(t_next, chunk_next, loadc);           // (past the end: the last step's tiles again)
    advance(ks + NSET + 1 < KS);
    const __bf16* Ab = As + buf * A_TILE + wm0 * LDA + frag_off;
    const __bf16* Bb = Bs + buf * B_TILE + wn0 * LDA + frag_off;
    bf16x8 a[2][TM], b[2][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) a[0][m] = *reinterpret_cast<const bf16x8*>(Ab + m * 32 * LDA);
#pragma unroll
    for (int n = 0; n < TN; ++n) b[0][n] = *reinterpret_cast<const bf16x8*>(Bb + n * 32 * LDA);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < BK / 16) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
          a[nxt][m] = *reinterpret_cast<const bf16x8*>(Ab + m * 32 * LDA + (kk + 1) * 16);
#pragma unroll
        for (int n = 0; n < TN; ++n)
          b[nxt][n] = *reinterpret_cast<const bf16x8*>(Bb + n * 32 * LDA + (kk + 1) * 16);
      }
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][m], b[cur][n], acc[m][n], 0, 0, 0);
    }
    store_tiles(buf ^ 1, storec);
    // buffer loads first (they are the critical path here), fragment reads ahead of the MFMAs
    __builtin_amdgcn_sched_group_barrier(0x020, A_PASSES + B_PASSES, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, (BK / 16) * (TM + TN), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, (BK / 16) * TM * TN, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, A_PASSES + B_PASSES, 0);
    __syncthreads();
  };
  if constexpr (NSET == 2) {
    for (int ks = 0; ks < KS; ks += 2) {
      k_step(ks, S0{}, S1{});
      if (ks + 1 < KS) k_step(ks + 1, S1{}, S0{});
    }
  } else {
    for (int ks = 0; ks < KS; ++ks) k_step(ks, S0{}, S0{});
  }

  // the K loop ended on a barrier: every LDS stage is free scratch from here on
  float* fsm = reinterpret_cast<float*>(smem_h);
  float* scratch = fsm;
  if (KG > 1) {   // partial sums of groups 1.. in [group][wave][register][lane] order
    constexpr int PART = 256 * 16 * TM * TN;
    static_assert(((KG - 1) * PART + 2 * (BM / WM) * BN) * 4 <= KG * 2 * (A_TILE + B_TILE) * 2,
                  "partial sums + reduction scratch fit in the stages");
    if (!lead) {
      float* dst = fsm + (grp - 1) * PART + wave * 64 * 16 * TM * TN + lane;
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
        const float* src = fsm + (g - 1) * PART + wave * 64 * 16 * TM * TN + lane;
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] += src[((m * TN + n) * 16 + r) * 64];
      }
    }
    scratch = fsm + (KG - 1) * PART;
  }

  // ---- epilogue: D row = (reg&3) + 8*(reg>>2) + 4*lh, column = li ----
  // BSTATS (uniform `bs`): needs every tile inside one image and all rows valid (dispatcher check)
  const bool direct = (p.sout == 1 && p.Hl == p.Hout && p.Wl == p.Wout);
  const bool bs = !FUSED && p.bs_partial;
  constexpr int WAVES_M_ = BM / WM;
  float2* bred = reinterpret_cast<float2*>(scratch);
  const int bimg = m0 / HlWl;
  TO* const outp = reinterpret_cast<TO*>(p.out);
  const TO* const ybase = reinterpret_cast<const TO*>(p.bs_y);
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
          for (int r = 0; r < 16; ++r)
            acc[m][n][r] += bv + (o[r] != kNoOut ? ld1(outp + o[r]) : 0.f);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] += bv;
        }
        float yv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          yv[r] = 0.f;
          if (o[r] != kNoOut) {
            st1(outp + o[r], acc[m][n][r]);
            yv[r] = ld1(ybase + o[r]);
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
        store_block16_off(outp, o, acc[m][n], bv, p.accumulate);
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
                   n0 + threadIdx.x] = out;
  }
  if (FUSED && p.stats) {   // uniform; statistics of the fp32 accumulators (before any rounding)
    constexpr int WAVES_M = BM / WM;
    float2* red = reinterpret_cast<float2*>(scratch);
    static_assert(WAVES_M * BN * 4 <= 2 * A_TILE, "stats scratch fits in the A tiles");
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = n0 + wn0 + n * 32 + li;
      const float bv = p.bias ? p.bias[col] : 0.f;
      const float2 mine = wave_col_stats<TM>([&](int m, int r) { return acc[m][n][r] + bv; });
      if (lh == 0 && lead) red[(wave / WAVES_N) * BN + wn0 + n * 32 + li] = mine;
    }
    float2 out;
    if (block_col_stats<BN, WAVES_M>(red, 0, 0, false, float2{0.f, 0.f}, 32.f * TM, out)) {
      const int img = m0 / HlWl;   // the whole tile lies in this image (dispatcher check)
      p.stats[((size_t)img * p.stats_tiles + (m0 - img * HlWl) / BM) * p.Ncols + n0 + threadIdx.x] = out;
    }
  }
}

// ---------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant: fp32-class accuracy on the bf16 matrix cores.
// Every fp32 operand x is split while it is staged into LDS into three bf16 terms
//   x = h + m + l,  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)      (|x - h - m - l| <= 2^-26 |x|)
// and a product a*b is evaluated as the six terms of weight >= 2^-16
//   a_h b_h + a_h b_m + a_m b_h + a_h b_l + a_m b_m + a_l b_h
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (each bf16 x bf16 product is exact in
// fp32); the three dropped terms are below 2^-24 |a b|, the size of one fp32 rounding.
// Six MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 32x32x16 block.
// Same gather-GEMM, tap table, tiles and epilogue as conv_igemm_kernel; BK = 16 so that the
// three planes of both double-buffered tiles fit twice per CU; the global loads run two K
// steps ahead of the MFMAs (registers), the split one step ahead (LDS).
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_split_kernel(const IgemmParams p) {
  constexpr int BK = 16;
  constexpr int LDA = BK + 8;          // bf16 elements per LDS row (48 B: conflict-free b128)
  constexpr int SEGS = BK / 4;         // 16-B fp32 segments per tile row
  constexpr int ROWS = 256 / SEGS;     // 64 tile rows per loader pass
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  static_assert(BM % ROWS == 0, "BM must be a multiple of 64");
  constexpr int A_PASSES = BM / ROWS;
  // weights arrive pre-split (three bf16 planes): slot = (plane, row, 8-element half row)
  constexpr int B_SLOTS = BN * 6;
  constexpr int B_PASSES = (B_SLOTS + 255) / 256;
  constexpr int A_TILE = BM * LDA, B_TILE = BN * LDA;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_h[];
  __bf16* As = smem_h;                    // [buf][plane][BM][LDA]
  __bf16* Bs = smem_h + 2 * 3 * A_TILE;   // [buf][plane][BN][LDA]

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
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(p.w3), 0, (int)p.w3_bytes, 0x00020000);
  // weight slots of this thread (slots past B_SLOTS alias an earlier slot: same bytes, same
  // destination, so the duplicate store is harmless and the loop needs no predicate)
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
  f32x4 ra[A_PASSES];                    // tile ks+1 (loaded one step ago)
  f32x4 na[A_PASSES];                    // tile ks+2 (in flight)
  i32x4 rb[B_PASSES], nb[B_PASSES];      // weight planes: raw bf16 bits
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int KS = p.ntaps * (Ktot / BK);

  auto load_tiles = [&](int t, int chunk, f32x4* qa, i32x4* qb) {
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
      const unsigned off = ((unsigned)((a_nb[i] + iy * p.Win + ix) * Cs + coff) * 4u) |
                           (ok ? 0u : 0x80000000u);
      qa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    }
    const unsigned woff = (unsigned)(wt * p.tap_stride + c) * 2u;
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j)
      qb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wslot_off[j] + woff, 0, 0);
  };
  auto store_tiles = [&](int buf) {
    __bf16* Ab = As + buf * 3 * A_TILE + lrow * LDA + lseg * 4;
    __bf16* Bb = Bs + buf * 3 * B_TILE;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      bf16x4 h, m, l;
      split3(ra[i], h, m, l);
      *reinterpret_cast<bf16x4*>(Ab + ROWS * i * LDA) = h;
      *reinterpret_cast<bf16x4*>(Ab + A_TILE + ROWS * i * LDA) = m;
      *reinterpret_cast<bf16x4*>(Ab + 2 * A_TILE + ROWS * i * LDA) = l;
    }
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) *reinterpret_cast<i32x4*>(Bb + wslot_lds[j]) = rb[j];
  };

  int t_next = 0, chunk_next = 0, issued = 0;
  auto advance = [&]() {  // branch-free; past the end it keeps re-staging the last tile
    const bool on = issued + 1 < KS;
    const int tn2 = t_next + 1;
    const bool wrap = tn2 == p.ntaps;
    t_next = on ? (wrap ? 0 : tn2) : t_next;
    chunk_next = on ? chunk_next + (wrap ? 1 : 0) : chunk_next;
    issued += on ? 1 : 0;
  };

  load_tiles(t_next, chunk_next, ra, rb);
  advance();
  store_tiles(0);
  load_tiles(t_next, chunk_next, ra, rb);
  advance();
  __syncthreads();

  const int frag_off = li * LDA + 8 * lh;
  for (int ks = 0; ks < KS; ++ks) {
    const int buf = ks & 1;
    load_tiles(t_next, chunk_next, na, nb);
    advance();
    const __bf16* Ab = As + buf * 3 * A_TILE + wm0 * LDA + frag_off;
    const __bf16* Bb = Bs + buf * 3 * B_TILE + wn0 * LDA + frag_off;
    bf16x8 a[3][TM], b[3][TN];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int m = 0; m < TM; ++m)
        a[pl][m] = *reinterpret_cast<const bf16x8*>(Ab + pl * A_TILE + m * 32 * LDA);
#pragma unroll
      for (int n = 0; n < TN; ++n)
        b[pl][n] = *reinterpret_cast<const bf16x8*>(Bb + pl * B_TILE + n * 32 * LDA);
    }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        f32x16 c = acc[m][n];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][m], b[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[2][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][m], b[0][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[1][n], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][m], b[0][n], c, 0, 0, 0);
        acc[m][n] = c;
      }
    store_tiles(buf ^ 1);
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) ra[i] = na[i];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) rb[j] = nb[j];
    __syncthreads();
  }

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

template <int BM, int BN, int WM, int WN>
int launch_igemm_bf16(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)(BM + BN) * 40 * sizeof(__bf16);
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const long long tiles = ceil_div64(M, BM) * (p.Ncols / BN);
  hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WM, WN>), dim3((unsigned)tiles), dim3(256), lds,
                     stream, p);
  UNET_CHECK_LAUNCH("conv_igemm_bf16");
  return UNET_OK;
}

// bf16 STORAGE (the real mixed-precision pipeline): sources and output are bf16 tensors.
// stats_px: the fused-layer forward (statistics epilogue where every tile lies in one image);
// bs_px: a data gradient whose output is final for a layer (BSTATS epilogue, same condition).
template <int BM, int BN, int WM, int WN, int KG = 1, int MODE = 0>
int launch_igemm_b16(IgemmParams p, hipStream_t stream, int* stats_px, int* bs_px = nullptr) {
  constexpr size_t lds = KG * 2 * (size_t)(BM + BN) * (MODE ? 72 : 40) * sizeof(__bf16);
  static_assert(lds <= 160 * 1024, "LDS stages of every K group fit one CU");
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const long long tiles = ceil_div64(M, BM) * (p.Ncols / BN);
  const int HlWl = p.Hl * p.Wl;
  if constexpr (MODE != 0) {
    if (stats_px) {   // WIDE fused layer forward
      const bool direct = p.sout == 1 && p.Hl == p.Hout && p.Wl == p.Wout;
      if (direct && p.stats && HlWl % BM == 0) { *stats_px = BM; p.stats_tiles = HlWl / BM; }
      else { *stats_px = 0; p.stats = nullptr; }
      p.bs_partial = nullptr;
      if constexpr (MODE == 1) {
        auto kern = conv_igemm_bf16_kernel<BM, BN, WM, WN, __bf16, __bf16, true, KG, 1>;
        UNET_SET_DYN_LDS(kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256 * KG), lds, stream, p);
      }
      UNET_CHECK_LAUNCH("conv_igemm_b16");
      return UNET_OK;
    }
    if (bs_px && p.bs_partial && HlWl % BM == 0 && M % BM == 0) {
      *bs_px = BM; p.bs_tiles = HlWl / BM * (p.sout * p.sout);
    } else {
      if (bs_px) *bs_px = 0;
      p.bs_partial = nullptr;
    }
    auto kern = conv_igemm_bf16_kernel<BM, BN, WM, WN, __bf16, __bf16, false, KG, MODE>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256 * KG), lds, stream, p);
  } else
  if (stats_px) {   // fused layer forward
    const bool direct = p.sout == 1 && p.Hl == p.Hout && p.Wl == p.Wout;
    if (direct && p.stats && HlWl % BM == 0) { *stats_px = BM; p.stats_tiles = HlWl / BM; }
    else { *stats_px = 0; p.stats = nullptr; }
    p.bs_partial = nullptr;
    auto kern = conv_igemm_bf16_kernel<BM, BN, WM, WN, __bf16, __bf16, true, KG>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256 * KG), lds, stream, p);
  } else {
    if (bs_px && p.bs_partial && HlWl % BM == 0 && M % BM == 0) {
      *bs_px = BM; p.bs_tiles = HlWl / BM * (p.sout * p.sout);
    } else {
      if (bs_px) *bs_px = 0;
      p.bs_partial = nullptr;
    }
    auto kern = conv_igemm_bf16_kernel<BM, BN, WM, WN, __bf16, __bf16, false, KG>;
    UNET_SET_DYN_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256 * KG), lds, stream, p);
  }
  UNET_CHECK_LAUNCH("conv_igemm_b16");
  return UNET_OK;
}

// Deep layers (at most one 64x64 tile per CU): number of K groups per block (1 = no split)
static int deep_k_groups_b16(const IgemmParams& p) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  if (p.Ncols % 64 != 0 || ceil_div64(M, 64) * (p.Ncols / 64) > 256) return 1;
  const int ks = p.ntaps * ((p.C0 + p.C1) / 32);
  int kg = 4;
  while (kg > 1 && (ks % kg != 0 || ks / kg < 4)) kg >>= 1;
  return kg;
}

template <int BM, int BN, int WM, int WN>
int launch_igemm_split(const IgemmParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * 3 * (size_t)(BM + BN) * 24 * sizeof(__bf16);
  auto kern = conv_igemm_split_kernel<BM, BN, WM, WN>;
  UNET_SET_DYN_LDS(kern, lds);
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const long long tiles = ceil_div64(M, BM) * (p.Ncols / BN);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, p);
  UNET_CHECK_LAUNCH("conv_igemm_split");
  return UNET_OK;
}

}  // namespace

int dispatch_igemm_bf16(const IgemmParams& p, hipStream_t stream) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  if (nc % 128 == 0 && ceil_div64(M, 128) * (nc / 128) >= 256)
    return launch_igemm_bf16<128, 128, 64, 64>(p, stream);
  if (nc % 64 == 0 && ceil_div64(M, 128) * (nc / 64) >= 256)
    return launch_igemm_bf16<128, 64, 64, 32>(p, stream);
  if (nc % 64 == 0 && M <= 128 * 256) return launch_igemm_bf16<64, 64, 32, 32>(p, stream);
  return launch_igemm_bf16<128, 32, 32, 32>(p, stream);
}

// bf16-storage gather-GEMM (sources / output bf16 in HBM); stats_px != nullptr: the fused-layer
// forward (activation on load, statistics epilogue where every tile lies in one image)
int dispatch_igemm_b16(const IgemmParams& p, hipStream_t stream, int* stats_px, int* bs_px) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  {   // stride-1 3x3 that tiles as 4 x 32 pixels: the patch-staged kernel (conv_patch.hip);
      // only it has the BSTATS epilogue (bs_px), the gather-GEMM forms report 0 tiles
    const int rc = launch_patch_b16_auto(p, stream, stats_px, bs_px);
    if (rc != 1) return rc;
    if (bs_px) *bs_px = 0;
  }
  if (stats_px && !bs_px) {   // the stride-2 fused forward: its own patch form (conv_patch.hip)
    const int rc = launch_patch_s2_b16_auto(p, stream, stats_px);
    if (rc != 1) return rc;
  }
  // "channel taps" on contiguous rows with the bf16 weight plane: the plain-GEMM (DENSE) form.
  // 64 x 64 tiles throughout - four workgroups per CU (37 KB of LDS, 94 registers): the K loop
  // is 5..72 steps short and what hides its prologue, epilogue and load latency is the other
  // workgroups of the CU (measured against 128 x 128 / 128 x 64 tiles at two per CU: 72 -> 60 us
  // at K = 576, 202 -> 140 at K = 288; profiles/r04_bf16_experiments.txt); four K groups per
  // tile where the tiles do not fill the chip.
  static const bool dense_off = [] { const char* e = getenv("UNET_B16_DENSE_TAPS"); return e && e[0] == '0'; }();
  if (!dense_off && !stats_px && p.w3 && p.tap_cstride == p.C0 && p.src0_pitch == 9 * p.C0 &&
      p.C1 == 0 && p.ntaps == 9 && (p.C0 & (p.C0 - 1)) == 0 && p.C0 >= 32 && p.sin == 1 &&
      p.sout == 1 && p.Hl == p.Hout && p.Wl == p.Wout && M % 64 == 0 && nc % 64 == 0) {
    if ((M / 64) * (nc / 64) <= 256 && (9 * p.C0) % 256 == 0)
      return launch_igemm_b16<64, 64, 32, 32, 4, 2>(p, stream, nullptr, bs_px);
    return launch_igemm_b16<64, 64, 32, 32, 1, 2>(p, stream, nullptr, bs_px);
  }
  if (nc % 128 == 0 && ceil_div64(M, 128) * (nc / 128) >= 256)
    return launch_igemm_b16<128, 128, 64, 64>(p, stream, stats_px, bs_px);
  if (nc % 64 == 0 && ceil_div64(M, 128) * (nc / 64) >= 256)
    return launch_igemm_b16<128, 64, 64, 32>(p, stream, stats_px, bs_px);
  // data gradients of the 1/32-resolution stage (at most one 64 x 64 tile per CU) with the bf16
  // weight plane: the gather form on 64-wide K steps (WIDE), four K groups
  static const bool wide_off = [] { const char* e = getenv("UNET_B16_WIDE_GATHER"); return e && e[0] == '0'; }();
  if (!wide_off && p.w3 && p.tap_cstride == 0 && nc % 64 == 0 && p.C0 % 64 == 0 &&
      p.C1 % 64 == 0 && ceil_div64(M, 64) * (nc / 64) <= 256 &&
      (p.ntaps * ((p.C0 + p.C1) / 64)) % 4 == 0 && p.ntaps * ((p.C0 + p.C1) / 64) >= 8)
    return launch_igemm_b16<64, 64, 32, 32, 4, 1>(p, stream, stats_px, bs_px);
  if (nc % 64 == 0 && M <= 128 * 256) {
    const int kg = deep_k_groups_b16(p);
    return kg == 4   ? launch_igemm_b16<64, 64, 32, 32, 4>(p, stream, stats_px, bs_px)
           : kg == 2 ? launch_igemm_b16<64, 64, 32, 32, 2>(p, stream, stats_px, bs_px)
                     : launch_igemm_b16<64, 64, 32, 32>(p, stream, stats_px, bs_px);
  }
  return launch_igemm_b16<128, 32, 32, 32>(p, stream, stats_px, bs_px);
}

int dispatch_igemm_split(const IgemmParams& p, hipStream_t stream) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  // images narrower than one 32-pixel patch row (the 1/32-resolution stage): too few tiles for
  // the split kernels to win, the fp32 matrix-core kernel is faster there
  if (p.Wl < 32 && p.sout == 1) return dispatch_igemm(p, stream);
  if (patch_split_applicable(p)) return launch_patch_split_auto(p, stream);
  if (nc % 128 == 0 && ceil_div64(M, 128) * (nc / 128) >= 256)
    return launch_igemm_split<128, 128, 64, 64>(p, stream);
  if (nc % 64 == 0 && ceil_div64(M, 128) * (nc / 64) >= 256)
    return launch_igemm_split<128, 64, 64, 32>(p, stream);
  if (nc % 64 == 0 && M <= 128 * 256) return launch_igemm_split<64, 64, 32, 32>(p, stream);
  return launch_igemm_split<128, 32, 32, 32>(p, stream);
}

}  // namespace unet_conv

// conv_wino.hip — stride-1 3x3 convolution (fused-layer forward and data gradient) as Winograd
// F(2x2, 3x3) on the fp32 matrix cores: 16 multiplies per 2x2 output tile and channel pair
// instead of 36, i.e. 2.25x fewer MFMA FLOPs than the direct kernels of conv_patch.hip, whose
// ceiling is the fp32 matrix-core peak itself.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          (Lavin & Gray 2015)
//
// per 4x4 input tile d (stride 2) and 3x3 filter g.  The 16 element positions xi are 16
// independent GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co]:
//   * U (= G g G^T, 16/9 of the weight bytes) is produced once per step by wino_pack_kernel in
//     exactly the byte order the kernel wants in LDS, so it goes global -> LDS by DMA
//     (global_load_lds_dwordx4: no registers, no ds_write);
//   * the input transform B^T d B runs on chip, fused behind the loader: a (8+2) x (32+2) pixel
//     patch of an 8-channel chunk is activated on load (InstanceNorm + LeakyReLU + dropout of the
//     producing layer: the fused pipeline), staged raw, and every thread transforms one
//     (tile, channel) - 8 LDS reads, 32 adds, 16 LDS writes - behind the MFMAs of the previous
//     chunk;
//   * one workgroup (8 waves, one per CU) owns 8 x 32 output pixels (4 x 16 tiles) x 64 output
//     channels: wave w holds ALL 16 xi of 16 tiles x 32 channels (v_mfma_f32_16x16x4_f32:
//     16 xi x 2 blocks x 4 registers = 128 accumulator VGPRs), so the output transform A^T M A
//     is register-local: no exchange between waves;
//   * epilogues as in the direct kernels: bias, per-tile InstanceNorm statistics (STATS, forward)
//     or the next layer's InstanceNorm-backward reductions (BSTATS, data gradient).
//
// LDS: V 2 x 32 KB + U 2 x 32 KB (double-buffered per 8-channel chunk, XOR-swizzled so the
// 8-byte fragment reads are conflict-free without padding) + 2 x 11.4 KB raw patch = 150.3 KB.
// Error: the transforms add a few fp32 roundings per output (coefficients 1 and 1/2 only);
// measured <= 3e-6 of max |y| against the direct kernel, well inside the 1e-4 logits bound.
//
// Replaces nn.Conv2d forward / aten::convolution_backward(data) of the C -> C stride-1 layers
// (Our_UNet/models/unet.py:106-115), reached through unet_conv_in_fwd_wino /
// unet_conv3x3_bwd_data_bs_wino.
#include "conv_params.h"
#include "lds_asm.h"
#include <utility>

namespace unet_conv {
namespace {

typedef int i32x4v __attribute__((ext_vector_type(4)));
template <int I> using wn_ic = std::integral_constant<int, I>;
template <int B, int... I, typename F>
__device__ __forceinline__ void wn_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(wn_ic<B + I>{}), ...);
}
template <int B, int E, typename F>   // compile-time loop: f(integral_constant<int, I>), I in [B, E)
__device__ __forceinline__ void wn_for(F&& f) {
  wn_for_impl<B>(std::make_integer_sequence<int, E - B>{}, f);
}

constexpr int WN_TH = 8, WN_TW = 32;        // output pixels of a workgroup
constexpr int WN_PW = WN_TW + 2, WN_PH = WN_TH + 2;
constexpr int WN_PPIX = WN_PH * WN_PW;      // 340 patch pixels
constexpr int WN_BN = 64;                   // output channels of a workgroup
constexpr int WN_KC = 8;                    // channels per chunk
constexpr int WN_RP = 360;                  // channel-plane pitch of the raw patch: 40 mod 64, the
                                            // transform's ds_read_b64 (32 lanes over 64 banks) are conflict-free
constexpr int WN_BUF = 16 * 64 * WN_KC;     // floats of one V / U stage (32 KB)
constexpr size_t WN_LDS = (size_t)(4 * WN_BUF + 2 * WN_KC * WN_RP) * sizeof(float);

// position of channel k (0..7) of row `row` (tile or output channel) inside its 8-float group:
// channel pairs XOR-swizzled by bits 2-3 of the row.  A fragment read is 16 rows x one pair per
// 16 lanes; hipcc fuses neighbouring ds_read_b64 into ds_read2(st64)_b64, which the LDS serves
// in 16-lane groups over 32 banks (rows r, r+4, r+8, r+12 share a bank residue: their pairs must
// differ), while a plain ds_read_b64 is served in 32-lane groups over 64 banks (rows r, r+8
// collide: the pair sets of the two lane halves must differ).  This swizzle is conflict-free
// under both (the first layout, XOR by bit 3 only, measured 0.43 conflict cycles per active
// LDS cycle once the reads were fused).
__host__ __device__ __forceinline__ int wn_swz(int row, int k) {
  return 2 * ((k >> 1) ^ ((row >> 2) & 3)) + (k & 1);
}

struct WinoParams {
  IgemmParams g;       // sources / activation coefficients / output / epilogue descriptors
  const float* wu;     // packed U: [n tile of 64][K chunk of 8][xi 16][n 64][k 8 swizzled]
  // DZ (data gradient only): source 0 is the gradient g = dL/da of the layer's ACTIVATED output;
  // the loader forms dz = dL/dy (InstanceNorm + LeakyReLU + dropout backward) from (g, y) and
  // the coefficient planes of in_bwd_coef_kernel while it stages the patch - the elementwise
  // in_bwd_apply pass of that layer is gone - and the workgroups of column tile 0 also WRITE dz
  // (interior pixels of their tile) for the layer's weight gradient; workgroup 0 emits the
  // layer's dgamma / dbeta / dbias from the per-image sums, as in_bwd_apply did.
  const float* dz_y;       // raw conv output y, shape of source 0
  const float* dz_coef;    // [5][N][C0]: a1, b1, P, Q, R
  float* dz_out;           // dz, shape of source 0 (must not alias g: halos are re-read)
  const float2* dz_sums;   // [N][C0] (S1, S2)
  const float* dz_gamma;   // [C0]
  const float* dz_rstd;    // [N][C0]
  float* dz_dgamma; float* dz_dbeta; float* dz_dbias;   // [C0] each (may be null)
  float dz_slope;
};

// UP: source 0 is the LOW-resolution tensor [N][H/2][W/2][C0] of a decoder stage's first
// convolution, conv3x3(cat(upsample2x(act(low)), act(skip))): its patch pixels are gathered
// bilinearly (align_corners=False at exactly 2x: taps {0.75, 0.25}, edge clamped; activation on
// the four taps, PyTorch's blend order, zero padding after the blend - as conv_patch_up_kernel)
// on their way into the raw patch, so the up-sampled tensor never exists.
template <bool ACT, bool STATS, bool BSTATS, bool UP = false, bool DZ = false>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const WinoParams wp) {
  static_assert(!DZ || (!ACT && !UP && !STATS), "DZ is a data-gradient loader");
  const IgemmParams& p = wp.g;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                    // [buf][xi][tile 64][8]
  float* Us = smem + 2 * WN_BUF;       // [buf][xi][n 64][8]
  float* Rs = smem + 4 * WN_BUF;       // [buf][channel 8][WN_RP]: raw (activated) patch, pixel-major

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.Hin, W = p.Win;
  const int tiles_n = p.Ncols / WN_BN, tiles_x = W / WN_TW, tiles_y = H / WN_TH;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % tiles_n; bid /= tiles_n;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int y0 = ty * WN_TH, x0 = tx * WN_TW, n0 = tn * WN_BN;
  const int Ktot = p.C0 + p.C1;
  const int chunks = Ktot / WN_KC;

  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src1 ? p.src1 : p.src0), 0, (int)p.src1_bytes, 0x00020000);
  // DZ: y and dz have the shape (and the slot offsets) of source 0
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(DZ ? wp.dz_y : p.src0), 0, (int)p.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(
      DZ ? wp.dz_out : const_cast<float*>(p.src0), 0, (int)p.src0_bytes, 0x00020000);
  if (DZ && blockIdx.x == 0) {   // parameter gradients of the layer (N x C0 sums to read)
    const float hw = (float)(H * W), inv = 1.f / hw;
    for (int c = tid; c < p.C0; c += 512) {
      float dg = 0.f, db = 0.f, dbi = 0.f;
      for (int q = 0; q < p.N; ++q) {
        const float2 v = wp.dz_sums[(size_t)q * p.C0 + c];
        db += v.x;
        dg += v.y;
        dbi += wp.dz_gamma[c] * wp.dz_rstd[(size_t)q * p.C0 + c] * (v.x - hw * (v.x * inv));
      }
      if (wp.dz_dgamma) wp.dz_dgamma[c] = dg;
      if (wp.dz_dbeta) wp.dz_dbeta[c] = db;
      if (wp.dz_dbias) wp.dz_dbias[c] = dbi;
    }
  }

  // ---- raw patch slots: 340 pixels x 2 channel halves; thread -> slots tid, tid + 512 ----
  // Everything the K loop needs per slot is loop-invariant and lives in a register or an
  // immediate: on this chip every VALU instruction of either wave of a SIMD takes ~2.6 cycles
  // away from the fp32 matrix pipe, SALU and LDS instructions ~1-3 (tools/micro/
  // mfma_piece_cost.hip), so the loop is written for instruction count: buffer loads take the
  // chunk's channel offset in their SCALAR offset, the zero padding is an EXEC mask on the LDS
  // stores (the padding slots are zeroed once), LDS addresses are immediates (the loop is
  // unrolled over the two stage parities), DMA / coefficient loads are scalar base + lane offset.
  const int half = tid & 1;
  unsigned voff0[UP ? 1 : 2], voff1[2];   // byte offsets into source 0 / 1, bit 31 = outside the image
  bool okslot[2], wrslot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = tid + 512 * i;
    const int pix = slot >> 1;
    const int prow = pix / WN_PW, pcol = pix - prow * WN_PW;
    const int iy = y0 - 1 + prow, ix = x0 - 1 + pcol;
    const bool ok = slot < 2 * WN_PPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const unsigned lin = (unsigned)((n * H + iy) * W + ix);
    okslot[i] = ok;
    // (DZ: the pixels a workgroup of column tile 0 writes dz for: the interior of its patch)
    wrslot[i] = DZ && ok && tn == 0 && prow >= 1 && prow <= WN_TH && pcol >= 1 && pcol <= WN_TW;
    if (!UP) voff0[i] = ok ? (lin * (unsigned)p.C0 + half * 4) * 4u : 0x80000000u;
    voff1[i] = ok ? (lin * (unsigned)p.C1 + half * 4) * 4u : 0x80000000u;
  }
  // UP: per slot the clamped low-resolution pixel of tap (0,0) and four flag bits: step to the
  // right / lower neighbour (0 where clamped) and the parity of the patch column / row
  int l_lin[2], l_meta[2];
  const int lw = W >> 1;
  if (UP) {
    const int h = H >> 1, w = W >> 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int slot = tid + 512 * i;
      const int pix = slot < 2 * WN_PPIX ? slot >> 1 : 0;
      const int prow = pix / WN_PW, pcol = pix - prow * WN_PW;
      int gy0 = (y0 >> 1) - 1 + (prow >> 1), gx0 = (x0 >> 1) - 1 + (pcol >> 1);
      int gy1 = gy0 + 1, gx1 = gx0 + 1;
      gy0 = gy0 < 0 ? 0 : (gy0 > h - 1 ? h - 1 : gy0);
      gy1 = gy1 < 0 ? 0 : (gy1 > h - 1 ? h - 1 : gy1);
      gx0 = gx0 < 0 ? 0 : (gx0 > w - 1 ? w - 1 : gx0);
      gx1 = gx1 < 0 ? 0 : (gx1 > w - 1 ? w - 1 : gx1);
      l_lin[i] = (n * h + gy0) * w + gx0;
      l_meta[i] = (gx1 - gx0) | ((gy1 - gy0) << 1) | ((pcol & 1) << 2) | ((prow & 1) << 3);
    }
  }
  // Register sets of the loader.  Plain variants: TWO sets, used alternately by chunk parity -
  // the loads of chunk c + 4 are issued in iteration c, behind its DMA pieces, and may stay in
  // flight across the next barrier (the loop top waits only for everything OLDER than them: the
  // DMA): two whole iterations before they are stored.  With one set (up-sampling and DZ
  // variants: no registers left) the loads of chunk c + 3 precede the DMA and the barrier waits
  // for them.
  constexpr int NSET = (UP || DZ) ? 1 : 2;
  f32x4 pr[NSET][2];
  f32x4 pt[UP ? 2 : 1][3];   // UP: the other three taps of a slot
  bool pup = false;           // the chunk in the registers is a low-resolution (gathered) one
  f32x4 ca[NSET], cb[NSET];
#pragma unroll
  for (int q = 0; q < NSET; ++q) { ca[q] = f32x4{1.f, 1.f, 1.f, 1.f}; cb[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 py[DZ ? 2 : 1];            // DZ: the y values of the two slots
  f32x4 cz[DZ ? 5 : 1];            // DZ: a1, b1, P, Q, R of this lane's four channels
  const unsigned hoff = (unsigned)half * 16u;   // this lane's 16 bytes of a chunk's coefficients
  // G: global -> registers, one slot per call.  Slot 1's call also fetches the chunk's coefficients
  // and source kind: it runs AFTER both slots of the previous chunk were stored
  auto load_raw = [&](int chunk, auto ic, auto setc) {
    constexpr int i = decltype(ic)::value;
    constexpr int Q = decltype(setc)::value;
    const int c = chunk * WN_KC;
    const bool first = c < p.C0;
    const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
    const int Cs = first ? p.C0 : p.C1;
    const int cc = first ? c : c - p.C0;      // (uniform: scalar registers)
    if (UP && first) {   // uniform
      if (i == 1) pup = true;
      const unsigned o = (unsigned)(l_lin[i] * Cs + cc + half * 4);
      const unsigned dx = (l_meta[i] & 1) ? (unsigned)Cs : 0u;
      const unsigned dy = (l_meta[i] & 2) ? (unsigned)(lw * Cs) : 0u;
      pr[Q][i] = buf_ld4<float>(rs, o, 0u);
      if constexpr (UP) {
        pt[i][0] = buf_ld4<float>(rs, o + dx, 0u);
        pt[i][1] = buf_ld4<float>(rs, o + dy, 0u);
        pt[i][2] = buf_ld4<float>(rs, o + dy + dx, 0u);
      }
    } else {
      if (UP && i == 1) pup = false;
      unsigned vo = voff1[i];
      if constexpr (!UP) vo = first ? voff0[i] : voff1[i];
      pr[Q][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, cc * 4, 0));
      if constexpr (DZ)
        py[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy, vo, cc * 4, 0));
    }
    if constexpr (DZ) {
      if (i == 1) {   // the chunk's coefficient planes (looked at when the patch is stored)
        const float* cf = wp.dz_coef + ((size_t)n * p.C0 + cc);
        const size_t plane = (size_t)p.N * p.C0;
#pragma unroll
        for (int k = 0; k < 5; ++k)
          cz[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(cf + k * plane) + hoff);
      }
    }
    if (ACT && i == 1) {   // (ACT: every source is an activated tensor - the entry points check)
      const float* al = (first ? p.act0_alpha : p.act1_alpha) + ((size_t)n * Cs + cc);
      const float* be = (first ? p.act0_beta : p.act1_beta) + ((size_t)n * Cs + cc);
      // the loaded coefficients are only looked at when the patch is stored, an iteration later
      // (compiler-tracked loads on purpose, although hipcc builds a 64-bit VALU address for
      // them: the values live across the loop's back edge, where the register allocator copies
      // them between the chunk bodies - an untracked asm load still in flight at such a copy
      // is copied stale and lands in a register that has been given to something else)
      ca[Q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(al) + hoff);
      cb[Q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(be) + hoff);
    }
  };
  // raw patch position of slot i (slots past the patch: never stored)
  const int rpix1 = tid < 2 * WN_PPIX - 512 ? 256 + (tid >> 1) : WN_PPIX + (tid & 15);
  float* const rdst0 = Rs + (half * 4) * WN_RP + (tid >> 1);
  float* const rdst1 = Rs + (half * 4) * WN_RP + rpix1;
  constexpr int RB = WN_KC * WN_RP;
  // the zero padding: a slot outside the image is never written in the loop - zero it once in
  // both raw buffers (the same thread owns the position for every chunk)
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (!okslot[i] && tid + 512 * i < 2 * WN_PPIX) {
      float* d = i == 0 ? rdst0 : rdst1;
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k * WN_RP] = d[RB + k * WN_RP] = 0.f;
    }
  // (DZ: `cst` = the chunk in the registers, or -1 when it is a re-staged tail chunk - its dz
  // was written already)
  auto store_raw = [&](auto ic, auto bc, auto setc, int cst = -1) {   // R: activate, registers -> LDS raw patch (buffer bc)
    constexpr int i = decltype(ic)::value;
    constexpr int B = decltype(bc)::value;
    constexpr int Q = decltype(setc)::value;
    f32x4 v = pr[Q][i];
    if constexpr (DZ) {
      // dz = (z > 0 ? P : P slope) g + (Q y + R),  z = y a1 + b1
      const f32x4 yv = py[i];
      const f32x4 z = yv * cz[0] + cz[1];
      const f32x4 ps = cz[2] * wp.dz_slope;
      f32x4 sel;
#pragma unroll
      for (int k = 0; k < 4; ++k) sel[k] = z[k] > 0.f ? cz[2][k] : ps[k];
      v = sel * v + (cz[3] * yv + cz[4]);
      if (wrslot[i] && cst >= 0)    // (uniform in cst; EXEC-masked in wrslot)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v, v), rso, voff0[i],
                                               cst * WN_KC * 4, 0);
    }
    if (ACT) {
      const f32x4 a1 = ca[Q], b1 = cb[Q];
      const float sl = p.slope;
      bool done = false;
      if constexpr (UP) {
        if (pup) {   // uniform: activate the four taps, then blend
          // odd patch row = even image row 2k: taps (k-1, k) weigh (0.25, 0.75); even patch
          // row = odd image row: (0.75, 0.25); columns alike (y0, x0 are even)
          const f32x4 p00 = act4f(v, a1, b1, sl, 1.f), p01 = act4f(pt[i][0], a1, b1, sl, 1.f);
          const f32x4 p10 = act4f(pt[i][1], a1, b1, sl, 1.f);
          const f32x4 p11 = act4f(pt[i][2], a1, b1, sl, 1.f);
          const float wx1 = (l_meta[i] & 4) ? 0.75f : 0.25f, wx0 = 1.f - wx1;
          const float wy1 = (l_meta[i] & 8) ? 0.75f : 0.25f, wy0 = 1.f - wy1;
          v = (p00 * wx0 + p01 * wx1) * wy0 + (p10 * wx0 + p11 * wx1) * wy1;
          done = true;
        }
      }
      if (!done) v = act4f(v, a1, b1, sl, 1.f);
    }
    if (okslot[i]) {   // (the zero padding: EXEC-masked stores, see above)
      float* d = (i == 0 ? rdst0 : rdst1) + B * RB;
      d[0] = v[0]; d[WN_RP] = v[1]; d[2 * WN_RP] = v[2]; d[3 * WN_RP] = v[3];
    }
  };
  // ---- input transform: thread -> (tile = tid >> 3, channel = tid & 7) ----
  const int t_tile = tid >> 3, t_ch = tid & 7;
  const unsigned t_ra = lds_addr(Rs + t_ch * WN_RP + (2 * (t_tile >> 4)) * WN_PW + 2 * (t_tile & 15));
  float* const t_vdst = Vs + t_tile * 8 + wn_swz(t_tile, t_ch);
  // T: V = (B^T d) B of this thread's 4 x 4 window, one OUTPUT row per piece, in two halves a
  // stage apart: t_rd issues the 8-byte reads of the two input rows the row combines, t_wr (a
  // stage later, behind that stage's fragment wait, which also covers these reads) combines
  // them, runs the column pass and writes - no LDS latency is waited for on the spot (with the
  // wait inside the piece both waves of a SIMD sat out an LDS round trip per row)
  f32x2v tr[4];
  auto t_rd = [&](auto ac, auto bc) {       // bc: raw buffer
    constexpr int a = decltype(ac)::value;
    constexpr int B = decltype(bc)::value;
    constexpr int r0 = a == 0 ? 0 : (a == 1 ? 1 : (a == 2 ? 2 : 1));   // first row
    constexpr int r1 = a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));   // second row
    tr[0] = lds_rd64<(B * RB + r0 * WN_PW) * 4>(t_ra);
    tr[1] = lds_rd64<(B * RB + r0 * WN_PW) * 4 + 8>(t_ra);
    tr[2] = lds_rd64<(B * RB + r1 * WN_PW) * 4>(t_ra);
    tr[3] = lds_rd64<(B * RB + r1 * WN_PW) * 4 + 8>(t_ra);
  };
  auto t_wr = [&](auto ac, auto bc) {       // bc: V stage
    constexpr int a = decltype(ac)::value;
    constexpr int B = decltype(bc)::value;
    lds_wait<15>(tr[0], tr[1], tr[2], tr[3]);   // (no-op wait: keeps the uses behind the stage's wait)
    // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    const f32x2v lo = a == 1 ? tr[0] + tr[2] : tr[0] - tr[2];
    const f32x2v hi = a == 1 ? tr[1] + tr[3] : tr[1] - tr[3];
    float* Vb = t_vdst + B * WN_BUF;
    Vb[(4 * a + 0) * 512] = lo[0] - hi[0];
    Vb[(4 * a + 1) * 512] = lo[1] + hi[0];
    Vb[(4 * a + 2) * 512] = hi[0] - lo[1];
    Vb[(4 * a + 3) * 512] = lo[1] - hi[1];
  };
  // ---- U chunk: 32 KB contiguous in global, by DMA (8 waves x 4 x 1 KB) ----
  // (uniform base + 32-bit lane offset: no 64-bit address arithmetic per lane)
  const float* ubase = wp.wu + (size_t)((p.n_off / WN_BN + tn) * chunks) * WN_BUF;
  const int uwave = __builtin_amdgcn_readfirstlane(wave) * 1024;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned u_m0 = __builtin_amdgcn_readfirstlane(lds_addr(Us + uwave));
  auto dma_u = [&](int chunk, auto bc, auto ic) {   // piece i of this wave's four, into U stage bc
    constexpr int i = decltype(ic)::value;
    constexpr int B = decltype(bc)::value;
    const float* src = ubase + (size_t)chunk * WN_BUF + uwave;          // (scalar)
    const unsigned m0v = u_m0 + B * WN_BUF * 4;
    // global_load_lds_dwordx4, scalar base + lane offset form; M0 = LDS base of the wave's part
    // of the stage; the instruction offset (piece i) applies to the global AND the LDS address
    dma16_sbase<i * 1024>(m0v, lane16, src);
  };

  // ---- MFMA fragments: wave -> tiles 16 tg .. +15, output channels 32 nh .. +31 ----
  const int tg = wave & 3, nh = wave >> 2;
  const int fm = lane & 15, fk = lane >> 4;
  const int fsw = 2 * (fk ^ ((fm >> 2) & 3));
  const unsigned va = lds_addr(Vs + (16 * tg + fm) * 8 + fsw);   // this lane's byte addresses
  const unsigned ub = lds_addr(Us + (32 * nh + fm) * 8 + fsw);
  f32x4 acc[16][2];
#pragma unroll
  for (int x = 0; x < 16; ++x)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragments: read one xi ahead of their MFMAs into the other register set (the up-sampling
  // variant has no registers to spare: one set, read right behind the MFMAs that used it)
  constexpr int FS = UP ? 1 : 2;
  f32x2v fa[FS], fb0[FS], fb1[FS];
  auto frag = [&](auto bc, auto xc) {
    constexpr int x = decltype(xc)::value;
    constexpr int B = decltype(bc)::value;
    constexpr int sl = x % FS;
    fa[sl] = lds_rd64<(B * WN_BUF + x * 512) * 4>(va);
    fb0[sl] = lds_rd64<(B * WN_BUF + x * 512) * 4>(ub);
    fb1[sl] = lds_rd64<(B * WN_BUF + x * 512 + 128) * 4>(ub);
  };
  auto mm = [&](auto xc) {
    constexpr int x = decltype(xc)::value;
    constexpr int sl = x % FS;
    acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][0], fb0[sl][0], acc[x][0], 0, 0, 0);
    acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][0], fb1[sl][0], acc[x][1], 0, 0, 0);
    acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][1], fb0[sl][1], acc[x][0], 0, 0, 0);
    acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][1], fb1[sl][1], acc[x][1], 0, 0, 0);
  };
  // One stage = one xi.  Two fragment sets: the reads of xi + 1 (and, in the transform stages,
  // the four raw-patch reads of a transform row: TR) are issued FIRST, then the wait for the
  // fragments of xi (issued a stage ago: everything but the newest 3 + 4 reads), then the four
  // MFMAs - which cover the LDS round trip of what was just issued.
  auto stage = [&](auto bc, auto xc, auto trc) {
    constexpr int x = decltype(xc)::value;
    constexpr int TR = decltype(trc)::value;     // 4 = a transform row was read ahead of the fragments
    constexpr int sl = x % FS;
    if constexpr (FS == 2) {
      if constexpr (x + 1 < 16) frag(bc, wn_ic<x + 1>{});
      lds_wait<(x + 1 < 16 ? 3 : 0) + TR>(fa[sl], fb0[sl], fb1[sl]);
      __builtin_amdgcn_sched_barrier(0);
      mm(xc);
    } else {
      lds_wait<0>(fa[sl], fb0[sl], fb1[sl]);
      __builtin_amdgcn_sched_barrier(0);
      mm(xc);
      if constexpr (x + 1 < 16) frag(bc, wn_ic<x + 1>{});
    }
  };

  // ---- prologue: chunk 0 transformed, chunk 1 raw in LDS, chunk 2 (and, with two register
  //      sets, chunk 3) in the registers ----
  using S0 = wn_ic<0>;
  using S1 = wn_ic<NSET - 1>;
  auto cl = [&](int c) { return c < chunks ? c : chunks - 1; };   // the tail re-stages the last chunk
  // (the coefficient loads and the DMA are not tracked by the compiler: explicit waits)
  auto vm_done = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < NSET; ++q) reg_anchor(ca[q], cb[q]);
    if constexpr (DZ) { reg_anchor(cz[0], cz[1]); reg_anchor(cz[2], cz[3]); reg_anchor(cz[4]); }
  };
  if constexpr (NSET == 2) {
    load_raw(0, wn_ic<0>{}, S0{});
    load_raw(0, wn_ic<1>{}, S0{});
    load_raw(cl(1), wn_ic<0>{}, S1{});
    load_raw(cl(1), wn_ic<1>{}, S1{});
    wn_for<0, 4>([&](auto ic) { dma_u(0, wn_ic<0>{}, ic); });
    vm_done();
    store_raw(wn_ic<0>{}, wn_ic<0>{}, S0{}, 0);
    store_raw(wn_ic<1>{}, wn_ic<0>{}, S0{}, 0);
    store_raw(wn_ic<0>{}, wn_ic<1>{}, S1{}, 1 < chunks ? 1 : -1);
    store_raw(wn_ic<1>{}, wn_ic<1>{}, S1{}, 1 < chunks ? 1 : -1);
    load_raw(cl(2), wn_ic<0>{}, S0{});
    load_raw(cl(2), wn_ic<1>{}, S0{});
    load_raw(cl(3), wn_ic<0>{}, S1{});
    load_raw(cl(3), wn_ic<1>{}, S1{});
    __syncthreads();
    wn_for<0, 4>([&](auto ac) {
      t_rd(ac, wn_ic<0>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t_wr(ac, wn_ic<0>{});
    });
  } else {
    load_raw(0, wn_ic<0>{}, S0{});
    load_raw(0, wn_ic<1>{}, S0{});
    wn_for<0, 4>([&](auto ic) { dma_u(0, wn_ic<0>{}, ic); });
    vm_done();
    store_raw(wn_ic<0>{}, wn_ic<0>{}, S0{}, 0);
    store_raw(wn_ic<1>{}, wn_ic<0>{}, S0{}, 0);
    load_raw(cl(1), wn_ic<0>{}, S0{});
    load_raw(cl(1), wn_ic<1>{}, S0{});
    __syncthreads();
    wn_for<0, 4>([&](auto ac) {
      t_rd(ac, wn_ic<0>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t_wr(ac, wn_ic<0>{});
    });
    vm_done();
    store_raw(wn_ic<0>{}, wn_ic<1>{}, S0{}, 1 < chunks ? 1 : -1);
    store_raw(wn_ic<1>{}, wn_ic<1>{}, S0{}, 1 < chunks ? 1 : -1);
    load_raw(cl(2), wn_ic<0>{}, S0{});
    load_raw(cl(2), wn_ic<1>{}, S0{});
  }
  // One chunk, stages of parity B (compile time: every LDS offset of the body is an immediate).
  // Branch-free, ONE barrier.  One stage per xi: its 4 MFMAs, the fragment reads of the next xi,
  // and one piece of the staging pipeline: the raw patch of chunk c + 2 from the registers to
  // LDS, the loads of chunk c + 3 (two register sets: c + 4), U of chunk c + 1 by DMA, the input
  // transform of chunk c + 1 (raw patch double-buffered, so it needs no barrier of its own) -
  // each stage fenced so nothing bunches up.
  // At the barrier: every DMA of this wave has landed, every wave is done with stage B ^ 1, with
  // raw buffer B (transformed an iteration ago) and has written raw buffer B ^ 1.
  // (bare s_barrier: __syncthreads() adds nothing we need)
  auto chunk_body = [&](int c, auto bc) {
    constexpr int B = decltype(bc)::value;
    using NB = wn_ic<B ^ 1>;
    using SB = wn_ic<(NSET == 2 ? B : 0)>;        // the register set this iteration stores / reloads
    if constexpr (NSET == 2) {
      // everything older than the previous iteration's loads (its DMA pieces): those loads - two
      // buffer loads and, ACT, two coefficient loads - may stay in flight
      if constexpr (ACT) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    reg_anchor(ca[SB::value], cb[SB::value]);   // (the coefficients are behind the wait)
    if constexpr (DZ) { reg_anchor(cz[0], cz[1]); reg_anchor(cz[2], cz[3]); reg_anchor(cz[4]); }
    frag(bc, wn_ic<0>{});
    wn_for<0, 16>([&](auto xc) {
      constexpr int x = decltype(xc)::value;
      if constexpr (FS == 2) {
        // transform row x - 8 of chunk c + 1: read ahead of this stage's MFMAs, combined and
        // written behind them
        constexpr bool T = x >= 8 && x < 12;
        if constexpr (T) t_rd(wn_ic<(T ? x - 8 : 0)>{}, NB{});
        stage(bc, xc, wn_ic<(T ? 4 : 0)>{});
        if constexpr (T) {
          lds_wait<3>(tr[0], tr[1], tr[2], tr[3]);     // (only the fragment reads stay in flight)
          t_wr(wn_ic<(T ? x - 8 : 0)>{}, NB{});
        }
      } else {
        // one fragment set: the row is read behind stage x - 1's MFMAs (with the fragments of
        // xi) and written behind this stage's, whose initial wait covers both
        stage(bc, xc, wn_ic<0>{});
        if constexpr (x >= 8 && x < 12) t_wr(wn_ic<(x >= 8 && x < 12 ? x - 8 : 0)>{}, NB{});
        if constexpr (x >= 7 && x < 11) t_rd(wn_ic<(x >= 7 && x < 11 ? x - 7 : 0)>{}, NB{});
      }
      if constexpr (NSET == 2) {
        // chunk c + 2 (loaded two iterations ago) to LDS, the DMA, then the loads of chunk c + 4
        if constexpr (x == 0) store_raw(wn_ic<0>{}, bc, SB{}, c + 2 < chunks ? c + 2 : -1);
        if constexpr (x == 1) store_raw(wn_ic<1>{}, bc, SB{}, c + 2 < chunks ? c + 2 : -1);
        if constexpr (x >= 2 && x < 6) dma_u(cl(c + 1), NB{}, wn_ic<(x >= 2 && x < 6 ? x - 2 : 0)>{});
        if constexpr (x == 6) load_raw(cl(c + 4), wn_ic<0>{}, SB{});
        if constexpr (x == 7) load_raw(cl(c + 4), wn_ic<1>{}, SB{});
      } else {
        if constexpr (x == 0) store_raw(wn_ic<0>{}, bc, SB{}, c + 2 < chunks ? c + 2 : -1);   // chunk c + 2 (loaded an iteration ago)
        if constexpr (x == 1) load_raw(cl(c + 3), wn_ic<0>{}, SB{});
        if constexpr (x == 2) store_raw(wn_ic<1>{}, bc, SB{}, c + 2 < chunks ? c + 2 : -1);
        if constexpr (x == 3) load_raw(cl(c + 3), wn_ic<1>{}, SB{});
        if constexpr (x >= 4 && x < 8) dma_u(cl(c + 1), NB{}, wn_ic<(x >= 4 && x < 8 ? x - 4 : 0)>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  for (int c = 0; c < chunks; c += 2) {
    chunk_body(c, wn_ic<0>{});
    if (c + 1 < chunks) chunk_body(c + 1, wn_ic<1>{});
  }

  // ---- epilogue: Y = A^T M A per (tile, channel), register-local ----
  // (tried: rows = channels / columns = tiles, i.e. the MFMA operands swapped, for 16-byte
  // stores of four consecutive channels per lane - the K loop itself ran 3-5 % slower with the
  // U fragment in the A slot, which cost more than the wider stores gained)
  // (lane geometry re-derived here - the empty asm hides it from common-subexpression
  // elimination - so that nothing epilogue-only stays in a register through the K loop)
  int etid = threadIdx.x;
  asm volatile("" : "+v"(etid));
  const int fm_e = etid & 15, fk_e = (etid >> 4) & 3, tg_e = (etid >> 6) & 3, nh_e = etid >> 8;
  // lane holds tiles 16 tg_e + 4 (lane >> 4) + r (r = 0..3) = tile row tg_e, tile column
  // 4 (lane >> 4) + r, of channels 32 nh_e + 16 b + (lane & 15)
  float yv[2][4][4];   // [block][r][2 dy + dx]
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = n0 + 32 * nh_e + 16 * b + fm_e;
    const float bv = p.bias ? p.bias[p.n_off + col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[i][0] = acc[4 * i + 0][b][r] + acc[4 * i + 1][b][r] + acc[4 * i + 2][b][r];
        s[i][1] = acc[4 * i + 1][b][r] - acc[4 * i + 2][b][r] - acc[4 * i + 3][b][r];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        yv[b][r][j] = s[0][j] + s[1][j] + s[2][j] + bv;
        yv[b][r][2 + j] = s[1][j] - s[2][j] - s[3][j] + bv;
      }
    }
  }
  const int oy = y0 + 2 * tg_e, ox = x0 + 2 * (4 * fk_e);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = n0 + 32 * nh_e + 16 * b + fm_e;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t pix = ((size_t)n * H + oy + (q >> 1)) * W + ox + 2 * r + (q & 1);
        p.out[pix * p.ldo + col] = yv[b][r][q];
      }
  }
  // reduction scratch: the K loop's last LDS reads precede the barrier inside the block helpers
  float2* red = reinterpret_cast<float2*>(Rs);
  if (STATS && p.stats) {   // uniform: (mean, M2) of this block's 256 pixels per channel
    __syncthreads();        // every wave is past its last fragment read
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) sm += yv[b][r][q];
      float mean = sm * (1.f / 16.f), m2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float d = yv[b][r][q] - mean;
          m2 = fmaf(d, d, m2);
        }
      // lanes l, l ^ 16, l ^ 32 hold the other tile columns of the same channel
      wf_merge_eq(mean, m2, __shfl_xor(mean, 16, 64), __shfl_xor(m2, 16, 64), 16.f);
      wf_merge_eq(mean, m2, __shfl_xor(mean, 32, 64), __shfl_xor(m2, 32, 64), 32.f);
      if (fk_e == 0) red[tg_e * WN_BN + 32 * nh_e + 16 * b + fm_e] = float2{mean, m2};
    }
    float2 out;
    if (block_col_stats<WN_BN, 4>(red, 0, 0, false, float2{0.f, 0.f}, 64.f, out))
      p.stats[((size_t)n * p.stats_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
  if (BSTATS && p.bs_partial) {   // uniform: reductions of the NEXT backward stage (IgemmParams)
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = n0 + 32 * nh_e + 16 * b + fm_e;
      const BwdCoef cf = bwd_coef(p, n, col);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const size_t pix = ((size_t)n * H + oy + (q >> 1)) * W + ox + 2 * r + (q & 1);
          const float y = p.bs_y[pix * p.ldo + col];
          const float z = fmaf(y, cf.A, cf.B0);
          const float gz = yv[b][r][q] * cf.mk * (z > 0.f ? 1.f : p.slope);
          s1 += gz;
          s2 = fmaf(gz, (y - cf.mu) * cf.rs, s2);
        }
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (fk_e == 0) red[tg_e * WN_BN + 32 * nh_e + 16 * b + fm_e] = float2{s1, s2};
    }
    float2 out;
    if (block_col_sums<WN_BN, 4>(red, out))
      p.bs_partial[((size_t)n * p.bs_tiles + ty * tiles_x + tx) * p.Ncols + n0 + tid] = out;
  }
}

// U = G g G^T in the kernel's LDS image order.  One thread = one output column n and one chunk of
// 8 reduction channels: it transforms 8 filters and writes, per xi, their 8 (swizzled) values as
// two 16-byte stores - consecutive lanes = consecutive n, so a wave writes 2 KB runs.
// dir 0 (blockIdx.y): forward, n = co, k = ci, g = w[co][ci]; dir 1: data gradient, n = ci,
// k = co, g = w rotated by 180 degrees (the correlation of dy with the flipped filter).
__device__ __forceinline__ void wino_pack_items(const float* __restrict__ w,
                                                float* __restrict__ uf, float* __restrict__ ud,
                                                int Cout, int Cin, int dir, long long block) {
  float* dst = dir == 0 ? uf : ud;
  if (!dst) return;
  const int Nn = dir == 0 ? Cout : Cin, Kt = dir == 0 ? Cin : Cout;
  const int kchunks = Kt / WN_KC;
  const long long items = (long long)Nn * kchunks;
  const long long idx = block * 256 + threadIdx.x;
  if (idx >= items) return;
  const int nl = (int)(idx & 63);
  const long long rest = idx >> 6;
  const int kc = (int)(rest % kchunks), tn = (int)(rest / kchunks);
  const int nn = tn * WN_BN + nl;
  float u[16][8];   // [xi][channel of the chunk]
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int kk = kc * WN_KC + k;
    const int co = dir == 0 ? nn : kk, ci = dir == 0 ? kk : nn;
    const float* g = w + ((size_t)co * Cin + ci) * 9;
    float t4[4][3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const float g0 = dir == 0 ? g[0 + v] : g[6 + 2 - v];
      const float g1 = dir == 0 ? g[3 + v] : g[3 + 2 - v];
      const float g2 = dir == 0 ? g[6 + v] : g[0 + 2 - v];
      t4[0][v] = g0;
      t4[1][v] = 0.5f * (g0 + g1 + g2);
      t4[2][v] = 0.5f * (g0 - g1 + g2);
      t4[3][v] = g2;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      u[4 * a + 0][k] = t4[a][0];
      u[4 * a + 1][k] = 0.5f * (t4[a][0] + t4[a][1] + t4[a][2]);
      u[4 * a + 2][k] = 0.5f * (t4[a][0] - t4[a][1] + t4[a][2]);
      u[4 * a + 3][k] = t4[a][2];
    }
  }
  // wn_swz: the channel PAIRS of this row are XOR-permuted by bits 2-3 of the row - two
  // conditional pair swaps (selects, no indexed register access)
  const bool s1 = (nl >> 2) & 1, s2 = (nl >> 3) & 1;
  float* blk = dst + ((size_t)tn * kchunks + kc) * WN_BUF + nl * 8;
#pragma unroll
  for (int x = 0; x < 16; ++x) {
    f32x2v p0 = {u[x][0], u[x][1]}, p1 = {u[x][2], u[x][3]}, p2 = {u[x][4], u[x][5]},
           p3 = {u[x][6], u[x][7]};
    const f32x2v q0 = s1 ? p1 : p0, q1 = s1 ? p0 : p1, q2 = s1 ? p3 : p2, q3 = s1 ? p2 : p3;
    p0 = s2 ? q2 : q0; p1 = s2 ? q3 : q1; p2 = s2 ? q0 : q2; p3 = s2 ? q1 : q3;
    *reinterpret_cast<f32x4*>(blk + x * 512) = f32x4{p0[0], p0[1], p1[0], p1[1]};
    *reinterpret_cast<f32x4*>(blk + x * 512 + 4) = f32x4{p2[0], p2[1], p3[0], p3[1]};
  }
}

__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w,
                                                        float* __restrict__ uf,
                                                        float* __restrict__ ud, int Cout, int Cin) {
  wino_pack_items(w, uf, ud, Cout, Cin, blockIdx.y, blockIdx.x);
}
// every layer in one launch: `block_begin` of the device table maps blockIdx.x to (layer, block)
__global__ __launch_bounds__(256) void wino_pack_batched_kernel(
    const unet_wino_pack_entry* __restrict__ tab, int n) {
  int k = 0;
  for (int q = 1; q < n; ++q)
    if (tab[q].block_begin <= (int)blockIdx.x) k = q;
  const unet_wino_pack_entry e = tab[k];
  wino_pack_items(e.w, e.uf, e.ud, e.Cout, e.Cin, blockIdx.y, (long long)blockIdx.x - e.block_begin);
}

template <bool ACT, bool STATS, bool BSTATS, bool UP = false, bool DZ = false>
int launch_wino(const WinoParams& wp, hipStream_t stream) {
  auto kern = conv_wino_kernel<ACT, STATS, BSTATS, UP, DZ>;
  UNET_SET_DYN_LDS(kern, WN_LDS);
  const IgemmParams& p = wp.g;
  const long long tiles =
      (long long)p.N * (p.Hin / WN_TH) * (p.Win / WN_TW) * (p.Ncols / WN_BN);
  const long long blocks = tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), WN_LDS, stream, wp);
  UNET_CHECK_LAUNCH("conv_wino");
  return UNET_OK;
}

// shapes the kernel tiles: image as 8 x 32 pixels, K in chunks of 8, 64-wide column tiles, every
// tensor inside one 2 GiB buffer descriptor, and enough workgroups / K depth to pay for the
// per-workgroup prologue (one workgroup per CU: nothing else overlaps it)
bool wino_shape_ok(int N, int H, int W, int K0, int K1, int ncols, int n_off, bool up = false) {
  const int K = K0 + K1;
  if (H % WN_TH || W % WN_TW || K0 % WN_KC || K1 % WN_KC || ncols % WN_BN || n_off % WN_BN)
    return false;
  if (K < 64 || (up && K0 < WN_KC)) return false;
  const long long px = (long long)N * H * W;
  const long long b0 = (up ? px / 4 : px) * K0 * 4, b1 = px * K1 * 4;
  if (b0 >= (1LL << 31) || b1 >= (1LL << 31) || px * ncols * 4 >= (1LL << 31)) return false;
  return px / (WN_TH * WN_TW) * (ncols / WN_BN) >= 256;
}

}  // namespace
}  // namespace unet_conv

using namespace unet_conv;

extern "C" int unet_conv_wino_supported(int N, int H, int W, int C0, int C1, int Cout) {
  return wino_shape_ok(N, H, W, C0, C1, Cout, 0) ? 1 : 0;
}

extern "C" int unet_conv_up_wino_supported(int N, int H, int W, int C0, int C1, int Cout) {
  return wino_shape_ok(N, H, W, C0, C1, Cout, 0, true) ? 1 : 0;
}

extern "C" size_t unet_wino_weight_floats(int Cout, int Cin) { return (size_t)16 * Cout * Cin; }

extern "C" int unet_pack_wino_weights(const float* w_oihw, float* uf, float* ud, int Cout, int Cin,
                                      unet_stream_t stream) {
  UNET_REQUIRE(w_oihw && (uf || ud) && Cout > 0 && Cin > 0, "pack_wino_weights: bad argument");
  UNET_REQUIRE(!uf || (Cout % WN_BN == 0 && Cin % WN_KC == 0),
               "pack_wino_weights: forward form needs Cout %% 64 == 0, Cin %% 8 == 0");
  UNET_REQUIRE(!ud || (Cin % WN_BN == 0 && Cout % WN_KC == 0),
               "pack_wino_weights: data-gradient form needs Cin %% 64 == 0, Cout %% 8 == 0");
  const long long items = (long long)Cout * Cin / WN_KC;     // (column, 8-channel chunk) per form
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((items + 255) / 256), 2), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, uf, ud, Cout, Cin);
  UNET_CHECK_LAUNCH("wino_pack");
  return UNET_OK;
}

// All layers of a table in one launch (13 launches of 15 us each per step otherwise).  Entry k
// covers blocks [block_begin_k, block_begin_k + Cout_k * Cin_k / 8 / 256 rounded up).
extern "C" int unet_pack_wino_weights_batched(const unet_wino_pack_entry* table_device, int n,
                                              int total_blocks, unet_stream_t stream) {
  UNET_REQUIRE(table_device && n > 0 && n <= 256 && total_blocks > 0,
               "pack_wino_weights_batched: bad argument");
  hipLaunchKernelGGL(wino_pack_batched_kernel, dim3((unsigned)total_blocks, 2), dim3(256), 0,
                     (hipStream_t)stream, table_device, n);
  UNET_CHECK_LAUNCH("wino_pack_batched");
  return UNET_OK;
}

// Fused layer forward (as unet_conv_in_fwd with ksize 3, stride 1) on the Winograd kernel;
// wu = the forward form of unet_pack_wino_weights.  The shape must satisfy
// unet_conv_wino_supported.  Workspace / stats_px_out as unet_conv_in_fwd.
static int conv_in_fwd_wino_impl(const unet_act_src* s0, const unet_act_src* s1, float slope,
                                 const float* wu, const float* bias, float* y, void* workspace,
                                 size_t workspace_bytes, int* stats_px_out, int N, int H, int W,
                                 int Cout, hipStream_t stream, bool up) {
  UNET_REQUIRE(s0 && s0->x && wu && y && stats_px_out, "conv_in_fwd_wino: null pointer");
  const int C0 = s0->C, C1 = s1 ? s1->C : 0;
  UNET_REQUIRE(C1 == 0 || s1->x, "conv_in_fwd_wino: second source is null with C1=%d", C1);
  UNET_REQUIRE(s0->alpha && s0->beta && (!s1 || (s1->alpha && s1->beta)),
               "conv_in_fwd_wino: the Winograd loader takes activated sources (alpha / beta set)");
  UNET_REQUIRE(wino_shape_ok(N, H, W, C0, C1, Cout, 0, up),
               "conv_in_fwd_wino: shape N=%d %dx%d C=(%d,%d)->%d not tiled by the Winograd kernel",
               N, H, W, C0, C1, Cout);
  UNET_REQUIRE(!up || (H % 2 == 0 && W % 2 == 0 && C1 > 0), "conv_up_in_fwd_wino: bad shape");
  const int tiles = H * W / 256;
  const size_t need = (size_t)N * tiles * Cout * sizeof(float2);
  if (workspace && workspace_bytes < need) {
    unet_set_error("conv_in_fwd_wino: workspace %zu < %zu bytes", workspace_bytes, need);
    return UNET_E_WORKSPACE;
  }
  WinoParams wp{};
  IgemmParams& p = wp.g;
  p.src0 = s0->x; p.src1 = s1 ? s1->x : nullptr; p.C0 = C0; p.C1 = C1;
  p.act0_alpha = s0->alpha; p.act0_beta = s0->beta;
  p.act1_alpha = s1 ? s1->alpha : nullptr;
  p.act1_beta = s1 ? s1->beta : nullptr;
  p.slope = slope;
  p.src0_bytes = (unsigned)((long long)N * (up ? H / 2 : H) * (up ? W / 2 : W) * C0 * 4);
  p.src1_bytes = (unsigned)((long long)N * H * W * C1 * 4);
  p.bias = bias; p.n_off = 0;
  p.out = y; p.ldo = Cout; p.N = N; p.Hin = p.Hl = p.Hout = H; p.Win = p.Wl = p.Wout = W;
  p.Ncols = Cout;
  p.stats = reinterpret_cast<float2*>(workspace);
  p.stats_tiles = tiles;
  wp.wu = wu;
  *stats_px_out = workspace ? 256 : 0;
  if (up) return launch_wino<true, true, false, true>(wp, stream);
  return launch_wino<true, true, false>(wp, stream);
}

extern "C" int unet_conv_in_fwd_wino(const unet_act_src* s0, const unet_act_src* s1, float slope,
                                     const float* wu, const float* bias, float* y, void* workspace,
                                     size_t workspace_bytes, int* stats_px_out, int N, int H,
                                     int W, int Cout, unet_stream_t stream) {
  return conv_in_fwd_wino_impl(s0, s1, slope, wu, bias, y, workspace, workspace_bytes,
                               stats_px_out, N, H, W, Cout, (hipStream_t)stream, false);
}

// unet_conv_up_in_fwd on the Winograd kernel: low = [N][H/2][W/2][C0], skip = [N][H][W][C1].
extern "C" int unet_conv_up_in_fwd_wino(const unet_act_src* low, const unet_act_src* skip,
                                        float slope, const float* wu, const float* bias, float* y,
                                        void* workspace, size_t workspace_bytes,
                                        int* stats_px_out, int N, int H, int W, int Cout,
                                        unet_stream_t stream) {
  return conv_in_fwd_wino_impl(low, skip, slope, wu, bias, y, workspace, workspace_bytes,
                               stats_px_out, N, H, W, Cout, (hipStream_t)stream, true);
}

// Data gradient (as unet_conv3x3_bwd_data_bs with stride 1, accumulate 0) on the Winograd
// kernel; ud = the data-gradient form of unet_pack_wino_weights of the WHOLE weight
// [Cout][Cin_total]; dx covers input channels [ci_offset, ci_offset + Ccols), ci_offset % 64 == 0.
// bs may be null (no reductions).  bs->tiles_out receives the reduction tiles per image.
extern "C" int unet_conv3x3_bwd_data_bs_wino(const float* dy, const float* ud, int Cin_total,
                                             int ci_offset, float* dx, int N, int H, int W,
                                             int Cout, int Ccols, unet_bwd_stats* bs,
                                             unet_stream_t stream) {
  UNET_REQUIRE(dy && ud && dx, "conv3x3_bwd_data_bs_wino: null pointer");
  UNET_REQUIRE(ci_offset >= 0 && ci_offset + Ccols <= Cin_total && Cin_total % WN_BN == 0,
               "conv3x3_bwd_data_bs_wino: bad channel slice");
  UNET_REQUIRE(wino_shape_ok(N, H, W, Cout, 0, Ccols, ci_offset),
               "conv3x3_bwd_data_bs_wino: shape N=%d %dx%d %d->%d not tiled by the Winograd kernel",
               N, H, W, Cout, Ccols);
  WinoParams wp{};
  IgemmParams& p = wp.g;
  p.src0 = dy; p.C0 = Cout; p.C1 = 0;
  p.src0_bytes = (unsigned)((long long)N * H * W * Cout * 4);
  p.n_off = ci_offset;
  p.out = dx; p.ldo = Ccols; p.N = N; p.Hin = p.Hl = p.Hout = H; p.Win = p.Wl = p.Wout = W;
  p.Ncols = Ccols;
  wp.wu = ud;
  const bool use_bs = bs && bs->y && bs->mean && bs->rstd && bs->gamma && bs->beta &&
                      bs->partial &&
                      bs->partial_bytes >= (size_t)N * (H * W / 256) * Ccols * sizeof(float2);
  if (bs) bs->tiles_out = 0;
  if (use_bs) {
    p.bs_y = bs->y; p.bs_mean = bs->mean; p.bs_rstd = bs->rstd; p.bs_gamma = bs->gamma;
    p.bs_beta = bs->beta; p.bs_mask = bs->mask; p.slope = bs->slope;
    p.bs_partial = reinterpret_cast<float2*>(bs->partial);
    p.bs_tiles = H * W / 256; p.bs_tile0 = 0;
    bs->tiles_out = p.bs_tiles;
    return launch_wino<false, false, true>(wp, (hipStream_t)stream);
  }
  return launch_wino<false, false, false>(wp, (hipStream_t)stream);
}

// unet_conv3x3_bwd_data_bs_wino with the InstanceNorm + LeakyReLU + dropout backward of the
// layer applied ON LOAD (DZ mode of the kernel): g = dL/da (the gradient of the layer's activated
// output), y = its raw convolution output, coef5 / sums from unet_instnorm_bwd_coefs.  Computes
// dx as the data gradient of dz = dL/dy, WRITES dz (dz_out: [N][H][W][Cout], must not alias g)
// for the layer's weight gradient, and the layer's dgamma / dbeta / dbias (each may be null).
extern "C" int unet_conv3x3_bwd_data_dz_wino(const float* g, const float* y, const float* coef5,
                                             const float* sums, const float* gamma,
                                             const float* rstd, float slope, float* dz_out,
                                             float* dgamma, float* dbeta, float* dbias,
                                             const float* ud, int Cin_total, int ci_offset,
                                             float* dx, int N, int H, int W, int Cout, int Ccols,
                                             unet_bwd_stats* bs, unet_stream_t stream) {
  UNET_REQUIRE(g && y && coef5 && sums && gamma && rstd && dz_out && ud && dx,
               "conv3x3_bwd_data_dz_wino: null pointer");
  UNET_REQUIRE(dz_out != g, "conv3x3_bwd_data_dz_wino: dz must not alias g (halo pixels are re-read)");
  UNET_REQUIRE(ci_offset >= 0 && ci_offset + Ccols <= Cin_total && Cin_total % WN_BN == 0,
               "conv3x3_bwd_data_dz_wino: bad channel slice");
  UNET_REQUIRE(wino_shape_ok(N, H, W, Cout, 0, Ccols, ci_offset),
               "conv3x3_bwd_data_dz_wino: shape N=%d %dx%d %d->%d not tiled by the Winograd kernel",
               N, H, W, Cout, Ccols);
  WinoParams wp{};
  IgemmParams& p = wp.g;
  p.src0 = g; p.C0 = Cout; p.C1 = 0;
  p.src0_bytes = (unsigned)((long long)N * H * W * Cout * 4);
  p.n_off = ci_offset;
  p.out = dx; p.ldo = Ccols; p.N = N; p.Hin = p.Hl = p.Hout = H; p.Win = p.Wl = p.Wout = W;
  p.Ncols = Ccols;
  wp.wu = ud;
  wp.dz_y = y; wp.dz_coef = coef5; wp.dz_out = dz_out;
  wp.dz_sums = reinterpret_cast<const float2*>(sums);
  wp.dz_gamma = gamma; wp.dz_rstd = rstd; wp.dz_slope = slope;
  wp.dz_dgamma = dgamma; wp.dz_dbeta = dbeta; wp.dz_dbias = dbias;
  const bool use_bs = bs && bs->y && bs->mean && bs->rstd && bs->gamma && bs->beta &&
                      bs->partial &&
                      bs->partial_bytes >= (size_t)N * (H * W / 256) * Ccols * sizeof(float2);
  if (bs) bs->tiles_out = 0;
  if (use_bs) {
    p.bs_y = bs->y; p.bs_mean = bs->mean; p.bs_rstd = bs->rstd; p.bs_gamma = bs->gamma;
    p.bs_beta = bs->beta; p.bs_mask = bs->mask; p.slope = bs->slope;
    p.bs_partial = reinterpret_cast<float2*>(bs->partial);
    p.bs_tiles = H * W / 256; p.bs_tile0 = 0;
    bs->tiles_out = p.bs_tiles;
    return launch_wino<false, false, true, false, true>(wp, (hipStream_t)stream);
  }
  return launch_wino<false, false, false, false, true>(wp, (hipStream_t)stream);
}

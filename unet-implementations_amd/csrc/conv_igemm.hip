// conv_igemm.hip — 3x3 convolution forward and data-gradient as one NHWC
// implicit-GEMM kernel on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
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
#include "common.h"

namespace {

constexpr int kMaxTaps = 9;

struct IgemmParams {
  const float* src0;
  const float* src1;
  int C0, C1;        // channels of the two (virtually concatenated) sources
  const float* w;    // packed weights, row (tap*Ktot + k), column stride 1, row stride ldw
  int ldw;
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
  int offy[kMaxTaps], offx[kMaxTaps], wtap[kMaxTaps];
  int Ncols;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const IgemmParams p) {
  constexpr int BK = 32;
  constexpr int LDA = BK + 4;  // 144-B rows: conflict-free ds_read_b128 across 16 pixel rows
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int A_PASSES = BM / 32;
  constexpr int B_PER_THREAD = (BK * BN / 4) / 256;
  constexpr int A_TILE = BM * LDA, B_TILE = BK * BN;
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

  // ---- A loader: thread -> (row lrow + 32*i, 16-B segment lseg) ----
  const int lrow = tid >> 3, lseg = tid & 7;
  int a_nb[A_PASSES], a_iy[A_PASSES], a_ix[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + lrow + 32 * i;
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
  // ---- B loader ----
  int b_row[B_PER_THREAD], b_c4[B_PER_THREAD];
#pragma unroll
  for (int j = 0; j < B_PER_THREAD; ++j) {
    const int idx = tid + 256 * j;
    b_row[j] = idx / (BN / 4);
    b_c4[j] = idx - b_row[j] * (BN / 4);
  }

  f32x4 ra[A_PASSES], rb[B_PER_THREAD];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int KS = p.ntaps * (Ktot / BK);

  auto load_tiles = [&](int t, int chunk) {
    const int oy = p.offy[t], ox = p.offx[t];
    const int c = chunk * BK;
    const float* src;
    int Cs, coff;
    if (c < p.C0) {
      src = p.src0; Cs = p.C0; coff = c;
    } else {
      src = p.src1; Cs = p.C1; coff = c - p.C0;
    }
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int iy = a_iy[i] + oy, ix = a_ix[i] + ox;
      const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok)
        v = *reinterpret_cast<const f32x4*>(
            src + (size_t)(a_nb[i] + iy * p.Win + ix) * Cs + coff + lseg * 4);
      ra[i] = v;
    }
    const float* wrow = p.w + ((size_t)p.wtap[t] * Ktot + c) * p.ldw + n0;
#pragma unroll
    for (int j = 0; j < B_PER_THREAD; ++j)
      rb[j] = *reinterpret_cast<const f32x4*>(wrow + (size_t)b_row[j] * p.ldw + b_c4[j] * 4);
  };
  auto store_tiles = [&](int buf) {
    float* Ab = As + buf * A_TILE;
    float* Bb = Bs + buf * B_TILE;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<f32x4*>(Ab + (lrow + 32 * i) * LDA + lseg * 4) = ra[i];
#pragma unroll
    for (int j = 0; j < B_PER_THREAD; ++j)
      *reinterpret_cast<f32x4*>(Bb + b_row[j] * BN + b_c4[j] * 4) = rb[j];
  };

  int t_next = 0, chunk_next = 0;
  auto advance = [&]() {
    if (++t_next == p.ntaps) { t_next = 0; ++chunk_next; }
  };

  load_tiles(t_next, chunk_next);
  advance();
  store_tiles(0);
  __syncthreads();

  for (int ks = 0; ks < KS; ++ks) {
    const int buf = ks & 1;
    const bool more = (ks + 1 < KS);
    if (more) {
      load_tiles(t_next, chunk_next);
      advance();
    }
    const float* Ab = As + buf * A_TILE + wm0 * LDA;
    const float* Bb = Bs + buf * B_TILE + wn0;
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) {
      f32x4 a[TM];
      float b[TN][4];
#pragma unroll
      for (int m = 0; m < TM; ++m)
        a[m] = *reinterpret_cast<const f32x4*>(Ab + (m * 32 + li) * LDA + kb + 4 * lh);
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) b[n][r] = Bb[(kb + 4 * lh + r) * BN + n * 32 + li];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][r], b[n][r], acc[m][n], 0, 0, 0);
    }
    if (more) store_tiles(buf ^ 1);
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
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int mg = m0 + row;
        if (mg < M) {
          size_t opix;
          if (direct) {
            opix = (size_t)mg;
          } else {
            const int nn = mg / HlWl;
            const int rr = mg - nn * HlWl;
            const int a = rr / p.Wl;
            const int b = rr - a * p.Wl;
            opix = ((size_t)nn * p.Hout + (a * p.sout + p.py)) * p.Wout + (b * p.sout + p.px);
          }
          float* o = p.out + opix * p.ldo + col;
          float v = acc[m][n][r] + bv;
          if (p.accumulate) v += *o;
          *o = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(const IgemmParams& p, hipStream_t stream) {
  constexpr int BK = 32, LDA = BK + 4;
  constexpr size_t lds = 2 * (size_t)(BM * LDA + BK * BN) * sizeof(float);
  static bool attr_set = false;
  auto kern = conv_igemm_kernel<BM, BN, WM, WN>;
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

int dispatch_igemm(const IgemmParams& p, hipStream_t stream) {
  const long long M = (long long)p.N * p.Hl * p.Wl;
  const int nc = p.Ncols;
  // Largest tile that still yields >= 256 workgroups (one per CU); otherwise the
  // small 64x64 tile.
  if (nc % 128 == 0 && ceil_div64(M, 128) * (nc / 128) >= 256)
    return launch_igemm<128, 128, 64, 64>(p, stream);
  if (nc % 64 == 0 && ceil_div64(M, 128) * (nc / 64) >= 256)
    return launch_igemm<128, 64, 64, 32>(p, stream);
  if (nc % 64 == 0 && M <= 128 * 256) return launch_igemm<64, 64, 32, 32>(p, stream);
  return launch_igemm<128, 32, 32, 32>(p, stream);
}

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
    const int k = i >> 5, c = i & 31;
    B[i] = (k < 27) ? wf[k * Cout + co0 + c] : 0.f;
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

void fill_fwd_taps(IgemmParams& p, int stride) {
  p.ntaps = 9;
  for (int t = 0; t < 9; ++t) {
    p.offy[t] = t / 3 - 1;
    p.offx[t] = t % 3 - 1;
    p.wtap[t] = t;
  }
  p.sin = stride;
  p.sout = 1;
  p.py = p.px = 0;
}

}  // namespace

extern "C" int unet_conv3x3_fwd(const float* x0, int C0, const float* x1, int C1, const float* wf,
                                const float* bias, float* y, int N, int H, int W, int Cout,
                                int stride, unet_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  UNET_REQUIRE(x0 && wf && y, "conv3x3_fwd: null pointer");
  UNET_REQUIRE(stride == 1 || stride == 2, "conv3x3_fwd: stride %d unsupported", stride);
  UNET_REQUIRE(N > 0 && H > 0 && W > 0, "conv3x3_fwd: bad shape");
  UNET_REQUIRE(Cout > 0 && Cout % 32 == 0, "conv3x3_fwd: Cout %d must be a multiple of 32", Cout);
  if (C0 == 3) {
    UNET_REQUIRE(C1 == 0 && stride == 1, "conv3x3_fwd: RGB stem is stride-1, single source");
    const long long M = (long long)N * H * W;
    dim3 grid((unsigned)ceil_div64(M, STEM_PIX), Cout / 32);
    hipLaunchKernelGGL(conv_stem_fwd_kernel, grid, dim3(256), 0, stream, x0, wf, bias, y, N, H, W,
                       Cout);
    UNET_CHECK_LAUNCH("conv_stem_fwd");
    return UNET_OK;
  }
  UNET_REQUIRE(C0 > 0 && C0 % 32 == 0 && C1 >= 0 && C1 % 32 == 0,
               "conv3x3_fwd: channel counts (%d,%d) must be multiples of 32", C0, C1);
  UNET_REQUIRE(C1 == 0 || x1, "conv3x3_fwd: x1 is null with C1=%d", C1);
  IgemmParams p{};
  p.src0 = x0; p.src1 = x1; p.C0 = C0; p.C1 = C1;
  p.w = wf; p.ldw = Cout; p.bias = bias;
  p.out = y; p.ldo = Cout; p.accumulate = 0;
  p.N = N; p.Hin = H; p.Win = W;
  p.Hl = p.Hout = (H - 1) / stride + 1;
  p.Wl = p.Wout = (W - 1) / stride + 1;
  p.Ncols = Cout;
  fill_fwd_taps(p, stride);
  return dispatch_igemm(p, stream);
}

extern "C" int unet_conv3x3_bwd_data(const float* dy, const float* wd, int ldw, float* dx, int N,
                                     int H, int W, int Cout, int Ccols, int stride, int accumulate,
                                     unet_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  UNET_REQUIRE(dy && wd && dx, "conv3x3_bwd_data: null pointer");
  UNET_REQUIRE(stride == 1 || stride == 2, "conv3x3_bwd_data: stride %d unsupported", stride);
  UNET_REQUIRE(Cout > 0 && Cout % 32 == 0 && Ccols > 0 && Ccols % 32 == 0 && ldw >= Ccols,
               "conv3x3_bwd_data: bad channel counts Cout=%d Ccols=%d ldw=%d", Cout, Ccols, ldw);
  UNET_REQUIRE(stride == 1 || (H % 2 == 0 && W % 2 == 0),
               "conv3x3_bwd_data: stride 2 needs even H, W");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  IgemmParams p{};
  p.src0 = dy; p.src1 = nullptr; p.C0 = Cout; p.C1 = 0;
  p.w = wd; p.ldw = ldw; p.bias = nullptr;
  p.out = dx; p.ldo = Ccols; p.accumulate = accumulate;
  p.N = N; p.Hin = Ho; p.Win = Wo;
  p.Hout = H; p.Wout = W;
  p.Ncols = Ccols;
  p.sin = 1;
  if (stride == 1) {
    p.Hl = H; p.Wl = W; p.sout = 1; p.py = p.px = 0;
    p.ntaps = 9;
    for (int t = 0; t < 9; ++t) {
      p.offy[t] = 1 - t / 3;
      p.offx[t] = 1 - t % 3;
      p.wtap[t] = t;
    }
    return dispatch_igemm(p, stream);
  }
  // stride 2: dx[2a+py][2b+px] = sum over ky with (py+1-ky) even of dy[a + (py+1-ky)/2][..]
  p.Hl = H / 2; p.Wl = W / 2; p.sout = 2;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      p.py = py; p.px = px;
      int nt = 0;
      for (int ky = 0; ky < 3; ++ky) {
        if ((py + 1 - ky) & 1) continue;
        for (int kx = 0; kx < 3; ++kx) {
          if ((px + 1 - kx) & 1) continue;
          p.offy[nt] = (py + 1 - ky) / 2;
          p.offx[nt] = (px + 1 - kx) / 2;
          p.wtap[nt] = ky * 3 + kx;
          ++nt;
        }
      }
      p.ntaps = nt;
      int rc = dispatch_igemm(p, stream);
      if (rc != UNET_OK) return rc;
    }
  return UNET_OK;
}

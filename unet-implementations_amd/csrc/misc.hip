// misc.hip — layout conversion, weight packing, bilinear 2x up-sampling,
// SGD-Nesterov and the library-level entry points (error string, version).
// All kernels here are HBM-bound streaming kernels (float4 per lane).
#include "conv_params.h"

static thread_local char g_err[512] = "";

void unet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* unet_last_error(void) { return g_err; }

namespace unet_conv {
long long& chunk_limit_bytes() {
  static long long lim = (1LL << 31) - 1;
  return lim;
}
}  // namespace unet_conv

extern "C" int unet_debug_set_chunk_limit(int64_t bytes) {
  unet_conv::chunk_limit_bytes() = (bytes > 0 && bytes < (1LL << 31)) ? bytes : (1LL << 31) - 1;
  return UNET_OK;
}
extern "C" int unet_abi_version(void) { return UNET_ABI_VERSION; }
extern "C" int unet_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

namespace {

// ---- NCHW <-> NHWC (small C) ------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                    long long HW, long long total) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    // i indexes the NHWC output: ((n*HW + p)*C + c)
    const int c = (int)(i % C);
    const long long np = i / C;
    const long long n = np / HW, p = np - n * HW;
    y[i] = x[(n * C + c) * HW + p];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                    long long HW, long long total) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    // i indexes the NCHW output: ((n*C + c)*HW + p)
    const long long p = i % HW;
    const long long nc = i / HW;
    const long long n = nc / C;
    const int c = (int)(nc - n * C);
    y[i] = x[(n * HW + p) * C + c];
  }
}

// ---- input pipeline: uint8 HWC image + uint8 mask -> normalised NHWC fp32 + int64 target ---
// image: ((v / 255) - mean[c]) / std[c] in the reference's operation order
// (Our_UNet/src/train.py:303-308); mask: values > 2 other than 255 become 0 (:300).
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ img,
                                                            const unsigned char* __restrict__ mask,
                                                            float* __restrict__ out,
                                                            long long* __restrict__ target,
                                                            long long pixels, float m0, float m1,
                                                            float m2, float s0, float s1, float s2) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += stride) {
    const unsigned char* s = img + p * 3;
    float* o = out + p * 3;
    o[0] = ((float)s[0] / 255.0f - m0) / s0;
    o[1] = ((float)s[1] / 255.0f - m1) / s1;
    o[2] = ((float)s[2] / 255.0f - m2) / s2;
    if (mask) {
      const unsigned char v = mask[p];
      target[p] = (v > 2 && v != 255) ? 0 : (long long)v;
    }
  }
}

// ---- 2-D transpose (1x1 weights for the data gradient) ------------------------------------
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src,
                                                          float* __restrict__ dst, int R, int C) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    t[ty + 8 * k][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < R && c < C) dst[(size_t)c * R + r] = t[tx][ty + 8 * k];
  }
}

// ---- weight packing -----------------------------------------------------------
// wf[t][co][ci] (forward: K = ci contiguous) and wd[t][ci][co] (dgrad: K = co contiguous)
// from w[co][ci][t]
__global__ void pack_w_kernel(const float* __restrict__ w, float* __restrict__ wf,
                              float* __restrict__ wd, int Cout, int Cin) {
  const long long total = (long long)9 * Cin * Cout;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    if (wd) {  // i = (t*Cin + ci)*Cout + co
      const int co = (int)(i % Cout);
      const long long r = i / Cout;
      const int ci = (int)(r % Cin), t = (int)(r / Cin);
      wd[i] = w[((size_t)co * Cin + ci) * 9 + t];
    }
    if (wf) {  // i = (t*Cout + co)*Cin + ci
      const int ci = (int)(i % Cin);
      const long long r = i / Cin;
      const int co = (int)(r % Cout), t = (int)(r / Cout);
      wf[i] = w[((size_t)co * Cin + ci) * 9 + t];
    }
  }
}

// bf16x3 form: the same two layouts, every weight split into its three bf16 terms
// (common.h split3); out = [plane][tap][row][col], planes 9*Cin*Cout elements apart.
__global__ void pack_w_split_kernel(const float* __restrict__ w, __bf16* __restrict__ wf3,
                                    __bf16* __restrict__ wd3, int Cout, int Cin) {
  const long long total = (long long)9 * Cin * Cout;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    if (wd3) {  // i = (t*Cin + ci)*Cout + co
      const int co = (int)(i % Cout);
      const long long r = i / Cout;
      const int ci = (int)(r % Cin), t = (int)(r / Cin);
      const float v = w[((size_t)co * Cin + ci) * 9 + t];
      const __bf16 h = (__bf16)v;
      const float r1 = v - (float)h;
      const __bf16 m = (__bf16)r1;
      wd3[i] = h; wd3[total + i] = m; wd3[2 * total + i] = (__bf16)(r1 - (float)m);
    }
    if (wf3) {  // i = (t*Cout + co)*Cin + ci
      const int ci = (int)(i % Cin);
      const long long r = i / Cin;
      const int co = (int)(r % Cout), t = (int)(r / Cout);
      const float v = w[((size_t)co * Cin + ci) * 9 + t];
      const __bf16 h = (__bf16)v;
      const float r1 = v - (float)h;
      const __bf16 m = (__bf16)r1;
      wf3[i] = h; wf3[total + i] = m; wf3[2 * total + i] = (__bf16)(r1 - (float)m);
    }
  }
}

// All layers in one launch.  A workgroup moves one [32 co][32 ci][9 taps] tile through LDS:
// the OIHW source is read as 32 runs of 288 contiguous floats, both packed layouts (and, on
// request, their three bf16 planes) are written in 128-byte segments.  `tile_begin` of the
// device table maps blockIdx.x to (layer, tile).
__global__ __launch_bounds__(256) void pack_w_batched_kernel(const unet_pack_entry* __restrict__ tab,
                                                             int n) {
  __shared__ float tile[32][289];   // [co][ci*9 + t]; 289 = 1 mod 32: conflict-free both ways
  int k = 0;
  for (int q = 1; q < n; ++q)
    if (tab[q].tile_begin <= (int)blockIdx.x) k = q;
  const unet_pack_entry e = tab[k];
  const int id = blockIdx.x - e.tile_begin;
  const int tiles_ci = (e.Cin + 31) / 32;
  const int co0 = (id / tiles_ci) * 32, ci0 = (id % tiles_ci) * 32;
  const int nci = min(32, e.Cin - ci0), run = nci * 9;
  const long long total = (long long)9 * e.Cin * e.Cout;
  for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
    const int co = idx / run, jj = idx - co * run;
    tile[co][jj] = e.w[((size_t)(co0 + co) * e.Cin + ci0) * 9 + jj];
  }
  __syncthreads();
  __bf16* wf3 = reinterpret_cast<__bf16*>(e.wf3);
  __bf16* wd3 = reinterpret_cast<__bf16*>(e.wd3);
  const bool one_plane = e.reserved == 1;   // (uniform) mixed precision: the rounded weight only
  auto put = [&](float* d, __bf16* d3, size_t i, float v) {
    if (d) d[i] = v;
    if (d3) {
      const __bf16 h = (__bf16)v;
      d3[i] = h;
      if (!one_plane) {
        const float r1 = v - (float)h;
        const __bf16 md = (__bf16)r1;
        d3[total + i] = md; d3[2 * total + i] = (__bf16)(r1 - (float)md);
      }
    }
  };
  if (e.wf || wf3)     // [t][co][ci]: consecutive lanes = consecutive ci
    for (int idx = threadIdx.x; idx < 9 * 32 * nci; idx += 256) {
      const int tp = idx / (32 * nci), rem = idx - tp * 32 * nci;
      const int co = rem / nci, ci = rem - co * nci;
      put(e.wf, wf3, ((size_t)tp * e.Cout + co0 + co) * e.Cin + ci0 + ci, tile[co][ci * 9 + tp]);
    }
  if (e.wd || wd3)     // [t][ci][co]: consecutive lanes = consecutive co
    for (int idx = threadIdx.x; idx < 9 * nci * 32; idx += 256) {
      const int tp = idx / (nci * 32), rem = idx - tp * nci * 32;
      const int ci = rem >> 5, co = rem & 31;
      put(e.wd, wd3, ((size_t)tp * e.Cin + ci0 + ci) * e.Cout + co0 + co, tile[co][ci * 9 + tp]);
    }
}

// ---- bilinear 2x, align_corners=False ---------------------------------------------
// out[2i]   = 0.25*in[i-1] + 0.75*in[i]   (i-1 clamped: out[0] = in[0])
// out[2i+1] = 0.75*in[i]   + 0.25*in[i+1] (i+1 clamped)
__device__ __forceinline__ void up_taps(int o, int n_in, int& i0, int& i1, float& w0, float& w1) {
  // PyTorch area_pixel_compute_source_index: src = max((o+0.5)*0.5 - 0.5, 0)
  float src = ((float)o + 0.5f) * 0.5f - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  w1 = src - (float)i0;
  w0 = 1.f - w1;
}

// alpha != nullptr (fused layer pipeline): x is a raw convolution output; its InstanceNorm +
// LeakyReLU + dropout is applied to each of the four taps before they are blended.
template <typename TS>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const TS* __restrict__ x,
                                                             TS* __restrict__ y, int h, int w,
                                                             int C, long long total4,
                                                             const float* __restrict__ alpha,
                                                             const float* __restrict__ beta,
                                                             float slope) {
  const int lpp = C >> 2;
  const int H2 = 2 * h, W2 = 2 * w;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const long long pix = i / lpp;
    const int c = (int)(i - pix * lpp) * 4;
    const int ox = (int)(pix % W2);
    const long long r = pix / W2;
    const int oy = (int)(r % H2);
    const long long n = r / H2;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    up_taps(oy, h, y0, y1, wy0, wy1);
    up_taps(ox, w, x0, x1, wx0, wx1);
    const TS* b = x + (size_t)n * h * w * C + c;
    f32x4 v00 = ld4(b + ((size_t)y0 * w + x0) * C);
    f32x4 v01 = ld4(b + ((size_t)y0 * w + x1) * C);
    f32x4 v10 = ld4(b + ((size_t)y1 * w + x0) * C);
    f32x4 v11 = ld4(b + ((size_t)y1 * w + x1) * C);
    if (alpha) {   // uniform
      const f32x4 al = *reinterpret_cast<const f32x4*>(alpha + (size_t)n * C + c);
      const f32x4 be = *reinterpret_cast<const f32x4*>(beta + (size_t)n * C + c);
      v00 = unet_conv::act4(v00, al, be, slope, true);
      v01 = unet_conv::act4(v01, al, be, slope, true);
      v10 = unet_conv::act4(v10, al, be, slope, true);
      v11 = unet_conv::act4(v11, al, be, slope, true);
    }
    const f32x4 o = (v00 * wx0 + v01 * wx1) * wy0 + (v10 * wx0 + v11 * wx1) * wy1;
    st4(y + i * 4, o);
  }
}

// The same for bf16 tensors (mixed-precision pipeline), organised by LOW-resolution pixel: one
// thread = one input pixel x 8 channels (16-byte loads / stores): its 3 x 3 neighbourhood is
// loaded and activated ONCE (9 loads for 4 outputs instead of 16, 9 activations instead of 16)
// and the 2 x 2 output block is blended with PyTorch's weights and operation order:
// output row 2i   = taps (i-1, i) x (0.25, 0.75)   [i = 0: the clamped source index is 0: (0, 1)]
// output row 2i+1 = taps (i, i+1) x (0.75, 0.25)   [i = h-1: tap i+1 is clamped onto i]
__global__ __launch_bounds__(256) void upsample2x_fwd_b16x8_kernel(
    const __bf16* __restrict__ x, __bf16* __restrict__ y, int h, int w, int C, long long total8,
    const float* __restrict__ alpha, const float* __restrict__ beta, float slope) {
  const int lpp = C >> 3;
  const int W2 = 2 * w;
  const long long stride = (long long)gridDim.x * 256;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total8; t += stride) {
    const long long pix = t / lpp;
    const int c = (int)(t - pix * lpp) * 8;
    const int j = (int)(pix % w);
    const long long r = pix / w;
    const int i = (int)(r % h);
    const long long n = r / h;
    f32x4 al[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, be[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (alpha) {   // uniform
      al[0] = *reinterpret_cast<const f32x4*>(alpha + (size_t)n * C + c);
      al[1] = *reinterpret_cast<const f32x4*>(alpha + (size_t)n * C + c + 4);
      be[0] = *reinterpret_cast<const f32x4*>(beta + (size_t)n * C + c);
      be[1] = *reinterpret_cast<const f32x4*>(beta + (size_t)n * C + c + 4);
    }
    const __bf16* b = x + (size_t)n * h * w * C + c;
    f32x4 v[3][3][2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      int yy = i - 1 + ky;
      yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int xx = j - 1 + kx;
        xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
        const bf16x8 q = *reinterpret_cast<const bf16x8*>(b + ((size_t)yy * w + xx) * C);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[ky][kx][0][k] = (float)q[k]; v[ky][kx][1][k] = (float)q[4 + k]; }
        if (alpha) {
          v[ky][kx][0] = unet_conv::act4(v[ky][kx][0], al[0], be[0], slope, true);
          v[ky][kx][1] = unet_conv::act4(v[ky][kx][1], al[1], be[1], slope, true);
        }
      }
    }
    const float wy[2][2] = {{i > 0 ? 0.25f : 0.f, i > 0 ? 0.75f : 1.f}, {0.75f, 0.25f}};
    const float wx[2][2] = {{j > 0 ? 0.25f : 0.f, j > 0 ? 0.75f : 1.f}, {0.75f, 0.25f}};
    __bf16* o = y + (((size_t)n * 2 * h + 2 * i) * W2 + 2 * j) * C + c;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        bf16x8 out;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x4 res = blend2x2(v[a][bb][hf], v[a][bb + 1][hf], v[a + 1][bb][hf],
                                                v[a + 1][bb + 1][hf], wx[bb][0], wx[bb][1],
                                                wy[a][0], wy[a][1]);
#pragma unroll
          for (int k = 0; k < 4; ++k) out[4 * hf + k] = (__bf16)res[k];
        }
        *reinterpret_cast<bf16x8*>(o + ((size_t)a * W2 + bb) * C) = out;
      }
  }
}

// gather form of the transpose: input row i receives
//   out[2i] * a(2i), out[2i+1] * a(2i+1), out[2i-1] * 0.25 (i>=1), out[2i+2] * 0.25 (i<=L-1)
// where the self-weights are 0.75 except at the clamped ends (1.0).
__device__ __forceinline__ int bwd_taps(int i, int n_in, int idx[4], float wt[4]) {
  const int L = n_in - 1;
  int k = 0;
  idx[k] = 2 * i;     wt[k] = (i == 0) ? 1.0f : 0.75f; ++k;
  idx[k] = 2 * i + 1; wt[k] = (i == L) ? 1.0f : 0.75f; ++k;
  if (i >= 1) { idx[k] = 2 * i - 1; wt[k] = 0.25f; ++k; }
  if (i < L)  { idx[k] = 2 * i + 2; wt[k] = 0.25f; ++k; }
  return k;
}

__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ gy,
                                                             float* gx, int h, int w, int C,
                                                             long long total4, int accumulate) {
  const int lpp = C >> 2;
  const int W2 = 2 * w;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const long long pix = i / lpp;
    const int c = (int)(i - pix * lpp) * 4;
    const int ix = (int)(pix % w);
    const long long r = pix / w;
    const int iy = (int)(r % h);
    const long long n = r / h;
    int ys[4], xs[4];
    float wy[4], wx[4];
    const int ny = bwd_taps(iy, h, ys, wy), nx = bwd_taps(ix, w, xs, wx);
    const float* b = gy + (size_t)n * (4 * (size_t)h * w) * C + c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a) {
      f32x4 row = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < nx; ++q)
        row += *reinterpret_cast<const f32x4*>(b + ((size_t)ys[a] * W2 + xs[q]) * C) * wx[q];
      acc += row * wy[a];
    }
    f32x4* o = reinterpret_cast<f32x4*>(gx + i * 4);
    if (accumulate) acc += *o;
    *o = acc;
  }
}

// ---- backward of conv3x3(upsample2x(a)) at LOW resolution -----------------------------------
// The bilinear up-sampling U is linear, so for y = conv3x3(U a) (zero padding):
//   dW[tap]  = sum_p (U a)[p + tap] (x) dy[p]            =  sum_q a[q] (x) D_tap[q]
//   dL/da[q] = U^T sum_tap W_tap^T shift_tap(dy)          =  sum_tap W_tap^T D_tap[q]
// with  D_tap = U^T shift_tap(dy)   (shift_tap(dy)[P] = dy[P - off_tap], zero outside the image),
// i.e. both gradients are plain GEMMs over the LOW-resolution pixels q (a quarter of the
// positions: 1/4 of the FLOPs of running the 3x3 weight / data gradient on the up-sampled
// tensor) once dy has been reduced to the nine tensors D_tap.  This kernel writes
// D[n][i][j][tap * C + c] (9C channels per low-resolution pixel) from dy[n][2h][2w][C].
// One thread = one low-resolution pixel x 4 channels: 6 x 6 dy pixels in, 9 x 4 sums out.
// V = float4 groups per thread: 1 (4 channels) or, for bf16 tensors, 2 (8 channels = 16-byte
// loads and stores: half the memory instructions of this load-issue-bound kernel).
template <typename TS, int V = 1>
__global__ __launch_bounds__(256) void upsample2x_bwd_taps_kernel(const TS* __restrict__ dy,
                                                                  TS* __restrict__ D, int h,
                                                                  int w, int C, long long total4) {
  const int lpp = C / (4 * V);
  const int H2 = 2 * h, W2 = 2 * w;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i4 = (long long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += stride) {
    const long long pix = i4 / lpp;
    const int c = (int)(i4 - pix * lpp) * 4 * V;
    const int j = (int)(pix % w);
    const long long r = pix / w;
    const int i = (int)(r % h);
    const long long n = r / h;
    // 1-D transposed stencil of low index i: up rows 2i-1, 2i, 2i+1, 2i+2 with these weights
    // (0 where the row does not exist or does not touch x[i])
    float wy[4], wx[4];
    wy[0] = i >= 1 ? 0.25f : 0.f;
    wy[1] = i == 0 ? 1.0f : 0.75f;
    wy[2] = i == h - 1 ? 1.0f : 0.75f;
    wy[3] = i < h - 1 ? 0.25f : 0.f;
    wx[0] = j >= 1 ? 0.25f : 0.f;
    wx[1] = j == 0 ? 1.0f : 0.75f;
    wx[2] = j == w - 1 ? 1.0f : 0.75f;
    wx[3] = j < w - 1 ? 0.25f : 0.f;
    const TS* b = dy + (size_t)n * H2 * W2 * C + c;
    f32x4 acc[3][3][V];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int q = 0; q < V; ++q) acc[ky][kx][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // dy row R' = R - (ky - 1) for up row R = 2i - 1 + a: rows 2i-2 .. 2i+3, same for columns
#pragma unroll
    for (int rr = 0; rr < 6; ++rr) {
      const int Ry = 2 * i - 2 + rr;
      if ((unsigned)Ry >= (unsigned)H2) continue;
      f32x4 v[6][V];
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        const int Rx = 2 * j - 2 + cc;
        const bool in = (unsigned)Rx < (unsigned)W2;
        if constexpr (V == 2) {     // (bf16 only) one 16-byte load
          bf16x8 q8 = {};
          if (in) q8 = *reinterpret_cast<const bf16x8*>(b + ((size_t)Ry * W2 + Rx) * C);
#pragma unroll
          for (int k = 0; k < 4; ++k) { v[cc][0][k] = (float)q8[k]; v[cc][1][k] = (float)q8[4 + k]; }
        } else {
          v[cc][0] = in ? ld4(b + ((size_t)Ry * W2 + Rx) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      // column stencil: for tap kx, up column S = 2j-1+bb reads dy column S-(kx-1) = index
      // cc = bb + 2 - kx of v
      f32x4 colsum[3][V];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            const int S = 2 * j - 1 + bb;           // up column (must exist)
            const float wgt = (unsigned)S < (unsigned)W2 ? wx[bb] : 0.f;
            t += v[bb + 2 - kx][q] * wgt;
          }
          colsum[kx][q] = t;
        }
      }
      // row stencil: dy row Ry serves tap ky for up row R = Ry + ky - 1 = 2i-1+a, a = rr+ky-2
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int a = rr + ky - 2;
        if (a < 0 || a > 3) continue;
        const int R = 2 * i - 1 + a;
        const float wgt = (unsigned)R < (unsigned)H2 ? wy[a] : 0.f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int q = 0; q < V; ++q) acc[ky][kx][q] += colsum[kx][q] * wgt;
      }
    }
    TS* o = D + (size_t)pix * 9 * C + c;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if constexpr (V == 2) {
          bf16x8 out;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            out[k] = (__bf16)acc[ky][kx][0][k];
            out[4 + k] = (__bf16)acc[ky][kx][1][k];
          }
          *reinterpret_cast<bf16x8*>(o + (size_t)(ky * 3 + kx) * C) = out;
        } else {
          st4(o + (size_t)(ky * 3 + kx) * C, acc[ky][kx][0]);
        }
      }
  }
}

// ---- general bilinear resize of NCHW planes (align_corners = False) -------------------------
// F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False) as SimpleLoss applies it
// to logits whose size differs from the target's (Our_UNet/models/losses.py:66-68) and CLIP_UNet
// to the bottleneck features (CLIP_UNet/models/unet.py:444-450).  PyTorch's source index:
// src = max((dst + 0.5) * (in / out) - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, in - 1),
// weights (1 - l, l) with l = src - i0; blend order w0y (w0x p00 + w1x p01) + w1y (...).
__device__ __forceinline__ void bil_src(int dst, float scale, int in, int& i0, int& i1, float& l) {
  float src = ((float)dst + 0.5f) * scale - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 > in - 1 ? in - 1 : i0;
  i1 = i0 + 1 > in - 1 ? in - 1 : i0 + 1;
  l = src - (float)i0;
}
__global__ __launch_bounds__(256) void resize_bilinear_fwd_kernel(const float* __restrict__ x,
                                                                  float* __restrict__ y, int h,
                                                                  int w, int H, int W,
                                                                  long long total) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int X = (int)(i % W);
    const long long r = i / W;
    const int Y = (int)(r % H);
    const long long pl = r / H;
    int y0, y1, x0, x1;
    float ly, lx;
    bil_src(Y, sy, h, y0, y1, ly);
    bil_src(X, sx, w, x0, x1, lx);
    const float* p = x + pl * (long long)h * w;
    const float top = (1.f - lx) * p[(size_t)y0 * w + x0] + lx * p[(size_t)y0 * w + x1];
    const float bot = (1.f - lx) * p[(size_t)y1 * w + x0] + lx * p[(size_t)y1 * w + x1];
    y[i] = (1.f - ly) * top + ly * bot;
  }
}
// Adjoint, gather form (deterministic: no float atomics): input pixel (iy, ix) sums the output
// pixels whose stencil touches it.  Output rows that can reference input row iy have
// src in (iy - 1, iy + 1), i.e. dst in ((iy - 0.5) / scale - 0.5, (iy + 1.5) / scale - 0.5).
__device__ __forceinline__ void bil_range(int i, float scale, int out, int& lo, int& hi) {
  const float inv = 1.f / scale;
  lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out - 1 ? out - 1 : hi;
}
__device__ __forceinline__ float bil_coef(int dst, float scale, int in, int i) {
  int i0, i1;
  float l;
  bil_src(dst, scale, in, i0, i1, l);
  float c = 0.f;
  if (i0 == i) c += 1.f - l;
  if (i1 == i) c += l;
  return c;
}
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ gy,
                                                                  float* __restrict__ gx, int h,
                                                                  int w, int H, int W,
                                                                  long long total) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int ix = (int)(i % w);
    const long long r = i / w;
    const int iy = (int)(r % h);
    const long long pl = r / h;
    int ylo, yhi, xlo, xhi;
    bil_range(iy, sy, H, ylo, yhi);
    bil_range(ix, sx, W, xlo, xhi);
    const float* g = gy + pl * (long long)H * W;
    float acc = 0.f;
    for (int Y = ylo; Y <= yhi; ++Y) {
      const float cy = bil_coef(Y, sy, h, iy);
      if (cy == 0.f) continue;
      float row = 0.f;
      for (int X = xlo; X <= xhi; ++X) {
        const float cx = bil_coef(X, sx, w, ix);
        if (cx != 0.f) row += cx * g[(size_t)Y * W + X];
      }
      acc += cy * row;
    }
    gx[i] = acc;
  }
}

// ---- SGD with Nesterov momentum ------------------------------------------------
// hyper != nullptr: {lr, mu, wd, gscale} are read from device memory (graph-captured steps)
__global__ __launch_bounds__(256) void sgd_nesterov_kernel(float* __restrict__ p,
                                                           const float* __restrict__ g,
                                                           float* __restrict__ buf, long long n,
                                                           float lr, float mu, float wd,
                                                           int first_step, float gscale,
                                                           const float* __restrict__ hyper) {
  if (hyper) { lr = hyper[0]; mu = hyper[1]; wd = hyper[2]; gscale = hyper[3]; }
  const long long stride = (long long)gridDim.x * 256;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4) * gscale + pv * wd;
    f32x4 bv;
    if (first_step) bv = gv;
    else bv = *reinterpret_cast<f32x4*>(buf + i * 4) * mu + gv;
    *reinterpret_cast<f32x4*>(buf + i * 4) = bv;
    gv += bv * mu;
    pv -= gv * lr;
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
  }
  // tail
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    float pv = p[i];
    float gv = g[i] * gscale + pv * wd;
    float bv = first_step ? gv : buf[i] * mu + gv;
    buf[i] = bv;
    gv += bv * mu;
    p[i] = pv - gv * lr;
  }
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a,
                                                          const float* __restrict__ b,
                                                          long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) a[i] += b[i];
}

unsigned stream_grid(long long items) {
  long long b = ceil_div64(items, 256);
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int unet_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W,
                                 unet_stream_t stream) {
  UNET_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad argument");
  const long long total = (long long)N * C * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(stream_grid(total)), dim3(256), 0,
                     (hipStream_t)stream, x, y, C, (long long)H * W, total);
  UNET_CHECK_LAUNCH("nchw_to_nhwc");
  return UNET_OK;
}

extern "C" int unet_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W,
                                 unet_stream_t stream) {
  UNET_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad argument");
  const long long total = (long long)N * C * H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(stream_grid(total)), dim3(256), 0,
                     (hipStream_t)stream, x, y, C, (long long)H * W, total);
  UNET_CHECK_LAUNCH("nhwc_to_nchw");
  return UNET_OK;
}

extern "C" int unet_pack_conv3x3_weights(const float* w, float* wf, float* wd, int Cout, int Cin,
                                         unet_stream_t stream) {
  UNET_REQUIRE(w && (wf || wd) && Cout > 0 && Cin > 0, "pack_conv3x3_weights: bad argument");
  const long long total = (long long)9 * Cin * Cout;
  hipLaunchKernelGGL(pack_w_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, w,
                     wf, wd, Cout, Cin);
  UNET_CHECK_LAUNCH("pack_w");
  return UNET_OK;
}

extern "C" int unet_pack_conv3x3_weights_bf16x3(const float* w, uint16_t* wf3, uint16_t* wd3,
                                                int Cout, int Cin, unet_stream_t stream) {
  UNET_REQUIRE(w && (wf3 || wd3) && Cout > 0 && Cin > 0,
               "pack_conv3x3_weights_bf16x3: bad argument");
  const long long total = (long long)9 * Cin * Cout;
  hipLaunchKernelGGL(pack_w_split_kernel, dim3(stream_grid(total)), dim3(256), 0,
                     (hipStream_t)stream, w, reinterpret_cast<__bf16*>(wf3),
                     reinterpret_cast<__bf16*>(wd3), Cout, Cin);
  UNET_CHECK_LAUNCH("pack_w_split");
  return UNET_OK;
}

extern "C" int unet_pack_conv3x3_weights_batched(const unet_pack_entry* table_device, int n,
                                                 int total_tiles, unet_stream_t stream) {
  UNET_REQUIRE(table_device && n > 0 && n <= 256 && total_tiles > 0,
               "pack_conv3x3_weights_batched: bad argument");
  hipLaunchKernelGGL(pack_w_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0,
                     (hipStream_t)stream, table_device, n);
  UNET_CHECK_LAUNCH("pack_w_batched");
  return UNET_OK;
}

extern "C" int unet_upsample2x_fwd(const float* x, float* y, int N, int h, int w, int C,
                                   unet_stream_t stream) {
  UNET_REQUIRE(x && y && N > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0,
               "upsample2x_fwd: bad argument");
  const long long total4 = (long long)N * 4 * h * w * (C / 4);
  hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, dim3(stream_grid(total4)), dim3(256), 0,
                     (hipStream_t)stream, x, y, h, w, C, total4, (const float*)nullptr,
                     (const float*)nullptr, 0.f);
  UNET_CHECK_LAUNCH("upsample2x_fwd");
  return UNET_OK;
}

extern "C" int unet_upsample2x_in_fwd(const unet_act_src* x, float slope, float* up, int N, int h,
                                      int w, unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && up && N > 0 && h > 0 && w > 0 && x->C > 0 && x->C % 4 == 0 &&
                   (!x->alpha || x->beta),
               "upsample2x_in_fwd: bad argument");
  const long long total4 = (long long)N * 4 * h * w * (x->C / 4);
  hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, dim3(stream_grid(total4)), dim3(256), 0,
                     (hipStream_t)stream, x->x, up, h, w, x->C, total4, x->alpha, x->beta, slope);
  UNET_CHECK_LAUNCH("upsample2x_fwd");
  return UNET_OK;
}

extern "C" int unet_upsample2x_in_fwd_b16(const unet_act_src* x, float slope, uint16_t* up, int N,
                                          int h, int w, unet_stream_t stream) {
  UNET_REQUIRE(x && x->x && up && N > 0 && h > 0 && w > 0 && x->C > 0 && x->C % 4 == 0 &&
                   (!x->alpha || x->beta),
               "upsample2x_in_fwd_b16: bad argument");
  if (x->C % 8 == 0) {   // one thread per low-resolution pixel x 8 channels
    const long long total8 = (long long)N * h * w * (x->C / 8);
    hipLaunchKernelGGL(upsample2x_fwd_b16x8_kernel, dim3(stream_grid(total8)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const __bf16*>(x->x),
                       reinterpret_cast<__bf16*>(up), h, w, x->C, total8, x->alpha, x->beta, slope);
    UNET_CHECK_LAUNCH("upsample2x_fwd(b16 x8)");
    return UNET_OK;
  }
  const long long total4 = (long long)N * 4 * h * w * (x->C / 4);
  hipLaunchKernelGGL(upsample2x_fwd_kernel<__bf16>, dim3(stream_grid(total4)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const __bf16*>(x->x),
                     reinterpret_cast<__bf16*>(up), h, w, x->C, total4, x->alpha, x->beta, slope);
  UNET_CHECK_LAUNCH("upsample2x_fwd(b16)");
  return UNET_OK;
}

extern "C" int unet_upsample2x_bwd(const float* gy, float* gx, int N, int h, int w, int C,
                                   int accumulate, unet_stream_t stream) {
  UNET_REQUIRE(gy && gx && N > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0,
               "upsample2x_bwd: bad argument");
  const long long total4 = (long long)N * h * w * (C / 4);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(stream_grid(total4)), dim3(256), 0,
                     (hipStream_t)stream, gy, gx, h, w, C, total4, accumulate);
  UNET_CHECK_LAUNCH("upsample2x_bwd");
  return UNET_OK;
}

extern "C" int unet_upsample2x_bwd_taps(const float* dy, float* D, int N, int h, int w, int C,
                                        unet_stream_t stream) {
  UNET_REQUIRE(dy && D && N > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0,
               "upsample2x_bwd_taps: bad argument");
  const long long total4 = (long long)N * h * w * (C / 4);
  hipLaunchKernelGGL(upsample2x_bwd_taps_kernel<float>, dim3(stream_grid(total4)), dim3(256), 0,
                     (hipStream_t)stream, dy, D, h, w, C, total4);
  UNET_CHECK_LAUNCH("upsample2x_bwd_taps");
  return UNET_OK;
}

extern "C" int unet_upsample2x_bwd_taps_b16(const uint16_t* dy, uint16_t* D, int N, int h, int w,
                                            int C, unet_stream_t stream) {
  UNET_REQUIRE(dy && D && N > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0,
               "upsample2x_bwd_taps_b16: bad argument");
  if (C % 8 == 0) {   // 8 channels per thread: 16-byte loads and stores
    const long long total8 = (long long)N * h * w * (C / 8);
    hipLaunchKernelGGL((upsample2x_bwd_taps_kernel<__bf16, 2>), dim3(stream_grid(total8)), dim3(256),
                       0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(dy),
                       reinterpret_cast<__bf16*>(D), h, w, C, total8);
    UNET_CHECK_LAUNCH("upsample2x_bwd_taps(b16 x8)");
    return UNET_OK;
  }
  const long long total4 = (long long)N * h * w * (C / 4);
  hipLaunchKernelGGL(upsample2x_bwd_taps_kernel<__bf16>, dim3(stream_grid(total4)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const __bf16*>(dy),
                     reinterpret_cast<__bf16*>(D), h, w, C, total4);
  UNET_CHECK_LAUNCH("upsample2x_bwd_taps(b16)");
  return UNET_OK;
}

extern "C" int unet_sgd_nesterov_step(float* params, const float* grads, float* momentum,
                                      int64_t n, float lr, float mu, float weight_decay,
                                      int first_step, float grad_scale, unet_stream_t stream) {
  UNET_REQUIRE(params && grads && momentum && n > 0, "sgd_nesterov_step: bad argument");
  UNET_REQUIRE(((uintptr_t)params % 16 == 0) && ((uintptr_t)grads % 16 == 0) &&
                   ((uintptr_t)momentum % 16 == 0),
               "sgd_nesterov_step: arenas must be 16-byte aligned");
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, params, grads, momentum, (long long)n, lr, mu,
                     weight_decay, first_step, grad_scale, (const float*)nullptr);
  UNET_CHECK_LAUNCH("sgd_nesterov");
  return UNET_OK;
}

extern "C" int unet_sgd_nesterov_step_dev(float* params, const float* grads, float* momentum,
                                          int64_t n, const float* hyper, int first_step,
                                          unet_stream_t stream) {
  UNET_REQUIRE(params && grads && momentum && hyper && n > 0, "sgd_nesterov_step_dev: bad argument");
  UNET_REQUIRE(((uintptr_t)params % 16 == 0) && ((uintptr_t)grads % 16 == 0) &&
                   ((uintptr_t)momentum % 16 == 0),
               "sgd_nesterov_step_dev: arenas must be 16-byte aligned");
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, params, grads, momentum, (long long)n, 0.f, 0.f, 0.f,
                     first_step, 1.f, hyper);
  UNET_CHECK_LAUNCH("sgd_nesterov(dev)");
  return UNET_OK;
}

extern "C" int unet_add_inplace(float* a, const float* b, int64_t n, unet_stream_t stream) {
  UNET_REQUIRE(a && b && n > 0, "add_inplace: bad argument");
  hipLaunchKernelGGL(add_inplace_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream,
                     a, b, (long long)n);
  UNET_CHECK_LAUNCH("add_inplace");
  return UNET_OK;
}

extern "C" int unet_transpose2d(const float* src, float* dst, int R, int C, unet_stream_t stream) {
  UNET_REQUIRE(src && dst && R > 0 && C > 0, "transpose2d: bad argument");
  hipLaunchKernelGGL(transpose2d_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, R, C);
  UNET_CHECK_LAUNCH("transpose2d");
  return UNET_OK;
}

extern "C" int unet_preprocess_u8(const uint8_t* image_hwc, const uint8_t* mask, float* out_nhwc,
                                  int64_t* target, int N, int H, int W, const float* mean3,
                                  const float* std3, unet_stream_t stream) {
  UNET_REQUIRE(image_hwc && out_nhwc && mean3 && std3 && N > 0 && H > 0 && W > 0 &&
                   (mask == nullptr || target != nullptr),
               "preprocess_u8: bad argument");
  const long long pixels = (long long)N * H * W;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(stream_grid(pixels)), dim3(256), 0,
                     (hipStream_t)stream, image_hwc, mask, out_nhwc,
                     reinterpret_cast<long long*>(target), pixels, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2]);
  UNET_CHECK_LAUNCH("preprocess_u8");
  return UNET_OK;
}

extern "C" int unet_resize_bilinear_fwd(const float* x, float* y, int planes, int h, int w, int H,
                                        int W, unet_stream_t stream) {
  UNET_REQUIRE(x && y && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0,
               "resize_bilinear_fwd: bad argument");
  const long long total = (long long)planes * H * W;
  hipLaunchKernelGGL(resize_bilinear_fwd_kernel, dim3(stream_grid(total)), dim3(256), 0,
                     (hipStream_t)stream, x, y, h, w, H, W, total);
  UNET_CHECK_LAUNCH("resize_bilinear_fwd");
  return UNET_OK;
}

extern "C" int unet_resize_bilinear_bwd(const float* gy, float* gx, int planes, int h, int w, int H,
                                        int W, unet_stream_t stream) {
  UNET_REQUIRE(gy && gx && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0,
               "resize_bilinear_bwd: bad argument");
  const long long total = (long long)planes * h * w;
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(stream_grid(total)), dim3(256), 0,
                     (hipStream_t)stream, gy, gx, h, w, H, W, total);
  UNET_CHECK_LAUNCH("resize_bilinear_bwd");
  return UNET_OK;
}

// conv_params.h — the gather-GEMM launch descriptor shared by the convolution kernels
// (conv_igemm.hip: fp32 gather-GEMM, row-fused, stride-2 dgrad, stem; conv_patch.hip: the
// patch-staged kernels; conv_lowp.hip: bf16 and split-bf16 gather-GEMMs) and the entry points
// those translation units offer each other.
#pragma once
#include "common.h"

namespace unet_conv {

struct IgemmParams {
  const float* src0;
  const float* src1;
  int C0, C1;        // channels of the two (virtually concatenated) sources
  const float* w;    // packed weights [tap][n][k]: w[tap*tap_stride + (n_off + n)*Ktot + k]
  int tap_stride;
  int n_off;
  unsigned src0_bytes, src1_bytes, w_bytes;  // buffer-descriptor ranges (each < 2 GiB)
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
  // tap table packed 8 bits per tap (4 taps per word): bits 0-1 = offy+1, 2-3 = offx+1,
  // 4-7 = weight tap index.  Lives in SGPRs: no scalar-memory load per K step.
  unsigned tapw[3];
  int Ncols;
  // split-bf16 path: the weights as three bf16 planes, each laid out like `w`
  const __bf16* w3;
  int w3_plane;       // elements per plane
  unsigned w3_bytes;
};

inline void set_tap(IgemmParams& p, int t, int oy, int ox, int wt) {
  const unsigned e = (unsigned)(oy + 1) | ((unsigned)(ox + 1) << 2) | ((unsigned)wt << 4);
  p.tapw[t >> 2] |= e << ((t & 3) * 8);
}

// Epilogue of one 32x32 accumulator block: o[r] = destination of register r (nullptr = out of
// range).  With `accumulate` all 16 old values are loaded before the first add, so the reads
// overlap (a per-element load/add/store chain costs one HBM round trip per register).
#ifdef __HIPCC__
__device__ __forceinline__ void store_block16(float* const (&o)[16], const f32x16& acc, float bv,
                                              int accumulate) {
  if (accumulate) {
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = o[r] ? *o[r] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (o[r]) *o[r] = acc[r] + bv + old[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (o[r]) *o[r] = acc[r] + bv;
  }
}
#endif  // __HIPCC__

// ---- dispatchers implemented in the other translation units -------------------------------
bool patch_f32_applicable(const IgemmParams& p);              // conv_patch.hip
int launch_patch_f32_auto(const IgemmParams& p, hipStream_t stream);   // returns 1 if no tile fits
int launch_patch_split_auto(const IgemmParams& p, hipStream_t stream);
bool patch_split_applicable(const IgemmParams& p);
int dispatch_igemm(const IgemmParams& p, hipStream_t stream);          // conv_igemm.hip
int dispatch_igemm_bf16(const IgemmParams& p, hipStream_t stream);     // conv_lowp.hip
int dispatch_igemm_split(const IgemmParams& p, hipStream_t stream);    // conv_lowp.hip

}  // namespace unet_conv

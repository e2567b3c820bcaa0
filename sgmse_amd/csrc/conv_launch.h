// Launchers of the convolution kernels (tile selection happens in the engine / choose_conv_plan).
#pragma once
#include "kernels_conv.h"
#include "kernels_conv_pipe.h"
#include "kernels_conv1x1.h"
#include "kernels_conv_split.h"
#include "kernels_conv_wino.h"
#include "kernels_conv_wino2d.h"
#include "kernels_conv_thin.h"

#ifndef SGMSE_CONV_SPLIT_DEFAULT
#define SGMSE_CONV_SPLIT_DEFAULT 2     // 0: fp32 MFMA everywhere, 1: bf16x3 on the wide levels, 2: fp16x2 on the wide levels
#endif
#ifndef SGMSE_CONV_PIPE_DEFAULT
#define SGMSE_CONV_PIPE_DEFAULT 1
#endif

namespace sgmse {

// Measurement knob SGMSE_CONV_VARIANT (also the `variant` argument of sgmse_bench_conv):
//   bit 0: operand prefetch off (compiler-ordered LDS reads)   bit 1: element-wise instead of float4 input staging
//   bit 2: toggle the software-pipelined kernel (kernels_conv_pipe.h) for the 128 x 256 tiles
//   bit 3: 1x1 convolutions with 128-channel output blocks through the streaming kernel (kernels_conv1x1.h; measured
//          slower than the LDS-tiled kernel, kept selectable for the record)   bit 4: its 64-channel-chunk shape
//   bit 6 / bit 7: (sgmse_bench_conv only) the bf16x3 / fp16x2 split 3x3 kernel regardless of the engine's per-layer rule
inline int conv_variant() {
  static int v = [] { const char* e = getenv("SGMSE_CONV_VARIANT"); return e ? atoi(e) : 0; }();
  return v;
}

// tiles along gridDim.x: B x tile rows x tile columns of the (widest) utterance, or tile rows x the tile columns that exist (ConvArgs::rag_cols)
inline int conv_grid_tiles(const ConvArgs& a, int rows) {
  const int ty = (a.H + rows - 1) / rows;
  return a.rag_cols ? ty * a.rag_ncols : a.B * ty * ((a.W + 31) / 32);
}

template <int KS, int WC, int FC, int FP>
inline void launch_conv_mfma_t(const ConvArgs& a, drt::stream_t st, int variant, int ksplit = 1) {
  using T = ConvTile<KS, WC, FC, FP>;
  const int tiles = conv_grid_tiles(a, T::ROWS);
  dim3 grid(tiles, (a.Cout + T::CO_T - 1) / T::CO_T, ksplit);
  if (variant < 0) variant = conv_variant();
  const bool vec = (a.rag_w ? a.rag_vec_ok != 0 : a.W % 4 == 0) && !(variant & 2) && (reinterpret_cast<uintptr_t>(a.src1) % 16 == 0) &&
                   (a.src2 == nullptr || reinterpret_cast<uintptr_t>(a.src2) % 16 == 0);
  const bool pref = !(variant & 1);
  if constexpr (FC * FP == 8) {
    const bool pipe = ((variant & 4) != 0) != (SGMSE_CONV_PIPE_DEFAULT != 0);
    if (vec && pipe) { DRT_LAUNCH((conv_mfma_pipe_kernel<KS, WC, FC, FP>), grid, dim3(256), st, a); return; }
  }
  if constexpr (KS == 1 && FC * FP == 8) {
    // the element-wise staged 1x1 kernel on its widest tile needs more than 256 registers (it spilled): inputs that cannot take the
    // float4 path (width not a multiple of 4, unaligned sources) run on the 4-row tile -- the same bits, every tile shape accumulates
    // an output in the same order
    if (!vec) { launch_conv_mfma_t<KS, WC, FC, FP / 2>(a, st, variant, ksplit); return; }
    if (pref) DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 1, 1>), grid, dim3(256), st, a);
    else DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 0, 1>), grid, dim3(256), st, a);
  } else {
    if (vec && pref) DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 1, 1>), grid, dim3(256), st, a);
    else if (vec) DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 0, 1>), grid, dim3(256), st, a);
    else if (pref) DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 1, 0>), grid, dim3(256), st, a);
    else DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 0, 0>), grid, dim3(256), st, a);
  }
  if constexpr (FC * FP <= 2) {
    if (ksplit > 1) DRT_LAUNCH((conv_splitk_reduce_kernel<KS, WC, FC, FP>), dim3(grid.x, grid.y, 1), dim3(256), st, a, ksplit);
  }
}

// ksplit > 1 (small tiles only, ConvArgs::kchunk_stages chunks): split-K over gridDim.z + the reduce/epilogue kernel
inline void launch_conv_mfma(const ConvArgs& a, int ks, const ConvPlan& pl, drt::stream_t st, int variant = -1, int ksplit = 1) {
  const int v = variant < 0 ? conv_variant() : variant;
  if (ks == 1 && pl.co_t == 128 && (v & 8)) {
    // streaming kernel chunk: 32 channels, or 64 with bit 4 (one wave = 128 co x 32 px of one row; an 8-row-per-workgroup
    // shape with two pixel fragments per wave needs > 256 registers and spills)
    const int kc = (v & 16) ? 64 : 32;
    const int Cin = a.C1 + a.C2;
    if (Cin % kc == 0 && (a.src2 == nullptr || a.C1 % kc == 0)) {
      const int tiles = conv_grid_tiles(a, 4);
      dim3 grid(tiles, a.Cout / 128, 1);
      if (kc == 64) DRT_LAUNCH((conv1x1_stream_kernel<1, 64>), grid, dim3(256), st, a);
      else DRT_LAUNCH((conv1x1_stream_kernel<1, 32>), grid, dim3(256), st, a);
      return;
    }
  }
#define SGMSE_CONV_CASE(KS_, CO_, ROWS_, WC_, FC_, FP_) \
  if (ks == KS_ && pl.co_t == CO_ && pl.rows == ROWS_) { launch_conv_mfma_t<KS_, WC_, FC_, FP_>(a, st, variant, ksplit); return; }
  SGMSE_CONV_CASE(3, 128, 8, 2, 2, 4)
  SGMSE_CONV_CASE(3, 64, 8, 2, 1, 4)
  SGMSE_CONV_CASE(3, 32, 8, 1, 1, 2)
  SGMSE_CONV_CASE(3, 128, 4, 2, 2, 2)
  SGMSE_CONV_CASE(3, 64, 4, 2, 1, 2)
  SGMSE_CONV_CASE(3, 32, 4, 1, 1, 1)
  SGMSE_CONV_CASE(1, 128, 8, 2, 2, 4)
  SGMSE_CONV_CASE(1, 64, 8, 2, 1, 4)
  SGMSE_CONV_CASE(1, 32, 8, 1, 1, 2)
  SGMSE_CONV_CASE(1, 128, 4, 2, 2, 2)
  SGMSE_CONV_CASE(1, 64, 4, 2, 1, 2)
  SGMSE_CONV_CASE(1, 32, 4, 1, 1, 1)
#undef SGMSE_CONV_CASE
}

// split-operand convolutions (kernels_conv_split.h); a.w = weights packed by pack_weights_split_kernel
inline bool conv_split_eligible(int ks, int C1, int C2, int Cout) {
  return (ks == 3 || ks == 1) && Cout % 128 == 0 && (C1 + C2) % 16 == 0 && (C2 == 0 || C1 % 16 == 0) && (C1 + C2) <= 512;
}
// thin 3x3 layers (the C->4 pyramid convolutions): one zero-padded 32-channel fragment, the waves split the pixels
inline bool conv_thin_split_eligible(int ks, int C1, int C2, int Cout) {
  return ks == 3 && Cout <= 32 && (C1 + C2) >= 64 && (C1 + C2) % 16 == 0 && (C2 == 0 || C1 % 16 == 0) && (C1 + C2) <= 512;
}
// mode 1: bf16x3, mode 2: fp16x2 (a.acc_scale must point at the factor stored behind the packed weights)
// rows4: the 4-row workgroup shape of the full 3x3 kernel (same results; for launches that cannot fill the chip)
// abl: measurement-only ablation instantiations of the dominant shape (fp16x2, 8 rows, SiLU producer), see the kernel
template <class S, int SHAPE>
inline void launch_conv3x3_split_t(const ConvArgs& a, dim3 grid, drt::stream_t st) {
  const bool act = a.in_scale && a.in_act;
  if constexpr (SHAPE != 1 && S::SCALED) {
    if (a.sc_w) {                                  // folded residual shortcut (fp16x2, SiLU producer, full-block shapes)
      DRT_LAUNCH((conv3x3_split_kernel<S, SHAPE, 1, 0, 1>), grid, dim3(256), st, a);
      return;
    }
  }
  if (act) DRT_LAUNCH((conv3x3_split_kernel<S, SHAPE, 1>), grid, dim3(256), st, a);
  else DRT_LAUNCH((conv3x3_split_kernel<S, SHAPE, 0>), grid, dim3(256), st, a);
}
// rows per GroupNorm-statistics sub-tile the split kernels emit for this layer (ConvArgs::stats_rows): 4, except the thin shape
// (two rows per wave; its layers emit no statistics in the network)
inline int conv_split_stats_rows(int ks, int Cout) { return (ks == 3 && Cout <= 32) ? 1 : 4; }
inline void launch_conv_split(const ConvArgs& a, int ks, int mode, drt::stream_t st, bool rows4 = false, int abl = 0, int ksplit = 1) {
  const int tiles = conv_grid_tiles(a, 8);
  if (ks == 3 && a.Cout > 32 && rows4 && a.kchunk_stages > 0 && mode == 2) {     // chunked accumulation / split-K (coarse levels)
    const dim3 grid(conv_grid_tiles(a, 4), a.Cout / 128, ksplit);
    if (a.in_scale && a.in_act) DRT_LAUNCH((conv3x3_split_kernel<SplitH2, 2, 1, 0, 0, 1>), grid, dim3(256), st, a);
    else DRT_LAUNCH((conv3x3_split_kernel<SplitH2, 2, 0, 0, 0, 1>), grid, dim3(256), st, a);
    if (ksplit > 1) DRT_LAUNCH((conv_splitk_reduce_kernel<3, 4, 1, 4, true>), dim3(grid.x, grid.y, 1), dim3(256), st, a, ksplit);
    return;
  }
  if (ks == 3 && a.Cout > 32 && rows4) {
    const dim3 grid(conv_grid_tiles(a, 4), a.Cout / 128, 1);
    if (mode == 2) launch_conv3x3_split_t<SplitH2, 2>(a, grid, st); else launch_conv3x3_split_t<SplitB3, 2>(a, grid, st);
    return;
  }
  if (ks == 3 && a.Cout <= 32) {
    if (mode == 2) launch_conv3x3_split_t<SplitH2, 1>(a, dim3(tiles, 1, 1), st); else launch_conv3x3_split_t<SplitB3, 1>(a, dim3(tiles, 1, 1), st);
    return;
  }
  if (ks == 1) {
    if (mode == 2) DRT_LAUNCH(conv1x1_split_kernel<SplitH2>, dim3(tiles, a.Cout / 128, 1), dim3(256), st, a);
    else DRT_LAUNCH(conv1x1_split_kernel<SplitB3>, dim3(tiles, a.Cout / 128, 1), dim3(256), st, a);
    return;
  }
  const dim3 grid(tiles, a.Cout / 128, 1);
  if (abl && mode == 2) {
#define SGMSE_ABL_CASE(V) if (abl == V) { DRT_LAUNCH((conv3x3_split_kernel<SplitH2, 0, 1, V>), grid, dim3(256), st, a); return; }
    // the measurement series behind profiles/r02_split_*.txt and tools/power_probe.py's ablation cases: `make ABLATION=1` only
    // (each instantiation of this kernel costs ~6 s of build time; the product library carries none of them)
#ifdef SGMSE_ABLATION_FULL
    SGMSE_ABL_CASE(8) SGMSE_ABL_CASE(59) SGMSE_ABL_CASE(64)
    SGMSE_ABL_CASE(3) SGMSE_ABL_CASE(4) SGMSE_ABL_CASE(24) SGMSE_ABL_CASE(56)
    SGMSE_ABL_CASE(1) SGMSE_ABL_CASE(2) SGMSE_ABL_CASE(67) SGMSE_ABL_CASE(72) SGMSE_ABL_CASE(128) SGMSE_ABL_CASE(256) SGMSE_ABL_CASE(512)
#endif
#undef SGMSE_ABL_CASE
  }
  if (mode == 2) launch_conv3x3_split_t<SplitH2, 0>(a, grid, st); else launch_conv3x3_split_t<SplitB3, 0>(a, grid, st);
}

// Winograd F(2,3) x fp16x2 kernel of the wide levels (kernels_conv_wino.h); a.w = fragments packed by pack_weights_wino_kernel,
// a.co_scale = the per-channel factors behind them.  rows4: the 4-row shape (same bits) for launches that cannot fill the chip.
inline bool conv_wino_eligible(int C1, int C2, int Cout, int W) {
  return Cout % 128 == 0 && (C1 + C2) % 16 == 0 && (C2 == 0 || C1 % 16 == 0) && (C1 + C2) <= 512 && W % 2 == 0;
}
inline void launch_conv_wino(const ConvArgs& a, drt::stream_t st, bool rows4, bool trace = false, int abl = 0) {
  const bool act = a.in_scale && a.in_act;
#ifdef SGMSE_ABLATION_FULL
#define SGMSE_WABL_CASE(V) if (abl == V) { DRT_LAUNCH((conv3x3_wino_kernel<8, 1, 0, 0, V>), dim3(conv_grid_tiles(a, 8), a.Cout / 128, 1), dim3(512), st, a); return; }
  SGMSE_WABL_CASE(1) SGMSE_WABL_CASE(2) SGMSE_WABL_CASE(4) SGMSE_WABL_CASE(8) SGMSE_WABL_CASE(16) SGMSE_WABL_CASE(24) SGMSE_WABL_CASE(3) SGMSE_WABL_CASE(32) SGMSE_WABL_CASE(20)
#undef SGMSE_WABL_CASE
  if (trace) { DRT_LAUNCH((conv3x3_wino_kernel<8, 1, 0, 1>), dim3(conv_grid_tiles(a, 8), a.Cout / 128, 1), dim3(512), st, a); return; }
#endif
  (void)abl; (void)trace;
  if (rows4) {
    const dim3 grid(conv_grid_tiles(a, 4), a.Cout / 128, 1);
    if (a.sc_w) DRT_LAUNCH((conv3x3_wino_kernel<4, 1, 1>), grid, dim3(512), st, a);
    else if (act) DRT_LAUNCH((conv3x3_wino_kernel<4, 1, 0>), grid, dim3(512), st, a);
    else DRT_LAUNCH((conv3x3_wino_kernel<4, 0, 0>), grid, dim3(512), st, a);
  } else {
    const dim3 grid(conv_grid_tiles(a, 8), a.Cout / 128, 1);
    if (a.sc_w) DRT_LAUNCH((conv3x3_wino_kernel<8, 1, 1>), grid, dim3(512), st, a);
    else if (act) DRT_LAUNCH((conv3x3_wino_kernel<8, 1, 0>), grid, dim3(512), st, a);
    else DRT_LAUNCH((conv3x3_wino_kernel<8, 0, 0>), grid, dim3(512), st, a);
  }
}

// 2-D Winograd F(2x2,3x3) x fp16x2 (kernels_conv_wino2d.h; round 6: built, verified and measured against the 1-D kernel -- not the product
// path); a.w = fragments packed by pack_weights_wino2d_kernel.  No folded shortcut.  abl: measurement only.
inline void launch_conv_wino2d(const ConvArgs& a, drt::stream_t st, int abl = 0) {
  const dim3 grid(conv_grid_tiles(a, 4), a.Cout / 128, 1);
  const bool act = a.in_scale && a.in_act;
  if (abl == 8) { DRT_LAUNCH((conv3x3_wino2d_kernel<1, 8>), grid, dim3(512), st, a); return; }
  if (abl == 16) { DRT_LAUNCH((conv3x3_wino2d_kernel<1, 16>), grid, dim3(512), st, a); return; }
  if (act) DRT_LAUNCH((conv3x3_wino2d_kernel<1>), grid, dim3(512), st, a);
  else DRT_LAUNCH((conv3x3_wino2d_kernel<0>), grid, dim3(512), st, a);
}

// exact-fp32 VALU kernel of the C -> 4 pyramid convolutions (kernels_conv_thin.h); a.w = weights packed by pack_weights_thin_kernel
inline void launch_conv_thin(const ConvArgs& a, drt::stream_t st) {
  const dim3 grid(conv_thin_grid_tiles(a, (a.W + ConvThinGeom::TW - 1) / ConvThinGeom::TW), 1, 1);
  if (a.in_scale && a.in_act) DRT_LAUNCH((conv3x3_thin_kernel<1>), grid, dim3(256), st, a);
  else DRT_LAUNCH((conv3x3_thin_kernel<0>), grid, dim3(256), st, a);
}

inline void launch_conv_direct(const ConvArgs& a, int ks, drt::stream_t st) {
  const int HW = a.H * a.W;
  if (a.Cout <= 4) {
    dim3 grid((HW + 255) / 256, 1, a.B);
    if (ks == 3) DRT_LAUNCH((conv_direct_kernel<3, 4>), grid, dim3(256), st, a);
    else DRT_LAUNCH((conv_direct_kernel<1, 4>), grid, dim3(256), st, a);
  } else {
    dim3 grid((HW + 255) / 256, (a.Cout + 15) / 16, a.B);
    if (ks == 3) DRT_LAUNCH((conv_direct_kernel<3, 16>), grid, dim3(256), st, a);
    else DRT_LAUNCH((conv_direct_kernel<1, 16>), grid, dim3(256), st, a);
  }
}

}  // namespace sgmse

// 1x1 convolution (NIN / residual-shortcut layers; reference layerspp.py:241-243 `conv1x1`, layers.py:568-581 `NIN`)
// for 128-channel output blocks, as a streaming implicit GEMM.  MEASURED SLOWER than the LDS-tiled kernel and therefore
// not on the default path (SGMSE_CONV_VARIANT bit 3 selects it): 256->128 @ 8x256x512, 0.88 ms vs 0.79 ms plain,
// 1.02 vs 0.89 ms with the fused producer + residual (profiles/r01_conv1x1_streaming.txt).
//
// Idea: a 1x1 convolution has no spatial reuse, and with Cout = 128 every input element is worth only 256 FLOPs, so
// the layer sits near the HBM / MFMA balance point.  Here the B operand (pixels) never touches LDS: the MFMA B-fragment
// layout [k = lane>>5][px = lane&31] is a coalesced 128-byte row segment per half-wave, so each lane loads its own
// operand straight from global memory one K-chunk ahead of the MFMAs, and the fused GroupNorm-affine + SiLU producer
// is applied once per element in registers.  One wave owns all 128 output channels (FC = 4) of one 32-pixel row
// segment; two segments per wave (the 8-row tile) need > 256 registers and spill.  Only the weights (A operand,
// shared by the four waves) are staged through LDS, double-buffered per chunk.  Why it loses: with one pixel fragment
// per wave every MFMA needs its own A read (4 LDS reads per 4 MFMAs vs 6 per 8), and the layer is MFMA-bound at this
// shape (1.75 ms of MFMA time vs 1.37 ms of HBM time at B = 32), so bytes in flight were not the limiter.
//
// Same k order (channel pairs ascending) and the same v_mfma_f32_32x32x2_f32 chain as conv_mfma_kernel<1,...>, same
// epilogue: results are bit-identical to the other tile shapes.
#pragma once
#include "kernels_conv.h"

namespace sgmse {

template <int FP, int KC>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(ConvArgs p) {
  constexpr int FC = 4, NSTEP = KC / 2, CO_T = 128;
  using T = ConvTile<1, 1, FC, FP, 1>;
  static_assert(T::CO_T == CO_T && T::ROWS == 4 * FP, "tile");
  __shared__ float s_w0[KC * CO_T];
  __shared__ float s_w1[KC * CO_T];
  __shared__ float s_sc[512];
  __shared__ float s_sh[512];

  const int tid = threadIdx.x;
  const int Cin = p.C1 + p.C2;
  const int tiles_xg = (p.W + 31) >> 5;       // grid layout (ragged launches: of the widest utterance)
  const int tiles_y = (p.H + T::ROWS - 1) / T::ROWS;
  int b, ty, tx;
  conv_tile_of(p, (int)blockIdx.x, (int)gridDim.x, tiles_xg, tiles_y, b, ty, tx);
  if (p.rag_w) { if (!conv_ragged_adjust(p, b, tx)) return; }
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  const int co_blk = blockIdx.y;
  const bool xform = p.in_scale != nullptr;
  for (int c = tid; c < Cin; c += 256) {
    s_sc[c] = xform ? p.in_scale[b * Cin + c] : 1.f;
    s_sh[c] = xform ? p.in_shift[b * Cin + c] : 0.f;
  }
  const bool act = xform && p.in_act;

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const size_t HW = (size_t)H * W;
  // pixel offsets of this lane's FP fragment rows, clamped into the image (out-of-image lanes compute garbage that the
  // epilogue never stores: a 1x1 convolution does not mix pixels)
  int poff[FP];
#pragma unroll
  for (int j = 0; j < FP; ++j) {
    int y = ty * T::ROWS + wave * FP + j, x = tx * 32 + l31;
    y = y < H ? y : H - 1;
    x = x < W ? x : W - 1;
    poff[j] = y * W + x;
  }

  f32x16 acc[FC][FP];
#pragma unroll
  for (int i = 0; i < FC; ++i)
#pragma unroll
    for (int j = 0; j < FP; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* wbase = p.w + (size_t)co_blk * Cin * CO_T;
  constexpr int NW4 = KC * CO_T / 4 / 256;   // float4 weight items per thread and chunk
  float xb[NSTEP][FP], xn[NSTEP][FP];
  f32x4 rw[NW4];

  // channel plane of (chunk base c0, k-step s, this lane's k half) in the virtual concat [src1 | src2]; a chunk never
  // straddles the sources (C1 % KC == 0 is required by the launcher)
  auto load_x = [&](int c0, float (&dst)[NSTEP][FP]) {
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((size_t)b * p.C1 + c0 + kh) * HW : p.src2 + ((size_t)b * p.C2 + (c0 - p.C1) + kh) * HW;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
      for (int j = 0; j < FP; ++j) dst[s][j] = base[(size_t)(2 * s) * HW + poff[j]];
  };
  auto load_w = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NW4; ++i) rw[i] = *reinterpret_cast<const f32x4*>(wbase + (size_t)c0 * CO_T + 4 * (tid + 256 * i));
  };
  auto store_w = [&](float* sw) {
#pragma unroll
    for (int i = 0; i < NW4; ++i) *reinterpret_cast<f32x4*>(sw + 4 * (tid + 256 * i)) = rw[i];
  };
  auto compute = [&](const float* sw, int c0) {
    float a[2][FC];
    auto lda = [&](int s, int slot) {
#pragma unroll
      for (int i = 0; i < FC; ++i) a[slot][i] = sw[(2 * s + kh) * CO_T + i * 32 + l31];
    };
    lda(0, 0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + 1 < NSTEP) lda(s + 1, (s + 1) & 1);
      const float sc = s_sc[c0 + 2 * s + kh], sh = s_sh[c0 + 2 * s + kh];
      float bb[FP];
#pragma unroll
      for (int j = 0; j < FP; ++j) {
        float t = xb[s][j] * sc + sh;
        bb[j] = act ? silu_f(t) : t;
      }
#pragma unroll
      for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i], bb[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nchunk = Cin / KC;
  load_x(0, xb);
  load_w(0);
  store_w(s_w0);
  __syncthreads();   // s_w0, s_sc, s_sh visible
#pragma unroll 1
  for (int ci = 0; ci < nchunk; ci += 2) {
    // even chunk: weights in s_w0; prefetch the odd chunk's pixels and weights while the MFMAs run
    const int cn = (ci + 1 < nchunk ? ci + 1 : ci) * KC;     // clamped: the tail re-loads the last chunk (harmless)
    load_x(cn, xn);
    load_w(cn);
    compute(s_w0, ci * KC);
    store_w(s_w1);
    __syncthreads();
    if (ci + 1 >= nchunk) break;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
      for (int j = 0; j < FP; ++j) xb[s][j] = xn[s][j];
    const int cm = (ci + 2 < nchunk ? ci + 2 : ci + 1) * KC;
    load_x(cm, xn);
    load_w(cm);
    compute(s_w1, (ci + 1) * KC);
    store_w(s_w0);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
      for (int j = 0; j < FP; ++j) xb[s][j] = xn[s][j];
  }

  conv_epilogue<T, FC, FP, 1>(p, acc, b, co_blk, tx, ty, tiles_x, 0, wave, l31, kh);
}

}  // namespace sgmse

// The C -> 4 convolutions of the output pyramid (reference ncsnpp.py:366-379: conv3x3(act(GroupNorm(h))) accumulated over the levels)
// as an exact-fp32 VALU kernel.
//
// On the matrix pipe these layers compute a 32-channel MFMA fragment for 4 real output channels (conv3x3_split_kernel's thin shape:
// 0.93 PFLOP of issued f16 MFMA per evaluation at batch 32 for 0.04 algorithmic, 1.26 ms for the full-resolution launch) -- seven
// eighths of the matrix work is padding.  On the vector ALU there is no padding: 4 co x 9 taps = 36 FMAs per input value, 19.3 GFMA for the
// full-resolution launch at batch 32 against 131 TFMA/s of fp32 FMA on the chip.  The kernel is bound by those FMAs and by the LDS reads
// that feed them, so each thread computes 4 neighbouring pixels (one 16-byte + one 8-byte LDS read per tile row and channel feed
// 4 px x 3 taps) and all 4 output channels (16 accumulators); the weights are wave-uniform and come through the scalar cache.
//
//   workgroup = 256 threads = 16 rows x 64 columns of one utterance (thread: row t / 16, columns 4 (t % 16) .. + 3), 3 per CU;
//   the FMAs are packed over output-channel pairs (v_pk_fma_f32: weight pair from SGPRs, the input value broadcast to both halves);
//   K-stages of 4 input channels: the tile with halo (18 x 66, row stride 68 floats so that a thread's 16-byte read is aligned) after the
//   fused producer (GroupNorm affine + SiLU, zero padding applied behind it), double-buffered in LDS (2 x 19.6 KB: three workgroups per
//   CU); the raw values of stage s + 1 are loaded (coalesced dwords, 20 per thread) before the FMAs of stage s and pass the producer behind them;
//   accumulation order of an output: input channels ascending, taps row-major, one fmaf each -- a function of the layer only, never of
//   tile, batch or utterance length; epilogue: + bias, + residual (the up-sampled pyramid), x out_scale, 16-byte stores, range bound.
#pragma once
#include "kernels_conv.h"

namespace sgmse {

struct ConvThinGeom {
  static constexpr int TH = 16, TW = 64, KC = 4;
  static constexpr int LR = TH + 2, LC = 68;                    // LDS rows / row stride (66 used)
  static constexpr int PLANE = LR * LC;
  static constexpr int NPOS = LR * (TW + 2);                    // staged positions per channel
  static constexpr int NI = (NPOS + 255) / 256;                 // per thread and channel
};

inline bool conv_thin_eligible(int ks, int C1, int C2, int Cout) {
  return ks == 3 && Cout <= 4 && (C1 + C2) % ConvThinGeom::KC == 0 && (C2 == 0 || C1 % ConvThinGeom::KC == 0) && (C1 + C2) <= 512;
}
inline int conv_thin_grid_tiles(const ConvArgs& a, int tiles_x_widest) { return a.B * ((a.H + ConvThinGeom::TH - 1) / ConvThinGeom::TH) * tiles_x_widest; }

template <int ACT>
__global__ __launch_bounds__(256, 3) void conv3x3_thin_kernel(ConvArgs p) {
  using G = ConvThinGeom;
  __shared__ alignas(16) float s_in[2][G::KC * G::PLANE];
  __shared__ float s_sc[512];
  __shared__ float s_sh[512];
  const int tid = threadIdx.x;
  const int Cin = p.C1 + p.C2;
  const int tiles_xg = (p.W + G::TW - 1) / G::TW;               // grid layout (ragged launches: of the widest utterance)
  const int tiles_y = (p.H + G::TH - 1) / G::TH;
  int bid = (int)blockIdx.x;
  const int tx = bid % tiles_xg; bid /= tiles_xg;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  if (p.rag_w) { if (!conv_ragged_adjust(p, b, 2 * tx)) return; }      // (its width test is in 32-pixel columns)
  const int H = p.H, W = p.W;
  const unsigned HW = (unsigned)H * (unsigned)W;
  const int x0 = tx * G::TW, y0 = ty * G::TH;
  const bool xform = p.in_scale != nullptr;
  for (int c = tid; c < Cin; c += 256) {
    s_sc[c] = xform ? p.in_scale[b * Cin + c] : 1.f;
    s_sh[c] = xform ? p.in_shift[b * Cin + c] : 0.f;
  }
  // staging elements of this thread: NI positions (tile row r, tile column c; c fastest: a wave's loads of a row are consecutive) of each
  // of the stage's KC channels -- the channel is uniform per load, so the position's offsets are shared by the channels
  int it_goff[G::NI], it_loff[G::NI];
  unsigned ok_mask = 0, live_mask = 0;
#pragma unroll
  for (int i = 0; i < G::NI; ++i) {
    int e = tid + 256 * i;
    const bool live = e < G::NPOS;
    e = live ? e : G::NPOS - 1;
    const int r = e / (G::TW + 2), c = e - r * (G::TW + 2);
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    it_goff[i] = ok ? gy * W + gx : 0;
    it_loff[i] = r * G::LC + c;
    ok_mask |= (unsigned)ok << i; live_mask |= (unsigned)live << i;
  }
  float rin[G::KC][G::NI];
  auto load_stage = [&](int c0) {
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((size_t)b * p.C1 + c0) * HW : p.src2 + ((size_t)b * p.C2 + (c0 - p.C1)) * HW;
#pragma unroll
    for (int k = 0; k < G::KC; ++k)
#pragma unroll
      for (int i = 0; i < G::NI; ++i) rin[k][i] = base[(size_t)k * HW + it_goff[i]];
  };
  auto store_stage = [&](int c0, float* sbuf) {
#pragma unroll
    for (int k = 0; k < G::KC; ++k) {
      const float sc = s_sc[c0 + k], sh = s_sh[c0 + k];
#pragma unroll
      for (int i = 0; i < G::NI; ++i) {
        float t = rin[k][i] * sc + sh;
        if constexpr (ACT == 1) t = silu_f(t);
        if ((live_mask >> i) & 1) sbuf[k * G::PLANE + it_loff[i]] = ((ok_mask >> i) & 1) ? t : 0.f;   // zero padding applies to the producer's OUTPUT
      }
    }
  };
  const int row = tid >> 4, cx = (tid & 15) * 4;                   // this thread's 4 pixels: tile row `row`, tile columns cx .. cx + 3
  f32x2 acc[2][4];                                                  // [output-channel pair][pixel]: one v_pk_fma_f32 updates both channels
#pragma unroll
  for (int cp = 0; cp < 2; ++cp)
#pragma unroll
    for (int px = 0; px < 4; ++px) acc[cp][px] = f32x2{0.f, 0.f};

  const int nst = Cin / G::KC;
  load_stage(0);
  __syncthreads();                    // s_sc / s_sh visible
  store_stage(0, s_in[0]);
  __syncthreads();
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const float* cur = s_in[st & 1];
    if (st + 1 < nst) load_stage((st + 1) * G::KC);               // in flight during the FMAs below
#pragma unroll 1
    for (int k = 0; k < G::KC; ++k) {                            // (not unrolled: one channel's 36 weights in SGPRs at a time)
      const int ci = st * G::KC + k;
      // wave-uniform weights of input channel ci: [co][tap] (OIHW source: scalar loads)
      f32x2 wv[2][9];
#pragma unroll
      for (int cp = 0; cp < 2; ++cp)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          wv[cp][t] = f32x2{2 * cp < p.Cout ? p.w[((size_t)(2 * cp) * Cin + ci) * 9 + t] : 0.f, 2 * cp + 1 < p.Cout ? p.w[((size_t)(2 * cp + 1) * Cin + ci) * 9 + t] : 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const float* q = cur + k * G::PLANE + (row + dy) * G::LC + cx;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(q);
        const float2 v2 = *reinterpret_cast<const float2*>(q + 4);
        const float in[6] = {v4[0], v4[1], v4[2], v4[3], v2.x, v2.y};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int cp = 0; cp < 2; ++cp)
#pragma unroll
            for (int px = 0; px < 4; ++px) acc[cp][px] = drt_fma2(wv[cp][dy * 3 + dx], f32x2{in[px + dx], in[px + dx]}, acc[cp][px]);
      }
    }
    if (st + 1 < nst) store_stage((st + 1) * G::KC, s_in[(st + 1) & 1]);
    __syncthreads();
  }

  // epilogue: + bias, + residual, x out_scale; 16-byte accesses where the row allows them
  const int y = y0 + row, x = x0 + cx;
  float vmax = 0.f;
  if (y < H && x < W) {
    const bool vec = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.out) | (p.res ? reinterpret_cast<uintptr_t>(p.res) : 0)) & 15) == 0;
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      if (co < p.Cout) {
        const size_t o = ((size_t)(b * p.Cout + co) * H + y) * W + x;
        const float bv = p.bias ? p.bias[co] : 0.f;
        float v[4];
        if (vec) {                               // (x is a multiple of 4: the four pixels are inside the row together)
          f32x4 r = {0.f, 0.f, 0.f, 0.f};
          if (p.res) r = *reinterpret_cast<const f32x4*>(p.res + o);
#pragma unroll
          for (int px = 0; px < 4; ++px) v[px] = (acc[co >> 1][px][co & 1] + bv + r[px]) * p.out_scale;
          *reinterpret_cast<f32x4*>(p.out + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            v[px] = 0.f;
            if (x + px < W) {
              v[px] = (acc[co >> 1][px][co & 1] + bv + (p.res ? p.res[o + px] : 0.f)) * p.out_scale;
              p.out[o + px] = v[px];
            }
          }
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) vmax = fmaxf(vmax, fabsf(v[px]));
      }
    }
  }
  if (p.amax_out) {      // uniform
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((tid & 63) == 0) drt_atomic_max_nonneg(p.amax_out + b * kAmaxSpread + ((blockIdx.x * 4 + (tid >> 6)) & (kAmaxSpread - 1)), vmax);
  }
}

}  // namespace sgmse

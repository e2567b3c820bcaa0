// The C -> 4 convolutions of the output pyramid (reference ncsnpp.py:366-379: conv3x3(act(GroupNorm(h))) accumulated over the levels)
// as an exact-fp32 VALU kernel.
//
// On the matrix pipe these layers compute a 32-channel MFMA fragment for 4 real output channels (conv3x3_split_kernel's thin shape:
// 0.93 PFLOP of issued f16 MFMA per evaluation at batch 32 for 0.04 algorithmic, 1.26 ms for the full-resolution launch) -- seven
// eighths of the matrix work is padding.  On the vector ALU there is no padding: 4 co x 9 taps = 36 FMAs per input value, 19.3 GFMA for the
// full-resolution launch at batch 32 against 131 TFMA/s of fp32 FMA on the chip.  The kernel is bound by those FMAs and by the LDS reads
// that feed them, so each thread computes 4 neighbouring pixels (one 16-byte + one 8-byte LDS read per tile row and channel feed
// 4 px x 3 taps) and all 4 output channels (16 accumulators); the weights (packed [ci][tap][4 co]) are staged through LDS with the tile
// and read as wave-uniform broadcasts.  (Through the scalar cache instead -- s_load one step ahead -- every step waited on lgkmcnt(0),
// which scalar loads force: 8.4 k cycles per 4-channel stage with one wave per SIMD, profiles/r04_thin_kernel.txt.)
//
//   workgroup = 256 threads = 16 rows x 64 columns of one utterance (thread: row t / 16, columns 4 (t % 16) .. + 3), 3 per CU;
//   the FMAs of an output-channel pair sit side by side (weight pair x the input value).  ROUND 6: as TWO v_fma_f32, no longer one
//   v_pk_fma_f32 -- the packed instruction's low element came out wrong in lanes 48-63 whenever another process (or another queue of this
//   one) ran on the GPU at the same time, see the note in fmas() below; same-box A B B A at batch 32: 0.726 / 0.719 ms (256 x 512) and
//   0.218 / 0.222 ms (128 x 256) per launch against 0.676 / 0.688 and 0.193 / 0.196 with the packed form: +6 % / +13 % on these two launches,
//   0.065 ms of a 94 ms evaluation -- the price of results that do not depend on the device's other tenants;
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
inline size_t packed_thin_elems(int Cin) { return (size_t)Cin * 36; }
// OIHW -> [ci][tap][4 co] (output channels beyond Cout are zero)
__global__ void pack_weights_thin_kernel(const float* __restrict__ oihw, float* __restrict__ dst, int Cin, int Cout) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Cin * 36) return;
  const int co = e & 3, t = (e >> 2) % 9, ci = e / 36;
  dst[e] = co < Cout ? oihw[((size_t)co * Cin + ci) * 9 + t] : 0.f;
}
inline int conv_thin_grid_tiles(const ConvArgs& a, int tiles_x_widest) { return a.B * ((a.H + ConvThinGeom::TH - 1) / ConvThinGeom::TH) * tiles_x_widest; }

template <int ACT>
__global__ __launch_bounds__(256, 3) void conv3x3_thin_kernel(ConvArgs p) {
  using G = ConvThinGeom;
  __shared__ alignas(16) float s_in[2][G::KC * G::PLANE];
  __shared__ alignas(16) float s_w[2][G::KC * 36];
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
  unsigned ok_mask = 0;
#pragma unroll
  for (int i = 0; i < G::NI; ++i) {
    int e = tid + 256 * i;
    e = e < G::NPOS ? e : G::NPOS - 1;                            // (surplus lanes repeat the last position: same value to the same address)
    const int r = e / (G::TW + 2), c = e - r * (G::TW + 2);
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    it_goff[i] = ok ? gy * W + gx : 0;
    it_loff[i] = r * G::LC + c;
    ok_mask |= (unsigned)ok << i;
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
    int okm = (int)ok_mask;
    DRT_PIN_INT(okm);                 // (re-derived per stage: hoisted out of the K loop the 5 lane masks and their uses cost more SGPRs than the kernel has)
#pragma unroll
    for (int k = 0; k < G::KC; ++k) {
      const float sc = s_sc[c0 + k], sh = s_sh[c0 + k];
#pragma unroll
      for (int i = 0; i < G::NI; ++i) {
        float t = rin[k][i] * sc + sh;
        if constexpr (ACT == 1) t = silu_f(t);
        sbuf[k * G::PLANE + it_loff[i]] = ((okm >> i) & 1) ? t : 0.f;   // zero padding applies to the producer's OUTPUT
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
  // one step = one input channel x one kernel row: 3 weight quads (wave-uniform LDS reads: broadcasts) and the thread's 6 staged
  // values of that row, both requested one step ahead of the 24 packed FMAs that use them
  auto load_wq = [&](f32x4 (&w)[3], const float* wcur, int q) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) w[dx] = *reinterpret_cast<const f32x4*>(wcur + (q * 3 + dx) * 4);
  };
  auto load_in = [&](float (&in)[6], const float* cur, int q) {          // q = 3 k + dy within the stage
    const float* src = cur + (q / 3) * G::PLANE + (row + q % 3) * G::LC + cx;
    const f32x4 v4 = *reinterpret_cast<const f32x4*>(src);
    const f32x2 v2 = *reinterpret_cast<const f32x2*>(src + 4);
    in[0] = v4[0]; in[1] = v4[1]; in[2] = v4[2]; in[3] = v4[3]; in[4] = v2[0]; in[5] = v2[1];
  };
  auto fmas = [&](const f32x4 (&w)[3], const float (&in)[6]) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const f32x2 w01 = {w[dx][0], w[dx][1]}, w23 = {w[dx][2], w[dx][3]};
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const f32x2 v = {in[px + dx], in[px + dx]};
        // ROUND 6: two v_fma_f32 per channel pair, NOT one v_pk_fma_f32.  With the packed instruction this kernel's results changed
        // whenever ANOTHER PROCESS ran on the same GPU (tools/probes/thin_under_load_probe.py: 18-25 of 40 launches identical to the
        // solo result; always the low element of the pair -- output channels 0 and 2 -- in lanes 48-63 of a wave, |diff| up to 0.35),
        // i.e. across the wave save / restore of a shared device; alone on the device it was deterministic, which is why no test of
        // rounds 4-5 saw it.  The same arithmetic as scalar FMAs: 40 of 40 identical under the same load, same bits as before alone.
        acc[0][px] = f32x2{fmaf(w01[0], v[0], acc[0][px][0]), fmaf(w01[1], v[1], acc[0][px][1])};
        acc[1][px] = f32x2{fmaf(w23[0], v[0], acc[1][px][0]), fmaf(w23[1], v[1], acc[1][px][1])};
      }
    }
  };
  constexpr int NQ = 3 * G::KC;
  static_assert(NQ % 2 == 0 && NQ * 3 <= 64, "alternating operand sets; one wave stages a stage's weight quads");
  // the stage's 36 weight quads travel with the tile: global -> register (threads 0..35) -> LDS
  const f32x4* wq = reinterpret_cast<const f32x4*>(p.w);          // packed [ci][tap][4 co]
  f32x4 rw = {0.f, 0.f, 0.f, 0.f};
  f32x4 w[2][3];
  float in[2][6];
  load_stage(0);
  if (tid < NQ * 3) rw = wq[tid];
  __syncthreads();                    // s_sc / s_sh visible
  store_stage(0, s_in[0]);
  if (tid < NQ * 3) *reinterpret_cast<f32x4*>(&s_w[0][tid * 4]) = rw;
  __syncthreads();
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const float* cur = s_in[st & 1];
    const float* wcur = s_w[st & 1];
    const bool more = st + 1 < nst;
    load_in(in[0], cur, 0); load_wq(w[0], wcur, 0);
    if (more) {                                                    // in flight during the FMAs below
      load_stage((st + 1) * G::KC);
      if (tid < NQ * 3) rw = wq[(st + 1) * (NQ * 3) + tid];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q + 1 < NQ) { load_in(in[(q + 1) & 1], cur, q + 1); load_wq(w[(q + 1) & 1], wcur, q + 1); }
      DRT_SCHED_FENCE();              // requests first, then the FMAs that cover their latency ...
      fmas(w[q & 1], in[q & 1]);
      // ... and each step stays where it is written (left alone, the scheduler issues the LDS reads of all 12 steps up front: 157 spilled VGPRs)
      DRT_PIN8(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
      DRT_SCHED_FENCE();
    }
    if (more) {
      store_stage((st + 1) * G::KC, s_in[(st + 1) & 1]);
      if (tid < NQ * 3) *reinterpret_cast<f32x4*>(&s_w[(st + 1) & 1][tid * 4]) = rw;
    }
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

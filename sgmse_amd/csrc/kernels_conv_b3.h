// 3x3 convolution as an fp32-accurate implicit GEMM on the bf16 matrix pipe ("bf16x3": three-way operand split).
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate, and the fp32 3x3 kernel (kernels_conv_pipe.h) already
// sits at ~81 % of that fp32 peak.  Here every fp32 operand x is split EXACTLY into three bf16 terms
//     x = hi + mid + lo,   hi = x & 0xffff0000,  mid = (x - hi) & 0xffff0000,  lo = x - hi - mid   (8 + 8 + 8 mantissa bits)
// and a product a*b is accumulated in fp32 from the six bf16 x bf16 partial products whose weight is >= 2^-16
//     hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi
// (each exact in fp32).  The dropped terms (mid*lo, lo*mid, lo*lo) are below 2^-23 relative to a*b -- the size of one
// fp32 rounding -- so the result has fp32 accuracy (tests: same 1e-5 per-op / 2e-5 per-network gates as the fp32
// kernels, and a direct comparison against an fp64 convolution).  Six v_mfma_f32_32x32x16_bf16 (6 x 32 cycles for
// K = 16) replace eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles): 2.67x the fp32 MFMA peak.
//
// Layout (one workgroup = 128 co x 256 px = 8 rows x 32 columns, 4 waves, two workgroups per CU):
//   * wave w owns the 32 output channels w of the block and all 8 pixel fragments (acc = 8 x 16 registers);
//   * B operand (pixels): the input tile with halo (10 x 34 px) of a 16-channel K-stage lives in LDS already split,
//     channels fastest: [row][col][k-group of 8 channels][split][8 bf16] = 96 B per pixel (stride 112 B: bank
//     conflicts), so an MFMA B fragment of tap (dy,dx) is one ds_read_b128 per lane and the tap is an address
//     offset.  Two stages (76 KB) are double buffered; the fused producer (GroupNorm affine + SiLU) and the split run once per staged element;
//   * A operand (weights): packed offline in fragment order [co block][stage][tap][split][wave][lane][8 bf16], so a
//     fragment is one coalesced 16-byte global load per lane (L2-resident), prefetched one tap ahead.  No LDS.
//   * K order: stage (16 channels) -> tap -> {6 split products}.  Same epilogue as the fp32 kernels (conv_epilogue).
#pragma once
#include "kernels_conv.h"

namespace sgmse {

struct ConvB3 {
  static constexpr int KC = 16, ROWS = 8, TROWS = 10, TCOLS = 34;
  // dwords per staged pixel: 2 k-groups x 3 splits x 4 = 24, padded to 28 -- with a 24-dword stride the 16 lanes that a
  // ds_read_b128 services per LDS cycle collide pairwise on the 64 banks (lanes n and n+8: 48 % of all LDS cycles were
  // conflict cycles, profiles/r01_pmc_conv_b3.json); 28 n mod 64 is distinct for every lane of a group
  static constexpr int PX_U32 = 28, PX_V = PX_U32 / 4;
  static constexpr int STAGE_U32 = TROWS * TCOLS * PX_U32;     // 9520 dwords = 38,080 B
  static constexpr int NITEM = TROWS * TCOLS * 2;              // (pixel, k-group) staging items per stage
  static constexpr int NIT = (NITEM + 255) / 256;              // per thread (3)
};

// exact three-way bf16 split of an fp32 value (truncating; all three parts carry the sign of x)
__device__ __forceinline__ void b3_split(float x, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  const uint32_t xb = __builtin_bit_cast(uint32_t, x);
  hi = xb & 0xffff0000u;
  const float r1 = x - __builtin_bit_cast(float, hi);
  mid = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
  const float r2 = r1 - __builtin_bit_cast(float, mid);
  lo = __builtin_bit_cast(uint32_t, r2) & 0xffff0000u;
}

// Weight packing for conv3x3_b3_kernel.  src: OIHW fp32 [Cout][Cin][3][3]; dst: u32x4 [nCoBlk][Cin/16][9][3][4][64].
// One thread per (.., lane) 16-byte fragment element.
struct PackB3Args { const float* src; uint32_t* dst; int cin, cout; size_t total; };

__global__ __launch_bounds__(256) void pack_weights_b3_kernel(PackB3Args p) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.total) return;
  const int lane = (int)(e & 63);
  size_t r = e >> 6;
  const int w = (int)(r & 3); r >>= 2;
  const int split = (int)(r % 3); r /= 3;
  const int tap = (int)(r % 9); r /= 9;
  const int nst = p.cin / 16;
  const int st = (int)(r % nst);
  const int blk = (int)(r / nst);
  const int co = blk * 128 + w * 32 + (lane & 31);
  const int c0 = st * 16 + 8 * (lane >> 5);
  uint32_t out[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t part[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = c0 + 2 * q + h;
      const float v = co < p.cout ? p.src[((size_t)co * p.cin + c) * 9 + tap] : 0.f;
      uint32_t hi, mid, lo;
      b3_split(v, hi, mid, lo);
      part[h] = split == 0 ? hi : (split == 1 ? mid : lo);
    }
    out[q] = (part[0] >> 16) | part[1];
  }
  u32x4 o = {out[0], out[1], out[2], out[3]};
  reinterpret_cast<u32x4*>(p.dst)[e] = o;
}

inline size_t packed_b3_u32(int cin, int cout) { return (size_t)((cout + 127) / 128) * (cin / 16) * 9 * 3 * 4 * 64 * 4; }

__global__ __launch_bounds__(256, 2) void conv3x3_b3_kernel(ConvArgs p) {
  using C = ConvB3;
  using T = ConvTile<3, 4, 1, 8, 1>;       // epilogue geometry: 4 channel-waves x 1 fragment, 8 pixel fragments
  static_assert(T::CO_T == 128 && T::ROWS == 8, "tile");
  __shared__ u32x4 s_in0[C::STAGE_U32 / 4];
  __shared__ u32x4 s_in1[C::STAGE_U32 / 4];
  __shared__ float s_sc[512];
  __shared__ float s_sh[512];

  const int tid = threadIdx.x;
  const int Cin = p.C1 + p.C2;
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  const int tiles_y = (H + 7) >> 3;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int co_blk = blockIdx.y;
  const int x0 = tx * 32, y0 = ty * 8;
  const bool xform = p.in_scale != nullptr;
  for (int c = tid; c < Cin; c += 256) {
    s_sc[c] = xform ? p.in_scale[b * Cin + c] : 1.f;
    s_sh[c] = xform ? p.in_shift[b * Cin + c] : 0.f;
  }
  const float actf = (xform && p.in_act) ? 1.f : 0.f;   // arithmetic blend below: a branch would split the MFMA loop body
  const size_t HW = (size_t)H * W;

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;

  // staging items of this thread: (k-group g, tile row r, tile column c), c fastest so that a wave's loads of one
  // channel are consecutive pixels.  Items past the end repeat the last one (same address, same value).
  int it_goff[C::NIT], it_loff[C::NIT], it_g[C::NIT];
  unsigned okmask = 0;
#pragma unroll
  for (int i = 0; i < C::NIT; ++i) {
    int it = tid + 256 * i;
    it = it < C::NITEM ? it : C::NITEM - 1;
    const int g = it / (C::TROWS * C::TCOLS);
    const int rem = it - g * (C::TROWS * C::TCOLS);
    const int r = rem / C::TCOLS, c = rem - r * C::TCOLS;
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    okmask |= (ok ? 1u : 0u) << i;
    it_goff[i] = ok ? gy * W + gx : 0;
    it_loff[i] = (r * C::TCOLS + c) * C::PX_V + g * 3;  // in u32x4 units
    it_g[i] = g;
  }

  float rin[C::NIT][8];
  auto load_item = [&](int i, int c0) {    // raw global loads of one staged item of stage c0 (consumed >= 3 taps later)
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((size_t)b * p.C1 + c0) * HW : p.src2 + ((size_t)b * p.C2 + (c0 - p.C1)) * HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) rin[i][e] = base[(size_t)(8 * it_g[i] + e) * HW + it_goff[i]];
  };
  auto store_item = [&](int i, int c0, u32x4* sbuf) {   // producer + split + LDS write of one staged item
    uint32_t hi[8], mid[8], lo[8];
    const bool ok = (okmask >> i) & 1u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = c0 + 8 * it_g[i] + e;
      float t = rin[i][e] * s_sc[ch] + s_sh[ch];
      const float sig = __builtin_amdgcn_rcpf(1.0f + __expf(-t));
      t *= actf * (sig - 1.0f) + 1.0f;      // SiLU when actf = 1, identity when 0
      t = ok ? t : 0.f;                     // zero padding applies to the producer's OUTPUT
      b3_split(t, hi[e], mid[e], lo[e]);
    }
    u32x4 vh, vm, vl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vh[q] = (hi[2 * q] >> 16) | hi[2 * q + 1];
      vm[q] = (mid[2 * q] >> 16) | mid[2 * q + 1];
      vl[q] = (lo[2 * q] >> 16) | lo[2 * q + 1];
    }
    sbuf[it_loff[i]] = vh; sbuf[it_loff[i] + 1] = vm; sbuf[it_loff[i] + 2] = vl;
  };

  f32x16 acc[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  const int nst = Cin / C::KC;
  // A fragments of this wave: [co_blk][stage][tap][split][wave][lane]
  const u32x4* wbase = reinterpret_cast<const u32x4*>(p.w) + (size_t)co_blk * nst * 9 * 3 * 4 * 64 + wave * 64 + lane;
  auto load_a = [&](int st, int tap, u32x4 (&a)[3]) {
    const u32x4* q = wbase + ((size_t)st * 9 + tap) * 3 * 4 * 64;
    a[0] = q[0]; a[1] = q[4 * 64]; a[2] = q[2 * 4 * 64];
  };
  // B fragment base of this lane inside a stage buffer (u32x4 units): pixel (row j + dy, col l31 + dx), k-group kg
  const int b_lane = l31 * C::PX_V + kg * 3;

  // one tap of one stage: 8 pixel fragments x 6 split products.  The three B reads of fragment j+1 are issued before
  // the MFMAs of fragment j (order pinned with sched_barrier: left alone, the compiler sinks every LDS read to just in
  // front of its first use and waits lgkmcnt(0) three times per fragment).
  auto compute_tap = [&](const u32x4* sbuf, int tap, const u32x4 (&a)[3]) {
    const int dy = tap / 3, dx = tap - 3 * dy;
    const u32x4* sb = sbuf + b_lane + (dy * C::TCOLS + dx) * C::PX_V;
    u32x4 bq[2][3];
    bq[0][0] = sb[0]; bq[0][1] = sb[1]; bq[0][2] = sb[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j + 1 < 8) {
        const u32x4* q = sb + (j + 1) * C::TCOLS * C::PX_V;
        bq[(j + 1) & 1][0] = q[0]; bq[(j + 1) & 1][1] = q[1]; bq[(j + 1) & 1][2] = q[2];
      }
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 bh = bq[j & 1][0], bm = bq[j & 1][1], bl = bq[j & 1][2];
      f32x16 c = acc[0][j];
      c = mfma_32x32x16_bf16(a[2], bh, c);      // small terms first
      c = mfma_32x32x16_bf16(a[0], bl, c);
      c = mfma_32x32x16_bf16(a[1], bm, c);
      c = mfma_32x32x16_bf16(a[1], bh, c);
      c = mfma_32x32x16_bf16(a[0], bm, c);
      c = mfma_32x32x16_bf16(a[0], bh, c);
      acc[0][j] = c;
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: stage 0 -> s_in0
#pragma unroll
  for (int i = 0; i < C::NIT; ++i) load_item(i, 0);
  __syncthreads();          // s_sc / s_sh visible
#pragma unroll
  for (int i = 0; i < C::NIT; ++i) store_item(i, 0, s_in0);
  __syncthreads();

  u32x4 a0[3], a1[3];
  load_a(0, 0, a0);
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    // the stage after this one, clamped: the last stage re-stages itself into the buffer nobody reads again, which
    // keeps the loop body free of branches
    const int stn = st + 1 < nst ? st + 1 : st;
    const u32x4* cur = (st & 1) ? s_in1 : s_in0;
    u32x4* nxt = (st & 1) ? s_in0 : s_in1;
    // Per tap: the A fragments of the next tap first (vmcnt retires in order: they must be OLDER than the raw HBM loads
    // of the next stage issued behind them, or every tap would wait for HBM), then one item of raw loads (taps 0-2);
    // producer + split + LDS write of those items three or more taps later (taps 4, 6, 8).
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ntap = tap + 1 < 9 ? tap + 1 : 0;
      const int nstg = tap + 1 < 9 ? st : stn;
      if (tap & 1) load_a(nstg, ntap, a0); else load_a(nstg, ntap, a1);
      if (tap < C::NIT) load_item(tap, stn * C::KC);
      __builtin_amdgcn_sched_barrier(0);
      if (tap & 1) compute_tap(cur, tap, a1); else compute_tap(cur, tap, a0);
      // (slicing this producer work between the MFMAs with sched_group_barrier was tried: the compiler pairs the
      // elements into packed-f32 ops, clusters them behind the MFMAs anyway and spills)
      if (tap >= 4 && (tap & 1) == 0) store_item((tap - 4) / 2, stn * C::KC, nxt);
    }
    __syncthreads();
    // nine taps: the register sets have swapped roles (tap 8 computed from a0 and prefetched the next stage into a1)
#pragma unroll
    for (int s = 0; s < 3; ++s) a0[s] = a1[s];
  }

  conv_epilogue<T, 1, 8, 4>(p, acc, b, co_blk, tx, ty, tiles_x, wave, 0, l31, kg);
}

}  // namespace sgmse

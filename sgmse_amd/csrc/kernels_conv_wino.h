// 3x3 convolution of the wide U-Net levels: fp16x2 operand split (kernels_conv_split.h) + 1-D Winograd F(2,3) along the frame axis.
//
// Why: conv3x3_split_kernel<SplitH2> is ENERGY-bound -- the f16 matrix pipe on random data pins the socket at its 1400 W cap, the
// clock settles at ~1.5-1.7 GHz, and the MFMAs are ~60 % of the dynamic energy of a launch (profiles/r02_power_probe.txt,
// r02_split_ablation_microbench.txt).  Re-ordering instructions does not move it; removing MFMAs does.  F(2,3) along T computes two
// neighbouring output columns from 4 transformed inputs x 4 transformed weights instead of 2 x 3 products: 12 instead of 18 MFMA
// K-steps per output pair and input row tap, i.e. 2/3 of the matrix work, for the price of an input transform in the producer
// (two adds per staged element) and an output transform in the epilogue.
//
//     d0..d3 = producer output at columns 2m-1 .. 2m+2 (zero outside the image), g0..g2 = the weights of one kernel row
//     V0 = d0 - d2    V1 = d1 + d2    V2 = d2 - d1    V3 = d1 - d3                      (fp32, then split into hi + lo fp16)
//     U0 = g0         U1 = (g0 + g1 + g2) / 2        U2 = (g0 - g1 + g2) / 2        U3 = g2      (fp64 at pack time, then hi + lo)
//     M_k = sum over input channels and kernel rows of U_k V_k                            (4 implicit GEMMs, fp32 accumulation)
//     y(2m) = (M0 + M1) + M2                 y(2m+1) = M1 - (M2 + M3)
//
// Accuracy: every partial product is exact in fp32 as in the direct fp16x2 kernel; what Winograd adds is the rounding of V (one fp32
// add) and U (fp64 -> hi + lo, residual <= 2^-22 |U|) and the cancellation in the output transform -- measured error against an fp64
// convolution 1.3-2x the direct kernel's, inside the 1e-5 per-op gate by an order of magnitude (tests: check_conv_wino).  Range: |V|
// <= 2 max|d| and |U| <= 1.5 max|g|, both covered by the headroom of the power-of-two scales (bounds in [2^13, 2^14), fp16 max 65504).
//
// Layout (one workgroup = 128 co x ROWS x 32 px, 8 waves = 512 threads, ONE workgroup per CU):
//   * a GEMM column is a POSITION (image row, column pair); an MFMA B fragment = 32 positions = 2 rows x 16 pairs.  The accumulators
//     of a tile are 4 (k) x ROWS/2 fragments per 32-channel block -- twice the direct kernel's per output, which is why the tile is
//     held by 8 waves: wave (cf, kh) owns channel fragment cf and the transformed components k in {2 kh, 2 kh + 1} (128 accumulator
//     registers for ROWS = 8).  Splitting by k, not by position, keeps every A fragment private to ONE wave: weight-fragment traffic
//     from L2 per output is 12/9 of the direct kernel's instead of 2.7x.
//   * B operand: V of a 16-channel K-stage in LDS, [row][pair][k][k-group][split][8 x 16 bit], 17 x 16 B per (row, pair): a
//     fragment is one ds_read_b128 per lane, a tap (kernel row dy, component k) is an address offset, the 16 lanes an LDS cycle
//     serves hit 16 distinct bank quads.  Two stages double-buffered (2 x 43.5 KB).
//   * staging: a lane holds an ALIGNED column pair (x0 - 2 + 2j, +1), j = 0..17, of 4 channels (one 8-byte load per channel); 18
//     lanes = one (4-channel group, tile row), 3 such rows per wave pass.  The input transform needs the odd column of lane - 1 and
//     the even column of lane + 1: two whole-wave DPP shifts per channel (wave_shr:1 / wave_shl:1), no LDS round trip and no
//     producer evaluated twice.
//   * A operand: [co block][stage][dy][k][split][cf][lane] 16-byte fragments from global memory (L2), ring of three, two taps ahead.
//   * epilogue: A-waves (kh = 0) write {M0 + M1, M1}, B-waves then add {M2, -(M2 + M3)} in LDS ([co][row][col] fp32, over the dead
//     stage buffers), every wave reads back 32 co x ROWS/2... rows in the shared epilogue's fragment layout (conv_epilogue: residual,
//     1/sqrt 2, GroupNorm partials per 4 rows, range bound).  The K order and the transform order are fixed, so the 8-row and the
//     4-row shape give the same bits (the 4-row shape is for launches that cannot fill the chip).
//   * weights carry a power-of-two scale PER OUTPUT CHANNEL (undone on the way out of the exchange buffer): one outlier weight
//     costs its own channel's small weights their low bits, not the whole layer's.
#pragma once
#include "kernels_conv_split.h"

#ifndef SGMSE_WINO_TAIL_STAGING
#define SGMSE_WINO_TAIL_STAGING 0
#endif

namespace sgmse {

template <int ROWS_>
struct WinoGeom {
  static constexpr int KC = 16, ROWS = ROWS_, TROWS = ROWS_ + 2, NF = ROWS_ / 2;
  static constexpr int PV = 17;                       // u32x4 per (row, pair): 4 k x 2 k-groups x 2 splits, + 1 (bank spread)
  static constexpr int ROW_V = 16 * PV;
  static constexpr int STAGE_V = TROWS * ROW_V;
  static constexpr int NSROW = 4 * TROWS;             // staging rows: (4-channel group q, tile row r), q fastest
  static constexpr int NPASS = (NSROW + 2) / 3;       // wave passes of 3 staging rows x 18 lanes
  static constexpr int NIT = (NPASS + 7) / 8;         // passes per wave
  static constexpr int CO_V = 512;                    // producer coefficient table, u32x4 per input channel
  static constexpr int NDIR = ROWS_ == 4 ? 1 : 2;     // epilogue exchange: B-waves -> A-waves only (4 rows), or both ways
  static constexpr int XCH_V = 4 * NDIR * 2 * 16 * 64 / 2;   // ... [channel fragment][direction][fragment slot][register][lane] float2, in u32x4
  static constexpr int LDS_V = (2 * STAGE_V + CO_V) > XCH_V ? (2 * STAGE_V + CO_V) : XCH_V;
};
template <int ROWS_>
struct WinoTile { static constexpr int CO_T = 128, ROWS = ROWS_; };

// Weight packing.  src: OIHW fp32 [Cout][Cin][3][3]; dst: u32x4 [nCoBlk][Cin/16][dy 3][k 4][split 2][cf 4][lane 64], followed by
// nCoBlk * 128 floats: per output channel the factor 2^-e that undoes its weights' scale.  One thread per 16-byte fragment element.
struct PackWinoArgs { const float* src; uint32_t* dst; int cin, cout; size_t total; };
inline size_t packed_wino_frags(int cin, int cout) { return (size_t)((cout + 127) / 128) * (cin / 16) * 12 * 2 * 4 * 64; }
inline size_t packed_wino_bytes(int cin, int cout) { return packed_wino_frags(cin, cout) * 16 + (size_t)((cout + 127) / 128) * 128 * 4; }

// max |transformed weight| of output channel co (<= 1.5 max |g|): the channel's power-of-two scale
__device__ __forceinline__ float wino_co_absmax(const float* src, int cin, int co) {
  float m = 0.f;
  for (int i = 0; i < cin * 3; ++i) {
    const float* g = src + ((size_t)co * cin * 3 + i) * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(g0), fabsf(g2)), fmaxf(fabsf(0.5f * (g0 + g1 + g2)), fabsf(0.5f * (g0 - g1 + g2)))));
  }
  return m;
}
__global__ __launch_bounds__(256) void wino_co_scale_kernel(const float* src, int cin, int cout, int cout_pad, float* inv_scale, float* scale) {
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (co >= cout_pad) return;
  const float s = co < cout ? h2_weight_scale(wino_co_absmax(src, cin, co)) : 1.f;
  scale[co] = s; inv_scale[co] = 1.f / s;
}
__global__ __launch_bounds__(256) void pack_weights_wino_kernel(PackWinoArgs p, const float* co_scale) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.total) return;
  const int lane = (int)(e & 63);
  size_t r = e >> 6;
  const int cf = (int)(r & 3); r >>= 2;
  const int split = (int)(r & 1); r >>= 1;
  const int k = (int)(r & 3); r >>= 2;
  const int dy = (int)(r % 3); r /= 3;
  const int nst = p.cin / 16;
  const int st = (int)(r % nst);
  const int blk = (int)(r / nst);
  const int co = blk * 128 + cf * 32 + (lane & 31);
  const int c0 = st * 16 + 8 * (lane >> 5);
  const double ws = co < p.cout ? (double)co_scale[co] : 1.0;
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t part[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = c0 + 2 * q + h;
      double u = 0.0;
      if (co < p.cout) {
        const float* g = p.src + (((size_t)co * p.cin + c) * 3 + dy) * 3;
        const double g0 = g[0], g1 = g[1], g2 = g[2];
        u = k == 0 ? g0 : k == 1 ? 0.5 * (g0 + g1 + g2) : k == 2 ? 0.5 * (g0 - g1 + g2) : g2;
      }
      u *= ws;
      // hi = RN_f16(u) through fp32 (a double rounding here only moves a sliver between hi and lo), lo = RN_f16(u - hi) from the exact residual
      const uint32_t hi = drt_f32_to_f16((float)u);
      const uint32_t lo = drt_f32_to_f16((float)(u - (double)drt_f16_to_f32(hi)));
      part[h] = split ? lo : hi;
    }
    o[q] = part[0] | (part[1] << 16);
  }
  reinterpret_cast<u32x4*>(p.dst)[e] = o;
}

// ACT: SiLU behind the GroupNorm affine of the fused producer (1) or the affine only (0).
// SC = 1: with the folded 1x1 residual shortcut (ConvArgs::sc_*, as conv3x3_split_kernel): its K-stages run first into M0 (even
// columns) and M3 (odd columns, negated), the accumulators are rescaled to the 3x3 stages' operand scaling and the stages continue.
// TRACE (measurement only, sgmse_bench_conv): phase time stamps per workgroup (ConvArgs::trace; tools/analyze_trace.py).
// ABL (measurement only, `make ABLATION=1`; results WRONG on purpose): 1 producer without the transcendental pair, 2 input transform without the
// cross-lane shifts, 4 no LDS writes of the staged tile, 8 no staging at all behind the prologue, 16 no raw loads in the K loop, 32 the split arithmetic without its LDS stores.
template <int ROWS, int ACT, int SC, int TRACE = 0, int ABL = 0>
__global__ __launch_bounds__(512, 1) void conv3x3_wino_kernel(ConvArgs p) {
  using G = WinoGeom<ROWS>;
  using T = WinoTile<ROWS>;
  using S = SplitH2;
  constexpr int NF = G::NF, NIT = G::NIT, PV = G::PV, NS = 2;
  __shared__ u32x4 s_all[G::LDS_V];
  u32x4* const s_in0 = s_all;
  u32x4* const s_in1 = s_all + G::STAGE_V;
  f32x4* const s_co = reinterpret_cast<f32x4*>(s_all + 2 * G::STAGE_V);

  const int tid = threadIdx.x;
  const int wave = drt_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
  const int cf = wave & 3, kh = wave >> 2;
  unsigned long long* trace = nullptr;
  if constexpr (TRACE) {
    if (tid == 0 && p.trace) {
      trace = p.trace + 32 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
      trace[0] = (unsigned long long)drt_hw_id() | ((unsigned long long)drt_xcc_id() << 32);
      trace[1] = drt_clock();
    }
  }
  // (One workgroup owns a CU and every workgroup of a launch takes the same time, so all 256 CUs run their K loops and then their
  //  epilogues in lock step.  De-phasing the first residency round by up to a tile time did not shorten the epilogue -- it is bound by
  //  the CU's own vector-memory issue rate, ~25 cycles per wave instruction, not by the simultaneous burst -- and cost its start-up
  //  delay: profiles/r04_wino_microbench_history.txt visit r04b, r04_wino_trace.txt; removed.  Static issue priority for one half of the
  //  workgroup (s_setprio 1 for the B-waves, or for the A-waves, before the K loop) measured within +-1 % on three shapes: same file, visit
  //  r04p.  16-byte output stores (neighbouring lanes trade column pairs by DPP so that each stores 4 columns of one channel: half the
  //  store instructions) measured 0.930 / 1.502 / 0.351 ms against 0.922 / 1.491 / 0.352: no gain, visit r04w -- and the B-waves' queued
  //  x4 stores then picked up first dwords that later VALU instructions had already overwritten (non-repeatable; 8-byte stores do not).)
  const int Cin = p.C1 + p.C2;
  const int tiles_xg = (p.W + 31) >> 5;
  const int tiles_y = (p.H + ROWS - 1) / ROWS;
  int b, ty, tx;
  conv_tile_of(p, (int)blockIdx.x, (int)gridDim.x, tiles_xg, tiles_y, b, ty, tx);
  if (p.rag_w) { if (!conv_ragged_adjust(p, b, tx)) return; }
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  const int co_blk = blockIdx.y;
  const int x0 = tx * 32, y0 = ty * ROWS;
  const unsigned HW = (unsigned)H * (unsigned)W;

  // prologue loads, all issued before the first is waited for: range bound, producer coefficients of channel tid, per-channel
  // weight scales and bias terms of this lane's 16 accumulator rows, the first stage's raw inputs, the first weight fragments
  float xb_raw = 0.f;
  if (p.xbound) xb_raw = p.xbound[b * kAmaxSpread + (tid & (kAmaxSpread - 1))];
  const bool cld = p.in_scale != nullptr && tid < Cin;
  const float csc = cld ? p.in_scale[b * Cin + (tid < Cin ? tid : 0)] : 1.f;
  const float csh = cld ? p.in_shift[b * Cin + (tid < Cin ? tid : 0)] : 0.f;

  // staging items of this thread (wave pass wave + 8 i): staging row (q, r) = 3 pass + lane / 18, aligned column pair j = lane % 18
  unsigned it_boff[NIT];
  int it_woff[NIT], it_q[NIT];
  bool it_ok[NIT], it_wr[NIT], it_run[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int pass = wave + 8 * i;
    const int sub = lane / 18, j = lane - 18 * sub;
    const int rr = 3 * pass + sub;
    const bool live = sub < 3 && rr < G::NSROW;
    const int q = rr & 3, r = live ? rr >> 2 : 0;
    const int gy = y0 - 1 + r, gx = x0 - 2 + 2 * j;
    const bool ok = live && gy >= 0 && gy < H && gx >= 0 && gx < W;        // (W is even: both columns of an aligned pair are in or out)
    it_boff[i] = ((unsigned)(4 * q) * HW + (ok ? (unsigned)(gy * W + gx) : 0u)) * 4u;
    it_woff[i] = ((r * 16 + (j - 1)) * PV + (q >> 1) * NS) * 2 + (q & 1);   // in 8-byte units: entry (r, pair j - 1, k-group q / 2), half q % 2
    it_q[i] = q; it_ok[i] = ok; it_wr[i] = live && j >= 1 && j <= 16;
    it_run[i] = 3 * pass < G::NSROW;                                        // wave-uniform: this pass holds at least one staging row
  }
  float rin[NIT][8];
  auto load_item = [&](int i, int c0) {
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((size_t)b * p.C1 + c0) * HW : p.src2 + ((size_t)b * p.C2 + (c0 - p.C1)) * HW;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base + (size_t)c * HW) + it_boff[i]);
      rin[i][2 * c] = v.x; rin[i][2 * c + 1] = v.y;
    }
  };
  // input transform of one channel of an item: V[k] from this lane's pair (e, o), the odd column of lane - 1 and the even column of lane + 1
  float V[4][4];
  // producer of one element from a prefetched coefficient vector {s kx, h kx, s (-log2 e), h (-log2 e)}
  auto produce_co = [&](float x, const f32x4& co, bool ok) -> float {
    float o = x * co[0] + co[1];
    if constexpr (ACT == 1) {
      const float u = x * co[2] + co[3];
      if constexpr (ABL & 1) o = o * u; else o = o * __builtin_amdgcn_rcpf(1.0f + drt_exp2(u));
    }
    return ok ? o : 0.f;
  };
  auto stage_chan_co = [&](int i, int c, const f32x4& co) {
    const float e = produce_co(rin[i][2 * c], co, it_ok[i]), o = produce_co(rin[i][2 * c + 1], co, it_ok[i]);
    const float ol = (ABL & 2) ? e : drt_wave_shr1(o), er = (ABL & 2) ? o : drt_wave_shl1(e);
    V[0][c] = ol - o; V[1][c] = e + o; V[2][c] = o - e; V[3][c] = e - er;
  };
  auto stage_chan = [&](int i, int c, int c0) { stage_chan_co(i, c, s_co[c0 + 4 * it_q[i] + c]); };
  // split + LDS write of component k of an item (4 channels: two packed pairs per split term)
  auto flush_k = [&](int i, int k, u32x4* sbuf) {
    if (it_wr[i] && (!(ABL & 4) || V[k][0] == 12345.678f)) {
      uint2* w = reinterpret_cast<uint2*>(sbuf) + it_woff[i];
      uint32_t d01[2], d23[2];
      S::split2(V[k][0], V[k][1], d01);
      S::split2(V[k][2], V[k][3], d23);
      if constexpr (ABL & 32) {               // the split arithmetic without the stores
        if ((d01[0] ^ d23[1]) == 0x12345678u) w[k * 8] = make_uint2(d01[0] ^ d01[1], d23[0] ^ d23[1]);
      } else {
        w[k * 8] = make_uint2(d01[0], d23[0]);          // (k: 4 u32x4 = 8 halves; split: 1 u32x4 = 2 halves)
        w[k * 8 + 2] = make_uint2(d01[1], d23[1]);
      }
    }
  };
  auto flush_item = [&](int i, u32x4* sbuf) {
#pragma unroll
    for (int k = 0; k < 4; ++k) flush_k(i, k, sbuf);
  };

  const int nst = Cin / G::KC;
  // per-channel weight scales (undone behind the exchange buffer) and additive terms of this lane's accumulator rows
  const float* cs_tab = p.co_scale + (size_t)co_blk * 128;
  float acc_raw[16];
  if constexpr (!SC) conv_acc_raw<T>(p, b, co_blk, cf, kg, acc_raw);      // (SC: loaded behind the shortcut stages, where they are used)
  auto load_cs = [&](float (&cs)[16]) {      // (loaded where they are used: 16 registers that must not live through the K loop)
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[r] = cs_tab[cf * 32 + 4 * kg + (r & 3) + 8 * (r >> 2)];
  };
  float sc_m1 = 0.f, sc_m2 = 0.f;
  if constexpr (SC) {
    sc_m1 = p.sc_amax1[b * kAmaxSpread + (tid & (kAmaxSpread - 1))];
    if (p.sc_amax2) sc_m2 = p.sc_amax2[b * kAmaxSpread + (tid & (kAmaxSpread - 1))];
  } else {
#pragma unroll
    for (int i = 0; i < NIT; ++i) if (it_run[i]) load_item(i, 0);
  }
  float kx = 1.f;
  if (p.xbound) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) xb_raw = fmaxf(xb_raw, __shfl_xor(xb_raw, o));
    kx = h2_weight_scale(xb_raw);
  }
  const float inv_kx = 1.f / kx;
  if (tid < Cin) {
    constexpr float nl2e = -1.4426950408889634f;
    f32x4 v;
    v[0] = csc * kx; v[1] = csh * kx; v[2] = csc * nl2e; v[3] = csh * nl2e;
    s_co[tid] = v;
  }
  // accumulators: [component kk of this wave: k = 2 kh + kk][position fragment].  y(2m) = (M0 + M1) + M2 and y(2m+1) = M1 - (M2 + M3):
  // the additive terms (bias + time-embedding row, in accumulator units) start in M0 and, negated, in M3
  f32x16 acc[2][NF];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    float init[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) init[r] = 0.f;
    if constexpr (!SC) {
      if (kk == kh) {
        float cs_inv[16];
        load_cs(cs_inv);
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float t = acc_raw[r] * (kx / cs_inv[r]); init[r] = kh ? -t : t; }
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kk][f][r] = init[r];
  }

  // A fragments of this wave: tap (dy, kk) of stage st -> [st][dy][2 kh + kk][split][cf][lane]
  const u32x4* wblk = reinterpret_cast<const u32x4*>(p.w) + (size_t)co_blk * nst * 12 * NS * 4 * 64;
  const unsigned a_boff = (unsigned)((cf * 64 + lane) * 16);
  auto load_a = [&](int st, int tap, u32x4 (&a)[NS]) {
    const int dy = tap >> 1, k = 2 * kh + (tap & 1);
    const u32x4* q = wblk + (size_t)((st * 3 + dy) * 4 + k) * NS * 4 * 64;
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(q + s * 4 * 64) + a_boff);
  };
  // B fragment base of this lane: position (row 2 f + (l31 >> 4) [+ dy], pair l31 & 15), k-group kg
  const int b_lane = ((l31 >> 4) * 16 + (l31 & 15)) * PV + kg * NS;

  // one tap = component k of kernel row dy: NF position fragments x 3 split products; behind each fragment's MFMAs a share of the
  // staging work of `item` (the next stage's tile): the four channel transforms spread over the fragments, the LDS writes last
  u32x4 bq[2][NS];
  auto compute_tap = [&](const u32x4* sbuf, int tap, int kq, const u32x4 (&a)[NS], int item, bool stage_here, int c0n, u32x4* nxt) {
    const int dy = tap >> 1, kk = tap & 1;
    const u32x4* sb = sbuf + b_lane + dy * G::ROW_V + kq * 4;
    // the item's four coefficient vectors first: a read issued right in front of its use waits out the whole LDS queue (eight waves'
    // operand reads) -- four exposed round trips per item were ~60 % of a staging tap (profiles/r04_wino_trace.txt)
    f32x4 co4[4];
    if (item >= 0 && stage_here) {
#pragma unroll
      for (int c = 0; c < 4; ++c) co4[c] = s_co[c0n + 4 * it_q[item] + c];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) bq[0][s] = sb[s];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (f + 1 < NF) {
        const u32x4* q = sb + (f + 1) * 2 * G::ROW_V;
#pragma unroll
        for (int s = 0; s < NS; ++s) bq[(f + 1) & 1][s] = q[s];
      }
      __builtin_amdgcn_sched_barrier(SGMSE_SPLIT_FENCE);
#pragma unroll
      for (int k = 0; k < S::NP; ++k) acc[kk][f] = S::mfma(a[S::pa(k)], bq[f & 1][S::pb(k)], acc[kk][f]);
      if (item >= 0 && stage_here) {
        constexpr int CPF = 4 / NF;                   // channels per fragment slot (1 for 8 rows, 2 for 4)
#pragma unroll
        for (int c = 0; c < CPF; ++c) stage_chan_co(item, f * CPF + c, co4[f * CPF + c]);
        if (f == NF - 1) flush_item(item, nxt);
      }
      __builtin_amdgcn_sched_barrier(SGMSE_SPLIT_FENCE);
    }
  };

  if constexpr (SC) {
    // ---- folded 1x1 shortcut: K-stages of 16 raw channels; even columns -> M0 (A-waves), negated odd columns -> M3 (B-waves) ----
    float m = fmaxf(sc_m1, sc_m2);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float xs = h2_weight_scale(m);
    const int nsts = (p.sc_C1 + p.sc_C2) / G::KC;
    // one item per thread and stage: 4-channel group q, interior row r, column pair jm (ROWS = 4: half of the threads)
    const int sq = tid & 3, sjm = (tid >> 2) & 15, sr = tid >> 6;
    const bool s_live = sr < ROWS;
    const int sgy = y0 + sr, sgx = x0 + 2 * sjm;
    const bool s_ok = s_live && sgy < H && sgx < W;
    const unsigned s_boff = ((unsigned)(4 * sq) * HW + (s_ok ? (unsigned)(sgy * W + sgx) : 0u)) * 4u;
    const int s_woff = (((sr + 1) * 16 + sjm) * PV + (sq >> 1) * NS) * 2 + (sq & 1);      // tile row sr + 1 (the centre tap reads dy = 1)
    // Two 16-channel stages per barrier ("super-stage"): the centre tap uses one component slot per wave group, so the first stage of
    // a pair goes to the slots k = 0 (even columns, A-waves) / k = 3 (negated odd columns, B-waves) of the tile buffer and the second
    // to k = 1 / k = 2 -- 24 MFMAs per wave between barriers instead of 12.  Raw inputs and
    // weight fragments one super-stage ahead (a fragment loaded right in front of its MFMAs exposed an L2 round trip per stage; two
    // register sets for two super-stages of cover spilled).
    const int nss = (nsts + 1) / 2;
    float rsa[16];
    auto load_sc = [&](int ss, float (&dst)[16]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int st = 2 * ss + h < nsts ? 2 * ss + h : nsts - 1;      // (odd stage count: the last half is loaded twice, stored and used once)
        const int c0 = st * G::KC;
        const bool first = c0 < p.sc_C1;
        const float* base = first ? p.sc_src1 + ((size_t)b * p.sc_C1 + c0) * HW : p.sc_src2 + ((size_t)b * p.sc_C2 + (c0 - p.sc_C1)) * HW;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base + (size_t)c * HW) + s_boff);
          dst[8 * h + 2 * c] = v.x; dst[8 * h + 2 * c + 1] = v.y;
        }
      }
    };
    auto store_sc = [&](const float (&src)[16], u32x4* sbuf, int ss) {
      if (!s_live) return;
      uint2* w = reinterpret_cast<uint2*>(sbuf) + s_woff;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && 2 * ss + 1 >= nsts) break;
        float ev[4], od[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ev[c] = s_ok ? src[8 * h + 2 * c] * xs : 0.f;
          od[c] = s_ok ? -(src[8 * h + 2 * c + 1] * xs) : 0.f;
        }
        uint32_t d01[2], d23[2];
        const int ke = h, ko = 3 - h;                 // component slots of the even / odd columns of this half
        S::split2(ev[0], ev[1], d01); S::split2(ev[2], ev[3], d23);
        w[ke * 8] = make_uint2(d01[0], d23[0]); w[ke * 8 + 2] = make_uint2(d01[1], d23[1]);
        S::split2(od[0], od[1], d01); S::split2(od[2], od[3], d23);
        w[ko * 8] = make_uint2(d01[0], d23[0]); w[ko * 8 + 2] = make_uint2(d01[1], d23[1]);
      }
    };
    const u32x4* wsc = reinterpret_cast<const u32x4*>(p.sc_w) + (size_t)co_blk * nsts * NS * 4 * 64;
    auto load_asc = [&](int ss, u32x4 (&a)[2][NS]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int st = 2 * ss + h < nsts ? 2 * ss + h : nsts - 1;
        const u32x4* q = wsc + (size_t)st * NS * 4 * 64;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) a[h][s2] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(q + s2 * 4 * 64) + a_boff);
      }
    };
    // kernel row 1 (taps 2, 3); A-waves: accumulator set 0 (M0), slots 0 then 1; B-waves: set 1 (M3), slots 3 then 2
    auto compute_sc = [&](const u32x4* sbuf, const u32x4 (&a)[2][NS], int ss) {
      const bool two = 2 * ss + 1 < nsts;
      if (kh == 0) { compute_tap(sbuf, 2, 0, a[0], -1, false, 0, nullptr); if (two) compute_tap(sbuf, 2, 1, a[1], -1, false, 0, nullptr); }
      else { compute_tap(sbuf, 3, 3, a[0], -1, false, 0, nullptr); if (two) compute_tap(sbuf, 3, 2, a[1], -1, false, 0, nullptr); }
    };
    u32x4 asc[2][NS];
    load_sc(0, rsa);
    load_asc(0, asc);
    store_sc(rsa, s_in0, 0);
    if (nss > 1) load_sc(1, rsa);
    __syncthreads();
    // super-stage ss: MFMAs from buffer ss & 1, then -- behind them -- the weight fragments of ss + 1, the split of ss + 1 (its raw
    // inputs are in registers) into the other buffer, and the raw loads of ss + 2 into the registers that just became free
#pragma unroll 1
    for (int ss = 0; ss < nss; ++ss) {
      const u32x4* cur = (ss & 1) ? s_in1 : s_in0;
      u32x4* nxt = (ss & 1) ? s_in0 : s_in1;
      compute_sc(cur, asc, ss);
      if (ss + 1 < nss) {
        load_asc(ss + 1, asc);
        store_sc(rsa, nxt, ss + 1);
        if (ss + 2 < nss) load_sc(ss + 2, rsa);
      }
      __syncthreads();
    }
    // from the shortcut's operand scaling (weights 2^k1 per layer, input xs) to the 3x3 stages' (weights per channel, input kx), plus
    // the additive terms; M3 holds the NEGATED shortcut, so its terms are negated too
    const float sgn = kh ? -1.f : 1.f;
    float cs_inv[16];
    load_cs(cs_inv);
    conv_acc_raw<T>(p, b, co_blk, cf, kg, acc_raw);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float as3 = cs_inv[r] * inv_kx;
      const float rho = (p.sc_scale[co_blk * 128 + cf * 32 + 4 * kg + (r & 3) + 8 * (r >> 2)] / xs) / as3;      // (the 1x1 layer's scale is per channel too)
      const int co = co_blk * 128 + cf * 32 + 4 * kg + (r & 3) + 8 * (r >> 2);
      const float add = sgn * (acc_raw[r] + (p.sc_bias ? p.sc_bias[co < p.Cout ? co : 0] : 0.f)) / as3;
      if (kh == 0) {
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[0][f][r] = acc[0][f][r] * rho + add;
      } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[1][f][r] = acc[1][f][r] * rho + add;
      }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) if (it_run[i]) load_item(i, 0);
  }
  __syncthreads();          // s_co visible (SC: and the last shortcut stage consumed)
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    if (it_run[i]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) stage_chan(i, c, 0);
      flush_item(i, s_in0);
      load_item(i, (nst > 1 ? 1 : 0) * G::KC);        // raw inputs one full stage ahead (see the K loop)
    }
  }
  __syncthreads();

  // raw inputs ONE FULL STAGE ahead: an item's registers are reloaded (stage st + 2) right behind the tap whose fragments carried
  // its producer + transform (stage st + 1), so an HBM load has six taps (~72 MFMAs per wave, x 2 waves per SIMD) to arrive
  if constexpr (TRACE) { if (trace) trace[2] = drt_clock(); }
  constexpr int AR = 3, AD = AR - 1, NTAP = 6;
  u32x4 ar[AR][NS];
#pragma unroll
  for (int t = 0; t < AD; ++t) load_a(0, t, ar[t]);
  unsigned long long tbar = 0, ttap[NTAP] = {0, 0, 0, 0, 0, 0};
  const bool tracer = TRACE && p.trace && (tid == 0 || tid == 256);       // one lane of an A-wave and of a B-wave time their taps
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int stn = st + 1 < nst ? st + 1 : st;       // the last stage re-stages itself into the buffer nobody reads again
    const int stl = st + 2 < nst ? st + 2 : nst - 1;
    const u32x4* cur = (st & 1) ? s_in1 : s_in0;
    u32x4* nxt = (st & 1) ? s_in0 : s_in1;
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      unsigned long long tc0 = 0;
      if constexpr (TRACE) tc0 = drt_clock();
      const int ntap = (tap + AD) % NTAP;
      const int nstg = tap + AD < NTAP ? st : stn;
      load_a(nstg, ntap, ar[(tap + AD) % AR]);        // issued before the raw loads below: vmcnt retires in order
      __builtin_amdgcn_sched_barrier(0);
      // staging of the next stage's items: the A-waves in the FIRST NIT taps, the B-waves in the LAST -- the two waves of a SIMD are
      // (cf, kh = 0) and (cf, kh = 1), and in lock step they would run their VALU-heavy taps together and then compete for the matrix
      // pipe together; de-phased, one wave's producer arithmetic issues beside the other's MFMAs.  (Measured alternatives, all slower:
      // B-waves in taps 2-3, +2-3 %; the staging cut into 16 units spread one per fragment slot over the stage, +5 %.)
      // (one call with a uniform predicate around the staging pieces only: two calls under a branch put every accumulator in a phi)
      constexpr int BT0 = NTAP - NIT;
      const bool ca = tap < NIT, cb = tap >= BT0 && tap < BT0 + NIT;   // (constants once the tap loop is unrolled)
      const int item = ca ? tap : cb ? tap - BT0 : -1;
      // Round 5: the LAST stage has nothing to stage and the last TWO have nothing to load -- until then the last stage re-staged itself
      // into the buffer nobody reads again (producer, transform, split and LDS stores of a full tile) and both reloaded the last stage's
      // raw inputs: 1/8 of a 128-channel layer's staging work and 1/4 of its raw loads, dead (SGMSE_WINO_TAIL_STAGING=1 restores it for the A/B)
      const bool stage_live = SGMSE_WINO_TAIL_STAGING || st + 1 < nst, load_live = SGMSE_WINO_TAIL_STAGING || st + 2 < nst;
      const bool stage_here = !(ABL & 8) && item >= 0 && it_run[item < 0 ? 0 : item] && (ca ? kh == 0 : kh == 1) && stage_live;
      compute_tap(cur, tap, 2 * kh + (tap & 1), ar[tap % AR], item, stage_here, stn * G::KC, nxt);
      if constexpr (!(ABL & 16)) { if (stage_here && load_live) load_item(item < 0 ? 0 : item, stl * G::KC); }
      if constexpr (TRACE) { __builtin_amdgcn_sched_barrier(0); ttap[tap] += drt_clock() - tc0; }
    }
    if constexpr (TRACE) {
      const unsigned long long tb = drt_clock();
      __syncthreads();
      tbar += drt_clock() - tb;
    } else {
      __syncthreads();
    }
  }
  if constexpr (TRACE) {
    if (trace) { trace[3] = drt_clock(); trace[6] = tbar; }
    if (tracer) {
      unsigned long long* tq = p.trace + 32 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x) + (tid == 0 ? 16 : 23);
#pragma unroll
      for (int i = 0; i < NTAP; ++i) tq[i] = ttap[i];
      tq[6] = tbar;
    }
  }

  // ---- output transform + epilogue in the accumulators' own (position) layout -----------------------------------------------------
  // A-waves hold {a0, a1} = {M0 + M1, M1}, B-waves {b0, b1} = {M2, M2 + M3}; y(2m) = a0 + b0, y(2m+1) = a1 - b1.  The partners swap
  // HALF of their partial sums through LDS (one barrier): the A-wave of a channel fragment finishes the fragments of rows 0-3, the
  // B-wave those of rows 4-7 (4-row shape: the A-wave takes everything) -- each wave ends with 32 channels x 4 rows x 32 columns, a
  // lane holding the column pair (2m, 2m+1) of row 2f + (l31 >> 4): 8-byte residual loads and output stores (16 lanes = one 128-byte
  // row segment), GroupNorm partials of the 4-row sub-tile by ONE 32-lane butterfly per wave, no second pass through LDS.
  // (the loop's last barrier has passed: nobody reads the stage buffers any more)
  constexpr int EF = 2;                                   // fragments (row pairs) a wave finishes
  constexpr bool ALL_TO_A = ROWS == 4;
  const bool fin_wave = !ALL_TO_A || kh == 0;
  const int ef0 = ALL_TO_A ? 0 : 2 * kh;                   // first fragment this wave finishes
  // (the lane's coordinates derived afresh from the thread id: kept alive across the K loop from the prologue they were spilled)
  int tid_e = (int)threadIdx.x;
  DRT_PIN_INT(tid_e);
  const int lane_e = tid_e & 63, l31_e = lane_e & 31, kg_e = lane_e >> 5;
  const int pos_row = l31_e >> 4, m2 = 2 * (l31_e & 15);
  const int yb = y0 + 2 * ef0 + pos_row, x = x0 + m2;      // row of fragment slot 0, first column of the pair
  const bool okc = x < W;
  const bool inside = x0 + 32 <= W && y0 + ROWS <= H;      // workgroup-uniform: no access of the tile needs a guard
  const size_t ubase = (size_t)b * p.Cout * HW;
  const int co_l = co_blk * 128 + cf * 32 + 4 * kg_e;        // + (r & 3) + 8 (r >> 2)
  // byte offset of fragment slot f of this lane_e inside the utterance (clamped into the image: guarded accesses never use the value)
  unsigned lane_boff[EF];
#pragma unroll
  for (int f = 0; f < EF; ++f) {
    const int yy = yb + 2 * f < H ? yb + 2 * f : H - 1;
    lane_boff[f] = (((unsigned)co_l * (unsigned)H + (unsigned)yy) * (unsigned)W + (unsigned)(okc ? x : 0)) * 4u;
  }
  auto soff = [&](int r) -> unsigned { return (unsigned)((r & 3) + 8 * (r >> 2)) * HW * 4u; };
  const drt_buf obuf = drt_make_buf(p.out + ubase), rbuf = drt_make_buf(p.res ? p.res + ubase : p.out);
  // residual rows first: their HBM latency passes behind the exchange.  Unpredicated (clamped addresses): a predicated load
  // compiles to a branch and a full vmcnt wait each
  const bool has_res = !SC && p.res != nullptr;          // (a folded shortcut replaces the residual: ConvArgs::sc_*)
  float2 rr[SC ? 1 : EF][SC ? 1 : 16];
  if constexpr (!SC) {
    if (fin_wave && has_res) {
#pragma unroll
      for (int f = 0; f < EF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) rr[f][r] = drt_buf_load2(rbuf, lane_boff[f], soff(r));
    }
  }
  if constexpr (TRACE) { if (trace) trace[7] = drt_clock() - trace[3]; }       // residual loads issued
  {
    // exchange slots: [channel fragment][direction: 0 = for the A-wave, 1 = for the B-wave][fragment slot][register][lane_e] float2
    float2* xs = reinterpret_cast<float2*>(s_all);
    auto slot = [&](int dir, int f, int r) -> float2* { return xs + ((((cf * G::NDIR + dir) * EF + f) * 16 + r) * 64 + lane_e); };
    if (kh == 0) {
      if constexpr (!ALL_TO_A) {
#pragma unroll
        for (int f = 0; f < EF; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float a1 = acc[1][EF + f][r]; *slot(1, f, r) = make_float2(acc[0][EF + f][r] + a1, a1); }
      }
    } else {
#pragma unroll
      for (int f = 0; f < EF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float b0 = acc[0][f][r]; *slot(0, f, r) = make_float2(b0, b0 + acc[1][f][r]); }
    }
    if constexpr (TRACE) { if (trace) trace[8] = drt_clock() - trace[3]; }     // partial sums written
    __syncthreads();
    if constexpr (TRACE) { if (trace) trace[9] = drt_clock() - trace[3]; }     // barrier passed
    if (!fin_wave) return;
    if (kh == 0) {
#pragma unroll
      for (int f = 0; f < EF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float2 bb = *slot(0, f, r);
          const float a1 = acc[1][f][r], a0 = acc[0][f][r] + a1;
          acc[0][f][r] = a0 + bb.x; acc[1][f][r] = a1 - bb.y;
        }
    } else {
      if constexpr (!ALL_TO_A) {
#pragma unroll
        for (int f = 0; f < EF; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float2 aa = *slot(1, f, r);
            const float b0 = acc[0][EF + f][r], b1 = b0 + acc[1][EF + f][r];
            acc[0][f][r] = aa.x + b0; acc[1][f][r] = aa.y - b1;       // (into the slots 0 .. EF - 1: the loop below reads those)
          }
      }
    }
  }
  if constexpr (TRACE) { if (trace) trace[5] = drt_clock() - trace[3]; }
  auto finish = [&](auto guard_tag) {
    constexpr bool GUARD = decltype(guard_tag)::value;
    float cs_inv[16];
    load_cs(cs_inv);
    float s1[16], s2[16], vmax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
    for (int f = 0; f < EF; ++f) {
      const bool ok = !GUARD || (okc && yb + 2 * f < H);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v0 = acc[0][f][r] * cs_inv[r] * inv_kx, v1 = acc[1][f][r] * cs_inv[r] * inv_kx;      // exact powers of two
        if constexpr (!SC) { if (has_res) { v0 += rr[f][r].x; v1 += rr[f][r].y; } }
        v0 *= p.out_scale; v1 *= p.out_scale;
        if (ok) drt_buf_store2(obuf, make_float2(v0, v1), lane_boff[f], soff(r));
        if (GUARD) { v0 = ok ? v0 : 0.f; v1 = ok ? v1 : 0.f; }
        vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
        s1[r] = (s1[r] + v0) + v1;
        s2[r] = (s2[r] + v0 * v0) + v1 * v1;
      }
      DRT_PIN_HERE(vmax);
    }
    if constexpr (TRACE) { if (trace) trace[10] = drt_clock() - trace[3]; }    // outputs stored (issued)
    if (p.stats_out && y0 + 2 * ef0 < H) {
      // {sum, sum of squares} of the wave's 4-row x 32-column sub-tile per channel: the 16 per-lane_e sums of each kind through one
      // exchange-add butterfly over the 32 lanes of a half wave (as conv_epilogue, once per wave instead of once per row)
      auto butterfly = [&](float (&sv)[16]) -> float {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = drt_xadd<16>(sv[k], sv[k + 8]);
        float c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = drt_xadd<8>(a[k], a[k + 4]);
        const float d0 = drt_xadd<7>(c[0], c[2]), d1 = drt_xadd<7>(c[1], c[3]);
        return drt_add_xor2(drt_xadd<1>(d0, d1));
      };
      const float e2 = butterfly(s2);
      __builtin_amdgcn_sched_barrier(0);
      const float e1 = butterfly(s1);
      const int r = ((l31_e >> 4) & 1) * 8 + ((l31_e >> 3) & 1) * 4 + ((l31_e >> 2) & 1) * 2 + (l31_e & 1);
      const int co = co_l + (r & 3) + 8 * (r >> 2);
      float* so = p.stats_out + ((size_t)(b * p.Cout + co) * p.stats_nsub + (size_t)((y0 + 2 * ef0) >> 2) * tiles_x + tx) * 2;
      so[0] = e1; so[1] = e2;
    }
    if (p.amax_out) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
      if (lane_e == 0) drt_atomic_max_nonneg(p.amax_out + b * kAmaxSpread + ((blockIdx.x * 8 + wave) & (kAmaxSpread - 1)), vmax);
    }
    DRT_CODE_MARKER(GUARD);
  };
  if (inside) finish(std::false_type{}); else finish(std::true_type{});
  if constexpr (TRACE) { if (trace) trace[4] = drt_clock(); }
}

}  // namespace sgmse

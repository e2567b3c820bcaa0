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
  static constexpr int XS = ROWS_ * 32 + 8;           // floats per output channel in the exchange buffer (+8: kg halves 32 banks apart)
  static constexpr int CO_V = 512;                    // producer coefficient table, u32x4 per input channel
  static constexpr int LDS_V = (2 * STAGE_V + CO_V) > (128 * XS / 4) ? (2 * STAGE_V + CO_V) : (128 * XS / 4);
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
template <int ROWS, int ACT, int SC, int TRACE = 0>
__global__ __launch_bounds__(512, 1) void conv3x3_wino_kernel(ConvArgs p) {
  using G = WinoGeom<ROWS>;
  using T = WinoTile<ROWS>;
  using S = SplitH2;
  constexpr int NF = G::NF, NIT = G::NIT, PV = G::PV, NS = 2;
  __shared__ u32x4 s_all[G::LDS_V];
  u32x4* const s_in0 = s_all;
  u32x4* const s_in1 = s_all + G::STAGE_V;
  f32x4* const s_co = reinterpret_cast<f32x4*>(s_all + 2 * G::STAGE_V);
  float* const s_x = reinterpret_cast<float*>(s_all);       // exchange buffer of the epilogue (the stage buffers are dead by then)

  const int tid = threadIdx.x;
  const int wave = drt_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
  const int cf = wave & 3, kh = wave >> 2;
  unsigned long long* trace = nullptr;
  if constexpr (TRACE) {
    if (tid == 0 && p.trace) {
      trace = p.trace + 16 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
      trace[0] = (unsigned long long)drt_hw_id() | ((unsigned long long)drt_xcc_id() << 32);
      trace[1] = drt_clock();
    }
  }
  // One workgroup owns a CU and every workgroup of a launch takes the same time: left alone, all 256 CUs stream their K loops and
  // then their epilogues in lock step -- HBM idles during the loops and is saturated by 256 simultaneous residual reads + output
  // writes behind them.  The workgroups of the FIRST residency round (one per CU) start phase * stagger sleep units late, phase in
  // 0..15 scattered over the CUs; the rounds that follow inherit the offsets.
  if (p.stagger > 0) {
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    if (lin < 256u) {
      const unsigned ph = (lin * 2654435761u) >> 28;
      for (unsigned i = 0; i < ph * (unsigned)p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
  }
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
  auto produce = [&](float x, int ch, bool ok) -> float {
    const f32x4 co = s_co[ch];
    float o = x * co[0] + co[1];
    if constexpr (ACT == 1) {
      const float u = x * co[2] + co[3];
      o = o * __builtin_amdgcn_rcpf(1.0f + drt_exp2(u));       // SiLU: t * sigmoid(t), exponent pre-multiplied by -log2 e
    }
    o = fminf(fmaxf(o, -32752.f), 32752.f);                    // (cannot bind: the scaled bound is < 2^14; keeps |V| finite in fp16)
    return ok ? o : 0.f;                                       // zero padding applies to the producer's OUTPUT
  };
  // input transform of one channel of an item: V[k] from this lane's pair (e, o), the odd column of lane - 1 and the even column of lane + 1
  float V[4][4];
  auto stage_chan = [&](int i, int c, int c0) {
    const int ch = c0 + 4 * it_q[i] + c;
    const float e = produce(rin[i][2 * c], ch, it_ok[i]), o = produce(rin[i][2 * c + 1], ch, it_ok[i]);
    const float ol = drt_wave_shr1(o), er = drt_wave_shl1(e);
    V[0][c] = ol - o; V[1][c] = e + o; V[2][c] = o - e; V[3][c] = e - er;
  };
  auto flush_item = [&](int i, u32x4* sbuf) {
    if (it_wr[i]) {
      uint2* w = reinterpret_cast<uint2*>(sbuf) + it_woff[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t d01[2], d23[2];
        S::split2(V[k][0], V[k][1], d01);
        S::split2(V[k][2], V[k][3], d23);
        w[k * 8] = make_uint2(d01[0], d23[0]);          // (k: 4 u32x4 = 8 halves; split: 1 u32x4 = 2 halves)
        w[k * 8 + 2] = make_uint2(d01[1], d23[1]);
      }
    }
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
  auto compute_tap = [&](const u32x4* sbuf, int tap, int kq, const u32x4 (&a)[NS], int item, int c0n, u32x4* nxt) {
    const int dy = tap >> 1, kk = tap & 1;
    const u32x4* sb = sbuf + b_lane + dy * G::ROW_V + kq * 4;
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
      if (item >= 0) {
        constexpr int CPF = 4 / NF;                   // channels per fragment slot (1 for 8 rows, 2 for 4)
#pragma unroll
        for (int c = 0; c < CPF; ++c) stage_chan(item, f * CPF + c, c0n);
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
    float rsa[8], rsb[8];
    auto load_sc = [&](int c0, float (&dst)[8]) {
      const bool first = c0 < p.sc_C1;
      const float* base = first ? p.sc_src1 + ((size_t)b * p.sc_C1 + c0) * HW : p.sc_src2 + ((size_t)b * p.sc_C2 + (c0 - p.sc_C1)) * HW;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base + (size_t)c * HW) + s_boff);
        dst[2 * c] = v.x; dst[2 * c + 1] = v.y;
      }
    };
    auto store_sc = [&](const float (&src)[8], u32x4* sbuf) {
      if (!s_live) return;
      float ev[4], od[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ev[c] = s_ok ? fminf(fmaxf(src[2 * c] * xs, -65504.f), 65504.f) : 0.f;
        od[c] = s_ok ? -fminf(fmaxf(src[2 * c + 1] * xs, -65504.f), 65504.f) : 0.f;
      }
      uint2* w = reinterpret_cast<uint2*>(sbuf) + s_woff;
      uint32_t d01[2], d23[2];
      S::split2(ev[0], ev[1], d01); S::split2(ev[2], ev[3], d23);
      w[0] = make_uint2(d01[0], d23[0]); w[2] = make_uint2(d01[1], d23[1]);                 // k = 0
      S::split2(od[0], od[1], d01); S::split2(od[2], od[3], d23);
      w[3 * 8] = make_uint2(d01[0], d23[0]); w[3 * 8 + 2] = make_uint2(d01[1], d23[1]);     // k = 3
    };
    const u32x4* wsc = reinterpret_cast<const u32x4*>(p.sc_w) + (size_t)co_blk * nsts * NS * 4 * 64;
    auto load_asc = [&](int st, u32x4 (&a)[NS]) {
      const u32x4* q = wsc + (size_t)st * NS * 4 * 64;
#pragma unroll
      for (int s = 0; s < NS; ++s) a[s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(q + s * 4 * 64) + a_boff);
    };
    load_sc(0, rsa);
    load_sc((nsts > 1 ? 1 : 0) * G::KC, rsb);
    store_sc(rsa, s_in0);
    __syncthreads();
    u32x4 asc[NS];
    // tap index 2 + kh: kernel row 1, accumulator set kh (M0 in the A-waves, M3 in the B-waves); component slot 3 kh in LDS
#pragma unroll 1
    for (int st = 0; st < nsts; st += 2) {
      load_asc(st, asc);
      if (st + 2 < nsts) load_sc((st + 2) * G::KC, rsa);
      if (kh == 0) compute_tap(s_in0, 2, 0, asc, -1, 0, s_in1); else compute_tap(s_in0, 3, 3, asc, -1, 0, s_in1);     // (accumulator set by a constant index)
      if (st + 1 < nsts) store_sc(rsb, s_in1);
      __syncthreads();
      if (st + 1 < nsts) {
        load_asc(st + 1, asc);
        if (st + 3 < nsts) load_sc((st + 3) * G::KC, rsb);
        if (kh == 0) compute_tap(s_in1, 2, 0, asc, -1, 0, s_in0); else compute_tap(s_in1, 3, 3, asc, -1, 0, s_in0);
        if (st + 2 < nsts) store_sc(rsa, s_in0);
        __syncthreads();
      }
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
      const float rho = (*p.sc_scale / xs) / as3;
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
  unsigned long long tbar = 0;
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int stn = st + 1 < nst ? st + 1 : st;       // the last stage re-stages itself into the buffer nobody reads again
    const int stl = st + 2 < nst ? st + 2 : nst - 1;
    const u32x4* cur = (st & 1) ? s_in1 : s_in0;
    u32x4* nxt = (st & 1) ? s_in0 : s_in1;
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int ntap = (tap + AD) % NTAP;
      const int nstg = tap + AD < NTAP ? st : stn;
      load_a(nstg, ntap, ar[(tap + AD) % AR]);        // issued before the raw loads below: vmcnt retires in order
      __builtin_amdgcn_sched_barrier(0);
      int item = tap - (NTAP - NIT);                  // the last NIT taps carry the staging of items 0 .. NIT - 1
      if (item >= 0 && !it_run[item]) item = -1;
      compute_tap(cur, tap, 2 * kh + (tap & 1), ar[tap % AR], item < 0 ? -1 : item, stn * G::KC, nxt);
      if (item >= 0) load_item(item, stl * G::KC);
    }
    if constexpr (TRACE) {
      const unsigned long long tb = drt_clock();
      __syncthreads();
      tbar += drt_clock() - tb;
    } else {
      __syncthreads();
    }
  }
  if constexpr (TRACE) { if (trace) { trace[3] = drt_clock(); trace[6] = tbar; } }

  // ---- output transform through LDS: [co 128][row][col] fp32, XS floats per channel ------------------------------------------
  // (the loop's last barrier has passed: nobody reads the stage buffers any more)
  {
    const int pos_row = l31 >> 4, m2 = 2 * (l31 & 15);
    float* xw = s_x + (cf * 32 + 4 * kg) * G::XS + pos_row * 32 + m2;
    if (kh == 0) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a1 = acc[1][f][r], a0 = acc[0][f][r] + a1;
          *reinterpret_cast<float2*>(xw + ((r & 3) + 8 * (r >> 2)) * G::XS + f * 64) = make_float2(a0, a1);
        }
    }
    __syncthreads();
    if (kh == 1) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float b0 = acc[0][f][r], b1 = b0 + acc[1][f][r];
          float2* q = reinterpret_cast<float2*>(xw + ((r & 3) + 8 * (r >> 2)) * G::XS + f * 64);
          const float2 v = *q;
          *q = make_float2(v.x + b0, v.y - b1);
        }
    }
    __syncthreads();
  }
  if constexpr (TRACE) { if (trace) trace[5] = drt_clock() - trace[3]; }
  // every wave takes 32 channels x 4 rows in the shared epilogue's layout (lane = column); ROWS = 4: the A-waves only
  constexpr int FP = 4;
  if (ROWS == 4 && kh == 1) return;
  f32x16 out[1][FP];
  {
    float cs_inv[16];
    load_cs(cs_inv);
    const float* xr = s_x + (cf * 32 + 4 * kg) * G::XS + (ROWS == 8 ? kh * 4 : 0) * 32 + l31;
#pragma unroll
    for (int j = 0; j < FP; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[0][j][r] = xr[((r & 3) + 8 * (r >> 2)) * G::XS + j * 32] * cs_inv[r];
  }
  ConvArgs q = p;
  q.acc_scale = nullptr;
  conv_epilogue<T, 1, FP, 4, 0, true, true>(q, out, b, co_blk, tx, ty, tiles_x, cf, ROWS == 8 ? kh : 0, l31, kg, inv_kx);
  if constexpr (TRACE) { if (trace) trace[4] = drt_clock(); }
}

}  // namespace sgmse

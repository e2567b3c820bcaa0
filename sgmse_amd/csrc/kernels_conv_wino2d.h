// 3x3 convolution: fp16x2 operand split + TWO-dimensional Winograd F(2x2, 3x3) (round 6; VERDICT r5 "next round" item 1).
//
// Why it was built: the 1-D kernel (kernels_conv_wino.h) issues 12 MFMA K-steps per output pair and kernel column where the direct form
// issues 18; F(2x2, 3x3) issues 16 per 2x2 output tile where the 1-D form issues 24 (-33 % matrix work), with transform constants 0, +-1,
// 1/2, 1/4 only.  What it costs -- and why it is NOT the product path of the wide levels -- is measured in profiles/r06_wino2d_*.txt and
// stated in DESIGN.md section 8: the 16 transformed components of a 2x2 tile need four accumulators per output (the 1-D form: two), so a
// 512-thread workgroup holds 128 co x 4 rows x 32 columns instead of 8 rows; every weight fragment then feeds ONE position fragment
// instead of four (2.7x the weight bytes per output from L2), the staged tile has 6 rows for 4 (halo 1.59x instead of 1.41x), every
// staged element fans out into 4 transformed values instead of 2 (split arithmetic and LDS stores x 2 per output), and the vertical
// transform needs the rows of a tile in one lane (a second pass through LDS).
//
//     d = producer output on the 4 x 4 window (rows 2tr-1 .. 2tr+2, columns 2tc-1 .. 2tc+2 of the tile; zero outside the image)
//     H_j = row transform   H0 = d0 - d2,  H1 = d1 + d2,  H2 = d2 - d1,  H3 = d1 - d3              (along the columns, as the 1-D kernel)
//     V_ij = column transform of the H rows with the same four combinations                          (fp32, then hi + lo fp16)
//     U = G g G^T,  G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]                               (fp64 at pack time, then hi + lo)
//     M_ij = sum over input channels of U_ij V_ij                                                    (16 implicit GEMMs, fp32 accumulation)
//     P_i0 = (M_i0 + M_i1) + M_i2,  P_i1 = M_i1 - (M_i2 + M_i3);   y(2tr, .) = (P_0 + P_1) + P_2,  y(2tr+1, .) = P_1 - (P_2 + P_3)
//
// Range: |V| <= 4 max|d|, so the input bound goes to [2^12, 2^13) (half the 1-D kernel's scale; exact powers of two throughout);
// |U| <= 2.25 max|g|, covered by the per-output-channel scale computed from the transformed weights themselves.
//
// Layout (one workgroup = 128 co x 4 rows x 32 columns, 8 waves, one workgroup per CU):
//   * a GEMM column is a 2x2 output tile (tr, tc), 32 per workgroup = ONE MFMA B fragment per component; wave (cf, kh) owns channel
//     fragment cf and the components i in {2 kh, 2 kh + 1} x j = 0..3 (8 accumulator fragments = 128 registers);
//   * B operand: [position][component 16][k-group 2][split 2] u32x4, 65 x 16 B per position (conflict-free ds_read_b128), two stages;
//   * staging in two phases: phase 1 -- a lane holds an aligned column pair of 4 channels of one tile row (as the 1-D kernel): producer,
//     row transform through two whole-wave DPP shifts, fp32 result to a scratch tile [row][tile column][hc][channel]; phase 2 -- a lane
//     owns (position, hc, 4 channels): four scratch rows -> column transform -> split -> 8-byte stores into the B tile.  Pipelined over
//     the stages (phase 1 of stage s + 2 and phase 2 of stage s + 1 beside the MFMAs of stage s), one barrier per stage;
//   * A operand: [co block][stage][component 16][split][cf][lane] 16-byte fragments from global memory (L2), ring of four (three components ahead);
//   * epilogue: column transform in registers, partner waves (kh = 0 / 1) swap half of their row-transform terms through LDS; the
//     A-wave ends with the even rows of the tile, the B-wave with the odd rows (a lane holds a column pair: 8-byte stores).
#pragma once
#include "kernels_conv_wino.h"

namespace sgmse {

struct Wino2Geom {
  static constexpr int KC = 16, ROWS = 4, TROWS = 6;
  static constexpr int PV = 65;                          // u32x4 per position: 16 components x 2 k-groups x 2 splits, + 1 (bank spread)
  static constexpr int STAGE_V = 32 * PV;                // 33 280 B
  static constexpr int SCR_E = 68;                       // floats per (tile row, tile column) scratch entry: 4 hc x 16 channels, + 4 (bank spread)
  static constexpr int SCR_F = TROWS * 16 * SCR_E;       // 26 112 B
  static constexpr int SCR_V = SCR_F / 4;
  static constexpr int CO_V = 512;                       // producer coefficient table, u32x4 per input channel
  static constexpr int XCH_V = 4 * 2 * 16 * 64 / 2;      // epilogue exchange [cf][direction][register][lane] float2, in u32x4 (over the dead stage buffers)
  static constexpr int LDS_V = 2 * STAGE_V + 2 * SCR_V + CO_V;      // 126 976 B
  static_assert(XCH_V <= 2 * STAGE_V, "exchange buffer must fit the stage buffers");
};
struct Wino2Tile { static constexpr int CO_T = 128, ROWS = 4; };

// U_ij = (G g G^T)_ij of one 3x3 kernel (row-major g), in fp64
__device__ __forceinline__ double wino2d_u(const float* g, int i, int j) {
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  double u = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) u += G[i][a] * (double)g[a * 3 + b] * G[j][b];
  return u;
}
inline size_t packed_wino2d_frags(int cin, int cout) { return (size_t)((cout + 127) / 128) * (cin / 16) * 16 * 2 * 4 * 64; }
inline size_t packed_wino2d_bytes(int cin, int cout) { return packed_wino2d_frags(cin, cout) * 16 + (size_t)((cout + 127) / 128) * 128 * 4; }

__global__ __launch_bounds__(256) void wino2d_co_scale_kernel(const float* src, int cin, int cout, int cout_pad, float* inv_scale, float* scale) {
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (co >= cout_pad) return;
  float m = 0.f;
  if (co < cout) {
    for (int c = 0; c < cin; ++c) {
      const float* g = src + ((size_t)co * cin + c) * 9;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf((float)wino2d_u(g, i, j)) * 1.0000002f);     // (the fp32 rounding of the bound never undershoots)
    }
  }
  const float s = co < cout ? h2_weight_scale(m) : 1.f;
  scale[co] = s; inv_scale[co] = 1.f / s;
}
// dst: u32x4 [nCoBlk][Cin/16][component i*4+j][split 2][cf 4][lane 64], followed by nCoBlk * 128 inverse channel scales
__global__ __launch_bounds__(256) void pack_weights_wino2d_kernel(PackWinoArgs p, const float* co_scale) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.total) return;
  const int lane = (int)(e & 63);
  size_t r = e >> 6;
  const int cf = (int)(r & 3); r >>= 2;
  const int split = (int)(r & 1); r >>= 1;
  const int comp = (int)(r & 15); r >>= 4;
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
      if (co < p.cout) u = wino2d_u(p.src + ((size_t)co * p.cin + c) * 9, comp >> 2, comp & 3);
      u *= ws;
      const uint32_t hi = drt_f32_to_f16((float)u);
      const uint32_t lo = drt_f32_to_f16((float)(u - (double)drt_f16_to_f32(hi)));
      part[h] = split ? lo : hi;
    }
    o[q] = part[0] | (part[1] << 16);
  }
  reinterpret_cast<u32x4*>(p.dst)[e] = o;
}

// ACT: SiLU behind the GroupNorm affine of the fused producer (1) or the affine only (0).
// ABL (measurement only; results WRONG on purpose): 8 no staging behind the prologue (the MFMA + operand-feed skeleton: what the form
// could reach if its staging were free), 16 no raw loads in the K loop.
template <int ACT, int ABL = 0>
__global__ __launch_bounds__(512, 1) void conv3x3_wino2d_kernel(ConvArgs p) {
  using G = Wino2Geom;
  using T = Wino2Tile;
  using S = SplitH2;
  constexpr int PV = G::PV, NS = 2, ROWS = 4;
  __shared__ u32x4 s_all[G::LDS_V];
  u32x4* const s_st0 = s_all;
  u32x4* const s_st1 = s_all + G::STAGE_V;
  float* const s_scr0 = reinterpret_cast<float*>(s_all + 2 * G::STAGE_V);
  float* const s_scr1 = s_scr0 + G::SCR_F;
  f32x4* const s_co = reinterpret_cast<f32x4*>(s_all + 2 * G::STAGE_V + 2 * G::SCR_V);

  const int tid = threadIdx.x;
  const int wave = drt_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
  const int cf = wave & 3, kh = wave >> 2;
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

  float xb_raw = 0.f;
  if (p.xbound) xb_raw = p.xbound[b * kAmaxSpread + (tid & (kAmaxSpread - 1))];
  const bool cld = p.in_scale != nullptr && tid < Cin;
  const float csc = cld ? p.in_scale[b * Cin + (tid < Cin ? tid : 0)] : 1.f;
  const float csh = cld ? p.in_shift[b * Cin + (tid < Cin ? tid : 0)] : 0.f;

  // ---- phase-1 item of this thread: staging row (q, r) = 3 wave + lane / 18 (q fastest), aligned column pair j = lane % 18 ----
  const int sub1 = lane / 18, j1 = lane - 18 * sub1;
  const int rr1 = 3 * wave + sub1;
  const bool live1 = sub1 < 3;
  const int q1 = rr1 & 3, r1 = live1 ? rr1 >> 2 : 0;
  const int gy1 = y0 - 1 + r1, gx1 = x0 - 2 + 2 * j1;
  const bool ok1 = live1 && gy1 >= 0 && gy1 < H && gx1 >= 0 && gx1 < W;          // (W is even: both columns of an aligned pair are in or out)
  const unsigned boff1 = ((unsigned)(4 * q1) * HW + (ok1 ? (unsigned)(gy1 * W + gx1) : 0u)) * 4u;
  const bool wr1 = live1 && j1 >= 1 && j1 <= 16;
  const int soff1 = (r1 * 16 + (wr1 ? j1 - 1 : 0)) * G::SCR_E + q1 * 4;          // floats; + hc * 16
  float rin[8];
  auto load_raw = [&](int c0) {
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((size_t)b * p.C1 + c0) * HW : p.src2 + ((size_t)b * p.C2 + (c0 - p.C1)) * HW;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base + (size_t)c * HW) + boff1);
      rin[2 * c] = v.x; rin[2 * c + 1] = v.y;
    }
  };
  auto produce_co = [&](float x, const f32x4& co, bool ok) -> float {
    float o = x * co[0] + co[1];
    if constexpr (ACT == 1) {
      const float u = x * co[2] + co[3];
      o = o * __builtin_amdgcn_rcpf(1.0f + drt_exp2(u));
    }
    return ok ? o : 0.f;
  };
  float Hh[4][4];                                         // [hc][channel of the item]
  auto phase1_chan = [&](int c, const f32x4& co) {
    const float e = produce_co(rin[2 * c], co, ok1), o = produce_co(rin[2 * c + 1], co, ok1);
    const float ol = drt_wave_shr1(o), er = drt_wave_shl1(e);
    Hh[0][c] = ol - o; Hh[1][c] = e + o; Hh[2][c] = o - e; Hh[3][c] = e - er;
  };
  auto phase1_store = [&](float* scr) {
    if (wr1) {
#pragma unroll
      for (int hc = 0; hc < 4; ++hc) {
        f32x4 v = {Hh[hc][0], Hh[hc][1], Hh[hc][2], Hh[hc][3]};
        *reinterpret_cast<f32x4*>(scr + soff1 + hc * 16) = v;
      }
    }
  };
  auto phase1 = [&](int c0, float* scr) {
    f32x4 co4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) co4[c] = s_co[c0 + 4 * q1 + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) phase1_chan(c, co4[c]);
    phase1_store(scr);
  };
  // ---- phase-2 item of this thread: position (tr, tc), row-transform component hc, 4-channel group q ----
  const int q2 = tid & 3, hc2 = (tid >> 2) & 3, tc2 = (tid >> 4) & 15, tr2 = tid >> 8;
  const int roff2 = ((2 * tr2) * 16 + tc2) * G::SCR_E + hc2 * 16 + q2 * 4;        // floats; + k * 16 * SCR_E for the window's row k
  const int woff2 = (((tr2 * 16 + tc2) * PV + hc2 * 4 + (q2 >> 1) * NS) * 2) + (q2 & 1);      // 8-byte units; + i * 32 (component i*4+hc), + 2 (lo term)
  f32x4 Vv[4];
  auto phase2_load = [&](const float* scr) {
    f32x4 R[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) R[k] = *reinterpret_cast<const f32x4*>(scr + roff2 + k * 16 * G::SCR_E);
    // (element by element: whole-vector arithmetic compiles to v_pk_add_f32, whose results are not safe on a shared device -- kernels_conv_thin.h)
#pragma unroll
    for (int c = 0; c < 4; ++c) { Vv[0][c] = R[0][c] - R[2][c]; Vv[1][c] = R[1][c] + R[2][c]; Vv[2][c] = R[2][c] - R[1][c]; Vv[3][c] = R[1][c] - R[3][c]; }
  };
  auto phase2_store = [&](int i, u32x4* stg) {
    uint2* w = reinterpret_cast<uint2*>(stg) + woff2 + i * 32;
    uint32_t d01[2], d23[2];
    S::split2(Vv[i][0], Vv[i][1], d01);
    S::split2(Vv[i][2], Vv[i][3], d23);
    w[0] = make_uint2(d01[0], d23[0]);
    w[2] = make_uint2(d01[1], d23[1]);
  };
  auto phase2 = [&](const float* scr, u32x4* stg) {
    phase2_load(scr);
#pragma unroll
    for (int i = 0; i < 4; ++i) phase2_store(i, stg);
  };

  const int nst = Cin / G::KC;
  const float* cs_tab = p.co_scale + (size_t)co_blk * 128;
  float acc_raw[16];
  conv_acc_raw<T>(p, b, co_blk, cf, kg, acc_raw);
  auto load_cs = [&](float (&cs)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[r] = cs_tab[cf * 32 + 4 * kg + (r & 3) + 8 * (r >> 2)];
  };
  load_raw(0);
  float kx = 1.f;
  if (p.xbound) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) xb_raw = fmaxf(xb_raw, __shfl_xor(xb_raw, o));
    kx = 0.5f * h2_weight_scale(xb_raw);                 // bound in [2^12, 2^13): |V| <= 4 max|d| stays below the fp16 maximum
  }
  const float inv_kx = 1.f / kx;
  if (tid < Cin) {
    constexpr float nl2e = -1.4426950408889634f;
    f32x4 v;
    v[0] = csc * kx; v[1] = csh * kx; v[2] = csc * nl2e; v[3] = csh * nl2e;
    s_co[tid] = v;
  }
  // accumulators: fragment c = ii * 4 + j, component (i = 2 kh + ii, j).  The additive terms (bias + time-embedding row, accumulator
  // units) reach y(2tr, 2tc) through M_00, y(2tr, 2tc+1) through -M_03, y(2tr+1, 2tc) through -M_30 and y(2tr+1, 2tc+1) through M_33
  f32x16 acc[8];
  {
    float init[16];
    float cs_inv[16];
    load_cs(cs_inv);
#pragma unroll
    for (int r = 0; r < 16; ++r) init[r] = acc_raw[r] * (kx / cs_inv[r]);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int ii = c >> 2, j = c & 3;
      const bool corner = (kh == 0 ? ii == 0 : ii == 1) && (j == 0 || j == 3);
      const float sgn = ((kh == 0) == (j == 0)) ? 1.f : -1.f;          // kh 0: +M_00, -M_03;  kh 1: -M_30, +M_33
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = corner ? sgn * init[r] : 0.f;
    }
  }

  const u32x4* wblk = reinterpret_cast<const u32x4*>(p.w) + (size_t)co_blk * nst * 16 * NS * 4 * 64;
  const unsigned a_boff = (unsigned)((cf * 64 + lane) * 16);
  auto load_a = [&](int st, int c, u32x4 (&a)[NS]) {
    const u32x4* q = wblk + (size_t)(st * 16 + 8 * kh + c) * NS * 4 * 64;
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(q + s * 4 * 64) + a_boff);
  };
  const int b_lane = l31 * PV + kg * NS + 8 * kh * 4;    // + c * 4 (component 8 kh + c), + split

  __syncthreads();                                        // s_co visible
  // prologue: stage 0 through both phases, stage 1 through phase 1, raw inputs of stage 2 in registers
  phase1(0, s_scr0);
  load_raw((nst > 1 ? 1 : 0) * G::KC);
  __syncthreads();
  phase2(s_scr0, s_st0);
  if (nst > 1) phase1(G::KC, s_scr1);
  load_raw((nst > 2 ? 2 : nst - 1) * G::KC);
  __syncthreads();

  // (ring of FOUR fragment pairs, three components ahead: the ring index must repeat with the stage, 8 % AR == 0)
  constexpr int AR = 4, AD = AR - 1, NC = 8;
  u32x4 ar[AR][NS];
#pragma unroll
  for (int t = 0; t < AD; ++t) load_a(0, t, ar[t]);
  u32x4 bq[2][NS];
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int stn = st + 1 < nst ? st + 1 : st;
    const u32x4* cur = (st & 1) ? s_st1 : s_st0;
    u32x4* nxt = (st & 1) ? s_st0 : s_st1;                // B tile of stage st + 1 (phase 2 writes it)
    const float* scr_rd = (st & 1) ? s_scr0 : s_scr1;     // scratch of stage st + 1
    float* scr_wr = (st & 1) ? s_scr1 : s_scr0;           // scratch of stage st + 2 (the one phase 2 read during stage st - 1)
    const bool p2_live = !(ABL & 8) && st + 1 < nst, p1_live = !(ABL & 8) && st + 2 < nst, ld_live = !(ABL & (8 | 16)) && st + 3 < nst;
    const u32x4* sb = cur + b_lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) bq[0][s] = sb[s];
    f32x4 co4[4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int nc = (c + AD) % NC;
      load_a(c + AD < NC ? st : stn, nc, ar[(c + AD) % AR]);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < NC) {
#pragma unroll
        for (int s = 0; s < NS; ++s) bq[(c + 1) & 1][s] = sb[(c + 1) * 4 + s];
      }
      // staging slot of this component: the A-waves stage behind their first four components, the B-waves behind their last four
      // (the two waves of a SIMD are (cf, 0) and (cf, 1): one's staging arithmetic issues beside the other's MFMAs)
      const int slot = kh == 0 ? c : c - 4;
      if (slot == 2 && p1_live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) co4[k] = s_co[(st + 2) * G::KC + 4 * q1 + k];
      }
      __builtin_amdgcn_sched_barrier(SGMSE_SPLIT_FENCE);
#pragma unroll
      for (int k = 0; k < S::NP; ++k) acc[c] = S::mfma(ar[c % AR][S::pa(k)], bq[c & 1][S::pb(k)], acc[c]);
      if (slot == 0 && p2_live) { phase2_load(scr_rd); phase2_store(0, nxt); phase2_store(1, nxt); }
      if (slot == 1 && p2_live) { phase2_store(2, nxt); phase2_store(3, nxt); }
      if (slot == 2 && p1_live) { phase1_chan(0, co4[0]); phase1_chan(1, co4[1]); }
      if (slot == 3 && p1_live) { phase1_chan(2, co4[2]); phase1_chan(3, co4[3]); phase1_store(scr_wr); }
      if (slot == 3 && ld_live) load_raw((st + 3) * G::KC);
      __builtin_amdgcn_sched_barrier(SGMSE_SPLIT_FENCE);
    }
    __syncthreads();
  }

  // ---- output transform + epilogue --------------------------------------------------------------------------------------------------
  // column transform in registers; then the partner waves swap the row-transform term the other one finishes:
  //   A-wave (i = 0, 1): keeps a0 = P_0 + P_1 (even rows), sends a1 = P_1;   B-wave (i = 2, 3): sends b0 = P_2, keeps b1 = P_2 + P_3 (odd rows)
  //   y(2tr) = a0 + b0,   y(2tr + 1) = a1 - b1
  f32x16 keep[2], send[2];
  {
    // (register by register, not whole-vector arithmetic: that compiles to v_pk_add_f32 -- see kernels_conv_thin.h)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float P[2][2];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        P[ii][0] = (acc[ii * 4 + 0][r] + acc[ii * 4 + 1][r]) + acc[ii * 4 + 2][r];
        P[ii][1] = acc[ii * 4 + 1][r] - (acc[ii * 4 + 2][r] + acc[ii * 4 + 3][r]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (kh == 0) { keep[c][r] = P[0][c] + P[1][c]; send[c][r] = P[1][c]; }
        else { send[c][r] = P[0][c]; keep[c][r] = P[0][c] + P[1][c]; }
      }
    }
  }
  int tid_e = (int)threadIdx.x;
  DRT_PIN_INT(tid_e);
  const int lane_e = tid_e & 63, l31_e = lane_e & 31, kg_e = lane_e >> 5;
  const int y = y0 + 2 * (l31_e >> 4) + kh, x = x0 + 2 * (l31_e & 15);
  const bool okc = x < W && y < H;
  const bool inside = x0 + 32 <= W && y0 + ROWS <= H;
  const size_t ubase = (size_t)b * p.Cout * HW;
  const int co_l = co_blk * 128 + cf * 32 + 4 * kg_e;
  const unsigned lane_boff = (((unsigned)co_l * (unsigned)H + (unsigned)(y < H ? y : H - 1)) * (unsigned)W + (unsigned)(x < W ? x : 0)) * 4u;
  auto soff = [&](int r) -> unsigned { return (unsigned)((r & 3) + 8 * (r >> 2)) * HW * 4u; };
  const drt_buf obuf = drt_make_buf(p.out + ubase), rbuf = drt_make_buf(p.res ? p.res + ubase : p.out);
  const bool has_res = p.res != nullptr;
  float2 rr[16];
  if (has_res) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rr[r] = drt_buf_load2(rbuf, lane_boff, soff(r));
  }
  {
    float2* xs = reinterpret_cast<float2*>(s_all);
    auto slotp = [&](int dir, int r) -> float2* { return xs + (((cf * 2 + dir) * 16 + r) * 64 + lane_e); };
#pragma unroll
    for (int r = 0; r < 16; ++r) *slotp(kh == 0 ? 1 : 0, r) = make_float2(send[0][r], send[1][r]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 o = *slotp(kh, r);
      if (kh == 0) { keep[0][r] = keep[0][r] + o.x; keep[1][r] = keep[1][r] + o.y; }
      else { keep[0][r] = o.x - keep[0][r]; keep[1][r] = o.y - keep[1][r]; }
    }
  }
  float cs_inv[16];
  load_cs(cs_inv);
  float s1[16], s2[16], vmax = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool ok = inside || okc;
    float v0 = keep[0][r] * cs_inv[r] * inv_kx, v1 = keep[1][r] * cs_inv[r] * inv_kx;      // exact powers of two
    if (has_res) { v0 += rr[r].x; v1 += rr[r].y; }
    v0 *= p.out_scale; v1 *= p.out_scale;
    if (ok) drt_buf_store2(obuf, make_float2(v0, v1), lane_boff, soff(r));
    v0 = ok ? v0 : 0.f; v1 = ok ? v1 : 0.f;
    vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
    s1[r] = v0 + v1;
    s2[r] = v0 * v0 + v1 * v1;
  }
  if (p.stats_out) {
    // {sum, sum of squares} of the tile's 4-row x 32-column sub-tile per channel: each wave reduces its two rows by one 32-lane
    // butterfly (as the 1-D kernel), the B-wave hands its pair to the A-wave through LDS, which adds and stores
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
    float2* ex = reinterpret_cast<float2*>(s_scr0) + cf * 64 + lane_e;
    if (kh == 1) *ex = make_float2(e1, e2);
    __syncthreads();
    if (kh == 0 && y0 < H) {
      const float2 o = *ex;
      const int r = ((l31_e >> 4) & 1) * 8 + ((l31_e >> 3) & 1) * 4 + ((l31_e >> 2) & 1) * 2 + (l31_e & 1);
      const int co = co_l + (r & 3) + 8 * (r >> 2);
      float* so = p.stats_out + ((size_t)(b * p.Cout + co) * p.stats_nsub + (size_t)(y0 >> 2) * tiles_x + tx) * 2;
      so[0] = e1 + o.x; so[1] = e2 + o.y;
    }
  }
  if (p.amax_out) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane_e == 0) drt_atomic_max_nonneg(p.amax_out + b * kAmaxSpread + ((blockIdx.x * 8 + wave) & (kAmaxSpread - 1)), vmax);
  }
}

}  // namespace sgmse

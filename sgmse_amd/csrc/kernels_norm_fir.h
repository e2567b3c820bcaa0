// GroupNorm statistics / affine and FIR resampling kernels (HBM-bound class).
//
// GroupNorm: nn.GroupNorm(min(C//4,32), C, eps=1e-6) of reference layerspp.py:67,219,231 and ncsnpp.py:218,230.
//   gn_chan_stats  : per-(b,c) plane sum and sum of squares (wave-shuffle + LDS tree), virtual concat of 2 sources
//   gn_finalize    : per-(b,group) mean / rstd (fp64 combine of the channel partials) folded with gamma/beta into
//                    scale[b][c], shift[b][c]; consumers (conv loaders, FIR kernels, gn_apply) evaluate x*scale+shift
//   gn_apply       : standalone act(x*scale+shift) (op-level API / tests; the network fuses this into consumers)
// FIR: upfirdn2d with the separable [1,3,3,1] kernel (reference up_or_down_sampling.py:195-257, op/upfirdn2d.py,
//   op/upfirdn2d_kernel.cu modes 3 and 5) in closed form, optional fused producer
//   act(x*scale+shift) with zero padding applied after it; upfirdn2d_generic covers every other
//   (kernel, up, down, pad) of the native op's contract.
#pragma once
#include <sgmse_devrt.h>
#include "kernels_conv.h"

namespace sgmse {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// grid = B*C blocks of 256 threads; stats[(b*C+c)*2 + {0,1}] = {sum, sumsq}
// ragged launch (rag.w): HW of utterance b = H * rag.w[b], its planes at rag.off[b] * C
__global__ __launch_bounds__(256) void gn_chan_stats_kernel(const float* src1, const float* src2, int C1, int C2, int HW,
                                                            float* stats, Rag rag, int H) {
  __shared__ float s_a[4];
  __shared__ float s_b[4];
  const int C = C1 + C2;
  const int b = blockIdx.x / C, c = blockIdx.x % C;
  size_t pix0 = (size_t)b * HW;
  if (rag.w) { HW = H * rag.w[b]; pix0 = (size_t)rag.off[b]; }
  const float* plane = (c < C1) ? src1 + pix0 * C1 + (size_t)c * HW : src2 + pix0 * C2 + (size_t)(c - C1) * HW;
  float s = 0.f, q = 0.f;
  if ((HW & 3) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(plane);
    for (int i = threadIdx.x; i < HW / 4; i += 256) {
      const float4 v = p4[i];
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256) { const float v = plane[i]; s += v; q += v * v; }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = s; s_b[threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[(size_t)blockIdx.x * 2 + 0] = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
    stats[(size_t)blockIdx.x * 2 + 1] = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
  }
}

// Per-(b,c) totals come as `nsub` partial {sum, sumsq} pairs per channel (nsub = 1 from gn_chan_stats_kernel, the number
// of statistics sub-tiles when a convolution epilogue produced them); the two sources of a virtual concat may differ.
// grid = (G, B): one workgroup per group.  The partials of a group's channels are contiguous per source, so the 256 threads
// stream them coalesced (the canonical slot order and fp64 tree of gn_sum_pairs, kernels_conv.h: deterministic), then lane c < cpg
// of wave 0 writes the folded coefficients of channel g*cpg + c.
// Ragged launch (rag.w): utterance b has H * rag.w[b] pixels per plane; a source whose partials come from a convolution
// epilogue (nsub > 1) holds them packed per utterance -- (H / rps) * ceil(w / 32) sub-tiles per channel from sub-tile rag.soff[b] / rps on,
// exactly the layout of the utterance's own launch; gn_chan_stats sources (nsub == 1) are one pair per (b, c) either way.
// Range bound for an fp16x2 consumer (bound_out, [B][kAmaxSpread], zeroed by the engine; null: not wanted): an upper bound of
// |x * scale + shift| -- and with it of its SiLU, |silu(t)| <= |t| -- over the utterance, from the utterance's own data
// (gn_group_coeffs).
__global__ __launch_bounds__(256) void gn_finalize_kernel(GnFin f) {
  __shared__ double s_s[256];
  __shared__ double s_q[256];
  const int g = blockIdx.x, b = blockIdx.y;
  const GnGroupSpan sp = gn_group_span(f, b, g);
  double s = 0.0, q = 0.0;
  if (sp.n1) gn_sum_pairs(sp.p1, sp.n1, threadIdx.x, s, q);
  if (sp.n2) gn_sum_pairs(sp.p2, sp.n2, threadIdx.x, s, q);
  s_s[threadIdx.x] = s; s_q[threadIdx.x] = q;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) { s_s[threadIdx.x] += s_s[threadIdx.x + m]; s_q[threadIdx.x] += s_q[threadIdx.x + m]; }
    __syncthreads();
  }
  if (threadIdx.x < 64) gn_group_coeffs(f, b, g, threadIdx.x, s_s[0], s_q[0], sp.HW);      // cpg <= 64: the group's channels sit in wave 0
}

// op-level entry points / micro-benchmarks: the bound gn_finalize_kernel would have left for a producer given as explicit
// per-channel coefficients: max_c |scale_c| m + |shift_c|, m = max|x| of the utterance (amax1 / amax2); no producer: m itself
__global__ __launch_bounds__(64) void xform_bound_kernel(const float* scale, const float* shift, int C, const float* amax1,
                                                         const float* amax2, float* bound_out) {
  const int b = blockIdx.x;
  float m = amax_read(amax1, b);
  if (amax2) m = fmaxf(m, amax_read(amax2, b));
  float bnd = scale ? 0.f : m;
  if (scale)
    for (int c = threadIdx.x; c < C; c += 64) bnd = fmaxf(bnd, fabsf(scale[(size_t)b * C + c]) * m + fabsf(shift[(size_t)b * C + c]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, o));
  bound_out[b * kAmaxSpread + threadIdx.x] = bnd * 1.0001f;
}

// out = act(x*scale+shift); grid = (ceil(HW/1024), B*C)
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* src1, const float* src2, int C1, int C2, int HW,
                                                       const float* scale, const float* shift, int act, float* out) {
  const int C = C1 + C2;
  const int bc = blockIdx.y, b = bc / C, c = bc % C;
  const float* plane = (c < C1) ? src1 + (size_t)(b * C1 + c) * HW : src2 + (size_t)(b * C2 + (c - C1)) * HW;
  float* o = out + (size_t)bc * HW;
  const float a = scale[bc], s = shift[bc];
  const int base = blockIdx.x * 1024;
  for (int k = 0; k < 4; ++k) {
    const int i = base + k * 256 + threadIdx.x;
    if (i < HW) {
      float v = plane[i] * a + s;
      if (act) v = silu_f(v);
      o[i] = v;
    }
  }
}

struct FirArgs {
  const float* src; float* out;
  const float* in_scale; const float* in_shift;  // [B*C] or null: out = FIR(act(x*scale+shift)), zero padding after it
  int in_act;
  int BC, H, W;    // input plane geometry
  float* out_raw;  // optional second result FIR(x) of the same input (the ResBlock shortcut branch, layerspp.py:247-248,
                   // 254-255 resamples both h = act(GN(x)) and x): one pass over x feeds both
  int C;           // channels per utterance (BC = B * C)
  Rag rag;         // ragged launch: geometry of the INPUT level (the output level's prefix is off / 4 or off * 4); W = widest utterance
};

// ragged launch: point this workgroup's argument copy at the utterance of plane bc (same construction as conv_ragged_adjust:
// the uniform addressing bc * H * W then lands on the utterance's own block with its own row stride)
__device__ __forceinline__ void fir_ragged_adjust(FirArgs& p, int bc, bool up) {
  const int b = bc / p.C;
  const int Wb = p.rag.w[b];
  const long long d = p.rag.off[b] - (long long)b * p.H * Wb;
  p.W = Wb;
  p.src += d * p.C;
  const long long dout = up ? d * 4 : d / 4;
  p.out += dout * p.C;
  if (p.out_raw) p.out_raw += dout * p.C;
}

// raw value (0 outside the image) and, through *xv, its fused-producer image (also 0 outside)
__device__ __forceinline__ float fir_fetch(const FirArgs& p, const float* plane, int y, int x, float a, float s, float* xv) {
  if (y < 0 || y >= p.H || x < 0 || x >= p.W) { *xv = 0.f; return 0.f; }
  const float v = plane[y * p.W + x];
  float t = v;
  if (p.in_scale) { t = v * a + s; if (p.in_act) t = silu_f(t); }
  *xv = t;
  return v;
}

// FIR /2: out[i][j] = sum_{a,b} k[a]k[b] x[2i+a-1][2j+b-1], k=[1,3,3,1]/8.  grid = (ceil(Ho*Wo/256), BC)
__global__ __launch_bounds__(256) void fir_down2_kernel(FirArgs p) {
  const int bc = blockIdx.y;
  if (p.rag.w) fir_ragged_adjust(p, bc, false);
  const int Ho = p.H / 2, Wo = p.W / 2;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= Ho * Wo) return;
  const int i = o / Wo, j = o - i * Wo;
  const float* plane = p.src + (size_t)bc * p.H * p.W;
  float a = 1.f, s = 0.f;
  if (p.in_scale) { a = p.in_scale[bc]; s = p.in_shift[bc]; }
  const float k[4] = {0.125f, 0.375f, 0.375f, 0.125f};
  float acc = 0.f, acc_raw = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float row = 0.f, row_raw = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float t;
      const float r = fir_fetch(p, plane, 2 * i + u - 1, 2 * j + v - 1, a, s, &t);
      row += k[v] * t;
      row_raw += k[v] * r;
    }
    acc += k[u] * row;
    acc_raw += k[u] * row_raw;
  }
  p.out[(size_t)bc * Ho * Wo + o] = acc;
  if (p.out_raw) p.out_raw[(size_t)bc * Ho * Wo + o] = acc_raw;
}

// FIR x2 (polyphase): out[2m] = (x[m-1] + 3x[m])/4, out[2m+1] = (3x[m] + x[m+1])/4 per axis.
// One thread per input pixel -> 2x2 outputs.  grid = (ceil(H*W/256), BC)
__device__ __forceinline__ void fir_up2_emit(const float (&v)[3][3], float* out, int Wo) {
  float he[3], ho[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    he[u] = 0.25f * v[u][0] + 0.75f * v[u][1];
    ho[u] = 0.75f * v[u][1] + 0.25f * v[u][2];
  }
  *reinterpret_cast<float2*>(out) = make_float2(0.25f * he[0] + 0.75f * he[1], 0.25f * ho[0] + 0.75f * ho[1]);
  *reinterpret_cast<float2*>(out + Wo) = make_float2(0.75f * he[1] + 0.25f * he[2], 0.75f * ho[1] + 0.25f * ho[2]);
}

__global__ __launch_bounds__(256) void fir_up2_kernel(FirArgs p) {
  const int bc = blockIdx.y;
  if (p.rag.w) fir_ragged_adjust(p, bc, true);
  const int H = p.H, W = p.W;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= H * W) return;
  const int i = o / W, j = o - i * W;
  const float* plane = p.src + (size_t)bc * H * W;
  float a = 1.f, s = 0.f;
  if (p.in_scale) { a = p.in_scale[bc]; s = p.in_shift[bc]; }
  float v[3][3], vr[3][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int w = 0; w < 3; ++w) vr[u][w] = fir_fetch(p, plane, i + u - 1, j + w - 1, a, s, &v[u][w]);
  const int Wo = 2 * W;
  const size_t off = (size_t)bc * 4 * H * W + (size_t)(2 * i) * Wo + 2 * j;
  fir_up2_emit(v, p.out + off, Wo);
  if (p.out_raw) fir_up2_emit(vr, p.out_raw + off, Wo);
}

// LDS-tiled forms of the two FIR resamplers for the wide U-Net levels (W % 4 == 0, W >= 64).  The per-pixel kernels
// above evaluate the fused producer (GroupNorm affine + SiLU) once per tap -- 16x (down) / 9x (up) per input element --
// and fetch with stride-2 addresses; they ran at ~15 % of the HBM rate (profiles/r01_bench_b32_kernel_stats.csv).
// Here a workgroup stages its input tile once with aligned float4 loads, applies the producer once per element, and
// the taps read LDS as aligned b64 pairs.
struct FirTileDown { static constexpr int TO_H = 8, TO_W = 64, IR = 2 * TO_H + 2, RS = 136; };   // LDS col = gx - (2*ox0 - 4)
struct FirTileUp { static constexpr int TI_H = 8, TI_W = 64, IR = TI_H + 2, RS = 72; };          // LDS col = gx - (ix0 - 4)

template <class TT, int ICOLS>
__device__ __forceinline__ void fir_stage_tile(const FirArgs& p, const float* plane, int gy0, int gx0, float a, float s,
                                               float* sx, float* sr, bool want_raw) {
  // rows gy0 .. gy0+IR-1; interior columns gx0 .. gx0+ICOLS-1 at LDS column 4; halo columns gx0-1 (col 3), gx0+ICOLS
  constexpr int Q = ICOLS / 4;
  constexpr int NL = (TT::IR * Q + 255) / 256;      // float4 items per thread (3 for /2, 1 for x2)
  const bool xf = p.in_scale != nullptr;
  // all of a thread's loads first (unconditional, from clamped addresses; masked afterwards), then the producer and the LDS
  // writes: issued one by one behind their uses, a thread had a single 16-byte load in flight and the /2 kernel ran at 3.9 of the
  // ~6 TB/s the x2 kernel's store stream reaches, with its HBM traffic already at the algorithmic minimum
  f32x4 ld[NL];
  bool okv[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    int it = threadIdx.x + 256 * k;
    it = it < TT::IR * Q ? it : TT::IR * Q - 1;
    const int r = it / Q, q = it - r * Q;
    const int gy = gy0 + r, gx = gx0 + 4 * q;
    okv[k] = gy >= 0 && gy < p.H && gx < p.W;      // W % 4 == 0: a float4 is inside or outside as a whole
    const int cy = gy < 0 ? 0 : (gy >= p.H ? p.H - 1 : gy), cx = gx < p.W ? gx : p.W - 4;
    ld[k] = *reinterpret_cast<const f32x4*>(plane + (size_t)cy * p.W + cx);
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int it = threadIdx.x + 256 * k;
    if (it < TT::IR * Q) {
      const int r = it / Q, q = it - r * Q;
      f32x4 v = {0.f, 0.f, 0.f, 0.f}, t = v;
      if (okv[k]) {
        v = ld[k];
        t = v;
        if (xf) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { float u = v[e] * a + s; t[e] = p.in_act ? silu_f(u) : u; }
        }
      }
      *reinterpret_cast<f32x4*>(sx + r * TT::RS + 4 + 4 * q) = t;
      if (want_raw) *reinterpret_cast<f32x4*>(sr + r * TT::RS + 4 + 4 * q) = v;
    }
  }
  for (int it = threadIdx.x; it < TT::IR * 2; it += 256) {
    const int r = it >> 1, side = it & 1;
    const int gy = gy0 + r, gx = side ? gx0 + ICOLS : gx0 - 1;
    float v = 0.f, t = 0.f;
    if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
      v = plane[(size_t)gy * p.W + gx];
      t = v;
      if (xf) { float u = v * a + s; t = p.in_act ? silu_f(u) : u; }
    }
    const int c = side ? 4 + ICOLS : 3;
    sx[r * TT::RS + c] = t;
    if (want_raw) sr[r * TT::RS + c] = v;
  }
}

// Workgroups are dispatched round-robin over the 8 XCDs by their linear id, and every XCD has its own L2: with the plain map the
// four tiles of a tile row (and the rows above / below) sit on different XCDs, so every halo row and halo column line is fetched
// from HBM once per tile -- fir_down2_tiled fetched 1.54 x its input (profiles/r02_hbm_traffic_conv3x3_split_h2_b8.json).  This
// map hands XCD k the k-th contiguous eighth of a plane's tiles (whole tile rows), so neighbours share their halo lines in L2.
// A bijection of the tile index: results unchanged.
__device__ __forceinline__ int fir_xcd_tile(int t, int nt) { return (nt & 7) == 0 ? (t & 7) * (nt >> 3) + (t >> 3) : t; }

// grid = (ceil(Wo/64) * ceil(Ho/8), BC); thread -> output column ox (64) and the output row pair 2*rp, 2*rp+1
__device__ __forceinline__ void fir_down2_tile_emit(const float* s, int rp, int oxl, float* out, int Wo, bool row1) {
  using TT = FirTileDown;
  float h[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const float* row = s + (4 * rp + r) * TT::RS + 2 * oxl + 2;      // taps at columns 2*oxl+3 .. 2*oxl+6
    const float2 p0 = *reinterpret_cast<const float2*>(row), p1 = *reinterpret_cast<const float2*>(row + 2),
                 p2 = *reinterpret_cast<const float2*>(row + 4);
    h[r] = 0.125f * p0.y + 0.375f * p1.x + 0.375f * p1.y + 0.125f * p2.x;
  }
  out[0] = 0.125f * h[0] + 0.375f * h[1] + 0.375f * h[2] + 0.125f * h[3];
  if (row1) out[Wo] = 0.125f * h[2] + 0.375f * h[3] + 0.375f * h[4] + 0.125f * h[5];
}

__global__ __launch_bounds__(256) void fir_down2_tiled_kernel(FirArgs p) {
  using TT = FirTileDown;
  __shared__ float sx[TT::IR * TT::RS];
  __shared__ float sr[TT::IR * TT::RS];
  const int tiles_x = (p.W / 2 + TT::TO_W - 1) / TT::TO_W;          // grid layout (ragged launches: of the widest utterance)
  const int tile = fir_xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int bc = blockIdx.y;
  if (p.rag.w) { fir_ragged_adjust(p, bc, false); if (tx * TT::TO_W >= p.W / 2) return; }
  const int Ho = p.H / 2, Wo = p.W / 2;
  const float* plane = p.src + (size_t)bc * p.H * p.W;
  float a = 1.f, s = 0.f;
  if (p.in_scale) { a = p.in_scale[bc]; s = p.in_shift[bc]; }
  const int ox0 = tx * TT::TO_W, oy0 = ty * TT::TO_H;
  const bool raw = p.out_raw != nullptr;
  fir_stage_tile<TT, 2 * TT::TO_W>(p, plane, 2 * oy0 - 1, 2 * ox0, a, s, sx, sr, raw);
  __syncthreads();
  const int oxl = threadIdx.x & 63, rp = threadIdx.x >> 6;
  const int ox = ox0 + oxl, oy = oy0 + 2 * rp;
  if (ox >= Wo || oy >= Ho) return;
  const size_t o = (size_t)bc * Ho * Wo + (size_t)oy * Wo + ox;
  fir_down2_tile_emit(sx, rp, oxl, p.out + o, Wo, oy + 1 < Ho);
  if (raw) fir_down2_tile_emit(sr, rp, oxl, p.out_raw + o, Wo, oy + 1 < Ho);
}

// grid = (ceil(W/64) * ceil(H/8), BC); thread -> input pixel pair (jx, jx+1) of input row iy -> 2 x 4 outputs
__device__ __forceinline__ void fir_up2_tile_emit(const float* s, int iyl, int jxl, float* out, int Wo) {
  using TT = FirTileUp;
  float he[3][2], ho[3][2];     // horizontal polyphase results of the three input rows, for the two input pixels
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float* row = s + (iyl + r) * TT::RS + jxl + 2;              // columns jxl+3 .. jxl+6 = pixels jx-1 .. jx+2
    const float2 p0 = *reinterpret_cast<const float2*>(row), p1 = *reinterpret_cast<const float2*>(row + 2),
                 p2 = *reinterpret_cast<const float2*>(row + 4);
    he[r][0] = 0.25f * p0.y + 0.75f * p1.x;  ho[r][0] = 0.75f * p1.x + 0.25f * p1.y;
    he[r][1] = 0.25f * p1.x + 0.75f * p1.y;  ho[r][1] = 0.75f * p1.y + 0.25f * p2.x;
  }
  f32x4 r0, r1;
  r0[0] = 0.25f * he[0][0] + 0.75f * he[1][0];  r0[1] = 0.25f * ho[0][0] + 0.75f * ho[1][0];
  r0[2] = 0.25f * he[0][1] + 0.75f * he[1][1];  r0[3] = 0.25f * ho[0][1] + 0.75f * ho[1][1];
  r1[0] = 0.75f * he[1][0] + 0.25f * he[2][0];  r1[1] = 0.75f * ho[1][0] + 0.25f * ho[2][0];
  r1[2] = 0.75f * he[1][1] + 0.25f * he[2][1];  r1[3] = 0.75f * ho[1][1] + 0.25f * ho[2][1];
  *reinterpret_cast<f32x4*>(out) = r0;
  *reinterpret_cast<f32x4*>(out + Wo) = r1;
}

__global__ __launch_bounds__(256) void fir_up2_tiled_kernel(FirArgs p) {
  using TT = FirTileUp;
  __shared__ float sx[TT::IR * TT::RS];
  __shared__ float sr[TT::IR * TT::RS];
  const int tiles_x = (p.W + TT::TI_W - 1) / TT::TI_W;              // grid layout (ragged launches: of the widest utterance)
  const int tile = fir_xcd_tile((int)blockIdx.x, (int)gridDim.x);
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int bc = blockIdx.y;
  if (p.rag.w) { fir_ragged_adjust(p, bc, true); if (tx * TT::TI_W >= p.W) return; }
  const int H = p.H, W = p.W;
  const float* plane = p.src + (size_t)bc * H * W;
  float a = 1.f, s = 0.f;
  if (p.in_scale) { a = p.in_scale[bc]; s = p.in_shift[bc]; }
  const int ix0 = tx * TT::TI_W, iy0 = ty * TT::TI_H;
  const bool raw = p.out_raw != nullptr;
  fir_stage_tile<TT, TT::TI_W>(p, plane, iy0 - 1, ix0, a, s, sx, sr, raw);
  __syncthreads();
  const int jxl = 2 * (threadIdx.x & 31), iyl = threadIdx.x >> 5;
  const int jx = ix0 + jxl, iy = iy0 + iyl;
  if (jx >= W || iy >= H) return;       // W even: the pixel pair is inside or outside as a whole
  const int Wo = 2 * W;
  const size_t off = (size_t)bc * 4 * H * W + (size_t)(2 * iy) * Wo + 2 * jx;
  fir_up2_tile_emit(sx, iyl, jxl, p.out + off, Wo);
  if (raw) fir_up2_tile_emit(sr, iyl, jxl, p.out_raw + off, Wo);
}

inline bool fir_use_tiled(int W) { return W % 4 == 0 && W >= 64; }

// Generic upfirdn2d (reference op/upfirdn2d.py:162-203 semantics): zero-insert by `up`, pad/crop, correlate with the
// flipped kernel, decimate by `down`.  grid = (ceil(Ho*Wo/256), BC).
struct UpfirdnArgs {
  const void* src; const void* kern; void* out;
  int BC, H, W, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, Ho, Wo;
};
// element types the reference op dispatches (AT_DISPATCH_FLOATING_TYPES_AND_HALF, op/upfirdn2d_kernel.cu:311): float, double, half.
// Half is stored as its 16 bits, read into fp32, accumulated in fp32 and rounded once on the way out (the reference's CUDA kernel
// accumulates in scalar_t, i.e. rounds the running sum to half after every tap: this result is the closer one to the exact sum).
struct UpfirdnF32 { using elem = float; using acc = float;
  static __device__ __forceinline__ acc load(const elem* p, size_t i) { return p[i]; }
  static __device__ __forceinline__ void store(elem* p, size_t i, acc v) { p[i] = v; } };
struct UpfirdnF64 { using elem = double; using acc = double;
  static __device__ __forceinline__ acc load(const elem* p, size_t i) { return p[i]; }
  static __device__ __forceinline__ void store(elem* p, size_t i, acc v) { p[i] = v; } };
struct UpfirdnF16 { using elem = uint16_t; using acc = float;
  static __device__ __forceinline__ acc load(const elem* p, size_t i) { return drt_f16_to_f32(p[i]); }
  static __device__ __forceinline__ void store(elem* p, size_t i, acc v) { p[i] = (uint16_t)drt_f32_to_f16(v); } };

template <class E>
__global__ __launch_bounds__(256) void upfirdn2d_generic_kernel(UpfirdnArgs p) {
  using T = typename E::elem;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= p.Ho * p.Wo) return;
  const int bc = blockIdx.y;
  const int oy = o / p.Wo, ox = o - oy * p.Wo;
  const T* plane = static_cast<const T*>(p.src) + (size_t)bc * p.H * p.W;
  const T* kern = static_cast<const T*>(p.kern);
  typename E::acc acc = 0;
  for (int a = 0; a < p.kh; ++a) {
    // position in the zero-inserted, padded signal
    const int zy = oy * p.down_y + a - p.pad_y0;
    if (zy < 0 || zy % p.up_y != 0) continue;
    const int iy = zy / p.up_y;
    if (iy >= p.H) continue;
    for (int c = 0; c < p.kw; ++c) {
      const int zx = ox * p.down_x + c - p.pad_x0;
      if (zx < 0 || zx % p.up_x != 0) continue;
      const int ix = zx / p.up_x;
      if (ix >= p.W) continue;
      acc += E::load(kern, (size_t)(p.kh - 1 - a) * p.kw + (p.kw - 1 - c)) * E::load(plane, (size_t)iy * p.W + ix);
    }
  }
  E::store(static_cast<T*>(p.out), (size_t)bc * p.Ho * p.Wo + o, acc);
}

}  // namespace sgmse

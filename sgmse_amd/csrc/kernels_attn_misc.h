// Self-attention core (MFMA), time-embedding MLP, network entry/exit and the sampler's element-wise updates.
#pragma once
#include <sgmse_devrt.h>
#include "kernels_conv.h"
#include "kernels_norm_fir.h"

namespace sgmse {

// ------------------------------------------------------------------------------------------------------
// Attention core of AttnBlockpp (reference layerspp.py:82-88):
//   w[s,r] = softmax_r( sum_c q[c,s] k[c,r] * C^-1/2 ),  o[c,s] = sum_r w[s,r] v[c,r]
// qkv: [B][3C][S] (q | k | v along channels, S = H*W contiguous), out: [B][C][S].
// One workgroup = 32 queries; its 4 waves take 32-key tiles round-robin with private online-softmax state and are merged
// through LDS at the end.  Both contractions run on v_mfma_f32_32x32x2_f32 (exact fp32):
//   QK^T:  D[i=key][j=query]  A = k[c][key] (lane = key), B = q[c][query] (lane = query)
//   P.V :  D[i=chan][j=query]  B operand of k-step r IS accumulator register r of the score fragment
//          (key(r,kh) = (r&3)+8*(r>>2)+4*kh is exactly the key the lane holds), so P never leaves registers.
// Operand paths (every global access is a 128-byte row segment; round 1 gathered V with stride S across the lanes and
// loaded q / k inside the MFMA loop: 9 % of the fp32 MFMA peak):
//   q: the workgroup's [C][32] tile staged ONCE in LDS, read as the B operand (lane = query: conflict-free);
//   k: [c][32 keys] rows straight from global memory (lane = key), 16 k-steps (32 channels) in registers, the next 16
//      loaded while the MFMAs of the current ones run;
//   v: a wave's [C][32 keys] tile loaded as rows (two rows per wave instruction) one 64-channel chunk ahead, transposed through a
//      wave-private LDS tile with rows padded to 33 floats (bank = channel + key: conflict-free both ways) in chunks of 64
//      channels; the A operand of P.V (lane = channel) is then one ds_read_b32.
// LDS: q 32 KB + 4 x 8.4 KB (C = 256) = 66 KB and at most 256 registers: two workgroups per CU.
struct AttnArgs { const float* qkv; float* out; int B, C, S; float scale; Rag rag; int H; };   // rag.w: S of utterance b = H * rag.w[b]

template <int CF>
__global__ __launch_bounds__(256, 2) void attn_core_kernel(AttnArgs p) {
  constexpr int C = CF * 32;
  constexpr int FH = CF < 2 ? CF : 2;          // channel fragments per V chunk (64 channels: two workgroups per CU fit)
  constexpr int NH = CF / FH;                  // V chunks per key tile
  constexpr int VS = 33;                       // padded row of the transposed V tile
  __shared__ float s_q[C * 32];                // q tile; reused as the cross-wave output accumulator at the end
  __shared__ float s_v[4][FH * 32 * VS];
  __shared__ float s_m[4][32];
  __shared__ float s_l[4][32];
  __shared__ float s_lt[32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int b = blockIdx.y, s0 = blockIdx.x * 32;
  const int S = p.rag.w ? p.H * p.rag.w[b] : p.S;                  // ragged launch: the grid covers the longest utterance
  if (s0 >= S) return;
  const size_t pix0 = p.rag.w ? (size_t)p.rag.off[b] : (size_t)b * S;
  const float* q = p.qkv + pix0 * 3 * C;
  const float* k = q + (size_t)C * S;
  const float* v = k + (size_t)C * S;
  const int nkt = (S + 31) / 32;
  const float NEG_INF = -INFINITY;
  const drt_buf kbuf = drt_make_buf(k), vbuf = drt_make_buf(v);
  const unsigned rowb = (unsigned)S * 4u;                          // bytes per channel row

  for (int e = tid; e < C * 32; e += 256) {                       // q tile: rows of 32 queries
    const int c = e >> 5, j = e & 31;
    s_q[e] = (s0 + j) < S ? q[(size_t)c * S + s0 + j] : 0.f;
  }
  __syncthreads();

  float m = NEG_INF, lsum = 0.f;
  f32x16 O[CF];
#pragma unroll
  for (int f = 0; f < CF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[f][r] = 0.f;
  float* sv = s_v[wave];

  for (int kt = wave; kt < nkt; kt += 4) {
    const int r0 = kt * 32;
    // row (channel 2 i + kh), column (key r0 + l31): one VGPR byte offset per lane + a uniform offset per row pair.  Keys past
    // the end are CLAMPED to the last one instead of predicated (a predicated load is a branch and a full vmcnt wait each):
    // their scores are set to -inf below, so their k values never matter and their (finite) v values meet a weight of exactly 0
    const int kcol = (r0 + l31) < S ? r0 + l31 : S - 1;
    const unsigned lane_boff = (unsigned)(kh * S + kcol) * 4u;
    // this wave's V tile: row (channel) 2 i + kh of chunk h, column (key) l31; in flight during the QK^T loop
    float vreg[FH * 16];
    auto load_v = [&](int h) {
#pragma unroll
      for (int i = 0; i < FH * 16; ++i) vreg[i] = drt_buf_load(vbuf, lane_boff, (unsigned)(h * FH * 32 + 2 * i) * rowb);
    };
    load_v(0);

    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
    // k in chunks of 16 k-steps (32 channels), two register sets: the loop is NOT unrolled across chunk pairs, or the compiler
    // hoists every chunk's loads to the top and spills
    float ka0[16], ka1[16];
    auto load_k = [&](int ch, float (&dst)[16]) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[i] = drt_buf_load(kbuf, lane_boff, (unsigned)(ch * 32 + 2 * i) * rowb);
    };
    auto qk = [&](int ch, const float (&src)[16]) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        sc = __builtin_amdgcn_mfma_f32_32x32x2f32(src[i], s_q[(ch * 32 + 2 * i + kh) * 32 + l31], sc, 0, 0, 0);
    };
    load_k(0, ka0);
    if constexpr (CF == 1) {
      qk(0, ka0);
    } else {
#pragma unroll 1
      for (int ch = 0; ch < CF; ch += 2) {
        load_k(ch + 1, ka1);
        qk(ch, ka0);
        if (ch + 2 < CF) load_k(ch + 2, ka0);
        qk(ch + 1, ka1);
      }
    }
    float mx = NEG_INF;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = r0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      const float val = (key < S) ? sc[r] * p.scale : NEG_INF;
      sc[r] = val;
      mx = fmaxf(mx, val);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx);
    const float alpha = expf(m - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = expf(sc[r] - m_new);
      sc[r] = pr;
      psum += pr;
    }
    psum += __shfl_xor(psum, 32);
    lsum = lsum * alpha + psum;
    m = m_new;
#pragma unroll
    for (int f = 0; f < CF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[f][r] *= alpha;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      drt_wave_sync();                                             // the previous chunk's reads are done (wave-private tile)
#pragma unroll
      for (int i = 0; i < FH * 16; ++i) sv[(2 * i + kh) * VS + l31] = vreg[i];
      drt_wave_sync();
      if (h + 1 < NH) load_v(h + 1);                               // in flight during this chunk's MFMAs
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * kh;
#pragma unroll
        for (int f = 0; f < FH; ++f)
          O[h * FH + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[(f * 32 + l31) * VS + key], sc[r], O[h * FH + f], 0, 0, 0);
      }
    }
  }

  if (kh == 0) { s_m[wave][l31] = m; s_l[wave][l31] = lsum; }
  __syncthreads();                                                 // also: every wave is done with s_q
  float mstar = fmaxf(fmaxf(s_m[0][l31], s_m[1][l31]), fmaxf(s_m[2][l31], s_m[3][l31]));
  const float fac = expf(m - mstar);
  if (wave == 0 && kh == 0) {
    float lt = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) lt += s_l[w][l31] * expf(s_m[w][l31] - mstar);
    s_lt[l31] = lt;
  }
#pragma unroll
  for (int f = 0; f < CF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[f][r] *= fac;
  float* s_O = s_q;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int f = 0; f < CF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ch = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (w == 0) s_O[ch * 32 + l31] = O[f][r];
          else s_O[ch * 32 + l31] += O[f][r];
        }
    }
    __syncthreads();
  }
  for (int e = tid; e < C * 32; e += 256) {
    const int ch = e >> 5, j = e & 31;
    if (s0 + j < S) p.out[(pix0 * C + (size_t)ch * S) + s0 + j] = s_O[e] / s_lt[j];
  }
}

inline bool launch_attn_core(const AttnArgs& a, drt::stream_t st) {
  dim3 grid((a.S + 31) / 32, a.B, 1);
  switch (a.C) {
    case 32: DRT_LAUNCH((attn_core_kernel<1>), grid, dim3(256), st, a); return true;
    case 64: DRT_LAUNCH((attn_core_kernel<2>), grid, dim3(256), st, a); return true;
    case 128: DRT_LAUNCH((attn_core_kernel<4>), grid, dim3(256), st, a); return true;
    case 256: DRT_LAUNCH((attn_core_kernel<8>), grid, dim3(256), st, a); return true;
    default: return false;
  }
}

// ------------------------------------------------------------------------------------------------------
// Time embedding (reference ncsnpp.py:265-284, layerspp.py:39-41): Fourier features of log t, Linear, SiLU, Linear;
// output is SiLU(temb) (every consumer applies act(temb) first, layerspp.py:263).  One workgroup per t value.
// log / sin / cos are evaluated in fp64 and rounded once so they agree with any correctly-rounded fp32 libm.
struct TembArgs {
  const float* t; int nt;
  const float* Wf; const float* w1; const float* b1; const float* w2; const float* b2;
  int nf;          // Fourier size; emb = 2*nf, hidden = 4*nf
  float* act_out;  // [nt][4*nf]
};

__global__ __launch_bounds__(256) void temb_mlp_kernel(TembArgs p) {
  __shared__ float s_emb[512];
  __shared__ float s_h[1024];
  const int row = blockIdx.x, nf = p.nf, ne = 2 * nf, nh = 4 * nf;
  const float logt = (float)log((double)p.t[row]);
  for (int i = threadIdx.x; i < nf; i += 256) {
    float pr = logt * p.Wf[i];
    pr = pr * 2.0f;
    pr = pr * 3.14159265358979323846f;
    s_emb[i] = (float)sin((double)pr);
    s_emb[nf + i] = (float)cos((double)pr);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < nh; o += 256) {
    float acc = 0.f;
    const float* wr = p.w1 + (size_t)o * ne;
    for (int i = 0; i < ne; ++i) acc = fmaf(wr[i], s_emb[i], acc);
    s_h[o] = silu_precise_f(acc + p.b1[o]);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < nh; o += 256) {
    float acc = 0.f;
    const float* wr = p.w2 + (size_t)o * nh;
    for (int i = 0; i < nh; ++i) acc = fmaf(wr[i], s_h[i], acc);
    p.act_out[(size_t)row * nh + o] = silu_precise_f(acc + p.b2[o]);
  }
}

// All per-ResBlock Dense_0 projections (layerspp.py:262-263) in one launch, with Conv_0's bias folded in:
//   table[row][off + co] = Dense_0.b[co] + Conv_0.bias[co] + sum_i Dense_0.W[co][i] * act[row][i]
struct DenseDesc { const float* W; const float* b; const float* cb; int cout; int off; };

__global__ __launch_bounds__(256) void temb_dense_kernel(const DenseDesc* descs, const float* act, int nh, float* table,
                                                         int row_stride) {
  __shared__ float s_a[1024];
  const DenseDesc d = descs[blockIdx.x];
  const int row = blockIdx.y;
  for (int i = threadIdx.x; i < nh; i += 256) s_a[i] = act[(size_t)row * nh + i];
  __syncthreads();
  for (int co = threadIdx.x; co < d.cout; co += 256) {
    float acc = 0.f;
    const float* wr = d.W + (size_t)co * nh;
    for (int i = 0; i < nh; ++i) acc = fmaf(wr[i], s_a[i], acc);
    table[(size_t)row * row_stride + d.off + co] = acc + d.b[co] + d.cb[co];
  }
}

// ------------------------------------------------------------------------------------------------------
// Score-wrapper coefficients (reference ScoreModel.forward, model.py:284-310): the network sees gamma*x_t, gamma*y and the
// score is  alpha * x_t + beta * F  with F the backbone output.  Old-code branch (ncsnpp / ncsnpp_48k, model.py:307-310):
// gamma = 1, alpha = 0, beta = -1.  New-code branch (ncsnpp_v2): gamma = c_in(t); score_matching: alpha = c_skip(t),
// beta = c_out(t) * s(t); denoiser: alpha = -1/sigma^2, beta = s(t)/sigma^2; s(t) = network_scaling (1, 1/sigma, 1/t).
// coef rows of 4 floats {gamma, alpha, beta, -}; row = step * sstride + b * bstride; coef == null -> {1, 0, sign}.
struct WrapCoef { const float* coef; int bstride, sstride; const int* step_ptr; float sign; };

__device__ __forceinline__ void wrap_coef(const WrapCoef& w, int b, float* gamma, float* alpha, float* beta) {
  if (!w.coef) { *gamma = 1.f; *alpha = 0.f; *beta = w.sign; return; }
  const int step = w.step_ptr ? *w.step_ptr : 0;
  const float* r = w.coef + ((size_t)step * w.sstride + (size_t)b * w.bstride) * 4;
  *gamma = r[0]; *alpha = r[1]; *beta = r[2];
}

// Network entry (ncsnpp.py:262-263 / ncsnpp_v2.py:247): complex x_t, y -> real [B][4][F*T] = gamma*(x.re, x.im, y.re, y.im).
// xr8 (optional): the same planes in a C8-channel tensor whose other channels are zero (input of the entry convolution on the MFMA kernels)
// Ragged launch (rag.w): utterance b has FT_b = F * rag.w[b] elements; x_b = x + rag.off[b] * xbs, y_b = y + rag.off[b] * ybs +
// y_plane * FT_b (xbs = ybs = 1 for the sampler's packed arrays; 2, 2 and y_plane = 1 for a packed [B][2][F][T_b] network input).
__global__ __launch_bounds__(256) void entry_kernel(const float2* x, long long xbs, const float2* y, long long ybs,
                                                    float* xr, float* xr8, int C8, int FT, WrapCoef w, Rag rag, int F, int y_plane) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  size_t xo = (size_t)b * xbs, yo = (size_t)b * ybs, po = (size_t)b * FT;
  if (rag.w) {
    FT = F * rag.w[b];
    xo = (size_t)rag.off[b] * xbs; yo = (size_t)rag.off[b] * ybs + (size_t)y_plane * FT; po = (size_t)rag.off[b];
  }
  if (i >= FT) return;
  float g, al, be;
  wrap_coef(w, b, &g, &al, &be);
  const float2 xv = x[xo + i], yv = y[yo + i];
  float* o = xr + po * 4 + i;
  const float v0 = g * xv.x, v1 = g * xv.y, v2 = g * yv.x, v3 = g * yv.y;
  o[0] = v0; o[FT] = v1; o[2 * (size_t)FT] = v2; o[3 * (size_t)FT] = v3;
  if (xr8) {
    float* q = xr8 + po * C8 + i;
    q[0] = v0; q[FT] = v1; q[2 * (size_t)FT] = v2; q[3 * (size_t)FT] = v3;
    for (int k = 4; k < C8; ++k) q[(size_t)k * FT] = 0.f;
  }
}

// Network exit (ncsnpp.py:402-419 / ncsnpp_48k.py:414-421 / ncsnpp_v2.py:388-394) fused with the score wrapper:
//   ncsnpp    : h = h4 / t ; F = conv1x1(4->2)(h)        ncsnpp_48k: F = conv1x1(h4) ; F = F / t      ncsnpp_v2: F = conv1x1(h4)
//   out = alpha * x_t + beta * complex(F0, F1)
struct ExitArgs {
  const float* h4; const float* ow; const float* ob;
  const float* tvals; int t_bstride, t_sstride; const int* step_ptr;
  int conv_first, scale_by_t;
  WrapCoef w; const float2* xt; long long xt_bs;
  float2* out; int FT;
  Rag rag; int F;            // ragged launch: FT_b = F * rag.w[b]; h4 / out at rag.off[b]; xt at rag.off[b] * xt_bs
};

__global__ __launch_bounds__(256) void exit_kernel(ExitArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  int FT = p.FT;
  size_t po = (size_t)b * FT, xo = (size_t)b * p.xt_bs;
  if (p.rag.w) { FT = p.F * p.rag.w[b]; po = (size_t)p.rag.off[b]; xo = po * p.xt_bs; }
  if (i >= FT) return;
  const int step = p.step_ptr ? *p.step_ptr : 0;
  const float t = p.scale_by_t ? p.tvals[(size_t)step * p.t_sstride + (size_t)b * p.t_bstride] : 1.f;
  const float* h = p.h4 + po * 4 + i;
  float hv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) hv[c] = h[(size_t)c * FT];
  if (p.scale_by_t && !p.conv_first) {
#pragma unroll
    for (int c = 0; c < 4; ++c) hv[c] = hv[c] / t;
  }
  float o[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = fmaf(p.ow[k * 4 + c], hv[c], acc);
    o[k] = acc + p.ob[k];
  }
  if (p.scale_by_t && p.conv_first) { o[0] = o[0] / t; o[1] = o[1] / t; }
  float g, al, be;
  wrap_coef(p.w, b, &g, &al, &be);
  float2 r = make_float2(be * o[0], be * o[1]);
  if (al != 0.f) { const float2 xv = p.xt[xo + i]; r.x += al * xv.x; r.y += al * xv.y; }
  p.out[po + i] = r;
}

// ------------------------------------------------------------------------------------------------------
// Sampler element-wise updates (reference sdes.py:224-229, correctors.py:75-79, predictors.py:61-64 over
// sdes.py:72-89,130-135).  Per-step scalars come from a device table built by the host with the reference's own
// fp32 expressions; the step index is read from a device counter so one captured graph serves every step.
enum { SC_T = 0, SC_DT = 1, SC_ALD_EPS = 2, SC_ALD_NOISE = 3, SC_G = 4, SC_G2 = 5, SC_STRIDE = 8 };

__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// complex standard normal (Re, Im ~ N(0, 1/2)), what torch.randn_like gives for complex64
// stream: 64-bit noise-stream id of the utterance; its low word enters the counter (as it always did, so ids below 2^32 keep their
// draws), its high word the key, so ids that differ only above bit 31 (hashed utterance ids) get different streams too
__device__ __forceinline__ float2 philox_cnormal(unsigned long long seed, unsigned long long idx, uint32_t draw, unsigned long long stream64 = 0) {
  const uint32_t stream = (uint32_t)stream64, shi = (uint32_t)(stream64 >> 32);
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32) ^ (stream * 0x9E3779B9u), c2 = draw, c3 = 0x5367534du ^ stream;
  uint32_t k0 = (uint32_t)seed ^ (shi * 0x85EBCA6Bu), k1 = (uint32_t)(seed >> 32) ^ shi;
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  const float u1 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float rad = sqrtf(-logf(u1));  // sqrt(-2 ln u) * sqrt(1/2)
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  return make_float2(rad * cs, rad * sn);
}

struct SamplerArgs {
  float2* x; float2* x_mean; const float2* y; const float2* score;
  const float2* noise;       // replayed noise [ndraws][B*FT] or null (Philox)
  // device words: seed[0] = the Philox seed (data, not a kernel argument, so a captured step serves every seed), seed[1 + b] =
  // the noise-stream id of utterance b.  An utterance's draws depend on (seed, stream id, element index INSIDE the utterance,
  // draw): with stream ids that name the utterance (the directory script: index in the file list) its noise, and with it the
  // enhanced file, does not depend on how the corpus was batched or sharded.  Default stream id: the batch slot.
  const unsigned long long* seed;
  const float* table; const int* step_ptr;
  int draw_base, draw_per_step;  // draw index = draw_base + step*draw_per_step
  float theta, score_w;     // score_w: 1 (reverse SDE) or 0.5 (probability flow)
  int add_noise;
  float std1;               // prior: std(T=1)
  int n;                    // B*FT
  // LangevinCorrector (correctors.py:44-56): step size from batch-mean norms
  float snr; int B; int per;        // per = FT (elements per utterance)
  float* partial;                   // [B][LANG_NBLK][2] partial sums of |grad|^2, |z|^2
  float* lang;                      // [2]: step_size, sqrt(2*step_size)
  const long long* rag_off; // ragged batch: [B + 1] element prefix of the packed utterances (null: B utterances of `per` elements)
};

__device__ __forceinline__ float2 sampler_noise(const SamplerArgs& p, int i, int draw) {
  if (p.noise) return p.noise[(size_t)draw * p.n + i];
  if (p.rag_off) {            // the utterance this element belongs to: last b with rag_off[b] <= i
    int lo = 0, hi = p.B;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.rag_off[mid] <= (long long)i) lo = mid; else hi = mid; }
    return philox_cnormal(p.seed[0], (unsigned long long)((long long)i - p.rag_off[lo]), (uint32_t)draw, p.seed[1 + lo]);
  }
  const int b = i / p.per;
  return philox_cnormal(p.seed[0], (unsigned long long)(i - b * p.per), (uint32_t)draw, p.seed[1 + b]);
}

__global__ __launch_bounds__(256) void sampler_prior_kernel(SamplerArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const float2 z = sampler_noise(p, i, p.draw_base);
  const float2 yv = p.y[i];
  p.x[i] = make_float2(yv.x + z.x * p.std1, yv.y + z.y * p.std1);
}

__global__ __launch_bounds__(256) void sampler_ald_kernel(SamplerArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const int step = *p.step_ptr;
  const float* tb = p.table + (size_t)step * SC_STRIDE;
  const float eps = tb[SC_ALD_EPS], ns = tb[SC_ALD_NOISE];
  const float2 z = sampler_noise(p, i, p.draw_base + step * p.draw_per_step);
  const float2 xv = p.x[i], g = p.score[i];
  const float2 xm = make_float2(xv.x + eps * g.x, xv.y + eps * g.y);
  p.x_mean[i] = xm;
  p.x[i] = make_float2(xm.x + z.x * ns, xm.y + z.y * ns);
}

__global__ __launch_bounds__(256) void sampler_revdiff_kernel(SamplerArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const int step = *p.step_ptr;
  const float* tb = p.table + (size_t)step * SC_STRIDE;
  const float dt = tb[SC_DT], G = tb[SC_G], G2 = tb[SC_G2] * p.score_w;
  const float2 xv = p.x[i], yv = p.y[i], g = p.score[i];
  const float fx = (p.theta * (yv.x - xv.x)) * dt - G2 * g.x;
  const float fy = (p.theta * (yv.y - xv.y)) * dt - G2 * g.y;
  const float2 xm = make_float2(xv.x - fx, xv.y - fy);
  p.x_mean[i] = xm;
  if (p.add_noise) {
    const float2 z = sampler_noise(p, i, p.draw_base + step * p.draw_per_step);
    p.x[i] = make_float2(xm.x + G * z.x, xm.y + G * z.y);
  } else {
    p.x[i] = xm;
  }
}

constexpr int LANG_NBLK = 64;

// pass 1: per-utterance partial sums of |score|^2 and |z|^2.  grid = (LANG_NBLK, B)
__global__ __launch_bounds__(256) void sampler_langevin_norms_kernel(SamplerArgs p) {
  __shared__ float s_a[4];
  __shared__ float s_b[4];
  const int b = blockIdx.y;
  const int step = *p.step_ptr;
  const int draw = p.draw_base + step * p.draw_per_step;
  float g2 = 0.f, z2 = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < p.per; i += LANG_NBLK * 256) {
    const int e = b * p.per + i;
    const float2 g = p.score[e];
    const float2 z = sampler_noise(p, e, draw);
    g2 += g.x * g.x + g.y * g.y;
    z2 += z.x * z.x + z.y * z.y;
  }
  g2 = wave_sum(g2); z2 = wave_sum(z2);
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = g2; s_b[threadIdx.x >> 6] = z2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = p.partial + ((size_t)b * LANG_NBLK + blockIdx.x) * 2;
    o[0] = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
    o[1] = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
  }
}

// pass 2 (one workgroup): grad_norm = mean_b ||score_b||, noise_norm = mean_b ||z_b||, step = 2 (snr noise_norm / grad_norm)^2
__global__ __launch_bounds__(64) void sampler_langevin_scalars_kernel(SamplerArgs p) {
  float gn = 0.f, zn = 0.f;
  for (int b = threadIdx.x; b < p.B; b += 64) {
    double g2 = 0.0, z2 = 0.0;
    for (int k = 0; k < LANG_NBLK; ++k) { g2 += p.partial[((size_t)b * LANG_NBLK + k) * 2]; z2 += p.partial[((size_t)b * LANG_NBLK + k) * 2 + 1]; }
    gn += sqrtf((float)g2); zn += sqrtf((float)z2);
  }
  gn = wave_sum(gn); zn = wave_sum(zn);
  if (threadIdx.x == 0) {
    gn /= (float)p.B; zn /= (float)p.B;
    const float r = p.snr * zn / gn;
    const float stepsz = r * r * 2.f;
    p.lang[0] = stepsz;
    p.lang[1] = sqrtf(stepsz * 2.f);
  }
}

// pass 3: x_mean = x + step*score ; x = x_mean + z*sqrt(2 step)
__global__ __launch_bounds__(256) void sampler_langevin_kernel(SamplerArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const int step = *p.step_ptr;
  const float eps = p.lang[0], ns = p.lang[1];
  const float2 z = sampler_noise(p, i, p.draw_base + step * p.draw_per_step);
  const float2 xv = p.x[i], g = p.score[i];
  const float2 xm = make_float2(xv.x + eps * g.x, xv.y + eps * g.y);
  p.x_mean[i] = xm;
  p.x[i] = make_float2(xm.x + z.x * ns, xm.y + z.y * ns);
}

// Schroedinger-bridge samplers (reference sampling/__init__.py:145-249): every step is the weighted sum
//   x <- w_prev * x + w_est * estimate + w_y * y + w_z * z      (ODE: w_z = 0; SDE: w_y = 0, w_z = 0 on the last step)
// with per-step weights from the device table (columns SB_*), estimate = ScoreModel.forward(x, y, t) (data prediction).
enum { SB_WPREV = 2, SB_WEST = 3, SB_WY = 4, SB_WZ = 5 };

__global__ __launch_bounds__(256) void sampler_sb_kernel(SamplerArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const int step = *p.step_ptr;
  const float* tb = p.table + (size_t)step * SC_STRIDE;
  const float wp = tb[SB_WPREV], we = tb[SB_WEST], wy = tb[SB_WY], wz = tb[SB_WZ];
  const float2 xv = p.x[i], ev = p.score[i], yv = p.y[i];
  // The reference's rounding sequence, product by product and sum by sum (sampling/__init__.py:200-206, 226-233: torch evaluates
  // w_prev*xt, w_est*est, their sum, w_y*y (or w_z*z), the sum), WITHOUT fused multiply-adds.  The ODE's first step is
  // x = 5457.15 y + 0.446 est - 5456.54 y (k = 2.6, c = 0.4, N = 4; x_0 = y): the estimate is added to a number ~5457 |y| and
  // thereby quantised to that number's ulp, 4.6e-4 |y|, before the cancellation.  Following the sequence makes this step a
  // function of `est` alone -- it does NOT bring the sampler closer to the reference's output (measured: 7.3e-5 before and
  // after): two implementations whose network outputs differ by a relative delta disagree by ~sqrt(delta * 0.45 * 4.6e-4) after
  // this step, a property of the reference's algorithm in fp32 (tools/sb_conditioning.py, profiles/r03_sb_conditioning.txt:
  // delta 1e-6 -> 1.2e-5, 1e-5 -> 3.5e-5, 3e-5 -> 6.7e-5; the CPU oracle itself is at 9.6e-6 of the reference).
  float2 r = make_float2(drt_add_rn(drt_add_rn(drt_mul_rn(wp, xv.x), drt_mul_rn(we, ev.x)), drt_mul_rn(wy, yv.x)),
                         drt_add_rn(drt_add_rn(drt_mul_rn(wp, xv.y), drt_mul_rn(we, ev.y)), drt_mul_rn(wy, yv.y)));
  if (p.add_noise) {
    const float2 z = sampler_noise(p, i, p.draw_base + step * p.draw_per_step);
    r.x = drt_add_rn(r.x, drt_mul_rn(wz, z.x)); r.y = drt_add_rn(r.y, drt_mul_rn(wz, z.y));
  }
  p.x[i] = r;
}

// the current step's row of the time-embedding bias table -> a fixed buffer, once per network evaluation: the ~50 convolutions
// that add it then read a known address instead of each chasing the device step counter first (a dependent memory round trip
// in front of every one of them, which a launch of few workgroups -- batch 1 -- cannot hide)
__global__ __launch_bounds__(256) void bias_select_kernel(const float* table, int sstride, const int* step_ptr, float* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = table[(size_t)(*step_ptr) * sstride + i];
}

__global__ void step_set_kernel(int* step, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *step = v; }
__global__ void step_inc_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step = *step + 1; }

}  // namespace sgmse

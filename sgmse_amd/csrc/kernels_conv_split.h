// 3x3 convolution as an fp32-accurate implicit GEMM on the 16-bit matrix pipe: every fp32 operand is split into a few
// 16-bit terms whose partial products (exact in fp32) are accumulated in fp32 by v_mfma_f32_32x32x16_{bf16,f16}.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the 16-bit MFMA rate, and the fp32 3x3 kernel (kernels_conv_pipe.h) already
// sits at ~81 % of that fp32 peak.  Two split policies:
//
//   SplitB3 ("bf16x3"): x = hi + mid + lo EXACTLY, hi = x & 0xffff0000, mid = (x - hi) & 0xffff0000, lo = x - hi - mid
//     (8 + 8 + 8 mantissa bits, truncating).  a*b is accumulated from the six partial products of weight >= 2^-16:
//     hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi; the dropped terms are < 2^-23 relative to a*b, the size of one
//     fp32 rounding.  Six MFMAs of K = 16 (6 x 32 cycles) replace eight fp32 MFMAs of K = 2 (8 x 64): 2.67x the fp32 peak.
//
//   SplitH2 ("fp16x2"): x * 2^s = hi + lo + e, hi = RN_f16(x 2^s), lo = RN_f16(x 2^s - hi), |e| <= 2^-24 |x 2^s| (two
//     round-to-nearest 11-bit terms cover 22-23 bits plus the sign of lo).  a*b from hi*hi + hi*lo + lo*hi (lo*lo <=
//     2^-24): relative error <= ~3 x 2^-24 per product, i.e. within one bit of fp32.  Three MFMAs: 5.3x the fp32 peak.
//     fp16 has a narrow exponent range, so both operands are pre-scaled by exact powers of two: the weights of a layer by
//     2^k with max|w| 2^k in [2^13, 2^14) (chosen by the packing kernel, which also stores 2^-k for the epilogue), the
//     activations -- always the output of the fused GroupNorm(+SiLU) producer here, or its FIR resampling -- PER UTTERANCE
//     by the power of two that puts an upper bound of |producer output| in [2^13, 2^14).  The bound comes from the
//     utterance's own data (gn_finalize_kernel: min(max|x| + |mean|, sqrt(N var)) rstd |gamma| + |beta| over the channels),
//     so no GroupNorm parameter and no utterance length is presumed, and an utterance's scale does not depend on its batch.
//     Small operands fall into fp16 subnormals; their ABSOLUTE error stays <= 2^-25 of the scaled unit, which is what
//     matters for a sum (fp32-equivalent as long as the bound is within 2^14 of the typical magnitude; the sqrt(N var) term
//     keeps it there when single channels or pixels are hot).
//
// Both are checked against an fp64 convolution next to the fp32-MFMA kernel (tests: check_conv_split) and pass the same
// 1e-5 per-op / 2e-5 per-network parity gates as the fp32 kernels.
//
// Layout (one workgroup = 128 co x 256 px = 8 rows x 32 columns, 4 waves, two workgroups per CU):
//   * wave w owns the 32 output channels w of the block and all 8 pixel fragments (acc = 8 x 16 registers);
//   * B operand (pixels): the input tile with halo (10 x 34 px) of a 16-channel K-stage lives in LDS already split,
//     channels fastest: [row][col][k-group of 8 channels][split][8 x 16 bit], so an MFMA B fragment of tap (dy,dx) is
//     one ds_read_b128 per lane and the tap is an address offset.  The pixel stride is padded (PX_V) so that the 16
//     lanes a ds_read_b128 services per LDS cycle hit distinct banks.  Two stages are double buffered; the fused
//     producer (GroupNorm affine + SiLU) and the split run once per staged element;
//   * A operand (weights): packed offline in fragment order [co block][stage][tap][split][wave][lane][8 x 16 bit], so a
//     fragment is one coalesced 16-byte global load per lane (L2-resident), in a ring of three register sets two taps ahead.
//     No LDS.
//   * K order: stage (16 channels) -> tap -> split products, small terms first.  Epilogue shared with the fp32 kernels.
//
// What bounds it (round 2, profiles/r02_power_probe.txt, r02_split_ablation_microbench.txt, r02_split_trace_*.txt): the f16
// matrix pipe on random data drives the socket to its 1400 W power cap, the clock settles at ~1.5 GHz, and the kernel's time is
// its dynamic ENERGY divided by (cap - static power): MFMAs 60 %, the epilogue's HBM traffic 17 %, input staging 12 %, weight
// fragments from L2 8 %.  Stalls are nearly free, bytes and MFMAs are not: structures that only re-order the instruction stream
// (finer MFMA / VALU interleave, a deeper fragment ring, a 2 x 4 register blocking with half the LDS reads, a start-up stagger
// of the workgroups, 4-row tiles at three workgroups per CU) measured within +-2 %; halving the producer's VALU work and an
// epilogue with a fifth of the instructions gained 4 %.
#pragma once
#include "kernels_conv.h"

// scheduling fence between the operand reads and the MFMAs of a pixel fragment: 0 pins everything (0x6, letting VALU /
// SALU instructions move across, changes nothing: the compiler still clusters the producer arithmetic behind the MFMAs)
#ifndef SGMSE_SPLIT_FENCE
#define SGMSE_SPLIT_FENCE 0
#endif

namespace sgmse {

struct SplitB3 {
  static constexpr int NS = 3;          // split terms per operand
  static constexpr int NP = 6;          // partial products
  static constexpr int PX_V = 7;        // u32x4 per staged pixel (2 k-groups x 3 splits = 6, padded: 28 n mod 64 distinct)
  static constexpr bool SCALED = false;
  __device__ static constexpr int pa(int k) { return k == 0 ? 2 : (k == 2 || k == 3) ? 1 : 0; }   // (a term, b term),
  __device__ static constexpr int pb(int k) { return k == 1 ? 2 : (k == 2 || k == 4) ? 1 : 0; }   // small first
  __device__ static __forceinline__ void split(float x, uint32_t (&t)[3]) {   // 16-bit patterns in the low halves
    const uint32_t xb = __builtin_bit_cast(uint32_t, x);
    const uint32_t hi = xb & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, hi);
    const uint32_t mid = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mid);
    t[0] = hi >> 16; t[1] = mid >> 16; t[2] = __builtin_bit_cast(uint32_t, r2) >> 16;
  }
  // two consecutive K-elements -> one dword per split term (element 0 in the low half)
  __device__ static __forceinline__ void split2(float x0, float x1, uint32_t (&d)[3]) {
    uint32_t a[3], b[3];
    split(x0, a); split(x1, b);
#pragma unroll
    for (int s = 0; s < 3; ++s) d[s] = a[s] | (b[s] << 16);
  }
  __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) { return mfma_32x32x16_bf16(a, b, c); }
};

struct SplitH2 {
  static constexpr int NS = 2;
  static constexpr int NP = 3;
  static constexpr int PX_V = 5;        // 2 k-groups x 2 splits = 4, padded: 20 n mod 64 distinct
  static constexpr bool SCALED = true;
  __device__ static constexpr int pa(int k) { return k == 0 ? 1 : 0; }
  __device__ static constexpr int pb(int k) { return k == 1 ? 1 : 0; }
  __device__ static __forceinline__ void split(float x, uint32_t (&t)[2]) {   // x already scaled and saturated
    const uint32_t hi = drt_f32_to_f16(x);
    const float r = x - drt_f16_to_f32(hi);
    t[0] = hi; t[1] = drt_f32_to_f16(r);
  }
  // two consecutive K-elements -> {hi pair, lo pair}: two packed round-to-nearest conversions (v_cvt_pk_f16_f32)
  __device__ static __forceinline__ void split2(float x0, float x1, uint32_t (&d)[2]) {
    const uint32_t hi = drt_f32x2_to_f16x2(x0, x1);
    d[0] = hi; d[1] = drt_f32x2_to_f16x2(drt_sub_f16_lo(x0, hi), drt_sub_f16_hi(x1, hi));      // four instructions per pair
  }
  __device__ static __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) { return mfma_32x32x16_f16(a, b, c); }
};

template <int ROWS_>
struct ConvSplitGeom {
  static constexpr int KC = 16, ROWS = ROWS_, TROWS = ROWS_ + 2, TCOLS = 34;
  static constexpr int NITEM = TROWS * TCOLS * 2;              // (pixel, k-group) staging items per stage
  static constexpr int NIT = (NITEM + 255) / 256;              // per thread (3 for 8 rows, 2 for 4)
};

// Weight packing.  src: OIHW fp32 [Cout][Cin][k][k] (taps = k*k = 9 or 1); dst: u32x4 [nCoBlk][Cin/16][taps][NS][4][64], followed (SplitH2) by
// nCoBlk * 128 floats: PER OUTPUT CHANNEL the factor 2^-k that undoes that channel's weight scale (the input's per-utterance scale is
// undone by the kernel) -- one outlier weight costs the small weights of its own channel their low bits, not the whole layer's.
// `co_scale` (device, SplitH2 only) = the channels' scales 2^k, from split_co_scale_kernel.  One thread per 16-byte fragment element.
struct PackSplitArgs { const float* src; uint32_t* dst; int cin, cout; size_t total; const float* co_scale; int taps; };
// per output channel: scale 2^k with max |w| 2^k in [2^13, 2^14), and its inverse (padded channels: 1)
__global__ __launch_bounds__(256) void split_co_scale_kernel(const float* src, int n_per_co, int cout, int cout_pad, float* inv_scale, float* scale) {
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (co >= cout_pad) return;
  float m = 0.f;
  if (co < cout) for (int i = 0; i < n_per_co; ++i) m = fmaxf(m, fabsf(src[(size_t)co * n_per_co + i]));
  const float sc = co < cout ? h2_weight_scale(m) : 1.f;
  scale[co] = sc; inv_scale[co] = 1.f / sc;
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* x, size_t n, float* out) {   // *out zeroed by the caller
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) drt_atomic_max_nonneg(out, m);
}

template <class S>
__global__ __launch_bounds__(256) void pack_weights_split_kernel(PackSplitArgs p) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.total) return;
  const int lane = (int)(e & 63);
  size_t r = e >> 6;
  const int w = (int)(r & 3); r >>= 2;
  const int split = (int)(r % S::NS); r /= S::NS;
  const int tap = (int)(r % p.taps); r /= p.taps;
  const int nst = p.cin / 16;
  const int st = (int)(r % nst);
  const int blk = (int)(r / nst);
  const int co = blk * 128 + w * 32 + (lane & 31);
  const int c0 = st * 16 + 8 * (lane >> 5);
  const float wscale = (S::SCALED && co < p.cout) ? p.co_scale[co] : 1.f;
  u32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t part[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = c0 + 2 * q + h;
      const float v = (co < p.cout ? p.src[((size_t)co * p.cin + c) * p.taps + tap] : 0.f) * wscale;
      uint32_t t[S::NS];
      S::split(v, t);
      uint32_t sel = t[0];
#pragma unroll
      for (int s = 1; s < S::NS; ++s) sel = split == s ? t[s] : sel;
      part[h] = sel;
    }
    o[q] = part[0] | (part[1] << 16);
  }
  reinterpret_cast<u32x4*>(p.dst)[e] = o;
}

template <class S>
inline size_t packed_split_frags(int cin, int cout, int taps = 9) { return (size_t)((cout + 127) / 128) * (cin / 16) * taps * S::NS * 4 * 64; }
template <class S>
inline size_t packed_split_bytes(int cin, int cout, int taps = 9) { return packed_split_frags<S>(cin, cout, taps) * 16 + (size_t)((cout + 127) / 128) * 128 * 8; }

// Workgroup shapes (all with the same K order, epilogue and per-row GroupNorm partials: bit-identical results):
//   SHAPE 0: the four waves own the four 32-channel fragments of a 128-channel block, each all 8 pixel fragments
//            (128 co x 8 rows x 32 px).
//   SHAPE 1 (Cout <= 32, the C->4 pyramid convolutions): one 32-channel fragment (zero-padded weights), the four waves
//            own two pixel fragments each -- 1/4 of the MFMA work of a full block for the same staged tile; these layers
//            are bound by the producer arithmetic and HBM, not by the matrix pipe.
//   SHAPE 2: as 0 with 4 rows (128 co x 4 rows x 32 px): twice the workgroups of half the duration, for launches that
//            cannot fill the chip (small batches: the per-file loop of enhancement.py).
// ACT: the fused producer applies SiLU after the affine (1) or only the affine (0: raw / FIR-resampled inputs, scale 1).
//
// Fused producer, per staged element (the VALU work of this kernel: it runs beside the MFMA stream and its instruction
// count sets the power the matrix pipe is left with):
//     t  = x * s + h            s, h: GroupNorm scale/shift of the channel, pre-multiplied by the utterance's fp16x2 input scale
//     u  = x * s2 + h2          s2, h2 = -log2(e) * the unscaled (s, h): the exponent of exp(-t) without a dependent multiply
//     o  = t / (1 + 2^u)        v_exp_f32, v_add, v_rcp_f32, v_mul
//     o  = clamp(o, +-65504)    (cannot bind: the scaled bound is < 2^14; keeps the conversion finite for non-finite inputs)
//     hi = f16(o), lo = f16(o - hi) for two elements at a time (v_cvt_pk_f16_f32)
// The four coefficients of a channel are one 16-byte LDS read.  Zero padding: a staged item outside the image never
// changes, so its LDS slots are written once (zeros) and its per-stage stores go to a dummy slot behind the tile.
//
// ABL: measurement-only instantiations reachable from sgmse_bench_conv (profiles/r02_split_ablation.txt), results WRONG on
// purpose: 1 no epilogue stores, 2 no residual read, 4 producer without the transcendental pair, 8 nothing staged after
// the first K-stage (no raw loads, producer, LDS writes), 16 B fragments read from LDS once, 32 A fragments loaded once;
// correct results: 64 phase time stamps per workgroup (ConvArgs::trace), 128 / 256 epilogue variants (conv_epilogue), 512 LDS
// padded to one workgroup per CU.
// (a 2 x 4 register blocking of a wave -- two channel fragments x half of the pixel fragments, half the LDS operand reads --
// measured the same as this 1 x 8 blocking, profiles/r02_conv_microbench_kernel_families.txt; removed in round 4)
// SC = 1: with the folded 1x1 residual shortcut (ConvArgs::sc_*): its K-stages run first on accumulators that start from
// zero; the accumulators are then rescaled by the (exact, power-of-two) ratio of the two operand scalings, receive the
// bias terms, and the 3x3 stages continue on top.
// CHK = 1 (4-row shape, levels with 2-7 tiles per image): chunked accumulation / split-K as in the fp32 kernels
// (ConvArgs::kchunk_stages, stages of 16 channels): the K reduction is the sum, in chunk order, of per-chunk partial sums; one
// workgroup runs all chunks (gridDim.z == 1) or one chunk each (gridDim.z == chunks, raw partial sums to ConvArgs::partial,
// summed and finished by conv_splitk_reduce_kernel<3, 4, 1, 4>) -- bit-identical.  The bias terms are added by the epilogue.
template <class S, int SHAPE = 0, int ACT = 1, int ABL = 0, int SC = 0, int CHK = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_split_kernel(ConvArgs p) {
  static_assert(!CHK || (SHAPE == 2 && !SC), "chunked accumulation exists for the plain 4-row shape");
  constexpr bool THIN = SHAPE == 1;
  constexpr int ROWS = SHAPE == 2 ? 4 : 8;
  using C = ConvSplitGeom<ROWS>;
  constexpr int WCW = THIN ? 1 : 4;                 // waves along the output channels
  constexpr int FCW = THIN ? 1 : 4 / WCW;           // channel fragments per wave
  constexpr int FPW = THIN ? 2 : ROWS / (4 / WCW);  // pixel fragments (image rows) per wave
  // epilogue geometry: WCW channel-waves x FCW fragments, 4 / WCW pixel-waves x FPW fragments
  using T = ConvTile<3, WCW, FCW, FPW, 1>;
  static_assert(T::CO_T == (THIN ? 32 : 128) && T::ROWS == ROWS, "tile");
  constexpr int NS = S::NS, PX_V = S::PX_V;
  constexpr int EPJ = 8 / FPW;             // producer elements staged behind each fragment's MFMAs
  constexpr int STAGE_V = C::TROWS * C::TCOLS * PX_V;
  constexpr int DUMMY_V = STAGE_V;         // scratch slot behind the tile for the stores of items outside the image
  __shared__ u32x4 s_in0[STAGE_V + NS];
  __shared__ u32x4 s_in1[STAGE_V + NS];
  // per input channel {s, h, s2, h2}; the bf16x3 policy keeps {s, h} only (its three split terms per element leave no LDS
  // for the wider table at two workgroups per CU) and derives the exponent with a dependent multiply
  constexpr int NCO = S::SCALED ? 4 : 2;
  __shared__ float s_co[512 * NCO];

  const int tid = threadIdx.x;
  if constexpr (ABL & 512) {          // one workgroup per CU: 48 KB of extra LDS
    __shared__ u32x4 s_pad[3072];
    if (p.ablate == 12345) s_pad[tid] = u32x4{0u, 0u, 0u, 0u};
  }
  unsigned long long* trace = nullptr;
  if constexpr (ABL & 64) {
    if (tid == 0 && p.trace) {
      trace = p.trace + 16 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
      trace[0] = (unsigned long long)drt_hw_id() | ((unsigned long long)drt_xcc_id() << 32);
      trace[1] = drt_clock();
    }
  }
  const int Cin = p.C1 + p.C2;
  const int tiles_xg = (p.W + 31) >> 5;       // grid layout (ragged launches: of the widest utterance)
  const int tiles_y = (p.H + ROWS - 1) / ROWS;
  int b, ty, tx;
  conv_tile_of(p, (int)blockIdx.x, (int)gridDim.x, tiles_xg, tiles_y, b, ty, tx);
  if (p.rag_w) { if (!conv_ragged_adjust(p, b, tx)) return; }
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  const int co_blk = blockIdx.y;
  const int x0 = tx * 32, y0 = ty * ROWS;
  // fp16x2 input scale of THIS utterance: the power of two that puts the bound gn_finalize_kernel derived from the utterance's
  // own statistics (ConvArgs::xbound) in [2^13, 2^14); undone exactly on the accumulator side (inv_kx)
  // Prologue latency (what a launch of few workgroups -- batch 1 -- pays per convolution): every INDEPENDENT global load of the
  // prologue is issued before the first result is waited for -- the range-bound word, this thread's producer coefficients
  // (channels tid and tid + 256; Cin <= 512), the bias / time-embedding terms, the first stage's raw inputs, the first weight
  // fragments -- so that they cost one memory round trip together instead of one each.
  float xb_raw = 0.f;
  if constexpr (S::SCALED) { if (p.xbound) xb_raw = p.xbound[b * kAmaxSpread + (tid & (kAmaxSpread - 1))]; }
  float csc[2], csh[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = tid + 256 * k;
    const bool ld = p.in_scale != nullptr && c < Cin;
    csc[k] = ld ? p.in_scale[b * Cin + (c < Cin ? c : 0)] : 1.f;
    csh[k] = ld ? p.in_shift[b * Cin + (c < Cin ? c : 0)] : 0.f;
  }
  const unsigned HW = (unsigned)H * (unsigned)W;

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;

  // staging items of this thread: (k-group g, tile row r, tile column c), c fastest so that a wave's loads of one
  // channel are consecutive pixels.  Items past the end repeat the last one (same address, same value).
  unsigned it_boff[C::NIT];                // byte offset of (channel 8 g, pixel) inside a 16-channel slab of the source
  int it_loff[C::NIT], it_g[C::NIT];
#pragma unroll
  for (int i = 0; i < C::NIT; ++i) {
    int it = tid + 256 * i;
    it = it < C::NITEM ? it : C::NITEM - 1;
    const int g = it / (C::TROWS * C::TCOLS);
    const int rem = it - g * (C::TROWS * C::TCOLS);
    const int r = rem / C::TCOLS, c = rem - r * C::TCOLS;
    const int gy = y0 - 1 + r, gx = x0 - 1 + c;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const int loff = (r * C::TCOLS + c) * PX_V + g * NS;     // in u32x4 units
    if (!ok) {                                                // zero padding applies to the producer's OUTPUT: written once
#pragma unroll
      for (int s = 0; s < NS; ++s) { s_in0[loff + s] = u32x4{0u, 0u, 0u, 0u}; s_in1[loff + s] = u32x4{0u, 0u, 0u, 0u}; }
    }
    it_boff[i] = ((unsigned)(8 * g) * HW + (ok ? (unsigned)(gy * W + gx) : 0u)) * 4u;
    it_loff[i] = ok ? loff : DUMMY_V;
    it_g[i] = g;
  }

  float rin[C::NIT][8];
  auto load_item = [&](int i, int c0) {    // raw global loads of one staged item of stage c0 (consumed >= 3 taps later)
    const bool first = c0 < p.C1;
    // uniform slab pointer + 32-bit lane offset: global_load with an SGPR base, no per-load address arithmetic
    const float* base = first ? p.src1 + ((size_t)b * p.C1 + c0) * HW : p.src2 + ((size_t)b * p.C2 + (c0 - p.C1)) * HW;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      rin[i][e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (size_t)e * HW) + it_boff[i]);
  };
  // producer of one element -> the value that is split (already scaled)
  auto produce = [&](float x, int ch) -> float {
    float o, u;
    if constexpr (NCO == 4) {
      const f32x4 co = reinterpret_cast<const f32x4*>(s_co)[ch];
      o = x * co[0] + co[1];
      u = x * co[2] + co[3];
    } else {
      o = x * s_co[2 * ch] + s_co[2 * ch + 1];
      u = o * -1.4426950408889634f;
    }
    if constexpr (ACT == 1) {
      if constexpr (ABL & 4) o = o * u;
      else o = o * __builtin_amdgcn_rcpf(1.0f + drt_exp2(u));       // SiLU: t * sigmoid(t)
    }
    if (S::SCALED) o = fminf(fmaxf(o, -65504.f), 65504.f);
    return o;
  };
  auto store_item = [&](int i, int c0, u32x4* sbuf) {   // producer + split + LDS write of one staged item (prologue)
    u32x4 v[NS];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = c0 + 8 * it_g[i] + 2 * q;
      uint32_t d[NS];
      S::split2(produce(rin[i][2 * q], ch), produce(rin[i][2 * q + 1], ch + 1), d);
#pragma unroll
      for (int s = 0; s < NS; ++s) v[s][q] = d[s];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) sbuf[it_loff[i] + s] = v[s];
  };

  // the same, one ELEMENT at a time: the eight elements of an item are spread over the pixel fragments of a tap (behind
  // a fragment's MFMAs and inside its scheduling fences), so that the producer arithmetic issues while the MFMA pipe
  // works instead of after the tap; the split runs on every second element (packed conversions)
  u32x4 pk[NS];
  float ev = 0.f;
  auto stage_elem = [&](int i, int e, int c0, const float (&raw)[C::NIT][8]) {
    const float o = produce(raw[i][e], c0 + 8 * it_g[i] + e);
    if ((e & 1) == 0) { ev = o; return; }
    uint32_t d[NS];
    S::split2(ev, o, d);
#pragma unroll
    for (int s = 0; s < NS; ++s) pk[s][e >> 1] = d[s];
  };
  auto flush_item = [&](int i, u32x4* sbuf) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sbuf[it_loff[i] + s] = pk[s];
  };

  const int wc = THIN ? 0 : wave % WCW, wp = THIN ? wave : wave / WCW;     // this wave's channel / pixel position in the block
  const int nst = Cin / C::KC;
  // this workgroup's K-stages: all of them, or (split-K) one chunk
  int st0 = 0, st1 = nst;
  if constexpr (CHK) {
    if (gridDim.z > 1) { st0 = (int)blockIdx.z * p.kchunk_stages; st1 = st0 + p.kchunk_stages < nst ? st0 + p.kchunk_stages : nst; }
  }
  // accumulators start from (bias + time-embedding row) / acc_scale: no bias work in the epilogue.  Raw terms now (loads),
  // scaled once the input scale is known.
  float acc_raw[(SC || CHK) ? 1 : FCW][16];
  if constexpr (!(SC || CHK)) {
#pragma unroll
    for (int i = 0; i < FCW; ++i) conv_acc_raw<T>(p, b, co_blk, wc * FCW + i, kg, acc_raw[i]);
  }
  float sc_m1 = 0.f, sc_m2 = 0.f;
  if constexpr (SC) {
    sc_m1 = p.sc_amax1[b * kAmaxSpread + (tid & (kAmaxSpread - 1))];
    if (p.sc_amax2) sc_m2 = p.sc_amax2[b * kAmaxSpread + (tid & (kAmaxSpread - 1))];
  } else {
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) load_item(i, st0 * C::KC);      // first stage's raw inputs (into s_in0 below)
  }
  // fp16x2 input scale of THIS utterance: the power of two that puts the bound gn_finalize_kernel derived from the utterance's
  // own statistics (ConvArgs::xbound) in [2^13, 2^14); undone exactly on the accumulator side (inv_kx)
  float kx = 1.f;
  if constexpr (S::SCALED) {
    if (p.xbound) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) xb_raw = fmaxf(xb_raw, __shfl_xor(xb_raw, o));
      kx = h2_weight_scale(xb_raw);
    }
  }
  const float inv_kx = 1.f / kx;
  {
    constexpr float nl2e = -1.4426950408889634f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = tid + 256 * k;
      if (c < Cin) {
        if constexpr (NCO == 4) {
          f32x4 v;
          v[0] = csc[k] * kx; v[1] = csh[k] * kx; v[2] = csc[k] * nl2e; v[3] = csh[k] * nl2e;
          reinterpret_cast<f32x4*>(s_co)[c] = v;
        } else {
          s_co[2 * c] = csc[k] * kx; s_co[2 * c + 1] = csh[k] * kx;
        }
      }
    }
  }
  f32x16 acc[FCW][FPW];
#pragma unroll
  for (int i = 0; i < FCW; ++i) {
    float init[16];
    if constexpr (SC || CHK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) init[r] = 0.f;
    } else {
      const int co_l = co_blk * T::CO_T + (wc * FCW + i) * 32 + 4 * kg;
#pragma unroll
      for (int r = 0; r < 16; ++r) init[r] = acc_raw[i][r] * (1.0f / (conv_as(p, co_l + (r & 3) + 8 * (r >> 2)) * inv_kx));      // (powers of two: exact)
    }
#pragma unroll
    for (int j = 0; j < FPW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = init[r];
  }

  // A fragments of this wave: [co_blk][stage][tap][split][wave][lane] (THIN: every wave uses fragment 0); uniform
  // fragment pointer + lane byte offset
  const u32x4* wblk = reinterpret_cast<const u32x4*>(p.w) + (size_t)co_blk * nst * 9 * NS * 4 * 64;
  const unsigned a_boff = (unsigned)((wc * FCW * 64 + lane) * 16);
  // THIN: the fragment is the layer's Cout (4) real output channels zero-padded to 32, and all four waves need the same one.
  // Loading the padding costs L2 bandwidth like real data -- 72 KB of fragments per workgroup and stage against 22 KB of staged
  // input, 9.4 GB per full-resolution launch at batch 32 -- so only the lanes that hold a real channel load (exec-masked: no
  // traffic for the others), the rest keep zeros.  (Less traffic, same time: see the thin loop below.)
  const bool a_real = !THIN || l31 < p.Cout;
  auto load_a = [&](int st, int tap, u32x4 (&a)[FCW][NS]) {
    const u32x4* q = wblk + (size_t)(st * 9 + tap) * NS * 4 * 64;
#pragma unroll
    for (int i = 0; i < FCW; ++i)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if constexpr (THIN) {
          a[i][s] = u32x4{0u, 0u, 0u, 0u};
          if (a_real) a[i][s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(q + s * 4 * 64 + i * 64) + a_boff);
        } else {
          a[i][s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(q + s * 4 * 64 + i * 64) + a_boff);
        }
      }
  };
  // B fragment base of this lane inside a stage buffer (u32x4 units): pixel (row j + dy, col l31 + dx), k-group kg
  const int b_lane = l31 * PX_V + kg * NS + wp * FPW * C::TCOLS * PX_V;

  // one tap of one stage: FPW pixel fragments x NP split products.  The B reads of fragment j+1 are issued before the
  // MFMAs of fragment j (order pinned with sched_barrier: left alone, the compiler sinks every LDS read to just in front
  // of its first use and waits lgkmcnt(0) for each).
  u32x4 bq[2][NS];
  auto compute_tap = [&](const u32x4* sbuf, int tap, const u32x4 (&a)[FCW][NS], int item, int c0n, u32x4* nxt, const float (&raw)[C::NIT][8]) {
    const int dy = tap / 3, dx = tap - 3 * dy;
    const u32x4* sb = sbuf + b_lane + (dy * C::TCOLS + dx) * PX_V;
    if constexpr (!(ABL & 16)) {
#pragma unroll
      for (int s = 0; s < NS; ++s) bq[0][s] = sb[s];
    }
#pragma unroll
    for (int j = 0; j < FPW; ++j) {
      if constexpr (!(ABL & 16)) {
        if (j + 1 < FPW) {
          const u32x4* q = sb + (j + 1) * C::TCOLS * PX_V;
#pragma unroll
          for (int s = 0; s < NS; ++s) bq[(j + 1) & 1][s] = q[s];
        }
      }
      __builtin_amdgcn_sched_barrier(SGMSE_SPLIT_FENCE);
#pragma unroll
      for (int k = 0; k < S::NP; ++k)
#pragma unroll
        for (int i = 0; i < FCW; ++i) acc[i][j] = S::mfma(a[i][S::pa(k)], bq[j & 1][S::pb(k)], acc[i][j]);
      if (item >= 0) {
#pragma unroll
        for (int e = 0; e < EPJ; ++e) stage_elem(item, j * EPJ + e, c0n, raw);
        if (j == FPW - 1) flush_item(item, nxt);
      }
      __builtin_amdgcn_sched_barrier(SGMSE_SPLIT_FENCE);
    }
  };

  if constexpr (SC) {
    // ---- folded 1x1 shortcut: K-stages of 16 channels of the raw block input, centre tap only -------------------------------
    static_assert(S::SCALED && !THIN, "the folded shortcut exists for the fp16x2 full-block shapes");
    float m = fmaxf(sc_m1, sc_m2);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float xs = h2_weight_scale(m);                 // max |x| 2^s in [2^13, 2^14) for this utterance
    const int nsts = (p.sc_C1 + p.sc_C2) / C::KC;
    // raw inputs two stages ahead in two register sets: a stage has only 24 MFMAs per wave to cover its loads
    float rsc[C::NIT][8];
    auto load_sc = [&](int c0, float (&dst)[C::NIT][8]) {
      const bool first = c0 < p.sc_C1;
      const float* base = first ? p.sc_src1 + ((size_t)b * p.sc_C1 + c0) * HW : p.sc_src2 + ((size_t)b * p.sc_C2 + (c0 - p.sc_C1)) * HW;
#pragma unroll
      for (int i = 0; i < C::NIT; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          dst[i][e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (size_t)e * HW) + it_boff[i]);
    };
    auto store_sc = [&](const float (&src)[C::NIT][8], u32x4* sbuf) {
#pragma unroll
      for (int i = 0; i < C::NIT; ++i) {
        u32x4 v[NS];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t d[NS];
          S::split2(fminf(fmaxf(src[i][2 * q] * xs, -65504.f), 65504.f), fminf(fmaxf(src[i][2 * q + 1] * xs, -65504.f), 65504.f), d);
#pragma unroll
          for (int s = 0; s < NS; ++s) v[s][q] = d[s];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) sbuf[it_loff[i] + s] = v[s];
      }
    };
    const u32x4* wsc = reinterpret_cast<const u32x4*>(p.sc_w) + (size_t)co_blk * nsts * NS * 4 * 64;
    auto load_asc = [&](int st, u32x4 (&a)[FCW][NS]) {
      const u32x4* q = wsc + (size_t)st * NS * 4 * 64;
#pragma unroll
      for (int i = 0; i < FCW; ++i)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          a[i][s] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(q + s * 4 * 64 + i * 64) + a_boff);
    };
    load_sc(0, rin);
    load_sc((nsts > 1 ? 1 : 0) * C::KC, rsc);
    __syncthreads();          // the zero padding visible
    store_sc(rin, s_in0);
    __syncthreads();
    u32x4 asc[FCW][NS];
    // stage st: MFMAs from buffer st & 1, stage st + 1 (already in registers) through the producer into the other buffer, stage
    // st + 2 loaded into the register set that just became free.  Two stages per iteration: the register sets alternate.
#pragma unroll 1
    for (int st = 0; st < nsts; st += 2) {
      load_asc(st, asc);
      if (st + 2 < nsts) load_sc((st + 2) * C::KC, rin);
      compute_tap(s_in0, 4, asc, -1, 0, s_in1, rin);
      if (st + 1 < nsts) store_sc(rsc, s_in1);
      __syncthreads();
      if (st + 1 < nsts) {
        load_asc(st + 1, asc);
        if (st + 3 < nsts) load_sc((st + 3) * C::KC, rsc);
        compute_tap(s_in1, 4, asc, -1, 0, s_in0, rin);
        if (st + 2 < nsts) store_sc(rin, s_in0);
        __syncthreads();
      }
    }
    // accumulator: from the shortcut's operand scaling (weights 2^k1, input xs) to the 3x3 stages' (acc_scale), plus the biases
#pragma unroll
    for (int i = 0; i < FCW; ++i) {
      float init[16], rho[16];
      conv_acc_init<T>(p, b, co_blk, wc * FCW + i, kg, inv_kx, init);
      const int co_l = co_blk * T::CO_T + (wc * FCW + i) * 32 + 4 * kg;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_l + (r & 3) + 8 * (r >> 2);
        const float as3 = conv_as(p, co) * inv_kx;                 // per output channel: the two layers' weight scales are per channel
        rho[r] = (p.sc_scale[co] / xs) / as3;
        init[r] += p.sc_bias ? p.sc_bias[co] / as3 : 0.f;
      }
#pragma unroll
      for (int j = 0; j < FPW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] * rho[r] + init[r];
    }
  }
  // prologue: first stage -> s_in0 (its raw loads were issued at the top; behind a folded shortcut they are issued here)
  if constexpr (SC) {
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) load_item(i, st0 * C::KC);
  }
  __syncthreads();          // s_co and the zero padding visible
#pragma unroll
  for (int i = 0; i < C::NIT; ++i) store_item(i, st0 * C::KC, s_in0);
  __syncthreads();

  if constexpr (ABL & 64) { if (trace) trace[2] = drt_clock(); }
  // A fragments: a ring of three register sets, loaded TWO taps (48 MFMAs of this wave, ~2 us of a shared matrix pipe) ahead.
  // Every wave of the CU shares one vector-memory queue: a fragment load (an L2 hit) regularly lands behind another wave's
  // HBM misses (raw inputs of the next stage, the co-resident workgroup's residual reads), and one tap of cover exposed that
  // latency on every tap (profiles/r02_split_ablation_microbench.txt: staging, fragment loads and the epilogue's memory
  // traffic each ADDED their time to the K loop).  9 taps % 3 == 0: the ring index of a tap is the same in every stage.
  unsigned long long tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // ABL 64: time per tap position and at the stage barrier (wave 0)
  // (a ring holding a whole stage -- 9 sets, 8 taps ahead -- for the 4-row shape, whose taps have half the MFMAs to cover a
  // load, measured no gain at batch 1 and costs the third workgroup per CU: profiles/r02_chunk_splitk.txt)
  if constexpr (THIN && !ABL) {
    // ---- thin shape (C -> 4 pyramid convolutions): a tap is only 6 MFMAs per wave (~190 cycles), so the fragment ring above (two
    // taps of cover, and vmcnt retires in order: a fragment wait also waits for the raw HBM loads issued in front of it) is
    // replaced by: the fragments of a WHOLE stage in registers (9 x 8 VGPRs), the fragment of tap t reloaded for the next stage
    // right after tap t's MFMAs (a full stage of cover), the raw inputs TWO stages ahead in two register sets, an item reloaded as
    // soon as the producer has consumed it.  Same K order, same arithmetic: bit-identical to the ring loop.
    // MEASURED (round 3, profiles/r03_prof_dump_b32.txt): neither this nor the exec-masked fragment loads (load_a) moved these
    // layers -- 128 -> 4 @ 32 x 256 x 512 stays at 1.41 ms.  They are not bound by memory latency or L2 bandwidth but by issue
    // slots: a workgroup stages the same 10 x 34 x 16 tile as a full 128-channel block (producer: ~66 wave-cycles per element,
    // 24 elements per thread and stage ~ 1600 cycles) and has only its 54 MFMAs per wave and stage (~1730 cycles) to put beside
    // it; 2 waves per SIMD x (1600 + 1730) = 6.7 k cycles per stage pair against 11 k measured.  What would help is less producer
    // work per output (the halo is 1.33 x) or a 16-row MFMA for the 4 real channels -- another kernel, for 1.9 % of the evaluation.
    static_assert(FCW == 1, "thin shape");
    float rin2[C::NIT][8];
    u32x4 af[9][FCW][NS];
#pragma unroll
    for (int t = 0; t < 9; ++t) load_a(0, t, af[t]);
    auto load_raw = [&](int i, int c0, float (&dst)[C::NIT][8]) {
      const bool first = c0 < p.C1;
      const float* base = first ? p.src1 + ((size_t)b * p.C1 + c0) * HW : p.src2 + ((size_t)b * p.C2 + (c0 - p.C1)) * HW;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        dst[i][e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (size_t)e * HW) + it_boff[i]);
    };
    auto clampst = [&](int st) { return st < nst ? st : nst - 1; };      // past the end: re-stage the last stage (never read)
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) { load_raw(i, clampst(1) * C::KC, rin); load_raw(i, clampst(2) * C::KC, rin2); }
    auto thin_stage = [&](int st, const u32x4* cur, u32x4* nxt, float (&raw)[C::NIT][8]) {
      const int stn = clampst(st + 1), str = clampst(st + 3);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int item = (tap >= 4 && (tap & 1) == 0 && (tap - 4) / 2 < C::NIT) ? (tap - 4) / 2 : -1;
        compute_tap(cur, tap, af[tap], item, stn * C::KC, nxt, raw);
        load_a(stn, tap, af[tap]);
        if (item >= 0) load_raw(item, str * C::KC, raw);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
#pragma unroll 1
    for (int st = 0; st < nst; st += 2) {
      thin_stage(st, s_in0, s_in1, rin);
      if (st + 1 < nst) thin_stage(st + 1, s_in1, s_in0, rin2);
    }
    conv_epilogue<T, FCW, FPW, WCW, 0, true, false, S::SCALED>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kg, inv_kx);
      return;
  }
  constexpr int AR = 3, AD = AR - 1;
  u32x4 ar[AR][FCW][NS];
#pragma unroll
  for (int t = 0; t < AD; ++t) load_a(st0, t, ar[t]);
  f32x16 tot[CHK ? FCW : 1][CHK ? FPW : 1];      // chunked accumulation: running total of the finished chunks
  bool first_chunk = true;
  if constexpr (ABL & 16) {
#pragma unroll
    for (int s = 0; s < NS; ++s) { bq[0][s] = s_in0[b_lane + s]; bq[1][s] = s_in0[b_lane + C::TCOLS * PX_V + s]; }
  }
  if constexpr (ABL & 32) load_a(0, 2, ar[2]);     // (ABL 32: ring of 3, all loaded once)
#pragma unroll 1
  for (int st = st0; st < st1; ++st) {
    // the stage after this one, clamped: the last stage re-stages itself into the buffer nobody reads again, which
    // keeps the loop body free of branches
    const int stn = st + 1 < st1 ? st + 1 : st;
    const u32x4* cur = ((st - st0) & 1) ? s_in1 : s_in0;
    u32x4* nxt = ((st - st0) & 1) ? s_in0 : s_in1;
    // Per tap: the A fragments of the tap after next first (vmcnt retires in order: they must be OLDER than the raw HBM
    // loads of the next stage issued behind them, or every tap would wait for HBM), then one item of raw loads (taps 0-2);
    // producer + split + LDS write of those items three or more taps later (taps 4, 6, 8).
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      unsigned long long tc0 = 0;
      if constexpr (ABL & 64) tc0 = drt_clock();
      const int ntap = (tap + AD) % 9;
      const int nstg = tap + AD < 9 ? st : stn;
      if constexpr (!(ABL & 32)) load_a(nstg, ntap, ar[(tap + AD) % AR]);
      if constexpr (!(ABL & 8)) { if (tap < C::NIT) load_item(tap, stn * C::KC); }
      __builtin_amdgcn_sched_barrier(0);
      int item = (tap >= 4 && (tap & 1) == 0 && (tap - 4) / 2 < C::NIT) ? (tap - 4) / 2 : -1;   // taps 4, 6, 8: items 0, 1, 2
      if constexpr (ABL & 8) item = -1;
      compute_tap(cur, tap, ar[tap % AR], item, stn * C::KC, nxt, rin);
      if constexpr (ABL & 64) { __builtin_amdgcn_sched_barrier(0); tsum[tap] += drt_clock() - tc0; }
    }
    if constexpr (ABL & 64) {
      const unsigned long long tb = drt_clock();
      __syncthreads();
      tsum[9] += drt_clock() - tb;
    } else {
      __syncthreads();
    }
    if constexpr (CHK) {
      if (gridDim.z == 1 && ((st + 1) % p.kchunk_stages == 0 || st + 1 == st1)) {      // a chunk is complete: total += chunk
#pragma unroll
        for (int i = 0; i < FCW; ++i)
#pragma unroll
          for (int j = 0; j < FPW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[i][j][r] = first_chunk ? acc[i][j][r] : tot[i][j][r] + acc[i][j][r]; acc[i][j][r] = 0.f; }
        first_chunk = false;
      }
    }
  }
  if constexpr (CHK) {
    if (gridDim.z == 1) {
#pragma unroll
      for (int i = 0; i < FCW; ++i)
#pragma unroll
        for (int j = 0; j < FPW; ++j) acc[i][j] = tot[i][j];
    } else {                                        // raw partial sums of this chunk; the reduce kernel runs the epilogue
      ConvArgs q = p;
      q.out = p.partial + (size_t)blockIdx.z * conv_partial_slab(p, H, W);
      q.bias = nullptr; q.bias2 = nullptr; q.res = nullptr; q.acc_scale = nullptr; q.co_scale = nullptr; q.out_scale = 1.f; q.stats_out = nullptr; q.amax_out = nullptr;
      conv_epilogue<T, FCW, FPW, WCW, 0, true>(q, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kg);
      // (conv_splitk_reduce_kernel follows)
      return;
    }
  }
  if constexpr (ABL & 64) {
    if (trace) {
#pragma unroll
      for (int i = 0; i < 10; ++i) trace[5 + i] = tsum[i];
    }
  }

  if constexpr (ABL & 64) { if (trace) trace[3] = drt_clock(); }
  conv_epilogue<T, FCW, FPW, WCW, ABL & (3 | 128 | 256), !CHK, FPW % 4 == 0, S::SCALED>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kg, inv_kx);
  if constexpr (ABL & 64) { if (trace) trace[4] = drt_clock(); }
}

// ---------------------------------------------------------------------------------------------------------------------
// 1x1 convolution (residual shortcuts `Conv_2`, NIN; reference layerspp.py:241-243, layers.py:537-555) with the same
// operand split.  Same workgroup shape and operand paths as the 3x3 kernel (wave = 32 output channels x 8 pixel
// fragments, B through LDS, A from global), but a K-stage is only one MFMA K-step deep (16 channels, no taps), so the
// global loads of a stage are issued TWO stages ahead (register sets rinA / rinB) and the producer + split of stage
// s+1 is spread over the pixel fragments of stage s.  Inputs of these layers are raw residual-stream activations of
// unknown range: bf16x3 needs no range; fp16x2 scales by an exact power of two derived per utterance from the range
// bounds the producing kernels left behind (ConvArgs::amax1/amax2), so that max|x| 2^s lies in [2^13, 2^14).
template <class S>
__global__ __launch_bounds__(256, 2) void conv1x1_split_kernel(ConvArgs p) {
  using T = ConvTile<1, 4, 1, 8, 1>;
  static_assert(T::CO_T == 128 && T::ROWS == 8, "tile");
  constexpr int NS = S::NS, PX_V = S::PX_V, KC = 16, NIT = 2;     // 256 px x 2 k-groups = 512 items = 2 per thread
  constexpr int STAGE_V = 256 * PX_V;
  __shared__ u32x4 s_in0[STAGE_V];
  __shared__ u32x4 s_in1[STAGE_V];
  __shared__ float s_sc[512];
  __shared__ float s_sh[512];

  const int tid = threadIdx.x;
  const int Cin = p.C1 + p.C2;
  const int tiles_xg = (p.W + 31) >> 5;       // grid layout (ragged launches: of the widest utterance)
  const int tiles_y = (p.H + 7) >> 3;
  int b, ty, tx;
  conv_tile_of(p, (int)blockIdx.x, (int)gridDim.x, tiles_xg, tiles_y, b, ty, tx);
  if (p.rag_w) { if (!conv_ragged_adjust(p, b, tx)) return; }      // (no GroupNorm tail in this kernel: its launches in the network emit no statistics)
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  const int co_blk = blockIdx.y;
  const int x0 = tx * 32, y0 = ty * 8;
  const bool xform = p.in_scale != nullptr;
  for (int c = tid; c < Cin; c += 256) {
    s_sc[c] = xform ? p.in_scale[b * Cin + c] : 1.f;
    s_sh[c] = xform ? p.in_shift[b * Cin + c] : 0.f;
  }
  const float actf = (xform && p.in_act) ? 1.f : 0.f;
  const size_t HW = (size_t)H * W;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kg = lane >> 5;
  // fp16x2 on a raw (residual-stream) input: exact power-of-two scale from the producers' per-utterance range bounds, so
  // that max |x| 2^s lies in [2^13, 2^14) whatever the range of the stream is
  float xs = 1.f;
  if (S::SCALED) {
    float m = p.amax1 ? amax_read(p.amax1, b) : 0.f;
    if (p.amax2) m = fmaxf(m, amax_read(p.amax2, b));
    xs = h2_weight_scale(m);
  }

  // staging items: (k-group g, row r, column c), c fastest; pixels outside the image are clamped (their outputs are
  // never stored and a 1x1 convolution does not mix pixels)
  int it_goff[NIT], it_loff[NIT], it_g[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int it = tid + 256 * i;
    const int g = it >> 8, r = (it >> 5) & 7, c = it & 31;
    int gy = y0 + r, gx = x0 + c;
    gy = gy < H ? gy : H - 1; gx = gx < W ? gx : W - 1;
    it_goff[i] = gy * W + gx;
    it_loff[i] = (r * 32 + c) * PX_V + g * NS;
    it_g[i] = g;
  }
  float rinA[NIT][8], rinB[NIT][8];
  auto load_items = [&](int c0, float (&dst)[NIT][8]) {
    const bool first = c0 < p.C1;
    const float* base = first ? p.src1 + ((size_t)b * p.C1 + c0) * HW : p.src2 + ((size_t)b * p.C2 + (c0 - p.C1)) * HW;
#pragma unroll
    for (int i = 0; i < NIT; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[i][e] = base[(size_t)(8 * it_g[i] + e) * HW + it_goff[i]];
  };
  u32x4 pk[NS];
  uint32_t ev[NS];
  auto stage_elem = [&](int i, int e, int c0, const float (&src)[NIT][8]) {
    const int ch = c0 + 8 * it_g[i] + e;
    float t = src[i][e] * s_sc[ch] + s_sh[ch];
    const float sig = __builtin_amdgcn_rcpf(1.0f + __expf(-t));
    t *= actf * (sig - 1.0f) + 1.0f;
    if (S::SCALED) t = fminf(fmaxf(t * xs, -65504.f), 65504.f);    // the clamp cannot bind when the range bound holds
    uint32_t t16[NS];
    S::split(t, t16);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if ((e & 1) == 0) ev[s] = t16[s]; else pk[s][e >> 1] = ev[s] | (t16[s] << 16);
    }
  };
  auto flush_item = [&](int i, u32x4* sbuf) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sbuf[it_loff[i] + s] = pk[s];
  };

  f32x16 acc[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  const int nst = Cin / KC;
  const u32x4* wbase = reinterpret_cast<const u32x4*>(p.w) + (size_t)co_blk * nst * NS * 4 * 64 + wave * 64 + lane;
  auto load_a = [&](int st, u32x4 (&a)[NS]) {
    const u32x4* q = wbase + (size_t)st * NS * 4 * 64;
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = q[s * 4 * 64];
  };
  const int b_lane = l31 * PX_V + kg * NS;

  // prologue: stage 0 -> LDS, stage 1 -> rinA
  load_items(0, rinA);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
#pragma unroll
    for (int e = 0; e < 8; ++e) stage_elem(i, e, 0, rinA);
    flush_item(i, s_in0);
  }
  load_items((nst > 1 ? 1 : 0) * KC, rinA);
  u32x4 a0[NS], a1[NS];
  load_a(0, a0);
  __syncthreads();

#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int st1 = st + 1 < nst ? st + 1 : st, st2 = st + 2 < nst ? st + 2 : nst - 1;   // clamped: branch-free tail
    const u32x4* cur = (st & 1) ? s_in1 : s_in0;
    u32x4* nxt = (st & 1) ? s_in0 : s_in1;
    load_a(st1, a1);                    // older than the HBM loads below (vmcnt retires in order)
    load_items(st2 * KC, rinB);
    __builtin_amdgcn_sched_barrier(0);
    const u32x4* sb = cur + b_lane;
    u32x4 bq[2][NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) bq[0][s] = sb[s];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j + 1 < 8) {
        const u32x4* q = sb + (j + 1) * 32 * PX_V;
#pragma unroll
        for (int s = 0; s < NS; ++s) bq[(j + 1) & 1][s] = q[s];
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x16 c = acc[0][j];
#pragma unroll
      for (int k = 0; k < S::NP; ++k) c = S::mfma(a0[S::pa(k)], bq[j & 1][S::pb(k)], c);
      acc[0][j] = c;
      // stage st+1 (raw values in rinA): item j / 4, elements 2 (j % 4) and 2 (j % 4) + 1
      stage_elem(j >> 2, 2 * (j & 3), st1 * KC, rinA);
      stage_elem(j >> 2, 2 * (j & 3) + 1, st1 * KC, rinA);
      if ((j & 3) == 3) flush_item(j >> 2, nxt);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) a0[s] = a1[s];
#pragma unroll
    for (int i = 0; i < NIT; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) rinA[i][e] = rinB[i][e];
  }

  conv_epilogue<T, 1, 8, 4, 0, false, true, S::SCALED>(p, acc, b, co_blk, tx, ty, tiles_x, wave, 0, l31, kg, 1.0f / xs);
}

}  // namespace sgmse

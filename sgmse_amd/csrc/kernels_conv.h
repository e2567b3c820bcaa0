// Convolution kernels of the NCSN++ score network for gfx950.
//
// Replaces F.conv2d as reached through ddpm_conv3x3 / ddpm_conv1x1 (reference
// sgmse/backbones/ncsnpp_utils/layers.py:100-124), NIN (layers.py:546-555), Combine (layerspp.py:52-57),
// and removes torch.cat (ncsnpp.py:350) by reading two source tensors.
//
// conv_mfma: implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fmaf chain).
//   GEMM view per workgroup:  D[co][px] = sum_{c,tap} Wp[c][tap][co] * X[c][px shifted by tap]
//   A operand (rows = output channels) comes from a packed weight chunk staged in LDS,
//   B operand (cols = 32 consecutive pixels of one image row) from an LDS input tile with a 1-pixel halo;
//   a 3x3 tap is an immediate LDS offset, so there is no im2col.  D's lane -> column map puts 32 consecutive
//   pixels on 32 consecutive lanes, so the epilogue (bias, time-embedding bias, residual, 1/sqrt2) stores and
//   residual loads are 128-byte coalesced rows of the NCHW tensor.
//   Fused producer: GroupNorm affine (+SiLU) applied while the tile is written to LDS (layerspp.py:243,264:
//   act(GroupNorm(x)) feeding Conv_0 / Conv_1), zero padding applied after it.
//   Fused consumer: per-channel {sum, sum of squares} of the stored output for the NEXT GroupNorm, reduced with
//   wave shuffles and written as deterministic per-tile partials (no atomics).
// conv_direct: VALU direct convolution for the thin 4->C layers and any shape the MFMA kernel does not cover;
//   same fused producer/epilogue.
//
// Structure measured on MI355X (profiles/r01_conv_microbench_*.txt, r01_pmc_conv_microbench.json): 128 co x 256 px
// tile, 8-channel K-stages, next stage prefetched into registers during the MFMAs, two workgroups per CU.  Tried and
// rejected (slower or equal): one workgroup per CU with a 512-register budget (-10 %), two half-size LDS stages with one
// barrier per stage (-8 %, spills), 4 MFMA waves + 4 staging waves per workgroup (-17 %: one workgroup per CU exposes
// every prologue/epilogue), iglp_opt(0) / s_setprio (0 %), start-up stagger of the second residency slot (0 %); finishing a launch's dependent step
// (GroupNorm coefficients, split-K reduce) in the last workgroup to arrive (+4-20 %, profiles/r03_arrive_last_ab.txt; removed in round 4).
#pragma once
#include <sgmse_devrt.h>
#include <type_traits>

namespace sgmse {

// Per-utterance geometry of one U-Net level in a ragged launch (see ConvArgs::rag_w): w[b] = width of utterance b, off[b] = its
// first pixel in the packed plane sets (sum over the utterances before it of H * w), soff[b] = the same prefix over its GroupNorm
// statistics sub-tiles (H * ceil(w / 32)).  w == nullptr: every utterance has the launch's W (uniform batch).
struct Rag { const int* w; const long long* off; const long long* soff; };

struct ConvArgs {
  const float* src1; const float* src2;  // NCHW; virtual concat [src1 | src2] along C (src2 may be null)
  int C1, C2;
  const float* w;          // conv_mfma: packed [nCoBlk][Cin][TAPS][CO_T]; conv_direct: OIHW [Cout][Cin][KS][KS]
  const float* bias;       // [Cout] or null
  const float* bias2;      // per-row bias table (time embedding + conv bias) or null; row = step*sstride + b*bstride
  int bias2_bstride, bias2_sstride;
  const int* step_ptr;     // device-side sampler step counter or null
  const float* in_scale; const float* in_shift;  // [B][C1+C2] fused GroupNorm affine, or null
  int in_act;              // 1: SiLU after the affine
  const float* res;        // residual [B][Cout][H][W] or null
  const float* acc_scale;  // device scalar multiplying the accumulator first (split kernels with pre-scaled operands), or null
  const float* co_scale;   // conv3x3_wino_kernel: per OUTPUT CHANNEL factor undoing the weights' power-of-two scale ([ceil(Cout/128)*128])
  float out_scale;         // out = (acc * acc_scale + bias + bias2 + res) * out_scale
  float* out;
  int Cout, B, H, W;
  // optional GroupNorm statistics of the stored output, fused into the epilogue: per (b, co, sub-tile) partial
  // {sum, sum of squares}; sub-tile = one image row x one 32-pixel segment, index y * ceil(W/32) + x/32 -- the unit every
  // tile shape of this kernel family decomposes into, so the partials (and everything computed from them) do not depend
  // on which tile shape a launch used.  Deterministic (no atomics); stats_nsub = H * ceil(W/32).
  float* stats_out;
  int stats_nsub;
  // rows per statistics sub-tile: 1 (the fp32 MFMA kernels, whose tile shapes go down to one row per wave) or 4 (the split-operand
  // kernels: every shape gives a wave 4 or 8 consecutive rows, so a wave sums its per-row results over 4 rows before storing --
  // a quarter of the partials to write and to read back; sub-tile index (y / 4) * ceil(W/32) + x/32, stats_nsub = ceil(H/4) *
  // ceil(W/32)).  Decided by the kernel FAMILY, which is decided per layer and level: still independent of tile shape and batch.
  int stats_rows;
  // optional per-utterance max |stored output| ([B][kAmaxSpread]: atomic max, order-independent and therefore
  // deterministic, spread over kAmaxSpread words per utterance so that the atomics of a launch do not serialise on one
  // address; zeroed by the engine before the forward): the range bound from which an fp16x2 consumer of this tensor picks
  // its exact power-of-two scale
  float* amax_out;
  // per-utterance range bounds of src1 / src2 ([B][kAmaxSpread] each, from the producers' amax_out) for a consumer that
  // scales its input dynamically (conv1x1_split_kernel<SplitH2>); null otherwise
  const float* amax1; const float* amax2;
  // per-utterance upper bound of |producer output| ([B][kAmaxSpread], written by gn_finalize_kernel from the utterance's own
  // statistics and range bound) for the fp16x2 3x3 kernel: its input scale is the power of two that puts this bound in
  // [2^13, 2^14) -- data-driven, per utterance, no presumption about GroupNorm parameters or utterance length
  const float* xbound;
  // measurement-only ablation switches of the fp32 kernels for sgmse_bench_conv (results are then WRONG on purpose):
  // bit 2 stage only the first K-stage, bit 3 skip the barriers (the epilogue's switches are compile-time: conv_epilogue<ABL>)
  int ablate;
  // Folded residual shortcut of the split 3x3 kernel (ResnetBlockBigGANpp with a Conv_2, layerspp.py:266-274): out =
  // (Conv_1(act(GN(h))) + Conv_2(x)) / sqrt 2 is ONE accumulation -- the 1x1 shortcut runs as extra K-stages (centre tap only)
  // over the raw block input x in front of the 3x3 stages, instead of a separate launch that writes a tensor the 3x3 kernel
  // reads back as its residual.  sc_src1 | sc_src2: x (virtual concat, sc_C1 + sc_C2 channels, same H x W); sc_w: its weights
  // in the split kernels' fragment order (one tap); sc_scale: device scalar undoing the weights' power-of-two scale; sc_bias:
  // Conv_2's bias; sc_amax1 / 2: range bounds of x, from which the input's power-of-two scale is derived per utterance.
  const float* sc_src1; const float* sc_src2; int sc_C1, sc_C2;
  const float* sc_w; const float* sc_scale; const float* sc_bias; const float* sc_amax1; const float* sc_amax2;
  // Chunked accumulation of the fp32 MFMA kernels' small tiles (coarse U-Net levels): kchunk_stages > 0 -> the reduction over the
  // input channels is the sum, in chunk order, of per-chunk partial sums (each an MFMA chain from zero over kchunk_stages K-stages).
  // With gridDim.z == 1 a workgroup runs all chunks itself; with gridDim.z == number of chunks each workgroup computes ONE chunk
  // and stores the raw partial sums to `partial` [chunk][B][Cout][H][W], summed in the same order (and run through the
  // epilogue) by conv_splitk_reduce_kernel: bit-identical results, but the chunks run on different CUs (small batches).
  int kchunk_stages;
  float* partial;
  // Ragged batches (utterances of different widths T_b in ONE launch, sgmse_set_frames): every tensor holds utterance b's planes
  // with ITS OWN row stride W_b, packed one utterance after the other ([sum_b C*H*W_b] floats).  rag_w[b] = W_b of this launch's
  // U-Net level, rag_off[b] = sum_{b' < b} H*W_b' (pixels of one channel plane set); W is then the widest utterance (the grid is
  // laid out for it, tiles beyond W_b exit) and rag_slab the elements of one split-K partial slab.  conv_ragged_adjust() rewrites
  // a workgroup's argument copy so that the uniform-batch addressing (b*C + c)*H*W lands on utterance b's block: what an
  // utterance computes is exactly what its single-utterance launch computes, bit for bit.
  const int* rag_w; const long long* rag_off; const long long* rag_soff; long long rag_slab;
  int rag_vec_ok;             // ragged: every utterance's width is a multiple of 4 (float4 staging allowed)
  // measurement (ABL bit 6 instantiation): per workgroup {hw_id | xcc_id << 32, t_start, t_loop, t_epilogue, t_end} (shader clock)
  unsigned long long* trace;
  // XCD-aware tile order (0: off, the default until measured).  Workgroups are dealt round-robin over the 8 XCDs by their linear
  // id and every XCD has its own L2: with the plain map a tile's vertical neighbour sits on the same XCD only when the tile row is a
  // multiple of 8 tiles long (T = 512: 16 tiles) -- other lengths fetch every halo row again (profiles/r03_length_sweep.txt: 3-4 %
  // per frame at T = 384 / 448 / 576 / 640).  With the map on, XCD k takes the k-th contiguous eighth of the launch's tiles, in order.
  // A pure permutation of which workgroup computes which tile: results cannot change.
  int xcd_map;
  // Ragged launch laid out over the tiles that EXIST (null: over the widest utterance's tile columns, tiles beyond an utterance's
  // width exit).  rag_cols[b] = sum over the utterances before b of ceil(W_b' / 32) (B + 1 entries), rag_ncols = rag_cols[B]:
  // gridDim.x = tiles_y * rag_ncols, utterance b owns the ids [tiles_y * rag_cols[b], tiles_y * rag_cols[b + 1]), row-major inside.
  // An exiting workgroup still has to be dispatched with the kernel's LDS and registers; a batch of 2 ... 6 s utterances launches
  // 24 576 tile columns for 16 896 used, and its convolutions take 1.14x per frame (profiles/r03_ragged_per_launch_dump.txt).
  const int* rag_cols; int rag_ncols;
};
// dispatch id -> tile id under ConvArgs::xcd_map (nt tiles along gridDim.x; needs nt % 8 == 0, which 8-row tiles of a 256-bin image give)
__device__ __forceinline__ int conv_xcd_tile(int t, int nt, int on) { return (on && (nt & 7) == 0) ? (t & 7) * (nt >> 3) + (t >> 3) : t; }
// dispatch id t of nt -> utterance b, tile row ty, tile column tx (every convolution kernel's first step)
__device__ __forceinline__ void conv_tile_of(const ConvArgs& p, int t, int nt, int tiles_xg, int tiles_y, int& b, int& ty, int& tx) {
  int bid = conv_xcd_tile(t, nt, p.xcd_map);
  if (p.rag_cols) {                        // (uniform: a handful of dependent scalar loads)
    int lo = 0, hi = p.B;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.rag_cols[mid] * tiles_y <= bid) lo = mid; else hi = mid; }
    const int c0 = p.rag_cols[lo], nx = p.rag_cols[lo + 1] - c0, r = bid - c0 * tiles_y;
    b = lo; ty = r / nx; tx = r - ty * nx;
  } else {
    tx = bid % tiles_xg; bid /= tiles_xg;
    ty = bid % tiles_y;
    b = bid / tiles_y;
  }
}

#ifndef SGMSE_SPLITK_FRAGMENT_MAJOR
#define SGMSE_SPLITK_FRAGMENT_MAJOR 1
#endif
constexpr int kAmaxSpread = 64;

// ragged launch (ConvArgs::rag_w): point this workgroup's argument copy at utterance b; false = the tile lies beyond its width
__device__ __forceinline__ bool conv_ragged_adjust(ConvArgs& p, int b, int tx) {
  const int Wb = p.rag_w[b];
  if (tx * 32 >= Wb) return false;
  const long long d = p.rag_off[b] - (long long)b * p.H * Wb;
  p.W = Wb;
  p.src1 += d * p.C1;
  if (p.src2) p.src2 += d * p.C2;
  p.out += d * p.Cout;
  if (p.res) p.res += d * p.Cout;
  if (p.sc_src1) p.sc_src1 += d * p.sc_C1;
  if (p.sc_src2) p.sc_src2 += d * p.sc_C2;
  if (p.partial) p.partial += d * p.Cout;
  if (p.stats_out) {            // statistics sub-tiles packed per utterance too: [soff[b] .. ) x Cout pairs, H * ceil(W_b / 32) per channel
    const int rps = p.stats_rows == 4 ? 4 : 1;          // (4-row sub-tiles: H % 4 == 0 in every ragged launch, so the prefix divides)
    const int nsub = (p.H / rps) * ((Wb + 31) >> 5);
    p.stats_out += (p.rag_soff[b] / rps - (long long)b * nsub) * p.Cout * 2;
    p.stats_nsub = nsub;
  }
  return true;
}
// elements between the split-K partial slabs of consecutive chunks
__device__ __forceinline__ size_t conv_partial_slab(const ConvArgs& p, int H, int W) {
  return p.rag_w ? (size_t)p.rag_slab : (size_t)p.B * p.Cout * H * W;
}

// max over the kAmaxSpread words of utterance b (every lane of the calling wave gets the result)
__device__ __forceinline__ float amax_read(const float* amax, int b) {
  float m = amax[b * kAmaxSpread + (threadIdx.x & (kAmaxSpread - 1))];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}

// exact power of two 2^k with m 2^k in [2^13, 2^14): the operand scale of the fp16x2 kernels for a tensor bounded by m
__device__ __forceinline__ float h2_weight_scale(float absmax) {
  if (!(absmax > 0.f)) return 1.f;
  const int e = (int)((__builtin_bit_cast(uint32_t, absmax) >> 23) & 0xff) - 127;    // floor(log2(absmax)) for normals
  int k = 13 - e;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  return __builtin_bit_cast(float, (uint32_t)(k + 127) << 23);
}

// ---- GroupNorm coefficients from partial sums (nn.GroupNorm(min(C//4,32), C, eps=1e-6), reference layerspp.py:67,219,231) --------
// One job: scale[b][c], shift[b][c] (x * scale + shift = GroupNorm(x) with gamma/beta folded in) of the virtual concat
// [source 1 | source 2], each source given as per-(b, c) partial {sum, sumsq} pairs (nsub per channel: 1 from gn_chan_stats_kernel,
// the number of statistics sub-tiles when a convolution epilogue produced them), plus the range bound an fp16x2 consumer scales by
// (bound_out, see gn_finalize_kernel).  rps: image rows per statistics sub-tile of the source (ragged launches need it).
struct GnFin {
  const float* st1; const float* st2; const float* amax1; const float* amax2;
  const float* gamma; const float* beta; float* scale; float* shift; float* bound_out;
  Rag rag;
  int C1, nsub1, rps1, C2, nsub2, rps2, G, HW, H;
  float eps;
};

// The canonical order of a group's sum: 256 SLOTS, slot j takes the elements j, j + 256, ... (two pairs per 16-byte load where the
// alignment allows) of source 1's share and then of source 2's, in fp64; the slots are then summed by the tree
// slot[t] += slot[t + m], m = 128 ... 1 (gn_finalize_kernel: a slot per thread).
__device__ __forceinline__ void gn_sum_pairs(const float* base, int n, int slot, double& s, double& q) {
  if ((reinterpret_cast<uintptr_t>(base) & 15) == 0 && (n & 1) == 0) {
    const float4* b4 = reinterpret_cast<const float4*>(base);
#pragma unroll 4
    for (int i = slot; i < n / 2; i += 256) {
      const float4 v = b4[i];
      s += (double)v.x + (double)v.z; q += (double)v.y + (double)v.w;
    }
  } else {
    for (int i = slot; i < n; i += 256) { s += (double)base[2 * i]; q += (double)base[2 * i + 1]; }
  }
}

struct GnGroupSpan { const float* p1; int n1; const float* p2; int n2; int HW; };
// where the partial pairs of group g of utterance b lie (source 1 covers channels [c_lo, min(c_hi, C1)), source 2 the rest: a group
// may straddle the concat boundary) and the utterance's pixels per plane
__device__ __forceinline__ GnGroupSpan gn_group_span(const GnFin& f, int b, int g) {
  const int cpg = (f.C1 + f.C2) / f.G;
  const int c_lo = g * cpg, c_hi = c_lo + cpg;
  int nsub1 = f.nsub1, nsub2 = f.nsub2, HW = f.HW;
  size_t sb1 = (size_t)b * nsub1, sb2 = (size_t)b * nsub2;       // first sub-tile of utterance b, per channel-set
  if (f.rag.w) {
    const int w = f.rag.w[b], ns = f.H * ((w + 31) >> 5);
    HW = f.H * w;
    if (nsub1 > 1) { nsub1 = ns / f.rps1; sb1 = (size_t)(f.rag.soff[b] / f.rps1); }
    if (nsub2 > 1) { nsub2 = ns / f.rps2; sb2 = (size_t)(f.rag.soff[b] / f.rps2); }
  }
  GnGroupSpan sp{nullptr, 0, nullptr, 0, HW};
  const int a_hi = c_hi < f.C1 ? c_hi : f.C1;
  if (c_lo < a_hi) { sp.p1 = f.st1 + (sb1 * f.C1 + (size_t)c_lo * nsub1) * 2; sp.n1 = (a_hi - c_lo) * nsub1; }
  const int b_lo = c_lo > f.C1 ? c_lo : f.C1;
  if (b_lo < c_hi) { sp.p2 = f.st2 + (sb2 * f.C2 + (size_t)(b_lo - f.C1) * nsub2) * 2; sp.n2 = (c_hi - b_lo) * nsub2; }
  return sp;
}

// the group's totals -> coefficients of its channels (lane < cpg of ONE wave; cpg <= 64) and the group's share of the range bound.
//     |x - mean| <= min(max|x| + |mean|, sqrt(N var))      max|x|: the producers' range bounds amax1 / amax2 (null: unknown)
//     bound_c = that * rstd * |gamma_c| + |beta_c|,        N var = sum of squared deviations of the group (>= any single one)
// reduced over the group's channels here and over the groups by an atomic max (order-independent: deterministic).
__device__ __forceinline__ void gn_group_coeffs(const GnFin& f, int b, int g, int lane, double s_tot, double q_tot, int HW) {
  const int C = f.C1 + f.C2, cpg = C / f.G, c_lo = g * cpg;
  float am = -1.f;                    // max |x| of the utterance over both sources; < 0: unknown
  if (f.bound_out && f.amax1 && (f.C2 == 0 || f.amax2)) {        // (whole wave: amax_read shuffles)
    am = amax_read(f.amax1, b);
    if (f.C2) am = fmaxf(am, amax_read(f.amax2, b));
  }
  float bnd = 0.f;
  if (lane < cpg) {
    const int c = c_lo + lane;
    const double n = (double)cpg * (double)HW;
    const double mean = s_tot / n;
    double var = q_tot / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float a = f.gamma[c] * rstd;
    f.scale[(size_t)b * C + c] = a;
    f.shift[(size_t)b * C + c] = f.beta[c] - (float)mean * a;
    // (q / n - mean^2 comes from fp32 partial sums: for |mean| >> std the cancellation can under-estimate n var by ~2^-21 n mean^2 --
    //  that much is added back, so the bound stays a bound; negligible next to n var whenever var is not itself lost in the rounding)
    double dev = sqrt(n * var + n * mean * mean * 9.5367431640625e-07);
    if (am >= 0.f) dev = fmin(dev, (double)am + fabs(mean));
    bnd = (float)(dev * (double)rstd * fabs((double)f.gamma[c]) + fabs((double)f.beta[c])) * 1.0001f;   // (margin for the fp32 affine's own rounding)
    if (!(bnd >= 0.f)) bnd = 3.0e38f;      // NaN statistics: the result is NaN whatever the scale
  }
  if (f.bound_out) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, o));
    if (lane == 0) drt_atomic_max_nonneg(f.bound_out + b * kAmaxSpread + (g & (kAmaxSpread - 1)), bnd);
  }
}

// SiLU x*sigmoid(x) (nn.SiLU, reference layers.py:38-39) on the hardware exp2/rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each,
// relative error of the result < 4e-7, far inside the 1e-5 per-op parity gate) -- the fused producers evaluate it for every
// staged element, so a libm-accurate expf + IEEE divide (~25 VALU instructions) would dominate the staging phase.
// (__frcp_rn is NOT that: it expands to the full div_scale / div_fmas / div_fixup sequence.)
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float silu_precise_f(float v) { return v / (1.0f + expf(-v)); }   // time-embedding MLP (tiny)

template <int KS, int WC, int FC, int FP, int VEC = 0>
struct ConvTile {
  static constexpr int WP = 4 / WC;                 // waves along pixels
  static constexpr int CO_T = WC * FC * 32;         // output channels per workgroup
  static constexpr int ROWS = WP * FP;              // image rows per workgroup (each pixel fragment = 32 px of a row)
  static constexpr int KC = (KS == 3) ? 8 : 32;     // input channels per LDS stage (Cin % KC == 0)
  static constexpr int HALO = KS / 2;
  // LDS row: scalar staging packs [halo | 32 px | halo] (stride 34 / 32); vector staging keeps the 32 interior pixels
  // 16-byte aligned at column 4 with the halo columns at 3 and 36 (stride 40) so they can be written as b128
  static constexpr int RS = (VEC && HALO) ? 40 : 32 + 2 * HALO;
  static constexpr int XOFF = (VEC && HALO) ? 3 : 0;   // column of the left-most tap of pixel 0
  static constexpr int TROWS = ROWS + 2 * HALO;
  static constexpr int PLANE = TROWS * RS;
  static constexpr int TAPS = KS * KS;
  static constexpr int IN_ELEMS = KC * PLANE;
  static constexpr int W_ELEMS = KC * TAPS * CO_T;
  static constexpr int NI = (IN_ELEMS + 255) / 256;    // scalar staging: elements per thread
  static constexpr int NVI = KC * TROWS * 8;           // vector staging: float4 items per stage
  static constexpr int NV = (NVI + 255) / 256;
  static constexpr int NHI = KC * TROWS * 2 * HALO;    // vector staging: halo scalars per stage (<= 256)
  static constexpr int NW4 = (W_ELEMS / 4 + 255) / 256;
};

// Accumulator initialisation with the per-channel additive terms of the epilogue, (bias + time-embedding row) / as, so that
// their loads overlap the first K-stage's loads instead of delaying the epilogue and the epilogue neither holds 16 bias
// registers nor adds them (as = the power of two that takes the accumulator back to the unscaled convolution: the
// division is exact).  Returns the value for accumulator register r of every pixel fragment of channel fragment `frag`.
// conv_acc_raw: the loads (bias + time-embedding row, unscaled) -- issue them with the prologue's other independent loads;
// conv_acc_init: the same followed by the division.
template <class T>
__device__ __forceinline__ void conv_acc_raw(const ConvArgs& p, int b, int co_blk, int frag, int kh, float (&init)[16]) {
  const float* b2 = nullptr;
  if (p.bias2) {
    const int step = p.step_ptr ? *p.step_ptr : 0;
    b2 = p.bias2 + (size_t)step * p.bias2_sstride + (size_t)b * p.bias2_bstride;
  }
  const int co_l = co_blk * T::CO_T + frag * 32 + 4 * kh;
#pragma unroll
  for (int r = 0; r < 16; ++r) init[r] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int co = co_l + (r & 3) + 8 * (r >> 2); init[r] = p.bias[co < p.Cout ? co : 0]; }
  }
  if (b2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int co = co_l + (r & 3) + 8 * (r >> 2); init[r] += b2[co < p.Cout ? co : 0]; }
  }
}
// factor that takes an accumulator of output channel co back to the unscaled convolution (before the per-utterance input factor):
// per channel for the fp16x2 kernels (ConvArgs::co_scale, a table padded to whole 128-channel blocks), else the layer's scalar or 1
__device__ __forceinline__ float conv_as(const ConvArgs& p, int co) { return p.co_scale ? p.co_scale[co] : (p.acc_scale ? *p.acc_scale : 1.0f); }
template <class T>
__device__ __forceinline__ void conv_acc_init(const ConvArgs& p, int b, int co_blk, int frag, int kh, float as_mul, float (&init)[16]) {
  const float* b2 = nullptr;
  if (p.bias2) {
    const int step = p.step_ptr ? *p.step_ptr : 0;
    b2 = p.bias2 + (size_t)step * p.bias2_sstride + (size_t)b * p.bias2_bstride;
  }
  const int co_l = co_blk * T::CO_T + frag * 32 + 4 * kh;
#pragma unroll
  for (int r = 0; r < 16; ++r) init[r] = 0.f;
  if (p.bias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int co = co_l + (r & 3) + 8 * (r >> 2); init[r] = p.bias[co < p.Cout ? co : 0]; }
  }
  if (b2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int co = co_l + (r & 3) + 8 * (r >> 2); init[r] += b2[co < p.Cout ? co : 0]; }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) init[r] *= 1.0f / (conv_as(p, co_l + (r & 3) + 8 * (r >> 2)) * as_mul);      // (powers of two: exact)
}

// Epilogue shared by the MFMA kernels: D[co = regs][px = lanes] -> NCHW rows, + bias, + time-embedding bias row,
// + residual, * out_scale, and (optionally) the per-(b, co, sub-tile) GroupNorm partial sums of the stored values.
//
// Addressing: a lane's byte offset inside its utterance is lane_boff + j W 4 (one VGPR per fragment row) plus
// ((r&3) + 8 (r>>2)) H W 4, the same for every lane (16 SGPRs shared by the output and the residual), so every access is
// one raw-buffer instruction `base + VGPR offset + SGPR offset` with no address arithmetic on the vector ALU.  GUARD = false is the fast path of a wave whose 32 output channels, FP rows and
// 32 columns all lie inside the tensor (wave-uniform test); GUARD = true predicates every access.
// GroupNorm partials: one {sum, sum of squares} per (channel, image row, 32-pixel segment), reduced over the 32 lanes
// of the segment by an exchange-add butterfly (drt_xadd: v_permlane16_swap / DPP, no LDS traffic): after the steps
// 16, 8, 7 (mirror), 1 a lane owns ONE of the 16 registers summed over 16 lanes, and the final ^2 step completes it.
// The order of the additions is a function of (row, segment) only -- not of the tile shape, wave or workgroup -- so the
// statistics, and everything computed from them, do not depend on which tile shape a launch used.
template <class T, int FC, int FP, int WC, bool GUARD, int ABL, bool PRE, bool RES, bool STAT4, bool CSV>
__device__ __forceinline__ float conv_epilogue_body(const ConvArgs& p, f32x16 (&acc)[FC][FP], int b, int co_blk, int tx, int ty,
                                                    int tiles_x, int wc, int wp, int l31, int kh, float as, float as_mul) {
  constexpr int CO_T = T::CO_T, ROWS = T::ROWS;
  constexpr int PF = (ABL & 128) ? 1 : (FP < 4 ? FP : 4);   // residual rows in flight (16 registers each)
  constexpr int AUX = (ABL & 256) ? 2 : 0;
  const int H = p.H, W = p.W;
  const unsigned HW = (unsigned)H * (unsigned)W;
  const int x = tx * 32 + l31;
  const int yb = ty * ROWS + wp * FP;            // first image row of this wave
  const bool xok = !GUARD || x < W;
  const size_t ubase = (size_t)b * p.Cout * HW;  // uniform
  const drt_buf obuf = drt_make_buf(p.out + ubase), rbuf = drt_make_buf(RES ? p.res + ubase : p.out);
  float vmax = 0.f;
  const float* b2 = nullptr;
  if (p.bias2) {
    const int step = p.step_ptr ? *p.step_ptr : 0;
    b2 = p.bias2 + (size_t)step * p.bias2_sstride + (size_t)b * p.bias2_bstride;
  }
  constexpr bool has_res = RES;
#pragma unroll
  for (int i = 0; i < FC; ++i) {
    const int co0 = co_blk * CO_T + (wc * FC + i) * 32;                       // uniform
    const int co_l = co0 + 4 * kh;                                            // + (r&3) + 8 (r>>2)
    // byte offset of (channel co_l, row yb, column x) inside the utterance; < 2^31 for every tensor of this network
    const unsigned lane_boff = (((unsigned)co_l * (unsigned)H + (unsigned)yb) * (unsigned)W + (unsigned)x) * 4u;
    auto soff = [&](int r) -> unsigned { return (unsigned)((r & 3) + 8 * (r >> 2)) * HW * 4u; };
    // CSV: the accumulator factor is per output channel (fp16x2 weights carry a power-of-two scale per channel: ConvArgs::co_scale)
    float asv[CSV ? 16 : 1];
    if constexpr (CSV) {
#pragma unroll
      for (int r = 0; r < 16; ++r) asv[r] = p.co_scale[co_l + (r & 3) + 8 * (r >> 2)] * as_mul;      // (table padded to whole blocks: no guard)
    }
    // per-channel additive terms (conv bias + time-embedding row): two batches of independent loads, each under ONE uniform
    // branch (a branch per element serialises 32 loads behind vmcnt(0) waits); PRE: already in the accumulators
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = 0.f;
    if constexpr (!PRE) {
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_l + (r & 3) + 8 * (r >> 2);
          bv[r] = p.bias[(!GUARD || co < p.Cout) ? co : 0];
        }
      }
      if (b2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_l + (r & 3) + 8 * (r >> 2);
          bv[r] += b2[(!GUARD || co < p.Cout) ? co : 0];
        }
      }
    }
    // residual rows are loaded PF - 1 fragment rows ahead of their use (independent loads; issued one by one behind their
    // dependent add + store the residual read cost 11 % of the fp32 kernel, profiles/r01_conv_ablation.txt)
    float st1 = 0.f, st2 = 0.f;     // STAT4: this lane's {sum, sum of squares} over the rows of the current 4-row sub-tile
    float rr[PF][16];
    auto load_res = [&](int j, int slot) {
      const bool rok = !GUARD || (yb + j < H && xok);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = rok && (!GUARD || co_l + (r & 3) + 8 * (r >> 2) < p.Cout);
        float t = 0.f;
        if (ok) t = drt_buf_load<AUX>(rbuf, lane_boff + (unsigned)j * (unsigned)W * 4u, soff(r));
        rr[slot][r] = t;
      }
    };
    if constexpr (has_res) {
#pragma unroll
      for (int j = 0; j < PF - 1; ++j) load_res(j, j);
    }
#pragma unroll
    for (int j = 0; j < FP; ++j) {
      if constexpr (has_res) { if (j + PF - 1 < FP) load_res(j + PF - 1, (j + PF - 1) % PF); }
      const int y = yb + j;
      const bool rok = !GUARD || (y < H && xok);
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] * (CSV ? asv[CSV ? r : 0] : as);
        if constexpr (!PRE) v += bv[r];
        if constexpr (has_res) v += rr[j % PF][r];
        v *= p.out_scale;
        const bool ok = rok && (!GUARD || co_l + (r & 3) + 8 * (r >> 2) < p.Cout);
        if (ok && (!(ABL & 1) || v == 12345.678f))      // ABL bit 0 (measurement only): keep the value live without storing it
          drt_buf_store<AUX>(obuf, v, lane_boff + (unsigned)j * (unsigned)W * 4u, soff(r));
        if (GUARD) v = ok ? v : 0.f;
        vmax = fmaxf(vmax, fabsf(v));
        sv[r] = v;
      }
      DRT_PIN_HERE(vmax);      // or the compiler sinks the whole max chain to its only use behind the epilogue, keeping all 128 values alive
      if (p.stats_out && (!GUARD || y < H)) {   // wave-uniform: GroupNorm partials of this 32-pixel row segment
        // the two butterflies one after the other (fenced): together their temporaries spill
        auto butterfly = [&](bool square) -> float {
          float a[8];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            a[k] = square ? drt_xadd<16>(sv[k] * sv[k], sv[k + 8] * sv[k + 8]) : drt_xadd<16>(sv[k], sv[k + 8]);
          float c[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) c[k] = drt_xadd<8>(a[k], a[k + 4]);
          const float d0 = drt_xadd<7>(c[0], c[2]), d1 = drt_xadd<7>(c[1], c[3]);
          return drt_add_xor2(drt_xadd<1>(d0, d1));
        };
        // squares first: the exchange instructions overwrite their operands, and the second pass may then consume sv itself
        const float e2 = butterfly(true);
        __builtin_amdgcn_sched_barrier(0);
        const float e1 = butterfly(false);
        // the register this lane ended up with: bit 3 of r from lane bit 4, bit 2 from bit 3, bit 1 from bit 2, bit 0 from bit 0
        const int r = ((l31 >> 4) & 1) * 8 + ((l31 >> 3) & 1) * 4 + ((l31 >> 2) & 1) * 2 + (l31 & 1);
        const int co = co_l + (r & 3) + 8 * (r >> 2);
        // lanes l and l ^ 2 hold the same sums and both store them (same address, same value): predicating one of them
        // away costs an exec-mask region per row, which the compiler gathers at the end of the epilogue with the sums spilled
        if constexpr (STAT4) {
          // rows in image order, ((r0 + r1) + r2) + r3 of the per-row butterfly results: the same in the 4-row and 8-row shapes
          st1 = (j & 3) == 0 ? e1 : st1 + e1;
          st2 = (j & 3) == 0 ? e2 : st2 + e2;
          if (((j & 3) == 3 || (GUARD && y + 1 >= H)) && (!GUARD || co < p.Cout)) {
            float* so = p.stats_out + ((size_t)(b * p.Cout + co) * p.stats_nsub + (size_t)(y >> 2) * tiles_x + tx) * 2;
            so[0] = st1; so[1] = st2;
          }
        } else {
          if (!GUARD || co < p.Cout) {
            float* so = p.stats_out + ((size_t)(b * p.Cout + co) * p.stats_nsub + (size_t)y * tiles_x + tx) * 2;
            so[0] = e1; so[1] = e2;
          }
        }
      }
    }
  }
  // four instantiations of this body meet behind the dispatch below: without a marker that differs per instantiation the
  // compiler sinks their (identical) tails into the common successor and carries every operand there in a phi (spills)
  DRT_CODE_MARKER(GUARD * 2 + RES);
  return vmax;
}

// ABL (measurement-only instantiations of sgmse_bench_conv; with bits 0 / 1 results are WRONG on purpose): bit 0 skip the
// global stores, bit 1 skip the residual read, bit 7 residual rows loaded right before their use, bit 8 non-temporal
// cache policy on the residual loads and output stores
// PRE: the kernel initialised its accumulators with (bias + bias2) / as (conv_acc_init), the epilogue adds no bias
// STAT4: GroupNorm partials per 4 image rows (ConvArgs::stats_rows == 4; needs FP % 4 == 0: a wave's rows are whole sub-tiles)
// CSV: per-output-channel accumulator factors (ConvArgs::co_scale; the fp16x2 kernels) instead of the layer's scalar
template <class T, int FC, int FP, int WC, int ABL = 0, bool PRE = false, bool STAT4 = false, bool CSV = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[FC][FP], int b, int co_blk, int tx, int ty,
                                              int tiles_x, int wc, int wp, int l31, int kh, float as_mul = 1.0f) {
  static_assert(!STAT4 || FP % 4 == 0, "4-row statistics sub-tiles need 4 or 8 rows per wave");
  const float as = (p.acc_scale ? *p.acc_scale : 1.0f) * as_mul;     // exact powers of two (or 1)
  // wave-uniform: are this wave's channels, rows and columns all inside the tensor?
  const bool inside = (co_blk * T::CO_T + (wc * FC + FC) * 32 <= p.Cout) && (ty * T::ROWS + wp * FP + FP <= p.H) && (tx * 32 + 32 <= p.W);
  const bool res = p.res != nullptr && !(ABL & 2);
  float vmax;
  if (inside) {
    if (res) vmax = conv_epilogue_body<T, FC, FP, WC, false, ABL, PRE, true, STAT4, CSV>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kh, as, as_mul);
    else vmax = conv_epilogue_body<T, FC, FP, WC, false, ABL, PRE, false, STAT4, CSV>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kh, as, as_mul);
  } else {
    if (res) vmax = conv_epilogue_body<T, FC, FP, WC, true, ABL, PRE, true, STAT4, CSV>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kh, as, as_mul);
    else vmax = conv_epilogue_body<T, FC, FP, WC, true, ABL, PRE, false, STAT4, CSV>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kh, as, as_mul);
  }
  if (p.amax_out) {      // wave-uniform
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((threadIdx.x & 63) == 0)
      drt_atomic_max_nonneg(p.amax_out + b * kAmaxSpread + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (kAmaxSpread - 1)), vmax);
  }
}

// Second half of a split-K convolution: the chunks' partial sums, summed in chunk order into the accumulator layout of the
// producing tile shape (elements outside the tensor: 0).
template <class T, int FC, int FP, int WC>
__device__ __forceinline__ void conv_splitk_sum(const ConvArgs& p, int nchunks, int b, int co_blk, int tx, int ty, int wc, int wp,
                                                int l31, int kh, f32x16 (&acc)[FC][FP]) {
  constexpr int ROWS = T::ROWS, CO_T = T::CO_T;
  const int H = p.H, W = p.W;
  const size_t slab = conv_partial_slab(p, H, W);
  const int x = tx * 32 + l31;
  // Up to 8 chunks (what the engine's chunking produces for 256 / 512 input channels): fragment loop OUTSIDE, the 16 x nchunks loads of one
  // fragment in flight together -- FC*FP memory round trips per workgroup instead of nchunks (round 5: the reduce launches of the coarse
  // levels are 53 x 17 us of a batch-1 evaluation, most of it the chunk loop's serial round trips to partial sums that the chunk
  // workgroups have just written through to memory).  An element still sums its chunks in chunk order: the same bits.
  // (element offsets fit 32 bits: inside a slab, and -- ragged launches rebase p.partial by the utterance's packed offset, so that
  //  (b Cout + co) H W is a VIRTUAL offset of up to B slabs -- over the whole batch)
  if (SGMSE_SPLITK_FRAGMENT_MAJOR && nchunks <= 8 && slab < ((size_t)1 << 28) && (size_t)p.B * slab < ((size_t)1 << 31)) {
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
      for (int j = 0; j < FP; ++j) {
        const int y = ty * ROWS + wp * FP + j;
        unsigned off[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_blk * CO_T + (wc * FC + i) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
          const bool ok = co < p.Cout && y < H && x < W;
          off[r] = ok ? (unsigned)((((size_t)b * p.Cout + co) * H + y) * W + x) : 0u;      // clamped, unpredicated
        }
        f32x16 v[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          const float* pz = p.partial + (size_t)(z < nchunks ? z : 0) * slab;     // (chunks past the end: re-read chunk 0, never added)
#pragma unroll
          for (int r = 0; r < 16; ++r) v[z][r] = pz[off[r]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = v[0][r];
#pragma unroll
        for (int z = 1; z < 8; ++z) {
          if (z < nchunks) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + v[z][r];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  } else
  // chunk loop OUTSIDE: the FC*FP*16 loads of one chunk are independent and in flight together; an element still sums its
  // chunks in chunk order (what makes split-K bit-identical to the chunked single-workgroup run)
  for (int z = 0; z < nchunks; ++z) {
    const float* pz = p.partial + (size_t)z * slab;
    f32x16 v[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
      for (int j = 0; j < FP; ++j) {
        const int y = ty * ROWS + wp * FP + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_blk * CO_T + (wc * FC + i) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
          const bool ok = co < p.Cout && y < H && x < W;
          v[i][j][r] = pz[ok ? ((size_t)(b * p.Cout + co) * H + y) * W + x : 0];      // clamped, unpredicated
        }
      }
    __builtin_amdgcn_sched_barrier(0);      // (left alone the compiler runs load - wait - add through ONE register, 16*FC*FP round trips)
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
      for (int j = 0; j < FP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = z == 0 ? v[i][j][r] : acc[i][j][r] + v[i][j][r];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < FC; ++i)
#pragma unroll
    for (int j = 0; j < FP; ++j) {
      const int y = ty * ROWS + wp * FP + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_blk * CO_T + (wc * FC + i) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
        if (!(co < p.Cout && y < H && x < W)) acc[i][j][r] = 0.f;
      }
    }
}

// PREF = 1: the LDS operand reads of k-step s+1 are issued before the MFMAs of k-step s (order pinned with
//           sched_group_barrier), so a wave does not park on lgkmcnt between MFMA groups.  PREF = 0: compiler order.
// VEC  = 1: float4 input staging (needs W % 4 == 0 and 16-byte aligned sources); 0: element-wise staging.
template <int KS, int WC, int FC, int FP, int PREF, int VEC>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs p) {
  using T = ConvTile<KS, WC, FC, FP, VEC>;
  constexpr int CO_T = T::CO_T, ROWS = T::ROWS, KC = T::KC, HALO = T::HALO, RS = T::RS, PLANE = T::PLANE,
                TAPS = T::TAPS, NI = VEC ? 1 : T::NI, NW4 = T::NW4, NV = VEC ? T::NV : 1, TROWS = T::TROWS;
  __shared__ float s_in[T::IN_ELEMS];
  __shared__ float s_w[T::W_ELEMS];
  __shared__ float s_sc[512];
  __shared__ float s_sh[512];

  const int tid = threadIdx.x;
  const int Cin = p.C1 + p.C2;
  const int tiles_xg = (p.W + 31) >> 5;       // grid layout (ragged launches: of the widest utterance)
  const int tiles_y = (p.H + ROWS - 1) / ROWS;
  int b, ty, tx;
  conv_tile_of(p, (int)blockIdx.x, (int)gridDim.x, tiles_xg, tiles_y, b, ty, tx);
  if (p.rag_w) { if (!conv_ragged_adjust(p, b, tx)) return; }
  const int tiles_x = (p.W + 31) >> 5;
  const int x0 = tx * 32, y0 = ty * ROWS;
  const int co_blk = blockIdx.y;
  const int H = p.H, W = p.W;
  const bool xform = p.in_scale != nullptr;

  // fused producer coefficients; identity when there is none, so the staging code below is branch-free
  for (int c = tid; c < Cin; c += 256) {
    s_sc[c] = xform ? p.in_scale[b * Cin + c] : 1.f;
    s_sh[c] = xform ? p.in_shift[b * Cin + c] : 0.f;
  }
  const bool act = xform && p.in_act;

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int wc = wave % WC, wp = wave / WC;
  const int a_off = kh * TAPS * CO_T + wc * FC * 32 + l31;
  const int b_off = kh * PLANE + (wp * FP) * RS + l31 + T::XOFF;

  f32x16 acc[FC][FP];
#pragma unroll
  for (int i = 0; i < FC; ++i)
#pragma unroll
    for (int j = 0; j < FP; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float rin[NI];
  f32x4 rv[NV];
  f32x4 rw[NW4];
  unsigned okmask = 0;
  int goff[NI];    // scalar items: offset inside the channel plane (clamped into the image: loads are unconditional)
  int goff_v[NV];  // vector items: same, for the 16-byte aligned float4 of 4 consecutive pixels
  int loff_v[NV];  // vector items: LDS offset inside the stage
  int loff_h = 0;  // halo item: LDS offset inside the stage

  const float* wbase = p.w + (size_t)co_blk * Cin * TAPS * CO_T;
  const size_t HW = (size_t)H * W;

  // Per-thread staging coordinates do not depend on the stage.  Threads past the end of the item list re-stage the last
  // item (same address, same value), which keeps every load and LDS store unconditional (no divergent branches).
  if constexpr (!VEC) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int e = tid + 256 * i;
      e = e < T::IN_ELEMS ? e : T::IN_ELEMS - 1;
      const int c = e / PLANE;
      const int rem = e - c * PLANE;
      const int r = rem / RS;
      const int x = rem - r * RS;
      const int gy = y0 - HALO + r, gx = x0 - HALO + x;
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
      okmask |= (ok ? 1u : 0u) << i;
      goff[i] = ok ? gy * W + gx : 0;
    }
  } else {
    // the 32 interior pixels of a tile row are 8 aligned float4 (all-or-nothing inside the image since W % 4 == 0);
    // the halo columns are scalars
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int it = tid + 256 * i;
      it = it < T::NVI ? it : T::NVI - 1;
      const int c = it / (TROWS * 8);
      const int rem = it - c * (TROWS * 8);
      const int r = rem >> 3, q = rem & 7;
      const int gy = y0 - HALO + r, gx = x0 + 4 * q;
      const bool ok = gy >= 0 && gy < H && gx < W;
      okmask |= (ok ? 1u : 0u) << i;
      goff_v[i] = ok ? gy * W + gx : 0;
      loff_v[i] = c * PLANE + r * RS + (HALO ? 4 : 0) + 4 * q;
    }
    if (HALO) {
      const int it = tid < T::NHI ? tid : T::NHI - 1;
      const int c = it / (TROWS * 2);
      const int rem = it - c * (TROWS * 2);
      const int r = rem >> 1, side = rem & 1;
      const int gy = y0 - HALO + r, gx = side ? x0 + 32 : x0 - 1;
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
      okmask |= (ok ? 1u : 0u) << 31;
      goff[0] = ok ? gy * W + gx : 0;
      loff_h = c * PLANE + r * RS + (side ? 36 : 3);
    }
  }

  auto plane_of = [&](int cg) -> const float* {
    return (cg < p.C1) ? p.src1 + (size_t)(b * p.C1 + cg) * HW : p.src2 + (size_t)(b * p.C2 + (cg - p.C1)) * HW;
  };

  // global -> registers (stage c0 .. c0+KC-1)
  auto load_stage = [&](int c0) {
    if constexpr (!VEC) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int e = tid + 256 * i;
        e = e < T::IN_ELEMS ? e : T::IN_ELEMS - 1;
        rin[i] = plane_of(c0 + e / PLANE)[goff[i]];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int it = tid + 256 * i;
        it = it < T::NVI ? it : T::NVI - 1;
        rv[i] = *reinterpret_cast<const f32x4*>(plane_of(c0 + it / (TROWS * 8)) + goff_v[i]);
      }
      if (HALO) {
        const int it = tid < T::NHI ? tid : T::NHI - 1;
        rin[0] = plane_of(c0 + it / (TROWS * 2))[goff[0]];
      }
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wbase + (size_t)c0 * TAPS * CO_T);
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      int idx = tid + 256 * i;
      idx = idx < T::W_ELEMS / 4 ? idx : T::W_ELEMS / 4 - 1;
      rw[i] = wsrc[idx];
    }
  };

  auto xf1 = [&](float v, float sc, float sh, bool ok) -> float {
    v = v * sc + sh;
    const float sv = silu_f(v);
    v = act ? sv : v;
    return ok ? v : 0.f;   // zero padding is applied after the fused producer
  };

  // registers -> LDS, through the fused producer
  auto store_stage = [&](int c0) {
    if constexpr (!VEC) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int e = tid + 256 * i;
        e = e < T::IN_ELEMS ? e : T::IN_ELEMS - 1;
        const int cg = c0 + e / PLANE;
        s_in[e] = xf1(rin[i], s_sc[cg], s_sh[cg], (okmask >> i) & 1u);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        int it = tid + 256 * i;
        it = it < T::NVI ? it : T::NVI - 1;
        const int cg = c0 + it / (TROWS * 8);
        const float sc = s_sc[cg], sh = s_sh[cg];
        const bool ok = (okmask >> i) & 1u;
        f32x4 o;
        o[0] = xf1(rv[i][0], sc, sh, ok); o[1] = xf1(rv[i][1], sc, sh, ok);
        o[2] = xf1(rv[i][2], sc, sh, ok); o[3] = xf1(rv[i][3], sc, sh, ok);
        *reinterpret_cast<f32x4*>(s_in + loff_v[i]) = o;
      }
      if (HALO) {
        const int it = tid < T::NHI ? tid : T::NHI - 1;
        const int cg = c0 + it / (TROWS * 2);
        s_in[loff_h] = xf1(rin[0], s_sc[cg], s_sh[cg], (okmask >> 31) & 1u);
      }
    }
    f32x4* wdst = reinterpret_cast<f32x4*>(s_w);
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      int idx = tid + 256 * i;
      idx = idx < T::W_ELEMS / 4 ? idx : T::W_ELEMS / 4 - 1;
      wdst[idx] = rw[i];
    }
  };

  // MFMAs of one stage: k-step = (channel pair, tap); lane (kh, l31) supplies channel 2*cp+kh
  auto compute = [&]() {
    const float* sw = s_w + a_off;
    const float* si = s_in + b_off;
    if constexpr (PREF) {
      constexpr int NS = (KC / 2) * TAPS;
      float a[2][FC], bb[2][FP];
      auto ld = [&](int s2, int slot) {
        const int cp = s2 / TAPS, tap = s2 % TAPS, dy = tap / KS, dx = tap % KS;
#pragma unroll
        for (int i = 0; i < FC; ++i) a[slot][i] = sw[i * 32 + (cp * 2 * TAPS + tap) * CO_T];
#pragma unroll
        for (int j = 0; j < FP; ++j) bb[slot][j] = si[j * RS + cp * 2 * PLANE + dy * RS + dx];
      };
      ld(0, 0);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        if (s2 + 1 < NS) ld(s2 + 1, (s2 + 1) & 1);
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
          for (int j = 0; j < FP; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2 & 1][i], bb[s2 & 1][j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, FC + FP, 0);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        if (s2 + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, FC + FP, 0);   // LDS reads of k-step s2+1
        __builtin_amdgcn_sched_group_barrier(0x008, FC * FP, 0);                    // MFMAs of k-step s2
      }
    } else {
#pragma unroll 1
      for (int cp = 0; cp < KC / 2; ++cp) {
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const int dy = tap / KS, dx = tap % KS;
          float a[FC], bb[FP];
#pragma unroll
          for (int i = 0; i < FC; ++i) a[i] = sw[i * 32 + (cp * 2 * TAPS + tap) * CO_T];
#pragma unroll
          for (int j = 0; j < FP; ++j) bb[j] = si[j * RS + cp * 2 * PLANE + dy * RS + dx];
#pragma unroll
          for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int j = 0; j < FP; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  const int nstages = Cin / KC;
  const bool abl_stage = p.ablate & 4, abl_bar = p.ablate & 8;
  if constexpr (FC * FP <= 2) {
    // small tiles (coarse levels): optional chunked accumulation / split-K, see ConvArgs::kchunk_stages
    const int cs = p.kchunk_stages > 0 ? p.kchunk_stages : nstages;
    const bool splitk = gridDim.z > 1;
    const int ci0 = splitk ? (int)blockIdx.z * cs : 0;
    const int ci1 = splitk ? (ci0 + cs < nstages ? ci0 + cs : nstages) : nstages;
    f32x16 tot[FC][FP];
    bool first = true;
    load_stage(ci0 * KC);
    __syncthreads();  // s_sc / s_sh visible
    for (int ci = ci0; ci < ci1; ++ci) {
      store_stage(ci * KC);
      __syncthreads();
      if (ci + 1 < ci1) load_stage((ci + 1) * KC);   // in flight during the MFMAs
      compute();
      __syncthreads();
      if (p.kchunk_stages > 0 && !splitk && ((ci + 1) % cs == 0 || ci + 1 == ci1)) {   // end of a chunk: total += partial
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
          for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[i][j][r] = first ? acc[i][j][r] : tot[i][j][r] + acc[i][j][r]; acc[i][j][r] = 0.f; }
        first = false;
      }
    }
    if (p.kchunk_stages > 0 && !splitk) {
#pragma unroll
      for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j) acc[i][j] = tot[i][j];
    }
    if (splitk) {                                      // raw partial sums of this chunk; the reduce kernel runs the epilogue
      ConvArgs q = p;
      q.out = p.partial + (size_t)blockIdx.z * conv_partial_slab(p, H, W);
      q.bias = nullptr; q.bias2 = nullptr; q.res = nullptr; q.acc_scale = nullptr; q.out_scale = 1.f; q.stats_out = nullptr; q.amax_out = nullptr;
      conv_epilogue<T, FC, FP, WC>(q, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kh);
      return;                                          // conv_splitk_reduce_kernel follows
    }
  } else {
    load_stage(0);
    __syncthreads();  // s_sc / s_sh visible
    for (int ci = 0; ci < nstages; ++ci) {
      if (!abl_stage || ci == 0) store_stage(ci * KC);
      if (!abl_bar) __syncthreads();
      if (ci + 1 < nstages && !abl_stage) load_stage((ci + 1) * KC);   // in flight during the MFMAs
      compute();
      if (!abl_bar) __syncthreads();
    }
  }

  conv_epilogue<T, FC, FP, WC>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kh);
}

// Second half of a split-K convolution (ConvArgs::kchunk_stages): sums the chunks' partial sums in chunk order into the
// accumulator layout of the producing tile shape and runs the ordinary epilogue (bias, time embedding, residual, scale,
// GroupNorm partials, range bound).  grid = (tiles, output-channel blocks) of the tile shape <KS, WC, FC, FP>.
template <int KS, int WC, int FC, int FP, bool STAT4 = false>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvArgs p, int nchunks) {
  using T = ConvTile<KS, WC, FC, FP, 1>;
  constexpr int ROWS = T::ROWS, CO_T = T::CO_T;
  const int tid = threadIdx.x;
  const int tiles_xg = (p.W + 31) >> 5;
  const int tiles_y = (p.H + ROWS - 1) / ROWS;
  int b, ty, tx;
  conv_tile_of(p, (int)blockIdx.x, (int)gridDim.x, tiles_xg, tiles_y, b, ty, tx);
  if (p.rag_w) { if (!conv_ragged_adjust(p, b, tx)) return; }
  const int H = p.H, W = p.W;
  const int tiles_x = (W + 31) >> 5;
  const int co_blk = blockIdx.y;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int wc = wave % WC, wp = wave / WC;
  f32x16 acc[FC][FP];
  conv_splitk_sum<T, FC, FP, WC>(p, nchunks, b, co_blk, tx, ty, wc, wp, l31, kh, acc);
  // chunks of the fp16x2 kernel carry its per-utterance input scale (ConvArgs::xbound); the fp32 kernels' chunks carry none
  float as_mul = 1.0f;
  if (p.xbound) as_mul = 1.0f / h2_weight_scale(amax_read(p.xbound, b));
  conv_epilogue<T, FC, FP, WC, 0, false, STAT4, STAT4>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kh, as_mul);      // (STAT4 = the chunked fp16x2 kernel's reduce: per-channel factors)
}

// Direct (VALU) convolution: one thread per output pixel, CG output channels per thread.
// grid = (ceil(H*W/256), ceil(Cout/CG), B).  Weights are OIHW and wave-uniform (scalar loads).
template <int KS, int CG>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs p) {
  constexpr int TAPS = KS * KS, HALO = KS / 2;
  const int b = blockIdx.z;
  if (p.rag_w) conv_ragged_adjust(p, b, 0);      // blocks past the utterance's pixels find inb == false below
  const int H = p.H, W = p.W, HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * CG;
  const int Cin = p.C1 + p.C2;
  const bool inb = pix < HW;
  const int y = inb ? pix / W : 0, x = inb ? pix - (pix / W) * W : 0;
  float acc[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) acc[g] = 0.f;
  const bool xform = p.in_scale != nullptr;
  // Input channels in groups of CB with ALL CB*TAPS loads of a group in flight before the first use (left alone the compiler
  // runs load - wait - fma through one register: 36 serial memory round trips for the 4-channel entry convolution, which made
  // a 256 x 512 image at batch 1 ten times slower than its output write).  Same summation order: channel, then tap.
  constexpr int CB = 4;
  int offs[TAPS];
  bool oks[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int gy = y + t / KS - HALO, gx = x + t % KS - HALO;
    oks[t] = inb && gy >= 0 && gy < H && gx >= 0 && gx < W;
    // unconditional load from a clamped address, masked afterwards: a predicated load is a branch plus a full vmcnt wait
    const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    offs[t] = gyc * W + gxc;
  }
  for (int c0 = 0; c0 < Cin; c0 += CB) {
    float v[CB][TAPS];
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) {
      const int c = c0 + cc < Cin ? c0 + cc : Cin - 1;
      const float* plane = (c < p.C1) ? p.src1 + (size_t)(b * p.C1 + c) * HW : p.src2 + (size_t)(b * p.C2 + (c - p.C1)) * HW;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) v[cc][t] = plane[offs[t]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) {
      const int c = c0 + cc;
      if (c < Cin) {
        if (xform) {
          const float sc = p.in_scale[b * Cin + c], sh = p.in_shift[b * Cin + c];
#pragma unroll
          for (int t = 0; t < TAPS; ++t) { float u = v[cc][t] * sc + sh; if (p.in_act) u = silu_f(u); v[cc][t] = u; }
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) v[cc][t] = oks[t] ? v[cc][t] : 0.f;
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const int co = co0 + g;
          if (co < p.Cout) {
            const float* wr = p.w + ((size_t)co * Cin + c) * TAPS;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[g] = fmaf(wr[t], v[cc][t], acc[g]);
          }
        }
      }
    }
  }
  const float* b2 = nullptr;
  if (p.bias2) {
    const int step = p.step_ptr ? *p.step_ptr : 0;
    b2 = p.bias2 + (size_t)step * p.bias2_sstride + (size_t)b * p.bias2_bstride;
  }
  float vmax = 0.f;
  float rr[CG];
  if (p.res) {                         // residual reads in one batch, from clamped addresses
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      const int co = co0 + g < p.Cout ? co0 + g : p.Cout - 1;
      rr[g] = p.res[(size_t)(b * p.Cout + co) * HW + (inb ? pix : 0)];
    }
  } else {
#pragma unroll
    for (int g = 0; g < CG; ++g) rr[g] = 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const int co = co0 + g;
    if (co < p.Cout) {                 // wave-uniform
      float v = acc[g];
      if (p.bias) v += p.bias[co];
      if (b2) v += b2[co];
      if (p.res) v += rr[g];
      v *= p.out_scale;
      if (inb) {
        p.out[(size_t)(b * p.Cout + co) * HW + pix] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  }
  if (p.amax_out) {      // every lane of the wave is still here (no early return above)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((threadIdx.x & 63) == 0)
      drt_atomic_max_nonneg(p.amax_out + b * kAmaxSpread + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (kAmaxSpread - 1)), vmax);
  }
}

// ---- host-side helpers ---------------------------------------------------------------------------------

// Which MFMA tile a layer uses.  co_t in {32,64,128}; rows in {8,4}.
struct ConvPlan { int co_t; int rows; bool mfma; };
inline int conv_plan_wp(int co_t) { return co_t == 32 ? 4 : 2; }            // pixel-waves per workgroup (ConvTile::WP)
inline int conv_plan_nsub(int H, int W, int rows = 1) { return ((H + rows - 1) / rows) * ((W + 31) / 32); }   // statistics sub-tiles per image (tile-independent)

inline ConvPlan choose_conv_plan(int ks, int cin, int cout, int H, int W) {
  ConvPlan pl{0, 0, false};
  const int kc = (ks == 3) ? 8 : 32;
  if ((ks != 1 && ks != 3) || cin % kc != 0 || cin > 512) return pl;
  // thin outputs (the C->4 pyramid convolutions) run on a zero-padded 32-channel MFMA tile when the reduction is deep
  // enough to pay for it; thin inputs (4->C) stay on the direct kernel
  if (cout % 32 != 0 && !(cout < 32 && cin >= 64)) return pl;
  pl.mfma = true;
  pl.co_t = (cout % 128 == 0) ? 128 : (cout % 64 == 0 ? 64 : 32);
  pl.rows = (H >= 8) ? 8 : 4;
  (void)W;
  return pl;
}

// Packed weight size (floats) for conv_mfma given OIHW source [cout][cin][ks][ks].
inline size_t packed_weight_elems(int ks, int cin, int cout, int co_t) {
  const int nblk = (cout + co_t - 1) / co_t;
  return (size_t)nblk * cin * ks * ks * co_t;
}

}  // namespace sgmse

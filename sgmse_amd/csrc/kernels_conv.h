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
//   Optional fused producer: GroupNorm affine (+SiLU) applied while the tile is written to LDS
//   (layerspp.py:243,264: act(GroupNorm(x)) feeding Conv_0 / Conv_1), zero padding applied after it.
// conv_direct: VALU direct convolution for the thin layers (4->C, C->4) and any shape the MFMA kernel does not
//   cover; same fused producer/epilogue.
#pragma once
#include <sgmse_devrt.h>

namespace sgmse {

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
  float out_scale;         // out = (acc + bias + bias2 + res) * out_scale
  float* out;
  int Cout, B, H, W;
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

template <int KS, int WC, int FC, int FP>
struct ConvTile {
  static constexpr int WP = 4 / WC;                 // waves along pixels
  static constexpr int CO_T = WC * FC * 32;         // output channels per workgroup
  static constexpr int ROWS = WP * FP;              // image rows per workgroup (each pixel fragment = 32 px of a row)
  static constexpr int KC = (KS == 3) ? 8 : 32;     // input channels per LDS stage
  static constexpr int HALO = KS / 2;
  static constexpr int RS = 32 + 2 * HALO;          // LDS row stride
  static constexpr int PLANE = (ROWS + 2 * HALO) * RS;
  static constexpr int TAPS = KS * KS;
  static constexpr int IN_ELEMS = KC * PLANE;
  static constexpr int W_ELEMS = KC * TAPS * CO_T;
  static constexpr int NI = (IN_ELEMS + 255) / 256;
  static constexpr int NW4 = (W_ELEMS / 4 + 255) / 256;
};

template <int KS, int WC, int FC, int FP, int MINW>
__global__ __launch_bounds__(256, MINW) void conv_mfma_kernel(ConvArgs p) {
  using T = ConvTile<KS, WC, FC, FP>;
  constexpr int CO_T = T::CO_T, ROWS = T::ROWS, KC = T::KC, HALO = T::HALO, RS = T::RS, PLANE = T::PLANE,
                TAPS = T::TAPS, NI = T::NI, NW4 = T::NW4;
  __shared__ float s_in[T::IN_ELEMS];
  __shared__ float s_w[T::W_ELEMS];
  __shared__ float s_sc[512];
  __shared__ float s_sh[512];

  const int tid = threadIdx.x;
  const int Cin = p.C1 + p.C2;
  const int tiles_x = (p.W + 31) >> 5;
  const int tiles_y = (p.H + ROWS - 1) / ROWS;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int x0 = tx * 32, y0 = ty * ROWS;
  const int co_blk = blockIdx.y;
  const int H = p.H, W = p.W;
  const bool xform = p.in_scale != nullptr;

  if (xform) {
    for (int c = tid; c < Cin; c += 256) {
      s_sc[c] = p.in_scale[b * Cin + c];
      s_sh[c] = p.in_shift[b * Cin + c];
    }
  }

  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
  const int wc = wave % WC, wp = wave / WC;
  const int a_off = kh * TAPS * CO_T + wc * FC * 32 + l31;
  const int b_off = kh * PLANE + (wp * FP) * RS + l31;

  f32x16 acc[FC][FP];
#pragma unroll
  for (int i = 0; i < FC; ++i)
#pragma unroll
    for (int j = 0; j < FP; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float rin[NI];
  f32x4 rw[NW4];
  unsigned okmask = 0;

  const float* wbase = p.w + (size_t)co_blk * Cin * TAPS * CO_T;

  // Per-thread staging coordinates do not depend on the chunk: precompute the validity mask once.
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + 256 * i;
    const int c = e / PLANE;
    const int rem = e - c * PLANE;
    const int r = rem / RS;
    const int x = rem - r * RS;
    const int gy = y0 - HALO + r, gx = x0 - HALO + x;
    const bool ok = (e < T::IN_ELEMS) && gy >= 0 && gy < H && gx >= 0 && gx < W;
    okmask |= (ok ? 1u : 0u) << i;
  }

  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tid + 256 * i;
      const int c = e / PLANE;
      const int rem = e - c * PLANE;
      const int r = rem / RS;
      const int x = rem - r * RS;
      const int gy = y0 - HALO + r, gx = x0 - HALO + x;
      const int cg = c0 + c;
      float v = 0.f;
      if ((okmask >> i) & 1u) {
        const float* sp = (cg < p.C1) ? p.src1 + ((size_t)(b * p.C1 + cg) * H + gy) * W + gx
                                      : p.src2 + ((size_t)(b * p.C2 + (cg - p.C1)) * H + gy) * W + gx;
        v = *sp;
      }
      rin[i] = v;
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wbase + (size_t)c0 * TAPS * CO_T);
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int idx = tid + 256 * i;
      if (idx < T::W_ELEMS / 4) rw[i] = wsrc[idx];
    }
  };

  auto store_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tid + 256 * i;
      if (e < T::IN_ELEMS) {
        float v = rin[i];
        if (xform && ((okmask >> i) & 1u)) {
          const int cg = c0 + e / PLANE;
          v = v * s_sc[cg] + s_sh[cg];
          if (p.in_act) v = silu_f(v);
        }
        s_in[e] = v;
      }
    }
    f32x4* wdst = reinterpret_cast<f32x4*>(s_w);
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int idx = tid + 256 * i;
      if (idx < T::W_ELEMS / 4) wdst[idx] = rw[i];
    }
  };

  const int nchunks = Cin / KC;
  load_chunk(0);
  __syncthreads();  // s_sc / s_sh visible
  for (int ci = 0; ci < nchunks; ++ci) {
    store_chunk(ci * KC);
    __syncthreads();
    if (ci + 1 < nchunks) load_chunk((ci + 1) * KC);
#pragma unroll 1
    for (int cp = 0; cp < KC / 2; ++cp) {
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int dy = tap / KS, dx = tap % KS;
        float a[FC], bb[FP];
#pragma unroll
        for (int i = 0; i < FC; ++i) a[i] = s_w[a_off + i * 32 + (cp * 2 * TAPS + tap) * CO_T];
#pragma unroll
        for (int j = 0; j < FP; ++j) bb[j] = s_in[b_off + j * RS + cp * 2 * PLANE + dy * RS + dx];
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
          for (int j = 0; j < FP; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue
  const int x = x0 + l31;
  const float* b2 = nullptr;
  if (p.bias2) {
    const int step = p.step_ptr ? *p.step_ptr : 0;
    b2 = p.bias2 + (size_t)step * p.bias2_sstride + (size_t)b * p.bias2_bstride;
  }
#pragma unroll
  for (int i = 0; i < FC; ++i) {
    const int co_base = co_blk * CO_T + (wc * FC + i) * 32 + 4 * kh;
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co_base + (r & 3) + 8 * (r >> 2);
      float t = 0.f;
      if (co < p.Cout) {
        if (p.bias) t += p.bias[co];
        if (b2) t += b2[co];
      }
      bv[r] = t;
    }
#pragma unroll
    for (int j = 0; j < FP; ++j) {
      const int y = y0 + wp * FP + j;
      if (y < H && x < W) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_base + (r & 3) + 8 * (r >> 2);
          if (co < p.Cout) {
            const size_t o = ((size_t)(b * p.Cout + co) * H + y) * W + x;
            float v = acc[i][j][r] + bv[r];
            if (p.res) v += p.res[o];
            p.out[o] = v * p.out_scale;
          }
        }
      }
    }
  }
}

// Direct (VALU) convolution: one thread per output pixel, CG output channels per thread.
// grid = (ceil(H*W/256), ceil(Cout/CG), B).  Weights are OIHW and wave-uniform (scalar loads).
template <int KS, int CG>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs p) {
  constexpr int TAPS = KS * KS, HALO = KS / 2;
  const int H = p.H, W = p.W, HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * CG;
  const int b = blockIdx.z;
  const int Cin = p.C1 + p.C2;
  const bool inb = pix < HW;
  const int y = inb ? pix / W : 0, x = inb ? pix - (pix / W) * W : 0;
  float acc[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) acc[g] = 0.f;
  const bool xform = p.in_scale != nullptr;
  for (int c = 0; c < Cin; ++c) {
    const float* plane = (c < p.C1) ? p.src1 + (size_t)(b * p.C1 + c) * HW : p.src2 + (size_t)(b * p.C2 + (c - p.C1)) * HW;
    float sc = 1.f, sh = 0.f;
    if (xform) { sc = p.in_scale[b * Cin + c]; sh = p.in_shift[b * Cin + c]; }
    float v[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int gy = y + t / KS - HALO, gx = x + t % KS - HALO;
      const bool ok = inb && gy >= 0 && gy < H && gx >= 0 && gx < W;
      float u = 0.f;
      if (ok) {
        u = plane[gy * W + gx];
        if (xform) { u = u * sc + sh; if (p.in_act) u = silu_f(u); }
      }
      v[t] = u;
    }
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      const int co = co0 + g;
      if (co < p.Cout) {
        const float* wr = p.w + ((size_t)co * Cin + c) * TAPS;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[g] = fmaf(wr[t], v[t], acc[g]);
      }
    }
  }
  if (!inb) return;
  const float* b2 = nullptr;
  if (p.bias2) {
    const int step = p.step_ptr ? *p.step_ptr : 0;
    b2 = p.bias2 + (size_t)step * p.bias2_sstride + (size_t)b * p.bias2_bstride;
  }
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const int co = co0 + g;
    if (co < p.Cout) {
      const size_t o = (size_t)(b * p.Cout + co) * HW + pix;
      float v = acc[g];
      if (p.bias) v += p.bias[co];
      if (b2) v += b2[co];
      if (p.res) v += p.res[o];
      p.out[o] = v * p.out_scale;
    }
  }
}

// ---- host-side helpers ---------------------------------------------------------------------------------

// Which MFMA tile a layer uses.  co_t in {32,64,128}; rows in {8,4}.
struct ConvPlan { int co_t; int rows; bool mfma; };

inline ConvPlan choose_conv_plan(int ks, int cin, int cout, int H, int W) {
  ConvPlan pl{0, 0, false};
  const int kc = (ks == 3) ? 8 : 32;
  if ((ks != 1 && ks != 3) || cin % kc != 0 || cin > 512 || cout % 32 != 0) return pl;
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

// dst[blk][c][tap][j] = src[blk*co_t + j][c][tap]   (zero where blk*co_t + j >= cout)
inline void pack_conv_weights(const float* src, float* dst, int ks, int cin, int cout, int co_t) {
  const int taps = ks * ks, nblk = (cout + co_t - 1) / co_t;
  for (int blk = 0; blk < nblk; ++blk)
    for (int c = 0; c < cin; ++c)
      for (int t = 0; t < taps; ++t)
        for (int j = 0; j < co_t; ++j) {
          const int co = blk * co_t + j;
          dst[(((size_t)blk * cin + c) * taps + t) * co_t + j] = co < cout ? src[((size_t)co * cin + c) * taps + t] : 0.f;
        }
}

// Register-allocation target of the 128x256 tile: 2 = two workgroups per CU (256 VGPRs), 1 = one (512 VGPRs).
// Chosen once per process from SGMSE_CONV_MINW (measurement knob; default 2).
inline int conv_minw() {
  static int v = [] { const char* e = getenv("SGMSE_CONV_MINW"); return (e && e[0] == '1') ? 1 : 2; }();
  return v;
}

template <int KS, int WC, int FC, int FP>
inline void launch_conv_mfma_t(const ConvArgs& a, drt::stream_t st) {
  using T = ConvTile<KS, WC, FC, FP>;
  const int tiles = a.B * ((a.H + T::ROWS - 1) / T::ROWS) * ((a.W + 31) / 32);
  dim3 grid(tiles, (a.Cout + T::CO_T - 1) / T::CO_T, 1);
  if (FC * FP == 8 && conv_minw() == 1) DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 1>), grid, dim3(256), st, a);
  else DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 2>), grid, dim3(256), st, a);
}

inline void launch_conv_mfma(const ConvArgs& a, int ks, const ConvPlan& pl, drt::stream_t st) {
#define SGMSE_CONV_CASE(KS_, CO_, ROWS_, WC_, FC_, FP_) \
  if (ks == KS_ && pl.co_t == CO_ && pl.rows == ROWS_) { launch_conv_mfma_t<KS_, WC_, FC_, FP_>(a, st); return; }
  SGMSE_CONV_CASE(3, 128, 8, 2, 2, 4)
  SGMSE_CONV_CASE(3, 64, 8, 2, 1, 4)
  SGMSE_CONV_CASE(3, 32, 8, 1, 1, 2)
  SGMSE_CONV_CASE(3, 128, 4, 2, 2, 2)
  SGMSE_CONV_CASE(3, 64, 4, 2, 1, 2)
  SGMSE_CONV_CASE(3, 32, 4, 1, 1, 1)
  SGMSE_CONV_CASE(1, 128, 8, 2, 2, 4)
  SGMSE_CONV_CASE(1, 64, 8, 2, 1, 4)
  SGMSE_CONV_CASE(1, 32, 8, 1, 1, 2)
  SGMSE_CONV_CASE(1, 128, 4, 2, 2, 2)
  SGMSE_CONV_CASE(1, 64, 4, 2, 1, 2)
  SGMSE_CONV_CASE(1, 32, 4, 1, 1, 1)
#undef SGMSE_CONV_CASE
}

inline void launch_conv_direct(const ConvArgs& a, int ks, drt::stream_t st) {
  const int HW = a.H * a.W;
  if (a.Cout <= 4) {
    dim3 grid((HW + 255) / 256, 1, a.B);
    if (ks == 3) DRT_LAUNCH((conv_direct_kernel<3, 4>), grid, dim3(256), st, a);
    else DRT_LAUNCH((conv_direct_kernel<1, 4>), grid, dim3(256), st, a);
  } else {
    dim3 grid((HW + 255) / 256, (a.Cout + 15) / 16, a.B);
    if (ks == 3) DRT_LAUNCH((conv_direct_kernel<3, 16>), grid, dim3(256), st, a);
    else DRT_LAUNCH((conv_direct_kernel<1, 16>), grid, dim3(256), st, a);
  }
}

}  // namespace sgmse

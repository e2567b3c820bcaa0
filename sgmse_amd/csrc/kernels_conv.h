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
#include <type_traits>

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
  // optional GroupNorm statistics of the stored output, fused into the epilogue: per (b, co, sub-tile) partial
  // {sum, sum of squares}; sub-tile = (tile index in the image) * WP + (pixel-wave index).  Deterministic (no atomics).
  float* stats_out;
  int stats_nsub;
  // Start-up stagger (units of s_sleep 127 ~ 8k cycles) applied to the workgroups that fill the second residency slot
  // of each CU: identical workgroups otherwise run their staging / prologue / epilogue phases in lock-step on both slots.
  int stagger;
};

// SiLU x*sigmoid(x) (nn.SiLU, reference layers.py:38-39) on the hardware exp2/rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each,
// relative error of the result < 4e-7, far inside the 1e-5 per-op parity gate) -- the fused producers evaluate it for every
// staged element, so a libm-accurate expf + IEEE divide (~25 VALU instructions) would dominate the staging phase.
__device__ __forceinline__ float silu_f(float v) { return v * __frcp_rn(1.0f + __expf(-v)); }
__device__ __forceinline__ float silu_precise_f(float v) { return v / (1.0f + expf(-v)); }   // time-embedding MLP (tiny)

template <int KS, int WC, int FC, int FP, int DB = 0, int VEC = 0>
struct ConvTile {
  static constexpr int WP = 4 / WC;                 // waves along pixels
  static constexpr int CO_T = WC * FC * 32;         // output channels per workgroup
  static constexpr int ROWS = WP * FP;              // image rows per workgroup (each pixel fragment = 32 px of a row)
  static constexpr int KC = (KS == 3) ? 8 : 32;     // input-channel granularity of the layer (Cin % KC == 0)
  static constexpr int NBUF = DB ? 2 : 1;           // LDS stages
  static constexpr int KCH = KC / NBUF;             // input channels per LDS stage
  static constexpr int HALO = KS / 2;
  // LDS row: scalar staging packs [halo | 32 px | halo] (stride 34 / 32); vector staging keeps the 32 interior pixels
  // 16-byte aligned at column 4 with the halo columns at 3 and 36 (stride 40) so they can be written as b128
  static constexpr int RS = (VEC && HALO) ? 40 : 32 + 2 * HALO;
  static constexpr int XOFF = (VEC && HALO) ? 3 : 0;   // column of the left-most tap of pixel 0
  static constexpr int TROWS = ROWS + 2 * HALO;
  static constexpr int PLANE = TROWS * RS;
  static constexpr int NVI = KCH * TROWS * 8;          // float4 items per stage (vector staging)
  static constexpr int NV = (NVI + 255) / 256;
  static constexpr int NHI = KCH * TROWS * 2 * HALO;   // halo scalars per stage (vector staging)
  static constexpr int TAPS = KS * KS;
  static constexpr int IN_ELEMS = KCH * PLANE;
  static constexpr int W_ELEMS = KCH * TAPS * CO_T;
  static constexpr int NI = (IN_ELEMS + 255) / 256;
  static constexpr int NW4 = (W_ELEMS / 4 + 255) / 256;
};

// DB = 0: one LDS stage of KC channels, two barriers per stage, next stage prefetched into registers during the MFMAs.
// DB = 1: two LDS stages of KC/2 channels (same LDS footprint): the stage after next is fetched into registers and the
//         next stage written to the idle LDS buffer inside the same basic block as the current stage's MFMAs (one
//         barrier per stage), so a wave's own staging work sits in the shadow of its own MFMAs.
// SCHED: 0 compiler default, 1 = __builtin_amdgcn_iglp_opt(0) on the MFMA block, 3 = s_setprio(1) around it,
//        4 = explicit operand prefetch one k-step ahead, order pinned with sched_group_barrier.
// WS = 1: wave-specialised workgroup of 512 threads (requires DB): waves 0-3 only read LDS and issue MFMAs, waves 4-7 only
//         stage (global loads, fused GroupNorm/SiLU, LDS writes) the next K-stage into the idle buffer.  The matrix pipe of
//         every SIMD is then fed by a wave that never leaves its MFMA stream, while its partner wave on the same SIMD uses the
//         VALU/LDS/VMEM issue slots in between (the two blocks-per-CU arrangement of WS = 0 runs its staging phases in
//         lock-step and leaves the pipe ~20 % idle: profiles/r01_pmc_conv_microbench.json).
template <int KS, int WC, int FC, int FP, int MINW, int DB, int SCHED, int VEC, int WS = 0>
__global__ __launch_bounds__(WS ? 512 : 256, MINW) void conv_mfma_kernel(ConvArgs p) {
  static_assert(!WS || DB, "wave specialisation needs the two-stage LDS layout");
  using T = ConvTile<KS, WC, FC, FP, DB, VEC>;
  constexpr int CO_T = T::CO_T, ROWS = T::ROWS, KCH = T::KCH, HALO = T::HALO, RS = T::RS, PLANE = T::PLANE,
                TAPS = T::TAPS, NI = VEC ? 1 : T::NI, NW4 = T::NW4, NBUF = T::NBUF, NV = VEC ? T::NV : 1,
                TROWS = T::TROWS;
  // the two stages are separate objects so that the compiler knows LDS writes of one never alias reads of the other
  __shared__ float s_in0[T::IN_ELEMS];
  __shared__ float s_w0[T::W_ELEMS];
  __shared__ float s_in1[DB ? T::IN_ELEMS : 4];
  __shared__ float s_w1[DB ? T::W_ELEMS : 4];
  __shared__ float s_sc[512];
  __shared__ float s_sh[512];

  constexpr int NT = WS ? 512 : 256;
  const bool producer = WS && threadIdx.x >= 256;
  const int tid = threadIdx.x & 255;   // staging index (producers) / MFMA index (consumers)
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

  if (p.stagger > 0) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    if (lin >= 256u && lin < 512u)
      for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // fused producer coefficients; identity when there is none, so the staging code below is branch-free
  for (int c = threadIdx.x; c < Cin; c += NT) {
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

  // Per-thread staging coordinates do not depend on the chunk.  Threads past the end of the item list re-stage the last
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
    // vector staging (W % 4 == 0): the 32 interior pixels of a tile row are 8 aligned float4; halo columns are scalars
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
      int it = tid < T::NHI ? tid : T::NHI - 1;
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

  auto load_chunk = [&](int c0) {
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
    return ok ? v : 0.f;
  };

  auto store_chunk = [&](int c0, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    float* din = BUF ? s_in1 : s_in0;
    if constexpr (!VEC) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        int e = tid + 256 * i;
        e = e < T::IN_ELEMS ? e : T::IN_ELEMS - 1;
        const int cg = c0 + e / PLANE;
        din[e] = xf1(rin[i], s_sc[cg], s_sh[cg], (okmask >> i) & 1u);
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
        *reinterpret_cast<f32x4*>(din + loff_v[i]) = o;
      }
      if (HALO) {
        const int it = tid < T::NHI ? tid : T::NHI - 1;
        const int cg = c0 + it / (TROWS * 2);
        din[loff_h] = xf1(rin[0], s_sc[cg], s_sh[cg], (okmask >> 31) & 1u);
      }
    }
    f32x4* wdst = reinterpret_cast<f32x4*>(BUF ? s_w1 : s_w0);
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      int idx = tid + 256 * i;
      idx = idx < T::W_ELEMS / 4 ? idx : T::W_ELEMS / 4 - 1;
      wdst[idx] = rw[i];
    }
  };

  auto compute = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    const float* sw = (BUF ? s_w1 : s_w0) + a_off;
    const float* si = (BUF ? s_in1 : s_in0) + b_off;
    if (SCHED == 1) __builtin_amdgcn_iglp_opt(0);
    if (SCHED == 3 && !WS) __builtin_amdgcn_s_setprio(1);
    if (SCHED == 4) {
      // Explicit operand pipeline: the LDS reads of k-step s+1 are issued before the MFMAs of k-step s, so a wave
      // never parks on lgkmcnt between MFMA groups (two co-resident waves otherwise reach their LDS waits in
      // lock-step and leave the matrix pipe idle).  The sched_group_barrier template pins that order.
      constexpr int NS = (KCH / 2) * TAPS;
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
        if (s2 + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, FC + FP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, FC * FP, 0);
      }
      return;
    }
    constexpr int UNR = DB ? KCH / 2 : 1;
#pragma unroll UNR
    for (int cp = 0; cp < KCH / 2; ++cp) {
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
    if (SCHED == 3 && !WS) __builtin_amdgcn_s_setprio(0);
  };

  const int nchunks = Cin / KCH;
  if (!WS || producer) load_chunk(0);
  __syncthreads();  // s_sc / s_sh visible
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  if (WS) {
    if (producer) { store_chunk(0, B0{}); }
    __syncthreads();
    if (SCHED == 3 && !producer) __builtin_amdgcn_s_setprio(1);
    for (int ci = 0; ci < nchunks; ci += 2) {
      if (producer) { if (ci + 1 < nchunks) { load_chunk((ci + 1) * KCH); store_chunk((ci + 1) * KCH, B1{}); } }
      else compute(B0{});
      __syncthreads();
      if (ci + 1 < nchunks) {
        if (producer) { if (ci + 2 < nchunks) { load_chunk((ci + 2) * KCH); store_chunk((ci + 2) * KCH, B0{}); } }
        else compute(B1{});
        __syncthreads();
      }
    }
    if (producer) return;
  } else if (DB) {
    store_chunk(0, B0{});  } else {
    for (int ci = 0; ci < nchunks; ++ci) {
      store_chunk(ci * KCH, B0{});
      __syncthreads();
      if (ci + 1 < nchunks) load_chunk((ci + 1) * KCH);
      compute(B0{});
      __syncthreads();
    }
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
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[r] = 0.f; ssq[r] = 0.f; }
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
            v *= p.out_scale;
            p.out[o] = v;
            ssum[r] += v;
            ssq[r] += v * v;
          }
        }
      }
    }
    if (p.stats_out) {   // wave-uniform
      const int sub = (ty * tiles_x + tx) * T::WP + wp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a1 = ssum[r], a2 = ssq[r];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) { a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
        const int co = co_base + (r & 3) + 8 * (r >> 2);
        if (l31 == 0 && co < p.Cout) {
          float* so = p.stats_out + ((size_t)(b * p.Cout + co) * p.stats_nsub + sub) * 2;
          so[0] = a1; so[1] = a2;
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
inline int conv_plan_wp(int co_t) { return co_t == 32 ? 4 : 2; }            // pixel-waves per workgroup (ConvTile::WP)
inline int conv_plan_nsub(int co_t, int rows, int H, int W) {                 // statistics sub-tiles per image
  return ((H + rows - 1) / rows) * ((W + 31) / 32) * conv_plan_wp(co_t);
}

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

// Kernel variant of the MFMA convolution, chosen once per process from SGMSE_CONV_VARIANT (measurement knob):
//   0 default | 1 double-buffered LDS stages | 4 operand prefetch pinned with sched_group_barrier | 5 = 1+4 |
//   6 s_setprio around the MFMA block | 8 wave-specialised (4 MFMA waves + 4 staging waves) | 9 = 8 + s_setprio for the
//   MFMA waves | 10 = 8 + operand prefetch | +256 scalar input staging.  (iglp_opt(1) crashes hipcc 7.2; iglp_opt(0) and a
//   one-workgroup-per-CU register target measured no better, see DESIGN.md.)
// Variants other than the default are compiled for the 128x256 tiles only.
#ifndef SGMSE_CONV_DEFAULT_STAGGER
#define SGMSE_CONV_DEFAULT_STAGGER 0
#endif
#ifndef SGMSE_CONV_DEFAULT_VARIANT
#define SGMSE_CONV_DEFAULT_VARIANT 4
#endif
inline int conv_variant() {
  static int v = [] { const char* e = getenv("SGMSE_CONV_VARIANT"); return e ? atoi(e) : SGMSE_CONV_DEFAULT_VARIANT; }();
  return v;
}

template <int KS, int WC, int FC, int FP, int VEC>
inline void launch_conv_mfma_v(const ConvArgs& a, drt::stream_t st, int variant) {
  using T = ConvTile<KS, WC, FC, FP>;
  const int tiles = a.B * ((a.H + T::ROWS - 1) / T::ROWS) * ((a.W + 31) / 32);
  dim3 grid(tiles, (a.Cout + T::CO_T - 1) / T::CO_T, 1);
  if (variant < 0) variant = conv_variant();
  if constexpr (FC * FP == 8) {
    switch (variant) {
#define SGMSE_V(ID, MINW_, DB_, SCHED_) \
      case ID: DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, MINW_, DB_, SCHED_, VEC, 0>), grid, dim3(256), st, a); return;
      SGMSE_V(1, 2, 1, 0) SGMSE_V(4, 2, 0, 4) SGMSE_V(5, 2, 1, 4) SGMSE_V(6, 2, 0, 3)
#undef SGMSE_V
      case 8: DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 2, 1, 0, VEC, 1>), grid, dim3(512), st, a); return;
      case 9: DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 2, 1, 3, VEC, 1>), grid, dim3(512), st, a); return;
      case 10: DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 2, 1, 4, VEC, 1>), grid, dim3(512), st, a); return;
      default: break;
    }
  }
  DRT_LAUNCH((conv_mfma_kernel<KS, WC, FC, FP, 2, 0, 0, VEC, 0>), grid, dim3(256), st, a);
}

// variant bit 8 (256): scalar (element-wise) input staging even when the row length allows float4 staging
inline int conv_stagger() {
  static int v = [] { const char* e = getenv("SGMSE_CONV_STAGGER"); return e ? atoi(e) : SGMSE_CONV_DEFAULT_STAGGER; }();
  return v;
}

template <int KS, int WC, int FC, int FP>
inline void launch_conv_mfma_t(const ConvArgs& a_in, drt::stream_t st, int variant = -1) {
  if (variant < 0) variant = conv_variant();
  ConvArgs a = a_in;
  if (variant & 1024) { a.stagger = (variant >> 11) & 31; variant &= 1023; }   // microbench: stagger in bits 11..15
  else a.stagger = conv_stagger();
  const bool vec = (a.W % 4 == 0) && !(variant & 256) && a.src1 != nullptr &&
                   (reinterpret_cast<uintptr_t>(a.src1) % 16 == 0) && (a.src2 == nullptr || reinterpret_cast<uintptr_t>(a.src2) % 16 == 0);
  if (vec) launch_conv_mfma_v<KS, WC, FC, FP, 1>(a, st, variant & 255);
  else launch_conv_mfma_v<KS, WC, FC, FP, 0>(a, st, variant & 255);
}

inline void launch_conv_mfma(const ConvArgs& a, int ks, const ConvPlan& pl, drt::stream_t st, int variant = -1) {
#define SGMSE_CONV_CASE(KS_, CO_, ROWS_, WC_, FC_, FP_) \
  if (ks == KS_ && pl.co_t == CO_ && pl.rows == ROWS_) { launch_conv_mfma_t<KS_, WC_, FC_, FP_>(a, st, variant); return; }
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

// Software-pipelined variant of the MFMA convolution (same tiles, layouts, fused producer and epilogue as
// conv_mfma_kernel in kernels_conv.h).
//
// PMC on the baseline kernel (profiles/r01_pmc_conv_microbench.json) shows the matrix pipe ~80 % busy and every wave
// parked ~15 % of its time at the two barriers and the staging phase of each K-stage: the compiler keeps global loads,
// the fused GroupNorm/SiLU transform and the LDS writes in their own basic blocks in front of the MFMA stream, and the
// two co-resident workgroups run those phases in lock-step.  Here the K-stage is split in two LDS buffers of KC/2
// channels and the staging of stage s+1 (registers -> LDS) and s+2 (global -> registers) is cut into per-item tasks
// that are placed, in source order and fenced with sched_barrier, BETWEEN the MFMA groups of stage s.  Everything in the
// steady state is one branch-free basic block with one workgroup barrier per stage.
#pragma once
#include "kernels_conv.h"

namespace sgmse {

template <int KS, int WC, int FC, int FP>
__global__ __launch_bounds__(256, 2) void conv_mfma_pipe_kernel(ConvArgs p) {
  using T = ConvTile<KS, WC, FC, FP, 1>;
  constexpr int CO_T = T::CO_T, ROWS = T::ROWS, HALO = T::HALO, RS = T::RS, PLANE = T::PLANE, TAPS = T::TAPS, TROWS = T::TROWS;
  constexpr int KH = T::KC / 2;                       // channels per LDS stage
  constexpr int IN_ELEMS = KH * PLANE, W_ELEMS = KH * TAPS * CO_T;
  constexpr int NVI = KH * TROWS * 8, NV = (NVI + 255) / 256;       // float4 input items per stage / per thread
  constexpr int NHI = KH * TROWS * 2 * HALO;                        // halo scalars per stage (<= 256)
  constexpr int NW4 = (W_ELEMS / 4 + 255) / 256;                    // float4 weight items per thread
  constexpr int NTASK = NV + (HALO ? 1 : 0) + NW4;
  constexpr int NS = (KH / 2) * TAPS;                               // k-steps per stage
  static_assert(NTASK <= NS - 1, "not enough MFMA groups to hide the staging tasks");
  static_assert(NHI <= 256, "halo items must fit one per thread");

  __shared__ float s_in0[IN_ELEMS];
  __shared__ float s_in1[IN_ELEMS];
  __shared__ float s_w0[W_ELEMS];
  __shared__ float s_w1[W_ELEMS];
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

  // staging registers (one stage) and the stage-independent per-thread item coordinates
  f32x4 rv[NV];
  f32x4 rw[NW4];
  float rh = 0.f;
  unsigned okmask = 0;
  int goff_v[NV], loff_v[NV], cv[NV];
  int goff_h = 0, loff_h = 0, ch = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int it = tid + 256 * i;
    it = it < NVI ? it : NVI - 1;
    const int c = it / (TROWS * 8);
    const int rem = it - c * (TROWS * 8);
    const int r = rem >> 3, q = rem & 7;
    const int gy = y0 - HALO + r, gx = x0 + 4 * q;
    const bool ok = gy >= 0 && gy < H && gx < W;
    okmask |= (ok ? 1u : 0u) << i;
    goff_v[i] = ok ? gy * W + gx : 0;
    loff_v[i] = c * PLANE + r * RS + (HALO ? 4 : 0) + 4 * q;
    cv[i] = c;
  }
  if (HALO) {
    const int it = tid < NHI ? tid : NHI - 1;
    const int c = it / (TROWS * 2);
    const int rem = it - c * (TROWS * 2);
    const int r = rem >> 1, side = rem & 1;
    const int gy = y0 - HALO + r, gx = side ? x0 + 32 : x0 - 1;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    okmask |= (ok ? 1u : 0u) << 31;
    goff_h = ok ? gy * W + gx : 0;
    loff_h = c * PLANE + r * RS + (side ? 36 : 3);
    ch = c;
  }
  const float* wbase = p.w + (size_t)co_blk * Cin * TAPS * CO_T;
  const size_t HW = (size_t)H * W;
  const int nst = Cin / KH;

  auto plane_of = [&](int cg) -> const float* {
    return (cg < p.C1) ? p.src1 + (size_t)(b * p.C1 + cg) * HW : p.src2 + (size_t)(b * p.C2 + (cg - p.C1)) * HW;
  };
  auto xf1 = [&](float v, float sc, float sh, bool ok) -> float {
    v = v * sc + sh;
    const float sv = silu_f(v);
    v = act ? sv : v;
    return ok ? v : 0.f;
  };

  // task `it` of a stage: write item `it` of the staged registers (stage st_store) into LDS buffer NB, then refill the
  // same registers with item `it` of stage st_load.  Stage indices past the end are clamped by the caller (the extra
  // loads re-read the last stage, the extra stores land in a buffer nobody reads again), so there are no branches.
  // Every task is cut into NPART parts so that a part (<= ~12 instructions) fits in the shadow of the MFMAs around it.
  constexpr int NPART = 4;
  f32x4 tmp_o;   // transformed float4 of the vector item being staged (lives across the parts of one task)
  auto task_part = [&](auto itc, auto partc, auto nbc, int st_store, int st_load) {
    constexpr int IT = decltype(itc)::value, PART = decltype(partc)::value, NB = decltype(nbc)::value;
    float* din = NB ? s_in1 : s_in0;
    float* dw = NB ? s_w1 : s_w0;
    if constexpr (IT < NV) {
      const int cg = st_store * KH + cv[IT];
      const bool ok = (okmask >> IT) & 1u;
      if constexpr (PART == 0) { const float sc = s_sc[cg], sh = s_sh[cg]; tmp_o[0] = xf1(rv[IT][0], sc, sh, ok); tmp_o[1] = xf1(rv[IT][1], sc, sh, ok); }
      if constexpr (PART == 1) { const float sc = s_sc[cg], sh = s_sh[cg]; tmp_o[2] = xf1(rv[IT][2], sc, sh, ok); tmp_o[3] = xf1(rv[IT][3], sc, sh, ok); }
      if constexpr (PART == 2) *reinterpret_cast<f32x4*>(din + loff_v[IT]) = tmp_o;
      if constexpr (PART == 3) rv[IT] = *reinterpret_cast<const f32x4*>(plane_of(st_load * KH + cv[IT]) + goff_v[IT]);
    } else if constexpr (HALO && IT == NV) {
      const int cg = st_store * KH + ch;
      if constexpr (PART == 0) din[loff_h] = xf1(rh, s_sc[cg], s_sh[cg], (okmask >> 31) & 1u);
      if constexpr (PART == 2) rh = plane_of(st_load * KH + ch)[goff_h];
    } else {
      constexpr int WI = IT - NV - (HALO ? 1 : 0);
      int idx = tid + 256 * WI;
      idx = idx < W_ELEMS / 4 ? idx : W_ELEMS / 4 - 1;
      if constexpr (PART == 0) reinterpret_cast<f32x4*>(dw)[idx] = rw[WI];
      if constexpr (PART == 2) rw[WI] = reinterpret_cast<const f32x4*>(wbase + (size_t)st_load * KH * TAPS * CO_T)[idx];
    }
  };
  auto task = [&](auto itc, auto nbc, int st_store, int st_load) {   // whole task (prologue only)
    task_part(itc, std::integral_constant<int, 0>{}, nbc, st_store, st_load);
    task_part(itc, std::integral_constant<int, 1>{}, nbc, st_store, st_load);
    task_part(itc, std::integral_constant<int, 2>{}, nbc, st_store, st_load);
    task_part(itc, std::integral_constant<int, 3>{}, nbc, st_store, st_load);
  };

  // MFMAs of the stage in buffer CB; the FC*FP MFMAs of k-step s2 (1 <= s2 <= NTASK) are issued in NPART slices with
  // part q of staging task s2-1 after slice q, each slice fenced with sched_barrier
  auto stage = [&](auto cbc, int st_store, int st_load) {
    constexpr int CB = decltype(cbc)::value;
    using NBc = std::integral_constant<int, CB ^ 1>;
    const float* sw = (CB ? s_w1 : s_w0) + a_off;
    const float* si = (CB ? s_in1 : s_in0) + b_off;
    float a[2][FC], bb[2][FP];
    auto ld = [&](int s2, int slot) {
      const int cp = s2 / TAPS, tap = s2 % TAPS, dy = tap / KS, dx = tap % KS;
#pragma unroll
      for (int i = 0; i < FC; ++i) a[slot][i] = sw[i * 32 + (cp * 2 * TAPS + tap) * CO_T];
#pragma unroll
      for (int j = 0; j < FP; ++j) bb[slot][j] = si[j * RS + cp * 2 * PLANE + dy * RS + dx];
    };
    ld(0, 0);
    constexpr int NM = FC * FP, SL = (NM + NPART - 1) / NPART;   // MFMAs per k-step, per slice
    auto group = [&](auto sc2) {
      constexpr int S2 = decltype(sc2)::value;
      if (S2 + 1 < NS) ld(S2 + 1, (S2 + 1) & 1);
      auto slice = [&](auto qc) {
        constexpr int Q = decltype(qc)::value;
#pragma unroll
        for (int m = Q * SL; m < (Q + 1) * SL && m < NM; ++m) {
          const int i = m / FP, j = m % FP;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[S2 & 1][i], bb[S2 & 1][j], acc[i][j], 0, 0, 0);
        }
        if constexpr (S2 >= 1 && S2 - 1 < NTASK) task_part(std::integral_constant<int, S2 - 1>{}, qc, NBc{}, st_store, st_load);
        __builtin_amdgcn_sched_barrier(0);
      };
      slice(std::integral_constant<int, 0>{});
      slice(std::integral_constant<int, 1>{});
      slice(std::integral_constant<int, 2>{});
      slice(std::integral_constant<int, 3>{});
    };
    // compile-time unrolled k-steps
    auto run = [&](auto self, auto sc2) -> void {
      constexpr int S2 = decltype(sc2)::value;
      if constexpr (S2 < NS) { group(sc2); self(self, std::integral_constant<int, S2 + 1>{}); }
    };
    run(run, std::integral_constant<int, 0>{});
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  auto clampst = [&](int s) { return s < nst ? s : nst - 1; };

  // prologue: stage 0 -> registers -> LDS buffer 0, stage 1 -> registers
  auto fill_regs = [&](int st) {
#pragma unroll
    for (int i = 0; i < NV; ++i) rv[i] = *reinterpret_cast<const f32x4*>(plane_of(st * KH + cv[i]) + goff_v[i]);
    if (HALO) rh = plane_of(st * KH + ch)[goff_h];
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      int idx = tid + 256 * i;
      idx = idx < W_ELEMS / 4 ? idx : W_ELEMS / 4 - 1;
      rw[i] = reinterpret_cast<const f32x4*>(wbase + (size_t)st * KH * TAPS * CO_T)[idx];
    }
  };
  fill_regs(0);
  __syncthreads();   // s_sc / s_sh visible
  {
    auto all_tasks = [&](auto self, auto itc) -> void {
      constexpr int IT = decltype(itc)::value;
      if constexpr (IT < NTASK) { task(itc, B0{}, 0, clampst(1)); self(self, std::integral_constant<int, IT + 1>{}); }
    };
    all_tasks(all_tasks, std::integral_constant<int, 0>{});   // store stage 0 into buffer 0, load stage 1
  }
  __syncthreads();

  int s = 0;
  for (; s + 1 < nst; s += 2) {
    stage(B0{}, clampst(s + 1), clampst(s + 2));
    __syncthreads();
    stage(B1{}, clampst(s + 2), clampst(s + 3));
    __syncthreads();
  }
  if (s < nst) stage(B0{}, clampst(s + 1), clampst(s + 2));

  conv_epilogue<T, FC, FP, WC>(p, acc, b, co_blk, tx, ty, tiles_x, wc, wp, l31, kh);
}

}  // namespace sgmse

// Host-side engine: weight store (reference state_dict names), activation arena, the NCSN++ forward as a static
// kernel sequence, and the predictor-corrector / probability-flow sampler loop (hipGraph-captured per step).
//
// Mirrors, by behaviour, reference sgmse/backbones/ncsnpp.py:50-253 (module list) and :256-419 (forward),
// ncsnpp_48k.py (ordering deltas), layerspp.py:62-91,212-274 (AttnBlockpp, ResnetBlockBigGANpp),
// sampling/__init__.py:52-68 (pc_sampler) with predictors.py:56-65 and correctors.py:60-81, and sdes.py:72-89,
// 130-135,188-229 (OUVE discretisation).  Nothing here is numerics except index bookkeeping: all arithmetic is in
// the kernels.
#pragma once
#include <map>
#include <string>
#include <vector>
#include <cstdio>
#include <stdexcept>

#include <deque>
#include <sgmse_devrt.h>
#include "kernels_conv.h"
#include "conv_launch.h"
#include "kernels_norm_fir.h"
#include "kernels_attn_misc.h"
#include "kernels_stft.h"

namespace sgmse {

struct EngineError : std::runtime_error { using std::runtime_error::runtime_error; };

#define SG_CHECK(expr) do { int _e = (expr); if (_e != 0) { char _b[256]; snprintf(_b, sizeof _b, "%s failed: %s (%d)", #expr, drt::error_string(_e), _e); throw EngineError(_b); } } while (0)
#define SG_REQUIRE(cond, msg) do { if (!(cond)) throw EngineError(std::string(msg)); } while (0)

struct NetCfg {
  int variant = 0;            // 0: ncsnpp, 1: ncsnpp_48k
  int nf = 128;
  int n_levels = 7;
  int ch_mult[8] = {1, 1, 2, 2, 2, 2, 2, 0};
  int num_res_blocks = 2;
  int n_attn = 1;
  int attn_res[8] = {16, 0, 0, 0, 0, 0, 0, 0};
  int image_size = 256;
  int progressive = 1;        // 0 none, 1 output_skip
  int progressive_input = 1;  // 0 none, 1 input_skip
  int scale_by_sigma = 1;
};

// ---- module layout (order and names of reference all_modules, ncsnpp.py:107-253) ---------------------------
struct Mod {
  enum Kind { FOURIER, LINEAR, CONV3, RES, ATTN, COMBINE, GN } kind;
  int idx, cin, cout;
  bool up, down;
};

inline bool cfg_has_attn(const NetCfg& c, int res) {
  for (int i = 0; i < c.n_attn; ++i) if (c.attn_res[i] == res) return true;
  return false;
}

inline std::vector<Mod> build_layout(const NetCfg& c) {
  std::vector<Mod> mods;
  auto add = [&](Mod::Kind k, int cin, int cout, bool up = false, bool down = false) {
    mods.push_back(Mod{k, (int)mods.size(), cin, cout, up, down});
  };
  const int nf = c.nf, L = c.n_levels, temb = 4 * nf, channels = 4;
  add(Mod::FOURIER, 0, nf);
  add(Mod::LINEAR, 2 * nf, temb);
  add(Mod::LINEAR, temb, temb);
  add(Mod::CONV3, channels, nf);
  std::vector<int> hs_c{nf};
  int in_ch = nf;
  for (int l = 0; l < L; ++l) {
    const int res = c.image_size >> l;
    for (int r = 0; r < c.num_res_blocks; ++r) {
      const int out_ch = nf * c.ch_mult[l];
      add(Mod::RES, in_ch, out_ch);
      in_ch = out_ch;
      if (cfg_has_attn(c, res)) add(Mod::ATTN, in_ch, in_ch);
      hs_c.push_back(in_ch);
    }
    if (l != L - 1) {
      add(Mod::RES, in_ch, in_ch, false, true);
      if (c.progressive_input == 1) add(Mod::COMBINE, channels, in_ch);
      hs_c.push_back(in_ch);
    }
  }
  in_ch = hs_c.back();
  add(Mod::RES, in_ch, in_ch);
  add(Mod::ATTN, in_ch, in_ch);
  add(Mod::RES, in_ch, in_ch);
  for (int l = L - 1; l >= 0; --l) {
    const int res = c.image_size >> l;
    for (int r = 0; r < c.num_res_blocks + 1; ++r) {
      const int out_ch = nf * c.ch_mult[l];
      const int cin = in_ch + hs_c.back();
      hs_c.pop_back();
      add(Mod::RES, cin, out_ch);
      in_ch = out_ch;
    }
    if (cfg_has_attn(c, res)) add(Mod::ATTN, in_ch, in_ch);
    if (c.progressive == 1) { add(Mod::GN, in_ch, in_ch); add(Mod::CONV3, in_ch, channels); }
    if (l != 0) add(Mod::RES, in_ch, in_ch, true, false);
  }
  SG_REQUIRE(hs_c.empty(), "layout: skip stack not empty");
  if (c.progressive != 1) { add(Mod::GN, in_ch, in_ch); add(Mod::CONV3, in_ch, channels); }
  return mods;
}

// expected parameter names and element counts, in named_parameters() order
inline std::vector<std::pair<std::string, size_t>> param_manifest(const NetCfg& c) {
  std::vector<std::pair<std::string, size_t>> out;
  out.push_back({"output_layer.weight", 8});
  out.push_back({"output_layer.bias", 2});
  const int temb = 4 * c.nf;
  for (const Mod& m : build_layout(c)) {
    const std::string p = "all_modules." + std::to_string(m.idx) + ".";
    auto put = [&](const char* k, size_t n) { out.push_back({p + k, n}); };
    switch (m.kind) {
      case Mod::FOURIER: put("W", c.nf); break;
      case Mod::LINEAR: put("weight", (size_t)m.cout * m.cin); put("bias", m.cout); break;
      case Mod::CONV3: put("weight", (size_t)m.cout * m.cin * 9); put("bias", m.cout); break;
      case Mod::GN: put("weight", m.cin); put("bias", m.cin); break;
      case Mod::COMBINE: put("Conv_0.weight", (size_t)m.cout * m.cin); put("Conv_0.bias", m.cout); break;
      case Mod::ATTN:
        put("GroupNorm_0.weight", m.cin); put("GroupNorm_0.bias", m.cin);
        for (int i = 0; i < 4; ++i) {
          out.push_back({p + "NIN_" + std::to_string(i) + ".W", (size_t)m.cin * m.cin});
          out.push_back({p + "NIN_" + std::to_string(i) + ".b", (size_t)m.cin});
        }
        break;
      case Mod::RES:
        put("GroupNorm_0.weight", m.cin); put("GroupNorm_0.bias", m.cin);
        put("Conv_0.weight", (size_t)m.cout * m.cin * 9); put("Conv_0.bias", m.cout);
        put("Dense_0.weight", (size_t)m.cout * temb); put("Dense_0.bias", m.cout);
        put("GroupNorm_1.weight", m.cout); put("GroupNorm_1.bias", m.cout);
        put("Conv_1.weight", (size_t)m.cout * m.cout * 9); put("Conv_1.bias", m.cout);
        if (m.cin != m.cout || m.up || m.down) { put("Conv_2.weight", (size_t)m.cout * m.cin); put("Conv_2.bias", m.cout); }
        break;
    }
  }
  return out;
}

// ---- weight packing on the device --------------------------------------------------------------------------
// dst[blk][c][tap][j] = W(co = blk*co_t + j, c, tap); the logical weight is the channel-concatenation of up to three
// sources, each either OIHW [cout_s][cin][taps] (io = 0) or NIN's [cin][cout_s] (io = 1, taps = 1).
struct PackArgs { const float* src[3]; int nsrc, cout_per_src, io, cin, taps, cout, co_t; float* dst; size_t total; };

__global__ __launch_bounds__(256) void pack_weights_kernel(PackArgs p) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= p.total) return;
  const int j = (int)(e % p.co_t);
  size_t r = e / p.co_t;
  const int t = (int)(r % p.taps); r /= p.taps;
  const int c = (int)(r % p.cin);
  const int blk = (int)(r / p.cin);
  const int co = blk * p.co_t + j;
  float v = 0.f;
  if (co < p.cout) {
    const int s = co / p.cout_per_src, cj = co % p.cout_per_src;
    v = p.io ? p.src[s][(size_t)c * p.cout_per_src + cj] : p.src[s][((size_t)cj * p.cin + c) * p.taps + t];
  }
  p.dst[e] = v;
}

// OIHW weights with the input channels zero-padded from cin to cin_pad (entry convolution on the MFMA kernels)
__global__ __launch_bounds__(256) void pad_cin_kernel(const float* src, float* dst, int cout, int cin, int cin_pad, int taps) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)cout * cin_pad * taps) return;
  const int t = (int)(e % taps);
  const int c = (int)((e / taps) % cin_pad);
  const int co = (int)(e / ((size_t)taps * cin_pad));
  dst[e] = c < cin ? src[((size_t)co * cin + c) * taps + t] : 0.f;
}

__global__ __launch_bounds__(256) void zero_floats_kernel(float* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

__global__ __launch_bounds__(256) void fill_random_kernel(float* dst, size_t n, uint32_t seed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (uint32_t)i * 2654435761u ^ seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  dst[i] = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

// Measurement only (sgmse_calib_stream): a stream of KNOWN size for calibrating the FETCH_SIZE / WRITE_SIZE counters in the run that
// reads them (MI355X_MICROARCH.md, HBM: gfx950's FETCH_SIZE under-reports coalesced streams by a width-dependent factor).  VEC = floats
// per lane and access (2: the 8-byte column pairs of the Winograd kernel's staging, 4: its 16-byte weight fragments); consecutive lanes
// read consecutive elements, a workgroup walks a contiguous chunk.  WRITE = 1: the same stream stored instead of loaded.
template <int VEC, int WRITE>
__global__ __launch_bounds__(256) void calib_stream_kernel(float* buf, size_t nvec, float* sink) {
  struct alignas(VEC * 4) vec_t { float v[VEC]; };
  vec_t* p = reinterpret_cast<vec_t*>(buf);
  const size_t per = (nvec + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < nvec ? lo + per : nvec;
  float acc = 0.f;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    if constexpr (WRITE) { vec_t v; for (int k = 0; k < VEC; ++k) v.v[k] = (float)(i & 1023); p[i] = v; }
    else { const vec_t v = p[i]; for (int k = 0; k < VEC; ++k) acc += v.v[k]; }
  }
  if (!WRITE && acc == 123456.789f) sink[0] = acc;
}

// Test hook (SGMSE_POISON_LDS=1): fill the whole LDS of every CU with NaN bit patterns in front of every launch of an eager forward, so
// that a kernel reading LDS it never wrote -- whose result then depends on what the PREVIOUS workgroup on that CU left there, i.e. on what
// else runs on the device -- shows up as a changed result (round 6: tools/probes/concurrency_bits_probe.py).  One workgroup owns a CU.
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned* sink, unsigned salt) {
  __shared__ unsigned s[40960];                    // 160 KiB
  for (int i = threadIdx.x; i < 40960; i += 256) s[i] = 0x7FC00000u | ((unsigned)i + salt);
  __syncthreads();
  if (s[(threadIdx.x * 97 + salt) % 40960] == 0x12345678u) sink[0] = 1u;
}

// [cin][cout] -> [cout][cin]
__global__ __launch_bounds__(256) void transpose_io_kernel(const float* src, float* dst, int cin, int cout) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= cin * cout) return;
  const int co = e / cin, c = e % cin;
  dst[e] = src[(size_t)c * cout + co];
}

// ---- arena ---------------------------------------------------------------------------------------------------
// Deterministic first-fit pool over one device allocation.  In "measure" mode nothing is backed by memory and only
// the high-water mark is tracked (used to size the real arena before the first run / graph capture).
class Arena {
 public:
  void reset() { free_.clear(); free_.push_back({0, cap_}); live_.clear(); deferring_ = false; deferred_.clear(); }
  void configure(char* base, size_t cap) { base_ = base; cap_ = cap; live_.clear(); reset(); }
  // measure mode hands out addresses from an unmapped fake range; they are never dereferenced (dry run)
  void measure_mode() { base_ = reinterpret_cast<char*>(uintptr_t(1) << 44); cap_ = size_t(1) << 42; high_ = 0; reset(); live_.clear(); }
  size_t high_water() const { return high_; }
  float* alloc(size_t nfloats) {
    size_t bytes = (nfloats * 4 + 255) / 256 * 256;
    if (bytes == 0) bytes = 256;
    for (size_t i = 0; i < free_.size(); ++i) {
      if (free_[i].second >= bytes) {
        const size_t off = free_[i].first;
        free_[i].first += bytes; free_[i].second -= bytes;
        if (free_[i].second == 0) free_.erase(free_.begin() + i);
        live_[off] = bytes;
        if (off + bytes > high_) high_ = off + bytes;
        return reinterpret_cast<float*>(base_ + off);
      }
    }
    throw EngineError("activation arena exhausted");
  }
  // Deferred releases (side branches of the forward on a second stream, Engine::side_begin): a buffer released by the side branch, or
  // by the main chain while the side branch may still read it, goes back to the pool only at the join -- nothing enqueued in between
  // can be handed memory that an unfinished kernel of the other stream still uses.
  void defer_begin() { deferring_ = true; }
  void defer_end() { deferring_ = false; std::vector<float*> d; d.swap(deferred_); for (float* p : d) release(p); }
  void release(float* p) {
    if (!p) return;
    if (deferring_) { deferred_.push_back(p); return; }
    const size_t off = size_t(reinterpret_cast<char*>(p) - base_);
    auto it = live_.find(off);
    SG_REQUIRE(it != live_.end(), "arena: bad release");
    size_t bytes = it->second;
    live_.erase(it);
    size_t i = 0;
    while (i < free_.size() && free_[i].first < off) ++i;
    free_.insert(free_.begin() + i, {off, bytes});
    if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) {
      free_[i].second += free_[i + 1].second; free_.erase(free_.begin() + i + 1);
    }
    if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) {
      free_[i - 1].second += free_[i].second; free_.erase(free_.begin() + i);
    }
  }
 private:
  char* base_ = nullptr;
  size_t cap_ = 0, high_ = 0;
  std::vector<std::pair<size_t, size_t>> free_;
  std::map<size_t, size_t> live_;
  bool deferring_ = false;
  std::vector<float*> deferred_;
};

// st: [B][C][nsub][2] GroupNorm partial sums; amax: [B] upper bounds of |x| per utterance (null: unknown range)
// srows: image rows per statistics sub-tile (1: the fp32 kernels, 4: the split kernels; ConvArgs::stats_rows)
struct Tensor { float* p = nullptr; int C = 0, H = 0, W = 0; float* st = nullptr; int nsub = 0; float* amax = nullptr; int srows = 1; };

struct ConvW {            // one convolution's parameters on the device
  const float* oihw = nullptr;   // [cout][cin][ks][ks] (for NIN: transposed copy)
  const float* packed = nullptr; // MFMA layout (null if the shape is not MFMA-eligible)
  const float* packed32 = nullptr; // second MFMA layout with 32-channel output tiles: 4x more workgroups for launches that
                                   // would otherwise leave most CUs idle (the 16x32 ... 4x8 levels of the U-Net)
  const float* packed_thin = nullptr;  // [ci][tap][4 co] of the C -> 4 layers (kernels_conv_thin.h)
  const float* packed_split = nullptr; // split-kernel fragment layout (kernels_conv_split.h) in the engine's split mode,
                                       // 3x3 with cout % 128 == 0, cin % 16 == 0
  const float* split_scale = nullptr;  // fp16x2: per output channel the factor undoing its weights' power-of-two scale (table behind the fragments)
  int split_mode = 0;                  // 1 bf16x3, 2 fp16x2 (0: no split layout)
  const float* packed_wino = nullptr;  // Winograd F(2,3) x fp16x2 fragment layout (kernels_conv_wino.h), 3x3 with cout % 128 == 0, cin % 16 == 0
  const float* wino_scale = nullptr;   // ... and the per-output-channel factors behind it
  const float* bias = nullptr;
  int ks = 1, cin = 0, cout = 0, co_t = 0;
};

struct ResW { const float *g0w, *g0b, *g1w, *g1b; ConvW c0, c1, c2; bool has_c2; int temb_off; };
struct AttnW { const float *gw, *gb; ConvW qkv, proj; };

struct SamplerCfg {
  int N = 30;
  int corrector = 1;        // 0 none, 1 ald, 2 langevin
  float snr = 0.f;          // langevin only (ALD's snr is folded into the step table)
  int corrector_steps = 1;
  int predictor = 1;        // 0 none, 1 reverse_diffusion
  int probability_flow = 0;
  int denoise = 1;
  float theta = 1.5f;
  float std1 = 0.f;         // OUVE._std(T)
  const float* t = nullptr; const float* dt = nullptr; const float* ald_eps = nullptr; const float* ald_noise = nullptr;
  const float* G = nullptr; const float* G2 = nullptr;   // host arrays [N]
  // score wrapper per step (host arrays [N], all three or none): network input scale, score = alpha*x_t + beta*F
  const float* in_scale = nullptr; const float* score_alpha = nullptr; const float* score_beta = nullptr;
  int use_graph = 1;
};

class Engine {
 public:
  explicit Engine(int device, void* stream) : device_(device) {
    SG_CHECK(drt::set_device(device));
    stream_ = reinterpret_cast<drt::stream_t>(stream);
    read_knobs();
  }
  ~Engine() {
    for (void* p : owned_) drt::free_dev(p);
    for (void* p : wowned_) drt::free_dev(p);
    if (graph_valid_ || graph_stale_) drt::graph_destroy(&graph_);
    if (hstage_) drt::free_host(hstage_);
    if (hstage_ev_init_) drt::event_destroy(&hstage_ev_);
    for (drt::event_t& e : side_ev_) drt::event_destroy(&e);
    if (side_stream_ready_) drt::stream_destroy(side_stream_);
    for (ProfRec& r : prof_recs_) { drt::event_destroy(&r.a); drt::event_destroy(&r.b); }
  }

  drt::stream_t stream() const { return stream_; }
  void set_stream(void* s) { stream_ = reinterpret_cast<drt::stream_t>(s); }  // a captured graph can be launched on any stream

  // ---- weights ----------------------------------------------------------------------------------------------
  void set_config(const NetCfg& c) { cfg_ = c; layout_ = build_layout(c); weights_ready_ = false; invalidate_graph(); read_knobs(); }
  const NetCfg& config() const { return cfg_; }

  void load_weights(const char* const* names, const void* const* ptrs, const long long* numels, int n, int on_device) {
    invalidate_graph();            // a captured graph refers to the old weight buffers
    weights_ready_ = false;
    free_weight_allocs();
    auto manifest = param_manifest(cfg_);
    std::map<std::string, size_t> want;
    size_t total = 0;
    for (auto& kv : manifest) { want[kv.first] = kv.second; total += (kv.second + 63) / 64 * 64; }
    float* blob = static_cast<float*>(dev_alloc_w(total * 4));
    blob_ = blob; blob_elems_ = total;
    std::map<std::string, std::pair<const void*, long long>> given;
    for (int i = 0; i < n; ++i) given[names[i]] = {ptrs[i], numels[i]};
    size_t off = 0;
    W_.clear();
    for (auto& kv : manifest) {
      auto it = given.find(kv.first);
      if (it == given.end()) throw EngineError("load_weights: missing parameter " + kv.first);
      if ((size_t)it->second.second != kv.second)
        throw EngineError("load_weights: size mismatch for " + kv.first + " (got " + std::to_string(it->second.second) +
                          ", want " + std::to_string(kv.second) + ")");
      if (on_device) SG_CHECK(drt::memcpy_d2d(blob + off, it->second.first, kv.second * 4, stream_));
      else SG_CHECK(drt::memcpy_h2d(blob + off, it->second.first, kv.second * 4, stream_));
      W_[kv.first] = blob + off;
      off += (kv.second + 63) / 64 * 64;
    }
    if (!on_device) SG_CHECK(drt::stream_sync(stream_));  // host buffers may be released by the caller
    read_knobs();                  // (kernel-family knobs are per load, so one process can compare settings)
    finalize_weights();
  }

  size_t param_count() const { size_t n = 0; for (auto& kv : param_manifest(cfg_)) n += kv.second; return n; }

  // ---- network forward ----------------------------------------------------------------------------------------
  // x: complex64 [B][1][F][T] with batch stride xbs (complex elements), y likewise; t: device fp32.
  // out: complex64 [B][1][F][T] = sign * NCSNpp(cat[x,y], t).
  struct FwdCtl {
    const float* bias_table; int bias_bstride, bias_sstride; const int* step_ptr;
    const float* tvals; int t_bstride, t_sstride; float sign;
    const float* coef = nullptr; int coef_bstride = 0, coef_sstride = 0;   // score-wrapper rows {gamma, alpha, beta, -} or null
    const int* bias_step = nullptr;   // step counter the convolutions read for their bias row (null: row 0 / per-utterance rows)
    int y_plane = 0;     // ragged launches: y of utterance b lies y_plane * F * T_b elements behind its x-relative position (entry_kernel)
  };

  void forward_xy(const float2* xy, const float* t_dev, float2* out, int B, int F, int T) {
    // public single-evaluation entry: xy is [B][2][F][T]; t is a device array [B]
    require_ready();
    ensure_shape(B, F, T, /*nrows=*/B);
    compute_temb(t_dev, B);
    FwdCtl ctl{bias_table_, tot_temb_, 0, nullptr, t_dev, 1, 0, 1.0f};
    const long long FT = (long long)F * T;
    arena_.reset();
    if (ragged()) {          // xy: utterance after utterance, each [2][F][T_b]; out: each [F][T_b]
      ctl.y_plane = 1;
      run_forward(xy, 2, xy, 2, out, B, F, T, ctl);
    } else {
      run_forward(xy, 2 * FT, xy + FT, 2 * FT, out, B, F, T, ctl);
    }
  }

  // ---- sampler --------------------------------------------------------------------------------------------------
  void pc_sample(const float2* Y, float2* out, int B, int F, int T, const SamplerCfg& sc, const float2* noise,
                 unsigned long long seed) {
    require_ready();
    SG_REQUIRE(sc.N >= 1 && sc.t && sc.dt && sc.G && sc.G2, "pc_sample: step table missing");
    SG_REQUIRE(sc.corrector != 1 || (sc.ald_eps && sc.ald_noise), "pc_sample: ALD table missing");
    ensure_shape(B, F, T, sc.N);
    SG_REQUIRE(!ragged() || sc.corrector != 2, "ragged batches: the Langevin corrector couples the utterances of a batch (correctors.py:50-52) and is not supported");
    SG_REQUIRE(!ragged() || !noise, "ragged batches: replayed noise is not supported (in-kernel noise only)");
    const size_t n = ragged() ? rag_pix_[0] : (size_t)B * F * T;     // complex elements of Y / the result, utterance after utterance
    // step table -> device
    std::vector<float> tab((size_t)sc.N * SC_STRIDE, 0.f), tv(sc.N);
    for (int i = 0; i < sc.N; ++i) {
      float* r = &tab[(size_t)i * SC_STRIDE];
      r[SC_T] = sc.t[i]; r[SC_DT] = sc.dt[i]; r[SC_G] = sc.G[i]; r[SC_G2] = sc.G2[i];
      if (sc.corrector == 1) { r[SC_ALD_EPS] = sc.ald_eps[i]; r[SC_ALD_NOISE] = sc.ald_noise[i]; }
      tv[i] = sc.t[i];
    }
    const bool affine = sc.in_scale && sc.score_alpha && sc.score_beta;
    SG_REQUIRE(affine || (!sc.in_scale && !sc.score_alpha && !sc.score_beta), "pc_sample: give all three score-wrapper arrays or none");
    std::vector<float> cf;
    if (affine) {
      cf.assign((size_t)sc.N * 4, 0.f);
      for (int i = 0; i < sc.N; ++i) { cf[4 * i] = sc.in_scale[i]; cf[4 * i + 1] = sc.score_alpha[i]; cf[4 * i + 2] = sc.score_beta[i]; }
    }
    const unsigned long long* seed_dev = upload_tables(tab, tv, cf, seed, B);      // no host synchronisation (pinned staging)
    compute_temb(tsteps_, sc.N);

    const int ncorr = sc.corrector ? sc.corrector_steps : 0;
    const bool pred_noise = sc.predictor == 1 && !sc.probability_flow;
    const int draws_per_step = ncorr + (pred_noise ? 1 : 0);

    SG_CHECK(drt::memcpy_d2d(sy_, Y, n * 8, stream_));   // engine-owned copy: the captured graph never refers to caller memory
    Y = sy_;
    SamplerArgs sa{};
    sa.x = sx_; sa.x_mean = sxm_; sa.y = Y; sa.score = sscore_; sa.noise = noise; sa.seed = seed_dev;
    sa.table = step_table_; sa.step_ptr = step_ctr_; sa.theta = sc.theta; sa.std1 = sc.std1; sa.n = (int)n;
    sa.score_w = sc.probability_flow ? 0.5f : 1.0f;
    sa.snr = sc.snr; sa.B = B; sa.per = F * T; sa.partial = lang_partial_; sa.lang = lang_scal_;
    sa.rag_off = ragged() ? rag_off_dev_[0] : nullptr;
    const dim3 eg((unsigned)((n + 255) / 256));

    sa.draw_base = 0; sa.draw_per_step = 0;
    DRT_LAUNCH(sampler_prior_kernel, eg, dim3(256), stream_, sa);
    SG_CHECK(drt::memcpy_d2d(sxm_, sx_, n * 8, stream_));
    DRT_LAUNCH(step_set_kernel, dim3(1), dim3(64), stream_, step_ctr_, 0);

    FwdCtl ctl{bias_table_, 0, tot_temb_, step_ctr_, tsteps_, 0, 1, -1.0f};
    if (affine) { ctl.coef = coef_table_; ctl.coef_bstride = 0; ctl.coef_sstride = 1; }
    const long long FT = ragged() ? 1 : (long long)F * T;          // batch stride of x / y (ragged: multiplier of the utterance's offset)
    auto step_body = [&]() {
      for (int cs = 0; cs < ncorr; ++cs) {
        arena_.reset();
        run_forward(sx_, FT, Y, FT, sscore_, B, F, T, ctl);
        SamplerArgs a = sa; a.draw_base = 1 + cs; a.draw_per_step = draws_per_step;
        if (sc.corrector == 2) {
          DRT_LAUNCH(sampler_langevin_norms_kernel, dim3(LANG_NBLK, B), dim3(256), stream_, a);
          DRT_LAUNCH(sampler_langevin_scalars_kernel, dim3(1), dim3(64), stream_, a);
          DRT_LAUNCH(sampler_langevin_kernel, eg, dim3(256), stream_, a);
        } else {
          DRT_LAUNCH(sampler_ald_kernel, eg, dim3(256), stream_, a);
        }
      }
      if (sc.predictor == 1) {
        arena_.reset();
        run_forward(sx_, FT, Y, FT, sscore_, B, F, T, ctl);
        SamplerArgs a = sa; a.draw_base = 1 + ncorr; a.draw_per_step = draws_per_step; a.add_noise = pred_noise ? 1 : 0;
        DRT_LAUNCH(sampler_revdiff_kernel, eg, dim3(256), stream_, a);
      }
      // NonePredictor.update_fn returns (x, x) (predictors.py:69-76): xt_mean := xt also after a corrector moved them apart
      if (sc.predictor == 0 && ncorr > 0) SG_CHECK(drt::memcpy_d2d(sxm_, sx_, n * 8, stream_));
      DRT_LAUNCH(step_inc_kernel, dim3(1), dim3(64), stream_, step_ctr_);
    };

    GraphKey key{B, F, T, sc.corrector, ncorr, sc.predictor, sc.probability_flow + (affine ? 2 : 0), (const void*)Y, (const void*)noise,
                 sc.theta + 1000.f * sc.snr * (sc.corrector == 2), draws_per_step};      // (set_frames invalidates the graph itself)
    const bool want_graph = sc.use_graph && drt::graphs_supported();
    if (want_graph) {
      if (!graph_valid_ || !(key == graph_key_)) {
        invalidate_graph();
        capture_step(step_body);
        graph_key_ = key;
      }
      for (int i = 0; i < sc.N; ++i) SG_CHECK(drt::graph_launch(&graph_, stream_));
    } else {
      for (int i = 0; i < sc.N; ++i) step_body();
    }
    SG_CHECK(drt::memcpy_d2d(out, sc.denoise ? sxm_ : sx_, n * 8, stream_));
    nfe_ = sc.N * (ncorr + 1);   // the reference counts N*(corrector.n_steps+1) whatever the predictor (sampling/__init__.py:67)
  }
  // get_sb_sampler(...)() (sampling/__init__.py:145-249): N steps of  x <- w_prev x + w_est model(x, y, t) + w_y y + w_z z,
  // x_0 = y.  Host arrays [N]; coef (in_scale / alpha / beta, [N] each or all null) = the ScoreModel.forward wrapper.
  void sb_sample(const float2* Y, float2* out, int B, int F, int T, int N, const float* t, const float* w_prev, const float* w_est,
                 const float* w_y, const float* w_z, const float* in_scale, const float* alpha, const float* beta, int stochastic,
                 const float2* noise, unsigned long long seed, int use_graph) {
    require_ready();
    SG_REQUIRE(N >= 1 && t && w_prev && w_est && w_y && w_z, "sb_sample: step table missing");
    SG_REQUIRE(!ragged(), "ragged batches: not supported by the Schroedinger-bridge sampler");
    ensure_shape(B, F, T, N);
    const size_t n = (size_t)B * F * T;
    std::vector<float> tab((size_t)N * SC_STRIDE, 0.f), tv(N);
    for (int i = 0; i < N; ++i) {
      float* r = &tab[(size_t)i * SC_STRIDE];
      r[SC_T] = t[i]; r[SB_WPREV] = w_prev[i]; r[SB_WEST] = w_est[i]; r[SB_WY] = w_y[i]; r[SB_WZ] = w_z[i];
      tv[i] = t[i];
    }
    const bool affine = in_scale && alpha && beta;
    SG_REQUIRE(affine || (!in_scale && !alpha && !beta), "sb_sample: give all three score-wrapper arrays or none");
    std::vector<float> cf;
    if (affine) {
      cf.assign((size_t)N * 4, 0.f);
      for (int i = 0; i < N; ++i) { cf[4 * i] = in_scale[i]; cf[4 * i + 1] = alpha[i]; cf[4 * i + 2] = beta[i]; }
    }
    const unsigned long long* seed_dev = upload_tables(tab, tv, cf, seed, B);
    compute_temb(tsteps_, N);
    SG_CHECK(drt::memcpy_d2d(sy_, Y, n * 8, stream_));
    SG_CHECK(drt::memcpy_d2d(sx_, Y, n * 8, stream_));          // x_0 = y
    DRT_LAUNCH(step_set_kernel, dim3(1), dim3(64), stream_, step_ctr_, 0);

    SamplerArgs sa{};
    sa.x = sx_; sa.x_mean = sxm_; sa.y = sy_; sa.score = sscore_; sa.noise = noise; sa.seed = seed_dev;
    sa.table = step_table_; sa.step_ptr = step_ctr_; sa.n = (int)n; sa.add_noise = stochastic ? 1 : 0; sa.B = B; sa.per = F * T;
    sa.draw_base = 0; sa.draw_per_step = 1;
    const dim3 eg((unsigned)((n + 255) / 256));
    FwdCtl ctl{bias_table_, 0, tot_temb_, step_ctr_, tsteps_, 0, 1, -1.0f};
    if (affine) { ctl.coef = coef_table_; ctl.coef_bstride = 0; ctl.coef_sstride = 1; }
    const long long FT = (long long)F * T;
    auto step_body = [&]() {
      arena_.reset();
      run_forward(sx_, FT, sy_, FT, sscore_, B, F, T, ctl);
      DRT_LAUNCH(sampler_sb_kernel, eg, dim3(256), stream_, sa);
      DRT_LAUNCH(step_inc_kernel, dim3(1), dim3(64), stream_, step_ctr_);
    };
    GraphKey key{B, F, T, 100 + stochastic, 0, 0, affine ? 2 : 0, nullptr, (const void*)noise, 0.f, 1};
    if (use_graph && drt::graphs_supported()) {
      if (!graph_valid_ || !(key == graph_key_)) {
        invalidate_graph();
        capture_step(step_body);
        graph_key_ = key;
      }
      for (int i = 0; i < N; ++i) SG_CHECK(drt::graph_launch(&graph_, stream_));
    } else {
      for (int i = 0; i < N; ++i) step_body();
    }
    SG_CHECK(drt::memcpy_d2d(out, sx_, n * 8, stream_));
    nfe_ = N;
  }
  void set_noise_streams(const unsigned long long* ids, int n) { streams_next_.assign(ids, ids + n); }
  void set_ragged_frames(const int* frames, int n) { set_frames(frames, n); }     // sgmse_set_frames
  int last_nfe() const { return nfe_; }
  int graph_captures() const { return graph_captures_; }     // how many times a step was captured + instantiated (tests, bench)
  int graph_updates() const { return graph_updates_; }       // ... captured and applied to the existing executable in place
  int split_mode() const { return split_mode_; }
  bool winograd() const { return split_mode_ == 2 && wino_; }
  size_t arena_bytes() const { return arena_cap_; }

  // ---- single ops (op-level C ABI + tests) ------------------------------------------------------------------
  // per-utterance max |x| of one or two NCHW tensors -> [B] (+ [B]) floats, for the op-level entry points
  float* input_bounds(const float* x1, int C1, const float* x2, int C2, int B, int HW) {
    const size_t per = (size_t)B * kAmaxSpread;
    float* out = static_cast<float*>(dev_alloc_tmp(2 * per * 4));
    SG_CHECK(drt::memset_dev(out, 0, 2 * per * 4, stream_));
    for (int b = 0; b < B; ++b) {
      const size_t n1 = (size_t)C1 * HW;
      DRT_LAUNCH(absmax_kernel, dim3((unsigned)std::min<size_t>((n1 + 255) / 256, 1024)), dim3(256), stream_, x1 + b * n1, n1,
                 out + (size_t)b * kAmaxSpread);
      if (x2) {
        const size_t n2 = (size_t)C2 * HW;
        DRT_LAUNCH(absmax_kernel, dim3((unsigned)std::min<size_t>((n2 + 255) / 256, 1024)), dim3(256), stream_, x2 + b * n2, n2,
                   out + per + (size_t)b * kAmaxSpread);
      }
    }
    return out;
  }

  // the bound gn_finalize_kernel leaves for an fp16x2 3x3 consumer, for a producer given as explicit coefficients (or none)
  float* producer_bound(const float* in_scale, const float* in_shift, int C, const float* amax1, const float* amax2, int B) {
    float* out = static_cast<float*>(dev_alloc_tmp((size_t)B * kAmaxSpread * 4));
    DRT_LAUNCH(xform_bound_kernel, dim3(B), dim3(64), stream_, in_scale, in_shift, C, amax1, amax2, out);
    return out;
  }

  void op_conv2d(const float* x, const float* w_oihw, const float* bias, const float* res, float* out, int B, int Cin,
                 int Cout, int H, int W, int ks, float out_scale, int force_direct, const float* in_scale,
                 const float* in_shift, int in_act, const float* x2, int C2) {
    ConvArgs a{};
    a.src1 = x; a.src2 = x2; a.C1 = Cin - C2; a.C2 = C2; a.bias = bias; a.res = res; a.out_scale = out_scale; a.out = out;
    a.Cout = Cout; a.B = B; a.H = H; a.W = W; a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = in_act;
    ConvPlan pl = choose_conv_plan(ks, Cin, Cout, H, W);
    if (force_direct == 6) {                               // exact-fp32 VALU kernel of the C -> 4 pyramid convolutions
      SG_REQUIRE(conv_thin_eligible(ks, a.C1, C2, Cout), "op_conv2d: shape is not eligible for the thin-output kernel");
      const float* pk = pack_thin(w_oihw, Cin, Cout, false);
      a.w = pk;
      launch_conv_thin(a, stream_);
      SG_CHECK(drt::stream_sync(stream_));
      free_tmp(const_cast<float*>(pk));
    } else if (force_direct == 7) {                        // 2-D Winograd F(2x2,3x3) x fp16x2 (kernels_conv_wino2d.h)
      SG_REQUIRE(ks == 3 && conv_wino_eligible(a.C1, C2, Cout, W), "op_conv2d: shape is not eligible for the 2-D Winograd kernel");
      const float* pk = pack_wino2d(w_oihw, Cin, Cout, false, &a.co_scale);
      a.w = pk;
      float* bounds = input_bounds(x, a.C1, x2, C2, B, H * W);
      const float* am2 = x2 ? bounds + (size_t)B * kAmaxSpread : nullptr;
      float* xb = producer_bound(in_scale, in_shift, Cin, bounds, am2, B);
      a.xbound = xb;
      launch_conv_wino2d(a, stream_);
      SG_CHECK(drt::stream_sync(stream_));
      free_tmp(const_cast<float*>(pk)); free_tmp(bounds); free_tmp(xb);
    } else if (force_direct == 4 || force_direct == 5) {          // Winograd F(2,3) x fp16x2: 4 = 8-row shape, 5 = 4-row shape
      SG_REQUIRE(ks == 3 && conv_wino_eligible(a.C1, C2, Cout, W), "op_conv2d: shape is not eligible for the Winograd kernel");
      const float* pk = pack_wino(w_oihw, Cin, Cout, false, &a.co_scale);
      a.w = pk;
      float* bounds = input_bounds(x, a.C1, x2, C2, B, H * W);
      const float* am2 = x2 ? bounds + (size_t)B * kAmaxSpread : nullptr;
      float* xb = producer_bound(in_scale, in_shift, Cin, bounds, am2, B);
      a.xbound = xb;
      launch_conv_wino(a, stream_, force_direct == 5);
      SG_CHECK(drt::stream_sync(stream_));
      free_tmp(const_cast<float*>(pk)); free_tmp(bounds); free_tmp(xb);
    } else if (force_direct == 2 || force_direct == 3) {          // the split kernels: 2 bf16x3, 3 fp16x2
      SG_REQUIRE(conv_split_eligible(ks, a.C1, C2, Cout) || conv_thin_split_eligible(ks, a.C1, C2, Cout),
                 "op_conv2d: shape is not eligible for the split kernels");
      const int smode = force_direct - 1;
      const float* pk = pack_split(w_oihw, ks, Cin, Cout, smode, false, &a.co_scale);
      a.w = pk;
      float *bounds = nullptr, *xb = nullptr;
      if (smode == 2) {      // dynamic input scale: range bounds as the producers would have left them
        SG_REQUIRE(ks == 3 || in_scale == nullptr, "op_conv2d: the fp16x2 1x1 kernel takes raw inputs (no fused producer)");
        bounds = input_bounds(x, a.C1, x2, C2, B, H * W);
        const float* am2 = x2 ? bounds + (size_t)B * kAmaxSpread : nullptr;
        if (ks == 1) { a.amax1 = bounds; a.amax2 = am2; }
        else { xb = producer_bound(in_scale, in_shift, Cin, bounds, am2, B); a.xbound = xb; }
      }
      launch_conv_split(a, ks, smode, stream_);
      SG_CHECK(drt::stream_sync(stream_));
      free_tmp(const_cast<float*>(pk));
      if (bounds) free_tmp(bounds);
      if (xb) free_tmp(xb);
    } else if (pl.mfma && !force_direct && (C2 == 0 || a.C1 % ((ks == 3) ? 8 : 32) == 0)) {
      const size_t ne = packed_weight_elems(ks, Cin, Cout, pl.co_t);
      float* pk = static_cast<float*>(dev_alloc_tmp(ne * 4));
      PackArgs pa{}; pa.src[0] = w_oihw; pa.nsrc = 1; pa.cout_per_src = Cout; pa.io = 0; pa.cin = Cin; pa.taps = ks * ks;
      pa.cout = Cout; pa.co_t = pl.co_t; pa.dst = pk; pa.total = ne;
      DRT_LAUNCH(pack_weights_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), stream_, pa);
      a.w = pk;
      launch_conv_mfma(a, ks, pl, stream_);
      SG_CHECK(drt::stream_sync(stream_));
      free_tmp(pk);
    } else {
      a.w = w_oihw;
      launch_conv_direct(a, ks, stream_);
    }
    check_launch();
  }

  void op_groupnorm(const float* x, const float* gamma, const float* beta, float* out, int B, int C, int H, int W, int act,
                    const float* x2, int C2) {
    const int HW = H * W, C1 = C - C2;
    float* stats = static_cast<float*>(dev_alloc_tmp((size_t)B * C * 2 * 4));
    float* sc = static_cast<float*>(dev_alloc_tmp((size_t)B * C * 4));
    float* sh = static_cast<float*>(dev_alloc_tmp((size_t)B * C * 4));
    float* stats2 = stats + (size_t)B * C1 * 2;
    DRT_LAUNCH(gn_chan_stats_kernel, dim3(B * C1), dim3(256), stream_, x, (const float*)nullptr, C1, 0, HW, stats, Rag{nullptr, nullptr, nullptr}, 0);
    if (C2) DRT_LAUNCH(gn_chan_stats_kernel, dim3(B * C2), dim3(256), stream_, x2, (const float*)nullptr, C2, 0, HW, stats2, Rag{nullptr, nullptr, nullptr}, 0);
    const int G = std::min(C / 4, 32);
    const GnFin f{stats, stats2, nullptr, nullptr, gamma, beta, sc, sh, nullptr, Rag{nullptr, nullptr, nullptr}, C1, 1, 1, C2, 1, 1, G, HW, 0, 1e-6f};
    DRT_LAUNCH(gn_finalize_kernel, dim3(G, B), dim3(256), stream_, f);
    DRT_LAUNCH(gn_apply_kernel, dim3((HW + 1023) / 1024, B * C), dim3(256), stream_, x, x2, C1, C2, HW, (const float*)sc,
               (const float*)sh, act, out);
    SG_CHECK(drt::stream_sync(stream_));
    check_launch();
    free_tmp(stats); free_tmp(sc); free_tmp(sh);
  }

  void op_fir(const float* x, float* out, int BC, int H, int W, int up, const float* in_scale, const float* in_shift, int in_act,
              float* out_raw) {
    FirArgs fa{x, out, in_scale, in_shift, in_act, BC, H, W, out_raw, 1, Rag{nullptr, nullptr, nullptr}};
    launch_fir(fa, up);
    check_launch();
  }

  // dtype: 0 float, 1 double, 2 half (the element types of the reference op, op/upfirdn2d_kernel.cu:311)
  void op_upfirdn2d(int dtype, const void* x, const void* kern, void* out, int BC, int H, int W, int kh, int kw, int up_x, int up_y,
                    int down_x, int down_y, int px0, int px1, int py0, int py1) {
    UpfirdnArgs a{x, kern, out, BC, H, W, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1, 0, 0};
    a.Ho = (H * up_y + py0 + py1 - kh) / down_y + 1;
    a.Wo = (W * up_x + px0 + px1 - kw) / down_x + 1;
    SG_REQUIRE(a.Ho > 0 && a.Wo > 0, "upfirdn2d: empty output");
    const dim3 grid((a.Ho * a.Wo + 255) / 256, BC);
    if (dtype == 0) DRT_LAUNCH(upfirdn2d_generic_kernel<UpfirdnF32>, grid, dim3(256), stream_, a);
    else if (dtype == 1) DRT_LAUNCH(upfirdn2d_generic_kernel<UpfirdnF64>, grid, dim3(256), stream_, a);
    else DRT_LAUNCH(upfirdn2d_generic_kernel<UpfirdnF16>, grid, dim3(256), stream_, a);
    check_launch();
  }

  void op_attention(const float* qkv, float* out, int B, int C, int S) {
    AttnArgs a{qkv, out, B, C, S, 1.0f / sqrtf((float)C), Rag{nullptr, nullptr, nullptr}, 0};
    SG_REQUIRE(launch_attn_core(a, stream_), "attention: C must be 32, 64, 128 or 256");
    check_launch();
  }

  void op_stft(const float* sig, const float* window, float2* spec, int B, int L, int n_fft, int hop) {
    SG_REQUIRE(n_fft <= kMaxNfft && n_fft >= 2 && hop >= 1 && L > n_fft / 2, "stft: unsupported geometry");
    const float2* tw = twiddle(n_fft);
    StftArgs a{sig, window, tw, spec, L, n_fft, hop, n_fft / 2 + 1, (L + 2 * (n_fft / 2) - n_fft) / hop + 1};
    DRT_LAUNCH(stft_kernel, dim3(a.K, B), dim3(256), stream_, a);
    check_launch();
  }

  void op_istft(const float2* spec, const float* window, float* out, int B, int K, int n_fft, int hop, int length) {
    SG_REQUIRE(n_fft <= kMaxNfft && n_fft >= 2 && hop >= 1, "istft: unsupported geometry");
    SG_REQUIRE(K >= 1 && length >= 1 && length + n_fft / 2 <= n_fft + hop * (K - 1), "istft: length exceeds the overlap-add span");
    const float2* tw = twiddle(n_fft);
    IstftArgs a{spec, window, tw, out, n_fft, hop, n_fft / 2 + 1, K, length};
    DRT_LAUNCH(istft_kernel, dim3((length + 255) / 256, B), dim3(256), stream_, a);
    check_launch();
  }

  void op_spec_xform(const float2* in, float2* out, size_t n, int type, float factor, float exponent, int inverse) {
    SpecXformArgs a{in, out, n, type, factor, exponent, inverse};
    DRT_LAUNCH(spec_xform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), stream_, a);
    check_launch();
  }

  void sync() { SG_CHECK(drt::stream_sync(stream_)); }

  // Kernel micro-benchmark of one MFMA convolution shape on random operands: `iters` back-to-back launches between two
  // events on this stream.  variant = SGMSE_CONV_VARIANT encoding (-1: process default).  Returns ms per launch.
  float bench_conv(int ks, int B, int Cin, int Cout, int H, int W, int variant, int iters, int fused) {
    ConvPlan pl = choose_conv_plan(ks, Cin, Cout, H, W);
    SG_REQUIRE(pl.mfma, "bench_conv: shape is not MFMA-eligible");
    if (variant >= 0 && (variant & 512)) { pl.rows = 4; variant &= ~512; }   // measurement knob: 128 co x 128 px tile
    int ablate = 0, abl_split = 0;
    bool split_rows4 = false;
    if (variant >= 0) {
      split_rows4 = (variant >> 23) & 1;         // measurement knob: 4-row workgroup shape of the split 3x3 kernel
      abl_split = (variant >> 12) & 1023;        // measurement knob: compile-time variant of the split 3x3 kernel, bits 12..21
      if (!(variant & (64 | 128))) { ablate = abl_split & 15; abl_split = 0; }   // fp32 kernels: run-time ablation bits 12..15
      variant &= 4095;
    }
    const bool wino = variant >= 0 && (variant & 1024);          // measurement: the Winograd F(2,3) x fp16x2 kernel (bit 23: its 4-row shape)
    const bool wino2d = variant >= 0 && (variant & 2048) && !wino;   // ... the 2-D F(2x2,3x3) x fp16x2 kernel (ablation bits 12..: 8 / 16)
    const int smode = variant < 0 ? 0 : ((variant & 128) ? 2 : ((variant & 64) ? 1 : 0));
    const bool b3 = smode != 0 && !wino && !wino2d;
    SG_REQUIRE(!(wino || wino2d) || (ks == 3 && conv_wino_eligible(Cin, 0, Cout, W)), "bench_conv: shape is not eligible for the Winograd kernel");
    SG_REQUIRE(!b3 || conv_split_eligible(ks, Cin, 0, Cout), "bench_conv: shape is not eligible for the split kernels");
    const size_t nx = (size_t)B * Cin * H * W, no = (size_t)B * Cout * H * W, nw = (size_t)Cout * Cin * ks * ks;
    const size_t ne = packed_weight_elems(ks, Cin, Cout, pl.co_t);
    float* x = static_cast<float*>(dev_alloc_tmp(nx * 4));
    float* o = static_cast<float*>(dev_alloc_tmp(no * 4));
    float* r = static_cast<float*>(dev_alloc_tmp(no * 4));
    float* w = static_cast<float*>(dev_alloc_tmp(nw * 4));
    float* pk = static_cast<float*>(dev_alloc_tmp(ne * 4));
    float* sc = static_cast<float*>(dev_alloc_tmp((size_t)B * Cin * 8));
    auto fill = [&](float* d, size_t n, uint32_t seed) {
      DRT_LAUNCH(fill_random_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), stream_, d, n, seed);
    };
    fill(x, nx, 1); fill(r, no, 2); fill(w, nw, 3); fill(sc, (size_t)B * Cin * 2, 4);
    PackArgs pa{}; pa.src[0] = w; pa.nsrc = 1; pa.cout_per_src = Cout; pa.io = 0; pa.cin = Cin; pa.taps = ks * ks; pa.cout = Cout;
    pa.co_t = pl.co_t; pa.dst = pk; pa.total = ne;
    DRT_LAUNCH(pack_weights_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), stream_, pa);
    ConvArgs a{};
    a.src1 = x; a.C1 = Cin; a.w = pk; a.out = o; a.Cout = Cout; a.B = B; a.H = H; a.W = W; a.out_scale = 1.f;
    a.ablate = ablate;
    if (fused) { a.in_scale = sc; a.in_shift = sc + (size_t)B * Cin; a.in_act = 1; a.res = r; a.bias = w; a.out_scale = 0.70710678f; }
    drt::event_t e0{}, e1{};
    drt::event_create(&e0); drt::event_create(&e1);
    const float* pk3 = nullptr;
    float *bounds = nullptr, *xbound = nullptr;
    if (b3) {
      pk3 = pack_split(w, ks, Cin, Cout, smode, false, &a.co_scale); a.w = pk3;
      if (smode == 2) {
        bounds = input_bounds(x, Cin, nullptr, 0, B, H * W);
        if (ks == 1) {
          a.in_scale = nullptr; a.in_shift = nullptr; a.in_act = 0;     // raw input, as in the network's shortcut layers
          a.amax1 = bounds;
        } else {
          xbound = producer_bound(a.in_scale, a.in_shift, Cin, bounds, nullptr, B); a.xbound = xbound;
        }
      }
    }
    if (wino || wino2d) {
      pk3 = wino2d ? pack_wino2d(w, Cin, Cout, false, &a.co_scale) : pack_wino(w, Cin, Cout, false, &a.co_scale); a.w = pk3;
      bounds = input_bounds(x, Cin, nullptr, 0, B, H * W);
      xbound = producer_bound(a.in_scale, a.in_shift, Cin, bounds, nullptr, B); a.xbound = xbound;
    }
    unsigned long long* trace_dev = nullptr;
    const size_t n_wg = (size_t)B * ((H + 7) / 8) * ((W + 31) / 32) * ((Cout + 127) / 128);
    if (abl_split & 64) {
      trace_dev = static_cast<unsigned long long*>(dev_alloc_tmp(n_wg * 256));      // (16 words per workgroup; the Winograd kernel's trace: 32)
      SG_CHECK(drt::memset_dev(trace_dev, 0, n_wg * 256, stream_));
      a.trace = trace_dev;
    }
    auto go = [&]() {
      if (wino2d) launch_conv_wino2d(a, stream_, abl_split & 63);
      else if (wino) launch_conv_wino(a, stream_, split_rows4, (abl_split & 64) != 0, abl_split & 63);
      else if (b3) launch_conv_split(a, ks, smode, stream_, split_rows4, abl_split);
      else launch_conv_mfma(a, ks, pl, stream_, variant);
    };
    for (int i = 0; i < 2; ++i) go();
    drt::event_record(&e0, stream_);
    for (int i = 0; i < iters; ++i) go();
    drt::event_record(&e1, stream_);
    drt::event_sync(&e1);
    const float ms = drt::event_elapsed_ms(e0, e1) / (float)iters;
    check_launch();
    drt::event_destroy(&e0); drt::event_destroy(&e1);
    if (trace_dev) {                    // phase time stamps of the LAST launch -> $SGMSE_TRACE_OUT (binary: 16 x u64 per workgroup)
      const size_t words = wino ? 32 : 16;
      std::vector<unsigned long long> h(n_wg * words);
      SG_CHECK(drt::memcpy_d2h(h.data(), trace_dev, n_wg * words * 8, stream_));
      SG_CHECK(drt::stream_sync(stream_));
      if (const char* path = getenv("SGMSE_TRACE_OUT")) { if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); } }
      free_tmp(trace_dev);
    }
    for (float* q : {x, o, r, w, pk, sc}) free_tmp(q);
    if (pk3) free_tmp(const_cast<float*>(pk3));
    if (bounds) free_tmp(bounds);
    if (xbound) free_tmp(xbound);
    return ms;
  }

  // Measurement only: one launch of a known-size stream (calib_stream_kernel) for the counter calibration of tools/gpu_visit.sh
  // STEPS=pmcbench.  mode 0 = read, 1 = write; bytes_per_lane 8 or 16.  Returns ms of the launch.
  float calib_stream(int mode, int bytes_per_lane, size_t total_bytes) {
    SG_REQUIRE((bytes_per_lane == 8 || bytes_per_lane == 16) && (mode == 0 || mode == 1) && total_bytes >= 4096, "calib_stream: bad arguments");
    float* buf = static_cast<float*>(dev_alloc_tmp(total_bytes + 256));
    SG_CHECK(drt::memset_dev(buf, 0, total_bytes + 256, stream_));
    const size_t nvec = total_bytes / (size_t)bytes_per_lane;
    drt::event_t e0{}, e1{};
    drt::event_create(&e0); drt::event_create(&e1);
    drt::event_record(&e0, stream_);
    const dim3 g(256 * 16), b(256);
    float* sink = buf + total_bytes / 4;
    if (bytes_per_lane == 8) { if (mode) DRT_LAUNCH((calib_stream_kernel<2, 1>), g, b, stream_, buf, nvec, sink); else DRT_LAUNCH((calib_stream_kernel<2, 0>), g, b, stream_, buf, nvec, sink); }
    else { if (mode) DRT_LAUNCH((calib_stream_kernel<4, 1>), g, b, stream_, buf, nvec, sink); else DRT_LAUNCH((calib_stream_kernel<4, 0>), g, b, stream_, buf, nvec, sink); }
    drt::event_record(&e1, stream_);
    drt::event_sync(&e1);
    const float ms = drt::event_elapsed_ms(e0, e1);
    check_launch();
    drt::event_destroy(&e0); drt::event_destroy(&e1);
    free_tmp(buf);
    return ms;
  }

  // per-kernel-class timing of one forward (eager, events on this stream); fills ms per class
  enum { TC_CONV3_BIG = 0, TC_CONV3, TC_CONV1, TC_DIRECT, TC_GN, TC_FIR, TC_ATTN, TC_MISC, TC_COUNT };
  // work[] = algorithmic FLOPs for the conv/attention classes, algorithmic bytes (each operand read once, each result
  // written once) for the HBM-bound classes (groupnorm statistics, FIR, misc)
  void profile_forward(const float2* xy, const float* t_dev, float2* out, int B, int F, int T, float* ms_out, double* work_out,
                       int* launches_out) {
    prof_ = true;
    for (int i = 0; i < TC_COUNT; ++i) { prof_ms_[i] = 0.f; prof_flops_[i] = 0.0; prof_n_[i] = 0; }
    prof_used_ = 0;
    forward_xy(xy, t_dev, out, B, F, T);
    prof_ = false;
    // the event pairs are read after the whole forward has been queued: launches then follow each other as in the replayed graph
    // (with a wait after every launch the device idled between them and each interval carried ~40 us of dispatch: the class times
    // were 4-5 % above the kernel durations rocprofv3 reports for the captured step)
    SG_CHECK(drt::stream_sync(stream_));
    for (size_t i = 0; i < prof_used_; ++i) {
      ProfRec& r = prof_recs_[i];
      if (r.cls < 0) continue;                      // (a tock() without its tick())
      const float ms = drt::event_elapsed_ms(r.a, r.b);
      prof_ms_[r.cls] += ms;
      prof_flops_[r.cls] += r.work;
      prof_n_[r.cls] += r.launches;
      if (prof_dump_ && r.note[0]) fprintf(stderr, "[sgmse-prof] %s %.4f ms %.1f Gwork/s\n", r.note, ms, r.work / ms * 1e-6);
    }
    for (int i = 0; i < TC_COUNT; ++i) { ms_out[i] = prof_ms_[i]; work_out[i] = prof_flops_[i]; launches_out[i] = prof_n_[i]; }
  }

 private:
  // ---- memory helpers ------------------------------------------------------------------------------------------
  // allocations that belong to the current weight set; released together when new weights are loaded (EMA swap, reload)
  void* dev_alloc_w(size_t bytes) {
    void* p = nullptr;
    SG_CHECK(drt::malloc_dev(&p, bytes ? bytes : 256));
    if (poison_) SG_CHECK(drt::memset_dev(p, 0xFF, bytes ? bytes : 256, stream_));
    wowned_.push_back(p);
    return p;
  }
  void free_weight_allocs() {
    if (wowned_.empty()) return;
    drt::stream_sync(stream_);
    for (void* p : wowned_) drt::free_dev(p);
    wowned_.clear();
  }
  // SGMSE_POISON=1 (tests): every allocation, and the activation arena before every forward, is filled with 0xFF bytes (a NaN
  // in every float), so that a kernel reading memory nobody wrote shows up as a NaN in the result instead of depending on
  // what the allocation happened to hold
  void* dev_alloc(size_t bytes) {
    void* p = nullptr;
    SG_CHECK(drt::malloc_dev(&p, bytes ? bytes : 256));
    if (poison_) SG_CHECK(drt::memset_dev(p, 0xFF, bytes ? bytes : 256, stream_));
    owned_.push_back(p);
    return p;
  }
  void* dev_alloc_tmp(size_t bytes) {
    void* p = nullptr;
    SG_CHECK(drt::malloc_dev(&p, bytes ? bytes : 256));
    if (poison_) SG_CHECK(drt::memset_dev(p, 0xFF, bytes ? bytes : 256, stream_));
    return p;
  }
  void free_tmp(void* p) { drt::free_dev(p); }
  void dev_free_owned(void* p) {
    for (size_t i = 0; i < owned_.size(); ++i) if (owned_[i] == p) { owned_.erase(owned_.begin() + i); break; }
    drt::free_dev(p);
  }
  void check_launch() { SG_CHECK(drt::last_error()); }
  void require_ready() { SG_REQUIRE(weights_ready_, "weights not loaded (call sgmse_load_weights first)"); }
  // A captured step that no longer matches (new shape, new ragged composition, new weights) is not destroyed but kept as the
  // target of an in-place update by the next capture (capture_step): same launch sequence, other arguments.
  void invalidate_graph() { if (graph_valid_) { graph_valid_ = false; graph_stale_ = true; } }
  void drop_graph() { if (graph_valid_ || graph_stale_) { drt::graph_destroy(&graph_); graph_valid_ = graph_stale_ = false; } }
  template <class Body>
  void capture_step(Body&& body) {
    SG_CHECK(drt::stream_sync(stream_));
    SG_CHECK(drt::graph_begin_capture(stream_));
    capturing_ = true;
    try { body(); } catch (...) {
      // leave the stream usable: end the capture, throw its graph away, and forget the executable a stale update would have reused
      capturing_ = false;
      drt::graph_abort_capture(stream_);
      drop_graph();
      throw;
    }
    capturing_ = false;
    if (graph_stale_) {
      const int r = drt::graph_end_capture_update(stream_, &graph_);
      if (r > 0) { drt::graph_destroy(&graph_); graph_stale_ = false; graph_valid_ = false; SG_CHECK(r); }     // (nothing left to leak or to reuse)
      if (r == 0) ++graph_updates_; else ++graph_captures_;
    } else {
      const int r = drt::graph_end_capture(stream_, &graph_);
      if (r != 0) { drt::graph_destroy(&graph_); graph_valid_ = false; SG_CHECK(r); }
      ++graph_captures_;
    }
    graph_stale_ = false;
    graph_valid_ = true;
  }
  bool graph_stale_ = false, capturing_ = false; int graph_updates_ = 0;

  const float2* twiddle(int n_fft) {
    auto it = twiddles_.find(n_fft);
    if (it != twiddles_.end()) return it->second;
    std::vector<float2> h(n_fft);
    for (int m = 0; m < n_fft; ++m) {
      const double ang = 2.0 * 3.14159265358979323846 * (double)m / (double)n_fft;
      h[m] = make_float2((float)cos(ang), (float)sin(ang));
    }
    float2* d = static_cast<float2*>(dev_alloc(sizeof(float2) * n_fft));
    SG_CHECK(drt::memcpy_h2d(d, h.data(), sizeof(float2) * n_fft, stream_));
    SG_CHECK(drt::stream_sync(stream_));
    twiddles_[n_fft] = d;
    return d;
  }

  const float* Wp(const std::string& name) const {
    auto it = W_.find(name);
    if (it == W_.end()) throw EngineError("internal: unknown parameter " + name);
    return it->second;
  }

  ConvW make_conv(const std::string& wname, const std::string& bname, int ks, int cin, int cout) {
    return make_conv(Wp(wname), bname.empty() ? nullptr : Wp(bname), ks, cin, cout);
  }
  ConvW make_conv(const float* oihw, const float* bias, int ks, int cin, int cout) {
    ConvW c; c.ks = ks; c.cin = cin; c.cout = cout;
    c.oihw = oihw; c.bias = bias;
    ConvPlan pl = choose_conv_plan(ks, cin, cout, 8, 32);
    if (pl.mfma) {
      c.co_t = pl.co_t;
      const size_t ne = packed_weight_elems(ks, cin, cout, pl.co_t);
      float* pk = static_cast<float*>(dev_alloc_w(ne * 4));
      PackArgs pa{}; pa.src[0] = c.oihw; pa.nsrc = 1; pa.cout_per_src = cout; pa.io = 0; pa.cin = cin; pa.taps = ks * ks;
      pa.cout = cout; pa.co_t = pl.co_t; pa.dst = pk; pa.total = ne;
      DRT_LAUNCH(pack_weights_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), stream_, pa);
      c.packed = pk;
      if (pl.co_t > 32) {
        const size_t ne32 = packed_weight_elems(ks, cin, cout, 32);
        float* pk32 = static_cast<float*>(dev_alloc_w(ne32 * 4));
        PackArgs pb = pa; pb.co_t = 32; pb.dst = pk32; pb.total = ne32;
        DRT_LAUNCH(pack_weights_kernel, dim3((unsigned)((ne32 + 255) / 256)), dim3(256), stream_, pb);
        c.packed32 = pk32;
      }
    }
    // fp16x2: the input scale is a per-utterance power of two derived at run time -- 3x3 layers from the bound of their GroupNorm
    // producer's output (gn_finalize_kernel), 1x1 layers (raw residual stream) from the producers' range bounds -- the stored
    // factor only undoes the weights'
    if (split_mode_ && (conv_split_eligible(ks, cin, 0, cout) || conv_thin_split_eligible(ks, cin, 0, cout))) {
      c.split_mode = split_mode_;
      c.packed_split = pack_split(c.oihw, ks, cin, cout, c.split_mode, true, &c.split_scale);
    }
    if (conv_thin_eligible(ks, cin, 0, cout)) c.packed_thin = pack_thin(c.oihw, cin, cout, true);
    // the wide levels run these layers on the Winograd F(2,3) x fp16x2 kernel (conv(): use_wino)
    if (split_mode_ == 2 && wino_ && ks == 3 && conv_wino_eligible(cin, 0, cout, 2)) c.packed_wino = pack_wino(c.oihw, cin, cout, true, &c.wino_scale);
    return c;
  }

  // weights in the fragment order of conv3x3_split_kernel (mode 1: bf16x3, 2: fp16x2 with the layer's power-of-two scale)
  const float* pack_split(const float* oihw, int ks, int cin, int cout, int mode, bool weight_owned, const float** scale_out) {
    const int taps = ks * ks;
    const size_t frags = mode == 2 ? packed_split_frags<SplitH2>(cin, cout, taps) : packed_split_frags<SplitB3>(cin, cout, taps);
    const int cout_pad = (cout + 127) / 128 * 128;
    const size_t bytes = frags * 16 + (size_t)cout_pad * 8;             // [fragments][per-channel inverse scales][scales (packing scratch)]
    uint32_t* pk = static_cast<uint32_t*>(weight_owned ? dev_alloc_w(bytes) : dev_alloc_tmp(bytes));
    PackSplitArgs pa{oihw, pk, cin, cout, frags, nullptr, taps};
    const dim3 grid((unsigned)((frags + 255) / 256));
    if (mode == 2) {
      float* inv = reinterpret_cast<float*>(pk) + frags * 4;
      float* sc = inv + cout_pad;
      DRT_LAUNCH(split_co_scale_kernel, dim3((unsigned)((cout_pad + 255) / 256)), dim3(256), stream_, oihw, cin * taps, cout, cout_pad, inv, sc);
      pa.co_scale = sc;
      DRT_LAUNCH(pack_weights_split_kernel<SplitH2>, grid, dim3(256), stream_, pa);
      *scale_out = inv;
    } else {
      DRT_LAUNCH(pack_weights_split_kernel<SplitB3>, grid, dim3(256), stream_, pa);
      *scale_out = nullptr;
    }
    return reinterpret_cast<const float*>(pk);
  }

  // weights in the fragment order of conv3x3_wino_kernel: transformed along the kernel's columns in fp64, scaled per output channel
  const float* pack_thin(const float* oihw, int cin, int cout, bool weight_owned) {
    const size_t ne = packed_thin_elems(cin);
    float* pk = static_cast<float*>(weight_owned ? dev_alloc_w(ne * 4) : dev_alloc_tmp(ne * 4));
    DRT_LAUNCH(pack_weights_thin_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), stream_, oihw, pk, cin, cout);
    return pk;
  }
  const float* pack_wino(const float* oihw, int cin, int cout, bool weight_owned, const float** scale_out) {
    const size_t frags = packed_wino_frags(cin, cout);
    const int cout_pad = (cout + 127) / 128 * 128;
    const size_t bytes = packed_wino_bytes(cin, cout) + (size_t)cout_pad * 4;       // [fragments][inverse scales][scales (packing scratch)]
    uint32_t* pk = static_cast<uint32_t*>(weight_owned ? dev_alloc_w(bytes) : dev_alloc_tmp(bytes));
    float* inv = reinterpret_cast<float*>(pk) + frags * 4;
    float* sc = inv + cout_pad;
    DRT_LAUNCH(wino_co_scale_kernel, dim3((unsigned)((cout_pad + 255) / 256)), dim3(256), stream_, oihw, cin, cout, cout_pad, inv, sc);
    PackWinoArgs pa{oihw, pk, cin, cout, frags};
    DRT_LAUNCH(pack_weights_wino_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), stream_, pa, (const float*)sc);
    *scale_out = inv;
    return reinterpret_cast<const float*>(pk);
  }

  const float* pack_wino2d(const float* oihw, int cin, int cout, bool weight_owned, const float** scale_out) {
    const size_t frags = packed_wino2d_frags(cin, cout);
    const int cout_pad = (cout + 127) / 128 * 128;
    const size_t bytes = packed_wino2d_bytes(cin, cout) + (size_t)cout_pad * 4;     // [fragments][inverse scales][scales (packing scratch)]
    uint32_t* pk = static_cast<uint32_t*>(weight_owned ? dev_alloc_w(bytes) : dev_alloc_tmp(bytes));
    float* inv = reinterpret_cast<float*>(pk) + frags * 4;
    float* sc = inv + cout_pad;
    DRT_LAUNCH(wino2d_co_scale_kernel, dim3((unsigned)((cout_pad + 255) / 256)), dim3(256), stream_, oihw, cin, cout, cout_pad, inv, sc);
    PackWinoArgs pa{oihw, pk, cin, cout, frags};
    DRT_LAUNCH(pack_weights_wino2d_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), stream_, pa, (const float*)sc);
    *scale_out = inv;
    return reinterpret_cast<const float*>(pk);
  }

  // NIN weights W[cin][cout] (layers.py:549); nsrc of them concatenated along cout (fused q|k|v projection)
  ConvW make_nin(const std::string& pre, const int* which, int nsrc, int C) {
    ConvW c; c.ks = 1; c.cin = C; c.cout = nsrc * C;
    float* tr = static_cast<float*>(dev_alloc_w((size_t)nsrc * C * C * 4));
    float* bb = static_cast<float*>(dev_alloc_w((size_t)nsrc * C * 4));
    for (int s = 0; s < nsrc; ++s) {
      const std::string n = pre + "NIN_" + std::to_string(which[s]);
      DRT_LAUNCH(transpose_io_kernel, dim3((C * C + 255) / 256), dim3(256), stream_, Wp(n + ".W"), tr + (size_t)s * C * C, C, C);
      SG_CHECK(drt::memcpy_d2d(bb + (size_t)s * C, Wp(n + ".b"), (size_t)C * 4, stream_));
    }
    c.oihw = tr; c.bias = bb;
    ConvPlan pl = choose_conv_plan(1, C, c.cout, 8, 32);
    if (pl.mfma) {
      c.co_t = pl.co_t;
      const size_t ne = packed_weight_elems(1, C, c.cout, pl.co_t);
      float* pk = static_cast<float*>(dev_alloc_w(ne * 4));
      PackArgs pa{}; pa.nsrc = nsrc; pa.cout_per_src = C; pa.io = 1; pa.cin = C; pa.taps = 1; pa.cout = c.cout; pa.co_t = pl.co_t;
      pa.dst = pk; pa.total = ne;
      for (int s = 0; s < nsrc; ++s) pa.src[s] = Wp(pre + "NIN_" + std::to_string(which[s]) + ".W");
      DRT_LAUNCH(pack_weights_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), stream_, pa);
      c.packed = pk;
      // (round 5) the 32-channel-tile layout too, as make_conv does: the attention projections sit at the 16 x 32 and 4 x 8 levels, where
      // conv() runs the fp32 layers on 32-channel tiles with chunked accumulation and, at small batches, split-K -- without this layout
      // q|k|v and the output projection stayed on 128-channel tiles: 24 / 8 workgroups running 16 serial K-stages at batch 1
      // (31 / 30 us per launch against 18 us for the 1x1 shortcut of the same size, profiles/r05_prof_dump_b1.txt)
      // (round 6: the compile-time switch of round 5's A/B is gone -- one build variant, the tested one.  Attention outputs changed in
      //  their last bits with that round: fixtures of these levels recorded before it are not bit-comparable.)
      if (pl.co_t > 32) {
        const size_t ne32 = packed_weight_elems(1, C, c.cout, 32);
        float* pk32 = static_cast<float*>(dev_alloc_w(ne32 * 4));
        PackArgs pb = pa; pb.co_t = 32; pb.dst = pk32; pb.total = ne32;
        DRT_LAUNCH(pack_weights_kernel, dim3((unsigned)((ne32 + 255) / 256)), dim3(256), stream_, pb);
        c.packed32 = pk32;
      }
    }
    if (split_mode_ && conv_split_eligible(1, C, 0, c.cout)) {
      c.split_mode = 1;
      c.packed_split = pack_split(c.oihw, 1, C, c.cout, 1, true, &c.split_scale);
    }
    return c;
  }

  void finalize_weights() {
    res_.clear(); attn_.clear(); conv_.clear(); gn_.clear();
    std::vector<DenseDesc> descs;
    int off = 0;
    for (const Mod& m : layout_) {
      const std::string p = "all_modules." + std::to_string(m.idx) + ".";
      if (m.kind == Mod::RES) {
        ResW r{};
        r.g0w = Wp(p + "GroupNorm_0.weight"); r.g0b = Wp(p + "GroupNorm_0.bias");
        r.g1w = Wp(p + "GroupNorm_1.weight"); r.g1b = Wp(p + "GroupNorm_1.bias");
        r.c0 = make_conv(p + "Conv_0.weight", p + "Conv_0.bias", 3, m.cin, m.cout);
        r.c1 = make_conv(p + "Conv_1.weight", p + "Conv_1.bias", 3, m.cout, m.cout);
        r.has_c2 = (m.cin != m.cout) || m.up || m.down;
        if (r.has_c2) r.c2 = make_conv(p + "Conv_2.weight", p + "Conv_2.bias", 1, m.cin, m.cout);
        r.temb_off = off;
        descs.push_back(DenseDesc{Wp(p + "Dense_0.weight"), Wp(p + "Dense_0.bias"), r.c0.bias, m.cout, off});
        off += m.cout;
        res_[m.idx] = r;
      } else if (m.kind == Mod::ATTN) {
        AttnW a{};
        a.gw = Wp(p + "GroupNorm_0.weight"); a.gb = Wp(p + "GroupNorm_0.bias");
        const int qkv[3] = {0, 1, 2}, pr[1] = {3};
        a.qkv = make_nin(p, qkv, 3, m.cin);
        a.proj = make_nin(p, pr, 1, m.cin);
        attn_[m.idx] = a;
      } else if (m.kind == Mod::CONV3) {
        conv_[m.idx] = make_conv(p + "weight", p + "bias", 3, m.cin, m.cout);
        if (m.cin == 4 && entry_mfma_) {
          // Entry convolution (4 -> nf on the full-resolution image, ncsnpp.py:286): as a direct VALU kernel it was bound by
          // its scalar weight loads (1.6 ms at batch 32 for a 0.36 ms output write, plus a 0.4 ms statistics pass).  With the
          // input channels zero-padded to one 8-channel K-stage it runs on the fp32 MFMA kernel, GroupNorm partials fused
          // (1.07 ms; 16 channels on the bf16x3 split kernel measured the same at batch 32 and twice the time at batch 1)
          constexpr int cp = 8;
          float* w8 = static_cast<float*>(dev_alloc_w((size_t)m.cout * cp * 9 * 4));
          DRT_LAUNCH(pad_cin_kernel, dim3((unsigned)(((size_t)m.cout * cp * 9 + 255) / 256)), dim3(256), stream_, Wp(p + "weight"), w8, m.cout, 4, cp, 9);
          entry8_ = make_conv(w8, Wp(p + "bias"), 3, cp, m.cout);
          entry8_idx_ = m.idx;
        }
      } else if (m.kind == Mod::COMBINE) {
        conv_[m.idx] = make_conv(p + "Conv_0.weight", p + "Conv_0.bias", 1, m.cin, m.cout);
      } else if (m.kind == Mod::GN) {
        gn_[m.idx] = {Wp(p + "weight"), Wp(p + "bias")};
      }
    }
    tot_temb_ = off;
    n_dense_ = (int)descs.size();
    dense_descs_ = static_cast<DenseDesc*>(dev_alloc_w(sizeof(DenseDesc) * descs.size()));
    SG_CHECK(drt::memcpy_h2d(dense_descs_, descs.data(), sizeof(DenseDesc) * descs.size(), stream_));
    SG_CHECK(drt::stream_sync(stream_));
    check_launch();
    weights_ready_ = true;
    shape_B_ = 0;
    temb_rows_ = 0;      // the time-embedding tables are sized by nf / the Dense layout of THIS net
  }

  // ---- shape-dependent buffers --------------------------------------------------------------------------------
  void ensure_shape(int B, int F, int T, int nrows) {
    SG_REQUIRE(B >= 1 && F >= 1 && T >= 1, "bad shape");
    const int down = 1 << (cfg_.n_levels - 1);
    SG_REQUIRE(F % down == 0 && T % down == 0, "F and T must be multiples of 2^(levels-1) (pad_spec pads T to a multiple of 64)");
    if (B != shape_B_ || F != shape_F_ || T != shape_T_ || (ragged() && !rag_built_)) {
      invalidate_graph();
      cur_F_ = F;
      if (ragged()) build_rag_tables(B, F, T);
      // size the arena by a dry run
      arena_.measure_mode();
      dry_ = true;
      {
        FwdCtl ctl{nullptr, 0, 0, nullptr, nullptr, 0, 0, 1.f};
        run_forward(nullptr, 0, nullptr, 0, nullptr, B, F, T, ctl);
      }
      dry_ = false;
      const size_t need = arena_.high_water() + (1 << 20);
      if (need > arena_cap_) {
        if (arena_base_) dev_free_owned(arena_base_);
        arena_base_ = static_cast<char*>(dev_alloc(need));
        arena_cap_ = need;
      }
      arena_.configure(arena_base_, arena_cap_);
      const size_t n = (size_t)B * F * T;      // (a ragged batch needs at most this)
      if (n > samp_n_) {
        for (float2** q : {&sx_, &sxm_, &sscore_, &sy_}) { if (*q) dev_free_owned(*q); *q = static_cast<float2*>(dev_alloc(n * 8)); }
        samp_n_ = n;
      }
      // range-bound slots: one per convolution output of the forward just dry-run
      const size_t need_amax = (size_t)amax_next_ * B * kAmaxSpread;
      amax_slots_ = amax_next_;
      if (need_amax > amax_pool_floats_) {
        if (amax_pool_) dev_free_owned(amax_pool_);
        amax_pool_ = static_cast<float*>(dev_alloc(need_amax * 4));
        amax_pool_floats_ = need_amax;
      }
      if (!step_ctr_) step_ctr_ = static_cast<int*>(dev_alloc(256));
      if (!lang_scal_) lang_scal_ = static_cast<float*>(dev_alloc(256));
      if (lang_partial_) dev_free_owned(lang_partial_);
      lang_partial_ = static_cast<float*>(dev_alloc((size_t)B * LANG_NBLK * 2 * 4));
      shape_B_ = B; shape_F_ = F; shape_T_ = T;
    }
    if (nrows > temb_rows_) {
      invalidate_graph();
      if (temb_act_) { dev_free_owned(temb_act_); dev_free_owned(bias_table_); dev_free_owned(step_table_); dev_free_owned(tsteps_); dev_free_owned(coef_table_); dev_free_owned(bias_cur_); }
      temb_rows_ = std::max(nrows, 64);
      temb_act_ = static_cast<float*>(dev_alloc((size_t)temb_rows_ * 4 * cfg_.nf * 4));
      bias_table_ = static_cast<float*>(dev_alloc((size_t)temb_rows_ * std::max(tot_temb_, 1) * 4));
      bias_cur_ = static_cast<float*>(dev_alloc((size_t)std::max(tot_temb_, 1) * 4));
      step_table_ = static_cast<float*>(dev_alloc((size_t)temb_rows_ * SC_STRIDE * 4));
      tsteps_ = static_cast<float*>(dev_alloc((size_t)temb_rows_ * 4));
      coef_table_ = static_cast<float*>(dev_alloc((size_t)temb_rows_ * 16));
    }
  }

  void compute_temb(const float* t_dev, int rows) {
    TembArgs ta{t_dev, rows, Wp("all_modules.0.W"), Wp("all_modules.1.weight"), Wp("all_modules.1.bias"),
                Wp("all_modules.2.weight"), Wp("all_modules.2.bias"), cfg_.nf, temb_act_};
    SG_REQUIRE(cfg_.nf <= 256, "nf > 256 not supported by temb kernel");
    DRT_LAUNCH(temb_mlp_kernel, dim3(rows), dim3(256), stream_, ta);
    DRT_LAUNCH(temb_dense_kernel, dim3(n_dense_, rows), dim3(256), stream_, (const DenseDesc*)dense_descs_,
               (const float*)temb_act_, 4 * cfg_.nf, bias_table_, tot_temb_);
    check_launch();
  }

  // ---- op wrappers used by the forward ------------------------------------------------------------------------
  // fused producer of a consumer's input.  `bound`: per-utterance upper bound of |producer output| (gn_coeffs), from which the
  // fp16x2 3x3 kernel derives its input scale.  A tensor that IS the FIR-resampled output of a producer applied upstream
  // (scale == nullptr) inherits that producer's bound: the [1,3,3,1] resamplers are convex combinations.
  struct Xform { const float* scale = nullptr; const float* shift = nullptr; int act = 0; const float* bound = nullptr; };

  void tick(int cls, double work, int launches = 1) {
    if (debug_sync_ && !dry_ && !capturing_) {      // SGMSE_DEBUG_SYNC=1: wait for every launch and name it (a device fault then names its kernel)
      const int e = drt::stream_sync(stream_);
      fprintf(stderr, "[sgmse-dbg] %s -> %s\n", prof_note_[0] ? prof_note_ : "(launch)", e ? drt::error_string(e) : "ok");
      fflush(stderr);
      if (!prof_) prof_note_[0] = 0;
    }
    if (!prof_ || prof_used_ == 0) return;
    ProfRec& r = prof_recs_[prof_used_ - 1];
    drt::event_record(&r.b, stream_);
    r.cls = cls; r.work = work; r.launches = launches;
    snprintf(r.note, sizeof r.note, "%s", prof_note_);
    prof_note_[0] = 0;
  }
  bool noting() const { return (prof_ && prof_dump_) || debug_sync_; }
  // one event pair per launch of a profiled forward (created on first use, reused by later profiles, destroyed with the engine)
  struct ProfRec { drt::event_t a{}, b{}; int cls = -1; double work = 0.0; int launches = 0; char note[160] = {0}; };
  void tock() {
    if (lds_poison_ && !dry_ && !capturing_ && !drt::is_emulator()) {
      const int idx = lds_poison_count_++;
      if (lds_poison_at_ < 0 || idx == lds_poison_at_ || (lds_poison_upto_ && idx <= lds_poison_at_)) {
        if (!lds_sink_) lds_sink_ = static_cast<unsigned*>(dev_alloc(256));
        DRT_LAUNCH(lds_poison_kernel, dim3(768), dim3(256), stream_, lds_sink_, (unsigned)idx);
      }
    }
    if (!prof_) return;
    if (prof_used_ == prof_recs_.size()) {
      prof_recs_.emplace_back();
      drt::event_create(&prof_recs_.back().a); drt::event_create(&prof_recs_.back().b);
    }
    ProfRec& r = prof_recs_[prof_used_++];
    r.cls = -1; r.note[0] = 0;
    drt::event_record(&r.a, stream_);
  }

  Tensor new_tensor(int C, int H, int W) { Tensor t; t.C = C; t.H = H; t.W = W; t.p = arena_.alloc((size_t)C * pix_total(H, W)); return t; }

  // ---- ragged batches (set_frames): utterances of different frame counts in one launch ------------------------------------
  // Every activation tensor holds utterance b's planes with its own row stride, packed one utterance after the other; the
  // per-level tables below give each kernel an utterance's width and first pixel (Rag, kernels_conv.h).  Which kernel FAMILY a
  // layer runs on is decided from the level alone (dec_W: the width a 512-frame utterance has there), never from the actual
  // widths: an utterance then goes through the same kernels, with the same arithmetic, alone, in a uniform batch or in a
  // ragged one.
  static constexpr int kNominalFrames = 512;
  bool ragged() const { return !rag_T_.empty(); }
  int level_of(int H) const { int l = 0; while ((cur_F_ >> l) > H && l < 30) ++l; return l; }
  int dec_W(int H) const { return std::max(1, kNominalFrames >> level_of(H)); }
  size_t pix_total(int H, int W) const { return ragged() ? rag_pix_.at(level_of(H)) : (size_t)B_ * H * W; }
  Rag rag_of(int H) const {
    if (!ragged()) return Rag{nullptr, nullptr, nullptr};
    const int l = level_of(H);
    return Rag{rag_w_dev_.at(l), rag_off_dev_.at(l), rag_soff_dev_.at(l)};
  }
  bool rag_all_mult2(int H) const {
    if (!ragged()) return true;
    const int l = level_of(H);
    for (int t : rag_T_) if ((t >> l) % 2) return false;
    return true;
  }
  bool rag_all_mult4(int H) const {
    if (!ragged()) return true;
    const int l = level_of(H);
    for (int t : rag_T_) if ((t >> l) % 4) return false;
    return true;
  }
  void set_frames(const int* frames, int n) {          // n == 0: uniform batches again
    std::vector<int> v(frames, frames + n);
    if (v != rag_T_) { rag_T_ = v; rag_built_ = false; shape_B_ = 0; invalidate_graph(); }
  }
  void build_rag_tables(int B, int F, int T) {
    const int L = cfg_.n_levels, down = 1 << (L - 1);
    SG_REQUIRE((int)rag_T_.size() == B, "ragged batch: sgmse_set_frames gave a different number of utterances");
    int tmax = 0;
    for (int t : rag_T_) { SG_REQUIRE(t >= down && t % down == 0, "ragged batch: frame counts must be multiples of 2^(levels-1)"); tmax = std::max(tmax, t); }
    SG_REQUIRE(tmax == T, "ragged batch: T must be the largest frame count");
    for (void* q : rag_owned_) dev_free_owned(q);
    rag_owned_.clear(); rag_w_dev_.clear(); rag_off_dev_.clear(); rag_soff_dev_.clear(); rag_pix_.clear(); rag_cols_dev_.clear(); rag_ncols_.clear();
    for (int l = 0; l < L; ++l) {
      const int H = F >> l;
      std::vector<int> w(B), cols(B + 1, 0);
      std::vector<long long> off(B + 1, 0), soff(B + 1, 0);
      for (int b = 0; b < B; ++b) {
        w[b] = rag_T_[b] >> l;
        off[b + 1] = off[b] + (long long)H * w[b];
        soff[b + 1] = soff[b] + (long long)H * ((w[b] + 31) / 32);
        cols[b + 1] = cols[b] + (w[b] + 31) / 32;             // tile columns that exist (ConvArgs::rag_cols)
      }
      int* wd = static_cast<int*>(dev_alloc((size_t)B * 4));
      long long* od = static_cast<long long*>(dev_alloc((size_t)(B + 1) * 8));
      long long* sd = static_cast<long long*>(dev_alloc((size_t)(B + 1) * 8));
      SG_CHECK(drt::memcpy_h2d(wd, w.data(), (size_t)B * 4, stream_));
      SG_CHECK(drt::memcpy_h2d(od, off.data(), (size_t)(B + 1) * 8, stream_));
      SG_CHECK(drt::memcpy_h2d(sd, soff.data(), (size_t)(B + 1) * 8, stream_));
      int* cd = static_cast<int*>(dev_alloc((size_t)(B + 1) * 4));
      SG_CHECK(drt::memcpy_h2d(cd, cols.data(), (size_t)(B + 1) * 4, stream_));
      SG_CHECK(drt::stream_sync(stream_));
      rag_owned_.push_back(cd); rag_cols_dev_.push_back(cd); rag_ncols_.push_back(cols[B]);
      rag_owned_.push_back(wd); rag_owned_.push_back(od); rag_owned_.push_back(sd);
      rag_w_dev_.push_back(wd); rag_off_dev_.push_back(od); rag_soff_dev_.push_back(sd);
      rag_pix_.push_back((size_t)off[B]);
    }
    rag_built_ = true;
  }
  std::vector<int> rag_T_;
  std::vector<int*> rag_w_dev_; std::vector<long long*> rag_off_dev_, rag_soff_dev_; std::vector<size_t> rag_pix_;
  std::vector<int*> rag_cols_dev_; std::vector<int> rag_ncols_;
  std::vector<void*> rag_owned_;
  bool rag_built_ = false;
  int cur_F_ = 0;
  void drop(Tensor& t) { arena_.release(t.p); t.p = nullptr; if (t.st) { arena_.release(t.st); t.st = nullptr; } }

  // GroupNorm coefficients of the virtual concat [a | b].  Per-channel partial sums come from the producing convolution's
  // epilogue when it emitted them (Tensor::st), otherwise from one streaming pass over the tensor.
  // *bound (optional): per-utterance upper bound of |GroupNorm(+SiLU) output| for an fp16x2 consumer (gn_finalize_kernel)
  void gn_coeffs(const Tensor& a, const Tensor* b, const float* gamma, const float* beta, float** sc, float** sh,
                 const float** bound = nullptr) {
    const int C = a.C + (b ? b->C : 0), HW = a.H * a.W;
    float* bslot = bound ? next_amax() : nullptr;
    if (bound) *bound = bslot;
    const float* st[2] = {a.st, b ? b->st : nullptr};
    int nsub[2] = {a.nsub, b ? b->nsub : 0};
    int srows[2] = {a.srows, b ? b->srows : 1};
    float* tmp[2] = {nullptr, nullptr};
    const Tensor* src[2] = {&a, b};
    for (int k = 0; k < 2; ++k) {
      if (!src[k] || st[k]) continue;
      tmp[k] = arena_.alloc((size_t)B_ * src[k]->C * 2);
      st[k] = tmp[k]; nsub[k] = 1; srows[k] = 1;
      if (!dry_) {
        tock();
        DRT_LAUNCH(gn_chan_stats_kernel, dim3(B_ * src[k]->C), dim3(256), stream_, (const float*)src[k]->p, (const float*)nullptr,
                   src[k]->C, 0, HW, tmp[k], rag_of(a.H), a.H);
        if (noting()) snprintf(prof_note_, sizeof prof_note_, "gn_chan_stats C=%d @%dx%dx%d", src[k]->C, B_, a.H, a.W);
        tick(TC_GN, 4.0 * B_ * (double)src[k]->C * HW, 1);
      }
    }
    *sc = arena_.alloc((size_t)B_ * C);
    *sh = arena_.alloc((size_t)B_ * C);
    if (!dry_) {
      tock();
      const int G = std::min(C / 4, 32);
      const GnFin f{st[0], st[1], a.amax, b ? b->amax : nullptr, gamma, beta, *sc, *sh, bslot, rag_of(a.H),
                    a.C, nsub[0], srows[0], b ? b->C : 0, nsub[1], srows[1], G, HW, a.H, 1e-6f};
      DRT_LAUNCH(gn_finalize_kernel, dim3(G, B_), dim3(256), stream_, f);
      if (noting()) snprintf(prof_note_, sizeof prof_note_, "gn_finalize C=%d nsub=%d @%dx%dx%d", C, nsub[0], B_, a.H, a.W);
      tick(TC_GN, 8.0 * B_ * ((double)a.C * nsub[0] + (b ? (double)b->C * nsub[1] : 0.0)), 1);
    }
    for (int k = 0; k < 2; ++k) if (tmp[k]) arena_.release(tmp[k]);
  }

  // residual shortcut folded into a 3x3 split launch (ConvArgs::sc_*): the 1x1 layer and its (raw) input
  struct Shortcut { const ConvW* w; const Tensor* a; const Tensor* b; };
  // would conv() run this 3x3 layer (GroupNorm producer in front) on the fp16x2 split kernel at this level?  (the rule below; W = dec_W)
  bool runs_on_h2_split3(const ConvW& w, int C, int H, int W) const {
    return w.packed && w.packed_split && w.split_mode == 2 && w.ks == 3 && w.cout > 32 && conv_split_eligible(3, C, 0, w.cout) &&
           (long)((H + 7) / 8) * ((W + 31) / 32) >= split_min_tiles_;
  }
  bool shortcut_foldable(const ConvW& c2, const Tensor& a, const Tensor* b) const {
    return fold_shortcut_ && c2.packed_split && c2.split_mode == 2 && c2.ks == 1 && conv_split_eligible(1, a.C, b ? b->C : 0, c2.cout) &&
           a.amax && (!b || b->amax);
  }

  Tensor conv(const ConvW& w, const Tensor& a, const Tensor* b, const Xform& xf, const float* bias, const float* bias2,
              const float* res, float out_scale, const FwdCtl& ctl, bool emit_stats = false, const Shortcut* sc = nullptr) {
    const int Cin = a.C + (b ? b->C : 0);
    SG_REQUIRE(Cin == w.cin, "conv: channel mismatch");
    Tensor o = new_tensor(w.cout, a.H, a.W);
    const int kc_ = (w.ks == 3) ? 8 : 32;
    const bool use_mfma = w.packed && (b == nullptr || a.C % kc_ == 0);
    // tile choice: the most efficient tile (widest channel block, 8 rows) that still gives the chip enough workgroups
    // (tile_min_blocks_, 2 per CU), falling back towards 32 channels x 4 rows for the coarse U-Net levels and for small
    // batches.  All tile shapes accumulate every output in the same order and emit the same per-row GroupNorm partials
    // (kernels_conv.h), so the choice -- and with it the batch size -- never changes a result bit.
    int co_t = w.co_t, rows_ = a.H >= 8 ? 8 : 4;
    if (use_mfma) {
      const int cand_co[2] = {w.co_t, w.packed32 ? 32 : w.co_t};
      long best = -1;
      bool done = false;
      for (int ci = 0; ci < 2 && !done; ++ci)
        for (int rows = (a.H >= 8 ? 8 : 4); rows >= 4 && !done; rows -= 4) {
          if (ci == 1 && cand_co[1] == cand_co[0]) break;
          const long nblk = (long)B_ * ((a.H + rows - 1) / rows) * ((a.W + 31) / 32) * ((w.cout + cand_co[ci] - 1) / cand_co[ci]);
          if (nblk > best) { best = nblk; co_t = cand_co[ci]; rows_ = rows; }
          if (nblk >= tile_min_blocks_) done = true;
        }
    }
    // fp32-accurate bf16x3 kernel for the wide levels.  Decided per layer and per LEVEL (never by the batch size or the utterance length), because
    // its results differ from the fp32-MFMA kernels in the last bits and an utterance must not depend on its batch.
    const int Wd = dec_W(a.H);          // family decisions by the level, not by the utterance length (see dec_W)
    const long tiles8 = (long)((a.H + 7) / 8) * ((Wd + 31) / 32);
    // Levels with at most chunk_max_tiles_ tiles per nominal image (32 x 64 and below): the fp16x2 split kernel in its 4-row shape with
    // CHUNKED accumulation, so that a small batch can spread the chunks over workgroups (split-K, bit-identical) instead of running 16-32
    // serial stages on 8-32 workgroups; full 3x3 blocks behind a GroupNorm producer only (not the launches with a folded
    // shortcut).  Decided per layer and level, never by the batch or the utterance length: chunking fixes the summation order.
    // The 8 x 16 and 4 x 8 levels joined in round 4: their tiles are half / three quarters empty, yet at batch 32 the layers take 0.057 /
    // 0.038 ms against 0.115 / 0.065 ms on the fp32 kernels (+1.7 % utterances/s); a single utterance pays 9 us per layer for the
    // four times wider output tile of a workgroup (434 -> 452 ms per utterance at batch 1; profiles/r04_coarse_levels.txt).
    const bool coarse_split = coarse_split_ && use_mfma && w.packed_split && w.split_mode == 2 && w.ks == 3 && w.cout > 32 && !sc &&
                              conv_split_eligible(3, a.C, b ? b->C : 0, w.cout) && tiles8 <= chunk_max_tiles_ && xf.bound != nullptr;
    const bool use_split = coarse_split || (use_mfma && w.packed_split &&
                        (conv_split_eligible(w.ks, a.C, b ? b->C : 0, w.cout) || conv_thin_split_eligible(w.ks, a.C, b ? b->C : 0, w.cout)) &&
                        tiles8 >= split_min_tiles_ &&
                        // fp16x2: 3x3 layers scale by the bound of their GroupNorm producer's output, 1x1 layers read the
                        // raw residual stream and scale by the producers' range bounds; either must be known
                        (w.split_mode != 2 || (w.ks == 3 ? xf.bound != nullptr
                                                         : (xf.scale == nullptr && a.amax && (!b || b->amax)))));
    // The wide levels (>= wino_min_tiles_ tiles per nominal image: 64 x 128 and up) run the full 3x3 blocks on the Winograd F(2,3) x
    // fp16x2 kernel: 2/3 of the matrix work of the direct split kernel, which is bound by the energy of its MFMAs (kernels_conv_wino.h).
    // Decided per layer and level like every kernel family; its 4-row shape (launches that cannot fill the chip) gives the same bits.
    const bool use_wino = use_split && !coarse_split && wino_ && w.packed_wino && w.ks == 3 && w.split_mode == 2 && xf.bound != nullptr &&
                          conv_wino_eligible(a.C, b ? b->C : 0, w.cout, 2) && level_of(a.H) <= 5 && tiles8 >= wino_min_tiles_;
    // (the kernel stages aligned column pairs: frame counts are multiples of 64, so every utterance's width is even down to level 5)
    SG_REQUIRE(!use_wino || (a.W % 2 == 0 && (!ragged() || rag_all_mult2(a.H))), "conv: odd width on a Winograd level");
    // Coarse levels (at most 512 pixels per nominal image) on the fp32 kernels: 32-channel tiles with CHUNKED accumulation (decided per
    // layer and image, never by the batch: it fixes the summation order), and -- when even those tiles leave most CUs idle
    // (small batches) -- the chunks spread over workgroups (split-K, bit-identical): a K loop of 32-64 serial stages was the
    // latency of these launches (60-120 us each at batch 1, profiles/r02_prof_dump_b1_per_launch.txt)
    int kchunk = 0, ksplit = 1;
    float* partial = nullptr;
    if (coarse_split) {
      const int nstages = Cin / 16;
      kchunk = std::max(2, (nstages + 7) / 8);
      const int nchunks = (nstages + kchunk - 1) / kchunk;
      const long nblk = (long)B_ * ((a.H + 3) / 4) * ((a.W + 31) / 32) * (w.cout / 128);
      if (nchunks > 1 && nblk * coarse_splitk_div_ <= tile_min_blocks_) {
        ksplit = nchunks;
        partial = arena_.alloc((size_t)nchunks * w.cout * pix_total(a.H, a.W));
      }
    }
    if (use_mfma && !use_split && coarse_chunked_ && (long)a.H * Wd <= 512 && (w.co_t == 32 || w.packed32)) {
      co_t = 32;
      const long nblk8 = (long)B_ * ((a.H + 7) / 8) * ((a.W + 31) / 32) * ((w.cout + 31) / 32);
      rows_ = (a.H >= 8 && nblk8 >= tile_min_blocks_) ? 8 : 4;
      const int nstages = Cin / kc_;
      kchunk = std::max(w.ks == 3 ? 4 : 2, (nstages + 7) / 8);
      const int nchunks = (nstages + kchunk - 1) / kchunk;
      const long nblk = (long)B_ * ((a.H + rows_ - 1) / rows_) * ((a.W + 31) / 32) * ((w.cout + 31) / 32);
      if (nchunks > 1 && nblk * 2 <= tile_min_blocks_) {
        ksplit = nchunks;
        partial = arena_.alloc((size_t)nchunks * w.cout * pix_total(a.H, a.W));
      }
    }
    if (emit_stats && use_mfma && fuse_gn_stats_) {
      // the split kernels emit one partial pair per 4 image rows, the fp32 kernels one per row (ConvArgs::stats_rows): a property
      // of the kernel family, which is a property of the layer and level
      o.srows = use_split ? conv_split_stats_rows(w.ks, w.cout) : 1;
      SG_REQUIRE(o.srows == 1 || !ragged() || a.H % 4 == 0, "ragged batch: 4-row statistics sub-tiles need H % 4 == 0");
      o.nsub = conv_plan_nsub(a.H, a.W, o.srows);
      o.st = arena_.alloc((size_t)B_ * w.cout * o.nsub * 2);
    } else if (emit_stats && fuse_gn_stats_) {
      // direct kernel: one statistics pass right behind it, kept with the tensor -- these outputs (entry conv, Combine)
      // are normalised twice, by the next block and again as skip connections on the way up
      o.nsub = 1;
      o.st = arena_.alloc((size_t)B_ * w.cout * 2);
    }
    o.amax = next_amax();
    if (dry_) { if (partial) arena_.release(partial); return o; }
    ConvArgs ca{};
    ca.xcd_map = conv_xcd_map_ ? 1 : 0;
    ca.stats_out = o.st; ca.stats_nsub = o.nsub; ca.stats_rows = o.srows;
    ca.amax_out = o.amax;
    ca.src1 = a.p; ca.src2 = b ? b->p : nullptr; ca.C1 = a.C; ca.C2 = b ? b->C : 0;
    ca.bias = bias; ca.bias2 = bias2; ca.bias2_bstride = ctl.bias_bstride; ca.bias2_sstride = ctl.bias_sstride;
    ca.step_ptr = bias2 ? ctl.bias_step : nullptr;
    ca.in_scale = xf.scale; ca.in_shift = xf.shift; ca.in_act = xf.act;
    ca.res = res; ca.out_scale = out_scale; ca.out = o.p; ca.Cout = w.cout; ca.B = B_; ca.H = a.H; ca.W = a.W;
    if (ragged()) {
      const Rag rg = rag_of(a.H);
      ca.rag_w = rg.w; ca.rag_off = rg.off; ca.rag_soff = rg.soff; ca.rag_slab = (long long)w.cout * (long long)pix_total(a.H, a.W);
      ca.rag_vec_ok = rag_all_mult4(a.H) ? 1 : 0;
      if (rag_prefix_) { const int l = level_of(a.H); ca.rag_cols = rag_cols_dev_.at(l); ca.rag_ncols = rag_ncols_.at(l); }
    }
    tock();
    double fl = 2.0 * B_ * (double)w.cout * Cin * w.ks * w.ks * a.H * a.W;
    if (sc) {
      SG_REQUIRE(use_split && w.ks == 3 && w.split_mode == 2 && w.cout > 32 && xf.scale && xf.act && !res, "conv: shortcut fold on an ineligible launch");
      const int Cs = sc->a->C + (sc->b ? sc->b->C : 0);
      SG_REQUIRE(Cs == sc->w->cin && sc->w->cout == w.cout && sc->a->H == a.H && sc->a->W == a.W, "conv: shortcut shape mismatch");
      ca.sc_src1 = sc->a->p; ca.sc_src2 = sc->b ? sc->b->p : nullptr; ca.sc_C1 = sc->a->C; ca.sc_C2 = sc->b ? sc->b->C : 0;
      ca.sc_w = sc->w->packed_split; ca.sc_scale = sc->w->split_scale; ca.sc_bias = sc->w->bias;
      ca.sc_amax1 = sc->a->amax; ca.sc_amax2 = sc->b ? sc->b->amax : nullptr;
      fl += 2.0 * B_ * (double)w.cout * Cs * a.H * a.W;
    }
    // the C -> 4 convolutions of the output pyramid: exact-fp32 VALU kernel (kernels_conv_thin.h) -- on the matrix pipe
    // seven eighths of their work was padding (decided by the layer's shape and U-Net level alone: never by batch or utterance length)
    // (the two finest levels: below them a launch at batch 1 has few 16 x 64 tiles and their 32-64 serial stages are slower than the MFMA path's split-K)
    // (conv3x3_thin_kernel has no time-embedding row, accumulator scale or folded shortcut: a layer that carries one stays on the MFMA shapes)
    const bool use_thin = w.packed_thin && conv_thin_eligible(w.ks, a.C, b ? b->C : 0, w.cout) && !emit_stats && level_of(a.H) <= 1 && !bias2 && !sc;
    if (use_thin) {
      ca.w = w.packed_thin; ca.stats_out = nullptr;
      launch_conv_thin(ca, stream_);
      if (noting()) snprintf(prof_note_, sizeof prof_note_, "conv3x3-thin %d->%d @%dx%dx%d%s%s", Cin, w.cout, B_, a.H, a.W, res ? " +res" : "", xf.scale ? " +gn" : "");
      tick(TC_CONV3, fl);
      if (partial) arena_.release(partial);
    } else if (use_split) {
      ca.w = w.packed_split; ca.co_scale = w.split_scale;      // (null for bf16x3: no scale)
      if (w.ks == 1 && w.split_mode == 2) { ca.amax1 = a.amax; ca.amax2 = b ? b->amax : nullptr; }
      if (w.ks == 3 && w.split_mode == 2) ca.xbound = xf.bound;
      // 4-row workgroups when 8-row ones would leave CUs idle (bit-identical results, so this may follow the batch size)
      const long nblk8 = (long)B_ * ((a.H + 7) / 8) * ((a.W + 31) / 32) * ((w.cout + 127) / 128);
      const bool rows4 = coarse_split || nblk8 < tile_min_blocks_;
      if (coarse_split) { ca.kchunk_stages = kchunk; ca.partial = partial; }
      if (use_wino) {
        ca.w = w.packed_wino; ca.co_scale = w.wino_scale; ca.acc_scale = nullptr;
        launch_conv_wino(ca, stream_, nblk8 < tile_min_blocks_);   // one 512-thread workgroup per CU: the 8-row shape from two rounds of the chip
      } else {
        launch_conv_split(ca, w.ks, w.split_mode, stream_, rows4, 0, ksplit);
      }
      if (coarse_split && partial) arena_.release(partial);
      if (noting()) {
        char scn[24] = "";
        if (sc) snprintf(scn, sizeof scn, " +shortcut(%d)", ca.sc_C1 + ca.sc_C2);      // (input channels of the folded 1x1)
        snprintf(prof_note_, sizeof prof_note_, "conv3x3-%s %d->%d @%dx%dx%d%s%s%s%s", use_wino ? "wino" : "split", Cin, w.cout, B_, a.H, a.W, res ? " +res" : "",
                 xf.scale ? " +gn" : "", scn, ca.rag_cols ? " existing-tiles" : "");
      }
      tick(w.ks == 3 ? (w.cout >= 128 ? TC_CONV3_BIG : TC_CONV3) : TC_CONV1, fl);
    } else if (use_mfma) {
      ConvPlan pl{co_t, rows_, true};
      ca.w = (co_t == w.co_t) ? w.packed : w.packed32;
      ca.kchunk_stages = kchunk; ca.partial = partial;
      launch_conv_mfma(ca, w.ks, pl, stream_, -1, ksplit);
      if (partial) arena_.release(partial);
      if (noting())
        snprintf(prof_note_, sizeof prof_note_, "conv%dx%d %d->%d @%dx%dx%d tile %dco x %drows%s%s%s%s", w.ks, w.ks, Cin, w.cout, B_, a.H, a.W,
                 co_t, rows_, res ? " +res" : "", xf.scale ? " +gn" : "", ksplit > 1 ? " split-K" : "", ca.rag_cols ? " existing-tiles" : "");
      // class "wide" = the dominant kernel family only: the split kernels, or (SGMSE_CONV_SPLIT=0) the fp32 128 x 256 tile
      tick(w.ks == 3 ? ((split_mode_ == 0 && co_t == 128 && pl.rows == 8) ? TC_CONV3_BIG : TC_CONV3) : TC_CONV1, fl);
    } else {
      ca.w = w.oihw;
      ca.stats_out = nullptr;
      launch_conv_direct(ca, w.ks, stream_);
      if (noting()) snprintf(prof_note_, sizeof prof_note_, "conv-direct%dx%d %d->%d @%dx%dx%d", w.ks, w.ks, Cin, w.cout, B_, a.H, a.W);
      tick(TC_DIRECT, fl);
      if (o.st) {
        tock();
        DRT_LAUNCH(gn_chan_stats_kernel, dim3(B_ * w.cout), dim3(256), stream_, (const float*)o.p, (const float*)nullptr, w.cout, 0,
                   a.H * a.W, o.st, rag_of(a.H), a.H);
        if (noting()) snprintf(prof_note_, sizeof prof_note_, "gn_chan_stats(direct) C=%d @%dx%dx%d", w.cout, B_, a.H, a.W);
        tick(TC_GN, 4.0 * B_ * (double)w.cout * a.H * a.W, 1);
      }
    }
    return o;
  }

  void launch_fir(const FirArgs& fa, bool up, bool by_level = false) {
    const int H = fa.H, W = fa.W, BC = fa.BC;
    // tiled or per-pixel kernel: by the LEVEL (dec_W) inside the network, so that an utterance's FIR never depends on its length
    const bool tiled = by_level ? (W % 4 == 0 && dec_W(H) >= 64 && rag_all_mult4(H)) : fir_use_tiled(W);
    if (tiled && !fir_scalar_) {
      if (up) DRT_LAUNCH(fir_up2_tiled_kernel, dim3(((W + 63) / 64) * ((H + 7) / 8), BC), dim3(256), stream_, fa);
      else DRT_LAUNCH(fir_down2_tiled_kernel, dim3(((W / 2 + 63) / 64) * ((H / 2 + 7) / 8), BC), dim3(256), stream_, fa);
    } else {
      if (up) DRT_LAUNCH(fir_up2_kernel, dim3((H * W + 255) / 256, BC), dim3(256), stream_, fa);
      else DRT_LAUNCH(fir_down2_kernel, dim3(((H / 2) * (W / 2) + 255) / 256, BC), dim3(256), stream_, fa);
    }
  }

  // FIR x2 / /2 of `a` through the fused producer `xf`; with `raw` also FIR(a) itself from the same pass
  Tensor fir(const Tensor& a, bool up, const Xform& xf, Tensor* raw = nullptr) {
    Tensor o = up ? new_tensor(a.C, a.H * 2, a.W * 2) : new_tensor(a.C, a.H / 2, a.W / 2);
    if (raw) {
      *raw = up ? new_tensor(a.C, a.H * 2, a.W * 2) : new_tensor(a.C, a.H / 2, a.W / 2);
      raw->amax = a.amax;      // the [1,3,3,1] resamplers are convex combinations (per output phase): max|FIR(x)| <= max|x|
    }
    if (dry_) return o;
    FirArgs fa{a.p, o.p, xf.scale, xf.shift, xf.act, B_ * a.C, a.H, a.W, raw ? raw->p : nullptr, a.C, rag_of(a.H)};
    tock();
    launch_fir(fa, up, true);
    if (noting()) snprintf(prof_note_, sizeof prof_note_, "fir-%s C=%d @%dx%dx%d%s", up ? "up" : "down", a.C, B_, a.H, a.W, raw ? " +raw" : "");
    tick(TC_FIR, 4.0 * B_ * (double)a.C * (a.H * a.W + (raw ? 2.0 : 1.0) * o.H * o.W));
    return o;
  }

  // ResnetBlockBigGANpp.forward (layerspp.py:242-274).  Inputs stay owned by the caller.
  Tensor res_block(const Mod& m, Tensor& a, Tensor* b, const FwdCtl& ctl) {
    const ResW& r = res_.at(m.idx);
    float *sc0, *sh0, *sc1, *sh1;
    const float *bd0, *bd1;
    gn_coeffs(a, b, r.g0w, r.g0b, &sc0, &sh0, &bd0);
    Xform x0{sc0, sh0, 1, bd0};
    const float* temb = ctl.bias_table ? ctl.bias_table + r.temb_off : nullptr;
    Tensor h, xs;       // xs: resampled shortcut input (only for up/down)
    bool have_xs = false;
    // The block's 1x1 shortcut Conv_2 (layerspp.py:266-269), where it is its own launch (not folded into Conv_1's launch: the levels
    // below the split kernels' threshold, and any layer the fold's conditions exclude): it needs the block input only and its result only
    // at the end of the block, so at small batches it runs on the side stream beside GroupNorm - Conv_0 - GroupNorm (round 6).
    Tensor sh_t; bool sc_side = false;
    auto shortcut_early = [&](const Tensor& sa, const Tensor* sb, int Ch, int Hh) {
      if (!r.has_c2) return;
      const bool fold = runs_on_h2_split3(r.c1, Ch, Hh, dec_W(Hh)) && shortcut_foldable(r.c2, sa, sb) && !((nofold_levels_ >> level_of(Hh)) & 1);
      if (fold || !side_enabled()) return;
      side_begin();
      sh_t = conv(r.c2, sa, sb, Xform{}, r.c2.bias, nullptr, nullptr, 1.f, ctl);
      side_end();
      sc_side = true;
    };
    if (m.up || m.down) {
      SG_REQUIRE(b == nullptr, "resample block with concat input");
      Tensor hr = fir(a, m.up, x0, &xs);
      have_xs = true;
      shortcut_early(xs, nullptr, r.c0.cout, hr.H);
      h = conv(r.c0, hr, nullptr, Xform{nullptr, nullptr, 0, bd0}, nullptr, temb, nullptr, 1.f, ctl, true);
      drop(hr);
    } else {
      shortcut_early(a, b, r.c0.cout, a.H);
      h = conv(r.c0, a, b, x0, nullptr, temb, nullptr, 1.f, ctl, true);
    }
    arena_.release(sc0); arena_.release(sh0);
    gn_coeffs(h, nullptr, r.g1w, r.g1b, &sc1, &sh1, &bd1);
    Xform x1{sc1, sh1, 1, bd1};
    Tensor out;
    const float inv_sqrt2 = 0.70710678118654752440f;
    if (r.has_c2) {
      const Tensor& sa = have_xs ? xs : a;
      const Tensor* sb = have_xs ? nullptr : b;
      if (runs_on_h2_split3(r.c1, h.C, h.H, dec_W(h.H)) && shortcut_foldable(r.c2, sa, sb) && !((nofold_levels_ >> level_of(h.H)) & 1)) {
        // (Conv_1(h) + Conv_2(x)) / sqrt 2 as one accumulation: the shortcut's K-stages run inside the 3x3 launch
        const Shortcut scin{&r.c2, &sa, sb};
        out = conv(r.c1, h, nullptr, x1, r.c1.bias, nullptr, nullptr, inv_sqrt2, ctl, true, &scin);
      } else {
        if (sc_side) side_join();                    // (also joins a pyramid branch still pending from the level above)
        else sh_t = conv(r.c2, sa, sb, Xform{}, r.c2.bias, nullptr, nullptr, 1.f, ctl);
        out = conv(r.c1, h, nullptr, x1, r.c1.bias, nullptr, sh_t.p, inv_sqrt2, ctl, true);
        drop(sh_t);
      }
    } else {
      SG_REQUIRE(b == nullptr, "identity shortcut with concat input");
      out = conv(r.c1, h, nullptr, x1, r.c1.bias, nullptr, a.p, inv_sqrt2, ctl, true);
    }
    if (have_xs) drop(xs);
    drop(h);
    arena_.release(sc1); arena_.release(sh1);
    return out;
  }

  // AttnBlockpp.forward (layerspp.py:75-91)
  Tensor attn_block(const Mod& m, Tensor& x, const FwdCtl& ctl) {
    const AttnW& w = attn_.at(m.idx);
    float *sc, *sh;
    gn_coeffs(x, nullptr, w.gw, w.gb, &sc, &sh);
    Tensor qkv = conv(w.qkv, x, nullptr, Xform{sc, sh, 0}, w.qkv.bias, nullptr, nullptr, 1.f, ctl);
    arena_.release(sc); arena_.release(sh);
    Tensor o = new_tensor(x.C, x.H, x.W);
    if (!dry_) {
      AttnArgs aa{qkv.p, o.p, B_, x.C, x.H * x.W, 1.0f / sqrtf((float)x.C), rag_of(x.H), x.H};
      tock();
      SG_REQUIRE(launch_attn_core(aa, stream_), "attention: unsupported channel count");
      if (noting()) snprintf(prof_note_, sizeof prof_note_, "attention C=%d S=%d B=%d", x.C, x.H * x.W, B_);
      tick(TC_ATTN, 4.0 * B_ * (double)x.C * (x.H * x.W) * (double)(x.H * x.W));
    }
    drop(qkv);
    Tensor out = conv(w.proj, o, nullptr, Xform{}, w.proj.bias, nullptr, x.p, 0.70710678118654752440f, ctl, true);
    drop(o);
    return out;
  }

  // ---- side stream (round 6) ---------------------------------------------------------------------------------------------------------
  // Launches that the main chain needs only later -- the output-pyramid branch of a level (run_forward), the 1x1 shortcut of a residual
  // block that is not folded into the block's second 3x3 launch (res_block) -- go to the engine's SECOND STREAM at small batches, where
  // they are latency, not throughput:
  //   side_begin() .. side_end(): the launches in between run on the side stream, ordered behind everything the main stream holds at
  //     side_begin() (an event recorded there); sections may follow each other before a join (the side stream keeps their order);
  //   side_join(): the main stream waits for everything the side stream holds.
  // From the first side_begin() to the join every arena release is DEFERRED (Arena::defer_begin): memory freed on one stream's behalf is
  // not handed to the other stream's later launches while its last user may still run -- in the dry run too, so that the planned arena
  // is the one the real run uses.  Same kernels, same arguments, same bits (test_sampler_graph_equals_eager, the batch-independence and
  // device-memory tests, tools/probes/side_stream_bits_probe.py).  Off above side_max_batch_ utterances (no deferral either: the arena of
  // the large batches is unchanged) and for instrumented evaluations (one stream: profile_forward times launches that follow each other).
  bool side_enabled() const { return side_stream_on_ && B_ <= side_max_batch_; }
  bool side_active() const { return side_enabled() && !prof_ && !dry_ && !drt::is_emulator(); }
  drt::event_t* side_event() {
    if (side_ev_next_ == side_ev_.size()) { side_ev_.emplace_back(); SG_CHECK(drt::event_create_order(&side_ev_.back())); }
    return &side_ev_[side_ev_next_++];
  }
  void side_begin() {
    SG_REQUIRE(!side_open_, "side stream: nested section");
    side_open_ = true;
    if (!side_enabled()) return;
    if (!side_pending_) arena_.defer_begin();
    if (!side_active()) return;
    if (!side_stream_ready_) { SG_CHECK(drt::stream_create(&side_stream_)); side_stream_ready_ = true; }
    drt::event_t* e = side_event();
    SG_CHECK(drt::event_record(e, stream_));
    SG_CHECK(drt::stream_wait_event(side_stream_, e));
    std::swap(stream_, side_stream_);
    side_swapped_ = true;
  }
  void side_end() {
    SG_REQUIRE(side_open_, "side stream: end without begin");
    side_open_ = false;
    if (!side_enabled()) return;
    side_pending_ = true;
    if (!side_swapped_) return;
    drt::event_t* e = side_event();
    SG_CHECK(drt::event_record(e, stream_));         // (stream_ is the side stream here)
    std::swap(stream_, side_stream_);
    side_swapped_ = false;
    side_join_ev_ = e;                               // (the side stream is in order: its latest event covers the earlier sections)
  }
  void side_join() {
    if (!side_pending_) return;
    side_pending_ = false;
    if (side_join_ev_) { SG_CHECK(drt::stream_wait_event(stream_, side_join_ev_)); side_join_ev_ = nullptr; }
    arena_.defer_end();
  }
  drt::stream_t side_stream_{}; bool side_stream_ready_ = false, side_open_ = false, side_pending_ = false, side_swapped_ = false;
  std::deque<drt::event_t> side_ev_; size_t side_ev_next_ = 0; drt::event_t* side_join_ev_ = nullptr;
  bool side_stream_on_ = true; int side_max_batch_ = 8;

  // NCSNpp.forward (ncsnpp.py:256-419) / NCSNpp_48k.forward
  void run_forward(const float2* x, long long xbs, const float2* y, long long ybs, float2* out, int B, int F, int T,
                   const FwdCtl& ctl_in) {
    side_ev_next_ = 0; side_open_ = side_pending_ = false; side_join_ev_ = nullptr;
    try { run_forward_main(x, xbs, y, ybs, out, B, F, T, ctl_in); }
    catch (...) {                 // never leave the engine on its side stream
      if (side_swapped_) { std::swap(stream_, side_stream_); side_swapped_ = false; }
      side_open_ = side_pending_ = false; side_join_ev_ = nullptr;
      throw;
    }
  }
  void run_forward_main(const float2* x, long long xbs, const float2* y, long long ybs, float2* out, int B, int F, int T,
                        const FwdCtl& ctl_in) {
    FwdCtl ctl = ctl_in;
    ctl.bias_step = ctl.step_ptr;
    if (!dry_ && ctl.step_ptr && ctl.bias_table && ctl.bias_bstride == 0 && tot_temb_ > 0) {
      // sampler loop: this step's bias row into a fixed buffer (bias_select_kernel), the convolutions read it without indirection
      DRT_LAUNCH(bias_select_kernel, dim3((tot_temb_ + 255) / 256), dim3(256), stream_, ctl.bias_table, ctl.bias_sstride, ctl.step_ptr,
                 bias_cur_, tot_temb_);
      ctl.bias_table = bias_cur_; ctl.bias_sstride = 0; ctl.bias_step = nullptr;
    }
    B_ = B;
    cur_F_ = F;
    amax_next_ = 0;
    lds_poison_count_ = 0;
    if (!dry_ && poison_ && arena_base_) SG_CHECK(drt::memset_dev(arena_base_, 0xFF, arena_cap_, stream_));
    if (!dry_ && amax_pool_) {
      const int n = amax_slots_ * B * kAmaxSpread;
      DRT_LAUNCH(zero_floats_kernel, dim3((n + 255) / 256), dim3(256), stream_, amax_pool_, n);
    }
    const NetCfg& c = cfg_;
    const int L = c.n_levels;
    size_t mi = 3;
    auto next = [&]() -> const Mod& { SG_REQUIRE(mi < layout_.size(), "layout exhausted"); return layout_[mi++]; };
    const int FT = F * T;
    std::vector<Tensor> hs;
    Tensor xr = new_tensor(4, F, T);
    const bool entry8 = entry_mfma_ && entry8_.packed && layout_[mi].idx == entry8_idx_;
    Tensor xr8{};
    if (entry8) xr8 = new_tensor(entry8_.cin, F, T);   // the same four planes + planes of zeros: one K-stage of the MFMA kernel
    if (!dry_) { tock(); const WrapCoef wc{ctl.coef, ctl.coef_bstride, ctl.coef_sstride, ctl.step_ptr, ctl.sign};
      DRT_LAUNCH(entry_kernel, dim3((FT + 255) / 256, B), dim3(256), stream_, x, xbs, y, ybs, xr.p, entry8 ? xr8.p : (float*)nullptr,
                 entry8 ? entry8_.cin : 0, FT, wc, rag_of(F), F, ctl.y_plane);
      tick(TC_MISC, (32.0 + (entry8 ? 4.0 * entry8_.cin : 0.0)) * B * FT); }
    {
      const Mod& m = next();
      const ConvW& w = entry8 ? entry8_ : conv_.at(m.idx);
      hs.push_back(conv(w, entry8 ? xr8 : xr, nullptr, Xform{}, w.bias, nullptr, nullptr, 1.f, ctl, true));
      if (entry8) drop(xr8);
    }
    Tensor pyr_in = xr;   // input pyramid (ncsnpp.py:293-296); released at the end / when replaced
    bool pyr_in_is_xr = true;
    for (int l = 0; l < L; ++l) {
      for (int rb = 0; rb < c.num_res_blocks; ++rb) {
        const Mod& m = next();
        Tensor h = res_block(m, hs.back(), nullptr, ctl);
        // the reference's forward places attention by the ACTUAL height (ncsnpp.py:308) while its module list was built from the
        // configured image_size (ncsnpp.py:189-193): an input of another height runs off the list there (TypeError); say why
        SG_REQUIRE(cfg_has_attn(c, h.H) == cfg_has_attn(c, c.image_size >> l),
                   "input height does not match the image_size the network was built for (attention placement, ncsnpp.py:308)");
        if (cfg_has_attn(c, h.H)) {
          const Mod& ma = next();
          Tensor h2 = attn_block(ma, h, ctl);
          drop(h);
          h = h2;
        }
        hs.push_back(h);
      }
      if (l != L - 1) {
        const Mod& m = next();
        Tensor h = res_block(m, hs.back(), nullptr, ctl);
        if (c.progressive_input == 1) {
          const Mod& mc = next();
          Tensor np = fir(pyr_in, false, Xform{});
          if (!pyr_in_is_xr) drop(pyr_in);
          pyr_in = np; pyr_in_is_xr = false;
          const ConvW& w = conv_.at(mc.idx);
          Tensor h2 = conv(w, pyr_in, nullptr, Xform{}, w.bias, nullptr, h.p, 1.f, ctl, true);
          drop(h);
          h = h2;
        }
        hs.push_back(h);
      }
    }
    if (!pyr_in_is_xr) drop(pyr_in);
    drop(xr);

    Tensor h = hs.back();   // not popped: still referenced by the skip stack (ncsnpp.py:337)
    {
      const Mod& m1 = next(); Tensor a = res_block(m1, h, nullptr, ctl);
      const Mod& m2 = next(); Tensor b2 = attn_block(m2, a, ctl); drop(a);
      const Mod& m3 = next(); h = res_block(m3, b2, nullptr, ctl); drop(b2);
    }
    Tensor pyramid; bool have_pyr = false;
    for (int l = L - 1; l >= 0; --l) {
      for (int rb = 0; rb < c.num_res_blocks + 1; ++rb) {
        const Mod& m = next();
        Tensor skip = hs.back(); hs.pop_back();
        Tensor o = res_block(m, h, &skip, ctl);
        drop(h); drop(skip);
        h = o;
      }
      SG_REQUIRE(cfg_has_attn(c, h.H) == cfg_has_attn(c, c.image_size >> l),
                 "input height does not match the image_size the network was built for (attention placement, ncsnpp.py:354)");
      if (cfg_has_attn(c, h.H)) {
        const Mod& ma = next();
        Tensor h2 = attn_block(ma, h, ctl);
        drop(h); h = h2;
      }
      if (c.progressive == 1) {
        // The output-pyramid branch of the level (GroupNorm finalize, C -> 4 convolution, 4-channel FIR; ncsnpp.py:358-379) reads h and the
        // previous level's pyramid only and rejoins the main chain at the network's exit: it runs on the engine's SIDE STREAM beside the
        // level's up-sampling block (round 6; small batches, where these launches are latency, not throughput).  Same kernels, same
        // arguments, same bits; the join sits in front of drop(h).
        const Mod& mg = next(); const Mod& mc = next();
        side_begin();
        float *sc, *sh;
        const float* bd;
        gn_coeffs(h, nullptr, gn_.at(mg.idx).first, gn_.at(mg.idx).second, &sc, &sh, &bd);
        const ConvW& w = conv_.at(mc.idx);
        Tensor up; bool have_up = false;
        if (have_pyr) { up = fir(pyramid, true, Xform{}); have_up = true; drop(pyramid); }
        pyramid = conv(w, h, nullptr, Xform{sc, sh, 1, bd}, w.bias, nullptr, have_up ? up.p : nullptr, 1.f, ctl);
        have_pyr = true;
        if (have_up) drop(up);
        arena_.release(sc); arena_.release(sh);
        side_end();
      }
      if (l != 0) {
        const Mod& m = next();
        Tensor o = res_block(m, h, nullptr, ctl);
        side_join();
        drop(h); h = o;
      }
    }
    side_join();
    SG_REQUIRE(hs.empty(), "skip stack not empty");
    Tensor h4;
    if (c.progressive == 1) { h4 = pyramid; drop(h); }
    else {
      const Mod& mg = next(); const Mod& mc = next();
      float *sc, *sh;
      const float* bd;
      gn_coeffs(h, nullptr, gn_.at(mg.idx).first, gn_.at(mg.idx).second, &sc, &sh, &bd);
      const ConvW& w = conv_.at(mc.idx);
      h4 = conv(w, h, nullptr, Xform{sc, sh, 1, bd}, w.bias, nullptr, nullptr, 1.f, ctl);
      arena_.release(sc); arena_.release(sh);
      drop(h);
    }
    SG_REQUIRE(mi == layout_.size(), "forward did not consume every module");
    if (!dry_) {
      ExitArgs ea{h4.p, Wp("output_layer.weight"), Wp("output_layer.bias"), ctl.tvals, ctl.t_bstride, ctl.t_sstride, ctl.step_ptr,
                  c.variant == 1 ? 1 : 0, c.scale_by_sigma,
                  WrapCoef{ctl.coef, ctl.coef_bstride, ctl.coef_sstride, ctl.step_ptr, ctl.sign}, x, xbs, out, FT, rag_of(F), F};
      tock();
      DRT_LAUNCH(exit_kernel, dim3((FT + 255) / 256, B), dim3(256), stream_, ea);
      tick(TC_MISC, 24.0 * B * FT);
      check_launch();
    }
    drop(h4);
  }

  // what a captured step depends on.  The Philox seed is NOT part of it: the sampler kernels read it from a device word
  // (upload_tables), so one capture serves every utterance / batch of a run (the Python samplers draw a fresh seed per call)
  struct GraphKey {
    int B, F, T, corr, ncorr, pred, pf; const void* y; const void* noise; float theta; int dps;
    bool operator==(const GraphKey& o) const {
      return B == o.B && F == o.F && T == o.T && corr == o.corr && ncorr == o.ncorr && pred == o.pred && pf == o.pf && y == o.y &&
             noise == o.noise && theta == o.theta && dps == o.dps;
    }
  };
  // Per-call sampler constants -> device: the step table, the time steps, the score-wrapper rows (may be empty) and the noise
  // table -- seed word + one noise-stream id per utterance (SamplerArgs::seed; ids given by set_noise_streams are consumed by the
  // next sampler call of the same batch size, otherwise utterance b gets stream b).  The host side of the copies is ONE page-locked
  // staging buffer, so the copies are asynchronous and a sampler call never synchronises the host with the stream (no
  // hidden host syncs): the caller may already enqueue the next batch's front end while this one samples.  The staging buffer is
  // reused by the next call once the event recorded behind these copies has passed (it has, long before: the copies are the
  // first thing a call enqueues).
  const unsigned long long* upload_tables(const std::vector<float>& tab, const std::vector<float>& tv, const std::vector<float>& cf,
                                          unsigned long long seed, int B) {
    SG_REQUIRE(B + 1 <= kMaxStreams, "batch too large for the noise-stream table");
    if (!seed_dev_) seed_dev_ = static_cast<unsigned long long*>(dev_alloc((size_t)kMaxStreams * 8));
    const size_t nb_seed = ((size_t)B + 1) * 8, nb_tab = tab.size() * 4, nb_tv = tv.size() * 4, nb_cf = cf.size() * 4;
    const size_t total = nb_seed + nb_tab + nb_tv + nb_cf;
    if (hstage_pending_) { SG_CHECK(drt::event_sync(&hstage_ev_)); hstage_pending_ = false; }
    if (total > hstage_cap_) {
      if (hstage_) drt::free_host(hstage_);
      hstage_cap_ = std::max(total, size_t(1) << 16);
      SG_CHECK(drt::malloc_host(reinterpret_cast<void**>(&hstage_), hstage_cap_));
    }
    unsigned long long* hs = reinterpret_cast<unsigned long long*>(hstage_);
    hs[0] = seed;
    const bool given = (int)streams_next_.size() == B;
    for (int b = 0; b < B; ++b) hs[1 + b] = given ? streams_next_[b] : (unsigned long long)b;
    streams_next_.clear();
    char* q = hstage_ + nb_seed;
    memcpy(q, tab.data(), nb_tab); memcpy(q + nb_tab, tv.data(), nb_tv);
    if (nb_cf) memcpy(q + nb_tab + nb_tv, cf.data(), nb_cf);
    SG_CHECK(drt::memcpy_h2d(seed_dev_, hs, nb_seed, stream_));
    SG_CHECK(drt::memcpy_h2d(step_table_, q, nb_tab, stream_));
    SG_CHECK(drt::memcpy_h2d(tsteps_, q + nb_tab, nb_tv, stream_));
    if (nb_cf) SG_CHECK(drt::memcpy_h2d(coef_table_, q + nb_tab + nb_tv, nb_cf, stream_));
    if (!hstage_ev_init_) { SG_CHECK(drt::event_create(&hstage_ev_)); hstage_ev_init_ = true; }
    SG_CHECK(drt::event_record(&hstage_ev_, stream_));
    hstage_pending_ = true;
    return seed_dev_;
  }
  char* hstage_ = nullptr; size_t hstage_cap_ = 0; drt::event_t hstage_ev_{}; bool hstage_ev_init_ = false, hstage_pending_ = false;
  static constexpr int kMaxStreams = 4096;
  std::vector<unsigned long long> streams_next_;
  unsigned long long* seed_dev_ = nullptr;
  int graph_captures_ = 0;

  int device_;
  drt::stream_t stream_;
  NetCfg cfg_;
  std::vector<Mod> layout_;
  std::map<std::string, const float*> W_;
  float* blob_ = nullptr; size_t blob_elems_ = 0;
  std::map<int, ResW> res_;
  std::map<int, AttnW> attn_;
  std::map<int, ConvW> conv_;
  std::map<int, std::pair<const float*, const float*>> gn_;
  DenseDesc* dense_descs_ = nullptr; int n_dense_ = 0; int tot_temb_ = 0;
  bool weights_ready_ = false;
  std::vector<void*> owned_, wowned_;
  std::map<int, const float2*> twiddles_;

  Arena arena_; char* arena_base_ = nullptr; size_t arena_cap_ = 0;
  bool dry_ = false;
  // measurement knobs, re-read from the environment at every configure (so one process can compare settings)
  // Runtime switches, re-read from the environment at every configure / weight load (so one process can compare settings).
  // User-facing: SGMSE_CONV_SPLIT, SGMSE_WINO, SGMSE_CONV_XCD_MAP, SGMSE_RAGGED_PREFIX, SGMSE_DEBUG_SYNC,
  // SGMSE_PROFILE_DUMP.  Test hooks (what the bitwise / parity tests toggle to reach a code path; not for users):
  // SGMSE_TILE_MIN_BLOCKS, SGMSE_SPLIT_MIN_TILES, SGMSE_WINO_MIN_TILES, SGMSE_FUSE_GN_STATS, SGMSE_POISON, SGMSE_CONV_VARIANT.
  // Everything else that used to be a switch is a constant now: the losing side of each was measured and removed (round 4).
  void read_knobs() {
    auto flag = [](const char* name, bool dflt) { const char* e = getenv(name); return e ? e[0] == '1' : dflt; };
    const char* e = getenv("SGMSE_CONV_SPLIT");          // 0: fp32 MFMA only, 1: bf16x3, 2: fp16x2 (+ Winograd, see SGMSE_WINO) on the wide levels
    split_mode_ = e ? atoi(e) : SGMSE_CONV_SPLIT_DEFAULT;
    SG_REQUIRE(split_mode_ >= 0 && split_mode_ <= 2, "SGMSE_CONV_SPLIT must be 0, 1 or 2");
    wino_ = flag("SGMSE_WINO", true);                    // Winograd F(2,3) x fp16x2 kernel on the wide levels (0: the direct fp16x2 split kernel there too)
    // ragged convolution launches over the tiles that exist (ConvArgs::rag_cols) and the XCD-aware tile order of the convolution
    // kernels (ConvArgs::xcd_map): both bit-identical, both measured in round 4 (profiles/r04_knobs_ab.txt: ragged batches 1.15 -> 1.04x
    // per frame; +3.5-4 % at T = 512 and T = 448) and on since
    rag_prefix_ = flag("SGMSE_RAGGED_PREFIX", true);
    conv_xcd_map_ = flag("SGMSE_CONV_XCD_MAP", true);
    debug_sync_ = flag("SGMSE_DEBUG_SYNC", false);       // synchronise after every launch of the forward and print its label (stderr)
    prof_dump_ = flag("SGMSE_PROFILE_DUMP", false);      // per-launch lines from profile_forward
    // ---- test hooks
    e = getenv("SGMSE_TILE_MIN_BLOCKS");
    tile_min_blocks_ = e ? atol(e) : 512L;               // workgroups below which a launch takes the smaller tile shape (profiles/r01_tile_sweep.txt)
    e = getenv("SGMSE_SPLIT_MIN_TILES");
    split_min_tiles_ = e ? atol(e) : 8L;                 // per-image 8x32 tiles from which a layer uses a split kernel (profiles/r01_b3_threshold.txt)
    e = getenv("SGMSE_WINO_MIN_TILES");
    wino_min_tiles_ = e ? atol(e) : 32L;                 // ... and the Winograd kernel (32: the 64 x 128 level and up)
    fuse_gn_stats_ = flag("SGMSE_FUSE_GN_STATS", true);  // GroupNorm partial sums in the conv epilogue (0: stand-alone statistics passes)
    lds_poison_ = flag("SGMSE_POISON_LDS", false);       // test hook: NaN patterns in every CU's LDS in front of every launch (lds_poison_kernel)
    e = getenv("SGMSE_POISON_LDS_AT");
    lds_poison_at_ = e ? atoi(e) : -1;                   // ... in front of launch number n only (bisection); SGMSE_POISON_LDS_UPTO=1: launches 0..n
    lds_poison_upto_ = flag("SGMSE_POISON_LDS_UPTO", false);
    side_stream_on_ = flag("SGMSE_SIDE_STREAM", true);   // output-pyramid branches on the second stream (run_forward) ...
    e = getenv("SGMSE_SIDE_MAX_BATCH");
    side_max_batch_ = e ? atoi(e) : 8;                   // ... for batches up to this size (batch 1: -1.4 %, batch 4 / 8: -0.8 %; nothing at 32: profiles/r06_side_stream.txt)
    e = getenv("SGMSE_NOFOLD_LEVELS");
    nofold_levels_ = e ? atoi(e) : 0;                    // measurement (round 6): bit l = the 1x1 shortcuts of U-Net level l as their own launch
    poison_ = flag("SGMSE_POISON", false);               // NaN patterns in every allocation and in the arena before every forward
  }
  // per-forward range-bound slots ([B][kAmaxSpread] floats each), handed out in program order; counted by the dry run
  float* amax_pool_ = nullptr; size_t amax_pool_floats_ = 0; int amax_slots_ = 0, amax_next_ = 0;
  float* next_amax() {
    const int i = amax_next_++;
    // dry run: a fake address (never dereferenced), non-null so that the kernel-family decisions that ask "is the bound known?"
    // come out as in the real run and the arena is sized for the launches that will really happen
    if (dry_) return reinterpret_cast<float*>(uintptr_t(1) << 43) + (size_t)i * kAmaxSpread;
    SG_REQUIRE(amax_pool_ && i < amax_slots_, "range-bound pool smaller than the forward needs");
    return amax_pool_ + (size_t)i * B_ * kAmaxSpread;
  }
  long tile_min_blocks_ = 512, split_min_tiles_ = 8;
  // (settled in rounds 2-3, switches removed in round 4: chunked accumulation of the coarse fp32 layers, shortcuts folded into the 3x3
  //  launches, chunked 4-row fp16x2 kernel on the 16 x 32 / 32 x 64 levels, entry convolution on the MFMA kernel)
  static constexpr bool coarse_chunked_ = true, fold_shortcut_ = true, coarse_split_ = true, entry_mfma_ = true;
  ConvW entry8_{}; int entry8_idx_ = -1;
  bool poison_ = false, debug_sync_ = false, conv_xcd_map_ = true, rag_prefix_ = true, wino_ = true;
  long wino_min_tiles_ = 32;
  int nofold_levels_ = 0;
  bool lds_poison_ = false, lds_poison_upto_ = false; int lds_poison_at_ = -1, lds_poison_count_ = 0; unsigned* lds_sink_ = nullptr;
  static constexpr long coarse_splitk_div_ = 4, chunk_max_tiles_ = 8;      // (profiles/r02_chunk_splitk.txt)
  int split_mode_ = SGMSE_CONV_SPLIT_DEFAULT;
  bool prof_dump_ = false;
  char prof_note_[160] = {0};
  static constexpr bool fir_scalar_ = false;      // (per-pixel FIR kernels everywhere: a round-1 measurement switch)
  bool fuse_gn_stats_ = true;
  int B_ = 0, shape_B_ = 0, shape_F_ = 0, shape_T_ = 0;
  float2 *sx_ = nullptr, *sxm_ = nullptr, *sscore_ = nullptr, *sy_ = nullptr; size_t samp_n_ = 0;
  int* step_ctr_ = nullptr;
  float *lang_partial_ = nullptr, *lang_scal_ = nullptr;
  float* bias_cur_ = nullptr;
  float *temb_act_ = nullptr, *bias_table_ = nullptr, *step_table_ = nullptr, *tsteps_ = nullptr, *coef_table_ = nullptr; int temb_rows_ = 0;
  drt::graph_t graph_{}; bool graph_valid_ = false; GraphKey graph_key_{};
  int nfe_ = 0;
  bool prof_ = false; std::vector<ProfRec> prof_recs_; size_t prof_used_ = 0; float prof_ms_[TC_COUNT]; double prof_flops_[TC_COUNT]; int prof_n_[TC_COUNT];
};

}  // namespace sgmse

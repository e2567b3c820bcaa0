// C-ABI (include/sgmse_hip.h) over the Engine.  Exceptions never cross the boundary.
#pragma once
#include <string>
#include "engine.h"
#include "../../include/sgmse_hip.h"

struct sgmse_ctx {
  sgmse::Engine* eng = nullptr;
  std::string err;
};

namespace {
template <class Fn>
int sg_guard(sgmse_ctx* ctx, Fn&& fn) {
  if (!ctx || !ctx->eng) return SGMSE_EINVAL;
  try {
    fn(*ctx->eng);
    return SGMSE_OK;
  } catch (const sgmse::EngineError& e) {
    ctx->err = e.what();
    const bool notready = ctx->err.find("weights not loaded") != std::string::npos;
    return notready ? SGMSE_ENOTREADY : SGMSE_ERUNTIME;
  } catch (const std::exception& e) {
    ctx->err = std::string("internal error: ") + e.what();
    return SGMSE_ERUNTIME;
  }
}
#define SG_ARG(ctx, cond, msg) do { if (!(cond)) { if (ctx) (ctx)->err = std::string("invalid argument: ") + msg; return SGMSE_EINVAL; } } while (0)
}  // namespace

extern "C" {

int sgmse_ctx_create(int device, void* hip_stream, sgmse_ctx** out) {
  if (!out) return SGMSE_EINVAL;
  *out = nullptr;
  sgmse_ctx* c = new sgmse_ctx();
  try {
    c->eng = new sgmse::Engine(device, hip_stream);
  } catch (const std::exception& e) {
    fprintf(stderr, "sgmse_ctx_create: %s\n", e.what());
    delete c;
    return SGMSE_ERUNTIME;
  }
  *out = c;
  return SGMSE_OK;
}

void sgmse_ctx_destroy(sgmse_ctx* ctx) {
  if (!ctx) return;
  delete ctx->eng;
  delete ctx;
}

int sgmse_set_stream(sgmse_ctx* ctx, void* s) { return sg_guard(ctx, [&](sgmse::Engine& e) { e.set_stream(s); }); }
int sgmse_sync(sgmse_ctx* ctx) { return sg_guard(ctx, [&](sgmse::Engine& e) { e.sync(); }); }
const char* sgmse_last_error(sgmse_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
const char* sgmse_backend(void) { return drt::backend_name(); }

int sgmse_configure(sgmse_ctx* ctx, const sgmse_net_cfg* cfg) {
  SG_ARG(ctx, cfg != nullptr, "cfg is null");
  SG_ARG(ctx, cfg->variant == 0 || cfg->variant == 1, "variant must be 0 (ncsnpp) or 1 (ncsnpp_48k)");
  SG_ARG(ctx, cfg->nf >= 32 && cfg->nf % 32 == 0 && cfg->nf <= 256, "nf must be a multiple of 32 in [32, 256]");
  SG_ARG(ctx, cfg->n_levels >= 1 && cfg->n_levels <= 8, "n_levels out of range");
  SG_ARG(ctx, cfg->num_res_blocks >= 1 && cfg->n_attn >= 0 && cfg->n_attn <= 8, "bad block counts");
  SG_ARG(ctx, cfg->progressive == 0 || cfg->progressive == 1, "progressive must be none or output_skip");
  SG_ARG(ctx, cfg->progressive_input == 0 || cfg->progressive_input == 1, "progressive_input must be none or input_skip");
  for (int i = 0; i < cfg->n_levels; ++i) SG_ARG(ctx, cfg->ch_mult[i] >= 1 && cfg->nf * cfg->ch_mult[i] <= 256, "ch_mult out of range");
  return sg_guard(ctx, [&](sgmse::Engine& e) {
    sgmse::NetCfg n;
    n.variant = cfg->variant; n.nf = cfg->nf; n.n_levels = cfg->n_levels; n.num_res_blocks = cfg->num_res_blocks;
    n.n_attn = cfg->n_attn; n.image_size = cfg->image_size; n.progressive = cfg->progressive;
    n.progressive_input = cfg->progressive_input; n.scale_by_sigma = cfg->scale_by_sigma;
    for (int i = 0; i < 8; ++i) { n.ch_mult[i] = cfg->ch_mult[i]; n.attn_res[i] = cfg->attn_res[i]; }
    e.set_config(n);
  });
}

int sgmse_load_weights(sgmse_ctx* ctx, const char* const* names, const void* const* ptrs, const long long* numels, int n,
                       int on_device) {
  SG_ARG(ctx, names && ptrs && numels && n > 0, "null table");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.load_weights(names, ptrs, numels, n, on_device); });
}

int sgmse_param_count(sgmse_ctx* ctx, long long* out) {
  SG_ARG(ctx, out != nullptr, "out is null");
  return sg_guard(ctx, [&](sgmse::Engine& e) { *out = (long long)e.param_count(); });
}

int sgmse_ncsnpp_forward(sgmse_ctx* ctx, const void* xy, const float* t, void* out, int B, int F, int T) {
  SG_ARG(ctx, xy && t && out && B > 0 && F > 0 && T > 0, "null pointer or non-positive shape");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.forward_xy((const float2*)xy, t, (float2*)out, B, F, T); });
}

int sgmse_pc_sample(sgmse_ctx* ctx, const void* Y, void* out, int B, int F, int T, const sgmse_sampler_cfg* cfg,
                    const void* noise, unsigned long long seed, int* nfe) {
  SG_ARG(ctx, Y && out && cfg && B > 0 && F > 0 && T > 0, "null pointer or non-positive shape");
  SG_ARG(ctx, cfg->N >= 1, "N must be >= 1");
  SG_ARG(ctx, cfg->corrector >= 0 && cfg->corrector <= 2, "corrector must be none, ald or langevin");
  SG_ARG(ctx, cfg->predictor == 0 || cfg->predictor == 1, "predictor must be none or reverse_diffusion");
  SG_ARG(ctx, cfg->corrector == 0 || cfg->corrector_steps >= 1, "corrector_steps must be >= 1");
  return sg_guard(ctx, [&](sgmse::Engine& e) {
    sgmse::SamplerCfg s;
    s.N = cfg->N; s.corrector = cfg->corrector; s.corrector_steps = cfg->corrector_steps; s.predictor = cfg->predictor;
    s.probability_flow = cfg->probability_flow; s.denoise = cfg->denoise; s.theta = cfg->theta; s.std1 = cfg->std1;
    s.t = cfg->t; s.dt = cfg->dt; s.ald_eps = cfg->ald_eps; s.ald_noise = cfg->ald_noise; s.G = cfg->G; s.G2 = cfg->G2;
    s.snr = cfg->snr;
    s.in_scale = cfg->in_scale; s.score_alpha = cfg->score_alpha; s.score_beta = cfg->score_beta;
    s.use_graph = cfg->use_graph;
    e.pc_sample((const float2*)Y, (float2*)out, B, F, T, s, (const float2*)noise, seed);
    if (nfe) *nfe = e.last_nfe();
  });
}

int sgmse_sb_sample(sgmse_ctx* ctx, const void* Y, void* out, int B, int F, int T, int N, const float* t, const float* w_prev,
                    const float* w_est, const float* w_y, const float* w_z, const float* in_scale, const float* score_alpha,
                    const float* score_beta, int stochastic, const void* noise, unsigned long long seed, int use_graph, int* nfe) {
  SG_ARG(ctx, Y && out && B > 0 && F > 0 && T > 0 && N >= 1, "null pointer or non-positive shape");
  SG_ARG(ctx, t && w_prev && w_est && w_y && w_z, "step weights missing");
  return sg_guard(ctx, [&](sgmse::Engine& e) {
    e.sb_sample((const float2*)Y, (float2*)out, B, F, T, N, t, w_prev, w_est, w_y, w_z, in_scale, score_alpha, score_beta, stochastic,
                (const float2*)noise, seed, use_graph);
    if (nfe) *nfe = e.last_nfe();
  });
}

int sgmse_stft(sgmse_ctx* ctx, const float* sig, const float* window, void* spec, int B, int L, int n_fft, int hop) {
  SG_ARG(ctx, sig && window && spec && B > 0 && L > 0, "null pointer or non-positive shape");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_stft(sig, window, (float2*)spec, B, L, n_fft, hop); });
}

int sgmse_istft(sgmse_ctx* ctx, const void* spec, const float* window, float* out, int B, int K, int n_fft, int hop, int length) {
  SG_ARG(ctx, spec && window && out && B > 0 && K > 0, "null pointer or non-positive shape");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_istft((const float2*)spec, window, out, B, K, n_fft, hop, length); });
}

int sgmse_spec_fwd(sgmse_ctx* ctx, const void* in, void* out, long long n, int type, float factor, float exponent) {
  SG_ARG(ctx, in && out && n > 0 && type >= 0 && type <= 2, "bad arguments");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_spec_xform((const float2*)in, (float2*)out, (size_t)n, type, factor, exponent, 0); });
}

int sgmse_spec_back(sgmse_ctx* ctx, const void* in, void* out, long long n, int type, float factor, float exponent) {
  SG_ARG(ctx, in && out && n > 0 && type >= 0 && type <= 2, "bad arguments");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_spec_xform((const float2*)in, (float2*)out, (size_t)n, type, factor, exponent, 1); });
}

int sgmse_upfirdn2d(sgmse_ctx* ctx, const float* input, const float* kernel, float* out, int BC, int H, int W, int kh, int kw,
                    int up_x, int up_y, int down_x, int down_y, int px0, int px1, int py0, int py1) {
  SG_ARG(ctx, input && kernel && out && BC > 0 && H > 0 && W > 0 && kh > 0 && kw > 0, "null pointer or non-positive shape");
  SG_ARG(ctx, up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "up/down factors must be >= 1");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_upfirdn2d(0, input, kernel, out, BC, H, W, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1); });
}

int sgmse_upfirdn2d_dtype(sgmse_ctx* ctx, int dtype, const void* input, const void* kernel, void* out, int BC, int H, int W, int kh, int kw,
                          int up_x, int up_y, int down_x, int down_y, int px0, int px1, int py0, int py1) {
  SG_ARG(ctx, dtype >= 0 && dtype <= 2, "dtype must be 0 (float), 1 (double) or 2 (half)");
  SG_ARG(ctx, input && kernel && out && BC > 0 && H > 0 && W > 0 && kh > 0 && kw > 0, "null pointer or non-positive shape");
  SG_ARG(ctx, up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1, "up/down factors must be >= 1");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_upfirdn2d(dtype, input, kernel, out, BC, H, W, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1); });
}

int sgmse_op_conv2d(sgmse_ctx* ctx, const float* x, const float* w, const float* bias, const float* res, float* out, int B, int Cin,
                    int Cout, int H, int W, int ks, float out_scale, int force_direct, const float* in_scale,
                    const float* in_shift, int in_act, const float* x2, int C2) {
  SG_ARG(ctx, x && w && out && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "null pointer or non-positive shape");
  SG_ARG(ctx, ks == 1 || ks == 3, "kernel size must be 1 or 3");
  SG_ARG(ctx, (x2 == nullptr) == (C2 == 0) && C2 >= 0 && C2 < Cin, "bad second source");
  return sg_guard(ctx, [&](sgmse::Engine& e) {
    e.op_conv2d(x, w, bias, res, out, B, Cin, Cout, H, W, ks, out_scale, force_direct, in_scale, in_shift, in_act, x2, C2);
  });
}

int sgmse_op_groupnorm(sgmse_ctx* ctx, const float* x, const float* gamma, const float* beta, float* out, int B, int C, int H,
                       int W, int act, const float* x2, int C2) {
  SG_ARG(ctx, x && gamma && beta && out && B > 0 && C >= 4 && H > 0 && W > 0, "null pointer or non-positive shape");
  SG_ARG(ctx, C % (C / 4 < 32 ? C / 4 : 32) == 0, "C must be divisible by min(C/4, 32)");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_groupnorm(x, gamma, beta, out, B, C, H, W, act, x2, C2); });
}

int sgmse_conv_split_mode(sgmse_ctx* ctx, int* out) {
  SG_ARG(ctx, out != nullptr, "out is null");
  return sg_guard(ctx, [&](sgmse::Engine& e) { *out = e.split_mode(); });
}

int sgmse_conv_winograd(sgmse_ctx* ctx, int* out) {
  SG_ARG(ctx, out != nullptr, "out is null");
  return sg_guard(ctx, [&](sgmse::Engine& e) { *out = e.winograd() ? 1 : 0; });
}

int sgmse_op_fir(sgmse_ctx* ctx, const float* x, float* out, int BC, int H, int W, int up, const float* in_scale,
                 const float* in_shift, int in_act, float* out_raw) {
  SG_ARG(ctx, x && out && BC > 0 && H > 0 && W > 0, "null pointer or non-positive shape");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_fir(x, out, BC, H, W, up, in_scale, in_shift, in_act, out_raw); });
}

int sgmse_op_attention(sgmse_ctx* ctx, const float* qkv, float* out, int B, int C, int S) {
  SG_ARG(ctx, qkv && out && B > 0 && C > 0 && S > 0, "null pointer or non-positive shape");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.op_attention(qkv, out, B, C, S); });
}

int sgmse_profile_forward(sgmse_ctx* ctx, const void* xy, const float* t, void* out, int B, int F, int T, float* ms, double* work,
                          int* launches) {
  SG_ARG(ctx, xy && t && out && ms && work && launches, "null pointer");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.profile_forward((const float2*)xy, t, (float2*)out, B, F, T, ms, work, launches); });
}

int sgmse_bench_conv(sgmse_ctx* ctx, int ks, int B, int Cin, int Cout, int H, int W, int variant, int iters, int fused, float* ms) {
  SG_ARG(ctx, ms && iters > 0 && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (ks == 1 || ks == 3), "bad arguments");
  return sg_guard(ctx, [&](sgmse::Engine& e) { *ms = e.bench_conv(ks, B, Cin, Cout, H, W, variant, iters, fused); });
}

int sgmse_calib_stream(sgmse_ctx* ctx, int mode, int bytes_per_lane, long long total_bytes, float* ms) {
  SG_ARG(ctx, ms != nullptr && total_bytes > 0, "bad arguments");
  return sg_guard(ctx, [&](sgmse::Engine& e) { *ms = e.calib_stream(mode, bytes_per_lane, (size_t)total_bytes); });
}

int sgmse_arena_bytes(sgmse_ctx* ctx, long long* out) {
  SG_ARG(ctx, out != nullptr, "out is null");
  return sg_guard(ctx, [&](sgmse::Engine& e) { *out = (long long)e.arena_bytes(); });
}

int sgmse_set_noise_streams(sgmse_ctx* ctx, const unsigned long long* ids, int n) {
  SG_ARG(ctx, (ids != nullptr && n > 0) || n == 0, "bad stream table");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.set_noise_streams(ids, n); });
}

int sgmse_set_frames(sgmse_ctx* ctx, const int* frames, int n) {
  SG_ARG(ctx, (frames != nullptr && n > 0) || n == 0, "bad frame table");
  return sg_guard(ctx, [&](sgmse::Engine& e) { e.set_ragged_frames(frames, n); });
}

int sgmse_graph_captures(sgmse_ctx* ctx, int* out) {
  SG_ARG(ctx, out != nullptr, "out is null");
  return sg_guard(ctx, [&](sgmse::Engine& e) { *out = e.graph_captures(); });
}

int sgmse_graph_updates(sgmse_ctx* ctx, int* out) {
  SG_ARG(ctx, out != nullptr, "out is null");
  return sg_guard(ctx, [&](sgmse::Engine& e) { *out = e.graph_updates(); });
}

}  // extern "C"

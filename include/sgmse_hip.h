/* sgmse_hip.h — C ABI of libsgmse_hip.so, the MI355X (gfx950) implementation of the reverse-SDE enhancement
 * sampler of sp-uhh/sgmse.
 *
 * The reference has no C ABI of its own; its native boundary is one pybind11 function
 * (sgmse/backbones/ncsnpp_utils/op/upfirdn2d.cpp:12-23) and everything else on the hot path is torch calls from
 * Python.  Each entry point below names the reference interface it replaces (paths relative to the upstream tree).
 *
 * Conventions
 *   - all pointers are caller-owned DEVICE pointers unless the parameter comment says "host";
 *   - complex tensors are interleaved (re, im) fp32 pairs == torch.complex64 storage;
 *   - tensors are dense and contiguous in the index order written in the comment;
 *   - work is enqueued on the context's stream (the caller's current HIP stream); only the functions marked
 *     "synchronises" wait for it;
 *   - return 0 on success, a negative sgmse_status otherwise; sgmse_last_error(ctx) gives the message.  The Python
 *     host layer re-raises SGMSE_EINVAL as ValueError and the rest as RuntimeError (reference: TORCH_CHECK ->
 *     RuntimeError, upfirdn2d.cpp:8-16; registry ValueError, util/registry.py:30).
 *   - a context is not re-entrant; use one per (device, stream).
 */
#ifndef SGMSE_HIP_H
#define SGMSE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgmse_ctx sgmse_ctx;

typedef enum { SGMSE_OK = 0, SGMSE_EINVAL = -1, SGMSE_ERUNTIME = -2, SGMSE_ENOTREADY = -3 } sgmse_status;

/* Constructor arguments of NCSNpp that change the graph (sgmse/backbones/ncsnpp.py:50-72,
 * ncsnpp_48k.py:50-72).  Fixed by this implementation, as in every shipped checkpoint: resblock_type='biggan',
 * fir=True, fir_kernel=[1,3,3,1], skip_rescale=True, resamp_with_conv=True (unused with biggan blocks),
 * combine_method='sum', embedding_type='fourier', conditional=True, centered=True, nonlinearity='swish',
 * init_scale=0, dropout=0. */
typedef struct {
  int variant;            /* 0: "ncsnpp" (ncsnpp.py) and, with scale_by_sigma = 0, "ncsnpp_v2" (ncsnpp_v2.py: same network, no /t);
                             1: "ncsnpp_48k" (ncsnpp_48k.py: output_layer before /t) */
  int nf;                 /* base width (128) */
  int n_levels;           /* len(ch_mult) (7) */
  int ch_mult[8];         /* (1,1,2,2,2,2,2) */
  int num_res_blocks;     /* 2 */
  int n_attn;             /* len(attn_resolutions) */
  int attn_res[8];        /* (16,) ; () for ncsnpp_48k */
  int image_size;         /* 256 */
  int progressive;        /* 0 'none', 1 'output_skip' */
  int progressive_input;  /* 0 'none', 1 'input_skip' */
  int scale_by_sigma;     /* 1 */
} sgmse_net_cfg;

/* One call of get_pc_sampler(...)() (sgmse/sampling/__init__.py:26-70).  The per-step scalars are HOST arrays of
 * length N computed by the caller with the reference's own fp32 expressions (sgmse/sdes.py:188-219,
 * sampling/__init__.py:56-62, correctors.py:77-79): t = linspace(T, eps, N); dt[i] = t[i]-t[i+1], dt[N-1] = t[N-1];
 * ald_eps = 2 (snr std(t))^2; ald_noise = sqrt(2 ald_eps); G = g(t) sqrt(dt); G2 = G^2. */
typedef struct {
  int N;
  int corrector;          /* 0 'none' (correctors.py:85-94), 1 'ald' (correctors.py:60-81), 2 'langevin' (correctors.py:37-56:
                             step size from batch-mean norms of score and noise) */
  int corrector_steps;
  int predictor;          /* 0 'none', 1 'reverse_diffusion' (predictors.py:56-65) */
  int probability_flow;   /* 1: fixed-step PF-ODE Euler step (sdes.py:130-135 with probability_flow=True), no noise */
  int denoise;            /* 1: return x_mean of the last step (sampling/__init__.py:66) */
  float theta;            /* OUVESDE.theta */
  float std1;             /* OUVESDE._std(T=1) for prior_sampling (sdes.py:224-229) */
  const float* t; const float* dt; const float* ald_eps; const float* ald_noise; const float* G; const float* G2;
  /* ScoreModel.forward as an affine wrapper of the backbone output F (model.py:284-310): the network sees in_scale*x_t and
   * in_scale*y, score = score_alpha*x_t + score_beta*F.  Host arrays [N], all three or all NULL.  NULL = old-code branch
   * (model.py:307-310): in_scale 1, alpha 0, beta -1.  ncsnpp_v2 models: in_scale = c_in(t); 'score_matching': alpha =
   * c_skip(t), beta = c_out(t)*s(t); 'denoiser': alpha = -1/std(t)^2, beta = s(t)/std(t)^2; s = network_scaling. */
  const float* in_scale; const float* score_alpha; const float* score_beta;
  float snr;              /* 'langevin' only: r in step = 2 (r * mean||z|| / mean||score||)^2 ('ald' has it folded into ald_eps) */
  int use_graph;          /* 1: capture one predictor-corrector step as a hipGraph and replay it N times */
} sgmse_sampler_cfg;

/* -- context ------------------------------------------------------------------------------------------------ */
int sgmse_ctx_create(int device, void* hip_stream, sgmse_ctx** out);
void sgmse_ctx_destroy(sgmse_ctx* ctx);
int sgmse_set_stream(sgmse_ctx* ctx, void* hip_stream);
int sgmse_sync(sgmse_ctx* ctx);                      /* synchronises */
const char* sgmse_last_error(sgmse_ctx* ctx);
const char* sgmse_backend(void);                     /* "hip-gfx950" for the product library */

/* -- model: replaces BackboneRegistry.get_by_name(name)(**kwargs) + load_state_dict (model.py:60-61,100-106) -- */
int sgmse_configure(sgmse_ctx* ctx, const sgmse_net_cfg* cfg);
/* names = the reference state_dict keys of the backbone ("output_layer.weight", "all_modules.3.weight", ...;
 * ncsnpp.py:105,253), fp32, shapes as in the reference; every key of the configured network must be present.
 * ptrs are host pointers (on_device = 0; synchronises) or device pointers (on_device = 1, e.g. the buffer an RCCL
 * broadcast just filled). */
int sgmse_load_weights(sgmse_ctx* ctx, const char* const* names, const void* const* ptrs, const long long* numels,
                       int n, int on_device);
int sgmse_param_count(sgmse_ctx* ctx, long long* out);

/* -- NCSNpp.forward(x, time_cond) (ncsnpp.py:256-419; ncsnpp_48k.py:260-424) ---------------------------------
 * xy: complex64 [B][2][F][T] (channel 0 = x_t, 1 = y); t: fp32 [B]; out: complex64 [B][1][F][T]. */
int sgmse_ncsnpp_forward(sgmse_ctx* ctx, const void* xy, const float* t, void* out, int B, int F, int T);

/* -- get_pc_sampler(...)() (sampling/__init__.py:52-68) with score_fn = ScoreModel.forward (model.py:307-310) --
 * Y, out: complex64 [B][1][F][T].  noise: complex64 [1 + N*draws_per_step][B][1][F][T] standard-normal draws in the
 * reference's call order (prior, then per step: corrector draws, predictor draw), or NULL for the in-kernel Philox
 * stream seeded by `seed`.  *nfe receives N*(corrector_steps+1). */
int sgmse_pc_sample(sgmse_ctx* ctx, const void* Y, void* out, int B, int F, int T, const sgmse_sampler_cfg* cfg,
                    const void* noise, unsigned long long seed, int* nfe);

/* -- get_sb_sampler(sde, model, y, ...)() (sampling/__init__.py:145-249), Schroedinger-bridge 'ode' and 'sde' samplers:
 *    x_0 = y; for i in 0..N-1:  x <- w_prev[i] x + w_est[i] model(x, y, t[i]) + w_y[i] y + w_z[i] z_i.  Host arrays [N] computed
 *    by the caller from SBVESDE._sigmas_alphas (sdes.py:276-288) with the reference's expressions (t = linspace(T, eps, N+1)[1:]);
 *    'ode': w_z = 0; 'sde' (stochastic = 1): w_y = 0, one draw per step, noise complex64 [N][B][1][F][T] or NULL (Philox).
 *    model = ScoreModel.forward with loss_type 'data_prediction': in_scale / score_alpha / score_beta as in sgmse_sampler_cfg. */
int sgmse_sb_sample(sgmse_ctx* ctx, const void* Y, void* out, int B, int F, int T, int N, const float* t, const float* w_prev,
                    const float* w_est, const float* w_y, const float* w_z, const float* in_scale, const float* score_alpha,
                    const float* score_beta, int stochastic, const void* noise, unsigned long long seed, int use_graph, int* nfe);

/* -- SpecsDataModule.stft / istft / spec_fwd / spec_back (data_module.py:162-188,212-218) ---------------------
 * sig fp32 [B][L] -> spec complex64 [B][n_fft/2+1][L/hop+1] (center=True, reflect pad, window fp32 [n_fft]). */
int sgmse_stft(sgmse_ctx* ctx, const float* sig, const float* window, void* spec, int B, int L, int n_fft, int hop);
int sgmse_istft(sgmse_ctx* ctx, const void* spec, const float* window, float* out, int B, int K, int n_fft, int hop,
                int length);
/* transform_type: 0 'exponent', 1 'log', 2 'none'; n = number of complex elements; in may equal out */
int sgmse_spec_fwd(sgmse_ctx* ctx, const void* in, void* out, long long n, int transform_type, float factor, float exponent);
int sgmse_spec_back(sgmse_ctx* ctx, const void* in, void* out, long long n, int transform_type, float factor, float exponent);

/* -- upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
 *    (op/upfirdn2d.cpp:12-23, op/upfirdn2d_kernel.cu:209-369).  input fp32 [BC][H][W], kernel fp32 [kh][kw],
 *    out fp32 [BC][(H*up_y+pad_y0+pad_y1-kh)/down_y+1][(W*up_x+pad_x0+pad_x1-kw)/down_x+1]. */
int sgmse_upfirdn2d(sgmse_ctx* ctx, const float* input, const float* kernel, float* out, int BC, int H, int W, int kh,
                    int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1);
/*    The same op over the other element types the reference dispatches (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 *    op/upfirdn2d_kernel.cu:311): dtype 0 = float (as above), 1 = double, 2 = half (IEEE binary16 bits; fp32 accumulation, one
 *    rounding on the way out).  input, kernel and out share the element type. */
int sgmse_upfirdn2d_dtype(sgmse_ctx* ctx, int dtype, const void* input, const void* kernel, void* out, int BC, int H, int W,
                          int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                          int pad_y1);

/* -- single layers, exported for per-op parity tests (all NCHW fp32) --------------------------------------------
 * conv2d: F.conv2d(cat[x, x2], w, bias, padding=ks/2) (layers.py:100-124), optionally with a fused per-(b,c)
 * affine(+SiLU) on the input, a fused residual and output scale: out = (conv + bias + res) * out_scale.
 * x holds Cin-C2 channels, x2 (may be NULL) the remaining C2.  force_direct = 1 selects the VALU kernel,
 * 2 / 3 the bf16x3 / fp16x2 split MFMA kernels (3x3, Cout % 128 == 0, channel counts % 16 == 0; error otherwise). */
int sgmse_op_conv2d(sgmse_ctx* ctx, const float* x, const float* w_oihw, const float* bias, const float* res, float* out,
                    int B, int Cin, int Cout, int H, int W, int ks, float out_scale, int force_direct,
                    const float* in_scale, const float* in_shift, int in_act, const float* x2, int C2);
/* act(GroupNorm(min(C/4,32), C, eps=1e-6)(cat[x, x2])) (layerspp.py:219,243); act: 0 none, 1 SiLU.  synchronises */
int sgmse_op_groupnorm(sgmse_ctx* ctx, const float* x, const float* gamma, const float* beta, float* out, int B, int C,
                       int H, int W, int act, const float* x2, int C2);
/* upsample_2d / downsample_2d with k=(1,3,3,1), factor 2 (up_or_down_sampling.py:195-257); x fp32 [BC][H][W].
 * Optional fused producer as inside the network: out = FIR(act(x * in_scale[bc] + in_shift[bc])) (in_scale NULL: none;
 * in_act 1: SiLU), and with out_raw != NULL also out_raw = FIR(x) from the same pass (layerspp.py:243-262). */
int sgmse_op_fir(sgmse_ctx* ctx, const float* x, float* out, int BC, int H, int W, int up, const float* in_scale,
                 const float* in_shift, int in_act, float* out_raw);
/* attention core of AttnBlockpp (layerspp.py:82-88): qkv fp32 [B][3C][S] -> out fp32 [B][C][S] */
int sgmse_op_attention(sgmse_ctx* ctx, const float* qkv, float* out, int B, int C, int S);

/* which kernel family the wide 3x3 layers use in this context: 0 fp32 MFMA, 1 bf16x3 split, 2 fp16x2 split
 * (SGMSE_CONV_SPLIT, read at configure time; kernels_conv_split.h) */
int sgmse_conv_split_mode(sgmse_ctx* ctx, int* out);
/* 1 when the levels of >= 32 tiles per image run their full 3x3 layers on the Winograd F(2,3) x fp16x2 kernel (split mode 2 and
 * SGMSE_WINO != 0; kernels_conv_wino.h), else 0.  Like the split mode it replaces nothing in the reference (F.conv2d picks its own
 * algorithm, layers.py:118-124); bench.py names the dominant kernel by it. */
int sgmse_conv_winograd(sgmse_ctx* ctx, int* out);

/* -- measurement: one eager forward with HIP events (on the context's stream) around every kernel launch, summed per
 *    kernel class.  ms, work and launches are host arrays of SGMSE_NCLASS entries:
 *      0 conv3x3 MFMA, 128-channel x 256-pixel tile (the dominant kernel)   1 conv3x3 MFMA, other tiles
 *      2 conv1x1 MFMA   3 direct (VALU) conv   4 groupnorm statistics   5 FIR resampling   6 attention core   7 entry/exit
 *    work = algorithmic FLOPs (classes 0-3, 6) or algorithmic bytes (classes 4, 5, 7).  synchronises */
#define SGMSE_NCLASS 8
int sgmse_profile_forward(sgmse_ctx* ctx, const void* xy, const float* t, void* out, int B, int F, int T, float* ms,
                          double* work, int* launches);
/* micro-benchmark of one MFMA convolution shape on random operands: ms per launch over `iters` back-to-back launches
 * (HIP events on the context's stream).  variant: kernel-variant id (see kernels_conv.h), -1 = default.
 * fused = 1 adds the GroupNorm-affine+SiLU producer and the bias/residual/scale epilogue.  synchronises */
int sgmse_bench_conv(sgmse_ctx* ctx, int ks, int B, int Cin, int Cout, int H, int W, int variant, int iters, int fused,
                     float* ms);
/* measurement only: ONE launch of a coalesced stream of exactly total_bytes (mode 0: read, 1: write; bytes_per_lane 8 or 16) on the
 * context's stream, so that the FETCH_SIZE / WRITE_SIZE counters of a rocprofv3 --pmc pass can be calibrated against a known byte
 * count in the pass that reads them (tools/gpu_visit.sh STEPS=pmcbench; nothing in the reference corresponds).  synchronises */
int sgmse_calib_stream(sgmse_ctx* ctx, int mode, int bytes_per_lane, long long total_bytes, float* ms);
int sgmse_arena_bytes(sgmse_ctx* ctx, long long* out);
/* Noise-stream ids (host array, one per utterance) for the NEXT sgmse_pc_sample / sgmse_sb_sample call of batch size n with
 * in-kernel (Philox) noise: an utterance's draws depend on (seed, stream id, element index inside the utterance, draw index)
 * only, so with ids that name the utterance its noise does not depend on batch composition or slot.  Default (no call, or a
 * different batch size): utterance b uses stream b.  Replaces nothing in the reference (torch.randn_like draws depend on the
 * global generator state, sdes.py:229, correctors.py:74, predictors.py:62). */
int sgmse_set_noise_streams(sgmse_ctx* ctx, const unsigned long long* ids, int n);
/* Ragged batches: frames[b] = number of spectrogram frames T_b of utterance b (host array; each a multiple of 2^(levels-1), i.e.
 * of 64 -- what pad_spec produces) for ALL FOLLOWING sgmse_ncsnpp_forward / sgmse_pc_sample calls of batch size n, until the next
 * call (n = 0: uniform batches again).  Those calls then take T = max_b T_b and PACKED tensors: utterance after utterance, each
 * with its own row stride -- xy: [2][F][T_b] per utterance, y / the results: [F][T_b] per utterance (sum_b F*T_b complex elements).
 * An utterance's result is bit-identical to the one its own single-utterance call gives (every kernel addresses an utterance's
 * block exactly as in that call; which kernel family a layer uses depends on the U-Net level only, never on T), so a corpus of
 * mixed lengths runs in full batches.  In-kernel noise only (noise == NULL), correctors "ald" / "none"; the Langevin corrector
 * couples the utterances of a batch (correctors.py:50-52) and the Schroedinger-bridge sampler are not supported.
 * Replaces the one-file-at-a-time loop of enhancement.py:57-103. */
int sgmse_set_frames(sgmse_ctx* ctx, const int* frames, int n);
/* how many times this context captured + instantiated a sampler step as a hipGraph (a run over many batches of one shape and
 * sampler configuration captures once: the Philox seed is device data, not a graph parameter) */
int sgmse_graph_captures(sgmse_ctx* ctx, int* out);
/* how many times a captured step was brought up to date IN PLACE (hipGraphExecUpdate: same launch sequence, other arguments --
 * a new ragged batch composition, a new batch size of the same kernel sequence) instead of being instantiated anew */
int sgmse_graph_updates(sgmse_ctx* ctx, int* out);

#ifdef __cplusplus
}
#endif
#endif /* SGMSE_HIP_H */

"""Kernel-logic parity on the CPU workgroup emulator (no GPU needed): the SAME kernel sources that hipcc builds for
gfx950, compiled with g++ against tests/emu/sgmse_devrt.h, against the oracle and the reference-made fixtures.
The `-m gpu` tests (test_gpu_parity.py) repeat these checks -- and larger ones -- through the real HIP library."""
import os

import numpy as np

import pytest
import torch

import parity as P
from oracle import stft_oracle as FO

SLOW = os.environ.get("SGMSE_SLOW", "0") == "1"


@pytest.mark.parametrize("shape", [
    (2, 32, 32, 16, 40, 3), (1, 64, 128, 9, 33, 3), (2, 32, 64, 4, 8, 3), (1, 96, 32, 8, 32, 1), (2, 64, 64, 5, 7, 1),
    (1, 64, 64, 3, 1, 3), (1, 32, 32, 1, 1, 1)])
def test_conv_mfma(emu, shape):
    P.check_conv(emu, *shape)


@pytest.mark.parametrize("shape", [(2, 4, 32, 10, 12, 3), (2, 32, 4, 10, 12, 3), (1, 4, 32, 6, 5, 1), (1, 24, 20, 7, 9, 3)])
def test_conv_direct(emu, shape):
    P.check_conv(emu, *shape, direct=True)


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 6, 8])
def test_conv_kernel_variants(emu, variant):
    """Every selectable structure of the MFMA convolution (operand prefetch on/off, float4 / element-wise staging, the
    software-pipelined kernel) computes the same convolution."""
    import subprocess, sys
    from conftest import EMU_LIB, ROOT
    env = dict(os.environ, SGMSE_CONV_VARIANT=str(variant), PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_check.py"), EMU_LIB, "cpu"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert "VARIANT-OK" in out.stdout, out.stdout + out.stderr


def test_conv_thin_output_on_padded_mfma_tile(emu):
    P.check_conv(emu, 1, 64, 4, 9, 33, 3, xform=True)
    P.check_conv(emu, 2, 128, 4, 4, 8, 3, xform=True)


def test_schroedinger_bridge_sampler_matches_reference(emu):
    P.check_sb_golden(emu, "sde", batch=1)
    P.check_sb_golden(emu, "ode", batch=1)      # (first step: 5457 y - 5456.5 y -- the kernel follows torch's rounding sequence)


def test_sampler_langevin_corrector_against_oracle(emu):
    """LangevinCorrector couples the batch through mean norms (correctors.py:50-52): B = 2."""
    P.check_sampler_oracle(emu, "ncsnpp", N=1, corrector="langevin", snr=0.5, F_=256, T=64, B=2)


def test_sampler_48k_variant_against_oracle(emu):
    P.check_sampler_oracle(emu, "ncsnpp_48k", N=1, snr=0.33, F_=192, T=64, B=1)


def test_conv3x3_bf16x3_kernel_has_fp32_accuracy(emu):
    P.check_conv_b3(emu, 1, 32, 128, 9, 33)
    P.check_conv_b3(emu, 2, 48, 128, 8, 32, xform=True)
    P.check_conv_b3(emu, 1, 64, 256, 5, 40, dual=32, xform=True)


def test_conv3x3_winograd_fp16x2_kernel_has_fp32_accuracy(emu):
    """Winograd F(2,3) x fp16x2 (kernels_conv_wino.h): odd tile counts, a width that is not a multiple of 32, two sources, raw input,
    utterances of different ranges, per-channel weight scales under outlier weights."""
    P.check_conv_wino(emu, 1, 32, 128, 9, 34)
    P.check_conv_wino(emu, 2, 48, 128, 8, 32, xmul=50.0)
    P.check_conv_wino(emu, 1, 64, 256, 5, 40, dual=32)
    P.check_conv_wino(emu, 1, 16, 128, 12, 64, xform=False, res=False)
    P.check_conv_wino(emu, 1, 32, 128, 8, 32, wmul=6)


def test_conv3x3_winograd_2d_fp16x2_kernel_has_fp32_accuracy(emu):
    """kernels_conv_wino2d.h (round 6: F(2x2,3x3) x fp16x2, measured against the 1-D kernel and not taken): the same accuracy gates."""
    P.check_conv_wino2d(emu, 1, 32, 128, 9, 34)
    P.check_conv_wino2d(emu, 2, 48, 128, 8, 32, xmul=50.0)
    P.check_conv_wino2d(emu, 1, 64, 256, 6, 40, dual=32)
    P.check_conv_wino2d(emu, 1, 16, 128, 12, 64, xform=False, res=False)
    P.check_conv_wino2d(emu, 1, 32, 128, 8, 32, wmul=6)


def test_conv3x3_thin_output_valu_kernel(emu):
    """C -> 4 pyramid convolutions on the exact-fp32 VALU kernel (kernels_conv_thin.h; the engine's path for these layers): fp32
    accuracy against float64 (no worse than 1.5x the fp32 MFMA kernel's own error), ragged tile edges, 2 output channels, dual input."""
    P.check_conv_b3(emu, 1, 64, 4, 9, 33, xform=True, split="thin", slack=1.5)
    P.check_conv_b3(emu, 2, 128, 4, 20, 70, xform=True, split="thin", slack=1.5)
    P.check_conv_b3(emu, 1, 64, 2, 17, 128, xform=False, split="thin", slack=1.5)
    P.check_conv_b3(emu, 1, 96, 4, 5, 32, dual=32, xform=True, split="thin", slack=1.5)


def test_conv3x3_thin_output_valu_kernel_is_batch_independent(emu):
    P.check_conv_thin_batch_independence(emu)


def test_conv3x3_thin_output_split_kernel(emu):
    """C -> 4 pyramid convolutions on the split kernel's thin variant (one padded 32-channel fragment, waves split pixels)."""
    P.check_conv_b3(emu, 1, 64, 4, 9, 33, xform=True, split="fp16x2", slack=3.0)
    P.check_conv_b3(emu, 2, 128, 4, 8, 40, xform=True, split="fp16x2", slack=3.0)
    P.check_conv_b3(emu, 1, 64, 4, 5, 32, xform=True, split="bf16x3")


def test_conv1x1_bf16x3_kernel_has_fp32_accuracy(emu):
    P.check_conv_b3(emu, 1, 32, 128, 9, 33, ks=1)
    P.check_conv_b3(emu, 2, 96, 256, 5, 40, ks=1, xform=True)
    P.check_conv_b3(emu, 1, 160, 128, 16, 20, ks=1, dual=64)
    P.check_conv_b3(emu, 1, 16, 128, 1, 1, ks=1)


def test_conv1x1_fp16x2_kernel_scales_by_the_input_range(emu):
    """1x1 layers read the raw residual stream: the fp16x2 kernel derives an exact power-of-two scale per utterance from
    range bounds (here computed by the op entry point, in the network left behind by the producing epilogues), so inputs
    of any magnitude -- far outside fp16's own range included -- keep fp32 accuracy."""
    P.check_conv_b3(emu, 1, 32, 128, 9, 33, ks=1, split="fp16x2", slack=3.0)
    P.check_conv_b3(emu, 2, 96, 256, 5, 40, ks=1, split="fp16x2", slack=3.0, xmul=1e6)
    P.check_conv_b3(emu, 2, 160, 128, 16, 20, ks=1, dual=64, split="fp16x2", slack=3.0, xmul=1e-6)


def test_conv3x3_fp16x2_kernel_is_within_one_bit_of_fp32(emu):
    P.check_conv_b3(emu, 1, 32, 128, 9, 33, xform=True, split="fp16x2", slack=3.0)
    P.check_conv_b3(emu, 2, 48, 128, 8, 32, xform=True, split="fp16x2", slack=3.0)
    P.check_conv_b3(emu, 1, 64, 256, 5, 40, dual=32, xform=True, split="fp16x2", slack=3.0)


def test_conv1x1_wide_output(emu):
    """1x1 convolutions with 128-channel output blocks: ragged edges, concat, fused producer (the streaming variant of
    the same shapes runs under SGMSE_CONV_VARIANT=8 in test_conv_kernel_variants)."""
    P.check_conv(emu, 1, 64, 128, 9, 33, 1)
    P.check_conv(emu, 2, 96, 256, 5, 40, 1, xform=True)
    P.check_conv(emu, 1, 160, 128, 16, 20, 1, dual=64, xform=True)
    P.check_conv(emu, 1, 32, 128, 1, 1, 1)


def test_conv_concat_and_fused_groupnorm_silu(emu):
    P.check_conv(emu, 2, 96, 32, 12, 36, 3, dual=64, xform=True)
    P.check_conv(emu, 1, 64, 32, 8, 8, 1, dual=32, xform=True)
    P.check_conv(emu, 1, 12, 4, 6, 6, 3, direct=True, dual=4, xform=True)


@pytest.mark.parametrize("shape", [(2, 32, 8, 8), (1, 96, 5, 7), (2, 128, 16, 16)])
def test_groupnorm(emu, shape):
    P.check_groupnorm(emu, *shape)
    P.check_groupnorm(emu, *shape, act=False)


def test_groupnorm_concat_group_straddles_sources(emu):
    P.check_groupnorm(emu, 2, 96, 6, 10, dual=32)     # 24 groups of 4: boundary at channel 64 is aligned
    P.check_groupnorm(emu, 1, 384, 4, 4, dual=128)    # 32 groups of 12: group 21 straddles channel 256


def test_fir(emu):
    P.check_fir(emu)
    P.check_fir(emu, 1, 2, 4, 1)
    P.check_fir(emu, 1, 2, 20, 72)
    P.check_fir_fused(emu)
    P.check_fir_fused(emu, 2, 3, 6, 12)
    P.check_fir_fused(emu, 1, 1, 16, 128)
    P.check_fir_golden(emu)


def test_upfirdn2d_refuses_what_the_reference_op_refuses(emu):
    """op/upfirdn2d_kernel.cu:311 dispatches float / double / half only, upfirdn2d.cpp:15-16 checks both tensors' device: other element types
    raise instead of running silently as float32."""
    import torch
    from sgmse_amd import ops
    x, k = torch.randn(1, 2, 8, 8), torch.ones(4, 4) / 16
    assert ops.upfirdn2d(x, k, down=2, pad=(1, 1)).dtype == torch.float32
    assert ops.upfirdn2d(x.double(), k, down=2, pad=(1, 1)).dtype == torch.float64        # (the kernel's dtype is cast to the input's)
    for bad in (x.to(torch.bfloat16), x.to(torch.int32)):
        with pytest.raises(ValueError):
            ops.upfirdn2d(bad, k, down=2, pad=(1, 1))
    with pytest.raises(ValueError):
        ops.upfirdn2d(x, k.to(torch.int64), down=2, pad=(1, 1))


@pytest.mark.parametrize("shape", [(2, 64, 40), (1, 32, 4), (1, 256, 100), (1, 64, 160)])
def test_attention(emu, shape):
    P.check_attention(emu, *shape)


def test_forward_matches_reference_ncsnpp(emu):
    P.check_forward_golden(emu, "fwd_nf32", batch=1)


def test_profile_of_one_evaluation_times_the_ordinary_forward(emu):
    P.check_profile_forward(emu, "fwd_nf32")


def test_conv_tile_shape_never_changes_a_bit(emu):
    P.check_tile_independence(emu, "fwd_nf32", batch=1)


def test_xcd_aware_tile_order_never_changes_a_bit(emu):
    P.check_xcd_map_bitwise(emu, "fwd_nf32", batch=1)
    P.check_xcd_map_bitwise(emu, "fwd_nf32", batch=1, knob="SGMSE_SIDE_STREAM")     # (round 6; on the emulator: the deferred arena releases and the early shortcut launch)


def test_weight_reload(emu):
    P.check_weight_reload(emu)


def test_forward_matches_reference_ncsnpp_v2(emu):
    P.check_forward_golden(emu, "fwd_v2_nf32")


def test_adaptive_ode_sampler_drift_matches_the_reference_at_its_evaluation_points(emu):
    """(the whole 92-evaluation trajectory on the emulator: SGMSE_SLOW=1; on the GPU it always runs)"""
    P.check_ode_rk45(emu, full=bool(os.environ.get("SGMSE_SLOW")))


def test_sampler_new_code_score_wrapper(emu):
    P.check_sampler_v2(emu, "denoiser", "1/sigma", "edm", "1", "0", N=1)


def test_forward_matches_reference_ncsnpp_48k(emu):
    P.check_forward_golden(emu, "fwd_48k_nf32")


@pytest.mark.skipif(not SLOW, reason="full-width network on the emulator takes minutes; SGMSE_SLOW=1")
@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("SGMSE_SLOW"), reason="full-width network on the emulator (minutes); set SGMSE_SLOW=1")
@pytest.mark.parametrize("mode", [None, 1])
def test_forward_with_split_kernels_on_every_eligible_layer(emu, mode):
    P.check_forward_b3_everywhere(emu, mode=mode)


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("SGMSE_SLOW"), reason="full-width network on the emulator (minutes); set SGMSE_SLOW=1")
def test_split_kernel_workgroup_shapes_give_the_same_bits(emu):
    P.check_split_workgroup_shapes_bitwise(emu)


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("SGMSE_SLOW"), reason="full-width network on the emulator (minutes); set SGMSE_SLOW=1")
def test_forward_matches_reference_full_width(emu):
    P.check_forward_golden(emu, "fwd_nf128")


def test_pc_sampler_matches_reference(emu):
    P.check_sampler_golden(emu, "pc_N4", batch=1)


def test_pc_sampler_with_two_corrector_steps_matches_reference(emu):
    P.check_sampler_golden(emu, "pc_N4_c2", batch=1)


@pytest.mark.skipif(not SLOW, reason="SGMSE_SLOW=1")
@pytest.mark.slow
@pytest.mark.parametrize("tag", ["pnone_N6", "pfode_N6"])
def test_corrector_free_samplers_match_reference(emu, tag):
    P.check_sampler_golden(emu, tag, batch=1)


@pytest.mark.parametrize("fc,L", [(FO.FrontCfg(), 4000), (FO.FrontCfg.ears_48k(), 9000)])
def test_front_end(emu, fc, L):
    P.check_front_end(emu, fc, L)


@pytest.mark.parametrize("name", ["hann_exponent", "sqrthann_log", "hann_none", "sqrthann_exponent_48k"])
def test_front_end_matches_the_reference_data_module(emu, name):
    from test_oracle_golden import FRONT_CASES
    P.check_front_golden(emu, name, *FRONT_CASES[name])


def test_enhance_end_to_end(emu):
    P.check_enhance(emu, L=8000, N=1)


def test_extreme_groupnorm_parameters_keep_the_fp16x2_kernels(emu, capfd):
    """The fp16x2 3x3 kernel scales its input per utterance by a power of two derived from the utterance's own GroupNorm statistics
    and range bound (gn_finalize_kernel -> ConvArgs::xbound): a checkpoint with extreme affine parameters stays on the fast family
    (sgmse_conv_split_mode == 2, nothing on stderr) and still matches the oracle."""
    cfg = P.NET_CASES["fwd_nf32"]
    Pm = P.synth.synth_params(cfg, seed=0)
    x = torch.randn(1, 2, 256, 64, dtype=torch.complex64, generator=torch.Generator().manual_seed(3)) * 0.3
    t = torch.tensor([0.5])
    Pbig = {k: v.clone() for k, v in Pm.items()}
    for k in Pbig:
        if "GroupNorm" in k and k.endswith("weight"):
            Pbig[k][0] = 9.0; Pbig[k][1] = -40.0
        elif "GroupNorm" in k and k.endswith("bias"):
            Pbig[k][2] = 300.0
    net2, _ = P.make_backbone(cfg, emu, P=Pbig)
    out = net2(x, t)
    assert net2.engine(torch.device(emu)).conv_split_mode() == 2
    assert "bf16x3" not in capfd.readouterr().err
    with torch.no_grad():
        ref = P.NO.ncsnpp_forward(Pbig, cfg, x, t)
    assert P.rel_l2(out, ref) < P.NET_TOL


def test_adversarial_checkpoint_wild_groupnorm_parameters(emu):
    P.check_adversarial_checkpoint(emu, "gn_wild", nf=32, expect_mode=2)


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("SGMSE_SLOW"), reason="full-width network on the emulator (minutes); set SGMSE_SLOW=1")
def test_adversarial_checkpoint_single_weights_times_a_million(emu):
    P.check_adversarial_checkpoint(emu, "single_weights", nf=128, expect_mode=2)


def test_adversarial_checkpoint_residual_stream_growth(emu):
    P.check_adversarial_checkpoint(emu, "growth", nf=32, expect_mode=2)


def test_enhancement_script_directory_to_directory(emu, tmp_path, monkeypatch):
    P.check_enhancement_script(emu, tmp_path, monkeypatch)


def test_reference_enhancement_script_runs_unmodified(emu, tmp_path, monkeypatch):
    P.check_reference_script_unmodified(emu, tmp_path, monkeypatch)


def test_results_do_not_depend_on_what_device_memory_held(emu):
    P.check_poison_independence(emu, "fwd_nf32")


def test_ragged_batch_gives_every_utterance_its_single_run_bits(emu):
    P.check_ragged_batch(emu, "fwd_nf32", frames=(128, 64), quick=True)


def test_ragged_launches_over_the_widest_utterances_grid_give_the_same_bits(emu, monkeypatch):
    """The ragged layout that was the default until round 4 -- convolution grids over the WIDEST utterance's tile columns, plain tile
    order -- against single runs (three utterances, widths that leave partial tiles at the coarse levels); the default since (launches
    over the tile columns that exist, XCD-aware tile order) is what every other ragged test runs."""
    monkeypatch.setenv("SGMSE_RAGGED_PREFIX", "0")
    monkeypatch.setenv("SGMSE_CONV_XCD_MAP", "0")
    # (forward only on the emulator -- the samplers add two minutes here; the GPU test of the same name runs them too)
    P.check_ragged_batch(emu, "fwd_nf32", frames=(128, 64, 192), sampler=bool(os.environ.get("SGMSE_SLOW")), quick=True)


@pytest.mark.skipif(not os.environ.get("SGMSE_SLOW"), reason="full-width network on the emulator: minutes (SGMSE_SLOW=1)")
def test_ragged_batch_full_width(emu):
    P.check_ragged_batch(emu, "fwd_nf128", frames=(128, 64), sampler=False)


def test_ragged_batch_other_variants_and_minibatch(emu):
    P.check_ragged_variants(emu)

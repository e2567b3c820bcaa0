"""Host-side logic and the C-ABI surface, CPU only (no compute on a GPU): registries, pad_spec, state-dict contract,
Lightning-checkpoint ingestion with the EMA swap, exported symbols, error behaviour, sharding arithmetic."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import HIP_LIB, ROOT, rel_l2


def test_registries_match_reference_names():
    from sgmse_amd.backbones import BackboneRegistry
    from sgmse_amd.sdes import SDERegistry
    from sgmse_amd.sampling import PredictorRegistry, CorrectorRegistry
    assert set(BackboneRegistry.get_all_names()) >= {"ncsnpp", "ncsnpp_48k", "ncsnpp_v2"}
    assert set(SDERegistry.get_all_names()) == {"ouve", "sbve"}
    assert set(PredictorRegistry.get_all_names()) == {"euler_maruyama", "reverse_diffusion", "none"}
    assert set(CorrectorRegistry.get_all_names()) == {"langevin", "ald", "none"}
    for reg in (BackboneRegistry, SDERegistry, PredictorRegistry, CorrectorRegistry):
        with pytest.raises(ValueError):
            reg.get_by_name("no-such-thing")


def test_pad_spec():
    from sgmse_amd.util.other import pad_spec
    from oracle import stft_oracle as FO
    Y = torch.randn(2, 1, 8, 501, dtype=torch.complex64)
    for mode in ("zero_pad", "reflection", "replication"):
        out = pad_spec(Y, mode)
        assert out.shape[-1] == 512 and torch.equal(out, FO.pad_spec(Y, mode))
    assert pad_spec(Y[..., :448]).shape[-1] == 448
    with pytest.raises(NotImplementedError):
        pad_spec(Y, "circular")


def test_state_dict_contract_full_width():
    """647 tensors / 65,590,822 parameters, output_layer first; 48 kHz variant 64,739,854."""
    from sgmse_amd.backbones import BackboneRegistry
    from oracle import ncsnpp_oracle as NO
    net = BackboneRegistry.get_by_name("ncsnpp")()
    sd = net.state_dict()
    want = NO.param_shapes(NO.NetCfg.for_variant("ncsnpp"))
    assert list(sd.keys()) == list(want.keys())
    assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    assert sum(v.numel() for v in sd.values()) == 65_590_822
    assert [k for k, p in net.named_parameters() if not p.requires_grad] == ["all_modules.0.W"]
    net48 = BackboneRegistry.get_by_name("ncsnpp_48k")()
    assert sum(v.numel() for v in net48.state_dict().values()) == 64_739_854
    with pytest.raises(NotImplementedError):
        BackboneRegistry.get_by_name("ncsnpp")(resblock_type="ddpm")


def test_ouve_matches_oracle_scalars():
    from sgmse_amd.sdes import OUVESDE
    from oracle import sde_oracle as SO
    for (th, a, b, N, snr) in ((1.5, 0.05, 0.5, 30, 0.5), (2.0, 0.1, 1.0, 50, 0.33)):
        tab = OUVESDE(th, a, b, N=N).step_table(0.03, snr)
        ref = SO.step_table(SO.OUVE(th, a, b, N), 0.03, snr)
        for k in ref:
            assert torch.equal(tab[k], ref[k]), k
    s = OUVESDE(1.5, 0.05, 0.5, N=30)
    c = s.copy()
    assert (c.theta, c.sigma_min, c.sigma_max, c.N, c.sampler_type) == (1.5, 0.05, 0.5, 30, "pc") and s.T == 1


def test_sbve_step_table_matches_oracle_loop():
    """SBVESDE.sb_step_table (vectorised weights handed to the HIP loop) against the step-by-step oracle restatement of
    get_sb_sampler, using a linear fake model so the weights are the only arithmetic."""
    from sgmse_amd.sdes import SBVESDE
    from oracle import sde_oracle as SO
    y = torch.randn(1, 1, 4, 8, dtype=torch.complex64, generator=torch.Generator().manual_seed(0))
    model = lambda x, yy, t: 0.5 * x + 0.25 * yy
    for stype in ("ode", "sde"):
        tab = SBVESDE(2.6, 0.4, N=6).sb_step_table(1e-4, stype)
        rep = SO.NoiseReplay(5)
        ref, _ = SO.sb_sample(SO.SBVE(2.6, 0.4, 6), model, y, rep, eps=1e-4, sampler_type=stype)
        x = y.clone()
        for i in range(6):
            z = rep.draws[i] if stype == "sde" else 0.0
            x = tab["w_prev"][i] * x + tab["w_est"][i] * model(x, y, None) + tab["w_y"][i] * y + tab["w_z"][i] * z
        assert rel_l2(x, ref) < 1e-6, stype


def test_generic_python_sampler_matches_oracle_with_fake_score():
    """The registry predictor/corrector classes (Python loop) against the oracle with an analytic score."""
    from sgmse_amd import sampling
    from sgmse_amd.sdes import OUVESDE
    from oracle import sde_oracle as SO
    y = torch.randn(2, 1, 8, 16, dtype=torch.complex64, generator=torch.Generator().manual_seed(0))
    score = lambda x, yy, t: (yy - x) * 0.7
    for pred, corr in (("reverse_diffusion", "ald"), ("reverse_diffusion", "langevin"), ("reverse_diffusion", "none")):
        rep = SO.NoiseReplay(3)
        ref, nfe_ref = SO.pc_sample(SO.OUVE(1.5, 0.05, 0.5, 5), score, y, rep, eps=0.03, snr=0.5, corrector=corr, predictor=pred)
        rep2 = SO.NoiseReplay(3)
        orig = torch.randn_like
        torch.randn_like = lambda like, **kw: rep2(like)
        try:
            out, nfe = sampling.get_pc_sampler(pred, corr, OUVESDE(1.5, 0.05, 0.5, N=5), score, y, eps=0.03, snr=0.5)()
        finally:
            torch.randn_like = orig
        assert nfe == nfe_ref and rel_l2(out, ref) < 1e-6, (pred, corr)


def test_checkpoint_ingestion_and_ema_swap(tmp_path):
    """Lightning .ckpt layout: state_dict with 'dnn.' prefix, hyper_parameters incl. the pickled
    data-module class, 'ema' shadow parameters used by eval() and restored by train()."""
    from sgmse_amd.model import ScoreModel
    from sgmse_amd.data_module import SpecsDataModule
    hp = dict(backbone="ncsnpp", sde="ouve", nf=32, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30, t_eps=0.03,
              data_module_cls=SpecsDataModule, n_fft=510, hop_length=128, spec_factor=0.15, spec_abs_exponent=0.5,
              no_wandb=True)
    src = ScoreModel(**{k: v for k, v in hp.items() if k != "no_wandb"})
    sd = {"dnn." + k: v.clone() for k, v in src.dnn.state_dict().items()}
    shadow = [p.detach() * 2 + 1 for p in src.dnn.parameters() if p.requires_grad]     # torch_ema tracks trainable params
    path = tmp_path / "m.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": hp, "ema": {"decay": 0.999, "num_updates": 1,
                                                                     "shadow_params": shadow, "collected_params": None}}, path)
    m = ScoreModel.load_from_checkpoint(str(path), map_location="cpu")
    assert m.backbone == "ncsnpp" and m.sde.__class__.__name__ == "OUVESDE" and m.t_eps == 0.03 and m.sr == 16000
    raw = [p.detach().clone() for p in m.dnn.parameters() if p.requires_grad]
    m.eval()
    assert all(torch.equal(p, s) for p, s in zip([q for q in m.dnn.parameters() if q.requires_grad], shadow))
    m.train(True)
    assert all(torch.equal(p, r) for p, r in zip([q for q in m.dnn.parameters() if q.requires_grad], raw))
    m.eval(no_ema=True)
    assert all(torch.equal(p, r) for p, r in zip([q for q in m.dnn.parameters() if q.requires_grad], raw))
    with pytest.warns(UserWarning):
        torch.save({"state_dict": sd, "hyper_parameters": hp}, path)
        ScoreModel.load_from_checkpoint(str(path))


def test_device_code_holds_no_packed_fp32_instruction(tmp_path):
    """Round 6: conv3x3_thin_kernel's v_pk_fma_f32 gave results that changed whenever another process shared the GPU (kernels_conv_thin.h);
    the kernel spells its FMAs out and the build passes -packed-fp32-ops: the gfx950 code object must not contain v_pk_{fma,mul,add}_f32."""
    import shutil
    from conftest import HIP_LIB
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(HIP_LIB) and os.path.exists(objdump)):
        pytest.skip("library or llvm-objdump missing")
    lib = tmp_path / "lib.so"
    shutil.copy(HIP_LIB, lib)
    subprocess.run([objdump, "--offloading", str(lib)], cwd=tmp_path, capture_output=True, check=True)
    co = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert len(co) == 1, os.listdir(tmp_path)
    dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(tmp_path / co[0])], capture_output=True, text=True, check=True).stdout
    assert dis.count("v_mfma_f32_32x32x16_f16") > 100          # (the disassembly is the real thing)
    bad = [l for l in dis.splitlines() if "v_pk_fma_f32" in l or "v_pk_mul_f32" in l or "v_pk_add_f32" in l]
    assert not bad, bad[:5]


def test_c_abi_exports_every_declared_symbol():
    """libsgmse_hip.so loads without a GPU and exports exactly what include/sgmse_hip.h declares."""
    assert os.path.exists(HIP_LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    hdr = open(os.path.join(ROOT, "include", "sgmse_hip.h")).read()
    declared = set(re.findall(r"\b(sgmse_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(HIP_LIB)
    for name in declared:
        assert hasattr(lib, name), name
    from sgmse_amd import _lib
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    lib.sgmse_backend.restype = ctypes.c_char_p
    assert lib.sgmse_backend() == b"hip-gfx950"


def test_product_path_fails_loudly_without_gpu():
    """No CPU / PyTorch fallback: with the HIP library loaded and no GPU visible, creating a context raises."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch\nfrom sgmse_amd import _lib\n_lib.load_library()\n"
            "try:\n    _lib.Context('cuda')\nexcept _lib.SgmseLibraryError as e:\n    print('RAISED', e)\n"
            "try:\n    _lib.Context('cpu')\nexcept _lib.SgmseLibraryError as e:\n    print('RAISED2', e)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED " in out.stdout and "RAISED2" in out.stdout, out.stdout + out.stderr


def test_error_codes_and_messages(emu):
    from sgmse_amd import _lib, ops
    ctx = _lib.Context("cpu")
    with pytest.raises(ValueError):
        ctx.configure(variant="dcunet", nf=32, ch_mult=(1,), num_res_blocks=1, attn_resolutions=(), image_size=256,
                      progressive="none", progressive_input="none")
    with pytest.raises(ValueError):
        ctx.configure(variant="ncsnpp", nf=33, ch_mult=(1,), num_res_blocks=1, attn_resolutions=(), image_size=256,
                      progressive="none", progressive_input="none")
    ctx.configure(variant="ncsnpp", nf=32, ch_mult=(1, 1), num_res_blocks=1, attn_resolutions=(), image_size=256,
                  progressive="none", progressive_input="none")
    with pytest.raises(RuntimeError, match="weights not loaded"):
        ctx.forward(torch.zeros(1, 2, 4, 4, dtype=torch.complex64), torch.ones(1))
    with pytest.raises(RuntimeError, match="missing parameter"):
        ctx.load_weights({"output_layer.weight": torch.zeros(2, 4, 1, 1)})
    with pytest.raises(ValueError):
        ops.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 5, 5))
    with pytest.raises(ValueError):
        ops.conv2d(torch.zeros(1, 8, 4, 4, dtype=torch.float64), torch.zeros(8, 8, 3, 3))
    with pytest.raises(RuntimeError, match="length"):
        ops.istft(torch.zeros(1, 256, 4, dtype=torch.complex64), 510, 128, torch.hann_window(510), length=10000)
    with pytest.raises(NotImplementedError):
        ops.spec_transform(torch.zeros(4, dtype=torch.complex64), "mel", 1.0, 1.0, False)
    # an input whose height is not the image_size the module list was built for: the reference's forward places attention by
    # the actual height (ncsnpp.py:308), runs off its module list and dies with a TypeError; here the error says why
    import parity as P
    from oracle import ncsnpp_oracle as NO
    net, _ = P.make_backbone(NO.NetCfg.for_variant("ncsnpp", nf=32), "cpu")
    with pytest.raises(RuntimeError, match="image_size"):
        net(torch.zeros(1, 2, 64, 64, dtype=torch.complex64), torch.ones(1))


def test_shard_range():
    from sgmse_amd.parallel import shard_range
    for n, w in ((256, 8), (10, 4), (3, 8), (7, 1)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert all(b - a == n // w for a, b in spans[:-1])


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from sgmse_amd import _lib
_lib.load_library({emu!r})
from sgmse_amd.model import ScoreModel
from sgmse_amd.parallel import shard_range, broadcast_backbone_weights
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT weights; rank 0's must win
m = ScoreModel('ncsnpp', 'ouve', nf=32, theta=1.5, sigma_min=0.05, sigma_max=0.5)
if rank != 0:
    with torch.no_grad():
        for p in m.dnn.parameters(): p.add_(1.0)
m.eval()
broadcast_backbone_weights(m.dnn, src=0)
g = torch.Generator().manual_seed(5)
wav = torch.randn(2, 8000, generator=g)              # the whole 2-utterance "file list"
a, b = shard_range(2, rank, world)
out, nfe = m.enhance_batch(wav[a:b], N=1, corrector='none', seed=11 + a)
gathered = [torch.zeros(1, 8000) for _ in range(world)]
dist.all_gather(gathered, out.contiguous())
if rank == 0:
    torch.save(torch.cat(gathered), {out!r})
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_utterance_sharding_gloo(emu, tmp_path):
    """World size 2 on CPU (gloo): weights broadcast from rank 0, utterances sharded contiguously, no collective on the
    data path; the gathered result equals a single-process run."""
    from conftest import EMU_LIB
    out = str(tmp_path / "gathered.pt")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, emu=EMU_LIB, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2", SGMSE_EMU_THREADS="4")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = torch.load(out)
    from sgmse_amd.model import ScoreModel
    torch.manual_seed(100)
    m = ScoreModel("ncsnpp", "ouve", nf=32, theta=1.5, sigma_min=0.05, sigma_max=0.5)
    m.eval()
    wav = torch.randn(2, 8000, generator=torch.Generator().manual_seed(5))
    ref = torch.cat([m.enhance_batch(wav[i:i + 1], N=1, corrector="none", seed=11 + i)[0] for i in range(2)])
    assert torch.equal(got, ref)


def test_reference_package_name_alias(tmp_path):
    """`sgmse_amd/compat` on the path makes the reference's import lines (enhancement.py:15-16) and a checkpoint that
    pickles `sgmse.data_module.SpecsDataModule` resolve to this implementation."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import torch, pickle\n"
            "from sgmse.model import ScoreModel\n"
            "from sgmse.util.other import pad_spec, set_torch_cuda_arch_list\n"
            "from sgmse.backbones.shared import BackboneRegistry\n"
            "from sgmse.data_module import SpecsDataModule\n"
            "import sgmse_amd.model\n"
            "assert ScoreModel is sgmse_amd.model.ScoreModel\n"
            "assert SpecsDataModule.__module__ == 'sgmse_amd.data_module'\n"
            "blob = b'csgmse.data_module\\nSpecsDataModule\\n.'\n"     # what a Lightning ckpt's hyper_parameters contain
            "assert pickle.loads(blob) is SpecsDataModule\n"
            "set_torch_cuda_arch_list(); print('ALIAS-OK', BackboneRegistry.get_all_names())\n"
            ) % (ROOT, os.path.join(ROOT, "sgmse_amd", "compat"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "ALIAS-OK" in out.stdout, out.stdout + out.stderr


def test_evaluation_metrics_match_reference_functions():
    """f4: the NumPy metrics against outputs of the reference's own functions (fixture from oracle/make_golden_metrics.py)."""
    from conftest import GOLDEN
    from sgmse_amd.util import other as O
    z = np.load(os.path.join(GOLDEN, "metrics.npz"))
    s, n, s_hat = z["s"], z["n"], z["s_hat"]
    for got, want in zip(O.si_sdr_components(s_hat, s, n), (z["s_target"], z["e_noise"], z["e_art"])):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.energy_ratios(s_hat, s, n), z["energy_ratios"], rtol=1e-12)
    np.testing.assert_allclose(O.si_sdr(s, s_hat), z["si_sdr"], rtol=1e-12)
    np.testing.assert_allclose(O.snr_dB(s, n), z["snr_dB"], rtol=1e-12)
    np.testing.assert_allclose(O.hp_filter(s_hat), z["hp"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.hp_filter(s_hat, cut_off=120, order=6, sr=48000), z["hp_48k"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(O.mean_conf_int(z["vals"]), z["mean_conf_int"], rtol=1e-12)
    np.testing.assert_allclose(O.mean_conf_int(z["vals"], 0.9), z["mean_conf_int_90"], rtol=1e-12)
    np.testing.assert_allclose(O.mean_std(z["with_nan"]), z["mean_std"], rtol=1e-12)
    m = O.Method("m", "/x", ["a"])
    for v in z["vals"]:
        m.append("a", v)
    np.testing.assert_allclose(m.get_mean_ci("a"), z["mean_conf_int"], rtol=1e-12)
    assert O.print_mean_std(z["with_nan"], 1) == "{:.1f} ± {:.1f}".format(*z["mean_std"])


def test_calc_metrics_on_a_wav_directory(tmp_path):
    """calc_metrics end to end: clean / noisy / enhanced wav directories (incl. the `<id>_<snr>dB.wav` naming rule and a
    sub-directory) -> per-file SI-SDR/SIR/SAR, csv and summary files."""
    from scipy.io import wavfile
    from sgmse_amd import calc_metrics as CM
    from sgmse_amd.util.other import energy_ratios
    rng = np.random.default_rng(0)
    dirs = {k: tmp_path / k for k in ("clean", "noisy", "enh")}
    for d in dirs.values():
        (d / "sub").mkdir(parents=True)
    expect = {}
    for name, clean_name in (("p1_5dB.wav", "p1.wav"), ("sub/p2.wav", "sub/p2.wav")):
        x = (0.3 * rng.standard_normal(3000)).astype(np.float32)
        n = (0.1 * rng.standard_normal(3000)).astype(np.float32)
        xh = (x + 0.3 * n + 0.01 * rng.standard_normal(3000)).astype(np.float32)
        wavfile.write(str(dirs["clean"] / clean_name), 16000, x)
        wavfile.write(str(dirs["noisy"] / name), 16000, x + n)
        wavfile.write(str(dirs["enh"] / name), 16000, xh)
        expect[name] = energy_ratios(xh.astype(np.float64), x.astype(np.float64), (x + n).astype(np.float64) - x.astype(np.float64))
    data = CM.main(["--clean_dir", str(dirs["clean"]), "--noisy_dir", str(dirs["noisy"]), "--enhanced_dir", str(dirs["enh"])])
    assert sorted(data["filename"]) == sorted(expect)
    for i, fn in enumerate(data["filename"]):
        np.testing.assert_allclose((data["si_sdr"][i], data["si_sir"][i], data["si_sar"][i]), expect[fn], rtol=1e-9)
    assert (dirs["enh"] / "_results.csv").read_text().splitlines()[0] == "filename,pesq,estoi,si_sdr,si_sir,si_sar"
    assert "SI-SDR:" in (dirs["enh"] / "_avg_results.txt").read_text()


def test_checkpoint_that_pickles_the_reference_package_name(tmp_path):
    """A checkpoint written by the reference's train.py pickles ``sgmse.data_module.SpecsDataModule`` (a GLOBAL opcode naming
    the reference package) inside hyper_parameters.  load_from_checkpoint must resolve it without sgmse_amd/compat being on
    PYTHONPATH (child process: no alias set up by the caller)."""
    import pickle
    from sgmse_amd.model import ScoreModel
    hp = dict(backbone="ncsnpp", sde="ouve", nf=32, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30, t_eps=0.03)
    src = ScoreModel(**hp)
    path = tmp_path / "ref.ckpt"
    torch.save({"state_dict": {"dnn." + k: v.clone() for k, v in src.dnn.state_dict().items()},
                "hyper_parameters": dict(hp, data_module_cls="PLACEHOLDER")}, path)
    # rewrite the placeholder into the pickle opcode sequence `GLOBAL sgmse.data_module SpecsDataModule`
    import zipfile
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        blobs = {n: z.read(n) for n in names}
    pkl = [n for n in names if n.endswith("data.pkl")][0]
    marker = pickle.dumps("PLACEHOLDER", protocol=2)[2:-1]          # BINUNICODE + payload (+ memo put), without PROTO / STOP
    assert blobs[pkl].count(marker[:16]) == 1
    head = marker[:5 + len("PLACEHOLDER")]
    blobs[pkl] = blobs[pkl].replace(head, b"csgmse.data_module\nSpecsDataModule\n", 1)
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        for n in names:
            z.writestr(n, blobs[n])
    code = ("import sys, warnings; sys.path.insert(0, %r)\n"
            "warnings.simplefilter('ignore')\n"
            "assert 'sgmse' not in sys.modules\n"
            "from sgmse_amd.model import ScoreModel\n"
            "from sgmse_amd.data_module import SpecsDataModule\n"
            "m = ScoreModel.load_from_checkpoint(%r)\n"
            "assert m.hparams['data_module_cls'] is SpecsDataModule and isinstance(m.data_module, SpecsDataModule)\n"
            "assert not any(p.endswith('compat') for p in sys.path)\n"
            "print('CKPT-OK', m.backbone)\n") % (ROOT, str(path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k != "PYTHONPATH"})
    assert "CKPT-OK ncsnpp" in out.stdout, out.stdout + out.stderr


def test_euler_maruyama_predictor_is_pinned_to_the_reference_behaviour():
    """'euler_maruyama' cannot run inside the PC loop, in the reference as here: the loop hands ``stepsize`` to update_fn,
    which forwards it into OUVESDE.sde(x, y, t) (predictors.py:49 -> sdes.py:114-120) -> TypeError.  Called directly it is
    the Euler-Maruyama step of the reverse SDE (oracle: euler_maruyama_update)."""
    from sgmse_amd import sampling
    from sgmse_amd.sdes import OUVESDE
    from oracle import sde_oracle as SO
    y = torch.randn(2, 1, 8, 16, dtype=torch.complex64, generator=torch.Generator().manual_seed(0))
    score = lambda x, yy, t: (yy - x) * 0.7
    sde = OUVESDE(1.5, 0.05, 0.5, N=5)
    with pytest.raises(TypeError):
        sampling.get_pc_sampler("euler_maruyama", "none", sde, score, y, eps=0.03)()
    pred = sampling.PredictorRegistry.get_by_name("euler_maruyama")(sde, score)
    t = torch.full((2,), 0.6)
    rep, rep2 = SO.NoiseReplay(3), SO.NoiseReplay(3)
    ref, ref_mean = SO.euler_maruyama_update(SO.OUVE(1.5, 0.05, 0.5, 5), score, y, y * 0.5, t, rep)
    orig = torch.randn_like
    torch.randn_like = lambda like, **kw: rep2(like)
    try:
        out, mean = pred.update_fn(y, y * 0.5, t)
    finally:
        torch.randn_like = orig
    assert rel_l2(out, ref) < 1e-6 and rel_l2(mean, ref_mean) < 1e-6


def test_adaptive_ode_sampler_is_the_scipy_solver_over_the_probability_flow_drift():
    """get_ode_sampler(denoise=False) -- the only way the reference's function completes -- integrates theta (y - x) - g^2 score / 2
    with scipy.integrate.solve_ivp from T to eps over the flattened complex state and returns the solver's own evaluation count;
    adaptive=True with denoise=True adds the predictor step the reference's code intends (the reference raises TypeError there)."""
    import numpy as np
    from scipy import integrate
    from sgmse_amd import sampling
    from sgmse_amd.sdes import OUVESDE
    g = torch.Generator().manual_seed(3)
    y = torch.complex(torch.randn(2, 1, 4, 8, generator=g), torch.randn(2, 1, 4, 8, generator=g))
    z0 = torch.complex(torch.randn(2, 1, 4, 8, generator=g), torch.randn(2, 1, 4, 8, generator=g))
    sde = OUVESDE(1.5, 0.05, 0.5, N=3)
    score = lambda x, yy, t: (yy - x) * (1.0 + t[:, None, None, None])
    out, nfe = sampling.get_ode_sampler(sde, score, y, denoise=False, rtol=1e-6, atol=1e-8)(z=z0)

    def f(t, xf):
        x = torch.from_numpy(xf.reshape(tuple(y.shape))).type(torch.complex64)
        tt = torch.ones(2) * t
        gg = sde.sde(x, y, tt)[1]
        d = 1.5 * (y - x) - gg[:, None, None, None] ** 2 * score(x, y, tt) * 0.5
        return d.numpy().reshape(-1)
    sol = integrate.solve_ivp(f, (sde.T, 0.03), z0.numpy().reshape(-1), rtol=1e-6, atol=1e-8, method="RK45")
    ref = torch.tensor(sol.y[:, -1]).reshape(y.shape).type(torch.complex64)
    assert nfe == sol.nfev and nfe > 3 and rel_l2(out, ref) < 1e-6
    # other solvers and their keyword arguments pass through
    out2, nfe2 = sampling.get_ode_sampler(sde, score, y, denoise=False, method="RK23", rtol=1e-4, atol=1e-6, first_step=1e-3)(z=z0)
    assert nfe2 != nfe and rel_l2(out2, ref) < 1e-3
    # adaptive + denoise: one reverse-diffusion predictor step of size eps at t = eps, its mean returned
    out3, nfe3 = sampling.get_ode_sampler(sde, score, y, adaptive=True, denoise=True, rtol=1e-6, atol=1e-8)(z=z0)
    te = torch.ones(2) * 0.03
    fz, gz = sde.discretize(out, y, te, torch.tensor(0.03))
    want = out - (fz - gz[:, None, None, None] ** 2 * score(out, y, te))
    assert nfe3 == nfe and rel_l2(out3, want) < 1e-6
    with pytest.raises(TypeError):
        sampling.get_ode_sampler(sde, score, [y[0], y[1]], denoise=False)


def test_ode_sampler_warns_about_ignored_solver_arguments():
    from sgmse_amd import sampling
    from sgmse_amd.sdes import OUVESDE
    y = torch.zeros(1, 1, 8, 16, dtype=torch.complex64)
    score = lambda x, yy, t: yy - x
    with pytest.warns(UserWarning, match="rtol"):
        sampling.get_ode_sampler(OUVESDE(1.5, 0.05, 0.5, N=3), score, y, rtol=1e-3)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out, nfe = sampling.get_ode_sampler(OUVESDE(1.5, 0.05, 0.5, N=3), score, y)()
    assert nfe == 3 and out.shape == y.shape


def test_minibatch_samplers_equal_one_big_batch(emu):
    """get_pc_sampler / get_ode_sampler(minibatch=m) (reference model.py:356-368, 378-390) run the batch in serial chunks;
    utterances are independent, so with replayed noise the chunks reproduce the big batch."""
    import parity as P
    from oracle import ncsnpp_oracle as NO, synth
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32, image_size=64)      # a quarter of the frequency bins: this test is about the host wrappers
    m, _ = P.make_model(cfg, emu)
    y = synth.synth_spec(3, 64, 64, seed=3)
    N = 1
    full, nfe = m.get_pc_sampler("reverse_diffusion", "ald", y, N=N, snr=0.5, noise=P.replay_noise(y.shape, 1 + 2 * N))()
    assert nfe == 2 * N
    # same draws per utterance: chunk i of size 2 replays rows [2i, 2i+2) of every draw
    outs = []
    for lo in (0, 2):
        yy = y[lo:lo + 2]
        noise = P.replay_noise(y.shape, 1 + 2 * N)[:, lo:lo + 2].contiguous()
        outs.append(m.get_pc_sampler("reverse_diffusion", "ald", yy, N=N, snr=0.5, noise=noise)()[0])
    assert torch.equal(torch.cat(outs), full)
    # the minibatch wrapper itself, in-kernel noise: a chunk's utterances keep the noise-stream ids of their position in the
    # whole batch, so the chunked run reproduces the unchunked one
    big, _ = m.get_pc_sampler("reverse_diffusion", "ald", y, N=N, snr=0.5, seed=5)()
    got, ns = m.get_pc_sampler("reverse_diffusion", "ald", y, N=N, snr=0.5, minibatch=2, seed=5)()
    assert ns == [2 * N, 2 * N] and torch.equal(got, big)
    # an utterance's noise is a function of (seed, stream id): any slot, any batch
    perm = [2, 0, 1]
    moved, _ = m.get_pc_sampler("reverse_diffusion", "ald", y[perm].contiguous(), N=N, snr=0.5, seed=5, streams=perm)()
    assert torch.equal(moved, big[perm])
    got, ns = m.get_ode_sampler(y, N=2, minibatch=2, seed=5)()
    assert ns == [2, 2] and torch.equal(got, m.get_ode_sampler(y, N=2, seed=5)()[0])


def test_none_predictor_behind_a_corrector_returns_the_noisy_iterate(emu):
    """NonePredictor.update_fn returns (x, x) (predictors.py:69-76): with a corrector in front, xt_mean is overwritten by the
    noisy xt, so denoise=True returns xt.  Fused loop == reference-style Python loop over the same network and noise."""
    import parity as P
    from oracle import ncsnpp_oracle as NO, sde_oracle as SO, synth
    cfg = NO.NetCfg.for_variant("ncsnpp", nf=32)
    m, Pm = P.make_model(cfg, emu)
    y = synth.synth_spec(1, 256, 64, seed=3)
    N = 2
    rep = SO.NoiseReplay(7)
    score = lambda a, b, c: NO.score_fn(Pm, cfg, a, b, c)
    ref, nfe_ref = SO.pc_sample(SO.OUVE(1.5, 0.05, 0.5, N), score, y, rep, eps=0.03, snr=0.5, corrector="ald", predictor="none")
    ref_noisy, _ = SO.pc_sample(SO.OUVE(1.5, 0.05, 0.5, N), score, y, SO.NoiseReplay(7), eps=0.03, snr=0.5, corrector="ald",
                                predictor="none", denoise=False)
    assert torch.equal(ref, ref_noisy)            # x_mean == x behind a NonePredictor
    noise = torch.stack(rep.draws)
    out, nfe = m.get_pc_sampler("none", "ald", y, N=N, snr=0.5, noise=noise)()
    assert nfe == nfe_ref and rel_l2(out, ref) < 1e-4
    out2, _ = m.get_pc_sampler("none", "ald", y, N=N, snr=0.5, noise=noise, use_graph=False)()
    assert torch.equal(out, out2)


def test_bench_gpus_2_spawns_two_ranks(emu):
    """`python bench.py --gpus 2` started WITHOUT torchrun starts its two ranks itself (one per GPU on the GPU box) and rank 0's
    JSON line says n_gpus = 2, with the weight-broadcast time, per-rank rates and the collective's world size.  Here the ranks
    run the --test-emulator mode (CPU tensors, gloo, reduced-width network): launcher and reporting logic only."""
    import json
    from conftest import EMU_LIB
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SGMSE_EMU_THREADS="4", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1",
                          "--seconds", "0.5", "--N", "1", "--test-emulator", EMU_LIB], capture_output=True, text=True, timeout=1500, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout + out.stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["unit"] == "utterances/s"
    assert d["collective_backend"]["world_size"] == 2 and len(d["per_rank_utt_per_s"]) == 2 and d["weight_broadcast_ms"] > 0
    assert d["communicator_setup_ms"] > 0 and d["weight_broadcast_gbps"] > 0      # (the warm-up broadcast is timed apart from the 262 MB one)
    assert d["config"]["batch_per_gpu"] == 1 and "TEST ONLY" in d["data"]
    assert abs(d["value"] - 2 * 1 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]


def test_bench_under_the_drivers_torch_distributed_run_line(emu):
    """The launch line of the round-end scaling bench: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from
    the environment, rank 0 prints the one JSON line.  (--test-emulator: CPU tensors, gloo, reduced-width network.)"""
    import json
    from conftest import EMU_LIB
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(SGMSE_EMU_THREADS="4", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29578", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--batch", "1", "--seconds", "0.5", "--N", "1", "--test-emulator", EMU_LIB],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout + out.stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective_backend"]["world_size"] == 2 and len(d["per_rank_utt_per_s"]) == 2
    assert d["weight_broadcast_ms"] > 0 and d["scaling"] == "weak"


def test_bench_at_eight_ranks_prints_one_line_for_the_whole_job(emu):
    """The driver's 8-GPU scaling run, rehearsed: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8`
    (CPU tensors, gloo, emulator, reduced-width network).  Exactly one JSON line, from rank 0; n_gpus = 8; eight per-rank rates;
    value = all ranks' utterances over the slowest rank's time; no rank prints a second line or a power / cpu_baseline object."""
    import json
    from conftest import EMU_LIB
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(SGMSE_EMU_THREADS="1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", "29591", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0",
                          "--batch", "1", "--seconds", "0.5", "--N", "1", "--test-emulator", EMU_LIB],
                         capture_output=True, text=True, timeout=2400, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-3000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["collective_backend"]["world_size"] == 8 and len(d["per_rank_utt_per_s"]) == 8
    assert d["scaling"] == "weak" and d["config"]["batch_per_gpu"] == 1 and d["weight_broadcast_ms"] > 0
    assert abs(d["value"] - 8 * 1 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]          # whole-job aggregate over the slowest rank
    assert min(d["per_rank_utt_per_s"]) * 8 <= d["value"] * (1 + 1e-6)
    assert "cpu_baseline" not in d and "power" not in d and "other_workloads" not in d        # rank-0-at-N=1-only legs stay out


def test_assign_files_balances_by_padded_frames():
    """--balance frames: longest-processing-time assignment by padded frame count; a partition of the file list that every rank
    computes identically; --balance contiguous is the reference's split (model.py:212-223)."""
    from sgmse_amd.enhancement import assign_files, padded_frames
    from sgmse_amd.parallel import shard_range
    rng = np.random.default_rng(3)
    frames = [padded_frames(int(n), 128) for n in rng.lognormal(np.log(48000), 0.6, size=101)]
    for world in (2, 3, 8):
        parts = [assign_files(frames, r, world, "frames") for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(len(frames)))
        loads = [sum(frames[i] for i in p) for p in parts]
        assert max(loads) / (sum(loads) / world) <= 1.1, loads
        cont = [assign_files(frames, r, world, "contiguous") for r in range(world)]
        assert [(p[0], p[-1] + 1) for p in cont] == [shard_range(len(frames), r, world) for r in range(world)]
    with pytest.raises(ValueError):
        assign_files(frames, 0, 2, "random")


def test_plan_batches_follows_the_cost_model():
    from sgmse_amd.enhancement import plan_batches
    frames = [512] * 70 + [576] * 3 + [640] * 2 + [1024]
    idx = list(range(len(frames)))
    plain = plan_batches(idx, frames, 32, ragged=False)
    assert sorted(i for b in plain for i in b) == idx and all(len({frames[i] for i in b}) == 1 for b in plain)
    rag = plan_batches(idx, frames, 32, ragged=True)
    assert sorted(i for b in rag for i in b) == idx
    assert sum(len(b) == 32 and len({frames[i] for i in b}) == 1 for b in rag) == 2       # full uniform batches stay uniform
    assert any(len({frames[i] for i in b}) > 1 for b in rag)                                # the leftovers were pooled: 12 files, one batch
    # a ragged batch priced above the small-batch penalty it avoids: the leftovers stay bucketed
    assert all(len({frames[i] for i in b}) == 1 for b in plan_batches(idx, frames, 32, ragged=True, ragged_factor=3.0))


_DIR_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch
from sgmse_amd import _lib
_lib.load_library({emu!r})
from sgmse_amd import enhancement as E
box = []
orig = E.load_model
def spy(*a):
    m = orig(*a); box.append(m); return m
E.load_model = spy
import warnings
warnings.simplefilter("ignore")
n = E.main({argv!r})
m = box[0]
aff = m.score_affine(torch.linspace(1.0, 0.03, 5))
torch.save(dict(n=n, affine=aff, hparams={{k: v for k, v in m.hparams.items() if k != 'data_module_cls'}},
                w0=next(m.dnn.parameters()).detach().clone()), {out!r} + os.environ['RANK'])
"""


def test_three_rank_directory_job_balanced_by_frames_with_an_edm_checkpoint(emu, tmp_path):
    """python -m sgmse_amd.enhancement under 3 ranks (gloo, emulator): rank 0 reads an ncsnpp_v2 checkpoint with the EDM
    preconditioning (c_in / c_out / c_skip = 'edm', network_scaling '1/sigma'), the other ranks are built from the broadcast
    hyper-parameters and must apply the SAME score wrapper (hparams used to drop those arguments); files of skewed
    lengths are assigned by padded frame count (max / mean frames per rank <= 1.1) and the enhanced files equal the
    single-process run bit for bit."""
    from scipy.io import wavfile
    from conftest import EMU_LIB
    from sgmse_amd import enhancement as E
    from sgmse_amd.model import ScoreModel
    from sgmse_amd.data_module import SpecsDataModule
    hp = dict(backbone="ncsnpp_v2", sde="ouve", nf=32, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=30, t_eps=0.03,
              loss_type="score_matching", c_in="edm", c_out="edm", c_skip="edm", network_scaling="1/sigma", sigma_data=0.2,
              data_module_cls=SpecsDataModule, n_fft=510, hop_length=128, spec_factor=0.15, spec_abs_exponent=0.5)
    torch.manual_seed(4)
    src = ScoreModel(**hp)
    assert ScoreModel(**src.hparams).c_in == "edm" and ScoreModel(**src.hparams).network_scaling == "1/sigma"
    ckpt = tmp_path / "m.ckpt"
    torch.save({"state_dict": {"dnn." + k: v.clone() for k, v in src.dnn.state_dict().items()}, "hyper_parameters": hp}, ckpt)
    noisy = tmp_path / "noisy"
    noisy.mkdir()
    rng = torch.Generator().manual_seed(5)
    lengths = [30000, 5000, 5000, 5000, 9000, 5000, 5000]          # padded frames 256, 64, 64, 64, 128, 64, 64 (reflection pad < frames)
    for i, L in enumerate(lengths):
        wavfile.write(str(noisy / f"f{i}.wav"), 16000, (0.1 * torch.randn(L, generator=rng)).numpy())
    frames = [E.padded_frames(L, 128) for L in lengths]
    loads = [sum(frames[i] for i in E.assign_files(frames, r, 3, "frames")) for r in range(3)]
    assert max(loads) / (sum(loads) / 3) <= 1.1 and max(sum(frames[i] for i in E.assign_files(frames, r, 3, "contiguous")) for r in range(3)) > max(loads)
    base = ["--test_dir", str(noisy), "--ckpt", str(ckpt), "--device", "cpu", "--N", "1", "--corrector", "none", "--seed", "7", "--batch_size", "4"]
    script = tmp_path / "worker.py"
    out = str(tmp_path / "rank")
    script.write_text(_DIR_WORKER.format(root=ROOT, emu=EMU_LIB, out=out, argv=base + ["--enhanced_dir", str(tmp_path / "o3"), "--balance", "frames"]))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="3", SGMSE_EMU_THREADS="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(3)]
    logs = [p.communicate(timeout=1500)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [torch.load(out + str(r), weights_only=False) for r in range(3)]
    assert sum(r["n"] for r in res) == len(lengths)
    for r in res[1:]:                                              # same wrapper, same weights on every rank
        assert r["hparams"] == res[0]["hparams"] and torch.equal(r["w0"], res[0]["w0"])
        assert all(torch.equal(a, b) for a, b in zip(r["affine"], res[0]["affine"]))
    assert res[0]["hparams"]["c_in"] == "edm" and not torch.equal(res[0]["affine"][0], torch.ones(5))
    with pytest.warns(UserWarning):                                # single process, contiguous, other batches: the same files
        assert E.main(base + ["--enhanced_dir", str(tmp_path / "o1")]) == len(lengths)
    for i in range(len(lengths)):
        a, b = wavfile.read(str(tmp_path / "o1" / f"f{i}.wav"))[1], wavfile.read(str(tmp_path / "o3" / f"f{i}.wav"))[1]
        assert np.array_equal(a, b) and np.isfinite(a).all() and np.abs(a).max() > 0, i


def test_bench_reads_the_committed_counters_of_its_own_run(tmp_path):
    """bench.py::pmc_of_the_bench_run takes roofline.traffic / mfma_busy from profiles/*pmc_bench_b<batch>.json (counters of the bench
    command's own launches, tools/summarize_pmc_bench.py) and falls back to nothing when no file matches; the summariser turns rocprofv3
    counter_collection CSVs + a per-launch listing into that file."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prefix = bench.DOMINANT[3][3]
    got = bench.pmc_of_the_bench_run(prefix, 32, "pc16k")
    assert got, "profiles/ holds no *pmc_bench_b32.json"
    assert 0.2 < got["mfma_busy"] < 1.0 and 1e9 < got["traffic"] < 1e10 and "pmc_bench_b32.json" in got["pmc_source"]
    assert 1.0 < got["effective_clock_ghz"] < 2.5 and got["valu_per_mfma"] > 1.0
    assert "committed" in got["pmc_source"] and isinstance(got["pmc_matches_this_build"], bool)     # (provenance, not "this run")
    assert bench.pmc_of_the_bench_run(prefix, 7, "pc16k") == {}
    assert bench.pmc_of_the_bench_run(prefix, 32, "ode16k") == {}      # counters of another workload's command are not applied
    # the summariser on a synthetic pass: two launches of one kernel, one counter group per directory
    rows = "Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n"
    k = '"void sgmse::conv3x3_wino_kernel<8, 1, 0, 0, 0>(sgmse::ConvArgs)"'
    d1, d2 = tmp_path / "a" / "host", tmp_path / "b" / "host"
    d1.mkdir(parents=True); d2.mkdir(parents=True)
    (d1 / "1_counter_collection.csv").write_text(rows + "".join(
        f"{i},{k},{n},{v},{1000 * i},{1000 * i + 2000000}\n" for i in (1, 2)
        for n, v in (("SQ_VALU_MFMA_BUSY_CYCLES", 2.0e9), ("GRBM_GUI_ACTIVE", 8 * 4.0e6), ("SQ_INSTS_VALU", 6e6), ("SQ_INSTS_MFMA", 1e6))))
    kc = '"void sgmse::calib_stream_kernel<2, 0>(float*, unsigned long, float*)"'      # the known-size 8 B/lane read stream of the same pass
    ka = '"void sgmse::attn_core_kernel<4>(sgmse::AttnArgs)"'
    (d2 / "1_counter_collection.csv").write_text(rows + "".join(f"{i},{k},FETCH_SIZE,1000000,{1000 * i},{1000 * i + 2000000}\n" for i in (1, 2))
                                                  + f"3,{kc},FETCH_SIZE,{(1 << 30) / 1024 / 2},5000,9000\n")
    (d1 / "2_counter_collection.csv").write_text(rows + "".join(
        f"7,{ka},{n},{v},1000,401000\n" for n, v in (("SQ_VALU_MFMA_BUSY_CYCLES", 4.0e8), ("GRBM_GUI_ACTIVE", 8 * 8.0e5))))
    dump = tmp_path / "dump.txt"
    dump.write_text("[sgmse-prof] conv3x3-wino 128->128 @32x256x512 +res +gn 3.4 ms 1.0 Gwork/s\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc_bench.py"), str(tmp_path / "a"), str(tmp_path / "b"),
                          "--dump", str(dump), "--meta", "command=bench.py --batch 32 --N 2", "workload=pc16k", "batch=32", "commit=abc1234",
                          "source_hash=0123"], capture_output=True, text=True, check=True).stdout
    d = json.loads(out)
    assert d["_meta"] == {"command": "bench.py --batch 32 --N 2", "workload": "pc16k", "batch": 32, "commit": "abc1234", "source_hash": "0123"}
    assert abs(d["_calibration"]["fetch_8B_per_lane"]["factor"] - 2.0) < 1e-9       # the counter reported half of the known gibibyte
    e = d["sgmse::conv3x3_wino_kernel<8, 1, 0, 0, 0>"]
    assert e["launches"] == 2 and abs(e["mfma_busy"] - 2.0e9 / (4.0e6 * 1024)) < 1e-9 and abs(e["valu_per_mfma"] - 6.0) < 1e-9
    assert e["fetch_bytes_per_launch_raw"] == 1000000 * 1024.0 and e["fetch_bytes_per_launch"] == 2.0 * 1000000 * 1024.0
    assert abs(e["effective_clock_ghz"] - 2.0) < 1e-9
    att = d["sgmse::attn_core_kernel<4>"]
    assert abs(att["mfma_busy"] - 4.0e8 / (8.0e5 * 1024)) < 1e-9
    # bench.py applies such a file: calibrated traffic, its provenance, the attention row
    import shutil
    prof = tmp_path / "root" / "profiles"
    prof.mkdir(parents=True)
    (prof / "r99_pmc_bench_b32.json").write_text(out)
    saved = bench.ROOT
    try:
        bench.ROOT = str(tmp_path / "root")
        os.makedirs(os.path.join(bench.ROOT, "sgmse_amd", "csrc"))
        got = bench.pmc_of_the_bench_run(prefix, 32, "pc16k")
    finally:
        bench.ROOT = saved
    assert got["traffic"] == 2.0 * 1000000 * 1024.0 and got["fetch_calibration"]["fetch_8B_per_lane"]["factor"] == 2.0
    assert "abc1234" in got["pmc_source"] and got["pmc_matches_this_build"] is False
    assert abs(got["attention"]["mfma_busy"] - 4.0e8 / (8.0e5 * 1024)) < 1e-9
    assert e["algorithmic_bytes_per_launch"] == 4.0 * 32 * 256 * 512 * (128 + 128 + 128) + 4.0 * 128 * 128 * 9

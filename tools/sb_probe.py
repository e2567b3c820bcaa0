import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import parity as P
from conftest import rel_l2
dev = "cuda"
def run(tag):
    for stype in ("ode", "sde"):
        z = P.load(f"sb_{stype}_N4")
        cfg = P.NO.NetCfg.for_variant("ncsnpp_v2", nf=32)
        m, _ = P.make_model(cfg, dev, sde="sbve", k=2.6, c=0.4, N=4, loss_type="data_prediction")
        y, ref = torch.from_numpy(z["y"]), torch.from_numpy(z["out"])
        noise = P.replay_noise(y.shape, 4).to(dev) if stype == "sde" else None
        out, n = m.get_sb_sampler(m.sde, y.to(dev), sampler_type=stype, n_steps=4, noise=noise)()
        print(tag, stype, rel_l2(out.cpu(), ref), flush=True)
run("default")
os.environ["SGMSE_FIR_SCALAR"] = "1"; run("fir_scalar"); del os.environ["SGMSE_FIR_SCALAR"]
os.environ["SGMSE_TILE_MIN_BLOCKS"] = "1024"; run("min1024"); os.environ["SGMSE_TILE_MIN_BLOCKS"] = "1"; run("min1")

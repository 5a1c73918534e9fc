"""Distribution of the scaled error of GPU and fp32-oracle against the fp64 oracle (teacher-forced single steps)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import oracle as O
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, AirbotTShape, domain_randomize
from test_parity_gpu import _np, _push, _scaled_err

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    tshape = len(sys.argv) > 2 and sys.argv[2] == "tshape"
    envdef = AirbotTShape() if tshape else AirbotPlayBase()
    dr = None if tshape else domain_randomize(envdef.sys, prng.split(prng.PRNGKey(5), n))
    env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
    orc = O.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
    o64 = O.Oracle(env.blob, "f64"); o64.set_ncon_cap(env.dims.ncon_max)
    keys = prng.split(prng.PRNGKey(0), n)
    st = orc.new_state(n, dr); orc.reset(st, keys); env.reset(keys)
    rng = np.random.default_rng(0)
    for depth in (0, 7, 53):
        for _ in range(depth):
            orc.step(st, np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32))
        if tshape:
            for k in st:
                if k in env._views and st[k] is not None and k != "stats" and env.view(k).numel() > 0:     # (skip fields of other env kinds)
                    env.view(k).copy_(torch.from_numpy(st[k].reshape(n, -1)))
        else:
            _push(env, st)
        st64 = {k: (v.copy() if v is not None else None) for k, v in st.items()}
        act = np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32)
        orc.step(st, act); o64.step(st64, act); env.step(None, act); torch.cuda.synchronize()
        print(f"=== +{depth} steps ===")
        for k in ("qpos", "qvel", "qacc_warmstart", "xpos", "obs", "reward"):
            g = _np(env, k, st[k])
            eg, ec, egc = _scaled_err(g, st64[k]), _scaled_err(st[k], st64[k]), _scaled_err(g, st[k])
            q = lambda e: " ".join(f"{np.quantile(e, p):.1e}" for p in (0.5, 0.9, 0.99, 0.999, 1.0))
            print(f"{k:16s} gpu-f64 [{q(eg)}]  cpu32-f64 [{q(ec)}]  gpu-cpu32 [{q(egc)}]  frac>1e-5: gpu-cpu32 {np.mean(egc>1e-5):.4f} cpu32-f64 {np.mean(ec>1e-5):.4f}")

if __name__ == "__main__":
    main()

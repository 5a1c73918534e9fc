"""Quick parity statistics + throughput probe on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import oracle as O
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize
from tools.gpu_debug import SHARED, to_np, push

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    envdef = AirbotPlayBase()
    keys = prng.split(prng.PRNGKey(0), n)
    dr = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(5), n))
    env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
    orc = O.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
    st = orc.new_state(n, dr)
    orc.reset(st, keys)
    s = env.reset(keys); torch.cuda.synchronize()
    rng = np.random.default_rng(0)
    for warm in (0, 10, 40):
        for _ in range(warm):
            orc.step(st, np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32))
        push(env, st)
        act = np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32)
        orc.step(st, act); env.step(s, act); torch.cuda.synchronize()
        print(f"--- teacher-forced step after +{warm} oracle steps (n={n}, n_frames=4, DR on) ---")
        for k in ("qpos", "qvel", "qacc_warmstart", "xpos", "obs", "reward", "done", "metrics", "info_new_cube_pos"):
            a, b = to_np(env, k, st).reshape(n, -1).astype(np.float64), st[k].reshape(n, -1).astype(np.float64)
            scale = np.maximum(1.0, np.abs(b).max(axis=1, keepdims=True))
            err = (np.abs(a - b) / scale).max(axis=1)
            print(f"{k:20s} scaled err: max {err.max():.2e}  p99 {np.quantile(err,0.99):.2e}  median {np.median(err):.2e}  frac>1e-5 {np.mean(err>1e-5):.4f}")
        print("ncon max", st["stats"][:, 2].max(), "drops", st["stats"][:, 3].sum(), "gpu drops", int(env.view("stats")[:, 3].sum()))
    # throughput probe
    for nn in (1024, 8192):
        e2 = envdef.batched(nn, episode_length=1200, auto_reset=True)
        k2 = prng.split(prng.PRNGKey(1), nn)
        s2 = e2.reset(k2)
        a = torch.clamp(torch.randn(nn, 5, device="cuda"), -1, 1)
        for _ in range(10): e2.step(s2, a)
        torch.cuda.synchronize(); t = time.time()
        K = 50
        for _ in range(K):
            a = torch.clamp(torch.randn(nn, 5, device="cuda"), -1, 1)
            e2.step(s2, a)
        torch.cuda.synchronize(); dt = time.time() - t
        print(f"N={nn}: {K} steps in {dt*1e3:.1f} ms -> {nn*K/dt:.3e} env-steps/s, {dt/K*1e3:.3f} ms/step")
        st_ = e2.view("stats").cpu().numpy()
        print("  stats mean [niter ls ncon drop]", st_.mean(axis=0))

if __name__ == "__main__":
    main()

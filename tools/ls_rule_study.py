"""CPU study of line-search stop rules on the fp32 oracle (test infrastructure): iterations per call and the distance of
the solver's qacc from the fp64 oracle's, for MJX's rule and the noise-floor variants (oracle.set_ls_rule).
usage: python tools/ls_rule_study.py [--tshape] [--envs N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs import config
from rsr_mjx_amd.mjcf import CompiledModel
from rsr_mjx_amd.model import model_fields, pack_blob

n = int(sys.argv[sys.argv.index("--envs") + 1]) if "--envs" in sys.argv else 256
tshape = "--tshape" in sys.argv
model = CompiledModel.load(os.path.join(ROOT, "rsr_mjx_amd", "assets", "airbot_tshape.npz" if tshape else "airbot_cube.npz"))
f = model_fields(model)
f.update((config.tshape_env_fields if tshape else config.cube_env_fields)(model))
blob = pack_blob(f)
O.build()
o32, o64 = O.Oracle(blob, "f32"), O.Oracle(blob, "f64")
st = o32.new_state(n)
o32.set_ls_rule(0)
o32.reset(st, prng.split(prng.PRNGKey(0), n))
rng = np.random.default_rng(0)
samples = [(st["qpos"].copy(), st["qvel"].copy(), st["ctrl"].copy(), st["qacc_warmstart"].copy())]      # straight after reset
for t in range(41):
    act = np.clip(rng.normal(size=(n, o32.nu)), -1, 1).astype(np.float32)
    o32.step(st, act)
    if t % 10 == 0:
        samples.append((st["qpos"].copy(), st["qvel"].copy(), st["ctrl"].copy(), st["qacc_warmstart"].copy()))
variants = [("mjx", 0, 1.0, 0, 0), ("mjx + cycle cut", 0, 1.0, 1, 0), ("floor x1 (kernel r1)", 1, 1.0, 0, 0), ("floor x1 + cycle cut", 1, 1.0, 1, 0),
            ("floor x1 |d| + cycle", 2, 1.0, 1, 0), ("floor x0.3 |d| + cycle", 2, 0.3, 1, 0), ("floor x0.1 |d| + cycle", 2, 0.1, 1, 0),
            ("floor x0.03 |d| + cycle", 2, 0.03, 1, 0), ("floor x0.01 |d| + cycle", 2, 0.01, 1, 0), ("floor x0.1 signed + cycle", 1, 0.1, 1, 0)]
truth = []
for (Q, V, U, W) in samples:
    for e in range(n):
        o64.forward(Q[e], V[e], U[e], W[e])
        truth.append(o64.get("qacc"))
truth = np.array(truth)
scale = np.maximum(1.0, np.abs(truth).max(axis=1))
print(f"{len(truth)} forward passes; |qacc|_inf median {np.median(np.abs(truth).max(axis=1)):.1f}")
base = None
for name, rule, noise, cyc, nn in variants:
    o32.set_ls_rule(rule, noise); o32.set_ls_cycle(cyc)
    o32.ls_counters(reset=True)
    got, niter = [], []
    for (Q, V, U, W) in samples:
        for e in range(n):
            o32.forward(Q[e], V[e], U[e], W[e])
            got.append(o32.get("qacc")); niter.append(o32.get("counts")[4])
    got = np.array(got)
    calls, iters, hist = o32.ls_counters()
    err = np.abs(got - truth).max(axis=1) / scale
    if base is None:
        base = got
    dif = np.abs(got - base).max(axis=1) / scale
    print(f"{name:20s} ls iters/call {iters / calls:6.2f} (calls {calls}, >=20 iters: {hist[20:].sum() / calls * 100:5.2f} %, max bin {np.nonzero(hist)[0].max()})  newton {np.mean(niter):.2f}"
          f" | err vs f64: median {np.median(err):.2e} p90 {np.percentile(err, 90):.2e} p99 {np.percentile(err, 99):.2e} max {err.max():.2e}"
          f" | vs mjx-f32: p99 {np.percentile(dif, 99):.2e} max {dif.max():.2e}")

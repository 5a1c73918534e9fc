"""Where do the HIP stepper and the fp32 oracle part on the T-shape env-step straight after reset?  One physics substep per
env-step (n_frames = 1), the kernel's stage dump against the oracle's intermediates, for the worst envs (debugging aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import oracle as O
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotTShape

np.set_printoptions(precision=6, suppress=True, linewidth=220)
n = 2048
env = AirbotTShape(n_frames=1).batched(n, episode_length=1000, auto_reset=True)
orc = O.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
o64 = O.Oracle(env.blob, "f64"); o64.set_ncon_cap(env.dims.ncon_max)
keys = prng.split(prng.PRNGKey(0), n)
st = orc.new_state(n); orc.reset(st, keys); s = env.reset(keys)
pre = {k: (v.copy() if v is not None else None) for k, v in st.items()}
dbg = env.enable_debug(True)
act = np.clip(np.random.default_rng(0).normal(size=(n, 5)), -1, 1).astype(np.float32)
orc.step(st, act); env.step(s, act); torch.cuda.synchronize()
g = lambda k: env.view(k).cpu().numpy().reshape(st[k].shape)
err = np.abs(g("qvel") - st["qvel"]).max(axis=1)
print("qvel abs err quantiles (one substep after reset):", np.quantile(err, [0.5, 0.9, 0.99, 0.999, 1.0]))
print("stats gpu [niter ls ncon drop] mean", env.view("stats").float().mean(0).tolist(), " cpu", st["stats"].mean(0).tolist())
d = dbg.cpu().numpy()
nv, nb = env.dims.nv, env.dims.nbody
rep = lambda tag, a, b: print(f"   {tag:26s} max_abs {np.abs(np.asarray(a, np.float64) - b).max():.3e}   scale {np.abs(b).max():.3e}")
for e in np.argsort(-err)[:3]:
    print(f"-- env {e}: qvel err {err[e]:.3e} --")
    nefc = orc.forward(pre["qpos"][e], pre["qvel"][e], st["ctrl"][e], pre["qacc_warmstart"][e])
    cnt = orc.get("counts")
    print("   counts cpu [nefc ne nf ncon niter ls]", cnt.tolist(), " gpu [nefc neq nf ncon niter ls nlim drop]", d[e, :8].tolist())
    rep("xpos", d[e, 16:16 + 3 * nb], orc.get("xpos"))
    rep("M", d[e, 128:128 + nv * nv], orc.get("M"))
    rep("qfrc_smooth", d[e, 736:736 + nv], orc.get("qfrc_smooth"))
    rep("qacc_smooth", d[e, 768:768 + nv], orc.get("qacc_smooth"))
    ncon = int(cnt[3])
    if ncon == int(d[e, 3]):
        con = orc.get("contacts").reshape(-1, 10)
        gg = d[e, 864:864 + 8 * ncon].reshape(-1, 8)
        rep("contact dist", gg[:, 0], con[:, 0]); rep("contact pos", gg[:, 1:4], con[:, 1:4]); rep("contact normal", gg[:, 4:7], con[:, 4:7])
        print("   contact pairs gpu", gg[:, 7].astype(int).tolist(), "cpu", con[:, 9].astype(int).tolist(), " dist", con[:, 0])
    else:
        print("   CONTACT COUNT DIFFERS")
    if int(cnt[0]) == int(d[e, 0]):
        rep("efc_aref", d[e, 1152:1152 + nefc], orc.get("efc_aref")); rep("efc_D", d[e, 1408:1408 + nefc], orc.get("efc_D"))
        rep("efc_J", d[e, 2048:2048 + nefc * nv], orc.get("efc_J"))
    qa32 = orc.get("qacc")
    print("   gpu start costs [smooth warm chosen]", d[e, 7400:7403].tolist())
    for it in range(int(d[e, 4])):
        print(f"   gpu iter {it} cost {d[e, 7410 + 4 * it]:.9g} alpha {d[e, 7411 + 4 * it]:.9g} ls {int(d[e, 7412 + 4 * it])} p0.d0 {d[e, 7413 + 4 * it]:.6g}")
    sys.stdout.flush(); os.environ["RSR_SOLVER_TRACE"] = "1"
    orc.forward(pre["qpos"][e], pre["qvel"][e], st["ctrl"][e], pre["qacc_warmstart"][e])
    o64.forward(pre["qpos"][e], pre["qvel"][e], st["ctrl"][e], pre["qacc_warmstart"][e])
    del os.environ["RSR_SOLVER_TRACE"]; sys.stderr.flush()
    orc.set_ls_rule(2, 1.0); orc.set_ls_cycle(True)
    orc.forward(pre["qpos"][e], pre["qvel"][e], st["ctrl"][e], pre["qacc_warmstart"][e])
    qa32k = orc.get("qacc"); cntk = orc.get("counts")
    orc.set_ls_rule(0); orc.set_ls_cycle(False)
    o64.forward(pre["qpos"][e], pre["qvel"][e], st["ctrl"][e], pre["qacc_warmstart"][e])
    qa64 = o64.get("qacc")
    rep("qacc vs oracle f32", d[e, 800:800 + nv], qa32)
    rep("qacc vs oracle f64", d[e, 800:800 + nv], qa64)
    print(f"   oracle f32 with the kernel's stop rules: counts {cntk.tolist()}, vs f64 max_abs {np.abs(qa32k - qa64).max():.3e}, vs gpu {np.abs(qa32k - d[e, 800:800 + nv]).max():.3e}")
    print(f"   oracle f32 vs f64 qacc max_abs {np.abs(qa32 - qa64).max():.3e}; f64 counts {o64.get('counts').tolist()}")
    print("   qacc gpu", d[e, 800:800 + nv]); print("   qacc f32", qa32); print("   qacc f64", qa64)

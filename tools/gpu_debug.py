"""Stage-by-stage comparison of the HIP step against the CPU oracle (debugging aid, run on the GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402
from rsr_mjx_amd import prng  # noqa: E402
from rsr_mjx_amd.envs.airbot import AirbotPlayBase  # noqa: E402

np.set_printoptions(precision=6, suppress=True, linewidth=200)
SHARED = ["qpos", "qvel", "ctrl", "qacc_warmstart", "time", "xpos", "site_xpos", "obs", "reward", "done", "metrics",
          "info_target_pos", "info_new_cube_pos", "info_site_pos", "info_cube_pos", "info_steps", "info_truncation",
          "info_episode_done", "info_episode_metrics", "first_qpos", "first_qvel", "first_ctrl", "first_warmstart",
          "first_time", "first_xpos", "first_site_xpos", "first_obs"]


def to_np(env, name, st):
    return env.view(name).detach().cpu().numpy().reshape(st[name].shape)


def push(env, st):
    for k in SHARED:
        env.view(k).copy_(torch.from_numpy(st[k].reshape(st[k].shape[0], -1)))


def report(tag, a, b, worst=3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) / (1e-6 + np.abs(b))
    print(f"{tag:28s} max_abs {np.abs(a - b).max():.3e}  max_rel {err.max():.3e}")
    return err


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    warm_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    envdef = AirbotPlayBase(n_frames=nfr)
    env = envdef.batched(n, episode_length=1200, auto_reset=True)
    orc = O.Oracle(env.blob)
    orc.set_ncon_cap(env.dims.ncon_max)
    keys = prng.split(prng.PRNGKey(0), n)
    st = orc.new_state(n)
    orc.reset(st, keys)
    s = env.reset(keys)
    torch.cuda.synchronize()
    print("== reset ==")
    for k in SHARED:
        report(k, to_np(env, k, st), st[k])
    print("stats gpu", env.view("stats")[:4].cpu().numpy().tolist(), "cpu", st["stats"][:4].tolist())
    rng = np.random.default_rng(0)
    for _ in range(warm_steps):
        orc.step(st, np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32))
    push(env, st)
    dbg = env.enable_debug(True)
    act = np.clip(rng.normal(size=(n, 5)), -1, 1).astype(np.float32)
    # oracle stage dump for env 0 before stepping
    q0, v0, w0 = st["qpos"][0].copy(), st["qvel"][0].copy(), st["qacc_warmstart"][0].copy()
    pre = {k: st[k].copy() for k in SHARED}
    orc.step(st, act)
    env.step(s, act)
    torch.cuda.synchronize()
    print(f"== one step (n_frames={nfr}) after {warm_steps} oracle steps ==")
    for k in SHARED:
        report(k, to_np(env, k, st), st[k])
    print("stats gpu", env.view("stats")[:8].cpu().numpy().tolist())
    print("stats cpu", st["stats"][:8].tolist())
    if nfr == 1:
        d = dbg.cpu().numpy()
        nv, nb = env.dims.nv, env.dims.nbody
        worst_env = int(np.argmax(np.abs(to_np(env, "qvel", st) - st["qvel"]).max(axis=1)))
        for e in sorted({0, worst_env}):
            print(f"-- stage dump env {e} --")
            ctrl = st["ctrl"][e]      # post-step ctrl = the clipped ctrl used during the step
            nefc = orc.forward(pre["qpos"][e], pre["qvel"][e], ctrl, pre["qacc_warmstart"][e])
            cnt = orc.get("counts")
            print("counts cpu [nefc ne nf ncon niter ls]", cnt, " gpu", d[e, :8])
            report("xpos", d[e, 16:16 + 3 * nb], orc.get("xpos"))
            report("xquat", d[e, 64:64 + 4 * nb], orc.get("xquat"))
            report("M", d[e, 128:128 + nv * nv], orc.get("M"))
            report("geom_xpos", d[e, 6400:6400 + 3 * env.dims.ngeom], orc.get("geom_xpos"))
            report("cdof", d[e, 6528:6528 + 6 * nv], orc.get("cdof"))
            report("cinert", d[e, 6656:6656 + 10 * nb], orc.get("cinert"))
            report("subtree_com", d[e, 6800:6800 + 3 * nb], orc.get("subtree_com"))
            report("cvel", d[e, 6848:6848 + 6 * nb], orc.get("cvel"))
            report("cdof_dot", d[e, 6944:6944 + 6 * nv], orc.get("cdof_dot"))
            report("qfrc_smooth", d[e, 736:736 + nv], orc.get("qfrc_smooth"))
            report("qacc_smooth", d[e, 768:768 + nv], orc.get("qacc_smooth"))
            ncon = int(cnt[3])
            if ncon == int(d[e, 3]):
                con = orc.get("contacts").reshape(-1, 10)
                g = d[e, 864:864 + 8 * ncon].reshape(-1, 8)
                report("contact dist/pos/normal", g[:, :7], con[:, :7])
                report("contact pair", g[:, 7], con[:, 9])
            if int(cnt[0]) == int(d[e, 0]):
                report("efc_aref", d[e, 1152:1152 + nefc], orc.get("efc_aref"))
                report("efc_D", d[e, 1408:1408 + nefc], orc.get("efc_D"))
                report("efc_J", d[e, 2048:2048 + nefc * nv], orc.get("efc_J"))
            report("qacc", d[e, 800:800 + nv], orc.get("qacc"))
            report("qfrc_constraint", d[e, 832:832 + nv], orc.get("qfrc_constraint"))
            print("qacc gpu", d[e, 800:800 + nv])
            print("qacc cpu", orc.get("qacc"))


if __name__ == "__main__":
    main()

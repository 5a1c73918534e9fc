"""Distribution of the scaled error of the HIP stepper against the fp32 CPU oracle (and of both against the fp64 oracle) on
teacher-forced single env-steps at several rollout depths, for all four workloads.  The numbers are what the per-field
envelopes of tests/parity_envelopes.py are derived from (x3 on the measured maximum / quantiles).
usage: python tools/gpu_parity_stats.py [cube|sf|tshape|go2|go2rough|handstand ...] [--n N] [--json out.json]
err = |a - b| / max(1, |b|_inf of that env's field), per env."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import oracle as O
from rsr_mjx_amd import prng

GNAME = {"priv_obs": "privileged_obs", "first_priv_obs": "first_privileged_obs"}
QS = (0.5, 0.9, 0.99, 0.999, 1.0)


def serr(a, b):
    n = a.shape[0]
    a, b = a.reshape(n, -1).astype(np.float64), b.reshape(n, -1).astype(np.float64)
    return (np.abs(a - b) / np.maximum(1.0, np.abs(b).max(axis=1, keepdims=True))).max(axis=1)


def make(kind, n, variant=0):
    if kind in ("cube", "tshape", "sf"):
        from rsr_mjx_amd.envs.airbot import AirbotPlayBase, AirbotPlaySF, AirbotTShape, domain_randomize
        envdef = AirbotTShape() if kind == "tshape" else (AirbotPlaySF() if kind == "sf" else AirbotPlayBase())
        dr = None if kind != "cube" else domain_randomize(envdef.sys, prng.split(prng.PRNGKey(5), n))
        env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
        odr, nu, astd, depths = dr, 5, 1.0, (0, 2, 7, 23, 53)
        fields = ["qpos", "qvel", "qacc_warmstart", "xpos", "site_xpos", "obs", "reward", "metrics", "ctrl"]
        fields += ["info_xita", "info_new_T_pos", "info_T_pos"] if kind == "tshape" else ["info_new_cube_pos", "info_site_pos", "info_cube_pos"]
    else:
        from rsr_mjx_amd.envs import go2
        if kind == "handstand":
            jenv = go2.load("Go2Footstand" if variant else "Go2Handstand")
        else:
            jenv = go2.load("Go2JoystickRoughTerrain" if kind == "go2rough" else "Go2JoystickFlatTerrain",
                            config_overrides={"pert_config": {"enable": True, "kick_wait_times": [0.1, 0.4], "velocity_kick": [1.0, 4.0]}})
        dr = None if variant else go2.domain_randomize(jenv.sys, prng.split(prng.PRNGKey(12), n))      # (the second variant: nominal model)
        env = go2.wrap_for_brax_training(jenv, n, episode_length=500 if kind == "handstand" else 1000, randomization_fn=(lambda sys: dr) if dr else None)
        odr = None if dr is None else {{"actuator_gainprm": "gainprm", "actuator_biasprm": "biasprm"}.get(k, k): v for k, v in dr.items()}
        nu, astd, depths = 12, (0.3 if kind == "handstand" else 0.5), (0, 2, 5, 17, 40)
        fields = ["qpos", "qvel", "qacc_warmstart", "xpos", "site_xpos", "obs", "reward", "metrics", "priv_obs"]
    return env, odr, nu, astd, depths, fields


def main():
    args = sys.argv[1:]
    n = int(args[args.index("--n") + 1]) if "--n" in args else 8192
    out_json = args[args.index("--json") + 1] if "--json" in args else None
    kinds = [a for a in args if a in ("cube", "sf", "tshape", "go2", "go2rough", "handstand")] or ["cube", "sf", "tshape", "go2", "go2rough", "handstand"]
    result = {}
    for kind, variant in [(k, v) for k in kinds for v in ((0, 1) if k == "handstand" else (0,))]:      # handstand: + Footstand on the nominal model
        env, odr, nu, astd, depths, fields = make(kind, n, variant)
        orc = O.Oracle(env.blob); orc.set_ncon_cap(env.dims.ncon_max)
        o64 = O.Oracle(env.blob, "f64"); o64.set_ncon_cap(env.dims.ncon_max)
        keys = prng.split(prng.PRNGKey(0), n)
        st = orc.new_state(n, odr); orc.reset(st, keys); state = env.reset(keys)
        rng = np.random.default_rng(0)
        res = result.get(kind) or {f: {"gpu_vs_f32": [], "gpu_vs_f32_same_rule": [], "f32_vs_f64": []} for f in fields}
        for depth in depths:
            orc.set_ls_rule(0); orc.set_ls_cycle(False)
            for _ in range(depth):
                orc.step(st, np.clip(rng.normal(size=(n, nu)) * astd, -1, 1).astype(np.float32))
            for k in st:
                v = GNAME.get(k, k)
                if v in env._views and st[k] is not None and k != "stats" and env.view(v).numel() > 0:
                    env.view(v).copy_(torch.from_numpy(st[k].reshape(n, -1)))
            copy = lambda s: {k: (v.copy() if v is not None else None) for k, v in s.items()}
            st64, st_same = copy(st), copy(st)
            act = np.clip(rng.normal(size=(n, nu)) * astd, -1, 1).astype(np.float32)
            orc.step(st, act); o64.step(st64, act)
            orc.set_ls_rule(2, 1.0); orc.set_ls_cycle(True)            # the kernel's line-search stop rules
            orc.step(st_same, act)
            env.step(state, act); torch.cuda.synchronize()
            print(f"=== {kind}{' (Footstand, nominal model)' if variant else ''}: +{depth} steps, {n} envs ===")
            for f in fields:
                g = env.view(GNAME.get(f, f)).cpu().numpy().reshape(st[f].shape)
                e1, e2, e3 = serr(g, st[f]), serr(g, st_same[f]), serr(st[f], st64[f])
                q = lambda e: " ".join(f"{np.quantile(e, p):.1e}" for p in QS)
                print(f"{f:18s} gpu-cpu32 [{q(e1)}] >1e-5: {np.mean(e1 > 1e-5):.4f} | same LS rule [{q(e2)}] >1e-5: {np.mean(e2 > 1e-5):.4f} | cpu32-f64 [{q(e3)}] >1e-5: {np.mean(e3 > 1e-5):.4f}")
                for name, e in (("gpu_vs_f32", e1), ("gpu_vs_f32_same_rule", e2), ("f32_vs_f64", e3)):
                    res[f][name].append({"depth": depth, "q": [float(np.quantile(e, p)) for p in QS], "frac_gt_1e-5": float(np.mean(e > 1e-5))})
        result[kind] = res
        del env
    if out_json:
        json.dump({"n": n, "quantiles": QS, "stats": result}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()

"""Bit-for-bit comparison of two builds of the library on rollouts of the four workloads (an optimisation that reorders no
arithmetic must not move a bit): python tools/gpu_bitcompare.py libA.so libB.so[@VAR=val,...] [--workloads cube,tshape,go2,go2rough,handstand] [--steps 40]
(`@VAR=val` sets environment variables for that arm, e.g. RSR_WHOLE_ENVS=0 RSR_UNITS=5: the same build under another schedule.)  Each build runs in its own process (the library path is read once per process) and dumps the records after the rollout."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(workloads, steps, out):
    import numpy as np, torch
    from rsr_mjx_amd import prng
    res = {}
    for wl in workloads:
        n = 4096
        if wl in ("cube", "tshape"):
            from rsr_mjx_amd.envs.airbot import AirbotPlayBase, AirbotTShape, domain_randomize
            envdef = AirbotTShape() if wl == "tshape" else AirbotPlayBase()
            dr = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(5), n)) if wl == "cube" else None
            env = envdef.batched(n, episode_length=25, auto_reset=True, randomization=dr)
            nu, astd = 5, 1.0
        else:
            from rsr_mjx_amd.envs import go2
            jenv = go2.load("Go2Handstand" if wl == "handstand" else ("Go2JoystickRoughTerrain" if wl == "go2rough" else "Go2JoystickFlatTerrain"))
            dr = go2.domain_randomize(jenv.sys, prng.split(prng.PRNGKey(12), n))
            env = go2.wrap_for_brax_training(jenv, n, episode_length=25, randomization_fn=lambda sys: dr)
            nu, astd = 12, 0.5
        s = env.reset(prng.split(prng.PRNGKey(3), n))
        gen = torch.Generator(device="cuda"); gen.manual_seed(7)
        acts = torch.clamp(torch.randn((steps, n, nu), generator=gen, device="cuda") * astd, -1, 1)
        for t in range(steps):
            s = env.step(s, acts[t])
        torch.cuda.synchronize()
        res[wl] = env.record.view(torch.int32).cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2].split(","), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    import numpy as np
    args = sys.argv[1:]
    wls = args[args.index("--workloads") + 1] if "--workloads" in args else "cube,tshape,go2,go2rough"
    steps = args[args.index("--steps") + 1] if "--steps" in args else "40"
    libs = [a for a in args if a.partition("@")[0].endswith(".so")]
    outs = []
    for lib in libs:
        out = tempfile.mktemp(suffix=".npz")
        name, _, kvs = lib.partition("@")
        env = dict(os.environ, RSR_MJX_LIB=os.path.join(ROOT, "rsr_mjx_amd", "csrc", name))
        for kv in filter(None, kvs.split(",")):
            k, _, v = kv.partition("="); env[k] = v
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", wls, steps, out], env=env)
        outs.append(np.load(out))
    ok = True
    for wl in wls.split(","):
        a = outs[0][wl]
        for lib, o in zip(libs[1:], outs[1:]):
            same = np.array_equal(a, o[wl])
            ok &= same
            extra = "" if same else f" ({int((a != o[wl]).any(axis=1).sum())} of {a.shape[0]} envs differ)"
            print(f"{wl}: {libs[0]} vs {lib}: {'bit-identical' if same else 'DIFFERENT'}{extra}")
    sys.exit(0 if ok else 1)

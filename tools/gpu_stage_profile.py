"""Per-stage cycle shares of the step kernel from the -DRSR_PROFILE diagnostic build (s_memtime stamps).
Never quote this build's run time: read the shares only."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "rsr_mjx_amd", "csrc")
PROF = os.path.join(CSRC, "librsrmjx_prof.so")
if not os.path.exists(PROF) or "--rebuild" in sys.argv:
    from rsr_mjx_amd.build import compile_lib
    compile_lib(PROF, extra_flags=["-DRSR_PROFILE"])
os.environ["RSR_MJX_LIB"] = PROF
import numpy as np
import torch
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize

NAMES = ["load", "kinematics", "com_crb_mass", "collision", "constraint_rows", "smooth_forces", "chol_M+solve",
         "solver_init", "hessian: solve only", "linesearch: p0, lo, iterations", "update_constraint", "integrate", "epilogue+store",
         "hessian: weights+compaction", "hessian: sparse rows", "hessian: contacts", "hessian: block exchange", "hessian: cholesky",
         "linesearch: setup (jdot, M.v, sums)", "x0 collision: SAT / primitives", "x1 collision: clip slots", "x2 collision: compaction",
         "x3 rows: limits, zeroing, sparse", "x4 rows: contact base rows", "x5 rows: friction coefficients", "x6 update: row forces and costs of the new point (rest of update = J^T f, gradient, the sums)", "integrate: factor of M + h D, solve (rest of integrate = advance)",
         "ls: prepare (row pieces)", "ls: point alpha=0", "ls: first Newton point", "ls: iterations", "solver exit: final costs and step of a one-iteration solve, exit test", "init: M.a, J.a, costs of both starts", "init: J^T f, gradient",
         "kin: record loads", "kin: local transforms (sincos)", "kin: tree composition", "kin: body stores + barrier",
         "crb: record loads", "crb: subtree com", "crb: cinert, cdof, M zero", "crb: composite inertia",
         "smooth: record loads", "smooth: cvel, cdof_dot", "smooth: cacc, cfrc", "smooth: subtree force sum",
         "go2 epilogue: sensors, accelerometer", "go2 epilogue: IMU FIFOs", "go2 epilogue: foot contacts", "go2 epilogue: obs + noise draws",
         "go2 epilogue: privileged obs", "go2 epilogue: reward terms", "go2 epilogue: bookkeeping, command draws (rest = stores)"]
n = 8192
if "--go2" in sys.argv:
    from rsr_mjx_amd.envs import go2
    env = go2.load("Go2JoystickFlatTerrain").batched(n, episode_length=1000, auto_reset=True)
    nu, astd = 12, 0.3
elif "--handstand" in sys.argv:
    from rsr_mjx_amd.envs import go2
    env = go2.load("Go2Handstand").batched(n, episode_length=500, auto_reset=True)
    nu, astd = 12, 0.3
elif "--tshape" in sys.argv:
    from rsr_mjx_amd.envs.airbot import AirbotTShape
    envdef = AirbotTShape()
    env = envdef.batched(n, episode_length=1200, auto_reset=True)
    nu, astd = 5, 1.0
else:
    envdef = AirbotPlayBase()
    dr = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(1), n))
    env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
    nu, astd = 5, 1.0
if hasattr(env, "set_schedule"):
    env.set_schedule(1)          # per-stage counters are written by the unit that ends the env-step: keep the step in one unit
s = env.reset(prng.split(prng.PRNGKey(0), n))
dbg = env.enable_debug(True)
tot = np.zeros(len(NAMES))
for t in range(40):
    env.step(s, torch.clamp(torch.randn(n, nu, device="cuda") * astd, -1, 1))
    if t >= 10:
        torch.cuda.synchronize()
        tot += dbg[:, 7200:7200 + len(NAMES)].double().mean(dim=0).cpu().numpy()
tot /= 30
print(f"mean cycles per env-step (wave lifetime, all substeps): {tot.sum():.0f}")
for nm, v in zip(NAMES, tot):
    print(f"  {nm:24s} {v:10.0f}  {100 * v / tot.sum():5.1f} %")
print("stats mean [niter ls ncon drop]", env.view("stats").float().mean(dim=0).tolist())

"""Wave timeline of one step launch (-DRSR_TIMELINE build: start / end stamps per wave, no per-stage stamps).
Prints: launch makespan, wave-lifetime distribution, residency (waves in flight over time), lifetime vs ncon / solver work.
usage: python tools/gpu_wave_timeline.py [--envs N] [--go2] [--tshape] [--rebuild] [extra -D flags ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "rsr_mjx_amd", "csrc")
LIB = os.path.join(CSRC, "librsrmjx_timeline.so")
extra = [a for a in sys.argv[1:] if a.startswith("-D") or a.startswith("-m")]
if not os.path.exists(LIB) or "--rebuild" in sys.argv:
    from rsr_mjx_amd.build import compile_lib
    compile_lib(LIB, extra_flags=["-DRSR_TIMELINE"] + extra)
os.environ["RSR_MJX_LIB"] = LIB
import numpy as np
import torch
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize

n = int(sys.argv[sys.argv.index("--envs") + 1]) if "--envs" in sys.argv else 8192
if "--go2" in sys.argv:
    from rsr_mjx_amd.envs import go2
    env = go2.load("Go2JoystickFlatTerrain").batched(n, episode_length=1000, auto_reset=True)
    nu, astd = 12, 0.3
else:
    envdef = AirbotPlayBase()
    dr = domain_randomize(envdef.sys, prng.split(prng.PRNGKey(1), n))
    env = envdef.batched(n, episode_length=1200, auto_reset=True, randomization=dr)
    nu, astd = 5, 1.0
units = int(sys.argv[sys.argv.index("--units") + 1]) if "--units" in sys.argv else 1
if hasattr(env, "set_schedule"):
    env.set_schedule(units)
s = env.reset(prng.split(prng.PRNGKey(0), n))
dbg = env.enable_debug(True)
for t in range(60):
    env.step(s, torch.clamp(torch.randn(n, nu, device="cuda") * astd, -1, 1))
torch.cuda.synchronize()
whole = int(sys.argv[sys.argv.index("--whole") + 1]) if "--whole" in sys.argv else None
if whole is not None:
    env.set_whole_envs(whole)
    dbg.zero_()
    for t in range(3):
        env.step(s, torch.clamp(torch.randn(n, nu, device="cuda") * astd, -1, 1))
    torch.cuda.synchronize()
raw = dbg[:, 7300:7300 + 8 * units].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
raw = raw.reshape(n, units, 8).transpose(1, 0, 2).reshape(n * units, 8)        # unit-major: all phase-0 units, then phase 1, ...
rt0 = raw[:, 0] | (raw[:, 1] << 32); rt1 = raw[:, 2] | (raw[:, 3] << 32)
cyc = raw[:, 4]; hw = raw[:, 5]; xcc = raw[:, 6] & 0xF
if whole is not None:
    # envs stepped as one unit write slot 0 only: keep the units that ran in the last launch
    ran = rt0 >= rt0[rt0 > 0].max() - 300000            # within 3 ms of the latest start
    n_whole = whole if whole >= 0 else max(0, n - 2 * 2048)       # the library's default: all but two resident rounds (8 waves x 256 CUs)
    is_whole = np.tile(np.arange(n) < n_whole, units)
    for name, sel in (("whole-step units", ran & is_whole), ("split units", ran & ~is_whole)):
        if sel.any():
            print(f"{name}: {int(sel.sum())}, lifetime mean {((rt1[sel] - rt0[sel]) / 100.0).mean():.1f} us p99 {np.percentile((rt1[sel] - rt0[sel]) / 100.0, 99):.1f}, start {((rt0[sel] - rt0[ran].min()) / 100.0).min():.1f}..{((rt0[sel] - rt0[ran].min()) / 100.0).max():.1f} us")
    rt0, rt1, cyc, hw, xcc = rt0[ran], rt1[ran], cyc[ran], hw[ran], xcc[ran]
    stats_sel = ran
elif units > 1:
    stats_sel = None
    for ph in range(units):
        sl = slice(ph * n, (ph + 1) * n)
        print(f"phase {ph}: unit start {((rt0[sl] - rt0.min()) / 100.0).min():.1f}..{((rt0[sl] - rt0.min()) / 100.0).max():.1f} us, lifetime mean {((rt1[sl] - rt0[sl]) / 100.0).mean():.1f} us, cycles mean {cyc[sl].mean():.0f}")
    gap = (rt0[n:2 * n] - rt1[:n]) / 100.0
    print(f"phase 1 start minus phase 0 end of the same env: min {gap.min():.1f} p50 {np.median(gap):.1f} max {gap.max():.1f} us")
else:
    stats_sel = None
t0 = (rt0 - rt0.min()) / 100.0; t1 = (rt1 - rt0.min()) / 100.0      # microseconds (100 MHz)
life = t1 - t0
print(f"envs {n}: makespan {t1.max():.1f} us; first start spread {t0.min():.1f}..{np.percentile(t0, 25):.1f} us (25% of waves)")
print("wave lifetime us: min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f mean %.1f" % (life.min(), *np.percentile(life, [10, 50, 90, 99]), life.max(), life.mean()))
print("wave shader cycles: p50 %.0f mean %.0f max %.0f ; implied clock GHz p50 %.2f" % (np.median(cyc), cyc.mean(), cyc.max(), np.median(cyc / life) / 1e3))
stats = np.tile(env.view("stats").cpu().numpy(), (units, 1))
if stats_sel is not None:
    stats = stats[stats_sel]
ncon = stats[:, 2]
for lo, hi in ((0, 4), (5, 8), (9, 12), (13, 16), (17, 99)):
    sel = (ncon >= lo) & (ncon <= hi)
    if sel.any():
        print(f"  ncon {lo:2d}..{hi:2d}: {sel.mean() * 100:5.1f} % of envs, lifetime mean {life[sel].mean():7.1f} us, newton iters (last substep) {stats[sel, 0].mean():.2f}, ls iters {stats[sel, 1].mean():.1f}")
# residency over time
grid = np.linspace(0, t1.max(), 41)
res = [(np.sum((t0 <= g) & (t1 > g))) for g in grid]
print("waves in flight at 2.5% steps of the makespan:", " ".join(str(r) for r in res))
order = np.argsort(t0)
print("start time of wave rank 2048/4096/6144/8191: " + " ".join(f"{t0[order[min(k, len(t0) - 1)]]:.1f}" for k in (2048, 4096, 6144, 8191)))
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; se = (hw >> 13) & 7
slot = ((xcc * 8 + se) * 16 + cu) * 4 + simd
u, c = np.unique(slot, return_counts=True)
print(f"distinct (xcc, se, cu, simd) slots used: {len(u)}; waves per SIMD over the launch: min {c.min()} max {c.max()} mean {c.mean():.2f}")
busy = np.zeros(len(u))
for i, sl in enumerate(u):
    busy[i] = life[slot == sl].sum()
print("per-SIMD sum of wave lifetimes / (2 x makespan): mean %.3f min %.3f max %.3f" % ((busy / (2 * t1.max())).mean(), (busy / (2 * t1.max())).min(), (busy / (2 * t1.max())).max()))
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "timeline.npz"), t0=t0, t1=t1, cyc=cyc, hw=hw, xcc=xcc, stats=stats)

if units > 1 or whole is not None:
    sys.exit(0)
# ---- how well does the previous step's lifetime predict this step's (longest-first dispatch keyed on it)?
import heapq
def _life():
    r = dbg[:, 7300:7307].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    return ((r[:, 2] | (r[:, 3] << 32)) - (r[:, 0] | (r[:, 1] << 32))) / 100.0, r[:, 4]
def _sched(L, slots=2048):
    h = [0.0] * slots; heapq.heapify(h)
    for l in L:
        heapq.heappush(h, heapq.heappop(h) + l)
    return max(h)
prev, prev_cyc = _life()
env.step(s, torch.clamp(torch.randn(n, nu, device="cuda") * astd, -1, 1)); torch.cuda.synchronize()
cur, cur_cyc = _life()
print(f"lifetime correlation between consecutive steps: {np.corrcoef(prev, cur)[0, 1]:.3f} (cycles: {np.corrcoef(prev_cyc, cur_cyc)[0, 1]:.3f})")
print(f"simulated makespan of this step's lifetimes on 2048 slots: index order {_sched(cur):.0f} us, longest-first by own lifetime {_sched(np.sort(cur)[::-1]):.0f}, "
      f"longest-first by the previous step's lifetime {_sched(cur[np.argsort(-prev)]):.0f}, by previous cycles {_sched(cur[np.argsort(-prev_cyc)]):.0f}, ideal {cur.sum() / 2048:.0f}")

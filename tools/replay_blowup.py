"""Replay a saved pre-instability state (tools/gpu_blowup_hunt.py) on the CPU oracle, fp32 and fp64, for a few steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_blob
from oracle import oracle as O
from rsr_mjx_amd.mjcf import CompiledModel
np.set_printoptions(precision=4, suppress=True, linewidth=220)
m = CompiledModel.load(os.path.join(ROOT, "rsr_mjx_amd", "assets", "airbot_cube.npz"))
blob = make_blob(m, "cube", episode_length=1200, auto_reset=True)
sizes = [("qpos", 22), ("qvel", 20), ("ctrl", 5), ("qacc_warmstart", 20), ("time", 1), ("xpos", 42), ("site_xpos", 3)]
off, o = {}, 0
for k, w in sizes: off[k] = (o, w); o += w
for k, w in sizes: off["first_" + k.replace("qacc_warmstart", "warmstart")] = (o, w); o += w
for k, w in (("obs", 23), ("first_obs", 23), ("reward", 1), ("done", 1), ("metrics", 3), ("info_target_pos", 3), ("info_new_cube_pos", 2),
             ("info_site_pos", 3), ("info_cube_pos", 3), ("info_last_action", 1), ("info_target_base_pos", 3), ("info_target_vertical_pos", 3),
             ("info_target_w", 1), ("info_new_T_pos", 2), ("info_T_pos", 3), ("info_xita", 1), ("info_steps", 1), ("info_truncation", 1),
             ("info_episode_done", 1), ("info_episode_metrics", 5)):
    off[k] = (o, w); o += w
for f in sys.argv[1:]:
    z = np.load(f)
    print("==", f, "step", int(z["step"]), "env", int(z["env"]), "action", z["action"], "prev action", z["prev_action"])
    for which, acts in (("record2", [z["prev_action"], z["action"]] + [z["action"]] * 4),):
        rec = z[which]
        print(" from", which, ": arm qpos", rec[0:8], "\n   arm qvel", rec[22:30], "ctrl", rec[42:47])
        for prec in ("f32", "f64"):
            orc = O.Oracle(blob, prec); orc.set_ncon_cap(24)
            dr = {k: z[k][None] for k in ("geom_friction", "body_mass", "dof_damping", "dof_frictionloss")}
            st = orc.new_state(1, dr)
            for k, (a, w) in off.items():
                if k in st and st[k] is not None:
                    st[k][...] = rec[a:a + w].reshape(st[k].shape)
            line = []
            for a in acts:
                orc.step(st, a[None])
                line.append(f"{np.abs(st['qvel'][0, :8]).max():.1f}/it{st['stats'][0][0]}/ls{st['stats'][0][1]}/c{st['stats'][0][2]}")
            print("  ", prec, "max|arm qvel| / newton iters / ls / ncon per step:", "  ".join(line))
    print("  GPU: after step", np.abs(z["after"][22:30]).max(), "before", np.abs(z["record"][22:30]).max())

"""Cube-env poses with more active contacts than the kernel's capacity (24): the arm folded into the table top, found by
random search on the fp64 oracle with the capacity lifted.  Written to tests/golden/cube_overcap_states.npz (inputs only:
qpos and the oracle's uncapped contact count); tests/test_parity_gpu.py::test_contact_capacity_overflow_is_reported uses them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_blob
from oracle import oracle as O
from rsr_mjx_amd import prng
from rsr_mjx_amd.mjcf import CompiledModel

m = CompiledModel.load(os.path.join(ROOT, "rsr_mjx_amd", "assets", "airbot_cube.npz"))
orc = O.Oracle(make_blob(m), "f64")
orc.set_ncon_cap(1000)
st = orc.new_state(1); orc.reset(st, prng.split(prng.PRNGKey(0), 1))
q0 = st["qpos"][0].astype(np.float64)
A = m.arrays
rng = np.random.default_rng(0)
found = []
for trial in range(20000):
    q = q0.copy()
    q[:6] = rng.uniform(A["jnt_range"][:6, 0], A["jnt_range"][:6, 1])
    q[8:11] = q0[8:11] + rng.uniform(-0.05, 0.05, 3) * [1, 1, 0]
    q[15:18] = q0[15:18] + rng.uniform(-0.15, 0.15, 3) * [1, 1, 0]
    orc.forward(q, np.zeros(20), q[[0, 1, 2, 4, 5]])
    ncon = int(orc.get("counts")[3])
    if 25 <= ncon <= 32:                      # a mild overflow: 1..8 contacts more than the capacity
        con = orc.get("contacts").reshape(-1, 10)
        if con[:, 0].min() > -0.03:
            found.append((ncon, q.astype(np.float32)))
    if len(found) >= 16:
        break
assert len(found) >= 8, len(found)
np.savez(os.path.join(ROOT, "tests", "golden", "cube_overcap_states.npz"), qpos=np.array([f[1] for f in found]), ncon=np.array([f[0] for f in found]))
print(len(found), [f[0] for f in found])

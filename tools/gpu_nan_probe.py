"""Do extreme / constant action streams ever produce non-finite state?  (robustness probe for the learner loop)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rsr_mjx_amd import prng
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, wrap, domain_randomize
n = 4096
env = wrap(AirbotPlayBase(), n, episode_length=1200, randomization_fn=lambda s: domain_randomize(s, prng.split(prng.PRNGKey(1), n)))
st = env.reset(prng.split(prng.PRNGKey(0), n))
g = torch.Generator(device="cuda"); g.manual_seed(0)
patterns = {"const +1": lambda t: torch.ones(n, 5, device="cuda"), "const -1": lambda t: -torch.ones(n, 5, device="cuda"),
            "sign flip": lambda t: torch.ones(n, 5, device="cuda") * (1 if (t // 7) % 2 else -1),
            "per-env const": None, "bernoulli +-1": lambda t: torch.sign(torch.randn(n, 5, device="cuda", generator=g))}
fixed = torch.sign(torch.randn(n, 5, device="cuda", generator=g))
patterns["per-env const"] = lambda t: fixed
for name, f in patterns.items():
    st = env.reset(prng.split(prng.PRNGKey(0), n))
    bad_obs = bad_rew = 0; dones = 0
    for t in range(1500):
        st = env.step(st, f(t))
        bad_obs += int((~torch.isfinite(st.obs)).any(dim=1).sum()); bad_rew += int((~torch.isfinite(st.reward)).sum()); dones += int(st.done.sum())
    q = env.view("qpos"); v = env.view("qvel")
    print(f"{name:14s} non-finite obs rows {bad_obs}, rewards {bad_rew}, dones {dones}, max|qvel| {float(v.abs().max()):.1f}, max|qpos| {float(q.abs().max()):.2f}, mean reward {float(st.reward.mean()):.3f}")

"""Where does the PPO run on the Airbot env first go non-finite?  (checks parameters, normaliser, gradients after every update)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize
from rsr_mjx_amd.learning import ppo_losses, ppo_train
orig_step = torch.optim.Adam.step
state = {"n": 0, "done": False}
def step(self, *a, **k):
    state["n"] += 1
    gmax = max(float(p.grad.abs().max()) for g in self.param_groups for p in g["params"] if p.grad is not None)
    r = orig_step(self, *a, **k)
    pmax = max(float(p.detach().abs().max()) for g in self.param_groups for p in g["params"])
    if not state["done"] and (not np.isfinite(pmax) or not np.isfinite(gmax) or state["n"] % 5000 == 0 or gmax > 1e3):
        print(f"adam step {state['n']}: grad max {gmax:.4g} param max {pmax:.4g}", flush=True)
        if not np.isfinite(pmax):
            state["done"] = True
            for g in self.param_groups:
                for i, p in enumerate(g["params"]):
                    st = self.state[p]
                    print("   param", i, tuple(p.shape), "finite", bool(torch.isfinite(p).all()), "exp_avg max", float(st["exp_avg"].abs().max()), "exp_avg_sq max", float(st["exp_avg_sq"].max()), "min", float(st["exp_avg_sq"].min()))
    return r
torch.optim.Adam.step = step
orig_upd = ppo_train.RunningStatistics.update
def upd(self, batch):
    orig_upd(self, batch)
    if not torch.isfinite(self.mean).all() or not torch.isfinite(self.std).all():
        print("normaliser non-finite: count", float(self.count), flush=True)
ppo_train.RunningStatistics.update = upd
ppo_train.train(AirbotPlayBase(), num_timesteps=15_000_000, num_evals=10, reward_scaling=0.1, episode_length=1200, normalize_observations=True,
                unroll_length=10, num_minibatches=32, num_updates_per_batch=8, discounting=0.96, learning_rate=1e-4, entropy_cost=2e-2, num_envs=1024,
                batch_size=256, rsr_loss_scale=0.0, randomization_fn=domain_randomize, seed=0,
                progress_fn=lambda s, m: print("eval", s, m["eval/episode_reward"], flush=True))

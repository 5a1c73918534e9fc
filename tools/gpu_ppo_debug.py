"""Where does the PPO run on the Airbot env first go non-finite?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize
from rsr_mjx_amd.learning import ppo_losses, ppo_train
orig = ppo_losses.compute_ppo_loss
state = {"n": 0, "reported": False}
def wrapped(policy, value, data, noise, **kw):
    total, m = orig(policy, value, data, noise, **kw)
    state["n"] += 1
    if not state["reported"] and (not torch.isfinite(total) or state["n"] % 4000 == 0):
        with torch.no_grad():
            obs = data.observation
            logits = policy(obs.transpose(0, 1))
            loc, scale = ppo_losses._split(logits)
            raw = data.extras["policy_extras"]["raw_action"]
            lp_b = data.extras["policy_extras"]["log_prob"]
            lp_t = ppo_losses.tanh_normal_log_prob(logits, raw.transpose(0, 1))
            print(f"minibatch {state['n']}: total {total.item():.4g} " + " ".join(f"{k}={v.item():.4g}" for k, v in m.items() if k != "total_loss"))
            print(f"   obs finite {bool(torch.isfinite(obs).all())} |obs|max {obs.abs().max().item():.3g}  loc |max| {loc.abs().max().item():.3g} scale [{scale.min().item():.3g}, {scale.max().item():.3g}] "
                  f"raw |max| {raw.abs().max().item():.3g}  lp_b [{lp_b.min().item():.3g}, {lp_b.max().item():.3g}]  lp_t-lp_b [{(lp_t - lp_b.transpose(0,1)).min().item():.3g}, {(lp_t - lp_b.transpose(0,1)).max().item():.3g}]", flush=True)
        if not torch.isfinite(total):
            state["reported"] = True
    return total, m
ppo_losses.compute_ppo_loss = wrapped
ppo_train.ppo_losses.compute_ppo_loss = wrapped
ppo_train.train(AirbotPlayBase(), num_timesteps=15_000_000, num_evals=10, reward_scaling=0.1, episode_length=1200, normalize_observations=True,
                unroll_length=10, num_minibatches=32, num_updates_per_batch=8, discounting=0.96, learning_rate=1e-4, entropy_cost=2e-2, num_envs=1024,
                batch_size=256, rsr_loss_scale=0.0, randomization_fn=domain_randomize, seed=0,
                progress_fn=lambda s, m: print("eval", s, m["eval/episode_reward"], flush=True))

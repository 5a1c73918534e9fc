"""PPO on Go2JoystickFlatTerrain / RoughTerrain / Go2Handstand / Go2Footstand with the hyper-parameters of reference
ppo_train/go2_training/mujoco_playground/config/locomotion_params.py:5-50 (8192 envs, unroll 20, batch 256 x 32 minibatches x 4
updates, lr 3e-4, entropy 1e-2, discount 0.97, max_grad_norm 1.0, policy / value MLPs (512, 256, 128), the critic on
`privileged_state`; episode length from the env's config: 1000 joystick, 500 handstand), with the domain randomisation of
go2/randomize.py, on the HIP stepper and the torch learner.

  python examples/train_go2_ppo.py --timesteps 200000000                       # the reference's budget (joystick)
  python examples/train_go2_ppo.py --env Go2Handstand --timesteps 100000000    # (handstand / footstand: :42-50)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rsr_mjx_amd.envs import go2  # noqa: E402
from rsr_mjx_amd.learning.ppo_train import train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="Go2JoystickFlatTerrain")
    ap.add_argument("--timesteps", type=int, default=30_000_000)
    ap.add_argument("--evals", type=int, default=6)
    ap.add_argument("--num-envs", type=int, default=8192)
    args = ap.parse_args()
    t0 = time.time()

    def progress(num_steps, m):
        print(f"[{time.time() - t0:7.1f} s] steps {num_steps:>11d}  eval/episode_reward {m['eval/episode_reward']:8.3f} +- {m['eval/episode_reward_std']:7.3f}"
              f"  len {m['eval/avg_episode_length']:6.1f}" + (f"  train sps {m['training/sps']:,.0f}" if "training/sps" in m else ""), flush=True)

    wrap = lambda env, n, ep, rf: go2.wrap_for_brax_training(env, n, episode_length=ep, randomization_fn=rf)
    env = go2.load(args.env)
    train(env, num_timesteps=args.timesteps, num_evals=args.evals, reward_scaling=1.0, episode_length=int(env._config["episode_length"]),
          normalize_observations=True, action_repeat=1, unroll_length=20, num_minibatches=32, num_updates_per_batch=4, discounting=0.97,
          learning_rate=3e-4, entropy_cost=1e-2, num_envs=args.num_envs, batch_size=256, max_grad_norm=1.0, rsr_loss_scale=0.0,
          policy_hidden_layer_sizes=(512, 256, 128), value_hidden_layer_sizes=(512, 256, 128), value_obs_key="privileged_state",
          randomization_fn=go2.domain_randomize, wrap_fn=wrap, seed=0, progress_fn=progress)
    print(f"total {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()

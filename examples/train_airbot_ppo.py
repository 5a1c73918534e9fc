"""PPO on the Airbot cube-push env with the hyper-parameters of reference ppo_train/airbot_training/train.py:45-56
(1024 envs, unroll 10, batch 256 x 32 minibatches x 8 updates, lr 1e-4, entropy 2e-2, discount 0.96, reward scaling 0.1,
observation normalisation, domain randomisation), on the HIP stepper and the torch learner.

  python examples/train_airbot_ppo.py --timesteps 15000000       # the reference's budget
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rsr_mjx_amd.envs.airbot import AirbotPlayBase, domain_randomize  # noqa: E402
from rsr_mjx_amd.learning.ppo_train import train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timesteps", type=int, default=3_000_000)
    ap.add_argument("--evals", type=int, default=6)
    ap.add_argument("--no-dr", action="store_true")
    ap.add_argument("--entropy-cost", type=float, default=2e-2)
    ap.add_argument("--learning-rate", type=float, default=1e-4)
    args = ap.parse_args()
    t0 = time.time()

    def progress(num_steps, metrics):
        print(f"[{time.time() - t0:7.1f} s] steps {num_steps:>10d}  eval/episode_reward {metrics['eval/episode_reward']:10.2f} "
              f"+- {metrics['eval/episode_reward_std']:8.2f}  len {metrics['eval/avg_episode_length']:7.1f}"
              + (f"  train sps {metrics['training/sps']:,.0f}" if "training/sps" in metrics else ""), flush=True)

    train(AirbotPlayBase(), num_timesteps=args.timesteps, num_evals=args.evals, reward_scaling=0.1, episode_length=1200,
          normalize_observations=True, action_repeat=1, unroll_length=10, num_minibatches=32, num_updates_per_batch=8,
          discounting=0.96, learning_rate=args.learning_rate, entropy_cost=args.entropy_cost, num_envs=1024, batch_size=256, rsr_loss_scale=0.0,
          randomization_fn=None if args.no_dr else domain_randomize, seed=0, progress_fn=progress)
    print(f"total {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()

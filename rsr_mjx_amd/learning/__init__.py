"""Learner-side pieces of RSR on torch-ROCm: the RSR distribution loss, GAE and the PPO / SAC losses
with the RSR term.  The env hot path does not depend on this package."""

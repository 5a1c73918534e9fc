"""Policy checkpoints of the torch learners: one .npz of named float arrays (no pickles).  Counterpart of the reference's
`model.save_params` / `model.load_params` calls (ppo_train/airbot_training/train.py:35-40, 98-99); the reference's Orbax /
pickle checkpoints are not read (they execute code on load)."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np


def _module_arrays(prefix: str, module) -> Dict[str, np.ndarray]:
    return {f"{prefix}/{k}": v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def save_params(path: str, params: Tuple[Any, ...]) -> None:
    """`params` as returned by ppo_train.train: (normalizer or None, PPONetworks), or by sac_train.train:
    (normalizer or None, policy_net, TwinQ)."""
    out: Dict[str, np.ndarray] = {}
    normalizer = params[0]
    if normalizer is not None:
        out.update({"normalizer/count": normalizer.count.cpu().numpy(), "normalizer/mean": normalizer.mean.cpu().numpy(),
                    "normalizer/summed_variance": normalizer.summed_variance.cpu().numpy(), "normalizer/std": normalizer.std.cpu().numpy()})
    if len(params) == 2:                                   # PPO
        out.update(_module_arrays("policy", params[1].policy)); out.update(_module_arrays("value", params[1].value))
    else:                                                  # SAC
        out.update(_module_arrays("policy", params[1])); out.update(_module_arrays("q1", params[2].q1)); out.update(_module_arrays("q2", params[2].q2))
    np.savez(path, **out)


def load_params(path: str, params: Tuple[Any, ...]) -> None:
    """Loads into already constructed modules of the same architecture (shapes are checked by load_state_dict)."""
    import torch
    z = np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False)
    def into(prefix, module):
        sd = {k[len(prefix) + 1:]: torch.as_tensor(z[k]) for k in z.files if k.startswith(prefix + "/")}
        module.load_state_dict(sd)
    normalizer = params[0]
    if normalizer is not None:
        for name in ("count", "mean", "summed_variance", "std"):
            getattr(normalizer, name).copy_(torch.as_tensor(z[f"normalizer/{name}"]))
    if len(params) == 2:
        into("policy", params[1].policy); into("value", params[1].value)
    else:
        into("policy", params[1]); into("q1", params[2].q1); into("q2", params[2].q2)

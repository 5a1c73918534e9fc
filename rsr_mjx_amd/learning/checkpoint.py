"""Policy checkpoints of the torch learners: one .npz of named float arrays (no pickles).  Counterpart of the reference's
`model.save_params` / `model.load_params` calls (ppo_train/airbot_training/train.py:35-40, 98-99); the reference's Orbax /
pickle checkpoints are not read (they execute code on load)."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np


def _module_arrays(prefix: str, module) -> Dict[str, np.ndarray]:
    return {f"{prefix}/{k}": v.detach().cpu().numpy() for k, v in module.state_dict().items()}


_STATS = ("count", "mean", "summed_variance", "std")


def _stats_arrays(prefix: str, rs) -> Dict[str, np.ndarray]:
    return {f"{prefix}/{name}": getattr(rs, name).cpu().numpy() for name in _STATS}


def save_params(path: str, params: Tuple[Any, ...]) -> None:
    """`params` as returned by ppo_train.train: (normalizer or None, PPONetworks), or by sac_train.train:
    (normalizer or None, policy_net, TwinQ).  The reference checkpoints the whole normalizer pytree, i.e. the statistics of
    every observation key: the critic's own normaliser (PPONetworks.value_normalizer, present when the critic reads another
    observation key such as Go2's privileged_state) is saved under value_normalizer/."""
    out: Dict[str, np.ndarray] = {}
    normalizer = params[0]
    if normalizer is not None:
        out.update(_stats_arrays("normalizer", normalizer))
    if len(params) == 2:                                   # PPO
        out.update(_module_arrays("policy", params[1].policy)); out.update(_module_arrays("value", params[1].value))
        if getattr(params[1], "value_normalizer", None) is not None:
            out.update(_stats_arrays("value_normalizer", params[1].value_normalizer))
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
    def stats(prefix, rs):
        for name in _STATS:
            getattr(rs, name).copy_(torch.as_tensor(z[f"{prefix}/{name}"]))
    normalizer = params[0]
    if normalizer is not None:
        stats("normalizer", normalizer)
    if len(params) == 2:
        into("policy", params[1].policy); into("value", params[1].value)
        vn = getattr(params[1], "value_normalizer", None)
        if vn is not None:
            if "value_normalizer/count" not in z.files:
                raise KeyError("checkpoint has no value_normalizer/ statistics but the critic normalises its own observation key")
            stats("value_normalizer", vn)
    else:
        into("policy", params[1]); into("q1", params[2].q1); into("q2", params[2].q2)

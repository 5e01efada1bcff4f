"""Generate tests/golden/reference_lr_vectors.json by RUNNING the reference's LR schedulers
(/root/reference/tzrec/optim/lr_scheduler.py:26-272) on a torch SGD optimizer with two parameter
groups (one with base rate 0: the freeze rule of BaseLR.get_lr).  Authoring container only:

    python tests/golden/make_reference_lr_vectors.py
"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_reference_module_vectors as mk  # noqa: E402

CASES = [
    ("ConstantLR", {}),
    ("ExponentialDecayLR", {"decay_size": 4, "decay_factor": 0.9}),
    ("ExponentialDecayLR", {"decay_size": 3, "decay_factor": 0.5, "staircase": False, "warmup_learning_rate": 0.001, "warmup_size": 5,
                            "min_learning_rate": 0.004}),
    ("ManualStepLR", {"schedule_sizes": [3, 7, 12], "learning_rates": [0.05, 0.02, 0.001]}),
    ("ManualStepLR", {"schedule_sizes": [4, 9], "learning_rates": [0.2, 0.01], "warmup": True}),
    ("CosineAnnealingLR", {"T_max": 10, "min_learning_rate": 0.001}),
    ("CosineAnnealingLR", {"T_max": 6, "warmup_learning_rate": 0.0, "warmup_size": 4}),
    ("CosineAnnealingWarmRestartsLR", {"T_0": 5}),
    ("CosineAnnealingWarmRestartsLR", {"T_0": 3, "T_mult": 2, "min_learning_rate": 0.002, "warmup_learning_rate": 0.01, "warmup_size": 2}),
]


def main():
    mk.install_reference_imports()
    L = importlib.import_module("tzrec.optim.lr_scheduler")
    out = []
    for name, kw in CASES:
        p0, p1 = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([{"params": [p0], "lr": 0.1}, {"params": [p1], "lr": 0.0}])
        sch = getattr(L, name)(opt, **kw)
        lrs = [[g["lr"] for g in opt.param_groups]]
        for _ in range(30):
            opt.step()
            sch.step()
            lrs.append([g["lr"] for g in opt.param_groups])
        out.append({"cls": name, "kwargs": kw, "by_epoch": bool(sch.by_epoch), "lrs": lrs})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_lr_vectors.json")
    json.dump({"generator": "tests/golden/make_reference_lr_vectors.py", "base_lrs": [0.1, 0.0], "cases": out}, open(path, "w"))
    print(f"wrote {path}: {len(out)} schedules x 31 steps")


if __name__ == "__main__":
    main()

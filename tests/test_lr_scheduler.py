"""Learning-rate schedules against OUTPUTS OF THE REFERENCE'S schedulers
(tests/golden/reference_lr_vectors.json, produced by running tzrec/optim/lr_scheduler.py:26-272;
generator: tests/golden/make_reference_lr_vectors.py), and their plumbing from a config block into
the fused sparse optimizer's device-side learning rate."""
import json
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torcheasyrec_amd import lr_scheduler as lrs  # noqa: E402
from torcheasyrec_amd.config import parse_text_proto  # noqa: E402

_G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_lr_vectors.json")))


class _Opt:
    def __init__(self, rates):
        self.param_groups = [{"lr": r} for r in rates]


@pytest.mark.parametrize("case", _G["cases"], ids=[f"{i}-{c['cls']}" for i, c in enumerate(_G["cases"])])
def test_schedule_matches_reference(case):
    opt = _Opt(_G["base_lrs"])
    sch = getattr(lrs, case["cls"])(opt, **case["kwargs"])
    assert sch.by_epoch == case["by_epoch"]
    got = [[g["lr"] for g in opt.param_groups]]
    for _ in range(len(case["lrs"]) - 1):
        sch.step()
        got.append([g["lr"] for g in opt.param_groups])
    assert got == case["lrs"]  # same double arithmetic: exact
    assert sch.get_last_lr() == case["lrs"][-1]


def test_state_dict_round_trip():
    opt = _Opt([0.1])
    sch = lrs.ExponentialDecayLR(opt, decay_size=2, decay_factor=0.5)
    for _ in range(5):
        sch.step()
    sd = sch.state_dict()
    opt2 = _Opt([0.1])
    sch2 = lrs.ExponentialDecayLR(opt2, decay_size=2, decay_factor=0.5)
    sch2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    sch.step(), sch2.step()
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]


def test_create_scheduler_from_config_block():
    blk = parse_text_proto("""
        adagrad_optimizer { lr: 0.1 }
        exponential_decay_learning_rate { decay_size: 4 decay_factor: 0.5 warmup_size: 2 warmup_learning_rate: 0.01 }
    """)
    opt = _Opt([0.1])
    sch = lrs.create_scheduler(opt, blk)
    assert isinstance(sch, lrs.ExponentialDecayLR) and not sch.by_epoch
    seq = [opt.param_groups[0]["lr"]]
    for _ in range(7):
        sch.step()
        seq.append(opt.param_groups[0]["lr"])
    w = float(torch.tensor(0.01, dtype=torch.float32))  # proto float
    assert seq[:3] == [w, (0.1 - w) * 0.5 + w, 0.1]
    assert seq[3:6] == [0.1, 0.1, 0.1] and seq[6] == 0.05
    assert isinstance(lrs.create_scheduler(_Opt([0.1]), parse_text_proto("adagrad_optimizer { lr: 0.1 } constant_learning_rate { }")), lrs.ConstantLR)
    assert isinstance(lrs.create_scheduler(_Opt([0.1]), None), lrs.ConstantLR)
    with pytest.raises(ValueError):
        lrs.create_scheduler(_Opt([0.1]), parse_text_proto("cosine_annealing_learning_rate { }"))


def test_schedule_drives_the_fused_sparse_update(dev):
    """A scheduled rate reaches the backward kernel through the optimizer's device scalar: SGD with
    ManualStepLR moves a row by exactly the scheduled rate each step."""
    from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    ebc = EmbeddingBagCollection([EmbeddingBagConfig("t", 8, 4, ["f"])], device=dev,
                                 optimizer=SparseOptimizerConfig(kind="sgd", lr=0.5))
    sch = lrs.ManualStepLR(ebc.fused_optimizer, schedule_sizes=[1, 2], learning_rates=[0.25, 0.125])
    kjt = KeyedJaggedTensor(["f"], torch.tensor([3], dtype=torch.int64, device=dev), torch.tensor([1], dtype=torch.int32, device=dev))
    w = ebc.table_weights()["t"]
    for want in (0.5, 0.5, 0.25, 0.125):  # steps 0, 1 (bisect_left: boundary still the old rate), 2, 3
        assert ebc.fused_optimizer.param_groups[0]["lr"] == want
        before = w.detach()[3].clone()
        out = ebc(kjt).values()
        out.sum().backward()
        torch.testing.assert_close(w.detach()[3], before - want, rtol=0, atol=1e-6)
        sch.step()

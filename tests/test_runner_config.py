"""Boundary shown, not claimed: the reference-schema YAML tree composes into the task constructors'
arguments without hydra, the stand-in Trainer deals batches / predict views to ranks and gathers them (2-rank gloo)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
CONFIG = ROOT / "contouring-uncertainty_amd" / "config"
for p in (str(ROOT), str(ROOT / "contouring-uncertainty_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

SIX = ["task.model.kernels=[[3,3],[3,3],[3,3],[3,3],[3,3],[3,3]]",
       "task.model.strides=[[1,1],[2,2],[2,2],[2,2],[2,2],[2,2]]"]


@pytest.mark.parametrize("task,target,name", [
    ("dsnt-al", "contour_uncertainty.task.regression.dsnt.dsnt_al.DSNTAleatoric", "dsnt-al-unet2-False"),
    ("dsnt-skew", "contour_uncertainty.task.regression.dsnt.dsnt_skew.DSNTSkew", "dsnt-skew-all-unet2-False"),
    ("dsnt-al2", "contour_uncertainty.task.regression.dsnt.dsnt_al.DSNTAleatoric", "dsnt-al2-unet2-False"),
])
def test_reference_schema_configs_compose_and_instantiate(task, target, name):
    """config/task/{dsnt-al,dsnt-skew,dsnt-al2}.yaml + task_default.yaml + default.yaml + model/unet2.yaml + optim/adam.yaml
    (the reference's files and keys) -> the composed `task` node -> the drop-in class, exactly as
    vital/vital/runner.py:110-112 instantiates it."""
    from contour_uncertainty._compat import instantiate
    from contour_uncertainty._config import compose
    cfg = compose(CONFIG, "default", [f"task={task}", "data=synthetic", "data.size=64", "data.batch_size=2"] + SIX)
    assert cfg.task._target_ == target and cfg.task.task_name == name
    assert cfg.choices == {"task": task, "data": "synthetic", "task/model": "unet2", "task/optim": "adam", "trainer": "default"}
    # every key of the reference's composed task node is there (task_default.yaml:5-19, default.yaml:6-19, dsnt-*.yaml)
    for key in ("model", "optim", "covar", "mse_weight", "log_penalty_weight", "psm_path", "seq_psm_path", "sequence_sampler",
                "t_a", "t_e", "train_ensemble", "log_figures", "task_name", "load_name", "name", "model_path",
                "best_model_save_path", "enable_model_summary", "train_log_kwargs", "val_log_kwargs"):
        assert key in cfg.task, key
    assert cfg.task.optim == {"_target_": "torch.optim.Adam", "lr": 0.001, "weight_decay": 0.001}
    assert cfg.task.t_a == 25 and cfg.task.t_e == 1 and cfg.task.covar is True
    assert cfg.name == f"synthetic-lv_{name}_10" and cfg.best_model_save_path.endswith(f"/10/{cfg.name}.ckpt")
    if task == "dsnt-skew":
        assert cfg.task.psm_path == "synthetic_psm_11_no_std.npy" and cfg.task.skew_indices is None
    dm = instantiate(cfg.data, _recursive_=False)
    model = instantiate(cfg.task, choices=cfg.choices, data_params=dm.data_params, _recursive_=False)
    assert type(model).__name__ == target.rsplit(".", 1)[1]
    assert model.hparams.task_name == name and model.hparams.choices["task"] == task
    assert len(model.model.downsamples) == 4 and model.hparams.data_params.out_shape == (21, 2)
    from cu_hip.optim import FusedAdam
    assert isinstance(model.configure_optimizers()["optimizer"], FusedAdam)


def test_config_overrides_and_resolvers():
    from contour_uncertainty._config import compose
    cfg = compose(CONFIG, "default", ["task=dsnt-al", "data=synthetic", "task.sequence_sampler=True", "task.model.drop_block=True",
                                      "seed=3", "trainer.devices=8", "data.labels=[bg,lv,myo]"])
    assert cfg.task.task_name == "dsnt-alsequence-unet2-True"            # the reference's `if` resolver (runner.py:25-27)
    assert cfg.id == "synthetic-lv-myo_dsnt-alsequence-unet2-True"       # ... and `labels` (runner.py:17-20)
    assert cfg.trainer.devices == 8 and cfg.trainer.fast_dev_run == 10 and cfg.seed == 3
    assert cfg.model_path.endswith("/3")
    with pytest.raises(ValueError):
        compose(CONFIG, "default", ["data=synthetic"])                   # task: ??? must be chosen
    full = compose(CONFIG, "default", ["task=dsnt-skew", "data=synthetic"])
    assert len(full.task.model.kernels) == 8 and full.task.model.patch_size == [256, 256]


# ---------------------------------------------------------------------------------------- Trainer: sharding + gather
class _StubResult:
    def __init__(self, view, draw):
        self.id, self.draw, self.view_index = f"view{view}", draw, None


class _StubTask(torch.nn.Module):
    """predict_step = one draw from torch's generator per view: what the PSM samplers consume"""
    trainer = None

    def on_predict_start(self):
        pass

    def predict_step(self, batch, idx):
        return _StubResult(idx, torch.randn(3).numpy() + float(batch["img"].sum()))


class _StubDM:
    def setup(self, stage):
        pass

    def predict_dataloader(self):
        return [{"img": torch.full((2, 1, 4, 4), float(i)), "id": f"v{i}"} for i in range(7)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _predict_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CONTOUR_DIST_BACKEND="gloo")
    from contour_uncertainty._compat import Trainer
    tr = Trainer(device="cpu")
    res = tr.predict(_StubTask(), datamodule=_StubDM(), gather=True, seed=5)
    ret[rank] = [(r.view_index, r.id, r.draw.tolist()) for r in res]
    dist.destroy_process_group()


def test_predict_views_are_sharded_over_ranks_and_gathered_in_order():
    """BASELINE config c5 / SURVEY 8e: views (ED/ES pairs) are dealt to ranks, every rank ends with ALL results in view
    order, and a view's draws do not depend on the sharding (per-view seeding)."""
    from contour_uncertainty._compat import Trainer
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    single = Trainer(device="cpu").predict(_StubTask(), datamodule=_StubDM(), gather=True, seed=5)
    assert [r.view_index for r in single] == list(range(7))
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_predict_worker, args=(2, port, ret), nprocs=2, join=True)
    for rank in (0, 1):
        got = ret[rank]
        assert [g[0] for g in got] == list(range(7)) and [g[1] for g in got] == [f"view{i}" for i in range(7)]
        for g, s in zip(got, single):
            assert np.allclose(g[2], s.draw)


def test_batches_are_dealt_round_robin():
    from contour_uncertainty._compat import Trainer
    tr = Trainer(device="cpu")
    tr.world, tr.rank = 3, 1
    assert list(tr._mine(range(10))) == [(0, 1), (1, 4), (2, 7)]          # the ragged tail (9) is dropped on every rank
    tr.world, tr.rank = 1, 0
    assert list(tr._mine("abc")) == [(0, "a"), (1, "b"), (2, "c")]

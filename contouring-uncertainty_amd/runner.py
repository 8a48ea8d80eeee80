#!/usr/bin/env python
"""Stand-in for the reference's ``runner.py`` + ``vital/vital/runner.py:94-145`` where hydra / pytorch_lightning are not
installed: composes the SAME YAML tree (``config/``), instantiates the datamodule and the task by their ``_target_`` with
the reference's arguments (``choices``, ``data_params``), fits, saves / loads checkpoints, predicts.

    python runner.py task=dsnt-skew data=synthetic data.size=64 trainer.fast_dev_run=2
    python -m torch.distributed.run --nproc-per-node 8 runner.py task=dsnt-al2 data=synthetic trainer.devices=8

With hydra + Lightning installed, the reference's own runner drives the drop-in classes of this package unchanged."""
from __future__ import annotations

import os
import random
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
if str(PKG) not in sys.path:
    sys.path.insert(0, str(PKG))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from contour_uncertainty._compat import Trainer, instantiate  # noqa: E402
from contour_uncertainty._config import compose  # noqa: E402


def seed_everything(seed: int) -> int:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


def run_system(cfg):
    """vital/vital/runner.py:62-145 without loggers / result processors; returns (model, trainer, predictions)."""
    seed_everything(int(cfg.seed))
    datamodule = instantiate(cfg.data, _recursive_=False)
    model = instantiate(cfg.task, choices=cfg.choices, data_params=datamodule.data_params, _recursive_=False)
    if cfg.get("ckpt"):
        model = type(model).load_from_checkpoint(cfg.ckpt, data_params=datamodule.data_params, strict=cfg.strict)
    elif cfg.get("weights"):
        state = torch.load(str(cfg.weights), map_location="cpu", weights_only=False)["state_dict"]
        model.load_state_dict(state, strict=cfg.strict)
    tcfg = {k: v for k, v in cfg.trainer.items() if k not in ("_target_",)}
    trainer = Trainer(**tcfg)
    trainer._model = model
    predictions = None
    if cfg.train:
        trainer.fit(model, datamodule=datamodule)
        if not cfg.trainer.get("fast_dev_run", False) and cfg.get("best_model_save_path"):
            Path(cfg.best_model_save_path).parent.mkdir(parents=True, exist_ok=True)
            trainer.save_checkpoint(cfg.best_model_save_path)
    if cfg.predict:
        predictions = trainer.predict(model, datamodule=datamodule, gather=True, seed=int(cfg.seed))
    return model, trainer, predictions


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = compose(Path(os.environ.get("CONTOUR_CONFIG_DIR", PKG / "config")), "default", argv)
    return run_system(cfg)


if __name__ == "__main__":
    main()

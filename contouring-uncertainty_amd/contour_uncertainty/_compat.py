"""Boundary shims: use the reference's own framework layers when they are installed, minimal stand-ins otherwise.

The reference's L3-L5 (pytorch_lightning, hydra, the ``vital`` submodule) are NOT re-implemented.
When they are importable the task classes inherit from the real ``vital.tasks.generic.SharedStepsTask`` and are
instantiated by the real ``hydra.utils.instantiate`` -- the drop-in case.  In a bare container (this image has none of
them) the stand-ins below honour the same surface: ``save_hyperparameters`` -> ``self.hparams``, ``log`` / ``log_dict``,
``configure_optimizers() -> {"optimizer": ...}``, ``training_step`` / ``validation_step`` key prefixes
(reference vital/vital/system.py:17-115, vital/vital/tasks/generic.py:10-41, vital/vital/utils/format/native.py:5-21).
"""
from __future__ import annotations

import importlib
import inspect
import os
from abc import ABC
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn


# ------------------------------------------------------------------------------------------------ config helpers
class AttrDict(dict):
    """dict with attribute access (stands in for OmegaConf DictConfig / Lightning AttributeDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(obj):
    if isinstance(obj, Mapping) and not isinstance(obj, AttrDict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)) and not isinstance(obj, str):
        return type(obj)(to_attr(v) for v in obj)
    return obj


try:  # pragma: no cover - exercised only where hydra is installed
    from hydra.utils import instantiate, to_absolute_path  # type: ignore
    HAVE_HYDRA = True
except ImportError:
    HAVE_HYDRA = False

    def _locate(path: str):
        mod, _, attr = path.rpartition(".")
        return getattr(importlib.import_module(mod), attr)

    def instantiate(cfg, *args, **kwargs):
        """Minimal ``hydra.utils.instantiate``: ``_target_`` + keyword merge, non-recursive."""
        cfg = dict(cfg)
        target = cfg.pop("_target_")
        cfg.pop("_recursive_", None)
        cfg.pop("_partial_", None)
        cfg.update(kwargs)
        return _locate(target)(*args, **cfg)

    def to_absolute_path(path: str) -> str:
        p = Path(path)
        return str(p if p.is_absolute() else Path(os.getcwd()) / p)


# ------------------------------------------------------------------------------------------------ data tags / params
class Tags:
    """vital.data.config.Tags (only the members the dsnt path reads)."""
    id = "id"
    group = "group"
    img = "img"
    gt = "gt"
    pred = "pred"


class ContourTags(Tags):
    """contour_uncertainty.data.config.ContourTags"""
    contour = "contour"


@dataclass
class DataParameters:
    """vital.data.config.DataParameters (reference vital/vital/data/config.py:96-109)."""
    in_shape: Tuple[int, ...]
    out_shape: Tuple[int, ...]
    labels: Optional[Sequence[Any]] = None


def prefix(d: Dict[str, Any], pre: str) -> Dict[str, Any]:
    """vital.utils.format.native.prefix"""
    return {f"{pre}{k}": v for k, v in d.items()}


# ------------------------------------------------------------------------------------------------ LightningModule
try:  # pragma: no cover
    import pytorch_lightning as pl  # type: ignore
    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except ImportError:
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        """The slice of ``pl.LightningModule`` the tasks touch."""

        def __init__(self):
            super().__init__()
            self._hparams = AttrDict()
            self.trainer = None
            self.current_epoch = 0
            self.logged: Dict[str, Any] = {}

        @property
        def hparams(self):
            return self._hparams

        def save_hyperparameters(self, *args, **_kw):
            """No args: capture the caller's ``__init__`` arguments (incl. **kwargs); one dict arg: merge it."""
            if args and isinstance(args[0], Mapping):
                self._hparams.update(to_attr(args[0]))
                return
            frame = inspect.currentframe().f_back
            info = inspect.getargvalues(frame)
            for name in info.args:
                if name != "self":
                    self._hparams[name] = to_attr(info.locals[name])
            if info.keywords and info.keywords in info.locals:
                for k, v in info.locals[info.keywords].items():
                    self._hparams[k] = to_attr(v)

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, name, value, **_kw):
            self.logged[name] = value

        # checkpoints in Lightning's layout: {"state_dict": ..., "hyper_parameters": ...}
        def save_checkpoint(self, path):
            hp = {k: v for k, v in self._hparams.items() if k != "task"}
            torch.save({"state_dict": self.state_dict(), "hyper_parameters": hp}, str(path))

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict: bool = True, **overrides):
            ckpt = torch.load(str(checkpoint_path), map_location=map_location or "cpu", weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.pop("task", None)
            hp.update(overrides)
            obj = cls(**hp)
            obj.load_state_dict(ckpt["state_dict"], strict=strict)
            return obj

        def log_dict(self, d, **_kw):
            self.logged.update(d)


try:  # pragma: no cover
    from vital.system import VitalSystem  # type: ignore
    from vital.tasks.generic import SharedStepsTask  # type: ignore
    HAVE_VITAL = True
except ImportError:
    HAVE_VITAL = False

    class VitalSystem(LightningModule, ABC):
        """reference vital/vital/system.py:17-115 (hparams capture + optimizer from config)."""

        def __init__(self, model=None, optim=None, choices=None, data_params: DataParameters = None, **kwargs):
            super().__init__()
            self.save_hyperparameters()
            self.save_hyperparameters({"task": {"_target_": f"{self.__class__.__module__}.{self.__class__.__name__}"}})

        def configure_model(self) -> nn.Module:
            return instantiate(self.hparams.model, input_shape=self.hparams.data_params.in_shape,
                               output_shape=self.hparams.data_params.out_shape)

        def configure_optimizers(self, params=None):
            if params is None:
                params = self.parameters()
            scheduler_cfg = None
            optim_cfg = self.hparams.optim
            if optim_cfg.get("optimizer"):
                scheduler_cfg = optim_cfg.get("lr_scheduler")
                optim_cfg = optim_cfg["optimizer"]
            out = {"optimizer": instantiate(optim_cfg, params=params)}
            if scheduler_cfg:
                out["lr_scheduler"] = instantiate(scheduler_cfg, optimizer=out["optimizer"])
            return out

    class SharedStepsTask(VitalSystem, ABC):
        """reference vital/vital/tasks/generic.py:10-41"""

        def _shared_step(self, *args, **kwargs) -> Dict[str, Tensor]:
            raise NotImplementedError

        def training_step(self, *args, **kwargs) -> Dict[str, Tensor]:
            result = prefix(self._shared_step(*args, **kwargs), "train/")
            self.log_dict(result, **(self.hparams.get("train_log_kwargs") or {}))
            result["loss"] = result["train/loss"]
            return result

        def validation_step(self, *args, **kwargs) -> Dict[str, Tensor]:
            result = prefix(self._shared_step(*args, **kwargs), "val/")
            self.log_dict(result, **(self.hparams.get("val_log_kwargs") or {}))
            return result

        def test_step(self, *args, **kwargs) -> Dict[str, Tensor]:
            result = prefix(self._shared_step(*args, **kwargs), "test/")
            self.log_dict(result, **(self.hparams.get("val_log_kwargs") or {}))
            return result


def resolve_model_checkpoint_path(checkpoint):
    """vital.utils.saving.resolve_model_checkpoint_path for local files (Comet model-registry lookups are host glue)."""
    from pathlib import Path
    path = Path(str(checkpoint))
    if not path.exists():
        raise FileNotFoundError(f"checkpoint {checkpoint} not found (only local checkpoint files are supported)")
    return path


def fused_optimizer_cfg(optim_cfg):
    """Map the reference's ``_target_: torch.optim.Adam`` (vital/vital/config/task/optim/adam.yaml) onto the fused HIP
    Adam with identical semantics; any other optimizer is left untouched."""
    cfg = dict(optim_cfg)
    if cfg.get("_target_") == "torch.optim.Adam" and not cfg.get("amsgrad", False):
        cfg["_target_"] = "cu_hip.optim.FusedAdam"
    return to_attr(cfg)

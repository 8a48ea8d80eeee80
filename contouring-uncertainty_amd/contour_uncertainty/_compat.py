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
        cfg.update(kwargs)
        target = cfg.pop("_target_")
        cfg.pop("_recursive_", None)
        cfg.pop("_partial_", None)
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

        # checkpoints in Lightning 1.8's layout (contour_uncertainty/utils/checkpoint.py): state_dict, hyper_parameters,
        # optimizer_states, epoch, global_step, ...
        def save_checkpoint(self, path, optimizer=None, epoch: int = 0, global_step: int = 0):
            from contour_uncertainty.utils.checkpoint import lightning_checkpoint
            ckpt = lightning_checkpoint(self, optimizer, epoch, global_step)
            ckpt["hyper_parameters"] = {k: v for k, v in self._hparams.items() if k != "task"}   # live objects: same-process reload
            torch.save(ckpt, str(path))

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict: bool = True, **overrides):
            from contour_uncertainty.utils.checkpoint import load_lightning_checkpoint
            ckpt = load_lightning_checkpoint(checkpoint_path, map_location or "cpu")
            hp = dict(ckpt.get("hyper_parameters") or {})
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


# ------------------------------------------------------------------------------------------------ Trainer stand-in
class GradSyncCallback:
    """Hooks the bucketed gradient exchange of ``cu_hip.ddp.GradSync`` into a training loop: a Lightning ``Callback`` by
    duck typing (``on_fit_start`` / ``on_before_optimizer_step`` are the hook names of pytorch_lightning >= 1.5) and the
    multi-process path of the stand-in ``Trainer`` below.  Use it with a strategy that does NOT wrap the module in
    ``torch.nn.parallel.DistributedDataParallel`` (under Lightning's own ``strategy="ddp"`` torch DDP already exchanges
    the gradients -- the HIP path works there unchanged -- and this callback stays passive)."""

    def __init__(self, bucket_elems: int = 8 * 1024 * 1024):
        self.bucket_elems = bucket_elems
        self.sync = None

    def on_fit_start(self, trainer, pl_module):
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel
        from cu_hip.ddp import GradSync
        wrapped = isinstance(getattr(getattr(trainer, "strategy", None), "model", None), DistributedDataParallel)
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1 or wrapped:
            return
        self.sync = GradSync(pl_module, self.bucket_elems)
        self.sync.overlap = int(getattr(trainer, "accumulate_grad_batches", 1) or 1) == 1
        self.sync.broadcast_parameters()

    def on_before_optimizer_step(self, trainer, pl_module, optimizer, *args):
        if self.sync is None:
            return
        self.sync.finish()
        if hasattr(optimizer, "grad_scale"):
            optimizer.grad_scale = self.sync.grad_scale
        else:                                   # any other optimizer: scale the summed gradients in place
            for group in optimizer.param_groups:
                for p in group["params"]:
                    if p.grad is not None:
                        p.grad.mul_(self.sync.grad_scale)


def _to_device(x, device):
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    if isinstance(x, Mapping):
        return type(x)((k, _to_device(v, device)) for k, v in x.items()) if not isinstance(x, dict) else \
            {k: _to_device(v, device) for k, v in x.items()}
    return x


class Trainer:
    """The slice of ``pytorch_lightning.Trainer`` the reference's runner touches (vital/vital/runner.py:94-145): ``fit`` and
    ``predict`` on ONE device per process.  Launched by ``torch.distributed.run`` (``trainer.devices`` ranks) the
    minibatches / the predict views are dealt round-robin to the ranks, gradients go through ``GradSyncCallback``
    (RCCL over xGMI with backend "nccl"), and ``predict`` can gather every rank's results."""

    def __init__(self, max_epochs: int = 1000, max_steps: int = -1, devices=1, fast_dev_run=False, accelerator="auto",
                 default_root_dir=None, logger=None, callbacks=None, accumulate_grad_batches: int = 1,
                 limit_val_batches=None, device=None, **_unused):
        self.max_epochs, self.max_steps = max_epochs, max_steps
        self.devices = devices
        self.fast_dev_run = int(fast_dev_run) if fast_dev_run else 0
        self.accumulate_grad_batches = accumulate_grad_batches
        self.limit_val_batches = limit_val_batches
        self.logger, self.callbacks = logger, list(callbacks or [])
        self.default_root_dir = default_root_dir
        self.datamodule = None
        self.callback_metrics: Dict[str, Any] = {}
        self.global_step = 0
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._forced_device = device     # tests of the loop logic with stub modules; the DSNT tasks need a GPU regardless

    # -- plumbing
    def _device(self):
        if self._forced_device is not None:
            return torch.device(self._forced_device)
        if not torch.cuda.is_available():
            from cu_hip.lib import ContourHipError
            raise ContourHipError("Trainer: no MI355X visible (the HIP path has no CPU fallback)")
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(self.local_rank % ndev)
        return torch.device("cuda", self.local_rank % ndev)

    def _init_distributed(self):
        import torch.distributed as dist
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("CONTOUR_DIST_BACKEND", "nccl"))

    def _call(self, hook, *args):
        for cb in self.callbacks:
            fn = getattr(cb, hook, None)
            if fn is not None:
                fn(self, *args)

    def _mine(self, iterable):
        """this rank's share of an iterable, dealt round-robin; every rank gets the same count (the tail is dropped)"""
        items = []
        for i, item in enumerate(iterable):
            items.append(item)
            if len(items) == self.world:
                yield i // self.world, items[self.rank]
                items = []

    # -- fit
    def fit(self, model, datamodule=None):
        device = self._device()
        self._init_distributed()
        self.datamodule = datamodule
        model.trainer = self
        try:
            datamodule.trainer = self       # Lightning attaches itself to the data module too (``trainer.training`` gates its hooks)
        except AttributeError:
            pass
        model.to(device)
        datamodule.setup("fit")
        if not any(isinstance(c, GradSyncCallback) for c in self.callbacks) and self.world > 1:
            self.callbacks.append(GradSyncCallback())
        model.on_fit_start()
        self._call("on_fit_start", model)
        optimizer = model.configure_optimizers()["optimizer"]
        self._model, self._optimizer = model, optimizer
        epochs = 1 if self.fast_dev_run else self.max_epochs
        done = False
        for epoch in range(epochs):
            model.current_epoch = self.current_epoch = epoch
            model.train()
            self.training = True
            for idx, batch in self._mine(datamodule.train_dataloader()):
                batch = _to_device(batch, device)
                hook = getattr(datamodule, "on_after_batch_transfer", None)      # Lightning's hook: on-device augmentation
                if hook is not None:
                    batch = hook(batch, idx)
                out = model.training_step(batch, idx)
                (out["loss"] / self.accumulate_grad_batches).backward()
                if (idx + 1) % self.accumulate_grad_batches == 0:
                    self._call("on_before_optimizer_step", model, optimizer)
                    optimizer.step()
                    optimizer.zero_grad(set_to_none=True)
                    self.global_step += 1
                self.callback_metrics.update({k: v.detach() if torch.is_tensor(v) else v for k, v in out.items()})
                if (self.fast_dev_run and idx + 1 >= self.fast_dev_run) or 0 < self.max_steps <= self.global_step:
                    done = True
                    break
            model.eval()
            self.training = False
            with torch.no_grad():
                for idx, batch in self._mine(datamodule.val_dataloader()):
                    batch = _to_device(batch, device)
                    hook = getattr(datamodule, "on_after_batch_transfer", None)      # called for every stage, as Lightning does
                    if hook is not None:
                        batch = hook(batch, idx)
                    out = model.validation_step(batch, idx)
                    self.callback_metrics.update({k: v.detach() if torch.is_tensor(v) else v for k, v in out.items()})
                    if self.fast_dev_run and idx + 1 >= self.fast_dev_run:
                        break
            if done:
                break
        if hasattr(model, "on_fit_end"):
            model.on_fit_end()
        return self.callback_metrics

    # -- predict: views are independent (an ED / ES pair stays on one rank); deterministic per view whatever the sharding
    def predict(self, model, datamodule=None, gather: bool = False, seed: int = 0):
        device = self._device()
        self._init_distributed()
        self.datamodule = datamodule
        model.trainer = self
        model.to(device).eval()
        self.training = False
        datamodule.setup("predict")
        model.on_predict_start()
        results = []
        loader = datamodule.predict_dataloader()
        with torch.no_grad():
            for view, batch in enumerate(loader):
                if view % self.world != self.rank:
                    continue
                if self.fast_dev_run and len(results) >= self.fast_dev_run:
                    break
                torch.manual_seed(seed + view)              # the samplers draw from torch's generator: same draws per view
                res = model.predict_step(_to_device(batch, device), view)
                res.view_index = view
                results.append(res)
        if gather and self.world > 1:
            import torch.distributed as dist
            slim = [_slim(r) for r in results]
            every = [None] * self.world
            dist.all_gather_object(every, slim)
            results = sorted((r for part in every for r in part), key=lambda r: r.view_index)
        return results

    def save_checkpoint(self, path):
        """Lightning-layout checkpoint (weights, hyper-parameters, optimizer state, epoch / global_step) from rank 0"""
        if self.rank == 0 and self._model is not None:
            self._model.save_checkpoint(path, optimizer=self._optimizer, epoch=getattr(self, "current_epoch", 0),
                                        global_step=self.global_step)

    _model = None
    _optimizer = None


def _slim(res):
    """a BatchResult without its device tensors (what travels between ranks)"""
    import copy
    out = copy.copy(res)
    for k, v in list(vars(out).items()):
        if torch.is_tensor(v):
            setattr(out, k, v.detach().cpu().numpy())
    return out

"""Lightning-format ``.ckpt`` interchange without Lightning (checkpoint half).

The reference saves and restores its tasks through ``pytorch_lightning`` (``ModelCheckpoint``; ``Trainer.save_checkpoint``;
``VitalSystem.load_from_checkpoint`` and the ``weights=`` branch of reference vital/vital/runner.py:113-120, which reads
``torch.load(path)["state_dict"]``).  A Lightning 1.8 checkpoint is one ``torch.save``-d dict::

    {"epoch", "global_step", "pytorch-lightning_version", "state_dict", "loops", "callbacks",
     "optimizer_states": [optimizer.state_dict(), ...], "lr_schedulers": [...],
     "hparams_name": "kwargs", "hyper_parameters": {...}}

with ``state_dict`` keys ``model.*`` (+ ``skew_block.model.*`` for dsnt-skew) -- the names this package keeps (strict load is
tested) -- and ``optimizer_states[0]`` in ``torch.optim.Adam``'s layout, which ``cu_hip.optim.FusedAdam`` keeps as well.
This module writes and reads that layout so that weights trained with the reference can be benchmarked / fine-tuned here
and the other way round.

Reading needs care: a checkpoint written by the reference pickles ``omegaconf.DictConfig`` / ``AttributeDict`` objects inside
``hyper_parameters`` (and callback state), whose classes may not be importable where the file is read (they are not in this
image).  :func:`load_lightning_checkpoint` therefore unpickles with a tolerant class resolver: classes of missing modules
become inert stand-ins, the tensors (``state_dict``, optimizer moments) load untouched, and the stand-ins are reduced to plain
containers where their pickled state allows it.
"""
from __future__ import annotations

import pickle
from collections.abc import Mapping
from pathlib import Path
from typing import Any, Dict, Optional

import torch

LIGHTNING_VERSION = "1.8.0"          # reference vital/pyproject.toml: pytorch-lightning ~1.8.0
_KEYS = ("epoch", "global_step", "pytorch-lightning_version", "state_dict", "loops", "callbacks", "optimizer_states",
         "lr_schedulers")


def _plain(obj):
    """hyper-parameters as plain picklable containers (AttrDict / DataParameters / tuples ...)"""
    if isinstance(obj, Mapping):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if hasattr(obj, "__dataclass_fields__"):
        return {k: _plain(getattr(obj, k)) for k in obj.__dataclass_fields__}
    if isinstance(obj, (str, int, float, bool)) or obj is None:
        return obj
    if isinstance(obj, Path):
        return str(obj)
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    return repr(obj)


def lightning_checkpoint(module, optimizer=None, epoch: int = 0, global_step: int = 0, callbacks: Optional[dict] = None,
                         lr_schedulers: Optional[list] = None) -> Dict[str, Any]:
    """The checkpoint dict ``Trainer.save_checkpoint`` of Lightning 1.8 would write for ``module`` (a dsnt task)."""
    sd = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    hp = {k: _plain(v) for k, v in dict(getattr(module, "hparams", {})).items()}
    ckpt = {
        "epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": LIGHTNING_VERSION,
        "state_dict": sd,
        "loops": {}, "callbacks": dict(callbacks or {}),
        "optimizer_states": [], "lr_schedulers": list(lr_schedulers or []),
        "hparams_name": "kwargs", "hyper_parameters": hp,
    }
    if optimizer is not None:
        osd = optimizer.state_dict()
        osd = {"state": {k: {n: (t.detach().cpu().clone() if torch.is_tensor(t) else t) for n, t in st.items()}
                         for k, st in osd["state"].items()},
               "param_groups": [dict(g) for g in osd["param_groups"]]}
        ckpt["optimizer_states"] = [osd]
    return ckpt


def save_lightning_checkpoint(module, path, optimizer=None, epoch: int = 0, global_step: int = 0, **kw) -> Path:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(lightning_checkpoint(module, optimizer, epoch, global_step, **kw), str(path))
    return path


class _Missing:
    """stand-in for an object whose class cannot be imported here (omegaconf.DictConfig, a Lightning callback state ...)"""

    def __init__(self, *args, **kwargs):
        self._args, self._state = args, None

    def __setstate__(self, state):
        self._state = state

    def __reduce_ex__(self, protocol):          # re-saving a loaded checkpoint keeps the inert object
        return (_Missing, ())


def _tolerant_pickle():
    import types
    mod = types.ModuleType("_contour_tolerant_pickle")

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return type(name, (_Missing,), {"__module__": module})

    mod.Unpickler = Unpickler
    mod.load = lambda f, **kw: Unpickler(f, **kw).load()
    mod.__name__ = "pickle"
    for k in ("dumps", "dump", "loads", "Pickler", "PickleError", "UnpicklingError", "HIGHEST_PROTOCOL", "DEFAULT_PROTOCOL"):
        setattr(mod, k, getattr(pickle, k))
    return mod


def _reduce_missing(obj):
    """inert stand-ins -> plain containers where their pickled state is one (DictConfig keeps its content under '_content')"""
    if isinstance(obj, _Missing):
        st = obj._state
        if isinstance(st, Mapping) and "_content" in st:
            return _reduce_missing(st["_content"])
        if isinstance(st, Mapping) and "_val" in st:          # omegaconf value nodes
            return _reduce_missing(st["_val"])
        return _reduce_missing(st) if st is not None else None
    if isinstance(obj, Mapping):
        return {k: _reduce_missing(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_reduce_missing(v) for v in obj)
    return obj


def load_lightning_checkpoint(path, map_location="cpu") -> Dict[str, Any]:
    """``torch.load`` of a Lightning checkpoint that does not need Lightning, omegaconf or the writer's classes."""
    ckpt = torch.load(str(path), map_location=map_location, weights_only=False, pickle_module=_tolerant_pickle())
    if not isinstance(ckpt, Mapping) or "state_dict" not in ckpt:
        raise ValueError(f"{path}: not a Lightning checkpoint (no 'state_dict')")
    ckpt = dict(ckpt)
    for k in ("hyper_parameters", "callbacks", "loops"):
        if k in ckpt:
            ckpt[k] = _reduce_missing(ckpt[k])
    return ckpt


def load_weights(module, path, strict: bool = True, map_location=None):
    """reference vital/vital/runner.py:117-120: ``model.load_state_dict(torch.load(weights)["state_dict"], strict=cfg.strict)``"""
    ckpt = load_lightning_checkpoint(path, map_location or "cpu")
    return module.load_state_dict(ckpt["state_dict"], strict=strict)


def restore(module, path, optimizer=None, strict: bool = True) -> Dict[str, Any]:
    """weights + (optionally) optimizer state of a checkpoint -> ``module`` / ``optimizer``; returns the checkpoint dict
    (``epoch``, ``global_step`` for the caller's loop).  The optimizer state is ``torch.optim.Adam``'s layout; ``FusedAdam``
    re-packs it into its flat moment buffers at the next ``step``."""
    ckpt = load_lightning_checkpoint(path)
    module.load_state_dict(ckpt["state_dict"], strict=strict)
    if optimizer is not None and ckpt.get("optimizer_states"):
        optimizer.load_state_dict(ckpt["optimizer_states"][0])
    return ckpt

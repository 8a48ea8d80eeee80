"""Batched, device-side form of the reference's augmentation protocol (reference augmentations/augmentation.py:6-157).

The reference draws ONE parameter set per call and transforms one item (image (1, H, W), mask (H, W), key points (K, 2)) on a
CPU worker.  Here an ``Augmentation`` draws one parameter set PER ITEM of a batch (``get_params(n)`` -> dict of (n,) float32
tensors, drawn exactly as the reference draws them: ``random.uniform`` / ``random.randint`` / ``torch.empty(1).uniform_`` item
by item) and every transform contributes its columns to a parameter table (n, 8) =
{angle (deg), tx, ty, brightness, contrast, gamma, 0, 0}; ``Compose`` of the data module's four transforms then runs ONE fused
kernel launch pair per batch for the images (``cu_augment_image``), one for the label maps (``cu_augment_labels``) and plain
tensor arithmetic for the 21 x 2 key points.  ``apply`` / ``un_apply`` keep the reference's dict protocol ("image", "mask",
"keypoints") and remember the parameters for the un-apply of test-time augmentation.

A ``Compose`` whose members are not in the reference order (rotation -> brightness/contrast -> gamma -> translation, the only
order the reference uses: data/camus/datamodule.py:46-55) applies its members one after the other (one launch pair each)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

IDENTITY = (0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 0.0, 0.0)      # angle, tx, ty, brightness, contrast, gamma
COL = {"angle": 0, "tx": 1, "ty": 2, "alpha": 3, "beta": 4, "gamma": 5}


def identity_table(n: int, device) -> torch.Tensor:
    return torch.tensor(IDENTITY, dtype=torch.float32, device=device).repeat(n, 1)


def _n_items(items: Dict[str, torch.Tensor]) -> int:
    return next(iter(items.values())).shape[0]


class Augmentation:
    def __init__(self):
        self.params: Optional[Dict[str, torch.Tensor]] = None       # kept for un_apply

    # ---- per-transform pieces -------------------------------------------------------------------------------------
    def get_params(self, n: int = 1) -> Dict[str, torch.Tensor]:
        return {}

    def fill(self, table: torch.Tensor, params: Dict[str, torch.Tensor], sign: float = 1.0):
        """write this transform's parameters into the (n, 8) table (``sign`` -1: the inverse geometric transform)"""

    geometric = False       # rotation / translation: also acts on masks and key points
    order = 0               # position in the fused kernel's pipeline (1 rotation, 2 colour, 3 gamma, 4 translation)

    def apply_keypoints(self, keypoints: torch.Tensor, params, sign: float = 1.0) -> torch.Tensor:
        return keypoints

    # ---- the reference's protocol ----------------------------------------------------------------------------------
    def __call__(self, image=None, mask=None, keypoints=None, *args, **kwargs):
        items = {k: v for k, v in (("image", image), ("mask", mask), ("keypoints", keypoints)) if v is not None}
        return self.apply(items)

    def _run(self, items: Dict[str, torch.Tensor], members: Sequence["Augmentation"], plist, sign: float):
        from cu_hip import ops
        n = _n_items(items)
        out = dict(items)
        dev = next(iter(items.values())).device
        table = identity_table(n, dev)
        for m, pr in zip(members, plist):
            m.fill(table, pr, sign)
        if "image" in out:
            img = out["image"]
            if sign < 0:      # un-apply: the colour transforms are not undone (reference: un_apply_img returns the image)
                table[:, 3:6] = 1.0
            out["image"] = ops.augment_image(img.float(), table)
        if "mask" in out:
            out["mask"] = ops.augment_labels(out["mask"].long(), table)
        if "keypoints" in out:
            kp = out["keypoints"]
            for m, pr in (zip(members, plist) if sign > 0 else reversed(list(zip(members, plist)))):
                kp = m.apply_keypoints(kp, pr, sign)
            out["keypoints"] = kp
        return out

    def apply(self, items: Dict[str, torch.Tensor], params=None) -> Dict[str, torch.Tensor]:
        if params is None:
            params = self.get_params(_n_items(items))
        self.params = params
        return self._run(items, [self], [params], 1.0)

    def un_apply(self, items: Dict[str, torch.Tensor], params=None) -> Dict[str, torch.Tensor]:
        params = self.params if params is None else params
        assert params is not None
        return self._run(items, [self], [params], -1.0)


class Compose(Augmentation):
    def __init__(self, transforms: List[Augmentation]):
        super().__init__()
        self.transforms = transforms

    def get_params(self, n: int = 1) -> List[Dict[str, torch.Tensor]]:
        return [t.get_params(n) for t in self.transforms]

    def _fusable(self, sign: float) -> bool:
        """one kernel pass = rotation, then colour, then gamma, then translation (each at most once); its inverse
        (translation first) is NOT that pipeline, so un-apply runs the members one by one in reverse"""
        orders = [t.order for t in self.transforms]
        return sign > 0 and orders == sorted(orders) and len(set(orders)) == len(orders) and all(orders)

    def apply(self, items, params: Optional[List[Dict]] = None):
        if params is None:
            params = self.get_params(_n_items(items))
        assert len(params) == len(self.transforms)
        self.params = params
        if self._fusable(1.0):
            return self._run(items, self.transforms, params, 1.0)
        for t, pr in zip(self.transforms, params):
            items = t.apply(items, pr)
        return items

    def un_apply(self, items, params: Optional[List[Dict]] = None):
        params = self.params if params is None else params
        assert params is not None and len(params) == len(self.transforms)
        for t, pr in list(zip(self.transforms, params))[::-1]:
            items = t.un_apply(items, pr)
        return items


def to_tuple(param, low=None, bias=None):
    """scalar -> (-v, +v) (or (low, v)); sequence -> tuple; + bias (the reference's helper, augmentation.py:130-157)"""
    if low is not None and bias is not None:
        raise ValueError("Arguments low and bias are mutually exclusive")
    if param is None:
        return param
    if isinstance(param, (int, float)):
        param = (-param, +param) if low is None else ((low, param) if low < param else (param, low))
    elif isinstance(param, Sequence):
        param = tuple(param)
    else:
        raise ValueError("Argument param must be either scalar (int, float) or tuple")
    return tuple(bias + x for x in param) if bias is not None else tuple(param)

"""TEST INFRASTRUCTURE (never imported by the product): CPU restatement of the reference's augmentations.

Reference: contour_uncertainty/augmentations/{affine,brightnesscontrast,gamma}.py, which call
``torchvision.transforms.functional`` (rotate, affine, adjust_brightness, adjust_contrast, adjust_gamma).  torchvision is a
third-party dependency that is ABSENT from this image (the reference's requirements do not pin it; it follows the pinned
torch ~1.12 -> torchvision 0.13), so its published algorithm is restated here on the very torch primitives it is built from
(torchvision/transforms/functional.py: ``_get_inverse_affine_matrix``; functional_tensor.py: ``_gen_affine_grid``,
``_apply_grid_transform`` = ``torch.nn.functional.grid_sample(mode='nearest', padding_mode='zeros', align_corners=False)``,
``_blend``, ``adjust_gamma``).  PARITY UNPINNED: the reference holds no test or golden vector for its augmentations and
torchvision cannot be run here, so nothing but this restatement (and the key-point formulas, which are the reference's own
torch code, restated line by line) stands behind the HIP kernels' expected values."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _inverse_affine(angle: float, translate, center=(0.0, 0.0)):
    """torchvision ``_get_inverse_affine_matrix(center, angle, translate, scale=1, shear=(0, 0))``"""
    rot = math.radians(angle)
    cx, cy = center
    tx, ty = translate
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _grid_transform(img: torch.Tensor, matrix) -> torch.Tensor:
    """img (C, H, W) float; torchvision ``_gen_affine_grid`` + ``grid_sample`` (nearest, zeros)"""
    c, h, w = img.shape
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, h, w, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    return F.grid_sample(img[None].float(), grid, mode="nearest", padding_mode="zeros", align_corners=False)[0]


def rotate(img: torch.Tensor, angle: float) -> torch.Tensor:
    """``F.rotate(img, angle)`` (nearest, no expand, zero fill): matrix of the inverse of a rotation by ``angle``"""
    return _grid_transform(img, _inverse_affine(-angle, (0.0, 0.0)))


def translate(img: torch.Tensor, tx: int, ty: int) -> torch.Tensor:
    """``F.affine(img, angle=0, translate=[tx, ty], scale=1, shear=0)``"""
    return _grid_transform(img, _inverse_affine(0.0, (float(tx), float(ty))))


def _blend(a: torch.Tensor, b: torch.Tensor, ratio: float) -> torch.Tensor:
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0)


def adjust_brightness(img, factor):
    return _blend(img, torch.zeros_like(img), factor)


def adjust_contrast(img, factor):
    mean = torch.mean(img.float(), dim=(-3, -2, -1), keepdim=True)        # one-channel image: its own mean
    return _blend(img, mean, factor)


def adjust_gamma(img, gamma, gain=1.0):
    return (gain * img ** gamma).clamp(0, 1)


def compose_image(img: torch.Tensor, angle, alpha, beta, gamma, tx, ty) -> torch.Tensor:
    """the data module's Compose on one (1, H, W) image (reference datamodule.py:46-55, brightnesscontrast.py:14-18)"""
    x = rotate(img, angle)
    x = adjust_contrast(adjust_brightness(x, alpha), beta)
    x = adjust_gamma(x, gamma)
    return translate(x, tx, ty)


def compose_mask(mask: torch.Tensor, angle, tx, ty) -> torch.Tensor:
    """(H, W) integer mask: rotate then translate (affine.py:18-23,73-78), nearest"""
    x = rotate(mask[None].float(), angle)
    return translate(x, tx, ty)[0].round().to(mask.dtype)


def rotate_keypoints(kp: torch.Tensor, angle: float, image_shape=(256, 256)) -> torch.Tensor:
    """reference affine.py:43-58"""
    ox, oy = image_shape[1] / 2, image_shape[0] / 2
    ax, ay = kp[..., 0] - ox, kp[..., 1] - oy
    a = torch.deg2rad(torch.tensor(angle))
    c, s = torch.cos(a), torch.sin(a)
    out = torch.zeros_like(kp)
    out[..., 0] = ox + c * ax + s * ay
    out[..., 1] = oy + -s * ax + c * ay
    return out


def translate_keypoints(kp: torch.Tensor, tx, ty) -> torch.Tensor:
    out = torch.clone(kp)
    out[..., 0] += tx
    out[..., 1] += ty
    return out

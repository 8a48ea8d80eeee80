"""On-device augmentation (cu_augment_image / cu_augment_labels, contour_uncertainty/augmentations) against the CPU oracle
(oracle/augment.py: restated torchvision ops, parity unpinned) item by item, and through the data module's hook."""
import random

import pytest
import torch

from oracle import augment as OA

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _compose(size):
    from contour_uncertainty.augmentations import (Compose, RandomBrightnessContrast, RandomGamma, RandomRotation,
                                                   RandomTranslation)
    return Compose([RandomRotation(3, (size, size)), RandomBrightnessContrast(0.2, 0.2), RandomGamma((0.8, 1.2)),
                    RandomTranslation(5, 5)])


@pytest.mark.parametrize("size", [64, 256])
def test_fused_batch_augmentation_equals_the_oracle_item_by_item(size):
    random.seed(11); torch.manual_seed(11)
    n = 6
    g = torch.Generator().manual_seed(5)
    img = torch.rand(n, 1, size, size, generator=g)
    gt = (torch.rand(n, size, size, generator=g) > 0.5).long() * torch.randint(1, 3, (n, 1, 1), generator=g)
    kp = torch.rand(n, 21, 2, generator=g) * (size - 1)
    c = _compose(size)
    out = c(image=img.to(DEV), mask=gt.to(DEV), keypoints=kp.to(DEV))
    p = c.params
    for i in range(n):
        a, al, be, ga = float(p[0]["angle"][i]), float(p[1]["alpha"][i]), float(p[1]["beta"][i]), float(p[2]["gamma"][i])
        tx, ty = int(p[3]["tx"][i]), int(p[3]["ty"][i])
        ref = OA.compose_image(img[i], a, al, be, ga, tx, ty)
        got = out["image"][i].cpu()
        diff = (got - ref).abs()
        # nearest sampling: a source coordinate within float rounding of .5 may pick the neighbouring pixel (random image:
        # any value); everything else agrees to float accuracy (powf / mean summation order)
        assert float((diff > 1e-4).float().mean()) < 2e-3, (i, float((diff > 1e-4).float().mean()))
        refm = OA.compose_mask(gt[i], a, tx, ty)
        assert float((out["mask"][i].cpu() != refm).float().mean()) < 2e-3
        refk = OA.translate_keypoints(OA.rotate_keypoints(kp[i], a, (size, size)), tx, ty)
        assert torch.allclose(out["keypoints"][i].cpu(), refk, atol=1e-3)
    # un-apply (test-time augmentation): the geometry comes back except for what left the image; colours are not undone
    back = c.un_apply({"mask": out["mask"], "keypoints": out["keypoints"]})
    assert torch.allclose(back["keypoints"].cpu(), kp, atol=2e-3)
    inner = slice(size // 4, 3 * size // 4)
    assert float((back["mask"].cpu()[:, inner, inner] != gt[:, inner, inner]).float().mean()) < 0.05


def test_identity_parameters_return_the_input():
    from cu_hip import ops
    from contour_uncertainty.augmentations.augmentation import identity_table
    img = torch.rand(3, 1, 128, 96, device=DEV)
    assert torch.equal(ops.augment_image(img, identity_table(3, DEV)), img)
    lab = torch.randint(0, 4, (3, 128, 96), device=DEV)
    assert torch.equal(ops.augment_labels(lab, identity_table(3, DEV)), lab)


def test_datamodule_hook_augments_training_batches_on_the_device():
    from contour_uncertainty.data.synthetic import SyntheticContourDataModule
    dm = SyntheticContourDataModule(size=64, batch_size=4, n_train=4, da=True)
    dm.setup("fit")
    batch = next(iter(dm.train_dataloader()))
    dev_batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    random.seed(0); torch.manual_seed(0)
    out = dm.on_after_batch_transfer(dev_batch, 0)
    assert out["img"].shape == batch["img"].shape and out["img"].is_cuda and out["gt"].dtype == torch.int64
    assert float(out["img"].min()) >= 0.0 and float(out["img"].max()) <= 1.0
    assert not torch.equal(out["img"].cpu(), batch["img"]) and not torch.allclose(out["contour"].cpu(), batch["contour"])
    moved = (out["contour"].cpu() - batch["contour"]).abs().max()
    assert float(moved) < 5 + 64 * 0.06 + 1            # +-5 px translation, +-3 degrees about the centre


def test_datamodule_hook_leaves_validation_batches_alone():
    """The reference augments its TRAINING subset only (data/camus/datamodule.py:46-55).  Lightning calls
    ``on_after_batch_transfer`` for every stage; outside training (``trainer.training`` False) the batch must come back
    untouched."""
    from types import SimpleNamespace
    from contour_uncertainty.data.synthetic import SyntheticContourDataModule
    dm = SyntheticContourDataModule(size=64, batch_size=4, n_train=4, n_val=4, da=True)
    dm.setup("fit")
    batch = next(iter(dm.val_dataloader()))
    dev_batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    dm.trainer = SimpleNamespace(training=False)
    out = dm.on_after_batch_transfer(dev_batch, 0)
    assert out is dev_batch
    dm.trainer = SimpleNamespace(training=True)
    out = dm.on_after_batch_transfer(dev_batch, 0)
    assert not torch.equal(out["img"], dev_batch["img"])

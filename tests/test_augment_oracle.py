"""CPU: the augmentation oracle (restated torchvision functional ops, PARITY UNPINNED -- see oracle/augment.py) obeys the
properties its algorithm implies, and the product's host-side halves (parameter draws, key-point transforms, the fused
parameter table) agree with it."""
import random

import torch

from oracle import augment as OA


def test_oracle_identities_and_shifts():
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 48, 64, generator=g)
    assert torch.equal(OA.rotate(img, 0.0), img) and torch.equal(OA.translate(img, 0, 0), img)
    t = OA.translate(img, 3, -2)                       # content moves +3 in x, -2 in y; zero fill behind it
    assert torch.equal(t[0, 0:46, 3:], img[0, 2:, :61]) and float(t[0, 46:].abs().max()) == 0 and float(t[0, :, :3].abs().max()) == 0
    r = OA.rotate(img, 180.0)                          # exact flip of both axes (nearest, centre between pixels)
    assert torch.equal(r, img.flip(-1, -2))
    sq = torch.rand(1, 32, 32, generator=g)
    assert torch.equal(OA.rotate(sq, 90.0), sq.rot90(1, (-2, -1)))        # counter-clockwise for positive angles
    assert torch.allclose(OA.adjust_gamma(img, 1.0), img) and torch.allclose(OA.adjust_brightness(img, 1.0), img)
    assert torch.allclose(OA.adjust_contrast(img, 1.0), img)
    assert torch.allclose(OA.adjust_contrast(img, 0.0), img.mean().expand_as(img))


def test_keypoints_follow_the_image():
    """a bright pixel and its key point land on the same place after rotate + translate (keypoint centre = size / 2 as in the
    reference, affine.py:44: half a pixel off the image centre the grid uses -- so within one pixel)"""
    img = torch.zeros(1, 256, 256)
    img[0, 100, 180] = 1.0
    kp = torch.tensor([[180.0, 100.0]])
    out = OA.translate(OA.rotate(img, 3.0), 4, -5)
    yx = torch.nonzero(out[0])
    q = OA.translate_keypoints(OA.rotate_keypoints(kp, 3.0), 4, -5)
    assert len(yx) >= 1 and float((yx[0].flip(0).float() - q[0]).abs().max()) <= 1.0


def test_product_parameter_draws_and_keypoints_match_the_reference_formulas():
    from contour_uncertainty.augmentations import (Compose, RandomBrightnessContrast, RandomGamma, RandomRotation,
                                                   RandomTranslation)
    from contour_uncertainty.augmentations.augmentation import identity_table
    c = Compose([RandomRotation(3), RandomBrightnessContrast(0.2, 0.2), RandomGamma((0.8, 1.2)), RandomTranslation(5, 5)])
    random.seed(3); torch.manual_seed(3)
    p = c.get_params(5)
    assert all(-3 <= a <= 3 for a in p[0]["angle"].tolist()) and all(0.8 <= v <= 1.2 for v in p[1]["alpha"].tolist())
    assert all(float(v).is_integer() and -5 <= v <= 5 for v in p[3]["tx"].tolist() + p[3]["ty"].tolist())
    # the draws are the reference's own calls in the reference's order (one item at a time per transform)
    random.seed(3); torch.manual_seed(3)
    ang = [float(torch.empty(1).uniform_(-3.0, 3.0).item()) for _ in range(5)]
    ab = [(1.0 + random.uniform(-0.2, 0.2), 1.0 + random.uniform(-0.2, 0.2)) for _ in range(5)]
    gm = [random.uniform(0.8, 1.2) for _ in range(5)]
    tt = [(random.randint(-5, 5), random.randint(-5, 5)) for _ in range(5)]
    assert torch.allclose(p[0]["angle"], torch.tensor(ang)) and torch.allclose(p[1]["alpha"], torch.tensor([a for a, _ in ab]))
    assert torch.allclose(p[2]["gamma"], torch.tensor(gm)) and p[3]["tx"].tolist() == [float(t[0]) for t in tt]
    # fused table and key points
    table = identity_table(5, "cpu")
    for t, pr in zip(c.transforms, p):
        t.fill(table, pr)
    assert torch.allclose(table[:, 0], p[0]["angle"]) and torch.allclose(table[:, 5], p[2]["gamma"]) and c._fusable(1.0)
    kp = torch.rand(5, 21, 2) * 255
    got = kp
    for t, pr in zip(c.transforms, p):
        got = t.apply_keypoints(got, pr)
    for i in range(5):
        ref = OA.translate_keypoints(OA.rotate_keypoints(kp[i], float(p[0]["angle"][i])), float(p[3]["tx"][i]), float(p[3]["ty"][i]))
        assert torch.allclose(got[i], ref, atol=1e-4)
    back = got
    for t, pr in list(zip(c.transforms, p))[::-1]:
        back = t.apply_keypoints(back, pr, -1.0)
    assert torch.allclose(back, kp, atol=1e-3)          # un-apply of the key points inverts apply

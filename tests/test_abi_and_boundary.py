"""CPU-side checks: the C-ABI library loads and exports every symbol include/contour_hip.h declares (no compute calls),
the drop-in classes keep the reference's names / shapes / hparams surface, and the product path fails loudly without a
GPU instead of falling back."""
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import unet as OU

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "contour_hip.h").read_text()
    return sorted(set(re.findall(r"\b(cu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cu_hip import lib
    handle = lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), f"libcontour_hip.so lacks {n}"
    assert sorted(lib.exported_symbols()) == names, "ctypes table and header disagree"
    assert handle.cu_arch().decode() == "gfx950"
    assert handle.cu_version() >= 100


def test_bad_arguments_return_error_codes_not_crashes():
    from cu_hip import lib
    h = lib.load()
    assert h.cu_conv_gemm(None, None, None, None, None, None, None, None, None, None, None, None) == -22
    assert b"null descriptor" in h.cu_last_error()
    d = lib.ConvDesc()
    d.dtype = 7
    assert h.cu_conv_gemm(d, None, None, None, None, None, None, None, None, None, None, None) == -22
    assert h.cu_dsnt_head_fwd(4, 16, 24, None, 1, None, None, None, None) == -22      # non-square map (utils.py:9)
    assert b"square" in h.cu_last_error()


def _task(kind="dsnt-skew", stages=6, size=64):
    from contour_uncertainty._compat import DataParameters
    from contour_uncertainty.task.regression.dsnt.dsnt_al import DSNTAleatoric
    from contour_uncertainty.task.regression.dsnt.dsnt_skew import DSNTSkew
    cfg = {"_target_": "contour_uncertainty.models.nnUnet.unet2.UNet", "kernels": [[3, 3]] * stages,
           "strides": [[1, 1]] + [[2, 2]] * (stages - 1), "patch_size": [256, 256], "drop_block": False,
           "deep_supervision": False}
    cls = DSNTSkew if kind == "dsnt-skew" else DSNTAleatoric
    return cls(model=cfg, optim={"_target_": "torch.optim.Adam", "lr": 1e-3, "weight_decay": 1e-3}, choices={"task": kind},
               data_params=DataParameters((1, size, size), (21, 2), [0, 1]), psm_path="camus-cont_psm_11_no_std.npy",
               seq_psm_path="camus-cont_sequence_psm_11_no_std.npy", sequence_sampler=False, t_a=25, t_e=1,
               log_figures=False, task_name="x", some_future_yaml_key=123)


def test_state_dict_matches_reference_names_and_shapes(golden_dir):
    task = _task("dsnt-skew", stages=8, size=256)
    sd = task.state_dict()
    ref_names = [str(n) for n in np.load(golden_dir / "unet_full.npz")["param_names"]]
    assert [k[len("model."):] for k in sd if k.startswith("model.")] == ref_names
    shapes = OU.param_shapes(OU.UNetSpec())
    for k, shp in shapes.items():
        assert tuple(sd["model." + k].shape) == tuple(shp), k
    skew = {k[len("skew_block."):]: tuple(v.shape) for k, v in sd.items() if k.startswith("skew_block.")}
    assert skew == {k: tuple(v) for k, v in OU.confidence_param_shapes(42).items()}
    assert sum(v.numel() for k, v in sd.items() if k.startswith("model.")) == 41298912
    # reference-shaped weights load strictly
    g = torch.Generator().manual_seed(0)
    task.model.load_state_dict(OU.init_unet_state(OU.UNetSpec(), g), strict=True)


def test_hparams_and_optimizer_surface():
    task = _task("dsnt-al")
    hp = task.hparams
    for key in ("model", "optim", "choices", "data_params", "covar", "mse_weight", "log_penalty_weight", "iterations",
                "psm_path", "seq_psm_path", "sequence_sampler", "t_a", "t_e", "train_ensemble", "ensemble_ckpt",
                "log_figures", "task_name", "some_future_yaml_key"):
        assert key in hp, key
    assert hp.covar is True and hp.t_a == 25 and hp.some_future_yaml_key == 123
    opt = task.configure_optimizers()
    assert set(opt.keys()) == {"optimizer"}
    from cu_hip.optim import FusedAdam
    assert isinstance(opt["optimizer"], FusedAdam)
    assert opt["optimizer"].defaults["lr"] == 1e-3 and opt["optimizer"].defaults["weight_decay"] == 1e-3
    s = task.get_cov_matrix(torch.ones(2, 21), 2 * torch.ones(2, 21), 0.5 * torch.ones(2, 21))
    assert s.shape == (2, 21, 2, 2) and float(s[0, 0, 0, 1]) == 0.5 and float(s[0, 0, 1, 1]) == 2.0


def test_unsupported_reference_options_are_refused_not_ignored():
    from contour_uncertainty.models.nnUnet.unet2 import UNet
    for kw in ({"attention": True}, {"residual": True}, {"deep_supervision": True}, {"ssn_rank": 5},
               {"normalization_layer": "batch"}):
        with pytest.raises(NotImplementedError):
            UNet((1, 64, 64), (21, 1, 64), [256, 256], [[3, 3]] * 6, [[1, 1]] + [[2, 2]] * 5, **kw)


def test_drop_block_layer_set_matches_reference(golden_dir):
    """task.model.drop_block=True: the ConvLayers that get a Dropout2d are exactly the ones the instantiated reference
    module flags (tests/golden/drop_block_layers.json, written by oracle/make_golden.py `drop` from the reference's
    use_drop_block attributes; unet2.py:302 selects only the last downsample block, :129-136 the bottleneck)."""
    import json
    from contour_uncertainty.models.nnUnet.unet2 import UNet
    ref = json.loads((golden_dir / "drop_block_layers.json").read_text())
    for n, layers in ref.items():
        n = int(n)
        net = UNet((1, 64, 64), (21, 1, 64), [256, 256], [[3, 3]] * n, [[1, 1]] + [[2, 2]] * (n - 1), drop_block=True)
        assert sorted(net.engine.drop_layers) == layers
        flagged = sorted(name for name, m in net.named_modules() if getattr(m, "use_drop_block", False))
        assert flagged == layers
    plain = UNet((1, 64, 64), (21, 1, 64), [256, 256], [[3, 3]] * 6, [[1, 1]] + [[2, 2]] * 5, drop_block=False)
    assert not plain.engine.drop_layers


def test_no_cpu_fallback():
    """The product path must fail loudly when there is no GPU (or the tensor is not on it)."""
    from cu_hip.lib import ContourHipError
    task = _task("dsnt-al")
    batch = {"img": torch.rand(2, 1, 64, 64), "contour": torch.rand(2, 21, 2) * 60}
    with pytest.raises(ContourHipError):
        task.training_step(batch, 0)


def test_product_package_never_imports_the_oracle():
    pkg = ROOT / "contouring-uncertainty_amd"
    for f in pkg.rglob("*.py"):
        text = f.read_text()
        assert "import oracle" not in text and "from oracle" not in text, f


def test_validation_mask_and_dice_helpers():
    from contour_uncertainty.utils.contour import contour_to_mask, linear_reconstruction
    from contour_uncertainty.utils.metrics import Dice
    t = np.linspace(0, 2 * np.pi, 21, endpoint=False)
    c = np.stack([32 + 12 * np.cos(t), 30 + 9 * np.sin(t)], -1)
    m = linear_reconstruction(c, (64, 64))
    area = np.pi * 12 * 9
    assert abs(m.sum() - area) / area < 0.08
    assert m[30, 32] and not m[5, 5]
    d = Dice(labels=[0, 1])
    assert d(m.astype(int)[None], m.astype(int)[None]) == 1.0
    assert contour_to_mask(c, (64, 64), apply_argmax=False).shape == (1, 64, 64)


def test_mask_and_umap_paths_have_no_cpu_fallback():
    """Rasterisation goes through cu_contour_masks or fails: no scipy fallback hides behind the drop-in functors; the
    branches this build does not serve raise instead of returning something else."""
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    from contour_uncertainty.data.camus.utils import USContourToMask, USSkewUmap, USUMap
    from contour_uncertainty.utils.contour import reconstruction
    t = np.linspace(0, np.pi, 21)
    c = np.stack([128 + 50 * np.cos(t), 170 - 80 * np.sin(t)], -1).astype(np.float32)
    for call in (lambda: reconstruction(c, 256, 256), lambda: USContourToMask()(c, (256, 256), [0, 1]),
                 lambda: USContourToMask.batch(c[None], (256, 256), [0, 1])):
        with pytest.raises(Exception) as e:
            call()
        assert "CUDA" in str(e.value) or "cuda" in str(e.value) or "HIP" in str(e.value) or "GPU" in str(e.value)
    # host-only pieces keep working without a GPU: linear reconstruction (validation Dice), projection
    assert USContourToMask()(c, (256, 256), [0, 1], reconstruction_type="linear").sum() > 1000
    # the LV + MYO branch (reference data/camus/utils.py:48-82) is host code: with the linear reconstruction it needs no GPU
    t2 = np.linspace(0, np.pi, 21)
    epi = np.stack([128 + 70 * np.cos(t2), 170 - 105 * np.sin(t2)], -1).astype(np.float32)
    seg = USContourToMask()(np.concatenate([c, epi]), (256, 256), [0, 1, 2], reconstruction_type="linear")
    assert seg.shape == (256, 256) and set(np.unique(seg)) == {0, 1, 2}
    assert seg[150, 128] == 1 and seg[150, 128 + 60] == 2 and seg[150, 128 - 60] == 2 and seg[20, 20] == 0
    assert (seg == 2).sum() > 0.5 * (seg == 1).sum()
    three = USContourToMask()(np.concatenate([c, epi]), (256, 256), [0, 1, 2], apply_argmax=False, reconstruction_type="linear")
    assert three.shape == (3, 256, 256) and int(three.sum(0).min()) == 1 and int(three.sum(0).max()) == 1


def test_polygon_fill_agrees_with_an_independent_point_in_polygon_test():
    """``polygon_mask`` restates skimage.draw.polygon (absent from this image); matplotlib's Path.contains_points is an
    independent implementation: the two may differ only on pixels whose centre lies ON the boundary (skimage includes
    them), i.e. within one pixel of an edge."""
    from matplotlib.path import Path as MplPath
    from scipy.ndimage import binary_dilation, binary_erosion
    from contour_uncertainty.utils.contour import polygon_mask
    rng = np.random.default_rng(3)
    for _ in range(3):
        ang = np.sort(rng.uniform(0, 2 * np.pi, 40))
        rad = rng.uniform(30, 90, 40)
        rows, cols = np.rint(128 + rad * np.sin(ang)).astype(int), np.rint(120 + rad * np.cos(ang)).astype(int)
        got = polygon_mask(rows, cols, (256, 256)).astype(bool)
        yy, xx = np.mgrid[0:256, 0:256]
        ref = MplPath(np.stack([cols, rows], 1)).contains_points(np.stack([xx.ravel(), yy.ravel()], 1)).reshape(256, 256)
        band = binary_dilation(ref, iterations=1) & ~binary_erosion(ref, iterations=1)
        odd = (got != ref) & ~band                         # only the tips of thin spikes: boundary pixels next to a vertex
        ys, xs = np.nonzero(odd)
        assert got[ys, xs].all() and all(np.min(np.hypot(rows - y, cols - x)) <= 1.5 for y, x in zip(ys, xs))
        assert got.sum() >= ref.sum() and (got != ref).sum() < 0.01 * ref.sum()
        assert got[rows, cols].all()                       # vertices belong to the polygon


def test_train_ensemble_takes_a_random_90_percent_subset():
    """reference task/uncertainty.py:76-80: on_fit_start swaps the TRAIN dataset for a 90 % random Subset; a missing
    datamodule is an error, not a silent no-op."""
    import random
    from enum import Enum
    from torch.utils.data import Subset as TorchSubset

    class Sub(Enum):
        TRAIN = "train"
        VAL = "val"

    class DM:
        def __init__(self):
            self._dataset = {Sub.TRAIN: list(range(50)), Sub.VAL: list(range(7))}
            self.umap_fn = self.contour_to_mask_fn = self.skew_umap_fn = staticmethod(lambda *a, **k: None)

    class Trainer:
        datamodule = DM()
        logger = None

    task = _task("dsnt-al")
    task.hparams.train_ensemble = True
    task.trainer = Trainer()
    random.seed(3)
    task.on_fit_start()
    sub = Trainer.datamodule._dataset[Sub.TRAIN]
    assert isinstance(sub, TorchSubset) and len(sub) == 45 and len(set(sub.indices)) == 45
    assert len(Trainer.datamodule._dataset[Sub.VAL]) == 7
    task2 = _task("dsnt-al")
    task2.hparams.train_ensemble = True
    with pytest.raises(RuntimeError):
        task2.on_fit_start()
    task3 = _task("dsnt-al")            # default: untouched
    task3.trainer = Trainer()
    before = Trainer.datamodule._dataset[Sub.TRAIN]
    task3.on_fit_start()
    assert Trainer.datamodule._dataset[Sub.TRAIN] is before


def test_product_library_has_no_tuning_hooks():
    """The shipped library must not honour timing-experiment environment variables: they are compiled
    in only under `make TUNING=1` (-DCU_TUNING), so their names do not even appear in the product .so."""
    from cu_hip import lib
    blob = Path(lib.LIB_PATH).read_bytes()
    for name in (b"CU_CONV_DBG", b"CU_CONV_NODMA", b"CU_CONV_NBMAX", b"CU_CONV_NO_NARROW", b"CU_WGRAD_NODMA",
                 b"CU_WGRAD_NW", b"CU_WGRAD_PC", b"CU_MASKS_DBG"):
        assert name not in blob, name


def test_synthetic_inputs_of_product_and_oracle_agree():
    """bench.py takes its inputs from the product's data/synthetic module; the oracle keeps its own statement of SURVEY
    8(d)'s generator.  Same seed -> same tensors, and the FLOP model of cu_hip.flops equals the oracle's layer count."""
    from contour_uncertainty.data.synthetic import synthetic_batch
    from cu_hip.flops import conv_macs_per_image
    from oracle.step import synthetic_batch as oracle_batch
    for n, size, seed in ((2, 64, 1234), (3, 32, 7)):
        a, b = synthetic_batch(n, size, 21, seed), oracle_batch(n, size, 21, seed)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for strides, size in (((1, 2, 2, 2, 2, 2, 2, 2), 256), ((1, 2, 2, 2, 2, 2), 64)):
        a, b = conv_macs_per_image(strides, size), OU.conv_macs_per_image(OU.UNetSpec(strides=strides), size)
        assert a == b
    assert abs(2 * conv_macs_per_image((1, 2, 2, 2, 2, 2, 2, 2), 256)["fwd"] - 28.98e9) < 0.01e9


def test_synthetic_datamodule_contract():
    """data=synthetic emits the reference's batch contract (dataset.py:100-149; predict items are whole views)."""
    from contour_uncertainty.data.synthetic import SyntheticContourDataModule
    dm = SyntheticContourDataModule(size=64, batch_size=2, n_train=5, n_val=3, n_predict=2)
    assert dm.data_params.in_shape == (1, 64, 64) and dm.data_params.out_shape == (21, 2)
    dm.setup("fit")
    b = next(iter(dm.train_dataloader()))
    assert b["img"].shape == (2, 1, 64, 64) and b["img"].dtype == torch.float32
    assert 0 <= float(b["img"].min()) and float(b["img"].max()) <= 1
    assert b["contour"].shape == (2, 21, 2) and b["gt"].shape == (2, 64, 64) and b["gt"].dtype == torch.int64
    assert len(b["id"]) == 2 and sum(len(x["id"]) for x in dm.train_dataloader()) == 5
    # the contour's x is the column, y the row: the landmarks lie on the boundary of the gt mask
    c = b["contour"][0].round().long()
    assert int(b["gt"][0][c[10, 1] + 3, c[10, 0]]) == 1          # just below the apex: inside
    dm.setup("predict")
    v = next(iter(dm.predict_dataloader()))
    assert v["img"].shape == (2, 1, 64, 64) and v["contour"].shape == (2, 21, 2) and isinstance(v["id"], str)
    # deterministic
    dm2 = SyntheticContourDataModule(size=64, batch_size=2, n_train=5, n_val=3)
    dm2.setup("fit")
    assert torch.equal(dm2.datasets["val"][1]["img"], dm.datasets["val"][1]["img"])


def test_parameter_list_cache_follows_replaced_parameters():
    """``UNet._params()`` / ``ConfidenceNet._params()`` cache the Parameter objects; replacing one without
    ``_apply`` (``load_state_dict(assign=True)``, ``conv.weight = nn.Parameter(...)``) must be noticed on the next call."""
    import torch
    from contour_uncertainty.models.nnUnet.unet2 import ConfidenceNet, UNet
    net = UNet((1, 64, 64), (21, 1, 64), [256, 256], [[3, 3]] * 4, [[1, 1]] + [[2, 2]] * 3)
    first = net._params()[0][0]
    assert net._params()[0][0] is first
    net.input_block.conv1.conv.weight = torch.nn.Parameter(torch.zeros_like(net.input_block.conv1.conv.weight))
    assert net._params()[0][0] is net.input_block.conv1.conv.weight and net._params()[0][0] is not first
    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()}, assign=True)
    named = dict(net.named_parameters())
    assert all(p is named[n] for p, n in zip(net._params()[0], net._pnames))
    head = ConfidenceNet(42)
    head._params()
    head.load_state_dict({k: v.clone() for k, v in head.state_dict().items()}, assign=True)
    assert head._params()[0] is head.model[0].weight

"""SURVEY.md 8f rank 4 (checkpoint half): Lightning-1.8-layout ``.ckpt`` files without Lightning.

  * a checkpoint written from the REFERENCE's module structure (tests/golden/ref_ckpt_tiny.ckpt, oracle/make_golden.py ckpt:
    the reference's own UNet inside a ``model`` attribute, stepped once with torch.optim.Adam) loads with ``strict=True``
    into this package's task, its optimizer state loads into the fused Adam, and (GPU) the network reproduces the
    reference's logits;
  * the 8-stage dsnt-skew task's ``state_dict`` names / order / shapes / dtypes and the layout of its optimizer state equal the
    reference's (tests/golden/ref_ckpt_structure.json);
  * what ``save_lightning_checkpoint`` writes has every key of a Lightning checkpoint and round-trips (weights, Adam moments,
    step counts, epoch / global_step), also through ``torch.optim.Adam`` -- the interchange the reference's ``weights=`` /
    resume paths need (reference vital/vital/runner.py:113-120);
  * a checkpoint whose pickled classes are not importable (a stand-in for omegaconf.DictConfig hyper-parameters) still loads.
"""
import json
import sys
import types

import numpy as np
import pytest
import torch


def _task(kind, stages, size, dtype="f32"):
    from contour_uncertainty._compat import DataParameters
    from contour_uncertainty.task.regression.dsnt.dsnt_al import DSNTAleatoric
    from contour_uncertainty.task.regression.dsnt.dsnt_skew import DSNTSkew
    cfg = {"_target_": "contour_uncertainty.models.nnUnet.unet2.UNet", "kernels": [[3, 3]] * stages,
           "strides": [[1, 1]] + [[2, 2]] * (stages - 1), "patch_size": [256, 256], "compute_dtype": dtype}
    cls = DSNTSkew if kind == "dsnt-skew" else DSNTAleatoric
    return cls(model=cfg, optim={"_target_": "torch.optim.Adam", "lr": 1e-3, "weight_decay": 1e-3}, choices={},
               data_params=DataParameters((1, size, size), (21, 2), [0, 1]), psm_path="unused.npy", seq_psm_path="unused.npy",
               t_a=25, t_e=1)


def test_state_dict_and_optimizer_layout_equal_the_reference(golden_dir):
    ref = json.loads((golden_dir / "ref_ckpt_structure.json").read_text())
    task = _task("dsnt-skew", 8, 256)
    mine = [[k, list(v.shape), str(v.dtype)] for k, v in task.state_dict().items()]
    assert mine == ref["state_dict"]
    assert [n for n, _ in task.named_parameters()] == ref["parameter_order"]
    # optimizer state layout: param ids = positions in parameters() order; the six deep-supervision heads never get a gradient
    names = ref["parameter_order"]
    no_grad = [i for i, n in enumerate(names) if n.startswith("model.deep_supervision_heads")]
    assert sorted(set(range(len(names))) - set(no_grad)) == ref["optimizer_state_ids"]
    assert ref["optimizer_state_keys"] == ["exp_avg", "exp_avg_sq", "step"]
    g = ref["optimizer_param_groups"][0]
    assert g["lr"] == 1e-3 and g["weight_decay"] == 1e-3 and tuple(g["betas"]) == (0.9, 0.999) and g["params"] == len(names)


def test_reference_structured_checkpoint_loads_strict(golden_dir):
    from contour_uncertainty.utils import checkpoint as C
    ckpt = C.load_lightning_checkpoint(golden_dir / "ref_ckpt_tiny.ckpt")
    assert ckpt["pytorch-lightning_version"].startswith("1.8") and ckpt["epoch"] == 3 and ckpt["global_step"] == 17
    task = _task("dsnt-al", 2, 16)
    res = C.load_weights(task, golden_dir / "ref_ckpt_tiny.ckpt", strict=True)          # runner.py:117-120
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in ckpt["state_dict"].items():
        assert torch.equal(task.state_dict()[k], v)
    # the optimizer state (torch.optim.Adam layout) into the fused Adam and into torch.optim.Adam alike
    from cu_hip.optim import FusedAdam
    for cls in (FusedAdam, torch.optim.Adam):
        opt = cls(task.parameters(), lr=1e-3, weight_decay=1e-3)
        opt.load_state_dict(ckpt["optimizer_states"][0])
        params = list(task.parameters())
        st = ckpt["optimizer_states"][0]["state"]
        assert len(opt.state) == len(st)
        for i, s in st.items():
            mine = opt.state[params[int(i)]]
            assert torch.equal(mine["exp_avg"], s["exp_avg"]) and torch.equal(mine["exp_avg_sq"], s["exp_avg_sq"])
            assert float(mine["step"]) == float(s["step"]) == 1.0


def test_written_checkpoint_has_lightning_layout_and_round_trips(tmp_path):
    from contour_uncertainty.utils import checkpoint as C
    torch.manual_seed(0)
    task = _task("dsnt-skew", 6, 64)
    opt = torch.optim.Adam(task.parameters(), lr=1e-3, weight_decay=1e-3)
    for p in task.parameters():
        if "deep_supervision" not in str(id(p)):
            p.grad = torch.randn_like(p) * 1e-2
    opt.step()
    path = C.save_lightning_checkpoint(task, tmp_path / "a" / "model.ckpt", optimizer=opt, epoch=4, global_step=123)
    raw = torch.load(str(path), map_location="cpu", weights_only=False)
    assert set(C._KEYS) <= set(raw) and raw["hparams_name"] == "kwargs"
    assert all(k.startswith(("model.", "skew_block.model.")) for k in raw["state_dict"])
    assert raw["hyper_parameters"]["covar"] is True and raw["hyper_parameters"]["t_a"] == 25
    assert raw["hyper_parameters"]["data_params"]["out_shape"] == [21, 2]
    # weights_only load works too: nothing but tensors and plain containers inside
    torch.load(str(path), map_location="cpu", weights_only=True)
    other = _task("dsnt-skew", 6, 64)
    opt2 = torch.optim.Adam(other.parameters(), lr=1e-3, weight_decay=1e-3)
    ck = C.restore(other, path, optimizer=opt2, strict=True)
    assert ck["epoch"] == 4 and ck["global_step"] == 123
    for (n, a), (_, b) in zip(task.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), n
    for pa, pb in zip(task.parameters(), other.parameters()):
        assert torch.equal(opt.state[pa]["exp_avg_sq"], opt2.state[pb]["exp_avg_sq"])
    # the task's own save / load_from_checkpoint use the same layout
    task.save_checkpoint(tmp_path / "b.ckpt", optimizer=opt, epoch=1, global_step=2)
    again = type(task).load_from_checkpoint(tmp_path / "b.ckpt")
    assert torch.equal(again.state_dict()["skew_block.model.7.bias"], task.state_dict()["skew_block.model.7.bias"])
    assert set(C._KEYS) <= set(torch.load(str(tmp_path / "b.ckpt"), weights_only=False))


def test_checkpoint_with_unimportable_classes_still_loads(tmp_path):
    """hyper_parameters of a reference-written file are omegaconf objects; their classes do not exist here"""
    from contour_uncertainty.utils import checkpoint as C
    mod = types.ModuleType("fake_omegaconf")
    exec("class DictConfig:\n"
         "    def __init__(self, content):\n"
         "        self.__dict__['_content'] = content\n"
         "    def __getstate__(self):\n"
         "        return {'_content': self._content, '_metadata': 'x'}\n"
         "    def __setstate__(self, st):\n"
         "        self.__dict__.update(st)\n", mod.__dict__)
    DictConfig = mod.DictConfig
    DictConfig.__module__ = "fake_omegaconf"
    sys.modules["fake_omegaconf"] = mod
    try:
        sd = {"model.output_block.conv.weight": torch.arange(6.0).view(1, 6, 1, 1)}
        torch.save({"state_dict": sd, "hyper_parameters": DictConfig({"covar": True, "model": DictConfig({"kernels": [[3, 3]]})}),
                    "epoch": 1, "global_step": 2, "pytorch-lightning_version": "1.8.0"}, str(tmp_path / "r.ckpt"))
    finally:
        del sys.modules["fake_omegaconf"]
    with pytest.raises(Exception):
        torch.load(str(tmp_path / "r.ckpt"), weights_only=False)
    ck = C.load_lightning_checkpoint(tmp_path / "r.ckpt")
    assert torch.equal(ck["state_dict"]["model.output_block.conv.weight"], sd["model.output_block.conv.weight"])
    assert ck["hyper_parameters"] == {"covar": True, "model": {"kernels": [[3, 3]]}}


@pytest.mark.gpu
def test_network_from_reference_checkpoint_reproduces_reference_logits(golden_dir):
    from contour_uncertainty.utils import checkpoint as C
    io = np.load(golden_dir / "ref_ckpt_tiny_io.npz")
    task = _task("dsnt-al", 2, 16)
    C.load_weights(task, golden_dir / "ref_ckpt_tiny.ckpt", strict=True)
    task = task.to("cuda").eval()
    with torch.no_grad():
        logits = task.model(torch.from_numpy(io["x"]).cuda())
    assert torch.allclose(logits.cpu(), torch.from_numpy(io["logits"]), rtol=1e-4, atol=2e-5)

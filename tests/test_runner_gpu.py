"""BASELINE config c1 end to end through the YAML runner: `task=dsnt-* data=synthetic` at 64 x 64 with
the 6-stage unet2, batch 2 -- composed from the reference-schema config tree, instantiated by `_target_`, two optimiser
steps, a validation pass with Dice, a checkpoint round trip and the predict loop with sampling + masks."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "contouring-uncertainty_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

SIX = ["task.model.kernels=[[3,3],[3,3],[3,3],[3,3],[3,3],[3,3]]",
       "task.model.strides=[[1,1],[2,2],[2,2],[2,2],[2,2],[2,2]]"]


def _argv(task, golden_dir, tmp, extra=()):
    return [f"task={task}", "data=synthetic", "data.size=64", "data.batch_size=2", "data.n_train=8", "data.n_val=4",
            "data.n_predict=3", "trainer.fast_dev_run=False", "trainer.max_epochs=1", "trainer.max_steps=2",
            f"model_path={tmp}", "task.t_a=5", "task.model.compute_dtype=f32",
            f"task.psm_path={golden_dir / 'camus-cont_psm_11_no_std.npz'}",
            f"task.seq_psm_path={golden_dir / 'camus-cont_sequence_psm_11_no_std.npz'}"] + SIX + list(extra)


@pytest.mark.parametrize("task", ["dsnt-skew", "dsnt-al", "dsnt-al2"])
def test_config_c1_fit_and_predict_through_the_runner(golden_dir, task, tmp_path):
    import runner
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    model, trainer, preds = runner.main(_argv(task, golden_dir, tmp_path))
    m = trainer.callback_metrics
    assert trainer.global_step == 2
    for key in ("train/loss", "train/distance_loss", "train/loss_term1", "train/loss_term2", "val/loss", "val/dice"):
        assert key in m and np.isfinite(float(m[key])), key
    if task == "dsnt-skew":
        assert "train/loss_term3" in m and "train/alpha_norm" in m
    assert 0.0 <= float(m["val/dice"]) <= 1.0
    # predict: 3 synthetic views of 2 instants each
    assert len(preds) == 3 and [p.view_index for p in preds] == [0, 1, 2]
    t_a = 25 if task == "dsnt-skew" else 5                  # hard-coded in the reference's skew predict step
    for p in preds:
        assert p.mu.shape == (2, 21, 2) and p.cov.shape == (2, 21, 2, 2)
        assert p.contour_samples.shape == (2, 1, t_a, 21, 2) and np.isfinite(p.contour_samples).all()
        assert p.pred_samples.shape == (2, 1, t_a, 64, 64) and p.entropy_map.shape == (2, 64, 64)
        assert p.post_cov.shape == (2, 21, 2, 2) and p.id.startswith("synthetic")
    # the runner saved the model where best_model_save_path says (Lightning's checkpoint layout); `ckpt=` of the runner
    # (vital/vital/runner.py:114-116) loads it back -> same predictions
    saved = list(Path(tmp_path).glob("*.ckpt"))
    assert len(saved) == 1 and saved[0].name.startswith("synthetic-lv_")
    ckpt = saved[0]
    _, _, again = runner.main(_argv(task, golden_dir, tmp_path, [f"ckpt={ckpt}", "train=False"]))
    for a, b in zip(preds, again):
        assert np.allclose(a.mu, b.mu, rtol=1e-4, atol=1e-3)
        # same per-view seed -> same draws; the skew sampler picks grid cells, so the ~1e-6 run-to-run noise of mu (f32
        # atomics in the InstanceNorm statistics) may move an isolated draw by a cell
        d = np.abs(a.contour_samples - b.contour_samples)
        assert (d > 2e-2).mean() < (0.02 if task == "dsnt-skew" else 1e-9)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, argv, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), CONTOUR_DIST_BACKEND="gloo")
    import runner
    _, _, preds = runner.main(argv)
    ret[rank] = [(p.view_index, p.mu, p.contour_samples, p.entropy_map) for p in preds]
    torch.distributed.destroy_process_group()


def test_frame_sharded_predict_equals_single_rank(golden_dir, tmp_path):
    """BASELINE config c5 in its sharded form: views are dealt to 2 ranks (gloo carries the gather on the
    one-GPU test box; RCCL with backend "nccl" on a node), each view keeps its ED/ES pair on one rank, and the gathered
    result equals the single-rank run view by view -- contour samples included (per-view seeding)."""
    import runner
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    model, _, _ = runner.main(_argv("dsnt-al", golden_dir, tmp_path, ["predict=False", "data.n_predict=4"]))
    ckpt = tmp_path / "w.ckpt"
    model.save_checkpoint(ckpt)
    argv = _argv("dsnt-al", golden_dir, tmp_path, [f"ckpt={ckpt}", "train=False", "data.n_predict=4", "task.t_a=16"])
    _, _, single = runner.main(argv)
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), argv, ret), nprocs=2, join=True)
    for rank in (0, 1):
        got = ret[rank]
        assert [g[0] for g in got] == [0, 1, 2, 3]
        for g, s in zip(got, single):
            assert np.allclose(g[1], s.mu, rtol=1e-4, atol=1e-3)
            assert np.allclose(g[2], s.contour_samples, atol=2e-2)
            # the two runs' contours agree to 2e-2 px (InstanceNorm sums are f32 atomics: last-bit noise), so a sampled contour
            # may round a boundary pixel of its mask differently: with 16 samples one flipped mask moves that pixel's entropy
            # by up to H(1/16) = 0.34 and a few flips at one pixel by more -- bound HOW MANY pixels move, not the largest move
            diff = np.abs(g[3] - s.entropy_map)
            assert (diff > 1e-3).mean() < 0.02 and diff.mean() < 5e-3

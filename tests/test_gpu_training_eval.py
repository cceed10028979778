"""-m gpu: the callers either side of the path: evaluate_model semantics on
device, the training loop with checkpoints in the reference's format, optimizer-state resume."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import metrics_ref
from rfi_toolbox_amd.models import UNet
from rfi_toolbox_amd.training import (evaluate_rfi_model, load_checkpoint, save_checkpoint,
                                      train_rfi_model)

pytestmark = pytest.mark.gpu


def _golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_f8_b4_s64.npz"))
    st = OrderedDict((k[8:], torch.from_numpy(g[k].copy())) for k in g.files if k.startswith("state40/"))
    return g, st


def test_evaluate_model_semantics(golden_dir):
    """mean of PER-BATCH metrics (evaluate_model.py:54-56) on the reference-trained weights."""
    g, st = _golden(golden_dir)
    m = UNet(3, 1, 8).load_state_dict(st)
    img, lab = torch.from_numpy(g["img"]), torch.from_numpy(g["lab"])
    got = evaluate_rfi_model(m, (img, lab), batch_size=2)
    pred = g["logits_eval40"][:, 0] > 0                       # sigmoid > 0.5
    want = [metrics_ref.evaluate_segmentation(pred[i:i + 2], g["lab"][i:i + 2]) for i in (0, 2)]
    for k in ("iou", "precision", "recall", "f1", "dice"):
        assert got[k] == pytest.approx(np.mean([w[k] for w in want]), abs=1e-3), k
    assert m.training                                         # mode restored
    tp, fp, fn = m.eval().eval_batch(img, lab)
    assert abs(tp / (tp + fp + fn) - float(g["iou"][-1])) <= 1e-3
    with pytest.raises(ValueError):
        evaluate_rfi_model(m, (img[:0], lab[:0]))


def test_checkpoint_round_trip_and_resume(tmp_path):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 32, 32, 3, generator=g)
    y = (torch.rand(4, 32, 32, generator=g) > 0.7).to(torch.uint8)
    torch.manual_seed(21)
    a = UNet(3, 1, 8)
    for _ in range(2):
        a.train_step(x, y, lr=1e-3)
    path = save_checkpoint(str(tmp_path / "ck" / "unet_rfi_epoch_2.pt"), a, epoch=2, loss=0.5, args={"lr": 1e-3},
                           optimizer_hyper=dict(lr=1e-3))
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss", "args"}    # train_model.py:177-183
    # the optimizer state is a genuine torch.optim.Adam state_dict
    params = [torch.nn.Parameter(p.clone()) for p in a.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5)
    opt.load_state_dict(ck["optimizer_state_dict"])
    assert float(opt.state[params[0]]["step"]) == 2
    b = UNet(3, 1, 8)
    assert load_checkpoint(path, b)["epoch"] == 2
    la, lb = a.train_step(x, y, lr=1e-3), b.train_step(x, y, lr=1e-3)
    assert la == lb
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    c = UNet(3, 1, 8).load_state_dict(ck)                     # wrapped dict accepted directly
    assert torch.equal(c.state_dict()["final_conv.weight"], ck["model_state_dict"]["final_conv.weight"])


def test_training_loop_reduces_loss_and_writes_reference_checkpoints(tmp_path):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(8, 32, 32, 3, generator=g)
    y = torch.zeros(8, 32, 32, dtype=torch.uint8)
    y[:, 10:14, :] = 1
    x[:, 10:14, :, 1] += 3.0
    torch.manual_seed(0)
    m = UNet(3, 1, 8)
    logs = []
    hist = train_rfi_model(m, (x[:6], y[:6]), (x[6:], y[6:]), num_epochs=6, batch_size=2, lr=1e-3,
                           checkpoint_dir=str(tmp_path / "ckpt"), log=logs.append)
    assert len(hist) == 6 and hist[-1]["train_loss"] < hist[0]["train_loss"]
    assert all(np.isfinite(h["val_loss"]) for h in hist) and len(logs) == 6
    files = sorted(os.listdir(tmp_path / "ckpt"))
    assert "unet_rfi_final.pt" in files and any(f.startswith("unet_rfi_epoch_") for f in files)
    final = torch.load(tmp_path / "ckpt" / "unet_rfi_final.pt", weights_only=False)
    assert set(final) == {"model_state_dict", "args"}                                          # train_model.py:190-193
    assert list(final["model_state_dict"])[0] == "encoder1.conv.conv.0.weight"
    # resume continues from the stored epoch
    best = sorted(f for f in files if f.startswith("unet_rfi_epoch_"))[-1]
    hist2 = train_rfi_model(UNet(3, 1, 8), (x[:6], y[:6]), (x[6:], y[6:]), num_epochs=7, batch_size=2, lr=1e-3,
                            resume_from=str(tmp_path / "ckpt" / best), log=lambda s: None)
    assert hist2[0]["epoch"] == int(best.split("_")[-1].split(".")[0]) + 1


def test_reference_shards_straight_to_hbm_and_one_step(golden_dir):
    """the reference-written shard fixture goes from the memory-mapped .pt files straight into one
    NHWC device buffer pair (`load_batches_device`), bit-identical to the host loader, and a training step run
    from those device buffers equals the step run from the host tensors and the CPU oracle."""
    from collections import OrderedDict as OD

    from oracle import unet_ref
    from rfi_toolbox_amd._lib import Hyper
    from rfi_toolbox_amd.datasets import load_batches, load_batches_device
    d = os.path.join(golden_dir, "ref_shards")
    host = load_batches(d)
    d_img, d_lab, meta = load_batches_device(d, device=0)
    assert d_img.shape == (16, 32, 32, 3) and d_lab.shape == (16, 32, 32) and meta["num_batches"] == 4
    np.testing.assert_array_equal(d_img.numpy(), host.images.numpy())
    np.testing.assert_array_equal(d_lab.numpy(), host.labels.numpy())
    torch.manual_seed(13)
    a = UNet(3, 1, 8).train()
    st0 = a.state_dict()
    b = UNet(3, 1, 8).load_state_dict(st0).train()
    la = a.train_step(host.images, host.labels, lr=1e-3)                       # host tensors in
    b.train_step_async(d_img.ptr, d_lab.ptr, 16, 32, 32, Hyper(1e-3, 0.9, 0.999, 1e-8, 1e-5, 1.0))   # device buffers in
    lb, _ = b.last_loss()
    assert la == lb
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    ost = OD((k, v.clone()) for k, v in st0.items())
    r = unet_ref.train_step(ost, unet_ref.new_adam_state(ost), unet_ref.nhwc_to_nchw(host.images),
                            host.labels.float().unsqueeze(1), lr=1e-3)
    assert la == pytest.approx(r["loss"], abs=5e-6)

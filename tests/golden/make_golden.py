#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (the reference never travels to the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 CI=1 \
        PYTHONPATH=/root/reference python tests/golden/make_golden.py

Every array written here is an input or an output of reference code
(preshanth/rfi_toolbox v0.2.0); nothing of the reference's source is stored.
``scripts/train_model.py`` itself is not importable here (it needs
``albumentations``), so the optimisation step below is driven with the very
torch objects that script constructs (train_model.py:120-151):
``nn.BCEWithLogitsLoss`` + its ``dice_loss`` formula, ``optim.Adam(lr, weight_decay)``,
``clip_grad_norm_(.., 1.0)``; autocast/GradScaler are disabled on CPU (:131,:144).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

os.environ.setdefault("CI", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))

from rfi_toolbox.config.loader import DataConfig                      # noqa: E402
from rfi_toolbox.data_generation.synthetic_generator import SyntheticDataGenerator  # noqa: E402
from rfi_toolbox.evaluation.metrics import evaluate_segmentation      # noqa: E402
from rfi_toolbox.models import UNet                                   # noqa: E402
from rfi_toolbox.models.unet import UNetBigger, UNetDifferentActivation, UNetOverfit   # noqa: E402
from rfi_toolbox.preprocessing.preprocessor import Preprocessor, patchify  # noqa: E402

torch.set_num_threads(4)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB")


# ---------------------------------------------------------------- G1: generator
def gen_sample(seed, nc, nt, npol, counts, bandpass=True):
    synth = {"rfi_type_counts": counts, "rfi_types": list(counts.keys())}
    cfg = DataConfig({"synthetic": synth, "processing": {}})
    g = SyntheticDataGenerator(cfg)
    rfi_cfg = g._parse_rfi_config(cfg.synthetic)
    np.random.seed(seed)
    w, m, _ = g._generate_single_sample(
        num_channels=nc, num_times=nt, noise_level=1.0, rfi_power_min=1000.0,
        rfi_power_max=10000.0, rfi_config=rfi_cfg, enable_bandpass=bandpass,
        bandpass_order=8, num_polarizations=npol, pol_corr=0.8, synth_config=cfg.synthetic)
    return w, m


COUNTS_A = {"narrowband_persistent": 2, "broadband_persistent": 1, "frequency_sweep": 1}
COUNTS_B = {"narrowband_persistent": 2, "broadband_persistent": 1, "frequency_sweep": 1,
            "narrowband_bursty": 2, "broadband_bursty": 1, "narrowband_intermittent": 1}


def golden_preprocess():
    out = {}
    # (a) 64x64, 1 pol, whole-waterfall patches (patch_size == image), 4 views
    w, m = gen_sample(1234, 64, 64, 1, COUNTS_A)
    np.random.seed(7)
    ds = Preprocessor(w, flags=m).create_dataset(patch_size=64, num_workers=0)
    out.update(a_w=w, a_m=m, a_img=ds.images.numpy(), a_lab=ds.labels.numpy())
    # (b) 96x80, 2 pols, tiling with zero padding, patch 32, 4 views, blank removal + shuffle
    w, m = gen_sample(1235, 96, 80, 2, COUNTS_B)
    np.random.seed(8)
    ds = Preprocessor(w, flags=m).create_dataset(patch_size=32, num_workers=0)
    out.update(b_w=w, b_m=m, b_img=ds.images.numpy(), b_lab=ds.labels.numpy())
    # (b2) same input, no augmentation, inference mode (order preserved, zero labels)
    ds = Preprocessor(w, flags=m).create_dataset(patch_size=32, num_workers=0,
                                                 enable_augmentation=False, inference_mode=True)
    out.update(b2_img=ds.images.numpy(), b2_lab=ds.labels.numpy())
    # (b3) 2 rotations, num_patches limit
    np.random.seed(9)
    ds = Preprocessor(w, flags=m).create_dataset(patch_size=32, num_workers=0,
                                                 augmentation_rotations=2, num_patches=10)
    out.update(b3_img=ds.images.numpy(), b3_lab=ds.labels.numpy())
    # (c) real-valued input: median normalise + stretch + MAD flags (no custom flags)
    wr = np.abs(w[0])                                   # (2, 96, 80) float64
    for tag, kw in (("c_sqrt", dict(stretch="SQRT")),
                    ("c_log", dict(stretch="LOG10", normalize_after_stretch=True)),
                    ("c_none", dict(stretch=None, normalize_before_stretch=False))):
        np.random.seed(10)
        ds = Preprocessor(wr, flags=None).create_dataset(patch_size=32, num_workers=0,
                                                         flag_sigma=5, **kw)
        out[f"{tag}_img"] = ds.images.numpy()
        out[f"{tag}_lab"] = ds.labels.numpy()
    out["c_w"] = wr
    # (d) hand-made patches straight through the channel extractors
    p = Preprocessor(np.zeros((1, 1, 4, 4), dtype=np.complex128))
    rng = np.random.default_rng(5)
    z = (rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8))) * 10.0 ** rng.uniform(-4, 5, (8, 8))
    z[2, 3] = 0.0                                       # |z| = 0 -> log floor
    zc = np.full((8, 8), 3.0 + 4.0j)                    # constant patch: gradient span 0
    zz = np.zeros((8, 8), dtype=np.complex128)
    out.update(d_z=z, d_zc=zc, d_zz=zz,
               d_z_ch=p._extract_channels_from_complex(z),
               d_zc_ch=p._extract_channels_from_complex(zc),
               d_zz_ch=p._extract_channels_from_complex(zz),
               d_r_ch=p._extract_channels_from_real(np.abs(z)),
               d_rc_ch=p._extract_channels_from_real(np.abs(zc)))
    # (e) the reference's own known-answer test (tests/test_preprocessing.py:22-33)
    pa = patchify(np.arange(16).reshape(4, 4), (2, 2), 2)
    out.update(e_first=pa[0, 0], e_last=pa[1, 1], e_shape=np.array(pa.shape))
    save("preprocess.npz", **out)


# ---------------------------------------------------------------- G4/G8: model + step
def dice_loss(pred, target, smooth=1.0):
    pred = torch.sigmoid(pred)
    iflat = pred.contiguous().view(-1)
    tflat = target.contiguous().view(-1)
    inter = (iflat * tflat).sum()
    return 1 - ((2.0 * inter + smooth) / (iflat.sum() + tflat.sum() + smooth))


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def make_batch(seed, size, n_views=4):
    """Reference generator -> reference Preprocessor -> (x NCHW f32, y (N,1,H,W) f32, nhwc, u8)."""
    w, m = gen_sample(seed, size, size, 1, COUNTS_A)
    np.random.seed(seed + 1)
    ds = Preprocessor(w, flags=m).create_dataset(patch_size=size, num_workers=0)
    img = ds.images.numpy()[:n_views]
    lab = ds.labels.numpy()[:n_views]
    return img, lab


def run_steps(model, img, lab, steps, lr, wd, record):
    x = torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()
    y = torch.from_numpy(lab).float().unsqueeze(1)
    crit = nn.BCEWithLogitsLoss()
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    losses, norms = [], []
    # distance of the nearest training-mode BatchNorm output to the ReLU threshold, per step: a
    # fixture with an element within float32 rounding of 0 makes every backward comparison a coin flip
    margins, cur = [], []
    hooks = [mod.register_forward_hook(lambda _m, _i, o: cur.append(float(o.detach().abs().min())) if _m.training else None)
             for mod in model.modules() if isinstance(mod, nn.BatchNorm2d)]
    if not hooks:                         # BN-less model: the ReLU inputs themselves
        hooks = [mod.register_forward_pre_hook(lambda _m, i: cur.append(float(i[0].detach().abs().min())) if _m.training else None)
                 for mod in model.modules() if isinstance(mod, nn.ReLU)]
    for s in range(1, steps + 1):
        model.train()
        opt.zero_grad()
        cur.clear()
        out = model(x)
        margins.append(min(cur))
        loss = crit(out, y) + dice_loss(out, y)
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        if s == 1:
            record["logits_train1"] = out.detach().numpy().copy()
        if s in record.get("_grad_steps", ()):
            # grads AFTER clipping (what Adam sees) and the pre-clip norm
            for k, p in model.named_parameters():
                record[f"grad{s}/{k}"] = p.grad.detach().numpy().copy()
        opt.step()
        losses.append(float(loss))
        norms.append(float(norm))
        if s in record.get("_state_steps", ()):
            for k, v in sd_np(model).items():
                record[f"state{s}/{k}"] = v
            if s in record.get("_adam_steps", ()):
                for k, p in model.named_parameters():
                    record[f"adam_m{s}/{k}"] = opt.state[p]["exp_avg"].numpy().copy()
                    record[f"adam_v{s}/{k}"] = opt.state[p]["exp_avg_sq"].numpy().copy()
            model.eval()
            with torch.no_grad():
                record[f"logits_eval{s}"] = model(x).numpy().copy()
    for h in hooks:
        h.remove()
    record["losses"] = np.array(losses, dtype=np.float64)
    record["grad_norms"] = np.array(norms, dtype=np.float64)
    record["relu_margin"] = np.array(margins, dtype=np.float64)
    return x, y


def golden_unet_small():
    """f=4, batch 4, 32x32: complete state before/after steps 1 and 3 (+Adam moments)."""
    img, lab = make_batch(1234, 32)
    for seed in range(1234, 1300):        # first init seed whose 3 steps stay clear of a ReLU threshold
        torch.manual_seed(seed)
        model = UNet(in_channels=3, out_channels=1, init_features=4)
        rec = {"_grad_steps": (1,), "_state_steps": (1, 3), "_adam_steps": (3,)}
        for k, v in sd_np(model).items():
            rec[f"state0/{k}"] = v
        model.eval()
        with torch.no_grad():
            rec["logits_eval0"] = model(torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()).numpy()
        run_steps(model, img, lab, 3, lr=1e-3, wd=1e-5, record=rec)
        print(f"unet_f4 init seed {seed}: relu margins {rec['relu_margin']}")
        if rec["relu_margin"].min() > 1e-5:
            break
    rec = {k: v for k, v in rec.items() if not k.startswith("_")}
    rec.update(img=img, lab=lab, names=np.array(list(model.state_dict().keys())),
               hyper=np.array([1e-3, 0.9, 0.999, 1e-8, 1e-5, 1.0]))
    save("unet_f4_b4_s32.npz", **rec)


def golden_unet_f8():
    """f=8, batch 4, 64x64: init state, step-1 logits/loss/norm/selected grads, and the
    40-step trajectory (loss each step, IoU every 10) of SURVEY G8 (lr 1e-3)."""
    img, lab = make_batch(4321, 64)
    torch.manual_seed(1234)
    model = UNet(in_channels=3, out_channels=1, init_features=8)
    rec = {"_grad_steps": (1,), "_state_steps": (), "_adam_steps": ()}
    for k, v in sd_np(model).items():
        rec[f"state0/{k}"] = v
    x = torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()
    y = torch.from_numpy(lab).float().unsqueeze(1)
    crit = nn.BCEWithLogitsLoss()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    losses, ious, norms = [], {}, []
    keep = ("encoder1.conv.conv.0.weight", "encoder1.conv.conv.1.weight", "encoder2.conv.conv.3.weight",
            "bottleneck.conv.0.weight", "bottleneck.conv.4.bias", "decoder4.up.weight",
            "decoder4.up.bias", "decoder1.conv.conv.0.weight", "final_conv.weight", "final_conv.bias")
    for s in range(1, 41):
        model.train()
        opt.zero_grad()
        out = model(x)
        loss = crit(out, y) + dice_loss(out, y)
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        if s == 1:
            rec["logits_train1"] = out.detach().numpy().copy()
            for k, p in model.named_parameters():
                if k in keep:
                    rec[f"grad1/{k}"] = p.grad.detach().numpy().copy()
        opt.step()
        losses.append(float(loss))
        norms.append(float(norm))
        if s % 10 == 0:
            model.eval()
            with torch.no_grad():
                lg = model(x)
                pred = (torch.sigmoid(lg) > 0.5).float()
            ious[s] = evaluate_segmentation(pred, y)
            if s == 40:
                rec["logits_eval40"] = lg.numpy().copy()
                for k, v in sd_np(model).items():      # reference-TRAINED weights (inference parity)
                    rec[f"state40/{k}"] = v
    rec = {k: v for k, v in rec.items() if not k.startswith("_")}
    rec.update(img=img, lab=lab, losses=np.array(losses), grad_norms=np.array(norms),
               iou_steps=np.array(sorted(ious)), iou=np.array([ious[s]["iou"] for s in sorted(ious)]),
               f1=np.array([ious[s]["f1"] for s in sorted(ious)]))
    save("unet_f8_b4_s64.npz", **rec)
    print("f8 trajectory IoU:", [round(ious[s]["iou"], 4) for s in sorted(ious)])


def golden_unet_bigger():
    """UNetBigger (5 levels) f=4, batch 2, 32x32: logits train/eval + loss (variant coverage)."""
    img, lab = make_batch(99, 32, n_views=2)
    torch.manual_seed(5)
    model = UNetBigger(in_channels=3, out_channels=1, init_features=4)
    rec = {}
    for k, v in sd_np(model).items():
        rec[f"state0/{k}"] = v
    rec2 = {"_grad_steps": (), "_state_steps": (1,), "_adam_steps": ()}
    run_steps(model, img, lab, 1, lr=1e-3, wd=1e-5, record=rec2)
    rec.update({k: v for k, v in rec2.items() if not k.startswith("_") and not k.startswith("state1/")})
    for k in ("encoder5.conv.conv.1.running_mean", "encoder5.conv.conv.1.num_batches_tracked",
              "bottleneck.conv.1.running_var", "bottleneck.conv.1.num_batches_tracked",
              "decoder5.up.weight", "final_conv.bias"):
        rec[f"state1/{k}"] = rec2[f"state1/{k}"]
    rec.update(img=img, lab=lab, names=np.array(list(model.state_dict().keys())))
    save("unetbigger_f4_b2_s32.npz", **rec)


# ---------------------------------------------------------------- N4: UNetOverfit / UNetDifferentActivation
def golden_unet_variants():
    """Reference variants (models/unet.py:120-268) through the reference's step: eval/train outputs,
    loss, step-1 gradients of a few tensors and the state after one step."""
    # UNetOverfit has five levels: 64 x 64 keeps 2 x 2 pixels (8 BatchNorm samples) at the bottleneck
    for tag, size, ctor, seed0 in (("overfit_f4_b2_s64", 64, lambda: UNetOverfit(in_channels=3, out_channels=1, init_features=4), 40),
                                   ("leaky_f4_b2_s32", 32, lambda: UNetDifferentActivation(in_channels=3, out_channels=1,
                                                                                        init_features=4,
                                                                                        activation=nn.LeakyReLU), 60)):
        img, lab = make_batch(777, size, n_views=2)
        for seed in range(seed0, seed0 + 40):
            torch.manual_seed(seed)
            model = ctor()
            rec = {"_grad_steps": (1,), "_state_steps": (1,), "_adam_steps": ()}
            for k, v in sd_np(model).items():
                rec[f"state0/{k}"] = v
            model.eval()
            with torch.no_grad():
                rec["logits_eval0"] = model(torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()).numpy()
            run_steps(model, img, lab, 1, lr=1e-3, wd=1e-5, record=rec)
            print(f"{tag} init seed {seed}: activation margins {rec['relu_margin']}")
            if rec["relu_margin"].min() > 1e-5:
                break
        rec = {k: v for k, v in rec.items() if not k.startswith("_")}
        rec.update(img=img, lab=lab, names=np.array(list(model.state_dict().keys())),
                   hyper=np.array([1e-3, 0.9, 0.999, 1e-8, 1e-5, 1.0]))
        save(f"unet_{tag}.npz", **rec)


# ---------------------------------------------------------------- A9: the "3-layer CNN segmenter"
class SimpleCNN(nn.Module):
    """NOT a reference class: the reference README's elided custom-model example
    (README.md:379-398) completed by the build as Conv3x3+ReLU, Conv3x3+ReLU, Conv1x1 -> logits.
    The data (generator + Preprocessor) and the optimisation step are the reference's."""

    def __init__(self, in_channels=3, out_channels=1, width=64):
        super().__init__()
        self.encoder = nn.Sequential(nn.Conv2d(in_channels, width, 3, padding=1), nn.ReLU(),
                                     nn.Conv2d(width, width, 3, padding=1), nn.ReLU())
        self.decoder = nn.Sequential(nn.Conv2d(width, out_channels, 1))

    def forward(self, x):
        return self.decoder(self.encoder(x))


def golden_cnn3():
    img, lab = make_batch(4321, 32)
    for seed in range(77, 140):
        torch.manual_seed(seed)
        model = SimpleCNN(3, 1, 16)
        rec = {"_grad_steps": (1,), "_state_steps": (1, 3), "_adam_steps": (3,)}
        for k, v in sd_np(model).items():
            rec[f"state0/{k}"] = v
        model.eval()
        with torch.no_grad():
            rec["logits_eval0"] = model(torch.from_numpy(img).permute(0, 3, 1, 2).contiguous()).numpy()
        run_steps(model, img, lab, 3, lr=1e-3, wd=1e-5, record=rec)
        print(f"cnn3 init seed {seed}: relu margins {rec['relu_margin']}")
        if rec["relu_margin"].min() > 5e-6:
            break
    rec = {k: v for k, v in rec.items() if not k.startswith("_")}
    rec.update(img=img, lab=lab, names=np.array(list(model.state_dict().keys())),
               hyper=np.array([1e-3, 0.9, 0.999, 1e-8, 1e-5, 1.0]))
    save("cnn3_c16_b4_s32.npz", **rec)


# ---------------------------------------------------------------- G6: metrics
def golden_metrics():
    rng = np.random.default_rng(3)
    cases = {}
    z, o = np.zeros((4, 4)), np.ones((4, 4))
    a = (rng.random((2, 1, 16, 16)) > 0.6)
    b = (rng.random((2, 1, 16, 16)) > 0.5)
    inputs = {
        "both_empty": (z, z), "pred_empty_true_full": (z, o), "pred_full_true_empty": (o, z),
        "both_full": (o, o), "random_bool": (a, b),
        "random_float_nonzero": (a.astype(np.float32) * 0.25, b.astype(np.float32) * 7.0),
        "torch_n1hw": (torch.from_numpy(a.astype(np.float32)), torch.from_numpy(b.astype(np.uint8))),
    }
    arrays = {}
    for name, (p, t) in inputs.items():
        r = evaluate_segmentation(p, t)
        cases[name] = {k: float(v) for k, v in r.items()}
        arrays[f"{name}/pred"] = np.asarray(p)
        arrays[f"{name}/true"] = np.asarray(t)
    save("metrics_inputs.npz", **arrays)
    with open(os.path.join(HERE, "metrics_expected.json"), "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    print("wrote metrics_expected.json")


if __name__ == "__main__":
    golden_metrics()
    golden_preprocess()
    golden_unet_small()
    golden_unet_f8()
    golden_unet_variants()
    golden_cnn3()
    golden_unet_bigger()

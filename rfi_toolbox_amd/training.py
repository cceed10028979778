"""The callers either side of the hot path, with the reference scripts' semantics.

* ``train_rfi_model``  -- the loop of scripts/train_model.py:133-193: per epoch shuffle, one
  ``train_step`` per batch (BCE-with-logits + dice, clip 1.0, Adam with coupled L2), validation
  loss in eval mode averaged over batches (:157-167), stop on NaN (:170-172), checkpoint dict
  ``{epoch, model_state_dict, optimizer_state_dict, loss, args}`` whenever the validation loss
  improves (:174-184) and a final ``{model_state_dict, args}`` (:189-193).  Unlike the reference
  the checkpoint directory is created and ``resume_from`` actually resumes.
* ``evaluate_rfi_model`` -- scripts/evaluate_model.py:18-58: eval mode, per batch
  ``sigmoid > 0.5`` -> ``evaluate_segmentation``, then the MEAN OF THE PER-BATCH metrics
  (:54-56; not the metric of the pooled counts).
Data are TorchDataset-like (``.images`` NHWC float32, ``.labels`` uint8) or (images, labels) pairs.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from .evaluation.metrics import _dice, _f1, _iou, _precision, _recall


def _pair(ds):
    if hasattr(ds, "images"):
        return ds.images, ds.labels
    return ds


def _batches(n, batch_size, order=None):
    idx = np.arange(n) if order is None else order
    for i in range(0, n, batch_size):
        yield idx[i:i + batch_size]


def evaluate_rfi_model(model, dataset, batch_size=4, threshold=0.5):
    """dict of mean per-batch iou / precision / recall / f1 / dice (evaluate_model.py:54-56)."""
    images, labels = _pair(dataset)
    was_training = model.training
    model.eval()
    per_batch = []
    try:
        for sel in _batches(len(images), batch_size):
            c = model.eval_batch(images[sel], labels[sel], threshold)
            per_batch.append({"iou": _iou(*c), "precision": _precision(*c), "recall": _recall(*c),
                              "f1": _f1(*c), "dice": _dice(*c)})
    finally:
        model.train(was_training)
    if not per_batch:
        raise ValueError("empty dataset")
    return {k: float(np.mean([m[k] for m in per_batch])) for k in per_batch[0]}


def save_checkpoint(path, model, epoch=None, loss=None, args=None, optimizer_hyper=None):
    """The dict train_model.py:177-183 writes (or :190-193 when epoch is None)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ck = {"model_state_dict": model.state_dict(), "args": args}
    if epoch is not None:
        ck.update(epoch=epoch, loss=loss, optimizer_state_dict=model.optimizer_state_dict(**(optimizer_hyper or {})))
    torch.save(ck, path)
    return path


def load_checkpoint(path, model, load_optimizer=True):
    ck = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["model_state_dict"] if "model_state_dict" in ck else ck)
    if load_optimizer and isinstance(ck, dict) and "optimizer_state_dict" in ck:
        model.load_optimizer_state_dict(ck["optimizer_state_dict"])
    return ck


def train_rfi_model(model, train_data, val_data=None, num_epochs=50, batch_size=4, lr=1e-4,
                    weight_decay=1e-5, checkpoint_dir=None, resume_from=None, args=None, log=print):
    """Returns the history [{epoch, train_loss, val_loss}].  Defaults = train_model.py:86-95."""
    tr_x, tr_y = _pair(train_data)
    start_epoch = 0
    if resume_from:
        start_epoch = int(load_checkpoint(resume_from, model).get("epoch", 0))
    hyper = dict(lr=lr, weight_decay=weight_decay)
    best, history = float("inf"), []
    for epoch in range(start_epoch, num_epochs):
        model.train()
        order = torch.randperm(len(tr_x)).numpy()            # DataLoader(shuffle=True), train_model.py:106
        losses = [model.train_step(tr_x[sel], tr_y[sel], **hyper) for sel in _batches(len(tr_x), batch_size, order)]
        rec = {"epoch": epoch + 1, "train_loss": float(np.mean(losses)), "val_loss": None}
        if val_data is not None:
            va_x, va_y = _pair(val_data)
            model.eval()
            vl = [model.loss(va_x[sel], va_y[sel]) for sel in _batches(len(va_x), batch_size)]
            rec["val_loss"] = float(np.mean(vl))
            log(f"Epoch [{epoch + 1}/{num_epochs}] - Train Loss: {rec['train_loss']:.4f} - Val Loss: {rec['val_loss']:.4f}")
            if math.isnan(rec["val_loss"]):
                log("Validation loss is NaN, stopping training.")
                history.append(rec)
                break
            if rec["val_loss"] < best and checkpoint_dir:
                best = rec["val_loss"]
                save_checkpoint(os.path.join(checkpoint_dir, f"unet_rfi_epoch_{epoch + 1}.pt"), model, epoch + 1,
                                losses[-1], args, hyper)
        history.append(rec)
    if checkpoint_dir:
        save_checkpoint(os.path.join(checkpoint_dir, "unet_rfi_final.pt"), model, args=args)
    return history

"""-m gpu: the ResNet-50-FPN backbone of the Mask R-CNN path (BASELINE.json configs[3]) on MI355X
against oracle/backbone_ref.py (plain torch.nn modules; parity unpinned by the reference, which has no detector): the five
pyramid levels of the forward pass and every parameter gradient of the backward pass from given d(loss)/d(P_i)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import backbone_ref as bref
from oracle import unet_ref
from rfi_toolbox_amd.models import ResNet50FPN

pytestmark = pytest.mark.gpu


def _state(w, f, seed):
    """Default conv init with non-trivial frozen BatchNorm buffers (every term of the affine matters)."""
    st = bref.init_state(3, w, f, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in st:
        if k.endswith("running_mean"):
            st[k] = 0.1 * torch.randn(st[k].shape, generator=g)
        elif k.endswith("running_var"):
            st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
        elif ".bn" in k or "downsample.1" in k:
            st[k] = (1 + 0.2 * torch.randn(st[k].shape, generator=g)) if k.endswith("weight") else 0.1 * torch.randn(st[k].shape, generator=g)
    return st


def test_state_dict_and_default_init():
    torch.manual_seed(3)
    m = ResNet50FPN(3, 8, 16)
    want = bref.init_state(3, 8, 16, seed=3)
    sd = m.state_dict()
    assert list(sd.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(sd[k], want[k]), k
    st = _state(8, 16, 4)
    m.load_state_dict(st)
    back = m.state_dict()
    for k, v in st.items():
        assert torch.equal(back[k], v), k
    n_param = sum(v.numel() for k, v in st.items() if "bn" not in k and "downsample.1" not in k)
    assert m.num_parameters() == n_param
    with pytest.raises(RuntimeError):
        m.forward_features(np.zeros((1, 96, 64, 3), np.float32))         # H, W multiples of 64
    with pytest.raises(ValueError):
        ResNet50FPN(3, 6, 16)


@pytest.mark.parametrize("mode,w,f,n,s", [("float32", 8, 16, 2, 64), ("float32_mfma", 8, 16, 2, 64), ("float32", 16, 32, 2, 128),
                                           ("float32", 64, 256, 1, 128)])
def test_features_and_gradients_vs_oracle(mode, w, f, n, s):
    st = _state(w, f, 11)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(n, s, s, 3, generator=g)
    mod = bref.ResNet50FPN(3, w, f)
    mod.load_state_dict(st)
    xo = unet_ref.nhwc_to_nchw(x)
    feats = mod(xo)
    dfe = [torch.randn(t.shape, generator=g) / t[0].numel() ** 0.5 for t in feats]
    params = [p for p in mod.parameters()]
    names = [k for k, _ in mod.named_parameters()]
    grads = torch.autograd.grad(sum((t * d).sum() for t, d in zip(feats, dfe)), params)
    mod64 = bref.ResNet50FPN(3, w, f).double()
    mod64.load_state_dict(OrderedDict((k, v.double()) for k, v in st.items()))
    feats64 = mod64(xo.double())
    grads64 = torch.autograd.grad(sum((t * d.double()).sum() for t, d in zip(feats64, dfe)), list(mod64.parameters()))

    m = ResNet50FPN(3, w, f).load_state_dict(st).set_compute_dtype(mode)
    got = m.forward_features(x.numpy())
    for i, (a, b) in enumerate(zip(got, feats)):
        b = b.detach().permute(0, 2, 3, 1).numpy()
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 5e-5 * max(1.0, np.abs(b).max()), i
    m.backward(x.numpy(), [d.permute(0, 2, 3, 1).numpy() for d in dfe])
    worst = []
    for k, g32, g64 in zip(names, grads, grads64):
        w64 = g64.numpy().ravel()
        nrm = np.linalg.norm(w64) + 1e-30
        rel_ref = np.linalg.norm(g32.numpy().ravel() - w64) / nrm
        rel_hip = np.linalg.norm(m.grad(k).ravel() - w64) / nrm
        worst.append((rel_hip, k, rel_ref))
        # a ReLU input within rounding of 0 takes the other branch than the oracle's and moves every gradient computed
        # after it by one element's share (see tests/test_gpu_resnet_unet.py): 2e-2 bounds that, the median pins the rest
        assert rel_hip <= max(4 * rel_ref, 2e-2), (k, rel_hip, rel_ref)
    assert np.median([t[0] for t in worst]) <= 5e-3, sorted(worst)[-3:]
    # frozen BatchNorm: buffers unchanged by a step, parameters move
    before = m.state_dict()
    m.apply_gradients(lr=1e-3, weight_decay=0.0)
    after = m.state_dict()
    for k in before:
        same = torch.equal(before[k], after[k])
        assert same == (".bn" in k or "downsample.1" in k), k

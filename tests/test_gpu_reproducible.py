"""Bitwise reproducibility of the backward pass (no float atomics, fixed-order reductions: DESIGN.md): the same model on
the same batch must give bit-identical gradients every time, although the weight-gradient kernels share the CUs with the
main stream's kernels.  This is the test that would have caught the packed-fp32 -> v_cvt_f64_f32 hazard (build.py,
tools/scan_pk_f64_hazard.py): a few lanes of a BatchNorm-backward sum changed in about 2 % of the passes."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


@pytest.mark.parametrize("mode,reps", [("bfloat16", 400), ("float32", 200)])
def test_backward_is_bitwise_reproducible(mode, reps):
    from race_probe import probe
    bad = probe(mode, reps)
    assert not bad, {k: v[:2] for k, v in bad.items()}


def test_bitwise_reproducible_with_bucketed_exchange():
    from race_probe import probe
    assert not probe("bfloat16", 200, emulate=2)


@pytest.mark.parametrize("mode,reps", [("bfloat16", 40), ("float32", 30)])
def test_bitwise_reproducible_at_the_benched_shape(mode, reps):
    """UNet(3,1,32) on 64 x 128 x 128: the kernel configurations and the co-residency of bench.py's step (float32: the
    wave-specialised conv / weight-gradient kernels and the stem kernels next to each other on two streams)."""
    from race_probe import probe
    assert not probe(mode, reps, features=32, batch=64, size=128)


@pytest.mark.parametrize("model,features,size", [("resnet", 16, 64), ("cnn3", 64, 64)])
def test_other_models_are_bitwise_reproducible(model, features, size):
    """The ResNet-18-encoder U-Net and the 3-layer CNN share the side stream, the slab workspace and the event rings with
    the U-Net code path: the same bit-identical-gradients property."""
    from race_probe import probe
    assert not probe("float32", 100, features=features, batch=4, size=size, model=model)


@pytest.mark.parametrize("kw", [dict(features=32, batch=8, size=64, emulate=2), dict(features=32, batch=2, size=64, emulate=2, model="resnet"),
                                dict(features=64, batch=1, size=256, model="resnet")])
def test_bf16_flow_with_transposed_convs_on_planes_is_bitwise_reproducible(kw):
    """Widths in whole 32-channel blocks: the transposed convs run on the plane kernels too (tap-group contraction, strided
    weight gradient on the side stream, BatchNorm-backward sums in the 2x2 stride-2 input gradient's epilogue), for the plain
    U-Net and for the ResNet-encoder model's bfloat16 flow (parity-class input gradients, owned dY planes), alone and with
    the bucketed gradient exchange emulated."""
    from race_probe import probe
    assert not probe("bfloat16", 40, **kw)

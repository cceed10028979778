"""CPU-only: the C-ABI library loads, exports every symbol include/rfi_hip.h declares, and the
host-side logic (parameter table, default init, argument validation, error strings) behaves.
No compute entry point is called here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import unet_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "rfi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rfi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rfi_toolbox_amd import _lib
    names = _header_functions()
    assert len(names) >= 50
    for n in names:
        assert hasattr(_lib.lib, n), f"librfi_hip.so lacks {n}"
    assert sorted(_lib.EXPORTED) == names, set(_lib.EXPORTED) ^ set(names)
    assert _lib.lib.rfi_abi_version() == 1


def test_host_only_entry_points():
    from rfi_toolbox_amd import _lib
    assert _lib.lib.rfi_profile_family_count() == 10
    fams = [_lib.lib.rfi_profile_family_name(i).decode() for i in range(10)]
    assert fams[0] == "conv_igemm_mfma" and "wgrad_igemm_mfma" in fams
    assert _lib.device_count() >= 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from rfi_toolbox_amd import _lib
    from rfi_toolbox_amd.models import UNet
    h = C.c_void_p()
    assert _lib.lib.rfi_ctx_create(0, C.byref(h)) != 0
    assert len(_lib.lib.rfi_last_error()) > 0
    with pytest.raises(RuntimeError):
        UNet(3, 1, 4)
    with pytest.raises(RuntimeError):
        UNet(3, 1, 4, device="cpu")


def test_parameter_table_and_default_init_match_reference(golden_dir):
    from rfi_toolbox_amd.models import default_init_state, unet_entries
    g = np.load(os.path.join(golden_dir, "unet_f8_b4_s64.npz"))
    ent = unet_entries(3, 1, 8)
    assert [(n, tuple(s)) for n, s, _ in ent] == [(n, tuple(s)) for n, s, _ in unet_ref.unet_entries(3, 1, 8)]
    torch.manual_seed(1234)                      # the seed the golden run used before UNet(3,1,8)
    sd = default_init_state(3, 1, 8)
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), g[f"state0/{k}"]), k      # bit-identical initial weights
    assert len(unet_entries(3, 1, 4, depth=5)) == len(unet_ref.unet_entries(3, 1, 4, depth=5))


def test_python_argument_validation_without_gpu():
    from rfi_toolbox_amd.models import UNet
    from rfi_toolbox_amd.preprocessing import Preprocessor, patchify
    with pytest.raises(ValueError):
        UNet(0, 1, 4)
    with pytest.raises(ValueError):
        UNet(3, 1, -2)
    with pytest.raises(ValueError):
        Preprocessor(np.zeros((4, 4)))
    p = patchify(np.arange(16).reshape(4, 4), (2, 2), 2)
    assert p[0, 0].tolist() == [[0, 1], [4, 5]] and p[1, 1].tolist() == [[10, 11], [14, 15]]


def test_batch_writer_shards_round_trip(tmp_path):
    """reference shard format (datasets/batched_dataset.py:126-175): batch_%03d.pt + metadata.json."""
    import json

    from rfi_toolbox_amd.datasets import BatchWriter, TorchDataset, load_batches
    imgs = torch.arange(7 * 4 * 4 * 3, dtype=torch.float32).reshape(7, 4, 4, 3)
    labs = (torch.arange(7 * 16) % 3 == 0).to(torch.uint8).reshape(7, 4, 4)
    w = BatchWriter(tmp_path / "shards", samples_per_batch=3)
    w.add_batch(TorchDataset(imgs[:2], labs[:2]))
    w.add_batch(TorchDataset(imgs[2:], labs[2:]))
    meta = w.finalize()
    assert sorted(p.name for p in (tmp_path / "shards").iterdir()) == ["batch_000.pt", "batch_001.pt", "batch_002.pt",
                                                                      "metadata.json"]
    assert meta["num_samples"] == 7 and meta["num_batches"] == 3 and meta["samples_per_batch"] == 3
    shard = torch.load(tmp_path / "shards" / "batch_000.pt", weights_only=False)
    assert set(shard) == {"images", "labels"} and len(shard["images"]) == 3
    ds = load_batches(tmp_path / "shards")
    assert torch.equal(ds.images, imgs) and torch.equal(ds.labels, labs)
    assert json.load(open(tmp_path / "shards" / "metadata.json"))["dtype"] == "float32"
    ds.save_to_disk(tmp_path / "one.pt")
    back = TorchDataset.load_from_disk(tmp_path / "one.pt")
    assert torch.equal(back.images, imgs) and back[1]["label"].shape == (4, 4)
    with pytest.raises(AssertionError):
        TorchDataset(imgs.double(), labs)


def test_load_batches_reads_reference_written_shards(golden_dir):
    """shards written by the REFERENCE's BatchWriter (tests/golden/make_shard_fixture.py): ragged
    shard sizes (5,3,5,3 -- the reference flushes everything it holds in samples_per_batch chunks), tensors that
    are views into a larger stored block, and a metadata.json whose shapes are the hard-coded 1024x1024."""
    import numpy as np
    from rfi_toolbox_amd.datasets import load_batches
    d = os.path.join(golden_dir, "ref_shards")
    want = np.load(os.path.join(d, "expected.npz"))
    ds = load_batches(d)
    assert len(ds) == 16 and tuple(ds.images.shape) == (16, 32, 32, 3) and tuple(ds.labels.shape) == (16, 32, 32)
    assert ds.images.dtype == torch.float32 and ds.labels.dtype == torch.uint8
    np.testing.assert_array_equal(ds.images.numpy(), want["images"])
    np.testing.assert_array_equal(ds.labels.numpy(), want["labels"])
    assert ds.metadata["num_samples"] == 16 and ds.metadata["num_batches"] == 4
    assert ds.metadata["image_shape"] == [1024, 1024, 3]        # the reference's constant, not the data's shape
    assert ds[3]["image"].shape == (32, 32, 3) and ds[3]["label"].dtype == torch.uint8

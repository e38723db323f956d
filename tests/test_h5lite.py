"""The pure-Python HDF5 reader against files written by a real HDF5 library (tests/golden/make_h5_fixtures.py: h5py 3.3 /
HDF5 1.10.6, Keras `save_weights` layout), and the Keras `.h5` -> C-ABI weight path built on it."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from tensorflowasr_amd import checkpoint, h5lite


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(os.path.join(GOLDEN, "keras_weights_small.npz")))


@pytest.mark.parametrize("name,count", [("keras_weights_small.h5", 38), ("keras_weights_gzip.h5", 59)])
def test_keras_h5_files_read_bit_exactly(ref, name, count):
    """contiguous datasets (h5py defaults) and chunked + shuffle + gzip ones; groups through v1 B-trees / local heaps /
    symbol nodes; fixed-length string array attributes and variable-length string attributes (global heap)."""
    path = os.path.join(GOLDEN, name)
    f = h5lite.H5File(path)
    assert f.attrs["backend"] == "tensorflow" and f.attrs["keras_version"] == "2.8.0"
    layers = [v.decode() for v in f.attrs["layer_names"]]
    assert "decoder_conformer_block_0" in layers and set(layers) == set(f.keys())
    w = h5lite.keras_weights(path)
    assert len(w) == count == len(f.visit())
    for k, v in w.items():
        assert v.dtype == ref[k].dtype and v.shape == ref[k].shape and np.array_equal(v, ref[k]), k
    g = f["decoder_conformer_block_0"]
    assert "ctc_decoder" in g.keys() and "nope" not in g
    ds = f["dense_53/ctc_decoder/dense_53/kernel:0"]
    assert ds.shape == (8, 8) and np.array_equal(ds.read(), ref["ctc_decoder/dense_53/kernel:0"])
    with pytest.raises(KeyError):
        f["dense_53/missing"]


def test_not_hdf5_and_unsupported_features_fail_loudly(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 64)
    with pytest.raises(h5lite.H5Error, match="not an HDF5 file"):
        h5lite.H5File(str(p))


def test_keras_h5_to_abi_names(ref):
    """variable names of the toy CTCDecoder file map onto the C-ABI names; values travel unchanged"""
    w = checkpoint.keras_h5_to_abi(os.path.join(GOLDEN, "keras_weights_small.h5"))
    blk = "decoder_conformer_block_0"
    expect = {"project/kernel": "ctc_decoder/dense_53/kernel:0",
              blk + "/ff_module_1/ffn1/kernel": "ctc_decoder/%s/ff_module_1/dense_54/kernel:0" % blk,
              blk + "/ff_module_2/ffn2/bias": "ctc_decoder/%s/ff_module_2/dense_57/bias:0" % blk,
              blk + "/mhsa_module/mha/projection_kernel": "ctc_decoder/%s/mhsa_module/multi_head_attention_13/projection_kernel:0" % blk,
              blk + "/conv_module/bn/moving_variance": "ctc_decoder/%s/conv_module/batch_normalization_13/moving_variance:0" % blk,
              blk + "/conv_module/dw_conv/depthwise_kernel": "ctc_decoder/%s/conv_module/dw_conv/depthwise_kernel:0" % blk,
              blk + "/ln/beta": "ctc_decoder/%s/layer_normalization_69/beta:0" % blk,
              "fully_connected/bias": "ctc_decoder/fully_connected/bias:0"}
    assert len(w) == 38
    for abi, keras in expect.items():
        assert np.array_equal(w[abi], ref[keras]), abi
    from tensorflowasr_amd.models import CTCDecoder
    dec = CTCDecoder(num_classes=1332, dmodel=144, num_blocks=1, head_size=36, num_heads=4, kernel_size=32)
    assert set(w) == {n for n, _ in dec._names_and_shapes()}          # exactly the tensors a CTCDecoder handle expects


def test_latest_format_bounds_file():
    """written with libver='latest': superblock 3, version-2 object headers (OHDR), link messages, compact attributes
    announced by an attribute-info message, version-4 contiguous layouts, a big-endian float64 dataset"""
    ref = dict(np.load(os.path.join(GOLDEN, "keras_weights_latest.npz")))
    w = h5lite.keras_weights(os.path.join(GOLDEN, "keras_weights_latest.h5"))
    assert set(w) == set(ref) == {"a/kernel:0", "a/bias:0", "b/kernel:0", "b/bias:0"}
    for k in ref:
        assert np.array_equal(w[k], ref[k]) and w[k].dtype.itemsize == ref[k].dtype.itemsize, k

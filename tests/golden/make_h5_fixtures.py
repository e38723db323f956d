"""Generates the HDF5 fixtures for tests/test_h5lite.py with a REAL HDF5 library, so that the pure-Python reader
(tensorflowasr_amd/h5lite.py) is checked against files it did not write.

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py      (h5py 3.3.0 / HDF5 1.10.6; the system python has no h5py)

Files (all small):
  keras_weights_small.h5   the layout of keras `Model.save_weights(path.h5)` (keras/saving/hdf5_format.py
                           save_weights_to_hdf5_group: root attrs layer_names / backend / keras_version, one group per
                           layer with attr weight_names and one dataset per weight, nested by the '/' in its name), with
                           the variable names of the reference's CTCDecoder at toy shapes; h5py defaults (contiguous).
  keras_weights_gzip.h5    the same tensors, chunked + shuffle + gzip, many layers (B-tree with several leaves, object
                           header continuation blocks).
  keras_weights_small.npz  the tensors of both files, for comparison."""
import os

import h5py
import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(0)


def layer_weights():
    d, H, hs, k, V = 8, 2, 4, 5, 11
    blk = "decoder_conformer_block_0"
    w = {}

    def add(layer, name, shape, dtype=np.float32):
        w.setdefault(layer, []).append(("%s/%s:0" % (layer if "/" not in name else name.split("|")[0], name), rng.standard_normal(shape).astype(dtype)))

    # keras names: "<layer scope>/<var>:0"; sub-layers of a custom layer keep their own scope below the top-level layer
    def var(layer, scope, vname, shape):
        w.setdefault(layer, []).append(("%s/%s:0" % (scope, vname), rng.standard_normal(shape).astype(np.float32)))

    var("dense_53", "ctc_decoder/dense_53", "kernel", (d, d))
    var("dense_53", "ctc_decoder/dense_53", "bias", (d,))
    for ff, (a, b), ln in (("ff_module_1", (54, 55), 65), ("ff_module_2", (56, 57), 68)):
        s = "ctc_decoder/%s/%s" % (blk, ff)
        var(blk, s + "/layer_normalization_%d" % ln, "gamma", (d,))
        var(blk, s + "/layer_normalization_%d" % ln, "beta", (d,))
        var(blk, s + "/dense_%d" % a, "kernel", (d, 4 * d))
        var(blk, s + "/dense_%d" % a, "bias", (4 * d,))
        var(blk, s + "/dense_%d" % b, "kernel", (4 * d, d))
        var(blk, s + "/dense_%d" % b, "bias", (d,))
    s = "ctc_decoder/%s/mhsa_module" % blk
    var(blk, s + "/layer_normalization_66", "gamma", (d,))
    var(blk, s + "/layer_normalization_66", "beta", (d,))
    for nm, shp in (("query_kernel", (H, d, hs)), ("key_kernel", (H, d, hs)), ("value_kernel", (H, d, hs)),
                    ("projection_kernel", (H, hs, d)), ("projection_bias", (d,))):
        var(blk, s + "/multi_head_attention_13", nm, shp)
    s = "ctc_decoder/%s/conv_module" % blk
    var(blk, s + "/layer_normalization_67", "gamma", (d,))
    var(blk, s + "/layer_normalization_67", "beta", (d,))
    var(blk, s + "/pw_conv_1", "kernel", (1, d, 2 * d))
    var(blk, s + "/pw_conv_1", "bias", (2 * d,))
    var(blk, s + "/dw_conv", "depthwise_kernel", (k, d, 1))
    var(blk, s + "/dw_conv", "pointwise_kernel", (1, d, 2 * d))
    var(blk, s + "/dw_conv", "bias", (2 * d,))
    for nm in ("gamma", "beta", "moving_mean", "moving_variance"):
        var(blk, s + "/batch_normalization_13", nm, (2 * d,))
    var(blk, s + "/pw_conv_2", "kernel", (1, 2 * d, d))
    var(blk, s + "/pw_conv_2", "bias", (d,))
    var(blk, "ctc_decoder/%s/layer_normalization_69" % blk, "gamma", (d,))
    var(blk, "ctc_decoder/%s/layer_normalization_69" % blk, "beta", (d,))
    var("fully_connected", "ctc_decoder/fully_connected", "kernel", (d, V))
    var("fully_connected", "ctc_decoder/fully_connected", "bias", (V,))
    return w


def save_keras_style(path, layers, **dset_kw):
    with h5py.File(path, "w") as f:
        f.attrs["layer_names"] = np.asarray([n.encode("utf8") for n in layers])      # as keras: array of bytes
        f.attrs["backend"] = "tensorflow"                                               # str -> variable-length string
        f.attrs["keras_version"] = "2.8.0"
        for lname, ws in layers.items():
            g = f.create_group(lname)
            g.attrs["weight_names"] = np.asarray([n.encode("utf8") for n, _ in ws])
            for n, val in ws:
                ds = g.create_dataset(n, val.shape, dtype=val.dtype, **dset_kw)
                if val.shape:
                    ds[:] = val
                else:
                    ds[()] = val


def main():
    layers = layer_weights()
    save_keras_style(os.path.join(OUT, "keras_weights_small.h5"), layers)
    # many layers + chunked / filtered datasets + a scalar and an int64 dataset
    big = dict(layers)
    for i in range(10):
        big["extra_layer_%02d" % i] = [("model/extra_layer_%02d/kernel:0" % i, rng.standard_normal((3, 5 + i)).astype(np.float32)),
                                       ("model/extra_layer_%02d/bias:0" % i, rng.standard_normal((5 + i,)).astype(np.float64))]
    big["counters"] = [("model/counters/step:0", np.arange(7, dtype=np.int64))]
    save_keras_style(os.path.join(OUT, "keras_weights_gzip.h5"), big, chunks=True, compression="gzip", shuffle=True)
    flat = {}
    for ws in big.values():
        for n, v in ws:
            flat[n] = v
    np.savez_compressed(os.path.join(OUT, "keras_weights_small.npz"), **flat)
    for p in ("keras_weights_small.h5", "keras_weights_gzip.h5", "keras_weights_small.npz"):
        print(p, os.path.getsize(os.path.join(OUT, p)))


if __name__ == "__main__":
    main()


def latest_format_fixture():
    """keras_weights_latest.h5: the same layout written with libver='latest' (superblock 3, version-2 object headers,
    link messages instead of symbol tables, version-4 contiguous layouts, a big-endian float64 dataset)."""
    r = np.random.default_rng(1)
    path = os.path.join(OUT, "keras_weights_latest.h5")
    vals = {}
    with h5py.File(path, "w", libver="latest") as f:
        f.attrs["layer_names"] = np.asarray([b"a", b"b"])
        f.attrs["backend"] = "tensorflow"
        for layer in ("a", "b"):
            g = f.create_group(layer)
            g.attrs["weight_names"] = np.asarray([("%s/kernel:0" % layer).encode(), ("%s/bias:0" % layer).encode()])
            vals["%s/kernel:0" % layer] = r.standard_normal((4, 5)).astype(np.float32)
            vals["%s/bias:0" % layer] = r.standard_normal((5,)).astype(">f8")
            g.create_dataset("%s/kernel:0" % layer, data=vals["%s/kernel:0" % layer])
            g.create_dataset("%s/bias:0" % layer, data=vals["%s/bias:0" % layer])
    np.savez(os.path.join(OUT, "keras_weights_latest.npz"), **vals)


if __name__ == "__main__":
    latest_format_fixture()

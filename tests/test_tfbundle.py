"""tensorflowasr_amd/tfbundle.py against the one TensorFlow tensor bundle the reference ships (the VAD SavedModel's
variables): the index parses, every block and every tensor passes its CRC32C, the object graph yields the Keras
variable names.  The file is read where it lies (not copied into the repository)."""
import os
import shutil

import numpy as np
import pytest

from tensorflowasr_amd import tfbundle

REF = "/root/reference/vad/online_vad_model"
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is only present in the build container")


def test_crc32c_known_answers():
    assert tfbundle.crc32c(b"") == 0
    assert tfbundle.crc32c(b"123456789") == 0xE3069283                  # the CRC-32C check value (RFC 3720)
    assert tfbundle.crc32c(bytes(32)) == 0x8A9136AA                      # RFC 3720 B.4: 32 bytes of zeros
    assert tfbundle.masked_crc32c(b"123456789") == (((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xA282EAD8) & 0xFFFFFFFF


@needs_ref
def test_reference_bundle_reads_with_all_checksums():
    b = tfbundle.Bundle(tfbundle.checkpoint_prefix(REF), verify=True)
    assert b.num_shards == 1 and len(b.entries) == 19
    assert b.entries["cnn1/kernel/.ATTRIBUTES/VARIABLE_VALUE"]["shape"] == (5, 80, 80)
    for k in b.keys():
        b.tensor(k)                                                       # raises on a checksum mismatch
    v = b.variables_by_name()
    assert len(v) == 18 and sum(a.size for a in v.values()) == 96801
    assert v["online_cnn_vad/conv1d/kernel:0"].shape == (5, 80, 80) and v["online_cnn_vad/conv1d/kernel:0"].dtype == np.float32
    assert np.array_equal(v["online_cnn_vad/conv1d/kernel:0"], b.tensor("cnn1/kernel/.ATTRIBUTES/VARIABLE_VALUE"))
    g = b.object_graph()
    assert len(g) == 120 and g[0][0]["cnn1"] == 2
    assert tfbundle.checkpoint_prefix(REF + "/variables/variables.index") == REF + "/variables/variables"


@needs_ref
def test_corruption_is_detected(tmp_path):
    for name in os.listdir(REF + "/variables"):
        shutil.copy(os.path.join(REF, "variables", name), tmp_path / name)
    for name in os.listdir(tmp_path):
        os.chmod(tmp_path / name, 0o644)
    data = tmp_path / "variables.data-00000-of-00001"
    raw = bytearray(data.read_bytes())
    raw[1000] ^= 0x01
    data.write_bytes(bytes(raw))
    b = tfbundle.Bundle(str(tmp_path / "variables"))
    with pytest.raises(tfbundle.BundleError, match="checksum mismatch"):
        for k in b.keys():
            b.tensor(k)
    idx = tmp_path / "variables.index"
    raw = bytearray(idx.read_bytes())
    raw[40] ^= 0x01
    idx.write_bytes(bytes(raw))
    with pytest.raises(tfbundle.BundleError):
        tfbundle.Bundle(str(tmp_path / "variables"))
    with pytest.raises(tfbundle.BundleError, match="bad table magic"):
        (tmp_path / "x.index").write_bytes(b"\0" * 100)
        tfbundle.Bundle(str(tmp_path / "x"))

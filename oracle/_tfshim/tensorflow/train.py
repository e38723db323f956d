"""tf.train: only list_variables, over the stand-in's checkpoints (keras/_impl.py Model.save_weights)."""
import numpy as np


def list_variables(ckpt_dir_or_file):
    d = np.load(ckpt_dir_or_file + ".shim-ckpt.npz")
    out = [(k.replace("|", "/"), list(d[k].shape)) for k in d.files]
    return sorted(out + [("_CHECKPOINTABLE_OBJECT_GRAPH", [])])


def load_variable(ckpt_dir_or_file, name):
    return np.load(ckpt_dir_or_file + ".shim-ckpt.npz")[name.replace("/", "|")]


class Feature:
    def __init__(self, **kw):
        self.__dict__.update(kw)


FloatList = Int64List = BytesList = Feature

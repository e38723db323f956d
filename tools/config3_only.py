"""BASELINE config 3 alone (bench.extra_config3): python tools/config3_only.py [steps]   -- one JSON line"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tensorflowasr_amd import _lib  # noqa: E402

lib = _lib.lib()
out = bench.extra_config3(lib, torch.device("cuda:0"), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20, with_cpu=False)
print(json.dumps(out))

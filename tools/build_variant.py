"""Builds libmi355asr.so variants that differ in one translation unit (kernel experiments on the GPU box):

    python tools/build_variant.py NAME SRC.hip [-DFLAG ...] [ENV=VALUE ...]

compiles tensorflowasr_amd/csrc/SRC.hip with the extra flags (ENV=VALUE pairs are exported first, e.g. PP_NPOOL=18 re-runs
tools/gen_pp.py), links it with the other objects of the last regular build into tensorflowasr_amd/build/variants/NAME.so
(select it with MI355ASR_LIB=<path>), and restores the generated sources.  The regular library is not touched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tensorflowasr_amd import build as b  # noqa: E402


def main():
    name, src = sys.argv[1], sys.argv[2]
    flags = [a for a in sys.argv[3:] if a.startswith("-")]
    env = dict(a.split("=", 1) for a in sys.argv[3:] if "=" in a and not a.startswith("-"))
    objdir = os.path.join(b.HERE, "build")
    vdir = os.path.join(objdir, "variants")
    os.makedirs(vdir, exist_ok=True)
    gen = [sys.executable, os.path.join(ROOT, "tools", "gen_pp.py")]
    if "PP_NPOOL" in env:
        subprocess.check_call(gen, env=dict(os.environ, **env), stderr=subprocess.DEVNULL)
    try:
        obj = os.path.join(vdir, name + "_" + src.replace(".hip", ".o"))
        subprocess.check_call([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", obj])
    finally:
        if "PP_NPOOL" in env:
            subprocess.check_call(gen, stderr=subprocess.DEVNULL)
    objs = [obj if s == src else os.path.join(objdir, s.replace(".hip", ".o")) for s in b.SOURCES]
    out = os.path.join(vdir, name + ".so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    print(out)


if __name__ == "__main__":
    main()

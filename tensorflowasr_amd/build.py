"""Builds libmi355asr.so (HIP kernels + C ABI) for gfx950 with hipcc.  hipcc cross-compiles without a GPU.

    python -m tensorflowasr_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi355asr.so")
SOURCES = ["api.hip", "api_chunk.hip", "api_translator.hip", "blocks.hip", "frontend.hip", "beam.hip", "fused.hip", "fft_stft.hip", "attention_lds.hip", "subconv.hip", "bf16.hip", "leaf.hip", "wavpick.hip"]
HEADERS = ["common.h", "launch.h", "beam.h", "wstream.h", "model.h", os.path.join("..", "..", "include", "mi355asr.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("MI355ASR_EXTRA_HIPCC_FLAGS", "").split()      # e.g. -DMI355ASR_DIAG_KERNELS (timing-only kernel variants)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_example(force, verbose)
    return LIB


EXAMPLE_SRC = os.path.join(HERE, "..", "examples", "asr_session.cpp")
EXAMPLE_BIN = os.path.abspath(os.path.join(HERE, "..", "examples", "asr_session"))   # git-ignored, travels with gpurun


def build_example(force=False, verbose=True):
    """examples/asr_session.cpp: the reference's C++ `Session` on the C ABI, linked against libmi355asr.so."""
    src = os.path.abspath(EXAMPLE_SRC)
    if not os.path.exists(src):
        return None
    inc = os.path.abspath(os.path.join(HERE, "..", "include"))
    if force or _stale(EXAMPLE_BIN, [src, LIB, os.path.join(inc, "mi355asr.h")]):
        cmd = [_hipcc(), "-std=c++17", "-O2", "-I" + inc, src, "-L" + HERE, "-lmi355asr", "-Wl,-rpath,$ORIGIN/../tensorflowasr_amd", "-Wl,-rpath," + HERE, "-o", EXAMPLE_BIN]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return EXAMPLE_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

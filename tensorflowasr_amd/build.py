"""Builds libmi355asr.so (HIP kernels + C ABI) for gfx950 with hipcc.  hipcc cross-compiles without a GPU.

    python -m tensorflowasr_amd.build [--force]
"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi355asr.so")
SOURCES = ["api.hip", "api_chunk.hip", "api_translator.hip", "blocks.hip", "frontend.hip", "beam.hip", "beam_device.hip", "fused.hip", "fused_pp.hip", "fused_ns.hip", "fft_stft.hip", "attention_lds.hip", "attention_split.hip", "attention_split64.hip", "subconv.hip", "bf16.hip", "stream256.hip", "gemm_ring.hip", "leaf.hip", "wavpick.hip"]
HEADERS = ["common.h", "launch.h", "beam.h", "wstream.h", "model.h", "prep_sched.inc", "prep2_sched.inc", "pp_units.inc", "pp_layout.inc", "refmath.h", "split_f16.h", "env.h", "refmath_tables.inc", os.path.join("..", "..", "include", "mi355asr.h")]
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


def _parse_resource_remarks(text):
    """`-Rpass-analysis=kernel-resource-usage` remarks -> {kernel: {vgprs, agprs, sgprs, scratch, lds, occupancy}}"""
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch",
            "LDS Size [bytes/block]": "lds", "Occupancy [waves/SIMD]": "occupancy"}
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(.*?): (\d+) \[-Rpass-analysis", line)
        if m and cur is not None and m.group(1) in keys:
            cur[keys[m.group(1)]] = int(m.group(2))
    return out


def resource_report():
    """{kernel: resources} of the last build: the reports of the current SOURCES only (a stale report of a removed source
    must not fail or pass the check)"""
    rep = {}
    objdir = os.path.join(HERE, "build")
    for src in SOURCES:
        f = os.path.join(objdir, src.replace(".hip", ".resources.json"))
        if os.path.exists(f):
            rep.update(json.load(open(f)))
    return rep


# A kernel that touches scratch is slow twice on this path: its own spill traffic, and the dispatches AROUND it pay for
# the scratch set-up (measured: dwconv 15 -> 30 us next to a spilling neighbour, profiles/r02_ring_experiments.md).
# Kernels allowed to keep a private segment (cold paths, listed with the reason):
SCRATCH_ALLOWED = ()   # none: every kernel of the library is scratch-free


def check_no_scratch(report=None, strict=None):
    """strict (default: MI355ASR_BUILD_STRICT != 0, i.e. on): raise when a kernel spills; otherwise only warn -- another
    compiler version or extra flags (-O0, -g) can make a kernel spill, and the library that was just linked is still usable."""
    if strict is None:
        strict = os.environ.get("MI355ASR_BUILD_STRICT", "1") != "0"
    bad = {k: v["scratch"] for k, v in (report or resource_report()).items()
           if v.get("scratch", 0) > 0 and not any(a in k for a in SCRATCH_ALLOWED)}
    if bad:
        msg = "kernels with scratch (register spills): %s" % bad
        if strict:
            raise RuntimeError(msg + "  (MI355ASR_BUILD_STRICT=0 turns this into a warning)")
        sys.stderr.write("warning: " + msg + "\n")


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        rep = obj[:-2] + ".resources.json"
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + hdrs) or not os.path.exists(rep):
            cmd = [hipcc] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            # the resource remarks are parsed below; everything else the compiler said (warnings) stays visible
            noise = ("remark:", "-Rpass-analysis", "^", "~")
            for ln in r.stderr.splitlines():
                if "warning:" in ln or "error:" in ln:
                    sys.stderr.write(ln + "\n")
            with open(rep, "w") as f:
                json.dump(_parse_resource_remarks(r.stderr), f, indent=1, sort_keys=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_example(force, verbose)
    check_no_scratch()
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

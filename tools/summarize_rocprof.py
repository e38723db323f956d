"""Turns rocprofv3 CSV output (gpurun_out/prof_*) into the committed summaries under profiles/.

  python tools/summarize_rocprof.py <round-tag> <stats_dir> [<fetch_dir> <write_dir> [<mfma_dir>]]

  profiles/<tag>_kernel_stats.csv   copy of rocprofv3 --kernel-trace --stats kernel_stats
  profiles/<tag>_summary.md         per-kernel table (calls, avg us, share) + PMC bytes per launch
  profiles/pmc_traffic.json         {bench kernel category: HBM bytes per launch} read by bench.py (roofline.traffic)

PMC handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in units of 1024 B
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024), collected in separate passes; on gfx950 FETCH_SIZE reports 1/2
of the bytes of wide (16 B/lane) coalesced reads, which is what every load on this path is, so it is doubled.
WRITE_SIZE is uncalibrated on gfx950 and reported as is.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CATEGORY = [
    ("stft_kernel", "stft"), ("fft_stft_split_kernel", "stft"), ("attention_lds_kernel", "attention"), ("attention_split_kernel", "attention"), ("subconv144_kernel", "subconv"), ("subconv144_split_kernel", "subconv"), ("leaf_conv_pool", "stft"), ("leaf_gather", "mel"), ("leaf_pcen", "mel"), ("utt_max_kernel", "utt_max"), ("mel_kernel", "mel"), ("mel_band_kernel", "mel"), ("db_norm_kernel", "mel"), ("subconv_kernel", "subconv"),
    ("stream_gemm_kernel", "sublinear"), ("attention_kernel", "attention"), ("dwconv_kernel", "dwconv"), ("dwconv_tile_kernel", "dwconv"),
    ("collapse_kernel", "collapse"), ("ff1_qkv_kernel", "ff1_qkv"), ("out_glu_kernel", "out_glu"),
    ("tail_ff2_kernel", "tail_ff2"), ("tail_ff1_ld_kernel", "tail_ff1"), ("tail_ff2_ld_kernel", "tail_ff2"), ("ff1_qkv_ld_kernel", "ff1_qkv"), ("out_glu_ld_kernel", "out_glu"), ("pp_out_glu_kernel", "out_glu"), ("pp_head_kernel", "ctc_head"), ("topn_reg_kernel", "topn"), ("tail_ff2_ring_kernel", "tail_ff2"), ("ff1_qkv_ring_kernel", "ff1_qkv"), ("out_glu_ring_kernel", "out_glu"), ("out_glu_split_kernel", "out_glu"), ("pp_sublinear_kernel", "sublinear"), ("sublinear_split_kernel", "sublinear"), ("sublinear_split_ld_kernel", "sublinear"), ("refmath_eval_kernel", "refmath"), ("topn_kernel", "topn"), ("head_ld_kernel", "ctc_head"), ("pick_kernel", "pick"), ("gather_kernel", "gather"), ("stream256_kernel", "enc_stack"),
]


def category(name):
    if "pp_block_kernel" in name:       # fused_pp.hip: <TAIL, FF1>
        a = [v.strip() for v in name[name.index("<") + 1:name.index(">")].split(",")]
        return {("true", "true"): "tail_ff1", ("true", "false"): "tail_ff2", ("false", "true"): "ff1_qkv"}[(a[0], a[1])]
    for key, cat in CATEGORY:
        if key in name:
            return cat
    if "<" not in name:
        return name
    args = [a.strip() for a in name[name.index("<") + 1:name.index(">")].split(",")]
    if "chain2_kernel" in name:
        return "ffn" if args[-1] == "0" else "conv_tail"
    if "gemm_ring_kernel" in name:      # <epilogue, LN, row tiles, ring slots>
        return "gemm_ring<" + ",".join(args[:4]) + ">"
    if "subconv_split_ring_kernel" in name:
        return "subconv" if args[1] == "144" else "subconv" + args[1]
    if "gemm_rows_kernel" in name:
        return {"0": "ctc_project", "1": "attn_out", "2": "qkv", "3": "pw1_glu", "4": "ctc_head"}[args[3]]
    return name


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    if not f:
        raise SystemExit("no file matches " + pattern)
    return f[0]


def pmc_avg(d, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(one(os.path.join(d, "**", "*counter_collection.csv"))) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return {k: tot[k] / n[k] for k in tot}


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = one(os.path.join(stats_dir, "**", "*kernel_stats.csv"))
    shutil.copy(ks, os.path.join(out, tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks)))
    fetch = write = busy = gui = {}
    if len(sys.argv) >= 5:
        fetch, write = pmc_avg(sys.argv[3], "FETCH_SIZE"), pmc_avg(sys.argv[4], "WRITE_SIZE")
    if len(sys.argv) >= 6:
        busy, gui = pmc_avg(sys.argv[5], "SQ_VALU_MFMA_BUSY_CYCLES"), pmc_avg(sys.argv[5], "GRBM_GUI_ACTIVE")
    lines = ["# rocprofv3 summary `%s`" % tag, "",
             "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --min-timed-s 0 "
             "--no-cpu-baseline ...` (tools/profile_round.sh; B=64 x 10 s, ConformerCTC(S), fp32); PMC columns from separate "
             "`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (FETCH doubled per MI355X_MICROARCH.md, HBM section).", "",
             "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), from a separate pass; "
             "GRBM_GUI_ACTIVE carries ~25k cycles of per-dispatch overhead under PMC, so the figure is pessimistic "
             "for kernels shorter than ~100 us.", "",
             "| kernel | category | calls | avg us | share % | FETCH MB/launch (x2) | WRITE MB/launch | HBM MB/launch | MFMA busy % |",
             "|---|---|---|---|---|---|---|---|---|"]
    traffic = {}
    for r in rows:
        name = r["Name"]
        cat = category(name)
        fb = 2 * fetch.get(name, 0.0) * 1024
        wb = write.get(name, 0.0) * 1024
        if name in fetch or name in write:
            traffic[cat] = round(fb + wb)
        mb = ("%.0f" % (100.0 * busy[name] / (1024.0 * gui[name] / 8.0))) if name in busy and gui.get(name) else "-"
        lines.append("| `%s` | %s | %s | %.1f | %s | %s | %s | %s | %s |" % (
            name.replace("void ", "").replace("(anonymous namespace)::", ""), cat, r["Calls"],
            float(r["AverageNs"]) / 1e3, r["Percentage"],
            ("%.1f" % (fb / 1e6)) if name in fetch else "-", ("%.1f" % (wb / 1e6)) if name in write else "-",
            ("%.1f" % ((fb + wb) / 1e6)) if (name in fetch or name in write) else "-", mb))
    # the traced steps (warm-up included): one collapse_kernel per step; kernel time per step under the profiler
    steps = next((int(r["Calls"]) for r in rows if "collapse_kernel" in r["Name"]), 0)
    if steps:
        total_ms = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
        lines += ["", "Kernel time per step under rocprofv3: **%.3f ms** (sum of all kernels' total duration / %d traced steps, warm-up "
                  "included).  The profiler slows the clock (MI355X_MICROARCH.md, DVFS note): compare with the un-profiled figure below, "
                  "not with another box's." % (total_ms / steps, steps)]
    open(os.path.join(out, tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
    if traffic:
        json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

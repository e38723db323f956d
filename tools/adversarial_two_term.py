"""How far does the two-term fp16 operand scheme (pp_block_kernel, pp_out_glu_kernel, attention_split_kernel<2>) drift from
the fp64 oracle when the power-of-two operand BOUNDS are far above the values -- and how far do the three-term bf16 kernels
(exact fp32 products) on the same weights?  One variant per process (the kernel family is read from the environment once):
    python tools/adversarial_two_term.py                       # the default family
    MI355ASR_PP=0 MI355ASR_PP_OUTGLU=0 MI355ASR_ATTN_TERMS=3 python tools/adversarial_two_term.py
prints one line per case: "CASE <name> <max|d| vs oracle> <max|oracle|>".  tests/test_gpu_parity.py runs both and compares.

Cases (block 1 of a two-block ConformerEncoder(S), 8 x 250 tokens so that the fused path runs):
  plain     Glorot weights, N(0, 1) input (the reference point)
  gamma30   every LayerNorm gamma x 30, beta = 3: all bounds grow with the values
  w1col50   one ffn1 column of each FFModule x 50: the hidden row's bound follows L1(that column), every other hidden
            value sits 50 x lower in its fp16 pair
  quiet1000 gamma_i x 1000 for one feature whose input equals the row mean (normalised value ~ 0): the static q / k / v bound
            counts 1000 |W_i| sqrt(143) that the values never reach -- bound / value >= 2^10
  all       the three together
"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, maxdiff, small_cfg  # noqa: E402
from tensorflowasr_amd.models import ConformerEncoder  # noqa: E402

P = "conformer_block_1/"
LNS = ["ff_module_1/ln", "ff_module_2/ln", "mhsa_module/ln", "conv_module/ln", "ln"]
QUIET = 17


def variant(w0, name):
    w = {k: np.array(v, copy=True) for k, v in w0.items()}
    if name in ("gamma30", "all"):
        for ln in LNS[:4]:
            w[P + ln + "/gamma"] *= 30.0
            w[P + ln + "/beta"] += 3.0
    if name in ("w1col50", "all"):
        for ff, col in (("ff_module_1", 5), ("ff_module_2", 300)):
            w[P + ff + "/ffn1/kernel"][:, col] *= 50.0
    if name in ("quiet1000", "all"):
        for ln in LNS[:4]:
            w[P + ln + "/gamma"][QUIET] *= 1000.0
    return w


def main():
    cfg = small_cfg(2)
    w0 = co.encoder_weights(cfg, seed=0)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((8, 250, 144)).astype(np.float32)
    xq = x.copy()
    xq[..., QUIET] = 0.0
    xq[..., QUIET] = xq.sum(-1) / 143.0                 # = the row mean: the normalised feature is ~ 1e-8
    for name in ("plain", "gamma30", "w1col50", "quiet1000", "all"):
        w = variant(w0, name)
        e = ConformerEncoder(**encoder_kwargs(cfg))
        e.load_weights(w, by_name=False)
        xi = xq if name in ("quiet1000", "all") else x
        ref = co.conformer_block(xi.astype(np.float64), w, "conformer_block_1", 36)
        got = e.conformer_block(1, xi).cpu().numpy()
        print("CASE %s %.4e %.4e" % (name, maxdiff(got, ref), float(np.abs(ref).max())), flush=True)
        del e


if __name__ == "__main__":
    main()

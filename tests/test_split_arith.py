"""The arithmetic behind the split-bf16 kernels (subconv.hip, fused.hip ring kernels, leaf.hip), emulated in NumPy:
an fp32 value is exactly the sum of three bf16 terms; products of bf16 terms are exact in fp32; the six term pairs with
i + j <= 2 reproduce a dot product to ~2^-24 of sum |x w| -- the accuracy class of an fp32 FMA chain -- while two
terms (three pairs) are ~2^-16.  (The device kernels split activations by truncation and weights by round-to-nearest;
both are exact decompositions.)"""
import numpy as np


def bf16_trunc(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def bf16_rne(x):
    u = x.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def split(x, n, rnd):
    terms, r = [], x.astype(np.float32).copy()
    for _ in range(n):
        t = rnd(r)
        terms.append(t)
        r = (r - t).astype(np.float32)            # exact: the remainder has fewer significant bits
    return terms, r


def test_three_terms_are_an_exact_decomposition():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-30, 30, 200000))).astype(np.float32)
    for rnd in (bf16_trunc, bf16_rne):
        terms, rest = split(x, 3, rnd)
        assert np.all(rest == 0)                                            # nothing left after three terms
        total = terms[0].astype(np.float64) + terms[1].astype(np.float64) + terms[2].astype(np.float64)
        assert np.array_equal(total.astype(np.float32), x)
        for t in terms:                                                    # every term is a bf16 value
            assert np.all((t.view(np.uint32) & 0xFFFF) == 0)


def test_term_products_are_exact_in_fp32_and_six_pairs_match_an_fp32_chain():
    rng = np.random.default_rng(1)
    K = 1296                                                               # the subsampling convolution's K
    x = np.maximum(rng.standard_normal((64, K)), 0).astype(np.float32)     # post-ReLU activations
    w = (rng.standard_normal((K, 16)) / np.sqrt(K)).astype(np.float32)
    xs, _ = split(x, 3, bf16_trunc)
    ws, _ = split(w, 3, bf16_rne)
    p = xs[0][:, :, None].astype(np.float32) * ws[1][None].astype(np.float32)
    assert np.array_equal(p.astype(np.float64), xs[0][:, :, None].astype(np.float64) * ws[1][None].astype(np.float64))
    exact = x.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)

    def pairs(order):
        acc = np.zeros_like(exact)
        for i in range(3):
            for j in range(3):
                if i + j <= order:
                    acc += xs[i].astype(np.float64) @ ws[j].astype(np.float64)
        return acc

    err6 = np.abs(pairs(2) - exact).max() / scale.max()                     # six pairs (three terms per operand)
    err3 = np.abs(pairs(1) - exact).max() / scale.max()                     # three pairs (two terms per operand)
    chain = np.zeros((64, 16), np.float32)                                  # an fp32 accumulation chain for comparison
    for k in range(K):
        chain += x[:, k:k + 1] * w[k:k + 1, :]
    err_chain = np.abs(chain.astype(np.float64) - exact).max() / scale.max()
    assert err6 < 2.0 ** -22 and err6 < err_chain                           # better than the fp32 chain's rounding
    assert 2.0 ** -20 < err3 < 2.0 ** -14                                   # two terms: the 1e-5 class, not fp32

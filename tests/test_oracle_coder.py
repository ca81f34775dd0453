"""CPU: self-consistency of the entropy-model restatement (parity unpinned, see oracle/tdvc_ref/coder.py)."""
import numpy as np
import pytest
import torch

from oracle.tdvc_ref import coder as oc
from tdvc_amd.synth import fill_parameters


def test_pmf_to_quantized_cdf_properties():
    rng = np.random.default_rng(0)
    for n in (3, 17, 64):
        p = rng.random(n) ** 4
        p[rng.integers(0, n, 3)] = 1e-12          # nearly-empty bins must still get freq >= 1
        p = p / p.sum()
        cdf = oc.pmf_to_quantized_cdf(p.tolist(), 16)
        assert cdf[0] == 0 and cdf[-1] == 65536 and len(cdf) == n + 1
        assert all(b > a for a, b in zip(cdf, cdf[1:]))


def test_rans_roundtrip_with_bypass():
    rng = np.random.default_rng(1)
    pmfs = [np.array([0.6, 0.3, 0.05, 0.05]), np.array([0.1, 0.1, 0.5, 0.2, 0.05, 0.05])]
    cdfs, sizes, offsets = [], [], [-1, -2]
    width = 8
    for p in pmfs:
        c = oc.pmf_to_quantized_cdf(p.tolist(), 16)
        sizes.append(len(c))
        cdfs.append(c + [0] * (width - len(c)))
    idx = rng.integers(0, 2, 5000).tolist()
    syms = []
    for i in idx:
        n = sizes[i] - 2                       # symbols offsets[i] .. offsets[i]+n-1 are in-table
        s = int(rng.integers(offsets[i] - 40, offsets[i] + n + 40)) if rng.random() < 0.05 else int(rng.integers(offsets[i], offsets[i] + n))
        syms.append(s)
    data = oc.rans_encode(syms, idx, cdfs, sizes, offsets)
    assert len(data) % 4 == 0
    assert oc.RansDecoder(data).decode(idx, cdfs, sizes, offsets) == syms


@pytest.fixture(scope="module")
def small_coder():
    torch.manual_seed(0)
    m = oc.MVCoder(N=128).eval()
    h = torch.nn.Module(); h.add_module("mvCoder", m); fill_parameters(h)
    m.update(force=True)
    return m


def test_tables(small_coder):
    m = small_coder
    eb, gc = m.entropy_bottleneck, m.gaussian_conditional
    assert eb._quantized_cdf.shape[0] == 128 and gc._quantized_cdf.shape[0] == 64
    for t, ln in ((eb._quantized_cdf, eb._cdf_length), (gc._quantized_cdf, gc._cdf_length)):
        for i in range(t.shape[0]):
            row = t[i, : int(ln[i])].tolist()
            assert row[0] == 0 and row[-1] == 65536 and all(b > a for a, b in zip(row, row[1:]))
    # scale table + index rule
    s = torch.tensor([0.05, 0.11, 0.5, 300.0])
    idx = gc.build_indexes(s)
    assert idx.tolist()[0] == 0 and idx.tolist()[-1] == 63


def test_compress_decompress_roundtrip(small_coder):
    m = small_coder
    x = torch.randn(1, 64, 64, 64, generator=torch.Generator().manual_seed(3)) * 0.5
    enc = m.compress(x)
    dec = m.decompress(enc["strings"], enc["shape"])
    y_hat_enc = enc["_debug"][0]["y_hat"]
    assert torch.equal(dec["y_hat"], y_hat_enc), "decoder must reproduce the encoder's y_hat exactly"
    assert len(enc["strings"][0][0]) > 8 and len(enc["strings"][1][0]) >= 8
    # forward pass: rate estimate is close to the actual coded size (same model, eval quantisation)
    with torch.no_grad():
        out = m(x)
    bits_est = float(sum((-torch.log2(l)).sum() for l in out["likelihoods"].values()))
    bits_act = 8.0 * (len(enc["strings"][0][0]) + len(enc["strings"][1][0]))
    assert 0.5 * bits_act < bits_est < 2.0 * bits_act + 512, (bits_est, bits_act)


def test_forward_train_mode_runs(small_coder):
    m = small_coder
    m.train()
    out = m(torch.randn(1, 64, 64, 64) * 0.5)
    assert out["x_hat"].shape == (1, 64, 64, 64) and float(m.aux_loss()) > 0
    m.eval()


def test_lower_bound_backward_rule():
    """compressai `LowerBoundFunction` (ops/bound_ops.py): forward max(x, bound); the gradient passes where
    x >= bound OR grad < 0.  Hand-computed vector at, below and above the bound, for the oracle's LowerBound and for
    the product's parameter-space twin (GDN beta / gamma chain)."""
    from tdvc_amd.model import coder as dc
    x0 = [0.05, 0.05, 0.11, 0.11, 0.20, 0.20]          # below, below, at, at, above, above   (bound 0.11)
    g0 = [-2.0, 3.0, -2.0, 3.0, -2.0, 3.0]
    want = [-2.0, 0.0, -2.0, 3.0, -2.0, 3.0]           # below the bound only the negative gradient passes
    for LB in (oc.LowerBound, dc.LowerBound):
        lb = LB(0.11)
        x = torch.tensor(x0, requires_grad=True)
        y = lb(x)
        assert torch.allclose(y, torch.tensor([0.11, 0.11, 0.11, 0.11, 0.20, 0.20]))
        y.backward(torch.tensor(g0))
        assert x.grad.tolist() == want, (LB.__module__, x.grad.tolist())
    # the Gaussian conditional's two bounds, hand-derived: bits = -log2 max(lik, 1e-9), lik = Phi((.5-v)/s) - Phi((-.5-v)/s),
    # s = max(scale, 0.11).  A scale of 0.05 (below the bound) with the sample one bin off the mean: d bits / d s < 0
    # (a wider Gaussian would code it cheaper), so the gradient must reach `scale`; with the sample ON the mean the
    # gradient is positive and must be blocked.
    gc = oc.GaussianConditional()
    sc = torch.tensor([0.05, 0.05, 0.5], requires_grad=True)
    y = torch.tensor([1.0, 0.0, 1.0])
    _, lik = gc(y, sc, torch.zeros(3), False)
    (-torch.log2(lik)).sum().backward()
    assert sc.grad[0] < 0 and sc.grad[1] == 0 and sc.grad[2] != 0, sc.grad
    s, v = 0.11, 1.0                                     # element 0 by hand (fp64), scale pinned at the bound
    import math
    a, b = (0.5 - v) / s, (-0.5 - v) / s
    Phi = lambda t: 0.5 * math.erfc(-t / math.sqrt(2))
    pdf = lambda t: math.exp(-0.5 * t * t) / math.sqrt(2 * math.pi)
    lik0 = Phi(a) - Phi(b)
    want0 = -(1.0 / (max(lik0, 1e-9) * math.log(2))) * (b * pdf(b) - a * pdf(a)) / s
    assert abs(float(sc.grad[0]) - want0) <= 2e-3 * abs(want0), (float(sc.grad[0]), want0)
    # likelihood floor: a sample 40 sigma out has lik < 1e-9; -log2 of the floored value still pushes it back
    sc2 = torch.tensor([0.5], requires_grad=True)
    y2 = torch.tensor([20.0], requires_grad=True)
    _, lik2 = gc(y2, sc2, torch.zeros(1), False)
    assert float(lik2) == pytest.approx(1e-9)
    (-torch.log2(lik2)).sum().backward()
    assert float(y2.grad) >= 0.0 and float(sc2.grad) <= 0.0      # (the tail pdf underflows in fp32: the sign is what is pinned)


def test_gaussian_conditional_is_the_discretised_normal():
    """An independent pin of the Gaussian rate term (no compressai here, no vectors in the reference): the likelihood of a
    quantised value q = round(y - mu) under N(mu, s^2) is the mass of [q - 1/2, q + 1/2], computed here in float64 with
    math.erf; s is bounded below by 0.11 and the likelihood by 1e-9 (compressai's GaussianConditional defaults)."""
    import math
    g = torch.Generator().manual_seed(3)
    gc = oc.GaussianConditional()
    y = torch.randn(4000, generator=g) * 6.0
    mu = torch.randn(4000, generator=g) * 2.0
    s = torch.rand(4000, generator=g) * 4.0 + 0.01                     # some below the 0.11 bound
    out, lik = gc(y.view(1, 1, 1, -1), s.view(1, 1, 1, -1), mu.view(1, 1, 1, -1), training=False)
    out, lik = out.flatten(), lik.flatten()
    Phi = lambda t: 0.5 * (1.0 + math.erf(t / math.sqrt(2.0)))
    worst = 0.0
    for i in range(0, 4000, 7):
        q = round(float(y[i]) - float(mu[i]))                             # Python rounds half to even, like torch.round
        assert float(out[i]) == pytest.approx(q + float(mu[i]), abs=1e-5)
        sd = max(float(s[i]), 0.11)
        want = max(Phi((q + 0.5) / sd) - Phi((q - 0.5) / sd), 1e-9)
        worst = max(worst, abs(float(lik[i]) - want) / want if want > 1e-6 else abs(float(lik[i]) - want))
    assert worst < 2e-3, worst                                            # fp32 erfc differences in the far tail
    # and it is a distribution: the masses of all integers sum to one
    ks = torch.arange(-400, 401, dtype=torch.float32)
    for sd in (0.11, 0.7, 3.0, 40.0):
        tot = float(gc.likelihood(ks, torch.full_like(ks, sd)).double().sum())
        assert abs(tot - 1.0) < 1e-5, (sd, tot)


def test_factorized_prior_is_a_distribution():
    """EntropyBottleneck: per channel the cumulative logits are increasing and the likelihoods of all integers sum to one
    (whatever the parameters: the construction guarantees it; a transcription error in the matrix / bias / factor chain does
    not survive this)"""
    torch.manual_seed(5)
    eb = oc.EntropyBottleneck(6)
    with torch.no_grad():
        for n, p_ in eb.named_parameters():
            if "quantiles" not in n:
                p_.add_(torch.randn_like(p_) * 0.3)                       # away from the initialisation
    x = torch.linspace(-60, 60, 2401).view(1, 1, -1).expand(6, 1, -1)
    with torch.no_grad():
        lc = eb.logits_cumulative(x)
        assert bool((lc[..., 1:] >= lc[..., :-1]).all())
        ks = torch.arange(-300, 301, dtype=torch.float32).view(1, 1, -1).expand(6, 1, -1)
        lik = eb.likelihood(ks).double()
    assert bool((lik >= 0).all())
    tot = lik.sum(dim=-1).flatten()
    assert float((tot - 1.0).abs().max()) < 1e-4, tot.tolist()

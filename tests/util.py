"""Shared helpers of the test-suite (tests may use the oracle; the product may not)."""
import os

import numpy as np

from oracle.spec import GnnSpec as OSpec
from oracle import compact as oc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 parity tolerances of the HIP path against the float64 oracle evaluated on the same
# fp32-rounded inputs and weights (north_star: "within a stated fp32 tolerance").
# forward:  |got - ref| <= FWD_RTOL*|ref| + FWD_ATOL*max(1, max|ref|)   (sums of ~150 fp32 products
# with cancellation: the absolute part has to follow the magnitude of the terms)
FWD_RTOL, FWD_ATOL = 2e-4, 2e-6
GRAD_RTOL, GRAD_ATOL_REL = 5e-4, 2e-5     # atol = GRAD_ATOL_REL * max|reference|


def ospec(spec):
    return OSpec(n_nodes=spec.n_nodes, n_channels=spec.n_channels, feat_dim=spec.feat_dim,
                 n_mp_layers=spec.n_mp_layers, share_weights=spec.share_weights)


def random_inputs(rng, B, N, C=4, ref_topology=True, density=0.5):
    """Feature statistics of SURVEY.md 8(d3) (measured from the reference simulator)."""
    x = np.concatenate([rng.normal(0.84, 0.39, size=(B, N, C)), rng.normal(0.60, 0.21, size=(B, N, C)),
                        np.full((B, N, 1), 10.0)], axis=2)
    e = rng.normal(0.88, 0.11, size=(B, N, C))
    if ref_topology and N > 2:
        adj = oc.random_topology(rng, B, N)
    else:
        adj = (rng.uniform(size=(B, N, N)) < density).astype(np.float64)
    return x.astype(np.float32), e.astype(np.float32), adj


def f32_params(spec, rng, random_bias=True):
    """Oracle params whose values are exactly representable in fp32 (kept as float64 arrays)."""
    P = oc.init_params(ospec(spec), rng, np.float64, random_bias=random_bias)
    return oc.cast_params(oc.cast_params(P, np.float32), np.float64)


def assert_close(got, ref, rtol, atol, what=""):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref)
    bound = atol + rtol * np.abs(ref)
    bad = err > bound
    assert not bad.any(), "%s: %d/%d out of tolerance, max err %.3e (ref scale %.3e)" % (
        what, bad.sum(), bad.size, err.max(), np.abs(ref).max())


def assert_fwd_close(got, ref, what=""):
    scale = max(1.0, float(np.abs(ref).max()))
    assert_close(got, ref, FWD_RTOL, FWD_ATOL * scale, what)


def assert_grad_close(got, ref, what=""):
    scale = float(np.abs(ref).max()) or 1.0
    assert_close(got, ref, GRAD_RTOL, GRAD_ATOL_REL * scale, what)


def golden_forward_cases():
    f = np.load(os.path.join(GOLDEN, "golden_forward_n4.npz"))
    cases = sorted({k.split('/')[0] for k in f.files})
    return f, cases


def golden_keras_list(f, case, tag):
    """Weights of a golden case, re-ordered from the reference's creation order
    (GNN: stage-major node-minor; Dense: node-major layer-minor) to the Keras-shaped list
    order used by the engine and the oracle (Dense: layer-major node-minor)."""
    ws = [f['%s/%s/w/%03d' % (case, tag, i)] for i in range(80)]
    lst = ws[:48]
    for i in range(4):
        for k in range(4):
            lst += [ws[48 + k * 8 + i * 2], ws[48 + k * 8 + i * 2 + 1]]
    return lst


def golden_feed(f, case):
    pre = case + '/in/'
    return {k[len(pre):]: f[k] for k in f.files if k.startswith(pre)}

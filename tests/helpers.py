"""Shared helpers for the test-suite (synthetic series + theta generators)."""
import numpy as np


from carma_pack_amd.synth import irregular_series, log_quads_from_roots, prior_like_theta, theta_batch  # noqa: F401,E402


def assert_parity(got, want, rtol=1e-10, what=""):
    """north_star bar: |got-want| <= 1e-10 |want| where finite; identical -inf/NaN pattern."""
    got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
    assert got.shape == want.shape
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), "%s: finite pattern differs at %s" % (
        what, np.flatnonzero(np.isfinite(got) != fin)[:10])
    assert np.array_equal(np.isneginf(got), np.isneginf(want)), "%s: -inf pattern differs" % what
    if fin.any():
        rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
        assert rel.max() <= rtol, "%s: max rel err %.3e at %d (got %r want %r)" % (
            what, rel.max(), np.flatnonzero(fin)[rel.argmax()], got[fin][rel.argmax()], want[fin][rel.argmax()])
        return rel.max()
    return 0.0

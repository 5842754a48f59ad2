"""The two-phase bisection of fss_sigma_kernel (scanpy_amd/csrc/fuzzy.hip) restated in numpy: float32 exponentials while
|psum - target| > 1e-5 + (k - 1) * 1e-6, float64 from the first step that comes closer.  The claim the kernel rests on:
every decision (stop / direction) is the float64 loop's, so the final `mid` is bit-identical -- also when the float32
exponential is several ulp worse than numpy's (the GPU's v_exp_f32 path).  No GPU, no oracle import: both loops live here;
the float64 one follows umap.umap_.smooth_knn_dist as SURVEY.md appendix A.1 states it."""
import numpy as np
import pytest


def _bisect_f64(d, start=None):
    n, k = d.shape
    target = np.log2(k)
    lo = np.zeros(n)
    hi = np.full(n, np.inf)
    mid = np.ones(n)
    it0 = np.zeros(n, dtype=np.int64)
    if start is not None:
        lo, hi, mid, it0 = (a.copy() for a in start)
    active = it0 < 64
    for it in range(64):
        run = active & (it0 <= it)
        if not run.any():
            continue
        with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
            terms = np.where(d[:, 1:] > 0, np.exp(-(d[:, 1:].astype(np.float64) / mid[:, None])), 1.0)
        psum = np.zeros(n)
        for j in range(k - 1):  # the kernel's summation order
            psum = psum + terms[:, j]
        stop = run & (np.abs(psum - target) < 1e-5)
        active &= ~stop
        run &= ~stop
        up = run & (psum > target)
        dn = run & ~(psum > target)
        hi = np.where(up, mid, hi)
        mid = np.where(up, (lo + hi) / 2.0, mid)
        lo = np.where(dn, mid, lo)
        mid = np.where(dn, np.where(np.isinf(hi), mid * 2.0, (lo + hi) / 2.0), mid)
    return mid


def _bisect_two_phase(d, rng, ulp_noise):
    n, k = d.shape
    target = np.log2(k)
    undecided = 1e-5 + 1e-6 * (k - 1)
    lo = np.zeros(n)
    hi = np.full(n, np.inf)
    mid = np.ones(n)
    it_at = np.zeros(n, dtype=np.int64)
    in_p1 = d[:, 1:].max(axis=1) < 1e30
    n_f32_steps = 0
    for it in range(64):
        run = in_p1 & (mid > 1e-30) & (mid < 1e30)
        in_p1 &= run
        if not run.any():
            break
        inv = (1.0 / mid).astype(np.float32)
        with np.errstate(over="ignore", invalid="ignore"):
            x = (d[:, 1:] * inv[:, None]).astype(np.float32)
            t32 = np.exp(-x).astype(np.float32)
        # a worse exponential than numpy's: +- ulp_noise ulp
        t32 = (t32 * (1.0 + rng.uniform(-ulp_noise, ulp_noise, t32.shape) * 2.0**-24)).astype(np.float32)
        terms = np.where(d[:, 1:] > 0, t32, np.float32(1.0)).astype(np.float64)
        psum = np.zeros(n)
        for j in range(k - 1):
            psum = psum + terms[:, j]
        close = run & (np.abs(psum - target) <= undecided)
        in_p1 &= ~close
        run &= ~close
        n_f32_steps += int(run.sum())
        up = run & (psum > target)
        dn = run & ~(psum > target)
        hi = np.where(up, mid, hi)
        mid = np.where(up, (lo + hi) / 2.0, mid)
        lo = np.where(dn, mid, lo)
        mid = np.where(dn, np.where(np.isinf(hi), mid * 2.0, (lo + hi) / 2.0), mid)
        it_at = np.where(run, it + 1, it_at)
    return _bisect_f64(d, start=(lo, hi, mid, it_at)), n_f32_steps


def _rows(rng, n, k, kind):
    if kind == "gauss":
        d = np.sort(np.abs(rng.standard_normal((n, k))).astype(np.float32) * rng.choice([1e-3, 1.0, 50.0], (n, 1)).astype(np.float32), axis=1)
    elif kind == "duplicates":
        d = np.sort(rng.integers(0, 4, (n, k)).astype(np.float32), axis=1)
    elif kind == "tight":  # nearly equal distances: sigma runs to tiny values
        d = np.sort((1.0 + rng.random((n, k)) * 1e-6).astype(np.float32), axis=1)
    else:  # wide dynamic range, subnormals, huge values
        d = np.sort((10.0 ** rng.uniform(-42, 35, (n, k))).astype(np.float32), axis=1)
    d[:, 0] = 0.0
    rho = np.where((d > 0).any(axis=1), np.where(d > 0, d, np.inf).min(axis=1), 0.0).astype(np.float32)
    return (d - rho[:, None]).astype(np.float32)  # the kernel bisects on d_j - rho (float32 subtraction)


@pytest.mark.parametrize("k", [3, 15, 16, 30, 100])
@pytest.mark.parametrize("kind", ["gauss", "duplicates", "tight", "wide"])
def test_two_phase_bisection_takes_the_float64_decisions(k, kind):
    rng = np.random.default_rng(k * 7 + len(kind))
    d = _rows(rng, 4000, k, kind)
    ref = _bisect_f64(d)
    got, n32 = _bisect_two_phase(d, rng, ulp_noise=4.0)
    assert np.array_equal(ref.view(np.int64), got.view(np.int64)), int((ref != got).sum())
    if kind == "gauss":
        assert n32 > 4000 * 5, n32  # (most steps are float32 ones, or the kernel gained nothing)

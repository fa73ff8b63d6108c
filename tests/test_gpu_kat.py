"""Device known-answer tests for the scalar functions the update kernels are built from (SURVEY.md section 8 rows a5,
a11), through the C ABI (slamgpu_kat), both kernel builds, against the reference's edge-case vectors
(tests/golden/kat_functions.npz, generated from the reference's own objects):

  trigonometricOffset (core.cpp:460-477): sweep over [-20, 20] plus +-pi, +-2pi, +-7, |a| > 2pi.  The strict build replays
      the reference's double-constant ladder (bit-exact expected); the fast build's wrap_pi is one rounding: same angle
      modulo 2 pi to 4e-6 rad and inside [-pi, pi] (the two conventions may pick opposite ends at exactly +-pi).
  gaussEvaluate D = 2, 3 (fastslam2.cpp:127-163), including nearly rank-deficient S: the device solves the triangular
      system by forward substitution (strict) or in closed form (fast) where the reference uses a JacobiSVD
      pseudo-inverse; compared on log w with a tolerance that scales with cond(S) and the size of the exponent (stated
      in the test).

and the degenerate-weights status bit (SLAMGPU_STATUS_DEGENERATE) of the resampling stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def sg():
    import slam_amd
    assert slam_amd.device_count() >= 1
    return slam_amd


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_trig_offset_edge_cases_on_device(sg, kat, math_mode):
    a, exp = kat["trig_in"], kat["trig_out"]
    got = sg.kat(math_mode, 0, a)
    if math_mode == 0:
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))  # same operations, same bits
        return
    assert np.all(np.abs(got) <= np.float32(np.pi) + 4e-6)
    d = (got.astype(np.float64) - exp.astype(np.float64) + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() <= 4e-6, np.abs(d).max()
    # away from the +-pi seam the two agree as numbers
    inner = np.abs(np.abs(exp) - np.pi) > 1e-4
    assert np.abs(got[inner] - exp[inner]).max() <= 4e-6


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
@pytest.mark.parametrize("D", [2, 3])
def test_gauss_evaluate_edge_cases_on_device(sg, kat, D, math_mode):
    """gaussEvaluate on the reference's vectors incl. near-singular S.  w = exp(E) / C; a relative perturbation delta of
    the Cholesky factor moves E by ~|E| * cond(S) * delta, so the comparison is on log w with
    tolerance (|E| + 1) * cond(S) * 2^-22 (strict) / 2^-20 (fast: rcp / rsq are 1-ulp, one FMA chain), floored at
    1e-5 / 1e-4; where the reference underflows to 0 (or below 1e-30) the device must be tiny too."""
    S, v, exp = kat["gauss%d_S" % D].astype(np.float64), kat["gauss%d_v" % D].astype(np.float64), kat["gauss%d_out" % D]
    n = S.shape[0]
    if D == 2:
        data = np.stack([v[:, 0], v[:, 1], S[:, 0, 0], S[:, 1, 0], S[:, 1, 1]], 1)
    else:
        data = np.stack([v[:, 0], v[:, 1], v[:, 2], S[:, 0, 0], S[:, 1, 0], S[:, 1, 1], S[:, 2, 0], S[:, 2, 1], S[:, 2, 2]], 1)
    got = sg.kat(math_mode, D - 1, data.astype(f32))
    ev = np.linalg.eigvalsh(S)
    cond = ev[:, -1] / np.maximum(ev[:, 0], 1e-300)
    E = np.array([0.5 * v[i] @ np.linalg.solve(S[i], v[i]) for i in range(n)])
    unit = 2.0 ** -22 if math_mode == 0 else 2.0 ** -20
    floor = 1e-5 if math_mode == 0 else 1e-4
    checked = 0
    for i in range(n):
        if not np.isfinite(exp[i]):
            continue  # garbage in the reference (failed LLT leaves the input behind, LLT.h:278-282): nothing to pin
        if exp[i] < 1e-30:
            assert got[i] < 1e-20, (i, got[i], exp[i])
            continue
        tol = max(floor, (abs(E[i]) + 1.0) * cond[i] * unit)
        if tol > 0.5:
            continue  # conditioning so bad that float32 pins nothing (cond ~ 1e6 with a large exponent)
        assert abs(np.log(float(got[i]) / float(exp[i]))) <= tol, (D, i, got[i], exp[i], cond[i], E[i], tol)
        checked += 1
    assert checked >= 48


def test_kat_rejects_bad_arguments(sg):
    with pytest.raises(sg.SlamGpuError):
        sg.kat(0, 7, np.zeros(4, f32))


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_degenerate_weights_are_flagged(sg, math_mode):
    """All likelihoods underflow (an observation 40 sigma away from every particle's prediction): sum w = 0, the
    reference divides by it (core.cpp:726-729) and so does the device (NaN weights), but the step carries
    SLAMGPU_STATUS_DEGENERATE in slamgpu_step_status and in the history; a healthy step carries 0."""
    R = np.array([0.01, 0, 0, 0.0003046], f32)
    Q = np.array([0.09, 0, 0, 0.0027415568], f32)
    N = 512
    s = sg.SlamGpu(N, 8, method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=1, math_mode=math_mode)
    ctl = np.array([[3.0, 0.0, 0.0]] * 8, f32)
    e = np.zeros((0, 2), f32)
    s.step(ctl, Q, 0.025, e, np.zeros(0, np.int32), np.array([[10.0, 0.3]], f32), R)  # one new landmark
    assert s.status() == 0
    s.step(ctl, Q, 0.025, np.array([[9.4, 0.31]], f32), np.array([0], np.int32), e, R)  # consistent re-observation
    assert s.status() == 0
    s.step(ctl, Q, 0.025, np.array([[30.0, -2.0]], f32), np.array([0], np.int32), e, R)  # absurd re-observation
    assert s.status() == 1
    _, _, _ = s.history_fetch()
    assert list(s.last_history_status) == [0, 0, 1]
    s.close()

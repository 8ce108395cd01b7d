"""GPU parity of the 'next' rows of SURVEY section 8(f): characteristic function from moments (rank 1) and the
drivers' .npz result format (rank 3)."""
import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import io as mio, synth
from mfs_amd.one_dim import filtering, moments, ss_models
from oracle import one_dim as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('N', [3, 7, 10, 20])
def test_characteristic_fn_matches_oracle_and_closed_form(N):
    rng = np.random.default_rng(N)
    zs = np.linspace(-2., 2., 257)
    mus, vs = rng.normal(scale=0.5, size=6), rng.uniform(0.3, 1.2, size=6)
    cms = np.array([[o.central_moment_of_normal(v, p) for p in range(2 * N)] for v in vs])
    cf = moments.characteristic_fn(zs, cms, mus)                       # central moments about mus
    assert cf.shape == (6, 257) and cf.dtype == np.complex128
    for b in range(6):
        npt.assert_allclose(cf[b], o.characteristic_fn(zs, cms[b], mus[b]), rtol=0, atol=1e-9)
        if N >= 7:  # the N-point rule reproduces the Gaussian characteristic function on this grid
            npt.assert_allclose(cf[b], np.exp(1j * zs * mus[b] - 0.5 * vs[b] * zs ** 2),
                                atol={7: 1e-3, 10: 1e-5, 20: 1e-9}[N])
    # scalar z, single moment vector, scaled mode
    scms = cms[2] / np.sqrt(vs[2]) ** np.arange(2 * N)
    one = moments.characteristic_fn(0.7, scms, mus[2], np.sqrt(vs[2]))
    assert np.ndim(one) == 0
    npt.assert_allclose(one, o.characteristic_fn(np.array([0.7]), scms, mus[2], np.sqrt(vs[2]))[0], atol=1e-10)


def test_post_processing_pipeline_on_filter_outputs(tmp_path):
    """dardel/benes_bernoulli/post_processing_mf.py:37-60 on this build's outputs: filter -> npz -> characteristic fn."""
    N, T, B = 7, 40, 3
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    _, c, _, mu, _ = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=4)
    cmss, means, nell = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys)
    assert mio.save_batch(str(tmp_path), 'central', N, (cmss, means, nell)) == B
    zs = np.linspace(-2, 2, 64)
    for k in range(B):
        l_cmss, l_means, l_nell = mio.load_filter_result(mio.result_filename(str(tmp_path), 'central', N, k), 'central')
        npt.assert_array_equal(l_cmss, cmss[k])
        assert float(l_nell) == nell[k]
        cf = moments.characteristic_fn(zs, l_cmss, l_means)            # (T, m), as cf_cms(zs, cmss, means) upstream
        assert cf.shape == (T, 64)
        ref = np.stack([o.characteristic_fn(zs, l_cmss[t], l_means[t]) for t in range(T)])
        npt.assert_allclose(cf, ref, atol=1e-9)
        npt.assert_allclose(np.abs(cf[:, 32 - 1:33]).max(), 1., atol=1e-2)   # |phi(0)| = 1
    with pytest.raises(ValueError):
        mio.save_filter_result(str(tmp_path / 'x.npz'), 'raw', cmss[0], means[0], nell[0])


def test_parameter_estimation_with_in_launch_gradients():
    """dardel/parameter_estimation/mf.py:37-73 as a drop-in: the objective closes over `drift(x, p1)` and
    `measurement_cond_pmf(y, x, p2)` exactly as upstream; the optimiser gets nell and a central-difference gradient from
    one batched launch (2P + 1 filters).  The reference's stored run (examples/parameter_estimation.ipynb cell 8) learnt
    [2.56, 3.36] from one T = 1000 data set generated at (3, 3); the same experiment here must land near the truth."""
    from mfs_amd import estimation
    N, T = 5, 1000
    dt, _, _, ic, drift, dispersion, emission, pmf, _ = ss_models.well_poisson(3., N)
    ys, _ = synth.well_poisson_batch(1, T, p1=3., p2=3., dt=dt, seed=12)
    calls = []

    def nell_batch(P):
        p1, p2 = P[:, 0], P[:, 1]
        _, c, _, mu, _ = moments.sde_cond_moments_tme_normal(lambda x: drift(x, p1), dispersion, dt, 2, N)
        _, _, nell = filtering.moment_filter_cms(c, mu, lambda y, x: pmf(y, x, p2), ic.cms, ic.mean,
                                                 np.repeat(ys, P.shape[0], axis=0))
        calls.append(P.shape[0])
        return nell

    f0, g0 = estimation.nell_and_grad(nell_batch, np.array([1.0, 1.0]))
    # the in-launch gradient agrees with a coarser, separately evaluated difference quotient
    fp = nell_batch(np.array([[1.0 + 1e-3, 1.0]]))[0]
    fm = nell_batch(np.array([[1.0 - 1e-3, 1.0]]))[0]
    npt.assert_allclose(g0[0], (fp - fm) / 2e-3, rtol=1e-3)
    res = estimation.minimise_nell(nell_batch, [1.0, 1.0], bounds=[(0.1, 8.), (0.1, 8.)], maxiter=60)
    assert res.fun < f0 - 10.
    assert abs(res.x[0] - 3.) < 1.2 and abs(res.x[1] - 3.) < 0.8, res.x
    assert set(calls) == {5, 1}  # every optimiser step was ONE launch of 2P + 1 filters

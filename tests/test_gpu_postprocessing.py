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

"""The forward-mode NLL gradient carried inside the kernel's time loop (C entry
mfs_filter_1d_grad, mfs_amd/csrc/filter1d_grad.hpp) -- what dardel/parameter_estimation/mf.py:37-54,70-73 obtains from JAX
autodiff through the lax.scan.  Checked against central differences of the PLAIN filter kernel (a different kernel: its
NLL also cross-checks the dual kernel's values) and of the NumPy oracle, for all three modes, both transition kinds and
parameters in the drift, the dispersion and the likelihood; then the reference's experiment: L-BFGS-B on well--Poisson."""
import math

import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import estimation, stats, sym, synth
from mfs_amd.one_dim import filtering, moments, ss_models
from oracle import one_dim as o, models as om, tme_sympy

pytestmark = pytest.mark.gpu


def _well(N):
    dt, _, _, ic, drift, dispersion, emission, pmf, _ = ss_models.well_poisson(3., N)
    return dt, ic, drift, dispersion, pmf


def test_well_poisson_gradient_matches_differences_of_the_plain_filter():
    """The parameter-estimation configuration itself: N = 7, central moments, TME-normal-2, theta = (p1 in the drift,
    p2 in the Poisson rate)."""
    N, T, B = 7, 400, 3
    dt, ic, drift, dispersion, pmf = _well(N)
    ys, _ = synth.well_poisson_batch(B, T, p1=3., p2=3., dt=dt, seed=5)

    def model(P):
        _, c, _, mu, _ = moments.sde_cond_moments_tme_normal(lambda x: drift(x, P[:, 0]), dispersion, dt, 2, N)
        return c, mu, (lambda y, x: pmf(y, x, P[:, 1]))

    theta = np.array([2.2, 2.7])
    nell, grad = estimation.nell_and_grad_forward(model, theta, ic.cms, ic.mean, ys)
    assert nell.shape == (B,) and grad.shape == (B, 2)

    def nell_batch(Pm, b):
        c, mu, lik = model(Pm)
        return filtering.moment_filter_cms(c, mu, lik, ic.cms, ic.mean, np.repeat(ys[b:b + 1], Pm.shape[0], axis=0))[2]

    for b in range(B):
        f0, g_fd = estimation.nell_and_grad(lambda Pm: nell_batch(Pm, b), theta, rel_step=1e-5)
        npt.assert_allclose(nell[b], f0, rtol=1e-9)              # two different kernels, one NLL
        npt.assert_allclose(grad[b], g_fd, rtol=1e-5, atol=1e-6)
    # one parameter point per replicate in the same launch: the rows of a multi-start / grid evaluation
    thetas = np.array([[2.2, 2.7], [3.0, 3.0], [1.5, 4.0]])
    nell_r, grad_r = estimation.nell_and_grad_forward(model, thetas, ic.cms, ic.mean, ys)
    npt.assert_allclose(nell_r[0], nell[0], rtol=1e-13)
    npt.assert_allclose(grad_r[0], grad[0], rtol=1e-10)
    n1, g1 = estimation.nell_and_grad_forward(model, thetas[2], ic.cms, ic.mean, ys[2])
    npt.assert_allclose(nell_r[2], n1, rtol=1e-13)
    npt.assert_allclose(grad_r[2], g1, rtol=1e-10)


@pytest.mark.parametrize('mode', ['raw', 'central', 'scaled'])
def test_operator_tables_tanh_drift_three_modes_against_the_oracle(mode):
    """Benes-type model with parameters in the drift amplitude, the dispersion and the logistic slope; TME-2 operator
    tables in u = tanh x.  Reference for the derivative: central differences of the NumPy oracle's NLL."""
    N, T = 5, 60
    dt = 1e-2
    ic = ss_models.benes_bernoulli(N)[3]
    ys, _ = synth.benes_bernoulli_batch(1, T, dt, seed=8)

    def model(P):
        a, s, k = P[:, 0], P[:, 1], P[:, 2]
        fns = moments.sde_cond_moments_tme(lambda x: a * sym.tanh(x), lambda _: s, dt, 2)
        lik = lambda y, x: stats.bernoulli_pmf(y, 1. / (1. + sym.exp(-k * x ** 3)))    # noqa: E731
        if mode == 'raw':
            return fns[0], lik
        return (fns[1], fns[3], lik) if mode == 'central' else (fns[2], fns[4], lik)

    def oracle_nell(th):
        import sympy as sp
        a, s, k = th
        f = tme_sympy.sde_cond_moments_tme_1d(lambda x: a * sp.tanh(x), lambda _: s, dt, 2, 2 * N)
        lik = lambda y, x: om.bernoulli_pmf(y, 1. / (1. + np.exp(-k * x ** 3)))     # noqa: E731
        if mode == 'raw':
            return o.moment_filter_rms(f[0], lik, ic.rms, ys[0])[1]
        if mode == 'central':
            return o.moment_filter_cms(f[1], f[3], lik, ic.cms, ic.mean, ys[0])[2]
        return o.moment_filter_scms(f[2], f[4], lik, ic.scms, ic.mean, math.sqrt(ic.variance), ys[0])[3]

    theta = np.array([0.9, 1.1, 0.25])
    ms0 = {'raw': ic.rms, 'central': ic.cms, 'scaled': ic.scms}[mode]
    nell, grad = estimation.nell_and_grad_forward(model, theta, ms0, None if mode == 'raw' else ic.mean, ys[0],
                                                  scale0=math.sqrt(ic.variance) if mode == 'scaled' else None, mode=mode)
    npt.assert_allclose(nell, oracle_nell(theta), rtol=1e-9)
    for j in range(3):
        h = 1e-4
        tp, tm = theta.copy(), theta.copy()
        tp[j] += h
        tm[j] -= h
        npt.assert_allclose(grad[j], (oracle_nell(tp) - oracle_nell(tm)) / (2 * h), rtol=2e-6, atol=1e-7)


def test_parameter_estimation_lbfgs_with_the_in_kernel_gradient():
    """dardel/parameter_estimation/mf.py:37-73: softplus-reparametrised (theta_1, theta_2), N = 7, TME-normal-2, T = 1000,
    L-BFGS-B.  The in-kernel gradient reaches the optimum the finite-difference objective reaches, with one filter per
    evaluation instead of 2P + 1, in no more evaluations."""
    N, T = 7, 1000
    dt, ic, drift, dispersion, pmf = _well(N)
    ys, _ = synth.well_poisson_batch(1, T, p1=3., p2=3., dt=dt, seed=12)
    softplus = lambda v: np.log1p(np.exp(v))                 # noqa: E731  (mf.py:39)

    def model(Pm):
        th = softplus(Pm)
        _, c, _, mu, _ = moments.sde_cond_moments_tme_normal(lambda x: drift(x, th[:, 0]), dispersion, dt, 2, N)
        return c, mu, (lambda y, x: pmf(y, x, th[:, 1]))

    def nell_batch(Pm):
        c, mu, lik = model(Pm)
        return filtering.moment_filter_cms(c, mu, lik, ic.cms, ic.mean, np.repeat(ys, Pm.shape[0], axis=0))[2]

    init = np.log(np.exp(np.array([1.0, 1.0])) - 1.)        # mf.py:69
    res_fw = estimation.minimise_nell_forward(model, init, ic.cms, ic.mean, ys[0], maxiter=80)
    res_fd = estimation.minimise_nell(nell_batch, init, maxiter=80)
    th_fw, th_fd = softplus(res_fw.x), softplus(res_fd.x)
    print('forward:', th_fw, res_fw.fun, res_fw.launches, ' finite differences:', th_fd, res_fd.fun, res_fd.launches)
    assert res_fw.nonfinite_evaluations == 0
    npt.assert_allclose(res_fw.fun, res_fd.fun, rtol=1e-7)
    npt.assert_allclose(th_fw, th_fd, rtol=2e-3)
    assert abs(th_fw[0] - 3.) < 1.2 and abs(th_fw[1] - 3.) < 0.8      # near the data-generating (3, 3), as the reference's stored run
    assert res_fw.launches <= res_fd.launches + 3
    g_end = estimation.nell_and_grad_forward(model, res_fw.x, ic.cms, ic.mean, ys[0])[1]
    assert np.max(np.abs(g_end)) < 1e-2 * max(1., abs(res_fw.fun))


def test_gradient_at_the_headline_order_N15_and_exact_parameter_tangents():
    """Benes--Bernoulli at N = 15 (the headline order), T = 100, theta = (drift gain, dispersion, logistic slope): the in-kernel
    forward-mode gradient against (i) central differences of the device's own plain filter in one launch and (ii) Richardson-
    extrapolated central differences of the NumPy oracle's NLL.  At N = 15 the oracle's NLL carries ~1e-10 of rounding, which a
    difference quotient of step 1e-3 turns into ~1e-6 of the gradient: (ii) is held to 2e-5, (i) -- same arithmetic on both
    sides -- to 2e-6.  The parameter tangents of the tables come from the complex-step trace (exact for ANY analytic
    dependence on theta); a model with exp(theta) in a coefficient is checked against the analytic derivative."""
    import sympy as sp
    N, T = 15, 100
    dt = 1e-2
    ic = ss_models.benes_bernoulli(N)[3]
    ys, _ = synth.benes_bernoulli_batch(1, T, dt, seed=8)

    def model(P):
        a, s, k = P[:, 0], P[:, 1], P[:, 2]
        fns = moments.sde_cond_moments_tme(lambda x: a * sym.tanh(x), lambda _: s, dt, 2)
        return fns[1], fns[3], (lambda y, x: stats.bernoulli_pmf(y, 1. / (1. + sym.exp(-k * x ** 3))))

    theta = np.array([0.9, 1.1, 0.25])
    nell, grad, fn = estimation.nell_and_grad_forward(model, theta, ic.cms, ic.mean, ys[0], return_first_nan=True)
    assert fn == -1 and np.isfinite(nell) and np.all(np.isfinite(grad))

    def dev_nell(P):
        c, mu, lik = model(P)
        return filtering.moment_filter_cms(c, mu, lik, ic.cms, ic.mean, np.repeat(ys, P.shape[0], axis=0))[2]

    h = 1e-4
    pts = np.repeat(theta[None], 7, axis=0)
    for j in range(3):
        pts[1 + 2 * j, j] += h
        pts[2 + 2 * j, j] -= h
    v = dev_nell(pts)
    npt.assert_allclose(nell, v[0], rtol=1e-9)
    fd = np.array([(v[1 + 2 * j] - v[2 + 2 * j]) / (2 * h) for j in range(3)])
    npt.assert_allclose(grad, fd, rtol=2e-6, atol=2e-6 * np.abs(fd).max())

    def oracle_nell(th):
        a, s, k = th
        f = tme_sympy.sde_cond_moments_tme_1d(lambda x: a * sp.tanh(x), lambda _: s, dt, 2, 2 * N)
        lik = lambda y, x: om.bernoulli_pmf(y, 1. / (1. + np.exp(-k * x ** 3)))     # noqa: E731
        return o.moment_filter_cms(f[1], f[3], lik, ic.cms, ic.mean, ys[0])[2]

    npt.assert_allclose(nell, oracle_nell(theta), rtol=1e-8)
    for j in range(3):
        d = []
        for hh in (2e-3, 1e-3):
            tp, tm = theta.copy(), theta.copy()
            tp[j] += hh
            tm[j] -= hh
            d.append((oracle_nell(tp) - oracle_nell(tm)) / (2 * hh))
        rich = (4 * d[1] - d[0]) / 3
        npt.assert_allclose(grad[j], rich, rtol=2e-5, atol=2e-5 * np.abs(grad).max())

    # exact tangents for a non-polynomial dependence on the parameter: drift gain exp(theta_0)
    def model_exp(P):
        fns = moments.sde_cond_moments_tme(lambda x: np.exp(P[:, 0]) * sym.tanh(x), lambda _: 1.0, dt, 2)
        return fns[1], fns[3], (lambda y, x: stats.bernoulli_pmf(y, 1. / (1. + sym.exp(-0.2 * x ** 3))))
    t_c = estimation._tables_and_tangents(model_exp, np.array([0.3]), 'central', 1e-3, 'complex-step')
    t_s = estimation._tables_and_tangents(model_exp, np.array([0.3]), 'central', 1e-3, 'stencil')
    # d/d theta of a table entry that is a polynomial in a = exp(theta): a d/da -- compare with differentiating in a directly
    def model_a(P):
        fns = moments.sde_cond_moments_tme(lambda x: P[:, 0] * sym.tanh(x), lambda _: 1.0, dt, 2)
        return fns[1], fns[3], (lambda y, x: stats.bernoulli_pmf(y, 1. / (1. + sym.exp(-0.2 * x ** 3))))
    t_a = estimation._tables_and_tangents(model_a, np.array([np.exp(0.3)]), 'central', 1e-3, 'complex-step')
    npt.assert_allclose(t_c[4], np.exp(0.3) * t_a[4], rtol=1e-14, atol=1e-300)
    npt.assert_allclose(t_s[4], t_c[4], rtol=1e-9, atol=1e-14)               # (the stencil route, kept as the fallback)


def test_nd_parameter_gradient_by_differences_in_one_launch():
    """d = 2: the gradient of the NLL with respect to model parameters (what `jax.grad` through `moment_filter_nd_cms` gives
    upstream, dardel/parameter_estimation/mf.py:37-54 carried to the N-D filter) from ONE launch of 2P + 1 filters with
    per-replicate tables (`batch_closures`, `coef_batched` / `lik_batched` of the N-D model).  theta = (dispersion scale,
    logistic offset) of the prey--predator model at N = 3; checked against central differences of the NumPy oracle's NLL."""
    from mfs_amd.multi_dims import filtering as fnd, moments as mnd, ss_models as snd
    from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, \
        gram_and_hankel_indices_graded_lexico
    from oracle import multi_dims as omd
    N, T = 3, 40
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    dt, _, _, gs, drift, _, _, _, _ = snd.prey_predator(mi)
    _, _, ogs, odrift, _, _, _ = omd.prey_predator(mi)
    ys, _ = synth.prey_predator_batch(1, T, dt, seed=21)
    theta = np.array([0.12, 0.8])

    def nell_batch(Pm):
        R = Pm.shape[0]
        per = [mnd.sde_cond_moments_tme(drift, (lambda s: (lambda x: np.array([[s * x[0], 0.], [0., s * x[1]]], dtype=object)))(s),
                                        dt, 2) for s in Pm[:, 0]]
        fns = mnd.batch_closures(per)
        pmf = lambda y, x: stats.bernoulli_pmf(y, 1. / (1. + sym.exp(-x[0] ** 3 + Pm[:, 1])))     # noqa: E731
        return fnd.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, np.repeat(ys, R, axis=0), (mi, inds), gs.cms,
                                        gs.mean)[2]

    f0, g = estimation.nell_and_grad(nell_batch, theta, rel_step=1e-5)

    def oracle_nell(th):
        odisp = lambda x: [[th[0] * x[0], 0], [0, th[0] * x[1]]]                                       # noqa: E731
        opmf = lambda y, x: om.bernoulli_pmf(y, 1. / (1. + np.exp(-x[0] ** 3 + th[1])))               # noqa: E731
        _, ocms, omean, _ = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, 2, mi)
        return omd.moment_filter_nd_cms((ocms, 'multi-index'), omean, opmf, ys[0], (mi, inds), ogs.cms, ogs.mean)[2]

    npt.assert_allclose(f0, oracle_nell(theta), rtol=1e-9)
    for i in range(2):
        h = 1e-4 * max(abs(theta[i]), 1.0)
        e = np.zeros(2); e[i] = h
        ref = (oracle_nell(theta + e) - oracle_nell(theta - e)) / (2 * h)
        npt.assert_allclose(g[i], ref, rtol=2e-5, atol=1e-7)
    assert np.all(np.abs(g) > 1e-3)

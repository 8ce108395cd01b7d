"""N-D test models, mirroring `mfs.multi_dims.ss_models` (same factory name and return tuple)."""
import numpy as np

from mfs_amd import sym, stats
from mfs_amd.utils import GaussianSumND


def prey_predator(multi_indices):
    """The prey--predator model (mfs/multi_dims/ss_models.py:40-95)."""
    dt = 1e-3
    T = 2000
    ts = np.linspace(dt, dt * T, T)
    alp, beta, delta, gamma, sigma = 4., 4., 4., 4., 0.1
    means = np.array([[1., 1.], [1., 1.]])
    covs = np.array([[[1., 0.], [0., 1.]], [[2., 0.], [0., 2.]]]) * 0.001
    weights = np.array([0.5, 0.5])
    gs = GaussianSumND.new(means, covs, weights, multi_indices)

    def drift(x):
        return x * (x[::-1] * np.array([-beta, delta]) + np.array([alp, -gamma]))

    def dispersion(x):
        return np.diag(sigma * x)

    def emission(x):
        return 1 / (1 + sym.exp(-x ** 3 + 1))

    def measurement_cond_pmf(y, x):
        return stats.bernoulli_pmf(y, emission(x[0]))

    def simulate(rng: np.random.Generator, integration_steps: int = 100):
        """Milstein path + Bernoulli measurements (:69-93); NumPy Generator instead of a JAX key."""
        ddt = dt / integration_steps
        x = gs.sampler(rng, 1)[0]
        x0 = x.copy()
        xs = np.empty((T, 2))
        for k in range(T):
            ddws = np.sqrt(ddt) * rng.standard_normal((integration_steps, 2))
            for ddw in ddws:
                x = x + drift(x) * ddt + sigma * x * ddw + 0.5 * sigma ** 2 * x * (ddw ** 2 - ddt)
            xs[k] = x
        p = 1 / (1 + np.exp(-xs[:, 0] ** 3 + 1))
        ys = (rng.random(T) < p).astype(np.float64)
        return x0, xs, ys

    return dt, T, ts, gs, drift, dispersion, emission, measurement_cond_pmf, simulate

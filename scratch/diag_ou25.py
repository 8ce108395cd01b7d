import sys, numpy as np
sys.path.insert(0, '.')
from mfs_amd import synth, stats
from mfs_amd.one_dim import filtering, moments, quadtures
from oracle import one_dim as o, models as om
for N in (20, 25):
    m = om.ou_gaussian(N)
    ys, _ = synth.ou_gaussian_batch(4, 200, seed=3)
    F, Sigma = m['F'], m['Sigma']
    _, cond_cms, _, cond_mean, _ = moments.sde_cond_moments_normal(lambda x: F * x, lambda x: Sigma)
    cmss, means, nell, fn = filtering.moment_filter_cms(cond_cms, cond_mean, lambda y, x: stats.norm_pdf(y, x, 1.), m['cms0'], m['mean0'], ys, return_first_nan=True)
    print(N, 'first_nan', fn, 'nell', nell)
    r = o.moment_filter_cms(m['cond_cms'], m['cond_mean'], m['pdf'], m['cms0'], m['mean0'], ys[0])
    print(' oracle nell', r[2], 'any nan', np.isnan(r[0]).any())
    k = fn[0] if fn[0] >= 0 else 199
    for t in (0, 1, 5, max(k-1,0)):
        rel = np.abs(cmss[0][t]-r[0][t])/np.maximum(np.abs(r[0][t]),1e-300)
        print('  t',t,'max rel moment err (even orders)', rel[::2].max(), 'mean err', abs(means[0][t]-r[1][t]))
    # quadrature on initial moments
    w,x = quadtures.moment_quadrature(m['cms0'][None,:])
    wr,xr = o.moment_quadrature(m['cms0'])
    print('  quad init: nodes err', np.abs(np.sort(x[0])-np.sort(xr)).max(), 'w err', np.abs(w[0][np.argsort(x[0])]-wr[np.argsort(xr)]).max())

import sys, numpy as np, struct
sys.path.insert(0, '.')
from mfs_amd import synth
from mfs_amd.multi_dims import filtering, moments, ss_models
from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, gram_and_hankel_indices_graded_lexico
N, T, B = 6, 100, 8
mi = generate_graded_lexico_multi_indices(2, 2*N-1); inds = gram_and_hankel_indices_graded_lexico(N, 2)
dt,_,_,gs,drift,disp,emis,pmf,sim = ss_models.prey_predator(mi)
kind = sys.argv[1] if len(sys.argv) > 1 else 'tme_2'
if kind == 'tme_normal_2':
    r,c,s,mu,mv = moments.sde_cond_moments_tme_normal(drift, disp, dt, 2, mi)
else:
    r,c,s,mu,mv = moments.sde_cond_moments_tme(drift, disp, dt, 2)
t = filtering._trace_transition((c,'index' if kind == 'tme_normal_2' else 'multi-index'),'central'); l = filtering._trace_likelihood(pmf, 2)
m, keep = filtering._model_struct(t, l)
ys,_ = synth.prey_predator_batch(B, T, dt, seed=0)
with open('tools/diag/nd_case_normal.bin' if kind == 'tme_normal_2' else 'tools/diag/nd_case.bin','wb') as f:
    f.write(struct.pack('6i', T, B, m.extent, m.n_terms, mi.shape[0], inds.shape[1]))
    f.write(keep[0].tobytes()); f.write(np.ascontiguousarray(keep[1][0], dtype=np.float64).tobytes())   # the single likelihood factor's [4] parameters
    f.write(inds.astype(np.int32).tobytes()); f.write(gs.cms.tobytes()); f.write(gs.mean.tobytes()); f.write(ys.tobytes())
print('dumped')

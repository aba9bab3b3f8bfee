# scratch: first GPU run -- parity of HIP curvefit / nnls vs oracle on golden fixtures
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from pyneapple_amd import api
from oracle import pnx_oracle as O
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests', 'golden')
def rel(a,b): return np.abs(a-b)/np.maximum(np.abs(b),1e-300)
mapping = {'g1_mono_b16':'mono','g1_mono_b8':'mono','g2_bi_reduced':'bi_reduced','g2_bi_s0':'bi_s0','g2_bi_full':'bi_full',
 'g3_tri_reduced':'tri_reduced','g3_tri_reduced_maxiter4':'tri_reduced','g3_tri_s0':'tri_s0','g3_tri_full':'tri_full',
 'g5_bi_pervoxel':'bi_reduced','g5_tri_pervoxel':'tri_reduced'}
for name, model in mapping.items():
    d = np.load(f'{G}/{name}.npz')
    if 'p0_arr' in d.files: p0,lo,hi = d['p0_arr'],d['lo_arr'],d['hi_arr']
    else: p0,lo,hi = d['p0_vals'],d['lo_vals'],d['hi_vals']
    for jac in ('fd','analytic'):
        t=time.time()
        r = api.curvefit(model, d['bvalues'], d['y'], p0, lo, hi, max_nfev=int(d['max_iter']), ftol=float(d['tol']), jac=jac)
        dt=time.time()-t
        o = O.curvefit(model, d['bvalues'], d['y'], p0, lo, hi, max_nfev=int(d['max_iter']), ftol=float(d['tol']), jac=jac)
        e_ref = rel(r['popt'].T, d['popt']).max(axis=1)
        e_or = rel(r['popt'], o['popt']).max(axis=0)
        print(f"{name:26s} {jac:8s} vs_ref max={e_ref.max():.2e} >1e-4:{(e_ref>1e-4).mean():.3f} | vs_oracle max={e_or.max():.2e} | status eq ref:{((r['status']>0)==d['success']).all()} eq oracle:{(r['status']==o['status']).mean():.3f} nfev eq:{(r['nfev']==o['nfev']).mean():.3f} {dt*1e3:.1f}ms", flush=True)
for name in ['g4_nnls_250_r2','g4_nnls_250_r1','g4_nnls_250_r3','g4_nnls_50_r2','g4_nnls_50_r0','g4_nnls_250_r2_maxiter20']:
    d = np.load(f'{G}/{name}.npz')
    t=time.time(); r = api.nnls(d['basis'], d['reg'], d['y'], int(d['max_iter'])); dt=time.time()-t
    o = O.nnls(d['basis'], d['reg'], d['y'], int(d['max_iter']))
    c, cr = r['coefficients'], d['coefficients']
    sc = np.abs(cr).max(axis=1, keepdims=True)+1e-300
    e = (np.abs(c-cr)/sc).max()
    er = (np.abs(r['residual']-d['residual'])/np.maximum(d['residual'],1e-300)).max()
    print(f"{name:28s} coeff err={e:.2e} rnorm err={er:.2e} status eq:{((r['status']==1)==d['success']).all()} iters eq oracle:{(r['iters']==o['iters']).mean():.3f} {dt*1e3:.1f}ms", flush=True)

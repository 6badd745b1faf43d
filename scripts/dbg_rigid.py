import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scenarios as S
from fluidlab_amd import _capi
hip = _capi.load_hip()
o64 = _capi.EngineLib('oracle/_build/libfe_oracle_f64.so'); o32 = _capi.EngineLib('oracle/_build/libfe_oracle_f32.so')
for variant, kw in dict(friction=dict(friction=0.5, softness=0.0), soft=dict(friction=0.1, softness=60.0), sticky=dict(friction=20.0, softness=0.0)).items():
    sc = S.stirrer_mini(shape='sphere', **kw)
    cot = S.random_cotangent(sc['N'])
    a = S.run_rigid(hip, sc, cot)
    b = S.run_rigid(o64, sc, {k: v.astype(np.float64) for k, v in cot.items()})
    c = S.run_rigid(o32, sc, cot)
    for name, r in (('hip-o64', (a, b)), ('o32-o64', (c, b)), ('hip-o32', (a, c))):
        e = np.abs(r[0]['final']['x'] - r[1]['final']['x']).max(1)
        print(variant, name, 'x err quantiles 50/90/99/max', np.quantile(e, [0.5, 0.9, 0.99, 1.0]), 'frac>1e-4', (e > 1e-4).mean(),
              'grad relL2', S.rel_l2(r[0]['action_grad'], r[1]['action_grad']), 'cos', S.cosine(r[0]['action_grad'], r[1]['action_grad']))

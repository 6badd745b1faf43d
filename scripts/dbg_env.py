import os, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import scenarios as S
from fluidlab_amd._capi import EngineLib, load_hip
from fluidlab_amd.envs import make
from fluidlab_amd.optimizer.recorder import Recorder
from fluidlab_amd.optimizer.solver import Solver
from fluidlab_amd.utils.config import load_config
from fluidlab_amd.fluidengine.effectors import Injector
import test_hip_env as T
base = np.random.RandomState(7).uniform(size=(20, 2, 3)).astype(np.float32)
Injector.random_vector_factory = staticmethod(lambda n, flux, dim: np.tile(base, (n // 20 + 1, 1, 1))[:n])
oracle64 = EngineLib('/root/repo/oracle/_build/libfe_oracle_f64.so')
hip = load_hip()
tgt = Recorder(make('LatteArt-v0', seed=0, loss=False, engine_lib=oracle64, **T.MINI)).record(write=False)
tgt32 = dict(tgt, x=[np.asarray(t, np.float32) for t in tgt['x']])
loss_o, g_o, _ = T._fwd_bwd(oracle64, tgt)
import fluidlab_amd.fluidengine.simulators.mpm_simulator as M
for store in (1, 0):
    for kw in (dict(max_substeps_local=None), dict(max_substeps_local=20, ckpt_dest='cpu'), dict(max_substeps_local=40, ckpt_dest='disk'), dict(max_substeps_local=40, ckpt_dest='cpu')):
        # engine option must be set after build: patch Engine creation
        orig = M._capi.Engine.__init__
        def patched(self, *a, **k):
            orig(self, *a, **k); self.set_option('grid_store', store)
        M._capi.Engine.__init__ = patched
        loss_g, g_g, env = T._fwd_bwd(hip, tgt32, **kw)
        M._capi.Engine.__init__ = orig
        print('store', store, kw, 'loss rel', abs(loss_g - loss_o) / abs(loss_o), 'grad relL2', S.rel_l2(g_g, g_o), 'cos', S.cosine(g_g, g_o))

import sys
sys.path.insert(0, '.')
from fluidlab_amd import _capi
from fluidlab_amd.envs import make
lib = _capi.load_hip()
env = make('LatteArt-v0', seed=0, engine_lib=lib, loss=False, quality=2, particle_density=4e6, n_pool=60000, max_substeps_local=50, ckpt_dest='cpu')
te = env.taichi_env
pol = env.demo_policy()
te.apply_agent_action_p(pol.get_actions_p())
for i in range(3):
    te.step(pol.get_action_v(i))
ws = te.simulator.engine.get_work_stats(0)
print({k: ws[k] for k in ('n_items', 'n_multi_item_workgroups', 'n_leftover_items', 'n_single_item_blocks', 'n_quad_items', 'n_quad_units', 'n_scatter_units', 'n_gather_units', 'packed', 'n_active_blocks')}, ws['items_by_size'])

"""MPMSimulator -- the host-side surface of fluidlab/fluidengine/simulators/mpm_simulator.py (`mpm:NNN`)
over the MI355X engine.

What was 29 @ti.kernel's and the Taichi runtime in the reference is one C-ABI library here
(fluidlab_amd/csrc, include/fluidengine.h) driven through ctypes.  The time indexing, action buffering,
chunked checkpoint/recompute protocol and state I/O keep the reference's semantics so TaichiEnv, Agent,
Loss, Solver and Recorder work against it unchanged.

MI355X-first differences:
  * `max_substeps_local=None` keeps the whole trajectory's particle frames resident in HBM (288 GB), so the
    backward pass needs no checkpoint reload and no second forward (mpm:856-912 is then never triggered);
  * the grid is not stored per frame: the backward pass recomputes P2G + grid_op of frame f;
  * particle adjoints are a ring of two frames inside the engine."""
import os
import pickle as pkl
import uuid

import numpy as np

from fluidlab_amd import _capi
from fluidlab_amd.configs.macros import DTYPE_NP, LAMDA, MAT_CLASS, MU
from fluidlab_amd.fluidengine.boundaries import create_boundary


class _Field:
    """Tiny stand-in for the Taichi fields other components peek into (`.to_numpy()`)."""

    def __init__(self, getter):
        self._getter = getter

    def to_numpy(self):
        return self._getter()


class _Namespace:
    pass


def _is_cuda_tensor(a):
    return hasattr(a, 'data_ptr') and getattr(a, 'is_cuda', False)


class MPMSimulator:
    def __init__(self, dim, quality, gravity, horizon, max_substeps_local, max_substeps_global, ckpt_dest,
                 engine_lib=None, device=0, dt=None):
        assert dim == 3, 'the MI355X engine is 3-D'
        self.dim = dim
        self.ckpt_dest = ckpt_dest
        self.sim_id = str(uuid.uuid4())
        self.gravity = tuple(float(g) for g in gravity)

        self.n_grid = int(64 * quality)                     # mpm:21
        self.dx = 1 / self.n_grid
        self.inv_dx = float(self.n_grid)
        # mpm:24 fixes dt = 2e-4 for every grid.  `dt` (not in the reference) is for grids finer than the 64^3 it runs: the stiff materials are
        # beyond their Courant limit there (ICECREAM at 256^3: 2.4) and leave the grid within a few hundred substeps on any implementation
        self.dt = 2e-4 if dt is None else float(dt)
        self.p_vol = (self.dx * 0.5) ** 2                   # mpm:25 (sic: squared in 3-D)
        self.res = (self.n_grid,) * self.dim
        self.max_substeps_global = max_substeps_global
        self.horizon = horizon
        self.n_substeps = int(round(2e-3 / self.dt))        # mpm:30
        if max_substeps_local is None:                      # whole trajectory resident in HBM
            max_substeps_local = self.n_substeps * (horizon + 1)
        self.max_substeps_local = max_substeps_local
        self.max_steps_local = int(self.max_substeps_local / self.n_substeps)

        assert self.n_substeps * self.horizon < self.max_substeps_global      # mpm:33
        assert self.max_substeps_local % self.n_substeps == 0                 # mpm:34

        self.boundary = None
        self.has_particles = False
        self.engine = None
        self._elib = engine_lib          # None -> the HIP library (raises when it is not built: no fallback)
        self._device = device

    def setup_boundary(self, **kwargs):
        self.boundary = create_boundary(**kwargs)

    # ------------------------------------------------------------------ build
    def engine_library(self):
        """the library this simulator runs on (loads the HIP library on first use; raises when it is not built)"""
        if self._elib is None:
            self._elib = _capi.load_hip()
        return self._elib

    def build(self, agent, smoke_field, statics, particles):
        if self.boundary is None:
            self.boundary = create_boundary()                # mpm:44-45
        self.n_statics = len(statics) if statics is not None else 0
        self.statics = statics

        if self._elib is None:
            self._elib = _capi.load_hip()
        elib = self._elib
        self.dtype = elib.dtype

        if particles is not None:
            self.has_particles = True
            self.n_particles = len(particles['x'])
        else:
            self.has_particles = False
            self.n_particles = 0

        self.engine = _capi.Engine(
            elib, n_grid=self.n_grid, n_particles=self.n_particles, max_substeps_local=self.max_substeps_local,
            n_substeps=self.n_substeps, max_action_steps=self.horizon, dt=self.dt, p_vol=self.p_vol,
            gravity=self.gravity, boundary=self.boundary.to_abi(elib), device=self._device)

        if self.has_particles:
            self.setup_ckpt_vars()
            self.init_particles_and_bodies(particles)
            # shims for the attribute peeks of Agent / Loss / Recorder (agent_injector.py:21, recorder.py:59)
            self.particles = _Namespace()
            self.particles_i = _Namespace()
            self.particles_i.mat = _Field(lambda: self.engine.get_mat())
            self.particles_ng = _Namespace()
            self.particles_ng.used = _Field(lambda: self.get_used(0)[None, :])
        else:
            self.particles = None

        # static SDF colliders of grid_op (mpm:386-390): voxels + world->voxel map + friction go to the engine
        if statics is not None:
            for st in statics:
                if st.has_dynamics:
                    st.prepare(self.engine.elib, self._device)
                    self.engine.add_static(st.sdf_voxels_np.astype(self.dtype), st.T_mesh_to_voxels_np, friction=st.friction,
                                           softness=st.softness)

        self.agent = agent
        self.smoke_field = smoke_field
        self.cur_substep_global = 0
        self.disable_grad()                                  # mpm:71

    def setup_ckpt_vars(self):
        """mpm:119-134"""
        if self.ckpt_dest == 'disk':
            N = self.n_particles
            self.x_np = np.zeros((N, 3), self.dtype); self.v_np = np.zeros((N, 3), self.dtype)
            self.C_np = np.zeros((N, 3, 3), self.dtype); self.F_np = np.zeros((N, 3, 3), self.dtype)
            self.used_np = np.zeros((N,), np.int32)
        else:
            self.ckpt_ram = dict()
        self.actions_buffer = []
        self.ckpt_dir = os.path.join('/tmp', 'fluidlab', self.sim_id)
        os.makedirs(self.ckpt_dir, exist_ok=True)

    def init_particles_and_bodies(self, particles):
        """mpm:136-148, 177-201"""
        mat = particles['mat'].astype(np.int32)
        body_id = particles['body_id'].astype(np.int32)
        self.n_bodies = particles['bodies']['n']
        assert self.n_bodies == np.max(body_id) + 1
        self.engine.init_particles(
            particles['x'].astype(DTYPE_NP), particles['used'].astype(np.int32), mat,
            np.array([MAT_CLASS[m] for m in mat], np.int32), np.array([MU[m] for m in mat]),
            np.array([LAMDA[m] for m in mat]), particles['rho'], body_id)

    # ------------------------------------------------------------------ grads
    def reset_grad(self):
        self.engine.reset_grad()                             # mpm:203-205 (+ effectors, done engine-side)

    def enable_grad(self):
        self.grad_enabled = True
        self.cur_substep_global = 0
        self.last_move_f = None

    def disable_grad(self):
        self.grad_enabled = False
        self.cur_substep_global = 0
        self.last_move_f = None

    # ------------------------------------------------------------------ time indexing, mpm:225-252
    def f_global_to_f_local(self, f_global):
        return f_global % self.max_substeps_local

    def f_local_to_s_local(self, f_local):
        return f_local // self.n_substeps

    def f_global_to_s_local(self, f_global):
        return self.f_local_to_s_local(self.f_global_to_f_local(f_global))

    def f_global_to_s_global(self, f_global):
        return f_global // self.n_substeps

    @property
    def cur_substep_local(self):
        return self.f_global_to_f_local(self.cur_substep_global)

    @property
    def cur_step_local(self):
        return self.f_global_to_s_local(self.cur_substep_global)

    @property
    def cur_step_global(self):
        return self.f_global_to_s_global(self.cur_substep_global)

    # ------------------------------------------------------------------ the hot path
    def substep(self, f, is_none_action):
        self.engine.substep(f, self.cur_substep_global, not is_none_action)        # mpm:515-533

    def substep_grad(self, f, is_none_action):
        self.engine.substep_grad(f, self.cur_substep_global, not is_none_action)   # mpm:535-552

    def step(self, action=None):
        """mpm:721-732"""
        if self.grad_enabled and self.cur_substep_local == 0:
            self.actions_buffer = []
        self.step_(action)
        if self.grad_enabled:
            self.actions_buffer.append(action)
        if self.cur_substep_local == 0:
            self.memory_to_cache()

    def step_(self, action=None):
        """mpm:735-753: one ABI crossing for the n_substeps loop."""
        self.step_begin(action)
        self.engine.step(*self.step_args(action))
        self.step_end(action)

    # step_ in three pieces, so that B simulators can share the middle one (fe_step_batch: optimizer/batch.py)
    def step_args(self, action):
        return self.cur_substep_local, self.cur_substep_global, self.n_substeps, action is not None

    def step_begin(self, action):
        if action is not None:
            self.agent.set_action(s=self.cur_step_local, s_global=self.cur_step_global, n_substeps=self.n_substeps, action=action)
        if self.smoke_field is not None:                    # smoke simulates at step level, not substep (mpm:744-747)
            self.smoke_field.step(s=self.cur_step_local, f=self.cur_substep_local)

    def step_end(self, action):
        if action is not None:
            self.last_move_f = self.cur_substep_local + self.n_substeps - 1        # Effector.latest_pos (effector.py:146-152)
        self.cur_substep_global += self.n_substeps
        assert self.cur_substep_global <= self.max_substeps_global

    def step_grad(self, action=None):
        """mpm:755-775"""
        self.step_grad_begin()
        self.engine.step_grad(*self.step_args(action))
        self.step_grad_end(action)

    def step_grad_begin(self):
        if self.cur_substep_local == 0:
            self.memory_from_cache()
        self.cur_substep_global -= self.n_substeps

    def step_grad_end(self, action):
        if self.smoke_field is not None:                    # mpm:765-767
            self.smoke_field.step_grad(s=self.cur_step_local, f=self.cur_substep_local)
        if action is not None:
            self.agent.set_action_grad(s=self.cur_substep_local // self.n_substeps, s_global=self.cur_substep_global // self.n_substeps,
                                       n_substeps=self.n_substeps, action=action)

    # ------------------------------------------------------------------ chunked checkpointing, mpm:777-912
    def memory_to_cache(self):
        if self.grad_enabled:
            ckpt_start_step = self.cur_substep_global - self.max_substeps_local
            ckpt_name = f'{ckpt_start_step:06d}'
            ckpt = {}
            if self.has_particles:
                if self.ckpt_dest == 'disk':
                    self.readframe(0, self.x_np, self.v_np, self.C_np, self.F_np, self.used_np)
                    ckpt.update(x=self.x_np, v=self.v_np, C=self.C_np, F=self.F_np, used=self.used_np)
                elif self.ckpt_dest == 'gpu' and self.engine.elib.backend.startswith('hip'):
                    # mpm:805-829: the checkpoint stays in HBM as torch tensors on the engine's device
                    import torch
                    dev = torch.device('cuda', self._device)
                    N = self.n_particles
                    # (empty, not zeros: the engine fills every element, and get_frame_dev fences torch's stream before it does)
                    st = dict(x=torch.empty((N, 3), dtype=torch.float32, device=dev), v=torch.empty((N, 3), dtype=torch.float32, device=dev),
                              C=torch.empty((N, 3, 3), dtype=torch.float32, device=dev), F=torch.empty((N, 3, 3), dtype=torch.float32, device=dev),
                              used=torch.empty((N,), dtype=torch.int32, device=dev))
                    self.readframe(0, st['x'], st['v'], st['C'], st['F'], st['used'])
                    ckpt.update(st)
                else:
                    st = self._frame_arrays()
                    self.readframe(0, st['x'], st['v'], st['C'], st['F'], st['used'])
                    ckpt.update(st)
                ckpt['actions'] = list(self.actions_buffer)
            if self.smoke_field is not None:
                ckpt['smoke_field'] = self.smoke_field.get_ckpt(ckpt_name)            # mpm:794-795
            if self.agent is not None:
                ckpt['agent'] = self.agent.get_ckpt()
            if self.ckpt_dest == 'disk':
                ckpt_file = os.path.join(self.ckpt_dir, f'{ckpt_name}.pkl')
                if os.path.exists(ckpt_file):
                    os.remove(ckpt_file)
                with open(ckpt_file, 'wb') as fh:
                    pkl.dump(ckpt, fh)
            elif self.ckpt_dest in ['cpu', 'gpu']:
                self.ckpt_ram[ckpt_name] = ckpt
            else:
                assert False
        # restart from frame 0 in memory (mpm:844-852)
        if self.has_particles:
            self.copy_frame(self.max_substeps_local, 0)
        if self.smoke_field is not None:
            self.smoke_field.copy_frame(self.max_steps_local, 0)                       # mpm:848-849
        if self.agent is not None:
            self.agent.copy_frame(self.max_substeps_local, 0)

    def memory_from_cache(self):
        assert self.grad_enabled
        L = self.max_substeps_local
        if self.has_particles:
            self.copy_frame(0, L)
            self.copy_grad(0, L)
            self.reset_grad_till_frame(L)
        if self.smoke_field is not None:                                               # mpm:863-866
            self.smoke_field.copy_frame(0, self.max_steps_local)
            self.smoke_field.copy_grad(0, self.max_steps_local)
            self.smoke_field.reset_grad_till_frame(self.max_steps_local)
        if self.agent is not None:
            self.agent.copy_frame(0, L)
            self.agent.copy_grad(0, L)
            self.agent.reset_grad_till_frame(L)

        ckpt_start_step = self.cur_substep_global - L
        ckpt_name = f'{ckpt_start_step:06d}'
        if self.ckpt_dest == 'disk':
            ckpt_file = os.path.join(self.ckpt_dir, f'{ckpt_name}.pkl')
            assert os.path.exists(ckpt_file)
            with open(ckpt_file, 'rb') as fh:
                ckpt = pkl.load(fh)
        elif self.ckpt_dest in ['cpu', 'gpu']:
            ckpt = self.ckpt_ram[ckpt_name]
        else:
            assert False
        if self.has_particles:
            self.setframe(0, ckpt['x'], ckpt['v'], ckpt['C'], ckpt['F'], ckpt['used'])
        if self.smoke_field is not None:
            self.smoke_field.set_ckpt(ckpt=ckpt['smoke_field'])
        if self.agent is not None:
            self.agent.set_ckpt(ckpt['agent'])
        # forward pass over the chunk to refill frames 1..L (mpm:906-909)
        self.cur_substep_global = ckpt_start_step
        for action in ckpt['actions']:
            self.step_(action)

    # ------------------------------------------------------------------ io, mpm:555-719
    def _frame_arrays(self):
        N = self.n_particles
        return dict(x=np.zeros((N, 3), self.dtype), v=np.zeros((N, 3), self.dtype), C=np.zeros((N, 3, 3), self.dtype),
                    F=np.zeros((N, 3, 3), self.dtype), used=np.zeros((N,), np.int32))

    def readframe(self, f, x, v, C, F, used):
        """mpm:555-564.  Like the reference's, accepts NumPy arrays or torch tensors; tensors living on the engine's GPU are
        filled in place by the engine without crossing PCIe."""
        if _is_cuda_tensor(x):
            self.engine.get_frame_dev(f, x, v, C, F, used)
        else:
            self.engine.get_frame(f, x, v, C, F, used)

    def setframe(self, f, x, v, C, F, used):
        if _is_cuda_tensor(x):
            self.engine.set_frame_dev(f, x, v, C, F, used)
        else:
            self.engine.set_frame(f, x, v, C, F, used)

    def set_x(self, f, x):
        self.engine.set_frame(f, x=x)

    def set_used(self, f, used):
        self.engine.set_frame(f, used=used)

    def copy_frame(self, source, target):
        self.engine.copy_frame(source, target)

    def copy_grad(self, source, target):
        self.engine.copy_grad(source, target)

    def reset_grad_till_frame(self, f):
        self.engine.reset_grad_till_frame(f)

    def get_state(self):
        f = self.cur_substep_local
        state = {}
        if self.has_particles:
            state.update(self._frame_arrays())
            self.readframe(f, state['x'], state['v'], state['C'], state['F'], state['used'])
        if self.agent is not None:
            state['agent'] = self.agent.get_state(f)
        if self.smoke_field is not None:                    # mpm:628-629
            state['smoke_field'] = self.smoke_field.get_state(self.cur_step_local)
        return state

    def set_state(self, f_global, state):
        f = self.f_global_to_f_local(f_global)
        self.last_move_f = None                              # a restored state starts a new episode: no move has happened in it yet
        if self.has_particles:
            self.setframe(f, state['x'], state['v'], state['C'], state['F'], state['used'])
        if self.agent is not None:
            self.agent.set_state(f, state['agent'])
        if self.smoke_field is not None:                    # mpm:643-644
            self.smoke_field.set_state(f // self.n_substeps, state['smoke_field'])

    def get_x(self, f=None):
        f = self.cur_substep_local if f is None else f
        x = np.zeros((self.n_particles, self.dim), dtype=self.dtype)
        if self.has_particles:
            self.engine.get_frame(f, x=x)
        return x

    def get_used(self, f=None):
        f = self.cur_substep_local if f is None else f
        used = np.zeros((self.n_particles,), dtype=np.int32)
        if self.has_particles:
            self.engine.get_frame(f, used=used)
        return used

    def get_v(self, f):
        v = np.zeros((self.n_particles, self.dim), dtype=self.dtype)
        if self.has_particles:
            self.engine.get_frame(f, v=v)
        return v

    def get_state_RL(self):
        f = self.cur_substep_local
        state = {}
        if self.has_particles:
            state['x'] = np.zeros((self.n_particles, 3), self.dtype)
            state['v'] = np.zeros((self.n_particles, 3), self.dtype)
            state['used'] = np.zeros((self.n_particles,), np.int32)
            self.engine.get_frame(f, x=state['x'], v=state['v'], used=state['used'])
        if self.agent is not None:
            state['agent'] = self.agent.get_state(f)
        if self.smoke_field is not None:                    # mpm:694-695
            state['smoke_field'] = self.smoke_field.get_state(self.cur_step_local)
        return state

    def get_state_render(self, f):
        return dict(x=self.get_x(f).astype(np.float32), used=self.get_used(f))

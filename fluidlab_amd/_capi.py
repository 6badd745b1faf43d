"""ctypes binding of the FluidEngine C ABI (include/fluidengine.h).

`load_hip()` is the product entry: it loads the in-tree gfx950 library
`fluidlab_amd/csrc/libfluidengine_hip.so` and raises if it is missing -- there is
no CPU fallback.  `EngineLib(path)` binds any library exporting the same ABI; the
tests use that to drive the host logic with the oracle build (tests only).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, 'csrc', 'libfluidengine_hip.so')

FE_BOUNDARY_CUBE, FE_BOUNDARY_CYLINDER = 0, 1
FE_EFF_PLAIN, FE_EFF_INJECTOR, FE_EFF_AIRCON = 0, 1, 2


class FeEngineError(RuntimeError):
    pass


def _structs(real):
    class FeBoundary(C.Structure):
        _fields_ = [('type', C.c_int), ('lower', real * 3), ('upper', real * 3),
                    ('xz_center', real * 2), ('xz_radius', real), ('restitution', real),
                    ('lock_dims', C.c_int)]

    class FeConfig(C.Structure):
        _fields_ = [('struct_size', C.c_int), ('n_grid', C.c_int), ('n_particles', C.c_int),
                    ('max_substeps_local', C.c_int), ('n_substeps', C.c_int),
                    ('max_action_steps', C.c_int), ('dt', real), ('p_vol', real),
                    ('gravity', real * 3), ('boundary', FeBoundary), ('device', C.c_int)]

    class FeEffectorDesc(C.Structure):
        _fields_ = [('struct_size', C.c_int), ('type', C.c_int), ('action_dim', C.c_int),
                    ('action_scale_v', real * 8), ('action_scale_p', real * 8),
                    ('boundary', FeBoundary), ('flux', C.c_int), ('radius', real),
                    ('inject_v', real * 3), ('inject_p', real * 3),
                    ('locally_random', C.c_int), ('randomize_inject_v', C.c_int),
                    ('random_length', C.c_int)]

    class FeSdfDesc(C.Structure):
        _fields_ = [('struct_size', C.c_int), ('res', C.c_int), ('T_mesh_to_voxels', real * 16),
                    ('friction', real), ('softness', real)]

    class FeSmokeConfig(C.Structure):
        _fields_ = [('struct_size', C.c_int), ('res', C.c_int), ('solver_iters', C.c_int), ('q_dim', C.c_int),
                    ('max_steps_local', C.c_int), ('dt', real), ('decay', real), ('high_T', real), ('low_T', real),
                    ('lower_y', C.c_int), ('higher_y', C.c_int)]

    return FeBoundary, FeConfig, FeEffectorDesc, FeSdfDesc, FeSmokeConfig


class FeStats(C.Structure):
    _fields_ = [('n_used', C.c_longlong), ('n_cells_touched', C.c_longlong),
                ('n_blocks_active', C.c_longlong), ('n_slow_path', C.c_longlong),
                ('bytes_state', C.c_longlong)]


# every symbol include/fluidengine.h declares (tests assert the libraries export all of them)
ABI_SYMBOLS = [
    'fe_create', 'fe_destroy', 'fe_last_error', 'fe_backend', 'fe_real_size', 'fe_sync',
    'fe_set_option', 'fe_get_option', 'fe_init_particles', 'fe_substep', 'fe_substep_grad', 'fe_step',
    'fe_step_grad', 'fe_step_batch', 'fe_step_grad_batch', 'fe_get_frame', 'fe_set_frame', 'fe_get_frame_dev', 'fe_set_frame_dev', 'fe_copy_frame', 'fe_copy_grad',
    'fe_reset_grad', 'fe_reset_grad_till_frame', 'fe_get_grad', 'fe_add_grad', 'fe_get_mat',
    'fe_add_static', 'fe_eff_set_mesh', 'fe_add_effector', 'fe_eff_set_act_range', 'fe_eff_get_state', 'fe_eff_set_state',
    'fe_eff_get_vw', 'fe_eff_set_vw', 'fe_eff_get_sr', 'fe_eff_set_sr', 'fe_eff_set_action', 'fe_eff_set_action_grad',
    'fe_eff_apply_action_p', 'fe_eff_apply_action_p_grad', 'fe_eff_get_action_grad',
    'fe_agent_copy_frame', 'fe_agent_copy_grad', 'fe_agent_reset_grad_till_frame', 'fe_agent_set_collector', 'fe_mesh_sdf', 'fe_add_grad_dev', 'fe_loss_alloc', 'fe_loss_set_target',
    'fe_loss_clear', 'fe_loss_step', 'fe_loss_step_grad', 'fe_loss_get', 'fe_get_stats', 'fe_get_work_stats', 'fe_get_work_stats_n',
    'fe_smoke_create', 'fe_smoke_step', 'fe_smoke_step_grad', 'fe_smoke_get_frame', 'fe_smoke_set_frame', 'fe_smoke_get_grad',
    'fe_smoke_add_grad', 'fe_smoke_copy_frame', 'fe_smoke_copy_grad', 'fe_smoke_reset_grad', 'fe_smoke_reset_grad_till_frame',
    'fe_timer_start', 'fe_timer_stop_ms', 'fe_profile_enable', 'fe_profile_read',
]


class EngineLib:
    """One loaded shared library exporting the FluidEngine ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FeEngineError(f'FluidEngine library not found: {path}')
        self.path = path
        if os.path.basename(path).endswith('_hip.so'):
            # torch wheels bundle their own libamdhip64; whichever HIP runtime is mapped first serves the whole process,
            # and torch sees no GPU when /opt/rocm's was mapped first.  The host layer shares device memory with torch
            # (ckpt_dest='gpu', the action-gradient all-reduce), so let torch map its runtime before the engine binds it.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self.lib = C.CDLL(path)
        lib = self.lib
        lib.fe_real_size.restype = C.c_int
        lib.fe_backend.restype = C.c_char_p
        self.real_size = lib.fe_real_size()
        self.backend = lib.fe_backend().decode()
        self.real = C.c_float if self.real_size == 4 else C.c_double
        self.dtype = np.float32 if self.real_size == 4 else np.float64
        self.FeBoundary, self.FeConfig, self.FeEffectorDesc, self.FeSdfDesc, self.FeSmokeConfig = _structs(self.real)
        lib.fe_create.restype = C.c_void_p
        lib.fe_create.argtypes = [C.c_void_p]
        lib.fe_destroy.restype = None
        lib.fe_destroy.argtypes = [C.c_void_p]
        lib.fe_last_error.restype = C.c_char_p
        lib.fe_last_error.argtypes = [C.c_void_p]
        lib.fe_timer_stop_ms.restype = C.c_double
        lib.fe_timer_stop_ms.argtypes = [C.c_void_p]
        lib.fe_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        if hasattr(lib, 'fe_get_option'):                 # (A/B builds of earlier rounds, scripts/ab_phases.py `lib=`, do not have it)
            lib.fe_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]

    def mesh_sdf(self, verts, faces, points, device=0):
        """fe_mesh_sdf: signed distance of points[M,3] to the triangle mesh (verts[nv,3], faces[nf,3]); float32 out."""
        v = np.ascontiguousarray(verts, np.float32); f = np.ascontiguousarray(faces, np.int32); p = np.ascontiguousarray(points, np.float32)
        assert v.ndim == 2 and v.shape[1] == 3 and f.ndim == 2 and f.shape[1] == 3 and p.ndim == 2 and p.shape[1] == 3
        out = np.empty((len(p),), np.float32)
        rc = self.lib.fe_mesh_sdf(int(device), v.ctypes.data_as(C.c_void_p), len(v), f.ctypes.data_as(C.c_void_p), len(f),
                                  p.ctypes.data_as(C.c_void_p), C.c_longlong(len(p)), out.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise FeEngineError(self.lib.fe_last_error(None).decode())
        return out

    def missing_symbols(self):
        return [s for s in ABI_SYMBOLS if not hasattr(self.lib, s)]

    def make_boundary(self, type='cube', lower=(0.05, 0.05, 0.05), upper=(0.95, 0.95, 0.95),
                      y_range=(0.05, 0.95), xz_center=(0.5, 0.5), xz_radius=0.45,
                      restitution=0.0, lock_dims=()):
        """boundaries.py:136-141 create_boundary(), as the ABI struct."""
        b = self.FeBoundary()
        if type == 'cube':
            b.type = FE_BOUNDARY_CUBE
            lo = np.asarray(lower, dtype=self.dtype)
            up = np.asarray(upper, dtype=self.dtype)
            assert (up >= lo).all()
            b.lower[:] = [float(t) for t in lo]
            b.upper[:] = [float(t) for t in up]
        elif type == 'cylinder':
            b.type = FE_BOUNDARY_CYLINDER
            yr = np.asarray(y_range, dtype=self.dtype)
            b.lower[:] = [0.0, float(yr[0]), 0.0]
            b.upper[:] = [1.0, float(yr[1]), 1.0]
            c = np.asarray(xz_center, dtype=self.dtype)
            b.xz_center[:] = [float(c[0]), float(c[1])]
            b.xz_radius = float(xz_radius)
        else:
            raise AssertionError(f'unknown boundary type {type}')
        b.restitution = float(restitution)
        b.lock_dims = sum(1 << int(d) for d in lock_dims)
        return b


_hip_lib = None


def load_hip():
    """The product library.  Fails loudly when the HIP extension is not built."""
    global _hip_lib
    if _hip_lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise FeEngineError(
                f'{HIP_LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
        _hip_lib = EngineLib(HIP_LIB_PATH)
    return _hip_lib


class Engine:
    """Thin object wrapper over one FeEngine handle.  All arrays are numpy, C-contiguous,
    in the library's real dtype; integer arrays are int32."""

    def __init__(self, elib, *, n_grid, n_particles, max_substeps_local, n_substeps,
                 max_action_steps, dt, p_vol, gravity, boundary, device=0):
        self.elib = elib
        self.lib = elib.lib
        self.dtype = elib.dtype
        cfg = elib.FeConfig()
        cfg.struct_size = C.sizeof(elib.FeConfig)
        cfg.n_grid = int(n_grid)
        cfg.n_particles = int(n_particles)
        cfg.max_substeps_local = int(max_substeps_local)
        cfg.n_substeps = int(n_substeps)
        cfg.max_action_steps = int(max_action_steps)
        cfg.dt = float(dt)
        cfg.p_vol = float(p_vol)
        cfg.gravity[:] = [float(g) for g in gravity]
        cfg.boundary = boundary
        cfg.device = int(device)
        self.device = int(device)
        self.cfg = cfg
        self.N = int(n_particles)
        self.h = self.lib.fe_create(C.byref(cfg))
        if not self.h:
            raise FeEngineError('fe_create failed: ' + self.lib.fe_last_error(None).decode())
        self.h = C.c_void_p(self.h)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.fe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers
    def _ck(self, rc):
        if rc != 0:
            raise FeEngineError(self.lib.fe_last_error(self.h).decode())

    def _r(self, a, shape=None):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if shape is not None:
            assert a.shape == tuple(shape), f'expected shape {shape}, got {a.shape}'
        return a, a.ctypes.data_as(C.c_void_p)

    def _i(self, a, shape=None):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.int32)
        if shape is not None:
            assert a.shape == tuple(shape), f'expected shape {shape}, got {a.shape}'
        return a, a.ctypes.data_as(C.c_void_p)

    def _out(self, a, shape, dtype=None):
        """Validate a caller-provided output array (must be writable in place)."""
        dtype = dtype or self.dtype
        assert isinstance(a, np.ndarray) and a.dtype == dtype and a.flags['C_CONTIGUOUS'] and a.shape == tuple(shape)
        return a.ctypes.data_as(C.c_void_p)

    # ---- lifecycle
    def sync(self):
        self._ck(self.lib.fe_sync(self.h))

    def set_option(self, name, value):
        self._ck(self.lib.fe_set_option(self.h, name.encode(), float(value)))

    OPTION_NAMES = ('sort_interval', 'item_max', 'grid_store', 'p2g_grad_waves', 'g2p_grad_v', 'loose_max', 'xcd_map', 'write_through',
                    'wave_sort', 'lane_split', 'fold_reorder', 'compact_F', 'fuse_g2p', 'fuse_bwd', 'quad_min_units', 'pgg_quad_min_units', 'quad_max', 'quad_fit', 'pack_units', 'wgrid_cap', 'wgrid_cap_g2p', 'wgrid_cap_pgg', 'ggrid_cap', 'collide_type')

    def get_option(self, name):
        v = C.c_double(0.0)
        self._ck(self.lib.fe_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def get_options(self):
        """every tunable as the engine holds it now: defaults, fe_set_option calls and FE_* environment variables alike"""
        return {n: self.get_option(n) for n in self.OPTION_NAMES}

    def init_particles(self, x, used, mat, mat_cls, mu, lam, rho, body_id):
        N = self.N
        k0, x_ = self._r(x, (N, 3)); k1, u_ = self._i(used, (N,)); k2, m_ = self._i(mat, (N,))
        k3, c_ = self._i(mat_cls, (N,)); k4, mu_ = self._r(mu, (N,)); k5, la_ = self._r(lam, (N,))
        k6, rh_ = self._r(rho, (N,)); k7, b_ = self._i(body_id, (N,))
        self._ck(self.lib.fe_init_particles(self.h, x_, u_, m_, c_, mu_, la_, rh_, b_))

    # ---- hot path
    def substep(self, f, f_global, act):
        self._ck(self.lib.fe_substep(self.h, int(f), int(f_global), int(bool(act))))

    def substep_grad(self, f, f_global, act):
        self._ck(self.lib.fe_substep_grad(self.h, int(f), int(f_global), int(bool(act))))

    def step(self, f0, f_global0, n, act):
        self._ck(self.lib.fe_step(self.h, int(f0), int(f_global0), int(n), int(bool(act))))

    def step_grad(self, f0, f_global0, n, act):
        self._ck(self.lib.fe_step_grad(self.h, int(f0), int(f_global0), int(n), int(bool(act))))

    # ---- state I/O
    def get_frame(self, f, x=None, v=None, C_=None, F=None, used=None):
        """Fill the given arrays in place (like readframe, mpm:555-564); None = skip."""
        N = self.N
        px = self._out(x, (N, 3)) if x is not None else None
        pv = self._out(v, (N, 3)) if v is not None else None
        pC = self._out(C_, (N, 3, 3)) if C_ is not None else None
        pF = self._out(F, (N, 3, 3)) if F is not None else None
        pu = self._out(used, (N,), np.int32) if used is not None else None
        self._ck(self.lib.fe_get_frame(self.h, int(f), px, pv, pC, pF, pu))

    def set_frame(self, f, x=None, v=None, C_=None, F=None, used=None):
        N = self.N
        k0, px = self._r(x, (N, 3)); k1, pv = self._r(v, (N, 3)); k2, pC = self._r(C_, (N, 3, 3))
        k3, pF = self._r(F, (N, 3, 3)); k4, pu = self._i(used, (N,))
        self._ck(self.lib.fe_set_frame(self.h, int(f), px, pv, pC, pF, pu))

    @staticmethod
    def _handles(engines):
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        return arr

    @staticmethod
    def step_batch(engines, f0, f_global0, n, act):
        """fe_step_batch: the same n substeps for every engine of the list (lockstep environments, one launch per phase)"""
        e0 = engines[0]
        e0._ck(e0.lib.fe_step_batch(Engine._handles(engines), len(engines), int(f0), int(f_global0), int(n), int(bool(act))))

    @staticmethod
    def step_grad_batch(engines, f0, f_global0, n, act):
        e0 = engines[0]
        e0._ck(e0.lib.fe_step_grad_batch(Engine._handles(engines), len(engines), int(f0), int(f_global0), int(n), int(bool(act))))

    def copy_frame(self, src, dst):
        self._ck(self.lib.fe_copy_frame(self.h, int(src), int(dst)))

    def copy_grad(self, src, dst):
        self._ck(self.lib.fe_copy_grad(self.h, int(src), int(dst)))

    def reset_grad(self):
        self._ck(self.lib.fe_reset_grad(self.h))

    def reset_grad_till_frame(self, f):
        self._ck(self.lib.fe_reset_grad_till_frame(self.h, int(f)))

    def get_grad(self, f):
        N = self.N
        gx = np.zeros((N, 3), self.dtype); gv = np.zeros((N, 3), self.dtype)
        gC = np.zeros((N, 3, 3), self.dtype); gF = np.zeros((N, 3, 3), self.dtype)
        self._ck(self.lib.fe_get_grad(self.h, int(f), gx.ctypes.data_as(C.c_void_p), gv.ctypes.data_as(C.c_void_p),
                                      gC.ctypes.data_as(C.c_void_p), gF.ctypes.data_as(C.c_void_p)))
        return gx, gv, gC, gF

    # ---- state that stays on the device: torch tensors on the engine's GPU (the reference's readframe/setframe accept them)
    @staticmethod
    def _tptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _torch_fence(self, *tensors):
        """The engine's stream is non-blocking: nothing orders it with torch's streams.  Before the engine reads or overwrites a
        torch tensor, whatever torch has queued on it -- the fill of a fresh allocation, an earlier read of a staging tensor,
        a producer on a side stream -- has to be finished: every stream of the tensors' devices is drained (the engine calls
        themselves return after the engine's stream has drained)."""
        done = set()
        for t in tensors:
            if t is not None and t.device not in done:
                import torch
                torch.cuda.synchronize(t.device)
                done.add(t.device)

    def get_frame_dev(self, f, x=None, v=None, C_=None, F=None, used=None):
        self._torch_fence(x, v, C_, F, used)
        self._ck(self.lib.fe_get_frame_dev(self.h, int(f), self._tptr(x), self._tptr(v), self._tptr(C_), self._tptr(F), self._tptr(used)))

    def set_frame_dev(self, f, x=None, v=None, C_=None, F=None, used=None):
        self._torch_fence(x, v, C_, F, used)
        self._ck(self.lib.fe_set_frame_dev(self.h, int(f), self._tptr(x), self._tptr(v), self._tptr(C_), self._tptr(F), self._tptr(used)))

    def add_grad_dev(self, f, gx=None, gv=None, gC=None, gF=None):
        """fe_add_grad_dev: torch tensors on the engine's GPU (their device is synchronised here, whichever stream produced them)"""
        self._torch_fence(gx, gv, gC, gF)
        self._ck(self.lib.fe_add_grad_dev(self.h, int(f), self._tptr(gx), self._tptr(gv), self._tptr(gC), self._tptr(gF)))

    def add_grad(self, f, gx=None, gv=None, gC=None, gF=None):
        N = self.N
        k0, px = self._r(gx, (N, 3)); k1, pv = self._r(gv, (N, 3)); k2, pC = self._r(gC, (N, 3, 3)); k3, pF = self._r(gF, (N, 3, 3))
        self._ck(self.lib.fe_add_grad(self.h, int(f), px, pv, pC, pF))

    def get_mat(self):
        m = np.zeros((self.N,), np.int32)
        self._ck(self.lib.fe_get_mat(self.h, m.ctypes.data_as(C.c_void_p)))
        return m

    # ---- effectors
    def add_effector(self, *, type, action_dim, action_scale_v, action_scale_p, boundary, flux=0,
                     radius=0.0, inject_v=(0, 0, 0), inject_p=(0, 0, 0), locally_random=False,
                     randomize_inject_v=False, random_vector=None):
        d = self.elib.FeEffectorDesc()
        d.struct_size = C.sizeof(self.elib.FeEffectorDesc)
        d.type = int(type)
        d.action_dim = int(action_dim)
        sv = list(action_scale_v) + [1.0] * 8
        sp = list(action_scale_p) + [1.0] * 8
        d.action_scale_v[:] = [float(t) for t in sv[:8]]
        d.action_scale_p[:] = [float(t) for t in sp[:8]]
        d.boundary = boundary
        d.flux = int(flux)
        d.radius = float(radius)
        d.inject_v[:] = [float(t) for t in inject_v]
        d.inject_p[:] = [float(t) for t in inject_p]
        d.locally_random = int(bool(locally_random))
        d.randomize_inject_v = int(bool(randomize_inject_v))
        keep, prv = None, None
        if random_vector is not None:
            keep, prv = self._r(random_vector)
            assert keep.ndim == 3 and keep.shape[1] == flux and keep.shape[2] == 3
            d.random_length = keep.shape[0]
        e = self.lib.fe_add_effector(self.h, C.byref(d), prv)
        if e < 0:
            raise FeEngineError(self.lib.fe_last_error(self.h).decode())
        return e

    def add_static(self, voxels, T_mesh_to_voxels, friction=0.0, softness=0.0):
        """statics.add_static (statics.py:14): one SDF collider for grid_op; returns its index."""
        keep, pv = self._r(voxels)
        assert keep.ndim == 3 and keep.shape[0] == keep.shape[1] == keep.shape[2]
        d = self.elib.FeSdfDesc()
        d.struct_size = C.sizeof(self.elib.FeSdfDesc)
        d.res = int(keep.shape[0])
        d.T_mesh_to_voxels[:] = [float(t) for t in np.asarray(T_mesh_to_voxels, np.float64).reshape(16)]
        d.friction = float(friction)
        d.softness = float(softness)
        i = self.lib.fe_add_static(self.h, C.byref(d), pv)
        if i < 0:
            raise FeEngineError(self.lib.fe_last_error(self.h).decode())
        return i

    # ---- AirCon strength / radius
    def eff_get_sr(self, e, f):
        s_, r_ = self.elib.real(), self.elib.real()
        self._ck(self.lib.fe_eff_get_sr(self.h, int(e), int(f), C.byref(s_), C.byref(r_)))
        return float(s_.value), float(r_.value)

    def eff_set_sr(self, e, f, s, r):
        self._ck(self.lib.fe_eff_set_sr(self.h, int(e), int(f), self.elib.real(float(s)), self.elib.real(float(r))))

    # ---- smoke field (smoke_field.py)
    def smoke_create(self, res=128, dt=0.03, solver_iters=500, q_dim=3, decay=0.99, max_steps_local=None, high_T=1.0, low_T=0.0,
                     lower_y=60, higher_y=68):
        c = self.elib.FeSmokeConfig()
        c.struct_size = C.sizeof(self.elib.FeSmokeConfig)
        c.res, c.solver_iters, c.q_dim, c.max_steps_local = int(res), int(solver_iters), int(q_dim), int(max_steps_local)
        c.dt, c.decay, c.high_T, c.low_T = float(dt), float(decay), float(high_T), float(low_T)
        c.lower_y, c.higher_y = int(lower_y), int(higher_y)
        self._ck(self.lib.fe_smoke_create(self.h, C.byref(c)))
        self.smoke_res, self.smoke_q_dim = int(res), int(q_dim)

    def smoke_step(self, s, f):
        self._ck(self.lib.fe_smoke_step(self.h, int(s), int(f)))

    def smoke_step_grad(self, s, f):
        self._ck(self.lib.fe_smoke_step_grad(self.h, int(s), int(f)))

    def smoke_get_frame(self, s, fields=('v', 'q')):
        n, qd = self.smoke_res, self.smoke_q_dim
        shapes = dict(v=(n, n, n, 3), v_tmp=(n, n, n, 3), div=(n, n, n), p=(n, n, n), q=(n, n, n, qd))
        out = {k: np.zeros(shapes[k], self.dtype) for k in fields}
        ptr = [out[k].ctypes.data_as(C.c_void_p) if k in out else None for k in ('v', 'v_tmp', 'div', 'p', 'q')]
        self._ck(self.lib.fe_smoke_get_frame(self.h, int(s), *ptr))
        return out

    def smoke_set_frame(self, s, **arrays):
        keep = {k: np.ascontiguousarray(a, self.dtype) for k, a in arrays.items() if a is not None}
        ptr = [keep[k].ctypes.data_as(C.c_void_p) if k in keep else None for k in ('v', 'v_tmp', 'div', 'p', 'q')]
        self._ck(self.lib.fe_smoke_set_frame(self.h, int(s), *ptr))

    def smoke_get_grad(self, s):
        n, qd = self.smoke_res, self.smoke_q_dim
        gv, gq = np.zeros((n, n, n, 3), self.dtype), np.zeros((n, n, n, qd), self.dtype)
        self._ck(self.lib.fe_smoke_get_grad(self.h, int(s), gv.ctypes.data_as(C.c_void_p), gq.ctypes.data_as(C.c_void_p)))
        return gv, gq

    def smoke_add_grad(self, s, gv=None, gq=None):
        kv = None if gv is None else np.ascontiguousarray(gv, self.dtype)
        kq = None if gq is None else np.ascontiguousarray(gq, self.dtype)
        self._ck(self.lib.fe_smoke_add_grad(self.h, int(s), None if kv is None else kv.ctypes.data_as(C.c_void_p),
                                            None if kq is None else kq.ctypes.data_as(C.c_void_p)))

    def smoke_copy_frame(self, src, dst):
        self._ck(self.lib.fe_smoke_copy_frame(self.h, int(src), int(dst)))

    def smoke_copy_grad(self, src, dst):
        self._ck(self.lib.fe_smoke_copy_grad(self.h, int(src), int(dst)))

    def smoke_reset_grad(self):
        self._ck(self.lib.fe_smoke_reset_grad(self.h))

    def smoke_reset_grad_till_frame(self, s):
        self._ck(self.lib.fe_smoke_reset_grad_till_frame(self.h, int(s)))

    def eff_set_mesh(self, e, voxels, T_mesh_to_voxels, friction=0.0, softness=0.0):
        """Rigid.setup_mesh (rigid.py:19-24): effector e becomes a moving SDF collider."""
        keep, pv = self._r(voxels)
        assert keep.ndim == 3 and keep.shape[0] == keep.shape[1] == keep.shape[2]
        d = self.elib.FeSdfDesc()
        d.struct_size = C.sizeof(self.elib.FeSdfDesc)
        d.res = int(keep.shape[0])
        d.T_mesh_to_voxels[:] = [float(t) for t in np.asarray(T_mesh_to_voxels, np.float64).reshape(16)]
        d.friction = float(friction)
        d.softness = float(softness)
        self._ck(self.lib.fe_eff_set_mesh(self.h, int(e), C.byref(d), pv))

    def eff_set_act_range(self, e, act_range):
        k, p = self._i(act_range)
        self._ck(self.lib.fe_eff_set_act_range(self.h, int(e), p, int(k.shape[0])))

    def eff_get_state(self, e, f):
        s = np.zeros((8,), self.dtype)
        self._ck(self.lib.fe_eff_get_state(self.h, int(e), int(f), s.ctypes.data_as(C.c_void_p)))
        return s

    def eff_set_state(self, e, f, state8):
        k, p = self._r(state8, (8,))
        self._ck(self.lib.fe_eff_set_state(self.h, int(e), int(f), p))

    def eff_get_vw(self, e, f):
        v = np.zeros((3,), self.dtype); w = np.zeros((3,), self.dtype)
        self._ck(self.lib.fe_eff_get_vw(self.h, int(e), int(f), v.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)))
        return v, w

    def eff_set_vw(self, e, f, v, w):
        k0, pv = self._r(v, (3,)); k1, pw = self._r(w, (3,))
        self._ck(self.lib.fe_eff_set_vw(self.h, int(e), int(f), pv, pw))

    def eff_set_action(self, e, s, s_global, n_substeps, action):
        k, p = self._r(action)
        self._ck(self.lib.fe_eff_set_action(self.h, int(e), int(s), int(s_global), int(n_substeps), p))

    def eff_set_action_grad(self, e, s, s_global, n_substeps):
        self._ck(self.lib.fe_eff_set_action_grad(self.h, int(e), int(s), int(s_global), int(n_substeps)))

    def eff_apply_action_p(self, e, action_p):
        k, p = self._r(action_p)
        self._ck(self.lib.fe_eff_apply_action_p(self.h, int(e), p))

    def eff_apply_action_p_grad(self, e):
        self._ck(self.lib.fe_eff_apply_action_p_grad(self.h, int(e)))

    def eff_get_action_grad(self, e, s, n, action_dim):
        g = np.zeros((n + 1, action_dim), self.dtype)
        self._ck(self.lib.fe_eff_get_action_grad(self.h, int(e), int(s), int(n), g.ctypes.data_as(C.c_void_p)))
        return g

    def agent_copy_frame(self, src, dst):
        self._ck(self.lib.fe_agent_copy_frame(self.h, int(src), int(dst)))

    def agent_copy_grad(self, src, dst):
        self._ck(self.lib.fe_agent_copy_grad(self.h, int(src), int(dst)))

    def agent_reset_grad_till_frame(self, f):
        self._ck(self.lib.fe_agent_reset_grad_till_frame(self.h, int(f)))

    def agent_set_collector(self, boundary=None, mat=-1):
        """collector_act_kernel (agent_pouring.py:30-41 every material, agent_jetbot.py:33-43 one material)."""
        self._ck(self.lib.fe_agent_set_collector(self.h, C.byref(boundary) if boundary is not None else None, int(mat)))

    # ---- loss
    def loss_alloc(self, max_loss_steps):
        self._ck(self.lib.fe_loss_alloc(self.h, int(max_loss_steps)))

    def loss_set_target(self, s, x):
        k, p = self._r(x, (self.N, 3))
        self._ck(self.lib.fe_loss_set_target(self.h, int(s), p))

    def loss_clear(self):
        self._ck(self.lib.fe_loss_clear(self.h))

    def loss_step(self, s, f, matching_mat, weight):
        self._ck(self.lib.fe_loss_step(self.h, int(s), int(f), int(matching_mat), self.elib.real(weight)))

    def loss_step_grad(self, s, f, matching_mat, weight, step_loss_grad):
        self._ck(self.lib.fe_loss_step_grad(self.h, int(s), int(f), int(matching_mat), self.elib.real(weight),
                                            self.elib.real(step_loss_grad)))

    def loss_get(self, n):
        a = np.zeros((n,), self.dtype)
        self._ck(self.lib.fe_loss_get(self.h, a.ctypes.data_as(C.c_void_p), int(n)))
        return a

    # ---- measurement
    def get_stats(self, f):
        st = FeStats()
        self._ck(self.lib.fe_get_stats(self.h, int(f), C.byref(st)))
        return {k: getattr(st, k) for k, _ in FeStats._fields_}

    def get_work_stats(self, f):
        out = (C.c_longlong * 24)()
        if hasattr(self.lib, 'fe_get_work_stats_n'):         # (the counted form: an A/B library of an earlier round writes its own, shorter list)
            self._ck(self.lib.fe_get_work_stats_n(self.h, int(f), out, 24))
        else:
            self._ck(self.lib.fe_get_work_stats(self.h, int(f), out))
        keys = ('n_items', 'tail_start', 'n_active_blocks', 'n_multi_item_workgroups', 'n_single_item_blocks')
        d = {k: int(out[i]) for i, k in enumerate(keys)}
        d['items_by_size'] = {k: int(out[5 + i]) for i, k in enumerate(('1', '2-4', '5-8', '9-16', '17-32', '33-64', '65-128'))}
        d['n_occupied_blocks'] = int(out[12])
        d['n_loose_particles'] = int(out[13])            # particles of blocks without a work item (engine option loose_max)
        d['n_quad_items'] = int(out[14])                 # single-item blocks of <= quad_max particles: four to a workgroup, one wave each ...
        d['n_quad_units'] = int(out[15])                 # ... when the order's unit list holds quad units at all (engine option quad_min_units): how many
        d['n_scatter_units'], d['n_gather_units'] = int(out[16]), int(out[17])     # work units of the two unit lists: the workgroups of a scatter / gather launch that have something to do
        d['packed'] = bool(out[18])                      # the scatter list has no idle halves (engine options pack_units, quad_fit)
        d['n_leftover_items'] = int(out[19]) + int(out[20])
        d['n_split9_waves'], d['n_split3_waves'] = int(out[21]), int(out[22])      # waves of <= 7 / 8..21 particles: nine / three lanes per particle (engine option lane_split)
        return d

    def timer_start(self):
        self._ck(self.lib.fe_timer_start(self.h))

    def timer_stop_ms(self):
        ms = self.lib.fe_timer_stop_ms(self.h)
        if ms < 0:
            raise FeEngineError(self.lib.fe_last_error(self.h).decode())
        return ms

    def profile_enable(self, on):
        self._ck(self.lib.fe_profile_enable(self.h, int(bool(on))))

    def profile_read(self, cap=32):
        buf = C.create_string_buffer(4096)
        ms = (C.c_double * cap)()
        cnt = (C.c_longlong * cap)()
        n = self.lib.fe_profile_read(self.h, buf, 4096, ms, cnt, cap)
        names = buf.value.decode().split('\n') if n > 0 else []
        return {names[i]: (ms[i], cnt[i]) for i in range(min(n, len(names)))}

"""Build-time particle sampling (fluidlab/fluidengine/bodies/bodies.py).

Pure host NumPy and part of the input contract: every add_body() re-seeds the global RNG with 0 and
restores it (bodies.py:27-28,44), so scenes are reproducible and identical to the reference's.
Mesh bodies (bodies.py:187-210) reject the samples of a box against the voxelised mesh; the voxelisation is the engine
library's `fe_mesh_sdf` (fluidlab_amd/utils/mesh.py) in the place of trimesh's."""
import numpy as np
from scipy.spatial.transform import Rotation

from fluidlab_amd.configs.macros import COLOR, MAT_NAME, NOWHERE, RHO
from fluidlab_amd.utils import mesh as mesh_utils


class Bodies:
    def __init__(self, dim, particle_density, elib=None, device=0):
        self.dim = dim
        self.particle_density = particle_density
        self._elib = elib        # callable returning the engine library (mesh bodies only)
        self._device = device
        self._parts = []        # one dict per body

    def __len__(self):
        return len(self._parts)

    def n_for_volume(self, volume):
        return round(volume * self.particle_density)

    def n_for_length(self, length):
        return round(length * np.cbrt(self.particle_density))

    # ---- samplers ----------------------------------------------------------------------
    def _box(self, lower, upper, filling):
        size = upper - lower
        if filling == 'random':
            return np.random.uniform(low=lower, high=upper, size=(self.n_for_volume(np.prod(size)), self.dim))
        if filling == 'grid':
            axes = [np.linspace(lower[d], upper[d], self.n_for_length(size[d]) + 1) for d in range(3)]
            return np.stack(np.meshgrid(*axes, indexing='ij'), -1).reshape((-1, 3))
        raise NotImplementedError(f'Unsupported filling type: {filling}.')

    def _ring(self, r, cx, cz, y):
        n = max(self.n_for_length(2 * np.pi * r), 1)
        ang = np.linspace(0, np.pi * 2, n + 1)[:-1]
        return np.vstack([np.cos(ang) * r + cx, np.repeat(y, n), np.sin(ang) * r + cz])

    def _sample(self, type, filling, **kw):
        if type == 'nowhere':
            return np.tile(np.array(NOWHERE), (kw.pop('n_particles'), 1)), False, kw
        if type == 'cube':
            lower = np.array(kw.pop('lower'))
            size = kw.pop('size', None)
            upper = lower + np.array(size) if size is not None else np.array(kw.pop('upper'))
            kw.pop('upper', None)
            assert (upper >= lower).all()
            return self._box(lower, upper, 'grid' if filling == 'natural' else filling), True, kw
        if type == 'cylinder':
            center, height, radius = np.array(kw.pop('center')), kw.pop('height'), np.array(kw.pop('radius'))
            if filling == 'natural':
                layers = [self._ring(r, center[0], center[2], y)
                          for y in np.linspace(center[1] - height / 2, center[1] + height / 2, self.n_for_length(height) + 1)
                          for r in np.linspace(0, radius, self.n_for_length(radius) + 1)]
                return np.hstack(layers).T, True, kw
            lo = np.array([center[0] - radius, center[1] - height / 2.0, center[2] - radius])
            hi = np.array([center[0] + radius, center[1] + height / 2.0, center[2] + radius])
            pts = self._box(lo, hi, filling)
            return pts[np.linalg.norm(pts[:, [0, 2]] - center[[0, 2]], axis=1) <= radius], True, kw
        if type == 'ball':
            center, radius = np.array(kw.pop('center')), kw.pop('radius')
            if filling == 'natural':
                layers = []
                for rs in np.linspace(0, radius, self.n_for_length(radius) + 1):
                    for a in np.linspace(-np.pi / 2, np.pi / 2, self.n_for_length(rs * np.pi) + 1):
                        y = center[1] + np.sin(a) * rs
                        layers.append(self._ring(np.sqrt(max(rs ** 2 - (center[1] - y) ** 2, 0)), center[0], center[2], y))
                return np.hstack(layers).T, True, kw
            pts = self._box(center - radius, center + radius, filling)
            return pts[np.linalg.norm(pts - center, axis=1) <= radius], True, kw
        if type == 'mesh':
            # add_mesh, bodies.py:187-210: sample the box pos +- scale / 2, keep what falls into filled voxels of the normalised mesh
            assert filling != 'natural', 'natural filling not supported for body type: mesh.'
            file, voxelize_res = kw.pop('file'), kw.pop('voxelize_res', 128)
            pos, scale = np.array(kw.pop('pos', (0.5, 0.5, 0.5))), np.array(kw.pop('scale', (1.0, 1.0, 1.0)))
            if self._elib is None:
                raise NotImplementedError('mesh bodies need the engine library for the voxelisation: Bodies(elib=...)')
            voxels = mesh_utils.load_or_voxelize(file, voxelize_res, self._elib(), self._device)
            pts = self._box(pos - scale * 0.5, pos + scale * 0.5, filling)
            return pts[voxels.is_filled((pts - pos) / scale)], True, kw
        raise NotImplementedError(f'Unsupported body type: {type}.')

    def add_body(self, type, filling='random', **kwargs):
        assert filling in ['random', 'grid', 'natural'], f'Unsupported filling type: {filling}.'
        state = np.random.get_state()
        np.random.seed(0)                                   # bodies.py:28
        try:
            pts, used, rest = self._sample(type, filling, **kwargs)
            self._register(type, pts, used=used, **rest)
        finally:
            np.random.set_state(state)

    def _register(self, type, particles, material, color=None, used=False, euler=(0.0, 0.0, 0.0)):
        n = len(particles)
        R = Rotation.from_euler('zyx', np.array(euler)[::-1], degrees=True).as_matrix()
        com = particles.mean(0)
        particles = (R @ (particles - com).T).T + com        # bodies.py:75-77
        self._parts.append(dict(
            x=particles, mat=np.full(n, material), used=np.full(n, used), rho=np.full(n, RHO[material]),
            color=np.tile(color if color is not None else COLOR[material], [n, 1]), body_id=np.full(n, len(self._parts))))
        print(f'===>  {n:7d} particles of {MAT_NAME[material]:>8} {type:>8} added.')

    def get(self):
        """bodies.py:208-234"""
        if not self._parts:
            return None
        out = {k: np.concatenate([p[k] for p in self._parts]) for k in ('x', 'mat', 'used', 'color', 'rho', 'body_id')}
        out['bodies'] = {
            'n': len(self._parts),
            'n_particles': [len(p['x']) for p in self._parts],
            'particle_ids': [np.sort(np.where(out['body_id'] == i)[0]) for i in range(len(self._parts))],
        }
        return out

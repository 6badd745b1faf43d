"""Static SDF colliders -- fluidlab/fluidengine/meshes/{mesh,static,statics}.py.

The reference turns a triangle mesh into a signed-distance voxel grid with `mesh_to_sdf` + `trimesh` (utils/mesh.py:
63-87) and caches it as a pickle {'voxels': [res^3], 'T_mesh_to_voxels': 4x4}.  Here the conversion is the engine library's
`fe_mesh_sdf` (fluidlab_amd/utils/mesh.py), used whenever `file` exists in the asset tree ($FLUIDLAB_ASSETS/meshes/raw).  The
reference's mesh assets are not part of this repository; without them a collider's SDF comes from one of:
  * `sdf=<path>`   a pickle in the reference's format (produced by the reference's tooling elsewhere),
  * `sdf=<dict>`   the same two arrays in memory,
  * `sdf=<callable>` an analytic signed distance f(points[N,3]) -> [N] in the *mesh frame* (the normalised frame the
    reference's meshes live in, roughly [-0.5, 0.5]^3); it is sampled on exactly the query lattice of
    compute_sdf_data (voxels_radius 0.6, `sdf_res` points per axis), so T_mesh_to_voxels is the reference's.
Pose handling follows Mesh.init_transform (mesh.py:97-127): T_mesh_to_voxels @ inverse(T_init), T_init = trans * rot *
scale with euler given in degrees as (x, y, z) and composed 'zyx'.  Collision itself (sdf_, normal_, collide:
static.py:25-104) runs inside the engine's grid_op; this class only prepares its inputs."""
import os
import pickle as pkl

import numpy as np
from scipy.spatial.transform import Rotation

from fluidlab_amd.configs import macros
from fluidlab_amd.configs.macros import FRICTION
from fluidlab_amd.utils import mesh as mesh_utils
from fluidlab_amd.utils.misc import eval_str

VOXELS_RADIUS = 0.6          # utils/mesh.py:69


def sample_sdf(fn, res):
    """compute_sdf_data (utils/mesh.py:63-87) with an analytic distance instead of mesh_to_sdf."""
    g = np.linspace(-VOXELS_RADIUS, VOXELS_RADIUS, res)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    pts = np.stack([X, Y, Z], axis=-1).reshape((-1, 3))
    voxels = np.asarray(fn(pts), dtype=np.float64).reshape([res, res, res])
    T = np.eye(4)
    T[:3, :3] *= (res - 1) / (VOXELS_RADIUS * 2)
    T[:3, 3] = (res - 1) / 2
    return {'voxels': voxels, 'T_mesh_to_voxels': T}


# ---- analytic shapes in the mesh frame ---------------------------------------------------------------------------
def sdf_sphere(radius=0.5, center=(0.0, 0.0, 0.0)):
    c = np.asarray(center, np.float64)
    return lambda p: np.linalg.norm(p - c, axis=1) - radius


def sdf_box(half_extents=(0.5, 0.5, 0.5)):
    h = np.asarray(half_extents, np.float64)

    def fn(p):
        q = np.abs(p) - h
        return np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)
    return fn


def sdf_cylinder(radius=0.5, half_height=0.5):
    """axis along y"""
    def fn(p):
        d = np.stack([np.hypot(p[:, 0], p[:, 2]) - radius, np.abs(p[:, 1]) - half_height], axis=1)
        return np.minimum(d.max(axis=1), 0) + np.linalg.norm(np.maximum(d, 0), axis=1)
    return fn


def sdf_cup(radius=0.5, half_height=0.5, wall=0.08):
    """an open-top cup (the role of the reference's cup/bowl/tank meshes): outer cylinder minus an inner cavity that
    reaches through the top"""
    outer = sdf_cylinder(radius, half_height)
    inner = sdf_cylinder(radius - wall, half_height)

    def fn(p):
        q = p.copy(); q[:, 1] -= wall                         # the cavity starts `wall` above the bottom
        return np.maximum(outer(p), -inner(q))
    return fn


def sdf_cone_tip(r_bottom=0.10, r_top=0.28, z_bottom=-0.25, z_top=0.10, dent=0.08):
    """stand-in for the reference's cone_tip.obj (the collision mesh of the ice-cream cone): a solid frustum whose axis is
    the mesh z axis (the env's euler (-90, 0, 30) turns it upright), wide end up, with a shallow conical dent in the top"""
    def fn(p):
        r = np.hypot(p[:, 0], p[:, 1])
        z = p[:, 2]
        t = np.clip((z - z_bottom) / (z_top - z_bottom), 0.0, 1.0)
        side = (r - (r_bottom + t * (r_top - r_bottom))) * np.cos(np.arctan2(r_top - r_bottom, z_top - z_bottom))
        body = np.maximum(side, np.maximum(z_bottom - z, z - z_top))                     # frustum (approximate distance)
        dent_surface = z_top - dent * np.clip(1.0 - r / r_top, 0.0, 1.0)                  # top surface is lower near the axis
        return np.maximum(body, z - dent_surface)
    return fn


def sdf_stirrer(r=0.035, half_height=0.5):
    """stand-in for the reference's stirrer.obj: a vertical rod (axis y) in the mesh frame.  The real mesh, normalised, is 1 tall
    and 0.031-0.04 in radius (measured with fe_mesh_sdf on the reference's asset)."""
    def fn(p):
        d = np.stack([np.hypot(p[:, 0], p[:, 2]) - r, np.abs(p[:, 1]) - half_height], axis=1)
        return np.minimum(d.max(axis=1), 0) + np.linalg.norm(np.maximum(d, 0), axis=1)
    return fn


class Static:
    """Static mesh-based collider (static.py).  Only the collision inputs are kept; vertices/colours are renderer data."""

    def __init__(self, material, file=None, sdf=None, sdf_res=128, pos=(0.0, 0.0, 0.0), euler=(0.0, 0.0, 0.0),
                 scale=(1.0, 1.0, 1.0), softness=0, has_dynamics=False, file_vis=None):
        self.pos = np.asarray(eval_str(pos), np.float64)
        self.euler = np.asarray(eval_str(euler), np.float64)
        self.scale = np.asarray(eval_str(scale), np.float64) * np.ones(3)
        self.raw_file = file
        self.sdf_res = sdf_res
        # yaml configs name the material by its macro (`material: CONE`); the reference eval()s the string with the macros in
        # scope (mesh.py:33) -- here only identifiers defined in configs/macros.py are resolved
        self.material = getattr(macros, material) if isinstance(material, str) and material.isidentifier() and hasattr(macros, material) \
            else eval_str(material)
        self.has_dynamics = has_dynamics
        self.softness = softness
        self._sdf_arg = sdf
        self._prepared = False
        if not has_dynamics:
            self._prepared = True
            return                                             # visual only: Static.collide is the identity (static.py:83)
        self.friction = FRICTION[self.material]                # mesh.py:60
        if sdf is not None and not self._asset_present():
            self.prepare(None)                                 # explicit SDF, no mesh file to prefer: ready now

    def _asset_present(self):
        return self.raw_file is not None and (os.path.exists(mesh_utils.get_raw_mesh_path(self.raw_file))
                                              or os.path.exists(mesh_utils.get_processed_sdf_path(self.raw_file, self.sdf_res)))

    def prepare(self, elib, device=0):
        """Mesh.load_file + init_transform (mesh.py:40-63, 97-127), deferred until an engine library can compute the SDF.  The
        mesh file wins when the asset tree has it (the reference's behaviour); the `sdf=` argument is the stand-in otherwise."""
        if self._prepared:
            return
        sdf = self._sdf_arg
        if self._asset_present():
            if elib is None:
                return                                          # wait for build(): the distance transform runs in the engine library
            sdf_data = mesh_utils.load_or_compute_sdf(self.raw_file, self.sdf_res, elib, device)
        elif sdf is None:
            raise NotImplementedError(
                f'collision mesh {self.raw_file!r} is not in the asset tree ({mesh_utils.get_mesh_dir("raw")}; the reference\'s '
                f'assets are not part of this repository) and no sdf=<pickle path | dict | analytic callable> stand-in was given')
        elif callable(sdf):
            sdf_data = sample_sdf(sdf, self.sdf_res)
        elif isinstance(sdf, dict):
            sdf_data = sdf
        else:
            with open(sdf, 'rb') as fh:
                sdf_data = pkl.load(fh)
        self.sdf_voxels_np = np.asarray(sdf_data['voxels'], np.float64)
        self.sdf_voxels_res = self.sdf_voxels_np.shape[0]
        # init_transform, mesh.py:97-103,121: scale, then rotate, then translate
        rot = Rotation.from_euler('zyx', self.euler[::-1], degrees=True).as_matrix()
        T_init = np.eye(4)
        T_init[:3, :3] = rot @ np.diag(self.scale)
        T_init[:3, 3] = self.pos
        self.T_mesh_to_voxels_np = np.asarray(sdf_data['T_mesh_to_voxels'], np.float64) @ np.linalg.inv(T_init)
        self._prepared = True

    def sdf(self, pos_world):
        """host mirror of Static.sdf (static.py:26-49) for tests and scene building"""
        T = self.T_mesh_to_voxels_np
        pv = np.atleast_2d(pos_world) @ T[:3, :3].T + T[:3, 3]
        res = self.sdf_voxels_res
        base = np.floor(pv).astype(int)
        outside = ((base >= res - 1) | (base < 0)).any(1)
        b = np.clip(base, 0, res - 2)
        sd = np.zeros(len(pv))
        for i in range(2):
            for j in range(2):
                for k in range(2):
                    vp = b + [i, j, k]
                    sd += np.prod(1 - np.abs(pv - vp), axis=1) * self.sdf_voxels_np[vp[:, 0], vp[:, 1], vp[:, 2]]
        return np.where(outside, 1.0, sd)


class Dynamic(Static):
    """Dynamic mesh of a Rigid effector (dynamic.py): the same SDF + pose inputs as a Static, with has_dynamics always on;
    the mesh frame is the effector's frame (the pose at frame f is applied by the engine, dynamic.py:29-36)."""

    def __init__(self, container, **kwargs):
        self.container = container
        kwargs['has_dynamics'] = True
        super().__init__(**kwargs)


class Statics:
    """statics.py"""

    def __init__(self):
        self.statics = []

    def add_static(self, **kwargs):
        self.statics.append(Static(**kwargs))

    def __getitem__(self, index):
        return self.statics[index]

    def __len__(self):
        return len(self.statics)

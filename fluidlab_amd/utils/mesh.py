"""Triangle meshes on the host (fluidlab/utils/mesh.py): asset paths, OBJ loading, normalisation, and the two conversions the
engine's inputs need -- mesh -> SDF voxels for colliders (compute_sdf_data, :63-87) and mesh -> occupancy for mesh-filled
particle bodies (voxelize_mesh, :89-96; bodies.py:187-210).

The reference delegates to trimesh and mesh_to_sdf (virtual scans; "this might take minutes", mesh.py:88).  Neither exists
here; the distance itself comes from the engine library's `fe_mesh_sdf` (exact point-triangle distance, winding-number sign:
fluidlab_amd/csrc/fe_mesh.h), everything around it is a few lines of numpy.  Cached results use the reference's file names
and pickle layout ({'voxels', 'T_mesh_to_voxels'}), so processed/*.sdf files are interchangeable.

Meshes are looked up under $FLUIDLAB_ASSETS/meshes or <package>/assets/meshes (raw/, processed/, voxelized/); the reference's
assets are not part of this repository."""
import os
import pickle as pkl

import numpy as np

from .misc import get_src_dir

VOXELS_RADIUS = 0.6          # utils/mesh.py:69


def get_mesh_dir(kind):
    root = os.environ.get('FLUIDLAB_ASSETS') or os.path.join(get_src_dir(), 'assets')
    return os.path.join(root, 'meshes', kind)


def get_raw_mesh_path(file):
    assert file.endswith('.obj')
    return os.path.join(get_mesh_dir('raw'), file)


def get_processed_sdf_path(file, sdf_res):
    assert file.endswith('.obj')
    return os.path.join(get_mesh_dir('processed'), f"{file.replace('.obj', '')}-{sdf_res}.sdf")


def get_voxelized_mesh_path(file, voxelize_res):
    assert file.endswith('.obj')
    return os.path.join(get_mesh_dir('voxelized'), f"{file.replace('.obj', '')}-{voxelize_res}.vox")


class TriMesh:
    """vertices [nv, 3] float64, faces [nf, 3] int32"""

    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, np.float64).reshape(-1, 3)
        self.faces = np.asarray(faces, np.int32).reshape(-1, 3)

    def copy(self):
        return TriMesh(self.vertices.copy(), self.faces.copy())


def load_mesh(file):
    """Wavefront OBJ: `v x y z` and `f a b c ...` records (a/b/c vertex/texture/normal triples, negative = relative indices,
    polygons fan-triangulated); everything else (normals, texture coordinates, materials, groups) is renderer data."""
    verts, faces = [], []
    with open(file) as fh:
        for line in fh:
            if line.startswith('v '):
                verts.append([float(t) for t in line.split()[1:4]])
            elif line.startswith('f '):
                idx = []
                for tok in line.split()[1:]:
                    k = int(tok.split('/')[0])
                    idx.append(k - 1 if k > 0 else len(verts) + k)
                for j in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[j], idx[j + 1]])
    if not verts or not faces:
        raise ValueError(f'{file}: no triangles found')
    mesh = TriMesh(verts, faces)
    if mesh.faces.min() < 0 or mesh.faces.max() >= len(mesh.vertices):
        raise ValueError(f'{file}: face index out of range')
    return mesh


def save_mesh(file, mesh):
    with open(file, 'w') as fh:
        for v in mesh.vertices:
            fh.write(f'v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n')
        for f in mesh.faces:
            fh.write(f'f {f[0] + 1} {f[1] + 1} {f[2] + 1}\n')


def normalize_mesh(mesh, mesh_actual=None):
    """utils/mesh.py:33-46: centre the bounding box of mesh_actual and scale its longest edge to 1 -> [-0.5, 0.5]."""
    if mesh_actual is None:
        mesh_actual = mesh
    lo, hi = mesh_actual.vertices.min(0), mesh_actual.vertices.max(0)
    out = mesh.copy()
    out.vertices = (out.vertices - (hi + lo) / 2) / (hi - lo).max()
    return out


def sdf_lattice(res):
    """the query points of compute_sdf_data (:66-73) and the matching T_mesh_to_voxels (:78-80)"""
    g = np.linspace(-VOXELS_RADIUS, VOXELS_RADIUS, res)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    T = np.eye(4)
    T[:3, :3] *= (res - 1) / (VOXELS_RADIUS * 2)
    T[:3, 3] = (res - 1) / 2
    return np.stack([X, Y, Z], axis=-1).reshape((-1, 3)), T


def compute_sdf_data(mesh, res, elib, device=0):
    """utils/mesh.py:63-87 with the engine's exact distance in the place of mesh_to_sdf's scan-based one"""
    pts, T = sdf_lattice(res)
    voxels = elib.mesh_sdf(mesh.vertices, mesh.faces, pts, device=device).astype(np.float64).reshape([res, res, res])
    return {'voxels': voxels, 'T_mesh_to_voxels': T}


def load_or_compute_sdf(file, res, elib, device=0):
    """Mesh.process_mesh, mesh.py:84-95: processed/<name>-<res>.sdf is a cache of compute_sdf_data(normalize(raw))"""
    path = get_processed_sdf_path(file, res)
    if os.path.exists(path):
        with open(path, 'rb') as fh:
            return pkl.load(fh)
    raw = get_raw_mesh_path(file)
    print(f'===> Computing sdf for {raw}.')
    data = compute_sdf_data(normalize_mesh(load_mesh(raw)), res, elib, device)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'wb') as fh:
            pkl.dump(data, fh)
        print(f'===> sdf saved as {path}.')
    except OSError:
        pass                                           # read-only asset tree: recompute next time
    return data


FILL_REACH = 0.7          # a voxel counts as touched by the surface when its centre is within this many pitches of it


class FilledVoxels:
    """What bodies.py:199-209 needs of trimesh's VoxelGrid: is_filled(points) on the normalised mesh.  trimesh marks the voxels
    of pitch 1/res that the surface passes through and fills the enclosed ones (mesh.voxelized(pitch).fill(), :95); voxel
    centres sit at integer multiples of the pitch and a point belongs to the voxel whose centre is nearest.  Here the occupancy
    of a voxel is decided at its centre: inside the mesh, or within FILL_REACH pitches of the surface (a cube of edge 1 is
    crossed by a plane at most sqrt(3)/2 from its centre).  Against the grid the reference ships for duck.obj
    (assets/meshes/voxelized/duck-128.vox, 825k voxels) this agrees on 99.6 % of the voxels; every difference is a surface
    voxel whose centre is 0.5-0.85 pitches from the surface (tests/test_mesh.py::test_voxelisation_against_reference_grid)."""

    def __init__(self, occupancy, res):
        self.occupancy = np.asarray(occupancy, bool)
        self.res = res
        self.n = self.occupancy.shape[0]
        self.half = self.n // 2                        # index of the voxel centred on the mesh frame's origin

    def centres(self):
        c = (np.arange(self.n) - self.half) / self.res
        X, Y, Z = np.meshgrid(c, c, c, indexing='ij')
        return np.stack([X, Y, Z], axis=-1).reshape((-1, 3))

    def is_filled(self, points):
        idx = np.round(np.asarray(points, np.float64) * self.res).astype(int) + self.half
        ok = ((idx >= 0) & (idx < self.n)).all(1)
        out = np.zeros(len(idx), bool)
        i = idx[ok]
        out[ok] = self.occupancy[i[:, 0], i[:, 1], i[:, 2]]
        return out


def voxelize_mesh(mesh, res, elib, device=0):
    """voxelize_mesh (:89-96) for an already normalised mesh: occupancy of the voxels centred at k / res, |k| <= res / 2 + 1"""
    n = 2 * (res // 2 + 1) + 1
    vox = FilledVoxels(np.zeros((n, n, n), bool), res)
    d = elib.mesh_sdf(mesh.vertices, mesh.faces, vox.centres(), device=device).reshape([n, n, n])
    vox.occupancy = d <= FILL_REACH / res
    return vox


def load_or_voxelize(file, res, elib, device=0):
    """bodies.py:190-199: voxelized/<name>-<res>.vox caches the filled voxel grid of the normalised raw mesh"""
    path = get_voxelized_mesh_path(file, res)
    if os.path.exists(path):
        with open(path, 'rb') as fh:
            obj = pkl.load(fh)
        if isinstance(obj, dict) and 'occupancy' in obj:
            return FilledVoxels(obj['occupancy'], obj['res'])
        return obj                                      # a trimesh VoxelGrid pickled by the reference (needs trimesh to load)
    raw = get_raw_mesh_path(file)
    print(f'===> Voxelizing mesh {raw}.')
    vox = voxelize_mesh(normalize_mesh(load_mesh(raw)), res, elib, device)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'wb') as fh:
            pkl.dump({'occupancy': vox.occupancy, 'res': vox.res}, fh)
    except OSError:
        pass
    return vox


# ---- procedural meshes (tests, stand-ins) ------------------------------------------------------------------------------------
def icosphere(subdivisions=2, radius=0.5):
    t = (1.0 + 5 ** 0.5) / 2
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return TriMesh(np.array(v) * radius, f)


def box_mesh(half_extents=(0.5, 0.5, 0.5)):
    h = np.asarray(half_extents, np.float64)
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64) * h
    q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = [t for a, b, c, d in q for t in ([a, b, c], [a, c, d])]
    return TriMesh(v, f)

"""Seeded scenes shared by the oracle tests, the GPU parity tests, smoke() and the golden
fixtures.  Every scene is a plain dict of float32-representable numpy arrays, so the fp64
oracle and the fp32 HIP engine see bit-identical inputs."""
import numpy as np

from fluidlab_amd._capi import Engine, FE_EFF_INJECTOR, FE_EFF_PLAIN  # noqa: F401
from fluidlab_amd.scenes import (WATER, MILK, COFFEE, ELASTIC, ICECREAM, RIGID, RIGID_HEAVY, MILK_VIS, MAT_LIQUID,  # noqa: F401
                                 MAT_PLASTO_ELASTIC, MAT_ELASTIC, MAT_RIGID, MATERIALS, f32, water_block, make_engine, get_state)


def mixed_materials(n_grid=16, n_particles=1500, seed=1):
    """All constitutive branches in one block, moving, with non-trivial C and F."""
    rng = np.random.RandomState(seed)
    N = n_particles
    mats = np.array([WATER, MILK_VIS, ELASTIC, ICECREAM], np.int32)[rng.randint(0, 4, N)]
    sc = dict(
        n_grid=n_grid, N=N, dt=2e-4, gravity=(0.0, -10.0, 0.0), n_substeps=10,
        boundary=dict(type='cube', lower=(0.2, 0.2, 0.2), upper=(0.8, 0.8, 0.8)),
        x=f32(rng.uniform(0.22, 0.6, (N, 3))), used=(rng.uniform(size=N) > 0.1).astype(np.int32), mat=mats,
    )
    sc['v'] = f32(rng.normal(0, 0.5, (N, 3)))
    sc['C'] = f32(rng.normal(0, 2.0, (N, 3, 3)))
    Fn = np.where((mats == ICECREAM)[:, None, None], 0.002, 0.03)
    sc['F'] = f32(np.eye(3)[None] + rng.normal(0, 1.0, (N, 3, 3)) * Fn)
    return sc


def rigid_in_water(n_grid=16, n_water=1200, n_rigid=(150, 90), seed=5):
    """Two MAT_RIGID shape-matching bodies (mpm:449-505) dropped into a water block, tumbling: one RIGID box, one
    RIGID_HEAVY box with a few unused particles (the reference divides the COM by the body's total count, mpm:201,461)."""
    rng = np.random.RandomState(seed)
    xs = [rng.uniform(0.3, 0.7, (n_water, 3)) * [1, 0.5, 1] + [0, 0.1, 0]]
    mats, bids, vs = [np.full(n_water, WATER)], [np.zeros(n_water)], [np.zeros((n_water, 3))]
    centres = [(0.42, 0.55, 0.45), (0.6, 0.6, 0.58)]
    for b, (cnt, c, m) in enumerate(zip(n_rigid, centres, (RIGID, RIGID_HEAVY))):
        rel = rng.uniform(-1, 1, (cnt, 3)) * [0.06, 0.035, 0.05]
        w = rng.normal(0, 6.0, 3)                                # angular velocity -> the body really rotates
        xs.append(np.asarray(c) + rel); vs.append(np.cross(w, rel) + rng.normal(0, 0.3, 3))
        mats.append(np.full(cnt, m)); bids.append(np.full(cnt, b + 1))
    x = np.concatenate(xs); N = len(x)
    used = np.ones(N, np.int32); used[-5:] = 0
    # F away from the identity: at F_tmp = I the singular values coincide and the reference's backward_svd
    # (1/clamp(s_j^2 - s_i^2), mpm:272-292) is not a derivative of anything
    Fp = f32(np.eye(3)[None] + rng.normal(0, 0.03, (N, 3, 3)))
    return dict(F=Fp, C=f32(rng.normal(0, 1.0, (N, 3, 3))), n_grid=n_grid, N=N, dt=2e-4, gravity=(0.0, -10.0, 0.0), n_substeps=10,
                boundary=dict(type='cube', lower=(0.1, 0.1, 0.1), upper=(0.9, 0.9, 0.9)),
                x=f32(x), v=f32(np.concatenate(vs)), used=used, mat=np.concatenate(mats).astype(np.int32),
                body_id=np.concatenate(bids).astype(np.int32))


def sphere_sdf(center, radius, res=24, lo=0.0, hi=1.0):
    """Voxelised signed distance of a sphere over the world box [lo, hi]^3, in the layout of the reference's .sdf
    pickles ({'voxels': [res^3], 'T_mesh_to_voxels': 4x4}, mesh.py:63-66): voxel index = (res - 1) * (x - lo) / (hi - lo)."""
    g = np.linspace(lo, hi, res)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    vox = np.sqrt((X - center[0]) ** 2 + (Y - center[1]) ** 2 + (Z - center[2]) ** 2) - radius
    sc = (res - 1) / (hi - lo)
    T = np.array([[sc, 0, 0, -lo * sc], [0, sc, 0, -lo * sc], [0, 0, sc, -lo * sc], [0, 0, 0, 1.0]])
    return f32(vox), T


def water_on_obstacles(n_grid=16, n_particles=1500, seed=7):
    """A water block falling sideways onto two static SDF colliders (static.py:82-103): a frictional sphere inside the
    block and a frictionless one it slides along -- nodes inside either surface take the contact branch."""
    sc = water_block(n_grid=n_grid, n_particles=n_particles, seed=seed, lo=0.3, hi=0.6, gravity=(3.0, -10.0, 1.0))
    rng = np.random.RandomState(seed)
    sc['v'] = f32(rng.normal(0, 0.4, (n_particles, 3)) + [0.5, -1.0, 0.2])
    v1, T1 = sphere_sdf((0.42, 0.30, 0.45), 0.12)
    v2, T2 = sphere_sdf((0.62, 0.40, 0.50), 0.10, res=20, lo=0.2, hi=0.9)
    sc['statics'] = [dict(voxels=v1, T=T1, friction=0.5), dict(voxels=v2, T=T2, friction=0.0)]
    return sc


def box_sdf_mesh(half=(0.12, 0.05, 0.08), res=28, radius=0.3):
    """SDF of a box centred at the origin of the *effector* frame, voxel box [-radius, radius]^3 (layout of
    compute_sdf_data, utils/mesh.py:63-87)."""
    g = np.linspace(-radius, radius, res)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    q = np.abs(np.stack([X, Y, Z], -1)) - np.asarray(half)
    vox = np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(axis=-1), 0)
    T = np.eye(4)
    T[:3, :3] *= (res - 1) / (2 * radius)
    T[:3, 3] = (res - 1) / 2
    return f32(vox), T


def offset_sphere_sdf_mesh(center=(0.05, 0.0, 0.03), r=0.1, res=28, radius=0.3):
    """SDF of a sphere that does not sit at the effector's origin (so rotating the effector moves it).  Unlike the box, its
    normal field has no medial-axis jumps inside the solid: fp32 and fp64 runs stay on the same contact branches."""
    g = np.linspace(-radius, radius, res)
    X, Y, Z = np.meshgrid(g, g, g, indexing='ij')
    vox = np.sqrt((X - center[0]) ** 2 + (Y - center[1]) ** 2 + (Z - center[2]) ** 2) - r
    T = np.eye(4)
    T[:3, :3] *= (res - 1) / (2 * radius)
    T[:3, 3] = (res - 1) / 2
    return f32(vox), T


def stirrer_mini(n_grid=16, n_particles=1500, seed=11, horizon=4, n_substeps=4, friction=0.5, softness=0.0, action_dim=6, shape='box'):
    """A Rigid effector (rigid.py, dynamic.py) -- a tilted box moving and rotating through a water block -- with a 6-dof
    action per step: particle-level agent.collide in g2p, collider velocity from the pose change, pose/quaternion adjoints."""
    rng = np.random.RandomState(seed)
    sc = water_block(n_grid=n_grid, n_particles=n_particles, seed=seed, lo=0.3, hi=0.62, gravity=(0.0, -10.0, 0.0))
    vox, T = box_sdf_mesh() if shape == 'box' else offset_sphere_sdf_mesh()
    L = horizon * n_substeps
    sc.update(n_substeps=n_substeps, horizon=horizon, max_substeps_local=L + n_substeps,
              rigid=dict(action_dim=action_dim, action_scale_v=(1, 1, 1, 1, 1, 1), action_scale_p=(1, 1, 1, 1, 1, 1),
                         boundary=dict(type='cube', lower=(0.1, 0.1, 0.1), upper=(0.9, 0.9, 0.9)),
                         voxels=vox, T=T, friction=friction, softness=softness,
                         init_state=[0.45, 0.47, 0.46, 0.9238795, 0.0, 0.3826834, 0.0]),       # 45 deg about y
              action_p=f32([0.45, 0.47, 0.46, 0, 0, 0][:action_dim]),
              actions=f32(np.concatenate([rng.uniform(-0.02, 0.02, (horizon, 3)), rng.uniform(-0.3, 0.3, (horizon, 3))], 1)[:, :action_dim]))
    sc['v'] = f32(rng.normal(0, 0.3, (n_particles, 3)))
    return sc


def pouring_mini(**kw):
    """AgentPouring in small (agent_pouring.py): one Rigid whose collider acts at the particles AND at the grid nodes
    (collide_type='both'), plus a collector that takes every particle leaving its box; the water drifts towards +z so
    particles keep crossing the collector's upper z face during the run."""
    sc = stirrer_mini(shape='sphere', **kw)
    sc['v'] = sc['v'] + f32([0.0, 0.0, 3.0])
    sc['collide_type'] = 3
    sc['collector'] = dict(boundary=dict(type='cube', lower=(0.0, 0.1, 0.0), upper=(1.0, 1.0, 0.58)), mat=-1)
    return sc


def run_rigid(elib, sc, cot, device=0, options=None, actions=None, action_p=None):
    """Forward over the horizon with per-step actions, cotangent `cot` on the final frame, backward, action gradient
    [(horizon + 1), action_dim] -- the Solver's pass (solver.py:23-59) for an AgentRigid scene, through the raw ABI."""
    eng = make_engine(elib, sc, device=device, options=options)
    r = sc['rigid']
    e = eng.add_effector(type=FE_EFF_PLAIN, action_dim=r['action_dim'], action_scale_v=r['action_scale_v'],
                         action_scale_p=r['action_scale_p'], boundary=elib.make_boundary(**r['boundary']))
    eng.eff_set_mesh(e, r['voxels'], r['T'], friction=r['friction'], softness=r['softness'])
    if 'collide_type' in sc:                                  # Agent.collide_type (agent.py:17-26): 1 particle, 2 grid, 3 both
        eng.set_option('collide_type', sc['collide_type'])
    if 'collector' in sc:                                     # AgentPouring / AgentJetBot collector
        eng.agent_set_collector(elib.make_boundary(**sc['collector']['boundary']), sc['collector'].get('mat', -1))
    st0 = eng.eff_get_state(e, 0)
    st0[:7] = r['init_state']
    eng.eff_set_state(e, 0, st0)
    H, ns = sc['horizon'], sc['n_substeps']
    actions = sc['actions'] if actions is None else actions
    eng.eff_apply_action_p(e, sc['action_p'] if action_p is None else action_p)
    for s in range(H):
        eng.eff_set_action(e, s, s, ns, actions[s])
        eng.step(s * ns, s * ns, ns, 1)
    final = get_state(eng, H * ns)
    eff_state = eng.eff_get_state(e, H * ns)
    used_hist = None
    if 'collector' in sc:                                    # used flags of every frame after the forward pass
        used_hist = np.stack([get_state(eng, f)['used'] for f in range(H * ns + 1)])
    eng.reset_grad()
    eng.add_grad(H * ns, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
    for s in reversed(range(H)):
        eng.step_grad(s * ns, s * ns, ns, 1)
        eng.eff_set_action_grad(e, s, s, ns)
    eng.eff_apply_action_p_grad(e)
    grad = eng.eff_get_action_grad(e, 0, H, r['action_dim'])
    gx0 = eng.get_grad(0)[0]
    eng.close()
    loss = float(sum((final[k].astype(np.float64) * cot[g]).sum() for k, g in zip('xvCF', ('gx', 'gv', 'gC', 'gF'))))
    return dict(final=final, action_grad=grad, eff_state=eff_state, loss=loss, gx0=gx0, used_hist=used_hist)


def latte_mini(n_grid=16, n_coffee=1200, n_pool=200, seed=2, horizon=6, n_substeps=4, flux=2):
    """A small LatteArt: coffee in a cylinder, a milk pool injected by an Injector effector
    (latteart_env.py:38-75, agent_latteart.yaml), squared-distance loss on the milk."""
    rng = np.random.RandomState(seed)
    c = np.array([0.5, 0.45, 0.5])
    pts = []
    while sum(len(p) for p in pts) < n_coffee:
        p = rng.uniform([0.5 - 0.3, 0.4, 0.5 - 0.3], [0.5 + 0.3, 0.5, 0.5 + 0.3], (4 * n_coffee, 3))
        pts.append(p[np.linalg.norm(p[:, [0, 2]] - c[[0, 2]], axis=1) <= 0.3])
    coffee = np.concatenate(pts)[:n_coffee]
    N = n_pool + n_coffee
    x = np.concatenate([np.tile([-100.0, -100.0, -100.0], (n_pool, 1)), coffee])      # NOWHERE pool first (latteart_env.py:54-66)
    L = horizon * n_substeps
    sc = dict(
        n_grid=n_grid, N=N, dt=2e-4, gravity=(0.0, -20.0, 0.0), n_substeps=n_substeps, horizon=horizon,
        max_substeps_local=L + n_substeps,
        boundary=dict(type='cylinder', xz_radius=0.32, xz_center=(0.5, 0.5), y_range=(0.38, 0.9)),
        x=f32(x), used=np.concatenate([np.zeros(n_pool, np.int32), np.ones(n_coffee, np.int32)]),
        mat=np.concatenate([np.full(n_pool, MILK, np.int32), np.full(n_coffee, COFFEE, np.int32)]),
        injector=dict(radius=0.02, flux=flux, inject_v=(0.0, -3.0, 0.0), inject_p=(0.0, 0.0, 0.0), action_dim=3,
                      action_scale_v=(1.0, 1.0, 1.0), action_scale_p=(1.0, 1.0, 1.0), locally_random=True,
                      boundary=dict(type='cylinder', xz_radius=0.32, xz_center=(0.5, 0.5), y_range=(0.62, 0.62)),
                      random_vector=f32(rng.uniform(size=(L + n_substeps, flux, 3)))),
        action_p=f32([0.42, 0.62, 0.5]),
        actions=f32(rng.uniform(-0.01, 0.01, (horizon, 3))),
        matching_mat=MILK,
    )
    sc['target'] = f32(rng.uniform(0.35, 0.65, (horizon, N, 3)))
    return sc


# ------------------------------------------------------------------------------------------
def run_forward(eng, n_sub, f0=0):
    for f in range(f0, f0 + n_sub):
        eng.substep(f, f, 0)
    return get_state(eng, f0 + n_sub)


def run_forward_backward(eng, n_sub, cot, ranged=False):
    """Forward n_sub substeps from frame 0, seed the cotangent `cot` (dict gx,gv,gC,gF) on the
    last frame, backward to frame 0.  Returns (final state, grads at frame 0).
    ranged: the reverse sweep as ONE fe_step_grad call (the HIP engine then writes the adjoint across a sort boundary in the next
    substep's order straight away, option `fold_reorder`) instead of one fe_substep_grad call per substep."""
    st = run_forward(eng, n_sub)
    eng.reset_grad()
    eng.add_grad(n_sub, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
    if ranged:
        eng.step_grad(0, 0, n_sub, 0)
    else:
        for f in reversed(range(n_sub)):
            eng.substep_grad(f, f, 0)
    gx, gv, gC, gF = eng.get_grad(0)
    return st, dict(gx=gx, gv=gv, gC=gC, gF=gF)


def random_cotangent(N, seed=5):
    rng = np.random.RandomState(seed)
    return dict(gx=f32(rng.normal(size=(N, 3))), gv=f32(rng.normal(size=(N, 3)) * 1e-2),
                gC=f32(rng.normal(size=(N, 3, 3)) * 1e-4), gF=f32(rng.normal(size=(N, 3, 3)) * 1e-2))


def jetbot_mini(**kw):
    """AgentJetBot's injector in small (agent_transporting.yaml): a 6-dof action turns the nozzle, so the injection point
    pos + R(quat) inject_p and the jet velocity R(quat) inject_v depend on the quaternion chain (injector.py:92-96)."""
    sc = latte_mini(**kw)
    rng = np.random.RandomState(5)
    H = sc['horizon']
    inj = sc['injector']
    inj.update(action_dim=6, action_scale_v=(1.0, 1.0, 1.0, 5.0, 5.0, 5.0), action_scale_p=(1.0,) * 6,
               inject_v=(-3.0, -1.0, 0.0), inject_p=(-0.07, 0.0, 0.02))
    sc['action_p'] = f32([0.5, 0.62, 0.5, 0.0, 0.0, 0.0])
    sc['actions'] = f32(np.concatenate([rng.uniform(-0.01, 0.01, (H, 3)), rng.uniform(-0.02, 0.02, (H, 3))], 1))
    return sc


def run_latte(elib, sc, device=0, options=None):
    """Full mini trajectory optimisation pass through the raw ABI, mirroring Solver.forward_backward
    (optimizer/solver.py:23-59): forward with loss, backward, action gradient."""
    eng = make_engine(elib, sc, device=device, options=options)
    inj = sc['injector']
    e = eng.add_effector(type=FE_EFF_INJECTOR, action_dim=inj['action_dim'], action_scale_v=inj['action_scale_v'],
                         action_scale_p=inj['action_scale_p'], boundary=elib.make_boundary(**inj['boundary']),
                         flux=inj['flux'], radius=inj['radius'], inject_v=inj['inject_v'], inject_p=inj['inject_p'],
                         locally_random=inj['locally_random'], random_vector=inj['random_vector'])
    eng.eff_set_act_range(e, np.where(sc['used'] == 0)[0].astype(np.int32))
    st0 = eng.eff_get_state(e, 0)
    st0[:7] = [0.5, 0.5, 0.5, 1.0, 0.0, 0.0, 0.0]
    eng.eff_set_state(e, 0, st0)
    H, ns = sc['horizon'], sc['n_substeps']
    eng.loss_alloc(H)
    for s in range(H):
        eng.loss_set_target(s, sc['target'][s])
    eng.loss_clear()
    eng.eff_apply_action_p(e, sc['action_p'])
    for s in range(H):
        eng.eff_set_action(e, s, s, ns, sc['actions'][s])
        eng.step(s * ns, s * ns, ns, 1)
        eng.loss_step(s, (s + 1) * ns, sc['matching_mat'], 1.0)
    step_loss = eng.loss_get(H)
    final = get_state(eng, H * ns)
    eng.reset_grad()
    for s in reversed(range(H)):
        eng.loss_step_grad(s, (s + 1) * ns, sc['matching_mat'], 1.0, 1.0)
        eng.step_grad(s * ns, s * ns, ns, 1)
        eng.eff_set_action_grad(e, s, s, ns)
    eng.eff_apply_action_p_grad(e)
    grad = eng.eff_get_action_grad(e, 0, H, inj['action_dim'])
    eff_state = eng.eff_get_state(e, H * ns)
    eng.close()
    return dict(step_loss=step_loss, final=final, action_grad=grad, eff_state=eff_state)


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))

"""CPU tests of the oracle (oracle/fe_oracle.cpp).

The reference has no tests or golden vectors for this path (parity unpinned), so
the oracle is pinned three ways here: against a second, independently written numpy restatement
(oracle/mpm_numpy.py), against the structural invariants of MLS-MPM, and -- for the hand-derived
adjoints the reference gets from Taichi autodiff -- against central finite differences."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import scenarios as S  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import mpm_numpy  # noqa: E402


def test_oracle_matches_numpy_restatement(oracle64):
    sc = S.mixed_materials()
    eng = S.make_engine(oracle64, sc)
    st = S.get_state(eng, 0)
    props = np.array([S.MATERIALS[int(m)] for m in sc['mat']])
    n = sc['n_grid']
    x, v, C, F = (st[k].astype(np.float64) for k in 'xvCF')
    for f in range(3):
        eng.substep(f, f, 0)
        x, v, C, F, _ = mpm_numpy.substep(x, v, C, F, sc['used'], props[:, 0], props[:, 1], (0.5 / n) ** 2 * props[:, 2],
                                          props[:, 3].astype(int), n, sc['dt'], (0.5 / n) ** 2, sc['gravity'], sc['boundary'])
    got = S.get_state(eng, 3)
    for k, ref in zip('xvCF', (x, v, C, F)):
        assert np.abs(got[k] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k


def test_rigid_shape_matching_matches_numpy(oracle64):
    """MAT_RIGID bodies (mpm:449-505): the C++ restatement against the independent numpy one (numpy SVD), plus the
    defining property -- a body's used particles move by one rigid motion per substep."""
    sc = S.rigid_in_water()
    eng = S.make_engine(oracle64, sc)
    st = S.get_state(eng, 0)
    props = np.array([S.MATERIALS[int(m)] for m in sc['mat']])
    n = sc['n_grid']
    x, v, C, F = (st[k].astype(np.float64) for k in 'xvCF')
    for f in range(4):
        eng.substep(f, f, 0)
        x, v, C, F, _ = mpm_numpy.substep(x, v, C, F, sc['used'], props[:, 0], props[:, 1], (0.5 / n) ** 2 * props[:, 2],
                                          props[:, 3].astype(int), n, sc['dt'], (0.5 / n) ** 2, sc['gravity'], sc['boundary'],
                                          body_id=sc['body_id'])
    got = S.get_state(eng, 4)
    for k, ref in zip('xvCF', (x, v, C, F)):
        assert np.abs(got[k] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k
    x0 = st['x'].astype(np.float64)
    for b in (1, 2):
        sel = (sc['body_id'] == b) & (sc['used'] == 1)
        d0 = np.linalg.norm(x0[sel][:, None] - x0[sel][None], axis=2)
        d4 = np.linalg.norm(got['x'][sel][:, None] - got['x'][sel][None], axis=2)
        assert np.abs(d4 - d0).max() < 1e-12                   # pairwise distances preserved: rigid
        assert np.abs(got['x'][sel] - x0[sel]).max() > 1e-4    # and the body did move


def test_static_colliders_match_numpy(oracle64):
    """Static SDF colliders (static.py:25-104) in grid_op: C++ restatement against the numpy one; the obstacles must
    actually be hit, and no colliding node may keep an inward normal velocity."""
    sc = S.water_on_obstacles()
    eng = S.make_engine(oracle64, sc)
    free = S.make_engine(oracle64, {k: v for k, v in sc.items() if k != 'statics'})
    st = S.get_state(eng, 0)
    props = np.array([S.MATERIALS[int(m)] for m in sc['mat']])
    n = sc['n_grid']
    x, v, C, F = (st[k].astype(np.float64) for k in 'xvCF')
    hit_nodes = 0
    for f in range(6):
        eng.substep(f, f, 0); free.substep(f, f, 0)
        x, v, C, F, aux = mpm_numpy.substep(x, v, C, F, sc['used'], props[:, 0], props[:, 1], (0.5 / n) ** 2 * props[:, 2],
                                            props[:, 3].astype(int), n, sc['dt'], (0.5 / n) ** 2, sc['gravity'], sc['boundary'],
                                            statics=sc['statics'])
        occ = np.argwhere(aux['grid_mass'] > 1e-12) / n
        for s_ in sc['statics']:
            T = np.asarray(s_['T'], np.float64)
            inside = mpm_numpy._sdf_sample(np.asarray(s_['voxels'], np.float64), occ @ T[:3, :3].T + T[:3, 3]) <= 0
            hit_nodes += int(inside.sum())
    got, got_free = S.get_state(eng, 6), S.get_state(free, 6)
    for k, ref in zip('xvCF', (x, v, C, F)):
        assert np.abs(got[k] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k
    assert hit_nodes > 100
    assert np.abs(got['v'] - got_free['v']).max() > 0.05           # the colliders changed the flow


@pytest.mark.parametrize('variant', ['friction', 'soft', 'sticky', 'friction-grid', 'friction-both'])
def test_dynamic_collider_matches_numpy(oracle64, variant):
    """A Rigid effector's moving SDF collider (dynamic.py:29-122): the C++ restatement against the independent numpy one, substep
    by substep with the effector poses the engine produced -- contact at the particles (mpm:418-422), at the grid nodes
    (mpm:393-395) or both, with friction, softness and the sticky (friction > 10) branch."""
    variant, _, where = variant.partition('-')
    kw = dict(friction=dict(friction=0.5, softness=0.0), soft=dict(friction=0.1, softness=60.0), sticky=dict(friction=20.0, softness=0.0))[variant]
    sc = S.stirrer_mini(shape='sphere', **kw)
    ct = dict(grid=2, both=3).get(where, 1)
    eng = S.make_engine(oracle64, sc)
    r = sc['rigid']
    e = eng.add_effector(type=S.FE_EFF_PLAIN, action_dim=6, action_scale_v=r['action_scale_v'], action_scale_p=r['action_scale_p'],
                         boundary=oracle64.make_boundary(**r['boundary']))
    eng.eff_set_mesh(e, r['voxels'], r['T'], friction=r['friction'], softness=r['softness'])
    eng.set_option('collide_type', ct)
    st0 = eng.eff_get_state(e, 0); st0[:7] = r['init_state']; eng.eff_set_state(e, 0, st0)
    eng.eff_apply_action_p(e, sc['action_p'])
    ns = sc['n_substeps']
    props = np.array([S.MATERIALS[int(m)] for m in sc['mat']])
    n = sc['n_grid']
    st = S.get_state(eng, 0)
    x, v, C, F = (st[k].astype(np.float64) for k in 'xvCF')
    touched = 0
    for s_ in range(2):
        eng.eff_set_action(e, s_, s_, ns, sc['actions'][s_])
        eng.step(s_ * ns, s_ * ns, ns, 1)
        for f in range(s_ * ns, (s_ + 1) * ns):
            p0, p1 = eng.eff_get_state(e, f), eng.eff_get_state(e, f + 1)
            dyn = dict(voxels=r['voxels'], T=r['T'], friction=r['friction'], softness=r['softness'],
                       pos0=p0[:3], quat0=p0[3:7], pos1=p1[:3], quat1=p1[3:7])
            free = mpm_numpy.substep(x, v, C, F, sc['used'], props[:, 0], props[:, 1], (0.5 / n) ** 2 * props[:, 2], props[:, 3].astype(int), n,
                                     sc['dt'], (0.5 / n) ** 2, sc['gravity'], sc['boundary'])[1]
            x, v, C, F, _ = mpm_numpy.substep(x, v, C, F, sc['used'], props[:, 0], props[:, 1], (0.5 / n) ** 2 * props[:, 2], props[:, 3].astype(int), n,
                                              sc['dt'], (0.5 / n) ** 2, sc['gravity'], sc['boundary'], dynamic=dyn, collide_type=ct)
            touched += int((np.abs(v - free).max(1) > 1e-9).sum())
    got = S.get_state(eng, 2 * ns)
    for k, ref in zip('xvCF', (x, v, C, F)):
        assert np.abs(got[k] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k
    assert touched > 50                                         # the collider did act on particles


def test_effector_and_injector_chain_matches_numpy(oracle64):
    """The action path of a 6-dof Injector (AgentJetBot): apply_action_p (effector.py:223-225), set_velocity (252-260),
    move_kernel with the quaternion update (157-161) and Injector.act (injector.py:80-105), checked on what the engine kept --
    effector states of every frame and the freshly injected particles -- against the numpy restatement."""
    sc = S.jetbot_mini(n_grid=8, n_coffee=150, n_pool=60, horizon=3, n_substeps=3)
    inj = sc['injector']
    eng = S.make_engine(oracle64, sc)
    e = eng.add_effector(type=S.FE_EFF_INJECTOR, action_dim=6, action_scale_v=inj['action_scale_v'], action_scale_p=inj['action_scale_p'],
                         boundary=oracle64.make_boundary(**inj['boundary']), flux=inj['flux'], radius=inj['radius'], inject_v=inj['inject_v'],
                         inject_p=inj['inject_p'], locally_random=inj['locally_random'], random_vector=inj['random_vector'])
    pool = np.where(sc['used'] == 0)[0].astype(np.int32)
    eng.eff_set_act_range(e, pool)
    st0 = eng.eff_get_state(e, 0); st0[:7] = [0.5, 0.5, 0.5, 1.0, 0.0, 0.0, 0.0]; eng.eff_set_state(e, 0, st0)
    eng.eff_apply_action_p(e, sc['action_p'])
    ns, H = sc['n_substeps'], sc['horizon']
    pos = mpm_numpy.impose_x(inj['boundary'], np.asarray(sc['action_p'][:3], np.float64) * np.asarray(inj['action_scale_p'][:3]))
    quat = np.array([1.0, 0.0, 0.0, 0.0])
    assert np.abs(eng.eff_get_state(e, 0)[:3] - pos).max() < 1e-12
    n_inj = 0
    for s_ in range(H):
        a = sc['actions'][s_].astype(np.float64)
        eng.eff_set_action(e, s_, s_, ns, a)
        eng.step(s_ * ns, s_ * ns, ns, 1)
        vv = a[:3] * np.asarray(inj['action_scale_v'][:3]) / ns
        ww = a[3:] * np.asarray(inj['action_scale_v'][3:]) / ns
        for f in range(s_ * ns, (s_ + 1) * ns):
            # Injector.act at frame f: flux pool particles appear in frame f+1 around pos[f] + R(quat[f]) inject_p
            nxt = S.get_state(eng, f + 1)
            ids = pool[n_inj:n_inj + inj['flux']]
            rv = inj['random_vector'][f].astype(np.float64)               # locally_random: row f
            ip = mpm_numpy._quat_rot(np.asarray(inj['inject_p'], np.float64)[None], quat)[0]
            iv = mpm_numpy._quat_rot(np.asarray(inj['inject_v'], np.float64)[None], quat)[0]
            assert (nxt['used'][ids] == 1).all() and (nxt['used'][pool[n_inj + inj['flux']:]] == 0).all()
            newest = nxt['x'][ids]                                        # (older particles have moved on; the newest are where they were put)
            assert np.abs(newest - ((rv * 2 - 1) * inj['radius'] + pos + ip)).max() < 1e-12, f
            assert np.abs(nxt['v'][ids] - iv).max() < 1e-12
            n_inj += inj['flux']
            pos, quat = mpm_numpy.effector_move(inj['boundary'], pos, quat, vv, ww)
            got = eng.eff_get_state(e, f + 1)
            assert np.abs(got[:3] - pos).max() < 1e-12 and np.abs(got[3:7] - quat).max() < 1e-12, f
    assert np.abs(quat - [1, 0, 0, 0]).max() > 1e-3                       # the nozzle did turn


def test_cylinder_boundary_matches_numpy(oracle64):
    sc = S.latte_mini()
    sc = dict(sc, used=np.where(sc['used'] == 1, 1, 0).astype(np.int32))
    eng = S.make_engine(oracle64, sc)
    st = S.get_state(eng, 0)
    props = np.array([S.MATERIALS[int(m)] for m in sc['mat']])
    n = sc['n_grid']
    x, v, C, F = (st[k].astype(np.float64) for k in 'xvCF')
    for f in range(12):
        eng.substep(f, f, 0)
        x, v, C, F, _ = mpm_numpy.substep(x, v, C, F, sc['used'], props[:, 0], props[:, 1], (0.5 / n) ** 2 * props[:, 2],
                                          props[:, 3].astype(int), n, sc['dt'], (0.5 / n) ** 2, sc['gravity'], sc['boundary'])
    got = S.get_state(eng, 12)
    act = sc['used'] == 1
    assert np.abs(got['x'][act] - x[act]).max() < 1e-10
    assert np.abs(got['v'][act] - v[act]).max() < 1e-8
    # unused pool particles are copied frame to frame (mpm:309-316)
    assert (got['x'][~act] == -100.0).all()


def test_invariants(oracle64):
    sc = S.water_block(n_grid=16, n_particles=600, lo=0.3, hi=0.6)
    props = np.array([S.MATERIALS[int(m)] for m in sc['mat']])
    n = sc['n_grid']
    x = sc['x'].astype(np.float64)
    N = sc['N']
    v = np.tile([0.3, -0.2, 0.1], (N, 1))
    C = np.zeros((N, 3, 3)); F = np.tile(np.eye(3), (N, 1, 1))
    mass = (0.5 / n) ** 2 * props[:, 2]
    _, v2, C2, F2, aux = mpm_numpy.substep(x, v, C, F, sc['used'], props[:, 0], props[:, 1], mass, props[:, 3].astype(int), n,
                                           sc['dt'], (0.5 / n) ** 2, (0, 0, 0), sc['boundary'])
    # P2G conserves mass and momentum; G2P of a constant field returns it with C = 0
    assert abs(aux['grid_mass'].sum() - mass.sum()) < 1e-12 * mass.sum()
    assert np.abs(aux['grid_v_in'].sum((0, 1, 2)) - (mass[:, None] * v).sum(0)).max() < 1e-12
    assert np.abs(v2 - v).max() < 1e-12 and np.abs(C2).max() < 1e-9
    # and the C++ oracle agrees
    eng = S.make_engine(oracle64, sc)
    eng.set_frame(0, v=v)
    eng.set_option('threads', 1)
    eng.substep(0, 0, 0)
    got = S.get_state(eng, 1)
    g = np.array(sc['gravity'])
    assert np.abs(got['v'] - (v + sc['dt'] * g)).max() < 1e-12
    assert np.abs(got['F'] - F2).max() < 1e-12


def _loss_of(eng, sc, n_sub, cot, x=None, v=None, C=None, F=None):
    eng.set_frame(0, x=x, v=v, C_=C, F=F)
    st = S.run_forward(eng, n_sub)
    return float((st['x'] * cot['gx']).sum() + (st['v'] * cot['gv']).sum() + (st['C'] * cot['gC']).sum() + (st['F'] * cot['gF']).sum())


@pytest.mark.parametrize('scene', ['mixed', 'water_wall', 'rigid', 'statics'])
def test_substep_adjoint_vs_finite_differences(oracle64, scene):
    if scene == 'mixed':
        sc = S.mixed_materials(n_grid=8, n_particles=40, seed=3)
        sc['x'] = S.f32(np.random.RandomState(3).uniform(0.3, 0.62, (40, 3)))
    elif scene == 'statics':
        # static SDF colliders in grid_op: contact branch (normal removal + Coulomb friction) and its adjoint
        sc = S.water_on_obstacles(n_grid=8, n_particles=60)
    elif scene == 'rigid':
        # MAT_RIGID shape matching: COM / covariance / SVD / rotation chain and its adjoint (mpm:436-505)
        sc = S.rigid_in_water(n_grid=8, n_water=30, n_rigid=(12, 9), seed=6)
    else:
        # particles pressed against the cube wall: exercises the boundary branch of grid_op's adjoint
        sc = S.water_block(n_grid=8, n_particles=40, lo=0.13, hi=0.4)
        sc['boundary'] = dict(type='cube', lower=(0.25, 0.25, 0.25), upper=(0.8, 0.8, 0.8))
        sc['v'] = S.f32(np.random.RandomState(4).normal(0, 1.0, (40, 3)))
    N = sc['N']
    eng = S.make_engine(oracle64, sc, max_substeps_local=8)
    eng.set_option('threads', 1)
    n_sub = 3
    base = {k: S.get_state(eng, 0)[k].astype(np.float64) for k in 'xvCF'}
    cot = {k: a.astype(np.float64) for k, a in S.random_cotangent(N).items()}
    _, g = S.run_forward_backward(eng, n_sub, cot)
    rng = np.random.RandomState(0)
    used = sc['used'] == 1
    for name, gname, h in [('x', 'gx', 1e-7), ('v', 'gv', 1e-6), ('C', 'gC', 1e-5), ('F', 'gF', 1e-7)]:
        diffs, mags = [], []
        for _ in range(10):
            p = rng.choice(np.where(used)[0])
            idx = (p,) + tuple(rng.randint(0, 3, base[name].ndim - 1))
            arrs = {k: a.copy() for k, a in base.items()}
            arrs[name][idx] += h
            lp = _loss_of(eng, sc, n_sub, cot, arrs['x'], arrs['v'], arrs['C'], arrs['F'])
            arrs[name][idx] -= 2 * h
            lm = _loss_of(eng, sc, n_sub, cot, arrs['x'], arrs['v'], arrs['C'], arrs['F'])
            fd = (lp - lm) / (2 * h)
            diffs.append(abs(fd - g[gname][idx])); mags.append(abs(fd))
        assert max(diffs) < 2e-5 * max(max(mags), 1e-6), (name, diffs, mags)


def test_action_gradient_vs_finite_differences(oracle64):
    """dL/d(action) through injector, effector move, set_velocity and apply_action_p."""
    sc = S.latte_mini(n_grid=8, n_coffee=150, n_pool=40, horizon=3, n_substeps=3)
    out = S.run_latte(oracle64, sc)
    g = out['action_grad']
    assert g.shape == (sc['horizon'] + 1, 3)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    # effector y is clamped by y_range=(0.62, 0.62): its gradient is killed (boundaries.py:68)
    assert np.abs(g[:, 1]).max() == 0.0

    def total(sc_):
        return float(S.run_latte(oracle64, sc_)['step_loss'].astype(np.float64).sum())

    h = 1e-6
    for (s, k) in [(0, 0), (1, 2), (2, 0)]:
        scp = dict(sc, actions=sc['actions'].astype(np.float64).copy()); scp['actions'][s, k] += h
        scm = dict(sc, actions=sc['actions'].astype(np.float64).copy()); scm['actions'][s, k] -= h
        fd = (total(scp) - total(scm)) / (2 * h)
        assert abs(fd - g[s, k]) < 1e-5 * max(abs(fd), 1e-3), (s, k, fd, g[s, k])
    for k in (0, 2):
        scp = dict(sc, action_p=sc['action_p'].astype(np.float64).copy()); scp['action_p'][k] += h
        scm = dict(sc, action_p=sc['action_p'].astype(np.float64).copy()); scm['action_p'][k] -= h
        fd = (total(scp) - total(scm)) / (2 * h)
        assert abs(fd - g[-1, k]) < 1e-5 * max(abs(fd), 1e-3), (k, fd, g[-1, k])


def test_turning_injector_gradient_vs_finite_differences(oracle64):
    """A 6-dof Injector (AgentJetBot, agent_transporting.yaml): the angular action reaches the loss through
    quat[f] -> R(quat) inject_p / R(quat) inject_v of the particles injected at f (injector.py:92-96)."""
    sc = S.jetbot_mini(n_grid=8, n_coffee=150, n_pool=40, horizon=3, n_substeps=3)
    out = S.run_latte(oracle64, sc)
    g = out['action_grad']
    assert g.shape == (sc['horizon'] + 1, 6) and np.abs(g[:3, 3:]).max() > 1e-6

    def total(sc_):
        return float(S.run_latte(oracle64, sc_)['step_loss'].astype(np.float64).sum())

    h = 1e-6
    for (s, k) in [(0, 3), (0, 5), (1, 4), (1, 0), (2, 5)]:
        scp = dict(sc, actions=sc['actions'].astype(np.float64).copy()); scp['actions'][s, k] += h
        scm = dict(sc, actions=sc['actions'].astype(np.float64).copy()); scm['actions'][s, k] -= h
        fd = (total(scp) - total(scm)) / (2 * h)
        assert abs(fd - g[s, k]) < 1e-5 * max(abs(fd), 1e-3), (s, k, fd, g[s, k])


@pytest.mark.parametrize('variant', ['friction', 'soft', 'sticky', 'friction-grid', 'friction-both'])
def test_rigid_effector_gradient_vs_finite_differences(oracle64, variant):
    """Dynamic.collide at particle level (dynamic.py:29-122, mpm:418-422) and the 6-dof pose chain (move_kernel's
    quaternion update, effector.py:157-161; set_velocity's w part, 252-260): dL/d(actions) against central differences."""
    variant, _, where = variant.partition('-')
    kw = dict(friction=dict(friction=0.5, softness=0.0), soft=dict(friction=0.1, softness=60.0), sticky=dict(friction=20.0, softness=0.0))[variant]
    sc = S.stirrer_mini(n_grid=8, n_particles=120, horizon=3, n_substeps=3, **kw)
    if where:                # the collider chain at the grid nodes too (mpm:393-395, AgentPouring's collide_type='both')
        sc['collide_type'] = dict(grid=2, both=3)[where]
    cot = {k: v.astype(np.float64) for k, v in S.random_cotangent(sc['N']).items()}
    out = S.run_rigid(oracle64, sc, cot)
    if where:
        ref = S.run_rigid(oracle64, dict(sc, collide_type=1), cot)
        assert np.abs(out['final']['v'] - ref['final']['v']).max() > 1e-2             # the node-level contact changes the flow
    g = out['action_grad']
    assert g.shape == (4, 6) and np.abs(g[:3, :3]).max() > 1e-3 and np.abs(g[:3, 3:]).max() > 1e-5
    # the collider must actually touch the water: same scene without the mesh contact gives another result
    # Contact is only piecewise smooth (voxel cells of the trilinear SDF, the sdf <= 0 / friction branches): a central
    # difference that straddles a kink is off by O(1), so each component is differenced at three step sizes and the best
    # one has to agree with the adjoint.
    rng = np.random.RandomState(0)

    def fd_err(run, exact, steps):
        errs = []
        for hstep in steps:
            fd = (run(+hstep) - run(-hstep)) / (2 * hstep)
            errs.append(abs(fd - exact) / max(abs(fd), 1e-3))
        return min(errs)

    worst = 0.0
    for _ in range(8):
        s_, k_ = rng.randint(0, 3), rng.randint(0, 6)

        def run(dh, s_=s_, k_=k_):
            a = sc['actions'].astype(np.float64).copy(); a[s_, k_] += dh
            return S.run_rigid(oracle64, sc, cot, actions=a)['loss']
        worst = max(worst, fd_err(run, g[s_, k_], (1e-5, 1e-6, 1e-7, 1e-8)))
    # softness > 0 adds a jump where influence crosses 0.1 (dynamic.py:99): differences only converge slowly there
    assert worst < (2e-2 if variant == 'soft' else 1e-5), worst
    for k_ in range(3):
        def run(dh, k_=k_):
            a = sc['action_p'].astype(np.float64).copy(); a[k_] += dh
            return S.run_rigid(oracle64, sc, cot, action_p=a)['loss']
        assert fd_err(run, g[3, k_], (1e-5, 1e-6, 1e-7, 1e-8)) < (5e-2 if variant == 'soft' else 2e-3)      # moves the whole collider: many kinks nearby


def test_collector_takes_particles_out(oracle64):
    """collector_act_kernel (agent_pouring.py:30-41): replayed substep by substep from the frames the engine kept, a
    particle is taken exactly when it was used and outside the collector box at the start of the substep; from then on it
    is unused, parked at NOWHERE, and -- as for any unused particle -- the adjoint passes straight through (mpm:551)."""
    sc = S.pouring_mini(n_grid=8, n_particles=300, horizon=3, n_substeps=3)
    cot = {k: v.astype(np.float64) for k, v in S.random_cotangent(sc['N']).items()}
    eng = S.make_engine(oracle64, sc)
    r = sc['rigid']
    e = eng.add_effector(type=S.FE_EFF_PLAIN, action_dim=6, action_scale_v=r['action_scale_v'], action_scale_p=r['action_scale_p'],
                         boundary=oracle64.make_boundary(**r['boundary']))
    eng.agent_set_collector(oracle64.make_boundary(**sc['collector']['boundary']), -1)
    lo, up = np.array(sc['collector']['boundary']['lower']), np.array(sc['collector']['boundary']['upper'])
    taken_total, later = 0, 0
    for f in range(9):
        before = S.get_state(eng, f)
        eng.substep(f, f, 1)
        cur, nxt = S.get_state(eng, f), S.get_state(eng, f + 1)
        out = ((before['x'] > up) | (before['x'] < lo)).any(1) & (before['used'] == 1)
        assert (cur['used'][out] == 0).all() and (nxt['used'][out] == 0).all() and (nxt['x'][out] == -100.0).all()
        assert (nxt['v'][out] == before['v'][out]).all() and (nxt['F'][out] == before['F'][out]).all()
        keep = (before['used'] == 1) & ~out
        assert (cur['used'][keep] == 1).all() and (nxt['used'][keep] == 1).all() and (nxt['x'][keep] > -1).all()
        gone = before['used'] == 0
        assert (nxt['used'][gone] == 0).all() and (nxt['x'][gone] == before['x'][gone]).all()
        taken_total += int(out.sum()); later += int(out.sum()) if f > 0 else 0
    assert taken_total > 20 and later > 3
    # a none-action substep does not run agent.act (mpm:318-320)
    eng2 = S.make_engine(oracle64, sc)
    eng2.agent_set_collector(oracle64.make_boundary(**sc['collector']['boundary']), -1)
    eng2.substep(0, 0, 0)
    assert (S.get_state(eng2, 1)['used'] == 1).all()
    # material filter (agent_jetbot.py:37): nothing here is of material 99
    eng3 = S.make_engine(oracle64, sc)
    eng3.agent_set_collector(oracle64.make_boundary(**sc['collector']['boundary']), 99)
    eng3.substep(0, 0, 1)
    assert (S.get_state(eng3, 1)['used'] == 1).all()
    # adjoint against central differences on the actions (collection events do not move under a 1e-7 nudge).  As in the
    # reference, process_unused_particles.grad also hands d/dx[f+1] of a taken particle back to x[f] although x[f+1] is the
    # constant NOWHERE; losses mask unused particles (pouring_loss.py:131-135), so the cotangent does too.
    out = S.run_rigid(oracle64, sc, cot)
    assert out['used_hist'][-1].sum() < sc['N'] - 20
    cot['gx'][out['used_hist'][-1] == 0] = 0.0
    out = S.run_rigid(oracle64, sc, cot)
    eng = None
    rng = np.random.RandomState(1)
    g = out['action_grad']
    for _ in range(4):
        s_, k_ = rng.randint(0, 3), rng.randint(0, 6)
        best = 1e9
        for hstep in (1e-5, 1e-6, 1e-7):
            a1 = sc['actions'].astype(np.float64).copy(); a1[s_, k_] += hstep
            a0 = sc['actions'].astype(np.float64).copy(); a0[s_, k_] -= hstep
            fd = (S.run_rigid(oracle64, sc, cot, actions=a1)['loss'] - S.run_rigid(oracle64, sc, cot, actions=a0)['loss']) / (2 * hstep)
            best = min(best, abs(fd - g[s_, k_]) / max(abs(fd), 1e-3))
        assert best < 1e-4, best


def test_oracle_f32_tracks_f64(oracle32, oracle64):
    sc = S.water_block(n_grid=16, n_particles=800)
    a = S.run_forward(S.make_engine(oracle32, sc), 20)
    b = S.run_forward(S.make_engine(oracle64, sc), 20)
    assert np.abs(a['x'] - b['x']).max() < 2e-6
    assert S.rel_l2(a['v'], b['v']) < 1e-3


def test_errors_are_reported_not_fatal(oracle64):
    sc = S.water_block(n_grid=8, n_particles=10)
    sc['x'][0] = [0.99, 0.99, 0.99]                     # stencil leaves the grid
    eng = S.make_engine(oracle64, sc, max_substeps_local=4)
    from fluidlab_amd._capi import FeEngineError
    with pytest.raises(FeEngineError, match='left the grid'):
        eng.substep(0, 0, 0)
    with pytest.raises(FeEngineError, match='out of range'):
        eng.substep(4, 4, 0)


def test_batch_entry_points_step_every_engine(oracle64):
    """fe_step_batch / fe_step_grad_batch on the oracle: the engines of the list are stepped in lockstep (one after the other here)"""
    scenes = [S.water_block(n_grid=16, n_particles=600, seed=sd) for sd in (0, 1)]
    L = 6
    solo = []
    for sc in scenes:
        eng = S.make_engine(oracle64, sc, max_substeps_local=L)
        eng.step(0, 0, L, 0)
        solo.append(S.get_state(eng, L)['x'])
        eng.close()
    engs = [S.make_engine(oracle64, sc, max_substeps_local=L) for sc in scenes]
    type(engs[0]).step_batch(engs, 0, 0, L, 0)
    cot = S.random_cotangent(600)
    for e in engs:
        e.reset_grad(); e.add_grad(L, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
    type(engs[0]).step_grad_batch(engs, 0, 0, L, 0)
    for e, x0 in zip(engs, solo):
        assert np.array_equal(S.get_state(e, L)['x'], x0)
        assert np.abs(e.get_grad(0)[0]).max() > 0
        e.close()

"""GPU parity: the gfx950 HIP engine against the fp64 CPU oracle on identical inputs, through the C ABI.

Tolerances (SURVEY 8c, fp32 engine vs fp64 oracle; both sides of the reference are themselves
order-nondeterministic in fp32 because of float atomics):
  one substep      max|dx| <= 1e-6,  max|dv| <= 1e-4 * max(1, |v|inf)
  50 substeps      rel L2(x) <= 1e-4
  adjoints         cosine >= 0.999 and rel L2 <= 1e-2 (mu = 0 liquids much tighter)
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import scenarios as S  # noqa: E402

pytestmark = pytest.mark.gpu


def test_native_library_is_the_one_loaded(hiplib):
    assert hiplib.backend == 'hip-gfx950'
    assert os.path.samefile(hiplib.path, os.path.join(os.path.dirname(S.__file__), '..', 'fluidlab_amd', 'csrc', 'libfluidengine_hip.so'))


def test_water_block_forward(hiplib, oracle64):
    sc = S.water_block(n_grid=32, n_particles=8192)
    g = S.make_engine(hiplib, sc)
    o = S.make_engine(oracle64, sc)
    a, b = S.run_forward(g, 1), S.run_forward(o, 1)
    assert np.abs(a['x'] - b['x']).max() <= 1e-6
    assert np.abs(a['v'] - b['v']).max() <= 1e-4 * max(1.0, np.abs(b['v']).max())
    assert np.abs(a['F'] - b['F']).max() <= 1e-6
    assert np.abs(a['C'] - b['C']).max() <= 1e-5          # C ~ 0 after one substep from rest
    a, b = S.run_forward(g, 49, f0=1), S.run_forward(o, 49, f0=1)
    assert S.rel_l2(a['x'], b['x']) <= 1e-4
    assert S.rel_l2(a['v'], b['v']) <= 1e-2
    assert (a['used'] == b['used']).all()
    g.sync()


def test_all_materials_forward(hiplib, oracle64):
    sc = S.mixed_materials()
    a = S.run_forward(S.make_engine(hiplib, sc), 10)
    b = S.run_forward(S.make_engine(oracle64, sc), 10)
    assert np.abs(a['x'] - b['x']).max() <= 1e-5
    assert S.rel_l2(a['v'], b['v']) <= 1e-3
    assert S.rel_l2(a['F'], b['F']) <= 1e-5
    assert S.rel_l2(a['C'], b['C']) <= 1e-2
    unused = sc['used'] == 0
    for k in 'xvCF':                                       # unused particles are carried unchanged
        assert (a[k][unused] == b[k][unused].astype(np.float32)).all()


# sort_interval K: 0 = never sort (every particle on the global-atomics path), 1 = sort every substep,
# 3 = sorts that do not line up with anything, 10 = default
@pytest.mark.parametrize('K', [0, 1, 3, 10])
@pytest.mark.parametrize('scene', ['water', 'mixed'])
def test_substep_adjoint(hiplib, oracle64, scene, K):
    if scene == 'water':
        sc = S.water_block(n_grid=16, n_particles=2000)
        sc['v'] = S.f32(np.random.RandomState(9).normal(0, 0.5, (2000, 3)))
        tol_l2 = 4e-6           # measured 1.1e-6 (gC)
    else:
        sc = S.mixed_materials()
        tol_l2 = 2e-3           # measured 5.4e-4 (gF: SVD / plastic-clamp adjoint in fp32); 1e-2 in round 1
    cot = S.random_cotangent(sc['N'])
    _, ga = S.run_forward_backward(S.make_engine(hiplib, sc, options={'sort_interval': K}), 6, cot)
    _, gb = S.run_forward_backward(S.make_engine(oracle64, sc), 6, {k: v.astype(np.float64) for k, v in cot.items()})
    print(f'MEASURED substep_adjoint[{scene}, K={K}]: ' + ' '.join(f'{k} relL2 {S.rel_l2(ga[k], gb[k]):.2e}' for k in ('gx', 'gv', 'gC', 'gF')))
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert np.isfinite(ga[k]).all(), k
        assert S.cosine(ga[k], gb[k]) >= 0.999, (k, S.cosine(ga[k], gb[k]))
        assert S.rel_l2(ga[k], gb[k]) <= tol_l2, (k, S.rel_l2(ga[k], gb[k]))


@pytest.mark.parametrize('variant', ['friction', 'soft', 'sticky'])
@pytest.mark.parametrize('K', [0, 10])
def test_rigid_effector(hiplib, oracle64, variant, K):
    """A Rigid effector's moving SDF collider (dynamic.py:29-122) at particle level in g2p (mpm:418-422), 6-dof action:
    forward state, effector pose, and dL/d(actions) through contact, collider velocity and the quaternion update."""
    kw = dict(friction=dict(friction=0.5, softness=0.0), soft=dict(friction=0.1, softness=60.0), sticky=dict(friction=20.0, softness=0.0))[variant]
    # friction/soft use a sphere: inside a box the SDF normal jumps across the medial axis, and fp32 vs fp64 runs (the
    # oracle's own f32 and f64 builds included) then put individual particles on different contact branches
    sc = S.stirrer_mini(shape='box' if variant == 'sticky' else 'sphere', **kw)
    cot = S.random_cotangent(sc['N'])
    a = S.run_rigid(hiplib, sc, cot, options={'sort_interval': K})
    b = S.run_rigid(oracle64, sc, {k: v.astype(np.float64) for k, v in cot.items()})
    free = S.run_forward(S.make_engine(hiplib, sc), sc['horizon'] * sc['n_substeps'])
    assert np.abs(a['final']['v'] - free['v']).max() > 0.05                      # the collider touches the water
    assert np.abs(a['eff_state'] - b['eff_state']).max() <= 1e-6
    assert np.abs(a['final']['x'] - b['final']['x']).max() <= 5e-6
    assert S.rel_l2(a['final']['v'], b['final']['v']) <= 2e-3
    ga, gb = a['action_grad'], b['action_grad']
    assert np.isfinite(ga).all()
    assert S.cosine(ga, gb) >= 0.999999, S.cosine(ga, gb)
    assert S.rel_l2(ga, gb) <= 1e-3, S.rel_l2(ga, gb)                                  # measured 5e-6 .. 9e-5
    assert S.cosine(a['gx0'], b['gx0']) >= 0.9999 and S.rel_l2(a['gx0'], b['gx0']) <= 1e-2


@pytest.mark.parametrize('where', ['grid', 'both'])
@pytest.mark.parametrize('K', [0, 10])
def test_rigid_effector_collides_at_grid_nodes(hiplib, oracle64, where, K):
    """Agent.collide_type 'grid' / 'both' (agent.py:17-26; AgentPouring uses 'both'): the effector's collider chain inside
    grid_op (mpm:393-395) and its adjoint into the node velocities and the effector pose."""
    sc = S.stirrer_mini(shape='sphere', friction=0.5, softness=0.0)
    sc['collide_type'] = dict(grid=2, both=3)[where]
    cot = S.random_cotangent(sc['N'])
    a = S.run_rigid(hiplib, sc, cot, options={'sort_interval': K})
    b = S.run_rigid(oracle64, sc, {k: v.astype(np.float64) for k, v in cot.items()})
    p = S.run_rigid(hiplib, dict(sc, collide_type=1), cot, options={'sort_interval': K})
    assert np.abs(a['final']['v'] - p['final']['v']).max() > 0.05                   # not the particle-only result
    assert np.abs(a['eff_state'] - b['eff_state']).max() <= 1e-6
    assert np.abs(a['final']['x'] - b['final']['x']).max() <= 5e-6
    assert S.rel_l2(a['final']['v'], b['final']['v']) <= 2e-3
    ga, gb = a['action_grad'], b['action_grad']
    assert np.isfinite(ga).all()
    assert S.cosine(ga, gb) >= 0.99999, S.cosine(ga, gb)
    assert S.rel_l2(ga, gb) <= 2e-3, S.rel_l2(ga, gb)
    assert S.cosine(a['gx0'], b['gx0']) >= 0.9999 and S.rel_l2(a['gx0'], b['gx0']) <= 1e-2


@pytest.mark.parametrize('K', [0, 3])
def test_collector(hiplib, oracle64, K):
    """AgentPouring in small: collide_type='both' plus the collector (agent_pouring.py:30-41).  The same particles are taken
    in the same substeps as in the oracle, they end parked at NOWHERE, and the action gradient agrees."""
    sc = S.pouring_mini()
    cot = S.random_cotangent(sc['N'])
    b = S.run_rigid(oracle64, sc, {k: v.astype(np.float64) for k, v in cot.items()})
    cot['gx'][b['used_hist'][-1] == 0] = 0.0                 # losses mask unused particles (pouring_loss.py:131-135)
    b = S.run_rigid(oracle64, sc, {k: v.astype(np.float64) for k, v in cot.items()})
    a = S.run_rigid(hiplib, sc, cot, options={'sort_interval': K})
    n_taken = int((b['used_hist'][-1] == 0).sum())
    assert n_taken > 50 and (b['used_hist'][1] == 0).sum() < n_taken
    # a particle within fp32 rounding of the collector face may be taken a substep apart; allow a handful
    assert (a['used_hist'] != b['used_hist']).any(0).sum() <= 2
    same = (a['used_hist'] == b['used_hist']).all(0)
    gone = same & (b['used_hist'][-1] == 0)
    assert (a['final']['x'][gone] == -100.0).all()
    assert S.rel_l2(a['final']['v'][gone], b['final']['v'][gone]) <= 1e-4       # carried over from the substep they were taken in
    live = same & (b['used_hist'][-1] == 1)
    assert np.abs(a['final']['x'][live] - b['final']['x'][live]).max() <= 5e-6
    ga, gb = a['action_grad'], b['action_grad']
    assert S.cosine(ga, gb) >= 0.9999, S.cosine(ga, gb)
    assert S.rel_l2(ga, gb) <= 1e-2, S.rel_l2(ga, gb)


@pytest.mark.parametrize('K', [0, 10])
def test_static_sdf_colliders(hiplib, oracle64, K):
    """grid_op's collide-with-statics (mpm:386-390, static.py:82-103): trilinear SDF, finite-difference normal, contact
    law with Coulomb friction -- forward and adjoint, two colliders in sequence."""
    sc = S.water_on_obstacles()
    cot = S.random_cotangent(sc['N'])
    sa, ga = S.run_forward_backward(S.make_engine(hiplib, sc, options={'sort_interval': K}), 8, cot)
    sb, gb = S.run_forward_backward(S.make_engine(oracle64, sc), 8, {k: v.astype(np.float64) for k, v in cot.items()})
    free = S.run_forward(S.make_engine(hiplib, {k: v for k, v in sc.items() if k != 'statics'}), 8)
    assert np.abs(sa['v'] - free['v']).max() > 0.05                      # the colliders act
    assert np.abs(sa['x'] - sb['x']).max() <= 2e-6
    assert S.rel_l2(sa['v'], sb['v']) <= 1e-3
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert np.isfinite(ga[k]).all(), k
        assert S.cosine(ga[k], gb[k]) >= 0.999, (k, S.cosine(ga[k], gb[k]))
        assert S.rel_l2(ga[k], gb[k]) <= 1e-2, (k, S.rel_l2(ga[k], gb[k]))


@pytest.mark.parametrize('K', [0, 3, 10])
def test_rigid_bodies(hiplib, oracle64, K):
    """MAT_RIGID shape matching (mpm:428-505): per-body COM/covariance reductions, 3x3 SVD, rotation, and the adjoint
    chain advect_kernel.grad -> compute_R.grad -> compute_H_svd_grad -> compute_H.grad -> compute_COM.grad."""
    sc = S.rigid_in_water()
    cot = S.random_cotangent(sc['N'])
    g = S.make_engine(hiplib, sc, options={'sort_interval': K})
    o = S.make_engine(oracle64, sc)
    sa, ga = S.run_forward_backward(g, 8, cot)
    sb, gb = S.run_forward_backward(o, 8, {k: v.astype(np.float64) for k, v in cot.items()})
    assert np.abs(sa['x'] - sb['x']).max() <= 2e-6
    assert S.rel_l2(sa['v'], sb['v']) <= 1e-3 and S.rel_l2(sa['F'], sb['F']) <= 1e-5
    x0 = sc['x'].astype(np.float64)
    for b in (1, 2):                                           # each body moved by one rigid motion, in fp32
        sel = (sc['body_id'] == b) & (sc['used'] == 1)
        d0 = np.linalg.norm(x0[sel][:, None] - x0[sel][None], axis=2)
        d1 = np.linalg.norm(sa['x'][sel].astype(np.float64)[:, None] - sa['x'][sel].astype(np.float64)[None], axis=2)
        assert np.abs(d1 - d0).max() <= 2e-6
    rigid = sc['body_id'] > 0
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert np.isfinite(ga[k]).all(), k
        assert S.cosine(ga[k], gb[k]) >= 0.999, (k, S.cosine(ga[k], gb[k]))
        assert S.rel_l2(ga[k], gb[k]) <= 1e-2, (k, S.rel_l2(ga[k], gb[k]))
        assert S.rel_l2(ga[k][rigid], gb[k][rigid]) <= 1e-2, (k, 'rigid particles', S.rel_l2(ga[k][rigid], gb[k][rigid]))


@pytest.mark.parametrize('K', [0, 1, 3, 10])
def test_latte_mini_trajectory_gradient(hiplib, oracle64, K):
    """Injector + cylinder boundary + loss + action gradient, end to end."""
    sc = S.latte_mini()
    a = S.run_latte(hiplib, sc, options={'sort_interval': K})
    b = S.run_latte(oracle64, sc)
    assert (a['final']['used'] == b['final']['used']).all()
    assert a['final']['used'].sum() == (sc['used'] == 1).sum() + sc['horizon'] * sc['n_substeps'] * sc['injector']['flux']
    assert S.rel_l2(a['final']['x'], b['final']['x']) <= 1e-5
    assert S.rel_l2(a['step_loss'], b['step_loss']) <= 1e-4
    assert np.abs(a['eff_state'] - b['eff_state']).max() <= 1e-6
    assert S.cosine(a['action_grad'], b['action_grad']) >= 0.999999
    assert S.rel_l2(a['action_grad'], b['action_grad']) <= 1e-4          # inviscid liquids: measured ~2e-7


@pytest.mark.parametrize('K', [0, 3])
def test_turning_injector(hiplib, oracle64, K):
    """A 6-dof Injector (AgentJetBot): injection point and jet velocity rotate with quat[f] (injector.py:92-96); the angular
    action's gradient runs through the quaternion chain."""
    sc = S.jetbot_mini()
    a = S.run_latte(hiplib, sc, options={'sort_interval': K})
    b = S.run_latte(oracle64, sc)
    assert (a['final']['used'] == b['final']['used']).all()
    assert S.rel_l2(a['final']['x'], b['final']['x']) <= 1e-5
    assert np.abs(a['eff_state'] - b['eff_state']).max() <= 1e-6
    assert np.abs(b['action_grad'][:-1, 3:]).max() > 1e-6
    assert S.cosine(a['action_grad'], b['action_grad']) >= 0.999999
    assert S.rel_l2(a['action_grad'], b['action_grad']) <= 1e-4
    assert S.rel_l2(a['action_grad'][:, 3:], b['action_grad'][:, 3:]) <= 1e-3


def test_fast_particles_leave_their_tiles(hiplib, oracle64):
    """Particles that outrun the 1-node drift margin of their LDS tile between two sorts must fall
    back to the global path and still match the oracle."""
    sc = S.water_block(n_grid=32, n_particles=4000, lo=0.3, hi=0.5, gravity=(0.0, 0.0, 0.0))
    # a coherent 12 m/s drift: 12 * 20 substeps * 2e-4 * 32 = 1.5 cells between the two sorts
    # (random per-particle velocities would be averaged away by the first P2G/G2P)
    sc['v'] = S.f32(np.tile([12.0, -9.0, 4.0], (4000, 1)))
    g = S.make_engine(hiplib, sc, options={'sort_interval': 20})
    o = S.make_engine(oracle64, sc)
    cot = S.random_cotangent(sc['N'])
    sa, ga = S.run_forward_backward(g, 20, cot)
    sb, gb = S.run_forward_backward(o, 20, {k: v.astype(np.float64) for k, v in cot.items()})
    assert S.rel_l2(sa['x'], sb['x']) <= 1e-5 and S.rel_l2(sa['v'], sb['v']) <= 1e-3
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.cosine(ga[k], gb[k]) >= 0.999 and S.rel_l2(ga[k], gb[k]) <= 1e-2, k
    assert g.get_stats(20)['n_slow_path'] > 0


@pytest.mark.parametrize('loose_max,quads', [(0, 0), (0, 1), (12, 0), (12, 1)])
@pytest.mark.parametrize('grid_store', [1, 0])
def test_scattered_droplets_and_untouched_blocks(hiplib, oracle64, grid_store, loose_max, quads):
    """Water that has come apart: droplets all over the box (most blocks of the active list get nothing, most items hold one or two
    particles) plus a dense clump, some of it fast enough to leave its tiles between two sorts (slow-path deposits into blocks of the
    active list: the dirty marks).  The grid kernels work on the marked entries only -- forward, backward from the stored grid
    (grid_store 1) and backward from the recompute (grid_store 0) must match the oracle, and so must the work list the stats report.
    loose_max 12: blocks with up to 12 particles get no work item, their particles are worked on in cell order by the global path.
    quads 1: the single-item blocks of at most 64 particles go four to a workgroup, one wave and one fixed-point LDS tile each (the
    scatter kernels' quad units; by default only orders with more than `quad_min_units` (1,400) workgroups of pairs get them)."""
    rng = np.random.RandomState(3)
    n_drop, n_clump = 1500, 2500
    sc = S.water_block(n_grid=64, n_particles=n_drop + n_clump, lo=0.40, hi=0.52)
    sc['x'][:n_drop] = S.f32(rng.uniform(0.08, 0.92, (n_drop, 3)))
    sc['v'] = S.f32(rng.normal(0, 1.0, (n_drop + n_clump, 3)))
    sc['v'][n_drop:] = S.f32(rng.normal(0, 0.2, (n_clump, 3)) + [12.0, -9.0, 4.0])       # the clump drifts 1.5 cells between two sorts
    g = S.make_engine(hiplib, sc, options={'sort_interval': 20, 'grid_store': grid_store, 'loose_max': loose_max, 'quad_min_units': 0 if quads else 1 << 30})
    o = S.make_engine(oracle64, sc)
    cot = S.random_cotangent(sc['N'])
    sa, ga = S.run_forward_backward(g, 30, cot)
    sb, gb = S.run_forward_backward(o, 30, {k: v.astype(np.float64) for k, v in cot.items()})
    assert (sa['used'] == sb['used']).all()
    assert S.rel_l2(sa['x'], sb['x']) <= 1e-5 and S.rel_l2(sa['v'], sb['v']) <= 1e-3 and S.rel_l2(sa['C'], sb['C']) <= 1e-3
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.cosine(ga[k], gb[k]) >= 0.999 and S.rel_l2(ga[k], gb[k]) <= 1e-2, k
    assert g.get_stats(30)['n_slow_path'] > 0
    ws = g.get_work_stats(21)
    assert ws['n_items'] == sum(ws['items_by_size'].values()) and ws['n_multi_item_workgroups'] + ws['n_single_item_blocks'] >= ws['n_occupied_blocks']
    if loose_max == 0:
        assert ws['items_by_size']['1'] + ws['items_by_size']['2-4'] > 800 and ws['n_active_blocks'] > 2 * ws['n_occupied_blocks']
        assert ws['n_quad_items'] > 800                          # (classified either way; `quads` decides whether the unit list uses them)
    else:                                                      # the droplets' blocks are loose: no item holds 12 particles or fewer
        assert ws['items_by_size']['1'] + ws['items_by_size']['2-4'] + ws['items_by_size']['5-8'] == 0 and ws['n_loose_particles'] > 1000
        assert ws['tail_start'] + ws['n_loose_particles'] == sc['N']


def test_quad_units_agree_with_pair_units(hiplib):
    """The same spray stepped with quad units (fixed-point LDS accumulators, quantum 2^-24 of a wave's largest contribution) and with
    pair units (fp64 accumulators): states and adjoints agree to fp32 rounding.  Velocities span five orders of magnitude inside
    single items, which is what a shared fixed-point scale has to survive."""
    rng = np.random.RandomState(11)
    N = 6000
    sc = S.water_block(n_grid=64, n_particles=N, lo=0.3, hi=0.7)
    sc['x'] = S.f32(rng.uniform(0.1, 0.9, (N, 3)))
    sc['x'][:3000] = S.f32(0.5 + rng.normal(0, 0.045, (3000, 3)))         # a cloud dense enough for items of 10-60 particles
    sc['v'] = S.f32(rng.normal(0, 1.0, (N, 3)) * 10.0 ** rng.uniform(-4, 0.5, (N, 1)))
    cot = S.random_cotangent(N, seed=5)
    out = {}
    for quads in (0, 1):
        g = S.make_engine(hiplib, sc, options={'sort_interval': 5, 'quad_min_units': 0 if quads else 1 << 30})
        out[quads] = S.run_forward_backward(g, 12, cot)
        ws = g.get_work_stats(0)
        assert ws['n_quad_items'] > 1000 and ws['items_by_size']['9-16'] + ws['items_by_size']['17-32'] + ws['items_by_size']['33-64'] > 30, ws
    (a, ga), (b, gb) = out[0], out[1]
    assert (a['used'] == b['used']).all()
    print('MEASURED quad vs pair units: x', np.abs(a['x'] - b['x']).max(), 'v', S.rel_l2(a['v'], b['v']), 'C', S.rel_l2(a['C'], b['C']),
          {k: S.rel_l2(ga[k], gb[k]) for k in ga})
    # measured: x one ulp (1.2e-7), v 2.7e-6, C 1.8e-5, gx 9.1e-5, gv 2.0e-5, gC 1.5e-5, gF 1.8e-5 (twelve substeps; the fast particles'
    # shell and slow-path deposits are fp32 global atomics in both runs, i.e. two pair-unit runs differ at this level too)
    assert np.abs(a['x'] - b['x']).max() <= 2.5e-7 and S.rel_l2(a['v'], b['v']) <= 1e-5 and S.rel_l2(a['C'], b['C']) <= 6e-5
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.rel_l2(ga[k], gb[k]) <= 3e-4, (k, S.rel_l2(ga[k], gb[k]))


@pytest.mark.parametrize('loose_max', [0, 2])
def test_quad_units_with_drifted_particles_in_p2g_grad(hiplib, oracle64, loose_max):
    """ADVICE r4.  k_p2g_grad's quad units keep two of their four gathered tiles where the pair units keep the stash of C and F:
    (i) a particle of a quad unit that has left its tile since the sort is redone on the global path -- which must not park C and F in
    the stash (it did: the recursion dropped NOSTASH and overwrote tile nodes other lanes were reading);  (ii) a TAIL unit that follows a
    quad unit in one workgroup does use the stash, so the workgroup has to meet at a barrier in between (loose_max 2: the blocks with
    one or two droplets have no item, their particles are the tail; 64-workgroup launches, so every workgroup walks quads, then tails).
    Droplets all over the box, every one of them drifting 1.5 cells between the two sorts, quad units forced on."""
    rng = np.random.RandomState(17)
    N = 8000
    sc = S.water_block(n_grid=64, n_particles=N, lo=0.10, hi=0.90, gravity=(0.0, 0.0, 0.0))
    sc['x'][:1500] = S.f32(0.45 + rng.normal(0, 0.05, (1500, 3)))             # a cloud with items of 5-40 particles among the one-particle ones
    sc['v'] = S.f32(rng.normal(0, 0.3, (N, 3)) + [12.0, -9.0, 4.0])
    opts = {'sort_interval': 20, 'quad_min_units': 0, 'loose_max': loose_max, 'wgrid_cap': 64, 'wgrid_cap_pgg': 64, 'wgrid_cap_g2p': 64}
    g = S.make_engine(hiplib, sc, options=opts)
    o = S.make_engine(oracle64, sc)
    cot = S.random_cotangent(N, seed=3)
    (a, ga), (b, gb) = S.run_forward_backward(g, 20, cot), S.run_forward_backward(o, 20, {k: v.astype(np.float64) for k, v in cot.items()})
    ws = g.get_work_stats(0)
    assert ws['n_quad_units'] > 100 and g.get_stats(20)['n_slow_path'] > 1000, ws
    if loose_max:
        assert ws['n_loose_particles'] > 500, ws
    assert (a['used'] == b['used']).all() and S.rel_l2(a['x'], b['x']) <= 1e-5 and S.rel_l2(a['v'], b['v']) <= 1e-3
    print('MEASURED quad units, drifted particles:', {k: (round(1 - S.cosine(ga[k], gb[k]), 9), round(S.rel_l2(ga[k], gb[k]), 7)) for k in ga})
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.cosine(ga[k], gb[k]) >= 0.99999 and S.rel_l2(ga[k], gb[k]) <= 3e-3, k


@pytest.mark.parametrize('materials', ['water', 'mixed'])
@pytest.mark.parametrize('quads', [0, 1])
def test_lane_split_small_waves_match_the_oracle(hiplib, oracle64, quads, materials):
    """Option lane_split (round 5): a wave with at most 21 / 7 particles gives every particle three / nine lanes, one per x plane /
    (x, y) column of its stencil -- in k_p2g, both passes of k_g2p_grad2 and (liquids) k_p2g_grad, in pair units (quads 0) and in quad
    units (quads 1: fixed-point tiles, the lean hand-over of the inner 6^3 nodes + the shell walk only where a particle sits on it).
    Droplets (one or two per block), loose clusters (8-20 per block) and a dense clump (full items, never split) in one scene, drifting
    1.2 cells per sort interval so that shells and a few slow-path particles take part: against the fp64 oracle, and against the same
    engine with the option off (the option is a bit per kernel: 7 = all three)."""
    rng = np.random.RandomState(23)
    n_drop, n_cl, n_dense = 3000, 2600, 1400
    N = n_drop + n_cl + n_dense
    sc = S.water_block(n_grid=64, n_particles=N, lo=0.40, hi=0.47)
    sc['x'][:n_drop] = S.f32(rng.uniform(0.1, 0.9, (n_drop, 3)))
    centres = rng.uniform(0.15, 0.85, (40, 3))
    sc['x'][n_drop:n_drop + n_cl] = S.f32(np.clip(centres[rng.randint(0, 40, n_cl)] + rng.normal(0, 0.022, (n_cl, 3)), 0.08, 0.92))
    sc['v'] = S.f32(rng.normal(0, 0.7, (N, 3)) + [9.0, -6.0, 3.0])
    if materials == 'mixed':                                   # the SVD kernels (k_p2g<., true>; k_p2g_grad<true> keeps one lane per particle)
        sc['mat'] = np.array([S.WATER, S.ELASTIC, S.ICECREAM], np.int32)[rng.randint(0, 3, N)]
        sc['F'] = S.f32(np.eye(3)[None] + rng.normal(0, 1.0, (N, 3, 3)) * np.where(sc['mat'] == S.ICECREAM, 0.002, 0.03)[:, None, None])     # (as scenarios.mixed_materials)
    cot = S.random_cotangent(N, seed=7)
    out = {}
    for split in (7, 0):
        g = S.make_engine(hiplib, sc, options={'sort_interval': 10, 'quad_min_units': 0 if quads else 1 << 30, 'lane_split': split})
        out[split] = S.run_forward_backward(g, 12, cot)
        ws = g.get_work_stats(10)
        if split:
            assert ws['n_split9_waves'] > 500 and ws['n_split3_waves'] > 30 and ws['items_by_size']['65-128'] > 5, ws
            assert (ws['n_quad_units'] > 100) == bool(quads), ws
            assert g.get_stats(12)['n_slow_path'] > 0
        else:
            assert ws['n_split9_waves'] == 0 and ws['n_split3_waves'] == 0
    o = S.make_engine(oracle64, sc)
    b, gb = S.run_forward_backward(o, 12, {k: v.astype(np.float64) for k, v in cot.items()})
    (a, ga), (c, gc) = out[7], out[0]
    assert (a['used'] == b['used']).all()
    print(f'MEASURED lane_split[{materials}, quads={quads}]: vs oracle x', np.abs(a['x'] - b['x']).max(), 'v', S.rel_l2(a['v'], b['v']),
          {k: round(S.rel_l2(ga[k], gb[k]), 8) for k in ga}, '| on vs off x', np.abs(a['x'] - c['x']).max(), {k: round(S.rel_l2(ga[k], gc[k]), 8) for k in ga})
    tol_g = 3e-3 if materials == 'water' else 2e-2
    assert np.abs(a['x'] - b['x']).max() <= 5e-6 and S.rel_l2(a['v'], b['v']) <= 1e-3 and S.rel_l2(a['F'], b['F']) <= 1e-5
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert np.isfinite(ga[k]).all() and S.cosine(ga[k], gb[k]) >= 0.999 and S.rel_l2(ga[k], gb[k]) <= tol_g, (k, S.rel_l2(ga[k], gb[k]))
        # the option changes the order of the fp32 adds inside a run and nothing else: no further from the unsplit engine than fp32 rounding
        assert S.rel_l2(ga[k], gc[k]) <= max(3e-4, 2.0 * S.rel_l2(gc[k], gb[k])), (k, S.rel_l2(ga[k], gc[k]))
    assert np.abs(a['x'] - c['x']).max() <= 1e-6


def test_dense_scene_with_small_items(hiplib, oracle64):
    """item_max = 64 on a box filled with water: every block holds 3-4 items, i.e. two pairs -- more pairs than half the items
    (the order's pair list used to be sized items / 2 and overflowed here).  Forward and backward against the oracle."""
    sc = S.water_block(n_grid=32, n_particles=77000, lo=0.06, hi=0.94, seed=4)
    opts = {'item_max': 64, 'sort_interval': 2}
    g = S.make_engine(hiplib, sc, options=opts)
    o = S.make_engine(oracle64, sc)
    cot = S.random_cotangent(sc['N'], seed=8)
    (a, ga), (b, gb) = S.run_forward_backward(g, 4, cot), S.run_forward_backward(o, 4, cot)
    ws = g.get_work_stats(0)
    assert ws['n_multi_item_workgroups'] > ws['n_items'] // 2 - 40 and ws['n_items'] > 1200, ws
    assert np.abs(a['x'] - b['x']).max() <= 5e-6 and (a['used'] == b['used']).all()
    print('MEASURED dense small items:', {k: (round(1 - S.cosine(ga[k], gb[k]), 8), round(S.rel_l2(ga[k], gb[k]), 6)) for k in ga})
    for k in ('gx', 'gv', 'gC', 'gF'):                        # measured: gC (values ~1e-6) relL2 3.8e-3, gF 1 - cos 3.4e-5 (the stiff volume term in fp32)
        tol_cos, tol_rel = {'gC': (0.99999, 1.2e-2), 'gF': (0.9999, 2.5e-2)}.get(k, (0.99999, 3e-3))
        assert S.cosine(ga[k], gb[k]) >= tol_cos and S.rel_l2(ga[k], gb[k]) <= tol_rel, k


def test_forward_is_independent_of_sort_interval(hiplib):
    sc = S.water_block(n_grid=32, n_particles=6000)
    ref = S.run_forward(S.make_engine(hiplib, sc, options={'sort_interval': 0}), 25)
    for K in (1, 7, 10):
        got = S.run_forward(S.make_engine(hiplib, sc, options={'sort_interval': K}), 25)
        assert (got['used'] == ref['used']).all()
        assert S.rel_l2(got['x'], ref['x']) <= 1e-6 and S.rel_l2(got['v'], ref['v']) <= 1e-3, K


def test_large_window_does_not_race_uploads(hiplib):
    """Regression: with a multi-GB frame window the zero-fill of the engine's buffers is still running on the engine's
    (non-blocking) stream when fe_create uploads the identity particle table; a null-stream copy used to be overwritten by
    the late memset, which only showed up at LatteArt 128^3 sizes.  K=0 keeps the identity table in use."""
    sc = S.water_block(n_grid=32, n_particles=8192)
    small = S.make_engine(hiplib, sc, max_substeps_local=10, options=dict(sort_interval=0))
    a = S.run_forward(small, 5)
    small.close()
    big = S.make_engine(hiplib, sc, max_substeps_local=24000, options=dict(sort_interval=0))     # ~20 GB of frames
    b = S.run_forward(big, 5)
    st = big.get_stats(5)
    big.close()
    assert st['n_used'] == 8192 and st['bytes_state'] > 15e9
    assert (a['used'] == b['used']).all()
    assert np.abs(a['x'] - b['x']).max() <= 1e-6          # slow path float atomics: order-dependent rounding only


def test_roundtrip_and_frame_ops(hiplib):
    sc = S.mixed_materials(n_particles=777)          # ragged: not a multiple of the wave size
    eng = S.make_engine(hiplib, sc, max_substeps_local=4)
    st = S.get_state(eng, 0)
    assert (st['x'] == sc['x']).all() and (st['v'] == sc['v']).all() and (st['C'] == sc['C']).all() and (st['F'] == sc['F']).all()
    assert (st['used'] == sc['used']).all()
    eng.copy_frame(0, 3)
    st3 = S.get_state(eng, 3)
    assert all((st3[k] == st[k]).all() for k in st)
    cot = S.random_cotangent(sc['N'])
    eng.reset_grad()
    eng.add_grad(2, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
    gx, gv, gC, gF = eng.get_grad(2)
    assert (gx == cot['gx']).all() and (gF == cot['gF']).all()
    stats = eng.get_stats(0)
    assert stats['n_used'] == int(sc['used'].sum()) and stats['n_cells_touched'] > 0
    from fluidlab_amd._capi import FeEngineError
    with pytest.raises(FeEngineError, match='out of range'):
        eng.substep(4, 4, 0)


def test_device_pointer_frame_io(hiplib):
    """fe_get_frame_dev / fe_set_frame_dev (readframe/setframe with ckpt_dest='gpu', mpm:555-587): the state moves between
    frames and caller-owned HBM arrays bit-exactly, in particle-id order, also after the engine has sorted its own storage."""
    import torch
    sc = S.mixed_materials(n_particles=777)
    eng = S.make_engine(hiplib, sc, max_substeps_local=8)
    S.run_forward(eng, 5)
    host = S.get_state(eng, 5)
    N = sc['N']
    dev = torch.device('cuda', 0)
    d = dict(x=torch.full((N, 3), -7.0, device=dev), v=torch.zeros((N, 3), device=dev), C_=torch.zeros((N, 3, 3), device=dev),
             F=torch.zeros((N, 3, 3), device=dev), used=torch.full((N,), -1, dtype=torch.int32, device=dev))
    eng.get_frame_dev(5, **d)
    torch.cuda.synchronize()
    for k, hk in (('x', 'x'), ('v', 'v'), ('C_', 'C'), ('F', 'F'), ('used', 'used')):
        assert (d[k].cpu().numpy() == host[hk]).all(), k
    eng.set_frame_dev(0, **d)
    st0 = S.get_state(eng, 0)
    assert all((st0[k] == host[k]).all() for k in host)
    a = S.run_forward(eng, 3)                    # stepping on from the device-restored frame == stepping on from the host-restored one
    eng.set_frame(0, host['x'], host['v'], host['C'], host['F'], host['used'])
    b = S.run_forward(eng, 3)
    # (the two runs re-sort from different in-cell orders, so sums round differently: tolerance, not bit equality)
    for k in a:
        err = np.abs(a[k].astype(np.float64) - b[k]).max()
        assert err <= 2e-5 * (1.0 + np.abs(b[k]).max()), (k, err)
    eng.get_frame_dev(3, x=d['x'])               # partial reads leave the other arrays alone
    assert (d['x'].cpu().numpy() == b['x']).all() and (d['v'].cpu().numpy() == host['v']).all()


def test_empty_and_all_unused(hiplib):
    sc = S.water_block(n_grid=8, n_particles=64)
    sc['used'] = np.zeros(64, np.int32)
    eng = S.make_engine(hiplib, sc, max_substeps_local=4)
    a = S.run_forward(eng, 3)
    assert (a['x'] == sc['x']).all() and (a['used'] == 0).all()
    eng.sync()


def test_stats_match_oracle(hiplib, oracle64):
    sc = S.water_block(n_grid=32, n_particles=5000)
    a = S.make_engine(hiplib, sc).get_stats(0)
    b = S.make_engine(oracle64, sc).get_stats(0)
    for k in ('n_used', 'n_cells_touched', 'n_blocks_active'):
        assert a[k] == b[k], k


def test_batched_envs_match_individual_runs(hiplib):
    """fe_step_batch / fe_step_grad_batch: three environments (different particle sets, one of them with a different particle
    count) stepped in lockstep through shared launches give what each gives on its own."""
    scenes = [S.water_block(n_grid=32, n_particles=n, seed=sd) for n, sd in ((6000, 0), (6000, 1), (4500, 2))]
    for k, sc in enumerate(scenes):
        sc['v'] = S.f32(np.random.RandomState(10 + k).normal(0, 0.4, (sc['N'], 3)))
    L = 24
    cots = [S.random_cotangent(sc['N'], seed=20 + k) for k, sc in enumerate(scenes)]

    def finish(eng, cot):
        st = S.get_state(eng, L)
        return st, eng.get_grad(0)

    solo = []
    for sc, cot in zip(scenes, cots):
        eng = S.make_engine(hiplib, sc, max_substeps_local=L, options={'fuse_g2p': 1, 'fuse_bwd': 1, 'fuse_grid': 0})
        eng.step(0, 0, L, 0)
        eng.reset_grad(); eng.add_grad(L, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
        eng.step_grad(0, 0, L, 0)
        solo.append(finish(eng, cot))
        eng.close()
    engs = [S.make_engine(hiplib, sc, max_substeps_local=L, options={'fuse_g2p': 1, 'fuse_bwd': 1, 'fuse_grid': 0}) for sc in scenes]      # (the launch counts asserted below are the fused launches': pinned against FE_* overrides)
    engs[0].profile_enable(True)                           # (a batch's launches are the leader's)
    type(engs[0]).step_batch(engs, 0, 0, L, 0)
    for eng, cot in zip(engs, cots):
        eng.reset_grad(); eng.add_grad(L, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
    type(engs[0]).step_grad_batch(engs, 0, 0, L, 0)
    prof = engs[0].profile_read()
    engs[0].profile_enable(False)
    # the batch runs the fused launches too (k_g2p_p2g_b, k_pgg_g2pg_b): every substep but the heads and tails of the sort intervals (K = 10: 0, 10, 20) and of the call
    assert prof['g2p_p2g'][1] == L - 3 and prof['p2g'][1] == 3 and prof['pgg_g2pg'][1] == L - 3 and prof['p2g_grad'][1] == 3 and prof['grid_op'][1] == L, prof
    for eng, cot, (st0, g0) in zip(engs, cots, solo):
        st, g = finish(eng, cot)
        assert np.array_equal(st['used'], st0['used'])
        # (not bit for bit: the rank of a particle inside its cell comes from an LDS atomic of the sort, so two runs of the same
        # engine already order the fp32 sums differently; measured x 1e-7, C 2.9e-6)
        for k, tol in (('x', 1e-6), ('v', 1e-5), ('C', 3e-5), ('F', 1e-6)):
            assert S.rel_l2(st[k], st0[k]) <= tol, k
        for a, b in zip(g, g0):
            assert S.rel_l2(a, b) <= 1e-4
        eng.close()


@pytest.mark.parametrize('pack_units,quad_fit', [(1, 40), (1, 90), (2, 4), (0, 40)])
def test_packed_unit_list_matches_the_oracle(hiplib, oracle64, pack_units, quad_fit):
    """Round 4's scatter list: the leftover item of a block with an odd number of items no longer sits alone in a workgroup -- it
    pairs with another block's item or goes into a quad unit with its own fixed-point tile -- and exactly as many small items go
    four to a workgroup as it takes to fit `quad_fit` workgroups (one resident round on the chip; small here).  A dense block (three
    to five items per 4^3 block, odd counts among them) in a cloud of droplets (single-item blocks of every size), forward and
    backward against the fp64 oracle; (1, 90) fits with a few quads, (1, 40) cannot fit and stays pairs, (2, 4) packs regardless."""
    rng = np.random.RandomState(21)
    core = []                                                                      # 27 blocks of 140 ... 600 particles: two to five items each
    for bi in range(3):
        for bj in range(3):
            for bk in range(3):
                lo = (4 * np.array([3 + bi, 3 + bj, 3 + bk]) + 0.5) / 32
                core.append(lo + rng.uniform(0.02, 0.98, (int(rng.randint(140, 600)), 3)) * (4 / 32))
    core = np.concatenate(core)
    n_drop = 1400
    drops = np.concatenate([c + rng.uniform(-r, r, (k, 3)) for c, r, k in
                            zip(rng.uniform(0.15, 0.85, (70, 3)), rng.choice([0.01, 0.02, 0.04], 70), [n_drop // 70] * 70)])
    x = S.f32(np.clip(np.concatenate([core, drops]), 0.08, 0.92))
    N = len(x)
    sc = dict(S.water_block(n_grid=32, n_particles=N, seed=3), x=x, v=S.f32(rng.normal(0, 0.4, (N, 3))))
    opts = {'sort_interval': 4, 'pack_units': pack_units, 'quad_fit': quad_fit, 'item_max': 128, 'quad_min_units': 1400}      # (the threshold pinned: FE_QUAD_MIN_UNITS=0 runs of the suite force quads elsewhere)
    g = S.make_engine(hiplib, sc, options=opts)
    o = S.make_engine(oracle64, sc)
    cot = S.random_cotangent(N, seed=4)
    sa, ga = S.run_forward_backward(g, 9, cot)
    sb, gb = S.run_forward_backward(o, 9, {k: v.astype(np.float64) for k, v in cot.items()})
    ws = g.get_work_stats(8)
    print('MEASURED packed unit list', (pack_units, quad_fit), {k: ws[k] for k in ('n_items', 'n_multi_item_workgroups', 'n_leftover_items', 'n_single_item_blocks', 'n_quad_items', 'n_quad_units', 'n_scatter_units', 'n_gather_units', 'packed')})
    assert ws['n_leftover_items'] >= 8 and ws['n_quad_items'] > 30
    if pack_units == 0:
        assert not ws['packed'] and ws['n_quad_units'] == 0 and ws['n_scatter_units'] == ws['n_gather_units']
    elif (pack_units, quad_fit) == (1, 90):
        assert ws['packed'] and 0 < ws['n_quad_units'] and ws['n_scatter_units'] <= 90 < ws['n_gather_units']
    elif pack_units == 2:
        assert ws['packed'] and ws['n_scatter_units'] < ws['n_gather_units']
    assert np.abs(sa['x'] - sb['x']).max() <= 2e-6
    assert S.rel_l2(sa['v'], sb['v']) <= 1e-4
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.cosine(ga[k], gb[k]) >= 0.99999 and S.rel_l2(ga[k], gb[k]) <= 1e-3, (k, S.rel_l2(ga[k], gb[k]))


@pytest.mark.parametrize('ggrid_cap', [1, 2, 5])
@pytest.mark.parametrize('grid_store', [0, 1])
def test_grid_kernels_long_list_road(hiplib, oracle64, ggrid_cap, grid_store):
    """`k_grid` / `k_grid_grad` on a list longer than four times their workgroups (option `ggrid_cap` shrinks the launch; by default only a
    spread-out scene at 128^3 gets there): candidates eight at a time per wave, several rounds of them, the marked ones shared out among
    the workgroup's four waves; droplets all over the box so that most entries of the list are unmarked, a few fast ones for dirty
    blocks; backward from the stored grid and from the recompute."""
    rng = np.random.RandomState(31)
    N = 1800
    c = rng.uniform(0.25, 0.75, (60, 3))
    x = S.f32(np.clip(np.repeat(c, N // 60, 0) + rng.uniform(-0.02, 0.02, (N, 3)), 0.22, 0.78))
    sc = dict(S.water_block(n_grid=32, n_particles=N, seed=5), x=x, v=S.f32(rng.normal(0, 0.5, (N, 3))))
    sc['v'][::37] *= 25.0                                                         # a few particles leave their tiles between sorts
    g = S.make_engine(hiplib, sc, options={'sort_interval': 6, 'ggrid_cap': ggrid_cap, 'grid_store': grid_store})
    o = S.make_engine(oracle64, sc)
    cot = S.random_cotangent(N, seed=6)
    sa, ga = S.run_forward_backward(g, 9, cot)
    sb, gb = S.run_forward_backward(o, 9, {k: v.astype(np.float64) for k, v in cot.items()})
    ws = g.get_work_stats(8)
    assert ws['n_active_blocks'] > 8 * 4 * ggrid_cap                              # several rounds of eight candidates per wave
    assert np.abs(sa['x'] - sb['x']).max() <= 2e-6 and S.rel_l2(sa['v'], sb['v']) <= 1e-4
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.cosine(ga[k], gb[k]) >= 0.99999 and S.rel_l2(ga[k], gb[k]) <= 1e-3, (k, S.rel_l2(ga[k], gb[k]))


@pytest.mark.parametrize('liquid_only', [True, False])
def test_adjoint_crosses_sort_boundaries_inside_p2g_grad(hiplib, oracle64, liquid_only):
    """`fold_reorder` (round 4): inside one fe_step_grad call the adjoint of a frame that a sort follows is written by k_p2g_grad in the
    order of the NEXT substep (slot_of_pid of that order, through the third adjoint buffer) and the reorder pass is skipped.  The same
    reverse sweep three ways -- one call per substep (nothing folded), one ranged call with the option off, one with it on -- must agree
    to rounding (the arithmetic per particle is the same; only where the result lands differs), and with the fp64 oracle.  Unused
    slots (their adjoint passes straight through, mpm:551), a sort every 3 substeps, particles fast enough to change blocks."""
    rng = np.random.RandomState(17)
    N = 6000
    x = S.f32(np.clip(np.repeat(rng.uniform(0.3, 0.7, (40, 3)), N // 40, 0) + rng.uniform(-0.05, 0.05, (N, 3)), 0.1, 0.9))
    sc = dict(S.water_block(n_grid=32, n_particles=N, seed=9), x=x, v=S.f32(rng.normal(0, 3.0, (N, 3))),
              used=(rng.rand(N) > 0.1).astype(np.int32))
    sc['v'][::7] *= 8.0
    if not liquid_only:
        sc['mat'] = np.array([S.WATER, S.ELASTIC, S.ICECREAM], np.int32)[rng.randint(0, 3, N)]
        sc['F'] = S.f32(np.eye(3)[None] + rng.normal(0, 1.0, (N, 3, 3)) * np.where((sc['mat'] == S.ICECREAM)[:, None, None], 0.002, 0.02))
    cot = S.random_cotangent(N, seed=8)
    n_sub = 11
    runs = {}
    for name, opts, ranged in (('per substep', {}, False), ('ranged, off', {'fold_reorder': 0}, True), ('ranged, on', {'fold_reorder': 1}, True)):
        g = S.make_engine(hiplib, sc, options=dict(opts, sort_interval=3))
        runs[name] = S.run_forward_backward(g, n_sub, cot, ranged=ranged)
        if name == 'ranged, on':
            assert g.get_option('fold_reorder') == 1.0
        g.close()
    o = S.make_engine(oracle64, sc)
    sb, gb = S.run_forward_backward(o, n_sub, {k: v.astype(np.float64) for k, v in cot.items()})
    ref = runs['per substep'][1]
    for name in ('ranged, off', 'ranged, on'):
        for k in ('gx', 'gv', 'gC', 'gF'):
            assert S.rel_l2(runs[name][1][k], ref[k]) <= 2e-5, (name, k, S.rel_l2(runs[name][1][k], ref[k]))      # (the fast particles' slow path adds with global fp32 atomics: order noise, measured 1.5e-6 through the SVD adjoint)
    tol_cos, tol_rel = (0.99999, 3e-3) if liquid_only else (0.999, 8e-2)
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.cosine(runs['ranged, on'][1][k], gb[k]) >= tol_cos and S.rel_l2(runs['ranged, on'][1][k], gb[k]) <= tol_rel, k


@pytest.mark.parametrize('n_grid', [12, 20, 28])
def test_grid_sizes_with_an_odd_number_of_blocks_per_axis(hiplib, oracle64, n_grid):
    """n / 4 blocks per axis, odd: the block count (27, 125, 343) is no multiple of the 16 blocks a wave of the sort's cell scan takes
    (k_sort_blk_partial), nor of the 4 a thread of the block scan does -- the guards at the end of the block range, forward and backward
    against the fp64 oracle with a sort every 3 substeps."""
    rng = np.random.RandomState(n_grid)
    N = 2500
    sc = dict(S.water_block(n_grid=n_grid, n_particles=N, seed=n_grid, lo=0.15, hi=0.85), v=S.f32(rng.normal(0, 1.0, (N, 3))))
    g = S.make_engine(hiplib, sc, options={'sort_interval': 3})
    o = S.make_engine(oracle64, sc)
    cot = S.random_cotangent(N, seed=2)
    sa, ga = S.run_forward_backward(g, 8, cot, ranged=True)
    sb, gb = S.run_forward_backward(o, 8, {k: v.astype(np.float64) for k, v in cot.items()})
    ws = g.get_work_stats(7)
    assert ws['n_items'] > 0 and ws['n_active_blocks'] > 0
    assert np.abs(sa['x'] - sb['x']).max() <= 2e-6 and S.rel_l2(sa['v'], sb['v']) <= 1e-4
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.cosine(ga[k], gb[k]) >= 0.99999 and S.rel_l2(ga[k], gb[k]) <= 1e-3, (k, S.rel_l2(ga[k], gb[k]))


def test_compact_F_of_liquids_changes_nothing(hiplib, oracle64):
    """Option compact_F (round 5): the SVD-free kernels keep an inviscid liquid's F as the one number it is (F = c I from the first substep on,
    mpm:359) -- k_p2g reads / writes 4 bytes of the 36, k_p2g_grad likewise and carries the adjoint's trace inside a ranged call -- and every API
    call that hands F or its adjoint out expands the planes first.  The same trajectory with the option on and off: states of an intermediate and
    of the last frame (F included) and all four adjoints of frame 0, through per-substep calls, a ranged call, and two ranged calls with a loss
    seeded on the frame between them; a frame-0 F that is NOT isotropic, unused pool slots, an injected particle's F.  The option only changes
    which bytes move: results agree to the order noise of the slow path's fp32 atomics (measured: identical), and with the fp64 oracle."""
    rng = np.random.RandomState(5)
    N = 5000
    sc = S.water_block(n_grid=32, n_particles=N, seed=3, lo=0.3, hi=0.6)
    sc['v'] = S.f32(rng.normal(0, 1.0, (N, 3)))
    sc['F'] = S.f32(np.eye(3)[None] + rng.normal(0, 0.02, (N, 3, 3)))            # general at frame 0: only its determinant matters to a liquid
    sc['used'] = (rng.rand(N) > 0.1).astype(np.int32)
    cot, cot_mid = S.random_cotangent(N, seed=2), S.random_cotangent(N, seed=4)
    n_sub, mid = 12, 7

    def run(lib, opts, mode):
        g = S.make_engine(lib, sc, options=opts)
        S.run_forward(g, n_sub)
        st_mid, st_end = S.get_state(g, mid), S.get_state(g, n_sub)
        g.reset_grad()
        g.add_grad(n_sub, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
        if mode == 'per substep':
            for f in reversed(range(mid, n_sub)):
                g.substep_grad(f, f, 0)
        else:
            g.step_grad(mid, mid, n_sub - mid, 0)
        g.add_grad(mid, cot_mid['gx'], cot_mid['gv'], cot_mid['gC'], cot_mid['gF'])       # a loss on the frame between the two calls: F's adjoint included
        if mode == 'per substep':
            for f in reversed(range(mid)):
                g.substep_grad(f, f, 0)
        else:
            g.step_grad(0, 0, mid, 0)
        gx, gv, gC, gF = g.get_grad(0)
        g.close()
        return st_mid, st_end, dict(gx=gx, gv=gv, gC=gC, gF=gF)

    base = {'sort_interval': 4}
    ref = run(hiplib, dict(base, compact_F=0), 'ranged')
    worst = 0.0
    for mode in ('per substep', 'ranged'):
        got = run(hiplib, dict(base, compact_F=1), mode)
        for a, b in ((got[0], ref[0]), (got[1], ref[1])):
            assert (a['used'] == b['used']).all()
            u = b['used'] > 0
            for k in 'xvCF':
                m = u if k == 'F' else slice(None)
                worst = max(worst, float(np.abs(a[k][m] - b[k][m]).max()))
                assert np.abs(a[k][m] - b[k][m]).max() <= 1e-5 * max(1.0, np.abs(b[k][m]).max()), (mode, k)      # (order noise of the shell's / slow path's fp32 atomics: two runs of ONE build differ by as much)
            # the one observable difference: an UNUSED liquid particle's general F is carried as det(F)^(1/3) I -- all that a liquid ever consumes of it
            assert np.abs(np.linalg.det(a['F'][~u].astype(np.float64)) - np.linalg.det(b['F'][~u].astype(np.float64))).max() <= 1e-6
        for k in ('gx', 'gv', 'gC', 'gF'):
            worst = max(worst, S.rel_l2(got[2][k], ref[2][k]))
            assert S.rel_l2(got[2][k], ref[2][k]) <= 2e-5, (mode, k, S.rel_l2(got[2][k], ref[2][k]))
    used_mid = ref[0]['used'] > 0
    Fm = ref[0]['F'][used_mid]
    assert np.abs(Fm - Fm[:, :1, :1] * np.eye(3)).max() == 0.0                  # from the first substep on a liquid's F is c I
    ob = run(oracle64, {}, 'per substep')
    print('MEASURED compact_F on vs off: largest difference', worst, '| vs fp64 oracle', {k: round(S.rel_l2(ref[2][k], ob[2][k]), 8) for k in ref[2]})
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert S.cosine(ref[2][k], ob[2][k]) >= 0.99999 and S.rel_l2(ref[2][k], ob[2][k]) <= 3e-3, k
    assert np.abs(ref[1]['x'] - ob[1]['x']).max() <= 5e-6


def test_compact_F_with_an_injector(hiplib, oracle64):
    """... and through a LatteArt-like pass: pool particles wait unused (their F = I is carried, compactly, until the Injector uses them), milk is
    injected, the loss is evaluated every step between the ranged backward calls."""
    sc = S.latte_mini()
    a = S.run_latte(hiplib, sc, options={'compact_F': 1, 'sort_interval': 3})
    b = S.run_latte(hiplib, sc, options={'compact_F': 0, 'sort_interval': 3})
    o = S.run_latte(oracle64, sc)
    assert (a['final']['used'] == b['final']['used']).all()
    for k in 'xvCF':
        assert np.abs(a['final'][k] - b['final'][k]).max() <= 1e-5 * max(1.0, np.abs(b['final'][k]).max()), k
    assert S.rel_l2(a['action_grad'], b['action_grad']) <= 2e-5 and S.rel_l2(a['step_loss'], b['step_loss']) <= 1e-6
    assert S.cosine(a['action_grad'], o['action_grad']) >= 0.999999 and S.rel_l2(a['action_grad'], o['action_grad']) <= 1e-4


def _well_conditioned(frames, dt=2e-4, gap=1e-2):
    """The particles whose SVD adjoint is well conditioned along a trajectory: backward_svd (mpm:272-292) divides by sigma_j^2 - sigma_i^2 of F_tmp = (I + dt C) F --
    here those whose squared singular values stay at least `gap` apart in every frame of `frames` (and every particle of a material that takes no SVD is told
    apart by the caller).  Their adjoints carry no amplified rounding noise: they are held to a MAX norm (a percentile passes a handful of wrong particles)."""
    ok = None
    for fr in frames:
        Ft = (np.eye(3)[None] + dt * fr['C'].astype(np.float64)) @ fr['F'].astype(np.float64)
        s2 = np.linalg.svd(Ft, compute_uv=False) ** 2
        g = np.minimum(np.minimum(np.abs(s2[:, 0] - s2[:, 1]), np.abs(s2[:, 1] - s2[:, 2])), np.abs(s2[:, 0] - s2[:, 2]))
        ok = (g >= gap) if ok is None else ok & (g >= gap)
    return ok


def _max_off(a, b, sel):
    """largest per-particle |a - b| among the particles `sel`, relative to the field's RMS over all particles"""
    a = np.asarray(a, np.float64).reshape(len(a), -1); b = np.asarray(b, np.float64).reshape(len(b), -1)
    rms = np.sqrt((b ** 2).sum(1).mean())
    return float(np.sqrt(((a - b) ** 2).sum(1))[sel].max() / rms) if sel.any() else 0.0


def _pct_off(a, b, q):
    """q-th percentile over the particles of |a - b| relative to the field's RMS: the SVD materials' adjoints amplify rounding noise without bound in the few
    particles whose singular values nearly coincide (backward_svd divides by their difference), which an L2 norm over all particles is then a measure of; a defect
    in a kernel shows in the median"""
    a = np.asarray(a, np.float64).reshape(len(a), -1); b = np.asarray(b, np.float64).reshape(len(b), -1)
    rms = np.sqrt((b ** 2).sum(1).mean())
    return float(np.percentile(np.sqrt(((a - b) ** 2).sum(1)), q) / rms)


def _droplet_scene(materials, seed=23):
    """(test_lane_split_small_waves_match_the_oracle's scene) droplets, loose clusters and a dense clump in a 64^3 box, drifting ~1.2 cells per sort interval"""
    rng = np.random.RandomState(seed)
    n_drop, n_cl, n_dense = 3000, 2600, 1400
    N = n_drop + n_cl + n_dense
    sc = S.water_block(n_grid=64, n_particles=N, lo=0.40, hi=0.47)
    sc['x'][:n_drop] = S.f32(rng.uniform(0.1, 0.9, (n_drop, 3)))
    centres = rng.uniform(0.15, 0.85, (40, 3))
    sc['x'][n_drop:n_drop + n_cl] = S.f32(np.clip(centres[rng.randint(0, 40, n_cl)] + rng.normal(0, 0.022, (n_cl, 3)), 0.08, 0.92))
    sc['v'] = S.f32(rng.normal(0, 0.7, (N, 3)) + [9.0, -6.0, 3.0])
    if materials == 'mixed':
        sc['mat'] = np.array([S.WATER, S.ELASTIC, S.ICECREAM], np.int32)[rng.randint(0, 3, N)]
        sc['F'] = S.f32(np.eye(3)[None] + rng.normal(0, 1.0, (N, 3, 3)) * np.where(sc['mat'] == S.ICECREAM, 0.002, 0.03)[:, None, None])
    sc['used'] = (rng.rand(N) > 0.05).astype(np.int32)
    return sc


@pytest.mark.parametrize('scene,opts', [('block', {}), ('block', {'compact_F': 0}), ('droplets-water', {'quad_min_units': 0}), ('droplets-water', {'quad_min_units': 1 << 30, 'loose_max': 3}),
                                        ('droplets-mixed', {'quad_min_units': 0}), ('droplets-mixed', {'quad_min_units': 1 << 30, 'lane_split': 0, 'loose_max': 3})])
def test_fused_g2p_p2g_launch_matches_separate_launches(hiplib, oracle64, scene, opts):
    """Option fuse_g2p (round 5): inside a fe_step call the g2p of substep f - 1 runs at the head of substep f's p2g launch (k_g2p_p2g) -- same unit list, the
    gathered x' v' C' written to frame f and kept in registers, v_out staged in the bytes of the unit's scatter tile.  Same trajectory as three launches per
    substep: EVERY frame of the window compared (the fused launch is what writes frames 1 ... n - 1, k_g2p only the frames in front of a sort and the last one),
    then the reverse sweep over the stored frames, against the unfused engine and the fp64 oracle.  Pair units, quad units (fixed-point tiles: the gather tile
    aliases four of them), split waves, tail units of loose blocks (global gather + global scatter), unused slots, the SVD materials, compact and full F."""
    if scene == 'block':
        rng = np.random.RandomState(5)
        N = 6000
        sc = S.water_block(n_grid=32, n_particles=N, seed=3, lo=0.3, hi=0.6)
        sc['v'] = S.f32(rng.normal(0, 1.0, (N, 3)) + [2.0, -3.0, 1.0])
        sc['used'] = (rng.rand(N) > 0.1).astype(np.int32)
        K = 4
    else:
        sc = _droplet_scene(scene.split('-')[1])
        N = len(sc['used'])
        K = 10
    n_sub = 13
    cot = S.random_cotangent(N, seed=7)

    def run(lib, o, ranged):
        g = S.make_engine(lib, sc, options=o)
        if ranged:
            g.step(0, 0, 6, 0)                                   # two calls: the boundary between them is an unfused g2p + p2g
            g.step(6, 6, n_sub - 6, 0)
        else:
            for f in range(n_sub):
                g.substep(f, f, 0)
        frames = [S.get_state(g, f) for f in range(1, n_sub + 1)]
        stats = g.get_stats(n_sub) if lib is hiplib else None
        ws = g.get_work_stats(K) if lib is hiplib else None
        g.reset_grad()
        g.add_grad(n_sub, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
        g.step_grad(0, 0, n_sub, 0)
        gx, gv, gC, gF = g.get_grad(0)
        g.close()
        return frames, dict(gx=gx, gv=gv, gC=gC, gF=gF), stats, ws

    base = dict({'sort_interval': K}, **opts)
    fa, ga, st, ws = run(hiplib, dict(base, fuse_g2p=1), True)
    fb, gb, _, _ = run(hiplib, dict(base, fuse_g2p=0), True)
    fc, gc, _, _ = run(hiplib, dict(base, fuse_g2p=0), True)           # the same engine once more: its own run-to-run noise
    fo, go, _, _ = run(oracle64, {}, False)
    if scene != 'block':
        assert st['n_slow_path'] > 0, st
        assert (ws['n_quad_units'] > 100) == (opts.get('quad_min_units') == 0), ws
    # The engine is not bit-reproducible from run to run: the counting sort ranks the particles of a cell with returning atomics, so the lanes of a run of
    # equal stencils change places, the fp32 segmented sums of the scatter change their order, and v' differs in the last bit from the first substep on
    # (C', a difference of nearly equal sums times 4 / dx, by 2e-5 of its range).  These violent scenes amplify that from substep to substep.  The fused
    # launch has to stay within that noise: no farther from the separate launches than those are from themselves (x4, + a few ulp).
    worst = {k: 0.0 for k in 'xvCF'}
    noise = {k: 0.0 for k in 'xvCF'}
    general = scene.endswith('mixed')
    for f, (a, b, c) in enumerate(zip(fa, fb, fc), start=1):
        assert (a['used'] == b['used']).all(), f
        u = b['used'] > 0
        for k in 'xvCF':
            scale = max(1.0, float(np.abs(b[k][u]).max()))
            d, nz = float(np.abs(a[k][u] - b[k][u]).max()) / scale, float(np.abs(c[k][u] - b[k][u]).max()) / scale
            worst[k] = max(worst[k], d); noise[k] = max(noise[k], nz)
            assert d <= 4.0 * nz + {'x': 5e-7, 'v': 4e-6, 'C': 8e-5, 'F': 4e-6}[k], (f, k, d, nz)       # (+ a few ulp of the field's range: one pair of runs is a small sample of the noise)
            assert (a[k][~u] == b[k][~u]).all() or k == 'F', (f, k)                # unused slots are carried, bit for bit
    print(f'MEASURED fuse_g2p[{scene}, {opts}]: fused vs separate launches, largest relative state difference over {n_sub} frames', {k: float(f'{v:.2g}') for k, v in worst.items()},
          'separate vs separate', {k: float(f'{v:.2g}') for k, v in noise.items()}, '| adjoints', {k: round(S.rel_l2(ga[k], gb[k]), 8) for k in ga}, 'separate vs separate',
          {k: round(S.rel_l2(gc[k], gb[k]), 8) for k in ga}, '| vs fp64 oracle x', np.abs(fa[-1]['x'] - fo[-1]['x']).max(), {k: round(S.rel_l2(ga[k], go[k]), 8) for k in ga})
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert np.isfinite(ga[k]).all()
        # (the SVD materials' adjoints amplify the state noise with a heavy tail -- one pair of runs is no bound on the next: there, by the median and the 95th percentile over the particles, _pct_off)
        if general:
            assert _pct_off(ga[k], gb[k], 50) <= 4.0 * _pct_off(gc[k], gb[k], 50) + 1e-5 and _pct_off(ga[k], gb[k], 95) <= 4.0 * _pct_off(gc[k], gb[k], 95) + 1e-3, \
                (k, [_pct_off(ga[k], gb[k], q) for q in (50, 95)], [_pct_off(gc[k], gb[k], q) for q in (50, 95)])
        else:
            assert S.rel_l2(ga[k], gb[k]) <= 4.0 * S.rel_l2(gc[k], gb[k]) + 2e-5, (k, S.rel_l2(ga[k], gb[k]), S.rel_l2(gc[k], gb[k]))
        assert S.cosine(ga[k], go[k]) >= 0.999 and S.rel_l2(ga[k], go[k]) <= (2e-2 if general else 3e-3), (k, S.rel_l2(ga[k], go[k]))
    if general:
        # ... and a MAX norm over the particles whose adjoint is well conditioned (the liquid takes no SVD; the others: singular values apart in every frame)
        frames0 = [dict(C=sc['C'] if 'C' in sc else np.zeros((N, 3, 3), np.float32), F=sc['F'])] + fb
        quiet = (sc['used'] > 0) & ((sc['mat'] == S.WATER) | _well_conditioned(frames0))
        mo = {k: (_max_off(ga[k], gb[k], quiet), _max_off(gc[k], gb[k], quiet), _max_off(ga[k], go[k], quiet)) for k in ga}
        print(f'MEASURED fuse_g2p[{scene}]: max norm over the {int(quiet.sum())} of {N} well-conditioned particles (fused vs separate, separate vs separate, fused vs oracle)',
              {k: [float(f'{x:.2g}') for x in v] for k, v in mo.items()})
        assert quiet.mean() >= 0.2
        for k, v in mo.items():
            assert v[0] <= 2e-2 and v[2] <= 4e-2, (k, v)            # (measured, two builds of the list: 3.2e-4 ... 6e-3 against the separate launches -- themselves 3e-4 ... 3.6e-3 apart from run to run --, 1.1e-2 ... 1.5e-2 against the oracle)
    assert (fa[-1]['used'] == fo[-1]['used']).all()
    assert np.abs(fa[-1]['x'] - fo[-1]['x']).max() <= 5e-6 and S.rel_l2(fa[-1]['v'], fo[-1]['v']) <= 1e-3


def test_fused_g2p_p2g_launch_counts_and_fallbacks(hiplib):
    """Which substeps fuse: inside one fe_step call, not across a sort, not in per-substep calls; never with a mesh effector at the particles, rigid bodies or
    a collector (those put a pass between g2p and the next p2g)."""
    sc = S.water_block(n_grid=32, n_particles=4000, seed=3, lo=0.3, hi=0.6)
    g = S.make_engine(hiplib, sc, options={'sort_interval': 5, 'fuse_g2p': 1, 'fuse_grid': 0})
    g.profile_enable(True)
    g.step(0, 0, 12, 0)                  # sorts at 0, 5, 10: p2g at 0, 5, 10; fused at 1-4, 6-9, 11; g2p at 4, 9, 11
    for f in range(12, 15):
        g.substep(f, f, 0)               # per-substep calls never fuse
    prof = g.profile_read()
    g.close()
    assert prof['g2p_p2g'][1] == 9 and prof['p2g'][1] == 3 + 3 and prof['g2p'][1] == 3 + 3 and prof['grid_op'][1] == 15, prof


def test_fused_g2p_p2g_with_an_injector(hiplib, oracle64):
    """... and through a LatteArt-like pass (ranged fe_step calls with act = 1): pool particles wait unused, milk is injected in the first substep of a step -- a
    particle that entered in substep f - 1 has no g2p of that substep, the fused launch reads its state from frame f."""
    sc = S.latte_mini()
    a = S.run_latte(hiplib, sc, options={'fuse_g2p': 1, 'sort_interval': 3})
    b = S.run_latte(hiplib, sc, options={'fuse_g2p': 0, 'sort_interval': 3})
    o = S.run_latte(oracle64, sc)
    assert (a['final']['used'] == b['final']['used']).all() and (a['final']['used'] == o['final']['used']).all()
    for k in 'xvCF':
        assert np.abs(a['final'][k] - b['final'][k]).max() <= 1e-5 * max(1.0, np.abs(b['final'][k]).max()), k
    assert S.rel_l2(a['action_grad'], b['action_grad']) <= 2e-5 and S.rel_l2(a['step_loss'], b['step_loss']) <= 1e-6
    assert S.cosine(a['action_grad'], o['action_grad']) >= 0.999999 and S.rel_l2(a['action_grad'], o['action_grad']) <= 1e-4


@pytest.mark.parametrize('scene,opts', [('block', {}), ('block', {'compact_F': 0}), ('block', {'grid_store': 0}), ('droplets-water', {'quad_min_units': 0}),
                                        ('droplets-water', {'quad_min_units': 0, 'lane_split': 0}), ('droplets-water', {'quad_min_units': 1 << 30, 'loose_max': 3}),
                                        ('droplets-mixed', {}), ('droplets-mixed', {'loose_max': 3})])
def test_fused_p2g_grad_g2p_grad_launch_matches_separate_launches(hiplib, oracle64, scene, opts):
    """Option fuse_bwd (round 5): inside a fe_step_grad call substep f's p2g_grad takes substep f - 1's g2p_grad along (k_pgg_g2pg) -- the adjoints of x, v, C of
    frame f go from one to the other in registers and reach memory only for particles somebody else reads them of.  The same reverse sweep as three launches per
    substep: two ranged calls with a loss seeded on the frame between them (the call boundary and every sort boundary are unfused substeps), against the unfused
    engine run twice (its own noise) and the fp64 oracle.  Pair units, quad units (the wave's words of the g2p_grad part inside the tile it gathered from), split
    waves, tail units, unused slots, drifted particles on both slow paths; without the per-frame grid store nothing fuses (the recompute comes first)."""
    if scene == 'block':
        rng = np.random.RandomState(5)
        N = 6000
        sc = S.water_block(n_grid=32, n_particles=N, seed=3, lo=0.3, hi=0.6)
        sc['v'] = S.f32(rng.normal(0, 1.0, (N, 3)) + [2.0, -3.0, 1.0])
        sc['used'] = (rng.rand(N) > 0.1).astype(np.int32)
        K = 4
    else:
        sc = _droplet_scene(scene.split('-')[1])
        N = len(sc['used'])
        K = 10
    n_sub, mid = 13, 6
    cot, cot_mid = S.random_cotangent(N, seed=7), S.random_cotangent(N, seed=9)

    def run(lib, o, ranged):
        g = S.make_engine(lib, sc, options=o)
        if lib is hiplib:
            g.profile_enable(True)
        g.step(0, 0, n_sub, 0) if ranged else [g.substep(f, f, 0) for f in range(n_sub)]
        g.reset_grad()
        g.add_grad(n_sub, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
        if ranged:
            g.step_grad(mid, mid, n_sub - mid, 0)
        else:
            [g.substep_grad(f, f, 0) for f in reversed(range(mid, n_sub))]
        at_mid = dict(zip(('gx', 'gv', 'gC', 'gF'), g.get_grad(mid)))
        g.add_grad(mid, cot_mid['gx'], cot_mid['gv'], cot_mid['gC'], cot_mid['gF'])
        if ranged:
            g.step_grad(0, 0, mid, 0)
        else:
            [g.substep_grad(f, f, 0) for f in reversed(range(mid))]
        out = dict(zip(('gx', 'gv', 'gC', 'gF'), g.get_grad(0)))
        prof = g.profile_read() if lib is hiplib else None
        stats = g.get_stats(n_sub) if lib is hiplib else None
        g.close()
        return at_mid, out, prof, stats

    base = dict({'sort_interval': K}, **opts)
    ma, ga, pa, st = run(hiplib, dict(base, fuse_bwd=1), True)
    mb, gb, pb, _ = run(hiplib, dict(base, fuse_bwd=0), True)
    mc, gc, _, _ = run(hiplib, dict(base, fuse_bwd=0), True)
    mo, go, _, _ = run(oracle64, {}, False)
    n_fused = pa.get('pgg_g2pg', (0, 0))[1]
    if opts.get('grid_store', 1) == 0:
        assert n_fused == 0
    else:
        # fused: every backward substep f whose predecessor f - 1 lies in the same call and the same sort interval
        want = sum(1 for lo, hi in ((mid, n_sub), (0, mid)) for f in range(lo + 1, hi) if f % K != 0)
        # (a frame whose grid store is incomplete -- a drifted particle reached a block outside the order's active list -- is recomputed, and its substep does not fuse)
        assert (n_fused == want if scene == 'block' else want - 3 <= n_fused <= want) and pa['p2g_grad'][1] == n_sub - n_fused and pa['g2p_grad'][1] == n_sub - n_fused, (pa, want)
    assert pb.get('pgg_g2pg', (0, 0))[1] == 0
    if scene != 'block':
        assert st['n_slow_path'] > 0, st
    print(f'MEASURED fuse_bwd[{scene}, {opts}]: {n_fused} fused launches; fused vs separate', {k: round(S.rel_l2(ga[k], gb[k]), 9) for k in ga}, 'separate vs separate',
          {k: round(S.rel_l2(gc[k], gb[k]), 9) for k in ga}, '| at the call boundary', {k: round(S.rel_l2(ma[k], mb[k]), 9) for k in ma}, '| vs fp64 oracle', {k: round(S.rel_l2(ga[k], go[k]), 8) for k in ga},
          '| median / 95th percentile over the particles, fused vs separate', {k: [float(f'{_pct_off(ga[k], gb[k], q):.2g}') for q in (50, 95)] for k in ga}, 'separate vs separate',
          {k: [float(f'{_pct_off(gc[k], gb[k], q):.2g}') for q in (50, 95)] for k in ga})
    general = scene.endswith('mixed')                          # (the SVD build: k_pgg_g2pg<3, true> on the pairs-only list, its own 52 KB arena)
    for got, ref, noise, orc in ((ma, mb, mc, mo), (ga, gb, gc, go)):
        for k in ('gx', 'gv', 'gC', 'gF'):
            assert np.isfinite(got[k]).all()
            # (the SVD materials: by the median and the 95th percentile over the particles, _pct_off)
            if general:
                assert _pct_off(got[k], ref[k], 50) <= 4.0 * _pct_off(noise[k], ref[k], 50) + 1e-5 and _pct_off(got[k], ref[k], 95) <= 4.0 * _pct_off(noise[k], ref[k], 95) + 1e-3, \
                    (k, [_pct_off(got[k], ref[k], q) for q in (50, 95)], [_pct_off(noise[k], ref[k], q) for q in (50, 95)])
            else:
                assert S.rel_l2(got[k], ref[k]) <= 4.0 * S.rel_l2(noise[k], ref[k]) + 2e-5, (k, S.rel_l2(got[k], ref[k]), S.rel_l2(noise[k], ref[k]))
            assert S.cosine(got[k], orc[k]) >= 0.999 and S.rel_l2(got[k], orc[k]) <= (2e-2 if general else 3e-3), (k, S.rel_l2(got[k], orc[k]))


def test_fused_p2g_grad_g2p_grad_with_an_injector(hiplib, oracle64):
    """... through a LatteArt-like pass: a particle injected in substep f - 1 is in use in frame f and not in frame f - 1 -- the fused launch stores its adjoint
    (Injector.act's adjoint reads it in the next launch); pool particles pass theirs through untouched."""
    sc = S.latte_mini()
    a = S.run_latte(hiplib, sc, options={'fuse_bwd': 1, 'sort_interval': 3})
    b = S.run_latte(hiplib, sc, options={'fuse_bwd': 0, 'sort_interval': 3})
    o = S.run_latte(oracle64, sc)
    assert S.rel_l2(a['action_grad'], b['action_grad']) <= 2e-5 and S.rel_l2(a['step_loss'], b['step_loss']) <= 1e-6
    assert S.cosine(a['action_grad'], o['action_grad']) >= 0.999999 and S.rel_l2(a['action_grad'], o['action_grad']) <= 1e-4


# ---------------------------------------------------------------------------------------------------------------------------------
# option fuse_grid (round 6): grid_op inside the forward scatter launches (the fused grid pass, FG kernels) -- off by default (measured
# slower than the launch it replaces: DESIGN.md section 10), kept as an option and held to the same parity as the separate k_grid
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('fuse_grid', [2, 6])            # 2 = wherever possible (late deposits included), 6 = ... and no wave ever waits: every entry takes the skipped road
@pytest.mark.parametrize('scene,opts', [('block', {}), ('droplets-water', {'quad_min_units': 0}), ('droplets-mixed', {'quad_min_units': 1 << 30, 'lane_split': 0}), ('fast', {})])
def test_fused_grid_pass_matches_separate_grid_kernel(hiplib, oracle64, scene, opts, fuse_grid):
    """The forward trajectory with grid_op riding on k_p2g / k_g2p_p2g (tile arrivals, owners, the final wave) against the same engine with k_grid as a launch of
    its own and against the fp64 oracle; then the reverse sweep over the frames the pass stored (its records are what the adjoint kernels read).
    'fast': a coherent drift of 1.5 cells and more between two sorts -- slow-path deposits outside the 27 neighbours of the particle's own block are LATE
    (ordered by no arrival): their planes, the late list, the final wave's fix-up of what the owners stored."""
    if scene == 'block':
        rng = np.random.RandomState(5)
        N = 6000
        sc = S.water_block(n_grid=32, n_particles=N, seed=3, lo=0.3, hi=0.6)
        sc['v'] = S.f32(rng.normal(0, 1.0, (N, 3)) + [2.0, -3.0, 1.0])
        sc['used'] = (rng.rand(N) > 0.1).astype(np.int32)
        K, n_sub = 4, 13
    elif scene == 'fast':
        N = 4000
        sc = S.water_block(n_grid=32, n_particles=N, lo=0.3, hi=0.5, gravity=(0.0, 0.0, 0.0))
        sc['v'] = S.f32(np.tile([30.0, -22.0, 10.0], (N, 1)))          # 30 m/s * 16 substeps * 2e-4 s * 32 = 3 cells between two sorts
        K, n_sub = 16, 16
    else:
        sc = _droplet_scene(scene.split('-')[1])
        N = len(sc['used'])
        K, n_sub = 10, 13
    cot = S.random_cotangent(N, seed=7)

    def run(lib, o):
        g = S.make_engine(lib, sc, options=o)
        if lib is hiplib:
            g.profile_enable(True)
            g.step(0, 0, 6, 0)
            g.step(6, 6, n_sub - 6, 0)
        else:
            for f in range(n_sub):
                g.substep(f, f, 0)
        frames = [S.get_state(g, f) for f in range(1, n_sub + 1)]
        g.reset_grad()
        g.add_grad(n_sub, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
        g.step_grad(0, 0, n_sub, 0)
        gx, gv, gC, gF = g.get_grad(0)
        prof = g.profile_read() if lib is hiplib else None
        g.sync()                                                    # (a wait of the pass that did not end is reported here)
        g.close()
        return frames, dict(gx=gx, gv=gv, gC=gC, gF=gF), prof

    base = dict({'sort_interval': K}, **opts)
    fa, ga, pa = run(hiplib, dict(base, fuse_grid=fuse_grid))
    fb, gb, pb = run(hiplib, dict(base, fuse_grid=0))
    fc, gc, _ = run(hiplib, dict(base, fuse_grid=0))                # the same engine once more: its own run-to-run noise
    fo, go, _ = run(oracle64, {})
    assert pa['grid_op'][1] == 0 and pb['grid_op'][1] == n_sub, (pa, pb)      # every forward substep of the window took the fused form
    if scene != 'fast':                                             # (there the front of the block runs into blocks off the order's active list: those frames are recomputed)
        assert pa['grid_op_keep'][1] == 0, pa                       # ... and left a complete record in the grid store: nothing recomputed on the way back
    general = scene.endswith('mixed')
    worst = {k: 0.0 for k in 'xvCF'}
    for f, (a, b, c) in enumerate(zip(fa, fb, fc), start=1):
        assert (a['used'] == b['used']).all(), f
        u = b['used'] > 0
        for k in 'xvCF':
            scale = max(1.0, float(np.abs(b[k][u]).max()))
            d, nz = float(np.abs(a[k][u] - b[k][u]).max()) / scale, float(np.abs(c[k][u] - b[k][u]).max()) / scale
            worst[k] = max(worst[k], d)
            assert d <= 4.0 * nz + {'x': 5e-7, 'v': 4e-6, 'C': 8e-5, 'F': 4e-6}[k], (f, k, d, nz)
    print(f'MEASURED fuse_grid={fuse_grid}[{scene}]: fused vs separate grid kernel, largest relative state difference', {k: float(f'{v:.2g}') for k, v in worst.items()},
          '| adjoints', {k: round(S.rel_l2(ga[k], gb[k]), 8) for k in ga}, '| vs fp64 oracle x', np.abs(fa[-1]['x'] - fo[-1]['x']).max(), {k: round(S.rel_l2(ga[k], go[k]), 8) for k in ga})
    for k in ('gx', 'gv', 'gC', 'gF'):
        assert np.isfinite(ga[k]).all()
        if general:
            assert _pct_off(ga[k], gb[k], 50) <= 4.0 * _pct_off(gc[k], gb[k], 50) + 1e-5 and _pct_off(ga[k], gb[k], 95) <= 4.0 * _pct_off(gc[k], gb[k], 95) + 1e-3, k
        else:
            assert S.rel_l2(ga[k], gb[k]) <= 4.0 * S.rel_l2(gc[k], gb[k]) + 2e-5, (k, S.rel_l2(ga[k], gb[k]), S.rel_l2(gc[k], gb[k]))
        assert S.cosine(ga[k], go[k]) >= 0.999 and S.rel_l2(ga[k], go[k]) <= (2e-2 if general else 1e-2), (k, S.rel_l2(ga[k], go[k]))
    assert (fa[-1]['used'] == fo[-1]['used']).all()
    assert np.abs(fa[-1]['x'] - fo[-1]['x']).max() <= 5e-6 and S.rel_l2(fa[-1]['v'], fo[-1]['v']) <= 1e-3


@pytest.mark.parametrize('fuse_grid', [1, 2])
def test_fused_grid_pass_with_an_injector(hiplib, oracle64, fuse_grid):
    """The injector's particles sit behind the order's work items until the next sort: with fuse_grid = 1 the substeps behind an injection keep the separate
    k_grid (fuse_grid_ok), with 2 their deposits are late ones.  Either way the LatteArt-like pass and its action gradient are those of the unfused engine."""
    sc = S.latte_mini()
    a = S.run_latte(hiplib, sc, options={'fuse_grid': fuse_grid, 'sort_interval': 3})
    b = S.run_latte(hiplib, sc, options={'fuse_grid': 0, 'sort_interval': 3})
    o = S.run_latte(oracle64, sc)
    assert (a['final']['used'] == b['final']['used']).all() and (a['final']['used'] == o['final']['used']).all()
    for k in 'xvCF':
        assert np.abs(a['final'][k] - b['final'][k]).max() <= 1e-5 * max(1.0, np.abs(b['final'][k]).max()), k
    assert S.rel_l2(a['action_grad'], b['action_grad']) <= 2e-5 and S.rel_l2(a['step_loss'], b['step_loss']) <= 1e-6
    assert S.cosine(a['action_grad'], o['action_grad']) >= 0.999999 and S.rel_l2(a['action_grad'], o['action_grad']) <= 1e-4


def test_incomplete_adjoint_slots_are_refused(hiplib):
    """After a fused fe_step_grad(f0, n > 1) only the adjoint of frame f0 is in memory: frame f0 + 1's slot holds F's adjoint and leftovers (k_pgg_g2pg passed
    x, v, C on in registers).  The API refuses that slot instead of returning it; fuse_bwd = 0 keeps every frame's; a reset clears the state."""
    sc = S.water_block(n_grid=32, n_particles=4000, seed=3, lo=0.3, hi=0.6)
    cot = S.random_cotangent(sc['N'], seed=2)
    for fuse in (1, 0):
        g = S.make_engine(hiplib, sc, options={'sort_interval': 10, 'fuse_bwd': fuse})
        g.step(0, 0, 6, 0)
        g.reset_grad()
        g.add_grad(6, cot['gx'], cot['gv'], cot['gC'], cot['gF'])
        g.step_grad(2, 2, 4, 0)                                    # frames 5 ... 2: the slot of frame 3 is the incomplete one
        g.get_grad(2)                                              # the call's first frame: defined
        if fuse:
            prof_ok = True
            for call in (lambda: g.get_grad(3), lambda: g.add_grad(3, cot['gx'], None, None, None), lambda: g.copy_grad(3, 4)):
                with pytest.raises(Exception, match='registers'):
                    call()
            g.step_grad(0, 0, 2, 0)                                # the sweep goes on from frame 2 as if nothing had happened
            g.get_grad(0)
            g.reset_grad()
            g.get_grad(3)                                          # a cleared ring has no incomplete slot
        else:
            g.get_grad(3)
        g.close()


def test_sort_keys_counted_in_g2p(hiplib):
    """Option sort_keys_in_g2p (round 6): the k_g2p launch in front of a sort frame counts the sort's keys, ranks and block counts while it writes the positions, and
    the sort skips k_sort_count.  The pre-counted keys describe ONE frame: they follow a copy of it (fluidlab's window wraps frame L to frame 0 and sorts that),
    they are dropped when the host rewrites the frame, and counts that no sort ever consumed are cleared before the next count.  Every road against the engine
    that always runs k_sort_count: the orders may differ in the ranks of a cell's particles, the trajectories only by the noise of the scatter's sums."""
    rng = np.random.RandomState(11)
    N = 6000
    sc = S.water_block(n_grid=32, n_particles=N, seed=3, lo=0.3, hi=0.6)
    sc['v'] = S.f32(rng.normal(0, 1.0, (N, 3)) + [2.0, -3.0, 1.0])
    sc['used'] = (rng.rand(N) > 0.1).astype(np.int32)

    def run(pre, road):
        g = S.make_engine(hiplib, sc, options={'sort_interval': 5, 'sort_keys_in_g2p': pre})
        g.profile_enable(True)
        g.step(0, 0, 10, 0)                                      # sorts at 0 and 5; substeps 4 and 9 end in a k_g2p in front of a sort frame
        if road == 'wrap':                                       # frame 10 becomes frame 0 and is sorted there
            g.copy_frame(10, 0)
            g.step(0, 0, 10, 0)
            out = S.get_state(g, 10)
        elif road == 'edit':                                     # the host moves particles of frame 10: its pre-counted keys are stale
            st = S.get_state(g, 10)
            x = st['x'].copy()
            x[::7] += np.float32(0.11)
            g.set_frame(10, x=np.clip(x, 0.1, 0.9).astype(np.float32))
            g.step(10, 10, 10, 0)
            out = S.get_state(g, 20)
        else:                                                    # the caller goes on somewhere else: the counts of frame 10 are never consumed
            st = S.get_state(g, 7)
            g.set_frame(30, x=st['x'], v=st['v'], C_=st['C'], F=st['F'], used=st['used'])
            g.step(30, 30, 10, 0)
            out = S.get_state(g, 40)
        prof = g.profile_read()
        g.sync()
        g.close()
        return out, prof

    for road in ('wrap', 'edit', 'elsewhere'):
        a, pa = run(1, road)
        b, _ = run(0, road)
        c, _ = run(0, road)
        assert (a['used'] == b['used']).all(), road
        u = b['used'] > 0
        for k in 'xvCF':
            scale = max(1.0, float(np.abs(b[k][u]).max()))
            d, nz = float(np.abs(a[k][u] - b[k][u]).max()) / scale, float(np.abs(c[k][u] - b[k][u]).max()) / scale
            assert d <= 4.0 * nz + {'x': 5e-7, 'v': 4e-6, 'C': 8e-5, 'F': 4e-6}[k], (road, k, d, nz)
        assert pa['sort'][1] == 4, (road, pa['sort'])


@pytest.mark.gpu
@pytest.mark.parametrize('n_grid', [32, 64])
def test_sort_one_scan_launch_matches_two(hiplib, oracle64, n_grid):
    """Option sort_one_scan (round 6): the sort's two scan launches as one -- the scan's workgroups publish their partial sums, wait for each other on a counter and go on
    (k_sort_blk_scan), the active list's length reaches the table through the last workgroup of that job.  64^3: five scan workgroups that wait for each other (32^3: one).
    A dense core among droplets (pairs, leftovers, singles of every size class), three sorts: the work lists of the two roads are identical, both follow the fp64 oracle
    (the launches themselves: profiles/r06_kernel_stats_*.csv list k_sort_blk_scan and neither k_sort_blk_partial nor k_sort_blk_final)."""
    rng = np.random.RandomState(41)
    core = np.concatenate([(4 * np.array([3 + bi, 3 + bj, 3 + bk]) + 0.5) / 32 + rng.uniform(0.02, 0.98, (int(rng.randint(140, 500)), 3)) * (4 / 32)
                           for bi in range(2) for bj in range(3) for bk in range(2)])
    drops = np.concatenate([c + rng.uniform(-r, r, (20, 3)) for c, r in zip(rng.uniform(0.15, 0.85, (60, 3)), rng.choice([0.01, 0.02, 0.04], 60))])
    x = S.f32(np.clip(np.concatenate([core, drops]), 0.08, 0.92))
    N = len(x)
    sc = dict(S.water_block(n_grid=n_grid, n_particles=N, seed=3), x=x, v=S.f32(rng.normal(0, 0.4, (N, 3))))
    cot = S.random_cotangent(N, seed=4)
    res = {}
    for one in (1, 0):
        g = S.make_engine(hiplib, sc, options={'sort_interval': 4, 'sort_one_scan': one})
        g.profile_enable(True)
        sa, ga = S.run_forward_backward(g, 9, cot)
        prof = g.profile_read()
        res[one] = (sa, ga, [g.get_work_stats(f) for f in (0, 4, 8)], prof['sort'][1])
        g.close()
    o = S.make_engine(oracle64, sc)
    sb, gb = S.run_forward_backward(o, 9, {k: v.astype(np.float64) for k, v in cot.items()})
    assert res[1][2] == res[0][2], (res[1][2], res[0][2])          # the same items, pairs, singles, units and active blocks, sort by sort
    assert res[1][2][2]['n_active_blocks'] > 0 and res[1][2][2]['n_items'] > 40
    assert res[1][3] == 3 and res[0][3] == 3, (res[1][3], res[0][3])              # three sorts each (frames 0, 4, 8)
    for one in (1, 0):
        sa, ga = res[one][0], res[one][1]
        assert np.abs(sa['x'] - sb['x']).max() <= 2e-6 and S.rel_l2(sa['v'], sb['v']) <= 1e-4, one
        for k in ('gx', 'gv', 'gC', 'gF'):
            assert S.cosine(ga[k], gb[k]) >= 0.99999 and S.rel_l2(ga[k], gb[k]) <= 1e-3, (one, k, S.rel_l2(ga[k], gb[k]))

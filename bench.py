#!/usr/bin/env python
"""bench.py -- MPM substep pairs/s (one forward + one backward substep) on the BASELINE workloads.

  python bench.py [--gpus N --steps K --warmup W]

N = 1 (BASELINE configs[1] inputs, the configuration the metric is quoted on): single-material water block,
  128^3 grid, 200,000 particles, x ~ U([0.30,0.53]^3) seed 0, g=(0,-10,0), CubeBoundary [0.05,0.95]^3, run forward AND backward.
  The block EVOLVES: one "step" = CHUNK forward substeps continuing from where the previous step ended (frame CHUNK is
  copied to frame 0: a rolling window) followed by CHUNK backward substeps (adjoint seeded by a squared-distance loss on
  the window's last frame).  The block falls, hits the floor and spreads while it is timed, so particles change cells and
  tiles, the order is re-sorted every K substeps and the active node set drifts (`n_slow_path`, `nc_*` in the line).
  All inputs are resident in HBM before the timed region; nothing crosses PCIe inside it.

N > 1 (BASELINE configs[3]): one LatteArt-v0 replica per GPU at 128^3 (the config-3 scene, injector
  randomness seeded by the rank), one process per GPU.  One "step" = one optimisation pass of fluidlab's Solver: 3300
  forward substeps with the loss, 3300 backward substeps, `agent.get_grad(horizon_action)`, ONE RCCL all-reduce of that
  (251 x 3) action gradient through EnvParallel.all_reduce_mean, the identical fp64 Adam step on every rank.  No other
  collective: weak scaling.  `python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_GRID, N_PARTICLES, CHUNK = 128, 200_000, 100
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # measured float4 copy on this part: `frac_of_measured_copy`
# rocprofv3 --pmc passes of THIS command line (`--steps 20 --warmup 5`), sliced by window (scripts/phase_profile.py): the traffic of
# the roofline kernel is read for the same windows it is timed in, or not at all
# (the files also record the engine's source hash and the device they were taken on: numbers of another build or another GPU are not printed)
PMC_TRAFFIC = os.path.join(ROOT, 'profiles', 'r06_pmc_traffic.json')
ROCPROF_TIMED = os.path.join(ROOT, 'profiles', 'r06_kernel_stats_timed_region.csv')
SRC_HASH_FILE = os.path.join(ROOT, 'fluidlab_amd', 'csrc', 'libfluidengine_hip.so.srchash')
# fixed substep windows of the evolving block, comparable across --steps and across rounds (window w = substeps [100 w, 100 w + 100))
PHASES = {'falling': (5, 11), 'impact': (11, 18), 'splash': (26, 34), 'layer': (80, 90)}

# algorithmic bytes per launch (DESIGN.md "Roofline accounting"; N = used particles, Nc = touched nodes).
# They sum to SURVEY 8d's B_fwd = 216 N + 72 Nc and B_bwd = 308 N + 132 Nc.
KERNEL_BYTES = {
    'p2g': (156, 16), 'grid_op': (0, 44), 'g2p': (60, 12),
    'g2p_p2g': (216, 28),                           # k_g2p_p2g: the g2p of one substep and the p2g of the next in one launch -- credited both (what it spares is the re-read of x' v' C')
    'p2g_recompute': (116, 16), 'grid_op_keep': (0, 28), 'g2p_grad': (60, 24), 'grid_op_grad': (0, 48), 'p2g_grad': (132, 16),
    'pgg_g2pg': (192, 40),                          # k_pgg_g2pg: a substep's p2g_grad and the next one's g2p_grad in one launch -- credited both (spared: the round trip of the adjoints of x, v, C)
    'sort': (0, 0), 'reorder_grad': (0, 0),          # overhead of the cell-sorted layout: no algorithmic bytes credited
}
FWD_KERNELS = ('p2g', 'grid_op', 'g2p')
# The backward pass reads the particle state of frame f again (116 bytes per particle: SURVEY 8d books them on the recompute of grid[f]).
# When the per-frame grid store spares that recompute, the read does not go away -- k_p2g_grad performs it (state 116 + adjoint in 96 +
# adjoint out 96 + info = 308 bytes per particle all told) -- so the per-kernel figures credit it to the kernel that ran: a p2g_grad launch
# without a p2g_recompute launch beside it moves (132 + 116) N + 16 Nc.  What is credited and never moved in such a frame is the grid part
# of the recompute, 44 Nc (16 accumulate-write + 28 grid_op_keep): `pair_roofline.frac_executed` leaves it out.
STATE_REREAD = 116
GRID_RECOMPUTE = 44


def fwd_substeps(prof):
    """forward substeps in a profile: every one has exactly one p2g, alone or at the tail of a k_g2p_p2g launch"""
    return prof.get('p2g', (0, 0))[1] + prof.get('g2p_p2g', (0, 0))[1]


def launch_bytes(name, cnt, prof, n_used, nc):
    """algorithmic bytes of `cnt` launches of kernel `name` in a window whose launch counts are `prof` ({name: (ms, launches)})"""
    bp, bc = KERNEL_BYTES.get(name, (0, 0))
    b = cnt * (bp * n_used + bc * nc)
    if name == 'p2g_grad':
        b += max(0, cnt - prof.get('p2g_recompute', (0.0, 0))[1]) * STATE_REREAD * n_used
    if name == 'pgg_g2pg':                           # (fused only where the grid store spared the recompute: its p2g_grad part reads the state)
        b += cnt * STATE_REREAD * n_used
    return b
METRIC = 'MPM substeps/sec (fwd+bwd), 128^3 grid / 200k particles'


# ------------------------------------------------------------------------------------------------------------------
# N = 1: the evolving water block
# ------------------------------------------------------------------------------------------------------------------
def build_block(elib, device, n_grid=N_GRID, n_particles=N_PARTICLES, L=CHUNK, mat=None, seed=0):
    from fluidlab_amd import scenes as S
    sc = S.water_block(n_grid=n_grid, n_particles=n_particles, seed=seed, **({} if mat is None else {'mat': mat}))
    eng = S.make_engine(elib, sc, max_substeps_local=L, device=device)
    eng.loss_alloc(1)
    rng = np.random.RandomState(1)
    eng.loss_set_target(0, (sc['x'] + rng.normal(0, 0.01, sc['x'].shape)).astype(np.float32))
    return eng, sc


def window_step(eng, chunk, mat=0, backward=True, roll=True):
    """CHUNK forward substeps from frame 0, CHUNK backward substeps, then the window rolls on (frame chunk -> frame 0)."""
    eng.step(0, 0, chunk, 0)
    if backward:
        eng.reset_grad()
        eng.loss_step_grad(0, chunk, mat, 1.0, 1.0)
        eng.step_grad(0, 0, chunk, 0)
    if roll:
        eng.copy_frame(chunk, 0)


def kernel_table(prof, n_used, nc):
    out = {}
    for name, (ms, cnt) in prof.items():
        if cnt:
            us = 1e3 * ms / cnt
            b = launch_bytes(name, cnt, prof, n_used, nc) / cnt
            out[name] = {'avg_us': round(us, 3), 'launches': cnt, 'alg_bytes': int(b), 'GBps': round(b / (us * 1e-6) / 1e9, 1)}
    return out


def probe_pass(elib, n_windows, timed, profiled):
    """The same evolving block once more on a fresh engine, untimed: the touched-node count Nc at every window boundary (the
    physics is deterministic, so this IS the timed run's Nc history), and per window either its wall time or -- for the windows in
    `profiled` -- the HIP-event time of every kernel.  Feeds pair_roofline (Nc integrated over the timed windows), roofline (the
    dominant kernel over the timed windows, bytes from those windows' Nc) and extra.phase_rates."""
    eng, _ = build_block(elib, 0)
    rec = []
    for w in range(n_windows):
        st = eng.get_stats(0)                                 # frame 0 = the state at substep 100 w (syncs; clears the slow counter)
        r = {'w': w, 'nc0': int(st['n_cells_touched']), 'n_used': int(st['n_used'])}
        if w in profiled:
            eng.profile_enable(True)
            window_step(eng, CHUNK)
            r['prof'] = eng.profile_read()
            eng.profile_enable(False)
        else:
            eng.sync(); t0 = time.perf_counter()
            window_step(eng, CHUNK)
            eng.sync(); r['dt'] = time.perf_counter() - t0
        rec.append(r)
    st = eng.get_stats(0)
    eng.close()
    for i, r in enumerate(rec):
        r['nc1'] = rec[i + 1]['nc0'] if i + 1 < len(rec) else int(st['n_cells_touched'])
        r['nc'] = 0.5 * (r['nc0'] + r['nc1'])
    return rec


def fold_windows(rec, lo, hi):
    """Windows [lo, hi) of a probe pass -> pairs/s of the plain-timed ones, per-kernel event times and algorithmic bytes of the
    profiled ones (bytes of a launch = its window's mean Nc), mean Nc."""
    ws = [r for r in rec if lo <= r['w'] < hi]
    if not ws:
        return None
    plain = [r for r in ws if 'dt' in r]
    ms, cnt, byt = {}, {}, {}
    for r in ws:
        for name, (m, c) in r.get('prof', {}).items():
            if c:
                ms[name] = ms.get(name, 0.0) + m; cnt[name] = cnt.get(name, 0) + c
                byt[name] = byt.get(name, 0.0) + launch_bytes(name, c, r['prof'], r['n_used'], r['nc'])
    kern = {k: {'avg_us': round(1e3 * ms[k] / cnt[k], 3), 'launches': cnt[k], 'alg_bytes': int(byt[k] / cnt[k]),
                'GBps': round(byt[k] / (ms[k] * 1e-3) / 1e9, 1)} for k in ms}
    nc = float(np.mean([r['nc'] for r in ws]))
    n_used = ws[0]['n_used']
    out = {'substeps': [lo * CHUNK, hi * CHUNK], 'nc_mean': int(nc), 'nc_min': int(min(min(r['nc0'], r['nc1']) for r in ws)),
           'nc_max': int(max(max(r['nc0'], r['nc1']) for r in ws)), 'kernels': kern}
    # share of the backward substeps whose grid[f] came from the per-frame store (no p2g_recompute / grid_op_keep launch)
    n_bwd = cnt.get('p2g_grad', 0) + cnt.get('pgg_g2pg', 0)          # backward substeps: every one has exactly one p2g_grad, alone or at the head of a k_pgg_g2pg launch
    stored = 1.0 - cnt.get('p2g_recompute', 0) / n_bwd if n_bwd else None
    out['bwd_frames_from_grid_store'] = None if stored is None else round(stored, 4)
    if plain:
        rate = len(plain) * CHUNK / sum(r['dt'] for r in plain)
        b_pair = 524 * n_used + 204 * nc
        out['pairs_per_s'] = round(rate, 1)
        out['pair_roofline_frac'] = round(b_pair * rate / 1e9 / HBM_PEAK_GBS, 4)
        if stored is not None:
            out['pair_roofline_frac_executed'] = round((b_pair - stored * GRID_RECOMPUTE * nc) * rate / 1e9 / HBM_PEAK_GBS, 4)
    return out


def source_hash():
    try:
        return open(SRC_HASH_FILE).read().strip()
    except OSError:
        return None


def cpu_baseline(budget_s=12.0):
    """The oracle (fp32 build, OpenMP over the host cores) on the same 128^3 / 200k workload: a bounded number of forward+backward
    substep pairs.  Its particle scatters (P2G, the adjoint scatter of G2P) run colour by colour over 4^3-cell blocks with plain adds
    (oracle option `scatter`): with `#pragma omp atomic` they got slower beyond ~16 threads.  Reported, never the target."""
    from fluidlab_amd import _capi
    elib = _capi.EngineLib(os.path.join(ROOT, 'oracle', '_build', 'libfe_oracle_f32.so'))
    L = 3
    eng, _ = build_block(elib, 0, L=L)
    eng.set_option('scatter', 1)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    window_step(eng, 1)                                      # warm (page faults)
    best, sweep = None, {}
    for c in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
        eng.set_option('threads', c)
        t = time.perf_counter(); window_step(eng, 1); t = time.perf_counter() - t
        sweep[c] = round(1.0 / t, 2)
        if best is None or t < best[0]:
            best = (t, c)
    cores = best[1]
    eng.set_option('threads', cores)
    t0 = time.perf_counter()
    pairs = 0
    while True:
        window_step(eng, L)
        pairs += L
        if time.perf_counter() - t0 > budget_s or pairs >= 150:
            break
    dt = time.perf_counter() - t0
    eng.close()
    return {'value': pairs / dt, 'unit': 'substep_pairs/s', 'cores': cores, 'kind': 'port', 'threads_sweep_pairs_per_s': sweep,
            'sample': f'{pairs} fwd+bwd substep pairs of the same 128^3/200k water block (evolving window), oracle fp32 + OpenMP '
                      f'({cores} of {ncpu} hardware threads, fastest of a short sweep), {dt:.1f}s',
            'scatter': 'coloured: particles bucketed by 4^3-cell block, eight colours one after the other, the blocks of a colour in parallel '
                       'with plain adds (no atomics); dense n^3 grid loops as in the reference'}


def extra_block(elib, device, name, n_grid, n, mat, L, reps):
    """A bounded side measurement: pairs/s, per-kernel event times and pair_roofline of an at-rest block, never part of `value`."""
    from fluidlab_amd import scenes as S
    rng = np.random.RandomState(0)
    side = (n / 8.0) ** (1 / 3) / n_grid                    # ~8 particles per cell
    sc = S.water_block(n_grid=n_grid, n_particles=n, seed=0, mat=mat)
    sc['x'] = S.f32(rng.uniform(0.3, 0.3 + side, (n, 3)))
    eng = S.make_engine(elib, sc, max_substeps_local=L, device=device)
    eng.loss_alloc(1); eng.loss_set_target(0, sc['x'])
    window_step(eng, L, mat=mat, roll=False); eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        window_step(eng, L, mat=mat, roll=False)
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    eng.profile_enable(True); window_step(eng, L, mat=mat, roll=False); prof = eng.profile_read(); eng.profile_enable(False)
    st = eng.get_stats(L // 2)
    b_pair = 524 * st['n_used'] + 204 * st['n_cells_touched']
    out = {'workload': name, 'pairs_per_s': round(L / dt, 1), 'n_used': int(st['n_used']), 'n_cells_touched': int(st['n_cells_touched']),
           'pair_roofline': {'alg_bytes_per_pair': b_pair, 'frac': round(b_pair * L / dt / 1e9 / HBM_PEAK_GBS, 4)},
           'kernels': kernel_table(prof, int(st['n_used']), int(st['n_cells_touched'])),
           'kernels_us': {k: round(1e3 * v[0] / v[1], 1) for k, v in prof.items() if v[1]}}
    eng.close()
    return out


def extra_evolving(elib, device, name, mat, dt, warm, n_windows, note):
    """A bounded side measurement on the metric's own size (128^3 / 200k) through the GENERAL kernels: the evolving block of `mat`, rolling
    windows like the headline -- `warm` windows untimed, `n_windows` timed (sorts included), then one more with HIP events around every
    kernel.  Never part of `value`."""
    from fluidlab_amd import scenes as S
    sc = S.water_block(n_grid=N_GRID, n_particles=N_PARTICLES, seed=0, mat=mat)
    sc['dt'] = dt
    eng = S.make_engine(elib, sc, max_substeps_local=CHUNK, device=device)
    eng.loss_alloc(1)
    eng.loss_set_target(0, (sc['x'] + np.random.RandomState(1).normal(0, 0.01, sc['x'].shape)).astype(np.float32))
    out = {'workload': name, 'dt': dt, 'note': note}
    try:
        for _ in range(warm):
            window_step(eng, CHUNK, mat=mat)
        eng.sync(); st0 = eng.get_stats(0); t0 = time.perf_counter()
        for _ in range(n_windows):
            window_step(eng, CHUNK, mat=mat)
        eng.sync(); dt_s = time.perf_counter() - t0
        st1 = eng.get_stats(0)
        eng.profile_enable(True); window_step(eng, CHUNK, mat=mat); prof = eng.profile_read(); eng.profile_enable(False)
        x = np.zeros((eng.N, 3), np.float32); eng.get_frame(0, x, None, None, None, None)
        nc = 0.5 * (st0['n_cells_touched'] + st1['n_cells_touched'])
        rate = n_windows * CHUNK / dt_s
        b_pair = 524 * st1['n_used'] + 204 * nc
        out.update({'pairs_per_s': round(rate, 1), 'timed_substeps': [warm * CHUNK, (warm + n_windows) * CHUNK], 'n_used': int(st1['n_used']),
                    'nc_start': int(st0['n_cells_touched']), 'nc_end': int(st1['n_cells_touched']), 'n_slow_path': int(st1['n_slow_path']),
                    'state_finite': bool(np.isfinite(x).all()),
                    'pair_roofline': {'alg_bytes_per_pair': int(b_pair), 'frac': round(b_pair * rate / 1e9 / HBM_PEAK_GBS, 4)},
                    'kernels': kernel_table(prof, int(st1['n_used']), int(st1['n_cells_touched'])),
                    'sorts_per_pair': round(prof.get('sort', (0, 0))[1] / max(1, fwd_substeps(prof)), 3)})
    except Exception as e:                                   # (a scene that leaves the grid must not take the bench line with it)
        out['error'] = str(e)[:200]
    eng.close()
    return out


def c5_injected(elib, device, t0_steps=125, passes=4):
    """BASELINE config 5 as SURVEY 8d C5 writes it: IceCreamDynamic-v0's scene at 256^3, a pool of 1,000,000 ICECREAM particles dispensed by the
    BallInjector (flux 10 per substep), the Rigid cone's SDF collider, the reference's 40-substep window; dt = 5e-5 (the reference's 2e-4 is past
    the solid's Courant limit on this grid: scripts/run_c5.py).  The demo policy runs `t0_steps` steps forward (50,000 particles in flight), then
    ONE step is differentiated `passes` times through fluidlab's Solver (forward with the loss, backward, agent.get_grad); the rate is taken from
    the Solver's own forward / backward clocks, which start after the state upload.  Never part of `value`."""
    import contextlib
    import io
    from fluidlab_amd.envs import make
    from fluidlab_amd.optimizer.policies import ActionsPolicy
    from fluidlab_amd.optimizer.solver import Solver
    out = {'workload': 'IceCreamDynamic-v0 scene at 256^3: 1M-particle ICECREAM pool, BallInjector flux 10, Rigid cone (analytic SDF), 40-substep window, dt 5e-5; '
                       'one step fwd+bwd through Solver.forward_backward after %d steps of the demo pour' % t0_steps}
    try:
        base = dict(quality=4, n_pool=1_000_000, inject_till=10**9, max_substeps_local=40, ckpt_dest='gpu', dt=5e-5, loss_type='default', device=device)
        quiet = lambda: contextlib.redirect_stdout(io.StringIO())
        with quiet():
            env = make('IceCreamDynamic-v0', seed=0, engine_lib=elib, loss=False, horizon=t0_steps + 1, **base)
            te = env.taichi_env
            table = env.demo_policy()
            te.apply_agent_action_p(table.get_actions_p())
            te.simulator.engine.sync(); t0 = time.perf_counter()
            for i in range(t0_steps):
                te.step(table.get_action_v(i))
            te.simulator.engine.sync()
        fwd_rate = t0_steps * te.simulator.n_substeps / (time.perf_counter() - t0)
        state = te.get_state()['state']
        te.simulator.engine.close()
        used0 = state['used'] > 0
        cone = np.asarray(state['agent'][1][:3], np.float64)
        acts = np.asarray(table.actions_v[t0_steps:t0_steps + 1], np.float64)
        tgt = state['x'] + np.random.RandomState(3).normal(0, 0.01, state['x'].shape)
        tgt[~used0] = [0.5, 0.78, 0.5]
        with quiet():
            env = make('IceCreamDynamic-v0', seed=0, engine_lib=elib, loss=True, horizon=1, **base)
            te = env.taichi_env
            te.loss.set_target({'x': tgt[None].astype(np.float32)})
            pol = ActionsPolicy(np.vstack([acts, (cone / np.asarray(te.agent.rigid.action_scale_p, np.float64)[:3])[None, :]]))
            pol.freeze_till = 0
            sol, secs, grad = Solver(env, None, None), [], None
            for _ in range(passes):
                info, grad = sol.forward_backward(state, pol, 1, 1)
                secs.append(info['forward_s'] + info['backward_s'])
        st = te.simulator.engine.get_stats(39)
        te.simulator.engine.close()
        ns = 40
        rate = ns / min(secs[1:])
        b_pair = 524 * st['n_used'] + 204 * st['n_cells_touched']
        out.update({'forward_only_substeps_per_s_during_the_pour': round(fwd_rate, 1), 'particles_in_flight': int(used0.sum()), 'n_used_end': int(st['n_used']),
                    'n_cells_touched': int(st['n_cells_touched']), 'pairs_per_s': round(rate, 1), 'pass_seconds': [round(x, 4) for x in secs],
                    'state_finite': bool(np.isfinite(state['x'][used0]).all()), 'action_grad_finite': bool(np.isfinite(np.asarray(grad)).all()),
                    'pair_roofline': {'alg_bytes_per_pair': int(b_pair), 'frac': round(b_pair * rate / 1e9 / HBM_PEAK_GBS, 4)},
                    'note': 'a window of 40 substeps per host crossing (the reference\'s memory model) with 5 % of the pool in flight: the per-call host work weighs more than in the resident runs'})
    except Exception as e:                                   # (never take the bench line down)
        out['error'] = str(e)[:300]
    return out


def run_single(args):
    import torch
    torch.cuda.set_device(0)
    from fluidlab_amd import _capi
    elib = _capi.load_hip()                               # no fallback: raises without the HIP library
    eng, sc = build_block(elib, 0)
    for o in args.opt:
        k, v = o.split('=')
        eng.set_option(k, float(v))

    def barrier():
        eng.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        window_step(eng, CHUNK)
    barrier()
    st0 = eng.get_stats(0)                                # (also clears the slow-path counter)
    barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(args.steps):
        window_step(eng, CHUNK)
    ev_ms = eng.timer_stop_ms()
    barrier()
    wall = time.perf_counter() - t0
    st1 = eng.get_stats(0)
    pairs = args.steps * CHUNK
    value = pairs / wall

    # ---- untimed: forward-only rate on the same engine, then the probe pass (Nc history, per-kernel event times, phase windows)
    t1 = time.perf_counter()
    nf = max(2, args.steps // 4)
    for _ in range(nf):
        window_step(eng, CHUNK, backward=False)
    barrier()
    fwd_rate = nf * CHUNK / (time.perf_counter() - t1)
    n_used = int(st1['n_used'])
    options = eng.get_options()                              # as the engine holds them: defaults, --opt, FE_* environment variables alike
    eng.close()
    w0, w1 = args.warmup, args.warmup + args.steps                     # the timed windows
    out = {
        'metric': METRIC, 'value': round(value, 1), 'unit': 'substep_pairs/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * wall / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'water block 128^3 grid, 200k particles (BASELINE configs[1] inputs), fwd+bwd, evolving (rolling window)',
                   'substeps_per_step': CHUNK, 'timed_pairs': pairs, 'timed_s': round(wall, 3), 'timed_substeps': [w0 * CHUNK, w1 * CHUNK],
                   'n_used': n_used, 'nc_start': int(st0['n_cells_touched']), 'nc_end': int(st1['n_cells_touched']),
                   'n_slow_path': int(st1['n_slow_path']), 'parallelism': '1 env, 1 GPU',
                   'engine_options': options, 'engine_env': {k: v for k, v in os.environ.items() if k.startswith('FE_')},
                   'engine_source_hash': source_hash()},
        'roofline': None, 'pair_roofline': None,
        'forward_only_substeps_per_s': round(fwd_rate, 1),
        'hip_event_ms_per_step': round(ev_ms / args.steps, 3),
    }
    rec = None
    if not args.no_probe:
        n_probe = w1 if args.no_extras else max(w1, max(b for _, b in PHASES.values()))
        # every second timed window is profiled with HIP events on the engine's stream (all of them when there is only one);
        # the others are wall-timed; outside the timed range likewise within the phase windows
        profiled = {w for w in range(n_probe) if (w - w0) % 2 == 1 or args.steps == 1}
        rec = probe_pass(elib, n_probe, range(w0, w1), profiled)
        tw = fold_windows(rec, w0, w1)
        kern = tw['kernels']
        # the kernel with the largest share of the profiled time; p2g and g2p_grad tie to within a per cent on this workload and would swap
        # places from run to run: among those within 2 % of the largest share, the one furthest below the roofline is reported
        share = {k: kern[k]['avg_us'] * kern[k]['launches'] for k in kern if kern[k]['alg_bytes'] > 0}
        dom = min((k for k in share if share[k] >= 0.98 * max(share.values())), key=lambda k: kern[k]['GBps'])
        traffic, rocprof_us = None, None
        try:                                              # PMC passes of this very command line, same windows, same build, same GPU (else: no claim)
            pj = json.load(open(PMC_TRAFFIC))
            same_build = pj.get('source_hash') == source_hash() and pj.get('device') == torch.cuda.get_device_name(0)
            if pj['steps'] == args.steps and pj['warmup'] == args.warmup and same_build:
                traffic = int(pj['timed_region']['kernels'][dom]['traffic_bytes'])
                # the committed rocprofv3 --kernel-trace of the same command, same windows (scripts/gpu_profile.sh wrote both files)
                import csv
                pre = {'p2g': 'k_p2g<true', 'g2p': 'k_g2p<', 'g2p_p2g': 'k_g2p_p2g', 'pgg_g2pg': 'k_pgg_g2pg', 'g2p_grad': 'k_g2p_grad', 'p2g_grad': 'k_p2g_grad', 'grid_op': 'k_grid<', 'grid_op_grad': 'k_grid_grad'}[dom]
                tot = cnt = 0
                for r in csv.DictReader(open(ROCPROF_TIMED)):
                    if r['Name'].replace('void ', '').startswith(pre):
                        tot += float(r['TotalDurationNs']); cnt += int(r['Calls'])
                rocprof_us = round(1e-3 * tot / cnt, 3) if cnt else None
        except Exception:
            pass
        b_pair = 524 * n_used + 204 * tw['nc_mean']
        sorts = kern.get('sort', {}).get('launches', 0)
        fwd_launches = max(1, kern.get('p2g', {}).get('launches', 0) + kern.get('g2p_p2g', {}).get('launches', 0))
        out['config'].update({'nc_mean_timed': tw['nc_mean'], 'nc_min_timed': tw['nc_min'], 'nc_max_timed': tw['nc_max'],
                              'sorts_per_pair': round(sorts / fwd_launches, 3)})
        desc = {'pgg_g2pg': "k_pgg_g2pg: substep f's p2g_grad and substep f-1's g2p_grad in one launch; credited both kernels' SURVEY bytes (192 N + 40 Nc + the 116 N state read)",
                'g2p_p2g': "k_g2p_p2g: substep f-1's g2p and substep f's p2g in one launch; credited both kernels' SURVEY bytes (216 N + 28 Nc)"}.get(dom)
        # (`achieved` is what the contract defines: ALGORITHMIC bytes -- SURVEY 8d's per-unit figure x the launch's units -- over the launch's time, i.e. an
        #  EFFECTIVE bandwidth: the fused launches are credited with both constituent kernels' bytes although they no longer move the 60 + 120 B per particle the
        #  fusion spares, nor the 64-96 B that compact_F does.  `traffic` is what the counters saw move.  ADVICE r5)
        out['roofline'] = {'bound': 'hbm', 'kernel': dom, **({'kernel_is': desc} if desc else {}), 'achieved': kern[dom]['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'achieved_is': 'effective: algorithmic bytes per launch / launch time (bytes the fusions and compact_F no longer move are still credited); see traffic',
                           'frac': round(kern[dom]['GBps'] / HBM_PEAK_GBS, 4), 'frac_of_measured_copy': round(kern[dom]['GBps'] / HBM_COPY_GBS, 4),
                           'traffic': traffic, 'alg_bytes_per_launch': kern[dom]['alg_bytes'], 'avg_launch_us': kern[dom]['avg_us'],
                           # (an event bracket serialises the launches around it: the profiled windows run ~13 % slower than the timed ones, so
                           #  `frac` is the conservative figure; the kernel's own begin-to-end time in the committed trace is given beside it)
                           'rocprof_avg_launch_us': rocprof_us,
                           'frac_rocprof': round(kern[dom]['alg_bytes'] / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rocprof_us else None,
                           'windows': f'HIP events over every second timed window (substeps {w0 * CHUNK}..{w1 * CHUNK}) of an untimed replay; bytes from those windows\' Nc'}
        stored = tw.get('bwd_frames_from_grid_store') or 0.0
        out['pair_roofline'] = {'alg_bytes_per_pair': int(b_pair), 'achieved_GBps': round(b_pair * value / 1e9, 1),
                                'frac': round(b_pair * value / 1e9 / HBM_PEAK_GBS, 4),
                                # (the SURVEY figure credits the backward recompute of grid[f] in full; the per-frame grid store spared it in
                                #  `bwd_frames_from_grid_store` of the frames: without those 44 Nc that nobody moved)
                                'frac_executed': round((b_pair - stored * GRID_RECOMPUTE * tw['nc_mean']) * value / 1e9 / HBM_PEAK_GBS, 4),
                                'bwd_frames_from_grid_store': tw.get('bwd_frames_from_grid_store'),
                                'frac_of_measured_copy': round(b_pair * value / 1e9 / HBM_COPY_GBS, 4),
                                'nc': 'mean over the timed windows (probe replay), not a post-region sample'}
        out['kernels'] = kern
    if not args.no_extras:
        extra = {}
        if rec is not None:
            extra['phase_rates'] = {name: fold_windows(rec, a, b) for name, (a, b) in PHASES.items() if b <= len(rec)}
        # round 1's headline for continuity: the same block restarted from rest every step (nothing moves, the sort is always fresh)
        e2, _ = build_block(elib, 0, L=50)
        for _ in range(3):
            window_step(e2, 50, roll=False)
        e2.sync(); t2 = time.perf_counter()
        for _ in range(20):
            window_step(e2, 50, roll=False)
        e2.sync()
        extra['restart_from_rest_pairs_per_s'] = round(20 * 50 / (time.perf_counter() - t2), 1)
        e2.close()
        # B environments stepped in lockstep through shared launches (fe_step_batch): the same evolving block with B different
        # particle sets; aggregate substep pairs/s of the B scenes against ONE scene on the same schedule (same warm-up, same number
        # of windows: the same part of the fall).  Never part of `value`.
        def batched_rate(B, nb):
            engs = [build_block(elib, 0, seed=sd)[0] for sd in range(B)]
            E = type(engs[0])

            def batch_window():
                E.step_batch(engs, 0, 0, CHUNK, 0)
                for e in engs:
                    e.reset_grad(); e.loss_step_grad(0, CHUNK, 0, 1.0, 1.0)
                E.step_grad_batch(engs, 0, 0, CHUNK, 0)
                for e in engs:
                    e.copy_frame(CHUNK, 0)
            for _ in range(args.warmup):
                batch_window()
            for e in engs:
                e.sync()
            t3 = time.perf_counter()
            for _ in range(nb):
                batch_window()
            for e in engs:
                e.sync()
            rate = B * nb * CHUNK / (time.perf_counter() - t3)
            for e in engs:
                e.close()
            return rate
        nb = max(4, args.steps // 4)
        r1, r4 = batched_rate(1, nb), batched_rate(4, nb)
        extra['batched_envs'] = {'n_envs': 4, 'pairs_per_s_all_envs': round(r4, 1), 'pairs_per_s_one_env_same_schedule': round(r1, 1),
                                 'ratio': round(r4 / r1, 3), 'timed_pairs_per_env': nb * CHUNK,
                                 'note': 'fe_step_batch: one launch per phase for all envs (gridDim.y = n_envs)'}
        from fluidlab_amd import scenes as S
        # the default run's figure (10,000 pairs: the block falls, splashes and settles into a layer) beside `value`, whatever --steps was
        if args.steps == 100 and args.warmup == 5:
            extra['value_full_run'] = {'pairs_per_s': out['value'], 'timed_substeps': [500, 10500], 'note': 'this run'}
        else:
            e3, _ = build_block(elib, 0)
            for _ in range(5):
                window_step(e3, CHUNK)
            e3.sync(); t3 = time.perf_counter()
            for _ in range(100):
                window_step(e3, CHUNK)
            e3.sync()
            extra['value_full_run'] = {'pairs_per_s': round(100 * CHUNK / (time.perf_counter() - t3), 1), 'timed_substeps': [500, 10500],
                                       'note': 'the default `python bench.py` workload (100 steps after 5), run untimed beside the line\'s own timed region'}
            e3.close()
        # the metric's size through the GENERAL kernels (svd3, backward_svd, multi-material stress: compiled out for inviscid liquids), EVOLVING:
        # the same block as the headline made of ICECREAM (plasto-elastic), the same rolling windows (500 warm-up substeps, 2,500 timed pairs with
        # their sorts: the fall and the impact, which comes after 2,200 substeps at this dt).  dt = 1e-4: at the reference's fixed 2e-4 the stiff solid is beyond its Courant limit on a 128^3 grid and leaves the grid
        # within a hundred substeps on every implementation, the oracle included (DESIGN section 6 caveats).
        extra['general_128_200k'] = extra_evolving(elib, 0, 'ICECREAM (plasto-elastic, SVD + plastic clamp + backward_svd) block 128^3, 200k particles, fwd+bwd, evolving (rolling windows)',
                                                   S.ICECREAM, 1e-4, 5, 25, 'GENERAL kernel variants at the size the metric is quoted on; dt halved for stability (see DESIGN)')
        extra['config5_injected_256_1M'] = c5_injected(elib, 0)
        extra['config5_water_256_1M'] = extra_block(elib, 0, 'water block 256^3, 1M particles, fwd+bwd', 256, 1_000_000, S.WATER, 40, 3)
        extra['config5_icecream_256_1M'] = extra_block(elib, 0, 'ICECREAM (plasto-elastic, SVD) block 256^3, 1M particles, fwd+bwd, 10 substeps', 256, 1_000_000, S.ICECREAM, 10, 3)
        out['extra'] = extra
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out))


# ------------------------------------------------------------------------------------------------------------------
# N > 1: LatteArt replicas, one per GPU, action-gradient all-reduce
# ------------------------------------------------------------------------------------------------------------------
C4_SCENES = {
    # BASELINE config 3's scene: 128^3, ~2 particles per cell, 60k milk pool -> 281,883 particles
    'config3': dict(quality=2, particle_density=4e6, n_pool=60000),
    # the reference's own LatteArt-v0 (64^3, 115,480 particles): tests only
    'as_shipped': dict(),
}


def run_replicas(args, rank, local_rank, world):
    import contextlib
    import io
    import torch
    from fluidlab_amd import _capi
    from fluidlab_amd.envs import make
    from fluidlab_amd.optimizer.distributed import EnvParallel
    from fluidlab_amd.optimizer.recorder import Recorder
    from fluidlab_amd.optimizer.solver import Solver
    from fluidlab_amd.utils.config import load_config
    dev = 0 if args.one_device else local_rank
    # every rank on its own cores: the host enqueues ~90 us of launches per 160 us substep pair, i.e. needs most of a core, and
    # the ranks of a node would otherwise migrate over each other's
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    cores = pin_to_cores(local_rank, local_world)
    torch.cuda.set_device(dev)
    par = EnvParallel(backend=args.dist_backend, device=dev, always=True)       # init_process_group('nccl' = RCCL), one rank per GPU
    assert par.world_size == world and par.dist is not None and par.dist.get_world_size() == world
    if args.envs_per_gpu > 1 and args.c4_scene == 'config3' and 'FE_GRID_STORE_GIB' not in os.environ:
        # two config-3 replicas per GPU: 93 GB of frames each; with the per-frame grid store at its default 64 GiB budget they do not fit
        # 288 GB, with 32 GiB (4,096 block slots per frame: the scene's active list has ~2,000 entries) they do
        os.environ['FE_GRID_STORE_GIB'] = '32'
    elib = _capi.load_hip()
    kw = dict(C4_SCENES[args.c4_scene], engine_lib=elib, device=dev)
    if args.c4_window > 0:                                             # the reference's memory model: a window of substeps, checkpoints in host memory, the chunk's forward re-run in backward (mpm:856-912)
        kw.update(max_substeps_local=args.c4_window, ckpt_dest='cpu')
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())        # the env / solver layers print progress lines
    with quiet():
        # the target pattern every replica pours towards: the demo policy's recording (seed 0 on every rank)
        rec = make('LatteArt-v0', seed=0, loss=False, **kw)
        tgt = Recorder(rec).record(write=False)
        rec.taichi_env.simulator.engine.close()                      # its resident trajectory is tens of GB: free it before the replica is built
        del rec
        B = max(1, args.envs_per_gpu)
        envs = [make('LatteArt-v0', seed=1000 + rank * B + i, loss=True, target=tgt, **kw) for i in range(B)]      # injector randomness differs per replica
        env = envs[0]
    for e in envs:                                                    # --opt name=value: engine options of every replica (tuning sweeps)
        for o in args.opt:
            k, v = o.split('=')
            e.taichi_env.simulator.engine.set_option(k, float(v))
    cfg = load_config('configs/exp_latteart.yaml').SOLVER
    # 128^3 sits at the stability edge of the reference's fixed dt: the Adam step is kept small so that W + K
    # passes stay in the stable regime.  Gradient, collective and update are the real ones.
    cfg.optim.lr = cfg.optim.lr * args.c4_lr_scale
    np.random.seed(0)                                                 # identical initial policy on every rank
    solver = Solver(env, None, cfg, parallel=par)
    policy = env.trainable_policy(cfg.optim, cfg.init_range)
    init = env.taichi_env.get_state()
    eng = env.taichi_env.simulator.engine
    n_frames = env.horizon * env.taichi_env.simulator.n_substeps      # substeps of one replica's trajectory
    sub = B * n_frames                                                # substep pairs per pass and rank (all B replicas)
    batch = None
    if B > 1:                                                         # the rank's replicas share every launch (optimizer/batch.py)
        from fluidlab_amd.optimizer.batch import EnvBatch
        batch = EnvBatch(envs)
        inits = [e.taichi_env.get_state()['state'] for e in envs]
    t_comp, t_coll, losses, skipped = [], [], [], [0]

    def one_pass():
        a = time.perf_counter()
        with quiet():
            if batch is None:
                info, g_local = solver.forward_backward(init['state'], policy, env.horizon, env.horizon_action)   # agent.get_grad inside
            else:                                                     # the mean over the rank's replicas; the all-reduce then averages the ranks
                res = batch.forward_backward(inits, [policy] * B, env.horizon, env.horizon_action)
                info = dict(res[0][0]); info['loss'] = float(np.mean([r[0]['loss'] for r in res]))
                g_local = np.mean([r[1] for r in res], axis=0)
        b = time.perf_counter()
        g_mean, (loss_mean,) = par.all_reduce_mean(g_local, [info['loss']])      # the path's one exchange
        c = time.perf_counter()
        if np.isfinite(g_mean).all():
            policy.optimize(g_mean, info)
        else:
            skipped[0] += 1
        t_comp.append(b - a); t_coll.append(c - b); losses.append(float(loss_mean))

    def barrier():
        eng.sync(); torch.cuda.synchronize()
        par.barrier()
        torch.cuda.synchronize()

    # the N = 1 figure of THIS scene, measured in this run: rank 0 alone (the others wait at the barrier), one warm and one timed
    # forward+backward pass, no collective, no update -- the denominator of `scaling_efficiency`
    barrier()
    n1_rate = 0.0
    if rank == 0:
        for k in range(2):
            eng.sync(); a = time.perf_counter()
            with quiet():
                solver.forward_backward(init['state'], policy, env.horizon, env.horizon_action)      # ONE replica, as `--gpus 1 --replicas` would run it
            eng.sync(); n1_rate = sub / B / (time.perf_counter() - a)
    # Where the host's time goes in that flow (is a rank host-bound?): one more pass of rank 0 with every C-ABI call timed.  Calls that wait for the GPU
    # (fe_sync, the read-backs) are `blocked`; everything else the host does -- the calls that only enqueue, and the Python between calls -- is what a faster GPU could not shorten.
    host_us = None
    if rank == 0 and B == 1:
        acc = {}

        class TimedLib:
            def __init__(self, lib): self._lib = lib
            def __getattr__(self, name):
                f = getattr(self._lib, name)
                if not name.startswith('fe_'):
                    return f
                def call(*a):
                    t = time.perf_counter(); r = f(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
                    return r
                return call
        real = eng.lib
        eng.lib = TimedLib(real)
        try:
            eng.sync(); a = time.perf_counter()
            with quiet():
                solver.forward_backward(init['state'], policy, env.horizon, env.horizon_action)
            eng.sync(); wall1 = time.perf_counter() - a
        finally:
            eng.lib = real
        waits = lambda n: n == 'fe_sync' or '_get' in n or n in ('fe_last_error',)
        blocked = sum(v for k, v in acc.items() if waits(k)); enq = sum(v for k, v in acc.items() if not waits(k))
        per = 1e6 / (sub / B)
        host_us = {'wall': round(wall1 * per, 1), 'enqueue_calls': round(enq * per, 1), 'python_between_calls': round((wall1 - blocked - enq) * per, 1),
                   'blocked_waiting_for_the_gpu': round(blocked * per, 1),
                   'largest_calls': {k: round(v * per, 1) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:6]},
                   'note': 'us per substep pair, one Solver pass of rank 0 with every fe_* call timed; host-bound would read blocked ~ 0'}
    barrier()
    for _ in range(args.warmup):
        one_pass()
    del t_comp[:], t_coll[:]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    barrier()
    wall = time.perf_counter() - t0
    dist = par.dist
    cdev = 'cuda' if args.dist_backend == 'nccl' else 'cpu'
    t = torch.tensor([wall], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    rss_gb = host_rss_gb()
    f_stats = min(n_frames, args.c4_window or n_frames) - 1        # a frame the engine holds (the window's last one when the trajectory is chunked)
    hbm_gb = sum(e.taichi_env.simulator.engine.get_stats(f_stats)['bytes_state'] for e in envs) / 2**30
    import hashlib
    dig = hashlib.sha256(np.ascontiguousarray(policy.comp_actions, dtype=np.float64).tobytes()).digest()      # the policy after the last Adam step
    h_lo, h_hi = float(int.from_bytes(dig[:4], 'little')), float(int.from_bytes(dig[4:8], 'little'))
    mine = torch.tensor([sub * args.steps / max(sum(t_comp), 1e-9), 1e6 * float(np.median(t_coll)), float(skipped[0]), rss_gb, hbm_gb, float(len(cores)), h_lo, h_hi],
                        dtype=torch.float64, device=cdev)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    # latency of the collective itself on a warm communicator (the per-pass figure above also waits for the slowest rank)
    probe = torch.zeros((env.horizon_action + 1) * 3 + 1, dtype=torch.float32, device=cdev)
    for _ in range(5):
        dist.all_reduce(probe)
    torch.cuda.synchronize(); par.barrier()
    p0 = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(probe)
    torch.cuda.synchronize()
    ar_us = 1e6 * (time.perf_counter() - p0) / 50
    st = eng.get_stats(f_stats)
    if rank == 0:
        per_rank = [float(a[0]) for a in allr]
        value = world * sub * args.steps / wall_max
        b_pair = 524 * st['n_used'] + 204 * st['n_cells_touched']
        out = {
            'metric': METRIC, 'value': round(value, 1), 'unit': 'substep_pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * wall_max / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{world} LatteArt-v0 replicas ({args.c4_scene}: {"128^3" if args.c4_scene == "config3" else "64^3"} grid, {eng.N} particles each), '
                                   'one per GPU: one step = one Solver pass (forward with loss, backward, action-gradient all-reduce, Adam)',
                       'substep_pairs_per_step_per_rank': sub, 'rccl_world_size': dist.get_world_size(), 'dist_backend': args.dist_backend,
                       'action_grad_shape': [env.horizon_action + 1, 3], 'lr_scale': args.c4_lr_scale, 'window_substeps': args.c4_window or n_frames,
                       'envs_per_gpu': B, 'engine_options': eng.get_options(), 'engine_env': {k: v for k, v in os.environ.items() if k.startswith('FE_')},
                       'parallelism': f'{world * B} env replicas, {B} per GPU' + (' sharing launches (fe_step_batch)' if B > 1 else '') + ', 1 all-reduce of the action gradient per pass'},
            'n1_same_scene_pairs_per_s': round(n1_rate, 1),
            'scaling_efficiency': round(value / (world * n1_rate), 4) if n1_rate > 0 else None,
            'per_rank_pairs_per_s_compute_only': [round(v, 1) for v in per_rank],
            'weak_scaling_efficiency_vs_rank_compute': round(value / sum(per_rank), 4),
            'host': {'rss_gb_per_rank': [round(float(a[3]), 2) for a in allr], 'cores_per_rank': [int(a[5]) for a in allr],
                     'hbm_state_gb_per_rank': [round(float(a[4]), 1) for a in allr], 'cpus_visible': os.cpu_count()},
            'host_time_per_pair_us': host_us,
            'allreduce_us': {'warm_latency': round(ar_us, 1), 'per_pass_median_incl_wait': [round(float(a[1]), 1) for a in allr]},
            'passes_skipped_nonfinite_grad': int(sum(float(a[2]) for a in allr)),
            'actions_identical_across_ranks': all(float(a[6]) == float(allr[0][6]) and float(a[7]) == float(allr[0][7]) for a in allr),
            'loss_mean_over_envs': [round(v, 3) for v in losses[-args.steps:]],
            'pair_roofline': {'alg_bytes_per_pair': b_pair, 'frac_per_gpu': round(b_pair * value / world / 1e9 / HBM_PEAK_GBS, 4)},
            'note': 'N=1 prints the water-block line (the configuration the metric is quoted on): its `value` is a different workload. The '
                    'N=1 rate of THIS scene is n1_same_scene_pairs_per_s (rank 0 alone, same run); scaling_efficiency = value / (n_gpus x that)',
        }
        print(json.dumps(out))
    par.barrier()
    for e in envs:
        e.taichi_env.simulator.engine.close()
    par.close()


def core_share(local_rank, local_world, cpus):
    """Rank `local_rank`'s equal, contiguous share of the cores `cpus` (fewer cores than ranks: everybody gets them all)."""
    cpus = sorted(cpus)
    per = len(cpus) // max(1, local_world)
    return cpus[local_rank * per:(local_rank + 1) * per] if per >= 1 else cpus


def pin_to_cores(local_rank, local_world):
    """Give this rank an equal, contiguous share of the cores the process may run on."""
    try:
        mine = core_share(local_rank, local_world, os.sched_getaffinity(0))
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):
        return []


def host_rss_gb():
    try:
        import psutil
        return psutil.Process().memory_info().rss / 2**30
    except Exception:
        return 0.0


def free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--no-probe', action='store_true', help='skip the untimed probe replay (profiler runs: the trace then holds one trajectory)')
    ap.add_argument('--dist-backend', default='nccl', help="'gloo' + --one-device: exercise the N>1 path on a 1-GPU box (tests)")
    ap.add_argument('--one-device', action='store_true', help='tests only: every rank uses GPU 0')
    ap.add_argument('--replicas', action='store_true', help='run the N > 1 workload (LatteArt replicas + all-reduce) even with one rank')
    ap.add_argument('--c4-scene', default='config3', choices=sorted(C4_SCENES))
    ap.add_argument('--c4-lr-scale', type=float, default=0.1)
    ap.add_argument('--c4-window', type=int, default=0, help='N > 1 workload: max_substeps_local (0 = the whole trajectory resident); tests use 50 so that eight ranks fit one device')
    ap.add_argument('--envs-per-gpu', type=int, default=1, help='N > 1 workload: B replicas per rank stepped in lockstep through fe_step_batch (they have to fit the HBM: the config-3 scene keeps ~150 GB per replica resident)')
    ap.add_argument('--opt', action='append', default=[], help='engine option name=value (tuning sweeps)')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver would
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.execvp(cmd[0], cmd)
    if args.gpus != world:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without a launcher)')
    if world == 1 and not args.replicas:
        args.steps = 100 if args.steps is None else args.steps            # 10,000 substep pairs: about a second of timed region
        args.warmup = 5 if args.warmup is None else args.warmup
        run_single(args)
    else:
        args.steps = 3 if args.steps is None else args.steps
        args.warmup = 1 if args.warmup is None else args.warmup
        run_replicas(args, int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), world)


if __name__ == '__main__':
    main()

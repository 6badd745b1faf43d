#!/usr/bin/env python
"""bench.py -- MPM substep pairs/s (forward + backward) on the BASELINE workload.

Workload (SURVEY 8d C2 = BASELINE.json configs[1] inputs, run forward AND backward as the metric
asks): single-material water block, 128^3 grid, 200,000 particles, x ~ U([0.30,0.53]^3) seed 0,
g=(0,-10,0), CubeBoundary [0.05,0.95]^3.  One "step" = CHUNK forward substeps from frame 0
followed by CHUNK backward substeps (adjoint seeded by the squared-distance loss on the last
frame); all inputs are resident in HBM before the timed region, nothing crosses PCIe inside it.

  python bench.py [--gpus N --steps K --warmup W]
N>1: launched by torch.distributed.run, one env replica per GPU (weak scaling), plus the path's
one real exchange: an RCCL all-reduce of the (horizon_action+1) x action_dim action gradient per
optimisation pass.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

N_GRID, N_PARTICLES, CHUNK = 128, 200_000, 50
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

# algorithmic bytes per launch (DESIGN.md "Roofline accounting"; N = used particles, Nc = touched nodes).
# They sum to SURVEY 8d's B_fwd = 216 N + 72 Nc and B_bwd = 308 N + 132 Nc.
KERNEL_BYTES = {
    'p2g': (156, 16), 'grid_op': (0, 44), 'g2p': (60, 12),
    'p2g_recompute': (116, 16), 'grid_op_keep': (0, 28), 'g2p_grad': (60, 24), 'grid_op_grad': (0, 48), 'p2g_grad': (132, 16),
    'sort': (0, 0), 'reorder_grad': (0, 0),          # overhead of the cell-sorted layout: no algorithmic bytes credited
}
FWD_KERNELS = ('p2g', 'grid_op', 'g2p')


def build_engine(elib, device, n_grid=N_GRID, n_particles=N_PARTICLES, L=CHUNK):
    import scenarios as S
    sc = S.water_block(n_grid=n_grid, n_particles=n_particles, seed=0)
    eng = S.make_engine(elib, sc, max_substeps_local=L, device=device)
    eng.loss_alloc(1)
    rng = np.random.RandomState(1)
    eng.loss_set_target(0, (sc['x'] + rng.normal(0, 0.01, sc['x'].shape)).astype(np.float32))
    return eng, sc


def one_step(eng, chunk, backward=True):
    eng.step(0, 0, chunk, 0)
    if backward:
        eng.reset_grad()
        eng.loss_step_grad(0, chunk, 0, 1.0, 1.0)        # mat 0 = WATER
        eng.step_grad(0, 0, chunk, 0)


def cpu_baseline(budget_s=12.0):
    """The oracle (fp32 build, OpenMP over all host cores) on the same 128^3 / 200k workload:
    a bounded number of forward+backward substep pairs.  Reported, never the target."""
    from fluidlab_amd import _capi
    path = os.path.join(ROOT, 'oracle', '_build', 'libfe_oracle_f32.so')
    elib = _capi.EngineLib(path)
    L = 3
    eng, _ = build_engine(elib, 0, L=L)
    ncpu = os.cpu_count() or 1
    one_step(eng, 1)                                      # warm (page faults)
    # the scatter uses float atomics: more threads is not always faster.  Use the best of a few counts.
    best = None
    for c in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), min(ncpu, 16)}, reverse=True):
        eng.set_option('threads', c)
        t = time.perf_counter(); one_step(eng, 1); t = time.perf_counter() - t
        if best is None or t < best[0]:
            best = (t, c)
    cores = best[1]
    eng.set_option('threads', cores)
    t0 = time.perf_counter()
    pairs = 0
    while True:
        one_step(eng, L)
        pairs += L
        if time.perf_counter() - t0 > budget_s or pairs >= 90:
            break
    dt = time.perf_counter() - t0
    eng.close()
    return {'value': pairs / dt, 'unit': 'substep_pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'{pairs} fwd+bwd substep pairs of the same 128^3/200k water block, oracle fp32 + OpenMP ({cores} of {ncpu} hardware threads, fastest of a short sweep), {dt:.1f}s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-replica-probe', action='store_true')
    ap.add_argument('--dist-backend', default='nccl', help="'gloo' + --one-device: exercise the N>1 control flow on a 1-GPU box")
    ap.add_argument('--one-device', action='store_true', help='testing only: every rank uses GPU 0')
    ap.add_argument('--opt', action='append', default=[], help='engine option name=value (tuning sweeps)')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if args.one_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(local_rank)

    from fluidlab_amd import _capi
    elib = _capi.load_hip()                               # no fallback: raises without the HIP library
    eng, sc = build_engine(elib, local_rank)
    for o in args.opt:
        k, v = o.split('=')
        eng.set_option(k, float(v))
    coll_dev = 'cuda' if args.dist_backend == 'nccl' else 'cpu'
    action_grad = torch.zeros((251, 3), device=coll_dev)  # LatteArt-sized action gradient

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        one_step(eng, CHUNK)
        if dist is not None:
            eng.sync()
            dist.all_reduce(action_grad)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    for _ in range(args.steps):
        step()
    ev_ms = eng.timer_stop_ms()
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # ---- untimed extras (rank 0 reports)
    fwd_rate = prof = stats = two = None
    if rank == 0:
        barrier_local = lambda: (eng.sync(), torch.cuda.synchronize())
        barrier_local()
        t1 = time.perf_counter()
        for _ in range(max(2, args.steps // 2)):
            one_step(eng, CHUNK, backward=False)
        barrier_local()
        fwd_rate = max(2, args.steps // 2) * CHUNK / (time.perf_counter() - t1)
        eng.profile_enable(True)
        for _ in range(2):
            one_step(eng, CHUNK)
        prof = eng.profile_read()
        eng.profile_enable(False)
        stats = eng.get_stats(CHUNK // 2)
        # How much of the chip one 200k-particle scene leaves idle: a second, independent replica of the same scene on the same
        # GPU (its own engine and HIP stream), both driven from this thread.  Reported beside `value`, never part of it.
        two = None
        if world == 1 and not args.no_replica_probe:
            eng2, _ = build_engine(elib, local_rank)
            for e in (eng, eng2):
                one_step(e, CHUNK)
            eng2.sync(); barrier_local()
            n2 = max(3, args.steps // 3)
            t2 = time.perf_counter()
            for _ in range(n2):
                one_step(eng, CHUNK); one_step(eng2, CHUNK)
            eng2.sync(); barrier_local()
            two = 2 * n2 * CHUNK / (time.perf_counter() - t2)
            eng2.close()
    if dist is not None:
        dist.barrier()

    if rank == 0:
        pairs = args.steps * CHUNK * world
        value = pairs / wall
        n_used, nc = stats['n_used'], stats['n_cells_touched']
        per_kernel = {}
        for name, (ms, cnt) in prof.items():
            if cnt:
                bp, bc = KERNEL_BYTES.get(name, (0, 0))
                us = 1e3 * ms / cnt
                b = bp * n_used + bc * nc
                per_kernel[name] = {'avg_us': round(us, 3), 'launches': cnt, 'alg_bytes': b, 'GBps': round(b / (us * 1e-6) / 1e9, 1)}
        # dominant kernel = largest share of the measured time among the kernels that carry algorithmic bytes
        dom = max((k for k in per_kernel if per_kernel[k]['alg_bytes'] > 0), key=lambda k: prof[k][0])
        # HBM traffic of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01final_pmc_traffic.json')))['kernels'][dom]
            traffic = int(pmc['traffic_bytes'])      # (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE correction, see the file's note
        except Exception:
            pass
        b_pair = 524 * n_used + 204 * nc
        out = {
            'metric': 'MPM substeps/sec (fwd+bwd), 128^3 grid / 200k particles', 'value': round(value, 1),
            'unit': 'substep_pairs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * wall / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'water block 128^3 grid, 200k particles (BASELINE configs[1] inputs), fwd+bwd',
                       'substeps_per_step': CHUNK, 'n_used': n_used, 'n_cells_touched': nc,
                       'parallelism': f'{world} env replica(s), one per GPU' + (', all-reduce of 251x3 action grad per step' if world > 1 else '')},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': per_kernel[dom]['GBps'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(per_kernel[dom]['GBps'] / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'alg_bytes_per_launch': per_kernel[dom]['alg_bytes'], 'avg_launch_us': per_kernel[dom]['avg_us']},
            'pair_roofline': {'alg_bytes_per_pair': b_pair, 'achieved_GBps': round(b_pair * value / world / 1e9, 1),
                              'frac': round(b_pair * value / world / 1e9 / HBM_PEAK_GBS, 4)},
            'forward_only_substeps_per_s': round(fwd_rate, 1),
            'two_replicas_one_gpu': None if two is None else {'value': round(two, 1), 'unit': 'substep_pairs/s (both scenes)', 'ratio_to_value': round(two / value, 3),
                                                              'note': 'two independent engines/streams on this GPU; not part of `value`'},
            'hip_event_ms_per_step_rank0': round(ev_ms / args.steps, 3),
            'kernels': per_kernel,
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""bench.py's launch contract: `--gpus N` without a launcher starts N ranks itself; a launcher / flag mismatch is an error; on the
GPU box two ranks run the real N > 1 path (LatteArt replicas, action-gradient all-reduce) sharing the one device over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(MASTER_ADDR='127.0.0.1', **kw)
    return env


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3'], env=_env(WORLD_SIZE='1'), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and '--gpus 3 but WORLD_SIZE=1' in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_spawns_two_ranks_and_all_reduces_the_action_gradient():
    """`python bench.py --gpus 2 ...` exactly as the driver would call it (no launcher around it): two ranks, one LatteArt-v0 replica
    each with its own injector randomness, Solver passes with the action-gradient all-reduce.  The box has one GPU: both ranks use
    it and the collective runs over gloo (the only test-only switches); the scene is the reference's own 64^3 LatteArt."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--dist-backend', 'gloo', '--one-device', '--c4-scene', 'as_shipped']
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['rccl_world_size'] == 2 and out['steps'] == 2 and out['scaling'] == 'weak'
    assert out['value'] > 0 and len(out['per_rank_pairs_per_s_compute_only']) == 2
    assert out['passes_skipped_nonfinite_grad'] == 0 and len(out['loss_mean_over_envs']) == 2
    assert all(v == v and v > 0 for v in out['loss_mean_over_envs'])
    assert out['config']['substep_pairs_per_step_per_rank'] == 3300 and out['config']['action_grad_shape'] == [251, 3]
    assert 0.2 < out['weak_scaling_efficiency_vs_rank_compute'] <= 1.05
    assert out['actions_identical_across_ranks'] is True                      # same averaged gradient, same fp64 Adam step on both ranks
    assert out['n1_same_scene_pairs_per_s'] > 0 and 0.1 < out['scaling_efficiency'] <= 1.1      # (two ranks share one GPU here)
    assert len(out['host']['rss_gb_per_rank']) == 2 and all(c >= 1 for c in out['host']['cores_per_rank'])


@pytest.mark.gpu
def test_bench_replica_path_over_rccl_with_one_rank():
    """The N > 1 workload on the real collective backend: one rank, `nccl` (= RCCL) process group bound to the device, the action
    gradient and the timing reductions as device tensors.  What the one-GPU box can show of the path the driver runs on eight."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--replicas', '--steps', '1', '--warmup', '1', '--c4-scene', 'as_shipped']
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['config']['dist_backend'] == 'nccl' and out['config']['rccl_world_size'] == 1 and out['n_gpus'] == 1
    assert out['value'] > 0 and out['passes_skipped_nonfinite_grad'] == 0 and out['allreduce_us']['warm_latency'] > 0


@pytest.mark.gpu
def test_bench_gpus_2_over_rccl_when_the_box_has_two_gpus():
    """The first box with two GPUs exercises the path the driver runs on eight: `python bench.py --gpus 2` with the default `nccl`
    (= RCCL) backend, one rank per GPU, LatteArt replicas, the action-gradient all-reduce -- and every rank must end up with the same
    policy (the averaged gradient and the fp64 Adam step are identical by construction)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (the 1-GPU box runs the gloo variant above and the one-rank RCCL group below)')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--c4-scene', 'as_shipped']
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['config']['dist_backend'] == 'nccl' and out['config']['rccl_world_size'] == 2 and out['n_gpus'] == 2
    assert out['value'] > 0 and out['passes_skipped_nonfinite_grad'] == 0
    assert out['actions_identical_across_ranks'] is True
    assert out['n1_same_scene_pairs_per_s'] > 0 and 0.2 < out['scaling_efficiency'] <= 1.1


@pytest.mark.gpu
def test_bench_replicas_with_two_envs_per_gpu():
    """`--envs-per-gpu 2`: the rank's two replicas step in lockstep through fe_step_batch (optimizer/batch.py), their gradients are
    averaged before the rank's all-reduce.  One rank over RCCL, the reference's own 64^3 scene (two of them fit the HBM resident)."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--replicas', '--envs-per-gpu', '2', '--steps', '1', '--warmup', '1', '--c4-scene', 'as_shipped']
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['config']['envs_per_gpu'] == 2 and out['config']['substep_pairs_per_step_per_rank'] == 6600
    assert out['value'] > 0 and out['passes_skipped_nonfinite_grad'] == 0 and out['loss_mean_over_envs'][0] > 0
    assert out['scaling_efficiency'] >= 0.9        # two replicas per launch are no slower per replica than one alone (measured ~1.2)


@pytest.mark.gpu
def test_bench_gpus_8_as_the_driver_launches_it_on_one_device():
    """Eight ranks -- the driver's node -- before the driver runs them: `python bench.py --gpus 8` spawns its eight ranks, every rank a
    LatteArt-v0 replica with its own injector noise, one all-reduce of the (251 x 3) action gradient per Solver pass.  The box has one
    GPU: all ranks share it, the collective runs over gloo, and the trajectory is kept as a 50-substep window with host checkpoints
    (`--c4-window`, the reference's memory model) so that eight replicas fit.  All eight policies must come out bit-identical."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '1',
           '--dist-backend', 'gloo', '--one-device', '--c4-scene', 'as_shipped', '--c4-window', '50']
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=2400, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 8 and out['config']['rccl_world_size'] == 8 and out['config']['window_substeps'] == 50
    assert out['value'] > 0 and len(out['per_rank_pairs_per_s_compute_only']) == 8 and out['passes_skipped_nonfinite_grad'] == 0
    assert out['actions_identical_across_ranks'] is True
    cores = out['host']['cores_per_rank']
    assert len(cores) == 8 and all(c >= 1 for c in cores) and sum(cores) <= out['host']['cpus_visible']

"""Data parallelism over environments (the reference is single-process, solver.py:17-72).

One process per GPU, one environment replica per process.  The only exchange on the path is the action
gradient -- (horizon_action+1) x action_dim float32, ~3 KB for LatteArt -- summed with one all-reduce per
optimisation pass (RCCL over xGMI when the backend is 'nccl'; 'gloo' on CPU for the tests).  It is latency
bound, so it is issued as a single small collective; every rank then applies the same fp64 Adam step and the
policies stay bit-identical without a broadcast."""
import os

import numpy as np


class EnvParallel:
    def __init__(self, backend=None, device=None, always=False):
        self.world_size = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.dist = None
        self.device = device
        if self.world_size > 1 or always:             # always: a process group even for one rank (bench.py --replicas)
            import torch
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:       # one rank without a launcher
                import socket
                with socket.socket() as sk:
                    sk.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
                os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            self.backend = backend
            if not dist.is_initialized():
                if backend == 'nccl':
                    torch.cuda.set_device(self.local_rank)
                    dist.init_process_group(backend, device_id=torch.device('cuda', self.local_rank))
                else:
                    dist.init_process_group(backend)
            self.dist = dist
            self._torch = torch
            self._dev = torch.device('cuda', self.local_rank) if backend == 'nccl' else torch.device('cpu')

    def all_reduce_mean(self, grad, scalars=()):
        """Average the action gradient (and a few loss scalars, for logging) over the replicas."""
        if self.dist is None:
            return np.asarray(grad), list(scalars)
        g = np.ascontiguousarray(grad, dtype=np.float32)
        buf = np.concatenate([g.ravel(), np.asarray(scalars, dtype=np.float32)])
        t = self._torch.from_numpy(buf).to(self._dev)
        self.dist.all_reduce(t)                        # one collective per optimisation pass
        out = (t / self.world_size).cpu().numpy()
        return out[:g.size].reshape(g.shape), list(out[g.size:])

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()

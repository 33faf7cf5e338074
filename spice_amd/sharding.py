"""Session sharding and the (tiny) cross-rank exchange of the multi-GPU path.

Display sessions are independent: GLZ state is per client (all display channels
of a client share one dictionary, server/red_worker.c:9677-9754), MJPEG state
per stream.  A session is therefore pinned to one rank / GPU and the data path
needs no collective.  Ranks only meet for the barrier round a timed region, the
max of their wall times and a gather of the per-codec counters
(stat_info_t, server/red_worker.c:183-193).  On GPUs the backend is RCCL
("nccl"); the same code runs on "gloo" for CPU tests.
"""
import os


def owner_of(client_id, world):
    """Rank that owns a client's sessions (stable, spreads consecutive ids)."""
    return (client_id * 2654435761 >> 7) % world if world > 1 else 0


def sessions_of(rank, world, client_ids):
    return [c for c in client_ids if owner_of(c, world) == rank]


class Dist:
    """Barrier / max / sum / gather over the ranks; trivial for one process."""

    def __init__(self, backend=None, use_cuda=True):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        import torch
        self.torch = torch
        self.use_cuda = use_cuda and torch.cuda.is_available()
        if self.use_cuda:
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        else:
            self.dev = torch.device("cpu")
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            backend = backend or ("nccl" if self.use_cuda else "gloo")
            kw = {"device_id": self.dev} if backend == "nccl" else {}
            dist.init_process_group(backend=backend, **kw)

    def barrier(self):
        if self.dist:
            self.dist.barrier()
        if self.use_cuda:
            self.torch.cuda.synchronize()

    def _reduce(self, v, op):
        if not self.dist:
            return float(v)
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, v):
        return self._reduce(v, self.dist.ReduceOp.MAX if self.dist else None)

    def sum(self, v):
        return self._reduce(v, self.dist.ReduceOp.SUM if self.dist else None)

    def gather_i64(self, values):
        """all_gather of a small fixed-length int64 vector; returns one list per rank."""
        if not self.dist:
            return [list(values)]
        t = self.torch.tensor(values, dtype=self.torch.int64, device=self.dev)
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.tolist() for o in outs]

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()

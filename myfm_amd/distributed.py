"""Row-sharded multi-GPU training: one process per GPU, `torch.distributed` over RCCL.

Every rank holds a contiguous slice of the training rows and a replica of the model; the device path
asks for ONE all-reduce per level of every sweep (plus the block statistics and sum e / sum e^2). The
all-reduce is `torch.distributed.all_reduce` on a zero-copy tensor view of the library's device buffer,
issued on the stream the library's kernels run on, so it is ordered with them without host syncs.
"""
import numpy as np
from scipy import sparse as sps


def row_range(n_rows: int, rank: int, world: int):
    """Contiguous, balanced partition of [0, n_rows)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rows(X, y, X_rel, rank: int, world: int, make_block=None):
    """This rank's slice of (X, y, relation blocks). `X_rel` is a list of RelationBlock or of
    (original_to_block, csr) pairs; blocks are returned in the same form (`make_block(map, csr)` builds a
    RelationBlock when given)."""
    n = X.shape[0] if X is not None else len(_map_of(X_rel[0]))
    lo, hi = row_range(n, rank, world)
    Xl = sps.csr_matrix(X)[lo:hi] if X is not None else None
    yl = np.asarray(y)[lo:hi]
    rel = []
    for b in X_rel:
        m, data = np.asarray(_map_of(b))[lo:hi], _data_of(b)
        rel.append(make_block([int(v) for v in m], data) if make_block else (m, data))
    return Xl, yl, rel, lo, n


def _map_of(b):
    return b.original_to_block if hasattr(b, "original_to_block") else b[0]


def _data_of(b):
    return b.data if hasattr(b, "data") else b[1]


class _DevView:
    """__cuda_array_interface__ view of `count` doubles at device address `ptr`."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchAllReduce:
    """all-reduce callback for `_myfm.GibbsSession(allreduce=...)` / `mfm_set_allreduce`."""

    def __init__(self, group=None, stream=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.calls = 0
        self.doubles = 0

    @property
    def stream_ptr(self):
        return self.stream.cuda_stream

    def __call__(self, ptr, count):
        t = self.torch.as_tensor(_DevView(ptr, count), device="cuda")
        with self.torch.cuda.stream(self.stream):
            self.dist.all_reduce(t, group=self.group)
        self.calls += 1
        self.doubles += int(count)

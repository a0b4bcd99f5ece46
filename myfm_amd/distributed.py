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

    def __init__(self, group=None, stream=None, device=None):
        import os

        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        # the stream and the tensor views must live on the device of the library's context (MYFM_AMD_DEVICE, else torch's
        # current device) -- not on whatever device torch happens to have current in this process
        if device is None:
            device = int(os.environ["MYFM_AMD_DEVICE"]) if "MYFM_AMD_DEVICE" in os.environ else torch.cuda.current_device()
        self.device = int(device)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self.calls = 0
        self.doubles = 0

    @property
    def stream_ptr(self):
        return self.stream.cuda_stream

    def __call__(self, ptr, count):
        with self.torch.cuda.device(self.device):
            t = self.torch.as_tensor(_DevView(ptr, count), device=self.torch.device("cuda", self.device))
            with self.torch.cuda.stream(self.stream):
                self.dist.all_reduce(t, group=self.group)
        self.calls += 1
        self.doubles += int(count)


# ---- native provider: the library calls RCCL itself (mfm_comm_init) -------------------------------------------------
_STATE = {"on": False, "group": None, "native": True, "peer_exchange": False}


def enable(group=None, set_device=True, native=True, peer_exchange=False):
    """Make `MyFM*.fit()` row-sharded over the ranks of `group` (default: the world group of an initialised
    torch.distributed): every rank calls fit() with the SAME full data, trains on its contiguous slice of the rows on its
    own GPU (LOCAL_RANK) with the all-reduces issued by libmyfm_hip.so through RCCL, and ends with the same samples.

    native=False: the library calls back into `torch.distributed.all_reduce` of `group` instead of opening its own RCCL
    communicator (any backend -- e.g. gloo with several ranks on one GPU, which RCCL refuses).

    peer_exchange=True: two-field one-hot tables train with the persistent sweep on every rank, the ranks' item sums exchanged
    inside the launch through IPC-mapped buffers (`connect_peers`) instead of one all-reduce per factor. Opt-in:
    the path has been tested with the ranks side by side on one GPU only."""
    import os

    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (init_process_group first)")
    _STATE.update(on=True, group=group, native=bool(native), peer_exchange=bool(peer_exchange))
    if set_device and "MYFM_AMD_DEVICE" not in os.environ:
        os.environ["MYFM_AMD_DEVICE"] = os.environ.get("LOCAL_RANK", "0")


def disable():
    _STATE.update(on=False, group=None)


def active():
    if not _STATE["on"]:
        return False
    import torch.distributed as dist

    return dist.is_initialized() and dist.get_world_size(_STATE["group"]) > 1


def comm_kwargs():
    """Keyword arguments of `_myfm.create_train_fm_sharded` that select the all-reduce provider."""
    group = _STATE["group"]
    extra = dict(peer_connect=lambda handle: connect_peers(handle, group)) if _STATE["peer_exchange"] else {}
    if _STATE["native"]:
        return dict(comm_id=native_comm_id(group), **extra)
    ar = TorchAllReduce(group=group)
    return dict(allreduce=ar, stream=ar.stream_ptr, **extra)


def rank_world():
    import torch.distributed as dist

    return dist.get_rank(_STATE["group"]), dist.get_world_size(_STATE["group"])


def native_comm_id(group=None):
    """The 128-byte RCCL id of a new communicator: created on rank 0 (ncclGetUniqueId through the library), broadcast over
    the process group (any backend)."""
    import torch.distributed as dist

    from . import _myfm

    box = [_myfm.comm_unique_id() if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return box[0]


def shard_cuts(first_col, world):
    """Row boundaries of `world` contiguous shards of a table whose rows are sorted by their first stored column
    (`first_col[t]`): balanced, snapped to the nearest boundary between two first-level columns so that (almost) every
    such column has all its rows on one rank. Returns world + 1 ascending row indices."""
    first_col = np.asarray(first_col)
    n = first_col.shape[0]
    edges = np.concatenate([[0], np.flatnonzero(first_col[1:] != first_col[:-1]) + 1, [n]])
    cuts = [0]
    for r in range(1, world):
        want = (n * r) // world
        k = int(np.searchsorted(edges, want))
        cand = [edges[max(k - 1, 0)], edges[min(k, len(edges) - 1)]]
        best = min(cand, key=lambda e: abs(int(e) - want))
        cuts.append(max(int(best), cuts[-1]))
    cuts.append(n)
    return cuts


def connect_peers(session, group=None):
    """Row-sharded persistent sweep: when `mfm_finalize` has built the sweep's layout on every rank
    (`session.peer_info()[0]`), exchange the ranks' IPC handles of their exchange buffers over `torch.distributed` and install
    them (`mfm_peer_export` / `mfm_peer_import`). Every rank must call this at the same point. Returns True when the persistent
    sweep is live on all ranks; on any failure every rank stays with the per-factor passes (same chain)."""
    import torch.distributed as dist

    _STATE["last_connect"] = False
    pending = bool(session.peer_info()[0])
    if not pending:  # (the same on every rank: mfm_finalize agrees on it)
        return False
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    try:
        mine, err = session.peer_export(), ""
    except Exception as ex:  # noqa: BLE001 -- every rank must take part in the gather below
        mine, err = b"", "%s: %s" % (type(ex).__name__, ex)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine, group=group)
    ok = all(len(h) == 256 for h in gathered)
    if ok:
        try:
            session.peer_import(world, rank, b"".join(gathered))
        except Exception as ex:  # noqa: BLE001
            ok, err = False, "%s: %s" % (type(ex).__name__, ex)
    oks = [None] * world
    dist.all_gather_object(oks, bool(ok), group=group)
    if not all(oks):
        if err:
            import sys

            print("myfm_amd.distributed.connect_peers: rank %d: %s" % (rank, err), file=sys.stderr)
        session.peer_drop()
        return False
    _STATE["last_connect"] = True  # (diagnostics: did the last fit / session take the in-launch exchange?)
    return True

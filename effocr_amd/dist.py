"""Multi-GPU recognizer: one process per GPU, crops sharded by rank, results all-gathered.

The reference's inference is single-device (infer_effocr.py:439 ``--device``) and has no collective
call site; the path shards embarrassingly over crops (every row of infer_effocr.py:313-319 is
independent), so each rank runs encoder + k-NN on its contiguous slice with the encoder weights and
the glyph index REPLICATED per GPU, and a single ``all_gather`` (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests) assembles the per-rank top-k ids — ``B/P * k * 8`` bytes per rank,
latency-bound.  No all-reduce exists on this path.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world_size):
    """Contiguous, balanced split of ``n`` crops: the first ``n % P`` ranks get one extra crop."""
    base, rem = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n, world_size):
    return [shard_bounds(n, r, world_size)[1] - shard_bounds(n, r, world_size)[0] for r in range(world_size)]


def all_gather_rows(local, n_total, group=None):
    """All-gather row blocks of unequal length: ``local`` [n_r, k] -> [n_total, k] on every rank, in
    rank order (= original crop order for ``shard_bounds`` slices).  Blocks are padded to the
    largest shard so that one fixed-size collective is used."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group) if _has_into_tensor(local) else \
        _all_gather_list(out, pad, world, mx, group)
    parts = [out[r * mx: r * mx + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def _has_into_tensor(t):
    # gloo (CPU tests) lacks all_gather_into_tensor on some builds; nccl/RCCL has it
    return t.is_cuda and dist.get_backend() == "nccl"


def _all_gather_list(out, pad, world, mx, group):
    if out.is_cuda:                                   # gloo with device tensors (debugging on a shared GPU): stage through the host
        host = [torch.empty((mx,) + tuple(pad.shape[1:]), dtype=pad.dtype) for _ in range(world)]
        dist.all_gather(host, pad.cpu().contiguous(), group=group)
        out.copy_(torch.cat(host, dim=0))
        return
    chunks = [out[r * mx:(r + 1) * mx] for r in range(world)]
    dist.all_gather(chunks, pad.contiguous(), group=group)


class ShardedRecognizer:
    """Wraps a per-rank ``neighbors(crops) -> (distances, indices)`` callable (e.g.
    ``effocr_amd.pipeline.Recognizer.neighbors``): every rank passes the SAME full batch (or only
    its own slice with ``presharded=True``) and gets the full ``[B,k]`` results back."""

    def __init__(self, neighbors_fn, group=None):
        self.neighbors_fn = neighbors_fn
        self.group = group

    def _rank_world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def __call__(self, crops, n_total=None, presharded=False):
        rank, world = self._rank_world()
        if presharded:
            if n_total is None:
                raise ValueError("presharded=True needs n_total")
            local = crops
        else:
            n_total = len(crops)
            lo, hi = shard_bounds(n_total, rank, world)
            local = crops[lo:hi]
        d, i = self.neighbors_fn(local)
        return all_gather_rows(d, n_total, self.group), all_gather_rows(i, n_total, self.group)

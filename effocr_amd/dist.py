"""Multi-GPU recognizer: one process per GPU, crops sharded by rank, results all-gathered.

The reference's inference is single-device (infer_effocr.py:439 ``--device``) and has no collective
call site; the path shards embarrassingly over crops (every row of infer_effocr.py:313-319 is
independent), so each rank runs encoder + k-NN on its contiguous slice with the encoder weights and
the glyph index REPLICATED per GPU, and a single ``all_gather`` (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests) assembles the per-rank top-k ids — ``B/P * k * 8`` bytes per rank,
latency-bound.  No all-reduce exists on this path.  ``ShardedIndexSearch`` is the optional variant that shards the index rows instead
 and merges per-shard top-k lists exactly.
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


def all_gather_rows(local, n_total, group=None, always_collective=False):
    """All-gather row blocks of unequal length: ``local`` [n_r, k] -> [n_total, k] on every rank, in
    rank order (= original crop order for ``shard_bounds`` slices).  Blocks are padded to the
    largest shard so that one fixed-size collective is used.  A group of one rank returns ``local``
    without a collective unless ``always_collective`` (tests: the RCCL call on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if dist.get_world_size(group) == 1 and not always_collective:
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


def pack_topk(d, i):
    """(scores fp32 [n,k], ids int64 [n,k]) -> ONE int32 buffer [n, 3k] (score bits | id low/high words): the per-rank result
    travels in a single collective, 12 bytes per neighbour, bit for bit."""
    d = d.contiguous()
    i = i.contiguous()
    if d.dtype != torch.float32 or i.dtype != torch.int64 or d.shape != i.shape:
        raise ValueError("pack_topk expects float32 scores and int64 ids of the same shape")
    n, k = d.shape                                   # explicit sizes: a rank's shard may be EMPTY (fewer crops than ranks)
    if n == 0:
        return torch.empty((0, 3 * k), dtype=torch.int32, device=d.device)
    return torch.cat([d.view(torch.int32).reshape(n, k), i.view(torch.int32).reshape(n, 2 * k)], dim=1)


def unpack_topk(buf, k):
    """Inverse of ``pack_topk``: int32 [n, 3k] -> (fp32 [n,k], int64 [n,k])."""
    n = buf.shape[0]
    if n == 0:
        return (torch.empty((0, k), dtype=torch.float32, device=buf.device), torch.empty((0, k), dtype=torch.int64, device=buf.device))
    d = torch.empty((n, k), dtype=torch.int32, device=buf.device).copy_(buf[:, :k]).view(torch.float32)   # fresh buffers: a view of a
    # sliced row keeps the slice's offset / stride, which an int32 -> int64 reinterpretation rejects
    i = torch.empty((n, 2 * k), dtype=torch.int32, device=buf.device).copy_(buf[:, k:]).view(torch.int64)
    return d, i


def all_gather_topk(d, i, n_total, group=None, always_collective=False):
    """ONE all-gather of the packed ``(score, id)[n_r, k]`` lists -> ``([n_total,k] fp32, [n_total,k] int64)`` on every rank."""
    k = d.shape[1]
    if k == 0:
        return all_gather_rows(d, n_total, group, always_collective), all_gather_rows(i, n_total, group, always_collective)
    return unpack_topk(all_gather_rows(pack_topk(d, i), n_total, group, always_collective), k)


class ShardedRecognizer:
    """Wraps a per-rank ``neighbors(crops) -> (distances, indices)`` callable (e.g.
    ``effocr_amd.pipeline.Recognizer.neighbors``): every rank passes the SAME full batch (or only
    its own slice with ``presharded=True``) and gets the full ``[B,k]`` results back.

    ``status_fn`` (optional; derived from a bound ``Recognizer.neighbors``): called on every rank AFTER the gather — the call's
    synchronisation point — and raises if this rank's encoder produced a non-finite embedding (f16 operand overflow,
    EFFOCR_EOVERFLOW); the collectives have completed on every rank by then, so a raising rank leaves no peer waiting."""

    def __init__(self, neighbors_fn, group=None, status_fn=None):
        self.neighbors_fn = neighbors_fn
        self.group = group
        if status_fn is None:
            owner = getattr(neighbors_fn, "__self__", None)
            enc = getattr(owner, "recongizer_encoder", None)
            if enc is not None:
                from .pipeline import check_encoder_status
                status_fn = lambda: check_encoder_status(enc)      # noqa: E731
        self.status_fn = status_fn

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
        out = all_gather_topk(d, i, n_total, self.group)          # ONE collective: scores and ids packed, 12 bytes per neighbour
        if self.status_fn is not None:
            self.status_fn()
        return out


def merge_topk(dists, ids, k):
    """Exact merge of per-shard top-k lists: ``dists`` / ``ids`` [P, B, k'] (k' >= what each shard returned; padding = (-FLT_MAX, -1)) ->
    the k best of every query's P * k' candidates in the product's order — score descending, equal scores by ascending GLOBAL id
    (the tie rule of ``effocr_knn_ip_topk``), padding last.  Two stable sorts: by id, then by score."""
    P, B, kk = dists.shape
    d = dists.permute(1, 0, 2).reshape(B, P * kk)
    i = ids.permute(1, 0, 2).reshape(B, P * kk)
    big = torch.iinfo(torch.int64).max
    order = torch.sort(torch.where(i < 0, torch.full_like(i, big), i), dim=1, stable=True)[1]      # ids ascending, padding last
    d, i = torch.gather(d, 1, order), torch.gather(i, 1, order)
    order = torch.sort(d, dim=1, descending=True, stable=True)[1]
    d, i = torch.gather(d, 1, order)[:, :k], torch.gather(i, 1, order)[:, :k]
    if d.shape[1] < k:                                   # fewer candidates than k in total: faiss pads (-FLT_MAX, -1)
        padn = k - d.shape[1]
        d = torch.cat([d, torch.full((B, padn), torch.finfo(torch.float32).min, dtype=d.dtype, device=d.device)], 1)
        i = torch.cat([i, torch.full((B, padn), -1, dtype=i.dtype, device=i.device)], 1)
    return d.contiguous(), i.contiguous()


class ShardedIndexSearch:
    """SURVEY 8(e)'s variant for indexes that outgrow one GPU: the glyph INDEX is sharded by rows instead of the crops.  Rank r
    holds rows ``shard_bounds(ntotal, r, P)`` in its own ``IndexFlatIP``; every rank searches ALL queries against its shard
    (``search_fn(q, k) -> (distances [B,k], LOCAL ids [B,k])``, e.g. ``index.search_device``), ONE all-gather of the
    ``(score fp32, global id int64)[B,k]`` lists (12 k B bytes per rank: 120 KiB at 1024 queries, k = 10) and the exact merge above give
    every rank the result of the single-index search — scores are the same fmaf chains, ties the same ascending-id rule, so ids and
    scores are bit-identical to one big index (``tests/test_dist_gloo.py``).  Not needed at BASELINE configs[3]'s 3 GB."""

    def __init__(self, search_fn, row_offset, group=None):
        self.search_fn = search_fn
        self.row_offset = int(row_offset)
        self.group = group

    def __call__(self, queries, k):
        d, i = self.search_fn(queries, k)
        i = torch.where(i >= 0, i + self.row_offset, i)      # local row -> global row; padding stays -1
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return d, i
        world = dist.get_world_size(self.group)
        B = d.shape[0]
        dg, ig = all_gather_topk(d, i, B * world, self.group)                    # equal blocks of B rows per rank, one collective
        return merge_topk(dg.reshape(world, B, k), ig.reshape(world, B, k), k)


def all_gather_texts(local_pairs, group=None):
    """All-gather of the per-rank transcriptions: ``local_pairs`` = list of (key, text) -> the concatenation over ranks in rank
    order, on every rank.  One collective (torch's object all-gather: the pickled strings travel as byte tensors over RCCL when
    the backend is "nccl", over gloo in the CPU tests); a few hundred bytes per text line — latency-bound, like the id gather."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(local_pairs)
    world = dist.get_world_size(group)
    out = [None] * world
    dist.all_gather_object(out, list(local_pairs), group=group)
    return [p for part in out for p in part]


class ShardedPipeline:
    """BASELINE configs[4] on N GPUs: text-LINE images are the independent units of ``run_effocr`` (a line's boxes, crops and
    characters depend on nothing else), so rank r runs the whole pipeline — localizer, crops, recognizer, k-NN, post-processing —
    on the contiguous slice ``shard_bounds(n_lines, r, N)`` of the line list with every model and the glyph index replicated,
    and ONE all-gather of the transcriptions (strings) gives every rank the full result.  No other collective.

    ``run_fn(images) -> (inference_results dict, anything)`` is a per-rank closure over ``effocr_amd.pipeline.run_effocr``
    (engines on this rank's device).  In-memory images are keyed by their GLOBAL position in the input list."""

    def __init__(self, run_fn, group=None):
        self.run_fn = run_fn
        self.group = group

    def __call__(self, coco_images):
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(self.group), dist.get_world_size(self.group)
        else:
            rank, world = 0, 1
        lo, hi = shard_bounds(len(coco_images), rank, world)
        local = list(coco_images[lo:hi])
        results = self.run_fn(local)[0] if local else {}
        pairs = []
        for j, img in enumerate(local):
            key = img if isinstance(img, str) else j             # run_effocr keys arrays by LOCAL position
            pairs.append((img if isinstance(img, str) else lo + j, results[key]))
        merged = {}
        for k, v in all_gather_texts(pairs, self.group):
            merged[k] = v
        return merged

"""Multi-GPU sharding of the resquiggle path: reads are independent, so the only shared state is
a host-side work queue of batches.  One process per GPU (torch.distributed launch: RANK /
LOCAL_RANK / WORLD_SIZE); each process owns one engine and pulls batch indices from an atomic
counter in the process group's key-value store -- no collective on the data path.
Results are gathered to rank 0 with gather_object (control plane only).

Replaces the reference's `resquiggle_all_reads` worker pool (resquiggle.py:1859-1950,
multiprocessing pipes/queues) for the compute part only; FAST5 I/O and mapping stay outside.
"""
import os


class BatchQueue(object):
    """Dynamic work queue over `n_batches` batch indices shared by all ranks."""

    def __init__(self, n_batches, store=None, key='tombo_amd/next_batch'):
        self.n_batches = int(n_batches)
        self.key = key
        self.store = store
        self._local = 0

    def next(self):
        if self.store is None:          # single process
            i = self._local
            self._local += 1
        else:
            i = self.store.add(self.key, 1) - 1
        return i if i < self.n_batches else None

    def __iter__(self):
        while True:
            i = self.next()
            if i is None:
                return
            yield i


def split_batches(n_reads, batch_size):
    """[(lo, hi), ...] contiguous read ranges of at most batch_size reads"""
    return [(lo, min(lo + batch_size, n_reads)) for lo in range(0, n_reads, batch_size)]


def default_store():
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    from torch.distributed import distributed_c10d
    return distributed_c10d._get_default_store()


def resquiggle_sharded(map_results, process_batch, batch_size=2048, gather=True, queue_key=None):
    """Run `process_batch(list_of_map_results) -> list_of_results` over all reads, batches pulled
    dynamically by every rank.  Returns the full ordered result list on rank 0 (None elsewhere)
    when `gather`, else {batch_index: results} of this rank.

    With the engine: process_batch = lambda mrs: resquiggle_batch(mrs, std_ref, params, ...).
    """
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    ranges = split_batches(len(map_results), batch_size)
    key = queue_key or 'tombo_amd/next_batch/%d' % len(map_results)
    q = BatchQueue(len(ranges), default_store() if distributed else None, key)
    mine = {}
    for b in q:
        lo, hi = ranges[b]
        mine[b] = process_batch(map_results[lo:hi])
    if not gather:
        return mine
    if not distributed:
        parts = [mine]
    else:
        parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
        dist.gather_object(mine, parts, dst=0)
        if dist.get_rank() != 0:
            return None
    merged = {}
    for p in parts:
        merged.update(p)
    out = []
    for b in range(len(ranges)):
        out.extend(merged[b])
    return out


def local_device():
    """HIP ordinal of this process (one process per GPU)."""
    return int(os.environ.get('LOCAL_RANK', '0'))

"""Multi-GPU sharding of a resquiggle job: one process per GPU, batches pulled from a shared
work queue, no collective on the data path.

The reference shards reads over worker processes through multiprocessing queues
(`resquiggle_all_reads`, tombo/resquiggle.py:1859-1950: a filler process enqueues reads, N
`_resquiggle_worker`s pull them one at a time).  Reads are independent, so here the unit of work
is a *batch* of reads (one pass of the HIP pipeline) and the queue is a single shared counter:
every rank atomically draws the next batch index until the job is exhausted.  Dynamic
assignment -- not a static split -- because batches differ in cost (read lengths) and GPUs in
speed.  A rank only ever materialises the batches it drew (`load_batch(b)` reads / generates
them); results stay on the rank that produced them unless a `sink` ships them elsewhere.

The counter lives in a `torch.distributed.TCPStore` of its own (public API; rank 0 lets the
kernel pick the port when it binds the store -- no bind / close / rebind race -- and announces
it through the default process group).  Without an initialised process group (world size 1)
the queue is a plain local counter.  The default process group only carries control messages
(the port, queue ids): any backend works, `gloo` is what a resquiggle job needs.
"""
import os

__all__ = ['BatchQueue', 'split_batches', 'run_sharded', 'resquiggle_sharded']

_STORE = None      # this process' connection to the queue store
_N_QUEUES = 0      # queues created so far (same order on every rank -> same key)


def _dist():
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def _queue_store():
    """TCPStore shared by all ranks of the default process group (created on first use)"""
    global _STORE
    if _STORE is not None:
        return _STORE
    import torch.distributed as dist
    from datetime import timedelta
    rank, world = dist.get_rank(), dist.get_world_size()
    host = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = [0]
    store = None
    if rank == 0:   # port 0: the listening socket is bound once, by the store itself
        store = dist.TCPStore(host, 0, world, is_master=True, timeout=timedelta(seconds=300),
                              wait_for_workers=False)
        port[0] = store.port
    dist.broadcast_object_list(port, src=0)
    if rank != 0:
        store = dist.TCPStore(host, int(port[0]), world, is_master=False,
                              timeout=timedelta(seconds=300))
    _STORE = store
    dist.barrier()
    return _STORE


def split_batches(n_reads, batch_size):
    """[(start, stop), ...] fixed-size batches over range(n_reads)"""
    return [(a, min(a + batch_size, n_reads)) for a in range(0, n_reads, batch_size)]


class BatchQueue(object):
    """Iterator over the batch indices this rank draws from the shared counter.

    Constructing a queue is a collective of the default process group: rank 0 announces the
    queue's id (its own count of queues so far, the job key and the batch count) and every rank
    checks it against its own -- a rank that skipped a job, or disagrees about its size, fails
    loudly here instead of silently drawing from another job's counter.  A counter is used once,
    so calling the same job function twice never sees a stale, already exhausted counter."""

    def __init__(self, n_batches, key=None):
        global _N_QUEUES
        self.n_batches = int(n_batches)
        self._dist = _dist()
        self._local = 0
        self.drawn = []
        if self._dist is not None and self._dist.get_world_size() > 1:
            self._store = _queue_store()
            mine = [_N_QUEUES, '' if key is None else str(key), self.n_batches]
            ann = list(mine)
            self._dist.broadcast_object_list(ann, src=0)
            if ann != mine:
                raise RuntimeError('work queues out of step: rank 0 opens queue %r, rank %d expected %r '
                                   '(every rank must run the same jobs in the same order)' %
                                   (ann, self._dist.get_rank(), mine))
            self._key = 'tombo_amd/queue/%d/%s' % (ann[0], ann[1])
            _N_QUEUES += 1
        else:
            self._store = None

    def __iter__(self):
        return self

    def __next__(self):
        if self._store is None:
            b = self._local
            self._local += 1
        else:
            b = self._store.add(self._key, 1) - 1   # atomic fetch-and-add on the store
        if b >= self.n_batches:
            raise StopIteration
        self.drawn.append(b)
        return b


def run_sharded(n_batches, load_batch, process_batch, sink=None, queue_key=None):
    """Process batches 0..n_batches-1 across the ranks of the default process group.

    load_batch(b)           -> the batch (only called on the rank that drew b)
    process_batch(batch)    -> its result
    sink(b, result)         -> consume / ship the result (default: keep it)
    Returns {b: result} of the batches this rank processed (empty values when a sink took them).
    """
    out = {}
    for b in BatchQueue(n_batches, key=queue_key):
        res = process_batch(load_batch(b))
        if sink is not None:
            sink(b, res)
            res = None
        out[b] = res
    return out


def resquiggle_sharded(map_results, process_batch, batch_size=4096, gather=True, queue_key=None):
    """List-based convenience form (small jobs, tests): `map_results` is the full read list,
    known to every rank; `process_batch(list_of_reads) -> list_of_results`.

    gather=True:  rank 0 returns the ordered result list (others None); the per-batch results
                  travel as pickled objects, which is fine for boundaries of a few thousand reads
                  and wrong for a million -- large jobs use `run_sharded` with a sink.
    gather=False: every rank returns {batch_index: results} of its own batches.
    """
    batches = split_batches(len(map_results), batch_size)
    mine = run_sharded(len(batches), lambda b: map_results[batches[b][0]:batches[b][1]],
                       process_batch, queue_key=queue_key)
    if not gather:
        return mine
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return [r for b in range(len(batches)) for r in mine[b]]
    rank, world = dist.get_rank(), dist.get_world_size()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in gathered:
        merged.update(part)
    missing = [b for b in range(len(batches)) if b not in merged]
    if missing:
        raise RuntimeError('work queue lost batches %r' % (missing,))
    return [r for b in range(len(batches)) for r in merged[b]]

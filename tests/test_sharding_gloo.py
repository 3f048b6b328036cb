"""The N>1 host path on CPU: two gloo processes pull batches from the shared work queue; every
batch is processed exactly once and the gathered result equals the single-process result.  The
per-batch compute is a stand-in (the CPU oracle on tiny reads): the queue / gather logic is what
is under test -- the engine itself needs a GPU."""
import os
import sys
import socket

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reads(n):
    from tombo_amd import synth, tombo_stats as ts, tombo_helper as th
    model = ts.TomboModel(seq_samp_type=th.seqSampleType('DNA', False))
    return model, [synth.synth_read(model, 260 + 7 * i, 77 + i, **synth.DNA_SYNTH)[:2]
                   for i in range(n)]


def _process(model):
    import oracle
    from tombo_amd import tombo_stats as ts, tombo_helper as th
    params = ts.load_resquiggle_parameters(th.seqSampleType('DNA', False))
    p = oracle.make_params(params)
    o = oracle.make_opts(6, 2, outlier_thresh=5.0, sig_match_thresh=1.1)

    def f(batch):
        out = []
        for seq, raw in batch:
            r = oracle.resquiggle_read(raw, ts.encode_seq(seq), model.level_means,
                                       model.level_sds, p, o)
            out.append((r['status'], r['segs'].tolist() if r['status'] == 0 else None))
        return out
    return f


def _worker(rank, world, port, n_reads, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from tombo_amd import sharding
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model, reads = _reads(n_reads)
    # same call twice with the default key: every call gets a fresh counter
    mine = sharding.resquiggle_sharded(reads, _process(model), batch_size=3, gather=False)
    dist.barrier()
    full = sharding.resquiggle_sharded(reads, _process(model), batch_size=3, gather=True)
    dist.barrier()
    # large-job form: batches are loaded by the rank that draws them, results go to a sink
    loaded, sunk = [], []
    f = _process(model)
    sharding.run_sharded(5, lambda b: (loaded.append(b), reads[3 * b:3 * b + 3])[1], f,
                         sink=lambda b, res: sunk.append((b, [st for st, _ in res])))
    assert loaded == [b for b, _ in sunk]
    q.put((rank, sorted(mine.keys()), full, sorted(loaded)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_work_queue_gloo():
    import torch.multiprocessing as mp
    n_reads, world = 14, 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got.sort()
    keys = got[0][1] + got[1][1]
    assert sorted(keys) == list(range(5)), 'every batch exactly once: %r' % (keys,)
    assert sorted(got[0][3] + got[1][3]) == list(range(5)), 'sink form: every batch exactly once'
    model, reads = _reads(n_reads)
    from tombo_amd import sharding
    single = sharding.resquiggle_sharded(reads, _process(model), batch_size=3)
    assert got[0][2] == single          # rank 0 holds the ordered, gathered result
    assert got[1][2] is None
    assert sum(1 for st, _ in single if st == 0) >= 12


def test_split_and_local_queue():
    from tombo_amd import sharding
    assert sharding.split_batches(10, 4) == [(0, 4), (4, 8), (8, 10)]
    assert sharding.split_batches(0, 4) == []
    assert list(sharding.BatchQueue(3)) == [0, 1, 2]

"""`python bench.py --gpus 2` end to end on CPU with a stub engine: the self-launch, the gloo
control plane, the shared work queue, the feeder / streaming slots and the JSON line of an N > 1
run (cpu_baseline, roofline and per_rank must be there at every N)."""
import os
import sys
import json
import subprocess

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _run(extra):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'TBA_STORE_PORT'):
        env.pop(k, None)
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '1', '--reads', '12',
         '--bases', '400', '--cpu-sample', '2', '--cpu-per-core', '0', '--stream-batch', '5',
         '--engine-stub', os.path.join(ROOT, 'tests', 'stub_engine.py')] + extra,
        cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [x for x in out.stdout.decode().splitlines() if x.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_two_rank_bench_line_is_complete():
    r = _run(['--gpus', '2'])
    assert r['n_gpus'] == 2 and r['steps'] == 4 and r['scaling'] == 'weak' and r['unit'] == 'reads/s'
    assert r['cpu_baseline']['kind'] == 'port' and r['cpu_baseline']['value'] > 0
    assert r['roofline']['bound'] == 'hbm' and r['roofline']['frac'] > 0
    assert 'traffic' in r['roofline'] and 'kernel' in r['roofline']
    pr = r['per_rank']
    assert [x['rank'] for x in pr] == [0, 1]
    assert all(k in pr[0] for k in ('device', 'steps', 'resident_reads_per_s', 'stream_reads_per_s'))
    # the queue hands out exactly K * N resident passes and the streamed batches, each once
    assert sum(x['steps'] for x in pr) == 8
    assert sum(x['stream_batches'] for x in pr) == r['end_to_end']['batches']
    assert r['end_to_end']['reads'] > 0 and r['end_to_end']['success_rate'] == 1.0
    assert 'gloo' in r['config']['parallelism'] and 'RCCL' not in json.dumps(r)
    assert r['value'] > 0 and r['per_rank_reads_per_s']['min'] <= r['per_rank_reads_per_s']['max']


def test_single_rank_bench_line_keys():
    r = _run(['--api-reads', '0'])
    assert r['n_gpus'] == 1 and len(r['per_rank']) == 1 and r['per_rank'][0]['steps'] == 4
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
              'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'end_to_end'):
        assert k in r, k


def test_four_ranks_under_torch_distributed_run():
    """the driver's own launch line for N > 1 (python -m torch.distributed.run ... bench.py --gpus N)"""
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'TBA_STORE_PORT'):
        env.pop(k, None)
    import socket
    with socket.socket() as s:      # a free port for the launcher's rendezvous
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4',
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
         '--gpus', '4', '--steps', '3', '--warmup', '1', '--reads', '12', '--bases', '400', '--cpu-sample', '2',
         '--cpu-per-core', '0', '--stream-batch', '5', '--engine-stub',
         os.path.join(ROOT, 'tests', 'stub_engine.py')],
        cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [x for x in out.stdout.decode().splitlines() if x.strip().startswith('{')]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r['n_gpus'] == 4 and r['steps'] == 3 and [x['rank'] for x in r['per_rank']] == [0, 1, 2, 3]
    assert sum(x['steps'] for x in r['per_rank']) == 12
    assert r['cpu_baseline']['value'] > 0 and r['roofline']['frac'] > 0


def test_cfg5_job_hands_out_every_distinct_batch_once():
    """--preset cfg5: the job's batches are keyed by their first read (read r of the job is a function
    of (job seed, r) alone); two ranks draw every batch exactly once between them and nobody
    synthesises anything on the host"""
    r = _run(['--gpus', '2', '--preset', 'cfg5', '--job-reads', '47', '--reads', '5'])
    assert r['n_gpus'] == 2 and r['scaling'] == 'strong' and r['unit'] == 'reads/s'
    job = r['distinct_read_job']
    assert job['job_reads'] == 47 and job['reads_done'] == 47 and job['batches'] == 10
    firsts = sorted(f for x in r['per_rank'] for f in x['first_reads_drawn'])
    assert firsts == list(range(0, 47, 5))               # distinct seeds (first reads), each once
    assert sum(x['job_reads'] for x in r['per_rank']) == 47
    assert sum(x['job_batches'] for x in r['per_rank']) == 10
    assert r['config']['setup_s']['synthesis'] == 0.0
    assert r['cpu_baseline']['value'] > 0 and r['roofline']['frac'] > 0
    # the default job size: 125 000 reads per rank (a million on eight), weak scaling
    r = _run(['--preset', 'cfg5', '--reads', '25000'])
    assert r['scaling'] == 'weak' and r['distinct_read_job']['job_reads'] == 125000 and r['steps'] == 5


def test_eight_ranks_share_the_host_and_the_job():
    """--gpus 8 --preset cfg5 (BASELINE.json's scaling configuration) with the stub engine: every batch of
    the job is drawn once, the per-rank sums add up, the ranks split the host's cores between them, ranks
    > 0 stay off the host until rank 0's CPU legs are over, and the set-up does not grow with N"""
    r1 = _run(['--preset', 'cfg5', '--job-reads', '83', '--reads', '5'])
    r8 = _run(['--gpus', '8', '--preset', 'cfg5', '--job-reads', '83', '--reads', '5'])
    job = r8['distinct_read_job']
    assert r8['n_gpus'] == 8 and job['job_reads'] == 83 and job['reads_done'] == 83 and job['batches'] == 17
    firsts = sorted(f for x in r8['per_rank'] for f in x['first_reads_drawn'])
    assert firsts == list(range(0, 83, 5))
    assert sum(x['job_reads'] for x in r8['per_rank']) == 83 and sum(x['job_batches'] for x in r8['per_rank']) == 17
    assert job['checksum_mod_2_40_summed'] == r1['distinct_read_job']['checksum_mod_2_40_summed']      # the sharding does not show in the results
    ncpu = os.cpu_count() or 8
    assert sum(x['host_threads'] for x in r8['per_rank']) <= max(ncpu, 8)
    assert all(x['waited_for_cpu_legs_s'] >= 0 for x in r8['per_rank'])
    assert r8['cpu_baseline']['value'] > 0
    # one CPU-leg child on rank 0 whatever N; the ranks come up in parallel behind it
    assert r8['config']['setup_s']['total_wall'] <= r1['config']['setup_s']['total_wall'] + 60.0


def test_default_two_rank_run_carries_the_distinct_read_job():
    r = _run(['--gpus', '2', '--bases', '10000', '--reads', '3', '--cpu-sample', '1'])
    job = r['distinct_read_job']
    assert job['job_reads'] == 24 and job['reads_done'] == 24 and job['batches'] == 8
    assert sum(x['job_reads'] for x in r['per_rank']) == 24


def test_make_reads_through_shared_memory_equals_inline():
    """bench.make_reads hands the workers' samples over through /dev/shm files: same reads, same order
    as the inline generator"""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    bases = np.array([300 + 7 * i for i in range(70)], np.int64)
    seqs, raws, dacs = bench.make_reads(bases, 4242, 2, 'DNA', True)
    assert len(seqs) == len(raws) == len(dacs) == 70
    for i in (0, 1, 33, 69):
        s, r, d = bench._gen((int(bases[i]), 4242 + i, 'DNA', True))
        assert seqs[i] == s and np.array_equal(raws[i], r) and np.array_equal(dacs[i], d)
        assert raws[i].dtype == np.float64 and dacs[i].dtype == np.int16
    seqs2, raws2, dacs2 = bench.make_reads(bases[:5], 4242, 2, 'DNA', False)   # (few reads: inline)
    assert dacs2[0] is None and np.array_equal(raws2[4], raws[4])

"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share
a pass on gfx950: TCC has 4 counter slots, FETCH_SIZE takes 3, WRITE_SIZE 2).

    python tools/pmc_traffic.py <fetch.db> <write.db> <reads> > profiles/rNN_pmc_hbm_traffic.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both
counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane)
coalesced streaming read, other access widths and WRITE_SIZE are uncalibrated.  Both the raw sum
and the sum with FETCH doubled (upper bound) are written; the bench line uses the raw one for
k_dp, whose traffic is 90 % writes (the packed move rows) and whose reads are 8 B/lane.
"""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    out = {}
    for name, v, k in c.execute(
            'select kernel_name, sum(value), count(*) from counters_collection '
            'where counter_name = ? group by kernel_name', (counter,)):
        out[name.split('(')[0].replace('void ', '')] = (float(v), int(k))
    return out


def main(fetch_db, write_db, reads):
    reads = int(reads)
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    kernels = {}
    tot_raw = tot_up = 0.0
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
        raw, up = (fk + wk) * 1024.0, (2 * fk + wk) * 1024.0
        kernels[k] = dict(FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk, dispatches=max(f.get(k, (0, 0))[1],
                                                                             w.get(k, (0, 0))[1]),
                          bytes_per_read=raw / reads, bytes_per_read_fetch_doubled=up / reads)
        if not k.startswith('__amd'):
            tot_raw += raw
            tot_up += up
    dp = [k for k in kernels if k.startswith('k_dp<') and kernels[k]['WRITE_SIZE_KB'] > 1e6]
    main_dp = max(dp, key=lambda k: kernels[k]['WRITE_SIZE_KB']) if dp else None
    json.dump(dict(reads=reads, main_dp_kernel=main_dp,
                   k_dp_bytes_per_read=kernels[main_dp]['bytes_per_read'] if main_dp else None,
                   k_dp_bytes_per_read_fetch_doubled=(
                       kernels[main_dp]['bytes_per_read_fetch_doubled'] if main_dp else None),
                   pipeline_bytes_per_read=tot_raw / reads,
                   pipeline_bytes_per_read_fetch_doubled=tot_up / reads, kernels=kernels),
              sys.stdout, indent=1, sort_keys=True)


if __name__ == '__main__':
    main(*sys.argv[1:4])

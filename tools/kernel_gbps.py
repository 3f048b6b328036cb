"""Per-kernel HBM rate table: bytes per launch from the bench line's in-run PMC passes
(roofline.traffic_bytes_per_read_by_kernel: FETCH_SIZE + WRITE_SIZE per read, two rocprofv3 --pmc
passes) divided by the kernel's average duration from a rocprofv3 --kernel-trace --stats summary of
the same configuration (tools/rocpd_summary.py output).
python tools/kernel_gbps.py <bench line .json> <kernel stats .txt> [reads per launch]"""
import json
import sys

d = json.load(open(sys.argv[1]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else d['config']['reads_per_gpu']
per_read = d['roofline']['traffic_bytes_per_read_by_kernel']
stats = {}
for line in open(sys.argv[2]).read().splitlines()[1:]:
    parts = line.rsplit(None, 4)
    if len(parts) == 5:
        stats[parts[0].strip()] = (int(parts[1]), float(parts[3]))
print('%-34s %6s %10s %12s %9s  (%d reads per launch; traffic: %s)' % (
    'kernel', 'calls', 'avg ms', 'MB / launch', 'GB/s', n, d['roofline']['traffic_scope'][:60] + '...'))
tot_b = tot_t = 0.0
for k, b in sorted(per_read.items(), key=lambda kv: -kv[1]):
    if k not in stats:
        continue
    calls, avg_us = stats[k]
    per_step = calls / max(min(c for c, _ in stats.values() if c >= 1), 1)
    steps = [c for kk, (c, _) in stats.items() if kk.startswith('k_normalize')]
    launches_per_step = calls / float(steps[0]) if steps else 1.0
    mb = b * n / 1e6
    ms = avg_us * launches_per_step / 1e3
    tot_b += mb
    tot_t += ms
    print('%-34s %6.0f %10.3f %12.1f %9.0f' % (k, launches_per_step, ms, mb, mb / ms if ms > 0 else 0.0))
print('%-34s %6s %10.3f %12.1f %9.0f' % ('all of the above', '', tot_t, tot_b, tot_b / tot_t))

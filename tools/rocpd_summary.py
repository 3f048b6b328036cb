"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel calls / total / average duration.

    python tools/rocpd_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats.txt

(rocprofv3 --kernel-trace --stats writes this database on ROCm 7.2; durations are in us.)
"""
import sys
import sqlite3


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute('select name, total_calls, total_duration, average, percentage '
                          'from top_kernels'))
    print('%-64s %6s %14s %14s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
    for name, calls, tot, avg, pct in rows:
        short = name.split('(')[0].replace('void ', '')
        print('%-64s %6d %14.1f %14.1f %7.2f' % (short[:64], calls, tot, avg, pct))
    try:
        rows = list(c.execute(
            'select kernel_name, counter_name, sum(value), count(*) from counters_collection '
            'group by kernel_name, counter_name order by sum(value) desc'))
        if rows:
            print('\n%-64s %-14s %18s %8s' % ('kernel', 'counter', 'sum', 'samples'))
            for name, cn, v, k in rows:
                short = name.split('(')[0].replace('void ', '')
                print('%-64s %-14s %18.1f %8d' % (short[:64], cn, v, k))
    except sqlite3.Error:
        pass


if __name__ == '__main__':
    main(sys.argv[1])

"""Per-kernel SQ counters from a rocprofv3 --pmc run (rocpd database) -> text table + the
instruction figures bench.py's roofline_valu entry uses.

    python tools/pmc_sq.py <results.db> <reads> <rows_per_read> [<key>]   # e.g. DNA_b10000_w500

Prints, per kernel, the sum of every collected counter, and for the main banded DP
(k_dp<CPL, false> with the most SQ_INSTS_VALU) the VALU wave-instructions per read and per DP
row; with <key> it also prints the JSON fragment for profiles/valu_counts.json."""
import sys
import json
import sqlite3


def main(db, reads, rows, key=None):
    reads, rows = int(reads), float(rows)
    c = sqlite3.connect(db)
    tab = {}
    for name, ctr, v, k in c.execute(
            'select kernel_name, counter_name, sum(value), count(*) from counters_collection '
            'group by kernel_name, counter_name'):
        short = name.split('(')[0].replace('void ', '')
        tab.setdefault(short, {})[ctr] = (float(v), int(k))
    ctrs = sorted({x for d in tab.values() for x in d})
    print('%-40s %s' % ('kernel', ' '.join('%18s' % x for x in ctrs)))
    for kname in sorted(tab, key=lambda n: -tab[n].get('SQ_INSTS_VALU', (0, 0))[0]):
        if kname.startswith('__amd'):
            continue
        print('%-40s %s' % (kname[:40], ' '.join('%18.4g' % tab[kname].get(x, (0, 0))[0] for x in ctrs)))
    dp = [k for k in tab if k.startswith(('k_dp<', 'k_dp_multi<')) and 'SQ_INSTS_VALU' in tab[k]]
    if dp:
        main_dp = max(dp, key=lambda k: tab[k]['SQ_INSTS_VALU'][0])
        v, n = tab[main_dp]['SQ_INSTS_VALU']
        per_read = v / n / reads
        print('\n%s: SQ_INSTS_VALU %.4g over %d dispatch(es) = %.0f wave-instructions per read = '
              '%.1f per DP row (%g rows per read)' % (main_dp, v, n, per_read, per_read / rows, rows))
        if key:
            print(json.dumps({key: dict(k_dp_valu_wave_insts_per_read=round(per_read, 1),
                                        valu_insts_per_row=round(per_read / rows, 1),
                                        source='rocprofv3 --pmc SQ_INSTS_VALU, %s, %d reads' % (main_dp, reads))}))


if __name__ == '__main__':
    main(*sys.argv[1:])

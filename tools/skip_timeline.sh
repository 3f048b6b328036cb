# When the RNA skip-window kernels of one cfg4 step ran (rocprofv3 --kernel-trace): start / duration / queue of k_skip_*
#   gpurun -- 'TBA_SKIP_FORK=2 bash tools/skip_timeline.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/trs
rocprofv3 --kernel-trace -d /tmp/trs -- python $R/bench.py --preset cfg4 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --e2e none --api-reads 0 > /tmp/trs.json 2>/dev/null
db=$(find /tmp/trs -name "*.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$db")
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt=[t for t in tabs if 'kernel_dispatch' in t][0]
ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=list(c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id=s.id order by d.start"%(kt,ks)))
last=[i for i,r in enumerate(rows) if 'k_skip_plan' in r[0]][-1]
t0=rows[last][1]
for n,s,e,q in rows[last:last+12]:
    print('%-60s queue %s start %7.3f ms dur %6.3f ms'%(n.split('(')[0].replace('void ','')[:60],q,(s-t0)/1e6,(e-s)/1e6))
PY

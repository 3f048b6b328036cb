# Kernel timeline (rocprofv3 --kernel-trace) of the long-tailed streaming run: every kernel longer
# than 8 ms with its start, duration and hardware queue, and the summed time per kernel.
#   gpurun -- 'bash tools/kernel_timeline.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr
rocprofv3 --kernel-trace -d /tmp/tr -- python $R/bench.py --preset longtail --steps 2 --no-cpu-baseline --no-pmc --api-reads 0 --slots 3 > /tmp/tr.json 2>/dev/null
db=$(find /tmp/tr -name "*.db" | head -1)
python - <<PY
import sqlite3, json
c=sqlite3.connect("$db")
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt=[t for t in tabs if 'kernel_dispatch' in t][0]
cols=[r[1] for r in c.execute("pragma table_info(%s)"%kt)]
print(kt, cols)
ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=list(c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id=s.id order by d.start"%(kt,ks)))
t0=rows[0][1]
# last third of the run = streaming phase; print long kernels
import collections
long=[(n.split('(')[0][:40],(s-t0)/1e6,(e-s)/1e6,q) for n,s,e,q in rows if (e-s)/1e6>8]
for x in long[-60:]: print('%-40s start %9.1f ms  dur %8.1f ms  queue %s'%x)
tot=collections.Counter()
for n,s,e,q in rows: tot[n.split('(')[0][:40]]+= (e-s)/1e6
print(sorted(tot.items(), key=lambda x:-x[1])[:12])
print('wall', (rows[-1][2]-t0)/1e6)
PY

# Kernel + copy timeline (rocprofv3 --kernel-trace --memory-copy-trace) of the LAST no-signal resquiggle_batch call of
# tools/api_profile.py (5 000 reads): per sub-batch stream, when its copies and its longest kernels ran.
#   gpurun -- 'bash tools/api_timeline.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr
NO_SIGNAL=1 API_ONE_CALL=1 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tr -- python $R/tools/api_profile.py 5000 > /tmp/tr.log 2>&1
tail -3 /tmp/tr.log
db=$(find /tmp/tr -name "*.db" | head -1)
python - <<PY
import sqlite3, collections
c=sqlite3.connect("$db")
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt=[t for t in tabs if 'kernel_dispatch' in t][0]
ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=list(c.execute("select s.kernel_name, d.start, d.end, d.queue_id from %s d join %s s on d.kernel_id=s.id order by d.start"%(kt,ks)))
mt=[t for t in tabs if 'memory_copy' in t]
copies=[]
if mt:
    cols=[r[1] for r in c.execute("pragma table_info(%s)"%mt[0])]
    copies=list(c.execute("select start, end, size from %s order by start"%mt[0]))
# the last call: everything after the last gap of > 30 ms in kernel activity
ends=[r[2] for r in rows]
cut=0
for i in range(1,len(rows)):
    if rows[i][1]-max(ends[:i][-50:])>30e6: cut=i
rows=rows[cut:]
t0=rows[0][1]
copies=[x for x in copies if x[0]>=t0-30e6]
if copies: t0=min(t0, copies[0][0])
print('last call: %d kernels, %d copies, GPU span %.1f ms'%(len(rows),len(copies),(max(r[2] for r in rows)-t0)/1e6))
for s,e,sz in copies:
    if sz>1e6: print('copy   start %7.1f ms dur %6.1f ms  %8.1f MB'%((s-t0)/1e6,(e-s)/1e6,sz/1e6))
for n,s,e,q in rows:
    if (e-s)/1e6>1.0: print('%-34s queue %s start %7.1f ms dur %6.1f ms'%(n.split('(')[0].replace('void ','')[:34],q,(s-t0)/1e6,(e-s)/1e6))
PY

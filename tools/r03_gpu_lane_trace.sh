#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export STARKHIP_POOL_BIG_LANE=1
rocprofv3 --kernel-trace -d $OUT/kt_lane -o kt -- python3 $R/bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-boundary --inflight 8 > $OUT/kt_lane.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3,glob
db=glob.glob('gpurun_out/kt_lane/**/*results.db',recursive=True)[0]
con=sqlite3.connect(db); cur=con.cursor()
rows=list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
t0=rows[0][1]
big=[(n,s-t0,e-t0) for n,s,e,g,w in rows if ('leaf_hash_lane' in n) and (e-s)>50e6]
print('lane hash launches', len(big))
for n,s,e in big[8:40]:
    # concurrency: how many other big hashes overlap the midpoint
    mid=(s+e)/2
    k=sum(1 for _,s2,e2 in big if s2<=mid<=e2)
    print('start %8.1f ms dur %7.1f ms concurrent %d'%(s/1e6,(e-s)/1e6,k))
others={}
for n,s,e,g,w in rows:
    if 'leaf_hash_lane' in n: continue
    key=n.split('(')[0][-40:]
    d=others.setdefault(key,[0,0]); d[0]+=1; d[1]+=(e-s)
for k,v in sorted(others.items(), key=lambda kv:-kv[1][1])[:6]: print(k, v[0], round(v[1]/v[0]/1e6,2),'ms avg')
PY
rm -rf $OUT/kt_lane
tail -1 $OUT/kt_lane.log | cut -c1-200

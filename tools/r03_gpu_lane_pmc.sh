#!/bin/bash
# SQ counters of the leaf hash in its quad and lane forms, eight proofs in flight (rocprofv3 gets python3 directly)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lane in 0 1; do
  export STARKHIP_POOL_BIG_LANE=$lane
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $OUT/pmc_lane$lane -o sq -- python3 $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-boundary --inflight 8 > $OUT/pmc_lane$lane.log 2>&1
  cd $R
  python3 tools/rocprof_export.py pmc $(find $OUT/pmc_lane$lane -name "*results.db" | head -1) $OUT/lane_pmc_$lane.csv
  rm -rf $OUT/pmc_lane$lane
  cd /tmp
done
cd $R
python3 - <<'PY'
import csv,collections
for lane in (0,1):
    rows=list(csv.DictReader(open('gpurun_out/lane_pmc_%d.csv'%lane)))
    print('lane',lane, rows[0].keys() if rows else None)
    for r in rows:
        if 'leaf_hash' in r.get('kernel','') and ('lane' in r['kernel'] or r['kernel'].startswith('leaf_hash_kernel') or 'leaf_hash_kernel' in r['kernel']):
            print({k:(v[:40] if isinstance(v,str) else v) for k,v in r.items()})
PY

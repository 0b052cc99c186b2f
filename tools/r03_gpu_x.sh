cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_demo8 -o d8 -- $R/build/signature_demo --batch 8 --operands $R/tests/golden/signature_operands_8.bin --steps 3 --warmup 1 > $O/r03_x_demo8_under_rocprof.json 2> $O/prof_demo8.log
cd $R
DB=$(find $O/prof_demo8 -name "*results.db" | head -1)
python3 tools/rocprof_export.py stats $DB $O/r03_x_demo8_kernel_stats.csv
python3 tools/rocprof_export.py bygrid $DB $O/r03_x_demo8_kernel_stats_by_grid.csv
rm -rf $O/prof_demo8
head -25 $O/r03_x_demo8_kernel_stats.csv
cut -c140-330 $O/r03_x_demo8_under_rocprof.json
python3 tools/air_latency.py > $O/r03_x_air_latency.json 2>/dev/null; python3 -c "
import json
d=json.load(open('$O/r03_x_air_latency.json'))
for k,v in d.items(): print(k, round(v['wall_ms'],1), {a:round(b,1) for a,b in v['phase_ms'].items() if b>1})"

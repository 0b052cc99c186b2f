#!/bin/bash
# kernel trace of one bench.py configuration: the timeline of its long kernels in the last part of the run.
# usage: bash tools/gpu_trace.sh TAG [window_s] [min_ms]    (configuration through exported environment variables and BENCH_ARGS)
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/trace_$TAG -o kt -- python3 $R/bench.py --steps ${STEPS:-24} --warmup 1 --no-cpu-baseline --no-boundary --no-solo ${BENCH_ARGS} > $OUT/trace_$TAG.log 2>&1 || { tail -5 $OUT/trace_$TAG.log; exit 1; }
cd $R
DB=$(find $OUT/trace_$TAG -name "*results.db" | head -1)
python3 tools/kernel_timeline.py $DB ${2:-1.5} ${3:-4.0} > $OUT/${TAG}_timeline.txt
python3 tools/rocprof_export.py bygrid $DB $OUT/${TAG}_bygrid.csv
python3 tools/kernel_overlap.py $DB > $OUT/${TAG}_overlap.txt 2>&1
rm -rf $OUT/trace_$TAG
tail -c 600 $OUT/trace_$TAG.log | cut -c1-300
head -8 $OUT/${TAG}_bygrid.csv
cat $OUT/${TAG}_overlap.txt

# per-kernel average durations with one proof in flight (rocprofv3 --kernel-trace --stats): bash tools/experiments/kernel_stats_one.sh TAG [name filter]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/ks_$1 -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-boundary --inflight 1 > $OUT/ks_$1.log 2>&1 || { tail -5 $OUT/ks_$1.log; exit 1; }
cd $R
python3 tools/rocprof_export.py stats $(find $OUT/ks_$1 -name "*results.db" | head -1) $OUT/$1_kernel_stats_one.csv
rm -rf $OUT/ks_$1
grep -E "${2:-.}" $OUT/$1_kernel_stats_one.csv | head -12

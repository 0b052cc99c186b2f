#!/bin/bash
# How the files under profiles/ are produced (run on the GPU box from the repo root; outputs land in gpurun_out/ as
# <tag>_*; copy the ones to be judged into profiles/):
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate PMC passes (never together with a trace),
#   the SQ instruction counters, the kernel-overlap figure of the default four-in-flight run, the integer-VALU issue
#   rates behind the "4 cycles per instruction" peak, and the default bench line.
# rocprofv3 gets `python3 ...` directly after `--` (no wrapper that would exec after the GPU is initialised).
# usage: profile_round.sh TAG [prof|sig|all]   (prof: rocprofv3 passes + bench line; sig: the signature-path measurements)
set -e
TAG=${1:-r03}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
if [ "$PART" != "sig" ]; then
cd /tmp && export TMPDIR=/tmp
ONE="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-boundary --inflight 1"
rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary --inflight 1 > $OUT/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- $ONE > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o write -- $ONE > $OUT/prof_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES -d $OUT/prof_sq -o sq -- $ONE > $OUT/prof_sq.log 2>&1
# the default run (eight in flight, lane-form commitment groups): kernel trace, then the counters of leaf_hash_lane_kernel in passes of their own
rocprofv3 --kernel-trace --stats -d $OUT/prof_ov -o ov -- python3 $R/bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-boundary --no-solo > $OUT/prof_ov.log 2>&1
LANE="python3 $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-boundary --no-solo"
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_lfetch -o fetch -- $LANE > $OUT/prof_lfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_lwrite -o write -- $LANE > $OUT/prof_lwrite.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES -d $OUT/prof_lsq -o sq -- $LANE > $OUT/prof_lsq.log 2>&1
cd $R
db() { find $OUT/$1 -name "*results.db" | head -1; }
python3 tools/rocprof_export.py stats $(db prof_kt) $OUT/${TAG}_kernel_stats.csv
python3 tools/rocprof_export.py bygrid $(db prof_kt) $OUT/${TAG}_kernel_stats_by_grid.csv
python3 tools/rocprof_export.py pmc $(db prof_fetch) $OUT/${TAG}_pmc_fetch_size.csv
python3 tools/rocprof_export.py pmc $(db prof_write) $OUT/${TAG}_pmc_write_size.csv
python3 tools/rocprof_export.py pmc $(db prof_sq) $OUT/${TAG}_pmc_sq_counters.csv
python3 tools/rocprof_export.py pmc $(db prof_lfetch) $OUT/${TAG}_pmc_fetch_size_in_flight.csv
python3 tools/rocprof_export.py pmc $(db prof_lwrite) $OUT/${TAG}_pmc_write_size_in_flight.csv
python3 tools/rocprof_export.py pmc $(db prof_lsq) $OUT/${TAG}_pmc_sq_counters_in_flight.csv
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_traffic.json $(db prof_fetch) $(db prof_write) $(db prof_lfetch) $(db prof_lwrite)
python3 tools/kernel_overlap.py $(db prof_ov) > $OUT/${TAG}_kernel_overlap.txt
python3 tools/rocprof_export.py bygrid $(db prof_ov) $OUT/${TAG}_kernel_stats_by_grid_in_flight.csv
python3 tools/rocprof_export.py stats $(db prof_ov) $OUT/${TAG}_kernel_stats_in_flight.csv
grep '^{"metric"' $OUT/prof_ov.log | tail -1 > $OUT/${TAG}_bench_in_flight_under_rocprof.json  # the bench line, not the profiler's last log line
python3 tools/lane_group_stats.py $(db prof_ov) $OUT/${TAG}_lane_groups.json > /dev/null
grep '^{"metric"' $OUT/prof_kt.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq $OUT/prof_ov $OUT/prof_lfetch $OUT/prof_lwrite $OUT/prof_lsq
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate_bench tools/valu_rate_bench.hip && /tmp/valu_rate_bench > $OUT/${TAG}_valu_rates.txt 2>&1 || true
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
head -8 $OUT/${TAG}_kernel_stats.csv; cat $OUT/${TAG}_kernel_overlap.txt; cat $OUT/${TAG}_pmc_traffic.json | head -40; tail -c 3000 $OUT/${TAG}_bench.json
fi
if [ "$PART" != "prof" ]; then
cd $R
# the signature path from compiled host code (tools/signature_demo.cpp on the library's proof pool) and from the Python harness
OPS=tests/golden/signature_operands_8.bin
build/signature_demo --batch 8 --operands $OPS --steps 5 --warmup 1 --timeline > $OUT/${TAG}_demo_batch8.json 2> $OUT/${TAG}_demo_batch8_timeline.txt || true
build/signature_demo --batch 1 --steps 10 --warmup 2 --timeline > $OUT/${TAG}_demo_batch1.json 2> $OUT/${TAG}_demo_batch1_timeline.txt || true
python3 tools/bench_signature.py --batch 8 --steps 3 > $OUT/${TAG}_bench_signature_batch8.json 2> /dev/null || true
python3 tools/bench_signature.py --batch 1 --steps 5 > $OUT/${TAG}_bench_signature.json 2> /dev/null || true
python3 tools/air_latency.py > $OUT/${TAG}_air_latency.json 2> /dev/null || true
cut -c1-400 $OUT/${TAG}_demo_batch8.json; cut -c1-400 $OUT/${TAG}_demo_batch1.json; cut -c1-300 $OUT/${TAG}_bench_signature_batch8.json; cut -c1-300 $OUT/${TAG}_bench_signature.json
fi

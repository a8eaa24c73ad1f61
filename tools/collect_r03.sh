#!/bin/bash
# Round-3 evidence on the GPU box, everything under gpurun_out/r03/ (copy the summaries into profiles/ afterwards):
#   rocprofv3 --kernel-trace --stats of the bench command itself (the process that prints the JSON line) for BASELINE
#   config 3 (300 x 1080p, default path and --path tiles) and config 4 (300 x 4K, default path = tiles);
#   the plain bench lines; the PMC passes of both kernels on both workloads (tools/pmc_r03.sh -> tools/pmc_record_r03.py).
#     bash tools/collect_r03.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() { # tag, frames kept, bench args...
  local tag=$1 keep=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o r -- python3 $ROOT/bench.py --no-cpu-baseline --deliver none "$@" > $OUT/r03_${tag}_bench_under_rocprof.json 2> $OUT/prof_$tag.err
  python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/prof_$tag/*results.db $OUT/prof_$tag/*/*results.db 2>/dev/null | head -1) --last $keep > $OUT/r03_${tag}_kernel_stats_timed.csv
  rm -rf $OUT/prof_$tag
}
echo "== pmc (first: the bench lines below read profiles/r03_pmc.json and check its source hash)"
cd $ROOT
for p in runs tiles; do
  bash tools/pmc_r03.sh base $p 1920 1080 300 > $OUT/pmc_1080p_$p.txt 2>&1
  bash tools/pmc_r03.sh base $p 3840 2160 300 > $OUT/pmc_4k_$p.txt 2>&1
done
python3 tools/pmc_record_r03.py gpurun_out/pmc3_base_runs_1920x1080 gpurun_out/pmc3_base_tiles_1920x1080 gpurun_out/pmc3_base_runs_3840x2160 gpurun_out/pmc3_base_tiles_3840x2160 > $OUT/pmc_record.log 2>&1
cp profiles/r03_pmc.json $OUT/r03_pmc.json
{ echo "# PMC passes of both encode kernels on the shipped tree (tools/pmc_r03.sh: one rocprofv3 --pmc run per counter group, --kernel-trace only"
  echo "# beside it), per-dispatch means; 300 frames per launch.  L1->L2 read requests = TCP_TCC_READ_REQ_sum; pixel lines = W*H*3*300/128."
  for d in runs_1920x1080 tiles_1920x1080 runs_3840x2160 tiles_3840x2160; do echo; echo "===== $d"; cat gpurun_out/pmc3_base_$d/summary.txt; done; } > $OUT/r03_memory_path_pmc.txt
cd /tmp
echo "== stats"
stats 1080p 200
stats 1080p_runs 200 --path runs
stats 4k 60 --width 3840 --height 2160 --steps 60 --warmup 20
stats 4k_runs 60 --width 3840 --height 2160 --steps 60 --warmup 20 --path runs
echo "== plain bench"
cd $ROOT
python3 bench.py > $OUT/r03_1080p_bench.json 2> /dev/null
python3 bench.py --path runs --no-cpu-baseline > $OUT/r03_1080p_runs_bench.json 2> /dev/null
python3 bench.py --no-cpu-baseline --width 3840 --height 2160 --steps 60 --warmup 20 > $OUT/r03_4k_bench.json 2> /dev/null
python3 bench.py --no-cpu-baseline --width 3840 --height 2160 --steps 60 --warmup 20 --path runs > $OUT/r03_4k_runs_bench.json 2> /dev/null
head -6 $OUT/r03_1080p_kernel_stats_timed.csv $OUT/r03_4k_kernel_stats_timed.csv
cat $OUT/r03_1080p_bench.json | cut -c1-600

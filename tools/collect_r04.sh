#!/bin/bash
# Round-4 evidence on the GPU box, everything under gpurun_out/r04/ (copy the summaries into profiles/ afterwards):
#   PMC passes of both encode kernels (and the assemble kernel behind them) on both workloads -> profiles/r04_pmc.json;
#   rocprofv3 --kernel-trace --stats of the bench command itself for BASELINE config 3 (300 x 1080p) and config 4 (300 x 4K);
#   the plain bench lines (the 1080p line carries `sustained` with sampled power and clock, and `config4`);
#   the bench exactly as the driver runs it, three times; sustained A/B against the round-3 library when build/libencoder_r03.so exists.
#     bash tools/collect_r04.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() { # tag, launches kept, bench args...
  local tag=$1 keep=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o r -- python3 $ROOT/bench.py --no-cpu-baseline --deliver none --sustained-s 0 --no-config4 "$@" > $OUT/r04_${tag}_bench_under_rocprof.json 2> $OUT/prof_$tag.err
  python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/prof_$tag/*results.db $OUT/prof_$tag/*/*results.db 2>/dev/null | head -1) --last $keep > $OUT/r04_${tag}_kernel_stats_timed.csv
  rm -rf $OUT/prof_$tag
}
PART=${1:-all}   # pmc | rest | all  (one gpurun call holds 20 minutes: the PMC passes alone take about ten)
if [ "$PART" != rest ]; then
echo "== pmc (first: the bench lines below read profiles/r04_pmc.json and check its source hash)"
cd $ROOT
for p in runs tiles; do
  bash tools/pmc_r04.sh base $p 1920 1080 300 > $OUT/pmc_1080p_$p.txt 2>&1
  bash tools/pmc_r04.sh base $p 3840 2160 300 > $OUT/pmc_4k_$p.txt 2>&1
done
python3 tools/pmc_record_r04.py gpurun_out/pmc4_base_runs_1920x1080 gpurun_out/pmc4_base_tiles_1920x1080 gpurun_out/pmc4_base_runs_3840x2160 gpurun_out/pmc4_base_tiles_3840x2160 > $OUT/pmc_record.log 2>&1
cp profiles/r04_pmc.json $OUT/r04_pmc.json
{ echo "# PMC passes of both encode kernels and of k_assemble behind them on the shipped tree (tools/pmc_r04.sh: one rocprofv3 --pmc run per counter"
  echo "# group, --kernel-trace only beside it), per-dispatch means; 300 frames per launch.  L1->L2 read requests = TCP_TCC_READ_REQ_sum; pixel lines = W*H*3*300/128."
  for d in runs_1920x1080 tiles_1920x1080 runs_3840x2160 tiles_3840x2160; do echo; echo "===== $d"; cat gpurun_out/pmc4_base_$d/summary.txt; done; } > $OUT/r04_memory_path_pmc.txt
fi
[ "$PART" = pmc ] && exit 0
cd /tmp
echo "== stats"
stats 1080p 200
stats 4k 60 --width 3840 --height 2160 --steps 60 --warmup 20
echo "== plain bench"
cd $ROOT
python3 bench.py > $OUT/r04_1080p_bench.json 2> /dev/null
python3 bench.py --path runs --no-cpu-baseline --no-config4 > $OUT/r04_1080p_runs_bench.json 2> /dev/null
python3 bench.py --no-cpu-baseline --width 3840 --height 2160 --steps 60 --warmup 20 > $OUT/r04_4k_bench.json 2> /dev/null
echo "== the bench as the driver runs it"
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d.get('sustained') or {}
print('driver-style run $i: fps', round(d['value']), 'ms_per_step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms'), 'frac', r['frac'], 'step_frac', r['step_frac'], '| sustained fps', s.get('value'), 'power', s.get('power'), '| config4 frac', (d.get('config4') or {}).get('frac'), 'step_frac', (d.get('config4') or {}).get('step_frac'))"; done > $OUT/r04_bench_driver_style.txt
cat $OUT/r04_bench_driver_style.txt
if [ -f build/libencoder_r03.so ]; then
  echo "== sustained A/B against the round-3 library"
  { echo "# tools/sustained.py r03 base: 500 untimed + 500 timed back-to-back steps per variant and round, wall time per step (encode + everything behind it)";
    python3 tools/sustained.py r03 base --rounds 4 2>&1 | tail -2;
    echo "# 100 x 3840x2160"; python3 tools/sustained.py r03 base --rounds 3 --w 3840 --h 2160 --n 100 2>&1 | tail -2; } > $OUT/r04_sustained_vs_r03.txt
  cat $OUT/r04_sustained_vs_r03.txt
fi
head -6 $OUT/r04_1080p_kernel_stats_timed.csv $OUT/r04_4k_kernel_stats_timed.csv
cut -c1-700 $OUT/r04_1080p_bench.json

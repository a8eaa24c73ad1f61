#!/bin/bash
# Round evidence on the GPU box: rocprofv3 kernel statistics of the bench command (same process that prints the JSON line)
# and the PMC passes, for BASELINE configs 3 (300 x 1080p) and 4 (300 x 4K).  Output under gpurun_out/<tag>/; copy the
# summaries into profiles/ afterwards (tools/pmc_record.py writes profiles/r02_pmc.json from the PMC directories).
#   bash tools/collect_profiles.sh r02
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== stats 1080p"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof1080 -o r -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_1080p_under_rocprof.json 2> $OUT/prof1080.err
python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/prof1080/*results.db $OUT/prof1080/*/*results.db 2>/dev/null | head -1) --last 200 > $OUT/${TAG}_1080p_kernel_stats_timed.csv
echo "== stats 4k"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof4k -o r -- python3 $ROOT/bench.py --no-cpu-baseline --width 3840 --height 2160 --steps 60 --warmup 20 > $OUT/bench_4k_under_rocprof.json 2> $OUT/prof4k.err
python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/prof4k/*results.db $OUT/prof4k/*/*results.db 2>/dev/null | head -1) --last 60 > $OUT/${TAG}_4k_kernel_stats_timed.csv
echo "== plain bench"; python3 $ROOT/bench.py > $OUT/bench_1080p.json 2> /dev/null
python3 $ROOT/bench.py --no-cpu-baseline --width 3840 --height 2160 --steps 60 --warmup 20 > $OUT/bench_4k.json 2> /dev/null
echo "== pmc"; cd $ROOT
bash tools/pmc.sh ${TAG}_1080p > $OUT/pmc_1080p.txt 2>&1
bash tools/pmc.sh ${TAG}_4k --width 3840 --height 2160 > $OUT/pmc_4k.txt 2>&1
head -4 $OUT/${TAG}_1080p_kernel_stats_timed.csv $OUT/${TAG}_4k_kernel_stats_timed.csv
cat $OUT/bench_1080p_under_rocprof.json $OUT/bench_4k.json

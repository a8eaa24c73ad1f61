#!/bin/bash
# The rest of the round-3 evidence, on the GPU box, into gpurun_out/r03x/ (copy into profiles/ afterwards):
#   run-kernel bench line + kernel statistics at 1080p, folder-level CLI bench, host-path / side-kernel bench, two-rank gloo
#   rehearsal, sustained runs-vs-tiles, power trace, phase stamps of the tile kernel, GPU test tail, fuzz soak.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03x
mkdir -p $OUT
cd $ROOT
echo "== pytest"; timeout -k 10 700 python3 -m pytest tests -q -m gpu > $OUT/r03_pytest_gpu.txt 2>&1; tail -2 $OUT/r03_pytest_gpu.txt
echo "== runs at 1080p"
python3 bench.py --path runs --no-cpu-baseline > $OUT/r03_1080p_runs_bench.json 2> /dev/null
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_runs -o r -- python3 $ROOT/bench.py --no-cpu-baseline --deliver none --path runs > $OUT/r03_1080p_runs_bench_under_rocprof.json 2> $OUT/prof_runs.err
  python3 $ROOT/tools/rocpd_stats.py $(ls $OUT/prof_runs/*results.db $OUT/prof_runs/*/*results.db 2>/dev/null | head -1) --last 200 > $OUT/r03_1080p_runs_kernel_stats_timed.csv; rm -rf $OUT/prof_runs )
echo "== sustained"
{ echo "# tools/sustained.py: us per step (encode + layout + gather), 500 settle + 500 timed launches per variant and round"
  timeout -k 10 200 python3 tools/sustained.py base:tiles base:runs --rounds 5 2>&1 | grep -v amdgpu.ids
  echo "# 3840x2160, 100 frames per step"; timeout -k 10 200 python3 tools/sustained.py base:tiles base:runs --w 3840 --h 2160 --n 100 --settle 300 --launches 300 2>&1 | grep -v amdgpu.ids
  echo "# 1280x720"; timeout -k 10 200 python3 tools/sustained.py base:tiles base:runs --w 1280 --h 720 2>&1 | grep -v amdgpu.ids; } > $OUT/r03_sustained.txt
echo "== power"
{ echo "# tools/power_trace.sh: bench.py --steps 6000 with rocm-smi sampled every 0.15 s: (shader clock) socket power in W; cap below"
  bash tools/power_trace.sh tiles; bash tools/power_trace.sh runs; rocm-smi --showmaxpower 2>/dev/null | grep -i "power (w)"
  echo "# tools/throttle_trace.sh tiles: amd-smi in the middle of a 9000-step pass"; bash tools/throttle_trace.sh tiles | grep -E "SOCKET_POWER|GFX_[0-7]:|^ *CLK: 2|HOTSPOT|UMC_ACT|GFX_ACTIVITY"; } > $OUT/r03_power_trace.txt 2>&1
echo "== stamps"
bash tools/mkvariant.sh stamps -DM1V_TILE_STAMPS > /dev/null 2>&1 && timeout -k 10 120 python3 tools/tile_stamps.py stamps 2>&1 | grep -v amdgpu.ids > $OUT/r03_tile_phase_stamps.txt
echo "== host path, cli, gloo"
timeout -k 10 300 python3 bench.py --host-path > $OUT/r03_host_path_and_side_kernels_bench.json 2> /dev/null
timeout -k 10 500 python3 bench.py --cli --frames 512 > $OUT/r03_cli_bench.json 2> $OUT/cli.err
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > $OUT/r03_n2_gloo_rehearsal_xgmi.json 2> /dev/null
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --backend gloo --gather host --steps 20 --warmup 5 > $OUT/r03_n2_gloo_rehearsal_host.json 2> /dev/null
echo "== fuzz"
timeout -k 10 300 python3 tests/fuzz_parity.py 240 3033 > $OUT/fuzz_b.txt 2>&1; tail -1 $OUT/fuzz_b.txt
ls -la $OUT | head -30

#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_round3.sh <tag>
# (1) kernel stats of the TIMED CYCLES ONLY (bench.py --no-secondary), (2) one lock-step batch of four evaluations:
# per-kernel totals + launch timeline, (3) K^-1 kernel traffic for B = 1 and B = 4 separately, (4) the bench line.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cycle_$tag -- python3 bench.py --steps 5 --warmup 2 --no-secondary > gpurun_out/bench_cycle_$tag.json 2> gpurun_out/bench_cycle_$tag.err || exit 1
cp "$(ls gpurun_out/prof_cycle_$tag/*/*kernel_stats.csv | head -1)" gpurun_out/cycle_kernel_stats_$tag.csv
for B in 1 4; do
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_eval_B${B}_$tag -- python3 tools/eval_kstats.py run 4096 $B > /dev/null 2>&1 || exit 1
  f="$(ls gpurun_out/trace_eval_B${B}_$tag/*/*kernel_trace.csv | head -1)"
  python3 tools/eval_kstats.py parse "$f" > gpurun_out/eval_kstats_B${B}_$tag.txt
  python3 tools/eval_kstats.py timeline "$f" > gpurun_out/eval_timeline_B${B}_$tag.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${c}_B${B}_$tag -- python3 tools/eval_kstats.py run 4096 $B > /dev/null 2>&1 || exit 1
  done
  python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE_B${B}_$tag gpurun_out/pmc_WRITE_SIZE_B${B}_$tag k_lauum_grad 4096 0 0 > gpurun_out/traffic_k_lauum_grad_B${B}_$tag.json
done
cat gpurun_out/eval_kstats_B4_$tag.txt
python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || exit 1
tail -c 1500 gpurun_out/bench_$tag.json

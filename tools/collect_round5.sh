#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_round5.sh <tag>      -> gpurun_out/r05_<tag>/
# (1) kernel stats of the TIMED CYCLES ONLY (bench.py --no-secondary) for the headline and config 2 (rocprofv3 --kernel-trace --stats),
# (2) PMC passes of the same command, each counter set in its own run, program directly after `--`:
#     FETCH_SIZE / WRITE_SIZE -> fabric traffic of k_trimul, k_cross_vv and the two assembly kernels (tools/pmc_traffic.py:
#     FETCH_SIZE x 2, WRITE_SIZE exact, KB -> bytes, MI355X_MICROARCH.md "HBM"); MFMA busy cycles -> utilisation,
# (3) the bench lines (headline with the CPU baseline, config 2, config 4 on one GPU, N = 8192).
tag=$1
out=gpurun_out/r05_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p $out
head=$(cat .git_head 2>/dev/null || echo "round 5")
cmd="python3 bench.py --steps 2 --warmup 1 --no-secondary"
if [ "$2" = "c5" ]; then
  # usage: bash tools/collect_round5.sh <tag> c5  -> the config-5 runs of profiles/r05_config5.txt (six seeds, the gated run,
  # thresholds 0.2 / 0.02, the 16-D run to the size cap) and the sampler kernels' time per step
  o=$out/config5_runs.txt
  python3 tools/config5_run.py steptime=1 > $out/sampler_step_times.txt 2>&1 || exit 1
  : > $o
  for s in 7 1 2 3 4 5; do python3 tools/config5_run.py seed=$s 2>&1 | tail -1 >> $o; done
  python3 tools/config5_run.py seed=7 clf=1 2>&1 | tail -1 >> $o
  python3 tools/config5_run.py seed=7 thr=0.2 ns_every=100 2>&1 | tail -1 >> $o
  python3 tools/config5_run.py seed=7 thr=0.02 ns_every=200 max_evals=4200 2>&1 | tail -1 >> $o
  python3 tools/config5_run.py dim=16 thr=0.5 max_evals=4400 max_gp=4096 ns_every=100 min_evals=800 seed=7 2>&1 | grep -v "^ " | cut -c1-420 >> $o
  cut -c1-330 $o
  exit 0
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cycle -- python3 bench.py --steps 5 --warmup 2 --no-secondary > $out/bench_cycle.json 2> $out/bench_cycle.err || exit 1
cp "$(ls $out/prof_cycle/*/*kernel_stats.csv | head -1)" $out/cycle_kernel_stats.csv
echo "kernel stats (headline) done" 
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cycle_small -- python3 bench.py --config small --steps 20 --warmup 3 --no-secondary > $out/bench_cycle_small.json 2> $out/bench_cycle_small.err || exit 1
cp "$(ls $out/prof_cycle_small/*/*kernel_stats.csv | head -1)" $out/cycle_kernel_stats_config2.csv
echo "kernel stats (config 2) done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-secondary > /dev/null 2>&1 || exit 1
  echo "pmc $c done"
done
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE "k_trimul(" 4096 65536 8192 "$cmd" "round 5, $head" > $out/traffic_k_trimul.json
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE "k_cross_vv" 4096 65536 8192 "$cmd" "round 5, $head" > $out/traffic_k_cross_vv.json
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE "k_kernel_matrix<0, false, 8, true>" 4096 65536 8192 "$cmd" "round 5, $head" > $out/traffic_k_kernel_matrix_0_false_8_true.json
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE "k_kernel_matrix<0, true, 8, true>" 4096 65536 8192 "$cmd" "round 5, $head" > $out/traffic_k_kernel_matrix_0_true_8_true.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-secondary > /dev/null 2>&1 || exit 1
python3 tools/pmc_mfma.py $out/pmc_mfma > $out/mfma_util.json
echo "pmc mfma done"
python3 bench.py --steps 20 --warmup 3 > $out/bench_headline.json 2> $out/bench_headline.err || exit 1
echo "bench headline done"
python3 bench.py --config small --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_config2.json 2> /dev/null || exit 1
python3 bench.py --config shard --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_config4_one_gpu.json 2> /dev/null || exit 1
python3 bench.py --config large --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_large.json 2> /dev/null || exit 1
rm -rf $out/prof_cycle $out/prof_cycle_small $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_mfma
tail -c 400 $out/bench_headline.json

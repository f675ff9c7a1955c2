#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_round6.sh <tag>      -> gpurun_out/r06_<tag>/
# (1) kernel stats of the TIMED CYCLES ONLY (bench.py --no-secondary): headline, config 2, and the headline with every factor
#     treated as ill conditioned (BOBE_REFINE_KAPPA=0: the blocked substitution's launches, k_blk_step),
# (2) PMC passes of the same command, each counter set in its own run, program directly after `--`:
#     FETCH_SIZE / WRITE_SIZE -> fabric traffic of k_trimul, k_blk_step,
#     k_cross_vv and the two assembly kernels (tools/pmc_traffic.py); MFMA busy cycles -> utilisation,
# (3) the bench lines (headline with the CPU baseline, config 2, config 4 on one GPU, N = 8192),
# (4) config 5 under the kernel trace + tools/config5_breakdown.py.
tag=$1
out=gpurun_out/r06_$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p $out
head=$(cat .git_head 2>/dev/null || echo "round 6")
cmd="python3 bench.py --steps 2 --warmup 1 --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cycle -- python3 bench.py --steps 5 --warmup 2 --no-secondary > $out/bench_cycle.json 2> $out/bench_cycle.err || exit 1
cp "$(ls $out/prof_cycle/*/*kernel_stats.csv | head -1)" $out/cycle_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cycle_small -- python3 bench.py --config small --steps 20 --warmup 3 --no-secondary > $out/bench_cycle_small.json 2> $out/bench_cycle_small.err || exit 1
cp "$(ls $out/prof_cycle_small/*/*kernel_stats.csv | head -1)" $out/cycle_kernel_stats_config2.csv
export BOBE_REFINE_KAPPA=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_cycle_subst -- python3 bench.py --steps 5 --warmup 2 --no-secondary > $out/bench_cycle_substitution.json 2> $out/bench_cycle_subst.err || exit 1
cp "$(ls $out/prof_cycle_subst/*/*kernel_stats.csv | head -1)" $out/cycle_kernel_stats_substitution.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmcs_$c -- python3 bench.py --steps 2 --warmup 1 --no-secondary > /dev/null 2>&1 || exit 1
done
python3 tools/pmc_traffic.py $out/pmcs_FETCH_SIZE $out/pmcs_WRITE_SIZE "k_blk_step" 4096 65536 32768 "BOBE_REFINE_KAPPA=0 $cmd" "round 6, $head" > $out/traffic_k_blk_step.json
unset BOBE_REFINE_KAPPA
echo "kernel stats + substitution traffic done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-secondary > /dev/null 2>&1 || exit 1
done
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE "k_trimul(" 4096 65536 8192 "$cmd" "round 6, $head" > $out/traffic_k_trimul.json
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE "k_cross_vv" 4096 65536 8192 "$cmd" "round 6, $head" > $out/traffic_k_cross_vv.json
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE "k_kernel_matrix<0, false, 8, true>" 4096 65536 8192 "$cmd" "round 6, $head" > $out/traffic_k_kernel_matrix_0_false_8_true.json
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE "k_kernel_matrix<0, true, 8, true>" 4096 65536 8192 "$cmd" "round 6, $head" > $out/traffic_k_kernel_matrix_0_true_8_true.json
# (the column-order A/B of k_trimul, profiles/r06_traffic_k_trimul_contig.json, was collected here with a knob that has since been removed)
echo "pmc traffic done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-secondary > /dev/null 2>&1 || exit 1
python3 tools/pmc_mfma.py $out/pmc_mfma > $out/mfma_util.json
export BOBE_REFINE_KAPPA=0
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma_s -- python3 bench.py --steps 2 --warmup 1 --no-secondary > /dev/null 2>&1 || exit 1
python3 tools/pmc_mfma.py $out/pmc_mfma_s > $out/mfma_util_substitution.json
unset BOBE_REFINE_KAPPA
echo "pmc mfma done"
python3 bench.py --steps 20 --warmup 3 > $out/bench_headline.json 2> $out/bench_headline.err || exit 1
echo "bench headline done"
python3 bench.py --config small --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_config2.json 2> /dev/null || exit 1
python3 bench.py --config shard --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_config4_one_gpu.json 2> /dev/null || exit 1
python3 bench.py --config large --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_large.json 2> /dev/null || exit 1
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c5 -- python3 tools/config5_run.py seed=7 > $out/config5_rocprof.txt 2>&1 || exit 1
cp "$(ls $out/prof_c5/*/*kernel_stats.csv | head -1)" $out/config5_kernel_stats.csv
python3 tools/config5_run.py seed=7 > $out/config5_plain.txt 2>&1 || exit 1
python3 tools/config5_breakdown.py $out/config5_kernel_stats.csv $out/config5_rocprof.txt $out/config5_plain.txt > $out/config5_breakdown.txt
rm -rf $out/prof_cycle $out/prof_cycle_small $out/prof_cycle_subst $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmcs_FETCH_SIZE $out/pmcs_WRITE_SIZE $out/pmc_mfma $out/pmc_mfma_s $out/prof_c5
head -12 $out/config5_breakdown.txt
tail -c 300 $out/bench_headline.json

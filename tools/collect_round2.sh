#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_round2.sh <tag>
# kernel stats of the headline bench (default command), PMC passes (MFMA utilisation with the fit in lock step; HBM
# traffic of the dominant kernel), a clean bench line with the CPU baseline, the lock-step scaling tables.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof_$tag.json 2> gpurun_out/bench_prof_$tag.err || exit 1
python tools/kstats.py $tag 16
echo "--- PMC: MFMA utilisation (fit in lock-step rounds)"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --fit-mode batch > gpurun_out/bench_pmc_$tag.json 2> gpurun_out/bench_pmc_$tag.err || exit 1
python tools/pmc_mfma.py gpurun_out/pmc_mfma_$tag > gpurun_out/mfma_util_$tag.json
head -c 1500 gpurun_out/mfma_util_$tag.json
echo "--- PMC: HBM traffic"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${c}_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --fit-concurrency 1 > gpurun_out/bench_pmc_${c}_$tag.json 2> gpurun_out/bench_pmc_${c}_$tag.err || exit 1
done
for k in "k_trimul(" "k_kernel_matrix<0, false" "k_kernel_matrix<0, true" "k_syrk_trail<64" "k_lauum_grad"; do
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE_$tag gpurun_out/pmc_WRITE_SIZE_$tag "$k" 4096 65536 8192 > "gpurun_out/traffic_$(echo $k | tr -c 'a-zA-Z0-9_\n' '_')_$tag.json"
done
cat gpurun_out/traffic_k_trimul__$tag.json
echo "--- bench with CPU baseline"
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || exit 1
cat gpurun_out/bench_$tag.json
echo "--- config 4 on one GPU, config 2"
python bench.py --config shard --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_shard_$tag.json 2>> gpurun_out/bench_$tag.err
python bench.py --config small --steps 5 --warmup 2 > gpurun_out/bench_small_$tag.json 2>> gpurun_out/bench_$tag.err
python tools/chol_step_check.py > gpurun_out/chol_lockstep_$tag.txt 2>&1
tail -22 gpurun_out/chol_lockstep_$tag.txt

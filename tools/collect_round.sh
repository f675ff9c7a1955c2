#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_round.sh <tag>
# GPU parity suite, kernel stats of the headline bench, one PMC pass for MFMA utilisation, a clean bench line with
# the CPU baseline, and the scaling tables.  Everything lands in gpurun_out/.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/pytest_gpu_$tag.log 2>&1 || { tail -20 gpurun_out/pytest_gpu_$tag.log; exit 1; }
tail -2 gpurun_out/pytest_gpu_$tag.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof_$tag.json 2> gpurun_out/bench_prof_$tag.err || exit 1
python tools/kstats.py $tag 14
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --fit-concurrency 1 > gpurun_out/bench_pmc_$tag.json 2> gpurun_out/bench_pmc_$tag.err || exit 1
python tools/pmc_mfma.py gpurun_out/pmc_mfma_$tag > gpurun_out/mfma_util_$tag.json
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || exit 1
cat gpurun_out/bench_$tag.json
python bench.py --steps 5 --warmup 2 --fit-concurrency 1 --no-cpu-baseline > gpurun_out/bench_seq_$tag.json 2>> gpurun_out/bench_$tag.err || exit 1
python tools/chol_scaling.py > gpurun_out/chol_scaling_$tag.txt 2>&1
python tools/batch_scaling.py 4096 > gpurun_out/batch_scaling_$tag.txt 2>&1
cat gpurun_out/chol_scaling_$tag.txt gpurun_out/batch_scaling_$tag.txt

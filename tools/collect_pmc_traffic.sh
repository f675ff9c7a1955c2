#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_pmc_traffic.sh <tag>
# two separate PMC passes (FETCH_SIZE, WRITE_SIZE) + one kernel-stats pass of the same bench command
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${c}_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --fit-concurrency 1 > gpurun_out/bench_pmc_${c}_$tag.json 2> gpurun_out/bench_pmc_${c}_$tag.err || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_seq_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --fit-concurrency 1 > /dev/null 2>&1 || exit 1
for k in "k_trimul(" "k_kernel_matrix<0, false" "k_kernel_matrix<0, true" "k_syrk_trail<64" "k_lauum_grad"; do
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE_$tag gpurun_out/pmc_WRITE_SIZE_$tag "$k" 4096 65536 8192 > "gpurun_out/traffic_$(echo $k | tr -c 'a-zA-Z0-9_\n' '_')_$tag.json"
done
python tools/kstats.py seq_$tag 24
ls gpurun_out | grep traffic_

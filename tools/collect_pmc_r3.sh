#!/bin/bash
# usage (GPU box, repo root): bash tools/collect_pmc_r3.sh <tag>
# PMC passes of the timed cycles only (bench.py --no-secondary): MFMA utilisation per kernel with the fit in lock step,
# HBM / fabric traffic of the sweep's GEMM and of the assembly kernels.  One counter group per pass, kernel trace only.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma_$tag -- python3 bench.py --steps 2 --warmup 1 --no-secondary --fit-mode batch > /dev/null 2> gpurun_out/pmc_mfma_$tag.err || exit 1
python3 tools/pmc_mfma.py gpurun_out/pmc_mfma_$tag > gpurun_out/mfma_util_$tag.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${c}_$tag -- python3 bench.py --steps 1 --warmup 0 --no-secondary --fit-concurrency 1 > /dev/null 2> gpurun_out/pmc_${c}_$tag.err || exit 1
done
for k in "k_trimul(" "k_kernel_matrix<0, false" "k_kernel_matrix<0, true" "k_syrk_trail<64"; do
  python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE_$tag gpurun_out/pmc_WRITE_SIZE_$tag "$k" 4096 65536 8192 > "gpurun_out/traffic_$(echo $k | tr -c 'a-zA-Z0-9_\n' '_')_$tag.json"
done
head -c 1200 gpurun_out/mfma_util_$tag.json; cat gpurun_out/traffic_k_trimul__$tag.json

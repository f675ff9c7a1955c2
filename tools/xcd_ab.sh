# A/B of the XCD tile shares (BOBE_XCD_SHARES=0/1): lock-step factorisation and value+gradient batches, then the
# per-kernel totals of one four-evaluation batch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for x in 0 1; do
  echo "== BOBE_XCD_SHARES=$x"
  BOBE_XCD_SHARES=$x timeout -k 10 300 python tools/chol_step_check.py 2>&1 | grep -E "N= 4096|N= 8192|B=4|B=8|B=1:|FAIL|ALL OK" 
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kst4 -- python tools/eval_kstats.py run 4096 4 > /dev/null 2>&1
python tools/eval_kstats.py parse $(ls gpurun_out/kst4/*/*kernel_trace.csv | head -1)

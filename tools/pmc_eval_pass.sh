# PMC passes over one lock-step batch of four evaluations at N=4096 (one counter group per pass)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in "LdsBankConflict" "MemUnitStalled" "LdsUtil" "MfmaUtil"; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python tools/eval_kstats.py run 4096 4 > /dev/null 2>&1 || echo "pass $c failed"
  python tools/pmc_generic.py /tmp/pmc_$c k_lauum k_trtri k_syrk_trail\<64 k_trimul
done

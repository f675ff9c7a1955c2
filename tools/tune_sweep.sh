#!/bin/bash
# usage (GPU box, repo root): bash tools/tune_sweep.sh   - A/B material: the lone factorisation and the sequential fit of the
# bench line under the filler switches
for v in "A=0" "BOBE_FILL=0" "BOBE_FILL=0 BOBE_FILL_INV=0" "BOBE_FILL_INV=0" "A=0"; do
  echo "$v: $(env $v python tools/lockstep_time.py 4096 2>&1 | grep -E 'x1' | sed 's/N=4096 //')"
done
run() { env "$@" python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['fit_ms'].items()}, {k: round(x,2) for k,x in d['sub_ms'].items()})"; }
run A=0
run BOBE_FILL=0
run BOBE_FILL=0 BOBE_FILL_INV=0
run A=0

"""The factorisation has several launch sequences (one-launch panel - with three or four row strips per workgroup - or
k_potf2 + k_trsm_panel; XCD tile shares or the row-major tile order; alone, on a slot, in a lock-step batch).  They must all produce the same bits: the library reads
its tuning environment once per process, so each variant runs tools/bits_snapshot.py in its own process and the digests
of L, the MLL value, its gradient, the batched values and the predictions are compared."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digests(extra_env):
    env = dict(os.environ, BITS_MAX_N="1500", **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bits_snapshot.py"), "print"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_every_launch_sequence_gives_the_same_bits():
    base = _digests({})
    assert len(base) >= 30
    for variant in ({"BOBE_XCD_SHARES": "0"}, {"BOBE_CHOL_LEGACY": "1"}, {"BOBE_PAIR_MIN": "0"}, {"BOBE_SWEEP_OVERLAP": "1"},
                    {"BOBE_PANEL_STRIPS": "4"}):
        other = _digests(variant)
        differing = [k for k in base if base[k] != other[k]]
        assert not differing, (variant, differing)


def test_deferred_updates_in_the_panel_shadows_do_not_move_a_bit():
    """potrf() lets trailing-update tiles of far block columns, and tiles of the inverse that follows (its diagonal blocks,
    the T / R stages of its recursion), ride in the panel launches as filler workgroups (k_chol_panel<., true>, chol_plan
    in bobe_gp.hip).  Where a tile is computed must not change what it holds: the factor, MLL, gradient, batch results and
    predictions with both kinds of filler forced on everywhere (BOBE_FILL=2, BOBE_FILL_INV=2: lone and lock-step, ragged and
    full sizes, both kernels) equal, bit for bit, those with every update and every inverse tile in its own launch (=0).
    The switches are read once per process, hence the child interpreters."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sizes = "300:3:rbf,1500:8:matern,2500:5:rbf,3200:6:matern,4096:8:rbf,5000:4:rbf"
    digests = {}
    for mode in ("0", "2", "1"):
        env = dict(os.environ, BOBE_FILL=mode, BOBE_FILL_INV=mode, BITS_SIZES=sizes, BOBE_LOCKSTEP_MIN_N="1024")
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "bits_snapshot.py"), "print"], env=env,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        digests[mode] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert digests["0"].keys() == digests["2"].keys() and len(digests["0"]) == 6 * 7 + 2      # (+ the sweeps at N <= 2048)
    assert digests["2"] == digests["0"]
    assert digests["1"] == digests["0"]                       # the default (fillers only where they pay)

"""The factorisation has several launch sequences (one-launch panel or k_potf2 + k_trsm_panel; XCD tile shares or the
row-major tile order; alone, on a slot, in a lock-step batch).  They must all produce the same bits: the library reads
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
    for variant in ({"BOBE_XCD_SHARES": "0"}, {"BOBE_CHOL_LEGACY": "1"}, {"BOBE_PAIR_MIN": "0"}):
        other = _digests(variant)
        differing = [k for k in base if base[k] != other[k]]
        assert not differing, (variant, differing)

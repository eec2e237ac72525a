"""The golden fixtures must be reproducible from the reference: `gen_golden.py --check` regenerates every fixture in a
temp dir (importing /root/reference, PYTHONHASHSEED pinned by the script itself) and compares array for array, bit for
bit, with what is committed.  Only runs where the reference exists (the build container), never on the GPU box."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("hashseed", ["", "4242"])
def test_committed_goldens_equal_a_regeneration(hashseed):
    env = dict(os.environ)
    env.pop("PYTHONHASHSEED", None)
    if hashseed:
        env["PYTHONHASHSEED"] = hashseed          # the script must override whatever the caller's environment says
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "gen_golden.py"), "--check"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "reproduces every committed fixture bit for bit" in r.stdout

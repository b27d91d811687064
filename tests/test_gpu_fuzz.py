"""A short run of the randomised parity sweep (tools/fuzz_parity.py): random sizes and kinds of content, default options
against the oracle's bytes, optional coders through the oracle's (or libwebp's) decoder, every stream decoded on the GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_images_for_a_quarter_of_a_minute():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "15", "2024"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failures" in r.stdout.splitlines()[-1]

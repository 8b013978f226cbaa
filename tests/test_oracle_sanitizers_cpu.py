"""CPU: every C entry point of the oracle under AddressSanitizer + UndefinedBehaviorSanitizer
(oracle/sanitize_check.c; SURVEY section 5 asks for it: the checker gets checked)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None and shutil.which("cc") is None, reason="no C compiler")
def test_oracle_under_asan_ubsan():
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "sanitize"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "sanitize_check: ok" in p.stdout

"""pgext/ndb_mirror_cache.h — the per-backend mirror cache of the PostgreSQL glue, compiled without PostgreSQL around the
harness tests/mirror_cache_harness.c (when every retired slot is pinned, a stale mirror must never be served)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stale_mirrors_are_never_served(tmp_path):
    exe = tmp_path / "mirror_cache_harness"
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                           os.path.join(ROOT, "tests", "mirror_cache_harness.c"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "mirror_cache_harness: OK" in out.stdout


def test_the_glue_uses_the_tested_cache():
    src = open(os.path.join(ROOT, "pgext", "ndbhip_glue.c")).read()
    assert '#include "ndb_mirror_cache.h"' in src and "ndb_mc_slot(" in src
    assert "mirrors[i].stamp = stamp" not in src          # the re-stamp after a failed drop is gone

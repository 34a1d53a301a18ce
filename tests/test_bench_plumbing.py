"""CPU: bench.py's MIOpen find-db seeding (plumbing; no GPU, MIOpen is never initialised here)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_seed_miopen_db_copies_the_shipped_db_to_a_scratch_dir(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    src = os.path.join(ROOT, "wssdl_bus_amd", "miopen_db", "config")
    assert os.path.isdir(src) and any(f.endswith(".ufdb.txt") for f in os.listdir(src))
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("MIOPEN_CUSTOM_CACHE_DIR", raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    tempfile.tempdir = None                                     # re-read TMPDIR
    try:
        dst = bench.seed_miopen_db()
        assert dst and dst.startswith(str(tmp_path))
        assert os.environ["MIOPEN_USER_DB_PATH"] == os.path.join(dst, "config")
        assert sorted(os.listdir(os.environ["MIOPEN_USER_DB_PATH"])) == sorted(os.listdir(src))
        assert os.path.isdir(os.environ["MIOPEN_CUSTOM_CACHE_DIR"])
        # a caller that already points MIOpen somewhere is left alone
        monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
        assert bench.seed_miopen_db() is None and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    finally:
        tempfile.tempdir = None

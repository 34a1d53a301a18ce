"""CPU: bench.py's MIOpen find-db seeding (plumbing; no GPU, MIOpen is never initialised here)."""
import importlib
import os
import stat
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_seed_miopen_db_copies_the_shipped_db_to_a_private_scratch_dir(tmp_path, monkeypatch):
    bench = _bench()
    src = os.path.join(ROOT, "wssdl_bus_amd", "miopen_db", "config")
    assert os.path.isdir(src) and any(f.endswith(".ufdb.txt") for f in os.listdir(src))
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("MIOPEN_CUSTOM_CACHE_DIR", raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    tempfile.tempdir = None                                     # re-read TMPDIR
    try:
        dst, note = bench.seed_miopen_db()
        assert dst and dst.startswith(str(tmp_path)) and "miopen_db" in note
        assert stat.S_IMODE(os.stat(dst).st_mode) == 0o700          # mkdtemp: private, unpredictable name
        assert os.environ["MIOPEN_USER_DB_PATH"] == os.path.join(dst, "config")
        assert sorted(os.listdir(os.environ["MIOPEN_USER_DB_PATH"])) == sorted(os.listdir(src))
        assert os.path.isdir(os.environ["MIOPEN_CUSTOM_CACHE_DIR"])
        assert bench.miopen_db_grew(dst) is False
        with open(os.path.join(dst, "config", os.listdir(src)[0]), "a") as fh:
            fh.write("x")                                           # what a search would do
        assert bench.miopen_db_grew(dst) is True
        # two calls never share a directory
        monkeypatch.delenv("MIOPEN_USER_DB_PATH")
        monkeypatch.delenv("MIOPEN_CUSTOM_CACHE_DIR")
        dst2, _ = bench.seed_miopen_db()
        assert dst2 and dst2 != dst
        # a caller that already points MIOpen somewhere is left alone
        monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
        assert bench.seed_miopen_db()[0] is None and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    finally:
        tempfile.tempdir = None


def test_seed_miopen_db_fails_closed_on_another_miopen_build(tmp_path, monkeypatch):
    """A find-db written by another MIOpen build would be ignored by MIOpen, which would then search for
    minutes on every rank: the seeding refuses and bench.py falls back to the heuristic solver choice."""
    bench = _bench()
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("MIOPEN_CUSTOM_CACHE_DIR", raising=False)
    fake = tmp_path / "repo"
    cfgdir = fake / "wssdl_bus_amd" / "miopen_db" / "config"
    cfgdir.mkdir(parents=True)
    (cfgdir / "gfx950100.HIP.9_9_9_19990101-1-1-gdeadbeef00.ufdb.txt").write_text("")
    monkeypatch.setattr(bench, "ROOT", str(fake))
    dst, note = bench.seed_miopen_db()
    assert dst is None and "9_9_9" in note
    assert "MIOPEN_USER_DB_PATH" not in os.environ

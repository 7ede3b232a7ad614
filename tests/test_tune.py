"""The table of measured launch configurations (ops/tune.py): exact and nearest-shape lookups, the shipped table's
sanity, writing a table out.  Host logic only -- the measurements themselves need a GPU (tools/tune_conv.py)."""
import importlib
import json
import os

import pytest


@pytest.fixture(autouse=True)
def _restore_module():
    """Leave ops.tune as a fresh import of the real environment behind every test."""
    saved = {k: os.environ.get(k) for k in ("DTFA_TUNE_DB", "DTFA_AUTOTUNE")}
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    from distributed_tensorflow_amd.ops import tune

    importlib.reload(tune)


def _fresh(monkeypatch, path, mode="0"):
    monkeypatch.setenv("DTFA_TUNE_DB", str(path))
    monkeypatch.setenv("DTFA_AUTOTUNE", mode)
    from distributed_tensorflow_amd.ops import tune

    return importlib.reload(tune)


def test_lookup_exact_nearest_and_off(tmp_path, monkeypatch):
    db = tmp_path / "t.json"
    db.write_text(json.dumps({"igemm": {"401408,256,2304,9,1,0,1,0": {"best": [256, 256, 64, 2, 1], "us": [500.0, 480.0]},
                                        "100352,256,2304,9,1,0,1,0": [128, 128, 32, 3, 0]},
                              "wgrad": {"401408,256,2304,256,1,1,0": {"best": [256, 128, 32, 56]}}}))
    tune = _fresh(monkeypatch, db)
    assert tune.lookup("igemm", "401408,256,2304,9,1,0,1,0") == (256, 256, 64, 2, 1)
    assert tune.lookup("igemm", "100352,256,2304,9,1,0,1,0") == (128, 128, 32, 3, 0)      # bare-list entries load too
    # another batch size of the same layer borrows the nearest measured row count within a factor of two ...
    assert tune.lookup("igemm", "301056,256,2304,9,1,0,1,0") == (256, 256, 64, 2, 1)
    assert tune.lookup("igemm", "150000,256,2304,9,1,0,1,0") == (128, 128, 32, 3, 0)
    # ... and nothing beyond that, nor across layers or kernels
    assert tune.lookup("igemm", "1605632,256,2304,9,1,0,1,0") is None
    assert tune.lookup("igemm", "401408,256,2304,9,1,0,3,0") is None
    assert tune.lookup("wgrad", "401408,256,2304,256,1,1,0") == (256, 128, 32, 56)
    assert tune.lookup("wgrad", "401408,256,2304,256,2,2,0") is None
    assert tune.stats()["igemm"] == 2
    off = _fresh(monkeypatch, db, mode="off")
    assert off.lookup("igemm", "401408,256,2304,9,1,0,1,0") is None
    # while measuring, a missing shape is measured rather than borrowed
    meas = _fresh(monkeypatch, db, mode="1")
    assert meas.measuring() and meas.lookup("igemm", "301056,256,2304,9,1,0,1,0") is None


def test_unreadable_table_means_rules(tmp_path, monkeypatch):
    db = tmp_path / "broken.json"
    db.write_text("{ not json")
    tune = _fresh(monkeypatch, db)
    assert tune.lookup("igemm", "1,2,3") is None and tune.stats()["igemm"] == 0


def test_save_merges_with_existing(tmp_path, monkeypatch):
    db = tmp_path / "t.json"
    db.write_text(json.dumps({"igemm": {"64,64,64,1,1,1,0,0": {"best": [64, 64, 32, 3, 0]}}, "wgrad": {}}))
    tune = _fresh(monkeypatch, db, mode="1")
    tune.lookup("igemm", "x")                         # loads the table
    tune._db["igemm"]["128,64,64,1,1,1,0,0"] = [128, 64, 32, 3, 0]
    out = tmp_path / "out.json"
    out.write_text(json.dumps({"igemm": {"9,9,9": {"best": [1, 2, 3, 4, 0]}}, "wgrad": {}}))
    tune.save(str(out))
    d = json.loads(out.read_text())
    assert set(d["igemm"]) == {"9,9,9", "64,64,64,1,1,1,0,0", "128,64,64,1,1,1,0,0"}


def test_shipped_table_is_well_formed(monkeypatch):
    """Every entry of the table in the package names a configuration the launchers can be asked for."""
    monkeypatch.delenv("DTFA_TUNE_DB", raising=False)
    monkeypatch.setenv("DTFA_AUTOTUNE", "0")
    from distributed_tensorflow_amd.ops import tune

    tune = importlib.reload(tune)
    path = os.path.join(os.path.dirname(tune.__file__), "tuned_gfx950.json")
    d = json.load(open(path))
    tiles = {(128, 192), (128, 128), (128, 64), (64, 64), (256, 128), (256, 256)}
    assert len(d["igemm"]) > 100 and len(d["wgrad"]) > 50
    for k, v in d["igemm"].items():
        assert len(k.split(",")) == 8 and all(p.lstrip("-").isdigit() for p in k.split(","))
        bm, bn, bk, nst, tail = v["best"]
        assert (bm, bn) in tiles and bk in (32, 64) and nst in (2, 3) and tail in (0, 1), (k, v)
    from distributed_tensorflow_amd.ops import conv as cv

    for k, v in d["wgrad"].items():
        assert len(k.split(",")) == 7
        bmw, bnw, bkp, splits = v["best"]
        assert (bmw, bnw, bkp) in cv._WGRAD_RESIDENT and 1 <= splits <= 4096, (k, v)


def test_merging_a_new_measurement_into_the_shipped_table(tmp_path):
    """tools/merge_tuned.py: an entry changes only when the new run's winner beats the shipped choice, timed in that same
    run, by more than the margin; notes beside the tables survive; entries without a reduction split keep five elements."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = tmp_path / "table.json"
    table.write_text(json.dumps({
        "_about": "note",
        "igemm": {"1,2,3,4,5,6,7,0": {"best": [128, 128, 32, 3, 1], "us": [10.0, 10.0]}},
        "wgrad": {"a": {"best": [256, 128, 32, 56], "us": [100.0, 100.0]},
                  "b": {"best": [128, 128, 32, 64], "us": [50.0, 50.0]}}}))
    measured = tmp_path / "new.json"
    measured.write_text(json.dumps({
        "igemm": {"1,2,3,4,5,6,7,0": {"best": [128, 128, 32, 3, 1, 1], "rule": [128, 128, 32, 3, 1, 1],
                                      "us": {"128,128,32,3,1,1": 9.0, "128,64,32,3,0": 9.5}}},
        "wgrad": {"a": {"best": [256, 256, 64, 56], "rule": [256, 128, 32, 56],     # 10 % faster: replaces
                        "us": {"256,128,32,56": 100.0, "256,256,64,56": 90.0}},
                  "b": {"best": [128, 128, 64, 64], "rule": [128, 128, 32, 64],     # 1 % faster: within the margin
                        "us": {"128,128,32,64": 50.0, "128,128,64,64": 49.5}},
                  "c": {"best": [64, 64, 32, 8], "rule": [64, 64, 32, 16],           # a new key
                        "us": {"64,64,32,16": 20.0, "64,64,32,8": 12.0}},
                  "d": {"best": [64, 64, 32, 8]}}}))                                  # carried over, not measured
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "merge_tuned.py"), str(measured), "--table", str(table)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(table.read_text())
    assert out["_about"] == "note"
    assert out["wgrad"]["a"] == {"best": [256, 256, 64, 56], "us": [100.0, 90.0]}
    assert out["wgrad"]["b"]["best"] == [128, 128, 32, 64]
    assert out["wgrad"]["c"] == {"best": [64, 64, 32, 8], "us": [20.0, 12.0]}
    assert "d" not in out["wgrad"]
    assert out["igemm"]["1,2,3,4,5,6,7,0"]["best"] == [128, 128, 32, 3, 1]

"""bench.py's measurement helpers that need no GPU: the committed counter summary is found and parsed, and the front-end's vector-ALU
fraction (roofline.valu) is formed from it as DESIGN.md section 4 states."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_committed_pmc_summary_feeds_the_roofline_block():
    b = _bench()
    entry, name = b.committed_pmc("k_lk_track_fb")
    assert name and name.endswith("_pmc_summary.json") and entry
    for key in ("hbm_bytes_per_launch", "grid_threads", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "launches"):
        assert key in entry, key
    # traffic per point as the bench forms it: within 10 % of the algorithmic 8 480 B (the XCD-chunked mapping keeps re-reads out)
    per_point = entry["hbm_bytes_per_launch"] / (entry["grid_threads"] / 64.0)
    assert 0.9 * 8480 < per_point < 1.1 * 8480


def test_frontend_valu_fraction():
    b = _bench()
    _, name = b.committed_pmc("k_lk_track_fb")
    v = b.frontend_valu(name, 8.0, 70000.0)
    assert v and 0.0 < v["frac"] < 1.0 and 0.3 < v["lk_share"] < 0.9
    allp = json.load(open(os.path.join(ROOT, "profiles", name)))
    lk = allp["k_lk_track_fb"]
    per_step = sum(e.get("SQ_INSTS_VALU", 0.0) * e.get("launches", 0.0) for k, e in allp.items() if k.startswith("k_") and k != "k_reproj_eval")
    assert abs(v["wave_instructions_per_frame"] - per_step / lk["launches"] / 8.0) < 1.0
    assert b.frontend_valu("no_such_summary.json", 8.0, 1.0) is None

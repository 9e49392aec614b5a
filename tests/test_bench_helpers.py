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
    assert name and "_pmc_summary" in name and name.endswith(".json") and entry
    # the newest round's summary at the launch shape of the run (192 streams per launch: the device engine) is the one picked
    e2, n2 = b.committed_pmc("k_lk_track_fb", 192.0)
    if b.pmc_provenance("r06_pmc_summary.json", 192.0) is None:  # (only while the kernel sources are the ones the counters were collected on)
        assert n2 == "r06_pmc_summary.json" and e2
    for key in ("hbm_bytes_per_launch", "grid_threads", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "launches"):
        assert key in entry, key
    # traffic per point as the bench forms it, against the algorithmic 8 480 B (the XCD-chunked mapping keeps re-reads out: 31.7 KB before
    # it).  0.98x with 8 groups x 64 streams (round 2), 1.16x with 12 x 64 (round 3), 1.85x in round 5 — 1.5-2 KB of that were register
    # spills (WRITE_SIZE 1.7 KB per point for a kernel that stores 21 bytes per point) — and 0.99x in round 6: the kernel has no scratch any
    # more (96 VGPRs, rounding constants in SGPRs, set-up cache and pair variant gone).  ADVICE r5: the guard is tight again, and the
    # written bytes are bounded on their own, so that a spill cannot hide inside the read traffic.
    meta = json.load(open(os.path.join(ROOT, "profiles", name))).get("_meta") or {}
    points = meta.get("lk_active_points_per_launch") or entry["grid_threads"] / 64.0
    per_point = entry["hbm_bytes_per_launch"] / points
    assert 0.9 * 8480 < per_point < 1.25 * 8480, per_point
    if name.startswith("r06"):
        assert entry["WRITE_SIZE"] * 1024.0 / points < 100.0, entry["WRITE_SIZE"] * 1024.0 / points  # (KB per launch: 21 B per point are results)


def test_frontend_valu_fraction():
    b = _bench()
    _, name = b.committed_pmc("k_lk_track_fb")
    allp = json.load(open(os.path.join(ROOT, "profiles", name)))
    spl = float((allp.get("_meta") or {}).get("streams_per_launch") or 8.0)  # the launch shape the counters were collected at
    v = b.frontend_valu(name, spl, 100000.0)
    assert v and 0.0 < v["frac"] < 1.0 and 0.3 < v["lk_share"] < 0.9
    lk = allp["k_lk_track_fb"]
    per_step = sum(e.get("SQ_INSTS_VALU", 0.0) * e.get("launches", 0.0) for k, e in allp.items() if k.startswith("k_") and k != "k_reproj_eval")
    assert abs(v["wave_instructions_per_frame"] - per_step / lk["launches"] / spl) < 1.0
    assert b.frontend_valu("no_such_summary.json", 8.0, 1.0) is None


def test_contract_line_is_compact_and_keeps_every_quoted_number():
    """The driver keeps the last 8 KB of bench.py's output: the contract line must hold the headline with its parity witness, roofline
    (incl. the exclusive-time ceiling), CPU baseline and the summary of every block, whatever the length of the per-group series."""
    b = _bench()
    committed = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles", "archive")) if f.startswith("r02_bench_driver_command"))
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", committed[-1])))  # a real (round-2) full record: ~15 KB
    full["parity"] = {"ok": True, "streams": [0, 383], "frames_per_stream": 73, "digest_gpu": ["0" * 16] * 2, "digest_oracle": ["0" * 16] * 2,
                      "checker": "x" * 120, "seconds": 3.2}
    full["roofline"].update({"exclusive_us": 61.0, "frac_exclusive": 0.04, "ceiling_frames_per_s": 120000.0, "exclusive_us_per_frame_all_kernels": 8.3,
                             "value_over_ceiling": 0.8, "achieved_exclusive": 330.0})
    full["host_ms_per_step"]["group_step_ms_per_group"] = [4.321] * 512
    full["step_stats"]["job_step_ms"]["series"] = [5.123] * 2000
    line = json.dumps(b.compact_line(full, os.path.join(ROOT, "gpurun_out", "bench_details.json")))
    assert len(line) < 7000, len(line)
    c = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "parity", "roofline", "cpu_baseline"):
        assert key in c, key
    assert c["parity"]["ok"] is True and c["roofline"]["ceiling_frames_per_s"] == 120000.0 and c["roofline"]["frac"] == full["roofline"]["frac"]
    assert c["solve"]["batched"]["value"] == full["solve"]["batched"]["value"]
    assert c["reproj"]["value"] == full["reproj"]["value"] and c["c4"]["frontend"]["value"] == full["c4"]["frontend"]["value"]
    assert c["pcie_inclusive"]["value"] == full["pcie_inclusive"]["value"] and c["details"] == "gpurun_out/bench_details.json"


def test_stale_counter_summaries_are_not_quoted(tmp_path, monkeypatch):
    """roofline.traffic / issue_frac / valu come from a committed rocprofv3 --pmc summary; they may only be quoted when that
    summary was collected on the kernel sources and launch shape of this build (profiles/summarize_pmc.py writes the provenance)."""
    b = _bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    src = tmp_path / "ic-gvins_amd" / "csrc"
    src.mkdir(parents=True)
    (src / "a.hip").write_text("kernel v1")
    sha = b.csrc_sha1()
    (prof / "r09_pmc_summary.json").write_text(json.dumps({"k_lk_track_fb": {}, "_meta": {"csrc_sha1": sha, "streams_per_launch": "32.0"}}))
    assert b.pmc_provenance("r09_pmc_summary.json", 32.0) is None
    assert "streams per launch" in b.pmc_provenance("r09_pmc_summary.json", 8.0)
    (src / "a.hip").write_text("kernel v2")
    assert "changed" in b.pmc_provenance("r09_pmc_summary.json", 32.0)
    (prof / "r08_pmc_summary.json").write_text(json.dumps({"k_lk_track_fb": {}}))
    assert "no provenance" in b.pmc_provenance("r08_pmc_summary.json", 32.0)
    # per-file record: only the files the quoted kernels are compiled from have to be the collected ones
    import hashlib
    (src / "a.hip").write_text("__global__ void k_a(int) {} v1")
    (src / "b.hip").write_text("__global__ void k_b(int) {} v1")
    (src / "common.h").write_text("// v1")
    (src / "ctx.hip").write_text("// ctx v1")
    per = {n: hashlib.sha1((src / n).read_bytes()).hexdigest() for n in ("a.hip", "b.hip", "common.h", "ctx.hip")}
    (prof / "r10_pmc_summary.json").write_text(json.dumps({"k_a": {}, "_meta": {"csrc_sha1": b.csrc_sha1(), "csrc_files": per, "streams_per_launch": "32.0"}}))
    assert b.kernel_source_files(["k_a"]) == ["a.hip", "common.h", "ctx.hip"]
    assert b.pmc_provenance("r10_pmc_summary.json", 32.0, ["k_a"]) is None
    (src / "b.hip").write_text("__global__ void k_b(int) {} v2")  # another kernel's file
    assert b.pmc_provenance("r10_pmc_summary.json", 32.0, ["k_a"]) is None
    assert "changed" in b.pmc_provenance("r10_pmc_summary.json", 32.0, ["k_a", "k_b"])
    assert "changed" in b.pmc_provenance("r10_pmc_summary.json", 32.0)  # (no kernel named: every file counts)
    assert "changed" in b.pmc_provenance("r10_pmc_summary.json", 32.0, ["k_missing"])
    (src / "common.h").write_text("// v2")  # a shared header ages every kernel
    assert "common.h" in b.pmc_provenance("r10_pmc_summary.json", 32.0, ["k_a"])


def test_whole_path_fraction_and_event_rates():
    """roofline.frac_whole_path = SURVEY 8(d)'s bytes per frame x frames/s of one GPU / peak (8.66 MB x 97.5 k / 8 TB/s
    = 0.106 for round 3's driver line); event rates per frame from the work counters."""
    b = _bench()
    assert abs(b.frame_bytes(1280, 720, 300) - 8.664e6) < 1e3            # C2
    assert abs(b.frame_bytes(640, 480, 100) - 2.888e6) < 1e3             # C1
    assert abs(b.frame_bytes(1920, 1080, 500) - 18.01e6) < 1e4           # C4
    assert abs(b.whole_path_fraction(1280, 720, 300, 97549.8, 8000.0) - 0.1056) < 5e-4
    r = b.event_rates({"lk_points": 58000, "lk_calls": 4, "detect_jobs": 150, "detect_calls": 4, "ransac_sets": 180, "ransac_calls": 9,
                       "frames": 200, "tri_points": 700}, keyframes=90, mappoints=640, frames=200)
    assert r == {"keyframes_per_frame": 0.45, "detections_per_frame": 0.75, "ransac_sets_per_frame": 0.9, "triangulated_points_per_frame": 3.5,
                 "mappoints_created_per_frame": 3.2, "lk_points_per_frame": 290.0}


def test_contract_line_carries_the_round4_honesty_fields():
    """frac on the exclusive duration with the under-load figure beside it, frac_whole_path, the forward-only control with both runs' event
    rates, and pcie_inclusive as a named configuration variant — all in the compact line, which still fits the driver's 8 KB tail."""
    b = _bench()
    committed = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles", "archive")) if f.startswith("r02_bench_driver_command"))
    full = json.load(open(os.path.join(ROOT, "profiles", "archive", committed[-1])))
    full["parity"] = {"ok": True}
    rates = {"keyframes_per_frame": 0.5, "detections_per_frame": 1.0, "ransac_sets_per_frame": 1.0, "triangulated_points_per_frame": 20.0,
             "mappoints_created_per_frame": 18.0, "lk_points_per_frame": 290.0}
    full["roofline"].update({"achieved": 437.6, "frac": 0.0547, "achieved_exclusive": 437.6, "frac_exclusive": 0.0547, "achieved_under_load": 141.6,
                             "frac_under_load": 0.0177, "frac_whole_path": 0.1071, "bytes_per_frame_algorithmic": 8664000, "exclusive_us": 373.7})
    full["rates"] = rates
    full["forward_control"] = {"value": 30000.0, "unit": "frames/s", "streams": 96, "groups": 12, "timed_steps": 40, "frames_per_stream": 84,
                               "rates": rates, "rates_pingpong_headline": rates, "tracking_state_fraction": 1.0, "note": "x" * 300}
    full["pcie_inclusive"].update({"config": {"workload": "C2", "streams_per_gpu": 384, "groups_per_gpu": 12, "input_residency": "pinned host"},
                                   "frac_whole_path": 0.055})
    c = b.compact_line(full, None)
    assert len(json.dumps(c)) < 8000
    assert c["roofline"]["frac"] == 0.0547 and c["roofline"]["frac_under_load"] == 0.0177 and c["roofline"]["frac_whole_path"] == 0.1071
    assert c["forward_control"]["rates"] == rates and c["rates"] == rates and "note" not in c["forward_control"]
    assert c["pcie_inclusive"]["config"]["input_residency"] == "pinned host"


def test_marg_batch_probe_child_process_schema():
    """bench.py's marg.batched block runs profiles/marg_batch_probe.py as a child process and reads these keys from its last line
    (here on the oracle-backed host layer: the CPU backend of the *_windows ABI)."""
    import os
    import subprocess
    import sys
    from stream_utils import ensure_oracle_host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ICG_PROBE_HOST_LIB=ensure_oracle_host())
    pr = subprocess.run([sys.executable, os.path.join(root, "profiles", "marg_batch_probe.py"), "--windows", "4"], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert pr.returncode == 0, pr.stderr[-500:]
    d = json.loads(pr.stdout.strip().splitlines()[-1])["4"]
    for k in ("windows_per_s", "batch_ms", "windows_per_s_one_by_one", "one_by_one_ms", "structured_dense", "max_rel_diff_Hp"):
        assert k in d
    assert d["structured_dense"] == [4, 0] and d["max_rel_diff_Hp"] < 1e-9
    # the bench's own wrapper around the child process (and what it does with a child that fails)
    b = _bench()
    os.environ["ICG_PROBE_HOST_LIB"] = env["ICG_PROBE_HOST_LIB"]
    try:
        blk = b.measure_marg_batched(4)
        assert blk.get("error") is None and blk["windows_per_batch"] == 4 and blk["value"] > 0 and blk["windows_structured_dense"] == [4, 0]
        committed = sorted(f for f in os.listdir(os.path.join(root, "profiles", "archive")) if f.startswith("r02_bench_driver_command"))
        full = json.load(open(os.path.join(root, "profiles", "archive", committed[-1])))  # a real full record
        full["parity"] = {"ok": True}
        full["marg"] = dict(full.get("marg") or {"value": 1.0, "unit": "ms per marginalization"}, batched=blk)
        assert b.compact_line(full, "details.json")["marg"]["batched"]["windows_per_batch"] == 4
        os.environ["ICG_PROBE_HOST_LIB"] = "/nonexistent/lib.so"
        assert "error" in b.measure_marg_batched(4)
    finally:
        os.environ.pop("ICG_PROBE_HOST_LIB", None)

"""bench.py's roofline object without a GPU: the arithmetic of price() / roofline() on synthetic launch records.

The contract (task statement (4)): `achieved` = algorithmic bytes per launch / average launch duration of
the kernel the timed region launches; `peak` 8000 GB/s; `frac` = achieved / peak; the counters' physical figures lead the
object; a note whenever SURVEY 8(d)'s bytes per step over ms_per_step pass the peak; PMC figures only where a committed
pass exists for the workload (profiles/traffic_r05.json) and still describes the kernels (its provenance block)."""
import hashlib
import importlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _tree_with_fresh_counters(tmp_path):
    """A copy of what bench.py looks at -- the kernel sources and a counter file whose provenance names exactly these
    sources (round 4's entries under round 5's provenance block) -- so that the arithmetic below does not depend on when
    the real passes were last collected."""
    bench = importlib.import_module("bench")
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "profiles"))
    os.makedirs(os.path.join(root, "prismarine-core_amd", "csrc"))
    sha = {}
    for rel in bench.KERNEL_SOURCES:
        shutil.copy(os.path.join(ROOT, rel), os.path.join(root, rel))
        sha[rel] = hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest()
    d = json.load(open(os.path.join(ROOT, "profiles", "traffic_r04.json")))
    d["provenance"] = {"sources_sha256": sha, "commit": "0123abc", "tags": ["test"]}
    json.dump(d, open(os.path.join(root, "profiles", "traffic_r05.json"), "w"))
    return bench, root


def test_roofline_object_arithmetic_and_key_order(tmp_path):
    bench, root = _tree_with_fresh_counters(tmp_path)
    R, V, T = 6_000_000, 300_000_000, 36_000_000
    launches, total_ms, steps = 9, 5.0, 1
    t = bench.price("rt_traverse<false, false, true>", launches, total_ms, R, V, T, 3, steps, "sponza_like", 1920, 1080, "synthetic", root=root)
    alg = R * 44 + V * 64 + T * 36
    assert abs(t["achieved"] - alg / (total_ms * 1e-3) / 1e9) < 1e-6
    assert abs(t["algorithmic_bytes_per_launch"] - alg / launches) < 1e-3
    assert abs(t["avg_launch_ms"] - total_ms / launches) < 1e-12 and t["launches_per_round"] == 3.0
    assert abs(t["own_record_bytes_per_launch"] - (R * 44 + V * 32 + T * 48) / launches) < 1e-3
    # the committed counters of the C3 workload are found for this kernel and give an HBM-side rate
    assert t["traffic"] and t["hbm_frac"] == t["hbm_gbs_of_one_launch"] / 8000.0 and 0.5 < t["l2_hit"] < 1.0
    whole = bench.price("rt_traverse<false, false, false>", 4, 3.0, R, V, T, 4, steps, "sponza_like", 1920, 1080, "synthetic", root=root)
    t["valu_issue_frac_of_step"] = 0.8      # (main() fills this from the step's instruction counts)
    roof = bench.roofline(t, whole, alg / steps, 2.5, 5200.0, root=root)
    keys = list(roof)
    # what binds the path comes first -- the VALU issue slots, not HBM -- then where the replayed counters come from
    assert keys[:5] == ["binding", "traffic_commit", "traffic_stale", "traffic_stale_why", "traffic_file"]
    assert roof["binding"] == {"roof": "valu_issue", "frac": 0.8, "lane_utilisation": t["valu_lane_utilisation"], "hbm_frac": t["hbm_frac"]}
    assert roof["traffic_commit"] == "0123abc" and roof["traffic_stale"] is False and roof["traffic_file"] == "profiles/traffic_r05.json"
    keys = keys[5:]
    assert keys[:5] == ["bound", "hbm_frac", "valu_issue_frac_of_step", "valu_issue_utilisation", "timed_schedule_valu_lane_utilisation"]   # physical figures first
    assert keys[5:11] == ["achieved", "peak", "unit", "frac", "traffic", "frac_note"]
    assert roof["peak"] == 8000.0 and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-12
    assert "not an HBM utilisation" in roof["frac_note"]              # 21.9 GB per 2.5 ms step > 8 TB/s
    assert roof["bound"].startswith("cache/VALU") and roof["kernel"] == "rt_traverse<false, false, true>"
    assert roof["single_launch_kernel_alone"]["kernel"] == "rt_traverse<false, false, false>"
    json.dumps(roof)
    # no committed pass for a workload -> no counters quoted, and none for tiles / split frames at all
    other = bench.price("rt_traverse<false, false, true>", launches, total_ms, R, V, T, 3, steps, "cornell", 640, 480, "synthetic", root=root)
    tile = bench.price("rt_traverse<false, false, true>", launches, total_ms, R, V, T, 3, steps, "sponza_like", 1920, 1080, "synthetic", use_pmc=False, root=root)
    for o in (other, tile):
        assert "traffic" not in o and "hbm_frac" not in o
    r2 = bench.roofline(other, whole, 1e9, 2.5, 5200.0, root=root)
    assert r2["bound"].startswith("unknown") and r2["hbm_frac"] is None and "frac_note" not in r2 and r2["traffic"] is None


def test_replayed_counters_are_flagged_stale_once_a_kernel_source_changes(tmp_path):
    """the PMC-derived fields of a bench line are replayed from a committed file. The file now names the
    kernel sources it was collected with; a traversal kernel edited since -- here: one byte appended to trace.hip in a copy of
    the tree -- turns the replay off (traffic_stale, no hbm_frac / valu_* / traffic) instead of quoting another kernel's counters."""
    bench, root = _tree_with_fresh_counters(tmp_path)
    R, V, T = 6_000_000, 300_000_000, 36_000_000
    args = ("rt_traverse<false, false, true>", 9, 5.0, R, V, T, 3, 1, "sponza_like", 1920, 1080, "synthetic")
    assert bench.traffic_provenance(root)[2] is False and "traffic" in bench.price(*args, root=root)
    with open(os.path.join(root, "prismarine-core_amd", "csrc", "trace.hip"), "a") as f:
        f.write(" ")
    path, prov, stale, why = bench.traffic_provenance(root)
    assert stale is True and "trace.hip has changed" in why and prov["commit"] == "0123abc"
    t = bench.price(*args, root=root)
    assert not any(k in t for k in ("traffic", "hbm_frac", "valu_lane_utilisation", "valu_insts_per_launch", "l2_hit"))
    roof = bench.roofline(t, t, 1e9, 2.5, 5200.0, root=root)
    assert roof["traffic_stale"] is True and roof["traffic_commit"] == "0123abc" and roof["binding"]["frac"] is None
    assert roof["hbm_frac"] is None and roof["bound"].startswith("unknown")
    # a counter file from before round 5 carries no hashes at all: stale by definition
    os.remove(os.path.join(root, "profiles", "traffic_r05.json"))
    shutil.copy(os.path.join(ROOT, "profiles", "traffic_r04.json"), os.path.join(root, "profiles", "traffic_r04.json"))
    assert bench.traffic_provenance(root)[2] is True and "no source hashes" in bench.traffic_provenance(root)[3]


def test_committed_traffic_file_matches_its_profiles():
    """profiles/traffic_r05.json (what bench.py quotes) holds the traversal kernels of C3 and C5 with the fields the roofline
    uses, its in-flight launch average agrees with the committed kernel-trace summary of the same run, and it names the kernel
    sources its passes ran with -- which are the sources of this tree (the counters were collected after the last kernel change)."""
    bench = importlib.import_module("bench")
    d = json.load(open(os.path.join(ROOT, "profiles", "traffic_r05.json")))
    assert set(d["provenance"]["sources_sha256"]) == set(bench.KERNEL_SOURCES) and d["provenance"]["commit"]
    path, prov, stale, why = bench.traffic_provenance()
    assert os.path.basename(path) == "traffic_r05.json" and not stale, why
    ents = {(e["scene"], e["kernel"]): e for e in d["entries"]}
    for scene in ("sponza_like", "stress"):
        for k in ("psm::rt_traverse<false, false, true>", "psm::rt_traverse<false, false, false>"):
            e = ents[(scene, k)]
            assert e["traffic_bytes_per_launch"] == 2.0 * e["fetch_bytes_per_launch_raw"] + e["write_bytes_per_launch"]
            assert 0.3 < e["valu_lane_utilisation"] < 0.7 and e["alone_avg_us"] > 0 and e["sq_per_launch"]["SQ_INSTS_VALU"] > 0
    tag = {"sponza_like": "c3", "stress": "c5"}
    for scene, c in tag.items():
        rows = [l.split() for l in open(os.path.join(ROOT, "profiles", "r05_%s_kt_stats.txt" % c)) if "false, false, tru" in l]
        avg_us = float(rows[0][-2])
        assert abs(avg_us - ents[(scene, "psm::rt_traverse<false, false, true>")]["in_flight_avg_us"]) < 0.01 * avg_us

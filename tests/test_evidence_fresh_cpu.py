"""Staleness guard of the round's committed evidence: profiles/r06_* must describe the code in the
tree.  tools/run_r06_prof.sh stamps its pass with the sha256 of the kernel sources and of the library
(profiles/r06_evidence.json, tools/evidence_stamp.py); this test recomputes the source hash from the working tree, so
a kernel edit after the evidence pass fails the CPU suite until the pass is run again.  Also: every kernel the stats
CSV names must be a symbol of the library, the line and the counter passes must come from ONE binary, and bench.py's
file lists must resolve to this round's files."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
PROF = os.path.join(ROOT, "profiles")
LIB = os.path.join(ROOT, "stark-perpetual_amd", "lib", "libstarkperp.so")


def _load(name):
    path = os.path.join(PROF, name)
    if not os.path.exists(path):
        pytest.skip("%s not collected yet this round (tools/run_r06_prof.sh)" % name)
    return json.load(open(path))


def test_evidence_was_collected_on_the_sources_in_the_tree():
    ev = _load("r06_evidence.json")
    import evidence_stamp
    assert ev["csrc_sha256"] == evidence_stamp.source_hash(), (
        "csrc/ changed after the evidence pass: run tools/run_r06_prof.sh again and copy gpurun_out/r06prof into profiles/")


def test_line_and_counter_passes_come_from_one_binary():
    ev = _load("r06_evidence.json")
    line = _load("r06_bench_default.json")
    detail = _load("r06_bench_detail.json")
    assert line["build"]["lib_sha256_16"] == ev["lib_sha256"][:16]
    assert detail["build"]["lib_sha256"] == ev["lib_sha256"]
    assert ev["lib_sha256"][:16] in _load("r06_valu_issue.json")["_source"]
    # the PMC traffic file was collected on the configuration the line reports
    from benchlib import roofline
    key = roofline.merkle_config_key(line["steps"], detail["config"]["timed_calls"], line["config"]["streams"],
                                     line["config"]["window_bits"])
    assert _load("r06_pmc_traffic.json")["config_key"] == key
    assert line["roofline"]["traffic_same_configuration"] is True
    assert sorted(line["roofline"]["sources"]) == ["profiles/r06_pmc_traffic.json", "profiles/r06_valu_issue.json"]
    assert len(json.dumps(line)) < 8192


def test_every_kernel_of_the_stats_file_is_a_symbol_of_the_library():
    path = os.path.join(PROF, "r06_kernel_stats.csv")
    if not os.path.exists(path):
        pytest.skip("r06_kernel_stats.csv not collected yet this round")
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    syms = subprocess.run(["nm", "-C", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    names = [r["Name"] for r in csv.DictReader(open(path))]
    ours = [n for n in names if "sp::" in n.split("(")[0] or n.split("(")[0].startswith("tree_")]
    assert len(ours) >= 12
    for n in ours:
        base = n.split("(")[0].replace("void ", "")
        assert base + "(" in syms, "kernel in the stats file that the library does not have: " + base
    flat = " ".join(ours)
    for must in ("sp::ped_accumulate_kernel", "sp::ped_finish_lds_kernel", "sp::ped_top_kernel", "sp::ntt_tile_kernel",
                 "sp::ecdsa_verify_keyed_kernel", "sp::ped_path_kernel"):
        assert must in flat, must + " missing from r06_kernel_stats.csv"


def test_bench_reads_this_rounds_files_first():
    _load("r06_evidence.json")
    from benchlib import roofline
    assert roofline.valu_counts(26)[2] == "r06_valu_issue.json"
    assert roofline.pmc_traffic("sp::ped_accumulate_kernel", "x")["source"] == "profiles/r06_pmc_traffic.json"
    a = roofline.pmc_traffic("sp::ped_accumulate_kernel", roofline.airfri_config_key(26), roofline.AIRFRI_PMC_FILES)
    assert a["source"] == "profiles/r06_pmc_traffic_airfri.json" and a["same_configuration_as_this_run"] is True
    c3 = roofline.c3_roofline({})
    assert c3 and c3["instr_source"] == "profiles/r06_c3_sq_counters.json" and c3["verify_keyed"]["instr_per_item"] > 5e4
    f = roofline.forest_instr_per_hash(26)
    assert f and f[1] == "r06_valu_issue.json" and 27000 < f[0] < 36000

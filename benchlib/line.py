"""The ONE JSON line bench.py prints, and the detail file beside it.

Round 5's line had grown to 20 KB and the driver could no longer parse it (`BENCH_r05.json: parsed null`).  The
rule now: bench.py assembles everything it measured into a DETAIL dict (the old line: telemetry windows, burst,
region percentiles, per-phase tables, provenance prose ...), `main_line()` derives from it the line proper - the
contract's keys, a numbers-only `roofline`, `cpu_baseline`, a small `airfri`, `summary` - and `emit()` REFUSES to
print more than MAX_LINE_BYTES: the detail goes to bench_detail.json beside bench.py and to stderr.
tests/test_bench_helpers_cpu.py::test_line_is_small_and_complete builds the line from a committed full-size detail
and holds it to the same bound."""
import json
import math
import os
import sys

from .common import DTYPE, ROOT

MAX_LINE_BYTES = 8192
DETAIL_FILE = "bench_detail.json"
REQUIRED_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def g(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _num(v, digits=9):
    """Numbers rounded to `digits` significant figures (the line is for reading and parsing, the detail file keeps
    full precision); everything else unchanged."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        if not math.isfinite(v):
            return None
        if v == 0.0:
            return 0.0
        return float("%.*g" % (digits, v))
    return v


def _clean(obj, digits=9):
    if isinstance(obj, dict):
        return {k: _clean(v, digits) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_clean(v, digits) for v in obj]
    return _num(obj, digits)


def _ms(v):
    return None if v is None else 1e3 * v


def _files(*names):
    return sorted({n for n in names if n})


def slim_roofline(roof, algo_bytes_per_launch=None):
    """Numbers and file names only."""
    if not roof:
        return None
    td = roof.get("traffic_detail") or {}
    out = {
        "bound": roof.get("bound"), "kernel": (roof.get("kernel") or "").split(" ")[0] or None,
        "instr_per_hash": roof.get("instr_per_hash"),
        "achieved": roof.get("achieved"), "peak": roof.get("peak"), "unit": roof.get("unit"), "frac": roof.get("frac"),
        "frac_at_held_clock": roof.get("frac_at_held_clock"), "held_clock_mhz": roof.get("held_clock_mhz"),
        "frac_at_2_cycle_peak": roof.get("frac_at_2_cycle_peak"),
        "cycles_per_instr_of_the_mix": roof.get("cycles_per_instr_of_the_mix"),
        "launches": roof.get("launches"), "hashes_per_launch": roof.get("hashes_per_launch"),
        "avg_launch_us": roof.get("avg_launch_us"),
        "traffic": roof.get("traffic"), "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE + WRITE_SIZE)",
        "traffic_same_configuration": td.get("same_configuration_as_this_run"),
        "algorithmic_bytes_per_launch": algo_bytes_per_launch,
        "hbm": {k: g(roof, "hbm", k) for k in ("bound", "achieved", "peak", "unit", "frac")} if roof.get("hbm") else None,
        "sources": _files(roof.get("instr_source"), td.get("source")),
    }
    if out["hbm"] is not None and out["hbm"].get("bound") is None:
        out["hbm"]["bound"] = "hbm"
    wr = roof.get("whole_region")
    if wr:
        out["whole_region"] = {"frac": wr.get("frac"), "frac_at_held_clock": wr.get("frac_at_held_clock"),
                               "instr_per_hash": wr.get("instr_per_hash")}
    return out


def slim_cpu_baseline(cb):
    if not cb:
        return cb
    if "error" in cb:
        return {"error": str(cb["error"])[:160]}
    out = {k: cb.get(k) for k in ("value", "unit", "cores", "kind") if k in cb}
    out["sample"] = str(cb.get("sample", ""))[:200]
    for k in ("matches_gpu", "root_matches_gpu"):
        if k in cb:
            out[k] = cb[k]
    return out


def slim_airfri(a):
    """<= 1 KB: both rates, the hash roofline of the job, the CPU leg scaled to 2^20 rows."""
    if not a:
        return a
    if "error" in a:
        return {"error": str(a["error"])[:160]}
    out = {"workload": "2^20-row trace: LDE x4, AIR, 16 FRI folds, 17 commits (BASELINE.json configs[3])",
           "commits_per_sec": a.get("commits_per_sec"), "commits_per_sec_burst": a.get("commits_per_sec_burst"),
           "seconds_per_job_one_stream": a.get("seconds_per_job_one_stream"),
           "pedersen_hashes_per_job": a.get("pedersen_hashes_per_job"),
           "timed_jobs": g(a, "timed", "jobs"), "timed_s": g(a, "timed", "seconds"),
           "held_clock_mhz": g(a, "timed", "telemetry", "sclk_mhz_median"),
           "roofline": {"bound": g(a, "roofline", "bound"), "kernel": "ped_accumulate_kernel",
                        "frac": g(a, "roofline", "frac"), "frac_at_held_clock": g(a, "roofline", "frac_at_held_clock"),
                        "avg_launch_us": g(a, "roofline", "avg_launch_us"), "traffic": g(a, "roofline", "traffic"),
                        "hbm_frac": g(a, "roofline", "hbm", "frac")}}
    ph = a.get("phases")
    if ph:
        out["phase_ms"] = {k.split("_2p")[0]: _ms(v.get("seconds")) for k, v in ph.items()}
        out["phase_hbm_frac"] = {k.split("_2p")[0]: v.get("hbm_frac_of_8_tb_per_s") for k, v in ph.items()}
    cb = a.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"commits_per_sec_scaled_to_2p20": g(cb, "scaled_to_2p20_rows", "commits_per_sec"),
                               "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": "one 2^10-row job in %.1f s, x 1024" % cb["seconds"] if cb.get("seconds") else None}
    for k in ("n_gpus", "scaling", "commits_per_sec_slowest_gpu"):
        if k in a:
            out[k] = a[k]
    return out


def slim_dist(d):
    """What proves that RCCL saw N ranks stays in the main line."""
    if not d:
        return d
    out = {k: d.get(k) for k in ("backend", "world_size", "rccl_version", "forced_at_one_gpu", "visible_devices",
                                 "window_plan_fallback") if k in d}
    prv = d.get("per_rank_value")
    if prv:
        out["per_rank_value"] = {"min": prv.get("min"), "max": prv.get("max")}
    ranks = [r for r in d.get("ranks", []) if isinstance(r, dict)]
    out["ranks_reported"] = len(ranks)
    out["devices"] = sorted({r.get("pci") for r in ranks if r.get("pci")})[:16]
    pa = d.get("peer_access")
    if isinstance(pa, list):
        out["peer_access_all"] = all(all(v in (1, None) for v in row) for row in pa)
    return out


def summary_object(result):
    """Compact digest, the LAST key of the line: both halves of BASELINE.json's metric and the figures the verdicts ask
    about, numbers only."""
    np_c3 = g(result, "extra", "c3_4096_orders_numpy_entry_points_seconds", "total")
    one = g(result, "extra", "c3_4096_orders_one_call_seconds") or {}
    return {
        "pedersen_hashes_per_sec": result.get("value"),
        "ms_per_step": result.get("ms_per_step"),
        "roofline_frac_bulk_launches": g(result, "roofline", "frac"),
        "roofline_frac_whole_region": g(result, "roofline", "whole_region", "frac"),
        "roofline_frac_at_held_clock": g(result, "roofline", "frac_at_held_clock"),
        "sclk_mhz_median": g(result, "telemetry", "sclk_mhz_median"),
        "power_w_median": g(result, "telemetry", "power_w_median"),
        "timed_total_s": g(result, "timed_regions", "total_s"),
        "burst_pedersen_hashes_per_sec": g(result, "burst", "value"),
        "sustained_over_burst": result.get("sustained_over_burst"),
        "lib_sha256_16": (g(result, "build", "lib_sha256") or "")[:16],
        "airfri_commits_per_sec": g(result, "airfri", "commits_per_sec"),
        "airfri_seconds_per_job": g(result, "airfri", "seconds_per_job_one_stream"),
        "airfri_roofline_frac": g(result, "airfri", "roofline", "frac"),
        "airfri_cpu_baseline_commits_per_sec": g(result, "airfri", "cpu_baseline", "scaled_to_2p20_rows", "commits_per_sec"),
        "single_tree_ms": g(result, "extra", "single_tree_rebuild_ms_one_stream"),
        "bulk_pedersen_hashes_per_sec": g(result, "extra", "bulk_pedersen_hashes_per_sec"),
        "c3_total_ms": _ms(np_c3),
        "c3_one_call_ms": _ms(one.get("median", one.get("best_of_3"))),
        "c3_one_call_ms_p90": _ms(one.get("p90")),
        "c3_one_call_ms_min": _ms(one.get("min", one.get("best_of_3"))),
        "c3_tree_update_ms": _ms(g(result, "extra", "c3_4096_orders_numpy_entry_points_seconds",
                                   "orders_tree_height64_update_on_existing_state")),
        "c3_verify_frac": g(result, "extra", "c3", "roofline", "verify_keyed", "frac"),
        "c3_verify_ladder_frac": g(result, "extra", "c3", "roofline", "verify_ladder", "frac"),
        "verify_keyed_frac_2p18": g(result, "extra", "c3", "roofline", "verify_keyed_2p18", "frac"),
        "verify_ladder_frac_2p18": g(result, "extra", "c3", "roofline", "verify_ladder_2p18", "frac"),
        "ecdsa_verifies_per_sec_ladder": g(result, "extra", "ecdsa_verifies_per_sec_x_only_2p16"),
        "ecdsa_verifies_per_sec_key_tables": g(result, "extra", "ecdsa_verifies_per_sec_key_tables_2p16"),
        "ecdsa_verifies_per_sec_ladder_2p18": g(result, "extra", "ecdsa_verifies_per_sec_x_only_2p18"),
        "ecdsa_verifies_per_sec_key_tables_2p18": g(result, "extra", "ecdsa_verifies_per_sec_key_tables_2p18"),
        "ecdsa_signs_per_sec": g(result, "extra", "ecdsa_signs_per_sec_2p16"),
        "ecdsa_signs_per_sec_list_api_host_inclusive": g(result, "extra", "ecdsa_signs_per_sec_2p16_host_inclusive"),
        "cpu_hashes_per_sec_python_port": g(result, "cpu_baseline", "value"),
        "cpu_hashes_per_sec_c_port": g(result, "cpu_baseline_c", "value"),
        "cpu_hashes_per_sec_optimised": g(result, "cpu_baseline_opt", "value"),
        "cpu_sign_per_sec_all_cores": g(result, "cpu_baseline_ecdsa", "sign_per_sec_all_cores"),
        "cpu_verify_per_sec_all_cores": g(result, "cpu_baseline_ecdsa", "verify_per_sec_all_cores"),
        "cpu_cores": g(result, "cpu_baseline", "cores"),
        "parity_in_run": {"level1_matches_gpu": g(result, "cpu_baseline", "matches_gpu"),
                          "root_matches_c_oracle": g(result, "cpu_baseline_c", "root_matches_gpu"),
                          "root_matches_optimised_cpu": g(result, "cpu_baseline_opt", "root_matches_gpu"),
                          "sign_matches_gpu": g(result, "cpu_baseline_ecdsa", "sign_matches_gpu"),
                          "verify_matches_gpu": g(result, "cpu_baseline_ecdsa", "verify_matches_gpu")},
    }


def main_line(detail):
    """The line proper from the full result of a merkle-workload run (bench.py main)."""
    cfg = detail.get("config") or {}
    tr = detail.get("timed_regions") or {}
    hpl = g(detail, "roofline", "hashes_per_launch")
    abh = g(detail, "roofline", "hbm", "algorithmic_bytes_per_hash")
    line = {k: detail.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                       "higher_is_better", "scaling", "vs_baseline")}
    line["dtype"] = DTYPE
    line["data"] = detail.get("data", "synthetic")
    line["config"] = {k: cfg.get(k) for k in ("workload", "tree_height", "leaves_per_gpu", "hashes_per_step",
                                              "trees_in_timed_call", "calls_per_region", "streams", "window_bits",
                                              "combine") if k in cfg}
    if cfg.get("table_mib") is not None:
        line["config"]["table_gib"] = cfg["table_mib"] / 1024.0
    line["timed"] = {"regions": tr.get("count"), "total_s": tr.get("total_s"), "median_s": tr.get("median_s"),
                     "p10_s": tr.get("p10_s"), "p90_s": tr.get("p90_s"), "preheat_s": tr.get("preheat_s"),
                     "value_is": "median region of the sustained window"}
    line["roofline"] = slim_roofline(detail.get("roofline"),
                                     None if hpl is None or abh is None else int(hpl * abh))
    line["cpu_baseline"] = slim_cpu_baseline(detail.get("cpu_baseline"))
    if "airfri" in detail:
        line["airfri"] = slim_airfri(detail["airfri"])
    line["telemetry"] = {k: g(detail, "telemetry", k) for k in ("sclk_mhz_median", "power_w_median", "power_cap_w")} \
        if detail.get("telemetry") else None
    if "dist" in detail:
        line["dist"] = slim_dist(detail["dist"])
        line["combine_matches_recomputed"] = detail.get("combine_matches_recomputed")
    if detail.get("failed_legs"):
        line["failed_legs"] = detail["failed_legs"]
    line["build"] = {"lib_sha256_16": (g(detail, "build", "lib_sha256") or "")[:16],
                     "bench_py_sha16": g(detail, "build", "bench_py_sha16")}
    line["detail"] = DETAIL_FILE
    line["summary"] = summary_object(detail)  # LAST key
    return _clean(line)


def check_line(line, roofline_required=True):
    """Raises ValueError unless `line` is a complete, strictly valid JSON object of at most MAX_LINE_BYTES."""
    text = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    if len(text.encode()) > MAX_LINE_BYTES:
        raise ValueError("bench line is %d bytes (limit %d)" % (len(text.encode()), MAX_LINE_BYTES))
    if "\n" in text:
        raise ValueError("bench line contains a newline")
    back = json.loads(text)
    missing = [k for k in REQUIRED_KEYS if k not in back]
    if missing:
        raise ValueError("bench line lacks %s" % missing)
    if roofline_required and isinstance(back.get("roofline"), dict):
        miss = [k for k in ROOFLINE_KEYS if k not in back["roofline"]]
        if miss:
            raise ValueError("roofline lacks %s" % miss)
    return text


def write_detail(detail, path=None):
    """Full-precision detail beside bench.py, or where STARKPERP_BENCH_DETAIL points (never raises: a read-only tree
    loses the file, not the measurement)."""
    path = path or os.environ.get("STARKPERP_BENCH_DETAIL") or os.path.join(ROOT, DETAIL_FILE)
    try:
        with open(path, "w") as f:
            json.dump(detail, f, indent=1, default=str)
            f.write("\n")
        return path
    except OSError as e:
        sys.stderr.write("bench: could not write %s (%s)\n" % (path, e))
        return None


def _truncate(obj, max_str=160, max_list=16):
    if isinstance(obj, dict):
        return {k: _truncate(v, max_str, max_list) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_truncate(v, max_str, max_list) for v in obj[:max_list]]
    if isinstance(obj, str) and len(obj) > max_str:
        return obj[:max_str - 3] + "..."
    return obj


def fallback_line(line):
    """What is printed when a line breaks the bound: the contract's keys (+ summary) with every string cut to 160
    characters and every list to 16 entries; if that is still too long, config shrinks to its workload and the
    nested objects to their scalar fields."""
    keep = _truncate({k: line.get(k) for k in REQUIRED_KEYS + ("summary",) if k in line})
    if len(json.dumps(keep).encode()) > MAX_LINE_BYTES:
        for k in ("config", "roofline", "cpu_baseline", "summary"):
            if isinstance(keep.get(k), dict):
                keep[k] = {kk: vv for kk, vv in keep[k].items() if not isinstance(vv, (dict, list))}
        if isinstance(keep.get("config"), dict):
            keep["config"] = {"workload": keep["config"].get("workload")}
    return keep


def emit(line, detail=None, stream=None):
    """Print ONE line on stdout (checked against the size bound first), the detail to its file and to stderr.  A line
    that fails the check is cut down (fallback_line) rather than not printed at all."""
    stream = stream or sys.stdout
    line = _clean(line)
    try:
        text = check_line(line, roofline_required=False)
    except ValueError as e:
        sys.stderr.write("bench: %s; printing the contract's keys only\n" % e)
        text = json.dumps(_clean(fallback_line(line)), allow_nan=False, separators=(", ", ": "))
    if detail is not None:
        write_detail(detail)
        sys.stderr.write("bench detail: " + json.dumps(detail, default=str) + "\n")
    sys.stderr.flush()
    stream.write(text + "\n")
    stream.flush()
    return text

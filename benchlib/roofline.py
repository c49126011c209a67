"""Roofline arithmetic of the hash kernels: VALU issue (the roof that binds them), HBM (the one the contract names),
PMC traffic from the committed rocprofv3 passes.  Every figure that comes from a file under profiles/ carries the file's
name, and a file is only called "the same configuration as this run" when its recorded config_key equals this run's."""
import json
import os

from .common import ALGO_BYTES_PER_HASH, HBM_PEAK_GBS, ROOT

VALU_PEAK_SIMDS, VALU_NOMINAL_GHZ = 1024, 2.4
# Committed counter files, newest round first (the end-of-round pass of THIS round comes first, and
# nothing older is stamped as this run's configuration unless its config_key says so).
PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json",
             "r02_pmc_traffic.json", "r01_pmc_traffic.json")
AIRFRI_PMC_FILES = ("r06_pmc_traffic_airfri.json", "r04_pmc_traffic_airfri.json", "r03_pmc_traffic_airfri.json",
                    "r02_pmc_traffic_airfri.json")
C3_COUNTER_FILES = ("r06_c3_sq_counters.json",)
VALU_ISSUE_FILES = ("r06_valu_issue.json", "r04_valu_issue.json", "r03_valu_issue.json", "r02_valu_issue.json",
                    "r01_valu_issue.json")
# Where the prose about the peak lives (it was 1.3 KB of every line in round 5)
PEAK_BASIS_DOC = "DESIGN.md section 4.2 (per-opcode issue intervals: profiles/r04_valu_rate_ubench.txt, tools/valu_mix.py)"


def _load(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:  # noqa: BLE001 - a missing or broken profile file is "no measurement", never a crash
        return None


def _valu_issue_file():
    for name in VALU_ISSUE_FILES:
        m = _load(name)
        if m is not None:
            return m, name
    return {}, None


def valu_cycles_per_instr():
    """Issue interval of the bulk hash kernel's instruction MIX in shader cycles per wave64 instruction per SIMD:
    every opcode of the kernel priced at the best interval tools/ubench/valu_rate.hip measured for it at any
    occupancy (profiles/r04_valu_rate_ubench.txt), weighted by the kernel's static histogram (tools/valu_mix.py).
    4.04 for ped_accumulate_kernel; rounds 1 - 3 used a flat 4."""
    m, _ = _valu_issue_file()
    return float(m.get("cycles_per_wave64_valu_instr", 4.0))


def valu_counts(window_bits):
    """(accumulate, finish) SQ_INSTS_VALU per hash of the bulk kernels and the file they come from (rocprofv3 --pmc,
    profiles/r0N_valu_issue.json, newest first); None when no file has this window width."""
    for name in VALU_ISSUE_FILES:
        m = _load(name)
        try:
            w = m["window_bits"][str(window_bits)]
            return w["accumulate_instr_per_hash"], w["finish_instr_per_hash"], name
        except Exception:  # noqa: BLE001
            continue
    return None


def valu_peak(clock_mhz=None):
    """wave64 VALU instructions per second the chip can issue with this kernel's mix, at the nominal clock or at
    `clock_mhz` (the clock the package held while it was measured)."""
    hz = VALU_NOMINAL_GHZ * 1e9 if clock_mhz is None else clock_mhz * 1e6
    return VALU_PEAK_SIMDS * hz / valu_cycles_per_instr()


def forest_instr_per_hash(window_bits):
    """VALU instructions per hash over EVERY kernel of the 20-tree forest build (accumulate, finish-lds, split, quad and
    top kernels; profiles/r06_valu_issue.json `forest_20`, measured at 26-bit windows) and its file, or None."""
    if window_bits != 26:
        return None
    for name in VALU_ISSUE_FILES:
        m = _load(name)
        try:
            return float(m["forest_20"]["instr_per_hash_all_kernels"]), name
        except Exception:  # noqa: BLE001
            continue
    return None


def valu_issue(hashes_per_sec, window_bits, workload, include_finish=True, whole_forest=False):
    """The roofline that binds the hash kernels: wave64 VALU instructions issued per
    second against the chip's issue peak.  whole_forest: price a hash at what the whole 20-tree build issues per hash
    (every kernel, measured) instead of accumulate + finish of the bulk batch."""
    c = valu_counts(window_bits)
    if c is None:
        return None
    per_hash = c[0] + (c[1] if include_finish else 0)
    if whole_forest:
        f = forest_instr_per_hash(window_bits)
        if f is not None:
            per_hash, c = f[0], (c[0], c[1], f[1])
    achieved = hashes_per_sec * per_hash / 64.0
    peak = valu_peak()
    return {"bound": "valu_issue", "workload": workload, "instr_per_hash": per_hash,
            "instr_source": "profiles/" + c[2], "achieved": achieved, "peak": peak,
            "unit": "wave64 VALU instr/s", "frac": achieved / peak,
            "cycles_per_instr_of_the_mix": valu_cycles_per_instr(),
            "frac_at_2_cycle_peak": achieved / (VALU_PEAK_SIMDS * VALU_NOMINAL_GHZ * 1e9 / 2.0),
            "frac_at_flat_4_cycle_peak": achieved / (VALU_PEAK_SIMDS * VALU_NOMINAL_GHZ * 1e9 / 4.0)}


def add_held_clock(roof, held_mhz):
    """The same issue rate against the clock the chip HELD while it was measured (the package sits at its power limit
    under the hash kernels): what the kernel reaches of the attainable issue rate."""
    if roof and held_mhz and roof.get("achieved"):
        roof["held_clock_mhz"] = held_mhz
        roof["frac_at_held_clock"] = roof["achieved"] / valu_peak(held_mhz)
    return roof


def hbm_object(hashes_per_sec):
    gbs = ALGO_BYTES_PER_HASH * hashes_per_sec / 1e9
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_hash": ALGO_BYTES_PER_HASH}


def pmc_traffic(kernel, this_config, files=PMC_FILES):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE
    and --pmc WRITE_SIZE runs, tools/pmc_traffic.py; units and gfx950 calibration in its docstring), and
    the configuration those passes ran - traffic is only comparable with this run when they agree."""
    for name in files:
        m = _load(name)
        try:
            k = m["kernels"][kernel]
            cfg = m.get("config", "bench.py r01 default: --steps 128 --warmup 16, 64 trees per call, 2 streams, "
                                  "26-bit windows (NOT this run's configuration)")
            return {"bytes_per_launch": k.get("hbm_bytes_per_launch_fetch_doubled", k["hbm_bytes_per_launch"]),
                    "fetch_size_doubled_for_coalesced_reads": "hbm_bytes_per_launch_fetch_doubled" in k,
                    "fetch_bytes_per_launch": k["fetch_bytes_per_launch"],
                    "write_bytes_per_launch": k["write_bytes_per_launch"], "launches_profiled": k["launches"],
                    "source": "profiles/" + name, "collected_on": cfg, "config_key": m.get("config_key"),
                    "same_configuration_as_this_run": bool(m.get("config_key")) and m.get("config_key") == this_config}
        except Exception:  # noqa: BLE001
            continue
    return None


def merkle_config_key(steps, timed_plan, n_streams, wbits):
    return "merkle:steps=%d:calls=%s:streams=%d:w=%d" % (steps, ",".join(map(str, timed_plan)), n_streams, wbits)


def airfri_config_key(wbits):
    """2^20-row jobs alternating over three streams (the `airfri` object and `--workload airfri`); the files of rounds
    2 - 4 carry the bare key "airfri" and so never compare equal."""
    return "airfri:rows=2^20:streams=3:w=%d" % wbits


def c3_roofline(live_rates):
    """Roofline of the kernels BASELINE.json configs[2] is made of: instructions per item from the
    committed SQ_INSTS_VALU pass (profiles/r06_c3_sq_counters.json, tools/c3_counters.py), the rate LIVE where
    bench.py measures one (`live_rates`: {"verify_ladder": signatures/s at 2^16, "verify_keyed": ...}) and from the
    counter pass's own launch otherwise.  achieved = wave64 VALU instructions per second; frac against the same issue
    peak as the hash kernels (all of them are fe_mul-dominated: the same instruction mix)."""
    m = None
    for name in C3_COUNTER_FILES:
        m = _load(name)
        if m is not None:
            src = "profiles/" + name
            break
    if m is None:
        return None
    peak = valu_peak()
    out = {"unit": "wave64 VALU instr/s", "peak": peak, "cycles_per_instr_of_the_mix": valu_cycles_per_instr(),
           "instr_source": src}

    def entry(kernel, grid):
        return ((m["kernels"].get(kernel) or {}).get("by_grid") or {}).get(str(grid))

    for key, kernel, live in (("verify_ladder", "sp::ecdsa_verify_kernel", live_rates.get("verify_ladder")),
                              ("verify_keyed", "sp::ecdsa_verify_keyed_kernel", live_rates.get("verify_keyed"))):
        e = entry(kernel, 1 << 16)
        if not e:
            continue
        ips = e["instr_per_signature"]
        rate = live if live else (1 << 16) / (e["duration_us_under_pmc"] * 1e-6)
        ach = ips * rate / 64.0
        out[key] = {"kernel": kernel, "items": 1 << 16, "instr_per_item": ips, "items_per_sec": rate,
                    "rate_is": "live (this run)" if live else "the counter pass's launch",
                    "achieved": ach, "frac": ach / peak, "waves_per_simd": e["waves_per_simd"],
                    "registers": {k: (m["kernels"][kernel].get(k)) for k in ("vgpr", "agpr", "scratch_bytes_per_lane")}}
    # the same kernels at 2^18 signatures (two waves per SIMD resident since round 6): live rates only
    for key, base, n_waves in (("verify_ladder_2p18", "verify_ladder", 2.0), ("verify_keyed_2p18", "verify_keyed", 2.0)):
        rate = live_rates.get(key)
        if rate and base in out:
            ach = out[base]["instr_per_item"] * rate / 64.0
            out[key] = {"kernel": out[base]["kernel"], "items": 1 << 18, "instr_per_item": out[base]["instr_per_item"],
                        "items_per_sec": rate, "rate_is": "live (this run)", "achieved": ach, "frac": ach / peak,
                        "waves_per_simd": n_waves}
    for key, kernel, grid, items, what in (
            ("verify_keyed_4096", "sp::ecdsa_verify_keyed_kernel", 4096, 4096, "signatures"),
            ("message_hash_chains", "sp::ped_chain_kernel<2>", 65536, 3 * 4096, "hashes (4096 chains x 3)"),
            ("tree_paths", "sp::ped_path_kernel<2, true>", 65536, None, "4096 paths x the unmerged levels")):
        e = entry(kernel, grid)
        if not e:
            continue
        ach = e["valu_wave_instr"] / (e["duration_us_under_pmc"] * 1e-6)
        out[key] = {"kernel": kernel, "items": items, "item": what, "valu_wave_instr_per_launch": e["valu_wave_instr"],
                    "instr_per_wave": e["instr_per_wave"], "launch_us": e["duration_us_under_pmc"],
                    "rate_is": "the counter pass's launch", "achieved": ach, "frac": ach / peak,
                    "waves_per_simd": e["waves_per_simd"],
                    "ns_per_dependent_instr": 1e3 * e["duration_us_under_pmc"] / e["instr_per_wave"]}
        if items:
            out[key]["instr_per_item"] = e["valu_wave_instr"] * 64.0 / items  # lane-instructions per item
    return out

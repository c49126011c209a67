"""Assembly of the headline workload's result (2^16-leaf Pedersen Merkle rebuilds, BASELINE.json configs[1]) from what
bench.py's timed region measured: the roofline of the dominant kernel, the sustained / burst windows, the config."""
import ctypes
import sys

from .common import ALGO_BYTES_PER_HASH, DTYPE, HEIGHT, median
from .roofline import (PEAK_BASIS_DOC, add_held_clock, hbm_object, merkle_config_key, pmc_traffic, valu_cycles_per_instr,
                       valu_issue, valu_peak)


def plan(k, cap):
    """K steps (trees) as the fewest lockstep calls of <= cap trees each, evenly sized.  Measured
    (tools/plan_sweep*.sh): the larger the forest the better - one call of 64 beats two of 32 on
    two streams (6.5 vs 6.1 x 10^8 hashes/s), and two calls of 64 on two streams overlap their
    latency-bound tops (7.5 x 10^8)."""
    if k <= 0:
        return []
    calls = (k + cap - 1) // cap
    base, rem = divmod(k, calls)
    return [base + (1 if i < rem else 0) for i in range(calls)]


def combine_check(slot, world, _lib):
    """N > 1: the job root of tree 0 of the last call issued on stream 0, recomputed from the gathered
    sub-roots (rank order) through the library's host-pointer tree entry point - a check of the
    exchange and of the tree-major transposition, independent of the lockstep device path."""
    try:
        from starkperp import batch
        nb = slot["last_nb"]
        top = slot["top"][: nb * (2 * world - 1)].cpu().numpy().astype("<i8")
        felts = _lib.unpack_felts((ctypes.c_uint64 * (4 * top.shape[0])).from_buffer_copy(top.tobytes()), top.shape[0])
        leaves, root = felts[:world], felts[nb * (2 * world - 1) - nb]
        return batch.merkle_root(leaves) == root
    except Exception as e:  # noqa: BLE001 - a failed self-check must not void the measurement
        sys.stderr.write("bench: combine check skipped (%s)\n" % e)
        return None


def kernel_roofline(prof, wbits, config_key, value_per_gpu, held_mhz, share_of_hashes):
    """`roofline` of ped_accumulate_kernel from the HIP events around its launches inside the timed region.
    prof = (total ms, launches, hashes) of sp_profile_end."""
    k_ms, k_launches, k_units = prof
    n_l = max(int(k_launches), 1)
    avg_launch_s = (k_ms / 1e3) / n_l
    hashes_per_launch = k_units / n_l
    kernel_rate = hashes_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0  # hashes/s inside the bulk launches
    roof = valu_issue(kernel_rate, wbits, "inside the ped_accumulate_kernel launches of the timed region (HIP events "
                                          "around each of them)", include_finish=False) or {
        "bound": "valu_issue", "achieved": None, "peak": valu_peak(), "unit": "wave64 VALU instr/s", "frac": None}
    traffic = pmc_traffic("sp::ped_accumulate_kernel", config_key)
    roof.update({
        "kernel": "ped_accumulate_kernel (one lane per hash: every level of more than 65 536 hashes; %.0f %% of "
                  "the hashes of this run)" % (100.0 * share_of_hashes),
        "peak_basis": PEAK_BASIS_DOC,
        "launches": int(k_launches), "hashes_per_launch": hashes_per_launch, "avg_launch_us": avg_launch_s * 1e6,
        "timing": "HIP events around every ped_accumulate_kernel launch inside the timed region, on the "
                  "stream it is launched on (sp_profile_begin/_end)",
        "traffic": (traffic or {}).get("bytes_per_launch"),
        "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE + WRITE_SIZE); algorithmic: %d" % int(
            ALGO_BYTES_PER_HASH * hashes_per_launch),
        "traffic_detail": traffic,
        "hbm": dict(hbm_object(kernel_rate),
                    note="the roofline the contract names; this kernel is integer-ALU bound (about 27 k VALU "
                         "instructions per 96 algorithmic bytes), so the HBM fraction says nothing about it"),
        "whole_region": valu_issue(value_per_gpu, wbits, "every kernel of the timed region: hashes/s per GPU over the "
                                                         "wall time (latency-bound upper levels included)", whole_forest=True),
        "frac_basis": "peak at the NOMINAL 2.4 GHz with c_mix = %.2f cycles per wave64 instruction" % valu_cycles_per_instr(),
    })
    add_held_clock(roof, held_mhz)
    if roof.get("whole_region"):
        add_held_clock(roof["whole_region"], held_mhz)
        roof["whole_region"].pop("held_clock_mhz", None)
    return roof


def merkle_detail(args, world, lib, tel, m):
    """The full result of the headline workload.  `m`: what the timed region measured (see bench.py main):
    regions, preheat_regions, burst_regions, burst_prof, prof, windows (perf_counter pairs), timed_plan, n_streams,
    trees_per_call_cap, idle_tel."""
    n_leaves = 1 << HEIGHT
    regions = m["regions"]
    srt = sorted(regions)
    elapsed = median(regions)
    hashes_per_step = world * (n_leaves - 1) + (world - 1)
    value = hashes_per_step * args.steps / elapsed
    wbits = int(lib.sp_window_bits())
    timed_plan, n_streams = m["timed_plan"], m["n_streams"]
    tel_sus = tel.window(*m["sustained_window"]) if tel else None
    tel_burst = tel.window(*m["burst_window"]) if tel else None
    held_mhz = tel_sus and tel_sus.get("sclk_mhz_median")
    share = m["prof"][2] / max(len(regions) * hashes_per_step * args.steps / max(world, 1), 1)
    roof = kernel_roofline(m["prof"], wbits, merkle_config_key(args.steps, timed_plan, n_streams, wbits),
                           value / max(world, 1), held_mhz, share)
    burst_med = median(m["burst_regions"])
    b_ms, _, b_u = m["burst_prof"]
    burst_value = hashes_per_step * args.steps / burst_med
    preheat = m["preheat_regions"]
    return {
        "metric": "pedersen_hashes_per_sec",
        "value": value,
        "unit": "hashes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "timed_regions": {"count": len(regions), "median_s": elapsed, "mean_s": sum(regions) / len(regions),
                          "min_s": srt[0], "max_s": srt[-1], "p10_s": srt[len(srt) // 10],
                          "p90_s": srt[(9 * len(srt)) // 10], "total_s": sum(regions),
                          "preheat_s": sum(preheat), "preheat_regions": len(preheat),
                          "value_from_mean_region": hashes_per_step * args.steps * len(regions) / sum(regions),
                          "note": "SUSTAINED: after %.2f s of the same call as pre-heat, the region (exactly "
                                  "--steps steps between barrier + synchronize fences) is repeated until %.1f s "
                                  "have been timed; value and ms_per_step come from the MEDIAN region of that "
                                  "window" % (sum(preheat), args.min_timed_s)},
        "burst": {"value": burst_value, "unit": "hashes/s", "median_s": burst_med, "count": len(m["burst_regions"]),
                  "total_s": sum(m["burst_regions"]),
                  "bulk_kernel_hashes_per_sec": (b_u / (b_ms / 1e3)) if b_ms > 0 else None,
                  "telemetry": tel_burst,
                  "note": "the %.0f ms window rounds 1 - 4 reported as `value`, taken first, straight after the CPU-only "
                          "set-up: the power controller has not settled yet" % (1e3 * args.burst_s)},
        "sustained_over_burst": value / burst_value if burst_value else None,
        "telemetry": dict(tel.describe(), sustained=tel_sus, burst=tel_burst, idle_before=m.get("idle_tel"),
                          sclk_mhz_median=held_mhz, power_w_median=tel_sus and tel_sus.get("power_w_median"),
                          note="rank 0's device, sampled every %.0f ms from a side thread while the regions run; "
                               "`sustained` covers exactly the window `value` comes from" % (1e3 * tel.period)) if tel else None,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE,
        "dtype_note": "29-bit limbs in 32-bit registers, 64-bit accumulators: full-width arithmetic mod p=2^251+17*2^192+1",
        "data": "synthetic",
        "config": {
            "workload": "2^16-leaf Pedersen Merkle rebuild per GPU (BASELINE.json configs[1]), %d rebuilds in lockstep"
                        % max(timed_plan),
            "tree_height": HEIGHT,
            "leaves_per_gpu": n_leaves,
            "hashes_per_step": hashes_per_step,
            "trees_in_timed_call": timed_plan[0] if len(set(timed_plan)) == 1 else max(timed_plan),
            "calls_per_region": len(timed_plan),
            "trees_per_call_cap": m["trees_per_call_cap"],
            "streams": n_streams,
            "timed_calls": timed_plan,
            "ms_per_step_note": "steps advance in lockstep: ms_per_step is wall time / steps, not the latency "
                                "of one rebuild (extra.single_tree_rebuild_ms_one_stream has that)",
            "window_bits": wbits,
            "table_mib": lib.sp_table_bytes() / 2**20,
            "combine": "none" if world == 1 else "all_gather of %d sub-roots (RCCL) + %d top hashes" % (world, world - 1),
        },
        "roofline": roof,
    }

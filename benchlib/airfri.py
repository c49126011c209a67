"""The AIR + FRI half of BASELINE.json's metric: the `airfri` object of the default line (configs[3]) and the
`--workload airfri` runs (configs[3] per GPU; configs[4] as ONE trace sharded over the ranks)."""
import ctypes
import time

from .common import ALGO_BYTES_PER_HASH, DTYPE, HBM_PEAK_GBS, seeded_felts
from . import telemetry as _tel
from .line import emit
from .provenance import build_provenance
from .roofline import (AIRFRI_PMC_FILES, PEAK_BASIS_DOC, add_held_clock, airfri_config_key, hbm_object, pmc_traffic,
                       valu_issue, valu_peak)


def run_airfri(args, torch, dist, lib, _lib, dev, rank, world, cpu_leg=None):
    """configs[3] (N = 1: independent 2^k-row jobs alternating over streams) and configs[4] (N > 1: ONE
    trace of N * 2^k rows sharded over the ranks by starkperp.sharded_prover - LDE units spread over the
    ranks, ONE bulk all-to-all into block-cyclic LDE-row shards with halos, per-block subtrees + all_gather
    of block roots, shard-local folds; every root equals the single-GPU root of the same trace)."""
    import random
    from starkperp import stark
    if not 10 <= args.log_rows <= 24:
        raise SystemExit("--log-rows must be in 10..24")
    total_log_rows = args.log_rows + (world.bit_length() - 1)
    if world > 1 and (world & (world - 1) or total_log_rows > 25):
        raise SystemExit("the sharded job needs a power-of-two world and at most 2^25 rows in all")
    m = 1 << (total_log_rows - 9)  # 512 trace rows per hash; N > 1: the WHOLE trace (every rank holds its inputs)
    log_lde = total_log_rows + 2
    P = stark.FIELD_PRIME
    xs, ys = seeded_felts(torch, m, 11, dev), seeded_felts(torch, m, 12, dev)
    rng = random.Random(13)
    alphas = [rng.randrange(P) for _ in range(stark.N_CONSTRAINTS)]
    betas = [rng.randrange(P) for _ in range(log_lde - 6)]
    trace = stark.pedersen_trace(xs, ys)  # witness generation is input preparation
    if dist is not None:  # N > 1, or --force-dist at N = 1: the sharded path on a process group
        return run_airfri_sharded(args, torch, dist, lib, _lib, dev, rank, world, stark, trace, alphas, betas,
                                  total_log_rows)
    per = stark.periodic_lde(512 * m, stark.FIELD_GEN, dev)
    n_roots = 2 + (log_lde - 7)  # trace, composition, every FRI layer above 64 points
    # Independent jobs alternate over the streams: the latency-bound tree tops of one job overlap
    # the throughput-bound row hashing of the next (inside one job every phase depends on the last).
    n_streams = args.streams if args.streams > 0 else 3
    slots = []
    for si in range(n_streams):
        slots.append({
            "stream": torch.cuda.current_stream() if n_streams == 1 else torch.cuda.Stream(device=dev),
            "roots": torch.zeros((n_roots, 4), dtype=torch.int64, device=dev),
        })
    job_counter = [0]

    def step():
        sl = slots[job_counter[0] % n_streams]
        job_counter[0] += 1
        with torch.cuda.stream(sl["stream"]):
            roots_dev = sl["roots"]
            k = 0
            job_trace = stark.pedersen_trace(xs, ys) if args.with_witness else trace
            t_lde = stark.lde(job_trace)
            roots_dev[k] = stark.commit_rows(t_lde)[-1]; k += 1
            comp = stark.air_eval(t_lde, per, 512 * m, alphas)
            roots_dev[k] = stark.commit_rows(comp.unsqueeze(0))[-1]; k += 1
            layer, sh, j = comp, stark.FIELD_GEN, 0
            while layer.shape[0] > 64:
                layer = stark.fri_fold(layer, betas[j], sh)
                sh = sh * sh % P
                j += 1
                if layer.shape[0] > 64:
                    roots_dev[k] = stark.commit_rows(layer.unsqueeze(0))[-1]; k += 1

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # the hash kernels dominate this job too (about 3/4 of its GPU time): same roofline leg as the
    # headline workload - HIP events around every accumulate launch of the timed region
    launches_per_step = 40 * (n_roots + 4)
    _lib.check(lib.sp_profile_begin(args.steps * launches_per_step), "profile_begin")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    k_ms, k_launches, k_units = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
    _lib.check(lib.sp_profile_end(ctypes.byref(k_ms), ctypes.byref(k_launches), ctypes.byref(k_units)),
               "profile_end")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tel_window = _tel.ACTIVE.window(t0, t1) if _tel.ACTIVE else None
    if rank == 0:
        avg_launch_s = (k_ms.value / 1e3) / max(k_launches.value, 1)
        achieved = (ALGO_BYTES_PER_HASH * k_units.value / max(k_launches.value, 1)) / avg_launch_s / 1e9 \
            if avg_launch_s > 0 else 0.0
        # trace rows: 3 chain hashes + 1 tree node per LDE row; then one tree per committed column
        hashes = 4 * (1 << log_lde) + (1 << log_lde) + sum((1 << k) for k in range(7, log_lde))
        emit({
            "metric": "air_fri_commits_per_sec", "value": world * args.steps / elapsed, "unit": "commits/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": "2^%d-row Pedersen-step trace per GPU: LDE x4 -> commit -> AIR -> commit -> "
                                   "%d FRI folds with %d layer commits (BASELINE.json configs[3] at 2^20; N GPUs "
                                   "= configs[4] as N disjoint row ranges; --log-rows 24 = configs[4] on one "
                                   "GPU)" % (args.log_rows, log_lde - 6, log_lde - 7),
                       "rows_per_gpu": 512 * m, "pedersen_hashes_per_job": hashes, "streams": n_streams,
                       "combine": "none" if world == 1 else "all_gather of 17 roots per rank + top hashes"},
            "roofline": dict(
                valu_issue((k_units.value / max(k_launches.value, 1)) / avg_launch_s if avg_launch_s > 0 else 0.0,
                           int(lib.sp_window_bits()), "inside the ped_accumulate_kernel launches of the timed region",
                           include_finish=False) or {"bound": "valu_issue", "achieved": None, "peak": valu_peak(), "frac": None},
                kernel="ped_accumulate_kernel (row chains and commit-tree levels above 65 536 hashes)",
                peak_basis=PEAK_BASIS_DOC, launches=int(k_launches.value),
                hashes_in_timed_launches=int(k_units.value), avg_launch_us=avg_launch_s * 1e6,
                timing="HIP events around every ped_accumulate_kernel launch inside the timed region",
                traffic=(pmc_traffic("sp::ped_accumulate_kernel", airfri_config_key(int(lib.sp_window_bits())), AIRFRI_PMC_FILES) or {}).get("bytes_per_launch"),
                traffic_detail=pmc_traffic("sp::ped_accumulate_kernel", airfri_config_key(int(lib.sp_window_bits())), AIRFRI_PMC_FILES),
                hbm={"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS}),
            "telemetry": dict(_tel.ACTIVE.describe(), timed=tel_window) if _tel.ACTIVE else None,
            "build": build_provenance(lib),
            "cpu_baseline": (cpu_leg(10) if (cpu_leg and world == 1 and not args.no_cpu_baseline) else None),
        })
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_airfri_sharded(args, torch, dist, lib, _lib, dev, rank, world, stark, trace, alphas, betas, total_log_rows):
    from starkperp import sharded_prover
    ops = sharded_prover.GpuOps(dev)
    log_lde = total_log_rows + 2
    n_roots = 2 + (log_lde - 7)
    out = {}
    stats = {}

    def step():
        out["roots"], out["final"] = sharded_prover.commit_job(ops, dist, trace, alphas, betas, stats=stats)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    _lib.check(lib.sp_profile_begin(args.steps * 40 * (n_roots + 4)), "profile_begin")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, k_launches, k_units = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
    _lib.check(lib.sp_profile_end(ctypes.byref(k_ms), ctypes.byref(k_launches), ctypes.byref(k_units)), "profile_end")
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist.get_backend() == "gloo":
        t = t.cpu()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    same = None
    if rank == 0 and total_log_rows <= 24:
        # the whole trace once more on this GPU alone (outside the timed region): the sharded roots must be
        # the single-GPU roots
        n = trace.shape[1]
        t_lde = stark.lde(trace)
        ref = [stark.root_of(stark.commit_rows(t_lde))]
        comp = stark.air_eval(t_lde, stark.periodic_lde(n, stark.FIELD_GEN, dev), n, alphas)
        del t_lde
        ref.append(stark.root_of(stark.commit_rows(comp.unsqueeze(0))))
        layer, sh, k = comp, stark.FIELD_GEN, 0
        while layer.shape[0] > 64:
            layer = stark.fri_fold(layer, betas[k], sh)
            sh = sh * sh % stark.FIELD_PRIME
            k += 1
            if layer.shape[0] > 64:
                ref.append(stark.root_of(stark.commit_rows(layer.unsqueeze(0))))
        same = bool(ref == out["roots"] and stark.tensor_to_felts(layer) == out["final"])
    if rank == 0:
        n_l = max(int(k_launches.value), 1)
        avg_launch_s = (k_ms.value / 1e3) / n_l
        rate = (k_units.value / n_l) / avg_launch_s if avg_launch_s > 0 else 0.0
        lde_bytes = 4 * (4 << total_log_rows) * 32
        emit({
            "metric": "air_fri_commits_per_sec", "value": world * args.steps / elapsed,
            "unit": "2^%d-row commits/s (one step = ONE proof of %d x 2^%d rows)" % (args.log_rows, world, args.log_rows),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": "ONE 2^%d-row Pedersen-step trace sharded over %d GPUs (BASELINE.json configs[4] "
                                   "shape; 2^24 rows = --log-rows 21 on 8 GPUs): 16 LDE units of 4 interpolations spread over "
                                   "the ranks, ONE all-to-all into block-cyclic LDE-row shards with halos, per-block subtrees "
                                   "+ all_gather of block roots, shard-local folds"
                                   % (total_log_rows, world),
                       "rows_total": 1 << total_log_rows, "rows_per_gpu": 1 << args.log_rows,
                       "exchange": {"lde_all_to_all_bytes_total": lde_bytes,
                                    "per_commit": "all_gather of the block roots (32 B per block of 2^%d rows) + the top "
                                                  "levels on every rank" % stats.get("log_block", 0),
                                    "per_fold": "none: block-cyclic row shards keep both members of every fold pair on one "
                                                "rank; one all_gather of 2^%d felts per rank before the replicated tail"
                                                % stats.get("log_block", 0),
                                    "bytes_sent_by_rank0_per_job": stats.get("bytes_sent_by_this_rank"),
                                    "interpolations_on_rank0": stats.get("interpolations"),
                                    "backend": dist.get_backend()}},
            "sharded_roots_match_single_gpu": same,
            "roofline": dict(
                valu_issue(rate, int(lib.sp_window_bits()), "inside the ped_accumulate_kernel launches of the timed "
                           "region on rank 0", include_finish=False)
                or {"bound": "valu_issue", "achieved": None, "peak": valu_peak(), "frac": None},
                kernel="ped_accumulate_kernel (row chains and commit-tree levels above 65 536 hashes)",
                peak_basis=PEAK_BASIS_DOC, launches=int(k_launches.value), avg_launch_us=avg_launch_s * 1e6,
                traffic=(pmc_traffic("sp::ped_accumulate_kernel", airfri_config_key(int(lib.sp_window_bits())), AIRFRI_PMC_FILES) or {}).get("bytes_per_launch"),
                traffic_detail=pmc_traffic("sp::ped_accumulate_kernel", airfri_config_key(int(lib.sp_window_bits())), AIRFRI_PMC_FILES),
                hbm={"bound": "hbm", "achieved": ALGO_BYTES_PER_HASH * rate / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ALGO_BYTES_PER_HASH * rate / 1e9 / HBM_PEAK_GBS}),
            "cpu_baseline": None,
        })
    dist.barrier()
    dist.destroy_process_group()


def airfri_object(torch, lib, _lib, dev, cpu_leg=None, brief=False, fence=None, min_timed_s=2.0, preheat_s=0.7):
    """BASELINE.json configs[3], the second half of the metric: one 2^20-row Pedersen-step trace ->
    4-column LDE to 2^22 -> commit -> composition -> commit -> 16 folds with 15 layer commits (25.2 M
    Pedersen hashes).  Inputs (the witness) resident in HBM.  commits_per_sec times independent jobs
    alternating over three streams, exactly what `--workload airfri` times per GPU.
    brief (the N > 1 form of the default line): every rank runs its own jobs - `fence` (barrier + synchronize)
    lines the ranks up in front of the timed jobs - and the object stops after the rates and the hash roofline."""
    import random
    from starkperp import stark
    m = 2048
    n_rows, n_lde, cols = 512 * m, 4 * 512 * m, 4
    xs, ys = seeded_felts(torch, m, 11, dev), seeded_felts(torch, m, 12, dev)
    rng = random.Random(13)
    P = stark.FIELD_PRIME
    alphas = [rng.randrange(P) for _ in range(stark.N_CONSTRAINTS)]
    betas = [rng.randrange(P) for _ in range(16)]
    trace = stark.pedersen_trace(xs, ys)  # witness generation is input preparation
    per = stark.periodic_lde(n_rows, stark.FIELD_GEN, dev)
    torch.cuda.synchronize()

    def timed(fn, iters):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    def job():
        t_lde = stark.lde(trace)
        stark.commit_rows(t_lde)
        comp = stark.air_eval(t_lde, per, n_rows, alphas)
        stark.commit_rows(comp.unsqueeze(0))
        layer, sh, k = comp, stark.FIELD_GEN, 0
        while layer.shape[0] > 64:
            layer = stark.fri_fold(layer, betas[k], sh)
            sh = sh * sh % P
            k += 1
            if layer.shape[0] > 64:
                stark.commit_rows(layer.unsqueeze(0))

    hashes = 4 * n_lde + n_lde + sum((1 << k) for k in range(7, 22))
    out = {"workload": "2^20-row trace, blowup 4, 11 constraints, folds down to 64 points (BASELINE.json configs[3])",
           "pedersen_hashes_per_job": hashes, "data": "synthetic", "dtype": DTYPE}
    job()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]

    def pipelined(njobs):
        for i in range(njobs):
            with torch.cuda.stream(streams[i % 3]):
                job()
        torch.cuda.synchronize()

    # Three windows, like the headline: a burst of 9 jobs straight away (what rounds 1 - 4
    # reported), pre-heat, then batches of 9 jobs until `min_timed_s` seconds have been timed: commits_per_sec is the
    # rate over that whole sustained window, with the clock and power the chip held in it.
    pipelined(3)
    if fence is not None:
        fence()
    t0 = time.perf_counter()
    pipelined(9)
    burst_rate = 9 / (time.perf_counter() - t0)
    heat_t0 = time.perf_counter()
    while time.perf_counter() - heat_t0 < preheat_s:
        pipelined(9)
    if fence is not None:
        fence()
    n_batches = max(1, int(min_timed_s * burst_rate / 9 + 0.999))  # fixed in advance: ranks stay in step
    sus_t0 = time.perf_counter()
    for _ in range(n_batches):
        pipelined(9)
    sus_t1 = time.perf_counter()
    out["commits_per_sec"] = 9 * n_batches / (sus_t1 - sus_t0)
    out["commits_per_sec_burst"] = burst_rate
    out["timed"] = {"jobs": 9 * n_batches, "seconds": sus_t1 - sus_t0, "preheat_s": sus_t0 - heat_t0,
                    "telemetry": _tel.ACTIVE.window(sus_t0, sus_t1) if _tel.ACTIVE else None}
    # one job after the other on one stream (the chip is hot now), with HIP events around the bulk hash launches
    _lib.check(lib.sp_profile_begin(3 * 64), "profile_begin")
    t_seq = timed(job, 2)
    k_ms, k_launches, k_units = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
    _lib.check(lib.sp_profile_end(ctypes.byref(k_ms), ctypes.byref(k_launches), ctypes.byref(k_units)), "profile_end")
    out["seconds_per_job_one_stream"] = t_seq
    out["commits_per_sec_note"] = "%d independent jobs alternating over 3 streams (tree tops of one job beside the row " \
                                  "hashing of the next) in %.2f s after %.2f s of pre-heat; the first 9 jobs out of idle " \
                                  "ran at %.1f commits/s; one job at a time: %.1f commits/s" % (
                                      9 * n_batches, sus_t1 - sus_t0, sus_t0 - heat_t0, burst_rate, 1.0 / t_seq)
    n_l = max(int(k_launches.value), 1)
    rate = (k_units.value / n_l) / ((k_ms.value / 1e3) / n_l) if k_ms.value > 0 else 0.0
    roof = valu_issue(rate, int(lib.sp_window_bits()),
                      "inside the ped_accumulate_kernel launches of three sequential jobs (HIP events)",
                      include_finish=False) or {"bound": "valu_issue", "achieved": None, "peak": valu_peak(), "frac": None}
    roof.update({"kernel": "ped_accumulate_kernel (row chains and the tree levels above 65 536 hashes: %.0f %% of the "
                           "job's hashes)" % (100.0 * k_units.value / (3.0 * hashes)),
                 "launches": int(k_launches.value), "avg_launch_us": (k_ms.value / n_l) * 1e3,
                 "traffic": (pmc_traffic("sp::ped_accumulate_kernel", airfri_config_key(int(lib.sp_window_bits())), AIRFRI_PMC_FILES) or {}).get("bytes_per_launch"),
                 "traffic_detail": pmc_traffic("sp::ped_accumulate_kernel", airfri_config_key(int(lib.sp_window_bits())), AIRFRI_PMC_FILES),
                 "hbm": hbm_object(rate)})
    held = ((out["timed"]["telemetry"] or {}).get("sclk_mhz_median")) if out.get("timed") else None
    add_held_clock(roof, held)
    if brief:
        out["roofline"] = roof
        return out
    out["witness_generation_seconds"] = timed(lambda: stark.pedersen_trace(xs, ys), 2)
    t_lde = stark.lde(trace)
    comp = stark.air_eval(t_lde, per, n_rows, alphas)
    phase_s = {
        "lde_4cols_2p20_to_2p22": timed(lambda: stark.lde(trace), 3),
        "commit_trace_lde_4cols_2p22_rows": timed(lambda: stark.commit_rows(t_lde), 2),
        "air_eval_2p22_points": timed(lambda: stark.air_eval(t_lde, per, n_rows, alphas), 3),
        "commit_composition_2p22_rows": timed(lambda: stark.commit_rows(comp.unsqueeze(0)), 2),
        "fri_fold_first_layer_2p22": timed(lambda: stark.fri_fold(comp, betas[0], stark.FIELD_GEN), 5),
    }
    # algorithmic bytes: an NTT pass reads and writes each felt once; the 2^20-point inverse
    # transform takes 2 passes, the 2^22-point forward one 3, the first of which reads the 2^20 coefficients
    # (coset scaling and zero padding happen in LDS) and writes 2^22 points; composition: 7 trace + 6
    # periodic reads and one write of 32 B per point; fold: 32 B read, 16 B written per input point; commit
    # of M rows of W felts: 32 W M read, 32 (2 M) written
    algo = {
        "lde_4cols_2p20_to_2p22": cols * (2 * 64 * n_rows + 32 * n_rows + 32 * n_lde + 2 * 64 * n_lde),
        "commit_trace_lde_4cols_2p22_rows": 32 * cols * n_lde + 64 * n_lde,
        "air_eval_2p22_points": (7 + 6 + 1) * 32 * n_lde,
        "commit_composition_2p22_rows": 32 * n_lde + 64 * n_lde,
        "fri_fold_first_layer_2p22": (32 + 16) * n_lde,
    }
    dominant = {"lde_4cols_2p20_to_2p22": ("ntt_tile_kernel", "valu (one 156-instruction multiplication per butterfly "
                                           "and 64 B; 0.30 of 8 TB/s would be 100 % VALU issue)"),
                "commit_trace_lde_4cols_2p22_rows": ("ped_accumulate_kernel", "valu_issue"),
                "air_eval_2p22_points": ("air_eval_kernel", "valu"),
                "commit_composition_2p22_rows": ("ped_accumulate_kernel", "valu_issue"),
                "fri_fold_first_layer_2p22": ("fri_fold_kernel", "valu / hbm")}
    out["phases"] = {k: {"seconds": phase_s[k], "dominant_kernel": dominant[k][0], "bound": dominant[k][1],
                         "algorithmic_bytes": algo[k], "hbm_gb_per_s": algo[k] / phase_s[k] / 1e9,
                         "hbm_frac_of_8_tb_per_s": algo[k] / phase_s[k] / 1e9 / HBM_PEAK_GBS,
                         "traffic_per_launch_of_dominant_kernel": pmc_traffic(
                             "sp::" + dominant[k][0], airfri_config_key(int(lib.sp_window_bits())), AIRFRI_PMC_FILES)} for k in phase_s}
    out["roofline"] = roof
    stark.prove(xs, ys, n_queries=8, seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stark.prove(xs, ys, n_queries=8, seed=1)
    torch.cuda.synchronize()
    out["prove_seconds_own_witness_8_queries"] = time.perf_counter() - t0
    out["cpu_baseline"] = cpu_leg(10) if cpu_leg else None
    return out

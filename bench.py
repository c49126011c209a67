#!/usr/bin/env python3
"""
bench.py - headline benchmark of the MI355X hot path (contract: see the build brief).

A "step" is one rebuild of a 2^16-leaf Pedersen Merkle tree per GPU (BASELINE.json configs[1]:
"2^16-leaf position-tree Merkle rebuild"), inputs resident in HBM.  Steps are independent, so the K
steps are advanced as lockstep forests (sp_merkle_forest_dev: one launch per level serves every
tree of the call) - the upper levels of one rebuild are latency-bound and would leave most of the
chip idle.  With N > 1 ranks every step is a tree of N * 2^16 leaves: each rank rebuilds its own
2^16-leaf subtree (no data-path collective), the N sub-roots are exchanged with one RCCL all_gather
(N x 32 bytes per tree) and the log2(N) top levels are hashed on every rank - weak scaling.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line of at most 8 KB on stdout (benchlib/line.py holds it to that: round 5's
20 KB line could not be parsed by the driver) and writes everything else it measured - telemetry
windows, burst, region percentiles, per-phase tables, configs[2] timings, provenance - to
bench_detail.json beside this script and to stderr.  `value` = Pedersen hashes/s over the whole job.
  roofline      the dominant kernel (ped_accumulate_kernel, the one-lane-per-hash bulk launches) against
                the roofline that binds it - VALU issue - from HIP events around those launches inside
                the timed region; the HBM fraction the contract names is the `hbm` field; `traffic` = HBM
                bytes per launch from rocprofv3 PMC passes (profiles/, named in `sources`).
  airfri        the second half of BASELINE.json's metric: 2^20-row AIR+FRI commit jobs per second
                (configs[3]) with its own roofline fraction and CPU leg.  With N > 1 ranks: independent
                jobs on every GPU, commits_per_sec = N x the slowest rank's rate; ONE trace sharded over
                the ranks is `--workload airfri`.
  cpu_baseline  the oracle (pure-Python restatement of the reference algorithm) on the host cores, a
                bounded sample of the same tree; the C port and the optimised CPU comparator are in the
                detail file and in `summary`.

This file keeps the argument parsing, the timed region, the CPU-baseline legs (the only code outside tests/ and
smoke() that touches oracle/) and the print; the plumbing lives in benchlib/ (unit-tested on CPU).
"""
import argparse
import ctypes
import os
import sys
import time

# One hardware queue per HIP stream in use (the runtime default of 4 is enough for the default 2).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from benchlib import telemetry as _telemetry  # noqa: E402  (no torch import, no GPU touched at import time)
from benchlib.common import HEIGHT, median, seeded_felts  # noqa: E402
from benchlib.launch import dist_report, init_library_one_plan, open_process_group, reduce_scalar, self_spawn  # noqa: E402
from benchlib.line import emit, main_line  # noqa: E402
from benchlib.merkle import combine_check, merkle_detail, plan  # noqa: E402
from benchlib.provenance import build_provenance  # noqa: E402
from benchlib.telemetry import Telemetry  # noqa: E402,F401  (re-exported: tests and tools read bench.Telemetry)


# ---- CPU-baseline legs: the oracle as the checker / the comparator, never the thing measured or shipped ----------

def _cpu_hash_chunk(pairs):
    from oracle import ref_py
    return [ref_py.pedersen_hash(a, b) for a, b in pairs]


def cpu_baseline(leaf_ints, budget_s=8.0):
    """Oracle ("port" of the reference algorithm: affine adds, one ext-Euclid inversion each) on
    the first level of the same tree, all host cores, bounded sample."""
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 64)
    # calibrate on one core
    t0 = time.time()
    _cpu_hash_chunk([(leaf_ints[0], leaf_ints[1])] * 4)
    per_hash = (time.time() - t0) / 4
    n = int(budget_s * cores / max(per_hash, 1e-6))
    n = max(cores * 4, min(n, len(leaf_ints) // 2))
    pairs = [(leaf_ints[2 * i], leaf_ints[2 * i + 1]) for i in range(n)]
    chunk = max(1, n // (cores * 4))
    chunks = [pairs[i : i + chunk] for i in range(0, n, chunk)]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_hash_chunk, [[pairs[0]]] * cores)  # spin up workers
        t0 = time.time()
        outs = pool.map(_cpu_hash_chunk, chunks)
        dt = time.time() - t0
    flat = [v for c in outs for v in c]
    return {
        "value": n / dt,
        "unit": "hashes/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d node hashes of level 1 of the same 2^16-leaf tree (oracle/ref_py.py, "
                  "multiprocessing over %d cores, %.1f s)" % (n, cores, dt),
    }, flat


def cpu_baseline_c(leaf_ints, gpu_root):
    """Second CPU baseline: the plain-C restatement of the same reference algorithm
    (oracle/starkref.c: affine adds, one inversion each), OpenMP over the host cores, on the WHOLE
    2^16-leaf rebuild - which is also a full-size parity check of the timed tree."""
    from oracle import cref
    t0 = time.time()
    levels = cref.merkle_levels(leaf_ints)
    dt = time.time() - t0
    return {"value": (len(leaf_ints) - 1) / dt, "unit": "hashes/s", "cores": cref.max_threads(),
            "kind": "port", "sample": "the complete 2^16-leaf rebuild (65535 hashes) in %.2f s, oracle/starkref.c "
                                      "with OpenMP" % dt,
            "root_matches_gpu": levels[-1][0] == gpu_root}


def cpu_baseline_opt(leaf_ints, gpu_root):
    """Third CPU baseline, the one a CPU library would ship: the same function
    with 8-bit fixed-base window tables and batched affine additions (one shared inversion per window and
    256 hashes; last section of oracle/starkref.c), OpenMP over the host cores, the WHOLE 2^16-leaf rebuild
    repeated until about two seconds have passed."""
    from oracle import cref
    levels, reps, dt = cref.opt_merkle_timed(leaf_ints, 2.0)
    return {"value": reps * (len(leaf_ints) - 1) / dt, "unit": "hashes/s", "cores": cref.max_threads(),
            "kind": "port", "algorithm": "optimised comparator: fixed-base 8-bit windows (63 table additions per "
                                         "hash) + Montgomery's trick over 256 hashes, affine coordinates",
            "sample": "%d complete 2^16-leaf rebuilds (65535 hashes each) in %.2f s inside the C library (window "
                      "table built and leaves marshalled before the clock starts)" % (reps, dt),
            "root_matches_gpu": levels[-1][0] == gpu_root}


def cpu_airfri_baseline(log_rows=10):
    """CPU side of the `airfri` object: the same commit job at 2^log_rows rows with oracle/stark_ref.py
    (plain-Python NTT / composition / folds over Python ints) and the C oracle's Pedersen hash for the
    commitments.  Build-defined like the GPU job (the reference has no prover): parity unpinned."""
    import random
    from oracle import cref, stark_ref as S
    P = S.P
    rng = random.Random(31)
    m = (1 << log_rows) // S.ROWS_PER_HASH
    inputs = [(rng.randrange(P), rng.randrange(P)) for _ in range(m)]
    trace = S.pedersen_trace(inputs)  # witness generation: input preparation, as on the GPU side
    n = len(trace[0])
    alphas = [rng.randrange(P) for _ in range(S.N_CONSTRAINTS)]
    log_lde = log_rows + 2
    betas = [rng.randrange(P) for _ in range(log_lde - 6)]

    def commit(columns):
        leaves = list(columns[0])
        for col in columns[1:]:
            leaves = cref.pedersen_hash_many(leaves, list(col))[0]
        return cref.merkle_levels(leaves)[-1][0]

    t0 = time.time()
    t_lde = [S.lde(col) for col in trace]
    roots = [commit(t_lde)]
    comp = S.composition_on_coset(t_lde, S.periodic_lde(n), n, alphas)
    roots.append(commit([comp]))
    layer, sh = comp, S.GEN
    for k in range(log_lde - 6):
        layer = S.fri_fold(layer, betas[k], sh)
        sh = sh * sh % P
        if len(layer) > 64:
            roots.append(commit([layer]))
    dt = time.time() - t0
    hashes = 4 * (1 << log_lde) + (1 << log_lde) + sum((1 << k) for k in range(7, log_lde))
    return {"value": 1.0 / dt, "unit": "commits/s of a 2^%d-row job" % log_rows, "cores": cref.max_threads(),
            "kind": "port", "parity": "build-defined, parity unpinned (the reference has no prover; BASELINE.md 3.5)",
            "sample": "one 2^%d-row job (LDE x4, 2 + %d commitments, composition, %d folds; %d Pedersen hashes "
                      "through oracle/starkref.c with OpenMP, transforms in plain Python) in %.1f s"
                      % (log_rows, log_lde - 7, log_lde - 6, hashes, dt),
            "seconds": dt,
            "scaled_to_2p20_rows": {"commits_per_sec": 1.0 / (dt * (1 << (20 - log_rows))),
                                    "how": "work is linear in the rows up to log factors: x %d" % (1 << (20 - log_rows))}}


def _c1_ecdsa_inputs():
    """BASELINE.json configs[0] / SURVEY 8(d) C1: 64 (z, d) pairs from random.Random(0)."""
    import random
    from oracle import ref_py
    rng = random.Random(0)
    return [(rng.randrange(2**251), rng.randrange(1, ref_py.EC_ORDER)) for _ in range(64)]


def _cpu_sign_chunk(items):
    from oracle import ref_py
    return [ref_py.sign(z, d) for z, d in items]


def _cpu_verify_chunk(items):
    from oracle import ref_py
    return [ref_py.verify(z, r, s, q) for z, r, s, q in items]


def cpu_baseline_ecdsa(budget_s=1.5):
    """CPU legs of the ECDSA figures: the oracle's `sign` and `verify`
    (oracle/ref_py.py - the reference algorithm: RFC 6979 nonce, one affine ladder per signature, three
    251-step ladders per verification, one ext-Euclid inversion per group operation) on C1's 64 + 64
    inputs, one core and all cores, and the same inputs through the GPU library for parity."""
    import multiprocessing as mp
    from oracle import ref_py
    from starkperp import batch
    cores = min(os.cpu_count() or 1, 64)
    items = _c1_ecdsa_inputs()
    t0 = time.time()
    sigs1 = _cpu_sign_chunk(items[:4])
    t_sign1 = (time.time() - t0) / 4
    pubs = batch.public_keys_many([d for _, d in items])
    vitems = None
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_sign_chunk, [[items[0]]] * cores)  # spin up workers
        reps = max(1, int(budget_s * cores / max(t_sign1, 1e-6) / len(items)))
        work = (items * reps)
        chunk = max(1, len(work) // (cores * 2))
        t0 = time.time()
        outs = pool.map(_cpu_sign_chunk, [work[i:i + chunk] for i in range(0, len(work), chunk)])
        dt_sign = time.time() - t0
        sigs = [v for c in outs for v in c][:len(items)]
        vitems = [(z, r, s, pubs[i][0]) for i, ((z, _), (r, s)) in enumerate(zip(items, sigs))]
        t0 = time.time()
        v1 = _cpu_verify_chunk(vitems[:2])
        t_ver1 = (time.time() - t0) / 2
        vreps = max(1, int(budget_s * cores / max(t_ver1, 1e-6) / len(vitems)))
        vwork = vitems * vreps
        vchunk = max(1, len(vwork) // (cores * 2))
        t0 = time.time()
        vouts = pool.map(_cpu_verify_chunk, [vwork[i:i + vchunk] for i in range(0, len(vwork), vchunk)])
        dt_ver = time.time() - t0
    verdicts = [v for c in vouts for v in c][:len(vitems)]
    gpu_sigs = batch.sign_many([z for z, _ in items], [d for _, d in items])
    gpu_ok = batch.verify_many([z for z, _ in items], [r for r, _ in sigs], [s for _, s in sigs], [q[0] for q in pubs])
    return {"kind": "port", "cores": cores, "unit": "operations/s",
            "sign_per_sec_one_core": 1.0 / t_sign1, "sign_per_sec_all_cores": len(work) / dt_sign,
            "verify_per_sec_one_core": 1.0 / t_ver1, "verify_per_sec_all_cores": len(vwork) / dt_ver,
            "sample": "C1's 64 (z, d) pairs from random.Random(0): %d signs in %.1f s and %d x-only-key verifications in "
                      "%.1f s over %d cores (oracle/ref_py.py)" % (len(work), dt_sign, len(vwork), dt_ver, cores),
            "sign_matches_gpu": bool(gpu_sigs == sigs), "verify_matches_gpu": bool(list(gpu_ok) == verdicts),
            "all_verified": bool(all(verdicts))}

def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--trees-per-call", type=int, default=64,
                    help="cap on the independent 2^16-leaf rebuilds advanced in lockstep by one library call "
                         "(sp_merkle_forest_dev); 1 = strictly one tree per call")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the lockstep calls / independent jobs are issued on round-robin "
                         "(0 = 2 for the merkle workload, 3 for airfri)")
    ap.add_argument("--workload", choices=["merkle", "airfri"], default="merkle",
                    help="merkle = BASELINE.json configs[1] (default, the headline line); airfri = one 2^20-row AIR+FRI "
                         "commit job per GPU per step (configs[3]; N GPUs: the N * 2^20-row trace of configs[4] sharded)")
    ap.add_argument("--plan", default="",
                    help="comma-separated call sizes (trees per lockstep call, round-robin over the streams) for the "
                         "TIMED steps; must sum to --steps.  Default: benchlib.merkle.plan")
    ap.add_argument("--window-bits", type=int, default=26,
                    help="log2 of the entries per signed window of the Pedersen tables: 26 = 75 GiB of the 288 GB HBM, "
                         "19 table entries per hash; 0 = the library default 21 = 4.3 GiB, 23 entries per hash.  If the "
                         "wide tables cannot be allocated the bench falls back to the library default and says so")
    ap.add_argument("--log-rows", type=int, default=20, help="airfri workload: log2 of the trace rows per GPU")
    ap.add_argument("--with-witness", action="store_true",
                    help="airfri workload: generate the trace (witness) inside every job instead of as input preparation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--force-dist", action="store_true", default=os.environ.get("STARKPERP_BENCH_FORCE_DIST") == "1",
                    help="--gpus 1 only: create a world-size-1 RCCL process group anyway and take every branch the N > 1 "
                         "runs take - the one-GPU rehearsal of the multi-GPU launch")
    ap.add_argument("--min-timed-s", type=float, default=float(os.environ.get("STARKPERP_BENCH_MIN_TIMED_S", "3.0")),
                    help="the timed regions (each EXACTLY --steps steps between fences) are repeated until this many "
                         "seconds have been timed; value = the median region of that sustained window")
    ap.add_argument("--preheat-s", type=float, default=float(os.environ.get("STARKPERP_BENCH_PREHEAT_S", "1.0")),
                    help="seconds of the timed call itself issued (untimed) in front of the sustained window, so that "
                         "the power controller has settled (profiles/r03_power_clock_bulk.txt: ~0.7 s)")
    ap.add_argument("--burst-s", type=float, default=0.05,
                    help="the 50 ms window of rounds 1 - 4, taken first, straight out of idle: reported as `burst`")
    ap.add_argument("--no-airfri", action="store_true", help="merkle workload: skip the `airfri` object")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver
            # would, and hand their single JSON line through
            return self_spawn(args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # Test hook for boxes with ONE GPU (tests/test_gpu_bench_ranks.py): every rank on device 0 and the
    # sub-root exchange over gloo, so that the N > 1 code path runs end to end.  Never set by the driver.
    share_gpu = os.environ.get("STARKPERP_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    forced_dist = bool(args.force_dist and world == 1)
    dist = open_process_group(torch, dev, world, forced_dist, share_gpu) if (world > 1 or forced_dist) else None

    from starkperp import _lib
    from starkperp.distributed import combine_forest_dev
    lib, wide_error = init_library_one_plan(torch, dist, dev, dev_index, args.window_bits)
    tel = _telemetry.ACTIVE = Telemetry(dev_index).start() if rank == 0 else None
    if args.workload == "airfri":
        from benchlib.airfri import run_airfri
        return run_airfri(args, torch, dist, lib, _lib, dev, rank, world, cpu_leg=cpu_airfri_baseline)

    n_leaves = 1 << HEIGHT
    n_streams = args.streams if args.streams > 0 else 2
    B = max(1, min(1024, int(args.trees_per_call)))  # independent rebuilds advanced in lockstep per call
    if args.plan:
        timed_plan = [int(v) for v in args.plan.split(",")]
        if sum(timed_plan) != args.steps or min(timed_plan) < 1 or max(timed_plan) > 1024:
            raise SystemExit("--plan must be positive call sizes <= 1024 summing to --steps")
    else:
        timed_plan = plan(args.steps, B)
    sizes = sorted(set([B] + plan(args.warmup, B) + timed_plan))
    max_b = sizes[-1]
    leaves = seeded_felts(torch, n_leaves * max_b, 1000 + rank, dev)  # distinct leaves for every tree
    slots = []
    for _ in range(n_streams):
        bufs = {}
        for nb in sizes:  # one forest buffer per call size; tree t always gets the same seeded leaves
            lv = torch.zeros((nb * (2 * n_leaves - 1), 4), dtype=torch.int64, device=dev)
            lv[: n_leaves * nb] = leaves[: n_leaves * nb]
            bufs[nb] = lv
        slots.append({
            "levels": bufs,
            "gathered": torch.zeros((max(world, 1) * max_b, 4), dtype=torch.int64, device=dev),
            "top": torch.zeros((2 * max(world, 1) * max_b - max_b, 4), dtype=torch.int64, device=dev),
            "stream": torch.cuda.current_stream() if n_streams == 1 else torch.cuda.Stream(device=dev),
        })
    stream = torch.cuda.current_stream().cuda_stream
    call_counter = [0]

    def issue(nb):
        """One lockstep call: nb complete 2^16-leaf rebuilds (+ the cross-rank combine)."""
        sl = slots[call_counter[0] % n_streams]
        call_counter[0] += 1
        sl["last_nb"] = nb
        with torch.cuda.stream(sl["stream"]):
            h = sl["stream"].cuda_stream
            buf = sl["levels"][nb]
            _lib.check(lib.sp_merkle_forest_dev(buf.data_ptr(), nb, HEIGHT, None, h), "forest")
            if dist is not None:
                combine_forest_dev(lib, dist, buf[buf.shape[0] - nb:], sl["gathered"], sl["top"], nb, h)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in slots:  # size every stream's scratch before the timed region
        issue(max_b)
    for nb in plan(args.warmup, B):
        issue(nb)
    fence()
    # The inner nodes of the buffers the TIMED calls write are zeroed first, so that the parity legs below
    # (cpu_baseline*, matches_gpu / root_matches_gpu) check what the timed region itself computed.
    first_timed = call_counter[0]
    timed_targets = []
    for i, nb in enumerate(timed_plan):
        sl = slots[(first_timed + i) % n_streams]
        if (id(sl), nb) not in [(id(a), b) for a, b in timed_targets]:
            timed_targets.append((sl, nb))
            sl["levels"][nb][n_leaves * nb:] = 0
    fence()
    # One region = EXACTLY --steps steps between two fences (barrier + synchronize on both sides); a region of 20
    # lockstep trees lasts ~1.7 ms.  Three windows of such regions, one after the other:
    #   burst      --burst-s (50 ms) straight after the CPU-only set-up: what rounds 1 - 4 reported;
    #   pre-heat   --preheat-s (1 s) of the same call, untimed: the power controller settles;
    #   sustained  regions repeated until --min-timed-s (3 s) have been timed: `value` = the MEDIAN region of this
    #              window, with the shader clock and package power sampled beside it (Telemetry).
    # Every rank takes the same decisions: a region's time is MAX-reduced over the ranks before it is used.
    MAX_REGIONS = 1 << 15
    local_regions = []  # this rank's own clock for every region (the MAX over ranks is what `value` uses)

    def run_regions(min_s, max_regions):
        regs = []
        w0 = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            for nb in timed_plan:
                issue(nb)
            fence()
            dt = time.perf_counter() - t0
            local_regions.append(dt)
            regs.append(reduce_scalar(torch, dist, dev, dt, "MAX"))
            if sum(regs) >= min_s or len(regs) >= max_regions:
                return regs, (w0, time.perf_counter())

    def profiled(min_s, max_regions, est_regions):
        """run_regions with HIP events around every ped_accumulate_kernel launch of the window."""
        # levels of more than 65 536 hashes per call: at most log2(trees) of them
        n_events = min(int(est_regions) * sum(nb.bit_length() + 1 for nb in timed_plan) + 64, 100000)
        _lib.check(lib.sp_profile_begin(n_events), "profile_begin")
        regs, window = run_regions(min_s, max_regions)
        k_ms, k_launches, k_units = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(lib.sp_profile_end(ctypes.byref(k_ms), ctypes.byref(k_launches), ctypes.byref(k_units)),
                   "profile_end")
        return regs, window, (k_ms.value, int(k_launches.value), int(k_units.value))

    m = {"timed_plan": timed_plan, "n_streams": n_streams, "trees_per_call_cap": B,
         "idle_tel": tel.window(time.perf_counter() - 0.5, time.perf_counter()) if tel else None}
    m["burst_regions"], m["burst_window"], m["burst_prof"] = profiled(args.burst_s, 64, 64)
    m["preheat_regions"] = run_regions(args.preheat_s, MAX_REGIONS)[0] if args.preheat_s > 0 else []
    est = args.min_timed_s / max(median(m["burst_regions"]), 1e-5) * 1.25 + 16
    del local_regions[:]
    m["regions"], m["sustained_window"], m["prof"] = profiled(args.min_timed_s, MAX_REGIONS, est)
    local_value = (n_leaves - 1) * args.steps / median(local_regions)  # this rank's own subtrees over its own clock

    # The AIR + FRI half of the metric at N > 1: independent 2^20-row jobs on every GPU (BASELINE north_star:
    # "independent order batches ... shard across the 8 GPUs"), no data-path collective; the ranks start their
    # timed jobs together and the job rate of the node is n_gpus x the slowest rank's rate.
    airfri_multi = None
    if dist is not None and not args.no_airfri:
        from benchlib.airfri import airfri_object
        loc = airfri_object(torch, lib, _lib, dev, brief=True, fence=fence, min_timed_s=args.min_timed_s,
                            preheat_s=0.7 * args.preheat_s)
        slowest = reduce_scalar(torch, dist, dev, loc["commits_per_sec"], "MIN")
        loc.update({"commits_per_sec_slowest_gpu": slowest, "commits_per_sec": world * slowest,
                    "seconds_per_job_one_stream": 1.0 / reduce_scalar(torch, dist, dev, 1.0 / loc["seconds_per_job_one_stream"], "MIN"),
                    "n_gpus": world, "scaling": "weak",
                    "sharding": "independent 2^20-row jobs on every GPU, no data-path collective: commits_per_sec = "
                                "n_gpus x the slowest rank's rate; ONE trace over all ranks is `--workload airfri`"})
        airfri_multi = loc
    dist_info = dist_report(torch, dist, dev, dev_index, world, rank, forced_dist, local_value, lib) if dist is not None else None

    if rank == 0:
        result = merkle_detail(args, world, lib, tel, m)
        if dist is not None:
            result["combine_matches_recomputed"] = combine_check(slots[0], world, _lib)
            result["dist"] = dist_info
            if wide_error:
                result["dist"]["window_plan_fallback"] = wide_error
        result["build"] = build_provenance(lib)

        # GPU legs and CPU legs alternate, so that the device's activity is spread over the run.  A secondary leg that
        # fails must not take the headline with it: it is reported under its key as {"error": ...} (traceback on
        # stderr) and the line is still printed.
        def leg(key, fn):
            try:
                result[key] = fn()
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                result[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                result.setdefault("failed_legs", []).append(key)

        def felts_of(t):
            return _lib.unpack_felts((ctypes.c_uint64 * (4 * t.shape[0])).from_buffer_copy(
                t.cpu().numpy().astype("<i8").tobytes()), t.shape[0])

        leaf_ints = gpu_root = None
        with_cpu = world == 1 and not args.no_cpu_baseline
        if with_cpu:
            def first_leg():
                # tree 0 of the forest buffer the FIRST TIMED call wrote (its inner nodes were zeroed before the timed
                # regions): the parity legs check the timed computation, not a warm-up forest
                nonlocal leaf_ints, gpu_root
                tsl, tnb = timed_targets[0]
                levels = tsl["levels"][tnb]
                leaf_ints = felts_of(levels[:n_leaves])
                base, cpu_out = cpu_baseline(leaf_ints)
                base["matches_gpu"] = felts_of(levels[n_leaves * tnb: n_leaves * tnb + len(cpu_out)]) == cpu_out
                base["compared_with"] = "level 1 of tree 0 in the %d-tree buffer written by the timed region" % tnb
                gpu_root = felts_of(levels[levels.shape[0] - tnb: levels.shape[0] - tnb + 1])[0]
                return base
            leg("cpu_baseline", first_leg)
        else:
            result["cpu_baseline"] = None
        if not args.no_airfri:
            if world == 1:
                from benchlib.airfri import airfri_object
                leg("airfri", lambda: airfri_object(torch, lib, _lib, dev,
                                                    cpu_leg=None if args.no_cpu_baseline else cpu_airfri_baseline,
                                                    min_timed_s=args.min_timed_s, preheat_s=0.7 * args.preheat_s))
            else:
                result["airfri"] = airfri_multi
            if forced_dist and airfri_multi is not None:  # the N > 1 reduction of the job rates, rehearsed at N = 1
                result["airfri_dist_rehearsal"] = {k: airfri_multi[k] for k in
                                                   ("commits_per_sec", "commits_per_sec_slowest_gpu", "n_gpus")}
        if with_cpu and leaf_ints is not None and gpu_root is not None:
            leg("cpu_baseline_c", lambda: cpu_baseline_c(leaf_ints, gpu_root))
            leg("cpu_baseline_opt", lambda: cpu_baseline_opt(leaf_ints, gpu_root))
        if world == 1 and not args.no_extras:
            from benchlib.extras import extras
            leg("extra", lambda: extras(torch, lib, _lib, dev, stream))
        if with_cpu:
            leg("cpu_baseline_ecdsa", cpu_baseline_ecdsa)
        emit(main_line(result), detail=result)
    if tel:
        tel.stop()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Multi-rank launch of bench.py: self-spawn through torch.distributed.run, process-group set-up, the group report."""
import ctypes
import json
import os
import sys

from .common import ROOT


def self_spawn(n):
    """bench.py --gpus N started without torch.distributed.run: launch N ranks on this node through it (one process
    per GPU, rendezvous on 127.0.0.1 at a free port), same arguments, stdout / stderr passed through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    sys.stderr.write("bench: --gpus %d without RANK/WORLD_SIZE: launching %s\n" % (n, " ".join(cmd[1:8])))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def dist_report(torch, dist, dev, dev_index, world, rank, forced, value, lib):
    """What the process group looked like - collective: every rank calls it.  Rank 0 gets
    {backend, world_size, rccl_version, per-rank device name / PCI bus / free HBM / window bits, the N x N
    hipDeviceCanAccessPeer matrix, the link types rocm-smi reports}; nothing here may break the line."""
    info = {"backend": dist.get_backend(), "world_size": world, "forced_at_one_gpu": forced}
    try:
        free_b, total_b = torch.cuda.mem_get_info(dev)
        pr = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device_index": dev_index, "name": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None),
                "pci": "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                             getattr(pr, "pci_device_id", 0)),
                "free_hbm_gib": free_b / 2**30, "total_hbm_gib": total_b / 2**30,
                "window_bits": int(lib.sp_window_bits()), "table_gib": lib.sp_table_bytes() / 2**30,
                "pid": os.getpid(), "cpus_allowed": len(os.sched_getaffinity(0)),
                "local_hashes_per_sec": value}
    except Exception as e:  # noqa: BLE001
        mine = {"rank": rank, "error": repr(e)}
    try:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        info["ranks"] = gathered
    except Exception as e:  # noqa: BLE001
        info["ranks"] = [mine]
        info["ranks_error"] = repr(e)
    if rank != 0:
        return info
    lv = [r.get("local_hashes_per_sec") for r in info["ranks"] if isinstance(r, dict) and r.get("local_hashes_per_sec")]
    if lv:
        info["per_rank_value"] = {"min": min(lv), "max": max(lv), "unit": "hashes/s on a rank's own clock (its 2^16-leaf "
                                  "subtrees per step; the job's value uses the slowest rank's region)"}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
    except Exception as e:  # noqa: BLE001
        info["rccl_version"] = "unknown (%s)" % type(e).__name__
    info["env"] = {k: os.environ[k] for k in ("NCCL_DEBUG", "NCCL_P2P_DISABLE", "NCCL_ALGO", "NCCL_PROTO", "RCCL_MSCCL_ENABLE",
                                               "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                               "GPU_MAX_HW_QUEUES") if k in os.environ}
    try:  # peer access between the devices the ranks run on, from this process (it sees all of them under torchrun)
        devs = [r.get("device_index", i) for i, r in enumerate(info["ranks"])]
        n_vis = torch.cuda.device_count()
        info["visible_devices"] = n_vis
        info["peer_access"] = [[(1 if a == b else int(torch.cuda.can_device_access_peer(a, b)))
                                if a < n_vis and b < n_vis else None for b in devs] for a in devs]
    except Exception as e:  # noqa: BLE001
        info["peer_access"] = "unavailable (%s)" % type(e).__name__
    if world > 1:
        try:
            import subprocess
            t = subprocess.run(["rocm-smi", "--showtopotype", "--json"], capture_output=True, text=True, timeout=30).stdout
            info["link_types"] = json.loads(t)
        except Exception as e:  # noqa: BLE001
            info["link_types"] = "unavailable (%s)" % type(e).__name__
    return info


def open_process_group(torch, dev, world, forced, share_gpu):
    """The process group of an N > 1 run (or of --force-dist at N = 1): RCCL ("nccl") with this rank's device, or gloo
    when the ranks share one GPU (the test hook).  Returns torch.distributed."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if forced:
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if share_gpu:
        dist.init_process_group("gloo")
        return dist
    dist.init_process_group("nccl", device_id=dev)
    # RCCL writes a version banner ("RCCL version : ...", five lines) to the C stdout when its first communicator
    # comes up; through a pipe it would sit in the stdio buffer and land AFTER the JSON line at exit.  Bring the
    # communicator up now with fd 1 pointed at stderr and flush: stdout carries ONE line.
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        t0 = torch.zeros(1, device=dev)
        dist.all_reduce(t0)
        torch.cuda.synchronize()
        ctypes.CDLL(None).fflush(None)
    finally:
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    return dist


def reduce_scalar(torch, dist, dev, value, op):
    """all_reduce of one float64 over the group (host tensor under gloo); `value` itself when there is no group."""
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    if dist.get_backend() == "gloo":
        t = t.cpu()
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
    return float(t.item())


def init_library_one_plan(torch, dist, dev, dev_index, window_bits):
    """sp_init with the bench's wide tables, falling back to the library default - and ONE table plan for the whole
    job: a single rank that cannot allocate the wide tables takes every rank to the default
    (ranks on different plans would still agree on every hash, but the weak-scaling figure would mix two kernels'
    rates).  Returns (lib, why_the_wide_tables_were_not_used or None)."""
    from starkperp import _lib
    wide_error = None
    try:
        lib = _lib.ensure_init(dev_index, window_bits or None)
    except _lib.StarkPerpError as e:
        if not window_bits:
            raise
        wide_error = str(e)
        lib = None
    if dist is not None and window_bits:
        ok = reduce_scalar(torch, dist, dev, 0.0 if lib is None else 1.0, "MIN")
        if ok == 0.0 and lib is not None:
            wide_error = "another rank could not allocate the %d-bit tables" % window_bits
            _lib.load().sp_shutdown()
            lib = None
    if lib is None:
        sys.stderr.write("bench: %d-bit tables unavailable (%s); using the library default\n" % (window_bits, wide_error))
        lib = _lib.ensure_init(dev_index, None)
    return lib, wide_error

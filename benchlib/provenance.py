"""Which binary produced a bench line."""
import ctypes
import os
import time

from .common import ROOT


def build_provenance(lib):
    """Which binary produced this line: sha256 of the loaded libstarkperp.so, what the
    library says it was compiled with (sp_build_info: compiler, HIP version, offload arch, compile date) and the
    toolchain found on THIS box."""
    import hashlib
    import subprocess
    from starkperp import _lib
    out = {"lib": os.path.relpath(_lib.LIB_PATH, ROOT)}
    try:
        h = hashlib.sha256()
        with open(_lib.LIB_PATH, "rb") as f:
            for blk in iter(lambda: f.read(1 << 20), b""):
                h.update(blk)
        out["lib_sha256"] = h.hexdigest()
        out["lib_bytes"] = os.path.getsize(_lib.LIB_PATH)
        out["lib_mtime_utc"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(os.path.getmtime(_lib.LIB_PATH)))
    except OSError as e:
        out["lib_sha256"] = "unreadable: %s" % e
    try:
        lib.sp_build_info.restype = ctypes.c_char_p
        out["compiled_with"] = lib.sp_build_info().decode()
    except Exception as e:  # noqa: BLE001
        out["compiled_with"] = "sp_build_info unavailable: %s" % e
    try:
        v = subprocess.run(["hipcc", "--version"], capture_output=True, text=True, timeout=20).stdout.splitlines()
        out["hipcc_on_this_box"] = "; ".join(l.strip() for l in v[:2])
    except Exception as e:  # noqa: BLE001
        out["hipcc_on_this_box"] = "not found (%s)" % type(e).__name__
    try:
        out["bench_py_sha16"] = hashlib.sha256(open(os.path.join(ROOT, "bench.py"), "rb").read()).hexdigest()[:16]
    except OSError:
        pass
    return out

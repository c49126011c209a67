import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "stark-perpetual_amd"))
import torch
torch.zeros(1, device="cuda")
from starkperp import _lib
lib = _lib.load()
mode = os.environ.get("STARKPERP_TABLE_BUILD", "doubling")
for rep in range(4):
    t = time.perf_counter(); _lib.check(lib.sp_init(0, 26), "init"); torch.cuda.synchronize(); a = time.perf_counter() - t
    t = time.perf_counter(); lib.sp_shutdown(); torch.cuda.synchronize(); b = time.perf_counter() - t
    if rep == 1: time.sleep(3)
    print(mode, "rep", rep, "init %.3f s, shutdown %.3f s" % (a, b), flush=True)

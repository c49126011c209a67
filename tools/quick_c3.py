#!/usr/bin/env python3
"""bench.py extras only (dev aid): prints the C3 entries."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stark-perpetual_amd"), os.path.join(ROOT, "tests")]
import importlib.util, torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from starkperp import _lib
lib = _lib.ensure_init(0, 26)
e = b.extras(torch, lib, _lib, torch.device("cuda", 0), torch.cuda.current_stream().cuda_stream)
for k in ("c3_4096_orders_host_inclusive_seconds", "c3_4096_orders_numpy_entry_points_seconds", "c3_4096_orders_one_call_seconds", "state_update_2048_positions_4096_orders_seconds", "single_tree_rebuild_ms_one_stream"):
    print(k, json.dumps(e[k]))

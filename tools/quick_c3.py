#!/usr/bin/env python3
"""bench.py's extras leg only (dev aid): prints the configs[2] entries (benchlib/extras.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "stark-perpetual_amd"), os.path.join(ROOT, "tests")]
import torch
from benchlib.extras import extras
from starkperp import _lib
lib = _lib.ensure_init(0, 26)
e = extras(torch, lib, _lib, torch.device("cuda", 0), torch.cuda.current_stream().cuda_stream)
for k in ("c3_4096_orders_host_inclusive_seconds", "c3_4096_orders_numpy_entry_points_seconds", "c3_4096_orders_one_call_seconds", "state_update_2048_positions_4096_orders_seconds", "single_tree_rebuild_ms_one_stream", "c3"):
    print(k, json.dumps(e[k]))

"""iteration counts of pllhip_eval_optimize_branches over two partitions: device loop / host loop, sharded / not"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pllhip_ctypes as pc
from test_eval_driver import build

lib = pc.PllLib(pc.PRODUCT_LIB)
for shards in (0, 3):
    for dev in ("1", "0"):
        os.environ["PLLHIP_EVAL_DEVICE_NEWTON"] = dev
        assert lib.lib.pllhip_set_sharding(shards, None)
        with build(lib, ntips=12, sizes=(700, 300)) as ev:
            if shards:
                ev.attach_comm(None)
            lnl = ev.loglh()
            opt = ev.optimize_branches(1e-4, 10.0, 0.01, 4, -1)
            print(f"shards {shards} device_newton {dev}: lnl {lnl!r} opt {opt!r} iterations {ev.newton_iterations()} scans {ev.counters()[2]}"
                  f" launches {[p.counters().derivative_calls for p in ev.parts]}", flush=True)
        lib.lib.pllhip_set_sharding(0, None)

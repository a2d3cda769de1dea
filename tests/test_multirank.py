"""N>1 path on CPU: two gloo ranks, sites sharded, results combined only through
the reference's reduce-callback interface."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("states", [4, 20])
def test_two_ranks_match_single_rank(oracle, tmp_path, states):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_multirank_worker.py"), str(states), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    # every rank holds the same reduced values (all-reduce semantics, in place)
    for k in ("lnl", "df", "ddf", "max", "min", "wsum"):
        assert r0[k] == r1[k], k
    assert abs(r0["lnl"] - r0["full_lnl"]) < 1e-9 * abs(r0["full_lnl"])
    assert abs(r0["df"] - r0["full_df"]) < 1e-9 * max(1.0, abs(r0["full_df"]))
    assert abs(r0["ddf"] - r0["full_ddf"]) < 1e-9 * max(1.0, abs(r0["full_ddf"]))
    assert r0["max"] == [2.0, 0.0] and r0["min"] == [1.0]
    assert r0["wsum"] == r0["full_wsum"]      # pattern_weight_sum is per slice (treeinfo.c:1166)

"""Multi-rank run of the real sharded path on the box's one GPU: `world` fresh child processes (torch.distributed.run,
gloo) each run tests/_sharded_rank_child.py — the product's ShardedHybrid over the C ABI's sharded search, exchanges
carried by the hosted transport — and compare their own results bit for bit with the CPU oracle.

The file name sorts first on purpose: the children must be started before this process has initialised the GPU
(starting another program from a process that holds the device is refused on the GPU pool), and pytest runs the
modules in name order."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _gpu_initialised_here():
    try:
        return any("kfd" in os.readlink(f"/proc/self/fd/{fd}") for fd in os.listdir("/proc/self/fd"))
    except OSError:
        return False


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_every_rank_of_the_sharded_search_matches_the_oracle(tmp_path, world):
    if _gpu_initialised_here():
        pytest.skip("this process already holds the GPU; run this module first (it sorts first by name)")
    import oracle as orc
    orc.build()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "_sharded_rank_child.py"), str(tmp_path)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    for r in range(world):
        rep = json.load(open(tmp_path / f"rank{r}.json"))
        assert rep["ok"], rep
        assert [c[0] for c in rep["checks"]] == ["all_gather", "all_to_all", "weak", "weak_in_flight", "strong", "strong_tiny"]
        assert rep["scan_fallbacks"] * 5 <= rep["queries_scanned"], rep  # the filter decides, not the exact rescan
    owned = [json.load(open(tmp_path / f"rank{r}.json"))["lists_owned"] for r in range(world)]
    assert sum(owned) == 24 and min(owned) > 0

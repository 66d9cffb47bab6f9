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
        assert [c[0] for c in rep["checks"]] == ["all_gather", "all_to_all", "weak", "weak_in_flight", "strong", "strong_tiny",
                                                   "weak_after_migration", "strong_after_migration"]
        assert rep["scan_fallbacks"] * 5 <= rep["queries_scanned"], rep  # the filter decides, not the exact rescan
    owned = [json.load(open(tmp_path / f"rank{r}.json"))["lists_owned"] for r in range(world)]
    assert sum(owned) == 24 and min(owned) > 0


def _bench(args, timeout=900):
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--n-vectors", "40000", "--nlist", "64", "--steps", "4", "--warmup", "1",
           "--query-batches", "2", "--select-batches", "1", "--train-sample", "10000", "--batch", "128", "--nprobe", "16", "--ef", "50",
           "--no-cpu-baseline", "--compare-host-walk", "0"] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)


def test_bench_gpus_n_launches_itself_from_a_plain_shell():
    # the driver's call: `python bench.py --gpus N ...` with no torchrun around it.  The parent starts the ranks as a
    # fresh child before touching the GPU and relays exactly one JSON line.  Two ranks share the box's one GPU here, so
    # the transport is the hosted one and has to be allowed explicitly.
    if _gpu_initialised_here():
        pytest.skip("this process already holds the GPU; run this module first (it sorts first by name)")
    p = _bench(["--gpus", "2", "--transport", "hosted", "--allow-hosted"])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[:2000]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["transport"] == "hosted" and j["value"] > 0
    assert j["config"]["global_batch"] == 2 * 128 and j["scaling"] == "weak"


def test_bench_refuses_a_multi_gpu_line_without_rccl():
    # RCCL cannot place two ranks on one device: bring-up fails on every rank, and without --allow-hosted the run must
    # exit non-zero and print no JSON line (an n_gpus > 1 line can only come from RCCL)
    if _gpu_initialised_here():
        pytest.skip("this process already holds the GPU; run this module first (it sorts first by name)")
    p = _bench(["--gpus", "2"], timeout=600)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")], p.stdout[:2000]


def test_dev_build_loopback_communicator_runs_a_rank_of_a_larger_job():
    # the capacity-planning communicator is compiled into the dev build only (make dev -> lib_dev/, -DFVDB_DEV_TOOLS):
    # a child process loads that build through FVDB_LIB_DIR and runs rank 1 of a pretended 4-rank job
    if _gpu_initialised_here():
        pytest.skip("this process already holds the GPU; run this module first (it sorts first by name)")
    root = os.path.dirname(HERE)
    if not os.path.exists(os.path.join(root, "fabstir-vectordb_amd", "lib_dev", "libfvdb_hip.so")):
        pytest.skip("dev build absent (make -C fabstir-vectordb_amd dev)")
    env = dict(os.environ, FVDB_LIB_DIR="lib_dev")
    p = subprocess.run([sys.executable, os.path.join(HERE, "_loopback_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "loopback child ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]

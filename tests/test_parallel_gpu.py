"""Data parallelism on the GPU path: two fresh processes sharing GPU 0 over gloo run the solver's
data-parallel train steps (tests/dp_worker.py) -- BASELINE.json configs[2]'s code path on the one GPU
a test box has; the RCCL run on 2/4/8 GPUs is bench.py's."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_one_gpu_solver_equivalence(dev):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FCD_HANG_DUMP="150")
    env.pop("RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "dp_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "weights identical across ranks: True" in r.stdout


def test_bench_starts_its_own_ranks(dev):
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two ranks itself (here over gloo, both on
    GPU 0) and report n_gpus 2."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--quick"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["world_size"] == 2 and res["config"]["global_batch"] == 32
    assert res["value"] > 0
    # the fields that make the first real multi-GPU run diagnose itself: span layout, the exposed (un-hidden) wait for
    # the gradient all-reduce per network, and the same loop with the sends from inside backward switched off
    cfg = res["config"]
    assert set(cfg["spans"]) == {"D", "G"} and sum(cfg["spans"]["D"]["spans_mb"]) > 170 and sum(cfg["spans"]["G"]["spans_mb"]) > 32
    assert max(cfg["spans"]["D"]["spans_mb"]) <= 135 and cfg["spans"]["D"]["piece_mb"] == 64.0
    assert set(cfg["allreduce_exposed_ms"]) == {"D", "G"} and all(v >= 0 for v in cfg["allreduce_exposed_ms"].values())
    for mode in ("on", "off"):
        assert cfg["bucket_overlap"][mode]["ms_per_step"] > 0
        assert abs(cfg["bucket_overlap"][mode]["allreduce_mbytes"]["D"] - 178.9) < 1.0
    assert cfg["spans"]["D"]["sent_from_inside_backward_last_step"] >= 1          # (described after the timed region: overlap on)

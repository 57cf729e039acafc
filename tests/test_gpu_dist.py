"""The multi-process path at HEAD on the one GPU of the test box: two ranks (gloo; the launcher starts before
any GPU call), wave engine + batched plans, Plan.estep -> all-reduce -> Plan.mstep for pooled channels and for
time shards with certified edges; and bench.py's multi-GPU modes (--config 4 --pooled, --time-sharded) end
to end with `valid: true`.  The 8-GPU runs are the driver's; RCCL needs one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_args, port, extra_env=None, timeout=560):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HMMSORT_BENCH_ONE_GPU="1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_pooled_and_time_sharded_statistics():
    p = _torchrun([os.path.join(ROOT, "tests", "_dist_gpu_worker.py")], 29741)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_GPU_OK world=2" in p.stdout


def _bench_line(p):
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("mode", ["config4_pooled", "config4", "config5", "time_sharded"])
def test_bench_multi_gpu_modes_two_ranks(mode):
    common = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--quick"]
    if mode.startswith("config4"):
        args = ["--config", "4", "--total-channels", "4", "--batch", "2", "--samples", "400000"]
        if mode.endswith("pooled"):
            args.append("--pooled")
    elif mode == "config5":
        args = ["--config", "5", "--total-channels", "2", "--samples", "1000000"]
    else:
        args = ["--time-sharded", "--samples", "2000000"]
    res = _bench_line(_torchrun([os.path.join(ROOT, "bench.py")] + common + args, 29742 + len(mode)))
    assert res["valid"] is True and res["n_gpus"] == 2
    assert res["value"] > 0 and res["ms_per_step"] > 0
    if mode.startswith("config"):
        assert res["scaling"] == "strong" and res["roofline"]["kernel"].startswith("kw_")
        assert res["config"]["pooled_allreduce"] == mode.endswith("pooled")
    else:
        assert res["scaling"] == "strong" and res["config"]["time_sharded"]

"""N > 1 path on CPU: world_size-2 gloo run of the channel sharding + statistics all-reduce."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "DIST_OK world=2" in p.stdout

"""world_size-2 rehearsal (gloo, CPU) of bench.py's multi-rank control flow: every rank simulates its own shard of reads,
ranks meet at a barrier, the job time is the MAX over ranks and rank 0 alone reports the whole-job rate; then the hand-off of
the end-to-end leg (rank 0 runs the command line tool with --gpus 2, here the oracle-bound host mapper, rank 1 parks in a
CPU-group barrier) whose mapped reads/s is the line's `value`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_selftest():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--dist-selftest"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert abs(out["elapsed"] - 0.5) < 1e-9                     # MAX over ranks (rank 1 reported 0.5 s)
    assert abs(out["replay_reads_per_s"] - 64 * 3 * 2 / 0.5) < 1e-6          # reads of ALL ranks / slowest rank
    # the headline at N > 1: rank 0 ran the command line tool with --gpus 2 while rank 1 waited; `value` is that figure
    assert out["mapped_reads_per_s"] and out["value"] == out["mapped_reads_per_s"]
    assert out["sam_md5"] == out["golden_md5"]
    assert len(set(out["shard_crc"])) == 2                      # the two ranks mapped different reads

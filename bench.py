#!/usr/bin/env python3
"""bench.py -- hot-path throughput of libalfa_hot on MI355X (see DESIGN.md "Measurement").

One *step* = one pass of the seed-and-extend hot path over one resident batch of synthetic reads:
  SMAM seed retrieval for both strands of every read (smam_count_kernel -> scan -> smam_emit_kernel)
  + the banded-DP tasks of those reads (ksw_nw, ksw_extend, ksw_global kernels).
All inputs of all steps are packed and copied to HBM before the timed region; the timed region only
launches kernels (alfa_seed_plan_launch / alfa_dp_batch_launch) and is bracketed by barrier +
torch.cuda.synchronize() + alfa_sync() on both sides.

Workload = BASELINE.json configs[2] ("synthetic 5 Mbp random reference, 1M simulated 250 bp reads at
2% error, 1 GPU"): the configuration the metric is quoted on that fits one GPU.  configs[0]/[1] are the
48 kbp lambda-phage walkthrough files (parity cases, not bench lines).  `--config c4` runs configs[3]
(100 Mbp reference, 2-10 kbp reads at 10 % error, `-e 0.12`) through the same steps; the driver's default is c3.

What the line says (one reader, one line): `value` = hot-path replay rate (kernels only, inputs resident);
`mapped_reads_per_s` = the SURVEY 8d figure, FASTQ in -> SAM out through the command line tool; `pcie_inclusive_reads_per_s`
= create + launch + fetch of one step through the C ABI (host packing, H2D, kernels, D2H, unpacking); `roofline` = the
seed kernel against HBM; `roofline_dp` = the DP kernels against the int32 VALU peak; `cpu_baseline` = the compiled
reference on this box's usable host cores (count stated), run at the same thread count as the GPU tool.

DP tasks are the real ones: before the timed region the product command line tool (alfalfa_amd align = staged host
mapper + HIP hot path, byte-identical SAM to the reference) maps each step's reads once with ALFA_TASK_LOG set and
every ksw_extend / ksw_nw / ksw_global task it issued is loaded, packed and made resident (mapper_dp_tasks()).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# hardware queues of the HIP runtime: the host chooses (include/alfa_hot.h: alfa_hw_queues_defaulted), and bench.py is the host
# here.  Its replay legs overlap the DP chains of several resident batches, which wants 16 queues (C4 step 10.1 ms with the
# box's default of 4, 6.6-7.9 with 16), so this process asks for ALFA_HW_QUEUES (default 16) before torch or the library
# initialise HIP.  The command line tools it starts get the environment's ORIGINAL value back (run_cli): the product runs as
# an operator would run it.  Both values are printed in the line (`hw_queues`, `hw_queues_cli`).
_ORIG_HWQ = os.environ.get("GPU_MAX_HW_QUEUES")
if os.environ.get("ALFA_HW_QUEUES", "16") != "0":
    os.environ["GPU_MAX_HW_QUEUES"] = str(min(32, max(1, int(os.environ.get("ALFA_HW_QUEUES", "16")))))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 256 CU x 4 SIMD32 x 2.4 GHz = 78.6 T int32 lane-ops/s

ACGT = np.frombuffer(b"acgt", dtype=np.uint8)
_RC = bytes.maketrans(b"acgtn", b"tgcan")
NT4 = np.full(256, 4, np.uint8)
for _i, _c in enumerate(b"acgt"):
    NT4[_c] = _i
    NT4[_c - 32] = _i


def make_reference(n, seed):
    rng = np.random.default_rng(seed)
    return ACGT[rng.integers(0, 4, n)]


def simulate_batch(rng, ref, n, rlen, err):
    """n reads: a window of rlen reference bases with per-base error err (60/20/20 sub/ins/del), 50/50 strand.
    Returns (list of read bytes as sequenced, list of their reverse complements, true positions, strands)."""
    L = len(ref)
    pos = rng.integers(0, L - rlen - 64, n)
    win = ref[pos[:, None] + np.arange(rlen)[None, :]]
    r = rng.random((n, rlen))
    sub = r < 0.6 * err
    ins = (r >= 0.6 * err) & (r < 0.8 * err)
    dele = (r >= 0.8 * err) & (r < err)
    win[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
    reps = (1 + ins.astype(np.int8) - dele.astype(np.int8)).astype(np.int64)
    flat = np.repeat(win.reshape(-1), reps.reshape(-1))
    ends = np.cumsum(reps.reshape(-1))
    ipos = ends[ins.reshape(-1)] - 1
    flat[ipos] = ACGT[rng.integers(0, 4, len(ipos))]
    row_end = ends.reshape(n, rlen)[:, -1]
    row_beg = np.concatenate([[0], row_end[:-1]])
    buf = flat.tobytes()
    strand = rng.random(n) < 0.5
    fw, rc = [], []
    for i in range(n):
        s = buf[row_beg[i]:row_end[i]]
        t = s[::-1].translate(_RC)
        if strand[i]:
            fw.append(s); rc.append(t)
        else:
            fw.append(t); rc.append(s)
    return fw, rc, pos, strand


TASK_DTYPE = np.dtype([("query", "<u8"), ("target", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("w", "<i4"), ("h0", "<i4")])


def usable_cores(world=1):
    """host cores the command line tools may use: the affinity mask, capped by the box's per-GPU share (OMP_NUM_THREADS when the
    box sets it: a GPU box shows the whole machine in os.cpu_count() but gives a one-GPU job a share of it) times the GPUs of
    the job.  torchrun exports OMP_NUM_THREADS=1 to its ranks when the variable was unset: that is not a share."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    if share <= 1:
        share = 16 if world > 1 else n
    return max(1, min(n, share * max(1, world)))


def write_fastq(path, reads, mode="w"):
    with open(path, mode + "b") as f:
        for lo in range(0, len(reads), 50000):
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (lo + i, r, b"I" * len(r)) for i, r in enumerate(reads[lo:lo + 50000])))


class GpuBusy:
    """mean / max of the kernel driver's gpu_busy_percent per card while a command runs (sampled every 50 ms)"""
    def __init__(self):
        import glob
        self.files = sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))
        self.samples = {f: [] for f in self.files}
        self._stop = False

    def _loop(self):
        while not self._stop:
            for f in self.files:
                try:
                    self.samples[f].append(float(open(f).read().strip()))
                except Exception:
                    pass
            time.sleep(0.05)

    def __enter__(self):
        import threading
        self._t = threading.Thread(target=self._loop, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self._t.join()

    def summary(self):
        out = {}
        for f, v in self.samples.items():
            if v and max(v) > 0:
                out[f.split("/")[4]] = {"mean": round(sum(v) / len(v), 1), "max": max(v), "samples": len(v)}
        return out or None


def run_cli(exe, index_prefix, fq, out, threads, extra=(), gpus=None, stats=False):
    """one `align` run of a command line tool: (wall seconds, return code, stderr text, per-GPU busy)"""
    import subprocess
    cmd = [exe, "align", "-i", index_prefix, "-0", fq, "-o", out, "-t", str(threads)] + list(extra)
    if gpus is not None:
        cmd += ["--gpus", str(gpus)]
    if stats:
        cmd += ["--stats"]
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    if _ORIG_HWQ is not None:
        env["GPU_MAX_HW_QUEUES"] = _ORIG_HWQ
    os.sync()          # what this script wrote just before (the FASTQ, the C4 block's files) leaves the page cache's dirty list first
    with GpuBusy() as gb:
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.perf_counter() - t0
    return dt, pr.returncode, pr.stderr.decode(errors="replace"), gb.summary()


def sam_record_md5(path, n=None):
    """md5 over the first n alignment records (all when n is None); header lines excluded.  With the default -a 1 a read
    has exactly one record and output is in input order, so the first n records belong to the first n reads."""
    import hashlib
    h = hashlib.md5()
    k = 0
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b"@"):
                continue
            if n is not None and k >= n:
                break
            h.update(line)
            k += 1
    return h.hexdigest(), k


def simulate_long(rng, ref_bytes, n, lo, hi, err):
    """n reads of length U[lo,hi] with per-base error err (60/20/20), 50/50 strand: (as sequenced, reverse complements)"""
    import seqgen
    fw = seqgen.sim_reads(rng, ref_bytes, n, lo, hi, err)
    return fw, [seqgen.revcomp(r) for r in fw]


def mapper_dp_tasks(workdir, index_prefix, reads, threads=8, device=0, extra=()):
    """The DP task stream the staged host mapper really issues for `reads`: run the product command line tool
    (alfalfa_amd align, HIP hot path) once with ALFA_TASK_LOG set and load every ksw_extend / ksw_nw / ksw_global
    task it submitted.  Returns {kind: (ctypes DpTask array, keepalive)} for kind in extend/nw/global."""
    import glob
    import subprocess
    from alfalfa_amd import ffi
    fq = os.path.join(workdir, "step.fq")
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write("@r%d\n%s\n+\n%s\n" % (i, r.decode(), "I" * len(r)))
    prefix = os.path.join(workdir, "tasks")
    for old in glob.glob(prefix + ".*"):
        os.remove(old)
    exe = os.path.join(ROOT, "alfalfa_amd", "alfalfa_amd")
    env = dict(os.environ, ALFA_TASK_LOG=prefix)
    # --host-trials: the trial loop on the host, so that every ksw call goes through the batch entry points and is logged (by
    # default whole reads are mapped inside one kernel, alfa_map_reads_batch, and only ksw_global calls would show up)
    subprocess.run([exe, "align", "-i", index_prefix, "-0", fq, "-o", os.path.join(workdir, "step.sam"), "-t", str(threads),
                    "--device", str(device), "--host-trials"] + list(extra),
                   check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return load_task_log(prefix)


def load_task_log(prefix):
    """every DP task an `alfalfa_amd align` run with ALFA_TASK_LOG=prefix submitted: {kind: (ctypes DpTask array, keepalive)}"""
    import glob
    from alfalfa_amd import ffi
    hdrs, seqs = [], []
    for hf in sorted(glob.glob(prefix + ".*.hdr")):
        hdrs.append(np.fromfile(hf, np.int32).reshape(-1, 5))
        seqs.append(np.fromfile(hf[:-4] + ".seq", np.uint8))
    hdr = np.concatenate(hdrs)
    seq = np.concatenate(seqs)
    off = np.concatenate([[0], np.cumsum(hdr[:, 1].astype(np.int64) + hdr[:, 2])[:-1]])
    base = seq.ctypes.data
    out = {}
    for name, kind in (("nw", 0), ("global", 1), ("extend", 2)):
        rows = np.nonzero(hdr[:, 0] == kind)[0]
        arr = np.zeros(len(rows), dtype=TASK_DTYPE)
        arr["query"] = base + off[rows]
        arr["target"] = base + off[rows] + hdr[rows, 1]
        arr["qlen"], arr["tlen"], arr["w"], arr["h0"] = hdr[rows, 1], hdr[rows, 2], hdr[rows, 3], hdr[rows, 4]
        tasks = (ffi.DpTask * len(rows)).from_buffer(arr) if len(rows) else (ffi.DpTask * 0)()
        out[name] = (tasks, (arr, seq))
    return out


# ---- multi-GPU plumbing (reads sharded over ranks, index replicated, no data-path collective)
def rank_rng(rank):
    """every rank simulates its own shard of reads"""
    return np.random.default_rng(1000 + rank)


def max_over_ranks(elapsed, world, device):
    """the job's time is the slowest rank's"""
    if world <= 1:
        return elapsed
    import torch
    import torch.distributed as dist
    tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def whole_job_value(batch, steps, world, elapsed):
    """reads all ranks processed / the slowest rank's time"""
    return batch * steps * world / elapsed


def dist_selftest(args, rank, world):
    """CPU rehearsal of the N>1 control flow (tests/test_bench_dist.py, gloo): shard -> barrier -> MAX -> rank 0 reports; then
    the hand-off around the end-to-end leg: the other ranks park in a CPU-group barrier while rank 0 runs the command line
    tool with --gpus N (here the host mapper bound to the oracle, tests/_cpu_mapper, on the lambda example: N worker groups)
    and reports mapped_reads_per_s as `value`."""
    import zlib
    import torch
    import torch.distributed as dist
    cpu_group = dist.new_group(backend="gloo")
    ref = make_reference(200000, 1)
    fw, _, _, _ = simulate_batch(rank_rng(rank), ref, 64, 100, 0.02)
    crc = zlib.crc32(b"".join(fw))
    dist.barrier()
    elapsed = max_over_ranks(0.25 * (rank + 1), world, "cpu")
    sums = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sums, torch.tensor([crc], dtype=torch.int64))
    if rank != 0:
        dist.barrier(group=cpu_group)
        dist.barrier(group=cpu_group)
        dist.destroy_process_group()
        return
    import golden_util as G
    import mapper_util as M
    dist.barrier(group=cpu_group)
    d = M.workdir()
    pre = os.path.join(d, "selftest_idx")
    import subprocess
    subprocess.run([M.CPU_EXE, "index", "-r", "reference.fasta", "-p", pre, "-s", "12"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    outp = os.path.join(d, "selftest.sam")
    dt, rcode, _err, _busy = run_cli(M.CPU_EXE, pre, os.path.join(d, "454reads.fq"), outp, 2 * world, [], gpus=world)
    dist.barrier(group=cpu_group)
    mapped = 100 / dt if rcode == 0 else None
    print(json.dumps({"n_gpus": world, "elapsed": elapsed, "replay_reads_per_s": whole_job_value(64, args.steps, world, elapsed),
                      "value": mapped, "mapped_reads_per_s": mapped, "sam_md5": M.md5(M.sam_body(outp)), "golden_md5": M.golden_md5("454_s12"),
                      "scaling": "weak", "shard_crc": [int(x.item()) for x in sums]}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=["c3", "c4"], help="BASELINE configs[2] (default) or configs[3]")
    ap.add_argument("--batch", type=int, default=0, help="reads per step per GPU (default 100000 for c3, 4000 for c4)")
    ap.add_argument("--ref-len", type=int, default=0)
    ap.add_argument("--read-len", type=int, default=250)
    ap.add_argument("--err", type=float, default=0.02)
    ap.add_argument("--sparseness", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=20000)
    ap.add_argument("--e2e-reads", type=int, default=-1, help="reads of the end-to-end (headline) run: default steps x batch x GPUs; 0 = skip, value is then the replay figure")
    ap.add_argument("--c4-e2e-reads", type=int, default=500000, help="reads of the C4 block's end-to-end run (default line, N = 1): what configs[3] names")
    ap.add_argument("--cpu-e2e-reads", type=int, default=-1, help="reads the compiled reference maps for cpu_baseline (a prefix of the same FASTQ; default min(e2e reads, 1000000 for c3 / 20000 for c4))")
    ap.add_argument("--no-c4", action="store_true", help="default (c3) run: do not append the C4 block (a second bench.py run with --config c4)")
    ap.add_argument("--e2e-threads", type=int, default=0, help="threads of both command line tools (default: min(usable host cores, 214))")
    ap.add_argument("--align-tasks", type=int, default=50000, help="mate-rescue shaped ksw_align tasks for the informative ksw_align section")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for --dist-selftest)")
    ap.add_argument("--dist-selftest", action="store_true", help="CPU rehearsal of the multi-rank control flow, no GPU work")
    args = ap.parse_args()
    c4 = args.config == "c4"
    args.batch = args.batch or (4000 if c4 else 100000)
    args.ref_len = args.ref_len or (100000000 if c4 else 5000000)
    if c4:
        args.err = 0.10
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.e2e_reads < 0:
        args.e2e_reads = args.steps * args.batch * world        # the K timed steps of the headline: K batches per GPU, FASTQ -> SAM
    if args.cpu_e2e_reads < 0:
        args.cpu_e2e_reads = min(args.e2e_reads, 20000 if c4 else 1000000)
    cores = usable_cores(world)
    if args.e2e_threads <= 0:
        args.e2e_threads = min(cores, 214)       # reference: bpCount = threads * 10^7 is an int (src/main.cpp:116)
    cli_extra = ["-e", "0.12"] if c4 else []     # the default 0.08 sends most 10 %-error reads through rescue

    # ---- the C4 block (BASELINE configs[3], the configuration the 1/2/4/8 metric is quoted on) of the default line: at N = 1
    # the whole line of a second run of this script with --config c4 (replay, rooflines, end to end, cpu_baseline), made BEFORE
    # this process touches the GPU (two processes with 16 hardware queues each would time-slice); at N > 1 the end-to-end
    # figure on N GPUs (c4_end_to_end, after the C3 legs)
    c4_block = None
    if not c4 and not args.no_c4 and world == 1 and not args.dist_selftest:
        import subprocess
        try:
            # (end to end on the 500 000 reads configs[3] names: the run's fixed part -- index load, HIP start-up, teardown,
            # 0.35 s -- is a third of a 100 000-read run)
            cmd = [sys.executable, os.path.abspath(__file__), "--config", "c4", "--steps", "25", "--warmup", "2", "--align-tasks", "0",
                   "--e2e-reads", str(args.c4_e2e_reads)]
            if args.no_cpu_baseline:
                cmd.append("--no-cpu-baseline")
            cenv = dict(os.environ)
            cenv.pop("GPU_MAX_HW_QUEUES", None)
            if _ORIG_HWQ is not None:
                cenv["GPU_MAX_HW_QUEUES"] = _ORIG_HWQ          # the child makes the same choice from the same starting point
            pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=cenv)
            line = [l for l in pr.stdout.decode().split("\n") if l.startswith("{")]
            c4_block = json.loads(line[-1]) if line else {"error": pr.stderr.decode()[-500:]}
        except Exception as e:   # the C4 block never fails the C3 line
            c4_block = {"error": repr(e)}
    import torch
    import torch.distributed as dist
    if args.dist_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        return dist_selftest(args, rank, world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    # hand-off barriers around the end-to-end leg go through a CPU (gloo) group: a rank parked in an RCCL barrier spins in a
    # kernel on its GPU, which would sit on the GPUs the command line tool is being measured on
    cpu_group = dist.new_group(backend="gloo") if world > 1 else None
    dev = local_rank if world > 1 else 0
    if os.environ.get("ALFA_ALIAS_DEVICES") and torch.cuda.is_available():
        dev %= torch.cuda.device_count()         # rehearsal of N ranks on fewer cards (tests/test_gpu_mapper.py; use --backend gloo)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libalfa_hot has no CPU fallback")
    torch.cuda.set_device(dev)

    from alfalfa_amd import ffi
    import seqgen
    K = args.sparseness
    ref = make_reference(args.ref_len, 2 if c4 else 1)        # PRNG seed 1 for C3, 2 for C4
    ref_bytes = ref.tobytes()
    ref_nt4 = NT4[ref]
    t0 = time.time()
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref_bytes], K=K)
    didx = ffi.DeviceIndex(hidx, dev)
    t_index = time.time() - t0
    import tempfile
    workdir = tempfile.mkdtemp(prefix="alfa_bench_r%d_" % rank)
    index_prefix = os.path.join(workdir, "idx")
    hidx.save(index_prefix)

    err_percent = 0.13 if c4 else 0.08           # reference default 0.08 (src/options.cpp:52); -e X stores X + 0.01 (:296)
    min_len, sparse_q = seqgen.seed_params(K, args.read_len, err_percent)

    def seed_queries(reads):
        if not c4:
            return ffi.make_seed_queries(reads, min_len, 10000, sparse_q)
        prm = [seqgen.seed_params(K, len(r), err_percent) for r in reads]     # min_len grows with the read length (src/mapper.cpp:1226)
        return ffi.make_seed_queries(reads, [a for a, _ in prm], 10000, [b for _, b in prm])
    P = ffi.default_params()
    nsteps = args.warmup + args.steps
    rng = rank_rng(rank)
    steps = []
    sample = None
    for s in range(nsteps):
        if c4:
            fw, rc = simulate_long(rng, ref_bytes, args.batch, 2000, 10000, args.err)
        else:
            fw, rc, pos, strand = simulate_batch(rng, ref, args.batch, args.read_len, args.err)
        queries = seed_queries(fw + rc)
        plan = ffi.SeedPlan(didx, ffi.SEED_SMAM, queries)
        tk = mapper_dp_tasks(workdir, index_prefix, fw, threads=8, device=dev, extra=cli_extra)
        (te, ke), (tn, kn), (tg, kg) = tk["extend"], tk["nw"], tk["global"]
        b_ext = ffi.DpBatch(ffi.DP_EXTEND, P, te, dev)
        b_nw = ffi.DpBatch(ffi.DP_NW, P, tn, dev)
        b_gl = ffi.DpBatch(ffi.DP_GLOBAL, P, tg, dev)
        steps.append((plan, b_ext, b_nw, b_gl))
        if s == args.warmup:                      # the first TIMED step: its results are the ones checked against the oracle
            sample = (fw, rc, te, tn, tg, (ke, kn, kg))
        else:
            del fw, rc

    def run(step):
        plan, b_ext, b_nw, b_gl = step
        plan.launch(); b_nw.launch(); b_ext.launch(); b_gl.launch()

    def barrier():
        if world > 1:
            dist.barrier()

    ffi.set_lazy_join(dev, True)      # successive DP batches overlap (their kernel bins are independent chains)
    for s in range(args.warmup):
        run(steps[s])
    ffi.sync(dev)
    torch.cuda.synchronize()
    barrier()
    ffi.sync(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ffi.timer_start(dev)
    for s in range(args.warmup, nsteps):
        run(steps[s])
    t_enqueue = time.perf_counter() - t0      # host time to issue every launch of the timed steps (no waiting)
    dev_ms = ffi.timer_stop(dev)
    ffi.sync(dev)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, "cuda")

    time.sleep(0.02)      # an idle gap in kernel traces between the timed region and the un-overlapped pass below
    timed = steps[args.warmup:]
    seed_ms = [st[0].kernel_ms for st in timed]
    # per-kind DP times (GCUPS): the same resident batches launched once more one at a time, each bracketed by its own
    # events (inside the timed region they overlap, so per-batch intervals would not be attributable)
    ffi.set_lazy_join(dev, False)
    ext_ms, nw_ms, gl_ms, seed_iso_ms = [], [], [], []
    for st in timed:
        st[0].launch()
        ffi.sync(dev)
        seed_iso_ms.append(st[0].kernel_ms)
        for b, acc in ((st[1], ext_ms), (st[2], nw_ms), (st[3], gl_ms)):
            b.launch()
            ffi.sync(dev)
            acc.append(b.kernel_ms)
    # what overlapping the seed kernel with the DP chains buys: the same timed steps once more with one half left out
    def half_loop(fn):
        ffi.sync(dev)
        torch.cuda.synchronize()
        time.sleep(0.005)
        t = time.perf_counter()
        for st in timed:
            fn(st)
        ffi.sync(dev)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / len(timed) * 1e3
    ffi.set_lazy_join(dev, True)
    seed_only_ms = half_loop(lambda st: st[0].launch())
    dp_only_ms = half_loop(lambda st: (st[2].launch(), st[1].launch(), st[3].launch()))
    ffi.sync(dev)
    ffi.set_lazy_join(dev, False)
    cells_ext = sum(st[1].cells for st in timed)
    cells_nw = sum(st[2].cells for st in timed)
    cells_gl = sum(st[3].cells for st in timed)
    queries_per_step = timed[0][0].queries

    # results of one TIMED step are fetched (and a sample of them compared with the oracle below, on rank 0)
    m, off = timed[0][0].fetch()
    seeds_per_read = len(m) / args.batch
    timed_ext = timed[0][1].fetch()
    timed_nw = timed[0][2].fetch()
    timed_gl = timed[0][3].fetch()
    fast_tasks = {"extend": timed[0][1].giveups(), "nw": timed[0][2].giveups(), "global": timed[0][3].giveups()}
    giveups_each = [st[1].giveups()[1] for st in timed]

    # PCIe-inclusive rate: create (host packing + H2D) + launch + fetch (D2H + unpacking) of the first step, through the
    # synchronous halves of the C ABI.  Never `value`.
    fw0, rc0, te0, tn0, tg0, _k0 = sample
    q0 = seed_queries(fw0 + rc0)              # python/ctypes marshalling: outside the measured interval
    tp0 = time.perf_counter()
    pl = ffi.SeedPlan(didx, ffi.SEED_SMAM, q0)
    pl.launch()
    pcie_m, _pcie_off = pl.fetch()
    tp1 = time.perf_counter()
    for mode, tk_ in ((ffi.DP_NW, tn0), (ffi.DP_EXTEND, te0), (ffi.DP_GLOBAL, tg0)):
        bb = ffi.DpBatch(mode, P, tk_, dev)
        bb.launch()
        bb.fetch()
        del bb
    tp2 = time.perf_counter()
    del pl
    pcie_batch_api = {"reads_per_s": args.batch / (tp2 - tp0), "seed_s": tp1 - tp0, "dp_s": tp2 - tp1,
                      "note": "one step through the per-kind entry points, create + launch + fetch each (host packing, H2D, kernels, D2H, "
                              "unpacking; python ctypes marshalling outside); dominated by host packing"}
    # ... and through the whole-read entry point (alfa_map_reads_batch: host reads in, alignments + chains + CIGAR operations
    # out; seeds, regions, the trial loop with its DP and the CIGAR recovery in between, nothing else crossing PCIe): the
    # drop-in boundary a caller of unpairedMatch binds.  Best of 3 calls on the first timed step's reads (the first call
    # also allocates the calling thread's buffers).
    mopt = ffi.default_map_options(err_percent)
    mreads = ffi.make_map_reads(fw0, rc0, K, mopt)
    map_call = {"seconds": None}
    for _rep in range(3):
        tq0 = time.perf_counter()
        m_st, m_aoff, m_al, m_soff, m_sd, m_ooff, m_ops = ffi.map_reads_batch(didx, mopt, P, mreads)
        dtm = time.perf_counter() - tq0
        if map_call["seconds"] is None or ffi.last_map_call_seconds < map_call["seconds"]:
            map_call = {"seconds": ffi.last_map_call_seconds, "seconds_with_numpy_copies": dtm, "map_kernel_ms": ffi.last_timing().kernel_ms}
    map_call.update({"reads": len(fw0), "reads_per_s": len(fw0) / map_call["seconds"], "done_on_device": int((m_st == 0).sum()),
                     "handed_back": int((m_st != 0).sum()), "alignments": int(len(m_al)), "with_cigar": int((m_al["op_count"] >= 0).sum()),
                     "cigar_ops": int(len(m_ops)),
                     "map_kernel_reads_per_s": len(fw0) / (map_call["map_kernel_ms"] * 1e-3) if map_call["map_kernel_ms"] else None,
                     "note": "alfa_map_reads_batch on one step's reads from one host thread, host buffers in, host results out (seconds = the "
                             "library call, what a C caller sees; seconds_with_numpy_copies adds this script's copies of the results); map_kernel_ms = "
                             "the per-read mapping kernel alone (trial loop + ksw_nw / ksw_extend / ksw_global + CIGARs, one wavefront per read)"})
    pcie = {"reads_per_s": map_call["reads_per_s"], "whole_read_call": map_call, "per_kind_batch_calls": pcie_batch_api,
            "note": "reads/s of ONE host thread through the C ABI with host buffers on both sides (packing, H2D, kernels, D2H, unpacking): "
                    "the whole-read call (alfa_map_reads_batch) is the figure; the per-kind batch calls of round 2 beside it"}

    if rank != 0:
        # the end-to-end leg runs the command line tool on every GPU of the job (rank 0 starts it): give the memory back, wait
        del steps, timed, sample
        import gc
        gc.collect()
        dist.barrier(group=cpu_group)
        dist.barrier(group=cpu_group)
        dist.destroy_process_group()
        return

    # ---- algorithmic bytes of the dominant kernel (SURVEY 8d census, counted by the oracle on a sample)
    import oracle_ffi as O
    fw, rc, te, tn, tg, _keep = sample
    ns = min(2000, len(fw))
    O.orc_counters_reset()
    qs = seed_queries(fw[:ns] + rc[:ns])
    O.orc_seed_batch(hidx.view, ffi.SEED_SMAM, qs)
    c = O.orc_counters()
    bytes_per_read = c["sa_bytes"] / ns
    bytes_per_launch = bytes_per_read * args.batch
    seed_avg_ms = float(np.mean(seed_ms))
    achieved = bytes_per_launch / (seed_avg_ms * 1e-3) / 1e9

    # HBM-side bytes per dispatch of that kernel from the committed rocprofv3 --pmc FETCH_SIZE pass (profiles/),
    # valid only for the batch size it was collected at; null otherwise
    traffic = None
    try:
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                      if f.endswith("_pmc_traffic.json") and (f.endswith("_c4_pmc_traffic.json") == c4))
        if prof:
            pj = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
            if pj.get("batch") == args.batch and (c4 or (args.ref_len == 5000000 and args.read_len == 250)):
                bd = pj["bytes_per_dispatch"]
                traffic = bd.get("smam_seed_path", bd.get("smam_count_kernel"))
    except Exception:
        traffic = None

    cpu = None
    executed_ratio = None
    # ---- a sample of the TIMED step's DP results against the oracle (the timed launches are the ones checked)
    nchk = 3000
    chk = {}
    if len(te) and len(tn) and len(tg):
        ke, kn, kg = min(nchk, len(te)), min(nchk, len(tn)), min(nchk, len(tg))
        chk["extend"] = bool(np.array_equal(timed_ext[:ke], O.orc_ksw_extend_batch(P, (ffi.DpTask * ke).from_address(C_addr(te)))))
        chk["nw"] = bool(np.array_equal(timed_nw[:kn], O.orc_ksw_nw_batch(P, (ffi.DpTask * kn).from_address(C_addr(tn)))))
        so, co, no_, oo = O.orc_ksw_global_batch(P, (ffi.DpTask * kg).from_address(C_addr(tg)))
        sg, cg, ng, og = timed_gl
        okg = np.array_equal(sg[:kg], so) and np.array_equal(ng[:kg], no_)
        for i in range(kg if okg else 0):
            if ng[i] > 0 and not np.array_equal(cg[og[i]:og[i] + ng[i]], co[oo[i]:oo[i] + no_[i]]):
                okg = False
                break
        chk["global"] = bool(okg)
    mo, oo_ = O.orc_seed_batch(hidx.view, ffi.SEED_SMAM, seed_queries(fw[:500] + rc[:500])) if not c4 else (None, None)
    if mo is not None:
        # the plan seeds fw[0..n) then rc[0..n): reads 0..499 are the first 500 queries
        chk["seeds_fw500"] = bool(np.array_equal(off[:501], oo_[:501]) and np.array_equal(m[:int(off[500])], mo[:int(oo_[500])]))
    if not args.no_cpu_baseline and world == 1:      # the CPU legs run at N=1 only
        # the oracle (scalar port) on one host core over a bounded sample of the same workload
        nsamp = min(args.cpu_sample, len(fw))
        frac = nsamp / len(fw)
        qs = seed_queries(fw[:nsamp] + rc[:nsamp])
        ne, nn, ng = int(len(te) * frac), int(len(tn) * frac), int(len(tg) * frac)
        tc0 = time.perf_counter()
        O.orc_seed_batch(hidx.view, ffi.SEED_SMAM, qs)
        tseed = time.perf_counter() - tc0
        O.orc_counters_reset()
        O.orc_ksw_extend_batch(P, (ffi.DpTask * ne).from_address(C_addr(te)))
        O.orc_ksw_nw_batch(P, (ffi.DpTask * nn).from_address(C_addr(tn)))
        O.orc_ksw_global_batch(P, (ffi.DpTask * ng).from_address(C_addr(tg)))
        tcpu = time.perf_counter() - tc0
        cc = O.orc_counters()
        cpu_cells = cc["cells_extend"] + cc["cells_nw"] + cc["cells_global"]
        # ksw_extend cells: SURVEY 8d counts the inner-loop iterations the reference executes; the GCUPS below divide the
        # static band envelope (what the kernels compute).  Their ratio on this sample:
        env_ext = ffi.DpBatch(ffi.DP_EXTEND, P, (ffi.DpTask * ne).from_address(C_addr(te)), dev).cells if ne else 0
        executed_ratio = cc["cells_extend"] / env_ext if env_ext else None
        cpu = {"value": nsamp / tcpu, "unit": "reads/s", "cores": 1, "kind": "port",
               "sample": "%d reads (both strands seeded + their DP tasks) of step 0, oracle/liboracle.so, 1 thread" % nsamp,
               "seed_reads_per_s": nsamp / tseed, "dp_gcups": cpu_cells / max(tcpu - tseed, 1e-9) / 1e9}

    # ---- THE HEADLINE (BASELINE's metric): mapped reads/s of the whole job, FASTQ in -> SAM out, through the
    # product command line tool on all N GPUs of the job (`align --gpus N -t T`: reads sharded over the GPUs, index replicated,
    # output gathered in input order; the reference's loop src/main.cpp:106-148).  The K timed steps are K batches per GPU,
    # i.e. steps x batch x N reads in one FASTQ; the W warm-up steps are an untimed run of the tool on W batches.  Index load
    # and HIP start-up are inside the timed interval.  Rank 0 starts the tool; the other ranks have freed their GPUs and wait.
    e2e = None
    gpu_exe = os.path.join(ROOT, "alfalfa_amd", "alfalfa_amd")
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "alfalfa")
    if args.e2e_reads > 0:
        erng = np.random.default_rng(77)
        fq = os.path.join(workdir, "e2e.fq")
        open(fq, "wb").close()
        # N > 1: the reads of ONE GPU's share (steps x batch) are simulated once and written N times (names r<copy>_<i>): the
        # simulator, not the mapper, would otherwise dominate the run; the mapper keeps nothing from read to read
        copies = world if (world > 1 and args.e2e_reads == args.steps * args.batch * world) else 1
        share = args.e2e_reads // copies
        done_reads = 0
        while done_reads < share:               # in chunks: the simulator holds a float per base
            nchunk = min(200000 if not c4 else 20000, share - done_reads)
            if c4:
                import seqgen
                er = seqgen.sim_reads(erng, ref_bytes, nchunk, 2000, 10000, args.err)      # (simulate_long without the reverse complements)
            else:
                er, _, _, _ = simulate_batch(erng, ref, nchunk, args.read_len, args.err)
            with open(fq, "ab") as f:
                for lo in range(0, nchunk, 50000):
                    f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (done_reads + lo + i, r, b"I" * len(r)) for i, r in enumerate(er[lo:lo + 50000])))
            done_reads += nchunk
            del er
        if copies > 1:
            block = open(fq, "rb").read()
            with open(fq, "ab") as f:
                for c in range(1, copies):
                    f.write(block.replace(b"@r", b"@r%d_" % c))
            del block
        e2e = {"reads": args.e2e_reads, "threads": args.e2e_threads, "gpus": world}
        if world > 1:
            dist.barrier(group=cpu_group)                 # every other rank has released its GPU
        if os.path.exists(gpu_exe):
            outp = os.path.join(workdir, "alfalfa_amd_gpu.sam")
            if args.warmup > 0:                           # W untimed warm-up batches through the same tool
                wfq = os.path.join(workdir, "warm.fq")
                nwarm = min(args.e2e_reads, args.warmup * args.batch)
                with open(fq, "rb") as f, open(wfq, "wb") as g:
                    for _ in range(4 * nwarm):
                        g.write(f.readline())
                run_cli(gpu_exe, index_prefix, wfq, os.path.join(workdir, "warm.sam"), args.e2e_threads, cli_extra, gpus=world)
            dt, rcode, err_txt, busy = run_cli(gpu_exe, index_prefix, fq, outp, args.e2e_threads, cli_extra, gpus=world, stats=True)
            md5_all, nrec = sam_record_md5(outp)
            e2e["alfalfa_amd_gpu"] = {"seconds": dt, "reads_per_s": args.e2e_reads / dt, "rc": rcode, "records": nrec, "sam_md5": md5_all,
                                      "gpu_busy_percent": busy,
                                      "stats": [l for l in err_txt.split("\n") if l.startswith("#")]}
        if world > 1:
            dist.barrier(group=cpu_group)
        if os.path.exists(ref_exe) and world == 1 and not args.no_cpu_baseline and args.cpu_e2e_reads > 0:
            ncpu = min(args.cpu_e2e_reads, args.e2e_reads)
            cfq = fq
            if ncpu < args.e2e_reads:
                cfq = os.path.join(workdir, "cpu.fq")
                with open(fq, "rb") as f, open(cfq, "wb") as g:
                    for _ in range(4 * ncpu):
                        g.write(f.readline())
            outp = os.path.join(workdir, "reference_cpu.sam")
            dt, rcode, _err, _busy = run_cli(ref_exe, index_prefix, cfq, outp, args.e2e_threads, cli_extra)
            md5_ref, nrec = sam_record_md5(outp)
            e2e["reference_cpu"] = {"seconds": dt, "reads": ncpu, "reads_per_s": ncpu / dt, "rc": rcode, "records": nrec, "sam_md5": md5_ref}
            if "alfalfa_amd_gpu" in e2e:
                md5_gpu_prefix, _n = sam_record_md5(os.path.join(workdir, "alfalfa_amd_gpu.sam"), ncpu)
                e2e["sam_identical"] = md5_ref == md5_gpu_prefix and nrec == ncpu
                e2e["sam_identical_on"] = "the first %d reads (all that the reference mapped)" % ncpu
    elif world > 1:
        dist.barrier(group=cpu_group)
        dist.barrier(group=cpu_group)

    # the baseline the north star asks for: the reference's own CPU path, multithreaded on this box's host cores.  It can
    # only be timed as the whole command line (it has no hot-path-only mode), on a bounded prefix of the same FASTQ;
    # the scalar port of the hot path alone (1 core) stays beside it.
    if e2e and e2e.get("reference_cpu", {}).get("rc") == 0:
        port = cpu
        cpu = {"value": e2e["reference_cpu"]["reads_per_s"], "unit": "reads/s", "cores": args.e2e_threads, "kind": "reference",
               "host_cores_usable": cores, "host_cpu_count": os.cpu_count(),
               "sample": "%d reads of the same workload through the compiled reference (oracle/_ref/alfalfa align -t %d), whole command "
                         "line FASTQ -> SAM, %.1f s wall" % (e2e["reference_cpu"]["reads"], args.e2e_threads, e2e["reference_cpu"]["seconds"]),
               "hot_path_port_1core": port}
    # ---- ksw_align (mate rescue): not part of `value`; kernel time of one batch of rescue-shaped tasks
    # (150-bp mate with 3 % substitutions placed in a window of 2-3 mate lengths + 200, the mapper's penalties and xtra)
    align = None
    try:
        if world > 1:
            raise RuntimeError("skipped at N > 1")
        arng = np.random.default_rng(4242)
        nt4 = np.full(256, 4, np.uint8)
        for i, ch in enumerate(b"acgt"):
            nt4[ch] = i
        refc = nt4[np.frombuffer(ref, np.uint8)]
        pairs = []
        for _ in range(args.align_tasks):
            wl = int(arng.integers(2 * 150, 3 * 150 + 200))
            o = int(arng.integers(0, len(refc) - wl))
            w = refc[o:o + wl]
            qo = int(arng.integers(0, wl - 150))
            qv = w[qo:qo + 150].copy()
            sub = arng.random(150) < 0.03
            qv[sub] = (qv[sub] + arng.integers(1, 4, int(sub.sum()))) % 4
            pairs.append((qv, w))
        pa = ffi.default_params()
        pa.gapo, pa.gape = -pa.gapo, -pa.gape          # ksw_align gets alnOptions.openGap / extendGap as they are (negative)
        xtra = ffi.KSW_XSTART | ffi.KSW_XBYTE
        atasks, _akeep = ffi.make_align_tasks(pairs, xtra)
        ffi.ksw_align_batch(pa, atasks)
        best = min(_align_ms(pa, atasks) for _ in range(3))
        acells = sum(len(a) * len(b) for a, b in pairs)
        align = {"tasks": len(pairs), "forward_cells": acells, "kernel_ms": best, "gcups": acells / best / 1e6,
                 "note": "cells of the forward pass only (the start-position pass runs too); 150-bp mates, byte kernel"}
        nchk = min(500, len(pairs))
        sub_t, _k2 = ffi.make_align_tasks(pairs[:nchk], xtra)
        tt = time.perf_counter()
        exp = O.orc_ksw_align_batch(pa, sub_t)
        align["oracle_1core_gcups"] = sum(len(a) * len(b) for a, b in pairs[:nchk]) / (time.perf_counter() - tt) / 1e9
        align["parity_sample_ok"] = bool(np.array_equal(ffi.ksw_align_batch(pa, sub_t), exp))
    except Exception as e:  # informative section: never fails the bench line
        align = {"error": str(e)}
    dp_ms = sum(ext_ms) + sum(nw_ms) + sum(gl_ms)

    def dp_roof(cells, ms, ops):
        g = cells / max(ms, 1e-9) / 1e6
        return {"gcups": g, "ops_per_cell": ops, "achieved": g * ops / 1e3, "frac": g * ops / 1e3 / VALU_PEAK_TOPS}
    roofline_dp = {
        "bound": "valu", "peak": VALU_PEAK_TOPS, "unit": "T int32 lane-op/s",
        "peak_note": "256 CU x 4 SIMD-32 x 2.4 GHz; algorithmic ops per cell from SURVEY 8d (12 / 8 / 16)",
        "kernels": "ksw_lane2_kernel<MODE> (band in registers, G skewed lanes per task: the short-read stream); ksw_wave_kernel / "
                   "ksw_group_kernel (+ backtrack kernels) for bands beyond 64 diagonals, irregular tasks and give-ups",
        "extend": dp_roof(cells_ext, sum(ext_ms), 12), "nw": dp_roof(cells_nw, sum(nw_ms), 8), "global": dp_roof(cells_gl, sum(gl_ms), 16),
        "cells": "static band envelope sum_i (min(i+w+1,qlen) - max(i-w,0)), what the kernels compute; the reference's executed "
                 "inner-loop iterations of ksw_extend (SURVEY 8d) are executed_over_envelope x that (oracle counter on the CPU sample)",
        "extend_executed_over_envelope": executed_ratio,
        "band_in_registers_tasks_and_giveups": fast_tasks,
        "extend_giveups_per_step": giveups_each,
        "timing": "each batch of the timed steps launched once more by itself, HIP events around its kernels",
    }
    # the product path's own kernel: map_kernel runs the same DP calls inside one wavefront per read (plus the trial loop's
    # control flow, which has no cell count); its rate in the same currency as roofline_dp, on the first timed step's reads
    roofline_map = None
    if map_call.get("map_kernel_ms"):
        ops0 = 12 * timed[0][1].cells + 8 * timed[0][2].cells + 16 * timed[0][3].cells
        ach = ops0 / (map_call["map_kernel_ms"] * 1e-3) / 1e12
        roofline_map = {"kernel": "map_kernel (csrc/map_kernels.hip): seeds + regions in, alignments + chains + CIGAR operations out, one wavefront per read",
                        "bound": "valu", "peak": VALU_PEAK_TOPS, "unit": "T int32 lane-op/s", "achieved": ach, "frac": ach / VALU_PEAK_TOPS,
                        "avg_launch_ms": map_call["map_kernel_ms"], "reads_per_launch": map_call["reads"],
                        "algorithmic_ops_per_launch": ops0,
                        "note": "algorithmic ops = SURVEY 8d's per-cell figures x the band-envelope cells of the step's ksw_extend / ksw_nw / "
                                "ksw_global calls (the same calls the batch kernels of roofline_dp execute); one task per wavefront, so the "
                                "short calls of this workload (bands of 10-40 diagonals) leave most lanes idle: latency-bound by the serial "
                                "trial loop, not by lane throughput"}
    occupancy = None
    try:
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_seed_limiter.json"))
        if prof:
            occupancy = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
    except Exception:
        occupancy = None
    e2e_gpu = (e2e or {}).get("alfalfa_amd_gpu", {})
    replay_value = whole_job_value(args.batch, args.steps, world, elapsed)
    mapped = e2e_gpu.get("reads_per_s") if e2e_gpu.get("rc") == 0 else None
    # ---- the C4 block (BASELINE configs[3], the configuration the 1/2/4/8 metric is quoted on): at N = 1 the whole line of a
    # second run with --config c4 (replay, rooflines, end to end, cpu_baseline); at N > 1 the end-to-end figure on N GPUs
    if not c4 and not args.no_c4 and world > 1:
        try:
            c4_block = c4_end_to_end(world, args.e2e_threads, dev)
        except Exception as e:   # the C4 block never fails the C3 line
            c4_block = {"error": repr(e)}
    out = {
        "metric": "mapped reads/sec (whole node, FASTQ -> SAM, SAM byte-identical to the reference) + banded-DP GCUPS",
        "value": mapped if mapped else replay_value,
        "value_is": ("mapped reads/s of the whole job: %d reads (steps x batch x GPUs) FASTQ in -> SAM out through `alfalfa_amd align --gpus %d "
                     "-t %d`, index load and HIP start-up included; compare with cpu_baseline.value (the compiled reference, same "
                     "thread count, same box).  The kernels-only figure is replay_reads_per_s" % (args.e2e_reads, world, args.e2e_threads))
                    if mapped else
                    "hot-path replay (no end-to-end run was made: --e2e-reads 0 or the tool is missing): kernel launches over resident batches",
        "mapped_reads_per_s": mapped,
        "mapped_reads_per_s_is": "SURVEY 8d: FASTQ in -> SAM out through alfalfa_amd align --gpus %d -t %d (index load and HIP start-up included), "
                                 "SAM identical to the reference: %s" % (world, args.e2e_threads, (e2e or {}).get("sam_identical")),
        "replay_reads_per_s": replay_value,
        "replay_is": "hot-path replay: kernel launches over batches already packed and resident in HBM (no H2D of reads/tasks, no D2H "
                     "of results, no host chaining between DP rounds), %d steps of %d reads per GPU, %.3f ms per step; what the "
                     "rooflines below are measured on.  NOT comparable with cpu_baseline" % (args.steps, args.batch, elapsed / args.steps * 1e3),
        "replay_ms_per_step": elapsed / args.steps * 1e3,
        "c4": c4_block,
        "pcie_inclusive_reads_per_s": pcie["reads_per_s"],
        "pcie_inclusive": pcie,
        "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
        "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
        "hw_queues_cli": _ORIG_HWQ if _ORIG_HWQ is not None else "unset (the tool then takes 4)",
        "timed_step_parity_vs_oracle": chk,
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": (e2e_gpu["seconds"] / args.steps * 1e3) if mapped else elapsed / args.steps * 1e3,
        "ms_per_step_is": "end-to-end seconds / steps (a step = one batch per GPU through the whole tool)" if mapped else "replay",
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/int32",
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[3]: %d bp iid random reference, 2-10 kbp reads at %.0f%% error (60/20/20 sub/ins/del), -e 0.12, "
                         "index sparseness K=%d, %d reads per step per GPU, %d timed steps" % (args.ref_len, args.err * 100, K, args.batch, args.steps))
                        if c4 else
                        ("BASELINE configs[2]: %d bp iid random reference, %d-bp reads at %.0f%% error (60/20/20 sub/ins/del), "
                         "index sparseness K=%d, %d reads per step per GPU, %d timed steps" %
                         (args.ref_len, args.read_len, args.err * 100, K, args.batch, args.steps)),
            "seed_type": ("SMAM (CLI --seed SMEM default), min_len and sparseQ per read length (src/mapper.cpp:1226, :533), maxCount=10000, fw+rc"
                          if c4 else "SMAM (CLI --seed SMEM default), min_len=%d sparseQ=%d maxCount=10000, fw+rc" % (min_len, sparse_q)),
            "dp_tasks": "every ksw_extend/ksw_nw/ksw_global task the staged host mapper (alfalfa_amd align) issues for the step's "
                        "reads, captured before the timed region and replayed resident; %.2f/%.2f/%.2f tasks per read" %
                        (len(timed[0][1]._tasks) / args.batch, len(timed[0][2]._tasks) / args.batch, len(timed[0][3]._tasks) / args.batch),
            "parallelism": "reads sharded over %d GPU(s), index replicated, no collective" % world,
        },
        "device_ms_total": dev_ms,
        "step_split": {"seed_only_ms_per_step": seed_only_ms, "dp_only_ms_per_step": dp_only_ms,
                       "note": "the timed steps issued once more with only the seed launches / only the DP launches (informative): "
                               "ms_per_step against their sum shows what running them side by side gains"},
        "host_enqueue_ms_per_step": t_enqueue / args.steps * 1e3,
        "dp_gcups": (cells_ext + cells_nw + cells_gl) / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else None,
        "dp_detail": {
            "extend": {"cells": cells_ext, "ms": sum(ext_ms), "gcups": cells_ext / max(sum(ext_ms), 1e-9) / 1e6, "ms_each": [round(x, 4) for x in ext_ms]},
            "nw": {"cells": cells_nw, "ms": sum(nw_ms), "gcups": cells_nw / max(sum(nw_ms), 1e-9) / 1e6, "ms_each": [round(x, 4) for x in nw_ms]},
            "global": {"cells": cells_gl, "ms": sum(gl_ms), "gcups": cells_gl / max(sum(gl_ms), 1e-9) / 1e6, "ms_each": [round(x, 4) for x in gl_ms]},
        },
        "roofline_dp": roofline_dp,
        "roofline_map": roofline_map,
        "seed_queries_per_step": queries_per_step,
        "seeds_per_read": seeds_per_read,
        "index_build_s": t_index,
        "roofline": {
            "kernel": ("sorted_keys_kernel + hipCUB radix sort + sorted_eval_kernel (queries in 10-mer order: the index is beyond the caches)"
                       if timed[0][0].sorted else "smam_count_kernel"),
            "bound": "hbm",
            "share_of_a_run": "this is the dominant kernel of the REPLAY (the batch entry points); in the end-to-end run that `value` measures the "
                              "seed kernels are 8 % (C3) / 4 % (C4) of the GPU time and map_kernel 47-57 % / 89 % (profiles/r03_cli_c*_kernel_trace_stats.csv): "
                              "see roofline_map",
            "limiter": "latency / instruction issue, not bandwidth: a gather kernel (FETCH_SIZE = algorithmic bytes, L2 hit rate "
                       "and wave wait fraction in profiles/*_seed_limiter.json)",
            "limiter_counters": occupancy,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_over_algorithmic": (traffic / bytes_per_launch) if traffic else None,
            "traffic_gbs": (traffic / (seed_avg_ms * 1e-3) / 1e9) if traffic else None,
            "traffic_gbs_isolated": (traffic / (float(np.mean(seed_iso_ms)) * 1e-3) / 1e9) if traffic else None,
            "traffic_note": "raw FETCH_SIZE x 1024 per dispatch (profiles/*_pmc_traffic.json, the latest round). Calibrated on this access pattern "
                            "(tools/fetch_calib.hip, profiles/r01_fetch_calibration.csv): an isolated 4/8/16-byte gather from a 2 GiB "
                            "table tallies 64 B, a wide streaming read tallies half its bytes (the guide's x2 case); the seed "
                            "kernels issue gathers, so the raw figure is used",
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "algorithmic_bytes_per_read": bytes_per_read,
            "avg_launch_ms": seed_avg_ms,
            "avg_launch_ms_isolated": float(np.mean(seed_iso_ms)),
            "note": "avg_launch_ms is measured inside the timed region, where the DP kernels of the previous step run "
                    "concurrently on side streams; the isolated figure is the same launch with the GPU to itself",
        },
        "cpu_baseline": cpu,
        "ksw_align": align,
        "end_to_end_cli": e2e,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def c4_end_to_end(world, threads, dev):
    """BASELINE configs[3] end to end on `world` GPUs: 100 Mbp reference (seed 2), index built on the GPU, 20 000 reads of
    2-10 kbp at 10 % per GPU, `alfalfa_amd align -e 0.12 --gpus world -t threads`: mapped reads/s of the whole job"""
    import tempfile
    from alfalfa_amd import ffi
    wd = tempfile.mkdtemp(prefix="alfa_bench_c4_")
    ref = make_reference(100000000, 2)
    refb = ref.tobytes()
    ffi.index_build_on_device(dev)
    try:
        hidx = ffi.HostIndex.build_mem(["chr1"], [refb], K=12)
    finally:
        ffi.index_build_on_device(-1)
    pre = os.path.join(wd, "idx")
    hidx.save(pre)
    del hidx
    share = 50000                     # reads per GPU: simulated once, written `world` times (names r<copy>_<i>)
    n = share * world
    fq = os.path.join(wd, "c4.fq")
    open(fq, "wb").close()
    rng = np.random.default_rng(78)
    for lo in range(0, share, 10000):
        er, _ = simulate_long(rng, refb, min(10000, share - lo), 2000, 10000, 0.10)
        with open(fq, "ab") as f:
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (lo + i, r, b"I" * len(r)) for i, r in enumerate(er)))
    block = open(fq, "rb").read()
    with open(fq, "ab") as f:
        for c in range(1, world):
            f.write(block.replace(b"@r", b"@r%d_" % c))
    del block
    exe = os.path.join(ROOT, "alfalfa_amd", "alfalfa_amd")
    dt, rcode, err_txt, busy = run_cli(exe, pre, fq, os.path.join(wd, "c4.sam"), threads, ["-e", "0.12"], gpus=world, stats=True)
    md5_all, nrec = sam_record_md5(os.path.join(wd, "c4.sam"))
    return {"workload": "BASELINE configs[3]: 100 Mbp iid reference, %d reads of 2-10 kbp at 10 %% error, -e 0.12, K=12" % n,
            "n_gpus": world, "threads": threads, "mapped_reads_per_s": n / dt if rcode == 0 else None, "seconds": dt, "rc": rcode,
            "records": nrec, "sam_md5": md5_all, "gpu_busy_percent": busy, "stats": [l for l in err_txt.split("\n") if l.startswith("#")]}


def _align_ms(params, tasks):
    from alfalfa_amd import ffi
    ffi.ksw_align_batch(params, tasks)
    return ffi.last_timing().kernel_ms


def C_addr(arr):
    import ctypes
    return ctypes.addressof(arr)


if __name__ == "__main__":
    main()

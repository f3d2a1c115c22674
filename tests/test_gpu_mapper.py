"""GPU parity, end to end: the product command line tool (host mapper + libalfa_hot HIP kernels) must write the
compiled reference's SAM records byte for byte (BASELINE configs[0] and [1] + variants)."""
import os
import subprocess

import numpy as np
import pytest

import mapper_util as M
import golden_util as G
import seqgen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(M.CONFIGS))
def test_gpu_mapper_matches_reference_sam(gpu, name):
    body = M.run_config(M.GPU_EXE, name, ["-t", "2"])
    assert M.md5(body) == M.golden_md5(name)


@pytest.mark.parametrize("name", sorted(M.SYNTH_CONFIGS))
def test_gpu_mapper_paired_modes_synthetic(gpu, name):
    """paired modes 1-6 on 3 000 synthetic pairs (--paired-rescue: hundreds of ksw_align calls): md5 of the compiled reference's records"""
    body, err = M.run_synth(M.GPU_EXE, name, ["-t", "4", "--stats"])
    assert M.md5(body) == M.synth_md5(name)
    assert ("rescue" not in name) or " align 0/0" not in err


@pytest.mark.parametrize("name", sorted(M.SYNTH_SE_CONFIGS))
def test_gpu_mapper_option_sweep_synthetic(gpu, name):
    """single-end option sweep (scores, seed types, bands, trial limits, strands, --full-dp): md5 of the reference's records"""
    assert M.md5(M.run_synth_single(M.GPU_EXE, name, ["-t", "4"])) == M.synth_md5(name)


def test_gpu_mapper_vs_oracle_mapper_synthetic(gpu):
    """larger synthetic set (two contigs, short + long reads): product binary vs the oracle-backed host mapper"""
    d = M.workdir()
    rng = np.random.default_rng(11)
    ref = seqgen.rand_ref(rng, 1500000)
    ref2 = seqgen.rand_ref(rng, 200000)
    with open(os.path.join(d, "syn.fa"), "w") as f:
        for name, s in (("chr1", ref), ("chr2 comment", ref2)):
            f.write(">%s\n" % name)
            s = s.decode().upper()
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    cat = ref + b"a" * 50 + ref2

    def fq(path, reads):
        with open(os.path.join(d, path), "w") as f:
            for i, r in enumerate(reads):
                f.write("@r%d/1\n%s\n+\n%s\n" % (i, r.decode().upper(), "I" * len(r)))
    fq("short.fq", seqgen.sim_reads(rng, cat, 20000, 250, 250, 0.02))
    fq("long.fq", seqgen.sim_reads(rng, cat, 300, 2000, 8000, 0.10))
    subprocess.run([M.GPU_EXE, "index", "-r", "syn.fa", "-p", "syn12", "-s", "12"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    subprocess.run([M.GPU_EXE, "index", "-r", "syn.fa", "-p", "syn12sl", "-s", "12", "--suflink"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    for tag, args in (("short", ["-0", "short.fq"]), ("short_suflink", ["-i", "syn12sl", "-0", "short.fq"]), ("long", ["-0", "long.fq", "-e", "0.12"]), ("longdef", ["-0", "long.fq"]),
                      ("short_mem", ["-0", "short.fq", "--seed", "MEM"]), ("long_psmem", ["-0", "long.fq", "-e", "0.12", "--seed", "PSMEM"])):
        outs = []
        for exe, thr in ((M.GPU_EXE, "4"), (M.CPU_EXE, "16")):
            out = os.path.join(d, "syn_%s_%s.sam" % (tag, os.path.basename(exe)))
            idx = [] if "-i" in args else ["-i", "syn12"]
            p = subprocess.run([exe, "align"] + idx + ["-o", out, "-t", thr] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            outs.append(M.sam_body(out))
        assert outs[0] == outs[1], tag
        assert outs[0].count(b"\n") > 100


def test_gpu_mapper_two_device_contexts(gpu):
    """`--gpus 2` on this one-GPU box with ALFA_ALIAS_DEVICES: two worker groups, two resident index copies, two sets of HIP
    contexts; output identical to the reference's and independent of the grouping"""
    os.environ["ALFA_ALIAS_DEVICES"] = "1"
    try:
        for name in ("454_s12", "pe_s12_rescue"):
            body = M.run_config(M.GPU_EXE, name, ["-t", "4", "--gpus", "2"])
            assert M.md5(body) == M.golden_md5(name), name
    finally:
        del os.environ["ALFA_ALIAS_DEVICES"]


def test_bench_two_ranks_report_mapped_reads_per_s(gpu):
    """bench.py --gpus 2 under torchrun, rehearsed on this one-GPU box (both ranks and both worker groups of the tool aliased
    onto device 0, gloo instead of RCCL: RCCL refuses two ranks on one card): the line's `value` is the mapped reads/s of
    `alfalfa_amd align --gpus 2` over steps x batch x 2 reads, n_gpus = 2, and the replay figure stands beside it."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ALFA_ALIAS_DEVICES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "5000",
           "--backend", "gloo", "--no-c4", "--no-cpu-baseline", "--align-tasks", "0"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert out["mapped_reads_per_s"] and out["value"] == out["mapped_reads_per_s"]
    e = out["end_to_end_cli"]
    assert e["reads"] == 2 * 5000 * 2 and e["gpus"] == 2 and e["alfalfa_amd_gpu"]["rc"] == 0 and e["alfalfa_amd_gpu"]["records"] == e["reads"]
    assert out["replay_reads_per_s"] > out["value"]
    assert all(out["timed_step_parity_vs_oracle"].values())

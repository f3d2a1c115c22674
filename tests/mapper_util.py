"""Helpers to run the mapper command line tools on the golden inputs and compare SAM records byte for byte."""
import gzip
import hashlib
import json
import os
import subprocess

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_EXE = os.path.join(ROOT, "alfalfa_amd", "alfalfa_amd")       # product: host mapper + libalfa_hot (HIP)
CPU_EXE = os.path.join(ROOT, "tests", "_cpu_mapper")              # checker: same host mapper + oracle

# name -> (K, arguments); the reference commands of example/EXAMPLE:46,94,102 and variants (tools/make_goldens.py)
CONFIGS = {
    "454_s1": (1, ["-0", "454reads.fq"]),
    "pe_s1": (1, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400"]),
    "pe_local_s1": (1, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--local"]),
    "454_s12": (12, ["-0", "454reads.fq"]),
    "pe_s12": (12, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400"]),
    "454_s1_mem": (1, ["-0", "454reads.fq", "--seed", "MEM"]),
    "454_s1_psmem": (1, ["-0", "454reads.fq", "--seed", "PSMEM"]),
    "454_s12_mem": (12, ["-0", "454reads.fq", "--seed", "MEM"]),
    "454_s12_psmem": (12, ["-0", "454reads.fq", "--seed", "PSMEM"]),
    "454_s1_fulldp": (1, ["-0", "454reads.fq", "--full-dp"]),
    "454_s1_local": (1, ["-0", "454reads.fq", "--local"]),
    "454_s12_e12": (12, ["-0", "454reads.fq", "-e", "0.12"]),
    "454_s1_a5": (1, ["-0", "454reads.fq", "-a", "5"]),
    # --paired-rescue reaches ksw_align (src/mapper.cpp:1571); goldens by tools/make_paired_goldens.py.  (The reference itself
    # segfaults on tighter insert ranges, e.g. -I 3700 -X 4300: its rescue window runs off the text, :1569.)
    "pe_s12_rescue": (12, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--paired-rescue"]),
    "pe_s1_rescue": (1, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--paired-rescue"]),
    "pe_s12_rescue_strict": (12, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--paired-rescue", "-e", "0.02",
                                  "--no-rescue", "-l", "60"]),
    "pe_s12_rescue_a3": (12, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--paired-rescue", "-a", "3", "--local"]),
    # paired modes 3-6 (src/mapper.cpp:1761-2373); mode 2 makes the reference segfault on this input
    "pe_s12_mode3": (12, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--paired-mode", "3"]),
    "pe_s12_mode4_rescue": (12, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--paired-mode", "4", "--paired-rescue"]),
    "pe_s12_mode5_rescue": (12, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--paired-mode", "5", "--paired-rescue"]),
    "pe_s1_mode6_rescue": (1, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3600", "-X", "4400", "--paired-mode", "6", "--paired-rescue"]),
    "pe_s12_rescue_wide": (12, ["-1", "mate1.fq", "-2", "mate2.fq", "-I", "3000", "-X", "5000", "--paired-rescue", "-e", "0.03"]),
}


def workdir():
    d = G.tmpdir()
    for f in ("reference.fasta", "454reads.fq", "mate1.fq", "mate2.fq"):
        G.input_path(f)
    return d


def sam_body(path):
    lines = [l for l in open(path, "rb").read().split(b"\n") if not l.startswith(b"@PG")]
    return b"\n".join(lines)


def golden_body(name):
    return gzip.open(os.path.join(G.GOLD, "sam_%s.sam.gz" % name), "rb").read()


def golden_md5(name):
    return json.load(open(os.path.join(G.GOLD, "sam_md5.json")))[name]["md5"]


def run_config(exe, name, extra=()):
    d = workdir()
    K, args = CONFIGS[name]
    pre = os.path.join(d, "idx_%s_%d" % (os.path.basename(exe), K))
    if not os.path.exists(pre + ".aux"):
        subprocess.run([exe, "index", "-r", "reference.fasta", "-p", pre, "-s", str(K)], cwd=d, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = os.path.join(d, "out_%s_%s.sam" % (os.path.basename(exe), name))
    p = subprocess.run([exe, "align", "-i", pre, "-o", out] + args + list(extra), cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return sam_body(out)


def md5(b):
    return hashlib.md5(b).hexdigest()


# ---- synthetic paired-end set for --paired-rescue (many ksw_align calls; the lambda example makes only 12).  Inputs are
# regenerated from a seed; the reference's SAM md5 is committed (tests/golden/sam_synth_md5.json, tools/make_paired_goldens.py)
SYNTH_CONFIGS = {
    "synpe_rescue": ["-I", "250", "-X", "650", "--paired-rescue"],
    "synpe_rescue_local": ["-I", "250", "-X", "650", "--paired-rescue", "-a", "3", "--local", "-e", "0.05"],
    "synpe_rescue_long": ["-I", "250", "-X", "650", "--paired-rescue", "-M", "2", "-U", "-3"],
    "synpe_mode2_rescue": ["-I", "250", "-X", "650", "--paired-rescue", "--paired-mode", "2"],
    "synpe_mode3": ["-I", "250", "-X", "650", "--paired-mode", "3"],
    "synpe_mode4_rescue": ["-I", "250", "-X", "650", "--paired-rescue", "--paired-mode", "4"],
    "synpe_mode5_rescue": ["-I", "250", "-X", "650", "--paired-rescue", "--paired-mode", "5"],
    "synpe_mode5": ["-I", "250", "-X", "650", "--paired-mode", "5", "-a", "2"],
    "synpe_mode6_rescue": ["-I", "250", "-X", "650", "--paired-rescue", "--paired-mode", "6"],
    # pairing options (isConcordant :1311, dpWindow :1464, discordant pairs, mixed output)
    "synpe_rf_rescue": ["-I", "250", "-X", "650", "--paired-rescue", "--orientation", "rf"],
    "synpe_ff_rescue": ["-I", "250", "-X", "650", "--paired-rescue", "--orientation", "ff"],
    "synpe_strict_pairs": ["-I", "350", "-X", "500", "--no-mixed", "--no-discordant", "-a", "2"],
    "synpe_geometry": ["-I", "100", "-X", "450", "--paired-rescue", "--dovetail", "--no-contain", "--no-overlap"],
    "synpe_mode5_ff": ["-I", "250", "-X", "650", "--paired-rescue", "--paired-mode", "5", "--orientation", "ff", "-a", "3"],
    "synpe_mode3_rf": ["-I", "250", "-X", "650", "--paired-mode", "3", "--orientation", "rf", "--no-discordant"],
}


def synth_paired():
    """writes synpe.fa / synpe_1.fq / synpe_2.fq into the work directory (once) and returns it"""
    import numpy as np
    import seqgen
    d = workdir()
    if os.path.exists(os.path.join(d, "synpe_2.fq")):
        return d
    rng = np.random.default_rng(2024)
    ref = seqgen.rand_ref(rng, 400000)
    with open(os.path.join(d, "synpe.fa"), "w") as f:
        f.write(">synpe\n")
        s = ref.decode().upper()
        for i in range(0, len(s), 70):
            f.write(s[i:i + 70] + "\n")
    # one mate is clean; the other is often seedless but alignable without gaps (what the rescue finds), noisy, or unrelated
    def noisy(r):
        u = r.random()
        return "periodic" if u < 0.35 else float(r.choice([0.01, 0.06, 0.12, 0.2]))
    m1, m2 = seqgen.sim_pairs(rng, ref, 3000, 150, 300, 600, 0.01, noisy, 3000)
    for k in range(0, 3000, 37):
        m2[k] = seqgen.rand_ref(rng, 150)
    for path, reads, tag in (("synpe_1.fq", m1, 1), ("synpe_2.fq", m2, 2)):
        with open(os.path.join(d, path), "w") as f:
            for i, r in enumerate(reads):
                f.write("@p%d/%d\n%s\n+\n%s\n" % (i, tag, r.decode().upper(), "I" * len(r)))
    return d


def run_synth(exe, name, extra=()):
    d = synth_paired()
    pre = os.path.join(d, "synpe_%s" % os.path.basename(exe))
    if not os.path.exists(pre + ".aux"):
        subprocess.run([exe, "index", "-r", "synpe.fa", "-p", pre, "-s", "12"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    out = os.path.join(d, "out_%s_%s.sam" % (os.path.basename(exe), name))
    p = subprocess.run([exe, "align", "-i", pre, "-o", out, "-1", "synpe_1.fq", "-2", "synpe_2.fq"] + SYNTH_CONFIGS[name] + list(extra),
                       cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return sam_body(out), p.stderr.decode()


# ---- synthetic single-end set: the command line options that steer the host logic (src/options.cpp), reference md5s committed
SYNTH_SE_CONFIGS = {
    "synse_default": [],
    "synse_scores": ["-M", "2", "-U", "-3", "-O", "-4", "-E", "-2"],
    "synse_minlen": ["-l", "25", "-c", "0.1", "-a", "3"],
    "synse_band_trials": ["-b", "20", "-f", "3", "--reset-failures", "--max-alignments", "50"],
    "synse_forward_only": ["--no-reverse", "--no-rescue"],
    "synse_reverse_only": ["--no-forward", "-e", "0.05"],
    "synse_seeds": ["--no-sparseness", "-m", "50", "--max-smems", "3", "--max-mems", "5", "--min-mem-length", "30", "--seed", "PSMEM"],
    "synse_mem_local": ["--seed", "MEM", "--local", "-a", "2", "-e", "0.1"],
    "synse_fulldp": ["--full-dp", "-a", "2"],
}


def synth_single():
    import numpy as np
    import seqgen
    d = workdir()
    if os.path.exists(os.path.join(d, "synse.fq")):
        return d
    rng = np.random.default_rng(4048)
    a = seqgen.rand_ref(rng, 250000)
    b = seqgen.rand_ref(rng, 60000)
    b = b + a[1000:3000] + b"acgt" * 50 + b[:500]            # repeats across and inside contigs
    with open(os.path.join(d, "synse.fa"), "w") as f:
        for name, sq in (("ctgA", a), ("ctgB some text", b)):
            f.write(">%s\n" % name)
            t = sq.decode().upper()
            for i in range(0, len(t), 80):
                f.write(t[i:i + 80] + "\n")
    cat = a + b"a" * 40 + b
    reads = seqgen.sim_reads(rng, cat, 2500, 80, 400, 0.03) + seqgen.sim_reads(rng, cat, 150, 800, 3000, 0.08)
    reads += [seqgen.rand_ref(rng, 150) for _ in range(30)] + [b"acgt" * 40, b"n" * 80, a[1500:1700]]
    with open(os.path.join(d, "synse.fq"), "w") as f:
        for i, r in enumerate(reads):
            f.write("@s%d\n%s\n+\n%s\n" % (i, r.decode().upper(), "5" * len(r)))
    return d


def run_synth_single(exe, name, extra=()):
    d = synth_single()
    pre = os.path.join(d, "synse_%s" % os.path.basename(exe))
    if not os.path.exists(pre + ".aux"):
        subprocess.run([exe, "index", "-r", "synse.fa", "-p", pre, "-s", "8"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    out = os.path.join(d, "out_%s_%s.sam" % (os.path.basename(exe), name))
    p = subprocess.run([exe, "align", "-i", pre, "-o", out, "-0", "synse.fq"] + SYNTH_SE_CONFIGS[name] + list(extra),
                       cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return sam_body(out)


def synth_md5(name):
    return json.load(open(os.path.join(G.GOLD, "sam_synth_md5.json")))[name]


# ---- C5-shaped paired long reads at reduced reference size (BASELINE configs[4]: 5 kbp pairs, insert 12 000 +- 2 000,
# 10 % error, -I 8000 -X 16000 -e 0.12).  Deterministic inputs; the compiled reference's md5 is a fixture
# (tools/make_c5_goldens.py).
C5_ARGS = ["-I", "8000", "-X", "16000", "-e", "0.12"]


def write_c5_paired(d, n_pairs=150, ref_len=3000000, seed=31):
    import numpy as np
    import seqgen
    rng = np.random.default_rng(seed)
    ref = seqgen.rand_ref(rng, ref_len)
    with open(os.path.join(d, "c5.fa"), "w") as f:
        f.write(">chr1\n")
        s = ref.decode().upper()
        for i in range(0, len(s), 60):
            f.write(s[i:i + 60] + "\n")
    m1, m2 = seqgen.sim_pairs(rng, ref, n_pairs, 5000, 10000, 14000, 0.10, 0.10, 200)
    for tag, reads in ((1, m1), (2, m2)):
        with open(os.path.join(d, "c5_%d.fq" % tag), "w") as f:
            for i, r in enumerate(reads):
                f.write("@p%d/%d\n%s\n+\n%s\n" % (i, tag, r.decode().upper(), "I" * len(r)))
    return "c5.fa", "c5_1.fq", "c5_2.fq"


def run_c5_paired(exe, d, threads="4", index_exe=None):
    pre = os.path.join(d, "c5idx_%s" % os.path.basename(exe))
    if not os.path.exists(pre + ".aux"):
        subprocess.run([index_exe or exe, "index", "-r", "c5.fa", "-p", pre, "-s", "12"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    out = os.path.join(d, "c5_%s.sam" % os.path.basename(exe))
    p = subprocess.run([exe, "align", "-i", pre, "-1", "c5_1.fq", "-2", "c5_2.fq", "-o", out, "-t", threads] + C5_ARGS, cwd=d,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return sam_body(out)


# ---- C4 at size (BASELINE configs[3]): 100 Mbp iid reference (PRNG seed 2), reads of 2-10 kbp at 10 % error
# (60/20/20), `-e 0.12`, index sparseness 12.  Deterministic inputs; the compiled reference's md5 is a fixture
# (tests/golden/c4_md5.json, tools/make_c4_goldens.py).
C4_ARGS = ["-e", "0.12"]
C4_REF_LEN = 100000000
C4_READS = 20000


def c4_reference():
    import numpy as np
    import seqgen
    return seqgen.ACGT[np.random.default_rng(2).integers(0, 4, C4_REF_LEN)]


def c4_reads(ref_bytes, n=C4_READS, seed=404):
    import numpy as np
    import seqgen
    return seqgen.sim_reads(np.random.default_rng(seed), ref_bytes, n, 2000, 10000, 0.10)


def write_c4(d, n=C4_READS):
    """writes c4.fa / c4.fq into d (once); returns (reference as uint8 array, reads)"""
    ref = c4_reference()
    refb = ref.tobytes()
    reads = c4_reads(refb, n)
    fa = os.path.join(d, "c4.fa")
    if not os.path.exists(fa):
        up = refb.upper()
        with open(fa, "wb") as f:
            f.write(b">chr1\n")
            f.write(b"\n".join(up[i:i + 60] for i in range(0, len(up), 60)) + b"\n")
    fq = os.path.join(d, "c4.fq")
    if not os.path.exists(fq):
        with open(fq, "wb") as f:
            for i, r in enumerate(reads):
                f.write(b"@r%d\n%s\n+\n%s\n" % (i, r.upper(), b"I" * len(r)))
    return ref, reads


def run_c4(exe, d, threads="16", index_args=(), env=None, extra=(), tag=""):
    pre = os.path.join(d, "c4idx_%s" % os.path.basename(exe))
    if not os.path.exists(pre + ".aux"):
        subprocess.run([exe, "index", "-r", "c4.fa", "-p", pre, "-s", "12"] + list(index_args), cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    out = os.path.join(d, "c4_%s%s.sam" % (os.path.basename(exe), tag))
    p = subprocess.run([exe, "align", "-i", pre, "-0", "c4.fq", "-o", out, "-t", threads] + C4_ARGS + list(extra), cwd=d, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    run_c4.last_stderr = p.stderr.decode()
    return sam_body(out)

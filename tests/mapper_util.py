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

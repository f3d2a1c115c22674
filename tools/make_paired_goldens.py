#!/usr/bin/env python3
"""Whole-SAM goldens of the compiled reference (oracle/_ref/alfalfa) for the paired-end configurations that reach
ksw_align (--paired-rescue).  Adds sam_<name>.sam.gz files and entries to tests/golden/sam_md5.json
for the configurations of tests/mapper_util.py whose name contains "rescue".  Runs only in the build container."""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G      # noqa: E402
import mapper_util as M      # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "alfalfa")


def main():
    d = M.workdir()
    md5 = json.load(open(os.path.join(G.GOLD, "sam_md5.json")))
    for name, (K, args) in sorted(M.CONFIGS.items()):
        if "rescue" not in name and "mode" not in name:
            continue
        pre = os.path.join(d, "ref_idx_%d" % K)
        if not os.path.exists(pre + ".aux"):
            subprocess.run([EXE, "index", "-r", "reference.fasta", "-p", pre, "-s", str(K)], cwd=d, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = os.path.join(d, "ref_%s.sam" % name)
        subprocess.run([EXE, "align", "-i", pre, "-o", out] + args, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        body = M.sam_body(out)
        md5[name] = {"md5": hashlib.md5(body).hexdigest(), "args": ["-s%d" % K] + args}
        with gzip.GzipFile(os.path.join(G.GOLD, "sam_%s.sam.gz" % name), "wb", mtime=0) as fo:
            fo.write(body)
        print(name, md5[name]["md5"], len(body))
    json.dump(md5, open(os.path.join(G.GOLD, "sam_md5.json"), "w"), indent=1, sort_keys=True)
    # synthetic paired-end set (inputs regenerated from a seed by tests/mapper_util.py): md5 of the reference's records only
    synth = {}
    for name in sorted(M.SYNTH_CONFIGS):
        body, _err = M.run_synth(EXE, name, ["-t", "8"])
        synth[name] = hashlib.md5(body).hexdigest()
        print(name, synth[name], len(body))
    for name in sorted(M.SYNTH_SE_CONFIGS):
        body = M.run_synth_single(EXE, name, ["-t", "8"])
        synth[name] = hashlib.md5(body).hexdigest()
        print(name, synth[name], len(body))
    json.dump(synth, open(os.path.join(G.GOLD, "sam_synth_md5.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

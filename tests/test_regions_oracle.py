"""CPU: oracle/regions_oracle.py (the Python restatement of identifyAndFilter / coveredBases that checks the device kernel)
against the host mapper's C++ restatement of the same two functions -- which is pinned by the reference's SAM goldens
(tests/test_mapper_cpu.py) -- on every read of the lambda examples: the regions the mapper actually builds, logged with
ALFA_REGION_LOG."""
import os
import subprocess
import sys

import pytest

import mapper_util as M

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import regions_oracle as RO  # noqa: E402


def _records(path):
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines) and lines[i]:
        tag, qlen, ed, nfw, nrc, nreg = lines[i].split()
        assert tag == "R"
        qlen, ed, nfw, nrc, nreg = int(qlen), int(ed), int(nfw), int(nrc), int(nreg)
        i += 1

        def take(k):
            nonlocal i
            out = [tuple(int(x) for x in lines[i + j].split()) for j in range(k)]
            i += k
            return out
        fw, rc = take(nfw), take(nrc)
        regs = []
        for _ in range(nreg):
            g = lines[i].split()
            assert g[0] == "G"
            i += 1
            regs.append((int(g[1]), int(g[2]), int(g[3]), int(g[4]), take(int(g[5]))))
        yield qlen, ed, fw, rc, regs


@pytest.mark.parametrize("name,extra", [("454_s12", []), ("454_s1", []), ("pe_s12", []), ("454_s12", ["-m", "3", "--max-mems", "2", "--min-mem-length", "30"])])
def test_python_restatement_equals_host_mapper(name, extra):
    d = M.workdir()
    log = os.path.join(d, "regions_%s_%d.log" % (name, len(extra)))
    env = dict(os.environ, ALFA_REGION_LOG=log)
    K, args = M.CONFIGS[name]
    pre = os.path.join(d, "idx_%s_%d" % (os.path.basename(M.CPU_EXE), K))
    if not os.path.exists(pre + ".aux"):
        subprocess.run([M.CPU_EXE, "index", "-r", "reference.fasta", "-p", pre, "-s", str(K)], cwd=d, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    p = subprocess.run([M.CPU_EXE, "align", "-i", pre, "-o", os.path.join(d, "regions.sam"), "-t", "2"] + args + extra, cwd=d, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    prm = {"max_smem_occ": 10000, "max_region_mems": 20, "min_mem_reg_length": 50}
    if extra:
        prm = {"max_smem_occ": 3, "max_region_mems": 2, "min_mem_reg_length": 30}
    n = nreg = 0
    for qlen, ed, fw, rc, regs in _records(log):
        exp = []
        amb = False
        for strand, ms in ((1, fw), (0, rc)):
            rr, a = RO.identify_and_filter(list(ms), qlen, ed, **prm)
            amb |= a
            for r in rr:
                sc, a2 = RO.covered_bases(r[2])
                amb |= a2
                exp.append((strand, r[0], r[1], sc, r[2]))
        if amb:
            continue            # std::sort's order of equal elements decides: not reproducible by a restatement
        assert len(exp) == len(regs), (n, len(exp), len(regs))
        for e, g in zip(exp, regs):
            assert e[:4] == g[:4] and list(e[4]) == list(g[4]), (n, e[:4], g[:4])
        n += 1
        nreg += len(regs)
    assert n >= 50 and nreg >= 50, (n, nreg)


def test_python_restatement_on_the_synthetic_set():
    """two contigs with repeats, 2 700 short and long reads"""
    d = M.synth_single()
    log = os.path.join(d, "regions_synse.log")
    os.environ["ALFA_REGION_LOG"] = log
    try:
        M.run_synth_single(M.CPU_EXE, "synse_default", ["-t", "2"])
    finally:
        del os.environ["ALFA_REGION_LOG"]
    n = nreg = 0
    for qlen, ed, fw, rc, regs in _records(log):
        exp = []
        amb = False
        for strand, ms in ((1, fw), (0, rc)):
            rr, a = RO.identify_and_filter(list(ms), qlen, ed)
            amb |= a
            for r in rr:
                sc, a2 = RO.covered_bases(r[2])
                amb |= a2
                exp.append((strand, r[0], r[1], sc, r[2]))
        if amb:
            continue
        assert len(exp) == len(regs), (n, len(exp), len(regs))
        for e, g in zip(exp, regs):
            assert e[:4] == g[:4] and list(e[4]) == list(g[4]), (n, e[:4], g[:4])
        n += 1
        nreg += len(regs)
    assert n > 2000 and nreg > 2000, (n, nreg)


# ---- the REFERENCE's regions: oracle/_ref/alfalfa_reglog is the reference executable compiled in place with a logger in front
# of calculateCandRegions' final sort (oracle/ref_region_logger.h), so the regions identifyAndFilter + coveredBases build for
# every read -- members, bounds, scores, creation order -- can be compared with the host mapper's (logged with ALFA_REGION_LOG),
# to which the Python restatement above is pinned with their input seeds.  Reference == host mapper == regions_oracle.py.
REF_REGLOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "alfalfa_reglog")


def _ref_records(path):
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines) and lines[i]:
        tag, nreg = lines[i].split()
        assert tag == "R"
        i += 1
        regs = []
        for _ in range(int(nreg)):
            g = lines[i].split()
            assert g[0] == "G"
            i += 1
            mem = [tuple(int(x) for x in lines[i + j].split()) for j in range(int(g[5]))]
            i += int(g[5])
            regs.append((int(g[1]), int(g[2]), int(g[3]), int(g[4]), tuple(mem)))
        yield tuple(regs)


@pytest.mark.skipif(not os.path.exists(REF_REGLOG), reason="compiled reference with the region logger not in the tree")
@pytest.mark.parametrize("name,extra", [("454_s12", []), ("454_s1", []), ("pe_s12", []), ("454_s12_e12", []),
                                        ("454_s12", ["-m", "3", "--max-mems", "2", "--min-mem-length", "30"])])
def test_reference_regions_equal_host_mapper_regions(name, extra):
    from collections import Counter
    d = M.workdir()
    K, args = M.CONFIGS[name]
    logs = {}
    for tag, exe, var in (("ref", REF_REGLOG, "ALFA_REF_REGION_LOG"), ("host", M.CPU_EXE, "ALFA_REGION_LOG")):
        log = os.path.join(d, "reglog_%s_%s_%d.log" % (tag, name, len(extra)))
        pre = os.path.join(d, "idx_%s_%d" % (os.path.basename(M.CPU_EXE), K))
        if not os.path.exists(pre + ".aux"):
            subprocess.run([M.CPU_EXE, "index", "-r", "reference.fasta", "-p", pre, "-s", str(K)], cwd=d, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
        p = subprocess.run([exe, "align", "-i", pre, "-o", os.path.join(d, "reglog_%s.sam" % tag), "-t", "1"] + args + extra, cwd=d,
                           env=dict(os.environ, **{var: log}), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        logs[tag] = log
    ref = Counter(_ref_records(logs["ref"]))
    host = Counter(tuple((fw, rb, re_, sc, tuple(mem)) for fw, rb, re_, sc, mem in regs) for _q, _e, _f, _r, regs in _records(logs["host"]))
    assert sum(ref.values()) >= 100 and sum(len(r) for r in ref.elements()) >= 100
    assert ref == host

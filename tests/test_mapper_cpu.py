"""CPU test of the HOST LOGIC only: the mapper (regions, chaining, extension bookkeeping, CIGAR, MAPQ, SAM) linked
against the oracle must reproduce the compiled reference's SAM byte for byte on every golden configuration.
(The product binary links libalfa_hot instead; its parity is tests/test_gpu_mapper.py.)"""
import os

import pytest

import mapper_util as M


@pytest.mark.skipif(not os.path.exists(M.CPU_EXE), reason="tests/_cpu_mapper not built (run __graft_entry__.build())")
@pytest.mark.parametrize("name", sorted(M.CONFIGS))
def test_host_mapper_with_oracle_matches_reference_sam(name):
    body = M.run_config(M.CPU_EXE, name, ["-t", "2", "--fibers", "64"])
    assert M.md5(body) == M.golden_md5(name)
    assert body == M.golden_body(name)


@pytest.mark.skipif(not os.path.exists(M.CPU_EXE), reason="tests/_cpu_mapper not built")
def test_output_independent_of_threads_and_fibers():
    a = M.run_config(M.CPU_EXE, "pe_s12", ["-t", "1", "--fibers", "1"])
    b = M.run_config(M.CPU_EXE, "pe_s12", ["-t", "5", "--fibers", "7"])
    assert a == b == M.golden_body("pe_s12")


@pytest.mark.skipif(not os.path.exists(M.CPU_EXE), reason="tests/_cpu_mapper not built")
@pytest.mark.parametrize("name", sorted(M.SYNTH_CONFIGS))
def test_host_mapper_paired_modes_synthetic(name):
    """3 000 synthetic pairs, one mate often seedless: paired modes 1-6; --paired-rescue issues hundreds of ksw_align calls"""
    body, err = M.run_synth(M.CPU_EXE, name, ["-t", "8", "--stats"])
    assert M.md5(body) == M.synth_md5(name)
    assert ("rescue" not in name) or " align 0/0" not in err


@pytest.mark.skipif(not os.path.exists(M.CPU_EXE), reason="tests/_cpu_mapper not built")
@pytest.mark.parametrize("name", sorted(M.SYNTH_SE_CONFIGS))
def test_host_mapper_option_sweep_synthetic(name):
    """2 700 synthetic single-end reads (two contigs with repeats, short and long reads): the options that steer the host logic"""
    assert M.md5(M.run_synth_single(M.CPU_EXE, name, ["-t", "8"])) == M.synth_md5(name)


@pytest.mark.skipif(not os.path.exists(M.CPU_EXE), reason="tests/_cpu_mapper not built")
def test_host_mapper_c5_paired_long_reads():
    """C5-shaped: paired 5 kbp reads, insert 12 000 +- 2 000, 10 % error, -I 8000 -X 16000 -e 0.12 (md5 of the compiled reference's SAM)"""
    import json
    import golden_util as G
    d = G.tmpdir()
    M.write_c5_paired(d)
    gold = json.load(open(os.path.join(G.GOLD, "c5_paired_md5.json")))
    assert M.md5(M.run_c5_paired(M.CPU_EXE, d, threads="8")) == gold["md5"]


@pytest.mark.skipif(not os.path.exists(M.CPU_EXE), reason="tests/_cpu_mapper not built")
def test_parallel_fastq_parse_equals_general_parser():
    """the reader thread only splits plain 4-line FASTQ records (workers build the reads); anything else -- CRLF,
    multi-line records, FASTA, junk between records -- goes through the kseq restatement; gzip input is inflated chunk by chunk on
    the reader thread and split the same way.
    Same SAM either way, and the same as with the fast path switched off."""
    import gzip
    import subprocess
    import golden_util as G
    d = M.workdir()
    recs = open(os.path.join(d, "454reads.fq")).read().rstrip("\n").split("\n")
    recs = [recs[i:i + 4] for i in range(0, len(recs), 4)][:40]
    variants = {
        "plain": "".join("\n".join(r) + "\n" for r in recs),
        "no_final_newline": "".join("\n".join(r) + "\n" for r in recs).rstrip("\n"),
        "crlf_from_10": "".join(("\r\n" if i >= 10 else "\n").join(r) + ("\r\n" if i >= 10 else "\n") for i, r in enumerate(recs)),
        "multiline_from_5": "".join((r[0] + "\n" + r[1][:50] + "\n" + r[1][50:] + "\n+\n" + r[3][:30] + "\n" + r[3][30:] + "\n") if i >= 5
                                    else "\n".join(r) + "\n" for i, r in enumerate(recs)),
        "names_with_comments": "".join("\n".join([r[0] + "/1 some comment\tx", r[1], "+" + r[0][1:], r[3]]) + "\n" for r in recs),
        "junk_between": "".join("\n".join(r) + "\n\n" for r in recs),
        "fasta": "".join(">" + r[0][1:] + "\n" + r[1] + "\n" for r in recs),
    }
    pre = os.path.join(d, "idx_pp")
    subprocess.run([M.CPU_EXE, "index", "-r", "reference.fasta", "-p", pre, "-s", "12"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)

    def run(path, fast):
        out = os.path.join(d, "pp.sam")
        env = dict(os.environ)
        if not fast:
            env["ALFA_NO_FAST_FASTQ"] = "1"
        p = subprocess.run([M.CPU_EXE, "align", "-i", pre, "-0", path, "-o", out, "-t", "3"], cwd=d, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        return M.sam_body(out)
    bodies = {}
    for name, txt in variants.items():
        path = os.path.join(d, "pp_%s.fq" % name)
        open(path, "w", newline="").write(txt)
        a, b = run(path, True), run(path, False)
        assert a == b, name
        assert a.count(b"\n") >= 40, name
        bodies[name] = a
    # gzip: the reader inflates chunk by chunk and splits the chunks (records cut by a chunk end, the general parser taking
    # over in the middle of a chunk, a file larger than one chunk)
    for name in ("plain", "no_final_newline", "crlf_from_10", "multiline_from_5", "fasta"):
        with gzip.open(os.path.join(d, "pp_%s.fq.gz" % name), "wt", newline="") as f:
            f.write(variants[name])
        assert run(os.path.join(d, "pp_%s.fq.gz" % name), True) == bodies[name], name
    big = variants["plain"] * 3000                                      # ~40 MB: more than one 32 MiB chunk
    with gzip.open(os.path.join(d, "pp_big.fq.gz"), "wt", compresslevel=1) as f:
        f.write(big)
    assert len(big) > (33 << 20)
    a, b = run(os.path.join(d, "pp_big.fq.gz"), True), run(os.path.join(d, "pp_big.fq.gz"), False)
    assert a == b and a.count(b"\n") >= 40 * 3000
    os.remove(os.path.join(d, "pp_big.fq.gz"))
    assert bodies["no_final_newline"] == bodies["plain"] == bodies["junk_between"]


@pytest.mark.skipif(not os.path.exists(M.CPU_EXE), reason="tests/_cpu_mapper not built")
@pytest.mark.parametrize("name", ["454_s12", "pe_s12", "pe_s12_rescue"])
def test_multi_gpu_worker_groups_keep_output_and_order(name):
    """`--gpus G` binds worker thread t to hot-path context t mod G (one index copy per context), batches are split
    over all workers and written in input order.  With G = 2 and 3 logical contexts (oracle-bound here) the SAM must be the
    reference's, byte for byte, whatever the thread count."""
    gold = M.golden_body(name)
    for gpus, threads in ((2, "4"), (3, "5"), (2, "2")):
        assert M.run_config(M.CPU_EXE, name, ["-t", threads, "--gpus", str(gpus)]) == gold, (gpus, threads)


def test_fiber_stack_overflow_is_reported_not_silent():
    """hot_sched.cpp: a mapping fiber that outgrows its stack is reported (poisoned zone at the bottom of every stack, checked at
    each switch) or faults on the inaccessible gap in front of its group of stacks; a fiber that stays inside runs clean."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cpu_mapper")
    ok = subprocess.run([exe, "--fiber-overflow-test", str(40 * 1024)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert ok.returncode == 0 and b"no overflow detected" in ok.stdout
    # 64 KiB stacks: 100 KiB of frames runs into the neighbour below -> reported by the canary check (abort)
    bad = subprocess.run([exe, "--fiber-overflow-test", str(100 * 1024)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode != 0
    assert b"overflowed" in bad.stderr or bad.returncode in (-11, -7, -6, 139, 135, 134), (bad.returncode, bad.stderr[-300:])
    # far beyond the group (16 stacks of 64 KiB = 1 MiB): must hit the inaccessible gap, i.e. a fault, never silent success
    worse = subprocess.run([exe, "--fiber-overflow-test", str(4 << 20)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert worse.returncode != 0 and b"no overflow detected" not in worse.stdout


# ---- csrc/map_core.h: the source of the mapping kernel's trial loop, compiled here with g++ and driven by the host mapper
# (`--core-trials`) with the oracle's DP: same SAM as the reference
@pytest.mark.parametrize("name", ["454_s12", "454_s1", "pe_s12", "454_s1_local", "454_s12_e12", "454_s1_a5", "pe_s12_rescue_a3"])
def test_map_core_trial_loop_matches_reference_sam(name):
    body = M.run_config(M.CPU_EXE, name, ["--core-trials", "-t", "2"])
    assert M.md5(body) == M.golden_md5(name)


@pytest.mark.parametrize("name", ["synse_default", "synse_scores", "synse_band_trials", "synse_minlen", "synse_mem_local"])
def test_map_core_trial_loop_option_sweep(name):
    body = M.run_synth_single(M.CPU_EXE, name, ["--core-trials", "-t", "4"])
    assert M.md5(body) == M.synth_md5(name)


def test_map_core_trial_loop_paired_and_its_hand_backs():
    import re
    body, err = M.run_synth(M.CPU_EXE, "synpe_rescue", ["--core-trials", "-t", "4", "--stats"])
    assert M.md5(body) == M.synth_md5("synpe_rescue")
    m = re.search(r"#CORE reads_done (\d+) handed_back (\d+)", err)
    assert m and int(m.group(1)) > 5000 and 0 < int(m.group(2)) < 200        # the re-seeding rescue is handed back


def test_map_core_sort_restates_libstdcxx():
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cpu_mapper")
    p = subprocess.run([exe, "--sort-selftest"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"sort selftest OK" in p.stdout, p.stdout[-300:]

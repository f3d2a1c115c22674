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

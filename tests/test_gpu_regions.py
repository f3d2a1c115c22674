"""GPU parity: alfa_seed_regions_batch (seeds -> candidate regions on the device) against the CPU
restatement of identifyAndFilter / coveredBases (oracle/regions_oracle.py), and the command line with --device-regions
against the reference's SAM."""
import os
import sys

import numpy as np
import pytest

from alfalfa_amd import ffi
import golden_util as G
import mapper_util as M
import seqgen

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import regions_oracle as RO  # noqa: E402

pytestmark = pytest.mark.gpu


def _check(dix, reads, min_len, sparse_q, edit, **prm):
    q = ffi.make_seed_queries(reads, min_len, 10000, sparse_q)
    m, moff, reg, roff, mem, boff, status, chains = dix.seed_regions_batch(ffi.SEED_SMAM, q, edit, **prm)
    err = prm.get("error_percent", 0.08)
    m2, moff2 = dix.seed_batch(ffi.SEED_SMAM, q)
    assert np.array_equal(moff, moff2) and np.array_equal(m, m2)
    ndev = 0
    for i, r in enumerate(reads):
        mine = [(int(x["ref"]), int(x["query"]), int(x["len"])) for x in m[moff[i]:moff[i + 1]]]
        exp, amb = RO.identify_and_filter(mine, len(r), int(edit[i]), prm.get("max_seed_candidates", 10000), prm.get("max_region_mems", 20),
                                          prm.get("min_mem_reg_length", 50))
        scores = []
        for e in exp:
            sc, a2 = RO.covered_bases(e[2])
            amb |= a2
            scores.append(sc)
        if status[i]:
            assert amb or len(mine) > 512 or len(exp) > 128 or sum(len(e[2]) for e in exp) > 2 * len(mine) + 8, (i, len(mine), len(exp))
            assert roff[i + 1] == roff[i]
            continue
        assert not amb, i
        ndev += 1
        got = reg[roff[i]:roff[i + 1]]
        assert len(got) == len(exp), (i, len(got), len(exp))
        for g, e, sc in zip(got, exp, scores):
            assert (int(g["ref_begin"]), int(g["ref_end"]), int(g["score"]), int(g["count"])) == (e[0], e[1], sc, len(e[2])), (i, g, e[:2], sc)
            mm = mem[boff[i] + g["first"]:boff[i] + g["first"] + g["count"]]
            assert [(int(x["ref"]), int(x["query"]), int(x["len"])) for x in mm] == e[2], i
            chain, anchor, ced, ccov = RO.first_chain(e[2], len(r), err)
            got_chain = chains[boff[i] + g["first"]:boff[i] + g["first"] + g["chain_count"]].tolist()
            assert (got_chain, int(g["chain_anchor"]), int(g["chain_edit"]), int(g["chain_coverage"])) == (chain, anchor, ced, ccov), (i, got_chain, chain)
    return ndev


def test_regions_random_reference(gpu):
    rng = np.random.default_rng(21)
    ref = seqgen.rand_ref(rng, 400000)
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref], K=12)
    dix = ffi.DeviceIndex(hidx, 0)
    reads = seqgen.sim_reads(rng, ref, 1500, 100, 400, 0.03) + seqgen.sim_reads(rng, ref, 60, 1500, 4000, 0.08)
    reads += [b"", ref[1000:1030], seqgen.rand_ref(rng, 200)]
    edit = np.array([int(0.08 * len(r)) + 1 for r in reads], np.int32)
    n = _check(dix, reads, 20, 1, edit)
    assert n > 1400
    _check(dix, reads, 40, 2, edit, max_seed_candidates=3, max_region_mems=2, min_mem_reg_length=30, error_percent=0.21)


def test_regions_repeats(gpu):
    """tandem and dispersed repeats: many matches per read, reserves, merged windows, strands handed back to the host"""
    rng = np.random.default_rng(22)
    a = seqgen.rand_ref(rng, 150000)
    rep = seqgen.rand_ref(rng, 300)
    unit = seqgen.rand_ref(rng, 37)
    ref = a[:30000] + rep + a[30000:60000] + unit * 40 + a[60000:90000] + rep + a[90000:120000] + rep + a[120000:]
    hidx = ffi.HostIndex.build_mem(["chr1"], [ref], K=3)
    dix = ffi.DeviceIndex(hidx, 0)
    reads = seqgen.sim_reads(rng, ref, 800, 150, 600, 0.02)
    reads += [ref[29900:30500], ref[60000 - 100:60000 + 37 * 40 + 100], rep, unit * 6]
    edit = np.array([int(0.08 * len(r)) + 1 for r in reads], np.int32)
    _check(dix, reads, 20, 1, edit)
    _check(dix, reads, 25, 2, edit, max_seed_candidates=5, max_region_mems=4, min_mem_reg_length=40)


@pytest.mark.parametrize("name", ["454_s12", "pe_s12_rescue", "454_s1", "454_s12_e12", "pe_s12_mode3", "pe_s1_mode6_rescue"])
def test_cli_device_regions_matches_reference_sam(gpu, name):
    assert M.md5(M.run_config(M.GPU_EXE, name, ["-t", "4", "--device-regions"])) == M.golden_md5(name)


@pytest.mark.parametrize("name", sorted(M.SYNTH_SE_CONFIGS))
def test_cli_device_regions_synthetic(gpu, name):
    """two contigs with repeats, short and long reads: md5 of the reference's records"""
    assert M.md5(M.run_synth_single(M.GPU_EXE, name, ["-t", "4", "--device-regions"])) == M.synth_md5(name)

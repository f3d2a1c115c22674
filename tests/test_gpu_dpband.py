"""GPU parity: the dynProg kernels (dpBandStatic / dpBandFull / dpBandScore / dpBasic through alfa_dp_band_batch) vs the CPU
oracle on random cases: every dp_type flag combination the variant accepts, affine and linear scores, symmetric and
asymmetric bands, empty and ragged inputs, upper/lower case and non-letter bytes."""
import numpy as np
import pytest

from alfalfa_amd import ffi
import oracle_ffi as O
import dpgen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_dpband_random(gpu, variant):
    rng = np.random.default_rng(50 + variant)
    n = 0
    for (m, mm, o, e) in dpgen.DPBAND_SCORES:
        if variant == 2 and o == 0:
            continue
        sc = ffi.DpbandScores(m, mm, o, e, int(o != 0))
        specs = dpgen.dpband_cases(rng, 250, variant)
        tasks = ffi.make_dpband_tasks(specs)
        got = ffi.dp_band_batch(variant, sc, tasks)
        exp = O.orc_dp_band_batch(variant, sc, tasks)
        for k, (g, x) in enumerate(zip(got, exp)):
            assert g == x, (variant, (m, mm, o, e), specs[k], g, x)
        n += len(specs)
    assert n >= 700


def test_dpband_rejects_what_the_reference_asserts_on(gpu):
    sc = ffi.DpbandScores(1, -4, -6, -1, 1)
    bad = [
        (0, dict(ref=b"acgt", query=b"acgt", flags=(0, 0, 0, 0, 1))),               # dpBandStatic: local
        (0, dict(ref=b"acgt", query=b"acgt", flags=(1, 1, 0, 0, 0))),               # free begin + free end
        (0, dict(ref=b"acgt", query=b"acgt", band1=0)),
        (1, dict(ref=b"acgt", query=b"acgt", band1=3, band2=0)),
        (1, dict(ref=b"acgt", query=b"acgt", refE=9)),                              # boundary outside the string
        (3, dict(ref=b"ac\xf0t", query=b"acgt")),                                   # scoreMatrix is 127 x 127
    ]
    for variant, sp in bad:
        with pytest.raises(ffi.AlfaError):
            ffi.dp_band_batch(variant, sc, ffi.make_dpband_tasks([sp]))
    with pytest.raises(ffi.AlfaError):
        ffi.dp_band_batch(2, ffi.DpbandScores(1, -1, 0, -1, 0), ffi.make_dpband_tasks([dict(ref=b"acgt", query=b"acgt")]))
    assert ffi.dp_band_batch(0, sc, ffi.make_dpband_tasks([])) == []

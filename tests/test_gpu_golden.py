"""GPU parity against the committed golden vectors (generated from the compiled reference)."""
import pytest

from alfalfa_amd import ffi
import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [1, 12])
def test_gpu_seeds_vs_reference_goldens(gpu, K):
    cache = {}

    def fn(hidx, mt, q):
        if id(hidx) not in cache:
            cache[id(hidx)] = ffi.DeviceIndex(hidx, gpu)
        return cache[id(hidx)].seed_batch(mt, q)
    assert G.check_seeds(fn, K, types=(0, 1, 2, 3)) > 300


def test_gpu_ksw_vs_captured_mapper_calls(gpu):
    assert G.check_ksw_calls(ffi.ksw_extend_batch, ffi.ksw_nw_batch, ffi.ksw_global_batch) > 3000


def test_gpu_ksw_vs_reference_edge_cases(gpu):
    assert G.check_ksw_random(ffi.ksw_extend_batch, ffi.ksw_nw_batch, ffi.ksw_global_batch) == 1500


def test_gpu_dynprog_vs_reference_goldens(gpu):
    assert G.check_dpband(ffi.dp_band_batch) > 1000

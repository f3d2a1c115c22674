"""CPU test: libalfa_hot loads without a GPU and exports every symbol declared in include/*.h;
compute entry points fail loudly (ALFA_ENODEV) instead of falling back to the CPU."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

from alfalfa_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        for m in re.finditer(r"\b(alfa_[a-z0-9_]+)\s*\(", txt):
            names.add(m.group(1))
    return sorted(names)


def test_every_declared_symbol_is_exported():
    lib = ffi.lib()
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_gpu():
    lib = ffi.lib()
    if lib.alfa_device_count() > 0:
        pytest.skip("a GPU is visible")
    P = ffi.default_params()
    tasks, keep = ffi.make_dp_tasks([(np.zeros(5, np.uint8), np.zeros(5, np.uint8))], 3, 0)
    out = np.zeros(1, np.int32)
    rc = lib.alfa_ksw_nw_batch(C.byref(P), tasks, 1, out.ctypes.data_as(C.c_void_p))
    assert rc == -19
    assert b"no CPU fallback" in lib.alfa_last_error()
    import golden_util as G
    hidx = G.lambda_index(12)
    with pytest.raises(ffi.AlfaError):
        ffi.DeviceIndex(hidx, 0)


def test_product_does_not_reference_oracle():
    """the product sources must never include or link anything under oracle/"""
    for f in glob.glob(os.path.join(ROOT, "alfalfa_amd", "**", "*"), recursive=True):
        if os.path.isfile(f) and f.endswith((".py", ".cpp", ".hip", ".h")):
            txt = open(f, errors="ignore").read()
            assert "oracle/" not in txt and "liboracle" not in txt and "alfa_oracle" not in txt, f


@pytest.mark.parametrize("knob,before,expect", [(None, None, "16"), (None, "4", "4"), ("8", None, "8"), ("8", "4", "8"), ("0", "5", "5"), ("0", None, None)])
def test_hw_queue_default_is_set_at_load(knob, before, expect):
    """the host chooses the hardware-queue count: loading the library never touches a GPU_MAX_HW_QUEUES that is already set;
    it installs 16 only when the variable is absent, and ALFA_HW_QUEUES=<n> is the explicit request that wins (0 = hands off):
    csrc/device_ctx.hip.  Checked in a child process through libc's getenv (the variable is set from C)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("ALFA_HW_QUEUES", "GPU_MAX_HW_QUEUES")}
    if knob is not None:
        env["ALFA_HW_QUEUES"] = knob
    if before is not None:
        env["GPU_MAX_HW_QUEUES"] = before
    code = ("import ctypes, sys; ctypes.CDLL(sys.argv[1]); libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "v = libc.getenv(b'GPU_MAX_HW_QUEUES'); print('unset' if v is None else v.decode())")
    out = subprocess.run([sys.executable, "-c", code, ffi.LIB_PATH], env=env, stdout=subprocess.PIPE, check=True).stdout.decode().strip()
    assert out == ("unset" if expect is None else expect)

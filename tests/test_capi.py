"""The C-ABI libraries load and export every symbol include/mpcqp.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

import mpcqp
from conftest import REPO, _have_gpu


def _declared_symbols():
    hdr = open(os.path.join(REPO, "include", "mpcqp.h")).read()
    return sorted(set(re.findall(r"\b(mpcqp_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_match_binding():
    assert set(_declared_symbols()) == set(mpcqp._capi.EXPORTED_SYMBOLS)


def test_product_library_exports_all_symbols():
    lib = mpcqp.product_library()            # fails loudly if the HIP library has not been built
    for sym in _declared_symbols():
        assert hasattr(lib.lib, sym), sym
    assert lib.version() == 0x00010300


def test_oracle_exports_same_symbols(oracle_lib):
    for sym in _declared_symbols():
        assert hasattr(oracle_lib.lib, sym), sym


def test_config_struct_layout_and_defaults(oracle_lib):
    plib = mpcqp.product_library()
    for lib in (plib, oracle_lib):
        cfg = lib.default_config()
        assert cfg.size == ctypes.sizeof(mpcqp.MpcQpConfig)
        # Lite3 constants hard-coded in the reference (src/mpc.py:45-46,71-76,122-134)
        assert cfg.N == 10 and abs(cfg.delta - 0.03) < 1e-15
        assert cfg.m == 8.885 and list(cfg.Ibody_inv) == [1 / 0.24, 1.0, 1.0]
        assert list(cfg.w) == [1e4, 2.7e4, 1e4, 2.7e5, 2.7e5, 2.7e5, 1e4, 1e4, 1e4, 1.6e4, 1.6e4, 1.6e4, 0.0]
        assert (cfg.f_min, cfg.f_max) == (3.0, 100.0)
        assert cfg.disc == mpcqp.DISC_EULER


def test_host_layer_scales_the_admm_block_with_the_horizon(oracle_lib):
    """The C default (100 iterations per block, cap 400) is tuned for N = 10; the Python host layer uses 10 N / 40 N for
    other horizons unless the caller says otherwise (include/mpcqp.h, `check_every`)."""
    for lib in (oracle_lib, mpcqp.product_library()):
        c10, c20 = lib.default_config(), lib.default_config(N=20)
        assert (c20.check_every, c20.max_iter, c20.polish_max) == (2 * c10.check_every, 2 * c10.max_iter, 2 * c10.polish_max)
        assert lib.default_config(N=20, polish_max=3).polish_max == 3
        c = lib.default_config(N=20, check_every=50)
        assert (c.check_every, c.max_iter) == (50, c10.max_iter)
        assert lib.default_config(N=20, max_iter=1000).max_iter == 1000
    p10 = mpcqp.product_library().default_config()
    assert (p10.check_every, p10.max_iter) == (100, 400)
    # other horizons run on the stage-wise engine: blocks of 5 N / 3 iterations (its first block is Anderson-accelerated), the polish budget of N = 20
    p60 = mpcqp.product_library().default_config(N=60, delta=0.01)
    assert (p60.check_every, p60.max_iter, p60.polish_max) == (100, 2400, 8)
    assert p60.accel == 0 and mpcqp.product_library().default_config(accel=-1).accel == -1     # (0 = the engine's default: on)
    flags = mpcqp.product_library().default_config(
        flags=mpcqp.FLAG_POLISH | mpcqp.FLAG_WARM_START | mpcqp.FLAG_WARM_SHIFT | mpcqp.FLAG_NATURAL_ORDER).flags
    assert flags == 1 | 2 | 16 | 8


def test_product_library_reads_no_environment():
    """Solver behaviour is a function of MpcQpConfig alone: the product library does not import getenv (round-2 review: developer
    knobs read from the environment could silently change a caller's results)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", mpcqp.product_library().path], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in out


def test_product_never_falls_back_to_cpu():
    """Without a gfx950 device the product library refuses to create an engine (MPCQP_ENODEV), it does not emulate."""
    if _have_gpu():
        pytest.skip("GPU present")
    lib = mpcqp.product_library()
    with pytest.raises(mpcqp.MpcQpError, match="-4"):
        mpcqp.Engine(lib, lib.default_config())
    with pytest.raises(mpcqp.MpcQpError):
        mpcqp.MPCBatch()


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(mpcqp.MpcQpError, match="not found"):
        mpcqp.Library(str(tmp_path / "libmpcqp.so"))


def test_bad_config_rejected(oracle_lib):
    cfg = oracle_lib.default_config()
    cfg.size = 8
    with pytest.raises(mpcqp.MpcQpError):
        mpcqp.Engine(oracle_lib, cfg)
    plib = mpcqp.product_library()
    for kw in (dict(N=0), dict(N=65), dict(precision=9), dict(relax=2.5), dict(delta=-1.0), dict(disc=5)):
        with pytest.raises(mpcqp.MpcQpError, match="-1"):
            mpcqp.Engine(plib, plib.default_config(**kw))

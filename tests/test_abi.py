"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol
include/mpcx.h declares, validates arguments, and its variant table matches the oracle's."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from oracle import mpc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from model_predictive_control_amd import _lib
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mpcx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcx_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_typed(lib):
    from model_predictive_control_amd import _lib
    names = _declared_symbols()
    assert "mpcx_solve_batch" in names and "mpcx_nearest_ref_batch" in names and len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mpcx.h but not exported"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.mpcx_abi_version() == _lib.ABI_VERSION == 2


def test_struct_layout_matches_header():
    from model_predictive_control_amd import _lib
    # 6 int32 + (2 + 4 + 4 + 3 + 1 + 3 + 4 + 4 + 4 + 1 + 4) doubles
    assert C.sizeof(_lib.CVariant) == 6 * 4 + 34 * 8


@pytest.mark.parametrize("name", list(O.VARIANTS))
def test_variant_table_matches_oracle(lib, name):
    from model_predictive_control_amd import variants
    v = variants.get(name)
    o = O.VARIANTS[name]
    assert (v.model, v.nu, v.T, v.eq_mode, v.chain) == (o.model, o.nu, o.T, o.eq_mode, o.chain)
    assert v.dt == o.dt and v.wb == o.wb and v.q_growth == o.q_growth and v.steer_fixed == o.steer_fixed
    for fld, pad in (("R", 0.0), ("Rd", 0.0), ("umax", math.inf), ("dumax", math.inf), ("first_step", math.inf),
                     ("cold_start", 0.0)):
        want = list(getattr(o, fld)) + [pad] * (4 - len(getattr(o, fld)))
        np.testing.assert_allclose(list(getattr(v, fld)), want, rtol=1e-15, err_msg=fld)
    assert list(v.Q) == list(o.Q) and list(v.Qf) == list(o.Qf)


def test_argument_validation_without_gpu(lib):
    from model_predictive_control_amd import _lib, variants
    bad = _lib.CVariant()
    assert lib.mpcx_variant_by_name(b"no_such_variant", C.byref(bad)) == -1
    assert b"unknown variant" in lib.mpcx_last_error()
    v = variants.get("diff_drive")
    assert lib.mpcx_workspace_bytes(C.byref(v), 65536) == 0            # T = 5: register-resident
    v30 = variants.get("dsr_ackermann", T=30)
    # small batch, T <= 32: one wavefront per instance in LDS, then a stage-wise pass over what that left unsettled
    assert lib.mpcx_workspace_bytes(C.byref(v30), 1000) == 55 * 30 * 8 * 1024
    big = lib.mpcx_workspace_bytes(C.byref(v30), 1 << 20)
    assert big == 262144 * 8 * 55 * 30                               # stage-wise kernel: 55 doubles per stage and lane, one launch of at most 262,144
    v40 = variants.get("dsr_ackermann", T=40)
    assert lib.mpcx_workspace_bytes(C.byref(v40), 1000) > 0
    vbad = variants.get("diff_drive", T=65)
    assert lib.mpcx_solve_batch(C.byref(vbad), 1, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.mpcx_solve_batch(C.byref(v), 4, None, None, None, None, None, None, None, None, None, None) == -1
    assert b"NULL" in lib.mpcx_last_error()
    vbad = variants.get("dsr_crab", nu=2)
    assert lib.mpcx_rollout_batch(C.byref(vbad), 1, None, None, None, None) == -1
    assert lib.mpcx_nearest_ref_batch(2, 5, 0.2, 2.0, 10, 1, 1, None, None, None, None, None, None, None, 1, 1,
                                      None, None, None, None, None) == -1


def test_chain_follows_horizon_override():
    from model_predictive_control_amd import variants
    assert variants.get("sim_ackermann", T=8).chain == 8
    assert variants.get("dsr_ackermann", T=8).chain == 1


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from model_predictive_control_amd import _lib, engine, variants, mpc_diff_drive
    with pytest.raises(_lib.MpcxError):
        engine.solve_batch(variants.get("diff_drive"), np.zeros((1, 3)), np.zeros((1, 3, 6)))
    with pytest.raises(_lib.MpcxError):
        mpc_diff_drive.get_linear_model_matrix(0.3, 0.7)


def test_header_is_plain_c_and_links(tmp_path):
    """A C99 program that includes include/mpcx.h, links the library and calls the entry points
    that need no GPU: what a cgo / JNI / C caller of the boundary would do."""
    import os
    import shutil
    import subprocess
    from model_predictive_control_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    _lib.load()
    libdir = os.path.join(root, "model_predictive_control_amd", "lib")
    src = tmp_path / "caller.c"
    src.write_text(
        "#include <stdio.h>\n#include <string.h>\n#include \"mpcx.h\"\n"
        "int main(void) {\n"
        "    mpcx_variant v;\n"
        "    if (mpcx_abi_version() != MPCX_ABI_VERSION) return 1;\n"
        "    if (mpcx_variant_by_name(\"dsr_ackermann\", &v) != 0) return 2;\n"
        "    if (v.T != 5 || v.nu != 4) return 3;\n"
        "    if (mpcx_workspace_bytes(&v, 65536) != 0) return 4;\n"
        "    if (mpcx_variant_by_name(\"nope\", &v) == 0 || !strstr(mpcx_last_error(), \"unknown\")) return 5;\n"
        "    puts(\"ok\");\n    return 0;\n}\n")
    exe = tmp_path / "caller"
    subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src),
                    "-L", libdir, "-lmpcx", "-Wl,-rpath," + libdir, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=60)
    assert out.stdout.strip() == "ok"

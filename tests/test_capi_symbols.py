"""The C-ABI library loads and exports every symbol include/moby_hip.h
declares; argument validation and host helpers work without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="moby_hip.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from moby_amd import _lib
    lib = _lib.load()
    declared = sorted(set(_declared_symbols() + _declared_symbols("moby_hip_impact.h") + _declared_symbols("moby_hip_stack.h") + _declared_symbols("moby_hip_artic.h")))
    assert "mh_lcp_solve_batch_dev" in declared and "mh_rand_seed" in declared and "mh_impact_batch_process" in declared
    for name in declared:
        assert hasattr(lib, name), "libmoby_hip.so does not export %s" % name
    assert set(declared) == set(_lib.SYMBOLS), "ctypes table out of sync with moby_hip.h / moby_hip_impact.h"
    assert lib.mh_version() >= 101          # 101: lu_work fills four columns per world (include/moby_hip.h)


def test_io_library_exports_every_declared_symbol():
    from moby_amd import io as mio
    lib = mio.load()
    declared = [n for n in _declared_symbols("moby_hip_io.h") if n.startswith("mh_io_")]
    assert sorted(declared) == ["mh_io_compare_trajs", "mh_io_format_row", "mh_io_last_error", "mh_io_load_sdf", "mh_io_load_urdf", "mh_io_load_xml", "mh_io_load_xml_artic"]
    for name in declared:
        assert hasattr(lib, name), "libmoby_hip_io.so does not export %s" % name


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both public headers compile as C99 (-pedantic -Werror) and as C++11."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "moby_hip.h"\n#include "moby_hip_io.h"\n#include "moby_hip_impact.h"\n#include "moby_hip_stack.h"\n#include "moby_hip_artic.h"\n'
                   'int main(void) { mh_scene s; mh_world_aux a; mh_io_scene io; mh_contact c; (void)s; (void)a; (void)io; (void)c;\n'
                   '  return (int)sizeof(mh_lcp_opts) == 0 || sizeof(mh_contact) != 96; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + inc, "-fsyntax-only", str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I" + inc, "-fsyntax-only", "-x", "c++", str(src)])


def test_ctypes_mirrors_have_the_c_struct_sizes(tmp_path):
    """moby_amd/{scene,stack,artic,io}.py mirror the structs of include/*.h by hand: compile a probe that prints sizeof / a few
    offsets and compare (a stale mirror would shift every field after the first mismatch)."""
    import subprocess
    from moby_amd import artic as A, scene as S, stack as K
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "moby_hip.h"\n#include "moby_hip_impact.h"\n#include "moby_hip_stack.h"\n#include "moby_hip_artic.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mh_scene), sizeof(mh_world_aux), sizeof(mh_big_scene), sizeof(mh_artic_model),\n'
                   '  offsetof(mh_artic_model, nspheres), offsetof(mh_artic_model, plane_R), offsetof(mh_artic_model, cp_nk)); return 0; }\n')
    exe = str(tmp_path / "sz")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = [int(x) for x in subprocess.check_output([exe]).split()]
    M = A.mh_artic_model
    want = [ctypes.sizeof(S.mh_scene), ctypes.sizeof(S.mh_world_aux), ctypes.sizeof(K.mh_big_scene), ctypes.sizeof(M), M.nspheres.offset, M.plane_R.offset, M.cp_nk.offset]
    assert got == want, (got, want)


def test_device_selection_is_validated_without_a_gpu():
    """Device affinity behind the C ABI (moby_hip.h, Devices): a batch lives on the device current at create and every entry
    point switches to it.  Without a GPU the selection entry points must refuse instead of crashing; with one, an index
    outside [0, mh_device_count()) is MH_ERR_INVALID_ARG and a null handle has no device."""
    from moby_amd import _lib
    lib = _lib.load()
    n = lib.mh_device_count()
    assert n >= 0
    if n == 0:
        assert lib.mh_device_set(0) == _lib.MH_ERR_NO_DEVICE and lib.mh_device_get() == _lib.MH_ERR_NO_DEVICE
        assert b"no HIP device" in lib.mh_last_error()
    else:
        assert lib.mh_device_set(n) == _lib.MH_ERR_INVALID_ARG and lib.mh_device_set(-1) == _lib.MH_ERR_INVALID_ARG
        assert lib.mh_device_set(0) == _lib.MH_OK and lib.mh_device_get() == 0
    for f in (lib.mh_world_batch_device, lib.mh_big_batch_device, lib.mh_artic_batch_device, lib.mh_impact_batch_device):
        assert f(None) == _lib.MH_ERR_INVALID_ARG


def test_multi_gpu_adapter_compiles_against_rccl(tmp_path):
    """moby_amd/cpp/example_multi_gpu.cpp (one process, one batch + stream per device, RCCL ncclAllReduce of the counters) builds
    in the CPU container; it runs inside `-m gpu` (tests/test_world_gpu.py::test_cpp_multi_gpu_example)."""
    import subprocess
    cpp = os.path.join(ROOT, "moby_amd", "cpp")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", os.path.join(cpp, "example_multi_gpu.cpp"), "-L" + os.path.join(ROOT, "moby_amd"),
                           "-lmoby_hip", "-lmoby_hip_io", "-lrccl", "-Wl,-rpath," + os.path.join(ROOT, "moby_amd"), "-o", str(tmp_path / "example_multi_gpu")])


def test_rand_seed_matches_libc():
    from moby_amd import _lib
    lib = _lib.load()
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 7):
        st = np.zeros(32, dtype=np.uint32)
        lib.mh_rand_seed(st.ctypes.data, seed)
        libc.srand(seed)
        assert [libc.rand() for _ in range(1000)] == [lib.mh_rand_next(st.ctypes.data) for _ in range(1000)]


def test_python_srand_state_equals_the_library():
    from moby_amd import _lib, scene as S
    lib = _lib.load()
    for seed in (1, 7, 0, 123456789, 2 ** 31 + 5):
        st = np.zeros(32, dtype=np.uint32)
        lib.mh_rand_seed(st.ctypes.data, ctypes.c_uint32(seed))
        np.testing.assert_array_equal(S.glibc_srand_state(seed), st, err_msg=str(seed))


def test_rand_state_layout_equals_oracle(oracle):
    from moby_amd import _lib
    lib = _lib.load()
    st = np.zeros(32, dtype=np.uint32)
    lib.mh_rand_seed(st.ctypes.data, 1)
    np.testing.assert_array_equal(st, oracle.rand_state(1))


def test_argument_validation_without_gpu():
    from moby_amd import _lib
    lib = _lib.load()
    buf = np.zeros(64)
    rc = lib.mh_lcp_solve_batch_dev(None, 99, 1, 2, buf.ctypes.data, 2, 4, buf.ctypes.data, buf.ctypes.data,
                                    None, None, buf.ctypes.data, buf.ctypes.data, None, None, 0, None, None)
    assert rc == _lib.MH_ERR_INVALID_ARG and b"kind" in lib.mh_last_error()
    rc = lib.mh_lcp_solve_batch_dev(None, 0, 1, 4097, buf.ctypes.data, 4097, 4097 * 4097, buf.ctypes.data, buf.ctypes.data,
                                    None, None, buf.ctypes.data, buf.ctypes.data, None, None, 0, None, None)
    assert rc == _lib.MH_ERR_UNSUPPORTED_N
    rc = lib.mh_lcp_solve_batch_dev(None, 0, 1, 4, buf.ctypes.data, 2, 16, buf.ctypes.data, buf.ctypes.data,
                                    None, None, buf.ctypes.data, buf.ctypes.data, None, None, 0, None, None)
    assert rc == _lib.MH_ERR_INVALID_ARG


def test_scheduling_switches_validate_their_values_without_gpu():
    """mh_debug_set: every documented key takes its documented values and refuses others; defaults restored."""
    from moby_amd import _lib
    lib = _lib.load()
    ranges = {2: (0, 5), 3: (0, 1), 4: (0, 4), 5: (0, 1), 6: (0, 1), 7: (0, 1), 8: (0, 4), 9: (0, 1), 10: (0, 1), 11: (0, 1)}
    defaults = {2: 0, 3: 1, 4: 3, 5: 1, 6: 1, 7: 1, 8: 0, 9: 0, 10: 1, 11: 0}
    try:
        for key, (lo, hi) in ranges.items():
            for v in range(lo, hi + 1):
                assert lib.mh_debug_set(key, v) == _lib.MH_OK, (key, v)
            assert lib.mh_debug_set(key, hi + 1) == _lib.MH_ERR_INVALID_ARG and lib.mh_debug_set(key, lo - 1) == _lib.MH_ERR_INVALID_ARG, key
        assert lib.mh_debug_set(99, 0) != _lib.MH_OK
    finally:
        for key, v in defaults.items():
            assert lib.mh_debug_set(key, v) == _lib.MH_OK


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (no CPU fallback); neither do the measurement scripts under tools/
    (checkers that run next to the oracle live in tests/tools/)."""
    for top in ("moby_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp", ".sh")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert "liboracle" not in txt and "oracle_api" not in txt, f
                    assert not re.search(r'#include\s+"[^"]*oracle', txt), f


def test_cpp_adapter_compiles_and_fails_loudly_without_gpu(tmp_path):
    """The reference-side C++ adapter builds against include/moby_hip.h with plain
    g++; with no device the call reports an error instead of falling back."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    cpp = os.path.join(ROOT, "moby_amd", "cpp")
    exe = str(tmp_path / "example_lcp")
    subprocess.check_call(["g++", "-std=c++11", os.path.join(cpp, "example_lcp.cpp"), "-L" + os.path.join(ROOT, "moby_amd"),
                           "-lmoby_hip", "-Wl,-rpath," + os.path.join(ROOT, "moby_amd"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, timeout=120)
    assert p.returncode == 2 and b"no HIP device" in p.stdout


@pytest.mark.parametrize("example,libs", [("example_stack", ["-lmoby_hip"]), ("example_joints", ["-lmoby_hip"]), ("example_articulated", ["-lmoby_hip", "-lmoby_hip_io"])])
def test_new_cpp_adapters_compile_and_fail_loudly_without_gpu(tmp_path, example, libs):
    """MobyHipStackSimulator.h (seams B5 / B3 for large worlds) and MobyHipArticulatedBody.h (seam B4) build with plain
    g++ against the C ABI; without a device they report the error (no fallback)."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU tests")
    cpp = os.path.join(ROOT, "moby_amd", "cpp")
    exe = str(tmp_path / example)
    subprocess.check_call(["g++", "-std=c++11", "-Wall", os.path.join(cpp, example + ".cpp"), "-L" + os.path.join(ROOT, "moby_amd")] + libs +
                          ["-Wl,-rpath," + os.path.join(ROOT, "moby_amd"), "-o", exe])
    args = [exe] + ([os.path.join(ROOT, "tests", "scenes", "ten_joint_arm.sdf")] if example == "example_articulated" else [])
    p = subprocess.run(args, capture_output=True, timeout=120)
    assert p.returncode == 1 and b"no HIP device" in p.stdout, p.stdout

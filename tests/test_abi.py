"""The drop-in boundary: libidkpt.so / libidkbvh.so load, export every symbol include/*.h declares, the struct
layouts compile under a plain C compiler, and (without a GPU) the library refuses loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix, macro):
    src = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(macro + r"\s+[\w\s\*]+?\b(" + prefix + r"\w+)\s*\(", src)))


def test_idkpt_exports_every_declared_symbol():
    from idkengine_amd import _lib
    declared = _declared("idkpt.h", "idkpt", "IDKPT_API")
    assert len(declared) >= 30
    assert sorted(_lib.SYMBOLS) == declared                      # the Python binding covers the whole ABI
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = sorted(set(re.findall(r" T (idkpt\w+)", out)))
    assert exported == declared                                  # nothing undeclared leaks out either


def test_idkbvh_exports_every_declared_symbol():
    from idkengine_amd import bvh
    declared = _declared("idkbvh.h", "idkbvh", "IDKBVH_API")
    assert sorted(bvh.SYMBOLS) == declared
    out = subprocess.check_output(["nm", "-D", "--defined-only", bvh.LIB_PATH]).decode()
    assert sorted(set(re.findall(r" T (idkbvh\w+)", out))) == declared


def test_headers_are_plain_c(tmp_path):
    """include/*.h must be consumable by a C compiler (P/Invoke / cgo / JNI generators read C): all the byte-layout
    static asserts (sizes cited from Source/GpuTypes/*.cs) are checked by gcc here."""
    c = tmp_path / "t.c"
    c.write_text('#include "idkpt.h"\n#include "idkbvh.h"\nint main(void){return (int)sizeof(idkpt_scene_desc) > 0 ? 0 : 1;}\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(c)])


def test_numpy_mirrors_match_c_layout(tmp_path):
    from idkengine_amd import gputypes as T
    names = ["GpuBlasNode", "GpuBlasTriangle", "GpuBlasDesc", "GpuBlasInstance", "GpuTlasNode", "GpuMeshTransform", "GpuMesh", "GpuMaterial", "GpuVertex",
             "GpuLight", "GpuWavefrontRay", "GpuUnskinnedVertex", "GpuSettings", "idkpt_settings", "idkpt_scene_desc", "idkpt_stats", "idkpt_texture"]
    c = tmp_path / "s.c"
    c.write_text('#include <stdio.h>\n#include "idkpt.h"\nint main(void){' + "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).decode().split()]
    py = [getattr(T, n).itemsize for n in names[:12]] + [C.sizeof(T.GpuSettings), C.sizeof(T.Settings), C.sizeof(T.SceneDesc), C.sizeof(T.Stats), C.sizeof(T.Texture)]
    assert sizes == py


def test_no_gpu_means_loud_failure_not_fallback():
    """Host logic without a GPU: argument validation works, and creating a context reports NO_DEVICE (status 5)."""
    from idkengine_amd import _lib
    L = _lib.load()
    assert L.idkptGetVersionString().startswith(b"idkpt")
    n = C.c_int32(-1)
    L.idkptGetDeviceCount(C.byref(n))
    ctx = C.c_void_p()
    assert L.idkptCreate(0, None, C.byref(ctx)) == 2 and L.idkptCreate(65, None, C.byref(ctx)) == 2     # 1..64 devices per context
    assert L.idkptCreate(1, None, None) == 2
    if n.value <= 0:
        assert L.idkptCreate(2, None, C.byref(ctx)) == 5 and not ctx.value                              # a multi-device context needs devices too
    if n.value <= 0:
        assert L.idkptCreate(1, None, C.byref(ctx)) == 5 and not ctx.value
        from idkengine_amd.pathtracer import PathTracer, IdkPtError
        with pytest.raises(IdkPtError):
            PathTracer(64, 64)
    assert L.idkptRender(None) == 2 and L.idkptDestroy(None) == 2


def test_plain_c_host_compiles_and_links():
    """tests/c_driver/abi_driver.c (the C11 host used by the GPU parity test) builds against include/idkpt.h and links libidkpt.so."""
    import subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from idkengine_amd import build
    build.build_hip()
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "abi_driver")
        subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_driver", "abi_driver.c"),
                               "-L", os.path.join(root, "idkengine_amd"), "-lidkpt", "-Wl,-rpath," + os.path.join(root, "idkengine_amd"), "-o", exe])
        assert os.path.exists(exe)


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path only: without a device it must stop with a clear message, not fall back to the CPU oracle."""
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--tris", "1000"], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs a GPU" in (out.stderr + out.stdout)
    assert not any(line.startswith("{") for line in out.stdout.splitlines())      # no metric line


def test_every_developer_option_is_documented_and_forwarded():
    """idkptSetDeveloperOption: every name the library accepts (csrc/host_options.hpp, csrc/idkpt_api.hpp) is documented in include/idkpt.h with its values and default, and the
    Python host mirror forwards IDKPT_<NAME> for exactly those names (graph_probe exists in developer builds only)."""
    import re
    csrc = os.path.join(ROOT, "idkengine_amd", "csrc")
    text = open(os.path.join(csrc, "host_options.hpp")).read() + open(os.path.join(csrc, "idkpt_api.hpp")).read()
    accepted = set(re.findall(r'n == "([a-z0-9_]+)"', text)) | set(re.findall(r'name\) == "([a-z0-9_]+)"', text))
    header = open(os.path.join(ROOT, "include", "idkpt.h")).read()
    from idkengine_amd.pathtracer import _OPTION_NAMES
    assert len(accepted) >= 40
    assert [n for n in sorted(accepted) if f'"{n}"' not in header] == []
    assert sorted(accepted - set(_OPTION_NAMES)) == ["graph_probe"]
    assert sorted(set(_OPTION_NAMES) - accepted) == []

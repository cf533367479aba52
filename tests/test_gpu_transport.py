"""RCCL as the transport of a multi-device context (csrc/transport_rccl.hpp; include/idkpt.h idkptGetTransportInfo / idkptTransportSelfTest).  A one-GPU box cannot form
an N-rank communicator (RCCL wants one rank per device), so what is checked here is everything short of that: the library is found and every call the transport uses
works on a one-rank communicator with the right bytes; a context whose members share a GPU reports peer copies and why; transport = 2 refuses instead of degrading;
transport = 1 never touches RCCL; frames are the oracle's under every setting."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, HERE)
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402
from idkengine_amd._lib import IdkPtError  # noqa: E402
from gpu_helpers import bits, oracle_render  # noqa: E402

pytestmark = pytest.mark.gpu


def test_rccl_is_loadable_and_its_calls_round_trip_on_one_rank():
    rc, ver, detail = PathTracer.transport_self_test(0)
    assert rc == 0 and detail == "ok", (rc, ver, detail)
    assert ver >= 21000          # RCCL 2.10+: grouped send / recv


def test_one_device_context_has_no_transport():
    pt = PathTracer(32, 32)
    info = pt.transport_info()
    assert info["transport"] == "none" and info["rccl_ranks"] == 0
    pt.Dispose()


def test_members_on_one_gpu_fall_back_to_peer_copies_and_say_why(oracle_mod, native_builder):
    sc = S.cornell_scene(native_builder, "mixed"); w = h = 96; cam = S.cornell_camera(w, h)
    o = oracle_render(oracle_mod, sc, cam, w, h, RayDepth=3)
    for opt in (0, 1):
        pt = PathTracer(w, h, devices=[0, 0]); pt.set_option("transport", opt)
        pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = 3; pt.Compute()
        assert (bits(pt.Result) == bits(o.image(0))).all()
        ptr, nbytes = pt.image_device_ptr(0)               # the gather on device 0
        assert nbytes == w * h * 16
        info = pt.transport_info()
        assert info["transport"] == "peer-copy" and info["rccl_ranks"] == 0
        assert ("share GPU" in info["detail"]) if opt == 0 else ("by option" in info["detail"]), info
        pt.Dispose()
    pt = PathTracer(w, h, devices=[0, 0]); pt.set_option("transport", 2)
    with pytest.raises(IdkPtError, match="RCCL"):
        pt.UploadScene(sc)
    pt.Dispose(); o.close()

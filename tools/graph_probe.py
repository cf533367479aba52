"""Developer tool (GPU box, developer build of the library): would a hipGraph of a batch's launches beat the launches?  For one-sample batches (the
launch-bound end: 13 dependent kernels per RayDepth-2 frame) three ways of rendering the same frame K times:
  (1) ResetAccumulation, Compute, Synchronize per frame (the SURVEY 8d protocol),  (2) the same frames queued back to back, one synchronisation at the end,
  (3) the batch captured into a hipGraph once and replayed K times (option "graph_probe": stderr line of the library).
usage: IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so python tools/graph_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

K = 200
cornell = S.cornell_scene(NativeBuilder(), "mixed")
soup = S.soup_scene(1000000, NativeBuilder(), seed=1)
cases = [("cornell 256x256 depth 2", cornell, S.cornell_camera, 256, 256, 2), ("cornell 256x256 depth 5", cornell, S.cornell_camera, 256, 256, 5),
         ("cornell 1920x1080 depth 5", cornell, S.cornell_camera, 1920, 1080, 5), ("headline 1920x1080 depth 2", soup, lambda w, h: S.Camera(w, h), 1920, 1080, 2)]
for name, sc, camf, w, h, d in cases:
    pt = PathTracer(w, h); pt.UploadScene(sc); pt.SetCamera(camf(w, h)); pt.RayDepth = d; pt.set_max_batch(1)
    for _ in range(20):
        pt.ResetAccumulation(); pt.Compute(); pt.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        pt.ResetAccumulation(); pt.Compute(); pt.synchronize()
    t_sync = (time.perf_counter() - t0) / K
    pt.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        pt.ResetAccumulation(); pt.Compute()
    pt.flush(); pt.synchronize()
    t_async = (time.perf_counter() - t0) / K
    print(f"{name}: synchronised per frame {t_sync * 1e6:7.1f} us, queued back to back {t_async * 1e6:7.1f} us per frame", flush=True)
    pt.ResetAccumulation(); pt.set_option("graph_probe", K); pt.Compute(); pt.synchronize()      # prints the graph line on stderr
    pt.Dispose()

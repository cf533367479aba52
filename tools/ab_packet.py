"""Round 6 experiment: the primary launch as a packet launch (option packet: 0 never / 2 always / 1 by the kernel's own counters; csrc/kernels_packet.hpp) against k_trace2.
Headline / interior / atrium views at 1920x1080, 32 and 20 samples in flight, A B A B on one box; RayDepth 1 rows time the primary launch alone.  Bit-identical frames are asserted.
usage: python tools/ab_packet.py [view ...]     (AB_MODES=0,2,0,2,1 by default)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402

W, H = bench.W, bench.H
modes = [int(x) for x in os.environ.get("AB_MODES", "0,2,0,2,1").split(",")]
want = set(sys.argv[1:])
pt = PathTracer(W, H); pt.enable_timing(True)
if os.environ.get("PACKET_WAVES"): pt.set_option("packet_waves", int(os.environ["PACKET_WAVES"]))
soup = S.soup_scene(bench.N_TRIS, NativeBuilder(), seed=1)
atrium = S.atrium_scene(bench.N_TRIS, NativeBuilder())
rows = (("interior_primary", soup, bench.view_camera(S, "interior", W, H), 1, 32), ("interior", soup, bench.view_camera(S, "interior", W, H), 2, 32), ("interior_20_samples", soup, bench.view_camera(S, "interior", W, H), 2, 20),
        ("interior_d5", soup, bench.view_camera(S, "interior", W, H), 5, 32), ("atrium_primary", atrium, S.atrium_camera(W, H), 1, 32), ("atrium", atrium, S.atrium_camera(W, H), 2, 32),
        ("headline_primary", soup, bench.view_camera(S, "headline", W, H), 1, 32), ("headline", soup, bench.view_camera(S, "headline", W, H), 2, 32), ("headline_20_samples", soup, bench.view_camera(S, "headline", W, H), 2, 20),
        ("interior_one_frame", soup, bench.view_camera(S, "interior", W, H), 2, 1), ("atrium_one_frame", atrium, S.atrium_camera(W, H), 2, 1))
for name, sc, cam, depth, B in rows:
    if want and name not in want:
        continue
    pt.UploadScene(sc); pt.SetCamera(cam); pt.RayDepth = depth
    row = {}; ref = None
    for opt in modes:
        pt.set_option("packet", opt)
        rays, dt = bench.timed_batch(pt, B, max(B, 16), reps=5)
        st = pt.stats()
        img = np.ascontiguousarray(pt.Result).view(np.uint32)
        if ref is None: ref = img.copy()
        assert (img == ref).all(), "frames differ"
        e = {"mray_s": round(rays / dt / 1e6, 1), "trace_ms_per_launch": round(st["trace_ms_total"] / max(st["trace_launches"], 1), 4)}
        if st["packet_packets"]:
            e.update(live=round(st["packet_live_lanes"] / (64.0 * max(1, st["packet_node_steps"])), 3), steps_per_packet=round(st["packet_node_steps"] / st["packet_packets"], 1),
                     tri_rounds_per_packet=round(st["packet_triangle_rounds"] / st["packet_packets"], 1), rays_per_packet=round(st["packet_rays_entered"] / st["packet_packets"], 1),
                     flagged=int(st["packet_flagged_rays"]), packets=int(st["packet_packets"]))
        row.setdefault(str(opt), []).append(e)
    print(json.dumps({name: row}), flush=True)
pt.Dispose()

"""Developer tool (GPU box): where k_trace2's wave cycles go — refill / node phase / leaf phase / other — per view, from the s_memtime-instrumented
instantiation of the developer build (libidkpt_dev.so, option trace_variant = 113).  Run as
    IDKPT_LIB_PATH=$PWD/idkengine_amd/libidkpt_dev.so python tools/phase_profile.py
The library prints the raw buckets to stderr ("[idkpt prof] ..."); this script adds the shares."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, ".."))
sys.path.insert(0, %r)
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from sweep_trace import run, W, H
view = sys.argv[1]
sc = S.atrium_scene(1000000, NativeBuilder()) if view == "atrium" else S.soup_scene(1000000, NativeBuilder(), seed=1)
cam = S.atrium_camera(W, H) if view == "atrium" else (S.Camera(W, H) if view == "headline" else S.Camera(W, H, position=(0.0, 0.0, 0.0)))
r, img, rays = run(sc, cam, int(os.environ.get("PHASE_VARIANT", "113")), int(sys.argv[2]), 64)
print("RESULT", view, sys.argv[2], r["mray_s"], r["trace_ms_per_frame"])
''' % (HERE, HERE)

if __name__ == "__main__":
    if "libidkpt_dev" not in os.environ.get("IDKPT_LIB_PATH", ""):
        sys.exit("set IDKPT_LIB_PATH to idkengine_amd/libidkpt_dev.so (python -c 'from idkengine_amd import build as B; B.build_hip(developer=True)')")
    for view in ("headline", "interior", "atrium"):
        for batch in (32, 1):
            p = subprocess.run([sys.executable, "-c", CHILD, view, str(batch)], capture_output=True, text=True, timeout=600)   # (PHASE_VARIANT=213 in the environment: the wide-node walk's instrumented instantiation)
            prof = [l for l in p.stderr.splitlines() if l.startswith("[idkpt prof]")]
            res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
            if not prof:
                print(view, batch, "no profile line", p.stderr[-400:]); continue
            v = [int(x) for x in re.findall(r"\d+", prof[-1])]
            refill, node, leaf, other, refills, refLanes, steps, stepLanes, leafPhases, leafLanes = v[:10]
            leafTests, leafTrips = (v[10], v[11]) if len(v) >= 12 else (0, 0)
            tot = float(refill + node + leaf + other)
            print(f"{view:9s} batch {batch:2d}: wave cycles refill {100 * refill / tot:5.1f} %  node phase {100 * node / tot:5.1f} %  leaf phase {100 * leaf / tot:5.1f} %  other {100 * other / tot:5.1f} % | "
                  f"{node / max(1, steps):7.0f} cycles per node step ({stepLanes / max(1, steps):4.1f} lanes), {leaf / max(1, leafPhases):7.0f} per leaf phase ({leafLanes / max(1, leafPhases):4.1f} lanes), "
                  f"{refill / max(1, refills):7.0f} per refill ({refLanes / max(1, refills):4.1f} lanes) | per leaf phase {leafTests / max(1, leafPhases):5.1f} (ray, triangle) tests in {leafTrips / max(1, leafPhases):4.2f} loop trips | {res[-1] if res else ''}", flush=True)

"""Developer tool (GPU box): A/B of whole library builds on one box — every library given on the command line renders the three bench views (soup headline, soup
interior, atrium; 1920x1080, RayDepth 2) with 32 samples in flight and one frame at a time, each in its own process (IDKPT_LIB_PATH), the libraries taken in turns
for `rounds` rounds so that clock / thermal drift hits all of them alike; images and ray state of every library are compared with the first one's.
usage: python tools/ab_libs.py [--rounds 2] [--env IDKPT_NAME=value ...] lib1.so lib2.so ...   -> gpurun_out/ab_libs.json"""
import json
import os
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, os, json, hashlib
sys.path.insert(0, os.path.join(%r, "..")); sys.path.insert(0, %r)
from idkengine_amd import scenes as S
from idkengine_amd.bvh import NativeBuilder
from sweep_trace import run, W, H
soup = S.soup_scene(1000000, NativeBuilder(), seed=1); atrium = S.atrium_scene(1000000, NativeBuilder())
views = {"headline": (soup, S.Camera(W, H)), "interior": (soup, S.Camera(W, H, position=(0.0, 0.0, 0.0))), "atrium": (atrium, S.atrium_camera(W, H))}
out = {}
for v, (sc, cam) in views.items():
    for batch, frames in ((32, 96), (1, 40)):
        r, img, rays = run(sc, cam, 0, batch, frames)
        out[f"{v}_b{batch}"] = {"mray_s": r["mray_s"], "trace_ms_per_frame": r["trace_ms_per_frame"], "sha": hashlib.sha256(img.tobytes() + rays.tobytes()).hexdigest()[:16]}
print("ABRESULT " + json.dumps(out))
''' % (HERE, HERE)

if __name__ == "__main__":
    args = sys.argv[1:]
    rounds = 2; env_extra = {}
    libs = []
    i = 0
    while i < len(args):
        if args[i] == "--rounds": rounds = int(args[i + 1]); i += 2
        elif args[i] == "--env": k, v = args[i + 1].split("=", 1); env_extra[k] = v; i += 2
        else: libs.append(args[i]); i += 1
    res = {l: [] for l in libs}
    for _ in range(rounds):
        for l in libs:
            p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=900, env=dict(os.environ, IDKPT_LIB_PATH=os.path.abspath(l), **env_extra))
            line = [x for x in p.stdout.splitlines() if x.startswith("ABRESULT ")]
            if not line:
                print(l, "FAILED", p.stderr[-600:]); continue
            res[l].append(json.loads(line[-1][9:]))
    base = libs[0]
    report = {}
    for key in res[base][0]:
        row = {}
        for l in libs:
            vals = [r[key]["mray_s"] for r in res[l]]
            row[os.path.basename(l)] = {"mray_s": round(statistics.median(vals), 1), "runs": [round(v, 1) for v in vals], "same_bits_as_first": all(r[key]["sha"] == res[base][0][key]["sha"] for r in res[l])}
        report[key] = row
        b = row[os.path.basename(base)]["mray_s"]
        print(f"{key:14s} " + "  ".join(f"{n}: {d['mray_s']:8.1f} (x{d['mray_s'] / b:.3f}{'' if d['same_bits_as_first'] else ' BITS DIFFER'})" for n, d in row.items()), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(report, open("gpurun_out/ab_libs.json", "w"), indent=1)

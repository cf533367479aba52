"""Developer tool (MI355X): randomised differential test of the HIP path against the CPU oracle.

Every case draws a scene (several BLASes of random triangle soups with random materials — diffuse, metallic, emissive, transmissive thin / volumetric,
alpha blend / cut-off — optional textures, optional rotated / translated / scaled instances, optional lights, constant or per-face sky), a camera
(inside or outside the geometry, random field of view, optional lens), a frame size (odd sizes included) and settings (RayDepth 1..7, ray sorting,
Russian roulette, lights, AOVs, TLAS, samples per pixel), renders it through libidkpt.so with a random number of samples in flight and through the
oracle, and compares image, AOV images, primary hits, ray state, alive queue and ray counts bit for bit.

    python tools/fuzz_parity.py [cases=60] [first_seed=0]
Prints one line per case and the number of mismatching cases; exit status 1 if any."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests")); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden"))
import torch  # noqa: E402,F401  (one HIP runtime per process)
import configs  # noqa: E402
from idkengine_amd import scenes as S  # noqa: E402
from idkengine_amd import gputypes as T  # noqa: E402
from idkengine_amd.bvh import NativeBuilder  # noqa: E402
from idkengine_amd.pathtracer import PathTracer  # noqa: E402
from oracle import oracle as O  # noqa: E402


def random_material(rng):
    kind = rng.integers(0, 7)
    col = tuple(rng.uniform(0.05, 1.0, 3)) + (1.0,)
    if kind == 0: return S.make_material(col, roughness=1.0)
    if kind == 1: return S.make_material(col, metallic=float(rng.uniform(0.2, 1.0)), roughness=float(rng.uniform(0.0, 0.8)))
    if kind == 2: return S.make_material(col, emissive=tuple(rng.uniform(0.0, 4.0, 3)))
    if kind == 3: return S.make_material(col, transmission=float(rng.uniform(0.4, 1.0)), roughness=float(rng.uniform(0.0, 0.4)), ior=float(rng.uniform(1.0, 1.8)),
                                         absorbance=tuple(rng.uniform(0.0, 1.5, 3)), volumetric=True)
    if kind == 4: return S.make_material(col, transmission=float(rng.uniform(0.4, 1.0)), roughness=float(rng.uniform(0.0, 0.4)), ior=float(rng.uniform(1.0, 1.8)), volumetric=False)
    if kind == 5: return S.make_material(col[:3] + (float(rng.uniform(0.1, 0.9)),), alpha_cutoff=2.0)                      # stochastic blend
    return S.make_material(col[:3] + (float(rng.uniform(0.1, 0.9)),), alpha_cutoff=float(rng.uniform(0.2, 0.8)))          # cut-off


def random_scene(rng, builder):
    # now and then (FUZZ_P_BIG, default 1.5 % of the cases; 10+ s each) the first mesh has more than 2^21 triangles: triangle ids beyond the 21 bits
    # the reference's sort key keeps (NHit/compute.glsl:81) and beyond what most index arithmetic of small scenes ever sees
    big = rng.random() < float(os.environ.get("FUZZ_P_BIG", "0.015"))
    nb = int(rng.integers(1, 5))
    if os.environ.get("FUZZ_BLASES"):                     # e.g. FUZZ_BLASES=2,14: a soak of the several-instance paths (other scenes than the default draw's: the stream shifts)
        lo_, hi_ = (int(v) for v in os.environ["FUZZ_BLASES"].split(",")); nb = int(rng.integers(lo_, hi_ + 1))
    extent = float(rng.choice([1.5, 4.0, 10.0]))
    use_tex = rng.random() < 0.3
    blases = []
    for k in range(nb):
        meshes = []
        for _ in range(int(rng.integers(1, 4))):
            n = int(rng.choice([1, 2, 7, 60, 400, 3000]))
            edge = float(rng.choice([0.15, 0.6, 2.0]))
            if big and k == 0 and not meshes:
                n, edge = 2_150_000, 0.02 * extent
            p, i, nrm, tan = S.flat_shaded(S.soup_triangles(n, int(rng.integers(1, 1 << 30)), extent, edge))
            mat = random_material(rng)
            uvs = None; kw = None
            if use_tex:
                uvs = (rng.uniform(-1.0, 2.0, (len(p), 2))).astype(np.float32)
                for slot in ("BaseColorTexture", "EmissiveTexture", "MetallicRoughnessTexture", "NormalTexture", "TransmissionTexture"):
                    if rng.random() < 0.4: mat[slot] = int(rng.integers(1, 4))
                if rng.random() < 0.5: kw = dict(NormalMapStrength=float(rng.uniform(0.0, 1.0)))
            if rng.random() < 0.25:
                kw = dict(kw or {}); kw.update(RoughnessBias=float(rng.uniform(-0.3, 0.3)), SpecularBias=float(rng.uniform(-0.3, 0.3)), EmissiveBias=float(rng.uniform(0.0, 0.2)),
                                               TransmissionBias=float(rng.uniform(-0.3, 0.3)), IORBias=float(rng.uniform(-0.2, 0.3)), TintOnTransmissive=int(rng.integers(0, 2)))
            meshes.append(S.MeshInput(p, i, mat, nrm, tan, uvs=uvs, mesh_kwargs=kw))
        tr = None
        if k > 0 and rng.random() < 0.8:
            sc = np.diag([float(rng.uniform(0.5, 1.6))] * 3 + [1.0])
            tr = S.rotation_y(float(rng.uniform(0, 360))) @ S.translation(tuple(rng.uniform(-0.3 * extent, 0.3 * extent, 3))) @ sc
        if os.environ.get("FUZZ_SAME_SPACE") and k > 0:     # a soak of the unified tree (k_trace_inst UNI): every BLAS under BLAS 1's transform (rotation, translation, scale — or none)
            tr = blases[1]["transform"] if k > 1 else tr
        blases.append({"meshes": meshes, "transform": tr, "refittable": bool(rng.random() < 0.3)})
    if os.environ.get("FUZZ_SAME_SPACE") and nb > 1:
        blases[0]["transform"] = blases[1]["transform"]
    lights = None
    if rng.random() < 0.4:
        lights = S.make_lights([(tuple(rng.uniform(-0.6 * extent, 0.6 * extent, 3)), float(rng.uniform(0.05, 0.2) * extent), tuple(rng.uniform(0.5, 8.0, 3))) for _ in range(int(rng.integers(1, 4)))])
    sky = tuple(rng.uniform(0.0, 1.5, 3)) if rng.random() < 0.7 else None
    sc = S.assemble(blases, builder, lights=lights, sky_color=sky if sky is not None else (0.5, 0.5, 0.5))
    if sky is None:
        s = int(rng.choice([1, 2, 5]))
        faces = np.zeros((6, s, s, 4), np.float32); faces[..., :3] = rng.uniform(0, 2, (6, s, s, 3)); faces[..., 3] = 1.0
        sc.sky_faces = faces
    if use_tex:
        sc.textures = [rng.uniform(0.0, 1.0, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 4)).astype(np.float32) for _ in range(3)]
        if os.environ.get("FUZZ_TEX_CONTRAST"):      # (oracle/glref/fuzz_reference.py: the same cases with the textures' contrast scaled about 0.5 — how much of a residue is texture gradient x input difference)
            sc.textures = [(np.float32(0.5) + (t - np.float32(0.5)) * np.float32(float(os.environ["FUZZ_TEX_CONTRAST"]))).astype(np.float32) for t in sc.textures]
    return sc, extent, lights is not None, nb


def draw_case(seed, builder):
    """Everything one_case draws, in the same order (also used by oracle/glref/fuzz_reference.py, which runs the same cases through the reference's shaders)."""
    rng = np.random.default_rng(seed)
    sc, extent, has_lights, nb = random_scene(rng, builder)
    w, h = int(rng.choice([17, 40, 64, 96, 131])), int(rng.choice([9, 33, 48, 77]))
    inside = rng.random() < 0.5
    pos = tuple(rng.uniform(-0.4 * extent, 0.4 * extent, 3)) if inside else tuple(rng.normal(size=3) / 1.0 * 0.2 * extent + np.float64([0, 0, 2.2 * extent]))
    cam = S.Camera(w, h, position=pos, view_dir=tuple(-np.float64(pos) + rng.normal(size=3) * 0.3 * extent) if not inside else tuple(rng.normal(size=3)), fovy_deg=float(rng.uniform(35, 115)))
    ov = dict(RayDepth=int(rng.integers(1, 8)), DoRaySorting=int(rng.random() < 0.4), DoRussianRoulette=int(rng.random() < 0.8), OutputAOVs=int(rng.random() < 0.4),
              DoTraceLights=int(has_lights and rng.random() < 0.8), UseTlas=int(nb > 1 and rng.random() < 0.5), SamplesPerPixel=int(rng.choice([1, 1, 2, 3])))
    if rng.random() < 0.3: ov.update(FocalLength=float(rng.uniform(0.5, 2.0) * extent), LenseRadius=float(rng.uniform(0.005, 0.05) * extent))
    if os.environ.get("FUZZ_BLASES") and rng.random() < 0.8: ov["UseTlas"] = 0
    frames = int(rng.integers(1, 6)); batch = int(rng.choice([1, 2, 5, 8]))
    st = configs.apply_settings(T.Settings.default(), ov)
    _legacy = (int(rng.choice([0, 0, 1, 2])), int(rng.integers(1, 6)), int(rng.choice([0, 0, 1, 2])))   # (draws of options that left the product in round 5: the cases of earlier rounds keep their scenes)
    opts = {"grid_rays_x4": int(rng.choice([6, 6, 0, 1, 64])), "grid_mid_waves": int(rng.choice([20, 20, 0, 3])), "defer_last": int(rng.choice([1, 1, 0])), "grid_hint": int(rng.choice([2, 2, 0, 1])), "trace_waves": int(rng.choice([0, 0, 1, 7])),
            "split": int(rng.choice([1, 2, 2, 3, 0])), "split_donor": int(rng.choice([0, 1]))}
    opts["fused"] = int(rng.choice([0, 2, 2, 1])); opts["fused_shade_min"] = int(rng.choice([16, 1, 8, 32, 64]))
    _legacy2 = (int(rng.choice([0, 0, 2])), int(rng.choice([0, 0, 7, 2]))); opts["query_scheduler"] = 1
    opts["wide"] = int(rng.choice([0, 1, 1])); opts["wide_cap"] = int(rng.choice([0, 0, 6]))     # the wide-node walk (kernels_wide.hpp) on the one-BLAS cases
    opts["inst_tlas_overlap"] = 100; opts["inst_tlas"] = int(rng.choice([2, 2, 0, 8])); opts["inst_sieve"] = int(rng.choice([0, 2])); opts["inst_sieve_overlap"] = 100; opts["gen_pixel_major"] = int(rng.choice([8, 2, 0])); opts["bounce_pixel_major"] = int(rng.choice([2, 1, 0]))                                            # the instance loop through the library's own TLAS (kernels_trace_inst.hpp) on the several-BLAS cases without UseTlas
    opts["packet"] = int(rng.choice([2, 2, 1, 0])); opts["packet_waves"] = int(rng.choice([0, 0, 1, 5])); opts["packet_min_live"] = int(rng.choice([60, 0, 100]))      # the packet walk of the primary launch (kernels_packet.hpp) on the one-BLAS cases
    opts["inst_unify"] = int(rng.choice([4096, 4096, 3, 40, 0])); opts["inst_braid"] = int(rng.choice([0, 0, 16])); opts["inst_general"] = int(rng.choice([2, 2, 0]))      # the unified tree of same-space scenes, entries under the own TLAS (drawn last: the cases of earlier rounds keep their scenes)
    # free choices of the implementation: never visible in the output
    return sc, cam, w, h, ov, st, opts, frames, batch, nb


def one_case(seed, builder):
    sc, cam, w, h, ov, st, opts, frames, batch, nb = draw_case(seed, builder)
    if os.environ.get("FUZZ_DEPTH"):                                   # (e.g. FUZZ_DEPTH=2: every case can take the fused FirstHit + NHit launch)
        ov["RayDepth"] = int(os.environ["FUZZ_DEPTH"]); st = configs.apply_settings(T.Settings.default(), ov)
    capture = not (opts.get("fused", 0) and ov["RayDepth"] == 2)       # (a host that captures primary hits keeps the two-launch schedule: those cases check everything else)
    pt = PathTracer(w, h, settings=st)
    for k_, v_ in opts.items():
        pt.set_option(k_, v_)
    pt.UploadScene(sc); pt.SetCamera(cam); pt.enable_primary_hit_capture(capture); pt.set_max_batch(batch)
    o = O.OraclePathTracer(sc, w, h); o.set_camera(cam); configs.apply_settings(o.settings, ov)
    for _ in range(frames):
        pt.Compute(); o.render()
    bits = lambda a: np.ascontiguousarray(a).view(np.uint32)  # noqa: E731
    bad = []
    if not (bits(pt.Result) == bits(o.image(0))).all(): bad.append("image")
    if ov["OutputAOVs"] and not ((bits(pt.AlbedoTexture) == bits(o.image(1))).all() and (bits(pt.NormalTexture) == bits(o.image(2))).all()): bad.append("aov")
    if capture:
        gt, gtri, gb = pt.primary_hits(); ot, otri, ob = o.primary_hits()
        if not ((gtri == otri).all() and (bits(gt) == bits(ot)).all() and (bits(gb) == bits(ob)).all()): bad.append("primary hits")
    if pt.rays().tobytes() != o.rays().tobytes(): bad.append("ray state")
    if not (pt.alive_queue().shape == o.alive_queue().shape and (pt.alive_queue() == o.alive_queue()).all()): bad.append("queue")
    if pt.stats()["rays_traced"] != o.stats()["rays_traced"]: bad.append("ray count")
    rays = pt.stats()["rays_traced"]; uni = pt.stats()["inst_unified_launches"]
    pt.Dispose()
    print(f"seed {seed}: {w}x{h} blases {nb} tris {len(sc.blas_triangles)} {ov} {opts} frames {frames} batch {batch} rays {rays} unified launches {uni}: {'OK' if not bad else 'MISMATCH ' + ', '.join(bad)}", flush=True)
    return not bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    builder = NativeBuilder()
    bad = [s for s in range(first, first + cases) if not one_case(s, builder)]
    print(f"cases {cases}, mismatching seeds: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

"""Cases the oracle (and through it the HIP path) is PINNED on against the reference's own GLSL shaders executed
by Mesa llvmpipe (oracle/glref/).  Shared by the generator (oracle/glref/make_vectors.py, runs only where
/root/reference exists), the CPU tests (oracle vs fixtures) and the GPU tests (HIP path vs fixtures).

name: (scene factory(builder), camera factory(w, h), w, h, settings overrides)
"""
import numpy as np
from idkengine_amd import scenes as S
import configs


def _lights_scene(b):
    from idkengine_amd import gputypes as T
    sc = S.cornell_scene(b, "mixed")
    lights = np.zeros(2, T.GpuLight)
    lights[0]["Position"] = (0.3, 0.2, 0.4); lights[0]["Radius"] = 0.18; lights[0]["Color"] = (6.0, 5.0, 3.0); lights[0]["PointShadowIndex"] = -1
    lights[1]["Position"] = (-0.5, -0.4, 0.1); lights[1]["Radius"] = 0.1; lights[1]["Color"] = (1.0, 2.0, 8.0); lights[1]["PointShadowIndex"] = -1
    sc.lights = lights
    return sc


def _alpha_scene(b):
    m = S.cornell_meshes("mixed")
    m["short"].material = S.make_material((0.9, 0.3, 0.3, 0.4), alpha_cutoff=2.0)     # stochastic blend
    m["tall"].material = S.make_material((0.3, 0.9, 0.3, 0.3), alpha_cutoff=0.5)      # cut-off: passes through
    return S.assemble([{"meshes": m["walls"] + [m["short"], m["tall"]]}], b, sky_color=(0.2, 0.2, 0.2))


def _textured_scene(b):
    rng = np.random.default_rng(5)
    m = S.cornell_meshes("diffuse")
    for part, scale in (("tall", 1.0), ("short", 3.0)):                                  # uv > 1 on "short": repeat wrap
        m[part].uvs = (rng.uniform(0, 1, (len(m[part].positions), 2)) * scale).astype(np.float32)
    m["tall"].material["BaseColorTexture"] = 1; m["tall"].material["EmissiveTexture"] = 2; m["tall"].material["EmissiveFactor"] = (0.5, 0.5, 0.5)
    m["short"].material["BaseColorTexture"] = 3; m["short"].material["MetallicRoughnessTexture"] = 1
    m["short"].material["MetallicFactor"] = 0.8; m["short"].material["RoughnessFactor"] = 0.6
    sc = S.assemble([{"meshes": m["walls"] + [m["short"], m["tall"]]}], b, sky_color=(0.3, 0.4, 0.6))
    sc.textures = [rng.uniform(0.2, 1.0, (8, 8, 4)).astype(np.float32), np.float32([[[0.2, 0.7, 0.1, 1.0]]]),
                   rng.uniform(0.0, 1.0, (5, 3, 4)).astype(np.float32)]
    return sc


def _lights_textured_scene(b):
    sc = _textured_scene(b)
    from idkengine_amd import gputypes as T
    lights = np.zeros(2, T.GpuLight)
    lights[0]["Position"] = (0.3, 0.2, 0.4); lights[0]["Radius"] = 0.18; lights[0]["Color"] = (6.0, 5.0, 3.0); lights[0]["PointShadowIndex"] = -1
    lights[1]["Position"] = (-0.5, -0.4, 0.1); lights[1]["Radius"] = 0.1; lights[1]["Color"] = (1.0, 2.0, 8.0); lights[1]["PointShadowIndex"] = -1
    sc.lights = lights
    return sc


def _bias_normalmap_scene(b):
    """The Surface.glsl / Shading.glsl branches the other cases leave out: thin-walled (non-volumetric) transmission without tint,
    every GpuMesh bias (SurfaceApplyModificatons), a tangent-space normal map blended by NormalMapStrength, a transmission texture."""
    rng = np.random.default_rng(21)
    m = S.cornell_meshes("mixed")
    m["short"].material = S.make_material((0.85, 0.9, 0.6, 1.0), transmission=0.9, roughness=0.15, ior=1.3, absorbance=(0.2, 0.4, 0.1), volumetric=False)
    m["short"].mesh_kwargs = dict(TintOnTransmissive=0, IORBias=0.2, RoughnessBias=0.1, TransmissionBias=-0.15, AbsorbanceBias=(0.05, -0.5, 0.1), EmissiveBias=0.02)
    m["short"].uvs = rng.uniform(0, 1, (len(m["short"].positions), 2)).astype(np.float32)
    m["short"].material["TransmissionTexture"] = 2
    m["tall"].material = S.make_material((0.8, 0.7, 0.7, 1.0), metallic=0.3, roughness=0.5)
    m["tall"].mesh_kwargs = dict(NormalMapStrength=0.7, SpecularBias=0.25, RoughnessBias=-0.2)
    m["tall"].uvs = (rng.uniform(0, 1, (len(m["tall"].positions), 2)) * 2.0).astype(np.float32)
    m["tall"].material["NormalTexture"] = 1
    sc = S.assemble([{"meshes": m["walls"] + [m["short"], m["tall"]]}], b, sky_color=(0.25, 0.3, 0.4))
    nm = np.zeros((6, 6, 4), np.float32); nm[..., 0:2] = rng.uniform(0.3, 0.7, (6, 6, 2)); nm[..., 2] = 1.0; nm[..., 3] = 1.0
    tr = np.ones((4, 4, 4), np.float32); tr[..., 0] = rng.uniform(0.2, 1.0, (4, 4))
    sc.textures = [nm, tr]
    return sc


def _sampler_scene(b, eight_bit=False):
    """Per-texture sampler state (ModelLoader.GetGLSamplerState, Utils/ModelLoader.cs:1166-1197; taps Surface.glsl:49-77): a wall of 3 x 6 quads, one per (wrapS, wrapT,
    magFilter) of {REPEAT, CLAMP_TO_EDGE, MIRRORED_REPEAT}^2 x {LINEAR, NEAREST}, each with its own non-square image as base colour and emission, uv running from -1.3 to 2.4
    across the quad (negative and > 1 on both axes: every branch of table 8.20's wrap functions), above a glossy floor that carries the bounces.  eight_bit: the same wall with
    RGBA8 images, all of them under GL_NEAREST — the one 8-bit configuration llvmpipe evaluates at full precision (it filters RGBA8 in 8-bit fixed point, up to 1.1 / 255 off the
    float filter, and decodes sRGB with a polynomial 2.5 % off the specification's function: tests/test_glref.py bounds both live, tests/test_oracle_kats.py pins the decode)."""
    from idkengine_amd import gputypes as T
    rng = np.random.default_rng(33)
    meshes, textures = [], []
    combos = [(ws, wt, mf) for mf in (0, 1) for ws in (0, 1, 2) for wt in (0, 1, 2)]
    for k, (ws, wt, mf) in enumerate(combos):
        cx, cy = (k % 6) - 2.5, (k // 6) - 1.0
        quad = S._quad((cx * 0.62 - 0.29, cy * 0.62 - 0.29, -0.6), (cx * 0.62 + 0.29, cy * 0.62 - 0.29, -0.6), (cx * 0.62 + 0.29, cy * 0.62 + 0.29, -0.6), (cx * 0.62 - 0.29, cy * 0.62 + 0.29, -0.6))
        p, i, n, t = S.flat_shaded(quad)
        uv = np.where(np.abs(p[:, :2] - np.float32([cx * 0.62, cy * 0.62])) > 0, (np.sign(p[:, :2] - np.float32([cx * 0.62, cy * 0.62])) * 0.5 + 0.5), 0.0)      # corner -> (0 / 1, 0 / 1)
        uv = (np.float32([-1.3, -0.7]) + uv.astype(np.float32) * np.float32([3.7, 3.1])).astype(np.float32)
        mat = S.make_material((1.0, 1.0, 1.0, 1.0), emissive=(0.6, 0.6, 0.6), roughness=0.9)
        mat["BaseColorTexture"] = len(textures) + 1; mat["EmissiveTexture"] = len(textures) + 1
        meshes.append(S.MeshInput(p, i, mat, n, t, uvs=uv))
        w_, h_ = (5, 3) if k % 2 else (4, 7)
        if eight_bit:
            img = rng.integers(0, 256, (h_, w_, 4), dtype=np.uint8)
            textures.append(T.TextureImage(img, ws, wt, T.IDKPT_FILTER_NEAREST))
        else:
            textures.append(T.TextureImage(rng.uniform(0.05, 1.0, (h_, w_, 4)).astype(np.float32), ws, wt, mf))
    floor = S._quad((-2.2, -1.0, -0.6), (-2.2, -1.0, 2.0), (2.2, -1.0, 2.0), (2.2, -1.0, -0.6))
    p, i, n, t = S.flat_shaded(floor)
    meshes.append(S.MeshInput(p, i, S.make_material((0.7, 0.7, 0.7, 1.0), metallic=0.6, roughness=0.3), n, t))
    sc = S.assemble([{"meshes": meshes}], b, sky_color=(0.1, 0.12, 0.15))
    sc.textures = textures
    return sc


def _sky_scene(b):
    """Every face a different 5x5 image: the seamless GL_LINEAR cube-map filter across face edges and corners."""
    rng = np.random.default_rng(9)
    sc = S.soup_scene(3000, b, seed=5, sky_color=None)
    sky = np.zeros((6, 5, 5, 4), np.float32); sky[..., :3] = rng.uniform(0, 2, (6, 5, 5, 3)); sky[..., 3] = 1.0
    sc.sky_faces = sky
    return sc


GLREF_CASES = dict(configs.CASES)
GLREF_CASES.update({
    "cornell_lights_d5": (_lights_scene, S.cornell_camera, 64, 64, dict(RayDepth=5, DoTraceLights=1)),
    "cornell_lights_sort_d4": (_lights_scene, S.cornell_camera, 64, 64, dict(RayDepth=4, DoTraceLights=1, DoRaySorting=1)),
    "cornell_alpha_d6": (_alpha_scene, S.cornell_camera, 64, 64, dict(RayDepth=6)),
    "cornell_textured_aov_d5": (_textured_scene, S.cornell_camera, 64, 64, dict(RayDepth=5, OutputAOVs=1)),
    "sampler_states_d3": (_sampler_scene, lambda w, h: S.Camera(w, h, position=(0.0, 0.1, 3.0), fovy_deg=48.0), 96, 56, dict(RayDepth=3, OutputAOVs=1)),
    "sampler_states_rgba8_d3": (lambda b: _sampler_scene(b, True), lambda w, h: S.Camera(w, h, position=(0.0, 0.1, 3.0), fovy_deg=48.0), 96, 56, dict(RayDepth=3, OutputAOVs=1)),
    "cornell_bias_normalmap_d6": (_bias_normalmap_scene, S.cornell_camera, 64, 64, dict(RayDepth=6)),
    "cornell_odd_size_d3": (lambda b: S.cornell_scene(b, "mixed"), S.cornell_camera, 53, 37, dict(RayDepth=3)),
    "soup_sky_linear_aov_d3": (_sky_scene, lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(1.0, 0.8, 0.9), fovy_deg=110.0), 96, 72, dict(RayDepth=3, OutputAOVs=1)),
    "soup_multi_tlas_d3": (lambda b: S.soup_scene_multi(6000, b, parts=3, seed=4), lambda w, h: S.Camera(w, h), 96, 54, dict(RayDepth=3, UseTlas=1)),
})


# ---- BASELINE-size cases (tests/golden/glref_full/, minted by oracle/glref/make_full_vectors.py): the reference's own shaders on llvmpipe on WHOLE frames of
# the workloads bench.py times.  A whole frame's ray records are 100 MB, so the fixture keeps (a) the sha256 of the state that was compared with the reference
# ray by ray at generation (the oracle's — and, bit for bit, the HIP path's), (b) the reference's records on a fixed sample of the rays for a direct comparison
# wherever the fixture travels, (c) every ray on which the reference's run and the oracle differ, with both results (glref_check.FULL_ALLOW names them).
# name: (scene key, camera factory(w, h), w, h, settings overrides); scenes by key, built once per test session.
FULL_SCENES = {
    "soup1m": lambda b: S.soup_scene(1000000, b, seed=1),
    "atrium262k": lambda b: S.atrium_scene(262000, b),
    "atrium1m": lambda b: S.atrium_scene(1000000, b),
    "soup4m": lambda b: S.soup_scene(4000000, b, seed=3),
    "soup2m3": lambda b: S.soup_scene(2300000, b, seed=2),
    "lucy": configs.lucy_scene,
    "helmet": configs.helmet_scene,
    "multi900k": lambda b: S.soup_scene_multi(900000, b, parts=3, seed=4),
    "cornell_lights_tex": lambda b: _lights_textured_scene(b),
}
FULL_CASES = {
    "full_headline_d2": ("soup1m", lambda w, h: S.Camera(w, h), 1920, 1080, dict(RayDepth=2)),                                            # BASELINE configs[2], the bench line
    "full_headline_sort_d5": ("soup1m", lambda w, h: S.Camera(w, h), 1920, 1080, dict(RayDepth=5, DoRaySorting=1)),                       # configs[3] (sort on), 4 bounces
    "full_interior_d3": ("soup1m", lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 0.0)), 1920, 1080, dict(RayDepth=3)),                 # every pixel traverses
    "full_atrium262k_d5": ("atrium262k", S.atrium_camera, 1920, 1080, dict(RayDepth=5)),                                                   # configs[1] stand-in
    "full_atrium1m_d2": ("atrium1m", S.atrium_camera, 1920, 1080, dict(RayDepth=2)),                                                       # the 1M-triangle atrium bench.py times beside the headline
    # the reference's own traversal-cost counter (BVHIntersect.glsl:45,60: +1 per node pair, +1.1 per triangle test) on the headline frame's primary rays:
    # pins the P and T that the roofline's algorithmic bytes are computed from against the reference's count, pixel by pixel and in total
    "full_headline_debugcost_d1": ("soup1m", lambda w, h: S.Camera(w, h), 1920, 1080, dict(RayDepth=1, DoDebugBVHTraversal=1)),
    # the two meshes the reference ships (shared vertices, slivers), at full HD
    "full_lucy_d5": ("lucy", configs.lucy_camera, 1080, 1920, dict(RayDepth=5)),
    "full_helmet_sort_d4": ("helmet", configs.helmet_camera, 1920, 1080, dict(RayDepth=4, DoRaySorting=1)),
    # several BLAS instances at scale: the instance loop (BVHIntersect.glsl:275-287) and the TLAS walk (:205-272) inside the persistent kernel
    "full_multi_instances_d3": ("multi900k", lambda w, h: S.Camera(w, h), 1920, 1080, dict(RayDepth=3)),
    "full_multi_tlas_sort_d3": ("multi900k", lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 0.0)), 1920, 1080, dict(RayDepth=3, UseTlas=1, DoRaySorting=1)),
    # lights as surfaces, an emissive ceiling, bilinear / repeat textures, at full HD
    "full_cornell_lights_textures_d4": ("cornell_lights_tex", S.cornell_camera, 1920, 1080, dict(RayDepth=4, DoTraceLights=1)),
    # more than 2^21 triangles with DoRaySorting: triangle ids alias in the reference's 21-bit sort key (NHit/compute.glsl:81) — the reference's own CountingSort on the aliased keys
    "full_soup2m3_sort_d4": ("soup2m3", lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 0.0)), 1920, 1080, dict(RayDepth=4, DoRaySorting=1)),
    "full_soup4m_4k_d9": ("soup4m", lambda w, h: S.Camera(w, h, position=(0.0, 0.0, 0.0), view_dir=(0.3, 0.1, -1.0)), 3840, 2160, dict(RayDepth=9)),   # configs[4] stand-in (sample 0 of its 4 spp)
}
FULL_SAMPLE_STRIDE = {None: 127, "full_soup4m_4k_d9": 1016}      # every n-th ray of a stage is kept in the fixture (a 4K frame of nine stages at 127 would be 7 MB)

"""glref — run the REFERENCE's own GLSL path tracer on the CPU (Mesa llvmpipe) to pin the oracle.

TEST INFRASTRUCTURE ONLY.  Works only where /root/reference and Mesa's swrast_dri.so exist (this container, not
the GPU box): the outputs it produces are committed as fixtures under tests/golden/glref/ by make_vectors.py.

What is the reference's and what is ours
  * the shader text is read at run time from /root/reference/IDKEngine/Resource/Shaders (never copied into the repo);
  * `preprocess()` does what BBG/Source/Objects/Shader.cs:177-285 does to it (AppInclude once-only includes,
    AppInsert values, unused-SSBO removal :292-335, the APP_* defines :199-201);
  * `adapt_for_llvmpipe()` applies the textual adaptations llvmpipe needs — listed in ADAPTATIONS below; none of them
    touches arithmetic, control flow or data layout of the path;
  * `ReferencePathTracer` issues the dispatches of Source/Render/PathTracer.cs:214-297 (Compute / RaySorting).

Queue order.  The reference enqueues surviving rays through atomicAdd, so its alive-queue order (and with it the
slot-seeded RNG of NHit, NHit/compute.glsl:54) depends on GPU scheduling.  Every order is a valid execution; the
oracle fixes ONE (FirstHit: increasing pixel index, NHit: increasing old slot).  With
canonical_order=True this host rewrites the queue (and the cached sort keys) into that order between dispatches —
a permutation of the reference's own output, nothing is recomputed.
"""
import ctypes as C
import os
import re
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SHADERS = "/root/reference/IDKEngine/Resource/Shaders"
_SO = os.path.join(_HERE, "..", "_ref", "libglref.so")

ADAPTATIONS = """
 A1  `#extension GL_ARB_bindless_texture : require` is dropped (llvmpipe has no bindless textures); sampler-typed
     members of structs / blocks (GpuMaterial's five samplers, SkyBoxUBO, GBufferDataUBO, GpuPointShadow) become
     `uvec2` — the same 8 bytes at the same offsets, so every buffer layout is unchanged.
 A2  `texture(gpuMaterial.X, uv)` -> `glrefMaterialTexture(gpuMaterial.X, uv)`, a switch over bound sampler2D units
     indexed by the handle's low word (0 = the 1x1 white default, Utils/ModelLoader.cs:1857-1877);
     `texture(skyBoxUBO.Albedo, d)` -> `texture(glrefSkyBox, d)`.
 A3  SSBO binding points >= GL_MAX_SHADER_STORAGE_BUFFER_BINDINGS are renumbered into free ones (host binds to match).
 A4  `#version 460 core` is compiled under MESA_GL(SL)_VERSION_OVERRIDE=4.6/460 (llvmpipe advertises 4.5).
 A5  GL_AMD_gpu_shader_half_float(_fetch) are `enable`-only in the source and absent here, so the shaders take
     their own sampler2D path (FirstHit/compute.glsl:7-9).
 A6  GL_KHR_shader_subgroup_arithmetic (CountingSort/BlellochScan/GroupWise only): if absent, subgroupExclusiveAdd /
     subgroupAdd are emulated through shared memory over groups of MIN_SUBGROUP_SIZE invocations (integer sums: the
     result does not depend on the subgroup size).
 A8  (ShadowsRayTraced only) `texelFetch(gBufferDataUBO.Depth/Normal, ..)` read bound sampler2D units and
     `image2D(pointShadow.RayTracedShadowMapImage)` is a bound rgba32f image (llvmpipe has no bindless images either).
 A9  (opt-in, ReferencePathTracer(dump_uv=True): the textured-stage check of oracle/glref/fuzz_reference.py) FirstHit and NHit additionally WRITE the texture
     coordinate they interpolated (`interpTexCoord`, FirstHit/compute.glsl:129, NHit/compute.glsl:115) to an extra SSBO indexed by the ray index: one store behind the
     statement that computes it, one global that carries the ray index into TraceRay; no statement of the reference is changed or removed.
 A7  (host order, only with DoRaySorting; sort_count_fix=True) PingPongIndex is uploaded BEFORE RaySorting() instead of
     after it.  Reference defect D1: PathTracer.cs:232-244 runs Reorder while the buffer still holds the previous
     bounce's PingPongIndex, so Reorder's GetItemCount() (Reorder/compute.glsl:44-47) returns the PREVIOUS bounce's
     (larger) count and its bounds check never fires; the up-to-31 tail invocations of the last work group then
     re-insert stale queue entries with stale keys, displacing live rays (the sorted queue is no longer a permutation
     of the alive queue).  The oracle and the product implement the evident intent (a stable counting sort of exactly
     the alive rays); sort_count_fix=False reproduces the defect for the record (tests/test_glref.py).
"""


def available():
    return os.path.isdir(REF_SHADERS) and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")


def build(force=False):
    src = os.path.join(_HERE, "glref.c")
    so = os.path.abspath(_SO)
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-Wno-comment", src, "-o", so, "-ldl"])
    return so


_GL = None


def gl():
    """The ctypes view of libglref.so with a current GL context (created on first use)."""
    global _GL
    if _GL is not None:
        return _GL
    os.environ.setdefault("MESA_GL_VERSION_OVERRIDE", "4.6")        # A4
    os.environ.setdefault("MESA_GLSL_VERSION_OVERRIDE", "460")
    os.environ.setdefault("LP_NUM_THREADS", "0")                     # workgroups run in order on this thread
    L = C.CDLL(build())
    L.glref_info.restype = C.c_char_p
    L.glref_compile_compute.restype = C.c_uint; L.glref_compile_compute.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.glref_buffer.restype = C.c_uint; L.glref_buffer.argtypes = [C.c_void_p, C.c_size_t]
    L.glref_buffer_write.argtypes = [C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p]
    L.glref_buffer_read.argtypes = [C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p]
    L.glref_buffer_copy.argtypes = [C.c_uint, C.c_uint, C.c_size_t, C.c_size_t, C.c_size_t]
    L.glref_bind_ssbo.argtypes = [C.c_uint, C.c_uint]; L.glref_bind_ubo.argtypes = [C.c_uint, C.c_uint]
    L.glref_texture2d.restype = C.c_uint; L.glref_texture2d.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.glref_cubemap.restype = C.c_uint; L.glref_cubemap.argtypes = [C.c_int, C.c_void_p, C.c_int]
    L.glref_texture2d_state.restype = C.c_uint; L.glref_texture2d_state.argtypes = [C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 4
    L.glref_texture_write.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_void_p]
    L.glref_texture_read.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_void_p]
    L.glref_bind_texture.argtypes = [C.c_uint, C.c_uint]; L.glref_bind_image.argtypes = [C.c_uint, C.c_uint]
    L.glref_dispatch.argtypes = [C.c_uint] * 4
    L.glref_dispatch_indirect.argtypes = [C.c_uint, C.c_uint, C.c_size_t]
    L.glref_has_extension.argtypes = [C.c_char_p]
    L.glref_set_uniform_1i.argtypes = [C.c_uint, C.c_char_p, C.c_int]
    L.glref_set_uniform_1ui.argtypes = [C.c_uint, C.c_char_p, C.c_uint]
    L.glref_get_integer.argtypes = [C.c_uint]; L.glref_get_integer_i.argtypes = [C.c_uint, C.c_uint]
    L.glref_delete_buffer.argtypes = [C.c_uint]; L.glref_delete_texture.argtypes = [C.c_uint]; L.glref_delete_program.argtypes = [C.c_uint]
    rc = L.glref_init()
    if rc != 0:
        raise RuntimeError(f"glref_init failed ({rc}): {L.glref_info().decode()}")
    _GL = L
    return L


# ------------------------------------------------------------------------------------------------ preprocessing

_KEYWORD = re.compile(r"(AppInsert|AppInclude)\((.*?)\)")


def _next_keyword(source, start):
    """First AppInsert(..)/AppInclude(..) at or after `start` that is not behind `//` on its line (Shader.cs:337-360)."""
    pos = start
    while True:
        m = _KEYWORD.search(source, pos)
        if m is None:
            return None
        line_start = source.rfind("\n", 0, m.start()) + 1
        if "//" in source[line_start:m.start()]:
            pos = m.end()
            continue
        return m


def preprocess(local_path, insertions, stage="COMPUTE", vendor="UNKNOWN", min_subgroup=8, source=None):
    """The reference's Preprocessor.PreProcess (BBG/Source/Objects/Shader.cs:177-285) on a shader of REF_SHADERS
    (or on `source`, a driver shader of ours that AppIncludes the reference's files)."""
    included = []

    def resolve(source):
        out = []
        pos = 0
        while True:
            m = _next_keyword(source, pos)
            if m is None:
                out.append(source[pos:])
                break
            out.append(source[pos:m.start()])
            kind, key = m.group(1), m.group(2)
            if kind == "AppInsert":
                out.append(str(insertions.get(key, "0")))          # "0" is the reference's fallback (:232-236)
            else:
                path = os.path.join(REF_SHADERS, key)
                if path in included:
                    out.append(f"// Omitted including \"{key}\" as it's already part of this file\n")
                else:
                    included.append(path)
                    with open(path, "r", encoding="utf-8-sig") as f:
                        text = f.read()
                    out.append(f"// Including \"{key}\"\n")
                    out.append(resolve(text))
                    out.append(f"\n// Included \"{key}\"\n")
            pos = m.end()
        return "".join(out)

    if source is None:
        with open(os.path.join(REF_SHADERS, local_path), "r", encoding="utf-8-sig") as f:
            source = f.read()
    text = resolve(source)
    text = _remove_unused_ssbos(text)
    m = re.search(r"#version .*\n*", text)
    after = m.end() if m else 0
    header = ("#extension GL_ARB_bindless_texture : require\n"
              "#extension GL_EXT_shader_image_load_formatted : require\n"
              f"#define APP_SHADER_STAGE_{stage} 1\n#define APP_VENDOR_{vendor} 1\n#define MIN_SUBGROUP_SIZE {min_subgroup}\n")
    return text[:after] + header + text[after:]


_SSBO_DECL = re.compile(r"layout\s*\([^)]*\)\s*(?:\b\w+\b\s*)*buffer\b[\s\S]*?}\s*(\w+)\s*;\s*")


def _remove_unused_ssbos(text):
    """Shader.cs:292-335: drop storage blocks whose instance name is never used as `name.` after the declaration starts."""
    out = []
    pos = 0
    while True:
        m = _SSBO_DECL.search(text, pos)
        if m is None:
            out.append(text[pos:])
            break
        used = re.search(r"\b%s\." % re.escape(m.group(1)), text[pos:]) is not None
        out.append(text[pos:m.end() if used else m.start()])
        pos = m.end()
    return "".join(out)


_SAMPLER_MEMBER = re.compile(r"^(\s*)(samplerCubeShadow|samplerCube|sampler2D|MATERIAL_SAMPLER_2D_TYPE)(\s+\w+\s*;)", re.M)
_SUBGROUP_EMULATION = """
// glref A6: llvmpipe of this Mesa has no GL_KHR_shader_subgroup.  Subgroups of MIN_SUBGROUP_SIZE consecutive
// invocations are emulated through shared memory; the integer sums do not depend on the subgroup size.
#define gl_SubgroupSize uint(MIN_SUBGROUP_SIZE)
#define gl_SubgroupInvocationID (gl_LocalInvocationIndex % uint(MIN_SUBGROUP_SIZE))
#define gl_SubgroupID (gl_LocalInvocationIndex / uint(MIN_SUBGROUP_SIZE))
#define gl_NumSubgroups ((gl_WorkGroupSize.x * gl_WorkGroupSize.y * gl_WorkGroupSize.z) / uint(MIN_SUBGROUP_SIZE))
shared uint glrefSubgroupScratch[1024];
uint subgroupExclusiveAdd(uint v)
{
    glrefSubgroupScratch[gl_LocalInvocationIndex] = v;
    barrier();
    uint first = gl_LocalInvocationIndex - gl_LocalInvocationIndex % uint(MIN_SUBGROUP_SIZE);
    uint sum = 0u;
    for (uint i = first; i < gl_LocalInvocationIndex; i++) sum += glrefSubgroupScratch[i];
    barrier();
    return sum;
}
uint subgroupAdd(uint v)
{
    glrefSubgroupScratch[gl_LocalInvocationIndex] = v;
    barrier();
    uint first = gl_LocalInvocationIndex - gl_LocalInvocationIndex % uint(MIN_SUBGROUP_SIZE);
    uint sum = 0u;
    for (uint i = first; i < first + uint(MIN_SUBGROUP_SIZE); i++) sum += glrefSubgroupScratch[i];
    barrier();
    return sum;
}
"""


def adapt_for_llvmpipe(src, n_textures, max_ssbo_bindings, have_image_formatted=True, have_subgroup=True):
    """Apply ADAPTATIONS A1-A3, A6.  Returns (source, ssbo_binding_remap)."""
    src = src.replace("#extension GL_ARB_bindless_texture : require\n", "")                       # A1
    if not have_image_formatted:
        src = src.replace("#extension GL_EXT_shader_image_load_formatted : require\n", "")
        src = re.sub(r"layout\(binding = (\d)\)(\s+restrict(?:\s+readonly)?(?:\s+writeonly)?\s+uniform\s+image2D)", r"layout(binding = \1, rgba32f)\2", src)
    src = _SAMPLER_MEMBER.sub(r"\1uvec2\3", src)                                                   # A1
    src = re.sub(r"\btexture\(\s*gpuMaterial\.(\w+)\s*,", r"glrefMaterialTexture(gpuMaterial.\1,", src)   # A2
    src = re.sub(r"\btexture\(\s*skyBoxUBO\.Albedo\s*,", "texture(glrefSkyBox,", src)
    cases = "".join(f"        case {i}u: return texture(glrefTex{i}, uv);\n" for i in range(n_textures + 1))
    decls = "".join(f"layout(binding = {1 + i}) uniform sampler2D glrefTex{i};\n" for i in range(n_textures + 1))
    helper = ("layout(binding = 0) uniform samplerCube glrefSkyBox;\n" + decls +
              "vec4 glrefMaterialTexture(uvec2 handle, vec2 uv)\n{\n    switch (handle.x)\n    {\n" + cases +
              "    }\n    return vec4(1.0);\n}\n")
    if not have_subgroup and "GL_KHR_shader_subgroup" in src:
        src = re.sub(r"#extension GL_KHR_shader_subgroup\w* : require\n", "", src)
        helper += _SUBGROUP_EMULATION
    at = max(m.end() for m in re.finditer(r"#define MIN_SUBGROUP_SIZE .*\n|#extension .*\n", src))   # after the last directive that must lead
    src = src[:at] + helper + src[at:]
    remap = {}                                                                                     # A3
    used = set(int(b) for b in re.findall(r"layout\(std430, binding = (\d+)", src))
    free = [b for b in range(max_ssbo_bindings) if b not in used]
    for b in sorted(used):
        if b >= max_ssbo_bindings:
            remap[b] = free.pop()
    for old, new in remap.items():
        src = re.sub(r"layout\(std430, binding = %d\b" % old, "layout(std430, binding = %d" % new, src)
    return src, remap


UV_DUMP_BINDING = 37      # A9: the extra SSBO (renumbered by A3 like every other binding beyond the limit)


def inject_uv_dump(src):
    """A9: see ADAPTATIONS.  Applied to the preprocessed FirstHit / NHit source BEFORE A1-A3."""
    decl = ("layout(std430, binding = %d) restrict writeonly buffer GlrefUvDumpSSBO { vec2 Uv[]; } glrefUvDumpSSBO;\nuint glrefRayIndex = 0u;\n" % UV_DUMP_BINDING)
    at = src.index("void main()")
    src = src[:at] + decl + src[at:]
    n0 = src.count("glrefRayIndex")
    src = src.replace("    bool continueRay = TraceRay(wavefrontRay, aovRay);", "    glrefRayIndex = uint(imgCoord.y * imageSize(ImgResult).x + imgCoord.x);\n    bool continueRay = TraceRay(wavefrontRay, aovRay);")   # FirstHit main (:76-78)
    src = src.replace("    bool continueRay = TraceRay(wavefrontRay, aovRay, sortingKey);", "    glrefRayIndex = rayIndex;\n    bool continueRay = TraceRay(wavefrontRay, aovRay, sortingKey);")                                        # NHit main (:56-62)
    assert src.count("glrefRayIndex") == n0 + 1, "A9: the call of TraceRay in main() was not found"
    m = re.search(r"^(\s*)vec2 interpTexCoord = Interpolate\([^;]*;\n", src, re.M)
    assert m and len(re.findall(r"vec2 interpTexCoord = ", src)) == 1, "A9: the interpolation of the texture coordinate was not found"
    return src[:m.end()] + m.group(1) + "glrefUvDumpSSBO.Uv[glrefRayIndex] = interpTexCoord;\n" + src[m.end():]


def compile_compute(src, what):
    L = gl()
    log = C.create_string_buffer(1 << 16)
    prog = L.glref_compile_compute(src.encode("utf-8"), log, len(log))
    if not prog:
        raise RuntimeError(f"llvmpipe failed to compile {what}:\n{log.value.decode(errors='replace')}")
    return prog


# ------------------------------------------------------------------------------------------------ the host schedule

HEADER_BYTES = 28          # GpuWavefrontPTHeader: DispatchCommand(12) Counts[2](8) PingPongIndex(4) AccumulatedSamples(4)
N_HIT_LOCAL_SIZE_X = 32    # PathTracing/include/Constants.glsl:5 (read back from the source in __init__)
GROUP_WISE_PROGRAM_STEPS, DOWN_UP_SWEEP_PROGRAM_STEPS = 10, 11         # PathTracer.cs:153-155
PREFIX_SUM_CAPACITY = 1 << (GROUP_WISE_PROGRAM_STEPS + DOWN_UP_SWEEP_PROGRAM_STEPS)

GL_MAX_SHADER_STORAGE_BUFFER_BINDINGS = 0x90DD
GL_MAX_COMPUTE_SHADER_STORAGE_BLOCKS = 0x90DB


class _ReferenceHost:
    """Scene buffers / uniform blocks / textures at the reference's binding points (include/StaticStorageBuffers.glsl,
    include/StaticUniformBuffers.glsl), shared by the hosts below.  `scene` is idkengine_amd.gputypes.Scene, whose arrays are
    byte-exact mirrors of the reference's GPU structs (include/idkpt_types.h)."""

    def _init_host(self, scene, use_tlas, blas_stack_size, extra_insertions=None):
        L = gl()
        self.L = L
        self._bufs, self._texs, self._progs = [], [], []
        self._max_b = L.glref_get_integer(GL_MAX_SHADER_STORAGE_BUFFER_BINDINGS)
        self._img_fmt = bool(L.glref_has_extension(b"GL_EXT_shader_image_load_formatted"))
        self._subgroup = bool(L.glref_has_extension(b"GL_KHR_shader_subgroup"))
        stack = int(blas_stack_size) if blas_stack_size and blas_stack_size > 0 else int(scene.blas_stack_size)
        self.insertions = {"USE_TLAS": "1" if use_tlas else "0", "BLAS_STACK_SIZE": str(stack)}               # Bvh/BVH.cs:25,43
        self.insertions.update(extra_insertions or {})
        self._ntex = len(scene.textures)
        self.remap = {}
        self.sources = {}

    def _program(self, path, source=None, post=None, pre=None):
        src = preprocess(path, self.insertions, source=source)
        if pre:
            src = pre(src)
        src, remap = adapt_for_llvmpipe(src, self._ntex, self._max_b, self._img_fmt, self._subgroup)
        if post:
            src = post(src)
        self.remap.update(remap)
        self.sources[path] = src
        p = compile_compute(src, path)
        self._progs.append(p)
        return p

    def _ssbo(self, binding, arr):
        arr = np.ascontiguousarray(arr)
        b = self.L.glref_buffer(arr.ctypes.data if arr.nbytes else None, arr.nbytes)
        self._bufs.append(b)
        self.L.glref_bind_ssbo(self.remap.get(binding, binding), b)
        return b

    def _ubo(self, binding, nbytes, data=None):
        b = self.L.glref_buffer(data.ctypes.data if data is not None else None, nbytes)
        self._bufs.append(b)
        self.L.glref_bind_ubo(binding, b)
        return b

    def _bind_scene(self, scene):
        L = self.L
        self.b_mesh = self._ssbo(2, scene.meshes); self.b_mat = self._ssbo(3, scene.materials); self.b_xf = self._ssbo(4, scene.mesh_transforms)
        self.b_vert = self._ssbo(7, scene.vertices); self.b_pos = self._ssbo(8, np.ascontiguousarray(scene.vertex_positions, np.float32))
        self.b_desc = self._ssbo(20, scene.blas_descs); self.b_inst = self._ssbo(21, scene.blas_instances)
        self.b_nodes = self._ssbo(22, scene.blas_nodes); self.b_tris = self._ssbo(23, scene.blas_triangles)
        self.b_tlas = self._ssbo(27, scene.tlas_nodes)
        # UBO 1 per-frame, UBO 2 lights (StaticUniformBuffers.glsl)
        self.b_frame = self._ubo(1, 576)
        lights = np.zeros(256 * 48 + 16, np.uint8)
        lb = np.ascontiguousarray(scene.lights).view(np.uint8).reshape(-1)
        lights[:lb.size] = lb
        lights[256 * 48:256 * 48 + 4] = np.array([len(scene.lights)], np.int32).view(np.uint8)
        self.b_lights = self._ubo(2, lights.nbytes, lights)
        # textures: unit 0 sky cube, unit 1 the 1x1 white default (handle 0), units 2.. the scene's texture table
        sky = scene.sky_faces if scene.sky_faces is not None else np.zeros((6, 1, 1, 4), np.float32)
        sky = np.ascontiguousarray(sky, np.float32)
        # GL_LINEAR, seamless (SkyBoxManager.cs:44,74).  SIX DIFFERENT 1x1 faces are the C-ABI's "constant colour per face" (include/idkpt.h): not a cube map the
        # reference would ever sample (its sky is an HDR or a computed atmosphere); a caller that compares such a case (oracle/glref/fuzz_reference.py) asks for GL_NEAREST
        # (`sky_nearest`), so that the comparison covers everything else of the case.
        t = L.glref_cubemap(sky.shape[1], sky.ctypes.data, 0 if getattr(self, "sky_nearest", False) else 1); self._texs.append(t); L.glref_bind_texture(0, t)
        # every image with ITS sampler state (GLSampler.SamplerState of the glTF sampler, ModelLoader.cs:1166-1197: wrap S / T, magnification filter) and storage format
        from idkengine_amd import gputypes as T
        white = np.ones((1, 1, 4), np.float32)
        for i, tex in enumerate([T.TextureImage(white)] + [T.TextureImage.of(x) for x in scene.textures]):
            t = L.glref_texture2d_state(tex.data.shape[1], tex.data.shape[0], tex.data.ctypes.data, tex.format, tex.wrap_s, tex.wrap_t, tex.mag_filter); self._texs.append(t); L.glref_bind_texture(1 + i, t)

    def _check_gl(self, what):
        err = self.L.glref_error()
        if err:
            raise RuntimeError(f"GL error 0x{err:x} during {what}")

    def close(self):
        for b in self._bufs:
            self.L.glref_delete_buffer(b)
        for t in self._texs:
            self.L.glref_delete_texture(t)
        for p in self._progs:
            self.L.glref_delete_program(p)
        self._bufs, self._texs, self._progs = [], [], []


class ReferencePathTracer(_ReferenceHost):
    """Source/Render/PathTracer.cs driven over llvmpipe.  `scene` is idkengine_amd.gputypes.Scene (whose arrays are
    byte-exact mirrors of the reference's GPU structs, include/idkpt_types.h), `settings` is gputypes.Settings."""

    def __init__(self, scene, width, height, settings, canonical_order=True, sort_count_fix=True, sky_nearest=False, dump_uv=False):
        self.sky_nearest = sky_nearest
        self.dump_uv = dump_uv
        self._init_host(scene, settings.UseTlas, settings.BlasStackSize,
                        {"PATH_TRACER_DO_RAY_SORTING": "1" if settings.DoRaySorting else "0",                 # PathTracer.cs:111
                         "PATH_TRACER_OUTPUT_AOVS": "1" if settings.OutputAOVs else "0"})                      # PathTracer.cs:123
        L = self.L
        self.W, self.H = width, height
        self.canonical = canonical_order
        self.sort_count_fix = sort_count_fix
        self.st = settings
        self.accumulated = 0
        self.alive_counts = []
        self.first_hit = self._program("PathTracing/FirstHit/compute.glsl", pre=inject_uv_dump if dump_uv else None)
        self.n_hit = self._program("PathTracing/NHit/compute.glsl", pre=inject_uv_dump if dump_uv else None)
        self.final_draw = self._program("PathTracing/FinalDraw/compute.glsl")
        if settings.DoRaySorting:
            self.reorder = self._program("PathTracing/CountingSort/Reorder/compute.glsl")
            self.down_up = self._program("PathTracing/CountingSort/BlellochScan/DownUpSweep/compute.glsl")
            self.group_wise = self._program("PathTracing/CountingSort/BlellochScan/GroupWise/compute.glsl")
        self._bind_scene(scene)
        n = width * height
        self.b_rays = self._ssbo(30, np.zeros(n * 48, np.uint8)); self.b_aov = self._ssbo(31, np.zeros(n * 32, np.uint8))
        hdr = np.zeros(HEADER_BYTES // 4 + n, np.uint32); hdr[1] = 1; hdr[2] = 1     # NumGroups = (0,1,1) (PathTracer.cs:323-326)
        self.b_pt = self._ssbo(32, hdr)
        self.b_sorted = self._ssbo(33, np.zeros(n, np.uint32)); self.b_keys = self._ssbo(34, np.zeros(n, np.uint32))
        self.b_wg_prefix = self._ssbo(35, np.zeros(PREFIX_SUM_CAPACITY, np.uint32))
        self.b_wg_sums = self._ssbo(36, np.zeros(PREFIX_SUM_CAPACITY >> GROUP_WISE_PROGRAM_STEPS, np.uint32))
        if dump_uv:
            self.b_uv = self._ssbo(UV_DUMP_BINDING, np.full(2 * n, np.nan, np.float32))     # A9: NaN = "this ray interpolated no texture coordinate in the stage"
        self.b_settings = self._ubo(0, 32)                                            # UBO 0 settings (std140: float float bool bool bool)
        # Result / Albedo / Normal images (PathTracer.cs:301-318)
        zero = np.zeros((height, width, 4), np.float32)
        self.img = []
        for _ in range(3):
            t = L.glref_texture2d(width, height, zero.ctypes.data, 1, 0); self._texs.append(t); self.img.append(t)
        self._check_gl("set-up")

    # ---- state ----
    def set_camera(self, cam):
        pf = np.zeros(576 // 4, np.float32)
        pf[32:48] = np.asarray(cam.inv_view, np.float32).reshape(16)            # InvView        @128
        pf[64:67] = np.asarray(cam.position, np.float32)                        # ViewPos        @256
        pf[84:100] = np.asarray(cam.inv_projection, np.float32).reshape(16)     # InvProjection  @336
        self.L.glref_buffer_write(self.b_frame, 0, pf.nbytes, pf.ctypes.data)

    def _u32(self, buf, off, count):
        out = np.zeros(count, np.uint32)
        if count:
            self.L.glref_buffer_read(buf, off, out.nbytes, out.ctypes.data)
        return out

    def _w32(self, buf, off, arr):
        arr = np.ascontiguousarray(arr, np.uint32)
        if arr.size:
            self.L.glref_buffer_write(buf, off, arr.nbytes, arr.ctypes.data)

    def header(self):
        h = self._u32(self.b_pt, 0, 7)
        return {"groups": h[0:3].astype(np.int32), "counts": h[3:5], "pingpong": int(h[5]), "accumulated": int(h[6])}

    def alive_queue(self, count=None):
        if count is None:
            count = self._last_count
        return self._u32(self.b_pt, HEADER_BYTES, count)

    def rays(self, dtype):
        out = np.zeros(self.W * self.H, dtype)
        self.L.glref_buffer_read(self.b_rays, 0, out.nbytes, out.ctypes.data)
        return out

    def take_uv_dump(self):
        """A9: the texture coordinates the last stage(s) interpolated, per ray index ((n, 2) float32, NaN where none was); the buffer is reset to NaN."""
        n = self.W * self.H
        out = np.zeros(2 * n, np.float32)
        self.L.glref_buffer_read(self.b_uv, 0, out.nbytes, out.ctypes.data)
        nan = np.full(2 * n, np.nan, np.float32)
        self.L.glref_buffer_write(self.b_uv, 0, nan.nbytes, nan.ctypes.data)
        return out.reshape(n, 2)

    def aov(self, dtype):
        out = np.zeros(self.W * self.H, dtype)
        self.L.glref_buffer_read(self.b_aov, 0, out.nbytes, out.ctypes.data)
        return out

    def image(self, which=0):
        out = np.zeros((self.H, self.W, 4), np.float32)
        self.L.glref_texture_read(self.img[which], self.W, self.H, out.ctypes.data)
        return out

    # ---- PathTracer.Compute, one sample (PathTracer.cs:214-271) ----
    def render(self, on_stage=None):
        L, g = self.L, self.st.Gpu
        s = np.zeros(8, np.uint32)
        s[0:2] = np.array([g.FocalLength, g.LenseRadius], np.float32).view(np.uint32)
        s[2], s[3], s[4] = int(g.DoDebugBVHTraversal), int(g.DoTraceLights), int(g.DoRussianRoulette)
        L.glref_buffer_write(self.b_settings, 0, 32, s.ctypes.data)                              # BBG.Cmd.SetUniforms(settings)
        self._w32(self.b_pt, 24, [self.accumulated])                                              # AccumulatedSamples setter (:28-36)
        for i in range(3):
            L.glref_bind_image(i, self.img[i])
        L.glref_dispatch(self.first_hit, (self.W + 7) // 8, (self.H + 7) // 8, 1)
        L.glref_barrier()
        self.alive_counts = []
        prev_in = None
        for j in range(1, int(self.st.RayDepth)):
            ping = j % 2
            count = int(self._u32(self.b_pt, 12 + 4 * ping, 1)[0])                                # Counts[pingPongIndex] as NHit j will read it
            if self.canonical:
                self._canonicalise(j, count, prev_in)
            if self.st.DoRaySorting:
                if j > 1:
                    if self.sort_count_fix:
                        self._w32(self.b_pt, 20, [ping])                                          # A7 (reference defect D1)
                    self._ray_sorting(count)
                self._w32(self.b_wg_prefix, 0, np.zeros(PREFIX_SUM_CAPACITY, np.uint32))          # workGroupPrefixSumBuffer.Fill(0)
            prev_in = self.alive_queue(count)
            self.alive_counts.append(count)
            self._last_count = count
            if on_stage:
                on_stage(j, self)
            self._w32(self.b_pt, 20, [ping])                                                      # PingPongIndex
            self._w32(self.b_pt, 12 + 4 * (1 - ping), [0])                                        # Counts[1 - ping] = 0
            L.glref_dispatch_indirect(self.n_hit, self.b_pt, 0)
            L.glref_barrier()
        final_ping = int(self.st.RayDepth) % 2 if int(self.st.RayDepth) > 1 else 1
        self._last_count = int(self._u32(self.b_pt, 12 + 4 * final_ping, 1)[0])
        if self.canonical and int(self.st.RayDepth) >= 1:
            self._canonicalise(int(self.st.RayDepth), self._last_count, prev_in)
        self.final_alive = self.alive_queue(self._last_count)
        L.glref_dispatch(self.final_draw, (self.W + 7) // 8, (self.H + 7) // 8, 1)
        L.glref_barrier()
        self.accumulated += 1
        err = L.glref_error()
        if err:
            raise RuntimeError(f"GL error 0x{err:x} during render")

    def run_nhit_from(self, rays, queue, j, aov=None, sort_first=False, keys=None):
        """One NHit dispatch (bounce j) from a GIVEN state: `rays` (W*H GpuWavefrontRay records) and `queue` (alive ray
        indices in slot order), as left by j-1 completed bounces.  Returns (rays after, surviving queue in canonical order).
        Lets a checker compare every bounce from identical inputs, so that one flipped discrete decision (a 1-ulp
        difference at an edge or at a Russian-roulette threshold) does not shift every later slot-seeded RNG stream.
        sort_first (DoRaySorting, j > 1): run RaySorting() on the queue first, with the keys and histogram the previous
        run_nhit_from call (bounce j-1, whose surviving queue must be `queue`) left in the buffers — or, with `keys` (one per queue entry), with exactly these
        keys: they are part of the state between two dispatches, and a caller that forces the state forces them too (the cached-key buffer and the histogram NHit
        j-1 builds with atomicAdd, NHit/compute.glsl:80-85, are rewritten from `keys`).  One ray whose closest hit is a tie between two copies of a pre-split
        triangle reports another TriangleId under another rounding of 1/dir; its key then moves it to another run and every slot in between shifts by one."""
        L = self.L
        ping = j % 2
        rays = np.ascontiguousarray(rays); queue = np.ascontiguousarray(queue, np.uint32)
        L.glref_buffer_write(self.b_rays, 0, rays.nbytes, rays.ctypes.data)
        if aov is not None:
            aov = np.ascontiguousarray(aov); L.glref_buffer_write(self.b_aov, 0, aov.nbytes, aov.ctypes.data)
        hdr = np.zeros(7, np.uint32)
        hdr[0] = (len(queue) + N_HIT_LOCAL_SIZE_X - 1) // N_HIT_LOCAL_SIZE_X; hdr[1] = hdr[2] = 1
        hdr[3 + ping] = len(queue); hdr[5] = ping; hdr[6] = self.accumulated
        self._w32(self.b_pt, 0, hdr)
        self._w32(self.b_pt, HEADER_BYTES, queue)
        g = self.st.Gpu
        s = np.zeros(8, np.uint32)
        s[0:2] = np.array([g.FocalLength, g.LenseRadius], np.float32).view(np.uint32)
        s[2], s[3], s[4] = int(g.DoDebugBVHTraversal), int(g.DoTraceLights), int(g.DoRussianRoulette)
        L.glref_buffer_write(self.b_settings, 0, 32, s.ctypes.data)
        if self.st.DoRaySorting:
            if sort_first and j > 1 and len(queue):
                if keys is not None:
                    k = np.ascontiguousarray(keys, np.uint32)
                    assert len(k) == len(queue) and (k.max() < PREFIX_SUM_CAPACITY if len(k) else True)
                    self._w32(self.b_keys, 0, k)
                    self._w32(self.b_wg_prefix, 0, np.bincount(k, minlength=PREFIX_SUM_CAPACITY).astype(np.uint32))
                self._ray_sorting(len(queue))
                queue = self.alive_queue(len(queue))
            self._w32(self.b_wg_prefix, 0, np.zeros(PREFIX_SUM_CAPACITY, np.uint32))
        if len(queue):
            L.glref_dispatch_indirect(self.n_hit, self.b_pt, 0)
            L.glref_barrier()
        count = int(self._u32(self.b_pt, 12 + 4 * (1 - ping), 1)[0])
        out = self.alive_queue(count)
        rank = np.full(self.W * self.H, -1, np.int64); rank[queue] = np.arange(len(queue))
        order = np.argsort(rank[out], kind="stable")
        self.last_out_keys = None
        if self.st.DoRaySorting and count:
            k = self._u32(self.b_keys, 0, count); self._w32(self.b_keys, 0, k[order])
            self.last_out_keys = k[order].copy()               # the keys this dispatch cached, entry by entry of the returned queue
        return self.rays(rays.dtype), out[order]

    def _canonicalise(self, j, count, prev_in):
        """Rewrite the queue NHit j is about to read into the oracle's canonical order (module docstring)."""
        q = self.alive_queue(count)
        if j == 1 or prev_in is None:
            order = np.argsort(q, kind="stable")                       # FirstHit: increasing pixel index
        else:
            rank = np.full(self.W * self.H, -1, np.int64)
            rank[prev_in] = np.arange(len(prev_in))                    # NHit: increasing old slot
            order = np.argsort(rank[q], kind="stable")
        self._w32(self.b_pt, HEADER_BYTES, q[order])
        if self.st.DoRaySorting and j > 1:
            k = self._u32(self.b_keys, 0, count)
            self._w32(self.b_keys, 0, k[order])

    def _ray_sorting(self, count=None):
        """PathTracer.RaySorting (PathTracer.cs:273-297).  Reorder places an entry at blockOffset + atomicAdd(PrefixSum[key], 1) (Reorder/compute.glsl:24-25):
        WHICH of the entries of one key gets which slot of the key's run is scheduling-dependent in the reference (llvmpipe runs work groups on several threads;
        with more than 2^21 triangles, where far-apart entries alias to one key, it shows).  Every such order is a valid execution; with canonical_order the host
        checks that the reference's output is a counting sort of exactly these keys — every entry inside the run of its own key — and rewrites each run into the
        order the oracle fixes (increasing old slot = the stable sort): again a permutation of the reference's own output, nothing is recomputed."""
        L = self.L
        q0 = k0 = None
        if self.canonical and count:
            q0 = self.alive_queue(count); k0 = self._u32(self.b_keys, 0, count)
        L.glref_dispatch(self.group_wise, PREFIX_SUM_CAPACITY >> GROUP_WISE_PROGRAM_STEPS, 1, 1); L.glref_barrier()
        L.glref_dispatch(self.down_up, 1, 1, 1); L.glref_barrier()
        L.glref_dispatch_indirect(self.reorder, self.b_pt, 0); L.glref_barrier()
        if q0 is not None and self.sort_count_fix:
            sq = self._u32(self.b_sorted, 0, count)
            key_of = np.zeros(self.W * self.H, np.uint32); key_of[q0] = k0
            order = np.argsort(k0, kind="stable")
            if not (np.array_equal(key_of[sq], k0[order]) and np.array_equal(np.sort(sq), np.sort(q0))):
                raise RuntimeError("the reference's Reorder output is not a counting sort of the cached keys")
            self.sort_entries_moved_inside_their_runs = getattr(self, "sort_entries_moved_inside_their_runs", 0) + int((sq != q0[order]).sum())
            self._w32(self.b_sorted, 0, q0[order])
        L.glref_buffer_copy(self.b_sorted, self.b_pt, 0, HEADER_BYTES, self.W * self.H * 4)


# ------------------------------------------------------------------------------------------------ ray queries (BVHIntersect.glsl)

_QUERY_DRIVER = """#version 460 core
// glref driver (ours): one invocation per query calls the REFERENCE's TraceRay / TraceRayAny (include/BVHIntersect.glsl:183-411).
AppInclude(include/Ray.glsl)
AppInclude(include/StaticStorageBuffers.glsl)
AppInclude(include/StaticUniformBuffers.glsl)
#define TRAVERSAL_STACK_USE_SHARED_STACK_SIZE 64
AppInclude(include/BVHIntersect.glsl)
layout(local_size_x = 64, local_size_y = 1, local_size_z = 1) in;
struct GlrefQuery { float ox, oy, oz, maxDist, dx, dy, dz; uint pad; };
struct GlrefHit { float t, bx, by; uint tri, xform, hit, p0, p1; };
layout(std430, binding = 18) restrict readonly buffer GlrefQuerySSBO { GlrefQuery q[]; } glrefQuerySSBO;
layout(std430, binding = 19) restrict writeonly buffer GlrefHitSSBO { GlrefHit h[]; } glrefHitSSBO;
uniform int GlrefCount;
uniform int GlrefAnyHit;
uniform int GlrefTraceLights;
void main()
{
    uint i = gl_GlobalInvocationID.x;
    if (i >= uint(GlrefCount)) return;
    GlrefQuery q = glrefQuerySSBO.q[i];
    Ray ray = Ray(vec3(q.ox, q.oy, q.oz), vec3(q.dx, q.dy, q.dz));
    HitInfo hitInfo;
    bool hit;
    if (GlrefAnyHit != 0) hit = TraceRayAny(ray, hitInfo, GlrefTraceLights != 0, q.maxDist);
    else hit = TraceRay(ray, hitInfo, GlrefTraceLights != 0, q.maxDist);
    GlrefHit o;
    o.t = hitInfo.T; o.bx = hitInfo.BaryXY.x; o.by = hitInfo.BaryXY.y; o.tri = hitInfo.TriangleId; o.xform = hitInfo.MeshTransformId;
    o.hit = hit ? 1u : 0u; o.p0 = 0u; o.p1 = 0u;
    glrefHitSSBO.h[i] = o;
}
"""


class ReferenceRayQuery(_ReferenceHost):
    """The reference's TraceRay / TraceRayAny (maxDist, traceLights, USE_TLAS on/off) on arbitrary rays: what idkptTraceRays restates."""

    def __init__(self, scene, use_tlas=False, blas_stack_size=0):
        self._init_host(scene, use_tlas, blas_stack_size)
        self.prog = self._program("glref/RayQuery/compute.glsl", source=_QUERY_DRIVER)
        self._bind_scene(scene)
        self._check_gl("set-up")

    def trace(self, rays, any_hit=False, trace_lights=False):
        """rays: gputypes.RayQuery records.  Returns gputypes.RayHit-shaped records; fields the reference leaves undefined on a miss
        (barycentrics, ids) carry whatever the shader's uninitialised `out` struct held."""
        L = self.L
        rays = np.ascontiguousarray(rays)
        n = len(rays)
        out = np.zeros(n, np.dtype([("T", "<f4"), ("BaryX", "<f4"), ("BaryY", "<f4"), ("TriangleId", "<u4"), ("MeshTransformId", "<u4"), ("Hit", "<u4"), ("_pad0", "<u4"), ("_pad1", "<u4")]))
        bq = self._ssbo(18, rays); bh = self._ssbo(19, out)
        L.glref_set_uniform_1i(self.prog, b"GlrefCount", n); L.glref_set_uniform_1i(self.prog, b"GlrefAnyHit", int(bool(any_hit)))
        L.glref_set_uniform_1i(self.prog, b"GlrefTraceLights", int(bool(trace_lights)))
        L.glref_dispatch(self.prog, (n + 63) // 64, 1, 1); L.glref_barrier()
        L.glref_buffer_read(bh, 0, out.nbytes, out.ctypes.data)
        self._check_gl("ray query")
        return out


# ------------------------------------------------------------------------------------------------ ShadowsRayTraced/compute.glsl

def _adapt_shadows(src):
    """A8 (this shader only): the G-buffer samplers of GBufferDataUBO and the bindless shadow image become bound units."""
    src = src.replace("texelFetch(gBufferDataUBO.Depth,", "texelFetch(glrefGBufferDepth,").replace("texelFetch(gBufferDataUBO.Normal,", "texelFetch(glrefGBufferNormal,")
    src = src.replace("image2D(pointShadow.RayTracedShadowMapImage)", "glrefShadowImage")
    decl = ("layout(binding = 30) uniform sampler2D glrefGBufferDepth;\nlayout(binding = 31) uniform sampler2D glrefGBufferNormal;\n"
            "layout(binding = 3, rgba32f) restrict uniform image2D glrefShadowImage;\n")
    at = src.index("vec4 glrefMaterialTexture(")
    return src[:at] + decl + src[at:]


class ReferenceShadows(_ReferenceHost):
    """Shaders/ShadowsRayTraced/compute.glsl (one point shadow, PointShadowManager.cs) over a caller-supplied G-buffer: what
    idkptTraceShadows restates.  params: gputypes.ShadowParams; NoiseIndex must be a multiple of RayTracingSamples (it is
    (Frame % SampleCount) * RayTracingSamples in the shader)."""

    def __init__(self, scene, use_tlas=False, blas_stack_size=0):
        self._init_host(scene, use_tlas, blas_stack_size)
        self.prog = self._program("ShadowsRayTraced/compute.glsl", post=_adapt_shadows)
        self._bind_scene(scene)
        self.b_shadows = self._ubo(3, 128 * 432 + 16)       # GpuPointShadow[128] (std140, 432 B each with 8-byte handles) + Count
        self.b_taa = self._ubo(4, 32)
        self._check_gl("set-up")

    def trace(self, p, depth, normal_oct, visibility=None):
        L = self.L
        w, h, rts = int(p.Width), int(p.Height), int(p.RayTracingSamples)
        assert int(p.NoiseIndex) % rts == 0
        pf = np.zeros(576 // 4, np.float32)
        pf[100:116] = np.frombuffer(bytes(p.InvProjView), np.float32)                       # InvProjView @400
        pf.view(np.uint32)[67] = int(p.NoiseIndex) // rts                                    # Frame @268
        L.glref_buffer_write(self.b_frame, 0, pf.nbytes, pf.ctypes.data)
        taa = np.zeros(8, np.uint32)
        taa[0:2] = np.array([p.TaaJitter[0], p.TaaJitter[1]], np.float32).view(np.uint32)
        taa[2] = 1 << 20; taa[4] = 1 if int(p.NoiseIndex) else 0                             # SampleCount, TemporalAntiAliasingMode (TAA / none)
        L.glref_buffer_write(self.b_taa, 0, 32, taa.ctypes.data)
        sh = np.zeros(432 // 4, np.uint32); sh[428 // 4] = np.uint32(int(p.LightIndex))      # PointShadows[0].LightIndex
        L.glref_buffer_write(self.b_shadows, 0, 432, sh.ctypes.data)
        cnt = np.array([1], np.int32); L.glref_buffer_write(self.b_shadows, 128 * 432, 4, cnt.ctypes.data)
        d4 = np.zeros((h, w, 4), np.float32); d4[..., 0] = np.asarray(depth, np.float32).reshape(h, w)
        n4 = np.zeros((h, w, 4), np.float32); n4[..., 0:2] = np.asarray(normal_oct, np.float32).reshape(h, w, 2)
        v4 = np.zeros((h, w, 4), np.float32)
        if visibility is not None:
            v4[...] = np.asarray(visibility, np.float32).reshape(h, w, 1)
        td = L.glref_texture2d(w, h, d4.ctypes.data, 0, 0); tn = L.glref_texture2d(w, h, n4.ctypes.data, 0, 0); tv = L.glref_texture2d(w, h, v4.ctypes.data, 0, 0)
        self._texs += [td, tn, tv]
        L.glref_bind_texture(30, td); L.glref_bind_texture(31, tn); L.glref_bind_image(3, tv)
        L.glref_set_uniform_1i(self.prog, b"RayTracingSamples", rts)
        L.glref_dispatch(self.prog, (w + 7) // 8, (h + 7) // 8, 1); L.glref_barrier()
        out = np.zeros((h, w, 4), np.float32)
        L.glref_texture_read(tv, w, h, out.ctypes.data)
        self._check_gl("shadows")
        return out[..., 0].copy()


# ------------------------------------------------------------------------------------------------ BLASRefit / Skinning

class ReferenceSceneUpdates(_ReferenceHost):
    """Shaders/BLASRefit/compute.glsl (Bvh/BVH.cs GpuBlasesRefit) and Shaders/Skinning/compute.glsl (ModelManager.cs) — what
    idkptRefitBlas and idkptSkin restate.  The refit shader synchronises parents with atomicExchange locks; its result does not
    depend on the order the leaves arrive in."""

    def __init__(self, scene):
        self._init_host(scene, False, 0)
        self.refit_prog = self._program("BLASRefit/compute.glsl")
        self.skin_prog = self._program("Skinning/compute.glsl")
        self._bind_scene(scene)
        self.b_parents = self._ssbo(24, np.ascontiguousarray(scene.blas_parent_indices, np.int32))
        self.b_leaves = self._ssbo(25, np.ascontiguousarray(scene.blas_leaf_indices, np.int32).view(np.uint32))
        self.b_locks = self._ssbo(26, np.zeros(max(len(scene.blas_nodes), 1), np.uint32))
        self.n_nodes = len(scene.blas_nodes); self.n_verts = len(scene.vertices)
        self.node_dtype = scene.blas_nodes.dtype; self.vertex_dtype = scene.vertices.dtype
        self.descs = scene.blas_descs
        self.b_prev = self._ssbo(17, np.zeros((self.n_verts, 3), np.float32))
        self._check_gl("set-up")

    def set_positions(self, positions):
        p = np.ascontiguousarray(positions, np.float32)
        self.L.glref_buffer_write(self.b_pos, 0, p.nbytes, p.ctypes.data)

    def refit(self, blas_index):
        L = self.L
        z = np.zeros(max(self.n_nodes, 1), np.uint32); L.glref_buffer_write(self.b_locks, 0, z.nbytes, z.ctypes.data)
        L.glref_set_uniform_1ui(self.refit_prog, b"BlasIndex", int(blas_index))
        leaves = int(self.descs[blas_index]["LeafIndicesCount"])
        L.glref_dispatch(self.refit_prog, (leaves + 63) // 64, 1, 1); L.glref_barrier()
        out = np.zeros(self.n_nodes, self.node_dtype)
        L.glref_buffer_read(self.b_nodes, 0, out.nbytes, out.ctypes.data)
        self._check_gl("refit")
        return out

    def skin(self, unskinned, joints, input_offset, output_offset, joint_offset, count):
        """Returns (positions, previous positions, vertex records) after the dispatch."""
        L = self.L
        un = np.ascontiguousarray(unskinned); jm = np.ascontiguousarray(joints, np.float32)
        self._ssbo(16, un); self._ssbo(15, jm)
        for name, v in ((b"InputVertexOffset", input_offset), (b"OutputVertexOffset", output_offset), (b"JointMatricesOffset", joint_offset), (b"VertexCount", count)):
            L.glref_set_uniform_1ui(self.skin_prog, name, int(v))
        L.glref_dispatch(self.skin_prog, (int(count) + 63) // 64, 1, 1); L.glref_barrier()
        pos = np.zeros((self.n_verts, 3), np.float32); prev = np.zeros((self.n_verts, 3), np.float32); verts = np.zeros(self.n_verts, self.vertex_dtype)
        L.glref_buffer_read(self.b_pos, 0, pos.nbytes, pos.ctypes.data); L.glref_buffer_read(self.b_prev, 0, prev.nbytes, prev.ctypes.data)
        L.glref_buffer_read(self.b_vert, 0, verts.nbytes, verts.ctypes.data)
        self._check_gl("skinning")
        return pos, prev, verts

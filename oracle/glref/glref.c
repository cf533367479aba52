/* glref — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A headless OpenGL 4.5 compute host for running the REFERENCE's own GLSL path-tracer shaders
 * (/root/reference/IDKEngine/Resource/Shaders/PathTracing/**) on the CPU through Mesa llvmpipe, so that the
 * CPU oracle (oracle/ref_pathtracer.cpp) can be pinned against outputs of the reference itself.
 *
 * This image has no X server, EGL or OSMesa, so the context is made by driving swrast_dri.so's DRI_SWRast
 * interface directly (GL/internal/dri_interface.h); GL entry points come from libglapi's dispatch.
 * The API below is a thin, stateless-looking wrapper: buffers, RGBA32F textures / cube maps, image units,
 * compute programs, dispatch (direct / indirect) and barriers — what Source/Render/PathTracer.cs drives.
 * Nothing here restates the algorithm: the shader text is read from /root/reference at run time by glref.py.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <dlfcn.h>
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

static void getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) { (void)d; (void)p; *x = *y = 0; *w = *h = 16; }
static void putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }

static const __DRIswrastLoaderExtension swrastLoader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = getDrawableInfo, .putImage = putImage, .getImage = getImage,
    .putImage2 = putImage2, .getImage2 = getImage2,
};
static const __DRIextension *loaderExt[] = { &swrastLoader.base, NULL };

static void *(*gpa)(const char *);
static char g_info[512];
static int g_ready;

#define GLFN(ret, name, ...) static ret (*p_##name)(__VA_ARGS__)
GLFN(const GLubyte *, glGetString, GLenum);
GLFN(const GLubyte *, glGetStringi, GLenum, GLuint);
GLFN(void, glGetIntegerv, GLenum, GLint *);
GLFN(void, glGetIntegeri_v, GLenum, GLuint, GLint *);
GLFN(GLenum, glGetError, void);
GLFN(GLuint, glCreateShader, GLenum);
GLFN(void, glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
GLFN(void, glCompileShader, GLuint);
GLFN(void, glGetShaderiv, GLuint, GLenum, GLint *);
GLFN(void, glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLFN(GLuint, glCreateProgram, void);
GLFN(void, glAttachShader, GLuint, GLuint);
GLFN(void, glLinkProgram, GLuint);
GLFN(void, glGetProgramiv, GLuint, GLenum, GLint *);
GLFN(void, glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLFN(void, glDeleteShader, GLuint);
GLFN(void, glDeleteProgram, GLuint);
GLFN(void, glUseProgram, GLuint);
GLFN(void, glCreateBuffers, GLsizei, GLuint *);
GLFN(void, glDeleteBuffers, GLsizei, const GLuint *);
GLFN(void, glNamedBufferData, GLuint, GLsizeiptr, const void *, GLenum);
GLFN(void, glNamedBufferSubData, GLuint, GLintptr, GLsizeiptr, const void *);
GLFN(void, glGetNamedBufferSubData, GLuint, GLintptr, GLsizeiptr, void *);
GLFN(void, glCopyNamedBufferSubData, GLuint, GLuint, GLintptr, GLintptr, GLsizeiptr);
GLFN(void, glBindBufferBase, GLenum, GLuint, GLuint);
GLFN(void, glBindBuffer, GLenum, GLuint);
GLFN(void, glCreateTextures, GLenum, GLsizei, GLuint *);
GLFN(void, glDeleteTextures, GLsizei, const GLuint *);
GLFN(void, glTextureStorage2D, GLuint, GLsizei, GLenum, GLsizei, GLsizei);
GLFN(void, glTextureSubImage2D, GLuint, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void *);
GLFN(void, glTextureSubImage3D, GLuint, GLint, GLint, GLint, GLint, GLsizei, GLsizei, GLsizei, GLenum, GLenum, const void *);
GLFN(void, glTextureParameteri, GLuint, GLenum, GLint);
GLFN(void, glGetTextureImage, GLuint, GLint, GLenum, GLenum, GLsizei, void *);
GLFN(void, glBindTextureUnit, GLuint, GLuint);
GLFN(void, glBindImageTexture, GLuint, GLuint, GLint, GLboolean, GLint, GLenum, GLenum);
GLFN(void, glDispatchCompute, GLuint, GLuint, GLuint);
GLFN(void, glDispatchComputeIndirect, GLintptr);
GLFN(void, glMemoryBarrier, GLbitfield);
GLFN(void, glFinish, void);
GLFN(GLint, glGetUniformLocation, GLuint, const GLchar *);
GLFN(void, glProgramUniform1i, GLuint, GLint, GLint);
GLFN(void, glProgramUniform1ui, GLuint, GLint, GLuint);
GLFN(void, glPixelStorei, GLenum, GLint);
GLFN(void, glEnable, GLenum);

#define LOAD(name) do { p_##name = gpa(#name); if (!p_##name) { snprintf(g_info, sizeof g_info, "missing GL entry point %s", #name); return -4; } } while (0)

/* 0 on success. Honour LP_NUM_THREADS from the environment (0 = run workgroups on the calling thread, in order). */
int glref_init(void)
{
    if (g_ready) return 0;
    const char *path = getenv("GLREF_SWRAST_DRI");
    if (!path) path = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { snprintf(g_info, sizeof g_info, "dlopen %s: %s", path, dlerror()); return -1; }
    const __DRIextension **(*getExt)(void) = dlsym(h, "__driDriverGetExtensions_swrast");
    if (!getExt) { snprintf(g_info, sizeof g_info, "no __driDriverGetExtensions_swrast"); return -1; }
    const __DRIextension **ext = getExt();
    const __DRIcoreExtension *core = NULL; const __DRIswrastExtension *sw = NULL;
    for (int i = 0; ext[i]; i++) {
        if (!strcmp(ext[i]->name, __DRI_CORE)) core = (const void *)ext[i];
        if (!strcmp(ext[i]->name, __DRI_SWRAST)) sw = (const void *)ext[i];
    }
    if (!core || !sw || sw->base.version < 4) { snprintf(g_info, sizeof g_info, "DRI_Core/DRI_SWRast v4 missing"); return -1; }
    const __DRIconfig **configs = NULL;
    __DRIscreen *scr = sw->createNewScreen2(0, loaderExt, ext, &configs, NULL);
    if (!scr || !configs || !configs[0]) { snprintf(g_info, sizeof g_info, "createNewScreen2 failed"); return -2; }
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5 };
    unsigned err = 0;
    __DRIcontext *ctx = sw->createContextAttribs(scr, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) { snprintf(g_info, sizeof g_info, "createContextAttribs(4.5 core) failed: %u", err); return -2; }
    __DRIdrawable *dr = sw->createNewDrawable(scr, configs[0], NULL);
    if (!dr || !core->bindContext(ctx, dr, dr)) { snprintf(g_info, sizeof g_info, "bindContext failed"); return -3; }
    gpa = dlsym(RTLD_DEFAULT, "_glapi_get_proc_address");
    if (!gpa) { snprintf(g_info, sizeof g_info, "_glapi_get_proc_address missing"); return -3; }
    LOAD(glGetString); LOAD(glGetStringi); LOAD(glGetIntegerv); LOAD(glGetIntegeri_v); LOAD(glGetError);
    LOAD(glCreateShader); LOAD(glShaderSource); LOAD(glCompileShader); LOAD(glGetShaderiv); LOAD(glGetShaderInfoLog);
    LOAD(glCreateProgram); LOAD(glAttachShader); LOAD(glLinkProgram); LOAD(glGetProgramiv); LOAD(glGetProgramInfoLog);
    LOAD(glDeleteShader); LOAD(glDeleteProgram); LOAD(glUseProgram);
    LOAD(glCreateBuffers); LOAD(glDeleteBuffers); LOAD(glNamedBufferData); LOAD(glNamedBufferSubData); LOAD(glGetNamedBufferSubData);
    LOAD(glCopyNamedBufferSubData); LOAD(glBindBufferBase); LOAD(glBindBuffer);
    LOAD(glCreateTextures); LOAD(glDeleteTextures); LOAD(glTextureStorage2D); LOAD(glTextureSubImage2D); LOAD(glTextureSubImage3D);
    LOAD(glTextureParameteri); LOAD(glGetTextureImage); LOAD(glBindTextureUnit); LOAD(glBindImageTexture);
    LOAD(glDispatchCompute); LOAD(glDispatchComputeIndirect); LOAD(glMemoryBarrier); LOAD(glFinish);
    LOAD(glGetUniformLocation); LOAD(glProgramUniform1i); LOAD(glProgramUniform1ui); LOAD(glPixelStorei); LOAD(glEnable);
    snprintf(g_info, sizeof g_info, "%s | %s", (const char *)p_glGetString(GL_VERSION), (const char *)p_glGetString(GL_RENDERER));
    p_glPixelStorei(GL_UNPACK_ALIGNMENT, 1); p_glPixelStorei(GL_PACK_ALIGNMENT, 1);
    p_glEnable(GL_TEXTURE_CUBE_MAP_SEAMLESS);        /* Texture.TryEnableSeamlessCubemap (Render/SkyBoxManager.cs:74) */
    g_ready = 1;
    return 0;
}

const char *glref_info(void) { return g_info; }
int glref_error(void) { return (int)p_glGetError(); }
int glref_get_integer(unsigned pname) { GLint v = 0; p_glGetIntegerv(pname, &v); return v; }
int glref_get_integer_i(unsigned pname, unsigned i) { GLint v = 0; p_glGetIntegeri_v(pname, i, &v); return v; }

int glref_has_extension(const char *name)
{
    GLint n = 0; p_glGetIntegerv(GL_NUM_EXTENSIONS, &n);
    for (GLint i = 0; i < n; i++) if (!strcmp((const char *)p_glGetStringi(GL_EXTENSIONS, (GLuint)i), name)) return 1;
    return 0;
}

/* returns the program name, or 0 with the compile/link log in `log` */
unsigned glref_compile_compute(const char *src, char *log, int logcap)
{
    if (log && logcap > 0) log[0] = 0;
    GLuint sh = p_glCreateShader(GL_COMPUTE_SHADER);
    p_glShaderSource(sh, 1, &src, NULL);
    p_glCompileShader(sh);
    GLint ok = 0; p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) { if (log) p_glGetShaderInfoLog(sh, logcap, NULL, log); p_glDeleteShader(sh); return 0; }
    GLuint prog = p_glCreateProgram();
    p_glAttachShader(prog, sh);
    p_glLinkProgram(prog);
    p_glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    p_glDeleteShader(sh);
    if (!ok) { if (log) p_glGetProgramInfoLog(prog, logcap, NULL, log); p_glDeleteProgram(prog); return 0; }
    return prog;
}
void glref_delete_program(unsigned prog) { p_glDeleteProgram(prog); }
int glref_set_uniform_1i(unsigned prog, const char *name, int v)
{
    GLint loc = p_glGetUniformLocation(prog, name);
    if (loc < 0) return -1;
    p_glProgramUniform1i(prog, loc, v);
    return 0;
}

int glref_set_uniform_1ui(unsigned prog, const char *name, unsigned v)
{
    GLint loc = p_glGetUniformLocation(prog, name);
    if (loc < 0) return -1;
    p_glProgramUniform1ui(prog, loc, v);
    return 0;
}

unsigned glref_buffer(const void *data, size_t size)
{
    GLuint b = 0; p_glCreateBuffers(1, &b);
    p_glNamedBufferData(b, (GLsizeiptr)(size ? size : 4), NULL, GL_DYNAMIC_COPY);
    if (data && size) p_glNamedBufferSubData(b, 0, (GLsizeiptr)size, data);
    return b;
}
void glref_delete_buffer(unsigned b) { p_glDeleteBuffers(1, &b); }
void glref_buffer_write(unsigned b, size_t off, size_t size, const void *data) { p_glNamedBufferSubData(b, (GLintptr)off, (GLsizeiptr)size, data); }
void glref_buffer_read(unsigned b, size_t off, size_t size, void *out) { p_glFinish(); p_glGetNamedBufferSubData(b, (GLintptr)off, (GLsizeiptr)size, out); }
void glref_buffer_copy(unsigned src, unsigned dst, size_t soff, size_t doff, size_t size) { p_glCopyNamedBufferSubData(src, dst, (GLintptr)soff, (GLintptr)doff, (GLsizeiptr)size); }
void glref_bind_ssbo(unsigned binding, unsigned b) { p_glBindBufferBase(GL_SHADER_STORAGE_BUFFER, binding, b); }
void glref_bind_ubo(unsigned binding, unsigned b) { p_glBindBufferBase(GL_UNIFORM_BUFFER, binding, b); }

/* RGBA32F 2D texture, one level; linear != 0 -> GL_LINEAR min/mag, repeat != 0 -> GL_REPEAT else CLAMP_TO_EDGE */
unsigned glref_texture2d(int w, int h, const float *rgba, int linear, int repeat)
{
    GLuint t = 0; p_glCreateTextures(GL_TEXTURE_2D, 1, &t);
    p_glTextureStorage2D(t, 1, GL_RGBA32F, w, h);
    if (rgba) p_glTextureSubImage2D(t, 0, 0, 0, w, h, GL_RGBA, GL_FLOAT, rgba);
    p_glTextureParameteri(t, GL_TEXTURE_MIN_FILTER, linear ? GL_LINEAR : GL_NEAREST);
    p_glTextureParameteri(t, GL_TEXTURE_MAG_FILTER, linear ? GL_LINEAR : GL_NEAREST);
    p_glTextureParameteri(t, GL_TEXTURE_WRAP_S, repeat ? GL_REPEAT : GL_CLAMP_TO_EDGE);
    p_glTextureParameteri(t, GL_TEXTURE_WRAP_T, repeat ? GL_REPEAT : GL_CLAMP_TO_EDGE);
    return t;
}
/* One image of the scene's texture table with the sampler state of its glTF sampler (ModelLoader.GetGLSamplerState, Utils/ModelLoader.cs:1166-1197): format 0 = GL_RGBA32F
 * (float texels), 1 = GL_RGBA8, 2 = GL_SRGB8_ALPHA8 (byte texels); wrapS / wrapT 0 = GL_REPEAT, 1 = GL_CLAMP_TO_EDGE, 2 = GL_MIRRORED_REPEAT; nearest != 0 -> GL_NEAREST.
 * One level (the compute shaders sample level 0: no derivatives), so the minification filter is the magnification filter and the texture is complete. */
unsigned glref_texture2d_state(int w, int h, const void *texels, int format, int wrapS, int wrapT, int nearest)
{
    static const GLenum wraps[3] = {GL_REPEAT, GL_CLAMP_TO_EDGE, GL_MIRRORED_REPEAT};
    GLuint t = 0; p_glCreateTextures(GL_TEXTURE_2D, 1, &t);
    p_glTextureStorage2D(t, 1, format == 0 ? GL_RGBA32F : (format == 1 ? GL_RGBA8 : GL_SRGB8_ALPHA8), w, h);
    if (texels) p_glTextureSubImage2D(t, 0, 0, 0, w, h, GL_RGBA, format == 0 ? GL_FLOAT : GL_UNSIGNED_BYTE, texels);
    p_glTextureParameteri(t, GL_TEXTURE_MIN_FILTER, nearest ? GL_NEAREST : GL_LINEAR);
    p_glTextureParameteri(t, GL_TEXTURE_MAG_FILTER, nearest ? GL_NEAREST : GL_LINEAR);
    p_glTextureParameteri(t, GL_TEXTURE_WRAP_S, wraps[wrapS]);
    p_glTextureParameteri(t, GL_TEXTURE_WRAP_T, wraps[wrapT]);
    return t;
}
/* RGBA32F cube map, faces in GL order (+X,-X,+Y,-Y,+Z,-Z), s x s texels each */
unsigned glref_cubemap(int s, const float *rgba6, int linear)
{
    GLuint t = 0; p_glCreateTextures(GL_TEXTURE_CUBE_MAP, 1, &t);
    p_glTextureStorage2D(t, 1, GL_RGBA32F, s, s);
    if (rgba6) p_glTextureSubImage3D(t, 0, 0, 0, 0, s, s, 6, GL_RGBA, GL_FLOAT, rgba6);
    p_glTextureParameteri(t, GL_TEXTURE_MIN_FILTER, linear ? GL_LINEAR : GL_NEAREST);
    p_glTextureParameteri(t, GL_TEXTURE_MAG_FILTER, linear ? GL_LINEAR : GL_NEAREST);
    p_glTextureParameteri(t, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    p_glTextureParameteri(t, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    p_glTextureParameteri(t, GL_TEXTURE_WRAP_R, GL_CLAMP_TO_EDGE);
    return t;
}
void glref_delete_texture(unsigned t) { p_glDeleteTextures(1, &t); }
void glref_texture_write(unsigned t, int w, int h, const float *rgba) { p_glTextureSubImage2D(t, 0, 0, 0, w, h, GL_RGBA, GL_FLOAT, rgba); }
void glref_texture_read(unsigned t, int w, int h, float *rgba) { p_glFinish(); p_glGetTextureImage(t, 0, GL_RGBA, GL_FLOAT, (GLsizei)((size_t)w * h * 16), rgba); }
void glref_bind_texture(unsigned unit, unsigned t) { p_glBindTextureUnit(unit, t); }
void glref_bind_image(unsigned unit, unsigned t) { p_glBindImageTexture(unit, t, 0, GL_FALSE, 0, GL_READ_WRITE, GL_RGBA32F); }

void glref_dispatch(unsigned prog, unsigned x, unsigned y, unsigned z) { p_glUseProgram(prog); p_glDispatchCompute(x, y, z); }
void glref_dispatch_indirect(unsigned prog, unsigned buf, size_t off)
{
    p_glUseProgram(prog);
    p_glBindBuffer(GL_DISPATCH_INDIRECT_BUFFER, buf);
    p_glDispatchComputeIndirect((GLintptr)off);
}
void glref_barrier(void) { p_glMemoryBarrier(GL_ALL_BARRIER_BITS); }
void glref_finish(void) { p_glFinish(); }

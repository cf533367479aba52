/* Plain-C host for libidkpt.so: proves the boundary is usable with nothing but include/idkpt.h and the shared object
 * (no Python, no torch, no C++).  Reads a scene dumped as raw arrays by tests/test_gpu_parity.py::test_plain_c_host_matches_python_host,
 * renders it through the C-ABI exactly like the reference's PathTracer would be driven (ctor -> uploads -> Compute -> read-back,
 * Source/Render/PathTracer.cs:170-271) and writes the Result image + stats for the test to compare bit for bit.
 *
 *   abi_driver <dir> <width> <height> <rayDepth> <samples> <useTlas> [devices]
 * devices > 1: ONE context on that many devices (idkptCreate(deviceCount = N)); ids wrap around the visible GPUs, so on a one-GPU box the
 * members share the GPU and the very same multi-device code runs.  The frame that comes back is the whole frame either way.
 * <dir>/manifest.txt: "<name> <count>" per line; <dir>/<name>.bin: raw bytes of that array.
 * Build (tests do this): gcc -std=c11 -O1 -I include tests/c_driver/abi_driver.c -L idkengine_amd -lidkpt -Wl,-rpath,idkengine_amd -o abi_driver
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "idkpt.h"

static void* load(const char* dir, const char* name, size_t* bytes)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s.bin", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { *bytes = 0; return NULL; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    void* p = malloc(n > 0 ? (size_t)n : 1);
    if (n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

struct cb_state { int calls; int32_t status; char message[512]; };
static void on_error(void* user, int32_t status, const char* message)
{
    struct cb_state* s = (struct cb_state*)user;
    s->calls++; s->status = status;
    strncpy(s->message, message ? message : "", sizeof s->message - 1);
}

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ != IDKPT_OK) { const char* m_ = ""; idkptGetLastError(ctx, &m_); \
    fprintf(stderr, "%s failed: %d (%s)\n", #call, (int)rc_, m_ ? m_ : ""); return 3; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: abi_driver dir w h depth samples useTlas\n"); return 1; }
    const char* dir = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]), depth = atoi(argv[4]), samples = atoi(argv[5]), useTlas = atoi(argv[6]);
    size_t n;
    idkpt_scene_desc sc; memset(&sc, 0, sizeof sc);
    sc.BlasNodes = (const GpuBlasNode*)load(dir, "blas_nodes", &n); sc.BlasNodeCount = (int32_t)(n / sizeof(GpuBlasNode));
    sc.BlasTriangles = (const GpuBlasTriangle*)load(dir, "blas_triangles", &n); sc.BlasTriangleCount = (int32_t)(n / sizeof(GpuBlasTriangle));
    sc.BlasDescs = (const GpuBlasDesc*)load(dir, "blas_descs", &n); sc.BlasDescCount = (int32_t)(n / sizeof(GpuBlasDesc));
    sc.BlasInstances = (const GpuBlasInstance*)load(dir, "blas_instances", &n); sc.BlasInstanceCount = (int32_t)(n / sizeof(GpuBlasInstance));
    sc.TlasNodes = (const GpuTlasNode*)load(dir, "tlas_nodes", &n); sc.TlasNodeCount = (int32_t)(n / sizeof(GpuTlasNode));
    sc.VertexPositions = (const float*)load(dir, "vertex_positions", &n); sc.VertexCount = (int32_t)(n / 12);
    sc.Vertices = (const GpuVertex*)load(dir, "vertices", &n);
    sc.Meshes = (const GpuMesh*)load(dir, "meshes", &n); sc.MeshCount = (int32_t)(n / sizeof(GpuMesh));
    sc.Materials = (const GpuMaterial*)load(dir, "materials", &n); sc.MaterialCount = (int32_t)(n / sizeof(GpuMaterial));
    sc.MeshTransforms = (const GpuMeshTransform*)load(dir, "mesh_transforms", &n); sc.MeshTransformCount = (int32_t)(n / sizeof(GpuMeshTransform));
    sc.Lights = (const GpuLight*)load(dir, "lights", &n); sc.LightCount = (int32_t)(n / sizeof(GpuLight));
    if (sc.LightCount == 0) sc.Lights = NULL;
    sc.SkyFaces = (const float*)load(dir, "sky_faces", &n); sc.SkyFaceSize = n ? 1 : 0;      /* the dump uses a constant (1x1 per face) sky */
    if (!n) sc.SkyFaces = NULL;
    float* cam = (float*)load(dir, "camera", &n);                                             /* invProj[16] invView[16] viewPos[3] */
    if (n != 35 * sizeof(float)) { fprintf(stderr, "bad camera dump\n"); return 2; }

    idkpt_ctx* ctx = NULL;
    int32_t devCount = 0;
    if (idkptGetDeviceCount(&devCount) != IDKPT_OK || devCount < 1) { fprintf(stderr, "no HIP device\n"); return 4; }
    const int devices = argc > 7 ? atoi(argv[7]) : 1;
    int32_t ids[64];
    if (devices < 1 || devices > 64) { fprintf(stderr, "bad device count\n"); return 1; }
    for (int d = 0; d < devices; d++) ids[d] = d % devCount;
    if (idkptCreate(devices, ids, &ctx) != IDKPT_OK || !ctx) { fprintf(stderr, "idkptCreate failed\n"); return 4; }
    CHECK(idkptSetSize(ctx, w, h));
    idkpt_settings st;
    CHECK(idkptGetSettings(ctx, &st));
    st.RayDepth = depth; st.SamplesPerPixel = samples; st.UseTlas = useTlas;
    CHECK(idkptSetSettings(ctx, &st));
    CHECK(idkptUploadScene(ctx, &sc));
    CHECK(idkptSetPerFrame(ctx, cam, cam + 16, cam + 32));
    CHECK(idkptEnableCounters(ctx, 1));
    CHECK(idkptResetAccumulation(ctx));
    CHECK(idkptRender(ctx));
    const size_t bytes = (size_t)w * h * 4 * sizeof(float);
    float* img = (float*)malloc(bytes);
    CHECK(idkptDownload(ctx, IDKPT_IMAGE_RESULT, img, bytes));
    idkpt_stats stats;
    CHECK(idkptGetStats(ctx, &stats));
    uint32_t acc = 0;
    CHECK(idkptGetAccumulatedSamples(ctx, &acc));
    /* an error path, from C: bad size must be reported through the status + last-error string, never abort */
    if (idkptSetSize(ctx, -1, 5) == IDKPT_OK) { fprintf(stderr, "expected an error for a negative size\n"); return 5; }
    const char* msg = NULL; idkptGetLastError(ctx, &msg);
    if (!msg || !msg[0]) { fprintf(stderr, "expected a last-error message\n"); return 5; }
    /* the optional error callback (oidnSetDeviceErrorFunction's pattern): same status, same message, before the failing call returns */
    struct cb_state cb = {0, 0, {0}};
    CHECK(idkptSetErrorCallback(ctx, on_error, &cb));
    const int32_t rcBad = idkptSetSize(ctx, 7, -3);
    if (rcBad == IDKPT_OK || cb.calls != 1 || cb.status != rcBad) { fprintf(stderr, "error callback: calls %d status %d (call returned %d)\n", cb.calls, cb.status, rcBad); return 6; }
    idkptGetLastError(ctx, &msg);
    if (!msg || strcmp(msg, cb.message) != 0) { fprintf(stderr, "error callback message differs from idkptGetLastError\n"); return 6; }
    CHECK(idkptSetErrorCallback(ctx, NULL, NULL));
    if (idkptSetSize(ctx, 7, -3) == IDKPT_OK || cb.calls != 1) { fprintf(stderr, "removed error callback was still called\n"); return 6; }

    char path[1024];
    snprintf(path, sizeof path, "%s/result.bin", dir);
    FILE* f = fopen(path, "wb"); fwrite(img, 1, bytes, f); fclose(f);
    snprintf(path, sizeof path, "%s/stats.txt", dir);
    f = fopen(path, "w");
    fprintf(f, "%llu %llu %llu %u\n", (unsigned long long)stats.RaysTraced, (unsigned long long)stats.NodePairVisits, (unsigned long long)stats.TriangleTests, acc);
    fclose(f);
    CHECK(idkptDestroy(ctx));
    printf("ok %s\n", idkptGetVersionString());
    return 0;
}

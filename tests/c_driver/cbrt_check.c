/* cbrt_check.c — CPU twin of the device cbrtf behind the GPU PreSplit (idkengine_amd/csrc/bvh_gpu_full.hpp: dev_cbrtf = glibc 2.35's algorithm,
 * written with explicit bit operations instead of frexpf / ldexpf), and the host's own cbrtf, over arrays.  tests/test_builder.py compares the twin
 * with the host on a stride through all 2^32 inputs (exhaustive: IDKPT_CBRT_EXHAUSTIVE=1); tests/test_gpu_builder.py compares the device with the host. */
#include <math.h>
#include <stdint.h>
#include <string.h>
static float twin_cbrtf(float x)
{
    uint32_t ix; memcpy(&ix, &x, 4);
    uint32_t ax = ix & 0x7fffffffu;
    if (ax == 0u || ax >= 0x7f800000u) return x + x;
    int xe; uint32_t m = ax;
    if (m < 0x00800000u) { int sh = __builtin_clz(m) - 8; m <<= sh; xe = (1 - sh) - 126; } else xe = (int)(m >> 23) - 126;
    uint32_t mb = (m & 0x007fffffu) | 0x3f000000u; float xm; memcpy(&xm, &mb, 4);
    float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
    float t2 = u * u * u;
    int r = xe % 3;
    double f = r == -2 ? 1.0 / 1.5874010519681994748 : r == -1 ? 1.0 / 1.2599210498948731648 : r == 0 ? 1.0 : r == 1 ? 1.2599210498948731648 : 1.5874010519681994748;
    float ym = (float)((double)u * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm) * f);
    uint32_t sb = (uint32_t)(127 + xe / 3) << 23; float s; memcpy(&s, &sb, 4);
    return (x > 0.0f ? ym : -ym) * s;
}
void host_cbrtf_array(const float* in, float* out, long n) { for (long i = 0; i < n; i++) out[i] = cbrtf(in[i]); }
/* number of bit patterns first, first + stride, ... (count of them) on which the twin and the host's cbrtf differ (NaN payloads ignored) */
long twin_vs_host_mismatches(uint32_t first, uint32_t stride, long count, uint32_t* firstBad)
{
    long bad = 0; uint32_t b = first;
    for (long i = 0; i < count; i++, b += stride) {
        float x; memcpy(&x, &b, 4);
        float a = twin_cbrtf(x), c = cbrtf(x);
        uint32_t ab, cb; memcpy(&ab, &a, 4); memcpy(&cb, &c, 4);
        if (ab != cb && !(a != a && c != c)) { if (!bad && firstBad) *firstBad = b; bad++; }
    }
    return bad;
}

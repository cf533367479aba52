/* Independent checker for tests/test_metamorphic.py: closest hit of every ray against EVERY triangle (no BVH), Moeller-Trumbore in
 * binary64.  Shares no code with the oracle or the HIP path.  gcc -O2 -fopenmp -shared -fPIC brute_force.c -o libbrute.so */
#include <math.h>
#include <stdint.h>
void brute_force(const double* tris /* n x 9 */, int64_t n, const double* org, const double* dir, int64_t rays, double* tBest, int64_t* iBest, double* tSecond)
{
    #pragma omp parallel for schedule(dynamic, 16)
    for (int64_t r = 0; r < rays; r++) {
        const double ox = org[3 * r], oy = org[3 * r + 1], oz = org[3 * r + 2], dx = dir[3 * r], dy = dir[3 * r + 1], dz = dir[3 * r + 2];
        double best = INFINITY, second = INFINITY; int64_t bi = -1;
        for (int64_t i = 0; i < n; i++) {
            const double* t = tris + 9 * i;
            const double e1x = t[3] - t[0], e1y = t[4] - t[1], e1z = t[5] - t[2], e2x = t[6] - t[0], e2y = t[7] - t[1], e2z = t[8] - t[2];
            const double px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
            const double det = e1x * px + e1y * py + e1z * pz;
            if (det == 0.0) continue;
            const double inv = 1.0 / det, tx = ox - t[0], ty = oy - t[1], tz = oz - t[2];
            const double u = (tx * px + ty * py + tz * pz) * inv;
            if (u < 0.0 || u > 1.0) continue;
            const double qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
            const double v = (dx * qx + dy * qy + dz * qz) * inv;
            if (v < 0.0 || u + v > 1.0) continue;
            const double tt = (e2x * qx + e2y * qy + e2z * qz) * inv;
            if (tt < 0.0) continue;
            if (tt < best) { second = best; best = tt; bi = i; } else if (tt < second) second = tt;
        }
        tBest[r] = best; iBest[r] = bi; tSecond[r] = second;
    }
}

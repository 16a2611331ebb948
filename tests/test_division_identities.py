"""The integrate kernel replaces IEEE float divisions by shorter exactly-rounded sequences (tsdf_common.hpp: div_known,
div_shared).  Their equality with `a / b` is a property of IEEE arithmetic, so it is checked here on the CPU in C
(gcc -ffp-contract=off -mfma): exhaustively over every float in [2^-40, 2^12] for the known divisors, and on 2e8 random
(a, b) pairs -- with the reciprocal estimate perturbed by +-1 ulp, the accuracy v_rcp_f32 guarantees -- for the shared one."""
import os
import subprocess
import sys
import tempfile

SRC = r"""
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static inline float div_known(float a, float b, float y) { float q = a * y; float r = fmaf(-b, q, a); return fmaf(r, y, q); }
static inline float div_shared(float a, float b, float y0) {
    float e = fmaf(-b, y0, 1.0f); float y1 = fmaf(y0, e, y0);
    float q0 = a * y1; float r0 = fmaf(-b, q0, a); float q1 = fmaf(r0, y1, q0);
    float r1 = fmaf(-b, q1, a); return fmaf(r1, y1, q1);
}
static uint64_t s = 88172645463325252ull;
static inline uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
int main() {
    uint64_t bad = 0;
    const float bs[4] = {255.0f, 32767.0f, 0.02f, 0.04f};
    for (int k = 0; k < 4; k++) {
        const float b = bs[k], y = 1.0f / b;
        uint32_t lo, hi; float flo = ldexpf(1.0f, -40), fhi = ldexpf(1.0f, 12);
        memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
        for (uint32_t u = lo; u <= hi; u++) {
            float a; memcpy(&a, &u, 4);
            if (a / b != div_known(a, b, y) || (-a) / b != div_known(-a, b, y)) bad++;
        }
    }
    printf("known %llu\n", (unsigned long long)bad);
    bad = 0;
    for (int bi = 1; bi <= 256; bi++)
        for (int d = -1; d <= 1; d++) {
            float b = (float)bi, y0 = 1.0f / b; uint32_t u; memcpy(&u, &y0, 4); u += d; memcpy(&y0, &u, 4);
            for (int k = 0; k < 100000; k++) {
                uint32_t r = rnd(); float a = ldexpf((float)(r & 0xFFFFFF) / 16777216.0f + 1.0f, (int)((r >> 24) % 40) - 30);
                if (r & 0x80000000u) a = -a;
                if (a / b != div_shared(a, b, y0)) bad++;
            }
        }
    for (int64_t k = 0; k < 120000000ll; k++) {
        uint32_t r1 = rnd(), r2 = rnd();
        float b = ldexpf((float)(r1 & 0xFFFFFF) / 16777216.0f + 1.0f, (int)((r1 >> 24) % 24) - 13);   /* [1e-4, 2e3] */
        float a = ldexpf((float)(r2 & 0xFFFFFF) / 16777216.0f + 1.0f, (int)((r2 >> 24) % 40) - 20);
        if (r2 & 0x80000000u) a = -a;
        float y0 = 1.0f / b; uint32_t u; memcpy(&u, &y0, 4); u += (int)(k % 3) - 1; memcpy(&y0, &u, 4);
        if (a / b != div_shared(a, b, y0)) bad++;
    }
    printf("shared %llu\n", (unsigned long long)bad);
    return 0;
}
"""


def test_short_division_sequences_equal_ieee_division():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write(SRC)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", exe, src, "-lm"])
        out = subprocess.check_output([exe], timeout=600).decode()
    assert "known 0" in out and "shared 0" in out, out

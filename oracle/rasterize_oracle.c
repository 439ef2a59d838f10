/* TEST INFRASTRUCTURE — the parity oracle for the 3DMM triangle rasterizer.
 *
 * A plain-C restatement of the reference's *sequential CPU* rasterizer, the path
 * BASELINE.json's north_star names as the bit-exactness target:
 *   forward   reference op/rasterize.cpp:21-67  + op/rasterize.h:10-167
 *   backward  reference op/rasterize.cpp:69-95  + op/rasterize.h:169-228
 *
 * PINNED against the reference itself: tests/test_oracle_rasterize.py compares this file
 * (a) with the golden vectors in tests/golden/raster_*.npz, which oracle/make_golden.py
 *     generated from the reference's own op/rasterize.cpp compiled where it lies
 *     (oracle/build_ref.py -> oracle/_ref/rasterize_ref.so), including the reference's only
 *     known-answer test (op/rasterize.py:83-107), and
 * (b) directly with oracle/_ref/rasterize_ref.so on random meshes when that file is present.
 * Both comparisons are bitwise (index, coeff, z-buffer, dcoeff).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (stylerenderer_amd/) never does.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (no -march: x86-64 baseline has no FMA),
 * see oracle/build_oracle.py.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define REAL float
#define NAME(x) x##_f32
#define CEIL ceilf
#define FLOOR floorf
#include "rasterize_oracle_impl.h"
#undef REAL
#undef NAME
#undef CEIL
#undef FLOOR

#define REAL double
#define NAME(x) x##_f64
#define CEIL ceil
#define FLOOR floor
#include "rasterize_oracle_impl.h"
#undef REAL
#undef NAME
#undef CEIL
#undef FLOOR

/* Attribute interpolation of the Python wrapper, reference op/rasterize.py:29-37:
 * out[p, ch] = sum_k tex[index[p, k], ch] * coeff[p, k]  (k = 0, 1, 2 accumulated in order). */
void oracle_rasterize_interp_f32(int64_t npix, int64_t c, const float *tex, const int64_t *index,
                                 const float *coeff, float *out) {
    for (int64_t p = 0; p < npix; ++p)
        for (int64_t ch = 0; ch < c; ++ch) {
            float a0 = tex[index[3 * p] * c + ch] * coeff[3 * p];
            float a1 = tex[index[3 * p + 1] * c + ch] * coeff[3 * p + 1];
            float a2 = tex[index[3 * p + 2] * c + ch] * coeff[3 * p + 2];
            float s = a0 + a1;
            out[p * c + ch] = s + a2;
        }
}

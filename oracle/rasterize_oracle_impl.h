/* TEST INFRASTRUCTURE — not part of the product.  Included twice by rasterize_oracle.c
 * (REAL = float / double).  See that file for the header comment.
 *
 * Every arithmetic statement is written so that a C compiler with -ffp-contract=off
 * evaluates exactly the rounding sequence of the reference's templates:
 *   setup            reference op/rasterize.h:10-75   (barycentric)
 *   weights          reference op/rasterize.h:77-124  (normalize_coeff)
 *   depth test       reference op/rasterize.h:126-167 (assign_buffer, host branch)
 *   gradient         reference op/rasterize.h:169-228 (barycentric_grad)
 *   loops            reference op/rasterize.cpp:21-67, 69-95
 */

typedef struct {
    REAL p[9];      /* vertices: x,y in screen space after setup, z untouched          */
    REAL e[9];      /* e[0..2] = edge constants, e[3..5] = d/dx, e[6..8] = d/dy         */
    REAL area;      /* |signed area sum| as stored by the reference through det_       */
    int64_t x0, x1, y0, y1;
} NAME(tri_t);

/* (int64_t) of a floating value as x86-64 cvttss2si/cvttsd2si defines it: out-of-range and
 * NaN give INT64_MIN.  The reference does a plain C cast (op/rasterize.h:40-43). */
static int64_t NAME(to_i64)(REAL f) {
    if (f >= (REAL)-9223372036854775808.0 && f < (REAL)9223372036854775808.0) return (int64_t)f;
    return INT64_MIN;
}

/* Returns 0 when the triangle is rejected.  `sw`/`sh` are the extents the reference's
 * `barycentric` sees as (w, h); the callers pass (h_arg, w_arg), reproducing the swap at
 * op/rasterize.cpp:38. */
static int NAME(tri_setup)(NAME(tri_t) * t, int64_t sw, int64_t sh, int perspective, REAL eps) {
    REAL lo_u = (REAL)sw, lo_v = (REAL)sh, hi_u = 0, hi_v = 0;
    for (int k = 0; k < 3; ++k) {
        REAL *q = t->p + 3 * k;
        if (perspective) {
            if (q[2] >= -eps) return 0;
            q[0] = q[0] / -q[2];
            q[1] = q[1] / -q[2];
        }
        /* "(1 + x) * w / 2 - .5": the .5 is a double literal, so the subtraction happens in
         * double and is rounded once on assignment (op/rasterize.h:21-22). */
        REAL sx = (1 + q[0]) * (REAL)sw / 2;
        REAL sy = (1 - q[1]) * (REAL)sh / 2;
        q[0] = (REAL)((double)sx - .5);
        q[1] = (REAL)((double)sy - .5);
        if (k == 0) {
            lo_u = hi_u = q[0];
            lo_v = hi_v = q[1];
        } else {
            if (lo_u > q[0]) lo_u = q[0];
            else if (hi_u < q[0]) hi_u = q[0];
            if (lo_v > q[1]) lo_v = q[1];
            else if (hi_v < q[1]) hi_v = q[1];
        }
    }
    t->x0 = NAME(to_i64)(CEIL(lo_u));
    t->x1 = NAME(to_i64)(FLOOR(hi_u));
    t->y0 = NAME(to_i64)(CEIL(lo_v));
    t->y1 = NAME(to_i64)(FLOOR(hi_v));
    if (t->x0 < 0) t->x0 = 0;
    if (t->x1 > sw - 1) t->x1 = sw - 1;
    if (t->y0 < 0) t->y0 = 0;
    if (t->y1 > sh - 1) t->y1 = sh - 1;
    if (t->x1 < t->x0 || t->y1 < t->y0) return 0;

    const REAL *p = t->p;
    REAL *e = t->e;
    REAL m0 = p[3] * p[7], m1 = p[4] * p[6];
    e[0] = m0 - m1;
    m0 = p[1] * p[6]; m1 = p[0] * p[7];
    e[1] = m0 - m1;
    m0 = p[0] * p[4]; m1 = p[1] * p[3];
    e[2] = m0 - m1;
    REAL det = e[0] + e[1];
    det = det + e[2];
    if (det > eps) return 0;            /* back face (front faces are CCW in NDC) */
    e[3] = p[4] - p[7];
    e[4] = p[7] - p[1];
    e[5] = p[1] - p[4];
    e[6] = p[6] - p[3];
    e[7] = p[0] - p[6];
    e[8] = p[3] - p[0];
    if (det < 0) {
        for (int k = 0; k < 9; ++k) e[k] = -e[k];
        t->area = -det;
    } else {
        t->area = det;
    }
    return 1;
}

/* Barycentric weights of pixel (px, py); `c` holds the three unnormalised edge values on
 * entry.  Returns 0 when the pixel is outside. */
static int NAME(pixel_weights)(const NAME(tri_t) * t, REAL px, REAL py, REAL c[3], REAL eps) {
    if (c[0] < -eps || c[1] < -eps || c[2] < -eps) return 0;
    if (t->area > eps) {
        REAL s = c[0] + c[1];
        s = s + c[2];
        c[0] = c[0] / s;
        c[1] = c[1] / s;
        c[2] = c[2] / s;
        return 1;
    }
    /* zero-area triangle: fall back to its longest edge, or to a point */
    const REAL *e = t->e, *p = t->p;
    REAL len[3];
    for (int k = 0; k < 3; ++k) {
        REAL a = e[3 + k] * e[3 + k], b = e[6 + k] * e[6 + k];
        len[k] = a + b;
    }
    int i = (len[0] > len[1]) ? 0 : 1;
    i = (len[i] > len[2]) ? i : 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    if (len[i] > eps) {
        REAL a = -(px - p[3 * k]) * e[6 + i];
        REAL b = (py - p[3 * k + 1]) * e[3 + i];
        REAL lj = a + b;
        a = (px - p[3 * j]) * e[6 + i];
        b = (py - p[3 * j + 1]) * e[3 + i];
        REAL lk = a - b;
        REAL li = lj + lk;
        c[i] = 0;
        c[j] = lj / li;
        c[k] = lk / li;
        return c[j] >= -eps && c[k] >= -eps;
    }
    c[j] = c[k] = 0;
    c[i] = 1;
    REAL dx = px - p[3 * i], dy = py - p[3 * i + 1];
    REAL a = dx * dx, b = dy * dy;
    return (a + b) < eps;
}

/* Depth of the pixel from its weights (mutates c in perspective mode).  Returns 0 when the
 * perspective depth is rejected. */
static int NAME(pixel_depth)(const NAME(tri_t) * t, REAL c[3], int perspective, REAL eps, REAL *z) {
    const REAL *p = t->p;
    if (perspective) {
        c[0] = c[0] / p[2];
        c[1] = c[1] / p[5];
        c[2] = c[2] / p[8];
        REAL s = c[0] + c[1];
        s = s + c[2];
        if (s >= -eps) return 0;
        c[0] = c[0] * s;
        c[1] = c[1] * s;
        c[2] = c[2] * s;
        *z = s;
    } else {
        REAL a = c[0] * p[2], b = c[1] * p[5], d = c[2] * p[8];
        REAL s = a + b;
        *z = s + d;
    }
    return 1;
}

/* Sequential z-buffer rasterizer.  Buffers must arrive initialised: index 0, coeff 0,
 * zbuf -MAX (what op/rasterize.cpp:128-132 allocates).  Returns triangles visited. */
int64_t NAME(oracle_rasterize_forward)(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w,
                                       int repeat_v, int repeat_f, int perspective,
                                       const REAL *v, const int64_t *f, int64_t *index,
                                       REAL *coeff, REAL *zbuf, REAL eps) {
    int64_t visited = 0;
    if (!v || !f) return 0;
    for (int64_t s = 0; s < b; ++s) {
        const REAL *vs = repeat_v ? v : v + s * nv * 3;
        const int64_t *fs = repeat_f ? f : f + s * nf * 3;
        int64_t *is = index ? index + s * h * w * 3 : NULL;
        REAL *cs = coeff ? coeff + s * h * w * 3 : NULL;
        REAL *zs = zbuf ? zbuf + s * h * w : NULL;
        const int64_t shift = repeat_v ? 0 : nv * s;
        for (int64_t ti = 0; ti < nf; ++ti) {
            const int64_t a0 = fs[3 * ti], a1 = fs[3 * ti + 1], a2 = fs[3 * ti + 2];
            if (a0 < 0 || a1 < 0 || a2 < 0 || a0 >= nv || a1 >= nv || a2 >= nv) continue;
            NAME(tri_t) t;
            for (int k = 0; k < 3; ++k) {
                t.p[k] = vs[3 * a0 + k];
                t.p[3 + k] = vs[3 * a1 + k];
                t.p[6 + k] = vs[3 * a2 + k];
            }
            ++visited;
            if (!NAME(tri_setup)(&t, h, w, perspective, eps)) continue;
            for (int64_t y = t.y0; y <= t.y1; ++y)
                for (int64_t x = t.x0; x <= t.x1; ++x) {
                    const int64_t pix = x + y * w;
                    if (pix >= h * w) continue; /* reference would write out of bounds (w > h) */
                    const REAL px = (REAL)x, py = (REAL)y;
                    REAL c[3];
                    for (int k = 0; k < 3; ++k) {
                        REAL gx = t.e[3 + k] * px, gy = t.e[6 + k] * py;
                        REAL acc = t.e[k] + gx;
                        c[k] = acc + gy;
                    }
                    if (!NAME(pixel_weights)(&t, px, py, c, eps)) continue;
                    REAL z;
                    if (!NAME(pixel_depth)(&t, c, perspective, eps, &z)) continue;
                    if (zs) {
                        if (!(zs[pix] < z)) continue;
                        zs[pix] = z;
                    }
                    if (cs) {
                        cs[3 * pix] = c[0];
                        cs[3 * pix + 1] = c[1];
                        cs[3 * pix + 2] = c[2];
                    }
                    if (is) {
                        is[3 * pix] = a0 + shift;
                        is[3 * pix + 1] = a1 + shift;
                        is[3 * pix + 2] = a2 + shift;
                    }
                }
        }
    }
    return visited;
}

/* d(3 weights)/d(3 vertices x xyz) at one pixel; `g` has 27 entries laid out
 * [weight][vertex][component].  Returns 0 (g untouched) for a degenerate triangle. */
static int NAME(weight_jacobian)(const REAL p[9], REAL px, REAL py, REAL sw, REAL sh, REAL g[27],
                                 int perspective, REAL eps) {
    REAL u = (px * 2 - sw + 1) / sw;
    REAL vv = (py * -2 + sh - 1) / sh;
    REAL e[9], det;
    REAL m0 = p[3] * p[7], m1 = p[4] * p[6];
    e[0] = m0 - m1;
    m0 = p[1] * p[6]; m1 = p[0] * p[7];
    e[1] = m0 - m1;
    m0 = p[0] * p[4]; m1 = p[1] * p[3];
    e[2] = m0 - m1;
    if (perspective) {
        if (p[2] >= -eps || p[5] >= -eps || p[8] >= -eps) return 0;
        REAL a = e[0] * p[2], b = e[1] * p[5], d = e[2] * p[8];
        det = a + b;
        det = det + d;
        if (det >= -eps && det <= eps) return 0;
        e[0] = -e[0];
        e[1] = -e[1];
        e[2] = -e[2];
        m0 = p[4] * p[8]; m1 = p[5] * p[7]; e[3] = m0 - m1;
        m0 = p[2] * p[7]; m1 = p[1] * p[8]; e[4] = m0 - m1;
        m0 = p[1] * p[5]; m1 = p[2] * p[4]; e[5] = m0 - m1;
        m0 = p[5] * p[6]; m1 = p[3] * p[8]; e[6] = m0 - m1;
        m0 = p[0] * p[8]; m1 = p[2] * p[6]; e[7] = m0 - m1;
        m0 = p[2] * p[3]; m1 = p[0] * p[5]; e[8] = m0 - m1;
    } else {
        det = e[0] + e[1];
        det = det + e[2];
        e[3] = p[4] - p[7];
        e[4] = p[7] - p[1];
        e[5] = p[1] - p[4];
        e[6] = p[6] - p[3];
        e[7] = p[0] - p[6];
        e[8] = p[3] - p[0];
    }
    if (!(det < -eps || det > eps)) return 0;
    for (int k = 0; k < 9; ++k) e[k] = e[k] / det;
    REAL c[3];
    for (int k = 0; k < 3; ++k) {
        REAL gx = e[3 + k] * u, gy = e[6 + k] * vv;
        REAL acc = e[k] + gx;
        c[k] = acc + gy;
    }
    if (!g) return 1;
    for (int l = 0; l < 27; ++l) {
        const int wi = l / 9, comp = (l + 1) % 3, vert = (l / 3) % 3;
        g[l] = -c[vert] * e[wi + comp * 3];
    }
    if (perspective) {
        REAL s = c[0] + c[1];
        s = s + c[2];
        for (int l = 0; l < 9; ++l) {
            REAL tot = g[l] + g[l + 9];
            tot = tot + g[l + 18];
            for (int wi = 0; wi < 3; ++wi) {
                REAL corr = c[wi] * tot / s;
                if (l % 3 == 2) g[l + wi * 9] = (-g[l + wi * 9] - corr) / s;
                else g[l + wi * 9] = (g[l + wi * 9] - corr) / s;
            }
        }
    } else {
        for (int l = 0; l < 9; ++l) g[2 + l * 3] = 0;
    }
    return 1;
}

/* Pixel loop of the backward op.  `dcoeff` ([b,h,w,27]) must arrive zeroed.
 * Reference: op/rasterize.cpp:69-95 (note: `v` is NOT advanced per sample; the stored ids
 * already carry the nv*batch offset). */
int64_t NAME(oracle_rasterize_backward)(int64_t b, int64_t n, int64_t h, int64_t w, int repeat_v,
                                        int perspective, const REAL *v, const int64_t *index,
                                        REAL *dcoeff, REAL eps) {
    (void)repeat_v;
    int64_t visited = 0;
    if (!v || !index) return 0;
    for (int64_t s = 0; s < b; ++s) {
        const int64_t *is = index + s * h * w * 3;
        REAL *gs = dcoeff ? dcoeff + s * h * w * 27 : NULL;
        for (int64_t t = 0; t < h * w; ++t) {
            const int64_t a0 = is[3 * t], a1 = is[3 * t + 1], a2 = is[3 * t + 2];
            if (a0 == a1 || a0 == a2 || a1 == a2) continue;
            if (a0 < 0 || a1 < 0 || a2 < 0 || a0 >= n * b || a1 >= n * b || a2 >= n * b) continue;
            REAL p[9];
            for (int k = 0; k < 3; ++k) {
                p[k] = v[3 * a0 + k];
                p[3 + k] = v[3 * a1 + k];
                p[6 + k] = v[3 * a2 + k];
            }
            /* (scalar)h and (scalar)w land in the (w, h) slots: op/rasterize.cpp:87 */
            NAME(weight_jacobian)(p, (REAL)(t % w), (REAL)(t / w), (REAL)h, (REAL)w,
                                  gs ? gs + t * 27 : NULL, perspective, eps);
            ++visited;
        }
    }
    return visited;
}

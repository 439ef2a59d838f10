// TEST INFRASTRUCTURE — not part of the product.
//
// Translation unit that turns the reference's CPU rasterizer into a loadable
// library *from the sources where they lie* (/root/reference/op, passed with -I).
// Nothing of the reference is copied here: this file only #includes
//   rasterize.h    (header-only arithmetic, reference op/rasterize.h:10-228)
//   rasterize.cpp  (sequential loops + pybind module, reference op/rasterize.cpp:21-245)
// and adds what the reference's rasterize.cu normally supplies for the host build
// (explicit host instantiations, op/rasterize.cu:140-160) plus `return false`
// definitions for the two GPU entry points (op/rasterize.cpp:14-19), which are never
// reached for CPU tensors.
//
// Extra extern "C" entry points expose the internal z-buffer, which the pybind
// `forward` allocates and drops (op/rasterize.cpp:128,177).
//
// Built only in the authoring container by oracle/build_ref.py into oracle/_ref/.
#include <cstdint>
#include <cstddef>
#include <cmath>
using std::ceil;
using std::floor;

#include "rasterize.h"
#include "rasterize.cpp"

template <typename scalar, typename index>
bool rasterize_gpu(index, index, index, index, index, bool, bool, bool,
                   const scalar*, const index*, index*, scalar*, scalar*, scalar) {
    return false;
}
template <typename scalar, typename index>
bool rasterize_gpu_backward(index, index, index, index, bool, bool,
                            const scalar*, const index*, scalar*, scalar) {
    return false;
}

template bool barycentric<float>(float*, int64_t, int64_t, int64_t*, float*, float*, bool, float);
template bool barycentric<double>(double*, int64_t, int64_t, int64_t*, double*, double*, bool, double);
template bool barycentric_grad<float>(const float*, float*, float, float, float*, bool, float);
template bool barycentric_grad<double>(const double*, double*, double, double, double*, bool, double);
template bool assign_buffer<float>(const float*, float*, const float*, float*, float*, bool,
                                   const float*, float, float);
template bool assign_buffer<double>(const double*, double*, const double*, double*, double*, bool,
                                    const double*, double, double);
template bool rasterize_gpu<float, int64_t>(int64_t, int64_t, int64_t, int64_t, int64_t, bool, bool,
                                            bool, const float*, const int64_t*, int64_t*, float*,
                                            float*, float);
template bool rasterize_gpu<double, int64_t>(int64_t, int64_t, int64_t, int64_t, int64_t, bool, bool,
                                             bool, const double*, const int64_t*, int64_t*, double*,
                                             double*, double);
template bool rasterize_gpu_backward<float, int64_t>(int64_t, int64_t, int64_t, int64_t, bool, bool,
                                                     const float*, const int64_t*, float*, float);
template bool rasterize_gpu_backward<double, int64_t>(int64_t, int64_t, int64_t, int64_t, bool, bool,
                                                      const double*, const int64_t*, double*, double);

extern "C" {
// Raw-pointer access to the reference's sequential loops, z-buffer included.
// Buffers must be pre-initialised by the caller exactly as op/rasterize.cpp:128-132 does
// (index = 0, coeff = 0, zB = -FLT_MAX / -DBL_MAX).
int64_t ref_rasterize_cpu_f32(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w,
                              int repeat_v, int repeat_f, int perspective,
                              const float* v, const int64_t* f, int64_t* i, float* c, float* zB,
                              float eps) {
    return rasterize_cpu<float, int64_t>(b, nv, nf, h, w, repeat_v != 0, repeat_f != 0,
                                         perspective != 0, v, f, i, c, zB, eps);
}
int64_t ref_rasterize_cpu_f64(int64_t b, int64_t nv, int64_t nf, int64_t h, int64_t w,
                              int repeat_v, int repeat_f, int perspective,
                              const double* v, const int64_t* f, int64_t* i, double* c, double* zB,
                              double eps) {
    return rasterize_cpu<double, int64_t>(b, nv, nf, h, w, repeat_v != 0, repeat_f != 0,
                                          perspective != 0, v, f, i, c, zB, eps);
}
int64_t ref_rasterize_cpu_backward_f32(int64_t b, int64_t n, int64_t h, int64_t w, int repeat_v,
                                       int perspective, const float* v, const int64_t* i,
                                       float* dcoeff, float eps) {
    return rasterize_cpu_backward<float, int64_t>(b, n, h, w, repeat_v != 0, perspective != 0, v, i,
                                                  dcoeff, eps);
}
int64_t ref_rasterize_cpu_backward_f64(int64_t b, int64_t n, int64_t h, int64_t w, int repeat_v,
                                       int perspective, const double* v, const int64_t* i,
                                       double* dcoeff, double eps) {
    return rasterize_cpu_backward<double, int64_t>(b, n, h, w, repeat_v != 0, perspective != 0, v, i,
                                                   dcoeff, eps);
}
}

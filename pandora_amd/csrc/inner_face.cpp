// inner_face.cpp - pandora_amd.inner_cpp: a pybind11 face with the reference's OWN native-function signatures over the C ABI of
// libpandora_amd.so (SURVEY 8b "inner" boundary; north_star: "a thin pybind11 C-ABI").
//
// The reference's python plugins call three pybind11 modules at these sites:
//     matching_cost_cpp.compute_matching_costs(img_left, imgs_right, cv, disps, w, h)       matching_cost/census.py:140
//     matching_cost_cpp.reverse_cost_volume(left_cv, disp_min)                               state_machine.py:438-448 (fast cross-checking)
//     matching_cost_cpp.reverse_disp_range(left_min, left_max)                               state_machine.py:673
//     aggregation_cpp.cross_support(image, len_arms, intensity)                              aggregation/cbca.py:237-291
//     aggregation_cpp.cbca(input, cross_left, cross_right, range_col, range_col_right)       aggregation/cbca.py:158
//     refinement_cpp.loop_refinement(cv, disp, mask, d_min, d_max, subpixel, measure, method, cst_invalid, cst_stopped)
//                                                                                            refinement/refinement.py:104
//     refinement_cpp.loop_approximate_refinement(... the same arguments, a right map on the left volume ...)   refinement/refinement.py:146
//     refinement_cpp.vfit_refinement_method / quadratic_refinement_method(cost, disp, measure, cst_stopped)   vfit.py / quadratic.py
// This module exports the same names with the same argument meaning and the same array conventions (arguments by value with
// implicit forcecast - a wrong dtype or a strided slice is silently copied, census WRITES INTO AND RETURNS the cv it was given,
// everything else returns new arrays; matching_cost_cpp.pyi:20-63, aggregation_cpp.pyi:21-58, refinement_cpp.pyi:22-143), each a
// thin call into the C ABI: the arrays go to the GPU, a HIP kernel does the work, the result comes back.  A maintainer swaps one
// import per file and keeps the reference's python untouched.  No torch, no numpy arithmetic here.
//
// tests/test_gpu_inner_face.py calls this module and the reference's compiled modules (oracle/_ref) with the same numpy arguments
// and compares the returns bit for bit.
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/pandora_amd.h"

namespace py = pybind11;
using farr = py::array_t<float, py::array::c_style | py::array::forcecast>;
using sarr = py::array_t<int16_t, py::array::c_style | py::array::forcecast>;
using larr = py::array_t<int64_t, py::array::c_style | py::array::forcecast>;

namespace {

pmx_ctx* ctx = nullptr;  // the face's one context; destroyed with the module (the capsule in PYBIND11_MODULE below)

pmx_ctx* context() {
    if (!ctx) {
        const char* dev = std::getenv("PANDORA_AMD_DEVICE");
        ctx = pmx_create(dev ? std::atoi(dev) : 0);
        if (!ctx) throw std::runtime_error(std::string("pandora_amd.inner_cpp: no GPU context (there is no CPU fallback): ") + pmx_last_error());
    }
    return ctx;
}

void ok(int rc, const char* what) {
    if (rc != PMX_OK) throw std::runtime_error(std::string(what) + ": " + pmx_last_error());
}

struct cv_guard {
    pmx_ctx* ctx;
    pmx_cv* cv;
    ~cv_guard() { if (cv) pmx_cv_free(ctx, cv); }
};

// matching_cost/cpp/src/census.cpp:97-180
py::array compute_matching_costs(farr img_left, py::list imgs_right, farr cv, farr disps, size_t census_width, size_t census_height) {
    if (img_left.ndim() != 2 || cv.ndim() != 3 || disps.ndim() != 1 || disps.shape(0) < 1) throw std::invalid_argument("compute_matching_costs: img_left 2-D, cv 3-D, disps 1-D");
    if (census_width != census_height) throw std::invalid_argument("compute_matching_costs: the device kernels take square census windows");
    const int H = (int)img_left.shape(0), W = (int)img_left.shape(1), D = (int)cv.shape(2);
    const int subpix = (int)py::len(imgs_right);
    if (subpix < 1 || cv.shape(0) != H || cv.shape(1) != W) throw std::invalid_argument("compute_matching_costs: shapes of cv / images disagree");
    std::vector<farr> rights;
    for (py::handle h : imgs_right) rights.push_back(farr::ensure(h));
    if (!rights[0] || rights[0].ndim() != 2 || rights[0].shape(0) != H || rights[0].shape(1) != W) throw std::invalid_argument("compute_matching_costs: right image shape");
    pmx_ctx* ctx = context();
    ok(pmx_set_images(ctx, img_left.data(), rights[0].data(), H, W, subpix), "pmx_set_images");
    ok(pmx_set_masks(ctx, nullptr, nullptr, 0, 1), "pmx_set_masks");
    ok(pmx_set_disparity_grids(ctx, nullptr, nullptr), "pmx_set_disparity_grids");
    for (int k = 1; k < subpix; ++k) {  // the caller's own shifted images (whatever interpolation made them), one column shorter
        if (!rights[k] || rights[k].ndim() != 2 || rights[k].shape(0) != H || rights[k].shape(1) != W - 1) throw std::invalid_argument("compute_matching_costs: shifted right image shape");
        ok(pmx_set_shifted_right(ctx, k, rights[k].data()), "pmx_set_shifted_right");
    }
    const int d0 = (int)std::lround(disps.data()[0]);  // census.cpp:110: only the first disparity is read
    cv_guard g{ctx, pmx_cv_alloc(ctx, D, d0)};
    if (!g.cv) throw std::runtime_error(std::string("pmx_cv_alloc: ") + pmx_last_error());
    ok(pmx_census(ctx, g.cv, (int)census_width), "pmx_census");
    std::vector<float> tmp((size_t)H * W * D);
    ok(pmx_cv_download(ctx, g.cv, tmp.data()), "pmx_cv_download");
    // the reference writes the cells it can compute and leaves the others as the caller passed them (census.cpp:134-171)
    float* out = cv.mutable_data();
    for (size_t i = 0; i < tmp.size(); ++i)
        if (tmp[i] == tmp[i]) out[i] = tmp[i];
    return std::move(cv);
}

// matching_cost/cpp/src/matching_cost.cpp:26-56: the right volume by re-indexing, (i, j, d) -> (i, j + d + disp_min, D - 1 - d),
// NaN where that column is outside the image.  (The reference reads the cells, never the coordinates: the volume's own first
// disparity does not enter.)
py::array_t<float> reverse_cost_volume(farr left_cv, int disp_min) {
    if (left_cv.ndim() != 3) throw std::invalid_argument("reverse_cost_volume: left_cv 3-D (row, col, disp)");
    const int H = (int)left_cv.shape(0), W = (int)left_cv.shape(1), D = (int)left_cv.shape(2);
    py::array_t<float> out({(py::ssize_t)H, (py::ssize_t)W, (py::ssize_t)D});
    if (H == 0 || W == 0 || D == 0) return out;
    pmx_ctx* ctx = context();
    std::vector<float> zeros((size_t)H * W, 0.f);  // (the context takes its geometry from the resident pair)
    ok(pmx_set_images(ctx, zeros.data(), zeros.data(), H, W, 1), "pmx_set_images");
    cv_guard l{ctx, pmx_cv_alloc(ctx, D, 0)};
    if (!l.cv) throw std::runtime_error(std::string("pmx_cv_alloc: ") + pmx_last_error());
    ok(pmx_cv_upload(ctx, l.cv, left_cv.data()), "pmx_cv_upload");
    cv_guard r{ctx, pmx_reverse_cost_volume(ctx, l.cv, disp_min)};
    if (!r.cv) throw std::runtime_error(std::string("pmx_reverse_cost_volume: ") + pmx_last_error());
    ok(pmx_cv_download(ctx, r.cv, out.mutable_data()), "pmx_cv_download");
    return out;
}

// matching_cost/cpp/src/matching_cost.cpp:59-132: per-pixel right disparity ranges from the left ones (NaN where no left pixel
// reaches the column).  The device kernel wants a bracket of every (int)left_min / (int)left_max: one scan of the two maps here.
std::tuple<py::array_t<float>, py::array_t<float>> reverse_disp_range(farr left_min, farr left_max) {
    if (left_min.ndim() != 2 || left_max.ndim() != 2 || left_min.shape(0) != left_max.shape(0) || left_min.shape(1) != left_max.shape(1))
        throw std::invalid_argument("reverse_disp_range: two 2-D maps of one shape");
    const int H = (int)left_min.shape(0), W = (int)left_min.shape(1);
    py::array_t<float> rmin({(py::ssize_t)H, (py::ssize_t)W}), rmax({(py::ssize_t)H, (py::ssize_t)W});
    if (H == 0 || W == 0) return {rmin, rmax};
    const float *a = left_min.data(), *b = left_max.data();
    bool any = false;
    int lo = 0, hi = 0;
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        if (a[i] != a[i] || b[i] != b[i]) continue;  // matching_cost.cpp:92-96: NaN ranges are skipped
        // (a range that cannot reach any column contributes nothing: clamp instead of converting a huge float)
        const float fa = std::fmax(std::fmin(a[i], (float)W), -(float)W), fb = std::fmax(std::fmin(b[i], (float)W), -(float)W);
        const int ia = (int)fa, ib = (int)fb;
        if (!any) { lo = ia; hi = ib; any = true; }
        lo = ia < lo ? ia : lo;
        hi = ib > hi ? ib : hi;
    }
    if (hi < lo) hi = lo;
    ok(pmx_reverse_disp_range(context(), a, b, H, W, lo, hi, rmin.mutable_data(), rmax.mutable_data()), "pmx_reverse_disp_range");
    return {rmin, rmax};
}

// aggregation/cpp/src/aggregation.cpp:224-321
py::array_t<int16_t> cross_support(farr image, int16_t len_arms, float intensity) {
    if (image.ndim() != 2) throw std::invalid_argument("cross_support: 2-D image");
    const int H = (int)image.shape(0), W = (int)image.shape(1);
    py::array_t<int16_t> out({(py::ssize_t)H, (py::ssize_t)W, (py::ssize_t)4});
    if (H > 0 && W > 0) ok(pmx_cross_support_image(context(), image.data(), H, W, len_arms, intensity, out.mutable_data()), "pmx_cross_support_image");
    return out;
}

// aggregation/cpp/src/aggregation.cpp:323-356
std::tuple<py::array_t<float>, py::array_t<float>> cbca(farr input, sarr cross_left, sarr cross_right, larr range_col, larr range_col_right) {
    if (input.ndim() != 2 || cross_left.ndim() != 3 || cross_right.ndim() != 3 || range_col.ndim() != 1 || range_col_right.ndim() != 1)
        throw std::invalid_argument("cbca: input 2-D, cross supports 3-D, column ranges 1-D");
    const int H = (int)input.shape(0), W = (int)input.shape(1), Wr = (int)cross_right.shape(1);
    if (cross_left.shape(0) != H || cross_left.shape(1) != W || cross_left.shape(2) != 4 || cross_right.shape(0) != H || cross_right.shape(2) != 4 ||
        range_col.shape(0) != range_col_right.shape(0))
        throw std::invalid_argument("cbca: shapes disagree");
    py::array_t<float> e({(py::ssize_t)H, (py::ssize_t)W}), n({(py::ssize_t)H, (py::ssize_t)W});
    if (H > 0 && W > 0)
        ok(pmx_cbca_slice(context(), input.data(), cross_left.data(), cross_right.data(), H, W, Wr, range_col.data(), range_col_right.data(),
                          (int)range_col.shape(0), e.mutable_data(), n.mutable_data()), "pmx_cbca_slice");
    return {e, n};
}

// refinement/cpp/src/vfit.cpp:28-56, quadratic.cpp:28-50, refinement_tools.cpp:25-56: the per-pixel callbacks of the reference's
// plugin classes (vfit.py:43-45).  Scalars in, scalars out - three float32 operations that the device path (loop_refinement
// below) never calls; they exist so that `Vfit.refinement_method` keeps its meaning for callers that probe it.
bool stopped(const float* c, const std::string& measure, float& c0, float& c1, float& c2) {
    c0 = c[0]; c1 = c[1]; c2 = c[2];
    if (c0 != c0 || c2 != c2) return true;
    if (measure == "min") return c1 > c0 || c1 > c2;
    return c1 < c0 || c1 < c2;
}
std::tuple<float, float, int64_t> vfit_refinement_method(farr cost, float /*disp*/, const std::string& measure, int64_t cst_stopped) {
    if (cost.size() < 3) throw std::invalid_argument("vfit_refinement_method: three costs");
    float c0, c1, c2;
    if (stopped(cost.data(), measure, c0, c1, c2)) return {0.f, c1, cst_stopped};
    float a = (measure == "min") ? (c0 > c2 ? c0 - c1 : c2 - c1) : (c0 < c2 ? c0 - c1 : c2 - c1);
    if (std::fabs((double)a) < 1.0e-15) return {0.f, c1, 0};
    const float x = (c0 - c2) / (2 * a);
    return {x, a * (x - 1) + c2, 0};
}
std::tuple<float, float, int64_t> quadratic_refinement_method(farr cost, float /*disp*/, const std::string& measure, int64_t cst_stopped) {
    if (cost.size() < 3) throw std::invalid_argument("quadratic_refinement_method: three costs");
    float c0, c1, c2;
    if (stopped(cost.data(), measure, c0, c1, c2)) return {0.f, c1, cst_stopped};
    const float alpha = (c0 - 2.f * c1 + c2) / 2.f, beta = (c2 - c0) / 2.f;
    float x = -beta / (2.f * alpha);
    x = (-1.f < x) ? x : -1.f;
    x = (x < 1.f) ? x : 1.f;
    return {x, (alpha * x * x) + (beta * x) + c1, 0};
}

// Which of the two device kernels does a caller's `method` stand for?  The reference passes `self.refinement_method`, a python
// staticmethod that forwards to one of the two functions above (vfit.py:43-45, quadratic.py): asked for the costs (3, 1, 2) the
// two give (0.25, 0.5) and (1/6, 0.958...).  Anything else cannot run on the device and is refused.
int identify_method(const py::object& method) {
    py::array_t<float> probe(3);
    probe.mutable_data()[0] = 3.f; probe.mutable_data()[1] = 1.f; probe.mutable_data()[2] = 2.f;
    py::object r;
    try {
        r = method(probe, 1.0f, "min");
    } catch (py::error_already_set&) {
        r = method(probe, 1.0f, "min", (int64_t)8);  // the module's own four-argument functions
    }
    py::tuple t = r.cast<py::tuple>();
    const float x = t[0].cast<float>(), y = t[1].cast<float>();
    if (x == 0.25f && y == 0.5f) return PMX_REFINE_VFIT;
    const float alpha = 1.5f, beta = -0.5f, qx = -beta / (2.f * alpha);
    if (x == qx && y == (alpha * qx * qx) + (beta * qx) + 1.f) return PMX_REFINE_QUADRATIC;
    throw std::invalid_argument("loop_refinement: only the vfit and quadratic refinement methods run on the device");
}

// refinement/cpp/src/refinement.cpp:28-99
std::tuple<py::array_t<float>, py::array_t<float>, py::array_t<int64_t>> loop_refinement(farr cv, farr disp, larr mask, double d_min, double d_max,
                                                                                         int subpixel, const std::string& measure, py::object method,
                                                                                         int64_t cst_invalid, int64_t cst_stopped) {
    if (cv.ndim() != 3 || disp.ndim() != 2 || mask.ndim() != 2) throw std::invalid_argument("loop_refinement: cv 3-D, disp and mask 2-D");
    const int H = (int)cv.shape(0), W = (int)cv.shape(1), D = (int)cv.shape(2);
    if (disp.shape(0) != H || disp.shape(1) != W || mask.shape(0) != H || mask.shape(1) != W) throw std::invalid_argument("loop_refinement: shapes disagree");
    if (cst_invalid != 0x3C3 || cst_stopped != 0x8) throw std::invalid_argument("loop_refinement: the device kernels carry pandora.constants' own mask values (963, 8)");
    if (subpixel < 1 || subpixel > 4 || std::llround((d_max - d_min) * subpixel) + 1 != D) throw std::invalid_argument("loop_refinement: [d_min, d_max] x subpixel does not match the volume's depth");
    if (measure != "min" && measure != "max") throw std::invalid_argument("loop_refinement: measure is 'min' or 'max'");
    if (d_min != std::floor(d_min)) throw std::invalid_argument("loop_refinement: d_min is an integer disparity");
    const int which = identify_method(method);
    pmx_ctx* ctx = context();
    std::vector<float> zeros((size_t)H * W, 0.f);  // (the context takes its geometry from the resident pair)
    ok(pmx_set_images(ctx, zeros.data(), zeros.data(), H, W, subpixel), "pmx_set_images");
    cv_guard g{ctx, pmx_cv_alloc(ctx, D, (int)d_min)};
    if (!g.cv) throw std::runtime_error(std::string("pmx_cv_alloc: ") + pmx_last_error());
    ok(pmx_cv_upload(ctx, g.cv, cv.data()), "pmx_cv_upload");
    ok(pmx_set_disparity(ctx, disp.data(), mask.data()), "pmx_set_disparity");
    ok(pmx_refine(ctx, g.cv, which, measure == "max"), "pmx_refine");
    py::array_t<float> itp({(py::ssize_t)H, (py::ssize_t)W}), dout({(py::ssize_t)H, (py::ssize_t)W});
    py::array_t<int64_t> mout({(py::ssize_t)H, (py::ssize_t)W});
    ok(pmx_get_disparity(ctx, dout.mutable_data(), mout.mutable_data(), itp.mutable_data()), "pmx_get_disparity");
    return {itp, dout, mout};
}

// refinement/cpp/src/refinement.cpp:103-182 (refinement_cpp.pyi:82-122): the right map of the "fast" cross-checking route, refined on
// the LEFT volume's diagonals.  disp / mask are the right map's, [d_min, d_max] the LEFT volume's range.
std::tuple<py::array_t<float>, py::array_t<float>, py::array_t<int64_t>> loop_approximate_refinement(
    farr cv, farr disp, larr mask, double d_min, double d_max, int subpixel, const std::string& measure, py::object method, int64_t cst_invalid,
    int64_t cst_stopped) {
    if (cv.ndim() != 3 || disp.ndim() != 2 || mask.ndim() != 2) throw std::invalid_argument("loop_approximate_refinement: cv 3-D, disp and mask 2-D");
    const int H = (int)cv.shape(0), W = (int)cv.shape(1), D = (int)cv.shape(2);
    if (disp.shape(0) != H || disp.shape(1) != W || mask.shape(0) != H || mask.shape(1) != W)
        throw std::invalid_argument("loop_approximate_refinement: shapes disagree");
    if (cst_invalid != 0x3C3 || cst_stopped != 0x8)
        throw std::invalid_argument("loop_approximate_refinement: the device kernels carry pandora.constants' own mask values (963, 8)");
    if (subpixel < 1 || subpixel > 4 || std::llround((d_max - d_min) * subpixel) + 1 != D)
        throw std::invalid_argument("loop_approximate_refinement: [d_min, d_max] x subpixel does not match the volume's depth");
    if (measure != "min" && measure != "max") throw std::invalid_argument("loop_approximate_refinement: measure is 'min' or 'max'");
    if (d_min != std::floor(d_min)) throw std::invalid_argument("loop_approximate_refinement: d_min is an integer disparity");
    const int which = identify_method(method);
    pmx_ctx* ctx = context();
    std::vector<float> zeros((size_t)H * W, 0.f);  // (the context takes its geometry from the resident pair)
    ok(pmx_set_images(ctx, zeros.data(), zeros.data(), H, W, subpixel), "pmx_set_images");
    cv_guard g{ctx, pmx_cv_alloc(ctx, D, (int)d_min)};
    if (!g.cv) throw std::runtime_error(std::string("pmx_cv_alloc: ") + pmx_last_error());
    ok(pmx_cv_upload(ctx, g.cv, cv.data()), "pmx_cv_upload");
    ok(pmx_set_disparity(ctx, disp.data(), mask.data()), "pmx_set_disparity");
    ok(pmx_refine_approximate(ctx, g.cv, which, measure == "max"), "pmx_refine_approximate");
    py::array_t<float> itp({(py::ssize_t)H, (py::ssize_t)W}), dout({(py::ssize_t)H, (py::ssize_t)W});
    py::array_t<int64_t> mout({(py::ssize_t)H, (py::ssize_t)W});
    ok(pmx_get_disparity(ctx, dout.mutable_data(), mout.mutable_data(), itp.mutable_data()), "pmx_get_disparity");
    return {itp, dout, mout};
}

}  // namespace

PYBIND11_MODULE(inner_cpp, m) {
    m.doc() = "The reference's native-function signatures (matching_cost_cpp / aggregation_cpp / refinement_cpp) over libpandora_amd.so";
    m.add_object("_context_guard", py::capsule(static_cast<void*>(&ctx), [](void*) {
                     if (ctx) pmx_destroy(ctx);
                     ctx = nullptr;
                 }));
    m.def("compute_matching_costs", &compute_matching_costs, py::arg("img_left"), py::arg("imgs_right"), py::arg("cv"), py::arg("disps"),
          py::arg("census_width"), py::arg("census_height"));
    m.def("reverse_cost_volume", &reverse_cost_volume, py::arg("left_cv"), py::arg("disp_min"));
    m.def("reverse_disp_range", &reverse_disp_range, py::arg("left_min"), py::arg("left_max"));
    m.def("cross_support", &cross_support, py::arg("image"), py::arg("len_arms"), py::arg("intensity"));
    m.def("cbca", &cbca, py::arg("input"), py::arg("cross_left"), py::arg("cross_right"), py::arg("range_col"), py::arg("range_col_right"));
    m.def("loop_refinement", &loop_refinement, py::arg("cv"), py::arg("disp"), py::arg("mask"), py::arg("d_min"), py::arg("d_max"),
          py::arg("subpixel"), py::arg("measure"), py::arg("method"), py::arg("cst_pandora_msk_pixel_invalid"),
          py::arg("cst_pandora_msk_pixel_stopped_interpolation"));
    m.def("loop_approximate_refinement", &loop_approximate_refinement, py::arg("cv"), py::arg("disp"), py::arg("mask"), py::arg("d_min"),
          py::arg("d_max"), py::arg("subpixel"), py::arg("measure"), py::arg("method"), py::arg("cst_pandora_msk_pixel_invalid"),
          py::arg("cst_pandora_msk_pixel_stopped_interpolation"));
    m.def("vfit_refinement_method", &vfit_refinement_method, py::arg("cost"), py::arg("disp"), py::arg("measure"),
          py::arg("cst_pandora_msk_pixel_stopped_interpolation"));
    m.def("quadratic_refinement_method", &quadratic_refinement_method, py::arg("cost"), py::arg("disp"), py::arg("measure"),
          py::arg("cst_pandora_msk_pixel_stopped_interpolation"));
}

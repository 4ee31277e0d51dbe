/* oracle.h - CPU restatement of the Pandora hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C, single-threaded restatement of the reference's
 * matching_cost -> aggregation -> optimization -> disparity -> refinement arithmetic.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (pandora_amd/csrc, libpandora_amd.so) never links, loads or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/pandora).  Pinning status per function is listed in oracle.c's header.
 *
 * Layout conventions (same as the reference): images are float32 [H][W] row-major; the cost
 * volume is float32 [H][W][D] with the disparity index innermost; disparity index k maps to
 * disparity d0 + k/subpix.
 */
#ifndef PANDORA_ORACLE_H
#define PANDORA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* img_tools.py:713-752 (scipy zoom order 1): out[H][W-1] = (1-f)*R[c] + f*R[c+1], f = k/subpix */
void orc_shift_right(const float* R, int H, int W, int subpix, int k, float* out);

/* census.cpp:32-95.  out: uint8 [H][W][nb_chars]; returns nb_chars */
int orc_census_nb_chars(int win);
void orc_census_transform(const float* img, int H, int W, int win, uint8_t* out);

/* census.cpp:97-180.  Rs = R (H*W) followed by subpix-1 shifted images (H*(W-1) each).
 * cv is filled in place (untouched cells keep the caller's value, normally NaN). */
void orc_census_cost(const float* L, const float* Rs, int H, int W, int D, int d0, int subpix,
                     int win, float* cv);

/* sad_ssd.py:75-207, 226-368 */
void orc_sad_ssd(const float* L, const float* Rs, int H, int W, int D, int d0, int subpix, int win,
                 int squared, float* cv);

/* zncc.py:114-277 + img_tools.py:834-952 */
void orc_zncc(const float* L, const float* Rs, int H, int W, int D, int d0, int subpix, int win,
              float* cv);

/* matching_cost.py:484-602: bad[H][W] = 1 where invalid or (dilated) no-data */
void orc_mask_dilatation(const int16_t* msk, int H, int W, int win, int valid_value, int nodata_value,
                         uint8_t* bad);

/* matching_cost.py:770-872 (steps 1-2: NaN injection).  mskL/mskR may be NULL; dmin/dmax grids
 * (double [H][W]) may be NULL. */
void orc_cv_masked(float* cv, int H, int W, int D, int d0, int subpix, int win, const int16_t* mskL,
                   const int16_t* mskR, int valid_value, int nodata_value, const double* dmin,
                   const double* dmax);

/* filter/median.py:134-179 (3x3 nanmedian) */
void orc_median3(const float* in, int H, int W, float* out);

/* aggregation.cpp:224-321.  cross: int16 [H][W][4] = left,right,up,down */
void orc_cross_support(const float* img, int H, int W, int len_arms, float intensity, int16_t* cross);

/* aggregation.cpp:28-221, 323-356 + cbca.py:127-177 for the whole volume (in place).
 * crossL int16 [Hc][Wc][4]; crossR: subpix arrays, image k is [Hc][Wk][4] with Wk = Wc (k=0) or Wc-1,
 * stored back to back; (Hc,Wc) = (H-2o, W-2o), o = offset_row_col. */
void orc_cbca(float* cv, int H, int W, int D, int d0, int subpix, int offset, const int16_t* crossL,
              const int16_t* crossRs);

/* SGM (external to the reference: pandora_plugin_libsgm==1.5.7 / libSGM; PARITY UNPINNED).
 * Conventions documented in oracle.c and DESIGN.md. out may alias cv. */
void orc_sgm(const float* cv, int H, int W, int D, float P1, float P2, int is_max, float invalid_cost,
             int overcounting, float* out);
/* the same with a subset of the eight paths (bit k = k-th path of the definition's order; 0xff = orc_sgm) */
void orc_sgm_p2maps(const float* cv, int H, int W, int D, float P1, const float* p2maps, int is_max, float invalid_cost,
                    int overcounting, float* out);
void orc_sgm_dirs(const float* cv, int H, int W, int D, float P1, float P2, int is_max, float invalid_cost,
                  int overcounting, int dir_mask, float* out);

/* disparity.py:399-516: WTA.  disp float32 [H][W], valid int64 [H][W] updated in place */
void orc_wta(const float* cv, int H, int W, int D, double d0, int subpix, int is_max,
             float invalid_disparity, float* disp, int64_t* validity);

/* refinement.cpp:28-99 + vfit.cpp:28-56 / quadratic.cpp:28-50 / refinement_tools.cpp:25-56.
 * method 0 = vfit, 1 = quadratic. disp/validity updated in place; itp written. */
void orc_refine(const float* cv, int H, int W, int D, double d_min, double d_max, int subpix, int is_max,
                int method, float* disp, int64_t* validity, float* itp);

/* matching_cost.cpp:26-56 (reverse_cost_volume): (i,j,d) -> (i, j+d, -d) */
void orc_reverse_cost_volume(const float* left_cv, int H, int W, int D, int min_disp, float* right_cv);


/* matching_cost.cpp:59-132 (reverse_disp_range) */
void orc_reverse_disp_range(const float* left_min, const float* left_max, int H, int W, float* right_min,
                            float* right_max);
/* validation/validation.py:226-371 (CrossCheckingAccurate.disparity_checking, both validation methods) */
void orc_cross_checking(const float* disp_left, int64_t* validity_left, const float* disp_right, int H, int W,
                        int dmin, int dmax, double threshold, float* conf);

/* filter/median.py:134-179 (median_filter, any odd size) and :94-131 (MedianFilter.filter_disparity) */
void orc_median_filter(const float* in, int H, int W, int size, float* out);
void orc_filter_median_disparity(float* disp, const int64_t* validity, int H, int W, int size);

/* filter/disparity_denoiser.py:223-313 (DisparityDenoiser.filter_disparity; the caller supplies np.gradient of the blurred map) */
void orc_denoise_disparity(float* disp, const int64_t* validity, const float* color, const float* grad_row, const float* grad_col,
                           int H, int W, int filter_size, double sigma_euclidian, double sigma_color, double sigma_planar);
/* filter/bilateral.py:100-255 (BilateralFilter.filter_disparity) */
void orc_filter_bilateral_disparity(float* disp, const int64_t* validity, int H, int W, double sigma_color,
                                    double sigma_space);

/* multiscale/fixed_zoom_pyramid.py:106-172 (FixedZoomPyramid.disparity_range, before the zoom) */
void orc_disparity_range(const float* disp, const int64_t* validity, int H, int W, int win, int marge, int gmin, int gmax,
                         float* out_min, float* out_max);

/* cost_volume_confidence/cpp/src/ambiguity.cpp:28-142 (integral of the ambiguity) */
void orc_ambiguity(const float* cv, int H, int W, int D, const float* etas, int nbr_etas, const int64_t* grid_min,
                   const int64_t* grid_max, const float* disp_range, float* amb);

/* cpp/src/img_tools.cpp:27-155 (interpolate_nodata_sgm) */
void orc_interpolate_nodata(const float* img, const int32_t* msk, int H, int W, int invalid_bits, int filled_value,
                            float* out_img, int32_t* out_msk);

/* validation/cpp/src/interpolated_disparity.cpp: pass 0/1 = occlusion/mismatch mc-cnn, 2/3 = occlusion/mismatch sgm */
void orc_interpolate_disparity(int pass, const float* disp, const int32_t* valid, int H, int W, float* out_disp, int32_t* out_valid);

/* OpenMP thread count of the census / SGM / WTA / refinement loops: n <= 0 = all cores; returns the count set */
int orc_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif

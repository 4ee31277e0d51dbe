/* oracle.c - CPU restatement of the Pandora hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Not product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so.  The HIP library (pandora_amd/csrc) never links against it.
 *
 * Citations are relative to /root/reference/src/pandora.  Pinning status:
 *   census / cross_support / cbca core / vfit / quadratic / loop_refinement / reverse_cost_volume
 *       : diffed against the reference's own compiled C++ (oracle/_ref, built by oracle/Makefile)
 *         and against the reference tests' golden vectors (tests/golden/).
 *   sad / ssd / zncc / cv_masked / median3 / wta / shift_right_img
 *       : restated from the reference Python; pinned by golden vectors transcribed from the
 *         reference tests (tests/golden/) - the Python itself cannot be imported here.
 *   sgm : the arithmetic lives in pandora_plugin_libsgm==1.5.7 (pyproject.toml:59-61), which is
 *         NOT in the reference tree and has no numeric test there.  PARITY UNPINNED.  The
 *         conventions below are this build's own, documented in DESIGN.md.
 */
#include "oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define IDX3(r, c, k, W, D) (((size_t)(r) * (size_t)(W) + (size_t)(c)) * (size_t)(D) + (size_t)(k))

/* width of the k-th shifted right image (img_tools.py:742: shifted images lose one column) */
static inline int shifted_width(int W, int k) { return k == 0 ? W : W - 1; }

/* pointer to the k-th image inside the back-to-back buffer Rs */
static inline const float* shifted_ptr(const float* Rs, int H, int W, int k) {
    if (k == 0) return Rs;
    return Rs + (size_t)H * W + (size_t)(k - 1) * H * (W - 1);
}

/* floor division of the disparity index: disparity of index k is d0 + k/subpix, its integer
 * part (floor) is d0 + k/subpix (k >= 0) and its sub-pixel phase is k % subpix. */

/* ---------------------------------------------------------------------------------------------
 * img_tools.py:713-752  shift_right_img: scipy.ndimage.zoom(order=1) sampled at k/subpix.
 * Linear interpolation evaluated in double and stored as float32 (zoom's output dtype follows
 * the float32 input).
 * ------------------------------------------------------------------------------------------- */
void orc_shift_right(const float* R, int H, int W, int subpix, int k, float* out) {
    double f = (double)k / (double)subpix;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W - 1; ++c) {
            double a = R[(size_t)r * W + c], b = R[(size_t)r * W + c + 1];
            out[(size_t)r * (W - 1) + c] = (float)((1.0 - f) * a + f * b);
        }
}

/* ---------------------------------------------------------------------------------------------
 * census.cpp:32-43 get_census_info: nb_chars = bits/8 + bits%8 (over-allocates, harmless)
 * ------------------------------------------------------------------------------------------- */
int orc_census_nb_chars(int win) {
    int bits = win * win;
    return bits / 8 + bits % 8;
}

/* census.cpp:45-95: strict '>' against the centre, row-major, MSB first inside each byte;
 * borders (half window) stay 0. */
void orc_census_transform(const float* img, int H, int W, int win, uint8_t* out) {
    int h = win / 2;
    int nb = orc_census_nb_chars(win);
    memset(out, 0, (size_t)H * W * nb);
#pragma omp parallel for schedule(static)
    for (int x = h; x < H - h; ++x)
        for (int y = h; y < W - h; ++y) {
            float val = img[(size_t)x * W + y];
            uint8_t* o = out + ((size_t)x * W + y) * nb;
            int bit = 0, chr = 0;
            for (int wx = x - h; wx <= x + h; ++wx)
                for (int wy = y - h; wy <= y + h; ++wy) {
                    if (img[(size_t)wx * W + wy] > val) o[chr] = (uint8_t)(o[chr] + (0x80u >> bit));
                    if (++bit >= 8) { ++chr; bit = 0; }
                }
        }
}

/* The loops above and below marked `omp parallel for` touch disjoint outputs per iteration, so results do not depend on the
 * thread count; orc_set_threads(1) gives the reference's serial execution (the 1-core CPU baseline), orc_set_threads(0) all cores. */
int orc_set_threads(int n) {
    if (n <= 0) n = omp_get_num_procs();
    omp_set_num_threads(n);
    return n;
}

static inline int popcount8(uint8_t v) { return __builtin_popcount((unsigned)v); }

/* census.cpp:97-180 compute_matching_costs */
void orc_census_cost(const float* L, const float* Rs, int H, int W, int D, int d0, int subpix,
                     int win, float* cv) {
    int h = win / 2;
    int nb = orc_census_nb_chars(win);
    uint8_t* cl = (uint8_t*)malloc((size_t)H * W * nb);
    uint8_t** cr = (uint8_t**)malloc(sizeof(uint8_t*) * (size_t)subpix);
    orc_census_transform(L, H, W, win, cl);
    for (int k = 0; k < subpix; ++k) {
        int Wk = shifted_width(W, k);
        cr[k] = (uint8_t*)malloc((size_t)H * Wk * nb);
        orc_census_transform(shifted_ptr(Rs, H, W, k), H, Wk, win, cr[k]);
    }
#pragma omp parallel for schedule(static)
    for (int row = h; row < H - h; ++row)
        for (int col = h; col < W - h; ++col) {
            const uint8_t* lp = cl + ((size_t)row * W + col) * nb;
            for (int disp = 0; disp < D; disp += subpix) {
                int right_x = col + disp / subpix + d0; /* census.cpp:138 */
                if (right_x < h || right_x >= W - h) continue;
                for (int id = 0; id < subpix && disp + id < D; ++id) {
                    const uint8_t* rp;
                    if (id != 0) {
                        if (right_x >= W - h - 1) break; /* census.cpp:149 */
                        rp = cr[id] + ((size_t)row * (W - 1) + right_x) * nb;
                    } else {
                        rp = cr[0] + ((size_t)row * W + right_x) * nb;
                    }
                    int weight = 0;
                    for (int c = 0; c < nb; ++c) weight += popcount8((uint8_t)(rp[c] ^ lp[c]));
                    cv[IDX3(row, col, disp + id, W, D)] = (float)weight;
                }
            }
        }
    free(cl);
    for (int k = 0; k < subpix; ++k) free(cr[k]);
    free(cr);
}

/* ---------------------------------------------------------------------------------------------
 * SAD / SSD.  sad_ssd.py:146-207 + 226-368 and matching_cost.py:429-482 (point_interval).
 * Cell (r,c,k) is finite iff the w x w window around (r,c) lies inside the image rows and inside
 * the column overlap of disparity k (NaN padding propagates through np.sum otherwise); the
 * right sample of left column p is column p + floor(d) of shifted image (k % subpix).
 * The window sum runs rows-then-columns in float32 (exact for integer-valued images, the only
 * case the reference tests pin; <=1e-5 relative otherwise).
 * ------------------------------------------------------------------------------------------- */
void orc_sad_ssd(const float* L, const float* Rs, int H, int W, int D, int d0, int subpix, int win,
                 int squared, float* cv) {
    int o = win / 2;
    for (int k = 0; k < D; ++k) {
        int ph = k % subpix, fl = d0 + k / subpix;
        int Wk = shifted_width(W, ph);
        const float* R = shifted_ptr(Rs, H, W, ph);
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c) {
                float v = NAN;
                int ok = (r - o >= 0) && (r + o < H) && (c - o >= 0) && (c + o < W) &&
                         (c - o + fl >= 0) && (c + o + fl < Wk);
                if (ok) {
                    float s = 0.f;
                    /* np.sum over the two leading (stride-sorted) window axes of the (disp,col,row) buffer
                     * (sad_ssd.py:340-368) adds sequentially in memory order: window columns outer, window
                     * rows inner - pinned by tests/golden/sad_float_order.npz. */
                    for (int j = -o; j <= o; ++j)
                        for (int i = -o; i <= o; ++i) {
                            float d = L[(size_t)(r + i) * W + c + j] - R[(size_t)(r + i) * Wk + c + j + fl];
                            s += squared ? d * d : fabsf(d);
                        }
                    v = s;
                }
                cv[IDX3(r, c, k, W, D)] = v;
            }
    }
}

/* ---------------------------------------------------------------------------------------------
 * ZNCC.  zncc.py:153-241, apply_divide_standard :244-277, img_tools.py:834-879 (mean raster,
 * float64 integral images) and :915-952 (std raster: squares rounded to float32 first, variance
 * clipped to 0 below 1e-15*|E[x^2]|).  The product L*R is a float32 product (zncc.py:209-212)
 * averaged in float64.  Direct float64 window sums are used instead of integral images: equal on
 * integer-valued images, <=1e-9 apart otherwise.
 * ------------------------------------------------------------------------------------------- */
static void window_stats(const float* img, int H, int Wd, int win, double* mean, double* std) {
    /* mean/std [H-2o][Wd-2o] of every full window */
    int o = win / 2, Ho = H - 2 * o, Wo = Wd - 2 * o;
    double n = (double)win * win;
    for (int r = 0; r < Ho; ++r)
        for (int c = 0; c < Wo; ++c) {
            double s = 0, s2 = 0;
            for (int i = 0; i < win; ++i)
                for (int j = 0; j < win; ++j) {
                    float x = img[(size_t)(r + i) * Wd + c + j];
                    float x2 = x * x; /* selected_band**2 stays float32 (img_tools.py:941) */
                    s += x;
                    s2 += x2;
                }
            double m = s / n, m2 = s2 / n;
            double var = m2 - m * m;
            if (var < 1e-15 * fabs(m2)) var = 0; /* img_tools.py:951 */
            mean[(size_t)r * Wo + c] = m;
            std[(size_t)r * Wo + c] = sqrt(var);
        }
}

void orc_zncc(const float* L, const float* Rs, int H, int W, int D, int d0, int subpix, int win,
              float* cv) {
    int o = win / 2;
    size_t ncell = (size_t)H * W * D;
    for (size_t i = 0; i < ncell; ++i) cv[i] = NAN;
    if (H - 2 * o <= 0 || W - 2 * o <= 0) return;
    int Ho = H - 2 * o;
    double* lm = (double*)malloc(sizeof(double) * (size_t)Ho * (W - 2 * o));
    double* ls = (double*)malloc(sizeof(double) * (size_t)Ho * (W - 2 * o));
    double** rm = (double**)calloc((size_t)subpix, sizeof(double*));
    double** rs = (double**)calloc((size_t)subpix, sizeof(double*));
    window_stats(L, H, W, win, lm, ls);
    for (int k = 0; k < subpix; ++k) {
        int Wk = shifted_width(W, k);
        if (Wk - 2 * o <= 0) continue;
        rm[k] = (double*)malloc(sizeof(double) * (size_t)Ho * (Wk - 2 * o));
        rs[k] = (double*)malloc(sizeof(double) * (size_t)Ho * (Wk - 2 * o));
        window_stats(shifted_ptr(Rs, H, W, k), H, Wk, win, rm[k], rs[k]);
    }
    double n = (double)win * win;
    for (int k = 0; k < D; ++k) {
        int ph = k % subpix, fl = d0 + k / subpix;
        int Wk = shifted_width(W, ph);
        if (!rm[ph]) continue;
        const float* R = shifted_ptr(Rs, H, W, ph);
        int WoL = W - 2 * o, WoR = Wk - 2 * o;
        for (int r = o; r < H - o; ++r)
            for (int c = o; c < W - o; ++c) {
                if (c - o + fl < 0 || c + o + fl >= Wk) continue;
                double s = 0;
                for (int i = -o; i <= o; ++i)
                    for (int j = -o; j <= o; ++j) {
                        float p = L[(size_t)(r + i) * W + c + j] * R[(size_t)(r + i) * Wk + c + j + fl];
                        s += p;
                    }
                double z = s / n;
                size_t il = (size_t)(r - o) * WoL + (c - o);
                size_t ir = (size_t)(r - o) * WoR + (c - o + fl);
                z -= lm[il] * rm[ph][ir];
                double dv = ls[il] * rs[ph][ir];
                if (dv > 0) z /= dv; else z = 0; /* zncc.py:273-277 */
                cv[IDX3(r, c, k, W, D)] = (float)z;
            }
    }
    free(lm); free(ls);
    for (int k = 0; k < subpix; ++k) { free(rm[k]); free(rs[k]); }
    free(rm); free(rs);
}

/* ---------------------------------------------------------------------------------------------
 * matching_cost.py:484-602 masks_dilatation: bad = (msk != valid && msk != nodata) ||
 * binary_dilation(msk == nodata, ones(win,win)).
 * ------------------------------------------------------------------------------------------- */
void orc_mask_dilatation(const int16_t* msk, int H, int W, int win, int valid_value, int nodata_value,
                         uint8_t* bad) {
    int o = win / 2;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            int16_t m = msk[(size_t)r * W + c];
            int b = (m != valid_value) && (m != nodata_value);
            for (int i = -o; i <= o && !b; ++i)
                for (int j = -o; j <= o && !b; ++j) {
                    int rr = r + i, cc = c + j;
                    if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                    if (msk[(size_t)rr * W + cc] == nodata_value) b = 1;
                }
            bad[(size_t)r * W + c] = (uint8_t)b;
        }
}

/* matching_cost.py:770-872 cv_masked, NaN injection part.
 *  (1) :815-839 per disparity: cost += mask_left[p] + mask_right[q]; for an integer disparity
 *      q = p + d inside the right image; for a fractional one the right mask is the 2-column
 *      aggregate (:575-593) at index p + floor(d) in [0, W-2] (mask_column_interval_without_step
 *      :658-712 resolves to exactly these intervals).
 *  (2) :845-860 NaN where disparity < dmin[r,c] or > dmax[r,c]. */
void orc_cv_masked(float* cv, int H, int W, int D, int d0, int subpix, int win, const int16_t* mskL,
                   const int16_t* mskR, int valid_value, int nodata_value, const double* dmin,
                   const double* dmax) {
    uint8_t* bl = NULL;
    uint8_t* br = NULL;
    if (mskL) { bl = (uint8_t*)malloc((size_t)H * W); orc_mask_dilatation(mskL, H, W, win, valid_value, nodata_value, bl); }
    if (mskR) { br = (uint8_t*)malloc((size_t)H * W); orc_mask_dilatation(mskR, H, W, win, valid_value, nodata_value, br); }
    if (bl || br) {
        for (int k = 0; k < D; ++k) {
            int ph = k % subpix, fl = d0 + k / subpix;
            for (int r = 0; r < H; ++r)
                for (int p = 0; p < W; ++p) {
                    int q = p + fl, bad = 0;
                    if (ph == 0) {
                        if (q < 0 || q > W - 1) continue;
                        bad = (bl && bl[(size_t)r * W + p]) || (br && br[(size_t)r * W + q]);
                    } else {
                        if (q < 0 || q > W - 2) continue;
                        bad = (bl && bl[(size_t)r * W + p]) ||
                              (br && (br[(size_t)r * W + q] || br[(size_t)r * W + q + 1]));
                    }
                    if (bad) cv[IDX3(r, p, k, W, D)] = NAN;
                }
        }
    }
    if (dmin && dmax) {
        for (int k = 0; k < D; ++k) {
            double d = (double)d0 + (double)k / (double)subpix;
            for (int r = 0; r < H; ++r)
                for (int c = 0; c < W; ++c)
                    if (d < dmin[(size_t)r * W + c] || d > dmax[(size_t)r * W + c])
                        cv[IDX3(r, c, k, W, D)] = NAN;
        }
    }
    free(bl); free(br);
}

/* ---------------------------------------------------------------------------------------------
 * filter/median.py:134-179: 3x3 np.nanmedian on interior pixels; 1-px border copied; NaN input
 * pixels stay NaN; even counts average the two middle values.
 * ------------------------------------------------------------------------------------------- */
void orc_median3(const float* in, int H, int W, float* out) {
    memcpy(out, in, sizeof(float) * (size_t)H * W);
    for (int r = 1; r < H - 1; ++r)
        for (int c = 1; c < W - 1; ++c) {
            float ctr = in[(size_t)r * W + c];
            if (isnan(ctr)) continue;
            float v[9];
            int n = 0;
            for (int i = -1; i <= 1; ++i)
                for (int j = -1; j <= 1; ++j) {
                    float x = in[(size_t)(r + i) * W + c + j];
                    if (!isnan(x)) v[n++] = x;
                }
            for (int a = 1; a < n; ++a) { /* insertion sort */
                float x = v[a];
                int b = a - 1;
                while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
                v[b + 1] = x;
            }
            float m;
            if (n & 1) m = v[n / 2];
            else m = (v[n / 2 - 1] + v[n / 2]) / 2.0f; /* np.mean of two float32 */
            out[(size_t)r * W + c] = m;
        }
}

/* aggregation.cpp:224-321 cross_support */
void orc_cross_support(const float* img, int H, int W, int len_arms, float intensity, int16_t* cross) {
    for (int row = 0; row < H; ++row)
        for (int col = 0; col < W; ++col) {
            float cur = img[(size_t)row * W + col];
            int16_t* o = cross + ((size_t)row * W + col) * 4;
            if (!isfinite(cur)) { o[0] = o[1] = o[2] = o[3] = 0; continue; }
            int16_t l = 0, rt = 0, up = 0, dn = 0;
            int lo = col - len_arms; if (lo < -1) lo = -1;
            for (int x = col - 1; x > lo; --x) {
                if (fabsf(cur - img[(size_t)row * W + x]) >= intensity) break;
                l++;
            }
            { int16_t m = (int16_t)(col >= 1 && isfinite(img[(size_t)row * W + col - 1])); if (m > l) l = m; }
            int hi = col + len_arms; if (hi > W) hi = W;
            for (int x = col + 1; x < hi; ++x) {
                if (fabsf(cur - img[(size_t)row * W + x]) >= intensity) break;
                rt++;
            }
            { int16_t m = (int16_t)(col < W - 1 && isfinite(img[(size_t)row * W + col + 1])); if (m > rt) rt = m; }
            lo = row - len_arms; if (lo < -1) lo = -1;
            for (int y = row - 1; y > lo; --y) {
                if (fabsf(cur - img[(size_t)y * W + col]) >= intensity) break;
                up++;
            }
            { int16_t m = (int16_t)(row >= 1 && isfinite(img[(size_t)(row - 1) * W + col])); if (m > up) up = m; }
            hi = row + len_arms; if (hi > H) hi = H;
            for (int y = row + 1; y < hi; ++y) {
                if (fabsf(cur - img[(size_t)y * W + col]) >= intensity) break;
                dn++;
            }
            { int16_t m = (int16_t)(row < H - 1 && isfinite(img[(size_t)(row + 1) * W + col])); if (m > dn) dn = m; }
            o[0] = l; o[1] = rt; o[2] = up; o[3] = dn;
        }
}

/* ---------------------------------------------------------------------------------------------
 * CBCA for the whole volume: cbca.py:127-177 driving aggregation.cpp:28-221 per disparity.
 * step 1 row running sum (NaN skipped), step 2 horizontal segment sum with combined arms,
 * step 3 column running sum, step 4 vertical segment sum + support count; then
 * out = NaN where in is NaN else step4 / (sum4 + 1).  step1(row,-1) is DEFINED as 0 here
 * (the reference reads the previous row's zero pad / out of bounds, aggregation.cpp:113-114).
 * ------------------------------------------------------------------------------------------- */
void orc_cbca(float* cv, int H, int W, int D, int d0, int subpix, int offset, const int16_t* crossL,
              const int16_t* crossRs) {
    int o = offset;
    int Hc = H - 2 * o, Wc = W - 2 * o;
    if (Hc <= 0 || Wc <= 0) return;
    float* s1 = (float*)malloc(sizeof(float) * (size_t)Hc * (Wc + 1));
    float* e2 = (float*)malloc(sizeof(float) * (size_t)Hc * Wc);
    float* n2 = (float*)malloc(sizeof(float) * (size_t)Hc * Wc);
    float* s3 = (float*)malloc(sizeof(float) * (size_t)(Hc + 1) * Wc);
    for (int k = 0; k < D; ++k) {
        int ph = k % subpix;
        double dd = (double)d0 + (double)k / (double)subpix;
        int Wr = shifted_width(Wc, ph);
        const int16_t* cr = crossRs;
        if (ph > 0) cr = crossRs + (size_t)Hc * Wc * 4 + (size_t)(ph - 1) * Hc * (Wc - 1) * 4;
        /* step 1 */
        for (int r = 0; r < Hc; ++r) {
            float acc = 0.f;
            for (int c = 0; c < Wc; ++c) {
                float v = cv[IDX3(r + o, c + o, k, W, D)];
                if (!isnan(v)) acc = acc + v;
                s1[(size_t)r * (Wc + 1) + c] = acc;
            }
            s1[(size_t)r * (Wc + 1) + Wc] = 0.f;
        }
        /* step 2 */
        for (size_t i = 0; i < (size_t)Hc * Wc; ++i) { e2[i] = 0.f; n2[i] = 0.f; }
        for (int r = 0; r < Hc; ++r)
            for (int c = 0; c < Wc; ++c) {
                double cr_col = (double)c + dd; /* cbca.py:156-157 */
                if (!(cr_col >= 0 && cr_col < (double)Wr)) continue;
                int q = (int)cr_col; /* astype(int) */
                const int16_t* al = crossL + ((size_t)r * Wc + c) * 4;
                const int16_t* ar = cr + ((size_t)r * Wr + q) * 4;
                int left = al[0] < ar[0] ? al[0] : ar[0];
                int right = al[1] < ar[1] ? al[1] : ar[1];
                int lo = c - left - 1;
                float a = s1[(size_t)r * (Wc + 1) + c + right];
                float b = lo < 0 ? 0.f : s1[(size_t)r * (Wc + 1) + lo];
                e2[(size_t)r * Wc + c] = a - b;
                n2[(size_t)r * Wc + c] += (float)(left + right);
            }
        /* step 3 */
        for (int c = 0; c < Wc; ++c) { s3[c] = e2[c]; s3[(size_t)Hc * Wc + c] = 0.f; }
        for (int r = 1; r < Hc; ++r)
            for (int c = 0; c < Wc; ++c)
                s3[(size_t)r * Wc + c] = s3[(size_t)(r - 1) * Wc + c] + e2[(size_t)r * Wc + c];
        /* step 4 + normalisation */
        for (int r = 0; r < Hc; ++r)
            for (int c = 0; c < Wc; ++c) {
                float step4 = 0.f, sum4 = n2[(size_t)r * Wc + c];
                double cr_col = (double)c + dd;
                if (cr_col >= 0 && cr_col < (double)Wr) {
                    int q = (int)cr_col;
                    const int16_t* al = crossL + ((size_t)r * Wc + c) * 4;
                    const int16_t* ar = cr + ((size_t)r * Wr + q) * 4;
                    int top = al[2] < ar[2] ? al[2] : ar[2];
                    int bot = al[3] < ar[3] ? al[3] : ar[3];
                    int sr = r - top - 1;
                    if (sr < 0) sr += Hc + 1; /* wraps to the zero row */
                    step4 = s3[(size_t)(r + bot) * Wc + c] - s3[(size_t)sr * Wc + c];
                    sum4 += (float)(top + bot);
                    if (top > 0) { float s = 0; for (int i = 1; i <= top; ++i) s += n2[(size_t)(r - i) * Wc + c]; sum4 += s; }
                    if (bot > 0) { float s = 0; for (int i = 1; i <= bot; ++i) s += n2[(size_t)(r + i) * Wc + c]; sum4 += s; }
                }
                sum4 += 1.f; /* cbca.py:166 */
                size_t id = IDX3(r + o, c + o, k, W, D);
                float in = cv[id];
                float base = in * 0.f; /* cbca.py:145-146: NaN stays NaN, finite -> 0 */
                cv[id] = (base + step4) / sum4;
            }
    }
    free(s1); free(e2); free(n2); free(s3);
}

/* ---------------------------------------------------------------------------------------------
 * SGM - this build's definition (PARITY UNPINNED; see header).  Energy of
 * docs/source/userguide/plugins/plugin_libsgm.rst:11 minimised along 8 paths (Hirschmuller 2008):
 *   C'(p,d)   = C(p,d) (negated for a "max" measure); NaN -> invalid_cost
 *   L_r(p,d)  = C'(p,d) + ( min( L_r(p-r,d), min(L_r(p-r,d-1), L_r(p-r,d+1)) + P1, M + P2 ) - M )
 *               with M = min_k L_r(p-r,k); L_r(p,d) = C'(p,d) when p-r is outside the image;
 *               d-1 / d+1 outside [0,D) count as +inf
 *   S(p,d)    = (S_H + S_D) + S_U, the paths summed FAMILY BY FAMILY in float32 [(drow,dcol) = the step from p-r to p]:
 *                 S_H = L(0,+1) + L(0,-1)                 the horizontal pair
 *                 S_D = (L(+1,0) + L(+1,+1)) + L(+1,-1)   the three downward paths
 *                 S_U = (L(-1,0) + L(-1,+1)) + L(-1,-1)   the three upward paths
 *               (each family on an accumulator of its own that starts at +0.)  Round 6: until then the eight paths were
 *               added onto ONE running sum in the order H, D, U (round 1: yet another order).  Nothing pins the order -
 *               libSGM is not vendored, it is invisible for integer-valued costs, and north_star asks 1e-5 of float
 *               costs - so the choice is this build's own, and a family-wise sum is what lets the three families of
 *               the GPU schedule run side by side instead of one after the other.
 *   dir_mask  : bit k set = the k-th path of the list (0,+1) (0,-1) (+1,0) (+1,+1) (+1,-1) (-1,0) (-1,+1) (-1,-1)
 *               contributes (0xff = the definition; subsets are a test hook that lets a strip of a large image be checked
 *               path family by family): a family without a path is left out of the sum
 *   overcounting: S -= 7*C'
 *   output    = S (negated back for "max"); NaN wherever the input was NaN.
 * All arithmetic is float32, in exactly the operation order written above.
 * ------------------------------------------------------------------------------------------- */
/* one pixel of one path: p-r = (pr, pc) holds Lq (or lies outside the image) */
static inline void sgm_pixel(const float* Cc, const float* Lq, int D, float P1, float P2, float* Lo, float* So) {
    if (!Lq) {
        for (int d = 0; d < D; ++d) { Lo[d] = Cc[d]; So[d] = So[d] + Cc[d]; }
        return;
    }
    float M = Lq[0];
    for (int d = 1; d < D; ++d) if (Lq[d] < M) M = Lq[d];
    float mp2 = M + P2;
    for (int d = 0; d < D; ++d) {
        float a = d > 0 ? Lq[d - 1] : INFINITY;
        float b = d < D - 1 ? Lq[d + 1] : INFINITY;
        float nb = (a < b ? a : b) + P1;
        float t = Lq[d] < nb ? Lq[d] : nb;
        t = t < mp2 ? t : mp2;
        float l = Cc[d] + (t - M);
        Lo[d] = l;
        So[d] = So[d] + l;
    }
}

/* Pixels are visited so that p-r comes before p.  Every cell sees exactly the operations written in the header, in
 * that order, whatever the thread count: horizontal paths run their rows in parallel (a private two-pixel buffer per
 * row), the others run the columns of a row in parallel against the finished previous row (rolling buffers prev / cur,
 * [W][D] each). */
/* p2map: NULL, or the P2 of every pixel for THIS direction ([H][W]: penalty methods that follow the image gradient along the
 * path, plugin_libsgm.rst:20-27); the pixel's own value enters its own update */
static void sgm_path(const float* Cp, int H, int W, int D, int dr, int dc, float P1, float P2, const float* p2map,
                     float* S, float* prev, float* cur) {
    int c0 = dc >= 0 ? 0 : W - 1, c1 = dc >= 0 ? W : -1, cs = dc >= 0 ? 1 : -1;
    if (dr == 0) {
#pragma omp parallel
        {
            float* two = (float*)malloc(sizeof(float) * 2 * (size_t)D);
#pragma omp for schedule(static)
            for (int r = 0; r < H; ++r) {
                int which = 0;
                for (int c = c0; c != c1; c += cs, which ^= 1) {
                    int pc = c - dc;
                    sgm_pixel(Cp + IDX3(r, c, 0, W, D), (pc < 0 || pc >= W) ? NULL : two + (size_t)(which ^ 1) * D, D, P1,
                              p2map ? p2map[(size_t)r * W + c] : P2, two + (size_t)which * D, S + IDX3(r, c, 0, W, D));
                }
            }
            free(two);
        }
        return;
    }
    int r0 = dr >= 0 ? 0 : H - 1, r1 = dr >= 0 ? H : -1, rs = dr >= 0 ? 1 : -1;
    float* Lrow[2] = {prev, cur};
    int which = 0;
    for (int r = r0; r != r1; r += rs, which ^= 1) {
        float* Lc = Lrow[which];
        const float* Lp = Lrow[which ^ 1];
        int pr = r - dr;
#pragma omp parallel for schedule(static)
        for (int c = 0; c < W; ++c) {
            int pc = c - dc;
            sgm_pixel(Cp + IDX3(r, c, 0, W, D), (pr < 0 || pr >= H || pc < 0 || pc >= W) ? NULL : Lp + (size_t)pc * D, D, P1,
                      p2map ? p2map[(size_t)r * W + c] : P2, Lc + (size_t)c * D, S + IDX3(r, c, 0, W, D));
        }
    }
}

static void sgm_all(const float* cv, int H, int W, int D, float P1, float P2, const float* p2maps, int is_max, float invalid_cost,
                    int overcounting, int dir_mask, float* out) {
    size_t n = (size_t)H * W * D;
    float* Cp = (float*)malloc(sizeof(float) * n);
    float* S = (float*)calloc(n, sizeof(float));
    float* b0 = (float*)malloc(sizeof(float) * (size_t)W * D);
    float* b1 = (float*)malloc(sizeof(float) * (size_t)W * D);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float v = cv[i];
        if (isnan(v)) v = invalid_cost; else if (is_max) v = -v;
        Cp[i] = v;
    }
    static const int dirs[8][2] = {{0, 1}, {0, -1}, {1, 0}, {1, 1}, {1, -1}, {-1, 0}, {-1, 1}, {-1, -1}};
    static const int fam_first[4] = {0, 2, 5, 8};
    float* F = (float*)malloc(sizeof(float) * n); /* the family being summed */
    for (int f = 0; f < 3; ++f) {
        int any = 0;
        for (int k = fam_first[f]; k < fam_first[f + 1]; ++k) any |= dir_mask >> k & 1;
        if (!any) continue;
        memset(F, 0, sizeof(float) * n);
        for (int k = fam_first[f]; k < fam_first[f + 1]; ++k)
            if (dir_mask >> k & 1)
                sgm_path(Cp, H, W, D, dirs[k][0], dirs[k][1], P1, P2, p2maps ? p2maps + (size_t)k * H * W : NULL, F, b0, b1);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) S[i] = S[i] + F[i];
    }
    free(F);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float s = S[i];
        if (overcounting) s = s - 7.0f * Cp[i];
        if (is_max) s = -s;
        out[i] = isnan(cv[i]) ? NAN : s;
    }
    free(Cp); free(S); free(b0); free(b1);
}

void orc_sgm_dirs(const float* cv, int H, int W, int D, float P1, float P2, int is_max, float invalid_cost,
                  int overcounting, int dir_mask, float* out) {
    sgm_all(cv, H, W, D, P1, P2, NULL, is_max, invalid_cost, overcounting, dir_mask, out);
}

/* P2 per pixel and direction: p2maps [8][H][W] in the definition's direction order */
void orc_sgm_p2maps(const float* cv, int H, int W, int D, float P1, const float* p2maps, int is_max, float invalid_cost,
                    int overcounting, float* out) {
    sgm_all(cv, H, W, D, P1, 0.f, p2maps, is_max, invalid_cost, overcounting, 0xff, out);
}

void orc_sgm(const float* cv, int H, int W, int D, float P1, float P2, int is_max, float invalid_cost,
             int overcounting, float* out) {
    orc_sgm_dirs(cv, H, W, D, P1, P2, is_max, invalid_cost, overcounting, 0xff, out);
}

/* ---------------------------------------------------------------------------------------------
 * disparity.py:399-480 WinnerTakesAll.to_disp + argmin_split :482-516 / argmax_split :518-553.
 * NaN counts as +inf (min) / -inf (max); first extremum wins; all-NaN pixels get
 * invalid_disparity and, if no INVALID bit is set yet, validity := PANDORA_MSK_PIXEL_INVALID.
 * ------------------------------------------------------------------------------------------- */
#define MSK_INVALID 0x3C3 /* constants.py:31  0b01111000011 */
#define MSK_STOPPED 0x8   /* constants.py:40 */

void orc_wta(const float* cv, int H, int W, int D, double d0, int subpix, int is_max,
             float invalid_disparity, float* disp, int64_t* validity) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const float* p = cv + IDX3(r, c, 0, W, D);
            int best = 0, allnan = 1;
            float bv = isnan(p[0]) ? (is_max ? -INFINITY : INFINITY) : p[0];
            if (!isnan(p[0])) allnan = 0;
            for (int k = 1; k < D; ++k) {
                float v = p[k];
                if (isnan(v)) v = is_max ? -INFINITY : INFINITY; else allnan = 0;
                if (is_max ? (v > bv) : (v < bv)) { bv = v; best = k; }
            }
            size_t i = (size_t)r * W + c;
            if (allnan) {
                disp[i] = invalid_disparity;
                if (validity && (validity[i] & MSK_INVALID) == 0) validity[i] = MSK_INVALID;
            } else {
                /* coords["disp"]: arange(dmin, dmax, 1/subpix) + [dmax] (matching_cost.py:409-427) */
                double d = d0 + (double)best / (double)subpix;
                disp[i] = (float)d;
            }
        }
}

/* refinement_tools.cpp:25-56 */
static int validate_costs(float c0, float c1, float c2, int is_max, float* ic0, float* ic1, float* ic2) {
    if (isnan(c0) || isnan(c2)) return 0;
    float inv = is_max ? -1.f : 1.f;
    *ic0 = inv * c0; *ic1 = inv * c1; *ic2 = inv * c2;
    if (*ic1 > *ic0 || *ic1 > *ic2) return 0;
    return 1;
}

/* vfit.cpp:28-56 */
static void vfit(float c0, float c1, float c2, int is_max, float* sd, float* sc, int* flag) {
    float ic0 = 0, ic1 = 0, ic2 = 0;
    if (!validate_costs(c0, c1, c2, is_max, &ic0, &ic1, &ic2)) { *sd = 0.f; *sc = c1; *flag = MSK_STOPPED; return; }
    float a = ic0 > ic2 ? c0 - c1 : c2 - c1;
    if (fabs((double)a) < 1.0e-15) { *sd = 0.f; *sc = c1; *flag = 0; return; }
    float sub = (c0 - c2) / (2 * a);
    *sd = sub;
    *sc = a * (sub - 1) + c2;
    *flag = 0;
}

/* quadratic.cpp:28-50 (std::min/std::max comparison semantics preserved for NaN/inf) */
static void quadratic(float c0, float c1, float c2, int is_max, float* sd, float* sc, int* flag) {
    float ic0 = 0, ic1 = 0, ic2 = 0;
    if (!validate_costs(c0, c1, c2, is_max, &ic0, &ic1, &ic2)) { *sd = 0.f; *sc = c1; *flag = MSK_STOPPED; return; }
    float alpha = (c0 - 2.f * c1 + c2) / 2.f;
    float beta = (c2 - c0) / 2.f;
    float x = -beta / (2.f * alpha);
    float mx = (-1.f < x) ? x : -1.f; /* std::max(-1.f, x) */
    float sub = (mx < 1.f) ? mx : 1.f; /* std::min(1.f, mx) */
    *sd = sub;
    *sc = (alpha * sub * sub) + (beta * sub) + c1;
    *flag = 0;
}

/* refinement.cpp:28-99 loop_refinement */
void orc_refine(const float* cv, int H, int W, int D, double d_min, double d_max, int subpix, int is_max,
                int method, float* disp, int64_t* validity, float* itp) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            size_t i = (size_t)r * W + c;
            if ((MSK_INVALID & validity[i]) != 0) { itp[i] = NAN; continue; }
            float raw = disp[i];
            int k = (int)(((double)raw - d_min) * subpix);
            float cost = cv[IDX3(r, c, k, W, D)];
            if (isnan(cost)) { itp[i] = cost; continue; }
            if ((double)raw == d_min || (double)raw == d_max) { itp[i] = cost; validity[i] += MSK_STOPPED; continue; }
            float sd, sc; int flag;
            float c0 = cv[IDX3(r, c, k - 1, W, D)], c2 = cv[IDX3(r, c, k + 1, W, D)];
            if (method == 0) vfit(c0, cost, c2, is_max, &sd, &sc, &flag);
            else quadratic(c0, cost, c2, is_max, &sd, &sc, &flag);
            disp[i] = raw + sd / (float)subpix;
            itp[i] = sc;
            validity[i] += flag;
        }
}

/* matching_cost.cpp:26-56 */
void orc_reverse_cost_volume(const float* left_cv, int H, int W, int D, int min_disp, float* right_cv) {
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j)
            for (int d = 0; d < D; ++d) {
                int col = j + d + min_disp;
                right_cv[IDX3(i, j, d, W, D)] =
                    (col < 0 || col >= W) ? NAN : left_cv[IDX3(i, col, D - 1 - d, W, D)];
            }
}

/* ---------------------------------------------------------------------------------------------
 * matching_cost.cpp:59-132 reverse_disp_range: per-pixel right disparity ranges from the left ones.
 * For every left pixel and every integer d in [(int)min, (int)max] the right pixel col+d (when inside the
 * row) receives -d into its running min / max; untouched right pixels end NaN.  NaN ranges are skipped.
 * ------------------------------------------------------------------------------------------- */
void orc_reverse_disp_range(const float* left_min, const float* left_max, int H, int W, float* right_min,
                            float* right_max) {
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        right_min[i] = INFINITY;
        right_max[i] = -INFINITY;
    }
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            float a = left_min[(size_t)r * W + c], b = left_max[(size_t)r * W + c];
            if (isnan(a) || isnan(b)) continue;
            int dmin = (int)a, dmax = (int)b;
            for (int d = dmin; d <= dmax; ++d) {
                int rc = c + d;
                if (rc < 0) continue;
                if (rc >= W) break;
                size_t k = (size_t)r * W + rc;
                if ((float)(-d) < right_min[k]) right_min[k] = (float)(-d);
                if ((float)(-d) > right_max[k]) right_max[k] = (float)(-d);
            }
        }
    for (size_t i = 0; i < (size_t)H * W; ++i)
        if (isinf(right_min[i])) {
            right_min[i] = NAN;
            right_max[i] = NAN;
        }
}

/* ---------------------------------------------------------------------------------------------
 * validation/validation.py:226-371 CrossCheckingAccurate.disparity_checking (also registered as
 * cross_checking_fast), per left pixel that is not PANDORA_MSK_PIXEL_INVALID:
 *   q = rint(col + disp_left) (numpy rint: half to even; float32 + int64 promotes to float64);
 *   q outside the row: nothing happens (the reference's `outside_right` test `(q < 0) & (q >= W)` is never
 *   true, validation.py:354-355 - reproduced);
 *   dist = |disp_right[q] + disp_left| in float32 with NaN -> +inf on both sides; conf = dist;
 *   dist > threshold: MISMATCH (bit 9) if some d of arange(dmin, dmax+1) has col+d inside the row and
 *   rint(disp_right[col+d]) == -d, else OCCLUSION (bit 8)  (:321-351).
 * conf starts NaN.  A valid pixel whose disparity is NaN makes the reference's index arrays lose alignment
 * (:278-280); the pipeline never produces one (invalid_disparity only lands on INVALID pixels) and this
 * restatement treats it as "q outside".  mask_border (:368-369) is applied by the caller.
 * ------------------------------------------------------------------------------------------- */
#define ORC_MSK_INVALID 0x3C3
#define ORC_MSK_OCCLUSION (1 << 8)
#define ORC_MSK_MISMATCH (1 << 9)
void orc_cross_checking(const float* disp_left, int64_t* validity_left, const float* disp_right, int H, int W,
                        int dmin, int dmax, double threshold, float* conf) {
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            size_t i = (size_t)r * W + c;
            conf[i] = NAN;
            if (validity_left[i] & ORC_MSK_INVALID) continue;
            float dl = disp_left[i];
            if (isnan(dl)) continue;
            double qf = rint((double)c + (double)dl);
            if (!(qf >= 0 && qf < W)) continue;
            int q = (int)qf;
            float dr = disp_right[(size_t)r * W + q];
            if (isnan(dr)) dr = INFINITY;
            float dist = fabsf(dr + dl);
            conf[i] = dist;
            if (!((double)dist > threshold)) continue;
            int mismatch = 0;
            for (int d = dmin; d <= dmax && !mismatch; ++d) {
                int cc = c + d;
                if (cc < 0 || cc >= W) continue;
                float v = disp_right[(size_t)r * W + cc];
                if (rintf(v) == (float)(-d)) mismatch = 1;
            }
            validity_left[i] += mismatch ? ORC_MSK_MISMATCH : ORC_MSK_OCCLUSION;
        }
}

/* ---------------------------------------------------------------------------------------------
 * filter/median.py:134-179 median_filter for any odd size: np.nanmedian over size x size on the pixels whose
 * window fits (the frame of size/2 pixels keeps its values); NaN input pixels stay NaN; even counts average
 * the two middle values in float32.  orc_median3 is the size-3 case kept for the CBCA path.
 * ------------------------------------------------------------------------------------------- */
void orc_median_filter(const float* in, int H, int W, int size, float* out) {
    int rad = size / 2;
    memcpy(out, in, sizeof(float) * (size_t)H * W);
    float* v = (float*)malloc(sizeof(float) * (size_t)size * size);
    for (int r = rad; r < H - rad; ++r)
        for (int c = rad; c < W - rad; ++c) {
            if (isnan(in[(size_t)r * W + c])) continue;
            int n = 0;
            for (int i = -rad; i <= rad; ++i)
                for (int j = -rad; j <= rad; ++j) {
                    float x = in[(size_t)(r + i) * W + c + j];
                    if (!isnan(x)) v[n++] = x;
                }
            for (int a = 1; a < n; ++a) {
                float x = v[a];
                int b = a - 1;
                while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
                v[b + 1] = x;
            }
            out[(size_t)r * W + c] = (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0f;
        }
    free(v);
}

/* filter/median.py:94-131 MedianFilter.filter_disparity: invalid pixels (validity & INVALID) -> NaN, median filter,
 * written back on the pixels that were finite; the disparity of invalid pixels is left as it was. */
void orc_filter_median_disparity(float* disp, const int64_t* validity, int H, int W, int size) {
    size_t n = (size_t)H * W;
    float* masked = (float*)calloc(n, sizeof(float));
    float* med = (float*)calloc(n, sizeof(float));
    for (size_t i = 0; i < n; ++i) masked[i] = (validity[i] & ORC_MSK_INVALID) ? NAN : disp[i];
    orc_median_filter(masked, H, W, size, med);
    for (size_t i = 0; i < n; ++i)
        if (isfinite(masked[i])) disp[i] = med[i];
    free(masked);
    free(med);
}

/* ---------------------------------------------------------------------------------------------
 * filter/bilateral.py:100-255 BilateralFilter.filter_disparity.  win = min(H, W, int(3*sigma_space + 1)), offset =
 * win / 2; window of the output pixel (r, c) = rows r-offset .. r-offset+win-1 (same for columns), only where it fits.
 * weight = gauss_spatial(i, j) * gauss_color(w - centre): the spatial kernel in float64 from the distance to
 * (win/2, win/2), the colour kernel exp() evaluated in float32 on the float32 difference (numpy keeps float32
 * through (x / sigma) ** 2 * 0.5 and np.exp) then scaled in float64; out = nansum(w * weight) / nansum(weight) with NaN
 * window elements ignored.  Invalid pixels count as NaN and keep their value; the result is written on finite pixels.
 * ------------------------------------------------------------------------------------------- */
void orc_filter_bilateral_disparity(float* disp, const int64_t* validity, int H, int W, double sigma_color,
                                    double sigma_space) {
    size_t n = (size_t)H * W;
    int win = (int)(3 * sigma_space + 1);
    if (win > H) win = H;
    if (win > W) win = W;
    int offset = win / 2;
    float* masked = (float*)calloc(n, sizeof(float));
    float* out = (float*)calloc(n, sizeof(float));
    double* gs = (double*)malloc(sizeof(double) * (size_t)win * win);
    const double two_pi_root = sqrt(2 * 3.14159265358979323846);
    for (int i = 0; i < win; ++i)
        for (int j = 0; j < win; ++j) {
            double dist = sqrt((double)((i - win / 2) * (i - win / 2) + (j - win / 2) * (j - win / 2)));
            gs[i * win + j] = exp(-((dist / sigma_space) * (dist / sigma_space)) * 0.5) / (sigma_space * two_pi_root);
        }
    for (size_t i = 0; i < n; ++i) masked[i] = (validity[i] & ORC_MSK_INVALID) ? NAN : disp[i];
    memcpy(out, masked, sizeof(float) * n);
    const float sc = (float)sigma_color;
    for (int r = offset; r + win - offset <= H; ++r)
        for (int c = offset; c + win - offset <= W; ++c) {
            float ctr = masked[(size_t)r * W + c];
            double num = 0, den = 0;
            for (int i = 0; i < win; ++i)
                for (int j = 0; j < win; ++j) {
                    float w = masked[(size_t)(r - offset + i) * W + (c - offset + j)];
                    float t = (w - ctr) / sc;
                    float g = expf(-(t * t) * 0.5f);
                    double weight = gs[i * win + j] * ((double)g / (sigma_color * two_pi_root));
                    if (isnan(weight)) continue; /* np.nansum over both */
                    num += (double)w * weight;
                    den += weight;
                }
            out[(size_t)r * W + c] = (float)(num / den);
        }
    for (size_t i = 0; i < n; ++i)
        if (isfinite(masked[i])) disp[i] = out[i];
    free(masked);
    free(out);
    free(gs);
}

/* ---------------------------------------------------------------------------------------------
 * filter/disparity_denoiser.py:223-313 DisparityDenoiser.filter_disparity after get_grad (:138-149: scipy's gaussian_filter and
 * np.gradient, which the caller runs - grad_row / grad_col are their two planes).  Per pixel, over the filter_size^2 window of
 * the maps padded with numpy's "reflect" (:151-166):
 *   dist(i,j)   = disp(i,j) - (i * grad_row(0,0) + j * grad_col(0,0))                float64            (:204-208)
 *   planar      = dist - disp(0,0),  centred = dist - mean(dist)                                         (:210-214)
 *   w(i,j)      = g(|(i,j)|, s_euclid) * g(colour(i,j) - colour(0,0), s_color) * g(centred, s_planar)    (:290-295)
 *                 with g(v, s) = exp(-(v/s)^2 / 2) (:38-48), the colour one in the image's float32
 *   out         = disp(0,0) + sum(planar * w / sum(w))                                                   (:228-232)
 * written where the pixel is not flagged invalid and finite (:297-303).  NaNs propagate as they do in numpy.
 * ------------------------------------------------------------------------------------------- */
static int orc_reflect(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - m;
}

void orc_denoise_disparity(float* disp, const int64_t* validity, const float* color, const float* grad_row, const float* grad_col,
                           int H, int W, int filter_size, double sigma_euclidian, double sigma_color, double sigma_planar) {
    const size_t n = (size_t)H * W;
    const int o = filter_size / 2, ws = filter_size;
    float* out = (float*)malloc(sizeof(float) * n);
    double* dist = (double*)malloc(sizeof(double) * (size_t)ws * ws);
    const float sc = (float)sigma_color;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const size_t at = (size_t)r * W + c;
            const double g0 = grad_row[at], g1 = grad_col[at];
            const float dc = disp[at], cc = color[at];
            double mean = 0.0;
            for (int i = -o; i <= o; ++i)
                for (int j = -o; j <= o; ++j) {
                    const size_t q = (size_t)orc_reflect(r + i, H) * W + orc_reflect(c + j, W);
                    const double d = (double)disp[q] - ((double)i * g0 + (double)j * g1);
                    dist[(i + o) * ws + (j + o)] = d;
                    mean += d;
                }
            mean /= (double)(ws * ws);
            double sw = 0.0, sp = 0.0;
            for (int i = -o; i <= o; ++i)
                for (int j = -o; j <= o; ++j) {
                    const size_t q = (size_t)orc_reflect(r + i, H) * W + orc_reflect(c + j, W);
                    const double d = dist[(i + o) * ws + (j + o)];
                    const double e = sqrt((double)(i * i + j * j)) / sigma_euclidian;
                    const float tc = (color[q] - cc) / sc;
                    const double pc = (d - mean) / sigma_planar;
                    const double w = exp(-(e * e) / 2.0) * (double)expf(-(tc * tc) / 2.0f) * exp(-(pc * pc) / 2.0);
                    sw += w;
                    sp += (d - (double)dc) * w;
                }
            out[at] = (float)((double)dc + sp / sw);
        }
    for (size_t i = 0; i < n; ++i)
        if (!(validity[i] & ORC_MSK_INVALID) && isfinite(disp[i])) disp[i] = out[i];
    free(out);
    free(dist);
}

/* ---------------------------------------------------------------------------------------------
 * multiscale/fixed_zoom_pyramid.py:106-172 FixedZoomPyramid.disparity_range before the zoom: invalid pixels -> NaN
 * (multiscale.py:129-153); interior pixels get nanmin(window) - marge / nanmax(window) + marge in float32; the frame of
 * window/2 pixels and the pixels that are NaN themselves get the global range (int(nanmin(disp_min)), int(nanmax(disp_max))).
 * ------------------------------------------------------------------------------------------- */
void orc_disparity_range(const float* disp, const int64_t* validity, int H, int W, int win, int marge, int gmin, int gmax,
                         float* out_min, float* out_max) {
    int off = (win - 1) / 2;
    size_t n = (size_t)H * W;
    float* m = (float*)calloc(n, sizeof(float));
    for (size_t i = 0; i < n; ++i) {
        m[i] = (validity[i] & ORC_MSK_INVALID) ? NAN : disp[i];
        out_min[i] = (float)gmin;
        out_max[i] = (float)gmax;
    }
    for (int r = off; r + win - off <= H; ++r)
        for (int c = off; c + win - off <= W; ++c) {
            if (isnan(m[(size_t)r * W + c])) continue;
            float lo = INFINITY, hi = -INFINITY;
            for (int i = 0; i < win; ++i)
                for (int j = 0; j < win; ++j) {
                    float v = m[(size_t)(r - off + i) * W + (c - off + j)];
                    if (isnan(v)) continue;
                    if (v < lo) lo = v;
                    if (v > hi) hi = v;
                }
            out_min[(size_t)r * W + c] = lo - (float)marge;
            out_max[(size_t)r * W + c] = hi + (float)marge;
        }
    free(m);
}

/* ---------------------------------------------------------------------------------------------
 * cost_volume_confidence/cpp/src/ambiguity.cpp:28-142 compute_ambiguity_and_sampled_ambiguity (integral only) with
 * min_max_cost / searchsorted of cost_volume_confidence_tools.cpp:22-88.  Costs normalised with the volume's global
 * min / max (NaN ignored); a NaN cost counts for every eta when its disparity lies inside the pixel's
 * [grid_min, grid_max] (-inf) and never otherwise (+inf); a pixel without any cost gets nbr_etas * D.
 * grid_min / grid_max: int64 [H][W].  disp_range: the D float32 disparity samples of the volume.
 * ------------------------------------------------------------------------------------------- */
static size_t orc_searchsorted(const float* arr, size_t n, float value) {
    size_t left = 0, right = n - 1;
    while (left < right) {
        size_t mid = left + (right - left) / 2;
        if (arr[mid] < value) left = mid + 1; else right = mid;
    }
    return left;
}

void orc_ambiguity(const float* cv, int H, int W, int D, const float* etas, int nbr_etas, const int64_t* grid_min,
                   const int64_t* grid_max, const float* disp_range, float* amb) {
    size_t npix = (size_t)H * W;
    float* minimg = (float*)malloc(sizeof(float) * npix);
    float min_cost = INFINITY, max_cost = -INFINITY;
    for (size_t p = 0; p < npix; ++p) {
        float lo = INFINITY, hi = -INFINITY;
        int any = 0;
        for (int k = 0; k < D; ++k) {
            float v = cv[p * D + k];
            if (!isnan(v)) { any = 1; if (v < lo) lo = v; if (v > hi) hi = v; }
        }
        if (!any) { minimg[p] = NAN; continue; }
        minimg[p] = lo;
        if (lo < min_cost) min_cost = lo;
        if (hi > max_cost) max_cost = hi;
    }
    float diff = max_cost - min_cost;
    float* nc = (float*)malloc(sizeof(float) * (size_t)D);
    for (size_t p = 0; p < npix; ++p) {
        float ne = (minimg[p] - min_cost) / diff;
        if (isnan(ne)) { amb[p] = (float)(nbr_etas * D); continue; }
        size_t i0 = orc_searchsorted(disp_range, (size_t)D, (float)grid_min[p]);
        size_t i1 = orc_searchsorted(disp_range, (size_t)D, (float)grid_max[p]) + 1;
        for (int k = 0; k < D; ++k) {
            float v = cv[p * D + k];
            if (isnan(v)) nc[k] = ((size_t)k >= i0 && (size_t)k < i1) ? -INFINITY : INFINITY;
            else nc[k] = (v - min_cost) / diff;
        }
        float sum = 0;
        for (int e = 0; e < nbr_etas; ++e) {
            float s = 0, thr = ne + etas[e];
            for (int k = 0; k < D; ++k) s += (nc[k] <= thr) ? 1.f : 0.f;
            sum += s;
        }
        amb[p] = sum;
    }
    free(nc);
    free(minimg);
}


/* ---------------------------------------------------------------------------------------------
 * cost_volume_confidence/cpp/src/risk.cpp:28-197 compute_risk_and_sampled_risk as risk.py:144-166 calls it: the sampled
 * ambiguity comes from ambiguity.cpp (float32 etas, float comparison), the risk itself compares in double
 * (float cost > float extremum + double eta).  For every eta the span of disparity indices whose normalised cost is
 * within eta of the pixel's minimum: risk_max = mean span, risk_min = mean (1 + span - sampled ambiguity),
 * disp_sup / disp_inf = mean of the disparities at the two ends.  Pixels without any cost get NaN.
 * The cost volume is the one of a "min" measure (callers negate similarity volumes, risk.py:139-141).
 * ------------------------------------------------------------------------------------------- */
void orc_risk(const float* cv, int H, int W, int D, const double* etas, int nbr_etas, const int64_t* grid_min,
              const int64_t* grid_max, const float* disp_range, float* risk_max, float* risk_min, float* disp_sup,
              float* disp_inf) {
    size_t npix = (size_t)H * W;
    float* minimg = (float*)malloc(sizeof(float) * npix);
    float min_cost = INFINITY, max_cost = -INFINITY;
    for (size_t p = 0; p < npix; ++p) {
        float lo = INFINITY, hi = -INFINITY;
        int any = 0;
        for (int k = 0; k < D; ++k) {
            float v = cv[p * D + k];
            if (!isnan(v)) { any = 1; if (v < lo) lo = v; if (v > hi) hi = v; }
        }
        if (!any) { minimg[p] = NAN; continue; }
        minimg[p] = lo;
        if (lo < min_cost) min_cost = lo;
        if (hi > max_cost) max_cost = hi;
    }
    float diff = max_cost - min_cost;
    float* nc = (float*)malloc(sizeof(float) * (size_t)D);
    for (size_t p = 0; p < npix; ++p) {
        float ne = (minimg[p] - min_cost) / diff;
        if (isnan(ne)) { risk_max[p] = risk_min[p] = disp_sup[p] = disp_inf[p] = NAN; continue; }
        size_t i0 = orc_searchsorted(disp_range, (size_t)D, (float)grid_min[p]);
        size_t i1 = orc_searchsorted(disp_range, (size_t)D, (float)grid_max[p]) + 1;
        for (int k = 0; k < D; ++k) {
            float v = cv[p * D + k];
            if (isnan(v)) nc[k] = ((size_t)k >= i0 && (size_t)k < i1) ? -INFINITY : INFINITY;
            else nc[k] = (v - min_cost) / diff;
        }
        float s_min = 0, s_max = 0, s_inf = 0, s_sup = 0;
        for (int e = 0; e < nbr_etas; ++e) {
            /* ambiguity.cpp:120-128, float32 eta */
            float samp = 0, thr = ne + (float)etas[e];
            for (int k = 0; k < D; ++k) samp += (nc[k] <= thr) ? 1.f : 0.f;
            /* risk.cpp:137-151, double eta */
            double thr_d = (double)ne + etas[e];
            int lo_k = -1, hi_k = -1;
            for (int k = 0; k < D; ++k) {
                if ((double)nc[k] > thr_d) continue;
                if (lo_k < 0) lo_k = k;
                hi_k = k;
            }
            if (lo_k < 0) { lo_k = 0; hi_k = 0; } /* cannot happen for eta >= 0: the minimum always qualifies */
            float span = (float)hi_k - (float)lo_k;
            s_sup += disp_range[hi_k];
            s_inf += disp_range[lo_k];
            s_min += 1 + span - samp;
            s_max += span;
        }
        risk_min[p] = s_min / nbr_etas;
        risk_max[p] = s_max / nbr_etas;
        disp_sup[p] = s_sup / nbr_etas;
        disp_inf[p] = s_inf / nbr_etas;
    }
    free(nc);
    free(minimg);
}

/* ---------------------------------------------------------------------------------------------
 * cost_volume_confidence/cpp/src/interval_bounds.cpp:28-161 compute_interval_bounds: inside the pixel's
 * [grid_min, grid_max] the normalised costs become a possibility  type_factor * norm + 1 - max(type_factor * norm)
 * (type_factor -1 for "min" measures, +1 for "max"); the interval is the span of disparities whose possibility reaches
 * the threshold, widened by one sample on a side whose end has possibility exactly 1 (room for the refinement).
 * No cost in the range -> NaN bounds.  disp_interval: the D float32 disparities written as bounds (== disp_range in
 * interval_bounds.py:163-170).
 * ------------------------------------------------------------------------------------------- */
void orc_interval_bounds(const float* cv, int H, int W, int D, const float* disp_interval, float possibility_threshold,
                         float type_factor, const int64_t* grid_min, const int64_t* grid_max, const float* disp_range,
                         float* interval_inf, float* interval_sup) {
    size_t npix = (size_t)H * W;
    float min_cost = INFINITY, max_cost = -INFINITY;
    for (size_t i = 0; i < npix * (size_t)D; ++i) {
        float v = cv[i];
        if (!isnan(v)) { if (v < min_cost) min_cost = v; if (v > max_cost) max_cost = v; }
    }
    float diff = max_cost - min_cost;
    float* poss = (float*)malloc(sizeof(float) * (size_t)D);
    for (size_t p = 0; p < npix; ++p) {
        size_t i0 = orc_searchsorted(disp_range, (size_t)D, (float)grid_min[p]);
        size_t i1 = orc_searchsorted(disp_range, (size_t)D, (float)grid_max[p]) + 1;
        float max_pix = -INFINITY;
        for (size_t k = i0; k < i1; ++k) {
            float v = cv[p * D + k];
            poss[k] = (v - min_cost) / diff;
            if (!isnan(v)) { float t = type_factor * poss[k]; if (t > max_pix) max_pix = t; }
        }
        interval_inf[p] = interval_sup[p] = NAN;
        if (isinf(max_pix)) continue;
        int lo_k = -1, hi_k = -1;
        for (size_t k = i0; k < i1; ++k) {
            if (!isnan(poss[k])) {
                volatile float t = type_factor * poss[k];
                volatile float u = t + 1.f;
                poss[k] = u - max_pix;
            }
            if (poss[k] >= possibility_threshold) { if (lo_k < 0) lo_k = (int)k; hi_k = (int)k; }
        }
        if (lo_k < 0) continue;
        if (lo_k > 0 && (int)poss[lo_k] == 1) --lo_k;
        if (hi_k < D - 1 && (int)poss[hi_k] == 1) ++hi_k;
        interval_inf[p] = disp_interval[lo_k];
        interval_sup[p] = disp_interval[hi_k];
    }
    free(poss);
}


/* ---------------------------------------------------------------------------------------------
 * cpp/src/img_tools.cpp:27-155 interpolate_nodata_sgm (+ find_valid_neighbors, compute_median): every pixel whose mask
 * has a bit of `invalid_bits` becomes the median of the first valid pixels met along the 8 directions (paths that leave
 * the image contribute nothing; no valid neighbour at all -> NaN) and gets the mask value `filled_value`.
 * ------------------------------------------------------------------------------------------- */
void orc_interpolate_nodata(const float* img, const int32_t* msk, int H, int W, int invalid_bits, int filled_value,
                            float* out_img, int32_t* out_msk) {
    static const int dcol[8] = {0, -1, -1, -1, 0, 1, 1, 1}, drow[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            size_t i = (size_t)r * W + c;
            if (!(msk[i] & invalid_bits)) { out_img[i] = img[i]; out_msk[i] = msk[i]; continue; }
            float v[8];
            int n = 0;
            for (int d = 0; d < 8; ++d) {
                int rr = r + drow[d], cc = c + dcol[d];
                while (rr >= 0 && rr < H && cc >= 0 && cc < W) {
                    if (!(msk[(size_t)rr * W + cc] & invalid_bits)) {
                        float x = img[(size_t)rr * W + cc];
                        if (!isnan(x)) v[n++] = x;
                        break;
                    }
                    rr += drow[d];
                    cc += dcol[d];
                }
            }
            for (int a = 1; a < n; ++a) {
                float x = v[a];
                int b = a - 1;
                while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
                v[b + 1] = x;
            }
            out_img[i] = n == 0 ? NAN : ((n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.f);
            out_msk[i] = filled_value;
        }
}


/* ---------------------------------------------------------------------------------------------
 * validation/cpp/src/interpolated_disparity.cpp: the four gather passes of AbstractInterpolation ("mc-cnn" / "sgm").
 * All read the input maps only (outputs are separate), so every pixel is independent.
 * pass 0 = interpolate_occlusion_mc_cnn (:232-296), 1 = interpolate_mismatch_mc_cnn (:298-393),
 *      2 = interpolate_occlusion_sgm (:101-139),    3 = interpolate_mismatch_sgm (:166-230).
 * Mask constants: constants.py (INVALID 0x3C3, OCCLUSION 1<<8, MISMATCH 1<<9, FILLED_OCCLUSION 1<<4, FILLED_MISMATCH 1<<5).
 * ------------------------------------------------------------------------------------------- */
#define ORC_INVALID 0x3C3
#define ORC_OCC (1 << 8)
#define ORC_MIS (1 << 9)
#define ORC_FILLED_OCC (1 << 4)
#define ORC_FILLED_MIS (1 << 5)

static float orc_median_nan_free(float* v, int count) { /* compute_median :141-163 */
    int n = 0;
    for (int a = 0; a < count; ++a)
        if (!isnan(v[a])) v[n++] = v[a];
    if (n == 0) return NAN;
    for (int a = 1; a < n; ++a) {
        float x = v[a];
        int b = a - 1;
        while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
        v[b + 1] = x;
    }
    return (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.f;
}

/* find_valid_neighbors :28-75: (drow, dcol) in this order; leaving the image gives NaN */
static void orc_valid_neighbors8(const float* disp, const int32_t* valid, int H, int W, int r, int c, float* out) {
    static const int drow[8] = {0, -1, -1, -1, 0, 1, 1, 1}, dcol[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    for (int d = 0; d < 8; ++d) {
        int rr = r + drow[d], cc = c + dcol[d];
        out[d] = NAN;
        while (rr >= 0 && rr < H && cc >= 0 && cc < W) {
            if (!(valid[(size_t)rr * W + cc] & ORC_INVALID)) { out[d] = disp[(size_t)rr * W + cc]; break; }
            rr += drow[d];
            cc += dcol[d];
        }
    }
}

void orc_interpolate_disparity(int pass, const float* disp, const int32_t* valid, int H, int W, float* out_disp, int32_t* out_valid) {
    /* :318-335, (dcol, drow) pairs: the first factor is applied to the column (:347-348) */
    static const float d16[32] = {0.0f, 1.0f, -0.5f, 1.0f, -1.0f, 1.0f, -1.0f, 0.5f, -1.0f, 0.0f, -1.0f, -0.5f, -1.0f, -1.0f, -0.5f, -1.0f,
                                  0.0f, -1.0f, 0.5f, -1.0f, 1.0f, -1.0f, 1.0f, -0.5f, 1.0f, 0.0f, 1.0f, 0.5f, 1.0f, 1.0f, 0.5f, 1.0f};
    const int maxlen = H > W ? H : W;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const size_t i = (size_t)r * W + c;
            const int32_t m = valid[i];
            out_disp[i] = disp[i];
            out_valid[i] = m;
            if (pass == 0 && (m & ORC_OCC)) {
                int found = -1;
                for (int k = c; k >= 0 && found < 0; --k)
                    if (!(valid[(size_t)r * W + k] & ORC_INVALID)) found = k;
                for (int k = c; k < W && found < 0; ++k)
                    if (!(valid[(size_t)r * W + k] & ORC_INVALID)) found = k;
                if (found >= 0) { /* nothing valid on the row: the pixel keeps its value and its flag (:276-287 with msk == col) */
                    out_disp[i] = disp[(size_t)r * W + found];
                    out_valid[i] = m - ORC_OCC + ORC_FILLED_OCC;
                }
            } else if (pass == 1 && (m & ORC_MIS)) {
                float v[16];
                for (int d = 0; d < 16; ++d) {
                    v[d] = 0.f;
                    for (int k = 0; k < maxlen; ++k) {
                        const int cc = c + (int)(d16[2 * d] * (float)k), rr = r + (int)(d16[2 * d + 1] * (float)k);
                        if (rr < 0 || rr >= H || cc < 0 || cc >= W) { v[d] = NAN; break; }
                        if (!(valid[(size_t)rr * W + cc] & ORC_INVALID)) { v[d] = disp[(size_t)rr * W + cc]; break; }
                    }
                }
                out_disp[i] = orc_median_nan_free(v, 16);
                out_valid[i] = m + ORC_FILLED_MIS - ORC_MIS;
            } else if (pass == 2 && (m & ORC_OCC)) {
                float v[8];
                orc_valid_neighbors8(disp, valid, H, W, r, c, v);
                /* get_second_min_val_abs :77-99: the value of second smallest magnitude (strict <, first come first), +inf if < 2 */
                float mn = INFINITY, mna = INFINITY, sm = INFINITY, sma = INFINITY;
                for (int d = 0; d < 8; ++d) {
                    const float a = fabsf(v[d]);
                    if (a < mna) { sma = mna; sm = mn; mna = a; mn = v[d]; }
                    else if (a < sma) { sma = a; sm = v[d]; }
                }
                out_disp[i] = sm;
                out_valid[i] = m + ORC_FILLED_OCC - ORC_OCC;
            } else if (pass == 3 && (m & ORC_MIS)) {
                int near_occ = 0; /* a mismatch that touches an occlusion becomes an occlusion (:189-209) */
                for (int rr = (r > 0 ? r - 1 : 0); rr <= (r + 1 < H ? r + 1 : H - 1); ++rr)
                    for (int cc = (c > 0 ? c - 1 : 0); cc <= (c + 1 < W ? c + 1 : W - 1); ++cc)
                        near_occ |= (valid[(size_t)rr * W + cc] & ORC_OCC) != 0;
                if (near_occ) { out_valid[i] = m - ORC_MIS + ORC_OCC; continue; }
                float v[8];
                orc_valid_neighbors8(disp, valid, H, W, r, c, v);
                out_disp[i] = orc_median_nan_free(v, 8);
                out_valid[i] = m + ORC_FILLED_MIS - ORC_MIS;
            }
        }
}

/*
 * oracle/loft_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the three mmcv==1.0.5 native ops that sit on the
 * reference's LOFT hot path (the reference pins mmcv to 1.0.5 in
 * mmdet/__init__.py:18-26; the mmcv sources are NOT in /root/reference):
 *
 *   RoIAlign  fwd/bwd  -- call sites roi_extractors/base_roi_extractor.py:49-54,
 *                         single_level_roi_extractor.py:67-76, core/mask/structures.py:286-287
 *   nms               -- via batched_nms, dense_heads/rpn_head.py:166-168
 *   soft_nms (linear) -- via batched_nms, core/post_processing/bbox_nms.py:63
 *
 * PARITY UNPINNED at this boundary: the reference tree holds no value-level
 * test for these ops (tests/test_masks.py:196-216 is shape-only), so this file
 * is the project's *definition* of their arithmetic, restated from the
 * published mmcv-1.0.5 algorithms.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/build_oracle.py).
 * -ffp-contract=off matters: the NMS suppression predicate must round exactly
 * like the HIP kernel's (compiled with the same flag) for bit-exact keep lists.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ RoIAlign */

typedef struct {
    int y_low, x_low, y_high, x_high;
    float w1, w2, w3, w4;
    int valid;
} bil_t;

/* mmcv roi_align bilinear_interpolate: outside (-1, size) -> 0; clamp to [0, size-1]. */
static bil_t bil_setup(float y, float x, int height, int width) {
    bil_t b;
    memset(&b, 0, sizeof(b));
    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
        b.valid = 0;
        return b;
    }
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= width - 1)  { x_high = x_low = width - 1;  x = (float)x_low; } else x_high = x_low + 1;
    float ly = y - (float)y_low, lx = x - (float)x_low;
    float hy = 1.f - ly, hx = 1.f - lx;
    b.y_low = y_low; b.x_low = x_low; b.y_high = y_high; b.x_high = x_high;
    b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
    b.valid = 1;
    return b;
}

typedef struct {
    float start_h, start_w, bin_h, bin_w;
    int grid_h, grid_w;
    float count;
    int batch;
} roi_geom_t;

static roi_geom_t roi_geom(const float* roi, float scale, int ph, int pw, int sampling_ratio, int aligned) {
    roi_geom_t g;
    float off = aligned ? 0.5f : 0.f;
    g.batch = (int)roi[0];
    g.start_w = roi[1] * scale - off;
    g.start_h = roi[2] * scale - off;
    float end_w = roi[3] * scale - off;
    float end_h = roi[4] * scale - off;
    float rw = end_w - g.start_w, rh = end_h - g.start_h;
    if (!aligned) { rw = rw > 1.f ? rw : 1.f; rh = rh > 1.f ? rh : 1.f; }
    g.bin_h = rh / (float)ph;
    g.bin_w = rw / (float)pw;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
    int c = g.grid_h * g.grid_w;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}

/* feat: [N,C,H,W] fp32 (NCHW, as the reference holds it); rois [K,5]; out [K,C,ph,pw]. */
void orc_roi_align_fwd(const float* feat, int N, int C, int H, int W, const float* rois, int K,
                       int ph, int pw, float scale, int sampling_ratio, int aligned, float* out) {
    (void)N;
    for (int k = 0; k < K; ++k) {
        roi_geom_t g = roi_geom(rois + 5 * k, scale, ph, pw, sampling_ratio, aligned);
        const float* fb = feat + (size_t)g.batch * C * H * W;
        for (int py = 0; py < ph; ++py)
            for (int px = 0; px < pw; ++px) {
                float* o = out + (((size_t)k * C) * ph + py) * pw + px;
                for (int c = 0; c < C; ++c) o[(size_t)c * ph * pw] = 0.f;
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    float y = g.start_h + py * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        float x = g.start_w + px * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                        bil_t b = bil_setup(y, x, H, W);
                        if (!b.valid) continue;
                        for (int c = 0; c < C; ++c) {
                            const float* f = fb + (size_t)c * H * W;
                            float v = b.w1 * f[b.y_low * W + b.x_low] + b.w2 * f[b.y_low * W + b.x_high] +
                                      b.w3 * f[b.y_high * W + b.x_low] + b.w4 * f[b.y_high * W + b.x_high];
                            o[(size_t)c * ph * pw] += v;
                        }
                    }
                }
                for (int c = 0; c < C; ++c) o[(size_t)c * ph * pw] /= g.count;
            }
    }
}

/* grad_out [K,C,ph,pw] -> grad_in [N,C,H,W] (accumulated; caller zeroes). */
void orc_roi_align_bwd(const float* grad_out, int N, int C, int H, int W, const float* rois, int K,
                       int ph, int pw, float scale, int sampling_ratio, int aligned, float* grad_in) {
    (void)N;
    for (int k = 0; k < K; ++k) {
        roi_geom_t g = roi_geom(rois + 5 * k, scale, ph, pw, sampling_ratio, aligned);
        float* gb = grad_in + (size_t)g.batch * C * H * W;
        for (int py = 0; py < ph; ++py)
            for (int px = 0; px < pw; ++px) {
                const float* go = grad_out + (((size_t)k * C) * ph + py) * pw + px;
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    float y = g.start_h + py * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        float x = g.start_w + px * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                        bil_t b = bil_setup(y, x, H, W);
                        if (!b.valid) continue;
                        for (int c = 0; c < C; ++c) {
                            float gv = go[(size_t)c * ph * pw] / g.count;
                            float* f = gb + (size_t)c * H * W;
                            f[b.y_low * W + b.x_low] += gv * b.w1;
                            f[b.y_low * W + b.x_high] += gv * b.w2;
                            f[b.y_high * W + b.x_low] += gv * b.w3;
                            f[b.y_high * W + b.x_high] += gv * b.w4;
                        }
                    }
                }
            }
    }
}

/* ------------------------------------------------------------------ NMS */

typedef struct { float s; int64_t i; } sc_t;
static int sc_cmp(const void* a, const void* b) {
    const sc_t* x = (const sc_t*)a; const sc_t* y = (const sc_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0); /* ties: lower original index first */
}

/* Project-defined total order for "sort by score, descending": score desc, index asc. */
void orc_argsort_desc(const float* scores, int64_t n, int64_t* order) {
    sc_t* t = (sc_t*)malloc(sizeof(sc_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { t[i].s = scores[i]; t[i].i = i; }
    qsort(t, (size_t)n, sizeof(sc_t), sc_cmp);
    for (int64_t i = 0; i < n; ++i) order[i] = t[i].i;
    free(t);
}

/* Suppression predicate of the mmcv-1.0.5 device kernel (offset = 0):
 * inter > thr * (Sa + Sb - inter), no division.  Every operation is a single
 * IEEE fp32 op (no FMA) so the HIP kernel reproduces it bit-for-bit. */
static int iou_gt(const float* a, const float* b, float thr) {
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    float sa = (a[2] - a[0]) * (a[3] - a[1]);
    float sb = (b[2] - b[0]) * (b[3] - b[1]);
    float uni = sa + sb - inter;
    return inter > thr * uni;
}

/* Suppression predicate of the mmcv-1.0.5 HOST kernel (nms_cpu, offset = 0): ovr = inter / (Sa + Sb - inter); ovr >= thr.
 * [mmcv-1.0.5, not in tree: restated from the published source; parity unpinned] */
static int iou_ge_div(const float* a, const float* b, float thr) {
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    float inter = w * h;
    float sa = (a[2] - a[0]) * (a[3] - a[1]);
    float sb = (b[2] - b[0]) * (b[3] - b[1]);
    float uni = sa + sb - inter;
    return inter / uni >= thr;
}

int64_t orc_nms_pred(const float* boxes, const float* scores, int64_t n, float thr, int pred, int64_t* keep);
/* Greedy NMS.  keep[] receives original indices in score-descending order; returns count. */
int64_t orc_nms(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep) {
    return orc_nms_pred(boxes, scores, n, thr, 0, keep);
}
/* pred 0: device predicate (inter > thr*union); pred 1: host predicate (inter/union >= thr). */
int64_t orc_nms_pred(const float* boxes, const float* scores, int64_t n, float thr, int pred, int64_t* keep) {
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    unsigned char* dead = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
    orc_argsort_desc(scores, n, order);
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        if (dead[_i]) continue;
        int64_t i = order[_i];
        keep[nk++] = i;
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            if (dead[_j]) continue;
            if (pred ? iou_ge_div(boxes + 4 * i, boxes + 4 * order[_j], thr) : iou_gt(boxes + 4 * i, boxes + 4 * order[_j], thr)) dead[_j] = 1;
        }
    }
    free(order); free(dead);
    return nk;
}

/* Linear / naive / gaussian soft-NMS, in-place max-selection order (mmcv-1.0.5 CPU op;
 * this op is CPU-only in that release).  dets [n,5], inds [n]; returns kept count. */
int64_t orc_soft_nms(const float* boxes, const float* scores, int64_t n, float iou_thr, float sigma,
                     float min_score, int method, float* dets, int64_t* inds) {
    size_t m = (size_t)(n > 0 ? n : 1);
    float* x1 = malloc(4 * m), *y1 = malloc(4 * m), *x2 = malloc(4 * m), *y2 = malloc(4 * m);
    float* sc = malloc(4 * m), *ar = malloc(4 * m);
    for (int64_t i = 0; i < n; ++i) {
        x1[i] = boxes[4 * i]; y1[i] = boxes[4 * i + 1]; x2[i] = boxes[4 * i + 2]; y2[i] = boxes[4 * i + 3];
        sc[i] = scores[i]; ar[i] = (x2[i] - x1[i]) * (y2[i] - y1[i]); inds[i] = i;
    }
    int64_t nb = n;
    for (int64_t i = 0; i < nb; ++i) {
        float max_score = sc[i]; int64_t max_pos = i;
        for (int64_t pos = i + 1; pos < nb; ++pos)
            if (max_score < sc[pos]) { max_score = sc[pos]; max_pos = pos; }
        float ix1 = x1[max_pos], iy1 = y1[max_pos], ix2 = x2[max_pos], iy2 = y2[max_pos];
        float isc = sc[max_pos], iar = ar[max_pos]; int64_t iind = inds[max_pos];
        dets[5 * i] = ix1; dets[5 * i + 1] = iy1; dets[5 * i + 2] = ix2; dets[5 * i + 3] = iy2; dets[5 * i + 4] = isc;
        x1[max_pos] = x1[i]; y1[max_pos] = y1[i]; x2[max_pos] = x2[i]; y2[max_pos] = y2[i];
        sc[max_pos] = sc[i]; ar[max_pos] = ar[i]; inds[max_pos] = inds[i];
        x1[i] = ix1; y1[i] = iy1; x2[i] = ix2; y2[i] = iy2; sc[i] = isc; ar[i] = iar; inds[i] = iind;
        int64_t pos = i + 1;
        while (pos < nb) {
            float xx1 = fmaxf(ix1, x1[pos]), yy1 = fmaxf(iy1, y1[pos]);
            float xx2 = fminf(ix2, x2[pos]), yy2 = fminf(iy2, y2[pos]);
            float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
            float inter = w * h;
            float ovr = inter / (iar + ar[pos] - inter);
            float weight = 1.f;
            if (method == 0) { if (ovr >= iou_thr) weight = 0.f; }
            else if (method == 1) { if (ovr >= iou_thr) weight = 1.f - ovr; }
            else { weight = expf(-(ovr * ovr) / sigma); }
            sc[pos] *= weight;
            if (sc[pos] < min_score) {
                x1[pos] = x1[nb - 1]; y1[pos] = y1[nb - 1]; x2[pos] = x2[nb - 1]; y2[pos] = y2[nb - 1];
                sc[pos] = sc[nb - 1]; ar[pos] = ar[nb - 1]; inds[pos] = inds[nb - 1];
                nb -= 1; pos -= 1;
            }
            pos += 1;
        }
    }
    free(x1); free(y1); free(x2); free(y2); free(sc); free(ar);
    return nb;
}

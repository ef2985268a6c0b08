/* ORACLE (test infrastructure only) -- plain-C restatement of the reference's instance post-processing.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load this file; the product
 * (cerberus_amd/) never does.
 *
 * Restates reference loader/postproc.py:268-407 (PostProcInstErodedContourMap) and the third-party routines it
 * calls, none of which are vendored under /root/reference:
 *   scipy.ndimage.label (default structure = 4-connectivity)           postproc.py:288,329,366,372,377
 *   skimage.morphology.remove_small_objects                            postproc.py:287,328,367,373
 *   scipy.ndimage.binary_fill_holes                                    postproc.py:304,345,376
 *   skimage.segmentation.watershed(image, markers, mask=...)           postproc.py:378
 *       (scikit-image 0.19.2 pinned by environment.yml:22; priority flood of _watershed_cy.pyx with the binary
 *        heap of _shared/heap_general.pxi -- restated from the published algorithm, incl. the heap's tie behaviour)
 *   cv2.getStructuringElement(MORPH_ELLIPSE) / cv2.erode / cv2.dilate  postproc.py:275,303,317,344,356,365
 *       (opencv-python 4.6.0.66 pinned by environment.yml:31; restated from OpenCV's documented semantics:
 *        row-span ellipse formula, anchor = ksize/2, constant border that never wins the min/max)
 *   misc/utils.py:82-91 get_bounding_box
 *
 * Pinning: oracle/gen_golden_postproc.py runs the reference's own post_process under /opt/conda python3.9 with the
 * REAL scikit-image 0.18.3 + scipy 1.7.1 and a cv2 stand-in (OpenCV is not installed anywhere in the container), and
 * checks this file against it on every fixture in tests/golden/pp_*.npz.  The cv2 pieces are therefore "parity
 * unpinned" against the real library (stated in DESIGN.md).
 *
 * Build: make -C oracle   ->  oracle/libpostproc_ref.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------ */
/* scipy.ndimage.label, 4-connectivity: ids in raster order of each component's first pixel.                      */
int ref_label4(const uint8_t* in, int H, int W, int32_t* out) {
    const long n = (long)H * W;
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    memset(out, 0, sizeof(int32_t) * n);
    int next = 0;
    for (long p = 0; p < n; ++p) {
        if (!in[p] || out[p]) continue;
        ++next;
        long sp = 0;
        stack[sp++] = (int32_t)p;
        out[p] = next;
        while (sp) {
            const long q = stack[--sp];
            const int y = (int)(q / W), x = (int)(q % W);
            if (y > 0 && in[q - W] && !out[q - W]) { out[q - W] = next; stack[sp++] = (int32_t)(q - W); }
            if (x > 0 && in[q - 1] && !out[q - 1]) { out[q - 1] = next; stack[sp++] = (int32_t)(q - 1); }
            if (x < W - 1 && in[q + 1] && !out[q + 1]) { out[q + 1] = next; stack[sp++] = (int32_t)(q + 1); }
            if (y < H - 1 && in[q + W] && !out[q + W]) { out[q + W] = next; stack[sp++] = (int32_t)(q + W); }
        }
    }
    free(stack);
    return next;
}

/* skimage.morphology.remove_small_objects on a label image: zero components with area < min_size, no renumbering. */
void ref_remove_small_labels(int32_t* lab, int nlab, long n, int min_size) {
    if (min_size == 0) return; /* skimage shortcut */
    long* cnt = (long*)calloc((size_t)nlab + 1, sizeof(long));
    for (long p = 0; p < n; ++p) cnt[lab[p]]++;
    for (long p = 0; p < n; ++p)
        if (lab[p] && cnt[lab[p]] < min_size) lab[p] = 0;
    free(cnt);
}

/* remove_small_objects on a bool image (labels internally with connectivity 1). */
void ref_remove_small_bool(uint8_t* m, int H, int W, int min_size) {
    const long n = (long)H * W;
    int32_t* lab = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    const int nl = ref_label4(m, H, W, lab);
    ref_remove_small_labels(lab, nl, n, min_size);
    for (long p = 0; p < n; ++p) m[p] = lab[p] != 0;
    free(lab);
}

/* cv2.getStructuringElement(MORPH_ELLIPSE, (k,k)): returns row spans [j1,j2) per row. */
void ref_ellipse_spans(int k, int* j1, int* j2) {
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; ++i) {
        const int dy = i - r;
        j1[i] = j2[i] = 0;
        if (abs(dy) <= r) {
            const int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2)); /* saturate_cast<int> == round-half-even */
            j1[i] = c - dx > 0 ? c - dx : 0;
            j2[i] = c + dx + 1 < k ? c + dx + 1 : k;
        }
    }
}

/* cv2.dilate(src u8, ellipse k x k, anchor (k/2,k/2), border never wins).  dst(y,x) = max src(y+i-a, x+j-a). */
void ref_dilate_ellipse(const uint8_t* src, int H, int W, int k, uint8_t* dst) {
    int j1[64], j2[64];
    const int a = k / 2;
    ref_ellipse_spans(k, j1, j2);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t v = 0;
            for (int i = 0; i < k && !v; ++i) {
                const int yy = y + i - a;
                if (yy < 0 || yy >= H) continue;
                for (int j = j1[i]; j < j2[i]; ++j) {
                    const int xx = x + j - a;
                    if (xx < 0 || xx >= W) continue;
                    if (src[(long)yy * W + xx]) { v = 1; break; }
                }
            }
            dst[(long)y * W + x] = v;
        }
}

/* cv2.erode with the 3x3 MORPH_ELLIPSE (= cross), border never wins the min. */
void ref_erode_cross3(const uint8_t* src, int H, int W, uint8_t* dst) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const long p = (long)y * W + x;
            uint8_t v = src[p];
            if (y > 0) v &= src[p - W];
            if (y < H - 1) v &= src[p + W];
            if (x > 0) v &= src[p - 1];
            if (x < W - 1) v &= src[p + 1];
            dst[p] = v;
        }
}

/* scipy.ndimage.binary_fill_holes (default 4-connected structure): background not connected to the border. */
void ref_fill_holes(uint8_t* m, int H, int W) {
    const long n = (long)H * W;
    if (n == 0) return;
    uint8_t* reach = (uint8_t*)calloc(n, 1);
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * n);
    long sp = 0;
#define SEED(P) do { const long p_ = (P); if (!m[p_] && !reach[p_]) { reach[p_] = 1; stack[sp++] = (int32_t)p_; } } while (0)
    for (int x = 0; x < W; ++x) { SEED(x); SEED((long)(H - 1) * W + x); }
    for (int y = 0; y < H; ++y) { SEED((long)y * W); SEED((long)y * W + W - 1); }
    while (sp) {
        const long q = stack[--sp];
        const int y = (int)(q / W), x = (int)(q % W);
        if (y > 0) SEED(q - W);
        if (x > 0) SEED(q - 1);
        if (x < W - 1) SEED(q + 1);
        if (y < H - 1) SEED(q + W);
    }
#undef SEED
    for (long p = 0; p < n; ++p) m[p] = m[p] || !reach[p];
    free(reach);
    free(stack);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* skimage.segmentation.watershed(image, markers, connectivity=1, mask=mask), compactness 0, no watershed line.   */
typedef struct { double value; int32_t age; int32_t index; int32_t source; } HItem;
typedef struct { HItem* a; long n, cap; } Heap;
static int smaller(const HItem* x, const HItem* y) {
    if (x->value != y->value) return x->value < y->value;
    return x->age < y->age;
}
static void hpush(Heap* h, HItem it) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 1024; h->a = (HItem*)realloc(h->a, sizeof(HItem) * h->cap); }
    long child = h->n++;
    h->a[child] = it;
    while (child > 0) {
        const long parent = (child + 1) / 2 - 1;
        if (smaller(&h->a[child], &h->a[parent])) { HItem t = h->a[child]; h->a[child] = h->a[parent]; h->a[parent] = t; child = parent; }
        else break;
    }
}
static HItem hpop(Heap* h) {
    HItem top = h->a[0];
    h->n--;
    if (h->n == 0) return top;
    h->a[0] = h->a[h->n];
    long i = 0, smallest = 0;
    for (;;) {
        const long l = 2 * i + 1, r = 2 * i + 2;
        if (l < h->n) {
            if (smaller(&h->a[l], &h->a[i])) smallest = l;
            if (r < h->n && smaller(&h->a[r], &h->a[smallest])) smallest = r;
        } else break;
        if (smallest == i) break;
        HItem t = h->a[i]; h->a[i] = h->a[smallest]; h->a[smallest] = t;
        i = smallest;
    }
    return top;
}

void ref_watershed(const float* image, const int32_t* markers, const uint8_t* mask, int H, int W, int32_t* out) {
    const int PW = W + 2, PH = H + 2;
    const long pn = (long)PH * PW;
    double* img = (double*)calloc(pn, sizeof(double));
    uint8_t* msk = (uint8_t*)calloc(pn, 1);
    int32_t* o = (int32_t*)calloc(pn, sizeof(int32_t));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const long p = (long)y * W + x, q = (long)(y + 1) * PW + x + 1;
            img[q] = (double)image[p];
            msk[q] = mask[p] != 0;
            o[q] = mask[p] ? markers[p] : 0; /* markers * mask (_validate_inputs) */
        }
    const long nb[4] = {-PW, -1, 1, PW}; /* _offsets_to_raveled_neighbors order for connectivity 1 */
    Heap h = {0, 0, 0};
    for (long q = 0; q < pn; ++q)
        if (o[q]) { HItem it = {img[q], 0, (int32_t)q, (int32_t)q}; hpush(&h, it); }
    int32_t age = 0;
    while (h.n > 0) {
        const HItem e = hpop(&h);
        for (int i = 0; i < 4; ++i) {
            const long nq = e.index + nb[i];
            if (!msk[nq]) continue;
            if (o[nq]) continue;
            age += 1;
            o[nq] = o[e.index];
            HItem ne = {img[nq], age, (int32_t)nq, e.source};
            hpush(&h, ne);
        }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) out[(long)y * W + x] = o[(long)(y + 1) * PW + x + 1];
    free(h.a); free(img); free(msk); free(o);
}

/* ------------------------------------------------------------------------------------------------------------ */
/* __proc_nuclei  (postproc.py:352-381).  inst: [H][W][2] float32.  Returns 1 when the `np.sum(inst_msk) > 0`
 * branch ran (int32 labels) and 0 for the all-zero float64 branch.                                               */
int ref_proc_nuclei(const float* inst, int H, int W, int32_t* out) {
    const long n = (long)H * W;
    uint8_t* msk = (uint8_t*)malloc(n ? n : 1);
    uint8_t* tmp = (uint8_t*)malloc(n ? n : 1);
    int32_t* lab = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    float* neg = (float*)malloc(sizeof(float) * (n ? n : 1));
    long any = 0;
    for (long p = 0; p < n; ++p) {
        const float raw = inst[2 * p] + inst[2 * p + 1]; /* float32 add, as numpy */
        msk[p] = raw > 0.5f;
        any += msk[p];
    }
    memset(out, 0, sizeof(int32_t) * n);
    if (any > 0) {
        ref_erode_cross3(msk, H, W, tmp);
        int nl = ref_label4(tmp, H, W, lab);
        ref_remove_small_labels(lab, nl, n, 8);
        for (long p = 0; p < n; ++p) msk[p] = lab[p] > 0;
        for (long p = 0; p < n; ++p) tmp[p] = inst[2 * p] > 0.5f;
        nl = ref_label4(tmp, H, W, lab);
        ref_remove_small_labels(lab, nl, n, 4);
        for (long p = 0; p < n; ++p) tmp[p] = lab[p] != 0;
        ref_fill_holes(tmp, H, W);
        ref_label4(tmp, H, W, lab);
        for (long p = 0; p < n; ++p) neg[p] = -inst[2 * p];
        ref_watershed(neg, lab, msk, H, W, out);
    }
    free(msk); free(tmp); free(lab); free(neg);
    return any > 0;
}

/* __proc_gland / __proc_lumen (postproc.py:269-350): thr = 0.55 / 0.5, min_size = int(1000*ds^2) / int(150*ds^2),
 * ksize = int(10*ds) / int(2*ds).  Output ids as int32 (the reference holds them in a float64 canvas).            */
static int proc_eroded_contour(const float* inst, int H, int W, float thr, int min_size, int ksize, int32_t* out) {
    const long n = (long)H * W;
    uint8_t* fg = (uint8_t*)malloc(n ? n : 1);
    int32_t* lab = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    for (long p = 0; p < n; ++p) {
        const float c = inst[2 * p + 1] > 0.5f ? 1.f : 0.f;
        fg[p] = (inst[2 * p] - c) > thr;
    }
    ref_remove_small_bool(fg, H, W, min_size);
    const int nl = ref_label4(fg, H, W, lab);
    memset(out, 0, sizeof(int32_t) * n);
    /* bounding boxes (misc/utils.py:82-91) */
    int* y1 = (int*)malloc(sizeof(int) * (nl + 1)); int* y2 = (int*)malloc(sizeof(int) * (nl + 1));
    int* x1 = (int*)malloc(sizeof(int) * (nl + 1)); int* x2 = (int*)malloc(sizeof(int) * (nl + 1));
    for (int i = 0; i <= nl; ++i) { y1[i] = H; x1[i] = W; y2[i] = 0; x2[i] = 0; }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int l = lab[(long)y * W + x];
            if (!l) continue;
            if (y < y1[l]) y1[l] = y;
            if (y + 1 > y2[l]) y2[l] = y + 1;
            if (x < x1[l]) x1[l] = x;
            if (x + 1 > x2[l]) x2[l] = x + 1;
        }
    const int pad = ksize * 2;
    for (int id = 1; id <= nl; ++id) {
        int ya = y1[id], yb = y2[id], xa = x1[id], xb = x2[id];
        ya = ya - pad >= 0 ? ya - pad : ya;
        xa = xa - pad >= 0 ? xa - pad : xa;
        xb = xb + pad <= W - 1 ? xb + pad : xb;
        yb = yb + pad <= H - 1 ? yb + pad : yb;
        const int ch = yb - ya, cw = xb - xa;
        uint8_t* crop = (uint8_t*)malloc((size_t)ch * cw);
        uint8_t* dil = (uint8_t*)malloc((size_t)ch * cw);
        for (int y = 0; y < ch; ++y)
            for (int x = 0; x < cw; ++x) crop[(long)y * cw + x] = lab[(long)(ya + y) * W + xa + x] == id;
        ref_dilate_ellipse(crop, ch, cw, ksize, dil);
        ref_fill_holes(dil, ch, cw);
        for (int y = 0; y < ch; ++y)
            for (int x = 0; x < cw; ++x)
                if (dil[(long)y * cw + x]) out[(long)(ya + y) * W + xa + x] = id;
        free(crop); free(dil);
    }
    free(y1); free(y2); free(x1); free(x2); free(fg); free(lab);
    return nl;
}
int ref_proc_gland(const float* inst, int H, int W, float ds, int32_t* out) {
    /* python: ksize = int((11-1)*ds); min_size = int(1000*(ds**2)) evaluated in double precision */
    return proc_eroded_contour(inst, H, W, 0.55f, (int)(1000.0 * ((double)ds * (double)ds)), (int)(10.0 * (double)ds), out);
}
int ref_proc_lumen(const float* inst, int H, int W, float ds, int32_t* out) {
    return proc_eroded_contour(inst, H, W, 0.5f, (int)(150.0 * ((double)ds * (double)ds)), (int)(2.0 * (double)ds), out);
}

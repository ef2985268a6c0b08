/* Developer tool (test infrastructure, CPU only): validates the RULES by which the GPU flood decides that a nuclei map's
 * watershed result does not depend on skimage's heap-layout tie order.
 *
 * sim_watershed floods every 4-connected mask component on its own with the total order (value, age, pixel index) -- what
 * the HIP kernels of cerberus_amd/csrc/postproc.hip do -- and raises `ambiguous` by the same rules; the checker
 * (tests/tools/dev_tie_rule_fuzz.py) compares it with the oracle's literal global binary heap (oracle/postproc_ref.c) on
 * tie-heavy maps: every map with ambiguous == 0 must be identical.
 *
 * Build: gcc -O2 -shared -fPIC -o tests/tools/_tie_rule_sim.so tests/tools/tie_rule_sim.c -lm
 */
#include "../../oracle/postproc_ref.c"

typedef struct { uint32_t val; uint32_t age; uint32_t idx; uint32_t unc; } SItem;
static uint32_t okey(float v) {
    if (v == 0.0f) v = 0.0f;
    uint32_t b;
    memcpy(&b, &v, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
static int sless(const SItem* a, const SItem* b) {
    if (a->val != b->val) return a->val < b->val;
    if (a->age != b->age) return a->age < b->age;
    return a->idx < b->idx;
}
typedef struct { SItem* a; long n, cap; } SHeap;
static void spush(SHeap* h, SItem it) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 1024; h->a = (SItem*)realloc(h->a, sizeof(SItem) * h->cap); }
    long c = h->n++;
    h->a[c] = it;
    while (c > 0) {
        long p = (c - 1) / 2;
        if (sless(&h->a[c], &h->a[p])) { SItem t = h->a[c]; h->a[c] = h->a[p]; h->a[p] = t; c = p; } else break;
    }
}
static SItem spop(SHeap* h) {
    SItem top = h->a[0];
    h->n--;
    if (h->n > 0) {
        h->a[0] = h->a[h->n];
        long i = 0;
        for (;;) {
            long l = 2 * i + 1, r = l + 1, s = i;
            if (l < h->n && sless(&h->a[l], &h->a[s])) s = l;
            if (r < h->n && sless(&h->a[r], &h->a[s])) s = r;
            if (s == i) break;
            SItem t = h->a[i]; h->a[i] = h->a[s]; h->a[s] = t;
            i = s;
        }
    }
    return top;
}

/* rule_mode 0: the round-1 rule (consecutive seed pops with equal value and different labels)
 * rule_mode 1: uncertain-age propagation + direct-conflict rule (round 2) */
int sim_watershed(const float* image, const int32_t* markers, const uint8_t* mask, int H, int W, int32_t* out, int rule_mode) {
    const long n = (long)H * W;
    int32_t* comp = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    const int nc = ref_label4(mask, H, W, comp);
    for (long p = 0; p < n; ++p) out[p] = mask[p] ? markers[p] : 0;
    /* bucket the seeds per component: active seeds only (labelled pixel with an unlabelled in-mask neighbour) */
    long* cnt = (long*)calloc((size_t)nc + 2, sizeof(long));
    for (long p = 0; p < n; ++p)
        if (out[p]) cnt[comp[p] + 1]++;
    for (int c = 1; c <= nc + 1; ++c) cnt[c] += cnt[c - 1];
    int32_t* seeds = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    long* fill = (long*)calloc((size_t)nc + 2, sizeof(long));
    for (long p = 0; p < n; ++p)
        if (out[p]) seeds[cnt[comp[p]] + fill[comp[p]]++] = (int32_t)p;
    int ambiguous = 0;
    SHeap h = {0, 0, 0};
    const int dy[4] = {-1, 0, 0, 1}, dx[4] = {0, -1, 1, 0};
    for (int c = 1; c <= nc; ++c) {
        h.n = 0;
        for (long s = cnt[c]; s < cnt[c] + fill[c]; ++s) {
            const long p = seeds[s];
            const int y = (int)(p / W), x = (int)(p % W);
            int active = 0;
            for (int k = 0; k < 4; ++k) {
                const int yy = y + dy[k], xx = x + dx[k];
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const long q = (long)yy * W + xx;
                if (mask[q] && !out[q]) active = 1;
            }
            if (!active) continue;
            SItem it = {okey(image[p]), 0, (uint32_t)p, 1};
            spush(&h, it);
        }
        uint32_t age = 0, prev_val = 0;
        int have_prev = 0, prev_lab = 0, have_prev_seed = 0, comp_amb = 0;
        uint32_t prev_seed_val = 0;
        while (h.n > 0) {
            const SItem e = spop(&h);
            const int lab = out[e.idx];
            if (rule_mode == 0 && e.age == 0) {
                if (have_prev_seed && e.val == prev_seed_val && lab != prev_lab) comp_amb = 1;
                have_prev_seed = 1; prev_seed_val = e.val; prev_lab = lab;
            }
            const int tied = (have_prev && prev_val == e.val) || (h.n > 0 && h.a[0].val == e.val);
            const int unc = e.unc && tied;
            have_prev = 1; prev_val = e.val;
            const int y = (int)(e.idx / W), x = (int)(e.idx % W);
            for (int k = 0; k < 4; ++k) {
                const int yy = y + dy[k], xx = x + dx[k];
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const long q = (long)yy * W + xx;
                if (!mask[q] || out[q]) continue;
                out[q] = lab;
                ++age;
                SItem ne = {okey(image[q]), age, (uint32_t)q, (uint32_t)unc};
                if (rule_mode == 1 && unc) {
                    if (ne.val < e.val) comp_amb = 1;                      /* (c) a child that overtakes the rest of the tied run */
                    for (int j = 0; j < 4; ++j) {                          /* (a) a tied pixel of another label touches q        */
                        const int y2 = yy + dy[j], x2 = xx + dx[j];
                        if (y2 < 0 || y2 >= H || x2 < 0 || x2 >= W) continue;
                        const long w = (long)y2 * W + x2;
                        if (out[w] && out[w] != lab && okey(image[w]) == e.val) comp_amb = 1;
                    }
                }
                spush(&h, ne);
            }
        }
        ambiguous += comp_amb;
    }
    free(h.a); free(comp); free(cnt); free(seeds); free(fill);
    return ambiguous;
}

/* __proc_nuclei with the simulated flood; returns the ambiguity count through *n_amb */
int sim_proc_nuclei(const float* inst, int H, int W, int32_t* out, int rule_mode, int* n_amb) {
    const long n = (long)H * W;
    uint8_t* msk = (uint8_t*)malloc(n ? n : 1);
    uint8_t* tmp = (uint8_t*)malloc(n ? n : 1);
    int32_t* lab = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    float* neg = (float*)malloc(sizeof(float) * (n ? n : 1));
    long any = 0;
    for (long p = 0; p < n; ++p) {
        const float raw = inst[2 * p] + inst[2 * p + 1];
        msk[p] = raw > 0.5f;
        any += msk[p];
    }
    memset(out, 0, sizeof(int32_t) * n);
    *n_amb = 0;
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
        *n_amb = sim_watershed(neg, lab, msk, H, W, out, rule_mode);
    }
    free(msk); free(tmp); free(lab); free(neg);
    return any > 0;
}

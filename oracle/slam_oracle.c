/*
 * slam_oracle.c — CPU restatement of the reference pipeline rows A1-A8.
 * TEST INFRASTRUCTURE — see slam_oracle.h for the rules on who may use it.
 * Written from the behavioural description in SURVEY.md §3/§8/Appendix A; every block cites
 * the reference lines (relative to /root/reference) whose arithmetic it must reproduce.
 */
#define _POSIX_C_SOURCE 200809L
#include "slam_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ A1 */

int orc_parse_frame(FILE *f, float *ranges, int nbeams)
{
    /* Subsystem_1/main.c:26-29 — "%f," per beam; a failed conversion leaves the slot's
     * previous content in place (the reference re-stores its stale local `value`; for a
     * well-formed file the two are indistinguishable, at EOF ours keeps the old frame). */
    int got = 0;
    for (int k = 0; k < nbeams; ++k) {
        float v;
        if (fscanf(f, "%f,", &v) == 1) {
            ranges[k] = v;
            ++got;
        }
    }
    return got;
}

/* ------------------------------------------------------------------ A2 */

void orc_beam_angles(float angle_min, float angle_inc, int nbeams, float *angles)
{
    /* main.c:53-57 — running float sum, NOT angle_min + k*inc (Appendix A.1) */
    float a = angle_min;
    for (int k = 0; k < nbeams; ++k) {
        angles[k] = a;
        a += angle_inc;
    }
}

int orc_clean_scan(const float *ranges, const float *angles, int nbeams, float range_min,
                   float usable_range, float *x, float *y)
{
    /* main.c:73-94 — gate is (r < range_min) | (r > (int)usable); NaN passes both tests and
     * is kept (Appendix A.2) */
    int n = 0;
    for (int k = 0; k < nbeams; ++k) {
        const float r = ranges[k];
        if ((r < range_min) | (r > usable_range)) continue;   /* the reference's int 24 converts exactly */
        const float a = angles[k];
        x[n] = r * cosf(a);
        y[n] = r * sinf(a);
        ++n;
    }
    return n;
}

/* ------------------------------------------------------------------ A3 */

void orc_transform(const float *x, const float *y, int n, const float pose[3], float *tx, float *ty)
{
    /* main.c:98-117 — world = R^T(theta) * p + t, i.e. the transposed rotation */
    const float c = cosf(pose[2]);
    const float s = sinf(pose[2]);
    for (int i = 0; i < n; ++i) {
        const float px = x[i], py = y[i];
        tx[i] = (c * px + s * py) + pose[0];
        ty[i] = (-s * px + c * py) + pose[1];
    }
}

/* ------------------------------------------------------------------ A4 */

int orc_local_map(const float *map_x, const float *map_y, int map_size, const float *tx,
                  const float *ty, int n, float border, float *loc_x, float *loc_y)
{
    /* main.c:156-182 — bounding box of the transformed scan, grown by the border */
    float lo_x = tx[0], hi_x = tx[0], lo_y = ty[0], hi_y = ty[0];
    for (int i = 1; i < n; ++i) {
        if (tx[i] < lo_x) lo_x = tx[i];
        if (tx[i] > hi_x) hi_x = tx[i];
        if (ty[i] < lo_y) lo_y = ty[i];
        if (ty[i] > hi_y) hi_y = ty[i];
    }
    lo_x = lo_x - border;
    lo_y = lo_y - border;
    hi_x = hi_x + border;
    hi_y = hi_y + border;
    /* main.c:185-198 — strictly-inside points, order kept */
    int m = 0;
    for (int i = 0; i < map_size; ++i) {
        const float qx = map_x[i], qy = map_y[i];
        if (qx > lo_x && qx < hi_x && qy > lo_y && qy < hi_y) {
            loc_x[m] = qx;
            loc_y[m] = qy;
            ++m;
        }
    }
    return m;
}

/* ------------------------------------------------------------------ A5 */

void orc_rasterise(const float *px, const float *py, int n, float pixel, int ld, int ld_rows,
                   int32_t *grid, orc_grid_meta *meta)
{
    /* main.c:272-290 — bounding box of the local map (seeded from element 0) */
    float lo[2] = { px[0], py[0] }, hi[2] = { px[0], py[0] };
    for (int a = 0; a < n; ++a) {
        if (px[a] < lo[0]) lo[0] = px[a];
        if (px[a] > hi[0]) hi[0] = px[a];
        if (py[a] < lo[1]) lo[1] = py[a];
        if (py[a] > hi[1]) hi[1] = py[a];
    }
    /* main.c:297-305 — pad three pixels, extent -> cell count (Appendix A.3) */
    int cells[2];
    for (int a = 0; a < 2; ++a) {
        lo[a] -= (3 * pixel);
        hi[a] += (3 * pixel);
        cells[a] = (int)roundf((hi[a] - lo[a]) / pixel) + 1;
    }
    meta->rows = cells[1];   /* main.c:313-314: y extent is the row count */
    meta->cols = cells[0];
    meta->ld = ld;
    meta->pixel = pixel;
    meta->min_x = lo[0];
    meta->min_y = lo[1];
    memset(grid, 0, sizeof(int32_t) * (size_t)ld * (size_t)ld_rows);   /* main.c:319-320 */
    /* main.c:330-353 — 1-based hit -> linear index -> (row, col) */
    for (int a = 0; a < n; ++a) {
        const int hx = (int)roundf((px[a] - lo[0]) / pixel) + 1;
        const int hy = (int)roundf((py[a] - lo[1]) / pixel) + 1;
        const int lin = ((hy - 1) * cells[0] + hx) - 1;
        const int r = lin / cells[0];
        const int c = lin % cells[0];
        if (r >= 0 && r < ld_rows && c >= 0 && c < ld)   /* the reference has no guard (Q8) */
            grid[(size_t)r * ld + c] = 1;
    }
}

/* ------------------------------------------------------------------ A6 */

void orc_edt_gather(const int32_t *occ, float *out, int ld, int rows, int cols, float cap)
{
    /* main.c:225-243 — running minimum kept as a float distance, compared through its square */
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            if (occ[(size_t)r * ld + c]) {
                out[(size_t)r * ld + c] = 0;
                continue;
            }
            float best = cap;
            for (int j = 0; j < rows; ++j)
                for (int i = 0; i < cols; ++i)
                    if (occ[(size_t)j * ld + i]) {
                        const int dx = c - i, dy = r - j;
                        const int d2 = dx * dx + dy * dy;
                        if ((float)d2 < best * best) best = sqrtf((float)d2);
                    }
            out[(size_t)r * ld + c] = best;
        }
}

void orc_edt_scatter(const int32_t *occ, float *out, int ld, int rows, int cols, float cap)
{
    /* main_accelerated.c:217-247 — every occupied cell (== 1) relaxes the whole rectangle;
     * the comparison there is done in double on a double d2 (main_accelerated.c:231-232) */
    float *dist = (float *)malloc(sizeof(float) * (size_t)rows * (size_t)cols);
    for (size_t k = 0; k < (size_t)rows * (size_t)cols; ++k) dist[k] = cap;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            if (occ[(size_t)r * ld + c] != 1) continue;
            for (int j = 0; j < rows; ++j)
                for (int i = 0; i < cols; ++i) {
                    const double d2 = (double)((r - j) * (r - j) + (c - i) * (c - i));
                    float *d = &dist[(size_t)j * cols + i];
                    if (d2 < (double)(*d * *d)) *d = (float)sqrt((double)(float)d2);
                }
        }
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) out[(size_t)r * ld + c] = dist[(size_t)r * cols + c];
    free(dist);
}

void orc_edt_window(const int32_t *occ, float *out, int ld, int rows, int cols, float cap)
{
    /* Same result as the two above: a candidate is only ever accepted while d2 < cap*cap
     * (SURVEY §3.2), so neighbours with |dx| or |dy| >= ceil(cap) can never matter. */
    const float cap2 = cap * cap;
    const int rad = (int)ceilf(cap);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            if (occ[(size_t)r * ld + c]) {
                out[(size_t)r * ld + c] = 0;
                continue;
            }
            int best = -1;
            const int j0 = r - rad < 0 ? 0 : r - rad, j1 = r + rad >= rows ? rows - 1 : r + rad;
            const int i0 = c - rad < 0 ? 0 : c - rad, i1 = c + rad >= cols ? cols - 1 : c + rad;
            for (int j = j0; j <= j1; ++j)
                for (int i = i0; i <= i1; ++i)
                    if (occ[(size_t)j * ld + i]) {
                        const int d2 = (c - i) * (c - i) + (r - j) * (r - j);
                        if (best < 0 || d2 < best) best = d2;
                    }
            out[(size_t)r * ld + c] = (best >= 0 && (float)best < cap2) ? sqrtf((float)best) : cap;
        }
}

/* ------------------------------------------------------------------ A7 */

float orc_score_pose(const orc_grid_meta *g, const float *edt, const float *bx, const float *by,
                     int nbeams, float x, float y, float ct, float st, float *hits, int *count)
{
    /* main.c:383-389 grid constants; :417-421 pixel scaling; :436-437 translation offset;
     * :462-463 rotation; :483,:501 rounding; :512-518 bounds test, gather, in-order sum */
    const float ipix = 1 / g->pixel;
    const float off_x = (x - g->min_x) * ipix;
    const float off_y = (y - g->min_y) * ipix;
    float total = 0;
    int n = 0;
    for (int b = 0; b < nbeams; ++b) {
        const float qx = bx[b] * ipix;
        const float qy = by[b] * ipix;
        const float rx = (qx * ct) + (qy * st);
        const float ry = (qx * (-st)) + (qy * ct);
        const int cx = (int)roundf(rx + off_x) + 1;
        const int cy = (int)roundf(ry + off_y) + 1;
        if (cx > 1 && cy > 1 && cx < g->cols && cy < g->rows) {
            const float h = edt[(size_t)(cy - 1) * g->ld + (cx - 1)];
            if (hits) hits[n] = h;
            total = total + h;
            ++n;
        }
    }
    *count = n;
    return total;
}

void orc_score_poses(const orc_grid_meta *g, const float *edt, const float *bx, const float *by,
                     int nbeams, const float *x, const float *y, const float *theta, int nposes,
                     float *score, int32_t *count)
{
    for (int p = 0; p < nposes; ++p) {
        int n;
        score[p] = orc_score_pose(g, edt, bx, by, nbeams, x[p], y[p], cosf(theta[p]), sinf(theta[p]), NULL, &n);
        count[p] = n;
    }
}

void orc_fastmatch(const orc_grid_meta *g, const float *edt, const float *bx, const float *by,
                   int nbeams, const float pose[3], const float res[3], float out_pose[3],
                   float *best_hits, int *best_hits_size, float *best_score)
{
    /* main.c:386-387 — res[0] steps both x and y, res[2] steps theta, res[1] is never read */
    const float t = res[0], r = res[2];
    /* main.c:424-426 — the lattice is laid out once around the INPUT pose (Q1) */
    const float th[3] = { pose[2] - r, pose[2], pose[2] + r };
    const float xs[3] = { pose[0] - t, pose[0], pose[0] + t };
    const float ys[3] = { pose[1] - t, pose[1], pose[1] + t };
    float c[3], s[3];
    for (int i = 0; i < 3; ++i) {   /* main.c:433-435 */
        c[i] = cosf(th[i]);
        s[i] = sinf(th[i]);
    }
    float best = INFINITY;
    float bp[3] = { pose[0], pose[1], pose[2] };
    int depth = 0;
    /* main.c:440-591 — up to 50 sweeps; a sweep without improvement deepens, four such end it.
     * Steps are never halved (main.c:577-580 are comments), so sweeps 2..5 repeat sweep 1. */
    for (int iter = 0; iter < 50; ++iter) {
        int unchanged = 1;
        for (int a = 0; a < 3; ++a)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    int n;
                    /* every candidate overwrites best_hits[] (main.c:515, Q2) */
                    const float sc = orc_score_pose(g, edt, bx, by, nbeams, xs[i], ys[j], c[a], s[a], best_hits, &n);
                    if (sc < best) {   /* main.c:549-563 */
                        unchanged = 0;
                        bp[0] = xs[i];
                        bp[1] = ys[j];
                        bp[2] = th[a];
                        *best_hits_size = n;
                        best = sc;
                    }
                }
        if (unchanged && ++depth > 3) break;   /* main.c:576-587 */
    }
    out_pose[0] = bp[0];
    out_pose[1] = bp[1];
    out_pose[2] = bp[2];
    if (best_score) *best_score = best;
}

/* ------------------------------------------------------------------ A8 */

enum { MAP_CAP = 20000, LOCAL_CAP = 25000, HITS_CAP = 2500 };   /* main.c:124, :148, :376 */

struct orc_slam {
    int nbeams;
    float *angles, *sx, *sy, *stx, *sty;
    int scan_n;
    float *map_x, *map_y;
    int map_n;
    float map_pose[3];
    float *loc_x, *loc_y;
    int loc_n;
    int32_t *occ[2];
    float *edt[2];
    orc_grid_meta meta[2];
    float *hits;
    int hits_n;
    float pose[3], prev[3];
    int mini_updated, frame;   /* frame = the reference's scan_iter */
    int edt_variant;
    orc_slam_params par;   /* main.c:832-839, :50, :846, :224, :943 */
    double edt_s, match_s;
    long edt_calls, match_calls;
    long partial_frames;   /* frames whose best candidate had beams out of bounds (the Q2 quirk matters there) */
};

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

orc_slam *orc_slam_create(int nbeams, float angle_min, float angle_inc)
{
    orc_slam *s = (orc_slam *)calloc(1, sizeof *s);
    s->nbeams = nbeams;
    s->angles = (float *)calloc((size_t)nbeams, sizeof(float));
    s->sx = (float *)calloc((size_t)nbeams, sizeof(float));
    s->sy = (float *)calloc((size_t)nbeams, sizeof(float));
    s->stx = (float *)calloc((size_t)nbeams, sizeof(float));
    s->sty = (float *)calloc((size_t)nbeams, sizeof(float));
    s->map_x = (float *)calloc(MAP_CAP + 4096, sizeof(float));
    s->map_y = (float *)calloc(MAP_CAP + 4096, sizeof(float));
    s->loc_x = (float *)calloc(LOCAL_CAP, sizeof(float));
    s->loc_y = (float *)calloc(LOCAL_CAP, sizeof(float));
    const int ld[2] = { 200, 400 };   /* main.c:201, :207 */
    for (int k = 0; k < 2; ++k) {
        s->occ[k] = (int32_t *)calloc((size_t)ld[k] * ld[k], sizeof(int32_t));
        s->edt[k] = (float *)calloc((size_t)ld[k] * ld[k], sizeof(float));
        s->meta[k].ld = ld[k];
    }
    s->hits = (float *)calloc(HITS_CAP > nbeams ? HITS_CAP : nbeams, sizeof(float));
    s->edt_variant = 1;
    orc_slam_params_default(&s->par);
    orc_beam_angles(angle_min, angle_inc, nbeams, s->angles);   /* main.c:845 */
    return s;
}

void orc_slam_params_default(orc_slam_params *p)
{
    const orc_slam_params d = { { 0.05f, 0.05f, 0.008727f },    /* main.c:832 */
                                { 0.025f, 0.025f, 0.004363f },  /* main.c:833 */
                                1.0f, 0.2f, 0.1f,               /* main.c:834-836 */
                                0.3f, 0.0872665f,               /* main.c:838-839 */
                                0.023f, 24.0f,                  /* main.c:50, :846 */
                                10.0f, 1.5f };                  /* main.c:224, :943 */
    *p = d;
}

void orc_slam_set_params(orc_slam *s, const orc_slam_params *p) { s->par = *p; }

void orc_slam_destroy(orc_slam *s)
{
    if (!s) return;
    free(s->angles); free(s->sx); free(s->sy); free(s->stx); free(s->sty);
    free(s->map_x); free(s->map_y); free(s->loc_x); free(s->loc_y);
    for (int k = 0; k < 2; ++k) { free(s->occ[k]); free(s->edt[k]); }
    free(s->hits);
    free(s);
}

void orc_slam_set_edt_variant(orc_slam *s, int v) { s->edt_variant = v; }
int orc_slam_map_size(const orc_slam *s) { return s->map_n; }
const float *orc_slam_map_x(const orc_slam *s) { return s->map_x; }
const float *orc_slam_map_y(const orc_slam *s) { return s->map_y; }

long orc_slam_partial_frames(const orc_slam *s) { return s->partial_frames; }

void orc_slam_timers(const orc_slam *s, double *edt_s, long *edt_calls, double *match_s, long *match_calls)
{
    *edt_s = s->edt_s; *edt_calls = s->edt_calls; *match_s = s->match_s; *match_calls = s->match_calls;
}

void orc_slam_first_frame(orc_slam *s, const float *ranges)
{
    /* main.c:844-858 — scan 0 at pose (0,0,0) seeds the map; the loop starts "mini-updated" */
    const float origin[3] = { 0, 0, 0 };
    s->scan_n = orc_clean_scan(ranges, s->angles, s->nbeams, s->par.range_min, s->par.usable_range, s->sx, s->sy);
    orc_transform(s->sx, s->sy, s->scan_n, origin, s->stx, s->sty);
    memcpy(s->map_x, s->stx, sizeof(float) * (size_t)s->scan_n);   /* main.c:136-145 */
    memcpy(s->map_y, s->sty, sizeof(float) * (size_t)s->scan_n);
    s->map_n = s->scan_n;
    memcpy(s->map_pose, origin, sizeof origin);
    memcpy(s->pose, origin, sizeof origin);
    memcpy(s->prev, origin, sizeof origin);
    s->mini_updated = 1;
    s->frame = 1;
}

static void build_grids(orc_slam *s)
{
    /* main.c:870-871 -> :155-198 and :271-363 (both resolutions, then both EDTs) */
    s->loc_n = orc_local_map(s->map_x, s->map_y, s->map_n, s->stx, s->sty, s->scan_n, s->par.border, s->loc_x, s->loc_y);
    const float pix[2] = { s->par.pixel, s->par.pixel2 };   /* main.c:835-836 */
    for (int k = 0; k < 2; ++k)
        orc_rasterise(s->loc_x, s->loc_y, s->loc_n, pix[k], s->meta[k].ld, s->meta[k].ld, s->occ[k], &s->meta[k]);
    for (int k = 0; k < 2; ++k) {
        const double t0 = now_s();
        const orc_grid_meta *m = &s->meta[k];
        if (s->edt_variant == 0) orc_edt_gather(s->occ[k], s->edt[k], m->ld, m->rows, m->cols, s->par.edt_cap);
        else if (s->edt_variant == 1) orc_edt_scatter(s->occ[k], s->edt[k], m->ld, m->rows, m->cols, s->par.edt_cap);
        else orc_edt_window(s->occ[k], s->edt[k], m->ld, m->rows, m->cols, s->par.edt_cap);
        s->edt_s += now_s() - t0;
        s->edt_calls++;
    }
}

static void match(orc_slam *s, int which, const float pose[3], const float res[3], float out[3])
{
    const double t0 = now_s();
    orc_fastmatch(&s->meta[which], s->edt[which], s->sx, s->sy, s->scan_n, pose, res, out, s->hits, &s->hits_n, NULL);
    s->match_s += now_s() - t0;
    s->match_calls++;
}

void orc_slam_next_frame(orc_slam *s, const float *ranges, float pose_out[3])
{
    const float *coarse = s->par.fast_res;    /* main.c:832 */
    const float *fine = s->par.fast_res2;     /* main.c:833 */

    s->scan_n = orc_clean_scan(ranges, s->angles, s->nbeams, s->par.range_min, s->par.usable_range, s->sx, s->sy);   /* :863 */
    int transformed = 0;
    if (s->mini_updated) {   /* main.c:865-872 — note: transformed with the OLD pose (Q3) */
        orc_transform(s->sx, s->sy, s->scan_n, s->pose, s->stx, s->sty);
        transformed = 1;
        build_grids(s);
    }
    /* main.c:875-898 — constant-velocity guess from the last two poses, no angle wrapping */
    float guess[3];
    for (int i = 0; i < 3; ++i)
        guess[i] = s->frame > 1 ? s->pose[i] + (s->pose[i] - s->prev[i]) : s->pose[i];
    /* main.c:901-918 — coarse search on the coarse grid after a map update, else on the fine
     * grid with the coarse step (Q4); then the refinement on the fine grid */
    float m1[3], m2[3];
    match(s, s->mini_updated ? 0 : 1, guess, coarse, m1);
    match(s, 1, m1, fine, m2);
    memcpy(s->prev, s->pose, sizeof s->prev);
    memcpy(s->pose, m2, sizeof s->pose);
    if (s->hits_n < s->scan_n) s->partial_frames++;

    /* main.c:928-961 — per-axis key-frame test against the pose of the last map update */
    const float dx = fabsf(s->pose[0] - s->map_pose[0]);
    const float dy = fabsf(s->pose[1] - s->map_pose[1]);
    const float dth = fabsf(s->pose[2] - s->map_pose[2]);
    if (dx > s->par.key_dt || dy > s->par.key_dt || dth > s->par.key_dr) {
        s->mini_updated = 1;
        if (!transformed) orc_transform(s->sx, s->sy, s->scan_n, s->pose, s->stx, s->sty);
        /* main.c:941-953 — hits of the LAST candidate, count of the BEST one, and the world
         * points indexed by in-bounds ordinal rather than beam number (Q2) */
        int added = 0;
        for (int j = 0; j < s->hits_n; ++j)
            if (s->hits[j] > s->par.new_point_threshold && s->map_n + added < MAP_CAP + 4096) {
                s->map_x[s->map_n + added] = s->stx[j];
                s->map_y[s->map_n + added] = s->sty[j];
                ++added;
            }
        s->map_n += added;
        memcpy(s->map_pose, s->pose, sizeof s->map_pose);
    } else {
        s->mini_updated = 0;
    }
    s->frame++;
    memcpy(pose_out, s->pose, sizeof s->pose);
}

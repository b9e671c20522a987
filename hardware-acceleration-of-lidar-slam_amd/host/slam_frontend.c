/* slam_frontend.c — see slam_frontend.h.  Float arithmetic follows the reference operation by
 * operation; build with -ffp-contract=off. */
#include "slam_frontend.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int fe_scan_init(fe_scan *s, int nbeams, float angle_min, float angle_inc)
{
    memset(s, 0, sizeof *s);
    s->nbeams = nbeams;
    float **arr[] = { &s->angle, &s->range, &s->bx, &s->by, &s->wx, &s->wy };
    for (unsigned k = 0; k < sizeof arr / sizeof arr[0]; ++k)
        if (!(*arr[k] = (float *)calloc((size_t)nbeams + 1, sizeof(float)))) return -1;
    /* main.c:53-57: the table is accumulated, each entry one float add after the previous */
    float acc = angle_min;
    for (int k = 0; k < nbeams; ++k, acc += angle_inc) s->angle[k] = acc;
    return 0;
}

void fe_scan_free(fe_scan *s)
{
    free(s->angle); free(s->range); free(s->bx); free(s->by); free(s->wx); free(s->wy);
    memset(s, 0, sizeof *s);
}

/* One `fscanf(f, "%f,", &v)` of main.c:26-29, without fscanf: parsing 1079 fields per frame through the scanf machinery was
 * more than half of a frame of the drop-in program (120 us of 150 on the build host).  The same stream position afterwards and
 * the same float:
 *  - leading white space is skipped; end of file before a field: nothing converted;
 *  - a plain decimal field — [+-] digits [. digits], what "%f" prints — is accumulated as an integer m and a digit count d;
 *    m < 2^24 and d <= 10 make m and 10^d exact floats, so the ONE float division m / 10^d is the correctly rounded value of
 *    the decimal string (Clinger's fast path), which is what glibc's strtof — the conversion behind "%f" — returns;
 *  - anything else (more digits, exponents, inf, nan, hexadecimal floats, garbage) goes to strtof itself on the collected
 *    characters, and what strtof does not accept is pushed back;
 *  - a ',' behind the field is consumed, any other character stays in the stream.
 * Returns 1 if a value was converted. */
static int fe_scan_field(FILE *f, float *out)
{
    static const float p10[11] = { 1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f };
    char tok[96];
    int n = 0, c;
    do c = getc_unlocked(f); while (c == ' ' || (c >= '\t' && c <= '\r'));
    if (c == EOF) return 0;
    int neg = 0;
    if (c == '+' || c == '-') {
        neg = c == '-';
        tok[n++] = (char)c;
        c = getc_unlocked(f);
    }
    uint64_t m = 0;
    int digits = 0, frac = 0, simple = 1;
    while (c >= '0' && c <= '9') {
        if (digits < 19) m = m * 10u + (unsigned)(c - '0'); else simple = 0;
        ++digits;
        if (n < 90) tok[n++] = (char)c; else simple = -1;
        c = getc_unlocked(f);
    }
    if (c == '.') {
        if (n < 90) tok[n++] = (char)c; else simple = -1;
        c = getc_unlocked(f);
        while (c >= '0' && c <= '9') {
            if (digits < 19) m = m * 10u + (unsigned)(c - '0'); else simple = 0;
            ++digits;
            ++frac;
            if (n < 90) tok[n++] = (char)c; else simple = -1;
            c = getc_unlocked(f);
        }
    }
    const int more = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');   /* an exponent, inf, nan, 0x...: not the plain form */
    if (digits > 0 && !more && simple == 1 && m < (1u << 24) && frac <= 10) {
        const float v = (float)(uint32_t)m / p10[frac];
        *out = neg ? -v : v;
    } else if (simple < 0) {   /* a field of more than 90 characters: let scanf have what is left of it (not reached by lidar data) */
        if (c != EOF) ungetc(c, f);
        return 0;
    } else {
        /* the general case: the rest of the field's characters, then strtof; what it leaves is pushed back (glibc takes back
         * any number of characters) */
        while (c != EOF && n < 90 && ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '.' || c == '+' ||
                                      c == '-' || c == '(' || c == ')' || c == '_')) {
            tok[n++] = (char)c;
            c = getc_unlocked(f);
        }
        tok[n] = 0;
        char *end = tok;
        const float v = strtof(tok, &end);
        if (c != EOF) ungetc(c, f);
        for (char *q = tok + n; q > end; --q) ungetc((unsigned char)q[-1], f);
        if (end == tok) return 0;   /* no conversion: like scanf, the stream stays in front of the offending character */
        *out = v;
        c = getc_unlocked(f);
    }
    if (c != ',' && c != EOF) ungetc(c, f);
    return 1;
}

int fe_read_frame(FILE *f, fe_scan *s)
{
    /* main.c:26-29 */
    int ok = 0;
    flockfile(f);
    for (int k = 0; k < s->nbeams; ++k) {
        float v;
        if (fe_scan_field(f, &v)) { s->range[k] = v; ++ok; }
    }
    funlockfile(f);
    return ok;
}

static const char k_magic[8] = { 'S', 'L', 'A', 'M', 'S', 'C', 'A', 'N' };

int fe_bin_open(FILE *f, int *nbeams)
{
    char magic[8];
    uint32_t hdr[2];
    const long at = ftell(f);
    if (fread(magic, 1, 8, f) == 8 && memcmp(magic, k_magic, 8) == 0 && fread(hdr, 4, 2, f) == 2 && hdr[0] == 1) {
        *nbeams = (int)hdr[1];
        return 0;
    }
    if (at >= 0) fseek(f, at, SEEK_SET);   /* not a binary stream: leave it where it was (text CSV) */
    return -1;
}

int fe_read_frame_bin(FILE *f, fe_scan *s)
{
    return (int)fread(s->range, sizeof(float), (size_t)s->nbeams, f);
}

int fe_bin_write_header(FILE *f, int nbeams)
{
    const uint32_t hdr[2] = { 1u, (uint32_t)nbeams };
    return fwrite(k_magic, 1, 8, f) == 8 && fwrite(hdr, 4, 2, f) == 2 ? 0 : -1;
}

int fe_bin_write_frame(FILE *f, const fe_scan *s)
{
    return fwrite(s->range, sizeof(float), (size_t)s->nbeams, f) == (size_t)s->nbeams ? 0 : -1;
}

void fe_clean(fe_scan *s, float range_min, float usable_range)
{
    /* main.c:77-94: both comparisons false keeps the beam (so NaN survives) */
    const float hi = (float)usable_range;
    int m = 0;
    for (int k = 0; k < s->nbeams; ++k) {
        const float r = s->range[k];
        const int bad = (r < range_min) | (r > hi);
        if (bad) continue;
        s->bx[m] = r * cosf(s->angle[k]);
        s->by[m] = r * sinf(s->angle[k]);
        ++m;
    }
    s->nscan = m;
}

void fe_to_world(fe_scan *s, const float pose[3])
{
    /* main.c:101-116: rows of the rotation are (ct, st) and (-st, ct) */
    const float ct = cosf(pose[2]), st = sinf(pose[2]);
    for (int k = 0; k < s->nscan; ++k) {
        s->wx[k] = (ct * s->bx[k] + st * s->by[k]) + pose[0];
        s->wy[k] = (-st * s->bx[k] + ct * s->by[k]) + pose[1];
    }
}

int fe_points_init(fe_points *p, int capacity)
{
    memset(p, 0, sizeof *p);
    p->capacity = capacity;
    p->x = (float *)calloc((size_t)capacity + 1, sizeof(float));
    p->y = (float *)calloc((size_t)capacity + 1, sizeof(float));
    return p->x && p->y ? 0 : -1;
}

void fe_points_free(fe_points *p)
{
    free(p->x); free(p->y);
    memset(p, 0, sizeof *p);
}

void fe_crop(const fe_points *map, const fe_scan *s, float border, fe_points *local)
{
    /* main.c:156-182 */
    float x0 = s->wx[0], x1 = s->wx[0], y0 = s->wy[0], y1 = s->wy[0];
    for (int k = 1; k < s->nscan; ++k) {
        const float px = s->wx[k], py = s->wy[k];
        if (px < x0) x0 = px;
        if (px > x1) x1 = px;
        if (py < y0) y0 = py;
        if (py > y1) y1 = py;
    }
    x0 = x0 - border; y0 = y0 - border; x1 = x1 + border; y1 = y1 + border;
    /* main.c:185-198 */
    int m = 0;
    for (int k = 0; k < map->size && m < local->capacity; ++k)
        if (map->x[k] > x0 && map->x[k] < x1 && map->y[k] > y0 && map->y[k] < y1) {
            local->x[m] = map->x[k];
            local->y[m] = map->y[k];
            ++m;
        }
    local->size = m;
}

int fe_grid_init(fe_grid *g, int ld)
{
    memset(g, 0, sizeof *g);
    g->meta.ld = ld;
    g->cell = (int32_t *)calloc((size_t)ld * ld, sizeof(int32_t));
    return g->cell ? 0 : -1;
}

void fe_grid_free(fe_grid *g)
{
    free(g->cell);
    memset(g, 0, sizeof *g);
}

int fe_rasterise(const fe_points *local, float pixel, fe_grid *g)
{
    const int ld = g->meta.ld;
    /* main.c:272-290 */
    float x0 = local->x[0], x1 = local->x[0], y0 = local->y[0], y1 = local->y[0];
    for (int k = 0; k < local->size; ++k) {
        if (local->x[k] < x0) x0 = local->x[k];
        if (local->x[k] > x1) x1 = local->x[k];
        if (local->y[k] < y0) y0 = local->y[k];
        if (local->y[k] > y1) y1 = local->y[k];
    }
    /* main.c:297-305 */
    x0 -= (3 * pixel); x1 += (3 * pixel);
    y0 -= (3 * pixel); y1 += (3 * pixel);
    const int nx = (int)roundf((x1 - x0) / pixel) + 1;
    const int ny = (int)roundf((y1 - y0) / pixel) + 1;
    g->meta.rows = ny;
    g->meta.cols = nx;
    g->meta.pixel = pixel;
    g->meta.min_x = x0;
    g->meta.min_y = y0;
    if (nx > ld || ny > ld || nx < 1 || ny < 1) return -1;
    memset(g->cell, 0, sizeof(int32_t) * (size_t)ld * ld);
    /* main.c:330-353: 1-based hit -> linear index -> (row, col) */
    for (int k = 0; k < local->size; ++k) {
        const int hx = (int)roundf((local->x[k] - x0) / pixel) + 1;
        const int hy = (int)roundf((local->y[k] - y0) / pixel) + 1;
        const int lin = ((hy - 1) * nx + hx) - 1;
        const int r = lin / nx, c = lin % nx;
        if (r >= 0 && r < ld && c >= 0 && c < ld) g->cell[(size_t)r * ld + c] = 1;
    }
    return 0;
}

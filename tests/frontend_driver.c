/* Test driver for the product's C front end (host/slam_frontend.c): runs the host-side stages on the first
 * two frames of a CSV and dumps every intermediate as raw float32/int32 so that the Python test can
 * compare them with the oracle and the reference's golden vectors.  Built with ASan + UBSan. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../hardware-acceleration-of-lidar-slam_amd/host/slam_frontend.h"

static void dump(FILE *f, const char *name, const void *p, size_t bytes)
{
    unsigned len = (unsigned)strlen(name);
    unsigned long long b = bytes;
    fwrite(&len, 4, 1, f); fwrite(name, 1, len, f); fwrite(&b, 8, 1, f); fwrite(p, 1, bytes, f);
}

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    FILE *in = fopen(argv[1], "r"), *out = fopen(argv[2], "wb");
    if (!in || !out) return 1;
    fe_scan s;
    fe_points map, local;
    fe_grid coarse, fine;
    if (fe_scan_init(&s, 1079, -2.351831f, 0.004363f) || fe_points_init(&map, FE_MAP_CAPACITY) ||
        fe_points_init(&local, FE_LOCAL_CAPACITY) || fe_grid_init(&coarse, FE_COARSE_LD) || fe_grid_init(&fine, FE_FINE_LD)) return 1;
    dump(out, "angles", s.angle, 4 * 1079);
    int got = fe_read_frame(in, &s);
    dump(out, "got0", &got, 4);
    dump(out, "ranges0", s.range, 4 * 1079);
    fe_clean(&s, 0.023f, 24);
    dump(out, "bx0", s.bx, 4 * (size_t)s.nscan);
    dump(out, "by0", s.by, 4 * (size_t)s.nscan);
    const float origin[3] = { 0, 0, 0 };
    fe_to_world(&s, origin);
    memcpy(map.x, s.wx, 4 * (size_t)s.nscan);
    memcpy(map.y, s.wy, 4 * (size_t)s.nscan);
    map.size = s.nscan;
    fe_read_frame(in, &s);
    fe_clean(&s, 0.023f, 24);
    const float pose[3] = { 0.15f, 0.004f, -0.024f };
    fe_to_world(&s, pose);
    dump(out, "wx1", s.wx, 4 * (size_t)s.nscan);
    dump(out, "wy1", s.wy, 4 * (size_t)s.nscan);
    fe_crop(&map, &s, 1.0f, &local);
    dump(out, "lx", local.x, 4 * (size_t)local.size);
    dump(out, "ly", local.y, 4 * (size_t)local.size);
    int rc0 = fe_rasterise(&local, 0.2f, &coarse), rc1 = fe_rasterise(&local, 0.1f, &fine);
    dump(out, "rc", &rc0, 4);
    dump(out, "rc1", &rc1, 4);
    dump(out, "meta0", &coarse.meta, sizeof coarse.meta);
    dump(out, "meta1", &fine.meta, sizeof fine.meta);
    dump(out, "grid0", coarse.cell, 4 * 200 * 200);
    dump(out, "grid1", fine.cell, 4 * 400 * 400);
    int eof_got = 0;
    fe_read_frame(in, &s);            /* third frame */
    eof_got = fe_read_frame(in, &s);  /* EOF: nothing converted, previous content kept */
    dump(out, "eof_got", &eof_got, 4);
    fclose(in); fclose(out);
    fe_grid_free(&coarse); fe_grid_free(&fine); fe_points_free(&map); fe_points_free(&local); fe_scan_free(&s);
    return 0;
}

/*
 * slam_main — the reference's SLAM program with its hot path on the MI355X engine.
 *
 * Same inputs and outputs as Subsystem_1/main_accelerated.c: scan-frame CSV in ("%f," x beams per
 * frame, main.c:22-30), "scan N" and "pose = %f  %f  %f" lines on stdout (main.c:860, :965), map
 * points as "%f,%f" lines out (main.c:982-985).  The frame loop below mirrors main.c:825-990; the two
 * hot stages are calls into the C ABI (include/slam_hip.h):
 *     OccupationalGrid's two EDTs   -> slam_grid_upload_host   (replaces main_accelerated.c:215-283)
 *     FastMatch / FastMatch2        -> slam_fastmatch_host     (replaces main.c:381-809)
 * The grids and the scan stay resident on the GPU between the calls of one frame.
 *
 * usage: slam_main [--mapper] dataset frames beams map_out.csv [angle_min angle_inc] [--params P x 15]
 *        --params: the 15 floats of slam_mapper_params in declaration order (fast_res[3] fast_res2[3] border pixel pixel2
 *        key_dt key_dr range_min usable_range edt_cap new_point_threshold); default: the reference's (main.c:832-839 ...)
 *        slam_main --to-binary dataset.csv frames beams dataset.bin      (no GPU needed)
 * --mapper runs the whole per-frame pipeline through slam_mapper_* (scan clean-up, local map, rasters and map
 * update on the device too) instead of the host front end below; same results.
 * `dataset` is the reference's CSV or the binary scan-frame stream of slam_frontend.h (auto-detected; the
 * binary form carries exactly the floats the CSV parser yields, results are identical, ingest is ~20x
 * cheaper).  Fails (non-zero exit) when no gfx950 GPU is available: there is no CPU fallback.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "slam_frontend.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int rc__ = (call);                                                                              \
        if (rc__ != SLAM_OK) {                                                                          \
            fprintf(stderr, "%s failed: %s (%s)\n", #call, slam_status_string(rc__), slam_last_error(eng)); \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

int main(int argc, char **argv)
{
    if (argc >= 6 && strcmp(argv[1], "--to-binary") == 0) {   /* CSV -> binary scan-frame stream */
        FILE *src = fopen(argv[2], "r"), *dst = fopen(argv[5], "wb");
        const int nfr = atoi(argv[3]), nb = atoi(argv[4]);
        fe_scan tmp;
        if (!src || !dst || fe_scan_init(&tmp, nb, 0, 0) || fe_bin_write_header(dst, nb)) return 1;
        for (int k = 0; k < nfr; ++k)
            if (fe_read_frame(src, &tmp) != nb || fe_bin_write_frame(dst, &tmp)) return 1;
        fclose(src);
        fclose(dst);
        fe_scan_free(&tmp);
        return 0;
    }
    int use_mapper = 0;
    if (argc >= 2 && strcmp(argv[1], "--mapper") == 0) {
        use_mapper = 1;
        --argc;
        ++argv;
    }
    slam_mapper_params par;
    slam_mapper_params_default(&par);
    for (int a = 1; a < argc; ++a)
        if (strcmp(argv[a], "--params") == 0) {
            if (argc - a - 1 < 15) { fprintf(stderr, "--params needs 15 values\n"); return 2; }
            float *f = (float *)&par;   /* 15 floats in declaration order */
            for (int k = 0; k < 15; ++k) f[k] = (float)atof(argv[a + 1 + k]);
            argc = a;
            break;
        }
    if (argc < 5) {
        fprintf(stderr, "usage: %s [--mapper] dataset frames beams map_out.csv [angle_min angle_inc] [--params P x 15]\n", argv[0]);
        return 2;
    }
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 1; }
    const int frames = atoi(argv[2]);
    int beams = atoi(argv[3]);
    int bin_beams = 0;
    const int binary = fe_bin_open(in, &bin_beams) == 0;
    if (binary && bin_beams != beams) {
        fprintf(stderr, "%s holds %d beams per frame, not %d\n", argv[1], bin_beams, beams);
        return 1;
    }
    const float angle_min = argc > 6 ? (float)atof(argv[5]) : -2.351831f;   /* main.c:47 */
    const float angle_inc = argc > 6 ? (float)atof(argv[6]) : 0.004363f;    /* main.c:49 */

    /* the reference's run-time parameters (main.c:830-839, :224, :943): slam_mapper_params, defaults = the reference's */
    const float *coarse_step = par.fast_res, *fine_step = par.fast_res2;
    const float border = par.border, pixel_coarse = par.pixel, pixel_fine = par.pixel2;
    const float key_dt = par.key_dt, key_dr = par.key_dr;
    const float edt_cap = par.edt_cap;

    slam_engine *eng = NULL;
    {
        int rc = slam_engine_create(0, &eng);
        if (rc != SLAM_OK) {
            fprintf(stderr, "slam_engine_create: %s\n", slam_status_string(rc));
            return 1;
        }
    }
    if (use_mapper) {   /* the device-resident form of the same loop */
        fe_scan rd;
        slam_mapper *mp = NULL;
        if (fe_scan_init(&rd, beams, angle_min, angle_inc)) return 1;
        CHECK(slam_mapper_create_ex(eng, beams, angle_min, angle_inc, &par, &mp));
        const double t0 = now_s();
        if (binary) fe_read_frame_bin(in, &rd); else fe_read_frame(in, &rd);
        CHECK(slam_mapper_first_frame(mp, rd.range));
        for (int k = 1; k < frames; ++k) {
            float p[3];
            printf("scan %d\n", k + 1);
            if (binary) fe_read_frame_bin(in, &rd); else fe_read_frame(in, &rd);
            CHECK(slam_mapper_next_frame(mp, rd.range, p));
            printf("pose = %f  %f  %f\n", p[0], p[1], p[2]);
        }
        fprintf(stderr, "frames %d  wall %.6f s  (mapper: front end on the device)\n", frames, now_s() - t0);
        int32_t n = 0;
        CHECK(slam_mapper_get_map_host(mp, NULL, NULL, 0, &n));
        float *mx = (float *)calloc((size_t)n + 1, sizeof(float)), *my = (float *)calloc((size_t)n + 1, sizeof(float));
        CHECK(slam_mapper_get_map_host(mp, mx, my, n, &n));
        FILE *mo = fopen(argv[4], "w");
        if (!mo) { perror(argv[4]); return 1; }
        for (int j = 0; j < n; ++j) fprintf(mo, "%f,%f\n", mx[j], my[j]);
        fclose(mo);
        fclose(in);
        free(mx); free(my);
        fe_scan_free(&rd);
        slam_mapper_destroy(mp);
        slam_engine_destroy(eng);
        return 0;
    }
    fe_scan scan;
    fe_points map, local;
    fe_grid coarse, fine;
    if (fe_scan_init(&scan, beams, angle_min, angle_inc) || fe_points_init(&map, FE_MAP_CAPACITY + beams) ||
        fe_points_init(&local, FE_LOCAL_CAPACITY) || fe_grid_init(&coarse, FE_COARSE_LD) || fe_grid_init(&fine, FE_FINE_LD)) {
        fprintf(stderr, "out of memory\n");
        return 1;
    }
    float *hits = (float *)calloc((size_t)beams + 1, sizeof(float));
    int32_t nhits = 0;

    double t_edt = 0, t_match = 0;
    long n_edt = 0, n_match = 0;
    const double t_begin = now_s();

    /* main.c:844-858: frame 0 at the origin seeds the map */
    float pose[3] = { 0, 0, 0 }, prev[3] = { 0, 0, 0 };
    if (binary) fe_read_frame_bin(in, &scan); else fe_read_frame(in, &scan);
    fe_clean(&scan, par.range_min, par.usable_range);
    fe_to_world(&scan, pose);
    memcpy(map.x, scan.wx, sizeof(float) * (size_t)scan.nscan);
    memcpy(map.y, scan.wy, sizeof(float) * (size_t)scan.nscan);
    map.size = scan.nscan;
    memcpy(map.pose, pose, sizeof pose);
    int mini_updated = 1;

    for (int k = 1; k < frames; ++k) {
        printf("scan %d\n", k + 1);
        if (binary) fe_read_frame_bin(in, &scan); else fe_read_frame(in, &scan);
        fe_clean(&scan, par.range_min, par.usable_range);
        CHECK(slam_scan_upload_host(eng, scan.bx, scan.by, scan.nscan));
        int in_world = 0;
        if (mini_updated) {   /* main.c:865-872 (world points from the OLD pose) */
            fe_to_world(&scan, pose);
            in_world = 1;
            fe_crop(&map, &scan, border, &local);
            if (fe_rasterise(&local, pixel_coarse, &coarse) || fe_rasterise(&local, pixel_fine, &fine)) {
                fprintf(stderr, "frame %d: map extent exceeds the %d/%d-cell grids\n", k + 1, FE_COARSE_LD, FE_FINE_LD);
                return 1;
            }
            const double t0 = now_s();
            CHECK(slam_grid_upload_host(eng, 0, coarse.cell, &coarse.meta, edt_cap, NULL));
            CHECK(slam_grid_upload_host(eng, 1, fine.cell, &fine.meta, edt_cap, NULL));
            t_edt += now_s() - t0;
            n_edt += 2;
        }
        /* main.c:875-898 */
        float guess[3];
        for (int a = 0; a < 3; ++a) guess[a] = k > 1 ? pose[a] + (pose[a] - prev[a]) : pose[a];
        /* main.c:901-918 (coarse step on the fine grid when the map was not just rebuilt, Q4) */
        float m1[3], m2[3];
        const double t1 = now_s();
        CHECK(slam_fastmatch_host(eng, mini_updated ? 0 : 1, guess, coarse_step, m1, hits, &nhits, NULL));
        CHECK(slam_fastmatch_host(eng, 1, m1, fine_step, m2, hits, &nhits, NULL));
        t_match += now_s() - t1;
        n_match += 2;
        memcpy(prev, pose, sizeof prev);
        memcpy(pose, m2, sizeof pose);

        /* main.c:928-961 */
        if (fabsf(pose[0] - map.pose[0]) > key_dt || fabsf(pose[1] - map.pose[1]) > key_dt ||
            fabsf(pose[2] - map.pose[2]) > key_dr) {
            mini_updated = 1;
            if (!in_world) fe_to_world(&scan, pose);
            int added = 0;
            for (int j = 0; j < nhits; ++j)   /* hits of the LAST candidate, count of the BEST (Q2) */
                if (hits[j] > par.new_point_threshold && map.size + added < map.capacity) {
                    map.x[map.size + added] = scan.wx[j];
                    map.y[map.size + added] = scan.wy[j];
                    ++added;
                }
            map.size += added;
            memcpy(map.pose, pose, sizeof pose);
        } else {
            mini_updated = 0;
        }
        printf("pose = %f  %f  %f\n", pose[0], pose[1], pose[2]);
    }
    const double wall = now_s() - t_begin;
    fclose(in);
    fprintf(stderr, "frames %d  wall %.6f s  edt %.6f s / %ld calls  match %.6f s / %ld calls\n", frames, wall, t_edt,
            n_edt, t_match, n_match);

    FILE *out = fopen(argv[4], "w");
    if (!out) { perror(argv[4]); return 1; }
    for (int j = 0; j < map.size; ++j) fprintf(out, "%f,%f\n", map.x[j], map.y[j]);
    fclose(out);

    free(hits);
    fe_grid_free(&coarse); fe_grid_free(&fine);
    fe_points_free(&map); fe_points_free(&local);
    fe_scan_free(&scan);
    slam_engine_destroy(eng);
    return 0;
}

/*
 * slam_pf_main — particle-filter form of the reference's SLAM program, host code in C.
 *
 * Same scan-frame CSV input and the same pose / map outputs as Subsystem_1/main_accelerated.c
 * ("scan N", "pose = %f  %f  %f" on stdout, "%f,%f" map lines; main.c:22-30, :860, :965, :982-985), and the
 * same map machinery (local map, two rasters, EDT on key frames; main.c:865-872, :928-961).  What changes
 * is the pose search: instead of the reference's two 27-pose lattice sweeps (main.c:901-918) every frame
 * runs one particle-filter step on the GPU — N particles are moved by the constant-velocity increment of
 * main.c:875-898 plus noise, each is scored against the fine EDT with the reference's own score function,
 * weights are normalised and the population is resampled; the frame's pose is the mean of the resampled
 * (hence equally weighted) population, accumulated in double in index order — or, with estimator "best", the
 * heaviest particle.
 * All of it goes through the C ABI (slam_pf_* in include/slam_hip.h); no HIP type appears here.
 *
 * usage: slam_pf_main dataset.csv frames beams map_out.csv particles [seed [mean|best]]
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

#define CHECK(call)                                                                                         \
    do {                                                                                                    \
        int rc__ = (call);                                                                                  \
        if (rc__ != SLAM_OK) {                                                                              \
            fprintf(stderr, "%s failed: %s (%s)\n", #call, slam_status_string(rc__), slam_last_error(eng)); \
            return 1;                                                                                       \
        }                                                                                                   \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: %s dataset.csv frames beams map_out.csv particles [seed]\n", argv[0]);
        return 2;
    }
    FILE *in = fopen(argv[1], "r");
    if (!in) { perror(argv[1]); return 1; }
    const int frames = atoi(argv[2]);
    const int beams = atoi(argv[3]);
    const int particles = atoi(argv[5]);
    const unsigned long long seed = argc > 6 ? strtoull(argv[6], NULL, 10) : 1;
    const int use_mean = !(argc > 7 && strcmp(argv[7], "best") == 0);
    float *px = (float *)calloc((size_t)particles, sizeof(float)), *py = (float *)calloc((size_t)particles, sizeof(float)),
          *pt = (float *)calloc((size_t)particles, sizeof(float));

    const float border = 1, pixel_coarse = 0.2f, pixel_fine = 0.1f;   /* main.c:834-836 */
    const float key_dt = 0.3f, key_dr = 0.0872665f, edt_cap = 10;     /* main.c:838-839, :224 */

    slam_engine *eng = NULL;
    {
        int rc = slam_engine_create(0, &eng);
        if (rc != SLAM_OK) {
            fprintf(stderr, "slam_engine_create: %s\n", slam_status_string(rc));
            return 1;
        }
    }
    slam_pf *pf = NULL;
    /* motion noise of the order of the reference's fine lattice step (0.025 m, 0.004363 rad; main.c:833) */
    const slam_pf_config cfg = { particles, 0, { 0.01f, 0.01f, 0.002f }, 1.0f, 0.25f, seed };
    CHECK(slam_pf_create(eng, &cfg, &pf));

    fe_scan scan;
    fe_points map, local;
    fe_grid coarse, fine;
    if (fe_scan_init(&scan, beams, -2.351831f, 0.004363f) || fe_points_init(&map, FE_MAP_CAPACITY + beams) ||
        fe_points_init(&local, FE_LOCAL_CAPACITY) || fe_grid_init(&coarse, FE_COARSE_LD) || fe_grid_init(&fine, FE_FINE_LD)) {
        fprintf(stderr, "out of memory\n");
        return 1;
    }
    float *hits = (float *)calloc((size_t)beams + 1, sizeof(float));
    int32_t nhits = 0;
    double t_step = 0;
    const double t_begin = now_s();

    float pose[3] = { 0, 0, 0 }, prev[3] = { 0, 0, 0 };
    fe_read_frame(in, &scan);
    fe_clean(&scan, 0.023f, 24);
    fe_to_world(&scan, pose);
    memcpy(map.x, scan.wx, sizeof(float) * (size_t)scan.nscan);
    memcpy(map.y, scan.wy, sizeof(float) * (size_t)scan.nscan);
    map.size = scan.nscan;
    memcpy(map.pose, pose, sizeof pose);
    CHECK(slam_pf_reset(pf, pose));
    int mini_updated = 1;

    for (int k = 1; k < frames; ++k) {
        printf("scan %d\n", k + 1);
        fe_read_frame(in, &scan);
        fe_clean(&scan, 0.023f, 24);
        CHECK(slam_scan_upload_host(eng, scan.bx, scan.by, scan.nscan));
        int in_world = 0;
        if (mini_updated) {
            fe_to_world(&scan, pose);
            in_world = 1;
            fe_crop(&map, &scan, border, &local);
            if (fe_rasterise(&local, pixel_coarse, &coarse) || fe_rasterise(&local, pixel_fine, &fine)) {
                fprintf(stderr, "frame %d: map extent exceeds the grids\n", k + 1);
                return 1;
            }
            CHECK(slam_grid_upload_host(eng, 0, coarse.cell, &coarse.meta, edt_cap, NULL));
            CHECK(slam_grid_upload_host(eng, 1, fine.cell, &fine.meta, edt_cap, NULL));
        }
        /* constant-velocity increment (main.c:875-898) drives the motion model of every particle */
        float dp[3];
        for (int a = 0; a < 3; ++a) dp[a] = k > 1 ? pose[a] - prev[a] : 0.0f;
        const double t0 = now_s();
        CHECK(slam_pf_step(pf, 1, dp, 0));
        float best[3];
        if (use_mean) {   /* posterior mean = plain mean of the resampled population */
            CHECK(slam_pf_get_poses_host(pf, px, py, pt));
            double sx = 0, sy = 0, st = 0;
            for (int i = 0; i < particles; ++i) { sx += px[i]; sy += py[i]; st += pt[i]; }
            best[0] = (float)(sx / particles); best[1] = (float)(sy / particles); best[2] = (float)(st / particles);
        } else {
            CHECK(slam_pf_best(pf, best, NULL, NULL));
        }
        t_step += now_s() - t0;
        memcpy(prev, pose, sizeof prev);
        memcpy(pose, best, sizeof pose);

        if (fabsf(pose[0] - map.pose[0]) > key_dt || fabsf(pose[1] - map.pose[1]) > key_dt ||
            fabsf(pose[2] - map.pose[2]) > key_dr) {
            mini_updated = 1;
            if (!in_world) fe_to_world(&scan, pose);
            /* hits of the frame's pose on the fine grid (the reference uses its matcher scratch here, main.c:942-948) */
            CHECK(slam_pose_hits_host(eng, 1, pose[0], pose[1], cosf(pose[2]), sinf(pose[2]), hits, &nhits));
            int added = 0;
            for (int j = 0; j < nhits; ++j)
                if (hits[j] > 1.5 && map.size + added < map.capacity) {
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
    fprintf(stderr, "frames %d  particles %d  wall %.6f s  pf steps %.6f s (%.3e particle-updates/s)\n", frames, particles,
            wall, t_step, (double)particles * (frames - 1) / t_step);

    FILE *out = fopen(argv[4], "w");
    if (!out) { perror(argv[4]); return 1; }
    for (int j = 0; j < map.size; ++j) fprintf(out, "%f,%f\n", map.x[j], map.y[j]);
    fclose(out);
    free(hits); free(px); free(py); free(pt);
    fe_grid_free(&coarse); fe_grid_free(&fine);
    fe_points_free(&map); fe_points_free(&local);
    fe_scan_free(&scan);
    slam_pf_destroy(pf);
    slam_engine_destroy(eng);
    return 0;
}

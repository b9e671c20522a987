/*
 * slam_pf_main — particle-filter form of the reference's SLAM program, host code in C, one or several GPUs.
 *
 * Same scan-frame CSV input and the same pose / map outputs as Subsystem_1/main_accelerated.c
 * ("scan N", "pose = %f  %f  %f" on stdout, "%f,%f" map lines; main.c:22-30, :860, :965, :982-985), and the
 * same map machinery (local map, two rasters, EDT on key frames; main.c:865-872, :928-961).  What changes
 * is the pose search: instead of the reference's two 27-pose lattice sweeps (main.c:901-918) every frame
 * runs one particle-filter step on the GPU — N particles are moved by the constant-velocity increment of
 * main.c:875-898 plus noise, each is scored against the fine EDT with the reference's own score function,
 * weights are normalised and the population is resampled; the frame's pose is the mean of the resampled
 * (hence equally weighted) population (exact fixed-point sums on the device) — or, with estimator "best", the
 * heaviest particle.
 * All of it goes through the C ABI (slam_pf_* in include/slam_hip.h); no HIP or RCCL type appears here.
 *
 * Several GPUs: the accelerator handle is created once per GPU and threaded through, the shape of the reference's
 * FPGA host (Submodule_2/Hadrware_acclereated.cpp:842-845, 284): one host thread per GPU owns one engine, one
 * communicator and one sharded session, runs this same frame loop, and the engine issues every exchange between
 * the GPUs itself (RCCL over xGMI).  The population is split in contiguous blocks; the output does not depend on
 * the number of GPUs (bit for bit: pose log and map file).
 *
 * usage: slam_pf_main dataset.csv frames beams map_out.csv particles [seed [mean|best]]
 *                     [--gpus N] [--transport rccl|local] [--same-device] [--ess F]
 *   particles      the whole population (a multiple of N)
 *   --ess F        ESS-gated resampling: resample only in frames whose effective sample size is below F * N
 *   --transport    rccl (default when N > 1): RCCL, one GPU per rank.  local: the in-process transport.
 *   --same-device  every rank on device 0 (only with the local transport; RCCL refuses two ranks on one GPU)
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
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

typedef struct {
    /* the run */
    const char *dataset, *map_out;
    int frames, beams, particles_total, use_mean;
    unsigned long long seed;
    float ess_frac;        /* --ess F: resample only when the effective sample size is below F * N (0: every frame) */
    /* the ranks */
    int world, use_rccl, same_device;
    uint8_t comm_id[SLAM_COMM_ID_BYTES];
    slam_local_group *group;
} run_t;

typedef struct {
    run_t *run;
    int rank, rc;
} rank_t;

#define CHECK(call)                                                                                               \
    do {                                                                                                          \
        int rc__ = (call);                                                                                        \
        if (rc__ != SLAM_OK) {                                                                                    \
            fprintf(stderr, "rank %d: %s failed: %s (%s)\n", rank, #call, slam_status_string(rc__),                \
                    eng ? slam_last_error(eng) : "");                                                             \
            goto fail;                                                                                            \
        }                                                                                                         \
    } while (0)

static void *rank_main(void *arg)
{
    rank_t *me = (rank_t *)arg;
    run_t *run = me->run;
    const int rank = me->rank, world = run->world;
    const int beams = run->beams, frames = run->frames;
    const int n = run->particles_total / world, n_total = run->particles_total;
    const float border = 1, pixel_coarse = 0.2f, pixel_fine = 0.1f;   /* main.c:834-836 */
    const float key_dt = 0.3f, key_dr = 0.0872665f, edt_cap = 10;     /* main.c:838-839, :224 */
    slam_engine *eng = NULL;
    slam_comm *comm = NULL;
    slam_pf *pf = NULL;
    FILE *in = NULL;
    float *hits = NULL;
    fe_scan scan = { 0 };
    fe_points map = { 0 }, local = { 0 };
    fe_grid coarse = { 0 }, fine = { 0 };
    me->rc = 1;

    in = fopen(run->dataset, "r");   /* every rank reads the (small) scan stream itself: no host-side broadcast */
    if (!in) { perror(run->dataset); goto fail; }
    {
        int rc = slam_engine_create(run->same_device ? 0 : rank, &eng);
        if (rc != SLAM_OK) {
            fprintf(stderr, "rank %d: slam_engine_create: %s\n", rank, slam_status_string(rc));
            eng = NULL;
            goto fail;
        }
    }
    /* motion noise of the order of the reference's fine lattice step (0.025 m, 0.004363 rad; main.c:833) */
    const slam_pf_config cfg = { n, 0, { 0.01f, 0.01f, 0.002f }, 1.0f, 0.25f, run->seed, run->ess_frac, SLAM_MAP_AUTO };
    if (world > 1 || run->use_rccl || run->group) {
        if (run->group) CHECK(slam_comm_create_local(eng, run->group, rank, &comm));
        else CHECK(slam_comm_create_rccl(eng, rank, world, run->comm_id, &comm));
        CHECK(slam_pf_create_sharded(eng, &cfg, comm, 0, &pf));
    } else {
        CHECK(slam_pf_create(eng, &cfg, &pf));
    }
    if (fe_scan_init(&scan, beams, -2.351831f, 0.004363f) || fe_points_init(&map, FE_MAP_CAPACITY + beams) ||
        fe_points_init(&local, FE_LOCAL_CAPACITY) || fe_grid_init(&coarse, FE_COARSE_LD) || fe_grid_init(&fine, FE_FINE_LD) ||
        !(hits = (float *)calloc((size_t)beams + 1, sizeof(float)))) {
        fprintf(stderr, "rank %d: out of memory\n", rank);
        goto fail;
    }
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
        if (rank == 0) printf("scan %d\n", k + 1);
        fe_read_frame(in, &scan);
        fe_clean(&scan, 0.023f, 24);
        CHECK(slam_scan_upload_host(eng, scan.bx, scan.by, scan.nscan));
        int in_world = 0;
        if (mini_updated) {   /* map, rasters and EDTs are replicated on every GPU (<= 16 MiB) */
            fe_to_world(&scan, pose);
            in_world = 1;
            fe_crop(&map, &scan, border, &local);
            if (fe_rasterise(&local, pixel_coarse, &coarse) || fe_rasterise(&local, pixel_fine, &fine)) {
                fprintf(stderr, "frame %d: map extent exceeds the grids\n", k + 1);
                goto fail;
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
        if (run->use_mean) {
            /* posterior mean = plain mean of the resampled population, over ALL ranks' particles: exact integer sums made on
             * the device (slam_pf_mean), so the result does not depend on the number of GPUs.  Headings are averaged on the
             * circle, around the predicted heading, so that a population straddling +-pi does not average to nonsense and
             * theta stays unwrapped (the reference never normalises angles). */
            CHECK(slam_pf_mean(pf, pose[2] + dp[2], best));
        } else {
            CHECK(slam_pf_best(pf, best, NULL, NULL));   /* sharded: the heaviest of the whole population, on every rank */
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
        if (rank == 0) printf("pose = %f  %f  %f\n", pose[0], pose[1], pose[2]);
    }
    if (rank == 0) {
        const double wall = now_s() - t_begin;
        fprintf(stderr, "frames %d  particles %d  gpus %d  wall %.6f s  pf steps %.6f s (%.3e particle-updates/s)\n", frames,
                n_total, world, wall, t_step, (double)n_total * (frames - 1) / t_step);
        FILE *out = fopen(run->map_out, "w");
        if (!out) { perror(run->map_out); goto fail; }
        for (int j = 0; j < map.size; ++j) fprintf(out, "%f,%f\n", map.x[j], map.y[j]);
        fclose(out);
    }
    me->rc = 0;
fail:
    if (me->rc && comm) slam_comm_abort(comm);   /* the other ranks must not wait for this one: their calls fail with SLAM_ERR_COMM */
    free(hits);
    fe_grid_free(&coarse); fe_grid_free(&fine);
    fe_points_free(&map); fe_points_free(&local);
    fe_scan_free(&scan);
    if (in) fclose(in);
    slam_pf_destroy(pf);
    slam_comm_destroy(comm);
    slam_engine_destroy(eng);
    return NULL;
}

int main(int argc, char **argv)
{
    run_t run;
    memset(&run, 0, sizeof run);
    run.world = 1;
    run.use_mean = 1;
    run.seed = 1;
    const char *pos[8];
    int npos = 0, transport_given = 0, use_local = 0;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "--gpus") && a + 1 < argc) run.world = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--transport") && a + 1 < argc) { use_local = !strcmp(argv[++a], "local"); transport_given = 1; }
        else if (!strcmp(argv[a], "--same-device")) run.same_device = 1;
        else if (!strcmp(argv[a], "--ess") && a + 1 < argc) run.ess_frac = (float)atof(argv[++a]);
        else if (npos < 8) pos[npos++] = argv[a];
    }
    if (npos < 5 || run.world < 1 || run.world > 16) {
        fprintf(stderr, "usage: %s dataset.csv frames beams map_out.csv particles [seed [mean|best]] [--gpus N] "
                        "[--transport rccl|local] [--same-device] [--ess F]\n", argv[0]);
        return 2;
    }
    run.dataset = pos[0];
    run.frames = atoi(pos[1]);
    run.beams = atoi(pos[2]);
    run.map_out = pos[3];
    run.particles_total = atoi(pos[4]);
    if (npos > 5) run.seed = strtoull(pos[5], NULL, 10);
    if (npos > 6 && !strcmp(pos[6], "best")) run.use_mean = 0;
    if (run.particles_total <= 0 || run.particles_total % run.world) {
        fprintf(stderr, "particles must be a positive multiple of the number of GPUs\n");
        return 2;
    }
    if (run.same_device && !use_local && run.world > 1) {
        fprintf(stderr, "--same-device needs --transport local (RCCL refuses two ranks on one GPU)\n");
        return 2;
    }
    run.use_rccl = transport_given && !use_local;   /* "--gpus 1 --transport rccl": the sharded path on a one-rank group */
    if (use_local) {
        if (slam_local_group_create(run.world, &run.group) != SLAM_OK) return 1;
    } else if (run.world > 1 || run.use_rccl) {
        if (slam_comm_unique_id(run.comm_id) != SLAM_OK) {
            fprintf(stderr, "slam_comm_unique_id failed\n");
            return 1;
        }
    }

    rank_t ranks[16];
    pthread_t th[16];
    for (int r = 0; r < run.world; ++r) { ranks[r].run = &run; ranks[r].rank = r; ranks[r].rc = 1; }
    if (run.world == 1) {
        rank_main(&ranks[0]);
    } else {
        for (int r = 0; r < run.world; ++r)
            if (pthread_create(&th[r], NULL, rank_main, &ranks[r]) != 0) {
                /* the ranks already running would wait for this one in their first collective: end the whole program */
                fprintf(stderr, "pthread_create failed for rank %d\n", r);
                exit(1);
            }
        for (int r = 0; r < run.world; ++r) pthread_join(th[r], NULL);
    }
    int rc = 0;
    for (int r = 0; r < run.world; ++r) rc |= ranks[r].rc;
    slam_local_group_destroy(run.group);
    return rc;
}

/*
 * main_cpu — whole-program form of the CPU restatement (slam_oracle.c), with the reference
 * program's inputs and outputs: scan-frame CSV in, "scan N" / "pose = x  y  th" lines on
 * stdout (/root/reference/Subsystem_1/main.c:860, :965), map CSV "%f,%f\n" out (:982-985).
 *
 * TEST INFRASTRUCTURE: used to pin the oracle against the reference's own stdout/map output
 * byte for byte, and as the `cpu_baseline` of bench.py.  Not part of the product.
 *
 * usage: main_cpu dataset.csv frames beams edt_variant map_out.csv [angle_min angle_inc] [--params P x 15]
 *        edt_variant: 0 = gather (main.c), 1 = scatter (main_accelerated.c), 2 = window
 *        --params: the 15 floats of orc_slam_params in declaration order (default: the reference's, main.c:832-839 ...)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "slam_oracle.h"

int main(int argc, char **argv)
{
    orc_slam_params par;
    orc_slam_params_default(&par);
    for (int a = 1; a < argc; ++a)
        if (strcmp(argv[a], "--params") == 0) {
            if (argc - a - 1 < 15) { fprintf(stderr, "--params needs 15 values\n"); return 2; }
            float *f = (float *)&par;
            for (int k = 0; k < 15; ++k) f[k] = (float)atof(argv[a + 1 + k]);
            argc = a;
            break;
        }
    if (argc < 6) {
        fprintf(stderr, "usage: %s dataset.csv frames beams edt_variant map_out.csv [angle_min angle_inc]\n", argv[0]);
        return 2;
    }
    FILE *in = fopen(argv[1], "r");
    if (!in) { perror(argv[1]); return 1; }
    const int frames = atoi(argv[2]);
    const int beams = atoi(argv[3]);
    const float amin = argc > 7 ? (float)atof(argv[6]) : -2.351831f;   /* main.c:47 */
    const float ainc = argc > 7 ? (float)atof(argv[7]) : 0.004363f;    /* main.c:49 */

    float *ranges = (float *)calloc((size_t)beams, sizeof(float));
    orc_slam *s = orc_slam_create(beams, amin, ainc);
    orc_slam_set_edt_variant(s, atoi(argv[4]));
    orc_slam_set_params(s, &par);

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    orc_parse_frame(in, ranges, beams);
    orc_slam_first_frame(s, ranges);
    for (int k = 1; k < frames; ++k) {
        float pose[3];
        printf("scan %d\n", k + 1);
        orc_parse_frame(in, ranges, beams);
        orc_slam_next_frame(s, ranges, pose);
        printf("pose = %f  %f  %f\n", pose[0], pose[1], pose[2]);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fclose(in);

    double edt_s, match_s;
    long edt_calls, match_calls;
    orc_slam_timers(s, &edt_s, &edt_calls, &match_s, &match_calls);
    fprintf(stderr, "frames %d  wall %.6f s  edt %.6f s / %ld calls  match %.6f s / %ld calls  partial-inbounds frames %ld\n",
            frames, (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec), edt_s, edt_calls, match_s,
            match_calls, orc_slam_partial_frames(s));

    FILE *out = fopen(argv[5], "w");
    if (!out) { perror(argv[5]); return 1; }
    const float *mx = orc_slam_map_x(s), *my = orc_slam_map_y(s);
    for (int j = 0; j < orc_slam_map_size(s); ++j) fprintf(out, "%f,%f\n", mx[j], my[j]);
    fclose(out);
    orc_slam_destroy(s);
    free(ranges);
    return 0;
}

/*
 * slam_frontend — host side of the SLAM frame loop, in C like the reference's host code.
 *
 * These are the cheap, once-per-frame stages that stay on the CPU:
 * frame parsing, scan clean-up, world transform, local-map crop, occupancy rasterisation and the
 * key-frame / map-append bookkeeping.  Everything hot (EDT build, scan-match score) goes through the
 * engine's C ABI (include/slam_hip.h).  Product code: it does not use anything from oracle/.
 *
 * Reference being mirrored (paths relative to the reference repository, Subsystem_1/):
 *   main.c:22-30 frame reader   :45-95 lidar table + scan clean-up   :97-118 transform
 *   :136-198 map init + local map   :271-354 rasterisation   :875-898 predict   :928-961 map update
 */
#ifndef SLAM_FRONTEND_H
#define SLAM_FRONTEND_H

#include <stdint.h>
#include <stdio.h>

#include "../../include/slam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FE_MAP_CAPACITY = 20000,    /* main.c:124 */
    FE_LOCAL_CAPACITY = 25000,  /* main.c:148 */
    FE_COARSE_LD = 200,         /* main.c:201 */
    FE_FINE_LD = 400            /* main.c:207 */
};

typedef struct {
    int nbeams;
    float *angle;             /* [nbeams] beam angle table */
    float *range;             /* [nbeams] current raw frame */
    float *bx, *by;           /* sensor-frame cartesian survivors */
    float *wx, *wy;           /* the same in the world frame (only valid after fe_to_world) */
    int nscan;
} fe_scan;

typedef struct {
    float *x, *y;
    int size, capacity;
    float pose[3];            /* pose of the last map update */
} fe_points;

typedef struct {
    int32_t *cell;            /* [ld][ld] occupancy, row-major */
    slam_grid_meta meta;
} fe_grid;

int fe_scan_init(fe_scan *s, int nbeams, float angle_min, float angle_inc);
void fe_scan_free(fe_scan *s);
/* One scan frame in the reference's text format (main.c:22-30: nbeams fields "%f,"); returns the number of values
 * converted (nbeams on a complete frame).  The fields are not parsed by fscanf but by a reader of this file that gives the
 * same floats and leaves the stream at the same place (plain decimals through one exact float division, everything else
 * through strtof; tests/test_host_frontend.py compares the two field by field): 16 instead of 135 us per 1079-beam frame on
 * the build host, which was more than half of a frame of the drop-in program. */
int fe_read_frame(FILE *f, fe_scan *s);
/* Binary scan-frame stream: a 16-byte header {"SLAMSCAN", uint32 version = 1,
 * uint32 nbeams} followed by frames of nbeams little-endian float32 ranges.  The values are exactly the
 * floats the text reader produces from the CSV, so both paths give identical results. */
int fe_bin_open(FILE *f, int *nbeams);                 /* 0 if `f` starts with a valid header (consumed) */
int fe_read_frame_bin(FILE *f, fe_scan *s);            /* values read (nbeams on a complete frame) */
int fe_bin_write_header(FILE *f, int nbeams);
int fe_bin_write_frame(FILE *f, const fe_scan *s);
void fe_clean(fe_scan *s, float range_min, float usable_range);
void fe_to_world(fe_scan *s, const float pose[3]);

int fe_points_init(fe_points *p, int capacity);
void fe_points_free(fe_points *p);
void fe_crop(const fe_points *map, const fe_scan *s, float border, fe_points *local);

int fe_grid_init(fe_grid *g, int ld);
void fe_grid_free(fe_grid *g);
/* returns 0, or -1 when the padded extent does not fit ld x ld (the reference would overrun, Q8) */
int fe_rasterise(const fe_points *local, float pixel, fe_grid *g);

#ifdef __cplusplus
}
#endif
#endif

/*
 * slam_oracle — CPU restatement of the reference's scan-matching SLAM hot path, in plain C.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may link, load or call anything under oracle/.  The product
 * (the HIP engine behind include/slam_hip.h) never routes through this file and has no CPU
 * fallback.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Parity status: rows A1-A8 are PINNED — tests/test_oracle_vs_reference.py
 * checks them bit-for-bit against golden vectors captured from the compiled, unmodified
 * reference (oracle/make_golden.py, oracle/_ref/).  The particle-filter stages (A9-A12:
 * motion sample, per-landmark EKF, weights, resample) have NO counterpart in the reference:
 * for them this file is the specification and their parity is UNPINNED (see slam_oracle_pf.c).
 *
 * All arithmetic is IEEE binary32 in the reference's operation order; build with
 * -ffp-contract=off (oracle/Makefile does).
 */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ A1-A4: front end */

/* One scan frame of `nbeams` ranges, text, "%f," each (Subsystem_1/main.c:22-30).
 * Returns the number of values actually converted; unconverted slots keep the previous
 * value, as the reference's fscanf loop does. */
int orc_parse_frame(FILE *f, float *ranges, int nbeams);

/* Beam angle table by repeated float addition (main.c:53-57). */
void orc_beam_angles(float angle_min, float angle_inc, int nbeams, float *angles);

/* Range gate + polar->cartesian + order-preserving compaction (main.c:71-95).
 * Keeps beam k unless r < range_min or r > usable_range (int).  Returns the survivor count. */
int orc_clean_scan(const float *ranges, const float *angles, int nbeams, float range_min,
                   float usable_range, float *x, float *y);

/* Sensor frame -> world frame with the reference's transposed rotation (main.c:97-118). */
void orc_transform(const float *x, const float *y, int n, const float pose[3], float *tx, float *ty);

/* Crop of the global map to the scan's bounding box +- border (main.c:155-198). */
int orc_local_map(const float *map_x, const float *map_y, int map_size, const float *tx,
                  const float *ty, int n, float border, float *loc_x, float *loc_y);

/* ------------------------------------------------------------------ A5-A6: grid + EDT */

typedef struct {
    int rows, cols;      /* used rectangle (main.c:313-317: rows = y extent, cols = x extent) */
    int ld;              /* leading dimension of the storage, 200 / 400 in the reference */
    float pixel;         /* metres per cell (main.c:357-358) */
    float min_x, min_y;  /* world coordinate of the padded top-left corner (main.c:359-362) */
} orc_grid_meta;

/* Rasterise points into a zeroed int grid[ld_rows][ld] (main.c:272-354, one resolution).
 * The whole ld_rows x ld storage is cleared first, like the reference's memset. */
void orc_rasterise(const float *px, const float *py, int n, float pixel, int ld, int ld_rows,
                   int32_t *grid, orc_grid_meta *meta);

/* Capped exact Euclidean distance transform over the used rows x cols rectangle; cells outside
 * it are left untouched.  Three formulations, identical results:
 *   gather  — main.c:223-245 (for every free cell scan every occupied cell)
 *   scatter — main_accelerated.c:215-247 (every occupied cell relaxes every cell)
 *   window  — search only |dx|,|dy| < ceil(cap): what the cap makes sufficient; the only one
 *             cheap enough for the 1024^2 / 2048^2 parity cases. */
void orc_edt_gather(const int32_t *occ, float *out, int ld, int rows, int cols, float cap);
void orc_edt_scatter(const int32_t *occ, float *out, int ld, int rows, int cols, float cap);
void orc_edt_window(const int32_t *occ, float *out, int ld, int rows, int cols, float cap);

/* ------------------------------------------------------------------ A7: scan-match score */

/* Score of ONE pose given cos/sin of its heading (main.c:417-438, 459-518).
 * bx/by are sensor-frame cartesian beams (NOT yet pixel-scaled).  Writes the in-bounds hit
 * values, in beam order, to hits[] when hits != NULL; *count = number of in-bounds beams. */
float orc_score_pose(const orc_grid_meta *g, const float *edt, const float *bx, const float *by,
                     int nbeams, float x, float y, float ct, float st, float *hits, int *count);

/* Batch form used for the particle filter measurement update: poses as SoA x,y,theta, heading
 * trig from libm cosf/sinf (what the reference uses for its 3 lattice headings). */
void orc_score_poses(const orc_grid_meta *g, const float *edt, const float *bx, const float *by,
                     int nbeams, const float *x, const float *y, const float *theta, int nposes,
                     float *score, int32_t *count);

/* The reference's 27-pose lattice search, quirks included (main.c:381-596 / 598-809):
 * lattice never re-centres, step never halves, strict '<' keeps the first best, best_hits[]
 * ends up holding the LAST candidate's hits while *best_hits_size belongs to the best pose
 * (SURVEY Q1, Q2, Q5). */
void orc_fastmatch(const orc_grid_meta *g, const float *edt, const float *bx, const float *by,
                   int nbeams, const float pose[3], const float res[3], float out_pose[3],
                   float *best_hits, int *best_hits_size, float *best_score);

/* ------------------------------------------------------------------ A8: frame loop */

typedef struct orc_slam orc_slam;   /* whole-pipeline state (the reference's globals) */

/* the reference's run-time parameters as one struct (same fields, same order as slam_mapper_params in include/slam_hip.h) */
typedef struct {
    float fast_res[3], fast_res2[3];   /* main.c:832-833 */
    float border, pixel, pixel2;       /* main.c:834-836 */
    float key_dt, key_dr;              /* main.c:838-839 */
    float range_min, usable_range;     /* main.c:50, :846 */
    float edt_cap, new_point_threshold;   /* main.c:224, :943 */
} orc_slam_params;
void orc_slam_params_default(orc_slam_params *p);
orc_slam *orc_slam_create(int nbeams, float angle_min, float angle_inc);
void orc_slam_destroy(orc_slam *s);
/* edt_variant: 0 gather (main.c), 1 scatter (main_accelerated.c), 2 window */
void orc_slam_set_edt_variant(orc_slam *s, int edt_variant);
void orc_slam_set_params(orc_slam *s, const orc_slam_params *p);
/* first frame: builds the initial map at pose (0,0,0) (main.c:844-852) */
void orc_slam_first_frame(orc_slam *s, const float *ranges);
/* every later frame (main.c:859-969); writes the matched pose */
void orc_slam_next_frame(orc_slam *s, const float *ranges, float pose_out[3]);
int orc_slam_map_size(const orc_slam *s);
const float *orc_slam_map_x(const orc_slam *s);
const float *orc_slam_map_y(const orc_slam *s);
/* frames whose matched pose had beams outside the grid (where SURVEY Q2's hit-scratch quirk has an effect) */
long orc_slam_partial_frames(const orc_slam *s);
/* instrumentation for the CPU baseline: seconds spent and calls made in EDT / matcher */
void orc_slam_timers(const orc_slam *s, double *edt_s, long *edt_calls, double *match_s, long *match_calls);

#ifdef __cplusplus
}
#endif
#endif

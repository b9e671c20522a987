/*
 * ref_compat — the reference-side binding: the four hot functions of the reference program, with the
 * reference's own names and signatures, implemented on the engine's C ABI.
 *
 * Link this file (plus libslam_hip.so) into Subsystem_1/main_accelerated.c in place of its own
 *   euclidean_distance_transform / euclidean_distance_transform2   (main_accelerated.c:215, :250)
 *   FastMatch / FastMatch2                                        (main_accelerated.c:396, :613)
 * and the unmodified rest of that program (main, readAScan, Transform, ExtractLocalMap,
 * OccupationalGrid, the map update) runs with its hot path on the MI355X.  INTEGRATION.md has the
 * recipe; oracle/Makefile target _ref/main_accel_dropin does exactly this with the reference object
 * (hot symbols weakened) and tests/test_gpu_scanmatch.py checks its pose log against the golden one.
 *
 * The reference passes everything but (POSE, searchResolution) through file-scope globals, so the
 * adapter declares those globals with the reference's layouts (column = 1079, main_accelerated.c:7).
 */
#include <stdio.h>
#include <stdlib.h>

#include "../../include/slam_hip.h"

typedef struct {                       /* main_accelerated.c:60-66 */
    float x[1079], y[1079], tx[1079], ty[1079];
    int size;
} ScanData;
typedef struct {                       /* main_accelerated.c:200-212 */
    int grid[200][200];
    int grid_size[2];
    float metric_grid[200][200];
    float pixel_size;
    float top_left_corner[2];
    int grid2[400][400];
    int grid_size2[2];
    float metric_grid2[400][400];
    float pixel_size2;
    float top_left_corner2[2];
} MyGrid;
typedef struct {                       /* main_accelerated.c:389-393 */
    float pose[3];
    float bestHits[2500];
    int bestHits_size;
} MyFastMatchParameters;

extern ScanData scan;
extern MyGrid occ_grid;
extern MyFastMatchParameters FastMatchParameters;

static slam_engine *engine(void)
{
    static slam_engine *eng;
    if (!eng) {
        int rc = slam_engine_create(0, &eng);
        if (rc != SLAM_OK) {   /* no CPU fallback: without the GPU the program cannot continue */
            fprintf(stderr, "ref_compat: slam_engine_create: %s\n", slam_status_string(rc));
            exit(1);
        }
    }
    return eng;
}

static void must(int rc, const char *what)
{
    if (rc != SLAM_OK) {
        fprintf(stderr, "ref_compat: %s: %s (%s)\n", what, slam_status_string(rc), slam_last_error(engine()));
        exit(1);
    }
}

/* The reference calls these as (grid, metric_grid, grid_size[1], grid_size[0]) (main_accelerated.c:370-371),
 * so `height` carries the column count and `width` the row count.  The EDT stays on the
 * device as grid slot 0 / 1 and is also copied back into the reference's metric grid. */
static void edt_into_slot(int slot, const int *in, float *out, int ld, int rows, int cols)
{
    const slam_grid_meta m = { rows, cols, ld, 1.0f, 0.0f, 0.0f };   /* pixel/corner are set after the EDTs (:372-377) */
    must(slam_grid_upload_host(engine(), slot, in, &m, 10.0f /* MAX_DIST, :217 */, out), "slam_grid_upload_host");
}

#ifndef SLAM_REF_EDT_WIDTH_HEIGHT
void euclidean_distance_transform(int input_map[200][200], float output_distance_map[200][200], int height, int width)
{
    edt_into_slot(0, &input_map[0][0], &output_distance_map[0][0], 200, width, height);
}

void euclidean_distance_transform2(int input_map[400][400], float output_distance_map[400][400], int height, int width)
{
    edt_into_slot(1, &input_map[0][0], &output_distance_map[0][0], 400, width, height);
}
#else
/* -DSLAM_REF_EDT_WIDTH_HEIGHT: the signature of the STAND-ALONE file,
 * Submodule_2/Accelereated_Euclidean_Distance_Transform.c:1 and :36 — (width, height) in that order, and its body
 * indexes [i < width][j < height], so the FIRST integer counts rows here.  Behind main.c's call site
 * (grid_size[1], grid_size[0]) that file is only right for square extents; this variant
 * reproduces exactly what it computes for a caller written against that file. */
void euclidean_distance_transform(int input_map[200][200], float output_distance_map[200][200], int width, int height)
{
    edt_into_slot(0, &input_map[0][0], &output_distance_map[0][0], 200, width, height);
}

void euclidean_distance_transform2(int input_map[400][400], float output_distance_map[400][400], int width, int height)
{
    edt_into_slot(1, &input_map[0][0], &output_distance_map[0][0], 400, width, height);
}
#endif

static void match_on_slot(int slot, const float POSE[3], const float searchResolution[3])
{
    const slam_grid_meta m = slot == 0
        ? (slam_grid_meta){ occ_grid.grid_size[0], occ_grid.grid_size[1], 200, occ_grid.pixel_size,
                            occ_grid.top_left_corner[0], occ_grid.top_left_corner[1] }
        : (slam_grid_meta){ occ_grid.grid_size2[0], occ_grid.grid_size2[1], 400, occ_grid.pixel_size2,
                            occ_grid.top_left_corner2[0], occ_grid.top_left_corner2[1] };
    must(slam_grid_set_meta(engine(), slot, &m), "slam_grid_set_meta");
    must(slam_scan_upload_host(engine(), scan.x, scan.y, scan.size), "slam_scan_upload_host");
    int32_t n = FastMatchParameters.bestHits_size;
    must(slam_fastmatch_host(engine(), slot, POSE, searchResolution, FastMatchParameters.pose,
                             FastMatchParameters.bestHits, &n, NULL), "slam_fastmatch_host");
    FastMatchParameters.bestHits_size = n;
}

void FastMatch(const float POSE[3], const float searchResolution[3]) { match_on_slot(0, POSE, searchResolution); }
void FastMatch2(const float POSE[3], const float searchResolution[3]) { match_on_slot(1, POSE, searchResolution); }

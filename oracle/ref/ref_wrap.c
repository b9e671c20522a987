/*
 * ref_wrap — turns the UNMODIFIED reference translation unit into a shared library so that the
 * golden-vector script (oracle/make_golden.py) and the oracle-pinning tests can call the
 * reference's own functions and read its file-scope globals.
 *
 * TEST INFRASTRUCTURE.  Built only in the build container, from the sources where they lie under
 * /root/reference (never copied into this repository); output goes to oracle/_ref/ (git-ignored).
 * Build: see oracle/Makefile, targets _ref/libref_main.so and _ref/libref_accel.so, which pass
 *   -DREF_SOURCE='"/root/reference/Subsystem_1/main.c"'  (or main_accelerated.c).
 *
 * Technique: rename the reference's main(), include the file, then undo
 * the `row` / `column` macros it defines (main.c:6-7).
 */
#define main reference_main
#include REF_SOURCE
#undef main
#undef row
#undef column

/* ---- raw frame / scan (main.c:10, 43, 69) ---- */
float *ref_ranges(void) { return test_input_memory; }
int ref_num_beams(void) { return (int)(sizeof test_input_memory / sizeof test_input_memory[0]); }
void ref_read_frame(const char *path, int frame_index)
{
    FILE *f = fopen(path, "r");
    for (int i = 0; i <= frame_index; ++i) readDatasetLineByLine(f);
    fclose(f);
}
void ref_set_lidar(void) { SetLidarParameters(); }
float *ref_angles(void) { return lidar.angles; }
int ref_read_scan(int usable_range) { readAScan(usable_range); return scan.size; }
float *ref_scan_x(void) { return scan.x; }
float *ref_scan_y(void) { return scan.y; }
float *ref_scan_tx(void) { return scan.tx; }
float *ref_scan_ty(void) { return scan.ty; }
int *ref_scan_size(void) { return &scan.size; }
void ref_transform(const float *pose) { Transform(pose); }

/* ---- map / local map (main.c:123-198) ---- */
void ref_initialise(const float *pose) { Initialise(pose); }
float *ref_map_x(void) { return map.x; }
float *ref_map_y(void) { return map.y; }
int *ref_map_size(void) { return &map.size; }
float *ref_map_pose(void) { return map.pose; }
void ref_extract_local_map(float border) { ExtractLocalMap(border); }
float *ref_local_x(void) { return local_map.x; }
float *ref_local_y(void) { return local_map.y; }
int *ref_local_size(void) { return &local_map.size; }

/* ---- occupancy grid + EDT (main.c:200-363) ---- */
void ref_occupancy_grid(float pix, float pix2) { OccupationalGrid(pix, pix2); }
int *ref_grid(int which) { return which ? &occ_grid.grid2[0][0] : &occ_grid.grid[0][0]; }
float *ref_metric(int which) { return which ? &occ_grid.metric_grid2[0][0] : &occ_grid.metric_grid[0][0]; }
int *ref_grid_size(int which) { return which ? occ_grid.grid_size2 : occ_grid.grid_size; }
float *ref_pixel_size(int which) { return which ? &occ_grid.pixel_size2 : &occ_grid.pixel_size; }
float *ref_top_left(int which) { return which ? occ_grid.top_left_corner2 : occ_grid.top_left_corner; }
int ref_grid_ld(int which) { return which ? 400 : 200; }
/* call the EDT exactly as OccupationalGrid does (main.c:355-356): (in, out, grid_size[1], grid_size[0]) */
void ref_edt(int which)
{
    if (which) euclidean_distance_transform2(occ_grid.grid2, occ_grid.metric_grid2, occ_grid.grid_size2[1], occ_grid.grid_size2[0]);
    else euclidean_distance_transform(occ_grid.grid, occ_grid.metric_grid, occ_grid.grid_size[1], occ_grid.grid_size[0]);
}

/* ---- scan matcher (main.c:374-809) ---- */
void ref_fastmatch(int which, const float *pose, const float *res)
{
    if (which) FastMatch2(pose, res); else FastMatch(pose, res);
}
float *ref_fm_pose(void) { return FastMatchParameters.pose; }
float *ref_fm_hits(void) { return FastMatchParameters.bestHits; }
int *ref_fm_hits_size(void) { return &FastMatchParameters.bestHits_size; }

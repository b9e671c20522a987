/*
 * ref_edt_standalone — the reference's STAND-ALONE scatter EDT file,
 *   Submodule_2/Accelereated_Euclidean_Distance_Transform.c:1 (200 x 200 form) and :36 (400 x 400 form),
 * compiled from the source where it lies (never copied) into oracle/_ref/libref_edt_standalone.so.
 *
 * TEST INFRASTRUCTURE (build container only; see oracle/Makefile target `ref`).  The file has no
 * #include of its own although it calls sqrt(), hence the <math.h> below.  Its parameter order is
 * (width, height) and its body indexes [i < width][j < height], i.e. `width` counts ROWS of the
 * row-major arrays — the reverse of main_accelerated.c:215, which declares (height, width) for the
 * same body.  With main.c's call site (grid_size[1], grid_size[0]) = (#cols, #rows) it would walk
 * #cols rows and #rows columns, so it is only meaningful for square extents;
 * the golden cases are square.
 */
#include <math.h>
#include REF_EDT_SOURCE

static int g200[200][200], g400[400][400];
static float m200[200][200], m400[400][400];

int *ref_sa_grid(int which) { return which ? &g400[0][0] : &g200[0][0]; }
float *ref_sa_metric(int which) { return which ? &m400[0][0] : &m200[0][0]; }
void ref_sa_edt(int which, int width, int height)
{
    if (which) euclidean_distance_transform2(g400, m400, width, height);
    else euclidean_distance_transform(g200, m200, width, height);
}

/*
 * gen_dataset — deterministic synthetic lidar dataset in the reference's scan-frame format.
 *
 * TEST INFRASTRUCTURE (oracle/): used by tests, bench.py's cpu_baseline leg and the golden
 * generation scripts only.  Never linked into the product library.
 *
 * Output format = what the reference's frame reader consumes
 * (/root/reference/Subsystem_1/main.c:22-30, `fscanf("%f,")` x `column` per frame):
 * one text line per scan frame, `beams` ranges printed with "%f" and separated by commas.
 *
 * World: axis-aligned room 15 x 11 m with two box obstacles, robot advancing
 * 4 mm and 0.6 mrad per frame, uniform +-5 mm range noise from a 32-bit LCG, a sprinkling of
 * dropouts (range 0 -> below range_min) and over-range returns (30 m -> above the usable 24 m)
 * so that the scan clean-up/compaction path (main.c:71-95) is exercised.
 *
 * Bit-reproducibility across machines: only IEEE-754 double +,-,*,/ are used (no libm; the
 * sine/cosine below is a Taylor kernel + three angle doublings), so the same command line gives
 * the same bytes on the build container and on the GPU box.  Build with -ffp-contract=off.
 *
 * usage: gen_dataset out.csv frames beams angle_min angle_inc seed [step_m turn_rad [world]]
 *        parity set : 1000 1079 -2.351831 0.004363 1            (arc, 4 mm + 0.6 mrad per frame)
 *        loop set   : 3480 1079 -2.351831 0.004363 2 0.004 0.0018  (one full 2.2 m-radius loop,
 *                     the frame count main_accelerated.c:6 is compiled for)
 *        hall set   : 1000 1079 -2.351831 0.004363 3 0.012 0.0002 1  (world 1 = 34 x 20 m hall: the far end is
 *                     beyond the 24 m usable range at first, so newly seen walls fall OUTSIDE the matcher's
 *                     grid for a while — beams out of bounds, the Q2 hit-scratch quirk in action)
 *        bench sets use 360 beams over 2*pi
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

static void det_sincos(double a, double *s, double *c)
{
    /* sin/cos(a/8) by Taylor series (|a| <= ~7 => |a/8| < 1), then three doublings */
    double x = a * 0.125, x2 = x * x;
    double ts = x, tc = 1.0, ss = x, cc = 1.0;
    for (int k = 1; k <= 12; ++k) {
        tc = -tc * x2 / (double)((2 * k - 1) * (2 * k));
        ts = -ts * x2 / (double)((2 * k) * (2 * k + 1));
        cc += tc;
        ss += ts;
    }
    for (int d = 0; d < 3; ++d) {
        double s2 = 2.0 * ss * cc;
        double c2 = cc * cc - ss * ss;
        ss = s2;
        cc = c2;
    }
    *s = ss;
    *c = cc;
}

typedef struct { double x0, y0, x1, y1; } box_t;

/* room first (hit from inside), then obstacles (hit from outside); world 0 = room, world 1 = long hall */
static box_t k_room = { -3.0, -5.5, 12.0, 5.5 };
static box_t k_obst[2] = { { 5.0, 2.0, 7.0, 3.5 }, { 2.0, -4.0, 3.0, -3.0 } };
static const box_t k_hall = { -3.0, -10.0, 31.0, 10.0 };
static const box_t k_hall_obst[2] = { { 9.0, 4.0, 11.0, 6.5 }, { 16.0, -7.0, 18.0, -5.0 } };

static double ray_box(double ox, double oy, double dx, double dy, const box_t *b, double best)
{
    /* nearest positive hit of the ray with the four edges of b, if closer than best */
    const double ex[2] = { b->x0, b->x1 }, ey[2] = { b->y0, b->y1 };
    for (int i = 0; i < 2; ++i) {
        if (dx != 0.0) {
            double t = (ex[i] - ox) / dx;
            if (t > 1e-9 && t < best) {
                double y = oy + t * dy;
                if (y >= b->y0 && y <= b->y1) best = t;
            }
        }
        if (dy != 0.0) {
            double t = (ey[i] - oy) / dy;
            if (t > 1e-9 && t < best) {
                double x = ox + t * dx;
                if (x >= b->x0 && x <= b->x1) best = t;
            }
        }
    }
    return best;
}

static uint32_t g_lcg;
static uint32_t lcg_next(void)
{
    g_lcg = g_lcg * 1664525u + 1013904223u;
    return g_lcg;
}

int main(int argc, char **argv)
{
    if (argc != 7 && argc != 9 && argc != 10) {
        fprintf(stderr, "usage: %s out.csv frames beams angle_min angle_inc seed [step_m turn_rad [world]]\n", argv[0]);
        return 2;
    }
    FILE *out = fopen(argv[1], "w");
    if (!out) { perror(argv[1]); return 1; }
    const int frames = atoi(argv[2]);
    const int beams = atoi(argv[3]);
    const double amin = atof(argv[4]);
    const double ainc = atof(argv[5]);
    g_lcg = (uint32_t)strtoul(argv[6], NULL, 10) * 2654435761u + 12345u;
    const double step = argc >= 9 ? atof(argv[7]) : 0.004;
    const double turn = argc >= 9 ? atof(argv[8]) : 0.0006;
    if (argc == 10 && atoi(argv[9]) == 1) {
        k_room = k_hall;
        k_obst[0] = k_hall_obst[0];
        k_obst[1] = k_hall_obst[1];
    }

    /* robot truth in the usual convention (heading phi, +CCW); the reference's theta is -phi */
    double px = 0.0, py = 0.0, phi = 0.0;
    for (int f = 0; f < frames; ++f) {
        for (int k = 0; k < beams; ++k) {
            double s, c;
            det_sincos(phi + (amin + (double)k * ainc), &s, &c);
            double t = ray_box(px, py, c, s, &k_room, 1e30);
            t = ray_box(px, py, c, s, &k_obst[0], t);
            t = ray_box(px, py, c, s, &k_obst[1], t);
            uint32_t u = lcg_next();
            double noise = ((double)(u >> 8) / 16777216.0 - 0.5) * 0.01;
            double r = t + noise;
            uint32_t v = lcg_next() >> 16;
            if (v % 97u == 0u) r = 0.0;          /* dropout: below range_min 0.023 */
            else if (v % 89u == 0u) r = 30.0;    /* over-range: above usable 24 */
            fprintf(out, k + 1 < beams ? "%f," : "%f\n", (double)(float)r);
        }
        double s, c;
        det_sincos(phi, &s, &c);
        px += step * c;
        py += step * s;
        phi += turn;
    }
    fclose(out);
    return 0;
}

/* Test driver: the same text parsed field by field with fscanf(f, "%f,", &v) — what the reference's reader does,
 * main.c:26-29 — and with the product's fe_read_frame (host/slam_frontend.c), both dumped as raw float32 together with the
 * number of conversions and the stream position afterwards.  Built with ASan + UBSan by tests/test_host_frontend.py. */
#include <stdio.h>
#include <stdlib.h>

#include "../hardware-acceleration-of-lidar-slam_amd/host/slam_frontend.h"

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    const int n = atoi(argv[2]);
    FILE *out = fopen(argv[3], "wb");
    if (!out || n <= 0) return 1;
    float *a = (float *)calloc((size_t)n + 1, sizeof(float));
    fe_scan s;
    if (!a || fe_scan_init(&s, n, 0.0f, 0.0f)) return 1;
    for (int k = 0; k < n; ++k) a[k] = s.range[k] = -12345.0f;   /* fields without a conversion keep this */
    FILE *f = fopen(argv[1], "r");
    if (!f) return 1;
    int ok_a = 0;
    for (int k = 0; k < n; ++k) {
        float v;
        if (fscanf(f, "%f,", &v) == 1) { a[k] = v; ++ok_a; }
    }
    long pos_a = ftell(f);
    fclose(f);
    f = fopen(argv[1], "r");
    if (!f) return 1;
    int ok_b = fe_read_frame(f, &s);
    long pos_b = ftell(f);
    fclose(f);
    fwrite(&ok_a, 4, 1, out); fwrite(&ok_b, 4, 1, out);
    fwrite(&pos_a, 8, 1, out); fwrite(&pos_b, 8, 1, out);
    fwrite(a, 4, (size_t)n, out); fwrite(s.range, 4, (size_t)n, out);
    fclose(out);
    free(a);
    fe_scan_free(&s);
    return 0;
}

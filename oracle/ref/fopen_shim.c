/*
 * fopen_shim — lets the UNMODIFIED reference programs run in this container.
 *
 * TEST INFRASTRUCTURE.  The reference opens two hard-coded Windows paths
 * (/root/reference/Subsystem_1/main.c:842 dataset, :982 map output).  The whole-program oracle is
 * compiled with -Dfopen=oracle_fopen, which sends every read-open to $ORACLE_DATASET and every
 * write-open to $ORACLE_MAP_OUT.  This file is compiled WITHOUT that define.
 */
#include <stdio.h>
#include <stdlib.h>

FILE *oracle_fopen(const char *path, const char *mode)
{
    (void)path;
    const char *p = getenv(mode[0] == 'r' ? "ORACLE_DATASET" : "ORACLE_MAP_OUT");
    if (!p) {
        fprintf(stderr, "oracle_fopen: set ORACLE_DATASET / ORACLE_MAP_OUT\n");
        exit(3);
    }
    return fopen(p, mode);
}

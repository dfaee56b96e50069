/* A plain C99 client of include/jjs_gpu.h: what a cgo / Rust-FFI / JNI binding sees.  Reads vectors
 *   <scheme>[_ext|_wire] <expected status> <hex fields in ABI order ...>
 * from the file given as argv[1] and verifies each through the blocking host-buffer entry points, one item per call:
 *   <scheme>       jjs_verify_single / _double / _vargen            points affine, 64 bytes
 *   <scheme>_ext   jjs_verify_*_ext (what INTEGRATION.md's shim calls) points U || V || Z, 96 bytes
 *   <scheme>_wire  jjs_verify_*_wire                                  sig, pk, m as the reference serialises them
 * With no argument it only checks that the header declares what it promises (used by the CPU test as a link check).
 * Test infrastructure, not product. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jjs_gpu.h"
#include "jjs_gpu_profiling.h"

static int unhex(const char* s, unsigned char* out, size_t n) {
    size_t i;
    if (strlen(s) != 2 * n) return -1;
    for (i = 0; i < n; ++i) {
        unsigned v;
        if (sscanf(s + 2 * i, "%2x", &v) != 1) return -1;
        out[i] = (unsigned char)v;
    }
    return 0;
}

int main(int argc, char** argv) {
    FILE* f;
    char line[4096];
    int total = 0, failures = 0;
    if (argc < 2) {
        printf("abi version %d\n", jjs_abi_version());
        return jjs_abi_version() >= 4 ? 0 : 1;
    }
    if (jjs_init(1) != JJS_OK) { fprintf(stderr, "jjs_init: %s\n", jjs_last_error()); return 2; }
    f = fopen(argv[1], "r");
    if (!f) return 2;
    while (fgets(line, sizeof line, f)) {
        /* 16-byte aligned, as the ABI asks */
        static unsigned char buf[6][96] __attribute__((aligned(16)));
        static unsigned char status[16] __attribute__((aligned(16)));
        uint64_t tally[4];
        char* tok[9];
        int n = 0, want, rc = -1;
        char* p = strtok(line, " \n");
        while (p && n < 9) { tok[n++] = p; p = strtok(NULL, " \n"); }
        if (n < 5) continue;
        want = atoi(tok[1]);
        ++total;
        if (!strcmp(tok[0], "single") && n == 6) {            /* u R PK m */
            if (unhex(tok[2], buf[0], 32) || unhex(tok[3], buf[1], 64) || unhex(tok[4], buf[2], 64) || unhex(tok[5], buf[3], 32)) { ++failures; continue; }
            rc = jjs_verify_single(buf[0], buf[1], buf[2], buf[3], 1, status, tally);
        } else if (!strcmp(tok[0], "double") && n == 8) {     /* u R R' PK PK' m */
            if (unhex(tok[2], buf[0], 32) || unhex(tok[3], buf[1], 64) || unhex(tok[4], buf[2], 64) || unhex(tok[5], buf[3], 64) ||
                unhex(tok[6], buf[4], 64) || unhex(tok[7], buf[5], 32)) { ++failures; continue; }
            rc = jjs_verify_double(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], 1, status, tally);
        } else if (!strcmp(tok[0], "vargen") && n == 7) {     /* u R PK Gen m */
            if (unhex(tok[2], buf[0], 32) || unhex(tok[3], buf[1], 64) || unhex(tok[4], buf[2], 64) || unhex(tok[5], buf[3], 64) ||
                unhex(tok[6], buf[4], 32)) { ++failures; continue; }
            rc = jjs_verify_vargen(buf[0], buf[1], buf[2], buf[3], buf[4], 1, status, tally);
        } else if (!strcmp(tok[0], "single_ext") && n == 6) {    /* u R PK m, points 96 bytes */
            if (unhex(tok[2], buf[0], 32) || unhex(tok[3], buf[1], 96) || unhex(tok[4], buf[2], 96) || unhex(tok[5], buf[3], 32)) { ++failures; continue; }
            rc = jjs_verify_single_ext(buf[0], buf[1], buf[2], buf[3], 1, status, tally);
        } else if (!strcmp(tok[0], "double_ext") && n == 8) {
            if (unhex(tok[2], buf[0], 32) || unhex(tok[3], buf[1], 96) || unhex(tok[4], buf[2], 96) || unhex(tok[5], buf[3], 96) ||
                unhex(tok[6], buf[4], 96) || unhex(tok[7], buf[5], 32)) { ++failures; continue; }
            rc = jjs_verify_double_ext(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], 1, status, tally);
        } else if (!strcmp(tok[0], "vargen_ext") && n == 7) {
            if (unhex(tok[2], buf[0], 32) || unhex(tok[3], buf[1], 96) || unhex(tok[4], buf[2], 96) || unhex(tok[5], buf[3], 96) ||
                unhex(tok[6], buf[4], 32)) { ++failures; continue; }
            rc = jjs_verify_vargen_ext(buf[0], buf[1], buf[2], buf[3], buf[4], 1, status, tally);
        } else if (!strcmp(tok[0], "single_wire") && n == 5) {   /* sig (u || R) pk m */
            if (unhex(tok[2], buf[0], 64) || unhex(tok[3], buf[1], 32) || unhex(tok[4], buf[2], 32)) { ++failures; continue; }
            rc = jjs_verify_single_wire(buf[0], buf[1], buf[2], 1, status, tally);
        } else if (!strcmp(tok[0], "double_wire") && n == 5) {   /* sig (u || R || R') pk (pk || pk') m */
            if (unhex(tok[2], buf[0], 96) || unhex(tok[3], buf[1], 64) || unhex(tok[4], buf[2], 32)) { ++failures; continue; }
            rc = jjs_verify_double_wire(buf[0], buf[1], buf[2], 1, status, tally);
        } else if (!strcmp(tok[0], "vargen_wire") && n == 5) {   /* sig (u || R) pk (pk || generator) m */
            if (unhex(tok[2], buf[0], 64) || unhex(tok[3], buf[1], 64) || unhex(tok[4], buf[2], 32)) { ++failures; continue; }
            rc = jjs_verify_vargen_wire(buf[0], buf[1], buf[2], 1, status, tally);
        } else { ++failures; continue; }
        if (rc != JJS_OK || status[0] != want || tally[want] != 1) {
            fprintf(stderr, "line %d: rc %d status %d want %d (%s)\n", total, rc, status[0], want, jjs_last_error());
            ++failures;
        }
    }
    fclose(f);
    jjs_shutdown();
    printf("%d vectors, %d failures\n", total, failures);
    return failures ? 1 : 0;
}

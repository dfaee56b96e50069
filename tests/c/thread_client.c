/* A multithreaded C99 client of include/jjs_gpu.h: what a service with a thread per request does to the engine, without
 * an interpreter lock between the threads.
 *
 *   thread_client <batch file> <threads,threads,...> <calls per thread> [check]
 *
 * The batch file holds one or more batches (written by tests/test_host_threads_gpu.py / tools/small_host_calls.py):
 *   magic "JJSB", u32 n_batches; per batch: u32 scheme (0 single, 1 double, 2 vargen), u32 format (0 affine, 1 ext, 2 wire),
 *   u32 n_items, u32 n_cols, n_cols x u32 width, then the columns (n_items x width each), then n_items expected status bytes.
 * Thread t calls batch (t + call) mod n_batches through the blocking host-buffer entry point of its scheme and format and
 * compares every status byte and the tally with the expectation.  For every thread count T of the list one JSON line:
 *   {"threads": T, "calls_per_s": ..., "items_per_s": ..., "mismatches": 0, "lane_launches": ..., "lane_calls": ...}
 * Test infrastructure, not product. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "jjs_gpu.h"

typedef struct {
    uint32_t scheme, format, n, n_cols, width[8];
    uint8_t* col[8];
    uint8_t* want;
    uint64_t want_tally[4];
} batch_t;

static batch_t* batches;
static uint32_t n_batches;
static int calls_per_thread, rotate;
static pthread_barrier_t start_line;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int verify(const batch_t* b, uint8_t* status, uint64_t tally[4]) {
    uint8_t* const* c = b->col;
    switch (b->scheme * 3 + b->format) {
    case 0: return jjs_verify_single(c[0], c[1], c[2], c[3], b->n, status, tally);
    case 1: return jjs_verify_single_ext(c[0], c[1], c[2], c[3], b->n, status, tally);
    case 2: return jjs_verify_single_wire(c[0], c[1], c[2], b->n, status, tally);
    case 3: return jjs_verify_double(c[0], c[1], c[2], c[3], c[4], c[5], b->n, status, tally);
    case 4: return jjs_verify_double_ext(c[0], c[1], c[2], c[3], c[4], c[5], b->n, status, tally);
    case 5: return jjs_verify_double_wire(c[0], c[1], c[2], b->n, status, tally);
    case 6: return jjs_verify_vargen(c[0], c[1], c[2], c[3], c[4], b->n, status, tally);
    case 7: return jjs_verify_vargen_ext(c[0], c[1], c[2], c[3], c[4], b->n, status, tally);
    case 8: return jjs_verify_vargen_wire(c[0], c[1], c[2], b->n, status, tally);
    default: return JJS_ERR_ARG;
    }
}

typedef struct { int id; long mismatches, errors; } worker_t;

static void* worker(void* arg) {
    worker_t* w = (worker_t*)arg;
    uint32_t most = 0, i;
    uint8_t* status;
    int c;
    for (i = 0; i < n_batches; ++i) most = batches[i].n > most ? batches[i].n : most;
    status = (uint8_t*)aligned_alloc(64, ((size_t)most + 63) & ~(size_t)63);
    pthread_barrier_wait(&start_line);
    for (c = 0; c < calls_per_thread; ++c) {
        const batch_t* b = &batches[rotate ? ((uint32_t)w->id + (uint32_t)c) % n_batches : (uint32_t)w->id % n_batches];
        uint64_t tally[4] = {0, 0, 0, 0};
        if (verify(b, status, tally) != JJS_OK) { ++w->errors; continue; }
        if (memcmp(status, b->want, b->n) != 0 || memcmp(tally, b->want_tally, sizeof tally) != 0) ++w->mismatches;
    }
    free(status);
    return NULL;
}

static int load(const char* path) {
    FILE* f = fopen(path, "rb");
    char magic[4];
    uint32_t i, k;
    if (!f) return -1;
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "JJSB", 4) != 0 || fread(&n_batches, 4, 1, f) != 1 || n_batches == 0) return -1;
    batches = (batch_t*)calloc(n_batches, sizeof(batch_t));
    for (i = 0; i < n_batches; ++i) {
        batch_t* b = &batches[i];
        if (fread(&b->scheme, 4, 1, f) != 1 || fread(&b->format, 4, 1, f) != 1 || fread(&b->n, 4, 1, f) != 1 || fread(&b->n_cols, 4, 1, f) != 1) return -1;
        if (b->scheme > 2 || b->format > 2 || b->n_cols > 8) return -1;
        if (fread(b->width, 4, b->n_cols, f) != b->n_cols) return -1;
        for (k = 0; k < b->n_cols; ++k) {
            const size_t bytes = (size_t)b->n * b->width[k];
            b->col[k] = (uint8_t*)aligned_alloc(64, (bytes + 63) & ~(size_t)63);
            if (fread(b->col[k], 1, bytes, f) != bytes) return -1;
        }
        b->want = (uint8_t*)malloc(b->n ? b->n : 1);
        if (fread(b->want, 1, b->n, f) != b->n) return -1;
        for (k = 0; k < b->n; ++k)
            if (b->want[k] < 4) ++b->want_tally[b->want[k]];
    }
    fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    char* list;
    char* tok;
    uint32_t i;
    if (argc < 4) { fprintf(stderr, "usage: %s <batch file> <threads,...> <calls per thread> [rotate]\n", argv[0]); return 2; }
    if (load(argv[1]) != 0) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    calls_per_thread = atoi(argv[3]);
    rotate = argc > 4 && !strcmp(argv[4], "rotate");
    if (jjs_init(1) != JJS_OK) { fprintf(stderr, "jjs_init: %s\n", jjs_last_error()); return 2; }
    for (i = 0; i < n_batches; ++i) {         /* first calls: buffers grow; and every batch is right when it is alone */
        uint8_t* st = (uint8_t*)aligned_alloc(64, ((size_t)batches[i].n + 63) & ~(size_t)63);
        uint64_t tally[4];
        int rc = verify(&batches[i], st, tally), r2 = verify(&batches[i], st, tally);
        if (rc != JJS_OK || r2 != JJS_OK || memcmp(st, batches[i].want, batches[i].n) != 0 || memcmp(tally, batches[i].want_tally, sizeof tally) != 0) {
            fprintf(stderr, "batch %u: wrong when alone (%s)\n", i, jjs_last_error());
            return 1;
        }
        free(st);
    }
    list = strdup(argv[2]);
    for (tok = strtok(list, ","); tok; tok = strtok(NULL, ",")) {
        const int T = atoi(tok);
        pthread_t th[64];
        worker_t w[64];
        uint64_t s0[JJS_PATH_STATS], s1[JJS_PATH_STATS];
        double t0, dt;
        long mismatches = 0, errors = 0;
        uint64_t items = 0;
        int t;
        if (T < 1 || T > 64) continue;
        pthread_barrier_init(&start_line, NULL, (unsigned)T + 1);
        for (t = 0; t < T; ++t) { w[t].id = t; w[t].mismatches = 0; w[t].errors = 0; pthread_create(&th[t], NULL, worker, &w[t]); }
        jjs_path_stats(s0);
        pthread_barrier_wait(&start_line);
        t0 = now_s();
        for (t = 0; t < T; ++t) pthread_join(th[t], NULL);
        dt = now_s() - t0;
        jjs_path_stats(s1);
        pthread_barrier_destroy(&start_line);
        for (t = 0; t < T; ++t) {
            int c;
            mismatches += w[t].mismatches; errors += w[t].errors;
            for (c = 0; c < calls_per_thread; ++c) items += batches[rotate ? ((uint32_t)t + (uint32_t)c) % n_batches : (uint32_t)t % n_batches].n;
        }
        printf("{\"threads\": %d, \"calls_per_thread\": %d, \"calls_per_s\": %.1f, \"items_per_s\": %.0f, \"mismatches\": %ld, \"errors\": %ld, "
               "\"lane_launches\": %llu, \"lane_calls\": %llu}\n",
               T, calls_per_thread, (double)T * calls_per_thread / dt, (double)items / dt, mismatches, errors,
               (unsigned long long)(s1[JJS_PATH_LANE_LAUNCHES] - s0[JJS_PATH_LANE_LAUNCHES]),
               (unsigned long long)(s1[JJS_PATH_LANE_CALLS] - s0[JJS_PATH_LANE_CALLS]));
        fflush(stdout);
        if (mismatches || errors) return 1;
    }
    jjs_shutdown();
    return 0;
}

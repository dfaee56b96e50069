/*
 * Extra entry points of the PROFILING build of the engine (jubjub_schnorr_amd/libjjs_gpu_prof.so, compiled from
 * the same sources with -DJJS_PROFILING).  They switch verification work off or let several logical devices share
 * one card, so they are compiled OUT of the product library libjjs_gpu.so (tests/test_abi.py checks that the
 * symbols are absent there).  Used only by jubjub_schnorr_amd/tools/phase_profile.py and tests/multidevice_child.py.
 */
#ifndef JJS_GPU_PROFILING_H
#define JJS_GPU_PROFILING_H

#include "jjs_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Ablations (results become meaningless): skip phases of the verify kernels in later launches; bit0 = point
 * validity, bit1 = challenge hash, bit2 = equations, bit3 = Euclid (stand-in scalars), bit4 = every window / comb
 * lookup of the throughput path, and every comb lookup of the key-table path, reads from a cache-resident subset of
 * its table (what the gathers cost); 0 restores the full path. */
int jjs_debug_skip_phases(unsigned mask);
/* Path selection for A/B timing (results stay exact): 0 = by batch size and key repetition (product behaviour),
 * 1 = never the latency path, 3 = never the latency path and never the key tables (every key a fresh variable
 * point), 2 = the latency path for every single / double call of at most 16 384 items; 0x42 / 0x82 = the
 * latency path with the scalars cut into 4 / 8 pieces whatever the size; 0x500 = 5-bit windows whenever the key
 * tables engage (the product takes 6-bit windows from 128 signatures per key); 0x1000 = the key-table path takes the
 * items in the caller's order instead of grouping them by key. */
int jjs_debug_force_path(int which);
/* Test mode for boxes with one GPU: a later jjs_init(k) with k above the visible device count creates k logical
 * devices (own stream, tables, workspace, staging each) that share the visible cards round-robin; the tallies are
 * then summed on the host, since two ranks on one card cannot form an RCCL clique. */
int jjs_debug_allow_virtual_devices(int allow);
/* on != 0: the next calls find that the pool of the per-key tables "cannot be allocated" (the branch a device short of
 * memory takes): they run the throughput path; 0 restores the product behaviour. */
int jjs_debug_fail_key_arena(int on);
/* on != 0: the dedup hash of the key tables runs with seed 0 instead of a fresh seed per call, so that a test can
 * present keys crafted to collide in it (what the seed keeps a sender from doing). */
int jjs_debug_pin_hash_seed(int on);
/* Where the last host-buffer call on one device spent its host time, in seconds: out[0] waiting for the staging copy
 * of a piece to finish (it runs one piece ahead on helper threads), out[1] starting the next one (includes waiting for
 * its pinned slot), out[2] the whole block, out[3] pieces, out[4] from the entry to the first upload being queued,
 * out[5] until everything was queued, out[6] waiting for the device to drain after that, out[7] copying the statuses
 * out. */
int jjs_debug_host_timing(double out[8]);

#ifdef __cplusplus
}
#endif
#endif /* JJS_GPU_PROFILING_H */

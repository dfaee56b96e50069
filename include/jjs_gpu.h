/*
 * libjjs_gpu -- batch Schnorr-on-JubJub verification on AMD MI355X (gfx950).
 *
 * C ABI that a host-language shim binds (Rust `extern "C"`, ctypes, ...).  It is the batch
 * drop-in for the `verify` hot path of dusk-network/jubjub-schnorr:
 *
 *   jjs_verify_single*  <->  PublicKey::verify        reference src/keys/public.rs:114-135
 *   jjs_verify_double*  <->  PublicKeyDouble::verify  reference src/keys/public/double.rs:86-117
 *   jjs_verify_vargen*  <->  PublicKeyVarGen::verify  reference src/keys/public/var_gen.rs:107-133
 *   status codes        <->  Result<(), Error>        reference src/error.rs:13-19
 *   jjs_challenge_*     <->  challenge_hash           reference src/signatures.rs:122-140,
 *                             src/signatures/double.rs:151-177, src/signatures/var_gen.rs:121-142
 *   jjs_sign_*          <->  SecretKey::sign & co.    reference src/keys/secret.rs:174-194,
 *                             src/keys/secret/double.rs:56-85, src/keys/secret/var_gen.rs:228-256
 *                             (test-vector / benchmark-input generator: NOT constant time)
 *
 * Data layout (all entry points): structure of arrays, item i of an array at offset i*size.
 *   field element / scalar : 32 bytes, little-endian, canonical (BlsScalar::to_bytes,
 *                            JubJubScalar::to_bytes)
 *   point                  : 64 bytes = affine u || v, each 32 bytes little-endian canonical
 *                            (what JubJubExtended::to_hash_inputs() returns, reference
 *                            src/signatures.rs:127-128)
 * Every buffer must be 16-byte aligned.
 *
 * status[i]: 0 = Ok, 1 = InvalidPoint, 2 = InvalidSignature (reference src/error.rs:17-19 with the
 * precedence of src/keys/public.rs:119-132), 3 = Malformed (a coordinate or message >= q, or
 * u >= r: unreachable through the Rust types, defined so that this ABI is total).
 * tally[k] = number of items with status k.
 *
 * Return value: 0 = success; negative = engine error, in which case outputs are unspecified:
 *   -1 bad argument, -2 HIP error, -3 collective (RCCL) error, -4 not initialised.
 * No exceptions, no aborts.  jjs_last_error() describes the last failure of the calling THREAD (the
 * buffer is thread-local: the pointer stays valid and is only rewritten by later failures of the same thread).
 *
 * Ownership: the caller owns every buffer passed in; the library keeps nothing after a blocking
 * call returns (after the stream has drained, for the *_dev calls).  The library owns its device
 * tables, workspaces, streams and RCCL communicators between jjs_init and jjs_shutdown.
 *
 * There are no switches, environment variables or hidden entry points that turn a check off: the profiling
 * ablations and the logical-device test mode live in a separate build (libjjs_gpu_prof.so, -DJJS_PROFILING,
 * include/jjs_gpu_profiling.h) that the product never loads.
 *
 * Threading: jjs_init / jjs_shutdown are not re-entrant.  All other calls may come from any host
 * thread.  One internal mutex guards the engine's state; it is held while a call is QUEUED, never while a call waits:
 * the *_dev calls are asynchronous, jjs_stream_sync waits outside it, and a blocking host-buffer call holds it only while
 * its launches are queued.  Host-buffer calls of at most 16 384 items run on staging lanes (eight per device), side by side
 * on the device; those of at most 4 096 items of one scheme and input format that arrive while another is running share
 * one launch (JJS_PATH_LANE_LAUNCHES / JJS_PATH_LANE_CALLS count them): four threads of 1 024-signature calls complete
 * about three times the calls per second of one thread, eight threads four times; 64 threads of ONE-signature calls about thirty
 * times (the callers of a shared launch other than the one that drives it sleep, so a service may have many more threads than
 * cores).  Larger host-buffer calls are pipelines of
 * uploads and launches that fill the device: they run one at a time per device (other threads' calls are queued meanwhile;
 * with several driven devices such a call holds the mutex for its duration).  A call takes one of the engine's call slots by size (six for calls of
 * at most 16 384 items, three for at most 131 072, two for larger ones): calls in different slots share no buffer and
 * overlap on the device when they are issued on different streams; calls in one slot are ordered on the device (each
 * waits for the previous one, also across streams).
 *
 * Method: the engine picks, per call, from the batch size and the repetition of its keys alone (no configuration):
 * at most 16 384 items -> latency path (a signature spread over many lanes: 0.37 ms single, 0.50 ms double, 0.41-0.47 ms
 * var-generator for a call of a few hundred items, 0.55 / 0.71 / 0.85 ms at 4 096); larger -> one signature per lane; at least 65 536 items (32 768 for double
 * and var-generator signatures) whose public keys (and per-item generators) repeat 16 times or more on average ->
 * per-key tables built inside the call.  The
 * status bytes are the same on every path.
 */
#ifndef JJS_GPU_H
#define JJS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JJS_OK 0
#define JJS_ERR_ARG (-1)
#define JJS_ERR_HIP (-2)
#define JJS_ERR_COLLECTIVE (-3)
#define JJS_ERR_NOT_INIT (-4)

#define JJS_STATUS_OK 0
#define JJS_STATUS_INVALID_POINT 1
#define JJS_STATUS_INVALID_SIGNATURE 2
#define JJS_STATUS_MALFORMED 3

/* Sets up the devices this process drives: builds the fixed-base tables for G and G' on each, allocates
 * the per-lane workspaces and one stream per device.
 *   device_count == 1 : the calling thread's current HIP device only (one process per GPU: what a
 *                       torch.distributed / torchrun rank uses);
 *   device_count == k : HIP devices 0 .. k-1 in this one process (2 <= k <= 16);
 *   device_count == 0 : every visible HIP device.
 * With several devices the host-buffer calls below cut each batch into one contiguous block of
 * ceil(n / devices) items per device, run the blocks concurrently and sum the four tally counters with
 * one RCCL all-reduce (4 x u64, the only collective; RCCL is loaded on demand and not at all for one
 * device).  The *_dev calls always act on the calling thread's current device, which must be one of the
 * devices set up here.  Idempotent for a repeated identical request. */
int jjs_init(int device_count);
void jjs_shutdown(void);
/* Number of devices the process drives (0 before jjs_init). */
int jjs_device_count(void);
/* Ranks of the in-library RCCL clique that sums the tallies of the host-buffer calls: the number of devices when
 * jjs_init set up more than one (real) device, else 0 (one device: nothing to sum). */
int jjs_collective_ranks(void);
const char* jjs_last_error(void);
/* ABI version: bumped on any signature change. */
int jjs_abi_version(void);

/* ---- host buffers, blocking: copy in, verify, copy out -------------------------------------- */
int jjs_verify_single(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* m, size_t n,
                      uint8_t* status, uint64_t tally[4]);
int jjs_verify_double(const uint8_t* u, const uint8_t* R, const uint8_t* R_prime, const uint8_t* PK,
                      const uint8_t* PK_prime, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]);
int jjs_verify_vargen(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen, const uint8_t* m,
                      size_t n, uint8_t* status, uint64_t tally[4]);

/* ---- device buffers (resident data), asynchronous on `stream` ------------------------------------
 * All pointers are device pointers on the engine's device.  `status` (n bytes) and `tally`
 * (4 x uint64, zeroed by the call) may each be NULL.  `stream` is a hipStream_t passed as void*
 * (NULL = the device's default stream, as in HIP).  The call enqueues and returns; use jjs_stream_sync or any
 * HIP synchronisation on that stream before reading the outputs. */
int jjs_verify_single_dev(const void* u, const void* R, const void* PK, const void* m, size_t n, void* status,
                          void* tally, void* stream);
int jjs_verify_double_dev(const void* u, const void* R, const void* R_prime, const void* PK, const void* PK_prime,
                          const void* m, size_t n, void* status, void* tally, void* stream);
int jjs_verify_vargen_dev(const void* u, const void* R, const void* PK, const void* Gen, const void* m, size_t n,
                          void* status, void* tally, void* stream);
int jjs_stream_sync(void* stream);

/* Which method the calls on the calling thread's device took since jjs_init (the engine chooses per call; see "Method"
 * above): out[k] = number of calls, k one of the JJS_PATH_* indices; out[JJS_PATH_KEY_POOL_BYTES] = device memory the
 * per-key tables hold right now.  A call that tried the key tables is counted under what the device decided for it
 * (WIDE / NARROW: tables with 6- / 5-bit windows; DO_NOT_REPEAT: fewer than 16 signatures per key on average;
 * PROBE_LIMIT: keys that collide in the dedup table; POOL_TOO_SMALL: the keys repeat but their tables did not fit -- the
 * pool has grown by the slot's next call; the last three ran the throughput path) once that call has finished; NO_MEMORY
 * counts allocations of the table pool that the device refused (those calls ran the throughput path, too).  So a caller
 * can see when its batches fall off the fast path.  Host-buffer calls count once per device block. */
#define JJS_PATH_LATENCY 0
#define JJS_PATH_THROUGHPUT 1
#define JJS_PATH_KEY_TABLES_WIDE 2
#define JJS_PATH_KEY_TABLES_NARROW 3
#define JJS_PATH_KEYS_DO_NOT_REPEAT 4
#define JJS_PATH_KEYS_PROBE_LIMIT 5
#define JJS_PATH_KEYS_POOL_TOO_SMALL 6
#define JJS_PATH_KEYS_NO_MEMORY 7
#define JJS_PATH_KEY_POOL_BYTES 8
/* blocking host-buffer calls of at most 131 072 items: launches on the staging lanes, and calls they served (calls of at
 * most 4 096 items of one scheme and format that arrive while such a launch runs share the next one) */
#define JJS_PATH_LANE_LAUNCHES 9
#define JJS_PATH_LANE_CALLS 10
#define JJS_PATH_STATS 11
int jjs_path_stats(uint64_t out[JJS_PATH_STATS]);

/* ---- pre-sizing, trimming -----------------------------------------------------------------------------------
 * The engine's buffers are grow-only and allocated on first use: the first call of a larger size than any before it
 * allocates (never waits for the device: a replaced buffer is kept until jjs_trim / jjs_shutdown), which costs that call
 * the allocation time.  A service that knows its call shapes pre-sizes at start-up:
 *   jjs_reserve(scheme, format, n_items, host_buffers) allocates what a verification call of that scheme (JJS_SCHEME_*),
 *   input format (JJS_FORMAT_*) and at most n_items items needs, in every call slot such a call can land in; with
 *   host_buffers != 0 also the staging of the blocking host-buffer entry point of that shape (for every driven device).
 * jjs_trim() waits for the devices to go idle and frees the retired buffers and the key-table pools (which come back,
 * at the size they had, with the next call that takes the key tables).  jjs_memory_stats: bytes held, by kind. */
#define JJS_SCHEME_SINGLE 0
#define JJS_SCHEME_DOUBLE 1
#define JJS_SCHEME_VARGEN 2
#define JJS_FORMAT_AFFINE 0
#define JJS_FORMAT_EXT 1
#define JJS_FORMAT_WIRE 2
int jjs_reserve(int scheme, int format, size_t n_items, int host_buffers);
int jjs_trim(void);
#define JJS_MEMORY_KEY_POOLS 0
#define JJS_MEMORY_SLOT_BUFFERS 1
#define JJS_MEMORY_HOST_STAGING 2
#define JJS_MEMORY_RETIRED 3
#define JJS_MEMORY_STATS 4
int jjs_memory_stats(uint64_t out[JJS_MEMORY_STATS]);

/* ---- wire formats (reference `to_bytes` / `from_bytes`), device buffers, asynchronous ------------------
 * Points travel compressed (32 bytes: little-endian v, parity of u in bit 255) and are decoded on the
 * device; an item with any undecodable point (v >= q, no square root, or u = 0 with the sign bit set)
 * or a scalar out of range gets status 3, where the Rust `from_bytes` would have returned an error.
 *   single : sig = n x 64 (u || R)        reference src/signatures.rs:104-119 ; pk = n x 32  src/keys/public.rs:83-93
 *   double : sig = n x 96 (u || R || R')  src/signatures/double.rs:126-148    ; pk = n x 64 (pk || pk')  src/keys/public/double.rs:169-186
 *   vargen : sig = n x 64 (u || R)        src/signatures/var_gen.rs:95-113    ; pk = n x 64 (pk || generator)  src/keys/public/var_gen.rs:54-79
 *   m      : n x 32 (BlsScalar::to_bytes)
 * 132 / 196 / 164 bytes per verification instead of 196 / 324 / 260. */
int jjs_verify_single_wire_dev(const void* sig, const void* pk, const void* m, size_t n, void* status, void* tally,
                               void* stream);
int jjs_verify_double_wire_dev(const void* sig, const void* pk, const void* m, size_t n, void* status, void* tally,
                               void* stream);
int jjs_verify_vargen_wire_dev(const void* sig, const void* pk, const void* m, size_t n, void* status, void* tally,
                               void* stream);
/* the same from host buffers, blocking (copy in, decode, verify, copy out) */
int jjs_verify_single_wire(const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]);
int jjs_verify_double_wire(const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]);
int jjs_verify_vargen_wire(const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]);
/* JubJubAffine::from_bytes / to_bytes in bulk: in n x 32 -> affine n x 64 + ok n bytes; affine n x 64 -> n x 32 */
int jjs_decompress_dev(const void* in, size_t n, void* affine_out, void* ok_out, void* stream);
int jjs_compress_dev(const void* affine, size_t n, void* out, void* stream);

/* ---- extended coordinates: what the Rust types hold --------------------------------------------------------
 * `PublicKey::verify(&self, &Signature, BlsScalar)` receives `JubJubExtended` points (reference src/keys/public.rs:
 * 114-118, src/signatures.rs:62-65) and normalises them itself with one field inversion per point
 * (`to_hash_inputs()`, src/signatures.rs:127-128).  These entry points take every point as 96 bytes =
 * U || V || Z (three canonical field elements; the affine point is (U/Z, V/Z); `get_u()/get_v()/get_z()` of the
 * Rust type -- reference tests/keys.rs:48-50 -- through `BlsScalar::to_bytes`) and normalise on the device with one inversion shared by many items,
 * so a shim does no field arithmetic at all.  Scalars and m as everywhere else.  Same statuses; additionally a
 * coordinate >= q gives 3 and Z = 0 (not a curve point in any representation) gives 1.  Argument order as the
 * affine entry points. */
int jjs_verify_single_ext_dev(const void* u, const void* R_ext, const void* PK_ext, const void* m, size_t n, void* status,
                              void* tally, void* stream);
int jjs_verify_double_ext_dev(const void* u, const void* R_ext, const void* R_prime_ext, const void* PK_ext,
                              const void* PK_prime_ext, const void* m, size_t n, void* status, void* tally, void* stream);
int jjs_verify_vargen_ext_dev(const void* u, const void* R_ext, const void* PK_ext, const void* Gen_ext, const void* m, size_t n,
                              void* status, void* tally, void* stream);
/* the same from host buffers, blocking */
int jjs_verify_single_ext(const uint8_t* u, const uint8_t* R_ext, const uint8_t* PK_ext, const uint8_t* m, size_t n,
                          uint8_t* status, uint64_t tally[4]);
int jjs_verify_double_ext(const uint8_t* u, const uint8_t* R_ext, const uint8_t* R_prime_ext, const uint8_t* PK_ext,
                          const uint8_t* PK_prime_ext, const uint8_t* m, size_t n, uint8_t* status, uint64_t tally[4]);
int jjs_verify_vargen_ext(const uint8_t* u, const uint8_t* R_ext, const uint8_t* PK_ext, const uint8_t* Gen_ext, const uint8_t* m,
                          size_t n, uint8_t* status, uint64_t tally[4]);

/* ---- multisignature: batch verify_share / combine (reference src/multisig.rs:284-387, 440-500) -------
 * Transcript t owns participants [offsets[t], offsets[t+1]) of the flattened device arrays z (N x 32),
 * PK, R, S (N x 64 affine); m is B x 32.  `offsets` is a HOST array of B + 1 non-decreasing entries starting at 0.
 * A transcript may have any number of participants (the reference takes any non-empty transcript); one without
 * participants gets transcript_status 5, the reference's InvalidMultisigTranscript, and does not affect the others.
 * (The two sponge tags of a transcript of more than 256 participants are computed on the device inside the call.  The hashes
 * of a transcript are sponge chains -- (2 + 2n) / 4 permutations for each of the n delinearisation hashes, which run side by
 * side, and (3 + 4n) / 4 for the binding hash, which is ONE chain -- so the time of a call grows linearly with its longest
 * transcript: about 0.2 ms per participant, 1 000 participants 0.21-0.24 s; a call with few items runs these chains on eight
 * lanes each.)
 * Outputs (device): share_status[i] = 0 when z_i*G + (c*d_i)*PK_i == R_i + a*S_i, 4 (InvalidMultisigShare)
 * when not, 3 for a non-canonical encoding (z_i, a coordinate, or the transcript's m); transcript_status[t]
 * (B bytes, nullable) = 0 when `combine` returns a signature, else the status of the transcript's first failing
 * share (the reference's `combine` stops there, src/multisig.rs:340-353), or 5; agg_pk[t] = aggregate_pk(pk_vec) (64 B
 * affine); and what `combine` returns: sig_u[t] = sum z_i (32 B), sig_R[t] = RSa (64 B affine) -- both all-zero
 * for a transcript whose status is not 0 (no signature comes out of bad shares).
 * Like the reference, the points are not validated.  Asynchronous on `stream`. */
#define JJS_STATUS_INVALID_SHARE 4
#define JJS_STATUS_INVALID_TRANSCRIPT 5
int jjs_multisig_combine_dev(const void* z, const void* PK, const void* R, const void* S, const void* m,
                             const uint32_t* offsets_host, size_t n_transcripts, void* share_status, void* transcript_status,
                             void* agg_pk, void* sig_u, void* sig_R, void* stream);

/* ---- transcript parity (debug export): c_out = n x 32 bytes, the 250-bit challenge per item ---- */
int jjs_challenge_single_dev(const void* R, const void* PK, const void* m, size_t n, void* c_out, void* stream);
int jjs_challenge_double_dev(const void* R, const void* R_prime, const void* PK, const void* PK_prime, const void* m,
                             size_t n, void* c_out, void* stream);
int jjs_challenge_vargen_dev(const void* R, const void* PK, const void* Gen, const void* m, size_t n, void* c_out,
                             void* stream);

/* ---- key derivation (reference `PublicKey::from(&SecretKey)` src/keys/public.rs:54-60, `PublicKeyDouble::from`
 * src/keys/public/double.rs:47-57): PK[i] = sk[i] * G and, when PKp_out is not NULL, PK'[i] = sk[i] * G'
 * (64 B affine each).  bad_out (nullable, n bytes) is set to 1 where sk[i] >= r.  Fixed-base comb, device
 * pointers, asynchronous on `stream`.  NOT constant time: for public test material, not for live secrets. */
int jjs_public_keys_dev(const void* sk, size_t n, void* PK_out, void* PKp_out, void* bad_out, void* stream);

/* ---- signing: generator of synthetic inputs (NOT constant time, not for production keys) -------
 * sk, rnd: scalars < r; m: field element < q.  rnd is the RNG draw the reference's hedged nonce
 * mixes in (reference src/nonce.rs:32-44).  Outputs: u (n x 32), points (n x 64 affine). */
int jjs_sign_single_dev(const void* sk, const void* rnd, const void* m, size_t n, void* u_out, void* R_out,
                        void* PK_out, void* stream);
int jjs_sign_double_dev(const void* sk, const void* rnd, const void* m, size_t n, void* u_out, void* R_out,
                        void* R_prime_out, void* PK_out, void* PK_prime_out, void* stream);
/* gen_scalar: per-item generator = gen_scalar * G (reference src/keys/secret/var_gen.rs:162-172) */
int jjs_sign_vargen_dev(const void* sk, const void* gen_scalar, const void* rnd, const void* m, size_t n, void* u_out,
                        void* R_out, void* PK_out, void* Gen_out, void* stream);

/* ---- primitives exposed for parity tests ------------------------------------------------------ */
/* out[i] = a[i] * b[i] mod q (canonical bytes in and out) */
int jjs_debug_fq_mul_dev(const void* a, const void* b, size_t n, void* out, void* stream);
/* out[i] = untruncated Poseidon digest of the k field elements at in[(i*k + j)*32] */
int jjs_debug_poseidon_dev(const void* in, size_t k, size_t n, void* out, void* stream);
/* out[i] bit0 = on curve, bit1 = torsion free (pairing test, as used by verify; identity counts as
 * torsion free), bit2 = identity, bit3 = torsion free by the reference's definition [r]P == O */
int jjs_debug_point_flags_dev(const void* points, size_t n, void* out, void* stream);
/* The half-size scalars the verify kernels derive from a challenge (csrc/verify_core.h half_size_scalars, the
 * device code path): for c[i] (n x 32 bytes, canonical, < r) a_out[i], b_out[i] (n x 16 bytes each, little-endian)
 * and b_neg_out[i] (n bytes) with a = +-b*c (mod r), a, |b| < 2^126: Euclid's algorithm on (r, c) stopped at the
 * first remainder below 2^126.  Device pointers, 16-byte aligned. */
int jjs_debug_half_scalars_dev(const void* c, size_t n, void* a_out, void* b_out, void* b_neg_out, void* stream);
/* copies the fixed-base table of G (which = 0) or G' (which = 1) to host memory; size in bytes via
 * jjs_debug_comb_table_bytes() */
size_t jjs_debug_comb_table_bytes(void);
int jjs_debug_comb_table(int which, void* host_out);
/* Loads RCCL, forms a one-rank clique on the current device and all-reduces a known 4 x u64 vector on the
 * engine stream: the call sequence of the multi-device tally reduction, runnable with a single GPU. */
int jjs_debug_rccl_selftest(void);

#ifdef __cplusplus
}
#endif
#endif /* JJS_GPU_H */

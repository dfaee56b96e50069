// Host-side construction of the wave-uniform scheme descriptors (verify_params) for the three
// reference schemes.  Argument order and meaning follow include/jjs_gpu.h.
//   single : PublicKey::verify        /root/reference/src/keys/public.rs:114-135
//   double : PublicKeyDouble::verify  src/keys/public/double.rs:86-117
//   vargen : PublicKeyVarGen::verify  src/keys/public/var_gen.rs:107-133
#pragma once
#include "verify_core.h"

namespace jjs {

inline fe_src pt_src(const uint8_t* base) { return fe_src{base, 64, 0}; }
inline fe_src coord_src(const uint8_t* base, uint32_t off) { return fe_src{base, 64, off}; }
inline fe_src fe32_src(const uint8_t* base) { return fe_src{base, 32, 0}; }

struct out_ptrs {
    uint8_t* status;
    unsigned long long* tally;
    uint8_t* c_out;
    uint32_t* workspace;
};

// transcript R.u, R.v, PK.u, PK.v, m  (src/signatures.rs:130-139)
inline verify_params params_single(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* m, uint64_t n,
                                   const uint32_t* comb_g, const out_ptrs& o) {
    verify_params P{};
    P.n_hash = 5; P.n_points = 2; P.n_eq = 1;
    P.hash_in[0] = coord_src(R, 0); P.hash_in[1] = coord_src(R, 32);
    P.hash_in[2] = coord_src(PK, 0); P.hash_in[3] = coord_src(PK, 32);
    P.hash_in[4] = fe32_src(m);
    P.points[0] = pt_src(PK); P.points[1] = pt_src(R);
    P.resolve_lanes = 2; P.resolve_lanes_keyed = 1;      // R and PK; on the key-table path R alone
    P.eq[0] = eq_desc{comb_g, fe_src{nullptr, 0, 0}, pt_src(PK), pt_src(R), 0, -1};
    P.key_points_mask = 1u;          // points[0] = PK
    P.u = fe32_src(u);
    P.n = n; P.status = o.status; P.tally = o.tally; P.c_out = o.c_out; P.workspace = o.workspace;
    return P;
}
// transcript TAG, R, R', PK, PK', m  (src/signatures/double.rs:162-176); tag = one shared element
inline verify_params params_double(const uint8_t* u, const uint8_t* R, const uint8_t* Rp, const uint8_t* PK,
                                   const uint8_t* PKp, const uint8_t* m, uint64_t n, const uint8_t* tag_words,
                                   const uint32_t* comb_g, const uint32_t* comb_gn, const out_ptrs& o) {
    verify_params P{};
    P.n_hash = 10; P.n_points = 4; P.n_eq = 2;
    P.hash_in[0] = fe_src{tag_words, 0, 0};
    P.hash_in[1] = coord_src(R, 0); P.hash_in[2] = coord_src(R, 32);
    P.hash_in[3] = coord_src(Rp, 0); P.hash_in[4] = coord_src(Rp, 32);
    P.hash_in[5] = coord_src(PK, 0); P.hash_in[6] = coord_src(PK, 32);
    P.hash_in[7] = coord_src(PKp, 0); P.hash_in[8] = coord_src(PKp, 32);
    P.hash_in[9] = fe32_src(m);
    P.points[0] = pt_src(PK); P.points[1] = pt_src(PKp); P.points[2] = pt_src(R); P.points[3] = pt_src(Rp);
    P.resolve_lanes = 4; P.resolve_lanes_keyed = 2;      // R, R', PK, PK'; on the key-table path R and R'
    P.eq[0] = eq_desc{comb_g, fe_src{nullptr, 0, 0}, pt_src(PK), pt_src(R), 0, -1};
    P.eq[1] = eq_desc{comb_gn, fe_src{nullptr, 0, 0}, pt_src(PKp), pt_src(Rp), 1, -1};
    P.key_points_mask = 3u;          // points[0] = PK, points[1] = PK'
    P.u = fe32_src(u);
    P.n = n; P.status = o.status; P.tally = o.tally; P.c_out = o.c_out; P.workspace = o.workspace;
    return P;
}
// transcript R, PK, Gen, m  (src/signatures/var_gen.rs:130-141)
inline verify_params params_vargen(const uint8_t* u, const uint8_t* R, const uint8_t* PK, const uint8_t* Gen,
                                   const uint8_t* m, uint64_t n, const out_ptrs& o) {
    verify_params P{};
    P.n_hash = 7; P.n_points = 3; P.n_eq = 1;
    P.hash_in[0] = coord_src(R, 0); P.hash_in[1] = coord_src(R, 32);
    P.hash_in[2] = coord_src(PK, 0); P.hash_in[3] = coord_src(PK, 32);
    P.hash_in[4] = coord_src(Gen, 0); P.hash_in[5] = coord_src(Gen, 32);
    P.hash_in[6] = fe32_src(m);
    P.points[0] = pt_src(PK); P.points[1] = pt_src(Gen); P.points[2] = pt_src(R);
    P.own_test_mask = 3u;            // PK and Gen are tested on their own; R rides on the equation
    P.resolve_lanes = 1; P.resolve_lanes_keyed = 1;
    P.eq[0] = eq_desc{nullptr, pt_src(Gen), pt_src(PK), pt_src(R), 0, 1};
    P.key_points_mask = 3u;          // points[0] = PK, points[1] = Gen
    P.u = fe32_src(u);
    P.n = n; P.status = o.status; P.tally = o.tally; P.c_out = o.c_out; P.workspace = o.workspace;
    return P;
}

}  // namespace jjs

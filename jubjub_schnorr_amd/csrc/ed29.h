// JubJub twisted Edwards curve  -u^2 + v^2 = 1 + d u^2 v^2  over Fq, extended coordinates.
//
// Replaces what the reference gets from dusk-jubjub 0.15 (Cargo.toml:26): `JubJubExtended`
// `+`, `* scalar`, `eq`, `is_identity`, `is_on_curve` behind
// /root/reference/src/keys/public.rs:128-130,159-164.  Unified a = -1 formulas; d is a
// non-square so they are complete and no input needs special-casing.  Bounds on every
// intermediate are enforced by the fe<L, A> types of fq29.h.
#pragma once
#include "fq29.h"

namespace jjs {

using fe_t = fe<1, 5>;  // a cached-addend coordinate (normalised limbs, value < 5q)

struct ext_pt {   // (X : Y : Z : T), x = X/Z, y = Y/Z, T = XY/Z
    fe_n x, y, z, t;
};
struct niels_pt {  // cached addend (Y+X, Y-X, Z, 2dT); Z = R' for affine entries
    fe_t ypx, ymx, z, t2d;
};

JJS_HD fe_n fe_n_one() { return fq_as<1, 2>(fq_one()); }
JJS_HD fe_n fe_n_zero() { return fq_as<1, 2>(fq_zero()); }

JJS_HD ext_pt ext_identity() {
    ext_pt p;
    p.x = fe_n_zero(); p.y = fe_n_one(); p.z = fe_n_one(); p.t = fe_n_zero();
    return p;
}
JJS_HD ext_pt ext_from_affine(const fe_n& u, const fe_n& v) {
    ext_pt p;
    p.x = u; p.y = v; p.z = fe_n_one(); p.t = fq_mul(u, v);
    return p;
}
JJS_HD niels_pt niels_identity() {
    niels_pt n;
    n.ypx = fq_as<1, 5>(fq_one()); n.ymx = n.ypx; n.z = n.ypx; n.t2d = fq_as<1, 5>(fq_zero());
    return n;
}
JJS_HD niels_pt to_niels(const ext_pt& p) {
    niels_pt n;
    n.ypx = fq_as<1, 5>(fq_norm(fq_add(p.y, p.x)));
    n.ymx = fq_norm(fq_sub(p.y, p.x));
    n.z = fq_as<1, 5>(p.z);
    n.t2d = fq_as<1, 5>(fq_mul(p.t, fe_from_const<1, 1>(JJS_D2)));
    return n;
}
JJS_HD niels_pt niels_select(bool c, const niels_pt& a, const niels_pt& b) {
    niels_pt r;
    r.ypx = fq_select(c, a.ypx, b.ypx); r.ymx = fq_select(c, a.ymx, b.ymx);
    r.z = fq_select(c, a.z, b.z); r.t2d = fq_select(c, a.t2d, b.t2d);
    return r;
}

// 2P: 3S + 5M (4M when the caller does not need T, i.e. the next operation is a doubling).
// E = 2XY is a product, not (X+Y)^2 - X^2 - Y^2: one multiply costs about what the two extra
// subtractions and their carry propagation would, and keeps E below 4q.
JJS_HD ext_pt ext_double(const ext_pt& p, bool need_t) {
    // The inlined product overwrites its first operand, so an operand that is still needed afterwards costs a
    // 9-register copy: each value below is the first operand of the LAST product that uses it.
    auto e = fq_dbl(fq_mul_hot(p.x, p.y));         // 2XY             <2,4>   (copies X once)
    fe_n xx = fq_sqr_hot(p.x);
    fe_n yy = fq_sqr_hot(p.y);
    auto c2 = fq_dbl(fq_sqr_hot(p.z));             // 2Z^2            <2,4>
    auto g = fq_norm(fq_add(yy, xx));          // Y^2 + X^2       <1,4>: normalised so that g*h and g*e fit the product
    auto h = fq_sub(yy, xx);                   // Y^2 - X^2       <3,5>: used in products only, no carry pass
    auto f = fq_norm(fq_sub(fq_add(c2, xx), yy));   // 2Z^2 - (Y^2 - X^2) = (2Z^2 + X^2) - Y^2   <1,9>
    ext_pt r;
    r.x = fq_mul_hot(e, f);                    // e survives only when T is wanted
    r.z = fq_mul_hot(f, h);
    r.y = fq_mul_hot(h, g);
    if (need_t) r.t = fq_mul_hot(g, e); else r.t = fe_n_zero();
    return r;
}

// P + N: 8M (7M without T).  neg adds -N instead (swap the two sums, negate 2dT).
JJS_HD ext_pt ext_add_niels(const ext_pt& p, const niels_pt& n, bool neg, bool need_t) {
    fe_t n_ymx = fq_select(neg, n.ypx, n.ymx);
    fe_t n_ypx = fq_select(neg, n.ymx, n.ypx);
    auto n_t2d = fq_select(neg, fq_neg(n.t2d), fq_as<2, 6>(n.t2d));   // <2,6>
    fe_n a = fq_mul_hot(fq_sub(p.y, p.x), n_ymx);      // <3,5> x <1,5>
    fe_n b = fq_mul_hot(fq_add(p.y, p.x), n_ypx);      // <2,4> x <1,5>
    fe_n c = fq_mul_hot(p.t, n_t2d);                   // <1,2> x <2,6>
    auto d = fq_dbl(fq_mul_hot(p.z, n.z));             // <2,4>
    auto e = fq_sub(b, a);                         // <3,5>
    auto f = fq_norm(fq_sub(d, c));                // <1,7>
    auto g = fq_add(d, c);                         // <3,6>
    auto h = fq_norm(fq_add(b, a));                // <1,4>: normalised for g*h and h*e
    ext_pt r;
    r.x = fq_mul_hot(e, f);                    // same ordering rule as in ext_double
    r.z = fq_mul_hot(f, g);
    r.y = fq_mul_hot(g, h);
    if (need_t) r.t = fq_mul_hot(h, e); else r.t = fe_n_zero();
    return r;
}
// P + N where N is affine (Z2 = 1): 7M (6M without T)
JJS_HD ext_pt ext_add_affine_niels(const ext_pt& p, const fe_t& n_ypx, const fe_t& n_ymx, const fe_t& n_t2d,
                                   bool need_t) {
    fe_n a = fq_mul_hot(fq_sub(p.y, p.x), n_ymx);
    fe_n b = fq_mul_hot(fq_add(p.y, p.x), n_ypx);
    fe_n c = fq_mul_hot(p.t, n_t2d);
    auto d = fq_dbl(p.z);                          // <2,4>
    auto e = fq_sub(b, a);
    auto f = fq_norm(fq_sub(d, c));
    auto g = fq_add(d, c);
    auto h = fq_norm(fq_add(b, a));
    ext_pt r;
    r.x = fq_mul_hot(e, f);
    r.z = fq_mul_hot(f, g);
    r.y = fq_mul_hot(g, h);
    if (need_t) r.t = fq_mul_hot(h, e); else r.t = fe_n_zero();
    return r;
}

JJS_HD bool ext_is_identity(const ext_pt& p) { return fq_is_zero(p.x) && fq_eq(p.y, p.z); }

// affine curve equation: is_on_curve at the boundary, where Z = 1
JJS_HD bool affine_on_curve(const fe_n& u, const fe_n& v) {
    fe_n u2 = fq_sqr(u), v2 = fq_sqr(v);
    auto lhs = fq_sub(v2, u2);                                                     // <3,5>
    auto rhs = fq_norm(fq_add(fq_mul(fq_mul(u2, v2), fe_from_const<1, 1>(JJS_D)), fq_one()));  // <1,3>
    return fq_eq(lhs, rhs);
}
JJS_HD bool affine_is_identity(const fe_n& u, const fe_n& v) { return fq_is_zero(u) && fq_eq(v, fq_one()); }

// projective equality with an affine point (`JubJubExtended::eq`): X == u Z and Y == v Z
JJS_HD bool ext_eq_affine(const ext_pt& p, const fe_n& u, const fe_n& v) {
    return fq_eq(p.x, fq_mul(u, p.z)) && fq_eq(p.y, fq_mul(v, p.z));
}

}  // namespace jjs

// A point doubling on FOUR adjacent lanes (a quad: DPP quad_perm reaches its lanes without LDS).  A doubling is two rounds of
// four independent products; lane j computes product j of each round and the results are broadcast in the quad, so the
// critical path of a doubling is two products instead of seven, plus 72 register moves.  Used where a call waits for a chain
// of doublings that nothing else can shorten: the bases of a key's window tables (key_tables.h kt_chain_key_quad) and the far
// positions of the latency path's chain lanes (small_batch.h sb_chain_lane_quad).  Same products on the same limbs as
// ext_double (a square is the product of a value with itself, limb for limb), so the results are bit-identical.  Device only.
#pragma once
#include "ed29.h"

namespace jjs {

#if defined(__HIPCC__)
template <int K>
__device__ __forceinline__ fe_n quad_bcast(const fe_n& v) {
    fe_n r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
        r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], K * 0x55, 0xf, 0xf, true);
#else
        r.l[i] = v.l[i];          // host pass of the compiler: never executed
#endif
    }
    return r;
}
// 2P on the quad's four lanes (j = lane within the quad); every lane holds P and gets 2P.  `own` = this lane's product of
// the second round: coordinate j of 2P (X, Y, Z, T).
__device__ __forceinline__ ext_pt ext_double_quad(const ext_pt& p, uint32_t j, fe_n& own) {
    const bool j0 = j == 0, j1 = j == 1, j2 = j == 2, j3 = j == 3;
    const fe_n a = fq_select(j0 || j1, p.x, fq_select(j2, p.y, p.z));       // X X Y Z
    const fe_n b = fq_select(j0, p.y, a);                                    // Y X Y Z
    const fe_n m = fq_mul_hot(a, b);
    const fe_n xy = quad_bcast<0>(m), xx = quad_bcast<1>(m), yy = quad_bcast<2>(m), zz = quad_bcast<3>(m);
    auto e = fq_dbl(xy);                           // 2XY             <2,4>
    auto c2 = fq_dbl(zz);                          // 2Z^2            <2,4>
    auto g = fq_norm(fq_add(yy, xx));          // <1,4>
    auto h = fq_sub(yy, xx);                   // <3,5>
    auto f = fq_norm(fq_sub(fq_add(c2, xx), yy));   // <1,9>
    // X3 = f e, Y3 = g h, Z3 = f h, T3 = g e: the first factors are f or g (normalised, < 9q), the second e or h
    const fe<1, 9> a2 = fq_select(j0 || j2, f, fq_as<1, 9>(g));
    const fe<3, 5> b2 = fq_select(j0 || j3, fq_as<3, 5>(e), h);
    own = fq_mul_hot(a2, b2);
    ext_pt r;
    r.x = quad_bcast<0>(own); r.y = quad_bcast<1>(own); r.z = quad_bcast<2>(own); r.t = quad_bcast<3>(own);
    return r;
}
#endif

}  // namespace jjs

// Device-vs-host self check of the arithmetic layers: the SAME __host__ __device__ source
// (csrc/*.h) is run on the CPU (hipcc's host pass) and on the GPU over identical random inputs and
// the raw limbs are compared stage by stage.  Localises a miscompile or an undefined-behaviour
// difference to one layer.  (The host pass is itself checked against the oracle in
// tests/test_hostbuild.py.)
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../csrc -o devcheck devcheck.hip ; run: ./devcheck
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "verify_core.h"

using namespace jjs;

// x^5 for an arbitrary normalised x (the product's sbox5 is specialised to lane + round constant)
template <int A>
JJS_HD fe_n sbox5_any(const fe<1, A>& x) {
    fe_n x2 = fq_sqr(x);
    return fq_mul(fq_sqr(x2), x);
}


#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

constexpr int N = 512;
constexpr int OUT_WORDS = 64;   // per item per stage

enum Stage { S_MUL, S_SQR, S_ADDSUB, S_REDUCE, S_DOT5, S_SBOX, S_ROUND_FULL, S_ROUND_PARTIAL, S_PERMUTE, S_DOUBLE, S_ADD,
             S_ONCURVE, S_TORSION, S_CANON, S_EUCLID, S_INVERSE, N_STAGES };
static const char* kStageNames[] = {"mul", "sqr", "add/sub/norm", "reduce", "dot5", "sbox", "hades round (full)",
                                    "hades round (partial)", "hades permute", "ext_double", "ext_add_niels", "on_curve",
                                    "torsion", "canon/to_words", "half-size scalars (Euclid)", "inverse (division steps / power)"};

__host__ __device__ inline void put(uint32_t* out, int& pos, const uint32_t* l, int n) { for (int i = 0; i < n; ++i) out[pos++] = l[i]; }

__host__ __device__ inline void one_round(hades_state& st, int rnd) {
    const bool full = (rnd < 4) || (rnd >= 64);
    fe<1, 3> t[5];
    if (full) {
        for (int i = 0; i < 4; ++i) t[i] = fq_as<1, 3>(sbox5_any(fq_norm(fq_add(st.s[i], fe_from_const<1, 1>(JJS_RC[5 * rnd + i])))));
    } else {
        for (int i = 0; i < 4; ++i) t[i] = fq_norm(fq_add(st.s[i], fe_from_const<1, 1>(JJS_RC[5 * rnd + i])));
    }
    t[4] = fq_as<1, 3>(sbox5_any(fq_norm(fq_add(st.s[4], fe_from_const<1, 1>(JJS_RC[5 * rnd + 4])))));
    for (int i = 0; i < 5; ++i) st.s[i] = fq_dot_const<5, 3>(JJS_MDS[i], t);
}

// in: 5 field elements (canonical words) per item
__host__ __device__ inline void run_stage(int stage, const uint32_t* in, uint32_t* out) {
    words8 w[5];
    for (int j = 0; j < 5; ++j) for (int i = 0; i < 8; ++i) w[j].w[i] = in[8 * j + i];
    fe_n a = fq_from_words(w[0]), b = fq_from_words(w[1]), c = fq_from_words(w[2]), d = fq_from_words(w[3]), e = fq_from_words(w[4]);
    int pos = 0;
    for (int i = 0; i < OUT_WORDS; ++i) out[i] = 0;
    switch (stage) {
    case S_MUL: { fe_n r = fq_mul(a, b); put(out, pos, r.l, 9); break; }
    case S_SQR: { fe_n r = fq_sqr(a); put(out, pos, r.l, 9); fe_n r2 = fq_sqr(fq_norm(fq_add(a, b))); put(out, pos, r2.l, 9); break; }
    case S_ADDSUB: {
        auto s = fq_norm(fq_add(a, b)); put(out, pos, s.l, 9);
        auto t = fq_norm(fq_sub(a, b)); put(out, pos, t.l, 9);
        auto u = fq_norm(fq_neg(a)); put(out, pos, u.l, 9);
        auto v = fq_norm(fq_dbl(a)); put(out, pos, v.l, 9);
        break; }
    case S_REDUCE: { fe_n r = fq_reduce(fq_norm(fq_add(a, b))); put(out, pos, r.l, 9); break; }
    case S_DOT5: {
        fe<1, 3> t[5] = {fq_as<1, 3>(a), fq_as<1, 3>(b), fq_as<1, 3>(c), fq_as<1, 3>(d), fq_as<1, 3>(e)};
        for (int i = 0; i < 5; ++i) { fe_n r = fq_dot_const<5, 3>(JJS_MDS[i], t); put(out, pos, r.l, 9); }
        break; }
    case S_SBOX: { fe_n r = sbox5_any(fq_norm(fq_add(a, fe_from_const<1, 1>(JJS_RC[7])))); put(out, pos, r.l, 9); break; }
    case S_ROUND_FULL: case S_ROUND_PARTIAL: {
        hades_state st; st.s[0] = a; st.s[1] = b; st.s[2] = c; st.s[3] = d; st.s[4] = e;
        one_round(st, stage == S_ROUND_FULL ? 1 : 10);
        for (int i = 0; i < 5; ++i) put(out, pos, st.s[i].l, 9);
        break; }
    case S_PERMUTE: {
        hades_state st; st.s[0] = a; st.s[1] = b; st.s[2] = c; st.s[3] = d; st.s[4] = e;
        hades_permute(st);
        for (int i = 0; i < 5; ++i) put(out, pos, st.s[i].l, 9);
        break; }
    case S_DOUBLE: {
        ext_pt p; p.x = a; p.y = b; p.z = c; p.t = d;
        ext_pt r = ext_double(p, true);
        put(out, pos, r.x.l, 9); put(out, pos, r.y.l, 9); put(out, pos, r.z.l, 9); put(out, pos, r.t.l, 9);
        break; }
    case S_ADD: {
        ext_pt p; p.x = a; p.y = b; p.z = c; p.t = d;
        ext_pt q; q.x = b; q.y = e; q.z = a; q.t = c;
        ext_pt r = ext_add_niels(p, to_niels(q), (in[0] & 1) != 0, true);
        put(out, pos, r.x.l, 9); put(out, pos, r.y.l, 9); put(out, pos, r.z.l, 9); put(out, pos, r.t.l, 9);
        break; }
    case S_ONCURVE: { out[0] = affine_on_curve(a, b); out[1] = affine_is_identity(a, b); out[2] = fq_is_zero(fq_sub(a, a)); break; }
    case S_TORSION: { out[0] = is_torsion_free(a, b); break; }
    case S_CANON: { words8 r = fq_to_words(fq_add(fq_add(a, b), c)); put(out, pos, r.w, 8); break; }
    case S_EUCLID: {
        // the device takes v_rcp_f64 estimates and wave ballots, the host 1.0 / y and per-lane control: same (a, b)
        words8 cw = w[0];
        cw.w[7] &= 0x03ffffffu;                                   // a challenge: 250 bits
        half_scalars h = half_size_scalars(cw);
        put(out, pos, h.a.w, 4); put(out, pos, h.b.w, 4); out[pos++] = h.b_neg ? 1u : 0u;
        break; }
    case S_INVERSE: {
        // csrc/fq_inv.h against the power it replaced, on the device as on the host; and a * (1/a) == 1
        const words8 r = fq_to_words(fq_inverse(a)), p = fq_to_words(fq_inverse_by_power(a)), one = fq_to_words(fq_mul(a, fq_inverse(a)));
        put(out, pos, r.w, 8); put(out, pos, p.w, 8); put(out, pos, one.w, 8);
        uint32_t diff = 0;
        for (int i = 0; i < 8; ++i) diff |= r.w[i] ^ p.w[i];
        out[pos++] = diff ? 0xBADu : 0u;
        break; }
    }
}

__global__ void dev_stage(int stage, const uint32_t* in, uint32_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) run_stage(stage, in + 40 * i, out + OUT_WORDS * i);
}

int main() {
    std::vector<uint32_t> in(40 * N);
    uint64_t s = 0x6a6a73;
    for (auto& x : in) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; x = (uint32_t)(s >> 32); }
    for (int i = 0; i < N; ++i) for (int j = 0; j < 5; ++j) in[40 * i + 8 * j + 7] &= 0x3fffffffu;   // < 2^254 < q
    for (int j = 0; j < 5; ++j) for (int k = 0; k < 8; ++k) in[8 * j + k] = JJS_Q_WORDS[k] - (k == 0 ? 1 + j : 0);   // q-1-j
    for (int k = 0; k < 40; ++k) in[40 + k] = 0;                                                              // zeros
    // item 2: the generator (on curve, torsion free)
    {
        fe_n gu = fq_as<1, 2>(fe_from_const<1, 1>(JJS_G[0])), gv = fq_as<1, 2>(fe_from_const<1, 1>(JJS_G[1]));
        words8 a = fq_to_words(gu), b = fq_to_words(gv);
        memcpy(&in[80], a.w, 32); memcpy(&in[88], b.w, 32);
    }
    // items 3..10: challenges with huge first quotients (r >> k) and remainders next to 2^126 (the Euclid stage)
    {
        const int shifts[6] = {26, 31, 32, 63, 120, 125};
        for (int t = 0; t < 6; ++t) {
            uint32_t x[8];
            for (int k = 0; k < 8; ++k) x[k] = JJS_FR_WORDS[k];
            for (int sft = 0; sft < shifts[t]; ++sft) {                  // x >>= 1
                for (int k = 0; k < 8; ++k) x[k] = (x[k] >> 1) | (k < 7 ? x[k + 1] << 31 : 0u);
            }
            x[0] += (t & 1);
            memcpy(&in[40 * (3 + t)], x, 32);
        }
        uint32_t p126[8] = {0, 0, 0, 0x40000000u, 0, 0, 0, 0}, m126[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu, 0, 0, 0, 0};
        memcpy(&in[40 * 9], p126, 32); memcpy(&in[40 * 10], m126, 32);
    }
    uint32_t *din, *dout;
    CHECK(hipMalloc(&din, in.size() * 4)); CHECK(hipMalloc(&dout, (size_t)OUT_WORDS * N * 4));
    CHECK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> host_out(OUT_WORDS * N), dev_out(OUT_WORDS * N);
    int bad_stages = 0;
    for (int st = 0; st < N_STAGES; ++st) {
        int n_host = (st == S_TORSION) ? 8 : (st == S_PERMUTE ? 64 : N);
        for (int i = 0; i < n_host; ++i) run_stage(st, &in[40 * i], &host_out[OUT_WORDS * i]);
        hipLaunchKernelGGL(dev_stage, dim3((N + 63) / 64), dim3(64), 0, 0, st, din, dout);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(dev_out.data(), dout, dev_out.size() * 4, hipMemcpyDeviceToHost));
        int mism = 0, first = -1;
        for (int i = 0; i < n_host; ++i)
            if (memcmp(&host_out[OUT_WORDS * i], &dev_out[OUT_WORDS * i], OUT_WORDS * 4) ||
                (st == S_INVERSE && dev_out[OUT_WORDS * i + 24] != 0)) { if (first < 0) first = i; ++mism; }      // [24]: the two inversions differ
        printf("stage %-24s items %4d mismatches %4d%s\n", kStageNames[st], n_host, mism, mism ? "  <-- DIFFERS" : "");
        if (mism) {
            ++bad_stages;
            printf("  first differing item %d\n  host:", first);
            for (int k = 0; k < 18; ++k) printf(" %08x", host_out[OUT_WORDS * first + k]);
            printf("\n  dev :");
            for (int k = 0; k < 18; ++k) printf(" %08x", dev_out[OUT_WORDS * first + k]);
            printf("\n");
        }
    }
    printf(bad_stages ? "DEVCHECK FAILED (%d stages)\n" : "DEVCHECK OK\n", bad_stages);
    return bad_stages ? 1 : 0;
}

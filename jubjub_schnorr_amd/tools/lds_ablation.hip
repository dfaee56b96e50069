// Row (g) of the scope table: BASELINE.json's north_star names "LDS staging of the Poseidon MDS/round constants".
// The product reads those constants at wave-uniform addresses through the scalar cache, as SGPR operands of the
// multiply-adds (csrc/hades29.h).  This program measures the alternative on the same permutation code: the
// 1 440 words of round constants (kappa, mu, full-round constants) copied into LDS at kernel start and read
// from there (uniform address: a broadcast ds_read per word).  Same arithmetic, same results (compared);
// one JSON line with both times.  The 9 distinct entries of the small-integer matrix stay scalar in both variants:
// they are operands of one asm block and fit the SGPR file outright.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../csrc -o lds_ablation lds_ablation.hip ; run: ./lds_ablation
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "hades29.h"

using namespace jjs;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

constexpr int N_KAPPA = 60 * 9, N_MU = 60 * 9, N_FULL = 8 * 5 * 9, N_WORDS = N_KAPPA + N_MU + N_FULL;

template <bool LDS>
__device__ __forceinline__ void permute(hades_state& st, const uint32_t* kappa, const uint32_t* mu, const uint32_t* full) {
    for (int r = 0; r < 68; ++r) {
        fe_n t[5];
        if (r < 4 || r >= 64) {
            const int fr = r < 4 ? r : r - 60;
#pragma unroll
            for (int i = 0; i < 5; ++i) t[i] = sbox5(fq_add(st.s[i], fe_from_const<1, 1>(full + (fr * 5 + i) * 9)));
        } else {
            const int k = r - 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = st.s[i];
            t[4] = fq_mul(sbox5(fq_add(st.s[4], fe_from_const<1, 1>(kappa + 9 * k))), fe_from_const<1, 1>(mu + 9 * k));
        }
        hades_matrix(st, t);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) st.s[i] = fq_mul(st.s[i], fe_from_const<1, 1>(JJS_HS_LAMBDA_END));
}

template <bool LDS>
__global__ __launch_bounds__(256, 4) void hades_kernel(uint32_t* out, int perms) {
    __shared__ uint32_t staged[N_WORDS];
    const uint32_t *kappa = &JJS_HS_KAPPA[0][0], *mu = &JJS_HS_MU[0][0], *full = &JJS_HS_RC_FULL[0][0][0];
    if (LDS) {
        for (int i = threadIdx.x; i < N_KAPPA; i += 256) { staged[i] = kappa[i]; staged[N_KAPPA + i] = mu[i]; }
        for (int i = threadIdx.x; i < N_FULL; i += 256) staged[N_KAPPA + N_MU + i] = full[i];
        __syncthreads();
    }
    hades_state st;
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        fe_c x = fq_zero();
        x.l[0] = (tid * 2654435761u + i) & MASK29; x.l[3] = tid & MASK29; x.l[7] = (tid ^ 0x5a5a5au) & 0xfffffu;
        st.s[i] = fq_mul(x, fe_from_const<1, 1>(JJS_R2));
    }
    for (int p = 0; p < perms; ++p) {
        if (LDS) permute<true>(st, staged, staged + N_KAPPA, staged + N_KAPPA + N_MU);
        else permute<false>(st, kappa, mu, full);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc = acc * 31u + fq_canon_limbs(st.s[i]).l[j];
    out[tid] = acc;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 4 * 4, perms = 8;       // four resident blocks per CU, four rounds of them
    const size_t n = (size_t)blocks * 256;
    uint32_t *a, *b;
    CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 4; ++rep) {
        for (int v = 0; v < 2; ++v) {
            CHECK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(hades_kernel<false>, dim3(blocks), dim3(256), 0, 0, a, perms);
            else hipLaunchKernelGGL(hades_kernel<true>, dim3(blocks), dim3(256), 0, 0, b, perms);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float t; CHECK(hipEventElapsedTime(&t, e0, e1));
            if (rep && t < ms[v]) ms[v] = t;
        }
    }
    std::vector<uint32_t> ha(n), hb(n);
    CHECK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(ha.data(), hb.data(), n * 4) == 0;
    printf("{\"what\": \"Hades permutation, round constants from the scalar cache (product) vs staged in LDS\", \"lanes\": %zu, "
           "\"permutations_per_lane\": %d, \"ms_scalar_cache\": %.4f, \"ms_lds\": %.4f, \"lds_over_scalar\": %.4f, "
           "\"lds_bytes_per_block\": %d, \"results_identical\": %s}\n",
           n, perms, ms[0], ms[1], ms[1] / ms[0], N_WORDS * 4, same ? "true" : "false");
    return same ? 0 : 1;
}

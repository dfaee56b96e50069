// Integer-multiply issue-rate microbenchmark for gfx950 (SURVEY.md section 7 "hard parts"):
// the verify path is bound by 32-bit multiply issue, and the microarchitecture guide gives no
// integer-multiply rates.  Each kernel runs 8 independent dependency chains of one instruction
// per lane; the host reports wave-instructions/s chip-wide and cycles per wave-instruction per
// SIMD at the nominal 2.4 GHz clock.
//
// build: hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip ; run: ./microbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, FMA_F64, ADD_U32, LSHL_ADD_U64, ADDC_U32,
          MUL_F64, MAD_U64_U32_SGPR, ADD3_U32, MAD_I64_I32, ASHR_I64, LSHR_B64, AND_B32, ALIGNBIT_B32, N_OPS };
static const char* kNames[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
                               "v_fma_f64", "v_add_u32", "v_lshl_add_u64", "v_addc_co_u32", "v_mul_f64",
                               "v_mad_u64_u32(sgpr b)", "v_add3_u32", "v_mad_i64_i32(sgpr b)", "v_ashrrev_i64", "v_lshrrev_b64",
                               "v_and_b32", "v_alignbit_b32"};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void ubench(uint64_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x * 0x9e3779b9u + seed, b = (threadIdx.x ^ seed) * 0x85ebca6bu + 12345u;
    uint64_t acc[8];
    double dacc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = a + i; dacc[i] = 1.0 + i * 1e-3; }
    double da = 1.0000001, db = 1e-9;
    uint32_t sb = __builtin_amdgcn_readfirstlane(b);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (OP == MAD_U64_U32) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
                REP8(X)
#undef X
            } else if constexpr (OP == MAD_U64_U32_SGPR) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(sb) : "vcc");
                REP8(X)
#undef X
            } else if constexpr (OP == MUL_LO_U32) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t) : "v"(b)); acc[i] = t; }
                REP8(X)
#undef X
            } else if constexpr (OP == MUL_HI_U32) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t) : "v"(b)); acc[i] = t; }
                REP8(X)
#undef X
            } else if constexpr (OP == MAD_U32_U24) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(t) : "v"(a), "v"(b)); acc[i] = t; }
                REP8(X)
#undef X
            } else if constexpr (OP == MUL_HI_U32_U24) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(t) : "v"(b)); acc[i] = t; }
                REP8(X)
#undef X
            } else if constexpr (OP == FMA_F64) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(dacc[i]) : "v"(da), "v"(db));
                REP8(X)
#undef X
            } else if constexpr (OP == MUL_F64) {
#define X(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(dacc[i]) : "v"(da));
                REP8(X)
#undef X
            } else if constexpr (OP == ADD_U32) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(b)); acc[i] = t; }
                REP8(X)
#undef X
            } else if constexpr (OP == ADD3_U32) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(t) : "v"(b), "v"(a)); acc[i] = t; }
                REP8(X)
#undef X
            } else if constexpr (OP == LSHL_ADD_U64) {
                uint64_t bb = ((uint64_t)a << 32) | b;
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(bb));
                REP8(X)
#undef X
            } else if constexpr (OP == MAD_I64_I32) {
#define X(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(sb) : "vcc");
                REP8(X)
#undef X
            } else if constexpr (OP == ASHR_I64) {
#define X(i) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(acc[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == LSHR_B64) {
#define X(i) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == AND_B32) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_and_b32 %0, %1, %0" : "+v"(t) : "v"(b)); acc[i] = t; }
                REP8(X)
#undef X
            } else if constexpr (OP == ALIGNBIT_B32) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(t) : "v"(b)); acc[i] = t; }
                REP8(X)
#undef X
            } else if constexpr (OP == ADDC_U32) {
#define X(i) { uint32_t t = (uint32_t)acc[i]; asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(t) : "v"(b) : "vcc"); acc[i] = t; }
                REP8(X)
#undef X
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i] + (uint64_t)dacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static int run(uint64_t* dout, int waves_per_simd, int iters) {
    int grid = 256 * waves_per_simd;  // 256 CUs x (4 waves = one per SIMD) per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(ubench<OP>, dim3(grid), dim3(256), 0, 0, dout, iters / 10, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(ubench<OP>, dim3(grid), dim3(256), 0, 0, dout, iters, 2u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double wave_instr = (double)grid * 4 * (double)iters * 32;  // 4 waves/block, 4x8 instr per iteration
    double per_s = wave_instr / (best * 1e-3);
    double cyc = 1024.0 * 2.4e9 / per_s;  // cycles per wave-instruction per SIMD at 2.4 GHz
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_instr_per_s\": %.4e, \"lane_ops_per_s\": %.4e, \"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.2f}\n",
           kNames[OP], waves_per_simd, best, per_s, per_s * 64, cyc);
    return 0;
}

int main() {
    uint64_t* dout;
    CHECK(hipMalloc(&dout, sizeof(uint64_t) * 256 * 8 * 256));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        if (run<MAD_U64_U32>(dout, w, iters)) return 1;
        if (run<MAD_U64_U32_SGPR>(dout, w, iters)) return 1;
        if (run<MUL_LO_U32>(dout, w, iters)) return 1;
        if (run<MUL_HI_U32>(dout, w, iters)) return 1;
        if (run<MAD_U32_U24>(dout, w, iters)) return 1;
        if (run<MUL_HI_U32_U24>(dout, w, iters)) return 1;
        if (run<FMA_F64>(dout, w, iters)) return 1;
        if (run<MUL_F64>(dout, w, iters)) return 1;
        if (run<ADD_U32>(dout, w, iters)) return 1;
        if (run<ADD3_U32>(dout, w, iters)) return 1;
        if (run<LSHL_ADD_U64>(dout, w, iters)) return 1;
        if (run<ADDC_U32>(dout, w, iters)) return 1;
        if (run<MAD_I64_I32>(dout, w, iters)) return 1;
        if (run<ASHR_I64>(dout, w, iters)) return 1;
        if (run<LSHR_B64>(dout, w, iters)) return 1;
        if (run<AND_B32>(dout, w, iters)) return 1;
        if (run<ALIGNBIT_B32>(dout, w, iters)) return 1;
    }
    CHECK(hipFree(dout));
    return 0;
}

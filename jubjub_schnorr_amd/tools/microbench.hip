// Integer-multiply issue-rate microbenchmark for gfx950 (SURVEY.md section 7 "hard parts"):
// the verify path is bound by 32-bit multiply issue, and the microarchitecture guide gives no
// integer-multiply rates.  Each kernel runs 8 independent dependency chains of one instruction
// per lane; the host reports wave-instructions/s chip-wide and cycles per wave-instruction per
// SIMD, both at the nominal 2.4 GHz clock and at the shader clock the chip HELD while the stage ran
// (librocm_smi64, sampled from a host thread beside the kernel: the figure bench.py's alu_roofline is built on).
// Three kinds of stage:
//   one opcode      8 independent chains of a single instruction: the issue cost of that opcode class;
//   mont-mix        the instruction mix of one column of the product's Montgomery block (csrc/mont_asm.inc):
//                   4 v_mad_u64_u32 (VGPR x VGPR), 3 v_mad_i64_i32 (VGPR x SGPR), v_and_b32, v_ashrrev_i64, v_add_u32,
//                   70 % multiply-adds, on 8 independent accumulators (the real block chains them: this is its floor);
//   fq_mul / fq_sqr the product's own field product and square (csrc/fq29.h), two independent chains per lane.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../csrc -o microbench microbench.hip -L/opt/rocm/lib -lrocm_smi64
// run: ./microbench
#include <hip/hip_runtime.h>
#include <rocm_smi/rocm_smi.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>
#include "fq29.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, FMA_F64, ADD_U32, LSHL_ADD_U64, ADDC_U32,
          MUL_F64, MAD_U64_U32_SGPR, ADD3_U32, MAD_I64_I32, ASHR_I64, LSHR_B64, AND_B32, ALIGNBIT_B32, MONT_MIX, N_OPS };
static const char* kNames[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
                               "v_fma_f64", "v_add_u32", "v_lshl_add_u64", "v_addc_co_u32", "v_mul_f64",
                               "v_mad_u64_u32(sgpr b)", "v_add3_u32", "v_mad_i64_i32(sgpr b)", "v_ashrrev_i64", "v_lshrrev_b64",
                               "v_and_b32", "v_alignbit_b32", "mont-mix (4 mad_u64_u32 + 3 mad_i64_i32(sgpr) + and + ashr64 + add)"};

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
            } else if constexpr (OP == MONT_MIX) {
                // ten instructions per accumulator, seven of them multiply-adds: what one column of mont_asm.inc issues
#define X(i) { uint32_t lo; \
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc"); \
                asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(sb) : "vcc"); \
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(a) : "vcc"); \
                asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "s"(sb) : "vcc"); \
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(a) : "vcc"); \
                asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(sb) : "vcc"); \
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(b) : "vcc"); \
                asm volatile("v_and_b32 %0, 0x1fffffff, %1" : "=v"(lo) : "v"((uint32_t)acc[i])); \
                asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(acc[i])); \
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(lo)); }
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

// ---- shader clock, sampled beside a kernel ------------------------------------------------------------------
static bool g_smi = false;
static double read_sclk_mhz() {
    if (!g_smi) return 0;
    rsmi_frequencies_t f;
    if (rsmi_dev_gpu_clk_freq_get(0, RSMI_CLK_TYPE_SYS, &f) != RSMI_STATUS_SUCCESS || f.current >= RSMI_MAX_NUM_FREQUENCIES) return 0;
    return (double)f.frequency[f.current] / 1e6;
}
struct clock_sampler {          // median of the samples taken between start() and stop()
    std::atomic<bool> go{false};
    std::vector<double> v;
    std::thread t;
    void start() {
        v.clear();
        go = true;
        t = std::thread([this] { while (go) { double s = read_sclk_mhz(); if (s > 0) v.push_back(s); std::this_thread::sleep_for(std::chrono::microseconds(200)); } });
    }
    double stop() {
        go = false;
        t.join();
        if (v.empty()) return 0;
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    }
};

static void report(const char* name, int waves_per_simd, float best_ms, double wave_instr, double sclk_mhz, int samples) {
    double per_s = wave_instr / (best_ms * 1e-3);
    double cyc = 1024.0 * 2.4e9 / per_s;  // cycles per wave-instruction per SIMD at 2.4 GHz
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_instr_per_s\": %.4e, \"lane_ops_per_s\": %.4e, "
           "\"cycles_per_wave_instr_per_simd_at_2.4GHz\": %.2f, \"sclk_mhz\": %.0f, \"sclk_samples\": %d, "
           "\"cycles_per_wave_instr_per_simd_at_sclk\": %.3f}\n",
           name, waves_per_simd, best_ms, per_s, per_s * 64, cyc, sclk_mhz, samples, sclk_mhz > 0 ? 1024.0 * sclk_mhz * 1e6 / per_s : 0.0);
    fflush(stdout);
}

template <int OP>
static int run(uint64_t* dout, int waves_per_simd, int iters) {
    int grid = 256 * waves_per_simd;  // 256 CUs x (4 waves = one per SIMD) per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(ubench<OP>, dim3(grid), dim3(256), 0, 0, dout, iters / 10, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    clock_sampler cs;
    cs.start();
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(ubench<OP>, dim3(grid), dim3(256), 0, 0, dout, iters, 2u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double sclk = cs.stop();
    const double per_iter = OP == MONT_MIX ? 4.0 * 8 * 10 : 32.0;   // wave-instructions per iteration
    report(kNames[OP], waves_per_simd, best, (double)grid * 4 * (double)iters * per_iter, sclk, (int)cs.v.size());
    return 0;
}

// The product's own field product / square: two independent chains per lane, values kept live through the output.
template <bool SQUARE>
__global__ __launch_bounds__(256) void fq_bench(uint32_t* out, int iters, uint32_t seed) {
    jjs::fe_n x, y, z;
#pragma unroll
    for (int i = 0; i < 9; ++i) { x.l[i] = (threadIdx.x * 2654435761u + seed + i) & 0x1fffffffu; y.l[i] = (x.l[i] * 40503u + 7u) & 0x1fffffffu; z.l[i] = (y.l[i] ^ 0x155555u) & 0x1fffffffu; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (SQUARE) { x = jjs::fq_sqr(x); y = jjs::fq_sqr(y); }
        else { x = jjs::fq_mul(x, z); y = jjs::fq_mul(y, z); }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) s += x.l[i] ^ y.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool SQUARE>
static int run_fq(uint32_t* dout, int waves_per_simd, int iters, int instr_per_op) {
    int grid = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fq_bench<SQUARE>, dim3(grid), dim3(256), 0, 0, dout, iters / 10, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    clock_sampler cs;
    cs.start();
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(fq_bench<SQUARE>, dim3(grid), dim3(256), 0, 0, dout, iters, 2u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double sclk = cs.stop();
    // instr_per_op: VALU instructions of one asm block (csrc/mont_asm.inc header comments); the loop adds nothing vector
    report(SQUARE ? "fq_sqr (csrc/mont_asm.inc: 161 VALU instructions, 117 multiply-adds)" : "fq_mul (csrc/mont_asm.inc: 189 VALU instructions, 153 multiply-adds)", waves_per_simd, best,
           (double)grid * 4 * (double)iters * 2 * instr_per_op, sclk, (int)cs.v.size());
    return 0;
}

int main() {
    g_smi = rsmi_init(0) == RSMI_STATUS_SUCCESS;
    uint64_t* dout;
    CHECK(hipMalloc(&dout, sizeof(uint64_t) * 256 * 8 * 256));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        if (run<MONT_MIX>(dout, w, iters / 8)) return 1;
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
    for (int w : {1, 2, 4, 8}) {
        if (run_fq<false>(reinterpret_cast<uint32_t*>(dout), w, 4000, 189)) return 1;
        if (run_fq<true>(reinterpret_cast<uint32_t*>(dout), w, 4000, 161)) return 1;
    }
    CHECK(hipFree(dout));
    return 0;
}
